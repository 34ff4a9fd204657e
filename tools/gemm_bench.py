"""python tools/gemm_bench.py -- times glare_gemm_nt_bf16 at the 1x1-conv / wgrad / attention-backward shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glare_amd import train_ops as T

dev = torch.device("cuda", 0)
def run(M, N, K, batch=1, reps=10, out_dtype=torch.bfloat16):
    a = torch.randn(batch, M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(batch, N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(batch, M, N, dtype=out_dtype, device=dev)
    for _ in range(2):
        T.gemm_nt(a, b, out=c)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        T.gemm_nt(a, b, out=c)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print("gemm_nt M=%7d N=%5d K=%6d batch=%3d : %.3f ms  %6.0f TFLOP/s" % (M, N, K, batch, ms, 2.0 * M * N * K * batch / ms / 1e9))

P = 8 * 105 * 155
for (M, N, K) in [(P, 512, 512), (P, 1024, 512), (P, 1536, 512), (4 * P, 256, 256), (8192, 8192, 8192), (4096, 4096, 4096)]:
    run(M, N, K)
run(6400, 6400, 512); run(6400, 512, 6400)
run(512, 4609, 2176, batch=6, out_dtype=torch.float32)     # split-K wgrad 512->512 3x3 at 80x80x2
run(128, 1153, 3200, batch=64, out_dtype=torch.float32)    # split-K wgrad 128->128 3x3 at 320x320x2
