import os, sys
sys.path.insert(0, "/root/repo")
import torch
from glare_amd import ops
dev = torch.device("cuda", 0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for name, B, H, W, ci, co in [("128->128 @256 B1", 1, 256, 256, 128, 128), ("256->256 @128 B1", 1, 128, 128, 256, 256), ("512->512 @64 B1", 1, 64, 64, 512, 512),
                              ("128->128 @320 B2", 2, 320, 320, 128, 128), ("256->256 @160 B2", 2, 160, 160, 256, 256), ("512->512 @80 B2", 2, 80, 80, 512, 512)]:
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    pc = ops.PackedConv(w, torch.zeros(co, device=dev))
    ms = timeit(lambda: ops.conv2d(x, pc))
    print("conv %-18s TN=%s: %.4f ms  %.0f TFLOP/s" % (name, os.environ.get("GLARE_FORCE_TN", "128"), ms, 2.0 * B * H * W * 9 * ci * co / ms / 1e9))
