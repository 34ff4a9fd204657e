"""Fixed cost per tile of the 3x3 conv kernel: time against Cin at one spatial size and Cout (the slope is the per-stage cost, the
intercept the prologue + epilogue of a tile).   python tools/probes/conv_k_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from glare_amd import ops

B, dev = 8, "cuda"


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for (h, w, co) in ((420, 620, 128), (210, 310, 256), (105, 155, 512)):
    tiles = B * ((h + 7) // 8) * ((w + 31) // 32) * (co // 128)
    rounds = tiles / 768.0
    pts = []
    for ci in (16, 32, 64, 128, 256, 512):
        x = torch.randn(B, h, w, ci, device=dev).to(torch.bfloat16)
        pc = ops.PackedConv(torch.randn(co, ci, 3, 3, device=dev) * 0.02, torch.zeros(co, device=dev))
        out = torch.empty(B, h, w, co, dtype=torch.bfloat16, device=dev)
        res = torch.randn(B, h, w, co, device=dev).to(torch.bfloat16)
        t0 = timeit(lambda: ops.conv2d(x, pc, out=out))
        t1 = timeit(lambda: ops.conv2d(x, pc, out=out, residual=res))
        pts.append((ci // 16 * 3, t0, t1))
        print("%dx%d co=%d ci=%3d: %.3f ms  (+res %.3f)  per round of 768 tiles: %.1f us (+res %.1f)" %
              (h, w, co, ci, t0, t1, t0 / rounds * 1e3, t1 / rounds * 1e3))
    (n0, a0, r0), (n1, a1, r1) = pts[2], pts[-1]
    slope = (a1 - a0) / (n1 - n0) / rounds * 1e3
    print("   -> per B stage %.2f us, intercept %.1f us (+res %.1f us) per round; %d tiles = %.2f rounds" %
          (slope, a0 / rounds * 1e3 - slope * n0, r0 / rounds * 1e3 - (r1 - r0) / (n1 - n0) / rounds * 1e3 * n0, tiles, rounds))
