"""3x3 convs whose launch does not fill the chip (training crops, batch 1): the default 128-wide output-channel tile against
64 / 32 (glare_conv_desc.cout_tile) and what glare_conv2d_cout_tile picks."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from glare_amd import _lib, ops

dev = torch.device("cuda", 0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for name, B, H, W, ci, co in [("128->128 @256 B1", 1, 256, 256, 128, 128), ("256->256 @128 B1", 1, 128, 128, 256, 256),
                              ("512->512 @64 B1", 1, 64, 64, 512, 512), ("128->128 @320 B2", 2, 320, 320, 128, 128),
                              ("256->256 @160 B2", 2, 160, 160, 256, 256), ("512->512 @80 B2", 2, 80, 80, 512, 512)]:
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    bias = torch.zeros(co, device=dev)
    res = []
    for tile in (0, 64, 32):
        pc = ops.PackedConv(w, bias, cout_tile=tile)
        res.append(2.0 * B * H * W * 9 * ci * co / timeit(lambda: ops.conv2d(x, pc)) / 1e9)
    i = ctypes.c_int
    pick = _lib.lib().glare_conv2d_cout_tile(i(B), i(H), i(W), i(co))
    print("conv %-18s TFLOP/s with the 128 / 64 / 32 tile: %4.0f / %4.0f / %4.0f   picked: %d" % (name, *res, pick))
