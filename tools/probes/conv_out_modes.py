import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from glare_amd import ops
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from kbench import timeit
B=8
for name, ci, co, h, w in (("128->108 @full", 128, 108, 420, 620), ("256->108 @half", 256, 108, 210, 310), ("256->128 @full (2 src)", 256, 128, 420, 620), ("512->256 @half (2 src)", 512, 256, 210, 310), ("128->3 @full f32", 128, 3, 420, 620)):
    x = torch.randn(B, h, w, ci, device="cuda").to(torch.bfloat16)
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.03
    b = torch.randn(co, device="cuda") * 0.1
    pc = ops.PackedConv(wt, b)
    fl = 2.0 * B * h * w * ci * co * 9
    plane = (h * w + 63) // 64 * 64
    for mode, kw in (("nhwc a16", {}), ("nhwc f32", {"out_mode": ops.OUT_NHWC_F32}), ("planar f32", {"out_mode": ops.OUT_PLANAR_F32, "plane_pitch": plane})):
        ms = timeit(lambda: ops.conv2d(x, pc, **kw))
        print("conv %-24s %-11s %.3f ms  %.0f TFLOP/s" % (name, mode, ms, fl / ms / 1e9), flush=True)
