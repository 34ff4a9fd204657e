"""Where the 3x3 conv family's time goes: one inference step with every conv launch timed by HIP events, grouped by shape."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from glare_amd import ops

dev = torch.device("cuda", 0)
netG, net_vq = bench.build_nets(dev)
lr = bench.build_inputs(8, dev)
recs = []
orig_conv2d = ops.conv2d


def conv2d(x, pc, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = orig_conv2d(x, pc, **kw)
    e.record()
    B, H, W, _ = x.shape
    st, up = kw.get("stride", 1), kw.get("upsample", False)
    OH, OW = (H // 2, W // 2) if st == 2 else ((2 * H, 2 * W) if up else (H, W))
    key = (pc.ksize, B, OH, OW, pc.cin, pc.cout, st, int(bool(up)), int(kw.get("residual") is not None), int(bool(kw.get("hilo"))),
           int(bool(kw.get("gn_stats"))), kw.get("out_mode", 0), kw.get("act", "none"))
    recs.append((key, s, e, 2.0 * B * OH * OW * pc.ksize ** 2 * pc.cin * pc.cout))
    return out


with torch.no_grad():
    for _ in range(3):
        netG.reverse_flow_nhwc(net_vq, lr)
    torch.cuda.synchronize()
    ops.conv2d = conv2d
    for it in range(5):
        netG.reverse_flow_nhwc(net_vq, lr)
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, s, e, fl in recs:
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += s.elapsed_time(e)
    a[2] += fl
print("k  B   OH   OW  cin cout st up res hilo gn mode act      | n/step  ms/launch  ms/step  TFLOP/s  frac")
tot = 0.0
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms / 5
    print("%d %2d %4d %4d %4d %4d  %d  %d  %d   %d    %d  %d   %-8s | %5.1f  %8.3f  %7.3f  %7.1f  %.3f" %
          (key + (n / 5, ms / n, ms / 5, fl / ms * 1e-9, fl / ms * 1e-9 / 2500)))
print("total ms/step", tot)
