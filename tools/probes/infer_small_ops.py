"""aten ops (the un-owned launches) issued by one inference step, per stage."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda", 0)
netG, net_vq = bench.build_nets(dev)
lr = bench.build_inputs(8, dev)
with torch.no_grad():
    for _ in range(2):
        netG.reverse_flow_nhwc(net_vq, lr)
    torch.cuda.synchronize()

    def run(name, fn):
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            r = fn()
            torch.cuda.synchronize()
        c = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith("aten::") and ev.name not in
                                ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::select", "aten::slice", "aten::empty_strided",
                                 "aten::reshape", "aten::permute", "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::t", "aten::transpose",
                                 "aten::_unsafe_view", "aten::expand", "aten::narrow", "aten::unflatten", "aten::flatten", "aten::lift_fresh", "aten::resize_"))
        print("%-10s %s" % (name, ", ".join("%s x%d" % (k[6:], v) for k, v in c.most_common(14))))
        return r

    from glare_amd import ops
    ops.use_precision(ops.inference_precision()).__enter__()   # the stages below under the entry point's precision (cached filters)
    enc = run("encoder", lambda: netG.RRDB.forward_nhwc(lr))
    lat = run("flow", lambda: netG.flowUpsamplerNet.decode_nhwc(enc["color_map"], enc["cond_feat"]))
    idx, _, feats = run("vq+dec", lambda: net_vq.decode_nhwc(lat, want_image=False))
    run("aft", lambda: netG.deformable_decoder.forward_nhwc(lat, feats, enc["mid_feat"]))
