import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
netG, net_vq = bench.build_nets(dev)
lr = bench.build_inputs(8, dev)
with torch.no_grad():
    enc = netG.RRDB.forward_nhwc(lr)
    which = sys.argv[1]
    for _ in range(10):
        if which == "flow":
            netG.flowUpsamplerNet.decode_nhwc(enc["color_map"], enc["cond_feat"])
        else:
            netG.RRDB.forward_nhwc(lr)
    torch.cuda.synchronize()
