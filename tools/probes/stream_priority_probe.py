"""Two-stream inference steps with equal and with unequal stream priorities (does a high-priority stream whose kernels are dispatched first, the
other filling its tails, beat two equal streams?).  python tools/probes/stream_priority_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from glare_amd import ops

dev = torch.device("cuda", 0)
ops.use_precision("fp16").__enter__()
netG, net_vq = bench.build_nets(dev)
lr = bench.build_inputs(8, dev)


def run(streams, steps=16):
    with torch.no_grad():
        for i in range(4):
            with torch.cuda.stream(streams[i % len(streams)]):
                netG.reverse_flow_nhwc(net_vq, lr)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(steps):
            with torch.cuda.stream(streams[i % len(streams)]):
                netG.reverse_flow_nhwc(net_vq, lr)
        torch.cuda.synchronize()
    return 8 * steps / (time.time() - t0)


with torch.no_grad():
    netG.reverse_flow_nhwc(net_vq, lr)
torch.cuda.synchronize()
for rep in range(3):
    for name, prios in (("one stream", (0,)), ("two equal", (0, 0)), ("high + low", (-1, 0))):
        print("%-12s %.2f images/s" % (name, run([torch.cuda.Stream(dev, priority=p) for p in prios])), flush=True)
