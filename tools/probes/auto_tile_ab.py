"""Inference step with and without the per-launch output-channel tile of the cached filters (ops.AUTO_COUT_TILE) at B = 1, 2, 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from glare_amd import ops
dev = torch.device("cuda", 0)
for B in (1, 2, 8):
    for auto in (False, True):
        ops.AUTO_COUT_TILE = auto
        netG, net_vq = bench.build_nets(dev)
        lr = bench.build_inputs(B, dev)
        with torch.no_grad():
            for _ in range(3):
                out = netG.reverse_flow_nhwc(net_vq, lr)["out"]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                out = netG.reverse_flow_nhwc(net_vq, lr)["out"]
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("B=%d auto_tile=%s: %.2f ms/step  %.1f img/s  checksum %.6f" % (B, auto, dt * 1e3, B / dt, float(out.double().sum())))
