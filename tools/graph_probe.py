"""Probe: B=1 latency of the fused inference graph, eager launches vs a captured hipGraph (torch.cuda.CUDAGraph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
netG, net_vq = bench.build_nets(dev)
lr = bench.build_inputs(B, dev)
def step():
    return netG.reverse_flow_nhwc(net_vq, lr)["out"]
with torch.no_grad():
    for _ in range(3):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10
    print("eager   B=%d: %.2f ms/step  %.1f img/s" % (B, eager * 1e3, B / eager))
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                out = step()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            gout = step()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        gr = (time.perf_counter() - t0) / 10
        print("graphed B=%d: %.2f ms/step  %.1f img/s   max|diff| vs eager %.3e" % (B, gr * 1e3, B / gr, float((gout - out).abs().max())))
    except Exception as e:
        print("graph capture failed:", repr(e)[:300])
