"""Times the training steps at the reference's per-GPU shapes (T2 = stage 2, 2 x 3x320x320 per GPU;
T3 = stage 3, 1 x 3x256x256 per GPU).  python tools/train_bench.py [stage2|stage3] [steps] [graph|flops]
flops: one extra eager step with the FLOP counters on -- the algorithmic FLOPs per step of each MFMA kernel family, which
tools/train_roofline.py divides by the rocprofv3 kernel times of the same step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from glare_amd import modules as M
from glare_amd.synthetic import seeded_init_
from glare_amd.train import GraphedStep, Stage2Trainer, Stage3Trainer

which = sys.argv[1] if len(sys.argv) > 1 else "stage2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
flops = len(sys.argv) > 3 and sys.argv[3] == "flops"
dev = torch.device("cuda", 0)
PREC = os.environ.get("TRAIN_PRECISION", "fp16")      # "fp16" (default): the reference's AMP form (loss scaling on the device); "bf16"
g = torch.Generator().manual_seed(10)
net_hq = seeded_init_(M.VQModel().eval(), 1).to(dev)
if which == "stage2":
    B, S = 2, 320
    tr = Stage2Trainer(seeded_init_(M.LLFlowVQGAN2().train(), 2).to(dev), net_hq, device_state=graph, precision=PREC)
else:
    B, S = 1, 256
    tr = Stage3Trainer(seeded_init_(M.VQLLFLOWDeformable().train(), 0).to(dev), net_hq, device_state=graph, precision=PREC)
gt = torch.rand(B, 3, S, S, generator=g).to(dev)
lr = (torch.randn(B, 3, S, S, generator=g) * 0.5 - 1.0).to(dev)
if graph:
    tr = GraphedStep(tr, gt, lr)
for _ in range(2):
    loss = tr.step(gt, lr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = tr.step_tensor(gt, lr)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("%s%s [%s]: B=%d %dx%d  %.1f ms/step  %.2f samples/s  loss %.4f  peak mem %.2f GB"
      % (which, " (hipGraph replay)" if graph else "", PREC, B, S, S, dt * 1e3, B / dt, float(loss), torch.cuda.max_memory_allocated() / 2**30))
if flops:
    from glare_amd import ops
    ops.FLOP_COUNTER = {}
    tr.step_tensor(gt, lr)
    torch.cuda.synchronize()
    for k, v in sorted(ops.FLOP_COUNTER.items()):
        print("flops_per_step %-10s %.4e" % (k, v))
    ops.FLOP_COUNTER = None
