"""Where the HOST time of an eager training step goes (the stage-3 step is paced by the host: ~1 400 launches of ~14 us): cProfile over 20
steps, top functions by own time.   python tools/probes/train_host_profile.py [stage3|stage2]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from glare_amd import modules as M
from glare_amd.synthetic import seeded_init_
from glare_amd.train import Stage2Trainer, Stage3Trainer

stage = sys.argv[1] if len(sys.argv) > 1 else "stage3"
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(10)
hq = seeded_init_(M.VQModel().eval(), 1).to(dev)
if stage == "stage2":
    B, S = 2, 320
    tr = Stage2Trainer(seeded_init_(M.LLFlowVQGAN2().train(), 2).to(dev), hq, precision="fp16")
else:
    B, S = 1, 256
    tr = Stage3Trainer(seeded_init_(M.VQLLFLOWDeformable().train(), 0).to(dev), hq, precision="fp16")
gt = torch.rand(B, 3, S, S, generator=g).to(dev)
lr = (torch.randn(B, 3, S, S, generator=g) * 0.5 - 1.0).to(dev)
for _ in range(3):
    tr.step_tensor(gt, lr)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.step_tensor(gt, lr)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
