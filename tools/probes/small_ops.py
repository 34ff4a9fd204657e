"""Where do the small stock kernels of a training step come from?  One eager step under torch.profiler with stacks; aten ops that
launch copy / elementwise kernels aggregated by the first glare_amd source line of their stack."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from glare_amd import modules as M
from glare_amd.synthetic import seeded_init_
from glare_amd.train import Stage2Trainer, Stage3Trainer

which = sys.argv[1] if len(sys.argv) > 1 else "stage2"
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(10)
net_hq = seeded_init_(M.VQModel().eval(), 1).to(dev)
if which == "stage2":
    B, S = 2, 320
    tr = Stage2Trainer(seeded_init_(M.LLFlowVQGAN2().train(), 2).to(dev), net_hq)
else:
    B, S = 1, 256
    tr = Stage3Trainer(seeded_init_(M.VQLLFLOWDeformable().train(), 0).to(dev), net_hq)
gt = torch.rand(B, 3, S, S, generator=g).to(dev)
lr = (torch.randn(B, 3, S, S, generator=g) * 0.5 - 1.0).to(dev)
for _ in range(2):
    tr.step(gt, lr)
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.step_tensor(gt, lr)
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::mul", "aten::add", "aten::add_", "aten::mul_", "aten::sum", "aten::cat", "aten::fill_", "aten::zero_",
                   "aten::exp", "aten::sub", "aten::div", "aten::neg", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy"):
        where = "?"
        for fr in ev.stack:
            if "glare_amd" in fr and "site-packages" not in fr:
                where = fr.strip().split("/repo/")[-1]
                break
        agg[(ev.name, where)] += 1
for (name, where), n in agg.most_common(45):
    print("%4d  %-18s %s" % (n, name, where))
