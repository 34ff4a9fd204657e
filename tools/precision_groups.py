#!/usr/bin/env python
"""CPU study (test-side tool, imports oracle/): WHICH modules' 16-bit rounding (filters + outputs, fp32 residual stream as in the
product's hi / lo form) the latent error of stages A + B comes from.  Baseline = everything rounded to fp16; then one group of
modules at a time is left in fp32.      python tools/precision_groups.py [h w]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from glare_amd.synthetic import representative_init_, synthetic_pair  # noqa: E402
from oracle import torch_ref as O  # noqa: E402
from precision_study import Rounding  # noqa: E402

h = int(sys.argv[1]) if len(sys.argv) > 2 else 100
w = int(sys.argv[2]) if len(sys.argv) > 2 else 156
torch.set_num_threads(min(32, os.cpu_count() or 1))
og, ov = representative_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), O.VQModel().eval(), 0)
names = [n for n, m in og.named_modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.GroupNorm)) and (n.startswith("RRDB") or n.startswith("flowUpsamplerNet"))]
print(len(names), "rounded modules; first / last:", names[:3], names[-3:])
groups = {}
for n in names:
    parts = n.split(".")
    if n.startswith("flowUpsamplerNet"):
        key = "flow"
    elif "down" in parts:
        key = "enc.down.%s" % parts[parts.index("down") + 1]
    elif "mid" in parts:
        key = "enc.mid"
    else:
        key = "enc." + ".".join(parts[1:3])
    groups.setdefault(key, []).append(n)
for k, v in groups.items():
    print("  group %-22s %3d modules" % (k, len(v)))
seeds = (123, 124, 125)
lrs = [O.preprocess(synthetic_pair(1, h, w, seed=s)[0][0]) for s in seeds]
refs = []
with torch.no_grad():
    for lr in lrs:
        refs.append(og.stages(ov, lr))


def measure(sel):
    errs, agr = [], []
    for lr, ref in zip(lrs, refs):
        with torch.no_grad(), Rounding(og, torch.float16, sel, True):
            enc = og.RRDB(lr, mid_feat=True)
            x, _ = og.flowUpsamplerNet.decode(enc["color_map"], enc["cond_feat"])
        with torch.no_grad():
            ov.decode(x)
        lat = ref["latent"]
        errs.append(float((x - lat).norm() / lat.norm()))
        agr.append(float((ov.last_indices == ref["indices"]).float().mean()))
    return sum(errs) / len(errs), sum(agr) / len(agr)


inAB = lambda n: n.startswith("RRDB") or n.startswith("flowUpsamplerNet")
e0, a0 = measure(inAB)
print("all rounded (fp16 operands, fp32 stream): latent rel %.3e  idx agree %.4f" % (e0, a0), flush=True)
for k, v in groups.items():
    vs = set(v)
    e, a = measure(lambda n: inAB(n) and n not in vs)
    print("  fp32 in %-22s latent rel %.3e (%.0f %% of the squared error)  idx agree %.4f" % (k, e, 100 * (1 - (e / e0) ** 2), a), flush=True)
