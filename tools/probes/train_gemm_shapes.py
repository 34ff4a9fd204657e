"""Which GEMMs (the im2col + GEMM weight-gradient form, attention fallbacks) a training step still issues, by shape."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from glare_amd import modules as M
from glare_amd import train_ops as T
from glare_amd.synthetic import seeded_init_
from glare_amd.train import Stage2Trainer, Stage3Trainer

which = sys.argv[1] if len(sys.argv) > 1 else "stage3"
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
recs = collections.Counter()
orig_gemm, orig_im2col = T.gemm_nt, T.im2col_t


def gemm_nt(a, b, *args, **kw):
    recs[("gemm_nt", tuple(a.shape), tuple(b.shape))] += 1
    return orig_gemm(a, b, *args, **kw)


def im2col_t(x, ksize, stride=1, *args, **kw):
    recs[("im2col_t", tuple(x.shape), ksize, stride, bool(kw.get("upsample", False)))] += 1
    return orig_im2col(x, ksize, stride, *args, **kw)


T.gemm_nt, T.im2col_t = gemm_nt, im2col_t
tr.step_tensor(gt, lr)
torch.cuda.synchronize()
for k, n in sorted(recs.items(), key=lambda kv: -kv[1]):
    print(n, k)
