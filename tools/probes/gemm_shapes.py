"""Lists the gemm_nt / reduce_parts calls of one stage-2 (or stage-3) training step with their shapes and durations."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from glare_amd import modules as M, train_ops as T
from glare_amd.synthetic import seeded_init_
from glare_amd.train import Stage2Trainer, Stage3Trainer
import traceback

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
log = []
def wrap(name, fn):
    def w(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = fn(*a, **k); e.record()
        caller = [f.name for f in traceback.extract_stack()[:-1] if f.name not in ("w",)][-3:]
        if name == "gemm_nt":
            A, Bm = a[0], a[1]
            shape = "A%s B%s K=%s" % (tuple(A.shape), tuple(Bm.shape), k.get("K"))
        else:
            shape = str(tuple(a[0].shape))
        log.append((name, shape, "/".join(caller), s, e))
        return r
    return w
T.gemm_nt = wrap("gemm_nt", T.gemm_nt)
T.reduce_parts = wrap("reduce_parts", T.reduce_parts)
tr.step(gt, lr)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, shape, caller, s, e in log:
    k = (name, shape, caller)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e)
tot = 0
for (name, shape, caller), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-13s x%-3d %7.3f ms  %-60s %s" % (name, n, ms, shape, caller)); tot += ms
print("total %.2f ms in %d calls" % (tot, len(log)))
