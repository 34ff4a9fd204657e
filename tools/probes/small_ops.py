"""Where do the small stock kernels of a training step come from?  One eager step under a TorchDispatchMode; every aten op that
launches a kernel is attributed to the innermost glare_amd source line of the Python stack that issued it (ops issued by built-in
autograd nodes have no Python frame: "<autograd>").   python tools/probes/small_ops.py [stage2|stage3]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

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

VIEWS = {"view", "as_strided", "select", "slice", "reshape", "permute", "unsqueeze", "squeeze", "detach", "alias", "t", "transpose",
         "_unsafe_view", "expand", "narrow", "unflatten", "flatten", "lift_fresh", "empty", "empty_like", "empty_strided", "resize_",
         "unbind", "split", "split_with_sizes", "chunk", "view_as", "_reshape_alias", "new_empty", "new_empty_strided", "is_same_size",
         "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "stride", "size", "numel", "dim", "is_contiguous", "diagonal",
         "movedim", "unfold", "_local_scalar_dense", "record_stream", "set_", "is_pinned", "real", "conj", "_conj", "item", "is_nonzero"}
agg = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        if name not in VIEWS:
            where = "<autograd>"
            for fr in reversed(traceback.extract_stack()):
                if "glare_amd" in fr.filename and "probes" not in fr.filename:
                    where = "%s:%d %s" % (fr.filename.split("glare_amd/")[-1], fr.lineno, fr.name)
                    break
            if where == "<autograd>" and name in ("copy_", "mul", "add") and args and torch.is_tensor(args[0]):
                where = "<autograd> %s %s" % (tuple(args[0].shape), str(args[0].dtype).replace("torch.", ""))
            agg[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    tr.step_tensor(gt, lr)
torch.cuda.synchronize()
tot = sum(agg.values())
print("%d kernel-launching aten ops in one %s step" % (tot, which))
for (name, where), n in agg.most_common(70):
    print("%4d  %-22s %s" % (n, name, where))
