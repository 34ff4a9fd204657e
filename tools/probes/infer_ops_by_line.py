"""Which stock torch ops does one INFERENCE step issue, and from which glare_amd source line?  (TorchDispatchMode, as small_ops.py does for
the training steps.)   python tools/probes/infer_ops_by_line.py"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench

dev = torch.device("cuda", 0)
netG, net_vq = bench.build_nets(dev)
lr = bench.build_inputs(8, dev)
VIEWS = {"view", "as_strided", "select", "slice", "reshape", "permute", "unsqueeze", "squeeze", "detach", "alias", "t", "transpose", "_unsafe_view",
         "expand", "narrow", "unflatten", "flatten", "lift_fresh", "empty", "empty_like", "empty_strided", "resize_", "unbind", "split",
         "split_with_sizes", "chunk", "view_as", "_reshape_alias", "new_empty", "new_empty_strided", "is_same_size", "sym_size", "sym_stride",
         "sym_numel", "sym_storage_offset", "stride", "size", "numel", "dim", "is_contiguous", "_local_scalar_dense", "set_", "item"}
agg = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        if name not in VIEWS:
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "glare_amd" in fr.filename:
                    where = "%s:%d %s" % (fr.filename.split("glare_amd/")[-1], fr.lineno, fr.name)
                    break
            shp = tuple(args[0].shape) if args and torch.is_tensor(args[0]) else ()
            agg[(name, where, str(shp)[:40])] += 1
        return func(*args, **(kwargs or {}))


with torch.no_grad():
    for _ in range(2):
        netG.reverse_flow_nhwc(net_vq, lr)
    torch.cuda.synchronize()
    with Spy():
        netG.reverse_flow_nhwc(net_vq, lr)
torch.cuda.synchronize()
print("%d stock aten ops in one inference step" % sum(agg.values()))
for (name, where, shp), n in agg.most_common(40):
    print("%4d  %-16s %-60s %s" % (n, name, where, shp))
