"""sha256 of every stage tensor of one seeded batch through the inference path (fp16 default): two BUILDS of the library that print the
same lines compute bit-identical results (A/B of a kernel change that must not move a bit).   python tools/path_hash.py [batch] [h w]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glare_amd import modules as M, ops  # noqa: E402
from glare_amd.synthetic import seeded_init_, synthetic_lowlight  # noqa: E402
from glare_amd.harness import preprocess_device  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (400, 600)
    dev = torch.device("cuda:0")
    netG = seeded_init_(M.VQLLFLOWDeformable().eval(), 0).to(dev)
    vq = seeded_init_(M.VQModel().eval(), 1).to(dev)
    lr = preprocess_device(torch.from_numpy(synthetic_lowlight(B, h, w)).to(dev))
    with torch.no_grad(), ops.use_precision(ops.inference_precision()):
        got = netG.reverse_flow_nhwc(vq, lr)
    torch.cuda.synchronize()

    def walk(prefix, v):
        if torch.is_tensor(v):
            t = v.detach().contiguous().cpu()
            raw = t.view(torch.uint8).numpy().tobytes() if t.dtype != torch.bool else t.numpy().tobytes()
            print("%-28s %-14s %s" % (prefix, str(tuple(t.shape)), hashlib.sha256(raw).hexdigest()[:16]))
        elif isinstance(v, dict):
            for k in sorted(v):
                walk(prefix + "." + str(k), v[k])
        elif isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                walk("%s[%d]" % (prefix, i), x)

    walk("out", got)


if __name__ == "__main__":
    main()
