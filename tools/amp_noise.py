"""CPU: how far the REFERENCE's own mixed-precision arithmetic moves the AFT decoder's parameter gradients (row a13).

The oracle (bit-identical restatement of the reference's modules, tests/test_oracle_vs_reference.py) is run on the stage-3 inputs of
tests/test_gpu_train.py::test_aft_decoder_backward_on_the_pipelines_own_inputs once in fp32 and once under torch.autocast -- the form
the reference trains in (LLFlow_model.py:236-241) -- and every MultiScaleDecoder2 parameter gradient is compared per tensor
(relative L2).  That is the noise floor of 16-bit training on this graph; the product's gradients are measured against the same
fp32 gradients by the GPU test.

Measured (256x256 crop, trained-like weights, 156 tensors; ~15 min for fp16 on 32 CPU threads):
    reference fp16 autocast : forward 9.5e-4 of max|out|, gradient median 3.3 %   (max: inf with a loss scale of 4096 -- CPU
                              autocast overflows in five tensors -- and meaningless without one: fp16 underflow)
    reference bf16 autocast : forward 6.6e-3,            gradient median 11.2 %
    product fp16 (MI355X)   : forward 4.8e-4,            gradient median 2.05 %, max 5.2 %
    product bf16 (MI355X)   : forward 3.9e-3,            gradient median 8.2 %,  max 32 %
i.e. the product's training gradients sit inside the reference's own mixed-precision noise in both formats.

usage: python tools/amp_noise.py [fp16|bf16] [loss_scale]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from glare_amd.synthetic import representative_init_, synthetic_pair  # noqa: E402
from oracle import torch_ref as O  # noqa: E402


def main():
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    og, ov = representative_init_(O.VQLLFLOWDeformable(per_sample_mean=False).eval(), O.VQModel().eval(), 0)
    lr = O.preprocess(synthetic_pair(1, 236, 236, seed=41)[0][0])
    with torch.no_grad():
        st = og.stages(ov, lr)
    z, code, enc = st["latent"], list(st["code_feats"]), list(st["enc"]["mid_feat"])
    ref = og.deformable_decoder.train()
    wgt = torch.randn(1, 3, z.shape[2] * 4, z.shape[3] * 4, generator=torch.Generator().manual_seed(12))

    def grads(dt):
        for p_ in ref.parameters():
            p_.grad = None
        t0 = time.time()
        if dt is None:
            out = ref(z, code, enc)
        else:
            with torch.autocast("cpu", dtype=dt):
                out = ref(z, code, enc)
        ((out.float() * wgt).sum() * scale).backward()
        print(dt, "%.0f s" % (time.time() - t0), flush=True)
        return out.detach().float(), {k: p_.grad.clone() / scale for k, p_ in ref.named_parameters() if p_.grad is not None}

    o32, g32 = grads(None)
    o, g = grads(dtype)
    errs = sorted((float((g[k] - g32[k]).norm() / (g32[k].norm() + 1e-30)), k) for k in g32)
    print("forward %.3g of max|out|; gradient relative L2 per tensor: median %.4f, max %.4g (%s), %d tensors"
          % (float((o - o32).abs().max() / o32.abs().max()), errs[len(errs) // 2][0], errs[-1][0], errs[-1][1], len(errs)))


if __name__ == "__main__":
    main()
