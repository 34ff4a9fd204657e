#!/usr/bin/env python
"""CPU study (test-side tool, imports oracle/): what activation precision in stages A (conditional encoder) and B (flow) does
the codebook search need?  The fp32 oracle is run with the outputs of every conv / norm / attention product of A + B rounded
to a storage format (bf16, fp16, or split per section), in both weight regimes (synthetic.seeded_init_ = adversarial,
synthetic.representative_init_ = trained-like).  Prints latent error, index agreement and the PSNR figures of the decoded
result (stages D + E always fp32 here: the question is the TOKENS).

    python tools/precision_study.py [h w]
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from glare_amd.synthetic import representative_init_, seeded_init_, synthetic_lowlight, synthetic_pair  # noqa: E402
from oracle import torch_ref as O  # noqa: E402


def rounder(dtype):
    return (lambda t: t) if dtype is None else (lambda t: t.to(dtype).float())


class Rounding:
    """Emulate 16-bit storage inside `root`: round the weights of the selected convs, the outputs of the selected leaf modules
    (conv, norm), the softmax probabilities, and -- unless `fp32_stream` -- the block outputs (the residual stream)."""

    def __init__(self, root, dtype, select=lambda name: True, fp32_stream=False):
        stream_sel = fp32_stream if callable(fp32_stream) else (lambda n, v=bool(fp32_stream): v)
        self.h, self.saved = [], []
        r = rounder(dtype)
        self.r = r
        hook = lambda mod, i, o: r(o) if torch.is_tensor(o) else o
        for name, m in root.named_modules():
            if not select(name):
                continue
            if isinstance(m, nn.Conv2d):
                self.saved.append((m.weight, m.weight.data.clone()))
                m.weight.data.copy_(r(m.weight.data))
            stream_conv = isinstance(m, nn.Conv2d) and (name.endswith("conv2") or name.endswith("proj_out") or name.endswith("nin_shortcut")
                                                        or name.endswith("conv_in") or name.endswith("downsample.conv"))
            if isinstance(m, (nn.Conv2d, nn.GroupNorm)) and not (stream_sel(name) and stream_conv):
                self.h.append(m.register_forward_hook(hook))
            if isinstance(m, (O.ResnetBlock, O.AttnBlock)) and not stream_sel(name):
                self.h.append(m.register_forward_hook(hook))

    def __enter__(self):
        self.soft = O.F.softmax
        O.F.softmax = lambda *a, **k: self.r(self.soft(*a, **k))
        return self

    def __exit__(self, *a):
        O.F.softmax = self.soft
        for h in self.h:
            h.remove()
        for p, v in self.saved:
            p.data.copy_(v)


def psnr_delta(out, ref, h):
    a, b = O.postprocess(out, h), O.postprocess(ref, h)
    rng = np.random.default_rng(5)
    gt = np.clip(b + rng.normal(0, 10 ** (-27 / 20), b.shape), 0, 1)
    gt = np.round(gt * 255).astype(np.uint8)
    pa, pb = O.psnr(gt / 255, O.postprocess(out, h, gt)), O.psnr(gt / 255, O.postprocess(ref, h, gt))
    return float(O.psnr(a, b)), float(abs(pa - pb))


def run(og, ov, lr, h, tag):
    with torch.no_grad():
        ref = og.stages(ov, lr)
    lat = ref["latent"]
    cb = ov.quantize.embedding.weight.detach()
    tok = lat.permute(0, 2, 3, 1).reshape(-1, 3)
    d = torch.cdist(tok, cb)
    top2 = d.topk(2, dim=1, largest=False).values
    print("== %s: latent |mean| %.3f std %.3f, codebook std %.3f, nearest-code dist median %.4f, margin median %.4f, codes used %d"
          % (tag, lat.abs().mean(), lat.std(), cb.std(), top2[:, 0].median(), (top2[:, 1] - top2[:, 0]).median(),
             ref["indices"].unique().numel()))
    isA = lambda n: n.startswith("RRDB")
    isB = lambda n: n.startswith("flowUpsamplerNet")
    plans = [("bf16 A+B", torch.bfloat16, lambda n: True, False), ("fp16 A+B", torch.float16, lambda n: True, False),
             ("fp16 A+B, fp32 stream", torch.float16, lambda n: True, True),
             ("fp16, fp32 stream q-res", torch.float16, lambda n: True, lambda n: ".down.2." in n or ".mid." in n),
             ("fp16, fp32 stream lvl0-1", torch.float16, lambda n: True, lambda n: not (".down.2." in n or ".mid." in n)),
             ("fp16 A, fp32 flow", torch.float16, isA, False), ("fp32 A, fp16 flow", torch.float16, isB, False),
             ("fp32 A, bf16 flow", torch.bfloat16, isB, False)]
    for name, dt, sel, stream in plans:
        with torch.no_grad(), Rounding(og, dt, lambda n: (isA(n) or isB(n)) and sel(n), stream):
            enc = og.RRDB(lr, mid_feat=True)
            x, _ = og.flowUpsamplerNet.decode(enc["color_map"], enc["cond_feat"])
        with torch.no_grad():
            rec, _, feats = ov.decode(x)
            idx = ov.last_indices
            out = og.deformable_decoder(x, list(feats), ref["enc"]["mid_feat"])
        agree = float((idx == ref["indices"]).float().mean())
        p, dl = psnr_delta(out, ref["out"], h)
        print("  %-26s latent rel %.2e  idx agree %.5f  PSNR(ours,oracle) %.2f dB  dPSNR@27dB %.4f"
              % (name, float((x - lat).norm() / lat.norm()), agree, p, dl), flush=True)


def main():
    h = int(sys.argv[1]) if len(sys.argv) > 2 else 100
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 156
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    lr_noise = O.preprocess(synthetic_lowlight(1, h, w, seed=21)[0])
    lr_pair = O.preprocess(synthetic_pair(1, h, w, seed=123)[0][0])
    og = seeded_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), 0)
    ov = seeded_init_(O.VQModel().eval(), 1)
    run(og, ov, lr_noise, h, "adversarial regime (seeded_init_), noise image")
    og, ov = representative_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), O.VQModel().eval(), 0)
    run(og, ov, lr_pair, h, "representative regime, scene image")
    run(og, ov, lr_noise, h, "representative regime, noise image")


if __name__ == "__main__":
    main()
