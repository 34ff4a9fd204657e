#!/usr/bin/env python
"""CPU study (test-side tool, imports oracle/): tools/winograd_study.py found that the fp16 rounding of the FILTERS, not of the
activations, is what stages D + E lose against the fp32 oracle (filters kept in fp32: |dPSNR vs GT| 0.0002-0.0018 dB instead of 0.015-0.018
on the third weight set): a filter's rounding error is the SAME perturbation at every pixel -- a coherent gain / offset error per output
channel -- where activation roundings average out over the image.  This tool prices roundings of the filter that cost nothing at run time:
  rne        : round-to-nearest-even, today's pack kernel;
  feedback   : error feedback along (cin, tap) per output channel -- w_i + carry rounded to fp16, carry = what the rounding dropped: each
               weight is still within one fp16 ulp of its value, and the SUM of the rounding errors of an output channel's filter is < 1/2 ulp
               (the response to the mean of the input is preserved);
    python tools/filter_rounding_study.py [h w] [seed ...]        (PARITY_WEIGHT_SEED as tools/parity_scenes.py; default 2)
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from glare_amd.synthetic import representative_init_, synthetic_pair  # noqa: E402
from oracle import torch_ref as O  # noqa: E402
from precision_study import psnr_delta  # noqa: E402
from winograd_study import Emulate, r16  # noqa: E402


def round_feedback(w):
    """fp32 [K, C, kh, kw] -> fp16-representable fp32 of the same shape: error feedback along the flattened (C, kh, kw) axis per output channel."""
    K = w.shape[0]
    flat = w.reshape(K, -1).double()
    out = torch.empty_like(flat)
    carry = torch.zeros(K, dtype=torch.float64)
    for i in range(flat.shape[1]):
        t = flat[:, i] + carry
        q = t.float().half().double()
        out[:, i] = q
        carry = t - q
    return out.float().reshape(w.shape)


_CACHE = {}


def conv_with(rounder):
    def fn(x, conv):
        key = (id(conv), rounder.__name__)
        if key not in _CACHE:
            _CACHE[key] = rounder(conv.weight.detach().float())
        return r16(F.conv2d(r16(x), _CACHE[key], conv.bias, 1, 1))
    return fn


def rne(w):
    return r16(w)


def main():
    args = [int(a) for a in sys.argv[1:]]
    h, w = (args[0], args[1]) if len(args) >= 2 else (400, 600)
    seeds = args[2:] or [11, 12, 13]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    wseed = int(os.environ.get("PARITY_WEIGHT_SEED", "2"))
    og, ov = representative_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), O.VQModel().eval(), wseed)
    print("== filter roundings priced on stages D + E of the fp32 oracle (fp16 activations, fp32 accumulate), %dx%d, weights seed %d" % (h, w, wseed))
    rows = []
    for s in seeds:
        lr = O.preprocess(synthetic_pair(1, h, w, seed=s)[0][0])
        t0 = time.time()
        with torch.no_grad():
            ref = og.stages(ov, lr)
            res = {}
            for tag, rd in (("rne", rne), ("feedback", round_feedback)):
                with Emulate([ov.decoder, og.deformable_decoder], conv_with(rd)):
                    _, _, code_feats = ov.decode(ref["latent"])
                    res[tag] = psnr_delta(og.deformable_decoder(ref["latent"], list(code_feats), ref["enc"]["mid_feat"]), ref["out"], h)
        rows.append(res)
        print("seed %3d (%.0f s): " % (s, time.time() - t0) + " | ".join("%s PSNR(out,oracle) %.2f dB, |dPSNR vs GT| %.4f dB" % (k, v[0], v[1]) for k, v in res.items()), flush=True)
    for k in rows[0]:
        print("-- %-9s over %d scenes: PSNR(out,oracle) min %.2f dB | |dPSNR vs GT| max %.4f mean %.4f dB"
              % (k, len(rows), min(r[k][0] for r in rows), max(r[k][1] for r in rows), sum(r[k][1] for r in rows) / len(rows)))


if __name__ == "__main__":
    main()
