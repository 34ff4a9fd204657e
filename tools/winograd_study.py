#!/usr/bin/env python
"""CPU study (test-side tool, imports oracle/): what would Winograd F(2x2, 3x3) on the two decoders' single-pass 3x3 convs cost in dB?
("price it in dB first".)

Stages D (VQGAN decoder) and E (AFT decoder) of the fp32 oracle are re-run from the oracle's own latent / indices / encoder features with
the PRODUCT's rounding sites of a stride-1 3x3 MFMA conv emulated two ways:
  direct16 : activation operand and filter each rounded once to fp16, fp32 accumulation, output stored as fp16 -- today's single-pass kernel;
  wino16   : the same conv as F(2x2, 3x3): U = G g G^T (computed in fp32 from the fp32 filter, rounded to fp16: the 16 filter planes the
             MFMA reads), V = B^T d B of the fp16 activation tile (sums of four fp16 values, rounded to fp16: the MFMA's other operand),
             M = sum_c U . V in fp32, Y = A^T M A in fp32, output stored as fp16.
Everything else in D / E stays fp32 in both (GroupNorm, attention, DCN, 1x1 convs): the table isolates what the transform's two extra
roundings add.  Reported per scene against the fp32 oracle: PSNR(out, oracle) and |dPSNR vs GT| (the BASELINE tolerance: 0.05 dB), on the
weight set the round-5 verdict names (PARITY_WEIGHT_SEED=2: the one whose margin is 2.5x, not 10x).

    PARITY_WEIGHT_SEED=2 python tools/winograd_study.py [h w] [seed ...]
"""
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from glare_amd.synthetic import representative_init_, synthetic_pair  # noqa: E402
from oracle import torch_ref as O  # noqa: E402
from precision_study import psnr_delta  # noqa: E402

BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])


def r16(t):
    return t.half().float()


def conv_direct16(x, conv):
    return r16(F.conv2d(r16(x), r16(conv.weight), conv.bias, 1, 1))


def conv_wino16(x, conv, round_v=True, round_u=True):
    """F(2x2, 3x3), pad 1, stride 1: tiles of 4x4 input -> 2x2 output."""
    B, C, H, W = x.shape
    K = conv.weight.shape[0]
    He, We = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = F.pad(r16(x), (1, 1 + We - W, 1, 1 + He - H))                       # fp16 activation, zero padding
    tiles = F.unfold(xp, kernel_size=4, stride=2)                              # [B, C*16, T]
    T = tiles.shape[-1]
    d = tiles.view(B, C, 4, 4, T)
    V = torch.einsum("ij,bcjkt,lk->bcilt", BT, d, BT)                          # B^T d B (fp32 sums of fp16 values)
    U = torch.einsum("ij,kcjl,ml->kcim", G, conv.weight.detach().float(), G)   # G g G^T, fp32 filter
    if round_v:
        V = r16(V)
    if round_u:
        U = r16(U)
    M = torch.einsum("kcim,bcimt->bkimt", U, V)                                # 16 independent [K x C] . [C x T] products, fp32 accumulate
    Y = torch.einsum("pi,bkimt,qm->bkpqt", AT, M, AT)                          # A^T M A -> [B, K, 2, 2, T]
    out = F.fold(Y.reshape(B, K * 4, T), output_size=(He, We), kernel_size=2, stride=2)[:, :, :H, :W]
    if conv.bias is not None:
        out = out + conv.bias.view(1, -1, 1, 1)
    return r16(out)


class Emulate:
    """Patches every stride-1 3x3 conv with >= 64 input and output channels under the given roots."""

    def __init__(self, roots, fn):
        self.saved = []
        for root in roots:
            for name, m in root.named_modules():
                if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.in_channels >= 64 and m.out_channels >= 64 \
                        and m.out_channels != 108:                                   # (conv_offset C -> 108 writes fp32 planes for the DCN)
                    self.saved.append((m, m.forward))
                    m.forward = (lambda x, mod=m: fn(x, mod))
        self.n = len(self.saved)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for m, f in self.saved:
            m.forward = f


def main():
    args = [int(a) for a in sys.argv[1:]]
    h, w = (args[0], args[1]) if len(args) >= 2 else (400, 600)
    seeds = args[2:] or [11, 12, 13, 101, 102, 103]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    wseed = int(os.environ.get("PARITY_WEIGHT_SEED", "2"))
    og, ov = representative_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), O.VQModel().eval(), wseed)
    print("== Winograd F(2x2,3x3) priced on stages D + E of the fp32 oracle, %dx%d, representative weights (seed %d)" % (h, w, wseed))
    # self-check of the emulation: without the two roundings it IS the direct conv
    c = nn.Conv2d(64, 64, 3, 1, 1)
    x = torch.randn(1, 64, 11, 15)
    with torch.no_grad():
        e = float((conv_wino16(x, c, False, False) - conv_direct16(x, c)).abs().max() / conv_direct16(x, c).abs().max())
    print("emulation self-check (transform without its roundings vs direct, both with fp16 in / out): %.1e" % e)
    rows = []
    for s in seeds:
        lr = O.preprocess(synthetic_pair(1, h, w, seed=s)[0][0])
        t0 = time.time()
        with torch.no_grad():
            ref = og.stages(ov, lr)

            def de():
                _, _, code_feats = ov.decode(ref["latent"])
                return og.deformable_decoder(ref["latent"], list(code_feats), ref["enc"]["mid_feat"])

            res = {}
            for tag, fn in (("direct16", conv_direct16), ("wino16", conv_wino16),
                            ("wino16, fp32 U", lambda x, m: conv_wino16(x, m, True, False)),
                            ("wino16, fp32 V", lambda x, m: conv_wino16(x, m, False, True))):
                with Emulate([ov.decoder, og.deformable_decoder], fn) as em:
                    res[tag] = psnr_delta(de(), ref["out"], h)
                    nconv = em.n
        rows.append((s, res))
        print("seed %3d (%d convs, %.0f s): " % (s, nconv, time.time() - t0) +
              " | ".join("%s PSNR(out,oracle) %.2f dB, |dPSNR vs GT| %.4f dB" % (k, v[0], v[1]) for k, v in res.items()), flush=True)
    for k in rows[0][1]:
        print("-- %-15s over %d scenes: PSNR(out,oracle) min %.2f dB | |dPSNR vs GT| max %.4f mean %.4f dB"
              % (k, len(rows), min(r[1][k][0] for r in rows), max(r[1][k][1] for r in rows), sum(r[1][k][1] for r in rows) / len(rows)))


if __name__ == "__main__":
    main()
