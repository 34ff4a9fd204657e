#!/usr/bin/env python
"""CPU study (test-side tool, imports oracle/): the fp32 oracle with the PRODUCT's rounding sites of stages A + B emulated one by
one -- every MFMA conv's activation operand, its filter, its stored output -- and each site either at 16 bits (11-bit fp16
mantissa: one MFMA pass) or at 22 bits (a hi / lo pair: one more MFMA pass per pair, or for a stored output just a second 16-bit
tensor).  Answers, before any kernel is written: which convs need the second / third MFMA pass for the codebook search to agree
with the fp32 oracle?

    python tools/precision_sites.py [h w] [seed ...]
"""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from glare_amd.synthetic import representative_init_, synthetic_pair  # noqa: E402
from oracle import torch_ref as O  # noqa: E402
from precision_study import psnr_delta  # noqa: E402


def r16(t):
    return t.half().float()


def r22(t):
    hi = t.half().float()
    return hi + (t - hi).half().float()


FEEDBACK16 = None      # set by main() under SITES_FEEDBACK=1: 16-bit filters rounded with error feedback (tools/filter_rounding_study.py) instead of RNE


DIRECT = ("RRDB.encoder.conv_in", "RRDB.color_conv", "RRDB.cond_conv.0")   # fp32 direct convs (conv_small.hip)
F32_OUT = ("RRDB.encoder.conv_out", "RRDB.color_conv")


def group_of(name):
    p = name.split(".")
    if name.startswith("flowUpsamplerNet"):
        return "flow"
    if "attn" in p or "attn_1" in p:
        return "attn"
    if "down" in p:
        return "down%s" % p[p.index("down") + 1]
    if "mid" in p:
        return "mid"
    return "out"


class Sites:
    """alo / wlo / olo: predicates on a conv's module name -> its activation operand / filter / stored output keeps 22 bits."""

    def __init__(self, root, alo=lambda n: False, wlo=lambda n: False, olo=lambda n: False, cond22=False, soft16=True):
        self.h, self.saved, self.soft16 = [], [], soft16
        for name, m in root.named_modules():
            if not (name.startswith("RRDB") or name.startswith("flowUpsamplerNet")):
                continue
            if isinstance(m, O.AttnBlock):
                # keys / values / the query conv's operand are the stream's hi half (the block's norm is folded into the filters)
                self.h.append(m.norm.register_forward_pre_hook(lambda mod, a, f=(r22 if alo(name + ".q") else r16): (f(a[0]),)))
            if not isinstance(m, nn.Conv2d):
                continue
            if name == "RRDB.cond_conv.0":
                self.h.append(m.register_forward_hook(lambda mod, i, o: o))   # its sigmoid output is rounded below (Sequential)
                continue
            if name in DIRECT:
                continue
            stream = name.endswith(("conv2", "proj_out", "nin_shortcut", "downsample.conv"))
            is_flow = name.startswith("flowUpsamplerNet")
            wf = r22 if wlo(name) else (FEEDBACK16 or r16)
            self.saved.append((m.weight, m.weight.data.clone()))
            w = m.weight.data
            if is_flow and name.endswith("fAffine.0"):
                w[:, 1:] = wf(w[:, 1:])                  # channel 0 = z1: fp32 direct in flow_h1_kernel
            else:
                w.copy_(wf(w))
            in_attn = ".attn" in name and name.endswith((".q", ".k", ".v"))
            if not in_attn and not (is_flow and name.endswith(".0")):
                af = r22 if alo(name) else r16
                self.h.append(m.register_forward_pre_hook(lambda mod, a, f=af: (f(a[0]),)))
            if name in F32_OUT:
                continue
            of = r22 if (stream or olo(name)) else r16
            self.h.append(m.register_forward_hook(lambda mod, i, o, f=of: f(o)))
        # cond_feat (sigmoid output, 16-bit NHWC tensor read by every coupling net)
        self.h.append(root.RRDB.cond_conv.register_forward_hook(lambda mod, i, o, f=(r22 if cond22 else r16): f(o)))

    def __enter__(self):
        self.soft = O.F.softmax
        if self.soft16:
            O.F.softmax = lambda *a, **k: r16(self.soft(*a, **k))
        return self

    def __exit__(self, *a):
        O.F.softmax = self.soft
        for h in self.h:
            h.remove()
        for p, v in self.saved:
            p.data.copy_(v)


def main():
    args = [int(a) for a in sys.argv[1:]]
    h, w = (args[0], args[1]) if len(args) >= 2 else (100, 156)
    seeds = args[2:] or [123, 124, 125]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    og, ov = representative_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), O.VQModel().eval(), 0)
    lrs = [O.preprocess(synthetic_pair(1, h, w, seed=s)[0][0]) for s in seeds]
    with torch.no_grad():
        refs = [og.stages(ov, lr) for lr in lrs]

    def measure(tag, full=False, **kw):
        errs, agr, dl = [], [], []
        for lr, ref in zip(lrs, refs):
            with torch.no_grad(), Sites(og, **kw):
                enc = og.RRDB(lr, mid_feat=True)
                x, _ = og.flowUpsamplerNet.decode(enc["color_map"], enc["cond_feat"])
            with torch.no_grad():
                rec, _, feats = ov.decode(x)
                idx = ov.last_indices
                if full:
                    out = og.deformable_decoder(x, list(feats), ref["enc"]["mid_feat"])
                    dl.append(psnr_delta(out, ref["out"], h))
            lat = ref["latent"]
            errs.append(float((x - lat).norm() / lat.norm()))
            agr.append(float((idx == ref["indices"]).float().mean()))
        extra = ""
        if full:
            extra = "  PSNR(ours,oracle) %s  dPSNR@27dB %s" % (" ".join("%.2f" % p for p, _ in dl), " ".join("%.4f" % d for _, d in dl))
        print("  %-58s latent rel %.3e  idx agree %.4f (min %.4f)%s" % (tag, sum(errs) / len(errs), sum(agr) / len(agr), min(agr), extra),
              flush=True)

    T, Fa = (lambda n: True), (lambda n: False)
    G = lambda *gs: (lambda n: group_of(n) in gs)
    enc = lambda n: n.startswith("RRDB")
    full = os.environ.get("SITES_FULL", "0") == "1"
    print("== %dx%d, seeds %s" % (h, w, seeds))
    measure("product today (16-bit operands, hi/lo stream)", full)
    measure("+ conv1 outputs hi/lo (no MFMA)", full, olo=enc)
    measure("+ filters hi/lo everywhere", full, wlo=T)
    measure("+ activations hi/lo everywhere", full, alo=T)
    measure("+ filters + activations (3 passes), conv1 out 16", full, alo=T, wlo=T)
    measure("+ filters + activations + conv1 out (fp32-class A+B)", full, alo=T, wlo=T, olo=T)
    measure("fp32-class A+B + cond_feat hi/lo", full, alo=T, wlo=T, olo=T, cond22=True)
    measure("fp32-class A+B + cond_feat hi/lo + fp32 softmax P", full, alo=T, wlo=T, olo=T, cond22=True, soft16=False)
    if os.environ.get("SITES_FEEDBACK", "0") == "1":
        # round 6: can the THIRD pass of the fp32-class convs (x_hi . w_lo, the filter's lo half) go if the 16-bit filter is rounded with error
        # feedback per output channel (ops.filter_feedback_round)?  Stages D / E gained 3 dB from it; here the question is the LATENT
        global FEEDBACK16
        from filter_rounding_study import round_feedback
        measure("reference point: fp32-class A+B + cond_feat hi/lo", full, alo=T, wlo=T, olo=T, cond22=True)
        measure("16-bit RNE filters everywhere (2 passes), acts / outputs hi/lo", full, alo=T, wlo=Fa, olo=T, cond22=True)
        FEEDBACK16 = lambda w: round_feedback(w) if w.dim() == 4 else r16(w)
        measure("16-bit FEEDBACK filters everywhere (2 passes), acts / outputs hi/lo", full, alo=T, wlo=Fa, olo=T, cond22=True)
        for g in ("down0", "down1", "down2", "mid", "attn", "out", "flow"):
            measure("16-bit FEEDBACK filters in %s only" % g, full, alo=T, wlo=(lambda n, g=g: group_of(n) != g), olo=T, cond22=True)
        FEEDBACK16 = None
        return
    if os.environ.get("SITES_PER_CONV", "0") == "1":
        # is there ANY conv of the conditional encoder whose third MFMA pass (x_hi . w_lo: the filter's lo half) or
        # second pass (x_lo . w_hi: the activation's lo half) can be dropped without moving the table?  One site at a time, everything
        # else at the full fp32-class scheme.
        names = [n for n, m in og.named_modules() if isinstance(m, nn.Conv2d) and n.startswith("RRDB") and n not in DIRECT]
        measure("reference point: fp32-class A+B + cond_feat hi/lo", full, alo=T, wlo=T, olo=T, cond22=True)
        for n in names:
            measure("16-bit FILTER at %s only (2 passes there)" % n[5:], full, alo=T, wlo=(lambda x, n=n: x != n), olo=T, cond22=True)
        for n in names:
            measure("16-bit ACTIVATION OPERAND at %s only" % n[5:], full, alo=(lambda x, n=n: x != n), wlo=T, olo=T, cond22=True)
        measure("16-bit filters in the whole flow only", full, alo=T, wlo=(lambda x: not x.startswith("flowUpsamplerNet")), olo=T, cond22=True)
        return
    if os.environ.get("SITES_SHORT", "0") == "1":
        return
    measure("fp32-class encoder, flow as today", full, alo=enc, wlo=enc, olo=enc)
    measure("fp32-class flow, encoder as today", full, alo=G("flow"), wlo=G("flow"), olo=G("flow"))
    for g in ("down0", "down1", "down2", "mid", "attn", "out", "flow"):
        others = [x for x in ("down0", "down1", "down2", "mid", "attn", "out", "flow") if x != g]
        measure("fp32-class everywhere except %s" % g, full, alo=G(*others), wlo=G(*others), olo=G(*others))
    measure("fp32-class except down0, down1", full, alo=G("down2", "mid", "attn", "out", "flow"), wlo=G("down2", "mid", "attn", "out", "flow"),
            olo=T)
    measure("filters everywhere + acts except down0/down1 + olo", full, alo=G("down2", "mid", "attn", "out", "flow"), wlo=T, olo=T)
    measure("acts everywhere + filters except down0/down1 + olo", full, wlo=G("down2", "mid", "attn", "out", "flow"), alo=T, olo=T)


if __name__ == "__main__":
    main()
