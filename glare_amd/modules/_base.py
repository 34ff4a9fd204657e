"""Shared plumbing of the HIP-backed modules: lazy weight packing and NCHW <-> NHWC entry/exit."""
import torch
import torch.nn as nn

from .. import ops


class HipModule(nn.Module):
    """Caches kernel-ready (packed) weights; `invalidate()` after changing parameters."""

    def _packed(self, key, builder):
        key = (key, ops.act_dtype())          # packed filters are in the 16-bit format of the precision in use
        cache = self.__dict__.setdefault("_hip_cache", {})
        if key not in cache:
            with torch.no_grad():
                cache[key] = builder()
        return cache[key]

    def invalidate(self):
        for m in self.modules():
            m.__dict__.pop("_hip_cache", None)

    def _apply(self, fn, *a, **k):  # .to()/.cuda()/.float() move parameters: packed images are stale
        self.__dict__.pop("_hip_cache", None)
        return super()._apply(fn, *a, **k)


def packed_conv(mod, conv, key=None, scale=None, split=0, feedback=True):
    """PackedConv of an nn.Conv2d owned by `mod` (optionally scaling weight and bias first; split: the fp32-class form of
    ops.PackedConv -- K segments [w_hi | w_hi | w_lo] for an activation hi / lo pair).
    feedback (default): a single-pass 16-bit filter is rounded with error feedback per output channel (ops.filter_feedback_round) instead
    of round-to-nearest per weight: a filter's rounding error is the same perturbation at every pixel, and where the input's channels share
    a mean (everything behind GroupNorm + swish) the errors of an output channel add up coherently over (cin, tap); a rounding whose
    errors sum to zero per output channel removes that part.  Measured on the full path (profiles/r06_filter_feedback.txt), third
    weight set: |dPSNR vs GT| 0.019 -> 0.006 dB, PSNR(ours, oracle) 62.8 -> 65.2 dB; first set unchanged.  Variants measured and not
    taken: feedback ALSO on the sub-pixel upsample phases (0.0085 dB) or on the per-image attention folds (0.0121 dB), and feedback ONLY on
    the convs that directly follow a GroupNorm + swish (0.0080 dB) -- all within the tolerance, none better than this default."""
    key = key or ("conv", id(conv), split, bool(feedback))

    def build():
        w, b = conv.weight, conv.bias
        if scale is not None:
            w = w * scale
            b = None if b is None else b * scale
        if feedback and not split and ops.FILTER_FEEDBACK and w.dim() == 4 and w.shape[1] >= 8:
            w = ops.filter_feedback_round(w)      # (the fp32-class filters -- split -- are hi / lo pairs and keep 22 bits anyway)
        return ops.PackedConv(w, b, split=split)

    return mod._packed(key, build)


def to_nhwc(x, bf16=True):
    ops.require_cuda(x)
    return ops.nchw_to_nhwc(x, bf16=bf16)


def to_nchw(x):
    return ops.nhwc_to_nchw(x)
