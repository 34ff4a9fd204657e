"""Synthetic weights and inputs (there is no network for checkpoints or datasets).

`seeded_init_` fills a module's state from its parameter NAMES, so any two modules with the same
state-dict keys (the reference import, the CPU oracle, the HIP-backed product modules) receive
bit-identical weights regardless of construction order.  Distributions follow the torch default
inits in scale, except where the reference's default is degenerate and would make a parity test
vacuous: the codebook (U(+-1/8192), quantize.py:230), Conv2dZeros (zeros,
flow.py:65-66), ActNorm (zeros, FlowActNorms.py:19-20), conv_offset (zeros, deform_conv.py:367-372).
"""
import math
import zlib

import numpy as np
import torch


def _gen(seed, name):
    return torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))


def seeded_init_(module, seed=0):
    sd = module.state_dict()
    new = {}
    for name in sorted(sd.keys()):
        t = sd[name]
        if not torch.is_floating_point(t):
            continue
        g = _gen(seed, name)
        shape = tuple(t.shape)
        leaf = name.rsplit(".", 1)[-1]

        def randn(std, mean=0.0):
            return torch.randn(shape, generator=g) * std + mean

        def uniform(a):
            return (torch.rand(shape, generator=g) * 2 - 1) * a

        if "mix." in name and leaf == "w":
            continue  # Mix keeps its constructor value (-1.0 / -0.6, deformableDecoder_arch.py:522-523)
        if name.endswith("embedding.weight"):
            v = randn(0.7)
        elif "invconv.weight" in name:
            q, _ = torch.linalg.qr(torch.randn(shape, generator=g, dtype=torch.float64))
            v = q.float()
        elif "actnorm." in name:
            v = randn(0.1)
        elif "conv_offset.weight" in name:
            v = randn(0.01)
        elif "conv_offset.bias" in name:
            v = uniform(2.0)
        elif leaf == "logs":  # Conv2dZeros.logs
            v = torch.zeros(shape)
        elif (".fAffine.4." in name or ".fFeatures.4." in name) and leaf == "weight":
            v = randn(0.02)
        elif (".fAffine." in name or ".fFeatures." in name) and leaf == "weight":
            v = randn(0.05)
        elif "norm" in name and leaf == "weight":
            v = randn(0.1, 1.0)
        elif "norm" in name and leaf == "bias":
            v = randn(0.1)
        elif leaf == "weight" and t.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = uniform(1.0 / math.sqrt(fan_in))
        elif leaf == "bias":
            v = uniform(0.05)
        else:
            v = randn(0.1)
        new[name] = v.to(t.dtype)
    sd.update(new)
    module.load_state_dict(sd)
    return module


def synthetic_lowlight(batch, h=400, w=600, seed=1234):
    """Dark LOL-like uint8 RGB images [B,h,w,3]: floor(255 * u^2.2 * 0.25), u ~ U(0,1)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand((batch, h, w, 3), generator=g)
    return torch.floor(255 * u.pow(2.2) * 0.25).to(torch.uint8).numpy()


def synthetic_gt(batch, h=400, w=600, seed=4321):
    g = torch.Generator().manual_seed(seed)
    return torch.floor(torch.rand((batch, h, w, 3), generator=g) * 256).clamp(0, 255).to(torch.uint8).numpy()
