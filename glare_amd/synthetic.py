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


def reset_actnorms_(module):
    """Put every ActNorm of `module` into the state of a freshly constructed flow (FlowActNorms.py:19-23: zero bias / logs,
    `inited = False`), so that the next TRAINING forward runs the data-dependent initialisation (:32-46)."""
    with torch.no_grad():
        for m in module.modules():
            if hasattr(m, "inited") and hasattr(m, "logs") and hasattr(m, "bias"):
                m.bias.zero_()
                m.logs.zero_()
                m.inited = False
    return module


def synthetic_lowlight(batch, h=400, w=600, seed=1234):
    """Dark LOL-like uint8 RGB images [B,h,w,3]: floor(255 * u^2.2 * 0.25), u ~ U(0,1)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand((batch, h, w, 3), generator=g)
    return torch.floor(255 * u.pow(2.2) * 0.25).to(torch.uint8).numpy()


def synthetic_gt(batch, h=400, w=600, seed=4321):
    g = torch.Generator().manual_seed(seed)
    return torch.floor(torch.rand((batch, h, w, 3), generator=g) * 256).clamp(0, 255).to(torch.uint8).numpy()


def synthetic_pair(batch, h=400, w=600, seed=77):
    """A LOL-like (low, normal) uint8 pair [B,h,w,3] of the SAME scene: the normal-light image is a smooth random field
    (1/8-resolution noise, bilinearly upsampled) plus fine texture; the low-light one is its gamma-darkened, noisy copy."""
    g = torch.Generator().manual_seed(seed)
    coarse = torch.rand((batch, 3, (h + 7) // 8 + 1, (w + 7) // 8 + 1), generator=g)
    img = torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=True)
    img = (0.8 * img + 0.2 * torch.rand((batch, 3, h, w), generator=g)).clamp(0, 1)
    low = (img.pow(2.2) * 0.25 + 0.01 * torch.randn((batch, 3, h, w), generator=g)).clamp(0, 1)
    to_u8 = lambda t: torch.floor(t * 255).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()
    return to_u8(low), to_u8(img)


def representative_init_(netG, net_vq, seed=0, batch=4, size=192, device=None, coupling_gain=0.25):
    """A weight regime shaped like a TRAINED GLARE out of seeded random weights (no checkpoints offline), for the end-to-end
    parity figures.  `seeded_init_` alone leaves the flow's reverse pass ~90 codebook radii away from every code (48 divisions
    by sigmoid(.)+1e-4 with unnormalised ActNorms), where a token's nearest code is decided by 1e-4 relative margins -- the
    adversarial regime.  Here, on a synthetic (low, normal) batch and with nothing but the modules' own forward passes:

      1. codebook <- 8192 sampled latents of `net_vq.encode(normal)`          (what VQGAN training converges to: codes ON the
                                                                               encoder's latent distribution, quantize.py:271-312)
      2. the flow's 124 ActNorms <- their data-dependent initialisation       (FlowActNorms.py:32-46: one training forward of
         on (latents, cond_feat), Conv2dZeros kept at their seeded values      `flowUpsamplerNet`, exactly what train_stage2.py's
                                                                               first step does)
      3. `RRDB.color_conv` rescaled so that color_map has zero mean / unit    (stage 2 trains z ~ N(color_map, 1): the mean of the
         variance per channel on the batch                                     Gaussian lives where the encoded latents live)

    so that `reverse_flow` lands INSIDE the codebook cloud.  Works on any module set with the reference's interface (the imported
    reference, the CPU oracle, the HIP product modules); the result is deterministic for a given module set, and tests copy
    the state dict across sets."""
    seeded_init_(netG, seed)
    seeded_init_(net_vq, seed + 1)
    flow = netG.flowUpsamplerNet
    reset_actnorms_(flow)
    with torch.no_grad():   # see the docstring: an untrained coupling at seeded_init_'s strength is unstable in reverse
        for name, p in flow.named_parameters():
            if (".fAffine.4." in name or ".fFeatures.4." in name) and name.endswith("weight"):
                p.mul_(coupling_gain)
    low, normal = synthetic_pair(batch, size, size, seed=seed * 31 + 7)
    gt = torch.from_numpy(normal).permute(0, 3, 1, 2).float() / 255
    lr = torch.from_numpy(low).permute(0, 3, 1, 2).float() / 255
    lr = torch.log(torch.clamp(lr + 1e-3, min=1e-3))
    if device is not None:
        gt, lr = gt.to(device), lr.to(device)
    was = (netG.training, net_vq.training)
    netG.eval()
    net_vq.eval()
    with torch.no_grad():
        lat = net_vq.encode(gt)
        lat = (lat[0] if isinstance(lat, (tuple, list)) else lat).float()
        tokens = lat.permute(0, 2, 3, 1).reshape(-1, lat.shape[1]).cpu()
        n_e = net_vq.quantize.embedding.weight.shape[0]
        assert tokens.shape[0] >= n_e, "batch too small for the codebook"
        pick = torch.randperm(tokens.shape[0], generator=torch.Generator().manual_seed(seed + 5))[:n_e]
        net_vq.quantize.embedding.weight.copy_(tokens[pick].to(net_vq.quantize.embedding.weight.device))
        enc = netG.RRDB(lr)
        cm = enc["color_map"].float()
        mu = cm.mean(dim=(0, 2, 3)).cpu()
        sd = cm.std(dim=(0, 2, 3)).cpu()
        cc = netG.RRDB.color_conv
        cc.bias.copy_(((cc.bias.detach().cpu() - mu) / sd).to(cc.bias.device))
        cc.weight.copy_((cc.weight.detach().cpu() / sd.view(-1, 1, 1, 1)).to(cc.weight.device))
        flow.train()
        flow(gt=lat, rrdbResults=enc, logdet=torch.zeros_like(lat[:, 0, 0, 0]), reverse=False)
    netG.train(was[0])
    net_vq.train(was[1])
    for m in (netG, net_vq):
        if hasattr(m, "invalidate"):   # HIP modules cache packed weights
            m.invalidate()
    return netG, net_vq
