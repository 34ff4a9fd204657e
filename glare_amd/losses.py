"""Stage-3 loss terms on the HIP path: MS-SSIM and the VGG16-feature perceptual loss that
VQLLFLOWDModel.optimize_parameters adds to the L1 term (code/models/VQLLFLOWD_model.py:217-223).

Mirrors code/models/modules/pytorch_msssim/__init__.py:8-98 (`gaussian`, `create_window`, `ssim`, `msssim`) and
code/models/modules/losses.py:12-40 (`PerceptualNetwork`).  Images are NHWC fp32 on the device; the ten SSIM level scalars
and the VGG features come from kernels (csrc/loss_ops.hip, csrc/conv_igemm.hip); the few scalar combinations below are
torch ops on 5-element tensors.
"""
from math import exp

import functools

import torch
import torch.nn as nn

from . import autograd as A
from .modules._base import HipModule

MSSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)   # pytorch_msssim/__init__.py:73
_WEIGHTS = {}


@functools.lru_cache(maxsize=None)      # five windows per MS-SSIM call, the same every step: built once (was ~15 small CPU tensor ops per step)
def gaussian(window_size, sigma=1.5):
    """The 1-D window of create_window() (:8-17) with the same float32 rounding: its outer product is the 2-D window."""
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return tuple((g / g.sum()).tolist())     # immutable: the cached object is shared by every caller


def msssim(sr, gt, window_size=11, normalize=False):
    """sr, gt: NHWC fp32 device tensors in [0, 1] (value range 1).  Returns the scalar of msssim() (:71-98), including its
    `prod(pow1[:-1] * pow2[-1])` combination."""
    B, H, W, C = sr.shape
    windows, h, w = [], H, W
    for _ in range(5):
        windows.append(gaussian(min(window_size, h, w)))     # real_size = min(window_size, height, width), :40-42
        h, w = h // 2, w // 2
    sims, css = A.msssim_terms(sr, gt, windows)
    weights = _WEIGHTS.get(sr.device)
    if weights is None:   # uploaded once per device (a pageable H2D copy is not allowed inside a captured step)
        weights = _WEIGHTS[sr.device] = torch.tensor(MSSSIM_WEIGHTS, dtype=torch.float32, device=sr.device)
    if normalize:                                             # :86-88
        sims, css = (sims + 1) / 2, (css + 1) / 2
    pow1, pow2 = css ** weights, sims ** weights
    x = pow1[:-1] * pow2[-1]                                  # :92 torch.prod(...), written out: prod's backward inspects
    return x[0] * x[1] * x[2] * x[3]                          # its input for zeros on the host (a sync; not graph-capturable)


class PerceptualNetwork(HipModule):
    """vgg16.features[:16] with taps after relu1_2 / relu2_2 / relu3_3 and the mean of the three feature MSEs
    (losses.py:12-40).  Parameter names are torchvision's (`vgg_model.<index>.weight`), so the pretrained file the reference
    downloads (`vgg16(pretrained=True)`) loads unchanged; offline the weights are whatever the caller initialises."""

    TAPS = (3, 8, 15)

    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in (64, 64, "M", 128, 128, "M", 256, 256, 256):     # torchvision cfg 'D', first 16 modules
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.vgg_model = nn.Sequential(*layers)
        for p in self.vgg_model.parameters():
            p.requires_grad = False

    def output_features(self, x):
        """x: NHWC fp32 image -> [relu1_2, relu2_2, relu3_3] as NHWC bf16."""
        feats, h, i, mods = [], x, 0, list(self.vgg_model)
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d):                              # conv + the ReLU that follows it, fused
                if m.in_channels == 3:
                    h = A.conv2d_small(h, m.weight, m.bias, layout="nhwc", act="relu")
                else:
                    h = A.conv2d(h, m.weight, m.bias, act="relu")
                i += 2
            else:
                h = A.maxpool2(h)
                i += 1
            if i - 1 in self.TAPS:
                feats.append(h)
        return feats

    def forward(self, dehaze, gt):
        f_sr = self.output_features(dehaze)
        with torch.no_grad():
            f_gt = self.output_features(gt)
        losses = [A.mse_loss(a, b) for a, b in zip(f_sr, f_gt)]
        return sum(losses) / len(losses)
