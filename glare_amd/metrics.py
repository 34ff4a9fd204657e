"""Evaluation metrics of the reference's dataset loops on the device: LPIPS (AlexNet) as `Measure.lpips` computes it
(code/Measure.py:17-30, called from infer_dataset_lol.py:153 / infer_dataset_lolv2-real.py:150 / test_stage3.py:205).  PSNR and the
MATLAB-style SSIM (utils2.py:32-36, 42-89) live in glare_amd.harness.

The reference builds `lpips.LPIPS(net='alex')` from the third-party `lpips` package (Zhang et al., "The Unreasonable Effectiveness of Deep
Features as a Perceptual Metric", CVPR 2018; PyPI lpips 0.1.x -- unpinned by the reference, absent from this image together with the
torchvision AlexNet weights and the package's linear heads it downloads).  `LPIPS` below mirrors that module's SURFACE -- constructor
default, `forward(in0, in1, normalize=False)` -> [N, 1, 1, 1], and its state-dict keys (`scaling_layer.shift / .scale`,
`net.slice{1..5}.{0,3,6,8,10}.weight / .bias` = torchvision alexnet.features indices, `lin{0..4}.model.1.weight` and their `lins.{k}` aliases)
so that a checkpoint saved from the package loads with `load_state_dict`; offline the weights are whatever the caller initialises
(glare_amd.synthetic.seeded_init_ in the tests).  Compute: csrc/metrics.hip through the C ABI (fp32 as the package; no CPU path).
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .modules._base import HipModule

_i, _ll, _f = ctypes.c_int, ctypes.c_longlong, ctypes.c_float
ALEX_CHNS = (64, 192, 384, 256, 256)                                    # lpips: self.chns for net='alex'
# (index in torchvision's alexnet.features, cin, cout, kernel, stride, padding) of the conv that ENDS each slice (ReLU follows; slices
# 2 and 3 start with MaxPool2d(3, 2))
ALEX_CONVS = ((0, 3, 64, 11, 4, 2), (3, 64, 192, 5, 1, 2), (6, 192, 384, 3, 1, 1), (8, 384, 256, 3, 1, 1), (10, 256, 256, 3, 1, 1))


def conv2d_direct(x, weight, bias=None, stride=1, padding=0, relu=False, in_shift=None, in_scale=None):
    """fp32 NCHW direct convolution (glare_conv2d_direct_f32): any kernel <= 16, stride, zero padding; optional per-channel input
    affine (x - shift) / scale applied before the padding."""
    _lib.require_cuda(x, weight, bias, in_shift, in_scale)
    x, weight = x.float().contiguous(), weight.detach().float().contiguous()
    B, Cin, H, W = x.shape
    Cout, cin2, k, k2 = weight.shape
    assert cin2 == Cin and k == k2
    OH, OW = (H + 2 * padding - k) // stride + 1, (W + 2 * padding - k) // stride + 1
    out = torch.empty(B, Cout, OH, OW, dtype=torch.float32, device=x.device)
    b = None if bias is None else bias.detach().float().contiguous()
    sh = None if in_shift is None else in_shift.detach().float().reshape(-1).contiguous()
    sc = None if in_scale is None else in_scale.detach().float().reshape(-1).contiguous()
    _lib.check(_lib.main_lib().glare_conv2d_direct_f32(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(b), _lib.ptr(out), _i(B), _i(Cin), _i(H), _i(W),
                                                       _i(Cout), _i(k), _i(stride), _i(padding), _i(int(relu)), _lib.ptr(sh), _lib.ptr(sc),
                                                       _lib.stream_handle()), "glare_conv2d_direct_f32")
    return out


def maxpool2d(x, kernel_size, stride):
    _lib.require_cuda(x)
    x = x.float().contiguous()
    B, C, H, W = x.shape
    out = torch.empty(B, C, (H - kernel_size) // stride + 1, (W - kernel_size) // stride + 1, dtype=torch.float32, device=x.device)
    _lib.check(_lib.main_lib().glare_maxpool2d_f32(_lib.ptr(x), _lib.ptr(out), _i(B), _i(C), _i(H), _i(W), _i(kernel_size), _i(stride),
                                                   _lib.stream_handle()), "glare_maxpool2d_f32")
    return out


class ScalingLayer(nn.Module):
    """lpips.ScalingLayer: (inp - shift) / scale with the package's constants (applied inside the first conv's loader here)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.Tensor([.458, .448, .450])[None, :, None, None])


class NetLinLayer(nn.Module):
    """lpips.NetLinLayer: Dropout (identity in eval) + 1x1 conv C -> 1 without bias; the conv is `model.1`."""

    def __init__(self, chn_in, chn_out=1, use_dropout=True):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)


class _AlexSlices(nn.Module):
    """lpips.pretrained_networks.alexnet: torchvision's alexnet.features cut into five slices, modules keyed by their index there."""

    def __init__(self):
        super().__init__()
        for s, (idx, cin, cout, k, st, pd) in enumerate(ALEX_CONVS):
            seq = nn.Sequential()
            if s in (1, 2):
                seq.add_module(str(idx - 1), nn.MaxPool2d(kernel_size=3, stride=2))
            seq.add_module(str(idx), nn.Conv2d(cin, cout, kernel_size=k, stride=st, padding=pd))
            seq.add_module(str(idx + 1), nn.ReLU(inplace=True))
            setattr(self, "slice%d" % (s + 1), seq)


class LPIPS(HipModule):
    """`lpips.LPIPS(net='alex', version='0.1', lpips=True, spatial=False)` on the HIP kernels."""

    def __init__(self, net="alex", version="0.1", use_dropout=True):
        super().__init__()
        if net not in ("alex", "alexnet") or version != "0.1":
            raise NotImplementedError("glare_amd.metrics.LPIPS implements net='alex', version='0.1' (Measure.py:20)")
        self.pnet_type, self.version, self.chns, self.L = net, version, list(ALEX_CHNS), len(ALEX_CHNS)
        self.scaling_layer = ScalingLayer()
        self.net = _AlexSlices()
        for k, c in enumerate(ALEX_CHNS):
            setattr(self, "lin%d" % k, NetLinLayer(c, use_dropout=use_dropout))
        self.lins = nn.ModuleList([getattr(self, "lin%d" % k) for k in range(self.L)])     # the package registers both names
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    def features(self, x):
        """The five ReLU taps of one batch (fp32 NCHW in [-1, 1] after `normalize`)."""
        feats, h = [], x
        for s, (idx, cin, cout, k, st, pd) in enumerate(ALEX_CONVS):
            conv = getattr(getattr(self.net, "slice%d" % (s + 1)), str(idx))
            if s in (1, 2):
                h = maxpool2d(h, 3, 2)
            if s == 0:
                h = conv2d_direct(h, conv.weight, conv.bias, st, pd, relu=True, in_shift=self.scaling_layer.shift, in_scale=self.scaling_layer.scale)
            else:
                h = conv2d_direct(h, conv.weight, conv.bias, st, pd, relu=True)
            feats.append(h)
        return feats

    @torch.no_grad()
    def forward(self, in0, in1, retPerLayer=False, normalize=False):
        _lib.require_cuda(in0, in1)
        if normalize:                      # images in [0, 1] -> [-1, 1]
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        f0, f1 = self.features(in0.float()), self.features(in1.float())
        B = in0.shape[0]
        lib = _lib.main_lib()
        lib.glare_lpips_tap_workspace_bytes.restype = ctypes.c_size_t
        total = torch.empty(B, dtype=torch.float64, device=in0.device)
        per_layer = []
        for k in range(self.L):
            a, b = f0[k], f1[k]
            C, HW = a.shape[1], a.shape[2] * a.shape[3]
            nws = lib.glare_lpips_tap_workspace_bytes(_i(B), _ll(HW))
            ws = torch.empty(nws, dtype=torch.uint8, device=a.device)
            lin = self.lins[k].model[-1].weight.detach().float().reshape(-1).contiguous()
            dst = torch.empty(B, dtype=torch.float64, device=a.device) if retPerLayer else total
            _lib.check(lib.glare_lpips_tap_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(lin), _i(B), _i(C), _ll(HW), _f(1e-10),
                                               _i(int(k > 0 and not retPerLayer)), _lib.ptr(dst), _lib.ptr(ws), ctypes.c_size_t(nws),
                                               _lib.stream_handle()), "glare_lpips_tap_f32")
            if retPerLayer:
                per_layer.append(dst.float().view(B, 1, 1, 1))
        if retPerLayer:
            val = sum(per_layer)
            return val, per_layer
        return total.float().view(B, 1, 1, 1)


def to_lpips_input(imgs_u8):
    """Measure.t (Measure.py:48-64): uint8 HWC image(s) on the device -> fp32 NCHW in [-1, 1] (x / 127.5 - 1)."""
    if imgs_u8.dim() == 3:
        imgs_u8 = imgs_u8[None]
    assert imgs_u8.dtype == torch.uint8 and imgs_u8.shape[-1] == 3
    return imgs_u8.permute(0, 3, 1, 2).float() / 127.5 - 1


class Measure:
    """Measure (Measure.py:17-30), the LPIPS leg: `lpips(imgA, imgB)` on uint8 HWC images (numpy or device tensors) -> float."""

    def __init__(self, net="alex", use_gpu=True, model=None):
        if not use_gpu:
            raise NotImplementedError("glare_amd metrics run on MI355X only (HIP kernels, no CPU path)")
        self.device = "cuda"
        self.model = (model if model is not None else LPIPS(net=net)).to(self.device)

    def lpips(self, imgA, imgB, model=None):
        tA, tB = (to_lpips_input(torch.as_tensor(np.ascontiguousarray(im) if isinstance(im, np.ndarray) else im).to(self.device)) for im in (imgA, imgB))
        return self.model.forward(tA, tB).item()
