"""Thin Python front-ends of the C ABI (include/glare_hip.h): argument checking, output allocation
with torch (device memory + current stream are torch's: plumbing, not compute), one ctypes call.
Every function raises on CPU tensors -- the HIP kernels are the only implementation."""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr, require_cuda, stream_handle

_i = ctypes.c_int
_ll = ctypes.c_longlong


def vq_nearest(z_tokens, codebook, want_zq=True):
    """z_tokens [N,3] fp32, codebook [K,3] fp32 -> (idx int64 [N], zq [N,3] or None).
    Restates quantize.py:280-285; indices bit-exact (see csrc/vq.hip)."""
    require_cuda(z_tokens, codebook)
    assert z_tokens.dtype == torch.float32 and codebook.dtype == torch.float32
    z_tokens = z_tokens.contiguous()
    codebook = codebook.contiguous()
    n, dim = z_tokens.shape
    idx = torch.empty(n, dtype=torch.int64, device=z_tokens.device)
    zq = torch.empty_like(z_tokens) if want_zq else None
    check(_lib.lib().glare_vq_nearest_f32(ptr(z_tokens), ptr(codebook), _ll(n), _i(codebook.shape[0]), _i(dim),
                                          ptr(idx), ptr(zq), stream_handle()), "glare_vq_nearest_f32")
    return idx, zq


# ---- convolution --------------------------------------------------------------------------------
ACT = {"none": 0, "relu": 1, "sigmoid": 2, "swish": 3}
OUT_NHWC_BF16, OUT_NHWC_F32, OUT_PLANAR_F32, OUT_PLANAR_BF16 = 0, 1, 2, 3


class ConvDesc(ctypes.Structure):
    """Mirror of `glare_conv_desc` (include/glare_hip.h)."""
    _fields_ = [("in_", ctypes.c_void_p), ("in2", ctypes.c_void_p), ("weight_packed", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("B", _i), ("H", _i), ("W", _i),
                ("Cin", _i), ("in_pitch", _i), ("in_off", _i),
                ("Cin2", _i), ("in2_pitch", _i), ("in2_off", _i),
                ("Cout", _i), ("out_pitch", _i), ("out_off", _i),
                ("res_pitch", _i), ("res_off", _i),
                ("ksize", _i), ("stride", _i), ("upsample", _i), ("act", _i), ("out_mode", _i),
                ("plane_pitch", _ll)]


class PackedConv:
    """A conv filter packed once for the MFMA kernel (weights bf16 stage-ordered, bias fp32)."""

    def __init__(self, weight_oihw, bias=None):
        require_cuda(weight_oihw)
        w = weight_oihw.detach().float().contiguous()
        self.cout, self.cin, kh, kw = w.shape
        assert kh == kw and kh in (1, 3)
        self.ksize = kh
        lib = _lib.lib()
        lib.glare_conv2d_packed_weight_elems.restype = _ll
        n = lib.glare_conv2d_packed_weight_elems(_i(self.cout), _i(self.cin), _i(kh))
        assert n > 0
        self.packed = torch.empty(n, dtype=torch.bfloat16, device=w.device)
        check(lib.glare_conv2d_pack_weight(ptr(w), _i(self.cout), _i(self.cin), _i(kh), ptr(self.packed),
                                           stream_handle()), "glare_conv2d_pack_weight")
        self.bias = None if bias is None else bias.detach().float().contiguous()


def conv2d(x, pc, *, x2=None, cin=None, in_off=0, cin2=None, in2_off=0, stride=1, upsample=False, act="none",
           residual=None, res_off=0, out=None, out_off=0, out_mode=OUT_NHWC_BF16, plane_pitch=0):
    """x: NHWC bf16 [B,H,W,pitch] (channels [in_off, in_off+cin) are used), optional x2 concatenated
    after it.  Returns (or fills `out`) per out_mode; planar outputs are [B, planes, plane_pitch]."""
    require_cuda(x, x2, residual, out)
    assert x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous()
    B, H, W, pitch = x.shape
    cin = pitch - in_off if cin is None else cin
    d = ConvDesc()
    d.in_, d.in2 = x.data_ptr(), (x2.data_ptr() if x2 is not None else None)
    d.B, d.H, d.W = B, H, W
    d.Cin, d.in_pitch, d.in_off = cin, pitch, in_off
    if x2 is not None:
        assert x2.dtype == torch.bfloat16 and x2.is_contiguous() and x2.shape[:3] == x.shape[:3]
        d.Cin2 = x2.shape[3] - in2_off if cin2 is None else cin2
        d.in2_pitch, d.in2_off = x2.shape[3], in2_off
    assert pc.cin == d.Cin + d.Cin2, (pc.cin, d.Cin, d.Cin2)
    IH, IW = (2 * H, 2 * W) if upsample else (H, W)
    OH, OW = ((IH + 1 - 3) // 2 + 1, (IW + 1 - 3) // 2 + 1) if stride == 2 else (IH, IW)
    if out is None:
        if out_mode == OUT_NHWC_BF16:
            out = torch.empty(B, OH, OW, pc.cout, dtype=torch.bfloat16, device=x.device)
        elif out_mode == OUT_NHWC_F32:
            out = torch.empty(B, OH, OW, pc.cout, dtype=torch.float32, device=x.device)
        else:
            pp = plane_pitch or OH * OW
            dt = torch.float32 if out_mode == OUT_PLANAR_F32 else torch.bfloat16
            out = torch.zeros(B, pc.cout, pp, dtype=dt, device=x.device)
    assert out.is_contiguous()
    if out_mode in (OUT_NHWC_BF16, OUT_NHWC_F32):
        d.out_pitch = out.shape[3]
    else:
        d.out_pitch = out.shape[1]
        d.plane_pitch = out.shape[2]
    d.out, d.Cout, d.out_off = out.data_ptr(), pc.cout, out_off
    d.weight_packed = pc.packed.data_ptr()
    d.bias = pc.bias.data_ptr() if pc.bias is not None else None
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.is_contiguous()
        d.residual, d.res_pitch, d.res_off = residual.data_ptr(), residual.shape[3], res_off
    d.ksize, d.stride, d.upsample, d.act, d.out_mode = pc.ksize, stride, int(bool(upsample)), ACT[act], out_mode
    check(_lib.lib().glare_conv2d_bf16(ctypes.byref(d), stream_handle()), "glare_conv2d_bf16")
    return out
