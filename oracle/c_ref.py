"""TEST INFRASTRUCTURE -- ctypes front-end of the plain-C oracle (oracle/dcn_ref.c, oracle/vq_ref.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else ctypes.c_void_p(0)


def vq_nearest(z_tokens, codebook, return_d=False):
    """z_tokens [N,3], codebook [K,3] -> (idx int64 [N], zq [N,3] [, d [N,K]])"""
    z, cb = _f32(z_tokens), _f32(codebook)
    n, k = z.shape[0], cb.shape[0]
    idx = np.empty(n, dtype=np.int64)
    zq = np.empty((n, 3), dtype=np.float32)
    d = np.empty((n, k), dtype=np.float32) if return_d else None
    rc = lib().vq_ref_nearest(_p(z), _p(cb), ctypes.c_longlong(n), ctypes.c_int(k), _p(idx), _p(zq), _p(d))
    assert rc == 0
    return (idx, zq, d) if return_d else (idx, zq)


def _geom(x, weight, stride, padding, dilation, groups, dg):
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    args = [B, C, H, W, Co, kh, kw, stride, stride, padding, padding, dilation, dilation, groups, dg]
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    return [ctypes.c_int(a) for a in args], Ho, Wo


def dcn_forward(x, offset, mask, weight, bias=None, stride=1, padding=1, dilation=1, groups=1, dg=1):
    x, offset, mask, weight = _f32(x), _f32(offset), _f32(mask), _f32(weight)
    bias = _f32(bias) if bias is not None else None
    ints, Ho, Wo = _geom(x, weight, stride, padding, dilation, groups, dg)
    out = np.empty((x.shape[0], weight.shape[0], Ho, Wo), dtype=np.float32)
    rc = lib().dcn_ref_forward(_p(x), _p(offset), _p(mask), _p(weight), _p(bias), _p(out), *ints)
    assert rc == 0
    return out


def dcn_backward(x, offset, mask, weight, grad_out, with_bias=True, stride=1, padding=1, dilation=1, groups=1, dg=1):
    x, offset, mask, weight, grad_out = _f32(x), _f32(offset), _f32(mask), _f32(weight), _f32(grad_out)
    ints, _, _ = _geom(x, weight, stride, padding, dilation, groups, dg)
    gx, goff, gmask = np.zeros_like(x), np.zeros_like(offset), np.zeros_like(mask)
    gw = np.zeros_like(weight)
    gb = np.zeros(weight.shape[0], dtype=np.float32) if with_bias else None
    rc = lib().dcn_ref_backward(_p(x), _p(offset), _p(mask), _p(weight), _p(grad_out), _p(gx), _p(goff), _p(gmask),
                                _p(gw), _p(gb), *ints)
    assert rc == 0
    return gx, goff, gmask, gw, gb
