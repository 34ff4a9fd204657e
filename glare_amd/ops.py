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
