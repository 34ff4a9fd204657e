"""Pre/post-processing of the inference harness (host side, as in the reference).

Mirrors code/infer_dataset_lol.py:113-153: reflect-pad 20 px at the bottom and the left (`impad`,
:71-72), HWC uint8 -> NCHW float /255 (`t`, :42), log(clamp(x + 1e-3, min=1e-3)) (:127-128,
`log_low: true`); after the network: crop `[:, :, :h, 20:]`, clamp to [0,1], GT-mean gain with the
cv2.COLOR_BGR2GRAY weights applied to RGB-ordered data (:142-144), PSNR (utils/utils2.py:32-36)."""
import ctypes
import math

import numpy as np
import torch

PAD = 20


# ---- device versions (csrc/harness.hip): batched, uint8 in, PSNR out ------------------------------------------
def preprocess_device(imgs_u8):
    """uint8 device tensor [B,H,W,3] -> fp32 [B,3,H+20,W+20] in the log domain (same arithmetic as preprocess())."""
    from . import _lib

    _lib.require_cuda(imgs_u8)
    assert imgs_u8.dtype == torch.uint8 and imgs_u8.dim() == 4 and imgs_u8.shape[3] == 3 and imgs_u8.is_contiguous()
    B, H, W, _ = imgs_u8.shape
    out = torch.empty(B, 3, H + PAD, W + PAD, dtype=torch.float32, device=imgs_u8.device)
    _lib.check(_lib.lib().glare_harness_preprocess_u8(_lib.ptr(imgs_u8), ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W),
                                                      ctypes.c_int(PAD), _lib.ptr(out), _lib.stream_handle()),
               "glare_harness_preprocess_u8")
    return out


def postprocess_device(out_nchw, h, w, gts_u8=None, want_nonfinite=False):
    """network output [B,3,Hp,Wp] (device) -> (restored float [B,h,w,3], psnr float64 [B] or None), all on the device.
    want_nonfinite: a third result, int32 [B] = inf / NaN values of the network output inside each crop BEFORE the clamp (which maps
    +inf to 1.0 and would hide an fp16 overflow behind a finite PSNR)."""
    from . import _lib

    _lib.require_cuda(out_nchw, gts_u8)
    out_nchw = out_nchw.float().contiguous()
    B, _, Hp, Wp = out_nchw.shape
    lib = _lib.lib()
    lib.glare_harness_postprocess_workspace_bytes.restype = ctypes.c_size_t
    nws = lib.glare_harness_postprocess_workspace_bytes(ctypes.c_int(B))
    ws = torch.empty(nws, dtype=torch.uint8, device=out_nchw.device)
    restored = torch.empty(B, h, w, 3, dtype=torch.float32, device=out_nchw.device)
    psnr_t = None
    if gts_u8 is not None:
        assert gts_u8.dtype == torch.uint8 and tuple(gts_u8.shape) == (B, h, w, 3) and gts_u8.is_contiguous()
        psnr_t = torch.empty(B, dtype=torch.float64, device=out_nchw.device)
    i = ctypes.c_int
    bad = torch.empty(B, dtype=torch.int32, device=out_nchw.device) if want_nonfinite else None
    _lib.check(lib.glare_harness_postprocess_flagged_f32(_lib.ptr(out_nchw), _lib.ptr(gts_u8), i(B), i(h), i(w), i(Hp), i(Wp), i(PAD),
                                                         _lib.ptr(restored), _lib.ptr(psnr_t), _lib.ptr(bad), _lib.ptr(ws), ctypes.c_size_t(nws),
                                                         _lib.stream_handle()), "glare_harness_postprocess_flagged_f32")
    if want_nonfinite:
        return restored, psnr_t, bad
    return restored, psnr_t


def to_ubyte_device(restored):
    """restored float [..] in [0,1] (device) -> uint8 of the same shape: img_as_ubyte, the image the reference's loop saves."""
    from . import _lib

    _lib.require_cuda(restored)
    restored = restored.float().contiguous()
    out = torch.empty(restored.shape, dtype=torch.uint8, device=restored.device)
    _lib.check(_lib.main_lib().glare_harness_to_ubyte(_lib.ptr(restored), ctypes.c_longlong(restored.numel()), _lib.ptr(out),
                                                      _lib.stream_handle()), "glare_harness_to_ubyte")
    return out


def ssim_device(restored, gts_u8):
    """SSIM as the reference's evaluation loop computes it (calculate_ssim(img_as_ubyte(target), img_as_ubyte(restored)),
    utils2.py:42-89 / infer_dataset_lol.py:152): restored float [B,h,w,3] in [0,1] and the uint8 ground truth, both on the device
    -> float64 [B] (device).  11x11 Gaussian (sigma 1.5) window at the valid positions, per channel, averaged."""
    from . import _lib
    from . import train_ops as T

    _lib.require_cuda(restored, gts_u8)
    assert restored.dtype == torch.float32 and gts_u8.dtype == torch.uint8 and restored.shape == gts_u8.shape
    restored, gts_u8 = restored.contiguous(), gts_u8.contiguous()
    B = restored.shape[0]
    x, y = torch.empty_like(restored), torch.empty_like(restored)
    _lib.check(_lib.lib().glare_harness_ubyte_planes_f32(_lib.ptr(restored), _lib.ptr(gts_u8), ctypes.c_longlong(restored.numel()),
                                                         _lib.ptr(x), _lib.ptr(y), _lib.stream_handle()), "glare_harness_ubyte_planes_f32")
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    g = (g / g.sum()).tolist()                                   # cv2.getGaussianKernel(11, 1.5)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = [T.ssim_forward(x[b:b + 1], y[b:b + 1], g, C1, C2)[1][0:1] for b in range(B)]   # mean ssim_map of one image
    return torch.cat(vals).double()


def preprocess(img_u8):
    """uint8 [H,W,3] -> fp32 [1,3,H+20,W+20] in the log domain."""
    img = np.pad(img_u8, [(0, PAD), (PAD, 0), (0, 0)], "reflect")
    t = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)[None]).astype(np.float32)) / 255
    return torch.log(torch.clamp(t + 1e-3, min=1e-3))


def preprocess_batch(imgs_u8):
    return torch.cat([preprocess(im) for im in imgs_u8], dim=0)


def gray_mean(img):
    img = np.asarray(img, dtype=np.float32)
    return float((0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).mean())


def postprocess(out_nchw, h, gt_u8=None):
    """network output [1,3,H+20,W+20] -> float [H,W,3] in [0,1]; with gt: mean-gray gain first."""
    r = torch.clamp(out_nchw[:, :, :h, PAD:], 0, 1).detach().cpu().permute(0, 2, 3, 1).squeeze(0).numpy()
    if gt_u8 is not None:
        r = np.clip(r * (gray_mean(gt_u8 / 255) / gray_mean(r)), 0, 1)
    return r


def psnr(img1, img2):
    mse = float(np.mean((img1 - img2) ** 2))
    return 100.0 if mse == 0 else 10 * math.log10(1 / mse)
