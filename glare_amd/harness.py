"""Pre/post-processing of the inference harness (host side, as in the reference).

Mirrors code/infer_dataset_lol.py:113-153: reflect-pad 20 px at the bottom and the left (`impad`,
:71-72), HWC uint8 -> NCHW float /255 (`t`, :42), log(clamp(x + 1e-3, min=1e-3)) (:127-128,
`log_low: true`); after the network: crop `[:, :, :h, 20:]`, clamp to [0,1], GT-mean gain with the
cv2.COLOR_BGR2GRAY weights applied to RGB-ordered data (:142-144), PSNR (utils/utils2.py:32-36)."""
import math

import numpy as np
import torch

PAD = 20


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
