"""LOL dataset loader (host side): mirrors `LoL_Dataset` (code/data/LoL_dataset.py:409-502) and its helpers
`random_crop` (:640-661), `random_flip` (:615-621), `random_rotation` (:631-637) with the same folder layout
(`<root>/our485|eval15/{low,high}/*.png`), option keys, sample dict and random-number call sequence, so that a seeded run draws
the crops / flips / rotations the reference draws.  PNG decoding is PIL (RGB order, what the reference gets from
cv2.imread + COLOR_BGR2RGB); the histogram-equalised side input (`concat_histeq` / `histeq_as_input`, off in every shipped
config) is not built.  The tensors it yields feed glare_amd.train.{Stage2Trainer,Stage3Trainer}.step and glare_amd.infer."""
import os
import random

import numpy as np
import torch


def random_crop(hr, lr, size_hr):
    """Same square crop from both images; np.random.randint is called for rows, then columns (LoL_dataset.py:640-661)."""
    sx, sy = lr.shape[0], lr.shape[1]
    x0 = np.random.randint(low=0, high=(sx - size_hr) + 1) if sx > size_hr else 0
    y0 = np.random.randint(low=0, high=(sy - size_hr) + 1) if sy > size_hr else 0
    return hr[x0:x0 + size_hr, y0:y0 + size_hr, :], lr[x0:x0 + size_hr, y0:y0 + size_hr, :]


def center_crop(img, size):
    assert img.shape[0] == img.shape[1], img.shape      # the reference asserts on dims 1/2 of an HWC image (:667); same intent
    border = (img.shape[0] - size) // 2
    return img[border:img.shape[0] - border, border:img.shape[1] - border, :]


def random_flip(hr, lr):
    keep = np.random.choice([True, False])
    return (hr, lr) if keep else (np.flip(hr, 1).copy(), np.flip(lr, 1).copy())


def random_rotation(hr, lr):
    k = np.random.choice([0, 1, 3])
    return np.rot90(hr, k, axes=(0, 1)).copy(), np.rot90(lr, k, axes=(0, 1)).copy()


def to_tensor(img_u8):
    """torchvision ToTensor on a uint8 HWC array: CHW float32 / 255."""
    return torch.from_numpy(np.ascontiguousarray(img_u8.transpose(2, 0, 1))).float().div(255)


class LoL_Dataset(torch.utils.data.Dataset):
    def __init__(self, opt, train, all_opt=None):
        from PIL import Image

        all_opt = all_opt or {}
        assert not all_opt.get("concat_histeq", False) and not all_opt.get("histeq_as_input", False), "histogram-equalised input: not built"
        self.opt = opt
        self.log_low = opt.get("log_low", False)
        self.use_flip, self.use_rot, self.use_crop = opt.get("use_flip", False), opt.get("use_rot", False), opt.get("use_crop", False)
        self.use_noise = opt.get("noise_prob", False)
        self.noise_prob = opt["noise_prob"] if self.use_noise else None
        self.noise_level = opt.get("noise_level", 0)
        self.center_crop_hr_size = opt.get("center_crop_hr_size", None)
        self.crop_size = opt.get("GT_size", None)
        self.root = os.path.join(opt["root"], "our485" if train else "eval15")
        self.pairs = []
        for f_name in filter(lambda x: "png" in x, os.listdir(os.path.join(self.root, "low"))):
            low = np.asarray(Image.open(os.path.join(self.root, "low", f_name)).convert("RGB"))
            high = np.asarray(Image.open(os.path.join(self.root, "high", f_name)).convert("RGB"))
            self.pairs.append([low, high, f_name.split(".")[0]])

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, item):
        lr, hr, f_name = self.pairs[item]
        if self.use_crop:
            hr, lr = random_crop(hr, lr, self.crop_size)
        if self.center_crop_hr_size:
            hr, lr = center_crop(hr, self.center_crop_hr_size), center_crop(lr, self.center_crop_hr_size)
        if self.use_flip:
            hr, lr = random_flip(hr, lr)
        if self.use_rot:
            hr, lr = random_rotation(hr, lr)
        hr, lr = to_tensor(hr), to_tensor(lr)
        if self.use_noise and random.random() < self.noise_prob:
            lr = torch.randn(lr.shape) * (self.noise_level / 255) + lr
        if self.log_low:
            lr = torch.log(torch.clamp(lr + 1e-3, min=1e-3))
        return {"LQ": lr, "GT": hr, "LQ_path": f_name, "GT_path": f_name}
