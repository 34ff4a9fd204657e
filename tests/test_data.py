"""Row f2 (host side): the LOL dataset loader's folder layout, sample dict, transforms and random-number call order
(code/data/LoL_dataset.py:409-502,615-661), on a synthetic folder."""
import os

import numpy as np
import torch
from PIL import Image

from glare_amd import data as D


def _make(root, split, n, h=40, w=56):
    rng = np.random.default_rng(0)
    for sub in ("low", "high"):
        os.makedirs(os.path.join(root, split, sub), exist_ok=True)
    imgs = []
    for i in range(n):
        low = rng.integers(0, 60, size=(h, w, 3), dtype=np.uint8)
        high = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        Image.fromarray(low).save(os.path.join(root, split, "low", "%d.png" % i))
        Image.fromarray(high).save(os.path.join(root, split, "high", "%d.png" % i))
        imgs.append((low, high))
    open(os.path.join(root, split, "low", "notes.txt"), "w").write("ignored")
    return imgs


def test_eval_split_is_untransformed_log_domain(tmp_path):
    imgs = _make(str(tmp_path), "eval15", 3)
    ds = D.LoL_Dataset({"root": str(tmp_path), "log_low": True}, train=False)
    assert len(ds) == 3
    for k in range(3):
        s = ds[k]
        i = int(s["LQ_path"])
        assert set(s) == {"LQ", "GT", "LQ_path", "GT_path"} and s["GT_path"] == s["LQ_path"]
        low, high = imgs[i]
        ref_lq = torch.log(torch.clamp(torch.from_numpy(low.transpose(2, 0, 1).copy()).float() / 255 + 1e-3, min=1e-3))
        assert torch.equal(s["LQ"], ref_lq)
        assert torch.equal(s["GT"], torch.from_numpy(high.transpose(2, 0, 1).copy()).float() / 255)


def test_train_split_draws_crop_flip_rotation_in_the_reference_order(tmp_path):
    imgs = _make(str(tmp_path), "our485", 1)
    ds = D.LoL_Dataset({"root": str(tmp_path), "use_crop": True, "use_flip": True, "use_rot": True, "GT_size": 32}, train=True)
    np.random.seed(123)
    s = ds[0]
    # replay the same draws: rows, columns (random_crop :645-646), flip (:616), rotation (:632)
    np.random.seed(123)
    low, high = imgs[0]
    x0 = np.random.randint(0, 40 - 32 + 1)
    y0 = np.random.randint(0, 56 - 32 + 1)
    keep = np.random.choice([True, False])
    k = np.random.choice([0, 1, 3])
    lo, hi = low[x0:x0 + 32, y0:y0 + 32], high[x0:x0 + 32, y0:y0 + 32]
    if not keep:
        lo, hi = np.flip(lo, 1), np.flip(hi, 1)
    lo, hi = np.rot90(lo, k, axes=(0, 1)), np.rot90(hi, k, axes=(0, 1))
    assert torch.equal(s["LQ"], torch.from_numpy(lo.transpose(2, 0, 1).copy()).float() / 255)
    assert torch.equal(s["GT"], torch.from_numpy(hi.transpose(2, 0, 1).copy()).float() / 255)
    assert s["LQ"].shape == (3, 32, 32)
