"""CPU: the metric restatements of the evaluation loop -- LPIPS (AlexNet) as `Measure.lpips` calls it
(Measure.py:17-30) and utils2.calculate_ssim (utils2.py:42-89).

LPIPS: the `lpips` package is a third-party dependency absent from /root/reference (and from this image): oracle.torch_ref.LPIPSAlex
restates its published forward; here its surface (state-dict keys = the product module's, shapes) and the metric's properties.
SSIM: the reference's OWN utils2.calculate_ssim / ssim are executed in place (build container only) with `cv2` reduced to the two
primitives the function uses, each implemented from cv2's documented definition (getGaussianKernel: G_i = a exp(-(i - (k-1)/2)^2 /
(2 sigma^2)), sum 1; filter2D: correlation with the kernel anchored at its centre, BORDER_REFLECT_101) -- the glue (float64, the
[5:-5, 5:-5] crop, C1 / C2 on 0-255, per-channel mean, the `border` argument, 2-D and 1-channel inputs) is then the reference's code
itself, and oracle.torch_ref.ssim_utils2 + the committed fixture are held against it."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import refimport as R
from oracle import torch_ref as O


def _seed_lpips_(m, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.startswith("lins."):
                continue
            if ".model." in name:                       # the 1x1 heads: non-negative, as the trained package's
                p.copy_(torch.rand(p.shape, generator=g) * 0.2)
            elif name.endswith("weight"):
                fan = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / fan) ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return m


def test_lpips_oracle_surface_and_properties():
    from glare_amd import metrics as MX

    o = _seed_lpips_(O.LPIPSAlex())
    prod = MX.LPIPS()
    assert list(o.state_dict().keys()) == list(prod.state_dict().keys())
    prod.load_state_dict(o.state_dict(), strict=True)
    keys = set(o.state_dict().keys())
    for k in ("scaling_layer.shift", "net.slice1.0.weight", "net.slice2.3.bias", "net.slice3.6.weight", "net.slice4.8.weight",
              "net.slice5.10.bias", "lin0.model.1.weight", "lins.4.model.1.weight"):
        assert k in keys, k
    assert tuple(o.state_dict()["net.slice1.0.weight"].shape) == (64, 3, 11, 11)
    assert tuple(o.state_dict()["lin2.model.1.weight"].shape) == (1, 384, 1, 1)
    g = torch.Generator().manual_seed(1)
    x, y = torch.rand(2, 3, 70, 90, generator=g) * 2 - 1, torch.rand(2, 3, 70, 90, generator=g) * 2 - 1
    with torch.no_grad():
        dxy, dyx, dxx = o(x, y), o(y, x), o(x, x)
        d01 = o((x + 1) / 2, (y + 1) / 2, normalize=True)
    assert dxy.shape == (2, 1, 1, 1)
    assert torch.equal(dxx, torch.zeros_like(dxx))
    assert torch.allclose(dxy, dyx, rtol=1e-6, atol=0) and (dxy > 0).all()
    assert torch.allclose(dxy, d01, rtol=1e-4)
    img = (np.random.RandomState(0).rand(8, 9, 3) * 255).astype(np.uint8)
    t = O.lpips_input(img)                               # Measure.t: /127.5 - 1, NCHW
    assert t.shape == (1, 3, 8, 9) and float(t.min()) >= -1 and float(t.max()) <= 1
    assert abs(float(t[0, 1, 2, 3]) - (img[2, 3, 1] / 127.5 - 1)) < 1e-6


def _cv2_documented():
    """The two cv2 primitives utils2.ssim uses, from OpenCV's documentation."""
    cv2 = types.ModuleType("cv2")

    def getGaussianKernel(ksize, sigma, ktype=None):
        i = np.arange(ksize, dtype=np.float64)
        g = np.exp(-((i - (ksize - 1) / 2.0) ** 2) / (2.0 * sigma ** 2))
        return (g / g.sum()).reshape(ksize, 1)

    def filter2D(src, ddepth, kernel):
        assert ddepth == -1 and src.ndim == 2
        kh, kw = kernel.shape
        ay, ax = kh // 2, kw // 2                         # default anchor: the kernel centre
        p = np.pad(src.astype(np.float64), ((ay, kh - 1 - ay), (ax, kw - 1 - ax)), mode="reflect")   # BORDER_REFLECT_101
        out = np.zeros(src.shape, dtype=np.float64)
        for y in range(kh):
            for x in range(kw):
                out += kernel[y, x] * p[y:y + src.shape[0], x:x + src.shape[1]]      # correlation
        return out

    cv2.getGaussianKernel, cv2.filter2D = getGaussianKernel, filter2D
    cv2.__getattr__ = lambda name: 0          # constants named in default arguments elsewhere in the file (COLORMAP_JET, ...): never used here
    return cv2


@pytest.mark.reference
@pytest.mark.skipif(not R.available(), reason="reference not present (build container only)")
def test_ssim_restatement_against_the_references_own_function(golden):
    path = os.path.join(R.REF_CODE, "utils", "utils2.py")
    saved = {k: sys.modules.get(k) for k in ("cv2", "skimage", "skimage.metrics", "natsort")}
    sys.modules["cv2"] = _cv2_documented()
    for k in ("skimage", "skimage.metrics", "natsort"):
        if k not in sys.modules:
            sys.modules[k] = types.ModuleType(k)
    try:
        import importlib.util

        spec = importlib.util.spec_from_file_location("ref_utils2", path)
        U = importlib.util.module_from_spec(spec)
        try:
            spec.loader.exec_module(U)
        except Exception as e:          # an import of the file's we cannot satisfy: say so instead of passing vacuously
            pytest.skip("utils2.py not importable here: %r" % (e,))
        g = golden("ssim_metric")
        tgt, res = g["target"], g["restored"]
        ref = U.calculate_ssim(tgt, res)
        assert abs(ref - O.ssim_utils2(tgt, res)) < 1e-12
        assert abs(ref - float(g["ssim"])) < 5e-6                       # the fixture pytorch_msssim produced (float32 arithmetic)
        rng = np.random.RandomState(11)
        a = rng.randint(0, 256, size=(33, 47, 3)).astype(np.uint8)
        b = np.clip(a.astype(np.int32) + rng.randint(-30, 31, size=a.shape), 0, 255).astype(np.uint8)
        assert abs(U.calculate_ssim(a, b) - O.ssim_utils2(a, b)) < 1e-12
        assert abs(U.calculate_ssim(a, a) - 1.0) < 1e-12
        # the `border` argument and the 2-D branch are the reference's own slicing around the same map
        assert abs(U.calculate_ssim(a, b, border=4) - O.ssim_utils2(a[4:-4, 4:-4], b[4:-4, 4:-4])) < 1e-12
        one = O.ssim_utils2(np.repeat(a[:, :, :1], 3, 2), np.repeat(b[:, :, :1], 3, 2))
        assert abs(U.calculate_ssim(a[:, :, 0], b[:, :, 0]) - one) < 1e-12
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
