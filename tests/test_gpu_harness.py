"""Device-side harness (row a14 / f2, csrc/harness.hip) against the host restatement (glare_amd/harness.py), which is itself
pinned to the reference's vectors (tests/golden/harness.npz, tests/test_harness.py)."""
import numpy as np
import pytest
import torch

from glare_amd import harness

pytestmark = pytest.mark.gpu


def test_preprocess_device_matches_reference_vector(golden):
    g = golden("harness")
    img = torch.from_numpy(g["img"][None]).cuda()
    got = harness.preprocess_device(img).cpu().numpy()
    np.testing.assert_allclose(got, g["pre"], rtol=3e-7, atol=3e-7)       # one ulp of logf


@pytest.mark.parametrize("B,H,W", [(3, 21, 37), (2, 400, 600)])
def test_preprocess_device_matches_host(B, H, W):
    rng = np.random.default_rng(H)
    imgs = rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    got = harness.preprocess_device(torch.from_numpy(imgs).cuda()).cpu()
    ref = harness.preprocess_batch(imgs)
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, rtol=3e-7, atol=3e-7)


@pytest.mark.parametrize("B,h,w", [(3, 20, 36), (2, 400, 600)])
def test_postprocess_device_gain_and_psnr(B, h, w):
    rng = np.random.default_rng(w)
    out = torch.from_numpy(rng.normal(0.45, 0.4, size=(B, 3, h + 20, w + 20)).astype(np.float32))
    gts = rng.integers(0, 256, size=(B, h, w, 3), dtype=np.uint8)
    restored, ps = harness.postprocess_device(out.cuda(), h, w, torch.from_numpy(gts).cuda())
    for i in range(B):
        ref = harness.postprocess(out[i:i + 1], h, gts[i])
        np.testing.assert_allclose(restored[i].cpu().numpy(), ref, rtol=2e-6, atol=2e-6)
        assert abs(float(ps[i]) - harness.psnr(gts[i] / 255.0, ref)) < 1e-4
    plain, none = harness.postprocess_device(out.cuda(), h, w)              # no GT: crop + clamp only
    assert none is None
    np.testing.assert_array_equal(plain[0].cpu().numpy(), harness.postprocess(out[0:1], h))


def test_harness_device_rejects_cpu():
    with pytest.raises(NotImplementedError):
        harness.preprocess_device(torch.zeros(1, 8, 8, 3, dtype=torch.uint8))


def test_ssim_metric_on_device(golden):
    """Row f4: the evaluation loop's SSIM (calculate_ssim on img_as_ubyte images, infer_dataset_lol.py:152) on the device against
    the reference-generated value (tests/golden/ssim_metric.npz) and the float64 oracle, batched, incl. the rounding of a float
    restored image to uint8."""
    import numpy as np
    import torch

    from glare_amd import harness
    from oracle import torch_ref as O

    g = golden("ssim_metric")
    tgt, res = g["target"], g["restored"]
    rng = np.random.RandomState(3)
    res2 = np.clip(tgt.astype(np.float64) + rng.randn(*tgt.shape) * 40.0, 0, 255) / 255.0       # a float image: rounded inside
    restored = torch.from_numpy(np.stack([res / 255.0, res2, tgt / 255.0]).astype(np.float32)).cuda()
    gts = torch.from_numpy(np.stack([tgt, tgt, tgt])).cuda()
    got = harness.ssim_device(restored, gts).cpu().numpy()
    assert abs(got[0] - float(g["ssim"])) < 5e-6
    assert abs(got[0] - O.ssim_utils2(tgt, res)) < 5e-6
    res2_u8 = np.rint(np.clip(res2.astype(np.float32), 0, 1) * 255).astype(np.uint8)
    assert abs(got[1] - O.ssim_utils2(tgt, res2_u8)) < 5e-6
    assert abs(got[2] - 1.0) < 1e-6
