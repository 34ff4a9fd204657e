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
