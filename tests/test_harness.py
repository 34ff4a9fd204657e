"""CPU: host-side pre/post-processing (glare_amd/harness.py) against the vectors captured from the
reference's own helpers (tests/golden/harness.npz) and against the oracle's restatement."""
import numpy as np
import torch

from glare_amd import harness
from oracle import torch_ref as O


def test_preprocess_matches_reference_vector(golden):
    g = golden("harness")
    got = harness.preprocess(g["img"]).numpy()
    np.testing.assert_array_equal(got, g["pre"])
    assert got.shape == (1, 3, 24 + 20, 28 + 20)


def test_psnr_and_postprocess(golden):
    g = golden("harness")
    assert abs(harness.psnr(g["a"], g["b"]) - float(g["psnr"])) < 1e-9
    assert harness.psnr(g["a"], g["a"]) == 100.0
    rng = np.random.RandomState(0)
    out = torch.from_numpy(rng.rand(1, 3, 44, 48).astype(np.float32) * 1.4 - 0.2)
    gt = rng.randint(0, 256, size=(24, 28, 3)).astype(np.uint8)
    np.testing.assert_allclose(harness.postprocess(out, 24, gt), O.postprocess(out, 24, gt), rtol=1e-6)
    r = harness.postprocess(out, 24)
    assert r.shape == (24, 28, 3) and r.min() >= 0 and r.max() <= 1
