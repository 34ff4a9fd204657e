"""CPU: the integer-contract audit (oracle/audit.py) accepts the flips a latent perturbation explains and rejects a defective search."""
import numpy as np

from oracle.audit import flip_audit


def _nearest(a, E):
    d = (a.astype(np.float32) ** 2).sum(1, keepdims=True) + (E ** 2).sum(1)[None] - 2 * a @ E.T      # quantize.py:280-283, fp32
    return d.argmin(1)


def test_flips_explained_by_the_latent_error_pass():
    rng = np.random.default_rng(0)
    E = rng.normal(size=(8192, 3)).astype(np.float32)
    z = (rng.normal(size=(20000, 3)) * 0.8).astype(np.float32)
    for noise in (2e-3, 1e-5):
        zp = (z + rng.normal(size=z.shape).astype(np.float32) * noise).astype(np.float32)
        r = flip_audit(z, zp, _nearest(z, E), _nearest(zp, E), E)
        assert r["tokens"] == 20000 and not r["violations"] and r["worst_ratio"] <= 1.0, r
    assert flip_audit(z, z, _nearest(z, E), _nearest(z, E), E)["flips"] == 0


def test_a_defective_search_is_caught():
    rng = np.random.default_rng(1)
    E = rng.normal(size=(8192, 3)).astype(np.float32)
    z = (rng.normal(size=(5000, 3)) * 0.8).astype(np.float32)
    zp = (z + rng.normal(size=z.shape).astype(np.float32) * 1e-5).astype(np.float32)
    io, iu = _nearest(z, E), _nearest(zp, E)
    iu[:5] = (iu[:5] + 17) % 8192                      # five tokens sent to an arbitrary code
    r = flip_audit(z, zp, io, iu, E)
    assert r["violations"][:5] == [0, 1, 2, 3, 4] and r["worst_ratio"] > 10.0, r
