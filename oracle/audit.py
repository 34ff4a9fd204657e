"""TEST INFRASTRUCTURE (checker only; imported by tests/ and tools/, never by the product).

Integer-contract audit of the end-to-end codebook indices (reference: quantize.py:280-285,
`min_encoding_indices = torch.argmin(d, dim=1)` over d = |z|^2 + |e|^2 - 2 z.e).

The product's codebook search is bit-exact GIVEN the latent (tests/test_gpu_vq.py); end to end its latent z' differs from the
oracle's z by ~1e-5 relative, and a handful of tokens per image land in another code.  This module shows that every such flip
is one the latent error EXPLAINS -- a near-tie -- and not a defect of the search:

    ours picked e_u = argmin_e |z' - e|^2, the oracle e_o = argmin_e |z - e|^2, hence |z' - e_u|^2 <= |z' - e_o|^2, and
    |z - e_u|^2 - |z - e_o|^2 = (|z' - e_u|^2 - |z' - e_o|^2) + 2 (z' - z).(e_u - e_o)  <=  2 |z' - z| |e_u - e_o|.

So for every flipped token the oracle-side margin  m = d(z, e_u) - d(z, e_o)  (>= 0 up to the oracle's own fp32 rounding) must
satisfy  m <= 2 |z' - z| |e_u - e_o| + slack, evaluated here in fp64; `slack` covers the fp32 evaluation of d in BOTH searches
(the reference's formula cancels |z|^2 + |e|^2 against 2 z.e: a few ulp of that magnitude).  A flip that violates the bound
would mean the search itself chose a worse code than its own latent allows.
"""
import numpy as np


def flip_audit(z_oracle, z_ours, idx_oracle, idx_ours, codebook):
    """z_*: [N, D] latents (tokens x channels), idx_*: [N] chosen codes, codebook: [K, D].  Returns a dict:
    flips, worst_ratio (max over flips of margin / bound; <= 1 means every flip is explained), worst_margin, violations (list of
    token indices with ratio > 1), max_margin_rel (margin relative to the winning distance)."""
    z = np.asarray(z_oracle, dtype=np.float64)
    zp = np.asarray(z_ours, dtype=np.float64)
    io = np.asarray(idx_oracle).reshape(-1).astype(np.int64)
    iu = np.asarray(idx_ours).reshape(-1).astype(np.int64)
    E = np.asarray(codebook, dtype=np.float64)
    assert z.shape == zp.shape and z.shape[0] == io.shape[0] == iu.shape[0] and z.shape[1] == E.shape[1]
    flips = np.nonzero(io != iu)[0]
    out = {"tokens": int(io.shape[0]), "flips": int(flips.size), "worst_ratio": 0.0, "worst_margin": 0.0, "violations": [],
           "max_margin_rel": 0.0}
    if flips.size == 0:
        return out
    zf, zpf, eo, eu = z[flips], zp[flips], E[io[flips]], E[iu[flips]]
    d_u = ((zf - eu) ** 2).sum(1)
    d_o = ((zf - eo) ** 2).sum(1)
    margin = d_u - d_o
    dz = np.sqrt(((zpf - zf) ** 2).sum(1))
    de = np.sqrt(((eu - eo) ** 2).sum(1))
    # fp32 evaluation of |z|^2 + |e|^2 - 2 z.e in both searches: <= ~4 ulp (2^-24 each) of the terms' magnitude, twice
    mag = (zf ** 2).sum(1) + np.maximum((eo ** 2).sum(1), (eu ** 2).sum(1)) + 2 * np.abs((zf * eo).sum(1))
    slack = 16.0 * 2.0 ** -24 * mag
    bound = 2.0 * dz * de + slack
    ratio = margin / bound
    out["worst_ratio"] = float(ratio.max())
    out["worst_margin"] = float(margin.max())
    out["max_margin_rel"] = float((margin / np.maximum(d_o, 1e-300)).max())
    out["violations"] = [int(t) for t in flips[ratio > 1.0]]
    return out
