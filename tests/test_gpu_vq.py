"""GPU: codebook retrieval kernel (csrc/vq.hip) through the C ABI against the C oracle and the
reference-generated golden vector.  Bar: bit-exact indices and z_q."""
import numpy as np
import pytest
import torch

from glare_amd import ops
from oracle import c_ref

pytestmark = pytest.mark.gpu


def _run(z, cb):
    idx, zq = ops.vq_nearest(torch.from_numpy(z).cuda(), torch.from_numpy(cb).cuda())
    torch.cuda.synchronize()
    return idx.cpu().numpy(), zq.cpu().numpy()


def test_golden_vector(golden):
    g = golden("vq")
    tokens = np.ascontiguousarray(g["z"].transpose(0, 2, 3, 1).reshape(-1, 3))
    idx, zq = _run(tokens, g["codebook"])
    assert np.array_equal(idx, g["idx"])  # includes exact ties (lowest index) and near ties
    ste = tokens + (zq - tokens)
    assert np.array_equal(ste.reshape(2, 8, 12, 3).transpose(0, 3, 1, 2), g["zq"])


@pytest.mark.parametrize("n,k", [(1, 8192), (511, 8192), (513, 8192), (4099, 1000), (300, 20000), (64, 1)])
def test_against_c_oracle(n, k):
    rng = np.random.RandomState(n + k)
    z = (rng.randn(n, 3) * 1.3).astype(np.float32)
    cb = (rng.randn(k, 3) * 0.7).astype(np.float32)
    idx, zq = _run(z, cb)
    ridx, rzq = c_ref.vq_nearest(z, cb)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(zq, rzq)


def test_empty_and_errors():
    cb = torch.zeros(8, 3, device="cuda")
    idx, zq = ops.vq_nearest(torch.zeros(0, 3, device="cuda"), cb)
    assert idx.numel() == 0
    with pytest.raises(Exception):
        ops.vq_nearest(torch.zeros(4, 4, device="cuda"), torch.zeros(8, 4, device="cuda"))  # dim != 3


def test_full_size_properties():
    """BASELINE config I8: 8 x 16 275 tokens.  Size-independent properties: every returned entry is
    the true nearest (distance to the chosen code <= distance to 64 random other codes), z_q is the
    indexed row, and re-quantising z_q is idempotent."""
    g = torch.Generator().manual_seed(1)
    z = torch.randn(8 * 16275, 3, generator=g).cuda()
    cb = (torch.randn(8192, 3, generator=g) * 0.7).cuda()
    idx, zq = ops.vq_nearest(z, cb)
    assert torch.equal(zq, cb[idx])
    d_best = ((z - zq) ** 2).sum(1)
    probe = torch.randint(0, 8192, (64,), generator=g).cuda()
    d_probe = ((z[:, None, :] - cb[probe][None]) ** 2).sum(2)
    assert bool((d_best[:, None] <= d_probe + 1e-4).all())
    idx2, _ = ops.vq_nearest(zq, cb)
    assert torch.equal(cb[idx2], zq)
