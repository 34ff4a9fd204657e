"""GPU: LPIPS (AlexNet) of the evaluation loop on the HIP kernels (glare_amd.metrics, csrc/metrics.hip) against the oracle's restatement of
the `lpips` package's published forward (oracle.torch_ref.LPIPSAlex) on seeded weights, and its building blocks against torch's fp32 ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from glare_amd import metrics as MX
from oracle import torch_ref as O
from test_metrics_oracle import _seed_lpips_

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [(3, 64, 11, 4, 2, 37, 53), (64, 192, 5, 1, 2, 12, 17), (192, 384, 3, 1, 1, 9, 11), (5, 7, 3, 2, 0, 16, 19),
                                 (8, 40, 1, 1, 0, 6, 70)])
def test_direct_conv_matches_torch_fp32(cfg):
    cin, cout, k, st, pd, H, W = cfg
    g = torch.Generator().manual_seed(cin * 100 + k)
    x = torch.randn(2, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, w, b, st, pd))
    got = MX.conv2d_direct(x.cuda(), w.cuda(), b.cuda(), st, pd, relu=True).cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err                        # fp32 both sides: summation order only (measured <= 5e-7)
    if cin == 3:                                  # the scaling layer in the loader: zero padding AFTER (x - shift) / scale
        sh, sc = torch.tensor([-.03, -.088, -.188]), torch.tensor([.458, .448, .45])
        ref2 = F.conv2d((x - sh.view(1, 3, 1, 1)) / sc.view(1, 3, 1, 1), w, None, st, pd)
        got2 = MX.conv2d_direct(x.cuda(), w.cuda(), None, st, pd, in_shift=sh.cuda(), in_scale=sc.cuda()).cpu()
        assert float((got2 - ref2).abs().max() / ref2.abs().max()) < 2e-6


def test_maxpool_matches_torch():
    x = torch.randn(2, 5, 23, 30)
    assert torch.equal(MX.maxpool2d(x.cuda(), 3, 2).cpu(), F.max_pool2d(x, 3, 2))


@pytest.mark.parametrize("shape", [(2, 64, 80), (1, 400, 600)])
def test_lpips_matches_oracle(shape):
    B, H, W = shape
    o = _seed_lpips_(O.LPIPSAlex())
    m = MX.LPIPS()
    m.load_state_dict(o.state_dict(), strict=True)
    m = m.cuda()
    rng = np.random.RandomState(H)
    a = rng.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.randint(-40, 41, size=a.shape), 0, 255).astype(np.uint8)
    ta, tb = MX.to_lpips_input(torch.from_numpy(a).cuda()), MX.to_lpips_input(torch.from_numpy(b).cuda())
    with torch.no_grad():
        ref = o(torch.cat([O.lpips_input(x) for x in a]), torch.cat([O.lpips_input(x) for x in b]))
    got = m(ta, tb).cpu()
    assert got.shape == (B, 1, 1, 1)
    err = float(((got - ref).abs() / ref.abs()).max())
    print("LPIPS %s: ours %s oracle %s rel %.2e" % (shape, got.flatten().tolist(), ref.flatten().tolist(), err))
    assert err < 1e-6, err                                               # fp32 both sides: measured 3.5e-7 / 0
    assert float(m(ta, ta).abs().max()) == 0.0                          # d(x, x) = 0
    val, per = m(ta, tb, retPerLayer=True)
    assert len(per) == 5 and torch.allclose(val.cpu(), got, rtol=1e-5)
    meas = MX.Measure(model=m)                                           # Measure.lpips on uint8 HWC images (Measure.py:25-29)
    assert abs(meas.lpips(a[0], b[0]) - float(ref[0])) < 2e-5 * abs(float(ref[0])) + 1e-9


def test_infer_lpips_column():
    from glare_amd import infer

    res = infer.run(2, batch=2, h=44, w=76, with_ssim=True, with_lpips=True)
    assert res.shape == (2, 3) and np.isfinite(res).all() and (res[:, 2] >= -1e-6).all()
