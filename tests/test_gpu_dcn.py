"""GPU: DCNv2 forward (csrc/dcn.hip) through the drop-in C entry point against the plain-C oracle
(oracle/dcn_ref.c).  Both compute in fp32; they differ in summation order (oracle accumulates the
contraction in double) -> tolerance 1e-4 * max|ref| absolute, 1e-4 relative."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from glare_amd import ops
from oracle import c_ref

pytestmark = pytest.mark.gpu


def _case(seed, B, C, H, W, Co, dg, off_scale=2.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, dg * 18, H, W, generator=g) * off_scale
    m = torch.rand(B, dg * 9, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * (1.0 / (C * 9) ** 0.5)
    b = torch.randn(Co, generator=g)
    return x, off, m, w, b


def _close(got, ref):
    tol = 1e-4 * float(np.abs(ref).max())
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=tol)


@pytest.mark.parametrize("B,C,H,W,Co,dg", [(1, 128, 9, 13, 128, 4), (2, 256, 6, 11, 256, 4), (1, 128, 5, 70, 64, 4),
                                            (1, 64, 8, 8, 64, 2)])
def test_forward_matches_c_oracle(B, C, H, W, Co, dg):
    x, off, m, w, b = _case(C + H, B, C, H, W, Co, dg)
    got = ops.mdcn_forward(x.cuda(), off.cuda(), m.cuda(), w.cuda(), b.cuda(), 1, 1, 1, 1, dg).cpu().numpy()
    ref = c_ref.dcn_forward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), b.numpy(), dg=dg)
    _close(got, ref)


def test_large_offsets_borders_and_no_bias():
    x, off, m, w, b = _case(3, 1, 128, 7, 9, 128, 4, off_scale=6.0)  # many samples leave the image
    off[0, 0] = -1.0   # exactly on the h = -1 boundary for tap 0 of group 0 at row 0 (h_im = -2 .. )
    got = ops.mdcn_forward(x.cuda(), off.cuda(), m.cuda(), w.cuda(), None, 1, 1, 1, 1, 4).cpu().numpy()
    ref = c_ref.dcn_forward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), None, dg=4)
    _close(got, ref)


def test_zero_offset_equals_conv2d_and_stride_dilation():
    x, off, m, w, b = _case(4, 1, 128, 10, 12, 128, 4)
    got = ops.mdcn_forward(x.cuda(), torch.zeros_like(off).cuda(), torch.ones_like(m).cuda(), w.cuda(), b.cuda(), 1, 1, 1, 1, 4)
    ref = F.conv2d(x.cuda(), w.cuda(), b.cuda(), 1, 1)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4)
    # stride 2 / dilation 2 geometry of the C ABI (v2 passes H before W, deform_conv.py:150-151)
    Ho = (10 + 2 * 2 - (2 * 2 + 1)) // 2 + 1
    Wo = (12 + 2 * 2 - (2 * 2 + 1)) // 2 + 1
    off2 = torch.randn(1, 72, Ho, Wo) * 1.5
    m2 = torch.rand(1, 36, Ho, Wo)
    got = ops.mdcn_forward(x.cuda(), off2.cuda(), m2.cuda(), w.cuda(), b.cuda(), 2, 2, 2, 1, 4).cpu().numpy()
    ref = c_ref.dcn_forward(x.numpy(), off2.numpy(), m2.numpy(), w.numpy(), b.numpy(), stride=2, padding=2, dilation=2, dg=4)
    _close(got, ref)


def test_nhwc_bf16_pipeline_entry():
    """The pipeline form: x bf16 NHWC inside a wider record, offsets + mask LOGITS in one planar buffer."""
    x, off, m, w, b = _case(5, 2, 128, 8, 10, 128, 4)
    xb = x.to(torch.bfloat16)
    logits = torch.randn(2, 36, 8, 10)
    om = torch.zeros(2, 108, 8 * 10 + 16)
    om[:, :72, :80] = off.reshape(2, 72, 80)
    om[:, 72:, :80] = logits.reshape(2, 36, 80)
    rec = torch.zeros(2, 8, 10, 256, dtype=torch.bfloat16)
    rec[..., 128:] = xb.permute(0, 2, 3, 1)
    pd = ops.PackedDcn(w.cuda(), b.cuda(), 4)
    got = ops.mdcn_forward_nhwc(rec.cuda(), om.cuda(), pd, x_off=128, C=128).permute(0, 3, 1, 2).cpu().numpy()
    ref = c_ref.dcn_forward(xb.float().numpy(), off.numpy(), torch.sigmoid(logits).numpy(), w.numpy(), b.numpy(), dg=4)
    _close(got, ref)


def test_errors():
    x, off, m, w, b = _case(6, 1, 24, 4, 4, 64, 4)  # cpg = 6: unsupported, must say so (not crash)
    with pytest.raises(Exception, match="unsupported"):
        ops.mdcn_forward(x.cuda(), off.cuda(), m.cuda(), w.cuda(), b.cuda(), 1, 1, 1, 1, 4)
    with pytest.raises(NotImplementedError):
        ops.mdcn_forward(x, off, m, w, b, 1, 1, 1, 1, 4)  # CPU tensors: no fallback
