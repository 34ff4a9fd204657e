"""CPU: pins the DCNv2 oracle (oracle/dcn_ref.c and the torch formulation in oracle/torch_ref.py)
by identities, because the reference's CUDA-only extension cannot execute in the build image
("parity unpinned by execution")."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import c_ref
from oracle import torch_ref as O


def _case(seed=0, B=2, C=8, H=7, W=9, Co=6, dg=4, off_scale=2.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, dg * 18, H, W, generator=g) * off_scale
    m = torch.rand(B, dg * 9, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.2
    b = torch.randn(Co, generator=g)
    return x, off, m, w, b, dg


def test_zero_offset_is_conv2d():
    x, off, m, w, b, dg = _case()
    ref = F.conv2d(x, w, b, 1, 1).numpy()
    got = c_ref.dcn_forward(x.numpy(), np.zeros_like(off.numpy()), np.ones_like(m.numpy()), w.numpy(), b.numpy(), dg=dg)
    np.testing.assert_allclose(got, ref, atol=1e-5)
    got_t = O.modulated_deform_conv(x, torch.zeros_like(off), torch.ones_like(m), w, b, 1, 1, 1, 1, dg)
    np.testing.assert_allclose(got_t.numpy(), ref, atol=1e-5)


def test_integer_offset_is_shifted_conv():
    x, off, m, w, b, dg = _case()
    dy, dx = 2, -1
    o = torch.zeros_like(off)
    o[:, 0::2] = dy
    o[:, 1::2] = dx
    # sampling x at (h+dy, w+dx) with zero fill == conv of the shifted, zero-filled image
    H, W = x.shape[2:]
    # a wide zero border, then the window shifted by (dy, dx): out-of-range taps read zeros
    xp = F.pad(x, (3, 3, 3, 3))
    ref = F.conv2d(xp[:, :, 3 + dy - 1:3 + dy - 1 + H + 2, 3 + dx - 1:3 + dx - 1 + W + 2], w, b)
    got = c_ref.dcn_forward(x.numpy(), o.numpy(), np.ones_like(m.numpy()), w.numpy(), b.numpy(), dg=dg)
    np.testing.assert_allclose(got, ref.numpy(), atol=1e-5)


def test_mask_linearity_and_bias():
    x, off, m, w, b, dg = _case(1)
    a = c_ref.dcn_forward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), None, dg=dg)
    a2 = c_ref.dcn_forward(x.numpy(), off.numpy(), (0.5 * m).numpy(), w.numpy(), None, dg=dg)
    np.testing.assert_allclose(a2, 0.5 * a, atol=1e-5)
    ab = c_ref.dcn_forward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), b.numpy(), dg=dg)
    np.testing.assert_allclose(ab, a + b.numpy().reshape(1, -1, 1, 1), atol=1e-5)


def test_border_partial_weights():
    """h_im in (-1, 0) keeps only the lower row with weight (1 - |h_im|); <= -1 gives exactly 0
    (deform_conv_cuda_kernel.cu:468-497, validity test :618)."""
    x = torch.ones(1, 1, 4, 4)
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 1, 1] = 1.0  # centre tap only
    m = torch.ones(1, 9, 4, 4)
    for shift, expect in ((-0.25, 0.75), (-1.0, 0.0), (-0.999, 0.001)):
        off = torch.zeros(1, 18, 4, 4)
        off[0, 2 * 4] = shift  # dh of the centre tap
        out = c_ref.dcn_forward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), None, dg=1)
        np.testing.assert_allclose(out[0, 0, 0], expect, atol=1e-6)  # first row samples h = shift
        np.testing.assert_allclose(out[0, 0, 2], 1.0, atol=1e-6)     # interior rows are unaffected
    off = torch.zeros(1, 18, 4, 4)
    off[0, 2 * 4] = 0.5  # last row samples h = 3.5: in (H-1, H) -> half weight
    out = c_ref.dcn_forward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), None, dg=1)
    np.testing.assert_allclose(out[0, 0, 3], 0.5, atol=1e-6)


def test_c_and_torch_formulations_agree_fwd_bwd():
    x, off, m, w, b, dg = _case(2, B=2, C=8, H=6, W=7, Co=5)
    xs = [t.clone().requires_grad_() for t in (x, off, m, w, b)]
    out = O.modulated_deform_conv(xs[0], xs[1], xs[2], xs[3], xs[4], 1, 1, 1, 1, dg)
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(9))
    out.backward(go)
    oc = c_ref.dcn_forward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), b.numpy(), dg=dg)
    np.testing.assert_allclose(oc, out.detach().numpy(), atol=2e-5)
    grads = c_ref.dcn_backward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), go.numpy(), dg=dg)
    for got, ref in zip(grads, xs):
        np.testing.assert_allclose(got, ref.grad.numpy(), atol=5e-5, rtol=1e-5)


def test_finite_differences_fp64():
    """All five gradients of the torch formulation against central differences in fp64 (offsets kept
    away from integer grid lines, where bilinear sampling is not differentiable)."""
    g = torch.Generator().manual_seed(4)
    B, C, H, W, Co, dg = 1, 4, 5, 5, 3, 2
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    off = (torch.rand(B, dg * 18, H, W, generator=g, dtype=torch.float64) * 0.6 + 0.2)
    m = torch.rand(B, dg * 9, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Co, C, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(Co, generator=g, dtype=torch.float64)
    ins = [t.requires_grad_() for t in (x, off, m, w, b)]
    assert torch.autograd.gradcheck(lambda *a: O.modulated_deform_conv(*a, 1, 1, 1, 1, dg), ins, eps=1e-6, atol=1e-5)
