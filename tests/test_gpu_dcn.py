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


@pytest.mark.parametrize("B,C,H,W,Co,off_scale", [(2, 128, 8, 10, 128, 0.3), (3, 128, 13, 21, 128, 4.0), (2, 256, 9, 11, 256, 1.5),
                                                   (1, 128, 40, 60, 256, 1.0), (2, 128, 210, 310, 128, 3.0), (4, 256, 210, 310, 256, 1.0)])
def test_fast_and_general_kernels_agree(B, C, H, W, Co, off_scale):
    """The bf16 pipeline entry has two kernels with the same arithmetic (glare_hip.h, flag GLARE_MDCN_GENERAL_KERNEL): the lean
    one must reproduce the general one on ragged tiles, image boundaries inside a tile and out-of-image samples."""

    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).cuda()
    plane = H * W + 16
    om = torch.zeros(B, 108, plane)
    om[:, :72, :H * W] = torch.randn(B, 72, H * W, generator=g) * off_scale
    om[:, 72:, :H * W] = torch.randn(B, 36, H * W, generator=g)
    om = om.cuda()
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.05
    pd = ops.PackedDcn(w.cuda(), torch.randn(Co, generator=g).cuda(), 4)
    fast = ops.mdcn_forward_nhwc(x, om, pd).cpu().numpy()
    for _ in range(3):   # run-to-run identical: the packed-fp32 version of this kernel was not (dcn.hip)
        assert np.array_equal(ops.mdcn_forward_nhwc(x, om, pd).cpu().numpy(), fast)
    general = ops.mdcn_forward_nhwc(x, om, pd, flags=ops.MDCN_GENERAL_KERNEL).cpu().numpy()   # a per-call flag: no library state
    # same products in the same order per (pixel, tap); only the fp32 accumulation grouping inside the MFMA chain is shared
    np.testing.assert_allclose(fast, general, rtol=0, atol=2e-5 * float(np.abs(general).max()))


@pytest.mark.parametrize("precision,tol", [("fp16", 1e-3), ("bf16", 8e-3)])
@pytest.mark.parametrize("B,C,H,W,Co,off_scale", [(2, 128, 13, 21, 128, 3.0), (1, 256, 9, 11, 256, 1.5), (2, 128, 105, 155, 128, 2.0),
                                                   (1, 256, 105, 155, 256, 1.0)])
def test_single_pass_form_against_the_split_form(precision, tol, B, C, H, W, Co, off_scale):
    """GLARE_MDCN_SINGLE_PASS: sample and filter rounded once to the precision's 16-bit format, one MFMA per product.  Against the
    split (fp32-class) form on the same 16-bit x: the difference is the rounding of the 9 * C products' operands (2^-11 each in
    half, 2^-8 in bf16), bounded here relative to max|ref|; run-to-run identical; and against the C oracle at the small size."""
    g = torch.Generator().manual_seed(B * 100 + C + H)
    with ops.use_precision(precision):
        dt = ops.act_dtype()
        x = torch.randn(B, H, W, C, generator=g).to(dt)
        plane = (H * W + 63) // 64 * 64
        off = torch.randn(B, 72, H * W, generator=g) * off_scale
        logit = torch.randn(B, 36, H * W, generator=g)
        om = torch.zeros(B, 108, plane)
        om[:, :72, :H * W] = off
        om[:, 72:, :H * W] = logit
        w = torch.randn(Co, C, 3, 3, generator=g) * (1.0 / (C * 9) ** 0.5)
        b = torch.randn(Co, generator=g)
        split = ops.mdcn_forward_nhwc(x.cuda(), om.cuda(), ops.PackedDcn(w.cuda(), b.cuda(), 4))
        pd1 = ops.PackedDcn(w.cuda(), b.cuda(), 4, single=True)
        assert pd1.packed.dtype == dt and pd1.packed.numel() == w.numel()
        one = ops.mdcn_forward_nhwc(x.cuda(), om.cuda(), pd1)
        for _ in range(2):
            assert torch.equal(ops.mdcn_forward_nhwc(x.cuda(), om.cuda(), pd1), one)
        ref = split.cpu().numpy()
        err = float(np.abs(one.cpu().numpy() - ref).max()) / float(np.abs(ref).max())
        print("\n[dcn single pass %s %dx%dx%d] max err / max|ref| = %.2e (bound %.0e)" % (precision, H, W, C, err, tol))
        assert err <= tol
        if H * W < 1000:
            cref = c_ref.dcn_forward(x.float().permute(0, 3, 1, 2).contiguous().numpy(), off.reshape(B, 72, H, W).numpy(),
                                     torch.sigmoid(logit).reshape(B, 36, H, W).numpy(), w.numpy(), b.numpy(), dg=4)
            got = one.permute(0, 3, 1, 2).cpu().numpy()
            assert float(np.abs(got - cref).max()) <= tol * float(np.abs(cref).max())
        # refusals: the flag needs the leaner kernel (16-bit x) and excludes the general-kernel flag
        with pytest.raises(RuntimeError):
            ops.mdcn_forward_nhwc(x.cuda(), om.cuda(), pd1, flags=ops.MDCN_GENERAL_KERNEL)


def test_nhwc_bf16_latent_size_against_c_oracle():
    """The pipeline entry at the latent resolution (105 x 155, 254 workgroups in flight) with scattered samples against the
    plain-C oracle: the small cases above fit in a handful of workgroups and cannot see a fault that depends on timing."""
    x, off, m, w, b = _case(21, 1, 128, 105, 155, 128, 4, off_scale=3.0)
    xb = x.to(torch.bfloat16)
    plane = (105 * 155 + 63) // 64 * 64
    om = torch.zeros(1, 108, plane)
    om[:, :72, :105 * 155] = off.reshape(1, 72, -1)
    om[:, 72:, :105 * 155] = m.reshape(1, 36, -1)
    pd = ops.PackedDcn(w.cuda(), b.cuda(), 4)
    got = ops.mdcn_forward_nhwc(xb.permute(0, 2, 3, 1).contiguous().cuda(), om.cuda(), pd, mask_is_logit=False)
    ref = c_ref.dcn_forward(xb.float().numpy(), off.numpy(), m.numpy(), w.numpy(), b.numpy(), dg=4)
    _close(got.permute(0, 3, 1, 2).cpu().numpy(), ref)


def test_errors():
    x, off, m, w, b = _case(6, 1, 24, 4, 4, 64, 4)  # 6 channels per deformable group: outside the MFMA kernels -> general kernels
    got = ops.mdcn_forward(x.cuda(), off.cuda(), m.cuda(), w.cuda(), b.cuda(), 1, 1, 1, 1, 4).cpu().numpy()
    _close(got, c_ref.dcn_forward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), b.numpy(), dg=4))
    with pytest.raises(Exception, match="invalid"):     # channels not divisible by the deformable groups (deform_conv_cuda.cpp:511-516)
        ops.mdcn_forward(x.cuda(), off.cuda(), m.cuda(), w.cuda(), b.cuda(), 1, 1, 1, 1, 5)
    with pytest.raises(NotImplementedError):
        ops.mdcn_forward(x, off, m, w, b, 1, 1, 1, 1, 4)  # CPU tensors: no fallback


# ---- backward (row a10) ---------------------------------------------------------------------------
def _bwd_case(B, C, H, W, Co, dg, seed, off_scale=1.5):
    x, off, m, w, b = _case(seed, B, C, H, W, Co, dg, off_scale)
    go = torch.randn(B, Co, H, W, generator=torch.Generator().manual_seed(seed + 1))
    return x, off, m, w, b, go


def _close_grad(got, ref, name):
    ref = np.asarray(ref)
    tol = 2e-4 * float(np.abs(ref).max()) + 1e-6
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=tol, err_msg=name)


@pytest.mark.parametrize("B,C,H,W,Co,dg", [(1, 128, 9, 13, 128, 4), (2, 256, 6, 11, 256, 4), (1, 128, 5, 70, 256, 4)])
def test_backward_matches_c_oracle(B, C, H, W, Co, dg):
    """All five gradients through the autograd Function (the reference's call path,
    deform_conv.py:155-174) against oracle/dcn_ref.c.  fp32 both; the oracle sums in double."""
    from glare_amd.modules.ops.dcn import modulated_deform_conv

    x, off, m, w, b, go = _bwd_case(B, C, H, W, Co, dg, seed=C + Co + H)
    ts = [t.clone().cuda().requires_grad_() for t in (x, off, m, w, b)]
    out = modulated_deform_conv(ts[0], ts[1], ts[2], ts[3], ts[4], 1, 1, 1, 1, dg)
    out.backward(go.cuda())
    ref = c_ref.dcn_backward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), go.numpy(), dg=dg)
    for t, r, name in zip(ts, ref, ("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias")):
        _close_grad(t.grad.cpu().numpy(), r, name)


def test_backward_borders_no_bias_and_accumulation():
    from glare_amd.modules.ops.dcn import deform_conv_ext

    x, off, m, w, b, go = _bwd_case(1, 128, 7, 9, 128, 4, seed=11, off_scale=5.0)  # many samples leave the image
    xc, oc, mc, wc, gc = [t.cuda().contiguous() for t in (x, off, m, w, go)]
    gi, goff, gm = torch.zeros_like(xc), torch.zeros_like(oc), torch.zeros_like(mc)
    gw = torch.full_like(wc, 0.5)  # grad_weight is ACCUMULATED into (deform_conv_cuda.cpp:655-660)
    e = xc.new_empty(0)
    deform_conv_ext.modulated_deform_conv_backward(xc, wc, xc.new_empty(1), e, oc, mc, e, gi, gw, xc.new_empty(1), goff, gm, gc,
                                                   3, 3, 1, 1, 1, 1, 1, 1, 1, 4, False)
    ref = c_ref.dcn_backward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), go.numpy(), with_bias=False, dg=4)
    _close_grad(gi.cpu().numpy(), ref[0], "grad_input")
    _close_grad(goff.cpu().numpy(), ref[1], "grad_offset")
    _close_grad(gm.cpu().numpy(), ref[2], "grad_mask")
    _close_grad(gw.cpu().numpy() - 0.5, ref[3], "grad_weight")


def test_backward_zero_offset_equals_conv_gradients():
    """offset = 0, mask = 1: grad_input / grad_weight / grad_bias are those of conv2d."""
    from glare_amd.modules.ops.dcn import modulated_deform_conv

    x, off, m, w, b, go = _bwd_case(1, 128, 8, 10, 128, 4, seed=12)
    ts = [t.clone().cuda().requires_grad_() for t in (x, torch.zeros_like(off), torch.ones_like(m), w, b)]
    modulated_deform_conv(ts[0], ts[1], ts[2], ts[3], ts[4], 1, 1, 1, 1, 4).backward(go.cuda())
    xr, wr, br = [t.clone().cuda().requires_grad_() for t in (x, w, b)]
    F.conv2d(xr, wr, br, 1, 1).backward(go.cuda())
    assert torch.allclose(ts[0].grad, xr.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(ts[3].grad, wr.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(ts[4].grad, br.grad, rtol=1e-4, atol=1e-4)


def test_dcn_v1_surface_matches_oracle_with_unit_mask():
    """Row f4: the three DCN v1 exports of deform_conv_ext (deform_conv_ext.cpp:52-104) through the reference-shaped
    DeformConvFunction.  v1 == v2 with mask == 1 and no bias, which is how the oracle evaluates it."""
    from glare_amd.modules.ops.dcn import DeformConvPack, deform_conv, deform_conv_ext

    x, off, m, w, b, go = _bwd_case(2, 128, 8, 10, 128, 4, seed=21)
    ones = np.ones_like(m.numpy())
    ts = [t.clone().cuda().requires_grad_() for t in (x, off, w)]
    out = deform_conv(ts[0], ts[1], ts[2], 1, 1, 1, 1, 4)
    ref = c_ref.dcn_forward(x.numpy(), off.numpy(), ones, w.numpy(), None, dg=4)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()))
    out.backward(go.cuda())
    gref = c_ref.dcn_backward(x.numpy(), off.numpy(), ones, w.numpy(), go.numpy(), with_bias=False, dg=4)
    _close_grad(ts[0].grad.cpu().numpy(), gref[0], "grad_input")
    _close_grad(ts[1].grad.cpu().numpy(), gref[1], "grad_offset")
    _close_grad(ts[2].grad.cpu().numpy(), gref[3], "grad_weight")
    # `scale` and accumulation of deform_conv_backward_parameters (deform_conv_cuda.cpp:478-482); W-before-H argument order
    gw = torch.full_like(ts[2].detach(), 0.25)
    e = x.new_empty(0).cuda()
    rc = deform_conv_ext.deform_conv_backward_parameters(x.cuda(), off.cuda(), go.cuda(), gw, e, e, 3, 3, 1, 1, 1, 1, 1, 1, 1, 4, 0.5, 2)
    assert rc == 1
    _close_grad((gw.cpu().numpy() - 0.25) / 0.5, gref[3], "grad_weight (scaled, accumulated)")
    with pytest.raises(RuntimeError):
        deform_conv_ext.deform_conv_forward(x, w, off, x.new_empty(1), e, e, 3, 3, 1, 1, 1, 1, 1, 1, 1, 4, 2)   # CPU input
    pack = DeformConvPack(128, 128, 3, padding=1, deformable_groups=4)
    assert sorted(k for k, _ in pack.named_parameters()) == ["conv_offset.bias", "conv_offset.weight", "weight"]


# ---- shapes outside the MFMA kernels' configurations: the general fp32 kernels (csrc/dcn_generic.hip) -----------------------
GENERIC_CASES = [  # B, C, H, W, Co, k, stride, pad, dil, groups, dg
    (2, 12, 7, 9, 8, 3, 1, 1, 1, 2, 3),      # conv groups AND deformable groups, channel counts off every tile size
    (1, 8, 10, 6, 6, 1, 1, 0, 1, 1, 2),      # 1x1 kernel
    (1, 16, 9, 11, 24, 3, 2, 2, 2, 4, 1),    # stride 2, dilation 2, four conv groups
    (1, 6, 6, 7, 70, 5, 1, 2, 1, 1, 6),      # 5x5 kernel, more output channels than a workgroup's 64, one channel per deformable group
    (2, 64, 8, 8, 64, 3, 1, 1, 1, 2, 2),     # MFMA-sized channel counts but groups = 2
    (1, 128, 6, 5, 96, 3, 1, 1, 1, 1, 4),    # Co not a multiple of 64 with the GLARE input geometry
]


@pytest.mark.parametrize("B,C,H,W,Co,k,stride,pad,dil,groups,dg", GENERIC_CASES)
def test_general_shapes_forward_and_backward_match_c_oracle(B, C, H, W, Co, k, stride, pad, dil, groups, dg):
    """deform_conv_ext.modulated_deform_conv_forward / _backward accept any group / deformable_group / channel / kernel /
    stride / padding / dilation configuration (deform_conv_cuda.cpp:497-516); so does the C ABI (no GLARE_ERR_UNSUPPORTED),
    through the reference-shaped autograd Function, against oracle/dcn_ref.c: output and all five gradients."""
    from glare_amd.modules.ops.dcn import modulated_deform_conv

    g = torch.Generator().manual_seed(C * 7 + Co + k)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=g) * 1.5
    m = torch.rand(B, dg * k * k, Ho, Wo, generator=g)
    w = torch.randn(Co, C // groups, k, k, generator=g) * (1.0 / (C // groups * k * k) ** 0.5)
    b = torch.randn(Co, generator=g)
    go = torch.randn(B, Co, Ho, Wo, generator=g)
    ts = [t.clone().cuda().requires_grad_() for t in (x, off, m, w, b)]
    out = modulated_deform_conv(ts[0], ts[1], ts[2], ts[3], ts[4], stride, pad, dil, groups, dg)
    ref = c_ref.dcn_forward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), b.numpy(), stride=stride, padding=pad, dilation=dil,
                            groups=groups, dg=dg)
    _close(out.detach().cpu().numpy(), ref)
    out.backward(go.cuda())
    gref = c_ref.dcn_backward(x.numpy(), off.numpy(), m.numpy(), w.numpy(), go.numpy(), stride=stride, padding=pad, dilation=dil,
                              groups=groups, dg=dg)
    for t, r, name in zip(ts, gref, ("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias")):
        _close_grad(t.grad.cpu().numpy(), r, name)


def test_general_shapes_c_abi_needs_no_workspace_and_rejects_bad_channel_counts():
    import ctypes

    from glare_amd import _lib

    lib = _lib.lib()
    B, C, H, W, Co, dg = 1, 12, 5, 6, 10, 3
    g = torch.Generator().manual_seed(1)
    x, off, m = torch.randn(B, C, H, W, generator=g).cuda(), torch.randn(B, dg * 18, H, W, generator=g).cuda(), torch.rand(B, dg * 9, H, W, generator=g).cuda()
    w = torch.randn(Co, C, 3, 3, generator=g).cuda()
    out = torch.empty(B, Co, H, W, device="cuda")
    i = ctypes.c_int
    args = lambda c, groups, dgs: (_lib.ptr(x), _lib.ptr(off), _lib.ptr(m), _lib.ptr(w), ctypes.c_void_p(0), _lib.ptr(out), i(B), i(c), i(H),
                                   i(W), i(Co), i(3), i(3), i(1), i(1), i(1), i(1), i(1), i(1), i(groups), i(dgs), ctypes.c_void_p(0),
                                   ctypes.c_size_t(0), _lib.stream_handle())
    assert lib.glare_mdcn_forward_f32(*args(C, 1, dg)) == 0                 # NULL workspace: the general kernels need none
    ref = c_ref.dcn_forward(x.cpu().numpy(), off.cpu().numpy(), m.cpu().numpy(), w.cpu().numpy(), None, dg=dg)
    _close(out.cpu().numpy(), ref)
    assert lib.glare_mdcn_forward_f32(*args(C, 1, 5)) != 0                  # C % deformable_group != 0 (deform_conv_cuda.cpp:511-516)
    assert lib.glare_mdcn_forward_f32(*args(C, 5, dg)) != 0                 # C % group != 0


def test_pack_modules_run_conv_offset_on_the_hip_kernels(monkeypatch):
    """ModulatedDeformConvPack / DeformConvPack .forward: `conv_offset` no longer goes through nn.Conv2d (MIOpen) -- the MFMA conv
    for the 3x3 'same' configuration, the general DCN kernel with zero offsets / unit mask for any other (stride 2 here)."""
    import torch.nn.functional as F

    from glare_amd.modules.ops.dcn import ModulatedDeformConvPack
    import importlib

    dc = importlib.import_module("glare_amd.modules.ops.dcn.deform_conv")   # the module (the package re-exports a function of that name)

    def boom(*a, **k):
        raise AssertionError("conv_offset ran through nn.Conv2d")

    monkeypatch.setattr(torch.nn.Conv2d, "forward", boom)
    g = torch.Generator().manual_seed(2)
    for stride in (1, 2):
        m = ModulatedDeformConvPack(16, 24, 3, stride=stride, padding=1, deformable_groups=2).cuda()
        with torch.no_grad():
            m.conv_offset.weight.copy_(torch.randn(m.conv_offset.weight.shape, generator=g) * 0.05)
            m.conv_offset.bias.copy_(torch.randn(m.conv_offset.bias.shape, generator=g) * 0.5)
        x = torch.randn(2, 16, 12, 14, generator=g).cuda()
        with torch.no_grad():
            got_off = dc._offset_conv(m.conv_offset, x)
            ref_off = F.conv2d(x, m.conv_offset.weight, m.conv_offset.bias, stride, 1)
            tol = 2e-2 if stride == 1 else 1e-4          # stride 1: the bf16 MFMA conv; stride 2: the fp32 general kernel
            assert float((got_off - ref_off).norm() / ref_off.norm()) < tol
            y = m(x)
        assert y.shape == (2, 24, 12 // stride, 14 // stride) and torch.isfinite(y).all()
        y2 = m(x.requires_grad_(True))                     # with a tape: the DCN form, gradients reach conv_offset
        y2.sum().backward()
        assert m.conv_offset.weight.grad is not None and float(m.conv_offset.weight.grad.abs().sum()) > 0


# ---- round 6: the pipeline's form of the call -- 16-bit output and per-tile sums from the epilogue, the mean rescale fed by them ------
def _nhwc_case(seed, B, C, H, W, precision):
    g = torch.Generator().manual_seed(seed)
    with ops.use_precision(precision):
        x = (torch.randn(B, H, W, C, generator=g) * 0.7).to(ops.act_dtype()).cuda()
        plane = (H * W + 63) // 64 * 64
        om = torch.zeros(B, 108, plane)
        om[:, :72, :H * W] = torch.randn(B, 72, H * W, generator=g) * 2.0
        om[:, 72:, :H * W] = torch.randn(B, 36, H * W, generator=g)
        w = torch.randn(C, C, 3, 3, generator=g) * (1.0 / (C * 9) ** 0.5)
        b = torch.randn(C, generator=g) * 0.3 + 0.2
    return x, om.cuda(), w.cuda(), b.cuda()


@pytest.mark.parametrize("B,C,H,W", [(3, 128, 13, 21), (2, 256, 12, 17), (1, 128, 16, 8)])
@pytest.mark.parametrize("precision,single", [("fp16", False), ("fp16", True), ("bf16", False)])
def test_fused_epilogue_is_the_plain_output_rounded_once_and_summed(B, C, H, W, precision, single):
    """glare_mdcn_forward_nhwc_fused against glare_mdcn_forward_nhwc on the same launch shape: the 16-bit output is the fp32 output
    rounded once (bit for bit -- same accumulators), the tile sums add up to the fp32 output's per-image sums (ragged last tiles:
    13 * 21 = 273 pixels per image against 64- / 128-pixel tiles), and an image's sums do not depend on the batch it sits in."""
    x, om, w, b = _nhwc_case(B * 100 + C + H, B, C, H, W, precision)
    with ops.use_precision(precision):
        pd = ops.PackedDcn(w, b, 4, single=single)
        ref = ops.mdcn_forward_nhwc(x, om, pd)
        out16, sums, tile = ops.mdcn_forward_nhwc_fused(x, om, pd, out16=True, want_sums=True)
        out32, sums2, _ = ops.mdcn_forward_nhwc_fused(x, om, pd, out16=False, want_sums=True)
        assert out16.dtype == ops.act_dtype() and torch.equal(out16, ref.to(ops.act_dtype()))
        assert torch.equal(out32, ref) and torch.equal(sums, sums2)
        assert tuple(sums.shape) == (B, (H * W + tile - 1) // tile)               # tiles are cut per image (the last one ragged)
        per_img = sums.double().sum(dim=1).cpu()
        want = ref.double().sum(dim=(1, 2, 3)).cpu()
        assert float(((per_img - want).abs() / want.abs()).max()) < 2e-6
        one16, one_sums, _ = ops.mdcn_forward_nhwc_fused(x[B - 1:].contiguous(), om[B - 1:].contiguous(), pd)       # the last image alone
        assert torch.equal(one16[0], out16[B - 1]) and torch.equal(one_sums[0], sums[B - 1])
        # the rescale fed by the producers' sums == the rescale with its own statistics pass (fp32 x_w), and its 16-bit-x_w form
        g = torch.Generator().manual_seed(7)
        e = (torch.randn(B, H, W, C, generator=g) * 0.5 + 0.3).to(ops.act_dtype()).cuda()
        h0 = (torch.randn(B, H, W, C, generator=g) * 0.5 + 0.3).to(ops.act_dtype()).cuda()
        h, hs = ops.mix_with_sums(e, h0, -0.6)
        assert torch.equal(h, ops.mix(e, h0, -0.6))
        for whole in (False, True):
            old = ops.mean_rescale(h, ref, whole_batch=whole)
            new32 = ops.mean_rescale_fused(h, out32, hs, sums, tile, whole_batch=whole)
            new16 = ops.mean_rescale_fused(h, out16, hs, sums, tile, whole_batch=whole)
            ulp = 2.0 ** (-10 if precision == "fp16" else -7)
            d32 = (new32.float() - old.float()).abs() / old.float().abs().clamp_min(1e-3)
            assert float(d32.max()) <= 1.01 * ulp and float((d32 > 0).float().mean()) < 1e-3      # the ratio's summation order: rare 1-ulp flips
            ratio = (h.double().sum(dim=(1, 2, 3)) if not whole else h.double().sum().expand(B)) / (ref.double().sum(dim=(1, 2, 3)) if not whole else ref.double().sum().expand(B))
            want16 = (h.double() + out16.double() * ratio.view(B, 1, 1, 1)).float()
            d16 = (new16.float() - want16).abs() / want16.abs().clamp_min(1e-3)
            assert float(d16.max()) <= 1.01 * ulp


def test_fused_call_on_images_smaller_than_a_tile():
    x, om, w, b = _nhwc_case(1, 3, 128, 5, 7, "fp16")        # 35 pixels per image: one ragged 64-pixel tile each
    with ops.use_precision("fp16"):
        pd = ops.PackedDcn(w, b, 4)
        ref = ops.mdcn_forward_nhwc(x, om, pd)
        out16, sums, tile = ops.mdcn_forward_nhwc_fused(x, om, pd)
        assert torch.equal(out16, ref.to(torch.float16)) and tuple(sums.shape) == (3, 1)
        assert float(((sums[:, 0].double().cpu() - ref.double().sum(dim=(1, 2, 3)).cpu()).abs() / ref.double().sum(dim=(1, 2, 3)).cpu().abs()).max()) < 2e-6
