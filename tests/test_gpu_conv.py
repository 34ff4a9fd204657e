"""GPU: implicit-GEMM convolution (csrc/conv_igemm.hip) through the C ABI against F.conv2d in fp32
on the same bf16-rounded operands.  Tolerance: fp32 accumulation of bf16 products -> the only
difference to the fp32 reference is summation order and the final bf16 rounding of the output:
|err| <= 2^-8 * |ref| + 1e-3 * max|ref|."""
import pytest
import torch
import torch.nn.functional as F

from glare_amd import ops

pytestmark = pytest.mark.gpu


def _rand(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale)


def _nhwc_bf16(x):  # NCHW fp32 -> NHWC bf16 on GPU
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()


def _check(got_nchw, ref, bf16_out=True):
    ref = ref.float().cpu()
    got = got_nchw.float().cpu()
    tol = (2.0 ** -8 if bf16_out else 1e-5) * ref.abs() + 2e-3 * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), "max err %g of max %g at %d elems" % (
        float((got - ref).abs().max()), float(ref.abs().max()), int(bad.sum()))


CASES = [
    # B, Cin, Cout, H, W, k
    (1, 128, 128, 16, 40, 3),
    (2, 64, 256, 9, 33, 3),     # ragged tile edges
    (1, 256, 128, 8, 32, 1),
    (2, 512, 512, 7, 45, 1),
    (1, 64, 64, 11, 35, 3),     # TN=64 variant
    (1, 64, 6, 10, 37, 3),      # TN=32 variant, Cout < 32
    (1, 128, 3, 20, 31, 3),
    (1, 64, 108, 12, 20, 3),    # conv_offset-like (108 channels)
    (1, 24, 40, 6, 10, 3),      # Cin not a multiple of 16 (zero-filled tail)
]


@pytest.mark.parametrize("B,Cin,Cout,H,W,k", CASES)
def test_conv_matches_fp32_reference(B, Cin, Cout, H, W, k):
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H)
    x = _rand((B, Cin, H, W), g)
    w = _rand((Cout, Cin, k, k), g, 1.0 / (Cin * k * k) ** 0.5)
    b = _rand((Cout,), g, 0.1)
    xb = x.to(torch.bfloat16).float()
    wb = w.to(torch.bfloat16).float()
    ref = F.conv2d(xb.cuda(), wb.cuda(), b.cuda(), 1, k // 2)
    pc = ops.PackedConv(w.cuda(), b.cuda())
    out = ops.conv2d(_nhwc_bf16(x), pc)
    _check(out.permute(0, 3, 1, 2), ref)


def test_residual_act_and_f32_output():
    g = torch.Generator().manual_seed(5)
    x = _rand((2, 128, 10, 34), g)
    w = _rand((128, 128, 3, 3), g, 0.03)
    b = _rand((128,), g, 0.1)
    r = _rand((2, 128, 10, 34), g)
    ref = F.conv2d(x.to(torch.bfloat16).float().cuda(), w.to(torch.bfloat16).float().cuda(), b.cuda(), 1, 1)
    ref = F.relu(ref + r.to(torch.bfloat16).float().cuda())
    pc = ops.PackedConv(w.cuda(), b.cuda())
    out = ops.conv2d(_nhwc_bf16(x), pc, residual=_nhwc_bf16(r), act="relu", out_mode=ops.OUT_NHWC_F32)
    _check(out.permute(0, 3, 1, 2), ref, bf16_out=False)


def test_upsample_and_downsample_fusions():
    g = torch.Generator().manual_seed(6)
    x = _rand((1, 64, 9, 21), g)
    w = _rand((64, 64, 3, 3), g, 0.04)
    b = _rand((64,), g, 0.1)
    xb, wb = x.to(torch.bfloat16).float().cuda(), w.to(torch.bfloat16).float().cuda()
    pc = ops.PackedConv(w.cuda(), b.cuda())
    # Upsample: nearest x2 then 3x3 (encoder_decoder.py:49-53)
    ref = F.conv2d(F.interpolate(xb, scale_factor=2.0, mode="nearest"), wb, b.cuda(), 1, 1)
    out = ops.conv2d(_nhwc_bf16(x), pc, upsample=True)
    _check(out.permute(0, 3, 1, 2), ref)
    # Downsample: pad (0,1,0,1) then 3x3 stride 2 (encoder_decoder.py:69-73), odd and even sizes
    for hw in ((9, 21), (10, 36)):
        x2 = _rand((1, 64) + hw, g)
        ref = F.conv2d(F.pad(x2.to(torch.bfloat16).float().cuda(), (0, 1, 0, 1)), wb, b.cuda(), 2, 0)
        out = ops.conv2d(_nhwc_bf16(x2), pc, stride=2)
        assert out.shape[1:3] == ref.shape[2:]
        _check(out.permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("B,C,Co,H,W", [(1, 64, 64, 9, 21), (2, 128, 128, 13, 45), (1, 256, 256, 8, 32), (1, 512, 512, 5, 33),
                                        (1, 64, 40, 1, 1), (1, 32, 128, 17, 70)])
def test_upsample_subpixel_form(B, C, Co, H, W):
    """desc.upsample = 2: nearest x2 + 3x3 as four 2x2 convs of the source with pre-summed taps.  Checked against the
    operator itself (fp32 weights; the regular path rounds each tap to bf16, this one rounds each tap SUM), against the
    regular fused path, and with a residual + fused GroupNorm statistics on ragged tiles."""
    g = torch.Generator().manual_seed(B * 100 + H)
    x = _rand((B, C, H, W), g)
    w = _rand((Co, C, 3, 3), g, 0.03)
    b = _rand((Co,), g, 0.1)
    xb = x.to(torch.bfloat16).float().cuda()
    ref = F.conv2d(F.interpolate(xb, scale_factor=2.0, mode="nearest"), w.cuda(), b.cuda(), 1, 1)
    pcs = ops.PackedConv(w.cuda(), b.cuda(), upsample_subpixel=True)
    sub = ops.conv2d(_nhwc_bf16(x), pcs, upsample=True)
    assert tuple(sub.shape) == (B, 2 * H, 2 * W, Co)
    _check(sub.permute(0, 3, 1, 2), ref)
    reg = ops.conv2d(_nhwc_bf16(x), ops.PackedConv(w.cuda(), b.cuda()), upsample=True)
    d = (sub.float() - reg.float()).norm() / reg.float().norm()
    assert float(d) < 6e-3, float(d)
    # residual + activation through the same scatter (the LDS-staged epilogue rounds the conv result to bf16 before the
    # residual add, so the element-wise bound of _check does not apply where the two cancel: compare in norm)
    r = _rand((B, Co, 2 * H, 2 * W), g)
    got = ops.conv2d(_nhwc_bf16(x), pcs, upsample=True, residual=_nhwc_bf16(r), act="relu").permute(0, 3, 1, 2).float()
    want = F.relu(ref + r.to(torch.bfloat16).float().cuda())
    assert float((got - want).norm() / want.norm()) < 4e-3
    assert float((got - want).abs().max()) < 2.0 ** -7 * float(ref.abs().max() + r.abs().max())
    if Co % 128 == 0:   # fused GroupNorm statistics: identical output, statistics equal to a fresh pass over it
        y1 = ops.conv2d(_nhwc_bf16(x), pcs, upsample=True, gn_stats=True)
        assert torch.equal(y1, sub)
        gamma, beta = _rand((Co,), g, 0.5).cuda() + 1.0, _rand((Co,), g, 0.2).cuda()
        n1 = ops.groupnorm(y1, gamma, beta, swish=True)
        n2 = ops.groupnorm(sub, gamma, beta, swish=True)
        assert torch.allclose(n1.float(), n2.float(), rtol=2 ** -7, atol=2e-3)


def test_upsample_subpixel_full_size():
    """The two decoder shapes at BASELINE size (256 ch 210x310 -> 420x620, 512 ch 105x155 -> 210x310): the sub-pixel form
    agrees with the loader-fused form (each rounds different bf16 filters: tap sums vs taps) and commutes with a flip of the
    image (a size-independent property: flipping rows swaps the roles of the two row phases)."""
    g = torch.Generator().manual_seed(12)
    for C, H, W in ((256, 210, 310), (512, 105, 155)):
        x = (torch.randn(1, H, W, C, generator=g)).to(torch.bfloat16).cuda()
        w = (torch.randn(C, C, 3, 3, generator=g) * 0.02).cuda()
        b = (torch.randn(C, generator=g) * 0.1).cuda()
        sub = ops.conv2d(x, ops.PackedConv(w, b, upsample_subpixel=True), upsample=True).float()
        reg = ops.conv2d(x, ops.PackedConv(w, b), upsample=True).float()
        assert float((sub - reg).norm() / reg.norm()) < 6e-3
        subf = ops.conv2d(x.flip(1).contiguous(), ops.PackedConv(w.flip(2).contiguous(), b, upsample_subpixel=True), upsample=True).float()
        assert float((subf.flip(1) - sub).norm() / sub.norm()) < 1e-6 + 2.0 ** -9   # same sums, other phase's code path


def test_concat_pitch_offsets_and_planar_output():
    g = torch.Generator().manual_seed(7)
    a = _rand((1, 32, 12, 33), g)
    c = _rand((1, 32, 12, 33), g)
    w = _rand((64, 64, 3, 3), g, 0.04)
    ref = F.conv2d(torch.cat([a, c], 1).to(torch.bfloat16).float().cuda(), w.to(torch.bfloat16).float().cuda(), None, 1, 1)
    pc = ops.PackedConv(w.cuda(), None)
    # source 1 lives at channel offset 16 of a 64-pitch buffer; source 2 is a separate tensor
    buf = torch.zeros(1, 12, 33, 64, dtype=torch.bfloat16, device="cuda")
    buf[..., 16:48] = _nhwc_bf16(a)
    out = torch.zeros(1, 12, 33, 96, dtype=torch.bfloat16, device="cuda")
    ops.conv2d(buf, pc, cin=32, in_off=16, x2=_nhwc_bf16(c), out=out, out_off=32)
    _check(out[..., 32:96].permute(0, 3, 1, 2), ref)
    assert float(out[..., :32].abs().max()) == 0.0
    # planar (NCHW) fp32 output with a padded plane pitch
    pl = ops.conv2d(buf, pc, cin=32, in_off=16, x2=_nhwc_bf16(c), out_mode=ops.OUT_PLANAR_F32, plane_pitch=12 * 33 + 20)
    _check(pl[:, :, :12 * 33].reshape(1, 64, 12, 33), ref, bf16_out=False)
    assert float(pl[:, :, 12 * 33:].abs().max()) == 0.0


def test_full_size_linearity():
    """BASELINE size (B=8, 128 ch, 420x620): conv(x1 + x2) == conv(x1) + conv(x2) up to bf16
    rounding, and a spot check of 64 random output pixels against a direct fp32 evaluation."""
    g = torch.Generator().manual_seed(8)
    B, C, H, W = 2, 128, 420, 620
    x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).cuda()
    w = _rand((C, C, 3, 3), g, 0.03)
    pc = ops.PackedConv(w.cuda(), None)
    y = ops.conv2d(x, pc, out_mode=ops.OUT_NHWC_F32)
    y2 = ops.conv2d((x.float() * 2).to(torch.bfloat16), pc, out_mode=ops.OUT_NHWC_F32)
    assert torch.allclose(y2, 2 * y, rtol=1e-5, atol=1e-5)  # scaling by 2 is exact in bf16
    wb = w.to(torch.bfloat16).float().cuda()
    xp = F.pad(x.float().permute(0, 3, 1, 2), (1, 1, 1, 1))
    for _ in range(64):
        b = int(torch.randint(0, B, (1,), generator=g)); yy = int(torch.randint(0, H, (1,), generator=g))
        xx = int(torch.randint(0, W, (1,), generator=g))
        ref = (xp[b, :, yy:yy + 3, xx:xx + 3][None] * wb).sum(dim=(1, 2, 3))
        assert torch.allclose(y[b, yy, xx], ref, rtol=1e-3, atol=2e-3)


def test_fused_groupnorm_statistics():
    """The conv epilogue's fused GroupNorm statistics equal the standalone statistics pass: the normalised
    output is identical up to fp32 summation order (ragged edges, residual, 128/256/512 channels)."""
    g = torch.Generator().manual_seed(21)
    for B, Cin, Cout, H, W in ((2, 64, 128, 13, 37), (1, 128, 256, 9, 33), (1, 64, 512, 8, 40)):
        x = _rand((B, Cin, H, W), g)
        w = _rand((Cout, Cin, 3, 3), g, 0.05)
        b = _rand((Cout,), g, 0.1)
        r = _rand((B, Cout, H, W), g)
        gamma = (_rand((Cout,), g, 0.2) + 1).cuda()
        beta = _rand((Cout,), g, 0.2).cuda()
        pc = ops.PackedConv(w.cuda(), b.cuda())
        y1 = ops.conv2d(_nhwc_bf16(x), pc, residual=_nhwc_bf16(r), gn_stats=True)
        assert hasattr(y1, "_gn_stats")
        y2 = ops.conv2d(_nhwc_bf16(x), pc, residual=_nhwc_bf16(r))
        assert torch.equal(y1, y2)
        n1 = ops.groupnorm(y1, gamma, beta, swish=True)   # apply only, fused statistics
        n2 = ops.groupnorm(y2, gamma, beta, swish=True)   # stats + apply
        assert torch.allclose(n1.float(), n2.float(), rtol=2 ** -7, atol=2e-3)


def test_conv_randomised_shapes_and_fusions():
    """Seeded sweep over ragged sizes and every fused loader / epilogue feature (stride-2 (0,1,0,1) padding, nearest x2 upsample,
    two concatenated sources read at channel offsets of wider buffers, residual, activations, fp32 / planar outputs)."""
    import random

    rnd = random.Random(2024)
    g = torch.Generator().manual_seed(2024)
    acts = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "swish": lambda t: t * torch.sigmoid(t)}
    for case in range(36):
        B, H, W = rnd.choice([1, 2, 3]), rnd.randint(1, 41), rnd.randint(1, 70)
        k = rnd.choice([1, 3, 3])
        cin = rnd.choice([8, 16, 24, 64, 72, 128])
        cout = rnd.choice([3, 8, 32, 64, 100, 128, 136, 256])
        stride = rnd.choice([1, 1, 2]) if (k == 3 and H >= 2 and W >= 2) else 1
        ups = stride == 1 and rnd.random() < 0.25
        two = rnd.random() < 0.3 and cin % (16 if k == 3 else 32) == 0   # a stage must not straddle the two sources
        act = rnd.choice(["none", "none", "relu", "sigmoid", "swish"])
        mode = rnd.choice([ops.OUT_NHWC_BF16, ops.OUT_NHWC_BF16, ops.OUT_NHWC_F32])
        res = mode == ops.OUT_NHWC_BF16 and act == "none" and rnd.random() < 0.4 and cout % 8 == 0
        # sources live inside wider buffers at channel offsets
        pad_l, pad_r = rnd.choice([0, 8, 16]), rnd.choice([0, 8])
        x = _rand((B, cin, H, W), g)
        xbuf = torch.zeros(B, H, W, pad_l + cin + pad_r)
        xbuf[..., pad_l:pad_l + cin] = x.permute(0, 2, 3, 1)
        xd = xbuf.to(torch.bfloat16).cuda()
        cin2 = rnd.choice([16, 32]) if two else 0
        x2 = _rand((B, cin2, H, W), g) if two else None
        w = _rand((cout, cin + cin2, k, k), g, 1.0 / ((cin + cin2) * k * k) ** 0.5)
        b = _rand((cout,), g, 0.1)
        xin = torch.cat([x, x2], 1) if two else x
        xin = xin.to(torch.bfloat16).float()
        if ups:
            xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
        wb = w.to(torch.bfloat16).float()
        ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), wb, b, stride=2) if stride == 2 else F.conv2d(xin, wb, b, padding=k // 2)
        r = _rand(ref.shape, g) if res else None
        ref = acts[act](ref)
        if res:
            ref = ref + r.to(torch.bfloat16).float()
        try:
            out = ops.conv2d(xd, ops.PackedConv(w.cuda(), b.cuda()), cin=cin, in_off=pad_l, x2=_nhwc_bf16(x2) if two else None,
                             stride=stride, upsample=ups, act=act, residual=_nhwc_bf16(r) if res else None, out_mode=mode)
            _check(out.permute(0, 3, 1, 2), ref, bf16_out=(mode == ops.OUT_NHWC_BF16))
        except Exception as e:
            raise AssertionError("case %d: B%d %dx%d k%d s%d ups%d cin%d+%d@%d cout%d act=%s res=%d mode=%d: %s"
                                 % (case, B, H, W, k, stride, ups, cin, cin2, pad_l, cout, act, res, mode, e))


def test_error_behaviour_of_the_new_entry_points():
    """Status codes instead of crashes: CPU tensors are refused before any launch, a sub-pixel filter cannot serve a plain
    conv, the sub-pixel form exists for stride 1 only, and the fused add refuses shapes its statistics pass cannot tile."""
    import ctypes

    from glare_amd import _lib

    w = torch.randn(64, 64, 3, 3) * 0.02
    x = torch.randn(1, 6, 7, 64).to(torch.bfloat16)
    with pytest.raises(NotImplementedError):
        ops.conv2d(x, ops.PackedConv(w.cuda(), None, upsample_subpixel=True), upsample=True)      # CPU activation
    pcs = ops.PackedConv(w.cuda(), None, upsample_subpixel=True)
    with pytest.raises(AssertionError):
        ops.conv2d(x.cuda(), pcs)                                                                   # sub-pixel filter, no upsample
    with pytest.raises(_lib.GlareError, match="unsupported"):
        ops.conv2d(x.cuda(), pcs, upsample=True, stride=2)
    with pytest.raises(_lib.GlareError, match="unsupported"):
        ops.conv2d(x.cuda(), pcs, upsample=True, out_mode=ops.OUT_NHWC_F32)                         # scatter lives in the bf16 epilogue
    a = torch.randn(1, 4, 4, 24).to(torch.bfloat16).cuda()                                          # 24 channels: not 32 groups
    with pytest.raises(_lib.GlareError, match="unsupported"):
        ops.add_bf16(a, a, gn_stats=True)
    lib = _lib.lib()
    lib.glare_conv2d_upsample_packed_weight_elems.restype = ctypes.c_longlong
    assert lib.glare_conv2d_upsample_packed_weight_elems(ctypes.c_int(0), ctypes.c_int(64)) < 0
    assert lib.glare_add_groupnorm_stats_bf16(None, None, None, ctypes.c_int(1), ctypes.c_longlong(16), ctypes.c_int(64), None,
                                              ctypes.c_size_t(0), None) != 0


# ---- 1x1 convolution, weight-stationary kernel (csrc/conv1x1.hip) ----------------------------------------------------------
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 512, 512, 7, 45), (1, 256, 128, 8, 32), (3, 128, 256, 5, 13), (1, 512, 1024, 9, 11),
                                             (2, 512, 512, 105, 155), (1, 256, 512, 1, 1), (9, 128, 128, 3, 11)])
def test_conv1x1_weight_stationary_matches_fp32_reference_and_the_igemm_kernel(B, Cin, Cout, H, W):
    """The persistent weight-stationary 1x1 kernel against F.conv2d (fp32, bf16-rounded operands) and against the implicit-GEMM
    kernel it replaces (identical products, fp32 accumulation in a different order -> equal up to the output's bf16 rounding);
    ragged row blocks (H*W not a multiple of 32), more images than pixel ranges, a single pixel."""
    g = torch.Generator().manual_seed(Cin + Cout + H * W)
    x = _rand((B, Cin, H, W), g)
    w = _rand((Cout, Cin, 1, 1), g, 1.0 / Cin ** 0.5)
    b = _rand((Cout,), g, 0.1)
    ref = F.conv2d(x.to(torch.bfloat16).float().cuda(), w.to(torch.bfloat16).float().cuda(), b.cuda())
    pc = ops.PackedConv(w.cuda(), b.cuda())
    assert pc.w16 is not None
    xn = _nhwc_bf16(x)
    out = ops.conv2d(xn, pc)
    _check(out.permute(0, 3, 1, 2), ref)
    ops.CONV1X1_WEIGHT_STATIONARY = False
    try:
        old = ops.conv2d(xn, pc)
    finally:
        ops.CONV1X1_WEIGHT_STATIONARY = True
    assert torch.allclose(out.float(), old.float(), rtol=2 ** -7, atol=2e-3 * float(ref.abs().max()))


def test_conv1x1_weight_stationary_epilogue_residual_act_offsets_and_gn_statistics():
    """Residual add + activation, channel sub-ranges of wider records on both sides, and the fused GroupNorm statistics: the norm
    that consumes them must equal the norm that computes its own."""
    g = torch.Generator().manual_seed(77)
    B, Cin, Cout, H, W = 2, 256, 256, 9, 21
    rec = _rand((B, H, W, Cin + 64), g).to(torch.bfloat16).cuda()          # the input lives at channels 64..
    w = _rand((Cout, Cin, 1, 1), g, 1.0 / Cin ** 0.5)
    b = _rand((Cout,), g, 0.1)
    res = _rand((B, H, W, Cout + 8), g).to(torch.bfloat16).cuda()          # residual at channels 8..
    pc = ops.PackedConv(w.cuda(), b.cuda())
    out = torch.zeros(B, H, W, Cout + 16, dtype=torch.bfloat16, device="cuda")
    ops.conv2d(rec, pc, cin=Cin, in_off=64, residual=res, res_off=8, act="relu", out=out, out_off=16)
    x = rec[..., 64:].float().permute(0, 3, 1, 2)
    ref = torch.relu(F.conv2d(x, w.to(torch.bfloat16).float().cuda(), b.cuda()).to(torch.bfloat16).float()
                     + res[..., 8:].float().permute(0, 3, 1, 2))
    _check(out[..., 16:].permute(0, 3, 1, 2), ref)
    assert float(out[..., :16].abs().max()) == 0.0                          # nothing outside the channel window
    # fused statistics
    xn = rec[..., 64:].contiguous()
    y = ops.conv2d(xn, pc, gn_stats=True)
    assert getattr(y, "_gn_stats", None) is not None
    gamma, beta = (_rand((Cout,), g, 0.3) + 1.0).cuda(), _rand((Cout,), g, 0.2).cuda()
    n1 = ops.groupnorm(y, gamma, beta, swish=True)
    n2 = ops.groupnorm(y.clone(), gamma, beta, swish=True)
    assert torch.allclose(n1.float(), n2.float(), rtol=2 ** -7, atol=2e-3)


def test_conv1x1_weight_stationary_is_deterministic_and_falls_back_outside_its_shapes():
    g = torch.Generator().manual_seed(3)
    x = _rand((8, 512, 105, 155), g).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    pc = ops.PackedConv((_rand((512, 512, 1, 1), g, 0.05)).cuda(), _rand((512,), g, 0.1).cuda())
    a = ops.conv2d(x, pc)
    for _ in range(3):
        assert torch.equal(ops.conv2d(x, pc), a)
    assert ops.PackedConv(_rand((1536, 512, 1, 1), g).cuda()).w16 is None        # 12 co-tiles do not divide the 32 slots of an XCD
    assert ops.PackedConv(_rand((64, 64, 1, 1), g).cuda()).w16 is None           # the flow's 64 -> 64 convs stay on the igemm kernel


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,stride,ups", [(1, 128, 128, 40, 40, 3, 1, False), (2, 64, 256, 24, 36, 3, 1, False),
                                                          (1, 72, 136, 17, 23, 3, 1, True), (1, 256, 256, 16, 16, 3, 2, False),
                                                          (1, 128, 192, 20, 20, 1, 1, False)])
def test_narrower_output_channel_tiles_give_the_same_bits(B, Cin, Cout, H, W, k, stride, ups):
    """glare_conv_desc.cout_tile: the same conv with 64- and 32-wide workgroup tiles (small launches: more workgroups) -- only the
    assignment of output channels to workgroups changes, every output's contraction order is the same, so the results are bit-equal;
    glare_conv2d_cout_tile picks the default for the BASELINE shapes and a narrower tile for the training crops."""
    from glare_amd import _lib
    import ctypes

    g = torch.Generator().manual_seed(B + Cin + Cout)
    x = _nhwc_bf16(_rand((B, Cin, H, W), g))
    w = _rand((Cout, Cin, k, k), g, (Cin * k * k) ** -0.5).cuda()
    b = _rand((Cout,), g).cuda()
    r = _nhwc_bf16(_rand((B, Cout, H, W), g)) if (stride == 1 and not ups) else None
    ref = ops.conv2d(x, ops.PackedConv(w, b), stride=stride, upsample=ups, act="relu" if r is None else "none", residual=r)
    for tile in (64, 32):
        pc = ops.PackedConv(w, b, cout_tile=tile)
        assert pc.cout_tile == tile and pc.w16 is None
        got = ops.conv2d(x, pc, stride=stride, upsample=ups, act="relu" if r is None else "none", residual=r)
        assert torch.equal(got, ref)
    pick = _lib.lib().glare_conv2d_cout_tile
    i = ctypes.c_int
    assert pick(i(8), i(420), i(620), i(128)) == 128 and pick(i(8), i(105), i(155), i(512)) == 128       # the BASELINE batch: default
    assert pick(i(1), i(256), i(256), i(128)) == 64 and pick(i(1), i(64), i(64), i(512)) == 32          # stage-3 crop
    assert pick(i(2), i(80), i(80), i(512)) == 64 and pick(i(2), i(80), i(80), i(64)) == 64 and pick(i(1), i(8), i(8), i(16)) == 32
    # fused GroupNorm statistics: the 64-wide tile keeps the 4-row wave slabs of the 128-wide one (same sums, another fp32 order);
    # the 32-wide tile's waves cover 2 rows each and are refused
    xs, ws = _nhwc_bf16(_rand((1, 128, 16, 32), g)), _rand((128, 128, 3, 3), g).cuda()
    s128 = ops.conv2d(xs, ops.PackedConv(ws), gn_stats=True)._gn_stats.double().sum(1)
    s64 = ops.conv2d(xs, ops.PackedConv(ws, cout_tile=64), gn_stats=True)._gn_stats.double().sum(1)
    assert float((s128 - s64).abs().max() / s128.abs().max()) < 1e-6
    with pytest.raises(_lib.GlareError):
        ops.conv2d(xs, ops.PackedConv(ws, cout_tile=32), gn_stats=True)


def test_groupnorm_fold_builds_the_per_image_filters_and_the_1x1_kernel_applies_them():
    """glare_attn_fold_groupnorm_f32 against the same algebra in fp64 torch; glare_conv1x1_ws_image_bf16 against eight separate
    convs with those filters; a batch that does not divide the kernel's 64 pixel ranges is refused (the caller then materialises
    the norm)."""
    from glare_amd import _lib

    g = torch.Generator().manual_seed(11)
    B, H, W, C = 4, 10, 14, 512
    x = _nhwc_bf16(_rand((B, C, H, W), g) * 1.5 + 0.7)
    xs = ops.add_bf16(x, torch.zeros_like(x), gn_stats=True)
    stats = xs._gn_stats
    gamma, beta = (_rand((C,), g) * 0.3 + 1.0).cuda(), (_rand((C,), g) * 0.2).cuda()
    wq, wo = (_rand((C, C), g) * C ** -0.5).cuda(), (_rand((C, C), g) * C ** -0.5).cuda()
    bq, bo = _rand((C,), g).cuda(), _rand((C,), g).cuda()
    wq_b, bq_b, wo_b, bo_b = ops.attn_fold_groupnorm(stats, H * W, gamma, beta, 1e-6, wq, bq, wo, bo)
    xf = x.double().view(B, H * W, 32, C // 32)
    mean, var = xf.mean(dim=(1, 3)), xf.var(dim=(1, 3), unbiased=False)                    # [B, 32]
    a = (gamma.double().view(32, -1) / torch.sqrt(var + 1e-6).unsqueeze(-1)).reshape(B, C)
    d = beta.double() - (mean.unsqueeze(-1) * a.view(B, 32, -1)).reshape(B, C)
    ref_wq = a.unsqueeze(2) * wq.double() * a.unsqueeze(1)
    ref_bq = a * (torch.einsum("oc,bc->bo", wq.double(), d) + bq.double())
    ref_wo = wo.double() * a.unsqueeze(1)
    ref_bo = torch.einsum("oc,bc->bo", wo.double(), d) + bo.double()
    rel = lambda u, v: float((u.double() - v).norm() / v.norm())
    assert rel(wq_b, ref_wq) < 3e-3 and rel(wo_b, ref_wo) < 3e-3            # bf16 rounding of the filters (2^-9)
    assert rel(bq_b, ref_bq) < 1e-5 and rel(bo_b, ref_bo) < 1e-5
    r = _nhwc_bf16(_rand((B, C, H, W), g))
    got = ops.conv1x1_per_image(x, wo_b, bo_b, residual=r, gn_stats=True)
    for b in range(B):
        pc = ops.PackedConv(wo_b[b].float()[:, :, None, None].contiguous(), bo_b[b])
        one = ops.conv2d(x[b:b + 1].contiguous(), pc, residual=r[b:b + 1].contiguous())
        assert torch.equal(got[b:b + 1], one)                                              # same kernel, same contraction order
    assert getattr(got, "_gn_stats", None) is not None
    x3 = x[:3].contiguous()
    with pytest.raises(_lib.GlareError):
        ops.conv1x1_per_image(x3, wo_b[:3].contiguous(), bo_b[:3].contiguous())            # 64 % 3 != 0


@pytest.mark.parametrize("k,cout,out_step,f32", [(1, 64, 64, False), (3, 6, 8, True), (3, 64, 64, False)])
def test_grouped_launch_gives_the_bits_of_the_per_group_launches(k, cout, out_step, f32):
    """glare_conv_desc.groups: n filters of one shape on channel slices of one tensor in ONE launch (the flow's z-independent
    coupling nets) == the n separate launches, bit for bit; untouched output channels stay untouched."""
    g = torch.Generator().manual_seed(77 + k + cout)
    G, B, H, W, cin = 5, 2, 13, 37, 64
    x = _nhwc_bf16(_rand((B, G * cin, H, W), g))
    w = _rand((G, cout, cin, k, k), g, 1.0 / (cin * k * k) ** 0.5).cuda()
    b = _rand((G, cout), g, 0.1).cuda()
    pcs = ops.packed_conv_batch(w, b)
    mode = ops.OUT_NHWC_F32 if f32 else ops.OUT_NHWC_BF16
    dt = torch.float32 if f32 else torch.bfloat16
    one = torch.full((B, H, W, G * out_step), 7.0, dtype=dt, device="cuda")
    sep = one.clone()
    ops.conv2d_grouped(x, pcs, cin=cin, in_step=cin, out=one, out_step=out_step, act="relu", out_mode=mode)
    for s in range(G):
        ops.conv2d(x, pcs[s], cin=cin, in_off=cin * s, act="relu", out=sep, out_off=out_step * s, out_mode=mode)
    assert torch.equal(one, sep)
    if out_step > cout:
        assert bool((one.view(B, H, W, G, out_step)[..., cout:] == 7.0).all())
    ref = F.conv2d(x[..., cin:2 * cin].permute(0, 3, 1, 2).float(), w[1].to(torch.bfloat16).float(), b[1], 1, k // 2).relu()
    _check(one[..., out_step:out_step + cout].permute(0, 3, 1, 2), ref, bf16_out=not f32)
    # refusals: channel slices past the pitch; fusions the grouped launch does not carry
    with pytest.raises(RuntimeError):
        ops.conv2d_grouped(x, pcs, cin=cin, in_step=cin, in_off=8, out=one, out_step=out_step, out_mode=mode)


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("B,C,Co,H,W,swish", [(2, 128, 128, 19, 45, True), (1, 256, 256, 9, 33, True), (1, 128, 256, 8, 32, False)])
def test_groupnorm_prologue_is_bit_identical_to_the_separate_pass(prec, B, C, Co, H, W, swish):
    """glare_conv_desc.gn_coef: GroupNorm (+ swish) applied to the halo tile in LDS by the conv's loader == conv(groupnorm(x)) bit for
    bit (same fp32 terms, same 16-bit rounding; zero padding stays zero), incl. ragged tile edges."""
    g = torch.Generator().manual_seed(B * 100 + C + H)
    with ops.use_precision(prec):
        x0 = (_rand((B, H, W, C), g) * 2.0 + 0.3).to(ops.act_dtype()).cuda()
        w0 = _rand((C, C, 3, 3), g, 0.03).cuda()
        x = ops.conv2d(x0, ops.PackedConv(w0, None), gn_stats=True)            # a producer that leaves fused statistics
        gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (_rand((C,), g, 0.2)).cuda()
        w = _rand((Co, C, 3, 3), g, 0.03).cuda()
        b = _rand((Co,), g, 0.1).cuda()
        pc = ops.PackedConv(w, b)
        ref = ops.conv2d(ops.groupnorm(x, gamma, beta, swish=swish), pc, gn_stats=True)
        got = ops.conv2d(x, pc, gn_prologue=(ops.groupnorm_coeffs(x, gamma, beta), swish), gn_stats=True)
    assert torch.equal(got, ref)
    assert torch.equal(got._gn_stats, ref._gn_stats)



def test_groupnorm_prologue_through_the_decoder_modules():
    """GLARE_GN_PROLOGUE=1 (encoder_decoder.GN_PROLOGUE): the VQGAN decoder with its <= 128-channel ResnetBlocks on the prologue form --
    the same bits as the default graph (the measured reason it is off by default is speed, profiles/r04_gn_prologue.txt)."""
    from glare_amd import modules as M
    from glare_amd.modules import encoder_decoder as ED
    from glare_amd.synthetic import seeded_init_

    pv = seeded_init_(M.VQModel().eval(), 1).cuda()
    z = (torch.randn(2, 9, 13, 3, generator=torch.Generator().manual_seed(3)) * 0.5).cuda()
    outs = []
    with torch.no_grad(), ops.use_precision("fp16"):
        for flag in (False, True):
            keep, ED.GN_PROLOGUE = ED.GN_PROLOGUE, flag
            try:
                img, feats = pv.decoder.forward_nhwc(z, want_image=True)
            finally:
                ED.GN_PROLOGUE = keep
            outs.append((img, feats))
    assert torch.equal(outs[0][0], outs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))


def test_exp_activation_without_residual_excludes_the_slab_epilogue_features():
    """include/glare_hip.h, glare_conv_desc.act: sigmoid / swish on the accumulators (no residual) run in the general
    epilogue, which has neither the fused GroupNorm statistics nor the sub-pixel upsample form -- those two combinations are a
    stated GLARE_ERR_UNSUPPORTED, everything around them works and agrees with the two-step form."""
    from glare_amd import _lib

    g = torch.Generator().manual_seed(3)
    x = (torch.randn(1, 16, 32, 128, generator=g) * 0.5).to(ops.act_dtype()).cuda()
    w = torch.randn(128, 128, 3, 3, generator=g).cuda() * 0.03
    b = torch.randn(128, generator=g).cuda() * 0.1
    pc = ops.PackedConv(w, b)
    plain = ops.conv2d(x, pc)                                     # none: slab epilogue
    sw = ops.conv2d(x, pc, act="swish")                           # swish, no residual: general epilogue
    ref = plain.float() * torch.sigmoid(plain.float())
    assert float((sw.float() - ref).abs().max() / ref.abs().max()) < 1.5e-2     # `plain` is rounded to 16 bits before the two-step swish
    with pytest.raises(_lib.GlareError) as e:
        ops.conv2d(x, pc, act="swish", gn_stats=True)
    assert e.value.status == _lib.ERR_UNSUPPORTED
    with pytest.raises(_lib.GlareError) as e:
        ops.conv2d(x, ops.PackedConv(w, b, upsample_subpixel=True), act="sigmoid", upsample=True)
    assert e.value.status == _lib.ERR_UNSUPPORTED
    res = (torch.randn(1, 16, 32, 128, generator=g) * 0.5).to(ops.act_dtype()).cuda()
    out = ops.conv2d(x, pc, act="swish", residual=res, gn_stats=True)            # with a residual the slab epilogue applies it: allowed
    assert hasattr(out, "_gn_stats") and bool(torch.isfinite(out.float()).all())


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_filter_feedback_rounding(precision):
    """ops.filter_feedback_round (round 6): the inference filters' rounding with error feedback per output channel -- bit for bit the
    sequential definition (q_i = round16(w_i + carry), carry = (w_i + carry) - q_i in double), every weight within one 16-bit ulp OF THE
    CHANNEL'S LARGEST WEIGHTS of its value (a tiny weight inherits the carry of its big neighbour), and the sum of a channel's rounding
    errors below half such an ulp (round-to-nearest: a random walk of sqrt(n) / 3.5 ulps)."""
    g = torch.Generator().manual_seed(2)
    w = torch.randn(24, 40, 3, 3, generator=g) * 0.03
    with ops.use_precision(precision):
        dt = ops.act_dtype()
        got = ops.filter_feedback_round(w.cuda()).cpu()
    flat = w.reshape(24, -1).double()
    want = torch.empty_like(flat)
    carry = torch.zeros(24, dtype=torch.float64)
    for i in range(flat.shape[1]):
        t = flat[:, i] + carry
        q = t.float().to(dt).double()
        want[:, i] = q
        carry = t - q
    assert torch.equal(got.reshape(24, -1).double(), want)
    assert torch.equal(got.to(dt).float(), got)                                   # 16-bit numbers: the pack kernel's own rounding is exact
    ulp = 2.0 ** (torch.floor(torch.log2(w.abs().clamp_min(1e-30))) - (10 if precision == "fp16" else 7))
    ulp_max = ulp.reshape(24, -1).max(1).values.view(24, 1, 1, 1)
    assert bool(((got - w).abs() <= ulp_max * 1.0001).all())                     # |error| <= 1/2 ulp(t_i) + |carry|: one ulp of the channel's largest weights
    err_fb = (got.double() - w.double()).reshape(24, -1).sum(1).abs()
    err_rne = (w.to(dt).double() - w.double()).reshape(24, -1).sum(1).abs()
    half_ulp_max = 0.5 * ulp.reshape(24, -1).max(1).values.double()
    assert bool((err_fb <= half_ulp_max * 1.0001).all()) and float(err_rne.mean()) > 4 * float(err_fb.mean())
