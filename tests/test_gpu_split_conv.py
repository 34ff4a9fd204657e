"""GPU: the fp32-class form of the MFMA convolution (glare_conv_desc.k_wrap, ops.PackedConv(split=3)): activation and filter
each a hi / lo pair of 16-bit tensors (22 mantissa bits), contracted as x_hi.w_hi + x_lo.w_hi + x_hi.w_lo in ONE accumulation
over three K segments -- against F.conv2d in fp32 on the UNROUNDED fp32 operands (what the reference's fp32 nn.Conv2d computes,
encoder_decoder.py:88-115).  Tolerance: the dropped x_lo.w_lo term (2^-22) + fp32 summation order: 3.5e-6 of max|ref| asserted
(measured 1.8e-6); the single-pass kernel on the same data is asserted to be >= 30x worse, so the test cannot pass on a launch
that silently ignored the lo halves."""
import pytest
import torch
import torch.nn.functional as F

from glare_amd import ops
from tolerances import within

pytestmark = pytest.mark.gpu


def _pair(x_nchw):
    """fp32 NCHW (cpu) -> the hi / lo pair in NHWC on the device (ops.split_hilo)."""
    x = x_nchw.permute(0, 2, 3, 1).contiguous().cuda()
    return ops.split_hilo(x)


def _val(t):
    lo = getattr(t, "_lo", None)
    return t.float() if lo is None else t.float() + lo.float()


def _err(got_nhwc, ref_nchw):
    ref = ref_nchw.permute(0, 2, 3, 1)
    return float((got_nhwc - ref).abs().max() / ref.abs().max())


CASES = [  # B, Cin, Cout, H, W, k, stride
    (1, 128, 128, 16, 40, 3, 1),
    (2, 64, 256, 9, 33, 3, 1),
    (1, 256, 128, 8, 32, 1, 1),      # nin_shortcut-like
    (1, 128, 128, 17, 35, 3, 2),     # Downsample
    (1, 512, 3, 12, 31, 3, 1),       # conv_out
    (1, 64, 64, 11, 35, 1, 1),       # the flow's 1x1 (64-wide tile)
    (1, 64, 4, 10, 37, 3, 1),        # the flow's last conv
]


@pytest.mark.parametrize("reuse", [True, False])
@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("B,Cin,Cout,H,W,k,stride", CASES)
def test_split3_conv_matches_the_fp32_conv(monkeypatch, prec, B, Cin, Cout, H, W, k, stride, reuse):
    """reuse: the K order k_wrap = 2 (every x_hi halo tile staged once for w_hi and w_lo: the 3x3 convs, the default) against round 4's
    segment order k_wrap = 1 -- the same three products in another accumulation order, the same bound."""
    monkeypatch.setattr(ops, "SPLIT_A_REUSE", reuse)
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H + k)
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn((Cout,), generator=g) * 0.1
    if stride == 2:   # pad (0,1,0,1) then valid 3x3 stride 2 (encoder_decoder.py:71-73)
        ref = F.conv2d(F.pad(x.cuda(), (0, 1, 0, 1)), w.cuda(), b.cuda(), 2, 0)
    else:
        ref = F.conv2d(x.cuda(), w.cuda(), b.cuda(), 1, k // 2)
    with ops.use_precision(prec):
        xp = _pair(x)
        pc = ops.PackedConv(w.cuda(), b.cuda(), split=3)
        assert pc.k_wrap == (2 if reuse and k == 3 and Cin % 16 == 0 else 1)
        out = ops.conv2d(xp, pc, stride=stride, out_mode=ops.OUT_NHWC_F32)
        plain = ops.conv2d(xp, ops.PackedConv(w.cuda(), b.cuda()), stride=stride, out_mode=ops.OUT_NHWC_F32)
    e3, e1 = _err(out, ref), _err(plain, ref)
    within(e3, 3.5e-6 if prec == "fp16" else 1.0e-5, prec)    # measured on MI355X (max over the cases): fp16 1.76e-6, bf16 5.0e-6
    assert e1 > 30 * e3, (e1, e3)


def test_split3_hilo_output_residual_and_statistics():
    """The ResnetBlock tail in fp32-class form: conv2(n2) + residual with every tensor a pair, GroupNorm statistics fused."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 128, 12, 40), generator=g)
    r = torch.randn((2, 128, 12, 40), generator=g)
    w = torch.randn((128, 128, 3, 3), generator=g) * 0.03
    b = torch.randn((128,), generator=g) * 0.1
    ref = F.conv2d(x.cuda(), w.cuda(), b.cuda(), 1, 1) + r.cuda()
    with ops.use_precision("fp16"):
        out = ops.conv2d(_pair(x), ops.PackedConv(w.cuda(), b.cuda(), split=3), residual=_pair(r), hilo=True, gn_stats=True)
        within(_err(_val(out), ref), 2.2e-6)          # measured 1.07e-6
        # GroupNorm of the pair, output as a pair
        gamma, beta = torch.rand(128, generator=g).cuda() + 0.5, torch.randn(128, generator=g).cuda() * 0.1
        y = ops.groupnorm(out, gamma, beta, swish=True, pair=True)
        yref = F.silu(F.group_norm(ref, 32, gamma, beta, 1e-6))
        within(_err(_val(y), yref), 2.3e-6)          # measured 1.13e-6
        assert float((y._lo.float().abs().max())) > 0


def test_tile_reuse_order_error_behaviour_and_agreement_with_the_segment_order(monkeypatch):
    """glare_conv_desc.k_wrap = 2 (include/glare_hip.h): 3x3 only, needs the lo half with the hi half's channel count; any
    other value is invalid.  Against k_wrap = 1 on the same operands (pair in, pair residual, pair out, statistics) the result differs
    by accumulation order only."""
    from glare_amd import _lib
    g = torch.Generator().manual_seed(21)
    x = torch.randn((1, 64, 9, 33), generator=g)
    w3 = torch.randn((32, 64, 3, 3), generator=g) * 0.05
    with ops.use_precision("fp16"):
        xp = _pair(x)
        pc = ops.PackedConv(w3.cuda(), None, split=3)
        assert pc.k_wrap == 2
        out = torch.empty(1, 9, 33, 32, dtype=torch.float16, device="cuda")
        d = ops.ConvDesc()
        d.in_, d.B, d.H, d.W, d.Cin, d.in_pitch = xp.data_ptr(), 1, 9, 33, 64, 64
        d.in2, d.Cin2, d.in2_pitch = xp._lo.data_ptr(), 64, 64
        d.out, d.Cout, d.out_pitch = out.data_ptr(), 32, 32
        d.weight_packed, d.ksize, d.stride, d.out_mode = pc.packed.data_ptr(), 3, 1, ops.OUT_NHWC_BF16
        lib = _lib.lib()
        import ctypes
        for k_wrap, ksize, in2, cin2, want in ((2, 3, True, 64, 0), (2, 1, True, 64, _lib.ERR_UNSUPPORTED), (3, 3, True, 64, _lib.ERR_INVALID),
                                               (-1, 3, True, 64, _lib.ERR_INVALID), (2, 3, False, 0, _lib.ERR_INVALID), (2, 3, True, 48, _lib.ERR_INVALID)):
            d.k_wrap, d.ksize = k_wrap, ksize
            d.in2, d.Cin2 = (xp._lo.data_ptr() if in2 else None), cin2
            assert lib.glare_conv2d_bf16(ctypes.byref(d), ops.stream_handle()) == want, (k_wrap, ksize, in2, cin2)
        torch.cuda.synchronize()
    x = torch.randn((2, 128, 12, 40), generator=g)
    r = torch.randn((2, 128, 12, 40), generator=g)
    w = torch.randn((128, 128, 3, 3), generator=g) * 0.03
    b = torch.randn((128,), generator=g) * 0.1
    res = {}
    with ops.use_precision("fp16"):
        for reuse in (True, False):
            monkeypatch.setattr(ops, "SPLIT_A_REUSE", reuse)
            o = ops.conv2d(_pair(x), ops.PackedConv(w.cuda(), b.cuda(), split=3), residual=_pair(r), hilo=True, gn_stats=True)
            res[reuse] = (_val(o), o._gn_stats.clone())
    ref = F.conv2d(x.cuda(), w.cuda(), b.cuda(), 1, 1) + r.cuda()
    within(_err(res[True][0], ref), 2.2e-6)
    within(_err(res[False][0], ref), 2.2e-6)
    assert float((res[True][0] - res[False][0]).abs().max() / ref.abs().max()) < 1.5e-6
    assert torch.allclose(res[True][1], res[False][1], rtol=1e-5, atol=1e-3)


def test_split3_1x1_hilo_output_both_tiles():
    g = torch.Generator().manual_seed(12)
    for cin, cout in ((128, 256), (64, 64)):
        x = torch.randn((1, cin, 9, 37), generator=g)
        w = torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5
        b = torch.randn((cout,), generator=g) * 0.1
        ref = F.relu(F.conv2d(x.cuda(), w.cuda(), b.cuda()))
        with ops.use_precision("fp16"):
            out = ops.conv2d(_pair(x), ops.PackedConv(w.cuda(), b.cuda(), split=3), act="relu", hilo=True)
        within(_err(_val(out), ref), 1.1e-6, cout)      # measured 5.3e-7 / 4.3e-7


def test_split3_grouped_launch_equals_per_group_launches():
    """The flow's z-independent nets: n filters of one shape on channel slices of one pair, one launch (blockIdx.y = group)."""
    g = torch.Generator().manual_seed(13)
    n = 3
    x = torch.randn((1, n * 64, 10, 33), generator=g)
    for k, cout, step, f32 in ((1, 64, 64, False), (3, 6, 8, True)):
        w = torch.randn((n, cout, 64, k, k), generator=g) / (64 * k * k) ** 0.5
        b = torch.randn((n, cout), generator=g) * 0.1
        with ops.use_precision("fp16"):
            xp = _pair(x)
            pcs = ops.packed_conv_batch(w.cuda(), b.cuda(), split=3)
            if f32:
                out = torch.zeros(1, 10, 33, n * step, dtype=torch.float32, device="cuda")
                ops.conv2d_grouped(xp, pcs, cin=64, in_step=64, out=out, out_step=step, out_mode=ops.OUT_NHWC_F32)
                got = out
            else:
                out = torch.zeros(1, 10, 33, n * step, dtype=torch.float16, device="cuda")
                ops.conv2d_grouped(xp, pcs, cin=64, in_step=64, out=out, out_step=step, out_lo=torch.zeros_like(out), act="relu")
                got = _val(out)
        for i in range(n):
            ref = F.conv2d(x[:, 64 * i:64 * i + 64].cuda(), w[i].cuda(), b[i].cuda(), 1, k // 2)
            if not f32:
                ref = F.relu(ref)
            within(_err(got[..., step * i:step * i + cout], ref), 1.5e-6, "%d/%d" % (k, i))     # measured <= 7.3e-7


def test_split2_filter_remainder_only():
    """A 16-bit activation against a 22-bit filter (K segments [x | x] . [w_hi | w_lo]): exact in the filter, x as stored."""
    g = torch.Generator().manual_seed(14)
    x = torch.randn((1, 128, 9, 35), generator=g).half().float()
    w = torch.randn((128, 128, 3, 3), generator=g) * 0.03
    ref = F.conv2d(x.cuda(), w.cuda(), None, 1, 1)
    with ops.use_precision("fp16"):
        xh = x.permute(0, 2, 3, 1).contiguous().cuda().half()
        out = ops.conv2d(xh, ops.PackedConv(w.cuda(), None, split=2), out_mode=ops.OUT_NHWC_F32)
    within(_err(out, ref), 2.0e-6)                  # measured 1.0e-6


def test_flow_h1_pair():
    g = torch.Generator().manual_seed(15)
    z = torch.randn((1, 9, 21, 3), generator=g).cuda()
    ftA = torch.randn((1, 9, 21, 128), generator=g).cuda()
    wz = torch.randn((64, 9), generator=g).cuda() * 0.2
    with ops.use_precision("fp16"):
        h1 = torch.empty(1, 9, 21, 64, dtype=torch.float16, device="cuda")
        h1._lo = torch.empty_like(h1)
        ops.flow_h1(z, ftA, 64, wz, out=h1)
        plain = ops.flow_h1(z, ftA, 64, wz)
    ref = F.relu(ftA[..., 64:] + F.conv2d(z[..., :1].permute(0, 3, 1, 2), wz.view(64, 1, 3, 3), None, 1, 1).permute(0, 2, 3, 1))
    assert torch.equal(h1, plain)
    within(float((_val(h1) - ref).abs().max() / ref.abs().max()), 4.5e-7)   # measured 2.2e-7


@pytest.mark.parametrize("B,N", [(8, 300), (1, 700)])       # one workgroup per query block / keys split over 4 workgroups
def test_attention_pair_output_keeps_the_accumulators(B, N):
    """attention_kv512(pair=True): hi + lo of the output equals softmax(q.x^T) x on the stored 16-bit operands to ~1e-5 (the
    probabilities are rounded to 16 bits inside the kernel: that is the kernel's arithmetic, not storage), where hi alone carries
    the 2^-11 storage rounding; hi is bit-identical to the plain call."""
    g = torch.Generator().manual_seed(N)
    x = (torch.randn((B, N, 512), generator=g)).half().cuda()
    q = (torch.randn((B, N, 512), generator=g) * 0.08).half().cuda()
    with ops.use_precision("fp16"):
        a = ops.attention_kv512(q, x, N, pair=True)
        plain = ops.attention_kv512(q, x, N)
    assert torch.equal(a, plain)
    s = torch.einsum("bid,bjd->bij", q.double(), x.double()) * 0.6931471805599453          # the kernel's scores are log2-domain
    ref = torch.einsum("bij,bjd->bid", torch.softmax(s, -1), x.double()).float()
    e_pair = float((_val(a) - ref).abs().max() / ref.abs().max())
    e_hi = float((a.float() - ref).abs().max() / ref.abs().max())
    within(e_pair, 4e-4, N)              # P rounded to fp16 before P.V: ~1e-4
    assert e_hi >= e_pair
    # the pair removes the STORAGE rounding: against the kernel's own fp32 result (hi + lo) hi is 2^-12-ish, lo exact to 2^-22
    assert float((a._lo.float().abs().max())) > 0
    assert float((a._lo.float().abs() / (a.float().abs() + 1e-3)).max()) < 2.0 ** -10


@pytest.mark.parametrize("cin,cout,H,W", [(128, 256, 9, 37), (256, 512, 7, 45), (512, 512, 11, 29), (512, 64, 5, 33), (256, 128, 8, 32)])
def test_split3_1x1_on_the_weight_stationary_kernel(cin, cout, H, W):
    """glare_conv1x1_ws_split_bf16 (round 4): the fp32-class 1x1 convs of the conditional encoder (nin_shortcut, AttnBlock's folded
    query / output projections) on the weight-stationary kernel, 64-cout tiles with (w_hi | w_lo) resident.  Three modes -- pair out,
    16-bit out, pair residual + fused statistics -- against F.conv2d in fp32 on the unrounded operands and against the implicit-GEMM
    launch of the same PackedConv (ops.CONV1X1_WEIGHT_STATIONARY = False); ragged pixel counts (H * W not a multiple of 32)."""
    g = torch.Generator().manual_seed(cin + cout + H)
    B = 2
    x = torch.randn((B, cin, H, W), generator=g)
    r = torch.randn((B, cout, H, W), generator=g)
    w = torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    ref = F.conv2d(x.cuda(), w.cuda(), b.cuda())
    with ops.use_precision("fp16"):
        xp, rp = _pair(x), _pair(r)
        pc = ops.PackedConv(w.cuda(), b.cuda(), split=3)
        assert pc.w16_lo is not None
        got = {}
        for ws in (True, False):
            ops.CONV1X1_WEIGHT_STATIONARY = ws
            try:
                pair = ops.conv2d(xp, pc, hilo=True)
                plain = ops.conv2d(xp, pc, act="relu")
                full = ops.conv2d(xp, pc, residual=rp, hilo=True, gn_stats=(cout % 128 == 0))
            finally:
                ops.CONV1X1_WEIGHT_STATIONARY = True
            got[ws] = (pair, plain, full)
        pair, plain, full = got[True]
        within(_err(_val(pair), ref), 1.9e-6, "pair")                                     # measured <= 9.5e-7
        assert plain._lo is None if hasattr(plain, "_lo") else True
        within(_err(plain.float(), F.relu(ref)), 8.8e-4, "16-bit")                         # measured 4.4e-4: one fp16 rounding of the output
        within(_err(_val(full), ref + r.cuda()), 1.6e-6, "residual")                     # measured <= 7.7e-7
        # the two kernels agree to fp32 summation order
        within(_err(_val(pair), _val(got[False][0]).permute(0, 3, 1, 2)), 2.3e-6, "vs igemm")          # measured <= 1.13e-6
        assert torch.equal(plain, got[False][1]) or _err(plain.float(), got[False][1].float().permute(0, 3, 1, 2)) < 6e-4
        if cout % 128 == 0:     # fused statistics: GroupNorm of the pair from them equals GroupNorm of the fp32 value
            gamma, beta = torch.rand(cout, generator=g).cuda() + 0.5, torch.randn(cout, generator=g).cuda() * 0.1
            y = ops.groupnorm(full, gamma, beta, swish=False, pair=True)
            yref = F.group_norm(ref + r.cuda(), 32, gamma, beta, 1e-6)
            within(_err(_val(y), yref), 1.5e-6, "gn")                                        # measured <= 7.2e-7
