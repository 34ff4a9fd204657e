"""GPU: the fp16 precision (libglare_hip_f16.so -- the inference kernels with IEEE-half activations and filters, the reference's own
autocast dtype) and the end-to-end parity table of both precisions in both weight regimes.

  * kernel level: every kernel family of the inference path under `ops.use_precision("fp16")` against an fp32 torch reference on
    the same fp16-rounded operands (tolerance: one fp16 rounding of the output, 2^-11, plus summation order);
  * graph level: every stage on the oracle's inputs (bounds = the bf16 suite's / 8: 8x less rounding was measured as 8x less error);
  * end to end, 100x156 and 400x600, adversarial (synthetic.seeded_init_) and representative (synthetic.representative_init_)
    weights, bf16 and fp16: codebook-index agreement, PSNR(ours, oracle), and |PSNR(ours, GT) - PSNR(oracle, GT)| on the FULL path
    (not only with the oracle's indices substituted) -- the north_star's tolerance as measured, each bound <= 2x its measurement
    (DESIGN.md section 4 has the table and why bit-exact indices end to end are out of reach of any 16-bit activation format).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from glare_amd import modules as M
from glare_amd import ops
from glare_amd.synthetic import representative_init_, seeded_init_, synthetic_lowlight, synthetic_pair
from oracle import torch_ref as O

pytestmark = pytest.mark.gpu

from tolerances import TOL, within  # noqa: E402


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def h16(x):
    return x.to(torch.float16).float()


def nhwc16(x):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.float16).cuda()


def _check16(got, ref, f32_out=False):
    ref, got = ref.float().cpu(), got.float().cpu()
    tol = (1e-5 if f32_out else 2.0 ** -11) * ref.abs() + 3e-4 * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), "max err %g of max %g at %d elems" % (float((got - ref).abs().max()), float(ref.abs().max()), int(bad.sum()))


# ---- kernel level ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Cin,Cout,H,W,k", [(1, 128, 128, 16, 40, 3), (2, 64, 256, 9, 33, 3), (2, 512, 512, 7, 45, 1),
                                              (1, 64, 6, 10, 37, 3), (1, 24, 40, 6, 10, 3)])
def test_fp16_conv_matches_fp32_reference(B, Cin, Cout, H, W, k):
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(h16(x).cuda(), h16(w).cuda(), b.cuda(), 1, k // 2)
    with ops.use_precision("fp16"):
        pc = ops.PackedConv(w.cuda(), b.cuda())
        assert pc.packed.dtype == torch.float16
        out = ops.conv2d(nhwc16(x), pc)
        assert out.dtype == torch.float16
        _check16(out.permute(0, 3, 1, 2), ref)
        # residual + activation + fused GroupNorm statistics path, fp32 output
        if Cout % 128 == 0 and k == 3:
            r = torch.randn(B, Cout, H, W, generator=g)
            out2 = ops.conv2d(nhwc16(x), pc, residual=nhwc16(r), act="swish", gn_stats=True)
            ref2 = F.silu(ref + h16(r).cuda())
            _check16(out2.permute(0, 3, 1, 2), ref2)
            y = ops.groupnorm(out2, torch.ones(Cout).cuda(), torch.zeros(Cout).cuda(), swish=False)
            refn = F.group_norm(out2.float().permute(0, 3, 1, 2), 32, eps=1e-6)
            assert rel(y.permute(0, 3, 1, 2), refn) < 6e-4
    with pytest.raises(AssertionError):   # a filter packed under one precision is refused under the other
        ops.conv2d(nhwc16(x).to(torch.bfloat16), pc)


def test_fp16_upsample_downsample_and_small_convs():
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 64, 9, 21, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.04
    b = torch.randn(64, generator=g) * 0.1
    xb, wb = h16(x).cuda(), h16(w).cuda()
    with ops.use_precision("fp16"):
        pc = ops.PackedConv(w.cuda(), b.cuda())
        ref = F.conv2d(F.interpolate(xb, scale_factor=2.0, mode="nearest"), wb, b.cuda(), 1, 1)
        _check16(ops.conv2d(nhwc16(x), pc, upsample=True).permute(0, 3, 1, 2), ref)
        sp = ops.PackedConv(w.cuda(), b.cuda(), upsample_subpixel=True)     # the pre-summed taps are rounded once more
        assert rel(ops.conv2d(nhwc16(x), sp, upsample=True).permute(0, 3, 1, 2), ref) < 1e-3
        ref = F.conv2d(F.pad(xb, (0, 1, 0, 1)), wb, b.cuda(), 2, 0)
        _check16(ops.conv2d(nhwc16(x), pc, stride=2).permute(0, 3, 1, 2), ref)
        # 3 -> 128 direct conv from an NCHW fp32 image (conv_in), fp16 output
        img = torch.randn(2, 3, 12, 20, generator=g).cuda()
        w3, b3 = (torch.randn(128, 3, 3, 3, generator=g) * 0.2).cuda(), (torch.randn(128, generator=g) * 0.1).cuda()
        out = ops.conv2d_smallcin(img, (3 * 240, 240, 20, 1), (2, 12, 20), w3, b3)
        assert out.dtype == torch.float16
        _check16(out.permute(0, 3, 1, 2), F.conv2d(img, w3, b3, 1, 1))
        # layout round trip
        t = torch.randn(2, 24, 5, 7, generator=g).cuda()
        assert torch.equal(ops.nhwc_to_nchw(ops.nchw_to_nhwc(t)), h16(t.cpu()).cuda())


def test_fp16_weight_stationary_1x1_and_attention():
    g = torch.Generator().manual_seed(9)
    B, H, W, C = 2, 9, 29, 512
    N = H * W
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 1, 1, generator=g) / C ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    with ops.use_precision("fp16"):
        pc = ops.PackedConv(w.cuda(), b.cuda())
        assert pc.w16 is not None and pc.w16.dtype == torch.float16
        out = ops.conv2d(nhwc16(x), pc)
        _check16(out.permute(0, 3, 1, 2), F.conv2d(h16(x).cuda(), h16(w).cuda(), b.cuda()))
        # shared-K/V attention against an fp32 softmax on the same fp16 operands (P is rounded to fp16 inside the kernel)
        q = (torch.randn(B, N, C, generator=g) * 0.15).to(torch.float16).cuda()
        kv = torch.randn(B, N, C, generator=g).to(torch.float16).cuda()
        got = ops.attention_kv512(q, kv, N, key_splits=1)
        s = torch.einsum("bic,bjc->bij", q.float(), kv.float()) * float(np.log(2.0))     # the kernel works in log2 units
        ref = torch.einsum("bij,bjc->bic", torch.softmax(s, dim=2), kv.float())
        assert got.dtype == torch.float16
        assert rel(got, ref) < 6e-4, rel(got, ref)
        got2 = ops.attention_kv512(q, kv, N, key_splits=3)
        assert rel(got2, ref) < 6e-4


def test_fp16_elementwise_and_dcn():
    g = torch.Generator().manual_seed(12)
    a = torch.randn(2, 10, 14, 128, generator=g).to(torch.float16).cuda()
    b = torch.randn(2, 10, 14, 128, generator=g).to(torch.float16).cuda()
    with ops.use_precision("fp16"):
        s = ops.add_bf16(a, b, gn_stats=True)                      # exported as glare_add_groupnorm_stats_f16
        assert torch.equal(s, (a.float() + b.float()).to(torch.float16))
        y = ops.groupnorm(s, torch.ones(128).cuda(), torch.zeros(128).cuda(), swish=True)
        refn = F.silu(F.group_norm(s.float().permute(0, 3, 1, 2), 32, eps=1e-6))
        assert rel(y.permute(0, 3, 1, 2), refn) < 6e-4
        m = ops.mix(a, b, 0.3)                                     # the argument is Mix.w, a logit (deformableDecoder_arch.py:587-590)
        f = 1.0 / (1.0 + np.exp(-0.3))
        assert rel(m, f * a.float() + (1 - f) * b.float()) < 6e-4
        xw = torch.randn(2, 10, 14, 128, generator=g).cuda()
        r = ops.mean_rescale(a, xw)
        ratio = a.float().mean(dim=(1, 2, 3), keepdim=True) / xw.mean(dim=(1, 2, 3), keepdim=True)
        assert rel(r, a.float() + xw * ratio) < 1e-3
        # DCN: fp16 x through the fast kernel == fp32 x through the general kernel up to the input rounding
        x = torch.randn(1, 20, 36, 128, generator=g)
        wd = (torch.randn(128, 128, 3, 3, generator=g) * 0.03).cuda()
        bd = (torch.randn(128, generator=g) * 0.1).cuda()
        plane = (20 * 36 + 63) // 64 * 64
        om = (torch.randn(1, 108, plane, generator=g) * 1.5).cuda()
        pd = ops.PackedDcn(wd, bd, 4)
        got = ops.mdcn_forward_nhwc(x.to(torch.float16).cuda(), om, pd)
        ref = ops.mdcn_forward_nhwc(h16(x).cuda(), om, pd)
        assert rel(got, ref) < 2e-4, rel(got, ref)
    from glare_amd import _lib

    with ops.use_precision("fp16"):   # since round 4 the half library is a build of every source: the training entry points resolve there
        assert _lib.lib().glare_adam_step_f32 is not None and _lib.lib().glare_gemm_nt_bf16 is not None
        with pytest.raises(_lib.GlareError):   # ... and a name it does not have is loud, never a silent bf16 kernel on fp16 data
            _lib.lib().glare_no_such_entry_bf16


# ---- graph level -------------------------------------------------------------------------------------------------------
def _product(og, ov):
    pg, pv = M.VQLLFLOWDeformable().eval(), M.VQModel().eval()
    pg.load_state_dict(og.state_dict(), strict=True)
    pv.load_state_dict(ov.state_dict(), strict=True)
    return pg.cuda(), pv.cuda()


def _oracle(regime):
    if regime.startswith("representative"):     # "representative2" / "3": a second / third trained-like weight set (weight seed 1 / 2)
        wseed = {"2": 1, "3": 2}.get(regime[-1], 0)
        return representative_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), O.VQModel().eval(), wseed)
    return seeded_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), 0), seeded_init_(O.VQModel().eval(), 1)


def _image(regime, h, w, seed):
    return O.preprocess(synthetic_pair(1, h, w, seed=seed)[0][0] if regime.startswith("representative") else synthetic_lowlight(1, h, w, seed=seed)[0])


_CACHE = {}


def setup(regime, h, w, seed):
    key = (regime, h, w, seed)
    if key not in _CACHE:
        threads = torch.get_num_threads()
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        try:
            og, ov = _oracle(regime)
            lr = _image(regime, h, w, seed)
            with torch.no_grad():
                ref = og.stages(ov, lr)
        finally:
            torch.set_num_threads(threads)
        pg, pv = _product(og, ov)
        _CACHE[key] = (og, ov, pg, pv, lr, ref)
    return _CACHE[key]


# relative L2 error of each stage on the oracle's inputs under fp16; bound = 2x measured.  Stages A + B run the fp32-class convs
# (round 4): their tensors are hi / lo pairs and are compared as hi + lo (the hi half alone carries fp16's 2.3e-4 storage rounding)
TOL16 = {"cond_feat": 5.4e-6,    # 2.67e-6   (round 3, 16-bit tensors: 2.45e-4)
         "color_map": 4.7e-5,    # 2.34e-5   (7.88e-4)
         "mid_feat0": 9.0e-7,    # 4.41e-7   (2.34e-4)
         "mid_feat1": 4.1e-6,    # 2.04e-6   (5.65e-4)
         "latent": 1.9e-6,       # 9.22e-7   (3.46e-4)
         "code_feat0": 2.58e-3,  # 1.29e-3
         "code_feat1": 3.97e-3,  # 1.99e-3
         "vq_rec": 4.17e-3,      # 2.09e-3
         "aft_out": 3.0e-3}      # 1.50e-3


def test_fp16_stage_parity_against_oracle():
    """Every stage on the oracle's inputs under fp16 (20x36 image): 8x less rounding than bf16 measures as 8-16x less error
    (tests/tolerances.py TOL holds the bf16 bounds)."""
    og, ov, pg, pv, lr, ref = setup("adversarial", 20, 36, 7)
    nhwc = lambda t, a16=True: ops.nchw_to_nhwc(t.cuda(), bf16=a16)
    nchw = lambda t: ops.nhwc_to_nchw(t).cpu()
    pair = lambda t: nchw(t) + nchw(t._lo)                      # the value of a hi / lo pair
    with torch.no_grad(), ops.use_precision("fp16"):
        enc = pg.RRDB.forward_nhwc(lr.cuda())
        assert enc["cond_feat"].dtype == torch.float16 and enc["cond_feat"]._lo is not None
        within(rel(pair(enc["cond_feat"]), ref["enc"]["cond_feat"]), TOL16["cond_feat"])
        within(rel(nchw(enc["color_map"]), ref["enc"]["color_map"]), TOL16["color_map"])
        for i, (a, b) in enumerate(zip(enc["mid_feat"], ref["enc"]["mid_feat"])):
            within(rel(pair(a), b), TOL16["mid_feat%d" % i], tag=i)
            assert rel(nchw(a), b) < 4e-4                       # the hi half alone (what the AFT decoder's skip inputs read)
        # the flow on the oracle's conditional features, handed over the way the encoder hands them over: as a pair
        ft = ops.split_hilo(nhwc(ref["enc"]["cond_feat"], False))
        z = pg.flowUpsamplerNet.decode_nhwc(nhwc(ref["enc"]["color_map"], False), ft)
        within(rel(nchw(z), ref["latent"]), TOL16["latent"])
        z16 = pg.flowUpsamplerNet.decode_nhwc(nhwc(ref["enc"]["color_map"], False), nhwc(ref["enc"]["cond_feat"]))
        assert rel(nchw(z16), ref["latent"]) < 6.9e-4           # a 16-bit cond_feat selects the single-pass nets (round 3: 3.46e-4)
        idx, img, feats = pv.decode_nhwc(nhwc(ref["latent"], False), want_image=True)
        assert torch.equal(idx.cpu(), ref["indices"])
        for i, (a, b) in enumerate(zip(feats, ref["code_feats"])):
            within(rel(nchw(a), b), TOL16["code_feat%d" % i], tag=i)
        within(rel(img.cpu(), ref["vq_rec"]), TOL16["vq_rec"])
        out = pg.deformable_decoder.forward_nhwc(nhwc(ref["latent"], False), [nhwc(f) for f in ref["code_feats"]],
                                                 [nhwc(f) for f in ref["enc"]["mid_feat"]])
        within(rel(out.cpu(), ref["out"]), TOL16["aft_out"])


def correlated_gt(ref_img, db=27.0, seed=5):
    rng = np.random.default_rng(seed)
    gt = np.clip(ref_img + rng.normal(0, 10 ** (-db / 20), ref_img.shape), 0, 1)
    return np.round(gt * 255).astype(np.uint8)


def audit_flips(r, ref, ov):
    """oracle/audit.py on one scene: every token whose index differs from the oracle's must be a near-tie the measured latent error
    explains (margin on the ORACLE's latent <= 2 |dz| |de|, fp64).  Asserts, and returns (flips, worst margin / bound ratio)."""
    from oracle.audit import flip_audit

    C = ref["latent"].shape[1]
    a = flip_audit(ref["latent"].permute(0, 2, 3, 1).reshape(-1, C).numpy(), r["latent"].float().cpu().reshape(-1, C).numpy(),
                   ref["indices"].reshape(-1).numpy(), r["indices"].cpu().reshape(-1).numpy(),
                   ov.quantize.embedding.weight.detach().float().cpu().numpy())
    assert not a["violations"], ("index flips the latent error cannot explain (quantize.py:280-285)", a)
    return a["flips"], a["worst_ratio"]


def e2e_metrics(out, out_ref, h):
    a, b = O.postprocess(out, h), O.postprocess(out_ref, h)
    gt = correlated_gt(b)
    pa, pb = O.psnr(gt / 255, O.postprocess(out, h, gt)), O.psnr(gt / 255, O.postprocess(out_ref, h, gt))
    return {"psnr_vs_oracle": float(O.psnr(a, b)), "delta": float(abs(pa - pb))}


# (regime, precision) -> bounds at 400x600 / 100x156: agree >=, PSNR(ours, oracle) >= [dB], full-path |dPSNR vs GT| <= [dB].
# Measured on MI355X (tools/parity_scenes.py / parity_probe.py, seed 11 / 21): see the comment on each row; every bound <= 2x the
# measured miss (the |dPSNR| figures of the fp16 rows sit at the noise floor of the metric, 0.001-0.005 dB whatever the indices:
# bound 0.01 dB = a fifth of BASELINE.json's 0.05).  fp16 = the inference default: fp32-class convs on hi / lo pairs in the
# conditional encoder and the flow (round 4), 16-bit single-pass convs in the two decoders, split fp32-class DCN.
BOUNDS = {
    (400, "adversarial", "bf16"): (0.970, 27.6, 3.4),        # 0.98415, 30.60 dB, 1.70 dB
    (400, "adversarial", "fp16"): (0.99988, 60.6, 0.01),     # 0.99994, 63.65 dB, 0.0020 dB (round 3: 0.99853, 40.57 dB, 0.22 dB)
    (400, "representative", "bf16"): (0.30, 32.2, 1.07),     # 0.47637, 35.19 dB, 0.53 dB
    (400, "representative", "fp16"): (0.9995, 61.8, 0.01),   # 0.99969, 64.85 dB, 0.0023 dB (round 3: 0.94015, 46.16 dB, 0.041 dB); 12 scenes: the next test
    (100, "adversarial", "bf16"): (0.950, 25.5, 5.1),        # 0.97576, 28.48 dB, 2.54 dB
    (100, "adversarial", "fp16"): (0.9984, 63.2, 0.01),      # 1.00000 (bit-exact; bound = 2 of 1 320 tokens), 66.21 dB, 0.0028 dB
    (100, "representative", "bf16"): (0.30, 33.4, 0.76),     # 0.51742, 36.47 dB, 0.38 dB
    (100, "representative", "fp16"): (0.9969, 54.9, 0.01),   # 0.99848 (2 tokens), 57.95 dB, 0.0042 dB
}


def check_e2e(regime, h, w, seed, capsys):
    og, ov, pg, pv, lr, ref = setup(regime, h, w, seed)
    rows = {}
    for prec in ("bf16", "fp16"):
        with torch.no_grad():
            r = pg.reverse_flow_nhwc(pv, lr.cuda(), precision=prec)
            assert r["enc"]["cond_feat"].dtype == (torch.float16 if prec == "fp16" else torch.bfloat16)
            agree = float((r["indices"].cpu() == ref["indices"]).float().mean())
            audit_flips(r, ref, ov)          # both precisions: whatever the latent error, a flipped token must be explained by it
            full = e2e_metrics(r["out"].cpu(), ref["out"], h)
            with ops.use_precision(prec):    # the same run with the VQ decoder fed the oracle's latent (=> its indices)
                _, _, feats_i = pv.decode_nhwc(ops.nchw_to_nhwc(ref["latent"].cuda(), bf16=False), want_image=False)
                out_i = pg.deformable_decoder.forward_nhwc(r["latent"], feats_i, r["enc"]["mid_feat"]).cpu()
            forced = e2e_metrics(out_i, ref["out"], h)
        rows[prec] = (agree, full, forced, float(rel(ops.nhwc_to_nchw(r["latent"]).cpu(), ref["latent"])))
        assert torch.isfinite(r["out"]).all()
    with capsys.disabled():
        for prec, (agree, full, forced, lat) in rows.items():
            print("\n[e2e %dx%d %-14s %s] latent rel %.2e | index agreement %.5f | full path: PSNR(ours,oracle) %.2f dB, |dPSNR vs GT| %.4f dB"
                  " | oracle's indices: %.2f dB, %.4f dB" % (h, w, regime, prec, lat, agree, full["psnr_vs_oracle"], full["delta"],
                                                              forced["psnr_vs_oracle"], forced["delta"]), end="")
        print()
    for prec, (agree, full, forced, lat) in rows.items():
        b_agree, b_psnr, b_delta = BOUNDS[(h, regime, prec)]
        assert agree >= b_agree, (regime, prec, agree)
        assert full["psnr_vs_oracle"] >= b_psnr, (regime, prec, full)
        assert full["delta"] <= b_delta, (regime, prec, full)       # the FULL path, our own indices
        # with the oracle's indices substituted: fp16 meets BASELINE.json's tolerance outright; bf16's smooth error of the two
        # decoders alone (47-48 dB) is worth up to 0.08 dB against a 27 dB ground truth (measured 0.0018-0.079)
        assert forced["delta"] <= (0.05 if prec == "fp16" else 0.16), (regime, prec, forced)
    # fp16 is the default of the inference entry point, and it is the better one on every figure
    with torch.no_grad():
        assert pg.reverse_flow_nhwc(pv, lr.cuda())["enc"]["cond_feat"].dtype == torch.float16
    assert rows["fp16"][0] > rows["bf16"][0] and rows["fp16"][1]["delta"] < rows["bf16"][1]["delta"]
    assert rows["fp16"][3] < rows["bf16"][3] / 4


@pytest.mark.parametrize("regime", ["adversarial", "representative"])
def test_end_to_end_mid_size_both_precisions(regime, capsys):
    check_e2e(regime, 100, 156, 21, capsys)


@pytest.mark.parametrize("regime", ["adversarial", "representative"])
def test_end_to_end_full_size_both_precisions(regime, capsys):
    """BASELINE shape (400x600, N = 16275 tokens): one ~30 s oracle run per regime."""
    check_e2e(regime, 400, 600, 11, capsys)


SCENES = tuple(range(11, 23))       # 12 scenes, incl. the two (13, 15) that missed 0.05 dB in round 3 (0.054 / 0.066 dB)


def test_end_to_end_full_size_twelve_scenes(capsys):
    """BASELINE.json: "codebook indices bit-exact and output PSNR within 0.05 dB of reference" -- the default path at 400x600 on
    TWELVE scenes against the fp32 oracle, its own codebook indices, asserted per scene.  Measured on MI355X
    (profiles/r04_parity_table.txt): index agreement 0.99926-0.99969 (5-12 tokens of 16 275 differ), |dPSNR vs GT| 0.0017-0.0048 dB,
    PSNR(ours, oracle) 57.2-65.0 dB, latent error 1.2-1.4e-5.  Bounds: 2x the measured miss; 0.01 dB (a fifth of the tolerance) for
    the PSNR delta, which sits at the metric's noise floor (the same figure with the oracle's indices forced: 0.0008-0.0047 dB)."""
    rows = []
    for seed in SCENES:
        og, ov, pg, pv, lr, ref = setup("representative", 400, 600, seed)
        with torch.no_grad():
            r = pg.reverse_flow_nhwc(pv, lr.cuda())
        agree = float((r["indices"].cpu() == ref["indices"]).float().mean())
        full = e2e_metrics(r["out"].cpu(), ref["out"], 400)
        lat = float(rel(ops.nhwc_to_nchw(r["latent"]).cpu(), ref["latent"]))
        flips, ratio = audit_flips(r, ref, ov)         # every flipped token is a near-tie (asserted inside)
        rows.append((seed, lat, agree, full["psnr_vs_oracle"], full["delta"], flips, ratio))
        _CACHE.pop(("representative", 400, 600, seed), None)          # 12 full-size reference sets would be ~6 GB of host memory
    with capsys.disabled():
        for row in rows:
            print("\n[e2e 400x600 representative seed %d] latent rel %.2e | index agreement %.5f | PSNR(ours,oracle) %.2f dB | |dPSNR vs GT| %.4f dB"
                  " | %d flipped tokens, all near-ties: worst margin / (2 |dz| |de|) = %.3f" % row, end="")
        print()
    for seed, lat, agree, psnr, delta, flips, ratio in rows:       # one record per quantity: tolerances.py keeps the maximum over the scenes
        within(lat, 2.8e-5)                        # measured max 1.42e-5
        within(1.0 - agree, 1.47e-3)               # measured max 7.4e-4 (12 tokens)
        assert psnr >= 54.2, (seed, psnr)          # measured min 57.22 dB
        within(delta, 0.0088)                      # measured max 0.0066 dB (seed 17; round 5: 0.0048); BASELINE: 0.05


@pytest.mark.parametrize("regime,seed", [("representative", 105), ("representative2", 101), ("representative2", 102), ("representative3", 101)])
def test_end_to_end_full_size_held_out(regime, seed, capsys):
    """Scenes and weights that played no part in choosing the precision scheme: a held-out scene on the usual weights (105: the worst
    of 12 held-out scenes, 12 tokens differ) and two scenes on a SECOND trained-like weight set (another codebook, other ActNorm states
    and filters; PARITY_WEIGHT_SEED=1 in tools/parity_scenes.py).  Measured (profiles/r04_parity_table.txt, second half): agreement
    0.99926 / 0.99988 (2 of 16 275 tokens) / 0.99957, |dPSNR vs GT| 0.0035 / 0.0012 / 0.0001 dB, latent 1.43e-5 / 1.05e-5 /
    0.95e-5.  Same bounds as the twelve-scene test (only the agreement bound is 2x this test's own worst case).
    Round 5: a THIRD weight set (weight seed 2), its worst scene of six: agreement 0.99982 but |dPSNR vs GT| 0.0190 dB (0.0201 with the
    oracle's indices forced: the decoders' 16-bit arithmetic behind the codebook, not the search) -- its own bound, 2x measured, still
    inside BASELINE's 0.05."""
    og, ov, pg, pv, lr, ref = setup(regime, 400, 600, seed)
    with torch.no_grad():
        r = pg.reverse_flow_nhwc(pv, lr.cuda())
    agree = float((r["indices"].cpu() == ref["indices"]).float().mean())
    full = e2e_metrics(r["out"].cpu(), ref["out"], 400)
    lat = float(rel(ops.nhwc_to_nchw(r["latent"]).cpu(), ref["latent"]))
    flips, ratio = audit_flips(r, ref, ov)
    _CACHE.pop((regime, 400, 600, seed), None)
    with capsys.disabled():
        print("\n[e2e 400x600 %s seed %d] latent rel %.2e | index agreement %.5f | PSNR(ours,oracle) %.2f dB | |dPSNR vs GT| %.4f dB"
              " | %d flipped tokens, all near-ties: worst margin / (2 |dz| |de|) = %.3f"
              % (regime, seed, lat, agree, full["psnr_vs_oracle"], full["delta"], flips, ratio))
    within(lat, 2.8e-5)                 # measured max 1.43e-5
    within(1.0 - agree, 1.47e-3)        # measured max 7.4e-4 (12 tokens)
    assert full["psnr_vs_oracle"] >= 54.2, full
    if regime == "representative3":
        within(full["delta"], 0.0125, "set3")   # measured 0.0061 dB (round 5: 0.0190 -- the filters' round-to-nearest; round 6 rounds them with error feedback); BASELINE: 0.05
    else:
        within(full["delta"], 0.0070)           # measured max 0.0030 dB; BASELINE: 0.05


@pytest.mark.parametrize("h,w", [(60, 92), (132, 72), (36, 28)])
def test_end_to_end_other_image_sizes(h, w, capsys):
    """The path is size-generic (infer_dataset_lol.py pads whatever it is given by 20): a small landscape, a portrait and a tiny image --
    ragged 8 x 32 conv tiles, partial attention key tiles, DCN tiles that straddle images -- full path vs the oracle, default precision."""
    og, ov, pg, pv, lr, ref = setup("representative", h, w, 33)
    with torch.no_grad():
        r = pg.reverse_flow_nhwc(pv, lr.cuda())
    agree = float((r["indices"].cpu() == ref["indices"]).float().mean())
    m = e2e_metrics(r["out"].cpu(), ref["out"], h)
    lat = float(rel(ops.nhwc_to_nchw(r["latent"]).cpu(), ref["latent"]))
    flips, ratio = audit_flips(r, ref, ov)
    with capsys.disabled():
        print("\n[e2e %dx%d] latent rel %.2e | index agreement %.5f | PSNR(ours,oracle) %.2f dB | |dPSNR vs GT| %.4f dB | %d flips, worst ratio %.3f"
              % (h, w, lat, agree, m["psnr_vs_oracle"], m["delta"], flips, ratio))
    within(lat, 5.5e-5)                # measured 1.97e-5 / 1.72e-5 / 2.73e-5
    tokens = ref["indices"].numel()
    assert (1.0 - agree) * tokens <= 6.5, (agree, tokens)     # measured: 0 / 3 (of 874) / 0 tokens differ -- two of the three sizes bit-exact
    within(m["delta"], 0.018)          # measured 0.0010 / 0.0008 / 0.0090 dB (round 5: 0.0012 / 0.0022 / 0.0010; the 36 x 28 image is 1 008 pixels: over six such scenes
                                       # the figure is 0.0029 mean / 0.0068 max with error-feedback filters, 0.0050 / 0.0123 with round-to-nearest -- scene noise)


def test_end_to_end_batch_of_8_against_eight_oracle_runs(capsys):
    """BASELINE configs[1] is a batch of 8: the product's ONE batched launch sequence (default launch configuration: keys not split
    at this size, as at 400x600 x 8) against the oracle run image by image, per image (100x156: eight 2 s oracle runs).  VERDICT r03:
    until now "config I8 vs oracle" was an inference from B = 1 comparisons plus the batch-invariance test."""
    og, ov, pg, pv, lr, ref = setup("representative", 100, 156, 21)
    lows = synthetic_pair(8, 100, 156, seed=51)[0]
    lrs = [O.preprocess(im) for im in lows]
    with torch.no_grad():
        r8 = pg.reverse_flow_nhwc(pv, torch.cat(lrs).cuda())
        refs = [og.stages(ov, x) for x in lrs]
    rows = []
    for i, rf in enumerate(refs):
        agree = float((r8["indices"].view(8, -1)[i].cpu() == rf["indices"].view(-1)).float().mean())
        m = e2e_metrics(r8["out"][i:i + 1].cpu(), rf["out"], 100)
        lat = float(rel(ops.nhwc_to_nchw(r8["latent"][i:i + 1]).cpu(), rf["latent"]))
        rows.append((i, lat, agree, m["psnr_vs_oracle"], m["delta"]))
    with capsys.disabled():
        for row in rows:
            print("\n[e2e batch of 8, image %d] latent rel %.2e | index agreement %.5f | PSNR(ours,oracle) %.2f dB | |dPSNR vs GT| %.4f dB" % row, end="")
        print()
    for i, lat, agree, psnr, delta in rows:
        within(lat, 3.7e-5)                # measured 1.5-1.9e-5 (one record per quantity: the maximum over the eight images)
        within(1.0 - agree, 3.0e-3)        # measured: five of the eight images with EVERY index equal, the others 1-2 of 1 320 tokens
        assert psnr >= 49.7, (i, psnr)     # measured 52.7-67.3 dB (ONE flipped token moves a 100x156 image from ~66 to ~53 dB of PSNR(ours, oracle))
        within(delta, 0.0133)              # measured max 0.0066 dB


def test_fp16_batch_of_8_equals_eight_single_runs():
    og, ov, pg, pv, lr, ref = setup("representative", 100, 156, 21)
    lows = synthetic_pair(8, 100, 156, seed=31)[0]
    lr8 = torch.cat([O.preprocess(im) for im in lows]).cuda()
    ops.ATTENTION_KEY_SPLITS_OVERRIDE = 1
    try:
        with torch.no_grad():
            r8 = pg.reverse_flow_nhwc(pv, lr8, precision="fp16")
            for i in (0, 5):
                r1 = pg.reverse_flow_nhwc(pv, lr8[i:i + 1], precision="fp16")
                assert torch.equal(r1["out"][0], r8["out"][i]) and torch.equal(r1["latent"][0], r8["latent"][i])
            again = pg.reverse_flow_nhwc(pv, lr8, precision="fp16")
            assert torch.equal(again["out"], r8["out"]) and torch.equal(again["indices"], r8["indices"])   # launch-to-launch determinism
    finally:
        ops.ATTENTION_KEY_SPLITS_OVERRIDE = None


def test_inference_driver_reruns_fp16_overflows_in_bf16(tmp_path):
    """fp16 (the inference default) has fp16's range.  A checkpoint whose first conv is scaled far past 65504 overflows it (inf ->
    GroupNorm -> NaN); bf16 carries the same activations.  glare_amd.infer must hand back finite PSNRs by re-running those images
    in bf16, and say which ones it re-ran (bf16 misses the end-to-end tolerance: a re-run is flagged per image)."""
    from glare_amd import checkpoint, infer

    netG = seeded_init_(M.VQLLFLOWDeformable().eval(), 0)
    with torch.no_grad():
        netG.RRDB.encoder.conv_in.weight.mul_(3.0e4)
    path = str(tmp_path / "net_G_overflow.pth")
    checkpoint.save_network(netG, path)
    lows = synthetic_lowlight(2, 40, 60, seed=5)
    gts = synthetic_lowlight(2, 40, 60, seed=6)
    psnr = infer.run(2, batch=2, pairs=(lows, gts), net_g=path)
    assert np.isfinite(psnr).all(), psnr
    assert infer.run.bf16_reruns == 2 and infer.run.bf16_rerun_images == [0, 1]     # WHICH images, not only how many
    # round 6: the trigger is the device-side count of non-finite output values before the clamp (every value of both crops here)
    assert infer.run.nonfinite_values == {0: 40 * 60 * 3, 1: 40 * 60 * 3}
    ref = infer.run(2, batch=2, pairs=(lows, gts), net_g=path, precision="bf16")
    assert infer.run.bf16_reruns == 0 and infer.run.bf16_rerun_images == [] and infer.run.nonfinite_values == {}
    assert np.allclose(psnr, ref, atol=0.05)      # single-image reruns split the attention keys differently: rounding-level changes


def test_overflow_flag_catches_what_the_clamp_hides():
    """+inf in the network output is clamped to 1.0 by the post-process and leaves a FINITE PSNR (torch.clamp, infer_dataset_lol.py:138;
    the reference masks NaNs only in its training loss, VQLLFLOWD_model.py:214-217): the device-side counter sees it before the clamp,
    and the driver keys the bf16 re-run on the counter."""
    from glare_amd import harness, infer

    g = torch.Generator().manual_seed(0)
    out = torch.rand(3, 3, 30, 50, generator=g)
    out[1, 0, 3, 25] = float("inf")            # inside image 1's crop [:, :, :24, 20:]
    out[1, 2, 7, 49] = float("-inf")
    out[2, 1, 5, 21] = float("nan")
    out[0, 0, 3, 5] = float("inf")             # in the padding columns: not part of any image
    out[0, 0, 28, 30] = float("inf")           # below the crop
    gts = (torch.rand(3, 24, 30, 3, generator=g) * 255).to(torch.uint8)
    restored, ps, bad = harness.postprocess_device(out.cuda(), 24, 30, gts.cuda(), want_nonfinite=True)
    assert bad.cpu().tolist() == [0, 2, 1]
    ps = ps.cpu().numpy()
    assert np.isfinite(ps[:2]).all() and not np.isfinite(ps[2])          # image 1: a finite PSNR over two clamped infinities
    assert bool(torch.isfinite(restored[1]).all()) and float(restored[1, 7, 29, 2]) == 0.0      # +inf -> 1.0 (x gain), -inf -> 0.0
    plain, _, bad2 = harness.postprocess_device(out.cuda(), 24, 30, want_nonfinite=True)         # no GT: crop + clamp only
    assert float(plain[1, 3, 5, 0]) == 1.0 and bad2.cpu().tolist() == [0, 2, 1]

    class Saturating(torch.nn.Module):         # a network whose fp16 run saturates one output value of image 1; its bf16 run is clean
        def __init__(self, net):
            super().__init__()
            self.net = net

        def reverse_flow_nhwc(self, net_vq, lr, precision=None):
            r = self.net.reverse_flow_nhwc(net_vq, lr, precision=precision)
            if precision != "bf16" and lr.shape[0] > 1:
                r["out"][1, 0, 4, 30] = float("inf")
            return r

    netG = Saturating(seeded_init_(M.VQLLFLOWDeformable().eval(), 0))
    net_vq = seeded_init_(M.VQModel().eval(), 1)
    lows, gts2 = synthetic_lowlight(2, 40, 60, seed=5), synthetic_lowlight(2, 40, 60, seed=6)
    psnr = infer.run(2, batch=2, pairs=(lows, gts2), nets=(netG, net_vq))
    assert np.isfinite(psnr).all()
    assert infer.run.bf16_rerun_images == [1] and infer.run.nonfinite_values == {1: 1}


# ---- hi / lo residual stream (fp16, conditional encoder) ----------------------------------------------------------------
def _pair(t32):
    hi = t32.to(torch.float16)
    lo = (t32 - hi.float()).to(torch.float16)
    return hi, lo


def test_hilo_kernels_keep_22_bits_through_the_residual_add():
    """glare_conv_desc.out_lo / glare_conv1x1_ws_hilo_f16 / glare_groupnorm_hilo_f16 / glare_split_hilo_f32: hi + lo of the output
    equals the fp32 result of conv(x) + bias + (res_hi + res_lo) to ~2^-20 (two 11-bit halves), hi alone is its fp16 rounding, and
    the fused GroupNorm statistics / the hi-lo GroupNorm are those of the 22-bit value."""
    g = torch.Generator().manual_seed(21)
    B, H, W, C = 2, 11, 37, 128
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    r32 = torch.randn(B, C, H, W, generator=g) * 3.0
    ref = F.conv2d(h16(x).cuda(), h16(w).cuda(), b.cuda(), 1, 1) + r32.cuda()
    with ops.use_precision("fp16"):
        r_hi, r_lo = _pair(r32.permute(0, 2, 3, 1).contiguous().cuda())
        assert float(((r_hi.float() + r_lo.float()) - r32.permute(0, 2, 3, 1).cuda()).abs().max()) < 2e-6 * 16
        r_hi._lo = r_lo
        pc = ops.PackedConv(w.cuda(), b.cuda())
        out = ops.conv2d(nhwc16(x), pc, residual=r_hi, gn_stats=True, hilo=True)
        v = out.float() + out._lo.float()
        refn = ref.permute(0, 2, 3, 1)
        assert float((v - refn).abs().max() / refn.abs().max()) < 4e-6, float((v - refn).abs().max() / refn.abs().max())
        assert torch.equal(out, refn.to(torch.float16)) or float((out.float() - refn).abs().max() / refn.abs().max()) < 6e-4
        plain = ops.conv2d(nhwc16(x), pc, residual=r_hi)          # the plain epilogue: 16-bit roundings before and after the add
        assert float((plain.float() - refn).abs().max()) > 20 * float((v - refn).abs().max())
        # GroupNorm of the pair from the fused statistics == group_norm of the 22-bit value
        gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
        y = ops.groupnorm(out, gam, bet, swish=True)
        yr = F.silu(F.group_norm(v.permute(0, 3, 1, 2), 32, gam, bet, eps=1e-6)).permute(0, 2, 3, 1)
        assert rel(y, yr) < 4e-4, rel(y, yr)
        # ... and without statistics on the tensor (split_hilo output: the standalone statistics pass reads hi + lo)
        s = ops.split_hilo(refn.contiguous())
        assert float(((s.float() + s._lo.float()) - refn).abs().max() / refn.abs().max()) < 4e-6
        y2 = ops.groupnorm(s, gam, bet, swish=True)
        assert rel(y2, yr) < 4e-4
        # stride-2 conv (Downsample) opens a pair as well
        d = ops.conv2d(nhwc16(x), pc, stride=2, hilo=True)
        dref = F.conv2d(F.pad(h16(x).cuda(), (0, 1, 0, 1)), h16(w).cuda(), b.cuda(), 2, 0).permute(0, 2, 3, 1)
        assert float(((d.float() + d._lo.float()) - dref).abs().max() / dref.abs().max()) < 4e-6
        # 1x1: the shared-filter and the per-image entry points
        C2 = 512
        x1 = torch.randn(B, C2, 9, 21, generator=g)
        w1 = torch.randn(C2, C2, 1, 1, generator=g) / C2 ** 0.5
        b1 = torch.randn(C2, generator=g) * 0.1
        r1 = torch.randn(B, 9, 21, C2, generator=g).cuda() * 2.0
        rh, rl = _pair(r1)
        rh._lo = rl
        pc1 = ops.PackedConv(w1.cuda(), b1.cuda())
        o1 = ops.conv2d(nhwc16(x1), pc1, residual=rh, gn_stats=True, hilo=True)
        ref1 = F.conv2d(h16(x1).cuda(), h16(w1).cuda(), b1.cuda()).permute(0, 2, 3, 1) + r1
        assert float(((o1.float() + o1._lo.float()) - ref1).abs().max() / ref1.abs().max()) < 4e-6
        st = o1._gn_stats.double().sum(dim=1)                                        # [B, 32, 2]: (sum, sum of squares) per group
        vv = (o1.float() + o1._lo.float()).double().reshape(B, -1, 32, C2 // 32)
        assert torch.allclose(st[..., 0], vv.sum(dim=(1, 3)), rtol=1e-4, atol=1e-2)
        assert torch.allclose(st[..., 1], (vv * vv).sum(dim=(1, 3)), rtol=1e-4)
        wb = torch.stack([w1[:, :, 0, 0], w1[:, :, 0, 0] * 0.5]).to(torch.float16).cuda().contiguous()
        bb = torch.stack([b1, b1 * 2]).cuda().contiguous()
        o2 = ops.conv1x1_per_image(nhwc16(x1), wb, bb, residual=rh, hilo=True)
        ref2 = torch.einsum("bhwc,bdc->bhwd", nhwc16(x1).float(), wb.float()) + bb[:, None, None, :] + r1
        assert float(((o2.float() + o2._lo.float()) - ref2).abs().max() / ref2.abs().max()) < 4e-6


def test_precision_ladder_of_the_conditional_encoder():
    """Stage A on one 100x156 image, fp16, three forms of the conditional encoder against the fp32 oracle: (a) 16-bit residual stream
    (round 2), (b) the stream as hi / lo pairs, single-pass convs (round 3), (c) fp32-class convs on hi / lo pairs (round 4, the
    default).  color_map -- what the flow and then the codebook search see -- improves at every step, (c) by more than 20x."""
    from glare_amd.modules import encoder_decoder as ED

    og, ov, pg, pv, lr, ref = setup("representative", 100, 156, 21)
    errs = {}
    with torch.no_grad(), ops.use_precision("fp16"):
        for name, stream, f32c in (("16-bit stream", False, False), ("hi/lo stream", True, False), ("fp32-class", True, True)):
            pg.RRDB.encoder.hilo_stream = stream
            keep, ED.FP32_CLASS = ED.FP32_CLASS, f32c
            try:
                enc = pg.RRDB.forward_nhwc(lr.cuda())
            finally:
                pg.RRDB.encoder.hilo_stream = True
                ED.FP32_CLASS = keep
            errs[name] = rel(ops.nhwc_to_nchw(enc["color_map"]).cpu(), ref["enc"]["color_map"])
            assert enc["mid_feat"][0].dtype == torch.float16
            assert (getattr(enc["cond_feat"], "_lo", None) is not None) == f32c
    print("\n[precision ladder] color_map rel err: %s" % errs)
    assert errs["hi/lo stream"] < 0.85 * errs["16-bit stream"]
    assert errs["fp32-class"] < errs["hi/lo stream"] / 20
    within(errs["fp32-class"], 3.7e-5)          # measured 1.83e-5 (hi/lo stream 1.77e-3, 16-bit stream 2.70e-3)
