"""GPU: the HIP-backed module graph against the CPU oracle with identical (name-seeded) weights.

Stage-isolated checks feed each product stage the ORACLE's inputs, so one stage's bf16 rounding
cannot hide in another's; the end-to-end tests then bound the accumulated difference directly:
PSNR(ours, oracle), codebook-index agreement, and the PSNR delta against a ground truth CORRELATED
with the output (oracle output + noise at 27 dB) -- the form of BASELINE's "output PSNR within
0.05 dB of the reference" that can fail.

Tolerances (TOL below) are <= 2x what was measured on MI355X (tools/parity_probe.py, DESIGN.md
section 4): activations are bf16 between kernels (relative rounding 2^-9 per store), accumulation
fp32; codebook indices are bit-exact given identical latent input."""
import os

import numpy as np
import pytest
import torch

from glare_amd import modules as M
from glare_amd import ops
from glare_amd.synthetic import seeded_init_, synthetic_gt, synthetic_lowlight
from oracle import torch_ref as O

pytestmark = pytest.mark.gpu

from tolerances import TOL, within  # noqa: E402  (tests/tolerances.py: per-stage bounds, <= 2x measured)


@pytest.fixture(autouse=True)
def bf16_precision():
    """This module is the bf16 suite: every tolerance below was measured with bf16 activations and filters.  The inference entry
    points default to fp16 (ops.inference_precision); tests/test_gpu_precision.py is the fp16 suite and holds the end-to-end
    table of both precisions in both weight regimes."""
    with ops.use_precision("bf16"):
        yield


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def nhwc(x, bf16=True):
    return ops.nchw_to_nhwc(x.cuda(), bf16=bf16)


def nchw(x):
    return ops.nhwc_to_nchw(x).cpu()


@pytest.fixture(scope="module")
def nets():
    torch.manual_seed(0)
    og = seeded_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), 0)
    ov = seeded_init_(O.VQModel().eval(), 1)
    pg = M.VQLLFLOWDeformable().eval()
    pv = M.VQModel().eval()
    pg.load_state_dict(og.state_dict(), strict=True)
    pv.load_state_dict(ov.state_dict(), strict=True)
    pg.cuda()
    pv.cuda()
    imgs = synthetic_lowlight(2, 20, 36, seed=7)  # -> 2 x 3 x 40 x 56 after the 20-px reflect pad
    lr = torch.cat([O.preprocess(im) for im in imgs])
    with torch.no_grad():
        ref = og.stages(ov, lr)
    return og, ov, pg, pv, lr, ref


def test_stage_a_conditional_encoder(nets):
    og, ov, pg, pv, lr, ref = nets
    enc = pg.RRDB.forward_nhwc(lr.cuda())
    within(rel(nchw(enc["cond_feat"]), ref["enc"]["cond_feat"]), TOL["cond_feat"])
    within(rel(nchw(enc["color_map"]), ref["enc"]["color_map"]), TOL["color_map"])
    for i, (a, b) in enumerate(zip(enc["mid_feat"], ref["enc"]["mid_feat"])):
        within(rel(nchw(a), b), TOL["mid_feat%d" % i])


def test_stage_b_flow_reverse(nets):
    og, ov, pg, pv, lr, ref = nets
    z = pg.flowUpsamplerNet.decode_nhwc(nhwc(ref["enc"]["color_map"], bf16=False), nhwc(ref["enc"]["cond_feat"]))
    within(rel(nchw(z), ref["latent"]), TOL["latent"])
    # reference-surface entry point (NCHW tensors, rrdbResults dict, reverse=True)
    x, _ = pg.flowUpsamplerNet(rrdbResults={k: v.cuda() for k, v in ref["enc"].items() if k != "mid_feat"},
                               z=ref["enc"]["color_map"].cuda(), eps_std=0, reverse=True)
    within(rel(x.cpu(), ref["latent"]), TOL["latent"])


def test_stage_c_codebook_indices_bit_exact(nets):
    og, ov, pg, pv, lr, ref = nets
    zq, loss, (_, _, idx) = pv.quantize(ref["latent"].cuda())
    assert torch.equal(idx.cpu(), ref["indices"])
    with torch.no_grad():
        rq, rloss, _ = ov.quantize(ref["latent"])
    assert torch.equal(zq.cpu(), rq)
    assert abs(float(loss) - float(rloss)) <= 1e-5 * abs(float(rloss))  # same formula, different reduction order


def test_stage_d_vq_decoder(nets):
    og, ov, pg, pv, lr, ref = nets
    idx, img, feats = pv.decode_nhwc(nhwc(ref["latent"], bf16=False), want_image=True)
    assert torch.equal(idx.cpu(), ref["indices"])
    for i, (a, b) in enumerate(zip(feats, ref["code_feats"])):
        within(rel(nchw(a), b), TOL["code_feat%d" % i])
    within(rel(img.cpu(), ref["vq_rec"]), TOL["vq_rec"])


def test_stage_e_aft_decoder(nets):
    og, ov, pg, pv, lr, ref = nets
    out = pg.deformable_decoder.forward_nhwc(nhwc(ref["latent"], bf16=False), [nhwc(f) for f in ref["code_feats"]],
                                             [nhwc(f) for f in ref["enc"]["mid_feat"]])
    within(rel(out.cpu(), ref["out"]), TOL["aft_out"])


def correlated_gt(ref_img, db=27.0, seed=5):
    """A ground truth correlated with the output: the oracle's post-processed image + Gaussian noise at `db` dB PSNR (LOL-trained
    GLARE scores ~27 dB against its real ground truth).  Against an independent random GT (what round 1 used) every output scores
    ~5.4 dB and a 26 dB disagreement moves that by 0.003 dB: that check could not fail."""
    rng = np.random.default_rng(seed)
    gt = np.clip(ref_img + rng.normal(0, 10 ** (-db / 20), ref_img.shape), 0, 1)
    return np.round(gt * 255).astype(np.uint8)


def e2e_metrics(out, out_ref, h):
    """PSNR(ours, oracle) on the harness's post-processed images and |PSNR(ours, GT) - PSNR(oracle, GT)| with the GT-mean gain."""
    a, b = O.postprocess(out, h), O.postprocess(out_ref, h)
    gt = correlated_gt(b)
    pa, pb = O.psnr(gt / 255, O.postprocess(out, h, gt)), O.psnr(gt / 255, O.postprocess(out_ref, h, gt))
    return {"psnr_vs_oracle": float(O.psnr(a, b)), "psnr_gt_ours": float(pa), "psnr_gt_oracle": float(pb), "delta": float(abs(pa - pb))}


def check_end_to_end(pg, pv, lr1, ref, h, tag, bounds, capsys):
    """The accumulated difference of the whole path on one image, asserted directly.
    (1) full path: index agreement and PSNR(ours, oracle) -- with name-seeded random weights the flow's 48 divisions by
        sigmoid(.)+1e-4 put the latent ~90 codebook radii away from the codes, where the nearest code is decided by relative
        margins of 1e-4: 1.5 % of the tokens flip under the path's 1.3 % latent error, and each flipped token redraws a 4x4-pixel
        patch (DESIGN.md section 4 has the budget);
    (2) the same run with the VQ decoder fed the oracle's indices (everything else ours): the north_star tolerance proper,
        |PSNR(ours, GT) - PSNR(oracle, GT)| <= 0.05 dB, holds whenever the indices agree."""
    r = pg.reverse_flow_nhwc(pv, lr1.cuda())
    agree = float((r["indices"].cpu() == ref["indices"]).float().mean())
    full = e2e_metrics(r["out"].cpu(), ref["out"], h)
    _, _, feats_i = pv.decode_nhwc(nhwc(ref["latent"], bf16=False), want_image=False)      # indices == the oracle's (stage C test)
    out_i = pg.deformable_decoder.forward_nhwc(r["latent"], feats_i, r["enc"]["mid_feat"]).cpu()
    forced = e2e_metrics(out_i, ref["out"], h)
    with capsys.disabled():
        print("\n[e2e %s] index agreement %.5f | full path: PSNR(ours,oracle) %.2f dB, PSNR vs GT %.3f / %.3f (delta %.4f) | "
              "oracle's indices: PSNR(ours,oracle) %.2f dB, delta %.4f dB"
              % (tag, agree, full["psnr_vs_oracle"], full["psnr_gt_ours"], full["psnr_gt_oracle"], full["delta"],
                 forced["psnr_vs_oracle"], forced["delta"]))
    assert torch.isfinite(r["out"]).all()
    assert agree >= bounds["agree"]
    assert full["psnr_vs_oracle"] >= bounds["psnr_full"]
    assert forced["psnr_vs_oracle"] >= bounds["psnr_forced"]
    assert forced["delta"] <= 0.05              # BASELINE.json: output PSNR within 0.05 dB of the reference
    return r


def test_end_to_end_against_oracle_mid_size(capsys):
    """100 x 156 image (latent 30 x 44 = 1320 tokens): big enough for stable statistics, small enough for a 2 s oracle run."""
    og = seeded_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), 0)
    ov = seeded_init_(O.VQModel().eval(), 1)
    pg, pv = M.VQLLFLOWDeformable().eval(), M.VQModel().eval()
    pg.load_state_dict(og.state_dict(), strict=True)
    pv.load_state_dict(ov.state_dict(), strict=True)
    pg.cuda()
    pv.cuda()
    h, w = 100, 156
    lr = O.preprocess(synthetic_lowlight(1, h, w, seed=21)[0])
    with torch.no_grad():
        ref = og.stages(ov, lr)
        # measured on MI355X: agreement 0.985, full 30-33 dB, forced 45-46 dB -> bounds = 2x the disagreement / -3 dB (2x the MSE)
        check_end_to_end(pg, pv, lr, ref, h, "100x156", {"agree": 0.970, "psnr_full": 27.0, "psnr_forced": 42.0}, capsys)


def test_reference_module_surface(nets):
    """forward() of the mirrored modules takes and returns the reference's NCHW fp32 tensors."""
    og, ov, pg, pv, lr, ref = nets
    rb_o, rb_p = og.RRDB.encoder.down[1].block[0], pg.RRDB.encoder.down[1].block[0]
    x = torch.randn(1, 128, 12, 20)
    with torch.no_grad():
        within(rel(rb_p(x.cuda(), None), rb_o(x)), 7.4e-3)   # measured 3.86e-03
        at_o, at_p = og.RRDB.encoder.mid.attn_1, pg.RRDB.encoder.mid.attn_1
        x5 = torch.randn(1, 512, 9, 13)
        within(rel(at_p(x5.cuda()), at_o(x5)), 4.5e-3)   # measured 2.35e-03
        out_p, lat_p = pg(net_vq=pv, lr=lr.cuda(), z=None, eps_std=0, reverse=True, reverse_with_grad=False)
    assert out_p.shape == ref["out"].shape and lat_p.shape == ref["latent"].shape
    with pytest.raises(NotImplementedError):
        pg.RRDB(lr)  # CPU tensor: no fallback


def test_row_a7_vqgan_encode(nets):
    """Row a7: VQModel.encode (Encoder + quant_conv, VQModel_arch.py:74-79) -- the frozen ground-truth encoder of stage 2 --
    through the reference-shaped entry point and the NHWC one."""
    og, ov, pg, pv, lr, ref = nets
    g = torch.Generator().manual_seed(17)
    gt = torch.rand(2, 3, 40, 56, generator=g)
    with torch.no_grad():
        z_o, _ = ov.encode(gt)
        z_p, none = pv.encode(gt.cuda())
        z_n = pv.encode_nhwc(gt.cuda())
    assert none is None and z_p.shape == z_o.shape
    within(rel(z_p.cpu(), z_o), 3e-2)
    assert torch.equal(nchw(z_n).cpu(), z_p.cpu())


def test_stage2_normal_flow_and_nll(nets):
    """Row a4: FlowUpsamplerNet.encode + log-determinant + Gaussian NLL against the oracle's normal_flow
    (which is bit-identical to the reference's, tests/test_oracle_vs_reference.py)."""
    og, ov, pg, pv, lr, ref = nets
    o2 = seeded_init_(O.LLFlowVQGAN2().eval(), 2)
    p2 = M.LLFlowVQGAN2().eval()
    p2.load_state_dict(o2.state_dict(), strict=True)
    p2.cuda()
    g = torch.Generator().manual_seed(3)
    gt = torch.randn(2, 3, 10, 14, generator=g) * 0.5
    with torch.no_grad():
        z_o, nll_o, ld_o = o2.normal_flow(gt, lr)
        # flow only, on the oracle's conditional features (isolates the flow kernels)
        enc = o2.RRDB(lr)
        z_p, ld_p, lp_p = p2.flowUpsamplerNet.encode_nhwc(nhwc(gt, bf16=False), nhwc(enc["cond_feat"]),
                                                           mean=nhwc(enc["color_map"], bf16=False))
        within(rel(nchw(z_p), z_o), 2.0e-3)   # measured 1.05e-03
        within(float((ld_p.float().cpu() - ld_o).abs().max() / ld_o.abs().max()), 1.6e-5, tag="logdet")   # measured 8.0e-06
        # whole stage-2 forward through the reference-shaped entry point
        z2, nll_p, _ = p2(gt=gt.cuda(), lr=lr.cuda(), reverse=False)
    within(rel(z2.cpu(), z_o), 2.2e-3)   # measured 1.13e-03
    within(float((nll_p.float().cpu() - nll_o).abs().max() / nll_o.abs().max()), 4.7e-5, tag="nll")   # measured 2.35e-05
    # invertibility on the HIP path itself: decode(encode(x)) == x -- with the four-launch reverse step, whose coupling nets are the SAME
    # kernels in the SAME 16-bit arithmetic as encode's (the roundings cancel exactly).  Round 6's fused reverse step
    # (csrc/flow_fused.hip) keeps h1 / h2 as hi / lo pairs: against this bf16 encode it differs by the bf16 rounding of h1 / h2, which
    # the reverse pass of an UNTRAINED flow amplifies ~100x (name-seeded weights; synthetic.representative_init_ documents the
    # instability -- the oracle's own fp32 reverse of the same latent is 9.9e-2 away from BOTH forms here).  The fused step's parity is
    # held where it is well-conditioned: tests/test_gpu_flow_fused.py (3.8e-7 of the four-launch form, 3.3e-6 of the oracle).
    import importlib
    FU = importlib.import_module("glare_amd.modules.FlowUpsamplerNet")
    ft = nhwc(enc["cond_feat"])
    FU.FUSED_STEP = False
    try:
        p2.flowUpsamplerNet.invalidate()
        back = p2.flowUpsamplerNet.decode_nhwc(z_p, ft)
    finally:
        FU.FUSED_STEP = True
        p2.flowUpsamplerNet.invalidate()
    within(rel(nchw(back), gt), 1.5e-3)   # measured 7.58e-04
    with torch.no_grad():
        back_f = p2.flowUpsamplerNet.decode_nhwc(z_p, ft)
    within(rel(nchw(back_f), gt), 0.15, tag="fused")   # measured 7.5e-02: the amplified bf16 rounding of encode's h1 / h2, see above


def test_inference_driver_matches_oracle_psnr():
    """glare_amd.infer (device harness + sharding + batching + fused graph) on 3 images, batch 2, against
    (a) the same network run one image at a time through the ORACLE's harness functions (numpy pad / log / crop / gain / PSNR):
        isolates the driver -- per-image PSNR equal to 1e-3 dB;
    (b) the oracle network run one image at a time (the reference's B = 1 loop), with a ground truth correlated with the
        output (oracle output + 27 dB noise): the accumulated model difference, bounded at 2x what was measured."""
    from glare_amd import harness, infer

    h, w = 60, 92
    lows = synthetic_lowlight(3, h, w, seed=77)
    og = seeded_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), 0)
    ov = seeded_init_(O.VQModel().eval(), 1)
    outs = []
    with torch.no_grad():
        for i in range(3):
            outs.append(og(ov, O.preprocess(lows[i]))[0])
    gts = np.stack([correlated_gt(O.postprocess(o, h), seed=50 + i) for i, o in enumerate(outs)])
    pg = seeded_init_(M.VQLLFLOWDeformable().eval(), 0).cuda()
    pv = seeded_init_(M.VQModel().eval(), 1).cuda()
    # one attention configuration for the batched and the single-image runs (a different key split is a different summation
    # order: a handful of near-tie tokens would flip and move the per-image PSNR by ~1 dB -- see check_end_to_end)
    ops.ATTENTION_KEY_SPLITS_OVERRIDE = 1
    try:
        psnrs = infer.run(3, batch=2, pairs=(lows, gts))
        mines = []
        for i in range(3):
            with torch.no_grad():     # the driver's own device pre-processing: its log() differs from numpy's in the last bit,
                lr_i = harness.preprocess_device(torch.from_numpy(np.ascontiguousarray(lows[i:i + 1])).cuda())   # which flips tokens
                mines.append(pg.reverse_flow_nhwc(pv, lr_i)["out"].cpu())
    finally:
        ops.ATTENTION_KEY_SPLITS_OVERRIDE = None
    for i in range(3):
        mine = mines[i]
        direct = O.psnr(gts[i] / 255, O.postprocess(mine, h, gts[i]))
        assert abs(psnrs[i] - direct) <= 1e-3, (i, psnrs[i], direct)                       # (a) the driver adds nothing
        ref = O.psnr(gts[i] / 255, O.postprocess(outs[i], h, gts[i]))
        print("image %d: driver %.3f dB, oracle %.3f dB" % (i, psnrs[i], ref))
        assert abs(psnrs[i] - ref) <= 4.0, (i, psnrs[i], ref)                              # (b) measured 1.0-1.8 dB (token flips)


@pytest.fixture(scope="module")
def full_size():
    """BASELINE configs[1]: a batch of 8 different 400 x 600 images (420 x 620 padded) through the fused graph, plus ONE of them
    through the CPU oracle (~20-60 s) -- one oracle image is enough because test_batch_of_8_equals_eight_single_runs proves the
    batched run is eight independent single-image runs, bit for bit."""
    torch.manual_seed(0)
    og = seeded_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), 0)
    ov = seeded_init_(O.VQModel().eval(), 1)
    pg, pv = M.VQLLFLOWDeformable().eval(), M.VQModel().eval()
    pg.load_state_dict(og.state_dict(), strict=True)
    pv.load_state_dict(ov.state_dict(), strict=True)
    pg.cuda()
    pv.cuda()
    imgs = synthetic_lowlight(8, 400, 600, seed=11)
    lr8 = torch.cat([O.preprocess(im) for im in imgs])
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    try:
        with torch.no_grad():
            ref = og.stages(ov, lr8[:1])
    finally:
        torch.set_num_threads(threads)
    return og, ov, pg, pv, lr8, ref


def test_full_size_stage_parity_against_oracle(full_size):
    """Every HIP stage on the ORACLE's inputs for that stage at the BASELINE shape: the kernels see their production launch shapes
    (attention N = 16275, full-resolution convs and DCN warps, 254+ workgroups in flight), where a timing-dependent fault would
    show and the small fixtures cannot."""
    og, ov, pg, pv, lr8, ref = full_size
    lr = lr8[:1]
    with torch.no_grad():
        enc = pg.RRDB.forward_nhwc(lr.cuda())
        within(rel(nchw(enc["cond_feat"]), ref["enc"]["cond_feat"]), TOL["cond_feat"])
        within(rel(nchw(enc["color_map"]), ref["enc"]["color_map"]), TOL["color_map"])
        for i, (a, b) in enumerate(zip(enc["mid_feat"], ref["enc"]["mid_feat"])):
            within(rel(nchw(a), b), TOL["mid_feat%d" % i])
        z = pg.flowUpsamplerNet.decode_nhwc(nhwc(ref["enc"]["color_map"], bf16=False), nhwc(ref["enc"]["cond_feat"]))
        within(rel(nchw(z), ref["latent"]), TOL["latent"])
        idx, img, feats = pv.decode_nhwc(nhwc(ref["latent"], bf16=False), want_image=True)
        assert torch.equal(idx.cpu(), ref["indices"])
        for i, (a, b) in enumerate(zip(feats, ref["code_feats"])):
            within(rel(nchw(a), b), TOL["code_feat%d" % i])
        within(rel(img.cpu(), ref["vq_rec"]), TOL["vq_rec"])
        out = pg.deformable_decoder.forward_nhwc(nhwc(ref["latent"], bf16=False), [nhwc(f) for f in ref["code_feats"]],
                                                 [nhwc(f) for f in ref["enc"]["mid_feat"]])
        within(rel(out.cpu(), ref["out"]), TOL["aft_out"])


def test_full_size_end_to_end_against_oracle(full_size, capsys):
    """The whole path at 400 x 600 against the oracle: measured index agreement 0.98495, PSNR(ours, oracle) 30.4 dB (46.2 dB and a
    0.001 dB PSNR delta with the oracle's indices)."""
    og, ov, pg, pv, lr8, ref = full_size
    with torch.no_grad():
        check_end_to_end(pg, pv, lr8[:1], ref, 400, "400x600", {"agree": 0.970, "psnr_full": 27.4, "psnr_forced": 43.2}, capsys)


def test_batch_of_8_equals_eight_single_runs(full_size):
    """BASELINE configs[1] is B = 8; the reference only ever runs B = 1 (infer_dataset_lol.py:113-135).  GroupNorm, attention, the
    flow, the codebook search and DCN are per sample and the mean rescale is per sample in inference, so a batch of
    8 must equal eight single-image runs BIT FOR BIT -- output, latent and indices -- given the same kernel configuration (a B = 1
    attention launch would otherwise split the keys over 4 workgroups: different summation order; pinned to 1 here, and the
    split path is compared to tolerance below)."""
    og, ov, pg, pv, lr8, ref = full_size
    with torch.no_grad():
        r8 = pg.reverse_flow_nhwc(pv, lr8.cuda())
        ops.ATTENTION_KEY_SPLITS_OVERRIDE = 1
        try:
            for i in (0, 3, 7):
                r1 = pg.reverse_flow_nhwc(pv, lr8[i:i + 1].cuda())
                assert torch.equal(r1["out"][0], r8["out"][i]), i
                assert torch.equal(r1["latent"][0], r8["latent"][i]), i
                n = r1["indices"].numel()
                assert torch.equal(r1["indices"], r8["indices"][i * n:(i + 1) * n]), i
        finally:
            ops.ATTENTION_KEY_SPLITS_OVERRIDE = None
        # the default B = 1 configuration (keys split 4 ways) agrees to rounding: same tokens except near-ties
        r1s = pg.reverse_flow_nhwc(pv, lr8[:1].cuda())
        n = r1s["indices"].numel()
        assert float((r1s["indices"] == r8["indices"][:n]).float().mean()) > 0.98     # measured 0.9899: summation order flips near-ties
        within(rel(r1s["enc"]["cond_feat"], r8["enc"]["cond_feat"][:1]), 2e-3)
    # and image 0 of the batch is the image the oracle comparison above was made on
    assert torch.equal(lr8[0], lr8[:1][0])


def test_full_size_attention_and_dcn_properties():
    """BASELINE shapes (N = 105*155 tokens; 420x620 warp): size-independent properties.
    attention: V = const  =>  out = const (softmax rows sum to 1) and permuting the keys does not change the
    output; DCN: zero offsets / unit mask => equals the MFMA conv of the same weights."""
    g = torch.Generator().manual_seed(0)
    N, C = 105 * 155, 512
    qk = (torch.randn(1, N, 2 * C, generator=g) * 0.2).to(torch.bfloat16).cuda()
    npad = (N + 63) // 64 * 64
    vt = torch.zeros(1, C, npad, dtype=torch.bfloat16, device="cuda")
    vt[:, :, :N] = 0.75
    out = ops.attention_d512(qk, qk[..., C:], vt, N, ldq=2 * C, ldk=2 * C)
    assert torch.allclose(out.float(), torch.full_like(out.float(), 0.75), atol=2e-2)
    v = torch.randn(1, N, C, generator=g).to(torch.bfloat16).cuda()
    vt[:, :, :N] = v.transpose(1, 2)
    ref = ops.attention_d512(qk, qk[..., C:], vt, N, ldq=2 * C, ldk=2 * C)
    perm = torch.randperm(N, generator=g).cuda()
    kp = qk[:, perm, C:].contiguous()
    vt2 = torch.zeros_like(vt)
    vt2[:, :, :N] = v[:, perm].transpose(1, 2)
    got = ops.attention_d512(qk, kp, vt2, N, ldq=2 * C, ldk=C)
    within(rel(got, ref), 6.1e-3)   # measured 3.17e-03
    # DCN at the full-resolution warp shape, B = 1
    x = torch.randn(1, 420, 620, 128, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(128, 128, 3, 3, generator=g) * 0.03).cuda()
    b = (torch.randn(128, generator=g) * 0.1).cuda()
    plane = (420 * 620 + 63) // 64 * 64
    om = torch.zeros(1, 108, plane, device="cuda")
    om[:, 72:] = 30.0  # mask logits -> sigmoid = 1
    got = ops.mdcn_forward_nhwc(x, om, ops.PackedDcn(w, b, 4))
    ref = ops.conv2d(x, ops.PackedConv(w, b), out_mode=ops.OUT_NHWC_F32)
    within(rel(got, ref), 3.2e-3)   # measured 1.66e-03
