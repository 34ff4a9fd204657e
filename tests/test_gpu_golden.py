"""GPU: the reference-generated fixtures (tests/golden/*.npz, captured by tests/golden/make_golden.py from the imported reference)
fed to the HIP path DIRECTLY -- no oracle in between.  vq.npz, harness.npz, msssim.npz and stage2_grads*.npz are consumed by
test_gpu_vq.py / test_gpu_harness.py / test_gpu_train.py; this file covers flow.npz, blocks.npz and graph.npz.
Tolerances are the bf16-activation bounds of tests/test_gpu_graph.py (<= 2x measured)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from glare_amd import modules as M
from glare_amd import ops
from glare_amd.modules import encoder_decoder as ED
from glare_amd.modules.FlowUpsamplerNet import FlowStep, FlowUpsamplerNet
from glare_amd.synthetic import seeded_init_
from tolerances import TOL, TOL_SLICE, within

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _load(module, npz, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(npz[k]) for k in npz.files if k.startswith(prefix)}
    module.load_state_dict(sd, strict=True)
    return module.eval().cuda()


def nhwc(x, bf16=True):
    return ops.nchw_to_nhwc(torch.as_tensor(x).cuda(), bf16=bf16)


def test_flow_steps_reference_vectors(golden):
    """flow.npz: two consecutive reference FlowSteps (FlowStep.py:75-119; one coupling-free, one CondAffineSeparatedAndCond) in
    both directions with their log-determinants.  The HIP flow is not per-step (the affine parts are pre-composed in fp64 and the
    conditional halves batched): the two steps are installed as the `layers` of a FlowUpsamplerNet and its decode / encode run."""
    g = golden("flow")
    net = FlowUpsamplerNet((80, 80, 3), 64, 12)
    net.layers = nn.ModuleList([_load(FlowStep(3, flow_coupling="noCoupling"), g, "s0."),
                                _load(FlowStep(3, flow_coupling="CondAffineSeparatedAndCond"), g, "s1.")])
    net.cuda()
    z, ft = nhwc(g["z"], bf16=False), nhwc(g["ft"])
    with torch.no_grad():
        rev = net.decode_nhwc(z, ft)                    # reverse order: s1 then s0
        fwd, logdet, _ = net.encode_nhwc(z, ft)         # normal order: s0 then s1
    e_rev, e_fwd = rel(ops.nhwc_to_nchw(rev), g["rev"]), rel(ops.nhwc_to_nchw(fwd), g["fwd"])
    print("flow.npz: reverse rel %.2e, forward rel %.2e, logdet %s vs %s" % (e_rev, e_fwd, logdet.tolist(), g["fwd_logdet"].tolist()))
    assert e_rev < 2e-3 and e_fwd < 2e-3                # measured ~5e-4: ft and the coupling nets' activations are bf16
    assert np.allclose(logdet.cpu().numpy(), g["fwd_logdet"], rtol=2e-3, atol=2e-2)
    with torch.no_grad():                               # invertibility on the HIP path itself
        back, _, _ = net.encode_nhwc(rev, ft)
    # round 6: decode runs the fused fp32-class step (csrc/flow_fused.hip), encode the 16-bit four-launch form -- the round trip now
    # measures the 16-bit form's rounding of h1 / h2 (bf16 here), not the cancellation of two bit-identical evaluations
    within(rel(ops.nhwc_to_nchw(back), g["z"]), 2.2e-4)   # measured 1.05e-04
    import importlib
    FU = importlib.import_module("glare_amd.modules.FlowUpsamplerNet")
    FU.FUSED_STEP = False                               # the four-launch form in both directions: identical nets, exact cancellation
    try:
        net.invalidate()
        with torch.no_grad():
            rev4 = net.decode_nhwc(z, ft)
            back4, _, _ = net.encode_nhwc(rev4, ft)
    finally:
        FU.FUSED_STEP = True
        net.invalidate()
    within(rel(ops.nhwc_to_nchw(back4), g["z"]), 1.5e-7)   # measured 7.75e-08
    within(rel(rev4, rev), 1e-3)                           # the two forms of the reverse step against each other (bf16 h1 / h2 in one)


def test_blocks_reference_vectors(golden):
    """blocks.npz: the reference's ResnetBlock(32 -> 64), AttnBlock(64), Downsample(32), Upsample(32) (encoder_decoder.py:38-192)
    outputs.  AttnBlock(64) is not the GLARE head size: it runs the module's general (materialised) form; the 512-channel
    block is additionally pinned through graph.npz below and through test_gpu_kernels.py against fp32 softmax."""
    g = golden("blocks")
    x32, x64 = torch.from_numpy(g["x32"]).cuda(), torch.from_numpy(g["x64"]).cuda()
    with torch.no_grad():
        res = _load(ED.ResnetBlock(in_channels=32, out_channels=64), g, "res.")(x32)
        attn = _load(ED.AttnBlock(64), g, "attn.")(x64)
        dn = _load(ED.Downsample(32), g, "down.")(x32)
        up = _load(ED.Upsample(32), g, "up.")(x32)
    errs = {k: rel(v, g[k]) for k, v in (("res", res), ("attn", attn), ("down", dn), ("up", up))}
    print("blocks.npz:", errs)
    assert errs["res"] < 1.2e-2 and errs["down"] < 8e-3 and errs["up"] < 8e-3   # bf16 input + output rounding, fp32 accumulate
    assert errs["attn"] < 1.2e-2, errs      # x + proj_out(softmax(q k^T) v) with 16-bit q / k / v / P: the residual x dominates the norm


def test_graph_reference_vectors(golden):
    """graph.npz: the reference's own stage outputs of the LOL.yml graph (A conditional encoder, B flow reverse, C codebook,
    D VQGAN decoder) on a 1x3x24x32 input with name-seeded weights.  A on the fixture's input; B, C, D each on the fixture's
    (i.e. the reference's) input for that stage."""
    g = golden("graph")
    pg = seeded_init_(M.VQLLFLOWDeformable().eval(), 0).cuda()
    pv = seeded_init_(M.VQModel().eval(), 1).cuda()
    with torch.no_grad():
        enc = pg.RRDB.forward_nhwc(torch.from_numpy(g["lr"]).cuda())
        within(rel(ops.nhwc_to_nchw(enc["cond_feat"]), g["cond_feat"]), TOL["cond_feat"])
        within(rel(ops.nhwc_to_nchw(enc["color_map"]), g["color_map"]), TOL["color_map"])
        within(rel(ops.nhwc_to_nchw(enc["mid_feat"][0])[:, :8], g["mid0"]), TOL_SLICE["mid0"])
        within(rel(ops.nhwc_to_nchw(enc["mid_feat"][1])[:, :8], g["mid1"]), TOL_SLICE["mid1"])
        z = pg.flowUpsamplerNet.decode_nhwc(nhwc(g["color_map"], bf16=False), nhwc(g["cond_feat"]))
        within(rel(ops.nhwc_to_nchw(z), g["latent"]), TOL["latent"])
        idx, img, feats = pv.decode_nhwc(nhwc(g["latent"], bf16=False), want_image=True)
        assert np.array_equal(idx.cpu().numpy(), g["idx"])                                # bit-exact on the reference's latent
        within(rel(img, g["rec"]), TOL["vq_rec"])
        within(rel(ops.nhwc_to_nchw(feats[0])[:, :8], g["code0"]), TOL_SLICE["code0"])
        within(rel(ops.nhwc_to_nchw(feats[1])[:, :8], g["code1"]), TOL_SLICE["code1"])
