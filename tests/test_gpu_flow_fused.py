"""GPU: the fused coupling step of the flow's reverse pass (csrc/flow_fused.hip, round 6) against the four-launch form it replaces
(flow_h1 -> 1x1 MFMA conv -> 3x3 MFMA conv -> flow_tail, csrc/flow.hip + conv_igemm) and against the CPU oracle's flow
(oracle/torch_ref.py, bit-identical to FlowUpsamplerNet.decode, FlowUpsamplerNet.py:290-326)."""
import importlib

import pytest
import torch

from glare_amd import modules as M

from glare_amd import ops
from glare_amd.synthetic import representative_init_
from oracle import torch_ref as O

FU = importlib.import_module("glare_amd.modules.FlowUpsamplerNet")   # the module (the package re-exports the class under this name)

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def nets():
    og, ov = representative_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), O.VQModel().eval(), 0, batch=4, size=192)
    pg = M.VQLLFLOWDeformable().eval()
    pg.load_state_dict(og.state_dict())
    return og, pg.cuda()


def _decode(pg, z, ft, fused):
    prev = FU.FUSED_STEP
    FU.FUSED_STEP = fused
    try:
        pg.flowUpsamplerNet.invalidate()
        with torch.no_grad():
            return pg.flowUpsamplerNet.decode_nhwc(z, ft)
    finally:
        FU.FUSED_STEP = prev
        pg.flowUpsamplerNet.invalidate()


@pytest.mark.parametrize("shape", [(2, 21, 37), (1, 8, 32), (3, 9, 33), (1, 105, 155)])
@pytest.mark.parametrize("precision,pair", [("fp16", True), ("fp16", False), ("bf16", False)])
def test_fused_step_matches_four_launch_form(nets, shape, precision, pair):
    """Same inputs, both forms, all 24 coupling steps (ragged tiles, one-tile and multi-tile images, full latent size).  Under the
    fp32-class precision both forms carry ~22 bits per operand: they differ by summation order only.  With a 16-bit cond_feat
    the old form rounds h1 / h2 to 16 bits where the fused kernel keeps pairs: the bound there is the old form's own error."""
    og, pg = nets
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + H)
    with ops.use_precision(precision):
        z = (torch.randn(B, H, W, 3, generator=g) * 0.7).cuda()
        ft32 = torch.sigmoid(torch.randn(B, H, W, 64, generator=g)).cuda()
        ft = ops.split_hilo(ft32) if pair else ft32.to(ops.act_dtype())
        a = _decode(pg, z, ft, True)
        b = _decode(pg, z, ft, False)
    assert torch.isfinite(a).all()
    e = rel(a, b)
    print("fused vs four-launch %s %s pair=%s: rel %.2e" % (shape, precision, pair, e))
    # measured: 3.8e-7 / 2.6e-4 / 2.1e-3
    assert e < (8e-7 if pair else (5.5e-4 if precision == "fp16" else 4.5e-3)), e


def test_fused_flow_against_oracle(nets):
    """The reverse flow on the oracle's conditional features: fused form vs the fp32 CPU oracle (and the four-launch form beside it)."""
    og, pg = nets
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 26, 41
    z = torch.randn(B, 3, H, W, generator=g) * 0.7
    ft = torch.sigmoid(torch.randn(B, 64, H, W, generator=g))
    with torch.no_grad():
        ref, _ = og.flowUpsamplerNet.decode(z, ft)
    with ops.use_precision("fp16"):
        zz = ops.nchw_to_nhwc(z.cuda(), bf16=False)
        fp = ops.split_hilo(ft.permute(0, 2, 3, 1).contiguous().cuda())
        a = ops.nhwc_to_nchw(_decode(pg, zz, fp, True))
        b = ops.nhwc_to_nchw(_decode(pg, zz, fp, False))
    ea, eb = rel(a, ref), rel(b, ref)
    print("flow reverse vs oracle: fused %.2e, four-launch %.2e" % (ea, eb))
    assert ea < 7e-6 and ea < 1.5 * eb, (ea, eb)        # measured 3.26e-6 / 3.27e-6


def test_fused_step_rejects_in_place():
    from glare_amd import _lib
    z = torch.zeros(1, 8, 32, 3, device="cuda")
    ftA = torch.zeros(1, 8, 32, 64, device="cuda")
    hF = torch.zeros(1, 8, 32, 8, device="cuda")
    img = torch.zeros(int(_lib.lib().glare_flow_step_fused_image_bytes()), dtype=torch.uint8, device="cuda")
    with pytest.raises(AssertionError):
        ops.flow_step_fused(z, z, ftA, 0, img, hF, 0, [1, 0, 0, 0, 1, 0, 0, 0, 1], [0, 0, 0])
