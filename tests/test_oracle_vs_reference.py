"""CPU, build container only: pins oracle/torch_ref.py against the reference EXECUTED in place
(oracle/refimport.py).  Skipped where /root/reference does not exist (the GPU box)."""
import numpy as np
import pytest
import torch

from glare_amd.synthetic import reset_actnorms_, seeded_init_, synthetic_lowlight
from oracle import refimport as R
from oracle import torch_ref as O

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not R.available(), reason="reference tree not present")]


@pytest.fixture(scope="module")
def nets():
    torch.manual_seed(0)
    np.random.seed(0)
    netG, opt = R.build_netG()
    net_vq, _ = R.build_vqgan(opt)
    seeded_init_(netG.eval(), 0)
    seeded_init_(net_vq.eval(), 1)
    oG = O.VQLLFLOWDeformable().eval()
    oV = O.VQModel().eval()
    oG.load_state_dict(netG.state_dict(), strict=True)  # 824 keys, names identical
    assert set(oV.state_dict().keys()) == set(net_vq.state_dict().keys())
    oV.load_state_dict(net_vq.state_dict(), strict=True)
    return netG, net_vq, oG, oV


def test_state_dict_surface(nets):
    netG, net_vq, oG, oV = nets
    assert len(netG.state_dict()) == 824
    assert sorted(oG.state_dict().keys()) == sorted(netG.state_dict().keys())


def test_inference_stages_bit_identical(nets):
    netG, net_vq, oG, oV = nets
    lr = O.preprocess(synthetic_lowlight(1, 12, 20, seed=5)[0])  # 1x3x32x40
    with torch.no_grad():
        er = netG.RRDB(lr, mid_feat=True)
        eo = oG.RRDB(lr, mid_feat=True)
        for k in ("cond_feat", "color_map"):
            assert torch.equal(er[k], eo[k])
        for a, b in zip(er["mid_feat"], eo["mid_feat"]):
            assert torch.equal(a, b)
        xr, _ = netG.flowUpsamplerNet(rrdbResults=er, z=er["color_map"], eps_std=0, reverse=True,
                                      logdet=torch.zeros(1))
        xo, _ = oG.flowUpsamplerNet.decode(eo["color_map"], eo["cond_feat"])
        assert torch.equal(xr, xo)
        rr, lr_, fr = net_vq.decode(xr)
        ro, lo, fo = oV.decode(xo)
        assert torch.equal(rr, ro) and torch.equal(lr_, lo)
        assert torch.equal(net_vq.quantize(xr)[2][2], oV.last_indices)
        for a, b in zip(fr, fo):
            assert torch.equal(a, b)


def test_stage_e_with_oracle_dcn(nets, monkeypatch):
    """MultiScaleDecoder2 glue (Mix, WarpBlock wiring, whole-tensor mean rescale): the reference runs
    with its CUDA-only DCN op swapped for the oracle's, everything else is the reference's code."""
    netG, net_vq, oG, oV = nets
    import models.modules.deformableDecoder_arch as dd

    monkeypatch.setattr(dd, "modulated_deform_conv", O.modulated_deform_conv)
    lr = torch.cat([O.preprocess(im) for im in synthetic_lowlight(2, 4, 12, seed=6)])  # 2x3x24x32
    with torch.no_grad(), R.cpu_only():
        ref_out, ref_lat = netG(net_vq=net_vq, lr=lr, z=None, eps_std=0, reverse=True, reverse_with_grad=False)
        out, lat = oG(oV, lr)
    assert torch.equal(ref_lat, lat)
    np.testing.assert_allclose(out.numpy(), ref_out.numpy(), atol=1e-5)


def test_stage2_normal_flow(nets):
    netG, net_vq, oG, oV = nets
    import models.modules.LLFlowVQGAN_arch as arch

    opt = R.load_opt()
    opt["train_gt_ratio"] = 0.0
    ref = arch.LLFlowVQGAN2(opt=opt, K=12).eval()
    seeded_init_(ref, 2)
    mine = O.LLFlowVQGAN2().eval()
    mine.load_state_dict(ref.state_dict(), strict=True)
    g = torch.Generator().manual_seed(3)
    lr = torch.log(torch.rand(2, 3, 32, 32, generator=g) * 0.3 + 1e-3)
    gt = torch.randn(2, 3, 8, 8, generator=g)
    lr_r, gt_r = lr.clone().requires_grad_(), gt.clone()
    z_r, nll_r, ld_r = ref(gt=gt_r, lr=lr_r, reverse=False)
    z_o, nll_o, ld_o = mine.normal_flow(gt, lr)
    assert torch.equal(z_r, z_o)
    np.testing.assert_allclose(nll_o.detach().numpy(), nll_r.detach().numpy(), rtol=1e-6)
    # backward of the stage-2 objective
    nll_r.mean().backward()
    nll_o.mean().backward()
    gr = dict(ref.named_parameters())
    for n, p in mine.named_parameters():
        if p.grad is not None:
            np.testing.assert_allclose(p.grad.numpy(), gr[n].grad.numpy(), rtol=1e-4, atol=1e-6)


def test_actnorm_data_dependent_init_bit_identical(nets):
    """First TRAINING forward of a flow whose ActNorms are all-zero (FlowActNorms.py:32-46,82-83): every one of the 28 step
    ActNorms and the 48 + 48 coupling-net ActNorms (flow.py:48-52) takes its bias / logs from the batch, each seeing the layers
    initialised before it."""
    import models.modules.LLFlowVQGAN_arch as arch

    opt = R.load_opt()
    opt["train_gt_ratio"] = 0.0
    ref = arch.LLFlowVQGAN2(opt=opt, K=12)
    seeded_init_(ref, 4)
    mine = O.LLFlowVQGAN2()
    mine.load_state_dict(ref.state_dict(), strict=True)
    reset_actnorms_(ref)
    reset_actnorms_(mine)
    ref.train()
    mine.train()
    g = torch.Generator().manual_seed(7)
    lr = torch.log(torch.rand(2, 3, 32, 32, generator=g) * 0.3 + 1e-3)
    gt = torch.randn(2, 3, 8, 8, generator=g) * 1.7 + 0.4
    z_r, nll_r, _ = ref(gt=gt.clone(), lr=lr.clone(), reverse=False)
    z_o, nll_o, _ = mine.normal_flow(gt, lr)
    sr, so = ref.state_dict(), mine.state_dict()
    n = 0
    for k in so:
        if "actnorm" in k:
            assert torch.equal(sr[k], so[k]), k
            assert (so[k] != 0).any(), k
            n += 1
    assert n == 2 * (28 + 4 * 24)
    assert torch.equal(z_r, z_o)
    np.testing.assert_allclose(nll_o.detach().numpy(), nll_r.detach().numpy(), rtol=1e-6)
    # a second forward leaves the parameters alone (inited), and eval mode never initialises
    before = {k: v.clone() for k, v in mine.state_dict().items()}
    mine.normal_flow(gt * 2, lr)
    assert all(torch.equal(before[k], v) for k, v in mine.state_dict().items())
    fresh = O.LLFlowVQGAN2().eval()
    fresh.normal_flow(gt, lr)
    assert all((m.bias == 0).all() for m in fresh.modules() if isinstance(m, O.ActNorm2d))


def test_msssim_bit_identical_to_reference():
    R.install()
    from models.modules import pytorch_msssim as PM

    g = torch.Generator().manual_seed(5)
    gt = torch.rand(1, 3, 64, 80, generator=g)
    a = (gt + 0.2 * torch.randn(1, 3, 64, 80, generator=g)).clamp(0, 1)
    ar, ao = a.clone().requires_grad_(True), a.clone().requires_grad_(True)
    vr, vo = PM.msssim(ar, gt, normalize=True), O.msssim(ao, gt, normalize=True)
    assert torch.equal(vr, vo)
    vr.backward()
    vo.backward()
    assert torch.equal(ar.grad, ao.grad)


def test_representative_regime_recipe_on_the_reference_itself():
    """synthetic.representative_init_ (the trained-like weight regime of the end-to-end parity table) uses nothing but the modules'
    own forward passes -- so it runs on the REFERENCE's modules unchanged, and must leave them in exactly the state it leaves the
    oracle in (codebook from the reference's VQ encoder, ActNorms from the reference's own data-dependent initialisation)."""
    from glare_amd.synthetic import representative_init_

    torch.manual_seed(0)
    np.random.seed(0)
    netG, opt = R.build_netG()
    net_vq, _ = R.build_vqgan(opt)
    oG, oV = O.VQLLFLOWDeformable().eval(), O.VQModel().eval()
    with R.cpu_only():
        representative_init_(netG.eval(), net_vq.eval(), 3, batch=4, size=192)   # 4 x 48 x 48 = 9216 tokens >= the 8192 codes
    representative_init_(oG, oV, 3, batch=4, size=192)
    a, b = netG.state_dict(), oG.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in b), [k for k in b if not torch.equal(a[k], b[k])][:4]
    av, bv = net_vq.state_dict(), oV.state_dict()
    assert all(torch.equal(av[k], bv[k]) for k in bv)
    assert all((m.bias != 0).any() for m in oG.flowUpsamplerNet.modules() if isinstance(m, O.ActNorm2d))
