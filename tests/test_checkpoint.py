"""Row f3: checkpoint files are interchangeable with the reference's (base_model.py:93-122,188-219) -- CPU-only checks of the
file formats; the optimizer arithmetic itself is covered on the GPU (tests/test_gpu_train.py::test_adam_matches_torch)."""
import torch

from glare_amd import checkpoint as C
from glare_amd.train import FlatAdam, FlatGroup


def _net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))


def test_network_roundtrip_and_dataparallel_prefix(tmp_path):
    a, b = _net(0), _net(1)
    C.save_network(a, str(tmp_path / "10_G.pth"))
    sd = torch.load(str(tmp_path / "10_G.pth"))
    assert list(sd) == list(a.state_dict()) and all(v.device.type == "cpu" for v in sd.values())
    C.load_network({"module." + k: v for k, v in sd.items()}, b)                # a file saved from nn.DataParallel
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))


def test_adam_state_is_torch_optim_layout(tmp_path):
    """A state written by torch.optim.Adam (what the reference saves) resumes FlatAdam, and the other way round."""
    net_t, net_f = _net(2), _net(2)
    ps = list(net_t.parameters())
    topt = torch.optim.Adam([{"params": ps[:2], "lr": 5e-4, "weight_decay": 0.0}, {"params": ps[2:], "lr": 5e-4, "weight_decay": 1e-5}])
    for _ in range(3):
        topt.zero_grad()
        net_t(torch.randn(4, 6)).square().mean().backward()
        topt.step()
    pf = list(net_f.parameters())
    fopt = FlatAdam([FlatGroup(pf[:2], 1.0), FlatGroup(pf[2:], 1.0)])
    C.load_adam_state_dict(fopt, topt.state_dict())
    assert fopt.t == 3 and fopt.groups[0].lr == 5e-4 and fopt.groups[1].weight_decay == 1e-5
    assert torch.equal(fopt.groups[0].m[:30].view(5, 6), topt.state[ps[0]]["exp_avg"])
    assert torch.equal(fopt.groups[1].v[-3:], topt.state[ps[3]]["exp_avg_sq"])
    # and back: torch.optim.Adam accepts what FlatAdam writes
    class Tr:
        opt = fopt
    C.save_training_state(str(tmp_path / "30.state"), Tr, epoch=1, iter_step=30)
    st = torch.load(str(tmp_path / "30.state"))
    assert set(st) == {"epoch", "iter", "schedulers", "optimizers", "scaler"}
    topt2 = torch.optim.Adam([{"params": ps[:2]}, {"params": ps[2:]}])
    topt2.load_state_dict(st["optimizers"][0])
    assert torch.equal(topt2.state[ps[1]]["exp_avg"], topt.state[ps[1]]["exp_avg"])
    assert topt2.param_groups[1]["weight_decay"] == 1e-5
    fopt2 = FlatAdam([FlatGroup(list(_net(2).parameters())[:2], 1.0), FlatGroup(list(_net(2).parameters())[2:], 1.0)])
    assert C.resume_training(str(tmp_path / "30.state"), type("T", (), {"opt": fopt2})) == (1, 30)
    assert fopt2.t == 3 and torch.equal(fopt2.groups[0].m, fopt.groups[0].m)


def _reference_adam(netG, wd_G=0.0, lr=5e-4):
    """torch.optim.Adam built the way the reference builds it (LLFlow_model.py:95-118 / VQLLFLOWD_model.py:101-121): two groups,
    [other, RRDB], from the parameters with requires_grad; the names carry nn.DataParallel's 'module.' prefix there, so
    '.RRDB.' in the name == our names starting with 'RRDB.'."""
    rrdb, other = [], []
    for k, v in netG.named_parameters():
        if v.requires_grad:
            (rrdb if ".RRDB." in "module." + k else other).append(v)
    return torch.optim.Adam([{"params": other, "lr": lr, "beta1": 0.9, "beta2": 0.99, "weight_decay": wd_G},
                             {"params": rrdb, "lr": lr, "beta1": 0.9, "beta2": 0.99, "weight_decay": 1e-5}])


def _fake_step(trainer, topt):
    """Give both optimizers the same state without a GPU: every used parameter gets a gradient; the parameters the graph
    never reaches (flowUpsamplerNet.f, deformable_decoder.{scale,bias,enc,conv_out}) get none, as after a real backward."""
    unused = ("flowUpsamplerNet.f.", "deformable_decoder.scale.", "deformable_decoder.bias.", "deformable_decoder.enc.",
              "deformable_decoder.conv_out.")
    names = {id(p): n for n, p in trainer.netG.named_parameters()}
    g = torch.Generator().manual_seed(0)
    for grp in trainer.opt.groups:
        grp.zero_grad()
        for p in grp.params:
            if not names[id(p)].startswith(unused):
                p.grad = torch.randn(p.shape, generator=g) * 1e-3
        grp.collect()
    topt.step()                         # p.grad of the used parameters are now views of the flat buffers: same values
    trainer.opt._t = 1
    for grp in trainer.opt.groups:      # the Adam arithmetic itself is a GPU kernel (tests/test_gpu_train.py); mirror torch's here
        off = 0
        for p in grp.params:
            k = p.numel()
            st = topt.state.get(p)
            if st:
                grp.m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                grp.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            off += k


def test_real_trainers_write_and_read_the_reference_state_layout(tmp_path):
    """The `.state` of the real Stage2 / Stage3 trainers against torch.optim.Adam + GradScaler built as the reference builds them:
    two param groups with an EMPTY RRDB group where the encoder is frozen, no state entry for parameters without a gradient,
    a scaler dict torch's GradScaler.load_state_dict accepts -- in both directions."""
    from glare_amd import modules as M
    from glare_amd.train import Stage2Trainer, Stage3Trainer

    cases = [("stage2", lambda: Stage2Trainer(M.LLFlowVQGAN2(), M.VQModel(), lr_G=5e-4)),
             ("stage2 frozen encoder", lambda: Stage2Trainer(M.LLFlowVQGAN2(), M.VQModel(), lr_G=5e-4, train_rrdb=False)),
             ("stage3", lambda: Stage3Trainer(M.VQLLFLOWDeformable(), M.VQModel(), lr_G=5e-4))]
    for tag, make in cases:
        tr = make()
        topt = _reference_adam(tr.netG)
        assert [len(g["params"]) for g in topt.param_groups] == [len(g.params) for g in tr.opt.groups], tag
        if tag != "stage2":
            assert len(tr.opt.groups) == 2 and len(tr.opt.groups[1].params) == 0, tag
        _fake_step(tr, topt)
        path = str(tmp_path / "s.state")
        C.save_training_state(path, tr, epoch=3, iter_step=1)
        st = torch.load(path)
        # ours -> the reference's resume path (base_model.py:207-219)
        topt2 = _reference_adam(tr.netG)
        topt2.load_state_dict(st["optimizers"][0])
        torch.amp.GradScaler("cpu").load_state_dict(st["scaler"])       # an empty dict raises here
        ref_sd = topt.state_dict()
        assert sorted(st["optimizers"][0]["state"]) == sorted(ref_sd["state"]), tag      # same parameters have state
        for k, v in ref_sd["state"].items():
            assert torch.equal(st["optimizers"][0]["state"][k]["exp_avg"], v["exp_avg"])
        # the reference's -> ours
        tr2 = make()
        ep, it = C.resume_training({"epoch": 3, "iter": 1, "schedulers": [], "optimizers": [ref_sd],
                                    "scaler": torch.amp.GradScaler("cpu").state_dict()}, tr2)
        assert (ep, it) == (3, 1) and tr2.opt.t == 1
        for a, b in zip(tr.opt.groups, tr2.opt.groups):
            assert torch.equal(a.m, b.m) and torch.equal(a.v, b.v) and a.has_grad == b.has_grad
        assert any(not all(g.has_grad) for g in tr2.opt.groups), "the never-used parameters must stay without state"
