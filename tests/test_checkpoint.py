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
