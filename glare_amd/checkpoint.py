"""Checkpoint compatibility with the reference.

Mirrors BaseModel.save_network / load_network / save_training_state / resume_training (code/models/base_model.py:93-122,
188-219) and the pretrained-weight loading of VQLLFLOWDModel.__init__ (code/models/VQLLFLOWD_model.py:42-63): network files are
plain CPU state dicts with the reference's 824 / 257 key names (a leading 'module.' from nn.DataParallel is stripped on load),
the training state is {'epoch', 'iter', 'schedulers', 'optimizers', 'scaler'} with each optimizer in torch.optim.Adam's own
state_dict layout -- so a `.state` file written by the reference resumes here and vice versa.
"""
from collections import OrderedDict

import torch


def save_network(network, path):
    network = getattr(network, "module", network)
    torch.save(OrderedDict((k, v.detach().cpu()) for k, v in network.state_dict().items()), path)


def load_network(path_or_state, network, strict=True, submodule=None):
    network = getattr(network, "module", network)
    if submodule is not None and str(submodule).lower() != "none":
        network = getattr(network, submodule)
    sd = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, str) else path_or_state
    clean = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in sd.items())
    result = network.load_state_dict(clean, strict=strict)
    if hasattr(network, "invalidate"):
        network.invalidate()          # packed kernel weights are stale
    return result


GRAD_SCALER_STATE = {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 0}
"""What an enabled torch.cuda.amp.GradScaler.state_dict() holds at its defaults.  The reference creates `GradScaler()`
(LLFlow_model.py:120, VQLLFLOWD_model.py:126) and its resume path calls `scaler.load_state_dict(state['scaler'])`, which raises on
an empty dict -- so the saved training state always carries these keys.  FlatAdam keeps the same state on the device (scale and
growth tracker move as GradScaler.update moves them; a step with inf / NaN gradients is skipped): saved from there, restored on resume."""


def adam_state_dict(opt):
    """FlatAdam -> the dict torch.optim.Adam.state_dict() would produce for the same groups (params indexed in order).  As in
    torch, parameters that never received a gradient have no state entry (FlatGroup.has_grad), and an empty group (the frozen
    RRDB group of stage 3 / stage 2 with train_RRDB: false) is emitted as an empty `params` list."""
    state, groups, idx = {}, [], 0
    for g in opt.groups:
        ids, off = [], 0
        used = getattr(g, "has_grad", None)
        for j, p in enumerate(g.params):
            k = p.numel()
            if used is None or used[j]:
                state[idx] = {"step": torch.tensor(float(opt.t)), "exp_avg": g.m[off:off + k].view(p.shape).detach().cpu().clone(),
                              "exp_avg_sq": g.v[off:off + k].view(p.shape).detach().cpu().clone()}
            ids.append(idx)
            idx += 1
            off += k
        groups.append({"lr": g.lr, "betas": tuple(opt.betas), "eps": opt.eps, "weight_decay": g.weight_decay, "amsgrad": False,
                       "maximize": False, "params": ids})
    return {"state": state, "param_groups": groups}


def load_adam_state_dict(opt, sd):
    """Accepts the layout above or one written by torch.optim.Adam in the reference (same group / parameter order)."""
    assert len(sd["param_groups"]) == len(opt.groups), "Wrong lengths of optimizers' param groups"
    step = 0
    for g, sg in zip(opt.groups, sd["param_groups"]):
        assert len(sg["params"]) == len(g.params), "parameter count of a group differs"
        g.lr, g.weight_decay = float(sg["lr"]), float(sg.get("weight_decay", 0.0))
        off = 0
        for p, pid in zip(g.params, sg["params"]):
            k = p.numel()
            st = sd["state"].get(pid)      # which parameters Adam updates is static (FlatGroup never_used), not read from the file
            if st is None:               # torch keeps no state for parameters that never received a gradient
                g.m[off:off + k].zero_()
                g.v[off:off + k].zero_()
            else:
                g.m[off:off + k].copy_(st["exp_avg"].reshape(-1).to(g.m.device))
                g.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1).to(g.v.device))
                step = max(step, int(float(st["step"])))
            off += k
    opt.t = step


def save_training_state(path, trainer, epoch, iter_step, schedulers=()):
    torch.save({"epoch": epoch, "iter": iter_step, "schedulers": [s.state_dict() for s in schedulers],
                "optimizers": [adam_state_dict(trainer.opt)],
                "scaler": trainer.opt.scaler_state_dict() if hasattr(trainer.opt, "scaler_state_dict") else dict(GRAD_SCALER_STATE)}, path)


def resume_training(path_or_state, trainer, schedulers=()):
    st = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, str) else path_or_state
    assert len(st["optimizers"]) == 1, "Wrong lengths of optimizers"
    assert len(st["schedulers"]) == len(schedulers), "Wrong lengths of schedulers"
    load_adam_state_dict(trainer.opt, st["optimizers"][0])
    if hasattr(trainer.opt, "load_scaler_state_dict"):
        trainer.opt.load_scaler_state_dict(st.get("scaler"))
    for s, sd in zip(schedulers, st["schedulers"]):
        s.load_state_dict(sd)
    return st["epoch"], st["iter"]
