"""CPU, world_size 2 over gloo: the N > 1 sharding + gather path of glare_amd/parallel.py."""
import os
import socket

import torch
import torch.multiprocessing as mp

from glare_amd import parallel


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 15, 32, 100):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, dev = parallel.init_from_env(backend="gloo")
    data = torch.arange(n_items, dtype=torch.float32).view(n_items, 1, 1).expand(n_items, 2, 3).contiguous()

    def enhance(lo, hi):  # stands in for the HIP pipeline: any per-image function
        return data[lo:hi] * 2 + 1

    local = parallel.run_sharded(n_items, enhance, r, w, batch=4)
    full = parallel.gather_results(local, n_items, r, w)
    if r == 0:
        q.put(full)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n = 13  # ragged: ranks get 7 and 6 images
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = torch.arange(n, dtype=torch.float32).view(n, 1, 1).expand(n, 2, 3) * 2 + 1
    assert torch.equal(full, ref)


def _grad_worker(rank, world, port, q):
    """The training step's data-parallel exchange (glare_amd/train.py): flat gradient buffers, one all-reduce per
    group; the Adam kernel itself is GPU-only and covered by tests/test_gpu_train.py."""
    from glare_amd.train import FlatGroup

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    grp = FlatGroup(list(net.parameters()), lr=1e-3)
    assert all(p.data_ptr() >= grp.w.data_ptr() for p in net.parameters())       # parameters are views of the flat buffer
    grp.zero_grad()
    x = torch.full((2, 5), float(rank + 1))
    net(x).sum().backward()
    grp.collect()                                                                 # gathers .grad into the flat buffer
    assert all(p.grad.data_ptr() >= grp.g.data_ptr() for p in net.parameters())
    local = grp.g.clone()
    w = grp.all_reduce()
    q.put((rank, w, local, grp.g.clone()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_flat_gradient_all_reduce_two_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, w0, l0, s0), (_, w1, l1, s1) = got
    assert w0 == w1 == 2
    assert float(l0.abs().sum()) > 0 and not torch.equal(l0, l1)
    assert torch.allclose(s0, l0 + l1) and torch.equal(s0, s1)


def _branch_worker(rank, world, port, q):
    """Two ranks draw DIFFERENT stage-2 branches (LLFlowVQGAN_arch.py:95): rank 1's graph does not reach the last layer's
    bias (stands in for RRDB.color_conv under mean = gt).  The set of parameters Adam updates must not depend on that."""
    from glare_amd.train import FlatGroup

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    unused = torch.nn.Parameter(torch.ones(7))                       # a parameter no graph ever reaches (flowUpsamplerNet.f)
    grp = FlatGroup(list(net.parameters()) + [unused], lr=1e-3, never_used=[unused])
    grp.zero_grad()
    x = torch.full((2, 5), float(rank + 1))
    h = net[0](x)
    out = h @ net[1].weight.t() + (net[1].bias if rank == 0 else 0.0)   # rank 1: no gradient for net[1].bias
    out.sum().backward()
    assert (net[1].bias.grad is None) == (rank == 1)
    grp.collect()
    grp.all_reduce()
    q.put((rank, grp.active_ranges(), grp.g.clone(), [p.grad is None for p in grp.params]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_active_parameter_set_is_rank_invariant():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_branch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, r0, g0, n0), (_, r1, g1, n1) = got
    assert r0 == r1 == [(0, 5 * 4 + 4 + 4 * 3 + 3)], (r0, r1)        # both ranks update everything but the never-used tail
    assert torch.equal(g0, g1)                                         # ... with the same all-reduced gradient
    assert n0 == n1 == [False, False, False, False, True]              # .grad stays None only for the never-used parameter
    assert float(g0[-10:-7].abs().sum()) > 0                           # the bias gradient of the rank that had one arrived


def test_never_used_parameter_with_a_gradient_is_an_error():
    import pytest

    from glare_amd.train import FlatGroup

    p = torch.nn.Parameter(torch.ones(3))
    grp = FlatGroup([p], lr=1e-3, never_used=[p])
    (p * 2).sum().backward()
    with pytest.raises(RuntimeError):
        grp.collect()


def _early_worker(rank, world, port, q):
    """The overlapped exchange of a training step (FlatGroup.arm_early_all_reduce): the LATE layers' group (the flow, in stage 2)
    starts its all-reduce from a multi-grad hook while the backward pass of the early layers' group (the conditional encoder) is
    still running; the result must equal the blocking path's, and .grad must end up as views of the all-reduced flat buffer."""
    from glare_amd.train import FlatGroup

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(0)
    early_layers, late_layers = torch.nn.Linear(5, 4), torch.nn.Linear(4, 3)       # "encoder" in front of the "flow"
    unused = torch.nn.Parameter(torch.ones(2))
    g_late = FlatGroup(list(late_layers.parameters()) + [unused], lr=1e-3, never_used=[unused])
    g_early = FlatGroup(list(early_layers.parameters()), lr=1e-3)
    x = torch.full((2, 5), float(rank + 1))
    order = []
    early_layers.weight.register_hook(lambda g: order.append("early-layer grad"))

    def run(overlap):
        for g in (g_late, g_early):
            g.zero_grad()
        loss = late_layers(torch.tanh(early_layers(x))).sum()
        if overlap:
            g_late.arm_early_all_reduce()
            fire = g_late._early["handle"]
            assert fire is not None
        loss.backward()
        out = []
        for g in (g_late, g_early):
            w = g.finish_early_all_reduce()
            if w is None:
                g.collect()
                w = g.all_reduce()
            out.append((w, g.g.clone()))
        return out

    blocking = run(False)
    overlapped = run(True)
    views = all(p.grad is not None and p.grad.data_ptr() >= g_late.g.data_ptr() and
                p.grad.data_ptr() < g_late.g.data_ptr() + g_late.g.numel() * 4 for p in late_layers.parameters())
    q.put((rank, [w for w, _ in blocking], [w for w, _ in overlapped],
           all(torch.equal(a[1], b[1]) for a, b in zip(blocking, overlapped)), views, unused.grad is None))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_early_all_reduce_equals_the_blocking_exchange():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_early_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, wb, wo, same, views, unused_none in got:
        assert wb == wo == [2, 2]
        assert same and views and unused_none


def test_early_all_reduce_is_a_no_op_without_a_process_group():
    from glare_amd.train import FlatGroup

    lin = torch.nn.Linear(3, 2)
    grp = FlatGroup(list(lin.parameters()), lr=1e-3)
    grp.zero_grad()
    grp.arm_early_all_reduce()
    lin(torch.ones(1, 3)).sum().backward()
    assert grp.finish_early_all_reduce() is None          # world size 1: the step's blocking path (and its hipGraph) is untouched
    grp.collect()
    assert grp.all_reduce() == 1
