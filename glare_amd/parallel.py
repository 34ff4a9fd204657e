"""Data-parallel inference over the GPUs of one node: one process per GPU, images sharded, no
collective on the data path (images are independent -- SURVEY.md section 8e); the only exchange is the
final gather of per-image results to rank 0 over RCCL (`torch.distributed`, backend "nccl" on ROCm;
"gloo" in the CPU tests).  Replaces the reference's single-process nn.DataParallel
(code/models/VQLLFLOWD_model.py:72-75), which replicates weights and gathers outputs every iteration."""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_items for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, world, device)."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world)
    return rank, world, device


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_launch_command(script, n_ranks, argv, port=None, module=False):
    """The `torch.distributed.run` command line that starts `script argv...` as n_ranks processes of ONE node, one per GPU
    (rendezvous on 127.0.0.1: the container hostname may not resolve)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_ranks)),
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port())]
    if module:
        cmd.append("-m")
    return cmd + [script] + list(argv)


def launch_ranks(script, n_ranks, argv, env=None, **popen_kw):
    """Self-launch used by `bench.py --gpus N` when it is started as a plain process (no WORLD_SIZE in the environment):
    re-executes the script under torch.distributed.run with n_ranks ranks, passes its stdout/stderr through and returns
    its exit code.  Rank 0 of the child job prints the result line."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what the host driver supports for RCCL between processes
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    return subprocess.run(rank_launch_command(script, n_ranks, argv), env=e, **popen_kw).returncode


class RankGather:
    """Per-step gather of equal-sized per-rank results to rank 0 with buffers allocated ONCE (the inference exchange of
    BASELINE configs[2]: "RCCL gather only").  `gather(t)` enqueues on the current stream; rank 0's `bufs` then hold every
    rank's tensor in rank order."""

    def __init__(self, like, rank, world, dst=0):
        self.rank, self.world, self.dst = rank, world, dst
        self.bufs = [torch.empty_like(like) for _ in range(world)] if (world > 1 and rank == dst) else None

    def gather(self, t):
        if self.world == 1:
            return t
        dist.gather(t, self.bufs, dst=self.dst)
        return self.bufs


def gather_results(local_tensor, n_total, rank, world, dst=0):
    """Gathers ragged per-rank slices [n_local, ...] back into dataset order on `dst` ([n_total, ...])."""
    if world == 1:
        return local_tensor
    q, r = divmod(n_total, world)
    n_max = q + (1 if r else 0)
    pad = torch.zeros((n_max,) + tuple(local_tensor.shape[1:]), dtype=local_tensor.dtype, device=local_tensor.device)
    pad[: local_tensor.shape[0]] = local_tensor
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = []
    for rk in range(world):
        lo, hi = shard_range(n_total, rk, world)
        out.append(bufs[rk][: hi - lo])
    return torch.cat(out, 0)


def run_sharded(n_items, fn, rank, world, batch=8, streams=1):
    """Calls fn(lo, hi) -> tensor [hi-lo, ...] over this rank's slice in batches; returns the local results.
    streams > 1 (CUDA only): consecutive batches are issued round-robin on that many streams, so the tail and the
    latency-bound sections of one batch run under the next batch's kernels (+3 % at batch 8); the side streams wait on the
    caller's stream first and the caller's stream waits on them before the results are joined (stream-ordered, no host sync)."""
    lo, hi = shard_range(n_items, rank, world)
    starts = list(range(lo, hi, batch))
    if streams > 1 and torch.cuda.is_available():
        cur = torch.cuda.current_stream()
        pool = [torch.cuda.Stream() for _ in range(streams)]
        for st in pool:
            st.wait_stream(cur)          # weights / inputs produced asynchronously on the caller's stream are ordered before us
        outs = []
        for i, s in enumerate(starts):
            with torch.cuda.stream(pool[i % streams]):
                o = fn(s, min(hi, s + batch))
                if o is not None:
                    o.record_stream(cur)  # consumed (concatenated) on the caller's stream below
                outs.append(o)
        for st in pool:
            cur.wait_stream(st)          # stream-ordered join: no device-wide synchronisation needed by the caller
    else:
        outs = [fn(s, min(hi, s + batch)) for s in starts]
    return torch.cat(outs, 0) if outs else None
