"""Data-parallel inference over the GPUs of one node: one process per GPU, images sharded, no
collective on the data path (images are independent -- SURVEY.md section 8e); the only exchange is the
final gather of per-image results to rank 0 over RCCL (`torch.distributed`, backend "nccl" on ROCm;
"gloo" in the CPU tests).  Replaces the reference's single-process nn.DataParallel
(code/models/VQLLFLOWD_model.py:72-75), which replicates weights and gathers outputs every iteration."""
import os

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
    latency-bound sections of one batch run under the next batch's kernels (+3 % at batch 8); the results are joined after a
    device synchronisation."""
    lo, hi = shard_range(n_items, rank, world)
    starts = list(range(lo, hi, batch))
    if streams > 1 and torch.cuda.is_available():
        pool = [torch.cuda.Stream() for _ in range(streams)]
        outs = []
        for i, s in enumerate(starts):
            with torch.cuda.stream(pool[i % streams]):
                outs.append(fn(s, min(hi, s + batch)))
        torch.cuda.synchronize()
    else:
        outs = [fn(s, min(hi, s + batch)) for s in starts]
    return torch.cat(outs, 0) if outs else None
