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
