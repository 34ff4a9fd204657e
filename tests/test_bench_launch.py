"""CPU: `bench.py --gpus N` started as a plain process must start N ranks itself (glare_amd.parallel.launch_ranks), and a
launched job really has WORLD_SIZE = N ranks that can talk (gloo here; nccl = RCCL on the GPU box)."""
import os
import subprocess
import sys
import textwrap

from glare_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rank_launch_command_shape():
    cmd = parallel.rank_launch_command("bench.py", 4, ["--gpus", "4", "--steps", "3"], port=29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == ["bench.py", "--gpus", "4", "--steps", "3"]


def test_launch_ranks_starts_world_size_processes(tmp_path):
    script = tmp_path / "probe.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        from glare_amd import parallel
        rank, world, dev = parallel.init_from_env(backend="gloo")
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        g = parallel.RankGather(torch.zeros(2), rank, world)
        bufs = g.gather(torch.full((2,), float(rank)))
        if rank == 0:
            print("WORLD", world, "SUM", int(t.item()), "GATHER", [int(b[0]) for b in bufs], sys.argv[1:], flush=True)
        dist.barrier(); dist.destroy_process_group()
    """ % ROOT))
    env = dict(os.environ, RANK="7", WORLD_SIZE="9")      # stale variables of an outer job must not leak into the launch
    r = subprocess.run([sys.executable, "-c",
                        "import sys; sys.path.insert(0, %r); from glare_amd import parallel; "
                        "sys.exit(parallel.launch_ranks(%r, 2, ['--flag', '1']))" % (ROOT, str(script))],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "WORLD 2 SUM 3 GATHER [0, 1] ['--flag', '1']" in r.stdout


def test_bench_refuses_a_mismatched_world(monkeypatch):
    """bench.py's own contract, read from its source (it cannot run without a GPU): self-launch when WORLD_SIZE is unset,
    and an assertion that the world size equals --gpus."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'args.gpus > 1 and "WORLD_SIZE" not in os.environ' in src and "parallel.launch_ranks(" in src
    assert "assert world == args.gpus" in src and "dist.get_world_size() == args.gpus" in src
    assert '"n_gpus": world' in src


import pytest


@pytest.mark.parametrize("literal", [False, True])
def test_bench_world_2_skeleton_runs_over_gloo(literal):
    """literal: BASELINE configs[2] as written -- a GLOBAL batch split over the ranks (`--global-batch`, strong scaling) instead of
    `--batch` images per GPU at every N.
    VERDICT r03: bench.py's own world > 1 body -- process-group init, the per-step RankGather inside the timed region, the barrier
    / synchronize brackets, the max over ranks, the train block's per-group all-reduce (early + blocking) -- executed for real by two
    ranks, on CPU over gloo, with the HIP pipeline replaced by tensor stand-ins (GLARE_BENCH_STUB=1).  Started as a plain process,
    so the self-launch is part of it."""
    import json

    env = dict(os.environ, GLARE_BENCH_STUB="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    shape = ["--global-batch", "4"] if literal else ["--batch", "2"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"] + shape,
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE line
    res = json.loads(lines[0])
    assert res["stub"] is True and res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1
    assert res["config"]["global_batch"] == 4 and res["config"]["parallelism"] == "dp2" and res["config"]["batch_per_gpu"] == 2
    assert res["scaling"] == ("strong" if literal else "weak")
    assert ("configs[2] as written" in res["config"]["workload"]) == literal
    assert res["value"] > 0 and abs(res["value"] - 4 * 3 / (res["ms_per_step"] * 3e-3)) / res["value"] < 1e-3
    assert res["train"]["world"] == 2 and res["train"]["stage2_ms_per_step"] > 0 and "error" not in res["train"]


def test_bench_prints_its_line_when_the_train_block_hangs():
    """The all-reduce leg of the train block has never run on more than one GPU: if it ever hangs there, the headline line must
    still come out.  A stub trainer that never returns (GLARE_BENCH_STUB_HANG=1) under `--train-timeout 3`: rank 0's watchdog prints
    the inference line with the error recorded in `train` and ends the process."""
    import json

    env = dict(os.environ, GLARE_BENCH_STUB="1", GLARE_BENCH_STUB_HANG="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--train-timeout", "3"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["value"] > 0 and "did not finish" in res["train"]["error"]


def _run_bench_stub(args, extra_env=None, timeout=900):
    import json

    env = dict(os.environ, GLARE_BENCH_STUB="1", **(extra_env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, [json.loads(l) for l in lines]


@pytest.mark.parametrize("n", [4, 8])
def test_bench_configs2_as_written_on_4_and_8_ranks(n):
    """BASELINE configs[2] as the driver would launch it on a full node -- `--gpus N --global-batch 32` -- through bench.py's own
    multi-rank body over gloo: every rank counted by the 4-byte all-reduce (`ranks_seen`), the per-rank rates
    gathered, one line."""
    r, lines = _run_bench_stub(["--gpus", str(n), "--steps", "2", "--warmup", "1", "--global-batch", "32", "--no-train"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    res = lines[0]
    cfg = res["config"]
    assert res["n_gpus"] == n and cfg["global_batch"] == 32 and cfg["batch_per_gpu"] == 32 // n and res["scaling"] == "strong"
    assert cfg["ranks_seen"] == n and len(cfg["per_rank_images_per_sec"]) == n and all(v > 0 for v in cfg["per_rank_images_per_sec"])
    assert isinstance(cfg["exchange"], str) and res["value"] > 0


def test_bench_prints_its_line_when_the_first_gather_hangs():
    """A rank that never reaches the per-step gather (GLARE_BENCH_STUB_HANG_GATHER=1: rank 1 sleeps in front of it): rank 0 sits in the
    collective of its FIRST step.  The watchdog over the whole N > 1 body (`--dist-timeout`) prints the line with the stage that hung in
    `config.exchange.error` and ends the job -- no line at all was the failure mode before."""
    r, lines = _run_bench_stub(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--dist-timeout", "8", "--no-train"],
                               {"GLARE_BENCH_STUB_HANG_GATHER": "1"}, timeout=300)
    assert len(lines) == 1, (r.stdout[-1500:], r.stderr[-1500:])
    res = lines[0]
    assert res["stub"] is True and res["n_gpus"] == 2 and res["value"] is None
    err = res["config"]["exchange"]["error"]
    assert "did not finish within 8 s" in err and ("first step" in err or "gather" in err)
