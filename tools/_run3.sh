cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout 1500 python -m pytest tests -m gpu -q 2>&1 > gpurun_out/r3c/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r3c/bench.log 2>&1
tail -4 gpurun_out/r3c/pytest.log
tail -2 gpurun_out/r3c/bench.log
