cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout 900 python -m pytest tests/test_gpu_precision.py tests/test_gpu_harness.py tests/test_gpu_graph.py -m gpu -q -s 2>&1 | grep -v Warn | tail -80 > gpurun_out/r3g/pytest.log
tail -60 gpurun_out/r3g/pytest.log
for seed in 13; do timeout 600 python tools/parity_probe.py 400 600 $seed fp16 representative 2>&1 | grep -v Warn | grep "==\|full path\|ORACLE\|latent_rel"; done > gpurun_out/r3g/parity.log 2>&1
cat gpurun_out/r3g/parity.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train > gpurun_out/r3g/bench.log 2>&1; tail -1 gpurun_out/r3g/bench.log | cut -c1-400
