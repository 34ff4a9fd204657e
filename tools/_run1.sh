set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3a/pytest.log
for prec in bf16 fp16; do for reg in adversarial representative; do
  timeout 600 python tools/parity_probe.py 400 600 11 $prec $reg > gpurun_out/r3a/probe_${prec}_${reg}.log 2>&1
done; done
timeout 600 python bench.py --steps 10 --warmup 3 --precision bf16 > gpurun_out/r3a/bench_bf16.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --precision fp16 --no-cpu-baseline > gpurun_out/r3a/bench_fp16.log 2>&1
tail -3 gpurun_out/r3a/*.log
