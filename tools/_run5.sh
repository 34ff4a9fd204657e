cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -m gpu -q -x -k "two_rank or attention or stage2" 2>&1 | tail -15 > gpurun_out/r3e/pytest.log
tail -4 gpurun_out/r3e/pytest.log
timeout 900 bash tools/attn_ab.sh "-DATTNKV_MAX3=0" 3 > gpurun_out/r3e/attn_ab.log 2>&1
cat gpurun_out/r3e/attn_ab.log
timeout 300 python tools/probes/small_ops.py stage2 > gpurun_out/r3e/small_ops_stage2.log 2>&1
timeout 300 python tools/probes/infer_small_ops.py > gpurun_out/r3e/small_ops_infer.log 2>&1
tail -5 gpurun_out/r3e/small_ops_infer.log
timeout 600 python tools/train_bench.py stage2 10 graph > gpurun_out/r3e/train2.log 2>&1; tail -1 gpurun_out/r3e/train2.log
