cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_precision.py tests/test_gpu_train.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r3d/pytest.log
tail -4 gpurun_out/r3d/pytest.log
timeout 900 bash tools/attn_ab.sh "-DATTNKV_MRUN_IN_ACC=0" 3 > gpurun_out/r3d/attn_ab.log 2>&1
cat gpurun_out/r3d/attn_ab.log
timeout 300 python tools/probes/small_ops.py stage2 > gpurun_out/r3d/small_ops_stage2.log 2>&1
timeout 300 python tools/probes/infer_small_ops.py > gpurun_out/r3d/small_ops_infer.log 2>&1
tail -30 gpurun_out/r3d/small_ops_infer.log
