cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "graphed or fall_back or two_rank or adam_skips or reference_crop" 2>&1 | tail -6 > gpurun_out/r3f/pytest.log
tail -3 gpurun_out/r3f/pytest.log
timeout 1500 bash tools/ablate.sh dcn.hip dcn DCN_ABL=1 DCN_ABL=2 DCN_ABL=4 DCN_ABL=8 DCN_ABL=3 DCN_ABL=11 > gpurun_out/r3f/dcn_ablate.log 2>&1
cat gpurun_out/r3f/dcn_ablate.log
