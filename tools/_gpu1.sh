set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_split_conv.py tests/test_gpu_determinism.py -x -q 2>&1 | tail -5) > gpurun_out/rs_tests.log 2>&1
python tools/path_hash.py 2 > gpurun_out/rs_hash1.log 2>&1
KB_REPS=20 python tools/kbench.py conv convsplit > gpurun_out/rs_kb1.log 2>&1
touch glare_amd/csrc/conv_igemm_kernel.h; GLARE_DEFS="-DGLARE_ROW_SKIP=0" python glare_amd/csrc/build.py > /dev/null 2>&1 || echo build failed
python tools/path_hash.py 2 > gpurun_out/rs_hash0.log 2>&1
KB_REPS=20 python tools/kbench.py conv convsplit > gpurun_out/rs_kb0.log 2>&1
python bench.py --steps 10 --warmup 3 --no-train --no-cpu-baseline --no-power > gpurun_out/rs_bench0.log 2>&1
touch glare_amd/csrc/conv_igemm_kernel.h; python glare_amd/csrc/build.py > /dev/null 2>&1 || echo build failed
KB_REPS=20 python tools/kbench.py conv convsplit > gpurun_out/rs_kb1b.log 2>&1
python bench.py --steps 10 --warmup 3 --no-train --no-cpu-baseline --no-power > gpurun_out/rs_bench1.log 2>&1
cmp gpurun_out/rs_hash0.log gpurun_out/rs_hash1.log && echo HASH_IDENTICAL
tail -3 gpurun_out/rs_tests.log
