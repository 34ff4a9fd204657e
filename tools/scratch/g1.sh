python -m pytest tests/test_gpu_conv.py  tests/test_gpu_precision.py tests/test_gpu_golden.py tests/test_gpu_kernels.py tests/test_gpu_train.py -q -m gpu 2>&1 | tail -15
python tools/kbench.py conv 2>&1 | tail -12
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>&1 | tail -1
