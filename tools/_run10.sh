cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_dcn.py tests/test_gpu_precision.py tests/test_gpu_determinism.py -m gpu -q 2>&1 | grep -v Warn | tail -8
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['rooflines'][0]
print(d['value'], d['ms_per_step'], 'attn', r['ms_per_launch'], r['frac'], 'conv', c['ms_per_step'], c['frac'])"
