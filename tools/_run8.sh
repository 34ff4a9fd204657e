cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
for i in 1 2; do
for hl in 1 0; do
  echo -n "hilo=$hl  "; GLARE_HILO_STREAM=$hl timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --breakdown 2>gpurun_out/r3h/err_$hl.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['rooflines'][0]
print(d['value'], d['ms_per_step'], 'attn', r['ms_per_launch'], r['frac'], 'conv', c['ms_per_step'], c['frac'])"; grep breakdown gpurun_out/r3h/err_$hl.log
done; done
echo -n "bf16    "; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --precision bf16 --breakdown 2>gpurun_out/r3h/err_bf.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['rooflines'][0]
print(d['value'], d['ms_per_step'], 'attn', r['ms_per_launch'], r['frac'], 'conv', c['ms_per_step'], c['frac'])"; grep breakdown gpurun_out/r3h/err_bf.log
