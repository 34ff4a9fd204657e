cd $GRAFT_REPO_ROOT
for b in 2 4 8 16; do
  echo -n "batch=$b  "; timeout 600 python bench.py --steps 10 --warmup 3 --batch $b --no-cpu-baseline --no-train 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['rooflines'][0]
print(d['value'], d['ms_per_step'], 'attn', r['ms_per_launch'], r['frac'], 'conv', c['ms_per_step'], c['frac'])"
done
