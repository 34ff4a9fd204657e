#!/bin/bash
# Runs on the GPU box: rebuilds csrc/attn_bwd.hip with each timing ablation (-DATTNBWD_ABL=<mask>; results are wrong on purpose)
# and times the fused attention backward at the stage-2 latent (tools/kbench.py attnbwd).
cd "$(dirname "$0")/.."
for m in 0 1 2 4 8 15; do
  touch glare_amd/csrc/attn_bwd.hip
  GLARE_DEFS="-DATTNBWD_ABL=$m" python glare_amd/csrc/build.py > /dev/null 2>&1 || { echo "ABL=$m: build failed"; continue; }
  echo -n "ABL=$m  "; KB_REPS=10 python tools/kbench.py attnbwd 2>&1 | grep attnbwd | sed 's/;.*//'
done
touch glare_amd/csrc/attn_bwd.hip; python glare_amd/csrc/build.py > /dev/null 2>&1
