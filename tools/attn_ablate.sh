#!/bin/bash
# Runs on the GPU box: rebuilds csrc/attn.hip with each ablation / variant macro and times the attention kernels
# (tools/kbench.py attn); the ATTNKV_PROFILE build also prints the per-phase cycle split of one wave.
cd "$(dirname "$0")/.."
run() {   # $1 = label, $2 = GLARE_DEFS, $3 = extra env
  touch glare_amd/csrc/attn.hip
  GLARE_DEFS="$2" python glare_amd/csrc/build.py > /dev/null 2>&1 || { echo "== $1: build failed"; return; }
  echo "== $1   [$2]"
  env $3 KB_REPS=10 python tools/kbench.py attn 2>&1 | grep attnkv
}
run baseline ""
run profile "-DATTNKV_PROFILE" KB_PROF=1
for v in "$@"; do run "$v" "$v"; done
touch glare_amd/csrc/attn.hip; python glare_amd/csrc/build.py > /dev/null 2>&1
