#!/bin/bash
# usage: tools/ablate.sh <file.hip> <kbench case> <DEF1> <DEF2> ...   -- rebuilds with each -D and times the kernel
f=$1; kcase=$2; shift 2
for d in "" "$@"; do
  touch glare_amd/csrc/$f
  GLARE_DEFS="${d:+-D$d}" python glare_amd/csrc/build.py > /dev/null 2>&1 || echo build failed
  echo "== ${d:-baseline}"; python tools/kbench.py $kcase 2>&1 | grep -v amdgpu.ids
done
touch glare_amd/csrc/$f; python glare_amd/csrc/build.py > /dev/null 2>&1
