#!/bin/bash
# usage: tools/ablate.sh <file.hip> <kbench case> <DEF1|-flag...> ...   -- rebuilds with each variant and times the kernel.
# An argument that starts with '-' is passed to hipcc verbatim (e.g. "-fno-slp-vectorize -DATTN_PIPELINED=1"), anything else
# becomes -D<arg>.
f=$1; kcase=$2; shift 2
for d in "" "$@"; do
  touch glare_amd/csrc/$f
  case "$d" in -*) defs="$d" ;; "") defs="" ;; *) defs="-D$d" ;; esac
  GLARE_DEFS="$defs" python glare_amd/csrc/build.py > /dev/null 2>&1 || echo build failed
  echo "== ${d:-baseline}"; python tools/kbench.py $kcase 2>&1 | grep -v amdgpu.ids
done
touch glare_amd/csrc/$f; python glare_amd/csrc/build.py > /dev/null 2>&1
