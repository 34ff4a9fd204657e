#!/bin/bash
# A/B of two builds of ONE source file on one box (clocks differ between boxes by up to 8 %): rebuilds with each set of extra compiler
# flags (GLARE_DEFS, passed to hipcc verbatim) and times a kbench case.   usage: tools/ab_build.sh <file.hip|header.h> <kbench case> "<flags>" ...
# The first build is always the plain one.  (Rounds 1-4 kept compiled-out timing ablations in the kernels for this; round 5 removed them --
# the switches that remain are real code paths, and an experiment lives on a branch, not in the product source.)
f=$1; kcase=$2; shift 2
for d in "" "$@"; do
  touch glare_amd/csrc/$f
  GLARE_DEFS="$d" python glare_amd/csrc/build.py > /dev/null 2>&1 || echo build failed
  echo "== ${d:-plain build}"; python tools/kbench.py $kcase 2>&1 | grep -v amdgpu.ids
done
touch glare_amd/csrc/$f; python glare_amd/csrc/build.py > /dev/null 2>&1
