#!/bin/bash
# Runs on the GPU box: A/B of two builds of csrc/attn.hip (default vs the macros in $1), alternated $2 times (default 3) so the
# clocks both see are the same; prints the shared-K/V attention time of each.
cd "$(dirname "$0")/.."
mkdir -p /tmp/ab
build() { touch glare_amd/csrc/attn.hip; GLARE_DEFS="$1" python glare_amd/csrc/build.py > /dev/null 2>&1 && cp glare_amd/libglare_hip.so "$2"; }
build "" /tmp/ab/a.so || { echo "build A failed"; exit 1; }
build "$1" /tmp/ab/b.so || { echo "build B failed"; exit 1; }
for i in $(seq 1 "${2:-3}"); do
  cp /tmp/ab/a.so glare_amd/libglare_hip.so; echo -n "A default     "; KB_REPS=10 python tools/kbench.py attn 2>&1 | grep attnkv
  cp /tmp/ab/b.so glare_amd/libglare_hip.so; echo -n "B $1  "; KB_REPS=10 python tools/kbench.py attn 2>&1 | grep attnkv
done
cp /tmp/ab/a.so glare_amd/libglare_hip.so
