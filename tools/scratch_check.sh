#!/bin/bash
# usage: tools/scratch_check.sh <file.hip> [extra hipcc flags]   -- VGPRs / scratch / occupancy of every kernel of one translation unit (fp16 build; no GPU needed)
root=$(cd "$(dirname "$0")/.." && pwd)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DGLARE_ACT_F16 "$@" -I $root/include -I $root/glare_amd/csrc \
  -Rpass-analysis=kernel-resource-usage -c $root/glare_amd/csrc/$f -o /tmp/scratch_check.o 2>&1 \
  | grep -E "error|Name:|VGPRs:|ScratchSize|Occupancy" | sed -e 's/.*remark: //' -e 's/ \[-Rpass.*//' -e 's/.*Name: //' | paste - - - -
