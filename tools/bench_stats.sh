#!/bin/bash
# rocprofv3 kernel-trace summary of a short bench run -> gpurun_out/<tag>_bench_kernel_stats.txt (+ the bench line)
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_prof.log 2>&1
python tools/rocpd_stats.py $(ls gpurun_out/prof_$tag/*/b_results.db gpurun_out/prof_$tag/b_results.db 2>/dev/null | head -1) > gpurun_out/${tag}_bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_$tag
head -16 gpurun_out/${tag}_bench_kernel_stats.txt
grep -o '"value": [0-9.]*' gpurun_out/${tag}_bench_prof.log
