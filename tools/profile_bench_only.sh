#!/bin/bash
# The first block of tools/profile_round.sh alone: the bench line, the rocprofv3 kernel-trace summary of the same command on one stream,
# the generated agreement check and the batch-4 line.   usage: tools/profile_bench_only.sh r06   -> gpurun_out/<tag>_*
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o b -- python bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-train --no-power > gpurun_out/${tag}_bench_prof.log 2>&1
python tools/rocpd_stats.py $(ls gpurun_out/prof_$tag/*/b_results.db gpurun_out/prof_$tag/b_results.db 2>/dev/null | head -1) > gpurun_out/${tag}_bench_kernel_stats.txt 2>&1
python tools/agreement_check.py gpurun_out/${tag}_bench_line.json gpurun_out/${tag}_bench_kernel_stats.txt > gpurun_out/${tag}_agreement.txt 2>&1
python bench.py --batch 4 --steps 20 --warmup 5 --no-train --no-cpu-baseline > gpurun_out/${tag}_bench_line_b4.json 2>> gpurun_out/${tag}_bench.err
rm -rf gpurun_out/prof_$tag
ls -la gpurun_out | grep $tag
