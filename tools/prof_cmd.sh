#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of one command, summarised per kernel.  usage: tools/prof_cmd.sh <tag> <command...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o p -- "$@" > gpurun_out/${tag}_cmd.log 2>&1
python tools/rocpd_stats.py $(ls gpurun_out/prof_$tag/*/p_results.db gpurun_out/prof_$tag/p_results.db 2>/dev/null | head -1) > gpurun_out/${tag}_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_$tag
head -${PROF_LINES:-25} gpurun_out/${tag}_kernel_stats.txt | cut -c1-150
