#!/bin/bash
# HBM traffic of the kbench kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slots),
# counters only.  usage: tools/pmc_traffic.sh <tag> <kbench args>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
KB_REPS=2 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/tr_f_$tag -o p -- python tools/kbench.py "$@" > gpurun_out/tr_$tag.log 2>&1
KB_REPS=2 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/tr_w_$tag -o p -- python tools/kbench.py "$@" >> gpurun_out/tr_$tag.log 2>&1
python tools/rocpd_pmc.py gpurun_out/tr_f_$tag/p_results.db gpurun_out/tr_w_$tag/p_results.db > gpurun_out/traffic_$tag.txt 2>&1
cat gpurun_out/traffic_$tag.txt
