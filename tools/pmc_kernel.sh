#!/bin/bash
# Collects SQ PMC counters for one kbench case in its own rocprofv3 run (counters only, no tracing
# domains besides the kernel trace), writes gpurun_out/pmc_<tag>.txt.  usage: tools/pmc_kernel.sh <tag> <kbench args>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
KB_REPS=2 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
  -d gpurun_out/pmc_$tag -o p -- python tools/kbench.py "$@" > gpurun_out/pmc_$tag.log 2>&1
KB_REPS=2 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_UNALIGNED_STALL \
  -d gpurun_out/pmc2_$tag -o p -- python tools/kbench.py "$@" >> gpurun_out/pmc_$tag.log 2>&1
python tools/rocpd_pmc.py gpurun_out/pmc_$tag/p_results.db gpurun_out/pmc2_$tag/p_results.db > gpurun_out/pmc_$tag.txt 2>&1
cat gpurun_out/pmc_$tag.txt
