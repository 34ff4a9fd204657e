#!/bin/bash
# One-call profile of a round on the GPU box: the bench line, its rocprofv3 kernel-trace summary, the PMC traffic / SQ counters of
# the kbench kernels (counters in their own runs, kernel trace only -- never with --sys-trace / hip / hsa domains) and the training
# steps.  usage: tools/profile_round.sh r02      -> gpurun_out/<tag>_*.txt / .json (copy the ones to keep into profiles/)
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o b -- python bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-train --no-power > gpurun_out/${tag}_bench_prof.log 2>&1
python tools/rocpd_stats.py $(ls gpurun_out/prof_$tag/*/b_results.db gpurun_out/prof_$tag/b_results.db 2>/dev/null | head -1) > gpurun_out/${tag}_bench_kernel_stats.txt 2>&1
python tools/agreement_check.py gpurun_out/${tag}_bench_line.json gpurun_out/${tag}_bench_kernel_stats.txt > gpurun_out/${tag}_agreement.txt 2>&1   # the README's "agreement check", generated
python bench.py --batch 4 --steps 20 --warmup 5 --no-train --no-cpu-baseline > gpurun_out/${tag}_bench_line_b4.json 2>> gpurun_out/${tag}_bench.err   # the per-rank workload of BASELINE configs[2] (32 images over 8 GPUs)
{
  echo "# rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, counters only), MI355X, B=8, $tag"
  echo "# units: KB per dispatch as reported; gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read stream -> x2 (MI355X_MICROARCH.md)"
  for k in attn conv gn dcn; do echo "== $k"; bash tools/pmc_traffic.sh ${tag}_$k $k 2>&1 | grep -v "amdgpu.ids"; done
} > gpurun_out/${tag}_pmc_traffic.txt
{
  echo "# rocprofv3 --pmc SQ counters (two passes of 8), kbench kernels at B=8, $tag"
  for k in attn conv dcn; do echo "== $k"; bash tools/pmc_kernel.sh ${tag}_$k $k 2>&1 | grep -v "amdgpu.ids"; done
} > gpurun_out/${tag}_pmc_kernels.txt
python tools/kbench.py attn conv convsplit gn dcn vq wgrad attnbwd > gpurun_out/${tag}_kbench.txt 2>&1
python tools/kbench.py flow winograd > gpurun_out/${tag}_kbench_flow_winograd.txt 2>&1        # round 6: the fused flow step against the four-launch form; the Winograd products priced
python tools/probes/infer_ops_by_line.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_infer_stock_ops.txt   # stock torch ops of one steady-state inference step, by source line
for st in stage2 stage3; do python tools/probes/small_ops.py $st 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_train_${st}_stock_ops.txt; done
python tools/probes/power_clock_probe.py 4 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${tag}_power_clock.txt   # the power wall: random vs all-zero operands
{
  echo "# rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum -- python tools/kbench.py dcn   (one shape per run)"
  for c in 128 256; do
    KB_DCN=$c KB_REPS=2 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum \
      -d gpurun_out/pmc_${tag}_dcnta$c -o p -- python tools/kbench.py dcn > gpurun_out/pmc_${tag}_dcnta$c.log 2>&1
    echo "== C = $c"; python tools/rocpd_pmc.py gpurun_out/pmc_${tag}_dcnta$c/p_results.db 2>&1 | grep -A7 "dcn_fwd"
  done
} > gpurun_out/${tag}_pmc_dcn_ta.txt 2>&1          # the DCN forward's texture-path counters (what binds it)
bash tools/pmc_shapes.sh $tag > /dev/null 2>&1          # per-SHAPE traffic of the conv / DCN launches -> ${tag}_pmc_shapes.json
# end-to-end parity: 12 scenes of the default path (tools/parity_scenes.py), then the precision ladder on three of them -- round 3's
# single-pass fp16 path (GLARE_FP32_CLASS=0) and bf16 -- and the single-pass DCN as an A/B
{
  python tools/parity_scenes.py 400 600 2>&1 | grep -v "Warn\|amdgpu.ids"
  GLARE_DCN_SINGLE_PASS=1 python tools/parity_scenes.py 400 600 11 13 15 2>&1 | grep -v "Warn\|amdgpu.ids"
  GLARE_FP32_CLASS=0 python tools/parity_scenes.py 400 600 11 13 15 2>&1 | grep -v "Warn\|amdgpu.ids"
  PARITY_PRECISION=bf16 python tools/parity_scenes.py 400 600 11 13 15 2>&1 | grep -v "Warn\|amdgpu.ids"
  echo; echo "# held-out scenes and a second trained-like weight set"
  python tools/parity_scenes.py 400 600 101 102 103 104 105 106 107 108 109 110 111 112 2>&1 | grep -v "Warn\|amdgpu.ids"
  PARITY_WEIGHT_SEED=1 python tools/parity_scenes.py 400 600 11 12 13 101 102 103 2>&1 | grep -v "Warn\|amdgpu.ids"
  echo; echo "# a THIRD trained-like weight set (PARITY_WEIGHT_SEED=2: another codebook, ActNorm states and filters), 6 scenes, and 6 more held-out scenes of the first (42 full-size scenes on the default path in all)"
  PARITY_WEIGHT_SEED=2 python tools/parity_scenes.py 400 600 11 12 13 101 102 103 2>&1 | grep -v "Warn\|amdgpu.ids"
  python tools/parity_scenes.py 400 600 201 202 203 204 205 206 2>&1 | grep -v "Warn\|amdgpu.ids"
} > gpurun_out/${tag}_parity_table.txt 2>&1
for st in stage2 stage3; do
  python tools/train_bench.py $st 20 graph > gpurun_out/${tag}_train_${st}_graph.txt 2>&1
  python tools/train_bench.py $st 10 > gpurun_out/${tag}_train_$st.txt 2>&1
  TRAIN_PRECISION=bf16 python tools/train_bench.py $st 10 >> gpurun_out/${tag}_train_$st.txt 2>&1      # bf16 beside the default (fp16 AMP)
  rocprofv3 --kernel-trace --stats -d gpurun_out/proft_${tag}_$st -o t -- python tools/train_bench.py $st 5 > /dev/null 2>&1
  python tools/rocpd_stats.py $(ls gpurun_out/proft_${tag}_$st/*/t_results.db gpurun_out/proft_${tag}_$st/t_results.db 2>/dev/null | head -1) > gpurun_out/${tag}_train_${st}_kernel_stats.txt 2>&1
  python tools/train_bench.py $st 1 flops 2>&1 | grep flops_per_step > gpurun_out/${tag}_train_${st}_flops.txt
  python tools/train_roofline.py gpurun_out/${tag}_train_${st}_kernel_stats.txt gpurun_out/${tag}_train_${st}_flops.txt 7 > gpurun_out/${tag}_train_${st}_roofline.txt 2>&1
done
rm -rf gpurun_out/prof_$tag gpurun_out/proft_${tag}_* gpurun_out/tr_f_* gpurun_out/tr_w_* gpurun_out/pmc_${tag}_*/ gpurun_out/pmc2_${tag}_*/ 2>/dev/null
ls -la gpurun_out | grep $tag
