#!/usr/bin/env python
"""Per-kernel-family roofline lines of a training step: algorithmic FLOPs per step (tools/train_bench.py <stage> <n> flops) over
the rocprofv3 kernel time per step of the kernels that do them (tools/rocpd_stats.py table of an n_steps-step run).
    python tools/train_roofline.py <kernel_stats.txt> <train_bench_flops.txt> <profiled steps incl. warm-up>"""
import re
import sys

PEAK = 2500.0   # dense bf16 MFMA peak, TFLOP/s (MI355X_MICROARCH.md)
FAMILIES = {"attn fwd": [r"^attn_fwd_kernel", r"^attn_kv_fwd_kernel", r"^attn_combine_kernel"], "attn bwd": [r"^attn_bwd_kernel", r"^attn_bwd_dsum"],
            "gemm_nt": [r"^gemm_nt_kernel"], "wgrad k3": [r"^wgrad_kernel<3>"], "wgrad k1": [r"^wgrad_kernel<1>"],
            "conv k3": [r"^conv_igemm k3", r"^conv_igemm k2"], "conv k1": [r"^conv_igemm k1", r"^conv1x1_ws_kernel"]}


def main(stats_path, flops_path, steps):
    times = {}
    for line in open(stats_path):
        m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            times[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    flops = {}
    for line in open(flops_path):
        t = line.split()
        if t and t[0] == "flops_per_step":
            flops[" ".join(t[1:-1])] = float(t[-1])
    print("# family: algorithmic TFLOP per step / kernel ms per step (rocprofv3, %d steps) = TFLOP/s, fraction of the %.0f TFLOP/s bf16 peak"
          % (steps, PEAK))
    for fam, pats in FAMILIES.items():
        ms = sum(t for name, (n, t) in times.items() if any(re.search(p, name) for p in pats)) / steps
        calls = sum(n for name, (n, t) in times.items() if any(re.search(p, name) for p in pats)) / steps
        if fam in flops and ms > 0:
            tf = flops[fam] / 1e12
            print("%-9s %8.4f TFLOP  %7.3f ms  %5.0f launches  %7.1f TFLOP/s  frac %.3f" % (fam, tf, ms, calls, tf / ms * 1e3, tf / ms * 1e3 / PEAK))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
