#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) per kernel: calls, total/avg/min/max
duration and share of GPU time -- the same table `rocprofv3 --stats` prints.  Kernel template
arguments are decoded so that the conv variants are distinguishable.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r01_bench_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"conv_igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (true|false))?>", name)
    if m:
        ks, st, mt, nt, wm, wn, kst = map(int, m.groups()[:7])
        return "conv_igemm k%d s%d TN%d%s" % (ks, st, wn * nt * 32, " hi/lo" if m.group(8) == "true" else "")
    name = re.sub(r"\(.*\)$", "", name)
    return name[:90]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % namecol).fetchall()
    agg = {}
    for name, s, e in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("%-62s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-62s %7d %12.3f %10.1f %10.1f %10.1f %6.2f" % (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3,
                                                              100.0 * a[1] / tot))
    print("%-62s %7d %12.3f" % ("TOTAL", sum(a[0] for a in agg.values()), tot / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
