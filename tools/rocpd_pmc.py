#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in one or more rocprofv3 rocpd databases."""
import collections
import re
import sqlite3
import sys


def main(paths):
    for path in paths:
        db = sqlite3.connect(path)
        cur = db.cursor()
        try:
            rows = cur.execute("select * from counters_collection limit 1").fetchall()
            cols = [d[0] for d in cur.description]
        except Exception as e:
            print("no counters_collection in", path, e)
            continue
        ncol = [c for c in cols if "kernel_name" in c or c == "name"]
        kcol = ncol[0] if ncol else "kernel_name"
        q = "select %s, counter_name, value, dispatch_id from counters_collection" % kcol
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(set)
        for name, cname, val, did in cur.execute(q):
            k = re.sub(r"\(anonymous namespace\)::", "", name)[:60]
            agg[k][cname] += val
            cnt[k].add(did)
        for k, d in agg.items():
            if "at::native" in k or "rocclr" in k or "pack_weight" in k:
                continue
            n = max(1, len(cnt[k]))
            print("%s  (%d dispatches)" % (k, n))
            for c, v in sorted(d.items()):
                print("    %-28s %16.0f per dispatch" % (c, v / n))


if __name__ == "__main__":
    main(sys.argv[1:])
