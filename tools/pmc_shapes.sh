#!/bin/bash
# Per-SHAPE HBM traffic of the conv / DCN launches (bench.py `rooflines[*].traffic`): for every key ONE launch shape under
# rocprofv3 --pmc FETCH_SIZE and, in a SEPARATE run, WRITE_SIZE (counters only, kernel trace only), then
# gpurun_out/<tag>_pmc_shapes.json = {key: {fetch_kb, write_kb, bytes = (2 x fetch + write) x 1024, algorithmic_bytes, ratio}}.
# usage: tools/pmc_shapes.sh r04     (copy the json into profiles/)
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/${tag}_pmc_shapes.txt
: > $out
for key in conv_128_full conv_256_half conv_512_q conv3_128_full conv3_256_half conv3_512_q dcn_128 dcn_256; do
  echo "== $key" >> $out
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/ps_f_$key -o p -- python tools/pmc_shape_probe.py $key 2>/dev/null | grep ALGO_BYTES >> $out
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/ps_w_$key -o p -- python tools/pmc_shape_probe.py $key > /dev/null 2>&1
  python tools/rocpd_pmc.py $(ls gpurun_out/ps_f_$key/*/p_results.db gpurun_out/ps_f_$key/p_results.db 2>/dev/null | head -1) \
                            $(ls gpurun_out/ps_w_$key/*/p_results.db gpurun_out/ps_w_$key/p_results.db 2>/dev/null | head -1) 2>&1 | grep -A1 "conv_igemm_kernel\|dcn_fwd" >> $out
  rm -rf gpurun_out/ps_f_$key gpurun_out/ps_w_$key
done
python - <<PY
import json, re
out = {}
key = None
for line in open("$out"):
    line = line.rstrip()
    if line.startswith("== "):
        key = line[3:]; out[key] = {}
    elif line.startswith("ALGO_BYTES"):
        out[key]["algorithmic_bytes"] = int(line.split()[1])
    else:
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+) per dispatch", line)
        if m and key:
            out[key]["fetch_kb" if m.group(1) == "FETCH_SIZE" else "write_kb"] = int(m.group(2))
for k, v in out.items():
    if "fetch_kb" in v and "write_kb" in v:
        v["bytes"] = int((2 * v["fetch_kb"] + v["write_kb"]) * 1024)
        if v.get("algorithmic_bytes"):
            v["ratio_to_algorithmic"] = round(v["bytes"] / v["algorithmic_bytes"], 3)
json.dump(out, open("gpurun_out/${tag}_pmc_shapes.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
