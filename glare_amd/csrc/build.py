"""Build libglare_hip.so (gfx950 only) from the .hip sources next to this file.

Used by __graft_entry__.build() and runnable directly: `python glare_amd/csrc/build.py`.
hipcc cross-compiles without a GPU; objects are rebuilt only when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), "libglare_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-fno-gpu-rdc"] + os.environ.get("GLARE_DEFS", "").split()  # GLARE_DEFS: ablation switches (tools/)
# vq.hip carries the bit-exactness contract: no fused contraction the source does not spell.
# dcn.hip: no compiler-formed packed-fp32 (v_pk_*_f32) ops next to its MFMAs -- see the note in dcn_fwd_fast_kernel.
PER_FILE = {"vq.hip": ["-ffp-contract=off"], "dcn.hip": ["-fno-slp-vectorize"], "dcn_bwd.hip": ["-fno-slp-vectorize"]}


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "glare_hip.h"))
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            jobs.append((src, obj, PER_FILE.get(s, [])))

    def cc(job):
        src, obj, extra = job
        cmd = [HIPCC] + COMMON + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return (src, r.returncode, r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        results = list(ex.map(cc, jobs))
    for src, rc, out in results:
        if out.strip() and verbose:
            print(out)
        if rc != 0:
            raise RuntimeError("hipcc failed on %s\n%s" % (src, out))
    if jobs or force or not os.path.exists(OUT):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed\n" + r.stdout + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
