"""Is the attention / conv K loop limited by the chip's power management rather than by its schedule?  The SAME launch is repeated for a
few seconds on random operands and on all-zero operands (identical instruction stream and memory traffic, far less switching power)
while `rocm-smi` samples the shader clock and the socket power in the background.  Measured on MI355X (round 4): see DESIGN.md section 3.

    python tools/probes/power_clock_probe.py [seconds per phase]"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glare_amd import ops  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
samples, phase, stop = [], ["idle"], [False]


def sampler():
    while not stop[0]:
        try:
            r = subprocess.run(["rocm-smi", "-c", "-P", "--json"], capture_output=True, text=True, timeout=10)
            d = json.loads(r.stdout)
            card = next(iter(d.values()))
            sclk = next((v for k, v in card.items() if k.lower().startswith("sclk")), "")
            pw = next((v for k, v in card.items() if "power" in k.lower()), "")
            samples.append((phase[0], sclk, pw))
        except Exception as e:      # the probe reports what it can read
            samples.append((phase[0], "err", repr(e)[:80]))
        time.sleep(0.2)


def run(tag, fn, flop):
    phase[0] = tag
    fn()
    torch.cuda.synchronize()
    t0, n = time.time(), 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    while time.time() - t0 < SECS:
        for _ in range(10):
            fn()
        n += 10
        torch.cuda.synchronize()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    mine = [x for x in samples if x[0] == tag][3:]      # skip the ramp
    print("%-28s %.3f ms  %5.0f TFLOP/s   sclk %s   power %s" % (tag, ms, flop / ms / 1e9,
          sorted(set(x[1] for x in mine))[:6], sorted(set(x[2] for x in mine))[:6]), flush=True)
    phase[0] = "idle"
    time.sleep(1.0)


def main():
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    B, N, C = 8, 105 * 155, 512
    with ops.use_precision("fp16"):
        x = torch.randn(B, N, C, device="cuda").half()
        q = (torch.randn(B, N, C, device="cuda") * 0.3).half()
        out = torch.empty_like(x)
        xz, qz = torch.zeros_like(x), torch.zeros_like(q)
        fl = 4.0 * B * N * N * C
        run("attention random", lambda: ops.attention_kv512(q, x, N, out=out, key_splits=1), fl)
        run("attention all-zero", lambda: ops.attention_kv512(qz, xz, N, out=out, key_splits=1), fl)
        run("attention random again", lambda: ops.attention_kv512(q, x, N, out=out, key_splits=1), fl)
        h, w, c = 210, 310, 512
        xc = torch.randn(B, h, w, c, device="cuda").half()
        wt = torch.randn(c, c, 3, 3, device="cuda") * 0.02
        pc, pz = ops.PackedConv(wt, torch.zeros(c, device="cuda")), ops.PackedConv(torch.zeros_like(wt), torch.zeros(c, device="cuda"))
        oc = torch.empty(B, h, w, c, dtype=torch.float16, device="cuda")
        xcz = torch.zeros_like(xc)
        fl = 2.0 * B * h * w * c * c * 9
        run("conv 512@half random", lambda: ops.conv2d(xc, pc, out=oc), fl)
        run("conv 512@half all-zero", lambda: ops.conv2d(xcz, pz, out=oc), fl)
        # one operand zero: which side carries the switching power?
        run("conv zero activations", lambda: ops.conv2d(xcz, pc, out=oc), fl)
        run("conv zero filter", lambda: ops.conv2d(xc, pz, out=oc), fl)
    stop[0] = True
    print("idle samples:", sorted(set((x[1], x[2]) for x in samples if x[0] == "idle"))[:4])


if __name__ == "__main__":
    main()
