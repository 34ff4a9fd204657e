#!/usr/bin/env python
"""ONE launch shape, launched a few times: the workload of tools/pmc_shapes.sh's per-shape counter passes.
    python tools/pmc_shape_probe.py <key>      key in SHAPES below
Prints the algorithmic bytes of one launch (inputs + outputs + filter) as `ALGO_BYTES <n>`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glare_amd import ops  # noqa: E402

B = 8
CONV = {"128_full": (128, 128, 420, 620), "256_half": (256, 256, 210, 310), "512_q": (512, 512, 105, 155)}


def main(key):
    dev = "cuda"
    kind, shape = key.split("_", 1)
    with ops.use_precision("fp16"):
        if kind in ("conv", "conv3"):
            ci, co, h, w = CONV[shape]
            wt = torch.randn(co, ci, 3, 3, device=dev) * 0.02
            if kind == "conv3":      # the fp32-class form: pair in, pair out
                x = ops.split_hilo(torch.randn(B, h, w, ci, device=dev))
                pc = ops.PackedConv(wt, torch.zeros(co, device=dev), split=3)
                out = torch.empty(B, h, w, co, dtype=torch.float16, device=dev)
                out._lo = torch.empty_like(out)
                fn = lambda: ops.conv2d(x, pc, out=out, hilo=True)
                algo = 2 * 2.0 * B * h * w * ci + 2 * 2.0 * B * h * w * co + 3 * 2.0 * 9 * ci * co
            else:
                x = torch.randn(B, h, w, ci, device=dev).half()
                pc = ops.PackedConv(wt, torch.zeros(co, device=dev))
                out = torch.empty(B, h, w, co, dtype=torch.float16, device=dev)
                fn = lambda: ops.conv2d(x, pc, out=out)
                algo = 2.0 * B * h * w * ci + 2.0 * B * h * w * co + 2.0 * 9 * ci * co
        elif kind == "dcn":
            c, h, w = {"128": (128, 420, 620), "256": (256, 210, 310)}[shape]
            x = torch.randn(B, h, w, c, device=dev).half()
            plane = (h * w + 63) // 64 * 64
            om = torch.randn(B, 108, plane, device=dev)
            pd = ops.PackedDcn(torch.randn(c, c, 3, 3, device=dev) * 0.02, torch.zeros(c, device=dev), 4)
            # the pipeline's call since round 6 (glare_mdcn_forward_nhwc_fused): 16-bit output + per-tile sums; out: 2 B / channel
            fn = lambda: ops.mdcn_forward_nhwc_fused(x, om, pd, x_off=0, C=c, out16=True, want_sums=True)
            algo = 2.0 * B * h * w * c + 4.0 * B * h * w * 108 + 2.0 * B * h * w * c + 4.0 * c * c * 9
        else:
            raise SystemExit("unknown key " + key)
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    print("ALGO_BYTES %d" % algo)


if __name__ == "__main__":
    main(sys.argv[1])
