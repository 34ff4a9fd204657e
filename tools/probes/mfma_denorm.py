#!/usr/bin/env python
"""Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL operands?  (The lo halves of hi / lo operand pairs live there: a filter of
magnitude 0.02 has a remainder of ~1e-5 < 6.1e-5.)  1x1 conv, Cin = 32, all-ones filter on a constant subnormal input, and the
mirrored case (subnormal filter, unit input); fp32 output."""
import torch

from glare_amd import ops

with ops.use_precision("fp16"):
    dev = torch.device("cuda:0")
    for name, xv, wv in (("subnormal activation", 2.0 ** -20, 1.0), ("subnormal filter", 1.0, 2.0 ** -20),
                         ("both normal", 2.0 ** -10, 2.0 ** -10), ("product of two subnormal-free small", 2.0 ** -14, 2.0 ** -14)):
        x = torch.full((1, 8, 32, 32), xv, dtype=torch.float16, device=dev)
        w = torch.full((32, 32, 1, 1), wv, dtype=torch.float32, device=dev)
        pc = ops.PackedConv(w)
        pc.w16 = None
        y = ops.conv2d(x, pc, out_mode=ops.OUT_NHWC_F32)
        torch.cuda.synchronize()
        print("%-40s expected %.6e got %.6e" % (name, 32 * xv * wv, float(y[0, 0, 0, 0])))
