"""Run-to-run determinism of the big kernels at production shapes: every launch is repeated and compared bit for bit
(no kernel on the inference path uses atomics, so any difference is a hazard or a race)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glare_amd import _lib, ops  # noqa: E402

REPS = int(os.environ.get("DET_REPS", "6"))
DEV = "cuda"


def check(name, fn):
    first = fn().clone()
    bad = 0
    for _ in range(REPS):
        out = fn()
        bad += int((out != first).sum())
    print("%-44s %s" % (name, "identical over %d launches" % (REPS + 1) if bad == 0 else "DIFFERS: %d elements" % bad), flush=True)
    return bad


def main():
    g = torch.Generator().manual_seed(0)
    total = 0
    B = 8
    for name, ci, co, h, w, k, ups in (("conv 128->128 3x3 420x620", 128, 128, 420, 620, 3, 0), ("conv 256->256 3x3 210x310", 256, 256, 210, 310, 3, 0),
                                       ("conv 512->512 3x3 105x155", 512, 512, 105, 155, 3, 0), ("conv 512->1024 1x1 105x155", 512, 1024, 105, 155, 1, 0),
                                       ("conv 256->128 1x1 420x620", 256, 128, 420, 620, 1, 0), ("conv 128->108 3x3 420x620", 128, 108, 420, 620, 3, 0),
                                       ("upsample 256->256 sub-pixel 210x310", 256, 256, 210, 310, 3, 2), ("upsample 512->512 loader 105x155", 512, 512, 105, 155, 3, 1),
                                       ("conv 64->1536 3x3 105x155", 64, 1536, 105, 155, 3, 0)):
        x = torch.randn(B, h, w, ci, generator=g).to(torch.bfloat16).to(DEV)
        wt = (torch.randn(co, ci, k, k, generator=g) * 0.02).to(DEV)
        pc = ops.PackedConv(wt, torch.zeros(co, device=DEV), upsample_subpixel=(ups == 2))
        total += check(name, lambda: ops.conv2d(x, pc, upsample=bool(ups)))
        if k == 3 and not ups and co % 8 == 0:
            r = torch.randn(B, h, w, co, generator=g).to(torch.bfloat16).to(DEV)
            total += check(name + " +res +gn", lambda: ops.conv2d(x, pc, residual=r, gn_stats=(co % 128 == 0)))
    x = torch.randn(B, 210, 310, 256, generator=g).to(torch.bfloat16).to(DEV)
    wt = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(DEV)
    total += check("conv 256->256 3x3 stride 2", lambda: ops.conv2d(x, ops.PackedConv(wt), stride=2))
    N, C = 105 * 155, 512
    qk = (torch.randn(B, N, 2 * C, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
    npad = (N + 63) // 64 * 64
    vt = torch.zeros(B, C, npad, dtype=torch.bfloat16, device=DEV)
    vt[:, :, :N] = torch.randn(B, C, N, generator=g).to(torch.bfloat16).to(DEV)
    total += check("attention B=8 N=16275", lambda: ops.attention_d512(qk, qk[..., C:], vt, N, ldq=2 * C, ldk=2 * C))
    total += check("attention B=1 split keys", lambda: ops.attention_d512(qk[:1], qk[:1, :, C:], vt[:1], N, ldq=2 * C, ldk=2 * C))
    for c, h, w in ((128, 420, 620), (256, 210, 310)):
        x = torch.randn(B, h, w, c, generator=g).to(torch.bfloat16).to(DEV)
        plane = (h * w + 63) // 64 * 64
        om = torch.randn(B, 108, plane, generator=g).to(DEV)
        pd = ops.PackedDcn((torch.randn(c, c, 3, 3, generator=g) * 0.02).to(DEV), torch.zeros(c, device=DEV), 4)
        total += check("dcn forward C=%d %dx%d" % (c, h, w), lambda: ops.mdcn_forward_nhwc(x, om, pd))
        total += check("dcn forward (general kernel) C=%d" % c, lambda: ops.mdcn_forward_nhwc(x, om, pd, flags=ops.MDCN_GENERAL_KERNEL))
        with ops.use_precision("fp16"):     # the single-pass form of the fp16 inference precision (128-pixel workgroups at C = 128)
            xh = x.to(torch.float16)
            pd1 = ops.PackedDcn((torch.randn(c, c, 3, 3, generator=g) * 0.02).to(DEV), torch.zeros(c, device=DEV), 4, single=True)
            total += check("dcn forward single pass fp16 C=%d" % c, lambda: ops.mdcn_forward_nhwc(xh, om, pd1))
    # the flow's grouped launch: 24 filters of one shape on channel slices of one tensor
    xg = torch.randn(B, 105, 155, 1536, generator=g).to(torch.bfloat16).to(DEV)
    pcs = ops.packed_conv_batch((torch.randn(24, 64, 64, 1, 1, generator=g) * 0.1).to(DEV), torch.randn(24, 64, generator=g).to(DEV))
    og = torch.empty_like(xg)
    total += check("grouped conv 24 x (64->64 1x1)", lambda: ops.conv2d_grouped(xg, pcs, cin=64, in_step=64, out=og, out_step=64, act="relu").clone())
    x = torch.randn(B, 420, 620, 128, generator=g).to(torch.bfloat16).to(DEV)
    gm, bt = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    total += check("groupnorm 128 420x620", lambda: ops.groupnorm(x, gm, bt))
    z = torch.randn(B * N, 3, generator=g).to(DEV)
    cb = (torch.randn(8192, 3, generator=g) * 0.7).to(DEV)
    total += check("vq nearest", lambda: ops.vq_nearest(z, cb)[0])
    try:
        from glare_amd import train_ops as T
        a = torch.randn(4096, 2304, generator=g).to(torch.bfloat16).to(DEV)
        b = torch.randn(512, 2304, generator=g).to(torch.bfloat16).to(DEV)
        total += check("gemm_nt 4096x512x2304", lambda: T.gemm_nt(a, b))
    except Exception as e:  # noqa: BLE001
        print("gemm_nt skipped:", e)
    print("TOTAL differing elements:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
