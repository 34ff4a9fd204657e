#!/usr/bin/env python
"""Micro-benchmarks of the individual HIP kernels at the shapes of the 400x600, B=8 path.
    python tools/kbench.py [attn] [conv] [gn] [dcn] [vq]   (default: all)
Prints one line per case: time per launch and achieved TFLOP/s or GB/s (algorithmic work)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glare_amd import ops  # noqa: E402

DEV = "cuda"
B = int(os.environ.get("KB_BATCH", "8"))
REPS = int(os.environ.get("KB_REPS", "5"))


def timeit(fn, reps=REPS, warm=2, warm_s=0.3):
    # warm up by TIME, not by count: the first case measured after an idle stretch (allocation, a rebuild) ran 15 % slow with two
    # warm-up launches (conv3 128 -> 128 @full: 2.01 ms first, 1.73 ms when measured again; round 4) -- clocks and first-touch mappings
    import time
    t0 = time.time()
    n = 0
    while n < warm or time.time() - t0 < warm_s:
        fn()
        torch.cuda.synchronize()
        n += 1
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def timeit_queued(fn, reps=50):
    """GPU time of `fn` when the host is AHEAD of the device (as inside the pipeline): a ~4 ms attention launch goes first, the `reps`
    launches queue up behind it, and the events bracket only them -- python / ctypes call overhead (10-20 us) is off the clock."""
    N, C = 105 * 155, 512
    q = torch.zeros(8, N, C, dtype=ops.act_dtype(), device=DEV)
    out = torch.empty_like(q)
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.attention_kv512(q, q, N, out=out, key_splits=1)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def bench_attn():
    N, C = 105 * 155, 512
    qk = (torch.randn(B, N, 2 * C, device=DEV) * 0.3).to(torch.bfloat16)
    npad = (N + 63) // 64 * 64
    vt = torch.zeros(B, C, npad, dtype=torch.bfloat16, device=DEV)
    vt[:, :, :N] = torch.randn(B, C, N, device=DEV).to(torch.bfloat16)
    out = torch.empty(B, N, C, dtype=torch.bfloat16, device=DEV)
    ms = timeit(lambda: ops.attention_d512(qk, qk[..., C:], vt, N, ldq=2 * C, ldk=2 * C, out=out))
    print("attn  B=%d N=%d d=512: %.3f ms  %.0f TFLOP/s" % (B, N, ms, 4.0 * B * N * N * C / ms / 1e9))
    x = torch.randn(B, N, C, device=DEV).to(torch.bfloat16)
    q = (torch.randn(B, N, C, device=DEV) * 0.3).to(torch.bfloat16)
    ms = timeit(lambda: ops.attention_kv512(q, x, N, out=out, key_splits=1))
    print("attnkv B=%d N=%d d=512 (shared K/V): %.3f ms  %.0f TFLOP/s" % (B, N, ms, 4.0 * B * N * N * C / ms / 1e9))
    if os.environ.get("KB_ZERO"):   # the same launch on all-zero operands: same instruction stream, far less switching power -- what the
        xz, qz = torch.zeros_like(x), torch.zeros_like(q)          # clock does to the figure (the kernel is power-limited)
        ms = timeit(lambda: ops.attention_kv512(qz, xz, N, out=out, key_splits=1))
        print("attnkv B=%d N=%d d=512 (shared K/V), ALL-ZERO operands: %.3f ms  %.0f TFLOP/s" % (B, N, ms, 4.0 * B * N * N * C / ms / 1e9))
        ms = timeit(lambda: ops.attention_kv512(q, x, N, out=out, key_splits=1))
        print("attnkv B=%d N=%d d=512 (shared K/V), random again: %.3f ms  %.0f TFLOP/s" % (B, N, ms, 4.0 * B * N * N * C / ms / 1e9))


def bench_conv():
    only = os.environ.get("KB_CONV")
    cases = [("512->512 3x3 @q", 512, 512, 105, 155, 3), ("256->256 3x3 @half", 256, 256, 210, 310, 3),
             ("128->128 3x3 @full", 128, 128, 420, 620, 3), ("512->512 3x3 @half", 512, 512, 210, 310, 3),
             ("256->256 3x3 @full", 256, 256, 420, 620, 3), ("512->1024 1x1 @q", 512, 1024, 105, 155, 1),
             ("512->512 1x1 @q", 512, 512, 105, 155, 1), ("64->1536 3x3 @q", 64, 1536, 105, 155, 3)]
    for name, ci, co, h, w, k in cases:
        if only and only not in name:
            continue
        x = torch.randn(B, h, w, ci, device=DEV).to(torch.bfloat16)
        wt = torch.randn(co, ci, k, k, device=DEV) * 0.02
        pc = ops.PackedConv(wt, torch.zeros(co, device=DEV))
        out = torch.empty(B, h, w, co, dtype=torch.bfloat16, device=DEV)
        ms = timeit(lambda: ops.conv2d(x, pc, out=out))
        fl = 2.0 * B * h * w * ci * co * k * k
        print("conv  %-22s: %.3f ms  %.0f TFLOP/s" % (name, ms, fl / ms / 1e9))
        if os.environ.get("KB_ZERO"):
            xz = torch.zeros_like(x)
            pz = ops.PackedConv(torch.zeros_like(wt), torch.zeros(co, device=DEV))
            msz = timeit(lambda: ops.conv2d(xz, pz, out=out))
            print("conv  %-22s: %.3f ms  %.0f TFLOP/s  (ALL-ZERO operands)" % (name, msz, fl / msz / 1e9))
        res = torch.randn(B, h, w, co, device=DEV).to(torch.bfloat16)
        ms = timeit(lambda: ops.conv2d(x, pc, out=out, residual=res))
        print("conv  %-22s: %.3f ms  %.0f TFLOP/s  (+residual)" % (name, ms, fl / ms / 1e9))
        if k == 1 and getattr(pc, "w16", None) is not None:   # the same through the implicit-GEMM kernel, and the HBM roofline
            ops.CONV1X1_WEIGHT_STATIONARY = False
            ms_old = timeit(lambda: ops.conv2d(x, pc, out=out))
            ops.CONV1X1_WEIGHT_STATIONARY = True
            ms_new = timeit(lambda: ops.conv2d(x, pc, out=out))
            by = 2.0 * B * h * w * (ci + co)
            print("conv  %-22s: igemm %.3f ms, weight-stationary %.3f ms = %.0f GB/s algorithmic (x in + out)" % (name, ms_old, ms_new, by / ms_new / 1e6))


def bench_convtile():
    """The output-channel tile (glare_conv_desc.cout_tile) at the path's shapes: 128 (default) against 64 -- twice the workgroups of
    half the accumulators (4 instead of 3 workgroups per CU), for the launches whose K loop is short (Cin = 128) or whose tile count
    rounds badly (8 480 tiles on 768 slots at full resolution)."""
    for name, ci, co, h, w in (("128->128 3x3 @full", 128, 128, 420, 620), ("256->256 3x3 @half", 256, 256, 210, 310),
                               ("512->512 3x3 @q", 512, 512, 105, 155), ("256->128 3x3 @full", 256, 128, 420, 620)):
        x = torch.randn(B, h, w, ci, device=DEV).to(ops.act_dtype())
        wt = torch.randn(co, ci, 3, 3, device=DEV) * 0.02
        res = torch.randn(B, h, w, co, device=DEV).to(ops.act_dtype())
        out = torch.empty(B, h, w, co, dtype=ops.act_dtype(), device=DEV)
        fl = 2.0 * B * h * w * ci * co * 9
        ref = None
        for tile in (0, 64, 0, 64):
            pc = ops.PackedConv(wt, torch.zeros(co, device=DEV), cout_tile=tile)
            ms = timeit(lambda: ops.conv2d(x, pc, out=out, gn_stats=True))
            msr = timeit(lambda: ops.conv2d(x, pc, out=out, residual=res, gn_stats=True))
            same = "" if ref is None else ("  bit-identical" if torch.equal(out, ref) else "  DIFFERS")
            ref = out.clone() if ref is None else ref
            print("convtile %-20s tile %3d: %.3f ms %.0f TFLOP/s | + residual %.3f ms %.0f TFLOP/s%s"
                  % (name, tile or 128, ms, fl / ms / 1e9, msr, fl / msr / 1e9, same))


def bench_convsplit():
    """The fp32-class form (glare_conv_desc.k_wrap; ops.PackedConv(split=3)) at the conditional encoder's shapes, fp16, pair in / pair
    out: TFLOP/s by ALGORITHMIC FLOPs (one fp32 conv) and by executed MFMA FLOPs (3 K segments)."""
    only = os.environ.get("KB_CONV")
    cases = [("128->128 3x3 @full", 128, 128, 420, 620, 3), ("256->256 3x3 @half", 256, 256, 210, 310, 3),
             ("512->512 3x3 @q", 512, 512, 105, 155, 3), ("512->512 1x1 @q", 512, 512, 105, 155, 1), ("64->1536 3x3 @q", 64, 1536, 105, 155, 3)]
    with ops.use_precision("fp16"):
        for name, ci, co, h, w, k in cases:
            if only and only not in name:
                continue
            x = ops.split_hilo(torch.randn(B, h, w, ci, device=DEV))
            wt = torch.randn(co, ci, k, k, device=DEV) * 0.02
            pc = ops.PackedConv(wt, torch.zeros(co, device=DEV), split=3)
            res = ops.split_hilo(torch.randn(B, h, w, co, device=DEV))
            out = torch.empty(B, h, w, co, dtype=torch.float16, device=DEV)
            out._lo = torch.empty_like(out)
            ms = timeit(lambda: ops.conv2d(x, pc, out=out, hilo=True))
            fl = 2.0 * B * h * w * ci * co * k * k
            print("conv3 %-22s: %.3f ms  %.0f TFLOP/s algorithmic, %.0f executed (pair out)" % (name, ms, fl / ms / 1e9, 3 * fl / ms / 1e9))
            ms = timeit(lambda: ops.conv2d(x, pc, out=out, residual=res, hilo=True, gn_stats=(co % 128 == 0)))
            print("conv3 %-22s: %.3f ms  %.0f TFLOP/s algorithmic, %.0f executed (+ pair residual, statistics)" % (name, ms, fl / ms / 1e9, 3 * fl / ms / 1e9))
            ms = timeit(lambda: ops.conv2d(x, pc, out=out))
            print("conv3 %-22s: %.3f ms  %.0f TFLOP/s algorithmic, %.0f executed (16-bit out: what the pair epilogue costs)" % (name, ms, fl / ms / 1e9, 3 * fl / ms / 1e9))
            if os.environ.get("KB_ORDER"):     # is the first case of a shape measured slow (clocks, placement)?  pair / 16-bit alternated
                for _ in range(2):
                    a = timeit(lambda: ops.conv2d(x, pc, out=out, hilo=True))
                    c = timeit(lambda: ops.conv2d(x, pc, out=out))
                    print("conv3 %-22s: again: pair out %.3f ms, 16-bit out %.3f ms" % (name, a, c))
            pc1 = ops.PackedConv(wt, torch.zeros(co, device=DEV))
            ms1 = timeit(lambda: ops.conv2d(x, pc1, out=out))
            print("conv3 %-22s: single pass, 16-bit out %.3f ms  %.0f TFLOP/s" % (name, ms1, fl / ms1 / 1e9))


def bench_gnpro():
    """GroupNorm + swish as the conv's loader prologue (glare_conv_desc.gn_coef) against the separate apply pass + conv, fp16."""
    with ops.use_precision("fp16"):
        for name, c, h, w in (("128->128 3x3 @full", 128, 420, 620), ("256->256 3x3 @half", 256, 210, 310), ("512->512 3x3 @q", 512, 105, 155)):
            x0 = torch.randn(B, h, w, c, device=DEV).half()
            x = ops.conv2d(x0, ops.PackedConv(torch.randn(c, c, 3, 3, device=DEV) * 0.02, None), gn_stats=True)
            g, b = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
            pc = ops.PackedConv(torch.randn(c, c, 3, 3, device=DEV) * 0.02, torch.zeros(c, device=DEV))
            out = torch.empty(B, h, w, c, dtype=torch.float16, device=DEV)
            t_gn = timeit(lambda: ops.groupnorm(x, g, b, swish=True))
            y = ops.groupnorm(x, g, b, swish=True)
            t_conv = timeit(lambda: ops.conv2d(y, pc, out=out))
            t_sep = timeit(lambda: ops.conv2d(ops.groupnorm(x, g, b, swish=True), pc, out=out))
            t_pro = timeit(lambda: ops.conv2d(x, pc, out=out, gn_prologue=(ops.groupnorm_coeffs(x, g, b), True)))
            print("gnpro %-20s: apply %.3f + conv %.3f = separate %.3f ms | prologue form %.3f ms (%+.1f %%)"
                  % (name, t_gn, t_conv, t_sep, t_pro, 100.0 * (t_pro - t_sep) / t_sep))


def bench_gn():
    for c, h, w in ((128, 420, 620), (256, 210, 310), (512, 105, 155)):
        x = torch.randn(B, h, w, c, device=DEV).to(torch.bfloat16)
        g = torch.ones(c, device=DEV)
        b = torch.zeros(c, device=DEV)
        ms = timeit(lambda: ops.groupnorm(x, g, b, swish=True))
        gb = 3.0 * x.numel() * 2 / 1e9
        print("gn    C=%d %dx%d: %.3f ms  %.0f GB/s (algorithmic 2 reads + 1 write)" % (c, h, w, ms, gb / ms * 1e3))


def bench_gnapply():
    """The GroupNorm + swish APPLY pass alone (statistics from the producer), fp16, the path's three shapes."""
    with ops.use_precision("fp16"):
        for c, h, w in ((128, 420, 620), (256, 210, 310), (512, 105, 155)):
            x = torch.randn(B, h, w, c, device=DEV).half()
            g, b = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
            x._gn_stats = torch.rand(B, 4, 32, 2, device=DEV) * 1000 + 1000
            ms = timeit(lambda: ops.groupnorm(x, g, b, swish=True), reps=10)
            print("gn apply C=%d %dx%d: %.3f ms  %.2f TB/s (1 read + 1 write)" % (c, h, w, ms, 2.0 * x.numel() * 2 / ms / 1e9))


def bench_dcn():
    only = os.environ.get("KB_DCN")      # "128" / "256": one shape (per-shape PMC passes)
    for prec in ("bf16", "fp16"):
        with ops.use_precision(prec):
            for c, h, w in ((128, 420, 620), (256, 210, 310)):
                if only and only != str(c):
                    continue
                x = torch.randn(B, h, w, c, device=DEV).to(ops.act_dtype())
                plane = (h * w + 63) // 64 * 64
                om = torch.randn(B, 108, plane, device=DEV)
                wt = torch.randn(c, c, 3, 3, device=DEV) * 0.02
                pd = ops.PackedDcn(wt, torch.zeros(c, device=DEV), 4)
                ms = timeit(lambda: ops.mdcn_forward_nhwc(x, om, pd))
                fl = 2.0 * B * h * w * c * c * 9 + 72.0 * B * h * w * c
                print("dcn   %s C=%d %dx%d split form: %.3f ms  %.1f TFLOP/s-equivalent" % (prec, c, h, w, ms, fl / ms / 1e9))
                if prec == "fp16":
                    pd1 = ops.PackedDcn(wt, torch.zeros(c, device=DEV), 4, single=True)
                    ms = timeit(lambda: ops.mdcn_forward_nhwc(x, om, pd1))
                    print("dcn   %s C=%d %dx%d single pass: %.3f ms" % (prec, c, h, w, ms))


def bench_dcnbwd():
    """DCNv2 backward at the stage-3 crop (B = 1, 256x256 and its half resolution), through the drop-in of
    modulated_deform_conv_backward; with and without grad_input (in stage 3 the sampled feature is frozen: no scatter)."""
    from glare_amd.modules.ops.dcn.deform_conv import deform_conv_ext
    for c, h, w in ((128, 256, 256), (256, 128, 128)):
        x = torch.randn(1, c, h, w, device=DEV)
        off = torch.randn(1, 72, h, w, device=DEV) * 2
        m = torch.rand(1, 36, h, w, device=DEV)
        wt = torch.randn(c, c, 3, 3, device=DEV) * 0.02
        b = torch.zeros(c, device=DEV)
        go = torch.randn(1, c, h, w, device=DEV)
        for want_gx in (False, True):
            gin = torch.zeros_like(x) if want_gx else None
            goff, gm, gw, gb = torch.zeros_like(off), torch.zeros_like(m), torch.zeros_like(wt), torch.zeros_like(b)
            ms = timeit(lambda: deform_conv_ext.modulated_deform_conv_backward(x, wt, b, x.new_empty(0), off, m, x.new_empty(0), gin, gw, gb,
                                                                               goff, gm, go, 3, 3, 1, 1, 1, 1, 1, 1, 1, 4, True))
            print("dcnbwd C=%d %dx%d grad_input=%d: %.3f ms" % (c, h, w, want_gx, ms))


def bench_wgrad():
    """Weight gradient of the 3x3 convs at the stage-2 crop (B = 2, 320x320 and its half / quarter resolutions): csrc/wgrad.hip
    plus the reduction of its split partials."""
    from glare_amd import train_ops as T
    for name, cin, cout, H, W in [("128->128 @320", 128, 128, 320, 320), ("256->256 @160", 256, 256, 160, 160),
                                  ("512->512 @80", 512, 512, 80, 80), ("128->256 @160", 128, 256, 160, 160),
                                  ("256->512 @80", 256, 512, 80, 80)]:
        x = torch.randn(2, H, W, cin, device=DEV).to(torch.bfloat16)
        g = torch.randn(2, H, W, cout, device=DEV).to(torch.bfloat16)
        ms = timeit(lambda: T.conv3x3_weight_grad(x, g, cout))
        fl = 2.0 * 2 * H * W * 9 * cin * cout
        print("wgrad %-15s: %.3f ms  %.0f TFLOP/s (NHWC kernel + reduction)" % (name, ms, fl / ms / 1e9))


def bench_attnbwd():
    """Attention backward at the stage-2 latent (B = 2, N = 80 x 80 tokens): the fused kernel against the materialised N^2 form."""
    from glare_amd import train_ops as T
    Bt, N, C = 2, 6400, 512
    q = (torch.randn(Bt, N, C, device=DEV) * 0.06).to(torch.bfloat16)
    k, v, do = [torch.randn(Bt, N, C, device=DEV).to(torch.bfloat16) for _ in range(3)]
    vt = T.transpose(v, (N + 63) // 64 * 64)
    lse = torch.empty(Bt, N, dtype=torch.float32, device=DEV)
    o = ops.attention_d512(q, k, vt, N, lse=lse)
    fl = 5 * 2.0 * Bt * N * N * C
    ms = timeit(lambda: T.attention_backward_fused(q, k, v, o, do, lse))
    ms_old = timeit(lambda: T.attention_backward(q, k, v, o, do))
    print("attnbwd B=%d N=%d d=512: fused %.3f ms  %.0f TFLOP/s algorithmic (5 products);  materialised %.3f ms" % (Bt, N, ms, fl / ms / 1e9, ms_old))


def bench_attnfold():
    """VERDICT r04 item 5 -- AttnBlock's two per-image 1x1 GEMMs as prologue / epilogue of attn_kv_fwd_kernel: what the separate
    launches cost today against what the same MFMAs would cost INSIDE the attention kernel (fp16, B = 8, N = 16 275).
    Inside the kernel a workgroup owns 128 query rows; producing its Q tile is 128 x 512 x 512 MACs = the score product against 512
    keys (16 key tiles), projecting its O tile the same again as a P.V product -- together ONE pass of the main loop over 16 more key
    tiles per workgroup, plus a 512 KB filter streamed through LDS per workgroup and product (the K / V stream is 16.7 MB)."""
    with ops.use_precision("fp16"):
        H, W, C = 105, 155, 512
        N = H * W
        x = torch.randn(B, H, W, C, device=DEV).half()
        stats = ops.conv2d(x, ops.PackedConv(torch.randn(C, C, 1, 1, device=DEV) * 0.04, None), gn_stats=True)._gn_stats
        wq, wo = torch.randn(C, C, device=DEV) * 0.04, torch.randn(C, C, device=DEV) * 0.04
        bq, bo, g, b = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        t_fold = timeit(lambda: ops.attn_fold_groupnorm(stats, N, g, b, 1e-6, wq, bq, wo, bo))
        wq_b, bq_b, wo_b, bo_b = ops.attn_fold_groupnorm(stats, N, g, b, 1e-6, wq, bq, wo, bo)
        t_q = timeit(lambda: ops.conv1x1_per_image(x, wq_b, bq_b))
        q = ops.conv1x1_per_image(x, wq_b, bq_b)
        out = torch.empty(B, N, C, dtype=torch.float16, device=DEV)
        t_a = timeit(lambda: ops.attention_kv512(q.view(B, N, C), x.view(B, N, C), N, out=out, key_splits=1))
        t_o = timeit(lambda: ops.conv1x1_per_image(out.view(B, H, W, C), wo_b, bo_b, residual=x, gn_stats=True))
        qb, kt = (N + 127) // 128, (N + 31) // 32
        per_tile = t_a / kt                       # one pass of every workgroup over one 32-key tile
        inside = 16 * per_tile                    # 16 tile-equivalents: Q prologue (as a score product) + O epilogue (as a P.V product)
        print("attnfold today, per block: fold kernel %.3f + q 1x1 %.3f + out 1x1 (+ residual, statistics) %.3f = %.3f ms in 3 launches; attention %.3f ms"
              % (t_fold, t_q, t_o, t_fold + t_q + t_o, t_a))
        print("attnfold inside the kernel: %d query blocks x %d key tiles at %.2f us per tile pass -> the two GEMMs = 16 tile passes = %.3f ms "
              "(+%.1f %% of the launch), before the 2 x 512 KB filter stream per workgroup, the Q / O layout changes through LDS and the "
              "residual + statistics epilogue that would move in with them" % (qb, kt, per_tile * 1e3, inside, 100 * inside / t_a))
        print("attnfold bound on the gain: %.3f - %.3f = %.3f ms per block (the fold kernel stays: it builds the per-image filters), x 11 blocks = %.2f ms per step of %d images"
              % (t_q + t_o, inside, t_q + t_o - inside, 11 * (t_q + t_o - inside), B))


def bench_convin():
    """conv_in 3 -> 128 on the NCHW fp32 image at full resolution (conv_small_kernel): HBM-bound on its bf16 output."""
    H, W, co = 420, 620, 128
    x = torch.randn(B, 3, H, W, device=DEV)
    w = torch.randn(co, 3, 3, 3, device=DEV) * 0.2
    b = torch.randn(co, device=DEV)
    out = torch.empty(B, H, W, co, dtype=torch.bfloat16, device=DEV)
    ms = timeit(lambda: ops.conv2d_smallcin(x, (3 * H * W, H * W, W, 1), (B, H, W), w, b, out=out))
    print("convin 3->128 3x3 @full: %.3f ms  %.0f GB/s (bf16 output + fp32 input)" % (ms, (out.numel() * 2 + x.numel() * 4) / ms / 1e6))


def bench_vq():
    n = B * 105 * 155
    z = torch.randn(n, 3, device=DEV)
    cb = torch.randn(8192, 3, device=DEV) * 0.7
    ms = timeit(lambda: ops.vq_nearest(z, cb))
    print("vq    %d tokens x 8192 codes: %.3f ms  %.2f Gtoken-code/s" % (n, ms, n * 8192 / ms / 1e6))


def bench_conv1x1():
    """Every 1x1 shape of the path on both kernels: the implicit-GEMM form (conv_igemm, KS = 1) against the weight-stationary one
    (conv1x1.hip), 16-bit and fp32-class (pair in / pair out)."""
    cases = [("128->256 @half", 128, 256, 210, 310), ("256->512 @q", 256, 512, 105, 155), ("512->512 @q", 512, 512, 105, 155),
             ("256->128 @full", 256, 128, 420, 620), ("512->256 @half", 512, 256, 210, 310), ("512->1024 @q", 512, 1024, 105, 155)]
    with ops.use_precision("fp16"):
        for name, ci, co, h, w in cases:
            wt = torch.randn(co, ci, 1, 1, device=DEV) * 0.02
            x = ops.split_hilo(torch.randn(B, h, w, ci, device=DEV))
            res = ops.split_hilo(torch.randn(B, h, w, co, device=DEV))
            out = torch.empty(B, h, w, co, dtype=torch.float16, device=DEV)
            out._lo = torch.empty_like(out)
            pc, pc3 = ops.PackedConv(wt, torch.zeros(co, device=DEV)), ops.PackedConv(wt, torch.zeros(co, device=DEV), split=3)
            t = {}
            for ws in (False, True):
                ops.CONV1X1_WEIGHT_STATIONARY = ws
                t[ws] = (timeit(lambda: ops.conv2d(x, pc, out=out)), timeit(lambda: ops.conv2d(x, pc, out=out, residual=res)),
                         timeit(lambda: ops.conv2d(x, pc3, out=out, hilo=True)), timeit(lambda: ops.conv2d(x, pc3, out=out, residual=res, hilo=True)))
            ops.CONV1X1_WEIGHT_STATIONARY = True
            print("conv1x1 %-16s: 16-bit igemm %.3f / ws %.3f | + residual %.3f / %.3f | fp32-class pair out %.3f / %.3f | + pair residual %.3f / %.3f ms"
                  % (name, t[False][0], t[True][0], t[False][1], t[True][1], t[False][2], t[True][2], t[False][3], t[True][3]))


def bench_winograd():
    """VERDICT r05 item 5, step 2 priced with the kernels that exist: F(2x2, 3x3) on 256 -> 256 @half (8 x 210 x 310) is 16 independent
    [T x 256] . [256 x 256] products over T = 8 x 105 x 155 tiles (2.25x fewer MACs than the direct conv) between an input transform that
    writes 4x the activation bytes and an output transform that reads 4x the output bytes.  Measured: the direct single-pass conv, and the
    16 products alone as ONE batched launch of gemm_nt (16-bit in / out, operands resident: no transform cost at all).  A fused Winograd
    kernel has to beat the direct conv by 1.5x INCLUDING both transforms; the products alone bound what is left for them."""
    from glare_amd import train_ops as T
    with ops.use_precision("fp16"):
        h, w, c = 210, 310, 256
        x = (torch.randn(B, h, w, c, device=DEV) * 0.5).to(torch.float16)
        pc = ops.PackedConv(torch.randn(c, c, 3, 3, device=DEV) * 0.02, torch.zeros(c, device=DEV))
        out = torch.empty(B, h, w, c, dtype=torch.float16, device=DEV)
        ms_d = timeit(lambda: ops.conv2d(x, pc, out=out))
        fl_d = 2.0 * B * h * w * c * c * 9
        tiles = B * (h // 2) * (w // 2)
        V = (torch.randn(16, tiles, c, device=DEV) * 0.5).to(torch.float16)
        U = (torch.randn(16, c, c, device=DEV) * 0.02).to(torch.float16)
        M = torch.empty(16, tiles, c, dtype=torch.float16, device=DEV)
        ms_g = timeit(lambda: T.gemm_nt(V, U, out=M, out_dtype=torch.float16))
        fl_g = 2.0 * 16 * tiles * c * c
        act = B * h * w * c * 2.0
        print("winograd 256->256 @half: direct conv %.3f ms (%.0f TFLOP/s) | the 16 Winograd products alone, one batched gemm_nt: %.3f ms (%.0f TFLOP/s on "
              "%.0f GFLOP = 1/2.25 of the direct conv's) | a 1.5x win needs <= %.3f ms in all: %.3f ms left for an input transform writing %.2f GB and an output "
              "transform reading %.2f GB if they went through HBM (%.3f ms at 5 TB/s)"
              % (ms_d, fl_d / ms_d / 1e9, ms_g, fl_g / ms_g / 1e9, fl_g / 1e9, ms_d / 1.5, ms_d / 1.5 - ms_g, 4 * act / 1e9, 4 * act / 1e9,
                 (act + 4 * act + 4 * act + act) / 5e12 * 1e3))


def bench_flow():
    """The flow's reverse pass at the path's latent size (8 x 105 x 155): round 6's fused step (one launch) against rounds 1-5's
    four launches per step, fp32-class (pair cond_feat) in fp16."""
    from glare_amd import modules as M
    import importlib
    FU = importlib.import_module("glare_amd.modules.FlowUpsamplerNet")
    from glare_amd.synthetic import seeded_init_
    net = seeded_init_(M.VQLLFLOWDeformable().eval(), 0).to(DEV).flowUpsamplerNet
    with ops.use_precision("fp16"), torch.no_grad():
        z = torch.randn(B, 105, 155, 3, device=DEV) * 0.7
        ft = ops.split_hilo(torch.sigmoid(torch.randn(B, 105, 155, 64, device=DEV)))
        for fused in (True, False, True, False):
            FU.FUSED_STEP = fused
            net.invalidate()
            ms = timeit(lambda: net.decode_nhwc(z, ft))
            print("flow decode B=%d 105x155 (24 coupling steps + the batched z-independent convs), %s: %.3f ms" % (B, "fused step" if fused else "four launches per step", ms))
        FU.FUSED_STEP = True
        net.invalidate()
        P = net._packed(("flow", 3), lambda: net._prepare(3))
        ftA = torch.randn(B, 105, 155, 64 * P["n"], device=DEV)
        hF = torch.randn(B, 105, 155, 8 * P["n"], device=DEV)
        z2 = torch.empty_like(z)
        st = P["steps"][0]
        ms = timeit(lambda: ops.flow_step_fused(z, z2, ftA, 0, st["image"], hF, 0, st["M"], st["t"], st["eps"]), reps=20)
        msq = timeit_queued(lambda: ops.flow_step_fused(z, z2, ftA, 0, st["image"], hF, 0, st["M"], st["t"], st["eps"]), reps=100)
        print("flow_step_fused alone: %.4f ms per step host-paced, %.4f ms queued behind a long launch (device time)" % (ms, msq))
        for fused in (True, False):
            FU.FUSED_STEP = fused
            net.invalidate()
            print("flow decode queued, %s: %.3f ms" % ("fused step" if fused else "four launches per step", timeit_queued(lambda: net.decode_nhwc(z, ft), reps=3)))
        FU.FUSED_STEP = True


if __name__ == "__main__":
    which = sys.argv[1:] or ["attn", "conv", "gn", "dcn", "vq", "wgrad", "attnbwd"]
    for w in which:
        globals()["bench_" + w]()
