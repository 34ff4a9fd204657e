"""torch.autograd.Function wrappers: HIP forward + HIP backward for the operators of the training step
(SURVEY.md section 8 rows a12 / a13).  torch's autograd engine is only the tape: every feature-map-sized computation
in forward() and backward() below is a kernel of libglare_hip.so, including the gradient accumulation at fan-out points
(ForkFn); torch ops touch filter-sized tensors, the 3-channel latent / image gradients and per-sample scalars only.

The reference gets these gradients from `loss.backward()` over cuDNN / ATen (LLFlow_model.py:231-236,
VQLLFLOWD_model.py:226-229); activations here are NHWC bf16, parameters and their gradients fp32.
"""
import torch
from . import _lib, ops
from . import train_ops as T


IMPLICIT_WGRAD = True   # 3x3 stride-1 weight gradients straight from the NHWC operands, csrc/wgrad.hip (False: im2col_t + GEMM)


def _rup(a, b):
    return (a + b - 1) // b * b


def _grad_bf16(g, y, act, cout):
    """Upstream gradient -> dense bf16 NHWC with a channel pitch that is a multiple of 8 (extra channels zero),
    after the fused activation's derivative."""
    if act != "none":
        g = g.contiguous().clone()
        T.act_backward_(g, y, act)
    g = g.contiguous()
    cp = _rup(cout, 8)
    if g.dtype == torch.float32:
        return T.cast_to_bf16(g, pitch=cp)
    if cp != cout:
        return T.cast_to_bf16(T.cast_to_f32(g), pitch=cp)
    return g


def _tile(B, OH, OW, cout, k):
    """Output-channel tile of a training conv: the crops are small, a 3x3 conv with the default 128-wide tile often has fewer
    workgroups than the chip has slots (ops.conv_cout_tile); 1x1 convs keep their default (weight-stationary kernel)."""
    return ops.conv_cout_tile(B, OH, OW, cout) if k == 3 else 0


def _data_grad(g16, weight, cout, stride, upsample, out_f32=False):
    """dx of a conv as a stride-1 conv of the (dilated) gradient with the flipped, transposed filter."""
    B, gh, gw, _ = g16.shape
    if stride == 2:
        gh, gw = 2 * gh, 2 * gw
    pc = ops.packed_for(weight, dgrad_pad=g16.shape[-1], cout_tile=_tile(B, gh, gw, weight.shape[1], weight.shape[2]))
    mode = ops.OUT_NHWC_F32 if out_f32 else ops.OUT_NHWC_BF16                # (flip / transpose / channel pad inside the pack kernel)
    if stride == 2:
        return ops.conv2d(T.dilate2(g16), pc, out_mode=mode)
    dx = ops.conv2d(g16, pc, out_mode=mode)
    return T.pool2_sum(dx) if upsample else dx


@_lib.keeps_precision
class Conv2dFn(torch.autograd.Function):
    """y = act(conv(x [, x2]) + bias) [+ residual]; x bf16 NHWC; weight OIHW fp32 (3x3 pad 1 / 1x1; stride 2 with the
    (0,1,0,1) padding; optional fused nearest x2 upsample of the input)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, x2, stride, upsample, act, out_f32):
        B, H, W, _ = x.shape
        OH, OW = (2 * H, 2 * W) if upsample else (H, W)
        if stride == 2:
            OH, OW = (OH + 1 - 3) // 2 + 1, (OW + 1 - 3) // 2 + 1
        pc = ops.packed_for(weight, bias, cout_tile=_tile(B, OH, OW, weight.shape[0], weight.shape[2]))
        y = ops.conv2d(x, pc, x2=x2, stride=stride, upsample=upsample, act=act, residual=residual,
                       out_mode=ops.OUT_NHWC_F32 if out_f32 else ops.OUT_NHWC_BF16)
        ctx.cfg = (stride, upsample, act, bias is not None, residual is not None, x2 is not None)
        ctx.save_for_backward(x, weight, y if act != "none" else None, x2)
        return y

    @staticmethod
    def backward(ctx, gy):
        stride, upsample, act, has_bias, has_res, has_x2 = ctx.cfg
        x, weight, y, x2 = ctx.saved_tensors
        cout, cin_tot, k, _ = weight.shape
        assert not (has_res and act != "none")
        g16 = _grad_bf16(gy, y, act, cout)
        dres = g16 if (has_res and ctx.needs_input_grad[3]) else None
        dw = db = dx = dx2 = None
        want_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
        c1 = x.shape[-1]
        implicit = (want_w and k in (1, 3) and stride == 1 and not upsample and cin_tot % 8 == 0 and cout % 8 == 0 and IMPLICIT_WGRAD
                    and (not has_x2 or (c1 % 8 == 0 and x2.shape[-1] % 8 == 0)))
        if implicit:
            # the NHWC weight-gradient kernel (no im2col matrix, no transposed copies).  Its fp32 partials grow with the number of
            # (image, strip, row range) splits and one image must stay below 2 GB: beyond either limit it reports
            # GLARE_ERR_UNSUPPORTED / needs a workspace past WGRAD_MAX_WORKSPACE, and the im2col + GEMM form below takes over
            try:
                # straight into the filter's own layout (round 6: the reduction of the kernel's partials writes OIHW, the bias row into
                # db): autograd takes a gradient laid out like its parameter over as it is -- as a permuted view of [row][cout] it
                # was cloned by one copy_ launch per parameter, and a two-source conv needed a torch.cat on top
                dw = torch.empty(cout, cin_tot, k, k, dtype=torch.float32, device=x.device)
                db = torch.empty(cout, dtype=torch.float32, device=x.device) if has_bias else None
                T.conv_weight_grad_oihw(k, x, g16, cout, dw, 0, db)
                if has_x2:      # torch.cat((x, x2), 1) as the conv's input: the filter's input-channel blocks, one launch per source
                    T.conv_weight_grad_oihw(k, x2, g16, cout, dw, c1, None)
            except _lib.GlareError:
                implicit = False
        if implicit:
            pass
        elif want_w:
            kk = k * k

            def build(ldp, ones_row):
                col = T.im2col_t(x, k, stride, upsample=upsample, ldp=ldp, ones_row=ones_row, rows=cin_tot * kk + 1)
                if has_x2:
                    T.im2col_t(x2, k, stride, upsample=upsample, ldp=ldp, col=col, row_base=c1 * kk)
                return col

            dwb = T.conv_weight_grad(build, g16, cout, cin_tot * kk)
            dw = dwb[:, :-1].unflatten(1, (cin_tot, k, k))      # views of the GEMM output: no copy
            db = dwb[:, -1] if has_bias else None
        if ctx.needs_input_grad[0] or (has_x2 and ctx.needs_input_grad[4]):
            dxa = _data_grad(g16, weight, cout, stride, upsample)
            if has_x2:
                c1 = x.shape[-1]
                dx, dx2 = dxa[..., :c1].contiguous(), dxa[..., c1:].contiguous()
            else:
                dx = dxa
        return dx, dw, db, dres, dx2, None, None, None, None


def conv2d(x, weight, bias=None, residual=None, x2=None, stride=1, upsample=False, act="none", out_f32=False):
    return Conv2dFn.apply(x, weight, bias, residual, x2, stride, upsample, act, out_f32)


@_lib.keeps_precision
class SmallConv2dFn(torch.autograd.Function):
    """Convs whose input has <= 4 channels (conv_in on the image, cond / color convs, quant convs): x fp32 read through
    element strides (NCHW image or NHWC latent)."""

    @staticmethod
    def forward(ctx, x, weight, bias, layout, act, out_f32):
        if layout == "nchw":
            B, C, H, W = x.shape
            strides = (C * H * W, H * W, W, 1)
        else:
            B, H, W, C = x.shape
            strides = (H * W * C, 1, W * C, C)
        y = ops.conv2d_smallcin(x, strides, (B, H, W), weight, bias, act=act, out_f32=out_f32)
        ctx.cfg = (layout, act, strides, (B, H, W), bias is not None)
        ctx.save_for_backward(x, weight, y if act != "none" else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        layout, act, strides, bhw, has_bias = ctx.cfg
        x, weight, y = ctx.saved_tensors
        cout, cin, k, _ = weight.shape
        g16 = _grad_bf16(gy, y, act, cout)
        dw = db = dx = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dwb = T.conv_weight_grad(lambda ldp, ones_row: T.im2col_t_f32(x, strides, bhw, cin, k, k // 2, ldp=ldp, ones_row=ones_row),
                                     g16, cout, cin * k * k)
            dw = dwb[:, :-1].unflatten(1, (cin, k, k))
            db = dwb[:, -1] if has_bias else None
        if ctx.needs_input_grad[0]:
            assert layout == "nhwc", "the image needs no gradient"
            dx = _data_grad(g16, weight, cout, 1, False, out_f32=True)
        return dx, dw, db, None, None, None


def conv2d_small(x, weight, bias=None, layout="nhwc", act="none", out_f32=False):
    return SmallConv2dFn.apply(x, weight, bias, layout, act, out_f32)


@_lib.keeps_precision
class GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, swish, eps):
        y, stats = T.groupnorm_forward(x, gamma.detach().float(), beta.detach().float(), swish, eps)
        ctx.cfg = (swish, eps)
        ctx.save_for_backward(x, stats, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, gy):
        swish, eps = ctx.cfg
        x, stats, gamma, beta = ctx.saved_tensors
        dx, dgamma, dbeta = T.groupnorm_backward(x, gy.contiguous(), stats, gamma.detach().float(), beta.detach().float(), swish, eps)
        return dx, dgamma, dbeta, None, None


def groupnorm(x, gamma, beta, swish=True, eps=1e-6):
    return GroupNormFn.apply(x, gamma, beta, swish, eps)


# Attention backward: "fused" = csrc/attn_bwd.hip (no N^2 tensor), "materialised" = gemm_nt + softmax2_rows + attn_ds with the N^2
# scores in HBM, "auto" = fused when its grid fills the chip (B * ceil(N / 64) >= 160 workgroups, one per CU: 1.59 vs 1.75 ms at the
# stage-2 latent, 2 x 6400 tokens) or when the scores of one sample would not fit comfortably (N > 8192: 4 x N^2 x 2-4 B), else
# materialised (64 workgroups at the stage-3 latent would leave 3/4 of the CUs idle).
FUSED_ATTENTION_BACKWARD = "auto"


def _use_fused_attention_backward(B, N):
    if FUSED_ATTENTION_BACKWARD in (True, "fused"):
        return True
    if FUSED_ATTENTION_BACKWARD in (False, "materialised"):
        return False
    return B * ((N + 63) // 64) >= 160 or N > 8192


@_lib.keeps_precision
class AttentionFn(torch.autograd.Function):
    """softmax(q k^T) v with q already carrying scale*log2(e) (base-2 logits): q, k, v bf16 [B, N, 512]."""

    @staticmethod
    def forward(ctx, q, k, v):
        B, N, d = q.shape
        assert d == 512 and q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
        vt = T.transpose(v, _rup(N, 64))
        lse = torch.empty(B, N, dtype=torch.float32, device=q.device) if _use_fused_attention_backward(B, N) else None
        o = ops.attention_d512(q, k, vt, N, lse=lse)
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, go):
        q, k, v, o, lse = ctx.saved_tensors
        if lse is not None:
            return T.attention_backward_fused(q, k, v, o, go.contiguous(), lse)
        return T.attention_backward(q, k, v, o, go.contiguous())


def attention(q, k, v):
    return AttentionFn.apply(q, k, v)


@_lib.keeps_precision
class MixFn(torch.autograd.Function):
    """Mix.forward (deformableDecoder_arch.py:587-590): sigmoid(w) a + (1 - sigmoid(w)) b."""

    @staticmethod
    def forward(ctx, a, b, w):
        ctx.save_for_backward(a, b, w)
        return ops.mix(a, b, w)                        # w stays on the device: no host synchronisation

    @staticmethod
    def backward(ctx, g):
        a, b, w = ctx.saved_tensors
        ga, gb, dw = T.mix_backward(g.contiguous(), a, b, w, want_ga=ctx.needs_input_grad[0])
        return ga, gb, dw.to(w.dtype).view_as(w)


def mix(a, b, w):
    return MixFn.apply(a, b, w)


@_lib.keeps_precision
class MeanRescaleFn(torch.autograd.Function):
    """h + x_w * (mean(h) / mean(x_w)) (deformableDecoder_arch.py:567); h bf16, x_w fp32."""

    @staticmethod
    def forward(ctx, h, xw, whole_batch):
        ctx.whole = whole_batch
        ctx.save_for_backward(h, xw)
        return ops.mean_rescale(h, xw, whole_batch=whole_batch)

    @staticmethod
    def backward(ctx, g):
        h, xw = ctx.saved_tensors
        gh, gxw = T.mean_rescale_backward(g, h, xw, ctx.whole)
        return gh, gxw, None


def mean_rescale(h, xw, whole_batch):
    return MeanRescaleFn.apply(h, xw, whole_batch)


@_lib.keeps_precision
class DcnFn(torch.autograd.Function):
    """DCNv2 (DCNv2Pack.forward, deformableDecoder_arch.py:141-152) on the NHWC path: x bf16 NHWC, om = conv_offset's
    output fp32 NHWC [B,H,W,3*dg*9] (offsets | mask logits; the chunk/cat of :146-147 is the identity on that order and
    the sigmoid is applied by the kernels).  Backward = glare_mdcn_backward_f32, the drop-in of
    modulated_deform_conv_cuda_backward, which always produces all five gradients (deform_conv.py:161-165)."""

    @staticmethod
    def forward(ctx, x, om, weight, bias, dg, padding):
        B, H, W, _ = x.shape
        om_p = ops.nhwc_to_nchw(om.contiguous())                               # planar [B, 108, H, W]
        pd = ops.PackedDcn(weight, bias, dg)
        out = ops.mdcn_forward_nhwc(x, om_p.view(B, om_p.shape[1], H * W), pd, mask_is_logit=True, padding=padding)
        ctx.cfg = (dg, padding, bias is not None)
        ctx.save_for_backward(x, om_p, weight, bias)
        return out

    @staticmethod
    def backward(ctx, g):
        from .modules.ops.dcn.deform_conv import deform_conv_ext

        dg, padding, with_bias = ctx.cfg
        x, om_p, weight, bias = ctx.saved_tensors
        B, H, W, C = x.shape
        co, _, kh, kw = weight.shape
        n_off = 2 * dg * kh * kw
        xin = ops.nhwc_to_nchw(x)
        offset = om_p[:, :n_off].contiguous()
        mask = T.sigmoid(om_p[:, n_off:].contiguous())
        gout = ops.nhwc_to_nchw(g.contiguous())
        w32 = weight.detach().float().contiguous()
        b32 = bias.detach().float().contiguous() if with_bias else xin.new_empty(1)
        # the reference always scatters grad_input (deform_conv.py:161-165); here it is produced only when the sampled
        # feature needs it (in stage 3 x is the frozen VQ decoder's feature): a NULL grad_input skips the atomics pass
        gin = torch.zeros_like(xin) if ctx.needs_input_grad[0] else None
        goff, gmask = torch.zeros_like(offset), torch.zeros_like(mask)
        gw, gb = torch.zeros_like(w32), torch.zeros_like(b32)
        deform_conv_ext.modulated_deform_conv_backward(xin, w32, b32, xin.new_empty(0), offset, mask, xin.new_empty(0), gin, gw, gb,
                                                       goff, gmask, gout, kh, kw, 1, 1, padding, padding, 1, 1, 1, dg, with_bias)
        T.act_backward_(gmask, mask, "sigmoid")                               # d / d mask logit
        gom = torch.empty(B, H, W, om_p.shape[1], dtype=torch.float32, device=x.device)
        ops.nchw_to_nhwc(goff, bf16=False, out=gom, out_off=0)
        ops.nchw_to_nhwc(gmask, bf16=False, out=gom, out_off=n_off)
        gx = ops.nchw_to_nhwc(gin, bf16=True) if ctx.needs_input_grad[0] else None
        return gx, gom, gw, (gb if with_bias else None), None, None


def dcn(x, om, weight, bias, dg, padding=1):
    return DcnFn.apply(x, om, weight, bias, dg, padding)


@_lib.keeps_precision
class L1ClampLossFn(torch.autograd.Function):
    """l1_loss of VQLLFLOWDModel.optimize_parameters (VQLLFLOWD_model.py:209-217) on the NHWC output."""

    @staticmethod
    def forward(ctx, rec, gt_nchw):
        loss, grad = T.l1_clamp_loss(rec.contiguous(), gt_nchw.float().contiguous())
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None     # g is the scalar 1 of loss.backward(): an image-sized (3-channel) scale


def l1_clamp_loss(rec, gt_nchw):
    return L1ClampLossFn.apply(rec, gt_nchw)


@_lib.keeps_precision
class Clamp01Fn(torch.autograd.Function):
    """sr = rec.clamp(0, 1); sr[isnan] = 0 (VQLLFLOWD_model.py:209-215)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return T.clamp01(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return T.clamp01_backward(x, g)


def clamp01(x):
    return Clamp01Fn.apply(x)


@_lib.keeps_precision
class MaxPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return T.maxpool2(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return T.maxpool2_backward(x, g)


def maxpool2(x):
    return MaxPool2Fn.apply(x)


@_lib.keeps_precision
class MseFn(torch.autograd.Function):
    """F.mse_loss(a, b) of 16-bit feature maps; only `a` is differentiated.  The gradient 2 (a - b) / n * g is formed in fp32 in
    backward, where the upstream g (with the GradScaler's scale under fp16) is known, and rounded to 16 bits once -- as the
    reference's autocast computes mse_loss in fp32 (losses.py:33-39); rounded before the scale, 2 d / n at n ~ 1e6 is an fp16
    subnormal."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        loss, _ = T.mse_loss(a, b, want_grad=False)
        if ctx.needs_input_grad[0]:        # only then is there a backward that reads them: a feature map per VGG tap otherwise kept for nothing
            ctx.save_for_backward(a, b)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[0]:
            return None, None
        a, b = ctx.saved_tensors
        return T.mse_backward(a, b, g.detach().float().reshape(1).contiguous()), None


def mse_loss(a, b):
    return MseFn.apply(a, b)


@_lib.keeps_precision
class MSSSIMTermsFn(torch.autograd.Function):
    """The ten level scalars of msssim() (pytorch_msssim/__init__.py:71-83): (sim[5], cs[5]) of sr vs gt, NHWC fp32, value range
    1 (sr is clamped to [0, 1], so ssim()'s data-dependent `L` is 1).  Backward: gradients of the ten scalars -> d / d sr."""

    @staticmethod
    def forward(ctx, x, y, windows):
        xs, ys, moms, outs = [x.contiguous()], [y.contiguous()], [], []
        C1, C2 = 0.01 ** 2, 0.03 ** 2
        for lvl in range(5):
            mom, out = T.ssim_forward(xs[lvl], ys[lvl], windows[lvl], C1, C2)
            moms.append(mom)
            outs.append(out)
            if lvl < 4:
                xs.append(T.avgpool2(xs[lvl]))
                ys.append(T.avgpool2(ys[lvl]))
        ctx.windows = windows
        ctx.save_for_backward(*xs, *ys, *moms)
        o = torch.stack(outs)                      # [5, 2]: ten scalars
        return o[:, 0].contiguous(), o[:, 1].contiguous()

    @staticmethod
    def backward(ctx, g_sim, g_cs):
        t = ctx.saved_tensors
        xs, ys, moms = t[0:5], t[5:10], t[10:15]
        g10 = torch.cat([g_sim.float().reshape(5), g_cs.float().reshape(5)]).contiguous()
        g_next = None
        for lvl in reversed(range(5)):
            g_next = T.ssim_backward(xs[lvl], ys[lvl], moms[lvl], ctx.windows[lvl], 0.01 ** 2, 0.03 ** 2, g10, lvl, g_next)
        return g_next, None, None


def msssim_terms(x, y, windows):
    return MSSSIMTermsFn.apply(x, y, windows)


@_lib.keeps_precision
class NhwcToNchwFn(torch.autograd.Function):
    """fp32 NHWC -> fp32 NCHW at the module surface, differentiable (backward = the opposite layout kernel)."""

    @staticmethod
    def forward(ctx, x):
        return ops.nhwc_to_nchw(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ops.nchw_to_nhwc(g.contiguous(), bf16=False)


def nhwc_to_nchw(x):
    return NhwcToNchwFn.apply(x)


def to_nhwc_f32(x_nchw):
    return ops.nchw_to_nhwc(x_nchw, bf16=False)


@_lib.keeps_precision
class ForkFn(torch.autograd.Function):
    """A fan-out point of the tape made explicit: returns n aliases of x; the backward sums their gradients with
    glare_add_bf16, so the accumulation of activation gradients runs on this library, not on the autograd engine's add."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        acc = gs[0]
        i = 1
        while i < len(gs):
            acc = T.add_bf16(acc, gs[i], gs[i + 1] if i + 1 < len(gs) else None)
            i += 2
        return acc, None


def fork(x, n=2):
    return ForkFn.apply(x, n)
