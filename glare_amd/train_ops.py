"""ctypes wrappers of the training-step kernels (backward of the hot path: rows a12 / a13 of SURVEY.md section 8).

Same rules as ops.py: HIP only, no fallback, CPU tensors raise NotImplementedError.
"""
import ctypes

import torch

from . import _lib
from ._lib import act_dtype, check, ptr, require_cuda, stream_handle

from .ops import count_flops as _count_flops

_i = ctypes.c_int
_ll = ctypes.c_longlong
_f = ctypes.c_float


def gemm_nt(a, b, out=None, alpha=1.0, out_dtype=torch.float32, accumulate=False, M=None, N=None, K=None):
    """C[.., m, n] = alpha * sum_k a[.., m, k] * b[.., n, k]  (+ C).  a, b: bf16, 2-D or 3-D (batched), last dim
    contiguous; rows may be strided (views of wider buffers are fine)."""
    require_cuda(a, b, out)
    assert a.dtype == act_dtype() and b.dtype == act_dtype() and a.stride(-1) == 1 and b.stride(-1) == 1
    if a.dim() == 2:
        a, b = a.unsqueeze(0), b.unsqueeze(0)
        squeeze = True
    else:
        squeeze = False
    batch = a.shape[0]
    M = a.shape[1] if M is None else M
    N = b.shape[1] if N is None else N
    K = a.shape[2] if K is None else K
    if out is None:
        out = torch.empty(batch, M, N, dtype=out_dtype, device=a.device)
        if accumulate:
            out.zero_()
    o3 = out if out.dim() == 3 else out.unsqueeze(0)
    assert o3.stride(-1) == 1 and o3.dtype in (torch.float32, act_dtype())
    _count_flops("gemm_nt", 2.0 * batch * M * N * K)
    check(_lib.lib().glare_gemm_nt_bf16(ptr(a), ptr(b), ptr(o3), _i(M), _i(N), _i(K), _ll(a.stride(1)), _ll(b.stride(1)),
                                        _ll(o3.stride(1)), _i(batch), _ll(a.stride(0) if batch > 1 else 0),
                                        _ll(b.stride(0) if batch > 1 else 0), _ll(o3.stride(0) if batch > 1 else 0),
                                        _f(alpha), _i(int(o3.dtype == act_dtype())), _i(int(accumulate)), stream_handle()),
          "glare_gemm_nt_bf16")
    return out[0] if squeeze and out.dim() == 3 else out


def reduce_parts(parts, scale=1.0, out=None, accumulate=False):
    """parts fp32 [S, ...] -> sum over S (deterministic)."""
    require_cuda(parts, out)
    assert parts.dtype == torch.float32 and parts.is_contiguous()
    S = parts.shape[0]
    n = parts[0].numel()
    if out is None:
        out = torch.empty(parts.shape[1:], dtype=torch.float32, device=parts.device)
    check(_lib.lib().glare_reduce_parts_f32(ptr(parts), _i(S), _ll(n), _f(scale), ptr(out), _i(int(accumulate)), stream_handle()),
          "glare_reduce_parts_f32")
    return out


_sz = ctypes.c_size_t


def _rup(a, b):
    return (a + b - 1) // b * b


def transpose(x2d, ld_out=None, out=None):
    """bf16 [.., R, C] (row stride arbitrary, last dim contiguous) -> [.., C, ld_out] with zero-filled tail columns."""
    require_cuda(x2d, out)
    assert x2d.dtype == act_dtype() and x2d.stride(-1) == 1
    x3 = x2d if x2d.dim() == 3 else x2d.unsqueeze(0)
    batch, R, C = x3.shape
    ld_out = _rup(R, 64) if ld_out is None else ld_out
    if out is None:
        out = torch.empty(batch, C, ld_out, dtype=act_dtype(), device=x2d.device)
    o3 = out if out.dim() == 3 else out.unsqueeze(0)
    check(_lib.lib().glare_transpose_bf16(ptr(x3), _ll(x3.stride(1)), _ll(x3.stride(0) if batch > 1 else 0), ptr(o3), _ll(o3.stride(1)),
                                          _ll(o3.stride(0) if batch > 1 else 0), _ll(R), _i(C), _i(batch), stream_handle()),
          "glare_transpose_bf16")
    return out if x2d.dim() == 3 else o3[0]


def im2col_t(x, ksize, stride=1, pad=None, upsample=False, cin=None, in_off=0, ldp=None, col=None, row_base=0, ones_row=-1,
             rows=None):
    """x bf16 NHWC [B,H,W,pitch] -> colT [rows, ldp] bf16 (row = c*k*k + tap), see include/glare_hip.h."""
    require_cuda(x, col)
    assert x.dtype == act_dtype() and x.is_contiguous()
    B, H, W, pitch = x.shape
    cin = pitch - in_off if cin is None else cin
    pad = (1 if (ksize == 3 and stride == 1) else 0) if pad is None else pad
    IH, IW = (2 * H, 2 * W) if upsample else (H, W)
    OH, OW = ((IH + 1 - ksize) // 2 + 1, (IW + 1 - ksize) // 2 + 1) if stride == 2 else (IH + 2 * pad - ksize + 1, IW + 2 * pad - ksize + 1)
    P = B * OH * OW
    ldp = _rup(P, 64) if ldp is None else ldp
    if col is None:
        nrows = (cin * ksize * ksize + (1 if ones_row >= 0 else 0)) if rows is None else rows
        col = torch.empty(nrows, ldp, dtype=act_dtype(), device=x.device)
    check(_lib.lib().glare_im2col_t_bf16(ptr(x), _i(B), _i(H), _i(W), _i(pitch), _i(in_off), _i(cin), _i(ksize), _i(stride), _i(pad),
                                         _i(int(upsample)), ptr(col), _ll(ldp), _i(row_base), _i(ones_row), stream_handle()),
          "glare_im2col_t_bf16")
    return col


def im2col_t_f32(x, strides, shape_bhw, cin, ksize, pad, ldp=None, ones_row=-1):
    require_cuda(x)
    assert x.dtype == torch.float32
    B, H, W = shape_bhw
    P = B * H * W
    ldp = _rup(P, 64) if ldp is None else ldp
    col = torch.empty(cin * ksize * ksize + (1 if ones_row >= 0 else 0), ldp, dtype=act_dtype(), device=x.device)
    sb, sc, sy, sx = strides
    check(_lib.lib().glare_im2col_t_f32(ptr(x), _ll(sb), _ll(sc), _ll(sy), _ll(sx), _i(B), _i(H), _i(W), _i(cin), _i(ksize), _i(pad),
                                        ptr(col), _ll(ldp), _i(0), _i(ones_row), stream_handle()), "glare_im2col_t_f32")
    return col


def dilate2(g):
    require_cuda(g)
    assert g.dtype == act_dtype() and g.is_contiguous()
    B, OH, OW, C = g.shape
    out = torch.empty(B, 2 * OH, 2 * OW, C, dtype=act_dtype(), device=g.device)
    check(_lib.lib().glare_dilate2_bf16(ptr(g), ptr(out), _i(B), _i(OH), _i(OW), _i(C), stream_handle()), "glare_dilate2_bf16")
    return out


def pool2_sum(g):
    require_cuda(g)
    assert g.dtype == act_dtype() and g.is_contiguous()
    B, H2, W2, C = g.shape
    out = torch.empty(B, H2 // 2, W2 // 2, C, dtype=act_dtype(), device=g.device)
    check(_lib.lib().glare_pool2_sum_bf16(ptr(g), ptr(out), _i(B), _i(H2 // 2), _i(W2 // 2), _i(C), stream_handle()),
          "glare_pool2_sum_bf16")
    return out


def act_backward_(g, y, act, C=None, g_off=0, y_off=0):
    """In place: g *= act'(y); g, y [.., pitch] bf16 or fp32."""
    require_cuda(g, y)
    from .ops import ACT
    C = g.shape[-1] - g_off if C is None else C
    pixels = g.numel() // g.shape[-1]
    check(_lib.lib().glare_act_backward(ptr(g), _i(int(g.dtype == torch.float32)), _i(g.shape[-1]), _i(g_off), ptr(y),
                                        _i(int(y.dtype == torch.float32)), _i(y.shape[-1]), _i(y_off), _ll(pixels), _i(C),
                                        _i(ACT[act]), stream_handle()), "glare_act_backward")
    return g


def cast_to_bf16(x, pitch=None):
    """fp32 [.., C] -> bf16 [.., pitch] (extra channels zero)."""
    require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    C = x.shape[-1]
    pitch = C if pitch is None else pitch
    out = (torch.zeros if pitch != C else torch.empty)(*x.shape[:-1], pitch, dtype=act_dtype(), device=x.device)
    check(_lib.lib().glare_cast_f32_bf16(ptr(x), _i(C), _i(0), ptr(out), _i(pitch), _i(0), _ll(x.numel() // C), _i(C), stream_handle()),
          "glare_cast_f32_bf16")
    return out


def cast_to_f32(x, C=None):
    require_cuda(x)
    assert x.dtype == act_dtype() and x.is_contiguous()
    pitch = x.shape[-1]
    C = pitch if C is None else C
    out = torch.empty(*x.shape[:-1], C, dtype=torch.float32, device=x.device)
    check(_lib.lib().glare_cast_bf16_f32(ptr(x), _i(pitch), _i(0), ptr(out), _i(C), _i(0), _ll(x.numel() // pitch), _i(C), stream_handle()),
          "glare_cast_bf16_f32")
    return out


def groupnorm_forward(x, gamma, beta, swish=True, eps=1e-6):
    """Training-mode GroupNorm: returns (y, stats) with stats = the [B][splits][32][2] block the backward consumes."""
    require_cuda(x, gamma, beta)
    assert x.dtype == act_dtype() and x.is_contiguous()
    B, H, W, C = x.shape
    lib = _lib.lib()
    lib.glare_groupnorm_workspace_bytes.restype = _sz
    nws = lib.glare_groupnorm_workspace_bytes(_i(B), _ll(H * W))
    stats = torch.empty(nws // 4, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    check(lib.glare_groupnorm_swish_bf16(ptr(x), _i(C), _i(0), ptr(gamma), ptr(beta), ptr(y), _i(B), _ll(H * W), _i(C), _f(eps),
                                         _i(int(swish)), ptr(stats), _sz(nws), stream_handle()), "glare_groupnorm_swish_bf16")
    return y, stats.view(B, -1, 32, 2)


def groupnorm_backward(x, dy, stats, gamma, beta, swish=True, eps=1e-6):
    """-> (dx bf16, dgamma fp32 [C], dbeta fp32 [C])."""
    require_cuda(x, dy, stats, gamma, beta)
    assert x.dtype == dy.dtype == act_dtype() and x.is_contiguous() and dy.is_contiguous()
    B, H, W, C = x.shape
    lib = _lib.lib()
    lib.glare_groupnorm_backward_workspace_bytes.restype = _sz
    nws = lib.glare_groupnorm_backward_workspace_bytes(_i(B), _ll(H * W), _i(C))
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=x.device)
    dx = torch.empty_like(x)
    per_image = torch.empty(B, 2, C, dtype=torch.float32, device=x.device)
    check(lib.glare_groupnorm_swish_backward_bf16(ptr(x), _i(C), _i(0), ptr(dy), ptr(stats), _i(stats.shape[1]), ptr(gamma), ptr(beta),
                                                  ptr(dx), ptr(per_image), _i(B), _ll(H * W), _i(C), _f(eps), _i(int(swish)), ptr(ws),
                                                  _sz(nws), stream_handle()), "glare_groupnorm_swish_backward_bf16")
    s = reduce_parts(per_image)
    return dx, s[1], s[0]


def softmax2_rows(S, n, ldp=None):
    require_cuda(S)
    assert S.dtype == torch.float32 and S.dim() == 2 and S.stride(1) == 1
    rows = S.shape[0]
    ldp = _rup(n, 64) if ldp is None else ldp
    P = torch.empty(rows, ldp, dtype=act_dtype(), device=S.device)
    check(_lib.lib().glare_softmax2_rows_f32(ptr(S), _ll(S.stride(0)), ptr(P), _ll(ldp), _ll(rows), _i(n), stream_handle()),
          "glare_softmax2_rows_f32")
    return P


def attention_ds(P, dP, dO, O, n, scale):
    require_cuda(P, dP, dO, O)
    rows, d = dO.shape
    dS = torch.empty(rows, P.shape[1], dtype=act_dtype(), device=P.device)
    check(_lib.lib().glare_attention_ds_bf16(ptr(P), _ll(P.stride(0)), ptr(dP), _ll(dP.stride(0)), ptr(dO), _i(dO.stride(0)), ptr(O),
                                             _i(O.stride(0)), _i(d), ptr(dS), _ll(dS.stride(0)), _ll(rows), _i(n), _f(scale),
                                             stream_handle()), "glare_attention_ds_bf16")
    return dS


def adam_step_(w, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    require_cuda(w, grad, exp_avg, exp_avg_sq)
    for t in (w, grad, exp_avg, exp_avg_sq):
        assert t.dtype == torch.float32 and t.is_contiguous()
    check(_lib.lib().glare_adam_step_f32(ptr(w), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), _ll(w.numel()), _f(lr), _f(betas[0]),
                                         _f(betas[1]), _f(eps), _f(weight_decay), _i(step), _f(grad_scale), stream_handle()),
          "glare_adam_step_f32")
    return w


# ---- composite backward passes ------------------------------------------------------------------------
# split-K of the GEMM-form weight gradients (1x1, strided and upsampling convs): about two workgroups per CU, each at least
# 2048 pixels deep -- more slices cost more in fp32 partials than they gain in occupancy (measured, tools/kbench.py wgrad)
WGRAD_TARGET_WGS = 512
WGRAD_MIN_K = 2048
def conv_weight_grad(x_col_builder, g_bf16, cout, n_rows, out=None):
    """dW|db = gO^T . colT^T by split-K NT GEMM.  x_col_builder(ldp, ones_row) -> colT [n_rows+1, ldp];
    g_bf16: bf16 [B,OH,OW,pitch>=cout] or a 2-D (row-strided) view [P, >=cout].  Returns fp32 [cout, n_rows+1]
    (last column = bias gradient)."""
    if g_bf16.dim() != 2:
        g_bf16 = g_bf16.view(-1, g_bf16.shape[-1])
    P = g_bf16.shape[0]
    # the GEMM tile is 256 (m) x 128 (n): put the filter's Cout on the 128 side when it would waste half a 256-row tile
    swap = out is None and cout <= 128 and n_rows + 1 > 128
    tiles = ((n_rows + 256) // 256) * ((cout + 127) // 128) if swap else ((cout + 255) // 256) * ((n_rows + 1 + 127) // 128)
    S = max(1, min((WGRAD_TARGET_WGS + tiles - 1) // tiles, P // WGRAD_MIN_K, 64))
    ldp = _rup(P, 64 * S)
    colT = x_col_builder(ldp, n_rows)
    gT = transpose(g_bf16, ld_out=ldp)[:cout]
    ks = ldp // S
    a3 = gT.as_strided((S, cout, ks), (ks, ldp, 1))
    b3 = colT.as_strided((S, n_rows + 1, ks), (ks, ldp, 1))
    if swap:   # C^T = colT . gT^T; the caller gets the transposed view (gradients may be strided)
        parts = gemm_nt(b3, a3)
        return (reduce_parts(parts) if S > 1 else parts[0]).t()
    if S == 1:
        return gemm_nt(a3[0], b3[0], out=out) if out is not None else gemm_nt(a3, b3)[0]
    return reduce_parts(gemm_nt(a3, b3), out=out)


def attention_backward_fused(q, k, v, o, do, lse, ln2_scale=0.6931471805599453):
    """Flash-style backward (csrc/attn_bwd.hip): q (pre-scaled: q.k^T are base-2 logits), k, v, o, do bf16 [B, N, 512] dense, lse fp32
    [B, N] from the forward -> (dq, dk, dv) bf16.  No N^2 tensor: scores are recomputed per tile in two passes."""
    require_cuda(q, k, v, o, do, lse)
    B, N, d = q.shape
    assert d == 512 and all(t.is_contiguous() and t.dtype == act_dtype() for t in (q, k, v, o, do))
    assert lse.dtype == torch.float32 and lse.is_contiguous() and lse.numel() == B * N
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _count_flops("attn bwd", 10.0 * B * N * N * d)     # the five products of the backward (the kernel recomputes two more)
    lib = _lib.lib()
    lib.glare_attention_d512_backward_workspace_bytes.restype = ctypes.c_size_t
    nws = lib.glare_attention_d512_backward_workspace_bytes(_i(B), _i(N))
    ws = torch.empty(nws, dtype=torch.uint8, device=q.device)
    check(lib.glare_attention_d512_backward_bf16(ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(dq), ptr(dk), ptr(dv), _i(B), _i(N),
                                                 _f(ln2_scale), ptr(ws), _sz(nws), stream_handle()), "glare_attention_d512_backward_bf16")
    return dq, dk, dv


def attention_backward(q, k, v, o, do, ln2_scale=0.6931471805599453):
    """q (pre-scaled so that q.k^T are base-2 logits), k, v, o, do: bf16 [B, N, 512] -> (dq, dk, dv) bf16.
    Materialised form, one sample at a time: N^2 scores live in HBM (82-164 MB at the training crops)."""
    B, N, d = q.shape
    npad = _rup(N, 64)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    for b in range(B):
        S = gemm_nt(q[b], k[b])                         # [N, N] fp32 base-2 logits
        P = softmax2_rows(S, N, npad)                   # [N, npad] bf16
        dP = gemm_nt(do[b], v[b], out=S)                # reuse the fp32 buffer
        dS = attention_ds(P, dP, do[b], o[b], N, ln2_scale)
        kT, qT, doT = transpose(k[b], npad), transpose(q[b], npad), transpose(do[b], npad)   # [d, npad]
        PT, dST = transpose(P[:, :N], npad), transpose(dS[:, :N], npad)                      # [N, npad]
        gemm_nt(dS, kT, out=dq[b], K=npad)
        gemm_nt(dST, qT, out=dk[b], K=npad)
        gemm_nt(PT, doT, out=dv[b], K=npad)
    return dq, dk, dv


# ---- flow (normal direction) backward -------------------------------------------------------------------
def _host3(M, t):
    return (ctypes.c_float * 9)(*[float(v) for v in M]), (ctypes.c_float * 3)(*[float(v) for v in t])


def flow_nll_backward(z, mean, g_logp):
    require_cuda(z, mean, g_logp)
    B = z.shape[0]
    gz, gmean = torch.empty_like(z), torch.empty_like(z)
    check(_lib.lib().glare_flow_nll_backward_f32(ptr(z), ptr(mean), ptr(g_logp), _i(B), _ll(z.numel() // 3 // B), ptr(gz), ptr(gmean),
                                                 stream_handle()), "glare_flow_nll_backward_f32")
    return gz, gmean


def flow_post_backward_(gz, z_pre, h4, g_logdet, eps, out=None):
    """-> gh4 bf16 [B,H,W,8] (4 used); gz[..., 1:] updated in place."""
    require_cuda(gz, z_pre, h4, g_logdet, out)
    B = gz.shape[0]
    gh4 = torch.empty(*gz.shape[:-1], 8, dtype=act_dtype(), device=gz.device) if out is None else out
    check(_lib.lib().glare_flow_fwd_post_backward_f32(ptr(gz), ptr(z_pre), ptr(h4), ptr(g_logdet), _i(B), _ll(gz.numel() // 3 // B),
                                                      _f(eps), ptr(gh4), stream_handle()), "glare_flow_fwd_post_backward_f32")
    return gh4


def flow_h1_backward_(gz, g, g_off, z_pre, wz, out=None):
    """gz[..., 0] += adjoint of the 1-channel conv; returns the filter gradient fp32 [64, 9]."""
    require_cuda(gz, g, z_pre, wz)
    B, H, W, _ = gz.shape
    lib = _lib.lib()
    nb = int(lib.glare_flow_bwd_blocks(_ll(B * H * W)))
    part = torch.empty(nb, 576, dtype=torch.float32, device=gz.device)
    check(lib.glare_flow_h1_backward_f32(ptr(gz), ptr(g), _i(g.shape[-1]), _i(g_off), ptr(z_pre), ptr(wz), _i(B), _i(H), _i(W), ptr(part),
                                         stream_handle()), "glare_flow_h1_backward_f32")
    return reduce_parts(part, out=None if out is None else out.view(576)).view(64, 9)


def flow_pre_backward_(gz, z_in, hF, hF_off, g_logdet, M, t, eps, ghF, ghF_off, out=None, Mt_dev=None):
    """gz updated in place to the step input's gradient; ghF slice written; returns (dM fp32 [3,3], dt fp32 [3])."""
    require_cuda(gz, z_in, hF, g_logdet, ghF, Mt_dev)
    B = gz.shape[0]
    lib = _lib.lib()
    npix = gz.numel() // 3
    nb = int(lib.glare_flow_bwd_blocks(_ll(npix)))
    part = torch.empty(nb, 12, dtype=torch.float32, device=gz.device)
    if Mt_dev is not None:
        check(lib.glare_flow_fwd_pre_backward_dev_f32(ptr(gz), ptr(z_in), ptr(hF), _i(hF.shape[-1]), _i(hF_off), ptr(g_logdet), _i(B),
                                                      _ll(npix // B), ptr(Mt_dev), _f(eps), ptr(ghF), _i(ghF.shape[-1]), _i(ghF_off),
                                                      ptr(part), stream_handle()), "glare_flow_fwd_pre_backward_dev_f32")
        r = reduce_parts(part, out=out)
        return r[:9].view(3, 3), r[9:]
    Ma, ta = _host3(M, t)
    check(lib.glare_flow_fwd_pre_backward_f32(ptr(gz), ptr(z_in), ptr(hF), _i(hF.shape[-1]), _i(hF_off), ptr(g_logdet), _i(B),
                                              _ll(npix // B), Ma, ta, _f(eps), ptr(ghF), _i(ghF.shape[-1]), _i(ghF_off), ptr(part),
                                              stream_handle()), "glare_flow_fwd_pre_backward_f32")
    r = reduce_parts(part, out=out)
    return r[:9].view(3, 3), r[9:]


# ---- a8 glue backward -----------------------------------------------------------------------------------
def mix_backward(g, a, b, w, want_ga=False):
    """-> (ga or None, gb, dw fp32 [1])."""
    require_cuda(g, a, b)
    assert g.dtype == a.dtype == b.dtype == act_dtype() and g.is_contiguous() and a.is_contiguous() and b.is_contiguous()
    gb = torch.empty_like(g)
    ga = torch.empty_like(g) if want_ga else None
    dw = torch.empty(1, dtype=torch.float32, device=g.device)
    ws = torch.empty(512, dtype=torch.float32, device=g.device)
    if torch.is_tensor(w):
        w32 = w.detach().float().reshape(1)
        check(_lib.lib().glare_mix_backward_dev_bf16(ptr(g), ptr(a), ptr(b), ptr(ga), ptr(gb), _ll(g.numel()), ptr(w32), ptr(dw), ptr(ws),
                                                     _sz(2048), stream_handle()), "glare_mix_backward_dev_bf16")
        return ga, gb, dw
    check(_lib.lib().glare_mix_backward_bf16(ptr(g), ptr(a), ptr(b), ptr(ga), ptr(gb), _ll(g.numel()), _f(float(w)), ptr(dw), ptr(ws),
                                             _sz(2048), stream_handle()), "glare_mix_backward_bf16")
    return ga, gb, dw


def mean_rescale_backward(g, h, xw, whole_batch):
    require_cuda(g, h, xw)
    assert g.dtype == h.dtype == act_dtype() and xw.dtype == torch.float32
    g, h, xw = g.contiguous(), h.contiguous(), xw.contiguous()
    B = h.shape[0]
    n = h.numel() // B
    lib = _lib.lib()
    lib.glare_mean_rescale_backward_workspace_bytes.restype = _sz
    nws = lib.glare_mean_rescale_backward_workspace_bytes(_i(B), _ll(n))
    ws = torch.empty(nws, dtype=torch.uint8, device=h.device)
    gh, gxw = torch.empty_like(h), torch.empty_like(xw)
    check(lib.glare_mean_rescale_backward_bf16(ptr(g), ptr(h), ptr(xw), ptr(gh), ptr(gxw), _i(B), _ll(n), _i(int(whole_batch)), ptr(ws),
                                               _sz(nws), stream_handle()), "glare_mean_rescale_backward_bf16")
    return gh, gxw


def sigmoid(x):
    require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().glare_sigmoid_f32(ptr(x), ptr(y), _ll(x.numel()), stream_handle()), "glare_sigmoid_f32")
    return y


def l1_clamp_loss(rec_nhwc, gt_nchw):
    """-> (loss fp32 [1], grad fp32 NHWC) of mean |clamp(rec,0,1) - gt| with the reference's NaN masking."""
    require_cuda(rec_nhwc, gt_nchw)
    assert rec_nhwc.dtype == gt_nchw.dtype == torch.float32 and rec_nhwc.is_contiguous() and gt_nchw.is_contiguous()
    B, H, W, C = rec_nhwc.shape
    assert tuple(gt_nchw.shape) == (B, C, H, W)
    loss = torch.empty(1, dtype=torch.float32, device=rec_nhwc.device)
    grad = torch.empty_like(rec_nhwc)
    ws = torch.empty(512, dtype=torch.float32, device=rec_nhwc.device)
    check(_lib.lib().glare_l1_clamp_loss_f32(ptr(rec_nhwc), ptr(gt_nchw), _i(B), _ll(H * W), _i(C), ptr(loss), ptr(grad), ptr(ws),
                                             _sz(2048), stream_handle()), "glare_l1_clamp_loss_f32")
    return loss, grad


# ---- stage-3 loss stack (row f1) ------------------------------------------------------------------------
def clamp01(x):
    require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().glare_clamp01_f32(ptr(x), ptr(y), _ll(x.numel()), stream_handle()), "glare_clamp01_f32")
    return y


def clamp01_backward(x, g):
    require_cuda(x, g)
    g = g.contiguous()
    gx = torch.empty_like(x)
    check(_lib.lib().glare_clamp01_backward_f32(ptr(x), ptr(g), ptr(gx), _ll(x.numel()), stream_handle()), "glare_clamp01_backward_f32")
    return gx


def avgpool2(x):
    require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    B, H, W, C = x.shape
    y = torch.empty(B, H // 2, W // 2, C, dtype=torch.float32, device=x.device)
    check(_lib.lib().glare_avgpool2_f32(ptr(x), ptr(y), _i(B), _i(H), _i(W), _i(C), stream_handle()), "glare_avgpool2_f32")
    return y


def _win(window):
    return (ctypes.c_float * len(window))(*[float(v) for v in window])


def ssim_forward(x, y, window, C1, C2):
    """x, y fp32 NHWC -> (moments [B,OH,OW,C,5], out fp32 [2] = (mean ssim_map, mean cs_map))."""
    require_cuda(x, y)
    B, H, W, C = x.shape
    ws = len(window)
    lib = _lib.lib()
    lib.glare_ssim_workspace_bytes.restype = _sz
    nws = lib.glare_ssim_workspace_bytes()
    wsb = torch.empty(nws, dtype=torch.uint8, device=x.device)
    mom = torch.empty(B, H - ws + 1, W - ws + 1, C, 5, dtype=torch.float32, device=x.device)
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    check(lib.glare_ssim_forward_f32(ptr(x), ptr(y), _i(B), _i(H), _i(W), _i(C), _win(window), _i(ws), _f(C1), _f(C2), ptr(mom), ptr(out),
                                     ptr(wsb), _sz(nws), stream_handle()), "glare_ssim_forward_f32")
    return mom, out


def ssim_backward(x, y, mom, window, C1, C2, g10, level, g_next):
    require_cuda(x, y, mom, g10, g_next)
    B, H, W, C = x.shape
    scratch = torch.empty(mom.shape[:-1] + (3,), dtype=torch.float32, device=x.device)
    gx = torch.empty_like(x)
    check(_lib.lib().glare_ssim_backward_f32(ptr(x), ptr(y), ptr(mom), _i(B), _i(H), _i(W), _i(C), _win(window), _i(len(window)), _f(C1),
                                             _f(C2), ptr(g10), _i(level), ptr(g_next), ptr(scratch), ptr(gx), stream_handle()),
          "glare_ssim_backward_f32")
    return gx


def maxpool2(x):
    require_cuda(x)
    assert x.dtype == act_dtype() and x.is_contiguous()
    B, H, W, C = x.shape
    y = torch.empty(B, H // 2, W // 2, C, dtype=act_dtype(), device=x.device)
    check(_lib.lib().glare_maxpool2_bf16(ptr(x), ptr(y), _i(B), _i(H), _i(W), _i(C), stream_handle()), "glare_maxpool2_bf16")
    return y


def maxpool2_backward(x, g):
    require_cuda(x, g)
    B, H, W, C = x.shape
    g = g.contiguous()
    gx = torch.empty_like(x)
    check(_lib.lib().glare_maxpool2_backward_bf16(ptr(x), ptr(g), ptr(gx), _i(B), _i(H), _i(W), _i(C), stream_handle()),
          "glare_maxpool2_backward_bf16")
    return gx


def mse_loss(a, b, want_grad=True):
    """bf16 feature maps -> (loss fp32 [1], grad_a bf16 or None)."""
    require_cuda(a, b)
    assert a.dtype == b.dtype == act_dtype() and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    ga = torch.empty_like(a) if want_grad else None
    ws = torch.empty(256, dtype=torch.float64, device=a.device)
    check(_lib.lib().glare_mse_loss_bf16(ptr(a), ptr(b), _ll(a.numel()), ptr(loss), ptr(ga), ptr(ws), _sz(2048), stream_handle()),
          "glare_mse_loss_bf16")
    return loss, ga


def mse_backward(a, b, g):
    """grad_a = round16(2 (a - b) / n * g); g: the upstream gradient, a one-element fp32 DEVICE tensor (no host read)."""
    require_cuda(a, b, g)
    assert a.dtype == b.dtype == act_dtype() and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    assert g.dtype == torch.float32 and g.numel() == 1
    ga = torch.empty_like(a)
    check(_lib.lib().glare_mse_backward_bf16(ptr(a), ptr(b), _ll(a.numel()), ptr(g), ptr(ga), stream_handle()), "glare_mse_backward_bf16")
    return ga


def adam_prepare_(step_dev, state3, betas):
    require_cuda(step_dev, state3)
    assert step_dev.dtype == torch.int32 and state3.dtype == torch.float32 and state3.numel() == 3
    check(_lib.lib().glare_adam_prepare(ptr(step_dev), ptr(state3), _f(betas[0]), _f(betas[1]), stream_handle()), "glare_adam_prepare")


def adam_step_dev_(w, grad, exp_avg, exp_avg_sq, state3, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    require_cuda(w, grad, exp_avg, exp_avg_sq, state3)
    check(_lib.lib().glare_adam_step_dev_f32(ptr(w), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), _ll(w.numel()), _f(lr), _f(betas[0]),
                                             _f(betas[1]), _f(eps), _f(weight_decay), ptr(state3), _f(grad_scale), stream_handle()),
          "glare_adam_step_dev_f32")
    return w


def grad_nonfinite_(grad, found):
    """found (int32 [1], device) |= any(grad is inf / NaN): GradScaler's found_inf without a host read."""
    require_cuda(grad, found)
    assert grad.dtype == torch.float32 and grad.is_contiguous() and found.dtype == torch.int32
    check(_lib.lib().glare_grad_nonfinite_f32(ptr(grad), _ll(grad.numel()), ptr(found), stream_handle()), "glare_grad_nonfinite_f32")


def adam_prepare_guarded_(step_dev, state3, betas, skip):
    require_cuda(step_dev, state3, skip)
    check(_lib.lib().glare_adam_prepare_guarded(ptr(step_dev), ptr(state3), _f(betas[0]), _f(betas[1]), ptr(skip), stream_handle()),
          "glare_adam_prepare_guarded")


def adam_step_dev_guarded_(w, grad, exp_avg, exp_avg_sq, state3, lr, betas, eps, weight_decay, grad_scale, skip, loss_scale=None):
    """loss_scale: fp32 device tensor [1] the loss was multiplied by (fp16 training: GradScaler's scale), divided out here."""
    require_cuda(w, grad, exp_avg, exp_avg_sq, state3, skip, loss_scale)
    if loss_scale is not None:
        check(_lib.lib().glare_adam_step_dev_scaled_f32(ptr(w), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), _ll(w.numel()), _f(lr),
                                                        _f(betas[0]), _f(betas[1]), _f(eps), _f(weight_decay), ptr(state3), _f(grad_scale),
                                                        ptr(loss_scale), ptr(skip), stream_handle()), "glare_adam_step_dev_scaled_f32")
        return w
    check(_lib.lib().glare_adam_step_dev_guarded_f32(ptr(w), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), _ll(w.numel()), _f(lr),
                                                     _f(betas[0]), _f(betas[1]), _f(eps), _f(weight_decay), ptr(state3), _f(grad_scale),
                                                     ptr(skip), stream_handle()), "glare_adam_step_dev_guarded_f32")
    return w


def gradscaler_update_(scale, tracker, found, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
    require_cuda(scale, tracker, found)
    assert scale.dtype == torch.float32 and tracker.dtype == torch.int32 and found.dtype == torch.int32
    check(_lib.lib().glare_gradscaler_update(ptr(scale), ptr(tracker), ptr(found), _f(growth_factor), _f(backoff_factor),
                                             _i(growth_interval), stream_handle()), "glare_gradscaler_update")


def actnorm_init_(x, C, bias, logs, off=0, scale=1.0):
    """ActNorm's data-dependent initialisation (FlowActNorms.py:32-46) from x fp32 [..., pitch] (channels [off, off + C)):
    writes bias / logs (fp32, C elements, any shape) in place."""
    require_cuda(x, bias, logs)
    assert x.dtype == torch.float32 and x.is_contiguous() and bias.dtype == logs.dtype == torch.float32
    assert bias.is_contiguous() and logs.is_contiguous() and bias.numel() == logs.numel() == C
    pitch = x.shape[-1]
    P = x.numel() // pitch
    lib = _lib.lib()
    lib.glare_actnorm_init_workspace_bytes.restype = ctypes.c_size_t
    nws = lib.glare_actnorm_init_workspace_bytes(_ll(P))
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
    check(lib.glare_actnorm_init_f32(ptr(x), _i(pitch), _i(off), _i(C), _ll(P), _f(scale), ptr(bias), ptr(logs), ptr(ws),
                                     ctypes.c_size_t(nws), stream_handle()), "glare_actnorm_init_f32")


def flow_h1_raw(z, ftA, ftA_off, wz):
    """The pre-activation of fAffine's first conv: ftA[:, off:off+64] + conv3x3(z[:, 0] -> 64), fp32 [B,H,W,64]."""
    require_cuda(z, ftA, wz)
    B, H, W, _ = z.shape
    out = torch.empty(B, H, W, 64, dtype=torch.float32, device=z.device)
    check(_lib.lib().glare_flow_h1_raw_f32(ptr(z), ptr(ftA), _i(ftA.shape[3]), _i(ftA_off), ptr(wz), ptr(out), _i(B), _i(H), _i(W),
                                           stream_handle()), "glare_flow_h1_raw_f32")
    return out


def flow_affine3_(z, M, t):
    """z = M z + t in place (fp32 [.., 3]); M (9 floats, row-major) and t (3 floats) are host sequences."""
    require_cuda(z)
    Ma = (ctypes.c_float * 9)(*[float(v) for v in M])
    ta = (ctypes.c_float * 3)(*[float(v) for v in t])
    check(_lib.lib().glare_flow_affine3_f32(ptr(z), _ll(z.numel() // 3), Ma, ta, stream_handle()), "glare_flow_affine3_f32")
    return z


def add_bf16(a, b, c=None):
    require_cuda(a, b, c)
    a, b = a.contiguous(), b.contiguous()
    c = None if c is None else c.contiguous()
    assert a.dtype == b.dtype == act_dtype() and a.shape == b.shape
    out = torch.empty_like(a)
    check(_lib.lib().glare_add_bf16(ptr(a), ptr(b), ptr(c), ptr(out), _ll(a.numel()), stream_handle()), "glare_add_bf16")
    return out


def colsum(g2d, C):
    """bf16 [P, pitch] -> fp32 [C] column sums (bias gradient)."""
    require_cuda(g2d)
    assert g2d.dtype == act_dtype() and g2d.dim() == 2 and g2d.stride(1) == 1
    out = torch.empty(C, dtype=torch.float32, device=g2d.device)
    ws = torch.empty(256 * C, dtype=torch.float32, device=g2d.device)
    check(_lib.lib().glare_colsum_bf16(ptr(g2d), _i(g2d.stride(0)), _ll(g2d.shape[0]), _i(C), ptr(out), ptr(ws), _sz(ws.numel() * 4),
                                       stream_handle()), "glare_colsum_bf16")
    return out


WGRAD_MAX_WORKSPACE = 2 << 30   # bytes of fp32 partials the NHWC weight-gradient kernel may ask for before callers fall back


def conv_weight_grad_nhwc(ksize, x, g16, cout, cin=None, in_off=0, groups=1, x_gstride=0, g_gstride=0, shape=None, out=None):
    """Weight + bias gradients of `groups` independent ksize x ksize (3 with pad 1, or 1) stride-1 convs straight from the NHWC
    operands (csrc/wgrad.hip).  x: bf16 NHWC [B,H,W,pitch] (cin channels at in_off, cin % 8 == 0); g16: bf16 NHWC
    [B,H,W,>=cout], cout % 8 == 0; group k reads x / g16 advanced by k * x_gstride / k * g_gstride elements (`shape` = (B,H,W) of
    one group when the tensors hold several).  Returns fp32 [groups, ksize^2*cin + 1, cout]: row (ty*ksize+tx)*cin + ci, last
    row = the bias gradient."""
    require_cuda(x, g16, out)
    assert x.dtype == act_dtype() and g16.dtype == act_dtype() and x.is_contiguous() and g16.is_contiguous()
    B, H, W = x.shape[:3] if shape is None else shape
    pitch, gpitch = x.shape[-1], g16.shape[-1]
    cin = pitch - in_off if cin is None else cin
    _count_flops("wgrad k%d" % ksize, 2.0 * groups * B * H * W * ksize * ksize * cin * cout)
    lib = _lib.lib()
    lib.glare_conv_wgrad_workspace_bytes.restype = ctypes.c_size_t
    nws = lib.glare_conv_wgrad_workspace_bytes(_i(ksize), _i(groups), _i(B), _i(H), _i(W), _i(cin), _i(cout))
    if nws > WGRAD_MAX_WORKSPACE:     # the split count grows with the batch (one workgroup per image, strip and row range)
        raise _lib.GlareError("glare_conv_wgrad_bf16: %.1f GB of fp32 partials (limit %.1f GB)" % (nws / 2 ** 30, WGRAD_MAX_WORKSPACE / 2 ** 30))
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=x.device)
    if out is None:
        out = torch.empty(groups, ksize * ksize * cin + 1, cout, dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == groups * (ksize * ksize * cin + 1) * cout
    check(lib.glare_conv_wgrad_bf16(_i(ksize), ptr(x), _i(pitch), _i(in_off), _ll(x_gstride), ptr(g16), _i(gpitch), _ll(g_gstride), ptr(out),
                                    _i(groups), _i(B), _i(H), _i(W), _i(cin), _i(cout), ptr(ws), _sz(nws), stream_handle()),
          "glare_conv_wgrad_bf16")
    return out


def conv_weight_grad_oihw(ksize, x, g16, cout, dw, ci_off=0, db=None):
    """The weight (and bias) gradient of ONE stride-1 conv straight into the filter's layout: dw fp32 [cout, ci_total, k, k] contiguous,
    this call filling input channels [ci_off, ci_off + x.shape[-1]); db fp32 [cout] or None (glare_conv_wgrad_oihw_bf16)."""
    require_cuda(x, g16, dw, db)
    assert x.dtype == act_dtype() and g16.dtype == act_dtype() and x.is_contiguous() and g16.is_contiguous()
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.shape[0] == cout and dw.shape[2] == dw.shape[3] == ksize
    B, H, W, cin = x.shape
    _count_flops("wgrad k%d" % ksize, 2.0 * B * H * W * ksize * ksize * cin * cout)
    lib = _lib.lib()
    lib.glare_conv_wgrad_oihw_workspace_bytes.restype = ctypes.c_size_t
    nws = lib.glare_conv_wgrad_oihw_workspace_bytes(_i(ksize), _i(B), _i(H), _i(W), _i(cin), _i(cout))
    if nws > WGRAD_MAX_WORKSPACE:
        raise _lib.GlareError("glare_conv_wgrad_oihw_bf16: %.1f GB of fp32 partials (limit %.1f GB)" % (nws / 2 ** 30, WGRAD_MAX_WORKSPACE / 2 ** 30))
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=x.device)
    check(lib.glare_conv_wgrad_oihw_bf16(_i(ksize), ptr(x), _i(cin), _i(0), ptr(g16), _i(g16.shape[-1]), ptr(dw), _i(dw.shape[1]), _i(ci_off), ptr(db),
                                         _i(B), _i(H), _i(W), _i(cin), _i(cout), ptr(ws), _sz(nws), stream_handle()), "glare_conv_wgrad_oihw_bf16")
    return dw


def conv3x3_weight_grad(x, g16, cout, cin=None, in_off=0):
    """-> (dW fp32 [cout,cin,3,3] (a permuted view), db fp32 [cout]) of one 3x3 / stride-1 / pad-1 conv."""
    cin = x.shape[-1] - in_off if cin is None else cin
    dwt = conv_weight_grad_nhwc(3, x, g16, cout, cin, in_off)[0]
    return dwt[:9 * cin].view(3, 3, cin, cout).permute(3, 2, 0, 1), dwt[9 * cin]


def conv1x1_weight_grad(x, g16, cout, cin=None, in_off=0):
    """-> (dW fp32 [cout,cin,1,1] (a transposed view), db fp32 [cout]) of one 1x1 conv."""
    cin = x.shape[-1] - in_off if cin is None else cin
    dwt = conv_weight_grad_nhwc(1, x, g16, cout, cin, in_off)[0]
    return dwt[:cin].t().unsqueeze(-1).unsqueeze(-1), dwt[cin]
