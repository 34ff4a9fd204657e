"""VQGAN encoder / decoder building blocks on HIP kernels.

Mirrors code/models/modules/encoder_decoder.py (ResnetBlock :78-137, AttnBlock :140-192, Upsample
:38-53, Downsample :56-75, Encoder :342-442, Decoder :445-551): same parameter names, same forward
signatures (NCHW fp32 in/out).  Internally activations are NHWC bf16; `forward_nhwc` chains modules
without layout conversions and is what the fused graphs use."""
import math
import os
import threading

import torch
import torch.nn as nn

from .. import autograd as A
from .. import ops
from ._base import HipModule, packed_conv, to_nchw, to_nhwc


GN_FUSED = True  # GroupNorm statistics gathered by the producing conv's epilogue (False: separate stats pass)


def Normalize(in_channels):  # encoder_decoder.py:34-35
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


def gn_swish(x, norm, swish=True, pair=False):
    return ops.groupnorm(x, norm.weight.detach().float(), norm.bias.detach().float(), swish=swish, eps=norm.eps, pair=pair)


def is_hilo(x):
    """Is this activation the hi half of a hi / lo pair (its remainder in `x._lo`)?  Encoder.hilo_stream starts the residual
    stream that way; every block then adds into it at 22 bits and hands the pair on (ops.conv2d(hilo=True))."""
    return getattr(x, "_lo", None) is not None


# Training mode (`train_nhwc` methods): the same graph through glare_amd.autograd, i.e. HIP forward + HIP backward
# with parameters taken live (packed per call) -- what `loss.backward()` differentiates in the reference.
def gn_swish_t(x, norm, swish=True):
    return A.groupnorm(x, norm.weight, norm.bias, swish=swish, eps=norm.eps)


def conv_t(x, conv, **kw):
    return A.conv2d(x, conv.weight, conv.bias, **kw)


# The conditional encoder's residual stream as hi / lo pairs under fp16 (Encoder.hilo_stream; GLARE_HILO_STREAM=0 switches it off for
# A/B measurements: the stream tensors are then plain 16-bit, as in every other network of the path).
HILO_STREAM = os.environ.get("GLARE_HILO_STREAM", "1") != "0"
# ... and, on top of the hi / lo stream, every MFMA contraction of the conditional encoder in the fp32-class form (round 4): the
# activation operand AND the filter as hi / lo pairs, x . w = x_hi . w_hi + x_lo . w_hi + x_hi . w_lo in one accumulation over three
# K segments (glare_conv_desc.k_wrap), every stored activation a hi / lo pair.  The reference runs these convs under fp16 autocast
# but is judged against its fp32 self (BASELINE: PSNR within 0.05 dB, indices bit-exact): tools/precision_sites.py shows the
# codebook search needs the latent to ~1e-4, which 11-bit MFMA operands miss 20-fold.  GLARE_FP32_CLASS=0:
# round 3's single-pass convs (A/B measurements).
FP32_CLASS = os.environ.get("GLARE_FP32_CLASS", "1") != "0"


_FP32_TLS = threading.local()     # a per-THREAD override of FP32_CLASS (None / unset: the module-level default above)


def fp32_class_on():
    """Is the fp32-class form in force for the calling thread?  The thread's own `with fp32_class(...)` if it is inside one, else the
    process default FP32_CLASS (the environment switch)."""
    v = getattr(_FP32_TLS, "value", None)
    return FP32_CLASS if v is None else v


class fp32_class:
    """`with fp32_class(False):` -- the conditional encoder and the flow's nets in the single-pass form inside the block (round 3's
    arithmetic = the reference's own fp16 autocast: one 16-bit MFMA pass per conv).  Used by Stage3Trainer for its FROZEN front: the
    fp32-class form exists for the inference contract (indices against the fp32 reference), which a training step does not have --
    the reference trains stage 3 with these nets under `@autocast()` (VQLLFLOWDeformable_arch.py:222).  THREAD-LOCAL (round 6; it
    used to flip the process-wide flag): an inference running on another thread or stream of the same process -- a validation pass
    beside a trainer -- keeps the index-parity form while a trainer's step is inside this block."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self._prev = getattr(_FP32_TLS, "value", None)
        _FP32_TLS.value = self.on
        return self

    def __exit__(self, *exc):
        _FP32_TLS.value = self._prev


# GroupNorm + swish in front of a ResnetBlock's 3x3 convs as the conv loader's PROLOGUE (glare_conv_desc.gn_coef; bit-identical to the
# separate apply pass).  Measured (tools/kbench.py gnpro, profiles/r04_gn_prologue.txt): -2 % on apply + conv at 128 channels, full
# resolution; +12 % / +20 % at 256 / 512 channels, where 2 / 4 output-channel tiles repeat the transform -- the in-LDS transform sits
# on each wave's critical path and costs half of the HBM pass it removes.  OFF by default (GLARE_GN_PROLOGUE=1: the <= 128-channel blocks).
GN_PROLOGUE = os.environ.get("GLARE_GN_PROLOGUE", "0") == "1"
SUBPIXEL_UPSAMPLE = True
FOLD_PROJ_INTO_V = True
# AttnBlock as attention with shared keys / values (csrc/attn.hip, attn_kv_fwd_kernel): the key projection folded into the
# query projection, the value projection folded into proj_out.  False: the q | k and v^T projections + the two-tensor kernel.
SHARED_KV_ATTENTION = True
# ... and with the block's GroupNorm folded into the two 1x1 convs as per-image filters (no activation follows that norm): the
# attention runs on the RAW input as keys / values and the normalised tensor is never written (ops.attn_fold_groupnorm).  Needs the
# producer's fused GroupNorm statistics on the input and a batch that divides the 1x1 kernel's 64 pixel ranges.
GN_FOLDED_ATTENTION = True


class Upsample(HipModule):
    def __init__(self, in_channels, with_conv=True):
        super().__init__()
        assert with_conv, "GLARE always resamples with a conv (resamp_with_conv=True)"
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)

    def forward_nhwc(self, x, **kw):
        # sub-pixel form of nearest x2 + 3x3 (four 2x2 convs of the source with pre-summed taps: 4/9 of the MACs, conv_igemm.hip);
        # GroupNorm statistics of the output fused in the epilogue.  SUBPIXEL_UPSAMPLE = False: upsample fused in the loader.
        if SUBPIXEL_UPSAMPLE:
            pc = self._packed(("conv_subpixel", id(self.conv)),
                              lambda: ops.PackedConv(self.conv.weight, self.conv.bias, upsample_subpixel=True))
        else:
            pc = packed_conv(self, self.conv)
        return ops.conv2d(x, pc, upsample=True, gn_stats=GN_FUSED and self.conv.out_channels % 128 == 0, **kw)

    def train_nhwc(self, x):
        return conv_t(x, self.conv, upsample=True)

    def forward(self, x):
        return to_nchw(self.forward_nhwc(to_nhwc(x)))


class Downsample(HipModule):
    def __init__(self, in_channels, with_conv=True):
        super().__init__()
        assert with_conv
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)

    def forward_nhwc(self, x, split=0):
        return ops.conv2d(x, packed_conv(self, self.conv, split=split), stride=2,  # pad (0,1,0,1) fused in the loader
                          gn_stats=GN_FUSED and self.conv.out_channels % 128 == 0, hilo=is_hilo(x))

    def train_nhwc(self, x):
        return conv_t(x, self.conv, stride=2)

    def forward(self, x):
        return to_nchw(self.forward_nhwc(to_nhwc(x)))


class ResnetBlock(HipModule):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0):
        super().__init__()
        assert temb_channels == 0 and not conv_shortcut and dropout == 0.0
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def forward_nhwc(self, x, out=None, out_off=0, split=0):
        fuse = GN_FUSED and self.out_channels % 128 == 0
        hl = is_hilo(x)      # the residual stream as a hi / lo pair (conditional encoder, fp16): x + h is added in 22 bits
        if split:            # fp32-class (FP32_CLASS): every operand and every stored activation of the block a hi / lo pair
            assert hl and out is None and split == 3
            h = ops.conv2d(gn_swish(x, self.norm1, pair=True), packed_conv(self, self.conv1, split=3), gn_stats=fuse, hilo=True)
            h = gn_swish(h, self.norm2, pair=True)
            res = x if self.in_channels == self.out_channels else ops.conv2d(x, packed_conv(self, self.nin_shortcut, split=3), hilo=True)
            return ops.conv2d(h, packed_conv(self, self.conv2, split=3), residual=res, gn_stats=fuse, hilo=True)
        if GN_PROLOGUE and not hl and self.in_channels <= 128 and getattr(x, "_gn_stats", None) is not None:
            coef = lambda t, n: ops.groupnorm_coeffs(t, n.weight.detach().float(), n.bias.detach().float(), n.eps)
            h = ops.conv2d(x, packed_conv(self, self.conv1), gn_stats=fuse, gn_prologue=(coef(x, self.norm1), True))
            if fuse and self.out_channels <= 128 and out is None:
                return ops.conv2d(h, packed_conv(self, self.conv2), gn_prologue=(coef(h, self.norm2), True),
                                  residual=x if self.in_channels == self.out_channels else ops.conv2d(x, packed_conv(self, self.nin_shortcut)),
                                  gn_stats=fuse)
        else:
            h = ops.conv2d(gn_swish(x, self.norm1), packed_conv(self, self.conv1), gn_stats=fuse)  # stats for norm2
        h = gn_swish(h, self.norm2)
        res = x if self.in_channels == self.out_channels else ops.conv2d(x, packed_conv(self, self.nin_shortcut), hilo=hl)
        # the block output feeds the next block's / attention's norm: its statistics ride along too
        return ops.conv2d(h, packed_conv(self, self.conv2), residual=res, out=out, out_off=out_off,
                          gn_stats=fuse and out is None, hilo=hl)

    def train_nhwc(self, x):
        x, xr = A.fork(x)                                         # x feeds the block body and the shortcut
        h = conv_t(gn_swish_t(x, self.norm1), self.conv1)
        h = gn_swish_t(h, self.norm2)
        res = xr if self.in_channels == self.out_channels else conv_t(xr, self.nin_shortcut)
        return conv_t(h, self.conv2, residual=res)

    def forward(self, x, temb=None):
        return to_nchw(self.forward_nhwc(to_nhwc(x)))


class AttnBlock(HipModule):
    def __init__(self, in_channels):
        super().__init__()
        # GLARE instantiates 512 only (the blockwise kernels of csrc/attn.hip); any other multiple of 32 -- the reference's class is
        # generic, encoder_decoder.py:140-192 -- runs the materialised form (_forward_general)
        assert in_channels % 32 == 0, "AttnBlock: channels must be a multiple of 32 (GroupNorm(32), 32-wide GEMM k-steps)"
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    def _qk(self):
        # softmax(q.k / sqrt(c)) == exp2-softmax with q pre-scaled by c^-0.5 * log2(e) (folded once here)
        s = float(self.in_channels) ** -0.5 * math.log2(math.e)
        w = torch.cat([self.q.weight * s, self.k.weight], 0)
        b = torch.cat([self.q.bias * s, self.k.bias], 0)
        return ops.PackedConv(w, b)

    def _v_proj(self):
        wp, wv = self.proj_out.weight[:, :, 0, 0].double(), self.v.weight[:, :, 0, 0].double()
        w = (wp @ wv).float()[:, :, None, None].contiguous()
        b = (wp @ self.v.bias.double() + self.proj_out.bias.double()).float()
        return ops.PackedConv(w, b)

    def _q_folded(self, split=0):
        # s_ij = (Wq h_i + bq).(Wk h_j + bk): the terms without j cancel in softmax_j, the rest is (Wk^T (Wq h_i + bq)) . h_j
        s = float(self.in_channels) ** -0.5 * math.log2(math.e)
        wq, wk = self.q.weight[:, :, 0, 0].double(), self.k.weight[:, :, 0, 0].double()
        w = (s * (wk.t() @ wq)).float()[:, :, None, None].contiguous()
        b = (s * (wk.t() @ self.q.bias.double())).float()
        return ops.PackedConv(w, b, split=split)

    def _out_folded(self, split=0):
        # Wp (sum_j P_ij (Wv h_j + bv)) + bp = (Wp Wv) (sum_j P_ij h_j) + Wp bv + bp   (softmax rows sum to 1)
        wp, wv = self.proj_out.weight[:, :, 0, 0].double(), self.v.weight[:, :, 0, 0].double()
        w = (wp @ wv).float()[:, :, None, None].contiguous()
        b = (wp @ self.v.bias.double() + self.proj_out.bias.double()).float()
        return ops.PackedConv(w, b, split=split)

    def _fold_mats(self):
        s = float(self.in_channels) ** -0.5 * math.log2(math.e)
        wq, wk = self.q.weight[:, :, 0, 0].double(), self.k.weight[:, :, 0, 0].double()
        wp, wv = self.proj_out.weight[:, :, 0, 0].double(), self.v.weight[:, :, 0, 0].double()
        return ((s * (wk.t() @ wq)).float().contiguous(), (s * (wk.t() @ self.q.bias.double())).float().contiguous(),
                (wp @ wv).float().contiguous(), (wp @ self.v.bias.double() + self.proj_out.bias.double()).float().contiguous(),
                self.norm.weight.detach().float().contiguous(), self.norm.bias.detach().float().contiguous())

    def _forward_general(self, x):
        """Any head size: h = GN(x); q, k, v = 1x1(h); w = softmax_j(q.k / sqrt(C)); out = x + proj_out(w v)
        (encoder_decoder.py:168-192) in the MATERIALISED form on the training step's kernels -- scores by `gemm_nt` (fp32 [N, N]), base-2
        row softmax, P.V by `gemm_nt` against the transposed values -- instead of the d = 512 blockwise kernel.  Not on the GLARE path
        (every AttnBlock there has 512 channels): it exists so that the operator surface equals the reference's, and it is what the
        reference-generated AttnBlock(64) vector of tests/golden/blocks.npz is checked on."""
        from .. import train_ops as T

        B, H, W, C = x.shape
        N = H * W
        if 6 * N * N > 8 << 30:     # the fp32 scores + their 16-bit softmax of ONE image: 6 N^2 bytes (N = 16 275, the path's size: 1.6 GB)
            raise NotImplementedError("AttnBlock(%d) at %d tokens: the general form materialises 6 N^2 = %.1f GB of scores per image; "
                                      "only the 512-channel block has the blockwise kernel" % (C, N, 6.0 * N * N / 2 ** 30))
        s = float(C) ** -0.5 * math.log2(math.e)
        hn = gn_swish(x, self.norm, swish=False)
        q = ops.conv2d(hn, self._packed("q_scaled", lambda: ops.PackedConv(self.q.weight * s, self.q.bias * s)))
        k = ops.conv2d(hn, packed_conv(self, self.k))
        v = ops.conv2d(hn, packed_conv(self, self.v))
        o = torch.empty(B, N, C, dtype=ops.act_dtype(), device=x.device)
        for b in range(B):                                    # per-sample attention: the N x N scores are per image
            S = T.gemm_nt(q[b].view(N, C), k[b].view(N, C))   # fp32 [N, N], base-2 logits
            P = T.softmax2_rows(S, N)                         # 16-bit [N, ldp], pad columns zero
            Vt = T.transpose(v[b].view(N, C), P.shape[1])     # [C, ldp], pad columns zero
            T.gemm_nt(P, Vt, out=o[b], out_dtype=ops.act_dtype())
        return ops.conv2d(o.view(B, H, W, C), packed_conv(self, self.proj_out), residual=x)

    def forward_nhwc(self, x, split=0):
        B, H, W, C = x.shape
        N = H * W
        if C != 512:
            assert not split, "the fp32-class form exists for the GLARE head size only"
            return self._forward_general(x)
        stats = getattr(x, "_gn_stats", None)
        if split:
            # fp32-class (FP32_CLASS): the norm is materialised as a hi / lo pair instead of folded into per-image filters, the two
            # folded 1x1 convs contract pairs against [w_hi | w_hi | w_lo], and the attention's fp32 accumulators leave as a pair.
            # Keys / values are the pair's hi half (tools/precision_sites.py: 16-bit keys / values cost 1.6e-5 of latent error, the
            # 16-bit attention output 2.8e-4, the two filters 1.3e-4 / 1.5e-4).
            assert split == 3 and is_hilo(x) and SHARED_KV_ATTENTION
            hn = gn_swish(x, self.norm, swish=False, pair=True)
            q = ops.conv2d(hn, self._packed("q_folded3", lambda: self._q_folded(split=3)))
            a = ops.attention_kv512(q, hn, N, pair=True)
            av = a.view(B, H, W, C)
            av._lo = a._lo.view(B, H, W, C)
            return ops.conv2d(av, self._packed("out_folded3", lambda: self._out_folded(split=3)), residual=x, gn_stats=GN_FUSED, hilo=True)
        if SHARED_KV_ATTENTION and GN_FOLDED_ATTENTION and stats is not None and 64 % B == 0:
            wq, bq, wo, bo, gamma, beta = self._packed("fold_mats", self._fold_mats)
            wq_b, bq_b, wo_b, bo_b = ops.attn_fold_groupnorm(stats, N, gamma, beta, self.norm.eps, wq, bq, wo, bo)
            q = ops.conv1x1_per_image(x, wq_b, bq_b)
            a = ops.attention_kv512(q, x, N)                                     # keys = values = the raw input
            return ops.conv1x1_per_image(a.view(B, H, W, C), wo_b, bo_b, residual=x, gn_stats=GN_FUSED, hilo=is_hilo(x))
        hn = gn_swish(x, self.norm, swish=False)
        if SHARED_KV_ATTENTION:
            q = ops.conv2d(hn, self._packed("q_folded", self._q_folded))
            a = ops.attention_kv512(q, hn, N)                                    # sum_j softmax_j(q'_i . h_j) h_j
            return ops.conv2d(a.view(B, H, W, C), self._packed("out_folded", self._out_folded), residual=x, gn_stats=GN_FUSED,
                              hilo=is_hilo(x))
        qk = ops.conv2d(hn, self._packed("qk", self._qk))                       # [B,H,W,1024]: q | k
        npad = (N + 63) // 64 * 64
        if FOLD_PROJ_INTO_V:
            # proj_out(softmax(S) V) = softmax(S) (Wp Wv h + Wp bv) + bp because every softmax row sums to 1: the output
            # projection is folded into the value projection (one 1x1 conv less per block), what remains of proj_out is "+ x"
            vt = ops.conv2d(hn, self._packed("v_proj", self._v_proj), out_mode=ops.OUT_PLANAR_BF16, plane_pitch=npad)
            o = ops.attention_d512(qk, qk[..., C:], vt, N, ldq=2 * C, ldk=2 * C)
            return ops.add_bf16(x, o.view(B, H, W, C), gn_stats=GN_FUSED)   # + the next norm's statistics
        vt = ops.conv2d(hn, packed_conv(self, self.v), out_mode=ops.OUT_PLANAR_BF16, plane_pitch=npad)  # V^T [B,512,npad]
        o = ops.attention_d512(qk, qk[..., C:], vt, N, ldq=2 * C, ldk=2 * C)     # [B,N,512]
        return ops.conv2d(o.view(B, H, W, C), packed_conv(self, self.proj_out), residual=x, gn_stats=GN_FUSED)

    def train_nhwc(self, x):
        B, H, W, C = x.shape
        if C != 512:    # the taped attention kernels (forward with log-sum-exp, fused / materialised backward) exist for the GLARE head size only
            raise NotImplementedError("AttnBlock.train_nhwc: the training kernels are built for 512 channels (got %d); "
                                      "forward_nhwc runs the general materialised form for other sizes" % C)
        s = float(self.in_channels) ** -0.5 * math.log2(math.e)   # the fold is a (differentiable) op on the filter
        x, xr = A.fork(x)
        hq, hk, hv = A.fork(gn_swish_t(x, self.norm, swish=False), 3)
        q = A.conv2d(hq, self.q.weight * s, self.q.bias * s)
        k, v = conv_t(hk, self.k), conv_t(hv, self.v)
        o = A.attention(q.view(B, H * W, C), k.view(B, H * W, C), v.view(B, H * W, C))
        return conv_t(o.view(B, H, W, C), self.proj_out, residual=xr)

    def forward(self, x):
        return to_nchw(self.forward_nhwc(to_nhwc(x)))


class _Level(nn.Module):
    pass


class Encoder(HipModule):
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4, 8), num_res_blocks=2, attn_resolutions=(16,), dropout=0.0,
                 resamp_with_conv=True, in_channels=3, resolution=256, z_channels=256, double_z=True):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            down = _Level()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res //= 2
            self.down.append(down)
        self.mid = _Level()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, 1, 1)

    def forward_nhwc(self, x_nchw):
        """x_nchw: fp32 NCHW image (read in place).  Returns (latent fp32 NHWC [B,h,w,zc], enc_feat list)."""
        x = x_nchw.float().contiguous()
        B, C, H, W = x.shape
        split = 0
        if getattr(self, "hilo_stream", False) and HILO_STREAM and ops.precision() == "fp16":
            # the residual stream as hi / lo pairs (22 mantissa bits): its rounding at every block output is the largest single
            # term of the latent error in fp16; everything that READS the stream (convs, attention) reads hi
            h = ops.conv2d_smallcin(x, (C * H * W, H * W, W, 1), (B, H, W), self.conv_in.weight, self.conv_in.bias, hilo=True)
            split = 3 if fp32_class_on() else 0      # ... and in the fp32-class form the convs contract hi / lo pairs (three K segments)
        else:
            h = ops.conv2d_smallcin(x, (C * H * W, H * W, W, 1), (B, H, W), self.conv_in.weight, self.conv_in.bias)
        kw = {"split": split} if split else {}
        feats = []
        for i_level in range(self.num_resolutions):
            lvl = self.down[i_level]
            for i_block in range(self.num_res_blocks):
                h = lvl.block[i_block].forward_nhwc(h, **kw)
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block].forward_nhwc(h, **kw)
            if i_level != self.num_resolutions - 1:
                feats.append(h)
                h = lvl.downsample.forward_nhwc(h, **kw)
        h = self.mid.block_2.forward_nhwc(self.mid.attn_1.forward_nhwc(self.mid.block_1.forward_nhwc(h, **kw), **kw), **kw)
        h = gn_swish(h, self.norm_out, pair=bool(split))
        z = ops.conv2d(h, packed_conv(self, self.conv_out, split=split), out_mode=ops.OUT_NHWC_F32)
        z._fp32_class = bool(split)          # ConEncoder1 hands cond_feat to the flow as a hi / lo pair then
        return z, feats

    def train_nhwc(self, x_nchw):
        h = A.conv2d_small(x_nchw.float().contiguous(), self.conv_in.weight, self.conv_in.bias, layout="nchw")
        feats = []
        for i_level in range(self.num_resolutions):
            lvl = self.down[i_level]
            for i_block in range(self.num_res_blocks):
                h = lvl.block[i_block].train_nhwc(h)
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block].train_nhwc(h)
            if i_level != self.num_resolutions - 1:
                feats.append(h)
                h = lvl.downsample.train_nhwc(h)
        h = self.mid.block_2.train_nhwc(self.mid.attn_1.train_nhwc(self.mid.block_1.train_nhwc(h)))
        h = gn_swish_t(h, self.norm_out)
        return conv_t(h, self.conv_out, out_f32=True), feats

    def forward(self, x, mid_feat=False):
        z, feats = self.forward_nhwc(x)
        z = to_nchw(z)
        return (z, [to_nchw(f) for f in feats]) if mid_feat else z


def build_decoder_trunk(mod, ch, ch_mult, num_res_blocks, attn_resolutions, resolution, z_channels, resamp_with_conv=True):
    """conv_in / mid / up of Decoder and MultiScaleDecoder2 (identical in the reference:
    encoder_decoder.py:445-513, deformableDecoder_arch.py:413-482)."""
    mod.num_resolutions, mod.num_res_blocks = len(ch_mult), num_res_blocks
    block_in = ch * ch_mult[mod.num_resolutions - 1]
    curr_res = resolution // 2 ** (mod.num_resolutions - 1)
    mod.z_shape = (1, z_channels, curr_res, curr_res)
    mod.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
    mod.mid = _Level()
    mod.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in)
    mod.mid.attn_1 = AttnBlock(block_in)
    mod.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in)
    ups = []
    for i_level in reversed(range(mod.num_resolutions)):
        block, attn = nn.ModuleList(), nn.ModuleList()
        block_out = ch * ch_mult[i_level]
        for _ in range(num_res_blocks + 1):
            block.append(ResnetBlock(in_channels=block_in, out_channels=block_out))
            block_in = block_out
            if curr_res in attn_resolutions:
                attn.append(AttnBlock(block_in))
        up = _Level()
        up.block, up.attn = block, attn
        if i_level != 0:
            up.upsample = Upsample(block_in, resamp_with_conv)
            curr_res *= 2
        ups.insert(0, up)
    mod.up = nn.ModuleList(ups)
    mod.norm_out = Normalize(block_in)
    return block_in


def decoder_stem(mod, z_nhwc_f32):
    """conv_in (3 -> 512, thin-input direct conv on the fp32 latent) + mid blocks."""
    B, H, W, C = z_nhwc_f32.shape
    h = ops.conv2d_smallcin(z_nhwc_f32, (H * W * C, 1, W * C, C), (B, H, W), mod.conv_in.weight, mod.conv_in.bias)
    return mod.mid.block_2.forward_nhwc(mod.mid.attn_1.forward_nhwc(mod.mid.block_1.forward_nhwc(h)))


class Decoder(HipModule):
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4, 8), num_res_blocks=2, attn_resolutions=(16,), dropout=0.0,
                 resamp_with_conv=True, in_channels=3, resolution=256, z_channels=256, give_pre_end=False):
        super().__init__()
        self.ch, self.temb_ch, self.resolution, self.in_channels = ch, 0, resolution, in_channels
        self.give_pre_end = give_pre_end
        block_in = build_decoder_trunk(self, ch, ch_mult, num_res_blocks, attn_resolutions, resolution, z_channels,
                                       resamp_with_conv)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)

    def forward_nhwc(self, z, want_image=True):
        """z: fp32 NHWC latent.  Returns (image fp32 NCHW or None, [feat@half, feat@full] bf16 NHWC).
        The reference always computes the RGB image and its caller drops it
        (VQLLFLOWDeformable_arch.py:246); want_image=False skips norm_out + conv_out."""
        h = decoder_stem(self, z)
        feats = []
        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                h = lvl.block[i_block].forward_nhwc(h)
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block].forward_nhwc(h)
            if i_level != 2:  # encoder_decoder.py:538-539
                feats.append(h)
            if i_level != 0:
                h = lvl.upsample.forward_nhwc(h)
        img = None
        if want_image and not self.give_pre_end:
            hh = gn_swish(h, self.norm_out)
            B, H, W, _ = hh.shape
            img = ops.conv2d(hh, packed_conv(self, self.conv_out), out_mode=ops.OUT_PLANAR_F32).view(B, -1, H, W)
        return img, feats

    def forward(self, z):
        img, feats = self.forward_nhwc(to_nhwc(z, bf16=False))
        return img, [to_nchw(f) for f in feats]
