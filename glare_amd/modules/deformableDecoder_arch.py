"""Adaptive-Feature-Transformation decoder on HIP kernels.

Mirrors code/models/modules/deformableDecoder_arch.py: DCNv2Pack (:132-152), WarpBlock (:279-290),
Mix (:579-590), MultiScaleDecoder2 (:413-576), including the parameters the reference builds but
never uses (scale / bias / enc, conv_out) so that checkpoints load unchanged."""
import torch
import torch.nn as nn

from .. import _lib
from .. import autograd as A
from .. import ops
from ._base import HipModule, packed_conv, to_nchw, to_nhwc
from .encoder_decoder import build_decoder_trunk, conv_t, decoder_stem, gn_swish, gn_swish_t, Normalize
from .ops.dcn import ModulatedDeformConvPack, modulated_deform_conv


import os

# The DCN contraction is the split fp32-class form everywhere by default (the reference contracts in fp32:
# deformableDecoder_arch.py:143,550-551, deform_conv_cuda.cpp:551-554).  GLARE_DCN_SINGLE_PASS=1 opts inference under the fp16
# precision into ONE half-precision MFMA per product (GLARE_MDCN_SINGLE_PASS: the arithmetic of every other convolution of the decoder;
# 1.1-1.5e-4 of max|out| against the split form, -0.9 / -0.6 ms per launch) -- an A/B switch, never the default (round 4).
DCN_SINGLE_PASS = os.environ.get("GLARE_DCN_SINGLE_PASS", "0") == "1"
# Round 6: inside the AFT decoder the warp's output x_w leaves the DCN kernel as a 16-bit tensor (rounded once from the fp32 accumulator;
# its only consumer, h + x_w * mean(h) / mean(x_w), rounds the sum to 16 bits anyway) and the two means of that rescale come from the
# producers' epilogues (Mix, DCN) instead of a pass over both tensors.  GLARE_DCN_OUT_16BIT=0: fp32 x_w (the sums stay fused);
# GLARE_RESCALE_FUSED=0: rounds 1-5's form (fp32 x_w, mean_rescale's own statistics pass).  The op-level API is unchanged (fp32 out).
DCN_OUT_16BIT = os.environ.get("GLARE_DCN_OUT_16BIT", "1") == "1"
RESCALE_FUSED = os.environ.get("GLARE_RESCALE_FUSED", "1") == "1"


class DCNv2Pack(ModulatedDeformConvPack, HipModule):
    """Offsets and masks come from a second feature map (deformableDecoder_arch.py:141-152)."""

    def forward(self, x, feat):  # NCHW fp32, the reference surface
        out = to_nchw(ops.conv2d(to_nhwc(feat), packed_conv(self, self.conv_offset), out_mode=ops.OUT_NHWC_F32))
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        return modulated_deform_conv(x.float(), offset, torch.sigmoid(mask), self.weight, self.bias, self.stride,
                                     self.padding, self.dilation, self.groups, self.deformable_groups)

    def forward_nhwc(self, x, feat, x_off=0, fused=False):
        """x: bf16 NHWC (the VQ-decoder feature), feat: bf16 NHWC.  conv_offset writes offsets and mask
        logits as fp32 planes; the DCN kernel applies the sigmoid while sampling.  -> fp32 NHWC.
        fused (round 6, the AFT decoder's call): -> (x_w, tile sums of its fp32 values, tile pixels) with x_w 16-bit when
        DCN_OUT_16BIT -- the DCN's epilogue feeds the mean rescale behind the warp; None where the lean kernel does not apply."""
        B, H, W, _ = feat.shape
        plane = (H * W + 63) // 64 * 64
        om = ops.conv2d(feat, packed_conv(self, self.conv_offset), out_mode=ops.OUT_PLANAR_F32, plane_pitch=plane)
        single = DCN_SINGLE_PASS and ops.precision() == "fp16" and x.dtype == torch.float16
        if fused and x.dtype == ops.act_dtype():
            pdf = (self._packed("dcn1", lambda: ops.PackedDcn(self.weight, self.bias, self.deformable_groups, single=True)) if single else
                   self._packed("dcn", lambda: ops.PackedDcn(self.weight, self.bias, self.deformable_groups, single=False)))
            try:
                return ops.mdcn_forward_nhwc_fused(x, om, pdf, x_off=x_off, C=self.in_channels, mask_is_logit=True, padding=self.padding,
                                                   out16=DCN_OUT_16BIT)
            except _lib.GlareError as e:          # outside the lean kernel's shapes: the plain call below, sums by mean_rescale's own pass
                if e.status != _lib.ERR_UNSUPPORTED:
                    raise
        if fused:
            return self._forward_plain(x, om, x_off, single), None, 0
        return self._forward_plain(x, om, x_off, single)

    def _forward_plain(self, x, om, x_off, single):
        if single:
            # the single-pass form lives in the fast kernel only (every tensor < 2 GB, < 2^31 pixels): shapes beyond it -- batch ~20 at
            # 400x600, a few 1080p images -- take the split form, whose call falls through to the general-extent kernel
            pd = self._packed("dcn1", lambda: ops.PackedDcn(self.weight, self.bias, self.deformable_groups, single=True))
            try:
                return ops.mdcn_forward_nhwc(x, om, pd, x_off=x_off, C=self.in_channels, mask_is_logit=True, padding=self.padding)
            except _lib.GlareError as e:
                if e.status != _lib.ERR_UNSUPPORTED:      # only "valid request outside the fast kernel" falls back to the split form
                    raise
        pd = self._packed("dcn", lambda: ops.PackedDcn(self.weight, self.bias, self.deformable_groups, single=False))
        return ops.mdcn_forward_nhwc(x, om, pd, x_off=x_off, C=self.in_channels, mask_is_logit=True, padding=self.padding)

    def train_nhwc(self, x, feat):
        om = conv_t(feat, self.conv_offset, out_f32=True)
        return A.dcn(x, om, self.weight, self.bias, self.deformable_groups, self.padding)


class WarpBlock(HipModule):
    def __init__(self, in_channel):
        super().__init__()
        self.offset = nn.Conv2d(in_channel * 2, in_channel, 3, stride=1, padding=1)
        self.dcn = DCNv2Pack(in_channel, in_channel, 3, padding=1, deformable_groups=4)

    def forward_nhwc(self, x_vq, x_residual, fused=False):
        r = ops.conv2d(x_vq, packed_conv(self, self.offset), x2=x_residual)  # torch.cat fused: two conv sources
        return self.dcn.forward_nhwc(x_vq, r, fused=fused)

    def train_nhwc(self, x_vq, x_residual):
        return self.dcn.train_nhwc(x_vq, conv_t(x_vq, self.offset, x2=x_residual))

    def forward(self, x_vq, x_residual):
        return to_nchw(self.forward_nhwc(to_nhwc(x_vq), to_nhwc(x_residual)))


class Mix(HipModule):
    def __init__(self, m=-0.80):
        super().__init__()
        self.w = nn.Parameter(torch.FloatTensor([m]))
        self.mix_block = nn.Sigmoid()

    def forward_nhwc(self, fea1, fea2, with_sums=False):
        # the scalar is read back once (cached with the packed weights): no host sync inside the launch sequence
        w = self._packed("w", lambda: float(self.w.detach()))
        if with_sums:          # + the per-block sums of the result: mean(h) of the rescale behind the warp (ops.mix_with_sums)
            return ops.mix_with_sums(fea1, fea2, w)
        return ops.mix(fea1, fea2, w)

    def train_nhwc(self, fea1, fea2):
        return A.mix(fea1, fea2, self.w)

    def forward(self, fea1, fea2):
        return to_nchw(self.forward_nhwc(to_nhwc(fea1), to_nhwc(fea2)))


class ResBlock(nn.Module):  # deformableDecoder_arch.py:157-180, parameters only (never called on the path)
    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.conv_out = nn.Conv2d(in_channels, out_channels, 1, 1, 0)


def _gate(c):  # :490-508, parameters only
    return nn.Sequential(nn.Conv2d(c, c, 3, padding=1), nn.LeakyReLU(0.2, True), nn.Conv2d(c, c, 3, padding=1), nn.Sigmoid())


class MultiScaleDecoder2(HipModule):
    def __init__(self, ch=64, out_ch=3, ch_mult=(1, 2, 4), num_res_blocks=2, attn_resolutions=(64,), dropout=0.0,
                 resamp_with_conv=True, in_channels=3, resolution=256, z_channels=3):
        super().__init__()
        self.ch, self.temb_ch, self.resolution, self.in_channels = ch, 0, resolution, in_channels
        block_in = build_decoder_trunk(self, ch, ch_mult, num_res_blocks, attn_resolutions, resolution, z_channels,
                                       resamp_with_conv)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)
        self.warp = nn.ModuleList([WarpBlock(ch * 2), WarpBlock(ch)])
        self.residual_conv = nn.Conv2d(ch, out_ch, 3, 1, 1)
        self.scale = nn.ModuleList([_gate(256), _gate(128)])
        self.bias = nn.ModuleList([_gate(256), _gate(128)])
        self.enc = nn.ModuleList([ResBlock(512, 256), ResBlock(256, 128)])
        self.mix = nn.ModuleList([Mix(m=-1.0), Mix(m=-0.6)])
        # inference: per-sample means so that a batched run equals the reference's B=1 runs
        # (SURVEY.md section 8e); True reproduces the reference's whole-tensor means (:567)
        self.whole_batch_mean = False

    def forward_nhwc(self, z, code_feats, enc_feats):
        """z: fp32 NHWC latent; code_feats: [bf16 NHWC @half (256), @full (128)] from the VQ decoder;
        enc_feats: [bf16 NHWC @full (128), @half (256)] from the conditional encoder -> fp32 NCHW image."""
        h = decoder_stem(self, z)
        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                h = lvl.block[i_block].forward_nhwc(h)
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block].forward_nhwc(h)
            if i_level != 2:  # :546-567
                x_code = code_feats[1 - i_level]
                e = enc_feats[i_level]
                if RESCALE_FUSED and e.is_contiguous() and h.is_contiguous() and e.shape == h.shape:
                    h, h_sums = self.mix[1 - i_level].forward_nhwc(e, h, with_sums=True)
                    x_w, x_sums, tile = self.warp[1 - i_level].forward_nhwc(x_code, h, fused=True)
                    if x_sums is not None:
                        h = ops.mean_rescale_fused(h, x_w, h_sums, x_sums, tile, whole_batch=self.whole_batch_mean)
                    else:
                        h = ops.mean_rescale(h, x_w, whole_batch=self.whole_batch_mean)
                else:
                    h = self.mix[1 - i_level].forward_nhwc(e, h)
                    x_w = self.warp[1 - i_level].forward_nhwc(x_code, h)
                    h = ops.mean_rescale(h, x_w, whole_batch=self.whole_batch_mean)
            if i_level != 0:
                h = lvl.upsample.forward_nhwc(h)
        h = gn_swish(h, self.norm_out)
        B, H, W, _ = h.shape
        return ops.conv2d(h, packed_conv(self, self.residual_conv), out_mode=ops.OUT_PLANAR_F32).view(B, -1, H, W)

    def train_nhwc(self, z, code_feats, enc_feats, whole_batch_mean=True):
        """forward_nhwc with a tape (stage-3 training, VQLLFLOWDeformable_arch.py:249): only this module's parameters
        are trained; z / code_feats / enc_feats come from frozen networks.  The mean of `h + x_w * mean(h)/mean(x_w)` is
        taken over the per-rank batch, as each nn.DataParallel replica does.  -> fp32 NHWC image."""
        B, H, W, C = z.shape
        h = A.conv2d_small(z, self.conv_in.weight, self.conv_in.bias, layout="nhwc")
        h = self.mid.block_2.train_nhwc(self.mid.attn_1.train_nhwc(self.mid.block_1.train_nhwc(h)))
        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                h = lvl.block[i_block].train_nhwc(h)
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block].train_nhwc(h)
            if i_level != 2:
                h, hw = A.fork(self.mix[1 - i_level].train_nhwc(enc_feats[i_level], h))   # read by the warp and the rescale
                x_w = self.warp[1 - i_level].train_nhwc(code_feats[1 - i_level], hw)
                h = A.mean_rescale(h, x_w, whole_batch_mean)
            if i_level != 0:
                h = lvl.upsample.train_nhwc(h)
        return conv_t(gn_swish_t(h, self.norm_out), self.residual_conv, out_f32=True)

    def forward(self, z, code_decoder_output, enc_feat):
        return self.forward_nhwc(to_nhwc(z, bf16=False), [to_nhwc(f) for f in code_decoder_output],
                                 [to_nhwc(f) for f in enc_feat])
