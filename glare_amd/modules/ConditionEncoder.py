"""Conditional encoder (mirrors code/models/modules/ConditionEncoder.py:14-55)."""
import torch.nn as nn

from .. import autograd as A
from .. import ops
from ._base import HipModule, to_nchw
from .encoder_decoder import Encoder


class ConEncoder1(HipModule):
    def __init__(self, resolution=256, double_z=False, z_channels=3, in_channels=3, out_ch=3, ch=128, ch_mult=(1, 2, 4),
                 num_res_blocks=2, attn_resolutions=(64,), dropout=0.0, opt=None, **unused):
        super().__init__()
        self.opt = opt
        self.encoder = Encoder(ch, out_ch, ch_mult=ch_mult, num_res_blocks=num_res_blocks,
                               attn_resolutions=attn_resolutions, in_channels=in_channels, resolution=resolution,
                               z_channels=z_channels, double_z=double_z)
        self.encoder.hilo_stream = True     # inference in fp16: the residual stream as hi / lo pairs (encoder_decoder.is_hilo)
        self.color_conv = nn.Conv2d(3, 3, 3, 1, 1)
        self.cond_conv = nn.Sequential(nn.Conv2d(3, 64, 3, 1, 1), nn.Sigmoid())

    def forward_nhwc(self, x_nchw):
        """-> dict(cond_feat bf16 NHWC [B,h,w,64], color_map fp32 NHWC [B,h,w,3], mid_feat [bf16 NHWC])."""
        enc, feats = self.encoder.forward_nhwc(x_nchw)
        B, H, W, C = enc.shape
        st = (H * W * C, 1, W * C, C)
        # fp32-class mode (encoder_decoder.FP32_CLASS): cond_feat as a hi / lo pair -- it is an MFMA operand of all 48 coupling nets
        cond = ops.conv2d_smallcin(enc, st, (B, H, W), self.cond_conv[0].weight, self.cond_conv[0].bias, act="sigmoid",
                                   hilo=getattr(enc, "_fp32_class", False))
        color = ops.conv2d_smallcin(enc, st, (B, H, W), self.color_conv.weight, self.color_conv.bias, out_f32=True)
        return {"cond_feat": cond, "color_map": color, "mid_feat": feats}

    def train_nhwc(self, x_nchw):
        """forward_nhwc with a tape (stage-2 training, LLFlowVQGAN_arch.py:66)."""
        enc, feats = self.encoder.train_nhwc(x_nchw)
        cond = A.conv2d_small(enc, self.cond_conv[0].weight, self.cond_conv[0].bias, layout="nhwc", act="sigmoid")
        color = A.conv2d_small(enc, self.color_conv.weight, self.color_conv.bias, layout="nhwc", out_f32=True)
        return {"cond_feat": cond, "color_map": color, "mid_feat": feats}

    def forward(self, x, mid_feat=False):
        r = self.forward_nhwc(x)
        out = {"cond_feat": to_nchw(r["cond_feat"]), "color_map": to_nchw(r["color_map"])}
        if mid_feat:
            out["mid_feat"] = [to_nchw(f) for f in r["mid_feat"]]
        return out
