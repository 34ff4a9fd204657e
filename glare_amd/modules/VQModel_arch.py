"""Frozen VQGAN prior `net_hq` (mirrors code/models/modules/VQModel_arch.py:14-91, encode/decode)."""
import torch.nn as nn

from .. import ops
from ._base import HipModule, to_nchw, to_nhwc
from .encoder_decoder import Decoder, Encoder
from .quantize import VectorQuantizer2


class VQModel(HipModule):
    def __init__(self, resolution=256, n_embed=8192, embed_dim=3, ckpt_path=None, double_z=False, z_channels=3,
                 in_channels=3, out_ch=3, ch=128, ch_mult=(1, 2, 4), num_res_blocks=2, attn_resolutions=(64,), dropout=0.0,
                 **unused):
        super().__init__()
        self.encoder = Encoder(ch, out_ch, ch_mult=ch_mult, num_res_blocks=num_res_blocks,
                               attn_resolutions=attn_resolutions, in_channels=in_channels, resolution=resolution,
                               z_channels=z_channels, double_z=double_z)
        self.decoder = Decoder(ch, out_ch, ch_mult=ch_mult, num_res_blocks=num_res_blocks,
                               attn_resolutions=attn_resolutions, in_channels=in_channels, resolution=resolution,
                               z_channels=z_channels)
        self.quantize = VectorQuantizer2(n_embed, embed_dim, beta=0.25)
        self.quant_conv = nn.Conv2d(z_channels, embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, z_channels, 1)
        self.conv_semantic = nn.Sequential(nn.Conv2d(3, 256, 1, 1, 0), nn.ReLU())  # state-dict surface only

    def _conv1x1_latent(self, conv, z):
        B, H, W, C = z.shape
        return ops.conv2d_smallcin(z, (H * W * C, 1, W * C, C), (B, H, W), conv.weight, conv.bias, out_f32=True)

    def encode_nhwc(self, x_nchw):
        z, _ = self.encoder.forward_nhwc(x_nchw)
        return self._conv1x1_latent(self.quant_conv, z)

    def encode(self, x):  # VQModel_arch.py:74-79
        return to_nchw(self.encode_nhwc(x)), None

    def decode_nhwc(self, z, want_image=False):
        """z: fp32 NHWC latent [B,h,w,3] -> (indices, image or None, [feat@half, feat@full])."""
        B, H, W, C = z.shape
        idx, zq = self.quantize.quantize_tokens(z.view(-1, C))
        quant2 = self._conv1x1_latent(self.post_quant_conv, zq.view(B, H, W, C))
        img, feats = self.decoder.forward_nhwc(quant2, want_image=want_image)
        return idx, img, feats

    def decode(self, h, vgg_feat=None):  # VQModel_arch.py:81-91
        assert vgg_feat is None
        quant, emb_loss, info = self.quantize(h)
        quant2 = self._conv1x1_latent(self.post_quant_conv, to_nhwc(quant, bf16=False))
        dec, feats = self.decoder.forward_nhwc(quant2, want_image=True)
        self.last_indices = info[2]
        return dec, emb_loss, [to_nchw(f) for f in feats]
