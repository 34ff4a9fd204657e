"""Inference / stage-3 graph: conditional encoder -> flow (reverse) -> codebook retrieval -> VQGAN
decoder -> AFT decoder.  Mirrors VQLLFLOWDeformable (code/models/modules/VQLLFLOWDeformable_arch.py:
ctor :19-52, forward :103-128, reverse_flow :222-250) with the same submodule names (RRDB,
flowUpsamplerNet, deformable_decoder) and therefore the same 824 state-dict keys."""
import torch
import torch.nn as nn

from ._base import HipModule, to_nchw
from .ConditionEncoder import ConEncoder1
from .deformableDecoder_arch import MultiScaleDecoder2
from .FlowUpsamplerNet import FlowUpsamplerNet


class VQLLFLOWDeformable(HipModule):
    def __init__(self, in_nc=3, out_nc=3, nf=32, nb=4, gc=32, scale=4, latent_size=64, latent_channel=512, K=None,
                 opt=None, step=None, fix_modules=("RRDB", "flowUpsamplerNet")):
        super().__init__()
        self.opt = opt
        self.RRDB = ConEncoder1(opt=opt)
        self.deformable_decoder = MultiScaleDecoder2(ch=128, out_ch=3, ch_mult=(1, 2, 4), num_res_blocks=2,
                                                     attn_resolutions=[64], resolution=256, z_channels=3)
        self.flowUpsamplerNet = FlowUpsamplerNet((80, 80, 3), 64, K or 12, flow_coupling="CondAffineSeparatedAndCond", opt=opt)
        if fix_modules is not None:
            for name in fix_modules:
                for p in getattr(self, name).parameters():
                    p.requires_grad = False

    def reverse_flow_nhwc(self, net_vq, lr):
        """lr: fp32 NCHW log-domain image batch.  Every intermediate stays NHWC on device."""
        enc = self.RRDB.forward_nhwc(lr)
        latent = self.flowUpsamplerNet.decode_nhwc(enc["color_map"], enc["cond_feat"])
        idx, _, code_feats = net_vq.decode_nhwc(latent, want_image=False)
        out = self.deformable_decoder.forward_nhwc(latent, code_feats, enc["mid_feat"])
        return {"out": out, "latent": latent, "indices": idx, "enc": enc, "code_feats": code_feats}

    def forward(self, net_vq=None, gt=None, lr=None, z=None, eps_std=None, reverse=True, epses=None,
                reverse_with_grad=True, lr_enc=None, add_gt_noise=False, step=None, y_label=None,
                align_condition_feature=False, get_color_map=False):
        if not reverse:
            raise NotImplementedError("normal flow is the stage-2 graph (LLFlowVQGAN2); not on HIP yet")
        assert lr.shape[1] == 3
        with torch.no_grad():
            r = self.reverse_flow_nhwc(net_vq, lr)
        return r["out"], to_nchw(r["latent"])
