"""Inference / stage-3 graph: conditional encoder -> flow (reverse) -> codebook retrieval -> VQGAN
decoder -> AFT decoder.  Mirrors VQLLFLOWDeformable (code/models/modules/VQLLFLOWDeformable_arch.py:
ctor :19-52, forward :103-128, reverse_flow :222-250) with the same submodule names (RRDB,
flowUpsamplerNet, deformable_decoder) and therefore the same 824 state-dict keys."""
import math

import torch
import torch.nn as nn

from .. import autograd as A
from .. import ops
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

    def reverse_flow_nhwc(self, net_vq, lr, precision=None):
        """lr: fp32 NCHW log-domain image batch.  Every intermediate stays NHWC on device.
        precision: the 16-bit format of activations and filters -- "fp16" (IEEE half, the reference's own autocast dtype:
        8x less rounding per stored tensor than bf16 at the same MFMA rate; what the end-to-end tolerance needs, DESIGN.md
        section 4) or "bf16" (fp32 range; +2.7 % images/s); None = the enclosing `ops.use_precision`, else fp16."""
        with ops.use_precision(ops.inference_precision(precision)):
            enc = self.RRDB.forward_nhwc(lr)
            latent = self.flowUpsamplerNet.decode_nhwc(enc["color_map"], enc["cond_feat"])
            idx, _, code_feats = net_vq.decode_nhwc(latent, want_image=False)
            out = self.deformable_decoder.forward_nhwc(latent, code_feats, enc["mid_feat"])
        return {"out": out, "latent": latent, "indices": idx, "enc": enc, "code_feats": code_feats}

    def reverse_flow_train_nhwc(self, net_vq, lr, whole_batch_mean=True):
        """reverse_flow with a tape where the reference has one (VQLLFLOWDeformable_arch.py:231-249): the conditional
        encoder, the flow and the VQGAN decoder run under no_grad there too (:231, :245), only `deformable_decoder` is
        differentiated.  -> (fp32 NHWC image with a tape, fp32 NHWC latent)."""
        with torch.no_grad():
            enc = self.RRDB.forward_nhwc(lr)
            latent = self.flowUpsamplerNet.decode_nhwc(enc["color_map"], enc["cond_feat"])
            _, _, code_feats = net_vq.decode_nhwc(latent, want_image=False)
        out = self.deformable_decoder.train_nhwc(latent, code_feats, enc["mid_feat"], whole_batch_mean=whole_batch_mean)
        return out, latent

    def forward(self, net_vq=None, gt=None, lr=None, z=None, eps_std=None, reverse=True, epses=None,
                reverse_with_grad=True, lr_enc=None, add_gt_noise=False, step=None, y_label=None,
                align_condition_feature=False, get_color_map=False):
        """The reference's entry point (VQLLFLOWDeformable_arch.py:103-128).  reverse=True, reverse_with_grad=True under
        autograd (VQLLFLOWD_model.py:205-208, stage 3) returns an image that carries the tape of `deformable_decoder`, so
        `total_loss.backward()` of the reference's optimize_parameters works on it; otherwise the fused no-grad graph runs.
        The normal direction of THIS class is never called by the reference (stage 2 uses LLFlowVQGAN2): it delegates to the
        same flow kernels for completeness."""
        if not reverse:
            assert gt is not None
            enc = self.RRDB.forward_nhwc(lr)
            zz, logdet, logp = self.flowUpsamplerNet.encode_nhwc(A.to_nhwc_f32(gt.detach()), enc["cond_feat"], mean=enc["color_map"])
            pixels = gt.shape[2] * gt.shape[3]
            nll = -(logdet + logp) / (math.log(2.0) * pixels)
            return to_nchw(zz), nll.float(), logdet.float()
        assert lr.shape[1] == 3
        if reverse_with_grad and torch.is_grad_enabled() and any(p.requires_grad for p in self.deformable_decoder.parameters()):
            out, latent = self.reverse_flow_train_nhwc(net_vq, lr)
            return A.nhwc_to_nchw(out), to_nchw(latent)
        with torch.no_grad():
            r = self.reverse_flow_nhwc(net_vq, lr)
        return r["out"], to_nchw(r["latent"])
