"""Stage-2 graph, forward (evaluation of the training objective): conditional encoder -> normal flow ->
negative log-likelihood.  Mirrors LLFlowVQGAN2 (code/models/modules/LLFlowVQGAN_arch.py:17-106; the
`...2_arch.py` twin fails to import upstream) with `train_gt_ratio: 0`
(confs/LOL.yml:12): mean = color_map.  `train_nll` is the same objective with a tape (HIP backward, row a12)."""
import math

import torch

from ._base import HipModule, to_nchw, to_nhwc
from .ConditionEncoder import ConEncoder1
from .FlowUpsamplerNet import FlowUpsamplerNet


class LLFlowVQGAN2(HipModule):
    def __init__(self, in_nc=3, out_nc=3, nf=32, nb=4, gc=32, scale=4, latent_size=64, latent_channel=512, K=None, opt=None,
                 step=None):
        super().__init__()
        self.opt = opt
        self.RRDB = ConEncoder1(opt=opt)
        self.flowUpsamplerNet = FlowUpsamplerNet((80, 80, 3), 64, K or 12, flow_coupling="CondAffineSeparatedAndCond", opt=opt)

    def normal_flow_nhwc(self, gt_latent, lr):
        """gt_latent: fp32 NHWC [B,h,w,3] (net_hq.encode of the ground truth); lr: fp32 NCHW image batch."""
        enc = self.RRDB.forward_nhwc(lr)
        z, logdet, logp = self.flowUpsamplerNet.encode_nhwc(gt_latent, enc["cond_feat"], mean=enc["color_map"])
        pixels = gt_latent.shape[1] * gt_latent.shape[2]
        nll = -(logdet + logp) / (math.log(2.0) * pixels)  # LLFlowVQGAN_arch.py:99-101
        return z, nll.float(), logdet.float()

    def train_nll(self, gt_latent, lr):
        """Per-sample NLL (float64 [B]) with a tape through the conditional encoder and the flow: what
        LLFlowModel.optimize_parameters differentiates (LLFlow_model.py:215-236)."""
        flow_params = self.flowUpsamplerNet._train_params()     # first: see train_nll_terms
        enc = self.RRDB.train_nhwc(lr)
        logdet, logp = self.flowUpsamplerNet.train_nll_terms(gt_latent, enc["cond_feat"], enc["color_map"], params=flow_params)
        pixels = gt_latent.shape[1] * gt_latent.shape[2]
        return -(logdet + logp) / (math.log(2.0) * pixels)

    def forward(self, gt=None, lr=None, z=None, eps_std=None, reverse=False, epses=None, reverse_with_grad=False, lr_enc=None,
                add_gt_noise=False, step=None, y_label=None, align_condition_feature=False, get_color_map=False):
        if reverse:
            enc = self.RRDB.forward_nhwc(lr)
            x = self.flowUpsamplerNet.decode_nhwc(enc["color_map"], enc["cond_feat"])
            return to_nchw(x), torch.zeros(lr.shape[0], device=lr.device)
        assert not add_gt_noise, "dequantisation noise is off in every shipped config (LLFlowVQGAN_arch.py:73-79)"
        zz, nll, logdet = self.normal_flow_nhwc(to_nhwc(gt, bf16=False), lr)
        return to_nchw(zz), nll, logdet
