"""Stage-2 graph: conditional encoder -> normal flow -> negative log-likelihood.  Mirrors LLFlowVQGAN2
(code/models/modules/LLFlowVQGAN_arch.py:17-106; the `...2_arch.py` twin fails to import upstream, SURVEY.md section 2
row 16).  The Gaussian's mean is the conditional encoder's color_map, or -- with probability `train_gt_ratio`
(opt['train_gt_ratio']: 0 in confs/LOL.yml:12, 0.2 in confs/train_stage2_LOL.yml:14), one host-side `random.random()` draw
per forward as in LLFlowVQGAN_arch.py:95 -- the ground-truth latent itself (then no gradient reaches `color_conv`).
`train_nll` is the objective with a tape (HIP backward, row a12); `forward(reverse=False)` is the reference-shaped entry
point and returns a differentiable `nll` whenever autograd is recording, so LLFlowModel.optimize_parameters
(LLFlow_model.py:215-241) runs on it unmodified."""
import math
import random

import torch

from ._base import HipModule, to_nchw, to_nhwc
from .ConditionEncoder import ConEncoder1
from .FlowUpsamplerNet import FlowUpsamplerNet


class LLFlowVQGAN2(HipModule):
    def __init__(self, in_nc=3, out_nc=3, nf=32, nb=4, gc=32, scale=4, latent_size=64, latent_channel=512, K=None, opt=None,
                 step=None):
        super().__init__()
        self.opt = opt
        ratio = opt.get("train_gt_ratio", 0.0) if hasattr(opt, "get") else 0.0
        self.train_gt_ratio = float(ratio or 0.0)
        self.RRDB = ConEncoder1(opt=opt)
        self.flowUpsamplerNet = FlowUpsamplerNet((80, 80, 3), 64, K or 12, flow_coupling="CondAffineSeparatedAndCond", opt=opt)

    def _mean_is_gt(self, mean_is_gt=None):
        """LLFlowVQGAN_arch.py:95: `mean = color_map if random.random() > train_gt_ratio else gt` -- exactly one draw from
        Python's `random` per forward, also when the ratio is 0 (the reference draws unconditionally); a forced branch draws nothing."""
        if mean_is_gt is not None:
            return bool(mean_is_gt)
        return not random.random() > self.train_gt_ratio

    def normal_flow_nhwc(self, gt_latent, lr, mean_is_gt=None):
        """gt_latent: fp32 NHWC [B,h,w,3] (net_hq.encode of the ground truth); lr: fp32 NCHW image batch."""
        enc = self.RRDB.forward_nhwc(lr)
        mean = gt_latent if self._mean_is_gt(mean_is_gt) else enc["color_map"]
        z, logdet, logp = self.flowUpsamplerNet.encode_nhwc(gt_latent, enc["cond_feat"], mean=mean)
        pixels = gt_latent.shape[1] * gt_latent.shape[2]
        nll = -(logdet + logp) / (math.log(2.0) * pixels)  # LLFlowVQGAN_arch.py:99-101
        return z, nll.float(), logdet.float()

    def train_nll(self, gt_latent, lr, mean_is_gt=None, want_z=False):
        """Per-sample NLL (float64 [B]) with a tape through the conditional encoder and the flow: what
        LLFlowModel.optimize_parameters differentiates (LLFlow_model.py:215-236).  mean_is_gt: None = draw as the reference
        does (train_gt_ratio), True / False = forced."""
        if self.flowUpsamplerNet.needs_actnorm_init():          # first training step of a fresh flow (FlowActNorms.py:82-83):
            with torch.no_grad():                                # the ActNorms take their statistics from this batch, untaped
                enc0 = self.RRDB.forward_nhwc(lr)
                self.flowUpsamplerNet.initialize_actnorms_nhwc(gt_latent.detach(), enc0["cond_feat"])
        flow_params = self.flowUpsamplerNet._train_params()     # first: see train_nll_terms
        enc = self.RRDB.train_nhwc(lr)
        mean = gt_latent.detach() if self._mean_is_gt(mean_is_gt) else enc["color_map"]
        logdet, logp, z = self.flowUpsamplerNet.train_nll_terms(gt_latent, enc["cond_feat"], mean, params=flow_params, want_z=True)
        pixels = gt_latent.shape[1] * gt_latent.shape[2]
        nll = -(logdet + logp) / (math.log(2.0) * pixels)
        return (z, nll, logdet) if want_z else nll

    def forward(self, gt=None, lr=None, z=None, eps_std=None, reverse=False, epses=None, reverse_with_grad=False, lr_enc=None,
                add_gt_noise=False, step=None, y_label=None, align_condition_feature=False, get_color_map=False):
        if reverse:
            enc = self.RRDB.forward_nhwc(lr)
            x = self.flowUpsamplerNet.decode_nhwc(enc["color_map"], enc["cond_feat"])
            return to_nchw(x), torch.zeros(lr.shape[0], device=lr.device)
        assert not add_gt_noise, "dequantisation noise is off in every shipped config (LLFlowVQGAN_arch.py:73-79)"
        gt_latent = to_nhwc(gt.detach(), bf16=False)        # the caller passes encoder_gt.detach() (LLFlow_model.py:215)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            zz, nll, logdet = self.train_nll(gt_latent, lr, want_z=True)       # taped: nll.mean().backward() works
            return to_nchw(zz), nll, logdet.detach().float()
        zz, nll, logdet = self.normal_flow_nhwc(gt_latent, lr)
        return to_nchw(zz), nll, logdet
