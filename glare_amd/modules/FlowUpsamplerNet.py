"""Conditional normalizing flow on HIP kernels.

Mirrors FlowUpsamplerNet (code/models/modules/FlowUpsamplerNet.py:17-326), FlowStep (FlowStep.py:16-127)
and CondAffineSeparatedAndCond (FlowAffineCouplingsAblation.py:10-151) at the configuration every
shipped yml uses (confs/LOL.yml:69-83: L=2, K=12, additionalFlowNoAffine=2, no squeeze, no split,
scale 1 -> conditional feature 'cond_feat').  State-dict keys are the reference's.

MI355X design of the reverse (sampling) pass, decode():
  * everything that does not depend on the latent z is hoisted out of the 24-step sequential loop and
    batched: the conditional half of every fAffine first conv (64 -> 24*64) and the whole fFeatures
    nets run as large MFMA convolutions over the conditional feature once;
  * ActNorm and the Conv2dZeros scale are folded into conv weights/biases on the host;
  * invconv^-1 (fp64 inverse, as Permutations.py:38), actnorm^-1 and the coupling-free steps that follow
    are pre-composed in fp64 into one 3x3 matrix + offset per coupling step (the reference recomputes
    a fp64 inverse and an slogdet per step per call);
  * per coupling step: flow_h1 -> 1x1 MFMA conv -> 3x3 MFMA conv -> flow_tail; z stays fp32.
"""
import os

import torch
import torch.nn as nn

from .. import _lib, ops
from .. import train_ops as T
from ._base import HipModule, to_nchw, to_nhwc
from .flow import ActNorm2d, Conv2d, Conv2dZeros, InvertibleConv1x1


# Round 6: a coupling step of the reverse pass is ONE launch (csrc/flow_fused.hip; the step's two MFMA convs, flow_h1 and flow_tail
# fused per pixel tile, fp32-class arithmetic in every precision).  GLARE_FLOW_FUSED=0 keeps rounds 1-5's four launches per step.
FUSED_STEP = os.environ.get("GLARE_FLOW_FUSED", "1") == "1"


def _opt_get(opt, keys, default=None):
    cur = opt
    for k in keys:
        if cur is None:
            return default
        cur = cur.get(k, None) if hasattr(cur, "get") else None
    return default if cur is None else cur


class CondAffineSeparatedAndCond(nn.Module):
    def __init__(self, in_channels, opt=None):
        super().__init__()
        self.need_features = True
        self.in_channels = in_channels
        self.in_channels_rrdb = _opt_get(opt, ["network_G", "flow", "conditionInFeaDim"], 64)
        self.hidden_channels = _opt_get(opt, ["network_G", "flow", "CondAffineSeparatedAndCond", "hidden_channels"], 64)
        self.affine_eps = _opt_get(opt, ["network_G", "flow", "CondAffineSeparatedAndCond", "eps"], 0.0001)
        self.channels_for_nn = in_channels // 2
        self.channels_for_co = in_channels - self.channels_for_nn
        self.fAffine = self.F(self.channels_for_nn + self.in_channels_rrdb, self.channels_for_co * 2)
        self.fFeatures = self.F(self.in_channels_rrdb, in_channels * 2)

    def F(self, in_channels, out_channels):
        h = self.hidden_channels
        return nn.Sequential(Conv2d(in_channels, h), nn.ReLU(inplace=False), Conv2d(h, h, kernel_size=[1, 1]),
                             nn.ReLU(inplace=False), Conv2dZeros(h, out_channels))


class FlowStep(nn.Module):
    def __init__(self, in_channels, hidden_channels=64, actnorm_scale=1.0, flow_permutation="invconv",
                 flow_coupling="CondAffineSeparatedAndCond", LU_decomposed=False, opt=None, **unused):
        super().__init__()
        assert flow_permutation == "invconv"
        self.flow_coupling = flow_coupling
        self.actnorm = ActNorm2d(in_channels, actnorm_scale)
        self.invconv = InvertibleConv1x1(in_channels, LU_decomposed=LU_decomposed)
        if flow_coupling == "CondAffineSeparatedAndCond":
            self.affine = CondAffineSeparatedAndCond(in_channels=in_channels, opt=opt)
        elif flow_coupling != "noCoupling":
            raise RuntimeError("coupling not Found:", flow_coupling)

    def forward_affine_fp64(self):
        """z -> A z + c of (actnorm, invconv) in fp64 (FlowStep.py:83-88) and the per-pixel log-determinant."""
        w = self.invconv.weight.detach().double().cpu()
        logs = self.actnorm.logs.detach().double().cpu().reshape(-1)
        bias = self.actnorm.bias.detach().double().cpu().reshape(-1)
        A = w * torch.exp(logs).view(1, -1)            # W diag(e^logs)
        c = A @ bias                                     # W ((z + b) e^logs) = A z + A b
        return A, c, float(logs.sum() + torch.slogdet(w)[1])

    def reverse_affine_fp64(self):
        """z -> A z + c of (invconv reverse, actnorm reverse) in fp64 (FlowStep.py:112-117)."""
        winv = torch.inverse(self.invconv.weight.detach().double().cpu())
        d = torch.exp(-self.actnorm.logs.detach().double().cpu().reshape(-1))
        return d.view(-1, 1) * winv, -self.actnorm.bias.detach().double().cpu().reshape(-1)


class FlowUpsamplerNet(HipModule):
    def __init__(self, image_shape=(80, 80, 3), hidden_channels=64, K=12, L=None, actnorm_scale=1.0,
                 flow_permutation=None, flow_coupling="CondAffineSeparatedAndCond", LU_decomposed=False, opt=None):
        super().__init__()
        self.opt = opt
        self.L = _opt_get(opt, ["network_G", "flow", "L"], 2 if L is None else L)
        K = _opt_get(opt, ["network_G", "flow", "K"], K)
        n_extra = int(_opt_get(opt, ["network_G", "flow", "additionalFlowNoAffine"], 2))
        H, W, self.C = image_shape
        assert self.C == 3
        self.layers = nn.ModuleList()
        for _ in range(self.L):
            for _ in range(n_extra):  # arch_additionalFlowAffine, FlowUpsamplerNet.py:175-187
                self.layers.append(FlowStep(self.C, hidden_channels, actnorm_scale, "invconv", "noCoupling", LU_decomposed, opt))
            for _ in range(K):        # arch_FlowStep, :128-148
                self.layers.append(FlowStep(self.C, hidden_channels, actnorm_scale, "invconv", flow_coupling, LU_decomposed, opt))
        blocks = _opt_get(opt, ["network_G", "flow", "stackRRDB", "blocks"], [1, 3, 5, 7])
        affine_in = (len(blocks or []) + 1) * 64
        self.f = nn.Sequential(nn.Conv2d(affine_in, 2 * 3 * 64, 3, 1, 1))  # built, never called (:113-116)

    # ---- host-side preparation (once per weight set) --------------------------------------------
    def _prepare(self, split=0):
        """split = 3: every MFMA conv of the coupling nets in the fp32-class form (ops.PackedConv(split=3): the conditional
        feature and the hidden activations are hi / lo pairs then, decode_nhwc)."""
        dev = self.layers[0].actnorm.bias.device
        order = list(reversed(range(len(self.layers))))
        steps = []  # one entry per coupling step, in execution (reverse) order
        for pos, li in enumerate(order):
            layer = self.layers[li]
            if layer.flow_coupling == "noCoupling":
                continue
            A, c = layer.reverse_affine_fp64()
            nxt = pos + 1
            while nxt < len(order) and self.layers[order[nxt]].flow_coupling == "noCoupling":
                A2, c2 = self.layers[order[nxt]].reverse_affine_fp64()
                A, c = A2 @ A, A2 @ c + c2
                nxt += 1
            steps.append({"layer": layer, "M": A.float().flatten().tolist(), "t": c.float().tolist()})
        assert self.layers[order[0]].flow_coupling != "noCoupling", "a leading coupling-free step has no host"
        n = len(steps)
        wa_ft, ba_ft, wf0, bf0 = [], [], [], []
        for s, st in enumerate(steps):
            aff = st["layer"].affine
            w0, b0 = aff.fAffine[0].folded()           # [64, 65, 3, 3]: channel 0 = z1, 1..64 = conditional
            wa_ft.append(w0[:, 1:])
            ba_ft.append(b0)
            st["wz"] = w0[:, 0].reshape(64, 9).float().contiguous()
            st["c2"] = ops.PackedConv(*aff.fAffine[2].folded(), split=split)      # the four-launch form and encode_nhwc
            st["c4"] = ops.PackedConv(*aff.fAffine[4].folded(), split=split)
            if FUSED_STEP:     # the whole step as one launch: the filters as the fused kernel's fragment image (csrc/flow_fused.hip)
                st["image"] = ops.flow_fused_image(st["wz"], *aff.fAffine[2].folded(), *aff.fAffine[4].folded())
            f0w, f0b = aff.fFeatures[0].folded()
            wf0.append(f0w)
            bf0.append(f0b)
            st["eps"] = float(aff.affine_eps)
        # the z-independent second and third layers of all n feature nets: ONE grouped launch each (ops.conv2d_grouped)
        f2 = [st["layer"].affine.fFeatures[2].folded() for st in steps]
        f4 = [st["layer"].affine.fFeatures[4].folded() for st in steps]
        return {"steps": steps, "n": n, "ftA": ops.PackedConv(torch.cat(wa_ft, 0), torch.cat(ba_ft, 0), split=split),
                "f0": ops.PackedConv(torch.cat(wf0, 0), torch.cat(bf0, 0), split=split), "dev": dev,
                "f2": ops.packed_conv_batch(torch.stack([w for w, _ in f2]), torch.stack([b for _, b in f2]), split=split),
                "f4": ops.packed_conv_batch(torch.stack([w for w, _ in f4]), torch.stack([b for _, b in f4]), split=split)}

    def decode_nhwc(self, z, ft):
        """z: fp32 NHWC [B,h,w,3] (color_map); ft: bf16 NHWC [B,h,w,64] (cond_feat) -> latent fp32 NHWC."""
        # a cond_feat that arrives as a hi / lo pair (ConEncoder1 in fp32-class mode) selects the fp32-class form of every conv here:
        # cond_feat, h1f / h2f and the per-step h1 / h2 are pairs, the filters [w_hi | w_hi | w_lo] (tools/precision_sites.py: the
        # 16-bit cond_feat alone is 6.5e-4 of latent error, the nets' first-layer filters 5.8e-4)
        pair = getattr(ft, "_lo", None) is not None
        split = 3 if pair else 0
        P = self._packed(("flow", split), lambda: self._prepare(split)) if split else self._packed("flow", self._prepare)
        n = P["n"]
        B, H, W, _ = z.shape
        z = z.clone()
        ftA = ops.conv2d(ft, P["ftA"], out_mode=ops.OUT_NHWC_F32)                 # [B,h,w,n*64] fp32
        h1f = ops.conv2d(ft, P["f0"], act="relu", hilo=pair)                       # [B,h,w,n*64] bf16
        h2f = torch.empty_like(h1f)
        hF = torch.empty(B, H, W, n * 8, dtype=torch.float32, device=z.device)   # 6 of every 8 written and read
        # z-independent, batched up front: the n second layers, then the n third layers, one grouped launch each
        ops.conv2d_grouped(h1f, P["f2"], cin=64, in_step=64, out=h2f, out_step=64, act="relu", out_lo=torch.empty_like(h2f) if pair else None)
        ops.conv2d_grouped(h2f, P["f4"], cin=64, in_step=64, out=hF, out_step=8, out_mode=ops.OUT_NHWC_F32)
        if FUSED_STEP:                                                             # the sequential part: one launch per step
            zs = [z, torch.empty_like(z)]
            for s, st in enumerate(P["steps"]):
                ops.flow_step_fused(zs[s & 1], zs[(s + 1) & 1], ftA, 64 * s, st["image"], hF, 8 * s, st["M"], st["t"], st["eps"])
            return zs[n & 1]
        h1 = torch.empty(B, H, W, 64, dtype=ops.act_dtype(), device=z.device)
        h2 = torch.empty_like(h1)
        if pair:
            h1._lo, h2._lo = torch.empty_like(h1), torch.empty_like(h2)
        h4 = torch.empty(B, H, W, 4, dtype=torch.float32, device=z.device)
        for s, st in enumerate(P["steps"]):                                        # the sequential part (rounds 1-5: four launches per step)
            ops.flow_h1(z, ftA, 64 * s, st["wz"], out=h1)
            ops.conv2d(h1, st["c2"], act="relu", out=h2, hilo=pair)
            ops.conv2d(h2, st["c4"], out=h4, out_mode=ops.OUT_NHWC_F32)
            ops.flow_tail(z, h4, hF, 8 * s, st["M"], st["t"], st["eps"])
        return z

    # ---- ActNorm data-dependent initialisation (row a12: the first training forward of a fresh flow) ----------------
    def actnorms(self):
        """Every ActNorm2d in the order a training forward reaches them (FlowStep.normal_flow, FlowStep.py:75-98: the step's
        own, then fFeatures', then fAffine's, FlowAffineCouplingsAblation.py:55-77)."""
        out = []
        for layer in self.layers:
            out.append(layer.actnorm)
            if layer.flow_coupling != "noCoupling":
                f, a = layer.affine.fFeatures, layer.affine.fAffine
                out += [f[0].actnorm, f[2].actnorm, a[0].actnorm, a[2].actnorm]
        return out

    def needs_actnorm_init(self):
        return self.training and any(not m.inited for m in self.actnorms())

    @torch.no_grad()
    def initialize_actnorms_nhwc(self, gt, ft):
        """_ActNorm.initialize_parameters (FlowActNorms.py:32-46) for every ActNorm a training forward has not seen yet, in the
        reference's order: layer by layer through the normal direction, each ActNorm taking `bias = -mean`, `logs =
        log(scale / (std + 1e-6))` of ITS input over (B, H, W) as produced by the layers already initialised before it.  An
        ActNorm whose bias is not all-zero is only marked initialised (:36-38: a loaded checkpoint).  One-off (the first
        training step), so the pass is plain: raw conv (no bias, fp32 out) -> statistics kernel -> the folded conv the
        training forward itself runs.  gt: fp32 NHWC [B,h,w,3]; ft: 16-bit NHWC [B,h,w,64] (cond_feat)."""
        pending = [m for m in self.actnorms() if not m.inited]
        if not pending:
            return
        nonzero = torch.stack([(m.bias != 0).any() for m in pending]).cpu().tolist()     # the one host read of the pass
        need = {id(m) for m, nz in zip(pending, nonzero) if not nz}
        for m in pending:
            m.inited = True
        if not need:
            return
        B, H, W, _ = gt.shape
        F32 = ops.OUT_NHWC_F32
        ft = ft.contiguous()
        z = gt.detach().clone().contiguous()
        eye, zero3 = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0], [0.0, 0.0, 0.0]
        bps = ops.flow_blocks_per_sample(H * W)
        partial = torch.zeros(B * bps, dtype=torch.float32, device=z.device)
        hF = torch.zeros(B, H, W, 8, dtype=torch.float32, device=z.device)
        h4 = torch.zeros(B, H, W, 4, dtype=torch.float32, device=z.device)

        def init(an, x, C):
            if id(an) in need:
                T.actnorm_init_(x, C, an.bias.data, an.logs.data, scale=an.scale)

        def tail(net, h1):   # conv 1x1 (+ActNorm) -> relu -> Conv2dZeros of a coupling net, given its first layer's output
            if id(net[2].actnorm) in need:
                init(net[2].actnorm, ops.conv2d(h1, ops.PackedConv(net[2].weight), out_mode=F32), 64)
            h2 = ops.conv2d(h1, ops.PackedConv(*net[2].folded()), act="relu")
            return h2

        for layer in self.layers:
            init(layer.actnorm, z, 3)
            A, c, _ = layer.forward_affine_fp64()            # actnorm . invconv of this step with the fresh parameters
            T.flow_affine3_(z, A.float().flatten().tolist(), c.float().tolist())
            if layer.flow_coupling == "noCoupling":
                continue
            aff = layer.affine
            f, a = aff.fFeatures, aff.fAffine
            eps = float(aff.affine_eps)
            # feature-conditional affine (FlowAffineCouplingsAblation.py:55-59)
            if id(f[0].actnorm) in need:
                init(f[0].actnorm, ops.conv2d(ft, ops.PackedConv(f[0].weight), out_mode=F32), 64)
            h1 = ops.conv2d(ft, ops.PackedConv(*f[0].folded()), act="relu")
            ops.conv2d(tail(f, h1), ops.PackedConv(*f[4].folded()), out=hF, out_off=0, out_mode=F32)
            ops.flow_fwd_pre(z, hF, 0, eye, zero3, eps, partial)
            # self-conditional affine (:62-77): the first conv sees cat[z1, ft]
            w0 = a[0].weight.detach()
            if id(a[0].actnorm) in need:
                raw = ops.conv2d(ft, ops.PackedConv(w0[:, 1:].contiguous()), out_mode=F32)
                init(a[0].actnorm, T.flow_h1_raw(z, raw, 0, w0[:, 0].reshape(64, 9).float().contiguous()), 64)
            w0f, b0f = a[0].folded()
            ftA = ops.conv2d(ft, ops.PackedConv(w0f[:, 1:].contiguous(), b0f), out_mode=F32)
            h1 = ops.flow_h1(z, ftA, 0, w0f[:, 0].reshape(64, 9).float().contiguous())
            ops.conv2d(tail(a, h1), ops.PackedConv(*a[4].folded()), out=h4, out_off=0, out_mode=F32)
            ops.flow_fwd_post(z, h4, eps, partial)
        self._share_actnorm_init([m for m in self.actnorms() if id(m) in need])
        self.invalidate()

    @staticmethod
    def _share_actnorm_init(actnorms):
        """Data-parallel training (one process per GPU): every rank has just initialised these ActNorms from ITS OWN crops, and the
        gradient all-reduce only keeps EQUAL replicas equal.  The reference's single-process nn.DataParallel initialises the one
        shared copy of the parameters from the replica on the first device (in-place `.data.copy_` on storage the replica shares
        with the module, FlowActNorms.py:43-44): rank 0's statistics are everybody's -- one broadcast of the flat (bias, logs)."""
        import torch.distributed as dist

        if not actnorms or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        tensors = [t for m in actnorms for t in (m.bias.data, m.logs.data)]
        flat = torch.cat([t.reshape(-1) for t in tensors])
        if dist.get_backend() == "gloo" and flat.is_cuda:      # two ranks sharing one GPU in the tests: gloo carries host tensors
            host = flat.cpu()
            dist.broadcast(host, src=0)
            flat = host.to(flat.device)
        else:
            dist.broadcast(flat, src=0)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()

    def _prepare_forward(self):
        """Host-side composition for the normal direction: per coupling step the affine map of its own
        actnorm . invconv and of the coupling-free steps before it; the data-independent log-determinant."""
        steps, const_ld = [], 0.0
        A, c = torch.eye(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64)
        for layer in self.layers:
            A2, c2, ld = layer.forward_affine_fp64()
            A, c = A2 @ A, A2 @ c + c2
            const_ld += ld
            if layer.flow_coupling != "noCoupling":
                steps.append({"layer": layer, "M": A.float().flatten().tolist(), "t": c.float().tolist()})
                A, c = torch.eye(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64)
        assert self.layers[len(self.layers) - 1].flow_coupling != "noCoupling", "a trailing coupling-free step has no host"
        return {"steps": steps, "const_logdet_per_pixel": const_ld}

    def encode_nhwc(self, gt, ft, mean=None):
        """Normal direction.  gt: fp32 NHWC latent [B,h,w,3]; ft: bf16 NHWC cond_feat.  Returns (z fp32 NHWC,
        logdet fp64 [B], logp fp64 [B] or None): FlowUpsamplerNet.encode (:228-274) + GaussianDiag.logp."""
        if self.needs_actnorm_init():                # a fresh flow in train mode: FlowActNorms.py:82-83
            self.initialize_actnorms_nhwc(gt, ft)
        P = self._packed("flow", self._prepare)      # packed convs are shared with the reverse direction
        F_ = self._packed("flow_fwd", self._prepare_forward)
        rev = {id(st["layer"]): st for st in P["steps"]}
        order = {id(st["layer"]): i for i, st in enumerate(P["steps"])}
        n = P["n"]
        B, H, W, _ = gt.shape
        z = gt.clone()
        ftA = ops.conv2d(ft, P["ftA"], out_mode=ops.OUT_NHWC_F32)
        h1f = ops.conv2d(ft, P["f0"], act="relu")
        h2f = torch.empty_like(h1f)
        hF = torch.empty(B, H, W, n * 8, dtype=torch.float32, device=z.device)   # 6 of every 8 written and read
        ops.conv2d_grouped(h1f, P["f2"], cin=64, in_step=64, out=h2f, out_step=64, act="relu")
        ops.conv2d_grouped(h2f, P["f4"], cin=64, in_step=64, out=hF, out_step=8, out_mode=ops.OUT_NHWC_F32)
        bps = ops.flow_blocks_per_sample(H * W)
        partial = torch.zeros(2 * n, B * bps, dtype=torch.float32, device=z.device)
        h1 = torch.empty(B, H, W, 64, dtype=ops.act_dtype(), device=z.device)
        h2 = torch.empty_like(h1)
        h4 = torch.empty(B, H, W, 4, dtype=torch.float32, device=z.device)
        for k, fs in enumerate(F_["steps"]):
            st = rev[id(fs["layer"])]
            s = order[id(fs["layer"])]                 # slot of this step in the batched (reverse-ordered) buffers
            ops.flow_fwd_pre(z, hF, 8 * s, fs["M"], fs["t"], st["eps"], partial[2 * k])
            ops.flow_h1(z, ftA, 64 * s, st["wz"], out=h1)
            ops.conv2d(h1, st["c2"], act="relu", out=h2)
            ops.conv2d(h2, st["c4"], out=h4, out_mode=ops.OUT_NHWC_F32)
            ops.flow_fwd_post(z, h4, st["eps"], partial[2 * k + 1])
        red = ops.flow_nll_reduce(z, mean if mean is not None else z, partial, 2 * n)
        logdet = red[:, 0] + F_["const_logdet_per_pixel"] * (H * W)
        return z, logdet, (red[:, 1] if mean is not None else None)

    # ---- training: the normal direction with a backward (stage 2, row a12) --------------------------------
    def _train_params(self):
        """Kernel-ready parameters as DIFFERENTIABLE functions of the module's parameters (filter-sized torch ops:
        ActNorm / Conv2dZeros folds on the device, the 3x3 affine compositions and slogdet in fp64 on the host), in
        forward execution order.  Autograd carries the kernels' gradients through these folds to the raw parameters."""
        dev = self.layers[0].actnorm.bias.device
        # actnorm . invconv of ALL layers at once, in fp64 ON THE DEVICE (no host round trip: the training step must not
        # synchronise): A = W diag(e^logs), c = A b (FlowStep.py:83-88); log|det W| by the closed 3x3 formula
        # (Permutations.py:27 uses slogdet), actnorm's logdet is sum(logs) (FlowActNorms.py:93-98)
        Wm = torch.stack([l.invconv.weight for l in self.layers]).double()                 # [L, 3, 3]
        logs = torch.stack([l.actnorm.logs.reshape(-1) for l in self.layers]).double()     # [L, 3]
        bias = torch.stack([l.actnorm.bias.reshape(-1) for l in self.layers]).double()
        A2 = Wm * torch.exp(logs).unsqueeze(1)
        c2 = (A2 @ bias.unsqueeze(-1)).squeeze(-1)
        w = Wm.reshape(-1, 9).unbind(1)          # one unbind (backward: one stack) instead of 18 selects
        det = w[0] * (w[4] * w[8] - w[5] * w[7]) - w[1] * (w[3] * w[8] - w[5] * w[6]) + w[2] * (w[3] * w[7] - w[4] * w[6])
        const_ld = logs.sum() + torch.log(det.abs()).sum()
        # One row [A | c] per coupling step.  A coupling layer that directly follows another one takes its own row of `rows`
        # unchanged: runs of such layers are SLICES of it (one slice / one slice_backward per run, not a select + cat per layer --
        # a per-layer formulation put ~400 filter-sized kernels into every training step); only a coupling layer behind
        # coupling-free layers (2 of the 24 steps) needs the composition with what is pending.
        rows = torch.cat([A2.reshape(-1, 9), c2], 1)                                       # [L, 12]
        pieces, steps, pend, run = [], [], None, None
        for i, layer in enumerate(self.layers):
            coupling = layer.flow_coupling != "noCoupling"
            if coupling and pend is None:
                run = [i, i + 1] if run is None else [run[0], i + 1]
                steps.append(layer.affine)
                continue
            if run is not None:
                pieces.append(rows[run[0]:run[1]])
                run = None
            A, c = A2[i], c2[i]
            if pend is not None:
                A, c = A @ pend[0], A @ pend[1] + c
            if coupling:
                pieces.append(torch.cat([A.reshape(9), c]).unsqueeze(0))
                steps.append(layer.affine)
                pend = None
            else:
                pend = (A, c)
        if run is not None:
            pieces.append(rows[run[0]:run[1]])
        assert self.layers[len(self.layers) - 1].flow_coupling != "noCoupling"

        # batched folds: stack the raw parameters of all coupling steps once, then ONE op per fold for all steps
        def fold(convs):
            w = torch.stack([c.weight for c in convs])                                   # [n, co, ci, k, k]
            if hasattr(convs[0], "actnorm"):                                             # flow.Conv2d: conv + ActNorm
                s_ = torch.exp(torch.stack([c.actnorm.logs.reshape(-1) for c in convs]))
                b = torch.stack([c.actnorm.bias.reshape(-1) for c in convs]) * s_
            else:                                                                        # Conv2dZeros
                s_ = torch.exp(torch.stack([c.logs.reshape(-1) for c in convs]) * convs[0].logscale_factor)
                b = torch.stack([c.bias for c in convs]) * s_
            return w * s_.view(s_.shape[0], -1, 1, 1, 1), b

        a0w, a0b = fold([a.fAffine[0] for a in steps])
        f0w, f0b = fold([a.fFeatures[0] for a in steps])
        c2w, c2b = fold([a.fAffine[2] for a in steps])
        c4w, c4b = fold([a.fAffine[4] for a in steps])
        f2w, f2b = fold([a.fFeatures[2] for a in steps])
        f4w, f4b = fold([a.fFeatures[4] for a in steps])
        P = {"Mt": torch.cat(pieces).float(),
             "wz": a0w[:, :, 0].reshape(len(steps), 64, 9),
             "ftA_w": a0w[:, :, 1:].reshape(-1, 64, 3, 3), "ftA_b": a0b.reshape(-1),
             "f0_w": f0w.reshape(-1, 64, 3, 3), "f0_b": f0b.reshape(-1),
             "c2_w": c2w, "c2_b": c2b, "c4_w": c4w, "c4_b": c4b, "f2_w": f2w, "f2_b": f2b, "f4_w": f4w, "f4_b": f4b}
        return P, const_ld, float(steps[0].affine_eps), dev

    def train_nll_terms(self, gt, ft, mean, params=None, want_z=False):
        """gt, mean: fp32 NHWC [B,h,w,3]; ft: bf16 NHWC [B,h,w,64] (both may carry a tape).  Returns per-sample
        (logdet, logp) float64 tensors on the device, differentiable w.r.t. ft, mean and every flow parameter (want_z: plus
        the encoded latent z, fp32 NHWC, without a tape).  `params` =
        a `_train_params()` result computed EARLIER in the step: its tape nodes are then older than the encoder's, so their
        (tiny) backward runs after the encoder's backward has been enqueued."""
        P, const_ld, eps, dev = self._train_params() if params is None else params
        ld_data, logp, z = FlowNLLFn.apply(ft, mean, gt, eps, *[P[k] for k in _FLOW_KEYS])
        pixels = gt.shape[1] * gt.shape[2]
        if want_z:
            return ld_data + const_ld * pixels, logp, z
        return ld_data + const_ld * pixels, logp

    def encode(self, gt, rrdbResults, logdet=0.0, epses=None, y_onehot=None):
        ft = rrdbResults["cond_feat"] if isinstance(rrdbResults, dict) else rrdbResults
        z, ld, _ = self.encode_nhwc(to_nhwc(gt, bf16=False), to_nhwc(ft, bf16=True))
        return to_nchw(z), logdet + ld.float()

    def decode(self, rrdbResults, z, eps_std=None, epses=None, logdet=0.0, y_onehot=None):
        ft = rrdbResults["cond_feat"] if isinstance(rrdbResults, dict) else rrdbResults
        x = self.decode_nhwc(to_nhwc(z, bf16=False), to_nhwc(ft, bf16=True))
        return to_nchw(x), logdet  # the reverse log-determinant is unused by every caller on the path

    def forward(self, gt=None, rrdbResults=None, z=None, epses=None, logdet=0.0, reverse=False, eps_std=None, y_onehot=None):
        if reverse:
            return self.decode(rrdbResults, z, eps_std, epses=epses, logdet=logdet, y_onehot=y_onehot)
        assert gt is not None
        return self.encode(gt, rrdbResults, logdet=logdet, epses=epses, y_onehot=y_onehot)


_FLOW_KEYS = ("Mt", "wz", "ftA_w", "ftA_b", "f0_w", "f0_b", "c2_w", "c2_b", "c4_w", "c4_b", "f2_w", "f2_b", "f4_w", "f4_b")


def _wt(w, pad_to=None):
    """Packed filter of the data-gradient conv (flip / transpose / channel pad inside the pack kernel)."""
    return ops.PackedConv(w, dgrad_pad=w.shape[0] if pad_to is None else pad_to)


@_lib.keeps_precision
class FlowNLLFn(torch.autograd.Function):
    """The whole normal-direction flow (FlowUpsamplerNet.encode, :228-274) + Gaussian term as one tape node: forward =
    FlowUpsamplerNet.encode_nhwc keeping every step's activations, backward = the adjoint sweep of csrc/flow_bwd.hip with
    the coupling nets' conv gradients on the MFMA conv / GEMM kernels."""

    @staticmethod
    def forward(ctx, ft, mean, gt, eps, Mt, wz, ftA_w, ftA_b, f0_w, f0_b, c2_w, c2_b, c4_w, c4_b, f2_w, f2_b, f4_w, f4_b):
        n = Mt.shape[0]
        B, H, W, _ = gt.shape
        dev = gt.device
        Mt = Mt.detach().float().contiguous()            # [n, 12] on the device: the kernels read it there
        wz = wz.detach().float().contiguous()
        ft = ft.contiguous()
        ftA = ops.conv2d(ft, ops.PackedConv(ftA_w, ftA_b), out_mode=ops.OUT_NHWC_F32)
        h1f = ops.conv2d(ft, ops.PackedConv(f0_w, f0_b), act="relu")
        h2f = torch.empty_like(h1f)
        hF = torch.empty(B, H, W, n * 8, dtype=torch.float32, device=dev)   # 6 of every 8 written and read
        f2p, f4p = ops.packed_conv_batch(f2_w, f2_b), ops.packed_conv_batch(f4_w, f4_b)      # one pack launch per conv type
        c2p, c4p = ops.packed_conv_batch(c2_w, c2_b), ops.packed_conv_batch(c4_w, c4_b)
        ops.conv2d_grouped(h1f, f2p, cin=64, in_step=64, out=h2f, out_step=64, act="relu")     # the n z-independent nets: one
        ops.conv2d_grouped(h2f, f4p, cin=64, in_step=64, out=hF, out_step=8, out_mode=ops.OUT_NHWC_F32)   # grouped launch per layer
        # every step's input and mid-step latent are kept for the backward: the kernels write them in place of copies --
        # z_in[k] -pre-> z_pre[k] -post-> z_in[k + 1]; slot n of z_in is the encoded latent
        z_in = torch.empty(n + 1, B, H, W, 3, dtype=torch.float32, device=dev)
        z_in[0].copy_(gt.detach())
        z_pre = torch.empty(n, B, H, W, 3, dtype=torch.float32, device=dev)
        h1s = torch.empty(n, B, H, W, 64, dtype=ops.act_dtype(), device=dev)
        h2s = torch.empty_like(h1s)
        h4s = torch.empty(n, B, H, W, 4, dtype=torch.float32, device=dev)
        bps = ops.flow_blocks_per_sample(H * W)
        partial = torch.zeros(2 * n, B * bps, dtype=torch.float32, device=dev)
        for k in range(n):
            ops.flow_fwd_pre(z_in[k], hF, 8 * k, None, None, eps, partial[2 * k], Mt_dev=Mt[k], out=z_pre[k])
            ops.flow_h1(z_pre[k], ftA, 64 * k, wz[k], out=h1s[k])
            ops.conv2d(h1s[k], c2p[k], act="relu", out=h2s[k])
            ops.conv2d(h2s[k], c4p[k], out=h4s[k], out_mode=ops.OUT_NHWC_F32)
            ops.flow_fwd_post(z_pre[k], h4s[k], eps, partial[2 * k + 1], out=z_in[k + 1])
        z = z_in[n]
        mean = mean.contiguous()
        red = ops.flow_nll_reduce(z, mean, partial, 2 * n)
        ctx.eps = eps
        ctx.save_for_backward(ft, mean, z, h1f, h2f, hF, z_in, z_pre, h1s, h2s, h4s, wz, ftA_w, f0_w, c2_w, c4_w, f2_w, f4_w, Mt)
        z_out = z.clone()                       # the encoded latent, handed out as a plain (non-differentiable) result
        ctx.mark_non_differentiable(z_out)
        return red[:, 0].clone(), red[:, 1].clone(), z_out

    @staticmethod
    def backward(ctx, g_logdet, g_logp, _g_z=None):
        ft, mean, z, h1f, h2f, hF, z_in, z_pre, h1s, h2s, h4s, wz, ftA_w, f0_w, c2_w, c4_w, f2_w, f4_w, Mt = ctx.saved_tensors
        eps = ctx.eps
        n, B, H, W, _ = z_pre.shape
        dev = z.device
        gld = g_logdet.float().contiguous()
        gz, gmean = T.flow_nll_backward(z, mean, g_logp.float().contiguous())
        gftA = torch.empty(B, H, W, n * 64, dtype=ops.act_dtype(), device=dev)
        ghF = torch.empty(B, H, W, n * 8, dtype=ops.act_dtype(), device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        dMt, dwz = torch.empty(n, 12, **f32), torch.empty(n, 64, 9, **f32)
        P = B * H * W
        gh4s = torch.empty(n, B, H, W, 8, dtype=ops.act_dtype(), device=dev)     # kept: their filter gradients are batched below
        gh2s = torch.empty(n, B, H, W, 64, dtype=ops.act_dtype(), device=dev)
        c4t, c2t = ops.packed_conv_batch(c4_w, dgrad_pad=8), ops.packed_conv_batch(c2_w, dgrad_pad=64)   # data-gradient filters
        f4t, f2t = ops.packed_conv_batch(f4_w, dgrad_pad=8), ops.packed_conv_batch(f2_w, dgrad_pad=64)
        for k in reversed(range(n)):                                   # the sequential adjoint sweep
            T.flow_post_backward_(gz, z_pre[k], h4s[k], gld, eps, out=gh4s[k])
            ops.conv2d(gh4s[k], c4t[k], out=gh2s[k])
            T.act_backward_(gh2s[k], h2s[k], "relu")
            ops.conv2d(gh2s[k], c2t[k], out=gftA, out_off=64 * k)
            T.act_backward_(gftA, h1s[k], "relu", C=64, g_off=64 * k)
            T.flow_h1_backward_(gz, gftA, 64 * k, z_pre[k], wz[k], out=dwz[k])
            T.flow_pre_backward_(gz, z_in[k], hF, 8 * k, gld, None, None, eps, ghF, 8 * k, out=dMt[k], Mt_dev=Mt[k])
        # filter gradients of the 2 x n coupling convs: one launch per conv type, the steps being the groups of the NHWC
        # weight-gradient kernel (step-major tensors: the group stride is a whole step)
        o4 = T.conv_weight_grad_nhwc(3, h2s, gh4s, 8, 64, groups=n, x_gstride=P * 64, g_gstride=P * 8, shape=(B, H, W))   # [n,577,8]
        o2 = T.conv_weight_grad_nhwc(1, h1s, gh2s, 64, 64, groups=n, x_gstride=P * 64, g_gstride=P * 64, shape=(B, H, W))  # [n,65,64]
        gh2f, gh1f = torch.empty_like(h1f), torch.empty_like(h1f)
        # the z-independent feature nets: data gradients, the n nets as the groups of one launch per layer
        ops.conv2d_grouped(ghF, f4t, cin=8, in_step=8, out=gh2f, out_step=64)
        T.act_backward_(gh2f, h2f, "relu")
        ops.conv2d_grouped(gh2f, f2t, cin=64, in_step=64, out=gh1f, out_step=64)
        T.act_backward_(gh1f, h1f, "relu")
        # ... and their filter gradients: step s owns a channel block of both tensors (group stride = the block)
        of4 = T.conv_weight_grad_nhwc(3, h2f, ghF, 8, 64, groups=n, x_gstride=64, g_gstride=8)      # [n,577,8], 6 of 8 used
        of2 = T.conv_weight_grad_nhwc(1, h1f, gh2f, 64, 64, groups=n, x_gstride=64, g_gstride=64)   # [n,65,64]
        def wgrad_oihw(g16):       # [n * 64, 64, 3, 3] laid out like the (stacked) filter: its 24 slices are taken over by autograd as they are
            dw = torch.empty(n * 64, ft.shape[-1], 3, 3, dtype=torch.float32, device=dev)
            db = torch.empty(n * 64, dtype=torch.float32, device=dev)
            T.conv_weight_grad_oihw(3, ft, g16, n * 64, dw, 0, db)
            return dw, db

        df0w, df0b = wgrad_oihw(gh1f)
        dftAw, dftAb = wgrad_oihw(gftA)

        # .contiguous(): ONE permuting copy per conv family here instead of one per LAYER later -- the fold's backward (mul) keeps its
        # operand's strides and stack's backward hands AccumulateGrad 24 slices each; a slice that is not laid out like its
        # parameter is cloned (a copy_ launch per layer: 170 of a stage-2 step's 438 stock launches, tools/probes/small_ops.py),
        # a contiguous one is taken over as it is
        def w3(o, co):   # [n, 9*64 + 1, 8] -> ([n, co, 64, 3, 3], [n, co])
            return o[:, :576, :co].unflatten(1, (3, 3, 64)).permute(0, 4, 3, 1, 2).contiguous(), o[:, 576, :co].contiguous()

        def w1(o):       # [n, 64 + 1, 64] -> ([n, 64, 64, 1, 1], [n, 64])
            return o[:, :64].transpose(1, 2).contiguous().unsqueeze(-1).unsqueeze(-1), o[:, 64].contiguous()

        (dc4w, dc4b), (dc2w, dc2b), (df4w, df4b), (df2w, df2b) = w3(o4, 4), w1(o2), w3(of4, 6), w1(of2)
        gft = ops.conv2d(gh1f, _wt(f0_w))
        gft = ops.conv2d(gftA, _wt(ftA_w), residual=gft)
        return (gft, gmean, None, None, dMt, dwz, dftAw, dftAb, df0w, df0b, dc2w, dc2b, dc4w, dc4b, df2w, df2b, df4w, df4b)
