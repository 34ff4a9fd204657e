"""Training steps on the HIP path: stage 2 (row a12: conditional encoder + flow, NLL objective) and the data-parallel
plumbing both stages share.

Reference: LLFlowModel.optimize_parameters (code/models/LLFlow_model.py:181-250) with its optimizer setup (:90-122):
torch.optim.Adam over two parameter groups -- the flow ("other", lr_G, weight_decay_G) and the conditional encoder
(names containing '.RRDB.', lr_RRDB or lr_G, weight decay 1e-5); the `beta1` / `beta2` keys it passes are not Adam's
`betas`, so torch's defaults (0.9, 0.999) apply.  `scaler.scale(loss).backward(); scaler.step(opt); scaler.update()`
(:236-241): bf16 activations keep fp32's range, so the loss is not scaled -- but GradScaler's other job is: a step whose
gradients hold an inf / NaN is skipped, decided on the device (FlatAdam.step), and the scale / growth tracker are kept as the
reference's scaler would keep them.

MI355X design: parameters, gradients and both Adam moments of a group live in four flat fp32 buffers (module parameters
and their .grad are views), so the optimizer is ONE kernel launch per group and the data-parallel gradient mean is ONE
RCCL all-reduce per group over xGMI (106 MB at stage 2) instead of nn.DataParallel's per-step replicate / scatter /
gather through GPU 0 (LLFlow_model.py:72-75).
"""
import torch
import torch.distributed as dist

from . import ops
from . import train_ops as T


class FlatGroup:
    """Flat fp32 storage for a parameter group: w / grad / exp_avg / exp_avg_sq; parameters become views."""

    def __init__(self, params, lr, weight_decay=0.0, device=None, never_used=()):
        """never_used: parameters the training graph never reaches (they never get a gradient: torch.optim.Adam keeps no
        state for them and applies no weight decay).  Declared STATICALLY so that the set of parameters Adam updates is
        the same on every rank and every step: a rank-local `p.grad is None` test is not -- stage 2's `train_gt_ratio`
        branch leaves `RRDB.color_conv` without a gradient on the ranks that drew `mean = gt`, and those ranks would skip an
        update the others apply to the all-reduced gradient.  (With nn.DataParallel the reference sums the replicas'
        gradients on one device: a parameter moves whenever any replica used it -- the same rule.)"""
        self.params = [p for p in params if p.requires_grad]
        skip = {id(p) for p in never_used}
        self.has_grad = [id(p) not in skip for p in self.params]
        self.lr, self.weight_decay = float(lr), float(weight_decay)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device(device or "cpu")
        self.w = torch.empty(n, dtype=torch.float32, device=dev)
        self.g = torch.zeros_like(self.w)
        self.m = torch.zeros_like(self.w)
        self.v = torch.zeros_like(self.w)
        off = 0
        for p in self.params:
            k = p.numel()
            self.w[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.w[off:off + k].view(p.shape)
            off += k

    def zero_grad(self):
        """Gradients are left None: autograd then hands over each gradient tensor without an accumulation kernel per
        parameter; collect() gathers them into the flat buffer with a few concatenations."""
        for p in self.params:
            p.grad = None

    def active_ranges(self):
        """[lo, hi) element ranges of the flat buffer that Adam updates: everything but the `never_used` parameters.
        torch.optim.Adam (the reference's optimizer) skips a parameter whose .grad is None -- no moment update, no weight
        decay, no state entry: the parameters the graph never uses (flowUpsamplerNet.f, deformable_decoder.{scale,bias,enc,
        conv_out}) must not move under weight decay.  Merged, these are 1-3 ranges, i.e. 1-3 Adam launches per group."""
        out, off = [], 0
        for p, used in zip(self.params, self.has_grad):
            k = p.numel()
            if used:
                if out and out[-1][1] == off:
                    out[-1][1] = off + k
                else:
                    out.append([off, off + k])
            off += k
        return [(a, b) for a, b in out]

    def collect(self, chunk=64, grads=None):
        """grads: {id(param): gradient tensor or None} -- the gradients as a multi-grad hook receives them (arm_early_all_reduce),
        BEFORE AccumulateGrad has stored them in .grad; None = read .grad."""
        gof = (lambda p: p.grad) if grads is None else (lambda p: grads.get(id(p)))
        for i, p in enumerate(self.params):
            if gof(p) is not None and not self.has_grad[i]:
                raise RuntimeError("a parameter declared never_used received a gradient (index %d, shape %s)" % (i, tuple(p.shape)))
        off = 0
        for i in range(0, len(self.params), chunk):
            ps = self.params[i:i + chunk]
            n = sum(p.numel() for p in ps)
            if all(gof(p) is None for p in ps):
                self.g[off:off + n].zero_()
            else:
                # runs of parameters WITH a gradient are concatenated into their slot, runs without one zeroed in place (one launch per
                # run; a torch.zeros_like per missing parameter inside the cat was 38 launches of a stage-3 step)
                o2, run, run_off, miss_off = off, [], off, None
                for p in ps + [None]:                                   # (the sentinel closes the last run)
                    gp = gof(p) if p is not None else None
                    if p is not None and gp is not None:
                        if miss_off is not None:
                            self.g[miss_off:o2].zero_()
                            miss_off, run_off = None, o2
                        run.append(gp.reshape(-1).to(torch.float32))
                    else:
                        if run:
                            torch.cat(run, out=self.g[run_off:o2])
                            run = []
                        if p is not None and miss_off is None:
                            miss_off = o2
                    if p is not None:
                        o2 += p.numel()
                if miss_off is not None:
                    self.g[miss_off:o2].zero_()
            off += n
        if grads is not None:     # inside the backward pass: AccumulateGrad has yet to run for these parameters, and a .grad that
            return                # aliases the flat buffer would be accumulated INTO, under the exchange (finish_early_all_reduce re-points)
        off = 0
        for p, used in zip(self.params, self.has_grad):   # expose the gathered gradient as .grad (a view of the flat buffer);
            k = p.numel()                                  # parameters the graph never reaches keep .grad = None, as in torch
            p.grad = self.g[off:off + k].view(p.shape) if used else None
            off += k

    def arm_early_all_reduce(self):
        """Data-parallel steps only: start THIS group's gradient exchange the moment its last gradient exists, while the backward
        pass of the layers in front of it is still running -- stage 2's flow group (53 MB) is complete when FlowNLLFn.backward and
        the fold nodes return, and the conditional encoder's whole backward (the larger half of the step) then hides the ring
        all-reduce over xGMI.  The reference reduces everything through GPU 0 after backward (nn.DataParallel, LLFlow_model.py:
        71-74).  Call before `loss.backward()`; FlatAdam.step() picks the result up (and falls back to the blocking path if the
        hook never fired).  No-op at world size 1, so the single-GPU step and its hipGraph capture are untouched."""
        stale, self._early = getattr(self, "_early", None), None
        if stale is not None:          # a backward that raised between arm and step(): its hook must not fire on a later backward
            stale["handle"].remove()   # (a stray collect + all_reduce would desynchronise the ranks' collective order)
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) or not self.params:
            return
        used = [p for p, u in zip(self.params, self.has_grad) if u]
        state = {"fired": False, "work": None}

        def fire(grads):
            state["fired"] = True
            by_id = {id(p): g for p, g in zip(used, grads)}
            self.collect(grads=by_id)
            state["work"] = self.all_reduce(async_op=True)

        state["handle"] = torch.autograd.graph.register_multi_grad_hook(used, fire, mode="all")
        self._early = state

    def finish_early_all_reduce(self):
        """-> the world size if the early exchange ran (the flat gradient buffer then holds the all-reduced sum and every .grad is
        its view), else None (nothing armed, or the hook did not fire: the caller collects and reduces as usual)."""
        state, self._early = getattr(self, "_early", None), None
        if state is None:
            return None
        state["handle"].remove()
        if not state["fired"]:
            return None
        work = state["work"]
        if work is not None and not isinstance(work, int):
            work.wait()
        off = 0
        for i, (p, used) in enumerate(zip(self.params, self.has_grad)):   # collect()'s check, which the hook's id-keyed dict cannot make:
            if not used and p.grad is not None:                            # a never_used parameter must not have received a gradient
                raise RuntimeError("a parameter declared never_used received a gradient (index %d, shape %s)" % (i, tuple(p.shape)))
        for p, used in zip(self.params, self.has_grad):   # AccumulateGrad has since stored the LOCAL gradient tensors in .grad
            k = p.numel()
            p.grad = self.g[off:off + k].view(p.shape) if used else None
            off += k
        return dist.get_world_size()

    def all_reduce(self, async_op=False):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if self.g.is_cuda and dist.get_backend() == "gloo":
                # test rigs only (two ranks on one GPU cannot use RCCL): the same exchange bounced through host memory
                host = self.g.cpu()
                dist.all_reduce(host)
                self.g.copy_(host)
            elif async_op:
                return dist.all_reduce(self.g, async_op=True)   # the caller waits on the returned work (finish_early_all_reduce)
            else:
                dist.all_reduce(self.g)        # sum (RCCL over xGMI); the 1/world of the mean is folded into the Adam kernel
            return dist.get_world_size()
        return 1


class FlatAdam:
    """torch.optim.Adam over flat groups with ALL optimizer state on the device -- step count, bias corrections, an lr multiplier,
    GradScaler's found_inf / scale / growth tracker: a step has no host-side state and no host synchronisation (the Python launch
    loop runs ahead of the GPU; the whole step replays from a hipGraph, GraphedStep).  `device_state` is accepted for
    compatibility with round 2's two modes; the host-state mode is gone (its inf / NaN check would have cost a sync per step)."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8, device_state=True, loss_scaling=False):
        """loss_scaling: fp16 training -- `scale_loss()` multiplies the loss by the scale before backward and step() divides it out
        of the gradients (scaler.scale / scaler.unscale_, LLFlow_model.py:236-241); bf16 training keeps fp32's range and only runs
        the scaler's bookkeeping."""
        self.groups, self.betas, self.eps = groups, betas, eps
        self.device_state = True
        self.loss_scaling = bool(loss_scaling)
        dev = next(g.w.device for g in groups if g.w.numel())
        # GradScaler's state (torch.cuda.amp.GradScaler defaults, LLFlow_model.py:120): found_inf of the step in progress, the
        # scale and the growth tracker
        self.found_inf = torch.zeros(1, dtype=torch.int32, device=dev)
        self.scale = torch.full((1,), 65536.0, dtype=torch.float32, device=dev)
        self.growth_tracker = torch.zeros(1, dtype=torch.int32, device=dev)
        self.growth_factor, self.backoff_factor, self.growth_interval = 2.0, 0.5, 2000
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.state3 = torch.tensor([1.0, 1.0, 1.0], dtype=torch.float32, device=dev)   # bc1, sqrt(bc2), lr multiplier

    @property
    def t(self):
        return int(self.step_dev.item())       # a host read: for checkpoints / tests, never inside a step

    @t.setter
    def t(self, v):
        self.step_dev.fill_(int(v))

    def set_lr_scale(self, s):
        """Scheduler hook: lr_effective = group lr * s (a device scalar the kernels read)."""
        self.state3[2] = float(s)

    def zero_grad(self):
        for g in self.groups:
            g.zero_grad()

    def scale_loss(self, loss):
        """scaler.scale(loss): a device-side multiply (no host read of the scale); the identity without loss scaling."""
        return loss * self.scale.to(loss.dtype).reshape(()) if self.loss_scaling else loss

    def step(self):
        """scaler.step(optimizer); scaler.update() (LLFlow_model.py:240-241): gather + all-reduce the gradients, then ONE Adam
        step over every group -- unless an inf / NaN sits anywhere in them (checked AFTER the all-reduce, so every rank
        decides alike): then nothing moves, the step count stays and the scale backs off.  `last_step_skipped()` reads the flag
        (a host synchronisation) for callers that want to log it."""
        self.found_inf.zero_()
        worlds = []
        for g in self.groups:
            if g.w.numel() == 0:
                worlds.append(1)
                continue
            early = g.finish_early_all_reduce()      # the exchange that started during the backward pass, if one was armed
            if early is None:
                g.collect()
                early = g.all_reduce()
            worlds.append(early)
            T.grad_nonfinite_(g.g, self.found_inf)
        T.adam_prepare_guarded_(self.step_dev, self.state3, self.betas, self.found_inf)
        for g, world in zip(self.groups, worlds):
            if g.w.numel() == 0:
                continue
            for lo, hi in g.active_ranges():       # never-used parameters are skipped, as torch.optim.Adam does
                T.adam_step_dev_guarded_(g.w[lo:hi], g.g[lo:hi], g.m[lo:hi], g.v[lo:hi], self.state3, g.lr, self.betas, self.eps,
                                         g.weight_decay, 1.0 / world, self.found_inf, self.scale if self.loss_scaling else None)
        T.gradscaler_update_(self.scale, self.growth_tracker, self.found_inf, self.growth_factor, self.backoff_factor,
                             self.growth_interval)

    def last_step_skipped(self):
        return bool(int(self.found_inf.item()))

    def scaler_state_dict(self):
        """What torch.cuda.amp.GradScaler.state_dict() holds (checkpoint.save_training_state)."""
        return {"scale": float(self.scale.item()), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(self.growth_tracker.item())}

    def load_scaler_state_dict(self, sd):
        if not sd:
            return
        self.scale.fill_(float(sd.get("scale", 65536.0)))
        self.growth_tracker.fill_(int(sd.get("_growth_tracker", 0)))
        self.growth_factor = float(sd.get("growth_factor", 2.0))
        self.backoff_factor = float(sd.get("backoff_factor", 0.5))
        self.growth_interval = int(sd.get("growth_interval", 2000))


# Parameters the reference constructs but its graphs never call (they never receive a gradient, so torch.optim.Adam never
# touches them): the 320 -> 384 conv of FlowUpsamplerNet (FlowUpsamplerNet.py:113-116) and, in MultiScaleDecoder2, the
# fuse / scale / shift layers of the unused gating path and its own conv_out (deformableDecoder_arch.py:490-523).
STAGE2_NEVER_USED = ("flowUpsamplerNet.f.",)
STAGE3_NEVER_USED = ("deformable_decoder.scale.", "deformable_decoder.bias.", "deformable_decoder.enc.", "deformable_decoder.conv_out.")


class Stage2Trainer:
    """One optimisation step of the flow objective: a7 (frozen VQGAN encoder, no tape) -> a1 -> a4 -> mean NLL -> backward ->
    gradient mean over ranks -> Adam."""

    def __init__(self, netG, net_hq, lr_G=5e-4, lr_RRDB=None, weight_decay_G=0.0, train_rrdb=True, device_state=False, precision="fp16"):
        """precision: the 16-bit format of activations and activation gradients.  "fp16" (the default) is the reference's own AMP form
        (`@autocast()` forward + `scaler.scale(loss).backward()`, LLFlowVQGAN_arch.py:36, LLFlow_model.py:236-241): IEEE-half
        activations, the loss multiplied by the device-resident GradScaler scale, and the precision every gradient-parity bound of
        tests/test_gpu_train.py is stated in.  "bf16" (fp32 range, no loss scaling, ~1 % faster) rounds every stored tensor 8x
        coarser and carries NO gradient-parity claim (AFT decoder on the pipeline's inputs: 8 % median / 32 % max against fp16's
        2 % / 5 %)."""
        assert precision in ("bf16", "fp16")
        self.precision = precision
        self.netG, self.net_hq = netG.train(), net_hq.eval()
        for p in net_hq.parameters():
            p.requires_grad_(False)
        rrdb = [p for n, p in netG.named_parameters() if n.startswith("RRDB.")]
        other = [p for n, p in netG.named_parameters() if not n.startswith("RRDB.")]
        if not train_rrdb:  # train_RRDB_delay (LLFlow_model.py:142-150): the encoder joins after a fraction of the run
            for p in rrdb:
                p.requires_grad_(False)
        # always the reference's two groups [other, RRDB] (LLFlow_model.py:110-118); the RRDB group is EMPTY while the
        # conditional encoder is frozen (train_RRDB: false / before train_RRDB_delay), exactly as in the reference's
        # optimizer -- so `.state` files are interchangeable (glare_amd/checkpoint.py)
        never = [p for n, p in netG.named_parameters() if n.startswith(STAGE2_NEVER_USED)]
        self.opt = FlatAdam([FlatGroup(other, lr_G, weight_decay_G, never_used=never),
                             FlatGroup(rrdb if train_rrdb else [], lr_G if lr_RRDB is None else lr_RRDB, 1e-5,
                                       device=other[0].device)], device_state=device_state, loss_scaling=precision == "fp16")
        self.pack_cache = ops.PackCache([p for p in netG.parameters() if p.requires_grad])   # packed filters live across steps

    def draw_branch(self):
        """The step's one host-side random decision (LLFlowVQGAN_arch.py:95): is the Gaussian's mean the ground truth?"""
        return self.netG._mean_is_gt(None)

    def step(self, gt_img, lr_img, mean_is_gt=None):
        """gt_img: fp32 NCHW ground-truth crop in [0,1]; lr_img: fp32 NCHW low-light crop (log domain).  Returns the loss."""
        return float(self.step_tensor(gt_img, lr_img, mean_is_gt))

    def step_tensor(self, gt_img, lr_img, mean_is_gt=None):
        """The step without any host synchronisation: returns the loss as a device tensor.  mean_is_gt: None = draw with
        probability train_gt_ratio as the reference does; True / False = forced (GraphedStep draws before choosing a graph)."""
        flag = self.draw_branch() if mean_is_gt is None else bool(mean_is_gt)
        with ops.use_precision(self.precision):
            with torch.no_grad(), ops.auto_cout_tile():
                gt_latent = self.net_hq.encode_nhwc(gt_img)            # LLFlow_model.py:200-201
            self.opt.zero_grad()
            with self.pack_cache:
                nll = self.netG.train_nll(gt_latent, lr_img, mean_is_gt=flag)   # :215
                loss = nll.mean()
                self.opt.groups[0].arm_early_all_reduce()              # N > 1: the flow group's all-reduce runs under the encoder's backward
                self.opt.scale_loss(loss).backward()                   # :236  scaler.scale(loss).backward()
            self.opt.step()                                            # :240-241  scaler.step(); scaler.update()
        self.netG.invalidate()                                     # packed inference weights are stale now
        return loss.detach()


class Stage3Trainer:
    """One optimisation step of stage 3 (row a13, VQLLFLOWDModel.optimize_parameters, VQLLFLOWD_model.py:187-232): the
    conditional encoder, the flow (reverse) and the VQGAN decoder run without a tape exactly as in inference
    (VQLLFLOWDeformable_arch.py:231-248); only `deformable_decoder` is trained.  Loss (:217-223): L1 + 0.01 * VGG16-feature
    perceptual + 0.2 * (1 - MS-SSIM(normalize=True)).  The perceptual network's weights are the caller's (`perceptual=`): the
    reference downloads torchvision's pretrained VGG16, which is not available offline; pass None to train on L1 + MS-SSIM."""

    def __init__(self, netG, net_hq, lr_G=5e-5, weight_decay_G=0.0, perceptual=None, use_msssim=True, device_state=False, precision="fp16",
                 frozen_fp32_class=False):
        """precision: as Stage2Trainer ("fp16" = the reference's AMP form, the default).  frozen_fp32_class: run the FROZEN conditional
        encoder + flow in the fp32-class inference form (hi / lo operand pairs, three MFMA passes per conv) instead of the single-pass
        fp16 form.  Default False: the reference runs these nets under `@autocast()` in the stage-3 step (VQLLFLOWDeformable_arch.py:
        222-248), i.e. single-pass fp16 is ITS arithmetic; the fp32-class form serves inference's index contract and costs the step
        ~3.5 ms (23.0 -> 19.5 ms at 1 x 256^2, round 4 / 5)."""
        from . import autograd as A
        from . import losses
        from .modules import encoder_decoder as ED

        assert precision in ("bf16", "fp16")      # as Stage2Trainer
        self.precision = precision
        self._front = lambda: ED.fp32_class(bool(frozen_fp32_class))

        self.A, self.losses, self.perceptual, self.use_msssim = A, losses, perceptual, use_msssim
        self.last_terms = {}
        self.netG, self.net_hq = netG, net_hq.eval()
        for p in net_hq.parameters():
            p.requires_grad_(False)
        for n, p in netG.named_parameters():
            p.requires_grad_(n.startswith("deformable_decoder."))
        dd = [p for n, p in netG.named_parameters() if n.startswith("deformable_decoder.")]
        # the reference's optimizer always has the two groups [other, RRDB] (VQLLFLOWD_model.py:113-121); at stage 3 the
        # conditional encoder is frozen (fix_modules), so its group is empty -- kept for `.state` file compatibility
        never = [p for n, p in netG.named_parameters() if n.startswith(STAGE3_NEVER_USED)]
        self.opt = FlatAdam([FlatGroup(dd, lr_G, weight_decay_G, never_used=never), FlatGroup([], lr_G, 1e-5, device=dd[0].device)],
                            device_state=device_state, loss_scaling=precision == "fp16")
        self.pack_cache = ops.PackCache(dd)

    def step(self, gt_img, lr_img):
        """gt_img: fp32 NCHW in [0,1]; lr_img: fp32 NCHW low-light crop (log domain).  Returns the loss."""
        return float(self.step_tensor(gt_img, lr_img))

    def step_tensor(self, gt_img, lr_img):
        G = self.netG
        with ops.use_precision(self.precision):
            with torch.no_grad(), ops.auto_cout_tile(), self._front():
                enc = G.RRDB.forward_nhwc(lr_img)
                lat = G.flowUpsamplerNet.decode_nhwc(enc["color_map"], enc["cond_feat"])
                _, _, feats = self.net_hq.decode_nhwc(lat, want_image=False)
            self.opt.zero_grad()
            with self.pack_cache:
                rec = G.deformable_decoder.train_nhwc(lat, feats, enc["mid_feat"], whole_batch_mean=True)
                loss, self.last_terms = stage3_loss(rec, gt_img, self.perceptual, self.use_msssim)
                self.opt.scale_loss(loss).backward()
            self.opt.step()
        G.deformable_decoder.invalidate()   # only its packed inference weights went stale; the frozen nets keep theirs
        return loss.detach()


def stage3_loss(rec_nhwc, gt_nchw, perceptual=None, use_msssim=True):
    """total = l1 + 0.01 * percep + 0.2 * (1 - msssim(sr, gt, normalize=True)) on the NHWC fp32 network output
    (VQLLFLOWD_model.py:209-223).  Returns (total, {term: float-able tensor})."""
    from . import autograd as A
    from . import losses, ops

    gt_nchw = gt_nchw.float().contiguous()
    terms = {"l1_loss": A.l1_clamp_loss(rec_nhwc, gt_nchw)}
    if perceptual is not None or use_msssim:
        sr = A.clamp01(rec_nhwc)
        gt_nhwc = ops.nchw_to_nhwc(gt_nchw, bf16=False)
        if perceptual is not None:
            terms["percep_loss"] = perceptual(sr, gt_nhwc) * 0.01
        if use_msssim:
            terms["ssim_loss"] = (1 - losses.msssim(sr, gt_nhwc, normalize=True)) * 0.2
    return sum(terms.values()), terms


class GraphedStep:
    """A training step captured once into a hipGraph (torch.cuda.CUDAGraph) and replayed: the ~5 000 kernel launches of a step
    at the reference's crop sizes are launch-bound from Python, the graph removes the host from the loop.  Needs a trainer
    built with device_state=True (no host-side optimizer state) and fixed input shapes; inputs are copied into static buffers.
    World size 1 only (a collective inside the captured region is not attempted).

    A stage-2 step has one host-side random decision (mean = ground truth with probability train_gt_ratio,
    LLFlowVQGAN_arch.py:95) that changes the captured kernel sequence: the decision is drawn on the host BEFORE the replay
    and selects one of two graphs, both captured up front (capture_all)."""

    def __init__(self, trainer, gt_img, lr_img, warmup=3):
        self.trainer = trainer
        self.branching = hasattr(trainer, "draw_branch")
        self.gt, self.lr = gt_img.clone(), lr_img.clone()
        self.warmup = warmup
        self.graphs = {}
        self.capture_all()

    def capture_all(self):
        """Warm up and capture EVERY branch now (stage 2: mean = color_map and mean = gt), never lazily mid-training: a capture
        bakes in host state of that moment -- the packed-filter job table of the PackCache (a branch that packs a filter the
        other never touched must have done so BEFORE any graph holds the table), allocator pools, lazily built folds.  Call again
        after `resume_training` / loading weights: the old graphs are dropped."""
        self.graphs = {}
        # the mean = gt branch exists only when it can be drawn (train_gt_ratio > 0: 0.2 in confs/train_stage2_LOL.yml, 0 in LOL.yml)
        both = self.branching and float(getattr(self.trainer.netG, "train_gt_ratio", 0.0)) > 0.0
        flags = (False, True) if both else (False,)
        # The warm-up runs REAL steps (allocator pools, lazy folds, the PackCache's job table must exist before the capture), on the
        # batch cloned at construction: the training state is snapshotted around it and put back, so that capturing -- at construction
        # or again after `resume_training` / a weight load -- never advances the optimisation (it used to apply
        # warmup x branches Adam updates on a stale batch and move the step counter and the scaler's tracker).  Only a FRESH flow's
        # ActNorm initialisation (a property of the first training forward, FlowActNorms.py:82-83) deliberately survives.
        opt = self.trainer.opt
        snap = [(t, t.clone()) for g in opt.groups for t in (g.w, g.m, g.v)]
        snap += [(t, t.clone()) for t in (opt.step_dev, opt.state3, opt.scale, opt.growth_tracker)]
        actnorms = [m for m in self.trainer.netG.modules() if hasattr(m, "inited") and hasattr(m, "logs")]
        # "fresh" as FlowActNorms.py:36-38 defines it: not yet initialised AND still all-zero (a seeded / loaded ActNorm is only marked)
        fresh = {id(m) for m in actnorms if not m.inited and not bool((m.bias != 0).any())}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            keep = []
            if fresh:
                # The data-dependent initialisation happens inside the FIRST training forward (FlowActNorms.py:82-83).  It is run here on
                # its own -- a forward without backward or optimizer step -- and cloned BEFORE any warm-up step: cloned after the warm-up
                # (as this code did) the values put back were the initialisation PLUS warmup x branches Adam updates on the construction
                # batch, with their moments and the step counter rewound to zero.
                self._init_forward(flags[0])
                keep = [(t, t.clone()) for m in actnorms if id(m) in fresh for t in (m.bias.data, m.logs.data)]
            for _ in range(self.warmup):            # allocator / lazy-init warm-up outside the capture, every branch
                for f in flags:
                    self._eager(f)
            for t, v in snap:
                t.copy_(v)
            for t, v in keep:                       # the parameters live inside the restored flat buffers: put the initialisation back
                t.copy_(v)
        torch.cuda.current_stream().wait_stream(side)
        self._invalidate()
        for f in flags:
            self._capture(f)

    def _eager(self, flag):
        return self.trainer.step_tensor(self.gt, self.lr, flag) if self.branching else self.trainer.step_tensor(self.gt, self.lr)

    def _init_forward(self, flag):
        """The forward half of a stage-2 step only (what initialises a fresh flow's ActNorms), no tape kept, nothing updated."""
        tr = self.trainer
        if not isinstance(tr, Stage2Trainer):
            return
        flow = tr.netG.flowUpsamplerNet
        with ops.use_precision(tr.precision), torch.no_grad():
            if flow.needs_actnorm_init():                      # exactly what train_nll() does first (LLFlowVQGAN_arch.train_nll)
                with ops.auto_cout_tile():
                    gt_latent = tr.net_hq.encode_nhwc(self.gt)
                enc0 = tr.netG.RRDB.forward_nhwc(self.lr)
                flow.initialize_actnorms_nhwc(gt_latent.detach(), enc0["cond_feat"])

    def _capture(self, flag):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = self._eager(flag)
        self.graphs[flag] = (g, loss)
        # the capture only RECORDS: it must not advance the optimizer.  (Capturing runs no kernels; the host-side step
        # counter lives on the device in device_state mode, so nothing to undo.)

    def _invalidate(self):
        # the host code inside step_tensor (netG.invalidate()) ran at capture time only: packed inference weights built
        # since then (a validation pass between steps) would be stale after this replay
        tr = self.trainer
        (tr.netG.deformable_decoder if isinstance(tr, Stage3Trainer) else tr.netG).invalidate()

    def step_tensor(self, gt_img, lr_img, mean_is_gt=None):
        flag = False
        if self.branching:
            flag = self.trainer.draw_branch() if mean_is_gt is None else bool(mean_is_gt)
        self.gt.copy_(gt_img)
        self.lr.copy_(lr_img)
        if flag not in self.graphs:      # a forced branch the configuration itself cannot draw (tests): captured on first use --
            self._capture(flag)          # safe, the PackCache keeps superseded job tables alive for the graphs that hold them
        g, loss = self.graphs[flag]
        g.replay()
        self._invalidate()
        return loss

    def step(self, gt_img, lr_img):
        return float(self.step_tensor(gt_img, lr_img))
