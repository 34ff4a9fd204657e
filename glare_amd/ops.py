"""Thin Python front-ends of the C ABI (include/glare_hip.h): argument checking, output allocation
with torch (device memory + current stream are torch's: plumbing, not compute), one ctypes call.
Every function raises on CPU tensors -- the HIP kernels are the only implementation."""
import ctypes
import os

import torch

from . import _lib
from ._lib import act_dtype, check, inference_precision, precision, ptr, require_cuda, stream_handle, use_precision  # noqa: F401

_i = ctypes.c_int
_ll = ctypes.c_longlong


def vq_nearest(z_tokens, codebook, want_zq=True):
    """z_tokens [N,3] fp32, codebook [K,3] fp32 -> (idx int64 [N], zq [N,3] or None).
    Restates quantize.py:280-285; indices bit-exact (see csrc/vq.hip)."""
    require_cuda(z_tokens, codebook)
    assert z_tokens.dtype == torch.float32 and codebook.dtype == torch.float32
    z_tokens = z_tokens.contiguous()
    codebook = codebook.contiguous()
    n, dim = z_tokens.shape
    idx = torch.empty(n, dtype=torch.int64, device=z_tokens.device)
    zq = torch.empty_like(z_tokens) if want_zq else None
    check(_lib.lib().glare_vq_nearest_f32(ptr(z_tokens), ptr(codebook), _ll(n), _i(codebook.shape[0]), _i(dim),
                                          ptr(idx), ptr(zq), stream_handle()), "glare_vq_nearest_f32")
    return idx, zq


# ---- convolution --------------------------------------------------------------------------------
ACT = {"none": 0, "relu": 1, "sigmoid": 2, "swish": 3}
OUT_NHWC_BF16, OUT_NHWC_F32, OUT_PLANAR_F32, OUT_PLANAR_BF16 = 0, 1, 2, 3


class ConvDesc(ctypes.Structure):
    """Mirror of `glare_conv_desc` (include/glare_hip.h)."""
    _fields_ = [("in_", ctypes.c_void_p), ("in2", ctypes.c_void_p), ("weight_packed", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("B", _i), ("H", _i), ("W", _i),
                ("Cin", _i), ("in_pitch", _i), ("in_off", _i),
                ("Cin2", _i), ("in2_pitch", _i), ("in2_off", _i),
                ("Cout", _i), ("out_pitch", _i), ("out_off", _i),
                ("res_pitch", _i), ("res_off", _i),
                ("ksize", _i), ("stride", _i), ("upsample", _i), ("act", _i), ("out_mode", _i),
                ("plane_pitch", _ll), ("gn_partial", ctypes.c_void_p), ("cout_tile", _i),
                ("residual_lo", ctypes.c_void_p), ("out_lo", ctypes.c_void_p),
                ("groups", _i), ("group_in_step", _i), ("group_out_step", _i), ("k_wrap", _i),
                ("gn_coef", ctypes.c_void_p), ("gn_swish", _i)]


# k_wrap = 2 (round 5): the K order of an fp32-class 3x3 conv in which every x_hi halo tile is staged ONCE for the two filter halves it
# meets -- per 16-channel chunk c the stages (x_hi(c), w_hi(c)), (x_hi(c), w_lo(c)), and behind all of them the x_lo segment against
# w_hi -- instead of the segment order [x_hi | x_lo | x_hi] that reads x_hi twice.  GLARE_SPLIT_A_REUSE=0: round 4's order.
SPLIT_A_REUSE = os.environ.get("GLARE_SPLIT_A_REUSE", "1") == "1"


def split_filter(w, split, reuse_kc=0):
    """K segments of an fp32-class conv (glare_conv_desc.k_wrap), along the input channels of an fp32 [..., cout, cin, k, k] filter:
    split = 3: [w_hi | w_hi | w_lo] for the operand segments [x_hi | x_lo | x_hi]; split = 2: [w_hi | w_lo] for [x_hi | x_hi]
    (a 16-bit activation against a 22-bit filter).  w_hi = round16(w) is what the pack kernel makes of the first segments by
    itself; w_lo = round16(w - w_hi).
    reuse_kc = the kernel's channels per stage (16 for 3x3): the k_wrap = 2 order [w_hi(c0) | w_lo(c0) | w_hi(c1) | w_lo(c1) | ... | w_hi]
    for the stages (x_hi(c), x_hi(c) again, ..., then x_lo): the same three products, x_hi staged once."""
    if not split:
        return w
    assert split in (2, 3)
    lo = w - w.to(act_dtype()).float()
    if reuse_kc:
        assert split == 3 and w.shape[-3] % reuse_kc == 0
        lead, (cin, kh, kw) = w.shape[:-3], w.shape[-3:]
        n = cin // reuse_kc
        inter = torch.stack([w.reshape(*lead, n, reuse_kc, kh, kw), lo.reshape(*lead, n, reuse_kc, kh, kw)], dim=len(lead) + 1)
        return torch.cat([inter.reshape(*lead, 2 * cin, kh, kw), w], dim=-3).contiguous()
    return torch.cat([w, w, lo] if split == 3 else [w, lo], dim=-3).contiguous()


# Round 6: inference rounds its single-pass 16-bit filters with error feedback (csrc/conv_igemm.hip filter_feedback_kernel) instead of
# round-to-nearest per weight: each weight within one ulp of the channel's largest weights, the rounding errors of an output channel summing
# to < 1/2 ulp (round-to-nearest: a random walk) -- the coherent per-channel gain error is what stages D / E lost (profiles/r06_filter_feedback.txt).
# GLARE_FILTER_FEEDBACK=0: round-to-nearest (rounds 1-5).
FILTER_FEEDBACK = os.environ.get("GLARE_FILTER_FEEDBACK", "1") == "1"


# ... but NOT the per-image attention filters (glare_attn_fold_groupnorm_f32's `feedback`): measured with it, the third weight set's |dPSNR vs GT|
# went 0.0061 -> 0.0121 dB and PSNR(ours, oracle) 65.2 -> 63.8 dB -- those filters multiply the RAW activation (per-channel means and scales
# all over the place), so the vanishing sum of a row's rounding errors buys nothing and the doubled per-weight variance costs.
# GLARE_ATTN_FOLD_FEEDBACK=1 switches it on for A/B runs.
ATTN_FOLD_FEEDBACK = os.environ.get("GLARE_ATTN_FOLD_FEEDBACK", "0") == "1"


def filter_feedback_round(w):
    """fp32 [cout, ...] filter -> fp32 tensor of 16-bit-representable values (current precision), error feedback along the trailing axes."""
    require_cuda(w)
    w = w.detach().float().contiguous()
    out = torch.empty_like(w)
    check(_lib.lib().glare_filter_feedback_round_bf16(ptr(w), ptr(out), _i(w.shape[0]), _ll(w.numel() // w.shape[0]), stream_handle()),
          "glare_filter_feedback_round_bf16")
    return out


class PackedConv:
    """A conv filter packed once for the MFMA kernel (weights bf16 stage-ordered, bias fp32)."""

    def __init__(self, weight_oihw, bias=None, dgrad_pad=None, upsample_subpixel=False, cout_tile=0, split=0, feedback=False):
        """dgrad_pad = P: pack the filter of the data-gradient conv (P >= cout input channels, cin outputs) instead.
        cout_tile: 0 = the default output-channel tile for this cout, or 64 / 32 for launches too small to fill the chip with it
        (conv_cout_tile(); the packed image is tile-specific and conv2d passes the tile on).
        upsample_subpixel: pack the four 2x2 sub-pixel filters of "nearest x2 upsample, then this 3x3 conv" (conv2d then runs
        desc.upsample = 2: 16 instead of 36 tap-MACs per source pixel); feedback (with it): round the four phase filters with error
        feedback per output channel (filter_feedback_round; plain filters are fed pre-rounded by modules._base.packed_conv)."""
        require_cuda(weight_oihw)
        w = weight_oihw.detach().float().contiguous()
        self.split = int(split)
        self.k_wrap = 1 if split else 0
        if split:       # the fp32-class form: conv2d() reads the activation's hi / lo pair against [w_hi | w_hi | w_lo]
            assert dgrad_pad is None and not upsample_subpixel and not cout_tile
            if split == 3 and SPLIT_A_REUSE and w.shape[-1] == 3 and w.shape[1] % 16 == 0:
                self.k_wrap = 2
                w = split_filter(w, split, reuse_kc=16)
            else:
                w = split_filter(w, split)
        cout, cin, kh, kw = w.shape
        assert kh == kw and kh in (1, 3)
        self.ksize = kh
        self.subpixel = bool(upsample_subpixel)
        self.cout_tile = 0
        # for_tile() re-packs a 3x3 filter with more than 64 output channels for a narrower tile on demand: only then is the fp32
        # source kept (for filters built on the fly -- AttnBlock's folded 1x1s, 9.4 MB sub-pixel upsample filters, the per-step
        # filters of training -- it would otherwise stay resident for the model's lifetime)
        self._src = (w, bias, dgrad_pad) if (kh == 3 and not upsample_subpixel and not split and (cout if dgrad_pad is None else cin) > 64) else None
        self._tiles = {}
        lib = _lib.lib()
        if cout_tile and not self.subpixel:
            one = packed_conv_batch(w.unsqueeze(0), None if bias is None else bias.detach().float().reshape(1, -1), dgrad_pad, cout_tile)[0]
            self.cout, self.cin, self.packed, self.bias, self.w16, self.cout_tile = one.cout, one.cin, one.packed, one.bias, None, cout_tile
            return
        if self.subpixel:
            assert kh == 3 and dgrad_pad is None
            self.cout, self.cin = cout, cin
            lib.glare_conv2d_upsample_packed_weight_elems.restype = _ll
            n = lib.glare_conv2d_upsample_packed_weight_elems(_i(cout), _i(cin))
            assert n > 0
            self.packed = torch.empty(n, dtype=act_dtype(), device=w.device)
            if feedback:
                # the four phase filters (the taps of the 3x3 filter that read one source pixel, summed) rounded with error feedback per
                # (phase, output channel) -- the sub-pixel form of filter_feedback_round: rows a = 0: r = 0 <- {0}, r = 1 <- {1, 2};
                # a = 1: r = 0 <- {0, 1}, r = 1 <- {2}; columns alike
                rows = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
                ph = torch.stack([torch.stack([torch.stack([w[:, :, list(rows[a][r])][:, :, :, list(rows[b][c])].sum(dim=(2, 3)) for c in (0, 1)], -1)
                                               for r in (0, 1)], -2) for a in (0, 1) for b in (0, 1)])           # [4, cout, cin, 2, 2]
                ph = filter_feedback_round(ph.reshape(4 * cout, cin, 2, 2))
                check(lib.glare_conv2d_pack_weight_upsample_phases(ptr(ph), _i(cout), _i(cin), ptr(self.packed), stream_handle()),
                      "glare_conv2d_pack_weight_upsample_phases")
            else:
                check(lib.glare_conv2d_pack_weight_upsample(ptr(w), _i(cout), _i(cin), ptr(self.packed), stream_handle()),
                      "glare_conv2d_pack_weight_upsample")
            self.bias = None if bias is None else bias.detach().float().contiguous()
            return
        lib.glare_conv2d_packed_weight_elems.restype = _ll
        self.cout, self.cin = (cout, cin) if dgrad_pad is None else (cin, dgrad_pad)
        n = lib.glare_conv2d_packed_weight_elems(_i(self.cout), _i(self.cin), _i(kh))
        assert n > 0
        self.packed = torch.empty(n, dtype=act_dtype(), device=w.device)
        if dgrad_pad is None:
            check(lib.glare_conv2d_pack_weight(ptr(w), _i(cout), _i(cin), _i(kh), ptr(self.packed), stream_handle()),
                  "glare_conv2d_pack_weight")
        else:
            check(lib.glare_conv2d_pack_weight_dgrad(ptr(w), _i(cout), _i(cin), _i(kh), _i(dgrad_pad), ptr(self.packed),
                                                     stream_handle()), "glare_conv2d_pack_weight_dgrad")
        self.bias = None if bias is None else bias.detach().float().contiguous()
        self.w16 = None
        self.w16_lo = None
        if split:
            self.cin = cin // split          # the conv's own input channels; the packed image holds `split` K segments of them
            if kh == 1 and split == 3 and lib.glare_conv1x1_ws_split_supported(_i(self.cin), _i(cout)):
                # the fp32-class form on the weight-stationary kernel: the two halves of the filter as plain [Cout][Cin] images
                # (segment 0 of `w` is the fp32 filter -- the pack kernel rounds it to w_hi --, segment 2 is w_lo)
                w2 = w.reshape(cout, 3, self.cin)
                self.w16, self.w16_lo = (torch.empty(cout, self.cin, dtype=act_dtype(), device=w.device) for _ in range(2))
                for src, dst in ((w2[:, 0].contiguous(), self.w16), (w2[:, 2].contiguous(), self.w16_lo)):
                    check(lib.glare_conv1x1_ws_pack_weight(ptr(src), _i(cout), _i(self.cin), ptr(dst), stream_handle()), "glare_conv1x1_ws_pack_weight")
        if kh == 1 and dgrad_pad is None and not split and lib.glare_conv1x1_ws_supported(_i(cin), _i(cout)):
            # the weight-stationary 1x1 kernel (csrc/conv1x1.hip) takes the filter as plain bf16 [Cout][Cin]
            self.w16 = torch.empty(cout, cin, dtype=act_dtype(), device=w.device)
            check(lib.glare_conv1x1_ws_pack_weight(ptr(w), _i(cout), _i(cin), ptr(self.w16), stream_handle()), "glare_conv1x1_ws_pack_weight")


def packed_conv_batch(weights, biases=None, dgrad_pad=None, cout_tile=0, split=0):
    """weights fp32 [n, cout, cin, k, k] (n filters of one shape) -> n PackedConv whose packed images come from ONE launch.
    biases: fp32 [n, cout] or None; dgrad_pad, cout_tile, split as in PackedConv."""
    require_cuda(weights, biases)
    w = weights.detach().float().contiguous()
    if split:
        assert dgrad_pad is None
        k_wrap = 2 if (split == 3 and SPLIT_A_REUSE and w.shape[-1] == 3 and w.shape[2] % 16 == 0) else 1
        w = split_filter(w, split, reuse_kc=16 if k_wrap == 2 else 0)
    else:
        k_wrap = 0
    n, cout, cin, kh, kw = w.shape
    assert kh == kw and kh in (1, 3)
    lib = _lib.lib()
    lib.glare_conv2d_packed_weight_elems_tile.restype = _ll
    oc, ic = (cout, cin) if dgrad_pad is None else (cin, dgrad_pad)
    elems = lib.glare_conv2d_packed_weight_elems_tile(_i(oc), _i(ic), _i(kh), _i(cout_tile))
    assert elems > 0
    packed = torch.empty(n, elems, dtype=act_dtype(), device=w.device)
    check(lib.glare_conv2d_pack_weight_batched(ptr(w), _i(n), _i(cout), _i(cin), _i(kh), _i(0 if dgrad_pad is None else dgrad_pad),
                                               _i(cout_tile), ptr(packed), stream_handle()), "glare_conv2d_pack_weight_batched")
    b = None if biases is None else biases.detach().float().contiguous()
    out = []
    for k in range(n):
        pc = PackedConv.__new__(PackedConv)
        pc.ksize, pc.subpixel, pc.cout, pc.cin, pc.packed, pc.w16, pc.cout_tile = kh, False, oc, ic // (split or 1), packed[k], None, cout_tile
        pc.split, pc.k_wrap = int(split), k_wrap
        pc.bias = None if b is None else b[k]
        out.append(pc)
    return out


PACK_BLOCK_ELEMS = 2048   # GLARE_PACK_BLOCK_ELEMS (include/glare_hip.h)


class _PackJob(ctypes.Structure):   # glare_pack_job (include/glare_hip.h)
    _fields_ = [("w", ctypes.c_void_p), ("out", ctypes.c_void_p), ("total", ctypes.c_longlong), ("block_begin", ctypes.c_longlong),
                ("cout", ctypes.c_int), ("cin", ctypes.c_int), ("ksize", ctypes.c_int), ("tn", ctypes.c_int), ("ksteps", ctypes.c_int),
                ("n_stages", ctypes.c_int), ("cin_real", ctypes.c_int), ("kind", ctypes.c_int)]


class PackCache:
    """Packed filters of the TRAINABLE convs kept across training steps.  The autograd conv nodes ask for (weight, dgrad_pad)
    through packed_for(); a weight that lives in one of `params` (the flat optimizer buffers: stable addresses) is packed on first
    use into a buffer that stays, and entering the context -- the trainer does at the start of every step's forward, so optimizer
    updates AND checkpoint loads are both seen -- re-packs ALL of them in ONE launch (glare_conv2d_pack_multi).  Filters derived
    from parameters on the fly (the flow's folded filters) are not cached: their tensors are new every step."""

    def __init__(self, params):
        self.ptrs = {p.data_ptr(): tuple(p.shape) for p in params if p.dim() == 4}
        self.entries = {}          # (data_ptr, dgrad_pad) -> PackedConv
        self.table = None          # (device job table, n_jobs, total_blocks)
        self._retired = []         # superseded job tables, kept alive

    def __enter__(self):
        global PACK_CACHE
        self._prev, PACK_CACHE = PACK_CACHE, self
        self.refresh()
        return self

    def __exit__(self, *exc):
        global PACK_CACHE
        PACK_CACHE = self._prev

    def get(self, weight, bias, dgrad_pad, cout_tile=0):
        key = (weight.data_ptr(), dgrad_pad, cout_tile)
        pc = self.entries.get(key)
        if pc is None:
            if self.ptrs.get(weight.data_ptr()) != tuple(weight.shape) or not weight.is_contiguous() or weight.dtype != torch.float32:
                return None
            pc = PackedConv(weight, None, dgrad_pad=dgrad_pad, cout_tile=cout_tile)
            pc._param = weight.detach()
            self.entries[key] = pc
            if self.table is not None:
                self._retired.append(self.table)   # a captured hipGraph may still launch the multi-shape pack from the old job
            self.table = None                      # table: it stays allocated (and correct for ITS entries) for the cache's lifetime
        pc.bias = None if bias is None else bias.detach().float().contiguous()
        return pc

    def refresh(self):
        if not self.entries:
            return
        if self.table is None:
            lib = _lib.lib()
            jobs, begin = [], 0
            for (_, dgrad_pad, cout_tile), pc in self.entries.items():
                cout, cin, kh, _ = pc._param.shape
                kinds = [(1, pc.packed)] if dgrad_pad is not None else [(0, pc.packed)] + ([(2, pc.w16)] if pc.w16 is not None else [])
                for kind, dst in kinds:
                    j = _PackJob()
                    check(lib.glare_conv2d_pack_job_init(ctypes.byref(j), _i(kind), ptr(pc._param), _i(cout), _i(cin), _i(kh),
                                                         _i(dgrad_pad or 0), _i(cout_tile), ptr(dst)), "glare_conv2d_pack_job_init")
                    j.block_begin = begin
                    begin += (j.total + PACK_BLOCK_ELEMS - 1) // PACK_BLOCK_ELEMS
                    jobs.append(j)
            arr = (_PackJob * len(jobs))(*jobs)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            dev = next(iter(self.entries.values())).packed.device
            self.table = (host.to(dev), len(jobs), begin)
        tab, n, blocks = self.table
        check(_lib.lib().glare_conv2d_pack_multi(ptr(tab), _i(n), _ll(blocks), stream_handle()), "glare_conv2d_pack_multi")


PACK_CACHE = None   # the PackCache of the training step in progress (None: every packed_for() packs)


def packed_for(weight, bias=None, dgrad_pad=None, cout_tile=0):
    """The packed image of a conv filter for the autograd nodes: from the step's PackCache when the filter is a trainable parameter
    it knows, packed on the spot otherwise."""
    if PACK_CACHE is not None:
        pc = PACK_CACHE.get(weight, bias, dgrad_pad, cout_tile)
        if pc is not None:
            return pc
    return PackedConv(weight, bias, dgrad_pad=dgrad_pad, cout_tile=cout_tile)


def conv_cout_tile(B, OH, OW, cout):
    """Output-channel tile (0 = the default) that gives a conv of this output size enough workgroups (glare_conv2d_cout_tile)."""
    t = _lib.lib().glare_conv2d_cout_tile(_i(B), _i(OH), _i(OW), _i(cout))
    return 0 if t >= (128 if cout > 64 else (64 if cout > 32 else 32)) else t


# Measurement hook (bench.py `rooflines`): when a dict {family: list}, launches of that family are bracketed by a pair of events
# recorded on the stream the kernel is launched on, and (start, end, algorithmic FLOPs, algorithmic bytes) is appended.
# Families: "conv3x3" (the 3x3 / sub-pixel implicit-GEMM launches), "dcn" (DCNv2 forward).  None (default) records nothing.
LAUNCH_EVENTS = None


class _timed_launch:
    def __init__(self, family, flops, nbytes):
        self.rec = LAUNCH_EVENTS.get(family) if LAUNCH_EVENTS is not None else None
        self.flops, self.nbytes = flops, nbytes

    def __enter__(self):
        if self.rec is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()

    def __exit__(self, *exc):
        if self.rec is not None:
            self.ev[1].record()
            self.rec.append((self.ev[0], self.ev[1], float(self.flops), float(self.nbytes)))


# Measurement hook (tools/train_bench.py flops): when a dict, every MFMA launch family adds its algorithmic FLOPs (2 x MACs) to
# FLOP_COUNTER[family]; None (the default) counts nothing.
FLOP_COUNTER = None


def count_flops(family, flops):
    if FLOP_COUNTER is not None:
        FLOP_COUNTER[family] = FLOP_COUNTER.get(family, 0.0) + float(flops)


def _for_tile(pc, tile):
    """The same filter packed for a narrower output-channel tile (cached on the PackedConv)."""
    other = pc._tiles.get(tile)
    if other is None:
        w, bias, dgrad_pad = pc._src
        other = pc._tiles[tile] = PackedConv(w, bias, dgrad_pad=dgrad_pad, cout_tile=tile)
    return other


# 3x3 convs of the cached (inference) filters pick the output-channel tile per launch (conv_cout_tile): a no-op at the BASELINE batch
# (every launch fills the chip with the 128-wide tile), +20..60 % on the convs of small launches.  OFF by default: with a narrower
# tile the fused GroupNorm statistics are summed in another fp32 order, and inference promises that a batch of 8 equals eight
# single-image runs bit for bit (tests/test_gpu_graph.py); the trainers switch it on around their frozen networks
# (`with ops.auto_cout_tile():`), where the batch is 1 or 2 and nothing is compared across batch sizes.
AUTO_COUT_TILE = False


class auto_cout_tile:
    def __enter__(self):
        global AUTO_COUT_TILE
        self._prev, AUTO_COUT_TILE = AUTO_COUT_TILE, True

    def __exit__(self, *exc):
        global AUTO_COUT_TILE
        AUTO_COUT_TILE = self._prev

CONV1X1_WEIGHT_STATIONARY = True   # False: every 1x1 conv through the implicit-GEMM kernel (conv_igemm.hip, KS = 1)


def _conv1x1_ws(x, pc, cin, in_off, act, residual, res_off, out, out_off, gn_stats, hilo=False):
    B, H, W, pitch = x.shape
    N = H * W
    if out is None:
        out = torch.empty(B, H, W, pc.cout, dtype=act_dtype(), device=x.device)
    lib = _lib.lib()
    gn_part = None
    if gn_stats:
        assert out_off == 0 and out.shape[3] == pc.cout
        lib.glare_conv1x1_ws_gn_partial_elems.restype = _ll
        gn_part = torch.empty(lib.glare_conv1x1_ws_gn_partial_elems(_i(B), _ll(N), _i(pc.cout)), dtype=torch.float32, device=x.device)
    if hilo:
        out_lo = torch.empty_like(out)
        rlo = getattr(residual, "_lo", None) if residual is not None else None
        check(lib.glare_conv1x1_ws_hilo_bf16(ptr(x), _i(pitch), _i(in_off), ptr(pc.w16), _ll(0), ptr(pc.bias), _i(0), ptr(residual), ptr(rlo),
                                             _i(residual.shape[3] if residual is not None else 0), _i(res_off), ptr(out), ptr(out_lo),
                                             _i(out.shape[3]), _i(out_off), _i(B), _ll(N), _i(cin), _i(pc.cout), _i(ACT[act]), ptr(gn_part),
                                             stream_handle()), "glare_conv1x1_ws_hilo_bf16")
        out._lo = out_lo
    else:
        _conv1x1_ws_plain(lib, x, pitch, in_off, pc, residual, res_off, out, out_off, B, N, cin, act, gn_part)
    if gn_stats:
        stats = torch.empty(B, 1, 32, 2, dtype=torch.float32, device=x.device)
        check(lib.glare_conv1x1_ws_gn_reduce(ptr(gn_part), ptr(stats), _i(B), _ll(N), _i(pc.cout), stream_handle()),
              "glare_conv1x1_ws_gn_reduce")
        out._gn_stats = stats
    return out


def _conv1x1_ws_split(x, pc, cin, in_off, act, residual, res_off, out, out_off, gn_stats, hilo):
    """The fp32-class 1x1 conv (PackedConv(split=3)) on the weight-stationary kernel (glare_conv1x1_ws_split_bf16): x and the filter as
    hi / lo pairs, the output a pair (hilo) or its 16-bit rounding."""
    xlo = getattr(x, "_lo", None)
    assert xlo is not None and xlo.shape == x.shape and xlo.dtype == x.dtype and xlo.is_contiguous(), "a split-3 filter contracts the activation's hi / lo pair"
    B, H, W, pitch = x.shape
    N = H * W
    if out is None:
        out = torch.empty(B, H, W, pc.cout, dtype=act_dtype(), device=x.device)
    lib = _lib.lib()
    gn_part = None
    if gn_stats:
        lib.glare_conv1x1_ws_gn_partial_elems.restype = _ll
        gn_part = torch.empty(lib.glare_conv1x1_ws_gn_partial_elems(_i(B), _ll(N), _i(pc.cout)), dtype=torch.float32, device=x.device)
    out_lo = rlo = None
    if hilo:
        out_lo = getattr(out, "_lo", None)
        if out_lo is None:
            out_lo = torch.empty_like(out)
    if residual is not None:
        assert residual.dtype == act_dtype() and residual.is_contiguous()
        rlo = getattr(residual, "_lo", None)
    check(lib.glare_conv1x1_ws_split_bf16(ptr(x), ptr(xlo), _i(pitch), _i(in_off), ptr(pc.w16), ptr(pc.w16_lo), ptr(pc.bias), ptr(residual),
                                          ptr(rlo), _i(residual.shape[3] if residual is not None else 0), _i(res_off), ptr(out), ptr(out_lo),
                                          _i(out.shape[3]), _i(out_off), _i(B), _ll(N), _i(cin), _i(pc.cout), _i(ACT[act]), ptr(gn_part),
                                          stream_handle()), "glare_conv1x1_ws_split_bf16")
    if gn_stats:
        stats = torch.empty(B, 1, 32, 2, dtype=torch.float32, device=x.device)
        check(lib.glare_conv1x1_ws_gn_reduce(ptr(gn_part), ptr(stats), _i(B), _ll(N), _i(pc.cout), stream_handle()),
              "glare_conv1x1_ws_gn_reduce")
        out._gn_stats = stats
    if out_lo is not None:
        out._lo = out_lo
    return out


def _conv1x1_ws_plain(lib, x, pitch, in_off, pc, residual, res_off, out, out_off, B, N, cin, act, gn_part):
    check(lib.glare_conv1x1_ws_bf16(ptr(x), _i(pitch), _i(in_off), ptr(pc.w16), ptr(pc.bias), ptr(residual),
                                    _i(residual.shape[3] if residual is not None else 0), _i(res_off), ptr(out), _i(out.shape[3]),
                                    _i(out_off), _i(B), _ll(N), _i(cin), _i(pc.cout), _i(ACT[act]), ptr(gn_part), stream_handle()),
          "glare_conv1x1_ws_bf16")


def conv1x1_per_image(x, w16, bias, residual=None, gn_stats=False, hilo=False):
    """1x1 conv with ONE FILTER PER IMAGE (csrc/conv1x1.hip): x bf16 NHWC [B,H,W,Cin], w16 bf16 [B,Cout,Cin], bias fp32 [B,Cout],
    optional residual bf16 [B,H,W,Cout] -> bf16 NHWC [B,H,W,Cout] (+ the fused GroupNorm statistics of the output).
    hilo: as conv2d(hilo=True) -- the result's remainder half rides along as `._lo`, the residual's `._lo` is added."""
    require_cuda(x, w16, bias, residual)
    B, H, W, cin = x.shape
    cout = w16.shape[1]
    assert x.dtype == w16.dtype == act_dtype() and x.is_contiguous() and w16.is_contiguous() and tuple(w16.shape) == (B, cout, cin)
    assert bias.dtype == torch.float32 and bias.is_contiguous() and tuple(bias.shape) == (B, cout)
    N = H * W
    count_flops("conv k1", 2.0 * B * N * cin * cout)
    out = torch.empty(B, H, W, cout, dtype=act_dtype(), device=x.device)
    lib = _lib.lib()
    gn_part = None
    if gn_stats:
        lib.glare_conv1x1_ws_gn_partial_elems.restype = _ll
        gn_part = torch.empty(lib.glare_conv1x1_ws_gn_partial_elems(_i(B), _ll(N), _i(cout)), dtype=torch.float32, device=x.device)
    if residual is not None:
        assert residual.dtype == act_dtype() and residual.is_contiguous() and residual.shape == out.shape
    if hilo:
        out_lo = torch.empty_like(out)
        rlo = getattr(residual, "_lo", None) if residual is not None else None
        check(lib.glare_conv1x1_ws_hilo_bf16(ptr(x), _i(cin), _i(0), ptr(w16), _ll(cout * cin), ptr(bias), _i(cout), ptr(residual), ptr(rlo),
                                             _i(cout if residual is not None else 0), _i(0), ptr(out), ptr(out_lo), _i(cout), _i(0), _i(B),
                                             _ll(N), _i(cin), _i(cout), _i(0), ptr(gn_part), stream_handle()), "glare_conv1x1_ws_hilo_bf16")
        out._lo = out_lo
    else:
        check(lib.glare_conv1x1_ws_image_bf16(ptr(x), _i(cin), _i(0), ptr(w16), _ll(cout * cin), ptr(bias), _i(cout), ptr(residual),
                                              _i(cout if residual is not None else 0), _i(0), ptr(out), _i(cout), _i(0), _i(B), _ll(N),
                                              _i(cin), _i(cout), _i(0), ptr(gn_part), stream_handle()), "glare_conv1x1_ws_image_bf16")
    if gn_stats:
        stats = torch.empty(B, 1, 32, 2, dtype=torch.float32, device=x.device)
        check(lib.glare_conv1x1_ws_gn_reduce(ptr(gn_part), ptr(stats), _i(B), _ll(N), _i(cout), stream_handle()),
              "glare_conv1x1_ws_gn_reduce")
        out._gn_stats = stats
    return out


def attn_fold_groupnorm(stats, HW, gamma, beta, eps, wq, bq, wo, bo):
    """Per-image filters of an AttnBlock whose GroupNorm is folded into its two 1x1 convs (glare_attn_fold_groupnorm_f32):
    stats fp32 [B, splits, 32, 2]; wq / wo fp32 [C, C], bq / bo fp32 [C] -> (wq_b bf16 [B,C,C], bq_b fp32 [B,C], wo_b, bo_b)."""
    require_cuda(stats, gamma, beta, wq, bq, wo, bo)
    B, splits = stats.shape[0], stats.shape[1]
    C = wq.shape[0]
    assert stats.dtype == torch.float32 and stats.is_contiguous() and tuple(stats.shape[2:]) == (32, 2)
    for t in (gamma, beta, wq, bq, wo, bo):
        assert t.dtype == torch.float32 and t.is_contiguous()
    wq_b = torch.empty(B, C, C, dtype=act_dtype(), device=stats.device)
    wo_b = torch.empty_like(wq_b)
    bq_b = torch.empty(B, C, dtype=torch.float32, device=stats.device)
    bo_b = torch.empty_like(bq_b)
    check(_lib.lib().glare_attn_fold_groupnorm_f32(ptr(stats), _i(splits), _i(B), _ll(HW), _i(C), ptr(gamma), ptr(beta), _f(eps), ptr(wq),
                                                   ptr(bq), ptr(wo), ptr(bo), ptr(wq_b), ptr(bq_b), ptr(wo_b), ptr(bo_b),
                                                   _i(int(ATTN_FOLD_FEEDBACK)), stream_handle()),
          "glare_attn_fold_groupnorm_f32")
    return wq_b, bq_b, wo_b, bo_b


def conv2d_grouped(x, pcs, *, cin, in_step, out, out_step, in_off=0, out_off=0, act="none", out_mode=OUT_NHWC_BF16, out_lo=None):
    """len(pcs) independent convs of ONE shape in a single launch (glare_conv_desc.groups): conv g reads channels
    [in_off + g * in_step, + cin) of x and writes channels [out_off + g * out_step, + cout) of `out`.  pcs: the PackedConvs that
    packed_conv_batch() returned for them (consecutive packed images, consecutive biases)."""
    require_cuda(x, out)
    G = len(pcs)
    p0 = pcs[0]
    assert x.dtype == act_dtype() and x.is_contiguous() and out.is_contiguous() and p0.packed.dtype == x.dtype
    stride_w = p0.packed.numel()
    assert all(pc.packed.data_ptr() == p0.packed.data_ptr() + g * stride_w * p0.packed.element_size() for g, pc in enumerate(pcs)), \
        "grouped conv: filters must come from one packed_conv_batch()"
    if p0.bias is not None:
        assert all(pc.bias.data_ptr() == p0.bias.data_ptr() + g * p0.cout * 4 for g, pc in enumerate(pcs))
    B, H, W, pitch = x.shape
    d = ConvDesc()
    d.in_, d.B, d.H, d.W = x.data_ptr(), B, H, W
    d.Cin, d.in_pitch, d.in_off = cin, pitch, in_off
    d.out, d.Cout, d.out_pitch, d.out_off = out.data_ptr(), p0.cout, out.shape[3], out_off
    d.weight_packed = p0.packed.data_ptr()
    d.bias = p0.bias.data_ptr() if p0.bias is not None else None
    d.cout_tile = getattr(p0, "cout_tile", 0)
    d.ksize, d.stride, d.upsample, d.act, d.out_mode = p0.ksize, 1, 0, ACT[act], out_mode
    d.groups, d.group_in_step, d.group_out_step = G, in_step, out_step
    split = getattr(p0, "split", 0)
    if split:                                  # fp32-class: the hi / lo operand pair against [w_hi | w_hi | w_lo] (k_wrap)
        d.k_wrap = getattr(p0, "k_wrap", 1) or 1
        if split == 3:
            xlo = getattr(x, "_lo", None)
            assert xlo is not None and xlo.shape == x.shape and xlo.dtype == x.dtype and xlo.is_contiguous(), "split-3 filter: x needs its lo half"
            d.in2, d.Cin2, d.in2_pitch, d.in2_off = xlo.data_ptr(), cin, pitch, in_off
    if out_lo is not None:
        assert out_mode == OUT_NHWC_BF16 and out_lo.shape == out.shape and out_lo.dtype == out.dtype and out_lo.is_contiguous()
        d.out_lo = out_lo.data_ptr()
        out._lo = out_lo
    # ALGORITHMIC FLOPs: an fp32-class (split) conv counts once, as in the LAUNCH_EVENTS path and the bench text; the two extra MFMA
    # passes it executes are recorded under their own family so that no MFU derived from "conv k*" is inflated 3x
    count_flops("conv k%d" % p0.ksize, 2.0 * G * B * H * W * p0.ksize ** 2 * p0.cin * p0.cout)
    if split:
        count_flops("conv k%d extra fp32-class passes (executed, not algorithmic)" % p0.ksize, 2.0 * G * B * H * W * p0.ksize ** 2 * p0.cin * p0.cout * (split - 1))
    check(_lib.lib().glare_conv2d_bf16(ctypes.byref(d), stream_handle()), "glare_conv2d_bf16")
    return out


def conv2d(x, pc, *, x2=None, cin=None, in_off=0, cin2=None, in2_off=0, stride=1, upsample=False, act="none",
           residual=None, res_off=0, out=None, out_off=0, out_mode=OUT_NHWC_BF16, plane_pitch=0, gn_stats=False, hilo=False,
           gn_prologue=None):
    """x: NHWC bf16 [B,H,W,pitch] (channels [in_off, in_off+cin) are used), optional x2 concatenated
    after it.  Returns (or fills `out`) per out_mode; planar outputs are [B, planes, plane_pitch].
    hilo: the output keeps 22 mantissa bits as a hi / lo pair -- the returned tensor is `hi` (what every consumer reads), its
    remainder rides along as `out._lo`; a `residual` carrying `._lo` is added with it (glare_conv_desc.out_lo)."""
    require_cuda(x, x2, residual, out)
    assert x.dtype == act_dtype() and x.dim() == 4 and x.is_contiguous(), (x.dtype, act_dtype())
    assert pc.packed.dtype == x.dtype, "filter packed under another precision"
    B, H, W, pitch = x.shape
    cin = pitch - in_off if cin is None else cin
    split = getattr(pc, "split", 0)
    if FLOP_COUNTER is not None:
        opix = B * H * W * (4 if upsample else 1) // (stride * stride)
        count_flops("conv k%d" % pc.ksize, 2.0 * opix * pc.ksize ** 2 * pc.cin * pc.cout)      # algorithmic: counted once (see conv2d_grouped)
        if split:
            count_flops("conv k%d extra fp32-class passes (executed, not algorithmic)" % pc.ksize, 2.0 * opix * pc.ksize ** 2 * pc.cin * pc.cout * (split - 1))
    if hilo and pc.ksize == 1 and not split:
        assert x2 is None and stride == 1 and not upsample and out is None and cin == pc.cin and getattr(pc, "w16", None) is not None
        return _conv1x1_ws(x, pc, cin, in_off, act, residual, res_off, None, 0, gn_stats, hilo=True)
    if (CONV1X1_WEIGHT_STATIONARY and getattr(pc, "w16", None) is not None and not split and x2 is None and stride == 1 and not upsample
            and out_mode == OUT_NHWC_BF16 and cin == pc.cin and pitch % 8 == 0 and in_off % 8 == 0
            and (out is None or (out.dtype == act_dtype() and out.shape[3] % 8 == 0 and out_off % 8 == 0))
            and (residual is None or (residual.shape[3] % 8 == 0 and res_off % 8 == 0)) and (not gn_stats or out is None)):
        if residual is not None:
            assert residual.dtype == act_dtype() and residual.is_contiguous()
        return _conv1x1_ws(x, pc, cin, in_off, act, residual, res_off, out, out_off, gn_stats)
    if (split == 3 and pc.ksize == 1 and CONV1X1_WEIGHT_STATIONARY and getattr(pc, "w16_lo", None) is not None and x2 is None and stride == 1
            and not upsample and out_mode == OUT_NHWC_BF16 and cin == pc.cin and pitch % 8 == 0 and in_off % 8 == 0 and gn_prologue is None
            and (out is None or (out.dtype == act_dtype() and out.shape[3] % 8 == 0 and out_off % 8 == 0 and (not hilo or (out_off == 0 and out.shape[3] == pc.cout))))
            and (residual is None or (residual.shape[3] % 8 == 0 and res_off % 8 == 0)) and (not gn_stats or (pc.cout % 128 == 0 and (out is None or (out_off == 0 and out.shape[3] == pc.cout))))):
        return _conv1x1_ws_split(x, pc, cin, in_off, act, residual, res_off, out, out_off, gn_stats, hilo)
    d = ConvDesc()
    d.in_, d.in2 = x.data_ptr(), (x2.data_ptr() if x2 is not None else None)
    d.B, d.H, d.W = B, H, W
    d.Cin, d.in_pitch, d.in_off = cin, pitch, in_off
    if split:      # fp32-class: K segments [x_hi | x_lo | x_hi] against the filter packed as [w_hi | w_hi | w_lo] (k_wrap)
        assert x2 is None and not upsample and cin == pc.cin, (cin, pc.cin)
        d.k_wrap = getattr(pc, "k_wrap", 1) or 1
        assert d.k_wrap != 2 or gn_prologue is None
        if split == 3:
            x2 = getattr(x, "_lo", None)
            assert x2 is not None and x2.shape == x.shape, "a split-3 filter contracts the activation's hi / lo pair: x._lo is missing"
            cin2, in2_off = cin, in_off
    if x2 is not None:
        assert x2.dtype == act_dtype() and x2.is_contiguous() and x2.shape[:3] == x.shape[:3]
        d.in2 = x2.data_ptr()
        d.Cin2 = x2.shape[3] - in2_off if cin2 is None else cin2
        d.in2_pitch, d.in2_off = x2.shape[3], in2_off
    assert pc.cin == d.Cin + (0 if split else d.Cin2), (pc.cin, d.Cin, d.Cin2)
    IH, IW = (2 * H, 2 * W) if upsample else (H, W)
    OH, OW = ((IH + 1 - 3) // 2 + 1, (IW + 1 - 3) // 2 + 1) if stride == 2 else (IH, IW)
    if out is None:
        if out_mode == OUT_NHWC_BF16:
            out = torch.empty(B, OH, OW, pc.cout, dtype=act_dtype(), device=x.device)
        elif out_mode == OUT_NHWC_F32:
            out = torch.empty(B, OH, OW, pc.cout, dtype=torch.float32, device=x.device)
        else:
            pp = plane_pitch or OH * OW
            dt = torch.float32 if out_mode == OUT_PLANAR_F32 else act_dtype()
            out = torch.empty(B, pc.cout, pp, dtype=dt, device=x.device)   # the conv writes every valid pixel of every plane;
            if pp > OH * OW:                                                # only the padding behind them is cleared (V^T of the
                out[:, :, OH * OW:].zero_()                                 # two-tensor attention reads whole 32-key groups): the
                                                                            # full 900 MB fill of the DCN's offset planes cost 0.1 ms
    assert out.is_contiguous()
    if out_mode in (OUT_NHWC_BF16, OUT_NHWC_F32):
        d.out_pitch = out.shape[3]
    else:
        d.out_pitch = out.shape[1]
        d.plane_pitch = out.shape[2]
    d.out, d.Cout, d.out_off = out.data_ptr(), pc.cout, out_off
    # (the hi / lo epilogue and the GroupNorm prologue exist for the 128-wide tile only: those launches keep it -- a narrower tile came
    # back as GLARE_ERR_UNSUPPORTED from the single-pass hi / lo stream at the training crops)
    if (AUTO_COUT_TILE and pc.ksize == 3 and pc.cout > 64 and getattr(pc, "cout_tile", 0) == 0 and not getattr(pc, "subpixel", False)
            and getattr(pc, "_src", None) is not None and not hilo and gn_prologue is None):
        tile = conv_cout_tile(B, OH, OW, pc.cout)
        if tile == 32 and gn_stats:
            tile = 64          # the fused GroupNorm statistics need 4-row wave slabs (128- and 64-wide tiles)
        if tile:
            pc = _for_tile(pc, tile)
    d.weight_packed = pc.packed.data_ptr()
    d.cout_tile = getattr(pc, "cout_tile", 0)
    d.bias = pc.bias.data_ptr() if pc.bias is not None else None
    if residual is not None:
        assert residual.dtype == act_dtype() and residual.is_contiguous()
        d.residual, d.res_pitch, d.res_off = residual.data_ptr(), residual.shape[3], res_off
    out_lo = None
    if hilo:
        assert out_mode == OUT_NHWC_BF16 and out_off == 0 and out.shape[3] == pc.cout
        out_lo = getattr(out, "_lo", None)          # a caller-provided pair (out=...) keeps its lo half
        if out_lo is None:
            out_lo = torch.empty_like(out)
        d.out_lo = out_lo.data_ptr()
        rlo = getattr(residual, "_lo", None) if residual is not None else None
        if rlo is not None:
            assert rlo.shape == residual.shape and rlo.dtype == residual.dtype and rlo.is_contiguous()
            d.residual_lo = rlo.data_ptr()
    if gn_prologue is not None:      # (coef fp32 [B, Cin, 2], swish): GroupNorm of `x` applied by the loader (groupnorm_coeffs)
        coef, swish = gn_prologue
        assert coef.dtype == torch.float32 and coef.is_contiguous() and tuple(coef.shape) == (B, cin, 2)
        assert pc.ksize == 3 and stride == 1 and x2 is None and not upsample and not split and not hilo and in_off == 0
        d.gn_coef, d.gn_swish = coef.data_ptr(), int(bool(swish))
    subpixel = getattr(pc, "subpixel", False)
    assert not subpixel or upsample, "a sub-pixel packed filter only implements the upsample conv"
    d.ksize, d.stride, d.upsample, d.act, d.out_mode = pc.ksize, stride, (2 if subpixel else int(bool(upsample))), ACT[act], out_mode
    lib = _lib.lib()
    gn_part = None
    if gn_stats:  # GroupNorm statistics of the output, gathered by the epilogue (saves the consumer's read pass)
        assert out_mode == OUT_NHWC_BF16 and pc.cout % 128 == 0 and out_off == 0 and out.shape[3] == pc.cout
        lib.glare_conv2d_gn_partial_elems.restype = _ll
        lib.glare_conv2d_upsample_gn_partial_elems.restype = _ll
        if subpixel:
            n = lib.glare_conv2d_upsample_gn_partial_elems(_i(B), _i(H), _i(W), _i(pc.cout))
        else:
            n = lib.glare_conv2d_gn_partial_elems(_i(B), _i(OH), _i(OW), _i(pc.cout))
        gn_part = torch.empty(n, dtype=torch.float32, device=x.device)
        d.gn_partial = gn_part.data_ptr()
    if LAUNCH_EVENTS is not None and pc.ksize == 3:
        opix = B * OH * OW
        esz = 2 if out_mode in (OUT_NHWC_BF16, OUT_PLANAR_BF16) else 4
        # 2*B*Ho*Wo*Cin*Cout*k^2 (an fp32-class launch counts ONCE -- its 3 MFMA passes are this kernel's way of doing
        # the reference's fp32 conv, not more algorithmic work); bytes in + out + filter
        with _timed_launch("conv3x3_split" if split else "conv3x3", 2.0 * opix * 9 * pc.cin * pc.cout,
                           2.0 * B * H * W * pc.cin * (2 if split == 3 else 1) + esz * opix * pc.cout * (2 if hilo else 1)
                           + 2.0 * 9 * pc.cin * pc.cout * (split or 1)):
            check(lib.glare_conv2d_bf16(ctypes.byref(d), stream_handle()), "glare_conv2d_bf16")
    else:
        check(lib.glare_conv2d_bf16(ctypes.byref(d), stream_handle()), "glare_conv2d_bf16")
    if gn_stats:
        stats = torch.empty(B, 1, 32, 2, dtype=torch.float32, device=x.device)
        if subpixel:
            check(lib.glare_conv2d_upsample_gn_reduce(ptr(gn_part), ptr(stats), _i(B), _i(H), _i(W), _i(pc.cout), stream_handle()),
                  "glare_conv2d_upsample_gn_reduce")
        else:
            check(lib.glare_conv2d_gn_reduce(ptr(gn_part), ptr(stats), _i(B), _i(OH), _i(OW), _i(pc.cout), stream_handle()),
                  "glare_conv2d_gn_reduce")
        out._gn_stats = stats  # consumed by groupnorm(); plain Python attribute, not a tensor property
    if out_lo is not None:
        out._lo = out_lo
    return out


# ---- thin convs, GroupNorm, glue -------------------------------------------------------------------
_f = ctypes.c_float
_sz = ctypes.c_size_t


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def conv2d_smallcin(x, strides, shape_bhw, weight, bias=None, act="none", out=None, out_off=0, out_f32=False, hilo=False):
    """Direct conv for Cin <= 4.  x: fp32 device tensor read through explicit element strides
    (sb, sc, sy, sx); shape_bhw = (B, H, W).  Returns NHWC [B,H,W,Cout] bf16 (or fp32); hilo: a hi / lo pair (`._lo`)."""
    require_cuda(x, weight, bias, out)
    B, H, W = shape_bhw
    w = weight.detach().float().contiguous()
    cout, cin, k, _ = w.shape
    b = None if bias is None else bias.detach().float().contiguous()
    if hilo:
        assert out is None and not out_f32
        hi = torch.empty(B, H, W, cout, dtype=act_dtype(), device=x.device)
        lo = torch.empty_like(hi)
        sb, sc, sy, sx = strides
        check(_lib.lib().glare_conv2d_smallcin_hilo_f32(ptr(x), _ll(sb), _ll(sc), _ll(sy), _ll(sx), ptr(w), ptr(b), ptr(hi), ptr(lo),
                                                        _i(B), _i(H), _i(W), _i(cin), _i(cout), _i(k), _i(cout), _i(0), _i(ACT[act]),
                                                        stream_handle()), "glare_conv2d_smallcin_hilo_f32")
        hi._lo = lo
        return hi
    if out is None:
        out = torch.empty(B, H, W, cout, dtype=torch.float32 if out_f32 else act_dtype(), device=x.device)
    sb, sc, sy, sx = strides
    check(_lib.lib().glare_conv2d_smallcin_f32(ptr(x), _ll(sb), _ll(sc), _ll(sy), _ll(sx), ptr(w), ptr(b), ptr(out),
                                               _i(B), _i(H), _i(W), _i(cin), _i(cout), _i(k), _i(out.shape[3]),
                                               _i(out_off), _i(ACT[act]), _i(int(out.dtype == torch.float32)),
                                               stream_handle()), "glare_conv2d_smallcin_f32")
    return out


def groupnorm(x, gamma, beta, swish=True, eps=1e-6, cin=None, in_off=0, pair=False):
    """x: bf16 NHWC [B,H,W,pitch]; returns dense bf16 NHWC [B,H,W,C].  If the producing conv left its fused
    statistics on the tensor (conv2d(..., gn_stats=True)), only the apply pass runs."""
    require_cuda(x, gamma, beta)
    assert x.dtype == act_dtype() and x.is_contiguous()
    B, H, W, pitch = x.shape
    C = pitch - in_off if cin is None else cin
    lib = _lib.lib()
    stats = getattr(x, "_gn_stats", None)
    xlo = getattr(x, "_lo", None)
    if xlo is not None:      # a hi / lo pair (the conditional encoder's residual stream in fp16): normalise the 22-bit value
        assert in_off == 0 and C == pitch and xlo.shape == x.shape and xlo.is_contiguous()
        y = torch.empty(B, H, W, C, dtype=act_dtype(), device=x.device)
        ylo = torch.empty_like(y) if pair else None      # pair: the output as a hi / lo pair too (operand of an fp32-class conv)
        ws, nws = None, 0
        if stats is None:
            lib.glare_groupnorm_workspace_bytes.restype = _sz
            nws = lib.glare_groupnorm_workspace_bytes(_i(B), _ll(H * W))
            ws = _workspace(nws, x.device)
        check(lib.glare_groupnorm_hilo_pair_bf16(ptr(x), ptr(xlo), _i(pitch), _i(0), ptr(gamma), ptr(beta), ptr(y), ptr(ylo), _i(B),
                                                 _ll(H * W), _i(C), _f(eps), _i(int(swish)), ptr(stats),
                                                 _i(int(stats.shape[1]) if stats is not None else 0), ptr(ws), _sz(nws), stream_handle()),
              "glare_groupnorm_hilo_pair_bf16")
        if pair:
            y._lo = ylo
        return y
    assert not pair, "groupnorm(pair=True) normalises a hi / lo pair (x._lo)"
    if stats is not None and in_off == 0 and C == pitch:
        y = torch.empty(B, H, W, C, dtype=act_dtype(), device=x.device)
        check(lib.glare_groupnorm_apply_bf16(ptr(x), _i(pitch), _i(0), ptr(gamma), ptr(beta), ptr(y), _i(B), _ll(H * W), _i(C),
                                             _f(eps), _i(int(swish)), ptr(stats), _i(int(stats.shape[1])), stream_handle()),
              "glare_groupnorm_apply_bf16")
        return y
    lib.glare_groupnorm_workspace_bytes.restype = _sz
    nws = lib.glare_groupnorm_workspace_bytes(_i(B), _ll(H * W))
    ws = _workspace(nws, x.device)
    y = torch.empty(B, H, W, C, dtype=act_dtype(), device=x.device)
    check(lib.glare_groupnorm_swish_bf16(ptr(x), _i(pitch), _i(in_off), ptr(gamma), ptr(beta), ptr(y), _i(B), _ll(H * W),
                                         _i(C), _f(eps), _i(int(swish)), ptr(ws), _sz(ws.numel()), stream_handle()),
          "glare_groupnorm_swish_bf16")
    return y


def groupnorm_coeffs(x, gamma, beta, eps=1e-6):
    """The (a, d) pairs of GroupNorm(32) of x -- y = act(a x + d) per (image, channel) -- from the fused statistics the producing
    conv left on x (`x._gn_stats`): fp32 [B, C, 2] for conv2d(..., gn_prologue=(coef, swish)), which then normalises x inside its
    loader instead of reading a normalised copy (glare_conv_desc.gn_coef)."""
    require_cuda(x, gamma, beta)
    stats = getattr(x, "_gn_stats", None)
    assert stats is not None, "groupnorm_coeffs needs the producer's fused statistics on x"
    B, H, W, C = x.shape
    coef = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
    check(_lib.lib().glare_groupnorm_coeffs_f32(ptr(stats), _i(int(stats.shape[1])), _i(B), _ll(H * W), _i(C), ptr(gamma), ptr(beta),
                                                _f(eps), ptr(coef), stream_handle()), "glare_groupnorm_coeffs_f32")
    return coef


def split_hilo(x32):
    """fp32 [..., C] (numel % 8 == 0) -> the 16-bit tensor hi = round16(x) with its remainder lo = round16(x - hi) as `hi._lo`."""
    require_cuda(x32)
    assert x32.dtype == torch.float32 and x32.is_contiguous() and x32.numel() % 8 == 0
    hi = torch.empty(x32.shape, dtype=act_dtype(), device=x32.device)
    lo = torch.empty_like(hi)
    check(_lib.lib().glare_split_hilo_f32(ptr(x32), _ll(x32.numel()), ptr(hi), ptr(lo), stream_handle()), "glare_split_hilo_f32")
    hi._lo = lo
    return hi


def mix(a, b, w, out=None, out_off=0, a_off=0, b_off=0, C=None):
    require_cuda(a, b, out)
    assert a.dtype == b.dtype == act_dtype()
    C = a.shape[-1] - a_off if C is None else C
    npix = a.numel() // a.shape[-1]
    if out is None:
        out = torch.empty(a.shape[:-1] + (C,), dtype=act_dtype(), device=a.device)
    if torch.is_tensor(w):   # device scalar: read by the kernel, no .item() synchronisation
        require_cuda(w)
        w32 = w.detach().float().reshape(1)
        check(_lib.lib().glare_mix_dev_bf16(ptr(a), _i(a.shape[-1]), _i(a_off), ptr(b), _i(b.shape[-1]), _i(b_off), ptr(out),
                                            _i(out.shape[-1]), _i(out_off), _ll(npix), _i(C), ptr(w32), stream_handle()),
              "glare_mix_dev_bf16")
        return out
    check(_lib.lib().glare_mix_bf16(ptr(a), _i(a.shape[-1]), _i(a_off), ptr(b), _i(b.shape[-1]), _i(b_off), ptr(out),
                                    _i(out.shape[-1]), _i(out_off), _ll(npix), _i(C), _f(float(w)), stream_handle()),
          "glare_mix_bf16")
    return out


def mean_rescale(h, xw, whole_batch=False):
    """h bf16 [B,...], xw fp32 same shape -> h + xw * (mean(h)/mean(xw)) (bf16)."""
    require_cuda(h, xw)
    assert h.dtype == act_dtype() and xw.dtype == torch.float32 and h.shape == xw.shape
    B = h.shape[0]
    n = h.numel() // B
    lib = _lib.lib()
    lib.glare_mean_rescale_workspace_bytes.restype = _sz
    ws = _workspace(lib.glare_mean_rescale_workspace_bytes(_i(B), _ll(n)), h.device)
    out = torch.empty_like(h)
    check(lib.glare_mean_rescale_bf16(ptr(h), ptr(xw), ptr(out), _i(B), _ll(n), _i(int(whole_batch)), ptr(ws),
                                      _sz(ws.numel()), stream_handle()), "glare_mean_rescale_bf16")
    return out


def nchw_to_nhwc(x, bf16=True, out=None, out_off=0):
    require_cuda(x, out)
    x = x.float().contiguous()
    B, C, H, W = x.shape
    if out is None:
        out = torch.empty(B, H, W, C, dtype=act_dtype() if bf16 else torch.float32, device=x.device)
    check(_lib.lib().glare_nchw_to_nhwc(ptr(x), ptr(out), _i(B), _i(C), _ll(H * W), _i(out.shape[3]), _i(out_off),
                                        _i(int(out.dtype != torch.float32)), stream_handle()), "glare_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x, C=None, off=0):
    require_cuda(x)
    assert x.is_contiguous() and x.dtype in (torch.float32, act_dtype()), (x.dtype, act_dtype())
    B, H, W, pitch = x.shape
    C = pitch - off if C is None else C
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    check(_lib.lib().glare_nhwc_to_nchw(ptr(x), ptr(out), _i(B), _i(C), _ll(H * W), _i(pitch), _i(off),
                                        _i(int(x.dtype != torch.float32)), stream_handle()), "glare_nhwc_to_nchw")
    return out


# ---- flow ---------------------------------------------------------------------------------------
def flow_h1(z, ftA, ftA_off, wz, out=None):
    """z fp32 [B,H,W,3]; ftA fp32 [B,H,W,pitch]; wz fp32 [64,9] -> h1 bf16 [B,H,W,64]."""
    require_cuda(z, ftA, wz, out)
    B, H, W, _ = z.shape
    if out is None:
        out = torch.empty(B, H, W, 64, dtype=act_dtype(), device=z.device)
    olo = getattr(out, "_lo", None)
    if olo is not None:        # h1 as a hi / lo pair: the operand of the fp32-class 1x1 conv that follows
        check(_lib.lib().glare_flow_h1_pair_f32(ptr(z), ptr(ftA), _i(ftA.shape[3]), _i(ftA_off), ptr(wz), ptr(out), ptr(olo), _i(B), _i(H),
                                                _i(W), stream_handle()), "glare_flow_h1_pair_f32")
        return out
    check(_lib.lib().glare_flow_h1_f32(ptr(z), ptr(ftA), _i(ftA.shape[3]), _i(ftA_off), ptr(wz), ptr(out), _i(B), _i(H),
                                       _i(W), stream_handle()), "glare_flow_h1_f32")
    return out


def flow_tail(z, h4, hF, hF_off, M, t, eps=1e-4):
    """In-place update of z fp32 [B,H,W,3]; M (9 floats) and t (3 floats) are host sequences."""
    require_cuda(z, h4, hF)
    Ma = (ctypes.c_float * 9)(*[float(v) for v in M])
    ta = (ctypes.c_float * 3)(*[float(v) for v in t])
    check(_lib.lib().glare_flow_tail_f32(ptr(z), ptr(h4), ptr(hF), _i(hF.shape[3]), _i(hF_off), _ll(z.numel() // 3), Ma, ta,
                                         _f(eps), stream_handle()), "glare_flow_tail_f32")
    return z


def flow_fused_image(wz, w2, b2, w4, b4):
    """The filters of one coupling step as the fragment image of csrc/flow_fused.hip (fp32-class: every filter a hi / lo pair of
    16-bit values in the current precision).  wz fp32 [64, 9] (fAffine[0]'s z1 column, tap = 3 ky + kx), w2 [64, 64(,1,1)], b2 [64],
    w4 [4, 64, 3, 3], b4 [4] -- all with ActNorm / Conv2dZeros already folded (flow.Conv2d.folded()).
    An A fragment is [half h][row m][8]; the kernel contracts k-step (j, u), half h, element i against channel
    32 j + 16 u + 8 (i >> 2) + 4 h + (i & 3) -- the order in which a lane's accumulator registers of the PREVIOUS product hold its
    pixel's channels -- and the 9-tap conv of z0 against tap 8 h + i.  The 3x3 conv 64 -> 4 is packed as 36 rows (tap * 4 + cout)."""
    require_cuda(wz, w2, b2, w4, b4)
    dev = wz.device
    dt = act_dtype()
    ks = torch.arange(4, device=dev).view(4, 1, 1)
    h = torch.arange(2, device=dev).view(1, 2, 1)
    i = torch.arange(8, device=dev).view(1, 1, 8)
    cidx = (32 * (ks // 2) + 16 * (ks % 2) + 8 * (i // 4) + 4 * h + (i % 4)).reshape(-1)        # [ks, h, i]

    def frags(Wm):            # fp32 [64, 64] -> [jt, ks, h, m, i]
        return Wm.reshape(2, 32, 64)[:, :, cidx].reshape(2, 32, 4, 2, 8).permute(0, 2, 3, 1, 4).contiguous()

    def pair(x):
        hi = x.to(dt)
        return hi, (x - hi.float()).to(dt)

    wzp = torch.zeros(64, 16, dtype=torch.float32, device=dev)
    wzp[:, :9] = wz.detach().float().reshape(64, 9)
    wzf = wzp.reshape(2, 32, 2, 8).permute(0, 2, 1, 3).contiguous()                              # [jt, h, m, i], tap = 8 h + i
    w4r = torch.zeros(64, 64, dtype=torch.float32, device=dev)
    w4r[:36] = w4.detach().float().permute(2, 3, 0, 1).reshape(36, 64)                           # row = tap * 4 + cout
    r = torch.arange(16, device=dev).view(1, 1, 16)
    bidx = (32 * torch.arange(2, device=dev).view(2, 1, 1) + 8 * (r // 4) + 4 * torch.arange(2, device=dev).view(1, 2, 1) + (r % 4)).reshape(-1)
    parts = []
    for t in (wzf, frags(w2.detach().float().reshape(64, 64)), frags(w4r)):
        parts += [p.reshape(-1).view(torch.uint8) for p in pair(t)]
    parts.append(b2.detach().float()[bidx].contiguous().view(torch.uint8))
    parts.append(b4.detach().float().contiguous().view(torch.uint8))
    img = torch.cat(parts)
    lib = _lib.lib()
    lib.glare_flow_step_fused_image_bytes.restype = _ll
    assert img.numel() == lib.glare_flow_step_fused_image_bytes(), img.numel()
    return img


def flow_step_fused(z_in, z_out, ftA, ftA_off, image, hF, hF_off, M, t, eps=1e-4):
    """One reverse coupling step in one launch (csrc/flow_fused.hip): z_in -> z_out, fp32 [B,H,W,3], different buffers."""
    require_cuda(z_in, z_out, ftA, image, hF)
    assert z_in.data_ptr() != z_out.data_ptr() and z_in.dtype == z_out.dtype == ftA.dtype == hF.dtype == torch.float32
    B, H, W, _ = z_in.shape
    Ma = (ctypes.c_float * 9)(*[float(v) for v in M])
    ta = (ctypes.c_float * 3)(*[float(v) for v in t])
    check(_lib.lib().glare_flow_step_fused_bf16(ptr(z_in), ptr(z_out), ptr(ftA), _i(ftA.shape[3]), _i(ftA_off), ptr(image), ptr(hF),
                                                _i(hF.shape[3]), _i(hF_off), _i(B), _i(H), _i(W), Ma, ta, _f(eps), stream_handle()),
          "glare_flow_step_fused_bf16")
    return z_out


# ---- attention ------------------------------------------------------------------------------------
def attention_key_splits(B, N):
    """Workgroups = B * ceil(N/128) query blocks; below two full rounds of the 256 CUs the keys are split to fill the chip."""
    blocks = B * ((N + 127) // 128)
    if blocks >= 512:
        return 1
    return max(1, min(4, (512 + blocks - 1) // blocks, (N + 31) // 32))


# Test / tool hook: force the number of key splits of every attention launch (None = attention_key_splits()).  A B = 1 run
# splits the keys over 4 workgroups per query block to fill the chip, a B = 8 run does not: the two differ in summation order
# (rounding), so the batch-invariance test pins both to one configuration.
ATTENTION_KEY_SPLITS_OVERRIDE = None

# Measurement hook (bench.py): when this is a list, every attention launch is bracketed by a pair of events recorded on the
# stream the kernel is launched on, and (start, end, B, N) is appended.  None (the default) records nothing.
ATTENTION_LAUNCH_EVENTS = None


def attention_d512(q, k, v_t, N, ldq=None, ldk=None, out=None, key_splits=None, lse=None):
    """q, k: bf16 [B, N, ld] views (d=512 used); v_t: bf16 [B, 512, v_pitch]; returns bf16 [B, N, 512].  lse: optional fp32 [B, N]
    that receives log2 sum_j 2^(q_i . k_j) of every query row (what the fused backward needs)."""
    require_cuda(q, k, v_t, out, lse)
    if lse is not None:
        assert lse.dtype == torch.float32 and lse.is_contiguous() and lse.numel() == v_t.shape[0] * N
        return _attention_d512(q, k, v_t, N, ldq, ldk, out, key_splits, lse)
    if ATTENTION_LAUNCH_EVENTS is not None:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        try:
            return _attention_d512(q, k, v_t, N, ldq, ldk, out, key_splits)
        finally:
            end.record()
            ATTENTION_LAUNCH_EVENTS.append((start, end, int(v_t.shape[0]), int(N)))
    return _attention_d512(q, k, v_t, N, ldq, ldk, out, key_splits)


def _attention_d512(q, k, v_t, N, ldq, ldk, out, key_splits, lse=None):
    B = v_t.shape[0]
    count_flops("attn fwd", 4.0 * B * N * N * 512)
    ldq = q.shape[-1] if ldq is None else ldq
    ldk = k.shape[-1] if ldk is None else ldk
    if out is None:
        out = torch.empty(B, N, 512, dtype=act_dtype(), device=v_t.device)
    if key_splits is None:
        key_splits = ATTENTION_KEY_SPLITS_OVERRIDE
    ks = attention_key_splits(B, N) if key_splits is None else key_splits
    if lse is not None:
        lib = _lib.lib()
        lib.glare_attention_d512_splitk_workspace_bytes.restype = _sz
        nws = lib.glare_attention_d512_splitk_workspace_bytes(_i(B), _i(N), _i(ks))
        ws = _workspace(nws, v_t.device) if nws else None
        check(lib.glare_attention_d512_lse_bf16(ptr(q), _i(ldq), ptr(k), _i(ldk), ptr(v_t), _ll(v_t.shape[2]), ptr(out), _i(out.shape[-1]),
                                                ptr(lse), _i(B), _i(N), _i(ks), ptr(ws), _sz(nws), stream_handle()),
              "glare_attention_d512_lse_bf16")
        return out
    if ks > 1:
        lib = _lib.lib()
        lib.glare_attention_d512_splitk_workspace_bytes.restype = _sz
        nws = lib.glare_attention_d512_splitk_workspace_bytes(_i(B), _i(N), _i(ks))
        ws = _workspace(nws, v_t.device)
        check(lib.glare_attention_d512_splitk_bf16(ptr(q), _i(ldq), ptr(k), _i(ldk), ptr(v_t), _ll(v_t.shape[2]), ptr(out),
                                                   _i(out.shape[-1]), _i(B), _i(N), _i(ks), ptr(ws), _sz(nws), stream_handle()),
              "glare_attention_d512_splitk_bf16")
        return out
    check(_lib.lib().glare_attention_d512_bf16(ptr(q), _i(ldq), ptr(k), _i(ldk), ptr(v_t), _ll(v_t.shape[2]), ptr(out),
                                               _i(out.shape[-1]), _i(B), _i(N), stream_handle()),
          "glare_attention_d512_bf16")
    return out


def attention_kv512(q, kv, N, ldq=None, ldkv=None, out=None, key_splits=None, pair=False):
    """Attention with SHARED keys / values: out[b,i] = sum_j softmax_j(q_i . kv_j) kv_j.  q, kv: bf16 [B, N, ld] views
    (d = 512 used; the caller folds 512^-0.5 * log2(e) and the key projection into q -- see AttnBlock); returns bf16 [B, N, 512]."""
    require_cuda(q, kv, out)
    assert q.dtype == kv.dtype == act_dtype()
    B = kv.shape[0]
    count_flops("attn fwd", 4.0 * B * N * N * 512)
    ldq = q.shape[-1] if ldq is None else ldq
    ldkv = kv.shape[-1] if ldkv is None else ldkv
    if out is None:
        out = torch.empty(B, N, 512, dtype=act_dtype(), device=kv.device)
    if key_splits is None:
        key_splits = ATTENTION_KEY_SPLITS_OVERRIDE
    ks = attention_key_splits(B, N) if key_splits is None else key_splits
    lib = _lib.lib()
    ws, nws = None, 0
    if ks > 1:
        lib.glare_attention_d512_splitk_workspace_bytes.restype = _sz
        nws = lib.glare_attention_d512_splitk_workspace_bytes(_i(B), _i(N), _i(ks))
        ws = _workspace(nws, kv.device)
    ev = None
    if ATTENTION_LAUNCH_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if pair:      # the output as a hi / lo pair (`out._lo`): the operand of an fp32-class output projection
        out_lo = torch.empty_like(out)
        check(lib.glare_attention_kv512_pair_bf16(ptr(q), _i(ldq), ptr(kv), _i(ldkv), ptr(out), ptr(out_lo), _i(out.shape[-1]), _i(B), _i(N),
                                                  _i(ks), ptr(ws), _sz(nws), stream_handle()), "glare_attention_kv512_pair_bf16")
        out._lo = out_lo
    else:
        check(lib.glare_attention_kv512_bf16(ptr(q), _i(ldq), ptr(kv), _i(ldkv), ptr(out), _i(out.shape[-1]), _i(B), _i(N), _i(ks),
                                             ptr(ws), _sz(nws), stream_handle()), "glare_attention_kv512_bf16")
    if ev is not None:
        ev[1].record()
        ATTENTION_LAUNCH_EVENTS.append((ev[0], ev[1], int(B), int(N)))
    return out


def add_bf16(a, b, out=None, gn_stats=False):
    """out = a + b on bf16 NHWC tensors of one shape (the residual add left of AttnBlock's folded output projection); with
    gn_stats the GroupNorm statistics of the sum ride along (consumed by groupnorm(), which then only runs its apply pass)."""
    require_cuda(a, b, out)
    assert a.dtype == b.dtype == act_dtype() and a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    if out is None:
        out = torch.empty_like(a)
    lib = _lib.lib()
    if gn_stats:
        B, H, W, C = a.shape
        lib.glare_groupnorm_workspace_bytes.restype = _sz
        nb = lib.glare_groupnorm_workspace_bytes(_i(B), _ll(H * W))
        stats = torch.empty(B, nb // (B * 256), 32, 2, dtype=torch.float32, device=a.device)
        check(lib.glare_add_groupnorm_stats_bf16(ptr(a), ptr(b), ptr(out), _i(B), _ll(H * W), _i(C), ptr(stats), _sz(nb),
                                                 stream_handle()), "glare_add_groupnorm_stats_bf16")
        out._gn_stats = stats
        return out
    check(lib.glare_add_bf16(ptr(a), ptr(b), None, ptr(out), _ll(a.numel()), stream_handle()), "glare_add_bf16")
    return out


# ---- DCNv2 ------------------------------------------------------------------------------------------
def mdcn_forward(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1, groups=1, deformable_groups=1):
    """Reference layouts (NCHW fp32) -> NCHW fp32; the drop-in entry point glare_mdcn_forward_f32."""
    require_cuda(x, offset, mask, weight, bias)
    x, offset, mask, weight = [t.float().contiguous() for t in (x, offset, mask, weight)]
    bias = None if bias is None else bias.float().contiguous()
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    out = torch.empty(B, Co, Ho, Wo, dtype=torch.float32, device=x.device)
    lib = _lib.lib()
    lib.glare_mdcn_workspace_bytes.restype = _sz
    ws = _workspace(lib.glare_mdcn_workspace_bytes(_i(B), _i(C), _i(H), _i(W), _i(Co), _i(kh), _i(kw)), x.device)
    check(lib.glare_mdcn_forward_f32(ptr(x), ptr(offset), ptr(mask), ptr(weight), ptr(bias), ptr(out), _i(B), _i(C), _i(H),
                                     _i(W), _i(Co), _i(kh), _i(kw), _i(stride), _i(stride), _i(padding), _i(padding),
                                     _i(dilation), _i(dilation), _i(groups), _i(deformable_groups), ptr(ws),
                                     _sz(ws.numel()), stream_handle()), "glare_mdcn_forward_f32")
    return out


class PackedDcn:
    """single=False: the split form (two 16-bit halves per operand, fp32-class).  single=True: the filter rounded once to the
    current precision's 16-bit format for GLARE_MDCN_SINGLE_PASS (mdcn_forward_nhwc passes the flag by itself)."""

    def __init__(self, weight_oihw, bias, deformable_groups, single=False):
        require_cuda(weight_oihw)
        w = weight_oihw.detach().float().contiguous()
        self.co, self.c, self.kh, self.kw = w.shape
        self.dg = deformable_groups
        self.single = bool(single)
        if self.single:
            self.packed = torch.empty(w.numel(), dtype=act_dtype(), device=w.device)
            check(_lib.lib().glare_mdcn_pack_weight_single_f32(ptr(w), ptr(self.packed), _i(self.co), _i(self.c), _i(self.kh),
                                                                _i(self.kw), _i(self.dg), stream_handle()), "glare_mdcn_pack_weight_single_f32")
        else:
            self.packed = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
            check(_lib.lib().glare_mdcn_pack_weight_f32(ptr(w), ptr(self.packed), _i(self.co), _i(self.c), _i(self.kh),
                                                         _i(self.kw), _i(self.dg), stream_handle()), "glare_mdcn_pack_weight_f32")
        self.bias = None if bias is None else bias.detach().float().contiguous()


MDCN_GENERAL_KERNEL = 1   # glare_hip.h: pin the general-extent MFMA kernel for this call
MDCN_SINGLE_PASS = 2      # glare_hip.h: one MFMA per product on 16-bit operands (PackedDcn(single=True))


def mdcn_forward_nhwc(x, om, pd, x_off=0, C=None, mask_is_logit=True, padding=1, flags=0):
    """x: NHWC bf16/fp32 [B,H,W,pitch]; om: planar fp32 [B, 3*dg*K, plane] (offsets then mask logits,
    the conv_offset output); returns NHWC fp32 [B,H,W,Co]."""
    require_cuda(x, om)
    assert x.dtype in (torch.float32, act_dtype()), (x.dtype, act_dtype())
    B, H, W, pitch = x.shape
    C = pd.c if C is None else C
    K = pd.kh * pd.kw
    plane = om.shape[2]
    out = torch.empty(B, H, W, pd.co, dtype=torch.float32, device=x.device)
    mask = om[:, 2 * pd.dg * K:]
    if getattr(pd, "single", False):
        assert pd.packed.dtype == act_dtype() == x.dtype, "single-pass DCN: filter and x in the current precision's 16-bit format"
        flags |= MDCN_SINGLE_PASS
    with _timed_launch("dcn", 2.0 * B * H * W * C * pd.co * K + 8.0 * K * B * H * W * C,      # contraction + sampling
                       (2.0 if x.dtype != torch.float32 else 4.0) * B * H * W * C + 4.0 * B * H * W * (3 * pd.dg * K + pd.co) + 4.0 * pd.co * C * K):
        _mdcn_forward_nhwc_launch(x, om, pd, out, mask, pitch, x_off, plane, mask_is_logit, B, C, H, W, padding, flags)
    return out


def _mdcn_forward_nhwc_launch(x, om, pd, out, mask, pitch, x_off, plane, mask_is_logit, B, C, H, W, padding, flags):
    check(_lib.lib().glare_mdcn_forward_nhwc(ptr(x), _i(int(x.dtype != torch.float32)), _i(pitch), _i(x_off), ptr(om),
                                             _ll(plane), _ll(om.shape[1] * plane), ctypes.c_void_p(mask.data_ptr()),
                                             _ll(plane), _ll(om.shape[1] * plane), _i(int(mask_is_logit)),
                                             ptr(pd.packed), ptr(pd.bias), ptr(out), _i(0), _i(pd.co), _i(0), _ll(0), _i(B),
                                             _i(C), _i(H), _i(W), _i(pd.co), _i(pd.kh), _i(pd.kw), _i(1), _i(1), _i(padding),
                                             _i(padding), _i(1), _i(1), _i(1), _i(pd.dg), _i(flags), stream_handle()),
          "glare_mdcn_forward_nhwc")
    return out


def mdcn_forward_nhwc_fused(x, om, pd, x_off=0, C=None, mask_is_logit=True, padding=1, out16=True, want_sums=True):
    """The pipeline's DCN call (glare_mdcn_forward_nhwc_fused, round 6): the output as a 16-bit NHWC tensor (out16) and / or the per-tile
    sums of its fp32 values for the mean rescale that follows.  Returns (out, tile_sums or None, tile_pixels).  Raises GlareError with
    status ERR_UNSUPPORTED outside the lean kernel's shapes -- the caller then takes mdcn_forward_nhwc + mean_rescale."""
    require_cuda(x, om)
    assert x.dtype == act_dtype(), (x.dtype, act_dtype())
    B, H, W, pitch = x.shape
    C = pd.c if C is None else C
    K = pd.kh * pd.kw
    plane = om.shape[2]
    flags = 0
    if getattr(pd, "single", False):
        assert pd.packed.dtype == act_dtype()
        flags |= MDCN_SINGLE_PASS
    lib = _lib.lib()
    tile = int(lib.glare_mdcn_tile_pixels(_i(C), _i(pd.co), _i(pd.dg), _i(flags)))
    out = torch.empty(B, H, W, pd.co, dtype=act_dtype() if out16 else torch.float32, device=x.device)
    sums = torch.empty(B, (H * W + tile - 1) // tile, dtype=torch.float32, device=x.device) if want_sums else None   # tiles cut per image
    mask = om[:, 2 * pd.dg * K:]
    with _timed_launch("dcn", 2.0 * B * H * W * C * pd.co * K + 8.0 * K * B * H * W * C,      # contraction + sampling; bytes: x 16-bit, offsets + mask logits fp32, output as written
                       2.0 * B * H * W * C + 4.0 * B * H * W * 3 * pd.dg * K + (2.0 if out16 else 4.0) * B * H * W * pd.co + 4.0 * pd.co * C * K):
        check(lib.glare_mdcn_forward_nhwc_fused(ptr(x), _i(pitch), _i(x_off), ptr(om), _ll(plane), _ll(om.shape[1] * plane),
                                                ctypes.c_void_p(mask.data_ptr()), _ll(plane), _ll(om.shape[1] * plane), _i(int(mask_is_logit)),
                                                ptr(pd.packed), ptr(pd.bias), ptr(None if out16 else out), ptr(out if out16 else None), _i(pd.co),
                                                _i(0), ptr(sums), _i(B), _i(C), _i(H), _i(W), _i(pd.co), _i(pd.kh), _i(pd.kw), _i(1), _i(1),
                                                _i(padding), _i(padding), _i(1), _i(1), _i(1), _i(pd.dg), _i(flags), stream_handle()),
              "glare_mdcn_forward_nhwc_fused")
    return out, sums, tile


def mix_with_sums(a, b, w):
    """Mix.forward on dense tensors + the per-block sums of the rounded output (glare_mix_sum_bf16): (out, sums [B * blocks])."""
    require_cuda(a, b)
    assert a.dtype == b.dtype == act_dtype() and a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    B = a.shape[0]
    n = a.numel() // B
    lib = _lib.lib()
    bps = int(lib.glare_mix_sum_blocks(_ll(n)))
    out = torch.empty_like(a)
    sums = torch.empty(B * bps, dtype=torch.float32, device=a.device)
    wdev = w.detach().float().reshape(1) if torch.is_tensor(w) else None
    check(lib.glare_mix_sum_bf16(ptr(a), ptr(b), ptr(out), _i(B), _ll(n), _f(0.0 if wdev is not None else float(w)), ptr(wdev), ptr(sums),
                                 stream_handle()), "glare_mix_sum_bf16")
    return out, sums


def mean_rescale_fused(h, xw, h_sums, xw_tile_sums, tile_pixels, whole_batch=False):
    """h + xw * (mean(h) / mean(xw)) from sums gathered by the producers (mix_with_sums, mdcn_forward_nhwc_fused): one tiny ratio launch
    + the apply pass; xw fp32 or 16-bit."""
    require_cuda(h, xw, h_sums, xw_tile_sums)
    assert h.dtype == act_dtype() and xw.dtype in (torch.float32, act_dtype()) and h.shape == xw.shape and h.is_contiguous() and xw.is_contiguous()
    B = h.shape[0]
    n = h.numel() // B
    pix = n // h.shape[-1]
    out = torch.empty_like(h)
    ratio = torch.empty(B, dtype=torch.float32, device=h.device)
    check(_lib.lib().glare_mean_rescale_fused_bf16(ptr(h), ptr(xw), _i(int(xw.dtype != torch.float32)), ptr(out), _i(B), _ll(n), _ll(pix),
                                                   ptr(h_sums), ptr(xw_tile_sums), _i(tile_pixels), _i(int(whole_batch)), ptr(ratio),
                                                   stream_handle()), "glare_mean_rescale_fused_bf16")
    return out


def flow_blocks_per_sample(pixels_per_sample):
    return int(_lib.lib().glare_flow_blocks_per_sample(_ll(pixels_per_sample)))


def flow_fwd_pre(z, hF, hF_off, M, t, eps, partial_row, Mt_dev=None, out=None):
    """M, t: host sequences -- or Mt_dev: fp32 device tensor [12] (3x3 row-major, then the offset), no host round trip.
    out (with Mt_dev): write the result there instead of over z."""
    require_cuda(z, hF, partial_row, Mt_dev, out)
    B = z.shape[0]
    if Mt_dev is not None:
        assert Mt_dev.dtype == torch.float32 and Mt_dev.numel() == 12 and Mt_dev.is_contiguous()
        dst = z if out is None else out
        assert dst.is_contiguous() and z.is_contiguous() and dst.shape == z.shape
        check(_lib.lib().glare_flow_fwd_pre_dev_io_f32(ptr(z), ptr(dst), ptr(hF), _i(hF.shape[3]), _i(hF_off), _i(B),
                                                       _ll(z.numel() // 3 // B), ptr(Mt_dev), _f(eps), ptr(partial_row), stream_handle()),
              "glare_flow_fwd_pre_dev_io_f32")
        return
    assert out is None
    Ma = (ctypes.c_float * 9)(*[float(v) for v in M])
    ta = (ctypes.c_float * 3)(*[float(v) for v in t])
    check(_lib.lib().glare_flow_fwd_pre_f32(ptr(z), ptr(hF), _i(hF.shape[3]), _i(hF_off), _i(B), _ll(z.numel() // 3 // B), Ma, ta,
                                            _f(eps), ptr(partial_row), stream_handle()), "glare_flow_fwd_pre_f32")


def flow_fwd_post(z, h4, eps, partial_row, out=None):
    require_cuda(z, h4, partial_row, out)
    B = z.shape[0]
    dst = z if out is None else out
    assert dst.is_contiguous() and z.is_contiguous() and dst.shape == z.shape
    check(_lib.lib().glare_flow_fwd_post_io_f32(ptr(z), ptr(dst), ptr(h4), _i(B), _ll(z.numel() // 3 // B), _f(eps), ptr(partial_row),
                                                stream_handle()), "glare_flow_fwd_post_io_f32")


def flow_nll_reduce(z, mean, partial, n_rows):
    """-> float64 [B, 2]: (data-dependent logdet, Gaussian log-likelihood)."""
    require_cuda(z, mean, partial)
    B = z.shape[0]
    out = torch.empty(B, 2, dtype=torch.float64, device=z.device)
    check(_lib.lib().glare_flow_nll_reduce_f32(ptr(z), ptr(mean), ptr(partial), _i(n_rows), _i(B), _ll(z.numel() // 3 // B),
                                               ptr(out), stream_handle()), "glare_flow_nll_reduce_f32")
    return out
