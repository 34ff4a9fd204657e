"""CPU: the host side of the fp32-class convolution -- ops.split_filter builds the K segments the kernel contracts (glare_conv_desc.k_wrap,
include/glare_hip.h).  The kernel's stage -> (source, channel chunk) map is restated here for both K orders and the contraction checked
against the fp32 conv on the widened operands: the packing logic is covered without a GPU (the kernels against it: tests/test_gpu_split_conv.py)."""
import torch
import torch.nn.functional as F

from glare_amd import ops


def _stage_sources(cin, k_wrap, kc=16):
    """(source, first channel) of every 16-channel K stage, as conv_igemm_kernel's issue_a walks them."""
    n = cin // kc
    if k_wrap == 2:      # stages 2c, 2c + 1: chunk c of x_hi (staged once); then the x_lo segment
        return [("hi", (s // 2) * kc) for s in range(2 * n)] + [("lo", c * kc) for c in range(n)]
    return [("hi", c * kc) for c in range(n)] + [("lo", c * kc) for c in range(n)] + [("hi", c * kc) for c in range(n)]


def _contract(x_hi, x_lo, wseg, k_wrap):
    cin = x_hi.shape[1]
    out = 0
    for s, (src, c0) in enumerate(_stage_sources(cin, k_wrap)):
        xs = (x_hi if src == "hi" else x_lo)[:, c0:c0 + 16]
        out = out + F.conv2d(xs.double(), wseg[:, 16 * s:16 * s + 16].double(), None, 1, 1)
    return out


def test_split_filter_segments_match_the_kernels_stage_order():
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 48, 7, 9), generator=g)
    w = torch.randn((8, 48, 3, 3), generator=g) * 0.1
    dt = ops.act_dtype()
    x_hi = x.to(dt).float()
    x_lo = (x - x_hi).to(dt).float()
    w_hi = w.to(dt).float()
    w_lo = (w - w_hi).to(dt).float()
    want = (F.conv2d(x_hi.double(), w_hi.double(), None, 1, 1) + F.conv2d(x_lo.double(), w_hi.double(), None, 1, 1)
            + F.conv2d(x_hi.double(), w_lo.double(), None, 1, 1))
    for k_wrap, reuse in ((1, 0), (2, 16)):
        seg = ops.split_filter(w, 3, reuse_kc=reuse)
        assert seg.shape == (8, 3 * 48, 3, 3)
        # what the pack kernel makes of the segments: round16 of every entry (w_hi stays w_hi, the remainder becomes w_lo)
        got = _contract(x_hi, x_lo, seg.to(dt).float(), k_wrap)
        assert float((got - want).abs().max()) < 1e-12, k_wrap
    # and the fp32 conv itself is met to the pair's 2^-22
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    assert float((want - ref).abs().max() / ref.abs().max()) < 4e-6 if dt == torch.float16 else 4e-5


def test_split_filter_two_segment_form_and_batches():
    g = torch.Generator().manual_seed(6)
    w = torch.randn((3, 4, 32, 3, 3), generator=g)          # a batch of filters (packed_conv_batch)
    dt = ops.act_dtype()
    seg = ops.split_filter(w, 3, reuse_kc=16)
    assert seg.shape == (3, 4, 96, 3, 3)
    for i in range(3):
        assert torch.equal(seg[i], ops.split_filter(w[i], 3, reuse_kc=16))
    two = ops.split_filter(w[0], 2)
    hi = w[0].to(dt).float()
    assert torch.equal(two[:, :32], w[0]) and torch.equal(two[:, 32:], w[0] - hi)
    w0 = w[0]
    assert ops.split_filter(w0, 0) is w0
