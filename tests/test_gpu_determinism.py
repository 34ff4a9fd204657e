"""GPU: run-to-run determinism of the MFMA kernels at the BASELINE shapes.

No kernel on the inference path uses atomics, so two launches on the same input must agree bit for bit; a difference is a
hazard or a race that tolerance-based parity checks can miss (it affected a fraction of a percent of the pixels of an
experimental DCN kernel).  Size-independent property, full-size inputs (tools/determinism_check.py is
the longer version)."""
import pytest
import torch

from glare_amd import ops

pytestmark = pytest.mark.gpu
REPS = 3


def _same(fn):
    first = fn().clone()
    for _ in range(REPS):
        assert torch.equal(fn(), first)


@pytest.mark.parametrize("ci,co,h,w,k,ups", [(128, 128, 420, 620, 3, 0), (256, 256, 210, 310, 3, 0), (512, 1024, 105, 155, 1, 0),
                                             (256, 256, 210, 310, 3, 2), (128, 108, 420, 620, 3, 0)])
def test_conv_launches_are_bit_identical(ci, co, h, w, k, ups):
    g = torch.Generator().manual_seed(ci + h)
    x = torch.randn(4, h, w, ci, generator=g).to(torch.bfloat16).cuda()
    wt = (torch.randn(co, ci, k, k, generator=g) * 0.02).cuda()
    pc = ops.PackedConv(wt, torch.zeros(co).cuda(), upsample_subpixel=(ups == 2))
    _same(lambda: ops.conv2d(x, pc, upsample=bool(ups), gn_stats=(co % 128 == 0)))


def test_attention_launches_are_bit_identical():
    g = torch.Generator().manual_seed(1)
    N, C = 105 * 155, 512
    qk = (torch.randn(2, N, 2 * C, generator=g) * 0.3).to(torch.bfloat16).cuda()
    npad = (N + 63) // 64 * 64
    vt = torch.zeros(2, C, npad, dtype=torch.bfloat16, device="cuda")
    vt[:, :, :N] = torch.randn(2, C, N, generator=g).to(torch.bfloat16).cuda()
    _same(lambda: ops.attention_d512(qk, qk[..., C:], vt, N, ldq=2 * C, ldk=2 * C))
    _same(lambda: ops.attention_d512(qk[:1], qk[:1, :, C:], vt[:1], N, ldq=2 * C, ldk=2 * C))   # split-key path


@pytest.mark.parametrize("c,h,w", [(128, 420, 620), (256, 210, 310)])
def test_dcn_launches_are_bit_identical(c, h, w):
    g = torch.Generator().manual_seed(c)
    x = torch.randn(4, h, w, c, generator=g).to(torch.bfloat16).cuda()
    plane = (h * w + 63) // 64 * 64
    om = (torch.randn(4, 108, plane, generator=g) * 2.0).cuda()     # scattered samples, many out of the image
    pd = ops.PackedDcn((torch.randn(c, c, 3, 3, generator=g) * 0.02).cuda(), torch.zeros(c).cuda(), 4)
    _same(lambda: ops.mdcn_forward_nhwc(x, om, pd))
