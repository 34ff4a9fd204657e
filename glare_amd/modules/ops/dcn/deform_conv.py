"""Deformable convolution operator surface on the HIP DCNv2 kernel.

Mirrors code/models/modules/ops/dcn/deform_conv.py: `ModulatedDeformConvFunction` (:121-184),
`modulated_deform_conv` (:188), `ModulatedDeformConv` (:289-333), `ModulatedDeformConvPack` (:336-379)
and the native module object `deform_conv_ext` they call (pybind functions of
src/deform_conv_ext.cpp:150-164).  `deform_conv_ext` here is a shim with the SAME function names
and positional argument orders, forwarding to the C ABI of libglare_hip.so via ctypes -- so the
reference's own deform_conv.py runs unmodified on top of it.

Like the reference (deform_conv.py:143-144) CPU tensors raise NotImplementedError: there is no CPU path.
"""
import ctypes
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair, _single

from .... import _lib
from ...._lib import check, ptr, stream_handle

_i, _sz = ctypes.c_int, ctypes.c_size_t


class _DeformConvExt:
    """Drop-in for the pybind module `deform_conv_ext` (deform_conv_ext.cpp:150-164)."""

    @staticmethod
    def modulated_deform_conv_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w,
                                      stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group,
                                      with_bias):
        if not input.is_cuda:  # deform_conv_ext.cpp:124
            raise RuntimeError("modulated deform conv is not implemented on CPU")
        if not (input.is_contiguous() and weight.is_contiguous()):  # TORCH_CHECK, deform_conv_cuda.cpp:497-498
            raise RuntimeError("input and weight tensors have to be contiguous")
        if input.dtype != torch.float32:
            raise RuntimeError("glare_amd DCNv2 computes in fp32 (as the GLARE path does, deformableDecoder_arch.py:143)")
        B, C, H, W = input.shape
        Co, Ck, kh, kw = weight.shape
        if (kh, kw) != (kernel_h, kernel_w):  # deform_conv_cuda.cpp:511-513
            raise RuntimeError("Input shape and kernel shape won't match: (%d x %d vs %d x %d)." % (kernel_h, kernel_w, kh, kw))
        if C != Ck * group:  # deform_conv_cuda.cpp:514-516
            raise RuntimeError("Input shape and kernel channels won't match: (%d vs %d)." % (C, Ck * group))
        lib = _lib.lib()
        lib.glare_mdcn_workspace_bytes.restype = _sz
        nws = lib.glare_mdcn_workspace_bytes(_i(B), _i(C), _i(H), _i(W), _i(Co), _i(kh), _i(kw))
        ws = torch.empty(int(nws), dtype=torch.uint8, device=input.device)  # replaces the callee-owned `columns`
        offset, mask = offset.contiguous(), mask.contiguous()
        check(lib.glare_mdcn_forward_f32(ptr(input), ptr(offset), ptr(mask), ptr(weight), ptr(bias if with_bias else None),
                                         ptr(output), _i(B), _i(C), _i(H), _i(W), _i(Co), _i(kh), _i(kw), _i(stride_h),
                                         _i(stride_w), _i(pad_h), _i(pad_w), _i(dilation_h), _i(dilation_w), _i(group),
                                         _i(deformable_group), ptr(ws), _sz(ws.numel()), stream_handle()),
              "glare_mdcn_forward_f32")

    @staticmethod
    def modulated_deform_conv_backward(input, weight, bias, ones, offset, mask, columns, grad_input, grad_weight,
                                       grad_bias, grad_offset, grad_mask, grad_output, kernel_h, kernel_w, stride_h,
                                       stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
        if not input.is_cuda:
            raise RuntimeError("modulated deform conv is not implemented on CPU")
        if not (input.is_contiguous() and weight.is_contiguous()):
            raise RuntimeError("input and weight tensors have to be contiguous")
        B, C, H, W = input.shape
        Co, Ck, kh, kw = weight.shape
        if (kh, kw) != (kernel_h, kernel_w) or C != Ck * group:
            raise RuntimeError("Input shape and kernel shape won't match")
        lib = _lib.lib()
        lib.glare_mdcn_backward_workspace_bytes.restype = _sz
        nws = lib.glare_mdcn_backward_workspace_bytes(_i(B), _i(C), _i(H), _i(W), _i(Co), _i(kh), _i(kw))
        ws = torch.empty(int(nws), dtype=torch.uint8, device=input.device)
        offset, mask, grad_output = offset.contiguous(), mask.contiguous(), grad_output.contiguous()
        check(lib.glare_mdcn_backward_f32(ptr(input), ptr(offset), ptr(mask), ptr(weight), ptr(grad_output), ptr(grad_input),
                                          ptr(grad_offset), ptr(grad_mask), ptr(grad_weight),
                                          ptr(grad_bias if with_bias else None), _i(B), _i(C), _i(H), _i(W), _i(Co), _i(kh),
                                          _i(kw), _i(stride_h), _i(stride_w), _i(pad_h), _i(pad_w), _i(dilation_h),
                                          _i(dilation_w), _i(group), _i(deformable_group), ptr(ws), _sz(ws.numel()),
                                          stream_handle()), "glare_mdcn_backward_f32")

    # ---- DCN v1 (deform_conv_ext.cpp:52-104): the unmodulated operator is DCNv2 with mask == 1 and no
    # bias (same sampling rule, deform_conv_cuda_kernel.cu:190-236 vs :571-633), so the three v1 entry points run on the
    # v2 kernels.  NOTE the v1 argument order: W before H (deform_conv.py:67-69).
    @staticmethod
    def _v1_geometry(input, weight, offset, kW, kH, group, deformable_group):
        if not input.is_cuda:
            raise RuntimeError("deform conv is not implemented on CPU")  # deform_conv_ext.cpp:67
        if input.dim() != 4 or offset.dim() != 4:
            raise RuntimeError("4D input and offset tensors expected")
        if (weight.shape[2], weight.shape[3]) != (kH, kW) or input.shape[1] != weight.shape[1] * group:
            raise RuntimeError("invalid kernel / channel configuration")    # shape_check, deform_conv_cuda.cpp:64-150
        if offset.shape[1] != 2 * deformable_group * kH * kW:
            raise RuntimeError("invalid number of channels of offset")
        ones = torch.ones(offset.shape[0], deformable_group * kH * kW, offset.shape[2], offset.shape[3], dtype=torch.float32,
                          device=input.device)
        return ones

    @staticmethod
    def deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
                            deformable_group, im2col_step):
        mask = _DeformConvExt._v1_geometry(input, weight, offset, kW, kH, group, deformable_group)
        _DeformConvExt.modulated_deform_conv_forward(input.contiguous(), weight.contiguous(), None, ones, offset, mask, output, columns,
                                                     kH, kW, dH, dW, padH, padW, dilationH, dilationW, group, deformable_group, False)
        return 1   # deform_conv_cuda.cpp:255

    @staticmethod
    def deform_conv_backward_input(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH, padW, padH,
                                   dilationW, dilationH, group, deformable_group, im2col_step):
        mask = _DeformConvExt._v1_geometry(input, weight, offset, kW, kH, group, deformable_group)
        gw, gm = torch.zeros_like(weight), torch.zeros_like(mask)
        _DeformConvExt.modulated_deform_conv_backward(input.contiguous(), weight.contiguous(), None, None, offset, mask, columns,
                                                      gradInput, gw, None, gradOffset, gm, gradOutput, kH, kW, dH, dW, padH, padW,
                                                      dilationH, dilationW, group, deformable_group, False)
        return 1

    @staticmethod
    def deform_conv_backward_parameters(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW, padH, dilationW,
                                        dilationH, group, deformable_group, scale, im2col_step):
        mask = torch.ones(offset.shape[0], deformable_group * kH * kW, offset.shape[2], offset.shape[3], dtype=torch.float32,
                          device=input.device)
        if (gradWeight.shape[2], gradWeight.shape[3]) != (kH, kW):
            raise RuntimeError("invalid kernel configuration")
        gw, go, gm = torch.zeros_like(gradWeight), torch.zeros_like(offset), torch.zeros_like(mask)
        w0 = torch.zeros_like(gradWeight)   # grad_weight does not depend on the filter values
        _DeformConvExt.modulated_deform_conv_backward(input.contiguous(), w0, None, None, offset, mask, columns, None, gw, None, go,
                                                      gm, gradOutput, kH, kW, dH, dW, padH, padW, dilationH, dilationW, group,
                                                      deformable_group, False)
        gradWeight.add_(gw, alpha=float(scale))   # filter-sized accumulate (deform_conv_cuda.cpp:478-482)
        return 1


deform_conv_ext = _DeformConvExt()


class ModulatedDeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1):
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation
        ctx.groups, ctx.deformable_groups = groups, deformable_groups
        ctx.with_bias = bias is not None
        if not ctx.with_bias:
            bias = input.new_empty(1)  # fake tensor (deform_conv.py:140-142)
        if not input.is_cuda:
            raise NotImplementedError
        if weight.requires_grad or mask.requires_grad or offset.requires_grad or input.requires_grad:
            ctx.save_for_backward(input, offset, mask, weight, bias)
        n, co = input.size(0), weight.size(0)
        kh, kw = weight.shape[2:4]
        ho = (input.size(2) + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
        wo = (input.size(3) + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
        output = input.new_empty((n, co, ho, wo))
        ctx._bufs = [input.new_empty(0), input.new_empty(0)]
        deform_conv_ext.modulated_deform_conv_forward(input, weight, bias, ctx._bufs[0], offset, mask, output, ctx._bufs[1],
                                                      kh, kw, stride, stride, padding, padding, dilation, dilation, groups,
                                                      deformable_groups, ctx.with_bias)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, bias = ctx.saved_tensors
        grads = [torch.zeros_like(t) for t in (input, offset, mask, weight, bias)]
        deform_conv_ext.modulated_deform_conv_backward(input, weight, bias, ctx._bufs[0], offset, mask, ctx._bufs[1],
                                                       grads[0], grads[3], grads[4], grads[1], grads[2], grad_output,
                                                       weight.shape[2], weight.shape[3], ctx.stride, ctx.stride,
                                                       ctx.padding, ctx.padding, ctx.dilation, ctx.dilation, ctx.groups,
                                                       ctx.deformable_groups, ctx.with_bias)
        return (grads[0], grads[1], grads[2], grads[3], grads[4] if ctx.with_bias else None, None, None, None, None, None)


modulated_deform_conv = ModulatedDeformConvFunction.apply


class DeformConvFunction(Function):
    """DCN v1 (deform_conv.py:33-118): not on the GLARE path, provided so that the `deform_conv_ext` surface is complete."""

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got %dD tensor instead." % input.dim())
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.groups, ctx.deformable_groups, ctx.im2col_step = groups, deformable_groups, im2col_step
        ctx.save_for_backward(input, offset, weight)
        output = input.new_empty(DeformConvFunction._output_size(input, weight, ctx.padding, ctx.dilation, ctx.stride))
        ctx.bufs_ = [input.new_empty(0), input.new_empty(0)]
        if not input.is_cuda:
            raise NotImplementedError
        step = min(ctx.im2col_step, input.shape[0])
        assert input.shape[0] % step == 0, "im2col step must divide batchsize"
        deform_conv_ext.deform_conv_forward(input, weight, offset, output, ctx.bufs_[0], ctx.bufs_[1], weight.size(3), weight.size(2),
                                            ctx.stride[1], ctx.stride[0], ctx.padding[1], ctx.padding[0], ctx.dilation[1],
                                            ctx.dilation[0], ctx.groups, ctx.deformable_groups, step)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        grad_input = grad_offset = grad_weight = None
        if not grad_output.is_cuda:
            raise NotImplementedError
        step = min(ctx.im2col_step, input.shape[0])
        geo = (weight.size(3), weight.size(2), ctx.stride[1], ctx.stride[0], ctx.padding[1], ctx.padding[0], ctx.dilation[1],
               ctx.dilation[0], ctx.groups, ctx.deformable_groups)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input, grad_offset = torch.zeros_like(input), torch.zeros_like(offset)
            deform_conv_ext.deform_conv_backward_input(input, offset, grad_output, grad_input, grad_offset, weight, ctx.bufs_[0], *geo, step)
        if ctx.needs_input_grad[2]:
            grad_weight = torch.zeros_like(weight)
            deform_conv_ext.deform_conv_backward_parameters(input, offset, grad_output, grad_weight, ctx.bufs_[0], ctx.bufs_[1], *geo, 1, step)
        return (grad_input, grad_offset, grad_weight, None, None, None, None, None)

    @staticmethod
    def _output_size(input, weight, padding, dilation, stride):
        size = (input.size(0), weight.size(0))
        for d in range(input.dim() - 2):
            kernel = dilation[d] * (weight.size(d + 2) - 1) + 1
            size += ((input.size(d + 2) + 2 * padding[d] - kernel) // stride[d] + 1,)
        if not all(s > 0 for s in size):
            raise ValueError("convolution input is too small (output would be %s)" % "x".join(map(str, size)))
        return size


deform_conv = DeformConvFunction.apply


class DeformConv(nn.Module):  # deform_conv.py:191-245
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                 bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0 and out_channels % groups == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.transposed, self.output_padding = False, _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups, self.deformable_groups)


def _offset_conv(conv, x):
    """`conv_offset(x)` of the two Pack modules on the HIP kernels: the MFMA conv for the configuration every user on the GLARE
    path has (3x3 / 1x1, stride 1, 'same' padding, no dilation, Cin % 8 == 0); any other nn.Conv2d configuration as what a conv
    IS in this file's terms -- a modulated deformable conv with zero offsets and a unit mask (the general-shape DCN kernel takes
    every stride / padding / dilation the reference's extension takes).  NCHW fp32 in and out."""
    from .... import ops

    ops.require_cuda(x)
    k, st, pd, dl = conv.kernel_size, conv.stride, conv.padding, conv.dilation
    taped = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)   # the DCN form below carries the tape
    kc = 16 if k[0] == 3 else 32       # the implicit-GEMM kernel's channel stage: a K segment of the fp32-class form is whole stages
    if (not taped and k[0] == k[1] and k[0] in (1, 3) and st == (1, 1) and dl == (1, 1) and pd == (k[0] // 2, k[0] // 2) and conv.groups == 1
            and conv.in_channels % kc == 0):
        # the reference's conv_offset is an fp32 nn.Conv2d (deform_conv.py:357-364): the fp32-class form of the MFMA conv -- activation and
        # filter as hi / lo pairs, three K segments (glare_conv_desc.k_wrap) -- so that this op-level API returns the same offsets /
        # masks (to ~1e-6) with and without a tape (the single-pass 16-bit conv differed from the taped path by ~1e-2)
        xp = ops.split_hilo(ops.nchw_to_nhwc(x.float(), bf16=False))
        y = ops.conv2d(xp, ops.PackedConv(conv.weight, conv.bias, split=3), out_mode=ops.OUT_NHWC_F32)
        return ops.nhwc_to_nchw(y)
    if st[0] != st[1] or pd[0] != pd[1] or dl[0] != dl[1]:
        raise NotImplementedError("conv_offset with different vertical / horizontal stride, padding or dilation")
    B, _, H, W = x.shape
    Ho = (H + 2 * pd[0] - (dl[0] * (k[0] - 1) + 1)) // st[0] + 1
    Wo = (W + 2 * pd[1] - (dl[1] * (k[1] - 1) + 1)) // st[1] + 1
    K = k[0] * k[1]
    zero = torch.zeros(B, 2 * K, Ho, Wo, dtype=torch.float32, device=x.device)
    one = torch.ones(B, K, Ho, Wo, dtype=torch.float32, device=x.device)
    return modulated_deform_conv(x.float(), zero, one, conv.weight, conv.bias, st[0], pd[0], dl[0], conv.groups, 1)


class DeformConvPack(DeformConv):  # deform_conv.py:248-286
    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels, self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride), padding=_pair(self.padding),
                                     dilation=_pair(self.dilation), bias=True)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        return deform_conv(x, _offset_conv(self.conv_offset, x), self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)


class ModulatedDeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups, self.with_bias = groups, deformable_groups, bias
        self.transposed, self.output_padding = False, _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.init_weights()

    def init_weights(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                     self.groups, self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels,
                                     self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride), padding=_pair(self.padding),
                                     dilation=_pair(self.dilation), bias=True)
        self.init_weights()

    def init_weights(self):
        super().init_weights()
        if hasattr(self, "conv_offset"):
            self.conv_offset.weight.data.zero_()
            self.conv_offset.bias.data.zero_()

    def forward(self, x):
        out = _offset_conv(self.conv_offset, x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        return modulated_deform_conv(x, offset, torch.sigmoid(mask), self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)
