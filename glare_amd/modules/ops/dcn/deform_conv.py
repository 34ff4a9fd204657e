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

    @staticmethod
    def deform_conv_forward(*args):
        raise NotImplementedError("DCN v1 is not on the GLARE path (SURVEY.md row f4)")

    deform_conv_backward_input = deform_conv_forward
    deform_conv_backward_parameters = deform_conv_forward


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
        out = self.conv_offset(x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        return modulated_deform_conv(x, offset, torch.sigmoid(mask), self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)
