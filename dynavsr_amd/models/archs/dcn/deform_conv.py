"""Op-level drop-in for codes/models/archs/dcn (the only native extension of the reference).

``modulated_deform_conv(input, offset, mask, weight, bias, stride, padding, dilation, groups,
deformable_groups)`` has the call contract of ``ModulatedDeformConvFunction.apply``
(deform_conv.py:97-154): contiguous NCHW fp32 CUDA tensors, first-order backward returning
(grad_input, grad_offset, grad_mask, grad_weight, grad_bias), RuntimeError on unsupported
shapes, NotImplementedError for CPU tensors (:109-110).  It calls dvsr_mdcn_forward /
dvsr_mdcn_backward through ctypes instead of the pybind11 module ``deform_conv_cuda``.

``ModulatedDeformConvPack`` (:258-291) keeps the reference's constructor, parameters
(weight, bias, conv_offset_mask.{weight,bias}), zero offset initialisation and the
``extra_offset_mask`` list input; its conv_offset_mask runs on the HIP conv kernel as well.
The DCNv1 classes (DeformConv*) are not provided: EDVR never instantiates them.
"""
import logging
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from dynavsr_amd import hipops

logger = logging.getLogger('base')


class ModulatedDeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1):
        if not input.is_cuda:
            raise NotImplementedError
        ctx.cfg = (stride, padding, dilation, groups, deformable_groups)
        ctx.with_bias = bias is not None
        args = [t.contiguous() for t in (input, offset, mask, weight)]
        ctx.save_for_backward(*args)
        return hipops.mdcn_forward(*args, bias.contiguous() if ctx.with_bias else None, *ctx.cfg)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight = ctx.saved_tensors
        gx, goff, gmask, gw, gb = hipops.mdcn_backward(input, offset, mask, weight, grad_output.contiguous(),
                                                       *ctx.cfg, with_bias=ctx.with_bias)
        return gx, goff, gmask, gw, gb, None, None, None, None, None


modulated_deform_conv = ModulatedDeformConvFunction.apply


class _Conv2dHip(Function):
    """3x3/1x1 conv on the HIP kernels with autograd (used for conv_offset_mask)."""

    @staticmethod
    def forward(ctx, x, w, b, stride):
        x, w = x.contiguous(), w.contiguous()
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return hipops.conv2d_forward(x, w, b, stride=stride)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx, _, gw, gb = hipops.conv2d_backward(gy.contiguous(), x, w, stride=ctx.stride)
        return gx, gw, gb, None


class ModulatedDeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups, self.with_bias = groups, deformable_groups, bias
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1. / math.sqrt(self.in_channels * self.kernel_size[0] * self.kernel_size[1])
        self.weight.data.uniform_(-bound, bound)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)


class _OffsetMaskConv(nn.Module):
    def __init__(self, cin, cout, ks, stride, padding):
        super().__init__()
        if ks not in (1, 3) or padding != ks // 2:
            raise NotImplementedError('conv_offset_mask: kernel %d / padding %d not supported' % (ks, padding))
        self.stride = stride
        self.weight = nn.Parameter(torch.zeros(cout, cin, ks, ks))
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x):
        return _Conv2dHip.apply(x, self.weight, self.bias, self.stride)


class ModulatedDeformConvPack(ModulatedDeformConv):
    def __init__(self, *args, extra_offset_mask=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.extra_offset_mask = extra_offset_mask
        k = self.kernel_size
        self.conv_offset_mask = _OffsetMaskConv(self.in_channels, self.deformable_groups * 3 * k[0] * k[1], k[0],
                                                self.stride, self.padding)
        self.init_offset()

    def init_offset(self):
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def forward(self, x):
        if self.extra_offset_mask:
            x, feat = x[0], x[1]
        else:
            feat = x
        out = self.conv_offset_mask(feat)
        n_off = out.shape[1] // 3 * 2
        offset, mask = out[:, :n_off].contiguous(), torch.sigmoid(out[:, n_off:]).contiguous()
        offset_mean = torch.mean(torch.abs(offset))
        if offset_mean > 100:       # same host-visible warning as deform_conv.py:285-287
            logger.warning('Offset mean is {}, larger than 100.'.format(offset_mean))
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)
