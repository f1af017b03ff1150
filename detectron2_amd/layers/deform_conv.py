"""DeformConv / ModulatedDeformConv -- mirrors detectron2/layers/deform_conv.py:16-502 (same
module parameters and names, so model-zoo weights load unchanged; same functional aliases,
`extra_repr`, empty-input path and error behaviour).  The five `_C.*deform_conv*` entry points
of the reference (csrc/deformable/deform_conv.h:116-375) are replaced by two C-ABI calls,
d2amd_deform_conv_forward / _backward: implicit-GEMM MFMA kernels without a column buffer, all
images batched (no im2col_step loop; the argument is accepted and ignored)."""
import ctypes

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import _C
from .wrappers import _NewEmptyTensorOp


def _params(x, weight, stride, padding, dilation, groups, deformable_groups):
    return _C.DcnParams(
        B=x.shape[0], C=x.shape[1], H=x.shape[2], W=x.shape[3], Co=weight.shape[0], kh=weight.shape[2],
        kw=weight.shape[3], stride_h=stride[0], stride_w=stride[1], pad_h=padding[0], pad_w=padding[1],
        dil_h=dilation[0], dil_w=dilation[1], groups=groups, deformable_groups=deformable_groups,
        dtype=_C.dtype_code(x))


def _output_size(input, weight, padding, dilation, stride):
    channels = weight.size(0)
    output_size = (input.size(0), channels)
    for d in range(input.dim() - 2):
        in_size = input.size(d + 2)
        pad = padding[d]
        kernel = dilation[d] * (weight.size(d + 2) - 1) + 1
        stride_ = stride[d]
        output_size += ((in_size + (2 * pad) - kernel) // stride_ + 1,)
    if not all(map(lambda s: s > 0, output_size)):
        raise ValueError(
            "convolution input is too small (output would be {})".format("x".join(map(str, output_size))))
    return output_size


def _check_shapes(x, offset, mask, weight, out_size, groups, deformable_groups):
    # deform_conv_cuda.cu:140-270 (shape_check) -> RuntimeError like TORCH_CHECK
    k2 = weight.shape[2] * weight.shape[3]
    if weight.dim() != 4:
        raise RuntimeError("4D weight tensor (nOutputPlane,nInputPlane,kH,kW) expected")
    if x.shape[1] != weight.shape[1] * groups:
        raise RuntimeError(
            f"invalid number of input planes, expected: {weight.shape[1] * groups}, but got: {x.shape[1]}")
    if x.shape[1] % deformable_groups != 0:
        raise RuntimeError("input channels must divide deformable group size")
    if offset.shape[0] != x.shape[0]:
        raise RuntimeError("invalid batch size of offset")
    if tuple(offset.shape[2:]) != tuple(out_size[2:]):
        raise RuntimeError(
            f"invalid spatial size of offset, expected height: {out_size[2]} width: {out_size[3]}, but got "
            f"height: {offset.shape[2]} width: {offset.shape[3]}")
    if offset.shape[1] != deformable_groups * 2 * k2:
        raise RuntimeError("invalid number of channels of offset")
    if mask is not None:
        if tuple(mask.shape[2:]) != tuple(out_size[2:]):
            raise RuntimeError(
                f"invalid spatial size of mask, expected height: {out_size[2]} width: {out_size[3]}, but got "
                f"height: {mask.shape[2]} width: {mask.shape[3]}")
        if mask.shape[1] != deformable_groups * k2 or mask.shape[0] != x.shape[0]:
            raise RuntimeError("invalid number of channels of mask")


def _same_dtype(ref, *ts):
    return [None if t is None else t.detach().to(ref.dtype).contiguous() for t in ts]


def _dcn_forward(x, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups):
    _C.require_gpu(x, offset, mask, weight, bias, op="deform_conv")
    out_size = _output_size(x, weight, padding, dilation, stride)
    _check_shapes(x, offset, mask, weight, out_size, groups, deformable_groups)
    x_ = x.detach().contiguous()
    offset_, mask_, weight_, bias_ = _same_dtype(x_, offset, mask, weight, bias)
    out = x_.new_empty(out_size)
    p = _params(x_, weight_, stride, padding, dilation, groups, deformable_groups)
    L = _C.lib()
    with _C.on_device(x_.device):
        ws_bytes = L.d2amd_deform_conv_workspace_bytes(ctypes.byref(p), 0)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x_.device)
        _C.check(L.d2amd_deform_conv_forward(ctypes.byref(p), _C.ptr(x_), _C.ptr(offset_), _C.ptr(mask_),
                                             _C.ptr(weight_), _C.ptr(bias_), _C.ptr(out), _C.ptr(ws), ws_bytes,
                                             _C.stream()))
    return out


def _dcn_backward(x, offset, mask, weight, grad_output, stride, padding, dilation, groups, deformable_groups,
                  need_input, need_weight, with_bias):
    _C.require_gpu(grad_output, op="deform_conv backward")
    x_ = x.detach().contiguous()
    offset_, mask_, weight_, go = _same_dtype(x_, offset, mask, weight, grad_output)
    gi = torch.empty_like(x_) if need_input else None
    goff = torch.empty_like(offset_) if need_input else None
    gm = torch.empty_like(mask_) if (need_input and mask_ is not None) else None
    gw = torch.empty_like(weight_) if need_weight else None
    gb = x_.new_empty(weight_.shape[0]) if with_bias else None
    p = _params(x_, weight_, stride, padding, dilation, groups, deformable_groups)
    L = _C.lib()
    with _C.on_device(x_.device):
        ws_bytes = L.d2amd_deform_conv_workspace_bytes(ctypes.byref(p), 1)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x_.device)
        _C.check(L.d2amd_deform_conv_backward(
            ctypes.byref(p), _C.ptr(x_), _C.ptr(offset_), _C.ptr(mask_), _C.ptr(weight_), _C.ptr(go), _C.ptr(gi),
            _C.ptr(goff), _C.ptr(gm), _C.ptr(gw), _C.ptr(gb), _C.ptr(ws), ws_bytes, _C.stream()))
    return gi, goff, gm, gw, gb


class _DeformConv(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        ctx.stride = _pair(stride)
        ctx.padding = _pair(padding)
        ctx.dilation = _pair(dilation)
        ctx.groups = groups
        ctx.deformable_groups = deformable_groups
        ctx.im2col_step = im2col_step  # accepted for API compatibility; all images are batched
        ctx.save_for_backward(input, offset, weight)
        if not input.is_cuda:
            raise NotImplementedError("Deformable Conv is not supported on CPUs!")
        return _dcn_forward(input, offset, None, weight, None, ctx.stride, ctx.padding, ctx.dilation, groups,
                            deformable_groups)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        if not grad_output.is_cuda:
            raise NotImplementedError("Deformable Conv is not supported on CPUs!")
        need_input = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_weight = ctx.needs_input_grad[2]
        gi, goff, _, gw, _ = _dcn_backward(input, offset, None, weight, grad_output, ctx.stride, ctx.padding,
                                           ctx.dilation, ctx.groups, ctx.deformable_groups, need_input, need_weight,
                                           False)
        return gi, goff, gw, None, None, None, None, None, None

    _output_size = staticmethod(_output_size)


class _ModulatedDeformConv(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1):
        ctx.stride = stride
        ctx.padding = padding
        ctx.dilation = dilation
        ctx.groups = groups
        ctx.deformable_groups = deformable_groups
        ctx.with_bias = bias is not None
        if not input.is_cuda:
            raise NotImplementedError("Deformable Conv is not supported on CPUs!")
        if weight.requires_grad or mask.requires_grad or offset.requires_grad or input.requires_grad:
            ctx.save_for_backward(input, offset, mask, weight)
        return _dcn_forward(input, offset, mask, weight, bias, _pair(stride), _pair(padding), _pair(dilation),
                            groups, deformable_groups)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError("Deformable Conv is not supported on CPUs!")
        input, offset, mask, weight = ctx.saved_tensors
        gi, goff, gm, gw, gb = _dcn_backward(input, offset, mask, weight, grad_output, _pair(ctx.stride),
                                             _pair(ctx.padding), _pair(ctx.dilation), ctx.groups,
                                             ctx.deformable_groups, True, True, ctx.with_bias)
        return gi, goff, gm, gw, gb, None, None, None, None, None

    @staticmethod
    def _infer_shape(ctx, input, weight):
        n = input.size(0)
        channels_out = weight.size(0)
        height, width = input.shape[2:4]
        kernel_h, kernel_w = weight.shape[2:4]
        height_out = (height + 2 * ctx.padding - (ctx.dilation * (kernel_h - 1) + 1)) // ctx.stride + 1
        width_out = (width + 2 * ctx.padding - (ctx.dilation * (kernel_w - 1) + 1)) // ctx.stride + 1
        return n, channels_out, height_out, width_out


deform_conv = _DeformConv.apply
modulated_deform_conv = _ModulatedDeformConv.apply


class DeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False, norm=None, activation=None):
        """Deformable convolution (DCNv1).  Arguments as nn.Conv2d plus `deformable_groups`,
        `norm` (nn.Module) and `activation` (callable) -- reference deform_conv.py:317-365."""
        super(DeformConv, self).__init__()
        assert not bias
        assert in_channels % groups == 0, "in_channels {} cannot be divisible by groups {}".format(
            in_channels, groups)
        assert out_channels % groups == 0, "out_channels {} cannot be divisible by groups {}".format(
            out_channels, groups)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.norm = norm
        self.activation = activation
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // self.groups, *self.kernel_size))
        self.bias = None
        nn.init.kaiming_uniform_(self.weight, nonlinearity="relu")

    def forward(self, x, offset):
        if x.numel() == 0:
            output_shape = [
                (i + 2 * p - (di * (k - 1) + 1)) // s + 1
                for i, p, di, k, s in zip(x.shape[-2:], self.padding, self.dilation, self.kernel_size, self.stride)
            ]
            output_shape = [x.shape[0], self.weight.shape[0]] + output_shape
            return _NewEmptyTensorOp.apply(x, output_shape)
        x = deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                        self.deformable_groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x

    def extra_repr(self):
        tmpstr = "in_channels=" + str(self.in_channels)
        tmpstr += ", out_channels=" + str(self.out_channels)
        tmpstr += ", kernel_size=" + str(self.kernel_size)
        tmpstr += ", stride=" + str(self.stride)
        tmpstr += ", padding=" + str(self.padding)
        tmpstr += ", dilation=" + str(self.dilation)
        tmpstr += ", groups=" + str(self.groups)
        tmpstr += ", deformable_groups=" + str(self.deformable_groups)
        tmpstr += ", bias=False"
        return tmpstr


class ModulatedDeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True, norm=None, activation=None):
        """Modulated deformable convolution (DCNv2) -- reference deform_conv.py:415-460.
        stride / padding / dilation are scalars here, as in the reference."""
        super(ModulatedDeformConv, self).__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.with_bias = bias
        self.norm = norm
        self.activation = activation
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.bias = None
        nn.init.kaiming_uniform_(self.weight, nonlinearity="relu")
        if self.bias is not None:
            nn.init.constant_(self.bias, 0)

    def forward(self, x, offset, mask):
        if x.numel() == 0:
            output_shape = [
                (i + 2 * p - (di * (k - 1) + 1)) // s + 1
                for i, p, di, k, s in zip(x.shape[-2:], _pair(self.padding), _pair(self.dilation), self.kernel_size,
                                          _pair(self.stride))
            ]
            output_shape = [x.shape[0], self.weight.shape[0]] + output_shape
            return _NewEmptyTensorOp.apply(x, output_shape)
        x = modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                  self.groups, self.deformable_groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x

    def extra_repr(self):
        tmpstr = "in_channels=" + str(self.in_channels)
        tmpstr += ", out_channels=" + str(self.out_channels)
        tmpstr += ", kernel_size=" + str(self.kernel_size)
        tmpstr += ", stride=" + str(self.stride)
        tmpstr += ", padding=" + str(self.padding)
        tmpstr += ", dilation=" + str(self.dilation)
        tmpstr += ", groups=" + str(self.groups)
        tmpstr += ", deformable_groups=" + str(self.deformable_groups)
        tmpstr += ", bias=" + str(self.with_bias)
        return tmpstr
