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


def _params(x, weight, stride, padding, dilation, groups, deformable_groups, layout=_C.NCHW):
    return _C.DcnParams(
        B=x.shape[0], C=x.shape[1], H=x.shape[2], W=x.shape[3], Co=weight.shape[0], kh=weight.shape[2],
        kw=weight.shape[3], stride_h=stride[0], stride_w=stride[1], pad_h=padding[0], pad_w=padding[1],
        dil_h=dilation[0], dil_w=dilation[1], groups=groups, deformable_groups=deformable_groups,
        dtype=_C.dtype_code(x), layout=layout)


def _is_nhwc(x):
    """A 16-bit channels_last activation (a `model.to(memory_format=torch.channels_last)` caller): the kernels' native
    layout -- d2amd_dcn_params.layout = NHWC skips the transposes in and out."""
    return (x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16) and x.shape[1] > 1 and
            x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous())


def _stage_nhwc(x, weight, stride, padding, dilation, groups, deformable_groups):
    """16-bit NCHW activations -- what an unmodified reference model hands a DCN block (backbone/resnet.py:303-327): the
    column + dense-GEMM path (csrc/dcn_colpath.hip) is the channels_last one, so an eligible shape is staged channels_last
    here (the library's tiled transpose) and its results are handed back NCHW, as the caller's layout implies.  -> the
    channels_last twin, or None (fp32 / already channels_last / a shape the column path does not serve: the NCHW entry)."""
    if x.dim() != 4 or x.dtype not in (torch.float16, torch.bfloat16) or _is_nhwc(x) or x.numel() == 0:
        return None
    p = _params(x, weight, stride, padding, dilation, groups, deformable_groups, _C.NHWC)
    if not _C.lib().d2amd_deform_conv_column_path(ctypes.byref(p)):
        return None
    from ..modeling.poolers import _to_nhwc

    return _to_nhwc(x.detach().contiguous())


def _conv_out_extent(size, pad, dil, kernel, stride):
    """Output extent of one spatial axis of a (deformable) convolution."""
    return (size + 2 * pad - (dil * (kernel - 1) + 1)) // stride + 1


def _output_size(input, weight, padding, dilation, stride):
    """(N, Co, Ho, Wo) of the convolution; ValueError with the reference's message when an extent is not positive."""
    spatial = tuple(_conv_out_extent(input.size(ax + 2), padding[ax], dilation[ax], weight.size(ax + 2), stride[ax])
                    for ax in range(input.dim() - 2))
    size = (input.size(0), weight.size(0)) + spatial
    if min(size) <= 0:
        raise ValueError("convolution input is too small (output would be {})".format("x".join(str(v) for v in size)))
    return size


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


def _columns(L, p, device, wanted):
    """The column buffer the training forward keeps for the weight gradient (d2amd_deform_conv_columns_bytes: 0 when
    the shape / dtype is not served) -- the reference's `columns` scratch tensor (deform_conv.py:97-98,248-254) as a
    saved activation."""
    if not wanted:
        return None
    n = L.d2amd_deform_conv_columns_bytes(ctypes.byref(p))
    return torch.empty(n, dtype=torch.uint8, device=device) if n else None


def _dcn_forward(x, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups,
                 save_columns=False):
    """-> (out, columns | None).  save_columns: the weight gradient will be asked for."""
    _C.require_gpu(x, offset, mask, weight, bias, op="deform_conv")
    out_size = _output_size(x, weight, padding, dilation, stride)
    _check_shapes(x, offset, mask, weight, out_size, groups, deformable_groups)
    L = _C.lib()
    staged = _stage_nhwc(x, weight, stride, padding, dilation, groups, deformable_groups)
    if staged is not None or _is_nhwc(x):  # channels_last in, channels_last out (D2AMD_EUNSUPPORTED: -> NCHW below)
        x_ = staged if staged is not None else x.detach()
        offset_, mask_, weight_, bias_ = _same_dtype(x_, offset, mask, weight, bias)
        out = torch.empty(out_size, dtype=x_.dtype, device=x_.device, memory_format=torch.channels_last)
        p = _params(x_, weight_, stride, padding, dilation, groups, deformable_groups, _C.NHWC)
        with _C.on_device(x_.device):
            cols = _columns(L, p, x_.device, save_columns)
            ws_bytes = L.d2amd_deform_conv_workspace_bytes(ctypes.byref(p), 2 if cols is not None else 0)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x_.device)
            rc = L.d2amd_deform_conv_forward_columns(ctypes.byref(p), _C.ptr(x_), _C.ptr(offset_), _C.ptr(mask_),
                                                     _C.ptr(weight_), _C.ptr(bias_), _C.ptr(out), _C.ptr(cols),
                                                     _C.ptr(ws), ws_bytes, _C.stream())
        if rc == 0:
            if staged is not None:
                from ..modeling.poolers import _to_nchw

                out = _to_nchw(out)
            return out, cols
        if rc != _C.EUNSUPPORTED:
            _C.check(rc)
    x_ = x.detach().contiguous()
    offset_, mask_, weight_, bias_ = _same_dtype(x_, offset, mask, weight, bias)
    out = x_.new_empty(out_size)
    p = _params(x_, weight_, stride, padding, dilation, groups, deformable_groups)
    with _C.on_device(x_.device):
        cols = _columns(L, p, x_.device, save_columns)
        ws_bytes = L.d2amd_deform_conv_workspace_bytes(ctypes.byref(p), 2 if cols is not None else 0)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x_.device)
        _C.check(L.d2amd_deform_conv_forward_columns(ctypes.byref(p), _C.ptr(x_), _C.ptr(offset_), _C.ptr(mask_),
                                                     _C.ptr(weight_), _C.ptr(bias_), _C.ptr(out), _C.ptr(cols),
                                                     _C.ptr(ws), ws_bytes, _C.stream()))
    return out, cols


def _dcn_backward(x, offset, mask, weight, grad_output, stride, padding, dilation, groups, deformable_groups,
                  need_input, need_weight, with_bias, columns=None):
    _C.require_gpu(grad_output, op="deform_conv backward")
    L = _C.lib()
    staged = _stage_nhwc(x, weight, stride, padding, dilation, groups, deformable_groups)
    if staged is not None or _is_nhwc(x):  # channels_last activations: gradients in and out stay channels_last
        x_ = staged if staged is not None else x.detach()
        offset_, mask_, weight_ = _same_dtype(x_, offset, mask, weight)
        if staged is not None:
            from ..modeling.poolers import _to_nchw, _to_nhwc

            go = _to_nhwc(grad_output.detach().to(x_.dtype).contiguous())
        else:
            go = grad_output.detach().to(x_.dtype).contiguous(memory_format=torch.channels_last)
        gi = torch.empty_like(x_) if need_input else None  # (preserves channels_last)
        goff = torch.empty_like(offset_) if need_input else None
        gm = torch.empty_like(mask_) if (need_input and mask_ is not None) else None
        gw = torch.empty_like(weight_) if need_weight else None
        gb = x_.new_empty(weight_.shape[0]) if with_bias else None
        p = _params(x_, weight_, stride, padding, dilation, groups, deformable_groups, _C.NHWC)
        with _C.on_device(x_.device):
            ws_bytes = L.d2amd_deform_conv_workspace_bytes(ctypes.byref(p), 1)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x_.device)
            rc = L.d2amd_deform_conv_backward_columns(
                ctypes.byref(p), _C.ptr(x_), _C.ptr(offset_), _C.ptr(mask_), _C.ptr(weight_), _C.ptr(go),
                _C.ptr(columns), _C.ptr(gi), _C.ptr(goff), _C.ptr(gm), _C.ptr(gw), _C.ptr(gb), _C.ptr(ws), ws_bytes,
                _C.stream())
        if rc == 0:
            if staged is not None and gi is not None:
                gi = _to_nchw(gi)
            return gi, goff, gm, gw, gb
        if rc != _C.EUNSUPPORTED:
            _C.check(rc)
    x_ = x.detach().contiguous()
    offset_, mask_, weight_, go = _same_dtype(x_, offset, mask, weight, grad_output)
    gi = torch.empty_like(x_) if need_input else None
    goff = torch.empty_like(offset_) if need_input else None
    gm = torch.empty_like(mask_) if (need_input and mask_ is not None) else None
    gw = torch.empty_like(weight_) if need_weight else None
    gb = x_.new_empty(weight_.shape[0]) if with_bias else None
    p = _params(x_, weight_, stride, padding, dilation, groups, deformable_groups)
    with _C.on_device(x_.device):
        ws_bytes = L.d2amd_deform_conv_workspace_bytes(ctypes.byref(p), 1)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x_.device)
        _C.check(L.d2amd_deform_conv_backward_columns(
            ctypes.byref(p), _C.ptr(x_), _C.ptr(offset_), _C.ptr(mask_), _C.ptr(weight_), _C.ptr(go), _C.ptr(columns),
            _C.ptr(gi), _C.ptr(goff), _C.ptr(gm), _C.ptr(gw), _C.ptr(gb), _C.ptr(ws), ws_bytes, _C.stream()))
    return gi, goff, gm, gw, gb


class _DeformConv(Function):
    """autograd entry of DCNv1; positional arguments as the reference's `deform_conv` (deform_conv.py:16-141)."""

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64, save_columns=None):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        if not input.is_cuda:
            raise NotImplementedError("Deformable Conv is not supported on CPUs!")
        # im2col_step is accepted for API compatibility only: there is no column buffer, all images are batched
        ctx.geom = (_pair(stride), _pair(padding), _pair(dilation), groups, deformable_groups)
        ctx.stride, ctx.padding, ctx.dilation = ctx.geom[:3]
        ctx.groups, ctx.deformable_groups, ctx.im2col_step = groups, deformable_groups, im2col_step
        ctx.save_for_backward(input, offset, weight)
        # (grad mode is always off inside Function.forward: `deform_conv` below decides outside whether a backward can
        # follow; a direct .apply caller gets the conservative answer)
        if save_columns is None:
            save_columns = bool(ctx.needs_input_grad[2])
        out, ctx.columns = _dcn_forward(input, offset, None, weight, None, *ctx.geom, save_columns=bool(save_columns))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError("Deformable Conv is not supported on CPUs!")
        input, offset, weight = ctx.saved_tensors
        wants = ctx.needs_input_grad
        gi, goff, _, gw, _ = _dcn_backward(input, offset, None, weight, grad_output, *ctx.geom,
                                           wants[0] or wants[1], wants[2], False, columns=ctx.columns)
        ctx.columns = None
        return (gi, goff, gw) + (None,) * 7

    _output_size = staticmethod(_output_size)


class _ModulatedDeformConv(Function):
    """autograd entry of DCNv2; positional arguments as the reference's `modulated_deform_conv` (:187-309)."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1, save_columns=None):
        if not input.is_cuda:
            raise NotImplementedError("Deformable Conv is not supported on CPUs!")
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation  # scalars, as the reference keeps them
        ctx.groups, ctx.deformable_groups, ctx.with_bias = groups, deformable_groups, bias is not None
        ctx.geom = (_pair(stride), _pair(padding), _pair(dilation), groups, deformable_groups)
        if any(t.requires_grad for t in (input, offset, mask, weight)):
            ctx.save_for_backward(input, offset, mask, weight)
        if save_columns is None:  # (a direct .apply caller; `modulated_deform_conv` below passes the grad mode in)
            save_columns = bool(weight.requires_grad)
        out, ctx.columns = _dcn_forward(input, offset, mask, weight, bias, *ctx.geom, save_columns=bool(save_columns))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError("Deformable Conv is not supported on CPUs!")
        input, offset, mask, weight = ctx.saved_tensors
        grads = _dcn_backward(input, offset, mask, weight, grad_output, *ctx.geom, True, True, ctx.with_bias,
                              columns=ctx.columns)
        ctx.columns = None
        return tuple(grads) + (None,) * 6


def _keeps_columns(weight):
    """The training forward keeps its column for the weight gradient -- only when a backward can follow: under
    torch.no_grad() / inference mode nothing is kept, whatever `weight.requires_grad` says (an eval-mode model's
    Parameters still require grad)."""
    return bool(torch.is_grad_enabled() and weight.requires_grad)


def deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
    """The reference's `deform_conv = _DeformConv.apply` (deform_conv.py:312), same positional arguments."""
    return _DeformConv.apply(input, offset, weight, stride, padding, dilation, groups, deformable_groups, im2col_step,
                             _keeps_columns(weight))


def modulated_deform_conv(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                          deformable_groups=1):
    """The reference's `modulated_deform_conv = _ModulatedDeformConv.apply` (deform_conv.py:313)."""
    return _ModulatedDeformConv.apply(input, offset, mask, weight, bias, stride, padding, dilation, groups,
                                      deformable_groups, _keeps_columns(weight))


class _DeformConvModule(nn.Module):
    """What DeformConv and ModulatedDeformConv share: the nn.Conv2d-style weight (same name and shape as the
    reference's, so checkpoints load unchanged), the empty-batch shortcut, the optional norm / activation tail and
    the repr.  `_REPR` lists the attributes `extra_repr` prints, in the reference's order (the exact strings are
    pinned by tests/layers/test_deformable.py:157-171)."""

    _REPR = ("in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation", "groups",
             "deformable_groups")

    def _setup(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups,
               norm, activation):
        for what, c in (("in_channels", in_channels), ("out_channels", out_channels)):
            assert c % groups == 0, "{} {} cannot be divisible by groups {}".format(what, c, groups)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups = groups, deformable_groups
        self.norm, self.activation = norm, activation
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        nn.init.kaiming_uniform_(self.weight, nonlinearity="relu")

    def _empty_result(self, x):
        """Zero images in: the (0, Co, Ho, Wo) result without touching the device kernels."""
        hw = [_conv_out_extent(x.shape[2 + ax], _pair(self.padding)[ax], _pair(self.dilation)[ax],
                               self.kernel_size[ax], _pair(self.stride)[ax]) for ax in (0, 1)]
        return _NewEmptyTensorOp.apply(x, [x.shape[0], self.weight.shape[0]] + hw)

    def _tail(self, y):
        for f in (self.norm, self.activation):
            if f is not None:
                y = f(y)
        return y

    def extra_repr(self):
        fields = ["{}={}".format(k, getattr(self, k)) for k in self._REPR]
        return ", ".join(fields + ["bias=" + str(self.bias is not None)])


class DeformConv(_DeformConvModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False, norm=None, activation=None):
        """Deformable convolution (DCNv1).  Arguments as nn.Conv2d plus `deformable_groups`,
        `norm` (nn.Module) and `activation` (callable) -- reference deform_conv.py:317-365."""
        super().__init__()
        assert not bias
        self._setup(in_channels, out_channels, kernel_size, _pair(stride), _pair(padding), _pair(dilation), groups,
                    deformable_groups, norm, activation)
        self.bias = None

    def forward(self, x, offset):
        if x.numel() == 0:
            return self._empty_result(x)
        return self._tail(deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                                      self.deformable_groups))


class ModulatedDeformConv(_DeformConvModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True, norm=None, activation=None):
        """Modulated deformable convolution (DCNv2) -- reference deform_conv.py:415-460.
        stride / padding / dilation are scalars here, as in the reference."""
        super().__init__()
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, norm,
                    activation)
        self.with_bias = bias
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x, offset, mask):
        if x.numel() == 0:
            return self._empty_result(x)
        return self._tail(modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                                self.dilation, self.groups, self.deformable_groups))
