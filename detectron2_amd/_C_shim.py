"""A module with the shape of the reference's compiled extension `detectron2._C` (csrc/vision.cpp:81-113) on top of
libd2amd.so, for "level 2" integration: an UNMODIFIED `detectron2/layers/deform_conv.py` (which calls `_C.*` with
caller-allocated outputs and scratch tensors) and `utils/collect_env.py` keep working when this module is installed
under that name:

    import detectron2_amd._C_shim as shim; shim.install()      # sys.modules["detectron2._C"] = shim

Same names, positional order and return values as the C++ signatures (deformable/deform_conv.h:116-375):
  * v1 functions return int 1, v2 functions return None;
  * v1 argument order is (kW, kH, dW, dH, padW, padH, dilationW, dilationH), v2 is (kernel_h, kernel_w, stride_h,
    stride_w, pad_h, pad_w, dilation_h, dilation_w);
  * outputs are WRITTEN INTO the caller's tensors: `output`, `grad_offset`, `grad_mask` are overwritten; `grad_input`,
    `grad_weight`, `grad_bias` are accumulated into, as the reference's kernels do (col2im atomics / addmm_ into
    buffers the Python side zero-fills, deform_conv.py:97-98,121,250-254);
  * `columns` / `ones` are the reference's scratch tensors: accepted and left untouched (there is no column buffer);
    `im2col_step` is accepted and ignored (all images are batched).
The torch.ops.detectron2.* half of the native surface (vision.cpp:115-120) is registered by detectron2_amd.layers.ops.
COCOeval (vision.cpp:104-112) is an evaluator speed-up outside the hot path and is not provided."""
import sys

import torch

from . import _C
from .layers.deform_conv import _dcn_backward, _dcn_forward

__all__ = ["get_compiler_version", "get_cuda_version", "has_cuda", "deform_conv_forward", "deform_conv_backward_input",
           "deform_conv_backward_filter", "modulated_deform_conv_forward", "modulated_deform_conv_backward", "install"]


def get_compiler_version() -> str:
    """vision.cpp:49-79: the compiler the extension was built with."""
    return _C.lib().d2amd_compiler_version().decode()


def get_cuda_version() -> str:
    """vision.cpp:16-39: "HIP <major>.<minor>" on ROCm builds."""
    return _C.lib().d2amd_hip_version().decode()


def has_cuda() -> bool:
    """vision.cpp:41-47: False under WITH_HIP (collect_env.py prints the ROCm line from get_cuda_version instead)."""
    return False


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("Not compiled with GPU support" if t is ts[0] else "tensor is not on GPU!")


def _into(dst, src, accumulate=False):
    if dst is None or src is None:
        return
    if tuple(dst.shape) != tuple(src.shape):
        dst.resize_(src.shape)
        if accumulate:
            dst.zero_()
    if accumulate:
        dst.add_(src.to(dst.dtype))
    else:
        dst.copy_(src)


def deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW, dilationH,
                        group, deformable_group, im2col_step):
    _need_gpu(input, weight, offset)
    assert tuple(weight.shape[2:]) == (kH, kW), "kernel size does not match the weight"
    # (the reference's `columns` argument is per-image scratch its C++ resizes and overwrites, deform_conv_cuda.cu:346-353:
    # its contents are not part of the contract; the saved-column entry is d2amd_deform_conv_forward_columns)
    y, _cols = _dcn_forward(input, offset, None, weight, None, (dH, dW), (padH, padW), (dilationH, dilationW), group,
                            deformable_group)
    _into(output, y)
    return 1


def deform_conv_backward_input(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH, padW,
                               padH, dilationW, dilationH, group, deformable_group, im2col_step):
    _need_gpu(gradOutput, input, weight, offset)
    gi, goff, _, _, _ = _dcn_backward(input, offset, None, weight, gradOutput, (dH, dW), (padH, padW),
                                          (dilationH, dilationW), group, deformable_group, True, False, False)
    _into(gradInput, gi, accumulate=True)
    _into(gradOffset, goff)
    return 1


def deform_conv_backward_filter(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW, padH,
                                dilationW, dilationH, group, deformable_group, scale, im2col_step):
    _need_gpu(gradOutput, input, offset)
    w_like = gradWeight.new_zeros(gradWeight.shape)
    _, _, _, gw, _ = _dcn_backward(input, offset, None, w_like.to(input.dtype), gradOutput, (dH, dW), (padH, padW),
                                       (dilationH, dilationW), group, deformable_group, False, True, False)
    _into(gradWeight, gw * scale, accumulate=True)
    return 1


def modulated_deform_conv_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w, stride_h,
                                  stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
    _need_gpu(input, weight, offset)
    assert tuple(weight.shape[2:]) == (kernel_h, kernel_w), "kernel size does not match the weight"
    y, _cols = _dcn_forward(input, offset, mask, weight, bias if with_bias else None, (stride_h, stride_w), (pad_h, pad_w),
                            (dilation_h, dilation_w), group, deformable_group)
    _into(output, y)


def modulated_deform_conv_backward(input, weight, bias, ones, offset, mask, columns, grad_input, grad_weight, grad_bias,
                                   grad_offset, grad_mask, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h,
                                   pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
    _need_gpu(grad_output, input, weight, offset)
    gi, goff, gm, gw, gb = _dcn_backward(input, offset, mask, weight, grad_output, (stride_h, stride_w),
                                             (pad_h, pad_w), (dilation_h, dilation_w), group, deformable_group, True,
                                             True, bool(with_bias))
    _into(grad_input, gi, accumulate=True)
    _into(grad_offset, goff)
    _into(grad_mask, gm)
    _into(grad_weight, gw, accumulate=True)
    if with_bias:
        _into(grad_bias, gb, accumulate=True)


def install(name: str = "detectron2._C"):
    """Make `import detectron2._C` (or `from detectron2 import _C`) resolve to this module."""
    mod = sys.modules[__name__]
    sys.modules[name] = mod
    parent = sys.modules.get(name.rsplit(".", 1)[0])
    if parent is not None:
        setattr(parent, name.rsplit(".", 1)[1], mod)
    return mod
