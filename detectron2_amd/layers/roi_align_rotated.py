"""Rotated ROIAlign: operator surface of detectron2/layers/roi_align_rotated.py:11-103 (`roi_align_rotated`,
`ROIAlignRotated`) on top of the HIP kernels.  The native calls go through
torch.ops.detectron2.roi_align_rotated_{forward,backward} (registered by ops.py from the C ABI), so a scripted or
traced module resolves to the same op names as the reference (:88-91)."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import ops  # noqa: F401  (registers torch.ops.detectron2.*)
from .wrappers import disable_torch_compiler

_fwd_op = lambda: torch.ops.detectron2.roi_align_rotated_forward  # resolved at call time (ops registered above)
_bwd_op = lambda: torch.ops.detectron2.roi_align_rotated_backward


class _ROIAlignRotated(Function):
    """Autograd pair around the two native ops; gradient flows to the feature map only (6 inputs -> 6 grads)."""

    @staticmethod
    @disable_torch_compiler
    def forward(ctx, features, rois, output_size, spatial_scale, sampling_ratio):
        pooled_h, pooled_w = _pair(output_size)
        ctx.save_for_backward(rois)
        ctx.geom = (float(spatial_scale), pooled_h, pooled_w, int(sampling_ratio)) + tuple(features.shape)
        return _fwd_op()(features, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_pooled):
        scale, pooled_h, pooled_w, sampling, n, c, h, w = ctx.geom
        d_features = _bwd_op()(grad_pooled, ctx.saved_tensors[0], scale, pooled_h, pooled_w, n, c, h, w, sampling)
        return (d_features,) + (None,) * 5


roi_align_rotated = _ROIAlignRotated.apply


class ROIAlignRotated(nn.Module):
    """`ROIAlignRotated(output_size, spatial_scale, sampling_ratio)`: pools (B, 6) rotated boxes
    [batch index, x_ctr, y_ctr, width, height, angle in degrees] from an NCHW map.  Pixel model: continuous
    coordinates, i.e. always "aligned" (reference :60-66).  sampling_ratio 0 = adaptive (ceil(roi / pooled))."""

    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input, rois):
        assert rois.dim() == 2 and rois.size(1) == 6
        result_dtype = input.dtype
        if result_dtype == torch.float16:  # the reference pools fp16 in fp32 and casts back (:80-83)
            input, rois = input.float(), rois.float()
        if torch.jit.is_scripting() or torch.jit.is_tracing():  # no autograd.Function inside a graph
            ph, pw = _pair(self.output_size)
            pooled = torch.ops.detectron2.roi_align_rotated_forward(input, rois, self.spatial_scale, ph, pw,
                                                                    self.sampling_ratio)
        else:
            pooled = roi_align_rotated(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)
        return pooled.to(dtype=result_dtype)

    def __repr__(self):
        return (f"{type(self).__name__}(output_size={self.output_size}, spatial_scale={self.spatial_scale}, "
                f"sampling_ratio={self.sampling_ratio})")
