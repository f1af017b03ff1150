"""ROIAlign -- mirrors detectron2/layers/roi_align.py:7-74 (torchvision.ops.roi_align semantics).

Differences from the reference, by design (SURVEY 7 "bf16"): ROI coordinates are always kept in
fp32 (the reference casts them to the input dtype, which under AMP torchvision undoes again);
16-bit features are sampled with fp32 weights and fp32 accumulation.  channels_last inputs are
consumed and produced natively (NHWC kernels)."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import _C
from .wrappers import disable_torch_compiler


def _layout_of(x):
    """NHWC when the tensor is channels_last-dense (and not also NCHW-dense)."""
    if x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last):
        return _C.NHWC
    return _C.NCHW


def _prep_input(x):
    layout = _layout_of(x)
    if layout == _C.NCHW:
        x = x.contiguous()
    return x, layout


def _empty_like_layout(x, shape, layout):
    if layout == _C.NHWC:
        return torch.empty(shape, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    return torch.empty(shape, dtype=x.dtype, device=x.device)


def f64_forward(input, rois, ph, pw, spatial_scale, sampling_ratio, aligned, rotated):
    """Double-precision entry (d2amd_roi_align_f64_forward): everything double, ROIs cast to the input dtype as the
    reference does (roi_align.py:60), NCHW.  The gradcheck path of the reference's tests, not a tuned one."""
    x = input.detach().contiguous()
    r = rois.detach().to(torch.float64).contiguous()
    n, c, h, w = x.shape
    out = torch.empty((r.shape[0], c, ph, pw), dtype=torch.float64, device=x.device)
    if out.numel() == 0:
        return out
    status = torch.zeros(1, dtype=torch.int32, device=x.device) if rotated else None
    with _C.on_device(x.device):
        _C.check(_C.lib().d2amd_roi_align_f64_forward(
            _C.ptr(x), _C.ptr(r), _C.ptr(out), n, c, h, w, r.shape[0], ph, pw, float(spatial_scale),
            int(sampling_ratio), int(bool(aligned)), int(bool(rotated)), _C.ptr(status), _C.stream()))
    if rotated and int(status.item()) != 0:  # ROIAlignRotated_cpu.cpp:236-238
        raise RuntimeError("ROIs in ROIAlignRotated do not have non-negative size!")
    return out


def f64_backward(grad, rois, shape, ph, pw, spatial_scale, sampling_ratio, aligned, rotated):
    g = grad.detach().to(torch.float64).contiguous()
    r = rois.detach().to(torch.float64).contiguous()
    n, c, h, w = shape
    gin = torch.empty(shape, dtype=torch.float64, device=g.device)
    if gin.numel() == 0:
        return gin
    with _C.on_device(g.device):
        _C.check(_C.lib().d2amd_roi_align_f64_backward(
            _C.ptr(g), _C.ptr(r), _C.ptr(gin), n, c, h, w, r.shape[0], ph, pw, float(spatial_scale),
            int(sampling_ratio), int(bool(aligned)), int(bool(rotated)), _C.stream()))
    return gin


class _ROIAlign(Function):
    @staticmethod
    @disable_torch_compiler
    def forward(ctx, input, rois, output_size, spatial_scale, sampling_ratio, aligned):
        _C.require_gpu(input, rois, op="roi_align")
        ph, pw = _pair(output_size)
        if input.dtype == torch.float64:
            ctx.save_for_backward(rois.detach())
            ctx.cfg = (ph, pw, float(spatial_scale), int(sampling_ratio), bool(aligned), tuple(input.shape), "f64")
            return f64_forward(input, rois, ph, pw, spatial_scale, sampling_ratio, aligned, False)
        x, layout = _prep_input(input)
        staged = False
        if layout == _C.NCHW and x.dtype in (torch.float16, torch.bfloat16) and x.shape[1] % 8 == 0 and x.numel():
            # 16-bit NCHW features -- what an unmodified reference model hands a level's ROIAlign (poolers.py:249-262): the
            # NHWC kernel reads a tap as contiguous channels, the NCHW kernel cannot and is 5 x slower (profiles/r01).  One
            # channels_last staging copy per feature tensor, SHARED by the poolers of an iteration (the box head and the
            # mask head pool the same FPN level: modeling/poolers.py keeps the copy while the tensor is unmodified), and
            # an NCHW result, as the caller's layout implies.
            from ..modeling.poolers import _staged_nhwc

            x, layout, staged = _staged_nhwc(x), _C.NHWC, True
        rois = _C.reference_roi_rounding(rois, input.dtype)
        n, c, h, w = x.shape
        k = rois.shape[0]
        out = _empty_like_layout(x, (k, c, ph, pw), layout)
        with _C.on_device(x.device):
            _C.check(_C.lib().d2amd_roi_align_forward(
                _C.ptr(x), _C.ptr(rois), _C.ptr(out), n, c, h, w, k, ph, pw, float(spatial_scale),
                int(sampling_ratio), int(bool(aligned)), _C.dtype_code(x), layout, _C.stream()))
        if staged:
            from ..modeling.poolers import _to_nchw

            out, layout = _to_nchw(out), _C.NCHW
        ctx.save_for_backward(rois)
        ctx.cfg = (ph, pw, float(spatial_scale), int(sampling_ratio), bool(aligned), tuple(x.shape), layout)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        ph, pw, scale, sr, aligned, shape, layout = ctx.cfg
        if layout == "f64":
            return f64_backward(grad_output, rois, shape, ph, pw, scale, sr, aligned, False), None, None, None, None, None
        n, c, h, w = shape
        k = rois.shape[0]
        fused = max(ph, pw) <= 32  # tile-gather backward (NHWC kernel; NCHW inputs get channels_last grads)
        if fused:
            layout = _C.NHWC
        if layout == _C.NHWC:
            g = grad_output.contiguous(memory_format=torch.channels_last)
        else:
            g = grad_output.contiguous()
        gin = _empty_like_layout(g, shape, layout)
        ws, ws_bytes = None, 0
        if fused:
            # per-ROI records + per-tile ROI lists (d2amd_roi_pooler_backward_workspace_bytes for one level)
            nt = ((h + 7) // 8) * ((w + 7) // 8) * n
            al = lambda x: (x + 255) // 256 * 256
            ws_bytes = al(48 * max(k, 1)) + al(4 * nt) + 64 * 32 * nt + 256
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=g.device)
        elif g.dtype != torch.float32:
            ws = torch.empty(n * c * h * w, dtype=torch.float32, device=g.device)
            ws_bytes = ws.numel() * 4
        with _C.on_device(g.device):
            _C.check(_C.lib().d2amd_roi_align_backward(
                _C.ptr(g), _C.ptr(rois), _C.ptr(gin), n, c, h, w, k, ph, pw, scale, sr, int(aligned),
                _C.dtype_code(g), layout, _C.ptr(ws), ws_bytes, _C.stream()))
        return gin, None, None, None, None, None


def roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """Functional form with torchvision.ops.roi_align's signature (re-exported by
    detectron2/layers/__init__.py:6).  `boxes`: Tensor[K,5] or a list of Tensor[L,4] per image."""
    if not isinstance(boxes, torch.Tensor):
        ids = torch.cat([torch.full_like(b[:, :1], i) for i, b in enumerate(boxes)], 0)
        boxes = torch.cat([ids, torch.cat(list(boxes), 0)], 1)
    return _ROIAlign.apply(input, boxes, output_size, spatial_scale, sampling_ratio, aligned)


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=True):
        """Same arguments as the reference (roi_align.py:8-37): `aligned=True` shifts the ROI by
        -0.5 px after scaling (the correct pixel model); `sampling_ratio=0` samples densely."""
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.aligned = aligned

    def forward(self, input, rois):
        """input: NCHW images; rois: Bx5 boxes (batch index, x1, y1, x2, y2)."""
        assert rois.dim() == 2 and rois.size(1) == 5
        if input.is_quantized:
            input = input.dequantize()
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)

    def __repr__(self):
        tmpstr = self.__class__.__name__ + "("
        tmpstr += "output_size=" + str(self.output_size)
        tmpstr += ", spatial_scale=" + str(self.spatial_scale)
        tmpstr += ", sampling_ratio=" + str(self.sampling_ratio)
        tmpstr += ", aligned=" + str(self.aligned)
        tmpstr += ")"
        return tmpstr
