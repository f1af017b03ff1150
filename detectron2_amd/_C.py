"""ctypes binding of libd2amd.so -- the product's only route to the hot path.

There is NO fallback: if the HIP library is missing or an op is given a CPU tensor, this raises.
(Reference counterpart: `from detectron2 import _C`, layers/deform_conv.py:505-514.)
"""
import contextlib
import ctypes
import os

import torch  # must be imported first: libd2amd.so binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("D2AMD_LIB_PATH") or os.path.join(_HERE, "lib", "libd2amd.so")  # (override: same-box A/B of two builds)
_lib = None

F32, F16, BF16 = 0, 1, 2
NCHW, NHWC = 0, 1
_DTYPES = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float
_d = ctypes.c_double
_sz = ctypes.c_size_t


class DcnParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "B", "C", "H", "W", "Co", "kh", "kw", "stride_h", "stride_w", "pad_h", "pad_w", "dil_h",
        "dil_w", "groups", "deformable_groups", "dtype", "layout")]


class NmsGather(ctypes.Structure):
    """d2amd_nms_gather (include/d2amd.h)"""
    _fields_ = [("count", ctypes.c_int), ("src", ctypes.c_void_p * 4), ("dst", ctypes.c_void_p * 4),
                ("row_bytes", ctypes.c_int * 4)]


class SampleImage(ctypes.Structure):
    """d2amd_sample_image (include/d2amd.h)"""
    _fields_ = [("proposals", ctypes.c_void_p), ("limits", ctypes.c_void_p), ("gt_boxes", ctypes.c_void_p),
                ("gt_classes", ctypes.c_void_p), ("keys", ctypes.c_void_p), ("max_proposals", ctypes.c_int),
                ("n_limits", ctypes.c_int), ("num_gt", ctypes.c_int), ("limit_stride", ctypes.c_int)]


class PoolerParams(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int) for n in ("num_levels", "N", "C")] +
                [("H", ctypes.c_int * 8), ("W", ctypes.c_int * 8), ("spatial_scale", ctypes.c_float * 8)] +
                [(n, ctypes.c_int) for n in ("pooled_h", "pooled_w", "sampling_ratio", "aligned", "dtype", "layout",
                                             "min_level", "max_level", "canonical_level")] +
                [("canonical_box_size", ctypes.c_float), ("roi_rounding", ctypes.c_int)])


_SIGNATURES = {
    "d2amd_version": (ctypes.c_char_p, []),
    "d2amd_compiler_version": (ctypes.c_char_p, []),
    "d2amd_hip_version": (ctypes.c_char_p, []),
    "d2amd_last_error": (ctypes.c_char_p, []),
    "d2amd_timing_enable": (None, [_i]),
    "d2amd_timing_select": (None, [ctypes.c_char_p]),
    "d2amd_timing_read": (_i, [ctypes.c_char_p, ctypes.POINTER(_d), ctypes.POINTER(_i)]),
    "d2amd_roi_align_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _vp]),
    "d2amd_roi_align_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _vp, _sz, _vp]),
    "d2amd_boxes_to_rois": (_i, [_vp, ctypes.POINTER(_i), _i, _i, _vp, _vp]),
    "d2amd_roi_pooler_supported": (_i, [ctypes.POINTER(PoolerParams), _i]),
    "d2amd_roi_pooler_forward": (_i, [ctypes.POINTER(PoolerParams), ctypes.POINTER(_vp), _vp, _vp, _i, _vp]),
    "d2amd_roi_pooler_rotated_supported": (_i, [ctypes.POINTER(PoolerParams)]),
    "d2amd_roi_pooler_rotated_forward": (_i, [ctypes.POINTER(PoolerParams), ctypes.POINTER(_vp), _vp, _vp, _i, _vp, _vp]),
    "d2amd_roi_pooler_rotated_backward_workspace_bytes": (_sz, [ctypes.POINTER(PoolerParams), _i]),
    "d2amd_roi_pooler_rotated_backward": (_i, [ctypes.POINTER(PoolerParams), _vp, _vp, ctypes.POINTER(_vp), _i, _vp, _sz,
                                               _vp]),
    "d2amd_roi_pooler_forward_box_lists": (_i, [ctypes.POINTER(PoolerParams), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                                ctypes.POINTER(_i), _i, _vp, _vp, _vp]),
    "d2amd_roi_pooler_workspace_bytes": (_sz, [_i]),
    "d2amd_roi_pooler_forward_workspace_bytes": (_sz, [_i]),
    "d2amd_roi_pooler_forward_ordered": (_i, [ctypes.POINTER(PoolerParams), ctypes.POINTER(_vp), _vp, _vp, _i, _vp, _sz,
                                              _vp]),
    "d2amd_roi_pooler_forward_box_lists_ordered": (_i, [ctypes.POINTER(PoolerParams), ctypes.POINTER(_vp),
                                                        ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _vp, _vp, _vp, _sz,
                                                        _vp]),
    "d2amd_roi_pooler_forward_pair": (_i, [ctypes.POINTER(PoolerParams), ctypes.POINTER(_vp), _vp, _vp, _i,
                                           ctypes.POINTER(PoolerParams), _vp, _vp, _i, _vp]),
    "d2amd_roi_pooler_forward_pair_box_lists": (_i, [ctypes.POINTER(PoolerParams), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                                     ctypes.POINTER(_i), _vp, _vp, ctypes.POINTER(PoolerParams),
                                                     ctypes.POINTER(_vp), ctypes.POINTER(_i), _vp, _vp, _i, _vp]),
    "d2amd_roi_pooler_backward_workspace_bytes": (_sz, [ctypes.POINTER(PoolerParams), _i]),
    "d2amd_roi_pooler_backward": (_i, [ctypes.POINTER(PoolerParams), _vp, _vp, ctypes.POINTER(_vp), _i, _vp, _sz,
                                       _vp]),
    "d2amd_roi_pooler_backward_accumulate": (_i, [ctypes.POINTER(PoolerParams), _vp, _vp, ctypes.POINTER(_vp), _i, _vp,
                                                  _sz, _vp]),
    "d2amd_roi_pooler_backward_pair_workspace_bytes": (_sz, [ctypes.POINTER(PoolerParams), _i, _i]),
    "d2amd_roi_pooler_backward_pair": (_i, [ctypes.POINTER(PoolerParams), _vp, _vp, _i, ctypes.POINTER(PoolerParams), _vp,
                                            _vp, _i, ctypes.POINTER(_vp), _vp, _sz, _vp]),
    "d2amd_roi_pooler_forward_pair_records": (_i, [ctypes.POINTER(PoolerParams), ctypes.POINTER(_vp), _vp, _vp, _i, ctypes.POINTER(PoolerParams),
                                                   _vp, _vp, _i, _vp, _sz, ctypes.POINTER(ctypes.c_int), _vp]),
    "d2amd_roi_pooler_backward_pair_phase": (_i, [ctypes.POINTER(PoolerParams), _vp, _vp, _i, ctypes.POINTER(PoolerParams),
                                                  _vp, _vp, _i, ctypes.POINTER(_vp), _vp, _sz, _i, _vp]),
    "d2amd_roi_pooler_backward_phase": (_i, [ctypes.POINTER(PoolerParams), _vp, _vp, ctypes.POINTER(_vp), _i, _vp, _sz,
                                             _i, _vp]),
    "d2amd_roi_align_rotated_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _vp, _vp]),
    "d2amd_roi_align_rotated_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _vp, _sz, _vp]),
    "d2amd_roi_align_f64_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _d, _i, _i, _i, _vp, _vp]),
    "d2amd_roi_align_f64_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _d, _i, _i, _i, _vp]),
    "d2amd_pairwise_iou": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "d2amd_box_iou_rotated": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "d2amd_matcher_workspace_bytes": (_sz, [_i]),
    "d2amd_match_boxes": (_i, [_vp, _i, _vp, _i, ctypes.POINTER(_f), ctypes.POINTER(ctypes.c_int8), _i, _i, _vp, _vp,
                               _vp, _sz, _vp]),
    "d2amd_match_boxes_batch_workspace_bytes": (_sz, [ctypes.POINTER(_i), _i]),
    "d2amd_match_boxes_batch": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _vp, _i, ctypes.POINTER(_f),
                                     ctypes.POINTER(ctypes.c_int8), _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "d2amd_label_and_sample_max_candidates": (_i, []),
    "d2amd_label_and_sample_proposals": (_i, [ctypes.POINTER(SampleImage), _i, ctypes.POINTER(_f),
                                              ctypes.POINTER(ctypes.c_int8), _i, _i, _i, _i64, _i, _vp, _vp, _vp, _vp,
                                              _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "d2amd_uniform_keys": (_i, [_vp, _vp, _i64, _vp]),
    "d2amd_subsample_labels_workspace_bytes": (_sz, [_i, _i64, _i, _i]),
    "d2amd_subsample_labels": (_i, [_vp, _i, _i, _i64, _vp, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d2amd_match_quality_matrix": (_i, [_vp, _i, _i, ctypes.POINTER(_f), ctypes.POINTER(ctypes.c_int8), _i, _i, _vp,
                                        _vp, _vp, _sz, _vp]),
    "d2amd_rpn_select_workspace_bytes": (_sz, [_i, _i]),
    "d2amd_rpn_select_proposals": (_i, [_vp, _vp, _vp, _i, _i, ctypes.POINTER(_i), _i, ctypes.POINTER(_i), _i, _f,
                                        ctypes.POINTER(_f), _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d2amd_rpn_select_proposals_levels": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i,
                                               ctypes.POINTER(_i), _i, ctypes.POINTER(_i), _i, _f, ctypes.POINTER(_f),
                                               _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d2amd_dense_select_workspace_bytes": (_sz, [_i, ctypes.POINTER(_i), _i, _i, _i]),
    "d2amd_dense_select_predictions": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i,
                                            ctypes.POINTER(_i), _i, _i, _f, _i, ctypes.POINTER(_f), _f, _vp, _vp, _vp,
                                            _vp, _vp, _vp, _vp, _sz, _vp]),
    "d2amd_nms_workspace_bytes": (_sz, [_i64, _i64, _i]),
    "d2amd_nms": (_i, [_vp, _vp, _vp, _i64, _d, _i, _i64, _vp, _vp, _vp, _sz, _vp]),
    "d2amd_nms_batched": (_i, [_i, _vp, _vp, _vp, _vp, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "d2amd_nms_batched_max_boxes": (_i, []),
    "d2amd_nms_runs": (_i, [_vp, _vp, _vp, _i64, ctypes.POINTER(_i), _i, _i, _i, _d, _i, _i64, _vp, _vp, _vp, _sz, _vp,
                            _vp]),
    "d2amd_nms_batched_runs": (_i, [_i, _vp, _vp, _vp, _vp, ctypes.POINTER(_i), _i, _i, _i, _d, _i, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp]),
    "d2amd_paste_masks": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _i, _vp]),
    "d2amd_bitmask_crop_and_resize": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "d2amd_bitmask_crop_and_resize_indexed": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "d2amd_transpose_batched": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "d2amd_transpose_multi": (_i, [_vp, _vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), _i, _i, _i, _vp]),
    "d2amd_polygon_crop_and_resize": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "d2amd_bitmask_crop_and_resize_batch": (_i, [_i, ctypes.POINTER(_vp), ctypes.POINTER(_i), ctypes.POINTER(_vp),
                                                 ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i, _vp, _vp, _vp]),
    "d2amd_fast_rcnn_filter_workspace_bytes": (_sz, [ctypes.POINTER(_i), _i]),
    "d2amd_fast_rcnn_filter": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i,
                                    ctypes.POINTER(_i), ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d2amd_fast_rcnn_park": (_i, [ctypes.POINTER(_i), _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "d2amd_fast_rcnn_take": (_i, [ctypes.POINTER(_i), _i, _i, _i, _i, ctypes.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _vp]),
    "d2amd_fast_rcnn_predict": (_i, [_vp, _vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i,
                                     ctypes.POINTER(ctypes.c_float), ctypes.c_float, _i, _vp, _vp, _vp]),
    "d2amd_proposals_pad": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _vp]),
    "d2amd_mask_rcnn_inference": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "d2amd_mask_rcnn_loss_workspace_bytes": (_sz, [_i]),
    "d2amd_mask_rcnn_loss_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "d2amd_mask_rcnn_loss_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "d2amd_mask_rcnn_loss_forward_masked": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "d2amd_mask_rcnn_loss_backward_masked": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "d2amd_deform_conv_workspace_bytes": (_sz, [ctypes.POINTER(DcnParams), _i]),
    "d2amd_deform_conv_forward": (_i, [ctypes.POINTER(DcnParams), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d2amd_deform_conv_backward": (_i, [ctypes.POINTER(DcnParams)] + [_vp] * 11 + [_sz, _vp]),
    "d2amd_deform_conv_columns_bytes": (_sz, [ctypes.POINTER(DcnParams)]),
    "d2amd_deform_conv_column_path": (_i, [ctypes.POINTER(DcnParams)]),
    "d2amd_deform_conv_forward_columns": (_i, [ctypes.POINTER(DcnParams)] + [_vp] * 8 + [_sz, _vp]),
    "d2amd_deform_conv_backward_columns": (_i, [ctypes.POINTER(DcnParams)] + [_vp] * 12 + [_sz, _vp]),
}


def lib():
    """Load libd2amd.so (fails loudly if it was not built: run `python -m detectron2_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"detectron2_amd: {LIB_PATH} not found. The HIP extension is required (there is no "
                "CPU/eager fallback); build it with `python -m detectron2_amd.build`.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return list(_SIGNATURES)


EUNSUPPORTED = -4


def check(rc):
    if rc != 0:
        msg = lib().d2amd_last_error().decode()
        raise RuntimeError(f"d2amd error {rc}: {msg}")


def dtype_code(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise RuntimeError(f"detectron2_amd: unsupported dtype {t.dtype} (float32/float16/bfloat16)")


# Strict reference parity for 16-bit features: the ROIs rounded to the FEATURE dtype, as the reference does
# (layers/roi_align.py:60: `rois.to(dtype=input.dtype)` -- under bf16 autocast a coordinate of 1,000 px lands on a multiple
# of 4 or 8).  Off by default -- the kernels take fp32 ROIs whatever the features are, which is the better numerics (fp16
# coordinates above 1,024 px are 1 px apart: up to 0.18 of the feature range, tests/test_gpu_pooler.py).  Read ONCE from
# D2AMD_REFERENCE_ROI_ROUNDING at import; set_reference_roi_rounding() changes it in a running process.
_REFERENCE_ROI_ROUNDING = os.environ.get("D2AMD_REFERENCE_ROI_ROUNDING") == "1"


def set_reference_roi_rounding(on: bool) -> bool:
    """-> the previous setting"""
    global _REFERENCE_ROI_ROUNDING
    prev, _REFERENCE_ROI_ROUNDING = _REFERENCE_ROI_ROUNDING, bool(on)
    return prev


def reference_roi_rounding_on() -> bool:
    return _REFERENCE_ROI_ROUNDING


def reference_roi_rounding(rois, feature_dtype):
    """The (single-level) ROIAlign layers' ROIs: rounded to the feature dtype when the strict mode is on.  -> fp32 tensor.
    (The fused multi-level pooler rounds in its kernels, BEHIND the level assignment: d2amd_pooler_params.roi_rounding.)"""
    r = rois.detach()
    if feature_dtype in (torch.float16, torch.bfloat16) and _REFERENCE_ROI_ROUNDING:
        r = r.to(feature_dtype)
    return r.float().contiguous()


def require_gpu(*tensors, op="op"):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NotImplementedError(
                f"detectron2_amd.{op}: got a {t.device} tensor. This package implements the hot path "
                "for MI355X only (HIP tensors); there is no CPU implementation or fallback.")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


_NULL_CTX = contextlib.nullcontext()


_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def on_device(device):
    """Context that makes `device` current for a C-ABI call (kernels launch on the current device).
    One process per GPU is the deployment model, so this is normally a no-op object."""
    if device.index is None or device.index == _get_device():
        return _NULL_CTX
    return torch.cuda.device(device)


def stream():
    """hipStream_t of torch's current stream on the current device, as a void*.  Uses the raw C accessor:
    torch.cuda.current_stream() builds a Stream object and costs ~15 us per call, more than some of the
    kernels launched on it."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(_get_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
