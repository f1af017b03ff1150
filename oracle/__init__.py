"""CPU ORACLE for the detectron2 detection hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  The product (``detectron2_amd``) never does.

``oracle.<fn>`` are numpy wrappers over ``oracle/_build/libd2oracle.so`` (plain-C restatement,
see d2_oracle.c for per-function reference citations and parity-pinning status).
``oracle.ref`` exposes the reference's own compiled C++ / Python where available.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libd2oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "d2_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_nms.restype = ctypes.c_int64
        _lib.orc_nms_rotated.restype = ctypes.c_int64
        _lib.orc_batched_nms.restype = ctypes.c_int64
        _lib.orc_single_box_iou_rotated.restype = ctypes.c_float
        _lib.orc_paste_sample.restype = ctypes.c_float
    return _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a, t=_f32p):
    return a.ctypes.data_as(t) if a is not None else None


def roi_align_forward(x, rois, output_size, spatial_scale, sampling_ratio, aligned):
    x, rois = _f32(x), _f32(rois).reshape(-1, 5)
    n, c, h, w = x.shape
    ph, pw = output_size
    out = np.zeros((rois.shape[0], c, ph, pw), np.float32)
    lib().orc_roi_align_forward(_p(x), _p(rois), rois.shape[0], c, h, w, ph, pw,
                                ctypes.c_float(spatial_scale), int(sampling_ratio), int(bool(aligned)),
                                _p(out))
    return out


def roi_align_backward(grad_out, rois, input_shape, spatial_scale, sampling_ratio, aligned):
    g, rois = _f32(grad_out), _f32(rois).reshape(-1, 5)
    n, c, h, w = input_shape
    ph, pw = g.shape[2], g.shape[3]
    gin = np.zeros((n, c, h, w), np.float32)
    lib().orc_roi_align_backward(_p(g), _p(rois), rois.shape[0], c, h, w, ph, pw,
                                 ctypes.c_float(spatial_scale), int(sampling_ratio),
                                 int(bool(aligned)), _p(gin))
    return gin


# cos(theta) with a float argument in ROIAlignRotated_cpu.cpp:233-234: which overload the
# reference build picks is pinned against the compiled reference in tests/test_oracle_golden.py.
ROT_TRIG_MODE = 1  # pinned: (float)cos((double)theta) is bit-exact vs oracle/_ref


def roi_align_rotated_forward(x, rois, output_size, spatial_scale, sampling_ratio, trig_mode=None):
    x, rois = _f32(x), _f32(rois).reshape(-1, 6)
    n, c, h, w = x.shape
    ph, pw = output_size
    out = np.zeros((rois.shape[0], c, ph, pw), np.float32)
    rc = lib().orc_roi_align_rotated_forward(
        _p(x), _p(rois), rois.shape[0], c, h, w, ph, pw, ctypes.c_float(spatial_scale),
        int(sampling_ratio), ROT_TRIG_MODE if trig_mode is None else trig_mode, _p(out))
    if rc != 0:
        raise RuntimeError("ROIs in ROIAlignRotated do not have non-negative size!")
    return out


def roi_align_rotated_backward(grad_out, rois, input_shape, spatial_scale, sampling_ratio,
                               trig_mode=None):
    g, rois = _f32(grad_out), _f32(rois).reshape(-1, 6)
    n, c, h, w = input_shape
    ph, pw = g.shape[2], g.shape[3]
    gin = np.zeros((n, c, h, w), np.float32)
    rc = lib().orc_roi_align_rotated_backward(
        _p(g), _p(rois), rois.shape[0], c, h, w, ph, pw, ctypes.c_float(spatial_scale),
        int(sampling_ratio), ROT_TRIG_MODE if trig_mode is None else trig_mode, _p(gin))
    if rc != 0:
        raise RuntimeError("ROIs in ROIAlignRotated do not have non-negative size!")
    return gin


def pairwise_iou(b1, b2, mode="iou"):
    b1, b2 = _f32(b1).reshape(-1, 4), _f32(b2).reshape(-1, 4)
    out = np.zeros((b1.shape[0], b2.shape[0]), np.float32)
    lib().orc_pairwise_iou(_p(b1), b1.shape[0], _p(b2), b2.shape[0],
                           {"iou": 0, "ioa": 1, "intersection": 2}[mode], _p(out))
    return out


def matcher(quality, thresholds, labels, allow_low_quality_matches=False):
    """Matcher(thresholds, labels, allow_low_quality_matches)(quality) -> (matches int64, labels int8)."""
    q = _f32(quality)
    m, n = q.shape
    thr = _f32(thresholds)
    lab = np.ascontiguousarray(np.asarray(labels, dtype=np.int8))
    assert lab.shape[0] == thr.shape[0] + 1
    matches = np.zeros(n, np.int64)
    out = np.zeros(n, np.int8)
    lib().orc_matcher(_p(q), m, n, _p(thr), lab.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)), thr.shape[0],
                      int(bool(allow_low_quality_matches)), _p(matches, _i64p),
                      out.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)))
    return matches, out


def box_iou_rotated(b1, b2):
    b1, b2 = _f32(b1).reshape(-1, 5), _f32(b2).reshape(-1, 5)
    out = np.zeros((b1.shape[0], b2.shape[0]), np.float32)
    lib().orc_box_iou_rotated(_p(b1), b1.shape[0], _p(b2), b2.shape[0], _p(out))
    return out


def nms(boxes, scores, iou_threshold):
    boxes, scores = _f32(boxes).reshape(-1, 4), _f32(scores).reshape(-1)
    keep = np.zeros(max(len(scores), 1), np.int64)
    k = lib().orc_nms(_p(boxes), _p(scores), ctypes.c_int64(len(scores)),
                      ctypes.c_double(iou_threshold), _p(keep, _i64p))
    return keep[:k].copy()


def nms_rotated(boxes, scores, iou_threshold):
    boxes, scores = _f32(boxes).reshape(-1, 5), _f32(scores).reshape(-1)
    keep = np.zeros(max(len(scores), 1), np.int64)
    k = lib().orc_nms_rotated(_p(boxes), _p(scores), ctypes.c_int64(len(scores)),
                              ctypes.c_double(iou_threshold), _p(keep, _i64p))
    return keep[:k].copy()


def batched_nms(boxes, scores, idxs, iou_threshold, rotated=False):
    bw = 5 if rotated else 4
    boxes, scores = _f32(boxes).reshape(-1, bw), _f32(scores).reshape(-1)
    idxs = np.ascontiguousarray(np.asarray(idxs, dtype=np.int64)).reshape(-1)
    keep = np.zeros(max(len(scores), 1), np.int64)
    k = lib().orc_batched_nms(_p(boxes), _p(scores), _p(idxs, _i64p), ctypes.c_int64(len(scores)),
                              ctypes.c_double(iou_threshold), int(rotated), _p(keep, _i64p))
    return keep[:k].copy()


def paste_masks_in_image(masks, boxes, image_shape, threshold=0.5, return_soft=False, skip_empty=True):
    """skip_empty=True: the reference's CPU path (bbox region only); False: its device path (whole image sampled,
    mask_ops.py:116-119) -- they differ outside the box for thresholds below 0.5 and the soft uint8 output."""
    masks, boxes = _f32(masks), _f32(boxes).reshape(-1, 4)
    n = masks.shape[0]
    h, w = image_shape
    out = np.zeros((n, h, w), np.uint8)
    soft = np.zeros((n, h, w), np.float32) if return_soft else None
    if n:
        lib().orc_paste_masks_ex(_p(masks), _p(boxes), n, masks.shape[1], masks.shape[2], h, w,
                                 ctypes.c_float(threshold), _p(out, _u8p), _p(soft), int(bool(skip_empty)))
    res = out.astype(bool) if threshold >= 0 else out
    return (res, soft) if return_soft else res


def _dcn_out_hw(h, w, kh, kw, stride, pad, dil):
    ho = (h + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    wo = (w + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    return ho, wo


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def deform_conv_forward(x, offset, weight, mask=None, bias=None, stride=1, padding=0, dilation=1,
                        groups=1, deformable_groups=1):
    """mask=None -> DCNv1 (DeformConv); mask given -> DCNv2 (ModulatedDeformConv)."""
    x, offset, weight = _f32(x), _f32(offset), _f32(weight)
    mask = _f32(mask) if mask is not None else None
    bias = _f32(bias) if bias is not None else None
    s, p, d = _pair(stride), _pair(padding), _pair(dilation)
    b, c, h, w = x.shape
    co, _, kh, kw = weight.shape
    ho, wo = _dcn_out_hw(h, w, kh, kw, s, p, d)
    out = np.zeros((b, co, ho, wo), np.float32)
    lib().orc_deform_conv_forward(_p(x), _p(offset), _p(mask), _p(weight), _p(bias), b, c, h, w, co,
                                  kh, kw, s[0], s[1], p[0], p[1], d[0], d[1], groups,
                                  deformable_groups, _p(out))
    return out


def deform_conv_backward(x, offset, weight, grad_out, mask=None, with_bias=False, stride=1,
                         padding=0, dilation=1, groups=1, deformable_groups=1):
    """Returns dict(grad_input, grad_offset, grad_mask|None, grad_weight, grad_bias|None)."""
    x, offset, weight, grad_out = _f32(x), _f32(offset), _f32(weight), _f32(grad_out)
    mask = _f32(mask) if mask is not None else None
    s, p, d = _pair(stride), _pair(padding), _pair(dilation)
    b, c, h, w = x.shape
    co, _, kh, kw = weight.shape
    gi, go, gw = np.zeros_like(x), np.zeros_like(offset), np.zeros_like(weight)
    gm = np.zeros_like(mask) if mask is not None else None
    gb = np.zeros((co,), np.float32) if with_bias else None
    lib().orc_deform_conv_backward(_p(x), _p(offset), _p(mask), _p(weight), _p(grad_out), b, c, h, w,
                                   co, kh, kw, s[0], s[1], p[0], p[1], d[0], d[1], groups,
                                   deformable_groups, _p(gi), _p(go), _p(gm), _p(gw), _p(gb))
    return dict(grad_input=gi, grad_offset=go, grad_mask=gm, grad_weight=gw, grad_bias=gb)


def polygons_to_bitmask(polygons, height, width):
    """detectron2/structures/masks.py:20-36 with pycocotools restated (orc_poly_to_mask): the union of the polygons'
    masks; polygons = list of flat (x0, y0, x1, y1, ...) arrays (double)."""
    out = np.zeros((height, width), np.uint8)
    for p in polygons:
        xy = np.ascontiguousarray(np.asarray(p, np.float64).reshape(-1))
        lib().orc_poly_to_mask(xy.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(xy) // 2, int(height), int(width),
                               _p(out, _u8p))
    return out.astype(bool)


def rasterize_polygons_within_box(polygons, box, mask_size):
    """masks.py:39-86, operation for operation: shift by the box origin, scale by mask_size / max(extent, 0.1) (the
    box is float32 as `box.numpy()` gives it, the polygons float64), rasterise on a mask_size x mask_size grid."""
    box = np.asarray(box, np.float32)
    w, h = box[2] - box[0], box[3] - box[1]
    polys = [np.array(p, np.float64).reshape(-1).copy() for p in polygons]
    for p in polys:
        p[0::2] = p[0::2] - box[0]
        p[1::2] = p[1::2] - box[1]
    ratio_h = mask_size / max(h, 0.1)
    ratio_w = mask_size / max(w, 0.1)
    if ratio_h == ratio_w:
        for p in polys:
            p *= ratio_h
    else:
        for p in polys:
            p[0::2] *= ratio_w
            p[1::2] *= ratio_h
    return polygons_to_bitmask(polys, mask_size, mask_size)
