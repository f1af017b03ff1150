"""Registers the reference's native-op surface -- torch.ops.detectron2.{nms_rotated,
box_iou_rotated, roi_align_rotated_forward, roi_align_rotated_backward} with the schemas of
detectron2/layers/csrc/vision.cpp:115-120 -- on top of the C ABI (libd2amd.so), plus
torch.ops.d2amd.* for the ops the reference gets from torchvision / plain torch.  Registered
through torch.library so they are TorchScript-callable and opaque to torch.compile, like the
reference's ops (SURVEY 8b "Threading / streams")."""
import torch

from .. import _C
from .roi_align import _empty_like_layout, _prep_input

_DEFS = {
    "nms_rotated": "(Tensor dets, Tensor scores, float iou_threshold) -> Tensor",
    "box_iou_rotated": "(Tensor boxes1, Tensor boxes2) -> Tensor",
    "roi_align_rotated_forward": "(Tensor input, Tensor rois, float spatial_scale, int pooled_height, "
                                 "int pooled_width, int sampling_ratio) -> Tensor",
    "roi_align_rotated_backward": "(Tensor grad, Tensor rois, float spatial_scale, int pooled_height, "
                                  "int pooled_width, int batch_size, int channels, int height, int width, "
                                  "int sampling_ratio) -> Tensor",
}


def _nms_launch(boxes, scores, idxs, iou_threshold, rotated, stream_ptr=None):
    """Allocate the outputs / workspace of one NMS on torch's current stream and enqueue the device
    pipeline (d2amd_nms) on `stream_ptr` (default: the current stream).  No host sync."""
    bw = 5 if rotated else 4
    assert boxes.dim() == 2 and boxes.shape[1] == bw, boxes.shape
    n = boxes.shape[0]
    _C.require_gpu(boxes, scores, idxs, op="nms")
    boxes = boxes.detach().float().contiguous()
    scores = scores.detach().float().contiguous()
    assert scores.shape[0] == n
    max_per_class = 0
    if idxs is not None:
        idxs = idxs.detach().to(torch.int64).contiguous()
        assert idxs.shape[0] == n
        if n > 16384:
            # the suppression bitmask is n x (largest category / 64) words: size it from the data
            # (one extra host sync, only for very large inputs)
            max_per_class = int(torch.unique(idxs, return_counts=True)[1].max().item())
    L = _C.lib()
    with _C.on_device(boxes.device):
        ws_bytes = L.d2amd_nms_workspace_bytes(n, max_per_class, int(rotated))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=boxes.device)
        keep = torch.empty(n, dtype=torch.int64, device=boxes.device)
        result = torch.empty(4, dtype=torch.int64, device=boxes.device)
        _C.check(L.d2amd_nms(_C.ptr(boxes), _C.ptr(scores), _C.ptr(idxs), n, float(iou_threshold), int(rotated),
                             max_per_class, _C.ptr(keep), _C.ptr(result), _C.ptr(ws), ws_bytes,
                             stream_ptr if stream_ptr is not None else _C.stream()))
    return keep, result, (boxes, scores, idxs, ws)  # the inputs / workspace must outlive the launch


def _nms_finish(keep, num, flags):
    if flags & 2:
        raise RuntimeError("batched_nms: category ids must be in [0, 65535]")
    if flags & 1:
        raise RuntimeError("batched_nms: internal error: category larger than max_per_class")
    return keep[:num]


def nms_impl(boxes, scores, idxs, iou_threshold, rotated):
    """Shared driver of nms / batched_nms / nms_rotated / batched_nms_rotated (d2amd_nms)."""
    if boxes.shape[0] == 0:  # nothing to compute on any device (nms.py:125-126)
        assert boxes.dim() == 2 and boxes.shape[1] == (5 if rotated else 4), boxes.shape
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    keep, result, _hold = _nms_launch(boxes, scores, idxs, iou_threshold, rotated)
    num, flags = result[:2].tolist()  # the only host sync of the NMS pipeline
    return _nms_finish(keep, num, flags)


_SIDE_STREAMS = {}
_BATCH_MAX = None


def _nms_images_batched(inputs, iou_threshold, rotated, defer=False):
    """All images through d2amd_nms_batched: one launch per pipeline stage for the whole batch, one
    [count, 2] result tensor, one host sync."""
    ct = _C.ctypes
    L = _C.lib()
    dev = inputs[0][0].device
    bw = 5 if rotated else 4
    cnt = len(inputs)
    hold = []
    arr = lambda vals: (ct.c_void_p * cnt)(*vals)
    with _C.on_device(dev):
        result = torch.empty((cnt, 4), dtype=torch.int64, device=dev)
        pb, ps, pi, pk, pr, pw = [], [], [], [], [], []
        ns, wb, keeps = [], [], []
        for k, (boxes, scores, idxs) in enumerate(inputs):
            assert boxes.dim() == 2 and boxes.shape[1] == bw, boxes.shape
            n = boxes.shape[0]
            _C.require_gpu(boxes, scores, idxs, op="nms")
            boxes = boxes.detach().float().contiguous()
            scores = scores.detach().float().contiguous()
            assert scores.shape[0] == n
            if idxs is not None:
                idxs = idxs.detach().to(torch.int64).contiguous()
                assert idxs.shape[0] == n
            nbytes = L.d2amd_nms_workspace_bytes(n, 0, int(rotated))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            keep = torch.empty(n, dtype=torch.int64, device=dev)
            hold.append((boxes, scores, idxs, ws))
            keeps.append(keep)
            pb.append(boxes.data_ptr()); ps.append(scores.data_ptr())
            pi.append(idxs.data_ptr() if idxs is not None else None)
            pk.append(keep.data_ptr()); pr.append(result.data_ptr() + 32 * k); pw.append(ws.data_ptr())
            ns.append(n); wb.append(nbytes)
        _C.check(L.d2amd_nms_batched(cnt, arr(pb), arr(ps), arr(pi), (ct.c_int64 * cnt)(*ns), float(iou_threshold),
                                     int(rotated), None, arr(pk), arr(pr), arr(pw), (ct.c_size_t * cnt)(*wb),
                                     _C.stream()))
    def finish(with_finite=False, extra=None):
        """with_finite: also return, per image, how many kept boxes have a score > -inf; extra: a device tensor of
        int64 values read in the same host transfer (returned as a list)."""
        flat = result.flatten() if extra is None else torch.cat([result.flatten(), extra.flatten().to(torch.int64)])
        vals = flat.tolist()  # the only host sync; `hold` keeps inputs / workspaces alive until here
        counts = [vals[4 * k:4 * k + 4] for k in range(cnt)]
        hold.clear()
        kept = [_nms_finish(keep, c[0], c[1]) for keep, c in zip(keeps, counts)]
        if not with_finite and extra is None:
            return kept
        return kept, [c[2] for c in counts], vals[4 * cnt:]

    return finish if defer else finish()


def nms_images(inputs, iou_threshold, rotated=False, defer=False):
    """NMS of every image of a batch in one call: `inputs` = [(boxes, scores, idxs | None), ...].
    defer=True: everything is enqueued and a callable is returned; calling it performs the one host sync and returns
    the kept indices -- the caller can enqueue independent work (e.g. the anchor labelling IoU) in between.
    The reference runs the RPN / RetinaNet NMS in a per-image Python loop, each iteration ending in a
    device->host sync (proposal_generator/proposal_utils.py:118-135, meta_arch/dense_detector.py:186-260).
    Images are independent: up to d2amd_nms_batched_max_boxes() boxes per image the whole batch runs as one
    device pipeline (d2amd_nms_batched); larger inputs are enqueued on separate HIP streams that fork from /
    join into the current stream.  Either way the kept counts are read with ONE sync."""
    global _BATCH_MAX
    if not inputs:
        return (lambda with_finite=False, extra=None: ([], [], []) if (with_finite or extra is not None) else []) if defer else []
    if _BATCH_MAX is None:
        _BATCH_MAX = int(_C.lib().d2amd_nms_batched_max_boxes())
    dev = inputs[0][0].device
    if all(b.shape[0] <= _BATCH_MAX and b.device == dev for b, _s, _i in inputs):
        return _nms_images_batched(inputs, iou_threshold, rotated, defer)
    cur = torch.cuda.current_stream(dev)
    pool = _SIDE_STREAMS.setdefault(dev.index, [])
    # every dtype / layout conversion of every image runs on `cur` BEFORE the fork event: the side streams wait for
    # that event only, so nothing they read may be produced after it (conversions inside _nms_launch are then no-ops)
    bw = 5 if rotated else 4
    prepared = []
    for boxes, scores, idxs in inputs:
        assert boxes.dim() == 2 and boxes.shape[1] == bw, boxes.shape
        prepared.append((boxes.detach().float().contiguous(), scores.detach().float().contiguous(),
                         None if idxs is None else idxs.detach().to(torch.int64).contiguous()))
    inputs = prepared
    fork = torch.cuda.Event()
    fork.record(cur)  # fork point: after the conversions, before any NMS kernel of this call
    launched = []
    k = 0
    for boxes, scores, idxs in inputs:
        if boxes.shape[0] == 0:
            launched.append(None)
            continue
        if k == 0:
            st = cur  # the first image stays on the current stream
        else:
            while len(pool) < min(k, 3):
                pool.append(torch.cuda.Stream(device=dev))
            st = pool[(k - 1) % len(pool)]
            st.wait_event(fork)  # inputs are ready; does not wait for the images launched above
        res = _nms_launch(boxes, scores, idxs, iou_threshold, rotated, stream_ptr=_C.ctypes.c_void_p(st.cuda_stream))
        if st is not cur:  # allocated on `cur`'s pool, used on `st`: tell the caching allocator
            for t in (res[0], res[1]) + tuple(x for x in res[2] if x is not None):
                t.record_stream(st)
        launched.append((res, st))
        k += 1
    for item in launched:
        if item is not None and item[1] is not cur:
            cur.wait_stream(item[1])  # join: everything enqueued later on `cur` sees the results
    live = [it for it in launched if it is not None]
    stacked = torch.stack([it[0][1] for it in live]) if live else None

    def finish(with_finite=False, extra=None):
        flat = stacked.flatten() if live else torch.zeros(0, dtype=torch.int64, device=dev)
        if extra is not None:
            flat = torch.cat([flat, extra.flatten().to(torch.int64)])
        vals = flat.tolist() if (live or extra is not None) else []  # ONE host sync
        out, fin, j = [], [], 0
        for (boxes, _s, _i), it in zip(inputs, launched):
            if it is None:
                out.append(torch.empty((0,), dtype=torch.int64, device=boxes.device))
                fin.append(0)
            else:
                c = vals[4 * j:4 * j + 4]
                out.append(_nms_finish(it[0][0], c[0], c[1]))
                fin.append(c[2])
                j += 1
        if not with_finite and extra is None:
            return out
        return out, fin, vals[4 * len(live):]

    return finish if defer else finish()


def _nms_rotated(dets, scores, iou_threshold):
    return nms_impl(dets, scores, None, iou_threshold, True)


def _box_iou_rotated(boxes1, boxes2):
    _C.require_gpu(boxes1, boxes2, op="box_iou_rotated")
    b1 = boxes1.detach().float().contiguous()
    b2 = boxes2.detach().float().contiguous()
    assert b1.dim() == 2 and b1.shape[1] == 5 and b2.dim() == 2 and b2.shape[1] == 5
    n, m = b1.shape[0], b2.shape[0]
    out = torch.empty((n, m), dtype=torch.float32, device=b1.device)  # always fp32 (box_iou_rotated_cpu.cpp:29)
    if n and m:
        with _C.on_device(b1.device):
            _C.check(_C.lib().d2amd_box_iou_rotated(_C.ptr(b1), n, _C.ptr(b2), m, _C.ptr(out), _C.stream()))
    return out


def _roi_align_rotated_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    _C.require_gpu(input, rois, op="roi_align_rotated_forward")
    assert rois.dim() == 2 and rois.shape[1] == 6
    x, layout = _prep_input(input.detach())
    r = rois.detach().float().contiguous()
    n, c, h, w = x.shape
    k = r.shape[0]
    out = _empty_like_layout(x, (k, c, pooled_height, pooled_width), layout)
    if out.numel() == 0:
        return out
    status = torch.zeros(1, dtype=torch.int32, device=x.device)
    with _C.on_device(x.device):
        _C.check(_C.lib().d2amd_roi_align_rotated_forward(
            _C.ptr(x), _C.ptr(r), _C.ptr(out), n, c, h, w, k, pooled_height, pooled_width, float(spatial_scale),
            int(sampling_ratio), _C.dtype_code(x), layout, _C.ptr(status), _C.stream()))
    # ROIAlignRotated_cpu.cpp:236-238: AT_ASSERTM(roi_width >= 0 && roi_height >= 0, ...) -> RuntimeError.  Reading the
    # status word is one host sync; the reference's device forward ends in cudaDeviceSynchronize()
    # (ROIAlignRotated_cuda.cu:379), so the call was never asynchronous for its callers.
    if int(status.item()) != 0:
        raise RuntimeError("ROIs in ROIAlignRotated do not have non-negative size!")
    return out


def _roi_align_rotated_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                                height, width, sampling_ratio):
    _C.require_gpu(grad, rois, op="roi_align_rotated_backward")
    layout = _C.NHWC if (grad.dim() == 4 and not grad.is_contiguous()
                         and grad.is_contiguous(memory_format=torch.channels_last)) else _C.NCHW
    g = grad.detach() if layout == _C.NHWC else grad.detach().contiguous()
    r = rois.detach().float().contiguous()
    shape = (batch_size, channels, height, width)
    gin = _empty_like_layout(g, shape, layout)
    if gin.numel() == 0:
        return gin
    ws, ws_bytes = None, 0
    if g.dtype != torch.float32:
        ws = torch.empty(gin.numel(), dtype=torch.float32, device=g.device)
        ws_bytes = ws.numel() * 4
    with _C.on_device(g.device):
        _C.check(_C.lib().d2amd_roi_align_rotated_backward(
            _C.ptr(g), _C.ptr(r), _C.ptr(gin), batch_size, channels, height, width, r.shape[0], pooled_height,
            pooled_width, float(spatial_scale), int(sampling_ratio), _C.dtype_code(g), layout, _C.ptr(ws),
            ws_bytes, _C.stream()))
    return gin


_IMPLS = {
    "nms_rotated": _nms_rotated,
    "box_iou_rotated": _box_iou_rotated,
    "roi_align_rotated_forward": _roi_align_rotated_forward,
    "roi_align_rotated_backward": _roi_align_rotated_backward,
}

_lib_handle = torch.library.Library("detectron2", "FRAGMENT")
for _name, _schema in _DEFS.items():
    try:
        _lib_handle.define(_name + _schema)
    except RuntimeError:
        pass  # a real detectron2 build already defined the schema; we only add the device kernel
    # "CUDA" is the dispatch key of HIP devices in PyTorch-ROCm
    _lib_handle.impl(_name, _IMPLS[_name], "CUDA")


def _cpu_stub(name):
    def f(*a, **k):
        raise NotImplementedError(
            f"torch.ops.{name if '.' in name else 'detectron2.' + name}: detectron2_amd implements this op for MI355X "
            f"(HIP tensors) only")
    return f


for _name in _DEFS:
    try:
        _lib_handle.impl(_name, _cpu_stub(_name), "CPU")
    except RuntimeError:
        pass


# ---- torch.ops.d2amd.*: the ops the reference gets from torchvision / plain torch, as registered ops, so that the
# Python surface around them (layers.nms.batched_nms, layers.mask_ops.paste_masks_in_image) is `torch.jit.script`-able
# like the reference's (tests/layers/test_nms.py:16-29, tests/layers/test_mask_ops.py:156-165)
_D2AMD_DEFS = {
    "nms": "(Tensor boxes, Tensor scores, float iou_threshold) -> Tensor",
    "batched_nms": "(Tensor boxes, Tensor scores, Tensor idxs, float iou_threshold) -> Tensor",
    "batched_nms_rotated": "(Tensor boxes, Tensor scores, Tensor idxs, float iou_threshold) -> Tensor",
    "paste_masks": "(Tensor masks, Tensor boxes, int img_h, int img_w, float threshold) -> Tensor",
}


def _paste_masks(masks, boxes, img_h, img_w, threshold):
    _C.require_gpu(masks, boxes, op="paste_masks_in_image")
    n = masks.shape[0]
    m = masks.detach()
    if m.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        m = m.float()
    m = m.contiguous()
    b = boxes.detach().float().contiguous()
    out = torch.empty((n, img_h, img_w), dtype=torch.uint8, device=m.device)
    with _C.on_device(m.device):
        _C.check(_C.lib().d2amd_paste_masks(_C.ptr(m), _C.ptr(b), n, m.shape[1], m.shape[2], img_h, img_w,
                                            float(threshold), _C.ptr(out), _C.dtype_code(m), _C.stream()))
    return out.view(torch.bool) if threshold >= 0 else out


_D2AMD_IMPLS = {
    "nms": lambda boxes, scores, thr: nms_impl(boxes, scores, None, thr, False),
    "batched_nms": lambda boxes, scores, idxs, thr: nms_impl(boxes, scores, idxs, thr, False),
    "batched_nms_rotated": lambda boxes, scores, idxs, thr: nms_impl(boxes, scores, idxs, thr, True),
    "paste_masks": _paste_masks,
}

def _cpu_nms_stub(name):
    stub = _cpu_stub("d2amd." + name)

    def f(boxes, *rest):
        if boxes.shape[0] == 0:  # nothing to compute on any device (nms.py:125-126)
            return torch.empty((0,), dtype=torch.int64, device=boxes.device)
        return stub()
    return f


_d2amd_lib = torch.library.Library("d2amd", "DEF")
for _name, _schema in _D2AMD_DEFS.items():
    _d2amd_lib.define(_name + _schema)
    _d2amd_lib.impl(_name, _D2AMD_IMPLS[_name], "CUDA")
    _d2amd_lib.impl(_name, _cpu_stub("d2amd." + _name) if _name == "paste_masks" else _cpu_nms_stub(_name), "CPU")
