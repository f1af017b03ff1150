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


def _runs_arg(runs):
    """(run_offsets, runs_are_categories[, num_categories]) -> ctypes arguments of the d2amd_nms*_runs entries."""
    off, are_cls = runs[0], runs[1]
    ncat = int(runs[2]) if len(runs) > 2 and runs[2] else 0
    off = [int(v) for v in off]
    return (_C.ctypes.c_int * len(off))(*off), len(off) - 1, int(bool(are_cls)), ncat


def _gather_arg(srcs):
    """tensors [n, ...] whose kept rows the NMS copies in keep order -> (d2amd_nms_gather, sources, destinations)"""
    assert 1 <= len(srcs) <= 4, "gather: 1..4 arrays"
    g = _C.NmsGather()
    g.count = len(srcs)
    src = [t.detach().contiguous() for t in srcs]
    dst = [torch.empty_like(t) for t in src]
    for t, (a, b) in enumerate(zip(src, dst)):
        rb = a.element_size() * (a[0].numel() if a.shape[0] else 1)
        assert rb % 4 == 0, "gather: rows must be a multiple of 4 bytes"
        g.src[t], g.dst[t], g.row_bytes[t] = a.data_ptr(), b.data_ptr(), rb
    return g, src, dst


def _regather(srcs, dsts, keep, num):
    """the gather of a redone image (general entry: no fused gather)"""
    for a, b in zip(srcs, dsts):
        b[:num] = a[keep[:num]]


def _runs_categories(runs, device):
    """the category ids a `runs_are_categories` input stands for (general-path fallback)"""
    off = torch.tensor(runs[0], dtype=torch.int64)
    return torch.repeat_interleave(torch.arange(len(off) - 1, dtype=torch.int64), off[1:] - off[:-1]).to(device)


# Boxes per category the suppression bitmask of a large input (n > 16,384) is pitched for without looking at the data.
# The reference path never needed such a bound (torchvision's mask is n x n / 64 words); sizing it from the data took a
# host sync + torch.unique (a device merge sort: 18 % of the RetinaNet selection's GPU time in the r02 profile).
# Now: launch with this bound, and only if a category turns out larger (error flag 1, read with the result anyway)
# run again with the exact size.  The mask kernel stops at the last live tile of a row block, so the bound costs
# address space (n x 257 words), not work.
_OPTIMISTIC_PER_CLASS = 16384


def _nms_launch(boxes, scores, idxs, iou_threshold, rotated, stream_ptr=None, runs=None, exact_bound=False,
                gather=None):
    """Allocate the outputs / workspace of one NMS on torch's current stream and enqueue the device
    pipeline (d2amd_nms) on `stream_ptr` (default: the current stream).  No host sync.
    runs = (run_offsets, runs_are_categories): the input is a sequence of pre-sorted runs (d2amd_nms_runs)."""
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
            # the suppression bitmask is n x (largest category / 64) words
            if exact_bound:  # second attempt: size it from the data (one extra host sync)
                max_per_class = int(torch.unique(idxs, return_counts=True)[1].max().item())
            else:
                max_per_class = _OPTIMISTIC_PER_CLASS
    elif runs is not None and runs[1] and n > 16384:
        max_per_class = max(int(b) - int(a) for a, b in zip(runs[0][:-1], runs[0][1:]))  # the runs are the categories
    L = _C.lib()
    with _C.on_device(boxes.device):
        ws_bytes = L.d2amd_nms_workspace_bytes(n, max_per_class, int(rotated))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=boxes.device)
        keep = torch.empty(n, dtype=torch.int64, device=boxes.device)
        result = torch.empty(4, dtype=torch.int64, device=boxes.device)
        st = stream_ptr if stream_ptr is not None else _C.stream()
        if runs is None:
            _C.check(L.d2amd_nms(_C.ptr(boxes), _C.ptr(scores), _C.ptr(idxs), n, float(iou_threshold), int(rotated),
                                 max_per_class, _C.ptr(keep), _C.ptr(result), _C.ptr(ws), ws_bytes, st))
        else:
            off, n_runs, are_cls, ncat = _runs_arg(runs)
            _C.check(L.d2amd_nms_runs(_C.ptr(boxes), _C.ptr(scores), _C.ptr(idxs), n, off, n_runs, are_cls, ncat,
                                      float(iou_threshold), int(rotated), max_per_class, _C.ptr(keep),
                                      _C.ptr(result), _C.ptr(ws), ws_bytes,
                                      _C.ctypes.byref(gather) if gather is not None else None, st))
    return keep, result, (boxes, scores, idxs, ws)  # the inputs / workspace must outlive the launch


def _nms_finish(keep, num, flags):
    if flags & 2:
        raise RuntimeError("batched_nms: category ids must be in [0, 65535]")
    if flags & 1:
        raise RuntimeError("batched_nms: internal error: category larger than max_per_class")
    return keep[:num]


def _nms_redo(boxes, scores, idxs, iou_threshold, rotated, runs, flags):
    """Second attempt of one image after the result flags said the first launch's assumptions did not hold:
    flag 4 = a run was not in order (rank from scratch), flag 1 = a category exceeds the optimistic bitmask pitch
    (size it from the data).  Its own host sync; -> (keep, num, flags, num_finite)."""
    if (flags & 4) or runs is None:
        if runs is not None and runs[1]:
            idxs = _runs_categories(runs, boxes.device)
        runs = None
    keep, result, _hold = _nms_launch(boxes, scores, idxs, iou_threshold, rotated, runs=runs, exact_bound=True)
    num, flags, fin = result[:3].tolist()
    return keep, num, flags, fin


def nms_impl(boxes, scores, idxs, iou_threshold, rotated):
    """Shared driver of nms / batched_nms / nms_rotated / batched_nms_rotated (d2amd_nms)."""
    if boxes.shape[0] == 0:  # nothing to compute on any device (nms.py:125-126)
        assert boxes.dim() == 2 and boxes.shape[1] == (5 if rotated else 4), boxes.shape
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    keep, result, _hold = _nms_launch(boxes, scores, idxs, iou_threshold, rotated)
    num, flags = result[:2].tolist()  # the only host sync of the NMS pipeline
    if (flags & 1) and idxs is not None and boxes.shape[0] > 16384:  # a category beyond the optimistic pitch
        keep, num, flags, _ = _nms_redo(_hold[0], _hold[1], _hold[2], iou_threshold, rotated, None, flags)
    return _nms_finish(keep, num, flags)


_SIDE_STREAMS = {}
_BATCH_MAX = None


_PINNED = {}


def _pinned_int32(n):
    """a pinned host buffer of n int32 for one in-flight result transfer (hipHostMalloc is slow: they are recycled once
    nobody holds them any more -- a captured graph keeps its buffer through the closure that reads it)"""
    import sys

    pool = _PINNED.setdefault(n, [])
    for t in pool:
        if sys.getrefcount(t) <= 3:  # the pool's reference, the loop variable, getrefcount's argument
            return t
    t = torch.empty(n, dtype=torch.int32, pin_memory=True)
    pool.append(t)
    return t


def _nms_images_batched(inputs, iou_threshold, rotated, defer=False, runs=None, gather=None, result_buffer=None,
                        host_mirror=True):
    """All images through d2amd_nms_batched: one launch per pipeline stage for the whole batch, one
    [count, 2] result tensor, one host sync."""
    ct = _C.ctypes
    L = _C.lib()
    dev = inputs[0][0].device
    bw = 5 if rotated else 4
    cnt = len(inputs)
    hold = []
    gathered = []  # per image: (sources, destinations) of the fused gather
    assert gather is None or runs is not None, "gather: only with pre-sorted runs (d2amd_nms*_runs)"
    arr = lambda vals: (ct.c_void_p * cnt)(*vals)
    with _C.on_device(dev):
        # result_buffer: the caller's int32 buffer, 8 words per image for the results + its own status words behind
        # them (written by the caller's kernels): everything the host needs comes back in ONE transfer, with no
        # torch.cat / dtype conversion launches in front of it
        if result_buffer is not None:
            assert result_buffer.dtype == torch.int32 and result_buffer.is_contiguous() and \
                result_buffer.numel() >= 8 * cnt and result_buffer.device == dev
            result = result_buffer[:8 * cnt].view(torch.int64).view(cnt, 4)
        else:
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
        if runs is None:
            _C.check(L.d2amd_nms_batched(cnt, arr(pb), arr(ps), arr(pi), (ct.c_int64 * cnt)(*ns),
                                         float(iou_threshold), int(rotated), None, arr(pk), arr(pr), arr(pw),
                                         (ct.c_size_t * cnt)(*wb), _C.stream()))
        else:
            off, n_runs, are_cls, ncat = _runs_arg(runs)
            garr = None
            if gather is not None:
                gs = [_gather_arg(g) for g in gather]
                garr = (_C.NmsGather * cnt)(*[g[0] for g in gs])
                gathered.extend((g[1], g[2]) for g in gs)
            _C.check(L.d2amd_nms_batched_runs(cnt, arr(pb), arr(ps), None if are_cls else arr(pi),
                                              (ct.c_int64 * cnt)(*ns), off, n_runs, are_cls, ncat,
                                              float(iou_threshold), int(rotated), None, arr(pk), arr(pr), arr(pw),
                                              (ct.c_size_t * cnt)(*wb), garr, _C.stream()))
    # result_buffer: the copy to the host is ENQUEUED here, into pinned memory, right behind the kernels (inside a
    # captured graph: a memcpy node); the host then only waits for the stream -- a blocking .tolist() is a
    # synchronous hipMemcpy of its own after that wait
    mirror = None
    if result_buffer is not None and host_mirror:  # (host_mirror=False: the caller reads the counts on the DEVICE; a
        # finish() call still works, through a synchronous read)
        mirror = _pinned_int32(result_buffer.numel())
        mirror.copy_(result_buffer, non_blocking=True)
        mirror_stream = torch.cuda.current_stream(dev)

    def finish(with_finite=False, extra=None):
        """with_finite: also return, per image, how many kept boxes have a score > -inf; extra: a device tensor of
        int64 values read in the same host transfer (returned as a list)."""
        if result_buffer is not None and mirror is None:
            assert extra is None
            v32 = result_buffer.tolist()
            vals = v32[0:8 * cnt:2] + v32[8 * cnt:]
        elif result_buffer is not None:  # (values are < 2^31: the low words; the tail = the caller's status words)
            assert extra is None
            # the only host sync (`hold` keeps inputs / workspaces alive until here): the stream the copy was enqueued
            # on -- and the current one: a replayed graph runs where it is launched, not where it was captured
            mirror_stream.synchronize()
            cur = torch.cuda.current_stream(dev)
            if cur != mirror_stream:
                cur.synchronize()
            v32 = mirror.tolist()
            vals = v32[0:8 * cnt:2] + v32[8 * cnt:]
        else:
            flat = result.flatten() if extra is None else torch.cat([result.flatten(), extra.flatten().to(torch.int64)])
            vals = flat.tolist()  # the only host sync; `hold` keeps inputs / workspaces alive until here
        counts = [vals[4 * k:4 * k + 4] for k in range(cnt)]
        kept = []
        for k, (keep, c) in enumerate(zip(keeps, counts)):
            if c[1] & 4:  # a run was not in order: rank this image from scratch (general entry, its own sync)
                keep, c[0], c[1], c[2] = _nms_redo(hold[k][0], hold[k][1], hold[k][2], iou_threshold, rotated, runs, c[1])
                if gathered:
                    _regather(gathered[k][0], gathered[k][1], keep, c[0])
            kept.append(_nms_finish(keep, c[0], c[1]))
        if not defer:
            hold.clear()  # (a deferred closure may be called once per replay of a captured graph: the inputs and
            # workspaces the captured kernels read, and the redo path above, stay alive as long as the closure does)
        if not with_finite and extra is None and result_buffer is None:
            return kept
        return kept, [c[2] for c in counts], vals[4 * cnt:]

    finish.gathered = [g[1] for g in gathered]  # per image: the arrays' rows in keep order (valid up to the kept count)
    finish.keeps = keeps  # per image: the kept indices [n] on the device, valid up to the kept count (result row, word 0)
    # the batched pipeline wrote every image's {kept, flags, finite, 0} row into result_buffer: a device-side consumer
    # may read the counts there (the per-image fallback of nms_images writes its results elsewhere)
    finish.result_on_device = result_buffer is not None
    return finish if defer else finish()


def nms_images(inputs, iou_threshold, rotated=False, defer=False, runs=None, gather=None, result_buffer=None,
               host_mirror=True):
    """NMS of every image of a batch in one call: `inputs` = [(boxes, scores, idxs | None), ...].
    runs = (run_offsets, runs_are_categories[, num_categories]), the same for every image: the rows are pre-sorted runs (per-level
    top-k lists; d2amd_nms_runs) -- with runs_are_categories the idxs of `inputs` are ignored (pass None).
    gather (with runs and defer): per image, up to 4 tensors [n, ...] whose kept rows are wanted in keep order; the
    returned callable carries them as `.gathered` (per image, full length: valid up to the kept count).
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
        return _nms_images_batched(inputs, iou_threshold, rotated, defer, runs, gather, result_buffer, host_mirror)
    extra_tail = None if result_buffer is None else result_buffer[8 * len(inputs):]  # (large inputs: separate transfers)
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
    assert gather is None or runs is not None, "gather: only with pre-sorted runs (d2amd_nms*_runs)"
    gathered = [_gather_arg(g) for g in gather] if gather is not None else []  # (allocated on `cur` before the fork)
    fork = torch.cuda.Event()
    fork.record(cur)  # fork point: after the conversions, before any NMS kernel of this call
    launched = []
    k = 0
    for img, (boxes, scores, idxs) in enumerate(inputs):
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
        res = _nms_launch(boxes, scores, None if (runs is not None and runs[1]) else idxs, iou_threshold, rotated,
                          stream_ptr=_C.ctypes.c_void_p(st.cuda_stream), runs=runs,
                          gather=gathered[img][0] if gathered else None)
        if st is not cur:  # allocated on `cur`'s pool, used on `st`: tell the caching allocator
            for t in (res[0], res[1]) + tuple(x for x in res[2] if x is not None):
                t.record_stream(st)
            if gathered:
                for t in gathered[img][1] + gathered[img][2]:
                    t.record_stream(st)
        launched.append((res, st))
        k += 1
    for item in launched:
        if item is not None and item[1] is not cur:
            cur.wait_stream(item[1])  # join: everything enqueued later on `cur` sees the results
    live = [it for it in launched if it is not None]
    stacked = torch.stack([it[0][1] for it in live]) if live else None

    def finish(with_finite=False, extra=None):
        if extra_tail is not None:
            extra = extra_tail
        flat = stacked.flatten() if live else torch.zeros(0, dtype=torch.int64, device=dev)
        if extra is not None:
            flat = torch.cat([flat, extra.flatten().to(torch.int64)])
        vals = flat.tolist() if (live or extra is not None) else []  # ONE host sync
        out, fin, j = [], [], 0
        for img, ((boxes, _s, _i), it) in enumerate(zip(inputs, launched)):
            if it is None:
                out.append(torch.empty((0,), dtype=torch.int64, device=boxes.device))
                fin.append(0)
            else:
                c = vals[4 * j:4 * j + 4]
                keep = it[0][0]
                if (c[1] & 4) or ((c[1] & 1) and boxes.shape[0] > 16384):  # (see _nms_redo)
                    keep, c[0], c[1], c[2] = _nms_redo(boxes, _s, _i, iou_threshold, rotated, runs, c[1])
                    if gathered:
                        _regather(gathered[img][1], gathered[img][2], keep, c[0])
                out.append(_nms_finish(keep, c[0], c[1]))
                fin.append(c[2])
                j += 1
        if not with_finite and extra is None:
            return out
        return out, fin, vals[4 * len(live):]

    finish.gathered = [g[2] for g in gathered]
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
    if input.dtype == torch.float64:  # the reference's double instantiation (gradcheck): d2amd_roi_align_f64_*
        from .roi_align import f64_forward

        return f64_forward(input, rois, pooled_height, pooled_width, spatial_scale, sampling_ratio, True, True)
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
    if grad.dtype == torch.float64:
        from .roi_align import f64_backward

        return f64_backward(grad, rois, (batch_size, channels, height, width), pooled_height, pooled_width,
                            spatial_scale, sampling_ratio, True, True)
    layout = _C.NHWC if (grad.dim() == 4 and not grad.is_contiguous()
                         and grad.is_contiguous(memory_format=torch.channels_last)) else _C.NCHW
    g = grad.detach() if layout == _C.NHWC else grad.detach().contiguous()
    r = rois.detach().float().contiguous()
    shape = (batch_size, channels, height, width)
    gin = _empty_like_layout(g, shape, layout)
    if gin.numel() == 0:
        return gin
    if layout == _C.NHWC and pooled_height * pooled_width <= 1024:
        # channels_last: the fused pooler's backward with ONE level -- a deterministic gather instead of the fp32 atomic
        # scatter (csrc/roi_pool_rot.hip; the level rule is not evaluated for a single level)
        p = _C.PoolerParams()
        p.num_levels, p.N, p.C = 1, batch_size, channels
        p.H[0], p.W[0], p.spatial_scale[0] = height, width, float(spatial_scale)
        p.pooled_h, p.pooled_w, p.sampling_ratio, p.aligned = pooled_height, pooled_width, int(sampling_ratio), 1
        p.dtype, p.layout = _C.dtype_code(g), _C.NHWC
        p.min_level = p.max_level = p.canonical_level = 0
        p.canonical_box_size = 224.0
        L = _C.lib()
        with _C.on_device(g.device):
            ws_bytes = L.d2amd_roi_pooler_rotated_backward_workspace_bytes(_C.ctypes.byref(p), int(r.shape[0]))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=g.device)
            _C.check(L.d2amd_roi_pooler_rotated_backward(_C.ctypes.byref(p), _C.ptr(g), _C.ptr(r),
                                                         (_C.ctypes.c_void_p * 1)(gin.data_ptr()), int(r.shape[0]),
                                                         _C.ptr(ws), ws_bytes, _C.stream()))
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
