"""ROIPooler -- mirrors detectron2/modeling/poolers.py:23-263 (same constructor, same
`forward(x: list[Tensor], box_lists: list[Boxes])`, same level-assignment rule), but the multi-level
case is ONE fused HIP launch per direction (d2amd_roi_pooler_forward / _backward) instead of the
reference's per-level `nonzero` (host sync) -> ROIAlign -> `index_put_` loop:

  * level assignment runs inside the kernel, in fp32, operation for operation as
    `assign_boxes_to_levels` (poolers.py:51-59);
  * channels_last (NHWC) features: forward = flattened-tap gather, backward = atomic-free,
    deterministic tile gather that writes every grad element once in the I/O dtype;
  * NCHW features: fused NCHW forward; backward re-lays dY out as NHWC, runs the same tile gather
    and returns channels_last-strided gradients.
`pooler_type` "ROIAlignRotated" is fused for channels_last features (csrc/roi_pool_rot.hip: one launch per direction), and
keeps the per-level loop otherwise; "ROIPool" is not part of the hot path.

CHAINED BACKWARD.  Mask R-CNN pools the same FPN features twice per iteration (box head 7x7, mask head 14x14:
roi_heads.py:780-846), so autograd sums two dense gradients per level with an elementwise kernel (r01: 4 launches,
274 MB of extra traffic per step).  Here the fused pooler also returns its feature inputs as (alias) outputs and
remembers them; a later pooler call on the SAME feature tensors (same objects, unmodified) pools from the aliases.
In the backward pass the later pooler hands its work to the first one (whose node is its autograd parent through the
aliases) instead of launching; the first pooler's backward writes its own gradient and lets the deferred tile
gathers ADD to it (d2amd_roi_pooler_backward_accumulate): no separate sum, empty tiles untouched, values = autograd's
sum.  A pooler whose result is unused contributes nothing, as in plain autograd.
"""
import ctypes
import os
import math
from typing import List

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from ..layers.roi_align import ROIAlign, _layout_of
from ..layers.roi_align_rotated import ROIAlignRotated
from ..layers.wrappers import disable_torch_compiler

__all__ = ["ROIPooler", "assign_boxes_to_levels", "convert_boxes_to_pooler_format"]


def _areas(box_lists):
    return torch.cat([b.area() if hasattr(b, "area") else (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
                      for b in box_lists])


def assign_boxes_to_levels(box_lists, min_level: int, max_level: int, canonical_box_size: int,
                           canonical_level: int):
    """Same contract as poolers.py:23-59: int64 level offsets (from `min_level`) of all boxes."""
    box_sizes = torch.sqrt(_areas(box_lists))
    level_assignments = torch.floor(canonical_level + torch.log2(box_sizes / canonical_box_size + 1e-8))
    level_assignments = torch.clamp(level_assignments, min=min_level, max=max_level)
    return level_assignments.to(torch.int64) - min_level


def convert_boxes_to_pooler_format(box_lists):
    """(M, 5) [batch index, x0, y0, x1, y1] (or (M, 6) for rotated boxes), poolers.py:74-104.
    On HIP tensors this is one torch.cat plus one kernel launch and -- unlike the reference's
    torch.repeat_interleave -- never synchronises with the host."""
    tensors = [b.tensor if hasattr(b, "tensor") else b for b in box_lists]
    boxes = torch.cat(tensors, dim=0) if len(tensors) != 1 else tensors[0]
    n_img, width = len(tensors), boxes.shape[1]
    if boxes.is_cuda and boxes.dtype == torch.float32 and n_img <= 64 and width in (4, 5):
        boxes = boxes.detach().contiguous()
        rois = torch.empty((boxes.shape[0], width + 1), dtype=torch.float32, device=boxes.device)
        counts = (ctypes.c_int * n_img)(*[int(t.shape[0]) for t in tensors])
        with _C.on_device(boxes.device):
            _C.check(_C.lib().d2amd_boxes_to_rois(_C.ptr(boxes), counts, n_img, width, _C.ptr(rois), _C.stream()))
        return rois
    sizes = torch.tensor([len(t) for t in tensors], device=boxes.device)
    indices = torch.repeat_interleave(torch.arange(len(sizes), dtype=boxes.dtype, device=boxes.device), sizes)
    return torch.cat([indices[:, None], boxes], dim=1)


_PARAMS_CACHE = {}


def _params(cfg, feats_shape, hw, dtype_code, layout):
    key = (cfg, feats_shape, tuple(hw), dtype_code, layout, _C.reference_roi_rounding_on())
    p = _PARAMS_CACHE.get(key)
    if p is not None:
        return p
    p = _PARAMS_CACHE[key] = _build_params(cfg, feats_shape, hw, dtype_code, layout)
    if len(_PARAMS_CACHE) > 256:
        _PARAMS_CACHE.clear()
    return p


def _build_params(cfg, feats_shape, hw, dtype_code, layout):
    out_hw, scales, sr, aligned, min_level, max_level, canon_size, canon_level = cfg
    p = _C.PoolerParams()
    p.num_levels = len(scales)
    p.N, p.C = feats_shape
    for l, ((h, w), s) in enumerate(zip(hw, scales)):
        p.H[l], p.W[l], p.spatial_scale[l] = h, w, s
    p.pooled_h, p.pooled_w = out_hw
    p.sampling_ratio, p.aligned = sr, int(aligned)
    p.dtype, p.layout = dtype_code, layout
    p.min_level, p.max_level, p.canonical_level = min_level, max_level, canon_level
    p.canonical_box_size = float(canon_size)
    p.roi_rounding = int(_C.reference_roi_rounding_on())  # (strict reference parity: _C.set_reference_roi_rounding)
    return p


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _transpose(src, dst, batch, rows, cols):
    with _C.on_device(src.device):
        _C.check(_C.lib().d2amd_transpose_batched(_C.ptr(src), _C.ptr(dst), batch, rows, cols, src.element_size(),
                                                  _C.stream()))
    return dst


def _to_nhwc(t):
    """The channels_last twin of a 4-D tensor.  NCHW-contiguous 2- / 4-byte tensors go through the library's tiled
    transpose (d2amd_transpose_batched); torch's permuting copy is the fallback for anything else."""
    if t.dim() != 4 or t.is_contiguous(memory_format=torch.channels_last):
        return t if t.dim() != 4 else t
    n, c, h, w = t.shape
    if t.is_cuda and t.is_contiguous() and t.element_size() in (2, 4) and t.numel() and not t.requires_grad:
        return _transpose(t, torch.empty_like(t, memory_format=torch.channels_last), n, c, h * w)
    return t.contiguous(memory_format=torch.channels_last)


def _to_nchw(t):
    """The NCHW-contiguous twin of a 4-D tensor (inverse of _to_nhwc)."""
    if t.dim() != 4 or t.is_contiguous():
        return t
    n, c, h, w = t.shape
    if (t.is_cuda and t.is_contiguous(memory_format=torch.channels_last) and t.element_size() in (2, 4) and t.numel()
            and not t.requires_grad):
        return _transpose(t, torch.empty(t.shape, dtype=t.dtype, device=t.device), n, h * w, c)
    return t.contiguous()


def _transpose_many(pairs, nhwc_to_nchw: bool):
    """[(src, dst)] 4-D tensors of one batch size / dtype / device, NCHW -> NHWC (or back) in ONE launch
    (d2amd_transpose_multi, <= 8 tensors per launch)."""
    for i in range(0, len(pairs), 8):
        chunk = pairs[i:i + 8]
        k = len(chunk)
        srcs, dsts = (ctypes.c_void_p * k)(), (ctypes.c_void_p * k)()
        rows, cols = (ctypes.c_int * k)(), (ctypes.c_int * k)()
        for j, (a, b) in enumerate(chunk):
            n, c, h, w = a.shape
            srcs[j], dsts[j] = a.data_ptr(), b.data_ptr()
            rows[j], cols[j] = (h * w, c) if nhwc_to_nchw else (c, h * w)
        a0 = chunk[0][0]
        with _C.on_device(a0.device):
            _C.check(_C.lib().d2amd_transpose_multi(srcs, dsts, rows, cols, k, int(a0.shape[0]), a0.element_size(), _C.stream()))


def _batchable(ts):
    t0 = ts[0]
    return all(t.dim() == 4 and t.is_cuda and t.dtype == t0.dtype and t.device == t0.device and t.shape[0] == t0.shape[0]
               and t.element_size() in (2, 4) and t.numel() and not t.requires_grad for t in ts)


def _to_nchw_many(ts):
    """_to_nchw of every tensor of the list; the channels_last ones of a batchable list in one launch."""
    todo = [i for i, t in enumerate(ts) if t is not None and t.dim() == 4 and not t.is_contiguous()
            and t.is_contiguous(memory_format=torch.channels_last)]
    if len(todo) < 2 or not _batchable([ts[i] for i in todo]):
        return [None if t is None else _to_nchw(t) for t in ts]
    out = [None if t is None else t for t in ts]
    pairs = []
    for i in todo:
        out[i] = torch.empty(ts[i].shape, dtype=ts[i].dtype, device=ts[i].device)
        pairs.append((ts[i], out[i]))
    _transpose_many(pairs, True)
    return [None if t is None else (t if i in todo else _to_nchw(t)) for i, t in enumerate(out)]


_NHWC_CACHE = {}  # id(feature tensor) -> (weakref, version, channels_last copy)


def _staged_nhwc_many(feats):
    """_staged_nhwc of every feature map; the copies that are not cached yet are made by ONE launch."""
    import weakref
    bases = [f._base if f._base is not None and f._base.shape == f.shape and f._base.stride() == f.stride() else f for f in feats]
    hit = []
    for b in bases:
        ent = _NHWC_CACHE.get(id(b))
        hit.append(ent[2] if ent is not None and ent[0]() is b and ent[1] == b._version else None)
    miss = [i for i, h in enumerate(hit) if h is None]
    srcs = [feats[i].detach() for i in miss]
    if len(miss) < 2 or not (_batchable(srcs) and all(t.is_contiguous() for t in srcs)):
        return [_staged_nhwc(f) for f in feats]
    pairs = [(t, torch.empty_like(t, memory_format=torch.channels_last)) for t in srcs]
    _transpose_many(pairs, False)
    for k in [k for k, e in _NHWC_CACHE.items() if e[0]() is None]:
        del _NHWC_CACHE[k]
    if len(_NHWC_CACHE) + len(miss) > 16:
        _NHWC_CACHE.clear()
    for i, (_, cl) in zip(miss, pairs):
        _NHWC_CACHE[id(bases[i])] = (weakref.ref(bases[i]), bases[i]._version, cl)
        hit[i] = cl
    return hit


def _staged_nhwc(f):
    """channels_last staging copy of an NCHW feature map, reused while the same (unmodified) tensor object is
    pooled again -- the box head and the mask head pool the same FPN features in one iteration (the second call may
    see it as the alias view of the chained backward: the cache is keyed on the view's base)."""
    import weakref
    base = f._base if f._base is not None and f._base.shape == f.shape and f._base.stride() == f.stride() else f
    ent = _NHWC_CACHE.get(id(base))
    if ent is not None and ent[0]() is base and ent[1] == base._version:
        return ent[2]
    cl = _to_nhwc(f.detach())
    for k in [k for k, e in _NHWC_CACHE.items() if e[0]() is None]:  # copies of features that no longer exist
        del _NHWC_CACHE[k]
    if len(_NHWC_CACHE) >= 16:
        _NHWC_CACHE.clear()
    _NHWC_CACHE[id(base)] = (weakref.ref(base), base._version, cl)
    return cl


import threading as _threading

# the tables below are shared by the forward (caller's thread) and the backward (autograd engine threads; several of
# them under DataParallel / multi-threaded backward): every access holds this lock
_TABLE_LOCK = _threading.RLock()
_ALIASES = {}  # id(feature tensor) -> (weakref to it, its version, alias output of the last pooler that took it, token)
_ALIAS_CAP = 16


_DEFERRED = {}     # token of a pooler call -> tile-gather work handed to it by the later poolers of its chain
_PLACEHOLDER = {}  # (dtype, device) -> zero scalar whose zero-stride expansions stand in for deferred gradients


def _placeholder(shape, dtype, device):
    z = _PLACEHOLDER.get((dtype, device))
    if z is None:
        z = _PLACEHOLDER[(dtype, device)] = torch.zeros((), dtype=dtype, device=device)
    return z.expand(shape)


def _is_placeholder(t):
    z = _PLACEHOLDER.get((t.dtype, t.device))
    return z is not None and t.data_ptr() == z.data_ptr() and all(st == 0 for st in t.stride())


def _chained_inputs(x):
    """(tensors to pool from, token of the pooler that produced them): the remembered alias outputs of an earlier pooler
    call on exactly these feature tensors, or (x itself, None)."""
    out, tokens = [], set()
    with _TABLE_LOCK:
        for f in x:
            ent = _ALIASES.get(id(f))
            if ent is None or ent[0]() is not f or ent[1] != f._version:
                return list(x), None
            out.append(ent[2])
            tokens.add(id(ent[3]))
        if len(tokens) != 1:
            return list(x), None
        return out, _ALIASES[id(x[0])][3]


def _remember_aliases(x, aliases, token):
    import weakref
    with _TABLE_LOCK:
        for k in [k for k, e in _ALIASES.items() if e[0]() is None]:  # features that no longer exist
            del _ALIASES[k]
        if len(_ALIASES) + len(x) > _ALIAS_CAP:
            _ALIASES.clear()
        for f, a in zip(x, aliases):
            _ALIASES[id(f)] = (weakref.ref(f), f._version, a, token)


def _forget_aliases(token):
    """The node that produced these aliases has run its backward: nothing may chain onto it any more."""
    with _TABLE_LOCK:
        for k in [k for k, e in _ALIASES.items() if e[3] is token]:
            del _ALIASES[k]


import os as _os

# Where the binning of the backward runs: "none" = in the backward, in front of each tile gather; "chained" / "all" =
# beside the forward of the chained / of every pooler (d2amd_roi_pooler_backward_phase: it depends on the ROIs alone).
# Measured inside the captured step (bench.py, same box, ms/step): none 0.524 / 0.526, chained 0.530 / 0.523, all 0.538 /
# 0.534 -- on one stream the small binning kernels cost the same wherever they sit, and the writing variant adds a
# zero-fill launch; the early binning only pays for a caller that puts the forward on its own stream.  Default: none.
_PREBIN_MODE = "none"  # (module attribute, no environment name: tests/test_gpu_pooler.py sets it)
# (Measured and REMOVED in r06 -- the switches had no test and no caller: the forward in a spatial PROCESSING ORDER
# (d2amd_roi_pooler_forward_ordered stays in the C ABI with its test: the box head's HBM fetch 1.7 x -> 1.07 x the features
# at equal kernel time, but 10-15 us of ordering launch on the critical path: 0.521 / 0.527 ms per step with it, 0.498 /
# 0.500 without); the side binnings joined in front of the first gather; the later gathers' binning NOT beside the first.)
_ROT_POOLER_LOOP = False  # (module attribute: the rotated pooler level by level -- the reference's structure, scripts/rrpn_ab.sh)
# The first two gathers of a chain in ONE pass over the tiles (d2amd_roi_pooler_backward_pair: box head 7x7 + mask head
# 14x14); D2AMD_POOL_PAIR=0 (read once, at import): one launch per pooler, the second one adding (the A/B and the
# bit-for-bit autograd sum; tests/test_gpu_pooler_pair.py sets the attribute).
_PAIR = _os.environ.get("D2AMD_POOL_PAIR", "1") != "0"
# The paired forward also writes the paired backward's per-ROI records and resets its work queues (r06:
# d2amd_roi_pooler_forward_pair_records): module attribute, tests/test_gpu_pooler_pair.py compares both settings bit for bit.
_FWD_RECORDS = _os.environ.get("D2AMD_POOL_FWD_RECORDS", "1") != "0"  # (read once, at import: the A/B)


def _PREBIN(head):
    return _PREBIN_MODE in ("all", "side") or (_PREBIN_MODE == "chained" and not head)


class _FusedROIPool(Function):
    @staticmethod
    @disable_torch_compiler
    def forward(ctx, rois, cfg, chain, head, upstream, *feats):
        # chain: None, or the token this call's aliases are remembered under; head: the inputs are the caller's
        # tensors; upstream: (not head) the token of the pooler whose aliases are this call's inputs
        # rois: the (M, 5) pooler-format tensor, or a tuple of per-image (n_i, 4) fp32 HIP box tensors -- then the
        # conversion happens inside the same C call (d2amd_roi_pooler_forward_box_lists: no torch.cat, one call less)
        box_lists = None
        # (strict-reference mode, _C.set_reference_roi_rounding: the kernels round the coordinates to the feature dtype
        # BEHIND the level assignment -- p.roi_rounding -- as ROIPooler.forward assigns levels from the fp32 boxes and
        # each level's ROIAlign casts its ROIs, poolers.py:240-262 / roi_align.py:60)
        if isinstance(rois, tuple):
            box_lists = rois
            rois = torch.empty((sum(int(b.shape[0]) for b in box_lists), 5), dtype=torch.float32,
                               device=feats[0].device)
        _C.require_gpu(rois, *feats, op="ROIPooler")
        layout = _layout_of(feats[0])
        nchw_in = layout == _C.NCHW and feats[0].dtype != torch.float32
        if nchw_in:
            # The NHWC kernel reads a tap as contiguous channels (fully coalesced); the NCHW kernel cannot, and is
            # 5x slower (profiles/r01).  For 16-bit NCHW features the forward therefore stages a channels_last
            # copy (one pass over the features, shared by the poolers of an iteration) and returns an NCHW
            # result, like the backward already does for the gradients.  fp32 keeps the dedicated NCHW kernel.
            feats_used, layout = _staged_nhwc_many(list(feats)), _C.NHWC
        else:
            feats_used = feats
        xs = [f if layout == _C.NHWC else f.contiguous() for f in feats_used]
        n, c = xs[0].shape[:2]
        hw = [tuple(x.shape[2:]) for x in xs]
        k = rois.shape[0]
        p = _params(cfg, (n, c), hw, _C.dtype_code(xs[0]), layout)
        ph, pw = cfg[0]
        mf = torch.channels_last if layout == _C.NHWC else torch.contiguous_format
        out = torch.empty((k, c, ph, pw), dtype=xs[0].dtype, device=xs[0].device, memory_format=mf)
        with _C.on_device(xs[0].device):
            if box_lists is None:
                _C.check(_C.lib().d2amd_roi_pooler_forward(ctypes.byref(p), _ptr_array(xs), _C.ptr(rois), _C.ptr(out), k,
                                                           _C.stream()))
            else:
                n_img = len(box_lists)
                counts = (ctypes.c_int * n_img)(*[int(b.shape[0]) for b in box_lists])
                _C.check(_C.lib().d2amd_roi_pooler_forward_box_lists(
                    ctypes.byref(p), _ptr_array(xs), _ptr_array(box_lists), counts, n_img, _C.ptr(rois), _C.ptr(out),
                    _C.stream()))
        ctx.save_for_backward(rois)
        ctx.cfg, ctx.hw, ctx.nc, ctx.layout = cfg, hw, (n, c), layout
        ctx.needs = [f.requires_grad for f in feats]
        # The binning of the backward (per-ROI records, per-tile ROI lists, work queues) depends on the ROIs alone: it
        # CAN be done here, beside the forward kernels (on whatever stream the forward runs), instead of in front of
        # the tile gathers of the backward; the gather that uses it later either adds (phase 2) or writes (phase 3: +
        # a zero fill of the untouched tiles).  Off by default: see _PREBIN_MODE.
        ctx.binned = None
        if layout == _C.NHWC and k > 0 and any(ctx.needs_input_grad[5:]) and _PREBIN(head):  # (False under no_grad)
            L = _C.lib()
            ws_bytes = L.d2amd_roi_pooler_backward_workspace_bytes(ctypes.byref(p), k)
            side = None
            if _PREBIN_MODE == "side":  # on a stream of its own, forked here, joined in front of the tile gather
                from ..streams import _streams

                dev = xs[0].device
                side = _streams(dev, 2, late=True)[1]  # (late stream 0: a caller's deferred fork_join branch)
                side.wait_stream(torch.cuda.current_stream(dev))
                rois.record_stream(side)
            with _C.on_device(xs[0].device):  # (pointers: only their alignment class matters to the binning)
                if side is None:
                    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xs[0].device)
                    rc = L.d2amd_roi_pooler_backward_phase(ctypes.byref(p), _C.ptr(out), _C.ptr(rois), _ptr_array(xs),
                                                           k, _C.ptr(ws), ws_bytes, 1, _C.stream())
                else:
                    with torch.cuda.stream(side):
                        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                        rc = L.d2amd_roi_pooler_backward_phase(ctypes.byref(p), _C.ptr(out), _C.ptr(rois),
                                                               _ptr_array(xs), k, _C.ptr(ws), ws_bytes, 1, _C.stream())
            if rc == 0:
                ctx.binned = (ws, ws_bytes) if side is None else (ws, ws_bytes, side)
            elif rc != _C.EUNSUPPORTED:
                _C.check(rc)
        ctx.chain, ctx.head, ctx.upstream, ctx.dtype = chain, head, upstream, xs[0].dtype
        ctx.set_materialize_grads(False)  # unused outputs (the aliases of the last pooler of a chain) arrive as None
        ctx.nchw_caller = _layout_of(feats[0]) == _C.NCHW  # gradients go back in the caller's layout
        if nchw_in:
            out = _to_nchw(out)  # NCHW-contiguous result, as the caller's layout implies
        if chain is not None:
            # the features come back as outputs: autograd turns a returned input into a view whose gradient arrives
            # in backward() below -- the hook a later pooler of the same features chains onto (module docstring)
            return (out,) + tuple(feats)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output, *held):
        """A chain of poolers over the same features runs ONE sequence of tile gathers, in the head's backward (the head
        is the pooler that took the caller's tensors; it runs last): a later pooler of the chain does not launch --
        it hands its work (dY, rois, configuration) to the pooler whose aliases it consumed and returns zero-stride
        placeholders as the aliases' gradients.  The head then writes its own gradient plainly (every tile once,
        empty tiles zero-filled) and the deferred ones ADD to it (d2amd_roi_pooler_backward_accumulate: tiles no ROI
        touches are neither read nor written).  The poolers with more ROIs come first in Mask R-CNN (box head, then
        mask head), so the accumulating launches are the small ones."""
        (rois,) = ctx.saved_tensors
        cfg, hw, (n, c) = ctx.cfg, ctx.hw, ctx.nc
        works = []
        if ctx.chain is not None:
            _forget_aliases(ctx.chain)
            with _TABLE_LOCK:
                works = _DEFERRED.pop(ctx.chain, [])
        if grad_output is not None:
            works.insert(0, (_to_nhwc(grad_output.detach()), rois, cfg, ctx.binned))  # own work first: the plain write
        real = [h for h in held if h is not None and not _is_placeholder(h)]  # a foreign consumer of the aliases
        if not ctx.head:
            if works:
                with _TABLE_LOCK:
                    _DEFERRED.setdefault(ctx.upstream, []).extend(works)
            if real:  # keep real gradients flowing; the deferred work is added upstream
                return (None, None, None, None, None) + tuple(h if need else None for h, need in zip(held, ctx.needs))
            ph = [(_placeholder((n, c) + tuple(s), ctx.dtype, rois.device) if works else None) for s in hw]
            return (None, None, None, None, None) + tuple(g if need else None for g, need in zip(ph, ctx.needs))
        # ---- head: launch everything
        back = _to_nchw if ctx.nchw_caller else (lambda t: t)
        grads = None
        if real:
            grads = [h if (h is not None and not _is_placeholder(h)) else None for h in held]
            # the tile gathers below ADD in place: never into the tensors autograd handed over (other nodes may hold
            # the same buffers) -- a channels_last gradient is cloned, any other layout is copied by the conversion
            grads = [(torch.zeros((n, c) + tuple(s), dtype=ctx.dtype, device=rois.device,
                                  memory_format=torch.channels_last) if g is None else
                      g.clone(memory_format=torch.channels_last) if g.is_contiguous(memory_format=torch.channels_last)
                      else _to_nhwc(g))
                     for g, s in zip(grads, hw)]
        L = _C.lib()
        dev = rois.device
        plain_first = grads is None and bool(works)  # the first work writes every tile; everything else adds
        if plain_first:
            g0 = works[0][0]
            grads = [torch.empty((n, c, h, w), dtype=g0.dtype, device=dev, memory_format=torch.channels_last)
                     for (h, w) in hw]
        with _C.on_device(dev):
            if (plain_first and _PAIR and len(works) >= 2 and works[0][3] is None and works[1][3] is None
                    and works[0][1].shape[0] > 0 and works[1][1].shape[0] > 0 and works[0][0].dtype == works[1][0].dtype):
                (g1, r1, c1, _), (g2, r2, c2, _) = works[0], works[1]
                p1 = _params(c1, (n, c), hw, _C.dtype_code(g1), _C.NHWC)
                p2 = _params(c2, (n, c), hw, _C.dtype_code(g2), _C.NHWC)
                wsb = L.d2amd_roi_pooler_backward_pair_workspace_bytes(ctypes.byref(p1), r1.shape[0], r2.shape[0])
                ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
                rc = L.d2amd_roi_pooler_backward_pair(ctypes.byref(p1), _C.ptr(g1), _C.ptr(r1), r1.shape[0],
                                                      ctypes.byref(p2), _C.ptr(g2), _C.ptr(r2), r2.shape[0],
                                                      _ptr_array(grads), _C.ptr(ws), wsb, _C.stream())
                if rc == 0:  # both are in `grads`: whatever follows adds
                    works, plain_first = works[2:], False
                elif rc != _C.EUNSUPPORTED:
                    _C.check(rc)
            # The binning of the chain's LATER gathers (records, per-tile lists, queues: ~20 us of small launches each,
            # a function of the ROIs alone) runs on a side stream beside the first gather instead of between the gathers.
            if len(works) > 1 and grads is not None:
                from ..streams import _streams

                cur = torch.cuda.current_stream(dev)
                side = _streams(dev, 1, late=True)[0]
                forked = False
                for j in range(1, len(works)):
                    g, r, wcfg, binned = works[j]
                    if binned is not None or r.shape[0] == 0:
                        continue
                    if not forked:
                        side.wait_stream(cur)
                        forked = True
                    p = _params(wcfg, (n, c), hw, _C.dtype_code(g), _C.NHWC)
                    with torch.cuda.stream(side):
                        ws_bytes = L.d2amd_roi_pooler_backward_workspace_bytes(ctypes.byref(p), r.shape[0])
                        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                        rc = L.d2amd_roi_pooler_backward_phase(ctypes.byref(p), _C.ptr(g), _C.ptr(r), _ptr_array(grads),
                                                               r.shape[0], _C.ptr(ws), ws_bytes, 1, _C.stream())
                    if rc == 0:
                        works[j] = (g, r, wcfg, (ws, ws_bytes, side))
                    elif rc != _C.EUNSUPPORTED:
                        _C.check(rc)
            for j, (g, r, wcfg, binned) in enumerate(works):
                k = r.shape[0]
                p = _params(wcfg, (n, c), hw, _C.dtype_code(g), _C.NHWC)
                if binned is not None and len(binned) == 3:  # binned on the side stream above: join it here
                    torch.cuda.current_stream(dev).wait_stream(binned[2])
                    binned = binned[:2]
                if plain_first and j == 0:
                    if binned is not None:  # binned beside its forward: zero fill of the untouched tiles + gather
                        ws, ws_bytes = binned
                        ws.record_stream(torch.cuda.current_stream(dev))
                        _C.check(L.d2amd_roi_pooler_backward_phase(ctypes.byref(p), _C.ptr(g), _C.ptr(r),
                                                                   _ptr_array(grads), k, _C.ptr(ws), ws_bytes, 3,
                                                                   _C.stream()))
                        continue
                    # per-ROI records + per-tile ROI lists (one wave per 8x8 tile bins the ROIs once per call)
                    ws_bytes = L.d2amd_roi_pooler_backward_workspace_bytes(ctypes.byref(p), k)
                    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                    _C.check(L.d2amd_roi_pooler_backward(ctypes.byref(p), _C.ptr(g), _C.ptr(r), _ptr_array(grads), k,
                                                         _C.ptr(ws), ws_bytes, _C.stream()))
                    continue
                if binned is not None:  # binned beside its forward (maybe on another stream): gather only
                    ws, ws_bytes = binned
                    ws.record_stream(torch.cuda.current_stream(dev))
                    rc = L.d2amd_roi_pooler_backward_phase(ctypes.byref(p), _C.ptr(g), _C.ptr(r), _ptr_array(grads), k,
                                                           _C.ptr(ws), ws_bytes, 2, _C.stream())
                else:
                    ws_bytes = L.d2amd_roi_pooler_backward_workspace_bytes(ctypes.byref(p), k)
                    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                    rc = L.d2amd_roi_pooler_backward_accumulate(ctypes.byref(p), _C.ptr(g), _C.ptr(r),
                                                                _ptr_array(grads), k, _C.ptr(ws), ws_bytes, _C.stream())
                if rc == _C.EUNSUPPORTED:  # configuration outside the staged tile gather: fresh buffers + a sum
                    extra = [torch.empty_like(t) for t in grads]
                    _C.check(L.d2amd_roi_pooler_backward(ctypes.byref(p), _C.ptr(g), _C.ptr(r), _ptr_array(extra), k,
                                                         _C.ptr(ws), ws_bytes, _C.stream()))
                    grads = [a + b for a, b in zip(grads, extra)]
                else:
                    _C.check(rc)
        if grads is None:
            return (None, None, None, None, None) + (None,) * len(hw)
        if ctx.nchw_caller:  # (all levels' gradients back to the caller's layout in one launch)
            res = _to_nchw_many([g if need else None for g, need in zip(grads, ctx.needs)])
            return (None, None, None, None, None) + tuple(res)
        return (None, None, None, None, None) + tuple(back(g) if need else None for g, need in zip(grads, ctx.needs))


class _FusedRotatedPool(Function):
    """Multi-level ROIAlignRotated in one launch per direction (csrc/roi_pool_rot.hip).  rois: (M, 6) fp32 pooler-format
    rotated boxes (image index, cx, cy, w, h, angle); features NHWC (channels_last) of one dtype."""

    @staticmethod
    @disable_torch_compiler
    def forward(ctx, rois, cfg, *feats):
        _C.require_gpu(rois, *feats, op="ROIPooler (ROIAlignRotated)")
        xs = list(feats)
        n, c = xs[0].shape[:2]
        hw = [tuple(x.shape[2:]) for x in xs]
        k = rois.shape[0]
        p = _params(cfg, (n, c), hw, _C.dtype_code(xs[0]), _C.NHWC)
        ph, pw = cfg[0]
        out = torch.empty((k, c, ph, pw), dtype=xs[0].dtype, device=xs[0].device, memory_format=torch.channels_last)
        status = torch.zeros(1, dtype=torch.int32, device=xs[0].device)
        if k > 0:
            with _C.on_device(xs[0].device):
                _C.check(_C.lib().d2amd_roi_pooler_rotated_forward(ctypes.byref(p), _ptr_array(xs), _C.ptr(rois),
                                                                   _C.ptr(out), k, _C.ptr(status), _C.stream()))
            # ROIAlignRotated_cpu.cpp:236-238 asserts non-negative sizes; reading the status word is one host sync, as the
            # per-level op's (layers/ops.py) -- the reference's device forward ends in a device synchronisation
            if int(status.item()) != 0:
                raise RuntimeError("ROIs in ROIAlignRotated do not have non-negative size!")
        ctx.save_for_backward(rois)
        ctx.cfg, ctx.hw, ctx.nc, ctx.dtype = cfg, hw, (n, c), xs[0].dtype
        ctx.needs = [f.requires_grad for f in feats]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        cfg, hw, (n, c) = ctx.cfg, ctx.hw, ctx.nc
        g = _to_nhwc(grad_output.detach())
        dev = rois.device
        grads = [torch.empty((n, c, h, w), dtype=ctx.dtype, device=dev, memory_format=torch.channels_last) for (h, w) in hw]
        p = _params(cfg, (n, c), hw, _C.dtype_code(g), _C.NHWC)
        L = _C.lib()
        with _C.on_device(dev):
            ws_bytes = L.d2amd_roi_pooler_rotated_backward_workspace_bytes(ctypes.byref(p), int(rois.shape[0]))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _C.check(L.d2amd_roi_pooler_rotated_backward(ctypes.byref(p), _C.ptr(g), _C.ptr(rois), _ptr_array(grads),
                                                         rois.shape[0], _C.ptr(ws), ws_bytes, _C.stream()))
        return (None, None) + tuple(gr if need else None for gr, need in zip(grads, ctx.needs))


class PairBackwardPlan:
    """The binning of a paired pooler backward (`pool_pair_rois(..., plan=...)`), issued AHEAD of the backward: it depends on
    the ROIs alone (per-ROI records, per-tile lists, work queues, zero fill of the gradient tiles no ROI touches), ~25 us of
    small launches that otherwise sit between the forward and the tile gather.  `prepare()` may be called on ANY stream
    once both ROI sets exist -- e.g. on the branch that builds the mask targets beside the poolers' forward -- and
    allocates the feature gradients it zero-fills; the backward then waits for its event and runs the gather alone
    (d2amd_roi_pooler_backward_pair_phase 1 / 2).  A plan that was never prepared, or whose call the library declines,
    changes nothing: the backward bins as usual."""

    def __init__(self):
        self.ready = None  # (event, ws, ws_bytes, grads, signature)

    def prepare(self, pooler1: "ROIPooler", pooler2: "ROIPooler", x: List[torch.Tensor], rois1: torch.Tensor,
                rois2: torch.Tensor):
        self.ready = None
        if not (_PAIR and len(x) > 0 and pooler1._fusable(x) and pooler2._fusable(x) and _layout_of(x[0]) == _C.NHWC
                and x[0].dtype in (torch.bfloat16, torch.float16) and rois1.shape[0] > 0 and rois2.shape[0] > 0):
            return False
        cfgs = [_pair_cfg(p) for p in (pooler1, pooler2)]
        n, c = x[0].shape[:2]
        hw = [tuple(t.shape[2:]) for t in x]
        dev, dt = x[0].device, x[0].dtype
        code = _C.dtype_code(x[0])
        p1, p2 = _params(cfgs[0], (n, c), hw, code, _C.NHWC), _params(cfgs[1], (n, c), hw, code, _C.NHWC)
        L = _C.lib()
        k1, k2 = int(rois1.shape[0]), int(rois2.shape[0])
        grads = [torch.empty((n, c, h, w), dtype=dt, device=dev, memory_format=torch.channels_last) for (h, w) in hw]
        wsb = L.d2amd_roi_pooler_backward_pair_workspace_bytes(ctypes.byref(p1), k1, k2)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        with _C.on_device(dev):
            # (no gradient exists yet: the workspace stands in for both dY pointers -- only their alignment is looked at)
            rc = L.d2amd_roi_pooler_backward_pair_phase(ctypes.byref(p1), _C.ptr(ws), _C.ptr(rois1), k1, ctypes.byref(p2),
                                                        _C.ptr(ws), _C.ptr(rois2), k2, _ptr_array(grads), _C.ptr(ws), wsb, 1,
                                                        _C.stream())
        if rc == _C.EUNSUPPORTED:
            return False
        _C.check(rc)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        # the plan KEEPS the ROI tensors (an address alone can be reused by the caching allocator after a skipped backward)
        # and their versions (an in-place edit between prepare() and the backward makes the binning stale)
        self.ready = (ev, ws, wsb, grads, (rois1, rois2, rois1._version, rois2._version, k1, k2, cfgs[0], cfgs[1], (n, c),
                                           tuple(hw), dt))
        return True

    def matches(self, ready, rois1, rois2, cfg1, cfg2, nc, hw, dt):
        r1, r2, v1, v2, k1, k2, c1, c2, nc0, hw0, dt0 = ready[4]
        same = lambda a, b, v: a.data_ptr() == b.data_ptr() and a.shape == b.shape and b._version == v and a.device == b.device
        return (same(rois1, r1, v1) and same(rois2, r2, v2) and (c1, c2, nc0, hw0, dt0) == (cfg1, cfg2, nc, tuple(hw), dt)
                and r1._version == v1 and r2._version == v2)


def _pair_cfg(p):
    return (tuple(p.output_size), tuple(p.scales), int(p.sampling_ratio), p.pooler_type == "ROIAlignV2", p.min_level,
            p.max_level, p.canonical_box_size, p.canonical_level)


class _FusedROIPoolPair(Function):
    """Two poolers of the same NHWC feature maps, one launch per direction (d2amd_roi_pooler_forward_pair /
    d2amd_roi_pooler_backward_pair): see `pool_pair`."""

    @staticmethod
    @disable_torch_compiler
    def forward(ctx, rois1, rois2, cfg1, cfg2, plan, *feats):
        # roisN: the (M, 5) pooler-format tensor, or a tuple of per-image (n_i, 4) fp32 HIP box tensors (converted inside
        # the same C call, both lists by one launch); plan: None or a PairBackwardPlan (looked at in the backward)
        ctx.plan = plan
        ctx.prep = None
        lists = None
        if isinstance(rois1, tuple):
            lists = (rois1, rois2)
            rois1, rois2 = (torch.empty((sum(int(b.shape[0]) for b in bl), 5), dtype=torch.float32, device=feats[0].device)
                            for bl in lists)
        _C.require_gpu(rois1, rois2, *feats, op="pool_pair")
        n, c = feats[0].shape[:2]
        hw = [tuple(x.shape[2:]) for x in feats]
        code = _C.dtype_code(feats[0])
        p1, p2 = _params(cfg1, (n, c), hw, code, _C.NHWC), _params(cfg2, (n, c), hw, code, _C.NHWC)
        k1, k2 = int(rois1.shape[0]), int(rois2.shape[0])
        dev, dt = feats[0].device, feats[0].dtype
        out1 = torch.empty((k1, c) + tuple(cfg1[0]), dtype=dt, device=dev, memory_format=torch.channels_last)
        out2 = torch.empty((k2, c) + tuple(cfg2[0]), dtype=dt, device=dev, memory_format=torch.channels_last)
        L = _C.lib()
        with _C.on_device(dev):
            if lists is not None:
                n_img = len(lists[0])
                cnt = [(ctypes.c_int * n_img)(*[int(b.shape[0]) for b in bl]) for bl in lists]
                rc = L.d2amd_roi_pooler_forward_pair_box_lists(
                    ctypes.byref(p1), _ptr_array(feats), _ptr_array(lists[0]), cnt[0], _C.ptr(rois1), _C.ptr(out1),
                    ctypes.byref(p2), _ptr_array(lists[1]), cnt[1], _C.ptr(rois2), _C.ptr(out2), n_img, _C.stream())
            elif _FWD_RECORDS and _PAIR and plan is None and k1 > 0 and k2 > 0 and any(ctx.needs_input_grad[5:]):
                # (training: the forward's workgroups also write the backward's per-ROI records and reset its queues into
                # the backward's workspace, allocated here and kept until then: the backward starts with its tile lists)
                wsb = L.d2amd_roi_pooler_backward_pair_workspace_bytes(ctypes.byref(p1), k1, k2)
                ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
                wrote = ctypes.c_int(0)
                rc = L.d2amd_roi_pooler_forward_pair_records(ctypes.byref(p1), _ptr_array(feats), _C.ptr(rois1), _C.ptr(out1),
                                                             k1, ctypes.byref(p2), _C.ptr(rois2), _C.ptr(out2), k2, _C.ptr(ws),
                                                             wsb, ctypes.byref(wrote), _C.stream())
                if rc == 0 and wrote.value:
                    ctx.prep = (ws, wsb)
            else:
                rc = L.d2amd_roi_pooler_forward_pair(ctypes.byref(p1), _ptr_array(feats), _C.ptr(rois1), _C.ptr(out1), k1,
                                                     ctypes.byref(p2), _C.ptr(rois2), _C.ptr(out2), k2, _C.stream())
            if rc == _C.EUNSUPPORTED:  # (an empty list, different level rules ...): one launch each
                _C.check(L.d2amd_roi_pooler_forward(ctypes.byref(p1), _ptr_array(feats), _C.ptr(rois1), _C.ptr(out1), k1,
                                                    _C.stream()))
                _C.check(L.d2amd_roi_pooler_forward(ctypes.byref(p2), _ptr_array(feats), _C.ptr(rois2), _C.ptr(out2), k2,
                                                    _C.stream()))
            else:
                _C.check(rc)
        ctx.save_for_backward(rois1, rois2)
        ctx.cfgs, ctx.hw, ctx.nc, ctx.dtype = (cfg1, cfg2), hw, (n, c), dt
        ctx.needs = [f.requires_grad for f in feats]
        ctx.set_materialize_grads(False)
        return out1, out2

    @staticmethod
    @once_differentiable
    def backward(ctx, g1, g2):
        rois1, rois2 = ctx.saved_tensors
        (cfg1, cfg2), hw, (n, c) = ctx.cfgs, ctx.hw, ctx.nc
        if g1 is None and g2 is None:
            return (None,) * (5 + len(hw))
        dev = rois1.device
        works = [(_to_nhwc(g.detach()), r, cfg) for g, r, cfg in ((g1, rois1, cfg1), (g2, rois2, cfg2)) if g is not None]
        L = _C.lib()
        code = _C.dtype_code(works[0][0])
        # binned ahead of time (PairBackwardPlan.prepare, for exactly these ROI tensors and poolers): the gather alone
        ready = ctx.plan.ready if ctx.plan is not None else None
        if ctx.plan is not None:
            ctx.plan.ready = None
        if (ready is not None and len(works) == 2 and _PAIR and
                ctx.plan.matches(ready, rois1, rois2, cfg1, cfg2, (n, c), hw, ctx.dtype)):
            ev, ws, wsb, grads, _ = ready
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)
            ws.record_stream(cur)
            for g in grads:
                g.record_stream(cur)
            (ga, ra, ca), (gb, rb, cb) = works
            pa, pb = _params(ca, (n, c), hw, code, _C.NHWC), _params(cb, (n, c), hw, code, _C.NHWC)
            with _C.on_device(dev):
                _C.check(L.d2amd_roi_pooler_backward_pair_phase(ctypes.byref(pa), _C.ptr(ga), _C.ptr(ra), ra.shape[0],
                                                                ctypes.byref(pb), _C.ptr(gb), _C.ptr(rb), rb.shape[0],
                                                                _ptr_array(grads), _C.ptr(ws), wsb, 2, _C.stream()))
            return (None, None, None, None, None) + tuple(g if need else None for g, need in zip(grads, ctx.needs))
        grads = [torch.empty((n, c, h, w), dtype=ctx.dtype, device=dev, memory_format=torch.channels_last) for (h, w) in hw]
        with _C.on_device(dev):
            done = False
            prep, ctx.prep = ctx.prep, None
            if len(works) == 2 and _PAIR and prep is not None:  # records + queue reset came with the forward
                (ga, ra, ca), (gb, rb, cb) = works
                pa, pb = _params(ca, (n, c), hw, code, _C.NHWC), _params(cb, (n, c), hw, code, _C.NHWC)
                ws, wsb = prep
                ws.record_stream(torch.cuda.current_stream(dev))
                rc = L.d2amd_roi_pooler_backward_pair_phase(ctypes.byref(pa), _C.ptr(ga), _C.ptr(ra), ra.shape[0],
                                                            ctypes.byref(pb), _C.ptr(gb), _C.ptr(rb), rb.shape[0],
                                                            _ptr_array(grads), _C.ptr(ws), wsb, 5, _C.stream())
                done = rc == 0
                if rc not in (0, _C.EUNSUPPORTED):
                    _C.check(rc)
            if not done and len(works) == 2 and _PAIR:
                (ga, ra, ca), (gb, rb, cb) = works
                pa, pb = _params(ca, (n, c), hw, code, _C.NHWC), _params(cb, (n, c), hw, code, _C.NHWC)
                wsb = L.d2amd_roi_pooler_backward_pair_workspace_bytes(ctypes.byref(pa), ra.shape[0], rb.shape[0])
                ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
                rc = L.d2amd_roi_pooler_backward_pair(ctypes.byref(pa), _C.ptr(ga), _C.ptr(ra), ra.shape[0], ctypes.byref(pb),
                                                      _C.ptr(gb), _C.ptr(rb), rb.shape[0], _ptr_array(grads), _C.ptr(ws), wsb,
                                                      _C.stream())
                done = rc == 0
                if rc not in (0, _C.EUNSUPPORTED):
                    _C.check(rc)
            if not done:  # the first writes every tile, the second adds
                for j, (g, r, cfg) in enumerate(works):
                    p = _params(cfg, (n, c), hw, code, _C.NHWC)
                    k = r.shape[0]
                    wsb = L.d2amd_roi_pooler_backward_workspace_bytes(ctypes.byref(p), k)
                    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
                    if j == 0:
                        _C.check(L.d2amd_roi_pooler_backward(ctypes.byref(p), _C.ptr(g), _C.ptr(r), _ptr_array(grads), k,
                                                             _C.ptr(ws), wsb, _C.stream()))
                        continue
                    rc = L.d2amd_roi_pooler_backward_accumulate(ctypes.byref(p), _C.ptr(g), _C.ptr(r), _ptr_array(grads), k,
                                                                _C.ptr(ws), wsb, _C.stream())
                    if rc == _C.EUNSUPPORTED:
                        extra = [torch.empty_like(t) for t in grads]
                        _C.check(L.d2amd_roi_pooler_backward(ctypes.byref(p), _C.ptr(g), _C.ptr(r), _ptr_array(extra), k,
                                                             _C.ptr(ws), wsb, _C.stream()))
                        grads = [a + b for a, b in zip(grads, extra)]
                    else:
                        _C.check(rc)
        return (None, None, None, None, None) + tuple(g if need else None for g, need in zip(grads, ctx.needs))


def pool_pair(pooler1: "ROIPooler", pooler2: "ROIPooler", x: List[torch.Tensor], box_lists1, box_lists2):
    """(pooler1(x, box_lists1), pooler2(x, box_lists2)) for two multi-level ROIAlign poolers of the SAME feature maps --
    Mask R-CNN's box head (7x7, the sampled proposals) and mask head (14x14, the foreground ones), roi_heads.py:780-846 --
    as ONE launch per direction where the features are channels_last HIP tensors: the forward runs both poolers'
    workgroups in one grid (the mask head's fill the slots the box head's largest ROIs leave idle), the backward bins both
    ROI sets together and gathers every gradient tile once for both (d2amd_roi_pooler_backward_pair).  The forward values
    are bit for bit the separate calls'; the gradient rounds the fp32 sum of both poolers once.  Anything else (other
    layouts, rotated poolers, an empty list, different level rules) takes the two separate calls."""
    ok = (isinstance(x, list) and len(x) > 0 and pooler1._fusable(x) and pooler2._fusable(x)
          and _layout_of(x[0]) == _C.NHWC and len(box_lists1) == len(box_lists2) == x[0].shape[0] and len(box_lists1) > 0
          and len(x) == len(pooler1.level_poolers) == len(pooler2.level_poolers) and _PAIR)
    if not ok:
        return pooler1(x, box_lists1), pooler2(x, box_lists2)
    cfgs = [(tuple(p.output_size), tuple(p.scales), int(p.sampling_ratio), p.pooler_type == "ROIAlignV2", p.min_level,
             p.max_level, p.canonical_box_size, p.canonical_level) for p in (pooler1, pooler2)]
    # fast path: per-image fp32 HIP box tensors go to the C ABI as they are
    dev = x[0].device
    bts = [tuple(b.tensor if hasattr(b, "tensor") else b for b in bl) for bl in (box_lists1, box_lists2)]
    if len(bts[0]) <= 64 and all(t.dtype == torch.float32 and t.device == dev and t.dim() == 2 and t.shape[1] == 4
                                 and t.is_contiguous() and not t.requires_grad and t.data_ptr() % 16 == 0
                                 for bt in bts for t in bt):
        return _FusedROIPoolPair.apply(bts[0], bts[1], cfgs[0], cfgs[1], None, *x)
    rois = [convert_boxes_to_pooler_format(bl).detach().float().contiguous() for bl in (box_lists1, box_lists2)]
    return _FusedROIPoolPair.apply(rois[0], rois[1], cfgs[0], cfgs[1], None, *x)


def pool_pair_rois(pooler1: "ROIPooler", pooler2: "ROIPooler", x: List[torch.Tensor], rois1: torch.Tensor,
                   rois2: torch.Tensor, plan: "PairBackwardPlan" = None):
    """`pool_pair` for boxes that already are in pooler format ((M, 5) fp32 rows = image index, x1, y1, x2, y2: what
    `label_and_sample_proposals_fixed` writes as "rois" / "head_rois"); = (pooler1.pool_rois(x, rois1),
    pooler2.pool_rois(x, rois2)).  plan: a `PairBackwardPlan` whose `prepare(pooler1, pooler2, x, rois1, rois2)` the caller
    issues (before the backward, on any stream): the backward's binning then runs there."""
    for r in (rois1, rois2):
        assert r.dim() == 2 and r.shape[1] == 5 and r.dtype == torch.float32 and r.is_contiguous()
    ok = (isinstance(x, list) and len(x) > 0 and pooler1._fusable(x) and pooler2._fusable(x)
          and _layout_of(x[0]) == _C.NHWC and len(x) == len(pooler1.level_poolers) == len(pooler2.level_poolers) and _PAIR
          and rois1.shape[0] > 0 and rois2.shape[0] > 0)
    if not ok:
        return pooler1.pool_rois(x, rois1), pooler2.pool_rois(x, rois2)
    cfgs = [(tuple(p.output_size), tuple(p.scales), int(p.sampling_ratio), p.pooler_type == "ROIAlignV2", p.min_level,
             p.max_level, p.canonical_box_size, p.canonical_level) for p in (pooler1, pooler2)]
    return _FusedROIPoolPair.apply(rois1.detach(), rois2.detach(), cfgs[0], cfgs[1], plan, *x)


class ROIPooler(nn.Module):
    """Region of interest feature map pooler over one or more feature maps (poolers.py:114-263)."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224,
                 canonical_level=4):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        assert len(output_size) == 2
        assert isinstance(output_size[0], int) and isinstance(output_size[1], int)
        self.output_size = output_size
        self.pooler_type = pooler_type
        self.sampling_ratio = sampling_ratio
        self.scales = [float(s) for s in scales]
        if pooler_type in ("ROIAlign", "ROIAlignV2"):
            aligned = pooler_type == "ROIAlignV2"
            self.level_poolers = nn.ModuleList(
                ROIAlign(output_size, spatial_scale=s, sampling_ratio=sampling_ratio, aligned=aligned)
                for s in scales)
        elif pooler_type == "ROIAlignRotated":
            self.level_poolers = nn.ModuleList(
                ROIAlignRotated(output_size, spatial_scale=s, sampling_ratio=sampling_ratio) for s in scales)
        elif pooler_type == "ROIPool":
            raise ValueError("ROIPool (torchvision.ops.RoIPool) is outside the MI355X hot path; use ROIAlignV2")
        else:
            raise ValueError("Unknown pooler type: {}".format(pooler_type))
        # strides must be powers of two forming a pyramid (poolers.py:187-201)
        min_level = -(math.log2(scales[0]))
        max_level = -(math.log2(scales[-1]))
        assert math.isclose(min_level, int(min_level)) and math.isclose(max_level, int(max_level)), \
            "Featuremap stride is not power of 2!"
        self.min_level = int(min_level)
        self.max_level = int(max_level)
        assert len(scales) == self.max_level - self.min_level + 1, \
            "[ROIPooler] Sizes of input featuremaps do not form a pyramid!"
        assert 0 <= self.min_level and self.min_level <= self.max_level
        self.canonical_level = canonical_level
        assert canonical_box_size > 0
        self.canonical_box_size = canonical_box_size

    def _pool_fused(self, rois, cfg, x):
        """One fused launch; in training the call is chained to earlier / later poolers of the same features."""
        chain = torch.is_grad_enabled() and all(t.requires_grad for t in x)
        if not chain:
            return _FusedROIPool.apply(rois, cfg, None, True, None, *x)
        token = object()
        ins, upstream = _chained_inputs(x)
        if upstream is None and len(_DEFERRED) > 8:
            _DEFERRED.clear()  # work deferred to poolers whose backward never ran (pruned graphs): drop it
        _placeholder((1,), x[0].dtype, x[0].device)  # exists before any backward (and before a graph capture)
        res = _FusedROIPool.apply(rois, cfg, token, upstream is None, upstream, *ins)
        _remember_aliases(x, res[1:], token)
        return res[0]

    def _fusable(self, x):
        if self.pooler_type not in ("ROIAlign", "ROIAlignV2") or len(x) > 8:
            return False
        if max(self.output_size) > 32 or not all(t.is_cuda for t in x):
            return False
        lay, dt = _layout_of(x[0]), x[0].dtype
        return all(_layout_of(t) == lay and t.dtype == dt and t.shape[:2] == x[0].shape[:2] for t in x)

    def _fusable_rotated(self, x):
        if self.pooler_type != "ROIAlignRotated" or len(x) > 8 or _ROT_POOLER_LOOP:
            return False
        if not all(t.is_cuda and t.dim() == 4 for t in x) or x[0].dtype not in (torch.float32, torch.bfloat16, torch.float16):
            return False
        dt = x[0].dtype
        return all(_layout_of(t) == _C.NHWC and t.dtype == dt and t.shape[:2] == x[0].shape[:2] for t in x)

    def pool_rois(self, x: List[torch.Tensor], rois: torch.Tensor):
        """forward() for boxes that already are in pooler format: rois (M, 5) fp32 = (image index, x1, y1, x2, y2), what
        `convert_boxes_to_pooler_format` returns and `label_and_sample_proposals_fixed` writes ("rois" / "head_rois") --
        no conversion launch in front of the pooling kernel."""
        assert rois.dim() == 2 and rois.shape[1] == 5 and rois.dtype == torch.float32 and rois.is_contiguous()
        assert self._fusable(x), "pool_rois: multi-level ROIAlign on HIP tensors of one layout / dtype only"
        cfg = (tuple(self.output_size), tuple(self.scales), int(self.sampling_ratio),
               self.pooler_type == "ROIAlignV2", self.min_level, self.max_level, self.canonical_box_size,
               self.canonical_level)
        return self._pool_fused(rois.detach(), cfg, x)

    def forward(self, x: List[torch.Tensor], box_lists):
        """x: list of NCHW feature maps (scales as constructed); box_lists: N Boxes / RotatedBoxes (image
        coordinates).  Returns (M, C, output_size, output_size), M = total number of boxes."""
        num_level_assignments = len(self.level_poolers)
        assert isinstance(x, list) and isinstance(box_lists, list), "Arguments to pooler must be lists"
        assert len(x) == num_level_assignments, \
            "unequal value, num_level_assignments={}, but x is list of {} Tensors".format(num_level_assignments, len(x))
        assert len(box_lists) == x[0].size(0), \
            "unequal value, x[0] batch dim 0 is {}, but box_list has length {}".format(x[0].size(0), len(box_lists))
        if len(box_lists) == 0:
            return torch.zeros((0, x[0].shape[1]) + tuple(self.output_size), dtype=x[0].dtype, device=x[0].device)
        fusable = self._fusable(x)
        if fusable:
            cfg = (tuple(self.output_size), tuple(self.scales), int(self.sampling_ratio),
                   self.pooler_type == "ROIAlignV2", self.min_level, self.max_level, self.canonical_box_size,
                   self.canonical_level)
            # fast path: per-image fp32 HIP box tensors go to the C ABI as they are
            bt = tuple(b.tensor if hasattr(b, "tensor") else b for b in box_lists)
            dev = x[0].device
            if len(bt) <= 64 and all(t.dtype == torch.float32 and t.device == dev and t.dim() == 2 and t.shape[1] == 4
                                     and t.is_contiguous() and not t.requires_grad and t.data_ptr() % 16 == 0 for t in bt):
                return self._pool_fused(bt, cfg, x)
        pooler_fmt_boxes = convert_boxes_to_pooler_format(box_lists)
        if fusable:
            if pooler_fmt_boxes.dtype != torch.float32:
                pooler_fmt_boxes = pooler_fmt_boxes.float()
            return self._pool_fused(pooler_fmt_boxes.detach(), cfg, x)
        if self._fusable_rotated(x) and pooler_fmt_boxes.shape[1] == 6:
            # multi-level ROIAlignRotated in one launch per direction (channels_last features): no per-level nonzero /
            # index_put_ (D2AMD_ROT_POOLER_LOOP=1: the reference's per-level structure below)
            cfg = (tuple(self.output_size), tuple(self.scales), int(self.sampling_ratio), True, self.min_level,
                   self.max_level, self.canonical_box_size, self.canonical_level)
            return _FusedRotatedPool.apply(pooler_fmt_boxes.detach().float().contiguous(), cfg, *x)
        if num_level_assignments == 1:
            return self.level_poolers[0](x[0], pooler_fmt_boxes)
        # reference structure (poolers.py:247-263)
        level_assignments = assign_boxes_to_levels(box_lists, self.min_level, self.max_level,
                                                   self.canonical_box_size, self.canonical_level)
        output = torch.zeros((pooler_fmt_boxes.shape[0], x[0].shape[1], self.output_size[0], self.output_size[0]),
                             dtype=x[0].dtype, device=x[0].device)
        for level, pooler in enumerate(self.level_poolers):
            inds = torch.nonzero(level_assignments == level, as_tuple=True)[0]
            output.index_put_((inds,), pooler(x[level], pooler_fmt_boxes[inds]))
        return output
