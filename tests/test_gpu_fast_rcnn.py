"""GPU parity of the fused box-head inference (csrc/box_head.hip + d2amd_nms_batched behind
detectron2_amd.modeling.fast_rcnn_inference_fused) against oracle/fast_rcnn.py (numpy restatement of
roi_heads/fast_rcnn.py:118-170, pinned to the reference's own source in tests/test_oracle_golden.py) and against the
reference's OWN function run unmodified on this package's batched_nms on the GPU.  Bar: bit-exact -- the kept boxes,
scores, classes, row indices and their ORDER."""
import numpy as np
import pytest
import torch

from _fast_rcnn_cases import CASES, make
from conftest import need_reference
from detectron2_amd import layers
from detectron2_amd.modeling import fast_rcnn_inference_device, fast_rcnn_inference_fused
from oracle import fast_rcnn as ofr
from oracle import ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def run_fused(case):
    boxes, scores, shapes, thr, nms, topk = make(case)
    res, rows = fast_rcnn_inference_fused([torch.from_numpy(b).to(DEV) for b in boxes],
                                          [torch.from_numpy(s).to(DEV) for s in scores], shapes, thr, nms, topk)
    return (boxes, scores, shapes, thr, nms, topk), res, rows


@pytest.mark.parametrize("case", CASES)
def test_fused_vs_oracle(case):
    (boxes, scores, shapes, thr, nms, topk), res, rows = run_fused(case)
    assert len(res) == len(boxes)
    total = 0
    for i in range(len(boxes)):
        wb, ws, wc, wr = ofr.fast_rcnn_inference_single_image(boxes[i], scores[i], shapes[i], thr, nms, topk)
        assert res[i].image_size == tuple(shapes[i])
        assert np.array_equal(res[i].pred_classes.cpu().numpy(), wc), (case, i)
        assert np.array_equal(rows[i].cpu().numpy(), wr), (case, i)
        assert np.array_equal(res[i].scores.cpu().numpy(), ws), (case, i)
        assert np.array_equal(res[i].pred_boxes.tensor.cpu().numpy(), wb), (case, i)
        total += len(ws)
    if case == "none_pass":
        assert total == 0
    elif case != "ragged":
        assert total > 0


@pytest.mark.parametrize("case", ["maskrcnn", "agnostic", "nonfinite", "ties"])
def test_fused_vs_the_references_own_function_on_the_gpu(case):
    """fast_rcnn.py's fast_rcnn_inference, loaded unmodified (bytecode staged by oracle/build_ref.py), running on HIP
    tensors with `detectron2.layers.batched_nms` = this package's: the per-image loop with its two host syncs per image
    gives the same detections as the fused batch call."""
    need_reference(ref.have_py(), "oracle/_ref/py (the reference's fast_rcnn.py)")
    m = ref.py_fast_rcnn(layers.batched_nms)
    (boxes, scores, shapes, thr, nms, topk), res, rows = run_fused(case)
    inst, kept = m.fast_rcnn_inference([torch.from_numpy(b).to(DEV) for b in boxes],
                                       [torch.from_numpy(s).to(DEV) for s in scores], shapes, thr, nms, topk)
    for i in range(len(boxes)):
        assert torch.equal(inst[i].pred_boxes.tensor, res[i].pred_boxes.tensor)
        assert torch.equal(inst[i].scores, res[i].scores)
        assert torch.equal(inst[i].pred_classes, res[i].pred_classes)
        assert torch.equal(kept[i], rows[i])


def test_filter_counts_and_order_without_the_nms():
    """The filter alone (the C entry): counts and the row-major candidate order of torch.nonzero, for the worst-case
    slices layout."""
    import ctypes

    from detectron2_amd import _C

    boxes, scores, shapes, thr, _, _ = make("nonfinite", seed=3)
    n = len(boxes)
    K = scores[0].shape[1] - 1
    rows = [b.shape[0] for b in boxes]
    base = np.concatenate([[0], np.cumsum([r * K for r in rows])]).astype(np.int64)
    bx = [torch.from_numpy(b).to(DEV) for b in boxes]
    sc = [torch.from_numpy(s).to(DEV) for s in scores]
    ob = torch.empty((int(base[-1]), 4), device=DEV)
    os_ = torch.empty(int(base[-1]), device=DEV)
    oc = torch.empty(int(base[-1]), dtype=torch.int64, device=DEV)
    orow = torch.empty(int(base[-1]), dtype=torch.int64, device=DEV)
    cnt = torch.zeros(n, dtype=torch.int64, device=DEV)
    L = _C.lib()
    rows_c = (ctypes.c_int * n)(*rows)
    ws_bytes = L.d2amd_fast_rcnn_filter_workspace_bytes(rows_c, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    hw = (ctypes.c_int * (2 * n))(*[int(v) for s in shapes for v in s])
    ptrs = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    _C.check(L.d2amd_fast_rcnn_filter(ptrs(bx), ptrs(sc), rows_c, n, K, K, hw, float(thr), _C.ptr(ob), _C.ptr(os_), _C.ptr(oc),
                                      _C.ptr(orow), _C.ptr(cnt), _C.ptr(ws), ws_bytes, _C.stream()))
    for i in range(n):
        valid = np.isfinite(boxes[i]).all(1) & np.isfinite(scores[i]).all(1)
        assert 0 < (~valid).sum() < rows[i]
        m = (scores[i][:, :-1] > np.float32(thr)) & valid[:, None]
        inds = np.argwhere(m)
        c = int(cnt[i])
        assert c == len(inds)
        sl = slice(int(base[i]), int(base[i]) + c)
        assert np.array_equal(orow[sl].cpu().numpy(), (np.cumsum(valid) - 1)[inds[:, 0]])  # index among the kept rows
        assert np.array_equal(oc[sl].cpu().numpy(), inds[:, 1])
        assert np.array_equal(os_[sl].cpu().numpy(), scores[i][:, :-1][m])
        h, w = shapes[i]
        want = boxes[i].reshape(rows[i], K, 4)[m].copy()
        want[:, 0::2] = np.clip(want[:, 0::2], 0, w); want[:, 1::2] = np.clip(want[:, 1::2], 0, h)
        assert np.array_equal(ob[sl].cpu().numpy(), want)


@pytest.mark.parametrize("case", ["maskrcnn", "agnostic", "ragged", "nonfinite", "none_pass", "ties", "many"])
@pytest.mark.parametrize("capacity", [None, 3000])
def test_device_path_equals_the_synchronous_one(case, capacity):
    """fast_rcnn_inference_device: no host sync until finish() (fixed-shape candidate windows, slots past the count
    parked; counts read by the NMS on the device) -- the detections, their order and the kept rows are those of
    fast_rcnn_inference_fused bit for bit; an image with more candidates than slots ("many": ~25,000 > 12,288; capacity
    3,000 for "maskrcnn") is recomputed by finish() through the synchronous path.  The fixed-shape tensors a captured
    chain consumes agree with the lists up to the count and hold 1 x 1 boxes / zeros behind it."""
    boxes, scores, shapes, thr, nms, topk = make(case)
    tb = [torch.from_numpy(b).to(DEV) for b in boxes]
    ts = [torch.from_numpy(s).to(DEV) for s in scores]
    want, want_rows = fast_rcnn_inference_fused(tb, ts, shapes, thr, nms, topk)
    dd = fast_rcnn_inference_device(tb, ts, shapes, thr, nms, topk, capacity=capacity)
    got, got_rows = dd.finish()
    counts = dd.counts.tolist()
    for i in range(len(boxes)):
        assert torch.equal(got[i].pred_boxes.tensor, want[i].pred_boxes.tensor), (case, i)
        assert torch.equal(got[i].scores, want[i].scores) and torch.equal(got[i].pred_classes, want[i].pred_classes)
        assert torch.equal(got_rows[i], want_rows[i])
        assert got[i].num_candidates == want[i].num_candidates
        m = len(want[i].scores)
        if got[i].num_candidates <= (capacity or 12288):  # served at fixed shape
            assert counts[i] == m and dd.boxes[i].shape == (topk, 4)
            assert torch.equal(dd.boxes[i][:m], want[i].pred_boxes.tensor) and torch.equal(dd.classes[i][:m], want[i].pred_classes)
            assert torch.equal(dd.boxes[i][m:], torch.tensor([0.0, 0.0, 1.0, 1.0], device=DEV).expand(topk - m, 4))
            assert not dd.scores[i][m:].any() and not dd.classes[i][m:].any()


def test_device_path_replays_in_a_hip_graph():
    """The whole call is capturable: replayed on new inputs (copied into the captured buffers) it gives what the eager
    synchronous path gives for them."""
    boxes, scores, shapes, thr, nms, topk = make("maskrcnn")
    tb = [torch.from_numpy(b).to(DEV) for b in boxes]
    ts = [torch.from_numpy(s).to(DEV) for s in scores]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fast_rcnn_inference_device(tb, ts, shapes, thr, nms, topk)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        dd = fast_rcnn_inference_device(tb, ts, shapes, thr, nms, topk)
    for seed in (1, 2):
        b2, s2, *_ = make("maskrcnn", seed=seed)
        for t, src in zip(tb + ts, b2 + s2):
            t.copy_(torch.from_numpy(src))
        g.replay()
        got, _rows = dd.finish()
        want, _ = fast_rcnn_inference_fused(tb, ts, shapes, thr, nms, topk)
        for a, b in zip(got, want):
            assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)


# ---- fast_rcnn_predict: FastRCNNOutputLayers.predict_boxes + predict_probs in one launch --------------------------------
def _torch_apply_deltas(deltas, boxes, weights, clamp):
    """box_regression.py:88-116 op for op in torch (the elementwise chain the fused launch replaces)."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    widths, heights = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    ctr_x, ctr_y = boxes[:, 0] + 0.5 * widths, boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx, dy, dw, dh = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy, deltas[:, 2::4] / ww, deltas[:, 3::4] / wh
    dw, dh = torch.clamp(dw, max=clamp), torch.clamp(dh, max=clamp)
    pcx, pcy = dx * widths[:, None] + ctr_x[:, None], dy * heights[:, None] + ctr_y[:, None]
    pw, ph = torch.exp(dw) * widths[:, None], torch.exp(dh) * heights[:, None]
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=-1).reshape(deltas.shape)


def _predict_inputs(rows, k_cls, kb, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    r = sum(rows)
    scores = (torch.randn(r, k_cls + 1, generator=g) * 3.0).to(dtype).to(DEV)
    deltas = (torch.randn(r, kb * 4, generator=g) * 0.5).to(dtype)
    deltas[::7, 2::4] = 30.0   # past scale_clamp
    deltas = deltas.to(DEV)
    xy = torch.rand(r, 2, generator=g) * 700
    wh = torch.rand(r, 2, generator=g) * 300 + 0.5
    props = torch.cat([xy, xy + wh], 1).to(DEV)
    return scores, deltas, list(props.split(rows))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,k_cls,agnostic", [([1000, 1000], 80, False), ([37, 0, 5], 80, True), ([3], 1, False),
                                                 ([129, 64], 200, False)])
def test_predict_boxes_and_probs_vs_the_elementwise_chain(dtype, rows, k_cls, agnostic):
    from detectron2_amd.modeling import fast_rcnn_predict

    kb = 1 if agnostic else k_cls
    scores, deltas, props = _predict_inputs(rows, k_cls, kb, dtype, 5)
    weights, clamp = (10.0, 10.0, 5.0, 5.0), float(np.log(1000.0 / 16))
    boxes, probs = fast_rcnn_predict(scores, deltas, props, weights, clamp)
    want_b = _torch_apply_deltas(deltas, torch.cat(props), weights, clamp)
    want_p = torch.softmax(scores, dim=-1)
    got_b, got_p = torch.cat(boxes), torch.cat(probs)
    assert [tuple(b.shape) for b in boxes] == [(r, kb * 4) for r in rows] and got_b.dtype == torch.float32
    assert [tuple(p.shape) for p in probs] == [(r, k_cls + 1) for r in rows] and got_p.dtype == dtype
    assert torch.equal(got_b, want_b), "decoded boxes differ from box_regression.py's expression order"
    # softmax: the sum's reduction order differs from ATen's -> a few fp32 ulps before the output rounding
    tol = {torch.float32: 1e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    assert torch.allclose(got_p.float(), want_p.float(), rtol=tol, atol=1e-9 if dtype == torch.float32 else tol * 1e-2)
    assert torch.allclose(got_p.float().sum(1), torch.ones(sum(rows), device=DEV), atol={torch.float32: 1e-5}.get(dtype, 2e-2))
    # sigmoid head (use_sigmoid_ce)
    _b, sig = fast_rcnn_predict(scores, deltas, props, weights, clamp, use_sigmoid_ce=True)
    assert torch.allclose(torch.cat(sig).float(), torch.sigmoid(scores.float()).to(dtype).float(), rtol=tol, atol=tol * 1e-2)


def test_predict_rows_behind_a_device_side_count():
    """limits = the NMS result rows of a device-side proposal list: rows at / behind min(kept, finite) predict background
    with probability 1 and zero boxes; the other rows are untouched by the argument."""
    from detectron2_amd.modeling import fast_rcnn_predict

    rows = [50, 40, 30]
    scores, deltas, props = _predict_inputs(rows, 80, 80, torch.float32, 9)
    lim = [torch.tensor(v, dtype=torch.int64, device=DEV) for v in ([20, 0, 50, 0], [99, 0, 7, 0], [0, 0, 0, 0])]
    live = [20, 7, 0]
    b0, p0 = fast_rcnn_predict(scores, deltas, props)
    b1, p1 = fast_rcnn_predict(scores, deltas, props, limits=lim)
    for i in range(3):
        assert torch.equal(b1[i][:live[i]], b0[i][:live[i]]) and torch.equal(p1[i][:live[i]], p0[i][:live[i]])
        assert (b1[i][live[i]:] == 0).all()
        assert (p1[i][live[i]:, :-1] == 0).all() and (p1[i][live[i]:, -1] == 1).all()


def test_proposals_pad():
    from detectron2_amd.modeling.proposal_utils import DeviceProposals

    boxes = [torch.rand(10, 4, device=DEV) + 5, torch.rand(6, 4, device=DEV) + 5]
    keep = [b.clone() for b in boxes]
    lim = [torch.tensor(v, dtype=torch.int64, device=DEV) for v in ([4, 0, 9, 0], [6, 0, 6, 0])]
    DeviceProposals(boxes, [None, None], lim, None, [(1, 1)] * 2).pad_()
    unit = torch.tensor([0.0, 0.0, 1.0, 1.0], device=DEV)
    assert torch.equal(boxes[0][:4], keep[0][:4]) and (boxes[0][4:] == unit).all()
    assert torch.equal(boxes[1], keep[1])
