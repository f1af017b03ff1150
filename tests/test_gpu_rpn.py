"""GPU parity of the fused RPN proposal selection (detectron2_amd/csrc/rpn.hip + modeling/proposal_utils.py,
SURVEY 8(f) row 2) against the reference's functions (tests/golden/rpn_proposals.npz) and the oracle:
selection indices and validity exact (distinct logits), decoded boxes to the rounding of exp()
(rtol 2e-6 / atol 2e-4 px), the NMS + top-k part exact on the device's own decoded boxes."""
import os

import numpy as np
import pytest
import torch

from oracle import rpn as orpn
from detectron2_amd.modeling import find_top_rpn_proposals_fused, rpn_select_proposals

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "rpn_proposals.npz"))
    anchors = [g[f"anchors{l}"] for l in range(3)]
    logits = [g[f"logits{l}"] for l in range(3)]
    deltas = [g[f"deltas{l}"] for l in range(3)]
    n = logits[0].shape[0]
    hw = [tuple(int(v) for v in g["image_hw"])] * n
    return g, anchors, logits, deltas, hw


def _dev(xs):
    return [torch.from_numpy(x).to(DEV) for x in xs]


def test_rpn_select_matches_oracle_and_reference_decode(golden_dir):
    g, anchors, logits, deltas, hw = _golden(golden_dir)
    topk, minsz = int(g["pre_nms_topk"]), float(g["min_box_size"])
    boxes, scores, valid, lv, flags = rpn_select_proposals(_dev(anchors), _dev(logits), _dev(deltas), hw, topk, minsz)
    exp = orpn.select(anchors, logits, deltas, hw, topk, minsz)
    assert int(flags.item()) == 0
    for i, (eb, es, ev, el, eidx) in enumerate(exp):
        assert np.array_equal(lv.cpu().numpy(), el)
        assert np.array_equal(valid[i].cpu().numpy(), ev)
        gs = scores[i].cpu().numpy()
        assert np.array_equal(gs[ev], es[ev]) and np.all(np.isneginf(gs[~ev]))
        gb = boxes[i].cpu().numpy()
        assert np.allclose(gb[ev], eb[ev], rtol=2e-6, atol=2e-4)
        assert np.all(gb[~ev] == 0)
        # the reference's own decode of the selected anchors, clipped
        off = 0
        for l in range(3):
            k = min(anchors[l].shape[0], topk)
            ref = g[f"decoded{l}"][i][eidx[off:off + k]].copy()
            ref[:, 0::2] = np.clip(ref[:, 0::2], 0, hw[i][1]); ref[:, 1::2] = np.clip(ref[:, 1::2], 0, hw[i][0])
            m = ev[off:off + k]
            assert np.allclose(gb[off:off + k][m], ref[m], rtol=2e-6, atol=2e-4)
            off += k


def test_find_top_rpn_proposals_fused_matches_reference(golden_dir):
    g, anchors, logits, deltas, hw = _golden(golden_dir)
    res = find_top_rpn_proposals_fused(_dev(anchors), _dev(logits), _dev(deltas), hw, float(g["nms_thresh"]),
                                       int(g["pre_nms_topk"]), int(g["post_nms_topk"]), float(g["min_box_size"]), False)
    for i, r in enumerate(res):
        assert np.array_equal(r.objectness_logits.cpu().numpy(), g[f"scores_img{i}"])   # same proposals, same order
        assert np.allclose(r.proposal_boxes.tensor.cpu().numpy(), g[f"boxes_img{i}"], rtol=2e-6, atol=2e-4)


def test_rpn_full_size_pipeline_and_nonfinite():
    """BASELINE configs[1] shapes: 5 FPN levels, 268,569 anchors x 2 images, pre/post top-k 2000/1000.  The NMS
    part is exact given the device's decoded boxes; NaN deltas raise in training and are dropped otherwise."""
    torch.manual_seed(3)
    sizes = [201600, 50400, 12600, 3150, 819]
    H, W = 800, 1344
    anchors, logits, deltas = [], [], []
    for l, a in enumerate(sizes):
        s = 32.0 * 2 ** l
        c = torch.rand(a, 2) * torch.tensor([W, H])
        wh = s * torch.exp(torch.rand(a, 2) - 0.5)
        anchors.append(torch.cat([c - wh / 2, c + wh / 2], 1))
        logits.append(torch.randn(2, a) + torch.arange(a) * 1e-7)
        deltas.append(torch.randn(2, a, 4) * torch.tensor([0.2, 0.2, 0.3, 0.3]))
    hw = [(H, W)] * 2
    A, Lg, D = [t.to(DEV) for t in anchors], [t.to(DEV) for t in logits], [t.to(DEV) for t in deltas]
    boxes, scores, valid, lv, flags = rpn_select_proposals(A, Lg, D, hw, 2000, 0.0)
    assert boxes.shape == (2, 8819, 4) and int(flags.item()) == 0
    exp = orpn.select([a.numpy() for a in anchors], [t.numpy() for t in logits], [t.numpy() for t in deltas], hw, 2000, 0.0)
    for i in range(2):
        assert np.array_equal(valid[i].cpu().numpy(), exp[i][2])
        assert np.array_equal(scores[i].cpu().numpy()[exp[i][2]], exp[i][1][exp[i][2]])
        assert np.allclose(boxes[i].cpu().numpy()[exp[i][2]], exp[i][0][exp[i][2]], rtol=2e-6, atol=5e-4)
    res = find_top_rpn_proposals_fused(A, Lg, D, hw, 0.7, 2000, 1000, 0.0, True)
    sel = [(boxes[i].cpu().numpy(), scores[i].cpu().numpy(), valid[i].cpu().numpy(), lv.cpu().numpy(), None)
           for i in range(2)]
    ref = orpn.find_top_rpn_proposals(None, None, None, hw, 0.7, 2000, 1000, 0.0, selected=sel)
    for r, (eb, es) in zip(res, ref):
        assert len(r) == len(es) <= 1000
        assert np.array_equal(r.objectness_logits.cpu().numpy(), es)
        assert np.array_equal(r.proposal_boxes.tensor.cpu().numpy(), eb)
    # small post_nms_topk values (every level keeps more than that): still the oracle's proposals, in the oracle's order
    for post in (1, 37, 64, 300):
        res = find_top_rpn_proposals_fused(A, Lg, D, hw, 0.7, 2000, post, 0.0, True)
        ref = orpn.find_top_rpn_proposals(None, None, None, hw, 0.7, 2000, post, 0.0, selected=sel)
        for r, (eb, es) in zip(res, ref):
            assert len(r) == len(es) == post
            assert np.array_equal(r.objectness_logits.cpu().numpy(), es)
            assert np.array_equal(r.proposal_boxes.tensor.cpu().numpy(), eb)
    # non-finite predictions
    D[0][1, :4000, 2] = float("nan")
    with pytest.raises(FloatingPointError):
        find_top_rpn_proposals_fused(A, Lg, D, hw, 0.7, 2000, 1000, 0.0, True)
    res = find_top_rpn_proposals_fused(A, Lg, D, hw, 0.7, 2000, 1000, 0.0, False)
    assert torch.isfinite(res[1].proposal_boxes.tensor).all() and len(res[1]) > 0


def test_concatenated_entry_equals_the_per_level_entry(golden_dir):
    """d2amd_rpn_select_proposals (levels concatenated, [N, Atot]) and d2amd_rpn_select_proposals_levels (the head's
    per-level tensors, what the Python mirror calls) are one implementation behind two layouts: identical outputs."""
    import ctypes

    from detectron2_amd import _C

    g, anchors, logits, deltas, hw = _golden(golden_dir)
    topk, minsz = int(g["pre_nms_topk"]), float(g["min_box_size"])
    A, Lg, D = _dev(anchors), _dev(logits), _dev(deltas)
    exp = rpn_select_proposals(A, Lg, D, hw, topk, minsz)
    n, sizes = Lg[0].shape[0], [a.shape[0] for a in A]
    atot, k = sum(sizes), sum(min(s, topk) for s in sizes)
    cl, cd, ca = torch.cat(Lg, 1).contiguous(), torch.cat(D, 1).contiguous(), torch.cat(A, 0).contiguous()
    boxes = torch.empty((n, k, 4), device=DEV)
    scores = torch.empty((n, k), device=DEV)
    valid = torch.empty((n, k), dtype=torch.bool, device=DEV)
    lv = torch.empty((k,), dtype=torch.int64, device=DEV)
    flags = torch.empty((1,), dtype=torch.int32, device=DEV)
    L = _C.lib()
    ws_bytes = L.d2amd_rpn_select_workspace_bytes(n, atot)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    _C.check(L.d2amd_rpn_select_proposals(
        _C.ptr(cl), _C.ptr(cd), _C.ptr(ca), n, atot, (ctypes.c_int * len(sizes))(*sizes), len(sizes),
        (ctypes.c_int * (2 * n))(*[int(v) for s_ in hw for v in s_]), topk, minsz, (ctypes.c_float * 4)(1, 1, 1, 1),
        float(np.log(1000.0 / 16)), _C.ptr(boxes), _C.ptr(scores), _C.ptr(valid), _C.ptr(lv), _C.ptr(flags),
        _C.ptr(ws), ws_bytes, _C.stream()))
    for got, want in zip((boxes, scores, valid, lv, flags), exp):
        assert torch.equal(got, want)


def test_multi_launch_selection_path_still_agrees():
    """The RPN's selection runs as one fused launch when its workgroups are co-resident (csrc/topk.hip:
    tk_fused_kernel); D2AMD_TOPK_MULTI=1 forces the launch-per-pass path that larger inputs take.  Both must pass the
    oracle / reference comparisons of this file and the dense detector's tie tests."""
    import subprocess
    import sys

    if os.environ.get("D2AMD_TOPK_MULTI"):
        pytest.skip("already the child")
    env = dict(os.environ, D2AMD_TOPK_MULTI="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_rpn.py",
                        "tests/test_gpu_dense.py", "-k", "oracle or reference or ties or tie"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_device_proposals_exist_only_when_the_batched_nms_filled_the_result_buffer():
    """`find_top_rpn_proposals_fused(..., defer=True).device` hands the proposal lists + their device-side counts to a
    sync-free consumer (bench.py's connected step).  The counts live in the batched NMS pipeline's result buffer: with
    more than 12,288 boxes per image (here 5 x 3,000) the NMS runs per image into its own buffers and `.device` must be
    None instead of pointing at memory nobody wrote; with the RPN's 8,819 it must agree with the synchronous result."""
    torch.manual_seed(4)
    sizes = [20000, 10000, 6000, 4000, 3000]
    H, W = 800, 1344
    A, Lg, D = [], [], []
    for l, a in enumerate(sizes):
        s = 32.0 * 2 ** l
        c = torch.rand(a, 2) * torch.tensor([W, H])
        wh = s * torch.exp(torch.rand(a, 2) - 0.5)
        A.append(torch.cat([c - wh / 2, c + wh / 2], 1).to(DEV))
        Lg.append((torch.randn(2, a) + torch.arange(a) * 1e-7).to(DEV))
        D.append((torch.randn(2, a, 4) * 0.2).to(DEV))
    hw = [(H, W)] * 2
    big = find_top_rpn_proposals_fused(A, Lg, D, hw, 0.7, 3000, 1000, 0.0, True, defer=True, host_result=False)
    assert big.device is None
    assert len(big()) == 2 and all(0 < len(p) <= 1000 for p in big())
    small = find_top_rpn_proposals_fused(A, Lg, D, hw, 0.7, 2000, 1000, 0.0, True, defer=True)
    dp = small.device
    assert dp is not None and int(dp.nonfinite_flag.item()) == 0
    counts = dp.counts().tolist()
    props = small()
    assert counts == [len(p) for p in props]
    for i, p in enumerate(props):
        assert torch.equal(dp.boxes[i][:counts[i]], p.proposal_boxes.tensor)
        assert int(dp.limits[i][1]) == 0  # no NMS flag
