"""GPU tests of the PAIRED pooler backward (d2amd_roi_pooler_backward_pair, csrc/roi_pool.hip: pool_bwd_mfma_kernel<T, 8,
true, 16>): the box head's 7x7 pooler and the mask head's 14x14 pooler of the same FPN features (roi_heads.py:780-846,
modeling/poolers.py:206-263 twice) binned together and gathered in ONE pass over the gradient's tiles.

Checked against (a) the oracle's two ROIAlign backwards summed (what autograd accumulates in the reference) and (b) the
library's own two-call sequence d2amd_roi_pooler_backward + d2amd_roi_pooler_backward_accumulate: equal bit for bit on
every tile only one pooler touches, within the roundings the two-call sequence adds where both do (it rounds each
pooler's sum to the I/O dtype, then their sum; the paired launch rounds the fp32 sum of both once)."""
import ctypes

import numpy as np
import pytest
import torch

from detectron2_amd import _C
from detectron2_amd.modeling import ROIPooler
from detectron2_amd.modeling import poolers as P
from detectron2_amd.structures import Boxes

from test_gpu_pooler import SCALES, make_inputs, oracle_pooler, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rois(boxes):
    allb = np.concatenate(boxes)
    bidx = np.concatenate([np.full(len(b), i, np.float32) for i, b in enumerate(boxes)])
    return torch.from_numpy(np.concatenate([bidx[:, None], allb], 1).astype(np.float32)).to(DEV)


def _cfg(out):
    return ((out, out), tuple(SCALES), 0, True, 2, 5, 224, 4)


def _nhwc(a, dtype):
    return torch.from_numpy(a).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)


def _run(kind, feats, boxes1, g1, boxes2, g2, dtype, out1=7, out2=14):
    """-> (rc, per-level gradients) of the paired call / of the two-call sequence"""
    L = _C.lib()
    n, c = feats[0].shape[:2]
    hw = [tuple(f.shape[2:]) for f in feats]
    code = _C.dtype_code(torch.empty(0, dtype=dtype))
    p1, p2 = P._params(_cfg(out1), (n, c), hw, code, _C.NHWC), P._params(_cfg(out2), (n, c), hw, code, _C.NHWC)
    r1, r2 = _rois(boxes1), _rois(boxes2)
    k1, k2 = int(r1.shape[0]), int(r2.shape[0])
    grads = [torch.full((n, c) + s, 7.0, dtype=dtype, device=DEV).contiguous(memory_format=torch.channels_last) for s in hw]
    with _C.on_device(grads[0].device):
        if kind == "pair":
            wsb = L.d2amd_roi_pooler_backward_pair_workspace_bytes(ctypes.byref(p1), k1, k2)
            ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
            rc = L.d2amd_roi_pooler_backward_pair(ctypes.byref(p1), _C.ptr(g1), _C.ptr(r1), k1, ctypes.byref(p2), _C.ptr(g2),
                                                  _C.ptr(r2), k2, P._ptr_array(grads), _C.ptr(ws), wsb, _C.stream())
        else:
            b1 = L.d2amd_roi_pooler_backward_workspace_bytes(ctypes.byref(p1), k1)
            b2 = L.d2amd_roi_pooler_backward_workspace_bytes(ctypes.byref(p2), k2)
            ws1, ws2 = torch.empty(b1, dtype=torch.uint8, device=DEV), torch.empty(b2, dtype=torch.uint8, device=DEV)
            rc = L.d2amd_roi_pooler_backward(ctypes.byref(p1), _C.ptr(g1), _C.ptr(r1), P._ptr_array(grads), k1, _C.ptr(ws1),
                                             b1, _C.stream())
            if rc == 0:
                rc = L.d2amd_roi_pooler_backward_accumulate(ctypes.byref(p2), _C.ptr(g2), _C.ptr(r2), P._ptr_array(grads),
                                                            k2, _C.ptr(ws2), b2, _C.stream())
    torch.cuda.synchronize()
    return rc, grads


def _ulps_apart(a, b):
    """distance in representable values of the 16-bit dtype (sign-magnitude bit patterns mapped to a line)"""
    def line(t):
        v = t.view(torch.int16).to(torch.int32)
        return torch.where(v < 0, -(v & 0x7FFF), v)
    return (line(a) - line(b)).abs()


def _clustered(rng, boxes, per_tile, img_h, img_w, size=(150, 210)):
    out = []
    for b0 in boxes:
        c = rng.uniform([150, 120], [300, 200])
        ctr = c + rng.uniform(-6, 6, (per_tile, 2))
        wh = rng.uniform(size[0], size[1], (per_tile, 2))
        b = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1)
        b[:, 0::2] = b[:, 0::2].clip(0, img_w)
        b[:, 1::2] = b[:, 1::2].clip(0, img_h)
        out.append(np.concatenate([b0, b.astype(np.float32)]))
    return out


CASES = {
    # the second pooler's boxes are a SUBSET of the first one's (Mask R-CNN: the foreground proposals)
    "subset": dict(per1=48, per2=12, subset=True),
    # independent boxes: tiles of all three kinds (first only / second only / both)
    "independent": dict(per1=40, per2=24, subset=False),
    # lists beyond SPLIT_MIN on a few tiles: cut into parts, some of which hold entries of both poolers
    "first_split": dict(per1=20, per2=16, subset=False, cluster1=64),
    # lists beyond the per-tile list capacity: in-kernel scan of the first, then of the second pooler's records
    "second_scans": dict(per1=24, per2=8, subset=False, cluster2=90),
    # both long
    "both_long": dict(per1=16, per2=8, subset=False, cluster1=50, cluster2=50),
    # few, small boxes: most tiles empty, zero-filled by the first pooler's binning
    "sparse": dict(per1=3, per2=2, subset=False),
}


def _case(name, dtype, C=64, img_h=320, img_w=448):
    cs = CASES[name]
    rng = np.random.default_rng(sum(map(ord, name)))
    feats, boxes1 = make_inputs(rng, 2, C, img_h, img_w, cs["per1"])
    if cs["subset"]:
        boxes2 = [b[:cs["per2"]].copy() for b in boxes1]
    else:
        _, boxes2 = make_inputs(rng, 2, C, img_h, img_w, cs["per2"])
    if "cluster1" in cs:
        boxes1 = _clustered(rng, boxes1, cs["cluster1"], img_h, img_w)
    if "cluster2" in cs:
        boxes2 = _clustered(rng, boxes2, cs["cluster2"], img_h, img_w, size=(160, 200))
    k1, k2 = sum(len(b) for b in boxes1), sum(len(b) for b in boxes2)
    g1 = _nhwc(rng.standard_normal((k1, C, 7, 7)).astype(np.float32), dtype)
    g2 = _nhwc(rng.standard_normal((k2, C, 14, 14)).astype(np.float32), dtype)
    return feats, boxes1, g1, boxes2, g2


def _tile_any(m):
    """[N, C, H, W] bool -> the same shape: True on every element of an 8 x 8-pixel tile (all channels) that holds a True."""
    n, c, h, w = m.shape
    hp, wp = -(-h // 8) * 8, -(-w // 8) * 8
    p = np.zeros((n, hp, wp), bool)
    p[:, :h, :w] = m.any(1)
    t = p.reshape(n, hp // 8, 8, wp // 8, 8).any((2, 4))
    e = np.repeat(np.repeat(t, 8, 1), 8, 2)[:, :h, :w]
    return np.broadcast_to(e[:, None], m.shape).copy()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name", list(CASES))
def test_pair_vs_oracle_and_vs_the_two_call_sequence(name, dtype):
    feats, boxes1, g1, boxes2, g2 = _case(name, dtype)
    rc, pair = _run("pair", feats, boxes1, g1, boxes2, g2, dtype)
    assert rc == 0, _C.lib().d2amd_last_error().decode()
    rc, two = _run("two", feats, boxes1, g1, boxes2, g2, dtype)
    assert rc == 0
    _, gin1, _ = oracle_pooler(feats, boxes1, 7, 0, True, grad=g1.float().cpu().numpy())
    _, gin2, _ = oracle_pooler(feats, boxes2, 14, 0, True, grad=g2.float().cpu().numpy())
    # the same scatter of |dY| (the weights are >= 0: A = sum |w dY| exactly): what an fp32 summation order is worth
    _, abs1, _ = oracle_pooler(feats, boxes1, 7, 0, True, grad=np.abs(g1.float().cpu().numpy()))
    _, abs2, _ = oracle_pooler(feats, boxes2, 14, 0, True, grad=np.abs(g2.float().cpu().numpy()))
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    differ = 0
    for l in range(4):
        want = gin1[l] + gin2[l]
        assert torch.isfinite(pair[l]).all()
        assert rel_err(pair[l].float().cpu().numpy(), want) < tol, l
        # tiles only one pooler touches: the same bits; both: one rounding of the sum against one per pooler + one of
        # the sum, i.e. |difference| <= ulp(sum1) / 2 + ulp(sum2) / 2 + ulp(result) (+ the fp32 accumulation order)
        d = _ulps_apart(pair[l], two[l])
        # (per 8 x 8-pixel TILE: the gather contracts a tile's whole list -- both poolers' bins side by side along the
        # k axis of one matrix product -- so where both poolers touch a tile every pixel of it sees another summation
        # order than in the two-call sequence, also a pixel only one of them reaches)
        both = torch.from_numpy(_tile_any(gin1[l] != 0) & _tile_any(gin2[l] != 0)).to(DEV)
        if "cluster1" not in CASES[name] and "cluster2" not in CASES[name]:  # (a combined list of 41..64 entries is walked in
            # PARTS -- partial fp32 sums added in part order -- where the single poolers' shorter lists are walked whole,
            # and the other way round: another summation order)
            assert int(d[~both].max() if (~both).any() else 0) == 0, l
        half = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11  # ulp(x) / 2 <= half * |x|
        # (fp32 accumulation order: 1e-6 of A, not of |sum| -- a tile's list is ONE contraction over all its bins, and
        # where 90 mask-head ROIs pile up on a tile (`second_scans`) thousands of terms cancel to a small sum)
        bound = half * (np.abs(gin1[l]) + np.abs(gin2[l])) + 2 * half * np.abs(want) + 1e-6 * (abs1[l] + abs2[l]) + 2.0 ** -23
        diff = (pair[l].double() - two[l].double()).abs().cpu().numpy()
        assert (diff <= bound).all(), (l, float((diff / bound).max()))
        differ += int((d > 0).sum())
        # and where they differ the single rounding is the closer one (or as close) to the fp64 sum
        if (d > 0).any():
            w = torch.from_numpy(want).to(DEV).double()
            ep, et = (pair[l].double() - w).abs(), (two[l].double() - w).abs()
            m = d > 0
            assert float((ep[m] <= et[m] + 1e-12 * w[m].abs()).float().mean()) > 0.9
    if name != "sparse":
        assert differ > 0  # (the case does exercise tiles both poolers touch)
    # deterministic
    _, again = _run("pair", feats, boxes1, g1, boxes2, g2, dtype)
    assert all(torch.equal(a, b) for a, b in zip(pair, again))


def test_pair_is_refused_outside_the_16_bit_tile_gather_and_launches_nothing():
    rng = np.random.default_rng(1)
    feats, boxes1 = make_inputs(rng, 2, 64, 160, 224, 8)
    boxes2 = [b[:3] for b in boxes1]
    # fp32
    g1 = _nhwc(rng.standard_normal((16, 64, 7, 7)).astype(np.float32), torch.float32)
    g2 = _nhwc(rng.standard_normal((6, 64, 14, 14)).astype(np.float32), torch.float32)
    rc, grads = _run("pair", feats, boxes1, g1, boxes2, g2, torch.float32)
    assert rc == _C.EUNSUPPORTED and all(bool((g == 7.0).all()) for g in grads)
    # the 14 x 14 pooler first: taken since r06 (a list entry of the K-concatenated gather carries its own pooled size)
    g1 = _nhwc(rng.standard_normal((16, 64, 14, 14)).astype(np.float32), torch.bfloat16)
    g2 = _nhwc(rng.standard_normal((6, 64, 7, 7)).astype(np.float32), torch.bfloat16)
    rc, grads = _run("pair", feats, boxes1, g1, boxes2, g2, torch.bfloat16, out1=14, out2=7)
    assert rc == 0, _C.lib().d2amd_last_error().decode()
    _, gin1, _ = oracle_pooler(feats, boxes1, 14, 0, True, grad=g1.float().cpu().numpy())
    _, gin2, _ = oracle_pooler(feats, boxes2, 7, 0, True, grad=g2.float().cpu().numpy())
    for l in range(4):
        assert rel_err(grads[l].float().cpu().numpy(), gin1[l] + gin2[l]) < 2.0 ** -7, l
    # channels not a multiple of 32
    feats40 = [f[:, :40].copy() for f in feats]
    g1 = _nhwc(rng.standard_normal((16, 40, 7, 7)).astype(np.float32), torch.bfloat16)
    g2 = _nhwc(rng.standard_normal((6, 40, 14, 14)).astype(np.float32), torch.bfloat16)
    rc, grads = _run("pair", feats40, boxes1, g1, boxes2, g2, torch.bfloat16)
    assert rc == _C.EUNSUPPORTED and all(bool((g == 7.0).all()) for g in grads)


@pytest.mark.parametrize("pair", [True, False])
def test_chained_poolers_take_the_paired_launch_through_autograd(pair, monkeypatch):
    """ROIPooler x 2 on the same leaves, one backward: with the pairing on, ONE paired gather runs; off
    (D2AMD_POOL_PAIR=0), one gather per pooler."""
    monkeypatch.setattr(P, "_PAIR", pair)
    feats, boxes1, g1, boxes2, g2 = _case("independent", torch.bfloat16)
    names = b"pool_bwd_pair,pool_bwd_staged_r7,pool_bwd_staged_r14"
    _C.lib().d2amd_timing_select(names)
    try:
        xs = [_nhwc(f, torch.bfloat16).requires_grad_(True) for f in feats]
        yb = ROIPooler(7, SCALES, 0, "ROIAlignV2")(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes1])
        ym = ROIPooler(14, SCALES, 0, "ROIAlignV2")(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes2])
        torch.autograd.backward([yb, ym], [g1, g2])
        torch.cuda.synchronize()
        cnt = {}
        for kn in names.split(b","):
            tot, c = ctypes.c_double(0.0), ctypes.c_int(0)
            _C.check(_C.lib().d2amd_timing_read(kn, ctypes.byref(tot), ctypes.byref(c)))
            cnt[kn.decode()] = c.value
    finally:
        _C.lib().d2amd_timing_select(None)
        P._ALIASES.clear()
    assert cnt == ({"pool_bwd_pair": 1, "pool_bwd_staged_r7": 0, "pool_bwd_staged_r14": 0} if pair else
                   {"pool_bwd_pair": 0, "pool_bwd_staged_r7": 1, "pool_bwd_staged_r14": 1}), cnt
    _, want = _run("pair" if pair else "two", feats, boxes1, g1, boxes2, g2, torch.bfloat16)
    assert all(torch.equal(x.grad, w) for x, w in zip(xs, want))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("name", ["subset", "independent", "sparse"])
def test_pool_pair_forward_is_the_two_forwards_bit_for_bit(name, dtype):
    """d2amd_roi_pooler_forward_pair(_box_lists): the same workgroups run the same code in one grid."""
    from detectron2_amd.modeling import pool_pair

    feats, boxes1, _, boxes2, _ = _case(name, torch.bfloat16)
    xs = [_nhwc(f, dtype) for f in feats]
    b1 = [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes1]
    b2 = [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes2]
    pa, pb = ROIPooler(7, SCALES, 0, "ROIAlignV2"), ROIPooler(14, SCALES, 0, "ROIAlignV2")
    _C.lib().d2amd_timing_select(b"pool_fwd_pair")
    try:
        ya, yb = pool_pair(pa, pb, xs, b1, b2)
        torch.cuda.synchronize()
        tot, c = ctypes.c_double(0.0), ctypes.c_int(0)
        _C.check(_C.lib().d2amd_timing_read(b"pool_fwd_pair", ctypes.byref(tot), ctypes.byref(c)))
    finally:
        _C.lib().d2amd_timing_select(None)
    assert c.value == 1  # the paired launch ran (fp32: 16-B vectors of 4 channels, the same kernel)
    wa, wb = pa(xs, b1), pb(xs, b2)
    assert ya.shape == wa.shape and yb.shape == wb.shape and torch.equal(ya, wa) and torch.equal(yb, wb)
    # the (M, 5) route (boxes that are not 16-B aligned fp32 HIP tensors: here fp64 boxes)
    ya2, yb2 = pool_pair(pa, pb, xs, [Boxes(b.tensor.double()) for b in b1], [Boxes(b.tensor.double()) for b in b2])
    assert torch.equal(ya2, wa) and torch.equal(yb2, wb)


def test_pool_pair_backward_is_the_paired_backward_and_falls_back_cleanly():
    from detectron2_amd.modeling import pool_pair

    feats, boxes1, g1, boxes2, g2 = _case("independent", torch.bfloat16)
    b1 = [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes1]
    b2 = [Boxes(torch.from_numpy(b).to(DEV)) for b in boxes2]
    pa, pb = ROIPooler(7, SCALES, 0, "ROIAlignV2"), ROIPooler(14, SCALES, 0, "ROIAlignV2")
    xs = [_nhwc(f, torch.bfloat16).requires_grad_(True) for f in feats]
    ya, yb = pool_pair(pa, pb, xs, b1, b2)
    torch.autograd.backward([ya, yb], [g1, g2])
    _, want = _run("pair", feats, boxes1, g1, boxes2, g2, torch.bfloat16)
    assert all(torch.equal(x.grad, w) for x, w in zip(xs, want))
    # only one of the two results is used: that pooler's plain backward
    for x in xs:
        x.grad = None
    ya, yb = pool_pair(pa, pb, xs, b1, b2)
    yb.backward(g2)
    xs2 = [_nhwc(f, torch.bfloat16).requires_grad_(True) for f in feats]
    pb(xs2, b2).backward(g2)
    assert all(torch.equal(x.grad, y.grad) for x, y in zip(xs, xs2))
    # NCHW features, an empty image list entry, fp32: the separate calls (same values as calling them)
    xs3 = [torch.from_numpy(f).to(DEV).requires_grad_(True) for f in feats]
    ya, yb = pool_pair(pa, pb, xs3, b1, b2)
    assert torch.equal(ya, pa(xs3, b1)) and torch.equal(yb, pb(xs3, b2))
    P._ALIASES.clear()


def test_pool_pair_rois_is_pool_rois_twice():
    from detectron2_amd.modeling import pool_pair_rois

    feats, boxes1, g1, boxes2, g2 = _case("subset", torch.bfloat16)
    pa, pb = ROIPooler(7, SCALES, 0, "ROIAlignV2"), ROIPooler(14, SCALES, 0, "ROIAlignV2")
    r1, r2 = _rois(boxes1), _rois(boxes2)
    xs = [_nhwc(f, torch.bfloat16).requires_grad_(True) for f in feats]
    ya, yb = pool_pair_rois(pa, pb, xs, r1, r2)
    with torch.no_grad():
        assert torch.equal(ya, pa.pool_rois(xs, r1)) and torch.equal(yb, pb.pool_rois(xs, r2))
    torch.autograd.backward([ya, yb], [g1, g2])
    _, want = _run("pair", feats, boxes1, g1, boxes2, g2, torch.bfloat16)
    assert all(torch.equal(x.grad, w) for x, w in zip(xs, want))
    P._ALIASES.clear()


@pytest.mark.parametrize("C,out1,out2,sr", [(320, 7, 14, 0),   # two channel slabs, the second one 64 channels wide
                                            (64, 6, 10, 0),    # other pooled sizes of the two bin classes
                                            (64, (7, 5), (12, 16), 2),  # non-square, fixed sampling ratio
                                            (32, 8, 9, 0)])    # the class boundaries: 8 and 9 bins per axis
def test_pair_other_shapes_vs_oracle(C, out1, out2, sr):
    rng = np.random.default_rng(C + 7)
    img_h, img_w = 256, 320
    feats, boxes1 = make_inputs(rng, 2, C, img_h, img_w, 30)
    _, boxes2 = make_inputs(rng, 2, C, img_h, img_w, 9)
    o1 = (out1, out1) if isinstance(out1, int) else out1
    o2 = (out2, out2) if isinstance(out2, int) else out2
    k1, k2 = sum(len(b) for b in boxes1), sum(len(b) for b in boxes2)
    g1 = _nhwc(rng.standard_normal((k1, C) + o1).astype(np.float32), torch.bfloat16)
    g2 = _nhwc(rng.standard_normal((k2, C) + o2).astype(np.float32), torch.bfloat16)
    L = _C.lib()
    hw = [tuple(f.shape[2:]) for f in feats]
    code = _C.dtype_code(g1)
    cfg = lambda o: (o, tuple(SCALES), sr, True, 2, 5, 224, 4)
    p1, p2 = P._params(cfg(o1), (2, C), hw, code, _C.NHWC), P._params(cfg(o2), (2, C), hw, code, _C.NHWC)
    r1, r2 = _rois(boxes1), _rois(boxes2)
    grads = [torch.full((2, C) + s, 3.0, dtype=torch.bfloat16, device=DEV).contiguous(memory_format=torch.channels_last)
             for s in hw]
    wsb = L.d2amd_roi_pooler_backward_pair_workspace_bytes(ctypes.byref(p1), k1, k2)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    with _C.on_device(grads[0].device):
        _C.check(L.d2amd_roi_pooler_backward_pair(ctypes.byref(p1), _C.ptr(g1), _C.ptr(r1), k1, ctypes.byref(p2), _C.ptr(g2),
                                                  _C.ptr(r2), k2, P._ptr_array(grads), _C.ptr(ws), wsb, _C.stream()))
    torch.cuda.synchronize()

    def oracle_grads(boxes, o, g):
        import oracle
        from test_tile_gather_math import assign_levels_restated
        allb = np.concatenate(boxes)
        rois = _rois(boxes).cpu().numpy()
        lv = assign_levels_restated(allb, 2, 5, 224, 4)
        res = []
        for l, f in enumerate(feats):
            sel = np.nonzero(lv == l)[0]
            res.append(oracle.roi_align_backward(np.ascontiguousarray(g[sel]), rois[sel], f.shape, SCALES[l], sr, True))
        return res

    w1 = oracle_grads(boxes1, o1, g1.float().cpu().numpy())
    w2 = oracle_grads(boxes2, o2, g2.float().cpu().numpy())
    for l in range(4):
        assert rel_err(grads[l].float().cpu().numpy(), w1[l] + w2[l]) < 2.0 ** -7, l


def test_three_chained_poolers_pair_the_first_two_and_add_the_third():
    """box 7x7 + mask 14x14 + a third 14x14 pooler (keypoint head, roi_heads.py:848-877) on the same leaves."""
    rng = np.random.default_rng(21)
    C = 64
    feats, boxes1 = make_inputs(rng, 2, C, 320, 448, 30)
    boxes2 = [b[:10] for b in boxes1]
    boxes3 = [b[5:12] for b in boxes1]
    gs = [rng.standard_normal((sum(len(b) for b in bl), C, o, o)).astype(np.float32)
          for bl, o in ((boxes1, 7), (boxes2, 14), (boxes3, 14))]
    xs = [_nhwc(f, torch.bfloat16).requires_grad_(True) for f in feats]
    ys = [ROIPooler(o, SCALES, 0, "ROIAlignV2")(xs, [Boxes(torch.from_numpy(b).to(DEV)) for b in bl])
          for bl, o in ((boxes1, 7), (boxes2, 14), (boxes3, 14))]
    _C.lib().d2amd_timing_select(b"pool_bwd_pair,pool_bwd_staged_r14")
    try:
        torch.autograd.backward(ys, [_nhwc(g, torch.bfloat16) for g in gs])
        torch.cuda.synchronize()
        cnt = []
        for kn in (b"pool_bwd_pair", b"pool_bwd_staged_r14"):
            tot, c = ctypes.c_double(0.0), ctypes.c_int(0)
            _C.check(_C.lib().d2amd_timing_read(kn, ctypes.byref(tot), ctypes.byref(c)))
            cnt.append(c.value)
    finally:
        _C.lib().d2amd_timing_select(None)
        P._ALIASES.clear()
    assert cnt == [1, 1], cnt
    want = [sum(t) for t in zip(*[oracle_pooler(feats, bl, o, 0, True, grad=torch.from_numpy(g).to(torch.bfloat16).float().numpy())[1]
                                  for bl, o, g in ((boxes1, 7, gs[0]), (boxes2, 14, gs[1]), (boxes3, 14, gs[2]))])]
    for x, w in zip(xs, want):
        assert rel_err(x.grad.float().cpu().numpy(), w) < 2.0 ** -6


def test_backward_plan_bins_ahead_on_another_stream_and_gives_the_same_bits():
    """PairBackwardPlan: the binning issued beside the forward (here: on a side stream, before the backward exists), the
    backward = the gather alone; a plan prepared for OTHER rois, or never prepared, is ignored."""
    from detectron2_amd.modeling import PairBackwardPlan, pool_pair_rois

    feats, boxes1, g1, boxes2, g2 = _case("independent", torch.bfloat16)
    pa, pb = ROIPooler(7, SCALES, 0, "ROIAlignV2"), ROIPooler(14, SCALES, 0, "ROIAlignV2")
    r1, r2 = _rois(boxes1), _rois(boxes2)
    _, want = _run("pair", feats, boxes1, g1, boxes2, g2, torch.bfloat16)

    def run(prepare):
        xs = [_nhwc(f, torch.bfloat16).requires_grad_(True) for f in feats]
        plan = PairBackwardPlan()
        ya, yb = pool_pair_rois(pa, pb, xs, r1, r2, plan=plan)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            prepared = prepare(plan, xs)
        names = b"pool_bwd_pair"
        _C.lib().d2amd_timing_select(names)
        try:
            torch.autograd.backward([ya, yb], [g1, g2])
            torch.cuda.synchronize()
            tot, c = ctypes.c_double(0.0), ctypes.c_int(0)
            _C.check(_C.lib().d2amd_timing_read(names, ctypes.byref(tot), ctypes.byref(c)))
        finally:
            _C.lib().d2amd_timing_select(None)
        assert c.value == 1
        assert all(torch.equal(x.grad, w) for x, w in zip(xs, want))
        return prepared, plan

    prepared, plan = run(lambda plan, xs: plan.prepare(pa, pb, xs, r1, r2))
    assert prepared is True and plan.ready is None  # consumed by the backward
    run(lambda plan, xs: None)                                                   # never prepared
    other = r2.clone()
    prepared, plan = run(lambda plan, xs: plan.prepare(pa, pb, xs, r1, other))   # prepared for another tensor: ignored
    assert prepared is True
    # fp32 features: declined (the plain backward runs)
    xs32 = [_nhwc(f, torch.float32) for f in feats]
    assert PairBackwardPlan().prepare(pa, pb, xs32, r1, r2) is False
    P._ALIASES.clear()


@pytest.mark.parametrize("name", ["subset", "independent", "both_long", "sparse"])
def test_forward_written_records_give_the_same_gradients(name, monkeypatch):
    """r06, d2amd_roi_pooler_forward_pair_records + d2amd_roi_pooler_backward_pair_phase(5): the paired forward's workgroups
    write the backward's per-ROI records and reset its work queues (the workspace is allocated with the forward); outputs
    and gradients are the bits of the plain forward + the one-call backward (`_FWD_RECORDS = False`), also when the same
    forward is differentiated twice in a row (a fresh workspace per forward) and when only one result is used."""
    from detectron2_amd.modeling import pool_pair_rois

    feats, boxes1, g1, boxes2, g2 = _case(name, torch.bfloat16)
    pa, pb = ROIPooler(7, SCALES, 0, "ROIAlignV2"), ROIPooler(14, SCALES, 0, "ROIAlignV2")
    r1, r2 = _rois(boxes1), _rois(boxes2)
    res = {}
    for flag in (False, True, True):
        monkeypatch.setattr(P, "_FWD_RECORDS", flag)
        xs = [_nhwc(f, torch.bfloat16).requires_grad_(True) for f in feats]
        ya, yb = pool_pair_rois(pa, pb, xs, r1, r2)
        assert (ya.grad_fn.prep is not None) == flag
        torch.autograd.backward([ya, yb], [g1, g2])
        out = (ya.detach(), yb.detach(), [x.grad for x in xs])
        if flag in res:
            assert torch.equal(out[0], res[flag][0]) and all(torch.equal(a, b) for a, b in zip(out[2], res[flag][2]))
        res[flag] = out
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert all(torch.equal(a, b) for a, b in zip(res[True][2], res[False][2]))
    _, want = _run("pair", feats, boxes1, g1, boxes2, g2, torch.bfloat16)
    assert all(torch.equal(a, w) for a, w in zip(res[True][2], want))
    # only the second result is used: the prepared workspace is dropped, that pooler's plain backward runs
    monkeypatch.setattr(P, "_FWD_RECORDS", True)
    xs = [_nhwc(f, torch.bfloat16).requires_grad_(True) for f in feats]
    ya, yb = pool_pair_rois(pa, pb, xs, r1, r2)
    yb.backward(g2)
    xs2 = [_nhwc(f, torch.bfloat16).requires_grad_(True) for f in feats]
    pb.pool_rois(xs2, r2).backward(g2)
    assert all(torch.equal(x.grad, y.grad) for x, y in zip(xs, xs2))
    P._ALIASES.clear()
