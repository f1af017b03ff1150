"""NMS from pre-sorted runs (d2amd_nms_runs / d2amd_nms_batched_runs; include/d2amd.h) against the general entries on
the same inputs: identical kept indices, counts and finite counts -- small (merged order instead of the n^2 ranking)
and large (instead of the radix sorts) inputs, runs = categories (the RPN's per-level NMS,
proposal_generator/proposal_utils.py:118-135), runs + class ids (DenseDetector inference,
meta_arch/dense_detector.py:186-260), no categories, rows parked at -inf anywhere, ties inside and between runs, a
run that is NOT in order (the flag + the general-path redo), and the optimistic bitmask pitch of large class-wise
inputs (no torch.unique host sync) with a category beyond it."""
import numpy as np
import pytest
import torch

from detectron2_amd.layers import ops
from detectron2_amd.layers.nms import batched_nms_images

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _boxes(g, n, extent, lo, hi):
    c = torch.rand(n, 2, generator=g) * extent
    wh = lo + torch.rand(n, 2, generator=g) * (hi - lo)
    return torch.cat([c - wh / 2, c + wh / 2], 1)


def _runs_input(seed, lens, extent=1000.0, parked=0.1, ties=True):
    """rows = runs of the given lengths, each in descending score order (scores drawn from a small set when `ties`),
    a fraction parked at -inf (zero boxes) at random positions"""
    g = torch.Generator().manual_seed(seed)
    n = sum(lens)
    boxes = _boxes(g, n, extent, 20, 120)
    parts = []
    for ln in lens:
        s = torch.randint(0, max(ln // 3, 2), (ln,), generator=g).float() / 7 if ties else torch.randn(ln, generator=g)
        parts.append(torch.sort(s, descending=True).values)
    scores = torch.cat(parts) if parts else torch.zeros(0)
    park = torch.rand(n, generator=g) < parked
    scores[park] = float("-inf")
    boxes[park] = 0
    off = np.concatenate([[0], np.cumsum(lens)]).tolist()
    cat = torch.repeat_interleave(torch.arange(len(lens)), torch.tensor(lens))
    return boxes.to(DEV), scores.to(DEV), cat.to(DEV), off


def _same(a, b):
    ka, fa, _ = a
    kb, fb, _ = b
    assert fa == fb
    for x, y in zip(ka, kb):
        assert torch.equal(x, y), (len(x), len(y))


@pytest.mark.parametrize("lens", [(2000, 2000, 2000, 2000, 819), (700, 0, 64, 1, 3000), (5,), (0, 0, 7)])
def test_runs_are_categories_small(lens):
    ins = [_runs_input(10 + i, lens) for i in range(2)]
    off = ins[0][3]
    gen = batched_nms_images([(b, s, c) for b, s, c, _ in ins], 0.7, defer=True)(with_finite=True)
    run = batched_nms_images([(b, s, None) for b, s, _c, _ in ins], 0.7, defer=True, runs=(off, True))(with_finite=True)
    _same(gen, run)
    assert all(len(k) > 0 for k in run[0])


def test_runs_are_categories_large_and_no_categories():
    lens = (6000, 6000, 6000, 2500)  # 20,500 rows: the radix path of the general entry
    b, s, c, off = _runs_input(3, lens, extent=3000.0)
    gen = batched_nms_images([(b, s, c)], 0.5, defer=True)(with_finite=True)
    run = batched_nms_images([(b, s, None)], 0.5, defer=True, runs=(off, True))(with_finite=True)
    _same(gen, run)
    # no categories at all: one segment, order merged from the runs (small and large)
    for lens2, ext in (((1500, 1500, 900), 2000.0), ((9000, 9000), 6000.0)):
        b, s, _c, off = _runs_input(4, lens2, extent=ext)
        gen = batched_nms_images([(b, s, None)], 0.6, defer=True)(with_finite=True)
        run = batched_nms_images([(b, s, None)], 0.6, defer=True, runs=(off, False))(with_finite=True)
        _same(gen, run)


def test_runs_with_class_ids_large_and_small():
    g = torch.Generator().manual_seed(5)
    for lens, ext in (((20000, 20000, 12000, 3000, 700), 9000.0), ((1000, 1000, 1000), 1500.0)):
        b, s, _c, off = _runs_input(6, lens, extent=ext, parked=0.05)
        cls = torch.randint(0, 80, (sum(lens),), generator=g).to(DEV)
        gen = batched_nms_images([(b, s, cls)], 0.5, defer=True)(with_finite=True)
        run = batched_nms_images([(b, s, cls)], 0.5, defer=True, runs=(off, False))(with_finite=True)
        _same(gen, run)


def test_unordered_run_is_detected_and_redone():
    lens = (500, 500, 300)
    b, s, c, off = _runs_input(7, lens, ties=False, parked=0.0)
    s2 = s.clone()
    s2[[10, 400]] = s2[[400, 10]]  # run 0 is no longer in order
    gen = batched_nms_images([(b, s2, c)], 0.7, defer=True)(with_finite=True)
    run = batched_nms_images([(b, s2, None)], 0.7, defer=True, runs=(off, True))(with_finite=True)
    _same(gen, run)
    # the flag itself
    keep, result, _hold = ops._nms_launch(b, s2, None, 0.7, False, runs=(off, True))
    assert int(result[1].item()) & 4
    keep, result, _hold = ops._nms_launch(b, s, None, 0.7, False, runs=(off, True))
    assert int(result[1].item()) == 0


def test_large_classwise_input_needs_no_count_of_the_categories(monkeypatch):
    """n > 16,384 with class ids: launched with the optimistic pitch (no torch.unique / host sync before the launch);
    a category beyond the pitch is detected and redone with the exact size.  Both equal the exactly-sized launch."""
    g = torch.Generator().manual_seed(8)
    n = 30000
    b = _boxes(g, n, 12000.0, 20, 100).to(DEV)
    s = torch.rand(n, generator=g).to(DEV)
    for big in (False, True):
        cls = torch.randint(1, 40, (n,), generator=g)
        if big:
            cls[:20000] = 0  # one category of 20,000 > 16,384
        cls = cls.to(DEV)
        keep, result, _h = ops._nms_launch(b, s, cls, 0.5, False, exact_bound=True)
        exact = keep[:int(result[0].item())]
        calls = []
        real_unique = torch.unique
        monkeypatch.setattr(torch, "unique", lambda *a, **k: (calls.append(1), real_unique(*a, **k))[1])
        got = ops.nms_impl(b, s, cls, 0.5, False)
        monkeypatch.undo()
        assert torch.equal(got, exact)
        assert len(calls) == (1 if big else 0)
        (via_images,) = batched_nms_images([(b, s, cls)], 0.5)
        assert torch.equal(via_images, exact)


@pytest.mark.parametrize("large", [False, True])
def test_gather_in_keep_order(large):
    """the fused gather of the runs entries: dst row j == src row keep[j] for every array, small and large path,
    and after a redo (a run out of order)"""
    lens = (9000, 9000, 4000) if large else (2000, 1500, 64)
    g = torch.Generator().manual_seed(11)
    for broken in (False, True):
        b, s, _c, off = _runs_input(12, lens, extent=5000.0 if large else 1200.0, ties=not broken, parked=0.05)
        if broken:
            s = s.clone()
            s[[3, 900]] = s[[900, 3]]
        cls = torch.randint(0, 30, (sum(lens),), generator=g).to(DEV)
        extra = torch.rand(sum(lens), 3, generator=g).to(DEV)
        done = batched_nms_images([(b, s, cls)], 0.5, defer=True, runs=(off, False), gather=[(b, s, cls, extra)])
        (keep,), _fin, _ = done(with_finite=True)
        (ref,) = batched_nms_images([(b, s, cls)], 0.5)
        assert torch.equal(keep, ref)
        for src, dst in zip((b, s, cls, extra), done.gathered[0]):
            assert torch.equal(dst[:len(keep)], src[keep])
