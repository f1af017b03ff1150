"""GPU parity of BitMasks.crop_and_resize (detectron2_amd/csrc/mask_targets.hip):
bit-exact against the reference pipeline restated with the oracle -- masks.to(float32) -> torchvision
roi_align((M, M), 1.0, 0, aligned=True) (oracle.roi_align_forward, pinned by the reference's known answers)
-> `>= 0.5` (detectron2/structures/masks.py:193-224)."""
import numpy as np
import pytest
import torch

import oracle
from detectron2_amd.layers import ROIAlign
from detectron2_amd.structures import BitMasks

pytestmark = pytest.mark.gpu
DEV = "cuda"


def blob_masks(rng, g, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.zeros((g, h, w), bool)
    for i in range(g):
        cy, cx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.2, 0.8) * w
        ry, rx = rng.uniform(0.05, 0.3) * h, rng.uniform(0.05, 0.3) * w
        m[i] = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        m[i] ^= rng.random((h, w)) < 0.02  # speckle: bins whose mean sits near 0.5
    return m


def reference_pipeline(masks, boxes, M):
    g = masks.shape[0]
    rois = np.concatenate([np.arange(g, dtype=np.float32)[:, None], boxes], 1)
    out = oracle.roi_align_forward(masks.astype(np.float32)[:, None], rois, (M, M), 1.0, 0, True)
    return out[:, 0] >= 0.5


@pytest.mark.parametrize("g,h,w,M", [(6, 97, 131, 28), (3, 64, 48, 14), (2, 33, 40, 7)])
def test_crop_and_resize_bit_exact(g, h, w, M):
    rng = np.random.default_rng(g * 100 + M)
    masks = blob_masks(rng, g, h, w)
    xy = rng.uniform(-5, [w * 0.6, h * 0.6], (g, 2))
    wh = rng.uniform(3, [w * 0.7, h * 0.7], (g, 2))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    boxes[0] = [0, 0, w, h]                    # whole image
    boxes[-1, 2:] = boxes[-1, :2] + 0.4        # box smaller than a pixel
    got = BitMasks(torch.from_numpy(masks).to(DEV)).crop_and_resize(torch.from_numpy(boxes).to(DEV), M)
    assert got.dtype == torch.bool and tuple(got.shape) == (g, M, M)
    assert np.array_equal(got.cpu().numpy(), reference_pipeline(masks, boxes, M))


def test_crop_and_resize_full_size_equals_roialign_path():
    """BASELINE configs[1]: 16 ground-truth masks of an 800 x 1333 image -> 28 x 28.  Besides the oracle, the
    op agrees with this package's own ROIAlign (fp32 masks) wherever the mean is not within 1e-4 of 0.5."""
    rng = np.random.default_rng(9)
    g, h, w, M = 16, 800, 1333, 28
    masks = blob_masks(rng, g, h, w)
    s = np.exp(rng.uniform(np.log(16), np.log(512), g))
    c = rng.uniform([0, 0], [w, h], (g, 2))
    boxes = np.concatenate([c - s[:, None] / 2, c + s[:, None] / 2], 1).astype(np.float32)
    mt, bt = torch.from_numpy(masks).to(DEV), torch.from_numpy(boxes).to(DEV)
    got = BitMasks(mt).crop_and_resize(bt, M).cpu().numpy()
    assert np.array_equal(got, reference_pipeline(masks, boxes, M))
    rois = torch.cat([torch.arange(g, device=DEV, dtype=torch.float32)[:, None], bt], 1)
    mean = ROIAlign((M, M), 1.0, 0, aligned=True)(mt.float()[:, None], rois)[:, 0].cpu().numpy()
    sure = np.abs(mean - 0.5) > 1e-4
    assert np.array_equal(got[sure], (mean >= 0.5)[sure])


def test_crop_and_resize_empty():
    out = BitMasks(torch.zeros(0, 20, 30, dtype=torch.bool, device=DEV)).crop_and_resize(torch.zeros(0, 4, device=DEV), 28)
    assert tuple(out.shape) == (0, 28, 28) and out.dtype == torch.bool


def test_crop_and_resize_indexed_equals_indexing_then_crop():
    """`gt_masks[sampled_targets].crop_and_resize(proposal_boxes, M)` (roi_heads.py:280-291 + mask_head.py:65-67)
    without the indexed copy: bit-identical to indexing first, both on the device and against the oracle pipeline;
    an index outside the masks is flagged and yields zeros."""
    rng = np.random.default_rng(21)
    g, h, w, M, n = 5, 120, 161, 28, 37
    masks = blob_masks(rng, g, h, w)
    idx = rng.integers(0, g, n)
    xy = rng.uniform(-5, [w * 0.6, h * 0.6], (n, 2))
    wh = rng.uniform(3, [w * 0.7, h * 0.7], (n, 2))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    bm = BitMasks(torch.from_numpy(masks).to(DEV))
    bt, it = torch.from_numpy(boxes).to(DEV), torch.from_numpy(idx).to(DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    got = bm.crop_and_resize_indexed(bt, it, M, status)
    assert got.dtype == torch.bool and tuple(got.shape) == (n, M, M) and int(status.item()) == 0
    assert torch.equal(got, bm[it].crop_and_resize(bt, M))
    assert np.array_equal(got.cpu().numpy(), reference_pipeline(masks[idx], boxes, M))
    it[3] = g  # out of range
    bad = bm.crop_and_resize_indexed(bt, it, M, status)
    assert int(status.item()) == 1 and not bool(bad[3].any()) and torch.equal(bad[4:], got[4:])
    assert tuple(bm.crop_and_resize_indexed(bt[:0], it[:0], M).shape) == (0, M, M)


@pytest.mark.parametrize("pattern", ["checkerboard", "stripes", "half_plane"])
def test_crop_and_resize_means_exactly_at_threshold(pattern):
    """Masks whose bin means sit EXACTLY on 0.5 (or a rounding error away from it) for most bins, with large boxes
    (hundreds of samples per bin): the two-tier kernel (parallel partial sums, sequential re-evaluation near 0.5)
    must reproduce the reference's sequential fp32 sum bit for bit."""
    rng = np.random.default_rng(31)
    g, h, w, M = 6, 400, 640, 28
    yy, xx = np.mgrid[0:h, 0:w]
    if pattern == "checkerboard":
        base = ((yy + xx) % 2).astype(bool)
    elif pattern == "stripes":
        base = (xx % 2).astype(bool)
    else:
        base = xx >= w // 2
    masks = np.repeat(base[None], g, 0)
    boxes = np.array([[0, 0, w, h], [10.5, 7.25, 610.5, 390.25], [0, 0, 560, 392], [33.3, 21.7, 500.1, 377.9],
                      [w / 2 - 112, 50, w / 2 + 112, 274], [w / 2 - 14, 10, w / 2 + 14, 38]], np.float32)
    got = BitMasks(torch.from_numpy(masks).to(DEV)).crop_and_resize(torch.from_numpy(boxes).to(DEV), M)
    assert np.array_equal(got.cpu().numpy(), reference_pipeline(masks, boxes, M))


def test_crop_and_resize_batch_equals_per_image_loop():
    """The whole batch in one launch (images with different numbers of masks and boxes, with and without an index)
    == the reference's per-image loop + cat, and == the oracle pipeline."""
    from detectron2_amd.structures import crop_and_resize_batch

    rng = np.random.default_rng(8)
    h, w, M = 150, 210, 28
    counts_m, counts_b = [4, 1, 7], [9, 0, 30]
    masks = [blob_masks(rng, g, h, w) for g in counts_m]
    boxes, idx = [], []
    for g, n in zip(counts_m, counts_b):
        s = np.exp(rng.uniform(np.log(3), np.log(180), (n, 1))) * rng.uniform(0.6, 1.6, (n, 2))
        c = rng.uniform([0, 0], [w, h], (n, 2))
        boxes.append(np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32))
        idx.append(rng.integers(0, g, n))
    bms = [BitMasks(torch.from_numpy(m).to(DEV)) for m in masks]
    bt = [torch.from_numpy(b).to(DEV) for b in boxes]
    it = [torch.from_numpy(i).to(DEV) for i in idx]
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    got = crop_and_resize_batch(bms, bt, M, it, status)
    want = torch.cat([m.crop_and_resize_indexed(b, i, M) for m, b, i in zip(bms, bt, it)])
    assert int(status.item()) == 0 and torch.equal(got, want)
    ref = np.concatenate([reference_pipeline(m[i], b, M) for m, b, i in zip(masks, boxes, idx) if len(b)])
    assert np.array_equal(got.cpu().numpy(), ref)
    # no index: box g of an image crops its mask g
    bt2 = [torch.from_numpy(b[:g] if len(b) >= g else np.tile(np.array([[1, 2, 30, 40]], np.float32), (g, 1))).to(DEV)
           for b, g in zip(boxes, counts_m)]
    got2 = crop_and_resize_batch(bms, bt2, M)
    assert torch.equal(got2, torch.cat([m.crop_and_resize(b, M) for m, b in zip(bms, bt2)]))


@pytest.mark.parametrize("h,w,box", [
    (1900, 48, [2.0, 3.0, 44.0, 1890.0]),     # bin rows of > 62 pixel rows: the band's row table does not fit
    (40, 2300, [1.0, 2.0, 2290.0, 37.0]),     # a band wider than 2,048 columns
    (1900, 2300, [5.0, 5.0, 2200.0, 1850.0]), # both
])
def test_crop_and_resize_rois_beyond_the_band_tables(h, w, box):
    """ROIs taller than ~1,800 px or wider than 2,048 px skip the tier-1 tables: every bin goes through the
    sequential tier (the path that had an unsynchronised read of the per-bin flags); repeated so that workgroups
    follow ones that left the flags decided."""
    rng = np.random.default_rng(h + w)
    masks = blob_masks(rng, 2, h, w)
    boxes = np.array([box, [3.0, 3.0, 30.0, 30.0]], np.float32)  # the second ROI takes the tier-1 path
    bm = BitMasks(torch.from_numpy(masks).to(DEV))
    bt = torch.from_numpy(boxes).to(DEV)
    want = reference_pipeline(masks, boxes, 28)
    for _ in range(3):
        assert np.array_equal(bm.crop_and_resize(bt, 28).cpu().numpy(), want)
