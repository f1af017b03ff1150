"""GPU parity of PolygonMasks.crop_and_resize (csrc/polygon_masks.hip) against the oracle's
restatement of the reference pipeline -- rasterize_polygons_within_box (detectron2/structures/masks.py:39-86) on top of
the restated pycocotools rasteriser (oracle/d2_oracle.c orc_poly_to_mask; parity with pycocotools itself is unpinned:
it is not available here).  Bar: bit-exact."""
import numpy as np
import pytest
import torch

import oracle
from detectron2_amd.structures import PolygonMasks

pytestmark = pytest.mark.gpu
DEV = "cuda"


def star(rng, cx, cy, r, n):
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = r * rng.uniform(0.4, 1.0, n)
    return np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1).reshape(-1)


def instances(rng, n, size=400):
    out = []
    for i in range(n):
        polys = [star(rng, rng.uniform(0.2, 0.8) * size, rng.uniform(0.2, 0.8) * size, rng.uniform(10, 0.4 * size),
                      int(rng.integers(3, 40))) for _ in range(int(rng.integers(1, 4)))]
        out.append(polys)
    return out


def want(polys, boxes, M, index=None):
    idx = range(len(boxes)) if index is None else index
    return np.stack([oracle.rasterize_polygons_within_box(polys[i], b, M) for i, b in zip(idx, boxes)])


@pytest.mark.parametrize("M", [28, 14, 7])
def test_polygon_crop_and_resize_bit_exact(M):
    rng = np.random.default_rng(100 + M)
    polys = instances(rng, 24)
    pm = PolygonMasks(polys)
    xy = rng.uniform(-20, 300, (24, 2))
    wh = np.exp(rng.uniform(np.log(4), np.log(420), (24, 2)))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    boxes[0] = [0, 0, 400, 400]
    boxes[1, 2:] = boxes[1, :2] + 0.05       # extent below 0.1: the ratio is mask_size / 0.1
    boxes[2] = [100, 100, 100, 180]          # zero width
    got = pm.crop_and_resize(torch.from_numpy(boxes).to(DEV), M)
    assert got.dtype == torch.bool and tuple(got.shape) == (24, M, M) and got.is_cuda
    assert np.array_equal(got.cpu().numpy(), want(polys, boxes, M))


def test_polygon_crop_indexed_and_reference_known_answer():
    """`gt_masks[idx].crop_and_resize(boxes, M)` without the re-packing; plus the reference's own known answer
    (tests/structures/test_masks.py:31-38): an integer box polygon rasterised at its own size fills [x0, x1) x [y0, y1)."""
    rng = np.random.default_rng(3)
    polys = instances(rng, 9)
    pm = PolygonMasks(polys)
    idx = rng.integers(0, 9, 40)
    xy = rng.uniform(0, 250, (40, 2))
    wh = rng.uniform(10, 300, (40, 2))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    bt, it = torch.from_numpy(boxes).to(DEV), torch.from_numpy(idx).to(DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    got = pm.crop_and_resize_indexed(bt, it, 28, status)
    assert int(status.item()) == 0
    assert np.array_equal(got.cpu().numpy(), want(polys, boxes, 28, idx))
    assert torch.equal(got, pm[it].crop_and_resize(bt, 28))  # the reference's two-step form
    it[5] = 9
    bad = pm.crop_and_resize_indexed(bt, it, 28, status)
    assert int(status.item()) == 1 and not bool(bad[5].any()) and torch.equal(bad[6:], got[6:])
    for box in ([1, 0, 4, 4], [1, 1, 3, 4]):
        b = np.array(box, np.float64)
        m = PolygonMasks([[b[[0, 1, 2, 1, 2, 3, 0, 3]]]]).crop_and_resize(torch.tensor([[0.0, 0, 4, 4]], device=DEV), 4)[0].cpu().numpy()
        exp = np.zeros((4, 4), bool)
        exp[box[1]:box[3], box[0]:box[2]] = True
        assert np.array_equal(m, exp)


def test_polygon_masks_container_contract():
    """Constructor errors, indexing and the empty case of the reference container (masks.py:270-395)."""
    with pytest.raises(ValueError):
        PolygonMasks("nope")
    with pytest.raises(ValueError):
        PolygonMasks([[np.array([0.0, 0, 1, 1])]])  # fewer than 3 points
    rng = np.random.default_rng(1)
    pm = PolygonMasks(instances(rng, 5) + [[]])
    assert len(pm) == 6 and pm.device == torch.device("cpu") and pm.to("cuda") is pm
    assert len(pm[2]) == 1 and len(pm[1:4]) == 3 and len(pm[[0, 5]]) == 2
    assert len(pm[torch.tensor([True, False, True, False, False, True])]) == 3
    assert pm.nonempty().tolist() == [True] * 5 + [False]
    bb = pm.get_bounding_boxes().tensor
    # (an instance without polygons: [inf, inf, 0, 0], like the reference's masks.py:327-336)
    assert bb.shape == (6, 4) and bb[5].tolist() == [float("inf"), float("inf"), 0.0, 0.0]
    assert torch.all(bb[:5, 2:] > bb[:5, :2])
    assert repr(pm) == "PolygonMasks(num_instances=6)"
    empty = PolygonMasks([]).crop_and_resize(torch.zeros(0, 4, device=DEV), 28)
    assert tuple(empty.shape) == (0, 28, 28) and empty.dtype == torch.bool
    # an instance without polygons rasterises to zeros (polygons_to_bitmask, masks.py:29-31)
    z = pm.crop_and_resize(torch.tensor([[0.0, 0, 50, 50]] * 6, device=DEV), 14)
    assert not bool(z[5].any())
