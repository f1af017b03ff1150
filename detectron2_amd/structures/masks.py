"""BitMasks -- the part of detectron2/structures/masks.py:88-224 on the hot path: the (G, H, W) bool holder
and `crop_and_resize`, the Mask R-CNN training-target rasteriser.  One fused HIP kernel
(d2amd_bitmask_crop_and_resize) instead of `to(float32)` + ROIAlign + `>= 0.5`."""
import ctypes
from typing import List, Optional

import numpy as np
import torch

from .. import _C


class BitMasks:
    def __init__(self, tensor):
        if isinstance(tensor, torch.Tensor):
            tensor = tensor.to(torch.bool)
        else:
            tensor = torch.as_tensor(tensor, dtype=torch.bool, device=torch.device("cpu"))
        assert tensor.dim() == 3, tensor.size()
        self.image_size = tensor.shape[1:]
        self.tensor = tensor

    def to(self, *args, **kwargs):
        return BitMasks(self.tensor.to(*args, **kwargs))

    @property
    def device(self):
        return self.tensor.device

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        if isinstance(item, int):
            return BitMasks(self.tensor[item].unsqueeze(0))
        m = self.tensor[item]
        assert m.dim() == 3, "Indexing on BitMasks with {} returns a tensor with shape {}!".format(item, m.shape)
        return BitMasks(m)

    def crop_and_resize(self, boxes: torch.Tensor, mask_size: int) -> torch.Tensor:
        """Crop each bitmask by its box and resize to (mask_size, mask_size) -> bool (N, mask_size, mask_size)."""
        assert len(boxes) == len(self), "{} != {}".format(len(boxes), len(self))
        _C.require_gpu(self.tensor, op="BitMasks.crop_and_resize")
        m = self.tensor.contiguous().view(torch.uint8)
        b = boxes.detach().to(device=m.device, dtype=torch.float32).contiguous()
        g, h, w = m.shape
        out = torch.empty((g, mask_size, mask_size), dtype=torch.uint8, device=m.device)
        if g:
            with _C.on_device(m.device):
                _C.check(_C.lib().d2amd_bitmask_crop_and_resize(_C.ptr(m), _C.ptr(b), g, h, w, int(mask_size),
                                                                _C.ptr(out), _C.stream()))
        return out.view(torch.bool)

    def crop_and_resize_indexed(self, boxes: torch.Tensor, mask_index: torch.Tensor, mask_size: int,
                                status: torch.Tensor = None) -> torch.Tensor:
        """`self[mask_index].crop_and_resize(boxes, mask_size)` without the (len(boxes), H, W) indexed copy: box i
        crops mask mask_index[i].  This is what Mask R-CNN training does per image with the matched ground truth
        of every sampled proposal (roi_heads.py:280-291 then mask_head.py:65-67).  An index outside
        [0, len(self)) sets bit 0 of `status` (int32[1] on the device, optional; read it where a sync is due)."""
        assert len(boxes) == len(mask_index), "{} != {}".format(len(boxes), len(mask_index))
        _C.require_gpu(self.tensor, op="BitMasks.crop_and_resize_indexed")
        m = self.tensor.contiguous().view(torch.uint8)
        b = boxes.detach().to(device=m.device, dtype=torch.float32).contiguous()
        idx = mask_index.detach().to(device=m.device, dtype=torch.int64).contiguous()
        g, h, w = m.shape
        n = b.shape[0]
        out = torch.empty((n, mask_size, mask_size), dtype=torch.uint8, device=m.device)
        if n:
            assert g > 0, "indexing an empty BitMasks"
            with _C.on_device(m.device):
                _C.check(_C.lib().d2amd_bitmask_crop_and_resize_indexed(_C.ptr(m), g, _C.ptr(b), _C.ptr(idx), n, h, w,
                                                                        int(mask_size), _C.ptr(out), _C.ptr(status),
                                                                        _C.stream()))
        return out.view(torch.bool)


def crop_and_resize_batch(gt_masks: List["BitMasks"], boxes: List[torch.Tensor], mask_size: int,
                          mask_index: Optional[List[torch.Tensor]] = None, status: torch.Tensor = None) -> torch.Tensor:
    """`torch.cat([m[idx].crop_and_resize(b, M) for m, b, idx in ...])` -- the loop `mask_rcnn_loss` runs over the
    images of a batch (mask_head.py:57-77) -- in ONE launch: gt_masks[i] are image i's BitMasks (all of one H x W),
    boxes[i] its (n_i, 4) boxes, mask_index[i] the mask every box crops (None: box g crops mask g)."""
    n_img = len(gt_masks)
    assert n_img == len(boxes) and (mask_index is None or len(mask_index) == n_img)
    assert 0 < n_img <= 64, n_img
    ms = [m.tensor.contiguous().view(torch.uint8) for m in gt_masks]
    _C.require_gpu(*ms, op="crop_and_resize_batch")
    dev = ms[0].device
    h, w = ms[0].shape[1:]
    assert all(tuple(m.shape[1:]) == (h, w) for m in ms), "all images of the batch share one (padded) size"
    bs = [b.detach().to(device=dev, dtype=torch.float32).contiguous() for b in boxes]
    ix = None if mask_index is None else [i.detach().to(device=dev, dtype=torch.int64).contiguous() for i in mask_index]
    nb = [int(b.shape[0]) for b in bs]
    if ix is None:
        assert all(n == int(m.shape[0]) for n, m in zip(nb, ms)), "without an index every mask needs exactly one box"
    else:
        assert all(int(i.shape[0]) == n for i, n in zip(ix, nb))
    out = torch.empty((sum(nb), mask_size, mask_size), dtype=torch.uint8, device=dev)
    if sum(nb):
        vp = lambda ts: (ctypes.c_void_p * n_img)(*[t.data_ptr() for t in ts])
        ci = lambda vs: (ctypes.c_int * n_img)(*vs)
        with _C.on_device(dev):
            _C.check(_C.lib().d2amd_bitmask_crop_and_resize_batch(
                n_img, vp(ms), ci([int(m.shape[0]) for m in ms]), vp(bs), vp(ix) if ix is not None else None, ci(nb),
                int(h), int(w), int(mask_size), _C.ptr(out), _C.ptr(status), _C.stream()))
    return out.view(torch.bool)


class PolygonMasks:
    """The polygon mask container of detectron2/structures/masks.py:265-420 (COCO's default mask format): a list of
    instances, each a list of flat float64 polygons [x0, y0, x1, y1, ...], held on the HOST like the reference's
    (`to()` is a no-op, `device` is cpu).  What differs is `crop_and_resize` -- the Mask R-CNN training-target
    rasteriser: the reference rasterises instance by instance on the CPU (pycocotools) and copies the result to the
    device; here the polygons are packed once per container into three device arrays and ONE kernel
    (d2amd_polygon_crop_and_resize) rasterises all boxes -- with an optional per-box instance index, so that
    `gt_masks[matched_idx].crop_and_resize(proposal_boxes, M)` needs no re-packing of the selected polygons."""

    def __init__(self, polygons):
        if not isinstance(polygons, list):
            raise ValueError("Cannot create PolygonMasks: Expect a list of list of polygons per image. "
                             "Got '{}' instead.".format(type(polygons)))
        self.polygons: List[List[np.ndarray]] = [self._instance(inst) for inst in polygons]
        self._packed = {}  # device -> (coords, poly_offsets, inst_offsets)

    @staticmethod
    def _instance(polys):
        if not isinstance(polys, list):
            raise ValueError("Cannot create polygons: Expect a list of polygons per instance. "
                             "Got '{}' instead.".format(type(polys)))
        out = []
        for t in polys:
            a = np.asarray(t.cpu().numpy() if isinstance(t, torch.Tensor) else t).astype("float64")
            if len(a) % 2 != 0 or len(a) < 6:
                raise ValueError(f"Cannot create a polygon from {len(a)} coordinates.")
            out.append(a)
        return out

    def to(self, *args, **kwargs) -> "PolygonMasks":
        return self

    @property
    def device(self) -> torch.device:
        return torch.device("cpu")

    def __len__(self) -> int:
        return len(self.polygons)

    def __iter__(self):
        return iter(self.polygons)

    def __repr__(self) -> str:
        return self.__class__.__name__ + "(num_instances={})".format(len(self.polygons))

    def __getitem__(self, item) -> "PolygonMasks":
        """int -> one instance; slice / list[int] / int64 vector / bool vector -> the selected instances."""
        if isinstance(item, int):
            chosen = [self.polygons[item]]
        elif isinstance(item, slice):
            chosen = self.polygons[item]
        elif isinstance(item, list):
            chosen = [self.polygons[i] for i in item]
        elif isinstance(item, torch.Tensor):
            if item.dtype == torch.bool:
                assert item.dim() == 1, item.shape
                ids = item.nonzero().squeeze(1).cpu().numpy().tolist()
            elif item.dtype in (torch.int32, torch.int64):
                ids = item.cpu().numpy().tolist()
            else:
                raise ValueError("Unsupported tensor dtype={} for indexing!".format(item.dtype))
            chosen = [self.polygons[i] for i in ids]
        else:
            raise ValueError("Unsupported index {} for PolygonMasks".format(type(item)))
        return PolygonMasks(chosen)

    def nonempty(self) -> torch.Tensor:
        return torch.from_numpy(np.asarray([1 if len(inst) > 0 else 0 for inst in self.polygons], dtype=bool))

    def get_bounding_boxes(self):
        """Tight boxes around the polygons (masks.py:322-336: float32; the minimum starts at +inf and the maximum at
        0, so an instance without polygons gets [inf, inf, 0, 0] like the reference's)."""
        from .boxes import Boxes

        boxes = torch.zeros(len(self.polygons), 4, dtype=torch.float32)
        for i, inst in enumerate(self.polygons):
            lo, hi = np.full(2, np.inf, np.float32), np.zeros(2, np.float32)
            if inst:
                xy = np.concatenate([p.reshape(-1, 2) for p in inst]).astype(np.float32)
                lo, hi = np.minimum(lo, xy.min(0)), np.maximum(hi, xy.max(0))
            boxes[i] = torch.from_numpy(np.concatenate([lo, hi]))
        return Boxes(boxes)

    @staticmethod
    def cat(polymasks_list):
        """masks.py:446-465 (what Instances.cat calls for gt_masks)."""
        import itertools

        assert isinstance(polymasks_list, (list, tuple))
        assert len(polymasks_list) > 0
        assert all(isinstance(pm, PolygonMasks) for pm in polymasks_list)
        return type(polymasks_list[0])(list(itertools.chain.from_iterable(pm.polygons for pm in polymasks_list)))

    def area(self) -> torch.Tensor:
        """Shoelace area per instance (masks.py:422-441)."""
        out = []
        for inst in self.polygons:
            a = 0.0
            for p in inst:
                x, y = p[0::2], p[1::2]
                a += 0.5 * np.abs(np.dot(x, np.roll(y, 1)) - np.dot(y, np.roll(x, 1)))
            out.append(a)
        return torch.tensor(out)

    # ---- the hot-path part ---------------------------------------------------------------------------------
    def _pack(self, device):
        """(coords f64, poly_offsets i64, inst_offsets i64) on `device`, built once per container and device."""
        key = str(device)
        if key not in self._packed:
            flat = [p for inst in self.polygons for p in inst]
            coords = np.concatenate(flat) if flat else np.zeros(0, np.float64)
            poly_off = np.zeros(len(flat) + 1, np.int64)
            np.cumsum([len(p) for p in flat], out=poly_off[1:])
            inst_off = np.zeros(len(self.polygons) + 1, np.int64)
            np.cumsum([len(inst) for inst in self.polygons], out=inst_off[1:])
            self._packed[key] = tuple(torch.from_numpy(a).to(device) for a in (coords, poly_off, inst_off))
        return self._packed[key]

    def crop_and_resize(self, boxes: torch.Tensor, mask_size: int) -> torch.Tensor:
        """(N, mask_size, mask_size) bool on boxes.device: instance i rasterised inside boxes[i] (masks.py:396-420)."""
        assert len(boxes) == len(self), "{} != {}".format(len(boxes), len(self))
        return self._crop(boxes, None, mask_size, None)

    def crop_and_resize_indexed(self, boxes: torch.Tensor, index: torch.Tensor, mask_size: int,
                                status: torch.Tensor = None) -> torch.Tensor:
        """`self[index].crop_and_resize(boxes, mask_size)` without building the selected container: box k is
        rasterised from instance index[k].  An index outside [0, len(self)) sets bit 0 of `status` (int32[1] on the
        device, optional)."""
        assert len(boxes) == len(index), "{} != {}".format(len(boxes), len(index))
        return self._crop(boxes, index, mask_size, status)

    def _crop(self, boxes, index, mask_size, status):
        _C.require_gpu(boxes, op="PolygonMasks.crop_and_resize")
        dev = boxes.device
        n = int(boxes.shape[0])
        out = torch.empty((n, mask_size, mask_size), dtype=torch.uint8, device=dev)
        if n == 0:
            return out.view(torch.bool)
        coords, poly_off, inst_off = self._pack(dev)
        b = boxes.detach().to(dtype=torch.float32).contiguous()
        idx = None if index is None else index.detach().to(device=dev, dtype=torch.int64).contiguous()
        with _C.on_device(dev):
            _C.check(_C.lib().d2amd_polygon_crop_and_resize(_C.ptr(coords), _C.ptr(poly_off), _C.ptr(inst_off), len(self),
                                                            _C.ptr(b), _C.ptr(idx), n, int(mask_size), _C.ptr(out),
                                                            _C.ptr(status), _C.stream()))
        return out.view(torch.bool)
