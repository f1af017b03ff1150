"""BitMasks -- the part of detectron2/structures/masks.py:88-224 on the hot path: the (G, H, W) bool holder
and `crop_and_resize`, the Mask R-CNN training-target rasteriser.  One fused HIP kernel
(d2amd_bitmask_crop_and_resize) instead of `to(float32)` + ROIAlign + `>= 0.5`."""
import ctypes
from typing import List, Optional

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
