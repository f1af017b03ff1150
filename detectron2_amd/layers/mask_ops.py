"""paste_masks_in_image -- mirrors detectron2/layers/mask_ops.py:74-147.  One HIP kernel writes
the (N, H, W) bool/uint8 result directly (no grid tensor, no fp32 intermediate, no chunking):
BYTES_PER_FLOAT / GPU_MEM_LIMIT are kept as public names for compatibility only."""
from typing import Tuple

import torch

from .. import _C

__all__ = ["paste_masks_in_image"]

BYTES_PER_FLOAT = 4
GPU_MEM_LIMIT = 1024**3  # unused: the kernel needs no intermediate memory


def paste_masks_in_image(masks: torch.Tensor, boxes, image_shape: Tuple[int, int], threshold: float = 0.5):
    """Paste fixed-resolution masks (Bimg, M, M) into the image at their boxes.

    Args / returns as the reference: boxes is a Boxes or Tensor (Bimg, 4); returns
    (Bimg, H, W) bool, or uint8 when threshold < 0 (values trunc(v * 255))."""
    assert masks.shape[-1] == masks.shape[-2], "Only square mask predictions are supported"
    N = len(masks)
    image_shape = (int(image_shape[0]), int(image_shape[1]))
    if N == 0:
        return masks.new_empty((0,) + image_shape, dtype=torch.uint8)
    if not isinstance(boxes, torch.Tensor):
        boxes = boxes.tensor
    assert len(boxes) == N, boxes.shape
    _C.require_gpu(masks, boxes, op="paste_masks_in_image")
    img_h, img_w = image_shape
    m = masks.detach()
    if not m.dtype.is_floating_point:
        m = m.float()
    if m.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        m = m.float()
    m = m.contiguous()
    b = boxes.detach().float().contiguous()
    out = torch.empty((N, img_h, img_w), dtype=torch.uint8, device=m.device)
    with _C.on_device(m.device):
        _C.check(_C.lib().d2amd_paste_masks(_C.ptr(m), _C.ptr(b), N, m.shape[1], m.shape[2], img_h, img_w,
                                            float(threshold), _C.ptr(out), _C.dtype_code(m), _C.stream()))
    return out.view(torch.bool) if threshold >= 0 else out
