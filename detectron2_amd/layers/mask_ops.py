"""Mask pasting -- mirrors detectron2/layers/mask_ops.py: `paste_masks_in_image` (:74-147) is one HIP kernel that
writes the (N, H, W) bool / uint8 result directly (no grid tensor, no fp32 intermediate, no chunking; registered as
torch.ops.d2amd.paste_masks, so the function stays `torch.jit.script`-able with bit-identical output,
tests/layers/test_mask_ops.py:156-165).  BYTES_PER_FLOAT / GPU_MEM_LIMIT are kept as public names for compatibility
only.  The module's other public helpers keep their reference behaviour: `pad_masks`, `scale_boxes`
(:219-262), `_paste_masks_tensor_shape` (:264-275) and the Detectron1-era `paste_mask_in_image_old` (:155-212, host
side, PIL)."""
from typing import Tuple

import numpy as np
import torch

from . import ops as _ops  # noqa: F401  (registers torch.ops.d2amd.paste_masks)

__all__ = ["paste_masks_in_image"]

BYTES_PER_FLOAT = 4
GPU_MEM_LIMIT = 1024**3  # unused: the kernel needs no intermediate memory


def paste_masks_in_image(masks: torch.Tensor, boxes: torch.Tensor, image_shape: Tuple[int, int],
                         threshold: float = 0.5):
    """Paste fixed-resolution masks (Bimg, M, M) into the image at their boxes.

    Args / returns as the reference: boxes is a Boxes or Tensor (Bimg, 4); returns (Bimg, H, W) bool, or uint8 when
    threshold < 0 (values trunc(v * 255)).  Device tensors follow the reference's device path (the whole image is
    sampled, mask_ops.py:116-119)."""
    assert masks.shape[-1] == masks.shape[-2], "Only square mask predictions are supported"
    N = len(masks)
    if N == 0:
        return masks.new_empty((0,) + image_shape, dtype=torch.uint8)
    if not isinstance(boxes, torch.Tensor):
        boxes = boxes.tensor
    assert len(boxes) == N, boxes.shape
    return torch.ops.d2amd.paste_masks(masks, boxes, int(image_shape[0]), int(image_shape[1]), threshold)


def paste_mask_in_image_old(mask, box, img_h, img_w, threshold):
    """One mask (Hmask, Wmask) pasted the Detectron1 way (mask_ops.py:155-212): the box is truncated to integer pixel
    coordinates, the mask is resized with PIL's bilinear filter to (x1 - x0 + 1, y1 - y0 + 1) SAMPLES and copied into
    a zero (img_h, img_w) uint8 plane; `> threshold` (or x255 when threshold < 0).  Host side by nature (PIL); kept
    for the callers that still use it with `pad_masks` / `scale_boxes`."""
    from PIL import Image

    x0, y0, x1, y1 = (int(v) for v in box.to(dtype=torch.int32).tolist())
    cols, rows = x1 - x0 + 1, y1 - y0 + 1  # numbers of pixel samples, not geometric sizes
    resized = np.asarray(Image.fromarray(mask.cpu().numpy()).resize((cols, rows), resample=Image.BILINEAR))
    if threshold >= 0:
        resized = torch.from_numpy(np.array(resized > threshold, dtype=np.uint8))
    else:
        resized = torch.from_numpy(resized * 255).to(torch.uint8)
    plane = torch.zeros((img_h, img_w), dtype=torch.uint8)
    cx0, cx1, cy0, cy1 = max(x0, 0), min(x1 + 1, img_w), max(y0, 0), min(y1 + 1, img_h)
    plane[cy0:cy1, cx0:cx1] = resized[cy0 - y0:cy1 - y0, cx0 - x0:cx1 - x0]
    return plane


def pad_masks(masks, padding):
    """(B, M, M) -> ((B, M + 2p, M + 2p) with a zero border of `padding` cells, (M + 2p) / M)."""
    m = masks.shape[-1]
    size = m + 2 * padding
    padded = masks.new_zeros((masks.shape[0], size, size))
    padded[:, padding:size - padding, padding:size - padding] = masks
    return padded, float(size) / m


def scale_boxes(boxes, scale):
    """(B, 4) xyxy boxes scaled by `scale` about their centres."""
    centre = (boxes[:, :2] + boxes[:, 2:]) * 0.5
    half = (boxes[:, 2:] - boxes[:, :2]) * 0.5 * scale
    return torch.cat([centre - half, centre + half], dim=1)


@torch.jit.script_if_tracing
def _paste_masks_tensor_shape(masks: torch.Tensor, boxes: torch.Tensor, image_shape: Tuple[torch.Tensor, torch.Tensor],
                              threshold: float = 0.5):
    """paste_masks_in_image with a tensor-valued image shape: during tracing the Tensor -> int conversion must be
    scripted, not traced (mask_ops.py:264-275)."""
    return paste_masks_in_image(masks, boxes, (int(image_shape[0]), int(image_shape[1])), threshold)
