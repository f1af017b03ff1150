"""The hot-path subset of `detectron2.layers` (reference: detectron2/layers/__init__.py:2-7)."""
from .deform_conv import DeformConv, ModulatedDeformConv, deform_conv, modulated_deform_conv
from .mask_ops import paste_masks_in_image
from .nms import batched_nms, batched_nms_images, batched_nms_rotated, nms, nms_rotated
from .roi_align import ROIAlign, roi_align
from .roi_align_rotated import ROIAlignRotated, roi_align_rotated
from .rotated_boxes import pairwise_iou_rotated

__all__ = [
    "ROIAlign", "roi_align", "ROIAlignRotated", "roi_align_rotated", "DeformConv", "ModulatedDeformConv",
    "deform_conv", "modulated_deform_conv", "nms", "batched_nms", "batched_nms_images", "nms_rotated",
    "batched_nms_rotated",
    "paste_masks_in_image", "pairwise_iou_rotated",
]
