from .boxes import Boxes, pairwise_intersection, pairwise_ioa, pairwise_iou
from .masks import BitMasks, crop_and_resize_batch

__all__ = ["Boxes", "BitMasks", "crop_and_resize_batch", "pairwise_iou", "pairwise_ioa", "pairwise_intersection"]
