from .boxes import Boxes, pairwise_intersection, pairwise_ioa, pairwise_iou
from .masks import BitMasks, PolygonMasks, crop_and_resize_batch

__all__ = ["Boxes", "BitMasks", "PolygonMasks", "crop_and_resize_batch", "pairwise_iou", "pairwise_ioa", "pairwise_intersection"]
