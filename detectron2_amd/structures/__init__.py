from .boxes import Boxes, pairwise_intersection, pairwise_ioa, pairwise_iou
from .masks import BitMasks

__all__ = ["Boxes", "BitMasks", "pairwise_iou", "pairwise_ioa", "pairwise_intersection"]
