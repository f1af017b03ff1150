from .boxes import Boxes, pairwise_intersection, pairwise_ioa, pairwise_iou

__all__ = ["Boxes", "pairwise_iou", "pairwise_ioa", "pairwise_intersection"]
