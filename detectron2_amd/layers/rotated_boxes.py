"""pairwise_iou_rotated -- mirrors detectron2/layers/rotated_boxes.py:6-21."""
import torch

from . import ops  # noqa: F401


def pairwise_iou_rotated(boxes1, boxes2):
    """IoU of Tensor[N,5] x Tensor[M,5] rotated boxes (x_center, y_center, width, height, angle)
    -> Tensor[N,M] float32."""
    return torch.ops.detectron2.box_iou_rotated(boxes1, boxes2)
