"""Hot-path callers mirrored from `detectron2.modeling` (only what SURVEY.md section 8 lists)."""
from .dense_detector import Detections, dense_detector_inference_fused, dense_select_predictions
from .fast_rcnn import DeviceDetections, fast_rcnn_inference_device, fast_rcnn_inference_fused, fast_rcnn_predict
from .mask_head import mask_rcnn_inference, mask_rcnn_loss, mask_rcnn_loss_from_targets
from .matcher import Matcher
from .proposal_utils import DeviceProposals, Proposals, find_top_rpn_proposals_fused, rpn_select_proposals
from .poolers import ROIPooler, assign_boxes_to_levels, convert_boxes_to_pooler_format, pool_pair, pool_pair_rois, PairBackwardPlan
from .sampling import (DeviceKeyGenerator, label_and_sample_proposals_fixed, subsample_anchor_labels_, subsample_labels,
                       subsample_labels_batch)

__all__ = ["Detections", "DeviceDetections", "fast_rcnn_inference_device", "fast_rcnn_inference_fused", "fast_rcnn_predict", "dense_detector_inference_fused", "dense_select_predictions", "mask_rcnn_loss", "mask_rcnn_inference", "mask_rcnn_loss_from_targets", "Matcher", "Proposals", "DeviceProposals", "find_top_rpn_proposals_fused", "rpn_select_proposals", "ROIPooler", "pool_pair", "pool_pair_rois", "PairBackwardPlan", "assign_boxes_to_levels", "convert_boxes_to_pooler_format", "subsample_labels", "subsample_labels_batch", "subsample_anchor_labels_", "DeviceKeyGenerator",
           "label_and_sample_proposals_fixed"]
