"""Hot-path callers mirrored from `detectron2.modeling` (only what SURVEY.md section 8 lists)."""
from .matcher import Matcher
from .proposal_utils import Proposals, find_top_rpn_proposals_fused, rpn_select_proposals
from .poolers import ROIPooler, assign_boxes_to_levels, convert_boxes_to_pooler_format

__all__ = ["Matcher", "Proposals", "find_top_rpn_proposals_fused", "rpn_select_proposals", "ROIPooler", "assign_boxes_to_levels", "convert_boxes_to_pooler_format"]
