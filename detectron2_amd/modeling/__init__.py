"""Hot-path callers mirrored from `detectron2.modeling` (only what SURVEY.md section 8 lists)."""
from .matcher import Matcher
from .poolers import ROIPooler, assign_boxes_to_levels, convert_boxes_to_pooler_format

__all__ = ["Matcher", "ROIPooler", "assign_boxes_to_levels", "convert_boxes_to_pooler_format"]
