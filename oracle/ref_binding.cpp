// Binding shim for the COMPILED-REFERENCE oracle (oracle/_ref/libd2ref.so).
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes the
// reference's own headers where they lie under /root/reference and re-exports the
// reference's CPU entry points under the torch op namespace `d2ref` (the reference's
// own vision.cpp registers `detectron2::*`, which would collide with the product's ops).
//   detectron2/layers/csrc/ROIAlignRotated/ROIAlignRotated.h:10-47   (*_cpu declarations)
//   detectron2/layers/csrc/box_iou_rotated/box_iou_rotated.h:8-18
//   detectron2/layers/csrc/nms_rotated/nms_rotated.h:8-20
#include <torch/library.h>
#include "ROIAlignRotated/ROIAlignRotated.h"
#include "box_iou_rotated/box_iou_rotated.h"
#include "nms_rotated/nms_rotated.h"

namespace {
// vision.cpp:115-120 registers the header dispatchers, whose scalar arguments are
// double / int64_t (ROIAlignRotated.h:50-113); the *_cpu entry points take float / int.
at::Tensor rar_fwd(const at::Tensor& input, const at::Tensor& rois, double spatial_scale,
                   int64_t pooled_height, int64_t pooled_width, int64_t sampling_ratio) {
  return detectron2::ROIAlignRotated_forward_cpu(
      input, rois, (float)spatial_scale, (int)pooled_height, (int)pooled_width, (int)sampling_ratio);
}
at::Tensor rar_bwd(const at::Tensor& grad, const at::Tensor& rois, double spatial_scale,
                   int64_t pooled_height, int64_t pooled_width, int64_t batch_size,
                   int64_t channels, int64_t height, int64_t width, int64_t sampling_ratio) {
  return detectron2::ROIAlignRotated_backward_cpu(
      grad, rois, (float)spatial_scale, (int)pooled_height, (int)pooled_width, (int)batch_size,
      (int)channels, (int)height, (int)width, (int)sampling_ratio);
}
at::Tensor nms_rot(const at::Tensor& dets, const at::Tensor& scores, double iou_threshold) {
  return detectron2::nms_rotated_cpu(dets.contiguous(), scores.contiguous(), iou_threshold);
}
at::Tensor iou_rot(const at::Tensor& boxes1, const at::Tensor& boxes2) {
  return detectron2::box_iou_rotated_cpu(boxes1.contiguous(), boxes2.contiguous());
}
} // namespace

TORCH_LIBRARY(d2ref, m) {
  m.def("nms_rotated", &nms_rot);
  m.def("box_iou_rotated", &iou_rot);
  m.def("roi_align_rotated_forward", &rar_fwd);
  m.def("roi_align_rotated_backward", &rar_bwd);
}
