// The arithmetic of pairwise_iou + Matcher shared by matcher.hip (fused matching) and label_sample.hip (proposal
// labelling + sampling).  Units that include this are compiled with -ffp-contract=off: the IoU expression must stay
// operation for operation structures/boxes.py:312-358.
#pragma once
#include "common.h"

namespace d2amd {

__device__ __forceinline__ float mt_tmin(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a < b ? a : b); }
__device__ __forceinline__ float mt_tmax(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a > b ? a : b); }

// structures/boxes.py:312-358, operation for operation (see iou.hip iou_one<D2AMD_IOU>)
__device__ __forceinline__ float mt_iou(float4 a, float area1, float4 b) {
  float w = mt_tmin(a.z, b.z) - mt_tmax(a.x, b.x);
  float h = mt_tmin(a.w, b.w) - mt_tmax(a.y, b.y);
  if (w < 0) w = 0;
  if (h < 0) h = 0;
  const float inter = w * h;
  const float area2 = (b.z - b.x) * (b.w - b.y);
  if (inter > 0) return inter / (area1 + area2 - inter);
  return 0.f;
}

// the same expression when no operand is NaN (torch.min / max then equal fminf / fmaxf): 4 instructions instead of 20.
// Callers check the boxes once per thread / ground-truth chunk and take mt_iou for anything containing a NaN.
__device__ __forceinline__ float mt_iou_fast(float4 a, float area1, float4 b, float area2) {
  float w = fminf(a.z, b.z) - fmaxf(a.x, b.x);
  float h = fminf(a.w, b.w) - fmaxf(a.y, b.y);
  if (w < 0) w = 0;
  if (h < 0) h = 0;
  const float inter = w * h;
  if (inter > 0) return inter / (area1 + area2 - inter);
  return 0.f;
}
__device__ __forceinline__ bool mt_has_nan(float4 v) { return v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w; }

struct MatchCfg {
  float thr[D2AMD_MATCHER_MAX_THRESHOLDS];
  int8_t lab[D2AMD_MATCHER_MAX_THRESHOLDS + 1];
  int T;
};

// matcher.py:96-101: labels start at 1; each interval [low, high) with low = -inf / thr[k-1],
// high = thr[k] / +inf overwrites (NaN matches no interval and keeps 1)
__device__ __forceinline__ int8_t mt_label(float v, const MatchCfg& c) {
  int8_t l = 1;
  for (int k = 0; k <= c.T; k++) {
    const bool ge_low = k == 0 ? (v >= -__builtin_inff()) : (v >= c.thr[k - 1]);
    const bool lt_high = k == c.T ? (v < __builtin_inff()) : (v < c.thr[k]);
    if (ge_low && lt_high) l = c.lab[k];
  }
  return l;
}

}  // namespace d2amd
