// Segmented top-k selection by radix select (internal; used by rpn.hip).  See topk.hip.
#pragma once
#include "common.h"

namespace d2amd {

constexpr int TOPK_MAX_LEVELS = 8;
constexpr int TOPK_MAX_K = 65536;  // per segment: ordered in LDS runs of 16,384 (128 KB) merged by rank (topk.hip)

// Scores of N images x L levels.  Element i of (image, level l) lives at ptr[l][image * stride[l] + i].
struct TopkInput {
  const float* ptr[TOPK_MAX_LEVELS];
  long stride[TOPK_MAX_LEVELS];
  int size[TOPK_MAX_LEVELS];      // elements per image on level l
  int k[TOPK_MAX_LEVELS];         // selection size of a (image, level) segment: min(size, topk)
  int koff[TOPK_MAX_LEVELS + 1];  // prefix of k
  int L, N;
};

size_t topk_workspace_bytes(const TopkInput& in);

// For every segment (image, level): the k[l] best candidates by stored value, best first, ties towards the lower
// element index.
//   use_thr:  only elements with value >= xmin are candidates (a segment may then select fewer than k[l]); a NaN
//             xmin admits none.  Callers that threshold a monotone function of the value (sigmoid) pass the
//             equivalent bound on the value itself: logit_lower_bound().
// Outputs: sel [N][Ktot] element index inside its level (rows [koff[l], koff[l] + cnt) of a segment are valid),
//          cnt [N][L] selected count per segment.  Nothing synchronises with the host.
// `clear_word`: one caller-owned int cleared by the launch that clears the workspace (saves the caller a launch).
int topk_select(const TopkInput& in, bool use_thr, float xmin, uint32_t* sel, int* cnt, void* ws, size_t ws_bytes,
                hipStream_t s, int* clear_word = nullptr);

// Smallest fp32 logit x whose sigmoid exceeds the fp32 threshold `thr` in exact arithmetic:
// sigmoid(x) > thr  <=>  x > log(thr / (1 - thr)), evaluated once on the host in double.  thr >= 1: none (NaN);
// thr < 0: all (-inf); thr == 0: every x whose fp32 sigmoid is not flushed to zero (exp(-x) finite in fp32).
float logit_lower_bound(float thr);

__device__ __forceinline__ uint32_t topk_desc_key(float s) {  // ascending key order = descending score; NaN first
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~u;
}

}  // namespace d2amd
