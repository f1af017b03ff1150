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

// Optional epilogue of the rank stage (segments of <= 2,048 selected pairs: the RPN's 2,000 per level): the thread that
// places pair r of segment (image, level) also DECODES its anchor -- Box2BoxTransform.apply_deltas + clip + validity of
// rpn.hip's rpn_decode_kernel -- so the selection ends one dependent launch earlier (the first three launches of a
// training step's critical path become two).  The arithmetic is rpn_decode_row() below in both places.
struct TopkRpnEpilogue {
  const float4* deltas[TOPK_MAX_LEVELS];   // [N][A_l] (element stride per image = TopkInput::stride)
  const float4* anchors[TOPK_MAX_LEVELS];  // [A_l]
  int img_h[16], img_w[16];                // N <= 16
  float wx, wy, ww, wh, scale_clamp, min_size;
  float4* boxes;       // [N][Ktot]
  float* scores;       // [N][Ktot]
  uint8_t* valid;      // [N][Ktot]
  int64_t* level_ids;  // [Ktot]
  int* flags;          // bit 0: a non-finite box or score was decoded
};

__device__ __forceinline__ bool rpn_finite(float v) { return fabsf(v) <= 3.402823466e+38f; }

// one proposal: anchor b + deltas d + objectness `score` of image (W x H) -> row t of the outputs, row j of its image
// (box_regression.py:88-116 in fp32; proposal_utils.py:98-112: finite filter, clip, nonempty)
__device__ __forceinline__ void rpn_decode_row(float4 b, float4 d, float score, float W, float H, float wx, float wy,
                                               float ww, float wh, float scale_clamp, float min_size, long t, int j,
                                               int level, bool first_image, float4* __restrict__ boxes,
                                               float* __restrict__ scores, uint8_t* __restrict__ valid,
                                               int64_t* __restrict__ level_ids, int* __restrict__ flags) {
  const float widths = b.z - b.x, heights = b.w - b.y;
  const float ctr_x = b.x + 0.5f * widths, ctr_y = b.y + 0.5f * heights;
  const float dx = d.x / wx, dy = d.y / wy;
  float dw = d.z / ww, dh = d.w / wh;
  dw = dw != dw ? dw : fminf(dw, scale_clamp);  // torch.clamp(max=) propagates NaN
  dh = dh != dh ? dh : fminf(dh, scale_clamp);
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
  const float pw = expf(dw) * widths, ph = expf(dh) * heights;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
  const bool fin = rpn_finite(x1) && rpn_finite(y1) && rpn_finite(x2) && rpn_finite(y2) && rpn_finite(score);
  if (!fin) atomicOr(flags, 1);
  x1 = fminf(fmaxf(x1, 0.f), W); y1 = fminf(fmaxf(y1, 0.f), H);
  x2 = fminf(fmaxf(x2, 0.f), W); y2 = fminf(fmaxf(y2, 0.f), H);
  const bool ok = fin && (x2 - x1 > min_size) && (y2 - y1 > min_size);
  boxes[t] = ok ? make_float4(x1, y1, x2, y2) : make_float4(0.f, 0.f, 0.f, 0.f);
  scores[t] = ok ? score : -__builtin_inff();
  valid[t] = ok ? 1 : 0;
  if (first_image) level_ids[j] = level;
}

size_t topk_workspace_bytes(const TopkInput& in);

// For every segment (image, level): the k[l] best candidates by stored value, best first, ties towards the lower
// element index.
//   use_thr:  only elements with value >= xmin are candidates (a segment may then select fewer than k[l]); a NaN
//             xmin admits none.  Callers that threshold a monotone function of the value (sigmoid) pass the
//             equivalent bound on the value itself: logit_lower_bound().
// Outputs: sel [N][Ktot] element index inside its level (rows [koff[l], koff[l] + cnt) of a segment are valid),
//          cnt [N][L] selected count per segment.  Nothing synchronises with the host.
// `clear_word`: one caller-owned int cleared by the launch that clears the workspace (saves the caller a launch).
// `rpn` (optional): decode the selected anchors in the rank stage; *rpn_done tells whether that happened (it does when
// every segment is ranked rather than sorted: k <= 2,048 per segment, N <= 16) -- if not, the caller decodes itself.
int topk_select(const TopkInput& in, bool use_thr, float xmin, uint32_t* sel, int* cnt, void* ws, size_t ws_bytes,
                hipStream_t s, int* clear_word = nullptr, const TopkRpnEpilogue* rpn = nullptr, bool* rpn_done = nullptr);

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
