// Fused anchor / proposal matching: pairwise_iou + Matcher without the M x N matrix.
//   replaces  detectron2/structures/boxes.py:312-358 (pairwise_iou) followed by
//             detectron2/modeling/matcher.py:62-127 (Matcher.__call__, set_low_quality_matches_)
//   as called from proposal_generator/rpn.py:307-364 (G x 268,569 anchors per image) and
//   roi_heads/roi_heads.py:257-295 (G x ~1,000 proposals).
// The reference materialises the matrix (17 MB per image for the RPN), then makes ~8 elementwise /
// reduction passes over it (max over dim 0, one mask per label interval, max over dim 1, equality,
// nonzero, index_put).  Here: pass 1 keeps the running column maximum in registers and the row maxima
// in LDS (wave max -> one LDS atomic per wave and ground-truth box -> one global atomic per block);
// pass 2 (allow_low_quality_matches only) re-evaluates the same fp32 IoU expression and compares with
// the row maximum.  Roofline: HBM, 16 N bytes read per pass + 9 N bytes written.
// Bit-exact: same IoU arithmetic as iou.hip (this unit is compiled with -ffp-contract=off), torch.max
// tie rule (first maximal index), NaN propagation of torch.max.
// The same kernels also serve Matcher.__call__(matrix) for callers that already hold a matrix.
#pragma clang fp contract(off)
#include "common.h"
#include "matcher_core.h"

namespace d2amd {

// order-preserving float -> uint key for non-negative values; NaN maps above every number so that a
// row containing NaN reports NaN as its maximum, like torch.max
__device__ __forceinline__ uint32_t mt_key(float v) { return v != v ? 0xffffffffu : (v <= 0.f ? 0u : __float_as_uint(v)); }

constexpr int MT_BLOCK = 256;
constexpr int MT_CHUNK = 512;  // ground-truth boxes staged per LDS pass
constexpr int MT_U = 8;        // ground-truth boxes evaluated together per prediction
constexpr int MT_MAX_IMAGES = 16;  // images per launch of the batched entry (blockIdx.y)

// The ground truth of a batch of images matched against the SAME predictions (the RPN's anchors: rpn.py:331-353 loops
// over the images in Python, one pairwise_iou + Matcher each): image blockIdx.y takes gt[y] / M[y] and writes row y.
struct MtBatch {
  const float4* gt[MT_MAX_IMAGES];
  int M[MT_MAX_IMAGES];
  long out_stride;  // elements between the images' rows of matches / labels
  int rm_stride;    // words between the images' row maxima
};
// (constant indices only into the kernel-argument struct: a dynamic one makes the compiler copy it to scratch)
__device__ __forceinline__ void mt_image(const MtBatch& B, int img, const float4*& gt, int& M) {
  gt = B.gt[0]; M = B.M[0];
#pragma unroll
  for (int q = 1; q < MT_MAX_IMAGES; q++)
    if (q == img) { gt = B.gt[q]; M = B.M[q]; }
}

// FUSED: IoU from boxes; else values from the row-major M x N matrix `q`
template <bool FUSED>
__global__ __launch_bounds__(MT_BLOCK) void match_pass1_kernel(const MtBatch B,
                                                              const float4* __restrict__ boxes, int N,
                                                              const float* __restrict__ q, MatchCfg cfg,
                                                              int64_t* __restrict__ matches, int8_t* __restrict__ labels,
                                                              uint32_t* __restrict__ rowmax) {
  const float4* gt;
  int M;
  mt_image(B, blockIdx.y, gt, M);
  matches += (long)blockIdx.y * B.out_stride;
  labels += (long)blockIdx.y * B.out_stride;
  if (rowmax) rowmax += (long)blockIdx.y * B.rm_stride;
  __shared__ float4 g4[MT_CHUNK];
  __shared__ float garea[MT_CHUNK];
  __shared__ uint32_t rmax[MT_CHUNK];
  __shared__ int s_gt_nan;
  const int tid = threadIdx.x, lane = tid & 63;
  const long n = (long)blockIdx.x * MT_BLOCK + tid;
  const bool valid = n < N;
  float4 b = make_float4(0, 0, 0, 0);
  if (FUSED && valid) b = boxes[n];
  const float barea = (b.z - b.x) * (b.w - b.y);
  const bool bnan = mt_has_nan(b);
  float best = 0.f;
  int besti = 0;
  bool have = false;
  for (int m0 = 0; m0 < M; m0 += MT_CHUNK) {
    const int mc = min(MT_CHUNK, M - m0);
    __syncthreads();
    if (tid == 0) s_gt_nan = 0;
    __syncthreads();
    for (int i = tid; i < mc; i += MT_BLOCK) {
      if (FUSED) {
        const float4 a = gt[m0 + i];
        g4[i] = a;
        garea[i] = (a.z - a.x) * (a.w - a.y);
        if (mt_has_nan(a)) s_gt_nan = 1;
      }
      rmax[i] = 0u;
    }
    __syncthreads();
    const bool slow = bnan || s_gt_nan != 0;  // a NaN coordinate somewhere: torch.min / max propagate it
    // MT_U ground-truth boxes per step: their IoUs, LDS reads and wave reductions are independent chains the
    // hardware overlaps (one box per step was a ~500-cycle dependent chain: 7 us even for 1,000 predictions)
    for (int i0 = 0; i0 < mc; i0 += MT_U) {
      float v[MT_U];
#pragma unroll
      for (int u = 0; u < MT_U; u++) {
        const int i = min(i0 + u, mc - 1);
        v[u] = 0.f;
        if (valid) v[u] = FUSED ? (slow ? mt_iou(g4[i], garea[i], b) : mt_iou_fast(g4[i], garea[i], b, barea))
                                : q[(long)(m0 + i) * N + n];
      }
#pragma unroll
      for (int u = 0; u < MT_U; u++) {
        if (valid && i0 + u < mc) {
          // torch.max(dim=0): first maximal value; NaN is maximal
          const bool better = !have || (v[u] > best) || (v[u] != v[u] && best == best);
          if (better) { best = v[u]; besti = m0 + i0 + u; have = true; }
        }
      }
      if (rowmax) {  // uniform.  row maxima: wave max, then one LDS atomic per wave and box
        // (tried: every lane with a non-zero quality issuing its own LDS atomicMax -- slower, 397 vs 229 us at
        // M = 256, scripts/matcher_scan.py: conflicting ds_max serialise)
        uint32_t k[MT_U];
        unsigned long long any = 0ull;
#pragma unroll
        for (int u = 0; u < MT_U; u++) {
          k[u] = (valid && i0 + u < mc) ? mt_key(v[u]) : 0u;
          any |= __ballot(k[u] != 0u);
        }
        // most (ground truth, 64 neighbouring predictions) pairs do not overlap at all: skip the reductions then
        if (any != 0ull) {
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
            for (int u = 0; u < MT_U; u++) k[u] = max(k[u], (uint32_t)__shfl_xor((int)k[u], o));
          }
          if (lane < MT_U && i0 + lane < mc) {
            uint32_t mine = 0u;
#pragma unroll
            for (int u = 0; u < MT_U; u++) mine = lane == u ? k[u] : mine;
            if (mine != 0u) atomicMax(&rmax[i0 + lane], mine);
          }
        }
      }
    }
    __syncthreads();
    if (rowmax)
      for (int i = tid; i < mc; i += MT_BLOCK)
        if (rmax[i] != 0u) atomicMax(&rowmax[m0 + i], rmax[i]);
  }
  if (valid) {
    matches[n] = besti;
    labels[n] = mt_label(best, cfg);
  }
}

// set_low_quality_matches_ (matcher.py:103-127): label 1 for every prediction whose quality with some
// ground truth equals that ground truth's row maximum (ties included; a row maximum of 0 matches every
// prediction with quality 0, exactly like the reference's `==` + nonzero)
template <bool FUSED>
__global__ __launch_bounds__(MT_BLOCK) void match_pass2_kernel(const MtBatch B,
                                                              const float4* __restrict__ boxes, int N,
                                                              const float* __restrict__ q,
                                                              const uint32_t* __restrict__ rowmax,
                                                              int8_t* __restrict__ labels) {
  const float4* gt;
  int M;
  mt_image(B, blockIdx.y, gt, M);
  labels += (long)blockIdx.y * B.out_stride;
  rowmax += (long)blockIdx.y * B.rm_stride;
  __shared__ float4 g4[MT_CHUNK];
  __shared__ float garea[MT_CHUNK];
  __shared__ uint32_t rmax[MT_CHUNK];
  __shared__ int s_gt_nan;
  const int tid = threadIdx.x;
  const long n = (long)blockIdx.x * MT_BLOCK + tid;
  const bool valid = n < N;
  float4 b = make_float4(0, 0, 0, 0);
  if (FUSED && valid) b = boxes[n];
  const float barea = (b.z - b.x) * (b.w - b.y);
  const bool bnan = mt_has_nan(b);
  bool hit = false;
  for (int m0 = 0; m0 < M; m0 += MT_CHUNK) {
    const int mc = min(MT_CHUNK, M - m0);
    __syncthreads();
    if (tid == 0) s_gt_nan = 0;
    __syncthreads();
    for (int i = tid; i < mc; i += MT_BLOCK) {
      if (FUSED) {
        const float4 a = gt[m0 + i];
        g4[i] = a;
        garea[i] = (a.z - a.x) * (a.w - a.y);
        if (mt_has_nan(a)) s_gt_nan = 1;
      }
      rmax[i] = rowmax[m0 + i];
    }
    __syncthreads();
    const bool slow = bnan || s_gt_nan != 0;
    if (valid) {
      for (int i0 = 0; i0 < mc; i0 += MT_U) {
#pragma unroll
        for (int u = 0; u < MT_U; u++) {
          const int i = min(i0 + u, mc - 1);  // (a repeated last box only repeats its own test)
          const float v = FUSED ? (slow ? mt_iou(g4[i], garea[i], b) : mt_iou_fast(g4[i], garea[i], b, barea))
                                : q[(long)(m0 + i) * N + n];
          // float equality like the reference (NaN never equal); the key of a non-negative value is its bits
          if (v == v && mt_key(v) == rmax[i] && rmax[i] != 0xffffffffu) hit = true;
        }
      }
    }
  }
  if (valid && hit) labels[n] = 1;
}

__global__ void match_fill_kernel(int64_t* matches, int8_t* labels, int N, int8_t lab0) {
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) { matches[n] = 0; labels[n] = lab0; }
}

static int match_cfg(const float* thresholds, const int8_t* labels, int T, MatchCfg& cfg) {
  D2_CHECK_ARG(T >= 0 && T <= D2AMD_MATCHER_MAX_THRESHOLDS && labels != nullptr && (T == 0 || thresholds != nullptr),
               "match: %d thresholds (max %d)", T, D2AMD_MATCHER_MAX_THRESHOLDS);
  cfg = MatchCfg{};
  cfg.T = T;
  for (int k = 0; k < T; k++) {
    D2_CHECK_ARG(thresholds[k] > 0.f && (k == 0 || thresholds[k - 1] <= thresholds[k]),
                 "match: thresholds must be positive and ascending");
    cfg.thr[k] = thresholds[k];
  }
  for (int k = 0; k <= T; k++) {
    D2_CHECK_ARG(labels[k] >= -1 && labels[k] <= 1, "match: labels must be in {-1, 0, 1}");
    cfg.lab[k] = labels[k];
  }
  return D2AMD_OK;
}

// `count` images (<= MT_MAX_IMAGES per launch) against the same N predictions: gt[i] [M[i], 4]; outputs [count][N].
// q (matrix mode): one image.  A ground-truth-free image takes the same kernels (no box: index 0, labels[0]:
// matcher.py:80-90).
static int match_impl(const float* const* gt, const int* M, int count, const float* boxes, const float* q, int N,
                      const float* thresholds, const int8_t* labels, int T, int allow_low, int64_t* matches,
                      int8_t* match_labels, void* workspace, size_t workspace_bytes, hipStream_t s) {
  D2_CHECK_ARG(count >= 0 && N >= 0, "match: negative size");
  MatchCfg cfg;
  { const int rc = match_cfg(thresholds, labels, T, cfg); if (rc) return rc; }
  if (N == 0 || count == 0) return D2AMD_OK;
  D2_CHECK_ARG(matches && match_labels, "match: null output");
  int mmax = 0;
  for (int i = 0; i < count; i++) {
    D2_CHECK_ARG(M[i] >= 0, "match: negative size");
    D2_CHECK_ARG(M[i] == 0 || q != nullptr || (gt[i] != nullptr && boxes != nullptr), "match: null input");
    mmax = M[i] > mmax ? M[i] : mmax;
  }
  const int grid = cdiv(N, MT_BLOCK);
  if (mmax == 0 && count == 1) {  // no ground truth -> index 0, labels[0]
    hipLaunchKernelGGL(match_fill_kernel, dim3(grid), dim3(MT_BLOCK), 0, s, matches, match_labels, N, cfg.lab[0]);
    D2_LAUNCH_OK();
    return D2AMD_OK;
  }
  const int rm_stride = (mmax + 63) / 64 * 64;
  uint32_t* rowmax = nullptr;
  if (allow_low && mmax > 0) {
    const size_t need = (size_t)count * rm_stride * 4;
    if (workspace == nullptr || workspace_bytes < need) {
      set_error("match: workspace too small (%zu < %zu)", workspace_bytes, need);
      return D2AMD_EWORKSPACE;
    }
    rowmax = (uint32_t*)workspace;
    { const int zrc = zero_async(rowmax, need, s); if (zrc) return zrc; }
  }
  for (int i0 = 0; i0 < count; i0 += MT_MAX_IMAGES) {
    const int c = count - i0 < MT_MAX_IMAGES ? count - i0 : MT_MAX_IMAGES;
    MtBatch B{};
    for (int i = 0; i < c; i++) { B.gt[i] = (const float4*)gt[i0 + i]; B.M[i] = M[i0 + i]; }
    B.out_stride = N;
    B.rm_stride = rm_stride;
    int64_t* mo = matches + (long)i0 * N;
    int8_t* lo = match_labels + (long)i0 * N;
    uint32_t* rm = rowmax ? rowmax + (long)i0 * rm_stride : nullptr;
    if (q)
      hipLaunchKernelGGL((match_pass1_kernel<false>), dim3(grid, c), dim3(MT_BLOCK), 0, s, B, nullptr, N, q, cfg, mo, lo, rm);
    else
      hipLaunchKernelGGL((match_pass1_kernel<true>), dim3(grid, c), dim3(MT_BLOCK), 0, s, B, (const float4*)boxes, N,
                         nullptr, cfg, mo, lo, rm);
    D2_LAUNCH_OK();
    if (rm) {
      if (q)
        hipLaunchKernelGGL((match_pass2_kernel<false>), dim3(grid, c), dim3(MT_BLOCK), 0, s, B, nullptr, N, q, rm, lo);
      else
        hipLaunchKernelGGL((match_pass2_kernel<true>), dim3(grid, c), dim3(MT_BLOCK), 0, s, B, (const float4*)boxes, N,
                           nullptr, rm, lo);
      D2_LAUNCH_OK();
    }
  }
  return D2AMD_OK;
}

}  // namespace d2amd

using namespace d2amd;

extern "C" size_t d2amd_matcher_workspace_bytes(int M) { return (size_t)((M > 0 ? M : 1) + 63) / 64 * 64 * 4; }

extern "C" int d2amd_match_boxes(const float* gt_boxes, int M, const float* boxes, int N, const float* thresholds,
                                 const int8_t* labels, int T, int allow_low_quality, int64_t* matches,
                                 int8_t* match_labels, void* workspace, size_t workspace_bytes, void* stream) {
  D2_CHECK_ARG(M >= 0, "match: negative size");
  return match_impl(&gt_boxes, &M, 1, boxes, nullptr, N, thresholds, labels, T, allow_low_quality, matches,
                    match_labels, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" size_t d2amd_match_boxes_batch_workspace_bytes(const int* M, int count) {
  int mmax = 1;
  for (int i = 0; i < count; i++) mmax = M[i] > mmax ? M[i] : mmax;
  return (size_t)(count > 0 ? count : 1) * ((mmax + 63) / 64 * 64) * 4;
}

extern "C" int d2amd_match_boxes_batch(const float* const* gt_boxes, const int* M, int count, const float* boxes, int N,
                                       const float* thresholds, const int8_t* labels, int T, int allow_low_quality,
                                       int64_t* matches, int8_t* match_labels, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  D2_CHECK_ARG(count == 0 || (gt_boxes != nullptr && M != nullptr), "match_boxes_batch: null image list");
  return match_impl(gt_boxes, M, count, boxes, nullptr, N, thresholds, labels, T, allow_low_quality, matches,
                    match_labels, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int d2amd_match_quality_matrix(const float* quality, int M, int N, const float* thresholds,
                                          const int8_t* labels, int T, int allow_low_quality, int64_t* matches,
                                          int8_t* match_labels, void* workspace, size_t workspace_bytes, void* stream) {
  D2_CHECK_ARG(quality != nullptr || (long)M * N == 0, "match_quality_matrix: null matrix");
  D2_CHECK_ARG(M >= 0, "match: negative size");
  const float* none = nullptr;
  return match_impl(&none, &M, 1, nullptr, quality, N, thresholds, labels, T, allow_low_quality, matches, match_labels,
                    workspace, workspace_bytes, (hipStream_t)stream);
}
