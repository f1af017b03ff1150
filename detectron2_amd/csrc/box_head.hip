// Box-head inference in front of the per-class NMS (the caller between the box pooler and batched_nms on the Mask R-CNN
// inference path):
//   replaces  detectron2/modeling/roi_heads/fast_rcnn.py:134-158 (fast_rcnn_inference_single_image up to the NMS: the
//             finite-row mask [two reductions + boolean index], Boxes.clip over all R*K boxes, `scores > thresh`,
//             `nonzero` [a host sync per image], two boolean-mask gathers) for ALL images of the batch, without a sync.
// Three small launches: per-row candidate counts (a wave per row), an exclusive scan of them per image, the ordered
// write -- the candidates come out in torch.nonzero's row-major (row, class) order, which is what makes batched_nms'
// result (keep order among equal scores) the reference's.  Nothing here is arithmetic: compares, a clamp, copies.
// Roofline: HBM; bytes = R * (K + 1 + 4 K_box) * 4 read twice (count, write) + 36 B per candidate.
#include "common.h"

namespace d2amd {

constexpr int BH_WAVES = 16;  // rows per workgroup (a wave per row)

struct BhImages {
  const float* boxes[D2AMD_POOLER_MAX_IMAGES];   // [R_i][Kb * 4]
  const float* scores[D2AMD_POOLER_MAX_IMAGES];  // [R_i][K + 1]
  int rows[D2AMD_POOLER_MAX_IMAGES];
  int row_base[D2AMD_POOLER_MAX_IMAGES + 1];     // prefix of rows: an image's slice of the per-row arrays
  float h[D2AMD_POOLER_MAX_IMAGES], w[D2AMD_POOLER_MAX_IMAGES];
  long cap_base[D2AMD_POOLER_MAX_IMAGES + 1];    // prefix of rows * K: an image's slice of the candidate arrays
  int n, K, Kb;
  float thr;
};

__device__ __forceinline__ bool bh_finite(float v) { return fabsf(v) <= 3.402823466e+38f; }  // (false for NaN / inf)

// the image of this workgroup's rows (constant indices only into the by-value struct)
__device__ __forceinline__ void bh_image(const BhImages& I, int img, const float*& boxes, const float*& scores, int& rows,
                                         int& row_base, long& cap_base, float& h, float& w) {
  boxes = I.boxes[0]; scores = I.scores[0]; rows = I.rows[0]; row_base = I.row_base[0]; cap_base = I.cap_base[0];
  h = I.h[0]; w = I.w[0];
#pragma unroll
  for (int q = 1; q < D2AMD_POOLER_MAX_IMAGES; q++)
    if (q == img) {
      boxes = I.boxes[q]; scores = I.scores[q]; rows = I.rows[q]; row_base = I.row_base[q]; cap_base = I.cap_base[q];
      h = I.h[q]; w = I.w[q];
    }
}

// fast_rcnn.py:134-137 + :148: rowcnt[row] = #classes with score > thr of a row whose boxes and scores are all finite
__global__ __launch_bounds__(64 * BH_WAVES) void bh_count_kernel(const BhImages I, int* __restrict__ rowcnt) {
  const int img = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *boxes, *scores;
  int rows, row_base; long cap_base; float h, w;
  bh_image(I, img, boxes, scores, rows, row_base, cap_base, h, w);
  const int r = blockIdx.x * BH_WAVES + wave;
  if (r >= rows) return;  // uniform per wave
  const float* s = scores + (long)r * (I.K + 1);
  const float* b = boxes + (long)r * (I.Kb * 4);
  bool fin = true;
  int cnt = 0;
  for (int k = lane; k <= I.K; k += 64) {
    const float v = s[k];
    fin = fin && bh_finite(v);
    cnt += (k < I.K && v > I.thr) ? 1 : 0;
  }
  for (int k = lane; k < I.Kb * 4; k += 64) fin = fin && bh_finite(b[k]);
  const bool all_fin = __ballot(!fin) == 0ull;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
  if (lane == 0) rowcnt[row_base + r] = all_fin ? cnt : -1;  // (-1: a dropped row)
}

// exclusive scans of an image's row counts and of its kept-row flags (one 1,024-thread workgroup per image; the two
// travel as one 64-bit sum); counts[img] = the candidates.  rowidx[row] = the row's index among the rows that are not
// dropped: what the reference reports as `filter_inds[:, 0]` (it indexes boxes[valid_mask], fast_rcnn.py:135-137)
__global__ __launch_bounds__(1024) void bh_scan_kernel(const BhImages I, const int* __restrict__ rowcnt,
                                                       int* __restrict__ rowoff, int* __restrict__ rowidx,
                                                       int64_t* __restrict__ counts) {
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *boxes, *scores;
  int rows, row_base; long cap_base; float h, w;
  bh_image(I, img, boxes, scores, rows, row_base, cap_base, h, w);
  __shared__ long long wsum[16];
  __shared__ long long s_run;
  if (tid == 0) s_run = 0;
  __syncthreads();
  for (int r0 = 0; r0 < rows; r0 += 1024) {  // uniform
    const int r = r0 + tid;
    const int c = r < rows ? rowcnt[row_base + r] : -1;
    const long long v = c >= 0 ? ((long long)c | (1ll << 32)) : 0ll;  // candidates | kept row << 32
    long long x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const long long y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    long long base = s_run, total = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const long long t = wsum[i];
      if (i < wave) base += t;
      total += t;
    }
    if (r < rows) {
      const long long e = base + x - v;
      rowoff[row_base + r] = (int)(e & 0xffffffffll);
      rowidx[row_base + r] = (int)(e >> 32);
    }
    __syncthreads();
    if (tid == 0) s_run += total;
    __syncthreads();
  }
  if (tid == 0) counts[img] = s_run & 0xffffffffll;
}

// fast_rcnn.py:141-158: the candidates of a row, classes ascending, at the row's offset: clipped box (Boxes.clip:
// x to [0, w], y to [0, h]), score, class, row
__global__ __launch_bounds__(64 * BH_WAVES) void bh_write_kernel(const BhImages I, const int* __restrict__ rowcnt,
                                                                 const int* __restrict__ rowoff,
                                                                 const int* __restrict__ rowidx,
                                                                 float4* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                                 int64_t* __restrict__ out_classes,
                                                                 int64_t* __restrict__ out_rows) {
  const int img = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *boxes, *scores;
  int rows, row_base; long cap_base; float h, w;
  bh_image(I, img, boxes, scores, rows, row_base, cap_base, h, w);
  const int r = blockIdx.x * BH_WAVES + wave;
  if (r >= rows) return;  // uniform per wave
  if (rowcnt[row_base + r] <= 0) return;  // (no candidate, or a dropped row)
  const int ridx = rowidx[row_base + r];
  const float* s = scores + (long)r * (I.K + 1);
  const float* b = boxes + (long)r * (I.Kb * 4);
  long pos = cap_base + rowoff[row_base + r];
  for (int k0 = 0; k0 < I.K; k0 += 64) {  // uniform
    const int k = k0 + lane;
    const float v = k < I.K ? s[k] : 0.f;
    const bool c = k < I.K && v > I.thr;
    const unsigned long long bal = __ballot(c);
    if (c) {
      const long p = pos + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
      const float* bk = b + (I.Kb == 1 ? 0 : k * 4);
      out_boxes[p] = make_float4(fminf(fmaxf(bk[0], 0.f), w), fminf(fmaxf(bk[1], 0.f), h), fminf(fmaxf(bk[2], 0.f), w),
                                 fminf(fmaxf(bk[3], 0.f), h));
      out_scores[p] = v;
      out_classes[p] = k;
      out_rows[p] = ridx;
    }
    pos += __builtin_popcountll(bal);
  }
}

// ---- the device-side continuation (no host sync between the filter and the NMS, none behind the NMS) -----------------
// The candidate WINDOW of an image = the first window[i] = min(window, rows[i] * K) slots of its slice of the candidate
// arrays (never beyond the slice: the next image's candidates start there).  Slots past the image's count are PARKED --
// zero box, score -inf, classes of their own (num_classes + slot / 64) -- so that an NMS over the whole window treats them as
// inert rows that sort last (the RPN path's convention for invalid proposals).
struct BhWindows {
  long cap_base[D2AMD_POOLER_MAX_IMAGES];
  int window[D2AMD_POOLER_MAX_IMAGES];
  int nms_row[D2AMD_POOLER_MAX_IMAGES];          // (take) the image's row of the NMS result; -1: no NMS ran (empty window)
  const int64_t* keep[D2AMD_POOLER_MAX_IMAGES];  // (take) the NMS's kept indices into the window
  int n, K, topk;
};
__device__ __forceinline__ void bh_window(const BhWindows& Wd, int img, long& base, int& window, int& nms_row,
                                          const int64_t*& keep) {
  base = Wd.cap_base[0]; window = Wd.window[0]; nms_row = Wd.nms_row[0]; keep = Wd.keep[0];
#pragma unroll
  for (int q = 1; q < D2AMD_POOLER_MAX_IMAGES; q++)
    if (q == img) { base = Wd.cap_base[q]; window = Wd.window[q]; nms_row = Wd.nms_row[q]; keep = Wd.keep[q]; }
}
__global__ __launch_bounds__(256) void bh_park_kernel(const BhWindows Wd, const int64_t* __restrict__ counts,
                                                      float4* __restrict__ boxes, float* __restrict__ scores,
                                                      int64_t* __restrict__ classes) {
  const int img = blockIdx.y, slot = blockIdx.x * 256 + threadIdx.x;
  long base; int window, nms_row; const int64_t* keep;
  bh_window(Wd, img, base, window, nms_row, keep);
  if (slot >= window || slot < counts[img]) return;
  boxes[base + slot] = make_float4(0.f, 0.f, 0.f, 0.f);
  scores[base + slot] = -INFINITY;
  classes[base + slot] = Wd.K + (slot >> 6);  // (a class per 64 parked slots: one 64 x 64 tile each for the NMS's mask, not (parked / 64)^2 / 2)
}
// fast_rcnn.py:161-170 at fixed shape: row t < topk of image i = candidate keep[t] of its window while t < min(kept,
// finite-score kept, topk); behind that a 1 x 1 box at the origin with score 0 / class 0 / row 0 (harmless for the mask
// pooler, mask inference and paste that follow).  nms_result: the NMS's {kept, flags, finite, 0} rows.
__global__ __launch_bounds__(256) void bh_take_kernel(const BhWindows Wd, const int64_t* __restrict__ nms_result,
                                                      const float4* __restrict__ boxes, const float* __restrict__ scores,
                                                      const int64_t* __restrict__ classes, const int64_t* __restrict__ rows,
                                                      float4* __restrict__ ob, float* __restrict__ os, int64_t* __restrict__ oc,
                                                      int64_t* __restrict__ orow, int64_t* __restrict__ ocount) {
  const int img = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  long base; int window, nms_row; const int64_t* keep;
  bh_window(Wd, img, base, window, nms_row, keep);
  int64_t nv = 0;
  if (nms_row >= 0) {
    const int64_t kept = nms_result[4 * nms_row], fin = nms_result[4 * nms_row + 2];
    nv = kept < fin ? kept : fin;
    nv = nv < 0 ? 0 : (nv > Wd.topk ? Wd.topk : nv);
  }
  if (t == 0) ocount[img] = nv;
  if (t >= Wd.topk) return;
  const long o = (long)img * Wd.topk + t;
  if (t < nv) {
    int64_t k = keep[t];
    k = k < 0 ? 0 : (k >= window ? window - 1 : k);
    const long p = base + k;
    ob[o] = boxes[p]; os[o] = scores[p]; oc[o] = classes[p]; orow[o] = rows[p];
  } else {
    ob[o] = make_float4(0.f, 0.f, 1.f, 1.f); os[o] = 0.f; oc[o] = 0; orow[o] = 0;
  }
}

// ---- FastRCNNOutputLayers.predict_boxes + predict_probs in one launch (fast_rcnn.py:524-568; box_regression.py:88-116) ----
// The reference runs ~45 elementwise launches here (slices, divisions, clamps, exp, stack, softmax); at 1,000 proposals per
// image every one of them is a launch latency.  A wave per proposal row: softmax (or sigmoid) of its K + 1 scores with
// two butterfly reductions, and the K_b class-specific boxes decoded with the reference's fp32 expression order (this
// file is compiled with -ffp-contract=off).  Rows at / behind an image's live count (a device-side proposal count: rows
// the RPN's NMS did not fill) predict nothing: all probability on the background column, zero boxes.
struct BhPredict {
  const float4* prop[D2AMD_POOLER_MAX_IMAGES];    // [R_i] proposal boxes (fp32 XYXY)
  const int64_t* limit[D2AMD_POOLER_MAX_IMAGES];  // nullable: {kept, flags, finite} of the NMS that produced the proposals
  int row_base[D2AMD_POOLER_MAX_IMAGES + 1];
  int n, K, Kb, sigmoid;
  float wx, wy, ww, wh, clamp;
};

constexpr int BHP_WAVES = 4;  // rows per workgroup of the predict kernel (2,000 rows -> 500 workgroups)
template <typename T>
__global__ __launch_bounds__(64 * BHP_WAVES) void bh_predict_kernel(const BhPredict P, const T* __restrict__ scores,
                                                                    const T* __restrict__ deltas,
                                                                    float4* __restrict__ boxes, T* __restrict__ probs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * BHP_WAVES + wave;
  if (row >= P.row_base[P.n]) return;  // uniform per wave
  const float4* prop = P.prop[0]; const int64_t* limit = P.limit[0]; int base = P.row_base[0], rows = P.row_base[1] - P.row_base[0];
#pragma unroll
  for (int q = 1; q < D2AMD_POOLER_MAX_IMAGES; q++)  // (constant indices only into the by-value struct; row_base[q >= n] = total)
    if (row >= P.row_base[q]) { prop = P.prop[q]; limit = P.limit[q]; base = P.row_base[q]; rows = P.row_base[q + 1] - P.row_base[q]; }
  const int r = row - base;
  const int K1 = P.K + 1;
  const T* s = scores + (long)row * K1;
  T* o = probs + (long)row * K1;
  float4* ob = boxes + (long)row * P.Kb;
  const vec4<T>* dv = reinterpret_cast<const vec4<T>*>(deltas + (long)row * (P.Kb * 4));
  // every load of the row is issued before anything is computed (the kernel is one memory round trip long): up to 128
  // scores and 128 class boxes live in registers; wider heads loop over the rest
  const bool small = K1 <= 128;
  float v0 = -INFINITY, v1 = -INFINITY;
  if (lane < K1) v0 = to_f32(s[lane]);
  if (lane + 64 < K1) v1 = to_f32(s[lane + 64]);
  vec4<T> d0{}, d1{};
  if (lane < P.Kb) d0 = dv[lane];
  if (lane + 64 < P.Kb) d1 = dv[lane + 64];
  const float4 b = prop[r];
  bool live = true;
  if (limit) {
    int64_t c = limit[0] < limit[2] ? limit[0] : limit[2];
    c = c < 0 ? 0 : (c > rows ? rows : c);
    live = r < c;
  }
  if (!live) {
    for (int k = lane; k < K1; k += 64) o[k] = from_f32<T>((k == P.K && !P.sigmoid) ? 1.f : 0.f);
    for (int k = lane; k < P.Kb; k += 64) ob[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  if (P.sigmoid) {
    for (int k = lane; k < K1; k += 64) {
      const float x = k == lane ? v0 : (k == lane + 64 ? v1 : to_f32(s[k]));
      o[k] = from_f32<T>(1.f / (1.f + expf(-x)));
    }
  } else {
    // softmax as ATen evaluates it: exp(x - max) / sum(exp(x - max)), fp32 accumulation
    float mx = fmaxf(v0, v1);
    if (!small)
      for (int k = lane + 128; k < K1; k += 64) mx = fmaxf(mx, to_f32(s[k]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    const float e0 = lane < K1 ? expf(v0 - mx) : 0.f, e1 = lane + 64 < K1 ? expf(v1 - mx) : 0.f;
    float sum = e0 + e1;
    if (!small)
      for (int k = lane + 128; k < K1; k += 64) sum += expf(to_f32(s[k]) - mx);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane < K1) o[lane] = from_f32<T>(e0 / sum);
    if (lane + 64 < K1) o[lane + 64] = from_f32<T>(e1 / sum);
    if (!small)
      for (int k = lane + 128; k < K1; k += 64) o[k] = from_f32<T>(expf(to_f32(s[k]) - mx) / sum);
  }
  const float widths = b.z - b.x, heights = b.w - b.y;
  const float ctr_x = b.x + 0.5f * widths, ctr_y = b.y + 0.5f * heights;
  for (int k = lane; k < P.Kb; k += 64) {
    float d[4];
    if (k == lane) unpack4(d0, d);
    else if (k == lane + 64) unpack4(d1, d);
    else unpack4(dv[k], d);
    // `deltas[:, 0::4] / wx` with a Python scalar on a device tensor is ATen's multiplication by the fp32 reciprocal
    // (BinaryDivTrueKernel: `a * (1 / b)` for a CPU-scalar divisor): P.wx .. P.wh hold 1.f / weight
    const float dx = d[0] * P.wx, dy = d[1] * P.wy;
    float dw = d[2] * P.ww, dh = d[3] * P.wh;
    dw = dw != dw ? dw : fminf(dw, P.clamp);   // torch.clamp(max=) keeps a NaN
    dh = dh != dh ? dh : fminf(dh, P.clamp);
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
    ob[k] = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
  }
}

// rows of a device-side proposal list at / behind its live count <- a 1 x 1 box at the origin (what the pooler and the
// decode that follow can take without looking at the count)
struct BhPad {
  float4* boxes[D2AMD_POOLER_MAX_IMAGES];
  const int64_t* limit[D2AMD_POOLER_MAX_IMAGES];
  int rows[D2AMD_POOLER_MAX_IMAGES];
};
__global__ __launch_bounds__(256) void bh_pad_kernel(const BhPad P) {
  const int img = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  float4* boxes = P.boxes[0]; const int64_t* limit = P.limit[0]; int rows = P.rows[0];
#pragma unroll
  for (int q = 1; q < D2AMD_POOLER_MAX_IMAGES; q++)
    if (q == img) { boxes = P.boxes[q]; limit = P.limit[q]; rows = P.rows[q]; }
  if (t >= rows) return;
  int64_t c = limit[0] < limit[2] ? limit[0] : limit[2];
  c = c < 0 ? 0 : c;
  if (t >= c) boxes[t] = make_float4(0.f, 0.f, 1.f, 1.f);
}

}  // namespace d2amd

using namespace d2amd;

static int bh_windows(BhWindows& Wd, const int* rows, int num_images, int num_classes, int window, const char* who) {
  D2_CHECK_ARG(num_images >= 1 && num_images <= D2AMD_POOLER_MAX_IMAGES, "%s: %d images (max %d)", who, num_images,
               D2AMD_POOLER_MAX_IMAGES);
  D2_CHECK_ARG(rows && num_classes >= 1 && window >= 1, "%s: bad arguments", who);
  Wd.n = num_images; Wd.K = num_classes;
  long base = 0;
  for (int i = 0; i < num_images; i++) {
    const long slice = (long)(rows[i] > 0 ? rows[i] : 0) * num_classes;
    Wd.cap_base[i] = base;
    Wd.window[i] = (int)(slice < window ? slice : window);
    Wd.nms_row[i] = -1;
    base += slice;
  }
  return D2AMD_OK;
}

extern "C" int d2amd_fast_rcnn_park(const int* rows, int num_images, int num_classes, int window, const int64_t* counts,
                                    float* out_boxes, float* out_scores, int64_t* out_classes, void* stream) {
  BhWindows Wd{};
  const int rc = bh_windows(Wd, rows, num_images, num_classes, window, "fast_rcnn_park");
  if (rc) return rc;
  D2_CHECK_ARG(counts && out_boxes && out_scores && out_classes, "fast_rcnn_park: null pointer");
  hipLaunchKernelGGL(bh_park_kernel, dim3(cdiv(window, 256), num_images), dim3(256), 0, (hipStream_t)stream, Wd, counts,
                     (float4*)out_boxes, out_scores, out_classes);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

extern "C" int d2amd_fast_rcnn_take(const int* rows, int num_images, int num_classes, int window, int topk,
                                    const int64_t* const* keep, const int64_t* nms_result, const float* cand_boxes,
                                    const float* cand_scores, const int64_t* cand_classes, const int64_t* cand_rows,
                                    float* det_boxes, float* det_scores, int64_t* det_classes, int64_t* det_rows,
                                    int64_t* det_counts, void* stream) {
  BhWindows Wd{};
  const int rc = bh_windows(Wd, rows, num_images, num_classes, window, "fast_rcnn_take");
  if (rc) return rc;
  D2_CHECK_ARG(topk >= 1 && keep && nms_result && cand_boxes && cand_scores && cand_classes && cand_rows && det_boxes &&
               det_scores && det_classes && det_rows && det_counts, "fast_rcnn_take: bad arguments");
  Wd.topk = topk;
  int row = 0;
  for (int i = 0; i < num_images; i++) {  // (images with an empty window took no part in the NMS: keep[i] may be null)
    if (Wd.window[i] == 0) continue;
    D2_CHECK_ARG(keep[i] != nullptr, "fast_rcnn_take: image %d: null keep", i);
    Wd.keep[i] = keep[i];
    Wd.nms_row[i] = row++;
  }
  hipLaunchKernelGGL(bh_take_kernel, dim3(cdiv(topk, 256), num_images), dim3(256), 0, (hipStream_t)stream, Wd, nms_result,
                     (const float4*)cand_boxes, cand_scores, cand_classes, cand_rows, (float4*)det_boxes, det_scores,
                     det_classes, det_rows, det_counts);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}
extern "C" size_t d2amd_fast_rcnn_filter_workspace_bytes(const int* rows, int num_images) {
  long total = 0;
  for (int i = 0; i < num_images; i++) total += rows[i] > 0 ? rows[i] : 0;
  return (size_t)(3 * total + 64) * sizeof(int);
}

extern "C" int d2amd_fast_rcnn_filter(const float* const* boxes, const float* const* scores, const int* rows,
                                      int num_images, int num_classes, int num_bbox_reg_classes, const int* image_hw,
                                      float score_thresh, float* out_boxes, float* out_scores, int64_t* out_classes,
                                      int64_t* out_rows, int64_t* counts, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  D2_CHECK_ARG(num_images >= 0 && num_images <= D2AMD_POOLER_MAX_IMAGES, "fast_rcnn_filter: %d images (max %d)",
               num_images, D2AMD_POOLER_MAX_IMAGES);
  if (num_images == 0) return D2AMD_OK;
  D2_CHECK_ARG(num_classes >= 1 && (num_bbox_reg_classes == 1 || num_bbox_reg_classes == num_classes),
               "fast_rcnn_filter: %d classes, boxes for %d", num_classes, num_bbox_reg_classes);
  D2_CHECK_ARG(boxes && scores && rows && image_hw && counts && workspace, "fast_rcnn_filter: null pointer");
  D2_CHECK_ARG(workspace_bytes >= d2amd_fast_rcnn_filter_workspace_bytes(rows, num_images),
               "fast_rcnn_filter: workspace too small");
  BhImages I{};
  I.n = num_images; I.K = num_classes; I.Kb = num_bbox_reg_classes; I.thr = score_thresh;
  int max_rows = 0;
  for (int i = 0; i < num_images; i++) {
    D2_CHECK_ARG(rows[i] >= 0 && (rows[i] == 0 || (boxes[i] && scores[i])), "fast_rcnn_filter: image %d: bad rows / pointers", i);
    D2_CHECK_ARG((long)rows[i] * num_classes < (1l << 31), "fast_rcnn_filter: too many (row, class) pairs");
    I.boxes[i] = boxes[i]; I.scores[i] = scores[i]; I.rows[i] = rows[i];
    I.h[i] = (float)image_hw[2 * i]; I.w[i] = (float)image_hw[2 * i + 1];
    I.row_base[i + 1] = I.row_base[i] + rows[i];
    I.cap_base[i + 1] = I.cap_base[i] + (long)rows[i] * num_classes;
    max_rows = rows[i] > max_rows ? rows[i] : max_rows;
  }
  hipStream_t st = (hipStream_t)stream;
  int* rowcnt = (int*)workspace;
  int* rowoff = rowcnt + I.row_base[num_images];
  int* rowidx = rowoff + I.row_base[num_images];
  if (max_rows == 0) {
    const int zrc = zero_async(counts, (size_t)num_images * 8, st);
    return zrc;
  }
  D2_CHECK_ARG(out_boxes && out_scores && out_classes && out_rows, "fast_rcnn_filter: null output");
  const dim3 grid(cdiv(max_rows, BH_WAVES), num_images);
  hipLaunchKernelGGL(bh_count_kernel, grid, dim3(64 * BH_WAVES), 0, st, I, rowcnt);
  D2_LAUNCH_OK();
  hipLaunchKernelGGL(bh_scan_kernel, dim3(num_images), dim3(1024), 0, st, I, (const int*)rowcnt, rowoff, rowidx, counts);
  D2_LAUNCH_OK();
  hipLaunchKernelGGL(bh_write_kernel, grid, dim3(64 * BH_WAVES), 0, st, I, (const int*)rowcnt, (const int*)rowoff,
                     (const int*)rowidx, (float4*)out_boxes, out_scores, out_classes, out_rows);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

extern "C" int d2amd_fast_rcnn_predict(const void* scores, const void* deltas, int dtype, const float* const* proposals,
                                       const int64_t* const* limits, const int* rows, int num_images, int num_classes,
                                       int num_bbox_reg_classes, const float* weights, float scale_clamp, int use_sigmoid,
                                       float* boxes_out, void* probs_out, void* stream) {
  D2_CHECK_ARG(num_images >= 1 && num_images <= D2AMD_POOLER_MAX_IMAGES, "fast_rcnn_predict: %d images (max %d)", num_images,
               D2AMD_POOLER_MAX_IMAGES);
  D2_CHECK_ARG(scores && deltas && proposals && rows && weights && boxes_out && probs_out, "fast_rcnn_predict: null pointer");
  D2_CHECK_ARG(num_classes >= 1 && (num_bbox_reg_classes == num_classes || num_bbox_reg_classes == 1),
               "fast_rcnn_predict: %d box classes for %d classes", num_bbox_reg_classes, num_classes);
  D2_CHECK_ARG(weights[0] != 0.f && weights[1] != 0.f && weights[2] != 0.f && weights[3] != 0.f, "fast_rcnn_predict: zero weight");
  BhPredict P{};
  P.n = num_images; P.K = num_classes; P.Kb = num_bbox_reg_classes; P.sigmoid = use_sigmoid ? 1 : 0;
  P.wx = 1.f / weights[0]; P.wy = 1.f / weights[1]; P.ww = 1.f / weights[2]; P.wh = 1.f / weights[3]; P.clamp = scale_clamp;
  int total = 0;
  for (int i = 0; i < num_images; i++) {
    D2_CHECK_ARG(rows[i] >= 0 && (rows[i] == 0 || proposals[i]), "fast_rcnn_predict: image %d: %d rows, proposals %p", i, rows[i],
                 (const void*)proposals[i]);
    P.prop[i] = (const float4*)proposals[i];
    P.limit[i] = limits ? limits[i] : nullptr;
    P.row_base[i] = total;
    total += rows[i];
  }
  for (int i = num_images; i <= D2AMD_POOLER_MAX_IMAGES; i++) P.row_base[i] = total;
  if (total == 0) return D2AMD_OK;
  const dim3 grid(cdiv(total, BHP_WAVES)), block(64 * BHP_WAVES);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case D2AMD_F32: hipLaunchKernelGGL(bh_predict_kernel<float>, grid, block, 0, st, P, (const float*)scores, (const float*)deltas, (float4*)boxes_out, (float*)probs_out); break;
    case D2AMD_F16: hipLaunchKernelGGL(bh_predict_kernel<f16_t>, grid, block, 0, st, P, (const f16_t*)scores, (const f16_t*)deltas, (float4*)boxes_out, (f16_t*)probs_out); break;
    case D2AMD_BF16: hipLaunchKernelGGL(bh_predict_kernel<bf16_t>, grid, block, 0, st, P, (const bf16_t*)scores, (const bf16_t*)deltas, (float4*)boxes_out, (bf16_t*)probs_out); break;
    default: D2_CHECK_ARG(false, "fast_rcnn_predict: dtype %d", dtype);
  }
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

extern "C" int d2amd_proposals_pad(float* const* boxes, const int64_t* const* limits, const int* rows, int num_images,
                                   void* stream) {
  D2_CHECK_ARG(num_images >= 1 && num_images <= D2AMD_POOLER_MAX_IMAGES, "proposals_pad: %d images (max %d)", num_images,
               D2AMD_POOLER_MAX_IMAGES);
  D2_CHECK_ARG(boxes && limits && rows, "proposals_pad: null pointer");
  BhPad P{};
  int mx = 0;
  for (int i = 0; i < num_images; i++) {
    D2_CHECK_ARG(rows[i] >= 0 && limits[i] && (rows[i] == 0 || boxes[i]), "proposals_pad: image %d: bad arguments", i);
    P.boxes[i] = (float4*)boxes[i]; P.limit[i] = limits[i]; P.rows[i] = rows[i];
    mx = rows[i] > mx ? rows[i] : mx;
  }
  if (mx == 0) return D2AMD_OK;
  hipLaunchKernelGGL(bh_pad_kernel, dim3(cdiv(mx, 256), num_images), dim3(256), 0, (hipStream_t)stream, P);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}
