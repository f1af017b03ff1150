// Mask-head training targets from POLYGONS: PolygonMasks.crop_and_resize on the device.
//   replaces  detectron2/structures/masks.py:396-420 (PolygonMasks.crop_and_resize: a Python loop over the instances,
//             each `rasterize_polygons_within_box` :39-86 = shift / scale the instance's polygons into the box's
//             mask_size x mask_size frame, then pycocotools frPyObjects + merge + decode on the CPU, stacked and copied
//             to the device -- COCO's default mask format; the GPU idles meanwhile).
// pycocotools is not part of the reference tree: the rasteriser restates the published cocoapi algorithm
// (common/maskApi.c rleFrPoly / rleMerge / rleDecode; see oracle/d2_oracle.c orc_poly_to_mask, which pins it to the
// reference's own known answer).  It is a boundary-crossing PARITY rule, which parallelises without any sort:
//   * coordinates are upsampled by 5 and truncated; every edge is walked densely (max(|dx|, |dy|) + 1 points, the
//     minor coordinate from the rounded slope); the walks of all edges form ONE point sequence;
//   * wherever consecutive points differ in x, a crossing (column, y) is recorded after downsampling (columns that do
//     not fall on a pixel centre or outside the mask are dropped, y is clamped to [0, h] and rounded up);
//   * the run-length code sorts the column-major positions col * h + y; a pixel with column-major index t is inside
//     iff an ODD number of crossings has position <= t.  Several polygons of an instance: OR of their masks.
// Here: one workgroup per output mask.  Threads take the points of the sequence (point g -> its edge by binary search
// in the prefix of the edges' point counts, its predecessor recomputed), crossings become +1 on an LDS counter per
// position, a workgroup scan turns the counters into parities.  Double precision where the reference uses it
// (polygons are float64, the box is float32 = `box.numpy()`, ratios are float32 `mask_size / max(extent, 0.1)`).
// Roofline: none worth naming -- a few KB per mask; the point is that the targets never leave the device pipeline.
#include "common.h"

namespace d2amd {

constexpr int PM_THREADS = 256;
constexpr int PM_MAX_VERTS = 4096;  // vertices per polygon (LDS: 3 x 16 KB)
constexpr int PM_MAX_M = 64;        // mask_size (LDS counters: M * M + 1)

__device__ __forceinline__ int pm_int_of(double v) {  // (int) of a double as the x86 reference does: NaN / range -> INT_MIN
  if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
  return (int)v;
}

// points of the walk of an edge: max(|dx|, |dy|) + 1
__device__ __forceinline__ long long pm_edge_points(int xs, int ys, int xe, int ye) {
  long long ax = (long long)xs - xe, ay = (long long)ys - ye;
  ax = ax < 0 ? -ax : ax;
  ay = ay < 0 ? -ay : ay;
  return (ax > ay ? ax : ay) + 1;
}

// point d of the walk of edge (xs, ys) -> (xe, ye)  [maskApi.c rleFrPoly]
__device__ __forceinline__ void pm_point(int xs, int ys, int xe, int ye, int d, int& u, int& v) {
  const int dx = abs(xe - xs), dy = abs(ys - ye);
  const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
  if (flip) { int t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
  if (dx >= dy) {
    const double s = (double)(ye - ys) / (double)dx;  // 0 / 0 = NaN for a degenerate edge, like the reference
    const int t = flip ? dx - d : d;
    u = t + xs;
    v = pm_int_of((double)ys + s * (double)t + .5);
  } else {
    const double s = (double)(xe - xs) / (double)dy;
    const int t = flip ? dy - d : d;
    v = t + ys;
    u = pm_int_of((double)xs + s * (double)t + .5);
  }
}

__global__ __launch_bounds__(PM_THREADS) void polygon_crop_kernel(
    const double* __restrict__ coords, const int64_t* __restrict__ poly_off, const int64_t* __restrict__ inst_off,
    int n_inst, const float* __restrict__ boxes, const int64_t* __restrict__ index, int M,
    uint8_t* __restrict__ out, int* __restrict__ status) {
  __shared__ int s_x[PM_MAX_VERTS + 1], s_y[PM_MAX_VERTS + 1];
  __shared__ int s_pre[PM_MAX_VERTS + 1];  // s_pre[j] = points of the edges before edge j
  __shared__ unsigned s_cnt[PM_MAX_M * PM_MAX_M + 1];
  __shared__ uint8_t s_mask[PM_MAX_M * PM_MAX_M];
  __shared__ int s_wave[PM_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int k_out = blockIdx.x;
  const int hw = M * M;
  uint8_t* o = out + (long)k_out * hw;
  long inst = k_out;
  if (index) inst = index[k_out];
  if (inst < 0 || inst >= n_inst) {  // torch indexing raises IndexError: flag, zeros (uniform)
    if (tid == 0 && status) atomicOr(status, 1);
    for (int t = tid; t < hw; t += PM_THREADS) o[t] = 0;
    return;
  }
  for (int t = tid; t < hw; t += PM_THREADS) s_mask[t] = 0;
  // masks.py:64-80: w, h in float32 (box.numpy()), ratio = mask_size / max(extent, 0.1)
  const float* b = boxes + (long)k_out * 4;
  const float bw = b[2] - b[0], bh = b[3] - b[1];
  const double rw = (double)bw > 0.1 ? (double)((float)M / bw) : (double)M / 0.1;
  const double rh = (double)bh > 0.1 ? (double)((float)M / bh) : (double)M / 0.1;
  const double ox = (double)b[0], oy = (double)b[1];
  for (long p = inst_off[inst]; p < inst_off[inst + 1]; p++) {
    const long c0 = poly_off[p];
    const int k = (int)((poly_off[p + 1] - c0) / 2);
    if (k <= 0) continue;  // uniform
    if (k > PM_MAX_VERTS) {  // uniform
      if (tid == 0 && status) atomicOr(status, 2);
      continue;
    }
    __syncthreads();  // previous polygon's readers are done
    for (int j = tid; j < k; j += PM_THREADS) {
      const double X = (coords[c0 + 2 * j] - ox) * rw, Y = (coords[c0 + 2 * j + 1] - oy) * rh;
      s_x[j] = pm_int_of(5.0 * X + .5);
      s_y[j] = pm_int_of(5.0 * Y + .5);
    }
    for (int t = tid; t <= hw; t += PM_THREADS) s_cnt[t] = 0u;
    __syncthreads();
    if (tid == 0) { s_x[k] = s_x[0]; s_y[k] = s_y[0]; }
    __syncthreads();
    // exclusive prefix of the edges' point counts (workgroup scan over chunks of consecutive edges)
    const int per = (k + PM_THREADS - 1) / PM_THREADS;
    const int j0 = min(tid * per, k), j1 = min(j0 + per, k);
    long long sum = 0;
    for (int j = j0; j < j1; j++) {
      sum += pm_edge_points(s_x[j], s_y[j], s_x[j + 1], s_y[j + 1]);
    }
    sum = sum < (1ll << 30) ? sum : (1ll << 30);  // saturate: 2^30 points is refused below
    long long x = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const long long y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) s_wave[wv] = (int)(x < 0x7fffffffll ? x : 0x7fffffffll);
    __syncthreads();
    long long base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < PM_THREADS / 64; i++) {
      if (i < wv) base += s_wave[i];
      total += s_wave[i];
    }
    if (total >= (1ll << 30)) {  // uniform: a walk of 2^30 points is out of any real use (and of int range)
      if (tid == 0 && status) atomicOr(status, 4);
      continue;
    }
    long long run = base + x - sum;
    for (int j = j0; j < j1; j++) {
      s_pre[j] = (int)run;
      run += pm_edge_points(s_x[j], s_y[j], s_x[j + 1], s_y[j + 1]);
    }
    if (tid == 0) s_pre[k] = (int)total;
    __syncthreads();
    const int m = (int)total;
    // crossings: every point g >= 1 of the sequence against its predecessor
    for (int g = tid + 1; g < m; g += PM_THREADS) {
      int lo = 0, hi = k;  // last edge e with s_pre[e] <= g
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_pre[mid] <= g) lo = mid; else hi = mid;
      }
      const int e = lo, d = g - s_pre[e];
      int u, v, pu, pv;
      pm_point(s_x[e], s_y[e], s_x[e + 1], s_y[e + 1], d, u, v);
      if (d > 0) pm_point(s_x[e], s_y[e], s_x[e + 1], s_y[e + 1], d - 1, pu, pv);
      else pm_point(s_x[e - 1], s_y[e - 1], s_x[e], s_y[e], s_pre[e] - s_pre[e - 1] - 1, pu, pv);
      if (u == pu) continue;
      double xd = (double)(u < pu ? u : u - 1);
      xd = (xd + .5) / 5.0 - .5;
      if (floor(xd) != xd || xd < 0 || xd > (double)(M - 1)) continue;
      double yd = (double)(v < pv ? v : pv);
      yd = (yd + .5) / 5.0 - .5;
      if (yd < 0) yd = 0; else if (yd > (double)M) yd = (double)M;
      yd = ceil(yd);
      atomicAdd(&s_cnt[(int)xd * M + (int)yd], 1u);
    }
    __syncthreads();
    // parity of the crossings at positions <= t (column-major t): workgroup scan of the counters
    const int per2 = (hw + PM_THREADS - 1) / PM_THREADS;
    const int t0 = min(tid * per2, hw), t1 = min(t0 + per2, hw);
    unsigned par = 0;
    for (int t = t0; t < t1; t++) par ^= s_cnt[t] & 1u;
    unsigned xs_ = par;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned y = (unsigned)__shfl_up((int)xs_, d, 64);
      if (lane >= d) xs_ ^= y;
    }
    __syncthreads();  // s_wave of the first scan has been read
    if (lane == 63) s_wave[wv] = (int)xs_;
    __syncthreads();
    unsigned before = xs_ ^ par;  // exclusive inside the wave
#pragma unroll
    for (int i = 0; i < PM_THREADS / 64; i++)
      if (i < wv) before ^= (unsigned)s_wave[i];
    unsigned cur = before;
    for (int t = t0; t < t1; t++) {
      cur ^= s_cnt[t] & 1u;
      if (cur) s_mask[(t % M) * M + (t / M)] = 1;  // t = column * h + row -> row-major output
    }
  }
  __syncthreads();
  for (int t = tid; t < hw; t += PM_THREADS) o[t] = s_mask[t];
}

}  // namespace d2amd

using namespace d2amd;

extern "C" int d2amd_polygon_crop_and_resize(const double* coords, const int64_t* poly_offsets,
                                             const int64_t* inst_offsets, int n_instances, const float* boxes,
                                             const int64_t* index, int n_boxes, int mask_size, uint8_t* out,
                                             int* status, void* stream) {
  D2_CHECK_ARG(n_boxes >= 0 && n_instances >= 0, "polygon_crop_and_resize: negative count");
  D2_CHECK_ARG(mask_size > 0 && mask_size <= PM_MAX_M, "polygon_crop_and_resize: mask_size %d (1..%d)", mask_size, PM_MAX_M);
  if (n_boxes == 0) return D2AMD_OK;
  D2_CHECK_ARG(poly_offsets && inst_offsets && boxes && out, "polygon_crop_and_resize: null pointer");
  D2_CHECK_ARG(index != nullptr || n_boxes == n_instances,
               "polygon_crop_and_resize: %d boxes for %d instances and no index", n_boxes, n_instances);
  hipLaunchKernelGGL(polygon_crop_kernel, dim3(n_boxes), dim3(PM_THREADS), 0, (hipStream_t)stream, coords, poly_offsets,
                     inst_offsets, n_instances, boxes, index, mask_size, out, status);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}
