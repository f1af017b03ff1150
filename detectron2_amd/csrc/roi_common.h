// Shared device code of the ROIAlign family (roi_align.hip, roi_pool.hip): per-ROI geometry, the
// per-axis bilinear tap classification and the separable per-bin tables.
//   geometry / taps <- torchvision.ops.roi_align as called by detectron2/layers/roi_align.py:58-65 and
//                      csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp:27-129,201-310
#pragma once
#include "common.h"

namespace d2amd {

// ------------------------------------------------------------------------------------------
// per-ROI geometry
struct RoiGeom {
  int batch;
  float start_w, start_h;   // axis-aligned: roi start; rotated: -w/2, -h/2 (relative to centre)
  float bin_w, bin_h;
  float roi_w, roi_h;       // extent after scaling (legacy mode: clamped to >= 1)
  int grid_w, grid_h;
  float center_w, center_h, cos_t, sin_t;  // rotated only
  bool bad;                                 // rotated: negative size
};

// axis-aligned geometry from the 5 ROI values (shared by every axis-aligned kernel so that the list
// scan, the forward and the backward see bit-identical starts / bins / grids)
__device__ __forceinline__ RoiGeom roi_geom_box(float b, float x1, float y1, float x2, float y2, float scale,
                                                int pooled_h, int pooled_w, int sampling_ratio, int aligned) {
  RoiGeom g;
  g.bad = false;
  g.batch = (int)b;
  // bits 1.. of `aligned`: the ROI coordinates rounded to fp16 (1) / bf16 (2) first -- d2amd_pooler_params.roi_rounding:
  // what the reference's cast of the ROIs to the feature dtype (layers/roi_align.py:60) makes of them
  const int rmode = aligned >> 1;
  aligned &= 1;
  if (rmode == 1) {
    x1 = (float)(_Float16)x1; y1 = (float)(_Float16)y1; x2 = (float)(_Float16)x2; y2 = (float)(_Float16)y2;
  } else if (rmode == 2) {
    x1 = (float)(__bf16)x1; y1 = (float)(__bf16)y1; x2 = (float)(__bf16)x2; y2 = (float)(__bf16)y2;
  }
  const float off = aligned ? 0.5f : 0.0f;
  g.start_w = x1 * scale - off;
  g.start_h = y1 * scale - off;
  const float end_w = x2 * scale - off, end_h = y2 * scale - off;
  float roi_w = end_w - g.start_w;
  float roi_h = end_h - g.start_h;
  if (!aligned) {  // legacy: force malformed ROIs to be 1x1
    roi_w = fmaxf(roi_w, 1.f);
    roi_h = fmaxf(roi_h, 1.f);
  }
  g.center_w = g.center_h = 0.f;
  g.cos_t = 1.f;
  g.sin_t = 0.f;
  g.roi_w = roi_w;
  g.roi_h = roi_h;
  g.bin_h = roi_h / (float)pooled_h;
  g.bin_w = roi_w / (float)pooled_w;
  g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / (float)pooled_h);
  g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / (float)pooled_w);
  return g;
}

template <bool ROT>
__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ rois, int k, float scale, int pooled_h,
                                            int pooled_w, int sampling_ratio, int aligned) {
  if (!ROT) {
    const float* r = rois + (long)k * 5;
    return roi_geom_box(r[0], r[1], r[2], r[3], r[4], scale, pooled_h, pooled_w, sampling_ratio, aligned);
  }
  RoiGeom g;
  g.bad = false;
  float roi_w, roi_h;
  {
    const float* r = rois + (long)k * 6;
    g.batch = (int)r[0];
    g.center_w = r[1] * scale - 0.5f;
    g.center_h = r[2] * scale - 0.5f;
    roi_w = r[3] * scale;
    roi_h = r[4] * scale;
    // ROIAlignRotated_cpu.cpp:232-234 with T=float: theta rounded to float, cos/sin in double
    const float theta = (float)((double)r[5] * 3.14159265358979323846 / 180.0);
    g.cos_t = (float)cos((double)theta);
    g.sin_t = (float)sin((double)theta);
    g.bad = !(roi_w >= 0.f && roi_h >= 0.f);
    g.start_h = (float)(-(double)roi_h / 2.0);
    g.start_w = (float)(-(double)roi_w / 2.0);
  }
  g.roi_w = roi_w;
  g.roi_h = roi_h;
  g.bin_h = roi_h / (float)pooled_h;
  g.bin_w = roi_w / (float)pooled_w;
  g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / (float)pooled_h);
  g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / (float)pooled_w);
  return g;
}

// one axis of the bilinear footprint (ROIAlignRotated_cpu.cpp:64-107 per axis)
struct AxisTap {
  int lo, hi;
  float wlo, whi;  // weight of lo / hi pixel; 0 when the sample is outside [-1, size]
  bool valid;
};
__device__ __forceinline__ AxisTap axis_tap(float y, int size) {
  AxisTap t;
  t.valid = !(y < -1.0f || y > (float)size);
  if (y < 0.f) y = 0.f;
  int lo = (int)y;
  int hi;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    y = (float)lo;
  } else {
    hi = lo + 1;
  }
  const float l = y - (float)lo;
  t.lo = lo; t.hi = hi;
  t.whi = l; t.wlo = 1.f - l;
  if (!t.valid) { t.lo = t.hi = 0; t.wlo = t.whi = 0.f; }
  return t;
}

__device__ __forceinline__ float sample_pos(float start, int p, float bin, int i, int grid) {
  // roi_start + ph*bin + (iy + .5f) * bin / grid   (same expression order as the reference)
  return start + (float)p * bin + ((float)i + .5f) * bin / (float)grid;
}

// ------------------------------------------------------------------------------------------
// SEPARABLE per-bin tables (axis-aligned).  For bin index t along one axis:
//   first[t]  first pixel row/col touched, span[t] number of consecutive pixels, wt[t*SPAN + j].
// Built by threads t < P of the workgroup; rows touched by the g samples of a bin are
// consecutive because samples are <= 1 px apart when g = ceil(bin) (and for fixed g they are
// inserted at lo-first offsets; span is bounded by SPAN = max offset + 1 and checked).
struct AxisTables {
  int* first;   // [P]
  int* span;    // [P]
  float* wt;    // [P * SPAN]
};

// returns false if some bin needs more than SPAN entries (caller falls back to direct kernel)
__device__ __forceinline__ bool build_axis(int t, float start, float bin, int grid, int size, int SPAN, int* first,
                                           int* span, float* wt) {
  float* w = wt + t * SPAN;
  for (int j = 0; j < SPAN; j++) w[j] = 0.f;
  int f = 0x7fffffff, l = -1;
  for (int i = 0; i < grid; i++) {
    const AxisTap a = axis_tap(sample_pos(start, t, bin, i, grid), size);
    if (!a.valid) continue;
    f = min(f, a.lo);
    l = max(l, a.hi);
  }
  if (l < 0) { first[t] = 0; span[t] = 0; return true; }
  first[t] = f;
  span[t] = l - f + 1;
  if (l - f + 1 > SPAN) return false;
  for (int i = 0; i < grid; i++) {
    const AxisTap a = axis_tap(sample_pos(start, t, bin, i, grid), size);
    if (!a.valid) continue;
    w[a.lo - f] += a.wlo;
    w[a.hi - f] += a.whi;
  }
  return true;
}

constexpr int SEP_SPAN = 12;   // table entries per bin and axis (covers g <= 10 with g = ceil(bin))
constexpr int SEP_MAXP = 32;   // max pooled size per axis on the fast path

struct SepShared {
  int firsty[SEP_MAXP], spany[SEP_MAXP], firstx[SEP_MAXP], spanx[SEP_MAXP];
  float wy[SEP_MAXP * SEP_SPAN], wx[SEP_MAXP * SEP_SPAN];
  int ok;
  int batch;
  float inv_count;
};

template <bool BWD>
__device__ __forceinline__ void sep_build(SepShared& S, const float* rois, int k, float scale, int PH, int PW,
                                          int sampling_ratio, int aligned, int H, int W) {
  if (threadIdx.x == 0) S.ok = 1;
  __syncthreads();
  const RoiGeom g = roi_geom<false>(rois, k, scale, PH, PW, sampling_ratio, aligned);
  const int t = threadIdx.x;
  bool ok = true;
  if (t < PH) ok = build_axis(t, g.start_h, g.bin_h, g.grid_h, H, SEP_SPAN, S.firsty, S.spany, S.wy);
  else if (t < PH + PW) ok = build_axis(t - PH, g.start_w, g.bin_w, g.grid_w, W, SEP_SPAN, S.firstx, S.spanx, S.wx);
  if (!ok) S.ok = 0;
  if (t == 0) {
    S.batch = g.batch;
    const int cnt = g.grid_h * g.grid_w;
    S.inv_count = 1.f / (float)(cnt > 0 ? cnt : 1);
  }
  __syncthreads();
}

// Inline fallback for one (ROI, channel range) when the separable tables overflow (very large
// bins with a fixed sampling_ratio): direct per-sample taps, executed by the same workgroup.
template <typename T, bool NHWC_>
__device__ void fwd_direct_range(const T* __restrict__ in, const float* __restrict__ rois, T* __restrict__ out,
                                 int k, int c0, int nc, int C, int H, int W, int PH, int PW, float scale,
                                 int sampling_ratio, int aligned, int b_lo = 0, int b_hi = -1) {
  const RoiGeom g = roi_geom<false>(rois, k, scale, PH, PW, sampling_ratio, aligned);
  const float count = (float)max(g.grid_h * g.grid_w, 1);
  const long plane = (long)H * W;
  const int bins = PH * PW;
  if (b_hi < 0) b_hi = bins;
  const int nb = b_hi - b_lo;  // bins [b_lo, b_hi) of this ROI
  for (int e = threadIdx.x; e < nc * nb; e += blockDim.x) {
    int c, b;
    if (NHWC_) { b = e / nc; c = c0 + (e - b * nc); b += b_lo; } else { c = c0 + e / nb; b = b_lo + e % nb; }
    const int ph = b / PW, pw = b - ph * PW;
    const T* base = NHWC_ ? in + (long)g.batch * plane * C + c : in + ((long)g.batch * C + c) * plane;
    const long pstride = NHWC_ ? C : 1;
    float acc = 0.f;
    for (int iy = 0; iy < g.grid_h; iy++) {
      const AxisTap ty = axis_tap(sample_pos(g.start_h, ph, g.bin_h, iy, g.grid_h), H);
      for (int ix = 0; ix < g.grid_w; ix++) {
        const AxisTap tx = axis_tap(sample_pos(g.start_w, pw, g.bin_w, ix, g.grid_w), W);
        acc += (ty.wlo * tx.wlo) * to_f32(base[((long)ty.lo * W + tx.lo) * pstride]) +
            (ty.wlo * tx.whi) * to_f32(base[((long)ty.lo * W + tx.hi) * pstride]) +
            (ty.whi * tx.wlo) * to_f32(base[((long)ty.hi * W + tx.lo) * pstride]) +
            (ty.whi * tx.whi) * to_f32(base[((long)ty.hi * W + tx.hi) * pstride]);
      }
    }
    const long o = NHWC_ ? ((long)k * bins + b) * C + c : ((long)k * C + c) * bins + b;
    out[o] = from_f32<T>(acc / count);
  }
}

template <typename T, bool NHWC_>
__device__ void bwd_direct_range(const T* __restrict__ gout, const float* __restrict__ rois,
                                 float* __restrict__ gin, int k, int c0, int nc, int C, int H, int W, int PH,
                                 int PW, float scale, int sampling_ratio, int aligned) {
  const RoiGeom g = roi_geom<false>(rois, k, scale, PH, PW, sampling_ratio, aligned);
  const float count = (float)(g.grid_h * g.grid_w);
  const long plane = (long)H * W;
  const int bins = PH * PW;
  for (int e = threadIdx.x; e < nc * bins; e += blockDim.x) {
    int c, b;
    if (NHWC_) { b = e / nc; c = c0 + (e - b * nc); } else { c = c0 + e / bins; b = e % bins; }
    const int ph = b / PW, pw = b - ph * PW;
    const long o = NHWC_ ? ((long)k * bins + b) * C + c : ((long)k * C + c) * bins + b;
    const float go = to_f32(gout[o]);
    float* base = NHWC_ ? gin + (long)g.batch * plane * C + c : gin + ((long)g.batch * C + c) * plane;
    const long pstride = NHWC_ ? C : 1;
    for (int iy = 0; iy < g.grid_h; iy++) {
      const AxisTap ty = axis_tap(sample_pos(g.start_h, ph, g.bin_h, iy, g.grid_h), H);
      for (int ix = 0; ix < g.grid_w; ix++) {
        const AxisTap tx = axis_tap(sample_pos(g.start_w, pw, g.bin_w, ix, g.grid_w), W);
        if (!(ty.valid && tx.valid)) continue;
        atomicAdd(base + ((long)ty.lo * W + tx.lo) * pstride, go * (ty.wlo * tx.wlo) / count);
        atomicAdd(base + ((long)ty.lo * W + tx.hi) * pstride, go * (ty.wlo * tx.whi) / count);
        atomicAdd(base + ((long)ty.hi * W + tx.lo) * pstride, go * (ty.whi * tx.wlo) / count);
        atomicAdd(base + ((long)ty.hi * W + tx.hi) * pstride, go * (ty.whi * tx.whi) / count);
      }
    }
  }
}

}  // namespace d2amd
