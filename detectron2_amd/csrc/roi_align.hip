// ROIAlign / ROIAlignRotated forward + backward for gfx950.
//   forward  <- torchvision.ops.roi_align as called by detectron2/layers/roi_align.py:58-65, and
//               csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp:201-310 (rotated)
//   backward <- torchvision's roi_align backward; ROIAlignRotated_cpu.cpp:312-416
// Roofline class: HBM gather/scatter.  Design:
//   * axis-aligned fast path = SEPARABLE per-bin taps.  For an axis-aligned ROI the g_h x g_w
//     bilinear samples of a bin factor into per-axis weights: bin = sum_r sum_c wy[r]*wx[c]*F[r,c]
//     over the <= (g+2)^2 distinct pixels the bin touches -- each feature pixel is loaded ONCE
//     per bin instead of once per sample tap (4*g^2 loads), ~2x fewer loads at g=2..4.
//     The per-ROI weight tables (validity and border clamping folded in) are built once per
//     workgroup in LDS and shared by all channels.
//   * NHWC (torch.channels_last) kernel: lanes = channels, 4 channels (8/16 B) per lane, so a
//     wave reads 512 B - 1 KiB contiguous per tap; tap weights are wave-uniform (LDS broadcast).
//   * NCHW kernel: workgroup = one ROI x channel slab, thread = one output element; outputs
//     coalesced, taps served from L1/L2 (the ROI footprint of a slab fits L1).
//   * rotated / oversized-table fallback: direct per-sample kernel (4 taps per sample).
//   * backward: the same per-bin separable weights scatter dY with fp32 atomics; 16-bit
//     gradients accumulate in an fp32 workspace and are rounded once.
// fp32 accumulation everywhere; rois are always fp32 (SURVEY 7 "bf16" policy).
#include "roi_common.h"

namespace d2amd {

// ------------------------------------------------------------------------------------------
// DIRECT kernels (per-sample taps): rotated boxes, and the fallback when LDS tables overflow.
// One thread per output element; NHWC_ selects which index is fastest.
template <typename T, bool ROT, bool NHWC_>
__global__ __launch_bounds__(256) void roi_align_fwd_direct_kernel(
    const T* __restrict__ in, const float* __restrict__ rois, T* __restrict__ out, int C, int H, int W, int K,
    int PH, int PW, float scale, int sampling_ratio, int aligned, int* status) {
  const long total = (long)K * C * PH * PW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int pw, ph, c, k;
    if (NHWC_) {
      c = (int)(idx % C); pw = (int)((idx / C) % PW); ph = (int)((idx / C / PW) % PH); k = (int)(idx / C / PW / PH);
    } else {
      pw = (int)(idx % PW); ph = (int)((idx / PW) % PH); c = (int)((idx / PW / PH) % C); k = (int)(idx / PW / PH / C);
    }
    const RoiGeom g = roi_geom<ROT>(rois, k, scale, PH, PW, sampling_ratio, aligned);
    if (ROT && g.bad) {
      if (status) atomicOr(status, 1);
      out[idx] = from_f32<T>(0.f);
      continue;
    }
    const float count = (float)max(g.grid_h * g.grid_w, 1);
    const long plane = (long)H * W;
    const T* base = NHWC_ ? in + (long)g.batch * plane * C + c : in + ((long)g.batch * C + c) * plane;
    const long pstride = NHWC_ ? C : 1;
    float acc = 0.f;
    for (int iy = 0; iy < g.grid_h; iy++) {
      const float yy = sample_pos(g.start_h, ph, g.bin_h, iy, g.grid_h);
      AxisTap ty;
      if (!ROT) ty = axis_tap(yy, H);
      for (int ix = 0; ix < g.grid_w; ix++) {
        const float xx = sample_pos(g.start_w, pw, g.bin_w, ix, g.grid_w);
        AxisTap tx;
        if (ROT) {
          const float y = yy * g.cos_t - xx * g.sin_t + g.center_h;
          const float x = yy * g.sin_t + xx * g.cos_t + g.center_w;
          const bool valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
          ty = axis_tap(y, H);
          tx = axis_tap(x, W);
          if (!valid) { ty.wlo = ty.whi = 0.f; ty.lo = ty.hi = 0; tx.lo = tx.hi = 0; }
        } else {
          tx = axis_tap(xx, W);
        }
        const float v1 = to_f32(base[((long)ty.lo * W + tx.lo) * pstride]);
        const float v2 = to_f32(base[((long)ty.lo * W + tx.hi) * pstride]);
        const float v3 = to_f32(base[((long)ty.hi * W + tx.lo) * pstride]);
        const float v4 = to_f32(base[((long)ty.hi * W + tx.hi) * pstride]);
        acc += (ty.wlo * tx.wlo) * v1 + (ty.wlo * tx.whi) * v2 + (ty.whi * tx.wlo) * v3 + (ty.whi * tx.whi) * v4;
      }
    }
    out[idx] = from_f32<T>(acc / count);
  }
}

// backward: scatter dY * w / count with fp32 atomics (gin is fp32: grad_input itself or workspace)
template <typename T, bool ROT, bool NHWC_>
__global__ __launch_bounds__(256) void roi_align_bwd_direct_kernel(
    const T* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gin, int C, int H, int W,
    int K, int PH, int PW, float scale, int sampling_ratio, int aligned) {
  const long total = (long)K * C * PH * PW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int pw, ph, c, k;
    if (NHWC_) {
      c = (int)(idx % C); pw = (int)((idx / C) % PW); ph = (int)((idx / C / PW) % PH); k = (int)(idx / C / PW / PH);
    } else {
      pw = (int)(idx % PW); ph = (int)((idx / PW) % PH); c = (int)((idx / PW / PH) % C); k = (int)(idx / PW / PH / C);
    }
    const RoiGeom g = roi_geom<ROT>(rois, k, scale, PH, PW, sampling_ratio, aligned);
    if (ROT && g.bad) continue;
    const float count = (float)(g.grid_h * g.grid_w);
    const float go = to_f32(gout[idx]);
    const long plane = (long)H * W;
    float* base = NHWC_ ? gin + (long)g.batch * plane * C + c : gin + ((long)g.batch * C + c) * plane;
    const long pstride = NHWC_ ? C : 1;
    for (int iy = 0; iy < g.grid_h; iy++) {
      const float yy = sample_pos(g.start_h, ph, g.bin_h, iy, g.grid_h);
      AxisTap ty;
      if (!ROT) ty = axis_tap(yy, H);
      for (int ix = 0; ix < g.grid_w; ix++) {
        const float xx = sample_pos(g.start_w, pw, g.bin_w, ix, g.grid_w);
        AxisTap tx;
        bool valid;
        if (ROT) {
          const float y = yy * g.cos_t - xx * g.sin_t + g.center_h;
          const float x = yy * g.sin_t + xx * g.cos_t + g.center_w;
          valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
          ty = axis_tap(y, H);
          tx = axis_tap(x, W);
        } else {
          tx = axis_tap(xx, W);
          valid = ty.valid && tx.valid;
        }
        if (!valid) continue;
        atomicAdd(base + ((long)ty.lo * W + tx.lo) * pstride, go * (ty.wlo * tx.wlo) / count);
        atomicAdd(base + ((long)ty.lo * W + tx.hi) * pstride, go * (ty.wlo * tx.whi) / count);
        atomicAdd(base + ((long)ty.hi * W + tx.lo) * pstride, go * (ty.whi * tx.wlo) / count);
        atomicAdd(base + ((long)ty.hi * W + tx.hi) * pstride, go * (ty.whi * tx.whi) / count);
      }
    }
  }
}

constexpr int SEP_THREADS = 256;

// ---- separable backward (fp32 atomics), both layouts ---------------------------------------------
template <typename T, bool NHWC_>
__global__ __launch_bounds__(SEP_THREADS) void roi_align_bwd_sep_kernel(
    const T* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gin, int C, int H, int W,
    int PH, int PW, float scale, int sampling_ratio, int aligned, int cslab) {
  __shared__ SepShared S;
  const int k = blockIdx.x;
  sep_build<true>(S, rois, k, scale, PH, PW, sampling_ratio, aligned, H, W);
  const int c0 = blockIdx.y * cslab;
  const int nc = min(cslab, C - c0);
  if (!S.ok) {
    bwd_direct_range<T, NHWC_>(gout, rois, gin, k, c0, nc, C, H, W, PH, PW, scale, sampling_ratio, aligned);
    return;
  }
  const int bins = PH * PW;
  const long plane = (long)H * W;
  const float inv = S.inv_count;
  for (int e = threadIdx.x; e < nc * bins; e += SEP_THREADS) {
    int c, b;
    if (NHWC_) { b = e / nc; c = e - b * nc; } else { c = e / bins; b = e - c * bins; }
    const int ph = b / PW, pw = b - ph * PW;
    const float go = NHWC_ ? to_f32(gout[((long)k * bins + b) * C + c0 + c])
                           : to_f32(gout[((long)k * C + c0 + c) * bins + b]);
    const float gsc = go * inv;
    const int fy = S.firsty[ph], sy = S.spany[ph], fx = S.firstx[pw], sx = S.spanx[pw];
    const float* wy = S.wy + ph * SEP_SPAN;
    const float* wx = S.wx + pw * SEP_SPAN;
    float* base = NHWC_ ? gin + (long)S.batch * plane * C + c0 + c : gin + ((long)S.batch * C + c0 + c) * plane;
    const long pstride = NHWC_ ? C : 1;
    for (int j = 0; j < sy; j++) {
      const float gy = gsc * wy[j];
      float* row = base + ((long)(fy + j) * W + fx) * pstride;
      for (int i = 0; i < sx; i++) {
        const float w = wx[i];
        if (w != 0.f) atomicAdd(row + (long)i * pstride, gy * w);
      }
    }
  }
}

template <typename T>
__global__ void f32_to_T_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = from_f32<T>(src[i]);
}

static int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  return (int)(g > 256L * 32 ? 256L * 32 : (g < 1 ? 1 : g));
}

// channel slab so that (ROI, slab) workgroups fill the chip (>> 256 CUs) yet keep tables amortised
static int pick_cslab(int C, int K) {
  int slab = C;
  while (slab > 16 && (long)K * ((C + slab - 1) / slab) < 2048 && slab % 2 == 0) slab /= 2;
  return slab;
}

template <typename T>
static int fwd_impl(const void* input, const float* rois, void* output, int N, int C, int H, int W, int K, int PH,
                    int PW, float scale, int sr, int aligned, int layout, bool rotated, int* status,
                    hipStream_t s) {
  const T* in = (const T*)input;
  T* out = (T*)output;
  const bool nhwc = layout == D2AMD_NHWC;
  const long total = (long)K * C * PH * PW;
  const int gsz = grid_for(total, 256);
  if (rotated) {
    const bool timed = timing_begin("roi_align_rot_fwd", s);
    if (nhwc)
      hipLaunchKernelGGL((roi_align_fwd_direct_kernel<T, true, true>), dim3(gsz), dim3(256), 0, s, in, rois, out, C,
                         H, W, K, PH, PW, scale, sr, aligned, status);
    else
      hipLaunchKernelGGL((roi_align_fwd_direct_kernel<T, true, false>), dim3(gsz), dim3(256), 0, s, in, rois, out,
                         C, H, W, K, PH, PW, scale, sr, aligned, status);
    if (timed) timing_end("roi_align_rot_fwd", s);
    D2_LAUNCH_OK();
    return D2AMD_OK;
  }
  // axis-aligned with pooled size > SEP_MAXP (the fused kernels of roi_pool.hip serve the rest)
  if (nhwc)
    hipLaunchKernelGGL((roi_align_fwd_direct_kernel<T, false, true>), dim3(gsz), dim3(256), 0, s, in, rois, out, C, H,
                       W, K, PH, PW, scale, sr, aligned, nullptr);
  else
    hipLaunchKernelGGL((roi_align_fwd_direct_kernel<T, false, false>), dim3(gsz), dim3(256), 0, s, in, rois, out, C,
                       H, W, K, PH, PW, scale, sr, aligned, nullptr);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

template <typename T>
static int bwd_impl(const void* grad_output, const float* rois, void* grad_input, int N, int C, int H, int W, int K,
                    int PH, int PW, float scale, int sr, int aligned, int layout, bool rotated, void* workspace,
                    size_t workspace_bytes, hipStream_t s) {
  const T* gout = (const T*)grad_output;
  const bool nhwc = layout == D2AMD_NHWC;
  const long numel = (long)N * C * H * W;
  constexpr bool is32 = sizeof(T) == 4;
  const size_t need = is32 ? 0 : (size_t)numel * 4;  // fp32 accumulation buffer for 16-bit grads
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) {
    set_error("roi_align_backward: workspace too small (%zu < %zu)", workspace_bytes, need);
    return D2AMD_EWORKSPACE;
  }
  float* acc = is32 ? (float*)grad_input : (float*)workspace;
  const bool timed = rotated && timing_begin("roi_align_rot_bwd", s);  // (the whole op: zero fill, scatter, conversion)
  { const int zrc = zero_async(acc, (size_t)numel * 4, s); if (zrc) return zrc; }
  const long total = (long)K * C * PH * PW;
  if (total > 0) {
    const bool sep_ok = !rotated && PH <= SEP_MAXP && PW <= SEP_MAXP;
    if (!sep_ok) {
      const int gsz = grid_for(total, 256);
      if (rotated) {
        if (nhwc) hipLaunchKernelGGL((roi_align_bwd_direct_kernel<T, true, true>), dim3(gsz), dim3(256), 0, s, gout, rois, acc, C, H, W, K, PH, PW, scale, sr, aligned);
        else hipLaunchKernelGGL((roi_align_bwd_direct_kernel<T, true, false>), dim3(gsz), dim3(256), 0, s, gout, rois, acc, C, H, W, K, PH, PW, scale, sr, aligned);
      } else {
        if (nhwc) hipLaunchKernelGGL((roi_align_bwd_direct_kernel<T, false, true>), dim3(gsz), dim3(256), 0, s, gout, rois, acc, C, H, W, K, PH, PW, scale, sr, aligned);
        else hipLaunchKernelGGL((roi_align_bwd_direct_kernel<T, false, false>), dim3(gsz), dim3(256), 0, s, gout, rois, acc, C, H, W, K, PH, PW, scale, sr, aligned);
      }
    } else {
      const int cslab = pick_cslab(C, K);
      dim3 grid(K, cdiv(C, cslab));
      if (nhwc) hipLaunchKernelGGL((roi_align_bwd_sep_kernel<T, true>), grid, dim3(SEP_THREADS), 0, s, gout, rois, acc, C, H, W, PH, PW, scale, sr, aligned, cslab);
      else hipLaunchKernelGGL((roi_align_bwd_sep_kernel<T, false>), grid, dim3(SEP_THREADS), 0, s, gout, rois, acc, C, H, W, PH, PW, scale, sr, aligned, cslab);
    }
    D2_LAUNCH_OK();
  }
  if (!is32) {
    hipLaunchKernelGGL((f32_to_T_kernel<T>), dim3(grid_for(numel, 256)), dim3(256), 0, s, acc, (T*)grad_input, numel);
    D2_LAUNCH_OK();
  }
  if (timed) timing_end("roi_align_rot_bwd", s);
  return D2AMD_OK;
}

// ------------------------------------------------------------------------------------------
// fp64 entries (d2amd_roi_align_f64_*): the reference's ops are instantiated for double
// (ROIAlignRotated_cuda.cu: AT_DISPATCH_FLOATING_TYPES_AND_HALF, torchvision likewise) and its tests
// gradcheck them in double (tests/layers/test_roi_align_rotated.py:107-172).  Not a performance path:
// one thread per output element, every quantity double as ROIAlignRotated_cpu.cpp:27-129,201-416
// with T = double (axis-aligned: the pixel model of layers/roi_align.py:15-35), ROIs double (the
// reference casts them to the input dtype, roi_align.py:60), NCHW, double atomics in the backward.
struct RoiGeom64 {
  int batch, grid_w, grid_h;
  double start_w, start_h, bin_w, bin_h, center_w, center_h, cos_t, sin_t;
  bool bad;
};
template <bool ROT>
__device__ __forceinline__ RoiGeom64 roi_geom64(const double* __restrict__ rois, int k, double scale, int PH, int PW,
                                                int sr, int aligned) {
  RoiGeom64 g;
  g.bad = false;
  double roi_w, roi_h;
  if (ROT) {
    const double* r = rois + (long)k * 6;
    g.batch = (int)r[0];
    g.center_w = r[1] * scale - 0.5;
    g.center_h = r[2] * scale - 0.5;
    roi_w = r[3] * scale;
    roi_h = r[4] * scale;
    const double theta = r[5] * 3.14159265358979323846 / 180.0;
    g.cos_t = cos(theta);
    g.sin_t = sin(theta);
    g.bad = !(roi_w >= 0.0 && roi_h >= 0.0);
    g.start_h = -roi_h / 2.0;
    g.start_w = -roi_w / 2.0;
  } else {
    const double* r = rois + (long)k * 5;
    g.batch = (int)r[0];
    const double off = aligned ? 0.5 : 0.0;
    g.start_w = r[1] * scale - off;
    g.start_h = r[2] * scale - off;
    roi_w = (r[3] * scale - off) - g.start_w;
    roi_h = (r[4] * scale - off) - g.start_h;
    if (!aligned) { roi_w = fmax(roi_w, 1.0); roi_h = fmax(roi_h, 1.0); }
    g.center_w = g.center_h = 0.0;
    g.cos_t = 1.0;
    g.sin_t = 0.0;
  }
  g.bin_h = roi_h / (double)PH;
  g.bin_w = roi_w / (double)PW;
  g.grid_h = sr > 0 ? sr : (int)ceil(roi_h / (double)PH);
  g.grid_w = sr > 0 ? sr : (int)ceil(roi_w / (double)PW);
  return g;
}
struct Tap64 { int lo, hi; double wlo, whi; bool valid; };
__device__ __forceinline__ Tap64 axis_tap64(double y, int size) {
  Tap64 t;
  t.valid = !(y < -1.0 || y > (double)size);
  if (y < 0.0) y = 0.0;
  int lo = (int)y, hi;
  if (lo >= size - 1) { hi = lo = size - 1; y = (double)lo; } else { hi = lo + 1; }
  const double l = y - (double)lo;
  t.lo = lo; t.hi = hi; t.whi = l; t.wlo = 1.0 - l;
  return t;
}

// BWD = false: data = input [N,C,H,W], res = output [K,C,PH,PW];  BWD = true: data = grad_output, res = grad_input (zeroed)
template <bool ROT, bool BWD>
__global__ __launch_bounds__(256) void roi_align_f64_kernel(const double* __restrict__ data, const double* __restrict__ rois,
                                                           double* __restrict__ res, int C, int H, int W, int K, int PH,
                                                           int PW, double scale, int sr, int aligned, int* status) {
  const long total = (long)K * C * PH * PW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % PW), ph = (int)((idx / PW) % PH), c = (int)((idx / PW / PH) % C), k = (int)(idx / PW / PH / C);
    const RoiGeom64 g = roi_geom64<ROT>(rois, k, scale, PH, PW, sr, aligned);
    if (ROT && g.bad) {
      if (!BWD) { if (status) atomicOr(status, 1); res[idx] = 0.0; }
      continue;
    }
    const double count = (double)max(g.grid_h * g.grid_w, 1);
    const long plane = (long)H * W;
    const long base = ((long)g.batch * C + c) * plane;
    const double go = BWD ? data[idx] : 0.0;
    double acc = 0.0;
    for (int iy = 0; iy < g.grid_h; iy++) {
      const double yy = g.start_h + (double)ph * g.bin_h + ((double)iy + 0.5) * g.bin_h / (double)g.grid_h;
      for (int ix = 0; ix < g.grid_w; ix++) {
        const double xx = g.start_w + (double)pw * g.bin_w + ((double)ix + 0.5) * g.bin_w / (double)g.grid_w;
        const double y = ROT ? yy * g.cos_t - xx * g.sin_t + g.center_h : yy;
        const double x = ROT ? yy * g.sin_t + xx * g.cos_t + g.center_w : xx;
        const Tap64 ty = axis_tap64(y, H), tx = axis_tap64(x, W);
        if (!(ty.valid && tx.valid)) continue;
        const long o1 = base + (long)ty.lo * W + tx.lo, o2 = base + (long)ty.lo * W + tx.hi;
        const long o3 = base + (long)ty.hi * W + tx.lo, o4 = base + (long)ty.hi * W + tx.hi;
        if (BWD) {
          atomicAdd(res + o1, go * (ty.wlo * tx.wlo) / count);
          atomicAdd(res + o2, go * (ty.wlo * tx.whi) / count);
          atomicAdd(res + o3, go * (ty.whi * tx.wlo) / count);
          atomicAdd(res + o4, go * (ty.whi * tx.whi) / count);
        } else {
          acc += (ty.wlo * tx.wlo) * data[o1] + (ty.wlo * tx.whi) * data[o2] + (ty.whi * tx.wlo) * data[o3] +
                 (ty.whi * tx.whi) * data[o4];
        }
      }
    }
    if (!BWD) res[idx] = acc / count;
  }
}

static int check_common(const char* who, int N, int C, int H, int W, int K, int PH, int PW, int dtype, int layout) {
  D2_CHECK_ARG(N >= 0 && C >= 0 && H >= 0 && W >= 0 && K >= 0 && PH > 0 && PW > 0, "%s: bad shape", who);
  D2_CHECK_ARG(layout == D2AMD_NCHW || layout == D2AMD_NHWC, "%s: bad layout %d", who, layout);
  D2_CHECK_ARG(dtype == D2AMD_F32 || dtype == D2AMD_F16 || dtype == D2AMD_BF16, "%s: bad dtype %d", who, dtype);
  return D2AMD_OK;
}

static d2amd_pooler_params single_level(int N, int C, int H, int W, int ph, int pw, float scale, int sr, int aligned,
                                        int dtype, int layout) {
  d2amd_pooler_params p{};
  p.num_levels = 1; p.N = N; p.C = C; p.H[0] = H; p.W[0] = W; p.spatial_scale[0] = scale;
  p.pooled_h = ph; p.pooled_w = pw; p.sampling_ratio = sr; p.aligned = aligned; p.dtype = dtype; p.layout = layout;
  p.min_level = p.max_level = p.canonical_level = 0; p.canonical_box_size = 1.f;
  return p;
}

}  // namespace d2amd

using namespace d2amd;

extern "C" int d2amd_roi_align_forward(const void* input, const float* rois, void* output, int N, int C, int H,
                                       int W, int K, int pooled_h, int pooled_w, float spatial_scale,
                                       int sampling_ratio, int aligned, int dtype, int layout, void* stream) {
  int rc = check_common("roi_align_forward", N, C, H, W, K, pooled_h, pooled_w, dtype, layout);
  if (rc) return rc;
  if ((long)K * C == 0) return D2AMD_OK;
  D2_CHECK_ARG(input && rois && output, "roi_align_forward: null pointer");
  if (pooled_h <= SEP_MAXP && pooled_w <= SEP_MAXP) {  // single-level case of the fused pooler
    const d2amd_pooler_params p = single_level(N, C, H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned,
                                               dtype, layout);
    const void* lv[1] = {input};
    return d2amd_roi_pooler_forward(&p, lv, rois, output, K, stream);
  }
  return D2_DISPATCH_DTYPE(dtype, [&]() -> int {
    return fwd_impl<scalar_t>(input, rois, output, N, C, H, W, K, pooled_h, pooled_w, spatial_scale, sampling_ratio,
                              aligned, layout, false, nullptr, (hipStream_t)stream);
  });
}

extern "C" int d2amd_roi_align_backward(const void* grad_output, const float* rois, void* grad_input, int N, int C,
                                        int H, int W, int K, int pooled_h, int pooled_w, float spatial_scale,
                                        int sampling_ratio, int aligned, int dtype, int layout, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  int rc = check_common("roi_align_backward", N, C, H, W, K, pooled_h, pooled_w, dtype, layout);
  if (rc) return rc;
  if ((long)N * C * H * W == 0) return D2AMD_OK;
  D2_CHECK_ARG(grad_input && (K == 0 || (grad_output && rois)), "roi_align_backward: null pointer");
  if (layout == D2AMD_NHWC && pooled_h <= SEP_MAXP && pooled_w <= SEP_MAXP) {  // atomic-free tile gather
    const d2amd_pooler_params p = single_level(N, C, H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned,
                                               dtype, layout);
    void* lv[1] = {grad_input};
    return d2amd_roi_pooler_backward(&p, grad_output, rois, lv, K, workspace, workspace_bytes, stream);
  }
  return D2_DISPATCH_DTYPE(dtype, [&]() -> int {
    return bwd_impl<scalar_t>(grad_output, rois, grad_input, N, C, H, W, K, pooled_h, pooled_w, spatial_scale,
                              sampling_ratio, aligned, layout, false, workspace, workspace_bytes, (hipStream_t)stream);
  });
}

extern "C" int d2amd_roi_align_rotated_forward(const void* input, const float* rois, void* output, int N, int C,
                                               int H, int W, int K, int pooled_h, int pooled_w, float spatial_scale,
                                               int sampling_ratio, int dtype, int layout, int* status, void* stream) {
  int rc = check_common("roi_align_rotated_forward", N, C, H, W, K, pooled_h, pooled_w, dtype, layout);
  if (rc) return rc;
  if ((long)K * C == 0) return D2AMD_OK;
  D2_CHECK_ARG(input && rois && output, "roi_align_rotated_forward: null pointer");
  return D2_DISPATCH_DTYPE(dtype, [&]() -> int {
    return fwd_impl<scalar_t>(input, rois, output, N, C, H, W, K, pooled_h, pooled_w, spatial_scale, sampling_ratio, 1,
                              layout, true, status, (hipStream_t)stream);
  });
}

extern "C" int d2amd_roi_align_rotated_backward(const void* grad_output, const float* rois, void* grad_input, int N,
                                                int C, int H, int W, int K, int pooled_h, int pooled_w,
                                                float spatial_scale, int sampling_ratio, int dtype, int layout,
                                                void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_common("roi_align_rotated_backward", N, C, H, W, K, pooled_h, pooled_w, dtype, layout);
  if (rc) return rc;
  if ((long)N * C * H * W == 0) return D2AMD_OK;
  D2_CHECK_ARG(grad_input && (K == 0 || (grad_output && rois)), "roi_align_rotated_backward: null pointer");
  return D2_DISPATCH_DTYPE(dtype, [&]() -> int {
    return bwd_impl<scalar_t>(grad_output, rois, grad_input, N, C, H, W, K, pooled_h, pooled_w, spatial_scale,
                              sampling_ratio, 1, layout, true, workspace, workspace_bytes, (hipStream_t)stream);
  });
}

extern "C" int d2amd_roi_align_f64_forward(const double* input, const double* rois, double* output, int N, int C, int H,
                                           int W, int K, int pooled_h, int pooled_w, double spatial_scale,
                                           int sampling_ratio, int aligned, int rotated, int* status, void* stream) {
  int rc = check_common("roi_align_f64_forward", N, C, H, W, K, pooled_h, pooled_w, D2AMD_F32, D2AMD_NCHW);
  if (rc) return rc;
  const long total = (long)K * C * pooled_h * pooled_w;
  if (total == 0) return D2AMD_OK;
  D2_CHECK_ARG(input && rois && output, "roi_align_f64_forward: null pointer");
  const int gsz = grid_for(total, 256);
  hipStream_t s = (hipStream_t)stream;
  if (rotated)
    hipLaunchKernelGGL((roi_align_f64_kernel<true, false>), dim3(gsz), dim3(256), 0, s, input, rois, output, C, H, W, K,
                       pooled_h, pooled_w, spatial_scale, sampling_ratio, 1, status);
  else
    hipLaunchKernelGGL((roi_align_f64_kernel<false, false>), dim3(gsz), dim3(256), 0, s, input, rois, output, C, H, W, K,
                       pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, nullptr);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

extern "C" int d2amd_roi_align_f64_backward(const double* grad_output, const double* rois, double* grad_input, int N,
                                            int C, int H, int W, int K, int pooled_h, int pooled_w,
                                            double spatial_scale, int sampling_ratio, int aligned, int rotated,
                                            void* stream) {
  int rc = check_common("roi_align_f64_backward", N, C, H, W, K, pooled_h, pooled_w, D2AMD_F32, D2AMD_NCHW);
  if (rc) return rc;
  const long numel = (long)N * C * H * W;
  if (numel == 0) return D2AMD_OK;
  D2_CHECK_ARG(grad_input && (K == 0 || (grad_output && rois)), "roi_align_f64_backward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  { const int zrc = zero_async(grad_input, (size_t)numel * 8, s); if (zrc) return zrc; }
  const long total = (long)K * C * pooled_h * pooled_w;
  if (total == 0) return D2AMD_OK;
  const int gsz = grid_for(total, 256);
  if (rotated)
    hipLaunchKernelGGL((roi_align_f64_kernel<true, true>), dim3(gsz), dim3(256), 0, s, grad_output, rois, grad_input, C,
                       H, W, K, pooled_h, pooled_w, spatial_scale, sampling_ratio, 1, nullptr);
  else
    hipLaunchKernelGGL((roi_align_f64_kernel<false, true>), dim3(gsz), dim3(256), 0, s, grad_output, rois, grad_input, C,
                       H, W, K, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, nullptr);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}
