// Fused multi-level ROIAlignRotated pooler (NHWC): ROIPooler.forward with pooler_type "ROIAlignRotated"
// (detectron2/modeling/poolers.py:206-263: level assignment -> per level `nonzero` [host sync] -> ROIAlignRotated ->
// index_put_) in ONE launch per direction, like the axis-aligned pooler of roi_pool.hip.
//   forward   workgroup = ROI.  A rotated sampling grid is not separable, but it is the same for every channel: the
//             workgroup builds the ROI's TAP TABLE once in LDS -- per bin the 4 bilinear taps of each of its
//             grid_h x grid_w samples as {element offset, weight / count} (ROIAlignRotated_cpu.cpp:22-125: the
//             pre-calculated bilinear table, here per ROI instead of per call), taps on the same pixel MERGED (the
//             samples of a bin are ~1 px apart: 2-3 x fewer distinct pixels than taps) -- and a lane then owns 16 B of
//             channels and walks a bin's pixels in batches of 8 independent 16-B loads.  The per-element kernel of roi_align.hip
//             recomputes rotation, validity and the four weights for every (bin, channel, sample): 104 us for the RRPN
//             box head against 45 us for the axis-aligned pooler on the same shapes.
//   backward  the reference's scatter (ROIAlignRotated_cpu.cpp:312-416) with fp32 atomics, all levels in one launch:
//             the same tap table, lanes add w * dY into an fp32 image of the gradients (workspace), one convert pass.
//             Not deterministic in the order of the additions -- like the reference's.
// Level assignment: poolers.py:51-59 on RotatedBoxes.area() = w * h, fp32, operation for operation.
// ROIs of negative size: zero rows and bit 0 of *status (the reference asserts: ROIAlignRotated_cpu.cpp:236-238).
#include "roi_common.h"

namespace d2amd {

constexpr int ROT_MAX_LEVELS = 8;
constexpr int ROT_THREADS = 512;
constexpr int ROT_TAPTAB = 6400;  // taps per workgroup (51 KB): 49 bins x 32 samples x 4, or 196 bins x 8 samples x 4
constexpr int ROT_U = 8;          // independent loads in flight per lane
constexpr int ROT_BINCAP_MIN = 16; // distinct pixels per bin the table holds (merged taps): min(64, ROT_TAPTAB / bins), at least this

struct RotLevels {
  void* data[ROT_MAX_LEVELS];  // forward: feature maps (read); backward: fp32 gradient images (workspace)
  int H[ROT_MAX_LEVELS], W[ROT_MAX_LEVELS];
  float scale[ROT_MAX_LEVELS];
  int num_levels, N, C, PH, PW, sr;
  int min_level, max_level, canonical_level;
  float canonical_size;
  int* status;
};

typedef unsigned int rraw16 __attribute__((ext_vector_type(4)));
template <typename T> struct RV { static constexpr int N = 16 / (int)sizeof(T); };
__device__ __forceinline__ void runpack(const rraw16& r, float (&f)[4], float) {
  f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
}
__device__ __forceinline__ void runpack(const rraw16& r, float (&f)[8], bf16_t) {
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
  f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ void runpack(const rraw16& r, float (&f)[8], f16_t) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f[2 * i] = to_f32(f16_t{(uint16_t)(w[i] & 0xffffu)});
    f[2 * i + 1] = to_f32(f16_t{(uint16_t)(w[i] >> 16)});
  }
}
__device__ __forceinline__ rraw16 rpack(const float (&f)[4], float) {
  return rraw16{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
}
template <typename T>
__device__ __forceinline__ rraw16 rpack(const float (&f)[8], T) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = (uint32_t)from_f32<T>(f[2 * i]).v | ((uint32_t)from_f32<T>(f[2 * i + 1]).v << 16);
  return rraw16{w[0], w[1], w[2], w[3]};
}

// poolers.py:51-59 for a rotated box (cx, cy, w, h, angle): area = w * h (structures/rotated_boxes.py: area())
__device__ __forceinline__ int rot_assign_level(const float* __restrict__ box, const RotLevels& L) {
#pragma clang fp contract(off)
  if (L.num_levels == 1) return 0;
  const float area = box[2] * box[3];
  const float size = sqrtf(area);
  float lv = floorf((float)L.canonical_level + log2f(size / L.canonical_size + 1e-8f));
  if (!(lv == lv)) return -1;
  lv = fminf(fmaxf(lv, (float)L.min_level), (float)L.max_level);
  return (int)lv - L.min_level;
}

// The four taps of sample s = (iy, ix) of bin (ph, pw): ROIAlignRotated_cpu.cpp:47-118.  -> element offsets (pixel * C)
// and weights (x inv); a sample outside [-1, H] x [-1, W] has weight 0 on pixel 0.
__device__ __forceinline__ void rot_sample_taps(const RoiGeom& g, int ph, int pw, int iy, int ix, int H, int W, int C,
                                                float inv, uint32_t (&ofs)[4], float (&wt)[4]) {
  const float yy = sample_pos(g.start_h, ph, g.bin_h, iy, g.grid_h);
  const float xx = sample_pos(g.start_w, pw, g.bin_w, ix, g.grid_w);
  const float y = yy * g.cos_t - xx * g.sin_t + g.center_h;
  const float x = yy * g.sin_t + xx * g.cos_t + g.center_w;
  const bool valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  AxisTap ty = axis_tap(y, H), tx = axis_tap(x, W);
  if (!valid) { ty.wlo = ty.whi = 0.f; ty.lo = ty.hi = 0; tx.lo = tx.hi = 0; }
  ofs[0] = (uint32_t)(ty.lo * W + tx.lo) * (uint32_t)C; wt[0] = (ty.wlo * tx.wlo) * inv;
  ofs[1] = (uint32_t)(ty.lo * W + tx.hi) * (uint32_t)C; wt[1] = (ty.wlo * tx.whi) * inv;
  ofs[2] = (uint32_t)(ty.hi * W + tx.lo) * (uint32_t)C; wt[2] = (ty.whi * tx.wlo) * inv;
  ofs[3] = (uint32_t)(ty.hi * W + tx.hi) * (uint32_t)C; wt[3] = (ty.whi * tx.whi) * inv;
}

// BWD = false: out[k] = pooled features; BWD = true: the fp32 gradient images += w * gout[k]
template <typename T, int VEC, bool BWD>
__global__ __launch_bounds__(ROT_THREADS) void pool_rot_kernel(RotLevels L, const float* __restrict__ rois,
                                                               T* __restrict__ io) {
  __shared__ uint2 taptab[ROT_TAPTAB];
  __shared__ int bincnt[ROT_TAPTAB / ROT_BINCAP_MIN];
  __shared__ int s_over;
  const int k = blockIdx.x, tid = threadIdx.x;
  const float* roi = rois + (long)k * 6;
  const int lvl = __builtin_amdgcn_readfirstlane(rot_assign_level(roi + 1, L));
  const int C = L.C, PH = L.PH, PW = L.PW, bins = PH * PW, CG = C / VEC;
  T* iok = io + (long)k * bins * C;  // forward: the output rows; backward: dY of this ROI
  if (lvl < 0) {  // no level (NaN size): forward rows stay zero, as the reference's zero-initialised output
    if (!BWD)
      for (int e = tid; e < bins * C; e += ROT_THREADS) iok[e] = from_f32<T>(0.f);
    return;
  }
  const int H = L.H[lvl], W = L.W[lvl];
  const RoiGeom g = roi_geom<true>(rois, k, L.scale[lvl], PH, PW, L.sr, 1);
  if (g.bad) {
    if (tid == 0 && L.status) atomicOr(L.status, 1);
    if (!BWD)
      for (int e = tid; e < bins * C; e += ROT_THREADS) iok[e] = from_f32<T>(0.f);
    return;
  }
  const int ns = g.grid_h * g.grid_w;            // samples per bin (uniform over the ROI)
  const float inv = 1.f / (float)max(ns, 1);
  const int cg_shift = (CG & (CG - 1)) == 0 ? __builtin_ctz(CG) : -1;  // uniform
  const uint32_t rcp_pw = (65536u + (uint32_t)PW - 1u) / (uint32_t)PW;
  char* img = (char*)L.data[lvl];
  const int ROT_BINCAP = min(64, ROT_TAPTAB / bins);  // uniform
  int stride = ROT_BINCAP;                            // table slots per bin (uniform)
  bool table = ns <= 64 && ROT_BINCAP >= ROT_BINCAP_MIN && (long)H * W * C < (1l << 31);  // uniform (32-bit offsets)
  if (tid == 0) s_over = 0;
  __syncthreads();
  if (table) {
    // ---- tap table with the taps of a bin MERGED by pixel: neighbouring samples of a bin are ~1 px apart, so its
    // 4 ns taps hit 2-3 x fewer distinct pixels.  A wave builds one bin at a time: lane s computes the four taps of
    // sample s; the taps are then broadcast one by one and the lane that already holds the pixel adds the weight, or
    // the next free lane takes it (lane = entry: the distinct list lives in registers; a bin with more distinct pixels
    // than its share of the table sends the whole ROI down the per-sample path: s_over).
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t rcp_gw = (65536u + (uint32_t)g.grid_w - 1u) / (uint32_t)g.grid_w;
    // (forward: the merge is a serial walk of a bin's taps by one wave -- measured 92 -> 155 us for the RRPN box head,
    // more than the 2.6 x fewer loads give back -- so the forward keeps every tap when they fit: thread = (bin, sample);
    // the backward's atomics pay for it: 1.42 -> 0.55 ms)
    const bool plain = !BWD && (long)bins * 4 * ns <= ROT_TAPTAB;  // uniform
    if (plain) stride = 4 * ns;
    if (plain) {
      for (int idx = tid; idx < bins * ns; idx += ROT_THREADS) {
        const int b = idx / ns, s2 = idx - b * ns;
        const int ph = (int)(((uint32_t)b * rcp_pw) >> 16), pw = b - ph * PW;
        const int iy = (int)(((uint32_t)s2 * rcp_gw) >> 16), ix = s2 - iy * g.grid_w;
        uint32_t ofs[4];
        float wt[4];
        rot_sample_taps(g, ph, pw, iy, ix, H, W, C, inv, ofs, wt);
#pragma unroll
        for (int t = 0; t < 4; t++) taptab[b * stride + 4 * s2 + t] = uint2{ofs[t], __float_as_uint(wt[t])};
        if (s2 == 0) bincnt[b] = 4 * ns;
      }
    }
    for (int b = plain ? bins : wave; b < bins; b += ROT_THREADS / 64) {  // uniform per wave
      const int ph = (int)(((uint32_t)b * rcp_pw) >> 16), pw = b - ph * PW;
      uint32_t ofs[4] = {0u, 0u, 0u, 0u};
      float wt[4] = {0.f, 0.f, 0.f, 0.f};
      if (lane < ns) {
        const int iy = (int)(((uint32_t)lane * rcp_gw) >> 16), ix = lane - iy * g.grid_w;
        rot_sample_taps(g, ph, pw, iy, ix, H, W, C, inv, ofs, wt);
      }
      uint32_t my_ofs = 0u;
      float my_w = 0.f;
      int cnt = 0;
      for (int s = 0; s < ns; s++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const float w = __shfl(wt[t], s);
          if (w == 0.f) continue;  // uniform: an invalid sample, or a tap the border clamping folded away
          const uint32_t o = (uint32_t)__shfl((int)ofs[t], s);
          const unsigned long long hit = __ballot(lane < cnt && my_ofs == o);
          if (hit) {
            if ((hit >> lane) & 1ull) my_w += w;
          } else if (cnt < ROT_BINCAP) {
            if (lane == cnt) { my_ofs = o; my_w = w; }
            cnt++;
          } else if (lane == 0) {
            s_over = 1;  // more distinct pixels than the table holds (bins wider than ~5 px): this ROI takes the per-sample path
          }
        }
      }
      if (lane < cnt) taptab[b * ROT_BINCAP + lane] = uint2{my_ofs, __float_as_uint(my_w)};
      if (lane == 0) bincnt[b] = cnt;
    }
    __syncthreads();
    table = s_over == 0;
  }
  if (table) {
    if (!BWD) {
      const T* inb = (const T*)img + (long)g.batch * H * W * C;
      for (int e = tid; e < bins * CG; e += ROT_THREADS) {
        const int b = cg_shift >= 0 ? (e >> cg_shift) : e / CG, q = e - b * CG;
        const uint2* tab = taptab + b * stride;
        const int nt = bincnt[b];
        const T* base = inb + q * VEC;
        float acc[VEC];
#pragma unroll
        for (int c = 0; c < VEC; c++) acc[c] = 0.f;
        for (int t0 = 0; t0 < nt; t0 += ROT_U) {  // (the tail of a batch re-reads the last tap with weight 0)
          float w[ROT_U];
          if constexpr (VEC > 1) {
            rraw16 raw[ROT_U];
#pragma unroll
            for (int u = 0; u < ROT_U; u++) {
              const int t = t0 + u;
              const uint2 tp = tab[min(t, max(nt - 1, 0))];
              w[u] = t < nt ? __uint_as_float(tp.y) : 0.f;
              raw[u] = *reinterpret_cast<const rraw16*>(base + tp.x);
            }
#pragma unroll
            for (int u = 0; u < ROT_U; u++) {
              float f[VEC];
              runpack(raw[u], f, T{});
#pragma unroll
              for (int c = 0; c < VEC; c++) acc[c] += w[u] * f[c];
            }
          } else {
            float f[ROT_U];
#pragma unroll
            for (int u = 0; u < ROT_U; u++) {
              const int t = t0 + u;
              const uint2 tp = tab[min(t, max(nt - 1, 0))];
              w[u] = t < nt ? __uint_as_float(tp.y) : 0.f;
              f[u] = to_f32(base[tp.x]);
            }
#pragma unroll
            for (int u = 0; u < ROT_U; u++) acc[0] += w[u] * f[u];
          }
        }
        T* o = iok + (long)b * C + (long)q * VEC;
        if constexpr (VEC > 1) *reinterpret_cast<rraw16*>(o) = rpack(acc, T{});
        else o[0] = from_f32<T>(acc[0]);
      }
    } else {
      // lane = ONE channel: an atomic instruction of a wave covers 64 consecutive floats of one pixel (with 16 B of
      // channels per lane it touched 64 pixels' worth of 32-B pieces: 11.2 ms for the RRPN box head against 0.36)
      float* gb = (float*)img + (long)g.batch * H * W * C;
      for (int e = tid; e < bins * C; e += ROT_THREADS) {
        const int b = e / C, c = e - b * C;
        const uint2* tab = taptab + b * ROT_BINCAP;
        const int nt = bincnt[b];
        const float go = to_f32(iok[e]);
        float* base = gb + c;
        for (int t = 0; t < nt; t++) {
          const uint2 tp = tab[t];
          atomicAdd(base + tp.x, go * __uint_as_float(tp.y));
        }
      }
    }
    return;
  }
  // the table does not fit (sampling grids above 4 x 4 for 7 x 7 bins): per-sample taps, recomputed per channel group
  const long plane = (long)H * W;
  for (int e = tid; e < bins * CG; e += ROT_THREADS) {
    const int b = cg_shift >= 0 ? (e >> cg_shift) : e / CG, q = e - b * CG;
    const int ph = (int)(((uint32_t)b * rcp_pw) >> 16), pw = b - ph * PW;
    float acc[VEC], go[VEC];
#pragma unroll
    for (int c = 0; c < VEC; c++) acc[c] = 0.f;
    if (BWD) {
      if constexpr (VEC > 1) runpack(*reinterpret_cast<const rraw16*>(iok + (long)b * C + (long)q * VEC), go, T{});
      else go[0] = to_f32(iok[(long)b * C + q]);
    }
    for (int iy = 0; iy < g.grid_h; iy++)
      for (int ix = 0; ix < g.grid_w; ix++) {
        const float yy = sample_pos(g.start_h, ph, g.bin_h, iy, g.grid_h);
        const float xx = sample_pos(g.start_w, pw, g.bin_w, ix, g.grid_w);
        const float y = yy * g.cos_t - xx * g.sin_t + g.center_h;
        const float x = yy * g.sin_t + xx * g.cos_t + g.center_w;
        if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
        const AxisTap ty = axis_tap(y, H), tx = axis_tap(x, W);
        const long o4[4] = {(long)ty.lo * W + tx.lo, (long)ty.lo * W + tx.hi, (long)ty.hi * W + tx.lo, (long)ty.hi * W + tx.hi};
        const float w4[4] = {(ty.wlo * tx.wlo) * inv, (ty.wlo * tx.whi) * inv, (ty.whi * tx.wlo) * inv, (ty.whi * tx.whi) * inv};
#pragma unroll
        for (int t = 0; t < 4; t++) {
          if (!BWD) {
            const T* p = (const T*)img + ((long)g.batch * plane + o4[t]) * C + (long)q * VEC;
            float f[VEC];
            if constexpr (VEC > 1) runpack(*reinterpret_cast<const rraw16*>(p), f, T{});
            else f[0] = to_f32(p[0]);
#pragma unroll
            for (int c = 0; c < VEC; c++) acc[c] += w4[t] * f[c];
          } else if (w4[t] != 0.f) {
            float* p = (float*)img + ((long)g.batch * plane + o4[t]) * C + (long)q * VEC;
#pragma unroll
            for (int c = 0; c < VEC; c++) atomicAdd(p + c, go[c] * w4[t]);
          }
        }
      }
    if (!BWD) {
      T* o = iok + (long)b * C + (long)q * VEC;
      if constexpr (VEC > 1) *reinterpret_cast<rraw16*>(o) = rpack(acc, T{});
      else o[0] = from_f32<T>(acc[0]);
    }
  }
}

// fp32 gradient image -> grad_input in the I/O dtype (one pass over all levels: `off` = prefix of elements)
struct RotCvt {
  const float* src;
  void* dst[ROT_MAX_LEVELS];
  long end[ROT_MAX_LEVELS];
  int n;
};
template <typename T>
__global__ void pool_rot_cvt_kernel(RotCvt c) {
  const long total = c.end[c.n - 1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int l = 0;
#pragma unroll
    for (int q = 1; q < ROT_MAX_LEVELS; q++)
      if (q < c.n && i >= c.end[q - 1]) l = q;
    T* d = (T*)c.dst[0];
    long e0 = 0;
#pragma unroll
    for (int q = 1; q < ROT_MAX_LEVELS; q++)
      if (q == l) { d = (T*)c.dst[q]; e0 = c.end[q - 1]; }
    d[i - e0] = from_f32<T>(c.src[i]);
  }
}

static int rot_check(const d2amd_pooler_params* p, const char* who) {
  D2_CHECK_ARG(p != nullptr, "%s: null params", who);
  D2_CHECK_ARG(p->num_levels >= 1 && p->num_levels <= ROT_MAX_LEVELS, "%s: %d levels", who, p->num_levels);
  D2_CHECK_ARG(p->N >= 0 && p->C >= 1 && p->pooled_h >= 1 && p->pooled_w >= 1, "%s: bad shape", who);
  return D2AMD_OK;
}
static bool rot_supported(const d2amd_pooler_params* p) {
  return p->layout == D2AMD_NHWC && p->pooled_h * p->pooled_w <= 1024 &&
      (p->dtype == D2AMD_BF16 || p->dtype == D2AMD_F16 || p->dtype == D2AMD_F32);
}
static RotLevels rot_levels(const d2amd_pooler_params* p) {
  RotLevels L{};
  L.num_levels = p->num_levels; L.N = p->N; L.C = p->C; L.PH = p->pooled_h; L.PW = p->pooled_w; L.sr = p->sampling_ratio;
  L.min_level = p->min_level; L.max_level = p->max_level; L.canonical_level = p->canonical_level;
  L.canonical_size = p->canonical_box_size;
  for (int l = 0; l < p->num_levels; l++) { L.H[l] = p->H[l]; L.W[l] = p->W[l]; L.scale[l] = p->spatial_scale[l]; }
  return L;
}

template <typename T>
static int rot_forward(const d2amd_pooler_params* p, const void* const* inputs, const float* rois, void* output, int K,
                       int* status, hipStream_t st) {
  RotLevels L = rot_levels(p);
  constexpr int VEC = RV<T>::N;
  bool vec = p->C % VEC == 0 && (uintptr_t)output % 16 == 0;
  for (int l = 0; l < p->num_levels; l++) { L.data[l] = const_cast<void*>(inputs[l]); vec = vec && (uintptr_t)inputs[l] % 16 == 0; }
  L.status = status;
  if (vec) hipLaunchKernelGGL((pool_rot_kernel<T, VEC, false>), dim3(K), dim3(ROT_THREADS), 0, st, L, rois, (T*)output);
  else hipLaunchKernelGGL((pool_rot_kernel<T, 1, false>), dim3(K), dim3(ROT_THREADS), 0, st, L, rois, (T*)output);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

template <typename T>
static int rot_backward(const d2amd_pooler_params* p, const void* grad_output, const float* rois, void* const* grad_inputs,
                        int K, void* workspace, hipStream_t st) {
  RotLevels L = rot_levels(p);
  constexpr int VEC = RV<T>::N;
  const bool f32 = sizeof(T) == 4;
  RotCvt cv{};
  cv.n = p->num_levels;
  long off = 0;
  for (int l = 0; l < p->num_levels; l++) {
    const long n = (long)p->N * p->H[l] * p->W[l] * p->C;
    L.data[l] = f32 ? grad_inputs[l] : (void*)((float*)workspace + off);  // fp32: the atomics go to grad_input itself
    cv.dst[l] = grad_inputs[l];
    off += n;
    cv.end[l] = off;
    if (f32) { const int zrc = zero_async(grad_inputs[l], (size_t)n * 4, st); if (zrc) return zrc; }
  }
  if (!f32) { const int zrc = zero_async(workspace, (size_t)off * 4, st); if (zrc) return zrc; }
  if (K > 0) {
    const bool vec = p->C % VEC == 0 && (uintptr_t)grad_output % 16 == 0;
    const bool timed = timing_begin("pool_rot_bwd", st);
    if (vec) hipLaunchKernelGGL((pool_rot_kernel<T, VEC, true>), dim3(K), dim3(ROT_THREADS), 0, st, L, rois, (T*)const_cast<void*>(grad_output));
    else hipLaunchKernelGGL((pool_rot_kernel<T, 1, true>), dim3(K), dim3(ROT_THREADS), 0, st, L, rois, (T*)const_cast<void*>(grad_output));
    if (timed) timing_end("pool_rot_bwd", st);
    D2_LAUNCH_OK();
  }
  if (!f32 && off > 0) {
    cv.src = (const float*)workspace;
    const long blocks = cdiv(off, 256) > 8192 ? 8192 : cdiv(off, 256);
    hipLaunchKernelGGL((pool_rot_cvt_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, cv);
    D2_LAUNCH_OK();
  }
  return D2AMD_OK;
}

}  // namespace d2amd

using namespace d2amd;

extern "C" int d2amd_roi_pooler_rotated_supported(const d2amd_pooler_params* p) {
  if (rot_check(p, "roi_pooler_rotated_supported")) return 0;
  return rot_supported(p) ? 1 : 0;
}

extern "C" int d2amd_roi_pooler_rotated_forward(const d2amd_pooler_params* p, const void* const* inputs, const float* rois,
                                                void* output, int K, int* status, void* stream) {
  int rc = rot_check(p, "roi_pooler_rotated_forward");
  if (rc) return rc;
  if (!rot_supported(p)) { set_error("roi_pooler_rotated_forward: NHWC fp32 / bf16 / fp16 only"); return D2AMD_EUNSUPPORTED; }
  if (K == 0) return D2AMD_OK;
  D2_CHECK_ARG(K > 0 && inputs && rois && output, "roi_pooler_rotated_forward: null pointer");
  const bool timed = timing_begin("pool_rot_fwd", (hipStream_t)stream);
  rc = D2_DISPATCH_DTYPE(p->dtype, [&]() -> int { return rot_forward<scalar_t>(p, inputs, rois, output, K, status, (hipStream_t)stream); });
  if (timed) timing_end("pool_rot_fwd", (hipStream_t)stream);
  return rc;
}

extern "C" size_t d2amd_roi_pooler_rotated_backward_workspace_bytes(const d2amd_pooler_params* p) {
  if (rot_check(p, "roi_pooler_rotated_backward_workspace_bytes")) return 0;
  if (p->dtype == D2AMD_F32) return 256;
  size_t n = 0;
  for (int l = 0; l < p->num_levels; l++) n += (size_t)p->N * p->H[l] * p->W[l] * p->C;
  return n * 4 + 256;
}

extern "C" int d2amd_roi_pooler_rotated_backward(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                                                 void* const* grad_inputs, int K, void* workspace, size_t workspace_bytes,
                                                 void* stream) {
  int rc = rot_check(p, "roi_pooler_rotated_backward");
  if (rc) return rc;
  if (!rot_supported(p)) { set_error("roi_pooler_rotated_backward: NHWC fp32 / bf16 / fp16 only"); return D2AMD_EUNSUPPORTED; }
  D2_CHECK_ARG(grad_inputs && (K == 0 || (grad_output && rois)), "roi_pooler_rotated_backward: null pointer");
  D2_CHECK_ARG(workspace_bytes >= d2amd_roi_pooler_rotated_backward_workspace_bytes(p) && (p->dtype == D2AMD_F32 || workspace),
               "roi_pooler_rotated_backward: workspace too small");
  return D2_DISPATCH_DTYPE(p->dtype, [&]() -> int {
    return rot_backward<scalar_t>(p, grad_output, rois, grad_inputs, K, workspace, (hipStream_t)stream);
  });
}
