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
//   backward  a deterministic GATHER (r05; the reference scatters with atomics, ROIAlignRotated_cpu.cpp:312-416, and so did
//             rounds 3-4: 320 M fp32 atomics into an fp32 image + zero fill + convert, 0.65 ms for the RRPN box head):
//             (1) a workgroup per ROI builds the same merged tap table, stores it ROI-major and COUNTS the entries of
//             every pixel; (2) a workgroup per 1,024 pixels carves the pixels' list space out of one buffer (local scan +
//             one atomic per workgroup: where a list lies is irrelevant to the result); (3) the stored tables are
//             scattered into the lists as {dY row, weight}; (4) a wave per pixel rank-sorts its list by dY row (a row
//             occurs once per pixel: the order is total) and (5) a wave per pixel accumulates w * dY[row] in that order in
//             fp32 and writes the pixel's C channels once, in the I/O dtype: no fp32 image, no zero fill, no convert, the
//             same bits on every run.  ROIs whose table does not fit (bins wider than ~8 px) are flagged by (1) and take
//             the old atomic scatter into an fp32 image the gather adds (kernels that exit at once when no ROI is
//             flagged); `D2AMD_ROT_BWD_ATOMICS=1`: the atomic path for everything (the A/B and its test).
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
  int merge;  // forward: 1 = the merged tap table even when every tap fits (D2AMD_ROT_FWD_MERGE, profiling builds)
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

// The MERGED tap table of bin (ph, pw), built by one wave: the distinct pixels the bin's samples touch, each with the sum
// of the weights of the taps that land on it (neighbouring samples of a bin are ~1 px apart: 2-3 x fewer pixels than
// taps).  Lane s computes the four taps of sample s; then lane = CELL of the bounding box of all taps, and every cell
// walks the samples in order adding the taps that hit it -- ns steps of broadcasts instead of the 4 ns serial
// insert-or-add steps of the first version (110 -> 60 us for the table pass of the RRPN box head), the same sums in the
// same order.  emit(index, pixel, weight) is called by the lane that owns an entry; -> entries, or -1 beyond `cap`.
// (smp != nullptr: the samples' taps are staged in 2 KB of LDS per wave and broadcast from there -- the walk over the
// samples is 8 v_readlane + their SGPR hazards per sample otherwise: 36 of the table pass's 68 us)
template <typename Emit>
__device__ __forceinline__ int rot_bin_merge(const RoiGeom& g, int ph, int pw, int ns, int H, int W, float inv, int cap,
                                             int lane, Emit emit, uint4* smp = nullptr) {
  uint32_t ofs[4] = {0u, 0u, 0u, 0u};
  float wt[4] = {0.f, 0.f, 0.f, 0.f};
  int ylo = 0x7fffffff, yhi = -1, xlo = 0x7fffffff, xhi = -1;
  if (lane < ns) {
    const int iy = lane / g.grid_w, ix = lane - iy * g.grid_w;
    const float yy = sample_pos(g.start_h, ph, g.bin_h, iy, g.grid_h);
    const float xx = sample_pos(g.start_w, pw, g.bin_w, ix, g.grid_w);
    const float y = yy * g.cos_t - xx * g.sin_t + g.center_h;
    const float x = yy * g.sin_t + xx * g.cos_t + g.center_w;
    if (!(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W)) {
      const AxisTap ty = axis_tap(y, H), tx = axis_tap(x, W);
      ofs[0] = (uint32_t)(ty.lo * W + tx.lo); wt[0] = (ty.wlo * tx.wlo) * inv;
      ofs[1] = (uint32_t)(ty.lo * W + tx.hi); wt[1] = (ty.wlo * tx.whi) * inv;
      ofs[2] = (uint32_t)(ty.hi * W + tx.lo); wt[2] = (ty.whi * tx.wlo) * inv;
      ofs[3] = (uint32_t)(ty.hi * W + tx.hi); wt[3] = (ty.whi * tx.whi) * inv;
      ylo = ty.lo; yhi = ty.hi; xlo = tx.lo; xhi = tx.hi;
    }
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    ylo = min(ylo, __shfl_xor(ylo, d, 64)); yhi = max(yhi, __shfl_xor(yhi, d, 64));
    xlo = min(xlo, __shfl_xor(xlo, d, 64)); xhi = max(xhi, __shfl_xor(xhi, d, 64));
  }
  if (yhi < 0) return 0;  // no sample inside the map (uniform)
  const int bw = xhi - xlo + 1, ncell = (yhi - ylo + 1) * bw;
  if (smp) {
    __builtin_amdgcn_wave_barrier();  // (the previous bin's readers of this wave's slots are done: one wave, in order)
    if (lane < ns) {
      smp[2 * lane] = uint4{ofs[0], __float_as_uint(wt[0]), ofs[1], __float_as_uint(wt[1])};
      smp[2 * lane + 1] = uint4{ofs[2], __float_as_uint(wt[2]), ofs[3], __float_as_uint(wt[3])};
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
  }
  int cnt = 0;
  for (int c0 = 0; c0 < ncell; c0 += 64) {  // uniform
    const int cell = c0 + lane;
    const int cy = cell / bw, cx = cell - cy * bw;
    const uint32_t mypix = cell < ncell ? (uint32_t)((ylo + cy) * W + xlo + cx) : 0xffffffffu;
    float wsum = 0.f;
    if (smp) {
      for (int s2 = 0; s2 < ns; s2++) {
        const uint4 a = smp[2 * s2], b = smp[2 * s2 + 1];  // (uniform address: broadcast)
        wsum += mypix == a.x ? __uint_as_float(a.y) : 0.f;
        wsum += mypix == a.z ? __uint_as_float(a.w) : 0.f;
        wsum += mypix == b.x ? __uint_as_float(b.y) : 0.f;
        wsum += mypix == b.z ? __uint_as_float(b.w) : 0.f;
      }
    } else {
      for (int s2 = 0; s2 < ns; s2++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)ofs[t], s2);
          const float w = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(wt[t]), s2));
          wsum += mypix == o ? w : 0.f;
        }
      }
    }
    const unsigned long long nz = __ballot(wsum != 0.f);
    const int idx = cnt + (int)__popcll(nz & ((1ull << lane) - 1ull));
    if (((nz >> lane) & 1ull) && idx < cap) emit(idx, mypix, wsum);
    cnt += (int)__popcll(nz);
  }
  return cnt > cap ? -1 : cnt;
}

// BWD = false: out[k] = pooled features; BWD = true: the fp32 gradient images += w * gout[k]
template <typename T, int VEC, bool BWD>
__global__ __launch_bounds__(ROT_THREADS) void pool_rot_kernel(RotLevels L, const float* __restrict__ rois,
                                                               T* __restrict__ io,
                                                               const uint8_t* __restrict__ only = nullptr) {
  __shared__ uint2 taptab[ROT_TAPTAB];
  __shared__ int bincnt[ROT_TAPTAB / ROT_BINCAP_MIN];
  __shared__ int s_over;
  const int k = blockIdx.x, tid = threadIdx.x;
  if (only && !only[k]) return;  // (backward: the ROIs the gather's table pass flagged)
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
    // forward: every tap when they fit (thread = (bin, sample): no merge pass), else -- and L.merge = 1: the A/B -- the
    // merged table, one bin per wave at a time (rot_bin_merge)
    const bool plain = !BWD && !L.merge && (long)bins * 4 * ns <= ROT_TAPTAB;  // uniform
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
      uint2* tb = taptab + b * ROT_BINCAP;
      const int cnt = rot_bin_merge(g, ph, pw, ns, H, W, inv, ROT_BINCAP, lane, [&](int idx, uint32_t pix, float w) {
        tb[idx] = uint2{pix * (uint32_t)C, __float_as_uint(w)};
      });
      if (lane == 0) {
        bincnt[b] = cnt < 0 ? 0 : cnt;
        if (cnt < 0) s_over = 1;  // more distinct pixels than the table holds (bins wider than ~5 px): the per-sample path
      }
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

// ---------------------------------------------------------------------------------------------------------------
// Backward as a gather (see the header).  Pixel ids are global over the levels: pix_base[l] + (n * H_l + y) * W_l + x.
struct RotGather {
  long pix_base[ROT_MAX_LEVELS + 1];
  int* count;      // [P] entries per pixel                  (zeroed per call, with cursor / flags)
  int* cursor;     // [P] fill cursor
  int* flags;      // [0] any ROI flagged, [2] list space handed out, [3] pixels in longp (one 64-bit word)
  int2* cs;        // [P] {first slot of the pixel's list, its length}
  int* longp;      // pixels whose list is longer than ROT_GL entries (flags[3] of them): the sort kernel's work list
  uint8_t* over;   // [K] 1: the ROI takes the atomic path
  uint2* tab;      // [K][bins][cap] {pixel id, weight}: the ROIs' merged tap tables
  int* bincnt;     // [K][bins]
  uint2* list;     // [K * bins * cap] {dY row, weight} in fill order
  uint2* sorted;   // the same, every pixel's list ordered by dY row
  float* img;      // fp32 image of the flagged ROIs' gradients (touched only when flags[0])
  long img_elems;
  int P, cap;
};

// (1) merged tap table of ROI k -> tab / bincnt, count[pixel]++ per entry; the ROI is flagged when the table cannot hold it
template <typename T>
__global__ __launch_bounds__(ROT_THREADS) void rot_bwd_table_kernel(RotLevels L, RotGather G, const float* __restrict__ rois) {
  __shared__ int s_over;
  __shared__ uint4 s_smp[ROT_THREADS / 64][128];  // per wave: the taps of the current bin's samples
  const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = L.C, PH = L.PH, PW = L.PW, bins = PH * PW, cap = G.cap;
  const float* roi = rois + (long)k * 6;
  int* bc = G.bincnt + (long)k * bins;
  const int lvl = __builtin_amdgcn_readfirstlane(rot_assign_level(roi + 1, L));
  bool none = lvl < 0;  // no level (NaN size): no gradient, as the forward's zero rows
  RoiGeom g{};
  if (!none) {
    g = roi_geom<true>(rois, k, L.scale[lvl], PH, PW, L.sr, 1);
    if (g.bad) {
      if (tid == 0 && L.status) atomicOr(L.status, 1);
      none = true;
    }
  }
  if (none) {
    for (int b = tid; b < bins; b += ROT_THREADS) bc[b] = 0;
    if (tid == 0) G.over[k] = 0;
    return;
  }
  const int H = L.H[lvl], W = L.W[lvl];
  const int ns = g.grid_h * g.grid_w;
  const float inv = 1.f / (float)max(ns, 1);
  const bool table = ns <= 64;  // uniform
  if (tid == 0) s_over = table ? 0 : 1;
  __syncthreads();
  if (table) {
    const uint32_t rcp_pw = (65536u + (uint32_t)PW - 1u) / (uint32_t)PW;
    const uint32_t pbase = (uint32_t)(G.pix_base[lvl] + (long)g.batch * H * W);
    uint2* tk = G.tab + (long)k * bins * cap;
    // pass A: the tables
    for (int b = wave; b < bins; b += ROT_THREADS / 64) {  // uniform per wave
      const int ph = (int)(((uint32_t)b * rcp_pw) >> 16), pw = b - ph * PW;
      uint2* tb = tk + (long)b * cap;
      const int cnt = rot_bin_merge(g, ph, pw, ns, H, W, inv, cap, lane, [&](int idx, uint32_t pix, float w) {
        tb[idx] = uint2{pbase + pix, __float_as_uint(w)};
      }, s_smp[wave]);
      if (lane == 0) {
        bc[b] = cnt < 0 ? 0 : cnt;
        if (cnt < 0) s_over = 1;
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  const bool over = s_over != 0;  // uniform
  if (over) {
    for (int b = tid; b < bins; b += ROT_THREADS) bc[b] = 0;
    if (tid == 0) { G.over[k] = 1; atomicOr(G.flags, 1); }
    return;
  }
  if (tid == 0) G.over[k] = 0;
  // pass B: the pixels' counts (the table is complete and the ROI is known to fit)
  const uint2* tk = G.tab + (long)k * bins * cap;
  for (int e = tid; e < bins * cap; e += ROT_THREADS) {
    const int b = e / cap, t = e - b * cap;
    if (t < bc[b]) atomicAdd(G.count + tk[e].x, 1);
  }
}

// (2) list space: cs[pixel] = {first slot, length}.  Local exclusive scan per 1,024 pixels + one atomic per workgroup (the
// ranges' positions differ from run to run; nothing that is computed depends on them); pixels with more than ROT_GL entries
// are appended to the sort kernel's work list
constexpr int ROT_GL = 8;  // lanes per pixel in the gather = the longest list it orders itself
__global__ __launch_bounds__(1024) void rot_bwd_alloc_kernel(RotGather G) {
  __shared__ int wsum[16], wlong[16];
  __shared__ int s_base, s_lbase;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = blockIdx.x * 1024 + tid;
  const int c = p < G.P ? G.count[p] : 0;
  int x = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  const unsigned long long lm = __ballot(c > ROT_GL);
  if (lane == 63) { wsum[wave] = x; wlong[wave] = (int)__popcll(lm); }
  __syncthreads();
  if (tid == 0) {
    int run = 0, lrun = 0;
    for (int q = 0; q < 16; q++) {
      const int v = wsum[q], lv = wlong[q];
      wsum[q] = run; wlong[q] = lrun;
      run += v; lrun += lv;
    }
    // list space and work-list slots in ONE atomic per workgroup (same-address atomics serialise in the L2: one per wave
    // was 16 us of this kernel): flags[2] = slots handed out, flags[3] = pixels listed
    unsigned long long got = 0ull;
    if (run | lrun) got = atomicAdd(reinterpret_cast<unsigned long long*>(G.flags + 2), (unsigned long long)(uint32_t)run | ((unsigned long long)(uint32_t)lrun << 32));
    s_base = (int)(uint32_t)got; s_lbase = (int)(got >> 32);
  }
  __syncthreads();
  if (p < G.P) G.cs[p] = int2{s_base + wsum[wave] + x - c, c};
  if (c > ROT_GL) G.longp[s_lbase + wlong[wave] + (int)__popcll(lm & ((1ull << lane) - 1ull))] = p;
}

// (3) the stored tables -> the pixels' lists: 8 lanes per table row (a bin holds ~8 entries of its 64 slots)
__global__ __launch_bounds__(256) void rot_bwd_fill_kernel(RotGather G, long rows) {
  const long kb = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;  // = k * bins + b: the dY row
  const int sub = threadIdx.x & 7;
  if (kb >= rows) return;
  const int cnt = G.bincnt[kb];
  const uint2* tb = G.tab + kb * G.cap;
  for (int t = sub; t < cnt; t += 8) {
    const uint2 en = tb[t];
    const int slot = G.cs[en.x].x + atomicAdd(G.cursor + en.x, 1);
    G.list[slot] = uint2{(uint32_t)kb, en.y};
  }
}

// (4) the lists of more than ROT_GL entries ordered by dY row into `sorted` (a row occurs at most once per pixel: the taps
// of a bin are merged by pixel); shorter lists are ordered by the gather itself.  A wave per listed pixel; rank of an
// entry = number of smaller keys: up to 64 entries in registers, longer lists with the keys broadcast from LDS four at a
// time (lists beyond the LDS share are ranked against global memory: correct for any length, slow beyond a few thousand).
constexpr int ROT_SORT_WAVES = 4, ROT_SORT_LDS = 2048, ROT_SORT_GRID = 2048;
__global__ __launch_bounds__(64 * ROT_SORT_WAVES) void rot_bwd_sort_kernel(RotGather G) {
  __shared__ uint32_t keys[ROT_SORT_WAVES][ROT_SORT_LDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nlong = G.flags[3];
  uint32_t* kw = keys[wave];
  for (int it = blockIdx.x * ROT_SORT_WAVES + wave; it < nlong; it += ROT_SORT_GRID * ROT_SORT_WAVES) {  // uniform per wave
    const int p = G.longp[it];
    const int2 c = G.cs[p];
    const int n = c.y;
    const uint2* src = G.list + c.x;
    uint2* dst = G.sorted + c.x;
    if (n <= 64) {
      const uint2 my = lane < n ? src[lane] : uint2{0xffffffffu, 0u};
      int rank = 0;
      for (int j = 0; j < n; j++) rank += (uint32_t)__builtin_amdgcn_readlane((int)my.x, j) < my.x ? 1 : 0;
      if (lane < n) dst[rank] = my;
      continue;
    }
    const bool in_lds = n <= ROT_SORT_LDS;
    if (in_lds) {
      const int n4 = (n + 3) & ~3;
      for (int j = lane; j < n4; j += 64) kw[j] = j < n ? src[j].x : 0xffffffffu;
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes (no other wave reads them)
      __builtin_amdgcn_wave_barrier();
    }
    for (int base = 0; base < n; base += 64) {
      const int e = base + lane;
      const uint2 my = e < n ? src[e] : uint2{0xffffffffu, 0u};
      int rank = 0;
      if (in_lds) {
        const uint4* k4 = reinterpret_cast<const uint4*>(kw);
        for (int j = 0; j < (n + 3) / 4; j++) {
          const uint4 q = k4[j];
          rank += (q.x < my.x ? 1 : 0) + (q.y < my.x ? 1 : 0) + (q.z < my.x ? 1 : 0) + (q.w < my.x ? 1 : 0);
        }
      } else {
        for (int j = 0; j < n; j++) rank += src[j].x < my.x ? 1 : 0;
      }
      if (e < n) dst[rank] = my;
    }
    __builtin_amdgcn_wave_barrier();  // (the next pixel's keys overwrite this one's)
  }
}

// (5) grad_input[pixel][:] = sum over the pixel's list, ordered by dY row, of weight * dY[row][:]  (+ the flagged ROIs' fp32
// image).  EIGHT pixels per wave side by side, 8 lanes each (a kernel of one pixel per wave is a chain of three dependent
// loads -- {start, length}, entries, rows -- at an occupancy-bound 5 us per pixel: 113 us; pixel after pixel in one wave
// is worse): lane t of a pixel's group loads entry t of the current chunk of 8 (lists of up to 8 entries are ordered right
// here: rank among the group, one ds_permute), the rows' 16-B pieces are spread over the group's lanes (8 lanes x 16 B =
// 128 contiguous bytes per piece), 8 rows in flight per piece, up to 4 pieces per lane and pass over the list.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void rot_bwd_gather_kernel(RotLevels L, RotGather G, const T* __restrict__ dY) {
  constexpr int NQ = 4;  // pieces per lane and pass
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 3, sub = lane & 7;
  const long p = ((long)blockIdx.x * 4 + wave) * 8 + grp;
  const bool live = p < G.P;
  const int C = L.C;
  const bool add_img = G.flags[0] != 0;  // uniform
  const int2 cs = live ? G.cs[p] : int2{0, 0};
  const int n = cs.y;
  int lvl = 0;
#pragma unroll
  for (int q = 1; q < ROT_MAX_LEVELS; q++)
    if (q < L.num_levels && p >= G.pix_base[q]) lvl = q;
  T* out = (T*)L.data[0];
  long pb = G.pix_base[0];
#pragma unroll
  for (int q = 1; q < ROT_MAX_LEVELS; q++)
    if (q == lvl) { out = (T*)L.data[q]; pb = G.pix_base[q]; }
  out += (p - pb) * C;
  const float* im = G.img + p * C;
  const uint2* lst = (n > ROT_GL ? G.sorted : G.list) + cs.x;
  int nmax = n;  // the longest list of the wave's 8 pixels (uniform trip count)
#pragma unroll
  for (int d = 8; d < 64; d <<= 1) nmax = max(nmax, __shfl_xor(nmax, d, 64));
  const int pieces = (C + VEC - 1) / VEC;  // 16-B pieces (or single elements) of a row
  for (int q0 = 0; q0 < pieces; q0 += 8 * NQ) {  // uniform
    float acc[NQ][VEC];
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int c = 0; c < VEC; c++) acc[q][c] = 0.f;
    for (int base = 0; base < nmax; base += 8) {  // uniform
      uint2 my = (base + sub < n) ? lst[base + sub] : uint2{0xffffffffu, 0u};
      if (n <= ROT_GL && base == 0) {  // the whole list: rank among the group's keys, every entry to the lane of its rank
        int rank = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) rank += (uint32_t)__shfl((int)my.x, (lane & ~7) + u) < my.x ? 1 : 0;
        if (sub >= n) rank = sub;  // (lanes past the list keep their place: the permutation stays a bijection)
        const int dstl = ((lane & ~7) + rank) * 4;
        my.x = (uint32_t)__builtin_amdgcn_ds_permute(dstl, (int)my.x);
        my.y = (uint32_t)__builtin_amdgcn_ds_permute(dstl, (int)my.y);
      }
      uint32_t row[8];
      float w[8];
      unsigned hasm = 0u;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t r = (uint32_t)__shfl((int)my.x, (lane & ~7) + u);
        const float wi = __uint_as_float((uint32_t)__shfl((int)my.y, (lane & ~7) + u));
        const bool has = base + u < n;
        row[u] = has ? r : 0u;   // (a group past its list re-reads row 0 with weight 0 -- and takes its VALUE as 0 too:
        w[u] = has ? wi : 0.f;   // 0 x Inf of an overflowed 16-bit gradient is NaN, the reference's scatter never reads it)
        hasm |= has ? 1u << u : 0u;
      }
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const int piece = q0 + q * 8 + sub;
        if (piece < pieces) {
          rraw16 v[8];
          T v1[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const T* r = dY + (long)row[u] * C + (long)piece * VEC;
            if constexpr (VEC > 1) v[u] = *reinterpret_cast<const rraw16*>(r);
            else v1[u] = r[0];
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            if constexpr (VEC > 1) {
              float f[VEC];
              runpack((hasm >> u) & 1u ? v[u] : rraw16{0u, 0u, 0u, 0u}, f, T{});
#pragma unroll
              for (int c = 0; c < VEC; c++) acc[q][c] += w[u] * f[c];
            } else {
              acc[q][0] += (hasm >> u) & 1u ? w[u] * to_f32(v1[u]) : 0.f;
            }
          }
        }
      }
    }
    if (!live) continue;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int piece = q0 + q * 8 + sub;
      if (piece >= pieces) continue;
      const long c = (long)piece * VEC;
      if (add_img) {
#pragma unroll
        for (int e = 0; e < VEC; e++) acc[q][e] += im[c + e];
      }
      if constexpr (VEC > 1) *reinterpret_cast<rraw16*>(out + c) = rpack(acc[q], T{});
      else out[c] = from_f32<T>(acc[q][0]);
    }
  }
}

// the fp32 image of the flagged ROIs is zeroed only when there is one
__global__ __launch_bounds__(256) void rot_bwd_zero_img_kernel(RotGather G) {
  if (G.flags[0] == 0) return;
  float4* d = reinterpret_cast<float4*>(G.img);
  const long n4 = G.img_elems / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) d[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < G.img_elems; i += 256) G.img[i] = 0.f;
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
  { const char* e = d2_prof_env("D2AMD_ROT_FWD_MERGE"); L.merge = e && e[0] == '1'; }
  if (vec) hipLaunchKernelGGL((pool_rot_kernel<T, VEC, false>), dim3(K), dim3(ROT_THREADS), 0, st, L, rois, (T*)output);
  else hipLaunchKernelGGL((pool_rot_kernel<T, 1, false>), dim3(K), dim3(ROT_THREADS), 0, st, L, rois, (T*)output);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

// workspace layout of the gather backward (sizes in bytes, every part 256-B aligned)
struct RotWs {
  size_t count, cursor, flags, cs, longp, over, bincnt, tab, list, sorted, img, total;
  long P, img_elems;
  int cap;
};
static RotWs rot_ws(const d2amd_pooler_params* p, int K) {
  RotWs w{};
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  long P = 0;
  for (int l = 0; l < p->num_levels; l++) P += (long)p->N * p->H[l] * p->W[l];
  w.P = P; w.img_elems = P * p->C; w.cap = 64;
  const size_t bins = (size_t)p->pooled_h * p->pooled_w, ent = (size_t)(K > 0 ? K : 0) * bins * w.cap;
  size_t o = 0;
  w.count = o; o += al((size_t)P * 4);
  w.cursor = o; o += al((size_t)P * 4);
  w.flags = o; o += 256;                      // (count, cursor, flags: one zero fill)
  w.cs = o; o += al((size_t)P * 8);
  w.longp = o; o += al((size_t)P * 4);
  w.over = o; o += al((size_t)(K > 0 ? K : 0));
  w.bincnt = o; o += al((size_t)(K > 0 ? K : 0) * bins * 4);
  w.tab = o; o += al(ent * 8);
  w.list = o; o += al(ent * 8);
  w.sorted = o; o += al(ent * 8);
  w.img = o; o += al((size_t)w.img_elems * 4);
  w.total = o;
  return w;
}
static bool rot_bwd_atomics() {
  const char* e = getenv("D2AMD_ROT_BWD_ATOMICS");  // (read per call: the test toggles it)
  return e && e[0] == '1';
}

template <typename T>
static int rot_backward_atomic(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                               void* const* grad_inputs, int K, void* workspace, hipStream_t st) {
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
    if (vec) hipLaunchKernelGGL((pool_rot_kernel<T, VEC, true>), dim3(K), dim3(ROT_THREADS), 0, st, L, rois, (T*)const_cast<void*>(grad_output), (const uint8_t*)nullptr);
    else hipLaunchKernelGGL((pool_rot_kernel<T, 1, true>), dim3(K), dim3(ROT_THREADS), 0, st, L, rois, (T*)const_cast<void*>(grad_output), (const uint8_t*)nullptr);
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

template <typename T>
static int rot_backward(const d2amd_pooler_params* p, const void* grad_output, const float* rois, void* const* grad_inputs,
                        int K, void* workspace, hipStream_t st) {
  const RotWs w = rot_ws(p, K);
  const bool aligned = [&] {
    bool ok = (uintptr_t)grad_output % 16 == 0;
    for (int l = 0; l < p->num_levels; l++) ok = ok && (uintptr_t)grad_inputs[l] % 16 == 0;
    return ok;
  }();
  if (rot_bwd_atomics() || !aligned || w.P >= (1l << 31) || (long)K * p->pooled_h * p->pooled_w * w.cap >= (1l << 31))
    return rot_backward_atomic<T>(p, grad_output, rois, grad_inputs, K, workspace, st);
  RotLevels L = rot_levels(p);
  RotGather G{};
  char* ws = (char*)workspace;
  long pb = 0;
  for (int l = 0; l < p->num_levels; l++) {
    L.data[l] = grad_inputs[l];
    G.pix_base[l] = pb;
    pb += (long)p->N * p->H[l] * p->W[l];
  }
  for (int l = p->num_levels; l <= ROT_MAX_LEVELS; l++) G.pix_base[l] = pb;
  if (w.P == 0) return D2AMD_OK;
  G.count = (int*)(ws + w.count); G.cursor = (int*)(ws + w.cursor); G.flags = (int*)(ws + w.flags);
  G.cs = (int2*)(ws + w.cs); G.longp = (int*)(ws + w.longp); G.over = (uint8_t*)(ws + w.over); G.bincnt = (int*)(ws + w.bincnt);
  G.tab = (uint2*)(ws + w.tab); G.list = (uint2*)(ws + w.list); G.sorted = (uint2*)(ws + w.sorted);
  G.img = (float*)(ws + w.img); G.img_elems = w.img_elems; G.P = (int)w.P; G.cap = w.cap;
  const int zrc = zero_async(ws + w.count, w.cs - w.count, st);
  if (zrc) return zrc;
  const int bins = p->pooled_h * p->pooled_w;
  T* dY = (T*)const_cast<void*>(grad_output);
  if (K > 0) {
    hipLaunchKernelGGL((rot_bwd_table_kernel<T>), dim3(K), dim3(ROT_THREADS), 0, st, L, G, rois);
    // the flagged ROIs (tables that do not fit): the atomic scatter into the fp32 image -- both kernels return at once
    // when no ROI is flagged
    hipLaunchKernelGGL(rot_bwd_zero_img_kernel, dim3(2048), dim3(256), 0, st, G);
    RotLevels La = L;
    long off = 0;
    for (int l = 0; l < p->num_levels; l++) { La.data[l] = (void*)(G.img + off); off += (long)p->N * p->H[l] * p->W[l] * p->C; }
    La.status = nullptr;  // (reported by the table pass)
    constexpr int VEC = RV<T>::N;
    if (p->C % VEC == 0) hipLaunchKernelGGL((pool_rot_kernel<T, VEC, true>), dim3(K), dim3(ROT_THREADS), 0, st, La, rois, dY, (const uint8_t*)G.over);
    else hipLaunchKernelGGL((pool_rot_kernel<T, 1, true>), dim3(K), dim3(ROT_THREADS), 0, st, La, rois, dY, (const uint8_t*)G.over);
  }
  hipLaunchKernelGGL(rot_bwd_alloc_kernel, dim3(cdiv(w.P, 1024)), dim3(1024), 0, st, G);
  if (K > 0) {
    const long rows = (long)K * bins;
    hipLaunchKernelGGL(rot_bwd_fill_kernel, dim3(cdiv(rows * 8, 256)), dim3(256), 0, st, G, rows);
    hipLaunchKernelGGL(rot_bwd_sort_kernel, dim3(ROT_SORT_GRID), dim3(64 * ROT_SORT_WAVES), 0, st, G);
  }
  {
    constexpr int VEC = RV<T>::N;
    if (p->C % VEC == 0) hipLaunchKernelGGL((rot_bwd_gather_kernel<T, VEC>), dim3(cdiv(w.P, 32)), dim3(256), 0, st, L, G, (const T*)dY);
    else hipLaunchKernelGGL((rot_bwd_gather_kernel<T, 1>), dim3(cdiv(w.P, 32)), dim3(256), 0, st, L, G, (const T*)dY);
  }
  D2_LAUNCH_OK();
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

extern "C" size_t d2amd_roi_pooler_rotated_backward_workspace_bytes(const d2amd_pooler_params* p, int K) {
  if (rot_check(p, "roi_pooler_rotated_backward_workspace_bytes")) return 0;
  return rot_ws(p, K).total + 256;
}

extern "C" int d2amd_roi_pooler_rotated_backward(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                                                 void* const* grad_inputs, int K, void* workspace, size_t workspace_bytes,
                                                 void* stream) {
  int rc = rot_check(p, "roi_pooler_rotated_backward");
  if (rc) return rc;
  if (!rot_supported(p)) { set_error("roi_pooler_rotated_backward: NHWC fp32 / bf16 / fp16 only"); return D2AMD_EUNSUPPORTED; }
  D2_CHECK_ARG(K >= 0 && grad_inputs && (K == 0 || (grad_output && rois)), "roi_pooler_rotated_backward: null pointer");
  D2_CHECK_ARG(workspace && workspace_bytes >= d2amd_roi_pooler_rotated_backward_workspace_bytes(p, K),
               "roi_pooler_rotated_backward: workspace too small");
  const bool timed = timing_begin("pool_rot_bwd", (hipStream_t)stream);
  rc = D2_DISPATCH_DTYPE(p->dtype, [&]() -> int {
    return rot_backward<scalar_t>(p, grad_output, rois, grad_inputs, K, workspace, (hipStream_t)stream);
  });
  if (timed) timing_end("pool_rot_bwd", (hipStream_t)stream);
  return rc;
}
