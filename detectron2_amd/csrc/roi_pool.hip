// Fused multi-level ROI pooler (ROIAlign / ROIAlignV2) for gfx950: ONE launch per direction for all
// FPN levels, level assignment included.
//   replaces  detectron2/modeling/poolers.py:206-263 (ROIPooler.forward: assign_boxes_to_levels ->
//             per-level nonzero / ROIAlign / index_put_) and the torchvision roi_align forward /
//             backward kernels it calls (detectron2/layers/roi_align.py:58-65).
// Roofline class: HBM (features read once, output written once; backward: dY read,
// grad_input written once).
//
// Forward (NHWC): workgroup = (ROI, chunk of bins).  The level is computed in-kernel (fp32, same
//   operation sequence as poolers.py:51-59), the separable per-bin tap tables are built once per
//   workgroup in LDS, and a lane owns 16 B of channels of one bin.  The <= (g+1)^2 taps of a bin
//   are walked as ONE flattened list in batches of 8 independent 16-B loads (the v0 kernel had a
//   single dependent load in flight per wave and was latency bound).
// Backward (NHWC): TILE GATHER -- no atomics, no fp32 staging buffer, no memset, deterministic.
//   workgroup = one 8x8-pixel tile of one image of one level (x 32 lanes of 16-B channel groups).
//   It (1) scans the ROI list and keeps, in order, the ROIs of its level/image whose footprint
//   touches the tile, (2) for each such ROI evaluates the per-axis weight matrices
//   Wy[8 rows][PH], Wx[8 cols][PW] (total bilinear weight the samples of a bin put on a pixel
//   row / column; the axis-aligned sampling grid is separable) and (3) accumulates
//   G[y,x,:] += sum_ph sum_pw Wy[y][ph] Wx[x][pw] dY[k,ph,pw,:] in registers, then writes every
//   pixel of grad_input exactly once in the I/O dtype.  tests/test_tile_gather_math.py checks this
//   formulation against the oracle's sample-by-sample scatter on the CPU.
#include "roi_common.h"

namespace d2amd {

constexpr int POOL_MAX_LEVELS = 8;
constexpr int POOL_THREADS = 256;

struct PoolLevels {
  const void* data[POOL_MAX_LEVELS];  // forward: feature maps; backward: grad_input (written)
  int H[POOL_MAX_LEVELS], W[POOL_MAX_LEVELS];
  float scale[POOL_MAX_LEVELS];
  int tile_base[POOL_MAX_LEVELS + 1];  // backward: prefix sum of tiles per level
  int num_levels, N, C, PH, PW, sr, aligned, K;
  int min_level, max_level, canonical_level;
  float canonical_size;
};

// detectron2/modeling/poolers.py:51-59 in fp32, operation for operation:
//   floor(canonical_level + log2(sqrt(area) / canonical_box_size + 1e-8)), clamped, - min_level.
// NaN sizes (negative area) map to -1 = "no level": forward rows stay zero, as in the reference
// where such a box matches no `level_assignments == level` mask.
__device__ __forceinline__ int assign_level(const float* __restrict__ box, const PoolLevels& L) {
#pragma clang fp contract(off)
  if (L.num_levels == 1) return 0;
  const float area = (box[2] - box[0]) * (box[3] - box[1]);
  const float size = sqrtf(area);
  float lv = floorf((float)L.canonical_level + log2f(size / L.canonical_size + 1e-8f));
  if (!(lv == lv)) return -1;
  lv = fminf(fmaxf(lv, (float)L.min_level), (float)L.max_level);
  return (int)lv - L.min_level;
}

// ---- 16-byte channel vectors ------------------------------------------------------------------
template <typename T> struct V16 { static constexpr int N = 16 / (int)sizeof(T); };

__device__ __forceinline__ void unpack16(const uint4& r, float (&f)[4], float) {
  f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
}
__device__ __forceinline__ void unpack16(const uint4& r, float (&f)[8], bf16_t) {
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
  f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ void unpack16(const uint4& r, float (&f)[8], f16_t) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f[2 * i] = to_f32(f16_t{(uint16_t)(w[i] & 0xffffu)});
    f[2 * i + 1] = to_f32(f16_t{(uint16_t)(w[i] >> 16)});
  }
}
__device__ __forceinline__ uint4 pack16(const float (&f)[4], float) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <typename T>
__device__ __forceinline__ uint4 pack16(const float (&f)[8], T) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = (uint32_t)from_f32<T>(f[2 * i]).v | ((uint32_t)from_f32<T>(f[2 * i + 1]).v << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// ------------------------------------------------------------------------------------------------
// FORWARD, NHWC.  grid = (K, nsplit); VEC = 16 B of channels per lane (or 1 for odd C / alignment)
template <typename T, int VEC>
__global__ __launch_bounds__(POOL_THREADS) void pool_fwd_nhwc_kernel(PoolLevels L, const float* __restrict__ rois,
                                                                    T* __restrict__ out, int nsplit) {
  __shared__ SepShared S;
  __shared__ int s_level;
  const int k = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) s_level = assign_level(rois + (long)k * 5 + 1, L);
  __syncthreads();
  const int lvl = __builtin_amdgcn_readfirstlane(s_level);
  const int C = L.C, PH = L.PH, PW = L.PW, bins = PH * PW;
  const int per = (bins + nsplit - 1) / nsplit;
  const int b_lo = blockIdx.y * per, b_hi = min(bins, b_lo + per);
  if (b_lo >= b_hi) return;
  const int CG = C / VEC;
  T* outk = out + (long)k * bins * C;
  if (lvl < 0) {  // reference: row of the zero-initialised output that no level fills
    for (int e = tid; e < (b_hi - b_lo) * C; e += POOL_THREADS) outk[(long)b_lo * C + e] = from_f32<T>(0.f);
    return;
  }
  const int H = L.H[lvl], W = L.W[lvl];
  const float scale = L.scale[lvl];
  const T* in = (const T*)L.data[lvl];
  sep_build<false>(S, rois, k, scale, PH, PW, L.sr, L.aligned, H, W);
  if (!S.ok) {  // a bin spans more than SEP_SPAN pixels: per-sample taps for this ROI
    fwd_direct_range<T, true>(in, rois, out, k, 0, C, C, H, W, PH, PW, scale, L.sr, L.aligned, b_lo, b_hi);
    return;
  }
  const T* inb = in + (long)S.batch * H * W * C;
  const float inv = S.inv_count;
  constexpr int U = 8;  // independent loads in flight per lane
  for (int e = tid; e < (b_hi - b_lo) * CG; e += POOL_THREADS) {
    const int bl = e / CG, q = e - bl * CG;
    const int b = b_lo + bl;
    const int ph = b / PW, pw = b - ph * PW;
    const int fy = S.firsty[ph], sy = S.spany[ph], fx = S.firstx[pw], sx = S.spanx[pw];
    const float* wy = S.wy + ph * SEP_SPAN;
    const float* wx = S.wx + pw * SEP_SPAN;
    const int nt = sy * sx;                        // <= SEP_SPAN^2 = 144 taps, row-major (j, i)
    const uint32_t rcp = sx > 0 ? (65536u + (uint32_t)sx - 1u) / (uint32_t)sx : 0u;  // t / sx == (t * rcp) >> 16 for t < 4096
    const T* base = inb + ((long)fy * W + fx) * C + (long)q * VEC;
    float acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; c++) acc[c] = 0.f;
    for (int t0 = 0; t0 < nt; t0 += U) {
      float w[U];
      if constexpr (VEC > 1) {
        uint4 raw[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int t = t0 + u, tt = min(t, nt - 1);
          const int j = (int)(((uint32_t)tt * rcp) >> 16), i = tt - j * sx;
          w[u] = t < nt ? wy[j] * wx[i] : 0.f;
          raw[u] = *reinterpret_cast<const uint4*>(base + ((long)j * W + i) * C);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          float f[VEC];
          unpack16(raw[u], f, T{});
#pragma unroll
          for (int c = 0; c < VEC; c++) acc[c] += w[u] * f[c];
        }
      } else {
        float f[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int t = t0 + u, tt = min(t, nt - 1);
          const int j = (int)(((uint32_t)tt * rcp) >> 16), i = tt - j * sx;
          w[u] = t < nt ? wy[j] * wx[i] : 0.f;
          f[u] = to_f32(base[((long)j * W + i) * C]);
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc[0] += w[u] * f[u];
      }
    }
#pragma unroll
    for (int c = 0; c < VEC; c++) acc[c] *= inv;
    T* o = outk + (long)b * C + (long)q * VEC;
    if constexpr (VEC > 1) {
      *reinterpret_cast<uint4*>(o) = pack16(acc, T{});
    } else {
      o[0] = from_f32<T>(acc[0]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// FORWARD, NCHW.  grid = (K, channel slabs); thread = one (channel, bin) of the slab, taps flattened
// and batched like the NHWC kernel (scalar loads: adjacent lanes = adjacent bins of one channel
// plane, whose footprint stays in L1).
template <typename T>
__global__ __launch_bounds__(POOL_THREADS) void pool_fwd_nchw_kernel(PoolLevels L, const float* __restrict__ rois,
                                                                    T* __restrict__ out, int cslab) {
  __shared__ SepShared S;
  __shared__ int s_level;
  const int k = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) s_level = assign_level(rois + (long)k * 5 + 1, L);
  __syncthreads();
  const int lvl = __builtin_amdgcn_readfirstlane(s_level);
  const int C = L.C, PH = L.PH, PW = L.PW, bins = PH * PW;
  const int c0 = blockIdx.y * cslab, nc = min(cslab, C - c0);
  T* outb = out + ((long)k * C + c0) * bins;
  if (lvl < 0) {
    for (int e = tid; e < nc * bins; e += POOL_THREADS) outb[e] = from_f32<T>(0.f);
    return;
  }
  const int H = L.H[lvl], W = L.W[lvl];
  const float scale = L.scale[lvl];
  const T* in = (const T*)L.data[lvl];
  sep_build<false>(S, rois, k, scale, PH, PW, L.sr, L.aligned, H, W);
  if (!S.ok) {
    fwd_direct_range<T, false>(in, rois, out, k, c0, nc, C, H, W, PH, PW, scale, L.sr, L.aligned);
    return;
  }
  const long plane = (long)H * W;
  const T* inb = in + ((long)S.batch * C + c0) * plane;
  const float inv = S.inv_count;
  constexpr int U = 8;
  for (int e = tid; e < nc * bins; e += POOL_THREADS) {
    const int c = e / bins, b = e - c * bins;
    const int ph = b / PW, pw = b - ph * PW;
    const int fy = S.firsty[ph], sy = S.spany[ph], fx = S.firstx[pw], sx = S.spanx[pw];
    const float* wy = S.wy + ph * SEP_SPAN;
    const float* wx = S.wx + pw * SEP_SPAN;
    const int nt = sy * sx;
    const uint32_t rcp = sx > 0 ? (65536u + (uint32_t)sx - 1u) / (uint32_t)sx : 0u;
    const T* base = inb + (long)c * plane + (long)fy * W + fx;
    float acc = 0.f;
    for (int t0 = 0; t0 < nt; t0 += U) {
      float w[U], f[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = t0 + u, tt = min(t, nt - 1);
        const int j = (int)(((uint32_t)tt * rcp) >> 16), i = tt - j * sx;
        w[u] = t < nt ? wy[j] * wx[i] : 0.f;
        f[u] = to_f32(base[(long)j * W + i]);
      }
#pragma unroll
      for (int u = 0; u < U; u++) acc += w[u] * f[u];
    }
    outb[e] = from_f32<T>(acc * inv);
  }
}

// ------------------------------------------------------------------------------------------------
// BACKWARD, NHWC: tile gather.
constexpr int TILE = 8;          // TILE x TILE pixels per workgroup
constexpr int LPP = 32;          // lanes (16-B channel groups) per pixel; 256 threads = 8 pixel columns
constexpr int LIST_CHUNK = 1024;  // ROIs scanned per list-building pass
constexpr int MAXP = SEP_MAXP;   // 32: one lane per bin along an axis

struct TileShared {
  int list[LIST_CHUNK];
  float Wy[TILE][MAXP], Wx[TILE][MAXP];  // Wx carries 1/count
  uint32_t ymask[TILE], xmask[TILE];
  int wave_cnt[POOL_THREADS / 64];
  int list_len;
};

// conservative footprint test: can ROI `g` put gradient on rows [y0, y0+TILE) x cols [x0, x0+TILE)?
__device__ __forceinline__ bool footprint_hits(const RoiGeom& g, int H, int W, int y0, int x0) {
  if (g.grid_h <= 0 || g.grid_w <= 0) return false;
  // samples lie strictly inside (start, start + roi); valid ones in [-1, size]; pixels touched are
  // floor(max(s, 0)) and +1, clamped to size - 1
  const float ylo = fmaxf(g.start_h, 0.f), yhi = g.start_h + g.roi_h;
  const float xlo = fmaxf(g.start_w, 0.f), xhi = g.start_w + g.roi_w;
  if (!(yhi >= -1.f && g.start_h <= (float)H && xhi >= -1.f && g.start_w <= (float)W)) return false;  // also NaN
  const int fy0 = (int)fminf(ylo, 1e9f), fy1 = min((int)fminf(fmaxf(yhi, 0.f), 1e9f) + 1, H - 1);
  const int fx0 = (int)fminf(xlo, 1e9f), fx1 = min((int)fminf(fmaxf(xhi, 0.f), 1e9f) + 1, W - 1);
  return fy1 >= y0 && fy0 < y0 + TILE && fx1 >= x0 && fx0 < x0 + TILE;
}

// total weight the `grid` samples of bin p put on pixel `pix` along one axis
__device__ __forceinline__ float axis_weight(float start, float bin, int grid, int p, int pix, int size) {
  float w = 0.f;
  for (int i = 0; i < grid; i++) {
    const AxisTap a = axis_tap(sample_pos(start, p, bin, i, grid), size);
    if (!a.valid) continue;
    if (a.lo == pix) w += a.wlo;
    if (a.hi == pix) w += a.whi;
  }
  return w;
}

template <typename T, int VEC>
__global__ __launch_bounds__(POOL_THREADS) void pool_bwd_nhwc_kernel(PoolLevels L, const float* __restrict__ rois,
                                                                    const T* __restrict__ gout, int nslab,
                                                                    int total_blocks) {
  __shared__ TileShared S;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so give each XCD one
  // contiguous run of tiles (neighbouring tiles share the dY rows of their ROIs in that XCD's L2)
  const int per_xcd = (total_blocks + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= total_blocks) return;
  const int slab = logical % nslab;
  const int tile = logical / nslab;
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < POOL_MAX_LEVELS; l++)
    if (l < L.num_levels && tile >= L.tile_base[l]) lvl = l;
  const int H = L.H[lvl], W = L.W[lvl];
  const float scale = L.scale[lvl];
  const int tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE;
  int tl = tile - L.tile_base[lvl];
  const int n = tl / (tiles_y * tiles_x);
  tl -= n * tiles_y * tiles_x;
  const int y0 = (tl / tiles_x) * TILE, x0 = (tl % tiles_x) * TILE;
  const int C = L.C, PH = L.PH, PW = L.PW, K = L.K;
  const int CG = C / VEC;
  const int col = tid >> 5;                 // pixel column of this thread inside the tile (0..7)
  const int cg = slab * LPP + (tid & 31);   // channel group
  const bool cg_ok = cg < CG;
  const long cofs = (long)cg * VEC;

  float acc[TILE][VEC];
#pragma unroll
  for (int i = 0; i < TILE; i++)
#pragma unroll
    for (int c = 0; c < VEC; c++) acc[i][c] = 0.f;

  for (int kbase = 0; kbase < K; kbase += LIST_CHUNK) {
    // ---- (1) ordered list of the ROIs of this chunk that touch the tile ----------------------
    __syncthreads();
    if (tid == 0) S.list_len = 0;
    __syncthreads();
    const int kend = min(K, kbase + LIST_CHUNK);
    for (int r0 = kbase; r0 < kend; r0 += POOL_THREADS) {
      const int r = r0 + tid;
      bool hit = false;
      if (r < kend) {
        const float* rr = rois + (long)r * 5;
        if ((int)rr[0] == n && assign_level(rr + 1, L) == lvl) {
          const RoiGeom g = roi_geom<false>(rois, r, scale, PH, PW, L.sr, L.aligned);
          hit = footprint_hits(g, H, W, y0, x0);
        }
      }
      const unsigned long long bal = __ballot(hit);
      if (lane == 0) S.wave_cnt[wid] = __builtin_popcountll(bal);
      __syncthreads();
      int off = S.list_len;
      for (int w = 0; w < wid; w++) off += S.wave_cnt[w];
      if (hit) S.list[off + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = r;
      __syncthreads();
      if (tid == 0) S.list_len += S.wave_cnt[0] + S.wave_cnt[1] + S.wave_cnt[2] + S.wave_cnt[3];
      __syncthreads();
    }
    const int nlist = S.list_len;
    // ---- (2)+(3) per ROI: axis weights for this tile, then the gather -------------------------
    for (int li = 0; li < nlist; li++) {
      const int k = S.list[li];
      const RoiGeom g = roi_geom<false>(rois, k, scale, PH, PW, L.sr, L.aligned);
      __syncthreads();  // previous ROI's weights are no longer read
      {
        const int r = tid >> 5, p = tid & 31;  // (tile row / column, bin)
        float wyv = 0.f, wxv = 0.f;
        if (p < PH && y0 + r < H) wyv = axis_weight(g.start_h, g.bin_h, g.grid_h, p, y0 + r, H);
        if (p < PW && x0 + r < W) wxv = axis_weight(g.start_w, g.bin_w, g.grid_w, p, x0 + r, W);
        const float inv = 1.f / (float)(g.grid_h * g.grid_w);
        S.Wy[r][p] = wyv;
        S.Wx[r][p] = wxv * inv;
        const unsigned long long by = __ballot(wyv != 0.f), bx = __ballot(wxv != 0.f);
        if (lane == 0) {
          S.ymask[2 * wid] = (uint32_t)by; S.ymask[2 * wid + 1] = (uint32_t)(by >> 32);
          S.xmask[2 * wid] = (uint32_t)bx; S.xmask[2 * wid + 1] = (uint32_t)(bx >> 32);
        }
      }
      __syncthreads();
      uint32_t yu = 0;
#pragma unroll
      for (int i = 0; i < TILE; i++) yu |= S.ymask[i];
      uint32_t xm = S.xmask[col];
      if (!cg_ok || yu == 0) xm = 0;
      const T* gk = gout + (long)k * PH * PW * C + cofs;
      while (xm) {
        const int pw = __builtin_ctz(xm);
        xm &= xm - 1;
        const float wxv = S.Wx[col][pw];
        uint32_t yb = yu;
        while (yb) {
          // up to 4 bins of this column per batch: 4 independent 16-B loads in flight
          int phs[4];
          bool ok[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            ok[u] = yb != 0;
            phs[u] = ok[u] ? __builtin_ctz(yb) : phs[0];
            if (ok[u]) yb &= yb - 1;
          }
          float f[4][VEC];
          if constexpr (VEC > 1) {
            uint4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; u++) raw[u] = *reinterpret_cast<const uint4*>(gk + ((long)phs[u] * PW + pw) * C);
#pragma unroll
            for (int u = 0; u < 4; u++) unpack16(raw[u], f[u], T{});
          } else {
#pragma unroll
            for (int u = 0; u < 4; u++) f[u][0] = to_f32(gk[((long)phs[u] * PW + pw) * C]);
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (!ok[u]) continue;
#pragma unroll
            for (int i = 0; i < TILE; i++) {
              const float wyi = S.Wy[i][phs[u]];
              if (wyi != 0.f) {  // rows are shared by the whole wave: uniform branch
                const float w = wyi * wxv;
#pragma unroll
                for (int c = 0; c < VEC; c++) acc[i][c] += w * f[u][c];
              }
            }
          }
        }
      }
    }
  }
  // ---- write the tile: every pixel of grad_input exactly once ---------------------------------
  if (cg_ok && x0 + col < W) {
    T* gi = (T*)L.data[lvl] + (((long)n * H + y0) * W + x0 + col) * C + cofs;
#pragma unroll
    for (int i = 0; i < TILE; i++) {
      if (y0 + i >= H) break;
      T* o = gi + (long)i * W * C;
      if constexpr (VEC > 1) {
        *reinterpret_cast<uint4*>(o) = pack16(acc[i], T{});
      } else {
        o[0] = from_f32<T>(acc[i][0]);
      }
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------
static int check_pooler(const d2amd_pooler_params* p, const char* who) {
  D2_CHECK_ARG(p != nullptr, "%s: null params", who);
  D2_CHECK_ARG(p->num_levels >= 1 && p->num_levels <= POOL_MAX_LEVELS, "%s: num_levels %d not in [1, %d]", who,
               p->num_levels, POOL_MAX_LEVELS);
  D2_CHECK_ARG(p->N >= 0 && p->C >= 0 && p->pooled_h > 0 && p->pooled_w > 0, "%s: bad shape", who);
  D2_CHECK_ARG(p->layout == D2AMD_NCHW || p->layout == D2AMD_NHWC, "%s: bad layout %d", who, p->layout);
  D2_CHECK_ARG(p->dtype == D2AMD_F32 || p->dtype == D2AMD_F16 || p->dtype == D2AMD_BF16, "%s: bad dtype %d", who,
               p->dtype);
  for (int l = 0; l < p->num_levels; l++)
    D2_CHECK_ARG(p->H[l] >= 0 && p->W[l] >= 0, "%s: bad level %d size", who, l);
  if (p->num_levels > 1) {
    D2_CHECK_ARG(p->max_level - p->min_level + 1 == p->num_levels && p->canonical_box_size > 0.f,
                 "%s: levels [%d, %d] do not match num_levels %d", who, p->min_level, p->max_level, p->num_levels);
  }
  return D2AMD_OK;
}

static PoolLevels make_levels(const d2amd_pooler_params* p, const void* const* data, int K) {
  PoolLevels L{};
  L.num_levels = p->num_levels; L.N = p->N; L.C = p->C; L.PH = p->pooled_h; L.PW = p->pooled_w;
  L.sr = p->sampling_ratio; L.aligned = p->aligned; L.K = K;
  L.min_level = p->min_level; L.max_level = p->max_level; L.canonical_level = p->canonical_level;
  L.canonical_size = p->canonical_box_size;
  int base = 0;
  for (int l = 0; l < p->num_levels; l++) {
    L.data[l] = data[l]; L.H[l] = p->H[l]; L.W[l] = p->W[l]; L.scale[l] = p->spatial_scale[l];
    L.tile_base[l] = base;
    base += cdiv(p->H[l], TILE) * cdiv(p->W[l], TILE) * p->N;
  }
  for (int l = p->num_levels; l <= POOL_MAX_LEVELS; l++) L.tile_base[l] = base;
  return L;
}

static bool all_aligned16(const void* const* data, int n, const void* extra) {
  uintptr_t a = (uintptr_t)extra;
  for (int l = 0; l < n; l++) a |= (uintptr_t)data[l];
  return (a & 15) == 0;
}

// can the fused kernels serve this configuration?  (else the caller loops over levels with the
// single-level entry points, which handle any pooled size through the direct kernels)
static bool pooler_fused_ok(const d2amd_pooler_params* p) { return p->pooled_h <= MAXP && p->pooled_w <= MAXP; }

template <typename T>
static int pool_fwd_impl(const d2amd_pooler_params* p, const void* const* inputs, const float* rois, void* output,
                         int K, hipStream_t s) {
  const PoolLevels L = make_levels(p, inputs, K);
  const int bins = p->pooled_h * p->pooled_w;
  constexpr int VEC = V16<T>::N;
  if (p->layout == D2AMD_NHWC) {
    const bool vec = (p->C % VEC == 0) && all_aligned16(inputs, p->num_levels, output);
    const int cg = vec ? p->C / VEC : p->C;
    const int passes = cdiv((long)bins * cg, POOL_THREADS);
    int nsplit = K > 0 ? 4096 / K : 1;
    nsplit = nsplit < 1 ? 1 : (nsplit > passes ? passes : nsplit);
    if (nsplit > bins) nsplit = bins;
    D2_CHECK_ARG(nsplit <= 65535, "roi_pooler_forward: internal split too large");
    dim3 grid(K, nsplit);
    if (vec)
      hipLaunchKernelGGL((pool_fwd_nhwc_kernel<T, VEC>), grid, dim3(POOL_THREADS), 0, s, L, rois, (T*)output, nsplit);
    else
      hipLaunchKernelGGL((pool_fwd_nhwc_kernel<T, 1>), grid, dim3(POOL_THREADS), 0, s, L, rois, (T*)output, nsplit);
  } else {
    int cslab = p->C;
    while (cslab > 16 && (long)K * cdiv(p->C, cslab) < 2048 && cslab % 2 == 0) cslab /= 2;
    dim3 grid(K, cdiv(p->C, cslab));
    D2_CHECK_ARG(grid.y <= 65535, "roi_pooler_forward: too many channel slabs");
    hipLaunchKernelGGL((pool_fwd_nchw_kernel<T>), grid, dim3(POOL_THREADS), 0, s, L, rois, (T*)output, cslab);
  }
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

template <typename T>
static int pool_bwd_nhwc_impl(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                              void* const* grad_inputs, int K, hipStream_t s) {
  const PoolLevels L = make_levels(p, (const void* const*)grad_inputs, K);
  constexpr int VEC = V16<T>::N;
  const bool vec = (p->C % VEC == 0) && all_aligned16((const void* const*)grad_inputs, p->num_levels, grad_output);
  const int cg = vec ? p->C / VEC : p->C;
  const int nslab = cdiv(cg, LPP);
  const long total = (long)L.tile_base[POOL_MAX_LEVELS] * nslab;
  if (total == 0) return D2AMD_OK;
  D2_CHECK_ARG(total < (1l << 30), "roi_pooler_backward: too many tiles");
  const int grid = (int)((total + 7) / 8) * 8;
  if (vec)
    hipLaunchKernelGGL((pool_bwd_nhwc_kernel<T, VEC>), dim3(grid), dim3(POOL_THREADS), 0, s, L, rois,
                       (const T*)grad_output, nslab, (int)total);
  else
    hipLaunchKernelGGL((pool_bwd_nhwc_kernel<T, 1>), dim3(grid), dim3(POOL_THREADS), 0, s, L, rois,
                       (const T*)grad_output, nslab, (int)total);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

}  // namespace d2amd

using namespace d2amd;

extern "C" int d2amd_roi_pooler_supported(const d2amd_pooler_params* p, int backward) {
  if (check_pooler(p, "roi_pooler_supported")) return 0;
  if (!pooler_fused_ok(p)) return 0;
  if (backward && p->layout != D2AMD_NHWC) return 0;
  return 1;
}

extern "C" int d2amd_roi_pooler_forward(const d2amd_pooler_params* p, const void* const* inputs, const float* rois,
                                        void* output, int K, void* stream) {
  int rc = check_pooler(p, "roi_pooler_forward");
  if (rc) return rc;
  D2_CHECK_ARG(K >= 0, "roi_pooler_forward: bad K");
  if ((long)K * p->C == 0) return D2AMD_OK;
  D2_CHECK_ARG(inputs && rois && output, "roi_pooler_forward: null pointer");
  if (!pooler_fused_ok(p)) {
    set_error("roi_pooler_forward: pooled size %dx%d exceeds the fused limit %d; use the per-level entry points",
              p->pooled_h, p->pooled_w, MAXP);
    return D2AMD_EUNSUPPORTED;
  }
  for (int l = 0; l < p->num_levels; l++)
    D2_CHECK_ARG(inputs[l] != nullptr || (long)p->N * p->H[l] * p->W[l] == 0, "roi_pooler_forward: null level %d", l);
  return D2_DISPATCH_DTYPE(p->dtype, [&]() -> int {
    return pool_fwd_impl<scalar_t>(p, inputs, rois, output, K, (hipStream_t)stream);
  });
}

extern "C" int d2amd_roi_pooler_backward(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                                         void* const* grad_inputs, int K, void* stream) {
  int rc = check_pooler(p, "roi_pooler_backward");
  if (rc) return rc;
  D2_CHECK_ARG(K >= 0, "roi_pooler_backward: bad K");
  D2_CHECK_ARG(grad_inputs && (K == 0 || (grad_output && rois)), "roi_pooler_backward: null pointer");
  if (!pooler_fused_ok(p) || p->layout != D2AMD_NHWC) {
    set_error("roi_pooler_backward: fused backward needs NHWC and pooled size <= %d; use the per-level entry points",
              MAXP);
    return D2AMD_EUNSUPPORTED;
  }
  if ((long)p->N * p->C == 0) return D2AMD_OK;
  return D2_DISPATCH_DTYPE(p->dtype, [&]() -> int {
    return pool_bwd_nhwc_impl<scalar_t>(p, grad_output, rois, grad_inputs, K, (hipStream_t)stream);
  });
}
