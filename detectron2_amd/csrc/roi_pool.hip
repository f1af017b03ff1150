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
#include <stdlib.h>

#include "roi_common.h"

namespace d2amd {

constexpr int POOL_MAX_LEVELS = 8;
constexpr int POOL_THREADS = 256;
#ifndef D2AMD_FWD_U
#define D2AMD_FWD_U 8
#endif
constexpr int FWD_U = D2AMD_FWD_U;

struct PoolLevels {
  const void* data[POOL_MAX_LEVELS];  // forward: feature maps; backward: grad_input (written)
  int H[POOL_MAX_LEVELS], W[POOL_MAX_LEVELS];
  float scale[POOL_MAX_LEVELS];
  int tile_base[POOL_MAX_LEVELS + 1];  // backward: prefix sum of tiles per level
  int num_levels, N, C, PH, PW, sr, aligned, K;
  int min_level, max_level, canonical_level;
  float canonical_size;
  unsigned long long* dbg;  // profiling only (D2AMD_PROFILE builds): cycle stamps of one workgroup
  int dbg_block;
  int ablate;  // profiling only (D2AMD_ABLATE): bit0 skip gather, bit1 skip weights, bit2 skip list scan; MFMA tile
               // gather: bit3 no pairing, bit4 no loads of dY, bit5 no weight images, bit6 no items (empty lists), bit7 no stores
  const int4* tile_geo;   // backward: per tile {level | image << 8, y0 | x0 << 16, H | W << 16, -} (tile_lists_kernel)
  const int* tile_cnt;    // backward: per tile, number of ROIs that touch it (nullptr: tiles scan the records)
  const void* tile_list;  // backward: [tile][TILE_CAP] TileEntry in ROI order (valid when tile_cnt[tile] <= TILE_CAP)
  unsigned long long* wgstamps;  // profiling only (D2AMD_POOL_STAMPS): per workgroup {start, lists done, loop done, end, #ROIs}
  int tab_off;       // forward: 1 = no 32-bit tap table (a level holds >= 2^32 elements per image)
  const int2* queue;  // backward: per-XCD work queues of this launch (tile_lists_kernel), [8][qcap] slots {tile | #ROIs
                      // << 24, part | parts << 8 | scratch slot << 16}; x = -1: unused; nullptr: static order
  int qcap;
  float* part_scratch;  // split tiles (tile_lists_kernel): fp32 partial accumulators, [slot][16][2][512] floats per slab ...
  int* part_tickets;    // ... and one arrival counter per split tile (at its first scratch slot)
  int* qctr;         // backward: the queues' counters (TileQueues::mem); non-null: persistent workgroups FETCH their tiles
                     // (take counter per XCD, then the other XCDs' queues) instead of serving slot blockIdx >> 3
  int accumulate;    // backward: 1 = grad_input already holds a gradient (another pooler's): add to it, skip empty tiles
  const int* perm;   // forward: ROI processing order (roi_order_kernel), nullptr: workgroup b pools ROI b
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
// native 4 x u32 vector: a first-class SSA value (arrays of HIP's raw16 struct that live across loop
// iterations were left in scratch memory by hipcc, which serialised the prefetch)
typedef unsigned int raw16 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void unpack16(const raw16& r, float (&f)[4], float) {
  f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
}
__device__ __forceinline__ void unpack16(const raw16& r, float (&f)[8], bf16_t) {
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
  f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ void unpack16(const raw16& r, float (&f)[8], f16_t) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f[2 * i] = to_f32(f16_t{(uint16_t)(w[i] & 0xffffu)});
    f[2 * i + 1] = to_f32(f16_t{(uint16_t)(w[i] >> 16)});
  }
}
__device__ __forceinline__ raw16 pack16(const float (&f)[4], float) {
  return raw16{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
}
// (bf16 by bit arithmetic, what from_f32<bf16_t> was before it went to the hardware converter: the LDS-staged fallback
// kernel below sits at its register cap and spills 3 VGPRs with the other instruction mix)
__device__ __forceinline__ uint32_t bf16_rne_bits(float x) {
  uint32_t u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ raw16 pack16_staged(const float (&f)[8], bf16_t) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = bf16_rne_bits(f[2 * i]) | (bf16_rne_bits(f[2 * i + 1]) << 16);
  return raw16{w[0], w[1], w[2], w[3]};
}
template <typename T>
__device__ __forceinline__ raw16 pack16(const float (&f)[8], T);
__device__ __forceinline__ raw16 pack16(const float (&f)[4], float);
template <typename T, int N>
__device__ __forceinline__ raw16 pack16_staged(const float (&f)[N], T t) { return pack16(f, t); }
template <typename T>
__device__ __forceinline__ raw16 pack16(const float (&f)[8], T) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = (uint32_t)from_f32<T>(f[2 * i]).v | ((uint32_t)from_f32<T>(f[2 * i + 1]).v << 16);
  return raw16{w[0], w[1], w[2], w[3]};
}

// Which ROI workgroup column b of the forward pools.  Workgroups are dealt round-robin to the 8 XCDs, each with its own
// L2: with the ROIs in list order, neighbours in the feature map are pooled on different XCDs at different times and
// every L2 fetches the same feature lines again (PMC: 1.6 x the features from HBM / Infinity Cache for the box head).
// roi_order_kernel sorts the ROIs by (level, image, Morton code of the centre's 8-px tile); XCD x = b & 7 then walks
// ONE contiguous range of that order (the b with b % 8 == x are (K - x + 7) / 8 many).
__device__ __forceinline__ int fwd_roi_of(const PoolLevels& L, int b, int K) {
  if (L.perm == nullptr) return b;
  // runs of 16 consecutive ROIs of the order are dealt round-robin to the XCDs (b & 7 = XCD, b >> 3 = its slot):
  // neighbours share an L2, and every XCD gets the same mix of levels (one contiguous range per XCD gave XCD 0 all
  // the small ROIs and XCD 7 all the large ones: slower than no order at all)
  const int K0 = K & ~127;
  if (b >= K0) return L.perm[b];
  const int x = b & 7, t = b >> 3;
  return L.perm[(((t >> 4) << 3) + x) * 16 + (t & 15)];
}

// Per-ROI record written once per backward call by roi_records_kernel: what a tile workgroup needs
// to decide "does this ROI touch my tile" with a few integer compares (the first versions evaluated
// the whole geometry -- two IEEE divisions, sqrt, log2 -- per candidate and per tile: ~6k cycles of
// every workgroup), plus the geometry the weights need.
struct HitGeo { float start_h, start_w, bin_h, bin_w, inv; int grid; };  // grid = grid_h | grid_w << 16
// global tile id (numbering of tile_lists_kernel) of the first tile of each level in this launch
struct PoolTileIds { int first[POOL_MAX_LEVELS]; };
struct RoiRec {
  int level, batch;        // level = -1: contributes nothing (no level, empty sampling grid, outside)
  int fy0, fy1, fx0, fx1;  // conservative pixel rectangle [fy0, fy1] x [fx0, fx1] that can receive gradient
  HitGeo g;
};  // 48 bytes

// conservative footprint rectangle of an ROI on its level; false if it cannot contribute
__device__ __forceinline__ bool footprint_rect(const RoiGeom& g, int H, int W, int& fy0, int& fy1, int& fx0,
                                               int& fx1) {
  if (g.grid_h <= 0 || g.grid_w <= 0) return false;
  // samples lie strictly inside (start, start + roi) -- or (start + roi, start) for an inverted ROI (x2 < x1 with
  // aligned = True and sampling_ratio > 0: negative bin size, the forward still samples there); valid ones in
  // [-1, size]; pixels touched are floor(max(s, 0)) and +1, clamped to size - 1
  const float y_a = fminf(g.start_h, g.start_h + g.roi_h), yhi = fmaxf(g.start_h, g.start_h + g.roi_h);
  const float x_a = fminf(g.start_w, g.start_w + g.roi_w), xhi = fmaxf(g.start_w, g.start_w + g.roi_w);
  const float ylo = fmaxf(y_a, 0.f), xlo = fmaxf(x_a, 0.f);
  if (!(yhi >= -1.f && y_a <= (float)H && xhi >= -1.f && x_a <= (float)W)) return false;  // also NaN
  fy0 = (int)fminf(ylo, 1e9f); fy1 = min((int)fminf(fmaxf(yhi, 0.f), 1e9f) + 1, H - 1);
  fx0 = (int)fminf(xlo, 1e9f); fx1 = min((int)fminf(fmaxf(xhi, 0.f), 1e9f) + 1, W - 1);
  return true;
}

// the record of one ROI (roi_records_kernel; r06: also written by the paired forward's workgroups, see PoolFwdPair)
__device__ __forceinline__ RoiRec make_roi_rec(const PoolLevels& L, const float* __restrict__ r, int PH, int PW) {
  RoiRec o{};
  o.level = -1;
  o.batch = (int)r[0];
  const int lvl = assign_level(r + 1, L);
  if (lvl >= 0) {
    const RoiGeom g = roi_geom_box(r[0], r[1], r[2], r[3], r[4], L.scale[lvl], PH, PW, L.sr, L.aligned);
    if (footprint_rect(g, L.H[lvl], L.W[lvl], o.fy0, o.fy1, o.fx0, o.fx1)) {
      o.level = lvl;
      o.g = HitGeo{g.start_h, g.start_w, g.bin_h, g.bin_w, 1.f / (float)(g.grid_h * g.grid_w),
                   (g.grid_h & 0xffff) | (g.grid_w << 16)};
    }
  }
  return o;
}

// ------------------------------------------------------------------------------------------------
// FORWARD, NHWC.  grid = (K, nsplit); VEC = 16 B of channels per lane (or 1 for odd C / alignment)
// PAIRED launch (d2amd_roi_pooler_forward_pair; P.rois2 != nullptr, grid = K1 * ns1 + K2 * ns2 workgroups in x): the
// workgroups of a SECOND pooler of the same feature maps follow the first one's in the same grid -- the box head's
// launch ends with a quarter of the chip waiting for its largest ROIs (1,024 workgroups, all resident at once: mean
// 21.9 us, longest 37.7), the mask head's workgroups fill those slots instead of starting behind the last one.
// r06, rec != nullptr: the launch ALSO does what roi_records_kernel does for the paired backward of the same ROIs -- the
// first workgroup of an ROI writes its record (records [0, K1): the first pooler's, then the second one's), all workgroups
// together reset the backward's work queues -- into the backward's workspace, which the caller allocated ahead
// (d2amd_roi_pooler_forward_pair_records): the backward then starts with its tile lists, 7.6 us + a dependency edge earlier.
template <typename T> struct PoolFwdPair { const float* rois2; T* out2; int K1, ns1, K2, ns2, PH2, PW2;
                                           RoiRec* rec; int* qmem; int qzero, qints; };
template <typename T, int VEC, int NTHR, int U = FWD_U, int WPE = 1, bool PIPE = false>
__global__ __launch_bounds__(NTHR, WPE) void pool_fwd_nhwc_kernel(PoolLevels L, const float* __restrict__ rois_,
                                                                 T* __restrict__ out_, int nsplit_, PoolFwdPair<T> P) {
  __shared__ SepShared S;
  __shared__ int s_level;
  const float* rois = rois_;
  T* out = out_;
  int nsplit = nsplit_, bx = (int)blockIdx.x, by = (int)blockIdx.y, gx = (int)gridDim.x, PH = L.PH, PW = L.PW;
  bool second = false;
  if (P.rois2) {  // uniform
    const int n1 = P.K1 * P.ns1;
    if (bx < n1) {
      by = bx / P.K1; bx -= by * P.K1; gx = P.K1; nsplit = P.ns1;
    } else {
      bx -= n1; by = bx / P.K2; bx -= by * P.K2; gx = P.K2; nsplit = P.ns2;
      rois = P.rois2; out = P.out2; PH = P.PH2; PW = P.PW2;
      second = true;
    }
  }
  const int k = fwd_roi_of(L, bx, gx), tid = threadIdx.x;
  if (P.rec) {  // uniform (paired launch only: perm == nullptr, k == bx)
    for (int i = (int)blockIdx.x * NTHR + tid; i < P.qints; i += (int)gridDim.x * NTHR) P.qmem[i] = i < P.qzero ? 0 : -1;
    if (by == 0 && tid == 0) P.rec[(second ? P.K1 : 0) + k] = make_roi_rec(L, rois + (long)k * 5, PH, PW);
  }
  unsigned long long* wst = (L.wgstamps && tid == 0) ? L.wgstamps + 5 * ((size_t)by * gx + k) : nullptr;
  if (wst) wst[0] = wall_clock64();
  // every thread evaluates the (wave-uniform) level itself: one broadcast load, no LDS round trip / barrier
  const int lvl = __builtin_amdgcn_readfirstlane(assign_level(rois + (long)k * 5 + 1, L));
  const int C = L.C, bins = PH * PW;
  const int per = (bins + nsplit - 1) / nsplit;
  const int b_lo = by * per, b_hi = min(bins, b_lo + per);
  if (b_lo >= b_hi) return;
  const int CG = C / VEC;
  T* outk = out + (long)k * bins * C;
  if (lvl < 0) {  // reference: row of the zero-initialised output that no level fills
    for (int e = tid; e < (b_hi - b_lo) * C; e += NTHR) outk[(long)b_lo * C + e] = from_f32<T>(0.f);
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
  // U = independent loads in flight per lane
  if (wst) { wst[1] = wall_clock64(); wst[4] = (unsigned long long)lvl; }
  // index arithmetic without integer divisions: CG is a power of two for the usual channel counts, and
  // b / PW == (b * rcp_pw) >> 16 for b < 1024 (PH, PW <= 32)
  const int cg_shift = (CG & (CG - 1)) == 0 ? __builtin_ctz(CG) : -1;  // uniform
  const uint32_t rcp_pw = (65536u + (uint32_t)PW - 1u) / (uint32_t)PW;
  // TAP TABLE (v9).  The gather loop is VALU-issue bound and, per tap, spent as many instructions on rebuilding the
  // tap (row / column from the flattened index, two weight reads and their product, the element offset) as on using
  // it (8 unpack + 4 packed FMA) -- identically in each of the 32 channel lanes of a bin and again in every pass.
  // The workgroup now builds each bin's taps ONCE: {32-bit element offset from the image base, weight x 1 / count},
  // ntcap slots per bin in LDS; the loop reads a tap with one ds_read_b64.  ROIs with a bin of more than ntcap taps
  // (bins wider than ~4 px) keep the arithmetic path below.
  if constexpr (VEC > 1) {
    constexpr int TAPTAB = 3200;  // 196 bins x 16 taps
    __shared__ uint2 taptab[TAPTAB];
    const int nbins = b_hi - b_lo;
    const int lgcap = nbins * 32 <= TAPTAB ? 5 : nbins * 16 <= TAPTAB ? 4 : nbins * 8 <= TAPTAB ? 3 : -1;  // uniform
    bool over = lgcap < 0 || L.tab_off;  // tab_off: the image is too large for 32-bit offsets (host)
    if (!over) {
      for (int bl = tid; bl < nbins; bl += NTHR) {
        const int b = b_lo + bl;
        const int ph = (int)(((uint32_t)b * rcp_pw) >> 16), pw = b - ph * PW;
        over |= S.spany[ph] * S.spanx[pw] > (1 << lgcap);
      }
    }
    if (!__syncthreads_or(over)) {
      const int cap = 1 << lgcap;
      for (int idx = tid; idx < nbins << lgcap; idx += NTHR) {
        const int bl = idx >> lgcap, t = idx & (cap - 1);
        const int b = b_lo + bl;
        const int ph = (int)(((uint32_t)b * rcp_pw) >> 16), pw = b - ph * PW;
        const int sx = S.spanx[pw], nt = S.spany[ph] * sx;
        if (t < nt) {
          const uint32_t rcp = (65536u + (uint32_t)sx - 1u) / (uint32_t)sx;
          const uint32_t j = __umul24((uint32_t)t, rcp) >> 16, i = (uint32_t)t - __umul24(j, (uint32_t)sx);
          const uint32_t pix = __umul24((uint32_t)S.firsty[ph] + j, (uint32_t)W) + (uint32_t)S.firstx[pw] + i;
          taptab[idx] = uint2{pix * (uint32_t)C, __float_as_uint(S.wy[ph * SEP_SPAN + j] * S.wx[pw * SEP_SPAN + i] * inv)};
        }
      }
      __syncthreads();
      for (int e = tid; e < nbins * CG; e += NTHR) {
        const int bl = cg_shift >= 0 ? (e >> cg_shift) : e / CG, q = e - bl * CG;
        const int b = b_lo + bl;
        const int ph = (int)(((uint32_t)b * rcp_pw) >> 16), pw = b - ph * PW;
        const int nt = S.spany[ph] * S.spanx[pw];
        const uint2* tab = taptab + (bl << lgcap);
        const T* base = inb + q * VEC;
        float acc[VEC];
#pragma unroll
        for (int c = 0; c < VEC; c++) acc[c] = 0.f;
        if constexpr (PIPE) {
        // r06: the NEXT round's U loads are requested before this round's are consumed (two register sets, the same sequence
        // of additions: bit-identical).  Stand-alone the paired launch is 1 us slower with it (52.1 against 51.2 us), inside the
        // connected step -- beside the targets branch -- the step is 5 us shorter (0.3185-0.3192 against 0.3228-0.3277 ms,
        // same box; U = 6 at two workgroups per CU: 58 us alone, 0.318-0.322 in the step; U = 2 / 3 at four: no gain).  PIPE is on
        // for the 512-thread shape (the paired launch and single poolers of >= ~1,000 workgroups); the 1,024- / 256-thread
        // shapes of small launches gain nothing from it (maskrcnn_infer 0.384-0.387 with it against 0.381-0.384).
        auto ld = [&](int t0, raw16 (&raw)[U], float (&w)[U]) {
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int t = t0 + u;
            const uint2 tp = tab[min(t, nt - 1)];
            w[u] = t < nt ? __uint_as_float(tp.y) : 0.f;
            raw[u] = *reinterpret_cast<const raw16*>(base + tp.x);
          }
        };
        auto mac = [&](const raw16 (&raw)[U], const float (&w)[U]) {
#pragma unroll
          for (int u = 0; u < U; u++) {
            float f[VEC];
            unpack16(raw[u], f, T{});
#pragma unroll
            for (int c = 0; c < VEC; c++) acc[c] += w[u] * f[c];
          }
        };
        raw16 ra[U], rb[U];
        float wa[U], wb[U];
        if (nt > 0) ld(0, ra, wa);
        for (int t0 = 0; t0 < nt; t0 += 2 * U) {
          if (t0 + U < nt) ld(t0 + U, rb, wb);
          mac(ra, wa);
          if (t0 + U < nt) {
            if (t0 + 2 * U < nt) ld(t0 + 2 * U, ra, wa);
            mac(rb, wb);
          }
        }
        } else {
        for (int t0 = 0; t0 < nt; t0 += U) {
          raw16 raw[U];
          float w[U];
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int t = t0 + u;
            const uint2 tp = tab[min(t, nt - 1)];
            w[u] = t < nt ? __uint_as_float(tp.y) : 0.f;
            raw[u] = *reinterpret_cast<const raw16*>(base + tp.x);
          }
#pragma unroll
          for (int u = 0; u < U; u++) {
            float f[VEC];
            unpack16(raw[u], f, T{});
#pragma unroll
            for (int c = 0; c < VEC; c++) acc[c] += w[u] * f[c];
          }
        }
        }
        *reinterpret_cast<raw16*>(outk + (long)b * C + (long)q * VEC) = pack16(acc, T{});
      }
      if (wst) { wst[2] = wall_clock64(); wst[3] = wst[2]; }
      return;
    }
  }
  for (int e = tid; e < (b_hi - b_lo) * CG; e += NTHR) {
    const int bl = cg_shift >= 0 ? (e >> cg_shift) : e / CG, q = e - bl * CG;
    const int b = b_lo + bl;
    const int ph = (int)(((uint32_t)b * rcp_pw) >> 16), pw = b - ph * PW;
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
        raw16 raw[U];
        // The kernel is VALU-issue bound (the v8 ISA had five quarter-rate / 64-bit integer multiplies per tap and a
        // branch + LDS wait around every weight read): 24-bit multiplies (full rate; t < 4096, rcp < 2^17, pixel offset
        // < 2^15, C <= 8192 checked by the host), one 32-bit element offset per tap, weights read unconditionally
        // (clamped tap) and selected.
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int t = t0 + u, tt = min(t, nt - 1);
          const uint32_t j = __umul24((uint32_t)tt, rcp) >> 16, i = (uint32_t)tt - __umul24(j, (uint32_t)sx);
          const float wv = wy[j] * wx[i];
          w[u] = t < nt ? wv : 0.f;
          raw[u] = *reinterpret_cast<const raw16*>(base + __umul24(__umul24(j, (uint32_t)W) + i, (uint32_t)C));
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
      *reinterpret_cast<raw16*>(o) = pack16(acc, T{});
    } else {
      o[0] = from_f32<T>(acc[0]);
    }
  }
  if (wst) { wst[2] = wall_clock64(); wst[3] = wst[2]; }
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
//
// What the measurements of the first two versions (profiles/r01, DESIGN.md log) say about this kernel:
// it is bound by the LATENCY of the longest per-tile ROI list, not by bandwidth.  Coarse FPN levels
// have few tiles and every large ROI covers many of them (p4 of the bench batch: 154 tiles, lists of
// up to 33 ROIs; p2: 2,100 tiles, lists <= 9), and a workgroup walks its list serially.  Hence:
//   * the list scan stores the geometry of every hit in LDS (no global load, no division in the
//     per-ROI loop) and loads candidate ROIs unconditionally (a conditional load compiles to
//     load + s_waitcnt vmcnt(0));
//   * per ROI, all dY loads of a lane are issued first (NB in flight), the axis weights of the NEXT
//     ROI are computed while they fly (weights are double-buffered), then the FMAs run: one
//     barrier and about one exposed L2 latency per ROI;
//   * a group is 512 threads: 2 row halves x 8 pixel columns x 32 channel lanes, 4 rows per thread
//     (32 accumulators: ~100 VGPRs, 4 waves / SIMD);
//   * GROUPS > 1 (coarse levels): the workgroup has GROUPS x 512 threads, group g walks list entries
//     g, g + GROUPS, ... with private accumulators, and the partial tiles are summed through LDS in a
//     fixed order (still deterministic) -- the critical path shrinks GROUPS-fold.
constexpr int TILE = 8;          // TILE x TILE pixels per workgroup
constexpr int LPP = 32;          // lanes (16-B channel groups) per pixel; 256 threads = 8 pixel columns
constexpr int LCH = 512;         // ROIs scanned per list-building pass
constexpr int MAXP = SEP_MAXP;   // 32: one lane per bin along an axis
constexpr int CT = 256;          // threads per row-split of a group: 8 pixel columns x 32 channel lanes
constexpr int QCTR = 320;        // work-queue counters: [pass][xcd] heads (0..15), [pass][xcd] tails (16..31) of the
                                 // binning; from QTAKE on, one 128-B line per XCD queue with the TAKE counter of the
                                 // persistent tile workgroups (all eight in one line: every fetch of the chip queued
                                 // up behind one memory channel)
constexpr int QTAKE = 64, QTAKE_PITCH = 32;  // (word 1 of a queue's line: scratch slots handed out to its split tiles)
// SPLIT TILES.  A tile's ROI list is walked by one workgroup, one item after the other; the box head's heaviest p4 tile
// holds 33 ROIs = 46-57 us of a kernel whose work, spread evenly, takes 49 us.  Lists longer than SPLIT_MIN entries are
// cut into PARTS of <= PART_LEN entries, each a queue entry of its own on the tile's XCD queue; a part leaves its fp32
// accumulators in a scratch slot and takes a ticket, the workgroup that draws the last ticket adds the parts IN PART
// ORDER (deterministic) and writes the tile.  Nobody waits for anybody.
// WHICH lists are split must not depend on the order in which the binning workgroups run (the parts' sums are added in
// part order, an unsplit list's items one after the other: the same tile must take the same route in every run), so the
// scratch budget is handed out in TILE ORDER by the last binning workgroup to finish (tile_lists_kernel).
// (Late round 3: for the bench's lists -- longest 33 -- cutting at 24 changes neither gather (65.1 / 38.9 us with and
// without, same box) while the planner costs the binning kernel 4.3 us of its 17.9 on the step's critical path.  Lists
// were cut from 40 entries on in round 4, and the planner only runs when a binning wave has SEEN such a list.)
// (Round 5, --rois clustered: a trained RPN piles the 1,000 proposals and the positives on 16 objects -- 211 tiles with
// 17-40 entries, 17 with more, longest 52 (bench.py: roi_tiles) -- and with the cut at 40 the paired gather took 133.8 us
// against 82 for the spread-out lists.  Same-box A/B of (SPLIT_MIN, PART_LEN): (40, 16) 133.8 | (24, 16) 116.4 | (20, 10)
// 104.2 | (16, 8) 101.6 | (12, 8) 114.8 | (12, 6) 122.1 us (finer cuts exhaust the scratch slots, and every part pays
// the tile's prologue and a scratch round trip); the spread-out lists (longest 16) never reach the planner: 80-82 us
// with every setting.)
constexpr int PART_LEN = 20, SPLIT_MIN = 40, MAX_PARTS = 6;  // (weight units: 16 bins of the entries' windows)
constexpr int SCR_PER_XCD_MAX = 96;  // scratch slots per XCD queue (64 px x C fp32 each: 48 MB at C = 256)
constexpr int SPLIT_MAX_SLABS = 4;   // channel slabs (of 256 channels, 16-bit) a split tile may have: one ticket each
constexpr int QTICKETS = 8 * SCR_PER_XCD_MAX * SPLIT_MAX_SLABS;

template <int GROUPS>
struct TileShared {
  int list[LCH];
  HitGeo geo[LCH];
  float WyT[GROUPS][2][MAXP][TILE];          // [group][buffer][bin][tile row]
  float Wx[GROUPS][2][TILE][MAXP];           // [group][buffer][tile col][bin], carries 1/count
  uint32_t ymask[GROUPS][2][TILE], xmask[GROUPS][2][TILE];
  int wave_cnt[LCH / 64];
};

// A SECOND pooler binned together with the first one (d2amd_roi_pooler_backward_pair): records [K1, L.K) are its ROIs,
// evaluated with its pooled size (same feature maps, level rule, sampling ratio and alignment: checked by the host).
struct PairBin { const float* rois2; int K1, PH2, PW2; };  // rois2 == nullptr: one pooler, K1 = L.K
// Also resets the work queues of the backward launches (counters = 0, slots = -1 "no tile").
__global__ void roi_records_kernel(PoolLevels L, const float* __restrict__ rois, RoiRec* __restrict__ rec,
                                   int* __restrict__ qmem, int qzero, int qints, PairBin B) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = k; i < qints; i += gridDim.x * blockDim.x) qmem[i] = i < qzero ? 0 : -1;  // counters + tickets | slots
  if (k >= L.K) return;
  const bool second = B.rois2 != nullptr && k >= B.K1;
  const float* r = second ? B.rois2 + (long)(k - B.K1) * 5 : rois + (long)k * 5;
  const int PH = second ? B.PH2 : L.PH, PW = second ? B.PW2 : L.PW;
  rec[k] = make_roi_rec(L, r, PH, PW);
}

// Per-tile ROI lists, built once per backward call by one WAVE per tile: the 64 lanes test 64 records at
// a time against the tile (a few integer compares on the record heads), ballot + prefix keep ROI order.
// Without them every tile workgroup scanned all K records itself: 2.2 us per 512 records and tile
// (profiles/r01/v4_pool_bwd_timeline.txt), 4.4 of the 5.5 us an EMPTY box-head tile took.  Tiles with
// more than TILE_CAP ROIs (clustered proposals) keep the in-kernel scan.
constexpr int TILE_CAP = 64;
// list entry = ROI index + the geometry the weights need (one dependent load less in the tile workgroup)
// inclusive prefix sum over the 64 lanes on the DPP network (row shifts inside the rows of 16, then the row broadcasts):
// six dependent VALU instructions where six __shfl_up are six LDS-crossbar round trips
__device__ __forceinline__ int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return x;
}

// (win: the window of bins that can touch the tile, ph_lo | nph << 8 | pw_lo << 16 | npw << 24 -- axis_window below)
struct __attribute__((aligned(16))) TileEntry { HitGeo g; int roi; int win; };  // 32 bytes
static_assert(sizeof(TileEntry) == 32, "TileEntry layout");
// CONSERVATIVE range of bins, along one axis, with a sample that can put weight on the pixels [t0, t0 + 8) of a map of
// `size` pixels: lo | n << 8.  Sample i of bin p lies at start + p bin + (i + 0.5) bin / grid (axis_weight); it reaches
// pixel x iff it is valid (in [-1, size]) and |clamp(y, 0, size - 1) - x| < 1, i.e. iff clamp(y) lies in the OPEN interval
// (t0 - 1, t0 + 8) -- widened to everything below / above for the first / last tile, where the clamp folds the border
// strip onto the edge pixel.  The bins whose sample span meets that interval are p in (a, b) with the bounds below
// (for a negative bin size -- an inverted ROI with a fixed sampling ratio -- first and last sample swap roles).  The
// bounds are widened by 0.02 bin: every bin with a non-zero weight is inside (the rounding of the bounds is orders of
// magnitude below that wherever they are not clamped), a bin too many only adds a zero weight.  Degenerate bins take
// the whole range.
__device__ __forceinline__ int axis_window(float start, float bin, int grid, int P, int t0, int size) {
  int p_lo = 0, p_hi = P - 1;
  if (fabsf(bin) >= 0.01f && fabsf(start) < 1e6f) {  // (NaN: the whole range)
    const float lo = t0 == 0 ? -2.f : (float)(t0 - 1), hi = t0 + TILE >= size ? (float)(size + 1) : (float)(t0 + TILE);
    // (v_rcp_f32 + a multiplication: RELATIVE error ~2e-7 -- where the bounds decide anything, |u|, |v| <= P + 1 <= 33 and
    // the error is < 1e-5 bin; larger values are clamped below.  The correctly rounded division the library is built
    // with costs ~10 x as many instructions, and this runs per (tile, ROI) pair on the critical path of the binning.)
    const float rg = __builtin_amdgcn_rcpf((float)grid), rb = __builtin_amdgcn_rcpf(bin);
    const float u = (lo - start) * rb, v = (hi - start) * rb;
    const float a = (bin > 0.f ? u : v) - ((float)grid - 0.5f) * rg;  // p > a
    const float b = (bin > 0.f ? v : u) - 0.5f * rg;                  // p < b
    p_lo = max(0, (int)floorf(fminf(fmaxf(a - 0.02f, -1.f), (float)P)) + 1);
    p_hi = min(P - 1, (int)ceilf(fminf(fmaxf(b + 0.02f, -1.f), (float)P + 1.f)) - 1);
  }
  return p_lo | (max(0, p_hi - p_lo + 1) << 8);
}
__device__ __forceinline__ int entry_window(const HitGeo& g, int PH, int PW, int y0, int x0, int H, int W) {
  return axis_window(g.start_h, g.bin_h, g.grid & 0xffff, PH, y0, H) |
      (axis_window(g.start_w, g.bin_w, g.grid >> 16, PW, x0, W) << 16);
}
// WEIGHT of a tile = (bins of all its entries' windows = the k's of its contraction) / 16.  (The gather's time per tile
// fits 4.9 us + 0.19 us per entry + 0.052 us per k, profiles/r06/pool_bwd_kcat.md; counting the entries in -- (k + 4 n) / 16
// -- moved tiles between the queues' heavy and light ends and cost the paired launch 3 us of 50, same box.)  What the
// heavy-first queues and the split planner count with -- a 7 x 7 pooler's entry is about one unit, a 14 x 14 pooler's 3-12 (a small
// ROI whose 196 bins all fall into one tile: 12), so entry counts misjudge a paired launch's lists by that much.
__device__ __forceinline__ int win_bins(int win) { return ((win >> 8) & 0xff) * ((win >> 24) & 0xff); }
// per-tile word of the binning: the list's length, and -- for the planner -- its weight (lists that fit only; 0 = no ROI)
__host__ __device__ __forceinline__ int tile_count_pack(int cnt, int wgt) {
  return cnt == 0 ? 0 : cnt <= 64 ? (1 << 30) | (wgt << 8) | cnt : cnt;  // (64 = TILE_CAP)
}
__host__ __device__ __forceinline__ int tile_count_of(int v) { return (v >> 30) & 1 ? v & 0xff : v; }
__host__ __device__ __forceinline__ int tile_weight_of(int v) { return (v >> 30) & 1 ? (v >> 8) & 0x3fffff : v; }
struct TileGeom { int lvl, n, y0, x0; };
__device__ __forceinline__ TileGeom tile_geom(const PoolLevels& L, int tile) {
  TileGeom g;
  g.lvl = 0;
#pragma unroll
  for (int l = 1; l < POOL_MAX_LEVELS; l++)
    if (l < L.num_levels && tile >= L.tile_base[l]) g.lvl = l;
  const int H = L.H[g.lvl], W = L.W[g.lvl];
  const int tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
  int tl = tile - L.tile_base[g.lvl];
  g.n = tl / (tiles_y * tiles_x);
  tl -= g.n * tiles_y * tiles_x;
  g.y0 = (tl / tiles_x) * 8;
  g.x0 = (tl % tiles_x) * 8;
  return g;
}

// Work queues (v8).  The static blockIdx -> tile order of the first versions gave each XCD one contiguous run of
// tiles; the levels are numbered one after the other, so the XCDs that got the coarser level (longer ROI lists)
// finished last while the others idled (profiles/r01/v5_pool_bwd_timeline.txt: 90 % of the workgroups had started
// after 46 us of a 98 us kernel), and ~12 % (box) / 55 % (mask) of the workgroups only wrote zeros.  Now the wave
// that bins the ROIs of a tile also schedules it:
//   * a tile with no ROI is zero-filled right here (16-B stores) and never reaches a tile workgroup;
//   * the others are pushed on the queue of their XCD: 4x4-tile blocks are dealt to the 8 XCDs (neighbouring
//     tiles share the dY rows of their ROIs in that XCD's L2), heavy tiles (>= thr ROIs) from the front, light
//     ones from the back, so the long lists start first (longest-processing-time-first) and the empty slots in
//     between cost one L2 round trip each.
// Workgroup b of a backward launch serves slot b >> 3 of the queue of XCD b & 7 (workgroups are dealt
// round-robin to the XCDs).  The order in which tiles are processed depends on atomics; the value written to
// every pixel does not (one workgroup per tile, fixed ROI order): the backward stays deterministic.
struct TileQueues {
  int* mem;                        // QCTR counters, then pass 0 queues [8][cap[0]], then pass 1 queues [8][cap[1]]
  int cap[2], thr[2];
  int pass_base[POOL_MAX_LEVELS];  // pass-local tile id of the first tile of each level
  int deal_shift[POOL_MAX_LEVELS]; // level l is dealt to the XCDs in blocks of (1 << shift) x (1 << shift) tiles
  int split_min, part_len;         // lists heavier than split_min (weight units) are cut into parts of about part_len
  int qbase;                       // ints from mem to the first queue slot (counters, then the split tiles' tickets)
  int scr_total;                   // scratch slots the split lists of a launch may take (0: lists are never split)
  unsigned coarse_mask;            // bit l: level l belongs to pass 1
  int esize, zero_fill;            // element size; 1: empty tiles are zero-filled here (16-B aligned rows)
};
// 4x4-tile blocks keep the dY rows of neighbouring tiles in ONE L2 -- right for the fine levels (thousands of tiles,
// 2-4 ROIs each).  A coarse level has few tiles with LONG lists (p4 of the bench: 154 tiles, 17 ROIs on average): dealt
// in 4x4 blocks, 16 of them land on one queue and the XCDs end 12 us apart; those levels are dealt tile by tile
// (shift 0) or in 2x2 blocks (shift 1): host, pool_deal_shift().
__host__ __device__ __forceinline__ int tile_xcd(int lvl, int n, int ty, int tx, int tiles_x, int shift) {
  return ((ty >> shift) * ((tiles_x + (1 << shift) - 1) >> shift) + (tx >> shift) + 3 * n + 5 * lvl) & 7;
}

// L.tile_base here numbers ALL tiles of all levels (make_levels); the two backward launches map their
// own tile numbering onto it through `first` (tile id of their first tile per level).
constexpr int LISTS_WAVES = 16;  // tiles (waves) per workgroup of tile_lists_kernel
// (tile_cnt1 != nullptr: two poolers binned together -- records [0, K1) are the first one's; the number of its entries in
// a tile's list goes to tile_cnt1, the second pooler's entries follow them in the list)
__global__ __launch_bounds__(64 * LISTS_WAVES) void tile_lists_kernel(PoolLevels L, const RoiRec* __restrict__ rec,
                                                                     int ntiles, int* __restrict__ tile_cnt,
                                                                     TileEntry* __restrict__ tile_list, TileQueues Q,
                                                                     int K1, int* __restrict__ tile_cnt1, int PH2, int PW2,
                                                                     int4* __restrict__ tile_geo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * LISTS_WAVES + wave;
  const bool live = tile < ntiles;  // uniform per wave; dead waves only take part in the barriers
  TileGeom g{};
  int cnt = 0, cnt1 = 0, wgt = 0;
  if (live) {
    g = tile_geom(L, tile);
    int kbins = 0;  // this lane's share of the list's window bins
    constexpr int UN = 4;  // record heads of 4 x 64 ROIs in flight (one L2 round trip instead of four)
    for (int k0 = 0; k0 < L.K; k0 += 64 * UN) {
      int4 ra[UN];
      int2 rb[UN];
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const int k = min(k0 + u * 64 + lane, L.K - 1);
        ra[u] = *reinterpret_cast<const int4*>(&rec[k].level);  // level, batch, fy0, fy1
        rb[u] = *reinterpret_cast<const int2*>(&rec[k].fx0);    // fx0, fx1
      }
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const int kk = k0 + u * 64 + lane;
        const bool hit = kk < L.K && ra[u].x == g.lvl && ra[u].y == g.n && ra[u].w >= g.y0 && ra[u].z < g.y0 + 8 &&
            rb[u].y >= g.x0 && rb[u].x < g.x0 + 8;
        const unsigned long long bal = __ballot(hit);
        const int pos = cnt + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        if (hit && pos < TILE_CAP) {
          TileEntry e;
          e.g = rec[kk].g;
          e.roi = kk;
          e.win = entry_window(e.g, kk < K1 ? L.PH : PH2, kk < K1 ? L.PW : PW2, g.y0, g.x0, L.H[g.lvl], L.W[g.lvl]);
          tile_list[(long)tile * TILE_CAP + pos] = e;
          kbins += win_bins(e.win);
        }
        cnt += __builtin_popcountll(bal);
        cnt1 += __builtin_popcountll(__ballot(hit && kk < K1));
      }
    }
    wgt = min((__builtin_amdgcn_readlane(wave_incl_scan(kbins), 63) + 15) >> 4, 0x3fffff);
    if (cnt > TILE_CAP) wgt = cnt;  // (the list did not fit: the gather scans the records; never split, always heavy)
    // The planner (another workgroup of THIS launch) reads the count and the weight: ONE device-scope word.  What only the
    // gather -- a later launch -- reads is stored plainly: three more write-through stores per tile, each waited for in
    // front of the planner's ticket, cost this kernel 6.5 us of 23 (same-box A/B, profiles/r06/pool_bwd_kcat.md).
    if (lane == 0) __hip_atomic_store(&tile_cnt[tile], tile_count_pack(cnt, wgt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0 && tile_cnt1) tile_cnt1[tile] = cnt1;
    if (lane == 0 && tile_geo) tile_geo[tile] = int4{g.lvl | (g.n << 8), g.y0 | (g.x0 << 16), L.H[g.lvl] | (L.W[g.lvl] << 16), wgt};
  }
  if (Q.mem == nullptr) return;  // uniform
  const int H = L.H[g.lvl], W = L.W[g.lvl];
  bool push = live;
  if (live && cnt == 0 && L.accumulate) {
    push = false;  // accumulate mode: a tile no ROI touches keeps what it holds -- no write, no workgroup
  } else if (live && cnt == 0 && Q.zero_fill) {  // nothing to gather: write the zeros here
    const int rows = min(8, H - g.y0), cols = min(8, W - g.x0);
    const long px = (long)L.C * Q.esize, rowbytes = cols * px;
    char* base = (char*)L.data[g.lvl] + (((long)g.n * H + g.y0) * W + g.x0) * px;
    for (int r = 0; r < rows; r++)
      for (long o = lane * 16; o < rowbytes; o += 64 * 16)
        *reinterpret_cast<uint4*>(base + (long)r * W * px + o) = uint4{0u, 0u, 0u, 0u};
    push = false;
  }
  // Queue slots: ONE atomic per workgroup and counter.  A returning atomic per tile on the 16 counters of a launch
  // (8 XCDs x heavy / light) serialised in L2 at ~0.1 us each: 2,500 tiles -> 16 us of this kernel.
  __shared__ int s_key[LISTS_WAVES];   // counter index of the wave's tile ([heavy / light][pass][xcd]), -1: none
  __shared__ int s_np[LISTS_WAVES];    // queue entries of the wave's tile (> 1: a split list)
  __shared__ int s_slot[LISTS_WAVES];  // queue position handed to the wave
  int key = -1, pass = 0, x = 0, np = 1, sbase = 0;
  bool heavy = false;
  if (push) {
    pass = (Q.coarse_mask >> g.lvl) & 1;
    int shift = Q.deal_shift[0];
#pragma unroll
    for (int l = 1; l < POOL_MAX_LEVELS; l++)
      if (l == g.lvl) shift = Q.deal_shift[l];
    x = tile_xcd(g.lvl, g.n, g.y0 >> 3, g.x0 >> 3, (W + 7) >> 3, shift);
    heavy = wgt >= Q.thr[pass];
    key = (heavy ? 0 : 16) + pass * 8 + x;
    // a list heavy enough to be split is queued by the planner below (the last workgroup), not here
    if (pass == 0 && Q.scr_total > 0 && wgt > Q.split_min && cnt >= 2 && cnt <= TILE_CAP) {
      push = false; key = -1;
      if (lane == 0) atomicOr(Q.mem + QTAKE + 2, 1);  // (device scope; acknowledged before the ticket below)
    }
  }
  if (lane == 0) { s_key[wave] = key; s_np[wave] = np; }
  __syncthreads();
  if (threadIdx.x < LISTS_WAVES) {  // lane t of wave 0 serves wave t's tile
    const int mine = s_key[threadIdx.x];
    int rank = 0, total = 0, first = LISTS_WAVES;
    for (int q = 0; q < LISTS_WAVES; q++) {
      const bool same = mine >= 0 && s_key[q] == mine;
      if (same && q < (int)threadIdx.x) rank += s_np[q];
      if (same) { total += s_np[q]; first = min(first, q); }
    }
    int basepos = 0;
    if (mine >= 0 && first == (int)threadIdx.x) basepos = atomicAdd(Q.mem + mine, total);
    // hand the base from the first wave with this key to the others
    basepos = __shfl(basepos, first < LISTS_WAVES ? first : 0, 64);
    s_slot[threadIdx.x] = basepos + rank;
  }
  __syncthreads();
  if (push && lane == 0) {
    const int u = Q.pass_base[g.lvl] + (tile - L.tile_base[g.lvl]);  // tile id inside its launch
    int2* q = reinterpret_cast<int2*>(Q.mem + Q.qbase) + (pass ? 8 * Q.cap[0] : 0) + x * Q.cap[pass];
    const int at = s_slot[wave];
    const int e = (min(cnt, 127) << 24) | u;  // positive: -1 marks an unused slot (127 > TILE_CAP)
    for (int part = 0; part < np; part++)     // (np == 1: the whole list)
      q[heavy ? at + part : Q.cap[pass] - 1 - at] = int2{e, part | (np << 8) | (sbase << 16)};
  }
  if (Q.scr_total <= 0) return;  // uniform
  // ---- split planner: the LAST workgroup to get here (every count is then visible: device-scope stores, acknowledged
  // before the ticket) walks the tiles in tile order, hands scratch slots to the lists longer than SPLIT_MIN until the
  // budget is spent -- a function of the counts alone -- and queues them: np parts, or whole if the budget is spent
  __shared__ int s_last, s_scan[LISTS_WAVES], s_run;
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(Q.mem + QTAKE + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  if (!s_last) return;  // uniform
  if (__hip_atomic_load(Q.mem + QTAKE + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;  // no long list
  constexpr int PT = 64 * LISTS_WAVES;
  for (int t0 = 0; t0 < ntiles; t0 += PT) {
    const int t = t0 + (int)threadIdx.x;
    int c = 0;
    int wgt = 0;
    if (t < ntiles) {
      const int v = __hip_atomic_load(&tile_cnt[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      c = tile_count_of(v);
      wgt = tile_weight_of(v);
    }
    TileGeom tg{};
    bool cand = wgt > Q.split_min && c >= 2 && c <= TILE_CAP;
    if (cand) {
      tg = tile_geom(L, t);
      cand = ((Q.coarse_mask >> tg.lvl) & 1) == 0;
    }
    // parts of equal entry counts, none empty: ceil(c / ceil(c / parts wanted by weight))
    int want = 0;
    if (cand) {
      const int w0 = min(min((wgt + Q.part_len - 1) / Q.part_len, MAX_PARTS), c);
      const int len = (c + w0 - 1) / w0;
      want = (c + len - 1) / len;
    }
    // exclusive scan of `want` in tile order: inside the wave, then over the 16 waves, then the running total
    int incl = want;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(incl, d, 64);
      if (lane >= d) incl += y;
    }
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    int before = s_run;
    for (int w2 = 0; w2 < wave; w2++) before += s_scan[w2];
    const int mybase = before + incl - want;
    if (cand) {
      const bool fits = mybase + want <= Q.scr_total;  // monotone in tile order: once spent, spent for all later tiles
      const int n_parts = fits ? want : 1;
      const int W2 = L.W[tg.lvl];
      int shift = Q.deal_shift[0];
#pragma unroll
      for (int l = 1; l < POOL_MAX_LEVELS; l++)
        if (l == tg.lvl) shift = Q.deal_shift[l];
      const int xq = tile_xcd(tg.lvl, tg.n, tg.y0 >> 3, tg.x0 >> 3, (W2 + 7) >> 3, shift);
      const int at = atomicAdd(Q.mem + xq, n_parts);  // heavy end of the XCD's queue (pass 0)
      const int u = Q.pass_base[tg.lvl] + (t - L.tile_base[tg.lvl]);
      int2* q = reinterpret_cast<int2*>(Q.mem + Q.qbase) + xq * Q.cap[0];
      for (int part = 0; part < n_parts; part++)
        q[at + part] = int2{(c << 24) | u, part | (n_parts << 8) | ((fits ? mybase : 0) << 16)};
    }
    __syncthreads();
    if (threadIdx.x == PT - 1) s_run = before + incl;  // (the last thread's inclusive total of this chunk)
    __syncthreads();
  }
}

// The zero fill of the tiles no ROI touches as its own launch: for a binning that ran ahead of the backward (beside
// the forward, in accumulate mode: no gradient tensor existed yet) and is now used for a WRITING gather.
__global__ __launch_bounds__(64 * LISTS_WAVES) void zero_empty_tiles_kernel(PoolLevels L, int ntiles,
                                                                           const int* __restrict__ tile_cnt, int esize) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * LISTS_WAVES + wave;
  if (tile >= ntiles || tile_cnt[tile] != 0) return;  // (tile_count_pack(0, .) == 0)
  const TileGeom g = tile_geom(L, tile);
  const int H = L.H[g.lvl], W = L.W[g.lvl];
  const int rows = min(8, H - g.y0), cols = min(8, W - g.x0);
  const long px = (long)L.C * esize, rowbytes = cols * px;
  char* base = (char*)L.data[g.lvl] + (((long)g.n * H + g.y0) * W + g.x0) * px;
  for (int r = 0; r < rows; r++)
    for (long o = lane * 16; o < rowbytes; o += 64 * 16)
      *reinterpret_cast<uint4*>(base + (long)r * W * px + o) = uint4{0u, 0u, 0u, 0u};
}

// total weight the `grid` samples of bin p put on pixel `pix` along one axis
// (axis_tap in closed form: a valid sample at y puts max(0, 1 - |clamp(y, 0, size - 1) - pix|) on pixel pix --
// for y in [lo, lo + 1) that is 1 - l on lo and l on lo + 1, at / beyond the last pixel and below 0 the full weight
// on the border pixel; the bin's sample spacing bin / grid is divided once.  Positions differ from the forward's
// expression order by an ulp, which moves a weight by ~1e-7: the backward's bar is 1e-4.  The tile kernels are VALU
// issue bound and spend a quarter of their instructions here.)
__device__ __forceinline__ float axis_weight(float start, float bin, int grid, int p, int pix, int size) {
  const float step = bin / (float)grid, y0 = start + (float)p * bin + 0.5f * step;
  const float fpix = (float)pix, last = (float)(size - 1), fsize = (float)size;
  float w = 0.f;
  for (int i = 0; i < grid; i++) {
    const float y = y0 + (float)i * step;
    const float t = 1.f - fabsf(fminf(fmaxf(y, 0.f), last) - fpix);
    w += (y >= -1.0f && y <= fsize) ? fmaxf(t, 0.f) : 0.f;
  }
  return w;
}

// RS = row splits of the tile inside a group (1: a thread owns all 8 rows of its pixel column,
// 2: 4 rows); a group has RS * 256 threads.  GROUPS = list splits.  NB = loads in flight per lane.
template <typename T, int VEC, int GROUPS, int RS, int NB>
// 4 waves / SIMD (<= 128 VGPRs): two 512-thread workgroups per CU; measured faster than the 146-178 VGPR builds
__global__ __launch_bounds__(CT * RS * GROUPS, 4) void pool_bwd_nhwc_kernel(PoolLevels L,
                                                                         const RoiRec* __restrict__ rec,
                                                                         const T* __restrict__ gout, int nslab,
                                                                         int total_blocks, PoolTileIds ids) {
  constexpr int GT = CT * RS, NT = GT * GROUPS, TR = TILE / RS;
  __shared__ TileShared<GROUPS> S;
  const int tid = threadIdx.x, lane = tid & 63;
  const int grp = tid / GT, t = tid % GT;  // group, thread in group
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so give each XCD one
  // contiguous run of tiles (neighbouring tiles share the dY rows of their ROIs in that XCD's L2)
  int logical, slab, tile, qcnt = -1;
  if (L.queue) {  // work queue of this XCD (tile_lists_kernel): heavy tiles first, empty tiles never arrive
    const int j = (int)(blockIdx.x >> 3);
    const int e = L.queue[(long)(blockIdx.x & 7) * L.qcap + j / nslab].x;  // (lists are split for the MFMA kernel only)
    if (e < 0) return;
    logical = (int)blockIdx.x;
    slab = j % nslab;
    tile = e & 0xffffff;
    qcnt = (int)((unsigned)e >> 24);
  } else {
    const int per_xcd = (total_blocks + 7) >> 3;
    logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= total_blocks) return;
    slab = logical % nslab;
    tile = logical / nslab;
  }
  unsigned long long* wst = (L.wgstamps && tid == 0) ? L.wgstamps + 5 * (size_t)logical : nullptr;
  if (wst) wst[0] = wall_clock64();
  unsigned long long wst_list = 0;
  int wst_n = 0;
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < POOL_MAX_LEVELS; l++)
    if (l < L.num_levels && tile >= L.tile_base[l]) lvl = l;
  const int H = L.H[lvl], W = L.W[lvl];
  const int tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE;
  int tl = tile - L.tile_base[lvl];
  const int n = tl / (tiles_y * tiles_x);
  tl -= n * tiles_y * tiles_x;
  const int y0 = (tl / tiles_x) * TILE, x0 = (tl % tiles_x) * TILE;
  const int C = L.C, PH = L.PH, PW = L.PW, K = L.K;
  const int CG = C / VEC;
  const int rh = t / CT, col = (t >> 5) & 7, lp = t & 31;  // row split / pixel column / channel lane
  const int cg = slab * LPP + lp;
  const bool cg_ok = cg < CG;
  const long cofs = (long)min(cg, CG - 1) * VEC;

  float acc[TR][VEC];  // rows rh*TR .. rh*TR + TR - 1 of column col
#pragma unroll
  for (int i = 0; i < TR; i++)
#pragma unroll
    for (int c = 0; c < VEC; c++) acc[i][c] = 0.f;

  // axis weights of list entry `li` -> buffer wb of this group.  RS == 1: every thread evaluates one
  // (tile row, bin) and one (tile col, bin) pair; RS == 2: threads 0-255 the rows, 256-511 the columns.
  auto compute_weights = [&](int li, int wb) __attribute__((always_inline)) {
    const HitGeo g = S.geo[li];
    const int grid_h = g.grid & 0xffff, grid_w = g.grid >> 16;
    const int r = (t >> 5) & 7, p = t & 31, w4 = (t >> 6) & 3;
    if (RS == 1 || t < CT) {
      float wv = 0.f;
      if (p < PH && y0 + r < H) wv = axis_weight(g.start_h, g.bin_h, grid_h, p, y0 + r, H);
      S.WyT[grp][wb][p][r] = wv;
      const unsigned long long bm = __ballot(wv != 0.f);
      if (lane == 0) { S.ymask[grp][wb][2 * w4] = (uint32_t)bm; S.ymask[grp][wb][2 * w4 + 1] = (uint32_t)(bm >> 32); }
    }
    if (RS == 1 || t >= CT) {
      float wv = 0.f;
      if (p < PW && x0 + r < W) wv = axis_weight(g.start_w, g.bin_w, grid_w, p, x0 + r, W);
      S.Wx[grp][wb][r][p] = wv * g.inv;
      const unsigned long long bm = __ballot(wv != 0.f);
      if (lane == 0) { S.xmask[grp][wb][2 * w4] = (uint32_t)bm; S.xmask[grp][wb][2 * w4 + 1] = (uint32_t)(bm >> 32); }
    }
  };

#ifdef D2AMD_PROFILE
  const bool dbg_on = L.dbg && logical == L.dbg_block && tid == 0;
  int dbg_n = 0;
#define STAMP() do { if (dbg_on && dbg_n < 120) L.dbg[dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP() do {} while (0)
#endif
  STAMP();
#ifdef D2AMD_PROFILE
  const unsigned long long rt0 = wall_clock64();
#endif
  // prepared list of this tile (tile_lists_kernel), if it fits.  The entries carry the geometry, so the
  // workgroup pays two dependent memory round trips (count, entries) before the weights instead of three
  // (count, indices, records).  Entries are only read when they were written: fetching all TILE_CAP slots
  // unconditionally (to overlap with the count) reads cold, never-written memory and was measured 2x slower.
  int tl_cnt = -1;
  if (L.tile_cnt) {
    const int gtile = ids.first[lvl] + (tile - L.tile_base[lvl]);
    const int c = qcnt >= 0 ? qcnt : tile_count_of(L.tile_cnt[gtile]);
    if (c <= TILE_CAP) {
      tl_cnt = c;
      if (tid < c) {
        const TileEntry e = ((const TileEntry*)L.tile_list)[(long)gtile * TILE_CAP + tid];
        S.list[tid] = e.roi;
        S.geo[tid] = e.g;
      }
    }
  }
  const bool prelist = tl_cnt >= 0;  // uniform
  for (int kbase = 0; kbase < (prelist ? 1 : K); kbase += LCH) {
    int nlist = 0;
    if (prelist) {
      nlist = tl_cnt;
    } else {
    // ---- (1) ordered list (+ geometry) of the ROIs of this chunk that touch the tile ----------
    const int kend = min(K, kbase + LCH);
    constexpr int ROUNDS = (LCH + NT - 1) / NT;
    int4 ra[ROUNDS];
    int2 rb[ROUNDS];
#pragma unroll
    for (int q = 0; q < ROUNDS; q++) {  // raw clamped loads of the record heads, all in flight together
      const long r = min(kbase + q * NT + tid, K - 1);
      ra[q] = *reinterpret_cast<const int4*>(&rec[r].level);  // level, batch, fy0, fy1
      rb[q] = *reinterpret_cast<const int2*>(&rec[r].fx0);    // fx0, fx1
    }
    bool hit[ROUNDS];
    unsigned long long bal[ROUNDS];
#pragma unroll
    for (int q = 0; q < ROUNDS; q++) {
      const int r = kbase + q * NT + tid;
      hit[q] = r < kend && q * NT + tid < LCH && ra[q].x == lvl && ra[q].y == n && ra[q].w >= y0 &&
          ra[q].z < y0 + TILE && rb[q].y >= x0 && rb[q].x < x0 + TILE;
      bal[q] = __ballot(hit[q]);
    }
    __syncthreads();  // previous chunk's readers of list / geo / wave_cnt are done
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < ROUNDS; q++) {
        const int slot = q * (NT / 64) + (tid >> 6);
        if (slot < LCH / 64) S.wave_cnt[slot] = __builtin_popcountll(bal[q]);
      }
    }
    __syncthreads();
    {
      int run = 0;
      constexpr int NSLOT = LCH / 64;  // waves of candidates, in ROI order
#pragma unroll
      for (int sl = 0; sl < NSLOT; sl++) {
        const int q = sl / (NT / 64), w = sl % (NT / 64);
        if (q < ROUNDS && w == (tid >> 6) && hit[q < ROUNDS ? q : 0])
          S.list[run + __builtin_popcountll(bal[q < ROUNDS ? q : 0] & ((1ull << lane) - 1ull))] =
              kbase + q * NT + tid;
        run += S.wave_cnt[sl];
      }
      nlist = run;
    }
    }  // !prelist
    if (L.ablate & 4) nlist = 0;
    if (wst) { wst_list = wall_clock64(); wst_n += nlist; }
    STAMP();
    if (nlist == 0) continue;  // uniform
    __syncthreads();           // list complete
    if (!prelist) {
      for (int i = tid; i < nlist; i += NT) S.geo[i] = rec[S.list[i]].g;  // geometry of the hits -> LDS
      __syncthreads();
    }

    // ---- (2)+(3) per group: software pipeline over its list entries grp, grp + GROUPS, ... ------
    const int iters = (nlist + GROUPS - 1) / GROUPS;
    if (grp < nlist) compute_weights(grp, 0);
    __syncthreads();
    STAMP();
    for (int it = 0; it < iters; it++) {
      const int li = grp + it * GROUPS, wb = it & 1;
      const bool active = li < nlist;
      const bool next_active = li + GROUPS < nlist;
      uint32_t yu = 0;  // bins that touch this thread's TR rows
#pragma unroll
      for (int i = 0; i < TR; i++) yu |= S.ymask[grp][wb][rh * TR + i];
      uint32_t xb = (active && cg_ok && !(L.ablate & 1)) ? S.xmask[grp][wb][col] : 0u;
      if (!active) yu = 0;
      const T* gk = gout + (long)S.list[active ? li : 0] * PH * PW * C + cofs;
      bool weights_done = false;
      // batches of NB = (NB/2 bins of this thread's rows) x (2 bins of its column)
      uint32_t yb = yu;
      do {  // at least once, so that the next ROI's weights are always computed
        int phs[NB / 2], npy = 0;
#pragma unroll
        for (int u = 0; u < NB / 2; u++) {
          phs[u] = yb ? __builtin_ctz(yb) : 0;
          if (yb) { npy++; yb &= yb - 1; }
        }
        uint32_t xq = xb;
        do {
          int pws[2];
          pws[0] = xq ? __builtin_ctz(xq) : 0;
          const bool okA = xq != 0;
          if (xq) xq &= xq - 1;
          pws[1] = xq ? __builtin_ctz(xq) : pws[0];
          const bool okB = xq != 0;
          if (xq) xq &= xq - 1;
          raw16 raw[NB];
          if constexpr (VEC > 1) {
#pragma unroll
            for (int u = 0; u < NB / 2; u++) {  // unconditional (clamped) loads: all NB in flight
              raw[u] = *reinterpret_cast<const raw16*>(gk + ((long)phs[u] * PW + pws[0]) * C);
              raw[NB / 2 + u] = *reinterpret_cast<const raw16*>(gk + ((long)phs[u] * PW + pws[1]) * C);
            }
          } else {
#pragma unroll
            for (int u = 0; u < NB / 2; u++) {
              raw[u] = raw16{__float_as_uint(to_f32(gk[((long)phs[u] * PW + pws[0]) * C])), 0u, 0u, 0u};
              raw[NB / 2 + u] = raw16{__float_as_uint(to_f32(gk[((long)phs[u] * PW + pws[1]) * C])), 0u, 0u, 0u};
            }
          }
          if (!weights_done) {  // uniform; overlaps the loads above
            STAMP();
            if (next_active) compute_weights(li + GROUPS, wb ^ 1);
            weights_done = true;
            STAMP();
          }
          const float wxA = okA ? S.Wx[grp][wb][col][pws[0]] : 0.f;
          const float wxB = okB ? S.Wx[grp][wb][col][pws[1]] : 0.f;
#pragma unroll
          for (int u = 0; u < NB / 2; u++) {
            if (u >= npy) break;  // uniform
            float wy[TR];
#pragma unroll
            for (int i = 0; i < TR; i += 4) {
              const float4 w4v = *reinterpret_cast<const float4*>(&S.WyT[grp][wb][phs[u]][rh * TR + i]);
              wy[i] = w4v.x; wy[i + 1] = w4v.y; wy[i + 2] = w4v.z; wy[i + 3] = w4v.w;
            }
            float fA[VEC], fB[VEC];
            if constexpr (VEC > 1) {
              unpack16(raw[u], fA, T{});
              unpack16(raw[NB / 2 + u], fB, T{});
            } else {
              fA[0] = __uint_as_float(raw[u].x);
              fB[0] = __uint_as_float(raw[NB / 2 + u].x);
            }
#pragma unroll
            for (int i = 0; i < TR; i++) {
              if (wy[i] != 0.f) {  // rows are shared by the whole wave: uniform branch
                const float a = wy[i] * wxA, b2 = wy[i] * wxB;
#pragma unroll
                for (int c = 0; c < VEC; c++) acc[i][c] += a * fA[c] + b2 * fB[c];
              }
            }
          }
        } while (xq);
      } while (yb);
      __syncthreads();  // weights[wb ^ 1] complete; everyone is done with weights[wb]
      STAMP();
    }
  }
  STAMP();
#ifdef D2AMD_PROFILE
  if (dbg_on) { L.dbg[127] = dbg_n; L.dbg[126] = wall_clock64() - rt0; L.dbg[125] = __builtin_readcyclecounter() - L.dbg[0]; }
#endif
  if (wst) { wst[1] = wst_list; wst[2] = wall_clock64(); wst[4] = (unsigned long long)wst_n; }
  // ---- partial tiles of groups 1.. are added to group 0 in a fixed order (deterministic) -------
  if constexpr (GROUPS > 1) {
    __shared__ float part[GT][TR * VEC + 1];
    for (int g = 1; g < GROUPS; g++) {
      __syncthreads();
      if (grp == g) {
#pragma unroll
        for (int i = 0; i < TR; i++)
#pragma unroll
          for (int c = 0; c < VEC; c++) part[t][i * VEC + c] = acc[i][c];
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int i = 0; i < TR; i++)
#pragma unroll
          for (int c = 0; c < VEC; c++) acc[i][c] += part[t][i * VEC + c];
      }
    }
  }
  // ---- write the tile: every pixel of grad_input exactly once ---------------------------------
  if (grp == 0 && cg_ok && x0 + col < W) {
    T* gi = (T*)L.data[lvl] + (((long)n * H + y0 + rh * TR) * W + x0 + col) * C + cofs;
#pragma unroll
    for (int i = 0; i < TR; i++) {
      if (y0 + rh * TR + i >= H) break;
      T* o = gi + (long)i * W * C;
      if constexpr (VEC > 1) {
        *reinterpret_cast<raw16*>(o) = pack16(acc[i], T{});
      } else {
        o[0] = from_f32<T>(acc[i][0]);
      }
    }
  }
  if (wst) wst[3] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// BACKWARD, NHWC: tile gather with the dY WINDOW STAGED IN LDS (v8).
//
// What the per-phase stamps of the kernel above say (profiles/r01/v8_pool_bwd_phase_stamps.txt): of the ~7,000
// cycles a tile workgroup spends per ROI, ~4,500 pass between the barrier and the moment its gather loads are
// issued.  Every lane asks for the dY vectors of "its" (ph, pw) pairs itself: the 16 threads that own the other
// pixel columns / row halves of the tile request the same vectors again, and the batches are padded to NB
// clamped loads, so a 512-thread group pushes 64 wave-loads of 1 KB through the CU's one texture-address path
// (64 B / clk: >= 1,000 cycles of pure issue) for ~8 KB of distinct data, and waits an L2 round trip for them.
// Here the group loads the WINDOW of bins that touch the tile (contiguous [ph_lo, ph_hi] x [pw_lo, pw_hi]:
// typically 3 x 3 ... 4 x 4 of the 7 x 7 / 14 x 14) ONCE, one 16-B load per thread and 16 bins, into LDS, one
// item ahead of the FMAs, and the lanes read their vectors from LDS (128 B / clk, ~100 cycles).
//   * item = (list entry, chunk of window rows holding <= WINCAP bins); almost always one item per ROI;
//   * one barrier per item: loads of item i + 1 are issued right after it, the axis weights of entry e + 2 are
//     computed while they fly (weights are TRIPLE buffered, the window of e + 1 must be known to issue its loads),
//     then the FMAs of item i run from LDS buffer i & 1, then the loaded vectors go to buffer (i + 1) & 1;
//   * one workgroup shape (512 threads) for all levels: a single launch, no side stream, tiles dealt by the
//     work queues (heavy first); long lists no longer need the list split (GROUPS) because an item costs less.
constexpr int WINCAP = 32;  // bins staged per item: 2 loads per thread
template <typename T>
struct StagedShared {
  int list[LCH];
  HitGeo geo[LCH];
  // axis weights of NSLOT = 3 * EPR list entries (EPR = 32 / PB entries are evaluated per round, PB = bins per
  // axis rounded up to 8 / 16 / 32):  WyT[(slot * PB + bin) * 8 + tile row],  Wx[(slot * 8 + tile col) * PB + bin]
  // (carries 1 / count)
  float WyT[3 * MAXP * TILE];
  float Wx[3 * MAXP * TILE];
  uint32_t ymask[12][TILE], xmask[12][TILE];
  uint32_t yall[12][4], xall[12][4];  // per weights wave: union of its rows' / columns' masks
  raw16 D[2][WINCAP][LPP];   // staged dY vectors: [buffer][bin of the item][channel lane]
  int wave_cnt[LCH / 64];
};
struct Window { int ph_lo, nph, pw_lo, npw, rpc, nitems; float rnpw; };

// PB = bins per axis rounded up to 8 / 16 / 32 (compile time: the slot / wave arithmetic folds to shifts, and the
// 7x7 and 14x14 poolers show up as separate kernels in rocprofv3)
template <typename T, int VEC, int PB>
__global__ __launch_bounds__(2 * CT, 4) void pool_bwd_staged_kernel(PoolLevels L, const RoiRec* __restrict__ rec,
                                                                   const T* __restrict__ gout, int nslab,
                                                                   int total_blocks, PoolTileIds ids) {
  constexpr int NT = 2 * CT, TR = TILE / 2;
  __shared__ StagedShared<T> S;
  const int tid = threadIdx.x, lane = tid & 63;
  int logical, slab, tile, qcnt = -1;
  if (L.queue) {
    const int j = (int)(blockIdx.x >> 3);
    const int e = L.queue[(long)(blockIdx.x & 7) * L.qcap + j / nslab].x;  // (lists are split for the MFMA kernel only)
    if (e < 0) return;
    logical = (int)blockIdx.x;
    slab = j % nslab;
    tile = e & 0xffffff;
    qcnt = (int)((unsigned)e >> 24);
  } else {
    const int per_xcd = (total_blocks + 7) >> 3;
    logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= total_blocks) return;
    slab = logical % nslab;
    tile = logical / nslab;
  }
  unsigned long long* wst = (L.wgstamps && tid == 0) ? L.wgstamps + 5 * (size_t)logical : nullptr;
  if (wst) wst[0] = wall_clock64();
  unsigned long long wst_list = 0;
  int wst_n = 0;
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < POOL_MAX_LEVELS; l++)
    if (l < L.num_levels && tile >= L.tile_base[l]) lvl = l;
  const int H = L.H[lvl], W = L.W[lvl];
  const int tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE;
  int tl = tile - L.tile_base[lvl];
  const int n = tl / (tiles_y * tiles_x);
  tl -= n * tiles_y * tiles_x;
  const int y0 = (tl / tiles_x) * TILE, x0 = (tl % tiles_x) * TILE;
  const int C = L.C, PH = L.PH, PW = L.PW, K = L.K;
  const int CG = C / VEC;
  const int rh = tid / CT, col = (tid >> 5) & 7, lp = tid & 31;  // row half / pixel column / channel lane
  const int cg = slab * LPP + lp;
  const bool cg_ok = cg < CG;
  const long cofs = (long)min(cg, CG - 1) * VEC;
  const int sb = tid >> 5;  // staging: this thread moves bins sb and sb + 16 of an item (channel lane lp)
#undef STAMP
#ifdef D2AMD_PROFILE
  const bool dbg_on = L.dbg && logical == L.dbg_block && tid == 0;
  int dbg_n = 0;
#define STAMP() do { if (dbg_on && dbg_n < 120) L.dbg[dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP() do {} while (0)
#endif
  STAMP();

  float acc[TR][VEC];
#pragma unroll
  for (int i = 0; i < TR; i++)
#pragma unroll
    for (int c = 0; c < VEC; c++) acc[i][c] = 0.f;

  // Axis weights.  PB = bins per axis rounded up to 8 / 16 / 32; one entry needs 8 x PB (row, bin) and 8 x PB
  // (column, bin) pairs = 2 * PB / 8 waves, so the 8 waves of the group evaluate EPR = 32 / PB list entries per
  // ROUND, all at the same time (box head: 4 entries, one wave per entry and axis).  Weights live in NSLOT = 3 EPR
  // slots: round r + 2 is evaluated during the first item of round r and overwrites round r - 1.
  constexpr int lg = PB == 8 ? 3 : PB == 16 ? 4 : 5;
  constexpr int EPR = 32 >> lg, NSLOT = 3 * EPR;
  const int wave = tid >> 6;
  int nlist = 0;
  auto compute_round = [&](int first) __attribute__((always_inline)) {
    constexpr int rows_per_wave = 64 >> lg, waves_per_axis = TILE >> (6 - lg);  // 8,1 / 4,2 / 2,4
    constexpr int wpe = 2 * waves_per_axis;                                      // waves per entry: 2 / 4 / 8
    const int li = first + wave / wpe;
    if (li >= nlist) return;  // uniform per wave
    const int slot = li % NSLOT;
    const int w2 = wave % wpe;
    const HitGeo g = S.geo[li];
    const bool is_x = w2 >= waves_per_axis;
    const int wa = is_x ? w2 - waves_per_axis : w2;  // wave within its axis
    const int p = lane & (PB - 1), r = wa * rows_per_wave + (lane >> lg);
    const int grid = is_x ? (g.grid >> 16) : (g.grid & 0xffff);
    const int P = is_x ? PW : PH, size = is_x ? W : H, pix = (is_x ? x0 : y0) + r;
    float wv = 0.f;
    if (p < P && pix < size) wv = axis_weight(is_x ? g.start_w : g.start_h, is_x ? g.bin_w : g.bin_h, grid, p, pix, size);
    if (is_x) S.Wx[((slot << 3) + r) * PB + p] = wv * g.inv;
    else S.WyT[((slot << lg) + p) * TILE + r] = wv;
    const unsigned long long bm = __ballot(wv != 0.f);
    if (lane < rows_per_wave) {
      const uint32_t m = (uint32_t)(bm >> (lane << lg)) & (PB == 32 ? 0xffffffffu : ((1u << PB) - 1u));
      if (is_x) S.xmask[slot][wa * rows_per_wave + lane] = m;
      else S.ymask[slot][wa * rows_per_wave + lane] = m;
    }
    if (lane == 0) {
      uint32_t u = 0;
      for (int q = 0; q < rows_per_wave; q++) u |= (uint32_t)(bm >> (q << lg)) & (PB == 32 ? 0xffffffffu : ((1u << PB) - 1u));
      if (is_x) S.xall[slot][wa] = u;
      else S.yall[slot][wa] = u;
    }
  };
  // window of bins of entry buffer wb that touch the tile (uniform; valid after the barrier that follows its weights)
  auto window_of = [&](int wb) __attribute__((always_inline)) {
    constexpr int waves_per_axis = PB / 8;
    uint32_t ya = S.yall[wb][0], xa = S.xall[wb][0];
    if (waves_per_axis > 1) { ya |= S.yall[wb][1]; xa |= S.xall[wb][1]; }
    if (waves_per_axis > 2) { ya |= S.yall[wb][2] | S.yall[wb][3]; xa |= S.xall[wb][2] | S.xall[wb][3]; }
    Window w;
    if (ya == 0 || xa == 0) { w.ph_lo = 0; w.nph = 0; w.pw_lo = 0; w.npw = 1; w.rpc = WINCAP; w.nitems = 1; w.rnpw = 1.f; return w; }
    w.ph_lo = __builtin_ctz(ya); w.nph = 32 - __builtin_clz(ya) - w.ph_lo;
    w.pw_lo = __builtin_ctz(xa); w.npw = 32 - __builtin_clz(xa) - w.pw_lo;
    w.rnpw = __builtin_amdgcn_rcpf((float)w.npw);
    w.rpc = (int)((WINCAP + 0.5f) * w.rnpw);  // WINCAP / npw;  npw <= MAXP = 32 = WINCAP: at least one row of bins
    w.nitems = (int)((w.nph + w.rpc - 0.5f) * __builtin_amdgcn_rcpf((float)w.rpc));  // ceil(nph / rpc)
    return w;
  };
  // issue the loads of item (entry li with window w, chunk c): bins sb and sb + 16 of the chunk, channel lane lp
  auto issue_loads = [&](int li, const Window& w, int c, raw16& r0, raw16& r1) __attribute__((always_inline)) {
    const int pa = w.ph_lo + c * w.rpc;
    const int nb = min(w.rpc, w.ph_lo + w.nph - pa) * w.npw;  // bins of this item (<= WINCAP); 0 for an empty window
    const T* gk = gout + (long)S.list[li] * PH * PW * C + cofs;
    if (nb > 0) {  // uniform
      const int j0 = min(sb, nb - 1);
      const int q0 = (int)((j0 + 0.5f) * w.rnpw);  // j0 / npw
      r0 = *reinterpret_cast<const raw16*>(gk + ((long)(pa + q0) * PW + w.pw_lo + (j0 - q0 * w.npw)) * C);
      if (nb > 16) {  // uniform
        const int j1 = min(sb + 16, nb - 1);
        const int q1 = (int)((j1 + 0.5f) * w.rnpw);
        r1 = *reinterpret_cast<const raw16*>(gk + ((long)(pa + q1) * PW + w.pw_lo + (j1 - q1 * w.npw)) * C);
      }
    }
    return nb;
  };

  int tl_cnt = -1;
  if (L.tile_cnt) {
    const int gtile = ids.first[lvl] + (tile - L.tile_base[lvl]);
    const int c = qcnt >= 0 ? qcnt : tile_count_of(L.tile_cnt[gtile]);
    if (c <= TILE_CAP) {
      tl_cnt = c;
      if (tid < c) {
        const TileEntry e = ((const TileEntry*)L.tile_list)[(long)gtile * TILE_CAP + tid];
        S.list[tid] = e.roi;
        S.geo[tid] = e.g;
      }
    }
  }
  const bool prelist = tl_cnt >= 0;  // uniform
  for (int kbase = 0; kbase < (prelist ? 1 : K); kbase += LCH) {
    nlist = 0;
    if (prelist) {
      nlist = tl_cnt;
    } else {
      // ordered list (+ geometry) of the ROIs of this chunk of records that touch the tile (tiles with more
      // than TILE_CAP ROIs, or no prepared lists)
      const int kend = min(K, kbase + LCH);
      const long r = min(kbase + tid, K - 1);
      const int4 ra = *reinterpret_cast<const int4*>(&rec[r].level);  // level, batch, fy0, fy1
      const int2 rb = *reinterpret_cast<const int2*>(&rec[r].fx0);    // fx0, fx1
      const bool hit = kbase + tid < kend && ra.x == lvl && ra.y == n && ra.w >= y0 && ra.z < y0 + TILE &&
          rb.y >= x0 && rb.x < x0 + TILE;
      const unsigned long long bal = __ballot(hit);
      __syncthreads();  // previous chunk's readers of list / geo / wave_cnt / weights / D are done
      if (lane == 0) S.wave_cnt[tid >> 6] = __builtin_popcountll(bal);
      __syncthreads();
      int run = 0;
#pragma unroll
      for (int sl = 0; sl < NT / 64; sl++) {
        if (sl == (tid >> 6) && hit) S.list[run + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = kbase + tid;
        run += S.wave_cnt[sl];
      }
      nlist = run;
    }
    if (wst) { wst_list = wall_clock64(); wst_n += nlist; }
    if (nlist == 0) continue;  // uniform
    __syncthreads();           // list complete
    if (!prelist) {
      for (int i = tid; i < nlist; i += NT) S.geo[i] = rec[S.list[i]].g;
      __syncthreads();
    }

    // ---- pipeline over the items of the list --------------------------------------------------------
    compute_round(0);
    compute_round(EPR);
    __syncthreads();
    raw16 r0 = raw16{0u, 0u, 0u, 0u}, r1 = raw16{0u, 0u, 0u, 0u};
    Window wc = window_of(0);
    {
      const int nb = issue_loads(0, wc, 0, r0, r1);
      if (sb < nb) S.D[0][sb][lp] = r0;
      if (sb + 16 < nb) S.D[0][sb + 16][lp] = r1;
    }
    int e = 0, c = 0, db = 0;
    while (true) {
      STAMP();
      __syncthreads();  // D[db] and the weights of e (and e + 1) are complete; everyone is done with D[db ^ 1]
      STAMP();
      // next item: the next chunk of this entry's window, or the first chunk of the next entry
      int e2 = e, c2 = c + 1;
      Window wn = wc;
      if (c2 >= wc.nitems) { e2 = e + 1; c2 = 0; }
      const bool have_next = e2 < nlist;
      int nb2 = 0;
      if (have_next) {
        if (e2 != e) wn = window_of(e2 % NSLOT);
        nb2 = issue_loads(e2, wn, c2, r0, r1);
      }
      STAMP();
      if (c == 0 && (e & (EPR - 1)) == 0) compute_round(e + 2 * EPR);  // overlaps the loads
      STAMP();
      // FMAs of item (e, c) from D[db]
      {
        const int wb = e % NSLOT;
        const int pa = wc.ph_lo + c * wc.rpc;
        const int nrow = min(wc.rpc, wc.ph_lo + wc.nph - pa);
        uint32_t yu = 0;
#pragma unroll
        for (int i = 0; i < TR; i++) yu |= S.ymask[wb][rh * TR + i];
        yu &= nrow > 0 ? (((nrow >= 32 ? 0u : (1u << nrow)) - 1u) << pa) : 0u;
        const uint32_t xb = (cg_ok && !(L.ablate & 1)) ? S.xmask[wb][col] : 0u;
        while (yu) {  // uniform per wave (a wave holds one row half)
          const int ph = __builtin_ctz(yu);
          yu &= yu - 1;
          const float4 w4v = *reinterpret_cast<const float4*>(&S.WyT[((wb << lg) + ph) * TILE + rh * TR]);
          const float wy[TR] = {w4v.x, w4v.y, w4v.z, w4v.w};
          const int dbase = (ph - pa) * wc.npw - wc.pw_lo;
          // separable: first the column taps of this bin row, t = sum_pw Wx[col][pw] dY[ph][pw], then the rows
          float t[VEC];
#pragma unroll
          for (int q = 0; q < VEC; q++) t[q] = 0.f;
          uint32_t xq = xb;
          while (xq) {
            const int pwA = __builtin_ctz(xq);
            xq &= xq - 1;
            const bool okB = xq != 0;
            const int pwB = okB ? __builtin_ctz(xq) : pwA;
            xq &= xq - 1;
            const raw16 va = S.D[db][dbase + pwA][lp], vb = S.D[db][dbase + pwB][lp];
            const float* wxr = &S.Wx[((wb << 3) + col) * PB];
            const float wxA = wxr[pwA];
            const float wxB = okB ? wxr[pwB] : 0.f;
            float fA[VEC], fB[VEC];
            unpack16(va, fA, T{});
            unpack16(vb, fB, T{});
#pragma unroll
            for (int q = 0; q < VEC; q++) t[q] += wxA * fA[q] + wxB * fB[q];
          }
#pragma unroll
          for (int i = 0; i < TR; i++) {
            if (wy[i] != 0.f) {  // rows are shared by the whole wave: uniform branch
#pragma unroll
              for (int q = 0; q < VEC; q++) acc[i][q] += wy[i] * t[q];
            }
          }
        }
      }
      STAMP();
      if (!have_next) break;
      if (sb < nb2) S.D[db ^ 1][sb][lp] = r0;
      if (sb + 16 < nb2) S.D[db ^ 1][sb + 16][lp] = r1;
      e = e2; c = c2; wc = wn; db ^= 1;
    }
  }
#ifdef D2AMD_PROFILE
  if (dbg_on) { L.dbg[127] = dbg_n; L.dbg[126] = 0; L.dbg[125] = __builtin_readcyclecounter() - L.dbg[0]; }
#endif
  if (wst) { wst[1] = wst_list; wst[2] = wall_clock64(); wst[4] = (unsigned long long)wst_n; }
  // ---- write the tile: every pixel of grad_input exactly once ---------------------------------
  if (cg_ok && x0 + col < W) {
    T* gi = (T*)L.data[lvl] + (((long)n * H + y0 + rh * TR) * W + x0 + col) * C + cofs;
#pragma unroll
    for (int i = 0; i < TR; i++) {
      if (y0 + rh * TR + i >= H) break;
      if (L.accumulate) {  // uniform: grad = round(held + round(own)), what autograd's add of two gradients gives
        float held[VEC], own[VEC];
        unpack16(*reinterpret_cast<const raw16*>(gi + (long)i * W * C), held, T{});
        unpack16(pack16_staged(acc[i], T{}), own, T{});
#pragma unroll
        for (int q = 0; q < VEC; q++) acc[i][q] = held[q] + own[q];
      }
      *reinterpret_cast<raw16*>(gi + (long)i * W * C) = pack16_staged(acc[i], T{});
    }
  }
  if (wst) wst[3] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// 16-bit I/O: operand types of the matrix-core tile gather below.
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 pbf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 pf16x8_t;
__device__ __forceinline__ f32x16_t pool_mma(s16x8_t a, s16x8_t b, f32x16_t c, bf16_t) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pbf16x8_t, a), __builtin_bit_cast(pbf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t pool_mma(s16x8_t a, s16x8_t b, f32x16_t c, f16_t) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pf16x8_t, a), __builtin_bit_cast(pf16x8_t, b), c, 0, 0, 0);
}

// PAIRED launch (d2amd_roi_pooler_backward_pair): a SECOND pooler of the same feature maps (Mask R-CNN: the mask head's
// 14 x 14 pooler behind the box head's 7 x 7) is binned TOGETHER with the first one -- records [0, K1) are the first
// pooler's ROIs, [K1, K1 + K2) the second one's, a tile has ONE list -- and gathered by ONE launch: one queue take, one
// prologue and one write of the tile for both, where two launches paid each of them twice and the second one read the
// tile back to add to it.
struct PoolPairArgs {
  const void* gout;     // the second pooler's dY [K2][PH][PW][C]
  const int* tile_cnt1; // per tile: entries of the FIRST pooler in its list (the rest are the second one's)
  int K1, K2, PH, PW;
};

// ------------------------------------------------------------------------------------------------
// BACKWARD, NHWC, 16-bit I/O: the K-CONCATENATED tile gather on the matrix cores (r06).
//
// Rounds 3-5 walked a tile's ROI list item by item -- per (tile, ROI) item one exposed global round trip, one barrier
// and ~2,700 cycles of bookkeeping for four MFMAs (profiles/r03/pool_bwd/README.md: items 6.3 us of a 10.8 us tile; the
// paired launch 78-85 us).  The gradient of a tile is ONE contraction, though:
//     G[64 px][C] = W[64 px][K] . D[K][C]      k = every (list entry, bin of its window) pair of the tile
// -- box-head and mask-head entries of a paired launch alike: an entry only decides which dY row a k reads and which
// axis weights form its column of W = Wy[row][ph] Wx[col][pw] / count.  The binning kernel hands every list entry its
// WINDOW of bins (axis_window: conservative, a bin too many only carries zero weights), so the rows to fetch are known
// before any weight is.  A tile is processed in ROUNDS of list entries and BATCHES of KCAP k's:
//   1. wave 0 lays out the round (lane = entry): k offsets (prefix of the window sizes, on the DPP network), dY row of
//      the window's first bin, offsets of the entry's axis weights in the pool;
//   2. the K TABLE, thread = k (one 5-step search in the k offsets): address of the k's dY row, byte offsets of its Wy /
//      Wx rows in the pool;
//   3. every dY row of the batch is requested at once by LDS-DMA (global_load_lds, 16 B per lane: two 512-B rows per
//      wave instruction, no staging registers) -- ONE exposed round trip per batch instead of one per item, issued
//      BEFORE the weights are evaluated;
//   4. while the rows fly: the axis weights of the windows (a wave per entry) and the weight image W of the batch as a
//      HIGH and a LOW 16-bit part (w = hi + lo keeps 16 significant bits; fp32 accumulation), lane = pixel;
//   5. acc += (Whi + Wlo) . D with v_mfma_f32_32x32x16: A = the staged dY through ds_read_b64_tr_b16 (the hardware 4 x 4
//      transpose: lane i of a 16-lane group addresses row i >> 2, columns 4 (i & 3) .. + 3 of a [4 k][16 channels] block
//      and receives column i -- scripts/probes/probe_tr16.hip), B = the weight image; wave w owns channels
//      [256 w / NW, 256 (w + 1) / NW) of the slab for all 64 pixels; rows past the batch are zero-filled (0 x stale NaN);
//   6. the epilogue transposes the accumulators through LDS: every pixel is written once as 16-B channel vectors.
// The NEXT tile is fetched during the current one by wave 0: the take on the queue counter right behind the tile's
// first barrier, the queue slot behind the first batch, the tile's geometry and first list entries (into registers)
// before the epilogue -- three of the four dependent round trips of a tile (counter -> slot -> list -> dY rows) leave the
// critical path.  (Two tiles ahead was measured too: 56.8 against 50 us -- a workgroup then sits on two tiles nobody
// else can take, and the launch ends as late as its slowest workgroup.)
// PERSISTENT workgroups: the grid is one resident wave of them, each fetching tile after tile from the queue of its own
// XCD (tile_lists_kernel: heavy tiles first).  SPLIT lists: a part leaves its fp32 accumulators in a scratch slot and
// takes a ticket, the workgroup that draws the last one adds the parts IN PART ORDER and writes the tile (device-scope
// relaxed atomics, nobody waits).  Tiles whose list exceeds TILE_CAP scan the records in chunks into an LDS hit buffer
// and feed the same rounds.  fp32 I/O keeps the VALU kernel above (fp32 MFMA runs at the vector rate).
// Measurements, what was tried and what the counters say: profiles/r06/pool_bwd_kcat.md.
// two fp32 -> one dword of two 16-bit values (hardware conversion, round to nearest even), and back
typedef float kc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t kc_pack2(float a, float b, bf16_t) {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(kc_f2{a, b}, v2));
}
__device__ __forceinline__ uint32_t kc_pack2(float a, float b, f16_t) {
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(kc_f2{a, b}, v2));
}
__device__ __forceinline__ kc_f2 kc_unpack2(uint32_t u, bf16_t) {
  return kc_f2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}
__device__ __forceinline__ kc_f2 kc_unpack2(uint32_t u, f16_t) {
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  return __builtin_convertvector(__builtin_bit_cast(v2, u), kc_f2);
}
// the HIGH part of a weight pair: bf16 by truncation (the upper halves of the two words, one v_perm; the low part
// = w - hi is exact in fp32 and rounded to nearest: hi + lo carries 16 significant bits either way), fp16 by conversion
__device__ __forceinline__ uint32_t kc_hi2(float a, float b, bf16_t) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ uint32_t kc_hi2(float a, float b, f16_t t) { return kc_pack2(a, b, t); }

template <int NW, int KCAP>
struct __attribute__((aligned(16))) KcShared {
  static constexpr int NT = 64 * NW, ECAP = 32, WPOOL = 1536, WP = KCAP + 8, TK = NT / KCAP * KCAP;
  char D[KCAP * 512];         // staged dY rows [k][256 channels], 16-bit; the epilogue's [64 px][256 ch] image lies
                              // over D + Whi + Wlo (+ wpool)
  uint16_t Whi[64][WP], Wlo[64][WP];  // weight image [pixel][k] (WP: conflict-free 16-B reads)
  float wpool[WPOOL];         // axis weights of the round: per entry [nph][8 rows], then [npw][8 cols] (x carries 1 / count)
  int4 ktab[NT];              // per k of the table: {address of the dY row's slab (lo, hi), Wy | Wx << 16 byte offsets in
                              // the pool, -}
  HitGeo geo[ECAP];
  int row0[ECAP];             // dY row of the window's first bin
  int woff[64];               // pool offset                                  } entries past the round: INT_MAX (the
  int pref[64];               // k offset (exclusive prefix of the window sizes) } searches read them unguarded)
  int dims[ECAP];             // nph | npw << 8 | PW << 16 | second pooler << 24
  int org[ECAP];              // ph_lo | pw_lo << 8
  int hits[ECAP + NT];        // record indices found by the in-kernel scan, in order
  int wcnt[NW];
  int ctl[16];                // tile: queue entry, slab, logical id, ticket, part info; round: entries [5], pool floats,
                              // k's; tile geometry (by wave 0, beside the previous tile): level [8], image, y0, x0, H, W
  int nbuf;
};

// (waves per SIMD the register allocation is held to = the workgroups the LDS admits per CU x NW / 4 SIMDs)
template <typename T, int NW, int KCAP>
__global__ __launch_bounds__(64 * NW, (160 * 1024 / (int)sizeof(KcShared<NW, KCAP>)) * NW / 4) void pool_bwd_kcat_kernel(
    PoolLevels L, const RoiRec* __restrict__ rec, const T* __restrict__ gout0, int nslab, PoolPairArgs P2) {
  using SH = KcShared<NW, KCAP>;
  constexpr int NT = SH::NT, NH = 8 / NW, CHW = 32 * NH, ECAP = SH::ECAP, WPOOL = SH::WPOOL, TK = SH::TK, VEC = 8;
  constexpr int RPT = TILE * 4 / NW;  // tile rows a thread stores (its pixel column, 16 B of channels)
  static_assert(NW == 4 || NW == 8, "waves per workgroup");
  static_assert(KCAP % 16 == 0 && KCAP * 512 + 4 * 64 * SH::WP + 4 * WPOOL >= 64 * 256 * 2, "the epilogue image lies over D + W + pool");
  static_assert(TK % 4 == 0 && KCAP % 4 == 0, "the weight image reads 4 table entries at once");
  static_assert(ECAP <= 64 && WPOOL * 4 < (1 << 15) && WPOOL >= 8 * 64, "entries: one wave; pool byte offsets: 15 bits; an entry fits");
  __shared__ SH S;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K1 = P2.gout ? P2.K1 : L.K;  // records [0, K1): the first pooler's ROIs, [K1, L.K): the second one's
  const int C = L.C, CG = C / VEC;
  const int myq = (int)blockIdx.x & 7, G = (int)(gridDim.x >> 3);
  const int nh = L.qctr[myq], nl = L.qctr[16 + myq];  // final: tile_lists_kernel is an earlier launch
#ifdef D2AMD_PROFILE
  int dbg_n = 0;
  bool dbg_on = false;
#define STAMP() do { if (dbg_on && dbg_n < 120) L.dbg[dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP() do {} while (0)
#endif

  // ---- wave 0: the queue and the next tile's list
  int nx_qe = -1, nx_pinfo = 1 << 8, nx_sl = 0, nx_lg = 0;  // the NEXT tile (uniform in wave 0)
  int4 nx_geo = int4{0, 0, 0, 0};                            // its geometry (tile_lists_kernel)
  uint4 pe0 = uint4{0u, 0u, 0u, 0u}, pe1 = pe0;             // its first round's list entries, lane = entry
  int take_i = (int)(blockIdx.x >> 3);  // entry of the home queue to fetch next (the first one without a take: 64
                                        // workgroups hitting one counter in the same microsecond are served in turn)
  int2 slot_v = int2{-1, 0};
  bool slot_pending = false;
  // queue entry `take_i` -> slot_v (a load in flight), nx_sl / nx_lg
  auto slot_issue = [&]() __attribute__((always_inline)) {
    const int i = __builtin_amdgcn_readfirstlane(take_i);
    const int ent = i / nslab;
    slot_v = int2{-1, 0};
    if (ent < nh + nl) {
      const int slot = ent < nh ? ent : L.qcap - 1 - (ent - nh);  // heavy from the front, light from the back
      slot_v = L.queue[(long)myq * L.qcap + slot];
      nx_sl = i - ent * nslab;
      nx_lg = ((slot * nslab + nx_sl) << 3) | myq;  // the workgroup id the static mapping gives this (slot, slab)
    }
    slot_pending = true;
  };
  // slot_v -> nx_qe / nx_pinfo, and the loads of the tile's first list entries
  auto entries_issue = [&]() __attribute__((always_inline)) {
    nx_qe = __builtin_amdgcn_readfirstlane(slot_v.x);
    nx_pinfo = __builtin_amdgcn_readfirstlane(slot_v.y);
    slot_pending = false;
    if (nx_qe >= 0) {
      const int c = (int)((unsigned)nx_qe >> 24);
      nx_geo = L.tile_geo[nx_qe & 0xffffff];  // (uniform address; used at the top of the next tile)
      if (c <= TILE_CAP) {
        const int part = nx_pinfo & 0xff, parts = (nx_pinfo >> 8) & 0xff;
        const int len = (c + parts - 1) / parts, lo = part * len, hi = min(c, lo + len);
        if (lane < min(hi - lo, ECAP)) {
          const uint4* ep = reinterpret_cast<const uint4*>((const TileEntry*)L.tile_list + (long)(nx_qe & 0xffffff) * TILE_CAP + lo + lane);
          pe0 = ep[0];
          pe1 = ep[1];
        }
      }
    }
  };
  // the layout of a round, lane = entry (wave 0): k offsets, pool offsets, rows; as many entries as the pool holds
  auto layout_round = [&](bool valid, int roi, const HitGeo& g, int win) __attribute__((always_inline)) {
    const int pid = roi >= K1 ? 1 : 0;
    const int PHe = pid ? P2.PH : L.PH, PWe = pid ? P2.PW : L.PW;
    const int ph_lo = win & 0xff, nph = (win >> 8) & 0xff, pw_lo = (win >> 16) & 0xff, npw = (win >> 24) & 0xff;
    const int nb = valid ? nph * npw : 0;
    const int sz = nb > 0 ? 8 * (nph + npw) : 0;  // (an entry without bins takes no pool space)
    const int isz = wave_incl_scan(sz), inb = wave_incl_scan(nb);
    const bool fits = valid && isz <= WPOOL;  // monotone: the round is a prefix of the available entries
    const int nr = __builtin_amdgcn_readfirstlane(__builtin_popcountll(__ballot(fits)));
    S.woff[lane] = fits ? isz - sz : 0x7fffffff;
    S.pref[lane] = fits ? inb - nb : 0x7fffffff;
    if (fits) {
      S.geo[lane] = g;
      S.row0[lane] = ((roi - (pid ? K1 : 0)) * PHe + ph_lo) * PWe + pw_lo;
      S.dims[lane] = nph | (npw << 8) | (PWe << 16) | (pid << 24);
      S.org[lane] = ph_lo | (pw_lo << 8);
    }
    const int used = __builtin_amdgcn_readlane(isz, max(nr - 1, 0)), ktot = __builtin_amdgcn_readlane(inb, max(nr - 1, 0));
    if (lane == 0) { S.ctl[5] = nr; S.ctl[6] = nr ? used : 0; S.ctl[7] = nr ? ktot : 0; }
  };
  auto unpack_entry = [&](const uint4& e0, const uint4& e1, HitGeo& g, int& roi, int& win) __attribute__((always_inline)) {
    g.start_h = __uint_as_float(e0.x); g.start_w = __uint_as_float(e0.y); g.bin_h = __uint_as_float(e0.z);
    g.bin_w = __uint_as_float(e0.w); g.inv = __uint_as_float(e1.x); g.grid = (int)e1.y;
    roi = (int)e1.z; win = (int)e1.w;
  };
  if (wave == 0) {
    slot_issue();
    entries_issue();
  }

  const int tid_k = tid, lane_k = lane;
  for (int round = 0;; round++) {
    // (the thread index passes through an opaque move per tile: what is derived from it -- LDS addresses of the epilogue,
    // store offsets -- is then recomputed per tile instead of being hoisted out of this loop, held across it and, at
    // the register cap, spilled to scratch)
    int tid_l = tid_k;
    asm volatile("" : "+v"(tid_l));
    const int tid = tid_l, lane = tid & 63;
    (void)lane_k;
    // ---- wave 0 publishes the tile and, for a prepared list, its first round
    if (wave == 0) {
      if (lane == 0) {
        S.ctl[0] = nx_qe; S.ctl[1] = nx_sl; S.ctl[2] = nx_lg; S.ctl[4] = nx_pinfo;
        S.ctl[8] = nx_geo.x & 0xff; S.ctl[9] = nx_geo.x >> 8; S.ctl[10] = nx_geo.y & 0xffff; S.ctl[11] = (int)((unsigned)nx_geo.y >> 16);
        S.ctl[12] = nx_geo.z & 0xffff; S.ctl[13] = (int)((unsigned)nx_geo.z >> 16);
      }
      const int c = (int)((unsigned)nx_qe >> 24);
      if (nx_qe >= 0 && c <= TILE_CAP) {
        const int part = nx_pinfo & 0xff, parts = (nx_pinfo >> 8) & 0xff;
        const int len = (c + parts - 1) / parts, lo = part * len, hi = min(c, lo + len);
        HitGeo g; int roi, win;
        unpack_entry(pe0, pe1, g, roi, win);
        layout_round(lane < min(hi - lo, ECAP), roi, g, win);
      }
    }
    __syncthreads();  // A0
    const int qe = __builtin_amdgcn_readfirstlane(S.ctl[0]);
    if (qe < 0) return;  // the queue is empty
    const int slab = __builtin_amdgcn_readfirstlane(S.ctl[1]);
    const int logical = __builtin_amdgcn_readfirstlane(S.ctl[2]);
    const int pinfo = __builtin_amdgcn_readfirstlane(S.ctl[4]);  // part | parts << 8 | scratch slot << 16
    const int tile = qe & 0xffffff, qcnt = (int)((unsigned)qe >> 24);
    // the take for the tile after this one
    if (wave == 0 && lane == 0) take_i = G + atomicAdd(L.qctr + QTAKE + QTAKE_PITCH * myq, 1);
#ifdef D2AMD_PROFILE
    dbg_on = L.dbg && logical == L.dbg_block && tid == 0;
#endif
    STAMP();
#define KST(k, v) do { if (L.wgstamps && tid == 0) L.wgstamps[5 * (size_t)logical + (k)] = (v); } while (0)
    KST(0, wall_clock64());
    const int lvl = __builtin_amdgcn_readfirstlane(S.ctl[8]), n = __builtin_amdgcn_readfirstlane(S.ctl[9]);
    const int y0 = __builtin_amdgcn_readfirstlane(S.ctl[10]), x0 = __builtin_amdgcn_readfirstlane(S.ctl[11]);
    const int H = __builtin_amdgcn_readfirstlane(S.ctl[12]), W = __builtin_amdgcn_readfirstlane(S.ctl[13]);
    const int lp = tid & 31;          // 16-B channel group of the slab this thread loads / stores
    const int cg = slab * LPP + lp;
    const bool cg_ok = cg < CG;
    const int cofs = min(cg, CG - 1) * VEC;
    bool ch_ok[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) ch_ok[h] = slab * (LPP * VEC) + CHW * wave + 32 * h < C;

    // the part of the tile's list this workgroup walks: entries [lo, hi) of the prepared list, or (lo < 0: more than
    // TILE_CAP ROIs) the records are scanned
    int lo = -1, hi = 0;
    if (qcnt <= TILE_CAP) {
      const int part = pinfo & 0xff, parts = (pinfo >> 8) & 0xff;
      const int len = (qcnt + parts - 1) / parts;
      lo = part * len;
      hi = min(qcnt, lo + len);
    }
    const bool prelist = lo >= 0;
    const TileEntry* tlist = (const TileEntry*)L.tile_list + (long)tile * TILE_CAP;

    // wave w accumulates channels [CHW w, CHW w + CHW) of the slab: acc[mt][h] = [32 channels of half h] x [pixels of
    // tile rows 4 mt .. 4 mt + 3]; lane = pixel, a register quad = 4 consecutive channels
    f32x16_t acc[2][NH];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int h = 0; h < NH; h++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[mt][h][i] = 0.f;

    int done = prelist ? lo : 0;  // prelist: next list entry; scan: next record
    int wst_n = 0;
    bool have_round = prelist;    // (the first round of a prepared list came with the tile)
    if (!prelist && tid == 0) S.nbuf = 0;
    while (true) {
      // ---- 1. the round's entries
      if (!have_round) {
        if (prelist) {
          if (done >= hi) break;  // uniform: the list is done
          if (wave == 0) {
            const bool valid = lane < min(hi - done, ECAP);
            HitGeo g{}; int roi = 0, win = 0;
            if (valid) {
              const uint4* ep = reinterpret_cast<const uint4*>(tlist + done + lane);
              const uint4 e0 = ep[0], e1 = ep[1];
              unpack_entry(e0, e1, g, roi, win);
            }
            layout_round(valid, roi, g, win);
          }
        } else {
          __syncthreads();  // nbuf / the compacted hit buffer are visible
          while (true) {
            const int nbuf = S.nbuf;
            if (nbuf >= ECAP || done >= L.K) break;  // uniform
            const int kk = done + tid;
            const long r = min(kk, L.K - 1);
            const int4 ra = *reinterpret_cast<const int4*>(&rec[r].level);  // level, batch, fy0, fy1
            const int2 rb = *reinterpret_cast<const int2*>(&rec[r].fx0);    // fx0, fx1
            const bool hit = kk < L.K && ra.x == lvl && ra.y == n && ra.w >= y0 && ra.z < y0 + TILE && rb.y >= x0 &&
                rb.x < x0 + TILE;
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) S.wcnt[wave] = __builtin_popcountll(bal);
            __syncthreads();
            int off = nbuf, tot = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) {
              const int c2 = S.wcnt[w2];
              if (w2 < wave) off += c2;
              tot += c2;
            }
            if (hit) S.hits[off + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = kk;
            __syncthreads();  // everyone has read nbuf and the counts
            if (tid == 0) S.nbuf = nbuf + tot;
            done += NT;
            __syncthreads();
          }
          const int navail = min(S.nbuf, ECAP);
          if (navail <= 0) break;  // uniform: the records are done
          if (wave == 0) {
            const bool valid = lane < navail;
            HitGeo g{}; int roi = 0, win = 0;
            if (valid) {
              roi = S.hits[lane];
              const uint2* gp = reinterpret_cast<const uint2*>(&rec[roi].g);
              const uint2 a = gp[0], b = gp[1], c = gp[2];
              g.start_h = __uint_as_float(a.x); g.start_w = __uint_as_float(a.y); g.bin_h = __uint_as_float(b.x);
              g.bin_w = __uint_as_float(b.y); g.inv = __uint_as_float(c.x); g.grid = (int)c.y;
              const int pid = roi >= K1 ? 1 : 0;
              win = entry_window(g, pid ? P2.PH : L.PH, pid ? P2.PW : L.PW, y0, x0, H, W);
            }
            layout_round(valid, roi, g, win);
          }
        }
        __syncthreads();  // A
      }
      have_round = false;
      const int nr = (L.ablate & 64) ? 0 : S.ctl[5];
      if (nr <= 0) break;  // uniform (an empty part, or the ablation)
      const int pool_used = S.ctl[6], ktot = S.ctl[7];
      if (L.wgstamps) { if (wst_n == 0) KST(1, wall_clock64()); wst_n += nr + (ktot << 12); }  // entries | k's << 12
      STAMP();

      // entry of k (of pool item i): the last e with pref[e] (woff[e]) <= k; entries without bins are skipped over
      auto entry_of = [&](const int* __restrict__ tab, int k) __attribute__((always_inline)) {
        int e = 0;
#pragma unroll
        for (int step = 16; step > 0; step >>= 1) e += tab[e + step] <= k ? step : 0;  // (tab[e >= nr] = INT_MAX)
        return e;
      };
      for (int kb0 = 0; kb0 < ktot; kb0 += KCAP) {
        const int kb = min(KCAP, ktot - kb0), kb16 = (kb + 15) & ~15;
        const bool new_table = kb0 % TK == 0;  // uniform
        if (new_table) {
          // ---- 2. the K table of k in [kb0, kb0 + TK): thread = k, one search in the k offsets
          if (tid < TK) {  // (k past the round's last: offsets 0 -- the weight image reads the whole batch unguarded)
            const int k = min(kb0 + tid, ktot - 1);
            const int e = entry_of(S.pref, k);
            const int il = k - S.pref[e], dims = S.dims[e];
            const int nph = dims & 0xff, npw = (dims >> 8) & 0xff, PWe = (dims >> 16) & 0xff, wo = S.woff[e];
            const int q = (int)(((float)il + 0.5f) * __builtin_amdgcn_rcpf((float)npw));  // il / npw  (il < 1024)
            const int pi = il - q * npw;
            const T* rowp = ((dims >> 24) ? (const T*)P2.gout : gout0) + (size_t)(unsigned)(S.row0[e] + q * PWe + pi) * (size_t)C +
                slab * (LPP * VEC);
            const int4 t = int4{(int)(unsigned)(size_t)rowp, (int)(unsigned)((size_t)rowp >> 32),
                                ((wo + q * 8) * 4) | (((wo + 8 * nph + pi * 8) * 4) << 16), 0};  // BYTE offsets into the pool
            S.ktab[tid] = kb0 + tid < ktot ? t : int4{0, 0, 0, 0};
          }
          __syncthreads();  // C
          STAMP();
        }
        // ---- 3. every dY row of the batch by LDS-DMA: lanes 0-31 / 32-63 of an instruction fetch rows k2 / k2 + 1.
        // (Source addresses first, then the instructions back to back, as inline assembly: the compiler orders an
        // LDS-DMA it knows about -- __builtin_amdgcn_global_load_lds -- before EVERY later LDS read with s_waitcnt
        // vmcnt(0), so the first version waited out a full round trip behind each instruction and again before the
        // weights.  Untracked, the rows are waited for once, in front of barrier D.)
        if (!(L.ablate & 16)) {
          constexpr int NI = KCAP / (2 * NW);
          const int t0 = kb0 % TK;
          uint2 rows[NI];
#pragma unroll
          for (int i = 0; i < NI; i++)
            rows[i] = *reinterpret_cast<const uint2*>(&S.ktab[t0 + min((i * NW + wave) * 2 + (lane >> 5), kb - 1)]);
          const unsigned d_lds = (unsigned)(size_t)S.D;  // (the low word of a flat LDS address is the LDS offset)
#pragma unroll
          for (int i = 0; i < NI; i++) {
            const int k2 = (i * NW + wave) * 2;  // uniform
            if (k2 < kb && k2 + (lane >> 5) < kb && cg_ok) {
              const char* src = (const char*)(((size_t)rows[i].y << 32) | rows[i].x) + lp * 16;
              asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                           ::"v"(src), "s"(__builtin_amdgcn_readfirstlane(d_lds + k2 * 512)) : "memory");
            }
          }
        }
        // rows [kb, kb16): zeros (their weights are 0; stale bits might be NaN)
        for (int z = tid; z < (kb16 - kb) * 32; z += NT)
          *reinterpret_cast<uint4*>(S.D + (kb + (z >> 5)) * 512 + (z & 31) * 16) = uint4{0u, 0u, 0u, 0u};
        STAMP();
        if (kb0 == 0) {
          // ---- 4a. (first batch, while the rows fly) the axis weights of the round's windows: a WAVE per entry (its
          // geometry is wave-uniform), lane = (bin of the window, tile row / column): [nph][8], then [npw][8]
          for (int e = wave; e < nr; e += NW) {
            const HitGeo g = S.geo[e];
            const int dims = S.dims[e], org = S.org[e], wo = S.woff[e];
            const int nph = dims & 0xff, npw = (dims >> 8) & 0xff;
            const int items = nph * npw > 0 ? 8 * (nph + npw) : 0;  // (an entry without bins owns no pool space)
            for (int j = lane; j < items; j += 64) {
              const bool is_x = j >= 8 * nph;
              const int jj = is_x ? j - 8 * nph : j;
              const int p = (is_x ? (org >> 8) : (org & 0xff)) + (jj >> 3), r = jj & 7;
              const int grid = is_x ? (g.grid >> 16) : (g.grid & 0xffff);
              const int size = is_x ? W : H, pix = (is_x ? x0 : y0) + r;
              float wv = 0.f;
              if (pix < size) wv = axis_weight(is_x ? g.start_w : g.start_h, is_x ? g.bin_w : g.bin_h, grid, p, pix, size);
              S.wpool[wo + j] = is_x ? wv * g.inv : wv;
            }
          }
          STAMP();
          __syncthreads();  // B: the weights are complete
          STAMP();
        }
        // ---- 4b. the weight image of the batch: lane = pixel, a wave builds 4 consecutive k's per step.  All KCAP
        // columns, unrolled and without a branch (k past the batch: offset 0, weight selected to 0): the table / pool
        // reads of every step are in flight together -- guarded per k they ran one LDS round trip after the other.
        if (!(L.ablate & 32)) {
          const int t0 = kb0 % TK;
          const char* pool = reinterpret_cast<const char*>(S.wpool);
          const int r4 = (lane >> 3) * 4, c4 = (lane & 7) * 4;
#pragma unroll
          for (int kq0 = 0; kq0 < KCAP; kq0 += 4 * NW) {
            const int kq = kq0 + 4 * wave;
            if (kq >= KCAP) break;  // uniform (only where 4 NW does not divide KCAP)
            const int codes[4] = {S.ktab[t0 + kq].z, S.ktab[t0 + kq + 1].z, S.ktab[t0 + kq + 2].z, S.ktab[t0 + kq + 3].z};
            float wv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float wy = *reinterpret_cast<const float*>(pool + (codes[j] & 0xffff) + r4);
              const float wx = *reinterpret_cast<const float*>(pool + ((unsigned)codes[j] >> 16) + c4);
              wv[j] = kq + j < kb ? wy * wx : 0.f;
            }
            // hi = w rounded to the I/O dtype, lo = (w - hi) rounded
            const uint32_t h01 = kc_hi2(wv[0], wv[1], T{}), h23 = kc_hi2(wv[2], wv[3], T{});
            const kc_f2 f01 = kc_unpack2(h01, T{}), f23 = kc_unpack2(h23, T{});
            const uint32_t l01 = kc_pack2(wv[0] - f01.x, wv[1] - f01.y, T{}), l23 = kc_pack2(wv[2] - f23.x, wv[3] - f23.y, T{});
            *reinterpret_cast<uint2*>(&S.Whi[lane][kq]) = uint2{h01, h23};
            *reinterpret_cast<uint2*>(&S.Wlo[lane][kq]) = uint2{l01, l23};
          }
        }
        STAMP();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA rows have landed (wave 0: and the take)
        if (wave == 0 && !slot_pending) slot_issue();  // (the tile's first batch: the take has returned)
        __syncthreads();                // D
        STAMP();
        // ---- 4. acc += (Whi + Wlo) . D
        if (!(L.ablate & 1)) {
          const int kh = lane >> 5;
          for (int ks = 0; ks < kb16 / 16; ks++) {
            s16x8_t bh[2], bl[2];
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
              const int px = 32 * mt + (lane & 31);
              bh[mt] = *reinterpret_cast<const s16x8_t*>(&S.Whi[px][16 * ks + 8 * kh]);
              bl[mt] = *reinterpret_cast<const s16x8_t*>(&S.Wlo[px][16 * ks + 8 * kh]);
            }
#pragma unroll
            for (int h = 0; h < NH; h++) {
              if (!ch_ok[h]) continue;  // uniform per wave
              // tr16 address of this lane inside a [4 k][16 channels] block of the half's 32 channels
              const char* bp = S.D + (16 * ks + 8 * kh + ((lane & 15) >> 2)) * 512 +
                  (CHW * wave + 32 * h + 16 * ((lane >> 4) & 1) + (lane & 3) * 4) * 2;
              const s16x4_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)bp);
              const s16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(bp + 4 * 512));
              const s16x8_t a = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
              for (int mt = 0; mt < 2; mt++) {
                acc[mt][h] = pool_mma(a, bh[mt], acc[mt][h], T{});
                acc[mt][h] = pool_mma(a, bl[mt], acc[mt][h], T{});
              }
            }
          }
        }
        STAMP();
        __syncthreads();  // E: D, W (and, behind the last batch, the table / the round's entries) may be overwritten
        STAMP();
      }
      // ---- next round
      if (prelist) {
        done += nr;
      } else {  // the round took the first nr hits: move the others to the front
        const int rem = S.nbuf - nr;
        int v0 = 0, v1 = 0;
        if (tid < rem) v0 = S.hits[nr + tid];
        if (tid + NT < rem) v1 = S.hits[nr + tid + NT];
        __syncthreads();
        if (tid < rem) S.hits[tid] = v0;
        if (tid + NT < rem) S.hits[tid + NT] = v1;
        if (tid == 0) S.nbuf = rem;
      }
    }
    KST(2, wall_clock64());
    if (wave == 0 && !slot_pending) slot_issue();  // (a tile without a batch)

    // ---- split list: this part's accumulators go to its scratch slot; the last part to arrive adds all parts in part
    // order and writes the tile
    if (((pinfo >> 8) & 0xff) > 1) {  // uniform
      const int part = pinfo & 0xff, parts = (pinfo >> 8) & 0xff, sbase = (int)((unsigned)pinfo >> 16);
      constexpr int NA = 32 * NH;  // accumulators per thread
      const size_t slot_floats = (size_t)nslab * NA * NT;
      float* mine = L.part_scratch + ((size_t)(sbase + part) * nslab + slab) * NA * NT + tid;
#pragma unroll
      for (int i = 0; i < NA; i++)
        __hip_atomic_store(mine + i * NT, acc[(i >> 4) & 1][i >> 5][i & 15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);  // acknowledged = visible
      __syncthreads();
      if (tid == 0)
        S.ctl[3] = __hip_atomic_fetch_add(L.part_tickets + sbase * SPLIT_MAX_SLABS + slab, 1, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (S.ctl[3] != parts - 1) {  // uniform: another part finishes this tile
        if (wave == 0) entries_issue();
        KST(3, wall_clock64());
        KST(4, (unsigned long long)wst_n | (unsigned long long)blockIdx.x << 32 | 1ull << 56);
        continue;
      }
      if (tid == 0)  // re-armed for a second gather over the same binned workspace
        __hip_atomic_store(L.part_tickets + sbase * SPLIT_MAX_SLABS + slab, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float* all = L.part_scratch + ((size_t)sbase * nslab + slab) * NA * NT + tid;
#pragma unroll
      for (int i = 0; i < NA; i++) acc[(i >> 4) & 1][i >> 5][i & 15] = 0.f;
      for (int q = 0; q < parts; q++) {  // (32 loads of a part in flight together: the combine is their latency)
#pragma unroll
        for (int i0 = 0; i0 < NA; i0 += 32) {
          float v[32];
#pragma unroll
          for (int i = 0; i < 32; i++)
            v[i] = __hip_atomic_load(all + (size_t)q * slot_floats + (i0 + i) * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int i = 0; i < 32; i++) acc[(i >> 4) & 1][i0 >> 5][i & 15] += v[i];
        }
      }
    }
    // ---- epilogue: accumulators -> LDS [pixel][channel] in the I/O dtype (a pixel's 32 16-B chunks at chunk ^
    // (pixel & 31)) -> 16-B channel vectors, every pixel of grad_input written exactly once
    T* obuf = reinterpret_cast<T*>(&S);
#pragma unroll
    for (int h = 0; h < NH; h++) {
      if (!ch_ok[h]) continue;
#pragma unroll
      for (int mt = 0; mt < 2; mt++) {
        const int px = 32 * mt + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int ch = CHW * wave + 32 * h + 8 * g + 4 * (lane >> 5);  // 4 consecutive channels: half a 16-B chunk
          uint2 w;
          w.x = (uint32_t)from_f32<T>(acc[mt][h][4 * g]).v | ((uint32_t)from_f32<T>(acc[mt][h][4 * g + 1]).v << 16);
          w.y = (uint32_t)from_f32<T>(acc[mt][h][4 * g + 2]).v | ((uint32_t)from_f32<T>(acc[mt][h][4 * g + 3]).v << 16);
          *reinterpret_cast<uint2*>(obuf + px * (LPP * VEC) + (((ch >> 3) ^ (px & 31)) << 3) + (ch & 4)) = w;
        }
      }
    }
    if (wave == 0) entries_issue();  // the next tile's queue slot has arrived: its list entries fly under the stores
    STAMP();
    const int col = (tid >> 5) & 7, r0 = (tid >> 8) * RPT;  // pixel column / first tile row of this thread
    const bool mine = cg_ok && x0 + col < W && !(L.ablate & 128);
    if (L.accumulate) {
      // the rows this thread adds to are fetched now (the accumulators are dead) and fly under the barrier
      raw16 held[RPT];
      T* gi = (T*)L.data[lvl] + (((long)n * H + y0 + r0) * W + min(x0 + col, W - 1)) * C + cofs;
#pragma unroll
      for (int i = 0; i < RPT; i++)
        held[i] = *reinterpret_cast<const raw16*>(gi + (long)max(min(i, H - 1 - (y0 + r0)), -r0) * W * C);
      __syncthreads();  // F
      if (mine) {
#pragma unroll
        for (int i = 0; i < RPT; i++) {
          if (y0 + r0 + i >= H) break;
          const int px = (r0 + i) * TILE + col;
          const raw16 v = *reinterpret_cast<const raw16*>(obuf + px * (LPP * VEC) + ((lp ^ (px & 31)) * VEC));
          float a[VEC], b[VEC];  // round(held + round(own)), what autograd's add of two gradients gives
          unpack16(held[i], a, T{});
          unpack16(v, b, T{});
#pragma unroll
          for (int q = 0; q < VEC; q++) a[q] += b[q];
          *reinterpret_cast<raw16*>(gi + (long)i * W * C) = pack16(a, T{});
        }
      }
    } else {
      __syncthreads();  // F
      STAMP();
      if (mine) {
        // (all rows out of LDS first, then the stores back to back: one register set for both made the compiler wait
        // for every store before the next LDS read could overwrite its data)
        T* gi = (T*)L.data[lvl] + (((long)n * H + y0 + r0) * W + x0 + col) * C + cofs;
        raw16 v[RPT];
#pragma unroll
        for (int i = 0; i < RPT; i++) {
          const int px = (r0 + i) * TILE + col;
          v[i] = *reinterpret_cast<const raw16*>(obuf + px * (LPP * VEC) + ((lp ^ (px & 31)) * VEC));
        }
#pragma unroll
        for (int i = 0; i < RPT; i++)
          if (y0 + r0 + i < H) *reinterpret_cast<raw16*>(gi + (long)i * W * C) = v[i];
      }
    }
    STAMP();
#ifdef D2AMD_PROFILE
    if (dbg_on) { L.dbg[127] = dbg_n; L.dbg[126] = (unsigned long long)wst_n; L.dbg[125] = __builtin_readcyclecounter() - L.dbg[0]; }
    dbg_n = 0;
#endif
    KST(3, wall_clock64());
    KST(4, (unsigned long long)wst_n | (unsigned long long)blockIdx.x << 32);
#undef KST
    // (no barrier here: the image is read before the stores issue; the next tile's first writes over it -- zero rows, DMA
    // rows -- lie behind its barrier A0, which every wave reaches only after its reads of the image)
  }
#undef STAMP
}

// ---- convert_boxes_to_pooler_format (poolers.py:62-104) in one launch, no host sync -----------------
struct ImgEnds { int n; int end[D2AMD_POOLER_MAX_IMAGES]; };  // exclusive prefix ends of the per-image box counts
__global__ void boxes_to_rois_kernel(const float* __restrict__ boxes, int K, int width, ImgEnds e,
                                     float* __restrict__ rois) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  int b = 0;
  for (int i = 0; i < e.n; i++) b += (k >= e.end[i]) ? 1 : 0;
  float* o = rois + (long)k * (width + 1);
  o[0] = (float)b;
  for (int c = 0; c < width; c++) o[1 + c] = boxes[(long)k * width + c];
}

// the same for per-image box tensors that were never concatenated (ROIPooler.forward's box_lists): saves the
// torch.cat and a separate call on the host
struct ImgBoxes { int n; int end[D2AMD_POOLER_MAX_IMAGES]; const float* ptr[D2AMD_POOLER_MAX_IMAGES]; };
__global__ void box_lists_to_rois_kernel(ImgBoxes e, int K, float* __restrict__ rois) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  int b = 0;
  for (int i = 0; i < e.n; i++) b += (k >= e.end[i]) ? 1 : 0;
  const float4 v = reinterpret_cast<const float4*>(e.ptr[b])[k - (b ? e.end[b - 1] : 0)];
  float* o = rois + (long)k * 5;
  o[0] = (float)b; o[1] = v.x; o[2] = v.y; o[3] = v.z; o[4] = v.w;
}

// (both box lists of a paired forward in one launch)
__global__ void box_lists_to_rois_pair_kernel(ImgBoxes e1, int K1, float* __restrict__ rois1, ImgBoxes e2, int K2,
                                              float* __restrict__ rois2) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K1 + K2) return;
  const bool second = k >= K1;
  const ImgBoxes& e = second ? e2 : e1;
  if (second) k -= K1;
  int b = 0;
  for (int i = 0; i < e.n; i++) b += (k >= e.end[i]) ? 1 : 0;
  const float4 v = reinterpret_cast<const float4*>(e.ptr[b])[k - (b ? e.end[b - 1] : 0)];
  float* o = (second ? rois2 : rois1) + (long)k * 5;
  o[0] = (float)b; o[1] = v.x; o[2] = v.y; o[3] = v.z; o[4] = v.w;
}

// ---- ROI processing order of the forward (see fwd_roi_of); K <= ROI_ORDER_MAX, one workgroup ----------------------
// A counting sort by bucket = (level, image, cell of the ROI centre at its level, cells in Morton order); the 14
// bucket bits go to the level and image numbers first, the rest (<= 10) to the cell: 16-px cells on the finest level
// for 2 images x 4 levels.  (A bitonic sort of full 32-bit keys was the first version: 8 us on the critical path
// against 3.8 us for the plain conversion, PMC fetch 1.04 x the features; 64-px cells: 1.34 x; list order: 1.7 x.)
// The order inside a bucket is whatever the LDS atomics give: the processing order is not deterministic, the results
// are (row k of the output is ROI k).
// LISTS: also does box_lists_to_rois_kernel's job (the per-image box tensors -> rois [K, 5]) on the way.
constexpr int ROI_ORDER_MAX = 4096, ROI_BUCKETS = 1 << 14;
template <bool LISTS>
__global__ __launch_bounds__(1024) void roi_order_kernel(PoolLevels L, ImgBoxes e, float* __restrict__ rois, int K,
                                                        int* __restrict__ perm) {
  __shared__ int hist[ROI_BUCKETS];  // counts, then exclusive offsets
  __shared__ int wtot[16];
  __shared__ float s_scale[POOL_MAX_LEVELS];
  __shared__ int s_sh[POOL_MAX_LEVELS];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int PER = ROI_ORDER_MAX / 1024;
  int bucket[PER], at[PER];
  int lb = 0, ib = 0;  // bits of the level / image number (uniform)
  while ((1 << lb) < L.num_levels) lb++;
  while ((1 << ib) < L.N) ib++;
  const int hb = min(5, (14 - lb - ib) >> 1);  // bits per cell coordinate (L.N <= 32, levels <= 8: >= 3)
  const int used = 1 << (lb + ib + 2 * hb);    // buckets in use (<= ROI_BUCKETS)
  // the boxes first: their round trip overlaps the zeroing
  float4 bx[PER];
  int bimg[PER];
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const int k = tid + q * 1024;
    bx[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    bimg[q] = 0;
    if (k < K) {
      if (LISTS) {
        int b = 0;
        for (int i = 0; i < e.n; i++) b += (k >= e.end[i]) ? 1 : 0;
        bx[q] = reinterpret_cast<const float4*>(e.ptr[b])[k - (b ? e.end[b - 1] : 0)];
        bimg[q] = b;
      } else {
        const float* r = rois + (long)k * 5;
        bimg[q] = (int)r[0];
        bx[q] = make_float4(r[1], r[2], r[3], r[4]);
      }
    }
  }
  for (int i = tid; i < used; i += 1024) hist[i] = 0;
  if (tid < POOL_MAX_LEVELS) {  // per level: scale and cell size 2^sh feature pixels (the larger side spans < 2^hb cells)
    int hmax = 1, sh = 0;
    float sc = 0.f;
#pragma unroll
    for (int l = 0; l < POOL_MAX_LEVELS; l++)
      if (l == tid) { hmax = max(L.H[l], L.W[l]); sc = L.scale[l]; }
    while ((hmax >> sh) >= (1 << hb)) sh++;
    s_scale[tid] = sc;
    s_sh[tid] = sh;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const int k = tid + q * 1024;
    bucket[q] = -1;
    if (k < K) {
      const float4 v = bx[q];
      const int img = bimg[q];
      if (LISTS) {
        float* o = rois + (long)k * 5;
        o[0] = (float)img; o[1] = v.x; o[2] = v.y; o[3] = v.z; o[4] = v.w;
      }
      const float box[4] = {v.x, v.y, v.z, v.w};
      int lvl = assign_level(box, L);
      if (lvl < 0) lvl = L.num_levels - 1;  // no level: pooled as zeros, anywhere
      const float sc = s_scale[lvl];
      const int sh = s_sh[lvl];
      const int tx = min(max((int)((box[0] + box[2]) * 0.5f * sc) >> sh, 0), (1 << hb) - 1);
      const int ty = min(max((int)((box[1] + box[3]) * 0.5f * sc) >> sh, 0), (1 << hb) - 1);
      int mort = 0;
#pragma unroll
      for (int j = 0; j < 5; j++) mort |= ((tx >> j) & 1) << (2 * j) | ((ty >> j) & 1) << (2 * j + 1);
      bucket[q] = (((lvl << ib) | min(img, (1 << ib) - 1)) << (2 * hb)) | mort;
      at[q] = atomicAdd(&hist[bucket[q]], 1);
    }
  }
  __syncthreads();
  // exclusive scan of the counts: 16 consecutive buckets per thread, wave scan, 16 wave totals
  constexpr int BPT = ROI_BUCKETS / 1024;
  int c[BPT], sum = 0;
#pragma unroll
  for (int j = 0; j < BPT; j++) { c[j] = tid * BPT + j < used ? hist[tid * BPT + j] : 0; sum += c[j]; }
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  if (lane == 63) wtot[wid] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int w = 0; w < wid; w++) run += wtot[w];
#pragma unroll
  for (int j = 0; j < BPT; j++) {
    if (tid * BPT + j < used) hist[tid * BPT + j] = run;
    run += c[j];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER; q++)
    if (bucket[q] >= 0) perm[hist[bucket[q]] + at[q]] = tid + q * 1024;
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
  L.sr = p->sampling_ratio; L.K = K;
  // (bits 1.. of L.aligned: the rounding of the ROI coordinates behind the level assignment, roi_geom_box)
  L.aligned = (p->aligned ? 1 : 0) | ((p->roi_rounding ? (p->dtype == D2AMD_F16 ? 1 : p->dtype == D2AMD_BF16 ? 2 : 0) : 0) << 1);
  L.min_level = p->min_level; L.max_level = p->max_level; L.canonical_level = p->canonical_level;
  L.canonical_size = p->canonical_box_size;
  { const char* e = d2_prof_env("D2AMD_ABLATE"); L.ablate = e ? atoi(e) : 0; }
  L.dbg = nullptr; L.dbg_block = -1; L.wgstamps = nullptr; L.tile_cnt = nullptr; L.tile_list = nullptr; L.queue = nullptr; L.qcap = 0; L.qctr = nullptr; L.part_scratch = nullptr; L.part_tickets = nullptr; L.tab_off = 0;
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

// (pair: a second pooler of the same feature maps in the same launch -- the NHWC 16-B vector kernel only, else
// EUNSUPPORTED with nothing launched)
// where the backward keeps its records and its queue words (pool_bwd_nhwc_impl, phase 4, fills it: no launch)
struct PoolRecPlan { RoiRec* rec; int* qmem; int qzero, qints; };
struct PoolFwdPairCall { const d2amd_pooler_params* p2; const float* rois2; void* out2; int K2; const PoolRecPlan* plan; };
template <typename T>
static int pool_fwd_impl(const d2amd_pooler_params* p, const void* const* inputs, const float* rois, void* output,
                         int K, hipStream_t s, const int* perm = nullptr, const PoolFwdPairCall* pair = nullptr) {
  PoolLevels L = make_levels(p, inputs, K);
  L.perm = p->layout == D2AMD_NHWC ? perm : nullptr;
  const int bins = p->pooled_h * p->pooled_w;
  constexpr int VEC = V16<T>::N;
  if (p->layout == D2AMD_NHWC) {
    int wmax = 0;
    for (int l = 0; l < p->num_levels; l++) wmax = p->W[l] > wmax ? p->W[l] : wmax;
    // the vector kernel's 32-bit tap offsets: (12 rows x W) x C elements must stay below 2^32
    const bool vec = (p->C % VEC == 0) && all_aligned16(inputs, p->num_levels, output) && p->C <= 8192 && wmax <= 16384;
    const int cg = vec ? p->C / VEC : p->C;
    // workgroup shape (profiles/r01/v5_pool_fwd_sweep.txt): the kernel is insensitive to it within +-10 % --
    // the per-workgroup prologue (ROI load, level, table build: ~4 us) and the tap loads trade off -- because
    // the bound is the ~5.5 TB/s of tap bytes requested through L1 (each pixel of a bin's footprint is
    // re-requested by the neighbouring bins); 512 threads and ~1,000 workgroups measured best
    int nthr = vec ? 512 : 256;
    { const char* e = d2_prof_env("D2AMD_FWD_THREADS"); if (e && (atoi(e) == 256 || atoi(e) == 512 || atoi(e) == 1024)) nthr = atoi(e); }
    if (!vec) nthr = 256;
    const int passes = cdiv((long)bins * cg, nthr);
    int nsplit = K > 0 ? 1024 / K : 1;
    { const char* e = d2_prof_env("D2AMD_FWD_NSPLIT"); if (e && atoi(e) > 0) nsplit = atoi(e); }  // profiling switch
    nsplit = nsplit < 1 ? 1 : (nsplit > passes ? passes : nsplit);
    if (nsplit > bins) nsplit = bins;
    D2_CHECK_ARG(nsplit <= 65535, "roi_pooler_forward: internal split too large");
    dim3 grid(K, nsplit);
    PoolFwdPair<T> P2{};
    if (pair) {
      static const bool wide_env = d2_prof_env("D2AMD_FWD_VARIANT") && atoi(d2_prof_env("D2AMD_FWD_VARIANT")) == 1;
      const bool ok = vec && nthr == 512 && !wide_env && perm == nullptr && d2_prof_env("D2AMD_POOL_STAMPS") == nullptr &&
          ((uintptr_t)pair->out2 & 15) == 0 && K > 0 && pair->K2 > 0;
      if (!ok) {
        set_error("roi_pooler_forward_pair: outside the paired forward (NHWC, 16-B channel vectors, list order)");
        return D2AMD_EUNSUPPORTED;
      }
      const int bins2 = pair->p2->pooled_h * pair->p2->pooled_w;
      const int passes2 = cdiv((long)bins2 * cg, nthr);
      int ns2 = 1024 / pair->K2;
      { const char* e = d2_prof_env("D2AMD_FWD_NSPLIT2"); if (e && atoi(e) > 0) ns2 = atoi(e); }  // profiling switch
      ns2 = ns2 < 1 ? 1 : (ns2 > passes2 ? passes2 : ns2);
      if (ns2 > bins2) ns2 = bins2;
      const long total = (long)K * nsplit + (long)pair->K2 * ns2;
      D2_CHECK_ARG(total < (1l << 30), "roi_pooler_forward_pair: too many workgroups");
      P2 = PoolFwdPair<T>{pair->rois2, (T*)pair->out2, K, nsplit, pair->K2, ns2, pair->p2->pooled_h, pair->p2->pooled_w,
                          nullptr, nullptr, 0, 0};
      if (pair->plan) { P2.rec = pair->plan->rec; P2.qmem = pair->plan->qmem; P2.qzero = pair->plan->qzero; P2.qints = pair->plan->qints; }
      grid = dim3((unsigned)total, 1);
    }
    PoolLevels Lf = L;
    for (int l = 0; l < p->num_levels; l++)
      if ((long)p->H[l] * p->W[l] * p->C >= (1l << 32)) Lf.tab_off = 1;
    const char* stamp_path = d2_prof_env("D2AMD_POOL_STAMPS");  // profiling only: per-workgroup timeline dump
    const long nwg = (long)K * nsplit;
    if (stamp_path) {
      D2_HIP_OK(hipMalloc(&Lf.wgstamps, (size_t)nwg * 5 * 8));
      D2_HIP_OK(hipMemsetAsync(Lf.wgstamps, 0, (size_t)nwg * 5 * 8, s));
    }
    const char* tname = pair ? "pool_fwd_pair" : p->pooled_h <= 7 ? "pool_fwd_r7" : "pool_fwd_r14";
    const bool timed = timing_begin(tname, s);
    if (vec && nthr == 1024)
      hipLaunchKernelGGL((pool_fwd_nhwc_kernel<T, VEC, 1024>), grid, dim3(1024), 0, s, Lf, rois, (T*)output, nsplit, PoolFwdPair<T>{});
    else if (vec && nthr == 512) {
      // 4 loads in flight per lane and <= 84 VGPRs: three 512-thread workgroups per CU instead of two (8 loads, 108
      // VGPRs): 48.3 -> 43.6 us (box), 34.9 -> 30.2 us (mask); D2AMD_FWD_VARIANT=1 selects the previous shape (A/B)
      static const bool wide = d2_prof_env("D2AMD_FWD_VARIANT") && atoi(d2_prof_env("D2AMD_FWD_VARIANT")) == 1;
      // (Tried in round 3 and removed, commit eb0e005: the forward as a GEMM on the matrix cores -- the ROI's footprint
      // staged once in chunks of pixel rows, weight image [bin][pixel] as hi + lo 16-bit parts, the backward's
      // contraction with bins and pixels swapped.  Within 1 ulp of this kernel on every case, and a workgroup needs
      // the same ~21 us per ROI (4.8 us tables + 13.5 chunks x 1.13 us) -- but with 57 KB of LDS and 100 VGPRs two
      // workgroups fit a CU where this kernel runs four: 57-65 us against 44 us for the box head.)
      if (wide)
        hipLaunchKernelGGL((pool_fwd_nhwc_kernel<T, VEC, 512>), grid, dim3(512), 0, s, Lf, rois, (T*)output, nsplit, PoolFwdPair<T>{});
      else
        hipLaunchKernelGGL((pool_fwd_nhwc_kernel<T, VEC, 512, 4, 6, true>), grid, dim3(512), 0, s, Lf, rois, (T*)output, nsplit, P2);
    }
    else if (vec)
      hipLaunchKernelGGL((pool_fwd_nhwc_kernel<T, VEC, 256>), grid, dim3(256), 0, s, Lf, rois, (T*)output, nsplit, PoolFwdPair<T>{});
    else
      hipLaunchKernelGGL((pool_fwd_nhwc_kernel<T, 1, 256>), grid, dim3(256), 0, s, Lf, rois, (T*)output, nsplit, PoolFwdPair<T>{});
    if (timed) timing_end(tname, s);
    if (stamp_path) {
      D2_HIP_OK(hipStreamSynchronize(s));
      unsigned long long* h = (unsigned long long*)malloc((size_t)nwg * 5 * 8);
      D2_HIP_OK(hipMemcpy(h, Lf.wgstamps, (size_t)nwg * 5 * 8, hipMemcpyDeviceToHost));
      char fn[512];
      snprintf(fn, sizeof(fn), "%s.fwd", stamp_path);
      FILE* f = fopen(fn, "w");
      if (f) {
        for (long i = 0; i < nwg; i++)
          fprintf(f, "%ld %llu %llu %llu %llu %llu\n", i, h[5 * i], h[5 * i + 1], h[5 * i + 2], h[5 * i + 3], h[5 * i + 4]);
        fclose(f);
      }
      free(h);
      (void)hipFree(Lf.wgstamps);
    }
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

// one side stream + fork/join events per device, created on first use (never destroyed: process lifetime)
struct SideStream { hipStream_t stream; hipEvent_t fork, join; };
static SideStream* side_stream() {
  static SideStream table[64];
  static bool made[64] = {};
  static const bool off = d2_prof_env("D2AMD_NO_SIDE_STREAM") != nullptr;
  int dev = 0;
  if (off || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!made[dev]) {
    SideStream t{};
    if (hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&t.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    table[dev] = t;
    made[dev] = true;
  }
  return &table[dev];
}

// workgroups of `fn` (block size `threads`, static LDS) that are resident at once on the current device
static long resident_workgroups(const void* fn, int threads) {
  struct Ent { const void* fn; int dev; long n; };
  static Ent cache[16];
  static int used = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  for (int i = 0; i < used; i++)
    if (cache[i].fn == fn && cache[i].dev == dev) return cache[i].n;
  int cus = 0, per = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, threads, 0) != hipSuccess) return 0;
  const long n = (long)cus * per;
  if (used < 16) cache[used++] = Ent{fn, dev, n};
  return n;
}
// shapes the K-concatenated tile gather takes (16-bit I/O): whole 32-channel groups, the tiles' geometry packed into
// 16 / 24 bits, dY row indices in 31 bits (checked per call where K is known)
static bool kcat_shape_ok(const d2amd_pooler_params* p) {
  bool ok = p->C % 32 == 0 && p->C <= 8192 && p->N < (1 << 23);
  for (int l = 0; l < p->num_levels; l++) ok = ok && p->H[l] < 65536 && p->W[l] < 65536;
  return ok;
}

// levels with at most this many tiles take the GROUPS > 1 kernel (few tiles <=> long ROI lists)
constexpr int COARSE_TILES = 512;

template <typename T, int VEC, int GROUPS, int RS>
static void launch_bwd(const PoolLevels& L, const RoiRec* rec, const void* gout, int nslab, long total,
                       hipStream_t s, const PoolTileIds& ids) {
  const int grid = L.queue ? 8 * L.qcap * nslab : (int)((total + 7) / 8) * 8;
  hipLaunchKernelGGL((pool_bwd_nhwc_kernel<T, VEC, GROUPS, RS, 8>), dim3(grid), dim3(CT * RS * GROUPS), 0, s, L, rec,
                     (const T*)gout, nslab, (int)total, ids);
}

static size_t pool_al(size_t x) { return (x + 255) / 256 * 256; }
// counters, tickets of the split tiles, then 8 queues of 64-bit slots: at most one slot per tile + the parts of the
// split ones (<= SCR_PER_XCD_MAX per queue)
static size_t pool_queue_bytes(long ntiles) {
  return pool_al((size_t)(QCTR + QTICKETS) * sizeof(int) + (size_t)8 * (ntiles + 8 * SCR_PER_XCD_MAX) * sizeof(int2));
}
static long pool_ntiles(const d2amd_pooler_params* p) {
  long n = 0;
  for (int l = 0; l < p->num_levels; l++) n += (long)cdiv(p->H[l], TILE) * cdiv(p->W[l], TILE) * p->N;
  return n;
}

// (pair: a second pooler of the same feature maps, binned and gathered together with this one -- phase 0 of the
// persistent MFMA tile gather of a pooler with <= 8 bins per axis only; probe: no launch at all, the return value says
// whether the call would take that kernel: OK / EUNSUPPORTED)
struct PoolPairCall { const d2amd_pooler_params* p2; const void* gout2; const float* rois2; int K2; };

template <typename T>
static int pool_bwd_nhwc_impl(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                              void* const* grad_inputs, int K_first, void* workspace, size_t workspace_bytes,
                              hipStream_t s, bool accumulate, int phase = 0, const PoolPairCall* pair = nullptr,
                              bool probe = false, PoolRecPlan* plan = nullptr) {
  const int K = K_first + (pair ? pair->K2 : 0);  // records: the first pooler's ROIs, then the second one's
  // phase 0: everything; 1: only the binning (records, per-tile ROI lists, work queues -- depends on the ROIs alone
  // in accumulate mode, so a caller can run it beside other work); 2: only the gather, ADDING, after a phase-1 call
  // with the same arguments and workspace; 3: the same but WRITING: the tiles without ROIs are zero-filled by their
  // own small launch, every other tile is written once (= what phase 0 without `accumulate` produces)
  // r06 -- 4: no launch, `plan` receives where the records and the queue words live (the paired FORWARD of the same ROIs
  // writes them: d2amd_roi_pooler_forward_pair_records); 5: phase 0 without the records launch, after such a forward
  constexpr int VEC = V16<T>::N;
  const bool vec = (p->C % VEC == 0) && all_aligned16((const void* const*)grad_inputs, p->num_levels, grad_output);
  const int cg = vec ? p->C / VEC : p->C;
  const int nslab = cdiv(cg, LPP);
  const size_t need = (size_t)(K > 0 ? K : 1) * sizeof(RoiRec);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("roi_pooler_backward: workspace too small (%zu < %zu)", workspace_bytes, need);
    return D2AMD_EWORKSPACE;
  }
  RoiRec* rec = (RoiRec*)workspace;
  // per-tile ROI lists live behind the records when the caller sized the workspace with
  // d2amd_roi_pooler_backward_workspace_bytes (the older K-only size still works: tiles then scan)
  const long ntiles = pool_ntiles(p);
  // (counts [+ the first pooler's counts of a pair], the tiles' geometry, the lists, the queues, the scratch slots)
  const size_t off_cnt = pool_al(need), off_geo = off_cnt + pool_al((size_t)ntiles * 4) * (pair ? 2 : 1);
  const size_t off_list = off_geo + pool_al((size_t)ntiles * sizeof(int4));
  const bool lists = K > 0 && ntiles > 0 && workspace_bytes >= off_list + (size_t)ntiles * TILE_CAP * sizeof(TileEntry) &&
      d2_prof_env("D2AMD_POOL_NOLISTS") == nullptr;
  int* tile_cnt = lists ? (int*)((char*)workspace + off_cnt) : nullptr;
  int* tile_cnt1 = lists && pair ? (int*)((char*)workspace + off_cnt + pool_al((size_t)ntiles * 4)) : nullptr;
  int4* tile_geo = lists ? (int4*)((char*)workspace + off_geo) : nullptr;
  TileEntry* tile_list = lists ? (TileEntry*)((char*)workspace + off_list) : nullptr;
  PoolLevels L0 = make_levels(p, (const void* const*)grad_inputs, K);
  L0.accumulate = (accumulate && phase != 3) ? 1 : 0;
  // work queues (see tile_lists_kernel): capacity per XCD = the tiles the 4x4-block deal gives it
  TileQueues Q{};
  const size_t off_q = off_list + pool_al((size_t)ntiles * TILE_CAP * sizeof(TileEntry));
  long nblocks4 = 0;
  for (int l = 0; l < p->num_levels; l++) nblocks4 += (long)cdiv(cdiv(p->H[l], TILE), 4) * cdiv(cdiv(p->W[l], TILE), 4) * p->N;
  const bool queues = lists && nblocks4 <= 8192 && ntiles < (1l << 24) &&
      workspace_bytes >= off_q + pool_queue_bytes(ntiles) && d2_prof_env("D2AMD_POOL_NOQUEUE") == nullptr;
  // one launch of the LDS-staged kernel for all levels (16-B channel vectors + work queues), else the two-launch
  // register-gather kernels
  const bool staged = queues && vec && d2_prof_env("D2AMD_POOL_NOSTAGED") == nullptr;
  if (accumulate && !staged) {
    set_error("roi_pooler_backward_accumulate: needs the staged tile gather (16-B aligned channel vectors, work queues)");
    return D2AMD_EUNSUPPORTED;
  }
  if (accumulate && K == 0 && phase != 3) return D2AMD_OK;  // nothing to add
  // lists are split only for the kernel that can add the parts: the 16-bit MFMA tile gather
  static const bool no_mfma_env = d2_prof_env("D2AMD_POOL_NOMFMA") != nullptr;
  const bool kcat_ok = staged && sizeof(T) == 2 && !no_mfma_env && kcat_shape_ok(p) &&
      (long)K_first * p->pooled_h * p->pooled_w < (1l << 31) &&
      (!pair || (long)pair->K2 * pair->p2->pooled_h * pair->p2->pooled_w < (1l << 31));
  const bool split_capable = kcat_ok && nslab <= SPLIT_MAX_SLABS;
  const size_t slot_bytes = (size_t)nslab * 32 * (2 * CT) * sizeof(float);  // 16 accumulators x 2 tiles x 512 threads
  if (pair || probe) {  // the paired gather exists in the K-concatenated tile gather only (either pooler may come first:
    // a list entry carries its own pooled size)
    const bool ok = kcat_ok && K_first > 0 && (phase <= 2 || phase == 4 || phase == 5) && !accumulate;
    if (!ok || probe) return ok ? D2AMD_OK : D2AMD_EUNSUPPORTED;
  }
  if (queues) {
    int per[2][8] = {};
    int base[2] = {0, 0};
    for (int l = 0; l < p->num_levels; l++) {
      const int ty = cdiv(p->H[l], TILE), tx = cdiv(p->W[l], TILE);
      const int pass = (!staged && ty * tx * p->N <= COARSE_TILES) ? 1 : 0;
      if (pass) Q.coarse_mask |= 1u << l;
      Q.pass_base[l] = base[pass];
      base[pass] += ty * tx * p->N;
      static const int deal_env = d2_prof_env("D2AMD_POOL_DEAL") ? atoi(d2_prof_env("D2AMD_POOL_DEAL")) : -1;  // A/B: fixed shift
      const int sh = deal_env >= 0 ? (deal_env > 2 ? 2 : deal_env) : (ty * tx * p->N <= 512 ? 0 : ty * tx * p->N <= 2048 ? 1 : 2);
      Q.deal_shift[l] = sh;
      const int bs = 1 << sh;
      for (int n = 0; n < p->N; n++)
        for (int by = 0; by < ty; by += bs)
          for (int bx = 0; bx < tx; bx += bs)
            per[pass][tile_xcd(l, n, by, bx, tx, sh)] += min(bs, ty - by) * min(bs, tx - bx);
    }
    for (int x = 0; x < 8; x++) { Q.cap[0] = max(Q.cap[0], per[0][x]); Q.cap[1] = max(Q.cap[1], per[1][x]); }
    // split lists (the MFMA tile gather only): scratch behind the queues, if the caller's workspace has it
    static const bool no_split = d2_prof_env("D2AMD_POOL_NOSPLIT") != nullptr;
    const int sx = 8 * min(SCR_PER_XCD_MAX, max(16, Q.cap[0] / 3));
    if (split_capable && !no_split && workspace_bytes >= off_q + pool_queue_bytes(ntiles) + (size_t)sx * slot_bytes) {
      Q.scr_total = sx;
      Q.cap[0] += sx;  // a queue holds at most one entry per tile + (all on one XCD) every part
    }
    Q.qbase = QCTR + QTICKETS;
    static const int thr_s = d2_prof_env("D2AMD_POOL_QTHR") ? atoi(d2_prof_env("D2AMD_POOL_QTHR")) : 6;
    static const int thr_f0 = d2_prof_env("D2AMD_POOL_QTHR_FINE") ? atoi(d2_prof_env("D2AMD_POOL_QTHR_FINE")) : 4;
    const int thr_f = staged ? thr_s : thr_f0;
    static const int thr_c = d2_prof_env("D2AMD_POOL_QTHR_COARSE") ? atoi(d2_prof_env("D2AMD_POOL_QTHR_COARSE")) : 16;
    Q.thr[0] = thr_f; Q.thr[1] = thr_c;
    static const int split_min = d2_prof_env("D2AMD_POOL_SPLIT_MIN") ? atoi(d2_prof_env("D2AMD_POOL_SPLIT_MIN")) : SPLIT_MIN;
    static const int part_len = d2_prof_env("D2AMD_POOL_PART_LEN") ? atoi(d2_prof_env("D2AMD_POOL_PART_LEN")) : PART_LEN;
    Q.split_min = split_min; Q.part_len = part_len > 0 ? part_len : PART_LEN;
    Q.mem = (int*)((char*)workspace + off_q);
    Q.esize = (int)sizeof(T);
    Q.zero_fill = vec ? 1 : 0;
  }
  const int qzero = QCTR + QTICKETS;
  const int qints = queues ? qzero + 2 * 8 * (Q.cap[0] + Q.cap[1]) : 0;
  D2_CHECK_ARG(!queues || (size_t)qints * sizeof(int) <= pool_queue_bytes(ntiles), "roi_pooler_backward: queue layout");
  if (phase == 3) {  // (binned in accumulate mode: empty tiles were neither queued nor written)
    D2_CHECK_ARG(lists && vec, "roi_pooler_backward: phase 3 without per-tile lists");
    hipLaunchKernelGGL(zero_empty_tiles_kernel, dim3(cdiv(ntiles, LISTS_WAVES)), dim3(64 * LISTS_WAVES), 0, s, L0,
                       (int)ntiles, (const int*)tile_cnt, (int)sizeof(T));
    D2_LAUNCH_OK();
  }
  if (phase == 4) {
    if (!(pair && queues && lists && plan)) return D2AMD_EUNSUPPORTED;
    *plan = PoolRecPlan{rec, Q.mem, qzero, qints};
    return D2AMD_OK;
  }
  if (phase == 5 && !(pair && queues && lists)) {
    set_error("roi_pooler_backward_pair: phase 5 outside the paired tile gather");
    return D2AMD_EUNSUPPORTED;
  }
  if (K > 0 && (phase < 2 || phase == 5)) {
    const PairBin PB2{pair ? pair->rois2 : nullptr, K_first, pair ? pair->p2->pooled_h : 0, pair ? pair->p2->pooled_w : 0};
    if (phase != 5) {
      hipLaunchKernelGGL(roi_records_kernel, dim3(cdiv(K, 256)), dim3(256), 0, s, L0, rois, rec, Q.mem, qzero, qints, PB2);
      D2_LAUNCH_OK();
    }
    if (lists) {
      hipLaunchKernelGGL(tile_lists_kernel, dim3(cdiv(ntiles, LISTS_WAVES)), dim3(64 * LISTS_WAVES), 0, s, L0, rec, (int)ntiles, tile_cnt,
                         tile_list, Q, K_first, tile_cnt1, pair ? pair->p2->pooled_h : p->pooled_h,
                         pair ? pair->p2->pooled_w : p->pooled_w, tile_geo);
      D2_LAUNCH_OK();
    }
  }
  if (phase == 1) return D2AMD_OK;
  // profiling switches: D2AMD_BWD_CFG = "<fine GROUPS><fine RS><coarse GROUPS><coarse RS>", e.g. 1122
  static const int cfg = d2_prof_env("D2AMD_BWD_CFG") ? atoi(d2_prof_env("D2AMD_BWD_CFG")) : 1222;
  const int fg = cfg / 1000 % 10, fr = cfg / 100 % 10, cgp = cfg / 10 % 10, cr = cfg % 10;
  // The coarse-level launch has few, long-running workgroups (latency bound) and the fine-level one
  // fills the chip at 16 waves / CU: they overlap on a library-owned side stream (fork / join with
  // events; both only read `rec` and dY and write disjoint grad tensors).
  if (staged) {
    PoolLevels L = L0;
    PoolTileIds ids{};
    for (int l = 0; l < p->num_levels; l++) ids.first[l] = L0.tile_base[l];
    L.tile_cnt = tile_cnt;
    L.tile_list = tile_list;
    L.tile_geo = tile_geo;
    L.queue = reinterpret_cast<const int2*>(Q.mem + Q.qbase);
    L.qcap = Q.cap[0];
    L.part_tickets = Q.mem + QCTR;
    L.part_scratch = (float*)((char*)workspace + off_q + pool_queue_bytes(ntiles));
    const long total = 8l * L.qcap * nslab;
    if (total == 0) return D2AMD_OK;
    D2_CHECK_ARG(total < (1l << 30), "roi_pooler_backward: too many tiles");
    const char* stamp_path = d2_prof_env("D2AMD_POOL_STAMPS");  // profiling only: per-workgroup timeline dump
    if (stamp_path) {
      D2_HIP_OK(hipMalloc(&L.wgstamps, (size_t)total * 5 * 8));
      D2_HIP_OK(hipMemsetAsync(L.wgstamps, 0, (size_t)total * 5 * 8, s));
    }
#ifdef D2AMD_PROFILE
    static unsigned long long* dbg_dev = nullptr;
    if (d2_prof_env("D2AMD_DBG_BLOCK")) {
      if (!dbg_dev) (void)hipMalloc(&dbg_dev, 128 * 8);
      (void)hipMemsetAsync(dbg_dev, 0, 128 * 8, s);
      L.dbg = dbg_dev;
      L.dbg_block = atoi(d2_prof_env("D2AMD_DBG_BLOCK"));
    }
#endif
    const char* tname = pair ? "pool_bwd_pair" : p->pooled_h <= 7 ? "pool_bwd_staged_r7" : "pool_bwd_staged_r14";
    const bool timed = timing_begin(tname, s);
    const int pmax = p->pooled_h > p->pooled_w ? p->pooled_h : p->pooled_w;
    bool mfma = false;
    if constexpr (sizeof(T) == 2) {  // 16-bit I/O: the K-concatenated tile gather on the matrix cores
      mfma = kcat_ok;
      if (mfma) {
        // persistent workgroups: one resident wave of them (a multiple of 8: every XCD gets the same number), each
        // fetching tiles until its queue is empty; never more than there are queue slots
        L.qctr = Q.mem;
        const PoolPairArgs P2 = pair ? PoolPairArgs{pair->gout2, tile_cnt1, K_first, pair->K2, pair->p2->pooled_h,
                                                    pair->p2->pooled_w} : PoolPairArgs{};
        // (4 waves x 48 k: three workgroups per CU.  Same-box A/B of the paired launch, profiles/r06/pool_bwd_kcat.md:
        // 4 x 48: 50.0 | 8 x 64, two per CU: 53.4 | 4 x 64, two per CU: 57-76 | 4 x 32, four per CU, spilling: 58.2 us)
        auto fn = pool_bwd_kcat_kernel<T, 4, 48>;
        const long r = resident_workgroups((const void*)fn, 256) & ~7l;
        const unsigned grid = (unsigned)(r >= 8 && r < total ? r : total);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, s, L, rec, (const T*)grad_output, nslab, P2);
      }
    }
    if (mfma) {
    } else if (pmax <= 8)
      hipLaunchKernelGGL((pool_bwd_staged_kernel<T, VEC, 8>), dim3((unsigned)total), dim3(2 * CT), 0, s, L, rec,
                         (const T*)grad_output, nslab, (int)total, ids);
    else if (pmax <= 16)
      hipLaunchKernelGGL((pool_bwd_staged_kernel<T, VEC, 16>), dim3((unsigned)total), dim3(2 * CT), 0, s, L, rec,
                         (const T*)grad_output, nslab, (int)total, ids);
    else
      hipLaunchKernelGGL((pool_bwd_staged_kernel<T, VEC, 32>), dim3((unsigned)total), dim3(2 * CT), 0, s, L, rec,
                         (const T*)grad_output, nslab, (int)total, ids);
    D2_LAUNCH_OK();
    if (timed) timing_end(tname, s);
#ifdef D2AMD_PROFILE
    if (L.dbg && L.dbg_block >= 0) {
      unsigned long long h[128];
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(h, L.dbg, sizeof(h), hipMemcpyDeviceToHost);
      fprintf(stderr, "[d2amd dbg] staged block %d cycles %llu; stamps %llu:", L.dbg_block, h[125], h[127]);
      for (unsigned i = 1; i < h[127] && i < 120; i++) fprintf(stderr, " %llu", h[i] - h[i - 1]);
      fprintf(stderr, "\n");
    }
#endif
    if (stamp_path) {
      D2_HIP_OK(hipStreamSynchronize(s));
      unsigned long long* h = (unsigned long long*)malloc((size_t)total * 5 * 8);
      D2_HIP_OK(hipMemcpy(h, L.wgstamps, (size_t)total * 5 * 8, hipMemcpyDeviceToHost));
      char fn[512];
      snprintf(fn, sizeof(fn), "%s.pass0", stamp_path);
      FILE* f = fopen(fn, "w");
      if (f) {
        for (long i = 0; i < total; i++)
          fprintf(f, "%ld %llu %llu %llu %llu %llu\n", i, h[5 * i], h[5 * i + 1], h[5 * i + 2], h[5 * i + 3], h[5 * i + 4]);
        fclose(f);
      }
      free(h);
      (void)hipFree(L.wgstamps);
      snprintf(fn, sizeof(fn), "%s.pass1", stamp_path);
      f = fopen(fn, "w");  // no second launch
      if (f) fclose(f);
    }
    return D2AMD_OK;
  }
  SideStream* side = side_stream();
  bool forked = false;
  for (int pass = 1; pass >= 0; pass--) {  // pass 1: coarse levels (side stream), pass 0: fine levels
    PoolLevels L = make_levels(p, (const void* const*)grad_inputs, K);
    PoolTileIds ids{};
    L.tile_cnt = tile_cnt;
    L.tile_list = tile_list;
    int base = 0;
    for (int l = 0; l < p->num_levels; l++) {
      const int tiles = cdiv(p->H[l], TILE) * cdiv(p->W[l], TILE) * p->N;
      const bool coarse = tiles <= COARSE_TILES;
      L.tile_base[l] = base;
      ids.first[l] = L0.tile_base[l];
      if (coarse == (pass == 1)) base += tiles;
    }
    for (int l = p->num_levels; l <= POOL_MAX_LEVELS; l++) L.tile_base[l] = base;
    long total = (long)base * nslab;
    if (total == 0) continue;
    if (queues) {
      L.queue = reinterpret_cast<const int2*>(Q.mem + Q.qbase) + (pass ? 8 * Q.cap[0] : 0);
      L.qcap = Q.cap[pass];
      total = 8l * L.qcap * nslab;  // workgroups of this launch (stamps are indexed by workgroup)
    }
    D2_CHECK_ARG(total < (1l << 30), "roi_pooler_backward: too many tiles");
    hipStream_t ls = s;
    if (pass == 1 && side) {
      D2_HIP_OK(hipEventRecord(side->fork, s));
      D2_HIP_OK(hipStreamWaitEvent(side->stream, side->fork, 0));
      ls = side->stream;
      forked = true;
    }
#ifdef D2AMD_PROFILE
    static unsigned long long* dbg_dev = nullptr;
    if (d2_prof_env("D2AMD_DBG_BLOCK")) {
      if (!dbg_dev) (void)hipMalloc(&dbg_dev, 128 * 8);
      (void)hipMemsetAsync(dbg_dev, 0, 128 * 8, ls);
      L.dbg = dbg_dev;
      L.dbg_block = (pass == atoi(d2_prof_env("D2AMD_DBG_PASS") ? d2_prof_env("D2AMD_DBG_PASS") : "0")) ? atoi(d2_prof_env("D2AMD_DBG_BLOCK")) : -1;
    }
#endif
    const char* stamp_path = d2_prof_env("D2AMD_POOL_STAMPS");  // profiling only: per-workgroup timeline dump
    if (stamp_path) {
      D2_HIP_OK(hipMalloc(&L.wgstamps, (size_t)total * 5 * 8));
      D2_HIP_OK(hipMemsetAsync(L.wgstamps, 0, (size_t)total * 5 * 8, ls));
    }
    const int g = pass == 0 ? fg : cgp, r = pass == 0 ? fr : cr;
    const char* tname = pass == 0 ? (p->pooled_h <= 7 ? "pool_bwd_fine_r7" : "pool_bwd_fine_r14")
                                  : (p->pooled_h <= 7 ? "pool_bwd_coarse_r7" : "pool_bwd_coarse_r14");
    const bool timed = timing_begin(tname, ls);
    if (!vec) {
      if (g == 1) launch_bwd<T, 1, 1, 1>(L, rec, grad_output, nslab, total, ls, ids);
      else launch_bwd<T, 1, 2, 1>(L, rec, grad_output, nslab, total, ls, ids);
    } else if (g == 1 && r == 1) launch_bwd<T, VEC, 1, 1>(L, rec, grad_output, nslab, total, ls, ids);
    else if (g == 1 && r == 2) launch_bwd<T, VEC, 1, 2>(L, rec, grad_output, nslab, total, ls, ids);
    else if (g == 2 && r == 1) launch_bwd<T, VEC, 2, 1>(L, rec, grad_output, nslab, total, ls, ids);
    else if (g == 2 && r == 2) launch_bwd<T, VEC, 2, 2>(L, rec, grad_output, nslab, total, ls, ids);
    else if (g == 4 && r == 1) launch_bwd<T, VEC, 4, 1>(L, rec, grad_output, nslab, total, ls, ids);
    else { set_error("roi_pooler_backward: bad D2AMD_BWD_CFG %d", cfg); return D2AMD_EINVAL; }
    D2_LAUNCH_OK();
    if (timed) timing_end(tname, ls);
    if (stamp_path) {
      D2_HIP_OK(hipStreamSynchronize(ls));
      unsigned long long* h = (unsigned long long*)malloc((size_t)total * 5 * 8);
      D2_HIP_OK(hipMemcpy(h, L.wgstamps, (size_t)total * 5 * 8, hipMemcpyDeviceToHost));
      char fn[512];
      snprintf(fn, sizeof(fn), "%s.pass%d", stamp_path, pass);
      FILE* f = fopen(fn, "w");
      if (f) {
        for (long i = 0; i < total; i++)
          fprintf(f, "%ld %llu %llu %llu %llu %llu\n", i, h[5 * i], h[5 * i + 1], h[5 * i + 2], h[5 * i + 3], h[5 * i + 4]);
        fclose(f);
      }
      free(h);
      (void)hipFree(L.wgstamps);
    }
    if (pass == 1 && forked) D2_HIP_OK(hipEventRecord(side->join, side->stream));
#ifdef D2AMD_PROFILE
    if (L.dbg && L.dbg_block >= 0) {
      unsigned long long h[128];
      (void)hipStreamSynchronize(ls);
      (void)hipMemcpy(h, L.dbg, sizeof(h), hipMemcpyDeviceToHost);
      fprintf(stderr, "[d2amd dbg] pass %d block %d realtime(100MHz) %llu cycles %llu -> %.0f MHz; stamps %llu:", pass,
              L.dbg_block, h[126], h[125], h[126] ? 100.0 * (double)h[125] / (double)h[126] : 0.0, h[127]);
      for (unsigned i = 1; i < h[127] && i < 120; i++) fprintf(stderr, " %llu", h[i] - h[i - 1]);
      fprintf(stderr, "\n");
    }
#endif
  }
  if (forked) D2_HIP_OK(hipStreamWaitEvent(s, side->join, 0));
  return D2AMD_OK;
}

}  // namespace d2amd

using namespace d2amd;

extern "C" int d2amd_boxes_to_rois(const float* boxes, const int* counts, int num_images, int width, float* rois,
                                   void* stream) {
  D2_CHECK_ARG(num_images >= 0 && num_images <= D2AMD_POOLER_MAX_IMAGES, "boxes_to_rois: %d images (max %d)",
               num_images, D2AMD_POOLER_MAX_IMAGES);
  D2_CHECK_ARG(width == 4 || width == 5, "boxes_to_rois: box width %d", width);
  ImgEnds e{};
  e.n = num_images;
  long K = 0;
  for (int i = 0; i < num_images; i++) {
    D2_CHECK_ARG(counts && counts[i] >= 0, "boxes_to_rois: bad count");
    K += counts[i];
    e.end[i] = (int)K;
  }
  if (K == 0) return D2AMD_OK;
  D2_CHECK_ARG(boxes && rois && K < (1l << 31), "boxes_to_rois: null pointer / too many boxes");
  hipLaunchKernelGGL(boxes_to_rois_kernel, dim3(cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, boxes, (int)K, width,
                     e, rois);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

extern "C" int d2amd_roi_pooler_supported(const d2amd_pooler_params* p, int backward) {
  if (check_pooler(p, "roi_pooler_supported")) return 0;
  if (!pooler_fused_ok(p)) return 0;
  if (backward && p->layout != D2AMD_NHWC) return 0;
  return 1;
}

static int pooler_forward_entry(const d2amd_pooler_params* p, const void* const* inputs, const float* rois,
                                void* output, int K, const int* perm, void* stream) {
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
    return pool_fwd_impl<scalar_t>(p, inputs, rois, output, K, (hipStream_t)stream, perm);
  });
}

// the ROI order of an ordered forward: K ints in the caller's workspace, or nullptr (K too large / not NHWC / no room)
static int* roi_order_ws(const d2amd_pooler_params* p, int K, void* workspace, size_t workspace_bytes) {
  // (few ROIs touch the features sparsely anyway -- the mask head's 256 fetch 0.7 x the features in list order -- and
  // the ordering launch costs 4-5 us more than the plain conversion)
  if (workspace == nullptr || K < 512 || K > ROI_ORDER_MAX || workspace_bytes < (size_t)K * sizeof(int)) return nullptr;
  if (p->layout != D2AMD_NHWC || p->N > 32 || ((uintptr_t)workspace & 3)) return nullptr;
  static const bool off = d2_prof_env("D2AMD_FWD_NO_ORDER") != nullptr;  // A/B switch
  return off ? nullptr : (int*)workspace;
}

extern "C" int d2amd_roi_pooler_forward(const d2amd_pooler_params* p, const void* const* inputs, const float* rois,
                                        void* output, int K, void* stream) {
  return pooler_forward_entry(p, inputs, rois, output, K, nullptr, stream);
}

// Two poolers of the SAME feature maps in ONE launch (the box head's and the mask head's, roi_heads.py:780-846): rows of
// output1 / output2 are exactly what d2amd_roi_pooler_forward(p1 ...) / (p2 ...) write -- the same workgroups run the same
// code -- but the second pooler's workgroups start in the slots the first one's free instead of behind its last one.
// EUNSUPPORTED (nothing launched) outside the NHWC 16-B vector kernel / for different level rules: two calls then.
static int pooler_forward_pair_entry(const d2amd_pooler_params* p1, const void* const* inputs, const float* rois1,
                                     void* output1, int K1, const d2amd_pooler_params* p2, const float* rois2,
                                     void* output2, int K2, const PoolRecPlan* rec_plan, void* stream) {
  int rc = check_pooler(p1, "roi_pooler_forward_pair");
  if (rc) return rc;
  rc = check_pooler(p2, "roi_pooler_forward_pair");
  if (rc) return rc;
  D2_CHECK_ARG(K1 >= 0 && K2 >= 0, "roi_pooler_forward_pair: bad K");
  D2_CHECK_ARG(inputs && (K1 == 0 || (rois1 && output1)) && (K2 == 0 || (rois2 && output2)),
               "roi_pooler_forward_pair: null pointer");
  bool same = p1->num_levels == p2->num_levels && p1->N == p2->N && p1->C == p2->C && p1->dtype == p2->dtype &&
      p1->layout == p2->layout;
  for (int l = 0; same && l < p1->num_levels; l++) same = p1->H[l] == p2->H[l] && p1->W[l] == p2->W[l];
  D2_CHECK_ARG(same, "roi_pooler_forward_pair: the two poolers must read the same feature maps");
  bool rule = p1->sampling_ratio == p2->sampling_ratio && p1->aligned == p2->aligned && p1->roi_rounding == p2->roi_rounding &&
      p1->min_level == p2->min_level &&
      p1->max_level == p2->max_level && p1->canonical_level == p2->canonical_level &&
      p1->canonical_box_size == p2->canonical_box_size;
  for (int l = 0; rule && l < p1->num_levels; l++) rule = p1->spatial_scale[l] == p2->spatial_scale[l];
  static const bool off = d2_prof_env("D2AMD_POOL_NO_PAIR") != nullptr;
  if (off || !rule || K1 == 0 || K2 == 0 || !pooler_fused_ok(p1) || !pooler_fused_ok(p2) || p1->layout != D2AMD_NHWC ||
      (long)p1->N * p1->C == 0) {
    set_error("roi_pooler_forward_pair: outside the paired forward (NHWC, both K > 0, the same level rule and sampling)");
    return D2AMD_EUNSUPPORTED;
  }
  for (int l = 0; l < p1->num_levels; l++)
    D2_CHECK_ARG(inputs[l] != nullptr || (long)p1->N * p1->H[l] * p1->W[l] == 0, "roi_pooler_forward_pair: null level %d", l);
  return D2_DISPATCH_DTYPE(p1->dtype, [&]() -> int {
    const PoolFwdPairCall pc{p2, rois2, output2, K2, rec_plan};
    return pool_fwd_impl<scalar_t>(p1, inputs, rois1, output1, K1, (hipStream_t)stream, nullptr, &pc);
  });
}
extern "C" int d2amd_roi_pooler_forward_pair(const d2amd_pooler_params* p1, const void* const* inputs, const float* rois1,
                                             void* output1, int K1, const d2amd_pooler_params* p2, const float* rois2,
                                             void* output2, int K2, void* stream) {
  return pooler_forward_pair_entry(p1, inputs, rois1, output1, K1, p2, rois2, output2, K2, nullptr, stream);
}

extern "C" size_t d2amd_roi_pooler_forward_workspace_bytes(int K) { return (size_t)(K > 0 ? K : 1) * sizeof(int); }

extern "C" int d2amd_roi_pooler_forward_ordered(const d2amd_pooler_params* p, const void* const* inputs,
                                                const float* rois, void* output, int K, void* workspace,
                                                size_t workspace_bytes, void* stream) {
  int rc = check_pooler(p, "roi_pooler_forward");
  if (rc) return rc;
  int* perm = (K > 0 && rois) ? roi_order_ws(p, K, workspace, workspace_bytes) : nullptr;
  if (perm) {
    const PoolLevels L = make_levels(p, inputs, K);
    hipLaunchKernelGGL(roi_order_kernel<false>, dim3(1), dim3(1024), 0, (hipStream_t)stream, L, ImgBoxes{},
                       const_cast<float*>(rois), K, perm);
    D2_LAUNCH_OK();
  }
  return pooler_forward_entry(p, inputs, rois, output, K, perm, stream);
}

static int box_lists_arg(ImgBoxes& e, long& K, const float* const* boxes, const int* counts, int num_images) {
  D2_CHECK_ARG(num_images >= 0 && num_images <= D2AMD_POOLER_MAX_IMAGES && (num_images == 0 || (boxes && counts)),
               "roi_pooler_forward_box_lists: %d images (max %d)", num_images, D2AMD_POOLER_MAX_IMAGES);
  e = ImgBoxes{};
  e.n = num_images;
  K = 0;
  for (int i = 0; i < num_images; i++) {
    D2_CHECK_ARG(counts[i] >= 0 && (counts[i] == 0 || (boxes[i] && ((uintptr_t)boxes[i] & 15) == 0)),
                 "roi_pooler_forward_box_lists: image %d: bad count / null or unaligned boxes", i);
    K += counts[i];
    e.end[i] = (int)K;
    e.ptr[i] = boxes[i];
  }
  D2_CHECK_ARG(K < (1l << 31), "roi_pooler_forward_box_lists: too many boxes");
  return D2AMD_OK;
}

extern "C" int d2amd_roi_pooler_forward_box_lists(const d2amd_pooler_params* p, const void* const* inputs,
                                                  const float* const* boxes, const int* counts, int num_images,
                                                  float* rois_out, void* output, void* stream) {
  ImgBoxes e;
  long K;
  const int rc = box_lists_arg(e, K, boxes, counts, num_images);
  if (rc) return rc;
  if (K == 0) return D2AMD_OK;
  D2_CHECK_ARG(rois_out != nullptr, "roi_pooler_forward_box_lists: null rois_out");
  hipLaunchKernelGGL(box_lists_to_rois_kernel, dim3(cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, e, (int)K, rois_out);
  D2_LAUNCH_OK();
  return d2amd_roi_pooler_forward(p, inputs, rois_out, output, (int)K, stream);
}

// d2amd_roi_pooler_forward_pair for box lists that were never concatenated: ONE conversion launch for both lists (rois1_out
// / rois2_out, needed again by the backward), then the paired forward -- or, where that does not apply, the two plain
// forwards: both outputs are always produced.
extern "C" int d2amd_roi_pooler_forward_pair_box_lists(const d2amd_pooler_params* p1, const void* const* inputs,
                                                       const float* const* boxes1, const int* counts1, float* rois1_out,
                                                       void* output1, const d2amd_pooler_params* p2,
                                                       const float* const* boxes2, const int* counts2, float* rois2_out,
                                                       void* output2, int num_images, void* stream) {
  ImgBoxes e1, e2;
  long K1, K2;
  int rc = box_lists_arg(e1, K1, boxes1, counts1, num_images);
  if (rc) return rc;
  rc = box_lists_arg(e2, K2, boxes2, counts2, num_images);
  if (rc) return rc;
  D2_CHECK_ARG((K1 == 0 || rois1_out) && (K2 == 0 || rois2_out), "roi_pooler_forward_pair_box_lists: null rois_out");
  if (K1 + K2 > 0) {
    hipLaunchKernelGGL(box_lists_to_rois_pair_kernel, dim3(cdiv(K1 + K2, 256)), dim3(256), 0, (hipStream_t)stream, e1, (int)K1,
                       rois1_out, e2, (int)K2, rois2_out);
    D2_LAUNCH_OK();
  }
  rc = d2amd_roi_pooler_forward_pair(p1, inputs, rois1_out, output1, (int)K1, p2, rois2_out, output2, (int)K2, stream);
  if (rc != D2AMD_EUNSUPPORTED) return rc;
  rc = d2amd_roi_pooler_forward(p1, inputs, rois1_out, output1, (int)K1, stream);
  if (rc) return rc;
  return d2amd_roi_pooler_forward(p2, inputs, rois2_out, output2, (int)K2, stream);
}

extern "C" int d2amd_roi_pooler_forward_box_lists_ordered(const d2amd_pooler_params* p, const void* const* inputs,
                                                          const float* const* boxes, const int* counts,
                                                          int num_images, float* rois_out, void* output,
                                                          void* workspace, size_t workspace_bytes, void* stream) {
  ImgBoxes e;
  long K;
  int rc = box_lists_arg(e, K, boxes, counts, num_images);
  if (rc) return rc;
  if (K == 0) return D2AMD_OK;
  D2_CHECK_ARG(rois_out != nullptr, "roi_pooler_forward_box_lists: null rois_out");
  rc = check_pooler(p, "roi_pooler_forward");
  if (rc) return rc;
  int* perm = roi_order_ws(p, (int)K, workspace, workspace_bytes);
  if (perm) {  // one workgroup: box lists -> rois AND the processing order
    const PoolLevels L = make_levels(p, inputs, (int)K);
    hipLaunchKernelGGL(roi_order_kernel<true>, dim3(1), dim3(1024), 0, (hipStream_t)stream, L, e, rois_out, (int)K, perm);
  } else {
    hipLaunchKernelGGL(box_lists_to_rois_kernel, dim3(cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, e, (int)K, rois_out);
  }
  D2_LAUNCH_OK();
  return pooler_forward_entry(p, inputs, rois_out, output, (int)K, perm, stream);
}

extern "C" size_t d2amd_roi_pooler_workspace_bytes(int K) { return (size_t)(K > 0 ? K : 1) * sizeof(RoiRec); }

extern "C" size_t d2amd_roi_pooler_backward_workspace_bytes(const d2amd_pooler_params* p, int K) {
  const size_t need = (size_t)(K > 0 ? K : 1) * sizeof(RoiRec);
  if (check_pooler(p, "roi_pooler_backward_workspace_bytes")) return need;
  const long ntiles = pool_ntiles(p);
  // ... + the scratch slots of split tile lists (16-bit I/O; see SPLIT TILES)
  const int vecn = 8, nslab = cdiv(cdiv(p->C, vecn), LPP);
  const size_t scratch = p->dtype == D2AMD_F32 ? 0 : (size_t)8 * SCR_PER_XCD_MAX * nslab * 32 * (2 * CT) * sizeof(float);
  return pool_al(need) + pool_al((size_t)ntiles * 4) + pool_al((size_t)ntiles * sizeof(int4)) +
      pool_al((size_t)ntiles * TILE_CAP * sizeof(TileEntry)) + pool_queue_bytes(ntiles) + scratch + 256;
}

static int pooler_backward_entry(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                                 void* const* grad_inputs, int K, void* workspace, size_t workspace_bytes,
                                 void* stream, bool accumulate, int phase = 0) {
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
    return pool_bwd_nhwc_impl<scalar_t>(p, grad_output, rois, grad_inputs, K, workspace, workspace_bytes,
                                        (hipStream_t)stream, accumulate, phase);
  });
}

extern "C" int d2amd_roi_pooler_backward(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                                         void* const* grad_inputs, int K, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  return pooler_backward_entry(p, grad_output, rois, grad_inputs, K, workspace, workspace_bytes, stream, false);
}

extern "C" int d2amd_roi_pooler_backward_accumulate(const d2amd_pooler_params* p, const void* grad_output,
                                                    const float* rois, void* const* grad_inputs, int K,
                                                    void* workspace, size_t workspace_bytes, void* stream) {
  return pooler_backward_entry(p, grad_output, rois, grad_inputs, K, workspace, workspace_bytes, stream, true);
}

// Two poolers of the SAME feature maps (the box head's and the mask head's), one gradient, ONE pass: both poolers' ROIs are
// binned together (records, per-tile lists -- the first pooler's entries in front of the second one's -- and one set of
// work queues over the tiles either touches) and pool_bwd_mfma_kernel<T, 8, true, 16> gathers a tile's two sublists
// into the same accumulators.  = d2amd_roi_pooler_backward(p1 ...) followed by d2amd_roi_pooler_backward_accumulate(p2
// ...), except that a tile both touch is rounded to the I/O dtype ONCE (the sum of both gathers in fp32) instead of once
// per pooler and once for their sum.  EUNSUPPORTED (nothing launched) outside the 16-bit MFMA tile gather with bins per
// axis <= 8 (first) and 9..16 (second) and equal level rule / sampling: the caller issues the two calls.
extern "C" size_t d2amd_roi_pooler_backward_pair_workspace_bytes(const d2amd_pooler_params* p1, int K1, int K2) {
  if (check_pooler(p1, "roi_pooler_backward_pair_workspace_bytes")) return 0;
  const long k = (long)(K1 > 0 ? K1 : 0) + (K2 > 0 ? K2 : 0);
  return d2amd_roi_pooler_backward_workspace_bytes(p1, (int)(k < (1l << 30) ? k : (1l << 30))) +
      pool_al((size_t)pool_ntiles(p1) * 4);  // + the per-tile count of the first pooler's entries
}
static int pooler_backward_pair_entry(const d2amd_pooler_params* p1, const void* grad_output1, const float* rois1, int K1,
                                      const d2amd_pooler_params* p2, const void* grad_output2, const float* rois2, int K2,
                                      void* const* grad_inputs, void* workspace, size_t workspace_bytes, int phase,
                                      void* stream, PoolRecPlan* plan = nullptr) {
  int rc = check_pooler(p1, "roi_pooler_backward_pair");
  if (rc) return rc;
  rc = check_pooler(p2, "roi_pooler_backward_pair");
  if (rc) return rc;
  D2_CHECK_ARG(K1 >= 0 && K2 >= 0 && (long)K1 + K2 < (1l << 30), "roi_pooler_backward_pair: bad K");
  D2_CHECK_ARG(grad_inputs && (K1 == 0 || (grad_output1 && rois1)) && (K2 == 0 || (grad_output2 && rois2)),
               "roi_pooler_backward_pair: null pointer");
  bool same = p1->num_levels == p2->num_levels && p1->N == p2->N && p1->C == p2->C && p1->dtype == p2->dtype &&
      p1->layout == p2->layout;
  for (int l = 0; same && l < p1->num_levels; l++) same = p1->H[l] == p2->H[l] && p1->W[l] == p2->W[l];
  D2_CHECK_ARG(same, "roi_pooler_backward_pair: the two poolers must read the same feature maps");
  // one set of records for both: the level rule, the scales and the sampling must be the same
  bool rule = p1->sampling_ratio == p2->sampling_ratio && p1->aligned == p2->aligned && p1->roi_rounding == p2->roi_rounding &&
      p1->min_level == p2->min_level &&
      p1->max_level == p2->max_level && p1->canonical_level == p2->canonical_level &&
      p1->canonical_box_size == p2->canonical_box_size;
  for (int l = 0; rule && l < p1->num_levels; l++) rule = p1->spatial_scale[l] == p2->spatial_scale[l];
  static const bool off = d2_prof_env("D2AMD_POOL_NO_PAIR") != nullptr;
  if (off || !rule || K1 == 0 || K2 == 0 || !pooler_fused_ok(p1) || !pooler_fused_ok(p2) || p1->layout != D2AMD_NHWC ||
      p1->dtype == D2AMD_F32 || (long)p1->N * p1->C == 0 || ((uintptr_t)grad_output2 & 15) != 0) {
    set_error("roi_pooler_backward_pair: outside the paired 16-bit tile gather (16-bit NHWC, pooled sizes <= %d, the same "
              "level rule and sampling)", MAXP);
    return D2AMD_EUNSUPPORTED;
  }
  return D2_DISPATCH_DTYPE(p1->dtype, [&]() -> int {
    const PoolPairCall pc{p2, grad_output2, rois2, K2};
    hipStream_t s = (hipStream_t)stream;
    // the call must take the persistent MFMA tile gather (probe: nothing is launched)
    const int r = pool_bwd_nhwc_impl<scalar_t>(p1, grad_output1, rois1, grad_inputs, K1, workspace, workspace_bytes, s, false,
                                               phase, &pc, true);
    if (r) {
      set_error("roi_pooler_backward_pair: outside the paired 16-bit tile gather (workspace, alignment or a profiling switch)");
      return r;
    }
    return pool_bwd_nhwc_impl<scalar_t>(p1, grad_output1, rois1, grad_inputs, K1, workspace, workspace_bytes, s, false, phase,
                                        &pc, false, plan);
  });
}
extern "C" int d2amd_roi_pooler_backward_pair(const d2amd_pooler_params* p1, const void* grad_output1, const float* rois1,
                                              int K1, const d2amd_pooler_params* p2, const void* grad_output2,
                                              const float* rois2, int K2, void* const* grad_inputs, void* workspace,
                                              size_t workspace_bytes, void* stream) {
  return pooler_backward_pair_entry(p1, grad_output1, rois1, K1, p2, grad_output2, rois2, K2, grad_inputs, workspace,
                                    workspace_bytes, 0, stream);
}
// The paired backward in two calls (as d2amd_roi_pooler_backward_phase): phase 1 bins both ROI sets -- it reads the rois,
// writes the workspace and ZERO-FILLS the tiles of grad_inputs no ROI touches, so the gradient tensors must exist, but no
// gradient value is needed: it can run beside the poolers' forward (grad_outputN: any pointers of the later ones'
// alignment class) -- phase 2, with the same arguments and workspace, is the tile gather alone.
extern "C" int d2amd_roi_pooler_backward_pair_phase(const d2amd_pooler_params* p1, const void* grad_output1,
                                                    const float* rois1, int K1, const d2amd_pooler_params* p2,
                                                    const void* grad_output2, const float* rois2, int K2,
                                                    void* const* grad_inputs, void* workspace, size_t workspace_bytes,
                                                    int phase, void* stream) {
  D2_CHECK_ARG(phase == 1 || phase == 2 || phase == 5, "roi_pooler_backward_pair_phase: phase must be 1 (bin + zero fill), 2 "
               "(gather) or 5 (lists + gather behind d2amd_roi_pooler_forward_pair_records)");
  return pooler_backward_pair_entry(p1, grad_output1, rois1, K1, p2, grad_output2, rois2, K2, grad_inputs, workspace,
                                    workspace_bytes, phase, stream);
}

extern "C" int d2amd_roi_pooler_backward_phase(const d2amd_pooler_params* p, const void* grad_output,
                                               const float* rois, void* const* grad_inputs, int K, void* workspace,
                                               size_t workspace_bytes, int phase, void* stream) {
  D2_CHECK_ARG(phase >= 1 && phase <= 3, "roi_pooler_backward_phase: phase must be 1 (bin), 2 (gather, adding) or 3 "
               "(gather, writing)");
  return pooler_backward_entry(p, grad_output, rois, grad_inputs, K, workspace, workspace_bytes, stream, true, phase);
}

// The paired forward that ALSO prepares the paired backward of the same ROIs (r06): the records and the reset of the work
// queues -- roi_records_kernel, the first launch of d2amd_roi_pooler_backward_pair -- are done by the forward's workgroups
// into `bwd_workspace` (d2amd_roi_pooler_backward_pair_workspace_bytes(p1, K1, K2) bytes, kept by the caller until the
// backward).  *records_written = 1: call d2amd_roi_pooler_backward_pair_phase(..., phase 5) with that workspace and the same
// ROIs (it starts with the tile lists); 0: the workspace was not touched (configuration outside the paired tile gather):
// call d2amd_roi_pooler_backward_pair as usual.  The outputs are d2amd_roi_pooler_forward_pair's either way.
extern "C" int d2amd_roi_pooler_forward_pair_records(const d2amd_pooler_params* p1, const void* const* inputs,
                                                     const float* rois1, void* output1, int K1,
                                                     const d2amd_pooler_params* p2, const float* rois2, void* output2,
                                                     int K2, void* bwd_workspace, size_t bwd_workspace_bytes,
                                                     int* records_written, void* stream) {
  D2_CHECK_ARG(records_written != nullptr, "roi_pooler_forward_pair_records: null pointer");
  *records_written = 0;
  PoolRecPlan plan{};
  bool have = false;
  if (bwd_workspace && inputs && rois1 && rois2 && K1 > 0 && K2 > 0 && check_pooler(p1, "roi_pooler_forward_pair_records") == 0 &&
      check_pooler(p2, "roi_pooler_forward_pair_records") == 0) {
    // (no gradient exists yet: the workspace stands in for both dY pointers and the features for the gradient tensors --
    // only their alignment class is looked at; phase 4 launches nothing)
    have = pooler_backward_pair_entry(p1, bwd_workspace, rois1, K1, p2, bwd_workspace, rois2, K2, (void* const*)inputs,
                                      bwd_workspace, bwd_workspace_bytes, 4, stream, &plan) == D2AMD_OK;
  }
  const int rc = pooler_forward_pair_entry(p1, inputs, rois1, output1, K1, p2, rois2, output2, K2, have ? &plan : nullptr, stream);
  if (rc == D2AMD_OK && have) *records_written = 1;
  return rc;
}
