// NMS / batched NMS / rotated NMS, entirely on the device (no D2H copy of the bitmask, no
// CPU sweep -- cf. the reference's csrc/nms_rotated/nms_rotated_cuda.cu:114-137 and
// torchvision's identical pattern).
//
// Pipeline (all on `stream`):
//   1. ORDER.  n <= RANK_MAX_N: one brute-force ranking kernel -- every box counts how many
//      64-bit composite keys (category | ~score | index) precede its own (n^2 compares from LDS,
//      ~8 us at n = 8,819) and scatters itself straight to its class-major / score-descending
//      slot; this replaces two radix sorts + three gather kernels (rocPRIM takes its merge-sort
//      path at this size: 8 launches, ~90 us).  Larger n: stable radix sort by score, then by
//      category (16 bits), then gathers.
//   2. wavefront bitmask kernel: one 64-lane wave per 64x64 tile of the (sorted) IoU matrix,
//        lane = row box, one uint64 word per lane; only tiles on/above the diagonal whose
//        category ranges overlap are evaluated (pair count = sum_c n_c^2/2, not N^2/2).
//        Diagonal tiles also emit the TRANSPOSED word (which earlier boxes of the tile suppress
//        this one), which turns step 3's diagonal resolution into a lane-parallel fixed point.
//   3. greedy reduction: one workgroup per category segment; see nms_reduce_kernel.
//   4. ordered compaction back to rank order -> original indices, count.
// Bit-exactness: IoU arithmetic is evaluated exactly as torchvision's CPU nms / the reference's
// nms_rotated_cpu.cpp (fp32, no FMA contraction, IEEE divide, threshold compare in double).
#pragma clang fp contract(off)
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "common.h"
#include "rotated_iou.h"

namespace d2amd {

typedef unsigned long long u64;

constexpr int RANK_MAX_N = 12288;  // brute-force ranking up to here (index must fit 16 bits)

struct NmsWorkspace {
  float* keys_out;    // [n] sorted scores                        (radix path)
  int* iota;          // [n] 0..n-1                               (radix path)
  int* order;         // [n] rank -> original index
  uint32_t* cls_r;    // [n] category in rank order               (radix path)
  uint32_t* cls_s;    // [n] category in segment order
  int* rankpos;       // [n] segment position -> rank
  float* boxes_s;     // [n64 * 8] boxes in segment order (aligned: float4; rotated: 5 of 8 floats)
  u64* mask;          // [n64 * wcap]
  u64* diagT;         // [n64] transposed diagonal word per row
  u64* keepbits;      // [nblocks]   } zeroed together
  int* counters;      // [4]: nseg, error flags }
  uint8_t* flag_r;    // [n] kept flag in rank order              (radix path)
  int* seg_start;     // [65536]
  void* sort_temp;
  size_t sort_temp_bytes;
  size_t zero_bytes;  // keepbits + counters
  size_t total;
};

static size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

static int wcap_for(int64_t n, int64_t max_per_class) {
  int64_t m = (max_per_class <= 0 || max_per_class > n) ? n : max_per_class;
  int64_t nblocks = (n + 63) / 64;
  int64_t w = (m + 63) / 64 + 1;
  return (int)(w < nblocks ? w : nblocks);
}

static size_t sort_temp_bytes(int64_t n) {
  if (n <= RANK_MAX_N) return 0;
  size_t a = 0, b = 0;
  (void)rocprim::radix_sort_pairs_desc(nullptr, a, (const float*)nullptr, (float*)nullptr, (const int*)nullptr,
                                 (int*)nullptr, (unsigned)n, 0, 32, (hipStream_t)0, false);
  (void)rocprim::radix_sort_pairs(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int*)nullptr,
                            (int*)nullptr, (unsigned)n, 0, 16, (hipStream_t)0, false);
  return a > b ? a : b;
}

static void carve(NmsWorkspace& w, void* base, int64_t n, int wcap) {
  char* p = (char*)base;
  size_t off = 0;
  int64_t n64 = (n + 63) / 64 * 64;
  auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
  w.keys_out = (float*)take(n * 4);
  w.iota = (int*)take(n * 4);
  w.order = (int*)take(n * 4);
  w.cls_r = (uint32_t*)take(n * 4);
  w.cls_s = (uint32_t*)take(n * 4);
  w.rankpos = (int*)take(n * 4);
  w.boxes_s = (float*)take(n64 * 8 * 4);
  w.mask = (u64*)take((size_t)n64 * wcap * 8);
  w.diagT = (u64*)take((size_t)n64 * 8);
  const size_t z0 = off;
  w.keepbits = (u64*)take((n64 / 64) * 8);
  w.counters = (int*)take(4 * 4);
  w.zero_bytes = off - z0;
  w.flag_r = (uint8_t*)take(n);
  w.seg_start = (int*)take(65536 * 4);
  w.sort_temp_bytes = sort_temp_bytes(n);
  w.sort_temp = take(w.sort_temp_bytes);
  w.total = off;
}

// ---- step 1a: brute-force ranking (n <= RANK_MAX_N) -------------------------------------------
// S = (~orderable(score)) << 16 | index        ascending S  = descending score, ties by lower index
// K = category << 48 | S                        ascending K  = class-major, score-descending inside
// (the orderable transform is the one radix sort uses: sign bit flipped for x >= 0, all bits for x < 0)
__device__ __forceinline__ u64 score_key(float s, int i) {
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((u64)(~u) << 16) | (u64)(uint32_t)i;
}

constexpr int RK_THREADS = 1024;
constexpr int RK_IB = RK_THREADS / 32;  // boxes ranked per workgroup (one per half-wave)

// All n keys are converted and staged in LDS by ONE bulk pass (every load in flight at once: one
// memory latency per workgroup), then each half-wave ranks one box against all of them from LDS.
template <int BW>
__global__ __launch_bounds__(RK_THREADS) void nms_rank_kernel(
    const float* __restrict__ boxes, const float* __restrict__ scores, const int64_t* __restrict__ idxs, int n,
    int* __restrict__ order, int* __restrict__ rankpos, uint32_t* __restrict__ cls_s, float* __restrict__ boxes_s,
    int* __restrict__ counters, int rounds) {
  __shared__ u64 Ks[RANK_MAX_N];  // 96 KiB
  const int tid = threadIdx.x, jp = tid & 31, il = tid >> 5;
  constexpr int PER = RANK_MAX_N / RK_THREADS;  // 12
  {
    float sc[PER];
    int64_t cc[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {  // raw clamped loads first, conversion after: keeps them all in flight
      const int j = min(tid + q * RK_THREADS, n - 1);
      sc[q] = scores[j];
      cc[q] = idxs ? idxs[j] : 0;
    }
    bool bad = false;
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int j = tid + q * RK_THREADS;
      int64_t c = cc[q];
      if (j < n && (c < 0 || c > 65535)) bad = true;
      if (c < 0 || c > 65535) c = 0;
      if (j < n) Ks[j] = ((u64)c << 48) | score_key(sc[q], j);
    }
    if (bad && blockIdx.x == 0) atomicOr(&counters[1], 2);
  }
  __syncthreads();
  // `rounds` boxes per half-wave: the launch has at most ~one workgroup per CU (the 96 KiB key table
  // allows only one resident workgroup per CU, so a 257th workgroup would cost a whole second round)
  constexpr int STRIDE = BW == 4 ? 4 : 8;
  const u64 M48 = 0x0000ffffffffffffull;
  for (int t = 0; t < rounds; t++) {
    const int i = (blockIdx.x * rounds + t) * RK_IB + il;
    if (i >= n) break;  // uniform per half-wave
    const u64 Ki = Ks[i];
    const u64 Si = Ki & M48;
    int r_s = 0, r_cm = 0;
    for (int j = jp; j < n; j += 32) {
      const u64 K = Ks[j];
      r_cm += (K < Ki) ? 1 : 0;          // class-major position
      r_s += ((K & M48) < Si) ? 1 : 0;   // global score rank (ties: lower index first)
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {  // the 32 lanes of a half-wave share box i
      r_s += __shfl_xor(r_s, o);
      r_cm += __shfl_xor(r_cm, o);
    }
    if (jp < BW) boxes_s[(long)r_cm * STRIDE + jp] = boxes[(long)i * BW + jp];
    if (jp == 0) {
      order[r_s] = i;
      rankpos[r_cm] = r_s;
      cls_s[r_cm] = (uint32_t)(Ki >> 48);
    }
  }
}

// segment starts of the class-major sequence (small path; the radix path finds them while gathering)
__global__ void nms_segments_kernel(const uint32_t* __restrict__ cls_s, int n, int* seg_start, int* counters) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t c = cls_s[p], cprev = cls_s[max(p - 1, 0)];
  if (p == 0 || c != cprev) {
    int pos = atomicAdd(&counters[0], 1);
    seg_start[pos] = p;
  }
}

// ---- step 1b helpers (radix path) ---------------------------------------------------------------
__global__ void nms_init_kernel(int* iota, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) iota[i] = i;
}

__global__ void nms_gather_cls_kernel(const int64_t* __restrict__ idxs, const int* __restrict__ order, int n,
                                      uint32_t* __restrict__ cls_r, int* counters) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int64_t c = idxs[order[r]];
  if (c < 0 || c > 65535) { atomicOr(&counters[1], 2); c = 0; }
  cls_r[r] = (uint32_t)c;
}

// boxes into segment order (+ segment starts).  BW = 4 (xyxy) or 5 (cxcywha); stored stride 4 / 8.
template <int BW>
__global__ void nms_gather_boxes_kernel(const float* __restrict__ boxes, const int* __restrict__ order,
                                        const int* __restrict__ rankpos, const uint32_t* __restrict__ cls_s,
                                        int n, float* __restrict__ boxes_s, int* seg_start, int* counters) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int r = rankpos ? rankpos[p] : p;
  int src = order[r];
  constexpr int STRIDE = BW == 4 ? 4 : 8;
#pragma unroll
  for (int k = 0; k < BW; k++) boxes_s[(long)p * STRIDE + k] = boxes[(long)src * BW + k];
  bool start = cls_s ? (p == 0 || cls_s[p] != cls_s[p - 1]) : (p == 0);
  if (start) {
    int pos = atomicAdd(&counters[0], 1);
    seg_start[pos] = p;
  }
}

// ---- step 2: wavefront bitmask ------------------------------------------------------------
__device__ __forceinline__ float bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// The j loop is deliberately NOT fully unrolled: every wave runs the body exactly once per j, so a
// 64x unrolled body (20 KB of straight-line code) made the kernel instruction-fetch bound (31 us
// for 2,640 live tiles; the arithmetic is ~3 us).
template <bool ROT>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes_s,
                                                      const uint32_t* __restrict__ cls_s, int n, int wcap,
                                                      double thr, u64* __restrict__ mask, u64* __restrict__ diagT) {
  const int rb = blockIdx.x, w = blockIdx.y;
  const int lane = threadIdx.x;
  const int cb = rb + w;
  const int row = rb * 64 + lane;
  const int col0 = cb * 64;
  const int nblocks = (n + 63) >> 6;
  if (cb >= nblocks) return;  // never read by the reduction
  u64 word = 0, wordT = 0;
  bool live = true;
  if (cls_s) {
    // categories ascend along the sorted sequence: tile is empty unless ranges touch
    uint32_t row_last = cls_s[min(rb * 64 + 63, n - 1)];
    uint32_t col_first = cls_s[col0];
    live = col_first <= row_last;
  }
  if (live) {
    const int rrow = min(row, n - 1), rcol = min(col0 + lane, n - 1);
    uint32_t my_cls = cls_s ? cls_s[rrow] : 0u;
    uint32_t col_cls = cls_s ? cls_s[rcol] : 0u;
    if constexpr (!ROT) {
      const float4 rbx = reinterpret_cast<const float4*>(boxes_s)[rrow];
      const float4 cbx = reinterpret_cast<const float4*>(boxes_s)[rcol];
      const float iarea = (rbx.z - rbx.x) * (rbx.w - rbx.y);
      const float carea = (cbx.z - cbx.x) * (cbx.w - cbx.y);
#pragma unroll 4
      for (int j = 0; j < 64; j++) {
        const float jx1 = bcast(cbx.x, j), jy1 = bcast(cbx.y, j), jx2 = bcast(cbx.z, j), jy2 = bcast(cbx.w, j);
        const float jarea = bcast(carea, j);
        const uint32_t jcls = (uint32_t)__builtin_amdgcn_readlane((int)col_cls, j);
        // torchvision nms_kernel_impl: std::max(a,b) = (a<b)?b:a, std::min(a,b) = (b<a)?b:a
        float xx1 = (rbx.x < jx1) ? jx1 : rbx.x;
        float yy1 = (rbx.y < jy1) ? jy1 : rbx.y;
        float xx2 = (jx2 < rbx.z) ? jx2 : rbx.z;
        float yy2 = (jy2 < rbx.w) ? jy2 : rbx.w;
        float ww = xx2 - xx1, hh = yy2 - yy1;
        ww = (0.f < ww) ? ww : 0.f;
        hh = (0.f < hh) ? hh : 0.f;
        float inter = ww * hh;
        // max/min, the product and a + b are symmetric in (row, col), so one value serves both
        // the word (row suppresses later col) and, on diagonal tiles, the transposed word
        float ovr = inter / (iarea + jarea - inter);
        const bool hit = ((double)ovr > thr) && (jcls == my_cls) && (col0 + j < n) && (row < n);
        word |= (hit && (col0 + j > row)) ? (1ull << j) : 0ull;
        wordT |= (hit && (col0 + j < row)) ? (1ull << j) : 0ull;
      }
    } else {
      __shared__ RotIouScratch<64> S;
      float rbx[5], cbx[5];
#pragma unroll
      for (int k = 0; k < 5; k++) { rbx[k] = boxes_s[(long)rrow * 8 + k]; cbx[k] = boxes_s[(long)rcol * 8 + k]; }
      for (int j = 0; j < 64; j++) {
        if (col0 + j >= n) break;  // uniform
        float jb[5];
#pragma unroll
        for (int k = 0; k < 5; k++) jb[k] = __shfl(cbx[k], j);
        const uint32_t jcls = (uint32_t)__shfl((int)col_cls, j);
        // every lane evaluates (uniform control flow); cheap rejection by category / triangle first.
        // The reference evaluates iou(kept box, later box) (nms_rotated_cpu.cpp:45-54) and the
        // polygon clip is not symmetric in floating point: keep that argument order in both words.
        const bool same = (jcls == my_cls) && (row < n);
        const bool cand = same && (col0 + j > row);
        float ovr = 0.f;
        if (cand) ovr = single_box_iou_rotated<64>(rbx, jb, S, lane);
        word |= (cand && ((double)ovr >= thr)) ? (1ull << j) : 0ull;  // nms_rotated_cpu.cpp:54
        if (w == 0) {  // uniform
          const bool candT = same && (col0 + j < row);
          float ovrT = 0.f;
          if (candT) ovrT = single_box_iou_rotated<64>(jb, rbx, S, lane);
          wordT |= (candT && ((double)ovrT >= thr)) ? (1ull << j) : 0ull;
        }
      }
    }
  }
  if (row < n) {
    mask[(long)row * wcap + w] = word;
    if (w == 0) diagT[row] = wordT;
  }
}

// ---- step 3: greedy reduction ----------------------------------------------------------------
// One 320-thread workgroup (5 waves) per category segment, walking its 64-row blocks in order.
//   wave 0     resolves the diagonal block: kept_j = cand_j && !(DT_j & kept), iterated as a
//              lane-parallel fixed point (DT_j = transposed diagonal word; position t is final
//              after t iterations and the typical depth is a handful -- v0/v1 ran a 64-step
//              scalar chain), publishes `kept`, and ORs word 1 of the kept rows into
//              removed[b+1] itself: the only data the NEXT diagonal block needs from this one.
//              Its inputs (DT, word 1) are staged in LDS a window of RED_WIN blocks at a time, so
//              the serial chain never touches global memory.
//   waves 1-4  "pushers": wave g owns the blocks b = g (mod 4), all 64 rows of it (128 VGPRs),
//              lane = word 2 + lane.  A pusher loads its whole block, then sits
//              out three barriers before `kept` of that block exists -- prefetch depth comes from
//              the other pushers' loads being in flight meanwhile.  (A register ring in one wave
//              does not work: hipcc emits s_waitcnt vmcnt(0) at every ring read inside divergent
//              / looped code, which drains the loads just issued -- measured 3 us per block.)
// One barrier per block; the pushers' LDS ORs land one barrier before wave 0 needs them.
constexpr int RED_THREADS = 320;
constexpr int RED_GROUPS = 4;
constexpr int RED_PUSH_ROWS = 64;   // rows per pusher wave (one wave per group)
constexpr int RED_WIN = 64;         // blocks of (DT, word 1) staged in LDS at a time (2 x 32 KiB)

__global__ __launch_bounds__(RED_THREADS) void nms_reduce_kernel(const u64* __restrict__ mask,
                                                                 const u64* __restrict__ diagT,
                                                                 const uint32_t* __restrict__ cls_s, int n, int wcap,
                                                                 int max_per_class, const int* __restrict__ seg_start,
                                                                 int* counters, u64* keepbits) {
  extern __shared__ __attribute__((aligned(16))) u64 removed[];  // [wcap]
  __shared__ u64 dt_s[RED_WIN * 64], w1_s[RED_WIN * 64];
  __shared__ u64 kept_s[2];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int grp = wid - 1;  // pushers only
  const int nseg = cls_s ? counters[0] : 1;
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    int s = cls_s ? seg_start[seg] : 0;
    int e = n;
    if (cls_s) {
      // upper bound of this category in the ascending cls_s (uniform work, done by every lane)
      uint32_t c = cls_s[s];
      int lo = s, hi = n;
      while (lo < hi) { int mid = (lo + hi) >> 1; if (cls_s[mid] <= c) lo = mid + 1; else hi = mid; }
      e = lo;
    }
    if (e - s > max_per_class) { if (tid == 0) atomicOr(&counters[1], 1); continue; }
    const int b0 = s >> 6, b1 = (e - 1) >> 6;
    const int nb = b1 - b0 + 1;  // <= wcap by construction
    __syncthreads();
    for (int w = tid; w < nb; w += RED_THREADS) removed[w] = 0;

    u64 rows[RED_PUSH_ROWS];  // pusher: word 2 + lane of the 64 rows of the block it currently owns
    const int wi = min(2 + lane, wcap - 1);
    auto load_block = [&](int b) {  // unconditional clamped loads, raw (validity applied at use)
#pragma unroll
      for (int r = 0; r < RED_PUSH_ROWS; r++) {
        // rows up to n64 - 1 are allocated (never kept if >= n): no clamp, so the row base is a
        // wave-uniform SGPR address and the lane offset is the only VGPR
        const u64* rowp = mask + (long)(b * 64 + r) * wcap;
        rows[r] = rowp[wi];
      }
    };
    if (wid != 0 && b0 + grp <= b1) load_block(b0 + grp);

    for (int b = b0; b <= b1; b++) {
      const int rel = b - b0;
      if ((rel % RED_WIN) == 0) {  // stage the next window of wave 0's inputs (uniform)
        __syncthreads();           // wave 0 is done with the previous window
        const int rows_w = min(RED_WIN, b1 - b + 1) * 64;
        for (int q = tid; q < rows_w; q += RED_THREADS) {
          const int row = b * 64 + q;
          const long rc = min(row, n - 1);
          const u64 d = diagT[rc], w1 = mask[rc * wcap + min(1, wcap - 1)];
          const bool valid = row >= s && row < e;
          dt_s[q] = valid ? d : 0ull;
          w1_s[q] = (valid && (row >> 6) < b1) ? w1 : 0ull;
        }
        __syncthreads();
      }
      if (wid == 0) {
        const int row = b * 64 + lane;
        const bool valid = row >= s && row < e;
        const u64 cand = ~removed[rel] & __ballot(valid);
        const u64 dt = dt_s[(rel % RED_WIN) * 64 + lane];
        const bool cj = (cand >> lane) & 1ull;
        u64 kept = cand;
        for (;;) {
          const u64 nk = __ballot(cj && (dt & kept) == 0ull);
          if (nk == kept) break;
          kept = nk;
        }
        if (lane == 0) {
          kept_s[b & 1] = kept;
          if (kept) atomicOr(&keepbits[b], kept);
        }
        const u64 w1 = w1_s[(rel % RED_WIN) * 64 + lane];
        if (((kept >> lane) & 1ull) && w1) atomicOr(&removed[rel + 1], w1);
      }
      __syncthreads();
      if (wid != 0 && (rel % RED_GROUPS) == grp) {  // this group's block: `kept` is known now
        const u64 kept = kept_s[b & 1];
        const int nlater = b1 - b;
        u64 acc = 0;
#pragma unroll
        for (int r = 0; r < RED_PUSH_ROWS; r++)  // kept rows lie in [s, e) by construction
          acc |= ((kept >> r) & 1ull) ? rows[r] : 0ull;
        if ((2 + lane) <= nlater && acc) atomicOr(&removed[rel + 2 + lane], acc);
        // categories with more than ~4200 boxes: remaining words, fetched now that `kept` is known
        for (int w = 66 + lane; w <= nlater; w += 64) {
          u64 a2 = 0;
          for (int r = 0; r < RED_PUSH_ROWS; r++) {
            if ((kept >> r) & 1ull) a2 |= mask[((long)b * 64 + r) * wcap + w];
          }
          if (a2) atomicOr(&removed[rel + w], a2);
        }
        if (b + RED_GROUPS <= b1) load_block(b + RED_GROUPS);  // in flight for the next 3 barriers
      }
    }
  }
}

// ---- step 4: compaction ------------------------------------------------------------------------
// small n: one workgroup does scatter-to-rank-order and ordered compaction out of LDS
constexpr int FIN_THREADS = 1024;
__global__ __launch_bounds__(FIN_THREADS) void nms_finalize_small_kernel(
    const u64* __restrict__ keepbits, const int* __restrict__ rankpos, const int* __restrict__ order, int n,
    int64_t* __restrict__ keep_out, const int* __restrict__ counters, int64_t* __restrict__ result) {
  __shared__ uint8_t flags[RANK_MAX_N];
  __shared__ int wave_tot[FIN_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int p = tid; p < n; p += FIN_THREADS) {
    const bool kept = (keepbits[p >> 6] >> (p & 63)) & 1ull;
    flags[rankpos ? rankpos[p] : p] = kept ? 1 : 0;
  }
  __syncthreads();
  constexpr int CH = RANK_MAX_N / FIN_THREADS;  // 12 consecutive ranks per thread
  const int r0 = tid * CH;
  int ord[CH];
  int cnt = 0;
  uint32_t fl = 0;
#pragma unroll
  for (int q = 0; q < CH; q++) {
    const int r = r0 + q;
    ord[q] = order[min(r, n - 1)];  // unconditional (clamped) loads, all in flight together
    const bool f = r < n && flags[min(r, n - 1)];
    fl |= f ? (1u << q) : 0u;
    cnt += f ? 1 : 0;
  }
  int incl = cnt;  // inclusive scan inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wave_tot[wid] = incl;
  __syncthreads();
  int off = incl - cnt;
  for (int w = 0; w < wid; w++) off += wave_tot[w];
#pragma unroll
  for (int q = 0; q < CH; q++)
    if ((fl >> q) & 1u) keep_out[off++] = (int64_t)ord[q];
  if (tid == FIN_THREADS - 1) { result[0] = off; result[1] = counters[1]; }
}

__global__ void nms_scatter_flags_kernel(const u64* __restrict__ keepbits, const int* __restrict__ rankpos, int n,
                                         uint8_t* __restrict__ flag_r) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  bool kept = (keepbits[p >> 6] >> (p & 63)) & 1ull;
  int r = rankpos ? rankpos[p] : p;
  flag_r[r] = kept ? 1 : 0;
}

constexpr int COMPACT_BLOCK = 1024;
__global__ __launch_bounds__(COMPACT_BLOCK) void nms_compact_kernel(const uint8_t* __restrict__ flag_r,
                                                                    const int* __restrict__ order, int n,
                                                                    int64_t* __restrict__ keep_out,
                                                                    const int* __restrict__ counters,
                                                                    int64_t* __restrict__ result) {
  __shared__ int wave_cnt[COMPACT_BLOCK / 64];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int start = 0; start < n; start += COMPACT_BLOCK) {
    int r = start + tid;
    bool f = r < n && flag_r[r];
    u64 bal = __ballot(f);
    int within = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wid] = __builtin_popcountll(bal);
    __syncthreads();
    int off = base_s;
    for (int k = 0; k < wid; k++) off += wave_cnt[k];
    if (f) keep_out[off + within] = (int64_t)order[r];
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int k = 0; k < COMPACT_BLOCK / 64; k++) t += wave_cnt[k];
      base_s += t;
    }
    __syncthreads();
  }
  if (tid == 0) { result[0] = base_s; result[1] = counters[1]; }
}

}  // namespace d2amd

using namespace d2amd;

extern "C" size_t d2amd_nms_workspace_bytes(int64_t n, int64_t max_per_class, int rotated) {
  (void)rotated;
  if (n <= 0) return 256;
  NmsWorkspace w;
  carve(w, nullptr, n, wcap_for(n, max_per_class));
  return w.total;
}

extern "C" int d2amd_nms(const float* boxes, const float* scores, const int64_t* idxs, int64_t n,
                         double iou_threshold, int rotated, int64_t max_per_class, int64_t* keep_out,
                         int64_t* result, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  D2_CHECK_ARG(n >= 0 && n < (1ll << 31) - 64, "nms: bad n %lld", (long long)n);
  D2_CHECK_ARG(result, "nms: null result");
  if (n == 0) {
    D2_HIP_OK(hipMemsetAsync(result, 0, 16, s));
    return D2AMD_OK;
  }
  D2_CHECK_ARG(boxes && scores && keep_out && workspace, "nms: null pointer");
  const int wcap = wcap_for(n, max_per_class);
  const int mpc = (max_per_class <= 0 || max_per_class > n) ? (int)n : (int)max_per_class;
  NmsWorkspace w;
  carve(w, workspace, n, wcap);
  if (workspace_bytes < w.total) {
    set_error("nms: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    return D2AMD_EWORKSPACE;
  }
  const int N = (int)n, nblocks = (N + 63) / 64;
  const int T = 256;
  const bool small = N <= RANK_MAX_N;
  D2_HIP_OK(hipMemsetAsync(w.keepbits, 0, w.zero_bytes, s));
  const int* rankpos = nullptr;
  const uint32_t* cls_s = nullptr;
  if (small) {
    const int rk_rounds = cdiv(cdiv(N, RK_IB), 240);  // <= 240 workgroups: one round on 256 CUs
    const int rk_grid = cdiv(N, RK_IB * rk_rounds);
    if (rotated)
      hipLaunchKernelGGL((nms_rank_kernel<5>), dim3(rk_grid), dim3(RK_THREADS), 0, s, boxes, scores, idxs, N,
                         w.order, w.rankpos, w.cls_s, w.boxes_s, w.counters, rk_rounds);
    else
      hipLaunchKernelGGL((nms_rank_kernel<4>), dim3(rk_grid), dim3(RK_THREADS), 0, s, boxes, scores, idxs, N,
                         w.order, w.rankpos, w.cls_s, w.boxes_s, w.counters, rk_rounds);
    D2_LAUNCH_OK();
    if (idxs) {
      rankpos = w.rankpos;
      cls_s = w.cls_s;
      hipLaunchKernelGGL(nms_segments_kernel, dim3(cdiv(N, T)), dim3(T), 0, s, w.cls_s, N, w.seg_start, w.counters);
      D2_LAUNCH_OK();
    }
  } else {
    hipLaunchKernelGGL(nms_init_kernel, dim3(cdiv(N, T)), dim3(T), 0, s, w.iota, N);
    D2_LAUNCH_OK();
    size_t tb = w.sort_temp_bytes;
    D2_HIP_OK(rocprim::radix_sort_pairs_desc(w.sort_temp, tb, scores, w.keys_out, w.iota, w.order, (unsigned)N, 0,
                                             32, s, false));
    if (idxs) {
      hipLaunchKernelGGL(nms_gather_cls_kernel, dim3(cdiv(N, T)), dim3(T), 0, s, idxs, w.order, N, w.cls_r,
                         w.counters);
      D2_LAUNCH_OK();
      tb = w.sort_temp_bytes;
      D2_HIP_OK(rocprim::radix_sort_pairs(w.sort_temp, tb, w.cls_r, w.cls_s, w.iota, w.rankpos, (unsigned)N, 0, 16,
                                          s, false));
      rankpos = w.rankpos;
      cls_s = w.cls_s;
    }
    if (rotated)
      hipLaunchKernelGGL((nms_gather_boxes_kernel<5>), dim3(cdiv(N, T)), dim3(T), 0, s, boxes, w.order, rankpos,
                         cls_s, N, w.boxes_s, w.seg_start, w.counters);
    else
      hipLaunchKernelGGL((nms_gather_boxes_kernel<4>), dim3(cdiv(N, T)), dim3(T), 0, s, boxes, w.order, rankpos,
                         cls_s, N, w.boxes_s, w.seg_start, w.counters);
    D2_LAUNCH_OK();
  }
  dim3 mgrid(nblocks, wcap);
  if (rotated)
    hipLaunchKernelGGL((nms_mask_kernel<true>), mgrid, dim3(64), 0, s, w.boxes_s, cls_s, N, wcap, iou_threshold,
                       w.mask, w.diagT);
  else
    hipLaunchKernelGGL((nms_mask_kernel<false>), mgrid, dim3(64), 0, s, w.boxes_s, cls_s, N, wcap, iou_threshold,
                       w.mask, w.diagT);
  D2_LAUNCH_OK();
  const int rgrid = idxs ? 512 : 1;
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(rgrid), dim3(RED_THREADS), (size_t)wcap * 8, s, w.mask, w.diagT, cls_s,
                     N, wcap, mpc, w.seg_start, w.counters, w.keepbits);
  D2_LAUNCH_OK();
  if (small) {
    hipLaunchKernelGGL(nms_finalize_small_kernel, dim3(1), dim3(FIN_THREADS), 0, s, w.keepbits, rankpos, w.order, N,
                       keep_out, w.counters, result);
  } else {
    hipLaunchKernelGGL(nms_scatter_flags_kernel, dim3(cdiv(N, T)), dim3(T), 0, s, w.keepbits, rankpos, N, w.flag_r);
    D2_LAUNCH_OK();
    hipLaunchKernelGGL(nms_compact_kernel, dim3(1), dim3(COMPACT_BLOCK), 0, s, w.flag_r, w.order, N, keep_out,
                       w.counters, result);
  }
  D2_LAUNCH_OK();
  return D2AMD_OK;
}
