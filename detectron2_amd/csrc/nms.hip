// NMS / batched NMS / rotated NMS, entirely on the device (no D2H copy of the bitmask, no
// CPU sweep -- cf. the reference's csrc/nms_rotated/nms_rotated_cuda.cu:114-137 and
// torchvision's identical pattern).
//
// Pipeline (all on `stream`):
//   1. ORDER.  n <= RANK_MAX_N: one brute-force ranking kernel -- every box counts how many
//      64-bit composite keys (category | ~score | index) precede its own (n^2 compares from LDS,
//      ~8 us at n = 8,819) and scatters itself straight to its class-major / score-descending
//      slot; this replaces two radix sorts + three gather kernels (rocPRIM takes its merge-sort
//      path at this size: 8 launches, ~90 us).  Larger n: stable counting-sort passes (cs_sort below: 8 bits per
//      pass, two launches each) by score, then by category (16 bits, or the caller's category count), then
//      gathers -- 25 us for the 7-bit class sort of 100,000 boxes against 49 us of rocPRIM's radix sort, which
//      this file no longer uses.  Callers whose input consists of pre-sorted RUNS (the per-level top-k
//      lists of the RPN / dense detectors: d2amd_nms_runs) get the order from a merge instead -- step 1c below;
//      for the RPN (runs = categories, n <= 12,288) the whole pipeline is 4 launches.
//   2. wavefront bitmask kernel: one 64-lane wave per 64x64 tile of the (sorted) IoU matrix,
//        lane = row box, one uint64 word per lane; only tiles on/above the diagonal whose
//        category ranges overlap are evaluated (pair count = sum_c n_c^2/2, not N^2/2).
//        Diagonal tiles also emit the TRANSPOSED word (which earlier boxes of the tile suppress
//        this one), which turns step 3's diagonal resolution into a lane-parallel fixed point.
//   3. greedy reduction: one workgroup per category segment; see nms_reduce_kernel.
//   4. ordered compaction back to rank order -> original indices, count (+ the kept rows of up to 4 caller arrays).
// Bit-exactness: IoU arithmetic is evaluated exactly as torchvision's CPU nms / the reference's
// nms_rotated_cpu.cpp (fp32, no FMA contraction, IEEE divide, threshold compare in double).
#pragma clang fp contract(off)
#include <cstring>

#include "common.h"
#include "rotated_iou.h"

namespace d2amd {

typedef unsigned long long u64;

constexpr int RK_GROUP = 8;  // keys per scalar-load group of the ranking kernel (2 x s_load_dwordx16)
constexpr int RK_JC = 512;   // keys per chunk: one (64-box block, chunk) task is ~2.5k VALU instructions per wave
constexpr int RANK_MAX_N = 12288;  // brute-force ranking up to here (index must fit 16 bits)
constexpr int CS_THREADS = 1024, CS_WAVES = CS_THREADS / 64, CS_MAX_WGS = 128;  // counting sort (large inputs)
constexpr int NMS_MAX_RUNS = 8;    // pre-sorted runs a caller may describe (RPN / dense detectors: one per feature level)

struct NmsWorkspace {
  float* keys_out;    // [n] sort keys, ping-pong partner         (large inputs)
  int* iota;          // [n] sort values, ping-pong partner       (large inputs)
  int* order;         // [n] rank -> original index
  uint32_t* cls_r;    // [n] category in rank order               (large inputs)
  uint32_t* cls_s;    // [n] category in segment order
  int* rankpos;       // [n] segment position -> rank
  float* boxes_s;     // [n64 * 8] boxes in segment order (aligned: float4; rotated: 5 of 8 floats)
  u64* mask;          // [n64 * wcap]
  u64* diagT;         // [n64] transposed diagonal word per row
  u64* w1T;           // [n64] transposed word 1: bit i = row i of the previous block suppresses this box
  u64* w2T;           // [n64] transposed word 2: same for the block before the previous one
  u64* keepbits;      // [nblocks]   } zeroed together
  int* counters;      // [4]: nseg, error flags }
  int* rk_cnt;        // [chunks][2][n64] partial rank counts (brute-force ranking path)
  void* rk_keys;      // [n padded to RK_GROUP] 16-B key records (brute-force ranking path)
  uint8_t* flag_r;    // [n] kept flag in rank order              (radix path)
  int* blk_cnt;       // [n / 1024 + 1] kept flags per compaction workgroup (radix path)
  int* seg_start;     // [65536]
  int* run_cnt;       // [NMS_MAX_RUNS] entries of a run with a score > -inf (pre-sorted runs path)
  int* cs_hist;  // counting sort: [workgroups][256] digit counts
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
  w.w1T = (u64*)take((size_t)n64 * 8);
  w.w2T = (u64*)take((size_t)n64 * 8);
  const size_t z0 = off;
  w.keepbits = (u64*)take((n64 / 64) * 8);
  w.counters = (int*)take(4 * 4);
  w.zero_bytes = off - z0;
  w.rk_keys = take(n <= RANK_MAX_N ? (size_t)(n + RK_GROUP) * 16 : 0);
  w.rk_cnt = (int*)take(n <= RANK_MAX_N ? (size_t)((n + RK_JC - 1) / RK_JC) * 2 * n64 * 4 : 0);
  w.flag_r = (uint8_t*)take(n);
  w.blk_cnt = (int*)take((size_t)(n / 1024 + 2) * 8);  // (8 B per chunk: the direct finalize publishes two counts)
  w.seg_start = (int*)take(65536 * 4);
  w.run_cnt = (int*)take(64 * 4);  // [0, 8): live entries per run; [16, 16 + 12): "a run is not in order" per order workgroup
  w.cs_hist = (int*)take(n <= RANK_MAX_N ? 0 : (size_t)CS_MAX_WGS * 256 * 4);
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

constexpr int RK_THREADS = 256;

// Box record in segment order (8 floats): axis-aligned { x1, y1, x2, y2, area, category bits, flags, 0 },
// rotated { cx, cy, w, h, angle, category bits, 0, 0 }.  flags bit 0 = a coordinate is NaN / inf.
constexpr int BOX_REC = 8;

template <int BW>
__device__ __forceinline__ void store_box_record(float* __restrict__ boxes_s, int p, const float* __restrict__ src,
                                                 uint32_t cls) {
  float* d = boxes_s + (long)p * BOX_REC;
  if constexpr (BW == 4) {
    const float4 b = *reinterpret_cast<const float4*>(src);
    const float area = (b.z - b.x) * (b.w - b.y);  // torchvision nms_kernel_impl: areas = (x2 - x1) * (y2 - y1)
    const bool fin = (fabsf(b.x) < INFINITY) && (fabsf(b.y) < INFINITY) && (fabsf(b.z) < INFINITY) &&
                     (fabsf(b.w) < INFINITY);
    reinterpret_cast<float4*>(d)[0] = b;
    reinterpret_cast<float4*>(d)[1] = make_float4(area, __uint_as_float(cls), __uint_as_float(fin ? 0u : 1u), 0.f);
  } else {
    float v[BOX_REC];
#pragma unroll
    for (int k = 0; k < BW; k++) v[k] = src[k];
    v[5] = __uint_as_float(cls); v[6] = 0.f; v[7] = 0.f;
    reinterpret_cast<float4*>(d)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(d)[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// Key record of one box (16 B): { K, S }, read by the ranking kernel through the scalar cache (wave-uniform
// addresses).  The array is padded to a multiple of RK_GROUP records with all-ones sentinels that never count.
// (Comparing S as a double + the category as u32 -- 5 full-rate VALU per pair -- measured 29 us against 18 us
// for the two v_cmp_lt_u64 per pair used here.)
struct RankGroup { uint4 k[RK_GROUP]; };

__device__ __forceinline__ uint4 rank_key(float score, int64_t c, int t) {
  const u64 S = score_key(score, t);
  const u64 K = ((u64)c << 48) | S;
  return make_uint4((uint32_t)K, (uint32_t)(K >> 32), (uint32_t)S, (uint32_t)(S >> 32));
}

// Zeroes the reduction's accumulators (replaces a memset launch) and builds the key records.
__device__ __forceinline__ void nms_prep_body(const float* __restrict__ scores, const int64_t* __restrict__ idxs, int n,
                                uint4* __restrict__ keys, uint32_t* __restrict__ zero, int zero_words) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int q = t; q < zero_words; q += gridDim.x * blockDim.x) zero[q] = 0u;
  const int npad = (n + RK_GROUP - 1) / RK_GROUP * RK_GROUP;
  if (t >= npad) return;
  if (t >= n) {
    keys[t] = make_uint4(~0u, ~0u, ~0u, ~0u);
    return;
  }
  int64_t c = idxs ? idxs[t] : 0;
  if (c < 0 || c > 65535) c = 0;  // (flagged by the ranking kernel: the counters are being zeroed here)
  keys[t] = rank_key(scores[t], c, t);
}

// Every box needs two counts over all n keys: r_s = #{S_j < S_i} (global score rank) and r_cm = #{K_j < K_i}
// (class-major position).  Work item = (block of 64 boxes, one box per lane) x (chunk of RK_JC keys): the keys
// come in through the scalar cache (wave-uniform addresses -> s_load, operands in SGPRs), so a pair costs only
// its compares and carry adds -- no per-lane memory traffic, no LDS (the first version walked a 96 KiB LDS key
// table with one read per lane per pair: 44 us at n = 8.8k).  Partial counts are added to global counters; the
// wave that completes a block's last chunk scatters that block.
template <int BW>
__device__ __forceinline__ void nms_rank_body(
    const float* __restrict__ boxes, const int64_t* __restrict__ idxs, const uint4* __restrict__ keys, int n,
    int* __restrict__ order, int* __restrict__ rankpos, uint32_t* __restrict__ cls_s, float* __restrict__ boxes_s,
    int* __restrict__ counters, int* __restrict__ rk_cnt, int nchunks) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n64 = (n + 63) & ~63;
  const int j0 = blockIdx.x * RK_JC;  // RK_JC is a multiple of RK_GROUP
  if (j0 >= n) return;                // (batched launch: the grid is sized for the largest image)
  const int jn = min(RK_JC, n - j0);
  const int ib = blockIdx.y * (RK_THREADS / 64) + wid;  // block of 64 boxes
  const int i = ib * 64 + lane;
  if (ib * 64 >= n) return;  // uniform per wave
  if (blockIdx.x == 0 && idxs && i < n) {
    const int64_t c = idxs[i];
    if (c < 0 || c > 65535) atomicOr(&counters[1], 2);
  }
  const uint4 mine = keys[min(i, n - 1)];
  const u64 m0 = ((u64)mine.y << 32) | mine.x, m1 = ((u64)mine.w << 32) | mine.z;
  int r_s = 0, r_cm = 0;
  const RankGroup* gp = reinterpret_cast<const RankGroup*>(keys + j0);
  const int ngroups = (jn + RK_GROUP - 1) / RK_GROUP;
  RankGroup cur = gp[0];
  for (int g = 0; g < ngroups; g++) {
    const RankGroup nxt = gp[min(g + 1, ngroups - 1)];  // in flight while this group is compared
#pragma unroll
    for (int q = 0; q < RK_GROUP; q++) {
      const uint4 k = cur.k[q];
      const u64 k0 = ((u64)k.y << 32) | k.x, k1 = ((u64)k.w << 32) | k.z;
      r_cm += (k0 < m0) ? 1 : 0;
      r_s += (k1 < m1) ? 1 : 0;
    }
    cur = nxt;
  }
  // partial counts of this chunk (plain stores; summed by nms_rank_scatter_kernel -- a last-arriver scheme with
  // device-scope fences per wave cost 40+ us here)
  int* part = rk_cnt + (long)blockIdx.x * 2 * n64;
  part[i] = r_s;
  part[n64 + i] = r_cm;
}

template <int BW>
__device__ __forceinline__ void nms_rank_scatter_body(const float* __restrict__ boxes, const uint4* __restrict__ keys, int n,
                                        const int* __restrict__ rk_cnt, int nchunks, int* __restrict__ order,
                                        int* __restrict__ rankpos, uint32_t* __restrict__ cls_s,
                                        float* __restrict__ boxes_s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int n64 = (n + 63) & ~63;
  int r_s = 0, r_cm = 0;
  for (int c = 0; c < nchunks; c++) {
    r_s += rk_cnt[(long)c * 2 * n64 + i];
    r_cm += rk_cnt[(long)c * 2 * n64 + n64 + i];
  }
  const uint4 mine = keys[i];
  const uint32_t cls = mine.y >> 16;
  order[r_s] = i;
  rankpos[r_cm] = r_s;
  cls_s[r_cm] = cls;
  store_box_record<BW>(boxes_s, r_cm, boxes + (long)i * BW, cls);
}

// segment starts of the class-major sequence (small path; the radix path finds them while gathering)
__device__ __forceinline__ void nms_segments_body(const uint32_t* __restrict__ cls_s, int n, int* seg_start, int* counters) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t c = cls_s[p], cprev = cls_s[max(p - 1, 0)];
  if (p == 0 || c != cprev) {
    int pos = atomicAdd(&counters[0], 1);
    seg_start[pos] = p;
  }
}

// ---- step 1c: ORDER from pre-sorted runs ----------------------------------------------------------
// The NMS callers of this path (find_top_rpn_proposals, DenseDetector inference) hand over candidates that the
// top-k selection has just sorted: the array is a sequence of RUNS (one per feature level), and inside a run the
// entries that are not parked at score -inf are in order (descending score, equal scores in array order).  Ranking
// such an input from scratch (n^2 key compares, 26 us for 2 x 8,819; or two radix sorts for 100,000) is wasted work:
// the global score rank of an entry is its position among the live entries of its own run plus, for every other
// run, the number of live entries that precede it there -- one binary search per other run.
//   nms_runs_scan:  per run, the live entries' keys compacted in order (cs), every entry's count of live
//                   predecessors in its run (vr), the live count (run_cnt); also zeroes the reduction's accumulators
//   nms_runs_rank:  rank -> order[]; parked entries follow all live ones in array order (what ranking them by
//                   (score, index) gives); with runs == categories the class-major position IS the array position,
//                   so the box records and segment starts are written here and nothing else is left of step 1.
// Scores are compared through the same order-preserving integer key as everywhere else (NaN first, -0 below +0).
// A run that is not in order is detected (flag bit 2 of the result's flags): the host then ranks from scratch.
struct NmsRuns { int n_runs, are_cls; int off[NMS_MAX_RUNS + 1]; int cat_bits; };  // cat_bits: host only (class sort)

__device__ __forceinline__ uint32_t run_key(float s) {  // ascending = descending score
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~u;
}
constexpr uint32_t RUN_KEY_PARKED = 0xff800000u;  // run_key(-inf)

constexpr int RUNS_SCAN_THREADS = 1024, RUNS_SCAN_WAVES = RUNS_SCAN_THREADS / 64;
// One workgroup per run, RUNS_SCAN_ROWS rows of 1,024 consecutive entries per pass (coalesced: entry = row * 1,024 +
// thread).  A row's live entries are ballot-counted per wave; ONE exclusive scan over the [row][wave] counts (in
// that order = array order) gives every wave of every row its base.  (A thread owning consecutive entries needs
// no such table but loads with a stride: 31-41 us for a run of 20,000.)
template <int RUNS_SCAN_ROWS>
__device__ __forceinline__ void nms_runs_scan_body(const float* __restrict__ scores, const NmsRuns& R,
                                                   uint32_t* __restrict__ cs, int* __restrict__ vr,
                                                   int* __restrict__ run_cnt, uint32_t* __restrict__ zero,
                                                   int zero_words, int* __restrict__ blk_zero, int blk_words) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int q = blockIdx.x * RUNS_SCAN_THREADS + tid; q < zero_words; q += gridDim.x * RUNS_SCAN_THREADS) zero[q] = 0u;
  if (blockIdx.x == 0 && tid < blk_words && blk_zero) blk_zero[tid] = 0;  // (direct finalize's chunk counts)
  const int r = blockIdx.x;
  if (r >= R.n_runs) return;
  const int lo = R.off[r], hi = R.off[r + 1];
  constexpr int CELLS = RUNS_SCAN_ROWS * RUNS_SCAN_WAVES;  // <= 256
  __shared__ int cell[CELLS + 1];
  int base = 0;
  for (int c = lo; c < hi; c += RUNS_SCAN_THREADS * RUNS_SCAN_ROWS) {  // uniform trip count
    uint32_t key[RUNS_SCAN_ROWS];
    unsigned long long bal[RUNS_SCAN_ROWS];
#pragma unroll
    for (int k = 0; k < RUNS_SCAN_ROWS; k++) {
      const int i = c + k * RUNS_SCAN_THREADS + tid;
      key[k] = i < hi ? run_key(scores[i]) : RUN_KEY_PARKED;
    }
    __syncthreads();  // cell[] of the previous pass has been read
#pragma unroll
    for (int k = 0; k < RUNS_SCAN_ROWS; k++) {
      bal[k] = __ballot(key[k] != RUN_KEY_PARKED);
      if (lane == 0) cell[k * RUNS_SCAN_WAVES + wid] = __builtin_popcountll(bal[k]);
    }
    __syncthreads();
    if (wid == 0) {  // exclusive scan of the CELLS counts: 64 lanes x CELLS / 64 consecutive cells
      constexpr int PER = (CELLS + 63) / 64;
      int v[PER], sum = 0;
#pragma unroll
      for (int q = 0; q < PER; q++) { v[q] = lane * PER + q < CELLS ? cell[lane * PER + q] : 0; sum += v[q]; }
      int incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(incl, d, 64);
        if (lane >= d) incl += y;
      }
      int run = incl - sum;
#pragma unroll
      for (int q = 0; q < PER; q++) {
        if (lane * PER + q < CELLS) cell[lane * PER + q] = run;
        run += v[q];
      }
      if (lane == 63) cell[CELLS] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RUNS_SCAN_ROWS; k++) {
      const int i = c + k * RUNS_SCAN_THREADS + tid;
      if (i < hi) {
        const int pos = base + cell[k * RUNS_SCAN_WAVES + wid] + __builtin_popcountll(bal[k] & ((1ull << lane) - 1ull));
        vr[i] = pos;
        if (key[k] != RUN_KEY_PARKED) cs[lo + pos] = key[k];
      }
    }
    base += cell[CELLS];
  }
  if (tid == 0) run_cnt[r] = base;
}

// Two-level search: every 2^shift-th live key of every run is staged in LDS (<= 12,288 samples: shift = 0, i.e. ALL
// keys, for the inputs of the batched path); a search first narrows to one 2^shift window there, then finishes in
// global memory inside that window (<= 2 cache lines).  A plain binary search over the runs is a chain of log2(run)
// dependent misses per run -- the keys were written by another workgroup a moment ago, every probe goes to memory:
// 12-20 us.  The searches of the different runs advance together (independent probes per step).
constexpr int RUNS_SAMPLES = 12288;  // 48 KB: every key of the batched path (n <= 12,288 - 8) is a sample
template <int BW>
__device__ __forceinline__ void nms_runs_rank_body(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                   int n, const NmsRuns& R, const uint32_t* __restrict__ cs,
                                                   const int* __restrict__ vr, const int* __restrict__ run_cnt,
                                                   int* __restrict__ order, int* __restrict__ rankpos,
                                                   uint32_t* __restrict__ cls_s, float* __restrict__ boxes_s,
                                                   int* __restrict__ seg_start, int* __restrict__ counters,
                                                   int records) {
  __shared__ uint32_t samp[RUNS_SAMPLES];
  __shared__ int s_sbase[NMS_MAX_RUNS + 1], s_cnt[NMS_MAX_RUNS], s_off[NMS_MAX_RUNS + 1];
  if (blockIdx.x * blockDim.x >= n) return;  // uniform (batched launch: the grid is sized for the largest image)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // this entry's own score and in-run position: requested first, they arrive while the samples are staged
  const float my_score = scores[min(i, n - 1)];
  const int mine = vr[min(i, n - 1)];
  int shift = 0;
  while ((n >> shift) + NMS_MAX_RUNS > RUNS_SAMPLES) shift++;  // uniform
  // (the run table goes through LDS: indexing the kernel-argument struct with a per-lane run number makes the
  // compiler copy the struct to scratch memory)
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q <= NMS_MAX_RUNS; q++) s_off[q] = R.off[q];
    int acc = 0;
    for (int q = 0; q < NMS_MAX_RUNS; q++) {
      const int c = q < R.n_runs ? run_cnt[q] : 0;
      s_cnt[q] = c;
      s_sbase[q] = acc;
      acc += (c + (1 << shift) - 1) >> shift;
    }
    s_sbase[NMS_MAX_RUNS] = acc;
  }
  __syncthreads();
  {  // sample e = sample j of run q; 8 independent loads in flight per thread (a plain loop issues one at a time)
    constexpr int SU = 12;  // (12,288 samples / 1,024 threads: one batch)
    const int total = s_sbase[NMS_MAX_RUNS];
    for (int e0 = threadIdx.x; e0 < total; e0 += blockDim.x * SU) {
      uint32_t val[SU];
#pragma unroll
      for (int u = 0; u < SU; u++) {
        const int e = min(e0 + u * (int)blockDim.x, total - 1);
        int q = 0;
#pragma unroll
        for (int t = 1; t < NMS_MAX_RUNS; t++)
          if (e >= s_sbase[t]) q = t;
        val[u] = cs[s_off[q] + ((e - s_sbase[q]) << shift)];
      }
#pragma unroll
      for (int u = 0; u < SU; u++)
        if (e0 + u * (int)blockDim.x < total) samp[e0 + u * blockDim.x] = val[u];
    }
  }
  __syncthreads();
  if (i >= n) return;
  int r = 0;
#pragma unroll
  for (int q = 1; q < NMS_MAX_RUNS; q++)
    if (i >= s_off[q] && q < R.n_runs) r = q;
  const uint32_t key = run_key(my_score);
  int rank;
  if (key != RUN_KEY_PARKED) {
    // in order inside the run?  (equal keys are fine: array order is the tie order; with shift == 0 every key is
    // a sample: the predecessor comes from LDS)
    if (mine > 0) {
      const uint32_t prev = shift == 0 ? samp[s_sbase[r] + mine - 1] : cs[s_off[r] + mine - 1];
      if (prev > key) atomicOr(&counters[1], 4);
    }
    rank = mine;
    // per other run: the number of its live keys that precede mine = first position whose key is > bound, where
    // earlier runs win ties (lower array index: keys <= key count) and later runs lose them (keys < key count).
    // (All loops over q are fully unrolled with the run number a constant: lo / hi / probe stay in registers.)
    int lo[NMS_MAX_RUNS], hi[NMS_MAX_RUNS];
    int most = 0;  // uniform: samples of the longest run
#pragma unroll
    for (int q = 0; q < NMS_MAX_RUNS; q++) {
      const bool search = q < R.n_runs && q != r && !(q > r && key == 0u);  // (no key is strictly better than key 0)
      lo[q] = 0;
      hi[q] = search ? s_sbase[q + 1] - s_sbase[q] : 0;
      most = max(most, s_sbase[q + 1] - s_sbase[q]);
    }
    for (int span = most; span > 0; span >>= 1) {  // first sample > bound, all runs together
      uint32_t probe[NMS_MAX_RUNS];
#pragma unroll
      for (int q = 0; q < NMS_MAX_RUNS; q++)
        if (q < R.n_runs) probe[q] = samp[min(s_sbase[q] + ((lo[q] + hi[q]) >> 1), RUNS_SAMPLES - 1)];
#pragma unroll
      for (int q = 0; q < NMS_MAX_RUNS; q++) {
        if (q < R.n_runs) {
          const int mid = (lo[q] + hi[q]) >> 1;
          const uint32_t bound = q < r ? key : key - 1u;
          const bool go = lo[q] < hi[q];
          if (go && probe[q] <= bound) lo[q] = mid + 1;
          else if (go) hi[q] = mid;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NMS_MAX_RUNS; q++) {  // sample index a = lo[q] -> the window ((a - 1) << shift, a << shift]
      const int a = lo[q];
      lo[q] = a > 0 ? ((a - 1) << shift) + 1 : 0;
      hi[q] = a > 0 ? min(a << shift, s_cnt[q]) : 0;
    }
    for (int step = 0; step < shift; step++) {  // the windows hold < 2^shift candidates
      uint32_t probe[NMS_MAX_RUNS];
#pragma unroll
      for (int q = 0; q < NMS_MAX_RUNS; q++)  // unconditional (clamped) loads: all in flight together
        if (q < R.n_runs) probe[q] = cs[min(R.off[q] + ((lo[q] + hi[q]) >> 1), n - 1)];
#pragma unroll
      for (int q = 0; q < NMS_MAX_RUNS; q++) {
        if (q < R.n_runs) {
          const int mid = (lo[q] + hi[q]) >> 1;
          const uint32_t bound = q < r ? key : key - 1u;
          const bool go = lo[q] < hi[q];
          if (go && probe[q] <= bound) lo[q] = mid + 1;
          else if (go) hi[q] = mid;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NMS_MAX_RUNS; q++) rank += lo[q];
  } else {
    int live_total = 0, live_before = 0;
#pragma unroll
    for (int q = 0; q < NMS_MAX_RUNS; q++) {
      const int c = s_cnt[q];
      live_total += c;
      if (q < r) live_before += c;
    }
    rank = live_total + (i - (live_before + mine));  // parked entries: after every live one, in array order
  }
  order[rank] = i;
  if (!R.are_cls) {
    // no categories (records == 1): segment order == rank order, one segment; categories by the caller's class
    // ids (records == 0): the class sort that follows places the records
    if (records) {
      store_box_record<BW>(boxes_s, rank, boxes + (long)i * BW, 0u);
      if (i == 0) { seg_start[0] = 0; counters[0] = 1; }
    }
  } else {  // class-major order == array order
    rankpos[i] = rank;
    cls_s[i] = (uint32_t)r;
    store_box_record<BW>(boxes_s, i, boxes + (long)i * BW, (uint32_t)r);
    if (i == s_off[r]) {
      const int pos = atomicAdd(&counters[0], 1);
      seg_start[pos] = i;
    }
  }
}

// ---- step 1c for the batched path (n <= RANK_MAX_N): scan AND rank in one launch -------------------------------------
// Every workgroup of 1,024 threads loads ALL n scores (coalesced; it staged all keys anyway), compacts the live keys
// in LDS by one flat scan -- runs are contiguous, so the compacted list of run r is the slice [g(off[r]), g(off[r+1]))
// of the flat compaction, g(i) = live entries before i -- and ranks the 1,024 entries of its own row.  No scan
// launch, no dependent load of the runs' counts, nothing written to and read back from memory in between
// (scan 5 us + rank 17 us -> one kernel).  It also zeroes the reduction's accumulators (plain stores: nothing else in
// this launch touches them) and reports "a run is not in order" through a per-workgroup word.
// Workgroups of 256 threads (35 of them for 8,819 entries): with 1,024-thread workgroups the kernel ran on 9 CUs and
// was bound by their VALUs (the 11-step search of 4 other runs is ~1,100 instructions per entry: 22 us); a workgroup
// walks the n keys twice (count, then place: the second pass re-reads them from cache) instead of holding them.
// Workgroup size of the order kernel.  Every workgroup stages ALL keys and walks all rows of ORDER_THREADS keys twice
// (ballot + scan, compaction) before it ranks its own ORDER_THREADS entries: with 256 threads that was 2 x 28 dependent
// LDS passes for the RPN's ~7,000 boxes per image, with 1,024 it is 2 x 7 -- the connected step went 0.418 -> 0.401 /
// 0.392 ms on one box (512 threads: 0.403 / 0.405).
#ifndef D2AMD_ORDER_THREADS
#define D2AMD_ORDER_THREADS 1024
#endif
constexpr int ORDER_THREADS = D2AMD_ORDER_THREADS, ORDER_ROWS = RANK_MAX_N / ORDER_THREADS, ORDER_WAVES = ORDER_THREADS / 64;
constexpr int ORDER_LG = ORDER_THREADS == 256 ? 8 : ORDER_THREADS == 512 ? 9 : 10;
static_assert((1 << ORDER_LG) == ORDER_THREADS, "ORDER_THREADS: 256, 512 or 1024");
template <int BW>
__device__ __forceinline__ void nms_runs_order_small_body(
    const float* __restrict__ boxes, const float* __restrict__ scores, int n, const NmsRuns& R,
    int* __restrict__ order, int* __restrict__ rankpos, uint32_t* __restrict__ cls_s, float* __restrict__ boxes_s,
    uint32_t* __restrict__ zero, int zero_words, int* __restrict__ blk_zero, int blk_words,
    int* __restrict__ blk_flag, int records) {
  __shared__ uint32_t samp[RANK_MAX_N], s_raw[RANK_MAX_N];  // live keys compacted / all keys in array order
  __shared__ int cell[ORDER_ROWS * ORDER_WAVES + 1];
  __shared__ int s_sbase[NMS_MAX_RUNS + 1], s_off[NMS_MAX_RUNS + 1];
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (blockIdx.x * ORDER_THREADS >= n) return;  // uniform (batched launch: the grid is sized for the largest image)
  for (int q = blockIdx.x * ORDER_THREADS + tid; q < zero_words; q += gridDim.x * ORDER_THREADS) zero[q] = 0u;
  if (blockIdx.x == 0 && tid < blk_words) blk_zero[tid] = 0;
  // my own entry's box: requested first, it arrives while the keys are compacted
  const int i = blockIdx.x * ORDER_THREADS + tid;
  const long bi = min(i, n - 1);
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float b5 = 0.f;
  if (BW == 4) {
    b4 = *reinterpret_cast<const float4*>(boxes + bi * 4);
  } else {
    const float* bp = boxes + bi * BW;
    b4 = make_float4(bp[0], bp[1], bp[2], bp[3]);
    b5 = bp[4];
  }
  if (tid == 0) {
    s_bad = 0;
#pragma unroll
    for (int q = 0; q <= NMS_MAX_RUNS; q++) s_off[q] = R.off[q];
  }
  const int rows = (n + ORDER_THREADS - 1) / ORDER_THREADS;  // uniform, <= ORDER_ROWS
  // The keys go through LDS in array order first: ALL rows are requested at once (every value is used exactly once,
  // by its LDS store -- kept in registers for two passes the compiler re-loaded them row by row: 10+ dependent round
  // trips, 11 us of a 25 us kernel), both passes then read LDS.
  {
    float sv[ORDER_ROWS];
#pragma unroll
    for (int k = 0; k < ORDER_ROWS; k++) {
      const int j = k * ORDER_THREADS + tid;
      sv[k] = (k < rows && j < n) ? scores[j] : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < ORDER_ROWS; k++)
      if (k < rows) s_raw[k * ORDER_THREADS + tid] = run_key(sv[k]);  // (-inf: parked)
  }
  __syncthreads();
  for (int k = 0; k < rows; k++) {  // pass 1: live entries per (row, wave)
    const unsigned long long bal = __ballot(s_raw[k * ORDER_THREADS + tid] != RUN_KEY_PARKED);
    if (lane == 0) cell[k * ORDER_WAVES + wid] = __builtin_popcountll(bal);
  }
  __syncthreads();
  const int cells = rows * ORDER_WAVES;
  if (wid == 0) {  // exclusive scan of the counts (row-major = array order)
    constexpr int PER = (ORDER_ROWS * ORDER_WAVES + 63) / 64;
    int v[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) { v[q] = lane * PER + q < cells ? cell[lane * PER + q] : 0; sum += v[q]; }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(incl, d, 64);
      if (lane >= d) incl += y;
    }
    int run = incl - sum;
#pragma unroll
    for (int q = 0; q < PER; q++) {
      if (lane * PER + q < cells) cell[lane * PER + q] = run;
      run += v[q];
    }
    if (lane == 63) cell[ORDER_ROWS * ORDER_WAVES] = incl;
  }
  __syncthreads();
  const int live_total = cell[ORDER_ROWS * ORDER_WAVES];
  int g_mine = 0;
  uint32_t keyv = RUN_KEY_PARKED;
  for (int k = 0; k < rows; k++) {  // pass 2: the flat compaction in LDS
    const uint32_t kk = s_raw[k * ORDER_THREADS + tid];
    const bool live = kk != RUN_KEY_PARKED;
    const unsigned long long bal = __ballot(live);
    const int g = cell[k * ORDER_WAVES + wid] + __builtin_popcountll(bal & ((1ull << lane) - 1ull));  // live before j
    if (live) samp[g] = kk;
    if (k == (int)blockIdx.x) { g_mine = g; keyv = kk; }
  }
  // start of run q in the flat compaction = live entries before off[q]: wave q % 4 recounts the 64 entries around
  // off[q] (comparing every entry of every row with every run start was 6 us)
  {
#pragma unroll
    for (int q = 0; q <= NMS_MAX_RUNS; q++) {
      const int oq = R.off[q];  // (constant index: an SGPR)
      if ((q & (ORDER_WAVES - 1)) != wid) continue;  // uniform per wave
      if (oq >= n) {
        if (lane == 0) s_sbase[q] = live_total;  // (a run that starts at the end: empty)
      } else {
        const int j = (oq & ~63) + lane;  // (rows are whole: s_raw holds "parked" behind n)
        const bool live = s_raw[j] != RUN_KEY_PARKED;
        const unsigned long long bal = __ballot(live);
        if (lane == 0)
          s_sbase[q] = cell[(oq >> ORDER_LG) * ORDER_WAVES + ((oq >> 6) & (ORDER_WAVES - 1))] +
              __builtin_popcountll(bal & ((1ull << (oq & 63)) - 1ull));
      }
    }
  }
  __syncthreads();
  if (i < n) {
    int r = 0;
#pragma unroll
    for (int q = 1; q < NMS_MAX_RUNS; q++)
      if (i >= s_off[q] && q < R.n_runs) r = q;
    int rank;
    if (keyv != RUN_KEY_PARKED) {
      const int mine = g_mine - s_sbase[r];
      if (mine > 0 && samp[g_mine - 1] > keyv) s_bad = 1;  // in order inside the run?  (equal keys: array order)
      rank = mine;
      // per other run: the number of its live keys that precede mine (see nms_runs_rank_body), all runs together
      int lo[NMS_MAX_RUNS], hi[NMS_MAX_RUNS], sb[NMS_MAX_RUNS + 1];
      int most = 0;
#pragma unroll
      for (int q = 0; q <= NMS_MAX_RUNS; q++) sb[q] = s_sbase[q];  // (registers: not re-read from LDS per probe)
#pragma unroll
      for (int q = 0; q < NMS_MAX_RUNS; q++) {
        const bool search = q < R.n_runs && q != r && !(q > r && keyv == 0u);
        lo[q] = 0;
        hi[q] = search ? sb[q + 1] - sb[q] : 0;
        most = max(most, q < R.n_runs ? sb[q + 1] - sb[q] : 0);
      }
      for (int span = most; span > 0; span >>= 1) {
        uint32_t probe[NMS_MAX_RUNS];
#pragma unroll
        for (int q = 0; q < NMS_MAX_RUNS; q++)
          if (q < R.n_runs) probe[q] = samp[min(sb[q] + ((lo[q] + hi[q]) >> 1), RANK_MAX_N - 1)];
#pragma unroll
        for (int q = 0; q < NMS_MAX_RUNS; q++) {
          if (q < R.n_runs) {
            const int mid = (lo[q] + hi[q]) >> 1;
            const uint32_t bound = q < r ? keyv : keyv - 1u;
            const bool go = lo[q] < hi[q];
            if (go && probe[q] <= bound) lo[q] = mid + 1;
            else if (go) hi[q] = mid;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < NMS_MAX_RUNS; q++) rank += lo[q];
    } else {
      rank = live_total + (i - g_mine);  // parked entries: after every live one, in array order
    }
    order[rank] = i;
    const bool to_rank = !R.are_cls;  // no categories: segment order == rank order; runs = categories: == array order
    if (R.are_cls || records) {
      const int p = to_rank ? rank : i;
      const uint32_t cls = to_rank ? 0u : (uint32_t)r;
      float* d = boxes_s + (long)p * BOX_REC;
      if (BW == 4) {
        const float area = (b4.z - b4.x) * (b4.w - b4.y);
        const bool fin = (fabsf(b4.x) < INFINITY) && (fabsf(b4.y) < INFINITY) && (fabsf(b4.z) < INFINITY) &&
                         (fabsf(b4.w) < INFINITY);
        reinterpret_cast<float4*>(d)[0] = b4;
        reinterpret_cast<float4*>(d)[1] = make_float4(area, __uint_as_float(cls), __uint_as_float(fin ? 0u : 1u), 0.f);
      } else {
        reinterpret_cast<float4*>(d)[0] = b4;
        reinterpret_cast<float4*>(d)[1] = make_float4(b5, __uint_as_float(cls), 0.f, 0.f);
      }
      if (!to_rank) { rankpos[i] = rank; cls_s[i] = cls; }
    }
  }
  __syncthreads();
  if (tid == 0) blk_flag[blockIdx.x] = s_bad ? 4 : 0;
}

// order[] entries as indices: a run that was announced as sorted but is not (flag 4) leaves ranks that are no
// permutation -- slots of order[] may then hold anything.  The results are unspecified in that case (the host redoes
// the image), but nothing may be read out of bounds.
__device__ __forceinline__ int order_at(const int* __restrict__ order, int r, int n) {
  const int o = order[r];
  return (unsigned)o < (unsigned)n ? o : 0;
}

// ---- step 1b helpers (large inputs) ----------------------------------------------------------------
// boxes into segment order (+ segment starts).  BW = 4 (xyxy) or 5 (cxcywha); stored stride 4 / 8.
template <int BW>
__global__ void nms_gather_boxes_kernel(const float* __restrict__ boxes, const int* __restrict__ order,
                                        const int* __restrict__ rankpos, const uint32_t* __restrict__ cls_s,
                                        int n, float* __restrict__ boxes_s, int* seg_start, int* counters) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int r = rankpos ? rankpos[p] : p;
  int src = order_at(order, r, n);
  store_box_record<BW>(boxes_s, p, boxes + (long)src * BW, cls_s ? cls_s[p] : 0u);
  bool start = cls_s ? (p == 0 || cls_s[p] != cls_s[p - 1]) : (p == 0);
  if (start) {
    int pos = atomicAdd(&counters[0], 1);
    seg_start[pos] = p;
  }
}

// ---- stable counting sort on one 8-bit digit (large inputs: the class sort, the score sort) -----------------------
// Two launches per digit.  cs_hist_kernel: every workgroup owns a contiguous range of the input and writes its digit
// histogram (the first pass also MAKES the keys: class ids through order[], or the descending-orderable score bits).
// cs_place_kernel: a workgroup derives its digit bases from the histogram table (digits below + the same digit in the
// workgroups before it), then walks its range 1,024 elements at a time: lanes with equal digits find each other with
// 8 ballots (position among them = the rank inside the wave), the 16 waves' digit counts are prefix-summed per digit
// in LDS, and every element lands at  base[digit] + waves before + rank in wave  -- stable by construction (workgroup,
// tile, wave and lane order are all input order).  16-bit class ids take two passes, 32-bit scores four (LSD).
struct CsPass {
  const uint32_t* keys_in;  // null in cs_hist_kernel of the first pass: keys are made (-> keys_made)
  const int* vals_in;       // null: the element's own position
  uint32_t* keys_out;
  int* vals_out;
  int n, shift, per_wg;
  int* hist;  // [workgroups][256]
  const int64_t* idxs;  // first pass of the class sort: key = idxs[order[i]] ...
  const int* order;
  const float* scores;  // ... of the score sort: key = ~orderable(scores[i]) (ascending key = descending score)
  uint32_t* keys_made;
  int* counters;
};

__global__ __launch_bounds__(CS_THREADS) void cs_hist_kernel(CsPass a) {
  __shared__ int h[256];
  const int tid = threadIdx.x, g = blockIdx.x;
  if (tid < 256) h[tid] = 0;
  __syncthreads();
  const int lo = g * a.per_wg, hi = min(a.n, lo + a.per_wg);
  for (int i = lo + tid; i < hi; i += CS_THREADS) {
    uint32_t k;
    if (a.keys_in) {
      k = a.keys_in[i];
    } else {
      if (a.idxs) {
        int64_t c = a.idxs[order_at(a.order, i, a.n)];
        if (c < 0 || c > 65535) { atomicOr(&a.counters[1], 2); c = 0; }
        k = (uint32_t)c;
      } else {
        uint32_t u = __float_as_uint(a.scores[i]);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        k = ~u;
      }
      a.keys_made[i] = k;
    }
    atomicAdd(&h[(k >> a.shift) & 255u], 1);
  }
  __syncthreads();
  if (tid < 256) a.hist[g * 256 + tid] = h[tid];
}

__global__ __launch_bounds__(CS_THREADS) void cs_place_kernel(CsPass a) {
  __shared__ int base[256], sc[256], wc[CS_WAVES][256], off[CS_WAVES][256];
  const int tid = threadIdx.x, g = blockIdx.x, G = gridDim.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < CS_WAVES * 256; i += CS_THREADS) (&wc[0][0])[i] = 0;
  // digit totals and the counts of the workgroups before this one: 4 threads per digit, each a quarter of the rows
  // (up to 128 dependent-looking loads per thread otherwise: 15 us of latency)
  if (tid < 256) { sc[tid] = 0; base[tid] = 0; }
  __syncthreads();
  {
    const int c = tid & 255;
    int tot = 0, bef = 0;
#pragma unroll 4
    for (int j = tid >> 8; j < G; j += CS_THREADS / 256) {
      const int v = a.hist[j * 256 + c];
      tot += v;
      bef += j < g ? v : 0;
    }
    atomicAdd(&sc[c], tot);
    atomicAdd(&base[c], bef);
  }
  __syncthreads();
  const int before = tid < 256 ? base[tid] : 0;
  if (w == 0) {  // exclusive scan of the 256 digit totals: 4 per lane + a wave scan
    const int v0 = sc[4 * lane], v1 = sc[4 * lane + 1], v2 = sc[4 * lane + 2], v3 = sc[4 * lane + 3];
    const int sum = v0 + v1 + v2 + v3;
    int incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    const int ex = incl - sum;
    sc[4 * lane] = ex; sc[4 * lane + 1] = ex + v0; sc[4 * lane + 2] = ex + v0 + v1; sc[4 * lane + 3] = ex + v0 + v1 + v2;
  }
  __syncthreads();
  if (tid < 256) base[tid] = sc[tid] + before;
  __syncthreads();
  const int lo = g * a.per_wg, hi = min(a.n, lo + a.per_wg);
  for (int t0 = lo; t0 < hi; t0 += CS_THREADS) {  // uniform
    const int i = t0 + tid;
    const bool act = i < hi;
    const uint32_t k = act ? a.keys_in[i] : 0u;
    const uint32_t d = (k >> a.shift) & 255u;
    u64 m = __ballot(act);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1u;
      const u64 bal = __ballot(act && bit);
      m &= bit ? bal : ~bal;
    }
    const int rin = __popcll(m & ((1ull << lane) - 1ull));
    if (act && rin == 0) wc[w][d] = __popcll(m);
    __syncthreads();
    if (tid < 256) {
      int run = base[tid];
#pragma unroll
      for (int x = 0; x < CS_WAVES; x++) {
        const int v = wc[x][tid];
        wc[x][tid] = 0;
        off[x][tid] = run;
        run += v;
      }
      base[tid] = run;
    }
    __syncthreads();
    if (act) {
      const int pos = off[w][d] + rin;
      a.keys_out[pos] = k;
      a.vals_out[pos] = a.vals_in ? a.vals_in[i] : i;
    }
  }
}

// LSD passes over `bits` key bits.  made: where the first pass stores the keys it makes; (tmp_k, tmp_v) the ping-pong
// partner of (out_k, out_v), which hold the result.  All five buffers are distinct, n elements each.
static int cs_sort(CsPass a, int bits, uint32_t* made, uint32_t* tmp_k, int* tmp_v, uint32_t* out_k, int* out_v,
                   hipStream_t s) {
  const int passes = (bits + 7) / 8;
  const int wgs0 = std::min(CS_MAX_WGS, cdiv(a.n, CS_THREADS));
  a.per_wg = cdiv(cdiv(a.n, wgs0), CS_THREADS) * CS_THREADS;
  const int wgs = cdiv(a.n, a.per_wg);
  a.keys_made = made;
  const uint32_t* in_k = nullptr;
  const int* in_v = nullptr;
  for (int p = 0; p < passes; p++) {
    const bool to_out = ((passes - 1 - p) & 1) == 0;
    a.shift = 8 * p;
    a.keys_in = in_k;
    a.vals_in = in_v;
    a.keys_out = to_out ? out_k : tmp_k;
    a.vals_out = to_out ? out_v : tmp_v;
    hipLaunchKernelGGL(cs_hist_kernel, dim3(wgs), dim3(CS_THREADS), 0, s, a);
    D2_LAUNCH_OK();
    if (p == 0) a.keys_in = made;
    hipLaunchKernelGGL(cs_place_kernel, dim3(wgs), dim3(CS_THREADS), 0, s, a);
    D2_LAUNCH_OK();
    in_k = a.keys_out;
    in_v = a.vals_out;
  }
  return D2AMD_OK;
}

// ---- step 2: wavefront bitmask ------------------------------------------------------------
// 64 x 64 bit transpose inside a wave: lane j receives bit j of every lane's word (bit i = lane i).
// The transposed diagonal / first / second off-diagonal words (row suppresses column -> column is
// suppressed by row) are exactly these transposes: the IoU test of a (row, column) pair is evaluated
// once, in the reference's argument order, and read from either side.
__device__ __forceinline__ u64 wave_transpose64(u64 word, int lane) {
  const uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32);
  u64 out = 0;
#pragma unroll
  for (int j = 0; j < 64; j++) {
    const u64 b = __ballot(((j < 32 ? lo : hi) >> (j & 31)) & 1u);
    out = (lane == j) ? b : out;
  }
  return out;
}

__device__ __forceinline__ void store_mask_words(u64 word, int row, int col0, int lane, int w, int n, int wcap,
                                                 u64* __restrict__ mask, u64* __restrict__ diagT,
                                                 u64* __restrict__ w1T, u64* __restrict__ w2T) {
  if (row < n) mask[(long)row * wcap + w] = word;
  if (w <= 2) {  // uniform
    const u64 t = wave_transpose64(word, lane);
    if (col0 + lane < n) (w == 0 ? diagT : w == 1 ? w1T : w2T)[col0 + lane] = t;
  }
}

// v_max_f32 / v_min_f32 without the compiler's canonicalisation; the second operand is wave-uniform (SGPR)
__device__ __forceinline__ float vmaxf_s(float a, float b) {
  float r;
  asm("v_max_f32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(b));
  return r;
}
__device__ __forceinline__ float vminf_s(float a, float b) {
  float r;
  asm("v_min_f32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(b));
  return r;
}
__device__ __forceinline__ float vmax0f(float a) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
  return r;
}

struct ColGroup { float v[4][BOX_REC]; };  // 4 column records = 2 x s_load_dwordx16 through the scalar cache

// Axis-aligned tile (rb, w): lane = row rb * 64 + lane against the 64 columns of block rb + w.
// torchvision's test is   ovr = inter / (iarea + jarea - inter);  ovr > thr   with float ovr, double thr.
// FAST path (finite boxes, 0 < denom < inf): with f = largest float <= thr, g = next float up and
// mid = (f + g) / 2, the correctly rounded quotient satisfies  ovr > thr  <=>  ovr >= g  <=>  x > mid, or
// x == mid when the tie rounds up to g (g even), for the real x = inter / denom; and  x > mid  <=>  inter > mid * denom
// evaluated EXACTLY in double (25-bit mid x 24-bit denom = 49 bits).  This replaces the ~12-instruction IEEE
// division by cvt/cvt/mul/cmp.  max/min: with finite operands (a < b) ? b : a equals v_max_f32 except for the
// sign of a zero, which cannot change a `> thr` outcome.  Anything else (a non-finite coordinate in the tile,
// denom outside (0, inf), an out-of-range threshold) takes the literal formula.
// Returns false when the tile lies beyond the last column block that shares a category with the row block: such
// words are never read by the reduction (it walks the blocks of the rows' own segment only), and neither is any
// tile further right -- the caller's loop over w stops there.
template <bool FAST, bool TIE_UP>
__device__ __forceinline__ bool nms_mask_tile(const float* __restrict__ boxes_s, int use_cls, int n, int wcap, int rb,
                                              int w, double thr, double mid, u64* __restrict__ mask,
                                              u64* __restrict__ diagT, u64* __restrict__ w1T,
                                              u64* __restrict__ w2T) {
  const int lane = threadIdx.x;
  const int cb = rb + w;
  const int row = rb * 64 + lane;
  const int col0 = cb * 64;
  const int nblocks = (n + 63) >> 6;
  if (cb >= nblocks || w >= wcap) return false;  // (or: batched grid larger than this image)
  u64 word = 0;
  if (use_cls) {
    // categories ascend along the sorted sequence: tile is empty unless ranges touch
    const uint32_t row_last = __float_as_uint(boxes_s[(long)min(rb * 64 + 63, n - 1) * BOX_REC + 5]);
    const uint32_t col_first = __float_as_uint(boxes_s[(long)col0 * BOX_REC + 5]);
    if (col_first > row_last) return false;
  }
  {
    const int rrow = min(row, n - 1), rcol = min(col0 + lane, n - 1);
    const float4 rbx = reinterpret_cast<const float4*>(boxes_s)[rrow * 2];
    const float4 rex = reinterpret_cast<const float4*>(boxes_s)[rrow * 2 + 1];
    const float iarea = rex.x;
    const uint32_t my_cls = __float_as_uint(rex.y);
    const uint32_t cflags = __float_as_uint(boxes_s[(long)rcol * BOX_REC + 6]);
    const bool weird = __ballot((__float_as_uint(rex.z) | cflags) & 1u) != 0ull;  // uniform
    const ColGroup* cg = reinterpret_cast<const ColGroup*>(boxes_s + (long)col0 * BOX_REC);
    // (records of the last block beyond n are allocated but unwritten: whatever they hold is masked below)
    bool exact = !FAST || weird;  // uniform
    if (!exact) {
      bool bad = false;
      // r06: the comparison  inter >= mid * denom  WITHOUT double arithmetic (four quarter-rate instructions per pair were
      // half of this kernel's VALU time).  With f = the float below mid and h = mid - f (half an ulp of f: a power of two),
      //     inter - mid * denom  =  A - e,      A = inter - f * denom,   e = h * denom  (exact in fp32),
      // and d = fma(-f, denom, inter) is A rounded ONCE: sign and magnitude to 2^-24.  So d > 1.001 e  =>  A > e and
      // d < 0.999 e  =>  A < e, decided in three fp32 instructions; a pair inside that band -- its IoU within an ulp of the
      // threshold: ~1e-7 of the pairs -- or with a vanishing e sends the tile to the literal formula below, like any
      // other irregular value.  (Bit-exactness is the bar: tests/test_gpu_parity.py, tests/test_gpu_nms_runs.py.)
      const float f_lo = __double2float_rd(mid), h_ulp = (float)(mid - (double)f_lo);
      // not unrolled: every wave runs the body once per group, a 64x unrolled body is instruction-fetch bound
#pragma unroll 1
      for (int g = 0; g < 16; g++) {
        const ColGroup c = cg[g];
        uint32_t nib = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float jx1 = c.v[q][0], jy1 = c.v[q][1], jx2 = c.v[q][2], jy2 = c.v[q][3], jarea = c.v[q][4];
          const uint32_t jcls = __float_as_uint(c.v[q][5]);
          const float xx1 = vmaxf_s(rbx.x, jx1), yy1 = vmaxf_s(rbx.y, jy1);
          const float xx2 = vminf_s(rbx.z, jx2), yy2 = vminf_s(rbx.w, jy2);
          const float ww = vmax0f(xx2 - xx1), hh = vmax0f(yy2 - yy1);
          const float inter = ww * hh;
          const float denom = iarea + jarea - inter;
          const float d = __builtin_fmaf(-f_lo, denom, inter), e = h_ulp * denom;
          const bool hit = d > e;
          // not (+normal | +denormal), or too close to call in fp32: redo the tile literally
          bad |= !__builtin_amdgcn_classf(denom, 0x180) | !(d > 1.001f * e || d < 0.999f * e) | !(e > 1e-30f);
          nib |= (hit && jcls == my_cls) ? (1u << q) : 0u;
        }
        word |= (u64)nib << (4 * g);
      }
      exact = __ballot(bad) != 0ull;  // (columns beyond n may raise it too: harmless)
    }
    if (exact) {
      word = 0;
#pragma unroll 1
      for (int g = 0; g < 16; g++) {
        const ColGroup c = cg[g];
        uint32_t nib = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float jx1 = c.v[q][0], jy1 = c.v[q][1], jx2 = c.v[q][2], jy2 = c.v[q][3], jarea = c.v[q][4];
          const uint32_t jcls = __float_as_uint(c.v[q][5]);
          // torchvision nms_kernel_impl: std::max(a,b) = (a<b)?b:a, std::min(a,b) = (b<a)?b:a
          float xx1 = (rbx.x < jx1) ? jx1 : rbx.x;
          float yy1 = (rbx.y < jy1) ? jy1 : rbx.y;
          float xx2 = (jx2 < rbx.z) ? jx2 : rbx.z;
          float yy2 = (jy2 < rbx.w) ? jy2 : rbx.w;
          float ww = xx2 - xx1, hh = yy2 - yy1;
          ww = (0.f < ww) ? ww : 0.f;
          hh = (0.f < hh) ? hh : 0.f;
          const float inter = ww * hh;
          const float ovr = inter / (iarea + jarea - inter);
          nib |= (((double)ovr > thr) && jcls == my_cls) ? (1u << q) : 0u;
        }
        word |= (u64)nib << (4 * g);
      }
    }
    // validity of the columns / rows and the strict upper triangle of the diagonal tile, applied once
    const int jmax = n - col0;  // > 0
    if (jmax < 64) word &= (1ull << jmax) - 1ull;
    if (w == 0) word &= (lane < 63) ? ~((2ull << lane) - 1ull) : 0ull;
    if (row >= n) word = 0;
  }
  store_mask_words(word, row, col0, lane, w, n, wcap, mask, diagT, w1T, w2T);
  return true;
}

// grid = (row blocks, min(wcap, MASK_GRID_Y)): a workgroup walks the tiles w = blockIdx.y, + gridDim.y, ... of its
// row block up to the last live one.  The host only knows an UPPER BOUND of the largest category (wcap sizes the
// row pitch of the mask); how far a row block really reaches is read from the records here, so a generous bound
// costs address space, not launches.
constexpr int MASK_GRID_Y = 64;
template <bool FAST, bool TIE_UP>
__device__ __forceinline__ void nms_mask_body(const float* __restrict__ boxes_s, int use_cls, int n, int wcap,
                                              double thr, double mid, u64* __restrict__ mask,
                                              u64* __restrict__ diagT, u64* __restrict__ w1T, u64* __restrict__ w2T) {
  for (int w = blockIdx.y; w < wcap; w += gridDim.y)
    if (!nms_mask_tile<FAST, TIE_UP>(boxes_s, use_cls, n, wcap, blockIdx.x, w, thr, mid, mask, diagT, w1T, w2T)) break;
}

// Rotated tile: the polygon clip is not symmetric in floating point and the reference evaluates
// iou(kept box, later box) (nms_rotated_cpu.cpp:45-54): rows are the earlier boxes, so that is the order
// used here, once per pair (the transposed words come from the bit transpose of the same results).
__device__ __forceinline__ bool nms_mask_rot_tile(const float* __restrict__ boxes_s, int n, int wcap, int rb, int w,
                                                  double thr, u64* __restrict__ mask, u64* __restrict__ diagT,
                                                  u64* __restrict__ w1T, u64* __restrict__ w2T,
                                                  RotIouScratch<64>& S) {
  const int lane = threadIdx.x;
  const int cb = rb + w;
  const int row = rb * 64 + lane;
  const int col0 = cb * 64;
  const int nblocks = (n + 63) >> 6;
  if (cb >= nblocks || w >= wcap) return false;
  u64 word = 0;
  const uint32_t row_last = __float_as_uint(boxes_s[(long)min(rb * 64 + 63, n - 1) * BOX_REC + 5]);
  const uint32_t col_first = __float_as_uint(boxes_s[(long)col0 * BOX_REC + 5]);
  if (col_first > row_last) return false;  // (see nms_mask_tile)
  {
    const int rrow = min(row, n - 1), rcol = min(col0 + lane, n - 1);
    float rbx[5], cbx[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { rbx[k] = boxes_s[(long)rrow * BOX_REC + k]; cbx[k] = boxes_s[(long)rcol * BOX_REC + k]; }
    const uint32_t my_cls = __float_as_uint(boxes_s[(long)rrow * BOX_REC + 5]);
    const uint32_t col_cls = __float_as_uint(boxes_s[(long)rcol * BOX_REC + 5]);
    for (int j = 0; j < 64; j++) {
      if (col0 + j >= n) break;  // uniform
      float jb[5];
#pragma unroll
      for (int k = 0; k < 5; k++) jb[k] = __shfl(cbx[k], j);
      const uint32_t jcls = (uint32_t)__shfl((int)col_cls, j);
      // every lane evaluates (uniform control flow); cheap rejection by category / triangle first
      const bool cand = (jcls == my_cls) && (row < n) && (col0 + j > row);
      float ovr = 0.f;
      if (cand) ovr = single_box_iou_rotated<64>(rbx, jb, S, lane);
      word |= (cand && ((double)ovr >= thr)) ? (1ull << j) : 0ull;  // nms_rotated_cpu.cpp:54
    }
  }
  store_mask_words(word, row, col0, lane, w, n, wcap, mask, diagT, w1T, w2T);
  return true;
}
// The same tile with the surviving pairs COMPACTED (thr > 0): a wave's 64 rows against one column ran the whole polygon
// clip as soon as ONE of its lanes had a nearby box -- for the RRPN's 2,000 boxes per level that is most columns, although
// ~98 % of the pairs are rejected by the centre-distance test (rotated_iou.h: rot_pair_is_zero, exactly the pairs whose
// IoU is +0.f).  Here every lane first tests its row against the 64 columns (the cheap test only), the pairs that pass
// are listed in LDS, and the clip then runs with lane = one listed pair: 64 clips per pass instead of 64 per column.
// Measured (8,819 boxes / 5 levels, same box): 0.61 -> see profiles/r04/LOG.md.
struct RotTileLds {
  float rowb[5][64], colb[5][64];  // the tile's boxes, component-major
  unsigned long long words[64];    // suppression word per row
  uint16_t list[64 * 64];          // row << 6 | column of the pairs that need the clip
};
__device__ __forceinline__ bool nms_mask_rot_tile_compact(const float* __restrict__ boxes_s, int n, int wcap, int rb, int w,
                                                          double thr, u64* __restrict__ mask, u64* __restrict__ diagT,
                                                          u64* __restrict__ w1T, u64* __restrict__ w2T,
                                                          RotIouScratch<64>& S, RotTileLds& T) {
  const int lane = threadIdx.x;
  const int cb = rb + w;
  const int row = rb * 64 + lane;
  const int col0 = cb * 64;
  const int nblocks = (n + 63) >> 6;
  if (cb >= nblocks || w >= wcap) return false;
  const uint32_t row_last = __float_as_uint(boxes_s[(long)min(rb * 64 + 63, n - 1) * BOX_REC + 5]);
  const uint32_t col_first = __float_as_uint(boxes_s[(long)col0 * BOX_REC + 5]);
  if (col_first > row_last) return false;  // (see nms_mask_tile)
  const int rrow = min(row, n - 1), rcol = min(col0 + lane, n - 1);
  float rbx[5], cbx[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    rbx[k] = boxes_s[(long)rrow * BOX_REC + k];
    cbx[k] = boxes_s[(long)rcol * BOX_REC + k];
    T.rowb[k][lane] = rbx[k];
    T.colb[k][lane] = cbx[k];
  }
  T.words[lane] = 0ull;
  const float thr_ratio = (float)(0.99 * thr);
  const uint32_t my_cls = __float_as_uint(boxes_s[(long)rrow * BOX_REC + 5]);
  const uint32_t col_cls = __float_as_uint(boxes_s[(long)rcol * BOX_REC + 5]);
  u64 cmask = 0;  // bit j: (row, column j) needs the clip
  for (int j = 0; j < 64; j++) {
    if (col0 + j >= n) break;  // uniform
    float jb[5];
#pragma unroll
    for (int k = 0; k < 5; k++) jb[k] = __shfl(cbx[k], j);
    const uint32_t jcls = (uint32_t)__shfl((int)col_cls, j);
    bool cand = (jcls == my_cls) && (row < n) && (col0 + j > row) && !rot_pair_is_zero(rbx, jb);
    // IoU = inter / (a1 + a2 - inter) <= min(a1, a2) / max(a1, a2): a pair whose area ratio is below the threshold by
    // more than 1 % cannot reach it -- for boxes whose sides are all >= 1 px.  The clip the reference runs accepts
    // vertices up to EPS / |side| outside a box (EPS = 1e-5 absolute, box_iou_rotated_utils.h:79-164): at side 0.01 that
    // is 10 % of the side and the intersection hull can exceed the smaller box by far more than 1 % (0.01 x
    // 0.01 against 0.011 x 0.019 -> reference IoU 0.59 at area ratio 0.48); at side >= 1 it is <= 1e-5 of it.  Smaller
    // boxes take the clip.
    {
      const float a1 = rbx[2] * rbx[3], a2 = jb[2] * jb[3];
      const bool sized = rbx[2] >= 1.f && rbx[3] >= 1.f && jb[2] >= 1.f && jb[3] >= 1.f;
      cand = cand && !(sized && fminf(a1, a2) < thr_ratio * fmaxf(a1, a2));
    }
    cmask |= cand ? (1ull << j) : 0ull;
  }
  // the listed pairs: exclusive prefix of the per-lane counts over the wave, then every lane appends its own
  const int mine = __builtin_popcountll(cmask);
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  const int total = __shfl(incl, 63);
  {
    int at = incl - mine;
    u64 m = cmask;
    while (m) {
      const int j = __builtin_ctzll(m);
      m &= m - 1;
      T.list[at++] = (uint16_t)((lane << 6) | j);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (one wave: LDS writes above are ordered before the reads below)
  for (int t0 = 0; t0 < total; t0 += 64) {  // uniform
    const int t = t0 + lane;
    const bool live = t < total;
    const int e = T.list[live ? t : total - 1];
    const int r = e >> 6, j = e & 63;
    float b1[5], b2[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { b1[k] = T.rowb[k][r]; b2[k] = T.colb[k][j]; }
    const float ovr = single_box_iou_rotated<64>(b1, b2, S, lane);  // (every lane: uniform control flow inside)
    if (live && (double)ovr >= thr) atomicOr(&T.words[r], 1ull << j);  // nms_rotated_cpu.cpp:54
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const u64 word = T.words[lane];
  store_mask_words(word, row, col0, lane, w, n, wcap, mask, diagT, w1T, w2T);
  return true;
}
// r06: the same tile by a workgroup of ROTW waves.  One wave per 64 x 64 tile left ~9 single-wave workgroups per CU,
// each a chain of 64 cheap tests + 2-3 clip passes (~50-100 us per tile, nothing to hide a latency behind: 42 G pairs/s
// against the IoU kernel's 72 G with its 4-16 rows per wave).  Here lane = COLUMN, wave v tests the rows 16 v .. 16 v + 15
// (row boxes are LDS broadcasts, no shuffles), the four waves append their survivors to ONE list and share its clip
// passes (64 pairs each, a scratch per wave); wave 0 stores the words.  Same tests, same clip, same operand order
// (row = the earlier box first: nms_rotated_cpu.cpp:45-54) -- the words are bit-identical to the one-wave tile's.
constexpr int ROTW = 4;
// pairs per clip pass of a wave (lanes beyond it idle) -- it sizes the wave's scratch -- and workgroups per CU the kernel
// is compiled for.  MEASURED (rrpn_micro's 2 x 8,819 boxes, same box, before the projection bound): 64 pairs / 2 per CU
// (61 KB of LDS) 126 us | 32 / 4 (37 KB, 106 VGPRs) 118.5 | 32 / 3 118 | 16 / 4 185 (twice the passes).
constexpr int ROT_PASS = 32;
constexpr int ROT_OCC = 4;
struct RotTileWg {
  float rowb[5][64], colb[5][64];
  float rowcs[2][64], colcs[2][64];  // cos, sin of the boxes' angles (the projection bound below)
  uint32_t rcls[64], ccls[64];
  unsigned long long words[64];
  uint16_t list[64 * 64];
  int total;
};
// Upper bound of the intersection area of two rotated boxes from their projections: the intersection lies inside the
// rectangle spanned -- in the frame of box 1 -- by the overlaps of the two boxes' projections onto box 1's axes, and
// likewise in the frame of box 2; it is also no larger than either box.  (rot_vertices: the width axis of a box is
// (cos, -sin), the height axis (sin, cos).)  SLACK px are added to every overlap length: the centre difference is taken
// in fp32 here (the reference shifts by the midpoint in double) and sin / cos are fp32 -- errors of ~1e-4 px at 1,000 px.
__device__ __forceinline__ float rot_inter_upper_bound(const float* __restrict__ b1, float c1, float s1,
                                                       const float* __restrict__ b2, float c2, float s2) {
  constexpr float SLACK = 1e-2f;
  const float dx = b2[0] - b1[0], dy = b2[1] - b1[1];
  const float cd = fabsf(c1 * c2 + s1 * s2), sd = fabsf(s1 * c2 - c1 * s2);
  const float hw1 = 0.5f * b1[2], hh1 = 0.5f * b1[3], hw2 = 0.5f * b2[2], hh2 = 0.5f * b2[3];
  auto frame = [&](float cu, float su, float hwa, float hha, float hwb, float hhb) {
    // box a's frame: box b's centre at (du, dv), its half extents along a's axes (eu, ev)
    const float du = dx * cu - dy * su, dv = dx * su + dy * cu;
    const float eu = hwb * cd + hhb * sd, ev = hwb * sd + hhb * cd;
    const float ou = fminf(hwa, du + eu) - fmaxf(-hwa, du - eu) + SLACK;
    const float ov = fminf(hha, dv + ev) - fmaxf(-hha, dv - ev) + SLACK;
    return fmaxf(ou, 0.f) * fmaxf(ov, 0.f);
  };
  // (in box 2's frame the centre difference changes sign: the overlaps are symmetric in it)
  return fminf(frame(c1, s1, hw1, hh1, hw2, hh2), frame(c2, s2, hw2, hh2, hw1, hh1));
}
__device__ __forceinline__ bool nms_mask_rot_tile_wg(const float* __restrict__ boxes_s, int n, int wcap, int rb, int w,
                                                     double thr, u64* __restrict__ mask, u64* __restrict__ diagT,
                                                     u64* __restrict__ w1T, u64* __restrict__ w2T,
                                                     RotIouScratch<ROT_PASS>* S, RotTileWg& T) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cb = rb + w;
  const int col0 = cb * 64;
  const int nblocks = (n + 63) >> 6;
  if (cb >= nblocks || w >= wcap) return false;
  const uint32_t row_last = __float_as_uint(boxes_s[(long)min(rb * 64 + 63, n - 1) * BOX_REC + 5]);
  const uint32_t col_first = __float_as_uint(boxes_s[(long)col0 * BOX_REC + 5]);
  if (col_first > row_last) return false;  // (see nms_mask_tile)
  __syncthreads();  // (the previous tile's LDS is no longer read)
  if (wv < 2) {
    const int rec = min((wv == 0 ? rb * 64 : col0) + lane, n - 1);
    float (&dst)[5][64] = wv == 0 ? T.rowb : T.colb;
#pragma unroll
    for (int k = 0; k < 5; k++) dst[k][lane] = boxes_s[(long)rec * BOX_REC + k];
    (wv == 0 ? T.rcls : T.ccls)[lane] = __float_as_uint(boxes_s[(long)rec * BOX_REC + 5]);
    {
      const float th = boxes_s[(long)rec * BOX_REC + 4] * 0.01745329251f;
      float (&cs)[2][64] = wv == 0 ? T.rowcs : T.colcs;
      cs[0][lane] = cosf(th);
      cs[1][lane] = sinf(th);
    }
    if (wv == 0) {
      T.words[lane] = 0ull;
      if (lane == 0) T.total = 0;
    }
  }
  __syncthreads();
  const float thr_ratio = (float)(0.99 * thr);
  float cbx[5];
#pragma unroll
  for (int k = 0; k < 5; k++) cbx[k] = T.colb[k][lane];
  const uint32_t col_cls = T.ccls[lane];
  const float ccos = T.colcs[0][lane], csin = T.colcs[1][lane];
  const bool cin = col0 + lane < n;
  unsigned live = 0u;  // bit i: (row 16 wv + i, this column) needs the clip
#pragma unroll 4
  for (int i = 0; i < 64 / ROTW; i++) {
    const int r = wv * (64 / ROTW) + i, row = rb * 64 + r;
    if (row >= n) break;  // uniform
    float rbx[5];
#pragma unroll
    for (int k = 0; k < 5; k++) rbx[k] = T.rowb[k][r];
    bool cand = cin && (col_cls == T.rcls[r]) && (col0 + lane > row) && !rot_pair_is_zero(rbx, cbx);
    {  // (the area-ratio bound of nms_mask_rot_tile_compact, same operands)
      const float a1 = rbx[2] * rbx[3], a2 = cbx[2] * cbx[3];
      const bool sized = rbx[2] >= 1.f && rbx[3] >= 1.f && cbx[2] >= 1.f && cbx[3] >= 1.f;
      cand = cand && !(sized && fminf(a1, a2) < thr_ratio * fmaxf(a1, a2));
      // r06: IoU = inter / (a1 + a2 - inter) grows with inter, and inter <= the projection bound: a pair whose BOUND gives
      // an IoU more than 1 % below the threshold cannot reach it (same size condition as above: at sides >= 1 px the
      // reference's vertex / edge tolerances are <= 1e-5 of a side).  The centre-distance test passes every pair within the
      // sum of the half DIAGONALS -- for the RRPN's elongated anchors most of those do not overlap by half.
      if (cand && sized) {
        const float ub = rot_inter_upper_bound(rbx, T.rowcs[0][r], T.rowcs[1][r], cbx, ccos, csin);
        cand = !(ub * (1.f + thr_ratio) < thr_ratio * (a1 + a2));
      }
    }
    live |= cand ? (1u << i) : 0u;
  }
  const int mine = __builtin_popcount(live);
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  int base = 0;
  if (lane == 63 && incl > 0) base = atomicAdd(&T.total, incl);
  base = __shfl(base, 63);
  {
    int at = base + incl - mine;
    unsigned m = live;
    while (m) {
      const int i = __builtin_ctz(m);
      m &= m - 1;
      T.list[at++] = (uint16_t)(((wv * (64 / ROTW) + i) << 6) | lane);
    }
  }
  __syncthreads();
  const int total = T.total;
  for (int t0 = wv * ROT_PASS; t0 < total; t0 += ROTW * ROT_PASS) {  // uniform per wave
    const int t = t0 + lane;
    const bool on = lane < ROT_PASS && t < total;
    const int e = T.list[on ? t : total - 1];
    const int r = e >> 6, j = e & 63;
    float b1[5], b2[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { b1[k] = T.rowb[k][r]; b2[k] = T.colb[k][j]; }
    if (ROT_PASS == 64 || lane < ROT_PASS) {
      const float ovr = single_box_iou_rotated<ROT_PASS>(b1, b2, S[wv], lane);
      if (on && (double)ovr >= thr) atomicOr(&T.words[r], 1ull << j);  // nms_rotated_cpu.cpp:54
    }
  }
  __syncthreads();
  if (wv == 0) store_mask_words(T.words[lane], rb * 64 + lane, col0, lane, w, n, wcap, mask, diagT, w1T, w2T);
  return true;
}
__device__ __forceinline__ void nms_mask_rot_wg_body(const float* __restrict__ boxes_s, int n, int wcap, double thr,
                                                     u64* __restrict__ mask, u64* __restrict__ diagT,
                                                     u64* __restrict__ w1T, u64* __restrict__ w2T) {
  __shared__ RotIouScratch<ROT_PASS> S[ROTW];
  __shared__ RotTileWg T;
  for (int w = blockIdx.y; w < wcap; w += gridDim.y)
    if (!nms_mask_rot_tile_wg(boxes_s, n, wcap, blockIdx.x, w, thr, mask, diagT, w1T, w2T, S, T)) break;
}
__device__ __forceinline__ void nms_mask_rot_body(const float* __restrict__ boxes_s, int n, int wcap, double thr,
                                                  u64* __restrict__ mask, u64* __restrict__ diagT,
                                                  u64* __restrict__ w1T, u64* __restrict__ w2T, int plain = 0) {
  __shared__ RotIouScratch<64> S;
  // a pair that the distance test rejects has IoU +0.f: it is "suppressed" only by a threshold <= 0 (uniform)
  if (thr > 0.0 && !plain) {
    __shared__ RotTileLds T;
    for (int w = blockIdx.y; w < wcap; w += gridDim.y)
      if (!nms_mask_rot_tile_compact(boxes_s, n, wcap, blockIdx.x, w, thr, mask, diagT, w1T, w2T, S, T)) break;
    return;
  }
  for (int w = blockIdx.y; w < wcap; w += gridDim.y)
    if (!nms_mask_rot_tile(boxes_s, n, wcap, blockIdx.x, w, thr, mask, diagT, w1T, w2T, S)) break;
}

// ---- step 3: greedy reduction ----------------------------------------------------------------
// One workgroup (wave 0 + RED_GROUPS pusher waves) per category segment, walking its 64-row blocks in order.
//   wave 0     resolves the diagonal block: kept_j = cand_j && !(DT_j & kept), iterated as a
//              lane-parallel fixed point (DT_j = transposed diagonal word; position t is final
//              after t iterations and the typical depth is 1-2), and publishes `kept`.  Suppression
//              by the previous two blocks comes from the TRANSPOSED words 1 and 2 (w1T, w2T: bit i =
//              row i of that block suppresses me) tested against the previous two `kept` values held
//              in registers: nothing on the block-to-block chain goes through an LDS atomic.  DT, w1T,
//              w2T are staged in LDS a window of RED_WIN blocks at a time.
//   pushers    wave 1 + g owns the half-block units u = g (mod RED_GROUPS): 32 rows in 64 VGPRs,
//              lane = word RED_NEAR + lane, loaded one round ahead; once `kept` of the unit's block is
//              published it ORs the kept rows into removed[] (LDS atomics).  Block rel only needs the
//              pushes of blocks <= rel - RED_NEAR, so a push has two blocks of slack.
// Synchronisation is by LDS flags (kept_ready / pushed[g]), not barriers: with one barrier per block
// every block waited for a pusher to issue its row loads (1.5 us per block; 0.3-0.4 us now).
constexpr int RED_NEAR = 3;          // wave 0 handles words 0 .. RED_NEAR - 1 itself (transposed words against the previous `kept`s)
constexpr int RED_GROUPS = 15;       // pusher waves; each owns every 8th half block, so its row loads have ~4 blocks of lead time
constexpr int RED_THREADS = 64 * (1 + RED_GROUPS);
constexpr int RED_SPLIT = 2;         // a block's 64 rows are pushed as 2 units of 32 rows: 64 VGPRs of rows per pusher
constexpr int RED_PUSH_ROWS = 64 / RED_SPLIT;  // (64-row units needed 128 VGPRs and spilled, which serialised the row loads)
constexpr int RED_WIN = 64;         // blocks of (DT, word 1) staged in LDS at a time (2 x 32 KiB)

__device__ __forceinline__ void nms_reduce_body(const u64* __restrict__ mask,
                                                                 const u64* __restrict__ diagT,
                                                                 const u64* __restrict__ w1T,
                                                                 const u64* __restrict__ w2T,
                                                                 const uint32_t* __restrict__ cls_s, int n, int wcap,
                                                                 int max_per_class, const int* __restrict__ seg_start,
                                                                 const NmsRuns& R, bool runs, int* counters,
                                                                 u64* keepbits, u64* dbg) {
  extern __shared__ __attribute__((aligned(16))) u64 removed[];  // [wcap]
  __shared__ u64 dt_s[RED_WIN * 64], wt_s[RED_WIN * 64], wu_s[RED_WIN * 64];
  __shared__ u64 kept_s[RED_WIN];
  __shared__ int flag_s[1 + RED_GROUPS];  // [0] blocks resolved by wave 0; [1 + g] blocks pushed by pusher g
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int grp = wid - 1;  // pushers only
  // runs = categories: the segments are the runs, known from the kernel arguments -- no count / start / end loads
  // (three dependent round trips) and no search (11 more) in front of the block loop
  // (the run table is read with constant indices only: a dynamic index -- or a pointer -- into the kernel-argument
  // struct makes the compiler copy the struct to scratch memory: 560 us)
  const int nseg = runs ? R.n_runs : (cls_s ? counters[0] : 1);
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    if (dbg && seg == 0 && tid == 0) dbg[120] = wall_clock64();
    int s = 0, e = n;
    if (runs) {
#pragma unroll
      for (int q = 0; q < NMS_MAX_RUNS; q++)
        if (q == seg) { s = R.off[q]; e = R.off[q + 1]; }
      if (e <= s) continue;  // (an empty run)
    } else if (cls_s) {
      s = seg_start[seg];
      // upper bound of this category in the ascending cls_s (uniform work, done by every lane)
      uint32_t c = cls_s[s];
      int lo = s, hi = n;
      while (lo < hi) { int mid = (lo + hi) >> 1; if (cls_s[mid] <= c) lo = mid + 1; else hi = mid; }
      e = lo;
    }
    if (e - s > max_per_class) { if (tid == 0) atomicOr(&counters[1], 1); continue; }
    const int b0 = s >> 6, b1 = (e - 1) >> 6;
    const int nb = b1 - b0 + 1;  // <= wcap by construction
    if (dbg && seg == 0 && tid == 0) dbg[121] = wall_clock64();
    __syncthreads();
    if (dbg && seg == 0 && tid == 0) dbg[126] = wall_clock64();
    for (int w = tid; w < nb; w += RED_THREADS) removed[w] = 0;

    // r06, HALF: a segment of <= RED_NEAR + 32 blocks (the RPN's 2,000 per level) needs 32 later words of a row at most,
    // so a pusher's lanes 32..63 hold 32 MORE rows instead of re-reading the last word: a unit is a whole block, the 15
    // pushers hold 15 blocks of rows (7.5 before) and a unit's loads are issued 15 blocks ahead of their use -- the
    // stamps showed wave 0 waiting 1-2 us for a push at 8 of the 32 blocks (profiles/r06/nms_reduce_stamps.txt)
    const bool half = nb <= RED_NEAR + 32;
    const int ushift = half ? 0 : 1;  // log2(units per block); RED_SPLIT == 2
    const int lane_w = half ? (lane & 31) : lane;
    const int hrow = half ? (lane >> 5) * RED_PUSH_ROWS : 0;
    const int wi = min(RED_NEAR + lane_w, wcap - 1);

    // Flag-synchronised pipeline instead of one workgroup barrier per block (measured: 1.5 us per block with
    // the barrier -- every block waited for a pusher to issue its 64 row loads -- against ~0.3 us of actual
    // dependency chain).  wave 0 publishes `kept` of block rel and bumps kept_ready; pusher g bumps
    // pushed[g] after OR-ing a block's rows into removed[]; wave 0 starts block rel once every block <= rel - 2
    // has been pushed (block rel - 1 reaches it through the transposed word 1 w1T and the previous `kept`).  All flags live in
    // LDS; the waves of a workgroup are co-resident, so the spin loops always make progress.
    if (tid <= RED_GROUPS) flag_s[tid] = 0;
    u64 prev_kept = 0, prev2_kept = 0;  // wave 0: `kept` of the previous two blocks of this segment
    for (int bw = b0; bw <= b1; bw += RED_WIN) {  // windows of wave 0's staged inputs (one for <= 4096 boxes)
      __syncthreads();
      if (dbg && seg == 0 && tid == 0) dbg[127] = wall_clock64();
      const int wend = min(b1, bw + RED_WIN - 1);
      const int rows_w = (wend - bw + 1) * 64;
      for (int q = tid; q < rows_w; q += RED_THREADS) {
        const int row = bw * 64 + q;
        const long rc = min(row, n - 1);
        const u64 d = diagT[rc], wt = w1T[rc], wu = w2T[rc];  // (w1T / w2T of the first blocks: never written nor used)
        const bool valid = row >= s && row < e;
        dt_s[q] = valid ? d : 0ull;
        wt_s[q] = (valid && (row >> 6) > b0) ? wt : 0ull;
        wu_s[q] = (valid && (row >> 6) > b0 + 1) ? wu : 0ull;
      }
      if (dbg && seg == 0 && lane == 0 && wid < 2) dbg[124 + wid] = wall_clock64();
      __syncthreads();
      if (dbg && seg == 0 && tid == 0) dbg[122] = wall_clock64();
      if (wid == 0) {
        for (int b = bw; b <= wend; b++) {
          const int rel = b - b0;
          if (dbg && seg == 0 && lane == 0 && rel < 120) dbg[rel] = wall_clock64();  // profiling only
          const u64 dt = dt_s[(b - bw) * 64 + lane], wt = wt_s[(b - bw) * 64 + lane], wu = wu_s[(b - bw) * 64 + lane];
          // wait for the pushers: group g must have finished its blocks <= rel - 2
          if (rel >= RED_NEAR) {
            const int g = lane < RED_GROUPS ? lane : 0;
            const int ulast = ((rel - RED_NEAR + 1) << ushift) - 1;  // last unit of block rel - RED_NEAR
            const int need = ulast >= g ? (ulast - g) / RED_GROUPS + 1 : 0;
            while (__ballot(__hip_atomic_load(&flag_s[1 + g], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need))
              __builtin_amdgcn_s_sleep(1);
          }
          const int row = b * 64 + lane;
          const bool valid = row >= s && row < e;
          const u64 rem = __hip_atomic_load(&removed[rel], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          // suppressed by a kept row two or more blocks back (pushers), or of the previous block (transposed
          // word 1 against the previous `kept`: no LDS atomic + re-read on the chain)
          const bool cj = valid && !((rem >> lane) & 1ull) && ((wt & prev_kept) | (wu & prev2_kept)) == 0ull;
          u64 kept = __ballot(cj);
          int iters = 0;
          for (;;) {
            const u64 nk = __ballot(cj && (dt & kept) == 0ull);
            iters++;
            if (nk == kept) break;
            kept = nk;
          }
          prev2_kept = prev_kept;
          prev_kept = kept;
          if (dbg && seg == 0 && lane == 0 && rel < 120) dbg[128 + rel] = (u64)iters;
          if (lane == 0) {
            kept_s[(b - bw)] = kept;
            if (kept) atomicOr(&keepbits[b], kept);
            __hip_atomic_store(&flag_s[0], rel + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        if (dbg && seg == 0 && lane == 0) dbg[123] = wall_clock64();
      } else {
        // pusher: `rows` = word 2 + lane of the 32 rows of the unit it owns next; the array lives only inside this
        // branch and is redefined by an unconditional (clamped) prefetch on every iteration -- carried across the
        // outer loops or loaded conditionally, the compiler kept two copies, spilled and drained vmcnt per unit
        const int u0 = (bw - b0) << ushift, u1 = ((wend - b0 + 1) << ushift) - 1;
        u64 rows[RED_PUSH_ROWS];
        auto load_unit = [&](int u) {
          // rows up to n64 - 1 are allocated (never kept if >= n)
          // lanes beyond the segment's last block re-read the last needed word (same cache line, masked at use):
          // 64 distinct words per row pulled 5 lines per row through this CU where ~2 are needed
          const int ub = u >> ushift;
          const int wl = min(wi, max(b1 - (b0 + ub), 0));
          const u64* base = mask + ((long)(b0 + ub) * 64 + (u - (ub << ushift)) * RED_PUSH_ROWS + hrow) * wcap + wl;
#pragma unroll
          for (int r = 0; r < RED_PUSH_ROWS; r++) rows[r] = base[(long)r * wcap];
        };
        const int ufirst = u0 + ((grp - u0) % RED_GROUPS + RED_GROUPS) % RED_GROUPS;
        // (MEASURED, r06: the first unit's rows issued in front of the staging barrier -- between the staging loads and
        // their LDS writes -- removed wave 0's 1.6 us wait at block RED_NEAR but the barrier then opened 4 us later: 480 row
        // loads of one CU queue in front of the last waves' staging loads.  26.0 against 24.1 us: not kept.)
        load_unit(min(ufirst, u1));
        for (int u = ufirst; u <= u1; u += RED_GROUPS) {
          const int rel = u >> ushift, b = b0 + rel;  // u % RED_GROUPS == grp
          while (__hip_atomic_load(&flag_s[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= rel)
            __builtin_amdgcn_s_sleep(1);
          const int r0 = (u - (rel << ushift)) * RED_PUSH_ROWS;
          const uint32_t kept = (uint32_t)(kept_s[(b - bw)] >> (r0 + hrow));  // (HALF: the upper lanes' 32 rows)
          const int nlater = b1 - b;
          u64 acc = 0;
#pragma unroll
          for (int r = 0; r < RED_PUSH_ROWS; r++)  // kept rows lie in [s, e) by construction
            acc |= ((kept >> r) & 1u) ? rows[r] : 0ull;
          if ((RED_NEAR + lane_w) <= nlater && acc) atomicOr(&removed[rel + RED_NEAR + lane_w], acc);
          // categories with more than ~4200 boxes: remaining words, fetched now that `kept` is known (never with HALF)
          for (int w = 64 + RED_NEAR + lane; w <= nlater; w += 64) {
            u64 a2 = 0;
            for (int r = 0; r < RED_PUSH_ROWS; r++) {
              if ((kept >> r) & 1u) a2 |= mask[((long)b * 64 + r0 + r) * wcap + w];
            }
            if (a2) atomicOr(&removed[rel + w], a2);
          }
          load_unit(min(u + RED_GROUPS, u1));
          // release: the LDS ORs above are complete before the count becomes visible
          if (lane == 0)
            __hip_atomic_store(&flag_s[1 + grp], u / RED_GROUPS + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
}

// ---- step 4: compaction ------------------------------------------------------------------------
// optional: rows `src` of up to 4 caller arrays copied to position `dst` (keep order) while the kept index is written
// -- what the callers would otherwise do with one torch index kernel per array after their host sync
__device__ __forceinline__ void gather_rows(const d2amd_nms_gather& G, int src, int dst) {
  for (int t = 0; t < G.count; t++) {
    const int words = G.row_bytes[t] >> 2;
    const uint32_t* a = (const uint32_t*)G.src[t] + (long)src * words;
    uint32_t* b = (uint32_t*)G.dst[t] + (long)dst * words;
    for (int q = 0; q < words; q++) b[q] = a[q];
  }
}

// small n: one workgroup does scatter-to-rank-order and ordered compaction out of LDS.
// The kernel is a chain of dependent memory round trips (~2 us each: everything it reads was written by other
// workgroups a moment ago), so the point is to have few of them:
//   general:  {keepbits, rankpos} -> LDS flags | order -> scores | payload -> stores           (4 round trips)
//   DIRECT (runs = categories: the class-major position of a row is the row, so order[r] is also where its kept
//            bit sits):  order -> {kept bit, score} | payload -> stores                          (3 round trips)
// (the first version of the gather added a pass over LDS indices and made it 5).
constexpr int FIN_THREADS = 1024;
template <int W, int CH>
__device__ __forceinline__ void fin_copy_rows(const uint32_t* __restrict__ a, uint32_t* __restrict__ b,
                                              const int (&ord)[CH], uint32_t fl, int at) {
  uint32_t val[CH][W];
#pragma unroll
  for (int q = 0; q < CH; q++)
#pragma unroll
    for (int j = 0; j < W; j++) val[q][j] = a[(long)ord[q] * W + j];
#pragma unroll
  for (int q = 0; q < CH; q++)
    if ((fl >> q) & 1u) {
#pragma unroll
      for (int j = 0; j < W; j++) b[(long)at * W + j] = val[q][j];
      at++;
    }
}
template <bool DIRECT>
__device__ __forceinline__ void nms_finalize_small_body(
    const u64* __restrict__ keepbits, const int* __restrict__ rankpos, const int* __restrict__ order, int n,
    int64_t* __restrict__ keep_out, const int* __restrict__ counters, int64_t* __restrict__ result,
    const float* __restrict__ scores, const d2amd_nms_gather& G, const int* __restrict__ blk_flag) {
  __shared__ uint8_t flags[DIRECT ? 4 : RANK_MAX_N];
  __shared__ int wave_tot[FIN_THREADS / 64];
  __shared__ int s_finite;
  if (threadIdx.x == 0) s_finite = 0;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int CH = RANK_MAX_N / FIN_THREADS;  // 12 consecutive ranks per thread
  const int r0 = tid * CH;
  int ord[CH];
#pragma unroll
  for (int q = 0; q < CH; q++) ord[q] = order_at(order, min(r0 + q, n - 1), n);  // (clamped) loads, all in flight
  if (!DIRECT) {  // kept bit of every segment position -> rank order; the loads of all 12 positions in flight together
    u64 kb[CH];
    int rp[CH];
#pragma unroll
    for (int q = 0; q < CH; q++) {
      const int p = min(tid + q * FIN_THREADS, n - 1);
      kb[q] = keepbits[p >> 6];
      rp[q] = rankpos ? rankpos[p] : p;
    }
#pragma unroll
    for (int q = 0; q < CH; q++) {
      const int p = tid + q * FIN_THREADS;
      if (p < n) flags[min(max(rp[q], 0), n - 1)] = (kb[q] >> (p & 63)) & 1ull ? 1 : 0;
    }
    __syncthreads();
  }
  float sc[CH];
  uint32_t kw[CH];
#pragma unroll
  for (int q = 0; q < CH; q++) {
    sc[q] = scores[ord[q]];  // unconditional: 12 loads in flight (ord is clamped)
    if (DIRECT) kw[q] = reinterpret_cast<const uint32_t*>(keepbits)[ord[q] >> 5];
  }
  int cnt = 0;
  uint32_t fl = 0;
#pragma unroll
  for (int q = 0; q < CH; q++) {
    const int r = r0 + q;
    const bool f = r < n && (DIRECT ? ((kw[q] >> (ord[q] & 31)) & 1u) != 0u : flags[min(r, n - 1)] != 0);
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
  const int off0 = off;
  int fin = 0;
#pragma unroll
  for (int q = 0; q < CH; q++)
    if ((fl >> q) & 1u) {
      keep_out[off++] = (int64_t)ord[q];
      fin += sc[q] > -INFINITY ? 1 : 0;
    }
  if (fin) atomicAdd(&s_finite, fin);
  // rows of the caller's arrays in keep order: this thread's 12 candidate rows are loaded UNCONDITIONALLY (ord is
  // clamped; loads under a per-row branch were issued one divergent branch at a time: 46 us), the kept ones stored
  for (int t = 0; t < G.count; t++) {
    const int words = G.row_bytes[t] >> 2;
    const uint32_t* a = (const uint32_t*)G.src[t];
    uint32_t* b = (uint32_t*)G.dst[t];
    if (words == 1) fin_copy_rows<1, CH>(a, b, ord, fl, off0);
    else if (words == 2) fin_copy_rows<2, CH>(a, b, ord, fl, off0);
    else if (words == 4) fin_copy_rows<4, CH>(a, b, ord, fl, off0);
    else {
      int at = off0;
      for (int q = 0; q < CH; q++)
        if ((fl >> q) & 1u) {
          for (int j = 0; j < words; j++) b[(long)at * words + j] = a[(long)ord[q] * words + j];
          at++;
        }
    }
  }
  __syncthreads();
  // result: {kept, error flags, kept with a score > -inf (callers park invalid rows at -inf: they sort last), 0}
  if (tid == FIN_THREADS - 1) {
    int fl2 = counters[1];
    if (blk_flag)
      for (int q = 0; q < (n + ORDER_THREADS - 1) / ORDER_THREADS; q++) fl2 |= blk_flag[q];
    // a raised flag means the host redoes / rejects this image; a DEVICE-side consumer of the counts (the connected
    // step's sampler reads min(kept, finite)) must then see NO proposals rather than rows of a wrong order
    result[0] = off; result[1] = fl2; result[2] = fl2 ? 0 : s_finite; result[3] = 0;
  }
}

// DIRECT mode across workgroups.  One workgroup doing all of it is bound by its CU's vector L1: every load here is a
// gather (one cache line per lane = 64 L1 cycles per wave-load), 12,288 ranks x (score, kept bit, 5 payload words)
// through ONE L1 is ~25 us whatever the number of round trips.  Here: 1,024 ranks per workgroup (one per thread), a
// workgroup publishes {kept, kept with a finite score} of its chunk and sums what the workgroups before it have
// published (decoupled look-back on device-scope atomics; the <= 12 workgroups of an image are co-resident).
__device__ __forceinline__ void nms_finalize_direct_body(
    const u64* __restrict__ keepbits, const int* __restrict__ order, int n, int64_t* __restrict__ keep_out,
    const int* __restrict__ counters, int64_t* __restrict__ result, const float* __restrict__ scores,
    const d2amd_nms_gather& G, int* __restrict__ blk_cnt, const int* __restrict__ blk_flag) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = blockIdx.x * FIN_THREADS + tid;
  if (blockIdx.x * FIN_THREADS >= n) return;  // uniform (batched launch: the grid is sized for the largest image)
  const int o = order_at(order, min(r, n - 1), n);
  const uint32_t kw = reinterpret_cast<const uint32_t*>(keepbits)[o >> 5];
  const float sc = scores[o];
  const bool f = r < n && ((kw >> (o & 31)) & 1u) != 0u;
  const bool fin = f && sc > -INFINITY;
  __shared__ int wave_cnt[FIN_THREADS / 64], wave_fin[FIN_THREADS / 64];
  __shared__ int s_before[2];
  const unsigned long long bal = __ballot(f), balf = __ballot(fin);
  if (lane == 0) { wave_cnt[wid] = __builtin_popcountll(bal); wave_fin[wid] = __builtin_popcountll(balf); }
  __syncthreads();
  int cnt = 0, cfin = 0, before_w = 0;
#pragma unroll
  for (int w = 0; w < FIN_THREADS / 64; w++) {
    if (w < wid) before_w += wave_cnt[w];
    cnt += wave_cnt[w];
    cfin += wave_fin[w];
  }
  // publish {kept + 1, finite} of this chunk (0 = not there yet), then add up the chunks before this one
  if (tid == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(blk_cnt) + blockIdx.x,
                       ((unsigned long long)(unsigned)cfin << 32) | (unsigned)(cnt + 1), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  if (wid == 0) {
    int bk = 0, bf = 0;
    for (int j = lane; j < (int)blockIdx.x; j += 64) {
      unsigned long long v;
      while (((v = __hip_atomic_load(reinterpret_cast<unsigned long long*>(blk_cnt) + j, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT)) & 0xffffffffull) == 0ull)
        __builtin_amdgcn_s_sleep(1);
      bk += (int)(v & 0xffffffffull) - 1;
      bf += (int)(v >> 32);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { bk += __shfl_xor(bk, d); bf += __shfl_xor(bf, d); }
    if (lane == 0) { s_before[0] = bk; s_before[1] = bf; }
  }
  __syncthreads();
  const int at = s_before[0] + before_w + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
  if (f) {
    keep_out[at] = (int64_t)o;
    if (G.count) gather_rows(G, o, at);
  }
  if (blockIdx.x == (unsigned)((n + FIN_THREADS - 1) / FIN_THREADS) - 1 && tid == 0) {
    int fl2 = counters[1];
    if (blk_flag)
      for (int q = 0; q < (n + ORDER_THREADS - 1) / ORDER_THREADS; q++) fl2 |= blk_flag[q];
    result[0] = s_before[0] + cnt; result[1] = fl2; result[2] = fl2 ? 0 : s_before[1] + cfin; result[3] = 0;  // (as above)
  }
}

__global__ void nms_scatter_flags_kernel(const u64* __restrict__ keepbits, const int* __restrict__ rankpos, int n,
                                         uint8_t* __restrict__ flag_r) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  bool kept = (keepbits[p >> 6] >> (p & 63)) & 1ull;
  int r = rankpos ? rankpos[p] : p;
  flag_r[r] = kept ? 1 : 0;
}

// Ordered compaction of the kept flags (rank order) for the large-n path, in two launches over n / 1024 workgroups:
// counts per workgroup, then every workgroup sums the counts before it and scatters.  (One workgroup walking all
// 100,000 ranks took 110-140 us per image, r02 profile.)
constexpr int COMPACT_BLOCK = 1024;
__global__ __launch_bounds__(COMPACT_BLOCK) void nms_compact_count_kernel(const uint8_t* __restrict__ flag_r, int n,
                                                                          int* __restrict__ blk_cnt,
                                                                          int64_t* __restrict__ result) {
  const int r = blockIdx.x * COMPACT_BLOCK + threadIdx.x;
  const int c = __syncthreads_count(r < n && flag_r[min(r, n - 1)]);
  if (threadIdx.x == 0) {
    blk_cnt[blockIdx.x] = c;
    if (blockIdx.x == 0) result[2] = 0;  // kept boxes with a score > -inf: accumulated by the scatter launch
  }
}
__global__ __launch_bounds__(COMPACT_BLOCK) void nms_compact_kernel(const uint8_t* __restrict__ flag_r,
                                                                    const int* __restrict__ order, int n,
                                                                    const int* __restrict__ blk_cnt,
                                                                    int64_t* __restrict__ keep_out,
                                                                    const int* __restrict__ counters,
                                                                    int64_t* __restrict__ result,
                                                                    const float* __restrict__ scores,
                                                                    const d2amd_nms_gather G) {
  __shared__ int wave_cnt[COMPACT_BLOCK / 64];
  __shared__ int s_red[COMPACT_BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // workgroups before this one
  int before = 0;
  for (int j = tid; j < (int)blockIdx.x; j += COMPACT_BLOCK) before += blk_cnt[j];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o);
  if (lane == 0) s_red[wid] = before;
  const int r = blockIdx.x * COMPACT_BLOCK + tid;
  const bool f = r < n && flag_r[min(r, n - 1)];
  const unsigned long long bal = __ballot(f);
  if (lane == 0) wave_cnt[wid] = __builtin_popcountll(bal);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < COMPACT_BLOCK / 64; k++) {
    off += s_red[k];
    if (k < wid) off += wave_cnt[k];
    tot += wave_cnt[k];
  }
  bool fin = false;
  if (f) {
    const int o = order_at(order, r, n);
    const int at = off + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
    keep_out[at] = (int64_t)o;
    if (G.count) gather_rows(G, o, at);
    fin = scores[o] > -INFINITY;
  }
  const int nfin = __syncthreads_count(fin);
  if (tid == 0) {
    if (nfin) atomicAdd((unsigned long long*)&result[2], (unsigned long long)nfin);
    if (blockIdx.x == gridDim.x - 1) {
      int base = 0;
      for (int k = 0; k < COMPACT_BLOCK / 64; k++) base += s_red[k];
      result[0] = base + tot; result[1] = counters[1]; result[3] = 0;
    }
  }
}

// ---- batched launch: blockIdx.z = image --------------------------------------------------------
// The per-image pipelines of a batch (RPN / RetinaNet NMS of every image) run as ONE launch per stage: half the
// launches and no stream fork/join on the host, and the latency-bound stages (reduction, compaction) of all
// images overlap on the device.  The grid of a stage is sized for the largest image; the bodies exit early.
constexpr int NMS_MAX_BATCH = 8;
struct NmsImg {
  const float* boxes;
  const float* scores;
  const int64_t* idxs;
  int n, wcap, mpc, rk_chunks;
  NmsWorkspace w;
  int64_t* keep_out;
  int64_t* result;
  const int* rankpos;      // null without categories: segment order == rank order
  const uint32_t* cls_s;
  d2amd_nms_gather gather; // rows of caller arrays to copy in keep order (count 0: none)
};
struct NmsBatch {
  double thr, mid;
  u64* dbg;
  int count;
  NmsRuns runs;  // n_runs == 0: no pre-sorted runs
  int order_flags;  // 1: nms_runs_order_small_kernel ran -- its per-workgroup "not in order" words join the flags
  int rot_plain;    // D2AMD_NMS_ROT_PLAIN (A/B and test switch): the rotated mask without pair compaction
  NmsImg img[NMS_MAX_BATCH];
};

__global__ void nms_prep_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_prep_body(I.scores, I.idxs, I.n, (uint4*)I.w.rk_keys, (uint32_t*)I.w.keepbits, (int)(I.w.zero_bytes / 4));
}
template <int BW>
__global__ __launch_bounds__(RK_THREADS) void nms_rank_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_rank_body<BW>(I.boxes, I.idxs, (const uint4*)I.w.rk_keys, I.n, I.w.order, I.w.rankpos, I.w.cls_s, I.w.boxes_s,
                    I.w.counters, I.w.rk_cnt, I.rk_chunks);
}
template <int BW>
__global__ void nms_rank_scatter_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_rank_scatter_body<BW>(I.boxes, (const uint4*)I.w.rk_keys, I.n, I.w.rk_cnt, I.rk_chunks, I.w.order, I.w.rankpos,
                            I.w.cls_s, I.w.boxes_s);
}
template <int PER>
__global__ __launch_bounds__(RUNS_SCAN_THREADS) void nms_runs_scan_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_runs_scan_body<PER>(I.scores, B.runs, (uint32_t*)I.w.keys_out, (int*)I.w.cls_r, I.w.run_cnt, (uint32_t*)I.w.keepbits,
                          (int)(I.w.zero_bytes / 4), I.n <= RANK_MAX_N ? I.w.blk_cnt : nullptr, 2 * (I.n / 1024 + 2));
}
template <int BW>
__global__ __launch_bounds__(ORDER_THREADS) void nms_runs_order_small_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_runs_order_small_body<BW>(I.boxes, I.scores, I.n, B.runs, I.w.order, I.w.rankpos, I.w.cls_s, I.w.boxes_s,
                                (uint32_t*)I.w.keepbits, (int)(I.w.zero_bytes / 4), I.w.blk_cnt, 2 * (I.n / 1024 + 2),
                                I.w.run_cnt + 16, I.idxs ? 0 : 1);
}
constexpr int RUNS_RANK_THREADS = 1024;  // every workgroup stages all samples: few, large workgroups
template <int BW>
__global__ __launch_bounds__(RUNS_RANK_THREADS) void nms_runs_rank_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_runs_rank_body<BW>(I.boxes, I.scores, I.n, B.runs, (const uint32_t*)I.w.keys_out, (const int*)I.w.cls_r,
                         I.w.run_cnt, I.w.order, I.w.rankpos, I.w.cls_s, I.w.boxes_s, I.w.seg_start, I.w.counters,
                         I.idxs ? 0 : 1);
}
__global__ void nms_segments_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  if (I.cls_s) nms_segments_body(I.w.cls_s, I.n, I.w.seg_start, I.w.counters);
}
template <bool FAST, bool TIE_UP>
__global__ __launch_bounds__(64) void nms_mask_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_mask_body<FAST, TIE_UP>(I.w.boxes_s, I.cls_s ? 1 : 0, I.n, I.wcap, B.thr, B.mid, I.w.mask, I.w.diagT, I.w.w1T,
                              I.w.w2T);
}
__global__ __launch_bounds__(64) void nms_mask_rot_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_mask_rot_body(I.w.boxes_s, I.n, I.wcap, B.thr, I.w.mask, I.w.diagT, I.w.w1T, I.w.w2T, B.rot_plain);
}
__global__ __launch_bounds__(64 * ROTW, ROT_OCC) void nms_mask_rot_wg_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_mask_rot_wg_body(I.w.boxes_s, I.n, I.wcap, B.thr, I.w.mask, I.w.diagT, I.w.w1T, I.w.w2T);
}
__global__ __launch_bounds__(RED_THREADS) void nms_reduce_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_reduce_body(I.w.mask, I.w.diagT, I.w.w1T, I.w.w2T, I.cls_s, I.n, I.wcap, I.mpc, I.w.seg_start,
                  B.runs, B.runs.are_cls != 0, I.w.counters, I.w.keepbits, blockIdx.z == 0 ? B.dbg : nullptr);
}
__global__ __launch_bounds__(FIN_THREADS) void nms_finalize_direct_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_finalize_direct_body(I.w.keepbits, I.w.order, I.n, I.keep_out, I.w.counters, I.result, I.scores, I.gather,
                           I.w.blk_cnt, B.order_flags ? I.w.run_cnt + 16 : nullptr);
}
template <bool DIRECT>
__global__ __launch_bounds__(FIN_THREADS) void nms_finalize_small_kernel(const NmsBatch B) {
  const NmsImg& I = B.img[blockIdx.z];
  nms_finalize_small_body<DIRECT>(I.w.keepbits, I.rankpos, I.w.order, I.n, I.keep_out, I.w.counters, I.result,
                                  I.scores, I.gather, B.order_flags ? I.w.run_cnt + 16 : nullptr);
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

// threshold constants of the division-free mask test (see nms_mask_body): f = largest float <= thr, g = next up
struct MaskThr { double mid; bool fast, tie_up; };
static MaskThr mask_thr(double thr) {
  float f = (float)thr;
  if ((double)f > thr) f = nextafterf(f, -INFINITY);
  const float g = nextafterf(f, INFINITY);
  uint32_t gbits;
  memcpy(&gbits, &g, 4);
  MaskThr m;
  m.fast = thr >= 1e-30 && thr <= 1e30;  // f, g normal
  m.mid = ((double)f + (double)g) * 0.5;
  m.tie_up = (gbits & 1u) == 0u;
  return m;
}

// mask + reduction of the images in B (every image already has its segment-ordered box records)
static int nms_mask_reduce(NmsBatch& B, int rotated, bool any_cls, hipStream_t s) {
  int nb_max = 0, wcap_max = 0;
  for (int k = 0; k < B.count; k++) {
    nb_max = std::max(nb_max, (B.img[k].n + 63) / 64);
    wcap_max = std::max(wcap_max, B.img[k].wcap);
  }
  // ~8,000 waves per image keep the chip busy; beyond that more workgroups per row block only add dispatches
  int gy = std::max(std::min(wcap_max, 8), std::min(std::min(wcap_max, MASK_GRID_Y), cdiv(8192, std::max(nb_max, 1))));
  if (const char* e = d2_prof_env("D2AMD_NMS_MASK_GY")) gy = std::max(1, std::min(atoi(e), wcap_max));  // (profiling builds)
  const dim3 mgrid(nb_max, gy, B.count);
  const bool timed_mask = timing_begin("nms_mask", s);
  if (rotated) {
    static const bool rot_plain = d2_prof_env("D2AMD_NMS_ROT_PLAIN") != nullptr;
    B.rot_plain = rot_plain ? 1 : 0;
    // thr > 0: the workgroup tile (r06); D2AMD_NMS_ROT_WAVE (profiling builds): r04's one-wave tile.  thr <= 0 "suppresses"
    // the pairs the distance test rejects too: the plain tile.
    static const bool rot_wave = d2_prof_env("D2AMD_NMS_ROT_WAVE") != nullptr;
    if (B.thr > 0.0 && !rot_plain && !rot_wave) hipLaunchKernelGGL(nms_mask_rot_wg_kernel, mgrid, dim3(64 * ROTW), 0, s, B);
    else hipLaunchKernelGGL(nms_mask_rot_kernel, mgrid, dim3(64), 0, s, B);
  } else {
    const MaskThr m = mask_thr(B.thr);
    B.mid = m.mid;
    static const bool mask_exact = d2_prof_env("D2AMD_NMS_MASK_EXACT") != nullptr;  // test switch: literal formula
    if (!m.fast || mask_exact) hipLaunchKernelGGL((nms_mask_kernel<false, false>), mgrid, dim3(64), 0, s, B);
    else if (m.tie_up) hipLaunchKernelGGL((nms_mask_kernel<true, true>), mgrid, dim3(64), 0, s, B);
    else hipLaunchKernelGGL((nms_mask_kernel<true, false>), mgrid, dim3(64), 0, s, B);
  }
  if (timed_mask) timing_end("nms_mask", s);
  D2_LAUNCH_OK();
  const int rgrid = any_cls ? 512 : 1;
  const char* red_stamps = d2_prof_env("D2AMD_NMS_STAMPS");  // profiling only: per-block stamps of segment 0 of image 0
  B.dbg = nullptr;
  if (red_stamps) {
    D2_HIP_OK(hipMalloc(&B.dbg, 256 * 8));
    D2_HIP_OK(hipMemsetAsync(B.dbg, 0, 256 * 8, s));
  }
  const bool timed_red = timing_begin("nms_reduce", s);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(rgrid, 1, B.count), dim3(RED_THREADS), (size_t)wcap_max * 8, s, B);
  if (timed_red) timing_end("nms_reduce", s);
  D2_LAUNCH_OK();
  if (red_stamps) {
    u64 h[256];
    D2_HIP_OK(hipStreamSynchronize(s));
    D2_HIP_OK(hipMemcpy(h, B.dbg, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[d2amd nms] per-block 10ns ticks (fixed-point iterations):");
    for (int i = 1; i < 120 && h[i]; i++) fprintf(stderr, " %llu(%llu)", h[i] - h[i - 1], h[128 + i - 1]);
    fprintf(stderr, "\n[d2amd nms] phases (10ns ticks): segment search %llu, staging %llu, block loop %llu; staging loads done wave0 +%llu, pusher +%llu; sync1 +%llu sync2 +%llu\n",
            h[121] - h[120], h[122] - h[121], h[123] - h[122], h[124] - h[121], h[125] - h[121], h[126] - h[121], h[127] - h[121]);
    (void)hipFree(B.dbg);
    B.dbg = nullptr;
  }
  return D2AMD_OK;
}

// whole pipeline of a batch of images that all take the brute-force ranking path (n <= RANK_MAX_N)
static int nms_run_small(NmsBatch& B, int rotated, hipStream_t s) {
  const int T = 256;
  int n_max = 0, chunks_max = 0;
  bool any_cls = false;
  for (int k = 0; k < B.count; k++) {
    n_max = std::max(n_max, B.img[k].n);
    chunks_max = std::max(chunks_max, B.img[k].rk_chunks);
    any_cls |= B.img[k].idxs != nullptr;
  }
  const dim3 ngrid(cdiv(n_max, T), 1, B.count);
  // pre-sorted runs: the order comes from one scan + one binary-search kernel.  (Runs + the caller's class ids on
  // this path would still need the class-major positions: the brute-force ranking below provides both.)
  const bool by_runs = B.runs.n_runs > 0 && (B.runs.are_cls || !any_cls);
  if (by_runs) {  // scan + rank + records + zeroing: one launch
    B.order_flags = 1;
    const dim3 ogrid(cdiv(n_max, ORDER_THREADS), 1, B.count);
    if (rotated) hipLaunchKernelGGL((nms_runs_order_small_kernel<5>), ogrid, dim3(ORDER_THREADS), 0, s, B);
    else hipLaunchKernelGGL((nms_runs_order_small_kernel<4>), ogrid, dim3(ORDER_THREADS), 0, s, B);
    D2_LAUNCH_OK();
    any_cls = B.runs.are_cls != 0;
  } else {
  const int npad = cdiv(n_max, RK_GROUP) * RK_GROUP;
  hipLaunchKernelGGL(nms_prep_kernel, dim3(cdiv(npad, T), 1, B.count), dim3(T), 0, s, B);
  D2_LAUNCH_OK();
  const dim3 rk_grid(chunks_max, cdiv(cdiv(n_max, 64), RK_THREADS / 64), B.count);
  if (rotated) hipLaunchKernelGGL((nms_rank_kernel<5>), rk_grid, dim3(RK_THREADS), 0, s, B);
  else hipLaunchKernelGGL((nms_rank_kernel<4>), rk_grid, dim3(RK_THREADS), 0, s, B);
  D2_LAUNCH_OK();
  if (rotated) hipLaunchKernelGGL((nms_rank_scatter_kernel<5>), ngrid, dim3(T), 0, s, B);
  else hipLaunchKernelGGL((nms_rank_scatter_kernel<4>), ngrid, dim3(T), 0, s, B);
  D2_LAUNCH_OK();
  if (any_cls) {
    hipLaunchKernelGGL(nms_segments_kernel, ngrid, dim3(T), 0, s, B);
    D2_LAUNCH_OK();
  }
  }
  const int rc = nms_mask_reduce(B, rotated, any_cls, s);
  if (rc != D2AMD_OK) return rc;
  if (by_runs && B.runs.are_cls)  // class-major position == row: order[r] is also the kept bit's position
    hipLaunchKernelGGL(nms_finalize_direct_kernel, dim3(cdiv(n_max, FIN_THREADS), 1, B.count), dim3(FIN_THREADS), 0, s, B);
  else
    hipLaunchKernelGGL(nms_finalize_small_kernel<false>, dim3(1, 1, B.count), dim3(FIN_THREADS), 0, s, B);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

static int nms_fill_img(NmsImg& I, const float* boxes, const float* scores, const int64_t* idxs, int64_t n,
                        int64_t max_per_class, int64_t* keep_out, int64_t* result, void* workspace,
                        size_t workspace_bytes) {
  D2_CHECK_ARG(n > 0 && n < (1ll << 31) - 64, "nms: bad n %lld", (long long)n);
  D2_CHECK_ARG(boxes && scores && keep_out && workspace && result, "nms: null pointer");
  I.boxes = boxes;
  I.scores = scores;
  I.idxs = idxs;
  I.n = (int)n;
  I.wcap = wcap_for(n, max_per_class);
  I.mpc = (max_per_class <= 0 || max_per_class > n) ? (int)n : (int)max_per_class;
  I.rk_chunks = cdiv((int)n, RK_JC);
  carve(I.w, workspace, n, I.wcap);
  if (workspace_bytes < I.w.total) {
    set_error("nms: workspace too small (%zu < %zu)", workspace_bytes, I.w.total);
    return D2AMD_EWORKSPACE;
  }
  I.keep_out = keep_out;
  I.result = result;
  I.rankpos = idxs ? I.w.rankpos : nullptr;
  I.cls_s = idxs ? I.w.cls_s : nullptr;
  return D2AMD_OK;
}

static int nms_gather_arg(d2amd_nms_gather& G, const d2amd_nms_gather& in) {
  D2_CHECK_ARG(in.count >= 0 && in.count <= 4, "nms: gather of %d arrays (max 4)", in.count);
  for (int t = 0; t < in.count; t++)
    D2_CHECK_ARG(in.src[t] && in.dst[t] && in.row_bytes[t] > 0 && in.row_bytes[t] % 4 == 0,
                 "nms: gather array %d: null pointer or a row size that is not a multiple of 4 bytes", t);
  G = in;
  return D2AMD_OK;
}

// `runs` (optional): the inputs are sequences of pre-sorted runs (see step 1c); runs->are_cls: the runs are the
// categories (idxs must be null)
static int nms_batched_impl(int count, const float* const* boxes, const float* const* scores,
                            const int64_t* const* idxs, const int64_t* n, double iou_threshold, int rotated,
                            const int64_t* max_per_class, int64_t* const* keep_out, int64_t* const* result,
                            void* const* workspace, const size_t* workspace_bytes, const NmsRuns* runs,
                            const d2amd_nms_gather* gather, hipStream_t s) {
  D2_CHECK_ARG(count >= 0 && boxes && scores && n && keep_out && result && workspace && workspace_bytes,
               "nms_batched: null pointer");
  for (int k0 = 0; k0 < count; k0 += NMS_MAX_BATCH) {
    NmsBatch B;
    memset(&B, 0, sizeof(B));
    B.thr = iou_threshold;
    if (runs) B.runs = *runs;
    for (int k = k0; k < count && k < k0 + NMS_MAX_BATCH; k++) {
      D2_CHECK_ARG(n[k] >= 0 && result[k], "nms_batched: bad image %d", k);
      if (n[k] == 0) {
        { const int zrc = zero_async(result[k], 32, s); if (zrc) return zrc; }
        continue;
      }
      D2_CHECK_ARG(n[k] <= RANK_MAX_N, "nms_batched: image %d has %lld boxes (> %d): use d2amd_nms", k,
                   (long long)n[k], RANK_MAX_N);
      D2_CHECK_ARG(!runs || n[k] == runs->off[runs->n_runs], "nms_batched: image %d has %lld boxes, the runs cover %d",
                   k, (long long)n[k], runs ? runs->off[runs->n_runs] : 0);
      NmsImg& I = B.img[B.count];
      const int rc = nms_fill_img(I, boxes[k], scores[k], idxs ? idxs[k] : nullptr, n[k],
                                  max_per_class ? max_per_class[k] : 0, keep_out[k], result[k], workspace[k],
                                  workspace_bytes[k]);
      if (rc != D2AMD_OK) return rc;
      if (runs && runs->are_cls) {
        D2_CHECK_ARG(I.idxs == nullptr, "nms_batched: runs are the categories: idxs must be null");
        I.rankpos = I.w.rankpos;
        I.cls_s = I.w.cls_s;
      }
      if (gather) {
        const int grc = nms_gather_arg(I.gather, gather[k]);
        if (grc) return grc;
      }
      B.count++;
    }
    if (B.count) {
      const int rc = nms_run_small(B, rotated, s);
      if (rc != D2AMD_OK) return rc;
    }
  }
  return D2AMD_OK;
}

static int nms_runs_arg(NmsRuns& R, const int* run_offsets, int n_runs, int runs_are_categories, int num_categories) {
  D2_CHECK_ARG(run_offsets && n_runs >= 1 && n_runs <= NMS_MAX_RUNS, "nms: %d pre-sorted runs (1..%d)", n_runs,
               NMS_MAX_RUNS);
  memset(&R, 0, sizeof(R));
  R.n_runs = n_runs;
  R.are_cls = runs_are_categories ? 1 : 0;
  D2_CHECK_ARG(run_offsets[0] == 0, "nms: run_offsets must start at 0");
  for (int r = 0; r <= n_runs; r++) {
    D2_CHECK_ARG(r == 0 || run_offsets[r] >= run_offsets[r - 1], "nms: run_offsets must not decrease");
    R.off[r] = run_offsets[r];
  }
  for (int r = n_runs + 1; r <= NMS_MAX_RUNS; r++) R.off[r] = run_offsets[n_runs];
  D2_CHECK_ARG(num_categories >= 0 && num_categories <= 65536, "nms: num_categories %d not in [0, 65536]", num_categories);
  R.cat_bits = 0;
  if (num_categories > 0)
    while ((1 << R.cat_bits) < num_categories) R.cat_bits++;
  if (num_categories == 1) R.cat_bits = 1;
  return D2AMD_OK;
}

extern "C" int d2amd_nms_batched(int count, const float* const* boxes, const float* const* scores,
                                 const int64_t* const* idxs, const int64_t* n, double iou_threshold, int rotated,
                                 const int64_t* max_per_class, int64_t* const* keep_out, int64_t* const* result,
                                 void* const* workspace, const size_t* workspace_bytes, void* stream) {
  return nms_batched_impl(count, boxes, scores, idxs, n, iou_threshold, rotated, max_per_class, keep_out, result,
                          workspace, workspace_bytes, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int d2amd_nms_batched_runs(int count, const float* const* boxes, const float* const* scores,
                                      const int64_t* const* idxs, const int64_t* n, const int* run_offsets,
                                      int n_runs, int runs_are_categories, int num_categories, double iou_threshold,
                                      int rotated, const int64_t* max_per_class, int64_t* const* keep_out,
                                      int64_t* const* result, void* const* workspace, const size_t* workspace_bytes,
                                      const d2amd_nms_gather* gather, void* stream) {
  NmsRuns R;
  const int rc = nms_runs_arg(R, run_offsets, n_runs, runs_are_categories, num_categories);
  if (rc) return rc;
  D2_CHECK_ARG(!(runs_are_categories && idxs), "nms_batched_runs: runs are the categories: idxs must be null");
  return nms_batched_impl(count, boxes, scores, idxs, n, iou_threshold, rotated, max_per_class, keep_out, result,
                          workspace, workspace_bytes, &R, gather, (hipStream_t)stream);
}

extern "C" int d2amd_nms_batched_max_boxes(void) { return RANK_MAX_N; }

static int nms_impl(const float* boxes, const float* scores, const int64_t* idxs, int64_t n, double iou_threshold,
                    int rotated, int64_t max_per_class, int64_t* keep_out, int64_t* result, void* workspace,
                    size_t workspace_bytes, const NmsRuns* runs, const d2amd_nms_gather* gather, hipStream_t s) {
  D2_CHECK_ARG(n >= 0 && n < (1ll << 31) - 64, "nms: bad n %lld", (long long)n);
  D2_CHECK_ARG(result, "nms: null result");
  if (n == 0) {
    { const int zrc = zero_async(result, 32, s); if (zrc) return zrc; }
    return D2AMD_OK;
  }
  D2_CHECK_ARG(!runs || n == runs->off[runs->n_runs], "nms: %lld boxes, the runs cover %d", (long long)n,
               runs ? runs->off[runs->n_runs] : 0);
  NmsBatch B;
  memset(&B, 0, sizeof(B));
  B.thr = iou_threshold;
  B.count = 1;
  if (runs) B.runs = *runs;
  NmsImg& I = B.img[0];
  int rc = nms_fill_img(I, boxes, scores, idxs, n, max_per_class, keep_out, result, workspace, workspace_bytes);
  if (rc != D2AMD_OK) return rc;
  if (gather) {
    rc = nms_gather_arg(I.gather, *gather);
    if (rc) return rc;
  }
  const bool run_cls = runs && runs->are_cls;
  if (run_cls) {
    D2_CHECK_ARG(idxs == nullptr, "nms: runs are the categories: idxs must be null");
    I.rankpos = I.w.rankpos;
    I.cls_s = I.w.cls_s;
  }
  if (n <= RANK_MAX_N) return nms_run_small(B, rotated, s);
  // ---- large inputs: counting-sort passes (or the runs' binary searches) instead of the brute-force ranking ----
  NmsWorkspace& w = I.w;
  const int N = (int)n;
  const int T = 256;
  if (runs) {  // order[] from the pre-sorted runs (the scan also zeroes the accumulators)
    hipLaunchKernelGGL(nms_runs_scan_kernel<16>, dim3(std::max(runs->n_runs, 16), 1, 1), dim3(RUNS_SCAN_THREADS), 0, s, B);
    D2_LAUNCH_OK();
    const dim3 rgrid(cdiv(N, RUNS_RANK_THREADS), 1, 1);
    if (rotated) hipLaunchKernelGGL((nms_runs_rank_kernel<5>), rgrid, dim3(RUNS_RANK_THREADS), 0, s, B);
    else hipLaunchKernelGGL((nms_runs_rank_kernel<4>), rgrid, dim3(RUNS_RANK_THREADS), 0, s, B);
    D2_LAUNCH_OK();
  } else {
    { const int zrc = zero_async(w.keepbits, w.zero_bytes, s); if (zrc) return zrc; }
    CsPass a{};  // score order: four 8-bit passes; rankpos / cls_r are free until the class sort
    a.n = N; a.hist = w.cs_hist; a.scores = scores; a.counters = w.counters;
    rc = cs_sort(a, 32, (uint32_t*)w.rankpos, w.cls_r, w.iota, (uint32_t*)w.keys_out, w.order, s);
    if (rc) return rc;
  }
  if (idxs) {
    // (a caller that knows its number of categories -- 80 classes: 7 bits -- saves the sort its second 8-bit pass)
    const unsigned cat_bits = runs && runs->cat_bits > 0 ? (unsigned)runs->cat_bits : 16u;
    CsPass a{};  // stable by class over the score order: rankpos[p] = rank of the element at class-major position p
    a.n = N; a.hist = w.cs_hist; a.idxs = idxs; a.order = w.order; a.counters = w.counters;
    rc = cs_sort(a, (int)cat_bits, w.cls_r, (uint32_t*)w.keys_out, w.iota, w.cls_s, w.rankpos, s);
    if (rc) return rc;
  }
  if (!runs || idxs) {  // (runs without class ids: the rank kernel has written the records)
    if (rotated)
      hipLaunchKernelGGL((nms_gather_boxes_kernel<5>), dim3(cdiv(N, T)), dim3(T), 0, s, boxes, w.order, I.rankpos,
                         I.cls_s, N, w.boxes_s, w.seg_start, w.counters);
    else
      hipLaunchKernelGGL((nms_gather_boxes_kernel<4>), dim3(cdiv(N, T)), dim3(T), 0, s, boxes, w.order, I.rankpos,
                         I.cls_s, N, w.boxes_s, w.seg_start, w.counters);
    D2_LAUNCH_OK();
  }
  rc = nms_mask_reduce(B, rotated, idxs != nullptr || run_cls, s);
  if (rc != D2AMD_OK) return rc;
  hipLaunchKernelGGL(nms_scatter_flags_kernel, dim3(cdiv(N, T)), dim3(T), 0, s, w.keepbits, I.rankpos, N, w.flag_r);
  D2_LAUNCH_OK();
  hipLaunchKernelGGL(nms_compact_count_kernel, dim3(cdiv(N, COMPACT_BLOCK)), dim3(COMPACT_BLOCK), 0, s, w.flag_r, N, w.blk_cnt,
                     result);
  D2_LAUNCH_OK();
  hipLaunchKernelGGL(nms_compact_kernel, dim3(cdiv(N, COMPACT_BLOCK)), dim3(COMPACT_BLOCK), 0, s, w.flag_r, w.order, N,
                     w.blk_cnt, keep_out, w.counters, result, scores, I.gather);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

extern "C" int d2amd_nms(const float* boxes, const float* scores, const int64_t* idxs, int64_t n,
                         double iou_threshold, int rotated, int64_t max_per_class, int64_t* keep_out,
                         int64_t* result, void* workspace, size_t workspace_bytes, void* stream) {
  return nms_impl(boxes, scores, idxs, n, iou_threshold, rotated, max_per_class, keep_out, result, workspace,
                  workspace_bytes, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int d2amd_nms_runs(const float* boxes, const float* scores, const int64_t* idxs, int64_t n,
                              const int* run_offsets, int n_runs, int runs_are_categories, int num_categories,
                              double iou_threshold, int rotated, int64_t max_per_class, int64_t* keep_out,
                              int64_t* result,
                              void* workspace, size_t workspace_bytes, const d2amd_nms_gather* gather,
                              void* stream) {
  NmsRuns R;
  const int rc = nms_runs_arg(R, run_offsets, n_runs, runs_are_categories, num_categories);
  if (rc) return rc;
  return nms_impl(boxes, scores, idxs, n, iou_threshold, rotated, max_per_class, keep_out, result, workspace,
                  workspace_bytes, &R, gather, (hipStream_t)stream);
}
