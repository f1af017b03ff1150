// NMS / batched NMS / rotated NMS, entirely on the device (no D2H copy of the bitmask, no
// CPU sweep -- cf. the reference's csrc/nms_rotated/nms_rotated_cuda.cu:114-137 and
// torchvision's identical pattern).
//
// Pipeline (all on `stream`):
//   1. radix sort of scores, descending (stable)                      -> rank order
//   2. (batched) stable radix sort of the rank-ordered boxes by category (16 bits)
//        -> category-major, score-descending inside each category
//   3. gather boxes into that order; find category segments
//   4. wavefront bitmask kernel: one 64-lane wave per 64x64 tile of the (sorted) IoU matrix,
//        lane = row box, one uint64 word per lane; only tiles on/above the diagonal whose
//        category ranges overlap are evaluated (pair count = sum_c n_c^2/2, not N^2/2)
//   5. greedy reduction: one wave per category segment; 64-row diagonal blocks are resolved
//        with scalar readlane/bit ops, surviving rows are OR-ed into an LDS "removed" bitset
//   6. ordered compaction back to rank order -> original indices, count
// Bit-exactness: IoU arithmetic is evaluated exactly as torchvision's CPU nms / the reference's
// nms_rotated_cpu.cpp (fp32, no FMA contraction, IEEE divide, threshold compare in double).
#pragma clang fp contract(off)
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "common.h"
#include "rotated_iou.h"

namespace d2amd {

typedef unsigned long long u64;

struct NmsWorkspace {
  float* keys_out;    // [n] sorted scores
  int* iota;          // [n] 0..n-1
  int* order;         // [n] rank -> original index
  uint32_t* cls_r;    // [n] category in rank order
  uint32_t* cls_s;    // [n] category in segment order
  int* rankpos;       // [n] segment position -> rank
  float* boxes_s;     // [n64 * 8] boxes in segment order (aligned: float4; rotated: 5 of 8 floats)
  u64* mask;          // [n64 * wcap]
  u64* keepbits;      // [nblocks]
  uint8_t* flag_r;    // [n] kept flag in rank order
  int* seg_start;     // [65536]
  int* counters;      // [4]: nseg, error flags
  void* sort_temp;
  size_t sort_temp_bytes;
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
  w.keepbits = (u64*)take((n64 / 64) * 8);
  w.flag_r = (uint8_t*)take(n);
  w.seg_start = (int*)take(65536 * 4);
  w.counters = (int*)take(4 * 4);
  w.sort_temp_bytes = sort_temp_bytes(n);
  w.sort_temp = take(w.sort_temp_bytes);
  w.total = off;
}

// ---- step 0/2 helpers ------------------------------------------------------------------
__global__ void nms_init_kernel(int* iota, int n, u64* keepbits, int nblocks, uint8_t* flag_r, int* counters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { iota[i] = i; flag_r[i] = 0; }
  if (i < nblocks) keepbits[i] = 0;
  if (i < 4) counters[i] = 0;
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

// ---- step 4: wavefront bitmask ------------------------------------------------------------
__device__ __forceinline__ float bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <bool ROT>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes_s,
                                                      const uint32_t* __restrict__ cls_s, int n, int wcap,
                                                      double thr, u64* __restrict__ mask) {
  const int rb = blockIdx.x, w = blockIdx.y;
  const int lane = threadIdx.x;
  const int cb = rb + w;
  const int row = rb * 64 + lane;
  const int col0 = cb * 64;
  const int nblocks = (n + 63) >> 6;
  if (cb >= nblocks) return;  // never read by the reduction
  u64 word = 0;
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
#pragma unroll
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
        float ovr = inter / (iarea + jarea - inter);
        bool sup = ((double)ovr > thr) && (jcls == my_cls) && (col0 + j > row) && (col0 + j < n);
        word |= sup ? (1ull << j) : 0ull;
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
        // every lane evaluates (uniform control flow); cheap rejection by category / triangle first
        bool cand = (jcls == my_cls) && (col0 + j > row) && (row < n);
        float ovr = 0.f;
        if (cand) ovr = single_box_iou_rotated<64>(rbx, jb, S, lane);
        bool sup = cand && ((double)ovr >= thr);  // nms_rotated_cpu.cpp:54
        word |= sup ? (1ull << j) : 0ull;
      }
    }
  }
  if (row < n) mask[(long)row * wcap + w] = word;
}

// ---- step 5: greedy reduction ----------------------------------------------------------------
__device__ __forceinline__ u64 bcast64(u64 v, int lane) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
  return ((u64)hi << 32) | lo;
}

// One 512-thread workgroup (8 waves) per category segment.  The serial part -- resolving the
// 64x64 diagonal block with a scalar chain -- runs on wave 0 only; the other waves exist to
// keep the suppression rows of the NEXT blocks in flight: wave v owns rows 8v..8v+7 of every
// 64-row block, lane = bitmask word, and a 4-deep register ring prefetches those rows three
// blocks ahead so that the HBM/fabric latency (~1-2 us) never sits on the chain.  After the
// chain, every wave ORs the rows that were kept into the LDS `removed` bitset (ds_or_b64).
constexpr int RED_THREADS = 512;
constexpr int RED_ROWS = 64 / (RED_THREADS / 64);  // suppression rows per thread and block
constexpr int RED_DEPTH = 4;  // ring depth (blocks in flight, including the current one)

__global__ __launch_bounds__(RED_THREADS) void nms_reduce_kernel(const u64* __restrict__ mask,
                                                                 const uint32_t* __restrict__ cls_s, int n, int wcap,
                                                                 int max_per_class, const int* __restrict__ seg_start,
                                                                 int* counters, u64* keepbits) {
  extern __shared__ __attribute__((aligned(16))) u64 removed[];  // [wcap]
  __shared__ u64 kept_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
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
    __syncthreads();

    // rows of block b owned by this thread: b*64 + RED_ROWS*wid + r, word 1 + lane (first 64 later words)
    auto fetch_rows = [&](int b, u64 (&v)[RED_ROWS]) {
      const bool in = b <= b1 && (1 + lane) <= (b1 - b);
#pragma unroll
      for (int r = 0; r < RED_ROWS; r++) {
        const int row = b * 64 + RED_ROWS * wid + r;
        v[r] = (in && row >= s && row < e) ? mask[(long)row * wcap + 1 + lane] : 0ull;
      }
    };
    auto fetch_diag = [&](int b) -> u64 {
      const int row = b * 64 + lane;
      return (wid == 0 && b <= b1 && row >= s && row < e) ? mask[(long)row * wcap] : 0ull;
    };

    u64 ring[RED_DEPTH][RED_ROWS];
    u64 dring[RED_DEPTH];
#pragma unroll
    for (int d = 0; d < RED_DEPTH - 1; d++) { fetch_rows(b0 + d, ring[d]); dring[d] = fetch_diag(b0 + d); }

    for (int bq = b0; bq <= b1; bq += RED_DEPTH) {
#pragma unroll
      for (int u = 0; u < RED_DEPTH; u++) {
        const int b = bq + u;
        if (b > b1) break;  // uniform
        fetch_rows(b + RED_DEPTH - 1, ring[(u + RED_DEPTH - 1) % RED_DEPTH]);
        dring[(u + RED_DEPTH - 1) % RED_DEPTH] = fetch_diag(b + RED_DEPTH - 1);
        if (wid == 0) {
          const int row = b * 64 + lane;
          const bool valid = row >= s && row < e;
          const u64 D = dring[u];
          const u64 validmask = __ballot(valid);
          u64 rem = removed[b - b0] | ~validmask;
          rem = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rem >> 32)) << 32) |
              (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rem);  // provably uniform -> SALU chain
#pragma unroll
          for (int i = 0; i < 64; i++) {
            const u64 Di = bcast64(D, i);
            rem |= ((rem >> i) & 1ull) ? 0ull : Di;
          }
          const u64 kept = ~rem;
          if (lane == 0) {
            kept_s = kept;
            if (kept) atomicOr(&keepbits[b], kept);
          }
        }
        __syncthreads();
        const u64 kept = kept_s;
        const int nlater = b1 - b;
        {
          u64 acc = 0;
#pragma unroll
          for (int r = 0; r < RED_ROWS; r++) acc |= ((kept >> (RED_ROWS * wid + r)) & 1ull) ? ring[u][r] : 0ull;
          if (acc) atomicOr(&removed[b - b0 + 1 + lane], acc);  // acc != 0 implies 1 + lane <= nlater
        }
        // categories with more than 4096 boxes: remaining words, fetched after the chain
        for (int w = 65 + lane; w <= nlater; w += 64) {
          u64 acc = 0;
#pragma unroll
          for (int r = 0; r < RED_ROWS; r++)
            if ((kept >> (RED_ROWS * wid + r)) & 1ull) acc |= mask[((long)b * 64 + RED_ROWS * wid + r) * wcap + w];
          if (acc) atomicOr(&removed[b - b0 + w], acc);
        }
        __syncthreads();
      }
    }
  }
}

// ---- step 6: compaction ------------------------------------------------------------------------
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
  hipLaunchKernelGGL(nms_init_kernel, dim3(cdiv(N, T)), dim3(T), 0, s, w.iota, N, w.keepbits, nblocks, w.flag_r,
                     w.counters);
  D2_LAUNCH_OK();
  size_t tb = w.sort_temp_bytes;
  D2_HIP_OK(rocprim::radix_sort_pairs_desc(w.sort_temp, tb, scores, w.keys_out, w.iota, w.order, (unsigned)N, 0,
                                           32, s, false));
  const int* rankpos = nullptr;
  const uint32_t* cls_s = nullptr;
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
  dim3 mgrid(nblocks, wcap);
  if (rotated)
    hipLaunchKernelGGL((nms_mask_kernel<true>), mgrid, dim3(64), 0, s, w.boxes_s, cls_s, N, wcap, iou_threshold,
                       w.mask);
  else
    hipLaunchKernelGGL((nms_mask_kernel<false>), mgrid, dim3(64), 0, s, w.boxes_s, cls_s, N, wcap, iou_threshold,
                       w.mask);
  D2_LAUNCH_OK();
  const int rgrid = idxs ? 512 : 1;
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(rgrid), dim3(RED_THREADS), (size_t)wcap * 8, s, w.mask, cls_s, N, wcap, mpc,
                     w.seg_start, w.counters, w.keepbits);
  D2_LAUNCH_OK();
  hipLaunchKernelGGL(nms_scatter_flags_kernel, dim3(cdiv(N, T)), dim3(T), 0, s, w.keepbits, rankpos, N, w.flag_r);
  D2_LAUNCH_OK();
  hipLaunchKernelGGL(nms_compact_kernel, dim3(1), dim3(COMPACT_BLOCK), 0, s, w.flag_r, w.order, N, keep_out,
                     w.counters, result);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}
