// Segmented top-k by RADIX SELECT: the k best of every (image, feature level) segment without sorting the
// segment.  Serves the pre-NMS selection of the RPN (proposal_generator/proposal_utils.py:62-80: `logits_i.topk`)
// and of RetinaNet / dense detectors (meta_arch/dense_detector.py:207-223: score threshold, `nonzero`, `topk`).
// The first RPN path radix-SORTED all N x 268,569 (key, value) pairs (rocprim, 35-bit keys): 0.25 of its 0.44 ms;
// for RetinaNet's N x 16 M class scores a sort is out of the question.  Here:
//   1-3. three histogram passes over the scores (11 + 11 + 10 key bits; LDS-privatised histograms flushed with
//        atomics; the last workgroup of a segment to finish scans the bins and narrows the key prefix) find the
//        exact 32-bit key T of the k-th best candidate, the number c_lt of strictly better ones and how many
//        ties at T are needed;
//   4.   (only if more ties than needed) ties are counted per workgroup and prefix-summed, so that the ones with
//        the lowest element index are taken: the selection is deterministic;
//   5.   one compaction pass writes the <= k selected (key, index) pairs;
//   6.   one workgroup per segment orders them in LDS (bitonic, 64-bit keys = score key : index).
// Every pass reads 4 B per element (HBM bound); nothing synchronises with the host.  Launches whose workgroups are
// all resident at once (the RPN: 138) run steps 1-5 as ONE kernel (tk_fused_kernel): the chunk stays in registers and
// the workgroups of a segment meet at barriers built on device-scope atomics -- without device-scope fences, which
// write the XCD's L2 back on this part.
#include <cmath>

#include "topk.h"

namespace d2amd {

constexpr int TK_THREADS = 256;
constexpr int TK_ITEMS = 16;
constexpr int TK_CHUNK = TK_THREADS * TK_ITEMS;  // elements per workgroup
constexpr int TK_BINS = 2048;

struct SegState {
  uint32_t prefix;  // key bits resolved so far
  int k_rem;        // still to take from the current bucket
  int c_lt;         // candidates strictly better than the current bucket
  int total;        // candidates of the segment
  int take_all;     // fewer candidates than k: all of them are selected
  int ties_total, need;
  int done[4];      // workgroups finished per pass (0-2: histogram passes, 3: tie count)
  int cnt_lt, cnt_tie;
  int cnt;          // selected = min(k, total)
  int c_def;        // candidates better than the bucket of pass 1 (key >> 10 < prefix): placed by a scan, see blk_def
  int pad[1];
};
static_assert(sizeof(SegState) == 64, "SegState layout");

struct TkParams {
  TopkInput in;
  int use_thr;
  float xmin;      // use_thr: an element is a candidate iff x >= xmin (NaN xmin: none is)
  int maxblk;      // workgroups per segment in the grid
  int tickets;     // 1: the last workgroup of a segment (atomic ticket) scans; 0: separate scan launches
  int reps;        // consecutive TK_CHUNK chunks per workgroup (keeps the workgroups of a segment <= ~256: every
                   // workgroup takes a ticket on ONE address per pass, and 3,000 returning atomics there cost 0.3 ms)
  int* blk_def;    // [segments][maxblk], large segments only (else null).  Pass 2 counts, per workgroup, the candidates
                   // that are selected whatever the last 10 key bits decide (key >> 10 < the 21-bit prefix); the scan
                   // launch turns the counts into offsets and the compaction places those candidates WITHOUT
                   // reserving through a returning atomic (thousands of them on one counter serialise).  Segments
                   // that select everything skip pass 2 and keep the counter.
};

// The key is the order-preserving image of the STORED value.  Dense detectors store class logits and the reference
// ranks sigmoid(logit): sigmoid is monotone, so ranking the logit gives the same order wherever the fp32 scores differ
// and a defined one (higher logit, then lower index) inside a group of equal fp32 scores -- independent of any exp()
// implementation.  The score threshold arrives as the equivalent bound on the stored value (logit_lower_bound()).
__device__ __forceinline__ bool tk_key(const TkParams& P, float x, uint32_t& key) {
  if (P.use_thr && !(x >= P.xmin)) return false;
  key = topk_desc_key(x);
  return true;
}

// exclusive prefix of one int per thread over the 256-thread workgroup (wave shuffles + 4 wave totals through LDS);
// a serial scan by one thread costs ~130 cycles of dependent LDS latency per element: 14 us for 256
__device__ __forceinline__ int tk_block_excl_scan(int v, int* lds4, int& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  __syncthreads();  // lds4 may still be read from a previous call
  if (lane == 63) lds4[w] = x;
  __syncthreads();
  int base = 0;
  total = 0;
#pragma unroll
  for (int i = 0; i < TK_THREADS / 64; i++) {
    const int t = lds4[i];
    if (i < w) base += t;
    total += t;
  }
  return base + x - v;
}

// the TK_ITEMS values of this thread, all loads in flight together (out of range: a value no test accepts)
__device__ __forceinline__ void tk_load(const float* __restrict__ x, long base, int size, float (&v)[TK_ITEMS], bool (&ok)[TK_ITEMS]) {
#pragma unroll
  for (int j = 0; j < TK_ITEMS; j++) {
    const long i = base + (long)j * TK_THREADS + threadIdx.x;
    ok[j] = i < size;
    v[j] = x[ok[j] ? i : (long)size - 1];
  }
}

template <typename TT>
__device__ __forceinline__ TT ld_agent(const TT* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// scan of a segment's bins (ascending key = best first) by one 256-thread workgroup: narrows the key prefix
// (-> true: pass 0 found fewer candidates than k, all of them are selected)
template <int PASS>
__device__ __forceinline__ bool tk_scan_bins(const TkParams& P, SegState* S, const int* gh, int l) {
  const int tid = threadIdx.x;
  const uint32_t prefix = PASS > 0 ? S->prefix : 0u;
  constexpr int BINS = PASS == 2 ? 1024 : 2048, PER = BINS / TK_THREADS;
  __shared__ int lds4[TK_THREADS / 64];
  __shared__ int s_bin, s_before;
  int loc[PER], sum = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) { loc[j] = ld_agent(&gh[tid * PER + j]); sum += loc[j]; }
  if (tid == 0) { s_bin = -1; s_before = 0; }
  int s_total;
  int run = tk_block_excl_scan(sum, lds4, s_total);
  int k_rem = PASS == 0 ? P.in.k[l] : S->k_rem;
  if (PASS == 0 && s_total < k_rem) {  // not enough candidates: everything is selected (uniform)
    if (tid == 0) { S->total = s_total; S->take_all = 1; S->cnt = s_total; }
    return true;
  }
#pragma unroll
  for (int j = 0; j < PER; j++) {
    if (run < k_rem && run + loc[j] >= k_rem) { s_bin = tid * PER + j; s_before = run; }
    run += loc[j];
  }
  __syncthreads();
  if (tid == 0) {
    const int b = s_bin;  // exists: k_rem >= 1 and the bins hold >= k_rem candidates
    if (PASS == 0) { S->total = s_total; S->cnt = k_rem; S->prefix = (uint32_t)b; }
    else if (PASS == 1) S->prefix = (prefix << 11) | (uint32_t)b;
    else S->prefix = (prefix << 10) | (uint32_t)b;
    if (PASS == 2) S->c_def = S->c_lt;
    S->c_lt = (PASS == 0 ? 0 : S->c_lt) + s_before;
    S->k_rem = k_rem - s_before;
    if (PASS == 2) { S->ties_total = ld_agent(&gh[b]); S->need = k_rem - s_before; }
  }
  return false;
}

// PASS 0: bins = key >> 21 of all candidates; 1: (key >> 10) & 2047 where key >> 21 == prefix; 2: key & 1023 where
// key >> 10 == prefix.
template <int PASS>
__global__ __launch_bounds__(TK_THREADS) void tk_hist_kernel(TkParams P, SegState* __restrict__ st, int* __restrict__ hist) {
  const int seg = blockIdx.y, l = seg % P.in.L, img = seg / P.in.L;
  const int size = P.in.size[l];
  const long span = (long)TK_CHUNK * P.reps;
  const long base0 = (long)blockIdx.x * span;
  if (base0 >= size) return;
  const int nblk = (int)((size + span - 1) / span);
  SegState* S = st + seg;
  const int tid = threadIdx.x;
  if (PASS > 0 && S->take_all) return;  // written by the previous launch
  __shared__ int h[TK_BINS];
  __shared__ int s_last, s_def;
  for (int i = tid; i < TK_BINS; i += TK_THREADS) h[i] = 0;
  if (tid == 0) s_def = 0;
  __syncthreads();
  int n_def = 0;
  const uint32_t prefix = PASS > 0 ? S->prefix : 0u;
  const float* x = P.in.ptr[l] + (long)img * P.in.stride[l];
  for (int rep = 0; rep < P.reps; rep++) {
    const long base = base0 + (long)rep * TK_CHUNK;
    if (base >= size) break;
    float v[TK_ITEMS];
    bool ok[TK_ITEMS];
    tk_load(x, base, size, v, ok);
#pragma unroll
    for (int j = 0; j < TK_ITEMS; j++) {
      uint32_t key;
      if (!ok[j] || !tk_key(P, v[j], key)) continue;
      if (PASS == 0) atomicAdd(&h[key >> 21], 1);
      else if (PASS == 1) { if ((key >> 21) == prefix) atomicAdd(&h[(key >> 10) & 2047u], 1); }
      else {
        if ((key >> 10) == prefix) atomicAdd(&h[key & 1023u], 1);
        n_def += (key >> 10) < prefix ? 1 : 0;
      }
    }
  }
  if (PASS == 2 && P.blk_def != nullptr) {  // uniform
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) n_def += __shfl_xor(n_def, d, 64);
    if ((tid & 63) == 0 && n_def) atomicAdd(&s_def, n_def);
  }
  __syncthreads();
  if (PASS == 2 && P.blk_def != nullptr && tid == 0) P.blk_def[(long)seg * P.maxblk + blockIdx.x] = s_def;
  int* gh = hist + ((long)seg * 3 + PASS) * TK_BINS;
  for (int i = tid; i < TK_BINS; i += TK_THREADS)
    if (h[i]) atomicAdd(&gh[i], h[i]);
  if (!P.tickets) return;  // tk_scan_kernel follows
  // (no device-scope fence: bins and ticket are device-scope atomics, performed at the memory side of the L2s, and
  // the scan reads the bins with device-scope loads -- see tk_segment_barrier; a fence costs an L2 write-back here)
  __builtin_amdgcn_s_waitcnt(0);  // this wave's bin atomics have been acknowledged
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(&S->done[PASS], 1) == nblk - 1;
  __syncthreads();
  if (!s_last) return;
  tk_scan_bins<PASS>(P, S, gh, l);
}

// exclusive prefix of the per-workgroup tie counts of one segment (one 256-thread workgroup; thread t owns a run)
__device__ __forceinline__ void tk_ties_scan(int* bt, int nblk) {
  const int tid = threadIdx.x;
  const int per = (nblk + TK_THREADS - 1) / TK_THREADS;
  const int lo = min(tid * per, nblk), hi = min(lo + per, nblk);
  int sum = 0;
  for (int i = lo; i < hi; i++) sum += ld_agent(&bt[i]);
  __shared__ int lds4[TK_THREADS / 64];
  int tot;
  int run = tk_block_excl_scan(sum, lds4, tot);
  for (int i = lo; i < hi; i++) { const int v = ld_agent(&bt[i]); bt[i] = run; run += v; }
}

// separate launch of the bin scan (large segments: thousands of tickets on one address would serialise in L2)
template <int PASS>
__global__ __launch_bounds__(TK_THREADS) void tk_scan_kernel(TkParams P, SegState* __restrict__ st, int* __restrict__ hist) {
  const int seg = blockIdx.x, l = seg % P.in.L;
  SegState* S = st + seg;
  if (P.in.size[l] == 0 || (PASS > 0 && S->take_all)) return;
  (void)tk_scan_bins<PASS>(P, S, hist + ((long)seg * 3 + PASS) * TK_BINS, l);
  if (PASS == 2 && P.blk_def != nullptr) {  // pass 2's per-workgroup counts -> offsets (TkParams::blk_def); uniform
    const long span = (long)TK_CHUNK * P.reps;
    __syncthreads();
    tk_ties_scan(P.blk_def + (long)seg * P.maxblk, (int)((P.in.size[l] + span - 1) / span));
  }
}

// per workgroup: number of ties (key == T); last workgroup: exclusive prefix over the workgroups of the segment
__global__ __launch_bounds__(TK_THREADS) void tk_ties_kernel(TkParams P, SegState* __restrict__ st, int* __restrict__ blk_ties) {
  const int seg = blockIdx.y, l = seg % P.in.L, img = seg / P.in.L;
  const int size = P.in.size[l];
  const long span = (long)TK_CHUNK * P.reps;
  const long base0 = (long)blockIdx.x * span;
  if (base0 >= size) return;
  const int nblk = (int)((size + span - 1) / span);
  SegState* S = st + seg;
  if (S->take_all || S->ties_total <= S->need) return;  // all ties are taken: no order needed
  const int tid = threadIdx.x;
  const uint32_t T = S->prefix;
  const float* x = P.in.ptr[l] + (long)img * P.in.stride[l];
  int c = 0;
  for (int rep = 0; rep < P.reps; rep++) {
    const long base = base0 + (long)rep * TK_CHUNK;
    if (base >= size) break;
    float v[TK_ITEMS];
    bool ok[TK_ITEMS];
    tk_load(x, base, size, v, ok);
#pragma unroll
    for (int j = 0; j < TK_ITEMS; j++) {
      uint32_t key;
      if (ok[j] && tk_key(P, v[j], key) && key == T) c++;
    }
  }
  __shared__ int red[TK_THREADS];
  __shared__ int s_last;
  red[tid] = c;
  __syncthreads();
  for (int s2 = TK_THREADS / 2; s2 > 0; s2 >>= 1) {
    if (tid < s2) red[tid] += red[tid + s2];
    __syncthreads();
  }
  int* bt = blk_ties + (long)seg * P.maxblk;
  if (tid == 0) {
    __hip_atomic_store(&bt[blockIdx.x], red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (P.tickets) {
      __builtin_amdgcn_s_waitcnt(0);  // the count above is a device-scope atomic store: acknowledged = visible
      s_last = atomicAdd(&S->done[3], 1) == nblk - 1;
    } else {
      s_last = 0;  // tk_ties_scan_kernel follows
    }
  }
  __syncthreads();
  if (!s_last) return;
  tk_ties_scan(bt, nblk);
}

__global__ __launch_bounds__(TK_THREADS) void tk_ties_scan_kernel(TkParams P, SegState* __restrict__ st, int* __restrict__ blk_ties) {
  const int seg = blockIdx.x, l = seg % P.in.L;
  const SegState* S = st + seg;
  const int size = P.in.size[l];
  if (size == 0 || S->take_all || S->ties_total <= S->need) return;
  const long span = (long)TK_CHUNK * P.reps;
  tk_ties_scan(blk_ties + (long)seg * P.maxblk, (int)((size + span - 1) / span));
}

// what the three histogram passes found for a segment
struct TkSel {
  uint32_t T;    // key of the k-th best candidate
  int c_lt;      // candidates strictly better than T
  int need;      // ties at T to take
  bool take_all; // fewer candidates than k: all of them
  bool ordered;  // more ties than needed: the ones with the lowest element index
};
struct TkCompactLds {
  int wcnt[TK_ITEMS][TK_THREADS / 64];
  int lds4[TK_THREADS / 64];
  int base_lt, base_tie;
};

// compaction of one TK_CHUNK of a segment (values already loaded): writes its selected (key : index) pairs.
// `before` = ties in the chunks before this one (ordered mode), advanced by this chunk's ties.
__device__ __forceinline__ void tk_compact_chunk(const TkParams& P, const TkSel& Z, SegState* S,
                                                 const float (&v)[TK_ITEMS], const bool (&ok)[TK_ITEMS], long base,
                                                 unsigned long long* __restrict__ out, int& before, TkCompactLds& L) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t keys[TK_ITEMS];
  unsigned lt_bits = 0, tie_bits = 0;  // bit j: element j of this thread is strictly better / a tie
  unsigned long long bal[TK_ITEMS];
#pragma unroll
  for (int j = 0; j < TK_ITEMS; j++) {
    uint32_t key = 0;
    const bool c = ok[j] && tk_key(P, v[j], key);
    keys[j] = key;
    if (c && (Z.take_all || key < Z.T)) lt_bits |= 1u << j;
    const bool tie = c && !Z.take_all && key == Z.T;
    if (tie) tie_bits |= 1u << j;
    if (Z.ordered) {  // uniform
      bal[j] = __ballot(tie);
      if (lane == 0) L.wcnt[j][wave] = __builtin_popcountll(bal[j]);
    }
  }
  if (!__syncthreads_or((lt_bits | tie_bits) != 0u)) return;  // uniform: nothing selected in this chunk (the usual case)
  // one atomic per workgroup and counter (2,000 returning atomics on ONE address serialise in L2: 20 us)
  int tot_lt, tot_tie = 0;
  const int my_lt = tk_block_excl_scan(__builtin_popcount(lt_bits), L.lds4, tot_lt);
  int my_tie = 0;
  if (!Z.ordered) my_tie = tk_block_excl_scan(__builtin_popcount(tie_bits), L.lds4, tot_tie);
  if (tid == 0) {
    L.base_lt = tot_lt ? atomicAdd(&S->cnt_lt, tot_lt) : 0;
    L.base_tie = tot_tie ? atomicAdd(&S->cnt_tie, tot_tie) : 0;
  }
  __syncthreads();
  {
    int p_lt = L.base_lt + my_lt, p_tie = Z.c_lt + L.base_tie + my_tie;
#pragma unroll
    for (int j = 0; j < TK_ITEMS; j++) {
      const long i = base + (long)j * TK_THREADS + tid;
      const unsigned long long e = ((unsigned long long)keys[j] << 32) | (uint32_t)i;
      if (lt_bits & (1u << j)) out[p_lt++] = e;
      else if (!Z.ordered && (tie_bits & (1u << j))) out[p_tie++] = e;
    }
  }
  if (!Z.ordered) return;
  // rank of a tie in element-index order: rows j ascending, inside a row waves then lanes ascending
  // (wcnt is complete: the scans above contain workgroup barriers)
#pragma unroll
  for (int j = 0; j < TK_ITEMS; j++) {
    int row_before = 0, row_total = 0;
#pragma unroll
    for (int w = 0; w < TK_THREADS / 64; w++) {
      const int c = L.wcnt[j][w];
      if (w < wave) row_before += c;
      row_total += c;
    }
    if (tie_bits & (1u << j)) {
      const int rank = before + row_before + __builtin_popcountll(bal[j] & ((1ull << lane) - 1ull));
      if (rank < Z.need) {
        const long i = base + (long)j * TK_THREADS + tid;
        out[Z.c_lt + rank] = ((unsigned long long)keys[j] << 32) | (uint32_t)i;
      }
    }
    before += row_total;
  }
}

// The usual (not tie-ordered) case for one chunk whose values are already loaded (`lim`: elements of the segment from
// `base` on; >= TK_CHUNK for an inner chunk).  A selected candidate is one of
//   definite   (scan_mode) key >> 10 below the 21-bit prefix: placed at the workgroup's scanned offset + rank, no
//              atomic (TkParams::blk_def);
//   late       better than T inside the last bucket (scan_mode), or any better-than-T candidate (otherwise): a
//              reservation on the segment's counter -- rare in scan_mode;
//   tie        key == T: all taken here (not ordered), behind the c_lt better ones.
__device__ __forceinline__ void tk_compact_cls(const TkParams& P, const TkSel& Z, SegState* S,
                                               const float (&v)[TK_ITEMS], long base, long lim,
                                               unsigned long long* __restrict__ out, TkCompactLds& L, bool scan_mode,
                                               int c_def, int& def_pos) {
  const int tid = threadIdx.x;
  unsigned df = 0u, lt = 0u, tie = 0u;
  const uint32_t pre = Z.T >> 10;
#pragma unroll
  for (int j = 0; j < TK_ITEMS; j++) {
    const uint32_t key = topk_desc_key(v[j]);
    const bool ok = j * TK_THREADS + tid < lim && (!P.use_thr || v[j] >= P.xmin);  // (tk_key)
    const bool sel = ok && (Z.take_all || key < Z.T);
    const bool is_def = sel && scan_mode && (key >> 10) < pre;
    df |= (unsigned)is_def << j;
    lt |= (unsigned)(sel && !is_def) << j;
    tie |= (unsigned)(ok && !sel && key == Z.T) << j;
  }
  const int n_def = __builtin_popcount(df);
  const int n_rare = __builtin_popcount(lt) + (__builtin_popcount(tie) << 16);  // (<= 4,096 each per workgroup)
  if (!__syncthreads_or((n_def | n_rare) != 0)) return;  // uniform
  int tot_def = 0, my_def = 0, tot_rare = 0, my_rare = 0;
  if (scan_mode) my_def = tk_block_excl_scan(n_def, L.lds4, tot_def);  // uniform
  const bool rare = __syncthreads_or(n_rare != 0);  // uniform
  if (rare) {
    my_rare = tk_block_excl_scan(n_rare, L.lds4, tot_rare);
    if (tid == 0) {
      L.base_lt = (tot_rare & 0xffff) ? atomicAdd(&S->cnt_lt, tot_rare & 0xffff) : 0;
      L.base_tie = (tot_rare >> 16) ? atomicAdd(&S->cnt_tie, tot_rare >> 16) : 0;
    }
    __syncthreads();
  }
  int p_def = def_pos + my_def;
  int p_late = rare ? c_def + L.base_lt + (my_rare & 0xffff) : 0;
  int p_tie = rare ? Z.c_lt + L.base_tie + (my_rare >> 16) : 0;
  def_pos += tot_def;
  if ((df | lt | tie) == 0u) return;  // (a thread selects ~1 % of its values)
#pragma unroll
  for (int j = 0; j < TK_ITEMS; j++) {
    const unsigned long long e = ((unsigned long long)topk_desc_key(v[j]) << 32) | (uint32_t)(base + j * TK_THREADS + tid);
    if (df & (1u << j)) out[p_def++] = e;
    else if (lt & (1u << j)) out[p_late++] = e;
    else if (tie & (1u << j)) out[p_tie++] = e;
  }
}

__global__ __launch_bounds__(TK_THREADS) void tk_compact_kernel(TkParams P, SegState* __restrict__ st,
                                                               const int* __restrict__ blk_ties,
                                                               unsigned long long* __restrict__ cand, int kmax) {
  const int seg = blockIdx.y, l = seg % P.in.L, img = seg / P.in.L;
  const int size = P.in.size[l];
  const long span = (long)TK_CHUNK * P.reps;
  const long base0 = (long)blockIdx.x * span;
  if (base0 >= size) return;
  SegState* S = st + seg;
  TkSel Z;
  Z.take_all = S->take_all != 0;
  Z.T = S->prefix;
  Z.c_lt = S->c_lt; Z.need = S->need;
  Z.ordered = !Z.take_all && S->ties_total > Z.need;  // uniform
  const float* x = P.in.ptr[l] + (long)img * P.in.stride[l];
  unsigned long long* out = cand + (long)seg * kmax;
  __shared__ TkCompactLds L;
  if (!Z.ordered) {  // uniform
    const bool scan_mode = P.blk_def != nullptr && !Z.take_all;  // (pass 2 does not run for a take-all segment)
    int def_pos = scan_mode ? P.blk_def[(long)seg * P.maxblk + blockIdx.x] : 0;
    const int c_def = scan_mode ? S->c_def : 0;
    // the next chunk's 16 loads are issued before this chunk's barriers.  (Measured on RetinaNet's 2 x 16.1M logits:
    // 90 us with a reservation per chunk, 55-69 us for every variant since -- one reservation per 4 chunks, none at
    // all, with and without this prefetch -- against 25-29 us for the histogram passes over the same data, which have
    // no barrier in their chunk loop.  What bounds this kernel is not identified yet: DESIGN.md 3.7b.)
    const int tid = threadIdx.x;
    auto load = [&](long base, float (&v)[TK_ITEMS]) {
      const long last = (long)size - 1 - base;  // >= 0
#pragma unroll
      for (int j = 0; j < TK_ITEMS; j++) v[j] = x[base + min((long)(j * TK_THREADS + tid), last)];
    };
    float cur[TK_ITEMS], nxt[TK_ITEMS];
    load(base0, cur);
    for (int rep = 0; rep < P.reps; rep++) {
      const long base = base0 + (long)rep * TK_CHUNK;
      if (base >= size) break;  // uniform
      const bool more = rep + 1 < P.reps && base + TK_CHUNK < size;  // uniform
      if (more) load(base + TK_CHUNK, nxt);
      if (rep) __syncthreads();  // the previous chunk's readers of L are done
      tk_compact_cls(P, Z, S, cur, base, (long)size - base, out, L, scan_mode, c_def, def_pos);
      if (!more) break;
#pragma unroll
      for (int j = 0; j < TK_ITEMS; j++) cur[j] = nxt[j];
    }
    return;
  }
  int before = blk_ties[(long)seg * P.maxblk + blockIdx.x];  // ties in the workgroups before this one
  for (int rep = 0; rep < P.reps; rep++) {
    const long base = base0 + (long)rep * TK_CHUNK;
    if (base >= size) break;  // uniform
    __syncthreads();          // the previous chunk's readers of L are done
    float v[TK_ITEMS];
    bool ok[TK_ITEMS];
    tk_load(x, base, size, v, ok);
    tk_compact_chunk(P, Z, S, v, ok, base, out, before, L);
  }
}

// ---- all of the above in ONE launch, for launches whose workgroups are co-resident (the RPN: 138 workgroups) ----
// Five dependent launches of ~140 workgroups spend most of their 18-20 us each on what surrounds the arithmetic:
// launch, the first load round trip, the ticket, the last workgroup's scan, the next launch.  Here every workgroup
// loads its chunk ONCE (16 values per thread stay in registers through all passes), and the workgroups of a segment
// meet at a segment-wide barrier (a counter in memory) after each histogram flush; after the barrier EVERY workgroup
// scans the segment's bins itself -- same inputs, same result -- so nothing is broadcast and no SegState travels
// between passes.  Precondition (host): reps == 1 and the whole grid fits on the device at once, so that spinning
// workgroups cannot starve the ones they wait for.
struct TkScanOut { int bin, before, total; };

template <int BINS>
__device__ __forceinline__ TkScanOut tk_scan_local(const int* gh, int k_rem, int* lds4, int* s_pair) {
  const int tid = threadIdx.x;
  constexpr int PER = BINS / TK_THREADS;
  int loc[PER], sum = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) { loc[j] = ld_agent(&gh[tid * PER + j]); sum += loc[j]; }
  int total;
  int run = tk_block_excl_scan(sum, lds4, total);
  if (tid == 0) { s_pair[0] = -1; s_pair[1] = 0; }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PER; j++) {
    if (run < k_rem && run + loc[j] >= k_rem) { s_pair[0] = tid * PER + j; s_pair[1] = run; }
    run += loc[j];
  }
  __syncthreads();
  TkScanOut o;
  o.bin = s_pair[0]; o.before = s_pair[1]; o.total = total;
  __syncthreads();  // s_pair is reused by the next scan
  return o;
}

// All workgroups of the segment have arrived.  Everything the workgroups exchange (histogram bins, tie counts, the
// counter itself) is written and read with DEVICE-SCOPE ATOMICS, which are performed at the memory side of the XCDs'
// L2s: no release / acquire fence is needed for them, and none is used -- on this part a device-scope fence writes the
// XCD's L2 back and invalidates it (the eight L2s are not coherent with each other), ~10 us per fence with 140
// workgroups doing it at once, which is what made the first fused version slower than five launches.  What is needed
// is that this workgroup's atomics have completed before its arrival is counted: s_waitcnt(0) in every wave, then the
// workgroup barrier, then the arrival.
__device__ __forceinline__ void tk_segment_barrier(int* counter, int nblk) {
  __builtin_amdgcn_s_waitcnt(0);  // every outstanding memory operation of this wave has been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblk) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
}

__global__ __launch_bounds__(TK_THREADS) void tk_fused_kernel(TkParams P, SegState* __restrict__ st, int* __restrict__ hist,
                                                             int* __restrict__ blk_ties,
                                                             unsigned long long* __restrict__ cand, int kmax) {
  const int seg = blockIdx.y, l = seg % P.in.L, img = seg / P.in.L;
  const int size = P.in.size[l];
  const long base = (long)blockIdx.x * TK_CHUNK;
  if (base >= size) return;
  const int nblk = (size + TK_CHUNK - 1) / TK_CHUNK;
  SegState* S = st + seg;
  const int tid = threadIdx.x;
  __shared__ int h[TK_BINS];
  __shared__ int lds4[TK_THREADS / 64];
  __shared__ int s_pair[2];
  __shared__ TkCompactLds L;
  const float* x = P.in.ptr[l] + (long)img * P.in.stride[l];
  float v[TK_ITEMS];
  bool ok[TK_ITEMS];
  tk_load(x, base, size, v, ok);
  uint32_t keys[TK_ITEMS];
  unsigned cand_bits = 0;
#pragma unroll
  for (int j = 0; j < TK_ITEMS; j++) {
    keys[j] = 0;
    if (ok[j] && tk_key(P, v[j], keys[j])) cand_bits |= 1u << j;
  }
  int* gh = hist + (long)seg * 3 * TK_BINS;
  TkSel Z{};
  int k_rem = P.in.k[l];
  uint32_t prefix = 0;
  // ---- pass 0: key >> 21 ------------------------------------------------------------------------
  for (int i = tid; i < TK_BINS; i += TK_THREADS) h[i] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TK_ITEMS; j++)
    if (cand_bits & (1u << j)) atomicAdd(&h[keys[j] >> 21], 1);
  __syncthreads();
  for (int i = tid; i < TK_BINS; i += TK_THREADS)
    if (h[i]) atomicAdd(&gh[i], h[i]);
  tk_segment_barrier(&S->done[0], nblk);
  TkScanOut o = tk_scan_local<2048>(gh, k_rem, lds4, s_pair);
  const int total = o.total;
  Z.take_all = total < k_rem;  // not enough candidates: everything is selected (uniform)
  if (!Z.take_all) {
    prefix = (uint32_t)o.bin; Z.c_lt = o.before; k_rem -= o.before;
    // ---- pass 1: (key >> 10) & 2047 where key >> 21 == prefix -----------------------------------
    for (int i = tid; i < TK_BINS; i += TK_THREADS) h[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TK_ITEMS; j++)
      if ((cand_bits & (1u << j)) && (keys[j] >> 21) == prefix) atomicAdd(&h[(keys[j] >> 10) & 2047u], 1);
    __syncthreads();
    for (int i = tid; i < TK_BINS; i += TK_THREADS)
      if (h[i]) atomicAdd(&gh[TK_BINS + i], h[i]);
    tk_segment_barrier(&S->done[1], nblk);
    o = tk_scan_local<2048>(gh + TK_BINS, k_rem, lds4, s_pair);
    prefix = (prefix << 11) | (uint32_t)o.bin; Z.c_lt += o.before; k_rem -= o.before;
    // ---- pass 2: key & 1023 where key >> 10 == prefix -------------------------------------------
    for (int i = tid; i < 1024; i += TK_THREADS) h[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TK_ITEMS; j++)
      if ((cand_bits & (1u << j)) && (keys[j] >> 10) == prefix) atomicAdd(&h[keys[j] & 1023u], 1);
    __syncthreads();
    for (int i = tid; i < 1024; i += TK_THREADS)
      if (h[i]) atomicAdd(&gh[2 * TK_BINS + i], h[i]);
    tk_segment_barrier(&S->done[2], nblk);
    o = tk_scan_local<1024>(gh + 2 * TK_BINS, k_rem, lds4, s_pair);
    prefix = (prefix << 10) | (uint32_t)o.bin; Z.c_lt += o.before;
    Z.need = k_rem - o.before;
    Z.T = prefix;
    const int ties_total = ld_agent(&gh[2 * TK_BINS + o.bin]);
    Z.ordered = ties_total > Z.need;
  }
  if (blockIdx.x == 0 && tid == 0) S->cnt = Z.take_all ? total : P.in.k[l];  // read by tk_sort_kernel
  // ---- ties in element-index order: the ties of the workgroups before this one ---------------------------------
  int before = 0;
  if (Z.ordered) {  // uniform over the segment
    int c = 0;
#pragma unroll
    for (int j = 0; j < TK_ITEMS; j++) c += ((cand_bits & (1u << j)) && keys[j] == Z.T) ? 1 : 0;
    int tot;
    (void)tk_block_excl_scan(c, lds4, tot);
    int* bt = blk_ties + (long)seg * P.maxblk;
    if (tid == 0) __hip_atomic_store(&bt[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tk_segment_barrier(&S->done[3], nblk);
    int mine = 0;
    for (int j = tid; j < (int)blockIdx.x; j += TK_THREADS) mine += ld_agent(&bt[j]);
    (void)tk_block_excl_scan(mine, lds4, before);
    __syncthreads();
  }
  tk_compact_chunk(P, Z, S, v, ok, base, cand + (long)seg * kmax, before, L);
}

// Final ordering of the <= k selected (key : index) pairs of a segment.  One workgroup bitonic-sorts a RUN of up to
// TK_RUN pairs in LDS (<= 128 KB); a segment with more (RetinaNet with TOPK_CANDIDATES_TEST 20000: BASELINE configs[3])
// is cut into runs, every run is sorted by its own workgroup and written back, and tk_merge_kernel places every
// element at  rank = position in its run + sum over the other runs of #elements below it  (binary searches; the
// 64-bit keys are unique, so the ranks are a permutation).
constexpr int TK_RUN_MAX = 16384;  // one run in LDS: 128 KB
// run length of a launch whose largest segment selects kmax pairs: one run while it fits, else 4,096 -- a 16,384-pair
// bitonic sort walks 16 pairs per thread through 105 steps (250 us for RetinaNet's 20,000 per level, r02 profile);
// five 4,096 runs sorted by five workgroups + the rank merge take a quarter of that
static inline int tk_run_for(int kmax) { return kmax <= 4096 ? TK_RUN_MAX : 4096; }  // (<= 4,096: one run)

__global__ __launch_bounds__(1024) void tk_sort_kernel(TkParams P, const SegState* __restrict__ st,
                                                      unsigned long long* __restrict__ cand, int kmax, int TK_RUN,
                                                      uint32_t* __restrict__ sel, int* __restrict__ cnt_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  const int seg = blockIdx.x, l = seg % P.in.L, img = seg / P.in.L;
  const int tid = threadIdx.x;
  const int total = st[seg].cnt;
  if (tid == 0 && blockIdx.y == 0) cnt_out[seg] = total;
  const int lo = blockIdx.y * TK_RUN;
  if (lo >= total && blockIdx.y > 0) return;
  const int n = min(total - lo, TK_RUN);
  int p2 = 1;
  while (p2 < n) p2 <<= 1;  // uniform
  unsigned long long* in = cand + (long)seg * kmax + lo;
  for (int i = tid; i < p2; i += 1024) sk[i] = i < n ? in[i] : ~0ull;
  __syncthreads();
  // One compare-exchange PAIR per thread and step (pair q of step j: i = q with a zero bit inserted at log2(j),
  // partner i | j) -- every thread works in every step; indexing by element left half of them idle and cost two
  // dependent LDS round trips per step.  The 64 pairs of a wave cover one aligned block of 128 elements for every
  // j <= 64, so those steps only need the wave's own LDS ordering; the workgroup barrier is needed around the steps
  // with j >= 128 (14 of the 66 steps of 2,048).
  const int half = p2 >> 1;
  for (int k2 = 2; k2 <= p2; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int q = tid; q < half; q += 1024) {
        const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1)), ix = i | j;
        const unsigned long long a = sk[i], b = sk[ix];
        const bool up = (i & k2) == 0;
        if ((a > b) == up) { sk[i] = b; sk[ix] = a; }
      }
      if (j >= 128 || (j == 1 && k2 >= 128)) __syncthreads();  // uniform
      else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
  __syncthreads();  // (short runs end on wave-ordered steps; the copy-out below reads across waves)
  if (total <= TK_RUN) {  // single run: done
    uint32_t* o = sel + (long)img * P.in.koff[P.in.L] + P.in.koff[l];
    for (int i = tid; i < n; i += 1024) o[i] = (uint32_t)sk[i];
  } else {
    for (int i = tid; i < n; i += 1024) in[i] = sk[i];  // the sorted run, in place (this workgroup owns the range)
  }
}

// Segments of <= TK_RANK_MAX selected pairs (the RPN's 2,000 per level, RetinaNet's 1,000): RANK instead of sort.
// The 64-bit keys are unique, so  rank = #keys below mine  is a permutation.  A workgroup stages the segment's keys in
// LDS and ranks 64 of them: lane = key, the four waves each count over a quarter of the segment (two keys per 16-B
// LDS broadcast read) and add their partial counts in LDS.  320 workgroups for 10 segments of 2,000 instead of the 10
// of the bitonic sort, and no barrier chain: 66 dependent steps there, one pass here.
constexpr int TK_RANK_MAX = 2048;
#ifndef D2AMD_TK_RANK_THREADS
#define D2AMD_TK_RANK_THREADS 1024
#endif
constexpr int TK_RANK_THREADS = D2AMD_TK_RANK_THREADS, TK_RANK_WAVES = TK_RANK_THREADS / 64;
template <bool RPN>
__global__ __launch_bounds__(TK_RANK_THREADS) void tk_rank_kernel(TkParams P, const SegState* __restrict__ st,
                                                     const unsigned long long* __restrict__ cand, int kmax,
                                                     uint32_t* __restrict__ sel, int* __restrict__ cnt_out,
                                                     const TopkRpnEpilogue E) {
  __shared__ __attribute__((aligned(16))) unsigned long long sk[TK_RANK_MAX];
  __shared__ int rk[64];
  const int seg = blockIdx.x, l = seg % P.in.L, img = seg / P.in.L, tid = threadIdx.x;
  const int n = min(st[seg].cnt, TK_RANK_MAX);
  if (tid == 0 && blockIdx.y == 0) cnt_out[seg] = n;
  const int base = blockIdx.y * 64;
  if (base >= n) return;  // uniform
  const unsigned long long* in = cand + (long)seg * kmax;
  // a wave's share is a whole number of key pairs; the padding ranks above every key
  const int np = (n + 2 * TK_RANK_WAVES - 1) / (2 * TK_RANK_WAVES) * (2 * TK_RANK_WAVES);
  for (int i = tid; i < np; i += TK_RANK_THREADS) sk[i] = i < n ? in[i] : ~0ull;
  if (tid < 64) rk[tid] = 0;
  __syncthreads();
  const int lane = tid & 63, w = tid >> 6, q = np / TK_RANK_WAVES, me = base + lane;
  const unsigned long long mine = sk[min(me, np - 1)];
  const ulonglong2* p = reinterpret_cast<const ulonglong2*>(sk + w * q);
  int r = 0;
  for (int i = 0; i < (q >> 1); i++) {
    const ulonglong2 v = p[i];
    r += (v.x < mine ? 1 : 0) + (v.y < mine ? 1 : 0);
  }
  atomicAdd(&rk[lane], r);
  __syncthreads();
  if (tid < 64 && me < n) {
    uint32_t* o = sel + (long)img * P.in.koff[P.in.L] + P.in.koff[l];
    const int r = rk[tid];
    o[r] = (uint32_t)mine;
    if (RPN) {  // decode the anchor this pair selects, into row r of its segment
      // (constant indices only into the kernel-argument structs)
      const float4* dl = E.deltas[0];
      const float4* an = E.anchors[0];
      const float* lg = P.in.ptr[0];
      long stride = P.in.stride[0];
#pragma unroll
      for (int q = 1; q < TOPK_MAX_LEVELS; q++)
        if (q == l) { dl = E.deltas[q]; an = E.anchors[q]; lg = P.in.ptr[q]; stride = P.in.stride[q]; }
      int W = E.img_w[0], H = E.img_h[0];
#pragma unroll
      for (int q = 1; q < 16; q++)
        if (q == img) { W = E.img_w[q]; H = E.img_h[q]; }
      const int a = (int)(uint32_t)mine, j = P.in.koff[l] + r;
      rpn_decode_row(an[a], dl[(long)img * stride + a], lg[(long)img * stride + a], (float)W, (float)H, E.wx, E.wy, E.ww,
                     E.wh, E.scale_clamp, E.min_size, (long)img * P.in.koff[P.in.L] + j, j, l, img == 0, E.boxes,
                     E.scores, E.valid, E.level_ids, E.flags);
    }
  }
}

__global__ __launch_bounds__(256) void tk_merge_kernel(TkParams P, const SegState* __restrict__ st,
                                                      const unsigned long long* __restrict__ cand, int kmax, int TK_RUN,
                                                      uint32_t* __restrict__ sel) {
  const int seg = blockIdx.x, l = seg % P.in.L, img = seg / P.in.L;
  const int total = st[seg].cnt;
  if (total <= TK_RUN) return;  // tk_sort_kernel wrote the result
  const int runs = (total + TK_RUN - 1) / TK_RUN;
  const unsigned long long* base = cand + (long)seg * kmax;
  uint32_t* o = sel + (long)img * P.in.koff[P.in.L] + P.in.koff[l];
  for (int i = blockIdx.y * 256 + threadIdx.x; i < total; i += gridDim.y * 256) {
    const unsigned long long key = base[i];
    const int own = i / TK_RUN;
    int rank = i - own * TK_RUN;
    for (int r = 0; r < runs; r++) {
      if (r == own) continue;
      const unsigned long long* run = base + (long)r * TK_RUN;
      int lo = 0, hi = min(total - r * TK_RUN, TK_RUN);  // first position whose key is >= key (keys are unique)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (run[mid] < key) lo = mid + 1; else hi = mid;
      }
      rank += lo;
    }
    o[rank] = (uint32_t)key;
  }
}

struct TkWs { SegState* st; int* hist; int* blk_ties; int* blk_def; unsigned long long* cand; size_t zero_bytes, total; int maxblk, kmax, reps, tickets; };
static size_t tk_al(size_t x) { return (x + 255) / 256 * 256; }
static TkWs tk_carve(const TopkInput& in, void* base) {
  TkWs w{};
  const long ns = (long)in.N * in.L;
  int maxsize = 0, kmax = 1;
  for (int l = 0; l < in.L; l++) { maxsize = in.size[l] > maxsize ? in.size[l] : maxsize; kmax = in.k[l] > kmax ? in.k[l] : kmax; }
  const int chunks = maxsize > 0 ? (maxsize + TK_CHUNK - 1) / TK_CHUNK : 1;
  // small segments (RPN): few workgroups, the last one of a segment scans (tickets) -- saves 4 launches; large ones
  // (RetinaNet: 3,000 chunks per segment): one chunk per workgroup for occupancy, separate scan launches
  w.tickets = chunks <= 256;
  // <= ~1,000 workgroups per large segment (amortises the LDS histogram), each of >= 4 chunks (the compaction loads one
  // chunk ahead)
  w.reps = w.tickets ? 1 : std::max(4, (chunks + 1023) / 1024);
  w.maxblk = (chunks + w.reps - 1) / w.reps;
  w.kmax = kmax;
  size_t off = 0;
  auto take = [&](size_t b) { void* r = base ? (char*)base + off : nullptr; off += tk_al(b); return r; };
  w.st = (SegState*)take(ns * sizeof(SegState));
  w.hist = (int*)take(ns * 3 * TK_BINS * sizeof(int));
  w.zero_bytes = off;  // states + histograms are zeroed per call
  w.blk_ties = (int*)take(ns * w.maxblk * sizeof(int));
  w.blk_def = w.tickets ? nullptr : (int*)take(ns * w.maxblk * sizeof(int));
  w.cand = (unsigned long long*)take(ns * kmax * sizeof(unsigned long long));
  w.total = off;
  return w;
}

float logit_lower_bound(float thr) {
  if (thr != thr || thr >= 1.f) return __builtin_nanf("");
  if (thr < 0.f) return -__builtin_inff();
  if (thr == 0.f) return -88.72283f;  // largest -x with expf(-x) finite: below it the fp32 score is 1 / inf = 0
  const double t = (double)thr, T = log(t / (1.0 - t));
  float f = (float)T;  // nearest fp32 to T: the bound is f if f > T, else its successor
  if (!((double)f > T)) f = nextafterf(f, __builtin_inff());
  return f;
}

size_t topk_workspace_bytes(const TopkInput& in) { return tk_carve(in, nullptr).total + 256; }

int topk_select(const TopkInput& in, bool use_thr, float xmin, uint32_t* sel, int* cnt, void* ws, size_t ws_bytes,
                hipStream_t s, int* clear_word, const TopkRpnEpilogue* rpn, bool* rpn_done) {
  if (rpn_done) *rpn_done = false;
  D2_CHECK_ARG(in.L >= 1 && in.L <= TOPK_MAX_LEVELS && in.N >= 1, "topk_select: bad segment layout");
  const TkWs w = tk_carve(in, ws);
  if (ws == nullptr || ws_bytes < w.total) {
    set_error("topk_select: workspace too small (%zu < %zu)", ws_bytes, w.total);
    return D2AMD_EWORKSPACE;
  }
  D2_CHECK_ARG(w.kmax <= TOPK_MAX_K, "topk_select: k = %d per segment exceeds %d", w.kmax, TOPK_MAX_K);
  D2_CHECK_ARG((long)in.N * in.L <= 65535, "topk_select: too many segments");
  TkParams P{};
  P.in = in;
  P.use_thr = use_thr; P.xmin = xmin;
  P.maxblk = w.maxblk;
  P.reps = w.reps;
  P.tickets = w.tickets;
  P.blk_def = w.blk_def;
  { const int zrc = zero_async(ws, w.zero_bytes, s, clear_word); if (zrc) return zrc; }
  dim3 grid(w.maxblk, in.N * in.L), block(TK_THREADS);
  const dim3 segs(in.N * in.L);
  // one launch when every workgroup of the grid is resident at once (see tk_fused_kernel): two 256-thread
  // workgroups per CU are always possible (8 KB + 1 KB LDS, < 128 VGPRs)
  static const int resident = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return 2 * cus;
  }();
  static const bool no_fused = getenv("D2AMD_TOPK_MULTI") != nullptr;  // A/B switch: the multi-launch path + sort
  long live_wgs = 0;  // workgroups that do not exit at once (the grid is sized for the largest segment)
  for (int l = 0; l < in.L; l++) live_wgs += (long)in.N * ((in.size[l] + TK_CHUNK - 1) / TK_CHUNK);
  if (w.tickets && w.reps == 1 && live_wgs <= resident && !no_fused) {
    hipLaunchKernelGGL(tk_fused_kernel, grid, block, 0, s, P, w.st, w.hist, w.blk_ties, w.cand, w.kmax);
    D2_LAUNCH_OK();
  } else {
  hipLaunchKernelGGL(tk_hist_kernel<0>, grid, block, 0, s, P, w.st, w.hist);
  if (!w.tickets) hipLaunchKernelGGL(tk_scan_kernel<0>, segs, block, 0, s, P, w.st, w.hist);
  hipLaunchKernelGGL(tk_hist_kernel<1>, grid, block, 0, s, P, w.st, w.hist);
  if (!w.tickets) hipLaunchKernelGGL(tk_scan_kernel<1>, segs, block, 0, s, P, w.st, w.hist);
  hipLaunchKernelGGL(tk_hist_kernel<2>, grid, block, 0, s, P, w.st, w.hist);
  if (!w.tickets) hipLaunchKernelGGL(tk_scan_kernel<2>, segs, block, 0, s, P, w.st, w.hist);
  hipLaunchKernelGGL(tk_ties_kernel, grid, block, 0, s, P, w.st, w.blk_ties);
  if (!w.tickets) hipLaunchKernelGGL(tk_ties_scan_kernel, segs, block, 0, s, P, w.st, w.blk_ties);
  hipLaunchKernelGGL(tk_compact_kernel, grid, block, 0, s, P, w.st, w.blk_ties, w.cand, w.kmax);
  }
  if (w.kmax <= TK_RANK_MAX && !no_fused) {  // (the A/B switch also keeps the bitonic sort under test)
    static const bool no_epi = getenv("D2AMD_RPN_NO_FUSED_DECODE") != nullptr;  // A/B switch
    if (rpn && rpn_done && in.N <= 16 && !no_epi) {
      hipLaunchKernelGGL(tk_rank_kernel<true>, dim3(in.N * in.L, cdiv(w.kmax, 64)), dim3(TK_RANK_THREADS), 0, s, P, w.st, w.cand,
                         w.kmax, sel, cnt, *rpn);
      *rpn_done = true;
    } else {
      hipLaunchKernelGGL(tk_rank_kernel<false>, dim3(in.N * in.L, cdiv(w.kmax, 64)), dim3(TK_RANK_THREADS), 0, s, P, w.st, w.cand,
                         w.kmax, sel, cnt, TopkRpnEpilogue{});
    }
    D2_LAUNCH_OK();
    return D2AMD_OK;
  }
  const int run = tk_run_for(w.kmax);
  int pow2 = 1;
  while (pow2 < w.kmax && pow2 < run) pow2 <<= 1;
  if ((size_t)pow2 * 8 > 64 * 1024)  // dynamic LDS beyond the default limit
    D2_HIP_OK(hipFuncSetAttribute((const void*)tk_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pow2 * 8));
  const int runs = (w.kmax + run - 1) / run;
  hipLaunchKernelGGL(tk_sort_kernel, dim3(in.N * in.L, runs), dim3(1024), (size_t)pow2 * 8, s, P, w.st, w.cand, w.kmax,
                     run, sel, cnt);
  D2_LAUNCH_OK();
  if (runs > 1) {
    hipLaunchKernelGGL(tk_merge_kernel, dim3(in.N * in.L, cdiv(w.kmax, 1024)), dim3(256), 0, s, P, w.st, w.cand, w.kmax,
                       run, sel);
    D2_LAUNCH_OK();
  }
  return D2AMD_OK;
}

}  // namespace d2amd
