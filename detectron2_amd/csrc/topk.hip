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
// LARGE segments (more than 256 chunks = 1 M scores: RetinaNet's class logits) read the scores TWICE instead (r04, see
// "large segments" below): pass 0 (tk_hist0_span_kernel), then ONE gather pass that writes what pass 0 already proves
// selected and collects the k-th candidate's 11-bit bucket into a small pool, whose radix select (remaining key bits,
// then the element index) never touches the scores again.
#include <cmath>

#include "topk.h"

#include <mutex>

namespace d2amd {

constexpr int TK_THREADS = 256;
constexpr int TK_ITEMS = 16;
constexpr int TK_CHUNK = TK_THREADS * TK_ITEMS;  // elements per workgroup
constexpr int TK_BINS = 2048;

struct SegState {
  uint32_t prefix;  // key bits resolved so far
  int k_rem;        // still to take from the current bucket
  int cnt_lt, cnt_tie;  // output reservations (adjacent and 8-byte aligned: tk_gather_kernel advances both with one atomic)
  int c_lt;         // candidates strictly better than the current bucket
  int total;        // candidates of the segment
  int take_all;     // fewer candidates than k: all of them are selected
  int ties_total, need;
  int done[4];      // workgroups finished per pass (0-2: histogram passes, 3: tie count)
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
  unsigned long long* stamps;  // profiling only (D2AMD_TOPK_STAMPS): [workgroup][6] wall-clock stamps of tk_gather_kernel
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
  // PASS 0 counts EVERY candidate, and a score distribution puts most of a wave's 64 values into a handful of bins:
  // lanes adding to the same LDS word are served one after the other.  Four copies of the bins, one per lane & 3 (8
  // words of padding between them: the same bin of two copies lies in different banks), quarter that.
  constexpr int COPIES = PASS == 0 ? 4 : 1, CPITCH = TK_BINS + 8;
  __shared__ int h[COPIES * CPITCH];
  __shared__ int s_last, s_def;
  for (int i = tid; i < COPIES * CPITCH; i += TK_THREADS) h[i] = 0;
  if (tid == 0) s_def = 0;
  __syncthreads();
  int* hc = h + (PASS == 0 ? (tid & 3) * CPITCH : 0);
  int n_def = 0;
  const uint32_t prefix = PASS > 0 ? S->prefix : 0u;
  const float* x = P.in.ptr[l] + (long)img * P.in.stride[l];
  float v[TK_ITEMS], nv[TK_ITEMS];
  bool ok[TK_ITEMS], nok[TK_ITEMS];
  tk_load(x, base0, size, v, ok);
  for (int rep = 0; rep < P.reps; rep++) {
    const long base = base0 + (long)rep * TK_CHUNK;
    // the next chunk's loads are in flight while this one is counted (the resident workgroups otherwise alternate
    // between a load phase and a count phase in lock step: the memory system idles during the second)
    const bool more = rep + 1 < P.reps && base + TK_CHUNK < size;  // uniform
    if (more) tk_load(x, base + TK_CHUNK, size, nv, nok);
#pragma unroll
    for (int j = 0; j < TK_ITEMS; j++) {
      uint32_t key;
      if (!ok[j] || !tk_key(P, v[j], key)) continue;
      if (PASS == 0) atomicAdd(&hc[key >> 21], 1);
      else if (PASS == 1) { if ((key >> 21) == prefix) atomicAdd(&h[(key >> 10) & 2047u], 1); }
      else {
        if ((key >> 10) == prefix) atomicAdd(&h[key & 1023u], 1);
        n_def += (key >> 10) < prefix ? 1 : 0;
      }
    }
    if (!more) break;
#pragma unroll
    for (int j = 0; j < TK_ITEMS; j++) { v[j] = nv[j]; ok[j] = nok[j]; }
  }
  if (PASS == 2 && P.blk_def != nullptr) {  // uniform
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) n_def += __shfl_xor(n_def, d, 64);
    if ((tid & 63) == 0 && n_def) atomicAdd(&s_def, n_def);
  }
  __syncthreads();
  if (PASS == 2 && P.blk_def != nullptr && tid == 0) P.blk_def[(long)seg * P.maxblk + blockIdx.x] = s_def;
  int* gh = hist + ((long)seg * 3 + PASS) * TK_BINS;
  for (int i = tid; i < TK_BINS; i += TK_THREADS) {
    int c = h[i];
    if (PASS == 0) c += h[CPITCH + i] + h[2 * CPITCH + i] + h[3 * CPITCH + i];
    if (c) atomicAdd(&gh[i], c);
  }
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

// (AGENT: the bins were written by other workgroups of THIS launch -- device-scope loads, performed at the memory side;
// bins of an earlier launch are read with plain loads: 7,390 workgroups x 2,048 device-scope loads of the same 80 KB
// made the first gather pass 98 us long)
template <int BINS, bool AGENT = true>
__device__ __forceinline__ TkScanOut tk_scan_local(const int* gh, int k_rem, int* lds4, int* s_pair) {
  const int tid = threadIdx.x;
  constexpr int PER = BINS / TK_THREADS;
  int loc[PER], sum = 0;
  if (AGENT) {
#pragma unroll
    for (int j = 0; j < PER; j++) { loc[j] = ld_agent(&gh[tid * PER + j]); sum += loc[j]; }
  } else {
    static_assert(PER == 8 || PER == 4, "two or one 16-B loads per thread");
    const int4* g4 = reinterpret_cast<const int4*>(gh + tid * PER);
#pragma unroll
    for (int j = 0; j < PER / 4; j++) {
      const int4 q = g4[j];
      loc[4 * j] = q.x; loc[4 * j + 1] = q.y; loc[4 * j + 2] = q.z; loc[4 * j + 3] = q.w;
      sum += q.x + q.y + q.z + q.w;
    }
  }
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
static inline int tk_run_for(int kmax) {  // (<= 4,096: one run)
  static const int run_env = d2_prof_env("D2AMD_TOPK_RUN") ? atoi(d2_prof_env("D2AMD_TOPK_RUN")) : 0;  // A/B switch: 1024 / 2048 / 4096
  // 2,048 while the other runs of a segment fit tk_merge_lds_kernel's LDS (k <= 20,480: RetinaNet's 20,000), else 4,096.
  // Measured for 10 x 20,000: sort + merge 36.2 + 9.9 us with 4,096, 21.1 + 15.1 with 2,048, 14.7 + 24.0 with 1,024.
  const int dflt = (long)((kmax + 2047) / 2048 - 1) * 2048 * 8 <= 152 * 1024 ? 2048 : 4096;
  const int run = (run_env == 1024 || run_env == 2048 || run_env == 4096) ? run_env : dflt;
  return kmax <= 4096 ? TK_RUN_MAX : run;
}

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

// ---- large segments (RetinaNet: 3,000 chunks per segment), r04: TWO reads of the scores instead of four / five ------------
// The legacy chain above reads every score in pass 0, 1, 2, (ties,) and the compaction: 129 MB x 4-5 per RetinaNet batch of
// two images, 164 us for a selection of 200,000 pairs (profiles/r03/final/retinanet_100k_kernel_stats.csv).  Pass 0 alone
// already says in WHICH 11-bit bucket b0 the k-th best candidate lies, how many candidates are better than the bucket
// (c_lt < k) and how many it holds (M).  So the second read of the scores is the last one:
//   tk_gather_kernel  every workgroup scans pass 0's bins itself (same bins -> same b0; no scan launch), then writes
//                     the candidates of better buckets straight into the selection ("definite": they are selected
//                     whatever the lower key bits say) and the candidates of bucket b0 into the segment's POOL as
//                     (key : index) pairs.  A workgroup loads its 4 chunks (64 values per thread) in one go and makes
//                     ONE reservation per counter.
//   tk_pool1_kernel / tk_pool_kernel
//                     the k - c_lt best of the M pool pairs: radix select on the remaining 21 key bits and -- only if
//                     there are more ties at the k-th key than needed -- on the element index (the pair as ONE 53-bit
//                     value: lower index first, as the reference's stable order; pairs are unique, so the select
//                     always ends with "take the whole bucket").  M is ~k for smooth score distributions (an 11-bit
//                     bucket is 19-25 % of the value wide): pools of <= 24,576 pairs are selected by one 1,024-thread
//                     workgroup per segment, pairs in registers; larger ones by 16 co-resident workgroups per segment
//                     with segment barriers (tk_segment_barrier).  The pool holds tk_pool_cap(k) pairs; a larger bucket
//                     (all scores nearly equal: an untrained head) is selected by the same kernel FROM THE SCORES
//                     (filtering bucket b0 on the fly: slow -- 16 workgroups per segment -- but exact and bounded).
// Measured (selection alone, 2 x 16.1 M logits, k = 20,000 x 5 levels, one box): 0.263 ms with the five-read chain,
// 0.135 ms with this one (pass 0 34 us, gather ~30 us, pool 13 us, sort 21 us, LDS-staged rank merge 15 us).
constexpr int TK_GSPAN = 8;        // chunks per workgroup of the gather pass (32,768 scores)
constexpr int TK_GGROUP = 2;       // chunks per load group: 32 values per thread, two groups in flight
constexpr int TK_STAGE = 1024;     // staged pairs per list and workgroup (3 % of its scores)
constexpr int TK_POOL_G_MAX = 32;  // workgroups per segment of the pool kernel (D2AMD_TOPK_POOL_G)
constexpr int TK_POOL_U = 8;       // pool pairs per thread and batch
constexpr int TK_POOL_PASSES = 5;  // 53 bits = 21 key bits + 32 index bits: digits of 11, 11, 11, 11, 9 bits
static inline int tk_pool_cap(int size, int k) { return std::min(size, std::max(4 * k, 131072)); }
struct TkPool {
  unsigned long long* mem;  // [N][sum of cap]: pool of segment (img, l) at img * per_img + off[l]
  int* hist;                // [segments][TK_POOL_PASSES][TK_BINS], then [segments][8] counters (barriers 0-4, 7: output)
  int cap[TOPK_MAX_LEVELS];
  long off[TOPK_MAX_LEVELS];
  long per_img;
  int no_small;             // D2AMD_TOPK_POOL_NO_SMALL (A/B and test switch): every pool goes to tk_pool_kernel
};

// Pass 0 of the large segments: the same counts as tk_hist_kernel<0> (bins = key >> 21 of every candidate), laid out like
// the gather pass: 8-chunk spans, 32 values per thread and group through immediate offsets, the next group in flight
// while this one is counted, no per-element bounds test or branch -- a value that is no candidate (below the bound)
// is counted in a spare word of the bins' padding.  (tk_hist_kernel<0> on the 2 x 16.1 M logits: 15 VALU + 9 SALU
// instructions per element and wave, 45 us; with four lane-interleaved copies of the bins 37 us.)
// LDS layout: what bounds the pass is lanes of a wave adding to the SAME LDS word (a score distribution puts a third of
// the values into one bin): COPIES copies of the bins, one per lane % COPIES, and -- a workgroup counts at most 32,768
// values -- 16-bit counters, bins b and b + 1,024 (the two signs: rarely both popular) sharing a word, so that 16 copies
// are 66 KB.
// VEC: a lane loads 16 B (four consecutive scores; 1 KB per wave and load instead of 256 B) -- needs 16-B aligned segments.
template <int COPIES, bool VEC>
__global__ __launch_bounds__(TK_THREADS) void tk_hist0_span_kernel(TkParams P, int* __restrict__ hist) {
  const int seg = blockIdx.y, l = seg % P.in.L, img = seg / P.in.L;
  const int size = P.in.size[l];
  constexpr long SPAN = (long)TK_GSPAN * TK_CHUNK;
  static_assert(SPAN < 65536, "16-bit counters");
  const long base0 = (long)blockIdx.x * SPAN;
  if (base0 >= size) return;
  const int tid = threadIdx.x;
  constexpr int WORDS = TK_BINS / 2, CPITCH = WORDS + 8;  // (word WORDS of a copy: values that are no candidates)
  __shared__ uint32_t h[COPIES * CPITCH];
  for (int i = tid; i < COPIES * CPITCH; i += TK_THREADS) h[i] = 0u;
  __syncthreads();
  uint32_t* hc = h + (tid % COPIES) * CPITCH;
  const float* x = P.in.ptr[l] + (long)img * P.in.stride[l];
  const bool thr = P.use_thr != 0;
  const float xmin = P.xmin;
  auto count = [&](float val) __attribute__((always_inline)) {
    const uint32_t b = topk_desc_key(val) >> 21;
    const bool c = !thr || val >= xmin;
    atomicAdd(&hc[c ? (b & (WORDS - 1)) : (uint32_t)WORDS], b >= (uint32_t)WORDS ? 0x10000u : 1u);
  };
  if (base0 + SPAN <= size) {  // uniform: every group of the span is complete
    constexpr int NV = TK_GGROUP * TK_ITEMS, G = TK_GSPAN / TK_GGROUP;
    float va[NV], vb[NV];
    auto load = [&](float (&v)[NV], int gi) __attribute__((always_inline)) {
      if (VEC) {
        const float4* xb = reinterpret_cast<const float4*>(x + base0 + (long)gi * (TK_GGROUP * TK_CHUNK)) + tid;
#pragma unroll
        for (int q = 0; q < NV / 4; q++) {
          const float4 t = xb[q * TK_THREADS];
          v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
      } else {
        const float* xb = x + base0 + (long)gi * (TK_GGROUP * TK_CHUNK) + tid;
#pragma unroll
        for (int j = 0; j < NV; j++) v[j] = xb[j * TK_THREADS];
      }
    };
    load(va, 0);
#pragma unroll 1
    for (int gi = 0; gi < G; gi++) {
      if (gi + 1 < G) load(vb, gi + 1);
#pragma unroll
      for (int j = 0; j < NV; j++) count(va[j]);
#pragma unroll
      for (int j = 0; j < NV; j++) va[j] = vb[j];
    }
  } else {  // the segment's last span
    for (long base = base0; base < size; base += TK_CHUNK) {
      float v[TK_ITEMS];
      bool ok[TK_ITEMS];
      tk_load(x, base, size, v, ok);
#pragma unroll
      for (int j = 0; j < TK_ITEMS; j++)
        if (ok[j]) count(v[j]);
    }
  }
  __syncthreads();
  int* gh = hist + (long)seg * 3 * TK_BINS;
  for (int i = tid; i < WORDS; i += TK_THREADS) {
    uint32_t lo = 0u, hi = 0u;
#pragma unroll
    for (int c = 0; c < COPIES; c++) { const uint32_t w = h[c * CPITCH + i]; lo += w & 0xffffu; hi += w >> 16; }
    if (lo) atomicAdd(&gh[i], (int)lo);
    if (hi) atomicAdd(&gh[i + WORDS], (int)hi);
  }
}

template <bool VEC>
__global__ __launch_bounds__(TK_THREADS, 4) void tk_gather_kernel(TkParams P, SegState* __restrict__ st,
                                                                 const int* __restrict__ hist,
                                                                 unsigned long long* __restrict__ cand, int kmax,
                                                                 const TkPool Q) {
  const int seg = blockIdx.y, l = seg % P.in.L, img = seg / P.in.L;
  const int size = P.in.size[l];
  constexpr long SPAN = (long)TK_GSPAN * TK_CHUNK;
  const long base0 = (long)blockIdx.x * SPAN;
  // a segment that selects a large share of its scores ("dense", decided below from pass 0's bins; a small segment by
  // nature) is handed out chunk by chunk when the grid has a workgroup per chunk: its workgroups stay for the scan
  const int chunks = (size + TK_CHUNK - 1) / TK_CHUNK;
  const bool by_chunk_possible = chunks <= (int)gridDim.x;
  if (base0 >= size && !(by_chunk_possible && (int)blockIdx.x < chunks)) return;
  SegState* S = st + seg;
  const int tid = threadIdx.x;
  __shared__ int lds4[TK_THREADS / 64];
  __shared__ int s_pair[2], s_base[2], s_cnt[2];
  __shared__ unsigned long long stage[2][TK_STAGE];
  const float* x = P.in.ptr[l] + (long)img * P.in.stride[l];
  unsigned long long* out_def = cand + (long)seg * kmax;
  int cap = Q.cap[0];
  long poff = Q.off[0];
#pragma unroll
  for (int q = 1; q < TOPK_MAX_LEVELS; q++)
    if (q == l) { cap = Q.cap[q]; poff = Q.off[q]; }
  unsigned long long* out_pool = Q.mem + (long)img * Q.per_img + poff;
#ifdef D2AMD_TOPK_STAMPS  // (profiling builds only: -DD2AMD_TOPK_STAMPS, scripts/topk_stamps.py)
#define TKST(k) do { if (P.stamps && tid == 0) P.stamps[((size_t)seg * gridDim.x + blockIdx.x) * 6 + (k)] = wall_clock64(); } while (0)
#else
#define TKST(k) do { } while (0)
#endif
  TKST(0);
  const bool whole = base0 + SPAN <= size;  // uniform: the span exists and every group of it is complete
  constexpr int NV = TK_GGROUP * TK_ITEMS;  // values per thread and group
  // group gi of the span: one base address, immediate offsets (a clamped index per load costs a 64-bit address each)
  // (VEC: 16 B per lane -- value j of a thread is element (j >> 2) * 1,024 + 4 tid + (j & 3) of the group)
  auto load = [&](float (&v)[NV], int gi) __attribute__((always_inline)) {
    if (VEC) {
      const float4* xb = reinterpret_cast<const float4*>(x + base0 + (long)gi * (TK_GGROUP * TK_CHUNK)) + tid;
#pragma unroll
      for (int q = 0; q < NV / 4; q++) {
        const float4 t = xb[q * TK_THREADS];
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      }
    } else {
      const float* xb = x + base0 + (long)gi * (TK_GGROUP * TK_CHUNK) + tid;
#pragma unroll
      for (int j = 0; j < NV; j++) v[j] = xb[j * TK_THREADS];
    }
  };
  float va[NV], vb[NV];
  if (whole) load(va, 0);  // (flies while the bins are scanned)
  // ---- what pass 0 found: every workgroup scans the bins itself
  uint32_t b0 = 0;
  float x0;  // no selected element is below it
  bool take_all, pool_on = false, bucket_def = false, dense_seg;
  {
    const int k = P.in.k[l];
    const TkScanOut o = tk_scan_local<2048, false>(hist + (long)seg * 3 * TK_BINS, k, lds4, s_pair);
    take_all = o.total < k;  // fewer candidates than k: all of them are selected
    int M = 0, k_rem = 0;
    if (!take_all) {
      b0 = (uint32_t)o.bin;
      k_rem = k - o.before;
      M = hist[(long)seg * 3 * TK_BINS + o.bin];
      bucket_def = M == k_rem;           // the whole bucket is selected: no pool
      pool_on = !bucket_def && M <= cap;  // else: tk_pool_kernel reads the scores
      // the value whose key is the bucket's worst, (b0 << 21) | 0x1fffff: topk_desc_key inverted
      const uint32_t m = ~((b0 << 21) | 0x1fffffu);
      x0 = __uint_as_float((m & 0x80000000u) ? (m ^ 0x80000000u) : ~m);
    } else {
      x0 = !P.use_thr ? -__builtin_inff() : (P.xmin != P.xmin) ? __builtin_inff() : P.xmin;
    }
    if (blockIdx.x == 0 && tid == 0) {
      S->total = o.total; S->take_all = take_all ? 1 : 0; S->cnt = take_all ? o.total : k;
      S->prefix = b0; S->c_lt = take_all ? 0 : o.before; S->k_rem = bucket_def ? 0 : k_rem;
      S->ties_total = bucket_def ? 0 : M; S->need = pool_on ? 1 : 0;
    }
    // dense: a span's expected share of either list would not fit the staging list (60 %: the share is not uniform)
    const long n_list = take_all ? o.total : max(o.before + (bucket_def ? M : 0), pool_on ? M : 0);
    dense_seg = n_list * SPAN > (long)size * (TK_STAGE * 6 / 10);
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
  }
  TKST(1);
  // exact class of a value: 1 = definite (selected whatever the lower key bits say), 2 = pool, 0 = neither
  auto cls = [&](float val, uint32_t& key) __attribute__((always_inline)) {
    key = 0;
    const bool c = tk_key(P, val, key);
    const uint32_t b = key >> 21;
    const bool d = c && (take_all || b < b0 || (bucket_def && b == b0));
    return d ? 1 : (c && pool_on && b == b0) ? 2 : 0;
  };
  // SPARSE (a workgroup of a large segment selects ~0.3 % of its values): a selected element is appended to a staging
  // list in LDS right where it is found, behind a wave-uniform test that is ONE float compare per element -- a selected
  // element is not below x0 (NaN passes and is classified exactly); 84 % of a wave's 64-element rows hold none.  The
  // lists go out as contiguous runs after ONE reservation per workgroup.  (Measured on the way, 2 x 16.1 M RetinaNet
  // logits: classify twice around a block scan 94 us; exact class of every element 87 us -- 73 issue cycles per element
  // and wave; float pretest 83 us, of which 8 us per workgroup waiting for its turn at the segment's counters: 739
  // workgroups x 2 returning atomics on one line, all at the same moment, and the load and compute phases of the
  // resident workgroups alternating in lock step.  Hence 16-chunk spans, double-buffered groups and one packed atomic.)
  auto sift = [&](const float (&v)[NV], int gi) __attribute__((always_inline)) {
    // (the element index is built from an opaque copy of the thread index: derived from `tid` the values tid | j << 8
    // were hoisted out of the loop and held in a register each)
    uint32_t ebase = (uint32_t)(base0 + (long)gi * (TK_GGROUP * TK_CHUNK)) + (uint32_t)tid * (VEC ? 4u : 1u);
    asm volatile("" : "+v"(ebase));
#pragma unroll
    for (int j = 0; j < NV; j++) {
      if (__ballot(!(v[j] < x0)) != 0ull) {  // uniform per wave
        uint32_t key;
        const int c = cls(v[j], key);
        if (c != 0) {
          const int pos = atomicAdd(&s_cnt[c - 1], 1);
          if (pos < TK_STAGE)
            stage[c - 1][pos] = ((unsigned long long)key << 32) | (ebase + (VEC ? (j >> 2) * (4 * TK_THREADS) + (j & 3) : j * TK_THREADS));
        }
      }
    }
  };
  // DENSE: one chunk (values re-loaded where the sparse attempt held them), block scan + one reservation per list
  auto dense_chunk = [&](long base) __attribute__((always_inline)) {
    float w[TK_ITEMS];
    const int lc = (int)min((long)size - base, (long)TK_CHUNK);
#pragma unroll
    for (int j = 0; j < TK_ITEMS; j++) w[j] = x[base + min(j * TK_THREADS + tid, lc - 1)];
    int m = 0;
#pragma unroll
    for (int j = 0; j < TK_ITEMS; j++) {
      uint32_t key;
      const int c = j * TK_THREADS + tid < lc ? cls(w[j], key) : 0;
      m += c == 1 ? 1 : c == 2 ? 0x10000 : 0;
    }
    int tot;
    const int my = tk_block_excl_scan(m, lds4, tot);
    if (tid == 0) {
      s_base[0] = (tot & 0xffff) ? atomicAdd(&S->cnt_lt, tot & 0xffff) : 0;
      s_base[1] = (tot >> 16) ? atomicAdd(&S->cnt_tie, tot >> 16) : 0;
    }
    __syncthreads();
    int p_def = s_base[0] + (my & 0xffff), p_pl = s_base[1] + (my >> 16);
#pragma unroll
    for (int j = 0; j < TK_ITEMS; j++) {
      uint32_t key = 0;
      const int c = j * TK_THREADS + tid < lc ? cls(w[j], key) : 0;
      const unsigned long long e = ((unsigned long long)key << 32) | (uint32_t)(base + j * TK_THREADS + tid);
      if (c == 1) out_def[p_def++] = e;
      else if (c == 2) out_pool[p_pl++] = e;
    }
    __syncthreads();  // s_base / lds4 are reused
  };
  if (dense_seg && by_chunk_possible) {  // uniform over the segment: workgroup = chunk
    TKST(5);
    dense_chunk((long)blockIdx.x * TK_CHUNK);
    TKST(4);
    return;
  }
  if (base0 >= size) return;  // (stayed for a chunk-wise hand-out that did not happen)
  if (whole && !dense_seg) {
    constexpr int G = TK_GSPAN / TK_GGROUP;
#pragma unroll 1
    for (int gi = 0; gi < G; gi++) {  // the next group's loads are in flight while this one is sifted
      if (gi + 1 < G) load(vb, gi + 1);
      sift(va, gi);
      // (a copy, not a second sift of vb in the same iteration: around the back edge the compiler's wait counts made
      // that one wait for the group issued AFTER it as well)
#pragma unroll
      for (int j = 0; j < NV; j++) va[j] = vb[j];
    }
    __syncthreads();
    TKST(2);
    const int t_def = s_cnt[0], t_pl = s_cnt[1];
    if (t_def <= TK_STAGE && t_pl <= TK_STAGE) {  // uniform
      if (t_def + t_pl == 0) return;
      if (tid == 0) {  // cnt_lt (low word) and cnt_tie (high word) advance together
        const unsigned long long r = atomicAdd(reinterpret_cast<unsigned long long*>(&S->cnt_lt),
                                               (unsigned long long)t_def | ((unsigned long long)t_pl << 32));
        s_base[0] = (int)(uint32_t)r; s_base[1] = (int)(r >> 32);
      }
      __syncthreads();
      TKST(3);
      for (int i = tid; i < t_def; i += TK_THREADS) out_def[s_base[0] + i] = stage[0][i];
      for (int i = tid; i < t_pl; i += TK_THREADS) out_pool[s_base[1] + i] = stage[1][i];
      TKST(4);
      return;
    }
    // a list overflowed (the selected scores are clustered): nothing has been written yet, the span is walked again
  }
  TKST(5);
  // the segment's last, partial span; a dense segment with more chunks than the grid is wide; the overflow above
  for (long base = base0; base < min((long)size, base0 + SPAN); base += TK_CHUNK) dense_chunk(base);  // uniform
  TKST(4);
#undef TKST
}

struct SegState;
__device__ __forceinline__ bool tk_pool_is_small(const SegState* S);
__global__ __launch_bounds__(TK_THREADS) void tk_pool_kernel(TkParams P, const SegState* __restrict__ st,
                                                            unsigned long long* __restrict__ cand, int kmax,
                                                            const TkPool Q) {
  const int TK_POOL_G = (int)gridDim.x;  // workgroups per segment
  const int seg = blockIdx.y, l = seg % P.in.L, img = seg / P.in.L, ns = P.in.N * P.in.L;
  const int size = P.in.size[l];
  if (size == 0) return;
  const SegState* S = st + seg;
  int k_rem = S->k_rem;
  if (S->take_all || k_rem == 0) return;  // (uniform over the segment: every selected candidate was definite)
  if (tk_pool_is_small(S) && !Q.no_small) return;  // tk_pool1_kernel's
  const int M = S->ties_total, c_lt = S->c_lt;
  const uint32_t b0 = S->prefix;
  const bool from_pool = S->need != 0;
  const int tid = threadIdx.x;
  long poff = Q.off[0];
#pragma unroll
  for (int q = 1; q < TOPK_MAX_LEVELS; q++)
    if (q == l) poff = Q.off[q];
  const unsigned long long* pool = Q.mem + (long)img * Q.per_img + poff;
  const float* x = P.in.ptr[l] + (long)img * P.in.stride[l];
  const long n_src = from_pool ? M : size;
  constexpr unsigned long long MASK53 = (1ull << 53) - 1ull;
  // pair i of the source, reduced to the 53 bits below the bucket (-> false: not a pair of bucket b0)
  auto fetch = [&](long i, unsigned long long& v) -> bool {
    if (from_pool) { v = pool[i] & MASK53; return true; }
    uint32_t key = 0;
    const bool c = tk_key(P, x[i], key) && (key >> 21) == b0;
    v = (((unsigned long long)key << 32) | (uint32_t)i) & MASK53;
    return c;
  };
  __shared__ int h[TK_BINS];
  __shared__ int lds4[TK_THREADS / 64];
  __shared__ int s_pair[2], s_base;
  int* gh = Q.hist + (long)seg * TK_POOL_PASSES * TK_BINS;
  int* ctr = Q.hist + (long)ns * TK_POOL_PASSES * TK_BINS + seg * 8;
  unsigned long long prefix = 0ull;  // the resolved high bits of the 53
  int shift = 53;                    // ... = bits [52 : shift]
  for (int pass = 0; pass < TK_POOL_PASSES; pass++) {
    const int width = pass == TK_POOL_PASSES - 1 ? 9 : 11;
    const int hi = shift;
    shift -= width;
    const uint32_t dmask = (1u << width) - 1u;
    for (int i = tid; i < TK_BINS; i += TK_THREADS) h[i] = 0;
    __syncthreads();
    for (long b = (long)blockIdx.x * (TK_THREADS * TK_POOL_U); b < n_src; b += (long)TK_POOL_G * (TK_THREADS * TK_POOL_U)) {
      unsigned long long vv[TK_POOL_U];
      bool ok[TK_POOL_U];
#pragma unroll
      for (int u = 0; u < TK_POOL_U; u++) {
        const long i = b + u * TK_THREADS + tid;
        ok[u] = fetch(min(i, n_src - 1), vv[u]) && i < n_src;
      }
#pragma unroll
      for (int u = 0; u < TK_POOL_U; u++)
        if (ok[u] && (hi >= 53 || (vv[u] >> hi) == prefix)) atomicAdd(&h[(uint32_t)(vv[u] >> shift) & dmask], 1);
    }
    __syncthreads();
    for (int i = tid; i < TK_BINS; i += TK_THREADS)
      if (h[i]) atomicAdd(&gh[pass * TK_BINS + i], h[i]);
    tk_segment_barrier(&ctr[pass], TK_POOL_G);
    const TkScanOut o = tk_scan_local<2048>(gh + pass * TK_BINS, k_rem, lds4, s_pair);
    const int in_bin = ld_agent(&gh[pass * TK_BINS + o.bin]);
    prefix = (prefix << width) | (unsigned long long)o.bin;
    k_rem -= o.before;
    if (in_bin == k_rem) break;  // the whole bucket is taken (uniform; at the latest in the last pass: pairs are unique)
  }
  // ---- the selected pairs: (v >> shift) <= prefix, behind the c_lt definite ones ----------------------------------
  unsigned long long* out = cand + (long)seg * kmax + c_lt;
  const unsigned long long top = (unsigned long long)b0 << 53;
  for (long b = (long)blockIdx.x * (TK_THREADS * TK_POOL_U); b < n_src; b += (long)TK_POOL_G * (TK_THREADS * TK_POOL_U)) {
    unsigned long long vv[TK_POOL_U];
    unsigned sel = 0u;
#pragma unroll
    for (int u = 0; u < TK_POOL_U; u++) {
      const long i = b + u * TK_THREADS + tid;
      const bool c = fetch(min(i, n_src - 1), vv[u]) && i < n_src;
      sel |= (unsigned)(c && (vv[u] >> shift) <= prefix) << u;
    }
    const int n = __builtin_popcount(sel);
    if (!__syncthreads_or(n != 0)) continue;  // uniform
    int tot;
    const int my = tk_block_excl_scan(n, lds4, tot);
    if (tid == 0) s_base = atomicAdd(&ctr[7], tot);
    __syncthreads();
    int p = s_base + my;
#pragma unroll
    for (int u = 0; u < TK_POOL_U; u++)
      if (sel & (1u << u)) out[p++] = top | vv[u];
    __syncthreads();  // s_base is reused
  }
}


// The rank merge with the OTHER runs of the segment staged in LDS (RetinaNet: 4 x 4,096 pairs = 128 KB): the 12-step
// binary searches read LDS instead of 48 dependent global loads per element (tk_merge_kernel: 33 us for 10 x 20,000).
// A workgroup covers 1,024 consecutive elements of ONE run (TK_RUN is a multiple of 1,024).
__global__ __launch_bounds__(1024) void tk_merge_lds_kernel(TkParams P, const SegState* __restrict__ st,
                                                           const unsigned long long* __restrict__ cand, int kmax,
                                                           int TK_RUN, uint32_t* __restrict__ sel) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long others[];  // [runs - 1][TK_RUN]
  const int seg = blockIdx.x, l = seg % P.in.L, img = seg / P.in.L;
  const int total = st[seg].cnt;
  if (total <= TK_RUN) return;  // tk_sort_kernel wrote the result
  const int i0 = blockIdx.y * 1024;
  if (i0 >= total) return;
  const int runs = (total + TK_RUN - 1) / TK_RUN, own = i0 / TK_RUN, tid = threadIdx.x;
  const unsigned long long* base = cand + (long)seg * kmax;
  for (int r = 0; r < runs; r++) {
    if (r == own) continue;
    const int n = min(total - r * TK_RUN, TK_RUN);
    unsigned long long* dst = others + (long)(r < own ? r : r - 1) * TK_RUN;
    for (int i = tid; i < n; i += 1024) dst[i] = base[(long)r * TK_RUN + i];
  }
  const int i = i0 + tid;
  const unsigned long long key = base[min(i, total - 1)];
  __syncthreads();
  if (i >= total) return;
  int rank = i - own * TK_RUN;
  for (int r = 0; r < runs; r++) {
    if (r == own) continue;
    const unsigned long long* run = others + (long)(r < own ? r : r - 1) * TK_RUN;
    int lo = 0, hi = min(total - r * TK_RUN, TK_RUN);  // first position whose key is >= key (keys are unique)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (run[mid] < key) lo = mid + 1; else hi = mid;
    }
    rank += lo;
  }
  uint32_t* o = sel + (long)img * P.in.koff[P.in.L] + P.in.koff[l];
  o[rank] = (uint32_t)key;
}

// The usual pool (M <= 24,576 pairs: smooth score distributions give M ~ k) is selected by ONE 1,024-thread workgroup per
// segment with the pairs in registers and the digit histograms in LDS: no segment barrier, no device-scope traffic -- the
// 16-workgroup version above spent ~8 us per digit on those (20 us for two digits + compaction on the bench's pools).
// tk_pool_kernel keeps the segments this one leaves (larger pools, buckets read from the scores).
constexpr int TK_P1_THREADS = 1024, TK_P1_U = 24;
__device__ __forceinline__ bool tk_pool_is_small(const SegState* S) { return S->need != 0 && S->ties_total <= TK_P1_THREADS * TK_P1_U; }

// exclusive prefix of one int per thread over the 1,024-thread workgroup
__device__ __forceinline__ int tk_p1_excl_scan(int v, int* wsum, int& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  __syncthreads();  // wsum may still be read from a previous call
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  int base = 0;
  total = 0;
#pragma unroll
  for (int i = 0; i < TK_P1_THREADS / 64; i++) {
    const int t = wsum[i];
    if (i < w) base += t;
    total += t;
  }
  return base + x - v;
}

__global__ __launch_bounds__(TK_P1_THREADS) void tk_pool1_kernel(TkParams P, const SegState* __restrict__ st,
                                                                unsigned long long* __restrict__ cand, int kmax,
                                                                const TkPool Q) {
  const int seg = blockIdx.x, l = seg % P.in.L, img = seg / P.in.L;
  if (P.in.size[l] == 0) return;
  const SegState* S = st + seg;
  int k_rem = S->k_rem;
  if (S->take_all || k_rem == 0 || !tk_pool_is_small(S) || Q.no_small) return;  // uniform
  const int M = S->ties_total, c_lt = S->c_lt;
  const int tid = threadIdx.x;
  long poff = Q.off[0];
#pragma unroll
  for (int q = 1; q < TOPK_MAX_LEVELS; q++)
    if (q == l) poff = Q.off[q];
  const unsigned long long* pool = Q.mem + (long)img * Q.per_img + poff;
  constexpr unsigned long long MASK53 = (1ull << 53) - 1ull;
  unsigned long long v[TK_P1_U];  // (invalid: all ones -- matches no prefix and is never selected)
#pragma unroll
  for (int u = 0; u < TK_P1_U; u++) {
    const int i = u * TK_P1_THREADS + tid;
    v[u] = i < M ? (pool[i] & MASK53) : ~0ull;
  }
  __shared__ int h[TK_BINS];
  __shared__ int wsum[TK_P1_THREADS / 64];
  __shared__ int s_pair[2];
  unsigned long long prefix = 0ull;
  int shift = 53;
  for (int pass = 0; pass < TK_POOL_PASSES; pass++) {
    const int width = pass == TK_POOL_PASSES - 1 ? 9 : 11;
    const int hi = shift;
    shift -= width;
    const uint32_t dmask = (1u << width) - 1u;
    h[tid] = 0; h[tid + TK_P1_THREADS] = 0;
    if (tid == 0) { s_pair[0] = -1; s_pair[1] = 0; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < TK_P1_U; u++)
      if ((v[u] >> 53) == 0ull && (hi >= 53 || (v[u] >> hi) == prefix)) atomicAdd(&h[(uint32_t)(v[u] >> shift) & dmask], 1);
    __syncthreads();
    const int l0 = h[2 * tid], l1 = h[2 * tid + 1];
    int total;
    const int run = tk_p1_excl_scan(l0 + l1, wsum, total);
    if (run < k_rem && run + l0 >= k_rem) { s_pair[0] = 2 * tid; s_pair[1] = run; }
    else if (run + l0 < k_rem && run + l0 + l1 >= k_rem) { s_pair[0] = 2 * tid + 1; s_pair[1] = run + l0; }
    __syncthreads();
    const int bin = s_pair[0], before = s_pair[1];
    const int in_bin = h[bin];
    __syncthreads();  // h / s_pair are rewritten by the next pass
    prefix = (prefix << width) | (unsigned long long)bin;
    k_rem -= before;
    if (in_bin == k_rem) break;  // the whole bucket is taken (uniform; at the latest in the last pass: pairs are unique)
  }
  unsigned long long* out = cand + (long)seg * kmax + c_lt;
  const unsigned long long top = (unsigned long long)S->prefix << 53;
  int n = 0;
#pragma unroll
  for (int u = 0; u < TK_P1_U; u++) n += ((v[u] >> 53) == 0ull && (v[u] >> shift) <= prefix) ? 1 : 0;
  int total;
  int p = tk_p1_excl_scan(n, wsum, total);
#pragma unroll
  for (int u = 0; u < TK_P1_U; u++)
    if ((v[u] >> 53) == 0ull && (v[u] >> shift) <= prefix) out[p++] = top | v[u];
}

struct TkWs { SegState* st; int* hist; int* blk_ties; int* blk_def; unsigned long long* cand; size_t zero_bytes, total; int maxblk, kmax, reps, tickets; TkPool pool; };
static size_t tk_al(size_t x) { return (x + 255) / 256 * 256; }
static int maxsize_of(const TopkInput& in) {
  int m = 0;
  for (int l = 0; l < in.L; l++) m = in.size[l] > m ? in.size[l] : m;
  return m;
}
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
  if (!w.tickets)  // large segments: the pool kernel's histograms and counters (see tk_pool_kernel)
    w.pool.hist = (int*)take(((size_t)ns * TK_POOL_PASSES * TK_BINS + (size_t)ns * 8) * sizeof(int));
  w.zero_bytes = off;  // states + histograms are zeroed per call
  w.blk_ties = (int*)take(ns * w.maxblk * sizeof(int));
  w.blk_def = w.tickets ? nullptr : (int*)take(ns * w.maxblk * sizeof(int));
  w.cand = (unsigned long long*)take(ns * kmax * sizeof(unsigned long long));
  if (!w.tickets) {
    long per = 0;
    for (int l = 0; l < in.L; l++) {
      w.pool.cap[l] = tk_pool_cap(in.size[l], in.k[l]);
      w.pool.off[l] = per;
      per += w.pool.cap[l];
    }
    w.pool.per_img = per;
    w.pool.mem = (unsigned long long*)take((size_t)in.N * per * sizeof(unsigned long long));
  }
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
  // large segments: two reads of the scores (tk_gather_kernel / tk_pool_kernel).  D2AMD_TOPK_LEGACY: the five-read
  // chain below (A/B and test switch).  The pool kernel's workgroups wait for each other per segment: 32 of them fit
  // on the device many times over, and the dispatcher hands out workgroups in order (segments are consecutive).
  static const bool legacy = getenv("D2AMD_TOPK_LEGACY") != nullptr;
  static const int pool_g = [] {
    const int g = d2_prof_env("D2AMD_TOPK_POOL_G") ? atoi(d2_prof_env("D2AMD_TOPK_POOL_G")) : 16;
    return g < 1 ? 1 : g > TK_POOL_G_MAX ? TK_POOL_G_MAX : g;
  }();
  // (every workgroup of the pool kernel must be resident: its segment barriers spin)
  const long nseg = (long)in.N * in.L;
  const int pool_g_fit = (int)std::max(1l, std::min((long)pool_g, resident / std::max(1l, nseg)));
  if (!w.tickets && !legacy && nseg <= resident) {
    const dim3 ggrid(cdiv(cdiv(maxsize_of(in), TK_CHUNK), TK_GSPAN), in.N * in.L);
    // 16-B loads: every segment starts on a 16-B boundary (spans start at multiples of 32,768 elements)
    static const bool no_vec = getenv("D2AMD_TOPK_NO_VEC") != nullptr;  // A/B and test switch
    bool vec = !no_vec;
    for (int l = 0; l < in.L; l++) vec = vec && ((uintptr_t)in.ptr[l] % 16 == 0) && (in.stride[l] % 4 == 0 || in.N == 1);
    static const bool old_hist0 = d2_prof_env("D2AMD_TOPK_OLD_HIST0") != nullptr;  // A/B switch
    if (old_hist0) hipLaunchKernelGGL(tk_hist_kernel<0>, grid, block, 0, s, P, w.st, w.hist);
    else {
      static const int copies = d2_prof_env("D2AMD_TOPK_HIST_COPIES") ? atoi(d2_prof_env("D2AMD_TOPK_HIST_COPIES")) : 8;
      if (copies <= 4) hipLaunchKernelGGL((tk_hist0_span_kernel<4, false>), ggrid, block, 0, s, P, w.hist);
      else if (!vec) hipLaunchKernelGGL((tk_hist0_span_kernel<8, false>), ggrid, block, 0, s, P, w.hist);
      else hipLaunchKernelGGL((tk_hist0_span_kernel<8, true>), ggrid, block, 0, s, P, w.hist);
    }
#ifdef D2AMD_TOPK_STAMPS
    const char* stamp_path = d2_prof_env("D2AMD_TOPK_STAMPS");  // profiling only: per-workgroup stamps of the gather pass
    const size_t stamp_n = (size_t)grid.x * grid.y * 6;
    if (stamp_path) {
      D2_HIP_OK(hipMalloc(&P.stamps, stamp_n * 8));
      D2_HIP_OK(hipMemsetAsync(P.stamps, 0, stamp_n * 8, s));
    }
#endif
    if (vec) hipLaunchKernelGGL(tk_gather_kernel<true>, ggrid, block, 0, s, P, w.st, w.hist, w.cand, w.kmax, w.pool);
    else hipLaunchKernelGGL(tk_gather_kernel<false>, ggrid, block, 0, s, P, w.st, w.hist, w.cand, w.kmax, w.pool);
#ifdef D2AMD_TOPK_STAMPS
    if (stamp_path) {
      D2_HIP_OK(hipStreamSynchronize(s));
      unsigned long long* h = (unsigned long long*)malloc(stamp_n * 8);
      D2_HIP_OK(hipMemcpy(h, P.stamps, stamp_n * 8, hipMemcpyDeviceToHost));
      FILE* f = fopen(stamp_path, "w");
      if (f) {
        for (size_t i = 0; i < stamp_n / 6; i++)
          if (h[6 * i])
            fprintf(f, "%zu %llu %llu %llu %llu %llu %llu\n", i, h[6 * i], h[6 * i + 1], h[6 * i + 2], h[6 * i + 3], h[6 * i + 4], h[6 * i + 5]);
        fclose(f);
      }
      free(h);
      (void)hipFree(P.stamps);
      P.stamps = nullptr;
    }
#endif
    static const bool no_small = getenv("D2AMD_TOPK_POOL_NO_SMALL") != nullptr;
    TkPool Q = w.pool;
    Q.no_small = no_small ? 1 : 0;
    hipLaunchKernelGGL(tk_pool1_kernel, segs, dim3(TK_P1_THREADS), 0, s, P, w.st, w.cand, w.kmax, Q);
    hipLaunchKernelGGL(tk_pool_kernel, dim3(pool_g_fit, in.N * in.L), block, 0, s, P, w.st, w.cand, w.kmax, Q);
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
  }
  if (w.kmax <= TK_RANK_MAX && !no_fused) {  // (the A/B switch also keeps the bitonic sort under test)
    static const bool no_epi = d2_prof_env("D2AMD_RPN_NO_FUSED_DECODE") != nullptr;  // A/B switch
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
    const size_t others = (size_t)(runs - 1) * run * 8;
    static const bool no_lds_merge = getenv("D2AMD_TOPK_MERGE_GLOBAL") != nullptr;  // A/B switch
    if (others <= 152 * 1024 && run % 1024 == 0 && !no_lds_merge) {
      {  // the opt-in for > 64 KB of dynamic LDS is PER DEVICE: one flag per device, under a mutex
        static std::mutex mu;
        static bool attr_set[64] = {};
        int dev = 0;
        D2_HIP_OK(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
          D2_HIP_OK(hipFuncSetAttribute((const void*)tk_merge_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
          if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
      }
      hipLaunchKernelGGL(tk_merge_lds_kernel, dim3(in.N * in.L, cdiv(w.kmax, 1024)), dim3(1024), others, s, P, w.st, w.cand,
                         w.kmax, run, sel);
    } else {
      hipLaunchKernelGGL(tk_merge_kernel, dim3(in.N * in.L, cdiv(w.kmax, 1024)), dim3(256), 0, s, P, w.st, w.cand, w.kmax,
                         run, sel);
    }
    D2_LAUNCH_OK();
  }
  return D2AMD_OK;
}

}  // namespace d2amd
