// subsample_labels on the device, whole batch, FIXED output shape, no host sync.
//   replaces  modeling/sampling.py:9-54 (subsample_labels: two nonzero() host syncs + two randperm sorts) as called by
//             proposal_generator/rpn.py:287-305 (RPN._subsample_labels: 268,569 anchor labels -> 256 per image, the
//             label vector rewritten in place to -1 / 0 / 1) and roi_heads/roi_heads.py:181-216 (_sample_proposals).
// Sampling rule (shared with label_sample.hip and oracle/sampling.py): one uniform key per element, the
// min(#positives, max_positives) SMALLEST keys among the positives (label not in {-1, bg_label}), then the
// min(#negatives, num_samples - sampled positives) smallest among the negatives (label == bg_label); ties towards
// the lower element index -- a uniform random subset of each group; the keys are the caller's (torch.rand), so the
// random stream stays torch's.  Output order inside a group: ascending (key, index).
//
// Pipeline (HBM-bound: every pass reads 4-5 B per element; nothing waits for the host):
//   1. ss_keys_kernel      label + key -> two candidate arrays (0 - key for the members of a group, NaN otherwise);
//                          RPN mode also fills the output label vector with -1
//   2. topk_select         the segmented radix select of topk.hip over 2 segments per image (positives: k =
//                          max_positives, negatives: k = num_samples): exact k-th key, ties by index, ordered result
//   3. ss_finish_kernel    sample sizes (sampling.py:42-47), index lists padded with -1, counts, and in RPN mode the
//                          scatter of 1 / 0 into the label vector (rpn.py:300-304)
#include <algorithm>
#include <cmath>

#include "common.h"
#include "topk.h"

namespace d2amd {

struct SsArgs {
  const void* labels;
  const float* keys;
  float *kpos, *kneg;  // [N][n] each (kpos null: no positive can be sampled)
  int8_t* labels_out;  // [N][n] or null
  long n;              // elements per image
  long total;          // N * n
  int64_t bg;
  int label_bytes;  // 1: int8, 8: int64
};

__global__ __launch_bounds__(256) void ss_keys_kernel(const SsArgs A) {
  const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= A.total) return;
  const float qnan = __builtin_nanf("");
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const long i = i0 + j;
    if (i >= A.total) break;
    const int64_t lab = A.label_bytes == 1 ? (int64_t)((const int8_t*)A.labels)[i] : ((const int64_t*)A.labels)[i];
    // 0 - key: the smallest key becomes the largest value; +0 and -0 both map to +0 (equal keys must stay ties)
    const float x = 0.f - A.keys[i];
    const bool neg = lab == A.bg, pos = lab != -1 && !neg;  // sampling.py:39-40
    if (A.kpos) A.kpos[i] = pos ? x : qnan;
    A.kneg[i] = neg ? x : qnan;
    if (A.labels_out) A.labels_out[i] = -1;  // rpn.py:300
  }
}

struct SsFinish {
  const uint32_t* sel;  // [N][ktot]
  const int* cnt;       // [N][L]
  int L, kpos, kneg, ktot;  // L = 2: level 0 positives, 1 negatives; L = 1: negatives only
  int num_samples, pos_cap, neg_cap;
  long n;
  int64_t *pos_idx, *neg_idx;  // [N][pos_cap] / [N][neg_cap] or null
  int* counts;                 // [N][2] or null
  int8_t* labels_out;          // or null
};

__global__ __launch_bounds__(256) void ss_finish_kernel(const SsFinish F) {
  const int img = blockIdx.x, tid = threadIdx.x;
  const uint32_t* sel = F.sel + (long)img * F.ktot;
  const int num_pos = F.L == 2 ? F.cnt[img * F.L] : 0;  // = min(#positives, max_positives): sampling.py:42-44
  const int have_neg = F.cnt[img * F.L + F.L - 1];      // = min(#negatives, num_samples)
  const int num_neg = min(have_neg, F.num_samples - num_pos);  // sampling.py:45-47
  for (int r = tid; r < F.pos_cap; r += 256) {
    const bool ok = r < num_pos;
    const long e = ok ? (long)sel[r] : -1;
    if (F.pos_idx) F.pos_idx[(long)img * F.pos_cap + r] = e;
    if (ok && F.labels_out) F.labels_out[(long)img * F.n + e] = 1;  // rpn.py:301
  }
  const uint32_t* nsel = sel + (F.L == 2 ? F.kpos : 0);
  for (int r = tid; r < F.neg_cap; r += 256) {
    const bool ok = r < num_neg;
    const long e = ok ? (long)nsel[r] : -1;
    if (F.neg_idx) F.neg_idx[(long)img * F.neg_cap + r] = e;
    if (ok && F.labels_out) F.labels_out[(long)img * F.n + e] = 0;  // rpn.py:302
  }
  if (tid == 0 && F.counts) {
    F.counts[2 * img] = num_pos;
    F.counts[2 * img + 1] = num_neg;
  }
}

// nothing can be drawn (no elements or num_samples 0): lists of -1, zero counts, every label -1
__global__ __launch_bounds__(256) void ss_empty_kernel(const SsFinish F, int N) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (F.pos_idx && t < (long)N * F.pos_cap) F.pos_idx[t] = -1;
  if (F.neg_idx && t < (long)N * F.neg_cap) F.neg_idx[t] = -1;
  if (F.counts && t < 2L * N) F.counts[t] = 0;
  if (F.labels_out && t < (long)N * F.n) F.labels_out[t] = -1;
}

static size_t ss_al(size_t x) { return (x + 255) / 256 * 256; }

struct SsLayout {
  TopkInput in;
  bool has_pos;
  size_t off_kpos, off_kneg, off_sel, off_cnt, off_tk, total;
};

static SsLayout ss_layout(int N, long n, int num_samples, int max_positives) {
  SsLayout y{};
  y.has_pos = max_positives > 0;
  const int kp = (int)std::min<long>(max_positives, n), kn = (int)std::min<long>(num_samples, n);
  TopkInput& in = y.in;
  in.N = N;
  in.L = y.has_pos ? 2 : 1;
  int l = 0, k = 0;
  if (y.has_pos) { in.size[l] = (int)n; in.stride[l] = n; in.k[l] = kp; in.koff[l] = k; k += kp; l++; }
  in.size[l] = (int)n; in.stride[l] = n; in.k[l] = kn; in.koff[l] = k; k += kn; l++;
  in.koff[l] = k;
  size_t off = 0;
  auto take = [&](size_t b) { const size_t r = off; off += ss_al(b); return r; };
  y.off_kpos = take(y.has_pos ? (size_t)N * n * 4 : 0);
  y.off_kneg = take((size_t)N * n * 4);
  y.off_sel = take((size_t)N * k * 4);
  y.off_cnt = take((size_t)N * in.L * 4);
  y.off_tk = off;
  y.total = off + topk_workspace_bytes(in);
  return y;
}

}  // namespace d2amd

using namespace d2amd;

static bool ss_sizes_ok(int N, long n, int num_samples, int max_positives) {
  return N >= 0 && n >= 0 && n <= 0x7fffffffL && (long)N * n <= (1L << 40) && num_samples >= 0 &&
      max_positives >= 0 && max_positives <= num_samples && num_samples <= TOPK_MAX_K;
}

extern "C" size_t d2amd_subsample_labels_workspace_bytes(int N, int64_t n, int num_samples, int max_positives) {
  if (!ss_sizes_ok(N, n, num_samples, max_positives) || N == 0 || n == 0 || num_samples == 0) return 256;
  return ss_layout(N, n, num_samples, max_positives).total + 256;
}

extern "C" int d2amd_subsample_labels(const void* labels, int label_bytes, int N, int64_t n, const float* keys,
                                      int num_samples, int max_positives, int64_t bg_label, int64_t* pos_idx_out,
                                      int64_t* neg_idx_out, int32_t* counts_out, int8_t* labels_out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  D2_CHECK_ARG(ss_sizes_ok(N, n, num_samples, max_positives),
               "subsample_labels: bad sizes (N %d, n %ld, num_samples %d <= %d, max_positives %d <= num_samples)", N,
               (long)n, num_samples, TOPK_MAX_K, max_positives);
  D2_CHECK_ARG(label_bytes == 1 || label_bytes == 8, "subsample_labels: labels must be int8 or int64");
  D2_CHECK_ARG(labels_out == nullptr || labels_out != labels || label_bytes == 1,
               "subsample_labels: an in-place label rewrite needs int8 labels");
  if (N == 0) return D2AMD_OK;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0 || num_samples == 0) {  // nothing to draw: empty lists, zero counts (labels_out: nothing / all -1)
    SsFinish F{};
    F.pos_cap = max_positives; F.neg_cap = num_samples; F.n = n;
    F.pos_idx = pos_idx_out; F.neg_idx = neg_idx_out; F.counts = counts_out; F.labels_out = labels_out;
    const long most = std::max<long>(std::max<long>((long)N * num_samples, 2L * N), labels_out ? (long)N * n : 0);
    hipLaunchKernelGGL(ss_empty_kernel, dim3(cdiv(most, 256)), dim3(256), 0, s, F, N);
    D2_LAUNCH_OK();
    return D2AMD_OK;
  }
  D2_CHECK_ARG(labels && keys, "subsample_labels: null input");
  const SsLayout y = ss_layout(N, n, num_samples, max_positives);
  if (workspace == nullptr || workspace_bytes < y.total) {
    set_error("subsample_labels: workspace too small (%zu < %zu)", workspace_bytes, y.total);
    return D2AMD_EWORKSPACE;
  }
  char* ws = (char*)workspace;
  SsArgs A{};
  A.labels = labels; A.keys = keys;
  A.kpos = y.has_pos ? (float*)(ws + y.off_kpos) : nullptr;
  A.kneg = (float*)(ws + y.off_kneg);
  A.labels_out = labels_out;
  A.n = n; A.total = (long)N * n; A.bg = bg_label; A.label_bytes = label_bytes;
  hipLaunchKernelGGL(ss_keys_kernel, dim3(cdiv(A.total, 1024)), dim3(256), 0, s, A);
  D2_LAUNCH_OK();
  TopkInput in = y.in;
  int l = 0;
  if (y.has_pos) in.ptr[l++] = A.kpos;
  in.ptr[l] = A.kneg;
  uint32_t* sel = (uint32_t*)(ws + y.off_sel);
  int* cnt = (int*)(ws + y.off_cnt);
  const int rc = topk_select(in, true, -__builtin_inff(), sel, cnt, ws + y.off_tk, workspace_bytes - y.off_tk, s);
  if (rc) return rc;
  SsFinish F{};
  F.sel = sel; F.cnt = cnt; F.L = in.L;
  F.kpos = y.has_pos ? in.k[0] : 0;
  F.kneg = in.k[in.L - 1];
  F.ktot = in.koff[in.L];
  F.num_samples = num_samples;
  F.pos_cap = max_positives; F.neg_cap = num_samples;
  F.n = n;
  F.pos_idx = pos_idx_out; F.neg_idx = neg_idx_out; F.counts = counts_out; F.labels_out = labels_out;
  hipLaunchKernelGGL(ss_finish_kernel, dim3(N), dim3(256), 0, s, F);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}
