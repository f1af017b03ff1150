// Proposal labelling + sampling for the ROI heads, whole batch, FIXED output size, no host sync.
//   replaces  roi_heads/roi_heads.py:219-295 (ROIHeads.label_and_sample_proposals): add_ground_truth_to_proposals
//             (proposal_generator/proposal_utils.py:138-205), pairwise_iou (structures/boxes.py:312-358), Matcher
//             (modeling/matcher.py:62-102, allow_low_quality_matches=False as roi_heads.py:176-180 builds it),
//             _sample_proposals (roi_heads.py:181-216) and subsample_labels (modeling/sampling.py:9-54)
// The reference runs this per image with data-dependent shapes: two nonzero() syncs inside subsample_labels, two
// randperm sorts, index gathers, two .item() reads for the logger.  The ROI heads of a captured training step cannot
// wait for the host, so this version has a fixed shape: S = batch_size_per_image rows per image -- the sampled
// foreground candidates first, then the background ones, then padding (class -1, index -1, zero box) -- and the two
// counts stay on the device.  The number of valid proposals of an image is read from device words (the NMS result
// buffer of find_top_rpn_proposals_fused), so nothing between the RPN and the poolers needs the host.
//
// Sampling rule = detectron2_amd/modeling/sampling.py: one uniform key per candidate, the num_pos smallest keys among
// the positives and the num_neg smallest among the negatives (a uniform random subset of each; ties by candidate
// index).  The keys come from the caller (torch.rand: the random stream stays torch's).  Output order inside a group:
// ascending key -- the reference's order is a random permutation, i.e. not defined.
//
// One 1,024-thread workgroup per image: candidates (<= LS_MAX, 4 per thread) keep their running best match in registers
// while the ground truth streams through LDS.  rank = number of same-group candidates with a smaller (key, index): the
// candidates are counting-sorted in LDS by (group, bucket of the key) -- 256 buckets per group, bucket = the key's
// position in [0, 1) -- so a candidate's rank is the number of candidates in the buckets below its own plus the few
// bucket mates it beats (any key distribution is handled: equal keys share a bucket, the worst case is one bucket
// holding everybody).  The first versions compared every candidate with every other one: 1,016^2 x ~6 instructions on
// ONE compute unit = 62 us (199 us with one dependent LDS load per compare), the longest op of the connected step
// (gpurun_out/r3h, r3j).  Matching arithmetic: matcher_core.h (bit-exact with d2amd_match_boxes; -ffp-contract=off).
#pragma clang fp contract(off)
#include <cstring>

#include "common.h"
#include "matcher_core.h"
#include "philox.h"

namespace d2amd {

constexpr int LS_THREADS = 1024, LS_PER = 4, LS_MAX = LS_THREADS * LS_PER, LS_GT_CHUNK = 512;
constexpr int LS_MAX_IMAGES = 16;  // per launch
constexpr int LS_BUCKETS = 256;    // key buckets per group of the rank's counting sort
static_assert(2 * LS_BUCKETS <= LS_THREADS && LS_MAX <= 65536, "label_sample bucket sort layout");

struct LsImage {
  const float4* props;
  const int64_t* limits;
  const float4* gt;
  const int64_t* gt_classes;
  const float* keys;
  int max_props, n_limits, num_gt, limit_stride;
  long key_base;  // keys drawn in the kernel (LsBatch::key_state): index of this image's first key in the call's draw
};

struct LsBatch {
  LsImage img[LS_MAX_IMAGES];
  MatchCfg cfg;
  int S, pos_max, append_gt;
  int64_t num_classes;
  float4* boxes;
  int64_t *classes, *gt_index, *index;
  int* counts;
  float *rois, *head_rois;  // optional: the rows in pooler format (image, x1, y1, x2, y2); the first head_rows of each image
  int64_t* head_classes;    // optional: the classes of those first head_rows rows, [image][head_rows] contiguous
  unsigned long long* key_state;  // optional: {seed, offset, ticket} of a device-resident generator (random_keys.hip) --
                                  // the keys are then DRAWN HERE (key c of image i = output key_base_i + c of
                                  // d2amd_uniform_keys at that state) and the last workgroup advances the offset
  int key_images;                 // workgroups of the whole call (the ticket's target)
  int head_rows, image0;    // image0: batch index of this launch's first image
};

// row `slot` of image `image` in pooler format (convert_boxes_to_pooler_format, poolers.py:62-104): what the box pooler
// (all rows) and the mask pooler (the first head_rows rows of every image: the positives come first) take as they are --
// no conversion launch between the sampler and the poolers
__device__ __forceinline__ void ls_write_rois(const LsBatch& B, int image, int slot, float4 b) {
  const float fi = (float)(B.image0 + image);
  if (B.rois) {
    float* r = B.rois + ((long)(B.image0 + image) * B.S + slot) * 5;
    r[0] = fi; r[1] = b.x; r[2] = b.y; r[3] = b.z; r[4] = b.w;
  }
  if (B.head_rois && slot < B.head_rows) {
    float* r = B.head_rois + ((long)(B.image0 + image) * B.head_rows + slot) * 5;
    r[0] = fi; r[1] = b.x; r[2] = b.y; r[3] = b.z; r[4] = b.w;
  }
}

__global__ __launch_bounds__(LS_THREADS) void label_sample_kernel(const LsBatch B) {
  __shared__ float s_key[LS_MAX];       // key of candidate c
  __shared__ uint16_t s_order[LS_MAX];  // candidate ids in (group, bucket) order
  __shared__ int s_bcnt[2 * LS_BUCKETS], s_bstart[2 * LS_BUCKETS + 1], s_wsum[LS_THREADS / 64];
  __shared__ float4 s_gt[LS_GT_CHUNK];
  __shared__ float s_garea[LS_GT_CHUNK];
  __shared__ int s_n, s_gt_nan;
  const int tid = threadIdx.x, image = blockIdx.x;
  // (constant indices only into the kernel-argument struct: a dynamic one makes the compiler copy it to scratch)
  LsImage I = B.img[0];
#pragma unroll
  for (int q = 1; q < LS_MAX_IMAGES; q++)
    if (q == image) I = B.img[q];
  if (tid == 0) {
    long n = I.max_props;
    for (int l = 0; l < I.n_limits; l++) {
      const long v = (long)I.limits[(long)l * I.limit_stride];
      n = v < n ? v : n;
    }
    s_n = (int)(n < 0 ? 0 : n);
  }
  __syncthreads();
  const int n = s_n, G = I.num_gt;
  unsigned long long kseed = 0ull, koffset = 0ull;  // (uniform) keys drawn here: the generator's state at this call
  if (B.key_state) { kseed = B.key_state[0]; koffset = B.key_state[1]; }
  const int ncand = n + (B.append_gt ? G : 0);
  // ---- candidates of this thread: c = tid + k * LS_THREADS
  float4 box[LS_PER];
  float area[LS_PER], best[LS_PER];
  int besti[LS_PER];
  bool have[LS_PER], bnan[LS_PER];
#pragma unroll
  for (int k = 0; k < LS_PER; k++) {
    const int c = tid + k * LS_THREADS;
    box[k] = make_float4(0, 0, 0, 0);
    if (c < n) box[k] = I.props[c];
    else if (c < ncand) box[k] = I.gt[c - n];
    area[k] = (box[k].z - box[k].x) * (box[k].w - box[k].y);
    bnan[k] = mt_has_nan(box[k]);
    best[k] = 0.f; besti[k] = 0; have[k] = false;
  }
  // ---- pairwise_iou + Matcher: first maximal ground truth per candidate (torch.max over dim 0; NaN is maximal)
  for (int m0 = 0; m0 < G; m0 += LS_GT_CHUNK) {
    const int mc = min(LS_GT_CHUNK, G - m0);
    __syncthreads();
    if (tid == 0) s_gt_nan = 0;
    __syncthreads();
    for (int i = tid; i < mc; i += LS_THREADS) {
      const float4 a = I.gt[m0 + i];
      s_gt[i] = a;
      s_garea[i] = (a.z - a.x) * (a.w - a.y);
      if (mt_has_nan(a)) s_gt_nan = 1;
    }
    __syncthreads();
    const bool gnan = s_gt_nan != 0;
    for (int i = 0; i < mc; i++) {
      const float4 g = s_gt[i];
      const float ga = s_garea[i];
#pragma unroll
      for (int k = 0; k < LS_PER; k++) {
        if (tid + k * LS_THREADS >= ncand) continue;
        const float v = (gnan || bnan[k]) ? mt_iou(g, ga, box[k]) : mt_iou_fast(g, ga, box[k], area[k]);
        const bool better = !have[k] || (v > best[k]) || (v != v && best[k] == best[k]);
        if (better) { best[k] = v; besti[k] = m0 + i; have[k] = true; }
      }
    }
  }
  // ---- _sample_proposals: class per candidate, group, key; counting sort by (group, key bucket)
  if (tid < 2 * LS_BUCKETS) s_bcnt[tid] = 0;
  __syncthreads();
  int64_t cls[LS_PER];
  float key[LS_PER];
  int bkt[LS_PER], at[LS_PER];  // bucket (group * LS_BUCKETS + key bucket; -1: not sampled from), arrival in the bucket
#pragma unroll
  for (int k = 0; k < LS_PER; k++) {
    const int c = tid + k * LS_THREADS;
    int grp = 2;
    cls[k] = -1;
    key[k] = 0.f;
    bkt[k] = -1;
    at[k] = 0;
    if (c < ncand) {
      if (G > 0) {
        const int8_t lab = mt_label(best[k], B.cfg);
        cls[k] = lab == 0 ? B.num_classes : (lab == -1 ? (int64_t)-1 : I.gt_classes[besti[k]]);
      } else {
        cls[k] = B.num_classes;  // roi_heads.py:207: no ground truth -> every proposal is background
      }
      grp = cls[k] == B.num_classes ? 1 : (cls[k] == -1 ? 2 : 0);  // sampling.py:39-40
      const int kc = c < n ? c : I.max_props + (c - n);
      key[k] = B.key_state ? philox_key(kseed, koffset, I.key_base + kc) : I.keys[kc];
      s_key[c] = key[k];
      if (grp < 2) {
        // monotone in the key (equal keys -> equal bucket; anything outside [0, 1) lands in the end buckets)
        const float f = key[k] * (float)LS_BUCKETS;
        const int b = f >= (float)(LS_BUCKETS - 1) ? LS_BUCKETS - 1 : (f > 0.f ? (int)f : 0);
        bkt[k] = grp * LS_BUCKETS + b;
        at[k] = atomicAdd(&s_bcnt[bkt[k]], 1);
      }
    }
  }
  __syncthreads();
  {  // exclusive scan of the 512 bucket counts (threads 0 .. 511: wave scans + the 8 wave totals)
    const int lane = tid & 63, wv = tid >> 6;
    const int v = tid < 2 * LS_BUCKETS ? s_bcnt[tid] : 0;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) s_wsum[wv] = x;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < wv; q++) base += s_wsum[q];
    if (tid < 2 * LS_BUCKETS) s_bstart[tid] = base + x - v;
    if (tid == 2 * LS_BUCKETS - 1) s_bstart[2 * LS_BUCKETS] = base + x;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < LS_PER; k++)
    if (bkt[k] >= 0) s_order[s_bstart[bkt[k]] + at[k]] = (uint16_t)(tid + k * LS_THREADS);
  __syncthreads();
  const int n_pos_all = s_bstart[LS_BUCKETS], n_neg_all = s_bstart[2 * LS_BUCKETS] - n_pos_all;
  const int num_pos = min(n_pos_all, B.pos_max);      // sampling.py:42-44
  const int num_neg = min(n_neg_all, B.S - num_pos);  // sampling.py:45-47
  // ---- rank inside the group by (key, index); the selected ones go to their slot
  float4* ob = B.boxes + (long)image * B.S;
  int64_t* oc = B.classes + (long)image * B.S;
  int64_t* og = B.gt_index + (long)image * B.S;
  int64_t* oi = B.index + (long)image * B.S;
#pragma unroll
  for (int k = 0; k < LS_PER; k++) {
    if (bkt[k] < 0) continue;
    const int c = tid + k * LS_THREADS;
    const bool pos = bkt[k] < LS_BUCKETS;
    const int want = pos ? num_pos : num_neg;
    const int lo = s_bstart[bkt[k]], hi = s_bstart[bkt[k] + 1];
    int rank = lo - (pos ? 0 : n_pos_all);  // same-group candidates in the buckets below
    if (rank >= want) continue;
    for (int p2 = lo; p2 < hi; p2++) {  // bucket mates: a handful for uniform keys
      const int j = s_order[p2];
      const float kj = s_key[j];
      rank += (kj < key[k] || (kj == key[k] && j < c)) ? 1 : 0;
    }
    if (rank >= want) continue;
    const int slot = (pos ? 0 : num_pos) + rank;
    ls_write_rois(B, image, slot, box[k]);
    ob[slot] = box[k];
    oc[slot] = cls[k];
    if (B.head_classes && slot < B.head_rows) B.head_classes[(long)(B.image0 + image) * B.head_rows + slot] = cls[k];
    og[slot] = besti[k];
    oi[slot] = c;
  }
  for (int t = num_pos + num_neg + tid; t < B.S; t += LS_THREADS) {  // padding
    ls_write_rois(B, image, t, make_float4(0, 0, 0, 0));
    ob[t] = make_float4(0, 0, 0, 0);
    oc[t] = -1;
    if (B.head_classes && t < B.head_rows) B.head_classes[(long)(B.image0 + image) * B.head_rows + t] = -1;
    og[t] = 0;
    oi[t] = -1;
  }
  if (tid == 0) {
    B.counts[2 * image] = num_pos;
    B.counts[2 * image + 1] = num_pos + num_neg;
  }
  // the last workgroup of the CALL to finish advances the generator (every other one has read the offset: it finished)
  if (B.key_state && tid == 0) {
    const unsigned long long t = atomicAdd(&B.key_state[2], 1ull);
    if (t == (unsigned long long)B.key_images - 1ull) {
      B.key_state[1] = koffset + 1ull;
      B.key_state[2] = 0ull;
    }
  }
}

}  // namespace d2amd

using namespace d2amd;

extern "C" int d2amd_label_and_sample_max_candidates(void) { return LS_MAX; }

extern "C" int d2amd_label_and_sample_proposals(const d2amd_sample_image* images, int count, const float* thresholds,
                                                const int8_t* labels, int T, int batch_size_per_image,
                                                int max_positives, int64_t num_classes, int append_gt,
                                                float* boxes_out, int64_t* classes_out, int64_t* gt_index_out,
                                                int64_t* index_out, int32_t* counts_out, float* rois_out,
                                                float* head_rois_out, int64_t* head_classes_out, int head_rows,
                                                uint64_t* key_state, void* stream) {
  D2_CHECK_ARG(count >= 0 && (count == 0 || images != nullptr), "label_and_sample: bad image list");
  D2_CHECK_ARG(T >= 0 && T <= D2AMD_MATCHER_MAX_THRESHOLDS && labels != nullptr && (T == 0 || thresholds != nullptr),
               "label_and_sample: %d thresholds (max %d)", T, D2AMD_MATCHER_MAX_THRESHOLDS);
  D2_CHECK_ARG(batch_size_per_image > 0 && max_positives >= 0 && max_positives <= batch_size_per_image &&
                   num_classes >= 0,
               "label_and_sample: bad sampling parameters");
  if (count == 0) return D2AMD_OK;
  D2_CHECK_ARG(boxes_out && classes_out && gt_index_out && index_out && counts_out, "label_and_sample: null output");
  LsBatch B;
  std::memset(&B, 0, sizeof(B));
  B.cfg.T = T;
  for (int k = 0; k < T; k++) {
    D2_CHECK_ARG(thresholds[k] > 0.f && (k == 0 || thresholds[k - 1] <= thresholds[k]),
                 "label_and_sample: thresholds must be positive and ascending");
    B.cfg.thr[k] = thresholds[k];
  }
  for (int k = 0; k <= T; k++) {
    D2_CHECK_ARG(labels[k] >= -1 && labels[k] <= 1, "label_and_sample: labels must be in {-1, 0, 1}");
    B.cfg.lab[k] = labels[k];
  }
  B.S = batch_size_per_image;
  B.pos_max = max_positives;  // the caller's int(num_samples * positive_fraction), sampling.py:42
  B.append_gt = append_gt != 0;
  B.num_classes = num_classes;
  D2_CHECK_ARG(head_rows >= 0 && head_rows <= batch_size_per_image, "label_and_sample: bad head_rows");
  B.rois = rois_out;
  B.head_rois = head_rows > 0 ? head_rois_out : nullptr;
  B.head_classes = head_rows > 0 ? head_classes_out : nullptr;
  B.head_rows = head_rows;
  B.key_state = (unsigned long long*)key_state;
  B.key_images = count;
  long key_base = 0;
  // EVERY image is validated before the first launch: a call that fails must not have launched a partial set of
  // workgroups (with key_state set they would leave the arrival ticket non-zero, and the generator's offset would stop
  // advancing for every later call -- the same keys redrawn silently)
  for (int i = 0; i < count; i++) {
    const d2amd_sample_image& s = images[i];
    D2_CHECK_ARG(s.max_proposals >= 0 && s.num_gt >= 0 && s.n_limits >= 0 && s.n_limits <= 4,
                 "label_and_sample: image %d: bad sizes", i);
    if ((long)s.max_proposals + (append_gt ? s.num_gt : 0) > LS_MAX) {
      set_error("label_and_sample: image %d has %d + %d candidates (max %d)", i, s.max_proposals, s.num_gt, LS_MAX);
      return D2AMD_EUNSUPPORTED;
    }
    D2_CHECK_ARG((s.max_proposals == 0 || s.proposals) && (s.n_limits == 0 || s.limits) &&
                     (s.num_gt == 0 || (s.gt_boxes && s.gt_classes)) &&
                     (s.max_proposals + s.num_gt == 0 || s.keys || key_state),
                 "label_and_sample: image %d: null pointer", i);
  }
  for (int i0 = 0; i0 < count; i0 += LS_MAX_IMAGES) {
    const int c = count - i0 < LS_MAX_IMAGES ? count - i0 : LS_MAX_IMAGES;
    for (int i = 0; i < c; i++) {
      const d2amd_sample_image& s = images[i0 + i];
      LsImage& I = B.img[i];
      I.props = (const float4*)s.proposals;
      I.limits = s.limits;
      I.gt = (const float4*)s.gt_boxes;
      I.gt_classes = s.gt_classes;
      I.keys = s.keys;
      I.max_props = s.max_proposals;
      I.n_limits = s.n_limits;
      I.num_gt = s.num_gt;
      I.limit_stride = s.limit_stride > 1 ? s.limit_stride : 1;
      I.key_base = key_base;
      key_base += (long)s.max_proposals + s.num_gt;
    }
    B.boxes = (float4*)boxes_out + (long)i0 * B.S;
    B.classes = classes_out + (long)i0 * B.S;
    B.gt_index = gt_index_out + (long)i0 * B.S;
    B.index = index_out + (long)i0 * B.S;
    B.counts = counts_out + 2 * i0;
    B.image0 = i0;
    hipLaunchKernelGGL(label_sample_kernel, dim3(c), dim3(LS_THREADS), 0, (hipStream_t)stream, B);
    D2_LAUNCH_OK();
  }
  return D2AMD_OK;
}
