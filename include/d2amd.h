/*
 * d2amd.h -- C ABI of libd2amd.so: the MI355X (gfx950) implementation of Detectron2's
 * per-image detection hot path.  This is the drop-in boundary: every entry point below is
 * what a binding of the reference's native-op surface would call.  The reference has no C
 * ABI of its own -- its boundary is a pybind11 module + TORCH_LIBRARY ops
 * (detectron2/layers/csrc/vision.cpp:81-120); each function cites the reference entry it
 * replaces.  INTEGRATION.md shows the Python (ctypes / torch.library) stub that binds them.
 *
 * Conventions
 *   - plain C, no torch/ATen types: device pointers + sizes + a HIP stream (hipStream_t
 *     passed as void*; NULL = the legacy default stream).
 *   - every pointer is a DEVICE pointer on the current HIP device unless it says (host).
 *   - calls are asynchronous on `stream`; nothing synchronises the device.
 *   - return 0 on success, a negative D2AMD_E* code otherwise; d2amd_last_error() (host,
 *     thread-local) describes the failure.  The reference raises RuntimeError through
 *     TORCH_CHECK/AT_ERROR (SURVEY 8b "Errors"); the Python layer turns codes into that.
 *   - tensors are dense; `layout` says whether a 4-D feature tensor is NCHW-contiguous
 *     (the reference's layout) or NHWC-contiguous (torch.channels_last).
 *   - floating-point I/O dtype is selected by `dtype`; all accumulation is fp32.
 */
#ifndef D2AMD_H_
#define D2AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D2AMD_OK 0
#define D2AMD_EINVAL (-1)    /* bad argument / shape (reference: TORCH_CHECK failures) */
#define D2AMD_ELAUNCH (-2)   /* HIP launch / runtime error */
#define D2AMD_EWORKSPACE (-3)/* workspace too small */
#define D2AMD_EUNSUPPORTED (-4)

enum d2amd_dtype { D2AMD_F32 = 0, D2AMD_F16 = 1, D2AMD_BF16 = 2 };
enum d2amd_layout { D2AMD_NCHW = 0, D2AMD_NHWC = 1 };
enum d2amd_iou_mode { D2AMD_IOU = 0, D2AMD_IOA = 1, D2AMD_INTERSECTION = 2 };

/* ---- introspection: vision.cpp:16-79 get_cuda_version / has_cuda / get_compiler_version */
const char* d2amd_version(void);          /* library version string (host) */
const char* d2amd_compiler_version(void); /* "clang x.y.z" like get_compiler_version() */
const char* d2amd_hip_version(void);      /* "HIP x.y" like get_cuda_version() under WITH_HIP */
const char* d2amd_last_error(void);       /* last error message of this thread (host) */

/* ---- kernel timing aid (measurement only: bench.py's `roofline` needs the duration of ONE kernel inside a multi-
 * kernel op, on the stream it is launched on -- some launches run on library-owned side streams that events
 * recorded by the caller cannot see).  For every selected kernel name, HIP events are recorded on the LAUNCH stream
 * right before / after each launch; d2amd_timing_read waits for them and returns the summed duration and the number
 * of launches since the last select / enable.  Names: "pool_bwd_staged_r7" / "_r14" (tile-gather backward of the
 * fused pooler, pooled size <= 7 / larger), "pool_fwd_r7" / "_r14", "dcn_fwd", "dcn_bwd_data", "dcn_bwd_gather", "dcn_bwd_weight",
 * "nms_mask", "nms_reduce" (axis-aligned and rotated), "paste_masks" (zero fill + region kernel), "iou_rotated",
 * "roi_align_rot_fwd", "roi_align_rot_bwd" (zero fill + scatter + conversion).  Off by default (no events, no cost). */
void d2amd_timing_select(const char* names_csv); /* comma-separated kernel names; NULL or "" = none */
void d2amd_timing_enable(int mask); /* legacy: bit 0 pool_bwd_*_r7 fine, 1 coarse, 2 / 3 the same for _r14; 0 = off */
int d2amd_timing_read(const char* kernel, double* total_ms, int* launches);

/* ---- layout conversion: batched 2-D transpose [batch][rows][cols] -> [batch][cols][rows] of 2- or 4-byte elements.
 * NCHW -> NHWC: rows = C, cols = H*W; NHWC -> NCHW: rows = H*W, cols = C.  What the Python layer uses to bring an
 * unmodified (NCHW) model's feature maps, pooled results and gradients in and out of the NHWC kernels; the reference
 * has no counterpart (torchvision's ops are NCHW throughout). */
int d2amd_transpose_batched(const void* src, void* dst, int batch, int rows, int cols, int element_size, void* stream);
/* The same for up to 8 tensors of one batch size and element size in ONE launch (the FPN levels of an NCHW model on the
 * way in, their gradients on the way out): src[t] [batch][rows[t]][cols[t]] -> dst[t] [batch][cols[t]][rows[t]]. */
int d2amd_transpose_multi(const void* const* src, void* const* dst, const int* rows, const int* cols, int count, int batch,
                          int element_size, void* stream);

/* ---- ROIAlign (axis-aligned).  Replaces torchvision.ops.roi_align as called from
 * detectron2/layers/roi_align.py:58-65 (forward) and its autograd backward.
 * input  [N,C,H,W] `dtype`, `layout`; rois [K,5] fp32 (b, x1, y1, x2, y2);
 * output [K,C,PH,PW] `dtype`, same `layout` convention as input (NCHW or NHWC dense). */
int d2amd_roi_align_forward(const void* input, const float* rois, void* output, int N, int C,
                            int H, int W, int K, int pooled_h, int pooled_w,
                            float spatial_scale, int sampling_ratio, int aligned, int dtype,
                            int layout, void* stream);
/* grad_input [N,C,H,W] is fully overwritten (zero-filled then accumulated).  For 16-bit
 * dtypes on the NCHW path `workspace` must hold N*C*H*W floats (fp32 accumulation, converted
 * once); the NHWC path (pooled size <= 32) needs d2amd_roi_pooler_workspace_bytes(K) instead. */
int d2amd_roi_align_backward(const void* grad_output, const float* rois, void* grad_input,
                             int N, int C, int H, int W, int K, int pooled_h, int pooled_w,
                             float spatial_scale, int sampling_ratio, int aligned, int dtype,
                             int layout, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused multi-level ROI pooler.  Replaces ROIPooler.forward
 * (detectron2/modeling/poolers.py:206-263: assign_boxes_to_levels -> per-level nonzero /
 * ROIAlign / index_put_) and its autograd backward with ONE launch per direction.
 * Level of a box = clamp(floor(canonical_level + log2(sqrt(area)/canonical_box_size + 1e-8)),
 * min_level, max_level) - min_level, evaluated in fp32 exactly like poolers.py:51-59 (ignored
 * when num_levels == 1).  rois [K,5] fp32 (batch index, x1, y1, x2, y2) in image coordinates;
 * inputs[l] is [N,C,H[l],W[l]] in `dtype`/`layout`; output [K,C,pooled_h,pooled_w], same layout
 * convention.  `inputs` / `grad_inputs` are HOST arrays of num_levels device pointers.
 * Backward (NHWC only) overwrites every element of every grad_inputs[l]; it uses no atomics, no
 * fp32 staging buffer (only d2amd_roi_pooler_workspace_bytes(K) of per-ROI records) and is
 * deterministic.  d2amd_roi_pooler_supported tells whether the fused
 * kernels serve a configuration (pooled size <= 32; backward: NHWC); otherwise loop over the
 * levels with d2amd_roi_align_forward/backward as the reference does. */
#define D2AMD_POOLER_MAX_LEVELS 8
#define D2AMD_POOLER_MAX_IMAGES 64
/* convert_boxes_to_pooler_format (poolers.py:62-104) in one launch and without the host sync of
 * torch.repeat_interleave: boxes [K,width] (width 4: xyxy, 5: rotated), the K boxes of image 0 first,
 * then image 1, ...; counts (host) [num_images] boxes per image -> rois [K,width+1] with the batch
 * index in column 0. */
int d2amd_boxes_to_rois(const float* boxes, const int* counts, int num_images, int width, float* rois,
                        void* stream);
typedef struct d2amd_pooler_params {
  int num_levels, N, C;
  int H[D2AMD_POOLER_MAX_LEVELS], W[D2AMD_POOLER_MAX_LEVELS];
  float spatial_scale[D2AMD_POOLER_MAX_LEVELS];
  int pooled_h, pooled_w, sampling_ratio, aligned, dtype, layout;
  int min_level, max_level, canonical_level;
  float canonical_box_size;
  int roi_rounding;  /* 0: fp32 ROIs as given (default).  1: strict reference parity for 16-bit features -- the ROI
                      * coordinates are rounded to the FEATURE dtype, as layers/roi_align.py:60 casts them
                      * (`rois.to(dtype=input.dtype)`), AFTER the level assignment, which ROIPooler.forward
                      * (poolers.py:240-247) makes from the unrounded boxes. */
} d2amd_pooler_params;
int d2amd_roi_pooler_supported(const d2amd_pooler_params* p, int backward);
int d2amd_roi_pooler_forward(const d2amd_pooler_params* p, const void* const* inputs, const float* rois,
                             void* output, int K, void* stream);
/* The same for box lists that were never concatenated (ROIPooler.forward's `box_lists`, poolers.py:206-263):
 * boxes (host array of num_images device pointers, each [counts[i], 4] fp32 xyxy, 16-byte aligned), counts (host).
 * Writes rois_out [K,5] (convert_boxes_to_pooler_format, needed again by the backward) and pools from it: one call
 * instead of torch.cat + d2amd_boxes_to_rois + d2amd_roi_pooler_forward. */
int d2amd_roi_pooler_forward_box_lists(const d2amd_pooler_params* p, const void* const* inputs,
                                       const float* const* boxes, const int* counts, int num_images,
                                       float* rois_out, void* output, void* stream);
/* Two poolers of the SAME feature maps in ONE launch (Mask R-CNN's box head 7x7 and mask head 14x14 poolers,
 * roi_heads.py:780-846): output1 / output2 are bit for bit what d2amd_roi_pooler_forward(p1 ...) and (p2 ...) write; the
 * second pooler's workgroups fill the slots the first one's free instead of starting behind its last (largest) ROI.
 * D2AMD_EUNSUPPORTED -- nothing launched -- outside NHWC with 16-byte channel vectors, for K1 or K2 == 0 or different
 * level rules / scales / sampling: the caller issues the two calls. */
int d2amd_roi_pooler_forward_pair(const d2amd_pooler_params* p1, const void* const* inputs, const float* rois1,
                                  void* output1, int K1, const d2amd_pooler_params* p2, const float* rois2,
                                  void* output2, int K2, void* stream);
/* The same for box lists that were never concatenated (as d2amd_roi_pooler_forward_box_lists): one conversion launch
 * for both lists (rois1_out [K1,5], rois2_out [K2,5]: the backward needs them), then the paired forward -- or the two
 * plain forwards where it does not apply: both outputs are always produced. */
int d2amd_roi_pooler_forward_pair_box_lists(const d2amd_pooler_params* p1, const void* const* inputs,
                                            const float* const* boxes1, const int* counts1, float* rois1_out,
                                            void* output1, const d2amd_pooler_params* p2, const float* const* boxes2,
                                            const int* counts2, float* rois2_out, void* output2, int num_images,
                                            void* stream);
/* Both forwards with a ROI PROCESSING ORDER: the ROIs are sorted by (level, image, Morton code of their centre's 8-px
 * tile) by one small launch (which also converts the box lists), and each XCD pools one contiguous range of that
 * order -- neighbours in the feature map share that XCD's L2 instead of being fetched by all eight.  Results are
 * identical (row k of the output is ROI k).  workspace: d2amd_roi_pooler_forward_workspace_bytes(K); without it, for
 * K > 4096 or more than 32 images the list order is used. */
size_t d2amd_roi_pooler_forward_workspace_bytes(int K);
int d2amd_roi_pooler_forward_ordered(const d2amd_pooler_params* p, const void* const* inputs, const float* rois,
                                     void* output, int K, void* workspace, size_t workspace_bytes, void* stream);
int d2amd_roi_pooler_forward_box_lists_ordered(const d2amd_pooler_params* p, const void* const* inputs,
                                               const float* const* boxes, const int* counts, int num_images,
                                               float* rois_out, void* output, void* workspace,
                                               size_t workspace_bytes, void* stream);
size_t d2amd_roi_pooler_workspace_bytes(int K); /* backward, minimum: per-ROI records (48 B each) */
/* backward, recommended: records + per-tile ROI lists (one wave per 8x8 tile bins the ROIs once per call;
 * with the minimum size every tile workgroup scans all K records itself, ~2 us per 512 records and tile) */
size_t d2amd_roi_pooler_backward_workspace_bytes(const d2amd_pooler_params* p, int K);
int d2amd_roi_pooler_backward(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                              void* const* grad_inputs, int K, void* workspace, size_t workspace_bytes,
                              void* stream);
/* Same, ADDING to grad_inputs instead of overwriting them: grad_inputs already hold the gradient another pooler of
 * the same feature maps produced (Mask R-CNN pools p2..p5 twice per iteration: box head 7x7, mask head 14x14 --
 * roi_heads.py:780-846; autograd would sum the two dense gradients with one elementwise kernel per level, 3 x the
 * feature bytes of extra traffic).  Every element = round(held + round(own)) in the I/O dtype -- what that sum
 * gives; tiles no ROI touches are neither read nor written.  D2AMD_EUNSUPPORTED when the staged tile gather cannot
 * serve the configuration (the caller then uses d2amd_roi_pooler_backward into a fresh buffer and adds). */
int d2amd_roi_pooler_backward_accumulate(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                                         void* const* grad_inputs, int K, void* workspace, size_t workspace_bytes,
                                         void* stream);
/* Both poolers of one set of feature maps in ONE pass over the gradient's tiles (Mask R-CNN: the box head's 7x7
 * pooler and the mask head's 14x14 one, roi_heads.py:780-846; modeling/poolers.py:206-263 twice): their ROIs are binned
 * together and a tile's list is ONE contraction over both poolers' bins into the same fp32 accumulators -- one queue
 * take, one prologue and ONE write per tile, where d2amd_roi_pooler_backward + d2amd_roi_pooler_backward_accumulate pay
 * each of them twice and read the tile back to add to it.  Result: every element = round(sum1 + sum2) in the I/O dtype
 * (fp32 sums; the two-call sequence gives round(round(sum1) + round(sum2)): the two differ by those extra roundings,
 * this one is the closer to the fp32 value).  workspace: d2amd_roi_pooler_backward_pair_workspace_bytes(p1, K1, K2).
 * D2AMD_EUNSUPPORTED -- nothing launched -- outside 16-bit NHWC (C a multiple of 32), pooled sizes <= 32 (either pooler
 * may come first), the same level rule, scales, sampling ratio and alignment, K1, K2 > 0: the caller issues the two
 * calls. */
size_t d2amd_roi_pooler_backward_pair_workspace_bytes(const d2amd_pooler_params* p1, int K1, int K2);
int d2amd_roi_pooler_backward_pair(const d2amd_pooler_params* p1, const void* grad_output1, const float* rois1, int K1,
                                   const d2amd_pooler_params* p2, const void* grad_output2, const float* rois2, int K2,
                                   void* const* grad_inputs, void* workspace, size_t workspace_bytes, void* stream);
/* The paired backward in two calls: phase 1 bins both ROI sets (reads the rois, writes the workspace and zero-fills
 * the tiles of grad_inputs no ROI touches: the gradient tensors must exist, no gradient value is needed -- it can run
 * on another stream beside the poolers' forward; grad_outputN: any pointers of the later ones' alignment class); phase 2,
 * same arguments and workspace, is the tile gather alone; phase 5 = the one-call backward without its records launch, behind
 * d2amd_roi_pooler_forward_pair_records (above).  D2AMD_EUNSUPPORTED as for the one-call entry. */
/* The paired forward that ALSO prepares the paired backward of the same ROIs: the per-ROI records and the reset of the
 * work queues -- the first launch of d2amd_roi_pooler_backward_pair -- are done by the forward's workgroups into
 * bwd_workspace (d2amd_roi_pooler_backward_pair_workspace_bytes(p1, K1, K2) bytes; the caller keeps it, and the ROI
 * tensors, unchanged until the backward).  *records_written = 1: run the backward as
 * d2amd_roi_pooler_backward_pair_phase(..., phase 5) with that workspace (it starts with the tile lists); 0: the workspace
 * was not touched (a configuration outside the paired tile gather): d2amd_roi_pooler_backward_pair as usual.  Outputs and
 * return codes are d2amd_roi_pooler_forward_pair's.  (modeling/poolers.py:206-263 twice + the head of their backward.) */
int d2amd_roi_pooler_forward_pair_records(const d2amd_pooler_params* p1, const void* const* inputs, const float* rois1,
                                          void* output1, int K1, const d2amd_pooler_params* p2, const float* rois2,
                                          void* output2, int K2, void* bwd_workspace, size_t bwd_workspace_bytes,
                                          int* records_written, void* stream);
int d2amd_roi_pooler_backward_pair_phase(const d2amd_pooler_params* p1, const void* grad_output1, const float* rois1,
                                         int K1, const d2amd_pooler_params* p2, const void* grad_output2,
                                         const float* rois2, int K2, void* const* grad_inputs, void* workspace,
                                         size_t workspace_bytes, int phase, void* stream);
/* The backward in two calls: phase 1 bins the ROIs (per-ROI records, per-tile ROI lists, work queues: reads `rois`
 * only, writes the workspace only -- it may run on another stream, long before the gradient exists: beside the
 * pooler's forward); a later call with the same arguments and workspace runs the gather: phase 2 ADDS to grad_inputs
 * (= d2amd_roi_pooler_backward_accumulate), phase 3 WRITES it (= d2amd_roi_pooler_backward: tiles no ROI touches are
 * zero-filled by a small launch of their own).  In phase 1 grad_output / grad_inputs may be any pointers of the same
 * alignment class as the later ones (only `& 15` is looked at).  D2AMD_EUNSUPPORTED as for the accumulate entry. */
int d2amd_roi_pooler_backward_phase(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                                    void* const* grad_inputs, int K, void* workspace, size_t workspace_bytes,
                                    int phase, void* stream);

/* ---- ROIAlignRotated.  Replaces torch.ops.detectron2.roi_align_rotated_forward/backward
 * (vision.cpp:118-119; csrc/ROIAlignRotated/ROIAlignRotated.h:50-113).
 * rois [K,6] fp32 (b, cx, cy, w, h, angle_degrees).  Negative w/h is reported through
 * `status` (device int32, set non-zero; mirrors the CPU AT_ASSERTM at
 * ROIAlignRotated_cpu.cpp:236-238); may be NULL. */
int d2amd_roi_align_rotated_forward(const void* input, const float* rois, void* output, int N,
                                    int C, int H, int W, int K, int pooled_h, int pooled_w,
                                    float spatial_scale, int sampling_ratio, int dtype,
                                    int layout, int* status, void* stream);
int d2amd_roi_align_rotated_backward(const void* grad_output, const float* rois,
                                     void* grad_input, int N, int C, int H, int W, int K,
                                     int pooled_h, int pooled_w, float spatial_scale,
                                     int sampling_ratio, int dtype, int layout, void* workspace,
                                     size_t workspace_bytes, void* stream);

/* ---- ROIAlign / ROIAlignRotated in DOUBLE precision.  The reference instantiates its ops for double
 * (ROIAlignRotated_cuda.cu:360,421: AT_DISPATCH_FLOATING_TYPES_AND_HALF; torchvision.ops.roi_align likewise) and
 * its tests gradcheck them there (tests/layers/test_roi_align_rotated.py:107-172).  input / output / rois are
 * all double (the reference casts the ROIs to the input dtype, layers/roi_align.py:60), NCHW; rotated != 0:
 * rois [K,6] and `aligned` is ignored; `status` as d2amd_roi_align_rotated_forward.  A correctness path
 * (one thread per output element), not a tuned one. */
int d2amd_roi_align_f64_forward(const double* input, const double* rois, double* output, int N, int C,
                                int H, int W, int K, int pooled_h, int pooled_w, double spatial_scale,
                                int sampling_ratio, int aligned, int rotated, int* status, void* stream);
int d2amd_roi_align_f64_backward(const double* grad_output, const double* rois, double* grad_input,
                                 int N, int C, int H, int W, int K, int pooled_h, int pooled_w,
                                 double spatial_scale, int sampling_ratio, int aligned, int rotated,
                                 void* stream);

/* ---- pairwise box IoU.  detectron2/structures/boxes.py:312-377 (pairwise_iou / _ioa /
 * _intersection): boxes1 [n,4], boxes2 [m,4] fp32 xyxy -> out [n,m] fp32. */
int d2amd_pairwise_iou(const float* boxes1, int n, const float* boxes2, int m, int mode,
                       float* out, void* stream);
/* torch.ops.detectron2.box_iou_rotated (vision.cpp:117; box_iou_rotated.h:20-33):
 * boxes [.,5] fp32 (cx, cy, w, h, angle_degrees) -> out [n,m] fp32. */
int d2amd_box_iou_rotated(const float* boxes1, int n, const float* boxes2, int m, float* out,
                          void* stream);

/* ---- proposal labelling + sampling for the ROI heads: ROIHeads.label_and_sample_proposals
 * (detectron2/modeling/roi_heads/roi_heads.py:219-295) = add_ground_truth_to_proposals
 * (proposal_generator/proposal_utils.py:138-205) + pairwise_iou + Matcher (allow_low_quality_matches=False,
 * roi_heads.py:176-180) + _sample_proposals (roi_heads.py:181-216) + subsample_labels (modeling/sampling.py:9-54), for
 * a batch, with a FIXED output shape and no host sync (the ROI half of a captured step cannot wait for the host).
 * Per image: candidates = the first n proposals, n = min(max_proposals, limits[0 .. n_limits)) with `limits` DEVICE
 * int64 words `limit_stride` apart (e.g. the kept / finite counts d2amd_nms_batched_runs left in its result row
 * {kept, flags, finite, 0}: n_limits 2, limit_stride 2; n_limits <= 4, may be 0), followed by the ground-truth boxes when append_gt.  keys: one uniform random fp32 per candidate SLOT --
 * [max_proposals] for the proposals, then [num_gt] for the appended boxes (no NaN).  Sampling rule: the
 * min(#positives, max_positives) smallest keys among the positives (class not in {-1, num_classes}), then the
 * min(#negatives, batch_size_per_image - sampled positives) smallest among the negatives; ties by candidate index.
 * Outputs, batch_size_per_image rows per image: positives first, then negatives (ascending key inside a group), then
 * padding -- boxes_out [count][S][4] (zero), classes_out [count][S] (ground-truth class, num_classes for background,
 * -1), gt_index_out [count][S] (the matched ground truth: matched_idxs[sampled_idxs]; 0 for padding / no ground
 * truth), index_out [count][S] (candidate index = the reference's sampled_idxs; -1), counts_out [count][2] =
 * (positives, rows).  thresholds / labels: Matcher's constructor arguments (host), as for d2amd_match_boxes.
 * rois_out (optional) [count * S][5]: the same rows in pooler format (image index, x1, y1, x2, y2:
 * convert_boxes_to_pooler_format, modeling/poolers.py:62-104), head_rois_out (optional) [count * head_rows][5]: the first
 * head_rows rows of every image (the mask head's rows: positives come first) -- what d2amd_roi_pooler_forward takes
 * as they are; head_classes_out (optional) [count][head_rows]: classes_out of those rows, contiguous (what the mask loss
 * takes as it is: a strided slice of classes_out would cost the caller a copy launch).
 * key_state (optional): the {seed, offset, ticket} words of a device-resident generator (d2amd_uniform_keys below).
 * The keys are then drawn INSIDE the kernel -- images[i].keys is ignored; key c of image i is output
 * (sum of max_proposals + num_gt of the images before i) + c of d2amd_uniform_keys at that state -- and the call
 * advances the generator like one d2amd_uniform_keys call: a captured step needs no key launch (and no cross-stream
 * wait for one) in front of the sampler.
 * max_proposals + num_gt <= d2amd_label_and_sample_max_candidates() per image, else D2AMD_EUNSUPPORTED. */
typedef struct {
  const float* proposals;    /* [max_proposals][4] fp32 xyxy, 16-byte aligned */
  const int64_t* limits;     /* [n_limits] device words, or NULL */
  const float* gt_boxes;     /* [num_gt][4], 16-byte aligned */
  const int64_t* gt_classes; /* [num_gt] */
  const float* keys;         /* [max_proposals + num_gt] */
  int max_proposals, n_limits, num_gt;
  int limit_stride;          /* words between the limits: limits[0], limits[limit_stride], ... (0 or 1: contiguous) */
} d2amd_sample_image;
int d2amd_label_and_sample_max_candidates(void);
int d2amd_label_and_sample_proposals(const d2amd_sample_image* images, int count, const float* thresholds,
                                     const int8_t* labels, int T, int batch_size_per_image, int max_positives,
                                     int64_t num_classes, int append_gt, float* boxes_out, int64_t* classes_out,
                                     int64_t* gt_index_out, int64_t* index_out, int32_t* counts_out, float* rois_out,
                                     float* head_rois_out, int64_t* head_classes_out, int head_rows,
                                     uint64_t* key_state, void* stream);

/* ---- uniform sampling keys from a DEVICE-resident generator state (what the samplers below consume; the reference
 * draws torch.randperm inside subsample_labels, modeling/sampling.py:49-50).  state: 3 device uint64 words {seed, offset,
 * 0}; every call writes n fp32 values in [0, 1) (Philox4x32-10, 24 random bits each) to `out` and advances the offset
 * on the device -- a captured HIP graph draws fresh keys at every replay without the two host-side fill launches torch's
 * generator needs in front of each hipGraphLaunch. */
int d2amd_uniform_keys(uint64_t* state, float* out, int64_t n, void* stream);

/* ---- subsample_labels (detectron2/modeling/sampling.py:9-54) for a batch, on the device, FIXED output shape, no
 * host sync -- the reference pays two nonzero() syncs and two randperm sorts per image.  Callers:
 * RPN._subsample_labels (proposal_generator/rpn.py:287-305: 268,569 anchor labels per image -> 256, label vector
 * rewritten to -1 / 0 / 1) and ROIHeads._sample_proposals (roi_heads/roi_heads.py:181-216).
 * labels [N][n] int8 (label_bytes 1: the Matcher's labels) or int64 (label_bytes 8: classes): -1 ignore, bg_label
 * negative, anything else positive.  keys [N][n] fp32, one uniform random number per element (no NaN): the sample is
 * the min(#positives, max_positives) SMALLEST keys among the positives, then the min(#negatives, num_samples - sampled
 * positives) smallest among the negatives, ties towards the lower index -- a uniform random subset of each group, the
 * random stream is the caller's.  max_positives = int(num_samples * positive_fraction) (sampling.py:42).
 * Outputs (each may be NULL): pos_idx_out [N][max_positives], neg_idx_out [N][num_samples] int64 element indices in
 * ascending (key, index) order, padded with -1; counts_out [N][2] = (sampled positives, sampled negatives);
 * labels_out [N][n] int8 = -1 everywhere, 1 at the sampled positives, 0 at the sampled negatives (rpn.py:300-304;
 * may alias `labels` when those are int8).  num_samples <= 65,536. */
size_t d2amd_subsample_labels_workspace_bytes(int N, int64_t n, int num_samples, int max_positives);
int d2amd_subsample_labels(const void* labels, int label_bytes, int N, int64_t n, const float* keys, int num_samples,
                           int max_positives, int64_t bg_label, int64_t* pos_idx_out, int64_t* neg_idx_out,
                           int32_t* counts_out, int8_t* labels_out, void* workspace, size_t workspace_bytes,
                           void* stream);

/* ---- anchor / proposal matching.  Matcher.__call__ + set_low_quality_matches_
 * (detectron2/modeling/matcher.py:62-127) fused with pairwise_iou (structures/boxes.py:312-358), as called
 * from proposal_generator/rpn.py:307-364 and roi_heads/roi_heads.py:257-295: the M x N matrix is never
 * written.  gt_boxes [M,4], boxes [N,4] fp32 xyxy; thresholds (host) [T] positive ascending, labels (host)
 * [T+1] in {-1,0,1} (Matcher's constructor arguments); matches [N] int64 = index of the first maximal-IoU
 * ground truth (0 when M == 0), match_labels [N] int8.  workspace: d2amd_matcher_workspace_bytes(M)
 * (row maxima, only used with allow_low_quality).  The reference's `assert torch.all(matrix >= 0)` is not
 * evaluated (it would cost a host sync); NaN qualities propagate like torch.max.
 * d2amd_match_quality_matrix is Matcher.__call__ for a caller that already holds the row-major M x N matrix. */
#define D2AMD_MATCHER_MAX_THRESHOLDS 8
size_t d2amd_matcher_workspace_bytes(int M);
int d2amd_match_boxes(const float* gt_boxes, int M, const float* boxes, int N, const float* thresholds,
                      const int8_t* labels, int T, int allow_low_quality, int64_t* matches,
                      int8_t* match_labels, void* workspace, size_t workspace_bytes, void* stream);
/* The same for a batch of images against the SAME boxes (the RPN's anchors; proposal_generator/rpn.py:331-353 loops over
 * the images in Python): gt_boxes (host array of `count` device pointers) [M[i]][4], M (host) [count]; outputs
 * matches [count][N], match_labels [count][N] -- ONE launch per pass for the batch.  workspace:
 * d2amd_match_boxes_batch_workspace_bytes(M, count) (row maxima, allow_low_quality only). */
size_t d2amd_match_boxes_batch_workspace_bytes(const int* M, int count);
int d2amd_match_boxes_batch(const float* const* gt_boxes, const int* M, int count, const float* boxes, int N,
                            const float* thresholds, const int8_t* labels, int T, int allow_low_quality,
                            int64_t* matches, int8_t* match_labels, void* workspace, size_t workspace_bytes,
                            void* stream);
int d2amd_match_quality_matrix(const float* quality, int M, int N, const float* thresholds,
                               const int8_t* labels, int T, int allow_low_quality, int64_t* matches,
                               int8_t* match_labels, void* workspace, size_t workspace_bytes, void* stream);

/* ---- RPN / RetinaNet proposal selection in front of NMS, all images and feature levels in one call.
 * Replaces proposal_generator/rpn.py:468-533 (_decode_proposals), modeling/box_regression.py:71-116
 * (Box2BoxTransform.apply_deltas) and proposal_generator/proposal_utils.py:62-120 (per-level topk + gather,
 * isfinite filter, Boxes.clip, Boxes.nonempty).
 *   logits  [N, Atot] fp32 objectness of every anchor, levels concatenated (level l = columns
 *           [sum(level_sizes[:l]), +level_sizes[l]));  deltas [N, Atot, 4];  anchors [Atot, 4] xyxy
 *   level_sizes (host) [L];  image_hw (host) [N][2] = (height, width);  weights (host) [4] = Box2BoxTransform
 *   weights (wx, wy, ww, wh);  scale_clamp = its clamp on dw / dh.
 * Per (image, level) the min(level_sizes[l], pre_nms_topk) highest logits are selected (ties: lower anchor
 * index first), decoded, clipped to the image; Ktot = sum of those counts.  Outputs, [N, Ktot] row-major
 * with levels concatenated in order and scores descending inside a level:
 *   boxes [N,Ktot,4], scores [N,Ktot], valid [N,Ktot] uint8 (finite and both sides > min_box_size after the
 *   clip; invalid rows hold a zero box and score -inf), level [Ktot] int64, flags [1] int32 (bit 0: a
 *   non-finite box or score was seen -- the reference raises FloatingPointError in training).
 * The selection is a segmented radix select (csrc/topk.hip); pre_nms_topk > 65536 per level: D2AMD_EUNSUPPORTED.
 * Nothing synchronises with the host. */
#define D2AMD_RPN_MAX_LEVELS 8
size_t d2amd_rpn_select_workspace_bytes(int N, int Atot);
int d2amd_rpn_select_proposals(const float* logits, const float* deltas, const float* anchors, int N, int Atot,
                               const int* level_sizes, int L, const int* image_hw, int pre_nms_topk,
                               float min_box_size, const float* weights, float scale_clamp, float* boxes_out,
                               float* scores_out, uint8_t* valid_out, int64_t* level_out, int* flags_out,
                               void* workspace, size_t workspace_bytes, void* stream);
/* The same selection on the RPN head's per-level outputs as rpn.py:431-449 holds them -- logits[l] [N, A_l],
 * deltas[l] [N, A_l, 4], anchors[l] [A_l, 4], fp32 and contiguous, device pointers in host arrays of length L
 * (level_sizes[l] = A_l): no concatenated copy of the inputs.  Same outputs, same workspace
 * (d2amd_rpn_select_workspace_bytes(N, sum of level_sizes)); pre_nms_topk <= 65536. */
int d2amd_rpn_select_proposals_levels(const float* const* logits, const float* const* deltas,
                                      const float* const* anchors, int N, const int* level_sizes, int L,
                                      const int* image_hw, int pre_nms_topk, float min_box_size,
                                      const float* weights, float scale_clamp, float* boxes_out, float* scores_out,
                                      uint8_t* valid_out, int64_t* level_out, int* flags_out, void* workspace,
                                      size_t workspace_bytes, void* stream);

/* ---- Dense-detector (RetinaNet) prediction selection in front of NMS, all images and levels in one call.
 * Replaces meta_arch/dense_detector.py:186-245 (_decode_per_level_predictions per level and image: `scores >
 * score_thresh`, `nonzero` [host sync], `topk`, gather, Box2BoxTransform.apply_deltas) and the `sigmoid_()` over
 * all class logits in front of it (meta_arch/retinanet.py:267).
 *   logits[l]  [N, A_l, K] fp32 class LOGITS of level l (device pointers in a host array), deltas[l] [N, A_l, 4],
 *   anchors[l] [A_l, 4] xyxy;  level_anchors (host) [L] = A_l;  K = num_classes.
 * Per (image, level): the candidates are the (anchor, class) pairs with sigmoid(logit) > score_thresh; the
 * min(topk_candidates, #candidates) best are selected by a segmented radix select (no sort of the A_l*K scores, no
 * host sync; ties: lower flattened index a*K + c first; topk_candidates <= 65536), decoded WITHOUT clipping.
 * Outputs, [N, Ktot] row-major, Ktot = sum_l min(A_l*K, topk_candidates), levels in order, best first inside a
 * level: boxes [N,Ktot,4], scores [N,Ktot] (= sigmoid(logit)), classes [N,Ktot] int64, valid [N,Ktot] uint8 (rows
 * past a segment's count: zero box, score -inf, class 0, valid 0), counts [N,L] int32, logits_out [N,Ktot] or NULL
 * (the selected logits: an exp()-independent ranking key for the NMS that follows; -inf past the count).
 * Ranking: by LOGIT descending, equal logits towards the lower flattened (anchor, class) index.  sigmoid is
 * monotone, so this is the reference's `topk` order wherever its fp32 scores differ, and a defined order inside a
 * group of equal fp32 scores (torch.topk leaves that unspecified).  Candidates: logit > log(t / (1 - t)) for the fp32
 * threshold t, evaluated in double (the exact-arithmetic form of `sigmoid(logit) > t`, dense_detector.py:207). */
size_t d2amd_dense_select_workspace_bytes(int N, const int* level_anchors, int L, int num_classes,
                                          int topk_candidates);
int d2amd_dense_select_predictions(const float* const* logits, const float* const* deltas,
                                   const float* const* anchors, int N, const int* level_anchors, int L,
                                   int num_classes, float score_thresh, int topk_candidates, const float* weights,
                                   float scale_clamp, float* boxes_out, float* scores_out, int64_t* classes_out,
                                   uint8_t* valid_out, int* counts_out, float* logits_out, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* ---- NMS.  One entry serves torchvision.ops.nms / batched_nms (detectron2/layers/nms.py:6,
 * 11-22) and torch.ops.detectron2.nms_rotated / batched_nms_rotated (vision.cpp:116,
 * nms_rotated.h:22-37, nms.py:96-147).
 *   boxes   [n,4] xyxy (rotated=0) or [n,5] cxcywha (rotated=1), fp32
 *   scores  [n] fp32;  idxs [n] int64 category per box, or NULL (single category)
 *   suppression: IoU >  iou_threshold (rotated=0, torchvision CPU semantics)
 *                IoU >= iou_threshold (rotated=1, nms_rotated_cpu.cpp:54); compare in double.
 *   keep_out [n] int64: kept ORIGINAL indices in decreasing score order (ties: lower index
 *            first);  result [4] int64 (device): {number kept, error flags, number kept whose score is > -inf,
 *            0} (callers that park invalid rows at score -inf read the third word: those rows sort last).  error flag bit 0:
 *            a category has more than `max_per_class` boxes, bit 1: category id out of range.
 *   max_per_class: upper bound on boxes in one category (sizes the suppression bitmask);
 *            <= 0 means "unknown" = n.   Category ids must lie in [0, 65535].
 * The whole pipeline (sort, bitmask, greedy reduction, compaction) runs on the device; the
 * caller's only host sync is reading `result`. */
size_t d2amd_nms_workspace_bytes(int64_t n, int64_t max_per_class, int rotated);
int d2amd_nms(const float* boxes, const float* scores, const int64_t* idxs, int64_t n,
              double iou_threshold, int rotated, int64_t max_per_class, int64_t* keep_out,
              int64_t* result, void* workspace, size_t workspace_bytes, void* stream);
/* The NMS of every image of a batch as ONE device pipeline (one launch per stage for all images): replaces the
 * per-image Python loops of find_top_rpn_proposals (proposal_generator/proposal_utils.py:118-135) and of
 * DenseDetector inference (meta_arch/dense_detector.py:186-260).  Arrays of `count` per-image arguments, each with
 * the meaning it has in d2amd_nms (idxs / max_per_class may be NULL; idxs[k] may be NULL); workspace[k] sized by
 * d2amd_nms_workspace_bytes(n[k], ...).  Every n[k] must be <= d2amd_nms_batched_max_boxes() (larger inputs go
 * through d2amd_nms, which sorts by radix).  Results are identical to `count` calls of d2amd_nms. */
int d2amd_nms_batched(int count, const float* const* boxes, const float* const* scores,
                      const int64_t* const* idxs, const int64_t* n, double iou_threshold, int rotated,
                      const int64_t* max_per_class, int64_t* const* keep_out, int64_t* const* result,
                      void* const* workspace, const size_t* workspace_bytes, void* stream);
int d2amd_nms_batched_max_boxes(void);
/* The same NMS for callers whose input is already ordered in RUNS -- find_top_rpn_proposals and DenseDetector
 * inference hand over the per-level top-k lists the selection has just sorted (proposal_utils.py:62-80,
 * dense_detector.py:207-223): run r = rows [run_offsets[r], run_offsets[r+1]) (host array of n_runs + 1 ints,
 * run_offsets[n_runs] == n; n_runs <= 8), and inside a run the rows whose score is not -inf are in decreasing score
 * order, equal scores in row order (rows at -inf -- parked invalid rows -- may sit anywhere).  The global order is then
 * obtained by binary searches between the runs instead of an n^2 ranking / two radix sorts; results are identical to
 * d2amd_nms / d2amd_nms_batched on the same input.  runs_are_categories = 1: the category of a row is the index of its
 * run (idxs must be NULL) -- the RPN's per-level NMS.  A run that is not in order sets error flag bit 2 (value 4) in
 * result[1]: keep_out is then unspecified and the caller must use the general entry.
 * gather (optional, NULL: none; one struct per image for the batched entry): up to 4 device arrays with n rows of
 * row_bytes[t] bytes each (a multiple of 4) whose kept rows are copied, in keep order, to dst[t] while keep_out is
 * written: dst[t] row j = src[t] row keep_out[j] for j < result[0] -- the `boxes[keep]`, `scores[keep]`,
 * `classes[keep]` gathers both callers issue after their host sync (proposal_utils.py:127-134,
 * dense_detector.py:254-259), without the extra launches.  (Unspecified, like keep_out, when flag 4 is set.) */
typedef struct {
  int count;
  const void* src[4];
  void* dst[4];
  int row_bytes[4];
} d2amd_nms_gather;
/* num_categories: 0 = unknown, else the category ids are < num_categories (the class sort of large inputs then
 * covers only the bits in use: 80 classes = one 8-bit pass instead of two; ids beyond it are the caller's error). */
int d2amd_nms_runs(const float* boxes, const float* scores, const int64_t* idxs, int64_t n, const int* run_offsets,
                   int n_runs, int runs_are_categories, int num_categories, double iou_threshold, int rotated,
                   int64_t max_per_class, int64_t* keep_out, int64_t* result, void* workspace, size_t workspace_bytes,
                   const d2amd_nms_gather* gather, void* stream);
int d2amd_nms_batched_runs(int count, const float* const* boxes, const float* const* scores,
                           const int64_t* const* idxs, const int64_t* n, const int* run_offsets, int n_runs,
                           int runs_are_categories, int num_categories, double iou_threshold, int rotated,
                           const int64_t* max_per_class,
                           int64_t* const* keep_out, int64_t* const* result, void* const* workspace,
                           const size_t* workspace_bytes, const d2amd_nms_gather* gather, void* stream);

/* ---- paste_masks_in_image.  detectron2/layers/mask_ops.py:74-147.
 * masks [n,mh,mw] `mask_dtype`; boxes [n,4] fp32; out [n,img_h,img_w] uint8:
 * threshold >= 0 -> 0/1 (torch.bool storage), threshold < 0 -> trunc(value*255). */
int d2amd_paste_masks(const void* masks, const float* boxes, int n, int mh, int mw, int img_h,
                      int img_w, float threshold, uint8_t* out, int mask_dtype, void* stream);

/* ---- BitMasks.crop_and_resize (detectron2/structures/masks.py:193-224): the Mask R-CNN training targets.
 * masks [G,H,W] uint8 / bool storage (non-zero = inside), boxes [G,4] fp32 xyxy (box g crops mask g) ->
 * out [G,mask_size,mask_size] uint8 0/1 = (ROIAlign((M,M), 1.0, 0, aligned=True)(mask as fp32) >= 0.5),
 * without the fp32 copy of the full-resolution masks; sampling and summation order are torchvision's CPU
 * roi_align, so the thresholded result is bit-identical to the reference pipeline. */
int d2amd_bitmask_crop_and_resize(const uint8_t* masks, const float* boxes, int G, int H, int W,
                                  int mask_size, uint8_t* out, void* stream);
/* Same, for the sampled proposals of an image: box i crops mask mask_index[i] of the image's n_masks ground-truth
 * masks -- `gt_masks[sampled_targets].crop_and_resize(proposal_boxes, M)` (roi_heads/roi_heads.py:280-291 +
 * mask_head.py:65-67) without the (n_boxes, H, W) indexed copy `BitMasks.__getitem__` (masks.py:122-142) makes.
 * status [1] int32 (device, zeroed by the caller, may be NULL): bit 0 = an index outside [0, n_masks) (torch
 * indexing raises IndexError; such rows are written as zeros). */
int d2amd_bitmask_crop_and_resize_indexed(const uint8_t* masks, int n_masks, const float* boxes,
                                          const int64_t* mask_index, int n_boxes, int H, int W, int mask_size,
                                          uint8_t* out, int* status, void* stream);
/* All images of a batch in ONE launch (mask_rcnn_loss loops over the images and concatenates, mask_head.py:57-77):
 * host arrays of num_images (<= 64) device pointers / counts; every image's masks are [n_masks[i], H, W]; mask_index
 * may be NULL (then n_boxes[i] == n_masks[i], box g of an image crops its mask g); out [sum n_boxes, M, M] in image
 * order. */
int d2amd_bitmask_crop_and_resize_batch(int num_images, const uint8_t* const* masks, const int* n_masks,
                                        const float* const* boxes, const int64_t* const* mask_index,
                                        const int* n_boxes, int H, int W, int mask_size, uint8_t* out, int* status,
                                        void* stream);

/* ---- PolygonMasks.crop_and_resize (detectron2/structures/masks.py:396-420 + rasterize_polygons_within_box :39-86 +
 * polygons_to_bitmask :20-36 = pycocotools frPyObjects / merge / decode): Mask R-CNN's training targets from COCO
 * polygons (the default MASK_FORMAT), rasterised on the device.
 *   coords        [total] float64 (device): x0, y0, x1, y1, ... of all polygons, concatenated
 *   poly_offsets  [P + 1] int64 (device): polygon p = coords[poly_offsets[p] .. poly_offsets[p + 1])
 *   inst_offsets  [n_instances + 1] int64 (device): instance i = polygons [inst_offsets[i], inst_offsets[i + 1])
 *   boxes [n_boxes,4] fp32 xyxy; index [n_boxes] int64 or NULL: box k crops instance index[k] (the matched ground
 *   truth of a sampled proposal), NULL: instance k (then n_boxes == n_instances)
 *   out [n_boxes, M, M] uint8 0/1.  status [1] int32 (device, zeroed by the caller, may be NULL): bit 0 an index
 *   outside [0, n_instances), bit 1 a polygon with more than 4,096 vertices (skipped), bit 2 a boundary walk of
 *   2^30 points (skipped).  mask_size <= 64.  The rasteriser restates cocoapi's published algorithm (pycocotools is
 *   not part of the reference tree: parity with it is unpinned). */
int d2amd_polygon_crop_and_resize(const double* coords, const int64_t* poly_offsets, const int64_t* inst_offsets,
                                  int n_instances, const float* boxes, const int64_t* index, int n_boxes,
                                  int mask_size, uint8_t* out, int* status, void* stream);

/* ---- Fused multi-level ROIAlignRotated pooler (csrc/roi_pool_rot.hip) --------------------------------------------------
 * ROIPooler.forward with pooler_type "ROIAlignRotated" (modeling/poolers.py:206-263 on layers/roi_align_rotated.py /
 * csrc/ROIAlignRotated) in one launch per direction: level assignment (poolers.py:51-59 on RotatedBoxes.area() = w * h)
 * inside the kernel, no per-level nonzero / index_put_.  Same d2amd_pooler_params as the axis-aligned pooler (`aligned`
 * is ignored: ROIAlignRotated always samples with the half-pixel shift); NHWC only (d2amd_roi_pooler_rotated_supported).
 *   rois [K][6] fp32 = (image index, cx, cy, w, h, angle in degrees); output / grad_output [K][PH][PW][C] (NHWC).
 *   status (device int, may be NULL): bit 0 is set when a ROI has a negative size -- its rows are zero; the reference
 *   asserts (ROIAlignRotated_cpu.cpp:236-238).
 * backward: a deterministic gather (no atomics, no fp32 image, bit-identical from run to run): the ROIs' merged tap tables
 * are turned into per-pixel lists of {dY row, weight}, ordered by row, and a wave per pixel accumulates them in fp32 and
 * writes the pixel once.  ROIs whose bins span more than 64 distinct pixels or 64 samples (bins wider than ~8 px at their
 * level) take the reference's atomic scatter into an fp32 image the gather adds -- only those are order-dependent, like
 * upstream.  workspace: d2amd_roi_pooler_rotated_backward_workspace_bytes(p, K).  grad_inputs are written completely. */
int d2amd_roi_pooler_rotated_supported(const d2amd_pooler_params* p);
int d2amd_roi_pooler_rotated_forward(const d2amd_pooler_params* p, const void* const* inputs, const float* rois,
                                     void* output, int K, int* status, void* stream);
size_t d2amd_roi_pooler_rotated_backward_workspace_bytes(const d2amd_pooler_params* p, int K);
int d2amd_roi_pooler_rotated_backward(const d2amd_pooler_params* p, const void* grad_output, const float* rois,
                                      void* const* grad_inputs, int K, void* workspace, size_t workspace_bytes,
                                      void* stream);

/* ---- Box-head inference in front of the per-class NMS (csrc/box_head.hip) ---------------------------------------------
 * Replaces detectron2/modeling/roi_heads/fast_rcnn.py:134-158 (fast_rcnn_inference_single_image up to `batched_nms`)
 * for a whole batch, without the reference's per-image host sync (`filter_mask.nonzero()`, :150):
 *   boxes[i]  [rows[i]][num_bbox_reg_classes * 4] fp32 (predict_boxes; num_bbox_reg_classes = 1: class-agnostic),
 *   scores[i] [rows[i]][num_classes + 1] fp32 (predict_probs: the last column is the background),
 *   image_hw  [num_images][2] = (height, width) the boxes are clipped to (Boxes.clip: x to [0, w], y to [0, h]).
 * A row with a non-finite box coordinate or score is dropped as a whole (:134-137).  The candidates -- (row, class)
 * pairs with score > score_thresh -- come out in torch.nonzero's order (row-major), image i's from row
 * sum_{j<i} rows[j] * num_classes on (worst-case slices: nothing overflows):
 *   out_boxes [.][4] (clipped), out_scores [.], out_classes [.] int64, out_rows [.] int64 (`filter_inds[:, 0]`: the row's
 *   index among the rows of its image that were NOT dropped -- the reference indexes boxes[valid_mask]), counts [num_images] int64 -- all on the device; rows past an image's count are not written.
 * workspace: d2amd_fast_rcnn_filter_workspace_bytes(rows, num_images).  num_images <= D2AMD_POOLER_MAX_IMAGES.
 * The caller then runs d2amd_nms_batched over the first counts[i] rows of every slice (class = category) and keeps the
 * first topk_per_image of its result (:161-164). */
size_t d2amd_fast_rcnn_filter_workspace_bytes(const int* rows, int num_images);
int d2amd_fast_rcnn_filter(const float* const* boxes, const float* const* scores, const int* rows, int num_images,
                           int num_classes, int num_bbox_reg_classes, const int* image_hw, float score_thresh,
                           float* out_boxes, float* out_scores, int64_t* out_classes, int64_t* out_rows,
                           int64_t* counts, void* workspace, size_t workspace_bytes, void* stream);

/* The same path WITHOUT a host sync between the filter, the NMS and the top-k cut (fast_rcnn.py:150-170 reads two sizes on
 * the host per image): every image works on a fixed WINDOW of its candidate slots.
 * window_i = min(window, rows[i] * num_classes) slots, the first ones of image i's slice of the candidate arrays (the slice
 * starts at sum_{j<i} rows[j] * num_classes).
 * d2amd_fast_rcnn_park: slots [counts[i], window_i) become inert rows -- zero box, score -inf, classes >= num_classes (one per 64 slots) -- so that
 * d2amd_nms_batched over window_i rows per image (idxs = the class array; images with window_i = 0 left out) gives the
 * reference's kept set for the live rows, in its order, with the parked rows last.
 * d2amd_fast_rcnn_take: row t < topk of image i = candidate keep[i][t] while t < min(kept, finite-score kept, topk) of its
 * NMS result row {kept, flags, finite, 0} (nms_result: int64 [images with window_i > 0][4], in image order; keep[i] may
 * be NULL for the others); rows behind that hold a 1 x 1 box at the origin, score 0, class 0, row 0.
 * det_*: [num_images][topk] (boxes x 4); det_counts: int64 [num_images]. */
int d2amd_fast_rcnn_park(const int* rows, int num_images, int num_classes, int window, const int64_t* counts,
                         float* out_boxes, float* out_scores, int64_t* out_classes, void* stream);
int d2amd_fast_rcnn_take(const int* rows, int num_images, int num_classes, int window, int topk,
                         const int64_t* const* keep, const int64_t* nms_result, const float* cand_boxes,
                         const float* cand_scores, const int64_t* cand_classes, const int64_t* cand_rows,
                         float* det_boxes, float* det_scores, int64_t* det_classes, int64_t* det_rows,
                         int64_t* det_counts, void* stream);

/* FastRCNNOutputLayers.predict_boxes + predict_probs for the batch in ONE launch (roi_heads/fast_rcnn.py:524-568:
 * `box2box_transform.apply_deltas(proposal_deltas, cat(proposal_boxes))` = box_regression.py:88-116, ~40 elementwise
 * launches; `F.softmax(scores, -1)` or `scores.sigmoid()` for use_sigmoid_ce).
 * scores [R, K + 1] and deltas [R, Kb * 4] in `dtype` (R = sum of rows[i], Kb = num_bbox_reg_classes = K or 1), contiguous;
 * proposals[i] fp32 [rows[i], 4]; weights[4] + scale_clamp = Box2BoxTransform's.  boxes_out fp32 [R, Kb * 4] (apply_deltas
 * computes in fp32 whatever the head's dtype: box_regression.py:88), probs_out [R, K + 1] in `dtype`.  Decode: the
 * reference's fp32 expression order as it evaluates on a GPU (no contraction; `deltas / w` for a Python scalar w is ATen's
 * multiplication by the fp32 reciprocal there, an IEEE division on the CPU: <= 1 ulp apart) -> bit-identical to the device run; softmax: exp(x - max) / sum with
 * a butterfly sum (last-ulp differences from ATen's reduction order).
 * limits (nullable; limits[i] nullable): the NMS result row {kept, flags, finite, ..} of the device-side proposal list the
 * rows came from (DeviceProposals.limits): rows at / behind min(kept, finite) predict nothing -- probability 1 on the
 * background column (0 everywhere with use_sigmoid), zero boxes -- without the host knowing the count. */
int d2amd_fast_rcnn_predict(const void* scores, const void* deltas, int dtype, const float* const* proposals,
                            const int64_t* const* limits, const int* rows, int num_images, int num_classes,
                            int num_bbox_reg_classes, const float* weights, float scale_clamp, int use_sigmoid,
                            float* boxes_out, void* probs_out, void* stream);
/* Rows of device-side proposal lists at / behind their live count min(limits[i][0], limits[i][2]) <- the box (0, 0, 1, 1),
 * in place: what lets the box pooler and the decode run at the fixed shape [rows[i], 4] (proposal_utils.py:67-135 returns a
 * list of the live length after a host read). */
int d2amd_proposals_pad(float* const* boxes, const int64_t* const* limits, const int* rows, int num_images, void* stream);

/* ---- Mask-head glue.  detectron2/modeling/roi_heads/mask_head.py:31-158.
 * logits [B,C,HW] `dtype` (HW = Hmask*Wmask, contiguous NCHW), classes [B] int64 or NULL (class-agnostic,
 * C == 1), gt_masks [B,HW] uint8 / bool storage (the output of d2amd_bitmask_crop_and_resize).
 * mask_rcnn_inference (mask_head.py:116-158): out [B,1,HW] = sigmoid(logits[b, classes[b]]), same dtype;
 *   a class outside [0, C) yields NaN rows (the reference's gather raises an index error).
 * mask_rcnn_loss forward (mask_head.py:31-113): loss_out[0] = mean over B*HW of
 *   binary_cross_entropy_with_logits(logits[b, gt_classes[b]], gt_masks) in fp32;
 *   stats_out[5] int64 = {#incorrect ((x > 0) != gt), #positive gt, #false positive, #false negative,
 *   #rows whose class is outside [0, C)} -- the counts behind mask_rcnn/accuracy, false_positive,
 *   false_negative (mask_head.py:88-95), left on the device (the reference reads each with .item()).
 *   Deterministic (fixed-order reduction).  B == 0 is the caller's case (`pred_mask_logits.sum() * 0`).
 * mask_rcnn_loss backward: grad_logits [B,C,HW] `dtype`, written completely: grad_loss[0] *
 *   (sigmoid(x) - gt) / (B*HW) in the class plane, 0 elsewhere.  grad_loss is a DEVICE scalar (fp32). */
int d2amd_mask_rcnn_inference(const void* logits, const int64_t* classes, int B, int C, int HW, int dtype,
                              void* out, void* stream);
size_t d2amd_mask_rcnn_loss_workspace_bytes(int B);
int d2amd_mask_rcnn_loss_forward(const void* logits, const int64_t* gt_classes, const uint8_t* gt_masks, int B,
                                 int C, int HW, int dtype, float* loss_out, int64_t* stats_out, void* workspace,
                                 size_t workspace_bytes, void* stream);
int d2amd_mask_rcnn_loss_backward(const void* logits, const int64_t* gt_classes, const uint8_t* gt_masks,
                                  const float* grad_loss, int B, int C, int HW, int dtype, void* grad_logits,
                                  void* stream);
/* The same over the rows that COUNT: a row whose class is outside [0, C) -- the background (num_classes) and padding
 * (-1) rows of d2amd_label_and_sample_proposals' fixed-size lists -- is ignored instead of reported: no loss, no
 * statistics, zero gradient; the mean is taken over the other rows (mask_head.py:47-113 on the foreground subset
 * select_foreground_proposals would have made, roi_heads.py:37-75).  stats_out[6]: [4] = ignored rows, [5] = rows
 * that count; no row -> loss 0 (the reference: `pred_mask_logits.sum() * 0`).  The backward reads the row count from
 * the DEVICE (`rows` = &stats_out[5]): nothing here needs the host (tests/test_gpu_label_sample.py,
 * tests/test_gpu_connected_step.py). */
int d2amd_mask_rcnn_loss_forward_masked(const void* logits, const int64_t* gt_classes, const uint8_t* gt_masks, int B,
                                        int C, int HW, int dtype, float* loss_out, int64_t* stats_out,
                                        void* workspace, size_t workspace_bytes, void* stream);
int d2amd_mask_rcnn_loss_backward_masked(const void* logits, const int64_t* gt_classes, const uint8_t* gt_masks,
                                         const float* grad_loss, const int64_t* rows, int B, int C, int HW, int dtype,
                                         void* grad_logits, void* stream);

/* ---- deformable convolution v1 / v2.  Replaces detectron2._C.deform_conv_forward,
 * deform_conv_backward_input, deform_conv_backward_filter, modulated_deform_conv_forward,
 * modulated_deform_conv_backward (vision.cpp:85-102; csrc/deformable/deform_conv.h:116-375).
 * x [B,C,H,W], offset [B,dg*2*kh*kw,Ho,Wo], mask [B,dg*kh*kw,Ho,Wo] or NULL (v1),
 * weight [Co,C/groups,kh,kw], bias [Co] or NULL, out [B,Co,Ho,Wo]; all `dtype`, NCHW.
 * Unlike the reference there is no column buffer in HBM and no per-image loop. */
typedef struct d2amd_dcn_params {
  int B, C, H, W, Co, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups,
      deformable_groups, dtype;
  int layout; /* D2AMD_NCHW (the reference's), or D2AMD_NHWC: x / out / grad_out / grad_input are [B,H,W,C]-contiguous
               * (torch.channels_last) -- the kernels' native layout: no transpose in, no transpose out; offset, mask
               * and the weights keep their NCHW / OIHW layout.  NHWC is served by the 16-bit MFMA path only
               * (D2AMD_EUNSUPPORTED otherwise: the caller converts). */
} d2amd_dcn_params;
/* backward: 0 = the forward, 1 = the backward, 2 = a forward that is handed `columns` to keep
 * (d2amd_deform_conv_forward_columns: no scratch column of its own -- 77 MB less for an R50 res3 block of 2 images). */
size_t d2amd_deform_conv_workspace_bytes(const d2amd_dcn_params* p, int backward);
int d2amd_deform_conv_forward(const d2amd_dcn_params* p, const void* x, const void* offset,
                              const void* mask, const void* weight, const void* bias, void* out,
                              void* workspace, size_t workspace_bytes, void* stream);
/* Any of the grad_* outputs may be NULL to skip it.  Non-NULL outputs are overwritten. */
int d2amd_deform_conv_backward(const d2amd_dcn_params* p, const void* x, const void* offset,
                               const void* mask, const void* weight, const void* grad_out,
                               void* grad_input, void* grad_offset, void* grad_mask,
                               void* grad_weight, void* grad_bias, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---- the column buffer as a SAVED ACTIVATION.  The reference's Python hands `columns` scratch tensors to
 * _C.deform_conv_forward / modulated_deform_conv_forward (layers/deform_conv.py:97-98,248-254; the C++ resizes and
 * refills them per image, deform_conv_cuda.cu:346-353,916-918) and recomputes the im2col in the backward
 * (:1160-1179).  Here the training forward can KEEP the column it builds -- 16-bit, modulation mask folded in; opaque to
 * the caller: [B*Ho*Wo positions][kh*kw*C] row-major in both layouts; NHWC layout (the column-kernel + dense-GEMM path):
 * followed by the weights packed per tap as [kh*kw*C][Co].
 * d2amd_deform_conv_columns_bytes = its size (77 MB + the weights for an R50 res3 block of 2 images: sized for 288 GB of
 * HBM), 0 when the shape / dtype is not served (fp32, groups > 1, deformable_groups > 1, C % 64 != 0: pass columns =
 * NULL) -- and the backward's weight gradient becomes a dense split-K GEMM dW = dY^T col on MFMA instead of a second
 * gather, its data gradient (NHWC) a dense GEMM against the kept weights.  The same d2amd_dcn_params (layout included) in
 * both calls.  columns = NULL in either call = the plain entry. */
size_t d2amd_deform_conv_columns_bytes(const d2amd_dcn_params* p);
/* 1 if a channels_last call of this shape / dtype takes the column + dense-GEMM path (csrc/dcn_colpath.hip), whatever
 * p->layout says: what a caller holding NCHW activations (the reference's DeformBottleneckBlock, backbone/resnet.py:303-327)
 * asks before it stages them channels_last itself -- detectron2_amd/layers/deform_conv.py does. */
int d2amd_deform_conv_column_path(const d2amd_dcn_params* p);
int d2amd_deform_conv_forward_columns(const d2amd_dcn_params* p, const void* x, const void* offset,
                                      const void* mask, const void* weight, const void* bias, void* out,
                                      void* columns, void* workspace, size_t workspace_bytes, void* stream);
int d2amd_deform_conv_backward_columns(const d2amd_dcn_params* p, const void* x, const void* offset,
                                       const void* mask, const void* weight, const void* grad_out,
                                       const void* columns, void* grad_input, void* grad_offset, void* grad_mask,
                                       void* grad_weight, void* grad_bias, void* workspace,
                                       size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D2AMD_H_ */
