// Mask-head training targets: BitMasks.crop_and_resize in one kernel.
//   replaces  detectron2/structures/masks.py:193-224: bit_masks.to(float32) [an fp32 copy of every full
//             resolution mask: 4 B per pixel], ROIAlign((M, M), 1.0, 0, aligned=True) on it, `>= 0.5`.
// The mask is read directly as bool / uint8 (1 B per tap).  A bin's adaptive sampling grid is summed in
// parallel by a group of lanes, and re-evaluated SEQUENTIALLY in torchvision's CPU order (sum over iy, ix of the
// 4-tap bilinear value, then / count) when the parallel mean is too close to 0.5 to decide (see the kernel): the
// thresholded result is bit-identical to the reference pipeline, including bins whose mean is exactly 0.5.  Roofline: HBM-light (the G boxes' footprints of the masks are read once from
// L2); the point of the kernel is the 4 B/px fp32 copy it does not make.
// Compiled with -ffp-contract=off.
#pragma clang fp contract(off)
#include "common.h"

namespace d2amd {

// ROIAlignRotated_cpu.cpp:64-125 / torchvision roi_align pre_calc: value of the bilinear sample at (y, x)
__device__ __forceinline__ float crop_sample(const uint8_t* __restrict__ m, int H, int W, float y, float x) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;  // w = 0: contributes exactly 0
  if (y < 0.f) y = 0.f;
  if (x < 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - (float)y_low, lx = x - (float)x_low;
  const float hy = (float)(1. - (double)ly), hx = (float)(1. - (double)lx);
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  const float v1 = m[(long)y_low * W + x_low] ? 1.f : 0.f, v2 = m[(long)y_low * W + x_high] ? 1.f : 0.f;
  const float v3 = m[(long)y_high * W + x_low] ? 1.f : 0.f, v4 = m[(long)y_high * W + x_high] ? 1.f : 0.f;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

// Two-tier evaluation.  The reference sums a bin's gh x gw samples sequentially (iy outer, ix inner) in fp32, and
// only the comparison `mean >= 0.5` reaches the output.  A box of 600 px has 22 x 22 = 484 samples per bin: one
// thread per bin walking them in order is a 484-deep dependent chain of byte loads (r01: 117 us for 128 ROIs).
// Tier 1: a group of CROP_LANES lanes shares a bin, each lane sums a strided subset of the SAME sample values and
// the group adds the partial sums (a different summation order: |difference of the means| <= 2 n u, u = 2^-24, the
// samples being in [0, 1]).  If the tier-1 mean is farther than that bound from 0.5, the reference's sequential sum
// is on the same side: the bit is decided.  Tier 2 (rare: a mean within ~1e-4 of 0.5): one lane of the group redoes
// the bin in the reference's order.  The result is bit-identical to the sequential kernel for every input.
constexpr int CROP_LANES = 16;
constexpr int CROP_THREADS = 256;

__global__ __launch_bounds__(CROP_THREADS) void bitmask_crop_kernel(const uint8_t* __restrict__ masks,
                                                                   const float* __restrict__ boxes,
                                                                   const int64_t* __restrict__ mask_index, int n_masks,
                                                                   int G, int H, int W, int M, uint8_t* __restrict__ out,
                                                                   int* __restrict__ status) {
  const int g = blockIdx.y;
  const int sub = threadIdx.x % CROP_LANES;
  const int bin = blockIdx.x * (CROP_THREADS / CROP_LANES) + threadIdx.x / CROP_LANES;
  const bool live = bin < M * M;  // whole groups are live or not: the shuffles below stay inside a group
  const int ph = live ? bin / M : 0, pw = live ? bin - ph * M : 0;
  // box g crops mask mask_index[g] (the matched ground truth of a sampled proposal: roi_heads.py:280-291 indexes
  // gt_masks[sampled_targets], a full-resolution copy per proposal for BitMasks) or mask g
  long mi = g;
  bool bad = false;
  if (mask_index) {
    mi = mask_index[g];
    bad = mi < 0 || mi >= n_masks;  // torch indexing raises IndexError: flag it, write zeros
    if (bad) mi = 0;
  }
  const float* b = boxes + (long)g * 4;
  // roi_align.py:21-35 with aligned = True, spatial_scale = 1, sampling_ratio = 0
  const float roi_start_w = b[0] * 1.0f - 0.5f, roi_start_h = b[1] * 1.0f - 0.5f;
  const float roi_end_w = b[2] * 1.0f - 0.5f, roi_end_h = b[3] * 1.0f - 0.5f;
  const float roi_width = roi_end_w - roi_start_w, roi_height = roi_end_h - roi_start_h;
  const float bin_size_h = roi_height / (float)M, bin_size_w = roi_width / (float)M;
  const int grid_h = (int)ceilf(roi_height / (float)M), grid_w = (int)ceilf(roi_width / (float)M);
  const float count = (float)max(grid_h * grid_w, 1);
  const uint8_t* m = masks + mi * H * W;
  const long n = grid_h > 0 && grid_w > 0 ? (long)grid_h * grid_w : 0;
  // tier 1: strided partial sums over the same sample values
  float part = 0.f;
  if (live && !bad) {
    for (long s = sub; s < n; s += CROP_LANES) {
      const int iy = (int)(s / grid_w), ix = (int)(s - (long)iy * grid_w);
      const float yy = roi_start_h + (float)ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)grid_h;
      const float xx = roi_start_w + (float)pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)grid_w;
      part += crop_sample(m, H, W, yy, xx);
    }
  }
#pragma unroll
  for (int d = 1; d < CROP_LANES; d <<= 1) part += __shfl_xor(part, d, 64);
  if (!live || sub != 0) return;
  const long obin = (long)g * M * M + bin;
  if (bad) {
    if (bin == 0 && status) atomicOr(status, 1);
    out[obin] = 0;
    return;
  }
  const float approx = part / count;
  const float tol = (float)n * 1.2e-7f + 1e-6f;  // 2 n u + the rounding of the divide, against means in [0, 1]
  if (fabsf(approx - 0.5f) > tol) {
    out[obin] = approx >= 0.5f ? 1 : 0;
    return;
  }
  // tier 2: the reference's order (torchvision roi_align CPU: iy outer, ix inner, then / count)
  float v = 0.f;
  for (int iy = 0; iy < grid_h; iy++) {
    const float yy = roi_start_h + (float)ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)grid_h;
    for (int ix = 0; ix < grid_w; ix++) {
      const float xx = roi_start_w + (float)pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)grid_w;
      v += crop_sample(m, H, W, yy, xx);
    }
  }
  v /= count;
  out[obin] = v >= 0.5f ? 1 : 0;
}

}  // namespace d2amd

using namespace d2amd;

static int crop_launch(const uint8_t* masks, const float* boxes, const int64_t* mask_index, int n_masks, int G, int H,
                       int W, int mask_size, uint8_t* out, int* status, void* stream) {
  D2_CHECK_ARG(G >= 0 && H >= 0 && W >= 0 && mask_size > 0 && n_masks >= 0, "bitmask_crop_and_resize: bad shape");
  if (G == 0) return D2AMD_OK;
  D2_CHECK_ARG(H > 0 && W > 0 && masks && boxes && out, "bitmask_crop_and_resize: null pointer / empty mask");
  D2_CHECK_ARG(G <= 65535, "bitmask_crop_and_resize: too many boxes (%d)", G);
  dim3 grid(cdiv((long)mask_size * mask_size, CROP_THREADS / CROP_LANES), G);
  hipLaunchKernelGGL(bitmask_crop_kernel, grid, dim3(CROP_THREADS), 0, (hipStream_t)stream, masks, boxes, mask_index, n_masks, G, H,
                     W, mask_size, out, status);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

extern "C" int d2amd_bitmask_crop_and_resize(const uint8_t* masks, const float* boxes, int G, int H, int W,
                                             int mask_size, uint8_t* out, void* stream) {
  return crop_launch(masks, boxes, nullptr, G, G, H, W, mask_size, out, nullptr, stream);
}

extern "C" int d2amd_bitmask_crop_and_resize_indexed(const uint8_t* masks, int n_masks, const float* boxes,
                                                     const int64_t* mask_index, int n_boxes, int H, int W,
                                                     int mask_size, uint8_t* out, int* status, void* stream) {
  D2_CHECK_ARG(n_boxes == 0 || mask_index, "bitmask_crop_and_resize_indexed: null index");
  return crop_launch(masks, boxes, mask_index, n_masks, n_boxes, H, W, mask_size, out, status, stream);
}
