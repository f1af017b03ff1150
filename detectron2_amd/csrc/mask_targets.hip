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
// only the comparison `mean >= 0.5` reaches the output.  A box of 600 px has 22 x 22 = 484 samples per bin; walked
// in order by one thread that is a 484-deep chain of dependent byte loads (r01: 117 us for 128 ROIs), and even in
// parallel ~140 VALU instructions and 4 scattered byte loads per sample.
// Tier 1 (decides almost every bin).  The sampling grid is axis aligned, so the sum over a bin's samples factorises
// exactly (real arithmetic):
//     sum_s value_s = sum_ix [ wlo(ix) CS[lo(ix)] + whi(ix) CS[hi(ix)] ],   CS[c] = sum_r Wy[r] mask[r][c],
// Wy[r] = total weight the bin's gh sample ROWS put on pixel row r (hy on y_low, ly on y_high; nothing for a sample
// row outside (-1, H)), and (lo, hi, wlo, whi)(ix) the reference's own per-column expression.  Wy and CS are shared
// by the 28 bins of a bin ROW: one workgroup = (ROI, bin row) builds Wy (gh evaluations), the column sums CS over
// the row's pixel band (threads along the columns: coalesced byte loads, every mask byte of the band read once), and
// the gw column samples of each bin pick their two CS entries.  Work ~ the ROI's pixel area, not 4 x its samples.
// In fp32 the two sums differ by at most (2 n + 4 (gh + gw) + 16) u of a mean in [0, 1] (u = 2^-24).  If
// the tier-1 mean is farther than that from 0.5, the reference's sequential sum is on the same side: decided.
// Tier 2 (a mean within ~1e-4 of 0.5, or a band too large for the LDS tables): the workgroup produces the bin's
// sample VALUES 256 at a time and one thread adds them in the reference's order (iy outer, ix inner, then / count).
// The result is bit-identical to the sequential evaluation for every input.
// Workgroup size.  A (box, bin row) is a chain of dependent steps (index -> box -> tables -> band -> bins) whose widest one
// uses ~100-200 threads (the band's columns): with 256-thread workgroups the 7,168 of a step's 256 boxes ran in 3.5 rounds
// of 8 per CU.  MEASURED (r06, same box): 256 threads 26.2 us alone / connected step 0.3156-0.3181 ms | 128: 17.5 us /
// 0.3065-0.3106 | 64: 19.1 us / 0.3131-0.3164.  (Several bin rows per workgroup -- the prologue once -- was worse: 2 rows
// 29 against 31 us alone, 4 rows 34, 7 rows 50.)
constexpr int CROP_THREADS = 128;
constexpr int CROP_LPB = CROP_THREADS / 64;  // lanes per bin in tier 1 (64 bins per row at most)
constexpr int CROP_MAX_IMAGES = 64;
constexpr int CROP_ROWS = 64;    // pixel rows of a bin row's band the Wy table holds
constexpr int CROP_COLS = 2048;  // pixel columns of the band the CS table holds
constexpr int CROP_MAX_M = 64;   // bins per row with a tier-1 lane group (mask_size <= 64; 28 in Mask R-CNN)

struct CropBatch {  // the images of one launch: box rows [end[i-1], end[i]) belong to image i
  const uint8_t* masks[CROP_MAX_IMAGES];
  const float* boxes[CROP_MAX_IMAGES];
  const int64_t* index[CROP_MAX_IMAGES];  // per image: matched mask of every box, or nullptr (box g crops mask g)
  int n_masks[CROP_MAX_IMAGES];
  int end[CROP_MAX_IMAGES];
  int n;
};

// one axis of crop_sample(): sample coordinate -> (low pixel, high pixel, weight on low, weight on high, in range)
struct CropAxis { int lo, hi; float wlo, whi; bool in; };
__device__ __forceinline__ CropAxis crop_axis(float y, int H) {
  CropAxis a;
  a.in = !(y < -1.0f || y > (float)H);
  if (y < 0.f) y = 0.f;
  int y_low = (int)y, y_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  const float ly = y - (float)y_low;
  a.lo = y_low; a.hi = y_high;
  a.whi = ly; a.wlo = (float)(1. - (double)ly);
  return a;
}

__global__ __launch_bounds__(CROP_THREADS) void bitmask_crop_kernel(CropBatch B, int H, int W, int M,
                                                                   uint8_t* __restrict__ out,
                                                                   int* __restrict__ status) {
  __shared__ float s_wy[CROP_ROWS];
  __shared__ float4 s_ys[CROP_ROWS];      // per y sample {low pixel, high pixel, weights}
  __shared__ float s_cs[CROP_COLS];
  __shared__ __attribute__((aligned(16))) float s_val[CROP_THREADS];   // tier 2 hand-off
  __shared__ int s_undecided[CROP_MAX_M];
  const int tid = threadIdx.x;
  const int g = blockIdx.y, ph = blockIdx.x;  // box row of the batch, bin row
  int img = 0;
#pragma unroll 1
  while (img + 1 < B.n && g >= B.end[img]) img++;
  const int gl = g - (img ? B.end[img - 1] : 0);
  // box g crops mask index[g] (the matched ground truth of a sampled proposal: roi_heads.py:280-291 indexes
  // gt_masks[sampled_targets], a full-resolution copy per proposal for BitMasks) or mask g
  long mi = gl;
  if (B.index[img]) {
    mi = B.index[img][gl];
    if (mi < 0 || mi >= B.n_masks[img]) {  // torch indexing raises IndexError: flag it, write zeros (uniform)
      if (ph == 0 && tid == 0 && status) atomicOr(status, 1);
      for (int pw = tid; pw < M; pw += CROP_THREADS) out[((long)g * M + ph) * M + pw] = 0;
      return;
    }
  }
  const float* b = B.boxes[img] + (long)gl * 4;
  // roi_align.py:21-35 with aligned = True, spatial_scale = 1, sampling_ratio = 0
  const float roi_start_w = b[0] * 1.0f - 0.5f, roi_start_h = b[1] * 1.0f - 0.5f;
  const float roi_end_w = b[2] * 1.0f - 0.5f, roi_end_h = b[3] * 1.0f - 0.5f;
  const float roi_width = roi_end_w - roi_start_w, roi_height = roi_end_h - roi_start_h;
  const float bin_size_h = roi_height / (float)M, bin_size_w = roi_width / (float)M;
  const int grid_h = (int)ceilf(roi_height / (float)M), grid_w = (int)ceilf(roi_width / (float)M);
  const float count = (float)max(grid_h * grid_w, 1);
  const long n = grid_h > 0 && grid_w > 0 ? (long)grid_h * grid_w : 0;
  const int ni = (int)n;  // < 2^31: the host bounds H * W, and grid <= (image extent / M) + 1 per axis
  const uint8_t* m = B.masks[img] + mi * H * W;
  const float y_base = roi_start_h + (float)ph * bin_size_h;
  uint8_t* orow = out + ((long)g * M + ph) * M;
  if (ni == 0) {  // no samples: the reference's mean is 0 / 1 -> below the threshold (uniform)
    for (int pw = tid; pw < M; pw += CROP_THREADS) orow[pw] = 0;
    return;
  }
  for (int pw = tid; pw < min(M, CROP_MAX_M); pw += CROP_THREADS) s_undecided[pw] = 1;
  // the band of the bin row: pixel rows of its y samples x pixel columns of the ROI's x samples
  const CropAxis x_first = crop_axis(roi_start_w + .5f * bin_size_w / (float)grid_w, W);
  const CropAxis x_last = crop_axis(roi_start_w + (float)(M - 1) * bin_size_w +
                                    ((float)(grid_w - 1) + .5f) * bin_size_w / (float)grid_w, W);
  const int c0 = x_first.lo, ncols = x_last.hi - x_first.lo + 1;
  const bool fast = M <= CROP_MAX_M && grid_h + 2 <= CROP_ROWS && ncols >= 1 && ncols <= CROP_COLS;  // uniform
  if (fast) {
    // ---- tier 1 ----
    if (tid < grid_h) {
      const CropAxis a = crop_axis(y_base + ((float)tid + .5f) * bin_size_h / (float)grid_h, H);
      s_ys[tid] = make_float4(__int_as_float(a.lo), __int_as_float(a.hi), a.in ? a.wlo : 0.f, a.in ? a.whi : 0.f);
    }
    __syncthreads();
    const int r0 = __float_as_int(s_ys[0].x);
    const int ny = __float_as_int(s_ys[grid_h - 1].y) - r0 + 1;  // <= grid_h + 2 (clamping only shrinks it)
    if (tid < ny) {
      float acc = 0.f;
      for (int i = 0; i < grid_h; i++) {
        const float4 e = s_ys[i];
        acc += (__float_as_int(e.x) - r0 == tid ? e.z : 0.f) + (__float_as_int(e.y) - r0 == tid ? e.w : 0.f);
      }
      s_wy[tid] = acc;
    }
    __syncthreads();
    // column sums: thread = pixel column (consecutive threads read consecutive bytes), 4 rows in flight
    const uint8_t* band = m + (long)r0 * W + c0;
    for (int c = tid; c < ncols; c += CROP_THREADS) {
      float acc = 0.f;
      int r = 0;
      for (; r + 4 <= ny; r += 4) {
        const uint8_t v0 = band[(long)r * W + c], v1 = band[(long)(r + 1) * W + c];
        const uint8_t v2 = band[(long)(r + 2) * W + c], v3 = band[(long)(r + 3) * W + c];
        acc += (v0 ? s_wy[r] : 0.f) + (v1 ? s_wy[r + 1] : 0.f) + (v2 ? s_wy[r + 2] : 0.f) + (v3 ? s_wy[r + 3] : 0.f);
      }
      for (; r < ny; r++) acc += band[(long)r * W + c] ? s_wy[r] : 0.f;
      s_cs[c] = acc;
    }
    __syncthreads();
    // bins: CROP_LPB lanes per bin (64 bins max), each lane takes every CROP_LPB-th column sample
    const int pw = tid / CROP_LPB, sub = tid % CROP_LPB;
    float part = 0.f;
    if (pw < M) {
      const float x_base = roi_start_w + (float)pw * bin_size_w;
      for (int ix = sub; ix < grid_w; ix += CROP_LPB) {
        const CropAxis a = crop_axis(x_base + ((float)ix + .5f) * bin_size_w / (float)grid_w, W);
        if (a.in) part += a.wlo * s_cs[a.lo - c0] + a.whi * s_cs[a.hi - c0];
      }
    }
#pragma unroll
    for (int d = 1; d < CROP_LPB; d <<= 1) part += __shfl_xor(part, d, 64);
    if (pw < M && sub == 0) {
      const float approx = part / count;
      const float tol = (float)(2 * ni + 4 * (grid_h + grid_w) + 16) * 1.2e-7f + 1e-6f;  // 2 x the bound, + the divide
      if (fabsf(approx - 0.5f) > tol) {
        orow[pw] = approx >= 0.5f ? 1 : 0;
        s_undecided[pw] = 0;
      }
    }
    __syncthreads();
  } else {
    __syncthreads();  // s_undecided is written by thread pw and read by every thread below (and a stale 0 from an
                      // earlier workgroup would make some waves skip the loop's barriers)
  }
  // ---- tier 2: the reference's order, for the bins tier 1 left open (all of them when the band does not fit) ----
  for (int pw = 0; pw < M; pw++) {
    if (pw < CROP_MAX_M && !s_undecided[pw]) continue;  // uniform
    const float x_base = roi_start_w + (float)pw * bin_size_w;
    float v = 0.f;
    for (int s0 = 0; s0 < ni; s0 += CROP_THREADS) {
      const int sidx = s0 + tid;
      const bool on = sidx < ni;
      const int iy = on ? sidx / grid_w : 0, ix = on ? sidx - iy * grid_w : 0;
      const float yy = y_base + ((float)iy + .5f) * bin_size_h / (float)grid_h;
      const float xx = x_base + ((float)ix + .5f) * bin_size_w / (float)grid_w;
      s_val[tid] = on ? crop_sample(m, H, W, yy, xx) : 0.f;
      __syncthreads();
      if (tid == 0) {
        // the entries past the last sample are 0: adding them leaves v unchanged (v >= 0), so the chain runs over
        // whole float4 reads, 16 values per batch of LDS loads
        const int cnt4 = (min(CROP_THREADS, ni - s0) + 3) >> 2;
        const float4* q = reinterpret_cast<const float4*>(s_val);
#pragma unroll 4
        for (int j = 0; j < cnt4; j++) {
          const float4 e = q[j];
          v += e.x; v += e.y; v += e.z; v += e.w;
        }
      }
      __syncthreads();
    }
    if (tid == 0) orow[pw] = (v / count) >= 0.5f ? 1 : 0;
  }
}

}  // namespace d2amd

using namespace d2amd;

static int crop_launch(const CropBatch& B, int total, int H, int W, int mask_size, uint8_t* out, int* status,
                       void* stream) {
  D2_CHECK_ARG(H >= 0 && W >= 0 && mask_size > 0, "bitmask_crop_and_resize: bad shape");
  if (total == 0) return D2AMD_OK;
  D2_CHECK_ARG(H > 0 && W > 0 && out, "bitmask_crop_and_resize: null pointer / empty mask");
  D2_CHECK_ARG(total <= 65535, "bitmask_crop_and_resize: too many boxes (%d)", total);
  D2_CHECK_ARG((long)H * W < (1l << 31), "bitmask_crop_and_resize: mask too large");
  D2_CHECK_ARG(mask_size <= 65535, "bitmask_crop_and_resize: mask_size too large");
  dim3 grid(mask_size, total);  // one workgroup per (box, bin row)
  hipLaunchKernelGGL(bitmask_crop_kernel, grid, dim3(CROP_THREADS), 0, (hipStream_t)stream, B, H, W, mask_size, out,
                     status);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

extern "C" int d2amd_bitmask_crop_and_resize(const uint8_t* masks, const float* boxes, int G, int H, int W,
                                             int mask_size, uint8_t* out, void* stream) {
  D2_CHECK_ARG(G >= 0, "bitmask_crop_and_resize: bad shape");
  D2_CHECK_ARG(G == 0 || (masks && boxes), "bitmask_crop_and_resize: null pointer / empty mask");
  CropBatch B{};
  B.n = 1; B.masks[0] = masks; B.boxes[0] = boxes; B.index[0] = nullptr; B.n_masks[0] = G; B.end[0] = G;
  return crop_launch(B, G, H, W, mask_size, out, nullptr, stream);
}

extern "C" int d2amd_bitmask_crop_and_resize_indexed(const uint8_t* masks, int n_masks, const float* boxes,
                                                     const int64_t* mask_index, int n_boxes, int H, int W,
                                                     int mask_size, uint8_t* out, int* status, void* stream) {
  D2_CHECK_ARG(n_boxes >= 0 && n_masks >= 0, "bitmask_crop_and_resize_indexed: bad shape");
  D2_CHECK_ARG(n_boxes == 0 || (mask_index && masks && boxes), "bitmask_crop_and_resize_indexed: null pointer");
  CropBatch B{};
  B.n = 1; B.masks[0] = masks; B.boxes[0] = boxes; B.index[0] = mask_index; B.n_masks[0] = n_masks; B.end[0] = n_boxes;
  return crop_launch(B, n_boxes, H, W, mask_size, out, status, stream);
}

extern "C" int d2amd_bitmask_crop_and_resize_batch(int num_images, const uint8_t* const* masks, const int* n_masks,
                                                   const float* const* boxes, const int64_t* const* mask_index,
                                                   const int* n_boxes, int H, int W, int mask_size, uint8_t* out,
                                                   int* status, void* stream) {
  D2_CHECK_ARG(num_images >= 0 && num_images <= CROP_MAX_IMAGES, "bitmask_crop_and_resize_batch: %d images (max %d)",
               num_images, CROP_MAX_IMAGES);
  D2_CHECK_ARG(num_images == 0 || (masks && n_masks && boxes && n_boxes), "bitmask_crop_and_resize_batch: null pointer");
  CropBatch B{};
  int total = 0;
  for (int i = 0; i < num_images; i++) {
    D2_CHECK_ARG(n_boxes[i] >= 0 && n_masks[i] >= 0, "bitmask_crop_and_resize_batch: negative count");
    D2_CHECK_ARG(n_boxes[i] == 0 || (masks[i] && boxes[i]), "bitmask_crop_and_resize_batch: null pointer (image %d)", i);
    D2_CHECK_ARG(mask_index ? (n_boxes[i] == 0 || mask_index[i] != nullptr) : n_boxes[i] == n_masks[i],
                 "bitmask_crop_and_resize_batch: image %d has %d boxes for %d masks and no index", i, n_boxes[i], n_masks[i]);
    B.masks[B.n] = masks[i]; B.boxes[B.n] = boxes[i]; B.index[B.n] = mask_index ? mask_index[i] : nullptr;
    B.n_masks[B.n] = n_masks[i];
    total += n_boxes[i];
    B.end[B.n] = total;
    B.n++;
  }
  if (B.n == 0) return D2AMD_OK;
  return crop_launch(B, total, H, W, mask_size, out, status, stream);
}
