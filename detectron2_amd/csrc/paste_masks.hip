// paste_masks_in_image: rasterise N soft masks (e.g. 28x28) into N full-image binary masks.
// Replaces detectron2/layers/mask_ops.py:17-147 (grid build + F.grid_sample + threshold +
// index_put) with ONE kernel that writes every output byte once (box regions sampled, the rest zeros): no fp32 grid
// tensor (8 B/px), no fp32 sampled image (4 B/px), no separate zero fill -- only the compulsory 1 B/px output.
// Bit-exact vs the reference's CPU path: same region rule (skip_empty=True, one mask per chunk,
// mask_ops.py:38-43,116-119) and the same fp32 evaluation order as ATen's CPU grid_sampler
// (see oracle/d2_oracle.c, orc_paste_sample).  Compiled with FP contraction off; the FMAs
// below are explicit.
#pragma clang fp contract(off)
#include "common.h"

namespace d2amd {

constexpr int PASTE_BLOCK = 256;
constexpr int PASTE_ROWS = 8;  // image rows per workgroup: 8 rows x 32 column lanes

// The output is 1 B / px and almost all of it is zeros (the boxes of the BASELINE shape cover ~5 % of the
// N x H x W pixels).  The first version gave every thread 16 consecutive pixels of the flattened image: a wave
// whose 1,024 pixels touched a box row evaluated the sampling path for all 16 of its iterations while the lanes
// outside the box columns idled -- 36 of its 55 us (zero fill alone: 18 us, profiles/r01/v8_paste_bench.txt).
// r01-r03: (1) the whole output zero-filled (15 us for 107 MB), (2) this kernel over the box REGIONS only, densely:
// workgroup = 8 image rows of one mask, lane = (row, column mod 32), the row terms (gy, iy, row weights) once per
// thread, then a walk over the region's columns: 28.6 us per image of 100 detections = 0.47 of the HBM peak, the
// region pixels written twice.  r04: the same workgroups also write the zeros of their rows (paste_zero_span) -- one
// launch, every byte once.  Same fp32 evaluation order as before (bit-exact vs ATen's CPU grid_sampler).
// zero `n` bytes at `p` with the 32 lanes of one image row's half-wave: single bytes up to the first 16-B boundary,
// 16-B stores, single bytes behind the last one (rows are img_w bytes apart: any alignment)
__device__ __forceinline__ void paste_zero_span(uint8_t* p, int n, int lane32) {
  if (n <= 0) return;
  const int head = min(n, (int)((16u - (unsigned)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u));
  if (lane32 < head) p[lane32] = 0;
  const int body = (n - head) >> 4;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  for (int i = lane32; i < body; i += 32) q[i] = make_uint4(0u, 0u, 0u, 0u);
  const int done = head + (body << 4);
  if (lane32 < n - done) p[done + lane32] = 0;
}

template <typename T>
__global__ __launch_bounds__(PASTE_BLOCK) void paste_region_kernel(
    const T* __restrict__ masks, const float* __restrict__ boxes, int mh, int mw, int img_h, int img_w,
    float threshold, uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smask[];  // [mh*mw]
  const int n = blockIdx.y;
  const float x0 = boxes[n * 4 + 0], y0 = boxes[n * 4 + 1], x1 = boxes[n * 4 + 2], y1 = boxes[n * 4 + 3];
  // Region that can hold a non-zero sample.  The reference's CPU path crops to [floor(x0) - 1, ceil(x1) + 1)
  // (mask_ops.py:38-43, skip_empty=True), its device path samples the whole image (skip_empty=False, :116-119):
  // with zero padding, bilinear values are non-zero up to half a mask pixel = extent / (2 M) image pixels outside
  // the box (14 px for an 800 px box, M = 28).  Invisible at threshold 0.5 for masks in [0, 1], visible for lower
  // thresholds and the uint8 soft output -- so the region is grown by that margin: everything outside it is exactly
  // 0 in the full-image evaluation too (the memset's value).
  const float mx = ceilf(fabsf(x1 - x0) / (2.f * (float)mw)) + 1.f, my = ceilf(fabsf(y1 - y0) / (2.f * (float)mh)) + 1.f;
  float fx0 = floorf(fminf(x0, x1)) - 1.f - mx, fy0 = floorf(fminf(y0, y1)) - 1.f - my;
  float fx1 = ceilf(fmaxf(x0, x1)) + 1.f + mx, fy1 = ceilf(fmaxf(y0, y1)) + 1.f + my;
  if (!(fx0 == fx0 && fx1 == fx1 && fy0 == fy0 && fy1 == fy1)) { fx0 = fy0 = 0.f; fx1 = (float)img_w; fy1 = (float)img_h; }  // NaN box
  fx0 = fx0 < 0.f ? 0.f : fx0;
  fy0 = fy0 < 0.f ? 0.f : fy0;
  fx1 = fx1 > (float)img_w ? (float)img_w : fx1;
  fy1 = fy1 > (float)img_h ? (float)img_h : fy1;
  const int rx0 = (int)fx0, ry0 = (int)fy0, rx1 = (int)fx1, ry1 = (int)fy1;
  const int row0 = blockIdx.x * PASTE_ROWS;
  const bool has_region = !(row0 >= ry1 || row0 + PASTE_ROWS <= ry0 || rx1 <= rx0);  // uniform
  if (has_region) {
    for (int i = threadIdx.x; i < mh * mw; i += PASTE_BLOCK) smask[i] = to_f32(masks[(long)n * mh * mw + i]);
    __syncthreads();
  }
  // EVERY output byte is written exactly once, by this launch: the region's pixels by the sampling walk below,
  // everything else of the row as zeros (16-B stores) -- there is no separate zero fill of the 107 MB any more
  const int py = row0 + (threadIdx.x >> 5), lane32 = threadIdx.x & 31;
  if (py >= img_h) return;
  {
    uint8_t* zrow = out + ((long)n * img_h + py) * img_w;
    if (!has_region || py < ry0 || py >= ry1) {
      paste_zero_span(zrow, img_w, lane32);
      return;
    }
    paste_zero_span(zrow, rx0, lane32);
    paste_zero_span(zrow + rx1, img_w - rx1, lane32);
  }
  const float sx = (float)mw / 2.f, sy = (float)mh / 2.f;
  const float dxw = x1 - x0, dyh = y1 - y0;
  const float gy = ((float)py + 0.5f - y0) / dyh * 2.f - 1.f;
  const float iy = __builtin_fmaf(gy + 1.f, sy, -0.5f);
  const float fly = floorf(iy);
  const bool yfin = fly > -4.0e8f && fly < 4.0e8f;
  const int yn = yfin ? (int)fly : 0;
  const float nn = iy - fly, s_ = 1.f - nn;
  const bool y0ok = yn >= 0 && yn < mh, y1ok = yn + 1 >= 0 && yn + 1 < mh;
  uint8_t* orow = out + ((long)n * img_h + py) * img_w;
  for (int px = rx0 + (threadIdx.x & 31); px < rx1; px += 32) {
    const float gx = ((float)px + 0.5f - x0) / dxw * 2.f - 1.f;
    const float ix = __builtin_fmaf(gx + 1.f, sx, -0.5f);
    const float flx = floorf(ix);
    float val = 0.f;
    if (flx > -4.0e8f && flx < 4.0e8f && yfin) {
      const int xw = (int)flx;
      const float w = ix - flx, e = 1.f - w;
      const float nw = s_ * e, ne = s_ * w, sw = nn * e, se = nn * w;
      const bool x0ok = xw >= 0 && xw < mw, x1ok = xw + 1 >= 0 && xw + 1 < mw;
      const float v00 = (y0ok && x0ok) ? smask[yn * mw + xw] : 0.f;
      const float v01 = (y0ok && x1ok) ? smask[yn * mw + xw + 1] : 0.f;
      const float v10 = (y1ok && x0ok) ? smask[(yn + 1) * mw + xw] : 0.f;
      const float v11 = (y1ok && x1ok) ? smask[(yn + 1) * mw + xw + 1] : 0.f;
      val = v00 * nw;
      val = __builtin_fmaf(v01, ne, val);
      val = __builtin_fmaf(v10, sw, val);
      val = __builtin_fmaf(v11, se, val);
    }
    orow[px] = threshold >= 0.f ? (uint8_t)(val >= threshold) : (uint8_t)(int)(val * 255.f);
  }
}

template <typename T>
static int launch_paste(const void* masks, const float* boxes, int n, int mh, int mw, int img_h, int img_w,
                        float threshold, uint8_t* out, hipStream_t s) {
  const size_t lds = (size_t)mh * mw * sizeof(float);
  const bool timed = timing_begin("paste_masks", s);
  dim3 grid(cdiv(img_h, PASTE_ROWS), n);
  hipLaunchKernelGGL((paste_region_kernel<T>), grid, dim3(PASTE_BLOCK), lds, s, (const T*)masks, boxes, mh, mw, img_h,
                     img_w, threshold, out);
  if (timed) timing_end("paste_masks", s);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

}  // namespace d2amd

using namespace d2amd;

extern "C" int d2amd_paste_masks(const void* masks, const float* boxes, int n, int mh, int mw, int img_h,
                                 int img_w, float threshold, uint8_t* out, int mask_dtype, void* stream) {
  D2_CHECK_ARG(n >= 0 && mh > 0 && mw > 0 && img_h >= 0 && img_w >= 0, "paste_masks: bad shape");
  D2_CHECK_ARG(mh == mw, "Only square mask predictions are supported");  // mask_ops.py:102
  if (n == 0 || img_h == 0 || img_w == 0) return D2AMD_OK;
  D2_CHECK_ARG(masks && boxes && out, "paste_masks: null pointer");
  D2_CHECK_ARG(n <= 65535, "paste_masks: n > 65535 unsupported");
  D2_CHECK_ARG((size_t)mh * mw * 4 <= 64 * 1024, "paste_masks: mask too large for LDS staging");
  hipStream_t s = (hipStream_t)stream;
  return D2_DISPATCH_DTYPE(mask_dtype, [&]() -> int {
    return launch_paste<scalar_t>(masks, boxes, n, mh, mw, img_h, img_w, threshold, out, s);
  });
}
