// paste_masks_in_image: rasterise N soft masks (e.g. 28x28) into N full-image binary masks.
// Replaces detectron2/layers/mask_ops.py:17-147 (grid build + F.grid_sample + threshold +
// index_put) with ONE kernel: no fp32 grid tensor (8 B/px), no fp32 sampled image (4 B/px) --
// only the compulsory 1 B/px output is written, 16 B per lane (coalesced "scatter").
// Bit-exact vs the reference's CPU path: same region rule (skip_empty=True, one mask per chunk,
// mask_ops.py:38-43,116-119) and the same fp32 evaluation order as ATen's CPU grid_sampler
// (see oracle/d2_oracle.c, orc_paste_sample).  Compiled with FP contraction off; the FMAs
// below are explicit.
#pragma clang fp contract(off)
#include "common.h"

namespace d2amd {

constexpr int PASTE_BLOCK = 256;

template <typename T, int VEC>
__global__ __launch_bounds__(PASTE_BLOCK) void paste_masks_kernel(
    const T* __restrict__ masks, const float* __restrict__ boxes, int mh, int mw, int img_h, int img_w,
    float threshold, uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smask[];  // [mh*mw]
  const int n = blockIdx.y;
  const long plane = (long)img_h * img_w;
  const long p0 = ((long)blockIdx.x * PASTE_BLOCK) * VEC;  // first pixel of this block
  if (p0 >= plane) return;
  const long p1 = min(plane, p0 + (long)PASTE_BLOCK * VEC);
  const float x0 = boxes[n * 4 + 0], y0 = boxes[n * 4 + 1], x1 = boxes[n * 4 + 2], y1 = boxes[n * 4 + 3];
  // region touched by the reference's CPU path (mask_ops.py:38-43), ints after clamp
  float fx0 = floorf(x0) - 1.f, fy0 = floorf(y0) - 1.f, fx1 = ceilf(x1) + 1.f, fy1 = ceilf(y1) + 1.f;
  fx0 = fx0 < 0.f ? 0.f : fx0;
  fy0 = fy0 < 0.f ? 0.f : fy0;
  fx1 = fx1 > (float)img_w ? (float)img_w : fx1;
  fy1 = fy1 > (float)img_h ? (float)img_h : fy1;
  const int rx0 = (int)fx0, ry0 = (int)fy0, rx1 = (int)fx1, ry1 = (int)fy1;
  const int row_first = (int)(p0 / img_w), row_last = (int)((p1 - 1) / img_w);
  const bool block_live = (row_last >= ry0) && (row_first < ry1) && (rx1 > rx0);  // uniform
  if (block_live) {
    for (int i = threadIdx.x; i < mh * mw; i += PASTE_BLOCK) smask[i] = to_f32(masks[(long)n * mh * mw + i]);
    __syncthreads();
  }
  const long p = p0 + (long)threadIdx.x * VEC;
  if (p >= plane) return;
  uint8_t res[VEC];
#pragma unroll
  for (int v = 0; v < VEC; v++) res[v] = 0;
  if (block_live) {
    const float sx = (float)mw / 2.f, sy = (float)mh / 2.f;
    const float dxw = x1 - x0, dyh = y1 - y0;
#pragma unroll
    for (int v = 0; v < VEC; v++) {
      const long q = p + v;
      if (q >= plane) break;
      const int py = (int)(q / img_w), px = (int)(q - (long)py * img_w);
      if (py < ry0 || py >= ry1 || px < rx0 || px >= rx1) continue;
      const float gy = ((float)py + 0.5f - y0) / dyh * 2.f - 1.f;
      const float gx = ((float)px + 0.5f - x0) / dxw * 2.f - 1.f;
      const float ix = __builtin_fmaf(gx + 1.f, sx, -0.5f);
      const float iy = __builtin_fmaf(gy + 1.f, sy, -0.5f);
      const float flx = floorf(ix), fly = floorf(iy);
      float val = 0.f;
      if (flx > -4.0e8f && flx < 4.0e8f && fly > -4.0e8f && fly < 4.0e8f) {
        const int xw = (int)flx, yn = (int)fly;
        const float w = ix - flx, e = 1.f - w;
        const float nn = iy - fly, s = 1.f - nn;
        const float nw = s * e, ne = s * w, sw = nn * e, se = nn * w;
        const bool x0ok = xw >= 0 && xw < mw, x1ok = xw + 1 >= 0 && xw + 1 < mw;
        const bool y0ok = yn >= 0 && yn < mh, y1ok = yn + 1 >= 0 && yn + 1 < mh;
        const float v00 = (y0ok && x0ok) ? smask[yn * mw + xw] : 0.f;
        const float v01 = (y0ok && x1ok) ? smask[yn * mw + xw + 1] : 0.f;
        const float v10 = (y1ok && x0ok) ? smask[(yn + 1) * mw + xw] : 0.f;
        const float v11 = (y1ok && x1ok) ? smask[(yn + 1) * mw + xw + 1] : 0.f;
        val = v00 * nw;
        val = __builtin_fmaf(v01, ne, val);
        val = __builtin_fmaf(v10, sw, val);
        val = __builtin_fmaf(v11, se, val);
      }
      res[v] = threshold >= 0.f ? (uint8_t)(val >= threshold) : (uint8_t)(int)(val * 255.f);
    }
  }
  uint8_t* o = out + (long)n * plane + p;
  if (VEC == 16) {
    uint4 pk;
    __builtin_memcpy(&pk, res, 16);
    *reinterpret_cast<uint4*>(o) = pk;
  } else if (VEC == 4) {
    uint32_t pk;
    __builtin_memcpy(&pk, res, 4);
    *reinterpret_cast<uint32_t*>(o) = pk;
  } else {
    o[0] = res[0];
  }
}

template <typename T>
static int launch_paste(const void* masks, const float* boxes, int n, int mh, int mw, int img_h, int img_w,
                        float threshold, uint8_t* out, hipStream_t s) {
  const long plane = (long)img_h * img_w;
  const size_t lds = (size_t)mh * mw * sizeof(float);
  const bool a16 = (plane % 16 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const bool a4 = (plane % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 3) == 0);
  dim3 block(PASTE_BLOCK);
  if (a16) {
    dim3 grid(cdiv(plane, (long)PASTE_BLOCK * 16), n);
    hipLaunchKernelGGL((paste_masks_kernel<T, 16>), grid, block, lds, s, (const T*)masks, boxes, mh, mw, img_h,
                       img_w, threshold, out);
  } else if (a4) {
    dim3 grid(cdiv(plane, (long)PASTE_BLOCK * 4), n);
    hipLaunchKernelGGL((paste_masks_kernel<T, 4>), grid, block, lds, s, (const T*)masks, boxes, mh, mw, img_h,
                       img_w, threshold, out);
  } else {
    dim3 grid(cdiv(plane, (long)PASTE_BLOCK), n);
    hipLaunchKernelGGL((paste_masks_kernel<T, 1>), grid, block, lds, s, (const T*)masks, boxes, mh, mw, img_h,
                       img_w, threshold, out);
  }
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
