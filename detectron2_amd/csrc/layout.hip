// Layout conversion NCHW <-> NHWC as a batched 2-D transpose: [B][R][S] -> [B][S][R].
// Callers that hold NCHW feature maps (an unmodified Detectron2 model) reach the NHWC kernels of this library through
// it: the fused ROIPooler stages its features (R = C, S = H*W), lays dY out for the tile gather and hands results /
// gradients back in the caller's layout.  torch's own permuting copy (`.contiguous(memory_format=...)`, a strided
// elementwise kernel) ran at ~1 TB/s on these shapes and was the top kernel of the NCHW bench step (r02 profile).
// 64 x 64 tiles through LDS, 16-byte global accesses on both sides (a row of 64 two-byte elements = 128 B).
// Roofline: HBM, 2 x the tensor's bytes.
#include "common.h"

namespace d2amd {

constexpr int LT_TILE = 64;

template <typename T>
__global__ __launch_bounds__(256) void layout_transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int S,
                                                              int vec_in, int vec_out) {
  constexpr int V = 16 / (int)sizeof(T);           // elements per 16-B access
  constexpr int PITCH = LT_TILE + (int)(4 / sizeof(T)) + 1;  // odd number of 32-bit words per row: conflict-light columns
  __shared__ T tile[LT_TILE][PITCH];               // tile[r][s]
  const long b = blockIdx.z;
  const int s0 = blockIdx.x * LT_TILE, r0 = blockIdx.y * LT_TILE;
  const T* src = in + b * (long)R * S;
  T* dst = out + b * (long)R * S;
  const int tid = threadIdx.x;
  // read: rows r of the tile, V consecutive s per access
  constexpr int VPR = LT_TILE / V;  // accesses per tile row
  for (int i = tid; i < LT_TILE * VPR; i += 256) {
    const int r = i / VPR, sv = (i % VPR) * V;
    const int gr = r0 + r, gs = s0 + sv;
    if (gr >= R) continue;
    if (vec_in && gs + V <= S) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + (long)gr * S + gs);
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int q = 0; q < V; q++) tile[r][sv + q] = e[q];
    } else {
#pragma unroll
      for (int q = 0; q < V; q++)
        if (gs + q < S) tile[r][sv + q] = src[(long)gr * S + gs + q];
    }
  }
  __syncthreads();
  // write: rows s of the output, V consecutive r per access
  for (int i = tid; i < LT_TILE * VPR; i += 256) {
    const int s = i / VPR, rv = (i % VPR) * V;
    const int gs = s0 + s, gr = r0 + rv;
    if (gs >= S) continue;
    if (vec_out && gr + V <= R) {
      uint4 v;
      T* e = reinterpret_cast<T*>(&v);
#pragma unroll
      for (int q = 0; q < V; q++) e[q] = tile[rv + q][s];
      *reinterpret_cast<uint4*>(dst + (long)gs * R + gr) = v;
    } else {
#pragma unroll
      for (int q = 0; q < V; q++)
        if (gr + q < R) dst[(long)gs * R + gr + q] = tile[rv + q][s];
    }
  }
}

template <typename T>
static int layout_launch(const void* in, void* out, int B, int R, int S, hipStream_t st) {
  constexpr int V = 16 / (int)sizeof(T);
  dim3 grid(cdiv(S, LT_TILE), cdiv(R, LT_TILE), B);
  D2_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "transpose: tensor too large for the launch grid");
  // 16-B accesses need every row start aligned: the row length must be a multiple of V and the base 16-B aligned
  const int vin = (S % V == 0) && (((uintptr_t)in & 15) == 0), vout = (R % V == 0) && (((uintptr_t)out & 15) == 0);
  hipLaunchKernelGGL((layout_transpose_kernel<T>), grid, dim3(256), 0, st, (const T*)in, (T*)out, R, S, vin, vout);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

}  // namespace d2amd

using namespace d2amd;

extern "C" int d2amd_transpose_batched(const void* src, void* dst, int batch, int rows, int cols, int element_size,
                                       void* stream) {
  D2_CHECK_ARG(batch >= 0 && rows >= 0 && cols >= 0, "transpose: negative size");
  if ((long)batch * rows * cols == 0) return D2AMD_OK;
  D2_CHECK_ARG(src && dst && src != dst, "transpose: null pointer / in place");
  D2_CHECK_ARG((long)rows * cols < (1l << 31), "transpose: plane too large");
  if (element_size == 2) return layout_launch<uint16_t>(src, dst, batch, rows, cols, (hipStream_t)stream);
  if (element_size == 4) return layout_launch<uint32_t>(src, dst, batch, rows, cols, (hipStream_t)stream);
  set_error("transpose: element size %d (2 or 4)", element_size);
  return D2AMD_EUNSUPPORTED;
}
