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

// Up to LT_MAX tensors of one batch size in ONE launch (the five FPN levels of an NCHW model, their gradients on the way
// back): five dependent launches of 3-40 us, the small ones all launch latency, were 0.17 ms of the NCHW drop-in step.
constexpr int LT_MAX = 8;
struct LayoutMulti {
  const void* src[LT_MAX];
  void* dst[LT_MAX];
  int R[LT_MAX], S[LT_MAX], tiles_x[LT_MAX], first[LT_MAX + 1], vin[LT_MAX], vout[LT_MAX];
  int count;
};
template <typename T>
__global__ __launch_bounds__(256) void layout_transpose_multi_kernel(LayoutMulti m) {
  constexpr int V = 16 / (int)sizeof(T);
  constexpr int PITCH = LT_TILE + (int)(4 / sizeof(T)) + 1;
  __shared__ T tile[LT_TILE][PITCH];
  int t = 0;
#pragma unroll
  for (int q = 1; q < LT_MAX; q++)
    if (q < m.count && (int)blockIdx.x >= m.first[q]) t = q;  // (constant indices: the table stays in SGPRs)
  const T* in = nullptr; T* out = nullptr;
  int R = 0, S = 0, tx = 1, first = 0, vec_in = 0, vec_out = 0;
#pragma unroll
  for (int q = 0; q < LT_MAX; q++)
    if (q == t) { in = (const T*)m.src[q]; out = (T*)m.dst[q]; R = m.R[q]; S = m.S[q]; tx = m.tiles_x[q]; first = m.first[q];
                  vec_in = m.vin[q]; vec_out = m.vout[q]; }
  const int tile_id = (int)blockIdx.x - first;
  const long b = blockIdx.z;
  const int s0 = (tile_id % tx) * LT_TILE, r0 = (tile_id / tx) * LT_TILE;
  const T* src = in + b * (long)R * S;
  T* dst = out + b * (long)R * S;
  const int tid = threadIdx.x;
  constexpr int VPR = LT_TILE / V;
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

extern "C" int d2amd_transpose_multi(const void* const* src, void* const* dst, const int* rows, const int* cols, int count,
                                     int batch, int element_size, void* stream) {
  D2_CHECK_ARG(count >= 0 && count <= LT_MAX && batch >= 0, "transpose_multi: %d tensors (max %d)", count, LT_MAX);
  if (count == 0 || batch == 0) return D2AMD_OK;
  D2_CHECK_ARG(src && dst && rows && cols, "transpose_multi: null pointer");
  D2_CHECK_ARG(element_size == 2 || element_size == 4, "transpose_multi: element size %d (2 or 4)", element_size);
  D2_CHECK_ARG(batch <= 65535, "transpose_multi: batch too large for the launch grid");
  const int V = 16 / element_size;
  LayoutMulti m{};
  long tiles = 0;
  for (int t = 0; t < count; t++) {
    D2_CHECK_ARG(rows[t] > 0 && cols[t] > 0 && (long)rows[t] * cols[t] < (1l << 31), "transpose_multi: bad plane %d", t);
    D2_CHECK_ARG(src[t] && dst[t] && src[t] != dst[t], "transpose_multi: null pointer / in place (tensor %d)", t);
    m.src[m.count] = src[t]; m.dst[m.count] = dst[t];
    m.R[m.count] = rows[t]; m.S[m.count] = cols[t];
    m.tiles_x[m.count] = cdiv(cols[t], LT_TILE);
    m.first[m.count] = (int)tiles;
    m.vin[m.count] = (cols[t] % V == 0) && (((uintptr_t)src[t] & 15) == 0);
    m.vout[m.count] = (rows[t] % V == 0) && (((uintptr_t)dst[t] & 15) == 0);
    tiles += (long)cdiv(cols[t], LT_TILE) * cdiv(rows[t], LT_TILE);
    m.count++;
  }
  D2_CHECK_ARG(tiles < (1l << 31), "transpose_multi: too many tiles");
  m.first[m.count] = (int)tiles;
  const dim3 grid((unsigned)tiles, 1, batch);
  if (element_size == 2) hipLaunchKernelGGL((layout_transpose_multi_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, m);
  else hipLaunchKernelGGL((layout_transpose_multi_kernel<uint32_t>), grid, dim3(256), 0, (hipStream_t)stream, m);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}
