// Pairwise box IoU kernels (axis-aligned and rotated).  Bit-exact vs the reference CPU ops:
// this translation unit is compiled with -ffp-contract=off and IEEE division.
//   d2amd_pairwise_iou    <- detectron2/structures/boxes.py:312-377
//   d2amd_box_iou_rotated <- torch.ops.detectron2.box_iou_rotated (csrc/box_iou_rotated/*)
// Roofline: HBM write of the n x m fp32 matrix (axis-aligned), VALU (rotated).
#pragma clang fp contract(off)
#include "common.h"
#include "rotated_iou.h"

namespace d2amd {

// torch.min / torch.max propagate NaN
__device__ __forceinline__ float tmin(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a < b ? a : b); }
__device__ __forceinline__ float tmax(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a > b ? a : b); }

template <int MODE>
__device__ __forceinline__ float iou_one(float4 a, float area1, float4 b) {
  float w = tmin(a.z, b.z) - tmax(a.x, b.x);
  float h = tmin(a.w, b.w) - tmax(a.y, b.y);
  if (w < 0) w = 0;  // clamp_(min=0): NaN stays NaN
  if (h < 0) h = 0;
  float inter = w * h;
  if (MODE == D2AMD_INTERSECTION) return inter;
  float area2 = (b.z - b.x) * (b.w - b.y);
  if (inter > 0) return MODE == D2AMD_IOU ? inter / (area1 + area2 - inter) : inter / area2;
  return 0.f;
}

// Each thread owns VEC consecutive columns (boxes2) and walks ROWS rows (boxes1, staged in LDS,
// broadcast reads).  Stores are 16 B per lane when VEC == 4 -> 1 KiB per wave instruction.
constexpr int IOU_BLOCK = 256;
constexpr int IOU_ROWS = 128;

template <int MODE, int VEC>
__global__ __launch_bounds__(IOU_BLOCK) void pairwise_iou_kernel(
    const float4* __restrict__ b1, int n, const float4* __restrict__ b2, int m, float* __restrict__ out) {
  __shared__ float4 rows[IOU_ROWS];
  __shared__ float areas[IOU_ROWS];
  const int row0 = blockIdx.y * IOU_ROWS;
  const int nrows = min(IOU_ROWS, n - row0);
  for (int i = threadIdx.x; i < nrows; i += IOU_BLOCK) {
    float4 a = b1[row0 + i];
    rows[i] = a;
    areas[i] = (a.z - a.x) * (a.w - a.y);
  }
  __syncthreads();
  const long col0 = ((long)blockIdx.x * IOU_BLOCK + threadIdx.x) * VEC;
  if (col0 >= m) return;
  float4 cb[VEC];
#pragma unroll
  for (int v = 0; v < VEC; v++) cb[v] = (col0 + v < m) ? b2[col0 + v] : make_float4(0, 0, 0, 0);
  for (int i = 0; i < nrows; i++) {
    float4 a = rows[i];
    float ar = areas[i];
    float r[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) r[v] = iou_one<MODE>(a, ar, cb[v]);
    float* o = out + (long)(row0 + i) * m + col0;
    if (VEC == 4) {
      *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
      for (int v = 0; v < VEC; v++) o[v] = r[v];
    }
  }
}

template <int MODE>
static int launch_pairwise(const float* b1, int n, const float* b2, int m, float* out, hipStream_t s) {
  // VEC=4 needs every row start 16-B aligned: m % 4 == 0 (and an aligned base)
  bool vec = (m % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  dim3 block(IOU_BLOCK);
  if (vec) {
    dim3 grid(cdiv(m, IOU_BLOCK * 4), cdiv(n, IOU_ROWS));
    hipLaunchKernelGGL((pairwise_iou_kernel<MODE, 4>), grid, block, 0, s, (const float4*)b1, n,
                       (const float4*)b2, m, out);
  } else {
    dim3 grid(cdiv(m, IOU_BLOCK), cdiv(n, IOU_ROWS));
    hipLaunchKernelGGL((pairwise_iou_kernel<MODE, 1>), grid, block, 0, s, (const float4*)b1, n,
                       (const float4*)b2, m, out);
  }
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

// ---- rotated: one thread per pair; 64-thread blocks own a 64-column strip of ROT_ROWS rows.
constexpr int ROT_BLOCK = 64;
constexpr int ROT_ROWS = 16;

// r04: the polygon clip runs on COMPACTED pairs.  A wave owns 64 columns x 16 rows; evaluated row by row, a row's clip
// ran with whatever lanes survived the exact distance test (rot_pair_is_zero: exactly the pairs whose IoU is +0.f) --
// in RRPN matching > 99 % of the pairs do not, but half of the rows still had a survivor somewhere in the wave.  Now
// every lane tests its column against the 16 rows first (zeros are stored right away), the survivors are listed in LDS
// and the clip runs with lane = one listed pair.
__global__ __launch_bounds__(ROT_BLOCK) void box_iou_rotated_kernel(
    const float* __restrict__ b1, int n, const float* __restrict__ b2, int m, float* __restrict__ out, int rows_per_block) {
  __shared__ RotIouScratch<ROT_BLOCK> S;
  __shared__ float rows[ROT_ROWS][5];
  __shared__ float cols[5][ROT_BLOCK];
  __shared__ uint16_t list[ROT_ROWS * ROT_BLOCK];  // row << 6 | lane
  const int lane = threadIdx.x;
  const int row0 = blockIdx.y * rows_per_block;
  const int nrows = min(rows_per_block, n - row0);
  for (int i = lane; i < nrows * 5; i += ROT_BLOCK) rows[i / 5][i % 5] = b1[(long)row0 * 5 + i];
  const long col0 = (long)blockIdx.x * ROT_BLOCK;
  const long col = col0 + lane;
  const bool in = col < m;
  float cb[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    cb[k] = b2[(in ? col : (long)m - 1) * 5 + k];
    cols[k][lane] = cb[k];
  }
  __syncthreads();
  unsigned live = 0u;  // bit i: (row i, this column) needs the clip
  for (int i = 0; i < nrows; i++) {
    float rb[5];
#pragma unroll
    for (int k = 0; k < 5; k++) rb[k] = rows[i][k];
    if (!in) continue;
    if (rot_pair_is_zero(rb, cb)) out[(long)(row0 + i) * m + col] = 0.f;
    else live |= 1u << i;
  }
  const int mine = __builtin_popcount(live);
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  const int total = __shfl(incl, 63);
  {
    int at = incl - mine;
    unsigned mbits = live;
    while (mbits) {
      const int i = __builtin_ctz(mbits);
      mbits &= mbits - 1;
      list[at++] = (uint16_t)((i << 6) | lane);
    }
  }
  __syncthreads();
  for (int t0 = 0; t0 < total; t0 += ROT_BLOCK) {  // uniform
    const int t = t0 + lane;
    const bool on = t < total;
    const int e = list[on ? t : total - 1];
    const int i = e >> 6, c = e & 63;
    float rb[5], cc[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { rb[k] = rows[i][k]; cc[k] = cols[k][c]; }
    const float v = single_box_iou_rotated<ROT_BLOCK>(rb, cc, S, lane);
    if (on) out[(long)(row0 + i) * m + col0 + c] = v;
  }
}

}  // namespace d2amd

using namespace d2amd;

extern "C" int d2amd_pairwise_iou(const float* boxes1, int n, const float* boxes2, int m, int mode,
                                  float* out, void* stream) {
  D2_CHECK_ARG(n >= 0 && m >= 0, "pairwise_iou: negative size");
  if (n == 0 || m == 0) return D2AMD_OK;
  D2_CHECK_ARG(boxes1 && boxes2 && out, "pairwise_iou: null pointer");
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    case D2AMD_IOU: return launch_pairwise<D2AMD_IOU>(boxes1, n, boxes2, m, out, s);
    case D2AMD_IOA: return launch_pairwise<D2AMD_IOA>(boxes1, n, boxes2, m, out, s);
    case D2AMD_INTERSECTION: return launch_pairwise<D2AMD_INTERSECTION>(boxes1, n, boxes2, m, out, s);
  }
  set_error("pairwise_iou: bad mode %d", mode);
  return D2AMD_EINVAL;
}

extern "C" int d2amd_box_iou_rotated(const float* boxes1, int n, const float* boxes2, int m, float* out,
                                     void* stream) {
  D2_CHECK_ARG(n >= 0 && m >= 0, "box_iou_rotated: negative size");
  if (n == 0 || m == 0) return D2AMD_OK;
  D2_CHECK_ARG(boxes1 && boxes2 && out, "box_iou_rotated: null pointer");
  // the larger set goes on the x (column) axis like the reference's operand swap
  // (box_iou_rotated_cuda.cu:89-100) -- here only grid.y is bounded (65535 * ROT_ROWS rows)
  // rows per 64-column wave: 16, or fewer while the launch would have less than ~16 waves per CU (16 x 268,569: 4)
  int rpb = ROT_ROWS;
  while (rpb > 1 && (long)cdiv(m, ROT_BLOCK) * cdiv(n, rpb) < 4096 * 4) rpb >>= 1;
  { const char* e = d2_prof_env("D2AMD_IOU_ROT_ROWS"); if (e && atoi(e) >= 1 && atoi(e) <= ROT_ROWS) rpb = atoi(e); }  // A/B
  D2_CHECK_ARG(cdiv(n, rpb) <= 65535, "box_iou_rotated: n too large (%d)", n);
  dim3 grid(cdiv(m, ROT_BLOCK), cdiv(n, rpb));
  const bool timed = timing_begin("iou_rotated", (hipStream_t)stream);
  hipLaunchKernelGGL(box_iou_rotated_kernel, grid, dim3(ROT_BLOCK), 0, (hipStream_t)stream, boxes1, n,
                     boxes2, m, out, rpb);
  if (timed) timing_end("iou_rotated", (hipStream_t)stream);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}
