// Deformable convolution, weight gradient from the column the FORWARD saved: a dense split-K GEMM on MFMA.
//
//   dW[co][(tap, ci)] = sum_p dY[p][co] * col[p][(tap, ci)]          (deform_conv_cuda.cu:1160-1196: the reference
//                                                                      re-runs im2col and calls at::addmm_ per image)
//
// Why: the weight gradient used to RE-GATHER the column (4 corners x 9 taps x C channels per position, ~25 VALU per
// sampled element: 80-105 us per R50 block, profiles/r03) although the forward had just produced exactly that matrix.
// MI355X has 288 GB of HBM: the training forward now keeps the column (16-bit, mask folded in; 77 MB for a res3 block
// of 2 images, 0.6 GB for all 13 blocks) -- the reference's own `columns` scratch tensor (layers/deform_conv.py:97-98)
// promoted from scratch to saved activation -- and the weight gradient is pure matrix work: 9.9 GFLOP per block,
// HBM-bound at ~86 MB (the column once + dY).
//
// Layouts.  col: [position p][tap * C + ci] (what dcn_col_kernel / dcn_fwd_tc_kernel write; r04 kept it in 32-channel
// chunks [q][p][32], r05 made it the row-major operand of the forward GEMM).  dY: NHWC [p][Co].  BOTH operands have the reduction
// index (p) as their ROW index, i.e. they are K-major where v_mfma_f32_32x32x16 wants 8 consecutive k per lane: the
// tiles are staged row-major in LDS by straight 16-B copies and the fragments are read with ds_read_b64_tr_b16, the
// gfx950 4x4 transpose read (lane mapping: scripts/probes/probe_tr16.hip, used the same way by pool_bwd_mfma_kernel).
// LDS rows are 320 B apart: the 16 rows x 64 B one wave-wide tr read touches fall on 4 x 4 distinct bank quarters.
//
// Workgroup = 128 output channels x 128 columns (4 chunks) x one K range; 4 waves of 64 x 64 (2 x 2 MFMA tiles, 64
// accumulator VGPRs); K step 32 positions, LDS double-buffered, next step's global loads in flight under the MFMAs.
// Split-K partial tiles are written with plain stores and summed IN SPLIT ORDER by bww_gemm_reduce_kernel (no atomics:
// dW is deterministic), which also converts to the caller's [Co][C][kh][kw] layout.
#include "dcn_gemm.h"

namespace d2amd {

typedef unsigned int g_raw16 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) short g_s16x4;
typedef __attribute__((ext_vector_type(8))) short g_s16x8;

constexpr int GM = 128, GN = 128, GK = 32;
constexpr int G_PITCH = 320;  // bytes per k row of a staged tile (256 B of data)

struct BwwGemmArgs {
  const void* dy;   // [P][Co]
  const void* col;  // [P][Q * 32]
  float* part;      // [ksplit][n_mt][n_nt][128][128]
  int P, Co, Q, n_mt, n_nt, ksplit, kchunk, total;
};

template <typename T>
__device__ __forceinline__ f32x16_t g_mma(g_s16x8 a, g_s16x8 b, f32x16_t c) {
  typedef typename Mma<T>::frag F;
  return Mma<T>::mma(__builtin_bit_cast(F, a), __builtin_bit_cast(F, b), c);
}

// fragment of a [32 k][.. columns] LDS image for v_mfma_f32_32x32x16: the lane receives column (lane & 31), k = 8 (lane >> 5)
// .. + 8 of the k-step whose first row is `rows`; col0 = first column (16-bit elements) of the 32-column tile
__device__ __forceinline__ g_s16x8 g_frag(const char* img, int col0, int lane) {
  const int ii = lane & 15, grp = (lane >> 4) & 1, kh = lane >> 5;
  const char* p = img + (8 * kh + (ii >> 2)) * G_PITCH + (col0 + grp * 16 + (ii & 3) * 4) * 2;
  const g_s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g_s16x4 __attribute__((address_space(3)))*)p);
  const g_s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g_s16x4 __attribute__((address_space(3)))*)(p + 4 * G_PITCH));
  return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void dcn_bww_gemm_kernel(BwwGemmArgs a) {
  __shared__ __attribute__((aligned(16))) char As[2][GK * G_PITCH];
  __shared__ __attribute__((aligned(16))) char Bs[2][GK * G_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  // XCD-aware decode: each XCD takes a contiguous range of logical ids; ids of one K range (and row tile) are adjacent,
  // so the dY rows a K range reads are fetched into ONE L2
  const int per_xcd = (a.total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= a.total) return;
  const int nt = logical % a.n_nt;
  const int mt = (logical / a.n_nt) % a.n_mt;
  const int ks = logical / (a.n_nt * a.n_mt);
  const int k0 = ks * a.kchunk;
  const int k1 = min(a.P, k0 + a.kchunk);
  const int nsteps = (k1 - k0 + GK - 1) / GK;  // >= 1 by construction of the plan

  // ---- global -> register staging: 2 x 16 B per thread and operand
  const char* dy = (const char*)a.dy;
  const char* col = (const char*)a.col;
  const int a_row[2] = {tid >> 4, (tid >> 4) + 16};            // 16 threads x 16 B = one 256-B row of 128 channels
  const int a_c16 = tid & 15;
  const bool a_ok = mt * GM + a_c16 * 8 < a.Co;                // (Co % 8 == 0: a 16-B group is in or out as a whole)
  // (B like A: 16 threads x 16 B = one 256-B piece of a column row, 128 of its (tap, ci) entries)
  const int q0 = nt * 4;
  g_raw16 ra[2], rb[2];
  auto issue = [&](int step) __attribute__((always_inline)) {
    const int kb = k0 + step * GK;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int p = kb + a_row[j];
      g_raw16 z = {0u, 0u, 0u, 0u};
      ra[j] = (a_ok && p < k1) ? *reinterpret_cast<const g_raw16*>(dy + ((size_t)p * a.Co + mt * GM + a_c16 * 8) * sizeof(T)) : z;
      const int q = q0 + (a_c16 >> 2);
      rb[j] = (q < a.Q && p < k1) ? *reinterpret_cast<const g_raw16*>(col + (((size_t)p * a.Q + q0) * 32 + a_c16 * 8) * sizeof(T)) : z;
    }
  };
  auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      *reinterpret_cast<g_raw16*>(&As[buf][a_row[j] * G_PITCH + a_c16 * 16]) = ra[j];
      *reinterpret_cast<g_raw16*>(&Bs[buf][a_row[j] * G_PITCH + a_c16 * 16]) = rb[j];
    }
  };
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  issue(0);
  stage(0);
  __syncthreads();
  for (int st = 0; st < nsteps; st++) {
    const int cur = st & 1;
    const bool more = st + 1 < nsteps;  // uniform
    if (more) issue(st + 1);
#pragma unroll
    for (int k16 = 0; k16 < 2; k16++) {
      const char* ai = &As[cur][16 * k16 * G_PITCH];
      const char* bi = &Bs[cur][16 * k16 * G_PITCH];
      g_s16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) af[i] = g_frag(ai, 64 * wm + 32 * i, lane);
#pragma unroll
      for (int j = 0; j < 2; j++) bf[j] = g_frag(bi, 64 * wn + 32 * j, lane);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = g_mma<T>(af[i], bf[j], acc[i][j]);
    }
    if (more) stage(cur ^ 1);
    __syncthreads();
  }

  // ---- partial tile [ks][mt][nt][128][128] fp32, plain stores (row = output channel, column = (tap, ci) index)
  float* out = a.part + ((size_t)(ks * a.n_mt + mt) * a.n_nt + nt) * (GM * GN);
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = 64 * wm + 32 * i + frag_row(r, lane);
        const int c = 64 * wn + 32 * j + (lane & 31);
        out[row * GN + c] = acc[i][j][r];
      }
}

// dW[co][ci][tap] = sum over the K splits of part[ks][mt][nt][co % 128][n % 128], n = tap * C + ci.  Workgroup = 64 groups
// of 4 consecutive n (16-B loads) x 4 interleaved shares of the splits; a share adds its splits in ascending order, the
// four shares are added in a fixed order through LDS: deterministic, and the chain of dependent loads is a quarter as
// long (the r04 kernel -- a thread per element walking all splits with 4-B loads -- took 14-24 us for 35 MB).
template <typename T>
__global__ __launch_bounds__(256) void bww_gemm_reduce_kernel(const float* __restrict__ part, T* __restrict__ gw, int Co, int C,
                                                             int K2, int n_mt, int n_nt, int ksplit) {
  __shared__ float4 sh[3][64];
  const int N = K2 * C;                       // (a multiple of 64: C % 64 == 0)
  const int quads = N >> 2;
  const int q = threadIdx.x & 63, share = threadIdx.x >> 6;
  const long item = (long)blockIdx.x * 64 + q;  // (co, quad of n)
  const long items = (long)Co * quads;
  const bool ok = item < items;
  const int co = ok ? (int)(item / quads) : 0;
  const int n = ok ? (int)(item - (long)co * quads) * 4 : 0;
  const size_t tile = (size_t)GM * GN, split = (size_t)n_mt * n_nt * tile;
  const float* p = part + ((size_t)(co / GM) * n_nt + (size_t)(n / GN)) * tile + (size_t)(co % GM) * GN + (size_t)(n % GN);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    int k = share;
    for (; k + 12 < ksplit; k += 16) {  // four loads in flight
      const float4 a0 = *reinterpret_cast<const float4*>(p + (size_t)k * split);
      const float4 a1 = *reinterpret_cast<const float4*>(p + (size_t)(k + 4) * split);
      const float4 a2 = *reinterpret_cast<const float4*>(p + (size_t)(k + 8) * split);
      const float4 a3 = *reinterpret_cast<const float4*>(p + (size_t)(k + 12) * split);
      acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w;
      acc.x += a1.x; acc.y += a1.y; acc.z += a1.z; acc.w += a1.w;
      acc.x += a2.x; acc.y += a2.y; acc.z += a2.z; acc.w += a2.w;
      acc.x += a3.x; acc.y += a3.y; acc.z += a3.z; acc.w += a3.w;
    }
    for (; k < ksplit; k += 4) {
      const float4 a0 = *reinterpret_cast<const float4*>(p + (size_t)k * split);
      acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w;
    }
  }
  if (share > 0) sh[share - 1][q] = acc;
  __syncthreads();
  if (share == 0 && ok) {
#pragma unroll
    for (int s2 = 0; s2 < 3; s2++) { const float4 o = sh[s2][q]; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    const int tap = n / C, ci = n - tap * C;  // (4 consecutive n share the tap: C % 4 == 0)
    T* dst = gw + ((long)co * C + ci) * K2 + tap;
    dst[0] = from_f32<T>(acc.x); dst[K2] = from_f32<T>(acc.y); dst[2 * K2] = from_f32<T>(acc.z); dst[3 * K2] = from_f32<T>(acc.w);
  }
}

BwwGemmPlan dcn_bww_gemm_plan(const DcnShape& s, int dtype, bool nhwc) {
  BwwGemmPlan pl{};
  pl.ok = false;
  if (getenv("D2AMD_DCN_NO_SAVED_COL")) return pl;  // A/B switch: the re-gathering weight-gradient kernels
  if (dtype != D2AMD_BF16 && dtype != D2AMD_F16) return pl;
  if (s.G != 1 || s.DG != 1 || s.C % 64 != 0 || s.Co % 8 != 0 || s.P <= 0) return pl;
  if (!(nhwc && dcn_colpath_plan(s, dtype).ok)) {  // who writes the column: dcn_col_kernel, or the fused forward kernel
    const TcPlan f = dcn_tc_plan_fwd(s, dtype);
    if (!f.ok || f.NKS != 2 || f.wave) return pl;
  }
  pl.Q = s.K2 * (s.C / 32);
  pl.n_mt = cdiv(s.Co, GM);
  pl.n_nt = cdiv(pl.Q, 4);
  const long tiles = (long)pl.n_mt * pl.n_nt;
  int ksplit = (int)((512 + tiles - 1) / tiles);  // ~2 workgroups per CU
  { const char* e = d2_prof_env("D2AMD_DCN_BWW_KSPLIT"); if (e && atoi(e) > 0) ksplit = atoi(e); }
  int kchunk = cdiv(cdiv(s.P, ksplit), GK) * GK;
  if (kchunk < 8 * GK) kchunk = 8 * GK;  // at least 8 K steps per workgroup
  pl.kchunk = kchunk;
  pl.ksplit = cdiv(s.P, kchunk);
  pl.col_bytes = (size_t)pl.Q * s.P * 32 * 2;
  pl.partial_bytes = (size_t)pl.ksplit * tiles * GM * GN * 4;
  pl.ok = true;
  return pl;
}

template <typename T>
int dcn_bww_gemm(const DcnShape& s, const BwwGemmPlan& pl, const void* dy_nhwc, const void* col, float* partials,
                 void* grad_weight, hipStream_t st) {
  BwwGemmArgs a{};
  a.dy = dy_nhwc; a.col = col; a.part = partials;
  a.P = s.P; a.Co = s.Co; a.Q = pl.Q; a.n_mt = pl.n_mt; a.n_nt = pl.n_nt; a.ksplit = pl.ksplit; a.kchunk = pl.kchunk;
  const long total = (long)pl.n_mt * pl.n_nt * pl.ksplit;
  D2_CHECK_ARG(total < (1l << 30), "deform_conv: too many weight-gradient tiles");
  a.total = (int)total;
  const int grid = (a.total + 7) / 8 * 8;
  const bool timed = timing_begin("dcn_bwd_weight", st);
  hipLaunchKernelGGL((dcn_bww_gemm_kernel<T>), dim3(grid), dim3(256), 0, st, a);
  if (timed) timing_end("dcn_bwd_weight", st);
  D2_LAUNCH_OK();
  const long items = (long)s.Co * s.C * s.K2 / 4;
  hipLaunchKernelGGL((bww_gemm_reduce_kernel<T>), dim3(cdiv(items, 64)), dim3(256), 0, st, (const float*)partials, (T*)grad_weight,
                     s.Co, s.C, s.K2, pl.n_mt, pl.n_nt, pl.ksplit);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

template int dcn_bww_gemm<bf16_t>(const DcnShape&, const BwwGemmPlan&, const void*, const void*, float*, void*, hipStream_t);
template int dcn_bww_gemm<f16_t>(const DcnShape&, const BwwGemmPlan&, const void*, const void*, float*, void*, hipStream_t);

}  // namespace d2amd
