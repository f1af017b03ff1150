// Dense 16-bit "NT" GEMM on the matrix cores -- the matrix work of the deformable convolution once its column exists:
//
//   out[m][n] = sum_k X[m][k] * Wn[n][k]  (+ bias[n])          X: [M][ldx], Wn: [N][ldw], both K-contiguous; out: [M][ldo]
//
//   forward        Y[p][co]      = col[p][(tap, ci)] . Wp[co][(tap, ci)]        M = P, N = Co,  K = 9C
//   backward-data  dcol[p][(tap, ci)] = dY[p][co] . Wt[(tap, ci)][co]           M = P, N = 9C,  K = Co
//
// (the reference does both with at::addmm_ per image on its column buffer: deform_conv_cuda.cu:393-404,455-558,
// 958-1004,1099-1140).  M is the long dimension (33,600 / 8,400 / 2,100 positions for R50 res3 / res4 / res5 at 2
// images), N and K are 128..4,608: few tiles per CU, so what matters is the latency of ONE tile and the balance over
// 256 CUs, not steady-state MFMA issue.
//
// Workgroup = 256 threads = 2 x 2 waves on a BM x BN tile (128 x 128 or 64 x 128), K step 64.  Both operand tiles are
// staged by LDS-DMA (global_load_lds, 16 B per lane: no staging registers, no ds_write pass) into [row][64 k] images
// whose 16-B slots are XOR-swizzled by (row >> 1) & 7 -- the DMA writes LDS linearly, so the swizzle is applied to the
// per-lane SOURCE address and again on the fragment read -- which makes the ds_read_b128 of a 32-row fragment
// conflict-free.  NST stages: loads run NST - 1 K steps ahead of the MFMAs (counted vmcnt, raw s_barrier).
// The MFMA A operand (rows i) comes from Wn, the B operand (columns j) from X: a lane then holds 4 CONSECUTIVE n of one
// m per register group, the epilogue packs them to 8 B, transposes the tile through LDS and writes full 16-B chunks
// of rows of `out`.  No split-K: a split measured +6..15 us on these shapes (partial tiles through HBM + an agent-scope
// release per workgroup, profiles/r05/LOG.md); short-M problems take 64 x 64 tiles instead.
#pragma once
#include "dcn_common.h"

namespace d2amd {

struct GemmNtArgs {
  const void* X;      // [M][ldx]
  const void* Wn;     // [N][ldw]
  void* out;          // [M][ldo]
  const void* bias;   // [N] or null
  int M, N, K, ldx, ldw, ldo;
  int n_mt, n_nt, total;
};

typedef __attribute__((ext_vector_type(4))) unsigned int gm_u32x4;
typedef __attribute__((ext_vector_type(4))) float gm_f32x4;

template <typename T>
__device__ __forceinline__ f32x16_t gm_mma(gm_u32x4 a, gm_u32x4 b, f32x16_t c) {
  typedef typename Mma<T>::frag F;
  return Mma<T>::mma(__builtin_bit_cast(F, a), __builtin_bit_cast(F, b), c);
}

template <typename T>
__device__ __forceinline__ uint32_t gm_pack2(float a, float b) {
  return (uint32_t)from_f32<T>(a).v | ((uint32_t)from_f32<T>(b).v << 16);
}

template <int N>
__device__ __forceinline__ void gm_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

static inline size_t gemm_nt_lds_bytes(int BM, int BN, int BK, int NST) {
  const size_t stages = (size_t)NST * (BM + BN) * BK * 2;
  const size_t epi = (size_t)BM * (BN * 2 + 16);
  return stages > epi ? stages : epi;
}

// WM x WN waves on the tile (4 or 8 waves: with two waves per SIMD one wave's MFMAs cover the other's LDS latency)
template <typename T, int BM, int BN, int BK, int NST, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (NST * (BM + BN) * BK * 2 <= 80 * 1024) ? 2 : 1) void gemm_nt_kernel(GemmNtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char gm_smem[];
  static_assert(BK == 32 || BK == 64, "K step");
  constexpr int NW = WM * WN, THREADS = 64 * NW;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);  // 32 x 32 MFMA blocks per wave along m / n
  static_assert(TM >= 1 && TN >= 1 && (BM / (64 / (BK / 8))) % NW == 0, "tile / wave layout");
  constexpr int ROWS = BM + BN;                     // rows of one stage: X rows, then Wn rows
  constexpr int RB = BK * 2;                        // bytes of one LDS row
  constexpr int SLOTS = BK / 8;                     // 16-B slots per row
  constexpr int RPI = 64 / SLOTS;                   // rows one DMA instruction covers
  constexpr int STAGE = ROWS * RB;
  constexpr int LPW = ROWS / RPI / NW;              // DMA instructions per wave and stage
  constexpr int KK = BK / 16;                       // MFMA k steps per stage
  // swizzle of a row's slots: the 16 lanes one ds_read_b128 cycle serves (16 consecutive rows, one slot) must fall on 16
  // distinct 16-B columns of the 256-B bank row
  auto swz = [](int row) __attribute__((always_inline)) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;

  // XCD-aware decode: an XCD takes a contiguous range of logical ids; the n tiles of an m tile are adjacent (the same
  // X rows are fetched into one L2)
  const int per_xcd = (a.total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= a.total) return;
  const int nt = logical % a.n_nt, mt = logical / a.n_nt;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = a.K / BK;

  // ---- DMA source addresses (per lane, per instruction), K offset added per step
  const char* src[LPW];
#pragma unroll
  for (int i = 0; i < LPW; i++) {
    const int q = wave + NW * i;                      // RPI-row block of the stage
    const bool isx = i < BM / RPI / NW;               // (BM / RPI is a multiple of NW: compile-time per i)
    const int r = q * RPI + lane / SLOTS - (isx ? 0 : BM);
    const int slot = (lane % SLOTS) ^ swz(r);
    const int grow = isx ? min(m0 + r, a.M - 1) : min(n0 + r, a.N - 1);
    const char* base = isx ? (const char*)a.X : (const char*)a.Wn;
    const int ld = isx ? a.ldx : a.ldw;
    src[i] = base + (size_t)grow * ld * 2 + slot * 16;
  }
  // DMA instructions [i0, i1) of a stage (a wave's instructions go through the CU's address unit at 16 cycles each, and
  // an in-order wave cannot issue anything behind a vector-memory instruction the unit has not accepted: all LPW at
  // once stalled the wave ~500 cycles before its first fragment read -- measured, scripts/probes/cu_probe.hip: k step
  // 768 cycles, DMA 593, both 1,285 -- so the K loop issues them in pieces between its MFMA groups)
  auto issue = [&](int stage, int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < LPW; i++) {
      if (i < i0 || i >= i1) continue;
      char* dst = gm_smem + stage * STAGE + (wave + NW * i) * 1024;  // wave-uniform; the DMA adds lane * 16 (= RPI rows)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      src[i] += BK * 2;
    }
  };

  // ---- fragment read offsets: row (lane & 31) of a 32-row block, slot (2 kk + (lane >> 5)) ^ swizzle
  const int sw = swz(lane & 31);
  int foff[KK];
#pragma unroll
  for (int kk = 0; kk < KK; kk++) foff[kk] = (lane & 31) * RB + (((2 * kk + (lane >> 5)) ^ sw) << 4);
  const int xrow0 = wm * (BM / WM) * RB;                  // this wave's X rows within the stage
  const int wrow0 = (BM + wn * (BN / WN)) * RB;           // this wave's Wn rows

  f32x16_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; i++)
#pragma unroll
    for (int j = 0; j < TM; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // ---- pipeline: NST - 1 stages in flight
#pragma unroll
  for (int s = 0; s < NST - 1; s++)
    if (s < nk) issue(s, 0, LPW);
  for (int kt = 0; kt < nk; kt++) {
    // stage kt has landed when at most the loads of the later stages issued so far are outstanding
    if constexpr (NST == 2) {
      gm_wait_vm<0>();
    } else {
      if (kt + NST - 2 < nk) gm_wait_vm<(NST - 2) * LPW>(); else gm_wait_vm<0>();   // (the tail drains everything)
    }
    __builtin_amdgcn_s_barrier();  // everybody's DMA of stage kt visible; everybody done reading stage kt - 1
    const bool more = kt + NST - 1 < nk;  // uniform
    const int nxt = (kt + NST - 1) % NST;
    const char* st = gm_smem + (kt % NST) * STAGE;
#pragma unroll
    for (int kk = 0; kk < KK; kk++) {
      gm_u32x4 wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; i++) wf[i] = *reinterpret_cast<const gm_u32x4*>(st + wrow0 + i * 32 * RB + foff[kk]);
#pragma unroll
      for (int j = 0; j < TM; j++) xf[j] = *reinterpret_cast<const gm_u32x4*>(st + xrow0 + j * 32 * RB + foff[kk]);
      if (more) issue(nxt, kk * LPW / KK, (kk + 1) * LPW / KK);  // this k step's share of stage kt + NST - 1
#pragma unroll
      for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++) acc[i][j] = gm_mma<T>(wf[i], xf[j], acc[i][j]);
    }
  }

  // ---- epilogue: (+ bias) -> 16-bit, transposed through LDS, rows of `out` written in 16-B chunks
  constexpr int OP = BN * 2 + 16;  // LDS pitch of an output row
  __syncthreads();                 // every wave is done with the stages
#pragma unroll
  for (int i = 0; i < TN; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int n = wn * (BN / WN) + i * 32 + 8 * q + 4 * (lane >> 5);  // first of this lane's 4 consecutive columns
      float b4[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
        const T* bp = (const T*)a.bias + min(n0 + n, a.N - 4);
#pragma unroll
        for (int e = 0; e < 4; e++) b4[e] = to_f32(bp[e]);
      }
#pragma unroll
      for (int j = 0; j < TM; j++) {
        const int m = wm * (BM / WM) + j * 32 + (lane & 31);
        uint2 w;
        w.x = gm_pack2<T>(acc[i][j][4 * q] + b4[0], acc[i][j][4 * q + 1] + b4[1]);
        w.y = gm_pack2<T>(acc[i][j][4 * q + 2] + b4[2], acc[i][j][4 * q + 3] + b4[3]);
        *reinterpret_cast<uint2*>(gm_smem + m * OP + n * 2) = w;
      }
    }
  __syncthreads();
  constexpr int CPR = BN / 8;  // 16-B chunks per row
#pragma unroll
  for (int i = 0; i < BM * CPR / THREADS; i++) {
    const int c = tid + THREADS * i;
    const int row = c / CPR, ch = c % CPR;
    if (m0 + row < a.M && n0 + ch * 8 < a.N) {
      const gm_u32x4 v = *reinterpret_cast<const gm_u32x4*>(gm_smem + row * OP + ch * 16);
      *reinterpret_cast<gm_u32x4*>((char*)a.out + ((size_t)(m0 + row) * a.ldo + n0 + ch * 8) * 2) = v;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct GemmNtPlan {
  bool ok;
  int BM, BN, BK, NST, WM, WN, n_mt, n_nt;
  size_t lds;
};

// tile shape: the largest of 128 x 128, 64 x 128, 64 x 64 that still gives ~2 workgroups per CU (profiles/r05/LOG.md)
GemmNtPlan gemm_nt_plan(int M, int N, int K);
template <typename T>
int gemm_nt_launch(const GemmNtPlan& pl, GemmNtArgs a, hipStream_t st);

// ---- the column path of the deformable convolution (dcn_colpath.hip) --------------------------------------------------
struct ColPathPlan {
  bool ok;
  GemmNtPlan fwd, bwd;  // Y = col Wp^T;  dcol = dY Wt^T
  int NP;               // positions per workgroup of the column / coordinate-gradient kernels
  size_t col_bytes, wpack_bytes;
};
ColPathPlan dcn_colpath_plan(const DcnShape& s, int dtype);
// col: [P][K2 * C] (kept by the caller in training, else scratch); wpack: wpack_bytes of scratch; wt_keep: null, or
// wpack_bytes behind the kept column where the backward's weight operand is left
template <typename T>
int dcn_colpath_forward(const DcnShape& s, const ColPathPlan& pl, const void* x_nhwc, const void* offset, const void* mask,
                        const void* weight, const void* bias, void* out_nhwc, void* col, void* wpack, void* wt_keep,
                        hipStream_t st);
// dcol: [P][K2 * C] out; wt_kept: what the forward left (null: packed here into wpack); goff / gmask (I/O dtype, the
// caller's tensors) may be null
template <typename T>
int dcn_colpath_backward_data(const DcnShape& s, const ColPathPlan& pl, const void* x_nhwc, const void* offset, const void* mask,
                              const void* weight, const void* gout_nhwc, void* dcol, void* wpack, const void* wt_kept,
                              void* goff, void* gmask, hipStream_t st);

}  // namespace d2amd
