// Shared pieces of the deformable-convolution kernels (deform_conv.hip: generic / fp32 path,
// deform_conv_tc.hip: 16-bit fragment-ordered MFMA path).
#pragma once
#include "common.h"

namespace d2amd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct DcnShape {
  int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, G, DG, Ho, Wo;
  int K2, L, P, Cg, Cog, cpg;
};

static DcnShape make_shape(const d2amd_dcn_params* p) {
  DcnShape s;
  s.B = p->B; s.C = p->C; s.H = p->H; s.W = p->W; s.Co = p->Co; s.kh = p->kh; s.kw = p->kw;
  s.sh = p->stride_h; s.sw = p->stride_w; s.ph = p->pad_h; s.pw = p->pad_w; s.dh = p->dil_h; s.dw = p->dil_w;
  s.G = p->groups; s.DG = p->deformable_groups;
  s.Ho = (s.H + 2 * s.ph - (s.dh * (s.kh - 1) + 1)) / s.sh + 1;
  s.Wo = (s.W + 2 * s.pw - (s.dw * (s.kw - 1) + 1)) / s.sw + 1;
  s.K2 = s.kh * s.kw; s.L = s.Ho * s.Wo; s.P = s.B * s.L;
  s.Cg = s.C / s.G; s.Cog = s.Co / s.G; s.cpg = s.C / s.DG;
  return s;
}

// ---- MFMA wrappers per element type ----------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KSTEP = 16, BK = 64, PAD = 8;
  typedef bf16x8_t frag;
  __device__ static __forceinline__ frag load(const bf16_t* row, int s, int lane) {
    return *reinterpret_cast<const frag*>(row + s * 16 + 8 * (lane >> 5));
  }
  __device__ static __forceinline__ f32x16_t mma(frag a, frag b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  static constexpr int KSTEP = 16, BK = 64, PAD = 8;
  typedef f16x8_t frag;
  __device__ static __forceinline__ frag load(const f16_t* row, int s, int lane) {
    return *reinterpret_cast<const frag*>(row + s * 16 + 8 * (lane >> 5));
  }
  __device__ static __forceinline__ f32x16_t mma(frag a, frag b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KSTEP = 2, BK = 32, PAD = 4;  // rows stay 16-B aligned for vector stores
  typedef float frag;
  __device__ static __forceinline__ frag load(const float* row, int s, int lane) { return row[s * 2 + (lane >> 5)]; }
  __device__ static __forceinline__ f32x16_t mma(frag a, frag b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

// C/D fragment of a 32x32 MFMA: element r of lane l is (row, col) =
// ((r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31)
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- 8 x 16-bit elements as one 16-B register quad -----------------------------------------------
typedef unsigned int raw16 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ __forceinline__ void tc_unpack(const raw16& r, float (&f)[8], bf16_t) {
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
  f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ void tc_unpack(const raw16& r, float (&f)[8], f16_t) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f[2 * i] = to_f32(f16_t{(uint16_t)(w[i] & 0xffffu)});
    f[2 * i + 1] = to_f32(f16_t{(uint16_t)(w[i] >> 16)});
  }
}
__device__ __forceinline__ raw16 tc_pack(const float (&f)[8], bf16_t) {
  raw16 r;
#pragma unroll
  for (int i = 0; i < 4; i++) {  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
    const f32x2_t v = {f[2 * i], f[2 * i + 1]};
    r[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  }
  return r;
}
__device__ __forceinline__ raw16 tc_pack(const float (&f)[8], f16_t) {
  raw16 r;
#pragma unroll
  for (int i = 0; i < 4; i++)
    r[i] = (uint32_t)from_f32<f16_t>(f[2 * i]).v | ((uint32_t)from_f32<f16_t>(f[2 * i + 1]).v << 16);
  return r;
}

// ---- layout kernels ---------------------------------------------------------------------------
// [B][R][S] -> [B][S][R] (32x32 LDS tiles): NCHW <-> NHWC with R = C, S = H*W.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ in, TO* __restrict__ out, int R, int S) {
  __shared__ float tile[32][33];
  const long b = blockIdx.z;
  const int s0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int r = r0 + j, s = s0 + tx;
    if (r < R && s < S) tile[j][tx] = to_f32(in[(b * R + r) * S + s]);
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int s = s0 + j, r = r0 + tx;
    if (r < R && s < S) out[(b * S + s) * R + r] = from_f32<TO>(tile[tx][j]);
  }
}

template <typename TI, typename TO>
static int launch_transpose(const TI* in, TO* out, int B, int R, int S, hipStream_t st) {
  if ((long)B * R * S == 0) return D2AMD_OK;
  dim3 grid(cdiv(S, 32), cdiv(R, 32), B);
  D2_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "deform_conv: tensor too large for transpose grid");
  hipLaunchKernelGGL((transpose_kernel<TI, TO>), grid, dim3(256), 0, st, in, out, R, S);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

// ---- 16-bit fragment-ordered MFMA path (deform_conv_tc.hip) ------------------------------------
struct TcPlan {
  bool ok;
  int MT, NWM, NWN, NKS, wave, ksplit, BM, BN, n_cot, n_pt, NCH, S, ndg;
  size_t wp_bytes, partial_bytes, lds;
};
TcPlan dcn_tc_plan_fwd(const DcnShape& s, int dtype);
template <typename T>
int dcn_tc_forward(const DcnShape& s, const TcPlan& pl, const void* x_nhwc, const void* offset, const void* mask,
                   const void* weight, const void* bias, void* out, void* wp, float* partial, hipStream_t st,
                   bool out_nhwc = false, void* col_out = nullptr);

// weight gradient from the column the forward saved (dcn_bww_gemm.hip): dense split-K GEMM
struct BwwGemmPlan {
  bool ok;
  int Q, n_mt, n_nt, ksplit, kchunk;   // Q = K2 * C / 32 (the column is [P][Q * 32]); 128 x 128 output tiles; K range per workgroup
  size_t col_bytes, partial_bytes;
};
BwwGemmPlan dcn_bww_gemm_plan(const DcnShape& s, int dtype, bool nhwc);
template <typename T>
int dcn_bww_gemm(const DcnShape& s, const BwwGemmPlan& pl, const void* dy_nhwc, const void* col, float* partials,
                 void* grad_weight, hipStream_t st);

struct TcBwPlan {
  bool ok;
  bool gather;  // dX by column gather (no atomics): dcn_tc_backward_data_gather
  int tiles_y, tiles_x, csplit;
  int R, PHt, PWt, csplit_patch;  // LDS-patch variant: displacement margin (-1: all-atomics kernel), patch size
  size_t lds, lds_patch, wp_bytes;
};
TcBwPlan dcn_tc_plan_bwd(const DcnShape& s, int dtype);
template <typename T>
int dcn_tc_backward_data(const DcnShape& s, const TcBwPlan& pl, const void* x_nhwc, const void* offset,
                         const void* mask, const void* weight, const void* gout_nhwc, float* gx, float* goff,
                         float* gmask, void* wp, hipStream_t st);

// workspace of the column-gather backward (deform_conv_tc.hip)
struct DcnGatherWs {
  void* col;    // [P][K2][C] I/O dtype
  int* cnt;     // [B*H*W + 1] entries per input pixel, then the overflow counter
  void* lists;  // [B*H*W][128] {sample id, weight}
  void* ovf;    // [P*K2*4] overflow entries (16 B each): room for every corner of every sample
};
static inline size_t dcn_gather_col_bytes(const DcnShape& s, size_t es) { return (size_t)s.P * s.K2 * s.C * es; }
static inline size_t dcn_gather_cnt_bytes(const DcnShape& s) { return ((size_t)s.B * s.H * s.W + 1) * 4; }
static inline size_t dcn_gather_list_bytes(const DcnShape& s) { return (size_t)s.B * s.H * s.W * 128 * 8; }
static inline size_t dcn_gather_ovf_bytes(const DcnShape& s) { return (size_t)s.P * s.K2 * 4 * 16; }
// Optional second stream of one backward call (deform_conv.hip: dcn_side()).  The sample binning -- needed by the column
// gather only -- and whatever `work` enqueues (the weight-gradient GEMM: it reads dY and the saved column, nothing of the
// data path) run there beside the data-gradient kernel, which is a latency chain that leaves the chip half idle.
struct DcnSide {
  hipStream_t stream;
  hipEvent_t fork, bin, join;
  int (*work)(void* ctx, hipStream_t side);  // enqueued on `stream` behind the binning (may be null)
  void* ctx;
};
template <typename T>
int dcn_tc_backward_data_gather(const DcnShape& s, const TcBwPlan& pl, const void* x_nhwc, const void* offset,
                                const void* mask, const void* weight, const void* gout_nhwc, void* gx_t, float* goff,
                                float* gmask, void* wp, const DcnGatherWs& gw, hipStream_t st,
                                const DcnSide* side = nullptr, const struct ColPathPlan* cp = nullptr, void* goff_t = nullptr,
                                void* gmask_t = nullptr, const void* wt_kept = nullptr);

struct TcBwwPlan {
  bool ok;
  int n_cot, n_cit, pch, ksteps_per_image;
  int share, pchw;       // cooperative kernel: output-channel tiles per workgroup (0: the wave-autonomous kernel), chunks
  size_t partial_bytes;  // the partial tiles the kernel writes: one 64 x 64 fp32 slot per workgroup
};
TcBwwPlan dcn_tc_plan_bww(const DcnShape& s, int dtype);
template <typename T>
int dcn_tc_backward_weight(const DcnShape& s, const TcBwwPlan& pl, const void* x_nhwc, const void* offset,
                           const void* mask, const void* gout_nchw, float* partials, void* grad_weight, hipStream_t st);

}  // namespace d2amd
