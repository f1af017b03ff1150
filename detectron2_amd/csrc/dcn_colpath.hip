// Deformable convolution v1 / v2, 16-bit channels_last path built AROUND a dense GEMM (round 5).
//
// The reference's own structure -- im2col to a column buffer, at::addmm_, col2im (deform_conv_cuda.cu:272-440,826-1221)
// -- is the right one for MI355X once the column is (a) 16-bit, (b) written and read exactly once per pass by streaming
// kernels and (c) contracted by a GEMM that runs at the vendor library's rate (dcn_gemm.h): the fused gather-MFMA
// kernels of rounds 1-4 (deform_conv_tc.hip) were latency chains at 0.04-0.07 of the MFMA peak (59 us forward, 80-134 us
// backward-data per R50 block), and the training forward stored the column anyway.
//
//   forward        col[p][(tap, ci)] = mask * bilinear(x)        dcn_col_kernel        (HBM writer: 77 / 39 / 19 MB)
//                  Y[p][co] = col . Wp^T (+ bias)                gemm_nt               (dcn_gemm.h)
//   backward-data  dcol[p][(tap, ci)] = dY[p][co] . Wt^T         gemm_nt
//                  d offset, d mask from dcol and x's corners    dcn_coord_grad_kernel (HBM reader)
//                  dX = per-pixel gather of dcol rows            dcn_bin_samples_kernel + dcn_gather_dx_kernel (r03/r04)
//   backward-weight dW = dY^T . col                              dcn_bww_gemm_kernel   (r04, the column the forward kept)
//
// Numerics are those of the reference's 16-bit path: the column and dcol are rounded to the I/O dtype once
// (deform_conv_cuda.cu keeps `columns` in the input dtype), every contraction accumulates in fp32.
#include "dcn_gemm.h"

namespace d2amd {

// ---- weights: W[co][ci][tap] -> Wp[co][tap * C + ci]  (forward: N = Co rows, K contiguous)
//                               -> Wt[tap * C + ci][co]  (backward-data: N = 9C rows, K = Co contiguous)
// Workgroup = a 32 co x 32 ci tile of ONE tap through LDS (C / 32 x Co / 32 x K2 workgroups): 64-B runs on both output
// sides.  The training forward packs BOTH (Wt rides in the tail of the column buffer the caller keeps for the backward):
// one pack launch per block and iteration.
template <typename T>
__device__ __forceinline__ void dcn_pack_body(const T* __restrict__ w, T* __restrict__ wp, T* __restrict__ wt, int Co, int C,
                                              int K2, int ci0, int co0, int tap, T (&tile)[32][33]) {
  const int a = threadIdx.x & 31, r = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int j = 0; j < 4; j++) tile[r + 8 * j][a] = w[((long)(co0 + r + 8 * j) * C + ci0 + a) * K2 + tap];
  __syncthreads();
  if (wp) {
#pragma unroll
    for (int j = 0; j < 4; j++) wp[((long)(co0 + r + 8 * j) * K2 + tap) * C + ci0 + a] = tile[r + 8 * j][a];
  }
  if (wt) {
#pragma unroll
    for (int j = 0; j < 4; j++) wt[((long)tap * C + ci0 + r + 8 * j) * Co + co0 + a] = tile[a][r + 8 * j];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void dcn_pack_weights_kernel(const T* __restrict__ w, T* __restrict__ wp, T* __restrict__ wt,
                                                              int Co, int C, int K2) {
  __shared__ T tile[32][33];  // [co][ci]
  dcn_pack_body<T>(w, wp, wt, Co, C, K2, blockIdx.x * 32, blockIdx.y * 32, blockIdx.z, tile);
}

// ---- sample tables ----------------------------------------------------------------------------------------------------
// (position, tap) -> the four corner pixels and what the kernels need of the bilinear weights, as
// deform_conv_cuda_kernel.cu:96-130,216-270 (v1) / 665-700,785-860 (v2): a sample inside (-1, H) x (-1, W), corners
// outside the image contribute 0.
struct __attribute__((aligned(16))) CpEntry {
  uint32_t pix[4];   // pixel index (b * H + y) * W + x of the corner; 0 when unused
  float w[4];        // bilinear weight (forward table: x modulation mask); 0 for corners / samples outside
  float lh, lw, m;   // (backward table) fractional parts, modulation mask
  uint32_t flags;    // bit c: corner c inside; bit 4: sample inside
  int oidx, midx;    // element index of d offset_h (d offset_w: + L) and of d mask for this sample; -1 beyond the last position
  int pad[2];
};
static_assert(sizeof(CpEntry) == 64, "CpEntry layout");

template <typename T, bool FOLD_MASK>
__device__ __forceinline__ CpEntry cp_make_entry(const DcnShape& s, const T* __restrict__ offset, const T* __restrict__ mask,
                                                 long p, int tap) {
  CpEntry e;
#pragma unroll
  for (int t = 0; t < 4; t++) { e.pix[t] = 0u; e.w[t] = 0.f; }
  e.lh = e.lw = 0.f; e.m = 0.f; e.flags = 0u;
  e.oidx = e.midx = -1; e.pad[0] = e.pad[1] = 0;
  if (p >= s.P) return e;
  const int b = (int)(p / s.L), l = (int)(p - (long)b * s.L);
  e.oidx = (int)(((long)b * 2 * s.K2 + 2 * tap) * s.L + l);
  e.midx = (int)(((long)b * s.K2 + tap) * s.L + l);
  const int ho = l / s.Wo, wo = l - ho * s.Wo;
  const int i = tap / s.kw, j = tap - i * s.kw;
  const long obase = (long)b * 2 * s.K2;
  const float off_h = to_f32(offset[(obase + 2 * tap) * s.L + l]);
  const float off_w = to_f32(offset[(obase + 2 * tap + 1) * s.L + l]);
  const float m = mask ? to_f32(mask[((long)b * s.K2 + tap) * s.L + l]) : 1.f;
  e.m = m;
  const float h_im = (float)(ho * s.sh - s.ph + i * s.dh) + off_h;
  const float w_im = (float)(wo * s.sw - s.pw + j * s.dw) + off_w;
  if (!(h_im > -1.f && w_im > -1.f && h_im < (float)s.H && w_im < (float)s.W)) return e;
  const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
  const float hh = 1.f - lh, hw = 1.f - lw;
  e.lh = lh; e.lw = lw; e.flags = 16u;
  const float f = FOLD_MASK ? m : 1.f;
  const long rowbase = (long)b * s.H;
  if (h_low >= 0 && w_low >= 0) { e.pix[0] = (uint32_t)((rowbase + h_low) * s.W + w_low); e.w[0] = hh * hw * f; e.flags |= 1u; }
  if (h_low >= 0 && w_high <= s.W - 1) { e.pix[1] = (uint32_t)((rowbase + h_low) * s.W + w_high); e.w[1] = hh * lw * f; e.flags |= 2u; }
  if (h_high <= s.H - 1 && w_low >= 0) { e.pix[2] = (uint32_t)((rowbase + h_high) * s.W + w_low); e.w[2] = lh * hw * f; e.flags |= 4u; }
  if (h_high <= s.H - 1 && w_high <= s.W - 1) { e.pix[3] = (uint32_t)((rowbase + h_high) * s.W + w_high); e.w[3] = lh * lw * f; e.flags |= 8u; }
  return e;
}

constexpr int CP_MAXS = 32 * 9;  // samples (positions x taps) of one workgroup's table

// sum over 8 channels of a * b, both 8 x 16-bit: four v_dot2_f32_{bf16,f16} (exact products, fp32 accumulation)
typedef __attribute__((ext_vector_type(2))) __bf16 cp_bf2;
typedef __attribute__((ext_vector_type(2))) _Float16 cp_h2;
// (components taken apart by name: with a[i] in an unrolled loop hipcc 7.2 fed the FIRST dword to all four instructions)
__device__ __forceinline__ float cp_dot8(const raw16& a, const raw16& b, bf16_t) {
  float acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(cp_bf2, (uint32_t)a.x), __builtin_bit_cast(cp_bf2, (uint32_t)b.x), 0.f, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(cp_bf2, (uint32_t)a.y), __builtin_bit_cast(cp_bf2, (uint32_t)b.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(cp_bf2, (uint32_t)a.z), __builtin_bit_cast(cp_bf2, (uint32_t)b.z), acc, false);
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(cp_bf2, (uint32_t)a.w), __builtin_bit_cast(cp_bf2, (uint32_t)b.w), acc, false);
}
__device__ __forceinline__ float cp_dot8(const raw16& a, const raw16& b, f16_t) {
  float acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(cp_h2, (uint32_t)a.x), __builtin_bit_cast(cp_h2, (uint32_t)b.x), 0.f, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(cp_h2, (uint32_t)a.y), __builtin_bit_cast(cp_h2, (uint32_t)b.y), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(cp_h2, (uint32_t)a.z), __builtin_bit_cast(cp_h2, (uint32_t)b.z), acc, false);
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(cp_h2, (uint32_t)a.w), __builtin_bit_cast(cp_h2, (uint32_t)b.w), acc, false);
}

// Sum over the LPS consecutive lanes of a sample (LPS = 8 .. 64, a power of two), left in the group's LAST lane: DPP row
// shifts (VALU rate, no LDS -- the ds_bpermute butterfly this replaces was the kernel's largest cost: 16 per item), then
// row broadcasts across the 16-lane rows.  Order of the additions: fixed.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float cp_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float cp_group_sum(float v, int LPS) {
  v += cp_dpp<0x111, 0xf>(v);               // row_shr:1
  v += cp_dpp<0x112, 0xf>(v);               // row_shr:2
  v += cp_dpp<0x114, 0xf>(v);               // row_shr:4  -> lanes 7 / 15 of a row: sums of 8
  if (LPS >= 16) v += cp_dpp<0x118, 0xf>(v);  // row_shr:8  -> lane 15: the row
  if (LPS >= 32) v += cp_dpp<0x142, 0xa>(v);  // row_bcast:15 into rows 1, 3 -> lanes 31 / 63: two rows
  if (LPS >= 64) v += cp_dpp<0x143, 0xc>(v);  // row_bcast:31 into rows 2, 3 -> lane 63: the wave
  return v;
}

// ---- forward: the column, written once, 16 B per lane, a wave stores 1 KB of consecutive column bytes ------------------
// Workgroup = NP consecutive positions x K2 taps; their tables once in LDS (one thread per sample); item = (sample, 8
// channels): four 16-B corner gathers (a sample's LPS = C / 8 lanes read C * 2 consecutive bytes of each corner pixel),
// the bilinear combination in fp32, one 16-B store.  XCD-aware: an XCD takes a contiguous range of positions, the pixels
// its workgroups gather stay in its L2.
template <typename T>
__global__ __launch_bounds__(256) void dcn_col_kernel(DcnShape s, const T* __restrict__ x, const T* __restrict__ offset,
                                                     const T* __restrict__ mask, T* __restrict__ col, int NP, int total,
                                                     const T* __restrict__ weight, T* __restrict__ wp, T* __restrict__ wt) {
  __shared__ CpEntry ent[CP_MAXS];
  const int tid = threadIdx.x;
  const int ncol = (total + 7) / 8 * 8;
  if ((int)blockIdx.x >= ncol) {
    // the weight pack (the GEMM that follows needs it, the column does not): the workgroups behind the column's, in the
    // same launch -- as its own launch in front of this one it was 4-6 us + an edge of every forward
    T(&tile)[32][33] = *reinterpret_cast<T(*)[32][33]>(&ent[0]);
    const int pi = (int)blockIdx.x - ncol, nci = s.C >> 5, nco = s.Co >> 5;
    const int tap = pi / (nci * nco), rem = pi - tap * (nci * nco);
    dcn_pack_body<T>(weight, wp, wt, s.Co, s.C, s.K2, (rem % nci) * 32, (rem / nci) * 32, tap, tile);
    return;
  }
  const int per_xcd = (total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= total) return;
  const long p0 = (long)logical * NP;
  const int nsamp = NP * s.K2;
  for (int i = tid; i < nsamp; i += 256) ent[i] = cp_make_entry<T, true>(s, offset, mask, p0 + i / s.K2, i % s.K2);
  __syncthreads();
  const int LPS = s.C >> 3;  // lanes per sample (a power of two: C in {64, 128, 256, 512})
  const int lps_shift = 31 - __builtin_clz(LPS);
  const int items = nsamp << lps_shift;
  const long rows = (s.P - p0) * s.K2;  // samples that exist from p0 on
  const char* xb = (const char*)x;
  const uint32_t pixbytes = (uint32_t)s.C * 2u;
  char* dst0 = (char*)col + (size_t)p0 * s.K2 * pixbytes;
  for (int it0 = 0; it0 < items; it0 += 512) {
    raw16 q[2][4];
    float w[2][4];
    bool ok[2];
    uint32_t fl[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int it = it0 + u * 256 + tid;
      const int sm = min(it >> lps_shift, nsamp - 1), sub = it & (LPS - 1);
      ok[u] = it < items && (it >> lps_shift) < rows;
      const CpEntry& e = ent[sm];
      fl[u] = e.flags;
      const uint4 px = *reinterpret_cast<const uint4*>(&e.pix[0]);
      const float4 wv = *reinterpret_cast<const float4*>(&e.w[0]);
      w[u][0] = wv.x; w[u][1] = wv.y; w[u][2] = wv.z; w[u][3] = wv.w;
      const uint32_t cofs = (uint32_t)sub * 16u;
      q[u][0] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.x * pixbytes + cofs));
      q[u][1] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.y * pixbytes + cofs));
      q[u][2] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.z * pixbytes + cofs));
      q[u][3] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.w * pixbytes + cofs));
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      float v[8];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        float f[8];
        // (a corner outside the image reads pixel 0 with weight 0: its VALUE must be 0 too -- 0 x Inf of an overflowed
        // 16-bit activation is NaN, where the reference's corner contributes exactly 0: deform_conv_cuda_kernel.cu:305-316)
        const raw16 qc = (fl[u] >> c) & 1u ? q[u][c] : raw16{0u, 0u, 0u, 0u};
        tc_unpack(qc, f, T{});
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = c == 0 ? w[u][c] * f[k] : v[k] + w[u][c] * f[k];
      }
      if (ok[u]) *reinterpret_cast<raw16*>(dst0 + (size_t)(it0 + u * 256 + tid) * 16) = tc_pack(v, T{});
    }
  }
}

// ---- backward: d(offset), d(mask) from dcol and the corners of x -------------------------------------------------------
// deform_conv_cuda_kernel.cu:454-520 (v1) / 1031-1064 (v2).  Same decomposition as the column kernel; a lane holds 8
// channels of one sample: D_c = sum over its channels of dcol * x[corner c] (4 dot products), summed over the sample's
// LPS lanes; then, with the per-sample weights,
//   d mask     = sum_c w_c D_c
//   d offset_h = m * (hw (D2 - D0) + lw (D3 - D1)),   d offset_w = m * (hh (D1 - D0) + lh (D3 - D2))
// (the derivative of the bilinear value by the sampling coordinate; corners outside the image contribute 0; a sample
// outside (-1, H) x (-1, W) gives zeros).  One owner per (position, tap): plain stores in the I/O dtype, no atomics,
// no fp32 staging tensor, no convert pass.
template <typename T>
__global__ __launch_bounds__(256) void dcn_coord_grad_kernel(DcnShape s, const T* __restrict__ x, const T* __restrict__ offset,
                                                            const T* __restrict__ mask, const T* __restrict__ dcol,
                                                            T* __restrict__ goff, T* __restrict__ gmask, int NP, int total) {
  __shared__ CpEntry ent[CP_MAXS];
  const int tid = threadIdx.x;
  const int per_xcd = (total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= total) return;
  const long p0 = (long)logical * NP;
  const int nsamp = NP * s.K2;
  for (int i = tid; i < nsamp; i += 256) ent[i] = cp_make_entry<T, false>(s, offset, mask, p0 + i / s.K2, i % s.K2);
  __syncthreads();
  const int LPS = s.C >> 3;
  const int lps_shift = 31 - __builtin_clz(LPS);
  const int items = nsamp << lps_shift;
  const long rows = (s.P - p0) * s.K2;
  const char* xb = (const char*)x;
  const uint32_t pixbytes = (uint32_t)s.C * 2u;
  const char* src0 = (const char*)dcol + (size_t)p0 * s.K2 * pixbytes;
  // (items is a multiple of 64 whenever LPS * K2 * NP is; a wave's lanes beyond `items` clamp to the last sample and
  // do not write)
  for (int it0 = 0; it0 < items; it0 += 512) {
    raw16 q[2][4], dq[2];
    int sraw[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {  // two items' loads in flight per thread
      const int it = it0 + u * 256 + tid;
      sraw[u] = it >> lps_shift;
      ok[u] = it < items && sraw[u] < rows;
      const CpEntry& e = ent[min(sraw[u], nsamp - 1)];
      const uint4 px = *reinterpret_cast<const uint4*>(&e.pix[0]);
      const uint32_t cofs = (uint32_t)(it & (LPS - 1)) * 16u;
      q[u][0] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.x * pixbytes + cofs));
      q[u][1] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.y * pixbytes + cofs));
      q[u][2] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.z * pixbytes + cofs));
      q[u][3] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.w * pixbytes + cofs));
      dq[u] = ok[u] ? *reinterpret_cast<const raw16*>(src0 + (size_t)it * 16) : raw16{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (it0 + u * 256 >= items) break;  // (uniform)
      const CpEntry& e = ent[min(sraw[u], nsamp - 1)];
      const uint32_t flags = e.flags;
      float D[4];
#pragma unroll
      for (int c = 0; c < 4; c++) D[c] = (flags & (1u << c)) ? cp_dot8(dq[u], q[u][c], T{}) : 0.f;
      // sum over the sample's lanes, left in the last of them
#pragma unroll
      for (int c = 0; c < 4; c++) D[c] = cp_group_sum(D[c], LPS);
      if (((it0 + u * 256 + tid) & (LPS - 1)) == LPS - 1 && ok[u]) {  // (the table holds the output indices: no divisions here)
        const float lh = e.lh, lw = e.lw, hh = 1.f - lh, hw = 1.f - lw, m = e.m;
        float gh = 0.f, gw = 0.f, gm = 0.f;
        if (flags & 16u) {
          gm = e.w[0] * D[0] + e.w[1] * D[1] + e.w[2] * D[2] + e.w[3] * D[3];
          gh = m * (hw * (D[2] - D[0]) + lw * (D[3] - D[1]));
          gw = m * (hh * (D[1] - D[0]) + lh * (D[3] - D[2]));
        }
        if (goff) {
          goff[e.oidx] = from_f32<T>(gh);
          goff[e.oidx + s.L] = from_f32<T>(gw);
        }
        if (gmask) gmask[e.midx] = from_f32<T>(gm);
      }
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------------------------
static int cp_positions_per_group(const DcnShape& s) {
  // >= ~8 workgroups per CU (32 resident waves each with 8 gathers in flight): the kernels are bound by the number of
  // loads in flight, not by their tables (profiles/r05/LOG.md)
  int np = 32;
  while (np > 1 && (long)s.P / np < 2048) np >>= 1;
  return np;
}

ColPathPlan dcn_colpath_plan(const DcnShape& s, int dtype) {
  ColPathPlan pl{};
  pl.ok = false;
  static const bool off = d2_prof_env("D2AMD_DCN_FUSED") != nullptr;  // A/B switch: the fused gather-MFMA kernels of r01-r04
  if (off) return pl;
  if (dtype != D2AMD_BF16 && dtype != D2AMD_F16) return pl;
  if (s.G != 1 || s.DG != 1 || s.K2 > 9 || s.P <= 0) return pl;
  if (!(s.C == 64 || s.C == 128 || s.C == 256 || s.C == 512) || s.Co % 64 != 0) return pl;
  if ((long)s.P * s.K2 * s.C >= (1l << 31) || (long)s.B * s.H * s.W * s.C * 2 >= (1l << 32)) return pl;  // (32-bit element indices)
  pl.fwd = gemm_nt_plan(s.P, s.Co, s.K2 * s.C);
  pl.bwd = gemm_nt_plan(s.P, s.K2 * s.C, s.Co);
  if (!pl.fwd.ok || !pl.bwd.ok) return pl;
  pl.NP = cp_positions_per_group(s);
  pl.col_bytes = (size_t)s.P * s.K2 * s.C * 2;
  pl.wpack_bytes = (size_t)s.Co * s.C * s.K2 * 2;
  pl.ok = true;
  return pl;
}

template <typename T>
int dcn_colpath_forward(const DcnShape& s, const ColPathPlan& pl, const void* x_nhwc, const void* offset, const void* mask,
                        const void* weight, const void* bias, void* out_nhwc, void* col, void* wpack, void* wt_keep,
                        hipStream_t st) {
  const int total = cdiv(s.P, pl.NP);
  {
    const int npack = (s.C / 32) * (s.Co / 32) * s.K2;
    const bool timed = timing_begin("dcn_fwd_col", st);
    hipLaunchKernelGGL((dcn_col_kernel<T>), dim3((total + 7) / 8 * 8 + npack), dim3(256), 0, st, s, (const T*)x_nhwc,
                       (const T*)offset, (const T*)mask, (T*)col, pl.NP, total, (const T*)weight, (T*)wpack, (T*)wt_keep);
    if (timed) timing_end("dcn_fwd_col", st);
    D2_LAUNCH_OK();
  }
  GemmNtArgs a{};
  a.X = col; a.Wn = wpack; a.out = out_nhwc; a.bias = bias;
  a.M = s.P; a.N = s.Co; a.K = s.K2 * s.C; a.ldx = a.K; a.ldw = a.K; a.ldo = s.Co;
  const bool timed = timing_begin("dcn_fwd_gemm", st);
  const int rc = gemm_nt_launch<T>(pl.fwd, a, st);
  if (timed) timing_end("dcn_fwd_gemm", st);
  return rc;
}

template <typename T>
int dcn_colpath_backward_data(const DcnShape& s, const ColPathPlan& pl, const void* x_nhwc, const void* offset, const void* mask,
                              const void* weight, const void* gout_nhwc, void* dcol, void* wpack, const void* wt_kept,
                              void* goff, void* gmask, hipStream_t st) {
  if (!wt_kept) {  // (no column was kept by the forward: pack here)
    hipLaunchKernelGGL((dcn_pack_weights_kernel<T>), dim3(s.C / 32, s.Co / 32, s.K2), dim3(256), 0, st, (const T*)weight, (T*)nullptr,
                       (T*)wpack, s.Co, s.C, s.K2);
    D2_LAUNCH_OK();
  }
  GemmNtArgs a{};
  a.X = gout_nhwc; a.Wn = wt_kept ? wt_kept : wpack; a.out = dcol; a.bias = nullptr;
  a.M = s.P; a.N = s.K2 * s.C; a.K = s.Co; a.ldx = s.Co; a.ldw = s.Co; a.ldo = a.N;
  {
    const bool timed = timing_begin("dcn_bwd_dcol_gemm", st);
    const int rc = gemm_nt_launch<T>(pl.bwd, a, st);
    if (timed) timing_end("dcn_bwd_dcol_gemm", st);
    if (rc) return rc;
  }
  if (goff || gmask) {
    const int total = cdiv(s.P, pl.NP);
    const bool timed = timing_begin("dcn_bwd_coord", st);
    hipLaunchKernelGGL((dcn_coord_grad_kernel<T>), dim3((total + 7) / 8 * 8), dim3(256), 0, st, s, (const T*)x_nhwc,
                       (const T*)offset, (const T*)mask, (const T*)dcol, (T*)goff, (T*)gmask, pl.NP, total);
    if (timed) timing_end("dcn_bwd_coord", st);
    D2_LAUNCH_OK();
  }
  return D2AMD_OK;
}

#define CP_INST(T)                                                                                                             \
  template int dcn_colpath_forward<T>(const DcnShape&, const ColPathPlan&, const void*, const void*, const void*, const void*, \
                                      const void*, void*, void*, void*, void*, hipStream_t);                                   \
  template int dcn_colpath_backward_data<T>(const DcnShape&, const ColPathPlan&, const void*, const void*, const void*,        \
                                            const void*, const void*, void*, void*, const void*, void*, void*, hipStream_t);
CP_INST(bf16_t)
CP_INST(f16_t)
#undef CP_INST

}  // namespace d2amd
