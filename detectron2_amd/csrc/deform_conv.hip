// Deformable convolution v1 / v2 (modulated) forward + backward for gfx950.
// Replaces detectron2/layers/csrc/deformable/deform_conv_cuda.cu:272-1221 and
// deform_conv_cuda_kernel.cu:216-452,785-1066 (im2col / col2im / col2im_coord + at::addmm_).
//
// Design (MI355X-first, not the reference's im2col-to-HBM + GEMM-per-image loop):
//   * implicit GEMM: the deformable "column" tile is gathered straight into LDS in MFMA operand
//     layout and consumed by v_mfma (32x32x16 bf16/f16, 32x32x2 f32 for exact-fp32 parity).
//     No column buffer in HBM, all images batched in one launch (N = B*Ho*Wo positions).
//   * K order is (tap, channel): one bilinear (idx, weight) table per position and tap serves
//     every channel of the deformable group; weights are repacked once to [g][tap][co][ci].
//   * activations are consumed as NHWC (a [B*H*W][C] matrix): a tap reads 4 x C contiguous
//     channels (8/16 B per lane, fully coalesced); NCHW inputs are transposed once per call
//     (<= 2 x |x| bytes of traffic against ~460 flop/B of work).
//   * backward = (1) dcol = W^T * dY by MFMA, consumed in-register/LDS by a fused epilogue that
//     scatters dX (fp32 atomics, NHWC) and reduces d(offset) / d(mask) over channels without a
//     column buffer; (2) dW = dY * col^T by MFMA with the re-gathered column tile, split over
//     position chunks (fp32 atomics); (3) d(bias) = row sums of dY.
// Accumulation is fp32 everywhere; column values are rounded to the I/O dtype before the MFMA
// exactly like the reference's column buffer.
#include <mutex>

#include "dcn_gemm.h"

namespace d2amd {

// weight (Co, Cg, K2) -> wr [g][tap][co][ci]  and  wt [g][tap][ci][co]
template <typename T>
__global__ void repack_weight_kernel(const T* __restrict__ w, T* __restrict__ wr, T* __restrict__ wt, int G, int Cog,
                                     int Cg, int K2) {
  long n = (long)G * Cog * Cg * K2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int tap = (int)(i % K2);
    int ci = (int)((i / K2) % Cg);
    int co = (int)((i / K2 / Cg) % Cog);
    int g = (int)(i / K2 / Cg / Cog);
    T v = w[i];
    if (wr) wr[(((long)g * K2 + tap) * Cog + co) * Cg + ci] = v;
    if (wt) wt[(((long)g * K2 + tap) * Cg + ci) * Cog + co] = v;
  }
}

// ---- per-position bilinear table ----------------------------------------------------------------
struct TapEntry {
  int idx[4];    // pixel index (b*H + y)*W + x of the 4 corners (0 when the corner is unused)
  float w[4];    // corner weights, 0 for out-of-range corners / samples (mask NOT folded in)
  float mask;    // modulation (1 for v1)
  float lh, lw;  // fractional parts (for the coordinate gradient)
  int inside;    // sample inside (-1, H) x (-1, W)
  int flags;     // bit t: corner t lies inside the image
};

template <typename T>
__device__ __forceinline__ TapEntry make_entry(const DcnShape& s, const T* __restrict__ offset,
                                               const T* __restrict__ mask, int p, int tap, int dgi) {
  TapEntry e;
#pragma unroll
  for (int t = 0; t < 4; t++) { e.idx[t] = 0; e.w[t] = 0.f; }
  e.mask = 0.f; e.lh = e.lw = 0.f; e.inside = 0; e.flags = 0;
  if (p >= s.P) return e;
  const int b = p / s.L, l = p - b * s.L;
  const int ho = l / s.Wo, wo = l - ho * s.Wo;
  const int i = tap / s.kw, j = tap - i * s.kw;
  const long obase = ((long)b * s.DG + dgi) * 2 * s.K2;
  const float off_h = to_f32(offset[(obase + 2 * tap) * s.L + l]);
  const float off_w = to_f32(offset[(obase + 2 * tap + 1) * s.L + l]);
  e.mask = mask ? to_f32(mask[(((long)b * s.DG + dgi) * s.K2 + tap) * s.L + l]) : 1.f;
  const float h_im = (float)(ho * s.sh - s.ph + i * s.dh) + off_h;
  const float w_im = (float)(wo * s.sw - s.pw + j * s.dw) + off_w;
  if (!(h_im > -1.f && w_im > -1.f && h_im < (float)s.H && w_im < (float)s.W)) return e;
  e.inside = 1;
  const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
  const float hh = 1.f - lh, hw = 1.f - lw;
  e.lh = lh; e.lw = lw;
  const long rowbase = (long)b * s.H;
  if (h_low >= 0 && w_low >= 0) { e.idx[0] = (int)((rowbase + h_low) * s.W + w_low); e.w[0] = hh * hw; e.flags |= 1; }
  if (h_low >= 0 && w_high <= s.W - 1) { e.idx[1] = (int)((rowbase + h_low) * s.W + w_high); e.w[1] = hh * lw; e.flags |= 2; }
  if (h_high <= s.H - 1 && w_low >= 0) { e.idx[2] = (int)((rowbase + h_high) * s.W + w_low); e.w[2] = lh * hw; e.flags |= 4; }
  if (h_high <= s.H - 1 && w_high <= s.W - 1) { e.idx[3] = (int)((rowbase + h_high) * s.W + w_high); e.w[3] = lh * lw; e.flags |= 8; }
  return e;
}

// valid[t] tells which corners exist (needed by the coordinate gradient, where a missing corner
// contributes 0 even though its bilinear weight is non-zero)
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}

// gather 4 channels (c .. c+3, absolute) of the sampled value for one table entry
template <typename T, bool VEC>
__device__ __forceinline__ void gather4(const T* __restrict__ x, int C, const TapEntry& e, int c, int cvalid,
                                        float (&val)[4]) {
  val[0] = val[1] = val[2] = val[3] = 0.f;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const float w = e.w[t];
    if (w == 0.f) continue;
    const T* px = x + (long)e.idx[t] * C + c;
    if (VEC) {
      float f[4];
      unpack4(*reinterpret_cast<const vec4<T>*>(px), f);
#pragma unroll
      for (int u = 0; u < 4; u++) val[u] += w * f[u];
    } else {
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (u < cvalid) val[u] += w * to_f32(px[u]);
    }
  }
}

// ---- forward ---------------------------------------------------------------------------------------
constexpr int FWD_BM = 128, FWD_BN = 64, FWD_THREADS = 256;

template <typename T, bool VEC>
__global__ __launch_bounds__(FWD_THREADS) void dcn_fwd_kernel(DcnShape s, const T* __restrict__ x /*NHWC*/,
                                                              const T* __restrict__ offset, const T* __restrict__ mask,
                                                              const T* __restrict__ wr, const T* __restrict__ bias,
                                                              T* __restrict__ out /*NCHW*/) {
  typedef Mma<T> M;
  constexpr int BK = M::BK, LD = BK + M::PAD;
  __shared__ __attribute__((aligned(16))) T As[FWD_BM * LD];
  __shared__ __attribute__((aligned(16))) T Bs[FWD_BN * LD];
  __shared__ TapEntry tab[FWD_BN];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int p0 = blockIdx.x * FWD_BN, co0 = blockIdx.y * FWD_BM, g = blockIdx.z;
  f32x16_t acc[2];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

  const int cg_lo = g * s.Cg, cg_hi = cg_lo + s.Cg;           // absolute channels of this conv group
  const int dg_first = cg_lo / s.cpg, dg_last = (cg_hi - 1) / s.cpg;
  for (int dgi = dg_first; dgi <= dg_last; dgi++) {
    const int c_lo = max(cg_lo, dgi * s.cpg), c_hi = min(cg_hi, (dgi + 1) * s.cpg);  // absolute sub-range
    for (int tap = 0; tap < s.K2; tap++) {
      __syncthreads();
      if (tid < FWD_BN) tab[tid] = make_entry<T>(s, offset, mask, p0 + tid, tap, dgi);
      __syncthreads();
      const T* wtap = wr + ((long)g * s.K2 + tap) * s.Cog * s.Cg;
      for (int cb = c_lo; cb < c_hi; cb += BK) {
        // ---- stage the weight tile As[m][kk] = W[co0+m][cb-cg_lo+kk]
        for (int e = tid; e < FWD_BM * (BK / 4); e += FWD_THREADS) {
          const int m = e / (BK / 4), q = e - m * (BK / 4);
          const int co = co0 + m, c = cb + 4 * q;
          float f[4] = {0.f, 0.f, 0.f, 0.f};
          if (co < s.Cog && c < c_hi) {
            const T* pw = wtap + (long)co * s.Cg + (c - cg_lo);
            if (VEC) {
              unpack4(*reinterpret_cast<const vec4<T>*>(pw), f);
            } else {
#pragma unroll
              for (int u = 0; u < 4; u++) if (c + u < c_hi) f[u] = to_f32(pw[u]);
            }
          }
          vec4<T> v;
          pack4(f, v);
          *reinterpret_cast<vec4<T>*>(&As[m * LD + 4 * q]) = v;
        }
        // ---- gather the column tile Bs[n][kk] (rounded to T like the reference's column buffer)
        for (int e = tid; e < FWD_BN * (BK / 4); e += FWD_THREADS) {
          const int n = e / (BK / 4), q = e - n * (BK / 4);
          const int c = cb + 4 * q;
          float val[4] = {0.f, 0.f, 0.f, 0.f};
          if (c < c_hi) {
            const TapEntry& te = tab[n];
            gather4<T, VEC>(x, s.C, te, c, c_hi - c, val);
            const float m = te.mask;
#pragma unroll
            for (int u = 0; u < 4; u++) val[u] *= m;
          }
          vec4<T> v;
          pack4(val, v);
          *reinterpret_cast<vec4<T>*>(&Bs[n * LD + 4 * q]) = v;
        }
        __syncthreads();
        // ---- MFMA: wave `wid` owns rows [32*wid, +32) x all 64 columns
        const T* arow = &As[(32 * wid + (lane & 31)) * LD];
        const T* brow0 = &Bs[(lane & 31) * LD];
        const T* brow1 = &Bs[(32 + (lane & 31)) * LD];
#pragma unroll
        for (int ks = 0; ks < BK / M::KSTEP; ks++) {
          const typename M::frag a = M::load(arow, ks, lane);
          const typename M::frag b0 = M::load(brow0, ks, lane);
          const typename M::frag b1 = M::load(brow1, ks, lane);
          acc[0] = M::mma(a, b0, acc[0]);
          acc[1] = M::mma(a, b1, acc[1]);
        }
        __syncthreads();
      }
    }
  }
  // ---- epilogue: out[b][g*Cog + co][l] (+ bias)
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int p = p0 + 32 * j + (lane & 31);
    if (p >= s.P) continue;
    const int b = p / s.L, l = p - b * s.L;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int co = co0 + 32 * wid + frag_row(r, lane);
      if (co >= s.Cog) continue;
      const int coa = g * s.Cog + co;
      float v = acc[j][r];
      if (bias) v += to_f32(bias[coa]);
      out[((long)b * s.Co + coa) * s.L + l] = from_f32<T>(v);
    }
  }
}

// ---- backward 1: dcol = W^T dY, fused with dX scatter + d(offset)/d(mask) reduction -------------------
// workgroup = (position tile of BN, tap, deformable group); loops over the channel blocks of the
// deformable group (M dimension, 64 channels at a time) and over Co (K dimension).
constexpr int BW_BM = 64, BW_BN = 64, BW_THREADS = 256;

template <typename T, bool VEC>
__global__ __launch_bounds__(BW_THREADS) void dcn_bwd_data_kernel(
    DcnShape s, const T* __restrict__ x /*NHWC*/, const T* __restrict__ offset, const T* __restrict__ mask,
    const T* __restrict__ wt /*[g][tap][ci][co]*/, const T* __restrict__ gout /*NHWC: [P][Co]*/,
    float* __restrict__ gx /*NHWC fp32*/, float* __restrict__ goff /*fp32 NCHW-like*/, float* __restrict__ gmask) {
  typedef Mma<T> M;
  constexpr int BK = M::BK, LD = BK + M::PAD;
  __shared__ __attribute__((aligned(16))) T As[BW_BM * LD];   // W^T tile: [ci][co]
  __shared__ __attribute__((aligned(16))) T Bs[BW_BN * LD];   // dY tile:  [n][co]
  __shared__ __attribute__((aligned(16))) float Cs[BW_BN][BW_BM + 4];  // dcol tile [n][ci]
  __shared__ TapEntry tab[BW_BN];
  __shared__ float red[3][BW_BN];                               // d(off_h), d(off_w), d(mask) per position
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int p0 = blockIdx.x * BW_BN, tap = blockIdx.y, dgi = blockIdx.z;
  if (tid < BW_BN) {
    tab[tid] = make_entry<T>(s, offset, mask, p0 + tid, tap, dgi);
    red[0][tid] = red[1][tid] = red[2][tid] = 0.f;
  }
  __syncthreads();
  const int d_lo = dgi * s.cpg, d_hi = d_lo + s.cpg;  // absolute channels of this deformable group
  // a deformable group may span several conv groups (and vice versa)
  for (int g = d_lo / s.Cg; g <= (d_hi - 1) / s.Cg; g++) {
    const int cg_lo = g * s.Cg;
    const int c_lo = max(d_lo, cg_lo), c_hi = min(d_hi, cg_lo + s.Cg);
    const T* wtap = wt + ((long)g * s.K2 + tap) * s.Cg * s.Cog;
    for (int cb = c_lo; cb < c_hi; cb += BW_BM) {
      f32x16_t acc[2];  // wave wid: rows (ci) [32*(wid&1), +32), cols (n) [32*(wid>>1), +32) ... 2x2 waves, 1 tile each
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
      const int wm = wid & 1, wn = wid >> 1;
      for (int k0 = 0; k0 < s.Cog; k0 += BK) {
        for (int e = tid; e < BW_BM * (BK / 4); e += BW_THREADS) {
          const int m = e / (BK / 4), q = e - m * (BK / 4);
          const int c = cb + m, co = k0 + 4 * q;
          float f[4] = {0.f, 0.f, 0.f, 0.f};
          if (c < c_hi && co < s.Cog) {
            const T* pw = wtap + (long)(c - cg_lo) * s.Cog + co;
            if (VEC) unpack4(*reinterpret_cast<const vec4<T>*>(pw), f);
            else {
#pragma unroll
              for (int u = 0; u < 4; u++) if (co + u < s.Cog) f[u] = to_f32(pw[u]);
            }
          }
          vec4<T> v; pack4(f, v);
          *reinterpret_cast<vec4<T>*>(&As[m * LD + 4 * q]) = v;
        }
        for (int e = tid; e < BW_BN * (BK / 4); e += BW_THREADS) {
          const int n = e / (BK / 4), q = e - n * (BK / 4);
          const int p = p0 + n, co = k0 + 4 * q;
          float f[4] = {0.f, 0.f, 0.f, 0.f};
          if (p < s.P && co < s.Cog) {
            const T* pg = gout + (long)p * s.Co + g * s.Cog + co;
            if (VEC) unpack4(*reinterpret_cast<const vec4<T>*>(pg), f);
            else {
#pragma unroll
              for (int u = 0; u < 4; u++) if (co + u < s.Cog) f[u] = to_f32(pg[u]);
            }
          }
          vec4<T> v; pack4(f, v);
          *reinterpret_cast<vec4<T>*>(&Bs[n * LD + 4 * q]) = v;
        }
        __syncthreads();
        const T* arow = &As[(32 * wm + (lane & 31)) * LD];
        const T* brow = &Bs[(32 * wn + (lane & 31)) * LD];
#pragma unroll
        for (int ks = 0; ks < BK / M::KSTEP; ks++)
          acc[0] = M::mma(M::load(arow, ks, lane), M::load(brow, ks, lane), acc[0]);
        __syncthreads();
      }
      // dcol tile -> LDS as [n][ci]
#pragma unroll
      for (int r = 0; r < 16; r++) Cs[32 * wn + (lane & 31)][32 * wm + frag_row(r, lane)] = acc[0][r];
      __syncthreads();
      // fused epilogue: item = (position n, channel quad q)
      for (int e = tid; e < BW_BN * (BW_BM / 4); e += BW_THREADS) {
        const int n = e / (BW_BM / 4), q = e - n * (BW_BM / 4);
        const int c = cb + 4 * q;
        const TapEntry& te = tab[n];
        float s_h = 0.f, s_w = 0.f, s_m = 0.f;
        if (c < c_hi && te.inside && (p0 + n) < s.P) {
          float dc[4];
          load4(&Cs[n][4 * q], dc);
          const int cv = min(4, c_hi - c);
          // corner values; corners outside the image contribute 0 (flags), also to the
          // coordinate gradient (deform_conv_cuda_kernel.cu:181-210)
          float v[4][4];
#pragma unroll
          for (int t = 0; t < 4; t++)
#pragma unroll
            for (int u = 0; u < 4; u++) v[t][u] = 0.f;
          const float lh = te.lh, lw = te.lw, hh = 1.f - lh, hw = 1.f - lw;
          const unsigned flags = (unsigned)te.flags;
#pragma unroll
          for (int t = 0; t < 4; t++) {
            if (!(flags & (1u << t))) continue;
            const T* px = x + (long)te.idx[t] * s.C + c;
            if (VEC) {
              float f[4];
              unpack4(*reinterpret_cast<const vec4<T>*>(px), f);
#pragma unroll
              for (int u = 0; u < 4; u++) v[t][u] = f[u];
            } else {
#pragma unroll
              for (int u = 0; u < 4; u++) if (u < cv) v[t][u] = to_f32(px[u]);
            }
          }
          const float m = te.mask;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (u >= cv) continue;
            const float d = dc[u];
            // value and coordinate derivatives of the bilinear sample (deform_conv_cuda_kernel.cu:164-214)
            const float val = te.w[0] * v[0][u] + te.w[1] * v[1][u] + te.w[2] * v[2][u] + te.w[3] * v[3][u];
            const float dvh = -hw * v[0][u] - lw * v[1][u] + hw * v[2][u] + lw * v[3][u];
            const float dvw = -hh * v[0][u] + hh * v[1][u] - lh * v[2][u] + lh * v[3][u];
            s_h += dvh * d * m;
            s_w += dvw * d * m;
            s_m += d * val;
            // dX scatter (deform_conv_cuda_kernel.cu:291-363): weight * dcol * mask
            const float dm = d * m;
#pragma unroll
            for (int t = 0; t < 4; t++)
              if ((flags & (1u << t)) && te.w[t] != 0.f) atomicAdd(gx + (long)te.idx[t] * s.C + c + u, te.w[t] * dm);
          }
        }
        // reduce the 16 quads of a position (consecutive lanes) and accumulate in LDS
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
          s_h += __shfl_xor(s_h, o);
          s_w += __shfl_xor(s_w, o);
          s_m += __shfl_xor(s_m, o);
        }
        if (q == 0) { red[0][n] += s_h; red[1][n] += s_w; red[2][n] += s_m; }
      }
      __syncthreads();
    }
  }
  // write d(offset) / d(mask) for (tap, dgi): exactly one workgroup owns each element -> plain stores
  if (tid < BW_BN) {
    const int p = p0 + tid;
    if (p < s.P) {
      const int b = p / s.L, l = p - b * s.L;
      const long ob = ((long)b * s.DG + dgi) * 2 * s.K2;
      if (goff) {
        goff[(ob + 2 * tap) * s.L + l] = red[0][tid];
        goff[(ob + 2 * tap + 1) * s.L + l] = red[1][tid];
      }
      if (gmask) gmask[(((long)b * s.DG + dgi) * s.K2 + tap) * s.L + l] = red[2][tid];
    }
  }
}

// ---- channel blocks: intersections of conv groups and deformable groups, cut into BN-channel blocks
__host__ __device__ inline int dcn_num_blocks(int C, int Cg, int cpg, int BN) {
  int n = 0;
  for (int c = 0; c < C;) {
    const int hi = min((c / Cg + 1) * Cg, (c / cpg + 1) * cpg);
    n += (hi - c + BN - 1) / BN;
    c = hi;
  }
  return n;
}
__device__ inline bool dcn_locate_block(int C, int Cg, int cpg, int BN, int blk, int& cb, int& c_hi) {
  for (int c = 0; c < C;) {
    const int hi = min((c / Cg + 1) * Cg, (c / cpg + 1) * cpg);
    const int nb = (hi - c + BN - 1) / BN;
    if (blk < nb) { cb = c + blk * BN; c_hi = hi; return true; }
    blk -= nb;
    c = hi;
  }
  return false;
}

// ---- backward 2: dW[g][tap][co][ci] += sum_p dY[p][co] * col[p][ci]  (split over position chunks) -------
constexpr int WG_BM = 64, WG_BN = 64, WG_THREADS = 256;

template <typename T, bool VEC>
__global__ __launch_bounds__(WG_THREADS) void dcn_bwd_weight_kernel(
    DcnShape s, const T* __restrict__ x /*NHWC*/, const T* __restrict__ offset, const T* __restrict__ mask,
    const T* __restrict__ gout_nchw, float* __restrict__ gwr /*[g][tap][co][ci] fp32*/, int pchunk, int nblk) {
  typedef Mma<T> M;
  constexpr int BK = M::BK, LD = BK + M::PAD;
  __shared__ __attribute__((aligned(16))) T As[WG_BM * LD];   // dY tile  [co][p]
  __shared__ __attribute__((aligned(16))) T Bs[WG_BN * LD];   // col tile [ci][p]
  __shared__ TapEntry tab[Mma<T>::BK];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tap = blockIdx.z;
  const int blk = blockIdx.y % nblk, co_tile = blockIdx.y / nblk;
  int cb, c_hi;
  if (!dcn_locate_block(s.C, s.Cg, s.cpg, WG_BN, blk, cb, c_hi)) return;
  const int g = cb / s.Cg, dgi = cb / s.cpg;
  const int co0 = co_tile * WG_BM;
  const int pbeg = blockIdx.x * pchunk, pend = min(s.P, pbeg + pchunk);
  const int wm = wid & 1, wn = wid >> 1;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  for (int pk = pbeg; pk < pend; pk += BK) {
    __syncthreads();
    if (tid < BK) tab[tid] = make_entry<T>(s, offset, mask, (pk + tid < pend) ? pk + tid : s.P, tap, dgi);
    // dY tile: As[m][kk] = dY[b][g*Cog + co0+m][l] for p = pk+kk  (coalesced along l)
    for (int e = tid; e < WG_BM * BK; e += WG_THREADS) {
      const int m = e / BK, kk = e - m * BK;
      const int p = pk + kk, co = co0 + m;
      T v = from_f32<T>(0.f);
      if (p < pend && co < s.Cog) {
        const int b = p / s.L, l = p - b * s.L;
        v = gout_nchw[((long)b * s.Co + g * s.Cog + co) * s.L + l];
      }
      As[m * LD + kk] = v;
    }
    __syncthreads();
    // col tile: Bs[ci][kk]; item = (position kk, channel quad q)
    for (int e = tid; e < BK * (WG_BN / 4); e += WG_THREADS) {
      const int kk = e / (WG_BN / 4), q = e - kk * (WG_BN / 4);
      const int c = cb + 4 * q;
      float val[4] = {0.f, 0.f, 0.f, 0.f};
      if (c < c_hi) {
        const TapEntry& te = tab[kk];
        gather4<T, VEC>(x, s.C, te, c, c_hi - c, val);
#pragma unroll
        for (int u = 0; u < 4; u++) val[u] = (c + u < c_hi) ? val[u] * te.mask : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) Bs[(4 * q + u) * LD + kk] = from_f32<T>(val[u]);
    }
    __syncthreads();
    const T* arow = &As[(32 * wm + (lane & 31)) * LD];
    const T* brow = &Bs[(32 * wn + (lane & 31)) * LD];
#pragma unroll
    for (int ks = 0; ks < BK / M::KSTEP; ks++) acc = M::mma(M::load(arow, ks, lane), M::load(brow, ks, lane), acc);
  }
  // rows = co, cols = ci (absolute channel cb + col) -> gwr[g][tap][co][ci_rel]
  const int c = cb + 32 * wn + (lane & 31);
  if (c < c_hi) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int co = co0 + 32 * wm + frag_row(r, lane);
      if (co < s.Cog) atomicAdd(gwr + (((long)g * s.K2 + tap) * s.Cog + co) * s.Cg + (c - g * s.Cg), acc[r]);
    }
  }
}

// gwr [g][tap][co][ci] fp32 -> grad_weight (Co, Cg, K2) T
template <typename T>
__global__ void unpack_gw_kernel(const float* __restrict__ gwr, T* __restrict__ gw, int G, int Cog, int Cg, int K2) {
  long n = (long)G * Cog * Cg * K2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int tap = (int)(i % K2);
    int ci = (int)((i / K2) % Cg);
    int co = (int)((i / K2 / Cg) % Cog);
    int g = (int)(i / K2 / Cg / Cog);
    gw[i] = from_f32<T>(gwr[(((long)g * K2 + tap) * Cog + co) * Cg + ci]);
  }
}

template <typename T>
__global__ void cvt_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = from_f32<T>(src[i]);
}

// two conversions in one launch (the offset and the mask gradient of a backward call: a launch is ~4.6 us of a step of
// small kernels)
template <typename T>
__global__ void cvt2_kernel(const float* __restrict__ a, T* __restrict__ da, long na, const float* __restrict__ b,
                            T* __restrict__ db, long nb) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (long)gridDim.x * blockDim.x) {
    if (i < na) da[i] = from_f32<T>(a[i]);
    else db[i - na] = from_f32<T>(b[i - na]);
  }
}
// d(bias)[co] = sum_{b,l} dY[b][co][l]: one workgroup per output channel
template <typename T>
__global__ __launch_bounds__(256) void dcn_bias_grad_kernel(const T* __restrict__ gout, T* __restrict__ gb, int B, int Co,
                                                            int L) {
  __shared__ float part[4];
  const int co = blockIdx.x;
  float acc = 0.f;
  for (int b = 0; b < B; b++) {
    const T* row = gout + ((long)b * Co + co) * L;
    for (int l = threadIdx.x; l < L; l += 256) acc += to_f32(row[l]);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) gb[co] = from_f32<T>(part[0] + part[1] + part[2] + part[3]);
}

// ---- host side -------------------------------------------------------------------------------------
static size_t al(size_t x) { return (x + 255) / 256 * 256; }

struct DcnWs {
  DcnGatherWs gather;  // column-gather backward (16-bit MFMA path, deformable_groups == 1)
  void* gx_t;          // its dX in the I/O dtype, NHWC
  bool use_gather;
  bool nhwc;  // the caller's x / out / grads are channels_last (d2amd_dcn_params::layout)
  void *x_nhwc, *wr, *wt, *gout_nhwc;
  float *gx, *goff, *gmask, *gwr;
  int dtype;
  void* tc_wp;        // fragment-ordered weights of the 16-bit MFMA path
  float* tc_partial;  // fp32 partial outputs when its reduction is split
  TcPlan tc;
  void* col_saved;    // the column the training forward saved / the backward is handed (d2amd_deform_conv_*_saved)
  ColPathPlan cp;     // channels_last 16-bit: column + dense GEMM (dcn_colpath.hip); cp.ok false: the fused kernels
  void* cp_col;       // forward: scratch column when the caller keeps none
  void* cp_wpack;     // packed weights of the GEMM
  size_t total;
};

// (keeps_col: the forward is handed a column to keep -- the column path then needs no scratch column of its own)
static DcnWs carve_ws(const DcnShape& s, int dtype, bool backward, bool nhwc, void* base, bool keeps_col = false) {
  DcnWs w{};
  w.dtype = dtype;
  w.nhwc = nhwc;
  if (nhwc) w.cp = dcn_colpath_plan(s, dtype);
  const size_t es = dtype_size(dtype);
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = base ? (char*)base + off : nullptr; off += al(bytes); return r; };
  w.x_nhwc = take((size_t)s.B * s.H * s.W * s.C * es);
  const size_t wbytes = (size_t)s.Co * s.Cg * s.K2 * es;
  if (!backward) {
    w.tc = dcn_tc_plan_fwd(s, dtype);
    if (w.cp.ok) {
      w.cp_wpack = take(w.cp.wpack_bytes);
      if (!keeps_col) w.cp_col = take(w.cp.col_bytes);
    } else if (w.tc.ok) {
      w.tc_wp = take(w.tc.wp_bytes);
      w.tc_partial = (float*)take(w.tc.partial_bytes);
    } else {
      w.wr = take(wbytes);
    }
  } else {
    w.wt = take(wbytes);
    w.gout_nhwc = take((size_t)s.P * s.Co * es);
    w.gx = (float*)take((size_t)s.B * s.H * s.W * s.C * 4);
    w.goff = (float*)take((size_t)s.B * s.DG * 2 * s.K2 * s.L * 4);
    w.gmask = (float*)take((size_t)s.B * s.DG * s.K2 * s.L * 4);
    const TcBwPlan bp = dcn_tc_plan_bwd(s, dtype);
    w.use_gather = bp.ok && bp.gather;
    // (the gather's pixel counters sit right behind the two accumulators: one zero launch covers the three regions)
    if (w.use_gather) w.gather.cnt = (int*)take(dcn_gather_cnt_bytes(s));
    {  // fp32 staging of the weight gradient: [g][tap][co][ci] (generic kernels: atomics), or the MFMA kernel's partial tiles
      const TcBwwPlan wp = dcn_tc_plan_bww(s, dtype);
      const BwwGemmPlan gp = dcn_bww_gemm_plan(s, dtype, nhwc);  // (the saved-column GEMM's split-K partial tiles)
      size_t stage = (size_t)s.Co * s.Cg * s.K2 * 4;
      if (wp.ok && wp.partial_bytes > stage) stage = wp.partial_bytes;
      if (gp.ok && gp.partial_bytes > stage) stage = gp.partial_bytes;
      w.gwr = (float*)take(stage);
    }
    if (w.use_gather) {
      w.gather.col = take(dcn_gather_col_bytes(s, es));
      w.gather.lists = take(dcn_gather_list_bytes(s));
      w.gather.ovf = take(dcn_gather_ovf_bytes(s));
      w.gx_t = take((size_t)s.B * s.H * s.W * s.C * es);
    }
  }
  w.total = off;
  return w;
}

static int check_params(const d2amd_dcn_params* p, DcnShape& s, const char* who) {
  D2_CHECK_ARG(p != nullptr, "%s: null params", who);
  // deform_conv_cuda.cu:140-270 (shape_check)
  D2_CHECK_ARG(p->kh > 0 && p->kw > 0, "kernel size should be greater than zero, but got kH: %d kW: %d", p->kh, p->kw);
  D2_CHECK_ARG(p->stride_h > 0 && p->stride_w > 0, "stride should be greater than zero, but got dH: %d dW: %d",
               p->stride_h, p->stride_w);
  D2_CHECK_ARG(p->dil_h > 0 && p->dil_w > 0, "dilation should be greater than 0, but got dilationH: %d dilationW: %d",
               p->dil_h, p->dil_w);
  D2_CHECK_ARG(p->B >= 0 && p->C > 0 && p->Co > 0 && p->H > 0 && p->W > 0, "%s: bad tensor shape", who);
  D2_CHECK_ARG(p->groups > 0 && p->C % p->groups == 0 && p->Co % p->groups == 0,
               "input/output channels must be divisible by groups");
  D2_CHECK_ARG(p->deformable_groups > 0 && p->C % p->deformable_groups == 0,
               "input channels must divide deformable group size");
  D2_CHECK_ARG(p->H + 2 * p->pad_h >= p->dil_h * (p->kh - 1) + 1 && p->W + 2 * p->pad_w >= p->dil_w * (p->kw - 1) + 1,
               "input image is smaller than kernel");
  s = make_shape(p);
  D2_CHECK_ARG(s.Ho >= 1 && s.Wo >= 1, "Calculated output size: (%d x %d). Output size is too small", s.Ho, s.Wo);
  D2_CHECK_ARG((long)s.B * s.H * s.W < (1l << 31), "%s: too many pixels for 32-bit indexing", who);
  return D2AMD_OK;
}

template <typename T>
static int cvt_grads(const DcnShape& s, const DcnWs& w, void* goffset, void* gmask, hipStream_t st) {
  const long no = goffset ? (long)s.B * s.DG * 2 * s.K2 * s.L : 0, nm = gmask ? (long)s.B * s.DG * s.K2 * s.L : 0;
  if (no + nm == 0) return D2AMD_OK;
  const long blocks = cdiv(no + nm, 256) > 8192 ? 8192 : cdiv(no + nm, 256);
  hipLaunchKernelGGL((cvt2_kernel<T>), dim3(blocks), dim3(256), 0, st, w.goff, (T*)goffset, no, w.gmask, (T*)gmask, nm);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

static bool vec_ok(const DcnShape& s) {
  return s.C % 4 == 0 && s.Cg % 4 == 0 && s.cpg % 4 == 0 && s.Cog % 4 == 0 && s.Co % 4 == 0;
}

static DcnSide* dcn_side(hipStream_t caller);  // (below)

template <typename T>
static int fwd_host(const DcnShape& s, const void* x, const void* offset, const void* mask, const void* weight,
                    const void* bias, void* out, const DcnWs& w, hipStream_t st) {
  if (s.B == 0) return D2AMD_OK;
  if (w.nhwc) {  // a channels_last caller: x is what the kernels read, out is written [position][Co]
    if constexpr (sizeof(T) == 2) {
      if (w.cp.ok && w.col_saved) (void)dcn_side(st);  // (a training forward: the backward's second stream is created and
                                                       // probed HERE -- two stream synchronisations the first time --
                                                       // not inside the first backward call: ADVICE r04)
      if (w.cp.ok)
        return dcn_colpath_forward<T>(s, w.cp, x, offset, mask, weight, bias, out, w.col_saved ? w.col_saved : w.cp_col,
                                      w.cp_wpack, w.col_saved ? (char*)w.col_saved + al(w.cp.col_bytes) : nullptr, st);
      if (w.tc.ok)
        return dcn_tc_forward<T>(s, w.tc, x, offset, mask, weight, bias, out, w.tc_wp, w.tc_partial, st, true, w.col_saved);
    }
    set_error("deform_conv_forward: NHWC input is served by the 16-bit MFMA path only (this shape / dtype is not)");
    return D2AMD_EUNSUPPORTED;
  }
  int rc = launch_transpose<T, T>((const T*)x, (T*)w.x_nhwc, s.B, s.C, s.H * s.W, st);
  if (rc) return rc;
  if constexpr (sizeof(T) == 2) {
    if (w.tc.ok)
      return dcn_tc_forward<T>(s, w.tc, w.x_nhwc, offset, mask, weight, bias, out, w.tc_wp, w.tc_partial, st, false,
                               w.col_saved);
  }
  const long nw = (long)s.Co * s.Cg * s.K2;
  hipLaunchKernelGGL((repack_weight_kernel<T>), dim3(cdiv(nw, 256) > 4096 ? 4096 : cdiv(nw, 256)), dim3(256), 0, st,
                     (const T*)weight, (T*)w.wr, (T*)nullptr, s.G, s.Cog, s.Cg, s.K2);
  D2_LAUNCH_OK();
  dim3 grid(cdiv(s.P, FWD_BN), cdiv(s.Cog, FWD_BM), s.G);
  D2_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "deform_conv: too many output channels / groups");
  if (vec_ok(s))
    hipLaunchKernelGGL((dcn_fwd_kernel<T, true>), grid, dim3(FWD_THREADS), 0, st, s, (const T*)w.x_nhwc,
                       (const T*)offset, (const T*)mask, (const T*)w.wr, (const T*)bias, (T*)out);
  else
    hipLaunchKernelGGL((dcn_fwd_kernel<T, false>), grid, dim3(FWD_THREADS), 0, st, s, (const T*)w.x_nhwc,
                       (const T*)offset, (const T*)mask, (const T*)w.wr, (const T*)bias, (T*)out);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

// ---- does a candidate side stream run BESIDE its caller's stream?  HIP maps streams onto a few hardware queues (4 by
// default, least-used first): in a process that has created dozens of streams a new one can land on its caller's queue,
// and the "second stream" of the backward then runs behind the first (dcn_r50 in the bench's in-line run: 3.88 ms
// against 3.28).  The probe: a kernel on the caller's stream waits -- at most 200 us -- for a flag that a kernel on the
// candidate sets; it sees the flag only if the two were on the device together.
__global__ void dcn_probe_wait_kernel(int* flag, unsigned long long max_ticks, int* saw) {
  const unsigned long long t0 = wall_clock64();
  int v = 0;
  while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && wall_clock64() - t0 < max_ticks)
    __builtin_amdgcn_s_sleep(16);
  *saw = v;
}
__global__ void dcn_probe_set_kernel(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 1: overlaps, 0: runs behind the caller's stream, -1: could not tell (a capture in progress, an error)
static int dcn_stream_runs_beside(hipStream_t caller, hipStream_t cand) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(caller, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return -1;
  }
  int* d = nullptr;
  if (hipMalloc(&d, 2 * sizeof(int)) != hipSuccess) return -1;
  int res = -1, h[2] = {0, 0};
  if (hipMemsetAsync(d, 0, 2 * sizeof(int), caller) == hipSuccess && hipStreamSynchronize(caller) == hipSuccess) {
    hipLaunchKernelGGL(dcn_probe_wait_kernel, dim3(1), dim3(1), 0, caller, d, 20000ull /* x 10 ns */, d + 1);
    hipLaunchKernelGGL(dcn_probe_set_kernel, dim3(1), dim3(1), 0, cand, d);
    if (hipStreamSynchronize(caller) == hipSuccess && hipStreamSynchronize(cand) == hipSuccess &&
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
      res = h[1] ? 1 : 0;
  }
  (void)hipFree(d);
  return res;
}

// the second stream of a backward call (DcnSide): one per (device, caller's stream), created on first use -- two host
// threads running backward passes on different streams of one device must not share events.  D2AMD_DCN_NO_SIDE: none.
// First use outside a graph capture costs a few hundred microseconds and two synchronisations of the caller's stream
// (the probe above, over up to 4 candidate streams); D2AMD_DCN_NO_PROBE: the first candidate, unprobed.
static DcnSide* dcn_side(hipStream_t caller) {
  struct Slot { int dev; hipStream_t caller; DcnSide side; };
  static Slot slots[32];
  static int nslots = 0;
  static std::mutex mu;
  static const bool off = d2_prof_env("D2AMD_DCN_NO_SIDE") != nullptr;
  int dev = 0;
  if (off || hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < nslots; i++)
    if (slots[i].dev == dev && slots[i].caller == caller) return &slots[i].side;
  if (nslots == 32) return nullptr;  // (more caller streams than anyone uses: those calls stay on one stream)
  DcnSide t{};
  {
    // Priority of the side stream (D2AMD_DCN_SIDE_PRIO = low | flat | high).  A LOW-priority stream looked right -- the side
    // work fills what the data-gradient kernel leaves idle -- and measures the same as the default in a process that runs
    // nothing else (dcn_r50 3.28-3.30 ms); in a process with other streams alive (the bench's in-line extra workloads
    // behind the Mask R-CNN step) it made the same step 8.5-8.6 ms, against 3.78 on one stream and 3.88 with the default
    // priority (profiles/r04/LOG.md).
    int lo = 0, hi = 0, prio = 0;
    static const char* mode = d2_prof_env("D2AMD_DCN_SIDE_PRIO");
    if (mode && mode[0] != 'f' && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) prio = mode[0] == 'l' ? lo : hi;
    static const bool no_probe = d2_prof_env("D2AMD_DCN_NO_PROBE") != nullptr;
    hipStream_t cand[4] = {nullptr, nullptr, nullptr, nullptr};
    int ncand = 0, pick = 0;
    for (; ncand < (no_probe ? 1 : 4); ncand++) {
      if (hipStreamCreateWithPriority(&cand[ncand], hipStreamNonBlocking, prio) != hipSuccess) break;
      if (no_probe) { ncand++; break; }
      const int r = dcn_stream_runs_beside(caller, cand[ncand]);
      if (r != 0) { pick = ncand; ncand++; break; }  // beside the caller (or unknown: nothing better to go by)
    }
    if (ncand == 0) return nullptr;
    if (pick >= ncand) pick = 0;
    t.stream = cand[pick];
    for (int i = 0; i < ncand; i++)
      if (i != pick) (void)hipStreamDestroy(cand[i]);
  }
  if (hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&t.bin, hipEventDisableTiming) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&t.join, hipEventDisableTiming) != hipSuccess) return nullptr;
  slots[nslots] = Slot{dev, caller, t};
  return &slots[nslots++].side;
}

template <typename T>
struct BwwGemmCall { const DcnShape* s; const BwwGemmPlan* gp; const void* gout; const void* col; float* gwr; void* gweight; };
template <typename T>
static int bww_gemm_on(void* ctx, hipStream_t side) {
  const BwwGemmCall<T>* c = (const BwwGemmCall<T>*)ctx;
  return dcn_bww_gemm<T>(*c->s, *c->gp, c->gout, c->col, c->gwr, c->gweight, side);
}

template <typename T>
static int bwd_host(const DcnShape& s, const void* x, const void* offset, const void* mask, const void* weight,
                    const void* gout, void* gin, void* goffset, void* gmask, void* gweight, void* gbias,
                    const DcnWs& w, hipStream_t st) {
  if (s.B == 0) {
    if (gweight) { const int zrc = zero_async(gweight, (size_t)s.Co * s.Cg * s.K2 * sizeof(T), st); if (zrc) return zrc; }
    if (gbias) { const int zrc = zero_async(gbias, (size_t)s.Co * sizeof(T), st); if (zrc) return zrc; }
    return D2AMD_OK;
  }
  constexpr bool is32 = sizeof(T) == 4;
  const bool vec = vec_ok(s);
  int rc = 0;
  if (w.nhwc) {
    // A channels_last caller (d2amd_dcn_params::layout = NHWC): x and grad_out ARE what the kernels read, the column
    // gather writes grad_input in place -- 3 of the 4 transposes of a block's forward + backward are gone; the
    // weight-gradient kernel still reads dY position-major per channel: one transpose into the workspace.
    if constexpr (is32) {
      set_error("deform_conv_backward: NHWC is served by the 16-bit MFMA path only");
      return D2AMD_EUNSUPPORTED;
    } else {
      const TcBwPlan bp = dcn_tc_plan_bwd(s, (int)w.dtype);
      const TcBwwPlan wp = dcn_tc_plan_bww(s, (int)w.dtype);
      const bool colpath = w.cp.ok && bp.ok && w.use_gather;
      if (!colpath && !(bp.ok && w.use_gather && wp.ok)) {
        set_error("deform_conv_backward: NHWC is served by the 16-bit MFMA path only (this shape is not)");
        return D2AMD_EUNSUPPORTED;
      }
      const bool need_data = gin || goffset || (gmask && mask);
      const BwwGemmPlan gp = dcn_bww_gemm_plan(s, (int)w.dtype, true);
      const bool gemm_w = gweight && w.col_saved && gp.ok;  // dW = dY^T col from the column the forward saved
      bool gemm_done = false;
      if (need_data) {
        float* goff_f = goffset ? w.goff : nullptr;
        float* gmask_f = (gmask && mask) ? w.gmask : nullptr;
        // second stream: the sample binning and the weight-gradient GEMM (dY and the saved column in, its own partial
        // tiles and grad_weight out: nothing the data path touches) run beside the data-gradient kernel
        DcnSide* sd = dcn_side(st);
        DcnSide side{};
        BwwGemmCall<T> call{&s, &gp, gout, w.col_saved, w.gwr, gweight};
        static const int side_mode = d2_prof_env("D2AMD_DCN_SIDE_MODE") ? atoi(d2_prof_env("D2AMD_DCN_SIDE_MODE")) : 3;  // A/B: 1 = binning only, 2 = GEMM only
        if (sd) {
          side = *sd;
          side.work = (gemm_w && (side_mode & 2)) ? bww_gemm_on<T> : nullptr;
          side.ctx = &call;
          side.fork = (side_mode & 1) ? side.fork : nullptr;  // (null: the binning stays on the caller's stream)
        }
        if (colpath) {  // (d offset / d mask leave the coordinate-gradient kernel in the I/O dtype: no fp32 staging, no convert)
          rc = dcn_tc_backward_data_gather<T>(s, bp, x, offset, mask, weight, gout, gin, nullptr, nullptr, w.wt, w.gather, st,
                                              sd ? &side : nullptr, &w.cp, goffset, (gmask && mask) ? gmask : nullptr,
                                              w.col_saved ? (const char*)w.col_saved + al(w.cp.col_bytes) : nullptr);
          if (rc) return rc;
        } else {
          rc = dcn_tc_backward_data_gather<T>(s, bp, x, offset, mask, weight, gout, gin, goff_f, gmask_f, w.wt, w.gather, st,
                                              sd ? &side : nullptr);
          if (rc) return rc;
          rc = cvt_grads<T>(s, w, goffset, (gmask && mask) ? gmask : nullptr, st);
          if (rc) return rc;
        }
        if (sd) {
          D2_HIP_OK(hipStreamWaitEvent(st, side.join, 0));
          gemm_done = side.work != nullptr;
        }
      }
      if ((gweight && !gemm_w) || gbias) {
        rc = launch_transpose<T, T>((const T*)gout, (T*)w.gout_nhwc, s.B, s.L, s.Co, st);  // -> [b][Co][l]
        if (rc) return rc;
      }
      if (gemm_done) {
      } else if (gemm_w) {
        rc = dcn_bww_gemm<T>(s, gp, gout, w.col_saved, w.gwr, gweight, st);
        if (rc) return rc;
      } else if (gweight) {
        rc = dcn_tc_backward_weight<T>(s, wp, x, offset, mask, w.gout_nhwc, w.gwr, gweight, st);
        if (rc) return rc;
      }
      if (gbias) {
        hipLaunchKernelGGL((dcn_bias_grad_kernel<T>), dim3(s.Co), dim3(256), 0, st, (const T*)w.gout_nhwc, (T*)gbias, s.B,
                           s.Co, s.L);
        D2_LAUNCH_OK();
      }
      return D2AMD_OK;
    }
  }
  rc = launch_transpose<T, T>((const T*)x, (T*)w.x_nhwc, s.B, s.C, s.H * s.W, st);
  if (rc) return rc;
  const bool need_data = gin || goffset || (gmask && mask);
  if (need_data) {
    rc = launch_transpose<T, T>((const T*)gout, (T*)w.gout_nhwc, s.B, s.Co, s.L, st);
    if (rc) return rc;
    float* goff_f = goffset ? (is32 ? (float*)goffset : w.goff) : nullptr;
    float* gmask_f = (gmask && mask) ? (is32 ? (float*)gmask : w.gmask) : nullptr;
    bool tc_done = false, gathered = false;
    if constexpr (!is32) {
      const TcBwPlan bp = dcn_tc_plan_bwd(s, sizeof(T) == 2 ? (int)w.dtype : D2AMD_F32);
      if (bp.ok && w.use_gather) {  // column gather: dX written once per pixel in the I/O dtype, no atomics, no zero fill
        rc = dcn_tc_backward_data_gather<T>(s, bp, w.x_nhwc, offset, mask, weight, w.gout_nhwc, gin ? w.gx_t : nullptr,
                                            goff_f, gmask_f, w.wt, w.gather, st);
        if (rc) return rc;
        tc_done = gathered = true;
      }
    }
    if (!gathered) { const int zrc = zero_async(w.gx, (size_t)s.B * s.H * s.W * s.C * 4, st); if (zrc) return zrc; }
    if constexpr (!is32) {
      const TcBwPlan bp = dcn_tc_plan_bwd(s, sizeof(T) == 2 ? (int)w.dtype : D2AMD_F32);
      if (bp.ok && !gathered) {  // 16-bit MFMA path with global atomics (deform_conv_tc.hip); w.wt holds its packed weights
        rc = dcn_tc_backward_data<T>(s, bp, w.x_nhwc, offset, mask, weight, w.gout_nhwc, w.gx, goff_f, gmask_f, w.wt, st);
        if (rc) return rc;
        tc_done = true;
      }
    }
    const long nw = (long)s.Co * s.Cg * s.K2;
    if (!tc_done) {
      hipLaunchKernelGGL((repack_weight_kernel<T>), dim3(cdiv(nw, 256) > 4096 ? 4096 : cdiv(nw, 256)), dim3(256), 0,
                         st, (const T*)weight, (T*)nullptr, (T*)w.wt, s.G, s.Cog, s.Cg, s.K2);
      D2_LAUNCH_OK();
    }
    dim3 grid(cdiv(s.P, BW_BN), s.K2, s.DG);
    D2_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "deform_conv: kernel / deformable groups too large");
    if (tc_done) {
    } else if (vec)
      hipLaunchKernelGGL((dcn_bwd_data_kernel<T, true>), grid, dim3(BW_THREADS), 0, st, s, (const T*)w.x_nhwc,
                         (const T*)offset, (const T*)mask, (const T*)w.wt, (const T*)w.gout_nhwc, w.gx, goff_f, gmask_f);
    else
      hipLaunchKernelGGL((dcn_bwd_data_kernel<T, false>), grid, dim3(BW_THREADS), 0, st, s, (const T*)w.x_nhwc,
                         (const T*)offset, (const T*)mask, (const T*)w.wt, (const T*)w.gout_nhwc, w.gx, goff_f, gmask_f);
    D2_LAUNCH_OK();
    if (gin) {
      if (gathered) rc = launch_transpose<T, T>((const T*)w.gx_t, (T*)gin, s.B, s.H * s.W, s.C, st);
      else rc = launch_transpose<float, T>(w.gx, (T*)gin, s.B, s.H * s.W, s.C, st);
      if (rc) return rc;
    }
    if (!is32) {
      rc = cvt_grads<T>(s, w, goffset, (gmask && mask) ? gmask : nullptr, st);
      if (rc) return rc;
    }
  }
  if (gweight) {
    bool tc_w = false;
    if constexpr (!is32) {
      const BwwGemmPlan gp = dcn_bww_gemm_plan(s, (int)w.dtype, false);
      if (w.col_saved && gp.ok) {  // dW = dY^T col from the column the forward saved (dY as [P][Co])
        if (!need_data) {
          rc = launch_transpose<T, T>((const T*)gout, (T*)w.gout_nhwc, s.B, s.Co, s.L, st);
          if (rc) return rc;
        }
        rc = dcn_bww_gemm<T>(s, gp, w.gout_nhwc, w.col_saved, w.gwr, gweight, st);
        if (rc) return rc;
        tc_w = true;
      }
    }
    if constexpr (!is32) {
      const TcBwwPlan wp = dcn_tc_plan_bww(s, (int)w.dtype);
      if (!tc_w && wp.ok) {  // 16-bit MFMA path (deform_conv_tc.hip): partial tiles + an ordered sum, no atomics, no zero fill
        rc = dcn_tc_backward_weight<T>(s, wp, w.x_nhwc, offset, mask, gout, w.gwr, gweight, st);
        if (rc) return rc;
        tc_w = true;
      }
    }
    if (!tc_w) {
      { const int zrc = zero_async(w.gwr, (size_t)s.Co * s.Cg * s.K2 * 4, st); if (zrc) return zrc; }
      const int nblk = dcn_num_blocks(s.C, s.Cg, s.cpg, WG_BN);
      const int co_tiles = cdiv(s.Cog, WG_BM);
      constexpr int BK = Mma<T>::BK;
      // split positions so that the launch has >= ~2048 workgroups; chunks are multiples of BK
      long base = (long)nblk * co_tiles * s.K2;
      int nchunks = (int)((2048 + base - 1) / base);
      int pchunk = cdiv(cdiv(s.P, nchunks), BK) * BK;
      if (pchunk < BK) pchunk = BK;
      nchunks = cdiv(s.P, pchunk);
      dim3 grid(nchunks, nblk * co_tiles, s.K2);
      D2_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "deform_conv: too many channel blocks");
      if (vec)
        hipLaunchKernelGGL((dcn_bwd_weight_kernel<T, true>), grid, dim3(WG_THREADS), 0, st, s, (const T*)w.x_nhwc,
                           (const T*)offset, (const T*)mask, (const T*)gout, w.gwr, pchunk, nblk);
      else
        hipLaunchKernelGGL((dcn_bwd_weight_kernel<T, false>), grid, dim3(WG_THREADS), 0, st, s, (const T*)w.x_nhwc,
                           (const T*)offset, (const T*)mask, (const T*)gout, w.gwr, pchunk, nblk);
      D2_LAUNCH_OK();
      const long nw = (long)s.Co * s.Cg * s.K2;
      hipLaunchKernelGGL((unpack_gw_kernel<T>), dim3(cdiv(nw, 256) > 4096 ? 4096 : cdiv(nw, 256)), dim3(256), 0, st,
                         w.gwr, (T*)gweight, s.G, s.Cog, s.Cg, s.K2);
      D2_LAUNCH_OK();
    }
  }
  if (gbias) {
    hipLaunchKernelGGL((dcn_bias_grad_kernel<T>), dim3(s.Co), dim3(256), 0, st, (const T*)gout, (T*)gbias, s.B, s.Co,
                       s.L);
    D2_LAUNCH_OK();
  }
  return D2AMD_OK;
}

}  // namespace d2amd

using namespace d2amd;

extern "C" size_t d2amd_deform_conv_workspace_bytes(const d2amd_dcn_params* p, int backward) {
  DcnShape s;
  if (check_params(p, s, "deform_conv_workspace_bytes")) return 0;
  return carve_ws(s, p->dtype, backward == 1, p->layout == D2AMD_NHWC, nullptr, backward == 2).total + 256;
}

static int dcn_forward_impl(const d2amd_dcn_params* p, const void* x, const void* offset, const void* mask,
                            const void* weight, const void* bias, void* out, void* columns, void* workspace,
                            size_t workspace_bytes, void* stream) {
  DcnShape s;
  int rc = check_params(p, s, "deform_conv_forward");
  if (rc) return rc;
  if (s.B == 0) return D2AMD_OK;
  D2_CHECK_ARG(x && offset && weight && out && workspace, "deform_conv_forward: null pointer");
  D2_CHECK_ARG(p->layout == D2AMD_NCHW || p->layout == D2AMD_NHWC, "deform_conv_forward: bad layout %d", p->layout);
  DcnWs w = carve_ws(s, p->dtype, false, p->layout == D2AMD_NHWC, workspace, columns != nullptr);
  w.col_saved = columns;
  if (workspace_bytes < w.total) {
    set_error("deform_conv_forward: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    return D2AMD_EWORKSPACE;
  }
  return D2_DISPATCH_DTYPE(p->dtype, [&]() -> int {
    return fwd_host<scalar_t>(s, x, offset, mask, weight, bias, out, w, (hipStream_t)stream);
  });
}

extern "C" int d2amd_deform_conv_forward(const d2amd_dcn_params* p, const void* x, const void* offset,
                                         const void* mask, const void* weight, const void* bias, void* out,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  return dcn_forward_impl(p, x, offset, mask, weight, bias, out, nullptr, workspace, workspace_bytes, stream);
}

extern "C" size_t d2amd_deform_conv_columns_bytes(const d2amd_dcn_params* p) {
  DcnShape s;
  if (check_params(p, s, "deform_conv_columns_bytes")) return 0;
  if (s.B == 0) return 0;
  const BwwGemmPlan gp = dcn_bww_gemm_plan(s, p->dtype, p->layout == D2AMD_NHWC);
  if (!gp.ok) return 0;
  // (the GEMM path also leaves the backward's packed weights behind the column: one pack launch per block and iteration)
  const ColPathPlan cp = p->layout == D2AMD_NHWC ? dcn_colpath_plan(s, p->dtype) : ColPathPlan{};
  return cp.ok ? al(gp.col_bytes) + cp.wpack_bytes : gp.col_bytes;
}

extern "C" int d2amd_deform_conv_column_path(const d2amd_dcn_params* p) {
  DcnShape s;
  if (check_params(p, s, "deform_conv_column_path") || s.B == 0) return 0;
  return dcn_colpath_plan(s, p->dtype).ok ? 1 : 0;
}

extern "C" int d2amd_deform_conv_forward_columns(const d2amd_dcn_params* p, const void* x, const void* offset,
                                                 const void* mask, const void* weight, const void* bias, void* out,
                                                 void* columns, void* workspace, size_t workspace_bytes,
                                                 void* stream) {
  D2_CHECK_ARG(columns == nullptr || d2amd_deform_conv_columns_bytes(p) > 0,
               "deform_conv_forward_columns: this shape / dtype keeps no column (d2amd_deform_conv_columns_bytes = 0)");
  return dcn_forward_impl(p, x, offset, mask, weight, bias, out, columns, workspace, workspace_bytes, stream);
}

static int dcn_backward_impl(const d2amd_dcn_params* p, const void* x, const void* offset, const void* mask,
                             const void* weight, const void* grad_out, const void* columns, void* grad_input,
                             void* grad_offset, void* grad_mask, void* grad_weight, void* grad_bias, void* workspace,
                             size_t workspace_bytes, void* stream) {
  DcnShape s;
  int rc = check_params(p, s, "deform_conv_backward");
  if (rc) return rc;
  D2_CHECK_ARG(s.B == 0 || (x && offset && weight && grad_out && workspace), "deform_conv_backward: null pointer");
  D2_CHECK_ARG(p->layout == D2AMD_NCHW || p->layout == D2AMD_NHWC, "deform_conv_backward: bad layout %d", p->layout);
  DcnWs w = carve_ws(s, p->dtype, true, p->layout == D2AMD_NHWC, workspace);
  w.col_saved = const_cast<void*>(columns);
  if (s.B > 0 && workspace_bytes < w.total) {
    set_error("deform_conv_backward: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    return D2AMD_EWORKSPACE;
  }
  return D2_DISPATCH_DTYPE(p->dtype, [&]() -> int {
    return bwd_host<scalar_t>(s, x, offset, mask, weight, grad_out, grad_input, grad_offset, grad_mask, grad_weight,
                              grad_bias, w, (hipStream_t)stream);
  });
}

extern "C" int d2amd_deform_conv_backward(const d2amd_dcn_params* p, const void* x, const void* offset,
                                          const void* mask, const void* weight, const void* grad_out,
                                          void* grad_input, void* grad_offset, void* grad_mask, void* grad_weight,
                                          void* grad_bias, void* workspace, size_t workspace_bytes, void* stream) {
  return dcn_backward_impl(p, x, offset, mask, weight, grad_out, nullptr, grad_input, grad_offset, grad_mask,
                           grad_weight, grad_bias, workspace, workspace_bytes, stream);
}

extern "C" int d2amd_deform_conv_backward_columns(const d2amd_dcn_params* p, const void* x, const void* offset,
                                                  const void* mask, const void* weight, const void* grad_out,
                                                  const void* columns, void* grad_input, void* grad_offset,
                                                  void* grad_mask, void* grad_weight, void* grad_bias, void* workspace,
                                                  size_t workspace_bytes, void* stream) {
  return dcn_backward_impl(p, x, offset, mask, weight, grad_out, columns, grad_input, grad_offset, grad_mask,
                           grad_weight, grad_bias, workspace, workspace_bytes, stream);
}
