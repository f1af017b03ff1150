// Deformable convolution v1 / v2, 16-bit (bf16 / f16) MFMA path for gfx950 -- forward.
// Replaces detectron2/layers/csrc/deformable/deform_conv_cuda.cu:272-440,826-1004 (im2col to an HBM
// column buffer + at::addmm_ per image) for the shapes of BASELINE config 5 (R50 res3-res5 3x3 DCN).
//
// Why this shape of kernel (numbers for res3: C = Co = 128, P = 2*100*168 positions, bf16):
//   * the dense contraction is 9.9 GFLOP = 4 us of MFMA; the deformable gather is 4 corners x 9 taps x
//     C channels per position = 310 MB of L2 reads and ~70 VALU per (position, 8 channels): the
//     kernel is bound by the gather's VALU + L1 rate, not by the matrix pipe.  So the gather is done
//     exactly ONCE per (position, tap, channel) per output-channel tile, with 128-B coalesced
//     segments, and everything else is arranged so that it costs (almost) no issue slots.
//   * both MFMA operands live in LDS in FRAGMENT ORDER ([tile][kstep][lane][16 B]): the weight tile
//     is a straight 16-B copy of a pre-packed global buffer, the gathered column tile is written by
//     the gathering lane into the slot the consuming lane reads; every ds_read_b128 / ds_write_b128
//     is bank-conflict free (33-slot pitch between the k-halves, see slot()).
//   * per-(position, tap) bilinear tables (4 byte offsets + 4 weights with the modulation mask
//     folded in) are built once per workgroup in LDS and shared by all channel chunks.
//   * a stage = (tap, 64 input channels).  Global loads of stage s+1 (16-B gathers + weight copy)
//     are issued before the MFMAs of stage s and land in the other LDS buffer afterwards: one
//     barrier per stage.
//   * blockIdx -> tile map is XCD-aware: each XCD gets one contiguous range of position tiles, so
//     the x pixels its workgroups gather stay in that XCD's 4 MiB L2.
//   * small feature maps (res5: 2,100 positions) do not fill 256 CUs with output tiles alone:
//     the (tap, channel) reduction is split over `ksplit` workgroups writing fp32 partials that a
//     tiny deterministic kernel sums (+ bias, -> 16 bit).
#include "dcn_gemm.h"

#include <stdlib.h>

namespace d2amd {

struct TcEntry {
  uint32_t off[4];  // byte offset of the corner pixel's channel 0 in x (NHWC); 0 when unused
  float w[4];       // bilinear weight x modulation mask; 0 for corners / samples outside the image
};
static_assert(sizeof(TcEntry) == 32, "TcEntry layout");

struct TcArgs {
  const void *x, *offset, *mask, *wp, *bias;
  void* out;
  float* partial;
  int n_pt, n_cot, ksplit, NCH, S, total;  // NCH: channel chunks (16*NKS channels) per conv group
  int ablate;  // profiling only (D2AMD_DCN_ABLATE): 1 no gather loads, 2 no combine, 4 no MFMA, 8 no weight copy
  int out_nhwc;  // 1: out (and the fp32 partials) are [position][Co] (a channels_last caller) instead of [b][Co][l]
  unsigned long long* stamps;  // profiling only (D2AMD_DCN_STAMPS): per workgroup {start, after tables, after loop, end} (100 MHz)
  void* col_out;  // training forward: the gathered column (mask folded in, I/O dtype) is ALSO stored, for the weight
                  // gradient's GEMM -- [position][tap * C + channel]; written by the workgroups of output-channel tile 0 only
};

// Column buffer: per 32-position tile, 2*NKS "subs" (kstep, k-half) of 32 consecutive 16-B slots each, at a
// pitch that spreads the 8 lanes of one ds_write_b128 group over the banks: NKS = 4: a group is one position
// x 8 subs -> odd pitch; NKS = 2: two positions x 4 subs -> pitch = 2 (mod 8).  Reads (32 consecutive slots
// per half-wave) are conflict free for any pitch.
template <int NKS> struct TcB {
  static constexpr int PS = NKS == 4 ? 33 : 34;
  static constexpr int TILE = 2 * NKS * PS;  // 16-B slots per 32-position tile
  __device__ static __forceinline__ int slot(int ntile, int sub, int n32) { return ntile * TILE + sub * PS + n32; }
};

// bilinear table entry of (position p, tap, deformable group), as deform_conv_cuda_kernel.cu:96-130,
// 216-270 (v1) / 665-700, 785-860 (v2): sample inside (-1, H) x (-1, W), corners outside contribute 0
template <typename T>
__device__ __forceinline__ TcEntry tc_make_entry(const DcnShape& s, const T* __restrict__ offset,
                                                 const T* __restrict__ mask, int p, int tap, int dgi) {
  TcEntry e;
#pragma unroll
  for (int t = 0; t < 4; t++) { e.off[t] = 0u; e.w[t] = 0.f; }
  if (p >= s.P) return e;
  const int b = p / s.L, l = p - b * s.L;
  const int ho = l / s.Wo, wo = l - ho * s.Wo;
  const int i = tap / s.kw, j = tap - i * s.kw;
  const long obase = ((long)b * s.DG + dgi) * 2 * s.K2;
  const float off_h = to_f32(offset[(obase + 2 * tap) * s.L + l]);
  const float off_w = to_f32(offset[(obase + 2 * tap + 1) * s.L + l]);
  const float m = mask ? to_f32(mask[(((long)b * s.DG + dgi) * s.K2 + tap) * s.L + l]) : 1.f;
  const float h_im = (float)(ho * s.sh - s.ph + i * s.dh) + off_h;
  const float w_im = (float)(wo * s.sw - s.pw + j * s.dw) + off_w;
  if (!(h_im > -1.f && w_im > -1.f && h_im < (float)s.H && w_im < (float)s.W)) return e;
  const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
  const float hh = 1.f - lh, hw = 1.f - lw;
  const uint32_t pix = (uint32_t)s.C * (uint32_t)sizeof(T);
  const long rowbase = (long)b * s.H;
  if (h_low >= 0 && w_low >= 0) { e.off[0] = (uint32_t)((rowbase + h_low) * s.W + w_low) * pix; e.w[0] = hh * hw * m; }
  if (h_low >= 0 && w_high <= s.W - 1) { e.off[1] = (uint32_t)((rowbase + h_low) * s.W + w_high) * pix; e.w[1] = hh * lw * m; }
  if (h_high <= s.H - 1 && w_low >= 0) { e.off[2] = (uint32_t)((rowbase + h_high) * s.W + w_low) * pix; e.w[2] = lh * hw * m; }
  if (h_high <= s.H - 1 && w_high <= s.W - 1) { e.off[3] = (uint32_t)((rowbase + h_high) * s.W + w_high) * pix; e.w[3] = lh * lw * m; }
  return e;
}

// ---- weight packing: (Co, Cg, K2) -> [g][co tile][stage = (tap, chunk)][m-tile][kstep < NKS][lane][8] ----
// element j of lane l = W[co = cot*BM + mt*32 + (l & 31)][ci = chunk*16*NKS + ks*16 + (l >> 5)*8 + j][tap]
// (the A operand of v_mfma_f32_32x32x16: lane l holds row l & 31, k = 8*(l >> 5) .. +8); rows >= Cog are 0.
template <typename T>
__global__ __launch_bounds__(256) void tc_pack_weight_kernel(const T* __restrict__ w, T* __restrict__ wp, int G,
                                                            int Cog, int Cg, int K2, int BM, int n_cot, int NCH,
                                                            int NKS) {
  const int S = K2 * NCH, MTA = BM / 32;
  const long total = (long)G * n_cot * S * MTA * NKS * 64;
  for (long gi = (long)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (long)gridDim.x * blockDim.x) {
    long r = gi;
    const int lane = (int)(r % 64); r /= 64;
    const int ks = (int)(r % NKS); r /= NKS;
    const int mt = (int)(r % MTA); r /= MTA;
    const int st = (int)(r % S); r /= S;
    const int cot = (int)(r % n_cot); r /= n_cot;
    const int g = (int)r;
    const int tap = st / NCH, cc = st - tap * NCH;
    const int co = cot * BM + mt * 32 + (lane & 31);
    const int ci0 = cc * 16 * NKS + ks * 16 + (lane >> 5) * 8;
    T v[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
      v[j] = co < Cog ? w[(((long)g * Cog + co) * Cg + ci0 + j) * K2 + tap] : from_f32<T>(0.f);
#pragma unroll
    for (int j = 0; j < 8; j++) wp[gi * 8 + j] = v[j];
  }
}

// ---- forward kernel ---------------------------------------------------------------------------------
// Wave-specialised workgroup: NWM x NWN MATRIX waves (wave (wm, wn) owns output rows
// (wm*MT .. +MT) * 32 of the BM = 32*MT*NWM row tile and positions wn*32 .. +32 of the BN = 32*NWN
// position tile; they also copy the weight tile) and NG GATHER waves (bilinear gather of the column
// tile, (BN*8) / (64*NG) items of (position, 8 channels) per lane and stage).  An in-order wave only
// overlaps the matrix pipe with VALU when the two alternate in its own instruction stream, which hipcc
// does not produce for this loop; a matrix wave and a gather wave resident on the same SIMD overlap
// by construction.  One s_barrier per stage orders column/weight buffer hand-off (double buffered).
// (4-wave workgroups with 32-channel stages are planned at THREE per CU -- dcn_tc_plan_fwd -- i.e. <= 168 VGPRs: the
// bound is stated so that an extra live value in the gather path cannot silently cost a third of the occupancy)
template <typename T, int MT, int NWM, int NWN, int NG, int NKS>
__global__ __launch_bounds__(64 * (NWM * NWN + NG), (NKS == 2 && NWM * NWN + NG == 4) ? 3 : 1)
void dcn_fwd_tc_kernel(DcnShape s, TcArgs a) {
  typedef Mma<T> M;
  typedef TcB<NKS> BL;
  constexpr int NMW = NWM * NWN, NT = 64 * (NMW + NG), BM = 32 * MT * NWM, BN = 32 * NWN;
  constexpr int NSUB = 2 * NKS, CHUNK = 16 * NKS;       // (kstep, k-half) subs per stage; channels per stage
  constexpr int ITEMS = (BN * NSUB) / (64 * NG);        // gather items per lane and stage
  constexpr int ACOPY = (BM * NSUB) / (64 * NMW);       // 16-B weight slots per matrix-wave lane and stage
  constexpr int ASLOTS = BM * NSUB, BSLOTS = NWN * BL::TILE;
  static_assert((BN * NSUB) % (64 * NG) == 0 && (BM * NSUB) % (64 * NMW) == 0, "tile / thread mismatch");
  static_assert(NKS == 2 || NKS == 4, "NKS");
  extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // XCD-aware decode: XCD x (= blockIdx % 8) works on one contiguous range of logical tiles
  const int per_xcd = (a.total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= a.total) return;
  const int inner = a.n_cot * s.G * a.ksplit;
  const int pt = logical / inner;
  int rr = logical - pt * inner;
  const int cot = rr % a.n_cot; rr /= a.n_cot;
  const int g = rr % s.G;
  const int kz = rr / s.G;
  const int p0 = pt * BN;
  const int dg_first = (g * s.Cg) / s.cpg, dg_last = ((g + 1) * s.Cg - 1) / s.cpg;
  const int ndg = dg_last - dg_first + 1;
  const int nrows = s.K2 * ndg;

  TcEntry* ent = reinterpret_cast<TcEntry*>(tc_smem);
  raw16* As = reinterpret_cast<raw16*>(tc_smem + (size_t)nrows * BN * sizeof(TcEntry));
  raw16* Bs = As + 2 * ASLOTS;
  unsigned long long* stamp = (a.stamps && tid == 0) ? a.stamps + 4 * (size_t)blockIdx.x : nullptr;
  if (stamp) stamp[0] = wall_clock64();

  {
    const T* offset = (const T*)a.offset;
    const T* mask = (const T*)a.mask;
    for (int e = tid; e < nrows * BN; e += NT) {
      const int row = e / BN, n = e - row * BN;
      const int tap = row / ndg, dgi = dg_first + (row - tap * ndg);
      if (a.ablate & 32) { TcEntry z{}; ent[e] = z; continue; }
      ent[e] = tc_make_entry<T>(s, offset, mask, p0 + n, tap, dgi);
    }
  }
  __syncthreads();  // barrier #0: tables ready
  if (stamp) stamp[1] = wall_clock64();

  const int s_lo = (int)((long)kz * a.S / a.ksplit), s_hi = (int)((long)(kz + 1) * a.S / a.ksplit);
  const int nst = s_hi - s_lo;  // >= 1 (ksplit <= S)

  if (wid >= NMW) {
    // ================================ GATHER waves ================================================
    const int gt = tid - 64 * NMW;  // thread index among the gather waves
    const char* xb = (const char*)a.x;
    raw16 graw[2][ITEMS][4];
    float gw[2][ITEMS][4];
    auto issue = [&](int st, raw16 (&raw)[ITEMS][4], float (&w)[ITEMS][4]) __attribute__((always_inline)) {
      const int tap = st / a.NCH, cc = st - tap * a.NCH;
      const int cabs = g * s.Cg + cc * CHUNK;
      const TcEntry* er = ent + (tap * ndg + (cabs / s.cpg - dg_first)) * BN;
#pragma unroll
      for (int it = 0; it < ITEMS; it++) {
        const int i = it * (64 * NG) + gt;
        const int n = i / NSUB, sub = i % NSUB;
        const raw16 eo = *reinterpret_cast<const raw16*>(&er[n].off[0]);
        const raw16 ew = *reinterpret_cast<const raw16*>(&er[n].w[0]);
        const uint32_t cofs = (uint32_t)(cabs + sub * 8) * (uint32_t)sizeof(T);
#pragma unroll
        for (int c = 0; c < 4; c++) {
          if (!(a.ablate & 1)) raw[it][c] = *reinterpret_cast<const raw16*>(xb + (eo[c] + cofs));
          else raw[it][c] = eo;
          w[it][c] = __uint_as_float(ew[c]);
        }
      }
    };
    char* const col_dst = (a.col_out && cot == 0) ? (char*)a.col_out : nullptr;  // uniform
    int cst = 0;  // the stage (tap * NCH + chunk) of the tile being combined: set by the caller of combine()
    auto combine = [&](int buf, const raw16 (&raw)[ITEMS][4], const float (&w)[ITEMS][4]) __attribute__((always_inline)) {
      raw16* Bb = Bs + buf * BSLOTS;
      if (a.ablate & 2) {
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
          const int i = it * (64 * NG) + gt;
          Bb[BL::slot(i / (32 * NSUB), i % NSUB, (i / NSUB) & 31)] = raw[it][0] ^ raw[it][1] ^ raw[it][2] ^ raw[it][3];
        }
        return;
      }
#pragma unroll
      for (int it = 0; it < ITEMS; it++) {
        const int i = it * (64 * NG) + gt;
        const int n = i / NSUB, sub = i % NSUB;
        float v[8];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          float f[8];
          tc_unpack(raw[it][c], f, T{});
          const float wc = w[it][c];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = c == 0 ? wc * f[u] : v[u] + wc * f[u];
        }
        const raw16 packed = tc_pack(v, T{});
        Bb[BL::slot(n >> 5, sub, n & 31)] = packed;
        // (row-major column [position][tap * C + channel], the layout of the GEMM path -- dcn_colpath.hip; a position's
        // NSUB lanes store CHUNK * 2 consecutive bytes)
        if (col_dst && p0 + n < s.P)
          *reinterpret_cast<raw16*>(col_dst + (((size_t)(p0 + n) * s.K2 * s.C + (size_t)cst * CHUNK) * sizeof(T)) + (size_t)sub * 16) = packed;
      }
    };
    // the gathers run TWO stages ahead of the matrix waves (register double buffer), so the L2
    // latency of stage j+2 hides behind the combine of stage j+1
    issue(s_lo, graw[0], gw[0]);
    if (nst > 1) issue(s_lo + 1, graw[1], gw[1]);
    cst = s_lo;
    combine(0, graw[0], gw[0]);
    if (nst > 2) issue(s_lo + 2, graw[0], gw[0]);
    __syncthreads();  // barrier #1: stage 0 staged
    // iteration j (matrix waves compute stage j): stage j+1 -> buffer (j+1)&1; registers (j+1)&1
    for (int j = 0; j < nst; j += 2) {
      if (j + 1 < nst) {
        cst = s_lo + j + 1;
        combine(1, graw[1], gw[1]);
        if (j + 3 < nst) issue(s_lo + j + 3, graw[1], gw[1]);
      }
      __syncthreads();
      if (j + 1 < nst) {
        if (j + 2 < nst) {
          cst = s_lo + j + 2;
          combine(0, graw[0], gw[0]);
          if (j + 4 < nst) issue(s_lo + j + 4, graw[0], gw[0]);
        }
        __syncthreads();
      }
    }
    return;
  }

  // ================================== MATRIX waves ==================================================
  const int wm = wid % NWM, wn = wid / NWM;
  const raw16* wsrc = (const raw16*)a.wp + ((size_t)(g * a.n_cot + cot) * a.S) * ASLOTS;
  f32x16_t acc[MT];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
  raw16 araw[ACOPY];
  auto a_issue = [&](int st) __attribute__((always_inline)) {
    if (a.ablate & 8) return;
    const raw16* src = wsrc + (size_t)st * ASLOTS;
#pragma unroll
    for (int q = 0; q < ACOPY; q++) araw[q] = src[q * (64 * NMW) + tid];
  };
  auto a_store = [&](int buf) __attribute__((always_inline)) {
    raw16* Ab = As + buf * ASLOTS;
#pragma unroll
    for (int q = 0; q < ACOPY; q++) Ab[q * (64 * NMW) + tid] = araw[q];
  };
  auto mfma_stage = [&](int buf) __attribute__((always_inline)) {
    if (a.ablate & 4) return;
    const raw16* Ab = As + buf * ASLOTS;
    const raw16* Bb = Bs + buf * BSLOTS;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      const typename M::frag b = __builtin_bit_cast(typename M::frag, Bb[BL::slot(wn, 2 * ks + (lane >> 5), lane & 31)]);
#pragma unroll
      for (int m = 0; m < MT; m++) {
        const typename M::frag av =
            __builtin_bit_cast(typename M::frag, Ab[((wm * MT + m) * NKS + ks) * 64 + lane]);
        acc[m] = M::mma(av, b, acc[m]);
      }
    }
  };
  a_issue(s_lo);
  a_store(0);
  __syncthreads();  // barrier #1
  for (int j = 0; j < nst; j++) {
    const int cur = j & 1;
    const bool more = j + 1 < nst;  // uniform
    if (more) a_issue(s_lo + j + 1);
    mfma_stage(cur);
    if (more) a_store(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: out[b][g*Cog + co][l] (+ bias), or fp32 partials when the reduction is split
  if (stamp) stamp[2] = wall_clock64();
  const int p = p0 + wn * 32 + (lane & 31);
  const T* bias = (const T*)a.bias;
  const bool add_bias = bias != nullptr && a.ksplit == 1;  // uniform
  if (add_bias) {  // all bias loads in flight together (clamped rows), then one pass of stores
#pragma unroll
    for (int m = 0; m < MT; m++) {
      float bv[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int co = min(cot * BM + (wm * MT + m) * 32 + frag_row(r, lane), s.Cog - 1);
        bv[r] = to_f32(bias[g * s.Cog + co]);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][r] += bv[r];
    }
  }
  if (p < s.P && !(a.ablate & 16)) {
    const int b = p / s.L, l = p - b * s.L;
    // NCHW: element (co, position) at (b Co + co) L + l; NHWC (a.out_nhwc): at position * Co + co -- a lane's four
    // consecutive accumulator rows are four consecutive channels there
    const long obase = a.out_nhwc ? (long)p * s.Co + (long)g * s.Cog : ((long)b * s.Co + (long)g * s.Cog) * s.L + l;
    const long cstride = a.out_nhwc ? 1 : s.L;
    T* outp = (T*)a.out;
    float* part = a.partial + (a.ksplit > 1 ? (long)kz * s.B * s.Co * s.L : 0l);
    if (a.out_nhwc && a.ksplit == 1 && (s.Cog & 3) == 0 && (s.Co & 3) == 0) {
      // channels_last output: accumulator rows 4 q .. 4 q + 3 of a lane are 4 consecutive channels of its position
#pragma unroll
      for (int m = 0; m < MT; m++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int co = cot * BM + (wm * MT + m) * 32 + frag_row(4 * q, lane);
          if (co < s.Cog) {  // (Cog % 4 == 0: the four rows are in range together)
            const float v4[4] = {acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]};
            vec4<T> pk;
            pack4(v4, pk);
            *reinterpret_cast<vec4<T>*>(outp + obase + co) = pk;
          }
        }
      }
    } else {
#pragma unroll
    for (int m = 0; m < MT; m++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int co = cot * BM + (wm * MT + m) * 32 + frag_row(r, lane);
        if (co < s.Cog) {
          const long o = obase + (long)co * cstride;
          if (a.ksplit == 1) outp[o] = from_f32<T>(acc[m][r]);
          else part[o] = acc[m][r];
        }
      }
    }
    }
  }
  if (stamp) stamp[3] = wall_clock64();
}

// ---- forward kernel, autonomous-wave form -------------------------------------------------------------
// One WAVE = one (32*MT output channels) x (32 positions) tile x one share of the (tap, channel) reduction,
// with NO workgroup barrier and NO LDS: lane (n, half) gathers the 8 channels k = 8*half.. of its own
// position n straight into the B-operand registers of v_mfma_f32_32x32x16 (the fragment layout of that
// instruction IS "one position, 8 consecutive k per lane"), and reads the A operand (weights, pre-packed in
// fragment order, 1 KiB coalesced per wave instruction) from L1/L2.  Measured motivation (profiles/r01,
// dcn_fwd timeline): the LDS-staged workgroup kernel above spends 2/3 of its time in per-stage barrier /
// LDS hand-off latency chains and in the tail of unevenly filled CUs, not on the matrix pipe, the VALU or
// L1 bandwidth.  Autonomous waves balance at wave granularity (1,050 waves on 1,024 SIMDs for res3) and
// keep two stages of operands in flight in registers.
template <typename T, int MT, int NKS>
__global__ __launch_bounds__(64, 2) void dcn_fwd_wave_kernel(DcnShape s, TcArgs a) {
#ifdef D2AMD_WAVE_ABLATE
  constexpr int WAB = D2AMD_WAVE_ABLATE;
#else
  constexpr int WAB = 0;
#endif
  typedef Mma<T> M;
  constexpr int CHUNK = 16 * NKS, ASLOTS = 32 * MT * 2 * NKS;  // channels per stage; 16-B weight slots per stage
  const int lane = threadIdx.x, n32 = lane & 31, half = lane >> 5;
  const int per_xcd = (a.total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= a.total) return;
  const int inner = a.n_cot * s.G * a.ksplit;
  const int pt = logical / inner;
  int rr = logical - pt * inner;
  const int cot = rr % a.n_cot; rr /= a.n_cot;
  const int g = rr % s.G;
  const int kz = rr / s.G;
  const int p = pt * 32 + n32;
  const bool pvalid = p < s.P;
  const int pc = pvalid ? p : s.P - 1;
  const int b = pc / s.L, l = pc - b * s.L;
  const int ho = l / s.Wo, wo = l - ho * s.Wo;
  const T* offset = (const T*)a.offset;
  const T* mask = (const T*)a.mask;
  const char* xb = (const char*)a.x;
  const raw16* wsrc = (const raw16*)a.wp + ((size_t)(g * a.n_cot + cot) * a.S) * ASLOTS + lane;
  const int s_lo = (int)((long)kz * a.S / a.ksplit), s_hi = (int)((long)(kz + 1) * a.S / a.ksplit);
  const uint32_t pix = (uint32_t)s.C * (uint32_t)sizeof(T);

  // raw (offset_h, offset_w, mask) of (tap, deformable group) for this lane's position
  auto tap_raw = [&](int tap, int dgi, float (&r)[3]) __attribute__((always_inline)) {
    const long obase = ((long)b * s.DG + dgi) * 2 * s.K2;
    r[0] = to_f32(offset[(obase + 2 * tap) * s.L + l]);
    r[1] = to_f32(offset[(obase + 2 * tap + 1) * s.L + l]);
    r[2] = mask ? to_f32(mask[(((long)b * s.DG + dgi) * s.K2 + tap) * s.L + l]) : 1.f;
  };
  // bilinear table of the current tap: byte offsets of the 4 corners + weights (x mask); see tc_make_entry
  uint32_t eoff[4];
  float ew[4];
  auto make_entry = [&](int tap, const float (&r)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; t++) { eoff[t] = 0u; ew[t] = 0.f; }
    const int i = tap / s.kw, j = tap - i * s.kw;
    const float h_im = (float)(ho * s.sh - s.ph + i * s.dh) + r[0];
    const float w_im = (float)(wo * s.sw - s.pw + j * s.dw) + r[1];
    if (pvalid && h_im > -1.f && w_im > -1.f && h_im < (float)s.H && w_im < (float)s.W) {
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const int h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
      const float hh = 1.f - lh, hw = 1.f - lw, m = r[2];
      const long rowbase = (long)b * s.H;
      if (h_low >= 0 && w_low >= 0) { eoff[0] = (uint32_t)((rowbase + h_low) * s.W + w_low) * pix; ew[0] = hh * hw * m; }
      if (h_low >= 0 && w_high <= s.W - 1) { eoff[1] = (uint32_t)((rowbase + h_low) * s.W + w_high) * pix; ew[1] = hh * lw * m; }
      if (h_high <= s.H - 1 && w_low >= 0) { eoff[2] = (uint32_t)((rowbase + h_high) * s.W + w_low) * pix; ew[2] = lh * hw * m; }
      if (h_high <= s.H - 1 && w_high <= s.W - 1) { eoff[3] = (uint32_t)((rowbase + h_high) * s.W + w_high) * pix; ew[3] = lh * lw * m; }
    }
  };

  f32x16_t acc[MT];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[m][r] = 0.f;

  // The stages of one bilinear TABLE (tap, deformable group) are consecutive: SPT = min(cpg, Cg) / CHUNK
  // of them (even, because both are multiples of 64).  The reduction share of this wave is a range of
  // whole tables.  All loads sit in straight-line code (clamped indices instead of branches) so that the
  // compiler's s_waitcnt counting stays exact and the stage-ahead prefetch really is asynchronous: with
  // loads inside data-dependent branches hipcc falls back to vmcnt(0) before every stage.
  const int dg_first = (g * s.Cg) / s.cpg, dg_last = ((g + 1) * s.Cg - 1) / s.cpg;
  const int ndg = dg_last - dg_first + 1;
  const int SPT = a.NCH / ndg;              // stages per table
  const int ntab = s.K2 * ndg;
  const int tb_lo = (int)((long)kz * ntab / a.ksplit), tb_hi = (int)((long)(kz + 1) * ntab / a.ksplit);

  struct Tab { uint32_t off[4]; float w[4]; };
  auto load_raw = [&](int tb, float (&r)[3]) __attribute__((always_inline)) {
    tb = min(tb, ntab - 1);
    const int tap = tb / ndg, dgi = dg_first + (tb - tap * ndg);
    tap_raw(tap, dgi, r);
  };
  auto build = [&](int tb, const float (&r)[3], Tab& t) __attribute__((always_inline)) {
    tb = min(tb, ntab - 1);
    make_entry(tb / ndg, r);
#pragma unroll
    for (int c = 0; c < 4; c++) { t.off[c] = eoff[c]; t.w[c] = ew[c]; }
  };
  raw16 graw[2][NKS][4], araw[2][MT][NKS];
  float gw[2][4];
  // loads of stage (table tb, chunk q) with table registers `t`
  auto issue = [&](int tb, int q, const Tab& t, raw16 (&gr)[NKS][4], raw16 (&ar)[MT][NKS], float (&w)[4]) __attribute__((always_inline)) {
    const int tap = tb / ndg, dl = tb - tap * ndg;
    const int cc = dl * SPT + q;                      // chunk index inside the conv group
    const int st = tap * a.NCH + cc;
    const uint32_t cofs = (uint32_t)(g * s.Cg + cc * CHUNK + half * 8) * (uint32_t)sizeof(T);
    if (!(WAB & 1)) {
#pragma unroll
      for (int ks = 0; ks < NKS; ks++)
#pragma unroll
        for (int c = 0; c < 4; c++)
          gr[ks][c] = *reinterpret_cast<const raw16*>(xb + (t.off[c] + cofs + (uint32_t)(ks * 16 * sizeof(T))));
    }
#pragma unroll
    for (int c = 0; c < 4; c++) w[c] = t.w[c];
    const raw16* src = wsrc + (size_t)st * ASLOTS;
    if (!(WAB & 8)) {
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) ar[m][ks] = src[(m * NKS + ks) * 64];
    }
  };
  auto compute = [&](const raw16 (&gr)[NKS][4], const raw16 (&ar)[MT][NKS], const float (&w)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      raw16 bp;
      if (WAB & 2) {
        bp = gr[ks][0] ^ gr[ks][1] ^ gr[ks][2] ^ gr[ks][3];
      } else {
        float v[8];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          float f[8];
          tc_unpack(gr[ks][c], f, T{});
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = c == 0 ? w[c] * f[u] : v[u] + w[c] * f[u];
        }
        bp = tc_pack(v, T{});
      }
      const typename M::frag bq = __builtin_bit_cast(typename M::frag, bp);
      if (WAB & 4) {
#pragma unroll
        for (int m = 0; m < MT; m++) acc[m][0] += __uint_as_float(bp.x ^ ar[m][ks].x);
      } else {
#pragma unroll
        for (int m = 0; m < MT; m++) acc[m] = M::mma(__builtin_bit_cast(typename M::frag, ar[m][ks]), bq, acc[m]);
      }
    }
  };
  auto select = [&](bool first, const Tab& x, const Tab& y, Tab& o) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; c++) { o.off[c] = first ? x.off[c] : y.off[c]; o.w[c] = first ? x.w[c] : y.w[c]; }
  };

  if (tb_lo < tb_hi) {
    Tab cur, nxt;
    float r0[3], r1[3], rn[3];
    load_raw(tb_lo, r0);
    load_raw(tb_lo + 1, r1);
    load_raw(tb_lo + 2, rn);
    build(tb_lo, r0, cur);
    build(tb_lo + 1, r1, nxt);
    issue(tb_lo, 0, cur, graw[0], araw[0], gw[0]);
    for (int tb = tb_lo; tb < tb_hi; tb++) {
      const int tbn = min(tb + 1, tb_hi - 1);  // table of the stage after this table's last one (clamped)
      for (int q = 0; q < SPT; q += 2) {
        // stage q + 1 is always in this table (SPT is even)
        issue(tb, q + 1, cur, graw[1], araw[1], gw[1]);
        compute(graw[0], araw[0], gw[0]);
        // stage q + 2: this table, or chunk 0 of the next one (the very last iteration re-issues a valid
        // stage of the last table; its operands are never consumed)
        const bool same = q + 2 < SPT;  // uniform
        Tab t;
        select(same, cur, nxt, t);
        issue(same ? tb : tbn, same ? q + 2 : 0, t, graw[0], araw[0], gw[0]);
        compute(graw[1], araw[1], gw[1]);
      }
      cur = nxt;
      build(tb + 2, rn, nxt);     // table tb + 2 (clamped), from the values loaded two tables ago
      load_raw(tb + 3, rn);
    }
  }

  // ---- epilogue: out[b][g*Cog + co][l] (+ bias), or fp32 partials when the reduction is split
  const T* bias = (const T*)a.bias;
  const bool add_bias = bias != nullptr && a.ksplit == 1;  // uniform
  constexpr int BM = 32 * MT;
  if (add_bias) {
#pragma unroll
    for (int m = 0; m < MT; m++) {
      float bv[16];
#pragma unroll
      for (int r = 0; r < 16; r++) bv[r] = to_f32(bias[g * s.Cog + min(cot * BM + m * 32 + frag_row(r, lane), s.Cog - 1)]);
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][r] += bv[r];
    }
  }
  if (pvalid && !(WAB & 16)) {
    const long obase = a.out_nhwc ? ((long)b * s.L + l) * s.Co + (long)g * s.Cog
                                  : ((long)b * s.Co + (long)g * s.Cog) * s.L + l;
    const long cstride = a.out_nhwc ? 1 : s.L;
    T* outp = (T*)a.out;
    float* part = a.partial + (a.ksplit > 1 ? (long)kz * s.B * s.Co * s.L : 0l);
#pragma unroll
    for (int m = 0; m < MT; m++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int co = cot * BM + m * 32 + frag_row(r, lane);
        if (co < s.Cog) {
          const long o = obase + (long)co * cstride;
          if (a.ksplit == 1) outp[o] = from_f32<T>(acc[m][r]);
          else part[o] = acc[m][r];
        }
      }
    }
  }
}

// out[i] = sum_kz partial[kz][i] (+ bias[co]) in a fixed order
template <typename T>
__global__ __launch_bounds__(256) void tc_reduce_partial_kernel(const float* __restrict__ partial, const T* __restrict__ bias,
                                                               T* __restrict__ out, long n, int ksplit, int Co, int L,
                                                               int nhwc) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int k = 0; k < ksplit; k++) v += partial[(long)k * n + i];
    if (bias) v += to_f32(bias[nhwc ? i % Co : (i / L) % Co]);
    out[i] = from_f32<T>(v);
  }
}

// ---- host side --------------------------------------------------------------------------------------
static int tc_env_cfg(int (&v)[6]) {
  const char* e = getenv("D2AMD_DCN_CFG");  // "MT,NWM,NWN,KSPLIT[,NKS[,WAVE]]" (profiling switch)
  if (!e) return 0;
  v[4] = 0; v[5] = -1;
  return sscanf(e, "%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) >= 4;
}

TcPlan dcn_tc_plan_fwd(const DcnShape& s, int dtype) {
  TcPlan pl{};
  pl.ok = false;
  if (getenv("D2AMD_DCN_V1")) return pl;
  if (dtype != D2AMD_BF16 && dtype != D2AMD_F16) return pl;
  if (s.Cg % 64 != 0 || s.cpg % 64 != 0 || s.P <= 0) return pl;
  if ((long)s.B * s.H * s.W * s.C * 2 >= (1l << 32)) return pl;   // 32-bit byte offsets into x
  int ndg_max = 1;
  for (int g = 0; g < s.G; g++) {
    const int n = ((g + 1) * s.Cg - 1) / s.cpg - (g * s.Cg) / s.cpg + 1;
    ndg_max = n > ndg_max ? n : ndg_max;
  }
  if (s.K2 * ndg_max > 32) return pl;
  pl.ndg = ndg_max;
  // Tile shape (measured, profiles/r01/dcn_fwd_sweep.txt): 128 output channels x 64 positions per workgroup
  // (2 matrix + 2 gather waves), 32-channel stages: 43 KB of LDS and <= 168 VGPRs keep three workgroups
  // per CU resident, which hides the per-stage barrier / LDS hand-off latency and lets every workgroup
  // of the launch start at once (a second round of workgroups costs a full workgroup lifetime).  Small
  // maps split the (tap, channel) reduction until there are ~2 workgroups per CU.
  int MT = s.Cog <= 64 ? 2 : 4, NWM = 1, NWN = 2, ks = 1, NKS = 2, wave = 0;
  // r04 experiment, MEASURED SLOWER and off (profiles/r04/dcn_fwd_tiles_ab.txt): 256 output channels per workgroup when
  // the layer has them, so that the column gather is repeated Co / 256 instead of Co / 128 times (res4 2 -> 1, res5 4 ->
  // 2), with 32-position tiles on small maps (mode 2) or a deeper reduction split (mode 1).  res4 forward 71 -> 95 /
  // 100 us per block, res5 76 -> 99 / 103: the kernel is bound by how many GATHER WAVES are in flight per CU (their
  // chains of L2 round trips), not by the number of gathered elements -- halving the workgroups halves those waves.
  // D2AMD_DCN_FWD_BIG = 1 / 2 selects the variants.
  {
    const char* e = d2_prof_env("D2AMD_DCN_FWD_BIG");
    const int mode = e ? atoi(e) : 0;
    if (s.Cog >= 256 && mode >= 1) {
      NWM = 2;
      if (mode >= 2 && (long)cdiv(s.P, 64) * cdiv(s.Cog, 256) * s.G < 480) NWN = 1;
    }
  }
  auto wgs = [&](int nwn, int k) { return (long)cdiv(s.P, 32 * nwn) * cdiv(s.Cog, 32 * MT * NWM) * s.G * k; };
  {
    const int stages = s.K2 * (s.Cg / 32);
    while (ks < 8 && wgs(NWN, ks) < 480 && stages / (ks + 1) >= 8) ks++;
  }
  int ev[6];
  if (tc_env_cfg(ev)) {
    MT = ev[0]; NWM = ev[1]; NWN = ev[2]; ks = ev[3];
    NKS = (ev[4] == 2 || ev[4] == 4) ? ev[4] : 4;
    wave = ev[5] == 1;
  }
  if (wave) {
    if (NWM != 1 || NWN != 1 || NKS != 2 || (MT != 4 && MT != 2)) return pl;  // no such instantiation
  }
  pl.wave = wave;
  pl.NKS = NKS;
  pl.NCH = s.Cg / (16 * NKS);
  pl.S = s.K2 * pl.NCH;
  if (ks < 1 || ks > pl.S) ks = 1;
  pl.MT = MT; pl.NWM = NWM; pl.NWN = NWN; pl.ksplit = ks;
  pl.BM = 32 * MT * NWM; pl.BN = 32 * NWN;
  pl.n_cot = cdiv(s.Cog, pl.BM);
  pl.n_pt = cdiv(s.P, pl.BN);
  pl.wp_bytes = (size_t)s.G * pl.n_cot * pl.S * pl.BM * 16 * NKS * 2;
  pl.partial_bytes = ks > 1 ? (size_t)ks * s.B * s.Co * s.L * 4 : 0;
  const size_t btile = NKS == 4 ? TcB<4>::TILE : TcB<2>::TILE;
  pl.lds = (size_t)s.K2 * pl.ndg * pl.BN * sizeof(TcEntry) + 2 * (size_t)pl.BM * 2 * NKS * 16 +
      2 * (size_t)pl.NWN * btile * 16;
  if (wave) pl.lds = 0;
  pl.ok = pl.lds <= 160 * 1024;
  return pl;
}

template <typename T, int MT, int NWM, int NWN, int NKS>
static int tc_launch_fwd2(const DcnShape& s, const TcPlan& pl, const TcArgs& a, hipStream_t st) {
  auto kern = dcn_fwd_tc_kernel<T, MT, NWM, NWN, NWN, NKS>;  // one gather wave per 32 positions
  if (pl.lds > 64 * 1024)
    D2_HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
  const int grid = (a.total + 7) / 8 * 8;
  const char* sp = d2_prof_env("D2AMD_DCN_STAMPS");  // profiling only: dump per-workgroup timestamps to this file
  TcArgs a2 = a;
  if (sp) {
    D2_HIP_OK(hipMalloc(&a2.stamps, (size_t)grid * 4 * 8));
    { const int zrc = zero_async(a2.stamps, (size_t)grid * 4 * 8, st); if (zrc) return zrc; }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (NWM * NWN + NWN)), pl.lds, st, s, a2);
  D2_LAUNCH_OK();
  if (sp) {
    D2_HIP_OK(hipStreamSynchronize(st));
    unsigned long long* h = (unsigned long long*)malloc((size_t)grid * 4 * 8);
    D2_HIP_OK(hipMemcpy(h, a2.stamps, (size_t)grid * 4 * 8, hipMemcpyDeviceToHost));
    FILE* f = fopen(sp, "w");
    if (f) {
      for (int i = 0; i < grid; i++) fprintf(f, "%d %llu %llu %llu %llu\n", i, h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
      fclose(f);
    }
    free(h);
    (void)hipFree(a2.stamps);
  }
  return D2AMD_OK;
}

template <typename T, int MT, int NWM, int NWN>
static int tc_launch_fwd(const DcnShape& s, const TcPlan& pl, const TcArgs& a, hipStream_t st) {
  return pl.NKS == 2 ? tc_launch_fwd2<T, MT, NWM, NWN, 2>(s, pl, a, st) : tc_launch_fwd2<T, MT, NWM, NWN, 4>(s, pl, a, st);
}

template <typename T>
int dcn_tc_forward(const DcnShape& s, const TcPlan& pl, const void* x_nhwc, const void* offset, const void* mask,
                   const void* weight, const void* bias, void* out, void* wp, float* partial, hipStream_t st,
                   bool out_nhwc, void* col_out) {
  {
    const long groups16 = (long)s.G * pl.n_cot * pl.S * (pl.BM / 32) * pl.NKS * 64;
    const int blocks = cdiv(groups16, 256) > 8192 ? 8192 : cdiv(groups16, 256);
    hipLaunchKernelGGL((tc_pack_weight_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)weight, (T*)wp, s.G,
                       s.Cog, s.Cg, s.K2, pl.BM, pl.n_cot, pl.NCH, pl.NKS);
    D2_LAUNCH_OK();
  }
  TcArgs a{};
  a.x = x_nhwc; a.offset = offset; a.mask = mask; a.wp = wp; a.bias = bias; a.out = out; a.partial = partial;
  a.n_pt = pl.n_pt; a.n_cot = pl.n_cot; a.ksplit = pl.ksplit; a.NCH = pl.NCH; a.S = pl.S;
  { const char* e = d2_prof_env("D2AMD_DCN_ABLATE"); a.ablate = e ? atoi(e) : 0; }
  a.out_nhwc = out_nhwc ? 1 : 0;
  a.col_out = (col_out && !pl.wave && pl.NKS == 2 && s.G == 1 && s.DG == 1) ? col_out : nullptr;  // (dcn_bww_gemm_plan)
  const long total = (long)pl.n_pt * pl.n_cot * s.G * pl.ksplit;
  D2_CHECK_ARG(total < (1l << 30), "deform_conv: too many tiles");
  a.total = (int)total;
  int rc = D2AMD_EUNSUPPORTED;
  const int key = pl.wave ? -(pl.MT * 10 + pl.NKS) : pl.MT * 100 + pl.NWM * 10 + pl.NWN;
  const int grid = (a.total + 7) / 8 * 8;
  const bool timed = timing_begin("dcn_fwd", st);
  switch (key) {
    case -42: hipLaunchKernelGGL((dcn_fwd_wave_kernel<T, 4, 2>), dim3(grid), dim3(64), 0, st, s, a); rc = 0; break;
    case -22: hipLaunchKernelGGL((dcn_fwd_wave_kernel<T, 2, 2>), dim3(grid), dim3(64), 0, st, s, a); rc = 0; break;
    case 414: rc = tc_launch_fwd<T, 4, 1, 4>(s, pl, a, st); break;
    case 412: rc = tc_launch_fwd<T, 4, 1, 2>(s, pl, a, st); break;
    case 411: rc = tc_launch_fwd<T, 4, 1, 1>(s, pl, a, st); break;
    case 422: rc = tc_launch_fwd<T, 4, 2, 2>(s, pl, a, st); break;
    case 421: rc = tc_launch_fwd<T, 4, 2, 1>(s, pl, a, st); break;
    case 222: rc = tc_launch_fwd<T, 2, 2, 2>(s, pl, a, st); break;
    case 214: rc = tc_launch_fwd<T, 2, 1, 4>(s, pl, a, st); break;
    case 212: rc = tc_launch_fwd<T, 2, 1, 2>(s, pl, a, st); break;
    case 211: rc = tc_launch_fwd<T, 2, 1, 1>(s, pl, a, st); break;
    default: set_error("deform_conv: no kernel for tile config MT=%d NWM=%d NWN=%d", pl.MT, pl.NWM, pl.NWN);
  }
  if (timed) timing_end("dcn_fwd", st);
  if (rc) return rc;
  D2_LAUNCH_OK();
  if (pl.ksplit > 1) {
    const long n = (long)s.B * s.Co * s.L;
    const int blocks = cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256);
    hipLaunchKernelGGL((tc_reduce_partial_kernel<T>), dim3(blocks), dim3(256), 0, st, (const float*)partial,
                       (const T*)bias, (T*)out, n, pl.ksplit, s.Co, s.L, out_nhwc ? 1 : 0);
    D2_LAUNCH_OK();
  }
  return D2AMD_OK;
}

template int dcn_tc_forward<bf16_t>(const DcnShape&, const TcPlan&, const void*, const void*, const void*, const void*,
                                    const void*, void*, void*, float*, hipStream_t, bool, void*);
template int dcn_tc_forward<f16_t>(const DcnShape&, const TcPlan&, const void*, const void*, const void*, const void*,
                                   const void*, void*, void*, float*, hipStream_t, bool, void*);

// =====================================================================================================
// Backward w.r.t. input / offset / mask, 16-bit path.
// Replaces deform_conv_cuda.cu:442-640,1006-1221 with deform_conv_cuda_kernel.cu:272-452,862-1066
// (dcol = W^T dY into an HBM column buffer, then col2im with one global atomic per (sample corner,
// channel) and a col2im_coord kernel that re-reads the column buffer).
//
// Workgroup = one 8x8 tile of output positions x one deformable group; stage = (64-channel chunk, tap).
// Per stage: dcol[64 ch][64 pos] = W^T[tap, chunk] dY by MFMA (operand fragments straight from L2:
// weights pre-packed in fragment order, dY as [position][Co]; K = Cog split over two wave halves that
// meet in LDS), then
//   phase A  lane = (position, 8 channels): re-gather the 4 corners of x, form the bilinear value and its
//            coordinate derivatives, reduce d(offset) / d(mask) over channels (wave shuffles; one owner
//            thread per (tap, position) accumulates in LDS -> plain stores at the end, no atomics);
//   phase B  dX: one wave instruction = one (position, corner) pair x 64 channels = ONE 256-B contiguous
//            global fp32 atomic; the pair's weight / target are wave-uniform (SGPRs), pairs with zero
//            weight are skipped by a scalar branch.
// Measured on MI355X (profiles/r01/dcn_*): device-scope fp32 atomics retire ~0.75 lane-ops/clk/CU when
// every instruction covers whole 128-B lines (4x the first version's strided pattern).  Accumulating in
// an LDS-resident patch first was tried and is SLOWER: ds_add_f32 retires ~0.2 lane-ops/clk/CU.
struct BwEntry {
  uint32_t pix[4];  // pixel index (b*H + y)*W + x of the corner; 0 when unused
  float w[4];       // bilinear corner weights (mask NOT folded in); 0 for corners outside
  float lh, lw, m;
  uint32_t flags;   // bit c: corner c inside the image; bit 4: sample inside (-1,H)x(-1,W) and position valid
};
static_assert(sizeof(BwEntry) == 48, "BwEntry layout");
struct BwPair { float wgt; uint32_t eofs; };  // weight * mask of a (position, corner); element offset pix * C

struct BwArgs {
  const void *x, *offset, *mask, *wp, *gout;  // x NHWC, gout NHWC [P][Co], wp = tc_pack_weight_t layout
  float *gx, *goff, *gmask;                   // gx fp32 NHWC (zero-filled), goff / gmask fp32 NCHW-like
  void* dcol;  // column-gather path: dcol[(position * K2 + tap) * C + channel] in the I/O dtype (then gx is null)
  int tiles_y, tiles_x, total, csplit;  // csplit: workgroups per (tile, deformable group), each a share of the channel chunks
  int ablate;  // profiling only: 1 no MFMA, 2 no phase A, 4 no phase B, 16 phase B without the atomics
};

// weight (Co, Cg, K2) -> [g][tap][c64][mt(2)][ks][lane][8]: element j of lane l =
// W[co = ks*16 + (l >> 5)*8 + j][ci = c64*64 + mt*32 + (l & 31)][tap]  (A operand rows = input channels, k = co)
template <typename T>
__global__ __launch_bounds__(256) void tc_pack_weight_t_kernel(const T* __restrict__ w, T* __restrict__ wp, int G,
                                                              int Cog, int Cg, int K2) {
  const int NC64 = Cg / 64, KS = Cog / 16;
  const long total = (long)G * K2 * NC64 * 2 * KS * 64;
  for (long gi = (long)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (long)gridDim.x * blockDim.x) {
    long r = gi;
    const int lane = (int)(r % 64); r /= 64;
    const int ks = (int)(r % KS); r /= KS;
    const int mt = (int)(r % 2); r /= 2;
    const int c64 = (int)(r % NC64); r /= NC64;
    const int tap = (int)(r % K2); r /= K2;
    const int g = (int)r;
    const int ci = c64 * 64 + mt * 32 + (lane & 31);
    const int co0 = ks * 16 + (lane >> 5) * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) wp[gi * 8 + j] = w[(((long)g * Cog + co0 + j) * Cg + ci) * K2 + tap];
  }
}

constexpr int BW_CPITCH = 68;  // floats per position row of the dcol tile (16-B aligned rows, spread banks)

// r04 -- a stage used to be a chain of exposed latencies (5 us per (64-channel chunk, tap) stage, 18 stages per res3
// workgroup, two workgroups per CU): offset / mask loads -> table -> MFMA operand loads -> MFMAs -> LDS meet -> corner
// gathers of phase A -> VALU -> barrier.  Now (a) the tables of ALL taps are built once per workgroup into LDS (they do
// not depend on the channel chunk: the offset / mask round trip left the stage loop); (b) the 4 corner gathers of phase A
// are issued at the TOP of the stage from the thread's own table entry (thread = (position, corner) in both places), so
// they fly under the MFMA section; (c) KHT > 0 (one conv group, Cog / 32 = KHT in {4, 8}): the wave's dY fragments --
// the same for every stage of the workgroup -- are loaded once and stay in registers.
// DCOL: the column-gather mode (a.dcol set, a.gx null: no phase B).  A separate instantiation because phase B's unrolled
// pair walk keeps ~64 loop-invariant LDS addresses in VGPRs (scripts/vgpr_liveness.py): in the one-kernel version they
// put BOTH modes at the 256-VGPR cap.
template <typename T, int KHT, bool DCOL>
__global__ __launch_bounds__(256, 2) void dcn_bwd_data_tc_kernel(DcnShape s, BwArgs a) {
  typedef Mma<T> M;
  extern __shared__ __attribute__((aligned(16))) unsigned char bw_smem[];
  __shared__ __attribute__((aligned(8))) BwPair pairs[256];
  __shared__ __attribute__((aligned(16))) float Cs[64 * BW_CPITCH];
  float* red = reinterpret_cast<float*>(bw_smem);  // [K2][64][3]
  BwEntry* ent_all = reinterpret_cast<BwEntry*>(bw_smem + (((size_t)s.K2 * 64 * 3 * 4 + 15) & ~(size_t)15));  // [K2][64]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int per_xcd = (a.total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= a.total) return;
  const int cz = logical % a.csplit;
  int tile = logical / a.csplit;
  const int dgi = tile % s.DG; tile /= s.DG;
  const int tx = tile % a.tiles_x; tile /= a.tiles_x;
  const int ty = tile % a.tiles_y;
  const int b = tile / a.tiles_y;

  for (int i = tid; i < s.K2 * 64 * 3; i += 256) red[i] = 0.f;

  const T* offset = (const T*)a.offset;
  const T* mask = (const T*)a.mask;
  const char* xb = (const char*)a.x;
  const T* gout = (const T*)a.gout;
  const raw16* wp = (const raw16*)a.wp;
  const int KS = s.Cog / 16, KH = KS / 2;  // ksteps in total / per wave half
  const int nt = wid & 1, khalf = wid >> 1;
  // this lane's position as B-operand column (n = nt*32 + (lane & 31))
  const int nB = nt * 32 + (lane & 31);
  const int hoB = ty * 8 + (nB >> 3), woB = tx * 8 + (nB & 7);
  const bool validB = hoB < s.Ho && woB < s.Wo;
  const long pB = ((long)b * s.Ho + hoB) * s.Wo + woB;
  const uint32_t pixbytes = (uint32_t)s.C * (uint32_t)sizeof(T);
  // this thread's (position, corner) for the tables
  const int nE = tid >> 2, cE = tid & 3;
  const int hoE = ty * 8 + (nE >> 3), woE = tx * 8 + (nE & 7);
  const bool validE = hoE < s.Ho && woE < s.Wo;
  const int lE = hoE * s.Wo + woE;

  // ---- (0) tables of every tap, once: thread (position nE, corner cE) evaluates the sample of its position; the
  //          corner-0 thread writes the entry
  for (int tap = 0; tap < s.K2; tap++) {
        BwEntry e;
#pragma unroll
        for (int t = 0; t < 4; t++) { e.pix[t] = 0u; e.w[t] = 0.f; }
        e.lh = e.lw = 0.f; e.m = 0.f; e.flags = 0u;
        if (validE) {
          const int i = tap / s.kw, j = tap - i * s.kw;
          const long obase = ((long)b * s.DG + dgi) * 2 * s.K2;
          const float off_h = to_f32(offset[(obase + 2 * tap) * s.L + lE]);
          const float off_w = to_f32(offset[(obase + 2 * tap + 1) * s.L + lE]);
          e.m = mask ? to_f32(mask[(((long)b * s.DG + dgi) * s.K2 + tap) * s.L + lE]) : 1.f;
          const float h_im = (float)(hoE * s.sh - s.ph + i * s.dh) + off_h;
          const float w_im = (float)(woE * s.sw - s.pw + j * s.dw) + off_w;
          if (h_im > -1.f && w_im > -1.f && h_im < (float)s.H && w_im < (float)s.W) {
            const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
            const float hh = 1.f - lh, hw = 1.f - lw;
            e.lh = lh; e.lw = lw; e.flags = 16u;
            const long rowbase = (long)b * s.H;
            if (h_low >= 0 && w_low >= 0) { e.pix[0] = (uint32_t)((rowbase + h_low) * s.W + w_low); e.w[0] = hh * hw; e.flags |= 1u; }
            if (h_low >= 0 && w_high <= s.W - 1) { e.pix[1] = (uint32_t)((rowbase + h_low) * s.W + w_high); e.w[1] = hh * lw; e.flags |= 2u; }
            if (h_high <= s.H - 1 && w_low >= 0) { e.pix[2] = (uint32_t)((rowbase + h_high) * s.W + w_low); e.w[2] = lh * hw; e.flags |= 4u; }
            if (h_high <= s.H - 1 && w_high <= s.W - 1) { e.pix[3] = (uint32_t)((rowbase + h_high) * s.W + w_high); e.w[3] = lh * lw; e.flags |= 8u; }
          }
        }
        if (cE == 0) ent_all[tap * 64 + nE] = e;
  }
  __syncthreads();  // (also orders the zeroing of `red` above against its first use)

  const bool doA = (a.goff || a.gmask) && !(a.ablate & 2);  // uniform
  const raw16 zero = {0u, 0u, 0u, 0u};
  // (c) the wave's dY fragments (B operand: its 32 positions x its K half), loaded once
  raw16 bh[KHT > 0 ? KHT : 1];
  if constexpr (KHT > 0) {
    const T* gsrc0 = gout + pB * s.Co + (lane >> 5) * 8 + khalf * KHT * 16;
#pragma unroll
    for (int u = 0; u < KHT; u++) bh[u] = validB ? *reinterpret_cast<const raw16*>(gsrc0 + u * 16) : zero;
  }

  const int nchunk = s.cpg >> 6;
  const int c_lo = dgi * s.cpg + 64 * (int)((long)cz * nchunk / a.csplit);
  const int c_hi = dgi * s.cpg + 64 * (int)((long)(cz + 1) * nchunk / a.csplit);
  for (int cabs = c_lo; cabs < c_hi; cabs += 64) {
    const int g = cabs / s.Cg, c64 = (cabs - g * s.Cg) >> 6;
    for (int tap = 0; tap < s.K2; tap++) {
      // ---- (1) this thread's table entry (its own position nE; corner cE for the pair record of phase B), and the
      //          corner gathers of phase A, issued now
      const BwEntry e = ent_all[tap * 64 + nE];
      if (!DCOL && a.gx) {
        const float wsel = cE == 0 ? e.w[0] : cE == 1 ? e.w[1] : cE == 2 ? e.w[2] : e.w[3];
        const uint32_t psel = cE == 0 ? e.pix[0] : cE == 1 ? e.pix[1] : cE == 2 ? e.pix[2] : e.pix[3];
        pairs[tid] = BwPair{wsel * e.m, psel * (uint32_t)s.C};
      }
      raw16 raw[2][4];
      const bool gatherA = doA && (e.flags & 16u);
      auto issue_gather = [&](int h) __attribute__((always_inline)) {
        const uint32_t cofs = (uint32_t)(cabs + (cE + 4 * h) * 8) * (uint32_t)sizeof(T);
#pragma unroll
        for (int c = 0; c < 4; c++)
          raw[h][c] = *reinterpret_cast<const raw16*>(xb + ((size_t)e.pix[c] * pixbytes + cofs));
      };
      // (both 8-channel halves up front when the registers allow -- KHT == 4: 256 VGPRs, no spill --, otherwise the
      // second half behind the MFMA section, where the weight fragments are dead)
      constexpr bool EARLY_BOTH = DCOL && KHT == 4;
      if (DCOL && gatherA) {  // (the atomics mode sits at its register cap: it gathers where it always did, in phase A)
        issue_gather(0);
        if (EARLY_BOTH) issue_gather(1);
      }
      // ---- (2) dcol tile by MFMA: rows = the 64 channels of the chunk (2 M-tiles), cols = this wave's 32
      //          positions, K = this wave half's share of the group's output channels
      f32x16_t acc[2];
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
      if (!(a.ablate & 1)) {
        const raw16* wsrc = wp + ((((size_t)g * s.K2 + tap) * (s.Cg >> 6) + c64) * 2 * KS) * 64 + lane;
        const T* gsrc = gout + pB * s.Co + (long)g * s.Cog + (lane >> 5) * 8;
        constexpr int KB = 2;
        if constexpr (KHT > 0) {
          // weights in groups of KB k-steps, one group ahead; dY from the registers; fully unrolled (static indices)
          raw16 aw[2][2][KB];
          const int kw0 = khalf * KHT;
          auto ldA = [&](int i, raw16 (&A)[2][KB]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < KB; u++) {
              A[0][u] = wsrc[(size_t)(kw0 + i + u) * 64];
              A[1][u] = wsrc[(size_t)(KS + kw0 + i + u) * 64];
            }
          };
          ldA(0, aw[0]);
#pragma unroll
          for (int i = 0; i < KHT; i += KB) {
            if (i + KB < KHT) ldA(i + KB, aw[((i / KB) + 1) & 1]);
#pragma unroll
            for (int u = 0; u < KB; u++) {
              const typename M::frag bq = __builtin_bit_cast(typename M::frag, bh[i + u]);
              acc[0] = M::mma(__builtin_bit_cast(typename M::frag, aw[(i / KB) & 1][0][u]), bq, acc[0]);
              acc[1] = M::mma(__builtin_bit_cast(typename M::frag, aw[(i / KB) & 1][1][u]), bq, acc[1]);
            }
          }
        } else {
        raw16 af[2][2][KB], bf[2][KB];
        auto ld = [&](int k0, raw16 (&A)[2][KB], raw16 (&Bq)[KB]) __attribute__((always_inline)) {
#pragma unroll
          for (int u = 0; u < KB; u++) {
            const int ks = min(k0 + u, KS - 1);
            A[0][u] = wsrc[(size_t)ks * 64];
            A[1][u] = wsrc[(size_t)(KS + ks) * 64];
            Bq[u] = validB ? *reinterpret_cast<const raw16*>(gsrc + ks * 16) : zero;
          }
        };
        auto mm = [&](int k0, int k_hi, const raw16 (&A)[2][KB], const raw16 (&Bq)[KB]) __attribute__((always_inline)) {
#pragma unroll
          for (int u = 0; u < KB; u++)
            if (k0 + u < k_hi) {  // uniform
              const typename M::frag bq = __builtin_bit_cast(typename M::frag, Bq[u]);
              acc[0] = M::mma(__builtin_bit_cast(typename M::frag, A[0][u]), bq, acc[0]);
              acc[1] = M::mma(__builtin_bit_cast(typename M::frag, A[1][u]), bq, acc[1]);
            }
        };
        const int k_lo = khalf * KH, k_hi = k_lo + KH;
        ld(k_lo, af[0], bf[0]);
        for (int k0 = k_lo; k0 < k_hi; k0 += 2 * KB) {
          if (k0 + KB < k_hi) ld(k0 + KB, af[1], bf[1]);
          mm(k0, k_hi, af[0], bf[0]);
          if (k0 + KB < k_hi) {
            if (k0 + 2 * KB < k_hi) ld(k0 + 2 * KB, af[0], bf[0]);
            mm(k0 + KB, k_hi, af[1], bf[1]);
          }
        }
        }
      }
      if (DCOL && !EARLY_BOTH && gatherA) issue_gather(1);
      // ---- (3) the two K halves meet in LDS: Cs[position][channel]
      if (khalf == 0) {
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
          for (int rg = 0; rg < 4; rg++)
            *reinterpret_cast<float4*>(&Cs[nB * BW_CPITCH + 32 * m + 8 * rg + 4 * (lane >> 5)]) =
                make_float4(acc[m][4 * rg], acc[m][4 * rg + 1], acc[m][4 * rg + 2], acc[m][4 * rg + 3]);
      }
      __syncthreads();
      if (khalf == 1) {
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
          for (int rg = 0; rg < 4; rg++) {
            float4* q = reinterpret_cast<float4*>(&Cs[nB * BW_CPITCH + 32 * m + 8 * rg + 4 * (lane >> 5)]);
            float4 v = *q;
            v.x += acc[m][4 * rg]; v.y += acc[m][4 * rg + 1]; v.z += acc[m][4 * rg + 2]; v.w += acc[m][4 * rg + 3];
            *q = v;
          }
      }
      __syncthreads();
      // ---- (3b) column-gather path: the dcol tile leaves as 16-bit rows col[position][tap][channel chunk] (128 B each)
      if (DCOL) {
        const int n = tid >> 2, q = tid & 3;  // position, 16-channel quarter of the chunk
        const int ho = ty * 8 + (n >> 3), wo = tx * 8 + (n & 7);
        if (ho < s.Ho && wo < s.Wo) {
          const long pp = ((long)b * s.Ho + ho) * s.Wo + wo;
          const float* cp = &Cs[n * BW_CPITCH + 16 * q];
          T* dst = (T*)a.dcol + (pp * s.K2 + tap) * s.C + cabs + 16 * q;
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const float4 d0 = *reinterpret_cast<const float4*>(cp + 8 * h), d1 = *reinterpret_cast<const float4*>(cp + 8 * h + 4);
            const float f[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            uint32_t wv[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
              wv[u] = (uint32_t)from_f32<T>(f[2 * u]).v | ((uint32_t)from_f32<T>(f[2 * u + 1]).v << 16);
            *reinterpret_cast<raw16*>(dst + 8 * h) = raw16{wv[0], wv[1], wv[2], wv[3]};
          }
        }
      }
      // ---- (4) phase A: d(offset), d(mask).  thread = (position n, channels q*8.. and (q+4)*8..)
      if (doA) {
        const int n = nE, q = cE;  // (the thread that issued the gathers above)
        const uint32_t flags = e.flags;
        float s_h = 0.f, s_w = 0.f, s_m = 0.f;
        if (flags & 16u) {
          if (!DCOL) { issue_gather(0); issue_gather(1); }
          const float lh = e.lh, lw = e.lw, hh = 1.f - lh, hw = 1.f - lw, m = e.m;
          const float w0 = e.w[0], w1 = e.w[1], w2 = e.w[2], w3 = e.w[3];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            float v[4][8];
#pragma unroll
            for (int c = 0; c < 4; c++) {
              tc_unpack(raw[h][c], v[c], T{});
              if (!(flags & (1u << c))) {
#pragma unroll
                for (int u = 0; u < 8; u++) v[c][u] = 0.f;
              }
            }
            const float* cp = &Cs[n * BW_CPITCH + (q + 4 * h) * 8];
            const float4 d0 = *reinterpret_cast<const float4*>(cp);
            const float4 d1 = *reinterpret_cast<const float4*>(cp + 4);
            const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const float val = w0 * v[0][u] + w1 * v[1][u] + w2 * v[2][u] + w3 * v[3][u];
              const float dvh = -hw * v[0][u] - lw * v[1][u] + hw * v[2][u] + lw * v[3][u];
              const float dvw = -hh * v[0][u] + hh * v[1][u] - lh * v[2][u] + lh * v[3][u];
              s_h += dvh * d[u] * m;
              s_w += dvw * d[u] * m;
              s_m += d[u] * val;
            }
          }
        }
        s_h += __shfl_xor(s_h, 1); s_w += __shfl_xor(s_w, 1); s_m += __shfl_xor(s_m, 1);
        s_h += __shfl_xor(s_h, 2); s_w += __shfl_xor(s_w, 2); s_m += __shfl_xor(s_m, 2);
        if (q == 0) {  // the only thread that touches red[tap][n]
          float* rp = red + (tap * 64 + n) * 3;
          rp[0] += s_h; rp[1] += s_w; rp[2] += s_m;
        }
      }
      // ---- (5) phase B: dX.  wave instruction = one (position, corner) pair x 64 channels
      if (!DCOL && a.gx && !(a.ablate & 4)) {
        float* gxc = a.gx + cabs + lane;
        constexpr int UB = 8;
        for (int it0 = 0; it0 < 64; it0 += UB) {
          float wg[UB], dv[UB];
          uint32_t eo[UB];
#pragma unroll
          for (int u = 0; u < UB; u++) {
            const int pid = (it0 + u) * 4 + wid;
            const BwPair pr = pairs[pid];
            wg[u] = pr.wgt; eo[u] = pr.eofs;
            dv[u] = Cs[(pid >> 2) * BW_CPITCH + lane];
          }
#pragma unroll
          for (int u = 0; u < UB; u++) {
            const float wgu = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wg[u])));
            const uint32_t eou = (uint32_t)__builtin_amdgcn_readfirstlane((int)eo[u]);
            if (wgu != 0.f) {  // scalar branch
              const float val = wgu * dv[u];
              if (a.ablate & 16) { if (val == 123.456f) Cs[0] = val; }
              else atomicAdd(gxc + eou, val);
            }
          }
        }
      }
      __syncthreads();  // ent / pairs / Cs are rewritten by the next stage
    }
  }
  // ---- d(offset) / d(mask): with csplit == 1 every (tap, position) of this deformable group has exactly
  //      one owner (plain stores); otherwise the csplit channel shares add into the zero-filled output
  __syncthreads();
  for (int i = tid; i < s.K2 * 64; i += 256) {
    const int tap = i >> 6, n = i & 63;
    const int ho = ty * 8 + (n >> 3), wo = tx * 8 + (n & 7);
    if (ho >= s.Ho || wo >= s.Wo) continue;
    const int l = ho * s.Wo + wo;
    const float* rp = red + i * 3;
    const long ob = ((long)b * s.DG + dgi) * 2 * s.K2;
    float* ph = a.goff ? a.goff + (ob + 2 * tap) * s.L + l : nullptr;
    float* pm = a.gmask ? a.gmask + (((long)b * s.DG + dgi) * s.K2 + tap) * s.L + l : nullptr;
    if (a.csplit == 1) {
      if (ph) { ph[0] = rp[0]; ph[s.L] = rp[1]; }
      if (pm) pm[0] = rp[2];
    } else {
      if (ph) { atomicAdd(ph, rp[0]); atomicAdd(ph + s.L, rp[1]); }
      if (pm) atomicAdd(pm, rp[2]);
    }
  }
}

// ---- backward data, column-gather mode, WAVE-SPECIALISED (r04) ------------------------------------------------
// Same mathematics and the same stage = (64-channel chunk, tap) as dcn_bwd_data_tc_kernel<.., DCOL = true>, but the
// two halves of a stage run CONCURRENTLY on different waves of the workgroup, one barrier per stage:
//   matrix waves 0, 1     dcol[64 channels][32 positions each] = W^T[tap, chunk] dY over the FULL K = Cog (no K halves
//                         meeting in LDS), written as fp32 into Cs[stage & 1];
//   consumer waves 2..5   stage - 1: the tile leaves as 16-bit column rows (what dcn_gather_dx_kernel reads) and phase A
//                         -- corner gathers of x, bilinear value and coordinate derivatives, d(offset) / d(mask)
//                         reduced over channels -- runs on Cs[(stage - 1) & 1]; the corner gathers of the NEXT half
//                         stage are in flight while the current one is combined (register double buffer).
// The one-wave-does-everything kernel is a chain of four barriers per stage with two waves per SIMD (5.5 us per stage
// measured, 1.6 us of it instruction issue: profiles/r04/dcn_bwd_data_ab.txt); moving loads earlier inside a wave does
// not help, because a wave's s_waitcnt retires its loads in order.  Here the matrix pipe, the L2 round trips of both
// sides and the VALU of phase A overlap by construction, as in the forward kernel.
// Tables of all taps are built once per workgroup (they do not depend on the channel chunk).
// GK: k-steps per operand request of the matrix waves; Cog / 16 is a multiple of it (4 for Cog % 64 == 0, else 2).
template <typename T, int GK>
__global__ __launch_bounds__(384) void dcn_bwd_data_ws_kernel(DcnShape s, BwArgs a) {
  typedef Mma<T> M;
  extern __shared__ __attribute__((aligned(16))) unsigned char bw_smem[];
  float* red = reinterpret_cast<float*>(bw_smem);  // [K2][64][3]
  const size_t red_bytes = ((size_t)s.K2 * 64 * 3 * 4 + 15) & ~(size_t)15;
  BwEntry* ent_all = reinterpret_cast<BwEntry*>(bw_smem + red_bytes);  // [K2][64]
  float* Cs2 = reinterpret_cast<float*>(bw_smem + red_bytes + (size_t)s.K2 * 64 * sizeof(BwEntry));  // [2][64 * BW_CPITCH]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int per_xcd = (a.total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= a.total) return;
  const int cz = logical % a.csplit;
  int tile = logical / a.csplit;
  const int dgi = tile % s.DG; tile /= s.DG;
  const int tx = tile % a.tiles_x; tile /= a.tiles_x;
  const int ty = tile % a.tiles_y;
  const int b = tile / a.tiles_y;

  for (int i = tid; i < s.K2 * 64 * 3; i += 384) red[i] = 0.f;
  const T* offset = (const T*)a.offset;
  const T* mask = (const T*)a.mask;
  // ---- tables of every tap, once (deform_conv_cuda_kernel.cu:785-860): entry (tap, position) by thread tap * 64 + n
  for (int i = tid; i < s.K2 * 64; i += 384) {
    const int tap = i >> 6, n = i & 63;
    const int ho = ty * 8 + (n >> 3), wo = tx * 8 + (n & 7);
    BwEntry e;
#pragma unroll
    for (int t = 0; t < 4; t++) { e.pix[t] = 0u; e.w[t] = 0.f; }
    e.lh = e.lw = 0.f; e.m = 0.f; e.flags = 0u;
    if (ho < s.Ho && wo < s.Wo) {
      const int l = ho * s.Wo + wo;
      const int ki = tap / s.kw, kj = tap - ki * s.kw;
      const long obase = ((long)b * s.DG + dgi) * 2 * s.K2;
      const float off_h = to_f32(offset[(obase + 2 * tap) * s.L + l]);
      const float off_w = to_f32(offset[(obase + 2 * tap + 1) * s.L + l]);
      e.m = mask ? to_f32(mask[(((long)b * s.DG + dgi) * s.K2 + tap) * s.L + l]) : 1.f;
      const float h_im = (float)(ho * s.sh - s.ph + ki * s.dh) + off_h;
      const float w_im = (float)(wo * s.sw - s.pw + kj * s.dw) + off_w;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)s.H && w_im < (float)s.W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        const float hh = 1.f - lh, hw = 1.f - lw;
        e.lh = lh; e.lw = lw; e.flags = 16u;
        const long rowbase = (long)b * s.H;
        if (h_low >= 0 && w_low >= 0) { e.pix[0] = (uint32_t)((rowbase + h_low) * s.W + w_low); e.w[0] = hh * hw; e.flags |= 1u; }
        if (h_low >= 0 && w_high <= s.W - 1) { e.pix[1] = (uint32_t)((rowbase + h_low) * s.W + w_high); e.w[1] = hh * lw; e.flags |= 2u; }
        if (h_high <= s.H - 1 && w_low >= 0) { e.pix[2] = (uint32_t)((rowbase + h_high) * s.W + w_low); e.w[2] = lh * hw; e.flags |= 4u; }
        if (h_high <= s.H - 1 && w_high <= s.W - 1) { e.pix[3] = (uint32_t)((rowbase + h_high) * s.W + w_high); e.w[3] = lh * lw; e.flags |= 8u; }
      }
    }
    ent_all[i] = e;
  }
  __syncthreads();

  const int nchunk = s.cpg >> 6;
  const int c_lo = dgi * s.cpg + 64 * (int)((long)cz * nchunk / a.csplit);
  const int c_hi = dgi * s.cpg + 64 * (int)((long)(cz + 1) * nchunk / a.csplit);
  const int nst = ((c_hi - c_lo) >> 6) * s.K2;  // stages of this workgroup: (chunk, tap), tap fastest
  const int KS = s.Cog / 16;

  if (wid < 2) {
    // ================================ MATRIX waves ====================================================
    const T* gout = (const T*)a.gout;
    const raw16* wp = (const raw16*)a.wp;
    const int nB = wid * 32 + (lane & 31);
    const int hoB = ty * 8 + (nB >> 3), woB = tx * 8 + (nB & 7);
    const bool validB = hoB < s.Ho && woB < s.Wo;
    const long pB = ((long)b * s.Ho + hoB) * s.Wo + woB;
    const raw16 zero = {0u, 0u, 0u, 0u};
    // Operand fragments come straight from L2 (weights pre-packed in fragment order, dY as [position][Co]).  r04, first
    // version: two k-steps in flight -> Cog / 32 DEPENDENT L2 round trips per stage; alone (consumers ablated) the matrix
    // side took 48 of the kernel's 86 us (profiles/r04/dcn_bwd_data_ws_ablation.txt).  Now a group of GK k-steps is
    // requested at once, and the next group -- of this stage or of the NEXT one -- is
    // requested as soon as the MFMAs have consumed the registers, i.e. before the tile is written and the barrier is
    // waited for: one exposed round trip per group, none at stage boundaries.
    raw16 ga[2][GK], gb[GK];
    const int ngrp = KS / GK;
    auto issue_group = [&](int st, int kg) __attribute__((always_inline)) {
      const int cabs = c_lo + 64 * (st / s.K2), tap = st % s.K2;
      const int g = cabs / s.Cg, c64 = (cabs - g * s.Cg) >> 6;
      const raw16* wsrc = wp + ((((size_t)g * s.K2 + tap) * (s.Cg >> 6) + c64) * 2 * KS) * 64 + lane;
      const T* gsrc = gout + pB * s.Co + (long)g * s.Cog + (lane >> 5) * 8;
      // (KS is a multiple of GK: no clamping, the addresses are base + constant -- base registers and immediate offsets
      // instead of an address pair per load)
      const raw16* w0 = wsrc + (size_t)(kg * GK) * 64;
      const raw16* w1 = wsrc + (size_t)(KS + kg * GK) * 64;
      const T* g0 = gsrc + kg * GK * 16;
#pragma unroll
      for (int u = 0; u < GK; u++) {
        ga[0][u] = w0[u * 64];
        ga[1][u] = w1[u * 64];
        gb[u] = validB ? *reinterpret_cast<const raw16*>(g0 + u * 16) : zero;
      }
    };
    if (!(a.ablate & 1)) issue_group(0, 0);
    for (int st = 0; st < nst; st++) {
      f32x16_t acc[2];
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
      if (!(a.ablate & 1)) {  // (profiling: D2AMD_DCN_ABLATE_BWD bit 0 = no operand loads / MFMAs)
        for (int kg = 0; kg < ngrp; kg++) {
#pragma unroll
          for (int u = 0; u < GK; u++) {
            const typename M::frag bq = __builtin_bit_cast(typename M::frag, gb[u]);
            acc[0] = M::mma(__builtin_bit_cast(typename M::frag, ga[0][u]), bq, acc[0]);
            acc[1] = M::mma(__builtin_bit_cast(typename M::frag, ga[1][u]), bq, acc[1]);
          }
          if (kg + 1 < ngrp) issue_group(st, kg + 1);
          else if (st + 1 < nst) issue_group(st + 1, 0);
        }
      }
      float* Cs = Cs2 + (st & 1) * (64 * BW_CPITCH);
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
          *reinterpret_cast<float4*>(&Cs[nB * BW_CPITCH + 32 * m + 8 * rg + 4 * (lane >> 5)]) =
              make_float4(acc[m][4 * rg], acc[m][4 * rg + 1], acc[m][4 * rg + 2], acc[m][4 * rg + 3]);
      __syncthreads();  // barrier st: tile st is in Cs[st & 1]; the consumers are done with Cs[(st + 1) & 1]
    }
    __syncthreads();  // barrier nst: the consumers' last stage
  } else {
    // ================================ CONSUMER waves ==================================================
    const int atid = tid - 128;
    const int n = atid >> 2, q = atid & 3;  // position, channel slot: channels (q + 4 h) * 8 .. + 8 of the chunk, h = 0, 1
    const int ho = ty * 8 + (n >> 3), wo = tx * 8 + (n & 7);
    const bool valid = ho < s.Ho && wo < s.Wo;
    const long pp = ((long)b * s.Ho + ho) * s.Wo + wo;
    const char* xb = (const char*)a.x;
    const uint32_t pixbytes = (uint32_t)s.C * (uint32_t)sizeof(T);
    const bool doA = (a.goff || a.gmask) && !(a.ablate & 2);  // uniform
    // half stages u = 2 st + h; the gathers of u + 1 fly while u is combined
    raw16 raw[2][4];
    auto issue = [&](int u, raw16 (&r)[4]) __attribute__((always_inline)) {
      const int st = u >> 1, h = u & 1;
      const int cabs = c_lo + 64 * (st / s.K2), tap = st % s.K2;
      const uint4 px = *reinterpret_cast<const uint4*>(&ent_all[tap * 64 + n].pix[0]);
      const uint32_t cofs = (uint32_t)(cabs + (q + 4 * h) * 8) * (uint32_t)sizeof(T);
      r[0] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.x * pixbytes + cofs));
      r[1] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.y * pixbytes + cofs));
      r[2] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.z * pixbytes + cofs));
      r[3] = *reinterpret_cast<const raw16*>(xb + ((size_t)px.w * pixbytes + cofs));
    };
    if (doA && nst > 0) issue(0, raw[0]);
    __syncthreads();  // barrier 0 (the matrix waves' stage 0)
    for (int st = 0; st < nst; st++) {
      const int cabs = c_lo + 64 * (st / s.K2), tap = st % s.K2;
      const float* Cs = Cs2 + (st & 1) * (64 * BW_CPITCH);
      // ---- the dcol tile leaves as 16-bit rows col[position][tap][channel chunk]: thread = (position, 16-channel quarter)
      if (valid && !(a.ablate & 8)) {  // (bit 3 = no column store)
        const float* cp = &Cs[n * BW_CPITCH + 16 * q];
        T* dst = (T*)a.dcol + (pp * s.K2 + tap) * s.C + cabs + 16 * q;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const float4 d0 = *reinterpret_cast<const float4*>(cp + 8 * h), d1 = *reinterpret_cast<const float4*>(cp + 8 * h + 4);
          const float f[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
          *reinterpret_cast<raw16*>(dst + 8 * h) = tc_pack(f, T{});
        }
      }
      // ---- phase A on the tile
      if (doA) {
        const BwEntry& e = ent_all[tap * 64 + n];
        const uint32_t flags = e.flags;
        const float lh = e.lh, lw = e.lw, hh = 1.f - lh, hw = 1.f - lw, m = e.m;
        const float w0 = e.w[0], w1 = e.w[1], w2 = e.w[2], w3 = e.w[3];
        float s_h = 0.f, s_w = 0.f, s_m = 0.f;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int u = 2 * st + h;
          if (u + 1 < 2 * nst && !(a.ablate & 32)) issue(u + 1, raw[(h + 1) & 1]);  // (uniform; always valid addresses; bit 5 = no gathers)
          if (flags & 16u) {
            float v[4][8];
#pragma unroll
            for (int c = 0; c < 4; c++) {
              tc_unpack(raw[h][c], v[c], T{});
              if (!(flags & (1u << c))) {
#pragma unroll
                for (int k = 0; k < 8; k++) v[c][k] = 0.f;
              }
            }
            const float* cp = &Cs[n * BW_CPITCH + (q + 4 * h) * 8];
            const float4 d0 = *reinterpret_cast<const float4*>(cp);
            const float4 d1 = *reinterpret_cast<const float4*>(cp + 4);
            const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
              const float val = w0 * v[0][k] + w1 * v[1][k] + w2 * v[2][k] + w3 * v[3][k];
              const float dvh = -hw * v[0][k] - lw * v[1][k] + hw * v[2][k] + lw * v[3][k];
              const float dvw = -hh * v[0][k] + hh * v[1][k] - lh * v[2][k] + lh * v[3][k];
              s_h += dvh * d[k] * m;
              s_w += dvw * d[k] * m;
              s_m += d[k] * val;
            }
          }
        }
        s_h += __shfl_xor(s_h, 1); s_w += __shfl_xor(s_w, 1); s_m += __shfl_xor(s_m, 1);
        s_h += __shfl_xor(s_h, 2); s_w += __shfl_xor(s_w, 2); s_m += __shfl_xor(s_m, 2);
        if (q == 0) {  // the only thread that touches red[tap][n]
          float* rp = red + (tap * 64 + n) * 3;
          rp[0] += s_h; rp[1] += s_w; rp[2] += s_m;
        }
      }
      __syncthreads();  // barrier st + 1
    }
  }
  // ---- d(offset) / d(mask): one owner per (tap, position) with csplit == 1 (plain stores), atomics otherwise
  __syncthreads();
  for (int i = tid; i < s.K2 * 64; i += 384) {
    const int tap = i >> 6, n = i & 63;
    const int ho = ty * 8 + (n >> 3), wo = tx * 8 + (n & 7);
    if (ho >= s.Ho || wo >= s.Wo) continue;
    const int l = ho * s.Wo + wo;
    const float* rp = red + i * 3;
    const long ob = ((long)b * s.DG + dgi) * 2 * s.K2;
    float* ph = a.goff ? a.goff + (ob + 2 * tap) * s.L + l : nullptr;
    float* pm = a.gmask ? a.gmask + (((long)b * s.DG + dgi) * s.K2 + tap) * s.L + l : nullptr;
    if (a.csplit == 1) {
      if (ph) { ph[0] = rp[0]; ph[s.L] = rp[1]; }
      if (pm) pm[0] = rp[2];
    } else {
      if (ph) { atomicAdd(ph, rp[0]); atomicAdd(ph + s.L, rp[1]); }
      if (pm) atomicAdd(pm, rp[2]);
    }
  }
}

// ---- backward data with an LDS-resident dX patch and conflict-free ownership --------------------------------
// EXPERIMENT, not the default (see dcn_tc_plan_bwd): measured slower than the all-atomics kernel above.
// Same stage structure as dcn_bwd_data_tc_kernel, 32-channel chunks.  dX contributions are not sent to global
// atomics one by one (the 0.75 lane-op/clk/CU floor that bounds that kernel) and not to LDS fp32 atomics either
// (ds_add_f32 is slower still, 0.2 lane-op/clk/CU): they are accumulated in an fp32 PATCH of the input in LDS --
// the pixels the 8x8 tile's taps can reach when displaced by up to R pixels -- with PLAIN read-modify-writes made
// race free by ownership: wave w owns channels 8w..8w+7 of the chunk, and one of its instructions covers the four
// corners of ONE sample x its 8 channels (four distinct pixels by construction); a wave's DS operations execute in
// order, different waves touch different channels.  The order of the additions is fixed (position, tap ascending):
// the patch part of dX is deterministic.  Samples displaced by more than R pixels fall back to global atomics.
// After the 9 taps of a chunk the patch is flushed with one 128-B coalesced global atomic per touched pixel.
struct BwEntryP {
  uint32_t pix[4];
  float w[4];
  float lh, lw, m;
  uint32_t flags;
  int py, px;   // patch coordinates of corner 0
  int pad[2];
};
static_assert(sizeof(BwEntryP) == 64, "BwEntryP layout");

struct BwpArgs {
  const void *x, *offset, *mask, *wp, *gout;
  float *gx, *goff, *gmask;
  int tiles_y, tiles_x, total, csplit, R, PHt, PWt;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void dcn_bwd_data_patch_kernel(DcnShape s, BwpArgs a) {
  typedef Mma<T> M;
  constexpr int CP = 36;  // floats per position row of the dcol tile
  extern __shared__ __attribute__((aligned(16))) unsigned char bwp_smem[];
  __shared__ __attribute__((aligned(16))) BwEntryP ent[64];
  __shared__ __attribute__((aligned(16))) float Cs[64 * CP];
  float* red = reinterpret_cast<float*>(bwp_smem);  // [K2][64][3]
  float* patch = red + s.K2 * 64 * 3;               // [PHt * PWt][32]
  const int npatch = a.PHt * a.PWt * 32;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int per_xcd = (a.total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= a.total) return;
  const int cz = logical % a.csplit;
  int tile = logical / a.csplit;
  const int dgi = tile % s.DG; tile /= s.DG;
  const int tx = tile % a.tiles_x; tile /= a.tiles_x;
  const int ty = tile % a.tiles_y;
  const int b = tile / a.tiles_y;
  const int oy = ty * 8 * s.sh - s.ph - a.R, ox = tx * 8 * s.sw - s.pw - a.R;

  for (int i = tid; i < s.K2 * 64 * 3 + npatch; i += 256) red[i] = 0.f;  // red and patch are contiguous

  const T* offset = (const T*)a.offset;
  const T* mask = (const T*)a.mask;
  const char* xb = (const char*)a.x;
  const T* gout = (const T*)a.gout;
  const raw16* wp = (const raw16*)a.wp;
  const int KS = s.Cog / 16, KH = KS / 2;
  const int nt = wid & 1, khalf = wid >> 1;
  const int nB = nt * 32 + (lane & 31);
  const int hoB = ty * 8 + (nB >> 3), woB = tx * 8 + (nB & 7);
  const bool validB = hoB < s.Ho && woB < s.Wo;
  const long pB = ((long)b * s.Ho + hoB) * s.Wo + woB;
  const uint32_t pixbytes = (uint32_t)s.C * (uint32_t)sizeof(T);
  const int nE = tid >> 2;
  const int hoE = ty * 8 + (nE >> 3), woE = tx * 8 + (nE & 7);
  const bool validE = hoE < s.Ho && woE < s.Wo;
  const int lE = hoE * s.Wo + woE;

  const int nchunk = s.cpg >> 5;
  const int c_lo = dgi * s.cpg + 32 * (int)((long)cz * nchunk / a.csplit);
  const int c_hi = dgi * s.cpg + 32 * (int)((long)(cz + 1) * nchunk / a.csplit);
  for (int cabs = c_lo; cabs < c_hi; cabs += 32) {
    const int g = cabs / s.Cg, crel = cabs - g * s.Cg, c64 = crel >> 6, mt = (crel >> 5) & 1;
    for (int tap = 0; tap < s.K2; tap++) {
      // ---- (1) table of this tap (4 threads per position evaluate it, one writes)
      {
        BwEntryP e;
#pragma unroll
        for (int t = 0; t < 4; t++) { e.pix[t] = 0u; e.w[t] = 0.f; }
        e.lh = e.lw = 0.f; e.m = 0.f; e.flags = 0u; e.py = e.px = -(1 << 20); e.pad[0] = e.pad[1] = 0;
        if (validE) {
          const int i = tap / s.kw, j = tap - i * s.kw;
          const long obase = ((long)b * s.DG + dgi) * 2 * s.K2;
          const float off_h = to_f32(offset[(obase + 2 * tap) * s.L + lE]);
          const float off_w = to_f32(offset[(obase + 2 * tap + 1) * s.L + lE]);
          e.m = mask ? to_f32(mask[(((long)b * s.DG + dgi) * s.K2 + tap) * s.L + lE]) : 1.f;
          const float h_im = (float)(hoE * s.sh - s.ph + i * s.dh) + off_h;
          const float w_im = (float)(woE * s.sw - s.pw + j * s.dw) + off_w;
          if (h_im > -1.f && w_im > -1.f && h_im < (float)s.H && w_im < (float)s.W) {
            const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
            const float hh = 1.f - lh, hw = 1.f - lw;
            e.lh = lh; e.lw = lw; e.flags = 16u;
            e.py = h_low - oy; e.px = w_low - ox;
            const long rowbase = (long)b * s.H;
            if (h_low >= 0 && w_low >= 0) { e.pix[0] = (uint32_t)((rowbase + h_low) * s.W + w_low); e.w[0] = hh * hw; e.flags |= 1u; }
            if (h_low >= 0 && w_high <= s.W - 1) { e.pix[1] = (uint32_t)((rowbase + h_low) * s.W + w_high); e.w[1] = hh * lw; e.flags |= 2u; }
            if (h_high <= s.H - 1 && w_low >= 0) { e.pix[2] = (uint32_t)((rowbase + h_high) * s.W + w_low); e.w[2] = lh * hw; e.flags |= 4u; }
            if (h_high <= s.H - 1 && w_high <= s.W - 1) { e.pix[3] = (uint32_t)((rowbase + h_high) * s.W + w_high); e.w[3] = lh * lw; e.flags |= 8u; }
          }
        }
        if ((tid & 3) == 0) ent[nE] = e;
      }
      // ---- (2) dcol tile by MFMA: 32 channels x this wave's 32 positions, K = this wave half's share of Co
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = 0.f;
      {
        const raw16* wsrc = wp + (((((size_t)g * s.K2 + tap) * (s.Cg >> 6) + c64) * 2 + mt) * KS) * 64 + lane;
        const T* gsrc = gout + pB * s.Co + (long)g * s.Cog + (lane >> 5) * 8;
        const raw16 zero = {0u, 0u, 0u, 0u};
        constexpr int KB = 4;
        raw16 af[2][KB], bf[2][KB];
        auto ld = [&](int k0, raw16 (&A)[KB], raw16 (&Bq)[KB]) __attribute__((always_inline)) {
#pragma unroll
          for (int u = 0; u < KB; u++) {
            const int ks = min(k0 + u, KS - 1);
            A[u] = wsrc[(size_t)ks * 64];
            Bq[u] = validB ? *reinterpret_cast<const raw16*>(gsrc + ks * 16) : zero;
          }
        };
        auto mm = [&](int k0, int k_hi, const raw16 (&A)[KB], const raw16 (&Bq)[KB]) __attribute__((always_inline)) {
#pragma unroll
          for (int u = 0; u < KB; u++)
            if (k0 + u < k_hi)
              acc = M::mma(__builtin_bit_cast(typename M::frag, A[u]), __builtin_bit_cast(typename M::frag, Bq[u]), acc);
        };
        const int k_lo = khalf * KH, k_hi = k_lo + KH;
        ld(k_lo, af[0], bf[0]);
        for (int k0 = k_lo; k0 < k_hi; k0 += 2 * KB) {
          if (k0 + KB < k_hi) ld(k0 + KB, af[1], bf[1]);
          mm(k0, k_hi, af[0], bf[0]);
          if (k0 + KB < k_hi) {
            if (k0 + 2 * KB < k_hi) ld(k0 + 2 * KB, af[0], bf[0]);
            mm(k0 + KB, k_hi, af[1], bf[1]);
          }
        }
      }
      // ---- (3) the two K halves meet in LDS: Cs[position][channel]
      if (khalf == 0) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
          *reinterpret_cast<float4*>(&Cs[nB * CP + 8 * rg + 4 * (lane >> 5)]) =
              make_float4(acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]);
      }
      __syncthreads();
      if (khalf == 1) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          float4* q = reinterpret_cast<float4*>(&Cs[nB * CP + 8 * rg + 4 * (lane >> 5)]);
          float4 v = *q;
          v.x += acc[4 * rg]; v.y += acc[4 * rg + 1]; v.z += acc[4 * rg + 2]; v.w += acc[4 * rg + 3];
          *q = v;
        }
      }
      __syncthreads();
      // ---- (4) phase A: d(offset), d(mask).  thread = (position n, 8 channels q*8..)
      if (a.goff || a.gmask) {
        const int n = tid >> 2, q = tid & 3;
        const BwEntryP& e = ent[n];
        const uint32_t flags = e.flags;
        float s_h = 0.f, s_w = 0.f, s_m = 0.f;
        if (flags & 16u) {
          const uint32_t cofs = (uint32_t)(cabs + q * 8) * (uint32_t)sizeof(T);
          raw16 raw[4];
#pragma unroll
          for (int c = 0; c < 4; c++) raw[c] = *reinterpret_cast<const raw16*>(xb + ((size_t)e.pix[c] * pixbytes + cofs));
          float v[4][8];
#pragma unroll
          for (int c = 0; c < 4; c++) {
            tc_unpack(raw[c], v[c], T{});
            if (!(flags & (1u << c))) {
#pragma unroll
              for (int u = 0; u < 8; u++) v[c][u] = 0.f;
            }
          }
          const float4 d0 = *reinterpret_cast<const float4*>(&Cs[n * CP + q * 8]);
          const float4 d1 = *reinterpret_cast<const float4*>(&Cs[n * CP + q * 8 + 4]);
          const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
          const float lh = e.lh, lw = e.lw, hh = 1.f - lh, hw = 1.f - lw, m = e.m;
          const float w0 = e.w[0], w1 = e.w[1], w2 = e.w[2], w3 = e.w[3];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const float val = w0 * v[0][u] + w1 * v[1][u] + w2 * v[2][u] + w3 * v[3][u];
            const float dvh = -hw * v[0][u] - lw * v[1][u] + hw * v[2][u] + lw * v[3][u];
            const float dvw = -hh * v[0][u] + hh * v[1][u] - lh * v[2][u] + lh * v[3][u];
            s_h += dvh * d[u] * m;
            s_w += dvw * d[u] * m;
            s_m += d[u] * val;
          }
        }
        s_h += __shfl_xor(s_h, 1); s_w += __shfl_xor(s_w, 1); s_m += __shfl_xor(s_m, 1);
        s_h += __shfl_xor(s_h, 2); s_w += __shfl_xor(s_w, 2); s_m += __shfl_xor(s_m, 2);
        if (q == 0) {
          float* rp = red + (tap * 64 + n) * 3;
          rp[0] += s_h; rp[1] += s_w; rp[2] += s_m;
        }
      }
      // ---- (5) phase B: dX into the patch.  wave w owns channels 8w..8w+7; lanes 0..31 = (corner, channel) of
      //          ONE sample per instruction (4 distinct pixels), positions in ascending order
      if (a.gx && lane < 32) {
        const int c = lane >> 3, ch = wid * 8 + (lane & 7);
#pragma unroll 4
        for (int n = 0; n < 64; n++) {
          const BwEntryP& e = ent[n];
          const float wgt = e.w[c] * e.m;
          if (wgt != 0.f && (e.flags & 16u)) {
            const float val = wgt * Cs[n * CP + ch];
            const int yy = e.py + (c >> 1), xx = e.px + (c & 1);
            if ((unsigned)yy < (unsigned)a.PHt && (unsigned)xx < (unsigned)a.PWt) {
              float* pp = &patch[(yy * a.PWt + xx) * 32 + ch];
              *pp = *pp + val;
            } else {
              atomicAdd(a.gx + (size_t)e.pix[c] * s.C + cabs + ch, val);
            }
          }
        }
      }
      __syncthreads();  // ent / Cs are rewritten by the next stage; patch rows are complete for the flush
    }
    // ---- flush the patch of this channel chunk: one coalesced atomic per touched (pixel, 32 channels)
    if (a.gx) {
      for (int i = tid; i < npatch; i += 256) {
        const float v = patch[i];
        if (v != 0.f) {
          patch[i] = 0.f;
          const int pixel = i >> 5, ch = i & 31;
          const int yy = pixel / a.PWt, xx = pixel - yy * a.PWt;
          const int hy = oy + yy, wx = ox + xx;
          if (hy >= 0 && hy < s.H && wx >= 0 && wx < s.W)
            atomicAdd(a.gx + (((size_t)b * s.H + hy) * s.W + wx) * s.C + cabs + ch, v);
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();
  for (int i = tid; i < s.K2 * 64; i += 256) {
    const int tap = i >> 6, n = i & 63;
    const int ho = ty * 8 + (n >> 3), wo = tx * 8 + (n & 7);
    if (ho >= s.Ho || wo >= s.Wo) continue;
    const int l = ho * s.Wo + wo;
    const float* rp = red + i * 3;
    const long ob = ((long)b * s.DG + dgi) * 2 * s.K2;
    float* ph = a.goff ? a.goff + (ob + 2 * tap) * s.L + l : nullptr;
    float* pm = a.gmask ? a.gmask + (((long)b * s.DG + dgi) * s.K2 + tap) * s.L + l : nullptr;
    if (a.csplit == 1) {
      if (ph) { ph[0] = rp[0]; ph[s.L] = rp[1]; }
      if (pm) pm[0] = rp[2];
    } else {
      if (ph) { atomicAdd(ph, rp[0]); atomicAdd(ph + s.L, rp[1]); }
      if (pm) atomicAdd(pm, rp[2]);
    }
  }
}

// KHT instantiation by shape: the dY fragments stay in registers for one conv group and Cog / 32 in {4, 8}
template <typename T>
static int launch_bwd_data_tc(const DcnShape& s, const BwArgs& a, int grid, size_t lds, hipStream_t st) {
  const int KH = s.Cog / 32;
  // MEASURED (profiles/r04/dcn_bwd_data_ab.txt, same box): hoisted dY fragments 103.7 us mean per block against 96.1
  // without (the fully unrolled weight pipeline is shallower than the rolling one) -- off unless D2AMD_DCN_BWD_HOIST=1
  const bool hoist = s.G == 1 && d2_prof_env("D2AMD_DCN_BWD_HOIST") != nullptr;
  auto launch = [&](auto kern) -> int {
    if (lds > 48 * 1024)
      D2_HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, s, a);
    return D2AMD_OK;
  };
  if (a.dcol && d2_prof_env("D2AMD_DCN_BWD_WS0") == nullptr) {  // wave-specialised (default); D2AMD_DCN_BWD_WS0: the one-role kernel
    const size_t lds_ws = lds + 2 * 64 * BW_CPITCH * sizeof(float);
    if (lds_ws <= 160 * 1024) {
      auto go = [&](auto kern) -> int {
        if (lds_ws > 48 * 1024)
          D2_HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ws));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(384), lds_ws, st, s, a);
        return D2AMD_OK;
      };
      // (MEASURED, profiles/r04/dcn_bwd_data_ws_ablation.txt: GK = 8 needs 168 VGPRs = exactly 3 waves per SIMD, which
      // the compiler only meets under amdgpu_waves_per_eu(3, 3) -- and then schedules the CONSUMER path worse: kernel
      // 86 -> 111 us.  GK = 4 stays at the consumers' 124 VGPRs: 80 us.)
      return (s.Cog / 16) % 4 == 0 ? go(dcn_bwd_data_ws_kernel<T, 4>) : go(dcn_bwd_data_ws_kernel<T, 2>);
    }
  }
  if (a.dcol) {
    if (hoist && KH == 4) return launch(dcn_bwd_data_tc_kernel<T, 4, true>);
    if (hoist && KH == 8) return launch(dcn_bwd_data_tc_kernel<T, 8, true>);
    return launch(dcn_bwd_data_tc_kernel<T, 0, true>);
  }
  return launch(dcn_bwd_data_tc_kernel<T, 0, false>);
}

TcBwPlan dcn_tc_plan_bwd(const DcnShape& s, int dtype) {
  TcBwPlan pl{};
  pl.ok = false;
  if (getenv("D2AMD_DCN_V1") || d2_prof_env("D2AMD_DCN_BWD_V1")) return pl;
  if (dtype != D2AMD_BF16 && dtype != D2AMD_F16) return pl;
  if (s.Cg % 64 != 0 || s.cpg % 64 != 0 || s.Cog % 32 != 0 || s.K2 > 64 || s.P <= 0) return pl;
  if ((long)s.B * s.H * s.W * s.C >= (1l << 31)) return pl;  // 32-bit element offsets into gx / x
  pl.tiles_y = cdiv(s.Ho, 8); pl.tiles_x = cdiv(s.Wo, 8);
  {  // small maps: split the channel chunks of a tile over several workgroups until ~4 per CU
    const long tiles = (long)s.B * pl.tiles_y * pl.tiles_x * s.DG;
    const int nchunk = s.cpg / 64;
    int cs = 1;
    // (the wave-specialised gather-mode kernel runs two 6-wave workgroups per CU = 512 slots: res3's 546 tiles stay
    // whole -- tables once per tile, d(offset) / d(mask) by plain stores, no zero fill; the one-role kernels fill 1,024)
    const bool ws = s.DG == 1 && (s.C == 64 || s.C == 128 || s.C == 256 || s.C == 512) && getenv("D2AMD_DCN_BWD_ATOMICS") == nullptr &&
        d2_prof_env("D2AMD_DCN_BWD_WS0") == nullptr;
    // (MEASURED: sizing for 512 slots leaves res3's 546 whole-tile workgroups in TWO rounds of full-length workgroups --
    // 1,092 half-length ones fill 2.13 rounds = 1.5 full lengths: bwd_res3 1.02 -> 1.29 ms per step.  1,024 stays.)
    const long want = 1024;
    (void)ws;
    while (cs < nchunk && tiles * cs < want) cs++;
    const char* e = getenv("D2AMD_DCN_CSPLIT");  // profiling switch
    if (e && atoi(e) >= 1) cs = atoi(e) < nchunk ? atoi(e) : nchunk;
    pl.csplit = cs;
  }
  pl.lds = (((size_t)s.K2 * 64 * 3 * 4 + 15) & ~(size_t)15) + (size_t)s.K2 * 64 * sizeof(BwEntry);  // red + the tables of all taps
  pl.wp_bytes = (size_t)s.Co * s.Cg * s.K2 * 2;
  // column-gather dX (no atomics): one deformable group, C = 64 * {1, 2, 4, 8}, sample ids fit 32 bits
  pl.gather = s.DG == 1 && (s.C == 64 || s.C == 128 || s.C == 256 || s.C == 512) && (long)s.P * s.K2 < (1l << 31) &&
      getenv("D2AMD_DCN_BWD_ATOMICS") == nullptr;
  // LDS patch variant (dcn_bwd_data_patch_kernel): the largest displacement margin R whose patch fits ~48 KB
  pl.R = -1; pl.PHt = pl.PWt = 0;
  {
    const size_t red_bytes = (size_t)s.K2 * 64 * 3 * 4;
    for (int r = 0; r <= 8; r++) {
      const long ph = 7l * s.sh + (long)(s.kh - 1) * s.dh + 2 + 2 * r, pw = 7l * s.sw + (long)(s.kw - 1) * s.dw + 2 + 2 * r;
      if (ph * pw * 32 * 4 + (long)red_bytes <= 54 * 1024) pl.R = r;
    }
    // MEASURED SLOWER than the all-atomics kernel (profiles/r01/v6_dcn_bwd_patch_sweep.txt: 890 vs 544 us for
    // res3, independent of R): a wave's 64 samples per stage form a serial chain of LDS read-modify-writes
    // (possible aliasing forbids batching them), ~150 cycles each.  Kept selectable for the record
    // (D2AMD_DCN_PATCH_R >= 0), off by default.
    const char* er = getenv("D2AMD_DCN_PATCH_R");
    pl.R = er ? (atoi(er) < pl.R ? atoi(er) : pl.R) : -1;
    if (pl.R >= 0) {
      pl.PHt = 7 * s.sh + (s.kh - 1) * s.dh + 2 + 2 * pl.R;
      pl.PWt = 7 * s.sw + (s.kw - 1) * s.dw + 2 + 2 * pl.R;
      pl.lds_patch = red_bytes + (size_t)pl.PHt * pl.PWt * 32 * 4;
      const long tiles = (long)s.B * pl.tiles_y * pl.tiles_x * s.DG;
      const int nchunk = s.cpg / 32;
      int cs = 1;
      while (cs < nchunk && tiles * cs < 1024) cs++;
      const char* e = getenv("D2AMD_DCN_CSPLIT");
      if (e && atoi(e) >= 1) cs = atoi(e) < nchunk ? atoi(e) : nchunk;
      pl.csplit_patch = cs;
    }
  }
  pl.ok = true;
  return pl;
}


// =====================================================================================================
// Backward w.r.t. the input WITHOUT atomics: column gather (r02).
// The kernel above spends 310 of its 535 us (res3) in device-scope fp32 atomics: 4 corners x 9 taps x C channels per
// position, retired at ~0.75 lane-ops / clk / CU.  Here dX is GATHERED, like the pooler backward:
//   (a) dcn_bin_samples_kernel: every sample (position, tap) appends {sample id, weight * mask} to the list of each
//       of its <= 4 corner PIXELS (one int atomic per corner -- 36 per position instead of 36 x C);
//   (b) dcn_bwd_data_tc_kernel with `dcol`: the same MFMA stages write the dcol tile as 16-bit rows
//       col[position][tap][C] (the column buffer of the reference's design, but 16-bit, backward only, and never read
//       by a GEMM) and still produce d(offset) / d(mask); phase B (the atomics) is gone;
//   (c) dcn_sort_lists_kernel + dcn_gather_dx_kernel: one wave per input pixel sorts its list by sample id (fixed
//       summation order: deterministic) and accumulates  dX[pixel, :] = sum_e w_e * col[sample_e, :]  in fp32 registers, every pixel
//       written once in the I/O dtype.  HBM / L2 bound: 36 rows of C x 2 B per pixel on average.
// Pixels that collect more than DG_CAP entries (adversarial offsets) park the excess in an overflow array that the
// gather scans when it is non-empty.  deformable_groups == 1 (R50's DCN); other shapes keep the kernel above.
constexpr int DG_CAP = 128;  // list entries per pixel: 2 per lane
struct __attribute__((aligned(8))) DgEntry { uint32_t sample; float w; };
struct __attribute__((aligned(16))) DgOverflow { uint32_t pix, sample; float w; uint32_t pad; };

template <typename T>
__global__ __launch_bounds__(256) void dcn_bin_samples_kernel(DcnShape s, const T* __restrict__ offset,
                                                             const T* __restrict__ mask, int* __restrict__ cnt,
                                                             DgEntry* __restrict__ lists, int* __restrict__ ovf_cnt,
                                                             DgOverflow* __restrict__ ovf) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)s.P * s.K2) return;
  const int tap = (int)(t % s.K2);
  const long pp = t / s.K2;
  const int b = (int)(pp / s.L), l = (int)(pp - (long)b * s.L);
  const int ho = l / s.Wo, wo = l - ho * s.Wo;
  const int i = tap / s.kw, j = tap - i * s.kw;
  // the table of dcn_bwd_data_tc_kernel, operation for operation (deformable group 0)
  const long obase = (long)b * 2 * s.K2;
  const float off_h = to_f32(offset[(obase + 2 * tap) * s.L + l]);
  const float off_w = to_f32(offset[(obase + 2 * tap + 1) * s.L + l]);
  const float m = mask ? to_f32(mask[((long)b * s.K2 + tap) * s.L + l]) : 1.f;
  const float h_im = (float)(ho * s.sh - s.ph + i * s.dh) + off_h;
  const float w_im = (float)(wo * s.sw - s.pw + j * s.dw) + off_w;
  if (!(h_im > -1.f && w_im > -1.f && h_im < (float)s.H && w_im < (float)s.W)) return;
  const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
  const float hh = 1.f - lh, hw = 1.f - lw;
  const int ys[4] = {h_low, h_low, h_high, h_high}, xs[4] = {w_low, w_high, w_low, w_high};
  const float ws[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
  for (int c = 0; c < 4; c++) {
    if (ys[c] < 0 || ys[c] > s.H - 1 || xs[c] < 0 || xs[c] > s.W - 1) continue;
    const float wg = ws[c] * m;
    if (wg == 0.f) continue;  // contributes nothing (the atomics kernel skips it too)
    const uint32_t pix = (uint32_t)(((long)b * s.H + ys[c]) * s.W + xs[c]);
    const int slot = atomicAdd(&cnt[pix], 1);
    if (slot < DG_CAP) {
      lists[(long)pix * DG_CAP + slot] = DgEntry{(uint32_t)t, wg};
    } else {
      const int o = atomicAdd(ovf_cnt, 1);
      ovf[o] = DgOverflow{pix, (uint32_t)t, wg, 0u};  // sized for every corner of every sample: never lost
    }
  }
}

// r06: the same binning with the slot counters of a TILE's neighbourhood in LDS.  The kernel above issues one device-scope
// atomic per list entry -- 1.2 M of them for a res3 block, ~40 G/s: 29 us.  A workgroup here takes the 4 x 8 output positions
// of a tile x K2 taps (one sample per thread); the corners of its samples fall, for all but extreme offsets, into a
// BW_H x BW_W window of input pixels around the tile's footprint: phase 1 counts them with LDS atomics (the returned value is
// the entry's rank among the workgroup's entries of that pixel), phase 2 reserves each touched pixel's slots with ONE global
// atomic (base = atomicAdd(cnt[pixel], local count)), phase 3 stores every entry at base + rank.  ~10 x fewer global
// atomics; corners outside the window take the direct path of the kernel above.  Slots beyond DG_CAP go to the overflow
// array exactly as there; the sort that follows makes the order of the slots irrelevant.
constexpr int BT_H = 4, BT_W = 8, BW_H = 24, BW_W = 32, BIN_THREADS = 320;  // <= 10 taps: one sample per thread
template <typename T>
__global__ __launch_bounds__(BIN_THREADS) void dcn_bin_samples_wg_kernel(DcnShape s, const T* __restrict__ offset,
                                                                       const T* __restrict__ mask, int* __restrict__ cnt,
                                                                       DgEntry* __restrict__ lists, int* __restrict__ ovf_cnt,
                                                                       DgOverflow* __restrict__ ovf, int tiles_y, int tiles_x,
                                                                       int fh, int fw) {
  __shared__ int lcnt[BW_H * BW_W];  // phase 1: entries per window pixel; phase 2: the slot base of the pixel
  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int b = tile / (tiles_y * tiles_x), tr = tile - b * (tiles_y * tiles_x);
  const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
  // window origin: the tile's footprint (fh x fw pixels at zero offsets) centred in the window
  const int oy = ty * BT_H * s.sh - s.ph - (BW_H - fh) / 2, ox = tx * BT_W * s.sw - s.pw - (BW_W - fw) / 2;
  for (int i = tid; i < BW_H * BW_W; i += BIN_THREADS) lcnt[i] = 0;
  const int pos = tid & (BT_H * BT_W - 1), tap = tid >> 5;
  const int ho = ty * BT_H + (pos >> 3), wo = tx * BT_W + (pos & 7);
  const bool live = tap < s.K2 && ho < s.Ho && wo < s.Wo;
  int widx[4], rank[4];
  uint32_t pixv[4];
  float wgt[4];
  bool on[4] = {false, false, false, false};
  long t = 0;
  if (live) {
    const int l = ho * s.Wo + wo;
    t = ((long)b * s.L + l) * s.K2 + tap;
    const int i = tap / s.kw, j = tap - i * s.kw;
    // the table of dcn_bwd_data_tc_kernel, operation for operation (deformable group 0)
    const long obase = (long)b * 2 * s.K2;
    const float off_h = to_f32(offset[(obase + 2 * tap) * s.L + l]);
    const float off_w = to_f32(offset[(obase + 2 * tap + 1) * s.L + l]);
    const float m = mask ? to_f32(mask[((long)b * s.K2 + tap) * s.L + l]) : 1.f;
    const float h_im = (float)(ho * s.sh - s.ph + i * s.dh) + off_h;
    const float w_im = (float)(wo * s.sw - s.pw + j * s.dw) + off_w;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)s.H && w_im < (float)s.W) {
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const int h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const int ys[4] = {h_low, h_low, h_high, h_high}, xs[4] = {w_low, w_high, w_low, w_high};
      const float ws[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
      for (int c = 0; c < 4; c++) {
        wgt[c] = ws[c] * m;
        on[c] = !(ys[c] < 0 || ys[c] > s.H - 1 || xs[c] < 0 || xs[c] > s.W - 1) && wgt[c] != 0.f;
        pixv[c] = (uint32_t)(((long)b * s.H + ys[c]) * s.W + xs[c]);
        const int wy = ys[c] - oy, wx = xs[c] - ox;
        widx[c] = (wy >= 0 && wy < BW_H && wx >= 0 && wx < BW_W) ? wy * BW_W + wx : -1;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; c++) {
    rank[c] = 0;
    if (on[c] && widx[c] >= 0) rank[c] = atomicAdd(&lcnt[widx[c]], 1);
  }
  __syncthreads();
  for (int i = tid; i < BW_H * BW_W; i += BIN_THREADS) {
    const int c = lcnt[i];
    if (c > 0) {
      const int py = oy + i / BW_W, px = ox + (i & (BW_W - 1));  // (inside the image: only such corners were counted)
      lcnt[i] = atomicAdd(&cnt[((long)b * s.H + py) * s.W + px], c);
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; c++) {
    if (!on[c]) continue;
    const int slot = widx[c] >= 0 ? lcnt[widx[c]] + rank[c] : atomicAdd(&cnt[pixv[c]], 1);
    if (slot < DG_CAP) {
      lists[(long)pixv[c] * DG_CAP + slot] = DgEntry{(uint32_t)t, wgt[c]};
    } else {
      const int o = atomicAdd(ovf_cnt, 1);
      ovf[o] = DgOverflow{pixv[c], (uint32_t)t, wgt[c], 0u};
    }
  }
}

// One wave per input pixel sorts its list by sample id, in place (all loads precede all stores): the gather below then adds a pixel's contributions in
// a fixed order whatever order the binning's atomics handed the slots out in (deterministic dX).  r05: a launch of its own
// behind the binning -- on the side stream when there is one -- instead of the first 60 % of the gather kernel's
// instructions on the critical path.  Lane i holds elements i and i + 64 of the <= 128 (sample, weight) pairs.
__global__ __launch_bounds__(256) void dcn_sort_lists_kernel(int npix, const int* __restrict__ cnt, DgEntry* __restrict__ lists) {
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;  // uniform per wave
  const int n = min(cnt[pix], DG_CAP);
  if (n < 2) return;
  DgEntry* lp = lists + pix * DG_CAP;
  uint32_t k0 = 0xffffffffu, k1 = 0xffffffffu;
  float w0 = 0.f, w1 = 0.f;
  if (lane < n) { const DgEntry e = lp[lane]; k0 = e.sample; w0 = e.w; }
  if (lane + 64 < n) { const DgEntry e = lp[lane + 64]; k1 = e.sample; w1 = e.w; }
  // rank of an element = the number of smaller keys (sample ids are distinct within a pixel's list): every key is
  // broadcast once (v_readlane with a uniform index: SGPR operand of the compare, no LDS, no dependent chain -- the
  // bitonic network this replaces was 21-28 stages of two ds_bpermute each), then the element is stored at its rank
  int r0 = 0, r1 = 0;
  const int n0 = min(n, 64);
  if (n <= 64) {  // (r06, the usual case -- 36 entries per pixel on average: one key per lane, three instructions per step instead of five)
    for (int j = 0; j < n; j++) {
      const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)k0, j);
      r0 += kj < k0;
    }
    if (lane < n) lp[r0] = DgEntry{k0, w0};
    return;
  }
  for (int j = 0; j < n0; j++) {
    const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)k0, j);
    r0 += kj < k0;
    r1 += kj < k1;
  }
  for (int j = 64; j < n; j++) {
    const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)k1, j - 64);
    r0 += kj < k0;
    r1 += kj < k1;
  }
  if (lane < n) lp[r0] = DgEntry{k0, w0};
  if (lane + 64 < n) lp[r1] = DgEntry{k1, w1};
}

// RL = lanes per column row = C / 8: every lane loads 16 B (8 channels), a wave instruction covers EPW = 64 / RL list
// entries (C = 128: four rows of 256 B per load).  Lane group g = lane / RL takes entries g, g + EPW, ... of the SORTED
// list -- read straight from memory, one 8-B load per group and entry, UN entries and their rows in flight per group --
// and the groups' partial sums are added in a fixed order at the end.
template <typename T, int RL>
__global__ __launch_bounds__(256) void dcn_gather_dx_kernel(int npix, int C, const int* __restrict__ cnt,
                                                           const DgEntry* __restrict__ lists,
                                                           const int* __restrict__ ovf_cnt,
                                                           const DgOverflow* __restrict__ ovf,
                                                           const T* __restrict__ col, T* __restrict__ gx) {
  constexpr int EPW = 64 / RL;  // entries per wave instruction
  constexpr int UN = 4;         // rows in flight per lane group
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;  // uniform per wave
  const int n = min(cnt[pix], DG_CAP);
  const DgEntry* lp = lists + pix * DG_CAP;
  const int grp = lane / RL, sub = lane - grp * RL;
  float acc[8];
#pragma unroll
  for (int v = 0; v < 8; v++) acc[v] = 0.f;
  const T* cbase = col + (long)sub * 8;
  for (int e0 = 0; e0 < n; e0 += EPW * UN) {
    DgEntry en[UN];
    raw16 q[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) en[u] = lp[min(e0 + u * EPW + grp, n - 1)];
#pragma unroll
    for (int u = 0; u < UN; u++) q[u] = *reinterpret_cast<const raw16*>(cbase + (long)en[u].sample * C);
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const float wv = (e0 + u * EPW + grp < n) ? en[u].w : 0.f;
      float f[8];
      tc_unpack(q[u], f, T{});
#pragma unroll
      for (int v = 0; v < 8; v++) acc[v] += wv * f[v];
    }
  }
  const int novf = *ovf_cnt;
  if (novf > 0) {  // uniform: some pixel collected more than DG_CAP entries (order of these additions: as stored)
    for (int o0 = 0; o0 < novf; o0 += 64) {
      const int o = o0 + lane;
      DgOverflow e{0xffffffffu, 0u, 0.f, 0u};
      if (o < novf) e = ovf[o];
      unsigned long long hit = __ballot(e.pix == (uint32_t)pix);
      while (hit) {
        const int src = __builtin_ctzll(hit);
        hit &= hit - 1;
        const uint32_t sk = (uint32_t)__shfl((int)e.sample, src);
        const float sw = __shfl(e.w, src);
        if (grp == 0) {
          float f[8];
          tc_unpack(*reinterpret_cast<const raw16*>(cbase + (long)sk * C), f, T{});
#pragma unroll
          for (int v = 0; v < 8; v++) acc[v] += sw * f[v];
        }
      }
    }
  }
  // the lane groups' partial sums, added in a fixed order (group 0 + 1, then + the pair above, ...)
#pragma unroll
  for (int d = RL; d < 64; d <<= 1) {
#pragma unroll
    for (int v = 0; v < 8; v++) acc[v] += __shfl_xor(acc[v], d);
  }
  if (grp == 0) {
    uint32_t wv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) wv[u] = (uint32_t)from_f32<T>(acc[2 * u]).v | ((uint32_t)from_f32<T>(acc[2 * u + 1]).v << 16);
    *reinterpret_cast<raw16*>(gx + pix * C + (long)sub * 8) = raw16{wv[0], wv[1], wv[2], wv[3]};
  }
}

template <typename T>
int dcn_tc_backward_data_gather(const DcnShape& s, const TcBwPlan& pl, const void* x_nhwc, const void* offset,
                                const void* mask, const void* weight, const void* gout_nhwc, void* gx_t, float* goff,
                                float* gmask, void* wp, const DcnGatherWs& gw, hipStream_t st, const DcnSide* side,
                                const ColPathPlan* cp, void* goff_t, void* gmask_t, const void* wt_kept) {
  const long npix = (long)s.B * s.H * s.W;
  bool goff_zeroed = false;
  {  // per-call state: pixel counters + the overflow counter (one region) -- and, when the tiles of a pixel are split
    // over channel groups (csplit > 1: atomics), the fp32 accumulators of the offset / mask gradient, which the
    // workspace keeps right in front of the counters: ONE zero launch instead of three
    char* z0 = (char*)gw.cnt;
    const char* z1 = (const char*)gw.cnt + (size_t)(npix + 1) * 4;
    bool merged = false;
    if (pl.csplit > 1 && goff && gmask && (char*)goff < (char*)gmask && (char*)gmask < z0 &&
        (size_t)(z0 - (char*)goff) <= (size_t)s.B * s.DG * 3 * s.K2 * s.L * 4 + 1024) {
      z0 = (char*)goff;
      merged = true;
    }
    const int zrc = zero_async(z0, (size_t)(z1 - z0), st);
    if (zrc) return zrc;
    if (merged) { goff_zeroed = true; }
  }
  int* ovf_cnt = gw.cnt + npix;
  const long nsamp = (long)s.P * s.K2;
  // the binning (behind the zero fill of its counters) -- on the side stream when there is one: only the column gather
  // at the end of this function reads the lists
  hipStream_t bst = st;
  // Every exit from here on -- an error return included -- leaves the caller's stream WAITING for whatever was enqueued
  // on the side stream (an unjoined fork would let the side stream read dY / the column / the workspace
  // after Python frees them, and leaves a stream capture with a dangling branch).  The success path joins in the
  // caller (deform_conv.hip: bwd_host) once the main-stream work is enqueued too.
  struct SideJoin {
    const DcnSide* sd; hipStream_t st; bool armed = false, ok = false;
    ~SideJoin() {
      if (!sd || !armed || ok) return;
      (void)hipEventRecord(sd->join, sd->stream);
      (void)hipStreamWaitEvent(st, sd->join, 0);
    }
  } side_join{side, st};
  if (side) {
    D2_HIP_OK(hipEventRecord(side->bin, st));  // (the fork point)
    D2_HIP_OK(hipStreamWaitEvent(side->stream, side->bin, 0));
    side_join.armed = true;
    if (side->fork) bst = side->stream;
  }
  {
    // the tile kernel: <= 10 taps (one sample per thread) and a footprint that leaves a margin in the LDS window;
    // D2AMD_DCN_BIN_DIRECT (profiling builds): one device-scope atomic per entry as in r02-r05
    const int fh = (BT_H - 1) * s.sh + (s.kh - 1) * s.dh + 2, fw = (BT_W - 1) * s.sw + (s.kw - 1) * s.dw + 2;
    const long tiles = (long)s.B * cdiv(s.Ho, BT_H) * cdiv(s.Wo, BT_W);
    if (s.K2 * BT_H * BT_W <= BIN_THREADS && fh + 4 <= BW_H && fw + 4 <= BW_W && tiles < (1l << 31) &&
        d2_prof_env("D2AMD_DCN_BIN_DIRECT") == nullptr)
      hipLaunchKernelGGL((dcn_bin_samples_wg_kernel<T>), dim3((unsigned)tiles), dim3(BIN_THREADS), 0, bst, s, (const T*)offset,
                         (const T*)mask, gw.cnt, (DgEntry*)gw.lists, ovf_cnt, (DgOverflow*)gw.ovf, cdiv(s.Ho, BT_H),
                         cdiv(s.Wo, BT_W), fh, fw);
    else
      hipLaunchKernelGGL((dcn_bin_samples_kernel<T>), dim3(cdiv(nsamp, 256)), dim3(256), 0, bst, s, (const T*)offset,
                         (const T*)mask, gw.cnt, (DgEntry*)gw.lists, ovf_cnt, (DgOverflow*)gw.ovf);
  }
  D2_LAUNCH_OK();
  if (gx_t) {  // the lists in sample order, for the gather at the end
    hipLaunchKernelGGL(dcn_sort_lists_kernel, dim3((unsigned)cdiv(npix, 4)), dim3(256), 0, bst, (int)npix, gw.cnt, (DgEntry*)gw.lists);
    D2_LAUNCH_OK();
  }
  if (side) {
    D2_HIP_OK(hipEventRecord(side->bin, side->stream));
    if (side->work) { const int wrc = side->work(side->ctx, side->stream); if (wrc) return wrc; }
    D2_HIP_OK(hipEventRecord(side->join, side->stream));
  }
  if (cp) {
    // r05: dcol = dY Wt^T on the dense GEMM, then d(offset) / d(mask) from dcol and x's corners, written in the I/O dtype
    const int crc = dcn_colpath_backward_data<T>(s, *cp, x_nhwc, offset, mask, weight, gout_nhwc, gw.col, wp, wt_kept, goff_t, gmask_t, st);
    if (crc) return crc;
  } else {
    {
      const long groups16 = (long)s.G * s.K2 * (s.Cg / 64) * 2 * (s.Cog / 16) * 64;
      const int blocks = cdiv(groups16, 256) > 8192 ? 8192 : cdiv(groups16, 256);
      hipLaunchKernelGGL((tc_pack_weight_t_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)weight, (T*)wp, s.G,
                         s.Cog, s.Cg, s.K2);
      D2_LAUNCH_OK();
    }
    BwArgs a{};
    a.x = x_nhwc; a.offset = offset; a.mask = mask; a.wp = wp; a.gout = gout_nhwc;
    a.gx = nullptr; a.goff = goff; a.gmask = gmask; a.dcol = gw.col;
    { const char* e = d2_prof_env("D2AMD_DCN_ABLATE_BWD"); a.ablate = e ? atoi(e) : 0; }
    a.tiles_y = pl.tiles_y; a.tiles_x = pl.tiles_x; a.csplit = pl.csplit;
    if (pl.csplit > 1 && !goff_zeroed) {
      if (goff) { const int zrc = zero_async(goff, (size_t)s.B * s.DG * 2 * s.K2 * s.L * 4, st); if (zrc) return zrc; }
      if (gmask) { const int zrc = zero_async(gmask, (size_t)s.B * s.DG * s.K2 * s.L * 4, st); if (zrc) return zrc; }
    }
    const long total = (long)s.B * pl.tiles_y * pl.tiles_x * s.DG * pl.csplit;
    D2_CHECK_ARG(total < (1l << 30), "deform_conv: too many tiles");
    a.total = (int)total;
    const bool timed = timing_begin("dcn_bwd_data", st);
    { const int lrc = launch_bwd_data_tc<T>(s, a, (a.total + 7) / 8 * 8, pl.lds, st); if (lrc) return lrc; }
    if (timed) timing_end("dcn_bwd_data", st);
    D2_LAUNCH_OK();
  }
  if (side) D2_HIP_OK(hipStreamWaitEvent(st, side->bin, 0));  // (also when nothing gathers: the counters are reused)
  if (gx_t) {
    const bool timed2 = timing_begin("dcn_bwd_gather", st);
    const dim3 grid((unsigned)cdiv(npix, 4));
    switch (s.C / 64) {
      case 1: hipLaunchKernelGGL((dcn_gather_dx_kernel<T, 8>), grid, dim3(256), 0, st, (int)npix, s.C, gw.cnt, (const DgEntry*)gw.lists, ovf_cnt, (const DgOverflow*)gw.ovf, (const T*)gw.col, (T*)gx_t); break;
      case 2: hipLaunchKernelGGL((dcn_gather_dx_kernel<T, 16>), grid, dim3(256), 0, st, (int)npix, s.C, gw.cnt, (const DgEntry*)gw.lists, ovf_cnt, (const DgOverflow*)gw.ovf, (const T*)gw.col, (T*)gx_t); break;
      case 4: hipLaunchKernelGGL((dcn_gather_dx_kernel<T, 32>), grid, dim3(256), 0, st, (int)npix, s.C, gw.cnt, (const DgEntry*)gw.lists, ovf_cnt, (const DgOverflow*)gw.ovf, (const T*)gw.col, (T*)gx_t); break;
      case 8: hipLaunchKernelGGL((dcn_gather_dx_kernel<T, 64>), grid, dim3(256), 0, st, (int)npix, s.C, gw.cnt, (const DgEntry*)gw.lists, ovf_cnt, (const DgOverflow*)gw.ovf, (const T*)gw.col, (T*)gx_t); break;
      default: set_error("deform_conv: column gather needs C in {64, 128, 256, 512}"); return D2AMD_EUNSUPPORTED;
    }
    if (timed2) timing_end("dcn_bwd_gather", st);
    D2_LAUNCH_OK();
  }
  side_join.ok = true;
  return D2AMD_OK;
}
template int dcn_tc_backward_data_gather<bf16_t>(const DcnShape&, const TcBwPlan&, const void*, const void*, const void*,
                                                 const void*, const void*, void*, float*, float*, void*,
                                                 const DcnGatherWs&, hipStream_t, const DcnSide*, const ColPathPlan*, void*, void*, const void*);
template int dcn_tc_backward_data_gather<f16_t>(const DcnShape&, const TcBwPlan&, const void*, const void*, const void*,
                                                const void*, const void*, void*, float*, float*, void*,
                                                const DcnGatherWs&, hipStream_t, const DcnSide*, const ColPathPlan*, void*, void*, const void*);

template <typename T>
int dcn_tc_backward_data(const DcnShape& s, const TcBwPlan& pl, const void* x_nhwc, const void* offset,
                         const void* mask, const void* weight, const void* gout_nhwc, float* gx, float* goff,
                         float* gmask, void* wp, hipStream_t st) {
  {
    const long groups16 = (long)s.G * s.K2 * (s.Cg / 64) * 2 * (s.Cog / 16) * 64;
    const int blocks = cdiv(groups16, 256) > 8192 ? 8192 : cdiv(groups16, 256);
    hipLaunchKernelGGL((tc_pack_weight_t_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)weight, (T*)wp, s.G,
                       s.Cog, s.Cg, s.K2);
    D2_LAUNCH_OK();
  }
  if (pl.R >= 0) {  // LDS patch + ownership variant
    BwpArgs a{};
    a.x = x_nhwc; a.offset = offset; a.mask = mask; a.wp = wp; a.gout = gout_nhwc;
    a.gx = gx; a.goff = goff; a.gmask = gmask;
    a.tiles_y = pl.tiles_y; a.tiles_x = pl.tiles_x; a.csplit = pl.csplit_patch;
    a.R = pl.R; a.PHt = pl.PHt; a.PWt = pl.PWt;
    if (a.csplit > 1) {
      if (goff) { const int zrc = zero_async(goff, (size_t)s.B * s.DG * 2 * s.K2 * s.L * 4, st); if (zrc) return zrc; }
      if (gmask) { const int zrc = zero_async(gmask, (size_t)s.B * s.DG * s.K2 * s.L * 4, st); if (zrc) return zrc; }
    }
    const long total = (long)s.B * pl.tiles_y * pl.tiles_x * s.DG * a.csplit;
    D2_CHECK_ARG(total < (1l << 30), "deform_conv: too many tiles");
    a.total = (int)total;
    auto kern = dcn_bwd_data_patch_kernel<T>;
    if (pl.lds_patch > 40 * 1024)
      D2_HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_patch));
    const bool timed = timing_begin("dcn_bwd_data", st);
    hipLaunchKernelGGL(kern, dim3((a.total + 7) / 8 * 8), dim3(256), pl.lds_patch, st, s, a);
    if (timed) timing_end("dcn_bwd_data", st);
    D2_LAUNCH_OK();
    return D2AMD_OK;
  }
  BwArgs a{};
  a.x = x_nhwc; a.offset = offset; a.mask = mask; a.wp = wp; a.gout = gout_nhwc;
  a.gx = gx; a.goff = goff; a.gmask = gmask;
  a.tiles_y = pl.tiles_y; a.tiles_x = pl.tiles_x; a.csplit = pl.csplit;
  { const char* e = d2_prof_env("D2AMD_DCN_ABLATE_BWD"); a.ablate = e ? atoi(e) : 0; }
  if (pl.csplit > 1) {  // channel shares accumulate d(offset) / d(mask) with atomics
    if (goff) { const int zrc = zero_async(goff, (size_t)s.B * s.DG * 2 * s.K2 * s.L * 4, st); if (zrc) return zrc; }
    if (gmask) { const int zrc = zero_async(gmask, (size_t)s.B * s.DG * s.K2 * s.L * 4, st); if (zrc) return zrc; }
  }
  const long total = (long)s.B * pl.tiles_y * pl.tiles_x * s.DG * pl.csplit;
  D2_CHECK_ARG(total < (1l << 30), "deform_conv: too many tiles");
  a.total = (int)total;
  const int grid = (a.total + 7) / 8 * 8;
  const bool timed = timing_begin("dcn_bwd_data", st);
  { const int lrc = launch_bwd_data_tc<T>(s, a, grid, pl.lds, st); if (lrc) return lrc; }
  if (timed) timing_end("dcn_bwd_data", st);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

template int dcn_tc_backward_data<bf16_t>(const DcnShape&, const TcBwPlan&, const void*, const void*, const void*,
                                          const void*, const void*, float*, float*, float*, void*, hipStream_t);
template int dcn_tc_backward_data<f16_t>(const DcnShape&, const TcBwPlan&, const void*, const void*, const void*,
                                         const void*, const void*, float*, float*, float*, void*, hipStream_t);

}  // namespace d2amd

namespace d2amd {

// =====================================================================================================
// Backward w.r.t. the weight, 16-bit path:  dW[co][ci][tap] = sum_p dY[co][p] * col[(tap, ci)][p].
// Replaces deform_conv_cuda.cu:642-824 / 1160-1221 (im2col into the HBM column buffer again, then one
// GEMM per image accumulating into grad_weight).
// The contraction runs over POSITIONS, so both MFMA operands need 8 consecutive positions per lane:
//   A  = dY, [co][position] in the reference's NCHW layout already: 16-B loads straight from global;
//   B  = the deformable column: a lane gathers (position, 8 channels) -- the transpose of what the
//        instruction wants -- so each wave turns its 16-position x 64-channel column tile through a
//        2.5 KB wave-private LDS buffer (8 x ds_write_b16 per item, 2 x ds_read_b64 per fragment; the
//        wave's own LDS operations are ordered, no barrier).
// One WAVE = 64 output channels x 64 input channels of one tap x a range of 16-position k-steps; waves
// are fully independent (no workgroup barrier).  All global loads of a k-step sit in straight-line code
// and run one k-step ahead in registers (clamped indices instead of branches: with loads inside
// data-dependent branches hipcc serialises the pipeline with s_waitcnt vmcnt(0)).  The bilinear table of a
// k-step is computed once per lane (its own position) and shared by the 64 channels.
// Result: fp32 atomics into the [g][tap][co][ci] staging buffer (4,096 per wave), unpacked by
// unpack_gw_kernel like the generic path.
struct BwwArgs {
  const void *x, *offset, *mask, *gout;  // x NHWC; gout NCHW [B][Co][L]
  float* gwr;                            // partial tiles [tile][chunk of 4 waves][64 co][64 ci] fp32 (see the epilogue)
  int n_cot, n_cit, pch, total, ksteps_per_image;
};

// AB: compile-time ablation for profiling (D2AMD_DCN_ABLATE_BWW selects the instantiation): 1 no gather loads,
// 2 no combine / LDS transpose, 4 no MFMA, 8 no dY loads, 16 no atomics, 32 no table loads
#ifndef D2AMD_BWW_DEPTH
#define D2AMD_BWW_DEPTH 2
#endif
constexpr int BWW_DEPTH = D2AMD_BWW_DEPTH;
template <typename T, int AB>
__global__ __launch_bounds__(256, 2) void dcn_bwd_weight_tc_kernel(DcnShape s, BwwArgs a) {
  typedef Mma<T> M;
  constexpr int TPITCH = 10;  // dwords per channel row of the transposition buffer: 16 positions x 2 B + 8 B pad
  __shared__ uint32_t tbuf_all[4][64 * TPITCH];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t* tbuf = tbuf_all[wid];
  const int per_xcd = (a.total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= a.total) return;
  // decode: position chunk fastest over the 4 waves of a workgroup, then (ci tile, co tile, tap, group)
  int r = logical * 4 + wid;
  const int pc = r % a.pch; r /= a.pch;
  const int cit = r % a.n_cit; r /= a.n_cit;
  const int cot = r % a.n_cot; r /= a.n_cot;
  const int tap = r % s.K2;
  const int g = r / s.K2;
  if (g >= s.G) return;
  const int cabs = g * s.Cg + cit * 64;     // first absolute input channel of this wave
  const int dgi = cabs / s.cpg;
  const int co0 = cot * 64;                 // first output channel inside the group
  const long nk = (long)s.B * a.ksteps_per_image;
  const int k_lo = (int)((long)pc * nk / a.pch), k_hi = (int)((long)(pc + 1) * nk / a.pch);

  const T* offset = (const T*)a.offset;
  const T* mask = (const T*)a.mask;
  const char* xb = (const char*)a.x;
  const T* gout = (const T*)a.gout;
  const uint32_t pix = (uint32_t)s.C * (uint32_t)sizeof(T);
  const int pos = lane & 15, cq = lane >> 4;        // gather role: position in the k-step, channel quarter
  const int n32 = lane & 31, khalf = lane >> 5;     // MFMA role
  const int ti = tap / s.kw, tj = tap - ti * s.kw;

  f32x16_t acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[m][n][q] = 0.f;

  struct Raw { float oh, ow, mk; };
  auto load_raw = [&](int k, Raw& rw) __attribute__((always_inline)) {
    k = min(k, (int)nk - 1);
    const int b = k / a.ksteps_per_image, l = min((k - b * a.ksteps_per_image) * 16 + pos, s.L - 1);
    const long obase = ((long)b * s.DG + dgi) * 2 * s.K2;
    if (AB & 32) { rw.oh = 0.3f; rw.ow = 0.4f; rw.mk = 0.5f; return; }
    rw.oh = to_f32(offset[(obase + 2 * tap) * s.L + l]);
    rw.ow = to_f32(offset[(obase + 2 * tap + 1) * s.L + l]);
    rw.mk = mask ? to_f32(mask[(((long)b * s.DG + dgi) * s.K2 + tap) * s.L + l]) : 1.f;
  };
  struct Ent { uint32_t off[4]; float w[4]; };
  auto build = [&](int k, const Raw& rw, Ent& e) __attribute__((always_inline)) {
    k = min(k, (int)nk - 1);
    const int b = k / a.ksteps_per_image, l = (k - b * a.ksteps_per_image) * 16 + pos;
#pragma unroll
    for (int t = 0; t < 4; t++) { e.off[t] = 0u; e.w[t] = 0.f; }
    const int lc = min(l, s.L - 1);
    const int ho = lc / s.Wo, wo = lc - ho * s.Wo;
    const float h_im = (float)(ho * s.sh - s.ph + ti * s.dh) + rw.oh;
    const float w_im = (float)(wo * s.sw - s.pw + tj * s.dw) + rw.ow;
    if (l < s.L && h_im > -1.f && w_im > -1.f && h_im < (float)s.H && w_im < (float)s.W) {
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const int h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
      const float hh = 1.f - lh, hw = 1.f - lw, m = rw.mk;
      const long rowbase = (long)b * s.H;
      if (h_low >= 0 && w_low >= 0) { e.off[0] = (uint32_t)((rowbase + h_low) * s.W + w_low) * pix; e.w[0] = hh * hw * m; }
      if (h_low >= 0 && w_high <= s.W - 1) { e.off[1] = (uint32_t)((rowbase + h_low) * s.W + w_high) * pix; e.w[1] = hh * lw * m; }
      if (h_high <= s.H - 1 && w_low >= 0) { e.off[2] = (uint32_t)((rowbase + h_high) * s.W + w_low) * pix; e.w[2] = lh * hw * m; }
      if (h_high <= s.H - 1 && w_high <= s.W - 1) { e.off[3] = (uint32_t)((rowbase + h_high) * s.W + w_high) * pix; e.w[3] = lh * lw * m; }
    }
  };
  // operands of k-step k: 2 gather items (this lane's position x channels (it*4 + cq)*8..) and 2 dY fragments
  auto issue = [&](int k, const Ent& e, raw16 (&gr)[2][4], raw16 (&ar)[2]) __attribute__((always_inline)) {
    k = min(k, (int)nk - 1);
    const int b = k / a.ksteps_per_image, l0 = (k - b * a.ksteps_per_image) * 16;
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const uint32_t cofs = (uint32_t)(cabs + (it * 4 + cq) * 8) * (uint32_t)sizeof(T);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        if (AB & 1) gr[it][c] = raw16{e.off[c], cofs, 0u, 0u};
        else gr[it][c] = *reinterpret_cast<const raw16*>(xb + (e.off[c] + cofs));
      }
    }
#pragma unroll
    for (int m = 0; m < 2; m++) {
      // 8 positions l0 + khalf*8 .. of output channel co0 + m*32 + n32.  The window is clamped to the row
      // ([L - 8, L) at the image tail) so that the load never leaves the row; compute() shifts it back
      const int co = min(co0 + m * 32 + n32, s.Cog - 1);
      const int lw = min(l0 + khalf * 8, s.L - 8);
      if (AB & 8) ar[m] = raw16{(uint32_t)lw, (uint32_t)co, 0u, 0u};
      else ar[m] = *reinterpret_cast<const raw16*>(gout + ((long)b * s.Co + (long)g * s.Cog + co) * s.L + lw);
    }
  };
  auto compute = [&](int k, const Ent& e, const raw16 (&gr)[2][4], const raw16 (&ar)[2]) __attribute__((always_inline)) {
    // combine -> transposed into the wave's LDS buffer: tbuf[channel][position].  The buffer is wave-private
    // and a wave's DS operations execute in order; the asm statements only stop the COMPILER from moving
    // the 16-bit stores across the 64-bit loads of the other type.
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2; it++) {
      if (AB & 2) { tbuf[lane] = gr[it][0].x ^ gr[it][1].x ^ gr[it][2].x ^ gr[it][3].x; continue; }
      float v[8];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        float f[8];
        tc_unpack(gr[it][c], f, T{});
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = c == 0 ? e.w[c] * f[u] : v[u] + e.w[c] * f[u];
      }
      const raw16 pk = tc_pack(v, T{});
      uint16_t* row = reinterpret_cast<uint16_t*>(tbuf) + ((it * 4 + cq) * 8) * (TPITCH * 2) + pos;
#pragma unroll
      for (int u = 0; u < 8; u++) row[u * (TPITCH * 2)] = (uint16_t)(u & 1 ? pk[u >> 1] >> 16 : pk[u >> 1] & 0xffffu);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // dY fragment: at the image tail the window was loaded from [L - 8, L); shift it down by `sh` elements
    // so that element j is position l + j again, zero-filling the positions past the row end
    const int kk = min(k, (int)nk - 1);
    const int lq = (kk - (kk / a.ksteps_per_image) * a.ksteps_per_image) * 16 + khalf * 8;
    const int sh = lq - min(lq, s.L - 8);  // 0 inside the row; >= 8: nothing valid
    raw16 af[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
      unsigned long long lo = (unsigned long long)ar[m].x | ((unsigned long long)ar[m].y << 32);
      unsigned long long hi = (unsigned long long)ar[m].z | ((unsigned long long)ar[m].w << 32);
      if (sh >= 8) { lo = 0ull; hi = 0ull; }
      else if (sh >= 4) { lo = sh == 4 ? hi : hi >> (16 * (sh - 4)); hi = 0ull; }
      else if (sh > 0) { lo = (lo >> (16 * sh)) | (hi << (64 - 16 * sh)); hi = hi >> (16 * sh); }
      af[m] = raw16{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
    }
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const uint32_t* rp = tbuf + (n * 32 + n32) * TPITCH + khalf * 4;
      const uint2 lo = *reinterpret_cast<const uint2*>(rp);
      const uint2 hi = *reinterpret_cast<const uint2*>(rp + 2);
      const raw16 bq = {lo.x, lo.y, hi.x, hi.y};
      const typename M::frag bfr = __builtin_bit_cast(typename M::frag, bq);
      if (AB & 4) {
#pragma unroll
        for (int m = 0; m < 2; m++) acc[m][n][0] += __uint_as_float(af[m].x ^ bq.x);
      } else {
#pragma unroll
        for (int m = 0; m < 2; m++) acc[m][n] = M::mma(__builtin_bit_cast(typename M::frag, af[m]), bfr, acc[m][n]);
      }
    }
    asm volatile("" ::: "memory");
  };

  // D k-steps in flight per wave: the gathers of k + 1 .. k + D - 1 are outstanding while k is combined and contracted,
  // the offset / mask rows are requested D steps ahead of the k-step that builds its table from them.  A wave is a chain
  // offsets -> addresses -> gather -> combine -> MFMA with two dependent memory round trips per k-step and only two waves
  // per SIMD to hide them (256 VGPRs each): with D = 2 (the first version) a k-step took ~3,300 cycles for ~600 cycles of
  // instruction issue.  Measured (round 3, D2AMD_BWW_DEPTH): D = 3 (256 VGPRs, no spill) 128.9 us against 127.5 us at
  // D = 2 over the 13 R50 blocks, D = 4 spills -- the outstanding loads are not what the k-step waits for; D stays 2.
  constexpr int D = BWW_DEPTH;
  if (k_lo < k_hi) {
    Raw rw[D];
    Ent en[D];
    raw16 gq[D][2][4], aq[D][2];
#pragma unroll
    for (int i = 0; i < D; i++) load_raw(k_lo + i, rw[i]);
#pragma unroll
    for (int i = 0; i < D - 1; i++) {
      build(k_lo + i, rw[i], en[i]);
      load_raw(k_lo + i + D, rw[i]);
      issue(k_lo + i, en[i], gq[i], aq[i]);
    }
    for (int k = k_lo; k < k_hi; k += D) {
#pragma unroll
      for (int u = 0; u < D; u++) {
        if (k + u >= k_hi) break;  // uniform
        const int sj = (u + D - 1) % D;  // slot of step k + u + D - 1 = the slot step k + u - 1 has just left
        build(k + u + D - 1, rw[sj], en[sj]);
        load_raw(k + u + 2 * D - 1, rw[sj]);
        issue(k + u + D - 1, en[sj], gq[sj], aq[sj]);
        compute(k + u, en[u], gq[u], aq[u]);
      }
    }
  }
  // ---- the 4 waves of the workgroup hold partial sums of the SAME 64 x 64 tile (4 consecutive position chunks: the
  // split over positions is what fills the chip -- 36 tiles for res3).  They are added up in LDS first and each wave
  // sends one quarter of the tile to memory: a quarter of the device-scope fp32 atomics (9.4 M per call before,
  // whatever the shape; the kernel time grows with the number of chunks: 92 / 127 / 210 / 395 us at 16 / 32 / 64 / 128
  // chunks for all blocks).  143 -> 128 us on average over the 13 blocks of R50.
  // (Also tried in r02: materialising the column once -- [b][tap][c][L], position-contiguous like dY -- and
  // contracting from it with 16-B loads only: column writer 55 us + contraction 80 us, L2-bandwidth bound at 32
  // flop/B per wave; not faster than gathering inside the contraction, removed.)
  __shared__ float red[4][4096];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int q = 0; q < 16; q++) red[wid][((m * 2 + n) * 16 + q) * 64 + lane] = acc[m][n][q];
  __syncthreads();
  // Round 3: the workgroup's sum goes to ITS OWN 64 x 64 slot of a partial-tile array with plain 128-B row stores and
  // unpack_gw_partials_kernel adds the slots of a tile in chunk order while it converts -- no device-scope fp32 atomics
  // (2.4 M per call, ~36 us of the kernel in the r01 ablation, more with more chunks), no zero fill of the staging
  // buffer, and a weight gradient that is bit-identical run to run.
  {
    const int m = wid >> 1, n = wid & 1;  // this wave's quarter: block (m, n) of the tile
    const long tile = (((long)g * s.K2 + tap) * a.n_cot + cot) * a.n_cit + cit;
    float* dst = a.gwr + (tile * (a.pch >> 2) + (pc >> 2)) * 4096;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int e = (wid * 16 + q) * 64 + lane;
      const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
      const int row = m * 32 + frag_row(q, lane);  // output channel inside the tile (rows past Cog are never read)
      if (AB & 16) { if (v == 123.456f) dst[0] = 1.f; }
      else dst[row * 64 + n * 32 + n32] = v;
    }
  }
}

// ---- cooperative form (r03): SH waves of a workgroup own SH OUTPUT-CHANNEL TILES of the same (tap, input-channel tile,
// position range) and share the deformable column.  In the kernel above every wave gathers the column tile of its
// k-step for itself -- the same [16 positions x 64 channels] tile is gathered Co / 64 times (2 / 4 / 8 for res3 / 4 / 5),
// and the ablation puts 63 + 20 + 13 of the kernel's 165 us (res3, r01) on the gather loads, the bilinear combine and
// the offset / mask table.  Here the k-steps of a range are dealt to the SH waves in batches of SH: wave j gathers
// k-step j of the batch into ITS LDS slot (the same transposed [channel][position] image), one barrier, then every
// wave contracts all SH k-steps of the batch with its own dY rows: one gather per SH x 4 MFMAs instead of per 4.
// SH = min(4, Co / 64 per group) (2: two gather groups per workgroup over different position ranges, summed in LDS).
template <typename T, int SH>
__global__ __launch_bounds__(256, 2) void dcn_bwd_weight_coop_kernel(DcnShape s, BwwArgs a) {
  typedef Mma<T> M;
  constexpr int NG = 4 / SH;   // gather groups per workgroup
  constexpr int TPITCH = 10;   // dwords per channel row of a transposed column tile: 16 positions x 2 B + 8 B pad
  __shared__ float red[4][4096];  // the epilogue's sum over the gather groups; the main loop's column tiles live in it
  uint32_t (*tb)[4][64 * TPITCH] = reinterpret_cast<uint32_t (*)[4][64 * TPITCH]>(&red[0][0]);  // [buffer][wave][channel row]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int gi = wid / SH, mem = wid % SH;
  const int per_xcd = (a.total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= a.total) return;  // (the whole workgroup)
  int r = logical;
  const int pcw = r % a.pch; r /= a.pch;
  const int cit = r % a.n_cit; r /= a.n_cit;
  const int ncg = a.n_cot / SH;
  const int cotg = r % ncg; r /= ncg;
  const int tap = r % s.K2;
  const int g = r / s.K2;
  if (g >= s.G) return;
  const int cot = cotg * SH + mem;
  const int cabs = g * s.Cg + cit * 64;  // first absolute input channel of the column tile
  const int dgi = cabs / s.cpg;
  const int co0 = cot * 64;              // first output channel (inside the group) of this wave's tile
  const long nk = (long)s.B * a.ksteps_per_image;
  const int nchunks = a.pch * NG, pc = pcw * NG + gi;
  const int k_lo = (int)((long)pc * nk / nchunks), k_hi = (int)((long)(pc + 1) * nk / nchunks);
  const int len_max = (int)((nk + nchunks - 1) / nchunks);
  const int nbatch = (len_max + SH - 1) / SH;  // the same for every group of the workgroup: one barrier per batch

  const T* offset = (const T*)a.offset;
  const T* mask = (const T*)a.mask;
  const char* xb = (const char*)a.x;
  const T* gout = (const T*)a.gout;
  const uint32_t pix = (uint32_t)s.C * (uint32_t)sizeof(T);
  const int pos = lane & 15, cq = lane >> 4;     // gather role: position in the k-step, channel quarter
  const int n32 = lane & 31, khalf = lane >> 5;  // MFMA role
  const int ti = tap / s.kw, tj = tap - ti * s.kw;

  f32x16_t acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[m][n][q] = 0.f;

  struct Raw { float oh, ow, mk; };
  auto load_raw = [&](int k, Raw& rw) __attribute__((always_inline)) {
    k = min(k, (int)nk - 1);
    const int b = k / a.ksteps_per_image, l = min((k - b * a.ksteps_per_image) * 16 + pos, s.L - 1);
    const long obase = ((long)b * s.DG + dgi) * 2 * s.K2;
    rw.oh = to_f32(offset[(obase + 2 * tap) * s.L + l]);
    rw.ow = to_f32(offset[(obase + 2 * tap + 1) * s.L + l]);
    rw.mk = mask ? to_f32(mask[(((long)b * s.DG + dgi) * s.K2 + tap) * s.L + l]) : 1.f;
  };
  struct Ent { uint32_t off[4]; float w[4]; };
  auto build = [&](int k, const Raw& rw, Ent& e) __attribute__((always_inline)) {
    k = min(k, (int)nk - 1);
    const int b = k / a.ksteps_per_image, l = (k - b * a.ksteps_per_image) * 16 + pos;
#pragma unroll
    for (int t = 0; t < 4; t++) { e.off[t] = 0u; e.w[t] = 0.f; }
    const int lc = min(l, s.L - 1);
    const int ho = lc / s.Wo, wo = lc - ho * s.Wo;
    const float h_im = (float)(ho * s.sh - s.ph + ti * s.dh) + rw.oh;
    const float w_im = (float)(wo * s.sw - s.pw + tj * s.dw) + rw.ow;
    if (l < s.L && h_im > -1.f && w_im > -1.f && h_im < (float)s.H && w_im < (float)s.W) {
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const int h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
      const float hh = 1.f - lh, hw = 1.f - lw, m = rw.mk;
      const long rowbase = (long)b * s.H;
      if (h_low >= 0 && w_low >= 0) { e.off[0] = (uint32_t)((rowbase + h_low) * s.W + w_low) * pix; e.w[0] = hh * hw * m; }
      if (h_low >= 0 && w_high <= s.W - 1) { e.off[1] = (uint32_t)((rowbase + h_low) * s.W + w_high) * pix; e.w[1] = hh * lw * m; }
      if (h_high <= s.H - 1 && w_low >= 0) { e.off[2] = (uint32_t)((rowbase + h_high) * s.W + w_low) * pix; e.w[2] = lh * hw * m; }
      if (h_high <= s.H - 1 && w_high <= s.W - 1) { e.off[3] = (uint32_t)((rowbase + h_high) * s.W + w_high) * pix; e.w[3] = lh * lw * m; }
    }
  };
  // the gather of a k-step: 2 items (this lane's position x channels (it * 4 + cq) * 8 ..) x 4 corners
  auto issue_gather = [&](const Ent& e, raw16 (&gr)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const uint32_t cofs = (uint32_t)(cabs + (it * 4 + cq) * 8) * (uint32_t)sizeof(T);
#pragma unroll
      for (int c = 0; c < 4; c++) gr[it][c] = *reinterpret_cast<const raw16*>(xb + (e.off[c] + cofs));
    }
  };
  // the dY fragments of k-step k for this wave's output-channel tile (window clamped to the row: see the kernel above)
  auto issue_a = [&](int k, raw16 (&ar)[2]) __attribute__((always_inline)) {
    k = min(k, (int)nk - 1);
    const int b = k / a.ksteps_per_image, l0 = (k - b * a.ksteps_per_image) * 16;
#pragma unroll
    for (int m = 0; m < 2; m++) {
      const int co = min(co0 + m * 32 + n32, s.Cog - 1);
      const int lw = min(l0 + khalf * 8, s.L - 8);
      ar[m] = *reinterpret_cast<const raw16*>(gout + ((long)b * s.Co + (long)g * s.Cog + co) * s.L + lw);
    }
  };
  // combine + transpose into this wave's slot: tbuf[channel][position]
  auto combine = [&](uint32_t* tbuf, const Ent& e, const raw16 (&gr)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 2; it++) {
      float v[8];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        float f[8];
        tc_unpack(gr[it][c], f, T{});
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = c == 0 ? e.w[c] * f[u] : v[u] + e.w[c] * f[u];
      }
      const raw16 pk = tc_pack(v, T{});
      uint16_t* row = reinterpret_cast<uint16_t*>(tbuf) + ((it * 4 + cq) * 8) * (TPITCH * 2) + pos;
#pragma unroll
      for (int u = 0; u < 8; u++) row[u * (TPITCH * 2)] = (uint16_t)(u & 1 ? pk[u >> 1] >> 16 : pk[u >> 1] & 0xffffu);
    }
  };
  auto mma_step = [&](const uint32_t* tbuf, int k, const raw16 (&ar)[2]) __attribute__((always_inline)) {
    const int kk = min(k, (int)nk - 1);
    const int lq = (kk - (kk / a.ksteps_per_image) * a.ksteps_per_image) * 16 + khalf * 8;
    const int sh = lq - min(lq, s.L - 8);  // 0 inside the row; >= 8: nothing valid
    raw16 af[2] = {ar[0], ar[1]};
    if (__builtin_amdgcn_ballot_w64(sh != 0) != 0ull) {  // (uniform: only the last k-step of an image row range shifts)
#pragma unroll
      for (int m = 0; m < 2; m++) {
        unsigned long long lo = (unsigned long long)ar[m].x | ((unsigned long long)ar[m].y << 32);
        unsigned long long hi = (unsigned long long)ar[m].z | ((unsigned long long)ar[m].w << 32);
        if (sh >= 8) { lo = 0ull; hi = 0ull; }
        else if (sh >= 4) { lo = sh == 4 ? hi : hi >> (16 * (sh - 4)); hi = 0ull; }
        else if (sh > 0) { lo = (lo >> (16 * sh)) | (hi << (64 - 16 * sh)); hi = hi >> (16 * sh); }
        af[m] = raw16{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
      }
    }
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const uint32_t* rp = tbuf + (n * 32 + n32) * TPITCH + khalf * 4;
      const uint2 lo = *reinterpret_cast<const uint2*>(rp);
      const uint2 hi = *reinterpret_cast<const uint2*>(rp + 2);
      const raw16 bq = {lo.x, lo.y, hi.x, hi.y};
      const typename M::frag bfr = __builtin_bit_cast(typename M::frag, bq);
#pragma unroll
      for (int m = 0; m < 2; m++) acc[m][n] = M::mma(__builtin_bit_cast(typename M::frag, af[m]), bfr, acc[m][n]);
    }
  };

  // batch b: k-steps k_lo + b * SH + j, j < SH; this wave gathers j = mem
  Raw rw;
  Ent en;
  raw16 gq[2][4], aq[SH][2];
  load_raw(k_lo + mem, rw);
  build(k_lo + mem, rw, en);
  issue_gather(en, gq);
  load_raw(k_lo + SH + mem, rw);
#pragma unroll
  for (int j = 0; j < SH; j++) issue_a(k_lo + j, aq[j]);
  for (int b = 0; b < nbatch; b++) {
    const int kb = k_lo + b * SH;
    uint32_t* mine = tb[b & 1][wid];
    if (kb + mem < k_hi) combine(mine, en, gq);  // (uniform per wave; a k-step past the range is skipped by everyone)
    // the gather of the next batch goes out before this one is contracted
    build(kb + SH + mem, rw, en);
    issue_gather(en, gq);
    load_raw(kb + 2 * SH + mem, rw);
    __syncthreads();  // the batch's SH column tiles are in LDS (buffer b & 1; b + 1 writes the other one)
#pragma unroll
    for (int j = 0; j < SH; j++)
      if (kb + j < k_hi) mma_step(tb[b & 1][gi * SH + j], kb + j, aq[j]);  // uniform
#pragma unroll
    for (int j = 0; j < SH; j++) issue_a(kb + SH + j, aq[j]);
  }
  // ---- partial tile of (this output-channel tile, this workgroup's position range): the NG gather groups hold the same
  // tiles over different ranges and are summed through LDS; plain row stores, the unpack kernel adds the workgroups'
  // slots of a tile in order
  __syncthreads();  // (the column tiles share the LDS of `red`: everyone is done reading them)
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int q = 0; q < 16; q++) red[wid][((m * 2 + n) * 16 + q) * 64 + lane] = acc[m][n][q];
  __syncthreads();
  if (wid < SH) {
    const long tile = (((long)g * s.K2 + tap) * a.n_cot + (cotg * SH + wid)) * a.n_cit + cit;
    float* dst = a.gwr + (tile * a.pch + pcw) * 4096;
#pragma unroll
    for (int mn = 0; mn < 4; mn++)
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int e = (mn * 16 + q) * 64 + lane;
        float v = red[wid][e];
        if (NG == 2) v += red[SH + wid][e];
        dst[((mn >> 1) * 32 + frag_row(q, lane)) * 64 + (mn & 1) * 32 + n32] = v;
      }
  }
}

// grad_weight (Co, Cg, K2) T = the partial tiles of dcn_bwd_weight_tc_kernel summed in chunk order.  One thread per
// (tile, row, column): the partials are read as coalesced rows.
template <typename T>
__global__ __launch_bounds__(256) void unpack_gw_partials_kernel(const float* __restrict__ part, T* __restrict__ gw, int G,
                                                                int Cog, int Cg, int K2, int n_cot, int n_cit, int nch4) {
  const long n = (long)G * K2 * n_cot * n_cit * 4096;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i & 63), row = (int)((i >> 6) & 63);
    long t = i >> 12;
    const int cit = (int)(t % n_cit); t /= n_cit;
    const int cot = (int)(t % n_cot); t /= n_cot;
    const int tap = (int)(t % K2);
    const int g = (int)(t / K2);
    const int co = cot * 64 + row, ci = cit * 64 + col;
    if (co >= Cog || ci >= Cg) continue;
    const float* p = part + (i >> 12) * (long)nch4 * 4096 + (i & 4095);
    float v = 0.f;
    for (int c = 0; c < nch4; c++) v += p[(long)c * 4096];
    gw[(((long)g * Cog + co) * Cg + ci) * K2 + tap] = from_f32<T>(v);
  }
}

TcBwwPlan dcn_tc_plan_bww(const DcnShape& s, int dtype) {
  TcBwwPlan pl{};
  pl.ok = false;
  if (getenv("D2AMD_DCN_V1") || d2_prof_env("D2AMD_DCN_BWW_V1")) return pl;
  if (dtype != D2AMD_BF16 && dtype != D2AMD_F16) return pl;
  if (s.Cg % 64 != 0 || s.cpg % 64 != 0 || s.P <= 0 || s.L < 8) return pl;
  if ((long)s.B * s.H * s.W * s.C * 2 >= (1l << 32)) return pl;
  pl.n_cot = cdiv(s.Cog, 64);
  pl.n_cit = s.Cg / 64;
  pl.ksteps_per_image = cdiv(s.L, 16);
  const long tiles = (long)s.G * s.K2 * pl.n_cot * pl.n_cit;
  const long nk = (long)s.B * pl.ksteps_per_image;
  long pch = (2304 + tiles - 1) / tiles;      // ~2 waves per SIMD in total (measured: profiles/r01/v5_dcn_bww_sweep.txt)
  if (pch < 8) pch = 8;
  if (pch > nk / 4) pch = nk / 4 > 0 ? nk / 4 : 1;  // at least 4 k-steps per wave
  pch = (pch + 3) / 4 * 4;                     // the 4 waves of a workgroup take consecutive chunks
  const char* e = getenv("D2AMD_DCN_BWW_PCH");  // profiling switch
  if (e && atoi(e) > 0) pch = (atoi(e) + 3) / 4 * 4;
  pl.pch = (int)pch;
  pl.partial_bytes = (size_t)tiles * (pch / 4) * 4096 * sizeof(float);
  // cooperative kernel: SH output-channel tiles per workgroup share the column gather (A/B: D2AMD_DCN_BWW_COOP=0)
  pl.share = 0;
  static const bool no_coop = d2_prof_env("D2AMD_DCN_BWW_COOP") && atoi(d2_prof_env("D2AMD_DCN_BWW_COOP")) == 0;
  if (!no_coop && pl.n_cot % 2 == 0) {
    pl.share = pl.n_cot % 4 == 0 ? 4 : 2;
    const int ng = 4 / pl.share;
    const long wg_tiles = tiles / pl.share;             // (tap, ci tile, group of SH co tiles)
    // ~2 workgroups per CU (share 2: ~4 -- measured per stage, res3 106.5 us at 64 chunks against 114.8 at 32)
    long pchw = (576 * ng + wg_tiles - 1) / wg_tiles;
    if (pchw * ng > nk / 4) pchw = nk / 4 / ng > 0 ? nk / 4 / ng : 1;  // at least 4 k-steps per gather group
    if (e && atoi(e) > 0) pchw = atoi(e);
    pl.pchw = (int)pchw;
    pl.partial_bytes = (size_t)tiles * pchw * 4096 * sizeof(float);
  }
  pl.ok = true;
  return pl;
}

template <typename T>
int dcn_tc_backward_weight(const DcnShape& s, const TcBwwPlan& pl, const void* x_nhwc, const void* offset,
                           const void* mask, const void* gout_nchw, float* gwr, void* grad_weight, hipStream_t st) {
  BwwArgs a{};
  a.x = x_nhwc; a.offset = offset; a.mask = mask; a.gout = gout_nchw; a.gwr = gwr;
  a.n_cot = pl.n_cot; a.n_cit = pl.n_cit; a.pch = pl.pch; a.ksteps_per_image = pl.ksteps_per_image;
  if (pl.share) {
    a.pch = pl.pchw;
    const long wgs = (long)s.G * s.K2 * (pl.n_cot / pl.share) * pl.n_cit * pl.pchw;
    D2_CHECK_ARG(wgs < (1l << 30), "deform_conv: too many tiles");
    a.total = (int)wgs;
    const int cgrid = (a.total + 7) / 8 * 8;
    const bool timed_c = timing_begin("dcn_bwd_weight", st);
    if (pl.share == 4) hipLaunchKernelGGL((dcn_bwd_weight_coop_kernel<T, 4>), dim3(cgrid), dim3(256), 0, st, s, a);
    else hipLaunchKernelGGL((dcn_bwd_weight_coop_kernel<T, 2>), dim3(cgrid), dim3(256), 0, st, s, a);
    if (timed_c) timing_end("dcn_bwd_weight", st);
    D2_LAUNCH_OK();
    const long n2 = (long)s.G * s.K2 * pl.n_cot * pl.n_cit * 4096;
    hipLaunchKernelGGL((unpack_gw_partials_kernel<T>), dim3(cdiv(n2, 256) > 8192 ? 8192 : cdiv(n2, 256)), dim3(256), 0, st,
                       (const float*)gwr, (T*)grad_weight, s.G, s.Cog, s.Cg, s.K2, pl.n_cot, pl.n_cit, pl.pchw);
    D2_LAUNCH_OK();
    return D2AMD_OK;
  }
  const long waves = (long)s.G * s.K2 * pl.n_cot * pl.n_cit * pl.pch;
  D2_CHECK_ARG(waves / 4 < (1l << 30), "deform_conv: too many tiles");
  a.total = (int)(waves / 4);
  const int grid = (a.total + 7) / 8 * 8;
  const char* ab = d2_prof_env("D2AMD_DCN_ABLATE_BWW");  // profiling only
  const bool timed = timing_begin("dcn_bwd_weight", st);
  switch (ab ? atoi(ab) : 0) {
#ifdef D2AMD_DCN_ABLATION_BUILD
    case 1: hipLaunchKernelGGL((dcn_bwd_weight_tc_kernel<T, 1>), dim3(grid), dim3(256), 0, st, s, a); break;
    case 2: hipLaunchKernelGGL((dcn_bwd_weight_tc_kernel<T, 2>), dim3(grid), dim3(256), 0, st, s, a); break;
    case 4: hipLaunchKernelGGL((dcn_bwd_weight_tc_kernel<T, 4>), dim3(grid), dim3(256), 0, st, s, a); break;
    case 8: hipLaunchKernelGGL((dcn_bwd_weight_tc_kernel<T, 8>), dim3(grid), dim3(256), 0, st, s, a); break;
    case 16: hipLaunchKernelGGL((dcn_bwd_weight_tc_kernel<T, 16>), dim3(grid), dim3(256), 0, st, s, a); break;
    case 32: hipLaunchKernelGGL((dcn_bwd_weight_tc_kernel<T, 32>), dim3(grid), dim3(256), 0, st, s, a); break;
    case 41: hipLaunchKernelGGL((dcn_bwd_weight_tc_kernel<T, 41>), dim3(grid), dim3(256), 0, st, s, a); break;
    case 47: hipLaunchKernelGGL((dcn_bwd_weight_tc_kernel<T, 47>), dim3(grid), dim3(256), 0, st, s, a); break;
    case 63: hipLaunchKernelGGL((dcn_bwd_weight_tc_kernel<T, 63>), dim3(grid), dim3(256), 0, st, s, a); break;
#endif
    default: hipLaunchKernelGGL((dcn_bwd_weight_tc_kernel<T, 0>), dim3(grid), dim3(256), 0, st, s, a);
  }
  if (timed) timing_end("dcn_bwd_weight", st);
  D2_LAUNCH_OK();
  const long n = (long)s.G * s.K2 * pl.n_cot * pl.n_cit * 4096;
  hipLaunchKernelGGL((unpack_gw_partials_kernel<T>), dim3(cdiv(n, 256) > 8192 ? 8192 : cdiv(n, 256)), dim3(256), 0, st,
                     (const float*)gwr, (T*)grad_weight, s.G, s.Cog, s.Cg, s.K2, pl.n_cot, pl.n_cit, pl.pch / 4);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

template int dcn_tc_backward_weight<bf16_t>(const DcnShape&, const TcBwwPlan&, const void*, const void*, const void*,
                                            const void*, float*, void*, hipStream_t);
template int dcn_tc_backward_weight<f16_t>(const DcnShape&, const TcBwwPlan&, const void*, const void*, const void*,
                                           const void*, float*, void*, hipStream_t);

}  // namespace d2amd
