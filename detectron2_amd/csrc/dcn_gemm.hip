// Host side of the dense NT GEMM (dcn_gemm.h): plan (tile shape, split) and launch.
#include "dcn_gemm.h"

#include <mutex>

namespace d2amd {

GemmNtPlan gemm_nt_plan(int M, int N, int K) {
  GemmNtPlan pl{};
  pl.ok = false;
  if (M <= 0 || N < 8 || N % 8 != 0 || K < 64 || K % 64 != 0) return pl;
  // Measured on the R50 shapes (2 images; scripts/probes/gemm_probe.hip, profiles/r05/gemm_probe_*.txt): a CU moves 64 B /
  // clk from L2 into LDS -- what a 128 x 128 x 64 stage needs for its 512 cycles of MFMA -- and a lone 4-wave workgroup
  // exposes every latency of its K loop.  So: many tiles -> 128 x 128, two 8-wave workgroups per CU; about one tile per
  // CU -> one workgroup per CU with a 4-stage ring (three stages in flight); between one and two rounds of 128-row
  // tiles (263 for res3's forward) -> 64-row tiles, two small workgroups per CU; few tiles -> smaller tiles.
  const long t128 = (long)cdiv(M, 128) * cdiv(N, 128), t64 = (long)cdiv(M, 64) * cdiv(N, 128);
  pl.BK = 64; pl.WM = 2; pl.WN = 2;
  if (t128 >= 480) {
    pl.BM = 128; pl.BN = 128; pl.NST = 2; pl.WN = 4;
    if (K <= 128) { pl.BK = 32; pl.NST = 3; pl.WN = 2; }  // (2-4 K steps per tile: finer stages start the MFMAs earlier)
  } else if (t128 > 256) {
    pl.BM = 64; pl.BN = 128; pl.NST = 2;
  } else if (t128 >= 128) {
    pl.BM = 128; pl.BN = 128; pl.NST = 4; pl.WN = 4;
  } else if (t64 >= 128) {
    pl.BM = 64; pl.BN = 128; pl.NST = 4; pl.WN = 4;
  } else {
    pl.BM = 64; pl.BN = 64; pl.NST = 4;
  }
  pl.n_mt = cdiv(M, pl.BM);
  pl.n_nt = cdiv(N, pl.BN);
  pl.lds = gemm_nt_lds_bytes(pl.BM, pl.BN, pl.BK, pl.NST);
  pl.ok = (long)pl.n_mt * pl.n_nt < (1l << 30);
  return pl;
}

template <typename T, int BM, int BN, int BK, int NST, int WM, int WN>
static int gemm_nt_go(const GemmNtPlan& pl, const GemmNtArgs& a, hipStream_t st) {
  // > 64 KB of dynamic LDS needs the opt-in, per device
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  D2_HIP_OK(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64 || !done[dev]) {
      D2_HIP_OK(hipFuncSetAttribute((const void*)gemm_nt_kernel<T, BM, BN, BK, NST, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)gemm_nt_lds_bytes(BM, BN, BK, NST)));
      if (dev >= 0 && dev < 64) done[dev] = true;
    }
  }
  const int grid = (a.total + 7) / 8 * 8;
  hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, BK, NST, WM, WN>), dim3(grid), dim3(64 * WM * WN), gemm_nt_lds_bytes(BM, BN, BK, NST), st, a);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

template <typename T>
int gemm_nt_launch(const GemmNtPlan& pl, GemmNtArgs a, hipStream_t st) {
  a.n_mt = pl.n_mt; a.n_nt = pl.n_nt;
  a.total = pl.n_mt * pl.n_nt;
  D2_CHECK_ARG(a.ldx % 8 == 0 && a.ldw % 8 == 0 && a.ldo % 8 == 0, "gemm_nt: leading dimensions must be multiples of 8");
#define GM_CASE(bm, bn, bk, nst, wm, wn)                                                                        \
  if (pl.BM == bm && pl.BN == bn && pl.BK == bk && pl.NST == nst && pl.WM == wm && pl.WN == wn)               \
    return gemm_nt_go<T, bm, bn, bk, nst, wm, wn>(pl, a, st)
  GM_CASE(128, 128, 64, 2, 2, 4); GM_CASE(128, 128, 32, 3, 2, 2); GM_CASE(64, 128, 64, 2, 2, 2);
  GM_CASE(128, 128, 64, 4, 2, 4); GM_CASE(64, 128, 64, 4, 2, 4); GM_CASE(64, 64, 64, 4, 2, 2);
#ifdef GM_PROBE_VARIANTS
  GM_CASE(128, 128, 64, 2, 2, 2); GM_CASE(128, 128, 64, 3, 2, 2); GM_CASE(128, 128, 64, 4, 2, 2); GM_CASE(128, 128, 64, 3, 2, 4);
  GM_CASE(128, 128, 64, 2, 4, 2); GM_CASE(128, 128, 64, 3, 4, 2); GM_CASE(128, 128, 64, 4, 4, 2);
  GM_CASE(128, 128, 32, 4, 2, 4); GM_CASE(128, 128, 32, 6, 2, 4);
  GM_CASE(64, 128, 64, 3, 2, 4); GM_CASE(64, 128, 64, 3, 2, 2); GM_CASE(64, 128, 64, 4, 2, 2);
  GM_CASE(64, 64, 64, 2, 2, 2); GM_CASE(64, 64, 64, 3, 2, 2); GM_CASE(64, 64, 64, 6, 2, 2);
#endif
#undef GM_CASE
  set_error("gemm_nt: no kernel for tile %d x %d x %d, %d stages", pl.BM, pl.BN, pl.BK, pl.NST);
  return D2AMD_EUNSUPPORTED;
}

template int gemm_nt_launch<bf16_t>(const GemmNtPlan&, GemmNtArgs, hipStream_t);
template int gemm_nt_launch<f16_t>(const GemmNtPlan&, GemmNtArgs, hipStream_t);

}  // namespace d2amd
