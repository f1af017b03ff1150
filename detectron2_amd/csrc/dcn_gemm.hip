// Host side of the dense NT GEMM (dcn_gemm.h): plan (tile shape, split) and launch.
#include "dcn_gemm.h"

#include <mutex>

namespace d2amd {

GemmNtPlan gemm_nt_plan(int M, int N, int K) {
  GemmNtPlan pl{};
  pl.ok = false;
  if (M <= 0 || N < 8 || N % 8 != 0 || K < 64 || K % 64 != 0) return pl;
  // Measured on the R50 shapes (2 images; scripts/probes/gemm_probe.hip, profiles/r05/LOG.md): a CU moves 64 B / clk from
  // L2 into LDS, exactly what a 128 x 128 x 64 stage needs for its 512 cycles of MFMA, and one workgroup alone exposes
  // the latency of every stage -- two resident workgroups per CU are worth more than a bigger tile.
  const long t128 = (long)cdiv(M, 128) * cdiv(N, 128), t64 = (long)cdiv(M, 64) * cdiv(N, 128);
  pl.BK = 64;
  if (t128 >= 480) {
    pl.BM = 128; pl.BN = 128; pl.NST = 2;
    if (K <= 128) { pl.BK = 32; pl.NST = 3; }  // (2-4 K steps per tile: finer stages start the MFMAs earlier)
  } else if (t64 >= 480) {
    pl.BM = 64; pl.BN = 128; pl.NST = 2;
  } else {
    pl.BM = 64; pl.BN = 64;
    pl.NST = (long)cdiv(M, 64) * cdiv(N, 64) >= 480 ? 2 : 4;
  }
  pl.n_mt = cdiv(M, pl.BM);
  pl.n_nt = cdiv(N, pl.BN);
  pl.lds = gemm_nt_lds_bytes(pl.BM, pl.BN, pl.BK, pl.NST);
  pl.ok = (long)pl.n_mt * pl.n_nt < (1l << 30);
  return pl;
}

template <typename T, int BM, int BN, int BK, int NST, int ABL = 0, bool PF = false>
static int gemm_nt_go(const GemmNtPlan& pl, const GemmNtArgs& a, hipStream_t st) {
  // > 64 KB of dynamic LDS needs the opt-in, per device
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  D2_HIP_OK(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64 || !done[dev]) {
      D2_HIP_OK(hipFuncSetAttribute((const void*)gemm_nt_kernel<T, BM, BN, BK, NST, ABL, PF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)gemm_nt_lds_bytes(BM, BN, BK, NST)));
      if (dev >= 0 && dev < 64) done[dev] = true;
    }
  }
  const int grid = (a.total + 7) / 8 * 8;
  hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, BK, NST, ABL, PF>), dim3(grid), dim3(GM_THREADS), gemm_nt_lds_bytes(BM, BN, BK, NST), st, a);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}

template <typename T>
int gemm_nt_launch(const GemmNtPlan& pl, GemmNtArgs a, hipStream_t st) {
  a.n_mt = pl.n_mt; a.n_nt = pl.n_nt;
  a.total = pl.n_mt * pl.n_nt;
  D2_CHECK_ARG(a.ldx % 8 == 0 && a.ldw % 8 == 0 && a.ldo % 8 == 0, "gemm_nt: leading dimensions must be multiples of 8");
#define GM_CASE(bm, bk, nst) if (pl.BM == bm && pl.BN == 128 && pl.BK == bk && pl.NST == nst) return gemm_nt_go<T, bm, 128, bk, nst>(pl, a, st)
#define GM_CASE64(bk, nst) if (pl.BM == 64 && pl.BN == 64 && pl.BK == bk && pl.NST == nst) return gemm_nt_go<T, 64, 64, bk, nst>(pl, a, st)
  GM_CASE(128, 64, 2); GM_CASE(128, 32, 3); GM_CASE(64, 64, 2); GM_CASE64(64, 2); GM_CASE64(64, 4);
#ifdef GM_PROBE_VARIANTS
  if (pl.BM == 128 && pl.BN == 128 && pl.BK == 64 && pl.NST == 2 && pl.variant) {
    switch (pl.variant) {
      case 1: return gemm_nt_go<T, 128, 128, 64, 2, 1>(pl, a, st);
      case 2: return gemm_nt_go<T, 128, 128, 64, 2, 2>(pl, a, st);
      case 3: return gemm_nt_go<T, 128, 128, 64, 2, 3>(pl, a, st);
      case 4: return gemm_nt_go<T, 128, 128, 64, 2, 4>(pl, a, st);
      case 5: return gemm_nt_go<T, 128, 128, 64, 2, 0, true>(pl, a, st);
    }
  }
  if (pl.BM == 128 && pl.BN == 128 && pl.BK == 64 && pl.NST == 3 && pl.variant) {
    switch (pl.variant) {
      case 1: return gemm_nt_go<T, 128, 128, 64, 3, 1>(pl, a, st);
      case 2: return gemm_nt_go<T, 128, 128, 64, 3, 2>(pl, a, st);
      case 3: return gemm_nt_go<T, 128, 128, 64, 3, 3>(pl, a, st);
      case 4: return gemm_nt_go<T, 128, 128, 64, 3, 4>(pl, a, st);
      case 5: return gemm_nt_go<T, 128, 128, 64, 3, 0, true>(pl, a, st);
    }
  }
  GM_CASE64(64, 3); GM_CASE64(64, 6); GM_CASE64(32, 4); GM_CASE64(32, 8);
  GM_CASE(128, 64, 3); GM_CASE(128, 64, 4); GM_CASE(64, 64, 3); GM_CASE(64, 64, 4); GM_CASE(64, 64, 5);
  GM_CASE(128, 32, 4); GM_CASE(128, 32, 5); GM_CASE(128, 32, 6); GM_CASE(128, 32, 8);
  GM_CASE(64, 32, 4); GM_CASE(64, 32, 6);
#endif
#undef GM_CASE
#undef GM_CASE64
  set_error("gemm_nt: no kernel for tile %d x %d x %d, %d stages", pl.BM, pl.BN, pl.BK, pl.NST);
  return D2AMD_EUNSUPPORTED;
}

template int gemm_nt_launch<bf16_t>(const GemmNtPlan&, GemmNtArgs, hipStream_t);
template int gemm_nt_launch<f16_t>(const GemmNtPlan&, GemmNtArgs, hipStream_t);

}  // namespace d2amd
