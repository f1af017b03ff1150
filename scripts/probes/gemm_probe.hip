// Stand-alone probe of the dense NT GEMM (detectron2_amd/csrc/dcn_gemm.h): correctness against a host reference
// (sampled entries, fp64 accumulation) and launch time at the DCN shapes, per tile shape / stage count / split.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scripts/probes/gemm_probe scripts/probes/gemm_probe.hip
//   ./scripts/probes/gemm_probe [M N K BM NST SPLITK]...
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../detectron2_amd/csrc/dcn_gemm.hip"

namespace d2amd {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fprintf(stderr, "\n");
}
}  // namespace d2amd
using namespace d2amd;

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint32_t rng_state = 12345u;
static float frand() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f;
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                   \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

static int run(int M, int N, int K, int BM, int BK, int NST, int splitk, bool bias, int BN = 128, int WM = 2, int WN = 2) {
  GemmNtPlan pl = gemm_nt_plan(M, N, K);
  if (!pl.ok) { printf("plan failed\n"); return 1; }
  if (BM > 0) {
    pl.BM = BM; pl.BN = BN; pl.BK = BK; pl.NST = NST; pl.WM = WM; pl.WN = WN; pl.n_mt = cdiv(M, BM); pl.n_nt = cdiv(N, BN);
  }
  std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K), hb(N), ho((size_t)M * N);
  for (auto& v : hx) v = f2bf(frand());
  for (auto& v : hw) v = f2bf(frand() * 0.1f);
  for (auto& v : hb) v = f2bf(frand());
  void *dx, *dw, *dout, *db;
  CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&dout, ho.size() * 2)); CK(hipMalloc(&db, N * 2));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dout, 0xff, ho.size() * 2));
  GemmNtArgs a{};
  a.X = dx; a.Wn = dw; a.out = dout; a.bias = bias ? db : nullptr;
  a.M = M; a.N = N; a.K = K; a.ldx = K; a.ldw = K; a.ldo = N;
  if (gemm_nt_launch<bf16_t>(pl, a, 0)) return 1;
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost));
  // sampled check: random entries + the last rows / columns
  int bad = 0;
  double worst = 0;
  const int NS = 6000;
  for (int s = 0; s < NS; s++) {
    int m, n;
    if (s < 4000) { rng_state = rng_state * 1664525u + 1013904223u; m = (rng_state >> 4) % M; rng_state = rng_state * 1664525u + 1013904223u; n = (rng_state >> 4) % N; }
    else if (s < 5000) { m = M - 1 - (s % 70); n = (s * 7) % N; }
    else { m = (s * 131) % M; n = N - 1 - (s % 40); }
    double ref = bias ? bf2f(hb[n]) : 0.0;
    for (int k = 0; k < K; k++) ref += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
    const double got = bf2f(ho[(size_t)m * N + n]);
    const double err = fabs(got - ref), tol = 0.01 * fabs(ref) + 0.02;
    if (err / tol > worst) worst = err / tol;
    if (!(err <= tol)) { if (bad < 5) printf("  MISMATCH m=%d n=%d got %g ref %g\n", m, n, got, ref); bad++; }
  }
  // repeat: results must be bit-identical run to run (split-K reduction in split order)
  std::vector<uint16_t> ho2(ho.size());
  int nondet = 0;
  for (int rep = 0; rep < 3; rep++) {
    if (gemm_nt_launch<bf16_t>(pl, a, 0)) return 1;
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ho2.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost));
    if (memcmp(ho.data(), ho2.data(), ho.size() * 2)) nondet++;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int WARM = getenv("GM_WARM") ? atoi(getenv("GM_WARM")) : 5;
  for (int i = 0; i < WARM; i++) gemm_nt_launch<bf16_t>(pl, a, 0);
  CK(hipEventRecord(e0, 0));
  const int REP = getenv("GM_REP") ? atoi(getenv("GM_REP")) : 50;
  for (int i = 0; i < REP; i++) gemm_nt_launch<bf16_t>(pl, a, 0);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / REP, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("M=%d N=%d K=%d tile %dx%dx%d nst %d waves %dx%d (lds %zu KB, %d tiles)%s: %.2f us  %.1f TF  bad %d/%d worst %.2f nondet %d\n", M, N, K,
         pl.BM, pl.BN, pl.BK, pl.NST, pl.WM, pl.WN, gemm_nt_lds_bytes(pl.BM, pl.BN, pl.BK, pl.NST) >> 10, pl.n_mt * pl.n_nt, bias ? " +bias" : "", us, tf, bad, NS, worst, nondet);
  hipFree(dx); hipFree(dw); hipFree(dout); hipFree(db);
  return bad != 0 || nondet != 0;
}

int main(int argc, char** argv) {
  int rc = 0;
  // small / ragged shapes first (edges, bias), then the DCN shapes of R50 at 2 images
  rc |= run(200, 64, 128, 64, 64, 2, 1, true);
  rc |= run(333, 200, 256, 128, 64, 2, 1, false);
  rc |= run(333, 200, 512, 64, 64, 4, 1, true, 64);
  rc |= run(1000, 136, 1024, 128, 64, 3, 1, false, 128, 2, 4);
  rc |= run(777, 264, 512, 128, 64, 4, 1, true, 128, 4, 2);
  const int P[3] = {33600, 8400, 2100}, C[3] = {128, 256, 512};
  for (int s = 0; s < 3; s++) {
    rc |= run(P[s], C[s], 9 * C[s], 0, 0, 0, 0, true);    // forward, planned
    rc |= run(P[s], 9 * C[s], C[s], 0, 0, 0, 0, false);   // backward-data, planned
  }
  // {BM, BN, BK, NST, WM, WN}
  const int V[][6] = {{128, 128, 64, 2, 2, 2}, {128, 128, 64, 3, 2, 2}, {128, 128, 64, 4, 2, 2},
                      {128, 128, 64, 2, 2, 4}, {128, 128, 64, 3, 2, 4}, {128, 128, 64, 4, 2, 4},
                      {128, 128, 64, 2, 4, 2}, {128, 128, 64, 3, 4, 2}, {128, 128, 64, 4, 4, 2},
                      {128, 128, 32, 4, 2, 4}, {128, 128, 32, 6, 2, 4},
                      {64, 128, 64, 2, 2, 2}, {64, 128, 64, 3, 2, 4}, {64, 128, 64, 4, 2, 4},
                      {64, 64, 64, 2, 2, 2}, {64, 64, 64, 4, 2, 2}};
  printf("---- no tail: 256 tiles of 128 rows\n");
  for (auto& v : V) rc |= run(32768, 128, 1152, v[0], v[2], v[3], 1, false, v[1], v[4], v[5]);
  for (int s = 0; s < 3; s++) {
    printf("---- forward stage %d\n", s + 3);
    for (auto& v : V) rc |= run(P[s], C[s], 9 * C[s], v[0], v[2], v[3], 1, false, v[1], v[4], v[5]);
    printf("---- backward-data stage %d\n", s + 3);
    for (auto& v : V) rc |= run(P[s], 9 * C[s], C[s], v[0], v[2], v[3], 1, false, v[1], v[4], v[5]);
  }
  printf(rc ? "FAILED\n" : "ALL OK\n");
  return rc;
}
