// Introspection + error plumbing of libd2amd.so.
// Replaces detectron2/layers/csrc/vision.cpp:16-79 (get_cuda_version / get_compiler_version).
#include <stdarg.h>
#include <string.h>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace d2amd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace d2amd

// ---- kernel timing aid -------------------------------------------------------------------------------
// HIP events recorded on the LAUNCH stream right before / after selected kernels (the side stream of the
// tile-gather backward is not visible to events a caller records on its own stream).  Off by default.
namespace d2amd {
struct TimingSlot { const char* name; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; };
static TimingSlot g_slots[8];
static int g_nslots = 0;
static int g_timing_mask = 0;  // bit i: time the i-th registered kernel name (registration order below)
static const char* const g_timing_names[] = {"pool_bwd_fine_r7", "pool_bwd_coarse_r7", "pool_bwd_fine_r14",
                                              "pool_bwd_coarse_r14", "pool_bwd_staged_r7", "pool_bwd_staged_r14"};
static const int g_timing_bits[] = {0, 1, 2, 3, 0, 2};  // the single staged launch answers to the "fine" bits
static TimingSlot* timing_slot(const char* name) {
  for (int i = 0; i < g_nslots; i++)
    if (!strcmp(g_slots[i].name, name)) return &g_slots[i];
  if (g_nslots >= 8) return nullptr;
  g_slots[g_nslots].name = name;
  return &g_slots[g_nslots++];
}
bool timing_begin(const char* name, hipStream_t s) {
  if (!g_timing_mask) return false;
  int bit = -1;
  for (int i = 0; i < 6; i++)
    if (!strcmp(g_timing_names[i], name)) bit = g_timing_bits[i];
  if (bit < 0 || !(g_timing_mask & (1 << bit))) return false;
  TimingSlot* t = timing_slot(name);
  if (!t || t->ev.size() >= 65536) return false;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
  t->ev.emplace_back(a, b);
  (void)hipEventRecord(a, s);
  return true;
}
void timing_end(const char* name, hipStream_t s) {
  TimingSlot* t = timing_slot(name);
  if (t && !t->ev.empty()) (void)hipEventRecord(t->ev.back().second, s);
}
}  // namespace d2amd

extern "C" {

void d2amd_timing_enable(int on) {
  using namespace d2amd;
  for (int i = 0; i < g_nslots; i++) {
    for (auto& p : g_slots[i].ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    g_slots[i].ev.clear();
  }
  g_timing_mask = on;
}

int d2amd_timing_read(const char* kernel, double* total_ms, int* launches) {
  using namespace d2amd;
  D2_CHECK_ARG(kernel && total_ms && launches, "timing_read: null pointer");
  *total_ms = 0.0;
  *launches = 0;
  for (int i = 0; i < g_nslots; i++) {
    if (strcmp(g_slots[i].name, kernel)) continue;
    for (auto& p : g_slots[i].ev) {
      float ms = 0.f;
      D2_HIP_OK(hipEventSynchronize(p.second));
      D2_HIP_OK(hipEventElapsedTime(&ms, p.first, p.second));
      *total_ms += ms;
      (*launches)++;
    }
  }
  return D2AMD_OK;
}

const char* d2amd_version(void) { return "d2amd 0.1 (gfx950)"; }

const char* d2amd_compiler_version(void) {
  static std::string s = std::string("clang ") + std::to_string(__clang_major__) + "." +
      std::to_string(__clang_minor__) + "." + std::to_string(__clang_patchlevel__);
  return s.c_str();
}

const char* d2amd_hip_version(void) {
  static std::string s;
  int v = 0;
  if (hipRuntimeGetVersion(&v) != hipSuccess) v = HIP_VERSION;
  // same formatting as vision.cpp:16-39: "HIP <major>.<minor>"
  s = std::string("HIP ") + std::to_string(v / 10000000) + "." + std::to_string(v / 100000 % 100);
  return s.c_str();
}

const char* d2amd_last_error(void) { return d2amd::g_err; }

}  // extern "C"
