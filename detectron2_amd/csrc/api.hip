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

namespace d2amd {
// head bytes up to 16-B alignment, 16-B vector stores, tail bytes
__global__ void zero_bytes_kernel(uint8_t* __restrict__ p, size_t head, size_t nvec, size_t tail, int* word) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (word != nullptr && t == 0) *word = 0;
  if (t < head) p[t] = 0;
  uint4* v = reinterpret_cast<uint4*>(p + head);
  for (size_t i = t; i < nvec; i += stride) v[i] = uint4{0u, 0u, 0u, 0u};
  if (t < tail) p[head + nvec * 16 + t] = 0;
}
int zero_async(void* ptr, size_t bytes, hipStream_t s, int* word) {
  if (bytes == 0 && word != nullptr) { ptr = word; bytes = sizeof(int); word = nullptr; }
  if (bytes == 0) return D2AMD_OK;
  D2_CHECK_ARG(ptr != nullptr, "zero_async: null pointer");
  size_t head = (16 - ((uintptr_t)ptr & 15)) & 15;
  if (head > bytes) head = bytes;
  const size_t nvec = (bytes - head) / 16, tail = bytes - head - nvec * 16;
  const size_t blocks = (nvec + 255) / 256;
  hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks < 8192 ? blocks : 8192))), dim3(256), 0, s,
                     (uint8_t*)ptr, head, nvec, tail, word);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}
}  // namespace d2amd

// ---- kernel timing aid -------------------------------------------------------------------------------
// HIP events recorded on the LAUNCH stream right before / after selected kernels (the side stream of the
// tile-gather backward is not visible to events a caller records on its own stream).  Off by default.
namespace d2amd {
struct TimingSlot { std::string name; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; };
static std::vector<TimingSlot> g_slots;
static std::vector<std::string> g_selected;  // d2amd_timing_select: kernel names to time
static int g_timing_mask = 0;  // d2amd_timing_enable: bit i times the pooler tile-gather launches below
static const char* const g_timing_names[] = {"pool_bwd_fine_r7", "pool_bwd_coarse_r7", "pool_bwd_fine_r14",
                                              "pool_bwd_coarse_r14", "pool_bwd_staged_r7", "pool_bwd_staged_r14"};
static const int g_timing_bits[] = {0, 1, 2, 3, 0, 2};  // the single staged launch answers to the "fine" bits
static TimingSlot* timing_slot(const char* name, bool create) {
  for (auto& t : g_slots)
    if (t.name == name) return &t;
  if (!create || g_slots.size() >= 64) return nullptr;
  g_slots.reserve(64);  // pointers into the vector stay valid
  g_slots.push_back(TimingSlot{name, {}});
  return &g_slots.back();
}
bool timing_begin(const char* name, hipStream_t s) {
  if (!g_timing_mask && g_selected.empty()) return false;
  bool on = false;
  for (const auto& n : g_selected) on = on || n == name;
  for (int i = 0; i < 6 && !on; i++)
    if (!strcmp(g_timing_names[i], name)) on = (g_timing_mask & (1 << g_timing_bits[i])) != 0;
  if (!on) return false;
  TimingSlot* t = timing_slot(name, true);
  if (!t || t->ev.size() >= 65536) return false;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
  t->ev.emplace_back(a, b);
  (void)hipEventRecord(a, s);
  return true;
}
void timing_end(const char* name, hipStream_t s) {
  TimingSlot* t = timing_slot(name, false);
  if (t && !t->ev.empty()) (void)hipEventRecord(t->ev.back().second, s);
}
static void timing_clear() {
  for (auto& t : g_slots) {
    for (auto& p : t.ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    t.ev.clear();
  }
}
}  // namespace d2amd

extern "C" {

void d2amd_timing_enable(int on) {
  d2amd::timing_clear();
  d2amd::g_timing_mask = on;
}

void d2amd_timing_select(const char* names_csv) {
  using namespace d2amd;
  timing_clear();
  g_selected.clear();
  if (!names_csv) return;
  std::string cur;
  for (const char* p = names_csv;; p++) {
    if (*p == ',' || *p == 0) {
      if (!cur.empty()) g_selected.push_back(cur);
      cur.clear();
      if (*p == 0) break;
    } else if (*p != ' ') {
      cur.push_back(*p);
    }
  }
}

int d2amd_timing_read(const char* kernel, double* total_ms, int* launches) {
  using namespace d2amd;
  D2_CHECK_ARG(kernel && total_ms && launches, "timing_read: null pointer");
  *total_ms = 0.0;
  *launches = 0;
  for (auto& t : g_slots) {
    if (t.name != kernel) continue;
    for (auto& p : t.ev) {
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
