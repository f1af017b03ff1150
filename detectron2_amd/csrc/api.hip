// Introspection + error plumbing of libd2amd.so.
// Replaces detectron2/layers/csrc/vision.cpp:16-79 (get_cuda_version / get_compiler_version).
#include <stdarg.h>
#include <string>

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

extern "C" {

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
