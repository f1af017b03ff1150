// Do the parallel branches of a captured HIP graph run concurrently on this stack (ROCm 7.2, gfx950)?
// Two spin kernels of `us` microseconds with `blocks` workgroups each: (a) back to back on one stream,
// (b) on two streams (fork / join with events), (c) captured as one-stream graph, (d) captured with the fork.
// build + run:  hipcc --offload-arch=gfx950 -O2 probe_graph_branches.hip -o /tmp/pgb && /tmp/pgb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

__global__ void spin(long ticks, int* sink) {
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  hipStream_t s0, s1;
  CK(hipStreamCreate(&s0));
  CK(hipStreamCreate(&s1));
  hipEvent_t fork, join;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  int* sink;
  CK(hipMalloc(&sink, 4));
  const long ticks = 100 * 100;  // wall_clock64: 100 MHz -> 100 us
  for (int blocks : {8, 256, 2048}) {
    auto serial = [&]() {
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, ticks, sink);
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, ticks, sink);
    };
    auto forked = [&]() {
      hipEventRecord(fork, s0);
      hipStreamWaitEvent(s1, fork, 0);
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, ticks, sink);
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, ticks, sink);
      hipEventRecord(join, s1);
      hipStreamWaitEvent(s0, join, 0);
    };
    auto timeit = [&](auto fn, int n) {
      fn();
      hipStreamSynchronize(s0);
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < n; i++) fn();
      hipStreamSynchronize(s0);
      return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    };
    hipGraph_t g;
    hipGraphExec_t ge_serial, ge_fork;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
    serial();
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge_serial, g, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
    forked();
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge_fork, g, nullptr, nullptr, 0));
    printf("blocks %4d x 256 threads, 2 x 100 us spin: one stream %.0f us | two streams %.0f us | graph, one stream %.0f us | graph, forked %.0f us\n",
           blocks, timeit(serial, 20), timeit(forked, 20), timeit([&]() { hipGraphLaunch(ge_serial, s0); }, 20),
           timeit([&]() { hipGraphLaunch(ge_fork, s0); }, 20));
  }
  return 0;
}
