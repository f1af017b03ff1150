// Uniform sampling keys generated ON THE DEVICE from a device-resident generator state.
//   serves    subsample_labels (detectron2/modeling/sampling.py:9-54) as called from proposal_generator/rpn.py:287-305
//             and roi_heads/roi_heads.py:181-216: the reference draws two torch.randperm per image; the device samplers
//             of this library (subsample.hip, label_sample.hip) take one uniform key per element instead.
// Why not torch.rand: inside a captured HIP graph torch's generator is fed by TWO fill launches that
// CUDAGraph.replay() issues in front of every hipGraphLaunch (seed and Philox offset written from the host side):
// ~10 us at the head of a 0.44 ms step (gpurun_out/r3q trace).  Here the state {seed, offset} lives in device memory:
// the kernel reads it, every thread derives its own Philox4x32-10 counter from (offset, thread), and the last workgroup
// to finish advances the offset -- a replayed graph draws fresh keys with no host involvement.
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator torch / curand use):
// counter = {thread index lo, hi, offset lo, hi}, key = seed.  u32 -> [0, 1): (x >> 8) * 2^-24 (24 random bits: every
// value is exactly representable, 0 included, 1 excluded -- the convention of torch.rand for fp32).
#include "common.h"
#include "philox.h"

namespace d2amd {

// state: [0] seed, [1] offset (in units of 4 outputs per thread-slot), [2] ticket of the running launch
__global__ __launch_bounds__(256) void uniform_keys_kernel(unsigned long long* __restrict__ state, float* __restrict__ out,
                                                          long n) {
  const unsigned long long seed = state[0], offset = state[1];
  const long q = (long)blockIdx.x * 256 + threadIdx.x;  // this thread writes outputs 4 q .. 4 q + 3
  if (4 * q < n) {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)((unsigned long long)q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = (float)(c[i] >> 8) * 5.9604644775390625e-08f;  // 2^-24
    if (4 * q + 3 < n && ((uintptr_t)out & 15) == 0) {
      *reinterpret_cast<float4*>(out + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int i = 0; i < 4 && 4 * q + i < n; i++) out[4 * q + i] = v[i];
    }
  }
  // the last workgroup to finish advances the offset (every other one has read it: it finished) and re-arms the ticket
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = atomicAdd(&state[2], 1ull);
    if (t == (unsigned long long)gridDim.x - 1) {
      state[1] = offset + 1;  // one offset step per launch: the counter's low words already separate the threads
      state[2] = 0;
    }
  }
}

}  // namespace d2amd

using namespace d2amd;

extern "C" int d2amd_uniform_keys(uint64_t* state, float* out, int64_t n, void* stream) {
  D2_CHECK_ARG(n >= 0, "uniform_keys: negative size");
  if (n == 0) return D2AMD_OK;
  D2_CHECK_ARG(state != nullptr && out != nullptr, "uniform_keys: null pointer");
  D2_CHECK_ARG(n < (1ll << 40), "uniform_keys: too many keys");
  const long quads = (n + 3) / 4;
  hipLaunchKernelGGL(uniform_keys_kernel, dim3(cdiv(quads, 256)), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long*)state, out, (long)n);
  D2_LAUNCH_OK();
  return D2AMD_OK;
}
