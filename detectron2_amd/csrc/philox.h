// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator torch / curand
// use) and this library's key convention: output i of a draw = word i % 4 of the block with counter
// {i / 4 (lo, hi), offset (lo, hi)} and key = seed; u32 -> [0, 1): (x >> 8) * 2^-24.  Shared by random_keys.hip
// (d2amd_uniform_keys) and the samplers that draw their keys in the kernel (label_sample.hip).
#pragma once
#include <cstdint>

namespace d2amd {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// key `i` of the draw at (seed, offset): what d2amd_uniform_keys writes to out[i]
__device__ __forceinline__ float philox_key(unsigned long long seed, unsigned long long offset, long i) {
  const unsigned long long q = (unsigned long long)i >> 2;
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t x = (i & 3) == 0 ? c[0] : (i & 3) == 1 ? c[1] : (i & 3) == 2 ? c[2] : c[3];
  return (float)(x >> 8) * 5.9604644775390625e-08f;  // 2^-24
}

}  // namespace d2amd
