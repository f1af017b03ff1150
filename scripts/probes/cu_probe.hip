// Micro-probe of one CU: core clock against the 100 MHz wall clock, MFMA issue interval, ds_read_b128 latency and
// throughput, s_barrier cost, global_load_lds round trip (L2 hit).  One workgroup of 256 threads unless noted.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scripts/probes/cu_probe scripts/probes/cu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long core_clock() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ unsigned long long wall_clock() { return wall_clock64(); }

// out[0..]: per test {core cycles, wall ticks}
__global__ __launch_bounds__(256) void probe_kernel(unsigned long long* out, const char* gsrc, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16384; i += 256) ((unsigned*)smem)[i] = i * 2654435761u;
  __syncthreads();
  unsigned long long c0, c1, w0, w1;
  // ---- 1. independent MFMAs, 4 accumulators, back to back
  {
    f32x16 acc[4];
    for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    bf16x8 x; for (int e = 0; e < 8; e++) x[e] = (__bf16)(float)(lane + e);
    __syncthreads();
    c0 = core_clock(); w0 = wall_clock();
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int a = 0; a < 4; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc[a], 0, 0, 0);
    }
    float s = 0; for (int a = 0; a < 4; a++) s += acc[a][0];
    asm volatile("" ::"v"(s));
    c1 = core_clock(); w1 = wall_clock();
    if (tid == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    if (s == 12345.f) sink[0] = s;
  }
  __syncthreads();
  // ---- 2. dependent ds_read_b128 chain (latency), wave 0 only; then all 4 waves, 4 independent reads per wait
  {
    unsigned addr = (lane * 16) & 0xfff0;
    __syncthreads();
    c0 = core_clock(); w0 = wall_clock();
    if (wave == 0)
      for (int i = 0; i < iters; i++) {
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + addr);
        addr = (v[0] & 0xff00) | (lane * 16 & 0xf0);
      }
    c1 = core_clock(); w1 = wall_clock();
    if (tid == 0) { out[2] = c1 - c0; out[3] = w1 - w0; }
    if (addr == 0xffffff) sink[1] = 1.f;
    __syncthreads();
    unsigned acc = 0;
    c0 = core_clock(); w0 = wall_clock();
    for (int i = 0; i < iters; i++) {
      u32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = *reinterpret_cast<const u32x4*>(smem + ((lane * 16 + j * 4096 + i * 1024) & 0xfff0));
#pragma unroll
      for (int j = 0; j < 4; j++) acc += v[j][0];
    }
    c1 = core_clock(); w1 = wall_clock();
    if (tid == 0) { out[4] = c1 - c0; out[5] = w1 - w0; }
    if (acc == 0xffffff) sink[2] = 1.f;
  }
  __syncthreads();
  // ---- 3. s_barrier alone
  {
    c0 = core_clock(); w0 = wall_clock();
    for (int i = 0; i < iters; i++) __builtin_amdgcn_s_barrier();
    c1 = core_clock(); w1 = wall_clock();
    if (tid == 0) { out[6] = c1 - c0; out[7] = w1 - w0; }
  }
  __syncthreads();
  // ---- 4. global_load_lds round trip (same 32 KB every time: L2 / L1 hit), 8 per wave then vmcnt(0)
  {
    c0 = core_clock(); w0 = wall_clock();
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int j = 0; j < 8; j++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)((wave * 8 + j) * 1024 + lane * 16)),
                                         (__attribute__((address_space(3))) void*)(smem + (wave * 8 + j) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    c1 = core_clock(); w1 = wall_clock();
    if (tid == 0) { out[8] = c1 - c0; out[9] = w1 - w0; }
  }
  __syncthreads();
  // ---- 5. one k step of the GEMM loop shape: 4 x (4 ds_read_b128 + 4 MFMA), no DMA, barrier per step
  {
    f32x16 acc[4];
    for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    c0 = core_clock(); w0 = wall_clock();
    for (int i = 0; i < iters; i++) {
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        u32x4 f[4];
#pragma unroll
        for (int j = 0; j < 4; j++) f[j] = *reinterpret_cast<const u32x4*>(smem + ((lane * 16 + j * 4096 + kk * 1024 + (i & 1) * 32768) & 0xfff0));
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[0]), __builtin_bit_cast(bf16x8, f[2]), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[0]), __builtin_bit_cast(bf16x8, f[3]), acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[1]), __builtin_bit_cast(bf16x8, f[2]), acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[1]), __builtin_bit_cast(bf16x8, f[3]), acc[3], 0, 0, 0);
      }
    }
    float s = 0; for (int a = 0; a < 4; a++) s += acc[a][0];
    asm volatile("" ::"v"(s));
    c1 = core_clock(); w1 = wall_clock();
    if (tid == 0) { out[10] = c1 - c0; out[11] = w1 - w0; }
    if (s == 12345.f) sink[3] = s;
  }
  __syncthreads();
  // ---- 6. the same k step WITH the DMA of a later stage in flight: 8 global_load_lds per wave into stage (i % 3) of a
  //         96 KB ring while the fragments are read from a fourth region; vmcnt(16) = two stages in flight
  {
    f32x16 acc[4];
    for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    c0 = core_clock(); w0 = wall_clock();
    for (int i = 0; i < iters; i++) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      char* ring = smem + 32768 + (i % 3) * 32768;
#pragma unroll
      for (int j = 0; j < 8; j++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)((wave * 8 + j) * 1024 + lane * 16) + (size_t)(i & 7) * 32768),
                                         (__attribute__((address_space(3))) void*)(ring + (wave * 8 + j) * 1024), 16, 0, 0);
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        u32x4 f[4];
#pragma unroll
        for (int j = 0; j < 4; j++) f[j] = *reinterpret_cast<const u32x4*>(smem + ((lane * 16 + j * 4096 + kk * 1024) & 0x7ff0));
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[0]), __builtin_bit_cast(bf16x8, f[2]), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[0]), __builtin_bit_cast(bf16x8, f[3]), acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[1]), __builtin_bit_cast(bf16x8, f[2]), acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[1]), __builtin_bit_cast(bf16x8, f[3]), acc[3], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0; for (int a = 0; a < 4; a++) s += acc[a][0];
    asm volatile("" ::"v"(s));
    c1 = core_clock(); w1 = wall_clock();
    if (tid == 0) { out[12] = c1 - c0; out[13] = w1 - w0; }
    if (s == 12345.f) sink[4] = s;
  }
  __syncthreads();
  // ---- 7. the DMA alone at that depth (no fragment reads / MFMA)
  {
    c0 = core_clock(); w0 = wall_clock();
    for (int i = 0; i < iters; i++) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      char* ring = smem + 32768 + (i % 3) * 32768;
#pragma unroll
      for (int j = 0; j < 8; j++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)((wave * 8 + j) * 1024 + lane * 16) + (size_t)(i & 7) * 32768),
                                         (__attribute__((address_space(3))) void*)(ring + (wave * 8 + j) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    c1 = core_clock(); w1 = wall_clock();
    if (tid == 0) { out[14] = c1 - c0; out[15] = w1 - w0; }
  }
}

int main(int argc, char** argv) {
  const int iters = 2000;
  unsigned long long* d; char* g; float* sink;
  CK(hipMalloc(&d, 64 * 8)); CK(hipMalloc(&g, 1 << 20)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(g, 1, 1 << 20));
  CK(hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int grid : {1, 256}) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(256), 131072, 0, d, g, iters, sink);
      CK(hipDeviceSynchronize());
    }
    unsigned long long h[16];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    int wf = 0; CK(hipDeviceGetAttribute(&wf, hipDeviceAttributeWallClockRate, 0));
    printf("grid %d (wall clock rate %d kHz)\n", grid, wf);
    const char* names[] = {"4 independent MFMA 32x32x16 per iter", "dependent ds_read_b128 chain (1 wave)", "4 ds_read_b128 + wait per iter (4 waves)",
                           "s_barrier per iter", "8 global_load_lds dwordx4 per wave + vmcnt(0) per iter (32 KB, cached)",
                           "GEMM k step: barrier + 4 x (4 ds_read_b128, 4 MFMA)",
                           "the same k step + 8 global_load_lds per wave, two stages in flight (vmcnt(16))",
                           "8 global_load_lds per wave per iter alone, two stages in flight"};
    for (int t = 0; t < 8; t++) {
      const double cyc = (double)h[2 * t] / iters, us = (double)h[2 * t + 1] / (wf * 1e-3) / iters;
      printf("  %-75s %8.1f core cycles/iter  %7.4f us/iter  -> core clock %.2f GHz\n", names[t], cyc, us, cyc / us * 1e-3);
    }
  }
  return 0;
}
