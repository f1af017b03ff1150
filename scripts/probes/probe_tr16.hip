// Probe: lane mapping of ds_read_tr16_b64 and a B-operand built with it for v_mfma_f32_32x32x16_bf16.
// hipcc --offload-arch=gfx950 -O2 probe_tr16.hip -o probe_tr16 && ./probe_tr16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(int* out, float* cout) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 256];  // image [row k][col c], 256 cols
  for (int i = threadIdx.x; i < 64 * 256; i += 64) lds[i] = (short)i;  // value = k * 256 + c
  __syncthreads();
  const int l = threadIdx.x;
  // hypothesis: lane i of a 16-lane group supplies the address of row (i >> 2), cols (i & 3) * 4 .. + 3 of a [4][16]
  // block; it receives column i, rows 0..3
  const int g = l >> 4, i = l & 15;
  const short* a = lds + ((i >> 2) * 256 + g * 16 + (i & 3) * 4);
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)a);
  for (int j = 0; j < 4; j++) out[l * 4 + j] = (unsigned short)r[j];
  // MFMA check: C[m][n] = sum_k A[m][k] B[k][n], A[m][k] = (m == k) (identity 32 x 16 -> C[m][n] = B[m][n] for m < 16)
  // B[k][n] = image value of row k, col n (n < 32), loaded with two tr reads per lane
  __shared__ __attribute__((aligned(16))) __bf16 img[16 * 256];
  for (int q = threadIdx.x; q < 16 * 256; q += 64) { const int kk = q / 256, nn = (q % 256) % 32; img[q] = (__bf16)((nn < 16 ? 1.f : -1.f) * (float)(kk * 16 + nn % 16 + 1)); }  // exact in bf16 (<= 256)
  __syncthreads();
  const int n = l & 31, kh = l >> 5;  // B operand: lane holds column n, k = 8 * kh .. + 8
  v8bf B, A;
  v4s t2[2];
  for (int rd = 0; rd < 2; rd++) {
    const int kbase = 8 * kh + 4 * rd;
    const int grp = (l >> 4) & 1, ii = l & 15;  // 16-lane group inside the 32 columns
    const __bf16* p = img + ((kbase + (ii >> 2)) * 256 + grp * 16 + (ii & 3) * 4);
    t2[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)p);
  }
  typedef short v8s __attribute__((ext_vector_type(8)));
  const v8s both = __builtin_shufflevector(t2[0], t2[1], 0, 1, 2, 3, 4, 5, 6, 7);
  B = __builtin_bit_cast(v8bf, both);
  for (int j = 0; j < 8; j++) A[j] = (__bf16)(((l & 31) == 8 * kh + j) ? 1.0f : 0.0f);  // A[m = l & 31][k = 8 kh + j]
  v16f c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c, 0, 0, 0);
  for (int r2 = 0; r2 < 16; r2++) cout[l * 16 + r2] = c[r2];
  for (int j = 0; j < 8; j++) cout[1024 + l * 8 + j] = (float)B[j];
}

int main() {
  int* d; float* dc;
  hipMalloc(&d, 64 * 4 * 4); hipMalloc(&dc, (64 * 16 + 64 * 8) * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, dc);
  int h[256]; float hc[1024 + 512];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; j++) { printf(" (k%d,c%d)", h[l * 4 + j] / 256, h[l * 4 + j] % 256); bad += h[l * 4 + j] != j * 256 + (l >> 4) * 16 + (l & 15); }
    printf("\n");
  }
  printf("tr16 hypothesis mismatches: %d\n", bad);
  // C layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); expect C[m][n] = m * 32 + n for m < 16, else 0
  int badc = 0;
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 16; r++) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
      const float want = row < 16 ? (col < 16 ? 1.f : -1.f) * (float)(row * 16 + col % 16 + 1) : 0.f;
      if (hc[l * 16 + r] != want) { if (badc < 8) printf("C mismatch lane %d r %d row %d col %d got %g want %g\n", l, r, row, col, hc[l * 16 + r], want); badc++; }
    }
  printf("mfma B-via-tr16 mismatches: %d\n", badc);
  for (int l = 0; l < 64; l += 9) { printf("B lane %d:", l); for (int j = 0; j < 8; j++) printf(" %g", hc[1024 + l * 8 + j]); printf("\n"); }
  return 0;
}
