// Checks the operand lane maps of v_mfma_i32_32x32x32_i8 on gfx950 with exact integer data.
// Hypothesis (by analogy with the bf16 32x32x16 map of the guide): lane l (r = l & 31, h = l >> 5) holds
// A[row r][k = 16 h + j] and B[k = 16 h + j][col r], j = 0..15 (16 bytes = 4 VGPRs);
// C/D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k(const signed char* A /*[32][32] row-major (row, k)*/, const signed char* Bt /*[32][32] (col, k)*/, int* C) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  v4i a = *reinterpret_cast<const v4i*>(A + r * 32 + 16 * h);
  v4i b = *reinterpret_cast<const v4i*>(Bt + r * 32 + 16 * h);
  v16i c = {0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int reg = 0; reg < 16; reg++) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h, col = r;
    C[row * 32 + col] = c[reg];
  }
}

int main() {
  signed char A[32 * 32], Bt[32 * 32];
  srand(1);
  for (int i = 0; i < 1024; i++) { A[i] = (signed char)(rand() % 255 - 127); Bt[i] = (signed char)(rand() % 255 - 127); }
  signed char *dA, *dB; int* dC;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, Bt, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  int C[1024];
  hipMemcpy(C, dC, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; i++)
    for (int j = 0; j < 32; j++) {
      int ref = 0;
      for (int t = 0; t < 32; t++) ref += (int)A[i * 32 + t] * (int)Bt[j * 32 + t];
      if (ref != C[i * 32 + j]) bad++;
    }
  printf("mfma_i32_32x32x32_i8 layout check: %d of 1024 elements differ\n", bad);
  return bad != 0;
}
