// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_max3_f32 / v_cndmask on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  float x[16];
  v2f y[8];
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
#pragma unroll
  for (int i = 0; i < 8; i++) y[i] = (v2f){x[2 * i], x[2 * i + 1]};
  const v2f a2 = {a, a}, b2 = {b, b};
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = fmaf(x[i], a, b);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; i++) y[i] = __builtin_elementwise_fma(y[i], a2, b2);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = fmaxf(fmaxf(x[i], a), x[(i + 1) & 15]);
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; i++) y[i] = y[i] * a2;
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 8; i++) y[i] = y[i] + a2;
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += x[i];
#pragma unroll
  for (int i = 0; i < 8; i++) s += y[i].x + y[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, float* d, int per_iter_instr) {
  const int blocks = 256 * 8, iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = blocks * 4.0, instr = waves * iters * per_iter_instr;
  // 1024 SIMDs
  printf("%-14s %.3f ms  -> %.2f cycles/instr/SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / (instr / 1024.0));
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_fma_f32", d, 16);
  run<1>("v_pk_fma_f32", d, 8);
  run<2>("v_max3_f32", d, 16);
  run<3>("v_pk_mul_f32", d, 8);
  run<4>("v_pk_add_f32", d, 8);
  return 0;
}
