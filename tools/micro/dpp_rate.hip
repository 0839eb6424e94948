// dpp_rate.hip -- issue cost of DPP moves by control code: quad_perm, row_shr:1, wave_shr:1, wave_shl:1 (the extrema
// scan's neighbour exchange uses the two wave shifts).
//   hipcc -O3 --offload-arch=gfx950 -o dpp_rate dpp_rate.hip && ./dpp_rate
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int CTRL>
__global__ __launch_bounds__(256) void dpp_kernel(float* out, int iters) {
  int v0 = threadIdx.x, v1 = threadIdx.x * 3, v2 = threadIdx.x * 5, v3 = threadIdx.x * 7;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {  // four independent chains, 32 DPP moves per iteration
      v0 = __builtin_amdgcn_update_dpp(0, v0, CTRL, 0xf, 0xf, false);
      v1 = __builtin_amdgcn_update_dpp(0, v1, CTRL, 0xf, 0xf, false);
      v2 = __builtin_amdgcn_update_dpp(0, v2, CTRL, 0xf, 0xf, false);
      v3 = __builtin_amdgcn_update_dpp(0, v3, CTRL, 0xf, 0xf, false);
    }
  }
  if ((v0 ^ v1 ^ v2 ^ v3) == 0x7fffffff) out[0] = 1.0f;
}

__global__ __launch_bounds__(256) void fma_kernel(float* out, int iters) {
  float v0 = threadIdx.x, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
    }
  }
  if (v0 + v1 + v2 + v3 == 12345.f) out[0] = 1.0f;
}

int main() {
  float* out;
  CHECK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 4096;
  auto run = [&](auto kern, int wgs_per_cu, const char* name) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double insts = 256.0 * wgs_per_cu * 4 * iters * 32;  // wave-instructions
    printf("%-14s %d wavefront(s)/SIMD: %7.3f ms  %6.0f Ginst/s  (%.2f cycles per instruction and SIMD at 2.1 GHz)\n", name, wgs_per_cu,
           best, insts / (best * 1e-3) / 1e9, 2.1e9 * best * 1e-3 / (iters * 32.0 * wgs_per_cu));
  };
  for (int w : {1, 2, 3, 4}) {
    run(fma_kernel, w, "v_fma_f32");
    run(dpp_kernel<0xB1>, w, "quad_perm");      // quad_perm:[1,0,3,2]
    run(dpp_kernel<0x111>, w, "row_shr:1");
    run(dpp_kernel<0x138>, w, "wave_shr:1");
    run(dpp_kernel<0x130>, w, "wave_shl:1");
  }
  return 0;
}
