// How many kernels of different streams run at the same time?  N streams, one long single-workgroup kernel each
// (a dependent chain of fmas, ~T us); wall time of the set / time of one = ceil(N / concurrency).
//   hipcc --offload-arch=gfx950 -O2 -o stream_overlap stream_overlap.hip && ./stream_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(float* out, int iters) {
  float v = threadIdx.x;
  for (int i = 0; i < iters; i++) v = fmaf(v, 1.0000001f, 0.5f);
  if (v == 12345.0f) out[0] = v;
}

int main() {
  float* d;
  hipMalloc(&d, 4);
  const int iters = 400000;
  for (int n = 1; n <= 8; n++) {
    std::vector<hipStream_t> st(n);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; rep++) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < 4; k++)
        for (auto& s : st) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, d, iters);
      hipDeviceSynchronize();
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rep) printf("streams %d: %.3f ms for 4 kernels per stream\n", n, ms);
    }
    for (auto& s : st) hipStreamDestroy(s);
  }
  return 0;
}
