// Cost of __threadfence() (agent-scope release + acquire: L2 write-back / invalidate on a multi-XCD part) per workgroup.
// 8192 workgroups x 256 threads; a thread spins ~10 us of fmas, stores 16 bytes, [fences], one atomic per workgroup.
//   hipcc --offload-arch=gfx950 -O2 -o fence_cost fence_cost.hip && ./fence_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

template <int FENCE>
__global__ void k(float4* out, int* counter, int iters) {
  float v = threadIdx.x;
  for (int i = 0; i < iters; i++) v = fmaf(v, 1.0000001f, 0.5f);
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = make_float4(v, v, v, v);
  if (FENCE == 1) __threadfence();
  if (FENCE == 2) __threadfence_system();
  if (threadIdx.x == 0) atomicAdd(counter + (blockIdx.x >> 4), 1);
}

int main() {
  float4* out; int* cnt;
  hipMalloc(&out, (size_t)8192 * 256 * 16); hipMalloc(&cnt, 4096);
  hipMemset(cnt, 0, 4096);
  for (int iters : {0, 4000}) {
    for (int f = 0; f < 3; f++) {
      double best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        if (f == 0) hipLaunchKernelGGL(k<0>, dim3(8192), dim3(256), 0, 0, out, cnt, iters);
        if (f == 1) hipLaunchKernelGGL(k<1>, dim3(8192), dim3(256), 0, 0, out, cnt, iters);
        if (f == 2) hipLaunchKernelGGL(k<2>, dim3(8192), dim3(256), 0, 0, out, cnt, iters);
        hipDeviceSynchronize();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        best = ms < best ? ms : best;
      }
      printf("iters %d, %s: %.3f ms\n", iters, f == 0 ? "no fence" : f == 1 ? "__threadfence()" : "__threadfence_system()", best);
    }
  }
  return 0;
}
