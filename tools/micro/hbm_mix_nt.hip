// hbm_mix_nt.hip -- the 1 read : 4 writes mix of a pyramid level launch with non-temporal stores / loads.
//   hipcc -O3 --offload-arch=gfx950 -o hbm_mix_nt hbm_mix_nt.hip && ./hbm_mix_nt
// Same shape as hbm_mix.hip (16-byte accesses, 256 threads x 2048 workgroups, grid-stride, 512 MB per stream);
// variants: plain, __builtin_nontemporal_store for the writes, non-temporal loads as well.
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NW, bool NTS, bool NTL>
__global__ __launch_bounds__(256) void mix_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    v4f v = NTL ? __builtin_nontemporal_load(src + i) : src[i];
#pragma unroll
    for (int k = 0; k < NW; k++) {
      v4f w = v; w.x += k;
      if (NTS) __builtin_nontemporal_store(w, dst + (size_t)k * n + i);
      else dst[(size_t)k * n + i] = w;
    }
  }
}

int main() {
  const size_t n = (size_t)32 << 20;
  v4f *src, *dst;
  CHECK(hipMalloc(&src, n * 16));
  CHECK(hipMalloc(&dst, 4 * n * 16));
  CHECK(hipMemset(src, 1, n * 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto time = [&](auto launch, double bytes, const char* name) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0, 0);
      launch();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-46s %8.3f ms  %7.0f GB/s total\n", name, best, bytes / (best * 1e-3) / 1e9);
  };
  const dim3 g(2048), b(256);
  time([&] { hipLaunchKernelGGL((mix_kernel<4, false, false>), g, b, 0, 0, src, dst, n); }, n * 80.0, "1 read : 4 writes, plain");
  time([&] { hipLaunchKernelGGL((mix_kernel<4, true, false>), g, b, 0, 0, src, dst, n); }, n * 80.0, "1 read : 4 writes, non-temporal stores");
  time([&] { hipLaunchKernelGGL((mix_kernel<4, true, true>), g, b, 0, 0, src, dst, n); }, n * 80.0, "1 read : 4 writes, non-temporal stores + loads");
  time([&] { hipLaunchKernelGGL((mix_kernel<1, false, false>), g, b, 0, 0, src, dst, n); }, n * 32.0, "copy, plain");
  time([&] { hipLaunchKernelGGL((mix_kernel<1, true, false>), g, b, 0, 0, src, dst, n); }, n * 32.0, "copy, non-temporal stores");
  time([&] { hipLaunchKernelGGL((mix_kernel<1, true, true>), g, b, 0, 0, src, dst, n); }, n * 32.0, "copy, non-temporal stores + loads");
  // smaller working set: 66 MB per stream (one pyramid level of a batch of eight 1080p images), producer then consumer
  const size_t m = (size_t)66 << 16;  // 66 MiB / 16
  time([&] { hipLaunchKernelGGL((mix_kernel<4, false, false>), g, b, 0, 0, src, dst, m); hipLaunchKernelGGL((mix_kernel<4, false, false>), g, b, 0, 0, dst, dst + 4 * m, m); }, m * 160.0, "two dependent 66 MB levels, plain");
  time([&] { hipLaunchKernelGGL((mix_kernel<4, true, false>), g, b, 0, 0, src, dst, m); hipLaunchKernelGGL((mix_kernel<4, true, false>), g, b, 0, 0, dst, dst + 4 * m, m); }, m * 160.0, "two dependent 66 MB levels, nt stores");
  return 0;
}
