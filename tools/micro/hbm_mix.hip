// hbm_mix.hip -- what HBM sustains for the read : write mixes of the Gaussian launches.
//
//   hipcc -O3 --offload-arch=gfx950 -o hbm_mix hbm_mix.hip && ./hbm_mix
//
// A pyramid level launch reads 4 B per pixel and writes 4 B (next level) + 4 B (det-H) + 8 B (gradient, theta): one part
// read to four parts written.  The 8 TB/s (spec) / 6.3 TB/s (float4 copy, MI355X_MICROARCH.md) figures are for one
// part read to one part written; this measures read-only, write-only, 1:1, 1:2, 1:3, 1:4 and 1:5 with 16-byte
// accesses, 256 threads x 2048 workgroups, grid-stride, arrays far larger than the Infinity Cache.
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NW>
__global__ __launch_bounds__(256) void mix_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, int do_read) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    if (do_read) v = src[i];
#pragma unroll
    for (int k = 0; k < NW; k++) dst[(size_t)k * n + i] = make_float4(v.x + k, v.y, v.z, v.w);
  }
}

__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ src, float* out, size_t n) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = src[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 12345.f) out[0] = s;
}

int main() {
  const size_t n = (size_t)32 << 20;  // 32 Mi float4 = 512 MB per stream
  float4 *src, *dst;
  float* out;
  CHECK(hipMalloc(&src, n * 16));
  CHECK(hipMalloc(&dst, 5 * n * 16));
  CHECK(hipMalloc(&out, 64));
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
    printf("%-34s %8.3f ms  %7.0f GB/s total\n", name, best, bytes / (best * 1e-3) / 1e9);
    return 0;
  };
  const dim3 g(2048), b(256);
  time([&] { hipLaunchKernelGGL(read_kernel, g, b, 0, 0, src, out, n); }, n * 16.0, "read only");
  time([&] { hipLaunchKernelGGL(mix_kernel<1>, g, b, 0, 0, src, dst, n, 0); }, n * 16.0, "write only");
  time([&] { hipLaunchKernelGGL(mix_kernel<1>, g, b, 0, 0, src, dst, n, 1); }, n * 32.0, "1 read : 1 write (copy)");
  time([&] { hipLaunchKernelGGL(mix_kernel<2>, g, b, 0, 0, src, dst, n, 1); }, n * 48.0, "1 read : 2 writes");
  time([&] { hipLaunchKernelGGL(mix_kernel<3>, g, b, 0, 0, src, dst, n, 1); }, n * 64.0, "1 read : 3 writes");
  time([&] { hipLaunchKernelGGL(mix_kernel<4>, g, b, 0, 0, src, dst, n, 1); }, n * 80.0, "1 read : 4 writes (level + det-H + gradient/theta)");
  time([&] { hipLaunchKernelGGL(mix_kernel<5>, g, b, 0, 0, src, dst, n, 1); }, n * 96.0, "1 read : 5 writes");
  time([&] { hipLaunchKernelGGL(mix_kernel<4>, g, b, 0, 0, src, dst, n, 0); }, n * 64.0, "4 write streams, no read");
  return 0;
}
