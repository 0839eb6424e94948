// hbm_mix_layout.hip -- does the LAYOUT of a level launch's four output streams matter to HBM?
//
//   hipcc -O3 --offload-arch=gfx950 -o hbm_mix_layout hbm_mix_layout.hip && ./hbm_mix_layout
//
// hbm_mix.hip: one read to four writes into four separate planes runs at 4.6 TB/s, one write stream alone at 6.1.  Here the
// four outputs of an element go (a) to four planes (the pipeline's layout), (b) to ONE region, interleaved in chunks of CH
// bytes per stream (a tile-major layout: a workgroup's level / det-H / gradient / theta tiles next to each other).
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// CHE = chunk size in float4 elements; 0 = separate planes, `stride` elements apart
template <int CHE>
__global__ __launch_bounds__(256) void mix_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, int do_read, int nontemporal,
                                                  size_t stride = 0) {
  if (stride == 0) stride = n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    if (do_read) v = src[i];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const size_t at = CHE == 0 ? (size_t)k * stride + i : (i / CHE) * (4 * (size_t)CHE) + (size_t)k * CHE + (i % CHE);
      const float4 o = make_float4(v.x + k, v.y, v.z, v.w);
      if (nontemporal) {
        __builtin_nontemporal_store(o.x, &dst[at].x); __builtin_nontemporal_store(o.y, &dst[at].y);
        __builtin_nontemporal_store(o.z, &dst[at].z); __builtin_nontemporal_store(o.w, &dst[at].w);
      } else {
        dst[at] = o;
      }
    }
  }
}

int main() {
  const size_t n = (size_t)32 << 20;  // 32 Mi float4 = 512 MB per stream
  float4 *src, *dst;
  CHECK(hipMalloc(&src, n * 16));
  CHECK(hipMalloc(&dst, 4 * n * 16 + (64 << 20)));
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
    printf("%-64s %8.3f ms  %7.0f GB/s total\n", name, best, bytes / (best * 1e-3) / 1e9);
    return 0;
  };
  const dim3 g(2048), b(256);
  for (int rd = 1; rd >= 0; rd--) {
    const double bytes = n * (rd ? 80.0 : 64.0);
    printf(rd ? "-- 1 read : 4 writes --\n" : "-- 4 writes, no read --\n");
    time([&] { hipLaunchKernelGGL(mix_kernel<0>, g, b, 0, 0, src, dst, n, rd, 0); }, bytes, "four planes");
    for (size_t pad : {(size_t)256, (size_t)4096, (size_t)(64 << 10) + 4096, (size_t)(1 << 20) + (68 << 10), (size_t)(5 << 20) + (332 << 10) + 256}) {
      char name[96];
      snprintf(name, sizeof name, "four planes, 512 MiB + %zu B apart", pad);
      time([&] { hipLaunchKernelGGL(mix_kernel<0>, g, b, 0, 0, src, dst, n, rd, 0, n + pad / 16); }, bytes, name);
    }
    time([&] { hipLaunchKernelGGL(mix_kernel<64>, g, b, 0, 0, src, dst, n, rd, 0); }, bytes, "one region, 1 KB chunks per stream (a wavefront's store)");
    time([&] { hipLaunchKernelGGL(mix_kernel<256>, g, b, 0, 0, src, dst, n, rd, 0); }, bytes, "one region, 4 KB chunks (a workgroup's store)");
    time([&] { hipLaunchKernelGGL(mix_kernel<512>, g, b, 0, 0, src, dst, n, rd, 0); }, bytes, "one region, 8 KB chunks (a 64x32 tile)");
    time([&] { hipLaunchKernelGGL(mix_kernel<4096>, g, b, 0, 0, src, dst, n, rd, 0); }, bytes, "one region, 64 KB chunks");
    time([&] { hipLaunchKernelGGL(mix_kernel<0>, g, b, 0, 0, src, dst, n, rd, 1); }, bytes, "four planes, non-temporal stores");
    time([&] { hipLaunchKernelGGL(mix_kernel<256>, g, b, 0, 0, src, dst, n, rd, 1); }, bytes, "one region, 4 KB chunks, non-temporal stores");
  }
  return 0;
}
