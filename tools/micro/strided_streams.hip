// strided_streams -- what the extrema scan's ACCESS PATTERN costs, without its arithmetic.
// 8 images x five det-H planes of 1920x1080 floats (the octave-0 part of the scan's input: 332 MB).  A wavefront walks
// down a 128-column strip (8 bytes per lane per plane and row = 512 bytes per load instruction), 26 rows per segment,
// exactly like extrema_stream_kernel; the loaded values only feed a running maximum.
//   layout A  [plane][image][row][col]   the product's layout: a wavefront's five loads of a row are 66 MB apart
//   layout B  [image][row][plane][col]   the five planes of a row are adjacent (38 KB per image row)
//   flat      the same bytes read as one stream of 16-byte accesses (the chip's read rate for this volume)
// Build: hipcc -O3 --offload-arch=gfx950 strided_streams.hip -o strided_streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

constexpr int W = 1920, H = 1080, B = 8, L = 5, ROWS = 24;

template <int LAYOUT, int DEPTH>
__global__ __launch_bounds__(256) void walk(const float* __restrict__ p, float* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int strips = W / 128, segs = H / ROWS;
  const int task = blockIdx.x * 4 + wv;
  const int b = blockIdx.y;
  if (task >= strips * segs) return;
  const int seg = task / strips, strip = task - seg * strips;
  const int x = strip * 128 + 2 * lane;
  float m = 0.0f;
  auto addr = [&](int l, int y) -> const float2* {
    const size_t o = LAYOUT == 0 ? (((size_t)l * B + b) * H + y) * W + x : (((size_t)b * H + y) * L + l) * W + x;
    return reinterpret_cast<const float2*>(p + o);
  };
  float2 v[DEPTH][L];
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
#pragma unroll
    for (int l = 0; l < L; l++) v[d][l] = *addr(l, min(seg * ROWS + d, H - 1));
  for (int r = 0; r < ROWS + 2; r += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      float2 cur[L];
#pragma unroll
      for (int l = 0; l < L; l++) cur[l] = v[d][l];
#pragma unroll
      for (int l = 0; l < L; l++) v[d][l] = *addr(l, min(seg * ROWS + r + d + DEPTH, H - 1));
#pragma unroll
      for (int l = 0; l < L; l++) m = fmaxf(m, fmaxf(cur[l].x, cur[l].y));
    }
  }
  if (m == 12345.0f) out[0] = m;
}

__global__ __launch_bounds__(256) void flat(const float4* __restrict__ p, size_t n, float* out) {
  float m = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = p[i];
    m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
  }
  if (m == 12345.0f) out[0] = m;
}

template <typename F>
float timeit(F f) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) f();
  hipEventRecord(a);
  for (int i = 0; i < 20; i++) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 20.0f * 1000.0f;
}

int main(int argc, char** argv) {
  const size_t n = (size_t)L * B * H * W;
  float *p, *q, *out;
  hipMalloc(&p, n * 4); hipMalloc(&q, (size_t)1 << 30); hipMalloc(&out, 64);
  hipMemset(p, 0, n * 4); hipMemset(q, 0, (size_t)1 << 30);
  const dim3 grid(((W / 128) * (H / ROWS) + 3) / 4, B);
  const double gb = (double)n * 4 / 1e9, gbh = gb * (ROWS + 2.0) / ROWS;
  // Evict the planes from the last-level cache between runs -- two ways (round 5): by a 1 GB FILL
  // (round 4's form: it leaves the caches full of DIRTY lines, so the timed launch may be paying their write-back)
  // or by a 1 GB READ pass (clean lines).  argv[1] = "read" selects the second.
  const bool by_read = argc > 1 && !strcmp(argv[1], "read");
  auto flush = [&] {
    if (by_read) hipLaunchKernelGGL(flat, dim3(256 * 16), dim3(256), 0, 0, (const float4*)q, ((size_t)1 << 30) / 16, out);
    else hipMemsetAsync(q, 1, (size_t)1 << 30, 0);
  };
  printf("eviction between the timed launches: 1 GB %s\n", by_read ? "READ pass (clean lines)" : "FILL (dirty lines)");
  printf("planes: %.1f MB (%.1f MB with the two halo rows per segment of %d)\n", gb * 1e3, gbh * 1e3, ROWS);
  float t;
  t = timeit([&] { flush(); });
  const float tf = t;
  t = timeit([&] { flush(); hipLaunchKernelGGL((walk<0, 1>), grid, dim3(256), 0, 0, p, out); }) - tf;
  printf("layout A, one row in flight : %7.1f us  %.2f TB/s of plane bytes\n", t, gb / t * 1e3);
  t = timeit([&] { flush(); hipLaunchKernelGGL((walk<0, 2>), grid, dim3(256), 0, 0, p, out); }) - tf;
  printf("layout A, two rows in flight: %7.1f us  %.2f TB/s\n", t, gb / t * 1e3);
  t = timeit([&] { flush(); hipLaunchKernelGGL((walk<1, 1>), grid, dim3(256), 0, 0, p, out); }) - tf;
  printf("layout B, one row in flight : %7.1f us  %.2f TB/s\n", t, gb / t * 1e3);
  t = timeit([&] { flush(); hipLaunchKernelGGL((walk<1, 2>), grid, dim3(256), 0, 0, p, out); }) - tf;
  printf("layout B, two rows in flight: %7.1f us  %.2f TB/s\n", t, gb / t * 1e3);
  t = timeit([&] { flush(); hipLaunchKernelGGL(flat, dim3(256 * 16), dim3(256), 0, 0, (const float4*)p, n / 4, out); }) - tf;
  printf("flat 16-byte stream         : %7.1f us  %.2f TB/s\n", t, gb / t * 1e3);
  return 0;
}
