// tile_mix.hip -- does the SHAPE of a workgroup's tile matter for the 1 read : 4 writes mix of a pyramid level launch?
//   hipcc -O3 --offload-arch=gfx950 -o tile_mix tile_mix.hip && ./tile_mix
// A plane is 1920 x 1080 floats x 8 images (66 MB).  A workgroup of 256 threads reads a TW x TH tile (TW*TH = 2048
// floats: 16-byte loads, one row of the tile = TW*4 contiguous bytes) of the source plane and writes the same tile of
// four destination planes (the det-H and the two gradient/theta halves optionally with streaming stores, as shipped).
// Tiles are numbered row-major over the image and dealt to XCDs in contiguous eighths like the Gaussian kernel's.
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int TW, bool NT>
__global__ __launch_bounds__(256) void tile_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, int tiles_x,
                                                   int tiles_per_img, int ntiles, size_t plane_all) {
  constexpr int TH = 2048 / TW, G = TW / 4;  // 16-byte groups per tile row
  const int per_xcd = (ntiles + 7) >> 3;
  const int t = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (t >= ntiles) return;
  const int img = t / tiles_per_img, r = t - img * tiles_per_img, ty = r / tiles_x, tx = r - ty * tiles_x;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int g = threadIdx.x + 256 * k;  // 512 groups per tile
    const int row = g / G, col = (g - row * G) * 4;
    const int x = tx * TW + col, y = ty * TH + row;
    if (x < w && y < h) {
      const size_t o = ((size_t)img * h + y) * w + x;
      const v4f v = *reinterpret_cast<const v4f*>(src + o);
      *reinterpret_cast<v4f*>(dst + o) = v;
      v4f a = v; a.x += 1;
      v4f b = v; b.y += 1;
      v4f c = v; c.z += 1;
      if (NT) {
        __builtin_nontemporal_store(a, reinterpret_cast<v4f*>(dst + plane_all + o));
        __builtin_nontemporal_store(b, reinterpret_cast<v4f*>(dst + 2 * plane_all + 2 * o));
        __builtin_nontemporal_store(c, reinterpret_cast<v4f*>(dst + 2 * plane_all + 2 * o + 4));
      } else {
        *reinterpret_cast<v4f*>(dst + plane_all + o) = a;
        *reinterpret_cast<v4f*>(dst + 2 * plane_all + 2 * o) = b;
        *reinterpret_cast<v4f*>(dst + 2 * plane_all + 2 * o + 4) = c;
      }
    }
  }
}

int main() {
  const int w = 1920, h = 1080, B = 8;
  const size_t plane_all = (size_t)w * h * B;
  float *src, *dst;
  CHECK(hipMalloc(&src, plane_all * 4));
  CHECK(hipMalloc(&dst, plane_all * 4 * 4));
  CHECK(hipMemset(src, 1, plane_all * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto run = [&](auto kern, int TW, const char* name) {
    const int TH = 2048 / TW, tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH, per = tiles_x * tiles_y, nt = per * B;
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(kern, dim3(((nt + 7) / 8) * 8), dim3(256), 0, 0, src, dst, w, h, tiles_x, per, nt, plane_all);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-40s %7.1f us  %6.0f GB/s (20 B per pixel)\n", name, best * 1e3, plane_all * 20.0 / (best * 1e-3) / 1e9);
  };
  run(tile_kernel<32, false>, 32, "tile 32 x 64, plain");
  run(tile_kernel<64, false>, 64, "tile 64 x 32, plain");
  run(tile_kernel<128, false>, 128, "tile 128 x 16, plain");
  run(tile_kernel<256, false>, 256, "tile 256 x 8, plain");
  run(tile_kernel<512, false>, 512, "tile 512 x 4, plain");
  run(tile_kernel<32, true>, 32, "tile 32 x 64, streaming side stores");
  run(tile_kernel<64, true>, 64, "tile 64 x 32, streaming side stores");
  run(tile_kernel<128, true>, 128, "tile 128 x 16, streaming side stores");
  run(tile_kernel<256, true>, 256, "tile 256 x 8, streaming side stores");
  run(tile_kernel<512, true>, 512, "tile 512 x 4, streaming side stores");
  return 0;
}
