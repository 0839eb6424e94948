// Microbenchmark: HBM store bandwidth on gfx950 by store shape, for the output stages of the Gaussian kernel.
//   hipcc -O3 --offload-arch=gfx950 -o store_rate store_rate.hip && ./store_rate
// Every kernel writes the same 1 GiB (rows of 7680 bytes = one 1920-pixel float row) once; only the mapping of
// lanes and instructions to bytes differs:
//   0  16 B per lane, lanes contiguous (1 KiB per instruction)          1  8 B per lane, lanes contiguous (512 B)
//   2  16 B per lane at a lane pitch of 32 B, two instructions fill it  3  16 B per lane at 64 B pitch, four instructions
//   4  8 B per lane: 32 lanes = 256 contiguous B, the two halves of the wavefront in rows 4 apart, 4 rows per lane
//   5  16 B per lane: 16 lanes = 256 contiguous B, the four quarters in rows 2 apart, 2 rows per lane
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr long long ROW = 7680;  // bytes
template <int MODE>
__global__ __launch_bounds__(256) void k(char* out, long long rows) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const float4 v4 = make_float4(1.f, 2.f, 3.f, (float)lane);
  const float2 v2 = make_float2(1.f, (float)lane);
  if (MODE <= 3) {  // a wavefront owns 4 KiB of consecutive bytes
    char* p = out + wave * 4096;
    if (MODE == 0) for (int i = 0; i < 4; i++) *reinterpret_cast<float4*>(p + i * 1024 + lane * 16) = v4;
    if (MODE == 1) for (int i = 0; i < 8; i++) *reinterpret_cast<float2*>(p + i * 512 + lane * 8) = v2;
    if (MODE == 2) for (int h = 0; h < 2; h++) for (int i = 0; i < 2; i++) *reinterpret_cast<float4*>(p + h * 2048 + lane * 32 + i * 16) = v4;
    if (MODE == 3) for (int i = 0; i < 4; i++) *reinterpret_cast<float4*>(p + lane * 64 + i * 16) = v4;
  } else {          // a workgroup owns a 64-pixel x 32-row tile of float rows (256 B x 32 rows), as the Gaussian kernel
    const long long tiles_x = ROW / 256, tile = blockIdx.x;
    const long long ty = tile / tiles_x, tx = tile - ty * tiles_x;
    char* p = out + ty * 32 * ROW + tx * 256;
    if (ty * 32 + 31 >= rows) return;
    const int tid = threadIdx.x;
    if (MODE == 4) { const int cg = tid & 31, rg = tid >> 5; for (int j = 0; j < 4; j++) *reinterpret_cast<float2*>(p + (rg * 4 + j) * ROW + cg * 8) = v2; }
    if (MODE == 5) { const int cg = tid & 15, rg = tid >> 4; for (int j = 0; j < 2; j++) *reinterpret_cast<float4*>(p + (rg * 2 + j) * ROW + cg * 16) = v4; }
  }
}
template <int MODE>
void run(const char* name, char* d, long long bytes) {
  const long long rows = bytes / ROW;
  const int blocks = MODE <= 3 ? (int)(bytes / 4096 / 4) : (int)((rows / 32) * (ROW / 256));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, rows);
  float best = 1e9f;
  for (int r = 0; r < 5; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double written = MODE <= 3 ? blocks * 4.0 * 4096 : (double)blocks * 256 * 32;
  printf("%-58s %.3f ms  %.0f GB/s\n", name, best, written / best / 1e6);
}
int main() {
  const long long bytes = 1ll << 30;
  char* d; hipMalloc(&d, bytes + (1 << 20));
  run<0>("0: 16 B/lane contiguous", d, bytes);
  run<1>("1: 8 B/lane contiguous", d, bytes);
  run<2>("2: 16 B/lane, 32 B lane pitch, 2 instructions", d, bytes);
  run<3>("3: 16 B/lane, 64 B lane pitch, 4 instructions", d, bytes);
  run<4>("4: tile rows, 8 B/lane (32 lanes per 256 B row segment)", d, bytes);
  run<5>("5: tile rows, 16 B/lane (16 lanes per 256 B row segment)", d, bytes);
  return 0;
}
