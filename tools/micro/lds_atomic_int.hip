// lds_atomic_int.hip -- LDS integer atomics (ds_add_u32 / ds_add_u64, no return) and a lane-private read-modify-write
// (ds_read_b32 + v_fma + ds_write_b32; inline asm so that the compiler keeps the accesses) on gfx950, by the number of
// lanes of one wave-instruction that hit the SAME address.  ds_add_f32 takes ~170-225 LDS cycles per wave-instruction
// whatever the addresses (lds_atomic.hip): a fixed-point accumulation through integer atomics is the alternative.
//   hipcc -O3 --offload-arch=gfx950 -o lds_atomic_int lds_atomic_int.hip && ./lds_atomic_int
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// MODE 0: ds_add_u32; 1: ds_add_u64; 2: private RMW (read, fma, write); 3: ds_add_rtn_u32 (value used)
// MODE 4 / 5: ds_add_u32 / ds_add_u64 with only some lanes active (SAME = pattern: 1 = odd lanes, 2 = a scattered half, 3 = lanes 0..31, 4 = a scattered quarter)
template <int MODE, int SAME>
__global__ __launch_bounds__(256) void lds_kernel(unsigned* out, int iters) {
  __shared__ unsigned long long h[4][8][64 + 8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = lane; i < 8 * 72; i += 64) (&h[wv][0][0])[i] = 0ull;
  __syncthreads();
  const int slot = (MODE == 2 || MODE >= 4) ? lane : lane / SAME;
  unsigned v = 1u + lane;
  unsigned acc = 0;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (MODE == 0) {
        unsigned* p = reinterpret_cast<unsigned*>(&h[wv][k][0]) + slot;
        (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 1) {
        (void)__hip_atomic_fetch_add(&h[wv][k][slot], (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 2) {
        volatile float* p = reinterpret_cast<volatile float*>(&h[wv][k][0]) + slot;
        *p = fmaf((float)v, 0.5f, *p);
      } else if (MODE == 3) {
        unsigned* p = reinterpret_cast<unsigned*>(&h[wv][k][0]) + slot;
        acc += __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        const unsigned hsh = ((unsigned)lane * 2654435761u) >> 13;
        const bool on = SAME == 1 ? (lane & 1) : SAME == 2 ? (hsh & 1) : SAME == 3 ? (lane < 32) : ((hsh & 3) == 0);
        if (on) {
          if (MODE == 4) (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned*>(&h[wv][k][0]) + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else (void)__hip_atomic_fetch_add(&h[wv][k][lane], (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    v += 3u;
  }
  __syncthreads();
  unsigned long long s = acc;
  for (int k = 0; k < 8; k++) s += h[wv][k][lane];
  if (s == 0x123456789ull) out[0] = (unsigned)s;
}

int main() {
  unsigned* out;
  CHECK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 2048;
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
    const double per_cu = (double)wgs_per_cu * 4 * iters * 8;
    printf("%-40s %d wg/CU: %7.3f ms  %6.2f cycles per wave-instruction and CU (2.1 GHz)\n", name, wgs_per_cu, best,
           2.1e9 * best * 1e-3 / per_cu);
  };
  for (int w : {1, 4}) {
    run(lds_kernel<2, 1>, w, "ds_read_b32 + fma + ds_write_b32 (private)");
    run(lds_kernel<0, 1>, w, "ds_add_u32, 64 addresses");
    run(lds_kernel<0, 2>, w, "ds_add_u32, 2 lanes per address");
    run(lds_kernel<0, 4>, w, "ds_add_u32, 4 lanes per address");
    run(lds_kernel<0, 8>, w, "ds_add_u32, 8 lanes per address");
    run(lds_kernel<0, 16>, w, "ds_add_u32, 16 lanes per address");
    run(lds_kernel<0, 64>, w, "ds_add_u32, 64 lanes per address");
    run(lds_kernel<1, 1>, w, "ds_add_u64, 64 addresses");
    run(lds_kernel<1, 4>, w, "ds_add_u64, 4 lanes per address");
    run(lds_kernel<1, 16>, w, "ds_add_u64, 16 lanes per address");
    run(lds_kernel<3, 1>, w, "ds_add_rtn_u32, 64 addresses");
    run(lds_kernel<3, 8>, w, "ds_add_rtn_u32, 8 lanes per address");
    run(lds_kernel<4, 1>, w, "ds_add_u32, odd lanes only");
    run(lds_kernel<4, 2>, w, "ds_add_u32, a scattered half");
    run(lds_kernel<4, 3>, w, "ds_add_u32, lanes 0..31");
    run(lds_kernel<4, 4>, w, "ds_add_u32, a scattered quarter");
    run(lds_kernel<5, 1>, w, "ds_add_u64, odd lanes only");
    run(lds_kernel<5, 2>, w, "ds_add_u64, a scattered half");
    run(lds_kernel<5, 3>, w, "ds_add_u64, lanes 0..31");
    run(lds_kernel<5, 4>, w, "ds_add_u64, a scattered quarter");
  }
  return 0;
}
