// valu_peak.hip -- what the vector ALUs of an MI355X sustain, per instruction class, occupancy and load.
//
//   hipcc -O3 --offload-arch=gfx950 -o valu_peak valu_peak.hip && ./valu_peak
//
// Reconciles tools/micro/valu_table.hip (3.0-3.5 "cycles" per v_fma_f32 at 8 wavefronts per SIMD, quoted at a nominal
// 2.4 GHz from wall time) with MI355X_MICROARCH.md (2 cycles per wave64 instruction and SIMD; 4 for one wavefront
// alone): every wavefront stamps s_memtime (shader cycles) and s_memrealtime (100 MHz, constant) around its loop, so
// the table separates CYCLES per instruction from the CLOCK the chip holds while the kernel runs:
//   cyc/inst/SIMD = median wave cycles / (instructions per wave x wavefronts per SIMD)
//   clock         = wave cycles / wave real time x 100 MHz
//   Ginst/s       = SIMDs in use x clock / (cyc/inst/SIMD)
// for 1, 2, 4 wavefronts per SIMD on ONE CU (one workgroup of 256/512/1024 threads) and 1, 2, 4, 8 wavefronts per SIMD on
// ALL CUs (256 or 512 workgroups, residency fixed by LDS size / the CU's thread limit, see cfgs[]).  Each configuration runs back to back
// for about half a second before the launch that is read (the clock settles under load).
// A `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` pass over this program gives the counter-derived clock beside it
// (tools/profile_valu_peak.sh).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Stamp { unsigned long long cyc, rt; };

#define KERNEL(NAME, ASM, INIT)                                                                        \
  __global__ __launch_bounds__(1024) void NAME(Stamp* st, float* out, float a, float b, int iters) {   \
    extern __shared__ float pad[];                                                                     \
    float x[16];                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 16; i++) x[i] = INIT;                                        \
    __syncthreads();                                                                                   \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
    for (int it = 0; it < iters; it++) {                                                               \
      _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                  \
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b)); \
      }                                                                                                \
    }                                                                                                  \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    float s = 0;                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 16; i++) s += x[i];                                          \
    if (s == 12345.678f) out[0] = s + pad[threadIdx.x & 15];                                           \
    if ((threadIdx.x & 63) == 0) {                                                                     \
      const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                      \
      st[w].cyc = c1 - c0; st[w].rt = r1 - r0;                                                         \
    }                                                                                                  \
  }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2", threadIdx.x * 0.001f + i + a)
KERNEL(k_mul, "v_mul_f32 %0, %0, %1", 1.0f + threadIdx.x * 1e-6f + i * 1e-7f)
KERNEL(k_add, "v_add_f32 %0, %0, %1", threadIdx.x * 0.001f + i + a)
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2", threadIdx.x * 0.001f + i + a)
KERNEL(k_cvt, "v_cvt_f32_i32 %0, %0", (float)(threadIdx.x + i))
KERNEL(k_rcp, "v_rcp_f32 %0, %0", 1.0f + threadIdx.x * 0.001f + i)
KERNEL(k_exp, "v_exp_f32 %0, %0", -0.001f * threadIdx.x - i)
KERNEL(k_dppmov, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", threadIdx.x * 0.001f + i + a)

// packed FP32: two registers per operand
__global__ __launch_bounds__(1024) void k_pkfma(Stamp* st, float* out, float a, float b, int iters) {
  extern __shared__ float pad[];
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = (v2){threadIdx.x * 0.001f + i + a, threadIdx.x * 0.002f + i + b};
  const v2 va = {a, a}, vb = {b, b};
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(vb));
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
  if (s == 12345.678f) out[0] = s + pad[threadIdx.x & 15];
  if ((threadIdx.x & 63) == 0) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    st[w].cyc = c1 - c0; st[w].rt = r1 - r0;
  }
}

typedef void (*Kern)(Stamp*, float*, float, float, int);

struct Config { const char* name; int blocks, threads, lds; int wps; bool all; };

int main() {
  int ncu = 0;
  CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  const int max_waves = 512 * 16;
  Stamp* st;
  float* out;
  CHECK(hipMalloc(&st, sizeof(Stamp) * max_waves));
  CHECK(hipMalloc(&out, 1 << 20));
  // Residency is fixed by the launch itself: 100 KB of LDS lets a CU hold ONE workgroup of the 1-, 2- and 4-wavefront
  // configurations (256 workgroups = one per CU); the 8-wavefront configuration is two 1024-thread workgroups per CU,
  // which the 2048-thread limit of a CU caps at exactly two (512 workgroups of 60 KB).  (With 80 KB the first version
  // of this table let some CUs take two workgroups and others none: stamps and wall time disagreed.)
  const Config cfgs[] = {
      {"one CU, 1 wave/SIMD", 1, 256, 0, 1, false},    {"one CU, 2 waves/SIMD", 1, 512, 0, 2, false},
      {"one CU, 4 waves/SIMD", 1, 1024, 0, 4, false},  {"all CUs, 1 wave/SIMD", ncu, 256, 100 * 1024, 1, true},
      {"all CUs, 2 waves/SIMD", ncu, 512, 100 * 1024, 2, true}, {"all CUs, 4 waves/SIMD", ncu, 1024, 100 * 1024, 4, true},
      {"all CUs, 8 waves/SIMD", 2 * ncu, 1024, 60 * 1024, 8, true},
  };
  struct K { const char* name; Kern fn; int per_iter; } kernels[] = {
      {"v_fma_f32", k_fma, 64},   {"v_mul_f32", k_mul, 64},         {"v_add_f32", k_add, 64},
      {"v_max3_f32", k_max3, 64}, {"v_cvt_f32_i32", k_cvt, 64},
      {"v_pk_fma_f32", k_pkfma, 64}, {"v_rcp_f32", k_rcp, 64},      {"v_exp_f32", k_exp, 64},
      {"v_mov_b32_dpp", k_dppmov, 64},
  };
  for (auto& k : kernels) CHECK(hipFuncSetAttribute((const void*)k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  printf("# MI355X vector-ALU issue: cycles per wave64 instruction and SIMD, clock held, sustained rate (CUs: %d)\n", ncu);
  printf("%-16s %-24s %10s %10s %12s %14s %14s\n", "instruction", "configuration", "cyc/inst", "clock GHz", "Ginst/s", "chip-equiv",
         "wall Ginst/s");
  hipEvent_t ev0, ev1;
  CHECK(hipEventCreate(&ev0));
  CHECK(hipEventCreate(&ev1));
  const int iters = 20000;  // 64 x 20000 = 1.28 M instructions per wave: 1-10 ms per launch
  for (auto& k : kernels) {
    for (auto& c : cfgs) {
      const auto t0 = std::chrono::steady_clock::now();
      int launches = 0;
      float wall_ms = 0;
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.4 || launches < 3) {
        CHECK(hipEventRecord(ev0, 0));
        hipLaunchKernelGGL(k.fn, dim3(c.blocks), dim3(c.threads), c.lds, 0, st, out, 1.0001f, 0.5f, iters);
        CHECK(hipEventRecord(ev1, 0));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&wall_ms, ev0, ev1));
        launches++;
      }
      const int nw = c.blocks * c.threads / 64;
      std::vector<Stamp> h(nw);
      CHECK(hipMemcpy(h.data(), st, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
      std::vector<double> cyc(nw), clk(nw);
      for (int i = 0; i < nw; i++) { cyc[i] = (double)h[i].cyc; clk[i] = h[i].rt ? (double)h[i].cyc / (double)h[i].rt * 0.1 : 0.0; }
      std::nth_element(cyc.begin(), cyc.begin() + nw / 2, cyc.end());
      std::nth_element(clk.begin(), clk.begin() + nw / 2, clk.end());
      const double insts = (double)k.per_iter * iters;
      const double cpi = cyc[nw / 2] / (insts * c.wps);
      const double ghz = clk[nw / 2];
      const int simds = (c.all ? ncu : 1) * 4;
      const double rate = simds * ghz / cpi;          // Ginst/s of the SIMDs in use
      const double chip = ncu * 4 * ghz / cpi;        // the same cycles and clock on every SIMD of the chip
      // independent of the stamps: every instruction of the launch over the launch's wall time (hipEvents): if the
      // workgroups of a configuration were not all resident at once this is what shows it
      const double wall = insts * (double)nw / (wall_ms * 1e-3) / 1e9;
      printf("%-16s %-24s %10.2f %10.3f %12.1f %14.1f %14.1f\n", k.name, c.name, cpi, ghz, rate, chip, wall);
      fflush(stdout);
    }
  }
  return 0;
}
