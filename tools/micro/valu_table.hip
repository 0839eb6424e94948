// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of common gfx950 VALU instructions.
// Each kernel runs 16 independent copies of one instruction per loop iteration via inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d\n", (int)e_); return 1; } } while (0)

#define KERNEL(NAME, ASM)                                                              \
  __global__ __launch_bounds__(256) void NAME(float* out, float a, float b, int iters) { \
    float x[16];                                                                       \
    _Pragma("unroll") for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i + a; \
    for (int it = 0; it < iters; it++) {                                               \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b)); \
    }                                                                                  \
    float s = 0;                                                                       \
    _Pragma("unroll") for (int i = 0; i < 16; i++) s += x[i];                           \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                           \
  }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_addu, "v_add_u32 %0, %0, %1")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_mul24, "v_mul_i32_i24 %0, %0, %1")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_cvtfi, "v_cvt_f32_i32 %0, %0")
KERNEL(k_cvtif, "v_cvt_i32_f32 %0, %0")
KERNEL(k_floor, "v_floor_f32 %0, %0")
KERNEL(k_rndne, "v_rndne_f32 %0, %0")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_exp, "v_exp_f32 %0, %0")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL(k_cmpx64, "v_cmp_lt_f32 s[40:41], %0, %1")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_dppq, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_readlane, "v_readlane_b32 s40, %0, 3")
KERNEL(k_divscale, "v_div_scale_f32 %0, vcc, %0, %1, %0")
KERNEL(k_divfixup, "v_div_fixup_f32 %0, %0, %1, %2")
KERNEL(k_subabs, "v_sub_f32_e64 %0, |%0|, %1")

typedef void (*kern_t)(float*, float, float, int);
struct Entry { const char* name; kern_t fn; };

int main() {
  float* d;
  CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
  const Entry tab[] = {
    {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_max_f32", k_max},
    {"v_max3_f32", k_max3}, {"v_med3_f32", k_med3}, {"v_add_u32", k_addu}, {"v_and_b32", k_and},
    {"v_lshlrev_b32", k_lshl}, {"v_lshl_add_u32", k_lshladd}, {"v_add3_u32", k_add3}, {"v_mul_i32_i24", k_mul24},
    {"v_mad_u32_u24", k_mad24}, {"v_mul_lo_u32", k_mullo}, {"v_cvt_f32_i32", k_cvtfi}, {"v_cvt_i32_f32", k_cvtif},
    {"v_floor_f32", k_floor}, {"v_rndne_f32", k_rndne}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_exp_f32", k_exp},
    {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp_lt_f32 sgpr", k_cmpx64}, {"v_bcnt_u32_b32", k_bcnt},
    {"v_mbcnt_lo", k_mbcnt}, {"v_mov_dpp wave_shr", k_dpp}, {"v_mov_dpp quad_perm", k_dppq}, {"v_mov_b32", k_mov},
    {"v_readlane_b32", k_readlane}, {"v_div_scale_f32", k_divscale}, {"v_div_fixup_f32", k_divfixup},
    {"v_sub_f32 |abs|", k_subabs},
  };
  const int blocks = 256 * 8, iters = 2000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  // reference: clock estimate from v_fma (documented 2 passes for wave64 on a 32-lane FP32 pipe is an assumption)
  for (const Entry& e : tab) {
    hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 10);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = blocks * 4.0 * iters * 16.0;
    printf("%-22s %.3f ms  %.2f cycles/instr/SIMD @2.4GHz\n", e.name, ms, ms * 1e-3 * 2.4e9 / (instr / 1024.0));
  }
  return 0;
}
