// hess_devmath.h -- device-side elementary functions of the Hessian/SIFT path (gfx950).
//
// The reference evaluates expf / atan2 / __sincosf / pow / rsqrt / __fdividef / __float2half_rn
// with the CUDA math library (ProgramCU.cu:559,865,1297,1359,1413,1698,1741,1989).  This build
// fixes each of them as an explicit sequence of IEEE binary32 operations (Cephes single-precision
// algorithms: range reduction + Horner polynomial in fmaf), accurate to <= 2 ulp on the ranges the
// path uses, so that results do not depend on a vendor math library.  Compile with
// -ffp-contract=off: every fused multiply-add below is written as fmaf().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hess {

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// e^x, x in [-87, 88]; 0 below (weights there are < 1.7e-38).
__device__ __forceinline__ float dm_expf(float x) {
  // branch-free form of `if (x < -87) return 0; if (x > 88) x = 88;`
  const bool under = x < -87.0f;
  x = (x > 88.0f) ? 88.0f : x;
  x = under ? -87.0f : x;  // keeps the exponent arithmetic in range; the result is discarded
  float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float z = r * r;
  float p = 1.9875691500E-4f;
  p = fmaf(p, r, 1.3981999507E-3f);
  p = fmaf(p, r, 8.3334519073E-3f);
  p = fmaf(p, r, 4.1665795894E-2f);
  p = fmaf(p, r, 1.6666665459E-1f);
  p = fmaf(p, r, 5.0000001201E-1f);
  p = fmaf(p, z, r);
  p = p + 1.0f;
  int e = (int)n + 127;
  const float v = p * u2f((uint32_t)e << 23);
  return under ? 0.0f : v;
}

// a^e from ln(a): the orientation kernel's pow(sigma_step, ds) (ProgramCU.cu:1297).
__device__ __forceinline__ float dm_powf_ln(float ln_a, float e) { return dm_expf(e * ln_a); }

// atan on [0,1]: t + t^3*q(t^2), q of degree 8 (relative-error-weighted least-squares fit).
__device__ __forceinline__ float dm_atan01(float t) {
  float z = t * t;
  float q = -0.0017890612361952662f;
  q = fmaf(q, z, 0.010897884145379066f);
  q = fmaf(q, z, -0.03115503303706646f);
  q = fmaf(q, z, 0.057945046573877335f);
  q = fmaf(q, z, -0.08403480052947998f);
  q = fmaf(q, z, 0.10952533036470413f);
  q = fmaf(q, z, -0.14264392852783203f);
  q = fmaf(q, z, 0.19998574256896973f);
  q = fmaf(q, z, -0.33333301544189453f);
  q = q * z;
  return fmaf(q, t, t);
}

// atan2(y,x) in [-pi, pi]; (0,0) -> 0.
__device__ __forceinline__ float dm_atan2f(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = ax > ay ? ax : ay;
  float mn = ax > ay ? ay : ax;
  if (mx == 0.0f) return 0.0f;
  float r = dm_atan01(mn / mx);
  if (ay > ax) r = 1.57079632679489662f - r;
  if (x < 0.0f) r = 3.14159265358979324f - r;
  if (y < 0.0f) r = -r;
  return r;
}

// ComputeHessian_Kernel for one pixel (ProgramCU.cu:536-559) from its 3x3 neighbourhood v<row><col> of the Gaussian
// level: det-Hessian * sigma^4 and (|gradient| / 2, theta).  The one scalar statement of these lines in the product
// (k_detect.hip's det-H kernels and the level-chain kernel use it; the tile kernel's fused stage is its two-wide form).
__device__ __forceinline__ float dm_deth(float v11, float v12, float v13, float v21, float v22, float v23, float v31,
                                         float v32, float v33, float norm) {
  const float Lxx = fmaf(-2.0f, v22, v21) + v23;      // ProgramCU.cu:536
  const float Lyy = fmaf(-2.0f, v22, v12) + v32;      // :537
  const float Lxy = (v13 - v11 + v31 - v33) * 0.25f;  // :538
  return fmaf(Lxx, Lyy, -(Lxy * Lxy)) * norm;         // :553
}
__device__ __forceinline__ float2 dm_grad_theta(float v12, float v21, float v23, float v32) {
  const float dx = v23 - v21, dy = v32 - v12;         // :556-557
  const float gradient = 0.5f * sqrtf(fmaf(dx, dx, dy * dy));
  return make_float2(gradient, (gradient == 0.0f) ? 0.0f : dm_atan2f(dy, dx));
}

// Two-wide forms of dm_atan01 / dm_atan2f: the same operations per component (IEEE division, fused
// multiply-adds in the same order) and branch-free: for (0,0) the quotient is 0/1 = 0 and every later
// step leaves 0, as the early return of the scalar form does.  The 2-vectors compile to packed FP32
// instructions; on gfx950 those issue at half the rate of the scalar ones (tools/micro/valu_rate.hip:
// v_pk_fma_f32 5.3 vs v_fma_f32 2.7 cycles per wavefront instruction), so this form is about straight-line
// code, not about arithmetic throughput.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f v2_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f v2_splat(float x) { return (v2f){x, x}; }

__device__ __forceinline__ v2f dm_atan01_x2(v2f t) {
  const v2f z = t * t;
  v2f q = v2_splat(-0.0017890612361952662f);
  q = v2_fma(q, z, v2_splat(0.010897884145379066f));
  q = v2_fma(q, z, v2_splat(-0.03115503303706646f));
  q = v2_fma(q, z, v2_splat(0.057945046573877335f));
  q = v2_fma(q, z, v2_splat(-0.08403480052947998f));
  q = v2_fma(q, z, v2_splat(0.10952533036470413f));
  q = v2_fma(q, z, v2_splat(-0.14264392852783203f));
  q = v2_fma(q, z, v2_splat(0.19998574256896973f));
  q = v2_fma(q, z, v2_splat(-0.33333301544189453f));
  q = q * z;
  return v2_fma(q, t, t);
}

__device__ __forceinline__ v2f dm_atan2f_x2(v2f y, v2f x) {
  const v2f ax = {fabsf(x.x), fabsf(x.y)}, ay = {fabsf(y.x), fabsf(y.y)};
  const bool g0 = ax.x > ay.x, g1 = ax.y > ay.y;
  const v2f mx = {g0 ? ax.x : ay.x, g1 ? ax.y : ay.y};
  const v2f mn = {g0 ? ay.x : ax.x, g1 ? ay.y : ax.y};
  const v2f den = {mx.x == 0.0f ? 1.0f : mx.x, mx.y == 0.0f ? 1.0f : mx.y};
  v2f r = dm_atan01_x2(mn / den);
  const v2f r1 = v2_splat(1.57079632679489662f) - r;
  r = (v2f){ay.x > ax.x ? r1.x : r.x, ay.y > ax.y ? r1.y : r.y};
  const v2f r2 = v2_splat(3.14159265358979324f) - r;
  r = (v2f){x.x < 0.0f ? r2.x : r.x, x.y < 0.0f ? r2.y : r.y};
  r = (v2f){y.x < 0.0f ? -r.x : r.x, y.y < 0.0f ? -r.y : r.y};
  return r;
}

__device__ __forceinline__ void dm_sincosf(float a, float* s, float* c) {
  float k = rintf(a * 0.636619772367581343f);
  float r = fmaf(k, -1.5703125f, a);
  r = fmaf(k, -4.837512969970703125e-4f, r);
  r = fmaf(k, -7.54978995489188216e-8f, r);
  float z = r * r;
  float ps = -1.9515295891E-4f;
  ps = fmaf(ps, z, 8.3321608736E-3f);
  ps = fmaf(ps, z, -1.6666654611E-1f);
  ps = ps * z;
  ps = fmaf(ps, r, r);
  float pc = 2.443315711809948E-005f;
  pc = fmaf(pc, z, -1.388731625493765E-003f);
  pc = fmaf(pc, z, 4.166664568298827E-002f);
  pc = pc * z;
  pc = fmaf(pc, z, fmaf(-0.5f, z, 1.0f));
  int q = ((int)k) & 3;
  float sv = (q & 1) ? pc : ps;
  float cv = (q & 1) ? ps : pc;
  if (q == 1) { cv = -cv; }
  else if (q == 2) { sv = -sv; cv = -cv; }
  else if (q == 3) { sv = -sv; }
  *s = sv;
  *c = cv;
}

// b / 255.0f for an integer-valued 0 <= b <= 255 (u8 luminance -> [0,1], GLTexImage.cpp:828), correctly rounded
// without the division sequence: q0 = b * fl(1/255) is within an ulp, the residual b - 255*q0 is exact in one fma,
// and one more fma adds the correction.  Equal to the IEEE quotient for all 256 inputs (checked exhaustively on the
// host in tests/test_oracle_math.py and on the device in tests/test_gpu_parity.py); 3 instructions instead of 11.
__device__ __forceinline__ float dm_u8_unit(float b) {
  const float r = 1.0f / 255.0f;  // constant-folded: fl(1/255)
  const float q0 = b * r;
  const float e = fmaf(q0, -255.0f, b);
  return fmaf(e, r, q0);
}

// binary32 -> binary16 bits, round to nearest even (__float2half_rn, ProgramCU.cu:865).
__device__ __forceinline__ uint32_t dm_f2h(float f) {
  uint32_t x = f2u(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x > 0x7f800000u) return sign | 0x7e00u;
  if (x >= 0x477ff000u) return sign | 0x7c00u;
  if (x >= 0x38800000u) {
    uint32_t m = x - 0x38000000u;
    m += 0x00000fffu + ((m >> 13) & 1u);
    return sign | (m >> 13);
  }
  if (x < 0x33000000u) return sign;
  uint32_t e = x >> 23;
  uint32_t m = (x & 0x007fffffu) | 0x00800000u;
  uint32_t shift = 126u - e;
  uint32_t half_lsb = 1u << shift;
  uint32_t rem = m & (half_lsb - 1u);
  uint32_t q = m >> shift;
  uint32_t halfway = half_lsb >> 1;
  if (rem > halfway || (rem == halfway && (q & 1u))) q++;
  return sign | q;
}

// binary16 bits -> binary32 (exact widening; __half2float / GlobalUtil.cpp:588-621).
__device__ __forceinline__ float dm_h2f(uint32_t h) {
  uint32_t sign = (h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  if (e == 0) {
    if (m == 0) return u2f(sign);
    int sh = 0;
    while (!(m & 0x400u)) { m <<= 1; sh++; }
    m &= 0x3ffu;
    return u2f(sign | ((uint32_t)(113 - sh) << 23) | (m << 13));
  }
  if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
  return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

}  // namespace hess
