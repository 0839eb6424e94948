/*
 * hess_math_ref.h -- ORACLE-SIDE (test infrastructure) definition of the elementary
 * functions the reference evaluates with the CUDA math library, which is proprietary and
 * not present here.  Each is a fixed sequence of IEEE-754 binary32
 * operations (+, *, fmaf, /, rintf) so that a second implementation of the same sequence --
 * the HIP kernels' own copy in hessgpu_amd/csrc/hess_devmath.h -- gives bit-identical
 * results.  Algorithms: Cephes single-precision expf / sinf / cosf (S. Moshier; atan: own fit,
 * public algorithm and coefficients), restated.  Accuracy is checked against libm in
 * tests/test_oracle_math.py (<= 2 ulp on the ranges the hot path uses), i.e. inside the
 * error bound CUDA documents for expf/atan2f/sinf/cosf.
 *
 * Models:  expf (ProgramCU.cu:1359,1741) -> om_expf ;  atan2 (ProgramCU.cu:559) -> om_atan2f ;
 *          __sincosf (ProgramCU.cu:1698) -> om_sincosf ;  pow (ProgramCU.cu:1297) -> om_powf_ln ;
 *          __float2half_rn / __half2float (ProgramCU.cu:865,2269) -> om_f2h / om_h2f ;
 *          rsqrt (ProgramCU.cu:1989) -> 1/sqrtf ;  __fdividef (ProgramCU.cu:38) -> IEEE division.
 */
#ifndef HESS_MATH_REF_H
#define HESS_MATH_REF_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t om_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float om_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* e^x for x in [-87, 88]; 0 below -87 (the weights there are < 1.7e-38). */
static inline float om_expf(float x) {
  if (x < -87.0f) return 0.0f;
  if (x > 88.0f) x = 88.0f;
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
  int e = (int)n + 127; /* 1..254 on the clamped range */
  return p * om_u2f((uint32_t)e << 23);
}

/* a^e given ln(a) rounded to float: the reference's pow(sigma_step, ds). */
static inline float om_powf_ln(float ln_a, float e) { return om_expf(e * ln_a); }

/* atan on [0,1] (argument already reduced to min/max): t + t^3*q(t^2), q = degree-8 least-squares
 * fit on Chebyshev nodes (relative-error weighted); measured <= 1.6 ulp, no range reduction. */
static inline float om_atan01(float t) {
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

/* atan2(y, x), result in [-pi, pi]; (0,0) -> 0. */
static inline float om_atan2f(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = ax > ay ? ax : ay;
  float mn = ax > ay ? ay : ax;
  if (mx == 0.0f) return 0.0f;
  float r = om_atan01(mn / mx);
  if (ay > ax) r = 1.57079632679489662f - r;
  if (x < 0.0f) r = 3.14159265358979324f - r;
  if (y < 0.0f) r = -r;
  return r;
}

/* sin and cos of a in [-8, 8] (the path only uses [0, 2pi]). */
static inline void om_sincosf(float a, float* s, float* c) {
  float k = rintf(a * 0.636619772367581343f); /* 2/pi */
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

/* binary32 -> binary16, round to nearest even (CUDA __float2half_rn). */
static inline uint16_t om_f2h(float f) {
  uint32_t x = om_f2u(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);  /* NaN */
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* >= 65520 -> inf */
  if (x >= 0x38800000u) {                                  /* normal half */
    uint32_t m = x - 0x38000000u;                          /* rebias exponent 127 -> 15 */
    m += 0x00000fffu + ((m >> 13) & 1u);
    return (uint16_t)(sign | (m >> 13));
  }
  if (x < 0x33000000u) return (uint16_t)sign;              /* < 2^-25 -> 0 */
  /* subnormal half: value = m * 2^(e-150), target unit 2^-24 */
  uint32_t e = x >> 23;
  uint32_t m = (x & 0x007fffffu) | 0x00800000u;
  uint32_t shift = 126u - e;                               /* 14..24 */
  uint32_t half_lsb = 1u << shift;
  uint32_t rem = m & (half_lsb - 1u);
  uint32_t q = m >> shift;
  uint32_t halfway = half_lsb >> 1;
  if (rem > halfway || (rem == halfway && (q & 1u))) q++;
  return (uint16_t)(sign | q);
}

/* binary16 -> binary32 bits (reference host routine half2float, GlobalUtil.cpp:588-621,
 * and device __half2float; both are the exact IEEE widening). */
static inline float om_h2f(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  if (e == 0) {
    if (m == 0) return om_u2f(sign);
    int sh = 0;
    while (!(m & 0x400u)) { m <<= 1; sh++; }
    m &= 0x3ffu;
    return om_u2f(sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13));
  }
  if (e == 31) return om_u2f(sign | 0x7f800000u | (m << 13));
  return om_u2f(sign | ((e + 112u) << 23) | (m << 13));
}

#endif /* HESS_MATH_REF_H */
