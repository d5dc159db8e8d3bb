/*
 * nova_math.h — the elementary functions of the backend's numeric contract.
 *
 * Nova's shading code calls sin, cos, atan, atan2, asin, acos, exp and pow (kernels/brdf.hpp:66,72,
 * 131; kernels/transforms.hpp:24-26,46-48; kernels/material.hpp:32).  OpenCL leaves their last
 * bits to the implementation, and a one-ulp difference between a host libm and a device library is
 * enough to send a path elsewhere (a shadow ray towards a sampled light point ends, up to
 * rounding, exactly ON the light's own triangle: kernels/raytrace.hpp:216-220).  To make results
 * reproducible bit for bit between the MI355X kernels and any host check, the backend defines
 * these functions itself, from IEEE +, -, *, /, sqrt and fma only, each operation written out
 * (no expression relies on the compiler's contraction setting).  Argument reduction and
 * polynomial coefficients are those of the Cephes single-precision library (S. Moshier);
 * measured error against float64 on the ranges the path uses: <= 3.1 ulp, pow(x, 1/2.2) <= 7 ulp
 * (tests/test_math.py),
 * inside OpenCL 1.2's full-profile bounds (sin/cos/asin/acos <= 4, atan <= 5, exp <= 3, pow <= 16).
 *
 * Plain C99 / C++; usable in host code and, under hipcc, in device code.
 */
#ifndef NOVA_MATH_H
#define NOVA_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
  #define NOVA_MATH_FN static __host__ __device__ __inline__ __attribute__((always_inline))
#else
  #define NOVA_MATH_FN static inline
#endif

NOVA_MATH_FN float nvm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
NOVA_MATH_FN float nvm_abs(float a) { return __builtin_fabsf(a); }
NOVA_MATH_FN uint32_t nvm_bits(float a) { uint32_t u; memcpy(&u, &a, 4); return u; }
NOVA_MATH_FN float nvm_from_bits(uint32_t u) { float a; memcpy(&a, &u, 4); return a; }

/* r = x - k*pi/2 in three fma steps (Cody-Waite), k = nearest integer to x*2/pi; good for
 * |x| up to a few thousand, far beyond the [0, 2*pi] the path uses. */
NOVA_MATH_FN float nvm_reduce_pio2(float x, int* quadrant) {
  const float k = __builtin_rintf(x * 0.63661977236758134308f);
  float r = nvm_fma(-k, 1.5703125f, x);
  r = nvm_fma(-k, 4.83751296997070312500e-4f, r);
  r = nvm_fma(-k, 7.54978995489188216e-8f, r);
  *quadrant = (int) k;
  return r;
}
NOVA_MATH_FN float nvm_sin_poly(float r) {
  const float s = r * r;
  float p = nvm_fma(-1.9515295891e-4f, s, 8.3321608736e-3f);
  p = nvm_fma(p, s, -1.6666654611e-1f);
  const float sr = s * r;
  return nvm_fma(p, sr, r);
}
NOVA_MATH_FN float nvm_cos_poly(float r) {
  const float s = r * r;
  float p = nvm_fma(2.443315711809948e-5f, s, -1.388731625493765e-3f);
  p = nvm_fma(p, s, 4.166664568298827e-2f);
  const float ss = s * s;
  const float q = nvm_fma(-0.5f, s, 1.0f);
  return nvm_fma(p, ss, q);
}
NOVA_MATH_FN float nova_sinf(float x) {
  int q;
  const float r = nvm_reduce_pio2(x, &q);
  const float v = (q & 1) ? nvm_cos_poly(r) : nvm_sin_poly(r);
  return (q & 2) ? -v : v;
}
NOVA_MATH_FN float nova_cosf(float x) {
  int q;
  const float r = nvm_reduce_pio2(x, &q);
  const float v = (q & 1) ? nvm_sin_poly(r) : nvm_cos_poly(r);
  return ((q + 1) & 2) ? -v : v;
}

NOVA_MATH_FN float nova_atanf(float x) {
  const float a = nvm_abs(x);
  float y0 = 0.0f, z = a;
  if (a > 2.414213562373095f) {
    y0 = 1.5707963267948966f;
    z = -1.0f / a;
  } else if (a > 0.4142135623730950f) {
    y0 = 0.7853981633974483f;
    const float num = a - 1.0f, den = a + 1.0f;
    z = num / den;
  }
  const float s = z * z;
  float p = nvm_fma(8.05374449538e-2f, s, -1.38776856032e-1f);
  p = nvm_fma(p, s, 1.99777106478e-1f);
  p = nvm_fma(p, s, -3.33329491539e-1f);
  const float sz = s * z;
  const float t = nvm_fma(p, sz, z);
  const float r = y0 + t;
  return (nvm_bits(x) >> 31) ? -r : r;
}

NOVA_MATH_FN float nova_atan2f(float y, float x) {
  const float pi = 3.14159265358979323846f, pio2 = 1.5707963267948966f;
  if (x == 0.0f) {
    if (y > 0.0f) return pio2;
    if (y < 0.0f) return -pio2;
    return 0.0f;
  }
  const float q = y / x;
  const float a = nova_atanf(q);
  if (x > 0.0f) return a;
  return (nvm_bits(y) >> 31) ? a - pi : a + pi;
}

NOVA_MATH_FN float nova_asinf(float x) {
  const float a = nvm_abs(x);
  if (a > 1.0f) return nvm_from_bits(0x7fc00000u);
  float z, w;
  const int big = a > 0.5f;
  if (big) {
    const float om = 1.0f - a;
    z = 0.5f * om;
    w = __builtin_sqrtf(z);
  } else {
    z = a * a;
    w = a;
  }
  float p = nvm_fma(4.2163199048e-2f, z, 2.4181311049e-2f);
  p = nvm_fma(p, z, 4.5470025998e-2f);
  p = nvm_fma(p, z, 7.4953002686e-2f);
  p = nvm_fma(p, z, 1.6666752422e-1f);
  const float zw = z * w;
  float r = nvm_fma(p, zw, w);
  if (big) r = nvm_fma(-2.0f, r, 1.5707963267948966f);
  return (nvm_bits(x) >> 31) ? -r : r;
}

NOVA_MATH_FN float nova_acosf(float x) {
  if (x < -0.5f) {
    const float h = nvm_fma(0.5f, x, 0.5f);
    const float s = nova_asinf(__builtin_sqrtf(h));
    return nvm_fma(-2.0f, s, 3.14159265358979323846f);
  }
  if (x > 0.5f) {
    const float h = nvm_fma(-0.5f, x, 0.5f);
    const float s = nova_asinf(__builtin_sqrtf(h));
    return 2.0f * s;
  }
  return 1.5707963267948966f - nova_asinf(x);
}

/* 2^n * m for the results below (m in [0.5, 2], n in [-150, 128]) */
NOVA_MATH_FN float nvm_scale2(float m, int n) {
  if (n > 127) { m = m * 1.7014118346046923e38f; n -= 127; }      /* 2^127 */
  if (n < -126) { m = m * 1.1754943508222875e-38f * 16777216.0f; n += 102; } /* 2^-126 * 2^24 */
  if (n > 127) n = 127;
  if (n < -126) return 0.0f;
  return m * nvm_from_bits((uint32_t) (n + 127) << 23);
}

NOVA_MATH_FN float nova_expf(float x) {
  if (!(x < 88.72283905206835f)) return x != x ? x : nvm_from_bits(0x7f800000u);
  if (x < -103.972077083991796f) return 0.0f;
  const float t = nvm_fma(x, 1.44269504088896341f, 0.5f);
  const float n = __builtin_floorf(t);
  float r = nvm_fma(-n, 0.693359375f, x);
  r = nvm_fma(-n, -2.12194440e-4f, r);
  const float z = r * r;
  float p = nvm_fma(1.9875691500e-4f, r, 1.3981999507e-3f);
  p = nvm_fma(p, r, 8.3334519073e-3f);
  p = nvm_fma(p, r, 4.1665795894e-2f);
  p = nvm_fma(p, r, 1.6666665459e-1f);
  p = nvm_fma(p, r, 5.0000001201e-1f);
  const float q = nvm_fma(p, z, r);
  const float m = q + 1.0f;
  return nvm_scale2(m, (int) n);
}

/* natural logarithm for finite x > 0 (subnormals included) */
NOVA_MATH_FN float nova_logf(float x) {
  uint32_t u = nvm_bits(x);
  int e = 0;
  if ((u >> 23) == 0) { /* subnormal: scale up by 2^24 */
    x = x * 16777216.0f;
    u = nvm_bits(x);
    e = -24;
  }
  e += (int) (u >> 23) - 126;
  float m = nvm_from_bits((u & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
  if (m < 0.707106781186547524f) {
    e -= 1;
    m = m + m;
  }
  const float f = m - 1.0f;
  const float z = f * f;
  float p = nvm_fma(7.0376836292e-2f, f, -1.1514610310e-1f);
  p = nvm_fma(p, f, 1.1676998740e-1f);
  p = nvm_fma(p, f, -1.2420140846e-1f);
  p = nvm_fma(p, f, 1.4249322787e-1f);
  p = nvm_fma(p, f, -1.6668057665e-1f);
  p = nvm_fma(p, f, 2.0000714765e-1f);
  p = nvm_fma(p, f, -2.4999993993e-1f);
  p = nvm_fma(p, f, 3.3333331174e-1f);
  const float fz = f * z;
  float y = p * fz;
  const float fe = (float) e;
  y = nvm_fma(-2.12194440e-4f, fe, y);
  y = nvm_fma(-0.5f, z, y);
  const float r = f + y;
  return nvm_fma(0.693359375f, fe, r);
}

/* pow for the path's two uses: x >= 0.  y == 5 (Schlick's Fresnel term, kernels/brdf.hpp:131)
 * is an exact product chain; anything else (the 1/2.2 gamma, kernels/transforms.hpp:26) goes
 * through exp(y log x). */
NOVA_MATH_FN float nova_powf(float x, float y) {
  if (y == 5.0f) {
    const float x2 = x * x;
    const float x4 = x2 * x2;
    return x4 * x;
  }
  if (y == 0.0f) return 1.0f;
  if (x != x || y != y) return x + y;
  if (x < 0.0f) return nvm_from_bits(0x7fc00000u);
  if (x == 0.0f) return y > 0.0f ? 0.0f : nvm_from_bits(0x7f800000u);
  if (x > 3.4028234663852886e38f) return y > 0.0f ? x : 0.0f;
  const float l = nova_logf(x);
  return nova_expf(y * l);
}

#endif /* NOVA_MATH_H */
