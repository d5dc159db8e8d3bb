// Device-side scalar/3-vector helpers for the gfx950 kernels.
//
// Floating-point contract of the backend (DESIGN.md "Numerics"): this code is compiled with
// -ffp-contract=off and every fused multiply-add is written out, so each result is defined by the
// source alone: IEEE binary32, correctly rounded / and sqrt (hipcc default), a*b+c fused exactly
// where Nova's kernel source has it in one expression (its OpenCL build enables -cl-mad-enable,
// src/backend/opencl/accelerator/accelerator.cpp:34-37), dot products as fma chains with the
// last component first, min/max = v_min_f32/v_max_f32 (minNum/maxNum).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nova_math.h"

namespace nova_hip {

struct V3 {
  float x, y, z;
};

#define NV_DEV __device__ __forceinline__

NV_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
NV_DEV float min_(float a, float b) { return __builtin_fminf(a, b); }
NV_DEV float max_(float a, float b) { return __builtin_fmaxf(a, b); }

NV_DEV V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
NV_DEV V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
NV_DEV V3 operator*(V3 a, V3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
NV_DEV V3 operator/(V3 a, V3 b) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
NV_DEV V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
NV_DEV V3 operator*(float s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }
NV_DEV V3 operator/(V3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
NV_DEV V3 operator-(V3 a) { return { -a.x, -a.y, -a.z }; }
NV_DEV V3 splat(float s) { return { s, s, s }; }
NV_DEV V3 fma3(V3 a, V3 b, V3 c) { return { fma_(a.x, b.x, c.x), fma_(a.y, b.y, c.y), fma_(a.z, b.z, c.z) }; }
NV_DEV V3 fma3(V3 a, float b, V3 c) { return { fma_(a.x, b, c.x), fma_(a.y, b, c.y), fma_(a.z, b, c.z) }; }
NV_DEV V3 max3(V3 a, V3 b) { return { max_(a.x, b.x), max_(a.y, b.y), max_(a.z, b.z) }; }
NV_DEV V3 clamp3(V3 a, float lo, float hi) {
  return { min_(max_(a.x, lo), hi), min_(max_(a.y, lo), hi), min_(max_(a.z, lo), hi) };
}
NV_DEV float dot(V3 a, V3 b) { return fma_(a.x, b.x, fma_(a.y, b.y, a.z * b.z)); }
NV_DEV V3 cross(V3 a, V3 b) {
  return { fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)) };
}
NV_DEV float length(V3 a) { return __builtin_sqrtf(dot(a, a)); }
NV_DEV V3 normalize(V3 a) {
  float inv = 1.0f / __builtin_sqrtf(dot(a, a));
  return a * inv;
}
NV_DEV V3 xyz(float4 v) { return { v.x, v.y, v.z }; }

struct M3 {
  V3 x, y, z;
};
NV_DEV V3 mul(const M3& m, V3 v) { return { dot(m.x, v), dot(m.y, v), dot(m.z, v) }; }
NV_DEV M3 transpose(const M3& m) {
  return { { m.x.x, m.y.x, m.z.x }, { m.x.y, m.y.y, m.z.y }, { m.x.z, m.y.z, m.z.z } };
}

// Wang hash and xorshift32 exactly as kernels/random.hpp:11-26 (uint32 arithmetic; the float
// conversion is round-to-nearest and the divisor (float)UINT_MAX is 2^32, so rand is in [0,1]).
NV_DEV uint32_t wang_hash(uint32_t s) {
  s = (s ^ 61u) ^ (s >> 16);
  s *= 9u;
  s ^= s >> 4;
  s *= 0x27d4eb2du;
  s ^= s >> 15;
  return s;
}
NV_DEV float rnd(uint32_t& st) {
  st ^= st << 13;
  st ^= st >> 17;
  st ^= st << 5;
  return static_cast<float>(st) * 0x1p-32f; // == st / 4294967296.0f exactly
}

constexpr float kRayEpsilon = 1e-2f; // kernels/constants.hpp:10
constexpr float kPi = 3.14159274101257f;
constexpr float kInvPi = 0.31830987334251f;
constexpr float kNonZeroEps = 1e-7f; // kernels/transforms.hpp:50
constexpr float kFltMax = 3.402823466e+38f;

// wave64 helpers
NV_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
NV_DEV uint32_t rank_in_mask(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

}
