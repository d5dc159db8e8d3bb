// TEST INFRASTRUCTURE ONLY — never linked into, imported by or shipped with the product path.
//
// CPU device backend for Nova's kernel headers (third sibling of the reference's own
// kernels/backend/opencl and kernels/backend/cuda shims, /root/reference/kernels/backend/).
//
// Nova's device logic (kernels/{raytrace,intersection,brdf,anti_aliasing,light,material,
// transforms,matrix,random,types}.hpp and kernels/kernels.cu) is written against a small device
// abstraction: qualifier macros, OpenCL-style vector types and math builtins, make_vector/swizzle
// helpers, read_image/write_image.  The dispatch headers kernels/backend/{kernel,vector,image,
// assertion,atomic}.hpp select an implementation by KERNEL_BACKEND_OPENCL / KERNEL_BACKEND_CUDA
// and include nothing when neither is defined (kernels/backend/kernel.hpp:4-8), and
// src/backend/types.hpp likewise (src/backend/types.hpp:4-10).  This header is force-included
// (-include) with NEITHER macro defined, so every reference header is compiled unmodified from
// where it lies, and this file supplies what a backend must supply.
//
// It also fixes the floating-point contract of the oracle (the reference builds with
// -cl-fast-relaxed-math -cl-mad-enable / --use_fast_math, so its own results are
// implementation-defined; src/backend/opencl/accelerator/accelerator.cpp:34-37,
// CMakeLists.txt:87-95):
//   * IEEE-754 binary32, round-to-nearest-even, subnormals kept, correctly rounded / and sqrt;
//   * a*b+c written in ONE source expression is fused (clang -ffp-contract=on), mirroring
//     -cl-mad-enable; nothing else is re-associated;
//   * the builtins below have the stated operation order (dot is an fma chain, normalize is a
//     multiply by 1/sqrt, min/max are minNum/maxNum as v_min_f32/v_max_f32 and fminf/fmaxf are);
//   * sin, cos, atan, atan2, asin, acos, exp, pow are the bit-reproducible implementations of
//     include/nova_math.h (Cephes polynomials, every operation written out), because a one-ulp
//     difference between a host libm and a device library is enough to send a path elsewhere.
#ifndef NOVA_ORACLE_REFSHIM_CPU_BACKEND_HPP
#define NOVA_ORACLE_REFSHIM_CPU_BACKEND_HPP

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <type_traits>

#include "nova_math.h"

// ---- qualifiers (kernels/backend/opencl/kernel.hpp:4-9 defines the same names) -----------------
#define DEVICE
#define HOST
#define KERNEL
#define GLOBAL
#define LOCAL
#define CONSTANT

#ifdef assert
  #undef assert
#endif
#define assert(expr)

// ---- scalar and vector types (OpenCL layout: 3-vectors occupy 16 bytes) ------------------------
typedef unsigned int uint;
typedef unsigned char uchar;

#define REFSHIM_VEC(T, N) typedef T T##N __attribute__((ext_vector_type(N)))
REFSHIM_VEC(float, 2);
REFSHIM_VEC(float, 3);
REFSHIM_VEC(float, 4);
REFSHIM_VEC(int, 2);
REFSHIM_VEC(int, 3);
REFSHIM_VEC(int, 4);
REFSHIM_VEC(uint, 2);
REFSHIM_VEC(uint, 3);
REFSHIM_VEC(uint, 4);
REFSHIM_VEC(uchar, 3);
REFSHIM_VEC(uchar, 4);
#undef REFSHIM_VEC

constexpr float M_PI_F = 3.14159274101257f;
constexpr float M_1_PI_F = 0.31830987334251f;

// ---- work-item id ------------------------------------------------------------------------------
namespace refshim {
extern thread_local int global_id[2];
}
inline size_t get_global_id(uint dim) { return static_cast<size_t>(refshim::global_id[dim]); }

// ---- math builtins -----------------------------------------------------------------------------
inline float min(float a, float b) { return __builtin_fminf(a, b); }
inline float max(float a, float b) { return __builtin_fmaxf(a, b); }
inline uint min(uint a, uint b) { return b < a ? b : a; }
inline uint max(uint a, uint b) { return a < b ? b : a; }
inline float3 min(float3 a, float3 b) { return { min(a.x, b.x), min(a.y, b.y), min(a.z, b.z) }; }
inline float3 max(float3 a, float3 b) { return { max(a.x, b.x), max(a.y, b.y), max(a.z, b.z) }; }
inline float fabs(float a) { return __builtin_fabsf(a); }
inline float sqrt(float a) { return __builtin_sqrtf(a); }
inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
inline float3 clamp(float3 x, float lo, float hi) {
  return { clamp(x.x, lo, hi), clamp(x.y, lo, hi), clamp(x.z, lo, hi) };
}

// dot: fma chain, last component first (so that a 4-vector dot with w == 1 is three fmas).
inline float dot(float3 a, float3 b) {
  return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z));
}
inline float dot(float4 a, float4 b) {
  return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, __builtin_fmaf(a.z, b.z, a.w * b.w)));
}
inline float length(float3 a) { return sqrt(dot(a, a)); }
inline float distance(float3 a, float3 b) { return length(a - b); }
inline float3 normalize(float3 a) {
  float inv = 1.0f / sqrt(dot(a, a));
  return a * inv;
}
inline float3 cross(float3 a, float3 b) {
  return { __builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
           __builtin_fmaf(a.x, b.y, -(a.y * b.x)) };
}
inline float3 mix(float3 a, float3 b, float t) {
  float3 d = b - a;
  return { __builtin_fmaf(d.x, t, a.x), __builtin_fmaf(d.y, t, a.y), __builtin_fmaf(d.z, t, a.z) };
}

// elementary functions: the backend's published numeric contract (include/nova_math.h)
inline float pow(float a, float b) { return nova_powf(a, b); }
inline float3 pow(float3 a, float b) { return { nova_powf(a.x, b), nova_powf(a.y, b), nova_powf(a.z, b) }; }
inline float exp(float a) { return nova_expf(a); }
inline float3 exp(float3 a) { return { nova_expf(a.x), nova_expf(a.y), nova_expf(a.z) }; }
inline float sin(float a) { return nova_sinf(a); }
inline float cos(float a) { return nova_cosf(a); }
inline float atan(float a) { return nova_atanf(a); }
inline float atan2(float a, float b) { return nova_atan2f(a, b); }
inline float asin(float a) { return nova_asinf(a); }
inline float acos(float a) { return nova_acosf(a); }

inline int isfinite(float a) { return __builtin_isfinite(a); }
inline int3 isfinite(float3 a) {
  return { -isfinite(a.x), -isfinite(a.y), -isfinite(a.z) };
}
inline int3 isless(float3 a, float3 b) { return a < b; }
inline int any(int3 a) { return (a.x | a.y | a.z) < 0; }
inline int all(int3 a) { return (a.x & a.y & a.z) < 0; }

// ---- images: linear float4 / uchar4 arrays, OpenCL 1.2 §8.2 sampling in software ---------------
#include "kernels/backend/common/image.hpp" // enum class AddressMode { WRAP, CLAMP }

namespace refshim {

struct Image2D {
  float4* texels; // row-major, width * height (* layers)
  int width;
  int height;
  int layers;
};
struct Image2DU8 {
  uchar4* texels;
  int width;
  int height;
};
struct Image2DArray : Image2D {};

// CLK_ADDRESS_CLAMP: out-of-range texels read the border colour (0,0,0,0) for a CL_RGBA image.
inline float4 fetch_clamp(const Image2D& im, int i, int j, int layer) {
  if (i < 0 || j < 0 || i >= im.width || j >= im.height) {
    return { 0.0f, 0.0f, 0.0f, 0.0f };
  }
  return im.texels[(static_cast<size_t>(layer) * im.height + j) * im.width + i];
}

// CLK_FILTER_LINEAR with CLK_NORMALIZED_COORDS_TRUE (OpenCL 1.2 spec §8.2, eq. for 2D images).
template <AddressMode A>
inline float4 sample_linear(const Image2D& im, float s, float t, int layer) {
  const float w = static_cast<float>(im.width);
  const float h = static_cast<float>(im.height);
  float u, v;
  int i0, j0, i1, j1;
  if constexpr (A == AddressMode::WRAP) {
    // CLK_ADDRESS_REPEAT
    u = (s - ::floorf(s)) * w;
    v = (t - ::floorf(t)) * h;
    i0 = static_cast<int>(::floorf(u - 0.5f));
    j0 = static_cast<int>(::floorf(v - 0.5f));
    i1 = i0 + 1;
    j1 = j0 + 1;
    if (i0 < 0) i0 += im.width;
    if (j0 < 0) j0 += im.height;
    if (i1 > im.width - 1) i1 -= im.width;
    if (j1 > im.height - 1) j1 -= im.height;
  } else {
    u = s * w;
    v = t * h;
    i0 = static_cast<int>(::floorf(u - 0.5f));
    j0 = static_cast<int>(::floorf(v - 0.5f));
    i1 = i0 + 1;
    j1 = j0 + 1;
  }
  const float a = (u - 0.5f) - ::floorf(u - 0.5f);
  const float b = (v - 0.5f) - ::floorf(v - 0.5f);
  const float4 t00 = fetch_clamp(im, i0, j0, layer);
  const float4 t10 = fetch_clamp(im, i1, j0, layer);
  const float4 t01 = fetch_clamp(im, i0, j1, layer);
  const float4 t11 = fetch_clamp(im, i1, j1, layer);
  const float w00 = (1.0f - a) * (1.0f - b);
  const float w10 = a * (1.0f - b);
  const float w01 = (1.0f - a) * b;
  const float w11 = a * b;
  return ((t00 * w00 + t10 * w10) + t01 * w01) + t11 * w11;
}

}

typedef const refshim::Image2D* image2d_read_t;
typedef const refshim::Image2DArray* image2d_array_read_t;
struct image2d_write_t {
  refshim::Image2D* f32;
  refshim::Image2DU8* u8;
};

namespace nova {

// make_vector / swizzles / reflect: what kernels/backend/opencl/vector.hpp:9-25 and swizzle.hpp
// give the OpenCL build.  (The 2-, 3- and 4-scalar forms come from the reference's own
// kernels/backend/common/vector.hpp, compiled as is.)
template <typename W, typename U, typename T,
          typename = std::enable_if_t<!std::is_arithmetic_v<U> && std::is_arithmetic_v<T>>>
constexpr W make_vector(U u, T t) {
  if constexpr (sizeof(U) / sizeof(u.x) == 2) {
    return W { u.x, u.y, t };
  } else {
    return W { u.x, u.y, u.z, t };
  }
}

template <typename W, typename U>
constexpr W make_vector(U u) {
  if constexpr (std::is_arithmetic_v<U>) {
    return static_cast<W>(u);
  } else {
    return __builtin_convertvector(u, W);
  }
}

template <typename W, typename U>
constexpr W xyz(U u) {
  return W { u.x, u.y, u.z };
}
template <typename W, typename U>
constexpr W xy(U u) {
  return W { u.x, u.y };
}

// kernels/backend/opencl/vector.hpp:23 — i - 2 (n.i) n, the last step fused as clang fuses it.
inline float3 reflect(const float3& i, const float3& n) {
  float k = -(2.0f * dot(n, i));
  return { __builtin_fmaf(k, n.x, i.x), __builtin_fmaf(k, n.y, i.y), __builtin_fmaf(k, n.z, i.z) };
}

// read_image / write_image: kernels/backend/opencl/image.hpp:23-105. Float coordinates use the
// LINEAR, normalised samplers (image.hpp:18-21); only the float4 forms are used by the kernels.
template <typename W, AddressMode A = AddressMode::CLAMP>
inline W read_image(image2d_read_t image, const float2& coords) {
  return refshim::sample_linear<A>(*image, coords.x, coords.y, 0);
}
template <typename W, AddressMode A = AddressMode::CLAMP>
inline W read_image(image2d_read_t image, const float2& coords, const float2& offset) {
  float2 c = coords + offset;
  return refshim::sample_linear<A>(*image, c.x, c.y, 0);
}
template <typename W, AddressMode A = AddressMode::CLAMP>
inline W read_image(image2d_array_read_t image, const float2& coords, int index) {
  return refshim::sample_linear<A>(*image, coords.x, coords.y, index);
}
inline void write_image(image2d_write_t image, const int2& coords, const float4& value) {
  if (coords.x < 0 || coords.y < 0 || coords.x >= image.f32->width || coords.y >= image.f32->height)
    return;
  image.f32->texels[static_cast<size_t>(coords.y) * image.f32->width + coords.x] = value;
}
inline void write_image(image2d_write_t image, const int2& coords, const uchar4& value) {
  if (coords.x < 0 || coords.y < 0 || coords.x >= image.u8->width || coords.y >= image.u8->height)
    return;
  image.u8->texels[static_cast<size_t>(coords.y) * image.u8->width + coords.x] = value;
}

}

#endif
