// Vector types of the HIP backend, for a -DBACKEND=HIP build of Nova.
// Replaces src/backend/cuda/types/vector_types.hpp:8-42 (which gets float3 & co from
// <cuda_runtime.h>).  The host side of this backend does not include any HIP header: it talks to
// the device only through the C ABI (include/nova_hip.h), so the global vector types the rest of
// Nova expects (src/kernel_types/*.hpp, src/vector/vector_conversions.hpp) are declared here as
// plain structs with the CUDA/HIP layout: 3-vectors 12 bytes, float2/uint2/int2 8-byte aligned,
// 4-vectors 16-byte aligned.
#ifndef HIP_VECTOR_TYPES_HPP
#define HIP_VECTOR_TYPES_HPP

#include <cstdint>

using uint = unsigned int;
using uchar = unsigned char;

#define NOVA_HIP_VEC2(T, N, A) struct alignas(A) N { T x, y; }
#define NOVA_HIP_VEC3(T, N)    struct N { T x, y, z; }
#define NOVA_HIP_VEC4(T, N, A) struct alignas(A) N { T x, y, z, w; }
NOVA_HIP_VEC2(float, float2, 8);
NOVA_HIP_VEC3(float, float3);
NOVA_HIP_VEC4(float, float4, 16);
NOVA_HIP_VEC2(int, int2, 8);
NOVA_HIP_VEC3(int, int3);
NOVA_HIP_VEC4(int, int4, 16);
NOVA_HIP_VEC2(uint, uint2, 8);
NOVA_HIP_VEC3(uint, uint3);
NOVA_HIP_VEC4(uint, uint4, 16);
NOVA_HIP_VEC2(uchar, uchar2, 2);
NOVA_HIP_VEC3(uchar, uchar3);
NOVA_HIP_VEC4(uchar, uchar4, 4);
#undef NOVA_HIP_VEC2
#undef NOVA_HIP_VEC3
#undef NOVA_HIP_VEC4

// launch geometry as Accelerator::call_kernel builds it (the HIP backend ignores it: wave64
// geometry is chosen inside the library, include/nova_hip.h)
struct dim3 {
  uint32_t x = 1, y = 1, z = 1;
};

namespace nova {

// component accessors used throughout Nova's host code (cuda/types/vector_types.hpp:10-40)
#define NOVA_HIP_ACCESSOR(c)                                   \
  template <typename V> inline auto& c(V& v) { return v.c; }   \
  template <typename V> inline const auto& c(const V& v) { return v.c; }
NOVA_HIP_ACCESSOR(x)
NOVA_HIP_ACCESSOR(y)
NOVA_HIP_ACCESSOR(z)
NOVA_HIP_ACCESSOR(w)
#undef NOVA_HIP_ACCESSOR

}

#endif // HIP_VECTOR_TYPES_HPP
