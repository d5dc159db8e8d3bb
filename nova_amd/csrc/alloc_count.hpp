// Counts the allocating HIP runtime calls this library makes (hipMalloc / hipFree / pinned host memory / events /
// streams), so that a test can check that nova_hip_trace_rays_noalloc really makes none
// (NOVA_HIP_OPT_ALLOC_CALLS, read-only).  Included right after <hip/hip_runtime.h> and before every other
// header of the library: the function-like macros below then wrap each call site that follows.
#pragma once

#include <atomic>
#include <cstdint>

namespace nova_hip {
inline std::atomic<uint64_t> g_alloc_calls { 0 };
inline void count_alloc_call() { g_alloc_calls.fetch_add(1, std::memory_order_relaxed); }
}

#define hipMalloc(...) (::nova_hip::count_alloc_call(), (hipMalloc)(__VA_ARGS__))
#define hipFree(...) (::nova_hip::count_alloc_call(), (hipFree)(__VA_ARGS__))
#define hipHostMalloc(...) (::nova_hip::count_alloc_call(), (hipHostMalloc)(__VA_ARGS__))
#define hipHostFree(...) (::nova_hip::count_alloc_call(), (hipHostFree)(__VA_ARGS__))
#define hipEventCreate(...) (::nova_hip::count_alloc_call(), (hipEventCreate)(__VA_ARGS__))
#define hipEventCreateWithFlags(...) (::nova_hip::count_alloc_call(), (hipEventCreateWithFlags)(__VA_ARGS__))
#define hipEventDestroy(...) (::nova_hip::count_alloc_call(), (hipEventDestroy)(__VA_ARGS__))
#define hipStreamCreateWithFlags(...) (::nova_hip::count_alloc_call(), (hipStreamCreateWithFlags)(__VA_ARGS__))
#define hipStreamDestroy(...) (::nova_hip::count_alloc_call(), (hipStreamDestroy)(__VA_ARGS__))
