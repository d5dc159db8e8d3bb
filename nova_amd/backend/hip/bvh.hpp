// BVH::build (src/bvh/bvh.cpp:17-29) and the TriangleData loop of IntersectableManager::build
// (src/intersectables/intersectable_manager.cpp:129-145) on the GPU, for a Nova built with BACKEND=hip:
// the same FlatBVHNode[] / TriangleData[] the host code makes, through one C-ABI call.
// INTEGRATION.md §7 shows the call site.
#ifndef HIP_BVH_HPP
#define HIP_BVH_HPP

#include <cstdint>
#include <vector>

#include "backend/hip/entry.hpp"

#if __has_include("util/exception/exception.hpp")
  #include "util/exception/exception.hpp"
#else
  #include "backend/hip/standalone.hpp"
#endif

namespace nova {

struct DeviceBuiltBVH {
  std::vector<FlatBVHNode> bvh;
  std::vector<TriangleData> triangle_data; // in leaf order
  std::vector<uint32_t> order;             // output triangle i = input triangle order[i]
};

// vertices: v1 v2 v3 of every triangle (9 floats each), in the order of the caller's meta data
inline DeviceBuiltBVH build_bvh(const std::vector<float>& vertices) {
  if (vertices.empty() || vertices.size() % 9 != 0) {
    throw AcceleratorException("Cannot build a BVH from an empty or ragged vertex list");
  }
  const uint64_t n = vertices.size() / 9;
  DeviceBuiltBVH out;
  out.bvh.resize(nova_hip_bvh_max_nodes(n));
  out.triangle_data.resize(n);
  out.order.resize(n);
  uint64_t num_nodes = 0;
  HIP_CHECK_AND_THROW(nova_hip_build_bvh(vertices.data(), n, reinterpret_cast<nova_bvh_node*>(out.bvh.data()),
                                         out.bvh.size(), &num_nodes,
                                         reinterpret_cast<nova_triangle_data*>(out.triangle_data.data()),
                                         out.order.data(), nullptr));
  out.bvh.resize(num_nodes);
  return out;
}

}

#endif // HIP_BVH_HPP
