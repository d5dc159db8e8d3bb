// TEST INFRASTRUCTURE ONLY.
//
// The one piece of the reference that builds in this image from its own files alone: the header-only
// nova::AABB (src/intersectables/aabb.hpp) on the glm the reference vendors (include/glm).  Both are
// compiled from where they lie under /root/reference (oracle/Makefile, target `ref`, output
// oracle/_ref/libnova_ref_aabb.so); nothing of the reference is copied and nothing is stubbed.
// Everything else on the path needs the OpenCL C or CUDA device toolchains (kernels/*), CL/CUDA host
// headers (src/kernel_types/*, src/backend/*) or assimp's generated config (src/bvh, src/intersectables
// /intersectable_manager.cpp) and is therefore unbuildable here.
//
// Used by tests/test_reference_kats.py to pin the host-side box helpers of the scene builder that
// feeds the hot path (surface area / SAH cost / centre / grow / shrink) bit for bit.
#include <cstdint>

#include "intersectables/aabb.hpp"

extern "C" {

// out = {surface area, cost(n), centre.xyz}
void ref_aabb(const float* top, const float* bottom, uint64_t n, float* out) {
  nova::AABB b { glm::vec3(top[0], top[1], top[2]), glm::vec3(bottom[0], bottom[1], bottom[2]) };
  out[0] = b.get_surface_area();
  out[1] = b.get_cost(n);
  glm::vec3 c = b.get_center();
  out[2] = c.x; out[3] = c.y; out[4] = c.z;
}

// mode 0: grow, 1: shrink.  out = {top.xyz, bottom.xyz}
void ref_aabb_combine(int mode, const float* top_a, const float* bottom_a, const float* top_b,
                      const float* bottom_b, float* out) {
  nova::AABB a { glm::vec3(top_a[0], top_a[1], top_a[2]), glm::vec3(bottom_a[0], bottom_a[1], bottom_a[2]) };
  nova::AABB b { glm::vec3(top_b[0], top_b[1], top_b[2]), glm::vec3(bottom_b[0], bottom_b[1], bottom_b[2]) };
  if (mode == 0) a.grow(b); else a.shrink(b);
  out[0] = a.top.x; out[1] = a.top.y; out[2] = a.top.z;
  out[3] = a.bottom.x; out[4] = a.bottom.y; out[5] = a.bottom.z;
}

// out = {top.xyz, bottom.xyz} of AABB::make_no_intersection()
void ref_aabb_empty(float* out) {
  nova::AABB a = nova::AABB::make_no_intersection();
  out[0] = a.top.x; out[1] = a.top.y; out[2] = a.top.z;
  out[3] = a.bottom.x; out[4] = a.bottom.y; out[5] = a.bottom.z;
}

}
