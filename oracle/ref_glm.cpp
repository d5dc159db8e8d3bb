// TEST INFRASTRUCTURE ONLY.
//
// The arithmetic of a vendored dependency that defines every TriangleData bit: the reference builds a
// triangle's Woop matrix with glm::cross / glm::normalize / glm::inverse / glm::transpose
// (src/intersectables/intersectable_manager.cpp:129-145) on the glm it vendors (include/glm).  This file
// only CALLS those glm functions, compiled from where glm lies under /root/reference (oracle/Makefile,
// target `ref`, output oracle/_ref/libnova_ref_glm.so); the caller's few assignments that put e1, e2,
// normal - v1 and v1 into the matrix columns are made by the test in numpy (float32 subtractions are exact
// the same everywhere), so nothing of the reference is restated here.
//
// Used by tests/test_reference_pins.py to pin the Woop matrices of the CPU scene builder
// (nova_amd/csrc/host/scene_build.cpp: glm_inverse) — and through the "same tree" tests the device builder's
// k_emit_triangles — bit for bit, and to write tests/golden/ref_pins.npz.
#include <cstdint>

#include <glm/glm.hpp>

extern "C" {

// columns in, columns out: out = transpose(inverse(m)); m[c * 4 + r] = column c, row r (glm's own storage)
void ref_glm_transpose_inverse(const float* m, uint64_t count, float* out) {
  for (uint64_t i = 0; i < count; i++) {
    glm::mat4 a;
    for (int c = 0; c < 4; c++) a[c] = glm::vec4(m[16 * i + 4 * c], m[16 * i + 4 * c + 1], m[16 * i + 4 * c + 2], m[16 * i + 4 * c + 3]);
    const glm::mat4 t = glm::transpose(glm::inverse(a));
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 4; r++) out[16 * i + 4 * c + r] = t[c][r];
  }
}

// out = normalize(cross(a, b)) per pair
void ref_glm_normalize_cross(const float* a, const float* b, uint64_t count, float* out) {
  for (uint64_t i = 0; i < count; i++) {
    const glm::vec3 n = glm::normalize(glm::cross(glm::vec3(a[3 * i], a[3 * i + 1], a[3 * i + 2]),
                                                  glm::vec3(b[3 * i], b[3 * i + 1], b[3 * i + 2])));
    out[3 * i] = n.x; out[3 * i + 1] = n.y; out[3 * i + 2] = n.z;
  }
}

}
