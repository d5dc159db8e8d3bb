// TEST INFRASTRUCTURE ONLY.
//
// C entry points onto the reference's flatten / pack templates (src/vector/vector_utils.hpp:8-51) in the
// instantiations the denoise hand-off uses (src/core/raytracer.cpp:219-221, :240): float4 -> 3 floats per
// pixel and back.  Compiled from where the header lies under /root/reference with the image's host OpenCL
// headers (oracle/Makefile, target `ref`, output oracle/_ref/libnova_ref_vector_utils.so).  Note what
// pack<float4, float, 3> does with the fourth component: the vector is value-initialised, so w = 0.
//
// Used by tests/test_reference_pins.py to write the golden vectors that pin nova_hip_denoise_read /
// nova_hip_denoise_write (the device half of that hand-off).
#include <cstdint>
#include <vector>

#include "vector/vector_utils.hpp"

extern "C" {

void ref_flatten_float4_to_3(const float* rgba, uint64_t pixels, float* out) {
  std::vector<float4> v(pixels);
  for (uint64_t i = 0; i < pixels; i++) {
    x(v[i]) = rgba[4 * i]; y(v[i]) = rgba[4 * i + 1]; z(v[i]) = rgba[4 * i + 2]; w(v[i]) = rgba[4 * i + 3];
  }
  const std::vector<float> f = nova::flatten<float4, float, 3>(v);
  for (uint64_t i = 0; i < f.size(); i++) out[i] = f[i];
}

void ref_pack_3_to_float4(const float* rgb, uint64_t pixels, float* out) {
  const std::vector<float> f(rgb, rgb + 3 * pixels);
  const std::vector<float4> v = nova::pack<float4, float, 3>(f);
  for (uint64_t i = 0; i < v.size(); i++) {
    out[4 * i] = x(v[i]); out[4 * i + 1] = y(v[i]); out[4 * i + 2] = z(v[i]); out[4 * i + 3] = w(v[i]);
  }
}

}
