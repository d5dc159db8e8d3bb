// TEST INFRASTRUCTURE ONLY.
//
// C entry points onto the reference's texture loaders, compiled from where their files lie under /root/reference
// (oracle/Makefile, target `ref`: src/material/material_loader.cpp, src/material/sky_loader.cpp,
// src/util/image/imageutils.cpp, src/util/exception/exception.cpp and the vendored stb implementation unit
// include/stb_image.cpp, as they are; output oracle/_ref/libnova_ref_material.so, build container only):
//   * nova::MaterialLoader — load_material(path, srgb) per texture in the order Model::process_mesh asks for them
//     (src/model/model.cpp:76-89, :181-209), then build(): stb decode, vertical flip, sRGB -> linear ON uint8
//     (material_loader.cpp:24-33), stbir_resize_uint8 to the mean width x height (:46-66), uint8 / 255.0f (:69-76);
//   * nova::SkyLoader — load_sky(path, srgb) + build() (sky_loader.cpp:12-35: stbi_loadf, flip, optional pow 2.2).
// Their outputs pin SURVEY.md §8f rank 4 (the texture path): tests/golden/make_ref_pins.py stores what these return
// for the aircraft's own JPEGs, tests/test_reference_pins.py and tests/test_gpu_configs.py check against it.
#include <cstdint>
#include <cstring>
#include <string>

#include "material/material_loader.hpp"
#include "material/sky_loader.hpp"

extern "C" {

// dims = { width, height, layers }.  out == nullptr: only the dimensions.  Returns 0, or 1 when a file does not load
// (the reference throws ImageException) / 2 when `out_floats` is too small.
int ref_material_build(const char* const* paths, const int* srgb, int n, uint32_t* dims, float* out, uint64_t out_floats) {
  try {
    nova::MaterialLoader loader;
    for (int i = 0; i < n; i++) {
      if (loader.load_material(paths[i], srgb[i] != 0) != i) return 1;
    }
    const nova::MaterialData d = loader.build();
    dims[0] = d.width;
    dims[1] = d.height;
    dims[2] = static_cast<uint32_t>(d.num_materials);
    if (!out) return 0;
    if (out_floats < d.data.size() * 4) return 2;
    for (size_t i = 0; i < d.data.size(); i++) {
      out[4 * i] = x(d.data[i]); out[4 * i + 1] = y(d.data[i]); out[4 * i + 2] = z(d.data[i]); out[4 * i + 3] = w(d.data[i]);
    }
    return 0;
  } catch (...) {
    return 1;
  }
}

// dims = { width, height }
int ref_sky_load(const char* path, int srgb, uint32_t* dims, float* out, uint64_t out_floats) {
  try {
    nova::SkyLoader loader;
    loader.load_sky(path, srgb != 0);
    const image_utils::image<float4> im = loader.build();
    dims[0] = im.width;
    dims[1] = im.height;
    if (!out) return 0;
    if (out_floats < im.data.size() * 4) return 2;
    for (size_t i = 0; i < im.data.size(); i++) {
      out[4 * i] = x(im.data[i]); out[4 * i + 1] = y(im.data[i]); out[4 * i + 2] = z(im.data[i]); out[4 * i + 3] = w(im.data[i]);
    }
    return 0;
  } catch (...) {
    return 1;
  }
}

}
