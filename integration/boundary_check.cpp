// Compile-time check of the drop-in boundary against Nova's REAL headers (run by tests/test_integration.py
// through integration/CMakeLists.txt; never part of libnova_hip.so):
//   * "backend/accelerator.hpp" / "backend/types.hpp" are Nova's dispatch headers with the BACKEND_HIP
//     branch of INTEGRATION.md §2 (integration/patch_nova.py), so nova::Accelerator etc. below are this
//     repository's nova_amd/backend/hip/** reached exactly as src/core/raytracer.cpp reaches a backend;
//   * "kernel_types/*.hpp" are Nova's own structs, laid out by this compiler with the HIP backend's vector
//     types: sizeof AND offsetof must equal the C ABI's (include/nova_hip.h) — entry.hpp reinterpret_casts;
//   * every Accelerator template is instantiated with the member types and argument lists of
//     src/core/raytracer.hpp:32-65 and raytracer.cpp:42-45, :63-165, :173-255.
#include <cstddef>
#include <cstdint>
#include <vector>

#include "backend/accelerator.hpp"
#include "backend/types.hpp"
#include "kernel_types/area_light.hpp"
#include "kernel_types/bvh_node.hpp"
#include "kernel_types/eye_coords.hpp"
#include "kernel_types/scene_params.hpp"
#include "kernel_types/triangle.hpp"
#include "nova_hip.h"

#ifndef BACKEND_HIP
  #error "compile with -DBACKEND_HIP"
#endif

using namespace nova;

#define SAME_SIZE(A, B) static_assert(sizeof(A) == sizeof(B), #A " size")
#define SAME_OFFSET(A, fa, B, fb) static_assert(offsetof(A, fa) == offsetof(B, fb), #A "::" #fa " offset")

SAME_SIZE(FlatBVHNode, nova_bvh_node);
SAME_OFFSET(FlatBVHNode, top_offset_left, nova_bvh_node, top_offset_left);
SAME_OFFSET(FlatBVHNode, bottom_num_right, nova_bvh_node, bottom_num_right);
SAME_SIZE(TriangleData, nova_triangle_data);
SAME_SIZE(TriangleMetaData, nova_triangle_meta);
SAME_OFFSET(TriangleMetaData, normal1, nova_triangle_meta, normal1);
SAME_OFFSET(TriangleMetaData, tangent1, nova_triangle_meta, tangent1);
SAME_OFFSET(TriangleMetaData, bitangent3, nova_triangle_meta, bitangent3);
SAME_OFFSET(TriangleMetaData, texture_coord1, nova_triangle_meta, texture_coord1);
SAME_OFFSET(TriangleMetaData, texture_coord3, nova_triangle_meta, texture_coord3);
SAME_OFFSET(TriangleMetaData, kD, nova_triangle_meta, kD);
SAME_OFFSET(TriangleMetaData, metallic, nova_triangle_meta, metallic);
SAME_OFFSET(TriangleMetaData, roughness, nova_triangle_meta, roughness);
SAME_OFFSET(TriangleMetaData, diffuse_index, nova_triangle_meta, diffuse_index);
SAME_OFFSET(TriangleMetaData, normal_index, nova_triangle_meta, normal_index);
SAME_OFFSET(TriangleMetaData, light_index, nova_triangle_meta, light_index);
SAME_SIZE(AreaLightData, nova_area_light);
SAME_OFFSET(AreaLightData, intensity, nova_area_light, intensity);
SAME_OFFSET(AreaLightData, rect, nova_area_light, rect);
SAME_OFFSET(AreaLightData, tri, nova_area_light, tri);
SAME_OFFSET(AreaLightData, type, nova_area_light, type);
SAME_SIZE(EyeCoords, nova_eye_coords);
SAME_OFFSET(EyeCoords, coord_scale, nova_eye_coords, coord_scale);
SAME_OFFSET(EyeCoords, coord_dims, nova_eye_coords, coord_dims);
SAME_OFFSET(EyeCoords, eye_pos, nova_eye_coords, eye_pos);
SAME_OFFSET(EyeCoords, eye_coord_frame, nova_eye_coords, eye_coord_frame);
SAME_SIZE(SceneParams, nova_scene_params);
SAME_OFFSET(SceneParams, eye_coords, nova_scene_params, eye_coords);
SAME_OFFSET(SceneParams, shading_diffuse, nova_scene_params, shading_diffuse);
SAME_OFFSET(SceneParams, shading_metallic, nova_scene_params, shading_metallic);
SAME_OFFSET(SceneParams, shading_roughness, nova_scene_params, shading_roughness);
SAME_OFFSET(SceneParams, path_tracing, nova_scene_params, path_tracing);
SAME_OFFSET(SceneParams, num_samples, nova_scene_params, num_samples);
SAME_OFFSET(SceneParams, exposure, nova_scene_params, exposure);
SAME_OFFSET(SceneParams, anti_aliasing, nova_scene_params, anti_aliasing);

// The device objects a renderer keeps (types of src/core/raytracer.hpp:32-65): default-constructible and
// move-assignable, because set_scene assigns fresh ones.
struct Renderer {
  Accelerator accelerator;
  uint32_t width = 0, height = 0;
  Image2DWrite<uchar4> color_img;
  Image2DRead<float4> prev_color_img;
  Image2DReadWrite<float4> temp_color_img1, temp_color_img2, albedo_img1, normal_img1, albedo_img2, normal_img2;
  Image2DRead<float4> prev_albedo_img, prev_normal_img;
  Wrapper<uint2> pixel_dims_wrapper;
  Wrapper<SceneParams> scene_params_wrapper;
  Buffer<TriangleData> triangle_buf;
  Buffer<TriangleMetaData> tri_meta_buf;
  Buffer<FlatBVHNode> bvh_buf;
  Buffer<AreaLightData> light_buf;
  Wrapper<uint32_t> num_lights_wrapper;
  Image2DArray<float4> material_imgs;
  Image2DRead<float4> sky_img;
  int sample_index = 0;

  void set_scene(uint32_t w, uint32_t h, std::vector<TriangleData>& triangle_data,
                 std::vector<TriangleMetaData>& triangle_meta_data, std::vector<FlatBVHNode>& bvh_data,
                 std::vector<AreaLightData>& light_data, std::vector<float4>& material_data, size_t mat_w, size_t mat_h,
                 size_t mat_layers, std::vector<float4>& sky_data, size_t sky_w, size_t sky_h, bool denoise_available) {
    scene_params_wrapper = accelerator.create_wrapper<SceneParams>(SceneParams {});
    color_img = accelerator.create_image2D_write<uchar4>(ImageChannelOrder::RGBA, ImageChannelType::UINT8, w, h);
    prev_color_img = accelerator.create_image2D_read<float4>(ImageChannelOrder::RGBA, ImageChannelType::FLOAT,
                                                             AddressMode::CLAMP, FilterMode::LINEAR, true, w, h);
    temp_color_img1 = accelerator.create_image2D_readwrite<float4>(ImageChannelOrder::RGBA, ImageChannelType::FLOAT,
                                                                   AddressMode::CLAMP, FilterMode::LINEAR, true, w, h);
    temp_color_img2 = accelerator.create_image2D_readwrite<float4>(ImageChannelOrder::RGBA, ImageChannelType::FLOAT,
                                                                   AddressMode::CLAMP, FilterMode::LINEAR, true, w, h);
    if (denoise_available) {
      albedo_img1 = accelerator.create_image2D_readwrite<float4>(ImageChannelOrder::RGBA, ImageChannelType::FLOAT,
                                                                 AddressMode::CLAMP, FilterMode::LINEAR, true, w, h);
      prev_albedo_img = accelerator.create_image2D_read<float4>(ImageChannelOrder::RGBA, ImageChannelType::FLOAT,
                                                                AddressMode::CLAMP, FilterMode::LINEAR, true, w, h);
    }
    pixel_dims_wrapper = accelerator.create_wrapper<uint2>(uint2 { w, h });
    triangle_buf = accelerator.create_buffer(MemFlags::READ_ONLY, triangle_data);
    tri_meta_buf = accelerator.create_buffer(MemFlags::READ_ONLY, triangle_meta_data);
    bvh_buf = accelerator.create_buffer(MemFlags::READ_ONLY, bvh_data);
    if (!light_data.empty()) light_buf = accelerator.create_buffer(MemFlags::READ_ONLY, light_data);
    num_lights_wrapper = accelerator.create_wrapper<uint32_t>(light_data.size());
    material_imgs = accelerator.create_image2D_array<float4>(ImageChannelOrder::RGBA, ImageChannelType::FLOAT,
                                                             AddressMode::WRAP, FilterMode::LINEAR, true, mat_layers,
                                                             mat_w, mat_h, material_data);
    sky_img = accelerator.create_image2D_read<float4>(ImageChannelOrder::RGBA, ImageChannelType::FLOAT, AddressMode::WRAP,
                                                      FilterMode::LINEAR, true, sky_w, sky_h, sky_data);
    width = w;
    height = h;
  }

  void start() {
    sample_index = 0;
    accelerator.fill_image2D(prev_color_img, width, height, float4 {});
  }

  std::vector<uchar4> raytrace(bool denoise_available) {
    auto sample_index_wrapper = accelerator.create_wrapper<int>(sample_index);
    auto time_wrapper = accelerator.create_wrapper<uint32_t>(12345u);
    auto denoise_available_wrapper = accelerator.create_wrapper<uint32_t>(denoise_available);
    uint2 global_dims { width, height };
    uint2 local_dims { 8, 4 };
    accelerator.call_kernel(RESOLVE_KERNEL(kernel_raytrace), global_dims, local_dims, scene_params_wrapper, time_wrapper,
                            temp_color_img1.write_access(), pixel_dims_wrapper, triangle_buf, tri_meta_buf, bvh_buf,
                            light_buf, num_lights_wrapper, material_imgs, sky_img, denoise_available_wrapper,
                            albedo_img1.write_access(), normal_img1.write_access());
    accelerator.call_kernel(RESOLVE_KERNEL(kernel_accumulate), global_dims, local_dims, sample_index_wrapper,
                            denoise_available_wrapper, temp_color_img1.read_access(), albedo_img1.read_access(),
                            normal_img1.read_access(), prev_color_img, prev_albedo_img, prev_normal_img,
                            temp_color_img2.write_access(), albedo_img2.write_access(), normal_img2.write_access(),
                            pixel_dims_wrapper);
    accelerator.copy_image2D(prev_color_img, temp_color_img2, width, height);
    if (denoise_available) {
      accelerator.copy_image2D(prev_albedo_img, albedo_img2, width, height);
      std::vector<float4> color_image = accelerator.read_image2D(temp_color_img2, width, height);
      accelerator.write_image2D(temp_color_img2, width, height, color_image);
    }
    accelerator.call_kernel(RESOLVE_KERNEL(kernel_post_process), global_dims, local_dims, scene_params_wrapper,
                            temp_color_img2.read_access(), color_img, pixel_dims_wrapper);
    sample_index++;
    return accelerator.read_image2D(color_img, width, height);
  }
};

int main() {
  // instantiation is the test; nothing here needs a device
  void (Renderer::*a)(uint32_t, uint32_t, std::vector<TriangleData>&, std::vector<TriangleMetaData>&,
                      std::vector<FlatBVHNode>&, std::vector<AreaLightData>&, std::vector<float4>&, size_t, size_t, size_t,
                      std::vector<float4>&, size_t, size_t, bool) = &Renderer::set_scene;
  void (Renderer::*b)() = &Renderer::start;
  std::vector<uchar4> (Renderer::*c)(bool) = &Renderer::raytrace;
  return (a && b && c) ? 0 : 1;
}
