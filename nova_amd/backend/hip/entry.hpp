// Kernel entry points of the HIP backend: the functions RESOLVE_KERNEL(kernel_raytrace) etc.
// name in Raytracer::raytrace (src/core/raytracer.cpp:184,194,247).  Same parameter lists as
// kernels/entrypoint/cuda/entrypoint.hpp:12-49, with the backend's image handles where CUDA has
// texture/surface objects.  Each forwards to one C-ABI launcher (include/nova_hip.h) and turns a
// non-zero return code into nova::Error, as CUDA_CHECK_AND_THROW does after a launch.
#ifndef HIP_ENTRY_HPP
#define HIP_ENTRY_HPP

#include <cstdint>
#include <vector>

#include "backend/hip/types/types.hpp"

#if __has_include("kernel_types/scene_params.hpp")
  #include "kernel_types/area_light.hpp"
  #include "kernel_types/bvh_node.hpp"
  #include "kernel_types/scene_params.hpp"
  #include "kernel_types/triangle.hpp"
#else
// stand-alone use (this repository's self-test): the C-ABI structs ARE the kernel types
namespace nova {
using SceneParams = nova_scene_params;
using TriangleData = nova_triangle_data;
using TriangleMetaData = nova_triangle_meta;
using FlatBVHNode = nova_bvh_node;
using AreaLightData = nova_area_light;
}
#endif

namespace nova {

static_assert(sizeof(SceneParams) == sizeof(nova_scene_params), "SceneParams layout");
static_assert(sizeof(TriangleData) == sizeof(nova_triangle_data), "TriangleData layout");
static_assert(sizeof(TriangleMetaData) == sizeof(nova_triangle_meta), "TriangleMetaData layout");
static_assert(sizeof(FlatBVHNode) == sizeof(nova_bvh_node), "FlatBVHNode layout");
static_assert(sizeof(AreaLightData) == sizeof(nova_area_light), "AreaLightData layout");

inline void kernel_raytrace(dim3, dim3, const SceneParams& scene_params, uint time,
                            ImageHandle temp_color1, uint2 pixel_dims, TriangleData* triangles,
                            TriangleMetaData* tri_meta, FlatBVHNode* bvh, AreaLightData* lights,
                            uint num_lights, ImageHandle materials, ImageHandle sky,
                            uint denoise_available, ImageHandle albedo_feature1,
                            ImageHandle normal_feature1) {
  HIP_CHECK_AND_THROW(nova_hip_launch_raytrace(
    reinterpret_cast<const nova_scene_params*>(&scene_params), time, temp_color1, pixel_dims.x,
    pixel_dims.y, reinterpret_cast<const nova_triangle_data*>(triangles),
    reinterpret_cast<const nova_triangle_meta*>(tri_meta),
    reinterpret_cast<const nova_bvh_node*>(bvh), reinterpret_cast<const nova_area_light*>(lights),
    num_lights, materials, sky, denoise_available, albedo_feature1, normal_feature1));
}

inline void kernel_accumulate(dim3, dim3, int sample_index, uint denoise_available,
                              ImageHandle temp_color1, ImageHandle albedo_feature1,
                              ImageHandle normal_feature1, ImageHandle prev_color,
                              ImageHandle prev_albedo_feature, ImageHandle prev_normal_feature,
                              ImageHandle temp_color2, ImageHandle albedo_feature2,
                              ImageHandle normal_feature2, uint2 pixel_dims) {
  HIP_CHECK_AND_THROW(nova_hip_launch_accumulate(
    sample_index, denoise_available, temp_color1, albedo_feature1, normal_feature1, prev_color,
    prev_albedo_feature, prev_normal_feature, temp_color2, albedo_feature2, normal_feature2,
    pixel_dims.x, pixel_dims.y));
}

inline void kernel_post_process(dim3, dim3, const SceneParams& scene_params, ImageHandle temp_color2,
                                ImageHandle pixels, uint2 pixel_dims) {
  HIP_CHECK_AND_THROW(nova_hip_launch_post_process(
    reinterpret_cast<const nova_scene_params*>(&scene_params), temp_color2, pixels, pixel_dims.x,
    pixel_dims.y));
}

// Extension (no kernel of this name in the reference): `times.size()` iterations of the body of
// Raytracer::raytrace (src/core/raytracer.cpp:169-262, without the OIDN hand-off and the per-sample
// read-back) as one device schedule with one host wait — include/nova_hip.h, nova_hip_render_samples.
// A renderer adopts it by replacing its per-sample loop; the images end up byte-identical.
inline void render_samples(const SceneParams& scene_params, const std::vector<uint32_t>& times,
                           int first_sample_index, const nova_hip_frame_images& images, uint2 pixel_dims,
                           TriangleData* triangles, TriangleMetaData* tri_meta, FlatBVHNode* bvh,
                           AreaLightData* lights, uint num_lights, ImageHandle materials, ImageHandle sky,
                           uint denoise_available) {
  HIP_CHECK_AND_THROW(nova_hip_render_samples(
    reinterpret_cast<const nova_scene_params*>(&scene_params), times.data(),
    static_cast<uint32_t>(times.size()), first_sample_index, &images, pixel_dims.x, pixel_dims.y,
    reinterpret_cast<const nova_triangle_data*>(triangles),
    reinterpret_cast<const nova_triangle_meta*>(tri_meta), reinterpret_cast<const nova_bvh_node*>(bvh),
    reinterpret_cast<const nova_area_light*>(lights), num_lights, materials, sky, denoise_available));
}

// Extension: the device half of the denoise hand-off (src/core/raytracer.cpp:211-242) — include/nova_hip.h,
// nova_hip_denoise_read / nova_hip_denoise_write.  The planes are pinned host memory the library owns.
inline nova_hip_denoise_planes denoise_read(ImageHandle color, ImageHandle albedo, ImageHandle normal, uint2 pixel_dims) {
  nova_hip_denoise_planes planes {};
  HIP_CHECK_AND_THROW(nova_hip_denoise_read(color, albedo, normal, pixel_dims.x, pixel_dims.y, &planes));
  return planes;
}
inline void denoise_write(ImageHandle color, uint2 pixel_dims) {
  HIP_CHECK_AND_THROW(nova_hip_denoise_write(color, pixel_dims.x, pixel_dims.y));
}

}

#endif // HIP_ENTRY_HPP
