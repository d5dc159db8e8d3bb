// The per-rank multi-GPU hook, written against Nova's REAL headers and compiled by integration/CMakeLists.txt next to
// boundary_check.cpp (tests/test_integration.py).  It is what a multi-process launch of Nova adds — one method on
// Raytracer and three lines in Scene::render_to_image — and everything below the two calls is
// inside libnova_hip.so: pack -> ncclAllGather (RCCL over xGMI) -> unpack on the library's stream.
//
//   // src/core/raytracer.hpp, in class Raytracer, #ifdef BACKEND_HIP
//   image_utils::image<uchar4> gather_frame();
//
//   // src/scene/scene.cpp:320-331, between the sample loop and image_utils::write_image, #ifdef BACKEND_HIP
//   img = raytracer.gather_frame();
//   if (!nova::multi_gpu::writes_the_frame()) return;     // ranks other than 0 have nothing to write
//
// Launch: one process per GPU,
//   NOVA_HIP_DEVICE=r NOVA_HIP_TILE_RANK=r NOVA_HIP_TILE_RANKS=N NOVA_HIP_COMM_FILE=/tmp/nova.id ./nova --headless
#include <cstdint>

#include "backend/accelerator.hpp"
#include "backend/types.hpp"
#include "kernel_types/scene_params.hpp"
#include "util/image/imageutils.hpp"

#include "backend/hip/multi_gpu.hpp"

namespace nova {

// Raytracer::gather_frame(), with the members it uses as parameters (src/core/raytracer.hpp:32-65: accelerator,
// color_img, temp_color_img2, width, height; the SceneParams are the ones set_scene made, raytracer.cpp:63-65).
image_utils::image<uchar4> gather_frame(Accelerator& accelerator, Image2DWrite<uchar4>& color_img,
                                        Image2DReadWrite<float4>& temp_color_img2, const SceneParams& scene_params,
                                        uint32_t width, uint32_t height) {
  // this rank's tiles of color_img / temp_color_img2 are final (the last Raytracer::raytrace returned); afterwards
  // color_img holds the whole post-processed frame on the rank that writes it
  multi_gpu::finish_frame(color_img.data(), temp_color_img2.write_access().data(), scene_params, uint2 { width, height });
  return { accelerator.read_image2D(color_img, width, height), width, height }; // as raytracer.cpp:253-261
}

}
