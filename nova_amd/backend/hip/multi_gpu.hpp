// Multi-GPU hook of the HIP backend (no counterpart in the reference: one device, no collective —
// src/backend/opencl/accelerator/accelerator.cpp:11-25).  One process per GPU, started by the launcher as
//   NOVA_HIP_DEVICE=r NOVA_HIP_TILE_RANK=r NOVA_HIP_TILE_RANKS=N NOVA_HIP_COMM_FILE=/path/shared/by/the/ranks ./nova --headless
// The Accelerator constructor reads the first three (the renderer is untouched: every rank renders its own 64x64
// tiles, no communication per sample).  What is left is ONE call per final frame, placed in Scene::render_to_image
// between the sample loop and image_utils::write_image (src/scene/scene.cpp:320-331):
//
//   #ifdef BACKEND_HIP
//     nova::multi_gpu::finish_frame(raytracer_color_image, raytracer_mean_image, scene_params, { width, height });
//     if (!nova::multi_gpu::writes_the_frame()) return;      // ranks other than 0 have nothing to write
//   #endif
//
// It gathers the finished tiles over xGMI with RCCL, inside libnova_hip.so (nova_hip_tiles_allgather: pack ->
// ncclAllGather -> unpack on the library's stream).  The ranks agree on the communicator through a small file:
// rank 0 draws the ncclUniqueId (nova_hip_tiles_unique_id) and writes it, the others wait for it to appear — no MPI,
// no sockets of our own; a launcher that has a better channel calls nova_hip_tiles_comm_init itself.
#ifndef HIP_MULTI_GPU_HPP
#define HIP_MULTI_GPU_HPP

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "backend/hip/entry.hpp"

namespace nova {
namespace multi_gpu {

inline uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* e = std::getenv(name);
  return e && *e ? static_cast<uint32_t>(std::strtoul(e, nullptr, 10)) : dflt;
}
inline uint32_t rank() { return env_u32("NOVA_HIP_TILE_RANK", 0); }
inline uint32_t ranks() { return env_u32("NOVA_HIP_TILE_RANKS", 1); }
inline bool writes_the_frame() { return rank() == 0; }

// Collective over the ranks.  Throws nova::Error (RCCL and HIP codes included) like every backend call.
inline void init_communicator(uint32_t rank, uint32_t ranks, const char* path) {
  if (!path || !*path) throw Error(NOVA_HIP_E_INVALID_ARGUMENT);
  unsigned char id[NOVA_HIP_UNIQUE_ID_BYTES];
  if (rank == 0) {
    HIP_CHECK_AND_THROW(nova_hip_tiles_unique_id(id));
    const std::string tmp = std::string(path) + ".tmp";
    std::FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id) || std::fclose(f) != 0) throw Error(NOVA_HIP_E_INVALID_ARGUMENT);
    if (std::rename(tmp.c_str(), path) != 0) throw Error(NOVA_HIP_E_INVALID_ARGUMENT); // appears whole or not at all
  } else {
    const uint32_t timeout_s = env_u32("NOVA_HIP_COMM_TIMEOUT", 120);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      std::FILE* f = std::fopen(path, "rb");
      if (f) {
        const size_t got = std::fread(id, 1, sizeof(id), f);
        std::fclose(f);
        if (got == sizeof(id)) break;
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) throw Error(NOVA_HIP_E_NO_COMM);
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  HIP_CHECK_AND_THROW(nova_hip_tiles_comm_init(id, rank, ranks));
}

// After the last sample: `pixels` (uchar4, what Raytracer::raytrace read back) and `mean` (float4, temp_color_img2)
// hold this rank's tiles.  Afterwards `pixels` holds the whole post-processed frame on the rank that writes it.
// FXAA reads across tile borders (kernels/anti_aliasing.hpp:99-116), so with anti_aliasing the float4 means travel
// and the writing rank post-processes the whole frame; without it the uchar4 pixels travel (a quarter of the bytes).
// (This overload expects a communicator: the one below makes it from the environment on first use.)
inline void finish_frame(ImageHandle pixels, ImageHandle mean, const SceneParams& scene_params, uint2 pixel_dims,
                         uint32_t rank, uint32_t ranks) {
  const nova_scene_params& p = reinterpret_cast<const nova_scene_params&>(scene_params);
  if (p.anti_aliasing) {
    HIP_CHECK_AND_THROW(nova_hip_tiles_allgather(mean));
    if (rank == 0) {
      HIP_CHECK_AND_THROW(nova_hip_set_option(NOVA_HIP_OPT_TILE_RANKS, 1)); // the image is whole now
      const int rc = nova_hip_launch_post_process(&p, mean, pixels, pixel_dims.x, pixel_dims.y);
      HIP_CHECK_AND_THROW(nova_hip_set_option(NOVA_HIP_OPT_TILE_RANKS, ranks));
      HIP_CHECK_AND_THROW(rc);
    }
  } else {
    HIP_CHECK_AND_THROW(nova_hip_tiles_allgather(pixels));
  }
}

inline void finish_frame(ImageHandle pixels, ImageHandle mean, const SceneParams& scene_params, uint2 pixel_dims) {
  if (ranks() <= 1) return;
  static bool have_comm = false;
  if (!have_comm) {
    init_communicator(rank(), ranks(), std::getenv("NOVA_HIP_COMM_FILE"));
    have_comm = true;
  }
  finish_frame(pixels, mean, scene_params, pixel_dims, rank(), ranks());
}

}
}

#endif // HIP_MULTI_GPU_HPP
