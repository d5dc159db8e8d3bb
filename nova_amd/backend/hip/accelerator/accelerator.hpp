// nova::Accelerator for -DBACKEND=HIP.  Same public method set as
// src/backend/cuda/accelerator/accelerator.hpp:16-232 (the only caller is nova::Raytracer,
// src/core/raytracer.cpp), header-only, every device action through the C ABI.
#ifndef HIP_ACCELERATOR_HPP
#define HIP_ACCELERATOR_HPP

#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "backend/hip/entry.hpp"
#include "backend/hip/types/types.hpp"

#if __has_include("backend/common/types/types.hpp")
  #include "backend/common/types/types.hpp"
  #include "backend/common/utils/utils.hpp"
  #include "util/exception/exception.hpp"
#else
  #include "backend/hip/standalone.hpp"
#endif

#define RESOLVE_KERNEL(kernel) kernel

namespace nova {

class Accelerator {
public:
  // The reference chooses its device here (opencl/accelerator/accelerator.cpp:11-25).  HIP-only
  // knobs come from the environment so that Nova's flow stays untouched:
  //   NOVA_HIP_DEVICE      device ordinal (default 0)
  //   NOVA_HIP_MAX_DEPTH   max path segments per sample, 0 = unbounded like the reference
  //   NOVA_HIP_FIXED_SEED  replaces the wall-clock `time` argument by seed + launch counter
  //   NOVA_HIP_TILE_RANK / NOVA_HIP_TILE_RANKS   multi-GPU: this process (one per GPU) renders the 64x64
  //                        tiles t with t % RANKS == RANK (nova_hip.h: NOVA_HIP_OPT_TILE_RANK/RANKS)
  //   NOVA_HIP_CLAMP_TO_EDGE=1  AddressMode::CLAMP as the CUDA backend's clamp-to-edge; default = the OpenCL
  //                        backend's border colour 0 (nova_hip.h: NOVA_HIP_OPT_CLAMP_TO_EDGE)
  Accelerator() {
    const char* dev = std::getenv("NOVA_HIP_DEVICE");
    HIP_CHECK_AND_THROW(nova_hip_init(dev ? std::atoi(dev) : 0));
    if (const char* d = std::getenv("NOVA_HIP_MAX_DEPTH")) {
      HIP_CHECK_AND_THROW(nova_hip_set_option(NOVA_HIP_OPT_MAX_DEPTH, std::atoll(d)));
    }
    if (const char* n = std::getenv("NOVA_HIP_TILE_RANKS")) {
      const char* r = std::getenv("NOVA_HIP_TILE_RANK");
      HIP_CHECK_AND_THROW(nova_hip_set_option(NOVA_HIP_OPT_TILE_RANKS, std::atoll(n)));
      HIP_CHECK_AND_THROW(nova_hip_set_option(NOVA_HIP_OPT_TILE_RANK, r ? std::atoll(r) : 0));
    }
    if (const char* c = std::getenv("NOVA_HIP_CLAMP_TO_EDGE")) {
      HIP_CHECK_AND_THROW(nova_hip_set_option(NOVA_HIP_OPT_CLAMP_TO_EDGE, std::atoll(c) != 0 ? 1 : 0));
    }
    if (const char* s = std::getenv("NOVA_HIP_FIXED_SEED")) {
      fixed_seed = true;
      seed = static_cast<uint32_t>(std::strtoul(s, nullptr, 10));
    }
  }

  void add_kernel(const std::string& kernel_name) { (void) kernel_name; }

  template <typename Kernel, typename... Args>
  void call_kernel(const Kernel& kernel, uint2 global_dims, uint2 local_dims, Args&&... args) {
    align_dims(global_dims, local_dims);
    dim3 block_size { local_dims.x, local_dims.y, 1 };
    dim3 num_blocks { global_dims.x / block_size.x, global_dims.y / block_size.y, 1 };
    kernel(num_blocks, block_size, std::forward<Args>(args).data()...);
  }

  // kernel_raytrace's second argument is the wall-clock seed (raytracer.cpp:175-176).  With
  // NOVA_HIP_FIXED_SEED set, Raytracer's time wrapper can be routed through this helper.
  uint32_t seed_for_launch(uint32_t wall_clock_ms) { return fixed_seed ? seed + launches++ : wall_clock_ms; }

  template <typename T>
  Image2DRead<T> create_image2D_read(ImageChannelOrder, ImageChannelType, AddressMode address_mode,
                                     FilterMode filter_mode, bool normalized_coords, size_t width,
                                     size_t height, std::vector<T>& data) const {
    if (data.empty() || width == 0 || height == 0) {
      throw AcceleratorException("Cannot build an empty Image2DRead");
    }
    return Image2DRead<T>(address_mode, filter_mode, normalized_coords, width, height, data);
  }
  template <typename T>
  Image2DRead<T> create_image2D_read(ImageChannelOrder, ImageChannelType, AddressMode address_mode,
                                     FilterMode filter_mode, bool normalized_coords, size_t width,
                                     size_t height) const {
    if (width == 0 || height == 0) {
      throw AcceleratorException("Cannot build an empty Image2DRead");
    }
    return Image2DRead<T>(address_mode, filter_mode, normalized_coords, width, height);
  }
  template <typename T>
  Image2DWrite<T> create_image2D_write(ImageChannelOrder, ImageChannelType, size_t width,
                                       size_t height) const {
    if (width == 0 || height == 0) {
      throw AcceleratorException("Cannot build an empty Image2DWrite");
    }
    return Image2DWrite<T>(width, height);
  }
  template <typename T>
  Image2DReadWrite<T> create_image2D_readwrite(ImageChannelOrder, ImageChannelType,
                                               AddressMode address_mode, FilterMode filter_mode,
                                               bool normalized_coords, size_t width, size_t height,
                                               std::vector<T>& data) const {
    if (data.empty() || width == 0 || height == 0) {
      throw AcceleratorException("Cannot build an empty Image2DReadWrite");
    }
    return Image2DReadWrite<T>(address_mode, filter_mode, normalized_coords, width, height, data);
  }
  template <typename T>
  Image2DReadWrite<T> create_image2D_readwrite(ImageChannelOrder, ImageChannelType,
                                               AddressMode address_mode, FilterMode filter_mode,
                                               bool normalized_coords, size_t width,
                                               size_t height) const {
    if (width == 0 || height == 0) {
      throw AcceleratorException("Cannot build an empty Image2DReadWrite");
    }
    return Image2DReadWrite<T>(address_mode, filter_mode, normalized_coords, width, height);
  }
  template <typename T>
  Image2DArray<T> create_image2D_array(ImageChannelOrder, ImageChannelType, AddressMode address_mode,
                                       FilterMode filter_mode, bool normalized_coords,
                                       size_t array_size, size_t width, size_t height,
                                       std::vector<T>& data) const {
    if (data.empty() || array_size == 0 || width == 0 || height == 0) {
      throw AcceleratorException("Cannot build an empty Image2DArray");
    }
    return Image2DArray<T>(address_mode, filter_mode, normalized_coords, array_size, width, height, data);
  }

  template <typename T>
  std::vector<T> read_image2D(const Image2DWrite<T>& image, size_t, size_t) const { return image.read(); }
  template <typename T>
  std::vector<T> read_image2D(const Image2DReadWrite<T>& image, size_t, size_t) const { return image.read(); }
  template <typename T>
  void write_image2D(Image2DReadWrite<T>& image, size_t, size_t, const std::vector<T>& data) const {
    image.write(data);
  }
  template <typename T>
  void copy_image2D(Image2DRead<T>& dst, const Image2DWrite<T>& src, size_t, size_t) const { dst.copy_from(src); }
  template <typename T>
  void copy_image2D(Image2DRead<T>& dst, const Image2DReadWrite<T>& src, size_t, size_t) const { dst.copy_from(src); }
  template <typename T>
  void fill_image2D(Image2DRead<T>& image, size_t, size_t, const T& t) const { image.fill(t); }

  template <typename T>
  Buffer<T> create_buffer(MemFlags, std::vector<T>& data) const {
    if (data.empty()) {
      throw AcceleratorException("Cannot build an empty Buffer");
    }
    return Buffer<T>(data.size(), data.data());
  }
  template <typename T>
  Buffer<T> create_buffer(MemFlags, size_t length) const {
    if (length == 0) {
      throw AcceleratorException("Cannot build an empty Buffer");
    }
    return Buffer<T>(length);
  }
  template <typename T>
  Buffer<T> create_buffer(MemFlags, T data) const { return Buffer<T>(1, &data); }

  template <typename T>
  void fill_buffer(Buffer<T>& buf, size_t length, const T& t) const { buf.fill(length, t); }
  template <typename T>
  void write_buffer(Buffer<T>& buf, const std::vector<T>& v) const { buf.write(v); }
  template <typename T>
  void write_buffer(Buffer<T>& buf, const T& t) const { buf.write(t); }
  template <typename T>
  std::vector<T> read_buffer(const Buffer<T>& buf, size_t length) const { return buf.read(length); }
  template <typename T>
  T read_buffer(const Buffer<T>& buf) const { return buf.read(); }

  template <typename T, typename... Args>
  Wrapper<T> create_wrapper(Args&&... args) const { return Wrapper<T>(std::forward<Args>(args)...); }

private:
  bool fixed_seed = false;
  uint32_t seed = 0;
  uint32_t launches = 0;
};

}

#endif // HIP_ACCELERATOR_HPP
