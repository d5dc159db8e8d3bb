// Image2DRead / Image2DWrite / Image2DReadWrite / Image2DArray for the HIP backend: linear HBM
// buffers with the sampler state kept in a nova_hip_image descriptor (CDNA has no image or
// sampler hardware).  Replace src/backend/cuda/types/image2D.hpp, image2D_read.hpp,
// image2D_write.hpp, image2D_readwrite.hpp and image2D_array.hpp (cudaArray + texture/surface
// objects).  `.data()` yields the handle the kernel entry points take; a default-constructed
// image yields nullptr, which the launchers accept for the denoise feature images when
// denoise_available == 0 (src/core/raytracer.cpp:81-100, :187, :196-198).
#ifndef HIP_IMAGE2D_HPP
#define HIP_IMAGE2D_HPP

#include <utility>
#include <vector>

#include "backend/hip/types/error.hpp"
#include "backend/hip/types/flags.hpp"

namespace nova {

using ImageHandle = nova_hip_image*;

template <typename T>
class Image2D {
public:
  Image2D() : handle(nullptr) {}
  Image2D(AddressMode address_mode, FilterMode filter_mode, bool normalized_coords, size_t width,
          size_t height, size_t layers, const std::vector<T>& data)
    : handle(nullptr) {
    static_assert(sizeof(T) == 16 || sizeof(T) == 4, "float4 or uchar4 texels");
    const uint32_t am = address_mode == AddressMode::WRAP ? NOVA_ADDRESS_WRAP : NOVA_ADDRESS_CLAMP;
    HIP_CHECK_AND_THROW(nova_hip_image_create(
      &handle, sizeof(T), static_cast<uint32_t>(width), static_cast<uint32_t>(height),
      static_cast<uint32_t>(layers), am, static_cast<uint32_t>(filter_mode), normalized_coords,
      data.empty() ? nullptr : data.data()));
  }
  ~Image2D() { HIP_CHECK(nova_hip_image_destroy(handle)); }

  Image2D(const Image2D&) = delete;
  Image2D& operator=(const Image2D&) = delete;
  Image2D(Image2D&& other) : handle(other.handle) { other.handle = nullptr; }
  Image2D& operator=(Image2D&& other) {
    std::swap(handle, other.handle);
    return *this;
  }

  std::vector<T> read() const {
    std::vector<T> host(handle ? static_cast<size_t>(handle->width) * handle->height * handle->layers : 0);
    if (handle) HIP_CHECK_AND_THROW(nova_hip_image_read(handle, host.data()));
    return host;
  }
  void write(const std::vector<T>& data) { HIP_CHECK_AND_THROW(nova_hip_image_write(handle, data.data())); }
  void fill(const T& t) { HIP_CHECK_AND_THROW(nova_hip_image_fill(handle, &t)); }
  template <typename Other>
  void copy_from(const Other& src) { HIP_CHECK_AND_THROW(nova_hip_image_copy(handle, src.raw())); }

  ImageHandle raw() const { return handle; }

protected:
  ImageHandle handle;
};

template <typename T>
class Image2DRead : public Image2D<T> {
public:
  Image2DRead() = default;
  Image2DRead(AddressMode a, FilterMode f, bool n, size_t w, size_t h, const std::vector<T>& data = {})
    : Image2D<T>(a, f, n, w, h, 1, data) {}
  ImageHandle& data() { return this->handle; }
  const ImageHandle& data() const { return this->handle; }
};

template <typename T>
class Image2DWrite : public Image2D<T> {
public:
  Image2DWrite() = default;
  Image2DWrite(size_t w, size_t h)
    : Image2D<T>(AddressMode::CLAMP, FilterMode::NEAREST, false, w, h, 1, {}) {}
  ImageHandle& data() { return this->handle; }
  const ImageHandle& data() const { return this->handle; }
};

template <typename T>
class Image2DReadWrite : public Image2D<T> {
public:
  Image2DReadWrite() = default;
  Image2DReadWrite(AddressMode a, FilterMode f, bool n, size_t w, size_t h, const std::vector<T>& data = {})
    : Image2D<T>(a, f, n, w, h, 1, data) {}

  // same accessor objects as cuda/types/image2D_readwrite.hpp:67-80; both views are the one buffer
  struct ReadAccessor {
    const Image2DReadWrite* image;
    ImageHandle data() const { return image->handle; }
  };
  struct WriteAccessor {
    const Image2DReadWrite* image;
    ImageHandle data() const { return image->handle; }
  };
  ReadAccessor read_access() const { return ReadAccessor { this }; }
  WriteAccessor write_access() const { return WriteAccessor { this }; }
};

template <typename T>
class Image2DArray : public Image2D<T> {
public:
  Image2DArray() = default;
  Image2DArray(AddressMode a, FilterMode f, bool n, size_t array_size, size_t w, size_t h,
               const std::vector<T>& data)
    : Image2D<T>(a, f, n, w, h, array_size, data) {}
  ImageHandle& data() { return this->handle; }
  const ImageHandle& data() const { return this->handle; }
};

}

#endif // HIP_IMAGE2D_HPP
