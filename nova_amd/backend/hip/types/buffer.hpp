// nova::Buffer<T> over nova_hip_malloc.  Replaces src/backend/cuda/types/buffer.hpp:11-86
// (same members, same RAII and move behaviour, synchronous uploads and reads).
#ifndef HIP_BUFFER_HPP
#define HIP_BUFFER_HPP

#include <utility>
#include <vector>

#include "backend/hip/types/error.hpp"

namespace nova {

template <typename T>
class Buffer {
public:
  Buffer() : buffer(nullptr), length(0) {}

  Buffer(size_t length, const T* data = nullptr) : buffer(nullptr), length(length) {
    void* p = nullptr;
    HIP_CHECK_AND_THROW(nova_hip_malloc(&p, length * sizeof(T)));
    buffer = static_cast<T*>(p);
    if (data) {
      HIP_CHECK_AND_THROW(nova_hip_memcpy_h2d(buffer, data, length * sizeof(T)));
    }
  }

  ~Buffer() { HIP_CHECK(nova_hip_free(buffer)); }

  Buffer(const Buffer&) = delete;
  Buffer& operator=(const Buffer&) = delete;
  Buffer(Buffer&& other) : buffer(other.buffer), length(other.length) {
    other.buffer = nullptr;
    other.length = 0;
  }
  Buffer& operator=(Buffer&& other) {
    std::swap(buffer, other.buffer);
    std::swap(length, other.length);
    return *this;
  }

  void fill(size_t count, const T& t) {
    std::vector<T> host(count, t);
    HIP_CHECK_AND_THROW(nova_hip_memcpy_h2d(buffer, host.data(), count * sizeof(T)));
  }
  void write(const std::vector<T>& v) {
    HIP_CHECK_AND_THROW(nova_hip_memcpy_h2d(buffer, v.data(), v.size() * sizeof(T)));
  }
  void write(const T& t) { HIP_CHECK_AND_THROW(nova_hip_memcpy_h2d(buffer, &t, sizeof(T))); }

  std::vector<T> read(size_t count) const {
    std::vector<T> host(count);
    HIP_CHECK_AND_THROW(nova_hip_memcpy_d2h(host.data(), buffer, count * sizeof(T)));
    return host;
  }
  T read() const {
    T t;
    HIP_CHECK_AND_THROW(nova_hip_memcpy_d2h(&t, buffer, sizeof(T)));
    return t;
  }

  T*& data() { return buffer; }
  T* const& data() const { return buffer; }

private:
  T* buffer;
  size_t length;
};

}

#endif // HIP_BUFFER_HPP
