// Used ONLY when these headers are compiled outside the Nova tree (this repository's self-test):
// minimal equivalents of the three Nova pieces the Accelerator leans on, so the drop-in headers
// can be exercised without Nova's sources.  Inside Nova, accelerator.hpp includes Nova's own
// backend/common/types/wrapper.hpp, backend/common/utils/utils.hpp and
// util/exception/exception.hpp instead (the __has_include switch there).
#ifndef HIP_STANDALONE_HPP
#define HIP_STANDALONE_HPP

#include <stdexcept>
#include <string>
#include <utility>

#include "backend/hip/types/vector_types.hpp"

namespace nova {

// by-value kernel argument holder with .data(), as Accelerator::create_wrapper returns
template <typename T>
class Wrapper {
public:
  Wrapper() = default;
  template <typename... Args>
  explicit Wrapper(Args&&... args) : value { std::forward<Args>(args)... } {}
  const T& data() const { return value; }
  T& data() { return value; }

private:
  T value {};
};

struct AcceleratorException : public std::runtime_error {
  explicit AcceleratorException(const std::string& what) : std::runtime_error(what) {}
};

// round the global size up to a multiple of the local size
inline void align_dims(uint2& global_dims, const uint2& local_dims) {
  global_dims.x = (global_dims.x + local_dims.x - 1) / local_dims.x * local_dims.x;
  global_dims.y = (global_dims.y + local_dims.y - 1) / local_dims.y * local_dims.y;
}

}

#endif // HIP_STANDALONE_HPP
