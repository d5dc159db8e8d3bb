// Replaces src/backend/cuda/types/error.hpp:10-31: nova::Error carries the code a C-ABI call
// returned (hipError_t > 0 or NOVA_HIP_E_* < 0); main.cpp:21-23 prints get_error_string(e.err()).
#ifndef HIP_ERROR_HPP
#define HIP_ERROR_HPP

#include <iostream>
#include <stdexcept>

#include "nova_hip.h"

namespace nova {

inline const char* get_error_string(int code) { return nova_hip_error_string(code); }

struct Error : public std::runtime_error {
  explicit Error(int error) : std::runtime_error(get_error_string(error)), error(error) {}
  int err() const { return error; }
  int error;
};

}

#define HIP_CHECK(result)                                                                       \
  do {                                                                                          \
    int nova_hip_rc_ = (result);                                                                \
    if (nova_hip_rc_ != 0) {                                                                    \
      std::cerr << __FILE__ << ": line " << __LINE__ << " " << nova::get_error_string(nova_hip_rc_) \
                << std::endl;                                                                   \
    }                                                                                           \
  } while (0)

#define HIP_CHECK_AND_THROW(result)                            \
  do {                                                         \
    int nova_hip_rc_ = (result);                               \
    if (nova_hip_rc_ != 0) {                                   \
      std::cerr << __FILE__ << ": line " << __LINE__ << " ";   \
      throw nova::Error(nova_hip_rc_);                         \
    }                                                          \
  } while (0)

#endif // HIP_ERROR_HPP
