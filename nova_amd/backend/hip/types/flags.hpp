// Replaces src/backend/cuda/types/flags.hpp:8-38; address/filter values are those of
// include/nova_hip.h (NOVA_ADDRESS_*, NOVA_FILTER_*).
#ifndef HIP_FLAGS_HPP
#define HIP_FLAGS_HPP

#include "nova_hip.h"

namespace nova {

enum class MemFlags { READ_ONLY, WRITE_ONLY, READ_WRITE };
enum class ImageChannelOrder { R, RGB, RGBA };
enum class ImageChannelType { UINT8, UINT32, INT8, INT32, FLOAT };

// CLAMP follows the reference's OpenCL path (border colour 0, kernels/backend/opencl/image.hpp:14-21);
// MIRROR and BORDER exist in the CUDA enum but no caller uses them, they map to CLAMP.
enum class AddressMode {
  WRAP = NOVA_ADDRESS_WRAP,
  CLAMP = NOVA_ADDRESS_CLAMP,
  MIRROR = NOVA_ADDRESS_CLAMP + 1,
  BORDER = NOVA_ADDRESS_CLAMP + 2,
};
enum class FilterMode { NEAREST = NOVA_FILTER_NEAREST, LINEAR = NOVA_FILTER_LINEAR };

}

#endif // HIP_FLAGS_HPP
