// Replaces src/backend/cuda/types/types.hpp (the umbrella header backend/types.hpp includes).
#ifndef HIP_TYPES_HPP
#define HIP_TYPES_HPP

#include "backend/hip/types/buffer.hpp"
#include "backend/hip/types/error.hpp"
#include "backend/hip/types/flags.hpp"
#include "backend/hip/types/image2D.hpp"
#include "backend/hip/types/vector_types.hpp"

#endif // HIP_TYPES_HPP
