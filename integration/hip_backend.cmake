# hip_backend.cmake — the `hip` branch of Nova's CMakeLists.txt as an include file.
#
#   set(BACKEND "hip" ...)                       # CMakeLists.txt:10-18: one more accepted value
#   include(path/to/hip_backend.cmake)
#   nova_hip_backend_configure()                 # instead of the OpenCL / CUDA find_package branch
#   ...
#   nova_hip_backend_sources(BACKEND_SOURCES)    # CMakeLists.txt:54-59: backend sources (headers only)
#   nova_hip_backend_target(${PROJECT_NAME})     # CMakeLists.txt:106-114 and :151-167: link + BACKEND_HIP
#
# Variables: NOVA_HIP_ROOT = directory of this repository (nova_amd/libnova_hip.so, include/, nova_amd/backend/hip).
# No kernels are compiled on the Nova side: the device code is inside libnova_hip.so, and Nova's sources
# are built by the ordinary host compiler (the C ABI has plain pointers and integers only).

set(NOVA_HIP_ROOT "" CACHE PATH "root of the nova_amd repository (holds nova_amd/libnova_hip.so and include/nova_hip.h)")

function(nova_hip_backend_configure)
  if (NOT NOVA_HIP_ROOT)
    message(FATAL_ERROR "BACKEND=hip needs -DNOVA_HIP_ROOT=<nova_amd repository>")
  endif()
  find_library(NOVA_HIP_LIB nova_hip HINTS ${NOVA_HIP_ROOT}/nova_amd ${NOVA_HIP_ROOT} NO_DEFAULT_PATH)
  if (NOT NOVA_HIP_LIB)
    message(FATAL_ERROR "libnova_hip.so not found under ${NOVA_HIP_ROOT}: build it with `python __graft_entry__.py`")
  endif()
  find_path(NOVA_HIP_INCLUDE nova_hip.h HINTS ${NOVA_HIP_ROOT}/include NO_DEFAULT_PATH)
  if (NOT NOVA_HIP_INCLUDE)
    message(FATAL_ERROR "nova_hip.h not found under ${NOVA_HIP_ROOT}/include")
  endif()
  set(HIP_BACKEND ON PARENT_SCOPE)
  message(STATUS "Nova HIP backend: ${NOVA_HIP_LIB}")
endfunction()

# the backend's headers, as src/backend/hip/* would be globbed (kept explicit so IDEs list them)
function(nova_hip_backend_sources OUT_VAR)
  file(GLOB_RECURSE _hip_headers ${NOVA_HIP_ROOT}/nova_amd/backend/hip/*.hpp)
  set(${OUT_VAR} ${_hip_headers} PARENT_SCOPE)
endfunction()

function(nova_hip_backend_target TARGET)
  target_compile_definitions(${TARGET} PRIVATE BACKEND_HIP)
  # "backend/hip/..." resolves inside nova_amd/; nova_hip.h and nova_math.h inside include/
  target_include_directories(${TARGET} PRIVATE ${NOVA_HIP_ROOT}/nova_amd ${NOVA_HIP_INCLUDE})
  target_link_libraries(${TARGET} PRIVATE ${NOVA_HIP_LIB})
  get_filename_component(_hip_libdir ${NOVA_HIP_LIB} DIRECTORY)
  set_property(TARGET ${TARGET} APPEND PROPERTY BUILD_RPATH ${_hip_libdir})
endfunction()
