/*
 * nova_hip.h — C ABI of the MI355X (gfx950) ray-tracing backend for Nova.
 *
 * This is the drop-in boundary: everything Nova's `src/backend` abstraction needs from a device
 * backend (buffers, images, the three kernel launches), as plain C entry points with plain
 * pointers and sizes.  The C++ header-only wrappers in nova_amd/backend/hip/ (the code a
 * `-DBACKEND=HIP` build of Nova compiles in place of src/backend/cuda/) call only these
 * functions.  Every entry point cites the reference interface it replaces; paths are relative
 * to the Nova source tree.
 *
 * Conventions
 *   - every function returns 0 on success, otherwise a hipError_t value (>0) or one of the
 *     NOVA_HIP_E_* codes (<0); nothing throws across this boundary;
 *   - what a caller can observe is what the reference's synchronous calls leave (`cudaDeviceSynchronize()` after
 *     every launch, src/backend/cuda/accelerator/accelerator.hpp:26-29): the four per-sample launches
 *     (nova_hip_launch_raytrace / _accumulate / _post_process, nova_hip_image_copy) return once enqueued and every other
 *     entry point waits for them first (NOVA_HIP_OPT_ASYNC_LAUNCH = 0: each waits itself); single host thread per device;
 *   - device pointers handed to the launchers must come from nova_hip_malloc (the library keeps
 *     a ptr -> size registry so that the launchers know how many nodes/triangles a buffer holds:
 *     the reference's kernels receive bare pointers, kernels/kernels.cu:13-16).
 */
#ifndef NOVA_HIP_H
#define NOVA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * POD layouts: the headers of src/kernel_types/ as a HIP (or CUDA) host compiler lays them out, i.e. a
 * 3-vector is 12 bytes, float2 is 8-byte aligned, float4 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
typedef struct { float x, y; } __attribute__((aligned(8))) nova_float2;
typedef struct { float x, y, z; } nova_float3;
typedef struct { float x, y, z, w; } __attribute__((aligned(16))) nova_float4;
typedef struct { uint8_t x, y, z, w; } __attribute__((aligned(4))) nova_uchar4;

/* src/kernel_types/bvh_node.hpp:13-16 */
typedef struct {
  nova_float4 top_offset_left;  /* xyz = box max; w = first triangle (leaf) */
  nova_float4 bottom_num_right; /* xyz = box min; w = right child (>= 0) or -count (leaf) */
} nova_bvh_node;                /* 32 bytes */

/* src/kernel_types/triangle.hpp:10-12 — rows of the Woop world->unit-triangle affine map */
typedef struct { nova_float4 x, y, z; } nova_triangle_data; /* 48 bytes */

/* src/kernel_types/triangle.hpp:14-35 */
typedef struct {
  nova_float3 normal1, normal2, normal3;
  nova_float3 tangent1, tangent2, tangent3;
  nova_float3 bitangent1, bitangent2, bitangent3;
  nova_float2 texture_coord1, texture_coord2, texture_coord3;
  nova_float3 kD;
  float metallic;
  float roughness;
  int32_t diffuse_index;
  int32_t metallic_index;
  int32_t roughness_index;
  int32_t normal_index;
  int32_t light_index;
} nova_triangle_meta; /* 176 bytes */

/* src/kernel_types/area_light.hpp:8-32 */
enum { NOVA_LIGHT_RECT = 0, NOVA_LIGHT_TRI = 1 };
typedef struct {
  nova_float3 intensity;
  union {
    struct { nova_float3 position, normal, dims; } rect;
    struct { nova_float3 v1, v2, v3; } tri;
  };
  uint8_t type; /* AreaLightType : bool */
} nova_area_light; /* 52 bytes */

/* src/kernel_types/eye_coords.hpp:9-14 and matrix.hpp:8-12 */
typedef struct { nova_float3 x, y, z; } nova_mat3x3;
typedef struct {
  nova_float2 coord_scale;
  nova_float2 coord_dims;
  nova_float3 eye_pos;
  nova_mat3x3 eye_coord_frame;
} nova_eye_coords; /* 64 bytes */

/* src/kernel_types/scene_params.hpp:9-18 */
typedef struct {
  nova_eye_coords eye_coords;
  nova_float3 shading_diffuse;
  float shading_metallic;
  float shading_roughness;
  char path_tracing;
  int32_t num_samples;
  float exposure;
  char anti_aliasing;
} nova_scene_params; /* 104 bytes */

/* ---------------------------------------------------------------------------------------------
 * Errors.  Replaces nova::Error / get_error_string (src/backend/cuda/types/error.hpp:10-17).
 * ------------------------------------------------------------------------------------------- */
enum {
  NOVA_HIP_OK = 0,
  NOVA_HIP_E_INVALID_ARGUMENT = -1, /* null/zero-sized/ill-formed argument */
  NOVA_HIP_E_UNKNOWN_BUFFER = -2,   /* device pointer not allocated by nova_hip_malloc */
  NOVA_HIP_E_BAD_SCENE = -3,        /* BVH does not parse (child index out of range, ...) */
  NOVA_HIP_E_NO_DEVICE = -4,        /* no HIP device / device ordinal out of range */
  NOVA_HIP_E_TOO_LARGE = -5,        /* request exceeds a documented maximum */
  NOVA_HIP_E_NOT_CACHED = -6,       /* nova_hip_trace_rays_noalloc: no traversal layout cached for the buffers */
  NOVA_HIP_E_NO_RCCL = -7,          /* nova_hip_tiles_comm_* / _allgather: librccl could not be opened */
  NOVA_HIP_E_NO_COMM = -8,          /* nova_hip_tiles_allgather without a communicator */
  NOVA_HIP_E_RCCL_BASE = -1000      /* an ncclResult_t r > 0 is returned as NOVA_HIP_E_RCCL_BASE - r */
};
const char* nova_hip_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * Device.  The reference picks its device in the Accelerator constructor
 * (src/backend/opencl/accelerator/accelerator.cpp:11-25; CUDA: implicit device 0).
 * nova_hip_init selects the device for the calling process and creates the library's per-device
 * context (stream, work queues).  It is idempotent.
 * ------------------------------------------------------------------------------------------- */
int nova_hip_init(int device_ordinal);
int nova_hip_device_count(int* count);
int nova_hip_device_name(char* out, size_t out_bytes);
int nova_hip_synchronize(void);
/* Frees everything the library owns on the device (cached scenes, work buffers, streams, events) and
 * forgets the device: a later nova_hip_init may pick another one.  Buffers and images the caller made
 * stay valid until the caller frees them. */
int nova_hip_shutdown(void);

/* ---------------------------------------------------------------------------------------------
 * Buffers.  Replace nova::Buffer<T> (src/backend/cuda/types/buffer.hpp:11-86): constructor
 * (cudaMalloc [+ upload]) -> nova_hip_malloc [+ nova_hip_memcpy_h2d]; destructor -> nova_hip_free;
 * write -> memcpy_h2d; read -> memcpy_d2h; fill -> nova_hip_fill.
 * ------------------------------------------------------------------------------------------- */
int nova_hip_malloc(void** dev_ptr, size_t bytes);
int nova_hip_free(void* dev_ptr);
/* Page-locked host memory for the destinations of nova_hip_memcpy_d2h / nova_hip_image_read.  No counterpart in
 * the reference, whose read_image2D / read_buffer return a fresh std::vector (src/backend/cuda/accelerator/
 * accelerator.hpp:130-142, :213-219), i.e. a pageable copy every sample (src/core/raytracer.cpp:253-255): a caller
 * that reads the frame back per sample keeps one such buffer instead. */
int nova_hip_host_alloc(void** host_ptr, size_t bytes);
int nova_hip_host_free(void* host_ptr);
int nova_hip_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int nova_hip_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int nova_hip_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes);
/* dst[i] = *pattern for i in [0, count); pattern_bytes in {1, 2, 4, 8, 16} */
int nova_hip_fill(void* dst_dev, const void* pattern, size_t pattern_bytes, size_t count);

/* ---------------------------------------------------------------------------------------------
 * Images.  CDNA has no image/sampler hardware, so an image is a linear row-major HBM buffer
 * plus the sampler state the reference attaches at creation.  Replace Image2DRead/Write/
 * ReadWrite/Array (src/backend/cuda/types/image2D*.hpp) and Accelerator::create_image2D_*,
 * read_image2D, write_image2D, copy_image2D, fill_image2D
 * (src/backend/cuda/accelerator/accelerator.hpp:32-177).
 * Sampling semantics follow the reference's OpenCL path (kernels/backend/opencl/image.hpp:14-21):
 * float coordinates = normalised, LINEAR filter; CLAMP = border colour 0; WRAP = REPEAT.
 * ------------------------------------------------------------------------------------------- */
enum { NOVA_ADDRESS_WRAP = 0, NOVA_ADDRESS_CLAMP = 1 };
enum { NOVA_FILTER_NEAREST = 0, NOVA_FILTER_LINEAR = 1 };
typedef struct nova_hip_image {
  void* data;          /* device pointer, width*height*layers*texel_bytes bytes */
  uint32_t width;
  uint32_t height;
  uint32_t layers;     /* 1 for a plain 2D image */
  uint32_t texel_bytes;/* 16 = float4, 4 = uchar4 */
  uint32_t address_mode;
  uint32_t filter_mode;
  uint32_t normalized_coords;
} nova_hip_image;

/* `host_data` may be NULL: the image then starts as zeros. */
int nova_hip_image_create(nova_hip_image** image, uint32_t texel_bytes, uint32_t width,
                          uint32_t height, uint32_t layers, uint32_t address_mode,
                          uint32_t filter_mode, uint32_t normalized_coords,
                          const void* host_data);
int nova_hip_image_destroy(nova_hip_image* image);
int nova_hip_image_read(const nova_hip_image* image, void* host_dst);
int nova_hip_image_write(nova_hip_image* image, const void* host_src);
int nova_hip_image_copy(nova_hip_image* dst, const nova_hip_image* src);
int nova_hip_image_fill(nova_hip_image* image, const void* texel);

/* ---------------------------------------------------------------------------------------------
 * The three kernels.  Argument lists are those of kernels/kernels.cu:9-22, :42-53, :94-97
 * (= kernels/entrypoint/cuda/entrypoint.hpp:12-49 without the launch geometry: the reference's
 * 8x4 work-groups would idle half of every wave64, so geometry is the backend's business).
 * Buffers are the device pointers of the reference's layouts above; the BVH is repacked into the
 * backend's own 64-byte node layout on first use and cached until a buffer is freed or rewritten.
 * `albedo_feature*`/`normal_feature*` may be NULL when denoise_available == 0 and `lights` may be
 * NULL when num_lights == 0 (src/core/raytracer.cpp:81-100, :129-132).
 * `materials` is the reference's float4 layered array, or (the texture path,
 * src/material/material_loader.cpp:39-80) a uchar4 one (texel_bytes == 4) holding the layers as
 * MaterialLoader has them BEFORE its last step: the uint8 -> float conversion x / 255.0f (:69-76) then happens
 * at sampling time on the device, with the same correctly rounded division, so every sampled value — and the
 * frame — is bit-identical while the array takes a quarter of the bytes and of the texture-gather traffic.
 * ------------------------------------------------------------------------------------------- */
int nova_hip_launch_raytrace(const nova_scene_params* params, uint32_t time,
                             nova_hip_image* temp_color1, uint32_t width, uint32_t height,
                             const nova_triangle_data* triangles,
                             const nova_triangle_meta* tri_meta, const nova_bvh_node* bvh,
                             const nova_area_light* lights, uint32_t num_lights,
                             const nova_hip_image* materials, const nova_hip_image* sky,
                             uint32_t denoise_available, nova_hip_image* albedo_feature1,
                             nova_hip_image* normal_feature1);

int nova_hip_launch_accumulate(int32_t sample_index, uint32_t denoise_available,
                               const nova_hip_image* temp_color1,
                               const nova_hip_image* albedo_feature1,
                               const nova_hip_image* normal_feature1,
                               const nova_hip_image* prev_color,
                               const nova_hip_image* prev_albedo_feature,
                               const nova_hip_image* prev_normal_feature,
                               nova_hip_image* temp_color2, nova_hip_image* albedo_feature2,
                               nova_hip_image* normal_feature2, uint32_t width, uint32_t height);

int nova_hip_launch_post_process(const nova_scene_params* params,
                                 const nova_hip_image* temp_color2, nova_hip_image* pixels,
                                 uint32_t width, uint32_t height);

/* ---------------------------------------------------------------------------------------------
 * Several samples in one call.  Replaces num_samples consecutive calls of
 * Raytracer::raytrace's device schedule (src/core/raytracer.cpp:169-262: kernel_raytrace,
 * kernel_accumulate, copy_image2D of the running mean and features, kernel_post_process) — the
 * reference waits for the device three times and copies one to three whole images per sample
 * (:195-210, :244-255).  Here every kernel of every sample is enqueued at once and the host waits
 * once; the running mean ping-pongs between prev_color and temp_color2 instead of being copied.
 * Sample k (k = 0 .. num_samples-1) is seeded with times[k] and accumulated with sample index
 * first_sample_index + k.  On return every image holds exactly the bytes the four-call schedule
 * would have left: temp_color1 / *_feature1 = the last sample, temp_color2 == prev_color = the mean
 * over all samples so far (likewise the features), pixels = kernel_post_process of that mean.
 * The feature images may be NULL when denoise_available == 0.  The OIDN hand-off (:211-242) is
 * not part of it: a caller that denoises does so on the final features.
 * ------------------------------------------------------------------------------------------- */
typedef struct nova_hip_frame_images {
  nova_hip_image* temp_color1;
  nova_hip_image* temp_color2;
  nova_hip_image* prev_color;
  nova_hip_image* albedo_feature1;
  nova_hip_image* normal_feature1;
  nova_hip_image* albedo_feature2;
  nova_hip_image* normal_feature2;
  nova_hip_image* prev_albedo_feature;
  nova_hip_image* prev_normal_feature;
  nova_hip_image* pixels; /* uchar4 */
} nova_hip_frame_images;
int nova_hip_render_samples(const nova_scene_params* params, const uint32_t* times,
                            uint32_t num_samples, int32_t first_sample_index,
                            const nova_hip_frame_images* images, uint32_t width, uint32_t height,
                            const nova_triangle_data* triangles,
                            const nova_triangle_meta* tri_meta, const nova_bvh_node* bvh,
                            const nova_area_light* lights, uint32_t num_lights,
                            const nova_hip_image* materials, const nova_hip_image* sky,
                            uint32_t denoise_available);

/* ---------------------------------------------------------------------------------------------
 * Denoise hand-off, the device half.  Replaces, in Raytracer::raytrace
 * (src/core/raytracer.cpp:211-242), read_image2D x3 + flatten<float4, float, 3> x3 (:216-221) by
 * nova_hip_denoise_read, and pack<float4, float, 3> + write_image2D (:240-241) by nova_hip_denoise_write
 * (templates: src/vector/vector_utils.hpp:8-51).  The float4 -> 3-float conversion runs on the device, the
 * three planes cross PCIe as ONE copy of 36 bytes per pixel (the reference moves 48 and then walks them on
 * the host) into pinned host memory the library owns; the denoiser (OIDN: setImage takes plain pointers,
 * :224-227) reads `color`, `albedo`, `normal` there and writes `output`, which nova_hip_denoise_write sends
 * back and expands into `color`'s float4 texels with w = 0, as pack's value-initialised vector leaves it.
 * The planes stay valid until the next nova_hip_denoise_read with a larger frame or nova_hip_shutdown.  OIDN
 * itself is not part of this library (its binary is absent from the reference tree).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  float* color;  /* HOST pointers, width * height * 3 floats each, row-major, x y z per pixel */
  float* albedo;
  float* normal;
  float* output; /* to be filled by the denoiser before nova_hip_denoise_write */
} nova_hip_denoise_planes;
int nova_hip_denoise_read(const nova_hip_image* color, const nova_hip_image* albedo, const nova_hip_image* normal,
                          uint32_t width, uint32_t height, nova_hip_denoise_planes* planes);
int nova_hip_denoise_write(nova_hip_image* color, uint32_t width, uint32_t height);

/* ---------------------------------------------------------------------------------------------
 * Tile stacks for the multi-GPU gather.  No counterpart in the reference (one device,
 * src/backend/opencl/accelerator/accelerator.cpp:11-25); SURVEY.md §8e.  A frame is cut into 64x64 tiles,
 * tile t belongs to rank t % ranks.  nova_hip_tiles_pack writes rank's tiles of a float4 or uchar4 image
 * into one contiguous stack [k][64][64] (tile id = rank + k * ranks; zeros beyond the image border and the
 * last tile) of nova_hip_tiles_stack_bytes bytes — the send buffer of ONE collective per final frame
 * (ncclAllGather / ncclSend-Recv to the root).  nova_hip_tiles_unpack scatters the `ranks` stacks, laid
 * out one after the other as an all-gather leaves them, into the whole image.  stack_dev / stacks_dev are
 * any device pointers of sufficient size (the collective's buffers need not come from nova_hip_malloc).
 * With anti_aliasing: gather the float4 running means (temp_color2), unpack on the root, set
 * NOVA_HIP_OPT_TILE_RANKS = 1 there and run nova_hip_launch_post_process on the whole frame.  Without:
 * post-process per rank and gather the uchar4 pixels (a quarter of the bytes).
 * ------------------------------------------------------------------------------------------- */
uint64_t nova_hip_tiles_stack_bytes(uint32_t width, uint32_t height, uint32_t ranks, uint32_t texel_bytes);
int nova_hip_tiles_pack(const nova_hip_image* image, uint32_t rank, uint32_t ranks, void* stack_dev);
int nova_hip_tiles_unpack(nova_hip_image* image, uint32_t ranks, const void* stacks_dev);

/* The gather itself: RCCL (ncclAllGather over xGMI) behind the C ABI.  No counterpart in the reference, which has
 * one device and no collective (src/backend/opencl/accelerator/accelerator.cpp:11-25); it is the per-rank hook a
 * multi-process launch of Nova runs once per final frame, before image_utils::write_image
 * (src/scene/scene.cpp:320-331; INTEGRATION.md §6, integration/multi_gpu_hook.cpp).  One process per GPU:
 *   rank 0:      nova_hip_tiles_unique_id(id)            (ncclGetUniqueId; hand the 128 bytes to the other ranks by
 *                                                          any means: a file, a socket, MPI, torch.distributed)
 *   every rank:  nova_hip_tiles_comm_init(id, rank, n)   (ncclCommInitRank on the library's device; collective)
 *                ... render with NOVA_HIP_OPT_TILE_RANK / _RANKS = rank / n ...
 *   every rank:  nova_hip_tiles_allgather(image)         (pack -> ncclAllGather -> unpack on the library's stream:
 *                                                          afterwards EVERY rank's image holds all ranks' tiles)
 * A caller that already has an ncclComm_t hands it over with nova_hip_tiles_comm_adopt (the library then never
 * destroys it; the handle must come from the librccl the library itself uses — nova_hip_comm_info.rccl_path — and
 * live on the library's device).  librccl is opened on first use, from the directory of the HIP runtime the library is bound to
 * (env NOVA_HIP_RCCL = full path overrides); single-GPU callers never load it.  The send and receive stacks are
 * the library's own buffers, grown on demand.  With anti_aliasing gather temp_color2 (float4), set
 * NOVA_HIP_OPT_TILE_RANKS = 1 on the rank that writes the frame and post-process there; without it post-process
 * per rank and gather the uchar4 pixels. */
#define NOVA_HIP_UNIQUE_ID_BYTES 128
typedef struct {
  uint32_t rank, ranks;   /* of the current communicator; ranks == 0: none */
  int32_t device;         /* HIP device ordinal of the library's context */
  int32_t rccl_version;   /* ncclGetVersion, 0 if librccl is not loaded */
  char pci_bus_id[32];    /* hipDeviceGetPCIBusId of that device: distinct per rank on a multi-GPU node */
  char rccl_path[256];    /* the librccl in use */
} nova_hip_comm_info;
int nova_hip_tiles_unique_id(void* id_out /* NOVA_HIP_UNIQUE_ID_BYTES */);
int nova_hip_tiles_comm_init(const void* unique_id, uint32_t rank, uint32_t ranks);
int nova_hip_tiles_comm_adopt(void* nccl_comm /* ncclComm_t */);
int nova_hip_tiles_comm_destroy(void);
int nova_hip_tiles_comm_info(nova_hip_comm_info* out);
int nova_hip_tiles_allgather(nova_hip_image* image);

/* ---------------------------------------------------------------------------------------------
 * Ray batches.  find_intersection (kernels/raytrace.hpp:38-102) over an array of rays, for
 * callers that generate their own rays (BASELINE config 4: 16 M incoherent rays per batch).
 * Each ray is built as the reference builds it at every call site (raytrace.hpp:132,216,239):
 * Ray(point, direction, RAY_EPSILON).  rays_point_tmax[i] = {point.xyz, tmax} (tmax = FLT_MAX for
 * an unbounded ray), rays_dir_flags[i] = {direction.xyz, as_float(flags)}, flags bit 0 = any-hit.
 * hits[i] = {t, barycentric.y, barycentric.z, as_float(int triangle index or -1)}.
 * All three arrays are device pointers from nova_hip_malloc.
 * ------------------------------------------------------------------------------------------- */
int nova_hip_trace_rays(const nova_triangle_data* triangles, const nova_bvh_node* bvh,
                        const nova_float4* rays_point_tmax, const nova_float4* rays_dir_flags,
                        nova_float4* hits, uint64_t num_rays);

/* nova_hip_trace_rays for callers that cannot allocate.  No counterpart in the reference: it is for tight
 * serving loops and harnesses that forbid hipMalloc (and, for code that builds HIP graphs, for the part
 * OUTSIDE a capture: the call is synchronous on the library's own stream — it enqueues there, not on the
 * caller's, and waits with hipStreamSynchronize — so it must not be made while a stream capture is active).
 * Same arguments, same kernels and bit-identical hits, but the call makes no hipMalloc / hipFree / event or
 * any other allocating call: the launch's queue heads and the traversal-stack entries beyond the LDS part live in
 * a static device array of 32 MiB + 8 KiB that every GPU the library runs on holds until the process
 * ends.  Conditions: nova_hip_init has run; the traversal layout of (bvh, triangles) is already cached,
 * i.e. nova_hip_trace_rays, nova_hip_launch_raytrace or nova_hip_build_bvh_device has seen these buffers
 * since they were last written (else NOVA_HIP_E_NOT_CACHED); num_rays < 2^31 and
 * (stack depth the tree needs - 16) x (CUs x 5 x 256 threads) <= 8 Mi entries, i.e. depth <= 41 on the
 * 256 CUs of an MI355X (else NOVA_HIP_E_TOO_LARGE).  All calls on one GPU share that array: calls that
 * overlap on one GPU race.  Kernel times are not added to nova_hip_counters. */
int nova_hip_trace_rays_noalloc(const nova_triangle_data* triangles, const nova_bvh_node* bvh,
                                const nova_float4* rays_point_tmax, const nova_float4* rays_dir_flags,
                                nova_float4* hits, uint64_t num_rays);

/* ---------------------------------------------------------------------------------------------
 * Backend options and counters (no counterpart in the reference, which has neither a bounce
 * cap, a fixed seed nor ray counters; SURVEY.md §7 "Hard parts").  Options are per process: they survive
 * nova_hip_shutdown (counters do not).
 * ------------------------------------------------------------------------------------------- */
enum {
  NOVA_HIP_OPT_MAX_DEPTH = 1,   /* max path segments per sample; 0 = unbounded (reference) */
  NOVA_HIP_OPT_TRAVERSAL = 2,   /* 0 = ordered + distance-culled (default);
                                   1 = reference walk: fixed order, no culling (validation) */
  NOVA_HIP_OPT_COUNT_WORK = 3,  /* 1 = also count node visits / triangle tests (slower) */
  NOVA_HIP_OPT_TILE_RANK = 4,   /* image-tile partition for multi-GPU: this process renders the */
  NOVA_HIP_OPT_TILE_RANKS = 5,  /* tiles with (tile_id % ranks) == rank; tiles are 64x64 pixels.  All three kernels
                                   then touch this rank's pixels only (its tiles, plus the few pixels of neighbouring
                                   tiles that its own pixels' bilinear reads weight with more than 0) and leave the
                                   rest of every image as it is; kernel_post_process and nova_hip_render_samples
                                   refuse anti_aliasing (FXAA reads across tiles): see nova_hip_tiles_pack */
  NOVA_HIP_OPT_TRAVERSE_KERNEL = 6, /* validation: 0 = production (persistent lanes on the four-wide tree; packet kernel for
                                       camera rays), 6 / 11 = the batch kernels on the four-wide / box-pair tree,
                                       12 = the packet kernel for every launch.  Same hits from all of them */
  NOVA_HIP_OPT_SORT_RAYS = 8,   /* nova_hip_trace_rays: bin the batch by origin cell and direction octant before tracing
                                   (same hits; fewer HBM lines per ray). 0 = for batches of >= 2^20 rays over scenes
                                   larger than 256 MB (default), 1 = never, 2 = always */
  NOVA_HIP_OPT_CLAMP_TO_EDGE = 9, /* what NOVA_ADDRESS_CLAMP means for the frame images that kernel_accumulate and
                                     kernel_post_process sample.  0 (default): the OpenCL backend's CLK_ADDRESS_CLAMP — texels
                                     beyond the image read as 0 (kernels/backend/opencl/image.hpp:14-21).  1: the CUDA backend's
                                     cudaAddressModeClamp — the edge texel repeats (src/backend/cuda/types/flags.hpp:30).
                                     Only FXAA's taps at the outermost pixel ring can tell the difference.  The address mode is
                                     all the switch changes: CUDA's 9-bit filter weights are not modelled. */
  NOVA_HIP_OPT_ALLOC_CALLS = 10, /* read-only (nova_hip_get_option): allocating HIP runtime calls the library has made so far
                                    (hipMalloc / hipFree / pinned memory / event and stream creation); what
                                    tests/test_zz_no_alloc.py checks nova_hip_trace_rays_noalloc against */
  NOVA_HIP_OPT_MIS_PRETEST = 11, /* 1 (default): a MIS ray (kernels/raytrace.hpp:234-263) is first tested against the
                                    triangles that carry a light index; it is traced only if it passes one of them, and
                                    then no farther than that.  The frame is the reference's bit for bit either way
                                    (nova_amd/csrc/shade.hpp gives the argument).  Likewise the shadow ray of a light
                                    sample whose term is +0 in every channel (light below the surface's horizon) is
                                    not traced: blocked or not, the same +0 is added.  0 traces every ray */
  NOVA_HIP_OPT_ASYNC_LAUNCH = 12, /* 1 (default): nova_hip_launch_raytrace / _accumulate / _post_process and
                                    nova_hip_image_copy return once their kernels are enqueued on the library's stream;
                                    every other entry point first waits for what they left there, so reads, writes,
                                    frees and option changes see what the blocking calls of the reference's backends
                                    (queue.finish() / cudaDeviceSynchronize() after every kernel) would have left.  Code
                                    that touches nova_hip_image::data with its own HIP calls calls nova_hip_synchronize
                                    first.  0: each of the four waits for its own kernels */
  NOVA_HIP_OPT_CALL_LANES = 13, /* with NOVA_HIP_OPT_ASYNC_LAUNCH: successive nova_hip_launch_raytrace calls take turns on
                                   this many lanes (own stream, path state and radiance planes each; 1-4, 0 = built-in 3),
                                   so that the kernel_raytrace of one sample runs beside the latency-bound tail and the
                                   image kernels of the sample before.  Each lane renders into planes of its own and
                                   copies them into temp_color1 / the feature images once every call made before it is
                                   done with those images: the images hold what one lane would have left.  With
                                   unbounded paths (NOVA_HIP_OPT_MAX_DEPTH 0) the library also keeps the image calls
                                   that follow a launch (accumulate, copy, post-process) until the sample's lane comes
                                   round, when it knows whether any path is still alive; they run in program order,
                                   at the latest when another entry point waits.  Costs one set of path state per
                                   extra lane (0.9 GB at 1080p) */
  NOVA_HIP_OPT_TUNE = 7         /* internal: bits 0-15 largest queue grab of a wave, in rays (0 = built-in);
                                   bits 16+ ray-queue flags for A/B runs (0 = production setting) */
};
int nova_hip_set_option(int option, int64_t value);
int nova_hip_get_option(int option, int64_t* value);

typedef struct {
  uint64_t rays_primary;      /* find_intersection calls for first path segments */
  uint64_t rays_bounce;       /* closest-hit calls for later segments and MIS rays (every MIS ray of the reference's
                                 schedule, traced or not: see rays_mis_pretested) */
  uint64_t rays_shadow;       /* any-hit calls (every one of the reference's schedule: see rays_shadow_skipped) */
  uint64_t node_visits;       /* 64-byte node lines fetched (NOVA_HIP_OPT_COUNT_WORK) */
  uint64_t triangle_tests;    /* intersects_triangle evaluations (NOVA_HIP_OPT_COUNT_WORK) */
  uint64_t path_segments;     /* shaded hits */
  double traverse_ms;         /* device time of traversal kernels since the last reset (kernels of
                                 nova_hip_render_samples overlap: there the sum exceeds wall time) */
  double shade_ms;            /* device time of generate/shade kernels since the last reset */
  uint64_t traverse_launches;
  double camera_traverse_ms;  /* the part of traverse_ms / traverse_launches spent on camera-ray */
  uint64_t camera_traverse_launches; /* launches (first path segments), which run their own kernel */
  uint64_t rays_mis_pretested; /* of rays_bounce: MIS rays answered by the light pre-test, no traversal
                                  (NOVA_HIP_OPT_MIS_PRETEST) */
  uint64_t rays_shadow_skipped; /* of rays_shadow: shadow rays of light samples whose term is +0, not traced */
} nova_hip_counters;
int nova_hip_get_counters(nova_hip_counters* out);
/* Roofline denominators measured on the device the library runs on (no counterpart in the reference;
 * SURVEY.md §8d): bytes read + written per second by a 1 GiB float4 copy kernel, and the FP32 rate of a
 * loop of independent fused multiply-adds (2 flop each).  Takes a few milliseconds and 2 GiB for its duration. */
typedef struct {
  double hbm_copy_gbs;
  double fp32_fma_tflops;
} nova_hip_device_rates;
int nova_hip_measure_device(nova_hip_device_rates* out);
int nova_hip_reset_counters(void);

/* ---------------------------------------------------------------------------------------------
 * BVH construction on the device.  Replaces BVH::build
 * (src/bvh/bvh.cpp:17-249: binned SAH, 1024/(depth+1) bins, stable split, DFS flattening with the
 * larger child first) and the TriangleData loop of IntersectableManager::build
 * (src/intersectables/intersectable_manager.cpp:129-145).  Same tree, node for node.
 * vertices: num_triangles x (v1 v2 v3) floats, HOST memory, in the order the caller's TriangleMetaData
 * is in.  Outputs are HOST arrays: bvh_out[bvh_capacity >= nova_hip_bvh_max_nodes(num_triangles)],
 * triangles_out[num_triangles] (leaf order), order_out[i] = input index of output triangle i (to
 * permute the meta data with).  At most 2^24 triangles: FlatBVHNode keeps offsets in floats.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  float device_ms;         /* device time of the build, without the copies in and out */
  uint32_t levels;         /* tree depth */
  uint64_t peak_bin_bytes; /* largest per-level bin storage */
  float pack_ms;           /* nova_hip_build_bvh_device: traversal layout from the tree (host clock), 0 if left to the first launch */
} nova_hip_build_stats;
uint64_t nova_hip_bvh_max_nodes(uint64_t num_triangles);
int nova_hip_build_bvh(const float* vertices, uint64_t num_triangles, nova_bvh_node* bvh_out,
                       uint64_t bvh_capacity, uint64_t* num_nodes, nova_triangle_data* triangles_out,
                       uint32_t* order_out, nova_hip_build_stats* stats /* may be NULL */);
/* The same with every array in DEVICE memory from nova_hip_malloc (vertices in, the rest out): the
 * tree is built where nova_hip_launch_raytrace / nova_hip_trace_rays read it, nothing crosses PCIe. */
int nova_hip_build_bvh_device(const float* vertices, uint64_t num_triangles, nova_bvh_node* bvh_out,
                              uint64_t bvh_capacity, uint64_t* num_nodes, nova_triangle_data* triangles_out,
                              uint32_t* order_out, nova_hip_build_stats* stats /* may be NULL */);
/* Validation hook (no counterpart in the reference): the traversal layout cached for these device
 * buffers — made on the device by nova_hip_build_bvh_device — against the one the host packer makes from
 * the same FlatBVHNode[]; *different_words = 32-bit words that differ (0 = identical). */
int nova_hip_validate_scene_pack(const nova_bvh_node* bvh, const nova_triangle_data* triangles,
                                 uint64_t* different_words);

#ifdef __cplusplus
}
#endif

#endif /* NOVA_HIP_H */
