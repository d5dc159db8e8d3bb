// TEST INFRASTRUCTURE ONLY — never linked into, imported by or shipped with the product path.
//
// Driver for the reference oracle: includes Nova's device code unmodified, from where it lies
// (-I /root/reference -I /root/reference/src; kernels/kernels.cu and everything it includes),
// on top of the CPU device backend in refshim/cpu_backend.hpp, and exposes it to Python (ctypes)
// with the same plain-C structs the product's C ABI uses (include/nova_hip.h).
//
// Built only inside the build container (oracle/Makefile -> oracle/_ref/libnova_ref.so); it is
// used to (1) generate the golden vectors committed under tests/golden/ and (2) validate the
// CPU restatement in oracle/nova_oracle.cpp.  /root/reference does not exist on the GPU box.
#include "kernels/kernels.cu" // the reference's three kernels + all device logic

#include <cstring>
#include <vector>

#include "nova_hip.h"

namespace refshim {
thread_local int global_id[2] = { 0, 0 };
}

namespace {

using namespace nova;

float3 f3(const nova_float3& v) { return { v.x, v.y, v.z }; }
float4 f4(const nova_float4& v) { return { v.x, v.y, v.z, v.w }; }
float2 f2(const nova_float2& v) { return { v.x, v.y }; }

SceneParams convert(const nova_scene_params& p) {
  SceneParams s;
  std::memset(&s, 0, sizeof(s));
  s.eye_coords.coord_scale = f2(p.eye_coords.coord_scale);
  s.eye_coords.coord_dims = f2(p.eye_coords.coord_dims);
  s.eye_coords.eye_pos = f3(p.eye_coords.eye_pos);
  s.eye_coords.eye_coord_frame = { f3(p.eye_coords.eye_coord_frame.x),
                                   f3(p.eye_coords.eye_coord_frame.y),
                                   f3(p.eye_coords.eye_coord_frame.z) };
  s.shading_diffuse = f3(p.shading_diffuse);
  s.shading_metallic = p.shading_metallic;
  s.shading_roughness = p.shading_roughness;
  s.path_tracing = p.path_tracing;
  s.num_samples = p.num_samples;
  s.exposure = p.exposure;
  s.anti_aliasing = p.anti_aliasing;
  return s;
}

struct Scene {
  std::vector<TriangleData> tris;
  std::vector<TriangleMetaData> meta;
  std::vector<FlatBVHNode> bvh;
  std::vector<AreaLightData> lights;
};

void convert_geometry(Scene& s, const nova_triangle_data* tris, size_t n_tris,
                      const nova_bvh_node* bvh, size_t n_nodes) {
  s.tris.resize(n_tris);
  for (size_t i = 0; i < n_tris; i++) {
    s.tris[i].transform = { f4(tris[i].x), f4(tris[i].y), f4(tris[i].z) };
  }
  s.bvh.resize(n_nodes);
  for (size_t i = 0; i < n_nodes; i++) {
    s.bvh[i].top_offset_left = f4(bvh[i].top_offset_left);
    s.bvh[i].bottom_num_right = f4(bvh[i].bottom_num_right);
  }
}

void convert_shading(Scene& s, const nova_triangle_meta* meta, size_t n_tris,
                     const nova_area_light* lights, size_t n_lights) {
  s.meta.resize(n_tris);
  for (size_t i = 0; i < n_tris; i++) {
    const nova_triangle_meta& m = meta[i];
    TriangleMetaData& d = s.meta[i];
    std::memset(&d, 0, sizeof(d));
    d.normal1 = f3(m.normal1);
    d.normal2 = f3(m.normal2);
    d.normal3 = f3(m.normal3);
    d.tangent1 = f3(m.tangent1);
    d.tangent2 = f3(m.tangent2);
    d.tangent3 = f3(m.tangent3);
    d.bitangent1 = f3(m.bitangent1);
    d.bitangent2 = f3(m.bitangent2);
    d.bitangent3 = f3(m.bitangent3);
    d.texture_coord1 = f2(m.texture_coord1);
    d.texture_coord2 = f2(m.texture_coord2);
    d.texture_coord3 = f2(m.texture_coord3);
    d.kD = f3(m.kD);
    d.metallic = m.metallic;
    d.roughness = m.roughness;
    d.diffuse_index = m.diffuse_index;
    d.metallic_index = m.metallic_index;
    d.roughness_index = m.roughness_index;
    d.normal_index = m.normal_index;
    d.light_index = m.light_index;
  }
  s.lights.resize(n_lights);
  for (size_t i = 0; i < n_lights; i++) {
    AreaLightData& d = s.lights[i];
    std::memset(&d, 0, sizeof(d));
    d.intensity = f3(lights[i].intensity);
    // rect {position, normal, dims} and tri {v1, v2, v3} alias (src/kernel_types/area_light.hpp:27-30)
    d.tri.v1 = f3(lights[i].tri.v1);
    d.tri.v2 = f3(lights[i].tri.v2);
    d.tri.v3 = f3(lights[i].tri.v3);
    d.type = lights[i].type ? AreaLightType::TRI : AreaLightType::RECT;
  }
}

struct HostImage {
  refshim::Image2DArray desc;
  HostImage(const float* rgba, int w, int h, int layers) {
    desc.texels = reinterpret_cast<float4*>(const_cast<float*>(rgba));
    desc.width = w;
    desc.height = h;
    desc.layers = layers;
  }
};

}

extern "C" {

// Struct sizes of the reference's kernel types as compiled here (OpenCL layout), for the tests.
void ref_struct_sizes(int32_t* out) {
  out[0] = sizeof(FlatBVHNode);
  out[1] = sizeof(TriangleData);
  out[2] = sizeof(TriangleMetaData);
  out[3] = sizeof(AreaLightData);
  out[4] = sizeof(SceneParams);
  out[5] = sizeof(EyeCoords);
}

// kernels/kernels.cu:9-40 over every pixel.  Images are float4 arrays (16-byte aligned).
int ref_kernel_raytrace(const nova_scene_params* params, uint32_t time, float* temp_color1,
                        uint32_t width, uint32_t height, const nova_triangle_data* tris,
                        const nova_triangle_meta* meta, uint64_t n_tris, const nova_bvh_node* bvh,
                        uint64_t n_nodes, const nova_area_light* lights, uint32_t num_lights,
                        const float* materials, uint32_t mat_w, uint32_t mat_h, uint32_t mat_layers,
                        const float* sky, uint32_t sky_w, uint32_t sky_h,
                        uint32_t denoise_available, float* albedo1, float* normal1) {
  Scene s;
  convert_geometry(s, tris, n_tris, bvh, n_nodes);
  convert_shading(s, meta, n_tris, lights, num_lights);
  SceneParams sp = convert(*params);
  HostImage mat(materials, mat_w, mat_h, mat_layers);
  HostImage sk(sky, sky_w, sky_h, 1);
  refshim::Image2D out_c { reinterpret_cast<float4*>(temp_color1), (int) width, (int) height, 1 };
  refshim::Image2D out_a { reinterpret_cast<float4*>(albedo1), (int) width, (int) height, 1 };
  refshim::Image2D out_n { reinterpret_cast<float4*>(normal1), (int) width, (int) height, 1 };
  uint2 dims = { width, height };
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = 0; y < (int) height; y++) {
    for (int x = 0; x < (int) width; x++) {
      refshim::global_id[0] = x;
      refshim::global_id[1] = y;
      kernel_raytrace(sp, time, image2d_write_t { &out_c, nullptr }, dims, s.tris.data(),
                      s.meta.data(), s.bvh.data(), s.lights.data(), num_lights, &mat.desc, &sk.desc,
                      denoise_available, image2d_write_t { &out_a, nullptr },
                      image2d_write_t { &out_n, nullptr });
    }
  }
  return 0;
}

// kernels/kernels.cu:42-92
int ref_kernel_accumulate(int32_t sample_index, uint32_t denoise_available, const float* temp_color1,
                          const float* albedo1, const float* normal1, const float* prev_color,
                          const float* prev_albedo, const float* prev_normal, float* temp_color2,
                          float* albedo2, float* normal2, uint32_t width, uint32_t height) {
  auto rd = [&](const float* p) {
    return refshim::Image2D { reinterpret_cast<float4*>(const_cast<float*>(p)), (int) width,
                              (int) height, 1 };
  };
  refshim::Image2D c1 = rd(temp_color1), a1 = rd(albedo1), n1 = rd(normal1), pc = rd(prev_color),
                   pa = rd(prev_albedo), pn = rd(prev_normal), c2 = rd(temp_color2),
                   a2 = rd(albedo2), n2 = rd(normal2);
  uint2 dims = { width, height };
#pragma omp parallel for
  for (int y = 0; y < (int) height; y++) {
    for (int x = 0; x < (int) width; x++) {
      refshim::global_id[0] = x;
      refshim::global_id[1] = y;
      kernel_accumulate(sample_index, denoise_available, &c1, &a1, &n1, &pc, &pa, &pn,
                        image2d_write_t { &c2, nullptr }, image2d_write_t { &a2, nullptr },
                        image2d_write_t { &n2, nullptr }, dims);
    }
  }
  return 0;
}

// kernels/kernels.cu:94-116
int ref_kernel_post_process(const nova_scene_params* params, const float* temp_color2,
                            uint8_t* pixels, uint32_t width, uint32_t height) {
  SceneParams sp = convert(*params);
  refshim::Image2D c2 { reinterpret_cast<float4*>(const_cast<float*>(temp_color2)), (int) width,
                        (int) height, 1 };
  refshim::Image2DU8 out { reinterpret_cast<uchar4*>(pixels), (int) width, (int) height };
  uint2 dims = { width, height };
#pragma omp parallel for
  for (int y = 0; y < (int) height; y++) {
    for (int x = 0; x < (int) width; x++) {
      refshim::global_id[0] = x;
      refshim::global_id[1] = y;
      kernel_post_process(sp, &c2, image2d_write_t { nullptr, &out }, dims);
    }
  }
  return 0;
}

// kernels/raytrace.hpp:38-102 on a ray batch; rays built as at raytrace.hpp:132/216/239.
// hits[i] = {t, bary.y, bary.z, as_float(tri_index)}; tmax = FLT_MAX -> Intersection().
int ref_find_intersection(const nova_triangle_data* tris, uint64_t n_tris, const nova_bvh_node* bvh,
                          uint64_t n_nodes, const float* point_tmax, const float* dir_flags,
                          float* hits, uint64_t n_rays) {
  Scene s;
  convert_geometry(s, tris, n_tris, bvh, n_nodes);
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < (int64_t) n_rays; i++) {
    float3 p = { point_tmax[4 * i], point_tmax[4 * i + 1], point_tmax[4 * i + 2] };
    float3 d = { dir_flags[4 * i], dir_flags[4 * i + 1], dir_flags[4 * i + 2] };
    float tmax = point_tmax[4 * i + 3];
    uint32_t flags;
    std::memcpy(&flags, &dir_flags[4 * i + 3], 4);
    Ray ray(p, d, RAY_EPSILON);
    Intersection intrs(tmax);
    find_intersection(s.tris.data(), s.bvh.data(), ray, intrs, (flags & 1u) != 0);
    hits[4 * i] = intrs.length;
    hits[4 * i + 1] = intrs.barycentric.y;
    hits[4 * i + 2] = intrs.barycentric.z;
    std::memcpy(&hits[4 * i + 3], &intrs.tri_index, 4);
  }
  return 0;
}

// Known-answer helpers for the small device functions.
uint32_t ref_hash(uint32_t seed) { return hash(seed); }            // kernels/random.hpp:11-18
float ref_rand(uint32_t* state) { return rand(*state); }            // kernels/random.hpp:21-26

// kernels/intersection.hpp:40-52; ray built by Ray(point, dir, RAY_EPSILON)
int ref_intersects_aabb(const float* point, const float* dir, const float* top, const float* bottom) {
  Ray ray(float3 { point[0], point[1], point[2] }, float3 { dir[0], dir[1], dir[2] }, RAY_EPSILON);
  return intersects_aabb(ray, float3 { top[0], top[1], top[2] },
                         float3 { bottom[0], bottom[1], bottom[2] });
}

// kernels/brdf.hpp:41-107 — sample / eval / pdf for one input; out = in_dir(3) eval(3) pdf(3)
void ref_brdf(const float* out_dir, const float* normal, const float* diffuse, float metallic,
              float roughness, uint32_t* rng_state, float* out) {
  CookTorranceBRDF b(float3 { out_dir[0], out_dir[1], out_dir[2] },
                     float3 { normal[0], normal[1], normal[2] },
                     float3 { diffuse[0], diffuse[1], diffuse[2] }, metallic, roughness);
  float3 in_dir = b.sample(*rng_state);
  float3 e = b.eval();
  float3 p = b.pdf();
  out[0] = in_dir.x; out[1] = in_dir.y; out[2] = in_dir.z;
  out[3] = e.x; out[4] = e.y; out[5] = e.z;
  out[6] = p.x; out[7] = p.y; out[8] = p.z;
}

}
