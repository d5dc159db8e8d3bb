// Wavefront path tracing stages for gfx950: generate and shade.  Together with traverse.hpp they
// replace trace_ray (reference kernels/raytrace.hpp:104-298) and the device helpers it calls:
// brdf.hpp, light.hpp, material.hpp, transforms.hpp, random.hpp.
//
// The reference runs one thread per pixel through an unbounded loop of
//   closest-hit ray -> shade -> [any-hit shadow ray + closest-hit MIS ray] -> sample BRDF -> RR.
// None of a segment's random draws depends on the outcome of its shadow/MIS rays, so one shade
// pass can draw everything for segment k (in the reference's order) and emit all three rays;
// the next pass first RESOLVES segment k's direct lighting from the shadow/MIS hits, then shades
// segment k+1.  Every path therefore advances one segment per (traverse, shade) pair and the
// per-pixel random sequence, arithmetic and result are those of the reference loop.
//
// Path state lives in HBM as six float4 planes (structure of arrays, 96 B per path), ping-ponged
// between passes; surviving paths are compacted with one wave-aggregated atomic per wave.
#pragma once

#include "device_math.hpp"
#include "nova_hip.h"
#include "traverse.hpp" // woop: the light pre-test of MIS rays is the traversal's own triangle test

namespace nova_hip {

struct ImageView {
  const float4* texels;
  int w, h, layers;
  // 8-bit storage: when set, the image is uchar4 texels and texel value = u8 / 255.0f,
  // the very conversion MaterialLoader::build applies on the host before upload (src/material/
  // material_loader.cpp:69-76) — so the sampled values are bit-identical and the array is a quarter of the bytes.
  // lut[x] = float(x) / 255.0f (k_shade fills it in LDS with the same correctly rounded division).
  const uchar4* u8;
  // (a generic pointer on purpose: an address_space(3) pointer is 4 bytes on the device and 8 on the host, and this
  // struct is part of kernel arguments — round 2's one GPU fault; kernel_args_check.hpp now asserts the
  // layout of every by-value kernel argument in both compiler passes)
  const float* lut;
  bool edge; // CLAMP repeats the edge texel (NOVA_HIP_OPT_CLAMP_TO_EDGE) instead of reading 0 beyond the image
};

struct PathPlanes {
  float4* a; // color.xyz, rng state
  float4* b; // throughput.xyz, pixel index (0xFFFFFFFF = empty)
  float4* c; // throughput before this segment's update .xyz, flags
  float4* d; // light-sample term (already multiplied by n.l) .xyz, MIS n.l
  float4* e; // MIS brdf value .xyz, light index of the shaded triangle
  float4* f; // MIS brdf pdf .xyz, slot of the MIS ray among the rays of kind 2 (kFlagHasMis)
};

constexpr uint32_t kFlagPending = 1u;    // direct lighting of the previous segment to resolve
constexpr uint32_t kFlagHasShadow = 2u;  // a light sample (shadow ray) was emitted
constexpr uint32_t kFlagTerminate = 4u;  // path ends after the resolve
constexpr uint32_t kFlagHasMis = 8u;     // a MIS ray was traced (it survived the light pre-test)
constexpr uint32_t kEmptyPixel = 0xFFFFFFFFu;

// Light pre-test of MIS rays (k_shade).  The reference traces a closest-hit ray along the BRDF sample and uses the
// hit only when it is a triangle with a light index other than the shaded triangle's (kernels/raytrace.hpp:234-263).
// Whatever a walk reaches, a triangle it ACCEPTS also passes `woop` on its own with h.t = FLT_MAX (during a walk the
// threshold is only ever lower).  With S = the triangles with such a light index that pass on their own:
//   S empty     -> the closest hit cannot contribute, whatever it is: the ray is not traced;
//   S not empty -> far = max t over S, and the ray is traced with tmax = nextafter(far).  If the reference's closest
//                  hit has t <= far the shortened ray returns that very hit (same triangles reached, same order of
//                  (t, index)); if not, that hit is not in S, contributes nothing, and the shortened ray misses.
// Either way the frame is the reference's bit for bit (oracle/nova_oracle.cpp restates this as `mis_pretest`; the CPU
// suite compares its frames with and without).  ~99 % of the MIS rays of the bench scenes go untraced: a third of
// the closest-hit rays of every bounce.  Scenes with more than kMaxLightTris triangles that carry a light index
// (large emissive meshes) trace every MIS ray as before.
constexpr uint32_t kMaxLightTris = 16;
// the list: word 0 = count (may exceed the cap), then from word 16 on one 16-word record per triangle:
// {triangle index, light index, -, -, Woop matrix rows x, y, z}
constexpr uint32_t kLightTriWords = 16 + 16 * kMaxLightTris;
constexpr int kBlockCounters = 8; // per k_shade block: bounce, MIS, shadow rays, shaded segments, MIS rays not traced, shadow rays not traced

struct FrameArgs {
  nova_scene_params params;
  uint32_t time;
  uint32_t width, height;
  uint32_t tile_rank, tile_ranks, tiles_x, n_my_tiles;
  // under a tile partition: pixels of other ranks' tiles that this rank renders as well, because the
  // pixel-centre bilinear reads of kernel_accumulate / kernel_post_process give them a non-zero weight
  // for one of its own pixels (TilePlan in nova_hip.hip); paths n_my_tiles * 4096 .. + n_extra - 1
  const uint32_t* extra_pixels;
  uint32_t n_extra;
  const nova_triangle_meta* meta;
  const nova_area_light* lights;
  uint32_t num_lights;
  ImageView materials, sky;
  uint32_t denoise;
  float4* out_color;
  float4* out_albedo;
  float4* out_normal;
  uint32_t stride;    // P: ray slot of kind k for path j is k * stride + j
  uint32_t max_depth; // 0 = unbounded
  uint32_t* n_active; // n_active[it] = paths entering pass `it`
  unsigned long long* block_counters; // [block][kBlockCounters]
  uint32_t* n_mis;             // n_mis[it] = MIS rays pass `it`'s traversal reads (kind 2, compacted)
  const uint32_t* light_tris;  // kLightTriWords words (k_find_light_tris), or null: no pre-test
};

// Paths per work-group of k_shade = paths per compaction atomic.  Measured on the 1080p frame (ms of generate + shade
// per sample): 1024 -> 0.550, 512 -> 0.447, 256 -> 0.422, 128 -> 0.543.  At 112 VGPRs a CU holds 16 waves: ONE group of
// 1024, whose waves load, wait at the barrier and store together; four groups of 256 overlap one another's phases.
// Below that the same-address atomics (~12 ns each, one per group) take over.
constexpr uint32_t kShadeBlock = 256;

// ---- software image sampling (CDNA has no sampler): OpenCL 1.2 §8.2 linear filter, normalised
// coordinates; repeat = CLK_ADDRESS_REPEAT, else CLK_ADDRESS_CLAMP (border colour 0), which is
// what the reference's float-coordinate samplers are (kernels/backend/opencl/image.hpp:14-21).
NV_DEV float4 texel(const ImageView& im, int i, int j, int layer) {
  if (i < 0 || j < 0 || i >= im.w || j >= im.h) {
    if (!im.edge) return make_float4(0.f, 0.f, 0.f, 0.f);
    i = min(max(i, 0), im.w - 1);
    j = min(max(j, 0), im.h - 1);
  }
  const size_t k = (static_cast<size_t>(layer) * im.h + j) * im.w + i;
  if (im.u8) {
    const uchar4 t = im.u8[k];
    // the table is in LDS (k_shade): read it as LDS — through the generic pointer every look-up is a flat load
    const float __attribute__((address_space(3)))* lut = (const float __attribute__((address_space(3)))*) im.lut;
    return make_float4(lut[t.x], lut[t.y], lut[t.z], lut[t.w]);
  }
  return im.texels[k];
}
template <bool REPEAT>
NV_DEV V3 sample_linear(const ImageView& im, float s, float t, int layer) {
  float u, v;
  int i0, j0, i1, j1;
  if (REPEAT) {
    u = (s - __builtin_floorf(s)) * static_cast<float>(im.w);
    v = (t - __builtin_floorf(t)) * static_cast<float>(im.h);
    i0 = static_cast<int>(__builtin_floorf(u - 0.5f));
    j0 = static_cast<int>(__builtin_floorf(v - 0.5f));
    i1 = i0 + 1;
    j1 = j0 + 1;
    if (i0 < 0) i0 += im.w;
    if (j0 < 0) j0 += im.h;
    if (i1 > im.w - 1) i1 -= im.w;
    if (j1 > im.h - 1) j1 -= im.h;
  } else {
    u = s * static_cast<float>(im.w);
    v = t * static_cast<float>(im.h);
    i0 = static_cast<int>(__builtin_floorf(u - 0.5f));
    j0 = static_cast<int>(__builtin_floorf(v - 0.5f));
    i1 = i0 + 1;
    j1 = j0 + 1;
  }
  const float a = (u - 0.5f) - __builtin_floorf(u - 0.5f);
  const float b = (v - 0.5f) - __builtin_floorf(v - 0.5f);
  const float4 t00 = texel(im, i0, j0, layer), t10 = texel(im, i1, j0, layer);
  const float4 t01 = texel(im, i0, j1, layer), t11 = texel(im, i1, j1, layer);
  const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  // ((T00 w00 + T10 w10) + T01 w01) + T11 w11 with the fusion of the numeric contract
  return { fma_(t11.x, w11, fma_(t01.x, w01, fma_(t00.x, w00, t10.x * w10))),
           fma_(t11.y, w11, fma_(t01.y, w01, fma_(t00.y, w00, t10.y * w10))),
           fma_(t11.z, w11, fma_(t01.z, w01, fma_(t00.z, w00, t10.z * w10))) };
}

NV_DEV V3 v3(const nova_float3& v) { return { v.x, v.y, v.z }; }

// kernels/transforms.hpp:32-40.  The reference compares fabs(n.y) with the double literal 0.1;
// no float lies between 0.1 and 0.1f, so `(double)f > 0.1` is `f >= 0.1f`.
NV_DEV M3 create_basis(V3 n) {
  V3 helper = (__builtin_fabsf(n.y) >= 0.1f) ? V3 { 1.f, 0.f, 0.f } : V3 { 0.f, 1.f, 0.f };
  V3 u = normalize(cross(n, helper));
  V3 w = cross(u, n);
  return { u, n, w };
}
NV_DEV V3 interp(V3 bc, V3 a, V3 b, V3 c) { return fma3(c, bc.z, fma3(a, bc.x, b * bc.y)); } // transforms.hpp:18-22
NV_DEV float nz(float x) { return max_(x, kNonZeroEps); }
NV_DEV V3 nz(V3 x) { return max3(x, splat(kNonZeroEps)); }
NV_DEV V3 reflect(V3 i, V3 n) { return fma3(n, -(2.0f * dot(n, i)), i); }
NV_DEV V3 spherical(float theta, float phi) {
  return { nova_sinf(theta) * nova_cosf(phi), nova_cosf(theta), nova_sinf(theta) * nova_sinf(phi) };
}

// kernels/material.hpp:15-29
NV_DEV V3 read_material(const ImageView& mats, const nova_triangle_meta& m, float tu, float tv, int index, V3 dflt) {
  if (m.diffuse_index == -1 && m.metallic_index == -1 && m.roughness_index == -1 && m.normal_index == -1) return dflt;
  if (index == -1) return dflt;
  return sample_linear<true>(mats, tu, tv, index);
}
// kernels/material.hpp:31-35
NV_DEV V3 read_sky(const ImageView& sky, V3 d) {
  float u = fma_(nova_atan2f(d.z, d.x), kInvPi * 0.5f, 0.5f);
  float v = fma_(nova_asinf(d.y), kInvPi, 0.5f);
  return sample_linear<true>(sky, u, v, 0);
}
// kernels/material.hpp:37-62
NV_DEV V3 shading_normal(const ImageView& mats, const nova_triangle_meta& m, float tu, float tv, V3 bc) {
  V3 n = normalize(interp(bc, v3(m.normal1), v3(m.normal2), v3(m.normal3)));
  if (m.normal_index != -1) {
    V3 t = normalize(interp(bc, v3(m.tangent1), v3(m.tangent2), v3(m.tangent3)));
    V3 b = normalize(interp(bc, v3(m.bitangent1), v3(m.bitangent2), v3(m.bitangent3)));
    M3 tbn = transpose({ t, b, n });
    V3 pn = read_material(mats, m, tu, tv, m.normal_index, { 0.f, 0.f, 0.f });
    pn = normalize(fma3(pn, 2.0f, splat(-1.0f)));
    n = normalize(mul(tbn, pn));
  }
  return n;
}

// kernels/light.hpp:12-44
NV_DEV V3 light_normal(const nova_area_light& l) {
  if (l.type == NOVA_LIGHT_RECT) return normalize(v3(l.rect.normal));
  return normalize(cross(v3(l.tri.v2) - v3(l.tri.v1), v3(l.tri.v3) - v3(l.tri.v1)));
}
NV_DEV float light_area(const nova_area_light& l) {
  if (l.type == NOVA_LIGHT_RECT) return l.rect.dims.x * l.rect.dims.y;
  return 0.5f * length(cross(v3(l.tri.v2) - v3(l.tri.v1), v3(l.tri.v3) - v3(l.tri.v1)));
}
NV_DEV V3 light_sample(const nova_area_light& l, uint32_t& rng) {
  if (l.type == NOVA_LIGHT_RECT) {
    float r0 = rnd(rng), r1 = rnd(rng);
    float ox = l.rect.dims.x * fma_(r0, 2.0f, -1.0f);
    float oy = l.rect.dims.y * fma_(r1, 2.0f, -1.0f);
    M3 basis = create_basis(normalize(v3(l.rect.normal)));
    return v3(l.rect.position) + mul(basis, { ox, 0.0f, oy });
  }
  float r1 = __builtin_sqrtf(rnd(rng));
  float r2 = rnd(rng);
  V3 b = (r1 * (1.0f - r2)) * v3(l.tri.v2);
  return fma3(v3(l.tri.v3), r1 * r2, fma3(v3(l.tri.v1), 1.0f - r1, b));
}

// Cook-Torrance GGX BRDF, kernels/brdf.hpp:20-203
struct Brdf {
  V3 in_dir, out_dir, normal, m_normal, diffuse;
  float metallic, roughness;
  float D, G, i_dot_m, i_dot_n, o_dot_n, n_dot_m;
  V3 F, k_d, t;

  NV_DEV static V3 f0_of(V3 dif, float met) { return fma3(dif - splat(0.04f), met, splat(0.04f)); }
  NV_DEV static V3 schlick(float c, V3 f0) {
    float p = nova_powf(max_(1.0f - c, 0.0f), 5.0f);
    return fma3(splat(1.0f) - f0, p, f0);
  }
  NV_DEV void init(V3 out, V3 n, V3 dif, float met, float rough) {
    out_dir = out;
    normal = n;
    diffuse = dif;
    metallic = met;
    roughness = rough;
  }
  NV_DEV void init_light(V3 in, V3 out, V3 n, V3 dif, float met, float rough) { // brdf.hpp:183-194
    init(out, n, dif, met, rough);
    in_dir = in;
    m_normal = normalize(in + out);
  }
  NV_DEV V3 sample(uint32_t& rng) { // brdf.hpp:41-77
    float xi0 = rnd(rng), xi1 = rnd(rng), xi2 = rnd(rng);
    V3 f0 = f0_of(diffuse, metallic);
    float odn = max_(dot(out_dir, normal), 0.0f);
    V3 ks = schlick(odn, f0);
    V3 kd = (splat(1.0f) - ks) * (1.0f - metallic);
    V3 ratio = clamp3(ks / (ks + kd), 0.0f, 1.0f);
    float est_t = max_(((ratio.x + ratio.y) + ratio.z) / 3.0f, 0.25f);
    M3 basis = transpose(create_basis(normal));
    float phi = (2.0f * kPi) * xi1;
    // (both members are assigned once, after the branch: assigned inside it, the compiler sinks the stores into one
    // store through a selected address and keeps four words of this struct in scratch memory)
    V3 mn, id;
    if (xi0 <= est_t) {
      float theta = nova_atanf(roughness * __builtin_sqrtf(xi2 / (1.0f - xi2)));
      mn = mul(basis, spherical(theta, phi));
      id = reflect(-out_dir, mn);
    } else {
      float theta = nova_acosf(__builtin_sqrtf(xi2));
      id = mul(basis, spherical(theta, phi));
      mn = normalize(id + out_dir);
    }
    m_normal = mn;
    in_dir = id;
    return in_dir;
  }
  NV_DEV void intermediates() { // brdf.hpp:134-159
    i_dot_m = dot(in_dir, m_normal);
    i_dot_n = dot(in_dir, normal);
    o_dot_n = max_(dot(out_dir, normal), 0.0f);
    n_dot_m = max_(dot(normal, m_normal), 0.0f);
    float alpha = roughness * roughness;
    float a2 = alpha * alpha, c2 = n_dot_m * n_dot_m;
    float x = fma_(c2, a2 - 1.0f, 1.0f);
    D = a2 / nz((kPi * x) * x);
    V3 f0 = f0_of(diffuse, metallic);
    F = schlick(i_dot_m, f0);
    float k = roughness + 1.0f;
    k = k * k / 8.0f;
    float g_i = i_dot_n / nz(fma_(i_dot_n, 1.0f - k, k));
    float g_o = o_dot_n / nz(fma_(o_dot_n, 1.0f - k, k));
    G = g_i * g_o;
    k_d = (splat(1.0f) - F) * (1.0f - metallic);
    t = clamp3(F / (F + k_d), 0.0f, 1.0f);
  }
  NV_DEV V3 eval() const { // brdf.hpp:85-96 (after intermediates())
    if (i_dot_m <= 0.0f || i_dot_n <= 0.0f) return { 0.f, 0.f, 0.f };
    V3 lambert = diffuse * kInvPi;
    V3 ct = ((D * F) * G) / nz((4.0f * o_dot_n) * i_dot_n);
    return fma3(k_d, lambert, ct);
  }
  NV_DEV V3 pdf() const { // brdf.hpp:103-107
    V3 spec = ((t * D) * n_dot_m) / nz(4.0f * i_dot_m);
    return nz(fma3((splat(1.0f) - t) * i_dot_n, kInvPi, spec));
  }
  NV_DEV float light_pdf(V3 ln, float dist, float area) const { // brdf.hpp:200-202
    return dist * dist / nz(__builtin_fabsf(dot(ln, in_dir)) * area);
  }
};
NV_DEV V3 power_heuristic(V3 p, V3 s) {
  V3 p2 = p * p, s2 = s * s;
  return p2 / (p2 + s2);
}

// path index -> pixel: 64x64 image tiles dealt round-robin over ranks; inside a tile 8x8 blocks,
// so that the 64 lanes of a wave start as one coherent 8x8 pixel block.
NV_DEV bool path_pixel(const FrameArgs& F, uint32_t p, uint32_t& x, uint32_t& y) {
  if (p >= F.n_my_tiles * 4096u) {
    const uint32_t pixel = F.extra_pixels[p - F.n_my_tiles * 4096u];
    y = pixel / F.width;
    x = pixel - y * F.width;
    return true;
  }
  const uint32_t t = p >> 12, q = p & 4095u;
  const uint32_t tile = F.tile_rank + t * F.tile_ranks;
  const uint32_t tx = tile % F.tiles_x, ty = tile / F.tiles_x;
  const uint32_t blk = q >> 6, l = q & 63u;
  x = tx * 64u + (blk & 7u) * 8u + (l & 7u);
  y = ty * 64u + (blk >> 3) * 8u + (l >> 3);
  return x < F.width && y < F.height;
}

// The triangles that carry a light index, for the pre-test: list[0] counts them all, the first kMaxLightTris are kept.
__global__ __launch_bounds__(256) void k_find_light_tris(const nova_triangle_meta* meta, const float4* tris, uint32_t n_tris,
                                                         uint32_t* list) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tris) return;
  const int light = meta[i].light_index;
  if (light == -1) return;
  const uint32_t k = atomicAdd(&list[0], 1u);
  if (k < kMaxLightTris) {
    uint32_t* rec = list + 16 + 16 * k;
    rec[0] = i;
    rec[1] = static_cast<uint32_t>(light);
    float4* m = reinterpret_cast<float4*>(rec + 4);
    m[0] = tris[3 * static_cast<size_t>(i)];
    m[1] = tris[3 * static_cast<size_t>(i) + 1];
    m[2] = tris[3 * static_cast<size_t>(i) + 2];
  }
}

// Stage 1: camera rays.  kernels/kernels.cu:23-27 (seed) and kernels/raytrace.hpp:117-127.
__global__ __launch_bounds__(256) void k_generate(FrameArgs F, PathPlanes S, float4* ray_o, float4* ray_d) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = F.n_my_tiles * 4096u + F.n_extra;
  if (p == 0) F.n_active[0] = total;
  if (p >= total) return;
  uint32_t x, y;
  if (!path_pixel(F, p, x, y)) {
    S.b[p] = make_float4(0.f, 0.f, 0.f, __uint_as_float(kEmptyPixel));
    ray_o[p] = make_float4(0.f, 0.f, 0.f, -1.0f);
    ray_d[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    const uint32_t pixel = y * F.width + x;
    uint32_t rng = wang_hash(pixel + wang_hash(F.time));
    const float jx = rnd(rng), jy = rnd(rng);
    const nova_eye_coords& E = F.params.eye_coords;
    const float ax = E.coord_scale.x * ((static_cast<float>(static_cast<int>(x)) - E.coord_dims.x) + jx);
    const float ay = E.coord_scale.y * ((static_cast<float>(static_cast<int>(y)) - E.coord_dims.y) + jy);
    const M3 frame_t = transpose({ v3(E.eye_coord_frame.x), v3(E.eye_coord_frame.y), v3(E.eye_coord_frame.z) });
    const V3 dir = normalize(mul(frame_t, { ax, -ay, -1.0f }));
    S.a[p] = make_float4(0.f, 0.f, 0.f, __uint_as_float(rng));
    S.b[p] = make_float4(1.f, 1.f, 1.f, __uint_as_float(pixel));
    S.c[p] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
    ray_o[p] = make_float4(E.eye_pos.x, E.eye_pos.y, E.eye_pos.z, kFltMax);
    ray_d[p] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(0u));
  }
}

// Stage 3: resolve the previous segment's direct lighting, shade the new hit, emit rays.
__global__ __launch_bounds__(kShadeBlock) void k_shade(FrameArgs F, uint32_t it, PathPlanes I, PathPlanes O,
                                               const float4* ray_o_in, const float4* ray_d_in,
                                               const float4* hits, float4* ray_o_out, float4* ray_d_out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_in = F.n_active[it];
  const uint32_t P = F.stride;
  __shared__ float s_u8_lut[256];
  if (F.materials.u8) { // uniform over the launch
    if (threadIdx.x < 256) s_u8_lut[threadIdx.x] = static_cast<float>(threadIdx.x) / 255.0f;
    __syncthreads();
    F.materials.lut = s_u8_lut;
  }
  bool alive = false;
  bool emit_shadow = false, emit_mis = false, emit_cont = false, shaded = false, shadow_live = false;

  // values carried to the append step
  V3 color { 0.f, 0.f, 0.f }, thr { 1.f, 1.f, 1.f }, thr_before { 0.f, 0.f, 0.f };
  V3 nee_term { 0.f, 0.f, 0.f }, mis_f { 0.f, 0.f, 0.f }, mis_pdf { 0.f, 0.f, 0.f };
  V3 point { 0.f, 0.f, 0.f }, shadow_dir { 0.f, 0.f, 0.f }, mis_dir { 0.f, 0.f, 0.f }, cont_dir { 0.f, 0.f, 0.f };
  float shadow_tmax = -1.0f, mis_ndl = 0.0f, mis_tmax = -1.0f;
  int here_light = -1;
  uint32_t rng = 0, pixel = kEmptyPixel, out_flags = 0;

  if (j < n_in) {
    const float4 sa = I.a[j], sb = I.b[j];
    pixel = __float_as_uint(sb.w);
    if (pixel != kEmptyPixel) {
      color = xyz(sa);
      rng = __float_as_uint(sa.w);
      thr = xyz(sb);
      const float4 sc = I.c[j];
      const uint32_t flags = __float_as_uint(sc.w);
      bool finished = false;

      if (flags & kFlagPending) {
        // direct lighting of the previous segment: kernels/raytrace.hpp:219-231 and :242-263
        const float4 sd = I.d[j], se = I.e[j], sf = I.f[j];
        V3 dc { 0.f, 0.f, 0.f };
        if (flags & kFlagHasShadow) {
          const float4 sh = hits[static_cast<size_t>(P) + j];
          if (__float_as_int(sh.w) < 0) dc = xyz(sd);
        }
        const size_t mis_slot = 2 * static_cast<size_t>(P) + __float_as_uint(sf.w);
        float4 mh = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        if (flags & kFlagHasMis) mh = hits[mis_slot];
        const int mis_tri = __float_as_int(mh.w);
        if (mis_tri >= 0) {
          const nova_triangle_meta& lm = F.meta[mis_tri];
          const int prev_light = __float_as_int(se.w);
          if (lm.light_index != -1 && lm.light_index != prev_light) {
            const nova_area_light& L = F.lights[lm.light_index];
            const V3 ln = light_normal(L);
            const float la = light_area(L);
            const V3 ldir = xyz(ray_d_in[mis_slot]);
            const float lpdf = mh.x * mh.x / nz(__builtin_fabsf(dot(ln, ldir)) * la);
            const V3 f = xyz(se), bpdf = xyz(sf);
            const V3 w = power_heuristic(bpdf, splat(lpdf));
            const V3 c = ((w * v3(L.intensity)) * f) / bpdf;
            dc = fma3(c, sd.w, dc);
          }
        }
        color = fma3(xyz(sc), dc, color);
      }
      if (flags & kFlagTerminate) finished = true;

      if (!finished) {
        const float4 hit = hits[j];
        const int tri = __float_as_int(hit.w);
        const V3 rd = xyz(ray_d_in[j]);
        const bool direct = (it == 0);
        if (tri < 0) {
          // kernels/raytrace.hpp:137-147
          if (direct) {
            color = read_sky(F.sky, rd);
            if (F.denoise) {
              const V3 al = color / (color + splat(1.0f));
              F.out_albedo[pixel] = make_float4(al.x, al.y, al.z, 1.0f);
              F.out_normal[pixel] = make_float4(0.f, 0.f, 0.f, 1.0f);
            }
          }
          finished = true;
        } else {
          shaded = true;
          const nova_triangle_meta& m = F.meta[tri];
          const V3 ro = fma3(rd, kRayEpsilon, xyz(ray_o_in[j])); // the Ray's offset origin
          point = fma3(rd, hit.x, ro);
          const V3 bc = { (1.0f - hit.y) - hit.z, hit.y, hit.z };
          const float tu = fma_(m.texture_coord3.x, bc.z, fma_(m.texture_coord1.x, bc.x, m.texture_coord2.x * bc.y));
          const float tv = fma_(m.texture_coord3.y, bc.z, fma_(m.texture_coord1.y, bc.x, m.texture_coord2.y * bc.y));
          const V3 diffuse = read_material(F.materials, m, tu, tv, m.diffuse_index, v3(F.params.shading_diffuse)) * v3(m.kD);
          const float metallic = read_material(F.materials, m, tu, tv, m.metallic_index,
                                               splat(m.metallic == -1.0f ? F.params.shading_metallic : m.metallic)).x;
          const float roughness = read_material(F.materials, m, tu, tv, m.roughness_index,
                                                splat(m.roughness == -1.0f ? F.params.shading_roughness : m.roughness)).x;
          here_light = m.light_index;
          if (!F.params.path_tracing) {
            // kernels/raytrace.hpp:170-177
            color = color + (here_light != -1 ? v3(F.lights[here_light].intensity) : diffuse);
            if (F.denoise) {
              // the reference leaves both features uninitialised on this path (kernels/kernels.cu:29-38
              // writes the locals that trace_ray never set); defined here as zero, like the oracle
              F.out_albedo[pixel] = make_float4(0.f, 0.f, 0.f, 1.0f);
              F.out_normal[pixel] = make_float4(0.f, 0.f, 0.f, 1.0f);
            }
            finished = true;
          } else {
            const V3 normal = shading_normal(F.materials, m, tu, tv, bc);
            const V3 out_dir = -rd;
            if (direct && F.denoise) {
              F.out_albedo[pixel] = make_float4(diffuse.x, diffuse.y, diffuse.z, 1.0f);
              F.out_normal[pixel] = make_float4(normal.x, normal.y, normal.z, 1.0f);
            }
            if (direct && here_light != -1) color = fma3(thr, v3(F.lights[here_light].intensity), color);

            Brdf b;
            if (F.num_lights > 1 || (F.num_lights == 1 && here_light == -1)) {
              // light sample (raytrace.hpp:198-231); the shadow ray is traced by the next pass
              uint32_t li;
              do {
                li = min(static_cast<uint32_t>(rnd(rng) * static_cast<float>(F.num_lights)), F.num_lights - 1u);
              } while (static_cast<uint32_t>(here_light) == li);
              const nova_area_light& L = F.lights[li];
              const V3 ln = light_normal(L);
              const float la = light_area(L);
              const V3 lp = light_sample(L, rng);
              shadow_dir = normalize(lp - point);
              const float ldist = length(lp - point);
              shadow_tmax = ldist - kRayEpsilon;
              b.init_light(shadow_dir, out_dir, normal, diffuse, metallic, roughness);
              b.intermediates();
              const V3 f = b.eval();
              const float lpdf = b.light_pdf(ln, ldist, la);
              const V3 bpdf = b.pdf();
              const V3 w = power_heuristic(splat(lpdf), bpdf);
              const V3 c = (((static_cast<float>(F.num_lights) * w) * v3(L.intensity)) * f) / lpdf;
              nee_term = c * max_(dot(normal, shadow_dir), 0.0f);
              emit_shadow = true;
              // a term of +0 in every channel (the light is below this surface's horizon: eval() is zero) is what the
              // resolve adds whether the shadow ray is blocked or not, so that ray is not traced (same option)
              shadow_live = !(F.light_tris && (__float_as_uint(nee_term.x) | __float_as_uint(nee_term.y) |
                                               __float_as_uint(nee_term.z)) == 0u);
              // BRDF sample for MIS (raytrace.hpp:234-263); resolved by the next pass
              b.init(out_dir, normal, diffuse, metallic, roughness);
              mis_dir = b.sample(rng);
              b.intermediates();
              mis_f = b.eval();
              mis_pdf = b.pdf();
              mis_ndl = max_(dot(normal, mis_dir), 0.0f);
              emit_mis = true;
              mis_tmax = kFltMax;
              if (F.light_tris) { // uniform; the list and the matrices come through the scalar cache
                const uint32_t n_lt = F.light_tris[0];
                if (n_lt <= kMaxLightTris) {
                  RayState lr;
                  lr.d = mis_dir;
                  lr.o = fma3(mis_dir, kRayEpsilon, point); // ray_setup
                  float far = -1.0f;
                  for (uint32_t k = 0; k < n_lt; k++) {
                    const uint32_t* rec = F.light_tris + 16 + 16 * k;
                    if (static_cast<int>(rec[1]) == here_light) continue;
                    const float4* mx = reinterpret_cast<const float4*>(rec + 4);
                    HitState lh { kFltMax, 0.0f, 0.0f, -1 };
                    if (woop(lr, lh, static_cast<int>(rec[0]), mx[0], mx[1], mx[2])) far = max_(far, lh.t);
                  }
                  mis_tmax = far < 0.0f ? -1.0f : __uint_as_float(__float_as_uint(max_(far, 1e-30f)) + 1u);
                }
              }
            }
            thr_before = thr;

            const bool capped = (F.max_depth != 0u && it + 1u >= F.max_depth);
            bool cont = !capped;
            if (cont) {
              // kernels/raytrace.hpp:269-289
              b.init(out_dir, normal, diffuse, metallic, roughness);
              cont_dir = b.sample(rng);
              b.intermediates();
              const V3 f = b.eval();
              const V3 pdf = b.pdf();
              thr = thr * ((f / pdf) * max_(dot(normal, cont_dir), 0.0f));
              if (!direct) {
                const float pr = min_(max_(thr.x, max_(thr.y, thr.z)), 1.0f);
                if (rnd(rng) > pr) {
                  cont = false;
                } else {
                  thr = thr * (1.0f / pr);
                }
              }
            }
            emit_cont = cont;
            if (emit_shadow || emit_mis) {
              out_flags = kFlagPending | (shadow_live ? kFlagHasShadow : 0u) | (cont ? 0u : kFlagTerminate) |
                          (mis_tmax >= 0.0f ? kFlagHasMis : 0u);
              alive = true;
            } else if (cont) {
              out_flags = 0u;
              alive = true;
            } else {
              finished = true;
            }
          }
        }
      }
      if (finished) F.out_color[pixel] = make_float4(color.x, color.y, color.z, 1.0f);
    }
  }

  // compaction: one atomic per block (same-address atomics serialise at ~12 ns each, so one per
  // wave would cost more than the shading itself); ray/segment counts go to per-block slots
  // with plain stores, summed on the host when counters are read.
  __shared__ uint32_t s_alive[kShadeBlock / 64];
  __shared__ uint32_t s_count[6][kShadeBlock / 64]; // [4] = MIS rays that are traced, [5] = shadow rays that are not
  __shared__ uint32_t s_base, s_base_mis;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const bool mis_live = alive && emit_mis && mis_tmax >= 0.0f;
  const uint64_t mask = __ballot(alive);
  const uint64_t m_cont = __ballot(alive && emit_cont), m_mis = __ballot(alive && emit_mis),
                 m_shadow = __ballot(alive && emit_shadow), m_shaded = __ballot(shaded), m_live = __ballot(mis_live),
                 m_dark = __ballot(alive && emit_shadow && !shadow_live);
  if (lane == 0) {
    s_alive[wave] = static_cast<uint32_t>(__popcll(mask));
    s_count[0][wave] = static_cast<uint32_t>(__popcll(m_cont));
    s_count[1][wave] = static_cast<uint32_t>(__popcll(m_mis));
    s_count[2][wave] = static_cast<uint32_t>(__popcll(m_shadow));
    s_count[3][wave] = static_cast<uint32_t>(__popcll(m_shaded));
    s_count[4][wave] = static_cast<uint32_t>(__popcll(m_live));
    s_count[5][wave] = static_cast<uint32_t>(__popcll(m_dark));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0, c[6] = { 0, 0, 0, 0, 0, 0 };
    for (uint32_t w = 0; w < kShadeBlock / 64; w++) {
      total += s_alive[w];
      for (int q = 0; q < 6; q++) c[q] += s_count[q][w];
    }
    s_base = total ? atomicAdd(&F.n_active[it + 1], total) : 0u;
    s_base_mis = c[4] ? atomicAdd(&F.n_mis[it + 1], c[4]) : 0u;
    unsigned long long* bc = F.block_counters + kBlockCounters * static_cast<size_t>(blockIdx.x);
    for (int q = 0; q < 4; q++)
      if (c[q]) bc[q] += c[q];
    if (c[1] != c[4]) bc[4] += c[1] - c[4];
    if (c[5]) bc[5] += c[5];
  }
  __syncthreads();
  if (alive) {
    uint32_t s = s_base + rank_in_mask(mask);
    uint32_t ms = s_base_mis + rank_in_mask(m_live);
    for (uint32_t w = 0; w < wave; w++) {
      s += s_alive[w];
      ms += s_count[4][w];
    }
    O.a[s] = make_float4(color.x, color.y, color.z, __uint_as_float(rng));
    O.b[s] = make_float4(thr.x, thr.y, thr.z, __uint_as_float(pixel));
    O.c[s] = make_float4(thr_before.x, thr_before.y, thr_before.z, __uint_as_float(out_flags));
    if (out_flags & kFlagPending) {
      O.d[s] = make_float4(nee_term.x, nee_term.y, nee_term.z, mis_ndl);
      O.e[s] = make_float4(mis_f.x, mis_f.y, mis_f.z, __int_as_float(here_light));
      O.f[s] = make_float4(mis_pdf.x, mis_pdf.y, mis_pdf.z, __uint_as_float(ms));
    }
    ray_o_out[s] = make_float4(point.x, point.y, point.z, emit_cont ? kFltMax : -1.0f);
    ray_d_out[s] = make_float4(cont_dir.x, cont_dir.y, cont_dir.z, __uint_as_float(0u));
    if (F.num_lights > 0) {
      const size_t s1 = static_cast<size_t>(P) + s;
      ray_o_out[s1] = make_float4(point.x, point.y, point.z, shadow_live ? shadow_tmax : -1.0f);
      ray_d_out[s1] = make_float4(shadow_dir.x, shadow_dir.y, shadow_dir.z, __uint_as_float(1u));
    }
    if (mis_live) {
      const size_t s2 = 2 * static_cast<size_t>(P) + ms;
      ray_o_out[s2] = make_float4(point.x, point.y, point.z, mis_tmax);
      ray_d_out[s2] = make_float4(mis_dir.x, mis_dir.y, mis_dir.z, __uint_as_float(0u));
    }
  }
}

}
