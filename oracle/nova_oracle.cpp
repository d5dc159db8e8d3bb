// TEST INFRASTRUCTURE ONLY — never linked into, imported by or shipped with the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// CPU restatement of Nova's ray-tracing hot path (scalar C++17, OpenMP over pixel rows).
// Every function cites the reference file:line it follows (paths relative to the Nova tree).
// It takes the same plain-C structs as the product's C ABI (include/nova_hip.h).
//
// PARITY UNPINNED.  The reference's own tests hold no vectors for this path (tests/src/intersectables/
// covers the AABB and Triangle host helpers only; those known answers are reproduced in
// tests/test_reference_kats.py, and nova::AABB itself — the one piece that builds here from the
// reference's own files — is compared bit for bit through oracle/_ref).  The kernels need the
// OpenCL C / CUDA device toolchains, which this image lacks, so no output of the reference exists
// to pin them: this file is a restatement checked by reading, by independent geometry
// (tests/test_oracle_properties.py: brute-force double-precision closest hits, energy and
// reciprocity properties) and frozen by the regression vectors in tests/golden/ (DESIGN.md §6).
//
// FLOATING-POINT CONTRACT: IEEE binary32, one rounding
// per operation, a*b+c fused exactly where the reference's source has it in one expression
// (written here as explicit fmaf; this file is compiled with -ffp-contract=off), dot = fma chain
// with the last component first, normalize = v * (1/sqrt(v.v)), min/max = fminf/fmaxf,
// sin/cos/atan/atan2/asin/acos/exp/pow = include/nova_math.h (bit-reproducible on host and device).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <omp.h>
#include <vector>

#include "nova_hip.h"
#include "nova_math.h"

namespace {

constexpr float kRayEpsilon = 1e-2f;        // kernels/constants.hpp:10
constexpr float kPi = 3.14159274101257f;
constexpr float kInvPi = 0.31830987334251f;
constexpr float kNonZeroEps = 1e-7f;        // kernels/transforms.hpp:50

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

inline f3 operator+(f3 a, f3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline f3 operator-(f3 a, f3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline f3 operator*(f3 a, f3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline f3 operator/(f3 a, f3 b) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
inline f3 operator*(f3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline f3 operator*(float s, f3 a) { return { s * a.x, s * a.y, s * a.z }; }
inline f3 operator/(f3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline f3 operator-(f3 a) { return { -a.x, -a.y, -a.z }; }
inline f3 splat(float s) { return { s, s, s }; }
inline f3 fma3(f3 a, f3 b, f3 c) { return { fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z) }; }
inline f3 fma3(f3 a, float b, f3 c) { return { fmaf(a.x, b, c.x), fmaf(a.y, b, c.y), fmaf(a.z, b, c.z) }; }
inline f3 max3(f3 a, f3 b) { return { fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z) }; }
inline f3 clamp3(f3 a, float lo, float hi) {
  return { fminf(fmaxf(a.x, lo), hi), fminf(fmaxf(a.y, lo), hi), fminf(fmaxf(a.z, lo), hi) };
}
inline float dot(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
inline f3 cross(f3 a, f3 b) {
  return { fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)) };
}
inline float length(f3 a) { return sqrtf(dot(a, a)); }
inline f3 normalize(f3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
inline f3 v3(const nova_float3& v) { return { v.x, v.y, v.z }; }
inline f3 xyz(const nova_float4& v) { return { v.x, v.y, v.z }; }

struct Mat3 { f3 x, y, z; };
inline f3 mul(const Mat3& m, f3 v) { return { dot(m.x, v), dot(m.y, v), dot(m.z, v) }; } // kernels/matrix.hpp:18-20
inline Mat3 transpose(const Mat3& m) {                                                    // matrix.hpp:22-26
  return { { m.x.x, m.y.x, m.z.x }, { m.x.y, m.y.y, m.z.y }, { m.x.z, m.y.z, m.z.z } };
}

// kernels/random.hpp:11-18
inline uint32_t wang_hash(uint32_t s) {
  s = (s ^ 61u) ^ (s >> 16);
  s *= 9u;
  s ^= s >> 4;
  s *= 0x27d4eb2du;
  s ^= s >> 15;
  return s;
}
// kernels/random.hpp:21-26 — UINT_MAX converts to 2^32, so the result lies in [0, 1]
inline float rnd(uint32_t& st) {
  st ^= st << 13;
  st ^= st >> 17;
  st ^= st << 5;
  return static_cast<float>(st) / 4294967296.0f;
}

// kernels/types.hpp:10-21
struct Ray {
  f3 origin, direction, inv_direction, nio;
  Ray(f3 point, f3 dir, float eps) {
    origin = fma3(dir, eps, point);
    direction = dir;
    inv_direction = { 1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z };
    nio = (-origin) * inv_direction;
  }
};
// kernels/types.hpp:23-31
struct Hit {
  f3 bary { 0, 0, 0 };
  float length = FLT_MAX;
  int tri = -1;
};

// kernels/intersection.hpp:40-52; also returns the slab interval for the ordered traversal
inline bool slab(const Ray& r, f3 top, f3 bottom, float& tmin, float& tmax) {
  f3 t1 = fma3(top, r.inv_direction, r.nio);
  f3 t2 = fma3(bottom, r.inv_direction, r.nio);
  f3 lo = { fminf(t1.x, t2.x), fminf(t1.y, t2.y), fminf(t1.z, t2.z) };
  f3 hi = { fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y), fmaxf(t1.z, t2.z) };
  tmin = fmaxf(lo.x, fmaxf(lo.y, lo.z));
  tmax = fminf(hi.x, fminf(hi.y, hi.z));
  return tmin <= tmax;
}

// kernels/intersection.hpp:14-37 (Woop unit-triangle test).  `tie_on_index` adds the explicit
// "equal t -> lower index wins" rule an out-of-order traversal needs to reproduce the
// reference, which visits triangles in increasing index order with a strict t < length test.
inline bool hit_triangle(const Ray& r, Hit& h, int index, const nova_triangle_data& m, bool tie_on_index) {
  const f3 o = r.origin, d = r.direction;
  float oz = fmaf(m.z.x, o.x, fmaf(m.z.y, o.y, fmaf(m.z.z, o.z, m.z.w)));
  float dz = fmaf(m.z.x, d.x, fmaf(m.z.y, d.y, m.z.z * d.z));
  float t = -oz / dz;
  if (t < 0.0f || !std::isfinite(t)) return false;
  if (t >= h.length) {
    if (!(tie_on_index && t == h.length && h.tri >= 0 && index < h.tri)) return false;
  }
  float ox = fmaf(m.x.x, o.x, fmaf(m.x.y, o.y, fmaf(m.x.z, o.z, m.x.w)));
  float oy = fmaf(m.y.x, o.x, fmaf(m.y.y, o.y, fmaf(m.y.z, o.z, m.y.w)));
  float dx = fmaf(m.x.x, d.x, fmaf(m.x.y, d.y, m.x.z * d.z));
  float dy = fmaf(m.y.x, d.x, fmaf(m.y.y, d.y, m.y.z * d.z));
  float u = fmaf(t, dx, ox);
  float v = fmaf(t, dy, oy);
  float w = (1.0f - u) - v;
  if (w < 0.0f || u < 0.0f || v < 0.0f) return false;
  h.length = t;
  h.bary = { w, u, v };
  h.tri = index;
  return true;
}

struct WorkCount {
  uint64_t nodes = 0, tris = 0;
};

// kernels/raytrace.hpp:38-102 — the reference walk: fixed order (left child first), no culling
// by distance, triangles of up to two pending leaves tested together, any-hit returns at once.
bool find_intersection_reference(const nova_triangle_data* tris, const nova_bvh_node* bvh, const Ray& ray,
                                 Hit& best, bool anyhit, WorkCount* wc) {
  std::vector<uint32_t> stack;
  stack.reserve(64);
  stack.push_back(0); // sentinel: traversal ends when it is popped
  uint32_t node_index = 0;
  do {
    uint32_t leaf_off[2] = { 0, 0 }, leaf_num[2] = { 0, 0 };
    do {
      const nova_bvh_node& n = bvh[node_index];
      if (wc) wc->nodes++;
      float a, b;
      if (!slab(ray, xyz(n.top_offset_left), xyz(n.bottom_num_right), a, b)) {
        node_index = stack.back();
        stack.pop_back();
        continue;
      }
      if (n.bottom_num_right.w >= 0.0f) {
        stack.push_back(static_cast<uint32_t>(n.bottom_num_right.w));
        node_index++;
      } else {
        uint32_t off = static_cast<uint32_t>(n.top_offset_left.w);
        uint32_t num = static_cast<uint32_t>(-n.bottom_num_right.w);
        node_index = stack.back();
        stack.pop_back();
        if (leaf_num[0] == 0) {
          leaf_off[0] = off;
          leaf_num[0] = num;
        } else {
          leaf_off[1] = off;
          leaf_num[1] = num;
          break;
        }
      }
    } while (node_index);
    for (int j = 0; j < 2; j++) {
      for (uint32_t i = leaf_off[j]; i < leaf_off[j] + leaf_num[j]; i++) {
        if (wc) wc->tris++;
        if (hit_triangle(ray, best, static_cast<int>(i), tris[i], false) && anyhit) return true;
      }
    }
  } while (node_index);
  return best.tri != -1;
}

// Cull slack of every subtree, for the ordered traversal below.  Nova's Woop matrices are
// ill-conditioned where n.v1 ~ 1 (src/intersectables/intersectable_manager.cpp:137-142: the third
// column of the inverted matrix is n - v1), so the float evaluation of a triangle test can accept
// points outside the triangle's box; an exact distance cull would then skip hits the reference
// finds.  slack(triangle) = how far the triangle the matrix really encodes (double-precision
// inverse) sticks out of its leaf box + the world-space effect of the barycentric rounding
// (<= 2^-20 (3R sum|m_i| + |m_w|) per coordinate, R = scene radius); slack(node) = max over its
// subtree + 2^-19 R for the slab test's own rounding.
struct Slack {
  std::vector<float> node;
  float inv_radius = 1.0f;
};

double triangle_slack(const nova_triangle_data& m, const nova_bvh_node& leaf, double radius) {
  const double a[3][3] = { { m.x.x, m.x.y, m.x.z }, { m.y.x, m.y.y, m.y.z }, { m.z.x, m.z.y, m.z.z } };
  const double w[3] = { m.x.w, m.y.w, m.z.w };
  double c[3][3];
  for (int r = 0; r < 3; r++)
    for (int k = 0; k < 3; k++) {
      const int r1 = (r + 1) % 3, r2 = (r + 2) % 3, k1 = (k + 1) % 3, k2 = (k + 2) % 3;
      c[r][k] = a[r1][k1] * a[r2][k2] - a[r1][k2] * a[r2][k1];
    }
  const double det = a[0][0] * c[0][0] + a[0][1] * c[0][1] + a[0][2] * c[0][2];
  if (!(det != 0.0) || !std::isfinite(det)) return HUGE_VAL;
  double e1[3], e2[3], v1[3];
  for (int r = 0; r < 3; r++) {
    e1[r] = c[0][r] / det;
    e2[r] = c[1][r] / det;
    v1[r] = -(c[0][r] * w[0] + c[1][r] * w[1] + c[2][r] * w[2]) / det;
  }
  const double top[3] = { leaf.top_offset_left.x, leaf.top_offset_left.y, leaf.top_offset_left.z };
  const double bot[3] = { leaf.bottom_num_right.x, leaf.bottom_num_right.y, leaf.bottom_num_right.z };
  double stick = 0.0;
  for (int r = 0; r < 3; r++) {
    const double p[3] = { v1[r], v1[r] + e1[r], v1[r] + e2[r] };
    for (double q : p) stick = std::max({ stick, q - top[r], bot[r] - q });
  }
  const double le1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
  const double le2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
  const double k = 0x1p-20, r3 = 3.0 * radius;
  const double du = k * (r3 * (std::fabs(a[0][0]) + std::fabs(a[0][1]) + std::fabs(a[0][2])) + std::fabs(w[0]));
  const double dv = k * (r3 * (std::fabs(a[1][0]) + std::fabs(a[1][1]) + std::fabs(a[1][2])) + std::fabs(w[1]));
  const double s = stick + du * le1 + dv * le2;
  return s == s ? s : HUGE_VAL;
}

double subtree_slack(const nova_triangle_data* tris, const nova_bvh_node* bvh, uint32_t i, double radius,
                     std::vector<float>& out) {
  if (out.size() <= i) out.resize(i + 1, 0.0f);
  const nova_bvh_node& n = bvh[i];
  double s = 0.0;
  if (n.bottom_num_right.w < 0.0f) {
    const uint32_t off = static_cast<uint32_t>(n.top_offset_left.w), num = static_cast<uint32_t>(-n.bottom_num_right.w);
    for (uint32_t t = off; t < off + num; t++) s = std::max(s, triangle_slack(tris[t], n, radius));
  } else {
    s = std::max(subtree_slack(tris, bvh, i + 1, radius, out),
                 subtree_slack(tris, bvh, static_cast<uint32_t>(n.bottom_num_right.w), radius, out));
  }
  const double total = s * 1.0000002 + 0x1p-19 * radius;
  out[i] = total < 3.0e38 ? static_cast<float>(total) : HUGE_VALF;
  return s;
}

Slack compute_slack(const nova_triangle_data* tris, const nova_bvh_node* bvh) {
  Slack sl;
  double radius = 0.0;
  const float rc[6] = { bvh[0].top_offset_left.x, bvh[0].top_offset_left.y, bvh[0].top_offset_left.z,
                        bvh[0].bottom_num_right.x, bvh[0].bottom_num_right.y, bvh[0].bottom_num_right.z };
  for (float c : rc) radius = std::max(radius, static_cast<double>(std::fabs(c)));
  if (!(radius > 0.0) || !std::isfinite(radius)) radius = 1.0;
  sl.inv_radius = static_cast<float>(1.0 / radius);
  subtree_slack(tris, bvh, 0, radius, sl.node);
  return sl;
}

// Ordered, hit-distance-culled traversal of the same tree (near child first; a subtree is
// skipped when its box, widened by the subtree's slack, lies wholly behind the origin or beyond
// the current hit).  Returns the same hit as the reference walk (min t, ties -> lowest index);
// used for the algorithmic work model N_node / N_tri of SURVEY.md §8(d).  One "node" = one box
// test, as in the reference.
bool find_intersection_ordered(const nova_triangle_data* tris, const nova_bvh_node* bvh, const Ray& ray,
                               Hit& best, bool anyhit, WorkCount* wc, const Slack& slack) {
  struct Entry { uint32_t node; float tmin; };
  std::vector<Entry> stack;
  stack.reserve(64);
  const float o_inf = fmaxf(fabsf(ray.origin.x), fmaxf(fabsf(ray.origin.y), fabsf(ray.origin.z)));
  const float scale = fmaxf(fabsf(ray.inv_direction.x), fmaxf(fabsf(ray.inv_direction.y), fabsf(ray.inv_direction.z))) *
                      fmaxf(1.0f, o_inf * slack.inv_radius);
  auto limit = [&]() { return fmaf(fabsf(best.length), 0x1p-20f, best.length); };
  float a, b;
  if (wc) wc->nodes++;
  if (!slab(ray, xyz(bvh[0].top_offset_left), xyz(bvh[0].bottom_num_right), a, b)) return false;
  stack.push_back({ 0u, -HUGE_VALF });
  while (!stack.empty()) {
    Entry e = stack.back();
    stack.pop_back();
    if (e.tmin > limit()) continue;
    const nova_bvh_node& n = bvh[e.node];
    if (n.bottom_num_right.w < 0.0f) {
      uint32_t off = static_cast<uint32_t>(n.top_offset_left.w);
      uint32_t num = static_cast<uint32_t>(-n.bottom_num_right.w);
      for (uint32_t i = off; i < off + num; i++) {
        if (wc) wc->tris++;
        if (hit_triangle(ray, best, static_cast<int>(i), tris[i], true) && anyhit) return true;
      }
      continue;
    }
    uint32_t c[2] = { e.node + 1, static_cast<uint32_t>(n.bottom_num_right.w) };
    float tm[2], tx[2];
    bool ok[2];
    for (int k = 0; k < 2; k++) {
      if (wc) wc->nodes++;
      ok[k] = slab(ray, xyz(bvh[c[k]].top_offset_left), xyz(bvh[c[k]].bottom_num_right), tm[k], tx[k]);
      const float es = slack.node[c[k]] * scale;
      tm[k] = tm[k] - es; // entry into the widened box
      ok[k] = ok[k] && tx[k] + es >= 0.0f && tm[k] <= limit();
    }
    if (ok[0] && ok[1]) {
      int near = tm[1] < tm[0] ? 1 : 0;
      stack.push_back({ c[1 - near], tm[1 - near] });
      stack.push_back({ c[near], tm[near] });
    } else if (ok[0]) {
      stack.push_back({ c[0], tm[0] });
    } else if (ok[1]) {
      stack.push_back({ c[1], tm[1] });
    }
  }
  return best.tri != -1;
}

// ---- images ------------------------------------------------------------------------------------
// What CLAMP means beyond the image: 0 = the OpenCL backend's border colour 0 (CLK_ADDRESS_CLAMP,
// kernels/backend/opencl/image.hpp:14-21), 1 = the CUDA backend's clamp to the edge texel
// (src/backend/cuda/types/flags.hpp:30).  oracle_set_clamp_to_edge.
int g_clamp_to_edge = 0;

struct Image {
  const float* texels; // rgba
  int w, h, layers;
  f4 at(int i, int j, int layer) const {
    if (i < 0 || j < 0 || i >= w || j >= h) {
      if (!g_clamp_to_edge) return { 0, 0, 0, 0 };
      i = std::min(std::max(i, 0), w - 1);
      j = std::min(std::max(j, 0), h - 1);
    }
    const float* p = texels + 4 * ((static_cast<size_t>(layer) * h + j) * w + i);
    return { p[0], p[1], p[2], p[3] };
  }
};

// OpenCL 1.2 §8.2 linear filtering with normalised coordinates; repeat = CLK_ADDRESS_REPEAT,
// otherwise CLK_ADDRESS_CLAMP (kernels/backend/opencl/image.hpp:14-21).
f3 sample_linear(const Image& im, float s, float t, int layer, bool repeat) {
  float u, v;
  int i0, j0, i1, j1;
  if (repeat) {
    u = (s - floorf(s)) * im.w;
    v = (t - floorf(t)) * im.h;
    i0 = (int) floorf(u - 0.5f);
    j0 = (int) floorf(v - 0.5f);
    i1 = i0 + 1;
    j1 = j0 + 1;
    if (i0 < 0) i0 += im.w;
    if (j0 < 0) j0 += im.h;
    if (i1 > im.w - 1) i1 -= im.w;
    if (j1 > im.h - 1) j1 -= im.h;
  } else {
    u = s * im.w;
    v = t * im.h;
    i0 = (int) floorf(u - 0.5f);
    j0 = (int) floorf(v - 0.5f);
    i1 = i0 + 1;
    j1 = j0 + 1;
  }
  float a = (u - 0.5f) - floorf(u - 0.5f);
  float b = (v - 0.5f) - floorf(v - 0.5f);
  f4 t00 = im.at(i0, j0, layer), t10 = im.at(i1, j0, layer), t01 = im.at(i0, j1, layer),
     t11 = im.at(i1, j1, layer);
  float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  // ((T00 w00 + T10 w10) + T01 w01) + T11 w11, fused as clang fuses it (see the FP contract above)
  return { fmaf(t11.x, w11, fmaf(t01.x, w01, fmaf(t00.x, w00, t10.x * w10))),
           fmaf(t11.y, w11, fmaf(t01.y, w01, fmaf(t00.y, w00, t10.y * w10))),
           fmaf(t11.z, w11, fmaf(t01.z, w01, fmaf(t00.z, w00, t10.z * w10))) };
}

// ---- shading helpers ---------------------------------------------------------------------------
// kernels/transforms.hpp:32-40 — the comparison is done in double, as the literal 0.1 is
inline Mat3 create_basis(f3 n) {
  f3 helper = (static_cast<double>(fabsf(n.y)) > 0.1) ? f3 { 1, 0, 0 } : f3 { 0, 1, 0 };
  f3 u = normalize(cross(n, helper));
  f3 w = cross(u, n);
  return { u, n, w };
}
// kernels/transforms.hpp:18-22: a*bx + b*by + c*bz
inline f3 interp(f3 bc, f3 a, f3 b, f3 c) { return fma3(c, bc.z, fma3(a, bc.x, b * bc.y)); }
inline f2 interp(f3 bc, nova_float2 a, nova_float2 b, nova_float2 c) {
  return { fmaf(c.x, bc.z, fmaf(a.x, bc.x, b.x * bc.y)), fmaf(c.y, bc.z, fmaf(a.y, bc.x, b.y * bc.y)) };
}
inline float nz(float x) { return fmaxf(x, kNonZeroEps); }       // transforms.hpp:52
inline f3 nz(f3 x) { return max3(x, splat(kNonZeroEps)); }      // transforms.hpp:54-56
inline f3 reflect(f3 i, f3 n) { return fma3(n, -(2.0f * dot(n, i)), i); } // opencl/vector.hpp:23
inline f3 spherical(float theta, float phi) {                   // transforms.hpp:46-48
  return { nova_sinf(theta) * nova_cosf(phi), nova_cosf(theta), nova_sinf(theta) * nova_sinf(phi) };
}

// kernels/material.hpp:15-29
f3 read_material(const Image& mats, const nova_triangle_meta& m, f2 tc, int index, f3 dflt) {
  if (m.diffuse_index == -1 && m.metallic_index == -1 && m.roughness_index == -1 && m.normal_index == -1)
    return dflt;
  if (index == -1) return dflt;
  return sample_linear(mats, tc.x, tc.y, index, true);
}
// kernels/material.hpp:31-35
f3 read_sky(const Image& sky, f3 d) {
  float u = fmaf(nova_atan2f(d.z, d.x), kInvPi * 0.5f, 0.5f);
  float v = fmaf(nova_asinf(d.y), kInvPi, 0.5f);
  return sample_linear(sky, u, v, 0, true);
}
// kernels/material.hpp:37-62
f3 shading_normal(const Image& mats, const nova_triangle_meta& m, f2 tc, f3 bc) {
  f3 n = normalize(interp(bc, v3(m.normal1), v3(m.normal2), v3(m.normal3)));
  if (m.normal_index != -1) {
    f3 t = normalize(interp(bc, v3(m.tangent1), v3(m.tangent2), v3(m.tangent3)));
    f3 b = normalize(interp(bc, v3(m.bitangent1), v3(m.bitangent2), v3(m.bitangent3)));
    Mat3 tbn = transpose({ t, b, n });
    f3 pn = read_material(mats, m, tc, m.normal_index, { 0, 0, 0 });
    pn = normalize(fma3(pn, 2.0f, splat(-1.0f)));
    n = normalize(mul(tbn, pn));
  }
  return n;
}

// kernels/light.hpp:12-44
f3 light_normal(const nova_area_light& l) {
  if (l.type == NOVA_LIGHT_RECT) return normalize(v3(l.rect.normal));
  return normalize(cross(v3(l.tri.v2) - v3(l.tri.v1), v3(l.tri.v3) - v3(l.tri.v1)));
}
float light_area(const nova_area_light& l) {
  if (l.type == NOVA_LIGHT_RECT) return l.rect.dims.x * l.rect.dims.y;
  return 0.5f * length(cross(v3(l.tri.v2) - v3(l.tri.v1), v3(l.tri.v3) - v3(l.tri.v1)));
}
f3 light_sample(const nova_area_light& l, uint32_t& rng) {
  if (l.type == NOVA_LIGHT_RECT) {
    float r0 = rnd(rng), r1 = rnd(rng);
    float ox = l.rect.dims.x * fmaf(r0, 2.0f, -1.0f);
    float oy = l.rect.dims.y * fmaf(r1, 2.0f, -1.0f);
    Mat3 basis = create_basis(normalize(v3(l.rect.normal)));
    return v3(l.rect.position) + mul(basis, { ox, 0.0f, oy });
  }
  float r1 = sqrtf(rnd(rng));
  float r2 = rnd(rng);
  f3 b = (r1 * (1.0f - r2)) * v3(l.tri.v2);
  return fma3(v3(l.tri.v3), r1 * r2, fma3(v3(l.tri.v1), 1.0f - r1, b));
}

// kernels/brdf.hpp:20-203
struct Brdf {
  f3 in_dir { 0, 0, 0 }, out_dir, normal, m_normal { 0, 0, 0 }, diffuse;
  float metallic, roughness;
  bool ready = false;
  float D = 0, G = 0, i_dot_m = 0, i_dot_n = 0, o_dot_n = 0, n_dot_m = 0;
  f3 F { 0, 0, 0 }, f0 { 0, 0, 0 }, k_d { 0, 0, 0 }, t { 0, 0, 0 };

  Brdf(f3 out, f3 n, f3 dif, float met, float rough)
    : out_dir(out), normal(n), diffuse(dif), metallic(met), roughness(rough) {}
  // CookTorranceLightBRDF (brdf.hpp:183-194)
  Brdf(f3 in, f3 out, f3 n, f3 dif, float met, float rough) : Brdf(out, n, dif, met, rough) {
    in_dir = in;
    m_normal = normalize(in + out);
  }
  static f3 mix004(f3 dif, float met) { return fma3(dif - splat(0.04f), met, splat(0.04f)); }
  static f3 schlick(float c, f3 f0) {                                      // brdf.hpp:129-132
    float p = nova_powf(fmaxf(1.0f - c, 0.0f), 5.0f);
    return fma3(splat(1.0f) - f0, p, f0);
  }
  f3 sample(uint32_t& rng) {                                               // brdf.hpp:41-77
    float xi0 = rnd(rng), xi1 = rnd(rng), xi2 = rnd(rng);
    f0 = mix004(diffuse, metallic);
    o_dot_n = fmaxf(dot(out_dir, normal), 0.0f);
    f3 ks = schlick(o_dot_n, f0);
    f3 kd = (splat(1.0f) - ks) * (1.0f - metallic);
    f3 ratio = clamp3(ks / (ks + kd), 0.0f, 1.0f);
    float est_t = fmaxf(((ratio.x + ratio.y) + ratio.z) / 3.0f, 0.25f);
    Mat3 basis = transpose(create_basis(normal));
    float phi = (2.0f * kPi) * xi1;
    if (xi0 <= est_t) {
      float theta = nova_atanf(roughness * sqrtf(xi2 / (1.0f - xi2)));
      m_normal = mul(basis, spherical(theta, phi));
      in_dir = reflect(-out_dir, m_normal);
    } else {
      float theta = nova_acosf(sqrtf(xi2));
      in_dir = mul(basis, spherical(theta, phi));
      m_normal = normalize(in_dir + out_dir);
    }
    return in_dir;
  }
  void intermediates() {                                                   // brdf.hpp:134-159
    if (ready) return;
    i_dot_m = dot(in_dir, m_normal);
    i_dot_n = dot(in_dir, normal);
    o_dot_n = fmaxf(dot(out_dir, normal), 0.0f);
    n_dot_m = fmaxf(dot(normal, m_normal), 0.0f);
    float alpha = roughness * roughness;
    float a2 = alpha * alpha, c2 = n_dot_m * n_dot_m;
    float x = fmaf(c2, a2 - 1.0f, 1.0f);
    D = a2 / nz((kPi * x) * x);                                             // brdf.hpp:111-116
    f0 = mix004(diffuse, metallic);
    F = schlick(i_dot_m, f0);
    float k = roughness + 1.0f;
    k = k * k / 8.0f;
    auto g1 = [&](float c) { return c / nz(fmaf(c, 1.0f - k, k)); };       // brdf.hpp:119-121
    G = g1(i_dot_n) * g1(o_dot_n);
    k_d = (splat(1.0f) - F) * (1.0f - metallic);
    t = clamp3(F / (F + k_d), 0.0f, 1.0f);
    ready = true;
  }
  f3 eval() {                                                              // brdf.hpp:85-96
    intermediates();
    if (i_dot_m <= 0.0f || i_dot_n <= 0.0f) return { 0, 0, 0 };
    f3 lambert = diffuse * kInvPi;
    f3 ct = ((D * F) * G) / nz((4.0f * o_dot_n) * i_dot_n);
    return fma3(k_d, lambert, ct);
  }
  f3 pdf() {                                                               // brdf.hpp:103-107
    intermediates();
    f3 spec = ((t * D) * n_dot_m) / nz(4.0f * i_dot_m);
    return nz(fma3((splat(1.0f) - t) * i_dot_n, kInvPi, spec));
  }
  float light_pdf(f3 ln, float dist, float area) {                         // brdf.hpp:200-202
    return dist * dist / nz(fabsf(dot(ln, in_dir)) * area);
  }
};
inline f3 power_heuristic(f3 p, f3 s) {                                    // brdf.hpp:205-210
  f3 p2 = p * p, s2 = s * s;
  return p2 / (p2 + s2);
}

struct SceneView {
  const nova_triangle_data* tris;
  const nova_triangle_meta* meta;
  const nova_bvh_node* bvh;
  const nova_area_light* lights;
  uint32_t num_lights;
  Image materials, sky;
  // light pre-test of MIS rays (oracle_set_mis_pretest): the triangles that carry a light index, or null
  const std::vector<std::pair<uint32_t, int>>* light_tris = nullptr;
};

struct RayCount {
  uint64_t primary = 0, bounce = 0, shadow = 0, segments = 0;
  uint64_t mis_pretested = 0, shadow_skipped = 0;
  WorkCount work;
};

// Light pre-test of a MIS ray — NOT in the reference; the restatement of what libnova_hip's k_shade does, kept here
// so that (a) CPU tests can show the frame does not change by a bit and (b) the per-ray work of the rays that are
// still traversed can be counted.  The MIS ray's closest hit matters only when it is a triangle with a light index
// other than the shaded triangle's (kernels/raytrace.hpp:247-262).  Whatever the walk reaches, a triangle it ACCEPTS
// passes intersects_triangle on its own with length = FLT_MAX (the walk's threshold is only ever lower).  So with
// S = the light triangles (light != the shaded one's) that pass on their own:
//   S empty      -> the closest hit cannot contribute: no find_intersection call at all;
//   S not empty  -> far = max t over S; the call runs with hit.length = nextafter(far) instead of FLT_MAX.  If the
//                   reference's closest hit has t <= far, this call returns the very same hit (same triangles reached,
//                   same order, same ties); if not, that hit is not in S, contributes nothing, and neither does
//                   whatever this call returns (a miss).
// Returns false when the ray need not be traced; else sets `length`.
inline bool mis_pretest(const SceneView& S, const Ray& ray, int shaded_light, float& length) {
  float far = -1.0f;
  for (const auto& lt : *S.light_tris) {
    if (lt.second == shaded_light) continue;
    Hit h;
    if (hit_triangle(ray, h, static_cast<int>(lt.first), S.tris[lt.first], false)) far = fmaxf(far, h.length);
  }
  if (far < 0.0f) return false;
  length = nextafterf(fmaxf(far, 1e-30f), FLT_MAX);
  return true;
}

// kernels/raytrace.hpp:104-298.  max_depth = 0 reproduces the reference (unbounded); max_depth
// = d stops a path after it has shaded (incl. direct lighting) its d-th segment.
f3 trace_ray(uint32_t& rng, const nova_scene_params& P, int px, int py, const SceneView& S,
             bool denoise, f3& albedo_feature, f3& normal_feature, uint32_t max_depth, const Slack* slack,
             RayCount& rc, bool count_work) {
  auto find = [&](const nova_triangle_data* t, const nova_bvh_node* b, const Ray& r, Hit& h, bool any, WorkCount* w) {
    return slack ? find_intersection_ordered(t, b, r, h, any, w, *slack)
                 : find_intersection_reference(t, b, r, h, any, w);
  };
  WorkCount* wc = count_work ? &rc.work : nullptr;
  float jx = rnd(rng), jy = rnd(rng);
  const nova_eye_coords& E = P.eye_coords;
  float ax = E.coord_scale.x * ((static_cast<float>(px) - E.coord_dims.x) + jx);
  float ay = E.coord_scale.y * ((static_cast<float>(py) - E.coord_dims.y) + jy);
  Mat3 frame_t = transpose({ v3(E.eye_coord_frame.x), v3(E.eye_coord_frame.y), v3(E.eye_coord_frame.z) });
  f3 ray_dir = normalize(mul(frame_t, { ax, -ay, -1.0f }));
  f3 ray_pos = v3(E.eye_pos);
  f3 color { 0, 0, 0 }, throughput { 1, 1, 1 };
  bool direct = true;
  uint32_t depth = 0;

  while (true) {
    Ray ray(ray_pos, ray_dir, kRayEpsilon);
    Hit hit;
    (direct ? rc.primary : rc.bounce)++;
    if (!find(S.tris, S.bvh, ray, hit, false, wc)) {
      if (direct) {
        color = read_sky(S.sky, ray_dir);
        if (denoise) {
          albedo_feature = color / (color + splat(1.0f));
          normal_feature = { 0, 0, 0 };
        }
      }
      break;
    }
    rc.segments++;
    depth++;
    const nova_triangle_meta& m = S.meta[hit.tri];
    f3 point = fma3(ray.direction, hit.length, ray.origin);
    f2 tc = interp(hit.bary, m.texture_coord1, m.texture_coord2, m.texture_coord3);
    f3 diffuse = read_material(S.materials, m, tc, m.diffuse_index, v3(P.shading_diffuse)) * v3(m.kD);
    float metallic = read_material(S.materials, m, tc, m.metallic_index,
                                   splat(m.metallic == -1.0f ? P.shading_metallic : m.metallic)).x;
    float roughness = read_material(S.materials, m, tc, m.roughness_index,
                                    splat(m.roughness == -1.0f ? P.shading_roughness : m.roughness)).x;
    if (!P.path_tracing) {
      color = color + (m.light_index != -1 ? v3(S.lights[m.light_index].intensity) : diffuse);
      break;
    }
    f3 normal = shading_normal(S.materials, m, tc, hit.bary);
    f3 out_dir = -ray_dir;
    if (direct && denoise) {
      albedo_feature = diffuse;
      normal_feature = normal;
    }
    if (direct && m.light_index != -1) color = fma3(throughput, v3(S.lights[m.light_index].intensity), color);

    f3 direct_color { 0, 0, 0 };
    if (S.num_lights > 1 || (S.num_lights == 1 && m.light_index == -1)) {
      {
        uint32_t li;
        do {
          li = std::min(static_cast<uint32_t>(rnd(rng) * static_cast<float>(S.num_lights)), S.num_lights - 1);
        } while (static_cast<uint32_t>(m.light_index) == li);
        const nova_area_light& L = S.lights[li];
        f3 ln = light_normal(L);
        float la = light_area(L);
        f3 lp = light_sample(L, rng);
        f3 ldir = normalize(lp - point);
        float ldist = length(lp - point);
        Ray shadow(point, ldir, kRayEpsilon);
        Hit sh;
        sh.length = ldist - kRayEpsilon;
        rc.shadow++;
        bool traced = true;
        if (S.light_tris) {
          // libnova_hip's second rule (not in the reference): the light sample's term is computed before the shadow
          // ray is traced; when it is +0 in all three channels (the light is below the surface's horizon: eval()
          // returns zero) the ray's answer cannot change direct_color — it stays +0 either way — and the ray is
          // not traced
          Brdf b0(ldir, out_dir, normal, diffuse, metallic, roughness);
          f3 f0 = b0.eval();
          float lpdf0 = b0.light_pdf(ln, ldist, la);
          f3 w0 = power_heuristic(splat(lpdf0), b0.pdf());
          f3 term = ((((static_cast<float>(S.num_lights) * w0) * v3(L.intensity)) * f0) / lpdf0) * fmaxf(dot(normal, ldir), 0.0f);
          uint32_t bits[3];
          std::memcpy(bits, &term, sizeof(bits));
          traced = (bits[0] | bits[1] | bits[2]) != 0;
          if (!traced) rc.shadow_skipped++;
        }
        if (traced && !find(S.tris, S.bvh, shadow, sh, true, wc)) {
          Brdf b(ldir, out_dir, normal, diffuse, metallic, roughness);
          f3 f = b.eval();
          float lpdf = b.light_pdf(ln, ldist, la);
          f3 bpdf = b.pdf();
          f3 w = power_heuristic(splat(lpdf), bpdf);
          f3 c = (((static_cast<float>(S.num_lights) * w) * v3(L.intensity)) * f) / lpdf;
          direct_color = fma3(c, fmaxf(dot(normal, ldir), 0.0f), direct_color);
        }
      }
      {
        Brdf b(f3 { 0, 0, 0 }, out_dir, normal, diffuse, metallic, roughness);
        f3 ldir = b.sample(rng);
        Ray mis(point, ldir, kRayEpsilon);
        Hit mh;
        rc.bounce++;
        bool traced = true;
        if (S.light_tris) {
          traced = mis_pretest(S, mis, m.light_index, mh.length);
          if (!traced) rc.mis_pretested++;
        }
        if (traced && find(S.tris, S.bvh, mis, mh, false, wc)) {
          const nova_triangle_meta& lm = S.meta[mh.tri];
          if (lm.light_index != -1 && lm.light_index != m.light_index) {
            const nova_area_light& L = S.lights[lm.light_index];
            f3 ln = light_normal(L);
            float la = light_area(L);
            f3 f = b.eval();
            float lpdf = b.light_pdf(ln, mh.length, la);
            f3 bpdf = b.pdf();
            f3 w = power_heuristic(bpdf, splat(lpdf));
            f3 c = ((w * v3(L.intensity)) * f) / bpdf;
            direct_color = fma3(c, fmaxf(dot(normal, ldir), 0.0f), direct_color);
          }
        }
      }
    }
    color = fma3(throughput, direct_color, color);
    if (max_depth != 0 && depth >= max_depth) break;

    Brdf b(out_dir, normal, diffuse, metallic, roughness);
    f3 in_dir = b.sample(rng);
    f3 f = b.eval();
    f3 pdf = b.pdf();
    throughput = throughput * ((f / pdf) * fmaxf(dot(normal, in_dir), 0.0f));
    if (!direct) {
      float p = fminf(fmaxf(throughput.x, fmaxf(throughput.y, throughput.z)), 1.0f);
      if (rnd(rng) > p) break;
      throughput = throughput * (1.0f / p);
    }
    ray_pos = point;
    ray_dir = in_dir;
    direct = false;
  }
  return color;
}

inline float luma(f3 c) { return dot(c, { 0.299f, 0.587f, 0.114f }); }       // transforms.hpp:28-30

// kernels/anti_aliasing.hpp:14-150
f3 fxaa(const Image& im, float inv_w, float inv_h, float u, float v) {
  auto tap = [&](float s, float t) { return sample_linear(im, s, t, 0, false); };
  f3 center_color = tap(u, v);
  float center = luma(center_color);
  float down = luma(tap(u + 0.0f * inv_w, v + -1.0f * inv_h));
  float left = luma(tap(u + -1.0f * inv_w, v + 0.0f * inv_h));
  float right = luma(tap(u + 1.0f * inv_w, v + 0.0f * inv_h));
  float up = luma(tap(u + 0.0f * inv_w, v + 1.0f * inv_h));
  float lmax = fmaxf(center, fmaxf(up, fmaxf(left, fmaxf(right, down))));
  float lmin = fminf(center, fminf(up, fminf(left, fminf(right, down))));
  float range = lmax - lmin;
  if (range < fmaxf(0.0312f, lmax * 0.125f)) return center_color;
  float dl = luma(tap(u + -1.0f * inv_w, v + -1.0f * inv_h));
  float dr = luma(tap(u + 1.0f * inv_w, v + -1.0f * inv_h));
  float ul = luma(tap(u + -1.0f * inv_w, v + 1.0f * inv_h));
  float ur = luma(tap(u + 1.0f * inv_w, v + 1.0f * inv_h));
  float down_up = down + up, left_right = left + right;
  float left_c = dl + ul, down_c = dl + dr, right_c = dr + ur, up_c = ul + ur;
  float edge_h = fmaf(2.0f, fabsf(fmaf(-2.0f, center, down_up)), fabsf(fmaf(-2.0f, left, left_c))) +
                 fabsf(fmaf(-2.0f, right, right_c));
  float edge_v = fmaf(2.0f, fabsf(fmaf(-2.0f, center, left_right)), fabsf(fmaf(-2.0f, up, up_c))) +
                 fabsf(fmaf(-2.0f, down, down_c));
  bool horiz = edge_h >= edge_v;
  float l1 = horiz ? down : left, l2 = horiz ? up : right;
  float g1 = l1 - center, g2 = l2 - center;
  float grad_scaled = 0.25f * fmaxf(fabsf(g1), fabsf(g2));
  float step = horiz ? inv_h : inv_w;
  float local_avg;
  if (fabsf(g1) >= fabsf(g2)) {
    step = -step;
    local_avg = 0.5f * (l1 + center);
  } else {
    local_avg = 0.5f * (l2 + center);
  }
  float cu = u, cv = v;
  if (horiz) cv = fmaf(0.5f, step, cv); else cu = fmaf(0.5f, step, cu);
  float ox = horiz ? inv_w : 0.0f, oy = horiz ? 0.0f : inv_h;
  float u1 = cu - ox, v1 = cv - oy, u2 = cu + ox, v2 = cv + oy;
  bool end1 = false, end2 = false;
  float le1 = 0, le2 = 0;
  const float mods[12] = { 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.5f, 2.0f, 2.0f, 2.0f, 2.0f, 4.0f, 8.0f };
  for (int i = 0; i < 12; i++) {
    if (!end1) {
      le1 = luma(tap(u1, v1)) - local_avg;
      end1 = fabsf(le1) >= grad_scaled;
      u1 = fmaf(-mods[i], ox, u1);
      v1 = fmaf(-mods[i], oy, v1);
    }
    if (!end2) {
      le2 = luma(tap(u2, v2)) - local_avg;
      end2 = fabsf(le2) >= grad_scaled;
      u2 = fmaf(mods[i], ox, u2);
      v2 = fmaf(mods[i], oy, v2);
    }
    if (end1 && end2) break;
  }
  float d1 = horiz ? (u - u1) : (v - v1);
  float d2 = horiz ? (u2 - u) : (v2 - v);
  float dmin = fminf(d1, d2);
  float thick = d1 + d2;
  float pix_off = -dmin / thick + 0.5f;
  bool center_smaller = center < local_avg;
  bool correct = ((d1 < d2 ? le1 : le2) < 0.0f) != center_smaller;
  float final_off = correct ? pix_off : 0.0f;
  float lavg = (1.0f / 12.0f) * (fmaf(2.0f, down_up + left_right, left_c) + right_c);
  float s1 = fminf(fmaxf(fabsf(lavg - center) / range, 0.0f), 1.0f);
  float s2 = (fmaf(-2.0f, s1, 3.0f) * s1) * s1;
  float sfin = (s2 * s2) * 0.75f;
  final_off = fmaxf(final_off, sfin);
  float fu = u, fv = v;
  if (horiz) fv = fmaf(final_off, step, fv); else fu = fmaf(final_off, step, fu);
  return tap(fu, fv);
}

int g_threads = 0;
uint32_t g_mis_pretest_tris = 0; // oracle_set_mis_pretest: number of triangles (0 = off, the reference's schedule)

}

extern "C" {

void oracle_set_threads(int n) { g_threads = n; }
void oracle_set_clamp_to_edge(int on) { g_clamp_to_edge = on ? 1 : 0; }
// n_tris > 0: oracle_kernel_raytrace applies libnova_hip's two rules for rays whose answer cannot matter — the light
// pre-test of MIS rays (mis_pretest above) and the shadow rays of light samples whose term is +0; counters[6] / [7] =
// MIS rays / shadow rays not traced.  counters[1] and [2] keep counting every ray of the reference's schedule.
void oracle_set_mis_pretest(uint32_t n_tris) { g_mis_pretest_tris = n_tris; }

uint32_t oracle_hash(uint32_t s) { return wang_hash(s); }
float oracle_rand(uint32_t* st) { return rnd(*st); }
int oracle_intersects_aabb(const float* p, const float* d, const float* top, const float* bot) {
  Ray r({ p[0], p[1], p[2] }, { d[0], d[1], d[2] }, kRayEpsilon);
  float a, b;
  return slab(r, { top[0], top[1], top[2] }, { bot[0], bot[1], bot[2] }, a, b);
}
void oracle_brdf(const float* o, const float* n, const float* dif, float met, float rough,
                 uint32_t* rng, float* out) {
  Brdf b({ o[0], o[1], o[2] }, { n[0], n[1], n[2] }, { dif[0], dif[1], dif[2] }, met, rough);
  f3 in = b.sample(*rng), e = b.eval(), p = b.pdf();
  float r[9] = { in.x, in.y, in.z, e.x, e.y, e.z, p.x, p.y, p.z };
  std::memcpy(out, r, sizeof(r));
}

// include/nova_math.h on arrays, for tests/test_math.py.
// fn: 0 sin 1 cos 2 atan 3 atan2(x,y) 4 asin 5 acos 6 exp 7 log 8 pow(x,y)
void oracle_math(int fn, const float* x, const float* y, float* out, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) {
    switch (fn) {
      case 0: out[i] = nova_sinf(x[i]); break;
      case 1: out[i] = nova_cosf(x[i]); break;
      case 2: out[i] = nova_atanf(x[i]); break;
      case 3: out[i] = nova_atan2f(x[i], y[i]); break;
      case 4: out[i] = nova_asinf(x[i]); break;
      case 5: out[i] = nova_acosf(x[i]); break;
      case 6: out[i] = nova_expf(x[i]); break;
      case 7: out[i] = nova_logf(x[i]); break;
      default: out[i] = nova_powf(x[i], y[i]); break;
    }
  }
}

// kernels/kernels.cu:9-40.  counters (optional, 8 x uint64): primary, bounce, shadow rays,
// shaded segments, node visits, triangle tests.  ordered = 1 selects the culled traversal.
int oracle_kernel_raytrace(const nova_scene_params* params, uint32_t time, float* temp_color1,
                           uint32_t width, uint32_t height, const nova_triangle_data* tris,
                           const nova_triangle_meta* meta, const nova_bvh_node* bvh,
                           const nova_area_light* lights, uint32_t num_lights, const float* materials,
                           uint32_t mat_w, uint32_t mat_h, uint32_t mat_layers, const float* sky,
                           uint32_t sky_w, uint32_t sky_h, uint32_t denoise_available, float* albedo1,
                           float* normal1, uint32_t max_depth, int ordered, int count_work,
                           uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t band_period,
                           uint64_t* counters) {
  SceneView S { tris, meta, bvh, lights, num_lights,
                { materials, (int) mat_w, (int) mat_h, (int) mat_layers }, { sky, (int) sky_w, (int) sky_h, 1 } };
  uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  row_end = std::min(row_end, height);
  std::vector<std::pair<uint32_t, int>> light_tris;
  if (g_mis_pretest_tris) {
    for (uint32_t i = 0; i < g_mis_pretest_tris; i++)
      if (meta[i].light_index != -1) light_tris.push_back({ i, meta[i].light_index });
    S.light_tris = &light_tris;
  }
  Slack slack;
  if (ordered) slack = compute_slack(tris, bvh);
  const Slack* slack_ptr = ordered ? &slack : nullptr;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : c0, c1, c2, c3, c4, c5, c6, c7) num_threads(g_threads > 0 ? g_threads : omp_get_max_threads())
  for (int y = (int) row_begin; y < (int) row_end; y++) {
    RayCount rc;
    // optional row sampling: bands of `band_rows` rows, one band in every `band_period`
    if (band_period > 1 && band_rows > 0 && ((static_cast<uint32_t>(y) - row_begin) / band_rows) % band_period != 0) continue;
    for (int x = 0; x < (int) width; x++) {
      uint32_t rng = wang_hash(static_cast<uint32_t>(y) * width + static_cast<uint32_t>(x) + wang_hash(time));
      f3 albedo { 0, 0, 0 }, normal { 0, 0, 0 };
      f3 c = trace_ray(rng, *params, x, y, S, denoise_available != 0, albedo, normal, max_depth,
                       slack_ptr, rc, count_work != 0);
      size_t o = 4 * (static_cast<size_t>(y) * width + x);
      temp_color1[o] = c.x; temp_color1[o + 1] = c.y; temp_color1[o + 2] = c.z; temp_color1[o + 3] = 1.0f;
      if (denoise_available) {
        albedo1[o] = albedo.x; albedo1[o + 1] = albedo.y; albedo1[o + 2] = albedo.z; albedo1[o + 3] = 1.0f;
        normal1[o] = normal.x; normal1[o + 1] = normal.y; normal1[o + 2] = normal.z; normal1[o + 3] = 1.0f;
      }
    }
    c0 += rc.primary; c1 += rc.bounce; c2 += rc.shadow; c3 += rc.segments;
    c4 += rc.work.nodes; c5 += rc.work.tris;
    c6 += rc.mis_pretested; c7 += rc.shadow_skipped;
  }
  if (counters) {
    counters[0] += c0; counters[1] += c1; counters[2] += c2; counters[3] += c3;
    counters[4] += c4; counters[5] += c5;
    counters[6] += c6; counters[7] += c7;
  }
  return 0;
}

// kernels/kernels.cu:42-92.  Like the reference, every input is read through the LINEAR, CLAMP
// sampler at the pixel's centre (kernels.cu:59-64, transforms.hpp:42-44): that is the texel itself
// except where (x+0.5)/W*W rounds off x+0.5 (weights within 1e-7 of 1/0) and where a neighbour
// holds NaN/inf (NaN * 0 = NaN: the renormalised zero normals of sky pixels spread, as they do in
// the reference).
int oracle_kernel_accumulate(int32_t sample_index, uint32_t denoise_available, const float* color1,
                             const float* albedo1, const float* normal1, const float* prev_color,
                             const float* prev_albedo, const float* prev_normal, float* color2,
                             float* albedo2, float* normal2, uint32_t width, uint32_t height) {
  const float si = static_cast<float>(sample_index), si1 = static_cast<float>(sample_index + 1);
#pragma omp parallel for
  for (int y = 0; y < (int) height; y++)
  for (int x = 0; x < (int) width; x++) {
    const size_t i = static_cast<size_t>(y) * width + x;
    const float u = (static_cast<float>(x) + 0.5f) / static_cast<float>(width);
    const float v = (static_cast<float>(y) + 0.5f) / static_cast<float>(height);
    auto ld = [&](const float* p) { return sample_linear(Image { p, (int) width, (int) height, 1 }, u, v, 0, false); };
    auto st = [&](float* p, f3 c3) { p[4 * i] = c3.x; p[4 * i + 1] = c3.y; p[4 * i + 2] = c3.z; p[4 * i + 3] = 1.0f; };
    f3 c = ld(color1);
    f3 a = denoise_available ? ld(albedo1) : f3 { 0, 0, 0 };
    f3 nn = denoise_available ? ld(normal1) : f3 { 0, 0, 0 };
    if (sample_index != 0) {
      c = fma3(ld(prev_color), si, c) / si1;
      if (denoise_available) {
        a = fma3(ld(prev_albedo), si, a) / si1;
        nn = normalize(fma3(ld(prev_normal), si, nn) / si1);
      }
    }
    st(color2, c);
    if (denoise_available) {
      st(albedo2, a);
      st(normal2, nn);
    }
  }
  return 0;
}

// kernels/kernels.cu:94-116 with transforms.hpp:14-26
int oracle_kernel_post_process(const nova_scene_params* params, const float* color2, uint8_t* pixels,
                               uint32_t width, uint32_t height) {
  Image im { color2, (int) width, (int) height, 1 };
  const float inv_w = 1.0f / static_cast<float>(width), inv_h = 1.0f / static_cast<float>(height);
#pragma omp parallel for
  for (int y = 0; y < (int) height; y++) {
    for (int x = 0; x < (int) width; x++) {
      f3 c;
      float u = (static_cast<float>(x) + 0.5f) / static_cast<float>(width);
      float v = (static_cast<float>(y) + 0.5f) / static_cast<float>(height);
      if (params->anti_aliasing) {
        c = fxaa(im, inv_w, inv_h, u, v);
      } else {
        c = sample_linear(im, u, v, 0, false);
      }
      f3 e = { nova_expf(-c.x * params->exposure), nova_expf(-c.y * params->exposure), nova_expf(-c.z * params->exposure) };
      f3 m = splat(1.0f) - e;
      const float g = 1.0f / 2.2f;
      f3 o = { nova_powf(m.x, g) * 255.0f, nova_powf(m.y, g) * 255.0f, nova_powf(m.z, g) * 255.0f };
      uint8_t* q = pixels + 4 * (static_cast<size_t>(y) * width + x);
      q[0] = static_cast<uint8_t>(static_cast<int>(o.x));
      q[1] = static_cast<uint8_t>(static_cast<int>(o.y));
      q[2] = static_cast<uint8_t>(static_cast<int>(o.z));
      q[3] = 255;
    }
  }
  return 0;
}

// find_intersection on a batch; mode 0 = reference walk, 1 = ordered + culled.
// counters (optional, 2 x uint64): node visits (box tests), triangle tests.
int oracle_trace_rays_work(const nova_triangle_data* tris, const nova_bvh_node* bvh, const float* point_tmax,
                           const float* dir_flags, float* hits, uint64_t n_rays, int mode, uint64_t* counters,
                           uint32_t* per_ray);
int oracle_trace_rays(const nova_triangle_data* tris, const nova_bvh_node* bvh, const float* point_tmax,
                      const float* dir_flags, float* hits, uint64_t n_rays, int mode, uint64_t* counters) {
  return oracle_trace_rays_work(tris, bvh, point_tmax, dir_flags, hits, n_rays, mode, counters, nullptr);
}
// per_ray (optional, 2 x uint32 per ray): box tests and triangle tests of that ray (work distribution)
int oracle_trace_rays_work(const nova_triangle_data* tris, const nova_bvh_node* bvh, const float* point_tmax,
                           const float* dir_flags, float* hits, uint64_t n_rays, int mode, uint64_t* counters,
                           uint32_t* per_ray) {
  uint64_t cn = 0, ct = 0;
  Slack slack;
  if (mode) slack = compute_slack(tris, bvh);
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : cn, ct) num_threads(g_threads > 0 ? g_threads : omp_get_max_threads())
  for (int64_t i = 0; i < (int64_t) n_rays; i++) {
    uint32_t flags;
    std::memcpy(&flags, &dir_flags[4 * i + 3], 4);
    Ray ray({ point_tmax[4 * i], point_tmax[4 * i + 1], point_tmax[4 * i + 2] },
            { dir_flags[4 * i], dir_flags[4 * i + 1], dir_flags[4 * i + 2] }, kRayEpsilon);
    Hit h;
    h.length = point_tmax[4 * i + 3];
    WorkCount wc;
    const bool want = counters != nullptr || per_ray != nullptr;
    if (mode) find_intersection_ordered(tris, bvh, ray, h, (flags & 1u) != 0, want ? &wc : nullptr, slack);
    else find_intersection_reference(tris, bvh, ray, h, (flags & 1u) != 0, want ? &wc : nullptr);
    cn += wc.nodes;
    ct += wc.tris;
    if (per_ray) {
      per_ray[2 * i] = static_cast<uint32_t>(wc.nodes);
      per_ray[2 * i + 1] = static_cast<uint32_t>(wc.tris);
    }
    hits[4 * i] = h.length;
    hits[4 * i + 1] = h.bary.y;
    hits[4 * i + 2] = h.bary.z;
    std::memcpy(&hits[4 * i + 3], &h.tri, 4);
  }
  if (counters) {
    counters[0] += cn;
    counters[1] += ct;
  }
  return 0;
}

}
