// Host-side scene preparation: produces the inputs of the hot path (FlatBVHNode[], TriangleData[],
// TriangleMetaData[], AreaLightData[]) in the layouts of include/nova_hip.h.
//
// In a real Nova build these arrays come from Nova's own, untouched host code
// (IntersectableManager::build, src/intersectables/intersectable_manager.cpp:102-186, and
// BVH::build, src/bvh/bvh.cpp:17-249).  That code cannot be compiled in this tree (it pulls in
// assimp's CMake-generated config.h), so tests and bench.py use this from-scratch builder, which
// follows the same algorithm and produces the same data format:
//   * binned-SAH BVH2, bins = 1024/(depth+1), leaves <= 8 triangles where a split pays
//     (bvh.cpp:136-203, src/constants.hpp:31-32);
//   * DFS pre-order flattening, left child adjacent, larger-surface-area child first, leaf
//     triangles appended in visiting order (bvh.cpp:211-249);
//   * Woop unit-triangle matrices = rows 0..2 of transpose(inverse([e1 e2 n-v1 v1]))
//     (intersectable_manager.cpp:129-145);
//   * rect lights and the ground plane become 2x2 / 4x4 grids of quads (:8-42, :55-94); every
//     emissive triangle becomes one TRI light (:149-159).
// Plain C++17 + OpenMP, no GPU code; exposed to Python through a C ABI (nova_amd/scene.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <vector>

#include "nova_hip.h"

namespace {

struct V3 {
  float x, y, z;
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline V3 operator*(float s, V3 a) { return a * s; }
inline V3 vmin(V3 a, V3 b) { return { std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z) }; }
inline V3 vmax(V3 a, V3 b) { return { std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z) }; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
  return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
inline V3 normalize(V3 a) {
  float inv = 1.0f / std::sqrt(dot(a, a));
  return a * inv;
}

// src/intersectables/aabb.hpp:10-41
struct Box {
  V3 top, bottom;
  static Box empty() {
    const float m = std::numeric_limits<float>::max();
    return { { -m, -m, -m }, { m, m, m } };
  }
  void grow(const Box& o) {
    top = vmax(top, o.top);
    bottom = vmin(bottom, o.bottom);
  }
  float surface_area() const {
    V3 d = top - bottom;
    return (d.x * d.y + d.y * d.z + d.z * d.x) * 2.0f;
  }
  float cost(size_t n) const { return surface_area() * static_cast<float>(n); }
  V3 center() const { return (top + bottom) * 0.5f; }
};

struct Tri {
  V3 v1, v2, v3;
  Box bounds() const { return { vmax(v1, vmax(v2, v3)), vmin(v1, vmin(v2, v3)) }; }
};

constexpr size_t kMinTrianglesPerLeaf = 8; // src/constants.hpp:31
constexpr float kMaxBins = 1024.0f;        // src/constants.hpp:32

struct BuildNode {
  Box box = Box::empty();
  uint32_t begin = 0, end = 0; // range of `order`
  std::unique_ptr<BuildNode> left, right;
};

struct Builder {
  const std::vector<Tri>& tris;
  std::vector<Box> bounds;
  std::vector<uint32_t> order;

  explicit Builder(const std::vector<Tri>& t) : tris(t), bounds(t.size()), order(t.size()) {
    for (size_t i = 0; i < t.size(); i++) bounds[i] = t[i].bounds();
    std::iota(order.begin(), order.end(), 0u);
  }

  struct Split {
    float cost = std::numeric_limits<float>::max();
    uint32_t index = 0;
    int axis = -1;
    Box left, right;
  };

  // bvh.cpp:136-203.  `cost < other.cost ? this : other`: on equal cost the later candidate wins.
  void split(BuildNode* node, int depth) {
    const uint32_t n = node->end - node->begin;
    if (n <= kMinTrianglesPerLeaf) return;
    const uint32_t num_bins = static_cast<uint32_t>(kMaxBins / static_cast<float>(depth + 1));
    if (num_bins <= 1) return;
    const uint32_t num_splits = num_bins - 1;

    Split best;
    best.cost = node->box.cost(n);

    const V3 start = node->box.bottom, stop = node->box.top;
    // bvh.cpp:152: glm's vec3 / float divides each component (include/glm/detail/type_vec3.inl:715)
    const float fbins = static_cast<float>(num_bins);
    const V3 d = stop - start;
    const V3 step = { d.x / fbins, d.y / fbins, d.z / fbins };
    const V3 inv_step = { 1.0f / step.x, 1.0f / step.y, 1.0f / step.z };

    std::vector<uint32_t> bin_of(static_cast<size_t>(n) * 3);
    for (uint32_t k = 0; k < n; k++) {
      const V3 c = bounds[order[node->begin + k]].center();
      for (int a = 0; a < 3; a++) {
        float f = (c[a] - start[a]) * inv_step[a];
        uint32_t b = (f > 0.0f) ? static_cast<uint32_t>(std::min(f, 4.0e9f)) : 0u; // NaN -> 0
        bin_of[3 * static_cast<size_t>(k) + a] = std::min(b, num_splits);
      }
    }

    std::vector<Box> bins(num_bins), right_bounds(num_bins);
    std::vector<uint32_t> counts(num_bins);
    for (int axis = 0; axis < 3; axis++) {
      if (std::fabs(stop[axis] - start[axis]) < 1e-4f) continue;
      std::fill(bins.begin(), bins.end(), Box::empty());
      std::fill(counts.begin(), counts.end(), 0u);
      for (uint32_t k = 0; k < n; k++) {
        uint32_t b = bin_of[3 * static_cast<size_t>(k) + axis];
        bins[b].grow(bounds[order[node->begin + k]]);
        counts[b]++;
      }
      // bvh.cpp:54-105
      right_bounds[num_bins - 1] = bins[num_bins - 1];
      for (int32_t i = static_cast<int32_t>(num_bins) - 2; i >= 0; i--) {
        right_bounds[i] = right_bounds[i + 1];
        right_bounds[i].grow(bins[i]);
      }
      Box left = bins[0];
      uint32_t left_n = counts[0];
      for (uint32_t s = 0; s < num_splits; s++) {
        const uint32_t right_n = n - left_n;
        if (left_n > 2 && right_n > 2) {
          // NOTE: the reference pairs left_bounds[s] with right_bounds[s] (bvh.cpp:77-81), i.e.
          // the right box still includes bin s.  Reproduced as is.
          const Box& right = right_bounds[s];
          const float total = left.cost(left_n) + right.cost(right_n);
          if (!(best.cost < total)) {
            best.cost = total;
            best.index = s;
            best.axis = axis;
            best.left = left;
            best.right = right;
          }
        }
        left.grow(bins[s + 1]);
        left_n += counts[s + 1];
      }
    }
    if (best.axis == -1) return;

    // bvh.cpp:107-134: stable split on the bin index along the best axis.
    std::vector<uint32_t> l, r;
    l.reserve(n);
    r.reserve(n);
    for (uint32_t k = 0; k < n; k++) {
      uint32_t id = order[node->begin + k];
      (bin_of[3 * static_cast<size_t>(k) + best.axis] <= best.index ? l : r).push_back(id);
    }
    std::copy(l.begin(), l.end(), order.begin() + node->begin);
    std::copy(r.begin(), r.end(), order.begin() + node->begin + l.size());
    bin_of.clear();
    bin_of.shrink_to_fit();

    node->left = std::make_unique<BuildNode>();
    node->right = std::make_unique<BuildNode>();
    node->left->box = best.left;
    node->right->box = best.right;
    node->left->begin = node->begin;
    node->left->end = node->begin + static_cast<uint32_t>(l.size());
    node->right->begin = node->left->end;
    node->right->end = node->end;

    BuildNode* lp = node->left.get();
    BuildNode* rp = node->right.get();
    if (n > 4096) {
#pragma omp task
      split(lp, depth + 1);
#pragma omp task
      split(rp, depth + 1);
#pragma omp taskwait
    } else {
      split(lp, depth + 1);
      split(rp, depth + 1);
    }
  }

  // bvh.cpp:211-249
  size_t flatten(BuildNode* node, std::vector<nova_bvh_node>& out, std::vector<uint32_t>& tri_order) {
    const size_t idx = out.size();
    nova_bvh_node f;
    f.top_offset_left = { node->box.top.x, node->box.top.y, node->box.top.z, 0.0f };
    f.bottom_num_right = { node->box.bottom.x, node->box.bottom.y, node->box.bottom.z, 0.0f };
    out.push_back(f);
    if (!node->left) {
      out[idx].top_offset_left.w = static_cast<float>(tri_order.size());
      out[idx].bottom_num_right.w = -static_cast<float>(node->end - node->begin);
      for (uint32_t k = node->begin; k < node->end; k++) tri_order.push_back(order[k]);
    } else {
      if (node->right->box.surface_area() > node->left->box.surface_area()) {
        std::swap(node->left, node->right);
      }
      flatten(node->left.get(), out, tri_order);
      out[idx].bottom_num_right.w = static_cast<float>(flatten(node->right.get(), out, tri_order));
    }
    return idx;
  }
};

// glm::inverse(mat4) with glm's own operation order (include/glm/detail/func_matrix.inl:294-352, the
// scalar path: mat4 is a packed type): m[c][r] is column c, row r.  No contraction (plain g++ -O2, no FMA).
void glm_inverse(const float m[4][4], float inv[4][4]) {
  const float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
  const float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
  const float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
  const float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
  const float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
  const float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
  const float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
  const float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
  const float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
  const float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
  const float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
  const float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
  const float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
  const float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
  const float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
  const float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
  const float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
  const float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
  const float fac0[4] = { c00, c00, c02, c03 }, fac1[4] = { c04, c04, c06, c07 }, fac2[4] = { c08, c08, c10, c11 };
  const float fac3[4] = { c12, c12, c14, c15 }, fac4[4] = { c16, c16, c18, c19 }, fac5[4] = { c20, c20, c22, c23 };
  const float vec0[4] = { m[1][0], m[0][0], m[0][0], m[0][0] }, vec1[4] = { m[1][1], m[0][1], m[0][1], m[0][1] };
  const float vec2[4] = { m[1][2], m[0][2], m[0][2], m[0][2] }, vec3[4] = { m[1][3], m[0][3], m[0][3], m[0][3] };
  const float sign_a[4] = { 1.0f, -1.0f, 1.0f, -1.0f }, sign_b[4] = { -1.0f, 1.0f, -1.0f, 1.0f };
  for (int r = 0; r < 4; r++) {
    inv[0][r] = (vec1[r] * fac0[r] - vec2[r] * fac1[r] + vec3[r] * fac2[r]) * sign_a[r];
    inv[1][r] = (vec0[r] * fac0[r] - vec2[r] * fac3[r] + vec3[r] * fac4[r]) * sign_b[r];
    inv[2][r] = (vec0[r] * fac1[r] - vec1[r] * fac3[r] + vec3[r] * fac5[r]) * sign_a[r];
    inv[3][r] = (vec0[r] * fac2[r] - vec1[r] * fac4[r] + vec2[r] * fac5[r]) * sign_b[r];
  }
  const float d0 = m[0][0] * inv[0][0], d1 = m[0][1] * inv[1][0], d2 = m[0][2] * inv[2][0], d3 = m[0][3] * inv[3][0];
  const float one_over_det = 1.0f / ((d0 + d1) + (d2 + d3));
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) inv[c][r] = inv[c][r] * one_over_det;
}

// intersectable_manager.cpp:8-42
void make_plane(V3 position, V3 normal, float dim_x, float dim_y, uint32_t div_x, uint32_t div_y,
                std::vector<Tri>& out) {
  V3 v = normal;
  V3 aux = std::fabs(v.y) > 0.1 ? V3 { 1.0f, 0.0f, 0.0f } : V3 { 0.0f, 1.0f, 0.0f };
  V3 u = normalize(cross(v, aux));
  V3 w = cross(u, v);
  V3 corner = position - (dim_x / 2.0f) * u - (dim_y / 2.0f) * w;
  float sx = dim_x / static_cast<float>(div_x), sy = dim_y / static_cast<float>(div_y);
  for (uint32_t j = 0; j < div_y; j++) {
    for (uint32_t i = 0; i < div_x; i++) {
      V3 p = corner + ((i + 0.5f) * sx) * u + ((j + 0.5f) * sy) * w;
      // (the reference offsets by a full division size either side, so neighbouring quads overlap)
      out.push_back({ p - sx * u - sy * w, p - sx * u + sy * w, p + sx * u + sy * w });
      out.push_back({ p + sx * u + sy * w, p + sx * u - sy * w, p - sx * u - sy * w });
    }
  }
}

struct Built {
  std::vector<nova_bvh_node> bvh;
  std::vector<nova_triangle_data> tris;
  std::vector<nova_triangle_meta> meta;
  std::vector<nova_area_light> lights;
  std::vector<uint32_t> tri_order; // output triangle i = input triangle tri_order[i]
  uint32_t max_depth = 0;
};

nova_float3 nf3(V3 v) { return { v.x, v.y, v.z }; }

}

extern "C" {

typedef struct {
  float intensity[3];
  float position[3];
  float normal[3];
  float dims[2];
} nova_scene_rect_light;

typedef struct {
  float position[3];
  float normal[3];
  float dims[2];
  float diffuse[3];
  float metallic;
  float roughness;
} nova_scene_ground_plane;

// verts: n*9 floats (v1 v2 v3); meta: n entries in device layout with light_index = -1;
// emissive: n*3 floats (kE) or NULL.  Rect lights and the optional ground plane are appended
// after the model triangles in the reference's order (src/core/raytracer.cpp:113-123).
void* nova_scene_build(const float* verts, const nova_triangle_meta* meta, const float* emissive,
                       uint64_t n, const nova_scene_rect_light* rect_lights, uint32_t n_rect_lights,
                       const nova_scene_ground_plane* ground) {
  std::vector<Tri> tris;
  std::vector<nova_triangle_meta> metas;
  std::vector<V3> ke;
  tris.reserve(n + 64);
  metas.reserve(n + 64);
  ke.reserve(n + 64);
  for (uint64_t i = 0; i < n; i++) {
    const float* p = verts + 9 * i;
    tris.push_back({ { p[0], p[1], p[2] }, { p[3], p[4], p[5] }, { p[6], p[7], p[8] } });
    metas.push_back(meta[i]);
    ke.push_back(emissive ? V3 { emissive[3 * i], emissive[3 * i + 1], emissive[3 * i + 2] }
                          : V3 { 0, 0, 0 });
  }

  auto built = std::make_unique<Built>();

  // add_light (intersectable_manager.cpp:55-84)
  for (uint32_t l = 0; l < n_rect_lights; l++) {
    const nova_scene_rect_light& rl = rect_lights[l];
    V3 nrm = { rl.normal[0], rl.normal[1], rl.normal[2] };
    std::vector<Tri> plane;
    make_plane({ rl.position[0], rl.position[1], rl.position[2] }, nrm, rl.dims[0], rl.dims[1], 2, 2,
               plane);
    nova_triangle_meta m;
    std::memset(&m, 0, sizeof(m));
    m.normal1 = m.normal2 = m.normal3 = nf3(nrm);
    m.kD = { 1.0f, 1.0f, 1.0f };
    m.metallic = -1.0f;
    m.roughness = -1.0f;
    m.diffuse_index = m.metallic_index = m.roughness_index = m.normal_index = -1;
    m.light_index = static_cast<int32_t>(l);
    for (const Tri& t : plane) {
      tris.push_back(t);
      metas.push_back(m);
      ke.push_back({ rl.intensity[0], rl.intensity[1], rl.intensity[2] });
    }
    nova_area_light al;
    std::memset(&al, 0, sizeof(al));
    al.intensity = { rl.intensity[0], rl.intensity[1], rl.intensity[2] };
    al.rect.position = { rl.position[0], rl.position[1], rl.position[2] };
    al.rect.normal = nf3(nrm);
    al.rect.dims = { rl.dims[0], rl.dims[1], 0.0f };
    al.type = NOVA_LIGHT_RECT;
    built->lights.push_back(al);
  }
  // add_ground_plane (intersectable_manager.cpp:86-94)
  if (ground) {
    V3 nrm = { ground->normal[0], ground->normal[1], ground->normal[2] };
    std::vector<Tri> plane;
    make_plane({ ground->position[0], ground->position[1], ground->position[2] }, nrm,
               ground->dims[0], ground->dims[1], 4, 4, plane);
    nova_triangle_meta m;
    std::memset(&m, 0, sizeof(m));
    m.normal1 = m.normal2 = m.normal3 = nf3(nrm);
    m.kD = { ground->diffuse[0], ground->diffuse[1], ground->diffuse[2] };
    m.metallic = ground->metallic;
    m.roughness = ground->roughness;
    m.diffuse_index = m.metallic_index = m.roughness_index = m.normal_index = -1;
    m.light_index = -1;
    for (const Tri& t : plane) {
      tris.push_back(t);
      metas.push_back(m);
      ke.push_back({ 0, 0, 0 });
    }
  }
  if (tris.empty()) return nullptr;

  // BVH::build (bvh.cpp:17-52)
  Builder b(tris);
  BuildNode root;
  root.begin = 0;
  root.end = static_cast<uint32_t>(tris.size());
  for (const Box& bb : b.bounds) root.box.grow(bb);
#pragma omp parallel
  {
#pragma omp single
    b.split(&root, 0);
  }
  built->bvh.reserve(tris.size() / 2);
  built->tri_order.reserve(tris.size());
  b.flatten(&root, built->bvh, built->tri_order);

  // intersectable_manager.cpp:129-183
  const size_t nt = built->tri_order.size();
  built->tris.resize(nt);
  built->meta.resize(nt);
  for (size_t i = 0; i < nt; i++) {
    const Tri& t = tris[built->tri_order[i]];
    V3 e1 = t.v2 - t.v1, e2 = t.v3 - t.v1;
    V3 nrm = normalize(cross(e1, e2));
    V3 c2 = nrm - t.v1;
    // transform[0..3] = (e1,0) (e2,0) (n-v1,0) (v1,1): columns, m[c][r]
    const float m[4][4] = { { e1.x, e1.y, e1.z, 0.0f }, { e2.x, e2.y, e2.z, 0.0f }, { c2.x, c2.y, c2.z, 0.0f },
                            { t.v1.x, t.v1.y, t.v1.z, 1.0f } };
    float inv[4][4];
    glm_inverse(m, inv);
    // transpose(inverse(M))[k] = row k of the inverse = (inv[0][k], inv[1][k], inv[2][k], inv[3][k])
    built->tris[i].x = { inv[0][0], inv[1][0], inv[2][0], inv[3][0] };
    built->tris[i].y = { inv[0][1], inv[1][1], inv[2][1], inv[3][1] };
    built->tris[i].z = { inv[0][2], inv[1][2], inv[2][2], inv[3][2] };

    nova_triangle_meta mm = metas[built->tri_order[i]];
    V3 e = ke[built->tri_order[i]];
    if (mm.light_index == -1 && std::sqrt(dot(e, e)) > 0.0f) {
      mm.light_index = static_cast<int32_t>(built->lights.size());
      nova_area_light al;
      std::memset(&al, 0, sizeof(al));
      al.intensity = nf3(e);
      al.tri.v1 = nf3(t.v1);
      al.tri.v2 = nf3(t.v2);
      al.tri.v3 = nf3(t.v3);
      al.type = NOVA_LIGHT_TRI;
      built->lights.push_back(al);
    }
    built->meta[i] = mm;
  }

  // tree depth (root = depth 1), for information
  {
    std::vector<std::pair<uint32_t, uint32_t>> st;
    st.push_back({ 0u, 1u });
    while (!st.empty()) {
      auto [i, d] = st.back();
      st.pop_back();
      built->max_depth = std::max(built->max_depth, d);
      if (built->bvh[i].bottom_num_right.w >= 0.0f) {
        st.push_back({ i + 1, d + 1 });
        st.push_back({ static_cast<uint32_t>(built->bvh[i].bottom_num_right.w), d + 1 });
      }
    }
  }
  return built.release();
}

// out = {num_nodes, num_triangles, num_lights, max_depth}
void nova_scene_sizes(const void* handle, uint64_t* out) {
  const Built* b = static_cast<const Built*>(handle);
  out[0] = b->bvh.size();
  out[1] = b->tris.size();
  out[2] = b->lights.size();
  out[3] = b->max_depth;
}

void nova_scene_copy(const void* handle, nova_bvh_node* bvh, nova_triangle_data* tris,
                     nova_triangle_meta* meta, nova_area_light* lights, uint32_t* tri_order) {
  const Built* b = static_cast<const Built*>(handle);
  if (bvh) std::memcpy(bvh, b->bvh.data(), b->bvh.size() * sizeof(nova_bvh_node));
  if (tris) std::memcpy(tris, b->tris.data(), b->tris.size() * sizeof(nova_triangle_data));
  if (meta) std::memcpy(meta, b->meta.data(), b->meta.size() * sizeof(nova_triangle_meta));
  if (lights && !b->lights.empty())
    std::memcpy(lights, b->lights.data(), b->lights.size() * sizeof(nova_area_light));
  if (tri_order) std::memcpy(tri_order, b->tri_order.data(), b->tri_order.size() * 4);
}

void nova_scene_free(void* handle) { delete static_cast<Built*>(handle); }

// Hooks for the known answers of the reference's own unit tests
// (tests/src/intersectables/aabb.cpp:7-64, triangle.cpp:8-21) on this builder's AABB / triangle
// helpers.  out = {surface area, cost(n), centre.xyz}
void nova_scene_test_aabb(const float* top, const float* bottom, uint64_t n, float* out) {
  Box b { { top[0], top[1], top[2] }, { bottom[0], bottom[1], bottom[2] } };
  out[0] = b.surface_area();
  out[1] = b.cost(n);
  V3 c = b.center();
  out[2] = c.x; out[3] = c.y; out[4] = c.z;
}
// out = grown box {top.xyz, bottom.xyz}
void nova_scene_test_aabb_grow(const float* top_a, const float* bottom_a, const float* top_b,
                               const float* bottom_b, float* out) {
  Box a { { top_a[0], top_a[1], top_a[2] }, { bottom_a[0], bottom_a[1], bottom_a[2] } };
  Box b { { top_b[0], top_b[1], top_b[2] }, { bottom_b[0], bottom_b[1], bottom_b[2] } };
  a.grow(b);
  out[0] = a.top.x; out[1] = a.top.y; out[2] = a.top.z;
  out[3] = a.bottom.x; out[4] = a.bottom.y; out[5] = a.bottom.z;
}
// out = bounds {top.xyz, bottom.xyz} of the triangle v[0..8]
void nova_scene_test_tri_bounds(const float* v, float* out) {
  Tri t { { v[0], v[1], v[2] }, { v[3], v[4], v[5] }, { v[6], v[7], v[8] } };
  Box b = t.bounds();
  out[0] = b.top.x; out[1] = b.top.y; out[2] = b.top.z;
  out[3] = b.bottom.x; out[4] = b.bottom.y; out[5] = b.bottom.z;
}

// Layout self-check for the Python bindings.
void nova_scene_struct_sizes(int32_t* out) {
  out[0] = sizeof(nova_bvh_node);
  out[1] = sizeof(nova_triangle_data);
  out[2] = sizeof(nova_triangle_meta);
  out[3] = sizeof(nova_area_light);
  out[4] = sizeof(nova_scene_params);
  out[5] = sizeof(nova_eye_coords);
}

}
