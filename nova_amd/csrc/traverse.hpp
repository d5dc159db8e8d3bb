// BVH traversal + Woop ray/triangle intersection for gfx950: the replacement for
// find_intersection / intersects_aabb / intersects_triangle
// (reference kernels/raytrace.hpp:38-102, kernels/intersection.hpp:14-52).
//
// Layout (built by Packer in bvh_pack.hpp from the reference's FlatBVHNode[]):
//   * one 64-byte, 64-byte-aligned Node64 per INNER node holding BOTH children's boxes, so one
//     cache line fetched per step yields two slab tests (the reference fetches a 32-byte node per
//     box test, dependent on the previous one);
//   * child reference: bit 31 = leaf; leaf: bits 24-30 = triangle count - 1, bits 0-23 = first
//     triangle (the reference stores indices as floats, so they are < 2^24 anyway); inner: index
//     of the child's Node64.  Inner nodes are renumbered densely in DFS order;
//   * triangles stay as the reference's 48-byte Woop matrices in leaf order.
//
// Execution: persistent work-groups pull 64-ray batches from a global queue head (one atomic
// per wave per batch); one ray per lane; the traversal stack lives in LDS as [entry][lane]
// (bank = lane mod 32: conflict-free for ds_read/write_b32), never in scratch memory.
//
// Semantics: the boxes tested are exactly the reference's (same operations, same order of
// operands), so the set of triangles reachable is the reference's.  Closest hit = min t over
// reachable triangles, ties -> lowest triangle index (the reference visits triangles in
// increasing index order with a strict `t < length`).  CULL=false is the reference's own walk
// (fixed order, no culling) and is kept as the validation mode.  CULL=true visits the nearer child
// first and skips subtrees wholly behind the origin or beyond the current hit.  That cull is made
// conservative with respect to the Woop ARITHMETIC, not the ideal geometry: Nova's Woop matrix
// [e1 e2 n-v1 v1]^-1 (src/intersectables/intersectable_manager.cpp:137-142) is singular where
// n.v1 = 1, so real scenes carry matrices with entries ~1e7 whose float evaluation accepts hits
// well outside the triangle's box.  At upload k_leaf_slack measures, for every triangle, how far
// the set of points its test can accept reaches beyond its leaf box: the triangle the float
// matrix really encodes (its inverse, in double) plus the rounding of the barycentric
// evaluation.  Each node stores e = max of that over its subtree (+ the slab test's own
// rounding); a box is culled against [0, best t] only after being widened by e on every side
// (tmin - e*max|1/d|, tmax + e*max|1/d|), so every hit the reference accepts is still visited.
#pragma once

#include "device_math.hpp"

namespace nova_hip {

struct __attribute__((aligned(64))) Node64 {
  float4 a; // child0 box max.xyz, w = child0 reference (bits)
  float4 b; // child0 box min.xyz, w = child1 reference (bits)
  float4 c; // child1 box max.xyz, w = cull slack e of this node's subtree (world units)
  float4 d; // child1 box min.xyz, w = unused
};

// Four-wide node (built at upload by collapsing the box-pair tree, see WideBuilder in bvh_pack.hpp):
// child k's box is component k of lo[axis] / hi[axis].  An unused slot has a NaN box (never hit).
// Why collapsing is exact: for a ray whose 1/d and -o/d are finite, the slab bounds
// fma(c, inv, nio) are monotone in the box coordinate c, so a box contained in another has its
// [tmin, tmax] contained too: "leaf box pierced" implies "every ancestor box pierced", and the set
// of triangles the reference reaches is exactly the set of leaves whose OWN box the line pierces —
// whatever hierarchy is put above them.  Rays with a non-finite 1/d or -o/d (a zero direction
// component) take the box-pair path, which tests every ancestor like the reference; trees whose
// boxes are not nested are not collapsed at all and are walked exhaustively (CULL = false).
struct __attribute__((aligned(128))) Node128 {
  float4 lo[3];
  float4 hi[3];
  uint32_t ref[4]; // child reference as in Node64; kEmptyRef for an unused slot
  float e[4];      // cull slack of the child's subtree (world units, slab rounding included)
};

struct LeafRec {
  uint32_t first, count;
  float top[3], bot[3];
};

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kEmptyRef = 0xFFFFFFFFu;
constexpr uint32_t kMaxLeafTris = 128;
constexpr float kCullRel = 0x1p-20f;          // relative slack on the current hit distance
constexpr float kEvalRounding = 0x1p-20f;     // bound on the fma-chain rounding (gamma_4 = 2^-22, x4)
constexpr float kSlabRoundingSlack = 0x1p-19f; // slab test rounding, as a fraction of the scene radius

struct TraverseArgs {
  const Node64* nodes;
  const float4* tris; // 3 x float4 per triangle
  float root_top[3];
  float root_bot[3];
  const float4* ray_o;      // {point.xyz, tmax}; tmax < 0 marks an empty slot
  const float4* ray_d;      // {direction.xyz, flags}; bit 0 = any-hit
  float4* hits;             // {t, u, v, triangle index}
  const uint32_t* n_items_ptr; // number of active items, read on the device (or null)
  uint32_t n_items_imm;
  uint32_t kinds;  // work items = n_items * kinds; item w -> slot (w / n) * stride + (w % n)
  uint32_t stride;
  uint32_t* queue_head;
  float inv_radius;                  // 1 / largest |coordinate| of the root box
  unsigned long long* work_counters; // [0] 64-byte node lines fetched, [1] triangle tests (COUNT only)
  const Node128* wide;               // four-wide tree, or null when the boxes are not nested
  uint32_t* spill;                   // stack entries beyond the LDS part: [entry][thread of the grid]
  uint32_t n_tris;
  uint32_t rays_per_wave;            // k_traverse_p: largest number of rays a wave takes from the queue at a time
  uint32_t flags;                    // bits 0-1 = RayQueue mode, bit 2 = ray kinds interleaved
  const uint32_t* perm;              // k_traverse_p: queue position w takes ray perm[w] (rays binned by origin cell and
                                     // direction octant, k_ray_keys), or null
  const uint32_t* n_extra_ptr;       // further work items after the n_items * kinds ones: *n_extra_ptr rays in slots
                                     // kinds * stride + i (the MIS rays of a frame that survive k_shade's light
                                     // pre-test, compacted), or null
};

// work items of a launch and the ray slot of item w
NV_DEV uint32_t items_total(const TraverseArgs& A, uint32_t n) {
  return n * A.kinds + (A.n_extra_ptr ? *A.n_extra_ptr : 0u); // the host guarantees n * (kinds + 1) < 2^31
}
NV_DEV size_t item_slot(const TraverseArgs& A, uint32_t n, uint32_t w, bool interleave) {
  const uint32_t main_items = n * A.kinds;
  if (w >= main_items) return static_cast<size_t>(A.kinds) * A.stride + (w - main_items);
  uint32_t kind = w / n;
  uint32_t j = w - kind * n;
  if (interleave) { // the rays of one path next to each other
    j = w / A.kinds;
    kind = w - j * A.kinds;
  }
  return static_cast<size_t>(kind) * A.stride + j;
}

// ---- ray binning for incoherent batches -------------------------------------------------------------
// A batch of random rays over a scene that does not fit the caches (BASELINE config 4: 0.6 GB of nodes and
// triangles) is bound by HBM: every ray pulls its own lines.  Rays that start in the same cell of a 32^3 grid
// over the root box and point into the same octant walk much the same nodes, so the queue is read through a
// permutation that puts them next to each other (counting sort on an 18-bit key: Morton code of the cell, then
// the octant).  Which ray a lane traces does not change the ray's answer; hits go to the ray's own slot.
constexpr uint32_t kRayKeyBits = 18;
__device__ __forceinline__ uint32_t spread5(uint32_t v) { // 5 bits -> every third bit
  v = (v | (v << 8)) & 0x100Fu;
  v = (v | (v << 4)) & 0x10C3u;
  v = (v | (v << 2)) & 0x1249u;
  return v;
}
__global__ __launch_bounds__(256) void k_ray_keys(const float4* __restrict__ ray_o, const float4* __restrict__ ray_d, uint32_t n,
                                                  float bx, float by, float bz, float sx, float sy, float sz,
                                                  uint32_t* __restrict__ keys, uint32_t* __restrict__ hist) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 o = ray_o[i], d = ray_d[i];
  const float fx = (o.x - bx) * sx, fy = (o.y - by) * sy, fz = (o.z - bz) * sz;
  const uint32_t cx = fx > 0.0f ? min(static_cast<uint32_t>(fminf(fx, 31.0f)), 31u) : 0u;
  const uint32_t cy = fy > 0.0f ? min(static_cast<uint32_t>(fminf(fy, 31.0f)), 31u) : 0u;
  const uint32_t cz = fz > 0.0f ? min(static_cast<uint32_t>(fminf(fz, 31.0f)), 31u) : 0u;
  const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
  const uint32_t key = ((spread5(cx) | (spread5(cy) << 1) | (spread5(cz) << 2)) << 3) | oct;
  keys[i] = key;
  atomicAdd(&hist[key], 1u);
}
__global__ __launch_bounds__(256) void k_ray_scatter(const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ cursor,
                                                     uint32_t* __restrict__ perm) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  perm[atomicAdd(&cursor[keys[i]], 1u)] = i;
}

// Per-leaf cull slack (see the header comment).  For a ray with |origin| <= R and a hit inside
// the scene box, the barycentric u = ox + t*dx is evaluated with absolute error
// <= kEvalRounding * (3R * sum|m.x_i| + |m.x.w|) (same for v); an error du moves the accepted
// point by du * |e1'|, e1' being the edge the matrix really encodes.  Rays starting farther away
// scale the slack by |origin|/R in trace_one.
__global__ __launch_bounds__(256) void k_leaf_slack(const float4* tris, const LeafRec* leaves,
                                                    uint32_t n_leaves, float radius, float* out) {
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_leaves) return;
  const LeafRec L = leaves[l];
  const double inf = __builtin_huge_val();
  double worst = 0.0;
  for (uint32_t i = L.first; i < L.first + L.count; i++) {
    const float4 mx = tris[3 * static_cast<size_t>(i)], my = tris[3 * static_cast<size_t>(i) + 1],
                 mz = tris[3 * static_cast<size_t>(i) + 2];
    const double a[3][3] = { { mx.x, mx.y, mx.z }, { my.x, my.y, my.z }, { mz.x, mz.y, mz.z } };
    const double w[3] = { mx.w, my.w, mz.w };
    // inverse of the 3x3 part by cofactors
    double c[3][3];
    c[0][0] = a[1][1] * a[2][2] - a[1][2] * a[2][1];
    c[0][1] = a[1][2] * a[2][0] - a[1][0] * a[2][2];
    c[0][2] = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    c[1][0] = a[0][2] * a[2][1] - a[0][1] * a[2][2];
    c[1][1] = a[0][0] * a[2][2] - a[0][2] * a[2][0];
    c[1][2] = a[0][1] * a[2][0] - a[0][0] * a[2][1];
    c[2][0] = a[0][1] * a[1][2] - a[0][2] * a[1][1];
    c[2][1] = a[0][2] * a[1][0] - a[0][0] * a[1][2];
    c[2][2] = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    const double det = a[0][0] * c[0][0] + a[0][1] * c[0][1] + a[0][2] * c[0][2];
    double slack = inf;
    if (det != 0.0 && det == det && __builtin_fabs(det) < inf) {
      const double id = 1.0 / det;
      // inv[r][k] = c[k][r] / det; column k of the inverse = image of unit vector k
      double e1[3], e2[3], v1[3];
      for (int r = 0; r < 3; r++) {
        e1[r] = c[0][r] * id;
        e2[r] = c[1][r] * id;
        v1[r] = -(c[0][r] * w[0] + c[1][r] * w[1] + c[2][r] * w[2]) * id;
      }
      double stick = 0.0;
      for (int r = 0; r < 3; r++) {
        const double p[3] = { v1[r], v1[r] + e1[r], v1[r] + e2[r] };
        for (int k = 0; k < 3; k++) {
          stick = __builtin_fmax(stick, p[k] - static_cast<double>(L.top[r]));
          stick = __builtin_fmax(stick, static_cast<double>(L.bot[r]) - p[k]);
        }
      }
      const double le1 = __builtin_sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
      const double le2 = __builtin_sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
      const double r3 = 3.0 * static_cast<double>(radius);
      const double du = kEvalRounding * (r3 * (__builtin_fabs(a[0][0]) + __builtin_fabs(a[0][1]) + __builtin_fabs(a[0][2])) + __builtin_fabs(w[0]));
      const double dv = kEvalRounding * (r3 * (__builtin_fabs(a[1][0]) + __builtin_fabs(a[1][1]) + __builtin_fabs(a[1][2])) + __builtin_fabs(w[1]));
      slack = stick + du * le1 + dv * le2;
      if (!(slack == slack)) slack = inf;
    }
    worst = __builtin_fmax(worst, slack);
  }
  out[l] = worst < 3.0e38 ? static_cast<float>(worst) * 1.0000002f + 1e-30f : __builtin_huge_valf();
}

struct RayState {
  V3 o, d, inv, nio;
};

NV_DEV bool slab(const RayState& r, float4 top, float4 bot, float& tmin, float& tmax) {
  float t1x = fma_(top.x, r.inv.x, r.nio.x), t1y = fma_(top.y, r.inv.y, r.nio.y), t1z = fma_(top.z, r.inv.z, r.nio.z);
  float t2x = fma_(bot.x, r.inv.x, r.nio.x), t2y = fma_(bot.y, r.inv.y, r.nio.y), t2z = fma_(bot.z, r.inv.z, r.nio.z);
  tmin = max_(min_(t1x, t2x), max_(min_(t1y, t2y), min_(t1z, t2z)));
  tmax = min_(max_(t1x, t2x), min_(max_(t1y, t2y), max_(t1z, t2z)));
  return tmin <= tmax;
}

struct HitState {
  float t, u, v;
  int id;
};

typedef float f2 __attribute__((ext_vector_type(2)));
NV_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); } // v_pk_fma_f32: two correctly rounded fmas

// kernels/intersection.hpp:14-37.  Returns true when the hit was accepted.
NV_DEV bool woop(const RayState& r, HitState& h, int index, float4 mx, float4 my, float4 mz) {
  float oz = fma_(mz.x, r.o.x, fma_(mz.y, r.o.y, fma_(mz.z, r.o.z, mz.w)));
  float dz = fma_(mz.x, r.d.x, fma_(mz.y, r.d.y, mz.z * r.d.z));
  float t = -oz / dz;
  if (t < 0.0f || !__builtin_isfinite(t)) return false;
  if (t >= h.t) {
    if (!(t == h.t && h.id >= 0 && index < h.id)) return false;
  }
  float ox = fma_(mx.x, r.o.x, fma_(mx.y, r.o.y, fma_(mx.z, r.o.z, mx.w)));
  float oy = fma_(my.x, r.o.x, fma_(my.y, r.o.y, fma_(my.z, r.o.z, my.w)));
  float dx = fma_(mx.x, r.d.x, fma_(mx.y, r.d.y, mx.z * r.d.z));
  float dy = fma_(my.x, r.d.x, fma_(my.y, r.d.y, my.z * r.d.z));
  float u = fma_(t, dx, ox);
  float v = fma_(t, dy, oy);
  float w = (1.0f - u) - v;
  if (w < 0.0f || u < 0.0f || v < 0.0f) return false;
  h.t = t;
  h.u = u;
  h.v = v;
  h.id = index;
  return true;
}

template <int STACK, bool CULL, bool COUNT>
__device__ __forceinline__ void trace_one(const TraverseArgs& A, uint32_t (*stk)[64], uint32_t lane,
                                          float4 o4, float4 d4, HitState& h, uint32_t& n_nodes,
                                          uint32_t& n_tris) {
  RayState r;
  r.d = xyz(d4);
  r.o = fma3(r.d, kRayEpsilon, xyz(o4)); // kernels/types.hpp:16-20
  r.inv = { 1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z };
  r.nio = (-r.o) * r.inv;
  const bool anyhit = (__float_as_uint(d4.w) & 1u) != 0;

  float tmin, tmax;
  const float4 rt = { A.root_top[0], A.root_top[1], A.root_top[2], 0.0f };
  const float4 rb = { A.root_bot[0], A.root_bot[1], A.root_bot[2], 0.0f };
  if (!slab(r, rt, rb, tmin, tmax)) return;

  // slack multiplier of this ray: max |1/d| (how far a slab bound moves when the box is widened
  // by one unit), scaled up for origins outside the scene radius
  const float o_inf = max_(__builtin_fabsf(r.o.x), max_(__builtin_fabsf(r.o.y), __builtin_fabsf(r.o.z)));
  const float slack_scale = max_(__builtin_fabsf(r.inv.x), max_(__builtin_fabsf(r.inv.y), __builtin_fabsf(r.inv.z))) *
                            max_(1.0f, o_inf * A.inv_radius);
  float far_limit = fma_(__builtin_fabsf(h.t), kCullRel, h.t);
  uint32_t sp = 0;
  uint32_t cur = 0;
  while (true) {
    if (!(cur & kLeafBit)) {
      const Node64* nd = A.nodes + cur;
      const float4 na = nd->a, nb = nd->b, nc = nd->c, ndd = nd->d;
      if (COUNT) n_nodes++;
      float tmin0, tmax0, tmin1, tmax1;
      bool h0 = slab(r, na, nb, tmin0, tmax0);
      bool h1 = slab(r, nc, ndd, tmin1, tmax1);
      if (CULL) {
        const float es = nc.w * slack_scale;
        h0 = h0 && (tmax0 + es >= 0.0f) && (tmin0 - es <= far_limit);
        h1 = h1 && (tmax1 + es >= 0.0f) && (tmin1 - es <= far_limit);
      }
      const uint32_t ref0 = __float_as_uint(na.w), ref1 = __float_as_uint(nb.w);
      if (h0 && h1) {
        const bool swap = CULL && (tmin1 < tmin0);
        stk[sp++][lane] = swap ? ref0 : ref1;
        cur = swap ? ref1 : ref0;
        continue;
      }
      if (h0) { cur = ref0; continue; }
      if (h1) { cur = ref1; continue; }
    } else {
      const uint32_t first = cur & 0x00FFFFFFu;
      const uint32_t count = ((cur >> 24) & 0x7Fu) + 1u;
      bool done = false;
      for (uint32_t i = first; i < first + count; i++) {
        const float4* m = A.tris + 3 * static_cast<size_t>(i);
        if (COUNT) n_tris++;
        if (woop(r, h, static_cast<int>(i), m[0], m[1], m[2])) {
          if (anyhit) { done = true; break; }
          far_limit = fma_(h.t, kCullRel, h.t);
        }
      }
      if (done) return;
    }
    if (sp == 0) return;
    cur = stk[--sp][lane];
  }
}

constexpr uint32_t kDone = 0x7FFFFFFFu; // `cur` of a lane without a ray (never a node index)

// Ray queue of one launch, as seen by one wave (all members wave-uniform).  The queue is split into
// kSubQueues heads, one per 128-byte line: 8192 waves taking 64 rays at a time from ONE head
// serialise on that address (~12-25 ns per atomic, measured: the camera-ray launch of a 1080p frame
// spent ~0.5 ms there).  Sub-queue k owns the contiguous ray range [k * per, (k + 1) * per); a wave
// starts at sub-queue (wave id mod kSubQueues) and moves on when it is exhausted, so the tail stays
// balanced and neighbouring rays stay together.
constexpr uint32_t kSubQueues = 64;
constexpr uint32_t kSubQueueStride = 32; // u32 words between heads (128 bytes)

struct RayQueue {
  uint32_t total, grab, per, q, q_tried, w_next, w_end;
  uint32_t nq, mode, xcd, wave_seq;
  bool dry;
  // Sub-queue tried after `t` failures.  mode 0: neighbours of the wave's own number.  The other
  // modes use that work-groups are dealt round-robin over the 8 XCDs (XCD = blockIdx.x % 8), so that
  // each XCD's L2 sees the rays of one part of the ray array (= of the frame: rays are stored in
  // pixel-tile order): 1 = sub-queues k with k % 8 == XCD first; 2 = the 8 consecutive sub-queues
  // 8 * XCD .. 8 * XCD + 7 first; 3 = one sub-queue per XCD (8 heads in all).
  NV_DEV uint32_t pick(uint32_t t) const {
    switch (mode) {
      case 1: return (xcd + 8u * ((wave_seq + t) & 7u) + (t >> 3)) % kSubQueues;
      case 2: return 8u * ((xcd + (t >> 3)) & 7u) + ((wave_seq + t) & 7u);
      case 3: return (xcd + t) & 7u;
      default: return (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) + t) % kSubQueues;
    }
  }
  NV_DEV void init(uint32_t total_, uint32_t grab_, uint32_t mode_ = 0) {
    total = total_;
    grab = grab_;
    mode = mode_;
    nq = mode == 3 ? 8u : kSubQueues;
    per = (((total + nq - 1u) / nq) + 63u) & ~63u;
    xcd = blockIdx.x & 7u;
    wave_seq = (blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // (tried, no effect: the waves of a work-group, or of a CU by HW_ID, sharing one sub-queue)
    q_tried = 0;
    q = pick(0);
    w_next = w_end = 0;
    dry = (total == 0);
  }
  // makes [w_next, w_end) non-empty, or sets `dry`.  (Tried and dropped: one vector load that shows
  // all 64 heads at once, to learn "drained" after one round trip instead of up to 64 — 6-15 %
  // slower on the 1080p frame: the head lines are hot in L2 and every wave reading all of them
  // serialises there.)
  NV_DEV void fill_window(uint32_t* heads, uint32_t lane) {
    if (dry || w_next < w_end) return;
    while (q_tried < nq) {
      const uint32_t lo = q * per;
      if (lo < total) {
        const uint32_t hi = min(lo + per, total);
        // a plain (L2) look first: an exhausted sub-queue costs a load, not an atomic
        uint32_t got = __hip_atomic_load(heads + q * kSubQueueStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        got = __builtin_amdgcn_readfirstlane(got);
        if (got < hi - lo) {
          if (lane == 0) got = atomicAdd(heads + q * kSubQueueStride, grab);
          got = __builtin_amdgcn_readfirstlane(got);
        }
        if (got < hi - lo) {
          w_next = lo + got;
          w_end = min(w_next + grab, hi);
          return;
        }
      }
      q_tried++;
      q = pick(q_tried);
    }
    dry = true;
  }
};

// ---------------------------------------------------------------------------------------------
// Four-wide traversal (production).  Stack: SLDS entries per lane in LDS as [entry][lane], the
// (rare) rest in a global spill array laid out [entry][thread] so that a wave's spill is coalesced.
// ---------------------------------------------------------------------------------------------
template <int SLDS>
struct HybridStack {
  uint32_t (*lds)[64];
  uint32_t* spill; // this thread's column
  uint32_t stride;
  uint32_t lane;
  uint32_t sp = 0;
  NV_DEV void push(uint32_t v) {
    if (sp < SLDS) lds[sp][lane] = v;
    else spill[static_cast<size_t>(sp - SLDS) * stride] = v;
    sp++;
  }
  NV_DEV uint32_t pop() {
    --sp;
    return sp < SLDS ? lds[sp][lane] : spill[static_cast<size_t>(sp - SLDS) * stride];
  }
  // The two below are for an LDS part of SLDS + 1 rows (k_traverse_p): row SLDS takes the writes that belong in
  // the spill area, so that the common case is straight-line LDS code.  (pop() above compiles to ONE flat load
  // from a selected address — LDS through the vector-memory path, in the dependent chain of every step.)
  NV_DEV uint32_t pop_lds() {
    --sp;
    uint32_t v = lds[min(sp, static_cast<uint32_t>(SLDS))][lane];
    asm volatile("" : "+v"(v)); // keeps this a ds_read: without it the two loads are merged into one flat load again
    if (sp >= SLDS) v = spill[static_cast<size_t>(sp - SLDS) * stride];
    return v;
  }
  // push c, b, a (in this order) where n > 3, n > 2, n > 1: entries above the top of the stack are free, so the
  // three LDS writes need no predicate, only the stack pointer does
  NV_DEV void push_upto3(uint32_t n, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t sp0 = sp;
    lds[min(sp, static_cast<uint32_t>(SLDS))][lane] = c;
    sp += n > 3u ? 1u : 0u;
    lds[min(sp, static_cast<uint32_t>(SLDS))][lane] = b;
    sp += n > 2u ? 1u : 0u;
    lds[min(sp, static_cast<uint32_t>(SLDS))][lane] = a;
    sp += n > 1u ? 1u : 0u;
    if (sp > SLDS) { // some of them belong in the spill area (rare)
      uint32_t q = sp0;
      if (n > 3u) { if (q >= SLDS) spill[static_cast<size_t>(q - SLDS) * stride] = c; q++; }
      if (n > 2u) { if (q >= SLDS) spill[static_cast<size_t>(q - SLDS) * stride] = b; q++; }
      if (n > 1u) { if (q >= SLDS) spill[static_cast<size_t>(q - SLDS) * stride] = a; q++; }
    }
  }
};

NV_DEV void ray_setup(float4 o4, float4 d4, RayState& r) {
  r.d = xyz(d4);
  r.o = fma3(r.d, kRayEpsilon, xyz(o4)); // kernels/types.hpp:16-20
  r.inv = { 1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z };
  r.nio = (-r.o) * r.inv;
}

NV_DEV void cswap(float& ka, uint32_t& ra, float& kb, uint32_t& rb) {
  const bool s = kb < ka;
  const float k0 = s ? kb : ka, k1 = s ? ka : kb;
  const uint32_t r0 = s ? rb : ra, r1 = s ? ra : rb;
  ka = k0; kb = k1; ra = r0; rb = r1;
}

NV_DEV void leaf_tests(const TraverseArgs& A, const RayState& r, HitState& h, uint32_t ref, bool anyhit,
                       float& far_limit, bool& done, uint32_t& n_tris) {
  const uint32_t first = ref & 0x00FFFFFFu;
  const uint32_t end = first + ((ref >> 24) & 0x7Fu) + 1u;
  // two triangles in flight per step (the second index is clamped so that the load stays inside the
  // array; its test is skipped when it is past the leaf)
  for (uint32_t i = first; i < end; i += 2) {
    const uint32_t i1 = min(i + 1u, A.n_tris - 1u);
    const float4* m = A.tris + 3 * static_cast<size_t>(i);
    const float4* q = A.tris + 3 * static_cast<size_t>(i1);
    const float4 m0 = m[0], m1 = m[1], m2 = m[2], q0 = q[0], q1 = q[1], q2 = q[2];
    n_tris++;
    if (woop(r, h, static_cast<int>(i), m0, m1, m2)) {
      if (anyhit) { done = true; return; }
      far_limit = fma_(h.t, kCullRel, h.t);
    }
    if (i + 1u < end) {
      n_tris++;
      if (woop(r, h, static_cast<int>(i + 1u), q0, q1, q2)) {
        if (anyhit) { done = true; return; }
        far_limit = fma_(h.t, kCullRel, h.t);
      }
    }
  }
}

template <int SLDS, bool COUNT>
__device__ __forceinline__ void trace_wide(const TraverseArgs& A, HybridStack<SLDS>& st, const RayState& r,
                                           bool anyhit, HitState& h, uint32_t& n_nodes, uint32_t& n_tris) {
  float tmin, tmax;
  const float4 rt = { A.root_top[0], A.root_top[1], A.root_top[2], 0.0f };
  const float4 rb = { A.root_bot[0], A.root_bot[1], A.root_bot[2], 0.0f };
  if (!slab(r, rt, rb, tmin, tmax)) return;
  const float o_inf = max_(__builtin_fabsf(r.o.x), max_(__builtin_fabsf(r.o.y), __builtin_fabsf(r.o.z)));
  const float slack_scale = max_(__builtin_fabsf(r.inv.x), max_(__builtin_fabsf(r.inv.y), __builtin_fabsf(r.inv.z))) *
                            max_(1.0f, o_inf * A.inv_radius);
  float far_limit = fma_(__builtin_fabsf(h.t), kCullRel, h.t);
  const float inf = __builtin_huge_valf();
  st.sp = 0;
  uint32_t cur = 0;
  while (true) {
    if (!(cur & kLeafBit)) {
      const Node128* nd = A.wide + cur;
      const float4 lx = nd->lo[0], ly = nd->lo[1], lz = nd->lo[2];
      const float4 hx = nd->hi[0], hy = nd->hi[1], hz = nd->hi[2];
      const uint4 rf = *reinterpret_cast<const uint4*>(nd->ref);
      const float4 es = *reinterpret_cast<const float4*>(nd->e);
      if (COUNT) n_nodes += 2;
      float key[4];
      uint32_t ref[4] = { rf.x, rf.y, rf.z, rf.w };
      const float lxs[4] = { lx.x, lx.y, lx.z, lx.w }, lys[4] = { ly.x, ly.y, ly.z, ly.w }, lzs[4] = { lz.x, lz.y, lz.z, lz.w };
      const float hxs[4] = { hx.x, hx.y, hx.z, hx.w }, hys[4] = { hy.x, hy.y, hy.z, hy.w }, hzs[4] = { hz.x, hz.y, hz.z, hz.w };
      const float ess[4] = { es.x, es.y, es.z, es.w };
      uint32_t n_hit = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // kernels/intersection.hpp:40-52 on child k's box (t1 from top, t2 from bottom)
        const float t1x = fma_(hxs[k], r.inv.x, r.nio.x), t1y = fma_(hys[k], r.inv.y, r.nio.y), t1z = fma_(hzs[k], r.inv.z, r.nio.z);
        const float t2x = fma_(lxs[k], r.inv.x, r.nio.x), t2y = fma_(lys[k], r.inv.y, r.nio.y), t2z = fma_(lzs[k], r.inv.z, r.nio.z);
        const float tn = max_(min_(t1x, t2x), max_(min_(t1y, t2y), min_(t1z, t2z)));
        const float tf = min_(max_(t1x, t2x), min_(max_(t1y, t2y), max_(t1z, t2z)));
        const float e = ess[k] * slack_scale;
        const bool hit = (tn <= tf) && (tf + e >= 0.0f) && (tn - e <= far_limit);
        key[k] = hit ? min_(tn, 3.402823466e38f) : inf; // misses sort after every hit
        n_hit += hit ? 1u : 0u;
      }
      if (n_hit == 0) {
        if (st.sp == 0) return;
        cur = st.pop();
        continue;
      }
      if (n_hit > 1) {
        cswap(key[0], ref[0], key[1], ref[1]);
        cswap(key[2], ref[2], key[3], ref[3]);
        cswap(key[0], ref[0], key[2], ref[2]);
        cswap(key[1], ref[1], key[3], ref[3]);
        cswap(key[1], ref[1], key[2], ref[2]);
        if (n_hit > 3) st.push(ref[3]);
        if (n_hit > 2) st.push(ref[2]);
        st.push(ref[1]);
        cur = ref[0];
      } else {
        cur = key[0] < inf ? ref[0] : key[1] < inf ? ref[1] : key[2] < inf ? ref[2] : ref[3];
      }
      continue;
    }
    bool done = false;
    uint32_t nt = 0;
    leaf_tests(A, r, h, cur, anyhit, far_limit, done, nt);
    if (COUNT) n_tris += nt;
    if (done || st.sp == 0) return;
    cur = st.pop();
  }
}

// Box-pair traversal on the hybrid stack: the path of rays with a non-finite 1/d or -o/d.
// Same steps as trace_one<.., CULL = true, ..>.
template <int SLDS, bool COUNT>
__device__ __forceinline__ void trace_pair(const TraverseArgs& A, HybridStack<SLDS>& st, const RayState& r,
                                           bool anyhit, HitState& h, uint32_t& n_nodes, uint32_t& n_tris) {
  float tmin, tmax;
  const float4 rt = { A.root_top[0], A.root_top[1], A.root_top[2], 0.0f };
  const float4 rb = { A.root_bot[0], A.root_bot[1], A.root_bot[2], 0.0f };
  if (!slab(r, rt, rb, tmin, tmax)) return;
  const float o_inf = max_(__builtin_fabsf(r.o.x), max_(__builtin_fabsf(r.o.y), __builtin_fabsf(r.o.z)));
  const float slack_scale = max_(__builtin_fabsf(r.inv.x), max_(__builtin_fabsf(r.inv.y), __builtin_fabsf(r.inv.z))) *
                            max_(1.0f, o_inf * A.inv_radius);
  float far_limit = fma_(__builtin_fabsf(h.t), kCullRel, h.t);
  st.sp = 0;
  uint32_t cur = 0;
  while (true) {
    if (!(cur & kLeafBit)) {
      const Node64* nd = A.nodes + cur;
      const float4 na = nd->a, nb = nd->b, nc = nd->c, ndd = nd->d;
      if (COUNT) n_nodes++;
      float tmin0, tmax0, tmin1, tmax1;
      bool h0 = slab(r, na, nb, tmin0, tmax0);
      bool h1 = slab(r, nc, ndd, tmin1, tmax1);
      const float es = nc.w * slack_scale;
      h0 = h0 && (tmax0 + es >= 0.0f) && (tmin0 - es <= far_limit);
      h1 = h1 && (tmax1 + es >= 0.0f) && (tmin1 - es <= far_limit);
      const uint32_t ref0 = __float_as_uint(na.w), ref1 = __float_as_uint(nb.w);
      if (h0 && h1) {
        const bool swap = tmin1 < tmin0;
        st.push(swap ? ref0 : ref1);
        cur = swap ? ref1 : ref0;
        continue;
      }
      if (h0) { cur = ref0; continue; }
      if (h1) { cur = ref1; continue; }
    } else {
      bool done = false;
      uint32_t nt = 0;
      leaf_tests(A, r, h, cur, anyhit, far_limit, done, nt);
      if (COUNT) n_tris += nt;
      if (done) return;
    }
    if (st.sp == 0) return;
    cur = st.pop();
  }
}

template <int SLDS, bool COUNT>
__global__ __launch_bounds__(256) void k_traverse4(TraverseArgs A) {
  __shared__ uint32_t lds_stack[4][SLDS][64];
  const uint32_t lane = threadIdx.x & 63u;
  HybridStack<SLDS> st;
  st.lds = lds_stack[threadIdx.x >> 6];
  st.lane = lane;
  st.stride = gridDim.x * blockDim.x;
  st.spill = A.spill + (blockIdx.x * blockDim.x + threadIdx.x);
  const uint32_t n = A.n_items_ptr ? *A.n_items_ptr : A.n_items_imm;
  const uint32_t total = items_total(A, n);
  uint32_t n_nodes = 0, n_tris = 0;

  RayQueue Q;
  Q.init(total, 64u);
  while (true) {
    Q.fill_window(A.queue_head, lane);
    if (Q.dry) break;
    const uint32_t w = Q.w_next + lane;
    const uint32_t w_end = Q.w_end;
    Q.w_next = Q.w_end;
    if (w < w_end) {
      const size_t slot = item_slot(A, n, w, false);
      const float4 o4 = A.ray_o[slot];
      const float4 d4 = A.ray_d[slot];
      HitState h { o4.w, 0.0f, 0.0f, -1 };
      if (!(o4.w < 0.0f)) {
        RayState r;
        ray_setup(o4, d4, r);
        const bool anyhit = (__float_as_uint(d4.w) & 1u) != 0;
        // finite 1/d and -o/d: x * 0 is 0 for finite x, NaN otherwise
        const float probe = fma_(r.inv.x, 0.0f, fma_(r.inv.y, 0.0f, r.inv.z * 0.0f)) +
                            fma_(r.nio.x, 0.0f, fma_(r.nio.y, 0.0f, r.nio.z * 0.0f));
        if (A.wide != nullptr && probe == 0.0f) trace_wide<SLDS, COUNT>(A, st, r, anyhit, h, n_nodes, n_tris);
        else trace_pair<SLDS, COUNT>(A, st, r, anyhit, h, n_nodes, n_tris);
      }
      A.hits[slot] = make_float4(h.t, h.u, h.v, __int_as_float(h.id));
    }
  }
  if (COUNT) {
    unsigned long long a = n_nodes, b = n_tris;
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_down(a, off);
      b += __shfl_down(b, off);
    }
    if (lane == 0) {
      atomicAdd(&A.work_counters[0], a);
      atomicAdd(&A.work_counters[1], b);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_traverse_p (production): persistent lanes on the four-wide tree.
//
// What bounds traversal on MI355X (rocprofv3 counters): every vector-memory
// wave-instruction occupies the CU's texture-data path for ~16-25 cycles and every VALU
// wave-instruction its SIMD for ~4, however few lanes are active — and a 64-ray batch that runs
// until its longest ray ends keeps only ~10 of 64 lanes busy.  So the unit of cost is the
// wave-instruction and the lever is lanes per instruction:
//   * a lane whose ray has ended is refilled from the wave's window of the ray queue as soon as
//     REFILL_MIN lanes are idle (ray set-up costs three divisions for the whole wave);
//   * each loop iteration executes ONE phase for the whole wave — a four-wide node step or a
//     three-triangle leaf step — chosen by majority among the lanes, so neither phase runs for a few
//     stragglers while the rest of the wave waits;
//   * the ray queue is split into kSubQueues heads (one 128-byte line each) with 64-ray grabs, so
//     that the queue atomics of 8192 waves do not serialise on one address and the tail stays
//     balanced (a wave whose sub-queue is empty moves on to the next one).
// Per-ray arithmetic is the same as in trace_wide; rays with a non-finite 1/d or -o/d are traced at
// once on the box-pair tree by their lane (rare).
// ---------------------------------------------------------------------------------------------
#ifndef NOVA_TRIS_PER_STEP
#define NOVA_TRIS_PER_STEP 3 // measured on MI355X: 2 -> 3 is 3-9 % faster, 4 needs > 96 VGPRs at 5 waves/SIMD
#endif
constexpr int kTrisPerStep = NOVA_TRIS_PER_STEP;

NV_DEV unsigned long long hit_key(const HitState& h) {
  // (t, triangle index) ordered like the reference's "closer, ties -> lower index"; t + 0 turns -0 into +0
  return h.id >= 0 ? (static_cast<unsigned long long>(__float_as_uint(h.t + 0.0f)) << 32) | static_cast<uint32_t>(h.id)
                   : ~0ull;
}

template <int SLDS, bool COUNT, int REFILL_MIN, int MIN_WAVES, bool HELP>
__global__ __launch_bounds__(256, MIN_WAVES) void k_traverse_p(TraverseArgs A) {
  constexpr bool FORK = true; // drain-mode forking (a build without it: 4.73 -> 4.02 ms per sample in round 1)
  constexpr int TPS = kTrisPerStep;
  __shared__ uint32_t lds_stack[4][SLDS + 1][64]; // row SLDS: see HybridStack::push_upto3
  // FORK: records of the rays that were split among several lanes of the wave (index = the lane
  // that loaded the ray): best (t, index) key, its t bits / u / v, members still running
  __shared__ unsigned long long rec_key_[4][64];
  __shared__ float rec_t_[4][64], rec_u_[4][64], rec_v_[4][64];
  __shared__ uint32_t rec_pending_[4][64], xch_[4][64], xch2_[4][HELP ? 64 : 1];
  unsigned long long* rec_key = rec_key_[threadIdx.x >> 6];
  float* rec_t = rec_t_[threadIdx.x >> 6];
  float* rec_u = rec_u_[threadIdx.x >> 6];
  float* rec_v = rec_v_[threadIdx.x >> 6];
  uint32_t* rec_pending = rec_pending_[threadIdx.x >> 6];
  uint32_t* xch = xch_[threadIdx.x >> 6];
  uint32_t* xch2 = xch2_[threadIdx.x >> 6];
  const uint32_t lane = threadIdx.x & 63u;
  HybridStack<SLDS> st;
  st.lds = lds_stack[threadIdx.x >> 6];
  st.lane = lane;
  st.stride = gridDim.x * blockDim.x;
  st.spill = A.spill + (blockIdx.x * blockDim.x + threadIdx.x);
  const uint32_t n = A.n_items_ptr ? *A.n_items_ptr : A.n_items_imm;
  const uint32_t total = items_total(A, n);
  uint32_t n_nodes = 0, n_tris = 0;
  const float4 rt = { A.root_top[0], A.root_top[1], A.root_top[2], 0.0f };
  const float4 rb = { A.root_bot[0], A.root_bot[1], A.root_bot[2], 0.0f };
  const float inf = __builtin_huge_valf();

  // A launch with few rays is latency-bound (the chip is nearly empty and a wave lasts as long as
  // its longest ray), so the grab shrinks until every wave of the grid gets some rays.
  RayQueue Q;
  Q.init(total, min(A.rays_per_wave, max(2u, total / (gridDim.x * 4u))), A.flags & 3u);
  const bool interleave = (A.flags & 4u) != 0;

  RayState r = {};
  HitState h = { 0.0f, 0.0f, 0.0f, -1 };
  float slack_scale = 0.0f, far_limit = 0.0f;
  float o_scale = 1.0f; // max(1, |origin| / scene radius): the slack is per unit of that (k_leaf_slack)
  bool anyhit = false;
  bool pair = false; // this lane's ray walks the box-pair tree (non-finite 1/d or -o/d)
  uint32_t cur = kDone;
  size_t slot = 0;
  st.sp = 0;
  bool fin = false;    // this lane's ray ended in this iteration
  bool forked = false; // this lane works on a ray shared with other lanes (record rec_*[fam])
  uint32_t fam = lane;

  while (true) {
    const uint64_t idle = __ballot(cur == kDone);
    if (!Q.dry && __popcll(idle) >= REFILL_MIN) {
      Q.fill_window(A.queue_head, lane);
      if (!Q.dry) {
        const uint32_t take = min(static_cast<uint32_t>(__popcll(idle)), Q.w_end - Q.w_next);
        const uint32_t my_rank = rank_in_mask(idle);
        if (cur == kDone && my_rank < take) {
          const uint32_t w = Q.w_next + my_rank;
          slot = item_slot(A, n, w, interleave);
          if (A.perm) slot = A.perm[w];
          const float4 o4 = A.ray_o[slot];
          const float4 d4 = A.ray_d[slot];
          h = { o4.w, 0.0f, 0.0f, -1 };
          forked = false;
          bool live = !(o4.w < 0.0f);
          if (live) {
            ray_setup(o4, d4, r);
            anyhit = (__float_as_uint(d4.w) & 1u) != 0;
            const float probe = fma_(r.inv.x, 0.0f, fma_(r.inv.y, 0.0f, r.inv.z * 0.0f)) +
                                fma_(r.nio.x, 0.0f, fma_(r.nio.y, 0.0f, r.nio.z * 0.0f));
            pair = !(probe == 0.0f);
            float tmin, tmax;
            live = slab(r, rt, rb, tmin, tmax);
          }
          if (live) {
            const float o_inf = max_(__builtin_fabsf(r.o.x), max_(__builtin_fabsf(r.o.y), __builtin_fabsf(r.o.z)));
            slack_scale = max_(__builtin_fabsf(r.inv.x), max_(__builtin_fabsf(r.inv.y), __builtin_fabsf(r.inv.z))) *
                          max_(1.0f, o_inf * A.inv_radius);
            o_scale = max_(1.0f, o_inf * A.inv_radius);
            far_limit = fma_(__builtin_fabsf(h.t), kCullRel, h.t);
            st.sp = 0;
            cur = 0;
          } else {
            A.hits[slot] = make_float4(h.t, h.u, h.v, __int_as_float(h.id));
          }
        }
        Q.w_next += take;
      }
      continue;
    }
    if (FORK && Q.dry && idle != 0) {
      // Drain: the queue is empty, so an idle lane can only help.  It takes the top stack entry (the
      // subtree its owner would visit next) of a running lane together with a copy of the ray and
      // of the best hit so far, and walks it as a ray of its own; the members of such a family
      // merge their results by (t, index) in the family record, the last one to finish writes the
      // hit.  The order of visits changes, the set of accepted hits' minimum does not.
      const bool can_give = cur != kDone && !pair && st.sp >= 1;
      const uint64_t donors = __ballot(can_give);
      const uint32_t nf = min(static_cast<uint32_t>(__popcll(idle)), static_cast<uint32_t>(__popcll(donors)));
      if (nf > 0) {
        const bool is_donor = can_give && rank_in_mask(donors) < nf;
        const bool is_helper = cur == kDone && rank_in_mask(idle) < nf;
        uint32_t give = 0;
        if (is_donor) {
          give = st.pop();
          if (!forked) {
            fam = lane;
            rec_key[lane] = hit_key(h);
            rec_t[lane] = h.t; rec_u[lane] = h.u; rec_v[lane] = h.v;
            rec_pending[lane] = 1;
            forked = true;
          }
          xch[rank_in_mask(donors)] = lane;
        }
        if (is_donor) atomicAdd(&rec_pending[fam], 1u);
        uint32_t src = lane;
        if (is_helper) src = xch[rank_in_mask(idle)];
        give = __shfl(give, src);
        r.o.x = __shfl(r.o.x, src); r.o.y = __shfl(r.o.y, src); r.o.z = __shfl(r.o.z, src);
        r.d.x = __shfl(r.d.x, src); r.d.y = __shfl(r.d.y, src); r.d.z = __shfl(r.d.z, src);
        r.inv.x = __shfl(r.inv.x, src); r.inv.y = __shfl(r.inv.y, src); r.inv.z = __shfl(r.inv.z, src);
        r.nio.x = __shfl(r.nio.x, src); r.nio.y = __shfl(r.nio.y, src); r.nio.z = __shfl(r.nio.z, src);
        h.t = __shfl(h.t, src); h.u = __shfl(h.u, src); h.v = __shfl(h.v, src); h.id = __shfl(h.id, src);
        slack_scale = __shfl(slack_scale, src);
        o_scale = __shfl(o_scale, src);
        far_limit = __shfl(far_limit, src);
        anyhit = __shfl(anyhit ? 1 : 0, src) != 0;
        fam = __shfl(fam, src);
        const uint32_t slot_lo = __shfl(static_cast<uint32_t>(slot), src);
        const uint32_t slot_hi = __shfl(static_cast<uint32_t>(static_cast<unsigned long long>(slot) >> 32), src);
        slot = static_cast<size_t>((static_cast<unsigned long long>(slot_hi) << 32) | slot_lo);
        if (is_helper) {
          cur = give;
          st.sp = 0;
          pair = false;
          forked = true;
        }
      }
    }
    const uint64_t m_node = __ballot(cur < kDone);
    const uint64_t m_leaf = __ballot((cur & kLeafBit) != 0);
    if ((m_node | m_leaf) == 0) {
      if (Q.dry) break;
      continue; // cannot happen with REFILL_MIN <= 64; kept as a guard
    }
    if (__popcll(m_node) >= __popcll(m_leaf)) {
      if (cur < kDone && pair) {
        // box-pair step (trace_pair's), for the rare rays the four-wide tree is not exact for
        const Node64* nd = A.nodes + cur;
        const float4 na = nd->a, nb = nd->b, nc = nd->c, ndd = nd->d;
        if (COUNT) n_nodes++;
        float tmin0, tmax0, tmin1, tmax1;
        bool h0 = slab(r, na, nb, tmin0, tmax0);
        bool h1 = slab(r, nc, ndd, tmin1, tmax1);
        const float es = nc.w * slack_scale;
        h0 = h0 && (tmax0 + es >= 0.0f) && (tmin0 - es <= far_limit);
        h1 = h1 && (tmax1 + es >= 0.0f) && (tmin1 - es <= far_limit);
        const uint32_t ref0 = __float_as_uint(na.w), ref1 = __float_as_uint(nb.w);
        if (h0 && h1) {
          const bool swap = tmin1 < tmin0;
          st.push(swap ? ref0 : ref1);
          cur = swap ? ref1 : ref0;
        } else if (h0) {
          cur = ref0;
        } else if (h1) {
          cur = ref1;
        } else if (st.sp != 0) {
          cur = st.pop();
        } else {
          fin = true;
        }
      } else if (cur < kDone) {
        if (FORK && forked) { // another member of the family may have found something closer
          const float ft = __uint_as_float(static_cast<uint32_t>(rec_key[fam] >> 32));
          if (ft < h.t) far_limit = min_(far_limit, fma_(ft, kCullRel, ft));
        }
        // The planes the ray meets first / last on each axis are read by the sign of 1/d (bottom when 1/d >= 0),
        // not found by min / max of both bounds: fma(c, 1/d, -o/d) is monotone in c and every box of a collapsed
        // tree has bottom <= top (Packer::ordered), so these ARE the reference's min(t1, t2) and max(t1, t2) —
        // 24 min / max fewer per step.  Children are taken in pairs so that the bounds are packed fmas.
        const char* nb = reinterpret_cast<const char*>(A.wide);
        const uint32_t base = cur << 7; // < 2^32: at most 2^24 nodes
        const uint32_t kx = (__float_as_uint(r.inv.x) >> 31) * 48u, ky = (__float_as_uint(r.inv.y) >> 31) * 48u,
                       kz = (__float_as_uint(r.inv.z) >> 31) * 48u;
        const float4 nx = *reinterpret_cast<const float4*>(nb + (base + kx));
        const float4 ny = *reinterpret_cast<const float4*>(nb + (base + 16u + ky));
        const float4 nz = *reinterpret_cast<const float4*>(nb + (base + 32u + kz));
        const float4 fx = *reinterpret_cast<const float4*>(nb + (base + 48u - kx));
        const float4 fy = *reinterpret_cast<const float4*>(nb + (base + 64u - ky));
        const float4 fz = *reinterpret_cast<const float4*>(nb + (base + 80u - kz));
        const uint4 rf = *reinterpret_cast<const uint4*>(nb + (base + 96u));
        const float4 es = *reinterpret_cast<const float4*>(nb + (base + 112u));
        if (COUNT) n_nodes += 2;
        float key[4];
        uint32_t ref[4] = { rf.x, rf.y, rf.z, rf.w };
        uint32_t n_hit = 0;
        // Near side of the cull: "the box widened by the child's slack e lies behind the origin".  The ray leaves
        // that box at min over the axes of far_a + e |1/d_a| — its slab test — which is far tighter than
        // far + e max |1/d| for a ray with one small direction component, and boxes just behind the origin
        // are what a bounce ray meets first: 4 % fewer node visits and 11 % fewer triangle tests on the
        // benchmark frame.  (The far side keeps the one-product bound: it decides 1.7 % of the visits, and the same
        // treatment there measured 0.6 %.)
        const f2 ix = { r.inv.x, r.inv.x }, iy = { r.inv.y, r.inv.y }, iz = { r.inv.z, r.inv.z };
        const f2 ox = { r.nio.x, r.nio.x }, oy = { r.nio.y, r.nio.y }, oz = { r.nio.z, r.nio.z };
        const float axs = __builtin_fabsf(r.inv.x) * o_scale, ays = __builtin_fabsf(r.inv.y) * o_scale, azs = __builtin_fabsf(r.inv.z) * o_scale;
        const f2 ax = { axs, axs }, ay = { ays, ays }, az = { azs, azs }, ss = { slack_scale, slack_scale };
#pragma unroll
        for (int p = 0; p < 2; p++) {
          const f2 e2 = p == 0 ? f2 { es.x, es.y } : f2 { es.z, es.w };
          const f2 tnx = fma2(p == 0 ? f2 { nx.x, nx.y } : f2 { nx.z, nx.w }, ix, ox);
          const f2 tny = fma2(p == 0 ? f2 { ny.x, ny.y } : f2 { ny.z, ny.w }, iy, oy);
          const f2 tnz = fma2(p == 0 ? f2 { nz.x, nz.y } : f2 { nz.z, nz.w }, iz, oz);
          const f2 tfx = fma2(p == 0 ? f2 { fx.x, fx.y } : f2 { fx.z, fx.w }, ix, ox);
          const f2 tfy = fma2(p == 0 ? f2 { fy.x, fy.y } : f2 { fy.z, fy.w }, iy, oy);
          const f2 tfz = fma2(p == 0 ? f2 { fz.x, fz.y } : f2 { fz.z, fz.w }, iz, oz);
          const f2 wx = fma2(e2, ax, tfx), wy = fma2(e2, ay, tfy), wz = fma2(e2, az, tfz);
          const f2 lim = e2 * ss;
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const int k = 2 * p + q;
            const float tn = max_(tnx[q], max_(tny[q], tnz[q]));
            const float tf = min_(tfx[q], min_(tfy[q], tfz[q]));
            const float near = min_(wx[q], min_(wy[q], wz[q]));
            const bool hit = (tn <= tf) && (near >= 0.0f) && (tn - lim[q] <= far_limit);
            key[k] = hit ? min_(tn, 3.402823466e38f) : inf;
            n_hit += hit ? 1u : 0u;
          }
        }
        cswap(key[0], ref[0], key[1], ref[1]);
        cswap(key[2], ref[2], key[3], ref[3]);
        cswap(key[0], ref[0], key[2], ref[2]);
        cswap(key[1], ref[1], key[3], ref[3]);
        cswap(key[1], ref[1], key[2], ref[2]);
        st.push_upto3(n_hit, ref[1], ref[2], ref[3]);
        if (n_hit > 0) {
          cur = ref[0];
        } else if (st.sp != 0) {
          cur = st.pop_lds();
        } else {
          fin = true;
        }
      }
    } else {
      // Leaf step with helpers.  A leaf phase runs with a quarter of the lanes (the rest want a node step or wait for
      // rays) and a leaf of a reference tree holds six triangles on average, i.e. two or three steps.  So a lane that
      // is not in a leaf takes the NEXT TPS triangles of a lane that is: it borrows that ray's origin, direction and
      // best hit, tests them, and hands back what it accepted; the owner merges by (t, index) — the same minimum the
      // in-order tests find — and moves on by 2 x TPS triangles.  (A second helper for the third TPS triangles
      // measured the same: 3.497 vs 3.498 ms.)
      if (HELP) {
        const bool is_leaf = (cur & kLeafBit) != 0;
        const bool want = is_leaf && ((cur >> 24) & 0x7Fu) >= static_cast<uint32_t>(TPS); // more than TPS triangles left
        const uint64_t m_want = __ballot(want), m_free = ~m_leaf;
        const uint32_t nh = min(static_cast<uint32_t>(__popcll(m_want)), static_cast<uint32_t>(__popcll(m_free)));
        const bool donor = want && rank_in_mask(m_want) < nh;
        const bool helper = !is_leaf && rank_in_mask(m_free) < nh;
        uint32_t src = lane, back = lane;
        if (nh > 0) { // (a copy of the plain leaf step for the steps nobody wants help in measured slower: 3.54 vs 3.50 ms)
          if (donor) xch[rank_in_mask(m_want)] = lane;
          if (helper) xch2[rank_in_mask(m_free)] = lane;
          if (helper) src = xch[rank_in_mask(m_free)];
          if (donor) back = xch2[rank_in_mask(m_want)];
        }
        const uint32_t lcur = __shfl(cur, src); // (through LDS beside the lane number instead: no faster)
        bool found = false, finished = false;
        const uint32_t first0 = lcur & 0x00FFFFFFu, count0 = ((lcur >> 24) & 0x7Fu) + 1u;
        const uint32_t first = first0 + (helper ? TPS : 0), count = count0 - (helper ? TPS : 0);
        // the ray this lane tests triangles for (its own unless it helps)
        RayState re;
        re.o = { __shfl(r.o.x, src), __shfl(r.o.y, src), __shfl(r.o.z, src) };
        re.d = { __shfl(r.d.x, src), __shfl(r.d.y, src), __shfl(r.d.z, src) };
        HitState he = { __shfl(h.t, src), h.u, h.v, __shfl(h.id, src) };
        if (is_leaf || helper) {
          float4 tm[TPS][3];
          const uint32_t off0 = __umul24(first, 48u), last_off = (A.n_tris - 1u) * 48u;
          uint32_t off[TPS];
#pragma unroll
          for (int k = 0; k < TPS; k++) {
            off[k] = min(off0 + 48u * k, last_off);
            asm volatile("" : "+v"(off[k]));
          }
#pragma unroll
          for (int k = 0; k < TPS; k++) {
            const float4* m = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(A.tris) + off[k]);
            tm[k][0] = m[0]; tm[k][1] = m[1]; tm[k][2] = m[2];
          }
#pragma unroll
          for (int k = 0; k < TPS; k++) {
            if (!finished && count > static_cast<uint32_t>(k)) {
              if (COUNT) n_tris++;
              if (woop(re, he, static_cast<int>(first + k), tm[k][0], tm[k][1], tm[k][2])) {
                found = true;
                finished = anyhit && !helper; // (a helper does not know whether its ray is an any-hit ray: it goes on)
              }
            }
          }
        }
        // what the helper accepted, seen from the owner's lane
        const float bt = __shfl(he.t, back), bu = __shfl(he.u, back), bv = __shfl(he.v, back);
        const int bid = __shfl(he.id, back);
        const bool bfound = __shfl(found ? 1 : 0, back) != 0;
        if (is_leaf) {
          h = he;
          if (donor && bfound && (h.id < 0 || bt < h.t || (bt == h.t && bid < h.id))) {
            h.t = bt; h.u = bu; h.v = bv; h.id = bid;
            found = true;
          }
          if (donor && bfound && anyhit) finished = true;
          if (found) far_limit = fma_(h.t, kCullRel, h.t);
          if (FORK && forked && h.id >= 0) atomicMin(&rec_key[fam], hit_key(h));
          const uint32_t used = donor ? 2u * TPS : static_cast<uint32_t>(TPS);
          if (!finished && count0 > used) {
            cur = kLeafBit | ((count0 - used - 1u) << 24) | (first0 + used);
          } else if (!finished && st.sp != 0) {
            cur = st.pop_lds();
          } else {
            fin = true;
          }
        }
      } else
      if (cur & kLeafBit) {
        // TPS triangles of the leaf per step (all loads in flight together; indices past
        // the leaf are clamped into the array and their tests skipped); the leaf's remaining range
        // is kept in `cur`
        const uint32_t first = cur & 0x00FFFFFFu;
        const uint32_t count = ((cur >> 24) & 0x7Fu) + 1u;
        bool finished = false;
        float4 tm[TPS][3];
        // byte offsets (n_tris <= 2^24, so they fit 32 bits and the one product is a 24-bit multiply)
        const uint32_t off0 = __umul24(first, 48u), last_off = (A.n_tris - 1u) * 48u;
        uint32_t off[TPS];
#pragma unroll
        for (int k = 0; k < TPS; k++) {
          off[k] = min(off0 + 48u * k, last_off);
          asm volatile("" : "+v"(off[k])); // every offset before the first load: a wait for an older load that the
        }                                  // scheduler puts between the loads would serialise them
#pragma unroll
        for (int k = 0; k < TPS; k++) {
          const float4* m = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(A.tris) + off[k]);
          // (non-temporal loads here were 1.45x slower: the three rows of a triangle share a line,
          // and without L1 allocation each row is its own L2 request)
          tm[k][0] = m[0]; tm[k][1] = m[1]; tm[k][2] = m[2];
        }
#pragma unroll
        for (int k = 0; k < TPS; k++) {
          if (!finished && count > static_cast<uint32_t>(k)) {
            if (COUNT) n_tris++;
            if (woop(r, h, static_cast<int>(first + k), tm[k][0], tm[k][1], tm[k][2])) {
              finished = anyhit;
              far_limit = fma_(h.t, kCullRel, h.t);
            }
          }
        }
        if (FORK && forked && h.id >= 0) atomicMin(&rec_key[fam], hit_key(h));
        if (!finished && count > static_cast<uint32_t>(TPS)) {
          cur = kLeafBit | ((count - TPS - 1u) << 24) | (first + TPS);
        } else if (!finished && st.sp != 0) {
          cur = st.pop_lds();
        } else {
          fin = true;
        }
      }
    }
    if (fin) {
      if (!FORK || !forked) {
        A.hits[slot] = make_float4(h.t, h.u, h.v, __int_as_float(h.id));
      } else {
        // LDS operations of one wave execute in program order, each for all its lanes: every member
        // finishing now has merged its key before any of them reads the record back
        const unsigned long long key = hit_key(h);
        if (h.id >= 0) atomicMin(&rec_key[fam], key);
        unsigned long long fk = rec_key[fam];
        if (h.id >= 0 && fk == key) {
          rec_t[fam] = h.t; rec_u[fam] = h.u; rec_v[fam] = h.v;
        }
        const uint32_t before = atomicSub(&rec_pending[fam], 1u);
        if (before == 1u) { // last member: the record is final
          fk = rec_key[fam];
          if (fk == ~0ull) A.hits[slot] = make_float4(h.t, 0.0f, 0.0f, __int_as_float(-1));
          else A.hits[slot] = make_float4(rec_t[fam], rec_u[fam], rec_v[fam], __int_as_float(static_cast<int>(fk & 0xFFFFFFFFull)));
        }
        forked = false;
      }
      cur = kDone;
      fin = false;
    }
  }
  if (COUNT) {
    unsigned long long a = n_nodes, b = n_tris;
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_down(a, off);
      b += __shfl_down(b, off);
    }
    if (lane == 0) {
      atomicAdd(&A.work_counters[0], a);
      atomicAdd(&A.work_counters[1], b);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_traverse_packet: camera rays.  A 64-ray batch is one 8x8-pixel block (k_generate), so its rays
// walk almost the same nodes.  The per-lane kernels then load the same 64 bytes 64 times through the
// texture path, which is what bounds them (5.1 vector-memory wave-instructions per camera ray,
// TD busy 67 %).  Here the WAVE walks the box-pair tree once for its 64 rays: the node and the
// triangles are fetched with wave-uniform addresses (scalar loads through the constant cache: no
// texture-path traffic at all), every lane tests its own ray against them, and a subtree is entered
// with the mask of lanes whose own test passed; the stack (one entry per level: reference + lane
// mask) is wave-uniform in LDS.  Each lane makes exactly the decisions of its own culled traversal
// (same slab test, same cull test against its own best hit, same triangle test), only the visiting
// order is the packet's (nearer child by majority), and the result does not depend on the order:
// min t, ties -> lowest triangle index.  Rays with non-finite 1/d need no special case here: the
// box-pair tree tests every ancestor like the reference.
// ---------------------------------------------------------------------------------------------
// NOVA_COUNT_STEPS (diagnostic builds only): the packet kernels count wave STEPS (node fetches, leaf rounds) instead
// of per-ray box and triangle tests
#ifdef NOVA_COUNT_STEPS
#define NOVA_STEP_COUNT_NODE(in) (lane == 0 ? 1u : 0u)
#define NOVA_STEP_COUNT_TRI(x) 0u
#define NOVA_STEP_COUNT_ROUND (lane == 0 ? 1u : 0u)
#else
#define NOVA_STEP_COUNT_NODE(in) ((in) ? 1u : 0u)
#define NOVA_STEP_COUNT_TRI(x) (x)
#define NOVA_STEP_COUNT_ROUND 0u
#endif
// (Round 3 measured the same packet on the FOUR-wide tree — two s_load_dwordx16 per step, four ballots, children
// entered nearest first: 19.9 instead of 37.7 node steps per packet on the benchmark frame, bit-identical hits, and
// the same 0.61 ms per launch at the same occupancy, slower at the 7 waves/SIMD its 105 scalar registers allow: the
// wide step issues ~250 instructions where two of these issue 2 x 110 and moves the same two lines.  The kernel is
// bound by each wave's own serial stream, not by the number of dependent loads.  profiles/r03_packet4_ab.txt; the
// code is in the repository's history, commit "packet4".)
constexpr int kPacketStack = 72; // box-pair tree depth <= 64 (checked at upload)

template <bool COUNT>
__global__ __launch_bounds__(256) void k_traverse_packet(TraverseArgs A, const Node64* __restrict__ nodes,
                                                         const float4* __restrict__ tris) {
  __shared__ uint32_t pstack[4][kPacketStack][4];
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t (*stk)[4] = pstack[threadIdx.x >> 6];
  const uint32_t n = A.n_items_ptr ? *A.n_items_ptr : A.n_items_imm;
  const uint32_t total = items_total(A, n);
  uint32_t n_nodes = 0, n_tris = 0;
  const float4 rt = { A.root_top[0], A.root_top[1], A.root_top[2], 0.0f };
  const float4 rb = { A.root_bot[0], A.root_bot[1], A.root_bot[2], 0.0f };

  RayQueue Q;
  Q.init(total, 64u, (A.flags & 8u) ? (A.flags & 3u) : 0u);
  while (true) {
    Q.fill_window(A.queue_head, lane);
    if (Q.dry) break;
    const uint32_t w = Q.w_next + lane;
    const bool valid = w < Q.w_end;
    Q.w_next = Q.w_end;
    size_t slot = 0;
    RayState r = {};
    HitState h = { 0.0f, 0.0f, 0.0f, -1 };
    bool alive = false, anyhit = false;
    float slack_scale = 0.0f, far_limit = 0.0f;
    if (valid) {
      slot = item_slot(A, n, w, false);
      const float4 o4 = A.ray_o[slot];
      const float4 d4 = A.ray_d[slot];
      h.t = o4.w;
      if (!(o4.w < 0.0f)) {
        ray_setup(o4, d4, r);
        anyhit = (__float_as_uint(d4.w) & 1u) != 0;
        float tmin, tmax;
        alive = slab(r, rt, rb, tmin, tmax);
        const float o_inf = max_(__builtin_fabsf(r.o.x), max_(__builtin_fabsf(r.o.y), __builtin_fabsf(r.o.z)));
        slack_scale = max_(__builtin_fabsf(r.inv.x), max_(__builtin_fabsf(r.inv.y), __builtin_fabsf(r.inv.z))) *
                      max_(1.0f, o_inf * A.inv_radius);
        far_limit = fma_(__builtin_fabsf(h.t), kCullRel, h.t);
      }
    }
    uint64_t act = __ballot(alive);
    uint32_t sp = 0;
    uint32_t cur = 0;
    while (act != 0) {
      const bool in = ((act >> lane) & 1ull) != 0;
      if (!(cur & kLeafBit)) {
        const Node64* nd = nodes + cur; // wave-uniform address: scalar loads
        const float4 na = nd->a, nb = nd->b, nc = nd->c, ndd = nd->d;
        if (COUNT) n_nodes += NOVA_STEP_COUNT_NODE(in);
        float tmin0, tmax0, tmin1, tmax1;
        bool h0 = slab(r, na, nb, tmin0, tmax0);
        bool h1 = slab(r, nc, ndd, tmin1, tmax1);
        const float es = nc.w * slack_scale;
        h0 = in && h0 && (tmax0 + es >= 0.0f) && (tmin0 - es <= far_limit);
        h1 = in && h1 && (tmax1 + es >= 0.0f) && (tmin1 - es <= far_limit);
        const uint64_t m0 = __ballot(h0), m1 = __ballot(h1);
        const uint32_t ref0 = __float_as_uint(na.w), ref1 = __float_as_uint(nb.w);
        if (m0 != 0 && m1 != 0) {
          const uint64_t both = m0 & m1;
          const uint64_t second_nearer = __ballot(h0 && h1 && tmin1 < tmin0);
          const bool swap = both != 0 ? (2 * __popcll(second_nearer) > __popcll(both)) : (__popcll(m1) > __popcll(m0));
          const uint32_t far_ref = swap ? ref0 : ref1;
          const uint64_t far_mask = swap ? m0 : m1;
          if (lane == 0) {
            stk[sp][0] = far_ref;
            stk[sp][1] = static_cast<uint32_t>(far_mask);
            stk[sp][2] = static_cast<uint32_t>(far_mask >> 32);
          }
          sp++;
          cur = swap ? ref1 : ref0;
          act = swap ? m1 : m0;
          continue;
        }
        if (m0 != 0) { cur = ref0; act = m0; continue; }
        if (m1 != 0) { cur = ref1; act = m1; continue; }
      } else {
        const uint32_t first = cur & 0x00FFFFFFu;
        const uint32_t count = ((cur >> 24) & 0x7Fu) + 1u;
        // two triangles per round: the rounds of a leaf are a chain of dependent scalar loads (the kernel's bound), and
        // the node's 16 scalar registers are free here.  (The node or the leaf staged through LDS by one vector load
        // instead measured 0.79-0.82 vs 0.74-0.78 ms and 0.628 vs 0.622 ms: no gain, removed.)
        const uint32_t end = first + count;
        for (uint32_t i = first; i < end; i += 2) {
          const float4* m = tris + 3 * static_cast<size_t>(i); // wave-uniform addresses
          const float4* q = tris + 3 * static_cast<size_t>(min(i + 1u, A.n_tris - 1u));
          const float4 m0 = m[0], m1 = m[1], m2 = m[2], q0 = q[0], q1 = q[1], q2 = q[2];
          if (COUNT) n_tris += NOVA_STEP_COUNT_ROUND;
          if (in && alive) {
            if (COUNT) n_tris += NOVA_STEP_COUNT_TRI(1u);
            if (woop(r, h, static_cast<int>(i), m0, m1, m2)) {
              if (anyhit) alive = false;
              far_limit = fma_(h.t, kCullRel, h.t);
            }
          }
          if (in && alive && i + 1u < end) {
            if (COUNT) n_tris += NOVA_STEP_COUNT_TRI(1u);
            if (woop(r, h, static_cast<int>(i + 1u), q0, q1, q2)) {
              if (anyhit) alive = false;
              far_limit = fma_(h.t, kCullRel, h.t);
            }
          }
        }
      }
      // pop the next subtree some lane still wants
      const uint64_t alive_mask = __ballot(alive);
      act = 0;
      while (sp != 0 && act == 0) {
        --sp;
        cur = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(stk[sp][0]));
        // readfirstlane returns int: go through uint32_t, or a mask with lane 31 set sign-extends into
        // lanes 32-63 (they would then test a popped LEAF's triangles without having passed its box —
        // found by scripts/gpu_soak.py: 1-3 pixels in 9 of 400 frames)
        const uint64_t mk = static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(stk[sp][1]))) |
                            (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(stk[sp][2]))) << 32);
        act = mk & alive_mask;
      }
    }
    if (valid) A.hits[slot] = make_float4(h.t, h.u, h.v, __int_as_float(h.id));
  }
  if (COUNT) {
    unsigned long long a = n_nodes, b = n_tris;
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_down(a, off);
      b += __shfl_down(b, off);
    }
    if (lane == 0) {
      atomicAdd(&A.work_counters[0], a);
      atomicAdd(&A.work_counters[1], b);
    }
  }
}

template <int STACK, bool CULL, bool COUNT>
__global__ __launch_bounds__(256) void k_traverse(TraverseArgs A) {
  __shared__ uint32_t lds_stack[4][STACK][64];
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t (*stk)[64] = lds_stack[threadIdx.x >> 6];
  const uint32_t n = A.n_items_ptr ? *A.n_items_ptr : A.n_items_imm;
  const uint32_t total = items_total(A, n);
  uint32_t n_nodes = 0, n_tris = 0;

  RayQueue Q;
  Q.init(total, 64u);
  while (true) {
    Q.fill_window(A.queue_head, lane);
    if (Q.dry) break;
    const uint32_t w = Q.w_next + lane;
    const uint32_t w_end = Q.w_end;
    Q.w_next = Q.w_end;
    if (w < w_end) {
      const size_t slot = item_slot(A, n, w, false);
      const float4 o4 = A.ray_o[slot];
      const float4 d4 = A.ray_d[slot];
      HitState h { o4.w, 0.0f, 0.0f, -1 };
      if (!(o4.w < 0.0f)) {
        trace_one<STACK, CULL, COUNT>(A, stk, lane, o4, d4, h, n_nodes, n_tris);
      }
      A.hits[slot] = make_float4(h.t, h.u, h.v, __int_as_float(h.id));
    }
  }
  if (COUNT) {
    // one atomic per wave per counter
    unsigned long long a = n_nodes, b = n_tris;
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_down(a, off);
      b += __shfl_down(b, off);
    }
    if (lane == 0) {
      atomicAdd(&A.work_counters[0], a);
      atomicAdd(&A.work_counters[1], b);
    }
  }
}

}
