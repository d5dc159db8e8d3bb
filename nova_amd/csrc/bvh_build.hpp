// GPU construction of Nova's BVH: the tree BVH::build makes (reference src/bvh/bvh.cpp:17-249 — binned
// SAH, bins = 1024/(depth+1), stable split on the bin index, DFS pre-order flattening with the
// larger-surface-area child first) and the Woop TriangleData of IntersectableManager::build
// (src/intersectables/intersectable_manager.cpp:129-145), built level by level on the device.
//
// The result is the SAME tree, node for node and triangle for triangle: every quantity the builder
// decides on is either an integer (bin counts), a min/max of input coordinates (boxes: exact in any
// order) or a float expression evaluated per node / per split with the operation order of the source
// (-ffp-contract=off, correctly rounded division and square root).  The only freedom is the sign of a
// zero box coordinate (std::max keeps the first of -0/+0, an atomic max keeps +0); it compares equal
// and no later result depends on it.
//
// Level-synchronous: all nodes of one depth are binned, evaluated and split together.
//   k_bin_sah    one wave per work item = up to `chunk` consecutive triangle positions of ONE node: bin
//                index per axis, counts and boxes accumulated in LDS.  A node that fits one item (all the
//                deep levels) is evaluated on the spot, bins never leaving LDS; a larger node's items
//                store their LDS bins to slabs that k_reduce_partials sums into the node's global bins
//                (no contended atomics per triangle)
//   k_sah_multi  one wave per (large node, axis): the same evaluation over the global bins
//                (evaluation = suffix boxes, prefix boxes and counts by wave scans, cost of every split
//                in its own lane, minimum with the later split winning a tie)
//   k_choose     one lane per node: best axis, child records
//   scan + k_scatter   stable partition of every splitting node's segment at once
// then sizes bottom-up, DFS indices top-down, and one pass that writes nodes, order and matrices.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>

#include "nova_hip.h"

namespace bvhb {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kMinTrianglesPerLeaf = 8; // src/constants.hpp:31
constexpr float kMaxBins = 1024.0f;          // src/constants.hpp:32
constexpr float kFltMax = 3.402823466e+38f;

struct BNode {
  float top[3], bot[3];
  uint32_t begin, end;
  uint32_t left, right; // child node ids; left == 0: leaf
  uint32_t slot;        // index among the nodes of its level that are being split, or kNone
  uint32_t size;        // nodes in the subtree
};

struct Bin { // all-zero = empty
  uint32_t count;
  uint32_t mx[3]; // enc(max)
  uint32_t mn[3]; // ~enc(min)
};

struct Slot {
  uint32_t node;
  uint32_t n;
  float start[3], inv[3];
  uint32_t axis_mask; // bit a: axis a is wide enough to be split (bvh.cpp:169)
  uint32_t multi;     // index of the node's global bins when it spans several work items, else kNone
  uint32_t part_base; // such a node's items write their bins to partial slabs part_base, part_base + 1, ...
};

struct AxisBest {
  float cost;
  uint32_t index, left_n;
  float ltop[3], lbot[3], rtop[3], rbot[3];
  int32_t axis; // -1: none
};

struct Choice {
  int32_t axis;
  uint32_t index, left_n, child; // child = id of the left child (right = child + 1)
};

// monotone float -> uint map: a < b  <=>  enc(a) < enc(b)
__device__ __forceinline__ uint32_t enc(float f) {
  uint32_t u = __float_as_uint(f);
  return u ^ ((u & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float dec(uint32_t e) {
  uint32_t u = e ^ ((e & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu);
  return __uint_as_float(u);
}
__device__ __forceinline__ float dec_max(uint32_t e) { return e == 0 ? -kFltMax : dec(e); }
__device__ __forceinline__ float dec_min(uint32_t e) { return e == 0 ? kFltMax : dec(~e); }
__device__ __forceinline__ void raise(uint32_t* p, uint32_t v) {
  if (*reinterpret_cast<volatile uint32_t*>(p) < v) atomicMax(p, v);
}

struct Box3 {
  float top[3], bot[3];
};
// AABB::grow (src/intersectables/aabb.hpp): max / min as std::max / std::min pick
__device__ __forceinline__ void grow(Box3& a, const Box3& b) {
  for (int i = 0; i < 3; i++) {
    a.top[i] = (a.top[i] < b.top[i]) ? b.top[i] : a.top[i];
    a.bot[i] = (b.bot[i] < a.bot[i]) ? b.bot[i] : a.bot[i];
  }
}
// AABB::get_surface_area / get_cost
__device__ __forceinline__ float surface_area(const float* top, const float* bot) {
  const float dx = top[0] - bot[0], dy = top[1] - bot[1], dz = top[2] - bot[2];
  return (dx * dy + dy * dz + dz * dx) * 2.0f;
}
__device__ __forceinline__ float cost(const float* top, const float* bot, uint32_t n) {
  return surface_area(top, bot) * static_cast<float>(n);
}
__device__ __forceinline__ Box3 load_bin(const Bin& b) {
  Box3 r;
  for (int i = 0; i < 3; i++) {
    r.top[i] = dec_max(b.mx[i]);
    r.bot[i] = dec_min(b.mn[i]);
  }
  return r;
}

// ---- triangle bounds and the root box -------------------------------------------------------------
// tbox[2 i] = (max xyz, -), tbox[2 i + 1] = (min xyz, -)   (Triangle::get_bounds)
__global__ void k_tri_bounds(const float* __restrict__ verts, uint32_t n, float4* __restrict__ tbox,
                             uint32_t* __restrict__ root /* 3 enc(max), 3 ~enc(min) */,
                             uint32_t* __restrict__ order, uint32_t* __restrict__ node_of) {
  uint32_t e[6] = { 0, 0, 0, 0, 0, 0 }; // enc(max) / ~enc(min): 0 = nothing
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = verts + 9ull * i;
    float top[3], bot[3];
    for (int a = 0; a < 3; a++) {
      const float v1 = p[a], v2 = p[3 + a], v3 = p[6 + a];
      const float m23 = (v2 < v3) ? v3 : v2, n23 = (v3 < v2) ? v3 : v2;
      top[a] = (v1 < m23) ? m23 : v1;
      bot[a] = (n23 < v1) ? n23 : v1;
    }
    tbox[2ull * i] = make_float4(top[0], top[1], top[2], 0.0f);
    tbox[2ull * i + 1] = make_float4(bot[0], bot[1], bot[2], 0.0f);
    order[i] = i;
    node_of[i] = 0;
    for (int a = 0; a < 3; a++) {
      const uint32_t et = enc(top[a]), eb = ~enc(bot[a]);
      e[a] = et > e[a] ? et : e[a];
      e[3 + a] = eb > e[3 + a] ? eb : e[3 + a];
    }
  }
  // one atomic per block and coordinate (all six words share a cache line: per-triangle atomics on them
  // cost more than the rest of the kernel)
  __shared__ uint32_t part[4][6];
  for (int a = 0; a < 6; a++) {
    uint32_t v = e[a];
    for (int d = 32; d > 0; d >>= 1) {
      const uint32_t o = __shfl_xor(v, d, 64);
      v = o > v ? o : v;
    }
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6][a] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    uint32_t v = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; w++) v = part[w][threadIdx.x] > v ? part[w][threadIdx.x] : v;
    if (v != 0) raise(root + threadIdx.x, v);
  }
}

// `lv` (where given): lv[d] .. lv[d + 1] are the nodes of level d — the level loop's ls / le kept on the device, so
// that the top levels can be enqueued without the host learning how many nodes each one has (build(): blind levels)
__global__ void k_root(const uint32_t* __restrict__ root, uint32_t n, BNode* __restrict__ nodes, uint32_t* __restrict__ lv) {
  lv[0] = 0;
  lv[1] = 1;
  BNode nd;
  for (int a = 0; a < 3; a++) {
    nd.top[a] = dec_max(root[a]);
    nd.bot[a] = dec_min(root[3 + a]);
  }
  nd.begin = 0;
  nd.end = n;
  nd.left = nd.right = 0;
  nd.slot = kNone;
  nd.size = 1;
  nodes[0] = nd;
}

// ---- exclusive scan of u32 (three passes; the block sums of up to 2^24 items fit one block's loop) ----
constexpr uint32_t kScanBlock = 256, kScanItems = 4, kScanTile = kScanBlock * kScanItems;
#ifndef NOVA_BVH_MIN_CHUNK
#define NOVA_BVH_MIN_CHUNK 1024
#endif
constexpr uint32_t kMinChunk = NOVA_BVH_MIN_CHUNK; // fewest positions of a k_bin_sah work item
constexpr uint32_t kFusedLevelNodes = 2048; // levels of at most this many nodes: k_level_setup / k_choose_scan

__device__ __forceinline__ uint32_t wave_inclusive(uint32_t v) {
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
// exclusive prefix of `v` over the block; *total = block sum.  blockDim.x == kScanBlock
__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wsum[kScanBlock / 64];
  const uint32_t inc = wave_inclusive(v);
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  __syncthreads(); // wsum may still be read by the previous call
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t base = 0, sum = 0;
  for (uint32_t i = 0; i < kScanBlock / 64; i++) {
    if (i < w) base += wsum[i];
    sum += wsum[i];
  }
  *total = sum;
  return base + inc - v;
}

__global__ void __launch_bounds__(kScanBlock) k_scan_reduce(const uint32_t* __restrict__ in, uint32_t n,
                                                             uint32_t* __restrict__ block_sums) {
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  uint32_t s = 0;
  for (uint32_t j = 0; j < kScanItems; j++)
    if (base + j < n) s += in[base + j];
  uint32_t total;
  (void) block_exclusive(s, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// one block: block_sums -> exclusive prefix in place; grand total to *total
__global__ void __launch_bounds__(kScanBlock) k_scan_sums(uint32_t* __restrict__ block_sums, uint32_t nb,
                                                           uint32_t* __restrict__ total) {
  constexpr uint32_t kPer = 16; // consecutive sums per thread and round (10 M triangles: 9766 sums, 3 rounds)
  uint32_t carry = 0;
  for (uint32_t start = 0; start < nb; start += kScanBlock * kPer) {
    const uint32_t base = start + threadIdx.x * kPer;
    uint32_t v[kPer], s = 0;
    for (uint32_t j = 0; j < kPer; j++) {
      v[j] = base + j < nb ? block_sums[base + j] : 0u;
      s += v[j];
    }
    uint32_t sum;
    uint32_t ex = carry + block_exclusive(s, &sum);
    for (uint32_t j = 0; j < kPer; j++) {
      if (base + j < nb) block_sums[base + j] = ex;
      ex += v[j];
    }
    carry += sum;
  }
  if (threadIdx.x == 0) *total = carry;
}

// the whole scan in one block, for the short arrays of the upper levels (one launch instead of three)
__global__ void __launch_bounds__(kScanBlock) k_scan_small(const uint32_t* __restrict__ in, uint32_t n,
                                                           uint32_t* __restrict__ out, uint32_t* __restrict__ total) {
  uint32_t carry = 0;
  for (uint32_t start = 0; start < n; start += kScanTile) {
    const uint32_t base = start + threadIdx.x * kScanItems;
    uint32_t v[kScanItems], s = 0;
    for (uint32_t j = 0; j < kScanItems; j++) {
      v[j] = (base + j < n) ? in[base + j] : 0u;
      s += v[j];
    }
    uint32_t sum;
    uint32_t ex = carry + block_exclusive(s, &sum);
    for (uint32_t j = 0; j < kScanItems; j++) {
      if (base + j < n) out[base + j] = ex;
      ex += v[j];
    }
    carry += sum;
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(kScanBlock) k_scan_apply(const uint32_t* __restrict__ in, uint32_t n,
                                                            const uint32_t* __restrict__ block_sums,
                                                            uint32_t* __restrict__ out) {
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  uint32_t v[kScanItems], s = 0;
  for (uint32_t j = 0; j < kScanItems; j++) {
    v[j] = (base + j < n) ? in[base + j] : 0u;
    s += v[j];
  }
  uint32_t total;
  uint32_t ex = block_exclusive(s, &total) + block_sums[blockIdx.x];
  for (uint32_t j = 0; j < kScanItems; j++) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}

// ---- one level ----------------------------------------------------------------------------------------
// flag[i] = 1 when node ls + i is to be split at this level (bvh.cpp:137-148)
// `small` = largest node (in triangles) left to k_subtree at this depth, 0 = none
__global__ void k_level_flags(const BNode* __restrict__ nodes, uint32_t ls, uint32_t m, uint32_t small,
                              uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const BNode& nd = nodes[ls + i];
  const uint32_t cnt = nd.end - nd.begin;
  flag[i] = (cnt > kMinTrianglesPerLeaf && cnt > small) ? 1u : 0u;
}

__global__ void k_slot_setup(BNode* __restrict__ nodes, uint32_t ls, uint32_t m, const uint32_t* __restrict__ flag,
                             const uint32_t* __restrict__ slot_of, uint32_t num_bins, uint32_t chunk,
                             Slot* __restrict__ slots, uint32_t* __restrict__ items,
                             uint32_t* __restrict__ multi_count, uint32_t* __restrict__ multi_list,
                             uint32_t* __restrict__ multi_items) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  BNode& nd = nodes[ls + i];
  if (!flag[i]) {
    nd.slot = kNone;
    return;
  }
  const uint32_t s = slot_of[i];
  nd.slot = s;
  Slot sl;
  sl.node = ls + i;
  sl.n = nd.end - nd.begin;
  sl.axis_mask = 0;
  const float fbins = static_cast<float>(num_bins);
  for (int a = 0; a < 3; a++) {
    const float step = (nd.top[a] - nd.bot[a]) / fbins; // bvh.cpp:152
    sl.start[a] = nd.bot[a];
    sl.inv[a] = 1.0f / step;                            // bvh.cpp:153
    if (!(fabsf(nd.top[a] - nd.bot[a]) < 1e-4f)) sl.axis_mask |= 1u << a; // bvh.cpp:169
  }
  sl.multi = sl.n > chunk ? atomicAdd(multi_count, 1u) : kNone; // which bins it gets does not matter
  const uint32_t n_items = (sl.n + chunk - 1) / chunk;
  sl.part_base = 0;
  if (sl.multi != kNone) {
    multi_list[sl.multi] = s;
    sl.part_base = atomicAdd(multi_items, n_items);
  }
  slots[s] = sl;
  items[s] = n_items;
}

// k_level_flags + scan + k_slot_setup + scan of the items in ONE work-group, for the levels of the top of the tree
// (a few thousand nodes at most): the six launches and two fills this replaces are ~25 us of a ~140 us level.
// Slots are numbered in node order as the scan of the flags numbers them.
// all_multi: every splitting node gets global bins and slabs, whatever its size (the two topmost levels: see build())
__global__ void __launch_bounds__(kScanBlock) k_level_setup(BNode* __restrict__ nodes, uint32_t ls, uint32_t m, uint32_t small,
                                                            uint32_t num_bins, uint32_t chunk, bool all_multi, Slot* __restrict__ slots,
                                                            uint32_t* __restrict__ item_base, uint32_t* __restrict__ n_slots_out,
                                                            uint32_t* __restrict__ n_items_out, uint32_t* __restrict__ multi_count,
                                                            uint32_t* __restrict__ multi_list, uint32_t* __restrict__ multi_items,
                                                            const uint32_t* __restrict__ lv) {
  if (lv) { // a blind level: the host passed bounds
    ls = lv[0];
    m = lv[1] - lv[0];
  }
  if (threadIdx.x == 0) {
    *multi_count = 0;
    *multi_items = 0;
  }
  __syncthreads();
  uint32_t slot_carry = 0, item_carry = 0;
  const float fbins = static_cast<float>(num_bins);
  for (uint32_t start = 0; start < m; start += kScanBlock) {
    const uint32_t i = start + threadIdx.x;
    uint32_t cnt = 0;
    if (i < m) cnt = nodes[ls + i].end - nodes[ls + i].begin;
    const bool take = i < m && cnt > kMinTrianglesPerLeaf && cnt > small;
    const uint32_t n_items = take ? (cnt + chunk - 1) / chunk : 0u;
    uint32_t slot_total, item_total;
    const uint32_t s = slot_carry + block_exclusive(take ? 1u : 0u, &slot_total);
    const uint32_t ib = item_carry + block_exclusive(n_items, &item_total);
    if (i < m) {
      BNode& nd = nodes[ls + i];
      if (!take) {
        nd.slot = kNone;
      } else {
        nd.slot = s;
        Slot sl;
        sl.node = ls + i;
        sl.n = cnt;
        sl.axis_mask = 0;
        for (int a = 0; a < 3; a++) {
          const float step = (nd.top[a] - nd.bot[a]) / fbins; // bvh.cpp:152
          sl.start[a] = nd.bot[a];
          sl.inv[a] = 1.0f / step;                            // bvh.cpp:153
          if (!(fabsf(nd.top[a] - nd.bot[a]) < 1e-4f)) sl.axis_mask |= 1u << a; // bvh.cpp:169
        }
        sl.multi = (cnt > chunk || all_multi) ? atomicAdd(multi_count, 1u) : kNone; // which bins it gets does not matter
        sl.part_base = 0;
        if (sl.multi != kNone) {
          multi_list[sl.multi] = s;
          sl.part_base = atomicAdd(multi_items, n_items);
        }
        slots[s] = sl;
        item_base[s] = ib;
      }
    }
    slot_carry += slot_total;
    item_carry += item_total;
  }
  if (threadIdx.x == 0) {
    *n_slots_out = slot_carry;
    *n_items_out = item_carry;
  }
}

// bvh.cpp:161-162: min(uvec3((center - bin_start) * inv_bin_step), num_splits)
__device__ __forceinline__ uint32_t bin_index(float c, float start, float inv, uint32_t num_splits) {
  const float f = (c - start) * inv;
  const uint32_t b = (f > 0.0f) ? static_cast<uint32_t>(fminf(f, 4.0e9f)) : 0u;
  return b < num_splits ? b : num_splits;
}

// ---- split evaluation by one wave (bvh.cpp:54-105) ---------------------------------------------------
// A count and a box as the bins hold them: max xyz as enc(), min xyz as ~enc() (so that both grow by an unsigned
// max), all-zero = empty.  The scans below stay in that form — one v_max_u32 with a DPP operand per value and
// step instead of move + compare + select on floats — and boxes are decoded only where a cost is computed.
// (enc() orders -0 below +0, std::max as AABB::grow uses it keeps whichever came first: a box coordinate that is
// a zero of both signs may come out with the other sign than on the host.  No comparison tells the two apart.)
struct CountBox {
  uint32_t n;
  uint32_t mx[3]; // enc(max)
  uint32_t mn[3]; // ~enc(min)
};
__device__ __forceinline__ CountBox empty_count_box() {
  CountBox c;
  c.n = 0;
  for (int i = 0; i < 3; i++) c.mx[i] = c.mn[i] = 0u;
  return c;
}
__device__ __forceinline__ void combine(CountBox& a, const CountBox& b) {
  a.n += b.n;
  for (int i = 0; i < 3; i++) {
    a.mx[i] = a.mx[i] < b.mx[i] ? b.mx[i] : a.mx[i];
    a.mn[i] = a.mn[i] < b.mn[i] ? b.mn[i] : a.mn[i];
  }
}
__device__ __forceinline__ Box3 decode(const CountBox& c) {
  Box3 r;
  for (int i = 0; i < 3; i++) {
    r.top[i] = dec_max(c.mx[i]);
    r.bot[i] = dec_min(c.mn[i]);
  }
  return r;
}
// Inclusive scan over the 64 lanes of a wave with data-parallel-primitive moves (no LDS traffic):
// row_shr 1, 2, 4, 8 inside each row of 16 lanes, then lane 15 of rows 0 and 2 into rows 1 and 3
// (row_bcast15), then lane 31 into rows 2 and 3 (row_bcast31).  A lane without a source gets the identity (0).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_step(CountBox& v) {
  CountBox o;
  o.n = dpp0<CTRL, ROW_MASK>(v.n);
  for (int j = 0; j < 3; j++) {
    o.mx[j] = dpp0<CTRL, ROW_MASK>(v.mx[j]);
    o.mn[j] = dpp0<CTRL, ROW_MASK>(v.mn[j]);
  }
  combine(v, o);
}
__device__ __forceinline__ void wave_scan(CountBox& v) {
  scan_step<0x111, 0xf>(v);
  scan_step<0x112, 0xf>(v);
  scan_step<0x114, 0xf>(v);
  scan_step<0x118, 0xf>(v);
  scan_step<0x142, 0xa>(v);
  scan_step<0x143, 0xc>(v);
}
__device__ __forceinline__ CountBox last_lane(const CountBox& v) {
  CountBox o;
  o.n = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v.n), 63));
  for (int j = 0; j < 3; j++) {
    o.mx[j] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v.mx[j]), 63));
    o.mn[j] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v.mn[j]), 63));
  }
  return o;
}
// the better of two (cost, index) pairs: smaller cost, the later split on a tie (SplitParams::min, bvh.hpp)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void best_step(float& key, uint32_t& idx) {
  const float ok = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(__builtin_inff()), __float_as_int(key), CTRL,
                                                              ROW_MASK, 0xf, false));
  const uint32_t oi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(idx), CTRL, ROW_MASK, 0xf, false));
  if (ok < key || (ok == key && oi > idx)) {
    key = ok;
    idx = oi;
  }
}

// `load(b)` returns bin b (b < num_bins) as it is stored.  rb: 6 * num_bins words of LDS, rb[j * num_bins + b].
// All 64 lanes of the wave call this; lane 0 writes *out.
// LDS written by a wave and read back by the same wave: a block barrier when the block is that one wave
// (k_bin_sah, k_sah_multi), else the wave's own in-order LDS queue plus a compiler fence (k_subtree)
template <bool BLOCK>
__device__ __forceinline__ void wave_lds_sync() {
  if (BLOCK) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// `load_again(s)` reads bin s for the second (left to right) pass; it may leave the bin cleared behind it.
template <class Load, bool BLOCK = true, class LoadAgain = Load>
__device__ __forceinline__ void wave_sah(uint32_t num_bins, uint32_t n, int axis, Load load, float* rb_f, AxisBest* out,
                                         LoadAgain load_again) {
  uint32_t* rb = reinterpret_cast<uint32_t*>(rb_f);
  const int lane = static_cast<int>(threadIdx.x & 63u);
  const uint32_t chunks = (num_bins + 63) / 64;
  // right_bounds[b] = box of bins b .. num_bins-1
  CountBox carry = empty_count_box();
  for (uint32_t c = chunks; c-- > 0;) {
    const uint32_t b = c * 64 + (63 - lane); // bins right to left over the lanes: a suffix is a prefix scan
    CountBox v = b < num_bins ? load(b) : empty_count_box();
    wave_scan(v);
    combine(v, carry);
    if (b < num_bins)
      for (int j = 0; j < 3; j++) {
        rb[j * num_bins + b] = v.mx[j];
        rb[(3 + j) * num_bins + b] = v.mn[j];
      }
    carry = last_lane(v);
  }
  wave_lds_sync<BLOCK>(); // the wave's LDS writes above are visible to its reads below
  // left_bounds[s] = box of bins 0 .. s; every split s < num_bins - 1 in its own lane
  bool has = false;
  float best_cost = kFltMax;
  uint32_t best_index = 0, best_left_n = 0;
  Box3 best_left, best_right;
  for (int i = 0; i < 3; i++) best_left.top[i] = best_left.bot[i] = best_right.top[i] = best_right.bot[i] = 0.0f;
  carry = empty_count_box();
  for (uint32_t c = 0; c < chunks; c++) {
    const uint32_t s = c * 64 + lane;
    CountBox v = s < num_bins ? load_again(s) : empty_count_box();
    wave_scan(v);
    CountBox left = carry;
    combine(left, v);
    if (s + 1 < num_bins) {
      const uint32_t left_n = left.n, right_n = n - left_n;
      if (left_n > 2 && right_n > 2) {
        Box3 right;
        for (int j = 0; j < 3; j++) {
          right.top[j] = dec_max(rb[j * num_bins + s]);
          right.bot[j] = dec_min(rb[(3 + j) * num_bins + s]);
        }
        const Box3 lbox = decode(left);
        const float total = cost(lbox.top, lbox.bot, left_n) + cost(right.top, right.bot, right_n);
        if (total <= kFltMax && !(best_cost < total)) {
          has = true;
          best_cost = total;
          best_index = s;
          best_left_n = left_n;
          best_left = lbox;
          best_right = right;
        }
      }
    }
    carry = last_lane(left);
  }
  // smallest cost, the later split on a tie (SplitParams::min, bvh.hpp)
  float key = has ? best_cost : __builtin_inff();
  uint32_t idx = best_index;
  best_step<0x111, 0xf>(key, idx);
  best_step<0x112, 0xf>(key, idx);
  best_step<0x114, 0xf>(key, idx);
  best_step<0x118, 0xf>(key, idx);
  best_step<0x142, 0xa>(key, idx);
  best_step<0x143, 0xc>(key, idx);
  key = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(key), 63)); // lane 63 holds the best of all
  idx = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(idx), 63));
  const bool any = key != __builtin_inff();
  if (any && has && best_index == idx && best_cost == key) {
    AxisBest r;
    r.cost = best_cost;
    r.index = best_index;
    r.left_n = best_left_n;
    r.axis = axis;
    for (int i = 0; i < 3; i++) {
      r.ltop[i] = best_left.top[i];
      r.lbot[i] = best_left.bot[i];
      r.rtop[i] = best_right.top[i];
      r.rbot[i] = best_right.bot[i];
    }
    *out = r;
  }
  if (!any && lane == 0) {
    AxisBest r;
    r.cost = kFltMax;
    r.index = r.left_n = 0;
    r.axis = -1;
    for (int i = 0; i < 3; i++) r.ltop[i] = r.lbot[i] = r.rtop[i] = r.rbot[i] = 0.0f;
    *out = r;
  }
}

template <class Load, bool BLOCK = true>
__device__ __forceinline__ void wave_sah(uint32_t num_bins, uint32_t n, int axis, Load load, float* rb_f, AxisBest* out) {
  wave_sah<Load, BLOCK, Load>(num_bins, n, axis, load, rb_f, out, load);
}

// One work-group of kBinWaves waves per work item (the waves share the item's positions: a wave alone spends the
// item waiting for its order -> box gathers).  Dynamic LDS: per axis held, 7 * num_bins words of bins (count,
// 3 enc(max), 3 ~enc(min), each its own array), then per axis held 6 * num_bins words of suffix boxes.  FUSED:
// the three axes are binned in one pass over the triangles (21 * num_bins words of bins) and, for a node that
// fits the item, evaluated by three waves side by side; otherwise one pass per axis.
constexpr uint32_t kBinWaves = 4;
template <bool FUSED>
__global__ void __launch_bounds__(kBinWaves * 64) k_bin_sah(const uint32_t* __restrict__ order, const BNode* __restrict__ nodes,
                                                const Slot* __restrict__ slots, const uint32_t* __restrict__ n_slots_ptr,
                                                const uint32_t* __restrict__ item_base,
                                                const uint32_t* __restrict__ n_items_ptr, uint32_t chunk,
                                                const float4* __restrict__ tbox, uint32_t num_bins,
                                                uint32_t* __restrict__ partials, uint32_t* __restrict__ part_owner,
                                                uint32_t* __restrict__ pbins, AxisBest* __restrict__ axis_best) {
  extern __shared__ uint32_t lds[];
  const uint32_t item = blockIdx.x;
  if (item >= *n_items_ptr) return;
  const uint32_t lane = threadIdx.x; // (of the work-group)
  const uint32_t wave = threadIdx.x >> 6;
  constexpr uint32_t kT = kBinWaves * 64;
  // the slot whose items include `item`: last s with item_base[s] <= item (every slot has >= 1 item)
  uint32_t lo = 0, hi = *n_slots_ptr - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (item_base[mid] <= item) lo = mid; else hi = mid - 1;
  }
  const uint32_t s = lo;
  const Slot sl = slots[s];
  const BNode& nd = nodes[sl.node];
  const uint32_t first = nd.begin + (item - item_base[s]) * chunk;
  const uint32_t last = first + chunk < nd.end ? first + chunk : nd.end;
  if (sl.multi != kNone && lane == 0) part_owner[sl.part_base + (item - item_base[s])] = sl.multi;
  constexpr int kHeld = FUSED ? 3 : 1; // axes whose bins are in LDS at once
  float* rb = reinterpret_cast<float*>(lds + kHeld * 7 * num_bins) + (FUSED ? wave : 0u) * 6 * num_bins; // wave a: axis a
  const uint32_t num_splits = num_bins - 1;

  auto add = [&](uint32_t* bins, int a, const float4& top, const float4& bot) -> uint32_t {
    const float tp = a == 0 ? top.x : (a == 1 ? top.y : top.z);
    const float bt = a == 0 ? bot.x : (a == 1 ? bot.y : bot.z);
    const uint32_t b = bin_index((tp + bt) * 0.5f, sl.start[a], sl.inv[a], num_splits); // AABB::get_center
    atomicAdd(&bins[b], 1u);
    atomicMax(&bins[num_bins + b], enc(top.x));
    atomicMax(&bins[2 * num_bins + b], enc(top.y));
    atomicMax(&bins[3 * num_bins + b], enc(top.z));
    atomicMax(&bins[4 * num_bins + b], ~enc(bot.x));
    atomicMax(&bins[5 * num_bins + b], ~enc(bot.y));
    atomicMax(&bins[6 * num_bins + b], ~enc(bot.z));
    return b;
  };

  for (int pass = 0; pass < (FUSED ? 1 : 3); pass++) {
    const uint32_t pass_mask = FUSED ? sl.axis_mask : (sl.axis_mask & (1u << pass));
    if (pass_mask) {
      for (uint32_t b = lane; b < kHeld * 7 * num_bins; b += kT) lds[b] = 0;
      __syncthreads();
      // four positions per lane in flight: the order -> box gathers are what this loop waits for
      for (uint32_t k0 = first + lane; k0 < last; k0 += 4 * kT) {
        uint32_t t[4];
        float4 top[4], bot[4];
        for (int j = 0; j < 4; j++) {
          const uint32_t k = k0 + kT * j;
          t[j] = k < last ? order[k] : kNone;
        }
        for (int j = 0; j < 4; j++)
          if (t[j] != kNone) {
            top[j] = tbox[2ull * t[j]];
            bot[j] = tbox[2ull * t[j] + 1];
          }
        for (int j = 0; j < 4; j++)
          if (t[j] != kNone) {
            // the three bin indices of a position, 10 bits each, for the partition (k_goes_left)
            const uint32_t k = k0 + kT * j;
            if (FUSED) {
              uint32_t packed = 0;
              for (int a = 0; a < 3; a++)
                if (pass_mask >> a & 1u) packed |= add(lds + a * 7 * num_bins, a, top[j], bot[j]) << (10 * a);
              pbins[k] = packed;
            } else {
              const uint32_t b = add(lds, pass, top[j], bot[j]) << (10 * pass);
              pbins[k] = (sl.axis_mask & ((1u << pass) - 1u)) ? (pbins[k] | b) : b; // first pass taken writes
            }
          }
      }
      __syncthreads();
    }
    for (int a = FUSED ? 0 : pass; a < (FUSED ? 3 : pass + 1); a++) {
      if (!(sl.axis_mask >> a & 1u)) {
        if (sl.multi == kNone && lane == 0) {
          AxisBest r;
          r.cost = kFltMax;
          r.index = r.left_n = 0;
          r.axis = -1;
          for (int i = 0; i < 3; i++) r.ltop[i] = r.lbot[i] = r.rtop[i] = r.rbot[i] = 0.0f;
          axis_best[s * 3 + a] = r;
        }
        continue;
      }
      const uint32_t* bins = lds + (FUSED ? a : 0) * 7 * num_bins;
      if (sl.multi == kNone) {
        auto load = [&](uint32_t b) {
          CountBox c;
          c.n = bins[b];
          for (int j = 0; j < 3; j++) {
            c.mx[j] = bins[(1 + j) * num_bins + b];
            c.mn[j] = bins[(4 + j) * num_bins + b];
          }
          return c;
        };
        // FUSED: wave a evaluates axis a (its own suffix boxes); else the first wave the pass's axis
        if (wave == (FUSED ? static_cast<uint32_t>(a) : 0u))
          wave_sah<decltype(load), false>(num_bins, sl.n, a, load, rb, &axis_best[s * 3 + a]);
      } else {
        // this item's bins go to its slab; k_reduce_partials adds the slabs of a node up
        const uint32_t pi = sl.part_base + (item - item_base[s]);
        uint32_t* dst = partials + static_cast<size_t>(pi) * 21 * num_bins + static_cast<size_t>(a) * 7 * num_bins;
        for (uint32_t b = lane; b < 7 * num_bins; b += kT) dst[b] = bins[b];
      }
    }
    __syncthreads();
  }
}

// Sum of the slabs of every large node into its global bins.  x: the 21 * num_bins words of a slab
// (axis, field, bin); y: groups of kReduceGroup consecutive slabs, one atomic per word, group and node.
constexpr uint32_t kReduceGroup = 32;
__global__ void k_reduce_partials(const uint32_t* __restrict__ partials, const uint32_t* __restrict__ part_owner,
                                  const uint32_t* __restrict__ n_partials, const Slot* __restrict__ slots,
                                  const uint32_t* __restrict__ multi_list, uint32_t num_bins, Bin* __restrict__ multi_bins) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= 21 * num_bins) return;
  const uint32_t a = w / (7 * num_bins), f = (w / num_bins) % 7, b = w % num_bins;
  const uint32_t i0 = blockIdx.y * kReduceGroup;
  const uint32_t total = *n_partials;
  const uint32_t i1 = i0 + kReduceGroup < total ? i0 + kReduceGroup : total;
  uint32_t owner = kNone, acc = 0;
  bool live = false;
  auto flush = [&]() {
    if (owner == kNone || acc == 0) return;
    uint32_t* dst = reinterpret_cast<uint32_t*>(multi_bins + (static_cast<size_t>(owner) * 3 + a) * num_bins + b) + f;
    if (f == 0) atomicAdd(dst, acc); else raise(dst, acc);
  };
  for (uint32_t i = i0; i < i1; i++) {
    const uint32_t o = part_owner[i];
    if (o != owner) {
      flush();
      owner = o;
      acc = 0;
      live = (slots[multi_list[o]].axis_mask >> a & 1u) != 0;
    }
    if (live) {
      const uint32_t v = partials[static_cast<size_t>(i) * 21 * num_bins + w];
      acc = f == 0 ? acc + v : (acc < v ? v : acc);
    }
  }
  flush();
}

// One wave per (node with global bins, axis).  Dynamic LDS: 6 * num_bins floats.
__global__ void __launch_bounds__(64) k_sah_multi(const Slot* __restrict__ slots,
                                                  const uint32_t* __restrict__ multi_list,
                                                  const uint32_t* __restrict__ multi_count, uint32_t num_bins,
                                                  const Bin* __restrict__ multi_bins, AxisBest* __restrict__ axis_best) {
  extern __shared__ uint32_t lds[];
  const uint32_t m = blockIdx.x / 3;
  const int a = static_cast<int>(blockIdx.x % 3);
  if (m >= *multi_count) return;
  const uint32_t s = multi_list[m];
  const Slot sl = slots[s];
  if (!(sl.axis_mask >> a & 1u)) {
    if (threadIdx.x == 0) {
      AxisBest r;
      r.cost = kFltMax;
      r.index = r.left_n = 0;
      r.axis = -1;
      for (int i = 0; i < 3; i++) r.ltop[i] = r.lbot[i] = r.rtop[i] = r.rbot[i] = 0.0f;
      axis_best[s * 3 + a] = r;
    }
    return;
  }
  const Bin* gb = multi_bins + (static_cast<size_t>(sl.multi) * 3 + a) * num_bins;
  auto load = [&](uint32_t b) {
    CountBox c;
    c.n = gb[b].count;
    for (int j = 0; j < 3; j++) {
      c.mx[j] = gb[b].mx[j];
      c.mn[j] = gb[b].mn[j];
    }
    return c;
  };
  wave_sah(num_bins, sl.n, a, load, reinterpret_cast<float*>(lds), &axis_best[s * 3 + a]);
}

// bvh.cpp:141-142, 184-190: the best of the node's own cost and the three axes, later wins a tie
__global__ void k_choose(const Slot* __restrict__ slots, const uint32_t* __restrict__ n_slots_ptr, uint32_t bound,
                         const BNode* __restrict__ nodes, const AxisBest* __restrict__ axis_best,
                         Choice* __restrict__ choice, uint32_t* __restrict__ split_flag) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= bound) return;
  if (s >= *n_slots_ptr) {
    split_flag[s] = 0;
    return;
  }
  const Slot& sl = slots[s];
  const BNode& nd = nodes[sl.node];
  float best_cost = cost(nd.top, nd.bot, sl.n);
  Choice c;
  c.axis = -1;
  c.index = 0;
  c.left_n = 0;
  c.child = 0;
  for (int a = 0; a < 3; a++) {
    const AxisBest& ab = axis_best[s * 3 + a];
    if (ab.axis < 0) continue;
    if (!(best_cost < ab.cost)) {
      best_cost = ab.cost;
      c.axis = a;
      c.index = ab.index;
      c.left_n = ab.left_n;
    }
  }
  choice[s] = c;
  split_flag[s] = c.axis >= 0 ? 1u : 0u;
}

// k_choose and the scan of its flags in one work-group (top of the tree): split_rank[s] = splitting slots before s
__global__ void __launch_bounds__(kScanBlock) k_choose_scan(const Slot* __restrict__ slots, const uint32_t* __restrict__ n_slots_ptr,
                                                            const BNode* __restrict__ nodes, const AxisBest* __restrict__ axis_best,
                                                            Choice* __restrict__ choice, uint32_t* __restrict__ split_rank,
                                                            uint32_t* __restrict__ total, uint32_t* __restrict__ lv) {
  const uint32_t n_slots = *n_slots_ptr;
  uint32_t carry = 0;
  for (uint32_t start = 0; start < n_slots; start += kScanBlock) {
    const uint32_t s = start + threadIdx.x;
    uint32_t f = 0;
    if (s < n_slots) {
      const Slot& sl = slots[s];
      const BNode& nd = nodes[sl.node];
      float best_cost = cost(nd.top, nd.bot, sl.n);
      Choice c;
      c.axis = -1;
      c.index = 0;
      c.left_n = 0;
      c.child = 0;
      for (int a = 0; a < 3; a++) {
        const AxisBest& ab = axis_best[s * 3 + a];
        if (ab.axis < 0) continue;
        if (!(best_cost < ab.cost)) {
          best_cost = ab.cost;
          c.axis = a;
          c.index = ab.index;
          c.left_n = ab.left_n;
        }
      }
      choice[s] = c;
      f = c.axis >= 0 ? 1u : 0u;
    }
    uint32_t sum;
    const uint32_t ex = carry + block_exclusive(f, &sum);
    if (s < n_slots) split_rank[s] = ex;
    carry += sum;
  }
  if (threadIdx.x == 0) {
    *total = carry;
    if (lv) lv[2] = lv[1] + 2u * carry; // where the next level ends
  }
}

// bvh.cpp:107-118: the two child records
__global__ void k_make_children(const Slot* __restrict__ slots, const uint32_t* __restrict__ n_slots_ptr,
                                BNode* __restrict__ nodes, const AxisBest* __restrict__ axis_best,
                                Choice* __restrict__ choice, const uint32_t* __restrict__ split_rank,
                                uint32_t level_end, const uint32_t* __restrict__ lv) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= *n_slots_ptr) return;
  if (lv) level_end = lv[1];
  Choice& c = choice[s];
  if (c.axis < 0) return;
  BNode& nd = nodes[slots[s].node];
  const AxisBest& ab = axis_best[s * 3 + c.axis];
  const uint32_t child = level_end + 2 * split_rank[s];
  BNode l, r;
  for (int i = 0; i < 3; i++) {
    l.top[i] = ab.ltop[i];
    l.bot[i] = ab.lbot[i];
    r.top[i] = ab.rtop[i];
    r.bot[i] = ab.rbot[i];
  }
  l.begin = nd.begin;
  l.end = nd.begin + c.left_n;
  r.begin = l.end;
  r.end = nd.end;
  l.left = l.right = r.left = r.right = 0;
  l.slot = r.slot = kNone;
  l.size = r.size = 1;
  nodes[child] = l;
  nodes[child + 1] = r;
  nd.left = child;
  nd.right = child + 1;
  c.child = child;
}

// 1 when position k moves into the left child (bvh.cpp:120-128)
__device__ __forceinline__ uint32_t goes_left_at(uint32_t k, const uint32_t* __restrict__ node_of,
                                                 const BNode* __restrict__ nodes, uint32_t ls,
                                                 const Choice* __restrict__ choice, const uint32_t* __restrict__ pbins) {
  const uint32_t nd = node_of[k];
  if (nd < ls) return 0u;
  const uint32_t s = nodes[nd].slot;
  if (s == kNone) return 0u;
  const Choice c = choice[s];
  if (c.axis < 0) return 0u;
  return ((pbins[k] >> (10 * c.axis)) & 1023u) <= c.index ? 1u : 0u;
}
__global__ void k_goes_left(const uint32_t* __restrict__ node_of, uint32_t n, const BNode* __restrict__ nodes,
                            uint32_t ls, const Choice* __restrict__ choice, const uint32_t* __restrict__ pbins,
                            uint32_t* __restrict__ goes_left, const uint32_t* __restrict__ lv) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  if (lv) ls = lv[0];
  goes_left[k] = goes_left_at(k, node_of, nodes, ls, choice, pbins);
}
// k_goes_left folded into the first and the last pass of the three-pass scan (k_scan_reduce, k_scan_apply): the flags
// are also written once, for k_scatter
__global__ void __launch_bounds__(kScanBlock) k_gl_scan_reduce(const uint32_t* __restrict__ node_of, uint32_t n,
                                                               const BNode* __restrict__ nodes, uint32_t ls,
                                                               const Choice* __restrict__ choice,
                                                               const uint32_t* __restrict__ pbins,
                                                               uint32_t* __restrict__ goes_left,
                                                               uint32_t* __restrict__ block_sums,
                                                               const uint32_t* __restrict__ lv) {
  if (lv) ls = lv[0];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  uint32_t s = 0;
  for (uint32_t j = 0; j < kScanItems; j++)
    if (base + j < n) {
      const uint32_t f = goes_left_at(base + j, node_of, nodes, ls, choice, pbins);
      goes_left[base + j] = f;
      s += f;
    }
  uint32_t total;
  (void) block_exclusive(s, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// stable split of every splitting node's segment (left part in order, then right part in order)
__global__ void k_scatter(const uint32_t* __restrict__ order, const uint32_t* __restrict__ node_of, uint32_t n,
                          const BNode* __restrict__ nodes, uint32_t ls, const Choice* __restrict__ choice,
                          const uint32_t* __restrict__ goes_left, const uint32_t* __restrict__ left_before,
                          uint32_t* __restrict__ order_out, uint32_t* __restrict__ node_of_out,
                          const uint32_t* __restrict__ lv) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  if (lv) ls = lv[0];
  const uint32_t nd = node_of[k];
  uint32_t pos = k, owner = nd;
  if (nd >= ls) {
    const BNode& node = nodes[nd];
    if (node.slot != kNone) {
      const Choice c = choice[node.slot];
      if (c.axis >= 0) {
        const uint32_t l = left_before[k] - left_before[node.begin];
        if (goes_left[k]) {
          pos = node.begin + l;
          owner = c.child;
        } else {
          pos = node.begin + c.left_n + (k - node.begin - l);
          owner = c.child + 1;
        }
      }
    }
  }
  order_out[pos] = order[k];
  node_of_out[pos] = owner;
}

// ---- small subtrees: one work-group finishes the whole subtree of a node ----------------------------------
// Level-synchronous construction pays ~20 dependent launches and a host wait per level, and from depth ~10 on
// the levels are millions of tiny nodes.  A node of at most kSmallTris triangles at depth >= kSmallMinDepth (so
// that a wave's bins fit LDS) is therefore taken out of the level loop ("small root") and k_subtree builds its
// whole subtree in one work-group: the triangle order of its segment lives in LDS (ping-pong), each wave takes
// a node of the current local level — bins, split evaluation (wave_sah), choice, child records, stable split —
// and the work-group meets once per local level.  Same arithmetic, same decisions, same tree.  Nodes are
// written to a per-root region of a temporary array with local ids (level by level); k_subtree_place then moves
// them to their final ids: "stage" l of the final array holds the nodes of local level l of every root, so that
// the stage list (top levels, then local levels 1, 2, ...) is still a layering with parents before children,
// which is all the later passes (sizes, flattening, bvh_pack_device.hpp) need of "levels".
constexpr uint32_t kMaxBlindLevels = 16;
constexpr uint32_t kSmallTris = 2048;
constexpr uint32_t kSmallMinDepth = 7;
constexpr uint32_t kSmallBins = 128;   // bins at kSmallMinDepth: 1024 / 8
constexpr uint32_t kSmallWaves = 4;
constexpr uint32_t kMaxStages = 1024;  // local levels (the build stops at depth 512: bins <= 1)

struct SmallRoot {
  uint32_t node;    // id of the root in the node array
  uint32_t depth;   // its depth
  uint32_t region;  // first record of its region in the temporary node array (capacity max_nodes(n))
};

// flag[i] = 1 for a node of the top tree that k_subtree takes: never visited by the level loop (slot == kNone,
// still a leaf) with more than 8 triangles; cap[i] = records its subtree may need
__global__ void k_small_flags(const BNode* __restrict__ nodes, uint32_t n_nodes, uint32_t* __restrict__ flag,
                              uint32_t* __restrict__ cap) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const BNode& nd = nodes[i];
  const uint32_t cnt = nd.end - nd.begin;
  const bool take = nd.left == 0 && nd.slot == kNone && cnt > kMinTrianglesPerLeaf;
  flag[i] = take ? 1u : 0u;
  const uint32_t leaves = cnt / 3 > 1 ? cnt / 3 : 1;
  cap[i] = take ? 2 * leaves - 1 : 0u;
}
__global__ void k_small_list(const BNode* __restrict__ nodes, uint32_t n_nodes, const uint32_t* __restrict__ flag,
                             const uint32_t* __restrict__ rank, const uint32_t* __restrict__ cap_before,
                             const uint32_t* __restrict__ level_start, uint32_t n_levels, SmallRoot* __restrict__ roots) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes || !flag[i]) return;
  uint32_t d = 0;
  while (d + 1 < n_levels && level_start[d + 1] <= i) d++;
  roots[rank[i]] = { i, d, cap_before[i] };
}

// ---- nodes of at most 16 triangles, four at a time ---------------------------------------------------------------
// Half of the inner nodes of a subtree have 9..16 triangles, and a wave that sweeps 48..128 bins for each of them
// spends ~1300 VALU instructions on it (k_subtree is bound by VALU issue).  Here a wave takes FOUR consecutive
// nodes of the level, one per row of 16 lanes, a lane per triangle: per axis the lanes of a row are sorted by bin
// (rank by fifteen row rotations, one ds_permute per value), boxes are scanned from both ends of the row, and a lane
// whose right neighbour lies in another bin evaluates the split between them — the splits s of wave_sah between
// two occupied bins all cut the node the same way and cost the same, and the last of them wins there, so that is
// the index reported.  Same boxes (unsigned max of the encoded words), same cost arithmetic, same tie rules as
// wave_sah / k_choose, hence the same tree.  Leaves among the four copy their segment.
constexpr uint32_t kTinyTris = 16;
constexpr uint32_t kMidTris = 32; // two nodes per wave, one per half: the same with two rows per node

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_row(uint32_t old, uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(old), static_cast<int>(v), CTRL, 0xf, 0xf, false));
}
template <int K>
__device__ __forceinline__ void rank_step(uint32_t mine, uint32_t theirs, uint32_t& rank) {
  const uint32_t other = dpp_row<0x120 + K>(theirs, theirs); // row_ror:K — `theirs` of lane (l + K) mod 16 of the row
  rank += other < mine ? 1u : 0u;
}
// number of lanes of the row whose `theirs` is smaller than this lane's `mine` (lane l itself included when SELF)
template <bool SELF>
__device__ __forceinline__ uint32_t row_rank(uint32_t mine, uint32_t theirs) {
  uint32_t rank = (SELF && theirs < mine) ? 1u : 0u;
  rank_step<1>(mine, theirs, rank); rank_step<2>(mine, theirs, rank); rank_step<3>(mine, theirs, rank);
  rank_step<4>(mine, theirs, rank); rank_step<5>(mine, theirs, rank); rank_step<6>(mine, theirs, rank);
  rank_step<7>(mine, theirs, rank); rank_step<8>(mine, theirs, rank); rank_step<9>(mine, theirs, rank);
  rank_step<10>(mine, theirs, rank); rank_step<11>(mine, theirs, rank); rank_step<12>(mine, theirs, rank);
  rank_step<13>(mine, theirs, rank); rank_step<14>(mine, theirs, rank); rank_step<15>(mine, theirs, rank);
  return rank;
}
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a < b ? b : a; }
__device__ __forceinline__ uint32_t lane_get(uint32_t src_lane, uint32_t v) { // v of lane src_lane
  return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(src_lane << 2), static_cast<int>(v)));
}
__device__ __forceinline__ uint32_t lane_put(uint32_t dst_lane, uint32_t v) { // this lane's v lands in lane dst_lane
  return static_cast<uint32_t>(__builtin_amdgcn_ds_permute(static_cast<int>(dst_lane << 2), static_cast<int>(v)));
}
// inclusive max scans over the G (16: a row, 32: two rows) lanes of a group; 0 is the identity
template <int G>
__device__ __forceinline__ uint32_t group_prefix_max(uint32_t v) {
  v = umax(v, dpp_row<0x111>(0u, v));
  v = umax(v, dpp_row<0x112>(0u, v));
  v = umax(v, dpp_row<0x114>(0u, v));
  v = umax(v, dpp_row<0x118>(0u, v));
  if (G == 32) v = umax(v, dpp0<0x142, 0xa>(v)); // lane 15 of the lower row into the upper row
  return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_suffix_max(uint32_t v, uint32_t lane) {
  v = umax(v, dpp_row<0x101>(0u, v));
  v = umax(v, dpp_row<0x102>(0u, v));
  v = umax(v, dpp_row<0x104>(0u, v));
  v = umax(v, dpp_row<0x108>(0u, v));
  if (G == 32) {
    const uint32_t upper = lane_get((lane & 32u) + 16u, v); // everything of the upper row
    if ((lane & 16u) == 0u) v = umax(v, upper);
  }
  return v;
}
// the value of the next / the previous lane of the group (`edge` beyond it)
template <int G>
__device__ __forceinline__ uint32_t group_next(uint32_t v, uint32_t edge, uint32_t lane) {
  uint32_t r = dpp_row<0x101>(edge, v);
  if (G == 32) {
    const uint32_t first_upper = lane_get((lane & 32u) + 16u, v);
    if ((lane & 31u) == 15u) r = first_upper;
  }
  return r;
}
template <int G>
__device__ __forceinline__ uint32_t group_prev(uint32_t v, uint32_t edge, uint32_t lane) {
  uint32_t r = dpp_row<0x111>(edge, v);
  if (G == 32) {
    const uint32_t last_lower = lane_get((lane & 32u) + 15u, v);
    if ((lane & 31u) == 16u) r = last_lower;
  }
  return r;
}

// j: the node of this lane's group (the same in its G lanes), of at most G triangles, or kNone
template <int G>
__device__ __forceinline__ void tiny_nodes(BNode* __restrict__ nodes, uint32_t j, uint32_t num_bins, bool may_split,
                                           const uint32_t* cur, uint32_t* nxt, const float4* __restrict__ tbox,
                                           uint32_t* s_next, uint32_t lane) {
  static_assert(G == 16 || G == 32, "a row or two");
  constexpr uint32_t kGroupMask = G == 32 ? 0xFFFFFFFFu : 0xFFFFu;
  const uint32_t l = lane & (G - 1u), row_base = lane & (64u - G);
  const bool have = j != kNone;
  BNode nd;
  for (int i = 0; i < 3; i++) nd.top[i] = nd.bot[i] = 0.0f;
  nd.begin = nd.end = 0;
  if (have) nd = nodes[j];
  const uint32_t n = nd.end - nd.begin;
  const bool mine = have && l < n;                                         // this lane holds a position
  const bool try_split = have && may_split && n > kMinTrianglesPerLeaf;   // (the same for the group)
  const uint32_t t = mine ? cur[nd.begin + l] : 0u;
  if (__ballot(try_split) == 0ull) { // leaves only: their segments stay as they are
    if (mine) nxt[nd.begin + l] = t;
    return;
  }
  float4 top = make_float4(0.f, 0.f, 0.f, 0.f), bot = top;
  if (mine && try_split) {
    top = tbox[2ull * t];
    bot = tbox[2ull * t + 1];
  }
  // k_slot_setup
  float start[3], inv[3];
  uint32_t axis_mask = 0;
  const float fbins = static_cast<float>(num_bins);
  for (int a = 0; a < 3; a++) {
    const float step = (nd.top[a] - nd.bot[a]) / fbins;
    start[a] = nd.bot[a];
    inv[a] = 1.0f / step;
    if (!(fabsf(nd.top[a] - nd.bot[a]) < 1e-4f)) axis_mask |= 1u << a;
  }
  const uint32_t num_splits = num_bins - 1;
  const bool data = mine && try_split;
  const uint32_t E[6] = { data ? enc(top.x) : 0u, data ? enc(top.y) : 0u, data ? enc(top.z) : 0u,
                          data ? ~enc(bot.x) : 0u, data ? ~enc(bot.y) : 0u, data ? ~enc(bot.z) : 0u };
  const uint32_t b3[3] = { bin_index((top.x + bot.x) * 0.5f, start[0], inv[0], num_splits),
                           bin_index((top.y + bot.y) * 0.5f, start[1], inv[1], num_splits),
                           bin_index((top.z + bot.z) * 0.5f, start[2], inv[2], num_splits) };

  float best_cost = cost(nd.top, nd.bot, n); // k_choose: the node's own cost first
  int chosen_axis = -1;
  uint32_t chosen_index = 0, chosen_pos = 0;
  uint32_t keep_l[6] = { 0, 0, 0, 0, 0, 0 }, keep_r[6] = { 0, 0, 0, 0, 0, 0 };
  for (int a = 0; a < 3; a++) {
    const bool on = data && (axis_mask >> a & 1u);
    // lanes of the group in bin order (lane number breaks ties; lanes without a triangle last)
    const uint32_t key = on ? (b3[a] << 5) | l : 0xFFFF00u | l;
    uint32_t rank = row_rank<false>(key, key);
    if (G == 32) rank += row_rank<true>(key, lane_get(lane ^ 16u, key)); // ... and the keys of the other row
    const uint32_t dst = row_base + rank;
    const uint32_t sb = lane_put(dst, on ? b3[a] : 0xFFFFu);
    uint32_t pre[6], suf[6];
    for (int i = 0; i < 6; i++) {
      const uint32_t v = lane_put(dst, on ? E[i] : 0u);
      // from here on lane l is POSITION l of the sorted group
      pre[i] = group_prefix_max<G>(v);
      suf[i] = group_suffix_max<G>(v, lane);
    }
    // The reference pairs left_bounds[s] with right_bounds[s], whose box still holds bin s (bvh.cpp:77-81).  So the
    // splits between this lane's bin v and the next occupied bin w come in two kinds: s = v with a right box from
    // the first triangle of bin v on, and v < s < w (if any) with the right box from bin w on; of the latter the
    // last one, w - 1, wins their tie.
    uint32_t right_w[6], right_v[6];
    for (int i = 0; i < 6; i++) right_w[i] = group_next<G>(suf[i], 0u, lane); // positions l + 1 ..
    const uint32_t next_bin = group_next<G>(sb, 0xFFFFu, lane);
    const uint32_t prev_bin = group_prev<G>(sb, 0xFFFFFFFFu, lane);
    const uint32_t first = group_prefix_max<G>(sb != prev_bin ? l : 0u); // position of the first triangle of this lane's bin
    for (int i = 0; i < 6; i++) right_v[i] = lane_get(row_base + first, suf[i]);
    const uint32_t left_n = l + 1, right_n = n - left_n;
    // (n is the group's node's count; the group's axis switch: the same for every lane of the group)
    const bool axis_on = try_split && (axis_mask >> a & 1u);
    const bool last_of_bin = axis_on && left_n < n && sb != next_bin && left_n > 2 && right_n > 2;
    Box3 lb, rv, rw;
    for (int i = 0; i < 3; i++) {
      lb.top[i] = dec_max(pre[i]);
      lb.bot[i] = dec_min(pre[3 + i]);
      rv.top[i] = dec_max(right_v[i]);
      rv.bot[i] = dec_min(right_v[3 + i]);
      rw.top[i] = dec_max(right_w[i]);
      rw.bot[i] = dec_min(right_w[3 + i]);
    }
    const float left_cost = cost(lb.top, lb.bot, left_n);
    const float total_v = left_cost + cost(rv.top, rv.bot, right_n);
    const float total_w = left_cost + cost(rw.top, rw.bot, right_n);
    const bool ok_v = last_of_bin && total_v <= kFltMax;
    const bool ok_w = last_of_bin && next_bin - 1u > sb && total_w <= kFltMax;
    const bool use_w = ok_w && !(ok_v && total_v < total_w); // the later split on a tie
    const bool valid = ok_v || ok_w;
    const float total = use_w ? total_w : total_v;
    const uint32_t my_index = use_w ? next_bin - 1u : sb;
    uint32_t right[6];
    for (int i = 0; i < 6; i++) right[i] = use_w ? right_w[i] : right_v[i];
    // smallest cost, the later split on a tie: the last lane of the group ends up with the best of the group
    float key_cost = valid ? total : __builtin_inff();
    uint32_t idx = valid ? my_index : 0u;
    best_step<0x111, 0xf>(key_cost, idx);
    best_step<0x112, 0xf>(key_cost, idx);
    best_step<0x114, 0xf>(key_cost, idx);
    best_step<0x118, 0xf>(key_cost, idx);
    if (G == 32) best_step<0x142, 0xa>(key_cost, idx);
    const float row_cost = __uint_as_float(lane_get(row_base + (G - 1u), __float_as_uint(key_cost)));
    const uint32_t row_index = lane_get(row_base + (G - 1u), idx);
    const bool winner = valid && total == row_cost && my_index == row_index;
    const uint32_t wmask = static_cast<uint32_t>((__ballot(winner) >> row_base) & kGroupMask);
    // k_choose: the later axis wins a tie
    if (wmask != 0u && !(best_cost < row_cost)) {
      best_cost = row_cost;
      chosen_axis = a;
      chosen_index = row_index;
      chosen_pos = static_cast<uint32_t>(__builtin_ctz(wmask));
      for (int i = 0; i < 6; i++) {
        keep_l[i] = pre[i];
        keep_r[i] = right[i];
      }
    }
  }
  // the winner's boxes to the whole group (its first lane writes the records)
  const uint32_t wl = row_base + chosen_pos;
  uint32_t wbox_l[6], wbox_r[6];
  for (int i = 0; i < 6; i++) {
    wbox_l[i] = lane_get(wl, keep_l[i]);
    wbox_r[i] = lane_get(wl, keep_r[i]);
  }
  const bool split = chosen_axis >= 0;
  const uint32_t left_count = chosen_pos + 1;
  uint32_t child = 0;
  if (split && l == 0) child = atomicAdd(s_next, 2u);
  child = lane_get(row_base, child);
  if (split && l == 0) {
    BNode lc, rc;
    for (int i = 0; i < 3; i++) {
      lc.top[i] = dec_max(wbox_l[i]);
      lc.bot[i] = dec_min(wbox_l[3 + i]);
      rc.top[i] = dec_max(wbox_r[i]);
      rc.bot[i] = dec_min(wbox_r[3 + i]);
    }
    lc.begin = nd.begin;
    lc.end = nd.begin + left_count;
    rc.begin = lc.end;
    rc.end = nd.end;
    lc.left = lc.right = rc.left = rc.right = 0;
    lc.slot = rc.slot = kNone;
    lc.size = rc.size = 1;
    nodes[child] = lc;
    nodes[child + 1] = rc;
    nodes[j].left = child;
    nodes[j].right = child + 1;
  }
  // stable split of the group's segment (lanes are in position order again), or the segment as it is
  const uint32_t my_bin = chosen_axis == 0 ? b3[0] : (chosen_axis == 1 ? b3[1] : b3[2]);
  const bool goes_left = mine && split && my_bin <= chosen_index;
  const bool goes_right = mine && split && !goes_left;
  const uint32_t below = (1u << l) - 1u;
  const uint32_t ml = static_cast<uint32_t>((__ballot(goes_left) >> row_base) & kGroupMask);
  const uint32_t mr = static_cast<uint32_t>((__ballot(goes_right) >> row_base) & kGroupMask);
  if (goes_left) nxt[nd.begin + static_cast<uint32_t>(__popc(ml & below))] = t;
  else if (goes_right) nxt[nd.begin + left_count + static_cast<uint32_t>(__popc(mr & below))] = t;
  else if (mine) nxt[nd.begin + l] = t;
}

// Dynamic LDS: ord[2][kSmallTris], pb[kSmallTris], per wave 27 * kSmallBins words (bins of three axes + suffix boxes)
__global__ void __launch_bounds__(kSmallWaves * 64) k_subtree(
    const SmallRoot* __restrict__ roots, const BNode* __restrict__ top_nodes, const float4* __restrict__ tbox,
    uint32_t* __restrict__ order /* in and out: the root's segment */, BNode* __restrict__ tmp,
    uint32_t* __restrict__ lvl_first /* [region + l]: first local id of local level l */,
    uint32_t* __restrict__ lvl_base /* [region + l]: rank of that level's first node in stage l */,
    uint32_t* __restrict__ root_levels /* [root]: local levels */, uint32_t* __restrict__ stage_count,
    uint32_t* __restrict__ max_depth) {
  extern __shared__ uint32_t lds[];
  uint32_t* ord0 = lds;
  uint32_t* ord1 = lds + kSmallTris;
  uint32_t* pb = lds + 2 * kSmallTris;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  uint32_t* bins = lds + 3 * kSmallTris + wave * 27 * kSmallBins;
  __shared__ AxisBest s_best[kSmallWaves][3];
  __shared__ uint32_t s_next;

  for (uint32_t b = lane; b < 21 * kSmallBins; b += 64) bins[b] = 0; // this wave's bins; kept clean by wave_sah's second pass
  const SmallRoot R = roots[blockIdx.x];
  const BNode root = top_nodes[R.node];
  const uint32_t n_root = root.end - root.begin;
  BNode* nodes = tmp + R.region;
  for (uint32_t k = threadIdx.x; k < n_root; k += blockDim.x) ord0[k] = order[root.begin + k];
  if (threadIdx.x == 0) {
    BNode r0 = root;
    r0.begin = 0;
    r0.end = n_root;
    r0.left = r0.right = 0;
    r0.size = 1;
    nodes[0] = r0;
    lvl_first[R.region] = 0;
    s_next = 1;
  }
  __syncthreads();

  uint32_t* cur = ord0;
  uint32_t* nxt = ord1;
  uint32_t lvl_begin = 0, lvl_end = 1, n_levels = 1;
  for (uint32_t l = 0;; l++) {
    const uint32_t depth = R.depth + l;
    const uint32_t num_bins = static_cast<uint32_t>(kMaxBins / static_cast<float>(depth + 1)); // bvh.cpp:145
    const bool may_split = num_bins > 1;
    auto big_node = [&](uint32_t j) {
      const BNode nd = nodes[j];
      const uint32_t n = nd.end - nd.begin;
      Choice c;
      c.axis = -1;
      c.index = c.left_n = c.child = 0;
      if (may_split && n > kMinTrianglesPerLeaf) {
        // k_slot_setup
        float start[3], inv[3];
        uint32_t axis_mask = 0;
        const float fbins = static_cast<float>(num_bins);
        for (int a = 0; a < 3; a++) {
          const float step = (nd.top[a] - nd.bot[a]) / fbins;
          start[a] = nd.bot[a];
          inv[a] = 1.0f / step;
          if (!(fabsf(nd.top[a] - nd.bot[a]) < 1e-4f)) axis_mask |= 1u << a;
        }
        const uint32_t num_splits = num_bins - 1;
        if (axis_mask) {
          // (the bins are all zero here: cleared once per wave at the start, and by the second pass of wave_sah after
          // every use)
          for (uint32_t k0 = nd.begin + lane; k0 < nd.end; k0 += 256) {
            uint32_t t[4];
            float4 top[4], bot[4];
            for (int q = 0; q < 4; q++) {
              const uint32_t k = k0 + 64 * q;
              t[q] = k < nd.end ? cur[k] : kNone;
            }
            for (int q = 0; q < 4; q++)
              if (t[q] != kNone) {
                top[q] = tbox[2ull * t[q]];
                bot[q] = tbox[2ull * t[q] + 1];
              }
            for (int q = 0; q < 4; q++)
              if (t[q] != kNone) {
                uint32_t packed = 0;
                for (int a = 0; a < 3; a++)
                  if (axis_mask >> a & 1u) {
                    uint32_t* ab = bins + a * 7 * num_bins;
                    const float tp = a == 0 ? top[q].x : (a == 1 ? top[q].y : top[q].z);
                    const float bt = a == 0 ? bot[q].x : (a == 1 ? bot[q].y : bot[q].z);
                    const uint32_t b = bin_index((tp + bt) * 0.5f, start[a], inv[a], num_splits);
                    atomicAdd(&ab[b], 1u);
                    atomicMax(&ab[num_bins + b], enc(top[q].x));
                    atomicMax(&ab[2 * num_bins + b], enc(top[q].y));
                    atomicMax(&ab[3 * num_bins + b], enc(top[q].z));
                    atomicMax(&ab[4 * num_bins + b], ~enc(bot[q].x));
                    atomicMax(&ab[5 * num_bins + b], ~enc(bot[q].y));
                    atomicMax(&ab[6 * num_bins + b], ~enc(bot[q].z));
                    packed |= b << (10 * a);
                  }
                pb[k0 + 64 * q] = packed;
              }
          }
          wave_lds_sync<false>();
        }
        float* rb = reinterpret_cast<float*>(bins + 21 * num_bins);
        for (int a = 0; a < 3; a++) {
          if (!(axis_mask >> a & 1u)) {
            if (lane == 0) s_best[wave][a].axis = -1;
            continue;
          }
          const uint32_t* ab = bins + a * 7 * num_bins;
          auto load = [&](uint32_t b) {
            CountBox cb;
            cb.n = ab[b];
            for (int q = 0; q < 3; q++) {
              cb.mx[q] = ab[(1 + q) * num_bins + b];
              cb.mn[q] = ab[(4 + q) * num_bins + b];
            }
            return cb;
          };
          uint32_t* abw = bins + a * 7 * num_bins;
          auto load_and_clear = [&](uint32_t b) { // the last read of a bin leaves it empty for the wave's next node
            const CountBox cb = load(b);
            for (int q = 0; q < 7; q++) abw[q * num_bins + b] = 0;
            return cb;
          };
          wave_sah<decltype(load), false, decltype(load_and_clear)>(num_bins, n, a, load, rb, &s_best[wave][a],
                                                                    load_and_clear);
          wave_lds_sync<false>(); // rb is reused by the next axis
        }
        wave_lds_sync<false>();
        // k_choose
        float best_cost = cost(nd.top, nd.bot, n);
        for (int a = 0; a < 3; a++) {
          const AxisBest& ab = s_best[wave][a];
          if (ab.axis < 0) continue;
          if (!(best_cost < ab.cost)) {
            best_cost = ab.cost;
            c.axis = a;
            c.index = ab.index;
            c.left_n = ab.left_n;
          }
        }
      }
      if (c.axis >= 0) {
        // k_make_children
        uint32_t child = 0;
        if (lane == 0) child = atomicAdd(&s_next, 2u);
        child = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(child)));
        if (lane == 0) {
          const AxisBest& ab = s_best[wave][c.axis];
          BNode lc, rc;
          for (int i = 0; i < 3; i++) {
            lc.top[i] = ab.ltop[i];
            lc.bot[i] = ab.lbot[i];
            rc.top[i] = ab.rtop[i];
            rc.bot[i] = ab.rbot[i];
          }
          lc.begin = nd.begin;
          lc.end = nd.begin + c.left_n;
          rc.begin = lc.end;
          rc.end = nd.end;
          lc.left = lc.right = rc.left = rc.right = 0;
          lc.slot = rc.slot = kNone;
          lc.size = rc.size = 1;
          nodes[child] = lc;
          nodes[child + 1] = rc;
          nodes[j].left = child;
          nodes[j].right = child + 1;
        }
        // stable split (k_goes_left + k_scatter): left part in order, then right part in order
        uint32_t lbase = nd.begin, rbase = nd.begin + c.left_n;
        for (uint32_t k0 = nd.begin; k0 < nd.end; k0 += 64) {
          const uint32_t k = k0 + lane;
          const bool in = k < nd.end;
          const bool left = in && ((pb[in ? k : nd.begin] >> (10 * c.axis)) & 1023u) <= c.index;
          const uint64_t ml = __ballot(left), mr = __ballot(in && !left);
          const uint64_t below = (1ull << lane) - 1ull;
          if (left) nxt[lbase + static_cast<uint32_t>(__popcll(ml & below))] = cur[k];
          else if (in) nxt[rbase + static_cast<uint32_t>(__popcll(mr & below))] = cur[k];
          lbase += static_cast<uint32_t>(__popcll(ml));
          rbase += static_cast<uint32_t>(__popcll(mr));
        }
      } else {
        for (uint32_t k = nd.begin + lane; k < nd.end; k += 64) nxt[k] = cur[k]; // a leaf: its segment stays
      }
    };
    // A wave takes up to 64 consecutive nodes of the level at a time, a lane per node: those of more than 32 triangles
    // one after the other by the whole wave (big_node), those of 17..32 two at a time, the others four at a time, one
    // per row of lanes (tiny_nodes) — the ones that may split first, so that their turns have all four rows at work
    // and the turns of the leaves that remain cost a copy.
    // (a short level is dealt to all the waves: `per` nodes at a time, at most 64)
    const uint32_t per_wave = (lvl_end - lvl_begin + kSmallWaves - 1) / kSmallWaves;
    const uint32_t per = per_wave < 64u ? per_wave : 64u;
    for (uint32_t j0 = lvl_begin + per * wave; j0 < lvl_end; j0 += per * kSmallWaves) {
      const uint32_t jn = j0 + lane;
      const bool in_turn = lane < per && jn < lvl_end;
      uint32_t cnt = 0;
      if (in_turn) cnt = nodes[jn].end - nodes[jn].begin;
      const bool is_big = in_turn && cnt > kMidTris;
      const bool is_mid = in_turn && !is_big && cnt > kTinyTris;
      const bool is_tiny = in_turn && !is_big && !is_mid && may_split && cnt > kMinTrianglesPerLeaf;
      const bool is_leaf = in_turn && !is_big && !is_mid && !is_tiny;
      const uint64_t m_tiny = __ballot(is_tiny), m_leaf = __ballot(is_leaf), m_mid = __ballot(is_mid);
      uint64_t m_big = __ballot(is_big);
      const uint64_t below = (1ull << lane) - 1ull;
      const uint32_t n_tiny = static_cast<uint32_t>(__popcll(m_tiny)), n_small = n_tiny + static_cast<uint32_t>(__popcll(m_leaf));
      const uint32_t n_mid = static_cast<uint32_t>(__popcll(m_mid));
      // lane k of `list` = the k-th of the small ones: the splittable ones, then the leaves; lane 63 - k = the k-th
      // of 17..32 triangles (a lane that holds none of these sends nothing anyone reads: the two lists meet at most)
      uint32_t slot = 63u - n_mid; // (a free lane between the lists, or — 64 nodes, all listed — nobody sends here)
      if (is_tiny) slot = static_cast<uint32_t>(__popcll(m_tiny & below));
      else if (is_leaf) slot = n_tiny + static_cast<uint32_t>(__popcll(m_leaf & below));
      else if (is_mid) slot = 63u - static_cast<uint32_t>(__popcll(m_mid & below));
      const uint32_t list = lane_put(slot, (is_tiny || is_leaf || is_mid) ? jn : kNone);
      for (uint32_t g = 0; g < n_small; g += 4) {
        const uint32_t k = g + (lane >> 4);
        const uint32_t j = lane_get(k < n_small ? k : 0u, list);
        tiny_nodes<16>(nodes, k < n_small ? j : kNone, num_bins, may_split, cur, nxt, tbox, &s_next, lane);
      }
      for (uint32_t g = 0; g < n_mid; g += 2) {
        const uint32_t k = g + (lane >> 5);
        const uint32_t j = lane_get(k < n_mid ? 63u - k : 0u, list);
        tiny_nodes<32>(nodes, k < n_mid ? j : kNone, num_bins, may_split, cur, nxt, tbox, &s_next, lane);
      }
      while (m_big) {
        const uint32_t r = static_cast<uint32_t>(__builtin_ctzll(m_big));
        m_big &= m_big - 1ull;
        big_node(j0 + r);
      }
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t made = s_next;
    uint32_t* t = cur;
    cur = nxt;
    nxt = t;
    if (made == lvl_end) break; // nothing split at this level
    lvl_begin = lvl_end;
    lvl_end = made;
    if (threadIdx.x == 0) lvl_first[R.region + n_levels] = lvl_begin;
    n_levels++;
    __syncthreads();
  }
  // the root's segment in its final order
  for (uint32_t k = threadIdx.x; k < n_root; k += blockDim.x) order[root.begin + k] = cur[k];
  // subtree sizes, bottom-up over the local levels
  const uint32_t n_local = lvl_end;
  for (uint32_t l = n_levels; l-- > 0;) {
    const uint32_t b = l == 0 ? 0u : lvl_first[R.region + l];
    const uint32_t e = l + 1 < n_levels ? lvl_first[R.region + l + 1] : n_local;
    for (uint32_t j = b + threadIdx.x; j < e; j += blockDim.x) {
      BNode& nd = nodes[j];
      nd.size = nd.left ? 1u + nodes[nd.left].size + nodes[nd.right].size : 1u;
    }
    __threadfence_block();
    __syncthreads();
  }
  // a place in every stage for this root's nodes of that local level
  for (uint32_t l = 1 + threadIdx.x; l < n_levels; l += blockDim.x) {
    const uint32_t b = lvl_first[R.region + l];
    const uint32_t e = l + 1 < n_levels ? lvl_first[R.region + l + 1] : n_local;
    lvl_base[R.region + l] = atomicAdd(&stage_count[l], e - b);
  }
  if (threadIdx.x == 0) {
    root_levels[blockIdx.x] = n_levels;
    lvl_first[R.region + n_levels] = n_local; // sentinel: end of the last level (capacity >= levels + 1: see k_small_flags)
    atomicMax(max_depth, R.depth + n_levels - 1);
  }
}

// Final ids: node j of local level l >= 1 of root r -> stage_first[l] + lvl_base[r][l] + (j - lvl_first[r][l]).
// Writes the final records (absolute positions, final child ids), the root's children and size, and node_of
// for the positions of every leaf.
__global__ void __launch_bounds__(256) k_subtree_place(const SmallRoot* __restrict__ roots, const BNode* __restrict__ tmp,
                                                       const uint32_t* __restrict__ lvl_first,
                                                       const uint32_t* __restrict__ lvl_base,
                                                       const uint32_t* __restrict__ root_levels,
                                                       const uint32_t* __restrict__ stage_first, BNode* __restrict__ nodes,
                                                       uint32_t* __restrict__ node_of) {
  const SmallRoot R = roots[blockIdx.x];
  const uint32_t n_levels = root_levels[blockIdx.x];
  const uint32_t* lf = lvl_first + R.region;
  const uint32_t* lb = lvl_base + R.region;
  const BNode* local = tmp + R.region;
  const uint32_t n_local = lf[n_levels];
  const uint32_t pos0 = nodes[R.node].begin;
  auto final_id = [&](uint32_t j, uint32_t l) { return l == 0 ? R.node : stage_first[l] + lb[l] + (j - lf[l]); };
  for (uint32_t l = 0; l < n_levels; l++) {
    const uint32_t b = lf[l], e = lf[l + 1];
    for (uint32_t j = b + threadIdx.x; j < e; j += blockDim.x) {
      BNode nd = local[j];
      const uint32_t me = final_id(j, l);
      if (nd.left) {
        nd.left = final_id(nd.left, l + 1);
        nd.right = final_id(nd.right, l + 1);
      }
      nd.begin += pos0;
      nd.end += pos0;
      nd.slot = kNone;
      if (l == 0) {
        BNode& root = nodes[me];
        root.left = nd.left;
        root.right = nd.right;
        root.size = nd.size;
      } else {
        nodes[me] = nd;
      }
      if (!nd.left)
        for (uint32_t k = nd.begin; k < nd.end; k++) node_of[k] = me;
    }
  }
  (void) n_local;
}

// k_flatten for the nodes below a small root, one work-group per root, local level by local level (the root's
// own record and its children's indices come from k_flatten over the top stages)
__global__ void __launch_bounds__(256) k_subtree_flatten(const SmallRoot* __restrict__ roots,
                                                         const uint32_t* __restrict__ lvl_first,
                                                         const uint32_t* __restrict__ lvl_base,
                                                         const uint32_t* __restrict__ root_levels,
                                                         const uint32_t* __restrict__ stage_first, BNode* __restrict__ nodes,
                                                         uint32_t* __restrict__ flat_index, uint32_t* __restrict__ tri_offset,
                                                         nova_bvh_node* __restrict__ out) {
  const SmallRoot R = roots[blockIdx.x];
  const uint32_t n_levels = root_levels[blockIdx.x];
  const uint32_t* lf = lvl_first + R.region;
  const uint32_t* lb = lvl_base + R.region;
  for (uint32_t l = 1; l < n_levels; l++) {
    const uint32_t first = stage_first[l] + lb[l], count = lf[l + 1] - lf[l];
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
      BNode& nd = nodes[first + i];
      const uint32_t me = flat_index[first + i], toff = tri_offset[first + i];
      nova_bvh_node f;
      f.top_offset_left = { nd.top[0], nd.top[1], nd.top[2], 0.0f };
      f.bottom_num_right = { nd.bot[0], nd.bot[1], nd.bot[2], 0.0f };
      if (!nd.left) {
        f.top_offset_left.w = static_cast<float>(toff);
        f.bottom_num_right.w = -static_cast<float>(nd.end - nd.begin);
      } else {
        uint32_t a = nd.left, b = nd.right;
        const BNode& ln = nodes[a];
        const BNode& rn = nodes[b];
        if (surface_area(rn.top, rn.bot) > surface_area(ln.top, ln.bot)) { // visit the larger child first
          a = nd.right;
          b = nd.left;
          nd.left = a;
          nd.right = b;
        }
        const BNode& fn = nodes[a];
        flat_index[a] = me + 1;
        flat_index[b] = me + 1 + fn.size;
        tri_offset[a] = toff;
        tri_offset[b] = toff + (fn.end - fn.begin);
        f.bottom_num_right.w = static_cast<float>(me + 1 + fn.size);
      }
      out[me] = f;
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---- flattening (bvh.cpp:211-249) --------------------------------------------------------------------
__global__ void k_sizes(BNode* __restrict__ nodes, uint32_t ls, uint32_t m) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  BNode& nd = nodes[ls + i];
  nd.size = nd.left ? 1u + nodes[nd.left].size + nodes[nd.right].size : 1u;
}

// flat_index / tri_offset of the children, and the node's own flat record
// (a node's children are left in visiting order: left = first)
__global__ void k_flatten(BNode* __restrict__ nodes, uint32_t ls, uint32_t m, uint32_t* __restrict__ flat_index,
                          uint32_t* __restrict__ tri_offset, nova_bvh_node* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  BNode& nd = nodes[ls + i];
  const uint32_t me = flat_index[ls + i], toff = tri_offset[ls + i];
  nova_bvh_node f;
  f.top_offset_left = { nd.top[0], nd.top[1], nd.top[2], 0.0f };
  f.bottom_num_right = { nd.bot[0], nd.bot[1], nd.bot[2], 0.0f };
  if (!nd.left) {
    f.top_offset_left.w = static_cast<float>(toff);
    f.bottom_num_right.w = -static_cast<float>(nd.end - nd.begin);
  } else {
    uint32_t first = nd.left, second = nd.right;
    const BNode& l = nodes[first];
    const BNode& r = nodes[second];
    if (surface_area(r.top, r.bot) > surface_area(l.top, l.bot)) { // visit the larger child first
      first = nd.right;
      second = nd.left;
      nd.left = first;
      nd.right = second;
    }
    const BNode& fn = nodes[first];
    flat_index[first] = me + 1;
    flat_index[second] = me + 1 + fn.size;
    tri_offset[first] = toff;
    tri_offset[second] = toff + (fn.end - fn.begin);
    f.bottom_num_right.w = static_cast<float>(me + 1 + fn.size);
  }
  out[me] = f;
}

// Triangle order of the flattened tree and the Woop matrices:
// rows 0..2 of glm::inverse([e1 e2 n-v1 v1; 0 0 0 1]) (intersectable_manager.cpp:129-145)
__global__ void k_emit_triangles(const uint32_t* __restrict__ order, const uint32_t* __restrict__ node_of, uint32_t n,
                                 const BNode* __restrict__ nodes, const uint32_t* __restrict__ tri_offset,
                                 const float* __restrict__ verts, uint32_t* __restrict__ tri_order,
                                 nova_triangle_data* __restrict__ tris) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t leaf = node_of[k];
  const uint32_t p = tri_offset[leaf] + (k - nodes[leaf].begin);
  const uint32_t t = order[k];
  tri_order[p] = t;
  const float* v = verts + 9ull * t;
  const float v1[3] = { v[0], v[1], v[2] };
  float e1[3], e2[3];
  for (int a = 0; a < 3; a++) {
    e1[a] = v[3 + a] - v1[a];
    e2[a] = v[6 + a] - v1[a];
  }
  float nr[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
  const float inv_len = 1.0f / sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]);
  // glm::inverse(mat4) in glm's own operation order (include/glm/detail/func_matrix.inl:294-352);
  // m[c][r]: columns (e1,0) (e2,0) (n-v1,0) (v1,1)
  float m[4][4];
  for (int a = 0; a < 3; a++) {
    m[0][a] = e1[a];
    m[1][a] = e2[a];
    m[2][a] = nr[a] * inv_len - v1[a];
    m[3][a] = v1[a];
  }
  m[0][3] = m[1][3] = m[2][3] = 0.0f;
  m[3][3] = 1.0f;
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
  float inv[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    inv[0][r] = (vec1[r] * fac0[r] - vec2[r] * fac1[r] + vec3[r] * fac2[r]) * sign_a[r];
    inv[1][r] = (vec0[r] * fac0[r] - vec2[r] * fac3[r] + vec3[r] * fac4[r]) * sign_b[r];
    inv[2][r] = (vec0[r] * fac1[r] - vec1[r] * fac3[r] + vec3[r] * fac5[r]) * sign_a[r];
    inv[3][r] = (vec0[r] * fac2[r] - vec1[r] * fac4[r] + vec2[r] * fac5[r]) * sign_b[r];
  }
  const float d0 = m[0][0] * inv[0][0], d1 = m[0][1] * inv[1][0], d2 = m[0][2] * inv[2][0], d3 = m[0][3] * inv[3][0];
  const float one_over_det = 1.0f / ((d0 + d1) + (d2 + d3));
  // transpose(inverse(M))[k] = row k of the inverse
  nova_triangle_data td;
  td.x = { inv[0][0] * one_over_det, inv[1][0] * one_over_det, inv[2][0] * one_over_det, inv[3][0] * one_over_det };
  td.y = { inv[0][1] * one_over_det, inv[1][1] * one_over_det, inv[2][1] * one_over_det, inv[3][1] * one_over_det };
  td.z = { inv[0][2] * one_over_det, inv[1][2] * one_over_det, inv[2][2] * one_over_det, inv[3][2] * one_over_det };
  tris[p] = td;
}

// ---- host driver -------------------------------------------------------------------------------------
struct DeviceBuffer {
  void* p = nullptr;
  size_t bytes = 0;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) {
      (void) hipFree(p);
      p = o.p;
      bytes = o.bytes;
      o.p = nullptr;
      o.bytes = 0;
    }
    return *this;
  }
  ~DeviceBuffer() { (void) hipFree(p); }
  hipError_t reserve(size_t want) {
    if (want <= bytes) return hipSuccess;
    (void) hipFree(p);
    p = nullptr;
    bytes = 0;
    const hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) bytes = want;
    return e;
  }
  template <typename T>
  T* as() const { return static_cast<T*>(p); }
};

// upper bound on the nodes of a tree over n triangles: every split leaves more than 2 on each side
inline uint64_t max_nodes(uint64_t n) {
  const uint64_t leaves = n / 3 > 1 ? n / 3 : 1;
  return 2 * leaves - 1;
}

#define BVHB_TRY(expr)                                   \
  do {                                                   \
    const hipError_t e_ = (expr);                        \
    if (e_ != hipSuccess) return static_cast<int>(e_);   \
  } while (0)

struct Scanner {
  DeviceBuffer sums;
  uint32_t* d_total = nullptr; // device word
  hipStream_t stream = nullptr;
  // out[i] = sum of in[0..i); the total lands in *d_total
  int run(const uint32_t* in, uint32_t n, uint32_t* out, uint32_t* total = nullptr) {
    const uint32_t nb = (n + kScanTile - 1) / kScanTile;
    if (nb <= 8) {
      k_scan_small<<<1, kScanBlock, 0, stream>>>(in, n, out, total ? total : d_total);
      return static_cast<int>(hipGetLastError());
    }
    BVHB_TRY(sums.reserve(static_cast<size_t>(nb) * 4));
    k_scan_reduce<<<nb, kScanBlock, 0, stream>>>(in, n, sums.as<uint32_t>());
    k_scan_sums<<<1, kScanBlock, 0, stream>>>(sums.as<uint32_t>(), nb, total ? total : d_total);
    k_scan_apply<<<nb, kScanBlock, 0, stream>>>(in, n, sums.as<uint32_t>(), out);
    return static_cast<int>(hipGetLastError());
  }
};

struct Result {
  uint32_t n_nodes = 0, levels = 0;
  float device_ms = 0.0f;
  float pack_ms = 0.0f; // traversal layout made from the tree (bvh_pack_device.hpp), host clock
  size_t peak_bin_bytes = 0;
};

// The builder's own tree, for a caller that goes on with it on the device (bvh_pack_device.hpp)
struct Tree {
  DeviceBuffer nodes;               // BNode[n_nodes]; left = the child visited first
  DeviceBuffer flat_index;          // index of each node in the FlatBVHNode array
  DeviceBuffer tri_offset;          // first triangle (output order) of each node's subtree
  // stage s = the id range [level_start[s], level_start[s + 1]): a layering of the tree, every node in a later
  // stage than its parent (the levels of the top of the tree, then the local levels below the small roots)
  std::vector<uint32_t> level_start;
  uint32_t n_nodes = 0;
  uint32_t depth = 0;               // depth of the deepest node (root = 0)
};

// d_verts: n*9 floats on the device.  d_bvh: room for max_nodes(n) records.  d_tris, d_tri_order: n.
inline int build(hipStream_t stream, const float* d_verts, uint32_t n, nova_bvh_node* d_bvh,
                 nova_triangle_data* d_tris, uint32_t* d_tri_order, Result* result, Tree* keep = nullptr) {
  constexpr uint32_t kB = 256;
  auto grid = [](uint64_t items) { return static_cast<uint32_t>((items + kB - 1) / kB); };
  const uint64_t cap = max_nodes(n);

  DeviceBuffer tbox, order[2], node_of[2], nodes, misc, flags, ranks, slots, items, item_base, multi_list, bins,
    partials, part_owner, pbins, axis_best, choice, gl, gls, flat_index, tri_offset;
  DeviceBuffer roots, tmp_nodes, lvl_first, lvl_base, root_levels, stage_words, d_level_start;
  BVHB_TRY(tbox.reserve(static_cast<size_t>(n) * 32));
  for (int i = 0; i < 2; i++) {
    BVHB_TRY(order[i].reserve(static_cast<size_t>(n) * 4));
    BVHB_TRY(node_of[i].reserve(static_cast<size_t>(n) * 4));
  }
  BVHB_TRY(nodes.reserve(cap * sizeof(BNode)));
  BVHB_TRY(misc.reserve(64 + 4 * kMaxBlindLevels + 16));
  BVHB_TRY(gl.reserve(static_cast<size_t>(n) * 4));
  BVHB_TRY(gls.reserve(static_cast<size_t>(n) * 4));
  BVHB_TRY(pbins.reserve(static_cast<size_t>(n) * 4));
  uint32_t* d_root = misc.as<uint32_t>();      // 6 words
  uint32_t* d_total = misc.as<uint32_t>() + 8; // scan total
  uint32_t* d_multi_count = misc.as<uint32_t>() + 9; // + 10: slabs handed out
  uint32_t* d_n_slots = misc.as<uint32_t>() + 11;
  uint32_t* d_n_items = misc.as<uint32_t>() + 12;
  uint32_t* d_lv = misc.as<uint32_t>() + 16; // kMaxBlindLevels + 3 words: the level list of the blind levels
  uint32_t* h_total = nullptr;
  BVHB_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_total), 64 + 4 * kMaxBlindLevels + 16));
  struct Pinned {
    uint32_t* p;
    ~Pinned() { (void) hipHostFree(p); }
  } pinned { h_total };
  Scanner scan;
  scan.d_total = d_total;
  scan.stream = stream;
  auto read_total = [&](uint32_t* out) -> int {
    BVHB_TRY(hipMemcpyAsync(h_total, d_total, 4, hipMemcpyDeviceToHost, stream));
    BVHB_TRY(hipStreamSynchronize(stream));
    *out = *h_total;
    return 0;
  };

  // NOVA_HIP_BVH_SMALL=0: every level by the level loop (the round-1 builder; A/B and fallback)
  static const uint32_t small_tris = [] {
    const char* e = std::getenv("NOVA_HIP_BVH_SMALL");
    return (e && e[0] == '0') ? 0u : kSmallTris;
  }();
  // Everything the level loop needs, reserved before the clock starts: a buffer that grows inside the loop
  // costs a hipFree (a device-wide wait) and a hipMalloc per level.  With the small-subtree path a level of the
  // top tree holds at most ~2 n / kSmallTris nodes above kSmallMinDepth; the in-loop reserves stay as a backstop.
  {
    const uint64_t m_bound = small_tris ? std::min<uint64_t>(cap, 2ull * (n / small_tris) + (2ull << kSmallMinDepth)) : cap;
    BVHB_TRY(flags.reserve(m_bound * 4));
    BVHB_TRY(ranks.reserve(m_bound * 4));
    BVHB_TRY(slots.reserve(m_bound * sizeof(Slot)));
    BVHB_TRY(items.reserve(m_bound * 4));
    BVHB_TRY(item_base.reserve(m_bound * 4));
    BVHB_TRY(axis_best.reserve(m_bound * 3 * sizeof(AxisBest)));
    BVHB_TRY(choice.reserve(m_bound * sizeof(Choice)));
    size_t max_part = 0, max_bins = 0, max_owner = 0, max_multi_list = 0;
    for (uint32_t d = 0; d < 64; d++) {
      const uint32_t nb = static_cast<uint32_t>(kMaxBins / static_cast<float>(d + 1));
      if (nb <= 1) break;
      uint32_t chunk = 16 * nb;
      const uint32_t spread = (n / 2048 + 63) & ~63u;
      if (chunk > spread) chunk = spread;
      if (chunk < kMinChunk) chunk = kMinChunk;
      const uint64_t level_nodes = std::min<uint64_t>(m_bound, 1ull << std::min<uint32_t>(d, 40u));
      const bool all_multi = 39ull * nb * 4 > 65536; // as in the level loop
      const uint64_t mm = all_multi ? level_nodes : std::min<uint64_t>(n / chunk, level_nodes);
      const uint64_t parts = mm ? 2ull * (n / chunk) + 1 + (all_multi ? level_nodes : 0) : 0;
      max_part = std::max<size_t>(max_part, parts * 21 * nb * 4 + 16);
      max_owner = std::max<size_t>(max_owner, parts * 4 + 16);
      max_bins = std::max<size_t>(max_bins, mm * 3 * nb * sizeof(Bin) + 16);
      max_multi_list = std::max<size_t>(max_multi_list, (mm + 1) * 4);
    }
    BVHB_TRY(partials.reserve(max_part));
    BVHB_TRY(part_owner.reserve(max_owner));
    BVHB_TRY(bins.reserve(max_bins));
    BVHB_TRY(multi_list.reserve(max_multi_list));
    BVHB_TRY(scan.sums.reserve((static_cast<size_t>(n) / kScanTile + 2) * 4));
    if (small_tris) { // a small root has more than 8 triangles; its region holds max_nodes(its triangles) records
      const size_t max_roots = n / (kMinTrianglesPerLeaf + 1) + 1;
      BVHB_TRY(roots.reserve(max_roots * sizeof(SmallRoot)));
      BVHB_TRY(root_levels.reserve(max_roots * 4));
      BVHB_TRY(tmp_nodes.reserve(cap * sizeof(BNode)));
      BVHB_TRY(lvl_first.reserve(cap * 4));
      BVHB_TRY(lvl_base.reserve(cap * 4));
      BVHB_TRY(stage_words.reserve((2 * kMaxStages + 1) * 4));
      BVHB_TRY(d_level_start.reserve(1024 * 4));
      BVHB_TRY(flat_index.reserve(cap * 4));
      BVHB_TRY(tri_offset.reserve(cap * 4));
    }
  }
  hipEvent_t ev0, ev1;
  BVHB_TRY(hipEventCreate(&ev0));
  BVHB_TRY(hipEventCreate(&ev1));
  struct Events {
    hipEvent_t a, b;
    ~Events() {
      (void) hipEventDestroy(a);
      (void) hipEventDestroy(b);
    }
  } events { ev0, ev1 };
  BVHB_TRY(hipEventRecord(ev0, stream));

  BVHB_TRY(hipMemsetAsync(d_root, 0, 64, stream));
  k_tri_bounds<<<grid(n) < 1024u ? grid(n) : 1024u, kB, 0, stream>>>(d_verts, n, tbox.as<float4>(), d_root, order[0].as<uint32_t>(),
                                          node_of[0].as<uint32_t>());
  k_root<<<1, 1, 0, stream>>>(d_root, n, nodes.as<BNode>(), d_lv);

  // Blind levels.  The host needs a level's node count only to size grids and to stop; waiting for it costs a round
  // trip per level (~25 us of a ~150 us level).  The first levels of a big build are sure to split — a node leaves
  // the loop at kSmallTris triangles, i.e. around depth log2(n / kSmallTris) — and hold at most 2^depth nodes, so
  // they are enqueued with that bound for the grids while ls / le stay on the device (d_lv); the host reads the
  // whole list once, when it takes over.  A tree that ended earlier just ran a few empty levels.
  uint32_t blind_levels = 0;
  if (small_tris) {
    while ((static_cast<uint64_t>(small_tris) << (blind_levels + 1)) <= n && (2u << blind_levels) <= kFusedLevelNodes &&
           blind_levels + 1 < kMaxBlindLevels)
      blind_levels++;
    static const bool off = std::getenv("NOVA_HIP_BVH_NO_BLIND") != nullptr; // A/B
    if (off) blind_levels = 0;
  }

  std::vector<uint32_t> level_start { 0u };
  uint32_t ls = 0, le = 1; // nodes of the current level
  int cur = 0;
  size_t peak_bins = 0;
  for (uint32_t depth = 0;; depth++) {
    const bool blind = depth < blind_levels;
    if (!blind && depth == blind_levels && blind_levels > 0) {
      // take over: lv[d] .. lv[d + 1] = level d for d <= depth; a level without nodes = the tree ended before it
      BVHB_TRY(hipMemcpyAsync(h_total + 16, d_lv, (depth + 2) * 4, hipMemcpyDeviceToHost, stream));
      BVHB_TRY(hipStreamSynchronize(stream));
      const uint32_t* lv = h_total + 16;
      uint32_t last = 0; // deepest level that has nodes
      while (last < depth && lv[last + 2] > lv[last + 1]) last++;
      level_start.assign(lv, lv + last + 1);
      ls = lv[last];
      le = lv[last + 1];
      if (last < depth) { // ended inside the blind levels (its last level split nothing)
        level_start.push_back(le);
        break;
      }
    }
    level_start.push_back(le);
    const uint32_t m = blind ? (1u << depth) : le - ls; // blind: an upper bound
    const uint32_t* lvp = blind ? d_lv + depth : nullptr;
    const uint32_t num_bins = static_cast<uint32_t>(kMaxBins / static_cast<float>(depth + 1)); // bvh.cpp:145
    if (num_bins <= 1) break;
    BVHB_TRY(flags.reserve(static_cast<size_t>(m) * 4));
    BVHB_TRY(ranks.reserve(static_cast<size_t>(m) * 4));
    const bool fused_level = m <= kFusedLevelNodes; // the small kernels of a level as two one-group launches
    int rc = 0;
    if (!fused_level) {
      k_level_flags<<<grid(m), kB, 0, stream>>>(nodes.as<BNode>(), ls, m, depth >= kSmallMinDepth ? small_tris : 0u,
                                               flags.as<uint32_t>());
      // the number of nodes to split stays on the device (d_n_slots); the host sizes everything by m
      rc = scan.run(flags.as<uint32_t>(), m, ranks.as<uint32_t>(), d_n_slots);
      if (rc) return rc;
    }
    const uint32_t n_slots = m; // upper bound

    // work items: `chunk` consecutive positions of one node; a node of more than one item has global bins
    // 16 x bins keeps the slab traffic below the triangles' own; fewer where that would leave the GPU short
    // of waves (about 2048 items wanted over the whole array)
    uint32_t chunk = 16 * num_bins;
    const uint32_t spread = (n / 2048 + 63) & ~63u;
    if (chunk > spread) chunk = spread;
    if (chunk < kMinChunk) chunk = kMinChunk;
    // The two topmost levels have too many bins for a work item to hold three axes of them AND the suffix boxes an
    // evaluation on the spot needs (39 * num_bins words of LDS).  Binning one axis per pass (k_bin_sah<false>) reads
    // every triangle three times; instead every node of such a level is treated like one of several items: three
    // axes in one pass (21 * num_bins words), slabs, evaluation from the global bins.
    const bool all_multi = fused_level && 39ull * num_bins * 4 > 65536;
    const uint32_t max_multi = all_multi ? n_slots : (n / chunk < n_slots ? n / chunk : n_slots); // nodes with global bins
    const uint32_t max_items = n / chunk + n_slots;
    // ceil(n_s / chunk) <= 2 n_s / chunk for n_s > chunk, one slab for a smaller node
    const uint32_t max_partials = max_multi ? 2 * (n / chunk) + 1 + (all_multi ? n_slots : 0) : 0;
    const size_t n_bins = static_cast<size_t>(max_multi) * 3 * num_bins;
    BVHB_TRY(slots.reserve(static_cast<size_t>(n_slots) * sizeof(Slot)));
    BVHB_TRY(items.reserve(static_cast<size_t>(n_slots) * 4));
    BVHB_TRY(item_base.reserve(static_cast<size_t>(n_slots) * 4));
    BVHB_TRY(multi_list.reserve(static_cast<size_t>(max_multi + 1) * 4));
    BVHB_TRY(bins.reserve(n_bins * sizeof(Bin) + 16));
    BVHB_TRY(partials.reserve(static_cast<size_t>(max_partials) * 21 * num_bins * 4 + 16));
    BVHB_TRY(part_owner.reserve(static_cast<size_t>(max_partials) * 4 + 16));
    BVHB_TRY(axis_best.reserve(static_cast<size_t>(n_slots) * 3 * sizeof(AxisBest)));
    BVHB_TRY(choice.reserve(static_cast<size_t>(n_slots) * sizeof(Choice)));
    const size_t bin_bytes = n_bins * sizeof(Bin) + static_cast<size_t>(max_partials) * 21 * num_bins * 4;
    peak_bins = bin_bytes > peak_bins ? bin_bytes : peak_bins;
    if (fused_level) {
      k_level_setup<<<1, kScanBlock, 0, stream>>>(nodes.as<BNode>(), ls, m, depth >= kSmallMinDepth ? small_tris : 0u, num_bins,
                                                 chunk, all_multi, slots.as<Slot>(), item_base.as<uint32_t>(), d_n_slots, d_n_items,
                                                 d_multi_count, multi_list.as<uint32_t>(), d_multi_count + 1, lvp);
    } else {
      BVHB_TRY(hipMemsetAsync(d_multi_count, 0, 8, stream));
      BVHB_TRY(hipMemsetAsync(items.p, 0, static_cast<size_t>(m) * 4, stream)); // nodes that are not split: no items
      k_slot_setup<<<grid(m), kB, 0, stream>>>(nodes.as<BNode>(), ls, m, flags.as<uint32_t>(), ranks.as<uint32_t>(),
                                              num_bins, chunk, slots.as<Slot>(), items.as<uint32_t>(), d_multi_count,
                                              multi_list.as<uint32_t>(), d_multi_count + 1);
      rc = scan.run(items.as<uint32_t>(), n_slots, item_base.as<uint32_t>(), d_n_items);
      if (rc) return rc;
    }
    if (n_bins) BVHB_TRY(hipMemsetAsync(bins.p, 0, n_bins * sizeof(Bin), stream));
    if (all_multi) {
      static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_sah<true>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 21 * 1024 * 4);
      BVHB_TRY(attr);
      k_bin_sah<true><<<max_items, kBinWaves * 64, 21 * num_bins * 4, stream>>>(
        order[cur].as<uint32_t>(), nodes.as<BNode>(), slots.as<Slot>(), d_n_slots, item_base.as<uint32_t>(), d_n_items, chunk,
        tbox.as<float4>(), num_bins, partials.as<uint32_t>(), part_owner.as<uint32_t>(), pbins.as<uint32_t>(),
        axis_best.as<AxisBest>());
    } else if (39 * num_bins * 4 <= 65536) // the bins of three axes and three sets of suffix boxes fit the LDS of one block
      k_bin_sah<true><<<max_items, kBinWaves * 64, 39 * num_bins * 4, stream>>>(
        order[cur].as<uint32_t>(), nodes.as<BNode>(), slots.as<Slot>(), d_n_slots, item_base.as<uint32_t>(), d_n_items, chunk,
        tbox.as<float4>(), num_bins, partials.as<uint32_t>(), part_owner.as<uint32_t>(), pbins.as<uint32_t>(),
        axis_best.as<AxisBest>());
    else
      k_bin_sah<false><<<max_items, kBinWaves * 64, 13 * num_bins * 4, stream>>>(
        order[cur].as<uint32_t>(), nodes.as<BNode>(), slots.as<Slot>(), d_n_slots, item_base.as<uint32_t>(), d_n_items, chunk,
        tbox.as<float4>(), num_bins, partials.as<uint32_t>(), part_owner.as<uint32_t>(), pbins.as<uint32_t>(),
        axis_best.as<AxisBest>());
    if (max_multi) {
      const dim3 rg((21 * num_bins + kB - 1) / kB, (max_partials + kReduceGroup - 1) / kReduceGroup);
      k_reduce_partials<<<rg, kB, 0, stream>>>(partials.as<uint32_t>(), part_owner.as<uint32_t>(), d_multi_count + 1,
                                              slots.as<Slot>(), multi_list.as<uint32_t>(), num_bins, bins.as<Bin>());
      k_sah_multi<<<max_multi * 3, 64, 6 * num_bins * 4, stream>>>(slots.as<Slot>(), multi_list.as<uint32_t>(),
                                                                  d_multi_count, num_bins, bins.as<Bin>(),
                                                                  axis_best.as<AxisBest>());
    }
    // flags / ranks are reused per slot from here on (n_slots <= m)
    if (fused_level) {
      k_choose_scan<<<1, kScanBlock, 0, stream>>>(slots.as<Slot>(), d_n_slots, nodes.as<BNode>(), axis_best.as<AxisBest>(),
                                                 choice.as<Choice>(), ranks.as<uint32_t>(), d_total,
                                                 blind ? d_lv + depth : nullptr);
    } else {
      k_choose<<<grid(n_slots), kB, 0, stream>>>(slots.as<Slot>(), d_n_slots, n_slots, nodes.as<BNode>(),
                                                axis_best.as<AxisBest>(), choice.as<Choice>(), flags.as<uint32_t>());
      rc = scan.run(flags.as<uint32_t>(), n_slots, ranks.as<uint32_t>());
      if (rc) return rc;
    }
    uint32_t n_split = 0;
    if (!blind) {
      rc = read_total(&n_split);
      if (rc) return rc;
      if (n_split == 0) break;
      if (static_cast<uint64_t>(le) + 2ull * n_split > cap) return NOVA_HIP_E_BAD_SCENE; // cannot happen
    }
    k_make_children<<<grid(n_slots), kB, 0, stream>>>(slots.as<Slot>(), d_n_slots, nodes.as<BNode>(),
                                                     axis_best.as<AxisBest>(), choice.as<Choice>(),
                                                     ranks.as<uint32_t>(), le, lvp);
    const uint32_t gl_blocks = (n + kScanTile - 1) / kScanTile;
    if (gl_blocks > 8) { // the flags are made inside the scan's first pass
      BVHB_TRY(scan.sums.reserve(static_cast<size_t>(gl_blocks) * 4));
      k_gl_scan_reduce<<<gl_blocks, kScanBlock, 0, stream>>>(node_of[cur].as<uint32_t>(), n, nodes.as<BNode>(), ls,
                                                            choice.as<Choice>(), pbins.as<uint32_t>(), gl.as<uint32_t>(),
                                                            scan.sums.as<uint32_t>(), lvp);
      k_scan_sums<<<1, kScanBlock, 0, stream>>>(scan.sums.as<uint32_t>(), gl_blocks, d_total);
      k_scan_apply<<<gl_blocks, kScanBlock, 0, stream>>>(gl.as<uint32_t>(), n, scan.sums.as<uint32_t>(), gls.as<uint32_t>());
    } else {
      k_goes_left<<<grid(n), kB, 0, stream>>>(node_of[cur].as<uint32_t>(), n, nodes.as<BNode>(), ls, choice.as<Choice>(),
                                             pbins.as<uint32_t>(), gl.as<uint32_t>(), lvp);
      rc = scan.run(gl.as<uint32_t>(), n, gls.as<uint32_t>());
      if (rc) return rc;
    }
    k_scatter<<<grid(n), kB, 0, stream>>>(order[cur].as<uint32_t>(), node_of[cur].as<uint32_t>(), n, nodes.as<BNode>(),
                                         ls, choice.as<Choice>(), gl.as<uint32_t>(), gls.as<uint32_t>(),
                                         order[cur ^ 1].as<uint32_t>(), node_of[cur ^ 1].as<uint32_t>(), lvp);
    cur ^= 1;
    if (!blind) {
      ls = le;
      le += 2 * n_split;
    }
  }
  // level_start[d] .. level_start[d + 1] are the nodes of depth d of the top of the tree
  const uint32_t n_top = le;
  const uint32_t n_top_levels = static_cast<uint32_t>(level_start.size()) - 1;
  uint32_t n_nodes = n_top, tree_depth = n_top_levels - 1, n_roots = 0;
  if (small_tris) {
    // the small roots: nodes of the top tree that the level loop left alone
    BVHB_TRY(flags.reserve(static_cast<size_t>(n_top) * 4));
    BVHB_TRY(ranks.reserve(static_cast<size_t>(n_top) * 4));
    BVHB_TRY(gl.reserve(static_cast<size_t>(n_top) * 4));
    BVHB_TRY(gls.reserve(static_cast<size_t>(n_top) * 4));
    k_small_flags<<<grid(n_top), kB, 0, stream>>>(nodes.as<BNode>(), n_top, flags.as<uint32_t>(), gl.as<uint32_t>());
    int rc = scan.run(flags.as<uint32_t>(), n_top, ranks.as<uint32_t>(), d_n_slots);
    if (rc) return rc;
    rc = scan.run(gl.as<uint32_t>(), n_top, gls.as<uint32_t>(), d_n_items);
    if (rc) return rc;
    BVHB_TRY(hipMemcpyAsync(h_total, d_n_slots, 8, hipMemcpyDeviceToHost, stream)); // d_n_slots, d_n_items are adjacent
    BVHB_TRY(hipStreamSynchronize(stream));
    n_roots = h_total[0];
    const uint32_t total_cap = h_total[1];
    if (n_roots) {
      BVHB_TRY(roots.reserve(static_cast<size_t>(n_roots) * sizeof(SmallRoot)));
      BVHB_TRY(tmp_nodes.reserve(static_cast<size_t>(total_cap) * sizeof(BNode)));
      BVHB_TRY(lvl_first.reserve(static_cast<size_t>(total_cap) * 4));
      BVHB_TRY(lvl_base.reserve(static_cast<size_t>(total_cap) * 4));
      BVHB_TRY(root_levels.reserve(static_cast<size_t>(n_roots) * 4));
      BVHB_TRY(stage_words.reserve((2 * kMaxStages + 1) * 4)); // counts, first ids, deepest node
      BVHB_TRY(d_level_start.reserve(level_start.size() * 4));
      BVHB_TRY(hipMemcpyAsync(d_level_start.p, level_start.data(), level_start.size() * 4, hipMemcpyHostToDevice, stream));
      BVHB_TRY(hipMemsetAsync(stage_words.p, 0, (2 * kMaxStages + 1) * 4, stream));
      uint32_t* d_stage_count = stage_words.as<uint32_t>();
      uint32_t* d_stage_first = d_stage_count + kMaxStages;
      uint32_t* d_max_depth = d_stage_count + 2 * kMaxStages;
      k_small_list<<<grid(n_top), kB, 0, stream>>>(nodes.as<BNode>(), n_top, flags.as<uint32_t>(), ranks.as<uint32_t>(),
                                                  gls.as<uint32_t>(), d_level_start.as<uint32_t>(), n_top_levels,
                                                  roots.as<SmallRoot>());
      constexpr size_t kSubtreeLds = (3 * kSmallTris + kSmallWaves * 27 * kSmallBins) * 4;
      static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_subtree),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kSubtreeLds));
      BVHB_TRY(attr);
      k_subtree<<<n_roots, kSmallWaves * 64, kSubtreeLds, stream>>>(
        roots.as<SmallRoot>(), nodes.as<BNode>(), tbox.as<float4>(), order[cur].as<uint32_t>(), tmp_nodes.as<BNode>(),
        lvl_first.as<uint32_t>(), lvl_base.as<uint32_t>(), root_levels.as<uint32_t>(), d_stage_count, d_max_depth);
      BVHB_TRY(hipGetLastError());
      std::vector<uint32_t> stage(2 * kMaxStages + 1);
      BVHB_TRY(hipMemcpyAsync(stage.data(), stage_words.p, stage.size() * 4, hipMemcpyDeviceToHost, stream));
      BVHB_TRY(hipStreamSynchronize(stream));
      // stage l >= 1 = the nodes of local level l of every root, after the top of the tree
      std::vector<uint32_t> first(kMaxStages, 0u);
      for (uint32_t l = 1; l < kMaxStages && stage[l]; l++) {
        first[l] = n_nodes;
        n_nodes += stage[l];
        level_start.push_back(n_nodes);
      }
      if (n_nodes > cap) return NOVA_HIP_E_BAD_SCENE; // cannot happen
      tree_depth = stage[2 * kMaxStages] > tree_depth ? stage[2 * kMaxStages] : tree_depth;
      BVHB_TRY(hipMemcpyAsync(d_stage_first, first.data(), kMaxStages * 4, hipMemcpyHostToDevice, stream));
      k_subtree_place<<<n_roots, 256, 0, stream>>>(roots.as<SmallRoot>(), tmp_nodes.as<BNode>(), lvl_first.as<uint32_t>(),
                                                  lvl_base.as<uint32_t>(), root_levels.as<uint32_t>(), d_stage_first,
                                                  nodes.as<BNode>(), node_of[cur].as<uint32_t>());
    }
  }
  // subtree sizes of the top of the tree (those below the small roots are set), then flattening top-down
  for (uint32_t d = n_top_levels; d-- > 0;) {
    const uint32_t s = level_start[d], m = level_start[d + 1] - s;
    if (m) k_sizes<<<grid(m), kB, 0, stream>>>(nodes.as<BNode>(), s, m);
  }
  BVHB_TRY(flat_index.reserve(static_cast<size_t>(n_nodes) * 4));
  BVHB_TRY(tri_offset.reserve(static_cast<size_t>(n_nodes) * 4));
  BVHB_TRY(hipMemsetAsync(flat_index.p, 0, 4, stream));
  BVHB_TRY(hipMemsetAsync(tri_offset.p, 0, 4, stream));
  for (uint32_t d = 0; d < n_top_levels; d++) {
    const uint32_t s = level_start[d], m = level_start[d + 1] - s;
    if (m)
      k_flatten<<<grid(m), kB, 0, stream>>>(nodes.as<BNode>(), s, m, flat_index.as<uint32_t>(),
                                           tri_offset.as<uint32_t>(), d_bvh);
  }
  if (n_roots)
    k_subtree_flatten<<<n_roots, 256, 0, stream>>>(roots.as<SmallRoot>(), lvl_first.as<uint32_t>(), lvl_base.as<uint32_t>(),
                                                  root_levels.as<uint32_t>(), stage_words.as<uint32_t>() + kMaxStages,
                                                  nodes.as<BNode>(), flat_index.as<uint32_t>(), tri_offset.as<uint32_t>(), d_bvh);
  k_emit_triangles<<<grid(n), kB, 0, stream>>>(order[cur].as<uint32_t>(), node_of[cur].as<uint32_t>(), n,
                                              nodes.as<BNode>(), tri_offset.as<uint32_t>(), d_verts, d_tri_order, d_tris);
  BVHB_TRY(hipGetLastError());
  BVHB_TRY(hipEventRecord(ev1, stream));
  BVHB_TRY(hipStreamSynchronize(stream));
  if (result) {
    result->n_nodes = n_nodes;
    result->levels = tree_depth + 1;
    result->peak_bin_bytes = peak_bins;
    BVHB_TRY(hipEventElapsedTime(&result->device_ms, ev0, ev1));
  }
  if (keep) {
    keep->nodes = std::move(nodes);
    keep->flat_index = std::move(flat_index);
    keep->tri_offset = std::move(tri_offset);
    keep->level_start = level_start;
    keep->n_nodes = n_nodes;
    keep->depth = tree_depth;
  }
  return 0;
}

} // namespace bvhb
