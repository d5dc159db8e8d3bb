// The traversal layout straight from the device BVH build: what bvh_pack.hpp (Packer, the slack sweep,
// WideBuilder) makes on the host from a FlatBVHNode[] copy, made on the device from the builder's own
// tree (bvh_build.hpp: Tree) — the same Node64 / Node128 arrays, word for word.  Used for scenes built
// by nova_hip_build_bvh_device; a FlatBVHNode[] that comes from the caller still takes the host path.
//
// Numbering is the host's: box-pair nodes = inner nodes in DFS pre-order, leaves in DFS order, wide nodes
// in DFS pre-order of the collapsed tree.  With subtree sizes known (a binary tree of `size` nodes has
// (size - 1) / 2 inner nodes and (size + 1) / 2 leaves) every rank is a top-down sum, level by level.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "bvh_build.hpp"
#include "traverse.hpp"

namespace nova_hip {
namespace packdev {

using bvhb::BNode;
using bvhb::DeviceBuffer;

// DFS ranks of the children among inner nodes (irank) and among leaves (lrank)
__global__ void k_ranks(const BNode* __restrict__ nodes, uint32_t ls, uint32_t m, uint32_t* __restrict__ irank,
                        uint32_t* __restrict__ lrank) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const BNode& nd = nodes[ls + i];
  if (!nd.left) return;
  const uint32_t me_i = irank[ls + i], me_l = lrank[ls + i];
  const uint32_t first_size = nodes[nd.left].size;
  irank[nd.left] = me_i + 1;
  irank[nd.right] = me_i + 1 + (first_size - 1) / 2;
  lrank[nd.left] = me_l;
  lrank[nd.right] = me_l + (first_size + 1) / 2;
}

__device__ __forceinline__ uint32_t leaf_ref(uint32_t first, uint32_t count) {
  return kLeafBit | ((count - 1u) << 24) | first;
}

// Packer::ref_of / set_child for every inner node at once; leaf records for k_leaf_slack
__global__ void k_nodes64(const BNode* __restrict__ nodes, uint32_t n_nodes, const uint32_t* __restrict__ irank,
                          const uint32_t* __restrict__ lrank, const uint32_t* __restrict__ tri_offset,
                          Node64* __restrict__ out, LeafRec* __restrict__ leaves, uint32_t* __restrict__ oversize) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const BNode& nd = nodes[i];
  if (!nd.left) return;
  uint32_t ref[2];
  const uint32_t child[2] = { nd.left, nd.right };
  for (int k = 0; k < 2; k++) {
    const BNode& c = nodes[child[k]];
    if (c.left) {
      ref[k] = irank[child[k]];
    } else {
      const uint32_t first = tri_offset[child[k]], count = c.end - c.begin;
      if (count > kMaxLeafTris) atomicOr(oversize, 1u); // the host packer chains such a leaf: left to it
      ref[k] = leaf_ref(first, count);
      LeafRec r;
      r.first = first;
      r.count = count;
      for (int a = 0; a < 3; a++) {
        r.top[a] = c.top[a];
        r.bot[a] = c.bot[a];
      }
      leaves[lrank[child[k]]] = r;
    }
  }
  const BNode& l = nodes[nd.left];
  const BNode& r = nodes[nd.right];
  Node64 o;
  o.a = make_float4(l.top[0], l.top[1], l.top[2], __uint_as_float(ref[0]));
  o.b = make_float4(l.bot[0], l.bot[1], l.bot[2], __uint_as_float(ref[1]));
  o.c = make_float4(r.top[0], r.top[1], r.top[2], 0.0f);
  o.d = make_float4(r.bot[0], r.bot[1], r.bot[2], 0.0f);
  out[irank[i]] = o;
}

// cull slack of a subtree = max over its leaves, one level at a time from the deepest (get_scene's sweep)
__global__ void k_slack_up(const BNode* __restrict__ nodes, uint32_t ls, uint32_t m, const uint32_t* __restrict__ irank,
                           const uint32_t* __restrict__ lrank, const float* __restrict__ leaf_e,
                           float* __restrict__ node_e, float e_round, Node64* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const BNode& nd = nodes[ls + i];
  if (!nd.left) return;
  float e = 0.0f;
  const uint32_t child[2] = { nd.left, nd.right };
  for (int k = 0; k < 2; k++) {
    const float x = nodes[child[k]].left ? node_e[irank[child[k]]] : leaf_e[lrank[child[k]]];
    e = (e < x) ? x : e;
  }
  node_e[irank[ls + i]] = e;
  out[irank[ls + i]].c.w = e + e_round;
}

// ---- four-wide tree (WideBuilder) ---------------------------------------------------------------------
struct WItem {
  float top[3], bot[3];
  uint32_t ref; // leaf reference, or (inner) first the BNode id, then the id of the child's record
  float e;
  uint32_t inner;
};
struct WRec {
  WItem it[4];
  uint32_t count, wsize, slot, pad;
};

struct WideCtx {
  const BNode* nodes;
  const uint32_t* irank;
  const uint32_t* lrank;
  const uint32_t* tri_offset;
  const float* leaf_e;
  const float* node_e;
  float e_round;
};

__device__ __forceinline__ void children_of(const WideCtx& C, uint32_t b, WItem* items, int& n) {
  const BNode& nd = C.nodes[b];
  const uint32_t child[2] = { nd.left, nd.right };
  for (int k = 0; k < 2; k++) {
    const BNode& c = C.nodes[child[k]];
    WItem it;
    for (int a = 0; a < 3; a++) {
      it.top[a] = c.top[a];
      it.bot[a] = c.bot[a];
    }
    it.inner = c.left ? 1u : 0u;
    it.ref = it.inner ? child[k] : leaf_ref(C.tri_offset[child[k]], c.end - c.begin);
    it.e = (it.inner ? C.node_e[C.irank[child[k]]] : C.leaf_e[C.lrank[child[k]]]) + C.e_round;
    items[n++] = it;
  }
}
__device__ __forceinline__ float item_area(const WItem& it) {
  const float dx = it.top[0] - it.bot[0], dy = it.top[1] - it.bot[1], dz = it.top[2] - it.bot[2];
  const float a = dx * dy + dy * dz + dz * dx;
  return a == a ? a : 0.0f;
}

// One wide level: every record of [first, first + count) collapses its subtree top and appends a record
// per inner child (WideBuilder::build up to the recursion)
__global__ void k_wide_expand(WideCtx C, WRec* __restrict__ recs, uint32_t* __restrict__ rec_bnode,
                              uint32_t* __restrict__ rec_above, uint32_t first, uint32_t count,
                              uint32_t* __restrict__ n_recs, uint32_t* __restrict__ stack_need) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t r = first + i;
  WItem items[5];
  int n = 0;
  children_of(C, rec_bnode[r], items, n);
  while (n < 4) {
    int best = -1;
    float best_area = -1.0f;
    for (int k = 0; k < n; k++) {
      if (items[k].inner) {
        const float a = item_area(items[k]);
        if (a > best_area) {
          best = k;
          best_area = a;
        }
      }
    }
    if (best < 0) break;
    const uint32_t child = items[best].ref;
    for (int k = best; k + 1 < n; k++) items[k] = items[k + 1];
    n--;
    children_of(C, child, items, n);
  }
  const uint32_t held = rec_above[r] + static_cast<uint32_t>(n) - 1u;
  atomicMax(stack_need, held);
  WRec out;
  out.count = static_cast<uint32_t>(n);
  out.wsize = 1;
  out.slot = 0;
  out.pad = 0;
  for (int k = 0; k < 4; k++) {
    if (k < n) {
      out.it[k] = items[k];
      if (items[k].inner) {
        const uint32_t id = atomicAdd(n_recs, 1u);
        rec_bnode[id] = items[k].ref;
        rec_above[id] = held;
        out.it[k].ref = id;
      }
    } else {
      out.it[k] = WItem {};
    }
  }
  if (r == 0) out.slot = 0;
  recs[r] = out;
}

__global__ void k_wide_sizes(WRec* __restrict__ recs, uint32_t first, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  WRec& r = recs[first + i];
  uint32_t s = 1;
  for (uint32_t k = 0; k < r.count; k++)
    if (r.it[k].inner) s += recs[r.it[k].ref].wsize;
  r.wsize = s;
}

// DFS pre-order slots of the children and this record's Node128
__global__ void k_wide_emit(WRec* __restrict__ recs, uint32_t first, uint32_t count, Node128* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const WRec& r = recs[first + i];
  const float qnan = __uint_as_float(0x7FC00000u);
  float lo[3][4], hi[3][4], e[4];
  uint32_t ref[4];
  uint32_t next = r.slot + 1;
  for (uint32_t k = 0; k < 4; k++) {
    if (k < r.count) {
      for (int a = 0; a < 3; a++) {
        lo[a][k] = r.it[k].bot[a];
        hi[a][k] = r.it[k].top[a];
      }
      e[k] = r.it[k].e;
      if (r.it[k].inner) {
        WRec& c = recs[r.it[k].ref];
        c.slot = next;
        ref[k] = next;
        next += c.wsize;
      } else {
        ref[k] = r.it[k].ref;
      }
    } else {
      for (int a = 0; a < 3; a++) lo[a][k] = hi[a][k] = qnan;
      ref[k] = kEmptyRef;
      e[k] = 0.0f;
    }
  }
  Node128 nd;
  for (int a = 0; a < 3; a++) {
    nd.lo[a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
    nd.hi[a] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
  }
  for (int k = 0; k < 4; k++) {
    nd.ref[k] = ref[k];
    nd.e[k] = e[k];
  }
  out[r.slot] = nd;
}

struct Packed {
  Node64* nodes = nullptr; // hipMalloc'ed, owned by the caller on success
  Node128* wide = nullptr;
  uint32_t n_packed = 0, n_wide = 0, stack_need = 0, stack_need_wide = 0;
  float radius = 1.0f;
  float root_top[3] = {}, root_bot[3] = {};
};

#define PACKDEV_TRY(expr)                               \
  do {                                                  \
    const hipError_t e_ = (expr);                       \
    if (e_ != hipSuccess) return static_cast<int>(e_);  \
  } while (0)

// Returns 0 and fills *out, 1 when this tree is left to the host packer (a single leaf, a leaf of more
// than kMaxLeafTris triangles), or a HIP error code.
inline int pack(hipStream_t stream, const bvhb::Tree& tree, const float4* d_tris, Packed* out) {
  constexpr uint32_t kB = 256;
  auto grid = [](uint64_t items) { return static_cast<uint32_t>((items + kB - 1) / kB); };
  const uint32_t n_nodes = tree.n_nodes;
  if (n_nodes < 3) return 1;
  const uint32_t n_inner = (n_nodes - 1) / 2, n_leaves = (n_nodes + 1) / 2;
  const uint32_t n_levels = static_cast<uint32_t>(tree.level_start.size()) - 1;
  const BNode* nodes = tree.nodes.as<BNode>();
  const uint32_t* tri_offset = tree.tri_offset.as<uint32_t>();

  DeviceBuffer irank, lrank, leaves, leaf_e, node_e, misc, recs, rec_bnode, rec_above, nodes64, wide;
  PACKDEV_TRY(irank.reserve(static_cast<size_t>(n_nodes) * 4));
  PACKDEV_TRY(lrank.reserve(static_cast<size_t>(n_nodes) * 4));
  PACKDEV_TRY(leaves.reserve(static_cast<size_t>(n_leaves) * sizeof(LeafRec)));
  PACKDEV_TRY(leaf_e.reserve(static_cast<size_t>(n_leaves) * 4));
  PACKDEV_TRY(node_e.reserve(static_cast<size_t>(n_inner) * 4));
  PACKDEV_TRY(misc.reserve(64));
  PACKDEV_TRY(nodes64.reserve(static_cast<size_t>(n_inner) * sizeof(Node64)));
  uint32_t* d_oversize = misc.as<uint32_t>();
  uint32_t* d_n_recs = misc.as<uint32_t>() + 1;
  uint32_t* d_stack_need = misc.as<uint32_t>() + 2;
  PACKDEV_TRY(hipMemsetAsync(misc.p, 0, 64, stream));
  PACKDEV_TRY(hipMemsetAsync(irank.p, 0, 4, stream));
  PACKDEV_TRY(hipMemsetAsync(lrank.p, 0, 4, stream));
  for (uint32_t d = 0; d < n_levels; d++) {
    const uint32_t s = tree.level_start[d], m = tree.level_start[d + 1] - s;
    if (m) k_ranks<<<grid(m), kB, 0, stream>>>(nodes, s, m, irank.as<uint32_t>(), lrank.as<uint32_t>());
  }
  k_nodes64<<<grid(n_nodes), kB, 0, stream>>>(nodes, n_nodes, irank.as<uint32_t>(), lrank.as<uint32_t>(), tri_offset,
                                             nodes64.as<Node64>(), leaves.as<LeafRec>(), d_oversize);
  // root box and the oversize flag
  BNode root;
  uint32_t oversize = 0;
  PACKDEV_TRY(hipMemcpyAsync(&root, nodes, sizeof(BNode), hipMemcpyDeviceToHost, stream));
  PACKDEV_TRY(hipMemcpyAsync(&oversize, d_oversize, 4, hipMemcpyDeviceToHost, stream));
  PACKDEV_TRY(hipStreamSynchronize(stream));
  if (oversize) return 1;
  // scene radius: largest |coordinate| of the root box (get_scene)
  float radius = 0.0f;
  for (int a = 0; a < 3; a++) {
    radius = std::max(radius, std::fabs(root.top[a]));
    out->root_top[a] = root.top[a];
  }
  for (int a = 0; a < 3; a++) {
    radius = std::max(radius, std::fabs(root.bot[a]));
    out->root_bot[a] = root.bot[a];
  }
  if (!(radius > 0.0f) || !std::isfinite(radius)) radius = 1.0f;
  const float e_round = radius * kSlabRoundingSlack;

  k_leaf_slack<<<grid(n_leaves), 256, 0, stream>>>(d_tris, leaves.as<LeafRec>(), n_leaves, radius, leaf_e.as<float>());
  for (uint32_t d = n_levels; d-- > 0;) {
    const uint32_t s = tree.level_start[d], m = tree.level_start[d + 1] - s;
    if (m)
      k_slack_up<<<grid(m), kB, 0, stream>>>(nodes, s, m, irank.as<uint32_t>(), lrank.as<uint32_t>(), leaf_e.as<float>(),
                                            node_e.as<float>(), e_round, nodes64.as<Node64>());
  }

  // the wide tree, one wide level at a time
  PACKDEV_TRY(recs.reserve(static_cast<size_t>(n_inner) * sizeof(WRec)));
  PACKDEV_TRY(rec_bnode.reserve(static_cast<size_t>(n_inner) * 4));
  PACKDEV_TRY(rec_above.reserve(static_cast<size_t>(n_inner) * 4));
  PACKDEV_TRY(hipMemsetAsync(rec_bnode.p, 0, 4, stream)); // record 0: the root, nothing above it
  PACKDEV_TRY(hipMemsetAsync(rec_above.p, 0, 4, stream));
  const uint32_t one = 1;
  PACKDEV_TRY(hipMemcpyAsync(d_n_recs, &one, 4, hipMemcpyHostToDevice, stream));
  WideCtx C { nodes, irank.as<uint32_t>(), lrank.as<uint32_t>(), tri_offset, leaf_e.as<float>(), node_e.as<float>(),
              e_round };
  std::vector<uint32_t> wide_level { 0u, 1u };
  for (;;) {
    const uint32_t first = wide_level[wide_level.size() - 2], count = wide_level.back() - first;
    k_wide_expand<<<grid(count), kB, 0, stream>>>(C, recs.as<WRec>(), rec_bnode.as<uint32_t>(), rec_above.as<uint32_t>(),
                                                 first, count, d_n_recs, d_stack_need);
    uint32_t total = 0;
    PACKDEV_TRY(hipMemcpyAsync(&total, d_n_recs, 4, hipMemcpyDeviceToHost, stream));
    PACKDEV_TRY(hipStreamSynchronize(stream));
    if (total > n_inner) return static_cast<int>(hipErrorUnknown); // cannot happen: one record per inner node at most
    if (total == wide_level.back()) break;
    wide_level.push_back(total);
  }
  const uint32_t n_wide = wide_level.back();
  PACKDEV_TRY(wide.reserve(static_cast<size_t>(n_wide) * sizeof(Node128)));
  for (size_t l = wide_level.size() - 1; l-- > 0;) {
    const uint32_t first = wide_level[l], count = wide_level[l + 1] - first;
    if (count) k_wide_sizes<<<grid(count), kB, 0, stream>>>(recs.as<WRec>(), first, count);
  }
  for (size_t l = 0; l + 1 < wide_level.size(); l++) {
    const uint32_t first = wide_level[l], count = wide_level[l + 1] - first;
    if (count) k_wide_emit<<<grid(count), kB, 0, stream>>>(recs.as<WRec>(), first, count, wide.as<Node128>());
  }
  uint32_t stack_need_wide = 0;
  PACKDEV_TRY(hipMemcpyAsync(&stack_need_wide, d_stack_need, 4, hipMemcpyDeviceToHost, stream));
  PACKDEV_TRY(hipGetLastError());
  PACKDEV_TRY(hipStreamSynchronize(stream));

  out->n_packed = n_inner;
  out->n_wide = n_wide;
  out->stack_need = n_levels - 1; // deepest leaf: Packer::max_depth
  out->stack_need_wide = stack_need_wide;
  out->radius = radius;
  out->nodes = nodes64.as<Node64>();
  out->wide = wide.as<Node128>();
  nodes64.p = nullptr; // handed over
  nodes64.bytes = 0;
  wide.p = nullptr;
  wide.bytes = 0;
  return 0;
}

} // namespace packdev
} // namespace nova_hip
