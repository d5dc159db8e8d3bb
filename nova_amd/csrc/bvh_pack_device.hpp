// The traversal layout straight from the device BVH build: what bvh_pack.hpp (Packer, the slack sweep,
// WideBuilder) makes on the host from a FlatBVHNode[] copy, made on the device from the builder's own
// tree (bvh_build.hpp: Tree) — the same Node64 / Node128 arrays, word for word — for scenes built by
// nova_hip_build_bvh_device, and from a caller's FlatBVHNode[] in device memory (pack_flat: the checks of
// Packer::ref_of node by node, depths and level lists, then the same kernels).  Arrays that are malformed,
// not nested, a single leaf, deeper than 64 or hold a leaf of more than 128 triangles go to the host packer.
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
// (the nodes of a level are ids[ls .. ls + m), or ls .. ls + m when the tree is stored level by level)
__global__ void k_ranks(const BNode* __restrict__ nodes, const uint32_t* __restrict__ ids, uint32_t ls, uint32_t m,
                        uint32_t* __restrict__ irank, uint32_t* __restrict__ lrank) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t me = ids ? ids[ls + i] : ls + i;
  const BNode& nd = nodes[me];
  if (!nd.left) return;
  const uint32_t me_i = irank[me], me_l = lrank[me];
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
__global__ void k_slack_up(const BNode* __restrict__ nodes, const uint32_t* __restrict__ ids, uint32_t ls, uint32_t m,
                           const uint32_t* __restrict__ irank, const uint32_t* __restrict__ lrank,
                           const float* __restrict__ leaf_e, float* __restrict__ node_e, float e_round,
                           Node64* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t me = ids ? ids[ls + i] : ls + i;
  const BNode& nd = nodes[me];
  if (!nd.left) return;
  float e = 0.0f;
  const uint32_t child[2] = { nd.left, nd.right };
  for (int k = 0; k < 2; k++) {
    const float x = nodes[child[k]].left ? node_e[irank[child[k]]] : leaf_e[lrank[child[k]]];
    e = (e < x) ? x : e;
  }
  node_e[irank[me]] = e;
  out[irank[me]].c.w = e + e_round;
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
struct TreeView {
  const BNode* nodes;       // left = the child visited first; size = nodes of the subtree; leaf: end - begin triangles
  const uint32_t* tri_offset; // first triangle of each leaf
  const uint32_t* ids;      // level lists, or null when level d is the id range [level_start[d], level_start[d + 1])
  const std::vector<uint32_t>* level_start; // stages: every node in a later stage than its parent
  uint32_t n_nodes;
  uint32_t depth;           // depth of the deepest node (root = 0)
};

inline int pack(hipStream_t stream, const TreeView& tree, const float4* d_tris, Packed* out) {
  constexpr uint32_t kB = 256;
  auto grid = [](uint64_t items) { return static_cast<uint32_t>((items + kB - 1) / kB); };
  const uint32_t n_nodes = tree.n_nodes;
  if (n_nodes < 3) return 1;
  const uint32_t n_inner = (n_nodes - 1) / 2, n_leaves = (n_nodes + 1) / 2;
  const std::vector<uint32_t>& level_start = *tree.level_start;
  const uint32_t n_levels = static_cast<uint32_t>(level_start.size()) - 1;
  if (tree.depth > 64) return 1; // the host path reports it (the reference's own STACK_SIZE)
  const BNode* nodes = tree.nodes;
  const uint32_t* tri_offset = tree.tri_offset;
  const uint32_t* ids = tree.ids;

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
  PACKDEV_TRY(hipMemsetAsync(irank.p, 0, 4, stream)); // the root is node 0 in both kinds of tree
  PACKDEV_TRY(hipMemsetAsync(lrank.p, 0, 4, stream));
  for (uint32_t d = 0; d < n_levels; d++) {
    const uint32_t s = level_start[d], m = level_start[d + 1] - s;
    if (m) k_ranks<<<grid(m), kB, 0, stream>>>(nodes, ids, s, m, irank.as<uint32_t>(), lrank.as<uint32_t>());
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
    const uint32_t s = level_start[d], m = level_start[d + 1] - s;
    if (m)
      k_slack_up<<<grid(m), kB, 0, stream>>>(nodes, ids, s, m, irank.as<uint32_t>(), lrank.as<uint32_t>(), leaf_e.as<float>(),
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
  out->stack_need = tree.depth; // deepest leaf: Packer::max_depth
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

inline int pack(hipStream_t stream, const bvhb::Tree& tree, const float4* d_tris, Packed* out) {
  const TreeView v { tree.nodes.as<BNode>(), tree.tri_offset.as<uint32_t>(), nullptr, &tree.level_start, tree.n_nodes, tree.depth };
  return pack(stream, v, d_tris, out);
}

// ---- a FlatBVHNode[] from the caller (src/bvh/bvh.cpp:211-249: DFS pre-order, left child adjacent) -------
// The checks of Packer::ref_of, node by node.  In pre-order the subtree of node i is [i, end(i)) with
// end(leaf) = i + 1 and end(inner) = end(right child); the array is a tree exactly when every inner node
// i < end(0) has its left subtree end where its right child begins: end(i + 1) == right(i).
constexpr uint32_t kFlagBad = 1, kFlagNotNested = 2, kFlagOversize = 4, kFlagDeep = 8;
constexpr uint32_t kMaxFlatDepth = 66;

struct FlatInfo {
  uint32_t end;   // 0: not a valid node
  uint32_t right; // 0: leaf
  uint32_t first, count;
};

__device__ __forceinline__ bool flat_classify(const nova_bvh_node* __restrict__ in, uint32_t n_alloc, uint32_t n_tris,
                                              uint32_t i, uint32_t* right, uint32_t* first, uint32_t* count) {
  const nova_bvh_node nd = in[i];
  if (!(nd.bottom_num_right.w >= 0.0f)) { // leaf (Packer::is_leaf)
    const float foff = nd.top_offset_left.w, fcnt = -nd.bottom_num_right.w;
    if (!(foff >= 0.0f) || !(fcnt >= 1.0f) || foff >= 16777216.0f || fcnt > 16777216.0f) return false;
    const uint32_t off = static_cast<uint32_t>(foff), cnt = static_cast<uint32_t>(fcnt);
    if (static_cast<uint64_t>(off) + cnt > n_tris) return false;
    *right = 0;
    *first = off;
    *count = cnt;
    return true;
  }
  const float fr = nd.bottom_num_right.w;
  if (!(fr < 16777216.0f)) return false;
  const uint32_t r = static_cast<uint32_t>(fr);
  if (static_cast<float>(r) != fr || r <= i + 1 || r >= n_alloc || i + 1 >= n_alloc) return false;
  *right = r;
  *first = *count = 0;
  return true;
}

__global__ void k_flat_end(const nova_bvh_node* __restrict__ in, uint32_t n_alloc, uint32_t n_tris,
                           FlatInfo* __restrict__ info) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_alloc) return;
  FlatInfo f;
  f.end = 0;
  if (flat_classify(in, n_alloc, n_tris, i, &f.right, &f.first, &f.count)) {
    // down the right spine to a leaf
    uint32_t j = i, r = f.right, a, b;
    bool ok = true;
    for (uint32_t step = 0; r != 0; step++) {
      if (step >= kMaxFlatDepth) { ok = false; break; }
      j = r;
      if (!flat_classify(in, n_alloc, n_tris, j, &r, &a, &b)) { ok = false; break; }
    }
    if (ok) f.end = j + 1;
  }
  info[i] = f;
}

__device__ __forceinline__ bool box_inside(const nova_bvh_node& c, const nova_bvh_node& p) {
  return c.top_offset_left.x <= p.top_offset_left.x && c.top_offset_left.y <= p.top_offset_left.y &&
         c.top_offset_left.z <= p.top_offset_left.z && c.bottom_num_right.x >= p.bottom_num_right.x &&
         c.bottom_num_right.y >= p.bottom_num_right.y && c.bottom_num_right.z >= p.bottom_num_right.z;
}

// structure checks, parents, build-node records
__global__ void k_flat_check(const nova_bvh_node* __restrict__ in, const FlatInfo* __restrict__ info, uint32_t n_nodes,
                             uint32_t* __restrict__ parent, BNode* __restrict__ nodes, uint32_t* __restrict__ tri_offset,
                             uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const FlatInfo f = info[i];
  const nova_bvh_node nd = in[i];
  BNode b;
  for (int a = 0; a < 3; a++) {
    b.top[a] = a == 0 ? nd.top_offset_left.x : (a == 1 ? nd.top_offset_left.y : nd.top_offset_left.z);
    b.bot[a] = a == 0 ? nd.bottom_num_right.x : (a == 1 ? nd.bottom_num_right.y : nd.bottom_num_right.z);
  }
  b.begin = f.first;
  b.end = f.first + f.count;
  b.left = b.right = 0;
  b.slot = 0;
  b.size = f.end - i;
  uint32_t flag = 0;
  // bottom <= top on every axis (Packer::ordered)
  if (!(b.bot[0] <= b.top[0]) || !(b.bot[1] <= b.top[1]) || !(b.bot[2] <= b.top[2])) flag |= kFlagNotNested;
  if (f.end == 0 || f.end > n_nodes) {
    flag |= kFlagBad;
  } else if (f.right) {
    if (f.right >= n_nodes || info[i + 1].end != f.right) {
      flag |= kFlagBad;
    } else {
      b.left = i + 1;
      b.right = f.right;
      parent[i + 1] = i;
      parent[f.right] = i;
      if (!box_inside(in[i + 1], nd) || !box_inside(in[f.right], nd)) flag |= kFlagNotNested;
    }
  } else if (f.count > kMaxLeafTris) {
    flag |= kFlagOversize;
  }
  if (flag) atomicOr(flags, flag);
  nodes[i] = b;
  tri_offset[i] = f.first;
}

__global__ void k_flat_depth(const uint32_t* __restrict__ parent, uint32_t n_nodes, uint32_t* __restrict__ depth,
                             uint32_t* __restrict__ level_count, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  uint32_t d = 0, j = i;
  while (j != 0 && d < kMaxFlatDepth) {
    j = parent[j];
    d++;
  }
  if (j != 0) {
    atomicOr(flags, kFlagDeep);
    d = kMaxFlatDepth;
  }
  depth[i] = d;
  atomicAdd(&level_count[d], 1u);
}

__global__ void k_flat_levels(const uint32_t* __restrict__ depth, uint32_t n_nodes, const uint32_t* __restrict__ level_first,
                              uint32_t* __restrict__ cursor, uint32_t* __restrict__ ids) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const uint32_t d = depth[i];
  ids[level_first[d] + atomicAdd(&cursor[d], 1u)] = i; // the order inside a level does not matter
}

// Returns 0 and fills *out, 1 when the array is left to the host packer (malformed, not nested, a single
// leaf, an oversized leaf, too deep: the host path handles or reports each), or a HIP error code.
inline int pack_flat(hipStream_t stream, const nova_bvh_node* d_bvh, uint32_t n_alloc, const float4* d_tris,
                     uint32_t n_tris, Packed* out) {
  constexpr uint32_t kB = 256;
  auto grid = [](uint64_t items) { return static_cast<uint32_t>((items + kB - 1) / kB); };
  if (n_alloc < 3) return 1;
  DeviceBuffer info, parent, nodes, tri_offset, depth, ids, misc;
  PACKDEV_TRY(info.reserve(static_cast<size_t>(n_alloc) * sizeof(FlatInfo)));
  k_flat_end<<<grid(n_alloc), kB, 0, stream>>>(d_bvh, n_alloc, n_tris, info.as<FlatInfo>());
  FlatInfo root;
  PACKDEV_TRY(hipMemcpyAsync(&root, info.p, sizeof(FlatInfo), hipMemcpyDeviceToHost, stream));
  PACKDEV_TRY(hipStreamSynchronize(stream));
  const uint32_t n_nodes = root.end;
  if (n_nodes < 3 || n_nodes > n_alloc || root.right == 0) return 1;
  constexpr uint32_t kWords = 2 * (kMaxFlatDepth + 1) + 2; // flags, level counts, cursors
  PACKDEV_TRY(parent.reserve(static_cast<size_t>(n_nodes) * 4));
  PACKDEV_TRY(nodes.reserve(static_cast<size_t>(n_nodes) * sizeof(BNode)));
  PACKDEV_TRY(tri_offset.reserve(static_cast<size_t>(n_nodes) * 4));
  PACKDEV_TRY(depth.reserve(static_cast<size_t>(n_nodes) * 4));
  PACKDEV_TRY(ids.reserve(static_cast<size_t>(n_nodes) * 4));
  PACKDEV_TRY(misc.reserve(2 * kWords * 4));
  uint32_t* d_flags = misc.as<uint32_t>();
  uint32_t* d_level_count = d_flags + 1;
  uint32_t* d_cursor = d_level_count + kMaxFlatDepth + 1;
  uint32_t* d_level_first = misc.as<uint32_t>() + kWords;
  PACKDEV_TRY(hipMemsetAsync(misc.p, 0, 2 * kWords * 4, stream));
  PACKDEV_TRY(hipMemsetAsync(parent.p, 0, static_cast<size_t>(n_nodes) * 4, stream));
  k_flat_check<<<grid(n_nodes), kB, 0, stream>>>(d_bvh, info.as<FlatInfo>(), n_nodes, parent.as<uint32_t>(),
                                                nodes.as<BNode>(), tri_offset.as<uint32_t>(), d_flags);
  uint32_t flags = 0;
  PACKDEV_TRY(hipMemcpyAsync(&flags, d_flags, 4, hipMemcpyDeviceToHost, stream));
  PACKDEV_TRY(hipStreamSynchronize(stream));
  if (flags) return 1; // parents are only trustworthy in a well-formed tree
  k_flat_depth<<<grid(n_nodes), kB, 0, stream>>>(parent.as<uint32_t>(), n_nodes, depth.as<uint32_t>(), d_level_count,
                                                d_flags);
  uint32_t host_words[kMaxFlatDepth + 2];
  PACKDEV_TRY(hipMemcpyAsync(host_words, d_flags, sizeof(host_words), hipMemcpyDeviceToHost, stream));
  PACKDEV_TRY(hipStreamSynchronize(stream));
  if (host_words[0]) return 1;
  std::vector<uint32_t> level_start { 0u };
  uint32_t level_first[kMaxFlatDepth + 1] = {};
  for (uint32_t d = 0; d <= kMaxFlatDepth; d++) {
    const uint32_t c = host_words[1 + d];
    if (c == 0) break;
    level_first[d] = level_start.back();
    level_start.push_back(level_start.back() + c);
  }
  if (level_start.back() != n_nodes) return 1;
  PACKDEV_TRY(hipMemcpyAsync(d_level_first, level_first, sizeof(level_first), hipMemcpyHostToDevice, stream));
  k_flat_levels<<<grid(n_nodes), kB, 0, stream>>>(depth.as<uint32_t>(), n_nodes, d_level_first, d_cursor,
                                                 ids.as<uint32_t>());
  const TreeView v { nodes.as<BNode>(), tri_offset.as<uint32_t>(), ids.as<uint32_t>(), &level_start, n_nodes,
                     static_cast<uint32_t>(level_start.size()) - 2 };
  return pack(stream, v, d_tris, out);
}

} // namespace packdev
} // namespace nova_hip
