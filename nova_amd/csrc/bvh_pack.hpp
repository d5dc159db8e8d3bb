// Host side of the BVH repack (no device calls): the reference's FlatBVHNode[] -> box-pair nodes
// (Node64) -> four-wide nodes (Node128).  Used by get_scene() in nova_hip.hip, which adds the
// per-leaf cull slack computed on the device (k_leaf_slack) between the two steps.
#pragma once

#include <algorithm>
#include <cstdint>
#include <limits>
#include <vector>

#include "nova_hip.h"
#include "traverse.hpp"

namespace nova_hip {

// ---- BVH repack: FlatBVHNode[] (DFS pre-order, left child adjacent; src/bvh/bvh.cpp:211-249)
// -> dense array of 64-byte box-pair nodes, plus the per-node cull slack (traverse.hpp).
struct Packer {
  const std::vector<nova_bvh_node>& in;
  uint32_t n_tris;
  std::vector<Node64> out;
  std::vector<LeafRec> leaves;
  // child k of packed node n: leaf index (>= 0) or -1 when it is packed node child_node[n][k]
  std::vector<int32_t> child_leaf[2];
  std::vector<uint32_t> child_node[2];
  uint32_t max_depth = 0;
  bool bad = false;
  bool nested = true; // every child box lies inside its parent's box, and every box has bottom <= top

  static bool inside(const nova_bvh_node& c, const nova_bvh_node& p) {
    return c.top_offset_left.x <= p.top_offset_left.x && c.top_offset_left.y <= p.top_offset_left.y &&
           c.top_offset_left.z <= p.top_offset_left.z && c.bottom_num_right.x >= p.bottom_num_right.x &&
           c.bottom_num_right.y >= p.bottom_num_right.y && c.bottom_num_right.z >= p.bottom_num_right.z;
  }

  // bottom <= top on every axis (false for NaN): the four-wide node step reads "the plane the ray meets first"
  // by the sign of 1/d instead of taking min / max of both, which is the same value only for such a box
  static bool ordered(const nova_bvh_node& n) {
    return n.bottom_num_right.x <= n.top_offset_left.x && n.bottom_num_right.y <= n.top_offset_left.y &&
           n.bottom_num_right.z <= n.top_offset_left.z;
  }

  static bool is_leaf(const nova_bvh_node& n) { return !(n.bottom_num_right.w >= 0.0f); }

  uint32_t new_node() {
    out.emplace_back();
    for (int k = 0; k < 2; k++) {
      child_leaf[k].push_back(-1);
      child_node[k].push_back(0);
    }
    return static_cast<uint32_t>(out.size() - 1);
  }

  void set_child(uint32_t slot, int k, const nova_float4& top, const nova_float4& bot, uint32_t ref, int32_t leaf) {
    Node64& nd = out[slot];
    if (k == 0) {
      nd.a = make_float4(top.x, top.y, top.z, __builtin_bit_cast(float, ref));
      nd.b.x = bot.x; nd.b.y = bot.y; nd.b.z = bot.z;
    } else {
      nd.c.x = top.x; nd.c.y = top.y; nd.c.z = top.z;
      nd.d = make_float4(bot.x, bot.y, bot.z, 0.0f);
      nd.b.w = __builtin_bit_cast(float, ref);
    }
    child_leaf[k][slot] = leaf;
    child_node[k][slot] = ref;
  }
  void set_empty_child(uint32_t slot, int k) {
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    nova_float4 v = { qnan, qnan, qnan, 0.0f };
    set_child(slot, k, v, v, kLeafBit, -2); // never entered: a NaN box fails `tmin <= tmax`
  }

  int32_t add_leaf(const nova_bvh_node& n, uint32_t off, uint32_t cnt) {
    LeafRec r;
    r.first = off;
    r.count = cnt;
    r.top[0] = n.top_offset_left.x; r.top[1] = n.top_offset_left.y; r.top[2] = n.top_offset_left.z;
    r.bot[0] = n.bottom_num_right.x; r.bot[1] = n.bottom_num_right.y; r.bot[2] = n.bottom_num_right.z;
    leaves.push_back(r);
    return static_cast<int32_t>(leaves.size() - 1);
  }

  // reference of the subtree rooted at original node i (depth = number of packed levels above);
  // *leaf receives the leaf index when the subtree is a single leaf reference
  uint32_t ref_of(uint32_t i, uint32_t depth, int32_t* leaf) {
    *leaf = -1;
    if (bad) return kLeafBit;
    const nova_bvh_node& n = in[i];
    if (!ordered(n)) nested = false;
    if (is_leaf(n)) {
      const float foff = n.top_offset_left.w, fcnt = -n.bottom_num_right.w;
      if (!(foff >= 0.0f) || !(fcnt >= 1.0f) || foff >= 16777216.0f || fcnt > 16777216.0f) { bad = true; return kLeafBit; }
      const uint32_t off = static_cast<uint32_t>(foff), cnt = static_cast<uint32_t>(fcnt);
      if (static_cast<uint64_t>(off) + cnt > n_tris) { bad = true; return kLeafBit; }
      max_depth = std::max(max_depth, depth);
      if (cnt <= kMaxLeafTris) {
        *leaf = add_leaf(n, off, cnt);
        return kLeafBit | ((cnt - 1u) << 24) | off;
      }
      // oversized leaf: chain of box-pair nodes, each holding one chunk and the rest
      return chain(n, off, cnt, depth);
    }
    const float fr = n.bottom_num_right.w;
    const uint32_t right = static_cast<uint32_t>(fr);
    if (static_cast<float>(right) != fr || right <= i + 1 || right >= in.size() || i + 1 >= in.size() ||
        out.size() >= in.size() + 16 || depth > 1100) {
      bad = true;
      return kLeafBit;
    }
    if (!inside(in[i + 1], n) || !inside(in[right], n)) nested = false;
    const uint32_t slot = new_node();
    int32_t l0, l1;
    const uint32_t r0 = ref_of(i + 1, depth + 1, &l0);
    const uint32_t r1 = ref_of(right, depth + 1, &l1);
    set_child(slot, 0, in[i + 1].top_offset_left, in[i + 1].bottom_num_right, r0, l0);
    set_child(slot, 1, in[right].top_offset_left, in[right].bottom_num_right, r1, l1);
    return slot;
  }

  uint32_t chain(const nova_bvh_node& leaf, uint32_t off, uint32_t cnt, uint32_t depth) {
    const uint32_t slot = new_node();
    const uint32_t head = kLeafBit | ((kMaxLeafTris - 1u) << 24) | off;
    const int32_t lh = add_leaf(leaf, off, kMaxLeafTris);
    const uint32_t rest = cnt - kMaxLeafTris;
    uint32_t r1;
    int32_t l1 = -1;
    if (rest <= kMaxLeafTris) {
      r1 = kLeafBit | ((rest - 1u) << 24) | (off + kMaxLeafTris);
      l1 = add_leaf(leaf, off + kMaxLeafTris, rest);
      max_depth = std::max(max_depth, depth + 1);
    } else {
      r1 = chain(leaf, off + kMaxLeafTris, rest, depth + 1);
    }
    set_child(slot, 0, leaf.top_offset_left, leaf.bottom_num_right, head, lh);
    set_child(slot, 1, leaf.top_offset_left, leaf.bottom_num_right, r1, l1);
    return slot;
  }
};

// ---- box-pair tree -> four-wide tree (traverse.hpp: Node128).  A node's child list starts as its
// two children; while it has a free slot, the inner child with the largest surface area is
// replaced by its own two children.  Wide nodes are numbered in DFS pre-order.
struct WideBuilder {
  const Packer& pk;
  const std::vector<float>& leaf_e;
  const std::vector<float>& node_e;
  float e_round;
  std::vector<Node128> out;
  uint32_t stack_need = 0;

  struct Item {
    float top[3], bot[3];
    uint32_t ref;
    float e;
    bool inner;
  };

  void children_of(uint32_t n, std::vector<Item>& items) const {
    const Node64& nd = pk.out[n];
    for (int k = 0; k < 2; k++) {
      const int32_t l = pk.child_leaf[k][n];
      if (l == -2) continue; // unused slot of the synthetic single-leaf root
      Item it;
      if (k == 0) {
        it.top[0] = nd.a.x; it.top[1] = nd.a.y; it.top[2] = nd.a.z;
        it.bot[0] = nd.b.x; it.bot[1] = nd.b.y; it.bot[2] = nd.b.z;
      } else {
        it.top[0] = nd.c.x; it.top[1] = nd.c.y; it.top[2] = nd.c.z;
        it.bot[0] = nd.d.x; it.bot[1] = nd.d.y; it.bot[2] = nd.d.z;
      }
      it.ref = pk.child_node[k][n];
      it.inner = (l == -1);
      it.e = (it.inner ? node_e[it.ref] : leaf_e[l]) + e_round;
      items.push_back(it);
    }
  }
  static float area(const Item& it) {
    const float dx = it.top[0] - it.bot[0], dy = it.top[1] - it.bot[1], dz = it.top[2] - it.bot[2];
    const float a = dx * dy + dy * dz + dz * dx;
    return a == a ? a : 0.0f;
  }

  // returns the wide index; `above` = stack entries already held when this node is entered
  uint32_t build(uint32_t n, uint32_t above) {
    std::vector<Item> items;
    items.reserve(4);
    children_of(n, items);
    while (items.size() < 4) {
      int best = -1;
      float best_area = -1.0f;
      for (size_t i = 0; i < items.size(); i++) {
        if (items[i].inner && area(items[i]) > best_area) { best = static_cast<int>(i); best_area = area(items[i]); }
      }
      if (best < 0) break;
      const uint32_t child = items[best].ref;
      items.erase(items.begin() + best);
      children_of(child, items);
    }
    const uint32_t slot = static_cast<uint32_t>(out.size());
    out.emplace_back();
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    float lo[3][4], hi[3][4], e[4];
    uint32_t ref[4];
    for (int k = 0; k < 4; k++) {
      for (int a = 0; a < 3; a++) { lo[a][k] = qnan; hi[a][k] = qnan; }
      ref[k] = kEmptyRef;
      e[k] = 0.0f;
    }
    const uint32_t held = above + static_cast<uint32_t>(items.size()) - 1u;
    stack_need = std::max(stack_need, held);
    for (size_t k = 0; k < items.size(); k++) {
      for (int a = 0; a < 3; a++) { lo[a][k] = items[k].bot[a]; hi[a][k] = items[k].top[a]; }
      e[k] = items[k].e;
      ref[k] = items[k].inner ? build(items[k].ref, held) : items[k].ref;
    }
    Node128& nd = out[slot];
    for (int a = 0; a < 3; a++) {
      nd.lo[a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
      nd.hi[a] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
    }
    for (int k = 0; k < 4; k++) { nd.ref[k] = ref[k]; nd.e[k] = e[k]; }
    return slot;
  }
};

}
