// k_traverse_q: four lanes per ray on the four-wide tree (production kernel for incoherent rays).
// Replaces find_intersection / intersects_aabb / intersects_triangle
// (reference kernels/raytrace.hpp:38-102, kernels/intersection.hpp:14-52), same per-ray decisions as
// k_traverse_p (traverse.hpp), different mapping of the work onto the wave.
//
// Why (rocprofv3 counters of k_traverse_p, profiles/r01_pmc_summary_v4.txt): with one ray per lane a
// node step is 8 vector loads whose 64 lanes each touch a different 128-byte line — the L1 looks up
// ~26 lines per wave-instruction and the CU's texture-data path is busy 82-88 % of the launch at
// ~35 cycles per wave-instruction.  The cost unit is the (line, wave-instruction) pair, so the
// layout is turned round: a QUAD of four adjacent lanes owns one ray, lane k owns child k of the
// node (or triangle k of the leaf):
//   * node step = 2 vector loads; the four lanes of a quad read one contiguous 64-byte half of the
//     node each time (16 lines per wave-instruction instead of 26-64, a quarter of the instructions);
//   * one slab test per lane instead of four in a row; the children are ordered with three DPP
//     compares inside the quad (no LDS, no sort network), pushed with ONE LDS store for the wave;
//   * leaf step = 3 vector loads for four triangles (TriQuad: row r of four consecutive triangles is
//     64 contiguous bytes), four Woop tests in parallel, distance merged with two DPP mins;
//   * 16 rays per wave: a quarter of the rays in flight for the same occupancy, so launches of a few
//     hundred thousand rays are no longer all tail, and a long ray's leaves take a quarter of the steps.
// Everything that decides a result is unchanged: the boxes and Woop matrices are the reference's
// bits, the slab/Woop arithmetic is traverse.hpp's, culling uses the same slack, and the closest
// hit is min (t, triangle index) — triangle slots are numbered in the order of the original indices,
// so "lowest slot" is "lowest index".  Only the visiting order differs, and no result depends on it.
#pragma once

#include "traverse.hpp"

namespace nova_hip {

// Four-wide node, child-major: lane k of a quad loads lo_ref[k] and hi_e[k].
struct __attribute__((aligned(128))) NodeQ {
  float4 lo_ref[4]; // child k: box min.xyz, w = child reference (bits); kEmptyRef + NaN box when unused
  float4 hi_e[4];   // child k: box max.xyz, w = cull slack of the child's subtree
};

// Four consecutive triangle slots, row-major: row[r][k] = row r of slot k's Woop matrix.
// Leaves start on a TriQuad boundary; padding slots hold zero matrices (t = -0/0 = NaN: never a hit)
// and are also masked by the leaf's count.
struct __attribute__((aligned(64))) TriQuad {
  float4 row[3][4];
};

struct QuadArgs {
  const NodeQ* wide;        // four-wide tree (finite 1/d and -o/d)
  const NodeQ* pair;        // the reference's own box-pair tree in the same format (children 2, 3 unused)
  const TriQuad* tris;
  const uint32_t* slot_tri; // triangle slot -> original triangle index
};

// leaf reference in the quad trees: bit 31, bits 24-30 = triangles - 1, bits 0-23 = first TriQuad

constexpr int kQx1 = 0xB1; // quad_perm [1,0,3,2]: lane ^ 1
constexpr int kQx2 = 0x4E; // quad_perm [2,3,0,1]: lane ^ 2
constexpr int kQx3 = 0x1B; // quad_perm [3,2,1,0]: lane ^ 3

template <int CTRL>
NV_DEV uint32_t qperm(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
NV_DEV float qpermf(float v) { return __uint_as_float(qperm<CTRL>(__float_as_uint(v))); }
NV_DEV uint32_t quad_or(uint32_t v) {
  v |= qperm<kQx1>(v);
  v |= qperm<kQx2>(v);
  return v;
}
NV_DEV uint32_t quad_add(uint32_t v) {
  v += qperm<kQx1>(v);
  v += qperm<kQx2>(v);
  return v;
}
NV_DEV float quad_min(float v) {
  v = min_(v, qpermf<kQx1>(v));
  v = min_(v, qpermf<kQx2>(v));
  return v;
}

// upload-time: TriQuad array from the reference's 48-byte matrices (slot_tri[s] = original index or ~0)
__global__ __launch_bounds__(256) void k_pack_tri_quads(const float4* tris, const uint32_t* slot_tri,
                                                        uint32_t n_slots, float4* out) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const uint32_t t = slot_tri[s];
  float4 m0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), m1 = m0, m2 = m0;
  if (t != 0xFFFFFFFFu) {
    const float4* m = tris + 3 * static_cast<size_t>(t);
    m0 = m[0]; m1 = m[1]; m2 = m[2];
  }
  float4* q = out + 12 * static_cast<size_t>(s >> 2) + (s & 3u);
  q[0] = m0; q[4] = m1; q[8] = m2;
}

template <int SLDS, bool COUNT, int REFILL_MIN, int MIN_WAVES>
__global__ __launch_bounds__(256, MIN_WAVES) void k_traverse_q(TraverseArgs A, QuadArgs G) {
  __shared__ uint32_t lds_stack[4][SLDS][16];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t k = lane & 3u;  // the child / triangle slot this lane owns
  const uint32_t quad = lane >> 2;
  uint32_t (*stk)[16] = lds_stack[threadIdx.x >> 6];
  // stack entries beyond the LDS part: [entry][quad of the grid]
  const uint32_t spill_stride = (gridDim.x * blockDim.x) >> 2;
  uint32_t* spill = A.spill + ((blockIdx.x * blockDim.x + threadIdx.x) >> 2);
  const uint32_t n = A.n_items_ptr ? *A.n_items_ptr : A.n_items_imm;
  const uint32_t total = n * A.kinds; // the host guarantees n * kinds < 2^31
  uint32_t n_nodes = 0, n_tris = 0;
  const float4 rt = { A.root_top[0], A.root_top[1], A.root_top[2], 0.0f };
  const float4 rb = { A.root_bot[0], A.root_bot[1], A.root_bot[2], 0.0f };

  RayQueue Q;
  Q.init(total, min(A.rays_per_wave, max(1u, total / (gridDim.x * 4u))));

  // all of this is uniform within a quad
  RayState r = {};
  float slack_scale = 0.0f, far_limit = 0.0f;
  bool anyhit = false, pair = false, fin = false;
  uint32_t cur = kDone, sp = 0, slot = 0;
  // this lane's own best hit among the triangles it has tested (merged over the quad at the end)
  HitState h = { 0.0f, 0.0f, 0.0f, -1 };

  const uint64_t leaders = 0x1111111111111111ull;
  while (true) {
    const uint64_t idle = __ballot(cur == kDone) & leaders;
    if (!Q.dry && __popcll(idle) >= REFILL_MIN) {
      Q.fill_window(A.queue_head, lane);
      if (!Q.dry) {
        const uint32_t take = min(static_cast<uint32_t>(__popcll(idle)), Q.w_end - Q.w_next);
        const uint32_t my_rank = static_cast<uint32_t>(__popcll(idle & ((1ull << (lane & ~3u)) - 1ull)));
        if (cur == kDone && my_rank < take) {
          const uint32_t w = Q.w_next + my_rank;
          const uint32_t kind = w / n;
          const uint32_t j = w - kind * n;
          slot = kind * A.stride + j;
          const float4 o4 = A.ray_o[slot];
          const float4 d4 = A.ray_d[slot];
          h = { o4.w, 0.0f, 0.0f, -1 };
          bool live = !(o4.w < 0.0f);
          if (live) {
            ray_setup(o4, d4, r);
            anyhit = (__float_as_uint(d4.w) & 1u) != 0;
            // finite 1/d and -o/d: x * 0 is 0 for finite x, NaN otherwise
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
            far_limit = fma_(__builtin_fabsf(h.t), kCullRel, h.t);
            sp = 0;
            cur = 0;
          } else if (k == 0) {
            A.hits[slot] = make_float4(h.t, 0.0f, 0.0f, __int_as_float(-1));
          }
        }
        Q.w_next += take;
      }
      continue;
    }
    const uint64_t m_node = __ballot(cur < kDone);
    const uint64_t m_leaf = __ballot((cur & kLeafBit) != 0);
    if ((m_node | m_leaf) == 0) {
      if (Q.dry) break;
      continue; // cannot happen with REFILL_MIN <= 16; kept as a guard
    }
    if (__popcll(m_node) >= __popcll(m_leaf)) {
      if (cur < kDone) {
        // ---- node step: lane k tests child k (kernels/intersection.hpp:40-52 on that box) ----------
        const NodeQ* nd = (pair ? G.pair : G.wide) + cur;
        const float4 lo = nd->lo_ref[k];
        const float4 hi = nd->hi_e[k];
        if (COUNT && k == 0) n_nodes += 2;
        const float t1x = fma_(hi.x, r.inv.x, r.nio.x), t1y = fma_(hi.y, r.inv.y, r.nio.y), t1z = fma_(hi.z, r.inv.z, r.nio.z);
        const float t2x = fma_(lo.x, r.inv.x, r.nio.x), t2y = fma_(lo.y, r.inv.y, r.nio.y), t2z = fma_(lo.z, r.inv.z, r.nio.z);
        const float tn = max_(min_(t1x, t2x), max_(min_(t1y, t2y), min_(t1z, t2z)));
        const float tf = min_(max_(t1x, t2x), min_(max_(t1y, t2y), max_(t1z, t2z)));
        const float e = hi.w * slack_scale;
        const bool hit = (tn <= tf) && (tf + e >= 0.0f) && (tn - e <= far_limit);
        // order key: entry distance as a sortable integer with the child number in the two low bits
        // (unique within the quad); a miss sorts after every hit
        const uint32_t tb = __float_as_uint(tn);
        const uint32_t sortable = tb ^ (static_cast<uint32_t>(static_cast<int32_t>(tb) >> 31) | 0x80000000u);
        const uint32_t key = hit ? ((sortable & ~3u) | k) : 0xFFFFFFFFu;
        const uint32_t k1 = qperm<kQx1>(key), k2 = qperm<kQx2>(key), k3 = qperm<kQx3>(key);
        const uint32_t rank = (k1 < key ? 1u : 0u) + (k2 < key ? 1u : 0u) + (k3 < key ? 1u : 0u);
        const uint32_t n_hit = quad_add(hit ? 1u : 0u);
        const uint32_t ref = __float_as_uint(lo.w);
        // nearest child continues, the others go on the stack with the farthest at the bottom
        if (hit && rank > 0) {
          const uint32_t idx = sp + (n_hit - 1u - rank);
          if (idx < SLDS) stk[idx][quad] = ref;
          else spill[static_cast<size_t>(idx - SLDS) * spill_stride] = ref;
        }
        const uint32_t first = quad_or((hit && rank == 0) ? ref : 0u);
        if (n_hit > 0) {
          cur = first;
          sp += n_hit - 1u;
        } else if (sp != 0) {
          --sp;
          cur = sp < SLDS ? stk[sp][quad] : spill[static_cast<size_t>(sp - SLDS) * spill_stride];
        } else {
          fin = true;
        }
      }
    } else {
      if (cur & kLeafBit) {
        // ---- leaf step: lane k tests triangle k of the leaf's current TriQuad ---------------------
        const uint32_t g = cur & 0x00FFFFFFu;
        const uint32_t count = ((cur >> 24) & 0x7Fu) + 1u;
        const TriQuad* tq = G.tris + g;
        const float4 m0 = tq->row[0][k], m1 = tq->row[1][k], m2 = tq->row[2][k];
        bool acc = false;
        if (k < count) {
          if (COUNT) n_tris++;
          acc = woop(r, h, static_cast<int>(4u * g + k), m0, m1, m2);
        }
        const uint32_t any_acc = quad_or(acc ? 1u : 0u);
        const float best = quad_min(h.t);
        far_limit = fma_(best, kCullRel, best);
        const bool finished = anyhit && any_acc != 0;
        if (!finished && count > 4u) {
          cur = kLeafBit | ((count - 5u) << 24) | (g + 1u);
        } else if (!finished && sp != 0) {
          --sp;
          cur = sp < SLDS ? stk[sp][quad] : spill[static_cast<size_t>(sp - SLDS) * spill_stride];
        } else {
          fin = true;
        }
      }
    }
    if (__ballot(fin) != 0) {
      // merge the quad's four records by (t, slot): the winning lane writes the hit
      const uint32_t khi = h.id >= 0 ? __float_as_uint(h.t + 0.0f) : 0xFFFFFFFFu; // t + 0 turns -0 into +0
      const uint32_t klo = static_cast<uint32_t>(h.id);
      uint32_t bhi = khi, blo = klo;
      {
        const uint32_t ohi = qperm<kQx1>(bhi), olo = qperm<kQx1>(blo);
        const bool less = ohi < bhi || (ohi == bhi && olo < blo);
        bhi = less ? ohi : bhi;
        blo = less ? olo : blo;
      }
      {
        const uint32_t ohi = qperm<kQx2>(bhi), olo = qperm<kQx2>(blo);
        const bool less = ohi < bhi || (ohi == bhi && olo < blo);
        bhi = less ? ohi : bhi;
        blo = less ? olo : blo;
      }
      if (fin) {
        const bool none = blo == 0xFFFFFFFFu;
        const bool writer = none ? (k == 0) : (klo == blo);
        if (writer) {
          int id = -1;
          if (!none) id = static_cast<int>(G.slot_tri[blo]);
          A.hits[slot] = make_float4(h.t, h.u, h.v, __int_as_float(id));
        }
        cur = kDone;
        fin = false;
      }
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

}
