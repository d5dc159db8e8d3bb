// libnova_hip.so — host side of the C ABI declared in include/nova_hip.h: per-process device
// context, buffer registry, linear images, BVH repack + scene cache, and the launch schedules of
// the wavefront pipeline (generate -> traverse -> shade ... ) and the streaming image kernels.
// gfx950 only.  No PyTorch, no CUDA shims, no CPU fallback: without a HIP device every entry
// point fails with NOVA_HIP_E_NO_DEVICE / the hipError_t.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: librccl is opened on first use (nova_hip_tiles_comm_*)

#include <dlfcn.h>

#include "alloc_count.hpp" // first: wraps the allocating runtime calls of everything below

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "bvh_build.hpp"
#include "bvh_pack.hpp"
#include "bvh_pack_device.hpp"
#include "image_kernels.hpp"
#include "kernel_args_check.hpp"
#include "nova_hip.h"
#include "shade.hpp"
#include "traverse.hpp"

using namespace nova_hip;

static_assert(sizeof(nova_bvh_node) == 32, "FlatBVHNode layout");
static_assert(sizeof(nova_triangle_data) == 48, "TriangleData layout");
static_assert(sizeof(nova_triangle_meta) == 176, "TriangleMetaData layout");
static_assert(sizeof(nova_area_light) == 52, "AreaLightData layout");
static_assert(sizeof(nova_eye_coords) == 64, "EyeCoords layout");
static_assert(sizeof(nova_scene_params) == 104, "SceneParams layout");
static_assert(offsetof(nova_triangle_meta, texture_coord1) == 112, "TriangleMetaData layout");
static_assert(offsetof(nova_scene_params, path_tracing) == 84, "SceneParams layout");
static_assert(sizeof(Node64) == 64, "Node64 layout");
static_assert(sizeof(Node128) == 128, "Node128 layout");

constexpr size_t kHeadWords = static_cast<size_t>(kSubQueues) * kSubQueueStride; // u32 words of one launch's queue heads

// Scratch of nova_hip_trace_rays_noalloc, the one entry point that allocates nothing: the queue heads of
// one traversal launch, then the traversal-stack entries beyond the LDS part ([entry][thread of the grid],
// as HybridStack lays them out).  32 MiB + 8 KiB per GPU, held until the process ends.  On the host this
// NAME is only a handle for hipGetSymbolAddress (nova_hip_init keeps the device address in
// Ctx::scratch_ptr); host code never passes the name itself to a kernel or a copy.
constexpr size_t kNoAllocSpillWords = static_cast<size_t>(8) << 20;
static __device__ __align__(256) uint32_t scratch[kHeadWords + kNoAllocSpillWords];

namespace {

constexpr uint32_t kMaxIterations = 1024; // hard cap on path length in unbounded mode
constexpr uint32_t kTileSize = 64;
constexpr size_t kQueueWords = 4096; // path counters [0, kMaxIterations + 2), MIS-ray counters from kMisWords on; a multiple of the head stride
constexpr size_t kMisWords = 2048;

#define HIP_TRY(expr)                        \
  do {                                       \
    hipError_t e_ = (expr);                  \
    if (e_ != hipSuccess) return static_cast<int>(e_); \
  } while (0)

struct Alloc {
  size_t bytes;
  uint64_t version;
};

struct PackedScene {
  const void* bvh_ptr = nullptr;
  const void* tri_ptr = nullptr;
  uint64_t bvh_version = 0, tri_version = 0;
  Node64* nodes = nullptr;
  Node128* wide = nullptr; // null when the input boxes are not nested (traverse.hpp)
  uint32_t n_wide = 0;
  uint32_t stack_need_wide = 0;
  uint32_t n_packed = 0;
  uint32_t n_tris = 0;
  uint32_t stack_need = 0;
  float radius = 1.0f;
  float root_top[3], root_bot[3];
};

struct FrameBuffers {
  uint32_t capacity = 0; // paths
  float4* planes[2][6] = {};
  float4* ray_o[2] = {};
  float4* ray_d[2] = {};
  float4* hits = nullptr;
  unsigned long long* block_counters = nullptr; // [capacity / kShadeBlock][kBlockCounters]
  uint32_t n_blocks = 0;
};

// What one sample in flight owns.  The context's own fields of these names are the CURRENT lane;
// nova_hip_render_samples keeps further ones and moves on between samples (switch_lane), so that the
// latency-bound tail of sample k (its last, nearly empty traversal launches) runs beside the start
// of sample k+1 on another stream.
constexpr int kMaxLanes = 4;
constexpr int kDefaultLanes = 3; // 1080p, 16 samples: 3.94 / 3.40 / 3.27 / 3.42 ms per sample with 1 / 2 / 3 / 4 lanes
struct Lane {
  hipStream_t stream = nullptr;
  FrameBuffers fb;
  uint32_t* d_queues = nullptr;
  uint32_t* d_heads = nullptr;
  uint32_t* h_pinned = nullptr;
  uint32_t* d_spill = nullptr;
  size_t spill_words = 0;
  hipEvent_t chain_tail = nullptr;
};

// What a rank of a tile partition keeps besides its own tiles.  kernel_accumulate and kernel_post_process read
// every pixel through a LINEAR sampler at the pixel centre (kernels/kernels.cu:59-64, :103-110); in float the
// centre (x + 0.5) / W * W is not always x + 0.5, so at a few columns and rows a neighbouring texel gets a
// weight of ~1e-6 instead of 0.  Where such a neighbour lies in another rank's tile this rank renders and
// accumulates it as well ("extra" pixels: the closure of its own pixels under "is read with a non-zero
// weight"), so that its own pixels hold exactly the single-GPU frame's values.  Zero-weight taps are not
// followed (0 * finite = 0): what a NaN texel of another rank's tile would do to its neighbour through them
// (the reference's behaviour for, e.g., the normal feature of sky pixels) is not reproduced across tiles.
struct TilePlan {
  uint32_t w = 0, h = 0, rank = 0, ranks = 1;
  uint8_t* d_mask = nullptr;   // [h][w]: bit 0 = own tile, bit 1 = extra pixel
  uint32_t* d_extra = nullptr; // pixel ids (y * w + x) of the extra pixels, row-major order
  uint32_t n_extra = 0;
};

struct Ctx {
  bool ready = false;
  int device = -1;
  int num_cus = 0;
  hipStream_t stream = nullptr;
  std::unordered_map<const void*, Alloc> allocs;
  uint64_t next_version = 1;
  std::vector<PackedScene> scenes;
  FrameBuffers fb;
  uint32_t* d_queues = nullptr;          // [0..kMaxIterations+1] n_active per iteration
  uint32_t* d_heads = nullptr;           // per iteration: kSubQueues queue heads, one per 128-byte line
  unsigned long long* d_counters = nullptr; // 8 x u64
  uint32_t* h_pinned = nullptr;          // readback word
  uint32_t* d_spill = nullptr;           // traversal stack entries beyond the LDS part
  size_t spill_words = 0;
  TilePlan plan;
  // ray binning of nova_hip_trace_rays (traverse.hpp: k_ray_keys): keys, permutation, histogram / cursors
  uint32_t* d_ray_keys = nullptr;
  uint32_t* d_ray_perm = nullptr;
  uint32_t* d_ray_hist = nullptr;
  size_t ray_sort_capacity = 0;
  bvhb::Scanner ray_scan;
  int64_t opt_sort_rays = 0;
  // denoise hand-off: four float3 planes (colour, albedo, normal, output) on the device and pinned on the host
  float* d_denoise = nullptr;
  float* h_denoise = nullptr;
  size_t denoise_pixels = 0;
  // RCCL gather of tile stacks (nova_hip_tiles_allgather): the communicator, whether the library made it, and the
  // send / receive stacks
  ncclComm_t comm = nullptr;
  bool comm_owned = false;
  uint32_t comm_rank = 0, comm_ranks = 0;
  void* d_gather = nullptr;              // [send stack][ranks receive stacks]
  size_t gather_bytes = 0;
  // light pre-test of MIS rays (shade.hpp): the triangles of `lt_meta` (at that version) that carry a light index
  uint32_t* d_light_tris = nullptr;      // kLightTriWords words
  const void* lt_meta = nullptr;
  const void* lt_tri_ptr = nullptr;
  uint64_t lt_version = 0, lt_tri_version = 0;
  uint32_t lt_tris = 0;
  int64_t opt_mis_pretest = 1;
  // NOVA_HIP_OPT_ASYNC_LAUNCH: the four per-sample calls return once their kernels are enqueued; `unsettled` = some
  // of that work has not been waited for yet (settle())
  int64_t opt_async = 1;
  int64_t opt_call_lanes = 0; // NOVA_HIP_OPT_CALL_LANES: 0 = built-in (kDefaultLanes, env NOVA_HIP_CALL_LANES)
  bool unsettled = false;
  int samples_in_flight = 1; // lanes of the schedule the current launch belongs to (launch_traverse_wide)
  uint32_t* scratch_ptr = nullptr;       // device address of `scratch` on this device (hipGetSymbolAddress at init)
  bool no_alloc = false;                 // inside nova_hip_trace_rays_noalloc: no events, no buffer growth
  // options
  int64_t opt_max_depth = 0, opt_traversal = 0, opt_count_work = 0, opt_tile_rank = 0,
          opt_tile_ranks = 1, opt_kernel = 0, opt_tune = 0, opt_clamp_to_edge = 0;
  // timing
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_traverse, ev_camera, ev_shade;
  hipEvent_t chain_tail = nullptr; // last event recorded on the stream with nothing enqueued after it
  double traverse_ms = 0, camera_ms = 0, shade_ms = 0;
  uint64_t traverse_launches = 0, camera_launches = 0;
  uint64_t host_primary = 0; // primary rays = valid pixels of the tiles rendered
  uint64_t folded[kBlockCounters] = {}; // block counters of released frame buffers
  // further lanes of nova_hip_render_samples (made on first use) and what orders the lanes.
  // lanes[cur] is stale: the current lane lives in the fields above (switch_lane).
  Lane lanes[kMaxLanes];
  int cur = 0;
  int lanes_ready = 1;
  hipEvent_t acc_done[2] = { nullptr, nullptr }; // recorded after kernel_accumulate of even / odd samples
  float4* stage[kMaxLanes - 1][3] = {}; // temp_color1 / albedo1 / normal1 of the lanes that do not render the last sample
  size_t stage_bytes = 0;
  // per-call overlap (nova_hip_launch_raytrace with NOVA_HIP_OPT_ASYNC_LAUNCH): every lane renders into planes of its
  // own and copies them into the caller's images once the image work enqueued before it is done (lane_mark)
  float4* call_stage[kMaxLanes][3] = {};
  size_t call_stage_bytes = 0;
  hipEvent_t lane_mark[kMaxLanes] = {};
};

Ctx g;
std::mutex g_mutex;

// a sample of nova_hip_render_samples whose first iterations are enqueued on its lane (see raytrace_impl)
struct Pending {
  bool active = false;
  FrameArgs F;
  const nova_bvh_node* bvh = nullptr;
  const nova_triangle_data* triangles = nullptr;
  uint32_t P = 0, next_it = 0, last_it = 0;
  bool nee = false;
};
Pending g_pending[kMaxLanes];

// Image work that waits for a sample whose paths are not all known to be dead (unbounded paths under lanes,
// nova_hip_launch_raytrace).  With NOVA_HIP_OPT_MAX_DEPTH 0 a sample's first iterations are enqueued at once and the
// survivor count is looked at when its lane comes round again — by then the answer is there without waiting, as in
// nova_hip_render_samples.  Until then the calls that follow the launch in the caller's program (the copy of the
// lane's planes into the caller's images, kernel_accumulate, copy_image2D, kernel_post_process) are kept here, in
// program order, and run when the sample is finished: at the launch that reuses its lane, or when any other entry
// point settles.  Each entry enqueues on its sample's lane; an event orders it behind the image work before it.
struct Deferred {
  uint64_t sample;
  int lane;
  std::function<int()> run;
};
std::deque<Deferred> g_deferred;
uint64_t g_sample_seq = 0;      // the latest sample launched in this mode
uint64_t g_sample_finished = 0; // the latest one whose remaining iterations have been enqueued
int g_image_lane = -1;          // lane of the last image work enqueued in this mode (its lane_mark is recorded behind it)
int g_issue_lane = 0;           // lane of the latest sample launched in this mode
int (*g_flush_deferred)(uint64_t through_sample) = nullptr; // flush_deferred, once the mode has been used

int require_ctx() {
  if (!g.ready) {
    int rc = nova_hip_init(0);
    if (rc != 0) return rc;
  }
  return 0;
}

const Alloc* find_alloc(const void* p) {
  auto it = g.allocs.find(p);
  return it == g.allocs.end() ? nullptr : &it->second;
}

void free_scene(PackedScene& s) {
  (void) hipFree(s.nodes);
  (void) hipFree(s.wide);
  s.nodes = nullptr;
  s.wide = nullptr;
}

void drop_scenes_using(const void* p) {
  for (size_t i = 0; i < g.scenes.size();) {
    if (g.scenes[i].bvh_ptr == p || g.scenes[i].tri_ptr == p) {
      free_scene(g.scenes[i]);
      g.scenes.erase(g.scenes.begin() + i);
    } else {
      i++;
    }
  }
}

// new contents at p: the allocation that CONTAINS p gets a new version (a write into the middle of a
// registered BVH or triangle buffer must invalidate the cached traversal layout too)
void touch(const void* p) {
  auto it = g.allocs.find(p);
  if (it == g.allocs.end()) {
    const char* c = static_cast<const char*>(p);
    for (it = g.allocs.begin(); it != g.allocs.end(); ++it) {
      const char* base = static_cast<const char*>(it->first);
      if (c >= base && c < base + it->second.bytes) break;
    }
  }
  if (it != g.allocs.end()) it->second.version = g.next_version++;
}

hipEvent_t next_event() {
  static const bool off = std::getenv("NOVA_HIP_NO_EVENTS") != nullptr; // diagnostics: launches without timing events
  if (off || g.no_alloc) return nullptr; // hipEventCreate allocates
  // a call that enqueues thousands of kernels before it waits (nova_hip_render_samples with many
  // samples) stops timing them rather than growing the pool without bound
  if (g.ev_used >= 8192) return nullptr;
  if (g.ev_used == g.ev_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g.ev_pool.push_back(e);
  }
  return g.ev_pool[g.ev_used++];
}

// Timing spans share their boundary events: the event recorded after one kernel is the start of
// the next span (half the event records of a pair per kernel: 0.05 ms less per 1080p frame; a span
// then includes the few microseconds of launch gap before its kernel).
void span_begin(hipEvent_t* e0) {
  if (g.chain_tail) {
    *e0 = g.chain_tail;
    return;
  }
  *e0 = next_event();
  if (*e0) (void) hipEventRecord(*e0, g.stream);
}
hipEvent_t span_end() {
  hipEvent_t e1 = next_event();
  if (e1) (void) hipEventRecord(e1, g.stream);
  g.chain_tail = e1;
  return e1;
}

void collect_events() {
  g.chain_tail = nullptr;
  for (auto& p : g.ev_traverse) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) g.traverse_ms += ms;
  }
  for (auto& p : g.ev_camera) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
      g.traverse_ms += ms;
      g.camera_ms += ms;
    }
  }
  g.ev_camera.clear();
  for (auto& p : g.ev_shade) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) g.shade_ms += ms;
  }
  g.ev_traverse.clear();
  g.ev_shade.clear();
  g.ev_used = 0;
}

// Every entry point other than the four per-sample launches starts here: whatever those left enqueued on the
// library's stream (NOVA_HIP_OPT_ASYNC_LAUNCH) is waited for, so that reads, writes, frees and option changes see the
// state a blocking call would have left.
int settle() {
  if (!g.ready || !g.unsettled) return 0;
  g.unsettled = false;
  if (g_flush_deferred && !g_deferred.empty()) {
    const int rc = g_flush_deferred(~0ull);
    if (rc) {
      g_deferred.clear();
      (void) hipDeviceSynchronize();
      collect_events();
      return rc;
    }
  }
  HIP_TRY(hipStreamSynchronize(g.stream)); // the current lane holds the newest work, which waited for the others'
  for (int l = 0; l < g.lanes_ready; l++)
    if (l != g.cur && g.lanes[l].stream) HIP_TRY(hipStreamSynchronize(g.lanes[l].stream));
  collect_events();
  return 0;
}
// a per-sample launch is done enqueuing: wait (blocking mode) or leave it to settle()
int launched() {
  if (g.opt_async && g.ev_used < 8192) { // (the event pool of the timing spans grows until someone waits)
    g.unsettled = true;
    return 0;
  }
  g.unsettled = false;
  HIP_TRY(hipStreamSynchronize(g.stream));
  collect_events();
  return 0;
}

void switch_lane(int to) {
  if (to == g.cur) return;
  Lane& o = g.lanes[g.cur];
  o.stream = g.stream; o.fb = g.fb; o.d_queues = g.d_queues; o.d_heads = g.d_heads; o.h_pinned = g.h_pinned;
  o.d_spill = g.d_spill; o.spill_words = g.spill_words; o.chain_tail = g.chain_tail;
  const Lane& n = g.lanes[to];
  g.stream = n.stream; g.fb = n.fb; g.d_queues = n.d_queues; g.d_heads = n.d_heads; g.h_pinned = n.h_pinned;
  g.d_spill = n.d_spill; g.spill_words = n.spill_words; g.chain_tail = n.chain_tail;
  g.cur = to;
}

void free_frame_buffers(FrameBuffers& fb) {
  for (int s = 0; s < 2; s++) {
    for (int k = 0; k < 6; k++) { (void) hipFree(fb.planes[s][k]); fb.planes[s][k] = nullptr; }
    (void) hipFree(fb.ray_o[s]); fb.ray_o[s] = nullptr;
    (void) hipFree(fb.ray_d[s]); fb.ray_d[s] = nullptr;
  }
  (void) hipFree(fb.hits); fb.hits = nullptr;
  (void) hipFree(fb.block_counters); fb.block_counters = nullptr;
  fb.capacity = 0;
  fb.n_blocks = 0;
}

// per-block ray counters of a lane's frame buffers, added to bc[]
int add_block_counters(const FrameBuffers& fb, uint64_t bc[kBlockCounters]) {
  if (!fb.block_counters) return 0;
  std::vector<unsigned long long> hb(kBlockCounters * static_cast<size_t>(fb.n_blocks));
  HIP_TRY(hipMemcpy(hb.data(), fb.block_counters, hb.size() * 8, hipMemcpyDeviceToHost));
  for (size_t b = 0; b < fb.n_blocks; b++)
    for (int q = 0; q < kBlockCounters; q++) bc[q] += hb[kBlockCounters * b + q];
  return 0;
}

int pack_scene_host(const nova_bvh_node* bvh, const nova_triangle_data* tris, const Alloc* ab, const Alloc* at,
                    PackedScene& s);

bool host_pack_only() {
  static const bool on = std::getenv("NOVA_HIP_HOST_PACK") != nullptr; // diagnostics
  return on;
}

void adopt_packed(const packdev::Packed& pd, const void* bvh, const void* tris, const Alloc* ab, const Alloc* at,
                  PackedScene& s) {
  s.bvh_ptr = bvh;
  s.tri_ptr = tris;
  s.bvh_version = ab->version;
  s.tri_version = at->version;
  s.nodes = pd.nodes;
  s.wide = pd.wide;
  s.n_wide = pd.n_wide;
  s.n_packed = pd.n_packed;
  s.n_tris = static_cast<uint32_t>(at->bytes / sizeof(nova_triangle_data));
  s.stack_need = pd.stack_need;
  s.stack_need_wide = pd.stack_need_wide;
  s.radius = pd.radius;
  for (int a = 0; a < 3; a++) {
    s.root_top[a] = pd.root_top[a];
    s.root_bot[a] = pd.root_bot[a];
  }
}

// 0: packed on the device; 1: left to the host packer; else an error
int pack_scene_device(const nova_bvh_node* bvh, const nova_triangle_data* tris, const Alloc* ab, const Alloc* at,
                      PackedScene& s) {
  const size_t n_alloc = ab->bytes / sizeof(nova_bvh_node);
  const size_t n_tris = at->bytes / sizeof(nova_triangle_data);
  if (n_alloc == 0 || n_tris == 0 || n_tris > (1u << 24) || n_alloc > (1u << 26)) return 1;
  packdev::Packed pd;
  const int rc = packdev::pack_flat(g.stream, bvh, static_cast<uint32_t>(n_alloc), reinterpret_cast<const float4*>(tris),
                                    static_cast<uint32_t>(n_tris), &pd);
  if (rc) return rc < 0 || rc > 1 ? rc : 1;
  adopt_packed(pd, bvh, tris, ab, at, s);
  return 0;
}

int get_scene(const nova_bvh_node* bvh, const nova_triangle_data* tris, PackedScene** out) {
  const Alloc* ab = find_alloc(bvh);
  const Alloc* at = find_alloc(tris);
  if (!ab || !at) return NOVA_HIP_E_UNKNOWN_BUFFER;
  for (auto& s : g.scenes) {
    if (s.bvh_ptr == bvh && s.tri_ptr == tris && s.bvh_version == ab->version && s.tri_version == at->version) {
      *out = &s;
      return 0;
    }
  }
  (void) hipDeviceSynchronize(); // an earlier launch on any lane, not yet awaited, may still walk a layout that is dropped here
  drop_scenes_using(bvh); // stale entries for a rewritten buffer
  PackedScene s;
  // on the device where the array is a well-formed, nested tree (bvh_pack_device.hpp); else on the host
  int rc = 1;
  if (!host_pack_only()) rc = pack_scene_device(bvh, tris, ab, at, s);
  if (rc == 1) rc = pack_scene_host(bvh, tris, ab, at, s);
  if (rc) return rc;
  g.scenes.push_back(s);
  *out = &g.scenes.back();
  return 0;
}

// The traversal layout of a FlatBVHNode[] in device memory, made on the host (bvh_pack.hpp)
int pack_scene_host(const nova_bvh_node* bvh, const nova_triangle_data* tris, const Alloc* ab, const Alloc* at,
                    PackedScene& s) {
  const size_t n_nodes = ab->bytes / sizeof(nova_bvh_node);
  const size_t n_tris = at->bytes / sizeof(nova_triangle_data);
  if (n_nodes == 0 || n_tris == 0) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (n_tris > (1u << 24)) return NOVA_HIP_E_TOO_LARGE;
  std::vector<nova_bvh_node> host(n_nodes);
  HIP_TRY(hipMemcpy(host.data(), bvh, n_nodes * sizeof(nova_bvh_node), hipMemcpyDeviceToHost));

  Packer pk { host, static_cast<uint32_t>(n_tris) };
  pk.out.reserve(n_nodes / 2 + 2);
  const nova_bvh_node& root = host[0];
  int32_t root_leaf = -1;
  if (Packer::is_leaf(root)) {
    // a single-leaf tree: give it a parent whose second child is never entered
    const uint32_t slot = pk.new_node();
    const uint32_t r0 = pk.ref_of(0, 1, &root_leaf);
    pk.set_child(slot, 0, root.top_offset_left, root.bottom_num_right, r0, root_leaf);
    pk.set_empty_child(slot, 1);
  } else {
    pk.ref_of(0, 0, &root_leaf);
  }
  if (pk.bad || pk.out.empty()) return NOVA_HIP_E_BAD_SCENE;
  if (pk.max_depth > 64) return NOVA_HIP_E_TOO_LARGE; // the reference's own STACK_SIZE (kernels/constants.hpp:8)

  // scene radius: largest |coordinate| of the root box
  float radius = 0.0f;
  const float rc[6] = { root.top_offset_left.x, root.top_offset_left.y, root.top_offset_left.z,
                        root.bottom_num_right.x, root.bottom_num_right.y, root.bottom_num_right.z };
  for (float c : rc) radius = std::max(radius, std::fabs(c));
  if (!(radius > 0.0f) || !std::isfinite(radius)) radius = 1.0f;

  // per-leaf slack on the device (it reads every Woop matrix), then max over subtrees on the host
  const size_t n_leaves = pk.leaves.size();
  std::vector<float> leaf_e(n_leaves);
  {
    LeafRec* d_leaves = nullptr;
    float* d_e = nullptr;
    HIP_TRY(hipMalloc(&d_leaves, n_leaves * sizeof(LeafRec)));
    hipError_t e = hipMalloc(&d_e, n_leaves * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_leaves, pk.leaves.data(), n_leaves * sizeof(LeafRec), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      const unsigned blocks = static_cast<unsigned>((n_leaves + 255) / 256);
      hipLaunchKernelGGL(k_leaf_slack, dim3(blocks), dim3(256), 0, g.stream,
                         reinterpret_cast<const float4*>(tris), d_leaves, static_cast<uint32_t>(n_leaves), radius, d_e);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (e == hipSuccess) e = hipMemcpy(leaf_e.data(), d_e, n_leaves * sizeof(float), hipMemcpyDeviceToHost);
    (void) hipFree(d_leaves);
    (void) hipFree(d_e);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  // packed nodes are in DFS pre-order (children after parents): one reverse sweep
  std::vector<float> node_e(pk.out.size(), 0.0f);
  const float e_round = radius * kSlabRoundingSlack;
  for (size_t n = pk.out.size(); n-- > 0;) {
    float e = 0.0f;
    for (int k = 0; k < 2; k++) {
      const int32_t l = pk.child_leaf[k][n];
      if (l >= 0) e = std::max(e, leaf_e[l]);
      else if (l == -1) e = std::max(e, node_e[pk.child_node[k][n]]);
    }
    node_e[n] = e;
    pk.out[n].c.w = e + e_round;
  }

  s.bvh_ptr = bvh;
  s.tri_ptr = tris;
  s.bvh_version = ab->version;
  s.tri_version = at->version;
  s.n_packed = static_cast<uint32_t>(pk.out.size());
  s.n_tris = static_cast<uint32_t>(n_tris);
  s.stack_need = pk.max_depth;
  s.radius = radius;
  s.root_top[0] = root.top_offset_left.x; s.root_top[1] = root.top_offset_left.y; s.root_top[2] = root.top_offset_left.z;
  s.root_bot[0] = root.bottom_num_right.x; s.root_bot[1] = root.bottom_num_right.y; s.root_bot[2] = root.bottom_num_right.z;
  HIP_TRY(hipMalloc(&s.nodes, pk.out.size() * sizeof(Node64)));
  hipError_t e = hipMemcpy(s.nodes, pk.out.data(), pk.out.size() * sizeof(Node64), hipMemcpyHostToDevice);
  if (e == hipSuccess && pk.nested) {
    WideBuilder wb { pk, leaf_e, node_e, e_round };
    wb.out.reserve(pk.out.size() / 2 + 2);
    wb.build(0, 0);
    s.n_wide = static_cast<uint32_t>(wb.out.size());
    s.stack_need_wide = wb.stack_need;
    e = hipMalloc(&s.wide, wb.out.size() * sizeof(Node128));
    if (e == hipSuccess) e = hipMemcpy(s.wide, wb.out.data(), wb.out.size() * sizeof(Node128), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    free_scene(s);
    return static_cast<int>(e);
  }
  return 0;
}

template <int STACK>
int launch_traverse_t(const TraverseArgs& A, bool cull, bool count) {
  const int blocks_per_cu = STACK <= 32 ? 4 : 2;
  dim3 grid(static_cast<unsigned>(g.num_cus * blocks_per_cu)), block(256);
  if (cull) {
    if (count) hipLaunchKernelGGL((k_traverse<STACK, true, true>), grid, block, 0, g.stream, A);
    else hipLaunchKernelGGL((k_traverse<STACK, true, false>), grid, block, 0, g.stream, A);
  } else {
    if (count) hipLaunchKernelGGL((k_traverse<STACK, false, true>), grid, block, 0, g.stream, A);
    else hipLaunchKernelGGL((k_traverse<STACK, false, false>), grid, block, 0, g.stream, A);
  }
  return static_cast<int>(hipGetLastError());
}

constexpr int kWideLdsStack = 16;       // LDS stack entries per lane of k_traverse4 (16 KB per block)
constexpr size_t kCacheResidentBytes = 256u << 20; // nodes + triangles up to this size are served by L2 / Infinity Cache
#ifndef NOVA_P_MINWAVES
#define NOVA_P_MINWAVES 5 // A/B builds only (build_ab/): waves per SIMD the persistent kernel is compiled for
#endif
#ifndef NOVA_P_REFILL
#define NOVA_P_REFILL 32   // A/B builds only: idle lanes that trigger a refill, helper lanes in the leaf step
#define NOVA_P_HELP true
#endif
constexpr int kWideBlocksPerCu = NOVA_P_MINWAVES; // = the resident work-groups at 5 waves/SIMD (8 was 0-10 % slower on launches below 0.3 M rays)

// k_traverse4: four-wide tree where the scene has one, hybrid LDS/global stack
int launch_traverse_wide(const PackedScene& s, TraverseArgs A, bool count) {
  const uint32_t need = std::max(s.stack_need, s.stack_need_wide);
  // Work-groups per CU: 5 (every wave slot the kernel's registers allow) when the launch has the GPU to itself; fewer
  // when kernels of several samples are in flight (lanes), so that they share every CU instead of queueing for it —
  // 1080p frame, three lanes: 2.13 / 2.08 / 2.03 / 2.02 ms per sample with 5 / 4 / 3 / 2 (1 work-group: 2.41); the other
  // configurations, four-call and fused: profiles/r03_blocks_per_cu_ab.txt.  The price is the end of a burst, where the
  // last sample has the GPU to itself at reduced occupancy (profiles/r03_burst_length_ab.txt: 4 samples of the 4K room
  // 9.4 / 9.7 / 10.2 ms with 5 / 3 / 2, 12 samples 9.27 / 9.14 / 9.37): 3 wins from 8 samples per burst on and loses
  // least below.
  static const unsigned shared_per_cu = [] {
    const char* e = std::getenv("NOVA_HIP_SHARED_BLOCKS_PER_CU");
    const int v = e ? std::atoi(e) : 0;
    return v >= 1 && v <= 8 ? static_cast<unsigned>(v) : 3u;
  }();
  const unsigned per_cu = ((g.opt_tune >> 12) & 15) != 0 ? static_cast<unsigned>((g.opt_tune >> 12) & 15) // A/B
                          : (g.samples_in_flight > 1 ? shared_per_cu : static_cast<unsigned>(kWideBlocksPerCu));
  const unsigned blocks = static_cast<unsigned>(g.num_cus) * per_cu;
  const size_t threads = static_cast<size_t>(blocks) * 256;
  const size_t words = need > static_cast<uint32_t>(kWideLdsStack) ? (need - kWideLdsStack) * threads : 0;
  if (words > g.spill_words) {
    if (g.no_alloc) return NOVA_HIP_E_TOO_LARGE; // the static scratch array is what there is
    (void) hipStreamSynchronize(g.stream);
    (void) hipFree(g.d_spill);
    g.d_spill = nullptr;
    g.spill_words = 0;
    HIP_TRY(hipMalloc(&g.d_spill, words * sizeof(uint32_t)));
    g.spill_words = words;
  }
  A.wide = s.wide;
  A.spill = g.d_spill;
  A.rays_per_wave = (g.opt_tune & 0xFFF) > 0 ? static_cast<uint32_t>(g.opt_tune & 0xFFF) : 64u;
  const dim3 grid(blocks), block(256);
  if (s.wide != nullptr && g.opt_kernel != 6) {
    // Persistent lanes (production).  96 VGPRs, 5 waves/SIMD (6 spills).  Leaf steps with helper lanes (traverse.hpp)
    // where the scene fits the caches: 1080p frame 3.62 -> 3.50 ms, C3 17.0 -> 15.9, C5 13.3 -> 13.0; on the 0.6 GB soup
    // of C4, where a step waits for HBM, the helpers' exchange only lengthens it (16.4 -> 17.7 ms): there the plain
    // leaf step with a refill at 16 idle lanes.  With helpers the lanes that wait for rays are not idle: refill at 32
    // (8 / 16 / 24 / 32 / 40 / 48 idle lanes measured 3.58 / 3.50 / 3.49 / 3.46 / 3.61 / 3.82 ms per 1080p sample).
    // The A/B history of the other variants (no forking, 6 waves/SIMD, two-round leaf loads, four lanes per ray) is in
    // DESIGN.md §7 and profiles/; their code left the library in round 3.
    const bool resident = static_cast<size_t>(s.n_wide) * sizeof(Node128) + static_cast<size_t>(s.n_tris) * 48 <= kCacheResidentBytes;
    if (!resident) {
      if (count) hipLaunchKernelGGL((k_traverse_p<kWideLdsStack, true, 16, NOVA_P_MINWAVES, false>), grid, block, 0, g.stream, A);
      else hipLaunchKernelGGL((k_traverse_p<kWideLdsStack, false, 16, NOVA_P_MINWAVES, false>), grid, block, 0, g.stream, A);
    } else {
      if (count) hipLaunchKernelGGL((k_traverse_p<kWideLdsStack, true, NOVA_P_REFILL, NOVA_P_MINWAVES, NOVA_P_HELP>), grid, block, 0, g.stream, A);
      else hipLaunchKernelGGL((k_traverse_p<kWideLdsStack, false, NOVA_P_REFILL, NOVA_P_MINWAVES, NOVA_P_HELP>), grid, block, 0, g.stream, A);
    }
    return static_cast<int>(hipGetLastError());
  }
  if (count) hipLaunchKernelGGL((k_traverse4<kWideLdsStack, true>), grid, block, 0, g.stream, A);
  else hipLaunchKernelGGL((k_traverse4<kWideLdsStack, false>), grid, block, 0, g.stream, A);
  return static_cast<int>(hipGetLastError());
}

int launch_traverse(const PackedScene& s, TraverseArgs A, bool coherent) {
  A.nodes = s.nodes;
  A.wide = nullptr;
  A.spill = nullptr;
  A.n_tris = s.n_tris;
  A.tris = reinterpret_cast<const float4*>(s.tri_ptr);
  std::memcpy(A.root_top, s.root_top, sizeof(A.root_top));
  std::memcpy(A.root_bot, s.root_bot, sizeof(A.root_bot));
  A.inv_radius = 1.0f / s.radius;
  A.work_counters = g.d_counters;
  // The distance cull argues from "hit point inside the leaf box => inside every ancestor's box"
  // (traverse.hpp); a tree whose boxes are not nested gets the reference's own exhaustive walk.
  const bool cull = g.opt_traversal == 0 && s.wide != nullptr, count = g.opt_count_work != 0;
  hipEvent_t e0 = nullptr;
  span_begin(&e0);
  int rc;
  // production (opt_kernel 0): persistent lanes on the four-wide tree, except for camera rays — an 8x8-pixel batch
  // per wave is coherent enough for the packet kernel.  Validation / regression switches kept:
  // 6 = k_traverse4 (batch kernel on the four-wide tree), 11 = k_traverse (batch kernel on the box-pair tree),
  // 12 = the packet kernel for every launch.
  const bool wide = cull && ((g.opt_kernel == 0 && !coherent) || g.opt_kernel == 6);
  const bool packet = cull && ((g.opt_kernel == 0 && coherent) || g.opt_kernel == 12);
  // ray-queue flags (traverse.hpp: TraverseArgs::flags).  Production = 0xD: sub-queues dealt to XCDs
  // (mode 1), the rays of a path next to each other, camera-ray kernel included — measured on
  // the 1080p frame: 3.79 -> 3.30 ms of traversal (each XCD's L2 then serves one part of the frame).
  // NOVA_HIP_OPT_TUNE bits 16+ override it for A/B runs (0x100 = all off).
  A.flags = (g.opt_tune >> 16) != 0 ? static_cast<uint32_t>(g.opt_tune >> 16) & 0xFFu : 0xDu;
  if (packet) {
    // node and triangles through the scalar cache (the wave-uniform stack in registers instead of LDS made no
    // difference; nor did the same packet on the four-wide tree: profiles/r03_packet4_ab.txt)
    const unsigned packet_per_cu = ((g.opt_tune >> 24) & 15) != 0 ? static_cast<unsigned>((g.opt_tune >> 24) & 15) : 8u; // A/B
    const dim3 grid(static_cast<unsigned>(g.num_cus) * packet_per_cu), block(256);
    if (count) hipLaunchKernelGGL((k_traverse_packet<true>), grid, block, 0, g.stream, A, A.nodes, A.tris);
    else hipLaunchKernelGGL((k_traverse_packet<false>), grid, block, 0, g.stream, A, A.nodes, A.tris);
    rc = static_cast<int>(hipGetLastError());
  } else if (wide) rc = launch_traverse_wide(s, A, count);
  else rc = s.stack_need <= 32 ? launch_traverse_t<32>(A, cull, count) : launch_traverse_t<64>(A, cull, count);
  hipEvent_t e1 = span_end();
  if (e0 && e1) (coherent ? g.ev_camera : g.ev_traverse).push_back({ e0, e1 });
  g.traverse_launches++;
  if (coherent) g.camera_launches++;
  return rc;
}

int ensure_frame_buffers(uint32_t paths) {
  FrameBuffers& fb = g.fb;
  if (fb.capacity >= paths) return 0;
  (void) hipStreamSynchronize(g.stream); // kernels of an earlier, not yet awaited launch may still use the old ones
  auto release = [&]() {
    for (int s = 0; s < 2; s++) {
      for (int k = 0; k < 6; k++) { (void) hipFree(fb.planes[s][k]); fb.planes[s][k] = nullptr; }
      (void) hipFree(fb.ray_o[s]); fb.ray_o[s] = nullptr;
      (void) hipFree(fb.ray_d[s]); fb.ray_d[s] = nullptr;
    }
    (void) hipFree(fb.hits); fb.hits = nullptr;
    (void) hipFree(fb.block_counters); fb.block_counters = nullptr;
    fb.capacity = 0;
    fb.n_blocks = 0;
  };
  // counts accumulated so far live in the old per-block slots: fold them into the totals first
  if (fb.block_counters) {
    std::vector<unsigned long long> hb(kBlockCounters * static_cast<size_t>(fb.n_blocks));
    if (hipMemcpy(hb.data(), fb.block_counters, hb.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      for (size_t b = 0; b < fb.n_blocks; b++)
        for (int q = 0; q < kBlockCounters; q++) g.folded[q] += hb[kBlockCounters * b + q];
    }
  }
  release();
  const size_t plane = static_cast<size_t>(paths) * sizeof(float4);
  for (int s = 0; s < 2; s++) {
    for (int k = 0; k < 6; k++) {
      hipError_t e = hipMalloc(&fb.planes[s][k], plane);
      if (e != hipSuccess) { release(); return static_cast<int>(e); }
    }
    hipError_t e = hipMalloc(&fb.ray_o[s], 3 * plane);
    if (e == hipSuccess) e = hipMalloc(&fb.ray_d[s], 3 * plane);
    if (e != hipSuccess) { release(); return static_cast<int>(e); }
  }
  hipError_t e = hipMalloc(&fb.hits, 3 * plane);
  if (e != hipSuccess) { release(); return static_cast<int>(e); }
  fb.n_blocks = (paths + kShadeBlock - 1) / kShadeBlock;
  e = hipMalloc(&fb.block_counters, kBlockCounters * static_cast<size_t>(fb.n_blocks) * 8);
  if (e == hipSuccess) e = hipMemset(fb.block_counters, 0, kBlockCounters * static_cast<size_t>(fb.n_blocks) * 8);
  if (e != hipSuccess) { release(); return static_cast<int>(e); }
  fb.capacity = paths;
  return 0;
}

PathPlanes planes_of(int side) {
  FrameBuffers& fb = g.fb;
  return { fb.planes[side][0], fb.planes[side][1], fb.planes[side][2],
           fb.planes[side][3], fb.planes[side][4], fb.planes[side][5] };
}

// texels (other than x itself) that the pixel-centre sample of pixel x reads with a non-zero weight along one
// axis of n texels: the arithmetic of k_accumulate / sample_linear<false>, operation for operation
void axis_deps(uint32_t n, std::vector<std::vector<uint32_t>>& deps) {
  deps.assign(n, {});
  for (uint32_t x = 0; x < n; x++) {
    const float u = (static_cast<float>(static_cast<int>(x)) + 0.5f) / static_cast<float>(n);
    const float uw = u * static_cast<float>(n);
    const float fl = std::floor(uw - 0.5f);
    const float a = (uw - 0.5f) - fl;
    const int i0 = static_cast<int>(fl);
    const int cand[2] = { i0, i0 + 1 };
    const float wt[2] = { 1.0f - a, a };
    for (int k = 0; k < 2; k++)
      if (wt[k] != 0.0f && cand[k] >= 0 && cand[k] < static_cast<int>(n) && cand[k] != static_cast<int>(x))
        deps[x].push_back(static_cast<uint32_t>(cand[k]));
  }
}

std::vector<uint32_t> axis_closure(uint32_t lo, uint32_t hi, const std::vector<std::vector<uint32_t>>& deps) {
  std::vector<uint32_t> set;
  std::vector<uint8_t> in(deps.size(), 0);
  for (uint32_t x = lo; x <= hi; x++) { set.push_back(x); in[x] = 1; }
  for (size_t k = 0; k < set.size(); k++)
    for (uint32_t t : deps[set[k]])
      if (!in[t]) { in[t] = 1; set.push_back(t); }
  return set;
}

void free_tile_plan() {
  (void) hipFree(g.plan.d_mask);
  (void) hipFree(g.plan.d_extra);
  g.plan = TilePlan {};
}

// the plan of the current partition options for a width x height frame (made on the host, cached)
int ensure_tile_plan(uint32_t width, uint32_t height) {
  const uint32_t rank = static_cast<uint32_t>(g.opt_tile_rank), ranks = static_cast<uint32_t>(g.opt_tile_ranks);
  TilePlan& P = g.plan;
  if (P.d_mask && P.w == width && P.h == height && P.rank == rank && P.ranks == ranks) return 0;
  HIP_TRY(hipDeviceSynchronize()); // kernels of any lane may still read the old one
  free_tile_plan();
  std::vector<std::vector<uint32_t>> dx, dy;
  axis_deps(width, dx);
  axis_deps(height, dy);
  const uint32_t tiles_x = (width + kTileSize - 1) / kTileSize, tiles_y = (height + kTileSize - 1) / kTileSize;
  std::vector<uint8_t> mask(static_cast<size_t>(width) * height, 0);
  for (uint32_t t = rank; t < tiles_x * tiles_y; t += ranks) {
    const uint32_t x0 = (t % tiles_x) * kTileSize, y0 = (t / tiles_x) * kTileSize;
    for (uint32_t y = y0; y < std::min(y0 + kTileSize, height); y++)
      std::memset(&mask[static_cast<size_t>(y) * width + x0], 1, std::min(kTileSize, width - x0));
  }
  std::vector<uint32_t> extra;
  for (uint32_t t = rank; t < tiles_x * tiles_y; t += ranks) {
    const uint32_t x0 = (t % tiles_x) * kTileSize, y0 = (t / tiles_x) * kTileSize;
    const uint32_t x1 = std::min(x0 + kTileSize, width) - 1, y1 = std::min(y0 + kTileSize, height) - 1;
    const std::vector<uint32_t> cols = axis_closure(x0, x1, dx), rows = axis_closure(y0, y1, dy);
    if (cols.size() == x1 - x0 + 1 && rows.size() == y1 - y0 + 1) continue;
    for (uint32_t y : rows)
      for (uint32_t x : cols) {
        uint8_t& m = mask[static_cast<size_t>(y) * width + x];
        if (m == 0) m = 2;
      }
  }
  for (size_t i = 0; i < mask.size(); i++)
    if (mask[i] == 2) extra.push_back(static_cast<uint32_t>(i));
  HIP_TRY(hipMalloc(&P.d_mask, mask.size()));
  hipError_t e = hipMemcpy(P.d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&P.d_extra, std::max<size_t>(extra.size(), 1) * sizeof(uint32_t));
  if (e == hipSuccess && !extra.empty())
    e = hipMemcpy(P.d_extra, extra.data(), extra.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) { free_tile_plan(); return static_cast<int>(e); }
  P.w = width; P.h = height; P.rank = rank; P.ranks = ranks;
  P.n_extra = static_cast<uint32_t>(extra.size());
  return 0;
}


// ---- RCCL, opened on first use ------------------------------------------------------------------
// Single-GPU callers never load the 0.5 GB library.  It must be the librccl that belongs to the HIP runtime this
// library is bound to — a process may hold two (PyTorch ships its own libamdhip64 + librccl next to ROCm's), and a
// stream of one runtime means nothing to the other — so it is opened by path, in the directory hipMalloc lives in.
struct Rccl {
  bool tried = false, ok = false;
  void* handle = nullptr;
  char path[512] = {};
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};
Rccl& rccl() {
  static Rccl R;
  return R;
}
bool load_rccl() {
  Rccl& R = rccl();
  if (R.tried) return R.ok;
  R.tried = true;
  std::string dir;
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
    dir = info.dli_fname;
    const size_t slash = dir.rfind('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
  }
  const char* env = std::getenv("NOVA_HIP_RCCL"); // full path of the library, for unusual installations
  std::vector<std::string> tries;
  if (env && *env) tries.push_back(env);
  if (!dir.empty()) {
    tries.push_back(dir + "librccl.so.1");
    tries.push_back(dir + "librccl.so");
  }
  tries.push_back("librccl.so.1"); // the loader's own search (an already loaded copy, LD_LIBRARY_PATH, RUNPATH)
  for (const std::string& t : tries) {
    R.handle = dlopen(t.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (R.handle) {
      std::snprintf(R.path, sizeof(R.path), "%s", t.c_str());
      break;
    }
  }
  if (!R.handle) return false;
  R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(dlsym(R.handle, "ncclGetUniqueId"));
  R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(dlsym(R.handle, "ncclCommInitRank"));
  R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(dlsym(R.handle, "ncclCommDestroy"));
  R.CommCount = reinterpret_cast<decltype(R.CommCount)>(dlsym(R.handle, "ncclCommCount"));
  R.CommUserRank = reinterpret_cast<decltype(R.CommUserRank)>(dlsym(R.handle, "ncclCommUserRank"));
  R.AllGather = reinterpret_cast<decltype(R.AllGather)>(dlsym(R.handle, "ncclAllGather"));
  R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.handle, "ncclGetErrorString"));
  R.GetVersion = reinterpret_cast<decltype(R.GetVersion)>(dlsym(R.handle, "ncclGetVersion"));
  R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.CommCount && R.CommUserRank && R.AllGather &&
         R.GetErrorString && R.GetVersion;
  return R.ok;
}
const char* rccl_error_string(int nccl_code) {
  if (rccl().ok) return rccl().GetErrorString(static_cast<ncclResult_t>(nccl_code));
  return "nova_hip: RCCL error";
}
#define RCCL_TRY(expr)                                                     \
  do {                                                                     \
    ncclResult_t r_ = (expr);                                              \
    if (r_ != ncclSuccess) return NOVA_HIP_E_RCCL_BASE - static_cast<int>(r_); \
  } while (0)

bool float_image_ok(const nova_hip_image* im, uint32_t w, uint32_t h) {
  return im && im->data && im->texel_bytes == 16 && im->width == w && im->height == h && im->layers >= 1;
}

}

extern "C" {

const char* nova_hip_error_string(int code) {
  switch (code) {
    case NOVA_HIP_OK: return "success";
    case NOVA_HIP_E_INVALID_ARGUMENT: return "nova_hip: invalid argument";
    case NOVA_HIP_E_UNKNOWN_BUFFER: return "nova_hip: device pointer was not allocated by nova_hip_malloc";
    case NOVA_HIP_E_BAD_SCENE: return "nova_hip: malformed BVH";
    case NOVA_HIP_E_NO_DEVICE: return "nova_hip: no usable HIP device";
    case NOVA_HIP_E_TOO_LARGE: return "nova_hip: request exceeds a documented maximum";
    case NOVA_HIP_E_NO_RCCL: return "nova_hip: librccl could not be opened next to the HIP runtime in use, or lacks a symbol";
    case NOVA_HIP_E_NO_COMM: return "nova_hip: no communicator (nova_hip_tiles_comm_init / _adopt comes first)";
    case NOVA_HIP_E_NOT_CACHED: return "nova_hip: no traversal layout cached for these buffers (an allocating call must see them first)";
    default: break;
  }
  if (code <= NOVA_HIP_E_RCCL_BASE && code > NOVA_HIP_E_RCCL_BASE - 100) return rccl_error_string(NOVA_HIP_E_RCCL_BASE - code);
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "nova_hip: unknown error";
}

int nova_hip_device_count(int* count) {
  if (!count) return NOVA_HIP_E_INVALID_ARGUMENT;
  *count = 0;
  hipError_t e = hipGetDeviceCount(count);
  if (e == hipErrorNoDevice) { *count = 0; return 0; }
  return static_cast<int>(e);
}

int nova_hip_init(int device_ordinal) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (g.ready && g.device == device_ordinal) return 0;
  if (g.ready) return NOVA_HIP_E_INVALID_ARGUMENT; // one device per process
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return NOVA_HIP_E_NO_DEVICE;
  if (device_ordinal < 0 || device_ordinal >= n) return NOVA_HIP_E_NO_DEVICE;
  HIP_TRY(hipSetDevice(device_ordinal));
  // every GPU holds its own copy of `scratch` at its own address: resolved here, with the device current
  HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&g.scratch_ptr), HIP_SYMBOL(scratch)));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
  g.num_cus = prop.multiProcessorCount;
  HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
  // one allocation, so that one fill per frame clears both: [kQueueWords counters][queue heads]
  HIP_TRY(hipMalloc(&g.d_queues, (kQueueWords + kHeadWords * (kMaxIterations + 2)) * sizeof(uint32_t)));
  g.d_heads = g.d_queues + kQueueWords;
  HIP_TRY(hipMalloc(&g.d_counters, 8 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(g.d_counters, 0, 8 * sizeof(unsigned long long)));
  HIP_TRY(hipHostMalloc(&g.h_pinned, 64, hipHostMallocDefault));
  HIP_TRY(hipMalloc(&g.d_light_tris, kLightTriWords * sizeof(uint32_t)));
  g.device = device_ordinal;
  g.ready = true;
  return 0;
}

int nova_hip_device_name(char* out, size_t out_bytes) {
  if (!out || out_bytes == 0) return NOVA_HIP_E_INVALID_ARGUMENT;
  int rc = require_ctx();
  if (rc) return rc;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, g.device));
  std::snprintf(out, out_bytes, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return 0;
}

int nova_hip_synchronize(void) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(g.stream));
  collect_events();
  return 0;
}

int nova_hip_shutdown(void) {
  if (int rc_settle = settle()) return rc_settle;
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!g.ready) return 0;
  (void) hipDeviceSynchronize();
  collect_events();
  switch_lane(0); // the context's own fields are lane 0 again; lanes[1..] hold the others
  for (auto& s : g.scenes) free_scene(s);
  g.scenes.clear();
  free_tile_plan();
  if (g.comm && g.comm_owned && rccl().ok) (void) rccl().CommDestroy(g.comm);
  (void) hipFree(g.d_gather);
  (void) hipFree(g.d_ray_keys);
  (void) hipFree(g.d_ray_perm);
  (void) hipFree(g.d_ray_hist);
  (void) hipFree(g.d_denoise);
  (void) hipHostFree(g.h_denoise);
  free_frame_buffers(g.fb);
  (void) hipFree(g.d_spill);
  (void) hipFree(g.d_queues);
  (void) hipFree(g.d_counters);
  (void) hipFree(g.d_light_tris);
  (void) hipHostFree(g.h_pinned);
  (void) hipStreamDestroy(g.stream);
  for (int l = 1; l < g.lanes_ready; l++) {
    Lane& o = g.lanes[l];
    free_frame_buffers(o.fb);
    (void) hipFree(o.d_spill);
    (void) hipFree(o.d_queues);
    (void) hipHostFree(o.h_pinned);
    if (o.stream) (void) hipStreamDestroy(o.stream);
  }
  for (auto& lane : g.stage)
    for (auto& p : lane) (void) hipFree(p);
  for (auto& e : g.acc_done)
    if (e) (void) hipEventDestroy(e);
  for (auto& e : g.lane_mark)
    if (e) (void) hipEventDestroy(e);
  for (auto& lane : g.call_stage)
    for (auto& p : lane) (void) hipFree(p);
  for (hipEvent_t e : g.ev_pool) (void) hipEventDestroy(e);
  for (auto& pd : g_pending) pd = Pending {};
  g_deferred.clear();
  g_image_lane = -1; // its event went with the context
  g_issue_lane = 0;
  // buffers from nova_hip_malloc belong to the caller and stay valid; the registry goes with them
  // ... and so do the options: they are per process (include/nova_hip.h), not per device context.  Counters
  // (rays, kernel times) are per context and start again at zero.
  std::unordered_map<const void*, Alloc> allocs = std::move(g.allocs);
  const uint64_t next_version = g.next_version;
  const int64_t opts[12] = { g.opt_max_depth, g.opt_traversal, g.opt_count_work, g.opt_tile_rank, g.opt_tile_ranks,
                             g.opt_kernel, g.opt_tune, g.opt_clamp_to_edge, g.opt_sort_rays, g.opt_mis_pretest,
                             g.opt_async, g.opt_call_lanes };
  g = Ctx {};
  g.allocs = std::move(allocs);
  g.next_version = next_version;
  g.opt_max_depth = opts[0]; g.opt_traversal = opts[1]; g.opt_count_work = opts[2];
  g.opt_tile_rank = opts[3]; g.opt_tile_ranks = opts[4]; g.opt_kernel = opts[5];
  g.opt_tune = opts[6]; g.opt_clamp_to_edge = opts[7]; g.opt_sort_rays = opts[8]; g.opt_mis_pretest = opts[9]; g.opt_async = opts[10]; g.opt_call_lanes = opts[11];
  return 0;
}

// ---- buffers -----------------------------------------------------------------------------------
int nova_hip_malloc(void** dev_ptr, size_t bytes) {
  if (!dev_ptr || bytes == 0) return NOVA_HIP_E_INVALID_ARGUMENT;
  int rc = require_ctx();
  if (rc) return rc;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, bytes));
  g.allocs[p] = { bytes, g.next_version++ };
  *dev_ptr = p;
  return 0;
}

int nova_hip_free(void* dev_ptr) {
  if (int rc_settle = settle()) return rc_settle;
  if (!dev_ptr) return 0;
  auto it = g.allocs.find(dev_ptr);
  if (it == g.allocs.end()) return NOVA_HIP_E_UNKNOWN_BUFFER;
  drop_scenes_using(dev_ptr);
  g.allocs.erase(it);
  HIP_TRY(hipFree(dev_ptr));
  return 0;
}

int nova_hip_host_alloc(void** host_ptr, size_t bytes) {
  if (!host_ptr || bytes == 0) return NOVA_HIP_E_INVALID_ARGUMENT;
  int rc = require_ctx();
  if (rc) return rc;
  HIP_TRY(hipHostMalloc(host_ptr, bytes, hipHostMallocDefault));
  return 0;
}

int nova_hip_host_free(void* host_ptr) {
  if (!host_ptr) return 0;
  HIP_TRY(hipHostFree(host_ptr));
  return 0;
}

int nova_hip_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes) {
  if (int rc_settle = settle()) return rc_settle;
  if (!dst_dev || !src_host) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (bytes == 0) return 0;
  HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  touch(dst_dev);
  return 0;
}

int nova_hip_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes) {
  if (int rc_settle = settle()) return rc_settle;
  if (!dst_host || !src_dev) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (bytes == 0) return 0;
  HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int nova_hip_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes) {
  if (int rc_settle = settle()) return rc_settle;
  if (!dst_dev || !src_dev) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (bytes == 0) return 0;
  HIP_TRY(hipMemcpy(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice));
  touch(dst_dev);
  return 0;
}

int nova_hip_fill(void* dst_dev, const void* pattern, size_t pattern_bytes, size_t count) {
  if (int rc_settle = settle()) return rc_settle;
  if (!dst_dev || !pattern) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (count == 0) return 0;
  int rc = require_ctx();
  if (rc) return rc;
  const unsigned blocks = static_cast<unsigned>(std::min<size_t>((count + 255) / 256, 4096));
  switch (pattern_bytes) {
    case 1: { uint8_t v; std::memcpy(&v, pattern, 1);
      hipLaunchKernelGGL(k_fill<uint8_t>, dim3(blocks), dim3(256), 0, g.stream, static_cast<uint8_t*>(dst_dev), v, count); break; }
    case 2: { uint16_t v; std::memcpy(&v, pattern, 2);
      hipLaunchKernelGGL(k_fill<uint16_t>, dim3(blocks), dim3(256), 0, g.stream, static_cast<uint16_t*>(dst_dev), v, count); break; }
    case 4: { uint32_t v; std::memcpy(&v, pattern, 4);
      hipLaunchKernelGGL(k_fill<uint32_t>, dim3(blocks), dim3(256), 0, g.stream, static_cast<uint32_t*>(dst_dev), v, count); break; }
    case 8: { uint2 v; std::memcpy(&v, pattern, 8);
      hipLaunchKernelGGL(k_fill<uint2>, dim3(blocks), dim3(256), 0, g.stream, static_cast<uint2*>(dst_dev), v, count); break; }
    case 16: { uint4 v; std::memcpy(&v, pattern, 16);
      hipLaunchKernelGGL(k_fill<uint4>, dim3(blocks), dim3(256), 0, g.stream, static_cast<uint4*>(dst_dev), v, count); break; }
    default: return NOVA_HIP_E_INVALID_ARGUMENT;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(g.stream));
  touch(dst_dev);
  return 0;
}

// ---- images ------------------------------------------------------------------------------------
int nova_hip_image_create(nova_hip_image** image, uint32_t texel_bytes, uint32_t width, uint32_t height,
                          uint32_t layers, uint32_t address_mode, uint32_t filter_mode,
                          uint32_t normalized_coords, const void* host_data) {
  if (!image || width == 0 || height == 0 || layers == 0 || (texel_bytes != 16 && texel_bytes != 4))
    return NOVA_HIP_E_INVALID_ARGUMENT;
  int rc = require_ctx();
  if (rc) return rc;
  const size_t bytes = static_cast<size_t>(width) * height * layers * texel_bytes;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, bytes));
  // an image without initial data starts as zeros: under a tile partition kernel_raytrace never writes
  // other ranks' pixels, and stale NaN/Inf words there must not reach this rank's readers
  hipError_t e = host_data ? hipMemcpy(p, host_data, bytes, hipMemcpyHostToDevice) : hipMemset(p, 0, bytes);
  if (e != hipSuccess) { (void) hipFree(p); return static_cast<int>(e); }
  nova_hip_image* im = new nova_hip_image { p, width, height, layers, texel_bytes, address_mode, filter_mode, normalized_coords };
  *image = im;
  return 0;
}

int nova_hip_image_destroy(nova_hip_image* image) {
  if (int rc_settle = settle()) return rc_settle;
  if (!image) return 0;
  hipError_t e = hipFree(image->data);
  delete image;
  return static_cast<int>(e);
}

static size_t image_bytes(const nova_hip_image* im) {
  return static_cast<size_t>(im->width) * im->height * im->layers * im->texel_bytes;
}

int nova_hip_image_read(const nova_hip_image* image, void* host_dst) {
  if (int rc_settle = settle()) return rc_settle;
  if (!image || !image->data || !host_dst) return NOVA_HIP_E_INVALID_ARGUMENT;
  HIP_TRY(hipMemcpy(host_dst, image->data, image_bytes(image), hipMemcpyDeviceToHost));
  return 0;
}

int nova_hip_image_write(nova_hip_image* image, const void* host_src) {
  if (int rc_settle = settle()) return rc_settle;
  if (!image || !image->data || !host_src) return NOVA_HIP_E_INVALID_ARGUMENT;
  HIP_TRY(hipMemcpy(image->data, host_src, image_bytes(image), hipMemcpyHostToDevice));
  return 0;
}

static int image_work(std::function<int()> run); // enqueue now, or behind a sample that is not whole yet (Deferred)

int nova_hip_image_copy(nova_hip_image* dst, const nova_hip_image* src) {
  if (!dst || !src || !dst->data || !src->data || image_bytes(dst) != image_bytes(src))
    return NOVA_HIP_E_INVALID_ARGUMENT;
  int rc = require_ctx();
  if (rc) return rc;
  void* const to = dst->data;
  const void* const from = src->data;
  const size_t bytes = image_bytes(dst);
  return image_work([=]() -> int {
    HIP_TRY(hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, g.stream));
    return 0;
  });
}

int nova_hip_image_fill(nova_hip_image* image, const void* texel) {
  if (!image || !image->data || !texel) return NOVA_HIP_E_INVALID_ARGUMENT;
  return nova_hip_fill(image->data, texel, image->texel_bytes,
                       static_cast<size_t>(image->width) * image->height * image->layers);
}

// ---- options / counters ------------------------------------------------------------------------
int nova_hip_set_option(int option, int64_t value) {
  if (int rc_settle = settle()) return rc_settle;
  switch (option) {
    case NOVA_HIP_OPT_MAX_DEPTH:
      if (value < 0 || value >= kMaxIterations) return NOVA_HIP_E_INVALID_ARGUMENT;
      g.opt_max_depth = value; return 0;
    case NOVA_HIP_OPT_TRAVERSAL:
      if (value != 0 && value != 1) return NOVA_HIP_E_INVALID_ARGUMENT;
      g.opt_traversal = value; return 0;
    case NOVA_HIP_OPT_COUNT_WORK: g.opt_count_work = value != 0; return 0;
    case NOVA_HIP_OPT_TILE_RANK:
      if (value < 0) return NOVA_HIP_E_INVALID_ARGUMENT;
      g.opt_tile_rank = value; return 0;
    case NOVA_HIP_OPT_TILE_RANKS:
      if (value < 1) return NOVA_HIP_E_INVALID_ARGUMENT;
      g.opt_tile_ranks = value; return 0;
    case NOVA_HIP_OPT_TRAVERSE_KERNEL:
      if (value != 0 && value != 6 && value != 11 && value != 12) return NOVA_HIP_E_INVALID_ARGUMENT;
      g.opt_kernel = value; return 0;
    case NOVA_HIP_OPT_TUNE: g.opt_tune = value; return 0;
    case NOVA_HIP_OPT_SORT_RAYS:
      if (value < 0 || value > 2) return NOVA_HIP_E_INVALID_ARGUMENT;
      g.opt_sort_rays = value; return 0;
    case NOVA_HIP_OPT_CLAMP_TO_EDGE:
      if (value < 0 || value > 1) return NOVA_HIP_E_INVALID_ARGUMENT;
      g.opt_clamp_to_edge = value; return 0;
    case NOVA_HIP_OPT_MIS_PRETEST:
      if (value < 0 || value > 1) return NOVA_HIP_E_INVALID_ARGUMENT;
      g.opt_mis_pretest = value; return 0;
    case NOVA_HIP_OPT_ASYNC_LAUNCH:
      if (value < 0 || value > 1) return NOVA_HIP_E_INVALID_ARGUMENT;
      g.opt_async = value; return 0;
    case NOVA_HIP_OPT_CALL_LANES:
      if (value < 0 || value > kMaxLanes) return NOVA_HIP_E_INVALID_ARGUMENT;
      g.opt_call_lanes = value; return 0;
    default: return NOVA_HIP_E_INVALID_ARGUMENT;
  }
}

int nova_hip_get_option(int option, int64_t* value) {
  if (!value) return NOVA_HIP_E_INVALID_ARGUMENT;
  switch (option) {
    case NOVA_HIP_OPT_MAX_DEPTH: *value = g.opt_max_depth; return 0;
    case NOVA_HIP_OPT_TRAVERSAL: *value = g.opt_traversal; return 0;
    case NOVA_HIP_OPT_COUNT_WORK: *value = g.opt_count_work; return 0;
    case NOVA_HIP_OPT_TILE_RANK: *value = g.opt_tile_rank; return 0;
    case NOVA_HIP_OPT_TILE_RANKS: *value = g.opt_tile_ranks; return 0;
    case NOVA_HIP_OPT_TRAVERSE_KERNEL: *value = g.opt_kernel; return 0;
    case NOVA_HIP_OPT_TUNE: *value = g.opt_tune; return 0;
    case NOVA_HIP_OPT_SORT_RAYS: *value = g.opt_sort_rays; return 0;
    case NOVA_HIP_OPT_CLAMP_TO_EDGE: *value = g.opt_clamp_to_edge; return 0;
    case NOVA_HIP_OPT_MIS_PRETEST: *value = g.opt_mis_pretest; return 0;
    case NOVA_HIP_OPT_ASYNC_LAUNCH: *value = g.opt_async; return 0;
    case NOVA_HIP_OPT_CALL_LANES: *value = g.opt_call_lanes; return 0;
    case NOVA_HIP_OPT_ALLOC_CALLS: *value = static_cast<int64_t>(g_alloc_calls.load()); return 0;
    default: return NOVA_HIP_E_INVALID_ARGUMENT;
  }
}

int nova_hip_get_counters(nova_hip_counters* out) {
  if (int rc_settle = settle()) return rc_settle;
  if (!out) return NOVA_HIP_E_INVALID_ARGUMENT;
  int rc = require_ctx();
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(g.stream));
  collect_events();
  unsigned long long h[8];
  HIP_TRY(hipMemcpy(h, g.d_counters, sizeof(h), hipMemcpyDeviceToHost));
  uint64_t bc[kBlockCounters];
  for (int q = 0; q < kBlockCounters; q++) bc[q] = g.folded[q];
  rc = add_block_counters(g.fb, bc);
  for (int l = 0; l < g.lanes_ready && rc == 0; l++)
    if (l != g.cur) rc = add_block_counters(g.lanes[l].fb, bc);
  if (rc) return rc;
  out->node_visits = h[0];
  out->triangle_tests = h[1];
  out->rays_primary = g.host_primary;
  out->rays_bounce = bc[0] + bc[1];
  out->rays_shadow = bc[2];
  out->rays_mis_pretested = bc[4];
  out->rays_shadow_skipped = bc[5];
  out->path_segments = bc[3];
  out->traverse_ms = g.traverse_ms;
  out->shade_ms = g.shade_ms;
  out->traverse_launches = g.traverse_launches;
  out->camera_traverse_ms = g.camera_ms;
  out->camera_traverse_launches = g.camera_launches;
  return 0;
}

int nova_hip_reset_counters(void) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(g.stream));
  collect_events();
  HIP_TRY(hipMemset(g.d_counters, 0, 8 * sizeof(unsigned long long)));
  if (g.fb.block_counters) HIP_TRY(hipMemset(g.fb.block_counters, 0, kBlockCounters * static_cast<size_t>(g.fb.n_blocks) * 8));
  for (int l = 0; l < g.lanes_ready; l++)
    if (l != g.cur && g.lanes[l].fb.block_counters)
      HIP_TRY(hipMemset(g.lanes[l].fb.block_counters, 0, kBlockCounters * static_cast<size_t>(g.lanes[l].fb.n_blocks) * 8));
  for (auto& f : g.folded) f = 0;
  g.host_primary = 0;
  g.traverse_ms = g.camera_ms = g.shade_ms = 0;
  g.traverse_launches = g.camera_launches = 0;
  return 0;
}

// ---- kernel_raytrace ---------------------------------------------------------------------------
// One path segment for every live path of the current lane: traverse the rays iteration `it` reads,
// then shade (resolve segment it-1's direct light, shade segment it, emit the next rays).
static int enqueue_iteration(const PackedScene& scene, const FrameArgs& F, uint32_t P, uint32_t it, bool nee) {
  const int in = it & 1, out = in ^ 1;
  TraverseArgs T;
  std::memset(&T, 0, sizeof(T));
  T.ray_o = g.fb.ray_o[in];
  T.ray_d = g.fb.ray_d[in];
  T.hits = g.fb.hits;
  T.n_items_ptr = g.d_queues + it;
  // bounce iterations: the continuation and shadow rays of every path (kinds 0 and 1), then the MIS rays that k_shade
  // did not answer itself, compacted into the slots of kind 2
  T.kinds = (nee && it > 0) ? 2u : 1u;
  T.n_extra_ptr = (nee && it > 0) ? g.d_queues + kMisWords + it : nullptr;
  T.stride = P;
  T.queue_head = g.d_heads + kHeadWords * it;
  int rc = launch_traverse(scene, T, it == 0);
  if (rc) return rc;
  const unsigned shade_blocks = (P + kShadeBlock - 1) / kShadeBlock;
  hipEvent_t e0 = nullptr;
  span_begin(&e0);
  hipLaunchKernelGGL(k_shade, dim3(shade_blocks), dim3(kShadeBlock), 0, g.stream, F, it, planes_of(in), planes_of(out),
                     g.fb.ray_o[in], g.fb.ray_d[in], g.fb.hits, g.fb.ray_o[out], g.fb.ray_d[out]);
  hipEvent_t e1 = span_end();
  if (e0 && e1) g.ev_shade.push_back({ e0, e1 });
  return static_cast<int>(hipGetLastError());
}

// A sample of nova_hip_render_samples whose first iterations are enqueued on its lane.  With unbounded
// paths (max_depth 0) only iterations 0..kSpeculativeLast are enqueued at once; whether any path is
// still alive is looked at when the lane comes round again (raytrace_finish) — by then the answer is
// there without waiting, and the few long paths' launches that follow run beside the other lanes' samples.
constexpr uint32_t kSpeculativeLast = 7; // NOVA_HIP_SPECULATIVE_LAST, NOVA_HIP_POLL_STEP: A/B overrides
constexpr uint32_t kPollStep = 2;        // iterations enqueued per look at the survivor count
static uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* e = std::getenv(name);
  return e && *e ? static_cast<uint32_t>(std::strtoul(e, nullptr, 10)) : dflt;
}

static int raytrace_finish(Pending& pd) {
  if (!pd.active) return 0;
  pd.active = false;
  if (pd.next_it > pd.last_it) return 0;
  PackedScene* scene = nullptr;
  int rc = get_scene(pd.bvh, pd.triangles, &scene);
  if (rc) return rc;
  while (pd.next_it <= pd.last_it) {
    HIP_TRY(hipMemcpyAsync(g.h_pinned, g.d_queues + pd.next_it, sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    g.chain_tail = nullptr;
    if (*g.h_pinned == 0) break; // nobody left to take segment next_it
    static const uint32_t step = std::max(1u, env_u32("NOVA_HIP_POLL_STEP", kPollStep));
    for (uint32_t k = 0; k < step && pd.next_it <= pd.last_it; k++) {
      rc = enqueue_iteration(*scene, pd.F, pd.P, pd.next_it++, pd.nee);
      if (rc) return rc;
    }
  }
  return 0;
}

// Runs the kept image work of every sample up to `through_sample`, oldest first (see Deferred).
static int flush_deferred(uint64_t through_sample) {
  while (!g_deferred.empty() && g_deferred.front().sample <= through_sample) {
    Deferred op = std::move(g_deferred.front());
    g_deferred.pop_front();
    switch_lane(op.lane);
    if (op.sample != g_sample_finished) { // the first piece of work that needs this sample whole
      g.samples_in_flight = g.lanes_ready;
      const int rc = raytrace_finish(g_pending[op.lane]);
      g.samples_in_flight = 1;
      if (rc) return rc;
      g_sample_finished = op.sample;
    }
    if (g_image_lane >= 0 && g_image_lane != op.lane) HIP_TRY(hipStreamWaitEvent(g.stream, g.lane_mark[g_image_lane], 0));
    const int rc = op.run();
    if (rc) return rc;
    g.chain_tail = nullptr;
    HIP_TRY(hipEventRecord(g.lane_mark[op.lane], g.stream));
    g_image_lane = op.lane;
  }
  return 0;
}
// image work of the four-call schedule: now, or behind the sample it belongs to
static int image_work(std::function<int()> run) {
  if (g_deferred.empty()) {
    const int rc = run();
    return rc ? rc : launched();
  }
  g_deferred.push_back({ g_sample_seq, g_deferred.back().lane, std::move(run) });
  g.unsettled = true;
  return 0;
}

// the list of triangles with a light index for k_shade's pre-test, made once per contents of the meta buffer
static int ensure_light_tris(const nova_triangle_meta* meta, const Alloc* am, const nova_triangle_data* tris, const Alloc* at,
                             uint32_t n_tris) {
  if (g.lt_meta == meta && g.lt_version == am->version && g.lt_tri_ptr == tris && g.lt_tri_version == at->version &&
      g.lt_tris == n_tris)
    return 0;
  HIP_TRY(hipDeviceSynchronize()); // kernels of any lane may still read the old list
  HIP_TRY(hipMemsetAsync(g.d_light_tris, 0, kLightTriWords * sizeof(uint32_t), g.stream));
  hipLaunchKernelGGL(k_find_light_tris, dim3((n_tris + 255) / 256), dim3(256), 0, g.stream, meta,
                     reinterpret_cast<const float4*>(tris), n_tris, g.d_light_tris);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(g.stream));
  g.chain_tail = nullptr;
  g.lt_meta = meta;
  g.lt_version = am->version;
  g.lt_tri_ptr = tris;
  g.lt_tri_version = at->version;
  g.lt_tris = n_tris;
  return 0;
}

static int ensure_lanes(int lanes, size_t plane_bytes, bool features);
// the current lane's own radiance planes into the caller's images (under a tile partition this rank's pixels only:
// kernel_raytrace leaves the others as they are)
static int copy_lane_planes(float4* const dst[3], int n_planes, uint32_t width, uint32_t height) {
  const size_t plane = static_cast<size_t>(width) * height * 16;
  if (g.opt_tile_ranks > 1) {
    int rc = ensure_tile_plan(width, height);
    if (rc) return rc;
    const uint32_t n_pixels = width * height;
    const unsigned blocks = std::min<unsigned>((n_pixels + 255) / 256, static_cast<unsigned>(g.num_cus) * 8u);
    for (int k = 0; k < n_planes; k++) {
      hipLaunchKernelGGL(k_copy_kept, dim3(blocks), dim3(256), 0, g.stream, dst[k], g.call_stage[g.cur][k], g.plan.d_mask, n_pixels);
      HIP_TRY(hipGetLastError());
    }
  } else {
    for (int k = 0; k < n_planes; k++)
      HIP_TRY(hipMemcpyAsync(dst[k], g.call_stage[g.cur][k], plane, hipMemcpyDeviceToDevice, g.stream));
  }
  g.chain_tail = nullptr;
  return 0;
}
// every lane's path state and stack spill as large as the current lane's (an allocation of ~1 GB waits for the
// device: better in the first call than in the middle of a caller's loop)
static int size_other_lanes(int L) {
  const uint32_t paths = g.fb.capacity;
  const size_t spill = g.spill_words;
  const int mine = g.cur;
  int rc = 0;
  for (int l = 0; l < L && rc == 0; l++) {
    if (l == mine || (g.lanes[l].fb.capacity >= paths && g.lanes[l].spill_words >= spill)) continue;
    switch_lane(l);
    rc = ensure_frame_buffers(paths);
    if (rc == 0 && g.spill_words < spill) { // the traversal stack beyond its LDS part (launch_traverse_wide)
      (void) hipStreamSynchronize(g.stream);
      (void) hipFree(g.d_spill);
      g.d_spill = nullptr;
      g.spill_words = 0;
      hipError_t e = hipMalloc(&g.d_spill, spill * sizeof(uint32_t));
      if (e == hipSuccess) g.spill_words = spill;
      else rc = static_cast<int>(e);
    }
  }
  switch_lane(mine);
  return rc;
}
static int ensure_call_lanes(int lanes, size_t plane_bytes, bool features) {
  int rc = ensure_lanes(lanes, 0, false);
  if (rc) return rc;
  for (int l = 0; l < lanes; l++)
    if (!g.lane_mark[l]) HIP_TRY(hipEventCreateWithFlags(&g.lane_mark[l], hipEventDisableTiming));
  if (plane_bytes > g.call_stage_bytes) {
    HIP_TRY(hipDeviceSynchronize());
    for (auto& lane : g.call_stage)
      for (auto& p : lane) { (void) hipFree(p); p = nullptr; }
    g.call_stage_bytes = plane_bytes;
  }
  for (int l = 0; l < lanes; l++)
    for (int k = 0; k < (features ? 3 : 1); k++)
      if (!g.call_stage[l][k]) {
        HIP_TRY(hipMalloc(&g.call_stage[l][k], g.call_stage_bytes));
        HIP_TRY(hipMemset(g.call_stage[l][k], 0, g.call_stage_bytes));
      }
  return 0;
}

// `sync` = false: everything is left enqueued on the library's stream (nova_hip_render_samples);
// `pend` != NULL: see Pending.
static int raytrace_impl(const nova_scene_params* params, uint32_t time, nova_hip_image* temp_color1,
                         uint32_t width, uint32_t height, const nova_triangle_data* triangles,
                         const nova_triangle_meta* tri_meta, const nova_bvh_node* bvh,
                         const nova_area_light* lights, uint32_t num_lights,
                         const nova_hip_image* materials, const nova_hip_image* sky,
                         uint32_t denoise_available, nova_hip_image* albedo_feature1,
                         nova_hip_image* normal_feature1, bool sync, Pending* pend = nullptr) {
  if (pend) pend->active = false;
  int rc = require_ctx();
  if (rc) return rc;
  if (!params || !triangles || !tri_meta || !bvh || width == 0 || height == 0) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (!float_image_ok(temp_color1, width, height)) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (denoise_available && (!float_image_ok(albedo_feature1, width, height) || !float_image_ok(normal_feature1, width, height)))
    return NOVA_HIP_E_INVALID_ARGUMENT;
  if (!materials || !materials->data || (materials->texel_bytes != 16 && materials->texel_bytes != 4) || !sky ||
      !sky->data || sky->texel_bytes != 16)
    return NOVA_HIP_E_INVALID_ARGUMENT;
  if (num_lights > 0 && !lights) return NOVA_HIP_E_INVALID_ARGUMENT;
  const Alloc* am = find_alloc(tri_meta);
  if (!am) return NOVA_HIP_E_UNKNOWN_BUFFER;
  if (num_lights > 0) {
    const Alloc* al = find_alloc(lights);
    if (!al) return NOVA_HIP_E_UNKNOWN_BUFFER;
    if (al->bytes < static_cast<size_t>(num_lights) * sizeof(nova_area_light)) return NOVA_HIP_E_INVALID_ARGUMENT;
  }
  PackedScene* scene = nullptr;
  rc = get_scene(bvh, triangles, &scene);
  if (rc) return rc;
  if (am->bytes < static_cast<size_t>(scene->n_tris) * sizeof(nova_triangle_meta)) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (g.opt_tile_rank >= g.opt_tile_ranks) return NOVA_HIP_E_INVALID_ARGUMENT;

  const uint32_t tiles_x = (width + kTileSize - 1) / kTileSize, tiles_y = (height + kTileSize - 1) / kTileSize;
  const uint32_t n_tiles = tiles_x * tiles_y;
  const uint32_t rank = static_cast<uint32_t>(g.opt_tile_rank), ranks = static_cast<uint32_t>(g.opt_tile_ranks);
  const uint32_t n_my_tiles = n_tiles > rank ? (n_tiles - rank + ranks - 1) / ranks : 0;
  if (n_my_tiles == 0) return 0;
  uint32_t n_extra = 0;
  if (ranks > 1) {
    rc = ensure_tile_plan(width, height);
    if (rc) return rc;
    n_extra = g.plan.n_extra;
  }
  if ((static_cast<uint64_t>(n_my_tiles) * 4096ull + n_extra) * 3ull >= (1ull << 31)) return NOVA_HIP_E_TOO_LARGE;
  const uint32_t P = n_my_tiles * 4096u + n_extra;
  rc = ensure_frame_buffers(P);
  if (rc) return rc;

  FrameArgs F;
  std::memset(&F, 0, sizeof(F));
  F.params = *params;
  F.time = time;
  F.width = width;
  F.height = height;
  F.tile_rank = rank;
  F.tile_ranks = ranks;
  F.tiles_x = tiles_x;
  F.n_my_tiles = n_my_tiles;
  F.extra_pixels = ranks > 1 ? g.plan.d_extra : nullptr;
  F.n_extra = n_extra;
  F.meta = tri_meta;
  F.lights = lights;
  F.num_lights = num_lights;
  F.materials = { static_cast<const float4*>(materials->data), static_cast<int>(materials->width),
                  static_cast<int>(materials->height), static_cast<int>(materials->layers),
                  materials->texel_bytes == 4 ? static_cast<const uchar4*>(materials->data) : nullptr, nullptr };
  F.sky = { static_cast<const float4*>(sky->data), static_cast<int>(sky->width), static_cast<int>(sky->height), 1,
            nullptr, nullptr };
  F.denoise = denoise_available ? 1u : 0u;
  F.out_color = static_cast<float4*>(temp_color1->data);
  F.out_albedo = denoise_available ? static_cast<float4*>(albedo_feature1->data) : nullptr;
  F.out_normal = denoise_available ? static_cast<float4*>(normal_feature1->data) : nullptr;
  F.stride = P;
  F.max_depth = static_cast<uint32_t>(g.opt_max_depth);
  F.n_active = g.d_queues;
  F.n_mis = g.d_queues + kMisWords;
  F.block_counters = g.fb.block_counters;
  F.light_tris = nullptr;
  if (g.opt_mis_pretest && params->path_tracing != 0 && num_lights > 0) {
    rc = ensure_light_tris(tri_meta, am, triangles, find_alloc(triangles), scene->n_tris);
    if (rc) return rc;
    F.light_tris = g.d_light_tris;
  }
  const bool path_tracing = params->path_tracing != 0;
  const bool nee = path_tracing && num_lights > 0;
  uint32_t last_it;
  if (!path_tracing) last_it = 0;
  else if (F.max_depth != 0) last_it = nee ? F.max_depth : F.max_depth - 1;
  else last_it = kMaxIterations - 1;
  // path counters and the queue heads of every iteration: one fill
  HIP_TRY(hipMemsetAsync(g.d_queues, 0, (kQueueWords + kHeadWords * (last_it + 1)) * sizeof(uint32_t), g.stream));
  const unsigned blocks = (P + 255) / 256;
  // primary rays = valid pixels of this rank's tiles
  for (uint32_t t = rank; t < n_tiles; t += ranks) {
    const uint32_t tx = t % tiles_x, ty = t / tiles_x;
    const uint32_t w = std::min(kTileSize, width - tx * kTileSize), hgt = std::min(kTileSize, height - ty * kTileSize);
    g.host_primary += static_cast<uint64_t>(w) * hgt;
  }
  g.host_primary += n_extra;
  {
    hipEvent_t e0 = nullptr;
    span_begin(&e0);
    hipLaunchKernelGGL(k_generate, dim3(blocks), dim3(256), 0, g.stream, F, planes_of(0), g.fb.ray_o[0], g.fb.ray_d[0]);
    hipEvent_t e1 = span_end();
    if (e0 && e1) g.ev_shade.push_back({ e0, e1 });
  }
  HIP_TRY(hipGetLastError());


  const bool unbounded = F.max_depth == 0 && path_tracing;
  uint32_t it = 0;
  for (; it <= last_it; it++) {
    static const uint32_t spec_last = env_u32("NOVA_HIP_SPECULATIVE_LAST", kSpeculativeLast);
    if (pend && unbounded && it > spec_last) break; // the rest when the lane comes round again
    rc = enqueue_iteration(*scene, F, P, it, nee);
    if (rc) return rc;
    // unbounded mode: poll the survivor count now and then (paths die by miss / Russian roulette)
    if (!pend && unbounded && it >= 3 && (it & 1u)) {
      HIP_TRY(hipMemcpyAsync(g.h_pinned, g.d_queues + it + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
      HIP_TRY(hipStreamSynchronize(g.stream));
      g.chain_tail = nullptr;
      if (*g.h_pinned == 0) break;
    }
  }
  if (pend) {
    pend->active = true;
    pend->F = F;
    pend->bvh = bvh;
    pend->triangles = triangles;
    pend->P = P;
    pend->next_it = it;
    pend->last_it = last_it;
    pend->nee = nee;
  }
  if (!sync) return 0;
  HIP_TRY(hipStreamSynchronize(g.stream));
  collect_events();
  if (std::getenv("NOVA_HIP_TRACE_QUEUES")) { // diagnostics: paths alive at the start of each iteration
    std::vector<uint32_t> na(std::min<uint32_t>(last_it + 2, 16));
    HIP_TRY(hipMemcpy(na.data(), g.d_queues, na.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "nova_hip: paths per iteration:");
    for (uint32_t v : na) std::fprintf(stderr, " %u", v);
    std::fprintf(stderr, "\n");
  }
  return 0;
}

int nova_hip_launch_raytrace(const nova_scene_params* params, uint32_t time, nova_hip_image* temp_color1,
                             uint32_t width, uint32_t height, const nova_triangle_data* triangles,
                             const nova_triangle_meta* tri_meta, const nova_bvh_node* bvh,
                             const nova_area_light* lights, uint32_t num_lights,
                             const nova_hip_image* materials, const nova_hip_image* sky,
                             uint32_t denoise_available, nova_hip_image* albedo_feature1,
                             nova_hip_image* normal_feature1) {
  int rc = require_ctx();
  if (rc) return rc;
  // Lanes in turn (as nova_hip_render_samples does inside one call): kernel_raytrace of this call runs on the next
  // lane's stream with that lane's path state, beside the latency-bound tail and the image kernels of the call before.
  // It renders into the lane's own planes; those are copied into the caller's images once everything enqueued before
  // this call — which may still read the images — is done (under a tile partition only this rank's pixels: kernel_raytrace
  // leaves the others of temp_color1 as they are).  With unbounded paths the copy and the image calls that follow are
  // kept until the lane comes round (Deferred).  One lane in blocking mode.
  static const int env_lanes = [] {
    const char* e = std::getenv("NOVA_HIP_CALL_LANES");
    const int v = e ? std::atoi(e) : 0;
    return v >= 1 && v <= kMaxLanes ? v : kDefaultLanes;
  }();
  const bool images_ok = float_image_ok(temp_color1, width, height) &&
                         (!denoise_available || (float_image_ok(albedo_feature1, width, height) &&
                                                 float_image_ok(normal_feature1, width, height)));
  const int want = g.opt_call_lanes >= 1 ? static_cast<int>(g.opt_call_lanes) : env_lanes;
  const bool lanes_ok = g.opt_async && images_ok && params && params->path_tracing && g.opt_tile_rank < g.opt_tile_ranks;
  const int L = lanes_ok ? want : 1;
  const int n_planes = denoise_available ? 3 : 1;
  if (!(L > 1 && g.opt_max_depth == 0) && !g_deferred.empty()) { // kept image work belongs in front of this launch
    g.unsettled = true;
    rc = settle();
    if (rc) return rc;
  }
  if (L > 1) {
    const bool busy = g.unsettled;
    const bool unbounded = g.opt_max_depth == 0;
    const size_t plane = static_cast<size_t>(width) * height * 16;
    rc = ensure_call_lanes(L, plane, denoise_available != 0);
    if (rc) return rc;
    int prev = g.cur, lane = (g.cur + 1) % L;
    if (unbounded) {
      // the sample that used the next lane last must be whole, and its image work enqueued, before the lane is reused
      g_flush_deferred = &flush_deferred;
      if (g_deferred.empty()) g_issue_lane = g.cur;
      lane = (g_issue_lane + 1) % L;
      if (g_sample_seq >= static_cast<uint64_t>(L)) {
        rc = flush_deferred(g_sample_seq + 1 - static_cast<uint64_t>(L));
        if (rc) {
          g.unsettled = true;
          (void) settle();
          return rc;
        }
      }
      prev = g.cur;
    } else {
      HIP_TRY(hipEventRecord(g.lane_mark[prev], g.stream));
    }
    switch_lane(lane);
    // earlier calls still in flight: this sample will share the GPU with them.  A caller that reads the frame after every
    // sample (Raytracer::raytrace does) finds the device idle here, and its sample gets the whole of every CU.
    g.samples_in_flight = busy ? L : 1;
    nova_hip_image color = *temp_color1, albedo, normal;
    color.data = g.call_stage[g.cur][0];
    if (denoise_available) {
      albedo = *albedo_feature1; albedo.data = g.call_stage[g.cur][1];
      normal = *normal_feature1; normal.data = g.call_stage[g.cur][2];
    }
    rc = raytrace_impl(params, time, &color, width, height, triangles, tri_meta, bvh, lights, num_lights, materials, sky,
                       denoise_available, denoise_available ? &albedo : nullptr, denoise_available ? &normal : nullptr, false,
                       unbounded ? &g_pending[lane] : nullptr);
    g.samples_in_flight = 1;
    if (rc == 0) rc = size_other_lanes(L);
    float4* const dst[3] = { static_cast<float4*>(temp_color1->data),
                             denoise_available ? static_cast<float4*>(albedo_feature1->data) : nullptr,
                             denoise_available ? static_cast<float4*>(normal_feature1->data) : nullptr };
    if (rc == 0 && unbounded) {
      // the copy into the caller's images, and every image call that follows, waits for the lane to come round (Deferred)
      g_issue_lane = lane;
      g_deferred.push_back({ ++g_sample_seq, lane, [=]() { return copy_lane_planes(dst, n_planes, width, height); } });
    } else if (rc == 0) {
      hipError_t e = hipStreamWaitEvent(g.stream, g.lane_mark[prev], 0);
      rc = e != hipSuccess ? static_cast<int>(e) : copy_lane_planes(dst, n_planes, width, height);
    }
    g.unsettled = true;
    if (rc) {
      (void) settle();
      return rc;
    }
    return unbounded ? 0 : launched();
  }
  rc = raytrace_impl(params, time, temp_color1, width, height, triangles, tri_meta, bvh, lights, num_lights,
                     materials, sky, denoise_available, albedo_feature1, normal_feature1, false);
  g.unsettled = true;
  if (rc) {
    (void) settle();
    return rc;
  }
  return launched();
}

// NOVA_HIP_OPT_CLAMP_TO_EDGE for an image created with NOVA_ADDRESS_CLAMP
static bool clamp_to_edge(const nova_hip_image* im) { return g.opt_clamp_to_edge != 0 && im->address_mode == NOVA_ADDRESS_CLAMP; }

// ---- kernel_accumulate -------------------------------------------------------------------------
static int accumulate_impl(int32_t sample_index, uint32_t denoise_available,
                           const nova_hip_image* temp_color1, const nova_hip_image* albedo_feature1,
                           const nova_hip_image* normal_feature1, const nova_hip_image* prev_color,
                           const nova_hip_image* prev_albedo_feature,
                           const nova_hip_image* prev_normal_feature, nova_hip_image* temp_color2,
                           nova_hip_image* albedo_feature2, nova_hip_image* normal_feature2,
                           uint32_t width, uint32_t height, bool sync, bool check_only = false) {
  int rc = require_ctx();
  if (rc) return rc;
  if (width == 0 || height == 0 || sample_index < 0) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (!float_image_ok(temp_color1, width, height) || !float_image_ok(temp_color2, width, height) ||
      !float_image_ok(prev_color, width, height))
    return NOVA_HIP_E_INVALID_ARGUMENT;
  if (denoise_available) {
    const nova_hip_image* need[] = { albedo_feature1, normal_feature1, prev_albedo_feature,
                                     prev_normal_feature, albedo_feature2, normal_feature2 };
    for (const nova_hip_image* im : need)
      if (!float_image_ok(im, width, height)) return NOVA_HIP_E_INVALID_ARGUMENT;
  }
  AccumulateArgs A;
  A.sample_index = sample_index;
  A.denoise = denoise_available ? 1u : 0u;
  auto ptr = [](const nova_hip_image* im) { return im ? static_cast<float4*>(im->data) : nullptr; };
  A.color1 = ptr(temp_color1);
  A.albedo1 = ptr(albedo_feature1);
  A.normal1 = ptr(normal_feature1);
  A.prev_color = ptr(prev_color);
  A.prev_albedo = ptr(prev_albedo_feature);
  A.prev_normal = ptr(prev_normal_feature);
  A.color2 = ptr(temp_color2);
  A.albedo2 = ptr(albedo_feature2);
  A.normal2 = ptr(normal_feature2);
  A.width = width;
  A.height = height;
  A.mask = nullptr;
  A.edge = clamp_to_edge(temp_color1);
  if (g.opt_tile_ranks > 1) { // this rank's pixels only (its tiles + their halo)
    if (g.opt_tile_rank >= g.opt_tile_ranks) return NOVA_HIP_E_INVALID_ARGUMENT;
    rc = ensure_tile_plan(width, height);
    if (rc) return rc;
    A.mask = g.plan.d_mask;
  }
  if (check_only) return 0;
  const unsigned blocks = std::min<unsigned>((width * height + 255) / 256, static_cast<unsigned>(g.num_cus) * 8u);
  g.chain_tail = nullptr; // a later timing span must not start before this kernel
  hipLaunchKernelGGL(k_accumulate, dim3(blocks), dim3(256), 0, g.stream, A);
  HIP_TRY(hipGetLastError());
  if (sync) HIP_TRY(hipStreamSynchronize(g.stream));
  return 0;
}

int nova_hip_launch_accumulate(int32_t sample_index, uint32_t denoise_available,
                               const nova_hip_image* temp_color1, const nova_hip_image* albedo_feature1,
                               const nova_hip_image* normal_feature1, const nova_hip_image* prev_color,
                               const nova_hip_image* prev_albedo_feature,
                               const nova_hip_image* prev_normal_feature, nova_hip_image* temp_color2,
                               nova_hip_image* albedo_feature2, nova_hip_image* normal_feature2,
                               uint32_t width, uint32_t height) {
  if (!g_deferred.empty()) { // behind a sample that is not whole yet: the arguments are checked now
    const int rc = accumulate_impl(sample_index, denoise_available, temp_color1, albedo_feature1, normal_feature1, prev_color,
                                   prev_albedo_feature, prev_normal_feature, temp_color2, albedo_feature2, normal_feature2,
                                   width, height, false, true);
    if (rc) return rc;
  }
  return image_work([=]() {
    return accumulate_impl(sample_index, denoise_available, temp_color1, albedo_feature1, normal_feature1, prev_color,
                           prev_albedo_feature, prev_normal_feature, temp_color2, albedo_feature2, normal_feature2,
                           width, height, false);
  });
}

// ---- kernel_post_process -----------------------------------------------------------------------
int nova_hip_launch_post_process(const nova_scene_params* params, const nova_hip_image* temp_color2,
                                 nova_hip_image* pixels, uint32_t width, uint32_t height) {
  int rc = require_ctx();
  if (rc) return rc;
  if (!params || width == 0 || height == 0 || !float_image_ok(temp_color2, width, height)) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (!pixels || !pixels->data || pixels->texel_bytes != 4 || pixels->width != width || pixels->height != height)
    return NOVA_HIP_E_INVALID_ARGUMENT;
  PostArgs A;
  A.color2 = { static_cast<const float4*>(temp_color2->data), static_cast<int>(width), static_cast<int>(height), 1,
               nullptr, nullptr, clamp_to_edge(temp_color2) };
  A.pixels = static_cast<uchar4*>(pixels->data);
  A.width = width;
  A.height = height;
  A.exposure = params->exposure;
  A.anti_aliasing = params->anti_aliasing ? 1u : 0u;
  A.mask = nullptr;
  if (g.opt_tile_ranks > 1) {
    // FXAA reads up to ~26 pixels along an edge (kernels/anti_aliasing.hpp:99-116), i.e. other ranks'
    // pixels: gather the float means (nova_hip_tiles_pack / _unpack) and post-process the whole frame
    // on one rank with NOVA_HIP_OPT_TILE_RANKS = 1 instead
    if (A.anti_aliasing) return NOVA_HIP_E_INVALID_ARGUMENT;
    if (g.opt_tile_rank >= g.opt_tile_ranks) return NOVA_HIP_E_INVALID_ARGUMENT;
    rc = ensure_tile_plan(width, height);
    if (rc) return rc;
    A.mask = g.plan.d_mask;
  }
  const unsigned n = width * height;
  const unsigned blocks = std::min<unsigned>((n + 255) / 256, static_cast<unsigned>(g.num_cus) * 8u);
  return image_work([=]() -> int {
    hipLaunchKernelGGL(k_post_process, dim3(blocks), dim3(256), 0, g.stream, A);
    HIP_TRY(hipGetLastError());
    return 0;
  });
}

// ---- several samples in one call ---------------------------------------------------------------
static int ensure_lanes(int lanes, size_t plane_bytes, bool features) {
  if (!g.acc_done[0])
    for (auto& e : g.acc_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  while (g.lanes_ready < lanes) { // lane numbers are handed out in order; the current one may be any of them
    Lane& o = g.lanes[g.lanes_ready];
    o = Lane {};
    hipError_t e = hipStreamCreateWithFlags(&o.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&o.d_queues, (kQueueWords + kHeadWords * (kMaxIterations + 2)) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc(&o.h_pinned, 64, hipHostMallocDefault);
    if (e != hipSuccess) { // nothing of a half-made lane is kept: a retry starts from Lane {}
      if (o.stream) (void) hipStreamDestroy(o.stream);
      (void) hipFree(o.d_queues);
      o = Lane {};
      return static_cast<int>(e);
    }
    o.d_heads = o.d_queues + kQueueWords;
    g.lanes_ready++;
  }
  if (plane_bytes > g.stage_bytes) {
    for (auto& lane : g.stage)
      for (auto& p : lane) { (void) hipFree(p); p = nullptr; }
    g.stage_bytes = plane_bytes;
  }
  if (plane_bytes == 0) return 0; // lanes only (ensure_call_lanes)
  for (int l = 0; l + 1 < lanes; l++)
    for (int k = 0; k < (features ? 3 : 1); k++)
      if (!g.stage[l][k]) HIP_TRY(hipMalloc(&g.stage[l][k], g.stage_bytes));
  return 0;
}

// Raytracer::raytrace (src/core/raytracer.cpp:169-262) called num_samples times, without what only
// the host needs between samples: every kernel of every sample is enqueued and the host waits once,
// at the end.  prev / temp2 ping-pong by pointer (sample k reads the plane sample k-1 wrote) and are
// made equal by one copy at the end instead of one per sample; kernel_post_process runs once, on the
// final mean.  Consecutive samples run on two lanes (own stream, path state and ray queues each):
// kernel_raytrace of sample k+1 does not depend on sample k, so it fills the GPU while sample k's
// last traversal launches — a few thousand rays each, one latency chain per launch — drain.  Only
// kernel_accumulate is ordered across lanes (an event per sample).  The lane that does not render
// the last sample writes its radiance to a library-owned plane instead of temp_color1.
// Same kernels, same order per pixel, so every image ends with the bytes the four-call schedule
// leaves (tests/test_gpu_render_samples.py).  NOVA_HIP_LANES=1 keeps one lane (A/B).
int nova_hip_render_samples(const nova_scene_params* params, const uint32_t* times, uint32_t num_samples,
                            int32_t first_sample_index, const nova_hip_frame_images* images,
                            uint32_t width, uint32_t height, const nova_triangle_data* triangles,
                            const nova_triangle_meta* tri_meta, const nova_bvh_node* bvh,
                            const nova_area_light* lights, uint32_t num_lights,
                            const nova_hip_image* materials, const nova_hip_image* sky,
                            uint32_t denoise_available) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (!params || !times || !images || num_samples == 0 || first_sample_index < 0) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (static_cast<int64_t>(first_sample_index) + num_samples > INT32_MAX) return NOVA_HIP_E_TOO_LARGE;
  const nova_hip_frame_images& I = *images;
  if (!float_image_ok(I.temp_color1, width, height) || !float_image_ok(I.temp_color2, width, height) ||
      !float_image_ok(I.prev_color, width, height))
    return NOVA_HIP_E_INVALID_ARGUMENT;
  if (!I.pixels || !I.pixels->data || I.pixels->texel_bytes != 4 || I.pixels->width != width || I.pixels->height != height)
    return NOVA_HIP_E_INVALID_ARGUMENT;
  if (denoise_available) {
    const nova_hip_image* need[] = { I.albedo_feature1, I.normal_feature1, I.albedo_feature2, I.normal_feature2,
                                     I.prev_albedo_feature, I.prev_normal_feature };
    for (const nova_hip_image* im : need)
      if (!float_image_ok(im, width, height)) return NOVA_HIP_E_INVALID_ARGUMENT;
  }
  if (g.opt_tile_ranks > 1 && params->anti_aliasing) return NOVA_HIP_E_INVALID_ARGUMENT; // see nova_hip_launch_post_process
  const size_t plane = static_cast<size_t>(width) * height * 16;
  // lanes: NOVA_HIP_LANES=1..4 (A/B); default below
  static const int env_lanes = [] {
    const char* e = std::getenv("NOVA_HIP_LANES");
    const int v = e ? std::atoi(e) : 0;
    return v >= 1 && v <= kMaxLanes ? v : kDefaultLanes;
  }();
  const uint32_t L = std::min<uint32_t>(static_cast<uint32_t>(env_lanes), num_samples);
  // local descriptors whose data pointers swap; [0] = the "prev" role, [1] = the "temp2" role
  nova_hip_image color[2] = { *I.prev_color, *I.temp_color2 };
  nova_hip_image albedo[2], normal[2];
  // radiance of one sample: [0] = the caller's images (the lane of the LAST sample), [r] = the library's
  // planes of the lane that is r samples short of it
  nova_hip_image one_color[kMaxLanes], one_albedo[kMaxLanes], one_normal[kMaxLanes];
  for (auto& d : one_color) d = *I.temp_color1;
  if (denoise_available) {
    albedo[0] = *I.prev_albedo_feature; albedo[1] = *I.albedo_feature2;
    normal[0] = *I.prev_normal_feature; normal[1] = *I.normal_feature2;
    for (auto& d : one_albedo) d = *I.albedo_feature1;
    for (auto& d : one_normal) d = *I.normal_feature1;
  }
  if (L > 1) {
    rc = ensure_lanes(static_cast<int>(L), plane, denoise_available != 0);
    if (rc) return rc;
    if (g.cur >= static_cast<int>(L)) switch_lane(0); // lanes 0..L-1 in turn: one plane set per lane
    for (uint32_t r = 1; r < L; r++) {
      one_color[r].data = g.stage[r - 1][0];
      if (denoise_available) {
        one_albedo[r].data = g.stage[r - 1][1];
        one_normal[r].data = g.stage[r - 1][2];
      }
      if (g.opt_tile_ranks > 1) {
        // kernel_raytrace leaves the pixels of other ranks' tiles as they are and kernel_accumulate reads
        // them: the library's planes start as copies of the caller's, as if every sample had used those
        HIP_TRY(hipMemcpyAsync(one_color[r].data, I.temp_color1->data, plane, hipMemcpyDeviceToDevice, g.stream));
        if (denoise_available) {
          HIP_TRY(hipMemcpyAsync(one_albedo[r].data, I.albedo_feature1->data, plane, hipMemcpyDeviceToDevice, g.stream));
          HIP_TRY(hipMemcpyAsync(one_normal[r].data, I.normal_feature1->data, plane, hipMemcpyDeviceToDevice, g.stream));
        }
      }
    }
    if (g.opt_tile_ranks > 1) HIP_TRY(hipStreamSynchronize(g.stream)); // other lanes may write them at once
  }
  auto wait_all = [&]() {
    (void) hipStreamSynchronize(g.stream);
    for (int l = 0; l < g.lanes_ready; l++)
      if (l != g.cur) (void) hipStreamSynchronize(g.lanes[l].stream);
  };
  int in = 0;
  const int lane0 = g.cur;
  auto lane_of = [&](uint32_t k) { return static_cast<int>((static_cast<uint32_t>(lane0) + k) % L); };
  // kernel_raytrace of sample k on the current lane; the planes are the same for every sample of a lane
  auto start_sample = [&](uint32_t k) -> int {
    const uint32_t r = (num_samples - 1 - k) % L;
    return raytrace_impl(params, times[k], &one_color[r], width, height, triangles, tri_meta, bvh, lights, num_lights,
                         materials, sky, denoise_available, denoise_available ? &one_albedo[r] : nullptr,
                         denoise_available ? &one_normal[r] : nullptr, false, &g_pending[g.cur]);
  };
  // what is left of sample k (long paths of the unbounded mode), then its kernel_accumulate; called in
  // the order of k, on sample k's lane
  auto finish_sample = [&](uint32_t k) -> int {
    const uint32_t r = (num_samples - 1 - k) % L;
    int e = raytrace_finish(g_pending[g.cur]);
    if (e) return e;
    if (L > 1 && k > 0) { // the running mean comes from the previous lane's kernel_accumulate
      hipError_t he = hipStreamWaitEvent(g.stream, g.acc_done[(k - 1) & 1u], 0);
      if (he != hipSuccess) return static_cast<int>(he);
    }
    e = accumulate_impl(first_sample_index + static_cast<int32_t>(k), denoise_available, &one_color[r],
                        denoise_available ? &one_albedo[r] : nullptr, denoise_available ? &one_normal[r] : nullptr,
                        &color[in], denoise_available ? &albedo[in] : nullptr, denoise_available ? &normal[in] : nullptr,
                        &color[in ^ 1], denoise_available ? &albedo[in ^ 1] : nullptr,
                        denoise_available ? &normal[in ^ 1] : nullptr, width, height, false);
    if (e) return e;
    in ^= 1;
    if (L > 1 && k + 1 < num_samples) {
      hipError_t he = hipEventRecord(g.acc_done[k & 1u], g.stream);
      if (he != hipSuccess) return static_cast<int>(he);
    }
    return 0;
  };
  g.samples_in_flight = static_cast<int>(L);
  // lane by lane: finish the sample that was started there L samples ago, start the next one
  for (uint32_t k = 0; k < num_samples && rc == 0; k++) {
    switch_lane(lane_of(k));
    if (k >= L) rc = finish_sample(k - L);
    if (rc == 0) rc = start_sample(k);
  }
  for (uint32_t k = num_samples > L ? num_samples - L : 0; k < num_samples && rc == 0; k++) {
    switch_lane(lane_of(k));
    rc = finish_sample(k);
  }
  g.samples_in_flight = 1;
  if (rc) {
    wait_all();
    for (auto& pd : g_pending) pd.active = false;
    collect_events();
    return rc;
  }
  // the current lane holds the last sample; color[in] the final mean, whose pair plane gets a copy
  // (raytracer.cpp:202-210)
  hipError_t e = hipMemcpyAsync(color[in ^ 1].data, color[in].data, plane, hipMemcpyDeviceToDevice, g.stream);
  if (e == hipSuccess && denoise_available) {
    e = hipMemcpyAsync(albedo[in ^ 1].data, albedo[in].data, plane, hipMemcpyDeviceToDevice, g.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(normal[in ^ 1].data, normal[in].data, plane, hipMemcpyDeviceToDevice, g.stream);
  }
  rc = e != hipSuccess ? static_cast<int>(e)
                       : nova_hip_launch_post_process(params, &color[in], I.pixels, width, height); // waits for this lane
  wait_all(); // the other lanes are done already: their last kernels were waited for by this one's
  collect_events();
  g.unsettled = false;
  return rc;
}

// ---- roofline denominators -----------------------------------------------------------------------
int nova_hip_measure_device(nova_hip_device_rates* out) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (!out) return NOVA_HIP_E_INVALID_ARGUMENT;
  const size_t n = (static_cast<size_t>(1) << 30) / sizeof(float4); // 1 GiB read + 1 GiB written per pass
  float4 *src = nullptr, *dst = nullptr;
  HIP_TRY(hipMalloc(&src, n * sizeof(float4)));
  hipError_t e = hipMalloc(&dst, n * sizeof(float4));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e == hipSuccess) e = hipMemsetAsync(src, 0, n * sizeof(float4), g.stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  float best_copy = 1e30f, best_fma = 1e30f;
  const int iterations = 4096;
  const unsigned fma_blocks = static_cast<unsigned>(g.num_cus) * 32u;
  for (int rep = 0; rep < 4 && e == hipSuccess; rep++) {
    (void) hipEventRecord(e0, g.stream);
    hipLaunchKernelGGL(k_measure_copy, dim3(static_cast<unsigned>(g.num_cus) * 32u), dim3(256), 0, g.stream, src, dst, n);
    (void) hipEventRecord(e1, g.stream);
    e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best_copy) best_copy = ms;
    if (e != hipSuccess) break;
    (void) hipEventRecord(e0, g.stream);
    hipLaunchKernelGGL(k_measure_fma, dim3(fma_blocks), dim3(256), 0, g.stream, reinterpret_cast<float*>(dst), iterations, 1.0f + rep);
    (void) hipEventRecord(e1, g.stream);
    e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best_fma) best_fma = ms;
  }
  if (e0) (void) hipEventDestroy(e0);
  if (e1) (void) hipEventDestroy(e1);
  (void) hipFree(src);
  (void) hipFree(dst);
  if (e != hipSuccess) return static_cast<int>(e);
  out->hbm_copy_gbs = 2.0 * static_cast<double>(n) * sizeof(float4) / (best_copy * 1e-3) / 1e9;
  out->fp32_fma_tflops = 2.0 * 8.0 * iterations * static_cast<double>(fma_blocks) * 256.0 / (best_fma * 1e-3) / 1e12;
  return 0;
}

// ---- denoise hand-off ---------------------------------------------------------------------------
// floats per plane slot: k_flatten3 / k_pack3 access the planes as float4, so every plane starts on 16 bytes
static size_t denoise_stride(size_t pixels) { return (3 * pixels + 3) & ~static_cast<size_t>(3); }

static int ensure_denoise_planes(size_t pixels) {
  if (g.denoise_pixels >= pixels) return 0;
  HIP_TRY(hipStreamSynchronize(g.stream));
  (void) hipFree(g.d_denoise);
  (void) hipHostFree(g.h_denoise);
  g.d_denoise = g.h_denoise = nullptr;
  g.denoise_pixels = 0;
  const size_t bytes = 4 * denoise_stride(pixels) * sizeof(float);
  HIP_TRY(hipMalloc(&g.d_denoise, bytes));
  hipError_t e = hipHostMalloc(&g.h_denoise, bytes, hipHostMallocDefault);
  if (e != hipSuccess) { (void) hipFree(g.d_denoise); g.d_denoise = nullptr; return static_cast<int>(e); }
  g.denoise_pixels = pixels;
  return 0;
}

int nova_hip_denoise_read(const nova_hip_image* color, const nova_hip_image* albedo, const nova_hip_image* normal,
                          uint32_t width, uint32_t height, nova_hip_denoise_planes* planes) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (!planes || width == 0 || height == 0 || !float_image_ok(color, width, height) ||
      !float_image_ok(albedo, width, height) || !float_image_ok(normal, width, height))
    return NOVA_HIP_E_INVALID_ARGUMENT;
  const size_t pixels = static_cast<size_t>(width) * height;
  if (pixels >= (1ull << 32)) return NOVA_HIP_E_TOO_LARGE;
  rc = ensure_denoise_planes(pixels);
  if (rc) return rc;
  const size_t stride = denoise_stride(g.denoise_pixels); // floats per plane slot (slots sized for the largest frame seen)
  const nova_hip_image* src[3] = { color, albedo, normal };
  const unsigned blocks = static_cast<unsigned>(std::min<size_t>((pixels / 4 + 255) / 256 + 1, static_cast<size_t>(g.num_cus) * 8));
  for (int k = 0; k < 3; k++)
    hipLaunchKernelGGL(k_flatten3, dim3(blocks), dim3(256), 0, g.stream, static_cast<const float4*>(src[k]->data),
                       g.d_denoise + k * stride, static_cast<uint32_t>(pixels));
  HIP_TRY(hipGetLastError());
  // one copy for the three planes when they are adjacent (the usual case: the slots fit the frame exactly)
  if (g.denoise_pixels == pixels) {
    HIP_TRY(hipMemcpyAsync(g.h_denoise, g.d_denoise, (2 * stride + 3 * pixels) * sizeof(float), hipMemcpyDeviceToHost, g.stream));
  } else {
    for (int k = 0; k < 3; k++)
      HIP_TRY(hipMemcpyAsync(g.h_denoise + k * stride, g.d_denoise + k * stride, pixels * 3 * sizeof(float),
                             hipMemcpyDeviceToHost, g.stream));
  }
  HIP_TRY(hipStreamSynchronize(g.stream));
  planes->color = g.h_denoise;
  planes->albedo = g.h_denoise + stride;
  planes->normal = g.h_denoise + 2 * stride;
  planes->output = g.h_denoise + 3 * stride;
  return 0;
}

int nova_hip_denoise_write(nova_hip_image* color, uint32_t width, uint32_t height) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (width == 0 || height == 0 || !float_image_ok(color, width, height)) return NOVA_HIP_E_INVALID_ARGUMENT;
  const size_t pixels = static_cast<size_t>(width) * height;
  if (g.denoise_pixels < pixels || !g.h_denoise) return NOVA_HIP_E_INVALID_ARGUMENT; // nova_hip_denoise_read comes first
  const size_t stride = denoise_stride(g.denoise_pixels);
  HIP_TRY(hipMemcpyAsync(g.d_denoise + 3 * stride, g.h_denoise + 3 * stride, pixels * 3 * sizeof(float),
                         hipMemcpyHostToDevice, g.stream));
  const unsigned blocks = static_cast<unsigned>(std::min<size_t>((pixels / 4 + 255) / 256 + 1, static_cast<size_t>(g.num_cus) * 8));
  hipLaunchKernelGGL(k_pack3, dim3(blocks), dim3(256), 0, g.stream, g.d_denoise + 3 * stride,
                     static_cast<float4*>(color->data), static_cast<uint32_t>(pixels));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(g.stream));
  return 0;
}

// ---- tile stacks for the multi-GPU gather ------------------------------------------------------
uint64_t nova_hip_tiles_stack_bytes(uint32_t width, uint32_t height, uint32_t ranks, uint32_t texel_bytes) {
  if (width == 0 || height == 0 || ranks == 0) return 0;
  const uint64_t n_tiles = static_cast<uint64_t>((width + kTileSize - 1) / kTileSize) * ((height + kTileSize - 1) / kTileSize);
  return (n_tiles + ranks - 1) / ranks * (kTileSize * kTileSize) * texel_bytes;
}

static bool tile_image_ok(const nova_hip_image* image) {
  return image && image->data && image->layers == 1 && (image->texel_bytes == 16 || image->texel_bytes == 4);
}

static int tiles_pack_async(const nova_hip_image* image, uint32_t rank, uint32_t ranks, void* stack_dev) {
  const uint32_t tiles_x = (image->width + kTileSize - 1) / kTileSize, tiles_y = (image->height + kTileSize - 1) / kTileSize;
  const uint32_t n_tiles = tiles_x * tiles_y, per_rank = (n_tiles + ranks - 1) / ranks;
  const unsigned blocks = static_cast<unsigned>(std::min<size_t>(static_cast<size_t>(per_rank) * 16, static_cast<size_t>(g.num_cus) * 16));
  if (image->texel_bytes == 16)
    hipLaunchKernelGGL(k_tiles_pack<uint4>, dim3(blocks), dim3(256), 0, g.stream, static_cast<const uint4*>(image->data),
                       image->width, image->height, tiles_x, n_tiles, rank, ranks, per_rank, static_cast<uint4*>(stack_dev));
  else
    hipLaunchKernelGGL(k_tiles_pack<uint32_t>, dim3(blocks), dim3(256), 0, g.stream, static_cast<const uint32_t*>(image->data),
                       image->width, image->height, tiles_x, n_tiles, rank, ranks, per_rank, static_cast<uint32_t*>(stack_dev));
  return static_cast<int>(hipGetLastError());
}

static int tiles_unpack_async(nova_hip_image* image, uint32_t ranks, const void* stacks_dev) {
  const uint32_t tiles_x = (image->width + kTileSize - 1) / kTileSize, tiles_y = (image->height + kTileSize - 1) / kTileSize;
  const uint32_t n_tiles = tiles_x * tiles_y, per_rank = (n_tiles + ranks - 1) / ranks;
  const unsigned blocks = static_cast<unsigned>(std::min<size_t>(static_cast<size_t>(per_rank) * ranks * 16, static_cast<size_t>(g.num_cus) * 16));
  if (image->texel_bytes == 16)
    hipLaunchKernelGGL(k_tiles_unpack<uint4>, dim3(blocks), dim3(256), 0, g.stream, static_cast<uint4*>(image->data),
                       image->width, image->height, tiles_x, n_tiles, ranks, per_rank, static_cast<const uint4*>(stacks_dev));
  else
    hipLaunchKernelGGL(k_tiles_unpack<uint32_t>, dim3(blocks), dim3(256), 0, g.stream, static_cast<uint32_t*>(image->data),
                       image->width, image->height, tiles_x, n_tiles, ranks, per_rank, static_cast<const uint32_t*>(stacks_dev));
  return static_cast<int>(hipGetLastError());
}

int nova_hip_tiles_pack(const nova_hip_image* image, uint32_t rank, uint32_t ranks, void* stack_dev) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (!tile_image_ok(image) || !stack_dev || ranks == 0 || rank >= ranks) return NOVA_HIP_E_INVALID_ARGUMENT;
  rc = tiles_pack_async(image, rank, ranks, stack_dev);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(g.stream));
  return 0;
}

int nova_hip_tiles_unpack(nova_hip_image* image, uint32_t ranks, const void* stacks_dev) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (!tile_image_ok(image) || !stacks_dev || ranks == 0) return NOVA_HIP_E_INVALID_ARGUMENT;
  rc = tiles_unpack_async(image, ranks, stacks_dev);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(g.stream));
  return 0;
}

// ---- the gather itself: RCCL behind the C ABI ---------------------------------------------------
int nova_hip_tiles_unique_id(void* id_out) {
  if (!id_out) return NOVA_HIP_E_INVALID_ARGUMENT;
  int rc = require_ctx();
  if (rc) return rc;
  if (!load_rccl()) return NOVA_HIP_E_NO_RCCL;
  static_assert(sizeof(ncclUniqueId) == NOVA_HIP_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  RCCL_TRY(rccl().GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

static void drop_comm() {
  if (g.comm && g.comm_owned && rccl().ok) (void) rccl().CommDestroy(g.comm);
  g.comm = nullptr;
  g.comm_owned = false;
  g.comm_rank = g.comm_ranks = 0;
}

int nova_hip_tiles_comm_init(const void* unique_id, uint32_t rank, uint32_t ranks) {
  if (int rc_settle = settle()) return rc_settle;
  if (!unique_id || ranks == 0 || rank >= ranks) return NOVA_HIP_E_INVALID_ARGUMENT;
  int rc = require_ctx();
  if (rc) return rc;
  if (!load_rccl()) return NOVA_HIP_E_NO_RCCL;
  HIP_TRY(hipStreamSynchronize(g.stream));
  drop_comm();
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  HIP_TRY(hipSetDevice(g.device)); // the communicator is made on the library's device
  ncclComm_t c = nullptr;
  RCCL_TRY(rccl().CommInitRank(&c, static_cast<int>(ranks), id, static_cast<int>(rank)));
  g.comm = c;
  g.comm_owned = true;
  g.comm_rank = rank;
  g.comm_ranks = ranks;
  return 0;
}

int nova_hip_tiles_comm_adopt(void* nccl_comm) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (!nccl_comm) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (!load_rccl()) return NOVA_HIP_E_NO_RCCL;
  HIP_TRY(hipStreamSynchronize(g.stream));
  int n = 0, r = 0;
  RCCL_TRY(rccl().CommCount(static_cast<ncclComm_t>(nccl_comm), &n));
  RCCL_TRY(rccl().CommUserRank(static_cast<ncclComm_t>(nccl_comm), &r));
  if (n <= 0 || r < 0 || r >= n) return NOVA_HIP_E_INVALID_ARGUMENT;
  drop_comm();
  g.comm = static_cast<ncclComm_t>(nccl_comm);
  g.comm_owned = false; // the caller's: never destroyed here
  g.comm_rank = static_cast<uint32_t>(r);
  g.comm_ranks = static_cast<uint32_t>(n);
  return 0;
}

int nova_hip_tiles_comm_destroy(void) {
  if (int rc_settle = settle()) return rc_settle;
  if (!g.ready) return 0;
  (void) hipStreamSynchronize(g.stream);
  drop_comm();
  return 0;
}

int nova_hip_tiles_comm_info(nova_hip_comm_info* out) {
  if (!out) return NOVA_HIP_E_INVALID_ARGUMENT;
  int rc = require_ctx();
  if (rc) return rc;
  std::memset(out, 0, sizeof(*out));
  out->rank = g.comm_rank;
  out->ranks = g.comm_ranks;
  out->device = g.device;
  HIP_TRY(hipDeviceGetPCIBusId(out->pci_bus_id, static_cast<int>(sizeof(out->pci_bus_id)), g.device));
  if (load_rccl()) {
    int v = 0;
    if (rccl().GetVersion(&v) == ncclSuccess) out->rccl_version = v;
    std::snprintf(out->rccl_path, sizeof(out->rccl_path), "%s", rccl().path);
  }
  return 0;
}

// pack -> ncclAllGather -> unpack, all on the library's stream (nothing to order against another stream); one
// collective of equal-sized stacks.  Afterwards EVERY rank's image holds every rank's tiles: an all-gather hands the
// stacks to everyone anyway, and the unpack is one streaming pass (4K float4: 0.13 GB).
int nova_hip_tiles_allgather(nova_hip_image* image) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (!tile_image_ok(image)) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (!g.comm) return NOVA_HIP_E_NO_COMM;
  const uint32_t ranks = g.comm_ranks, rank = g.comm_rank;
  const size_t stack = nova_hip_tiles_stack_bytes(image->width, image->height, ranks, image->texel_bytes);
  const size_t need = stack * (static_cast<size_t>(ranks) + 1);
  if (need > g.gather_bytes) {
    HIP_TRY(hipStreamSynchronize(g.stream));
    (void) hipFree(g.d_gather);
    g.d_gather = nullptr;
    g.gather_bytes = 0;
    HIP_TRY(hipMalloc(&g.d_gather, need));
    g.gather_bytes = need;
  }
  char* send = static_cast<char*>(g.d_gather);
  char* recv = send + stack;
  rc = tiles_pack_async(image, rank, ranks, send);
  if (rc) return rc;
  RCCL_TRY(rccl().AllGather(send, recv, stack, ncclChar, g.comm, g.stream)); // rank r's stack lands at recv + r * stack
  rc = tiles_unpack_async(image, ranks, recv);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(g.stream));
  return 0;
}

// ---- ray batches -------------------------------------------------------------------------------
int nova_hip_trace_rays(const nova_triangle_data* triangles, const nova_bvh_node* bvh,
                        const nova_float4* rays_point_tmax, const nova_float4* rays_dir_flags,
                        nova_float4* hits, uint64_t num_rays) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (!triangles || !bvh || !rays_point_tmax || !rays_dir_flags || !hits) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (num_rays == 0) return 0;
  if (num_rays >= (1ull << 31)) return NOVA_HIP_E_TOO_LARGE;
  const Alloc* a0 = find_alloc(rays_point_tmax);
  const Alloc* a1 = find_alloc(rays_dir_flags);
  const Alloc* a2 = find_alloc(hits);
  if (!a0 || !a1 || !a2) return NOVA_HIP_E_UNKNOWN_BUFFER;
  if (a0->bytes < num_rays * 16 || a1->bytes < num_rays * 16 || a2->bytes < num_rays * 16)
    return NOVA_HIP_E_INVALID_ARGUMENT;
  PackedScene* scene = nullptr;
  rc = get_scene(bvh, triangles, &scene);
  if (rc) return rc;
  uint32_t* queue_heads = g.d_heads;
  HIP_TRY(hipMemsetAsync(queue_heads, 0, kHeadWords * sizeof(uint32_t), g.stream));
  TraverseArgs T;
  std::memset(&T, 0, sizeof(T));
  T.ray_o = reinterpret_cast<const float4*>(rays_point_tmax);
  T.ray_d = reinterpret_cast<const float4*>(rays_dir_flags);
  T.hits = reinterpret_cast<float4*>(hits);
  T.n_items_ptr = nullptr;
  T.n_items_imm = static_cast<uint32_t>(num_rays);
  T.kinds = 1;
  T.stride = 0;
  T.queue_head = queue_heads;
  // large batches over a scene that does not fit the caches: bin the rays first (NOVA_HIP_OPT_SORT_RAYS)
  const size_t scene_bytes = static_cast<size_t>(scene->n_wide) * sizeof(Node128) + static_cast<size_t>(scene->n_tris) * 48;
  const bool sort = scene->wide != nullptr && g.opt_traversal == 0 && g.opt_kernel == 0 &&
                    (g.opt_sort_rays == 2 || (g.opt_sort_rays == 0 && num_rays >= (1u << 20) && scene_bytes > kCacheResidentBytes));
  if (sort) {
    const uint32_t n = static_cast<uint32_t>(num_rays);
    const uint32_t n_keys = 1u << kRayKeyBits;
    if (g.ray_sort_capacity < num_rays) {
      HIP_TRY(hipStreamSynchronize(g.stream));
      (void) hipFree(g.d_ray_keys);
      (void) hipFree(g.d_ray_perm);
      g.d_ray_keys = g.d_ray_perm = nullptr;
      g.ray_sort_capacity = 0;
      HIP_TRY(hipMalloc(&g.d_ray_keys, num_rays * 4));
      HIP_TRY(hipMalloc(&g.d_ray_perm, num_rays * 4));
      if (!g.d_ray_hist) HIP_TRY(hipMalloc(&g.d_ray_hist, (2 * n_keys + 16) * 4));
      g.ray_sort_capacity = num_rays;
    }
    uint32_t* hist = g.d_ray_hist;
    uint32_t* cursor = g.d_ray_hist + n_keys;
    HIP_TRY(hipMemsetAsync(hist, 0, n_keys * 4, g.stream));
    float s[3];
    for (int a = 0; a < 3; a++) {
      const float ext = scene->root_top[a] - scene->root_bot[a];
      s[a] = ext > 0.0f ? 32.0f / ext : 0.0f;
    }
    hipEvent_t e0 = nullptr;
    span_begin(&e0);
    hipLaunchKernelGGL(k_ray_keys, dim3((n + 255) / 256), dim3(256), 0, g.stream, T.ray_o, T.ray_d, n, scene->root_bot[0],
                       scene->root_bot[1], scene->root_bot[2], s[0], s[1], s[2], g.d_ray_keys, hist);
    g.ray_scan.stream = g.stream;
    g.ray_scan.d_total = cursor + n_keys;
    rc = g.ray_scan.run(hist, n_keys, cursor);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ray_scatter, dim3((n + 255) / 256), dim3(256), 0, g.stream, g.d_ray_keys, n, cursor, g.d_ray_perm);
    HIP_TRY(hipGetLastError());
    hipEvent_t e1 = span_end();
    if (e0 && e1) g.ev_traverse.push_back({ e0, e1 }); // part of the batch's traversal time
    T.perm = g.d_ray_perm;
  }
  rc = launch_traverse(*scene, T, false);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(g.stream));
  collect_events();
  touch(hits);
  return 0;
}

// nova_hip_trace_rays for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc):
// same kernels, same arguments, but the queue heads and the stack-spill area live in the static `scratch` array, no
// timing events are made and the traversal layout must already be cached.  Calls that overlap on one
// GPU race on `scratch`.
int nova_hip_trace_rays_noalloc(const nova_triangle_data* triangles, const nova_bvh_node* bvh,
                                const nova_float4* rays_point_tmax, const nova_float4* rays_dir_flags,
                                nova_float4* hits, uint64_t num_rays) {
  if (int rc_settle = settle()) return rc_settle;
  if (!g.ready || !g.scratch_ptr) return NOVA_HIP_E_NO_DEVICE; // nova_hip_init makes the context (it allocates)
  if (!triangles || !bvh || !rays_point_tmax || !rays_dir_flags || !hits) return NOVA_HIP_E_INVALID_ARGUMENT;
  if (num_rays == 0) return 0;
  if (num_rays >= (1ull << 31)) return NOVA_HIP_E_TOO_LARGE;
  const Alloc* a0 = find_alloc(rays_point_tmax);
  const Alloc* a1 = find_alloc(rays_dir_flags);
  const Alloc* a2 = find_alloc(hits);
  const Alloc* ab = find_alloc(bvh);
  const Alloc* at = find_alloc(triangles);
  if (!a0 || !a1 || !a2 || !ab || !at) return NOVA_HIP_E_UNKNOWN_BUFFER;
  if (a0->bytes < num_rays * 16 || a1->bytes < num_rays * 16 || a2->bytes < num_rays * 16)
    return NOVA_HIP_E_INVALID_ARGUMENT;
  const PackedScene* scene = nullptr;
  for (const auto& s : g.scenes)
    if (s.bvh_ptr == bvh && s.tri_ptr == triangles && s.bvh_version == ab->version && s.tri_version == at->version)
      scene = &s;
  if (!scene) return NOVA_HIP_E_NOT_CACHED;
  uint32_t* const heads = g.scratch_ptr;
  // the array keeps the last call's contents: the heads must start at zero (the spill area is written
  // before it is read)
  HIP_TRY(hipMemsetAsync(heads, 0, kHeadWords * sizeof(uint32_t), g.stream));
  TraverseArgs T;
  std::memset(&T, 0, sizeof(T));
  T.ray_o = reinterpret_cast<const float4*>(rays_point_tmax);
  T.ray_d = reinterpret_cast<const float4*>(rays_dir_flags);
  T.hits = reinterpret_cast<float4*>(hits);
  T.n_items_ptr = nullptr;
  T.n_items_imm = static_cast<uint32_t>(num_rays);
  T.kinds = 1;
  T.stride = 0;
  T.queue_head = heads;
  uint32_t* const saved_spill = g.d_spill;
  const size_t saved_words = g.spill_words;
  g.d_spill = g.scratch_ptr + kHeadWords;
  g.spill_words = kNoAllocSpillWords;
  g.no_alloc = true;
  g.chain_tail = nullptr;
  int rc = launch_traverse(*scene, T, false);
  g.no_alloc = false;
  g.d_spill = saved_spill;
  g.spill_words = saved_words;
  const hipError_t e = hipStreamSynchronize(g.stream);
  g.chain_tail = nullptr;
  if (rc) return rc;
  if (e != hipSuccess) return static_cast<int>(e);
  touch(hits);
  return 0;
}

uint64_t nova_hip_bvh_max_nodes(uint64_t num_triangles) { return bvhb::max_nodes(num_triangles); }

int nova_hip_build_bvh_device(const float* vertices, uint64_t num_triangles, nova_bvh_node* bvh_out,
                              uint64_t bvh_capacity, uint64_t* num_nodes, nova_triangle_data* triangles_out,
                              uint32_t* order_out, nova_hip_build_stats* stats) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (!vertices || !bvh_out || !num_nodes || !triangles_out || !order_out || num_triangles == 0)
    return NOVA_HIP_E_INVALID_ARGUMENT;
  if (num_triangles > (1ull << 24)) return NOVA_HIP_E_TOO_LARGE;
  if (bvh_capacity < bvhb::max_nodes(num_triangles)) return NOVA_HIP_E_INVALID_ARGUMENT;
  const Alloc* av = find_alloc(vertices);
  const Alloc* ab = find_alloc(bvh_out);
  const Alloc* at = find_alloc(triangles_out);
  const Alloc* ao = find_alloc(order_out);
  if (!av || !ab || !at || !ao) return NOVA_HIP_E_UNKNOWN_BUFFER;
  if (av->bytes < num_triangles * 36 || ab->bytes < bvh_capacity * sizeof(nova_bvh_node) ||
      at->bytes < num_triangles * sizeof(nova_triangle_data) || ao->bytes < num_triangles * 4)
    return NOVA_HIP_E_INVALID_ARGUMENT;
  bvhb::Result res;
  bvhb::Tree tree;
  rc = bvhb::build(g.stream, vertices, static_cast<uint32_t>(num_triangles), bvh_out, triangles_out, order_out, &res,
                   &tree);
  if (rc) return rc;
  // the scene cache keys on pointer + version: new contents in these buffers
  touch(bvh_out);
  touch(triangles_out);
  touch(order_out);
  drop_scenes_using(bvh_out);
  drop_scenes_using(triangles_out);
  // the traversal layout from the builder's own tree, so that the first launch does not repack on the host
  if (!host_pack_only() && tree.depth <= 64) {
    packdev::Packed pd;
    const auto t0 = std::chrono::steady_clock::now();
    rc = packdev::pack(g.stream, tree, reinterpret_cast<const float4*>(triangles_out), &pd);
    if (rc < 0 || rc > 1) return rc;
    if (rc == 0) {
      PackedScene s;
      adopt_packed(pd, bvh_out, triangles_out, find_alloc(bvh_out), at, s);
      g.scenes.push_back(s);
      res.pack_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  *num_nodes = res.n_nodes;
  if (stats) {
    stats->device_ms = res.device_ms;
    stats->levels = res.levels;
    stats->peak_bin_bytes = res.peak_bin_bytes;
    stats->pack_ms = res.pack_ms;
  }
  return 0;
}

int nova_hip_build_bvh(const float* vertices, uint64_t num_triangles, nova_bvh_node* bvh_out,
                       uint64_t bvh_capacity, uint64_t* num_nodes, nova_triangle_data* triangles_out,
                       uint32_t* order_out, nova_hip_build_stats* stats) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (!vertices || !bvh_out || !num_nodes || !triangles_out || !order_out || num_triangles == 0)
    return NOVA_HIP_E_INVALID_ARGUMENT;
  // FlatBVHNode keeps triangle offsets and child indices in floats (src/kernel_types/bvh_node.hpp:13-16)
  if (num_triangles > (1ull << 24)) return NOVA_HIP_E_TOO_LARGE;
  if (bvh_capacity < bvhb::max_nodes(num_triangles)) return NOVA_HIP_E_INVALID_ARGUMENT;
  const uint32_t n = static_cast<uint32_t>(num_triangles);
  bvhb::DeviceBuffer d_verts, d_bvh, d_tris, d_order;
  HIP_TRY(d_verts.reserve(static_cast<size_t>(n) * 36));
  HIP_TRY(d_bvh.reserve(bvhb::max_nodes(n) * sizeof(nova_bvh_node)));
  HIP_TRY(d_tris.reserve(static_cast<size_t>(n) * sizeof(nova_triangle_data)));
  HIP_TRY(d_order.reserve(static_cast<size_t>(n) * 4));
  HIP_TRY(hipMemcpyAsync(d_verts.p, vertices, static_cast<size_t>(n) * 36, hipMemcpyHostToDevice, g.stream));
  bvhb::Result res;
  rc = bvhb::build(g.stream, d_verts.as<float>(), n, d_bvh.as<nova_bvh_node>(), d_tris.as<nova_triangle_data>(),
                   d_order.as<uint32_t>(), &res);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(bvh_out, d_bvh.p, static_cast<size_t>(res.n_nodes) * sizeof(nova_bvh_node),
                         hipMemcpyDeviceToHost, g.stream));
  HIP_TRY(hipMemcpyAsync(triangles_out, d_tris.p, static_cast<size_t>(n) * sizeof(nova_triangle_data),
                         hipMemcpyDeviceToHost, g.stream));
  HIP_TRY(hipMemcpyAsync(order_out, d_order.p, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost, g.stream));
  HIP_TRY(hipStreamSynchronize(g.stream));
  *num_nodes = res.n_nodes;
  if (stats) {
    stats->device_ms = res.device_ms;
    stats->levels = res.levels;
    stats->peak_bin_bytes = res.peak_bin_bytes;
    stats->pack_ms = res.pack_ms;
  }
  return 0;
}

int nova_hip_validate_scene_pack(const nova_bvh_node* bvh, const nova_triangle_data* triangles,
                                 uint64_t* different_words) {
  if (int rc_settle = settle()) return rc_settle;
  int rc = require_ctx();
  if (rc) return rc;
  if (!bvh || !triangles || !different_words) return NOVA_HIP_E_INVALID_ARGUMENT;
  const Alloc* ab = find_alloc(bvh);
  const Alloc* at = find_alloc(triangles);
  if (!ab || !at) return NOVA_HIP_E_UNKNOWN_BUFFER;
  const PackedScene* have = nullptr;
  for (const auto& s : g.scenes)
    if (s.bvh_ptr == bvh && s.tri_ptr == triangles && s.bvh_version == ab->version && s.tri_version == at->version)
      have = &s;
  if (!have) return NOVA_HIP_E_INVALID_ARGUMENT; // nothing cached for these buffers
  PackedScene ref;
  rc = pack_scene_host(bvh, triangles, ab, at, ref);
  if (rc) return rc;
  uint64_t diff = 0;
  diff += have->n_packed != ref.n_packed;
  diff += have->n_wide != ref.n_wide;
  diff += have->stack_need != ref.stack_need;
  diff += have->stack_need_wide != ref.stack_need_wide;
  diff += have->n_tris != ref.n_tris;
  diff += __builtin_bit_cast(uint32_t, have->radius) != __builtin_bit_cast(uint32_t, ref.radius);
  diff += (have->wide == nullptr) != (ref.wide == nullptr);
  auto compare = [&](const void* a, const void* b, size_t bytes) -> int {
    std::vector<uint32_t> ha(bytes / 4), hb(bytes / 4);
    HIP_TRY(hipMemcpy(ha.data(), a, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hb.data(), b, bytes, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ha.size(); i++) diff += ha[i] != hb[i];
    return 0;
  };
  if (diff == 0) {
    rc = compare(have->nodes, ref.nodes, static_cast<size_t>(ref.n_packed) * sizeof(Node64));
    if (rc == 0 && ref.wide) rc = compare(have->wide, ref.wide, static_cast<size_t>(ref.n_wide) * sizeof(Node128));
  }
  free_scene(ref);
  if (rc) return rc;
  *different_words = diff;
  return 0;
}

}
