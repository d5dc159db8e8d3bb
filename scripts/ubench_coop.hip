// Microbenchmark for DESIGN.md §7 "cooperative node loads": a wave whose `active` lanes each need a different
// random 128-byte record fetches them either (a) one lane per record, eight 16-byte loads each (what k_traverse_p
// does for a Node128), or (b) cooperatively: eight lanes fetch the eight pieces of one record (a fully coalesced
// 128-byte line per eight lanes), park them in LDS, and every ray-lane reads its record back from LDS.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_coop.hip -o scripts/ubench_coop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <bool COOP, int EXTRA_LDS_KB>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ table, uint32_t n_records, int steps, int active, uint32_t* out) {
  __shared__ uint4 stage[4][8][64];                 // [wave][piece][ray lane], rotated by piece
  __shared__ uint32_t pad[EXTRA_LDS_KB * 256 + 1];  // occupancy control
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t cur = (gid * 2654435761u) % n_records;
  uint32_t acc = 0;
  if (threadIdx.x == 0) pad[0] = 0;
  const bool ray = static_cast<int>(lane) < active;
  for (int s = 0; s < steps; s++) {
    uint4 v[8];
    if (COOP) {
      const int rounds = (active + 7) / 8;
      for (int r = 0; r < rounds; r++) {
        const uint32_t src = 8u * r + (lane >> 3), piece = lane & 7u;
        const uint32_t idx = __shfl(cur, src);
        if (static_cast<int>(src) < active) {
          const uint4 x = table[static_cast<size_t>(idx) * 8u + piece];
          stage[wave][piece][(src + piece) & 63u] = x;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (ray) {
#pragma unroll
        for (int p = 0; p < 8; p++) v[p] = stage[wave][p][(lane + p) & 63u];
      }
      __builtin_amdgcn_wave_barrier();
    } else if (ray) {
      const uint4* rec = table + static_cast<size_t>(cur) * 8u;
#pragma unroll
      for (int p = 0; p < 8; p++) v[p] = rec[p];
    }
    if (ray) {
      uint32_t x = 0;
#pragma unroll
      for (int p = 0; p < 8; p++) x ^= v[p].x + v[p].y + v[p].z + v[p].w;
      acc += x;
      cur = v[0].x % n_records;
    }
  }
  if (acc == 0x12345678u) out[gid] = acc + pad[0];
}

template <bool COOP, int KB>
float run(int grid, const uint4* t, uint32_t n, int steps, int active, uint32_t* out, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<COOP, KB>), dim3(grid), dim3(256), 0, 0, t, n, steps, active, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, 8u << 20));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (size_t mb : { size_t(8), size_t(48), size_t(600) }) {
    const uint32_t n = static_cast<uint32_t>((mb << 20) / 128);
    std::vector<uint4> host(static_cast<size_t>(n) * 8);
    uint32_t st = 12345u;
    for (uint32_t r = 0; r < n; r++) {
      st ^= st << 13; st ^= st >> 17; st ^= st << 5;
      for (int p = 0; p < 8; p++) host[static_cast<size_t>(r) * 8 + p] = make_uint4(st, st * 3u + p, st * 5u, st * 7u);
    }
    uint4* d;
    CHECK(hipMalloc(&d, host.size() * sizeof(uint4)));
    CHECK(hipMemcpy(d, host.data(), host.size() * sizeof(uint4), hipMemcpyHostToDevice));
    const int steps = 300;
    for (int active : { 16, 32, 48, 64 }) {
      // blocks per CU limited by LDS: 32 KB of stage per block -> 5 blocks (20 waves); + 21 KB pad -> 3 blocks (12 waves)
      const float a = run<false, 0>(cus * 5, d, n, steps, active, d_out, e0, e1);
      const float b = run<true, 0>(cus * 5, d, n, steps, active, d_out, e0, e1);
      const float c = run<true, 21>(cus * 3, d, n, steps, active, d_out, e0, e1);
      const double rec5 = static_cast<double>(cus) * 5 * 4 * steps * active, rec3 = static_cast<double>(cus) * 3 * 4 * steps * active;
      std::printf("table %4zu MB active %2d: per-lane 8 loads %6.2f G rec/s | coop 20 waves/CU %6.2f G rec/s | coop 12 waves/CU %6.2f G rec/s\n",
                  mb, active, rec5 / a / 1e6, rec5 / b / 1e6, rec3 / c / 1e6);
    }
    CHECK(hipFree(d));
  }
  return 0;
}
