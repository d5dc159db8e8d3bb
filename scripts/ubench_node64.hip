// Microbenchmark for DESIGN.md §7 "a 64-byte quantised node": what a node step of k_traverse_p would cost on the
// memory side with four 16-byte loads per lane from a 64-byte record instead of eight from a 128-byte one, for the
// SAME number of records (the tree does not change, only the bytes per node), with `valu` fused multiply-adds per
// lane and step standing in for the slab tests (and, for the small record, the extra decode work).
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_node64.hip -o scripts/ubench_node64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int PIECES, int VALU>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ table, uint32_t n_records, int steps, int active, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t cur = (gid * 2654435761u) % n_records;
  float f0 = 1.0f, f1 = 2.0f, f2 = 3.0f, f3 = 4.0f;
  uint32_t acc = 0;
  const bool ray = static_cast<int>(lane) < active;
  for (int s = 0; s < steps; s++) {
    if (ray) {
      uint4 v[PIECES];
      const uint4* rec = table + static_cast<size_t>(cur) * PIECES;
#pragma unroll
      for (int p = 0; p < PIECES; p++) v[p] = rec[p];
      uint32_t x = 0;
#pragma unroll
      for (int p = 0; p < PIECES; p++) x ^= v[p].x + v[p].y + v[p].z + v[p].w;
      const float m = __uint_as_float((x & 0x007FFFFFu) | 0x3F000000u);
#pragma unroll
      for (int q = 0; q < VALU / 4; q++) {
        f0 = __builtin_fmaf(f0, m, 0.25f);
        f1 = __builtin_fmaf(f1, m, 0.25f);
        f2 = __builtin_fmaf(f2, m, 0.25f);
        f3 = __builtin_fmaf(f3, m, 0.25f);
      }
      acc += x;
      cur = v[0].x % n_records;
    }
  }
  if (acc == 0x12345678u || f0 + f1 + f2 + f3 == 12345.0f) out[gid] = acc;
}

template <int PIECES, int VALU>
double run(int grid, const uint4* t, uint32_t n, int steps, int active, uint32_t* out, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<PIECES, VALU>), dim3(grid), dim3(256), 0, 0, t, n, steps, active, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return static_cast<double>(grid) * 4 * steps * active / best / 1e6; // G records / s
}

template <int PIECES>
uint4* make_table(uint32_t n) {
  std::vector<uint4> host(static_cast<size_t>(n) * PIECES);
  uint32_t st = 12345u;
  for (uint32_t r = 0; r < n; r++) {
    st ^= st << 13; st ^= st >> 17; st ^= st << 5;
    for (int p = 0; p < PIECES; p++) host[static_cast<size_t>(r) * PIECES + p] = make_uint4(st, st * 3u + p, st * 5u, st * 7u);
  }
  uint4* d;
  CHECK(hipMalloc(&d, host.size() * sizeof(uint4)));
  CHECK(hipMemcpy(d, host.data(), host.size() * sizeof(uint4), hipMemcpyHostToDevice));
  return d;
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
  const int grid = cus * 5, steps = 300;
  for (uint32_t n : { 400000u, 5000000u }) {
    uint4* t8 = make_table<8>(n);
    uint4* t4 = make_table<4>(n);
    for (int active : { 24, 36, 64 }) {
      std::printf("%7u records, %2d lanes: 128 B/8 loads valu 0 %6.2f | 160 %6.2f | 240 %6.2f || 64 B/4 loads valu 0 %6.2f | 160 %6.2f | 240 %6.2f | 320 %6.2f  (G records/s)\n",
                  n, active,
                  run<8, 0>(grid, t8, n, steps, active, d_out, e0, e1), run<8, 160>(grid, t8, n, steps, active, d_out, e0, e1),
                  run<8, 240>(grid, t8, n, steps, active, d_out, e0, e1),
                  run<4, 0>(grid, t4, n, steps, active, d_out, e0, e1), run<4, 160>(grid, t4, n, steps, active, d_out, e0, e1),
                  run<4, 240>(grid, t4, n, steps, active, d_out, e0, e1), run<4, 320>(grid, t4, n, steps, active, d_out, e0, e1));
      std::fflush(stdout);
    }
    CHECK(hipFree(t8));
    CHECK(hipFree(t4));
  }
  return 0;
}
