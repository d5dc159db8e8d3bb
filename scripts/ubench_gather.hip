// Microbenchmark behind DESIGN.md "what bounds traversal": cost of divergent, dependent 16-byte
// gathers on MI355X as a function of (active lanes per wave, loads per 128-byte record, lanes that
// share a record, table size).  Every step of a lane loads L consecutive 16-byte pieces of one
// 128-byte record and derives the next record from what it read (a pointer chase, like a BVH walk).
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_gather.hip -o scripts/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int L, bool WAIT_FIRST = false>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ table, uint32_t n_records, int steps,
                                                int active, int share, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  if (static_cast<int>(lane) < active) {
    // lanes of one `share` group walk the same records and read different pieces of them
    uint32_t cur = ((gid / share) * 2654435761u) % n_records;
    const uint32_t piece0 = (lane % share) * L % 8u;
    for (int s = 0; s < steps; s++) {
      const uint4* rec = table + static_cast<size_t>(cur) * 8u;
      uint4 v[L];
      if (WAIT_FIRST) {
        // first piece alone, wait for it, then the rest (they then hit a line that has arrived)
        v[0] = rec[piece0 & 7u];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[0].z), "+v"(v[0].w) : : "memory");
#pragma unroll
        for (int k = 1; k < L; k++) v[k] = rec[(piece0 + k) & 7u];
      } else {
#pragma unroll
      for (int k = 0; k < L; k++) v[k] = rec[(piece0 + k) & 7u];
      }
      uint32_t x = 0;
#pragma unroll
      for (int k = 0; k < L; k++) x ^= v[k].x + v[k].y + v[k].z + v[k].w;
      acc += x;
      // next record: the table holds random numbers; all lanes of a share group must agree, so they
      // combine what the group read (piece-independent: word 0 of the record is replicated in every piece's .x)
      cur = v[0].x % n_records;
    }
  }
  if (acc == 0x12345678u) out[gid] = acc;
}

int main(int argc, char** argv) {
  const int blocks_per_cu = argc > 1 ? std::atoi(argv[1]) : 5;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate / 1e6;
  std::printf("device %s, %d CUs, %.2f GHz, %d blocks (of 4 waves) per CU\n", prop.gcnArchName, cus, ghz, blocks_per_cu);
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, 4u << 20));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const size_t sizes_mb[] = { 1, 48 };
  for (size_t mb : sizes_mb) {
    const uint32_t n_records = static_cast<uint32_t>((mb << 20) / 128);
    std::vector<uint4> host(static_cast<size_t>(n_records) * 8);
    uint32_t st = 12345u;
    for (uint32_t r = 0; r < n_records; r++) {
      st ^= st << 13; st ^= st >> 17; st ^= st << 5;
      for (int p = 0; p < 8; p++) host[static_cast<size_t>(r) * 8 + p] = make_uint4(st, st * 3u + p, st * 5u, st * 7u);
    }
    uint4* d_table;
    CHECK(hipMalloc(&d_table, host.size() * sizeof(uint4)));
    CHECK(hipMemcpy(d_table, host.data(), host.size() * sizeof(uint4), hipMemcpyHostToDevice));
    const int steps = 400;
    const int grid = cus * blocks_per_cu;
    for (int share : { 1, 4 }) {
      for (int active : { 8, 16, 32, 64 }) {
        for (int Lx : { 1, 2, 8, 9 }) {
          const int L = Lx == 9 ? 8 : Lx; // 9 = 8 loads, the first one waited for before the others are issued
          if (share == 4 && L == 8) continue;
          float best = 1e30f;
          for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (Lx == 1) hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), 0, 0, d_table, n_records, steps, active, share, d_out);
            else if (Lx == 2) hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(256), 0, 0, d_table, n_records, steps, active, share, d_out);
            else if (Lx == 8) hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(256), 0, 0, d_table, n_records, steps, active, share, d_out);
            else hipLaunchKernelGGL((k_gather<8, true>), dim3(grid), dim3(256), 0, 0, d_table, n_records, steps, active, share, d_out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
          }
          const double wave_instr_per_cu = static_cast<double>(blocks_per_cu) * 4 * steps * L;
          const double cycles = best * 1e-3 * ghz * 1e9;
          const double records = static_cast<double>(grid) * 4 * steps * (active / share);
          std::printf("table %4zu MB share %d active %2d loads/step %d%s: %7.3f ms  %6.1f cycles per wave-instr per CU  %7.2f G records/s  %6.2f TB/s useful\n",
                      mb, share, active, L, Lx == 9 ? "w" : " ", best, cycles / wave_instr_per_cu, records / best / 1e6,
                      static_cast<double>(grid) * 4 * steps * active * L * 16 / best / 1e9);
        }
      }
    }
    CHECK(hipFree(d_table));
  }
  return 0;
}
