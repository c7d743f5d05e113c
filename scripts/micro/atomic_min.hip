// Micro-benchmark: what the BestFit vote's per-row / per-column reductions cost as device-scope 64-bit atomic minima issued by the
// contraction tiles (16 x 16 tiles of 64 x 64, every tile one minimum per row and per column it covers: 16-way contention per
// word, 32 k atomics in flight as the single round of tiles retires) against the plain partial stores the tiles do today.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_min atomic_min.hip && /tmp/atomic_min
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>  // 0: nothing, 1: partial stores (f64 + u32 per row and per column), 2: u64 atomic min, no return
__global__ __launch_bounds__(256) void k_tiles(uint64_t* rowb, uint64_t* colb, double* rpw, uint32_t* rpt, double* cpw, uint32_t* cpq,
                                               uint32_t gx, uint32_t N, uint32_t T, uint32_t spin) {
  const uint32_t bx = blockIdx.x % gx, by = blockIdx.x / gx;
  // stand-in for the tile's contraction: a dependent chain of ~spin cycles
  uint32_t x = threadIdx.x;
  for (uint32_t i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
  const uint32_t t = threadIdx.x;
  if (t < 64) {
    const uint32_t row = by * 64 + t;
    const uint64_t key = ((uint64_t)(x | 1u) << 32) | (bx * 64 + (x & 63u));
    if (row < N) {
      if (MODE == 1) { rpw[(size_t)bx * N + row] = (double)x; rpt[(size_t)bx * N + row] = (uint32_t)key; }
      if (MODE == 2) __hip_atomic_fetch_min(rowb + row, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (t < 128) {
    const uint32_t col = bx * 64 + (t - 64);
    const uint64_t key = ((uint64_t)(x | 1u) << 32) | (by * 64 + (x & 63u));
    if (col < T) {
      if (MODE == 1) { cpw[(size_t)by * T + col] = (double)x; cpq[(size_t)by * T + col] = (uint32_t)key; }
      if (MODE == 2) __hip_atomic_fetch_min(colb + col, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (x == 0xdeadbeefu) rowb[0] = x;
}

int main() {
  const uint32_t N = 1000, T = 1000, gx = 16, gy = 16;
  uint64_t *rowb, *colb; double *rpw, *cpw; uint32_t *rpt, *cpq;
  hipMalloc(&rowb, 8 * 1024); hipMalloc(&colb, 8 * 1024);
  hipMalloc(&rpw, 8 * 16 * 1024); hipMalloc(&cpw, 8 * 16 * 1024); hipMalloc(&rpt, 4 * 16 * 1024); hipMalloc(&cpq, 4 * 16 * 1024);
  hipMemset(rowb, 0xff, 8 * 1024); hipMemset(colb, 0xff, 8 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (uint32_t spin : {0u, 2000u}) {
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f, sum = 0;
      for (int rep = 0; rep < 30; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) {
          if (mode == 0) hipLaunchKernelGGL(k_tiles<0>, dim3(gx * gy), dim3(256), 0, 0, rowb, colb, rpw, rpt, cpw, cpq, gx, N, T, spin);
          if (mode == 1) hipLaunchKernelGGL(k_tiles<1>, dim3(gx * gy), dim3(256), 0, 0, rowb, colb, rpw, rpt, cpw, cpq, gx, N, T, spin);
          if (mode == 2) hipLaunchKernelGGL(k_tiles<2>, dim3(gx * gy), dim3(256), 0, 0, rowb, colb, rpw, rpt, cpw, cpq, gx, N, T, spin);
        }
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        if (rep >= 10) sum += ms;
      }
      printf("spin %u mode %d (%s): best %.2f us, mean %.2f us per launch\n", spin, mode,
             mode == 0 ? "nothing" : mode == 1 ? "partial stores" : "u64 atomic min", best * 100.f, sum / 20 * 100.f);
    }
  }
  return 0;
}
