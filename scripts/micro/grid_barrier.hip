// Micro-benchmark: cost of a grid-wide barrier built from one device-scope counter, against the cost of a dependent launch.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier grid_barrier.hip && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_barriers(unsigned* counter, int rounds, unsigned* sink) {
  unsigned acc = 0;
  for (int r = 0; r < rounds; ++r) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * gridDim.x;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    acc += r;
  }
  if (acc == 0xffffffffu) *sink = acc;
}
__global__ void k_empty(unsigned* sink) { if (threadIdx.x == 9999) *sink = 1; }

int main() {
  unsigned *counter, *sink;
  hipMalloc(&counter, 4); hipMalloc(&sink, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int blocks : {64, 256, 512}) {
    for (int rounds : {1, 11, 101}) {
      float best = 1e9f;
      for (int rep = 0; rep < 20; ++rep) {
        hipMemset(counter, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(256), 0, 0, counter, rounds, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      printf("blocks %d rounds %d: %.2f us total\n", blocks, rounds, best * 1000.f);
    }
  }
  // chain of dependent empty launches
  for (int n : {1, 11, 101}) {
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(a);
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, sink);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    printf("%d dependent empty launches: %.2f us total\n", n, best * 1000.f);
  }
  return 0;
}
