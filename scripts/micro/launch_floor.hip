// Micro-benchmark: what a dependent launch costs on this stack whatever it computes — the floor under the positional / tail launches
// (C4: k_frame 13 us + k_assign_label 4.4 + k_assign_solve 5.5; C3: k_frame 8 + k_assign_small 5.8).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_floor launch_floor.hip && /tmp/launch_floor
// Per shape (blocks x 256 threads, optional LDS, optional chain of dependent global loads per block): the average per-launch time of a
// chain of 200 back-to-back launches on one stream (event to event), and the dispatch's own begin -> end (hipExtLaunchKernelGGL
// events: the clock rocprofv3 reads).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

template <int LDS_BYTES, int LOADS>
__global__ __launch_bounds__(256) void k_floor(const unsigned* __restrict__ chain, unsigned* sink) {
  __shared__ unsigned char lds[LDS_BYTES > 0 ? LDS_BYTES : 4];
  unsigned v = blockIdx.x;
  // LOADS dependent global loads (each one's address comes from the previous one's value: a trip to L2 / memory per step)
#pragma unroll 1
  for (int i = 0; i < LOADS; ++i) v = chain[(v + threadIdx.x) & 0xffffu];
  if (LDS_BYTES > 0) { lds[threadIdx.x] = (unsigned char)v; __syncthreads(); v += lds[(threadIdx.x + 1) & 255]; }
  if (v == 0xffffffffu) *sink = v;
}

template <int LDS_BYTES, int LOADS>
static void run(const char* what, int blocks, const unsigned* chain, unsigned* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int n = 200;
  float best = 1e9f;
  for (int rep = 0; rep < 10; ++rep) {
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k_floor<LDS_BYTES, LOADS>), dim3(blocks), dim3(256), 0, 0, chain, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  float own = 0.f;
  for (int rep = 0; rep < 50; ++rep) {
    hipExtLaunchKernelGGL((k_floor<LDS_BYTES, LOADS>), dim3(blocks), dim3(256), 0, 0, a, b, 0, chain, sink);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    own += ms;
  }
  printf("%-58s blocks %5d: %6.2f us per launch in a chain, %6.2f us dispatch begin->end\n", what, blocks, best * 1000.f / n, own * 1000.f / 50);
}

int main() {
  unsigned *chain, *sink;
  hipMalloc(&chain, 65536 * 4);
  hipMalloc(&sink, 4);
  unsigned* h = (unsigned*)malloc(65536 * 4);
  for (int i = 0; i < 65536; ++i) h[i] = (unsigned)((i * 2654435761u) >> 16);
  hipMemcpy(chain, h, 65536 * 4, hipMemcpyHostToDevice);
  for (int blocks : {1, 8, 256, 1000, 4000}) {
    run<0, 0>("empty", blocks, chain, sink);
    run<0, 1>("one global load per thread", blocks, chain, sink);
    run<0, 3>("three dependent global loads", blocks, chain, sink);
    run<32768, 3>("32 KB of LDS + three dependent loads + barrier", blocks, chain, sink);
    run<32768, 8>("32 KB of LDS + eight dependent loads + barrier", blocks, chain, sink);
  }
  return 0;
}
