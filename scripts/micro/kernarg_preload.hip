// Micro-benchmark: does kernarg preloading (first SGPRs filled by the command processor instead of an s_load at wave start)
// shorten a dependent launch?  Build twice:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/kp0 kernarg_preload.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o /tmp/kp1 kernarg_preload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
struct Desc { const unsigned* src; unsigned* dst; unsigned n; unsigned pad; };
// one dependent chain per kernel: kernarg -> descriptor -> data -> store   (the shape of the engine's small kernels)
__global__ void k_desc(const Desc* d, unsigned* sink) {
  const Desc D = *d;
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.n) D.dst[i] = D.src[i] + 1u;
  if (i == 0xffffffffu) *sink = 1;
}
// the same with the hot pointers in the kernarg itself: kernarg -> data -> store
__global__ void k_args(const unsigned* src, unsigned* dst, unsigned n, unsigned* sink) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] + 1u;
  if (i == 0xffffffffu) *sink = 1;
}
int main() {
  unsigned *a, *b, *sink; Desc* d;
  const unsigned n = 1024;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&sink, 4); hipMalloc(&d, sizeof(Desc));
  hipMemset(a, 0, n * 4);
  Desc h{a, b, n, 0};
  hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 2; ++variant) {
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 200; ++i) {
        if (variant == 0) hipLaunchKernelGGL(k_desc, dim3(4), dim3(256), 0, 0, d, sink);
        else hipLaunchKernelGGL(k_args, dim3(4), dim3(256), 0, 0, (const unsigned*)a, b, n, sink);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%s: %.2f us per dependent launch\n", variant == 0 ? "kernarg -> descriptor -> data" : "kernarg -> data", best * 1000.f / 200.f);
  }
  return 0;
}
