// h2d_rate.hip — how fast does one frame's worth of detections (2 MB of features at C2) cross PCIe, and by which means?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/h2d_rate.hip -o /tmp/h2d_rate && /tmp/h2d_rate
// Variants: one hipMemcpyAsync per frame on one stream | the frame split over 2 / 4 streams (several SDMA engines) | a copy KERNEL
// reading the pinned block through its device mapping (the shader engines pull over PCIe) with 64..1024 workgroups.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
  const size_t sizes[] = {256u << 10, 2u << 20, 8u << 20, 32u << 20};
  hipStream_t st[4];
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (size_t bytes : sizes) {
    void *h = nullptr, *d = nullptr, *hd = nullptr;
    CK(hipHostMalloc(&h, bytes, hipHostMallocPortable | hipHostMallocMapped));
    CK(hipMalloc(&d, bytes));
    CK(hipHostGetDevicePointer(&hd, h, 0));
    memset(h, 1, bytes);
    const int iters = bytes > (8u << 20) ? 50 : 300;
    auto run = [&](const char* name, auto body) {
      for (int i = 0; i < 5; ++i) body();
      for (auto& s : st) hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < iters; ++i) body();
      for (auto& s : st) hipStreamSynchronize(s);
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
      printf("%8zu KB  %-34s %8.1f us  %6.1f GB/s\n", bytes >> 10, name, us, bytes / us * 1e-3);
    };
    run("memcpyAsync x1 stream", [&] { hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0]); });
    run("memcpyAsync split over 2 streams", [&] { for (int k = 0; k < 2; ++k) hipMemcpyAsync((char*)d + k * bytes / 2, (char*)h + k * bytes / 2, bytes / 2, hipMemcpyHostToDevice, st[k]); });
    run("memcpyAsync split over 4 streams", [&] { for (int k = 0; k < 4; ++k) hipMemcpyAsync((char*)d + k * bytes / 4, (char*)h + k * bytes / 4, bytes / 4, hipMemcpyHostToDevice, st[k]); });
    for (int blocks : {32, 64, 128, 256, 1024}) {
      char nm[64];
      snprintf(nm, sizeof nm, "copy kernel, %d x 256 threads", blocks);
      run(nm, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st[0], (const float4*)hd, (float4*)d, bytes / 16); });
    }
    // one-at-a-time latency (sync after each): what a synchronous sa_associate pays
    {
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 50; ++i) { hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0]); hipStreamSynchronize(st[0]); }
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 50;
      printf("%8zu KB  %-34s %8.1f us  %6.1f GB/s\n", bytes >> 10, "memcpyAsync + sync each", us, bytes / us * 1e-3);
    }
    hipFree(d);
    hipHostFree(h);
  }
  return 0;
}
