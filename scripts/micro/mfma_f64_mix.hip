// mfma_f64_mix.hip — what does a wave of dependent f64 vector work cost beside a wave that keeps the SIMD's matrix pipe busy, and
// does the LENGTH of the matrix instruction matter?  (a positional tile lives 33.7 k cycles beside the contraction's tiles,
// 14 k in a launch of its own.)
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_f64_mix.hip -o /tmp/mfma_f64_mix && /tmp/mfma_f64_mix
// One launch of 512 blocks x 256 threads on 256 CUs: the first 256 blocks issue NM matrix instructions per wave (kind 1: v_mfma_f32_32x32x2_f32,
// one accumulator chain, 64 cycles each; kind 2: v_mfma_f32_16x16x4_f32, four accumulators round-robin, 32 cycles each; kind 3:
// 32x32x2 on two alternating accumulators), the other 256 run a chain of CH dependent f64 fused multiply-adds per lane.  Each block
// stamps s_memtime at start and end; printed: average / maximum cycles per kind and the launch time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k_mix(uint64_t* stamps, float* sink, double* dsink, int nm, int ch, int f64on, int f64waves) {
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  const bool matrix = blockIdx.x < gridDim.x / 2;  // workgroups go round the XCDs and then round each XCD's CUs: the first half of the grid puts
                                                   // one block on every CU, the second half a second one
  if (matrix) {
    if (KIND == 0) return;
    float a = 1.0f + threadIdx.x * 1e-3f, b = 1.0f - threadIdx.x * 1e-3f;
    if (KIND == 1) {
      f32x16 acc = {0};
      for (int i = 0; i < nm; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[15];
    } else if (KIND >= 4) {
      // the matrix wave steps aside between its instructions: s_nop keeps it away from the vector issue port while the matrix
      // pipe works (a dependent v_mfma waiting at the port keeps every other wave's vector instructions out)
      f32x16 acc = {0};
      for (int i = 0; i < nm; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (KIND == 4) asm volatile("s_nop 15");
        if (KIND == 5) asm volatile("s_nop 15\n s_nop 15");
        if (KIND == 6) asm volatile("s_nop 15\n s_nop 15\n s_nop 15");
        if (KIND == 7) asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 7");
        __builtin_amdgcn_sched_barrier(0);
      }
      sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[15];
    } else if (KIND == 3) {
      f32x16 acc0 = {0}, acc1 = {0};
      for (int i = 0; i < nm; i += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
      }
      sink[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[15];
    } else if (KIND == 2) {
      f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      for (int i = 0; i < nm / 2; ++i) {  // four 16x16x4 = 4096 multiply-accumulates = two 32x32x2
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, c3, 0, 0, 0);
      }
      sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
  } else {
    if (!f64on) return;
    if ((int)(threadIdx.x >> 6) >= f64waves) return;
    if (f64on == 3) __builtin_amdgcn_s_setprio(3);
    if (f64on == 2) {  // the same chain in f32, for comparison
      float x = 1.0f + threadIdx.x * 1e-6f, y = 0.999999f, z = 1e-7f;
#pragma unroll 16
      for (int i = 0; i < ch; ++i) x = __builtin_fmaf(x, y, z);
      sink[blockIdx.x * 256 + threadIdx.x] = x;
    } else {
      double x = 1.0 + threadIdx.x * 1e-9, y = 0.999999, z = 1e-7;
#pragma unroll 16
      for (int i = 0; i < ch; ++i) x = __builtin_fma(x, y, z);  // dependent chain
      dsink[blockIdx.x * 256 + threadIdx.x] = x;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime(); }
}

int main() {
  const int NB = 512;
  uint64_t* d_st; float* d_sink; double* d_ds;
  CK(hipMalloc(&d_st, NB * 16)); CK(hipMalloc(&d_sink, NB * 256 * 4)); CK(hipMalloc(&d_ds, NB * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<uint64_t> h(2 * NB);
  const int nm = 256, ch = 1008;  // 256 x 64 cycles = 16.4 k cycles of matrix issue, ~1000 dependent f64 steps: the C2 tile pair
  auto run = [&](const char* name, int kind, int f64on, int f64waves) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(d_st, 0, NB * 16));
      CK(hipEventRecord(e0));
      switch (kind) {
        case 0: hipLaunchKernelGGL(k_mix<0>, dim3(NB), dim3(256), 0, 0, d_st, d_sink, d_ds, nm, ch, f64on, f64waves); break;
        case 1: hipLaunchKernelGGL(k_mix<1>, dim3(NB), dim3(256), 0, 0, d_st, d_sink, d_ds, nm, ch, f64on, f64waves); break;
        case 2: hipLaunchKernelGGL(k_mix<2>, dim3(NB), dim3(256), 0, 0, d_st, d_sink, d_ds, nm, ch, f64on, f64waves); break;
        case 4: hipLaunchKernelGGL(k_mix<4>, dim3(NB), dim3(256), 0, 0, d_st, d_sink, d_ds, nm, ch, f64on, f64waves); break;
        case 5: hipLaunchKernelGGL(k_mix<5>, dim3(NB), dim3(256), 0, 0, d_st, d_sink, d_ds, nm, ch, f64on, f64waves); break;
        case 6: hipLaunchKernelGGL(k_mix<6>, dim3(NB), dim3(256), 0, 0, d_st, d_sink, d_ds, nm, ch, f64on, f64waves); break;
        case 7: hipLaunchKernelGGL(k_mix<7>, dim3(NB), dim3(256), 0, 0, d_st, d_sink, d_ds, nm, ch, f64on, f64waves); break;
        case 3: hipLaunchKernelGGL(k_mix<3>, dim3(NB), dim3(256), 0, 0, d_st, d_sink, d_ds, nm, ch, f64on, f64waves); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(h.data(), d_st, NB * 16, hipMemcpyDeviceToHost));
    double sm = 0, sf = 0; uint64_t mm = 0, mf = 0; int cm = 0, cf = 0;
    for (int b = 0; b < NB; ++b) {
      if (!h[2 * b + 1]) continue;
      uint64_t d = h[2 * b + 1] - h[2 * b];
      if (b >= NB / 2) { sf += d; mf = std::max(mf, d); ++cf; } else { sm += d; mm = std::max(mm, d); ++cm; }
    }
    printf("%-58s launch %7.2f us | matrix blocks avg %7.0f max %7llu | f64 blocks avg %7.0f max %7llu (s_memtime ticks)\n", name, ms * 1e3,
           cm ? sm / cm : 0.0, (unsigned long long)mm, cf ? sf / cf : 0.0, (unsigned long long)mf);
    return 0;
  };
  run("f64 chain alone (4 waves)", 0, 1, 4);
  run("f64 chain alone (1 wave)", 0, 1, 1);
  run("32x32x2 alone", 1, 0, 4);
  run("32x32x2 two accumulators alone", 3, 0, 4);
  run("16x16x4 alone", 2, 0, 4);
  run("32x32x2 + f64 chain (4 waves)", 1, 1, 4);
  run("32x32x2 two accumulators + f64 chain (4 waves)", 3, 1, 4);
  run("16x16x4 + f64 chain (4 waves)", 2, 1, 4);
  run("32x32x2 + f64 chain at s_setprio 3 (4 waves)", 1, 3, 4);
  run("32x32x2 + s_nop 15 alone", 4, 0, 4);
  run("32x32x2 + 2 x s_nop 15 alone", 5, 0, 4);
  run("32x32x2 + 3 x s_nop 15 alone", 6, 0, 4);
  run("32x32x2 + 3 x s_nop 15 + s_nop 7 alone", 7, 0, 4);
  run("32x32x2 + s_nop 15 + f64 chain (4 waves)", 4, 1, 4);
  run("32x32x2 + 2 x s_nop 15 + f64 chain (4 waves)", 5, 1, 4);
  run("32x32x2 + 3 x s_nop 15 + f64 chain (4 waves)", 6, 1, 4);
  run("32x32x2 + 3 x s_nop 15 + s_nop 7 + f64 chain (4 waves)", 7, 1, 4);
  run("32x32x2 + 3 x s_nop 15 + f64 chain (1 wave)", 6, 1, 1);
  run("f32 chain alone (4 waves)", 0, 2, 4);
  run("32x32x2 + f32 chain (4 waves)", 1, 2, 4);
  run("16x16x4 + f32 chain (4 waves)", 2, 2, 4);
  run("32x32x2 + f64 chain (1 wave)", 1, 1, 1);
  run("16x16x4 + f64 chain (1 wave)", 2, 1, 1);
  return 0;
}
