// Micro-benchmark: issue rate of the packed f32 vector instructions the euclidean kernel is built from.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk pk_fma_rate.hip && /tmp/pk
// Each variant runs ITER trips of 32 instructions per wave on NACC independent accumulators; waves per SIMD = 1, 2 or 4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int NACC>  // 0: v_pk_fma_f32, 1: v_fma_f32 (scalar), 2: v_pk_add_f32 (SGPR src) + v_pk_fma_f32 pairs, 3: v_pk_mul_f32
__global__ __launch_bounds__(256) void k(float* out, int iters, float s0) {
  f32x2 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x2{(float)threadIdx.x + i, 1.0f + i};
  const f32x2 b = f32x2{1.0001f, 0.9999f};
  f32x2 sv = f32x2{s0, s0 + 1.0f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(b));
        if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[i][0]) : "v"(b[0]));
        if (MODE == 2) {
          if ((r & 1) == 0) { f32x2 d; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_fma_f32 %3, %0, %0, %3" : "=&v"(d) : "s"(sv), "v"(b), "v"(acc[i])); }
        }
        if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
      }
  }
  float r = 0;
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1];
  if (r == 123.456f) out[0] = r;
}

template <int MODE, int NACC>
void run(const char* name, float* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4000;
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps;  // 256 CUs, 4 waves per block -> wps waves per SIMD
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    const double instr_per_simd = (double)iters * 32 * wps;
    printf("%-34s NACC %2d  waves/SIMD %d: %.2f cycles per wave-instruction at 2.4 GHz (%.3f ms)\n", name, NACC, wps,
           best * 1e-3 * 2.4e9 / instr_per_simd, best);
  }
}

int main() {
  float* out; hipMalloc(&out, 4);
  run<0, 16>("v_pk_fma_f32", out);
  run<0, 4>("v_pk_fma_f32", out);
  run<0, 1>("v_pk_fma_f32 (dependent chain)", out);
  run<1, 16>("v_fma_f32", out);
  run<1, 1>("v_fma_f32 (dependent chain)", out);
  run<3, 16>("v_pk_mul_f32", out);
  run<2, 16>("v_pk_add(sgpr)+v_pk_fma dependent", out);
  return 0;
}
