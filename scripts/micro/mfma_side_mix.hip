// mfma_side_mix.hip — ONE wave per SIMD running the contraction's instruction mix: how many cycles per 2048 multiply-accumulates does
// the matrix pipe see when the wave also has to issue its LDS fragment reads, its staging stores and its global loads — with
// v_mfma_f32_32x32x2_f32 (one accumulator chain, 64 cycles per instruction) and with v_mfma_f32_16x16x4_f32 (four accumulators, 32 cycles)?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_side_mix.hip -o /tmp/mfma_side_mix && /tmp/mfma_side_mix
// Per 16 values of k of a 32 x 32 wave tile both forms need 4 ds_read_b128 (A and B fragments), and the staging of the next chunk adds
// ~1 ds_write_b128 + 1 global_load_dwordx4; the work is 8 x 32x32x2 or 16 x 16x16x4 = 512 cycles of matrix pipe.  SIDE = 0: matrix
// instructions only; 1: + fragment reads; 2: + staging store and load; 3: + one s_barrier per 32 values of k (a 256-thread block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int SIDE>
__global__ __launch_bounds__(256) void k_mix(uint64_t* stamps, float* sink, const f32x4* __restrict__ src, int iters) {
  __shared__ f32x4 lds[2][1024];
  const uint32_t t = threadIdx.x;
  lds[0][t] = f32x4{1.f, 2.f, 3.f, 4.f}; lds[0][t + 256] = f32x4{1.f, 2.f, 3.f, 4.f}; lds[0][t + 512] = f32x4{.5f, .25f, 1.f, 2.f}; lds[0][t + 768] = f32x4{1.f, 1.f, 1.f, 1.f};
  lds[1][t] = lds[0][t]; lds[1][t + 256] = lds[0][t]; lds[1][t + 512] = lds[0][t]; lds[1][t + 768] = lds[0][t];
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  f32x4 stage = src[blockIdx.x * 256 + t];
  float out = 0.f;
  if (KIND == 1) {
    f32x16 acc = {0};
    f32x4 fa = lds[0][t], fb = lds[0][t + 256];
    for (int i = 0; i < iters; ++i) {  // one iteration = 16 values of k: 8 matrix instructions, 4 fragment reads
      const int st = i & 1;
      f32x4 fa2, fb2, fa3, fb3;
      if (SIDE >= 1) { fa2 = lds[st][(t + 64) & 1023]; fb2 = lds[st][(t + 320) & 1023]; }
      else { fa2 = fa; fb2 = fb; }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
      if (SIDE >= 2) { lds[st ^ 1][(t * 3 + i) & 1023] = stage; stage = src[((blockIdx.x * 256 + t) + (i + 1) * 4096) & 0xfffff]; }
      if (SIDE >= 1) { fa3 = lds[st][(t + 128) & 1023]; fb3 = lds[st][(t + 384) & 1023]; }
      else { fa3 = fa2; fb3 = fb2; }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa2[e], fb2[e], acc, 0, 0, 0);
      if (SIDE >= 3 && (i & 1)) __syncthreads();
      fa = fa3; fb = fb3;
    }
    out = acc[0] + acc[7] + acc[15];
  } else {
    f32x4 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    f32x4 a0 = lds[0][t], a1 = lds[0][t + 64], b0 = lds[0][t + 256], b1 = lds[0][t + 320];
    for (int i = 0; i < iters; ++i) {  // one iteration = 16 values of k: 16 matrix instructions, 4 fragment reads
      const int st = i & 1;
      f32x4 na0, na1, nb0, nb1;
      if (SIDE >= 1) { na0 = lds[st][(t + 128) & 1023]; nb0 = lds[st][(t + 384) & 1023]; }
      else { na0 = a0; nb0 = b0; }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b1[e], c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b0[e], c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], c11, 0, 0, 0);
      }
      if (SIDE >= 2) { lds[st ^ 1][(t * 3 + i) & 1023] = stage; stage = src[((blockIdx.x * 256 + t) + (i + 1) * 4096) & 0xfffff]; }
      if (SIDE >= 1) { na1 = lds[st][(t + 192) & 1023]; nb1 = lds[st][(t + 448) & 1023]; }
      else { na1 = a1; nb1 = b1; }
#pragma unroll
      for (int e = 2; e < 4; ++e) {
        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b1[e], c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b0[e], c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], c11, 0, 0, 0);
      }
      if (SIDE >= 3 && (i & 1)) __syncthreads();
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    out = c00[0] + c01[1] + c10[2] + c11[3];
  }
  sink[blockIdx.x * 256 + t] = out + stage[0];
  __syncthreads();
  if (t == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime(); }
}

int main() {
  const int NB = 256, iters = 32;  // 32 x 16 = 512 values of k: the C2 tile
  uint64_t* d_st; float* d_sink; f32x4* d_src;
  CK(hipMalloc(&d_st, NB * 16)); CK(hipMalloc(&d_sink, NB * 256 * 4)); CK(hipMalloc(&d_src, (1u << 20) * 16 + 4096 * 16 * 64));
  CK(hipMemset(d_src, 0, (1u << 20) * 16));
  std::vector<uint64_t> h(2 * NB);
  auto report = [&](const char* name) {
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_st, NB * 16, hipMemcpyDeviceToHost);
    double s = 0; uint64_t m = 0;
    for (int b = 0; b < NB; ++b) { uint64_t d = h[2 * b + 1] - h[2 * b]; s += d; m = std::max(m, d); }
    printf("%-52s avg %7.0f max %7llu cycles per block = %6.1f per 2048 multiply-accumulates (64 = the pipe's rate)\n", name, s / NB, (unsigned long long)m, s / NB / (iters * 8));
  };
#define RUN(K_, S_, name) do { for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_mix<K_, S_>), dim3(NB), dim3(256), 0, 0, d_st, d_sink, d_src, iters); report(name); } while (0)
  RUN(1, 0, "32x32x2, matrix instructions only");
  RUN(2, 0, "16x16x4, matrix instructions only");
  RUN(1, 1, "32x32x2 + fragment reads");
  RUN(2, 1, "16x16x4 + fragment reads");
  RUN(1, 2, "32x32x2 + fragment reads + staging");
  RUN(2, 2, "16x16x4 + fragment reads + staging");
  RUN(1, 3, "32x32x2 + fragment reads + staging + barrier");
  RUN(2, 3, "16x16x4 + fragment reads + staging + barrier");
  return 0;
}
