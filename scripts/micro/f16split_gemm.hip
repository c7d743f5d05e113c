// Micro-benchmark for DESIGN §7: the feature contraction with f16-split operands on the f16 matrix cores.
//   C[M][N] = A[M][K] . B[N][K]^T, A and B f32 in HBM.  Every element is split in the kernel, x = hi + lo / 2048 with hi, lo in
//   f16 (22 bits of x), and the three products hi.hi + (hi.lo + lo.hi) / 2048 run on v_mfma_f32_32x32x16_f16 with f32 accumulation.
//   Prints time, algorithmic TFLOP/s (2MNK) and the largest error against an f64 reference on a sample of cells.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f16split f16split_gemm.hip && /tmp/f16split
// Not product code: no edge handling (M, N multiples of 128, K of 32), unit-norm rows assumed (no per-row scaling).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
// byte offset of 16-B slot q (8 halves) of row r in a [rows][32 halves] tile; slots XOR-swizzled so that 8 consecutive rows
// hit 8 different 16-B bank groups
__device__ __forceinline__ uint32_t slot_off(uint32_t r, uint32_t q) { return r * 64u + ((q ^ ((r >> 1) & 3u)) << 4); }

__global__ __launch_bounds__(256) void k_f16split(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                  int M, int N, int K) {
  // stage = A_hi | A_lo | B_hi | B_lo, each [128][32] halves = 8 KB
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][4][BM * 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, wm = w >> 1, wn = w & 1u, lr = lane & 31u, lh = lane >> 5;
  const uint32_t m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  // staging: 128 rows x 8 float4 per operand = 1024 float4 -> 4 per thread per operand
  f4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t c = tid + 256u * i, row = c >> 3, kc = c & 7u;
      ra[i] = *(const f4*)(A + (size_t)(m0 + row) * K + k0 + kc * 4);
      rb[i] = *(const f4*)(B + (size_t)(n0 + row) * K + k0 + kc * 4);
    }
  };
  auto split_store = [&](int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t c = tid + 256u * i, row = c >> 3, kc = c & 7u;
      const uint32_t off = slot_off(row, kc >> 1) + (kc & 1u) * 8u;
      h4 ah, al, bh, bl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xa = ra[i][e], xb = rb[i][e];
        ah[e] = (_Float16)xa; al[e] = (_Float16)((xa - (float)ah[e]) * 2048.0f);
        bh[e] = (_Float16)xb; bl[e] = (_Float16)((xb - (float)bh[e]) * 2048.0f);
      }
      *(h4*)(lds[st][0] + off) = ah; *(h4*)(lds[st][1] + off) = al;
      *(h4*)(lds[st][2] + off) = bh; *(h4*)(lds[st][3] + off) = bl;
    }
  };
  f16v acc[2][2], accx[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[m][n][e] = 0.f; accx[m][n][e] = 0.f; }
  const int nchunks = K / BK;
  gload(0);
  split_store(0);
  if (nchunks > 1) gload(BK);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int st = c & 1;
    if (c + 1 < nchunks) split_store(st ^ 1);          // chunk c+1 (in registers) -> other stage
    if (c + 2 < nchunks) gload((c + 2) * BK);          // chunk c+2 -> registers
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const uint32_t r = wm * 64 + m * 32 + lr, o = slot_off(r, 2 * ks + lh);
        ah[m] = *(const h8*)(lds[st][0] + o); al[m] = *(const h8*)(lds[st][1] + o);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const uint32_t r = wn * 64 + n * 32 + lr, o = slot_off(r, 2 * ks + lh);
        bh[n] = *(const h8*)(lds[st][2] + o); bl[n] = *(const h8*)(lds[st][3] + o);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
          accx[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], accx[m][n], 0, 0, 0);
          accx[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], accx[m][n], 0, 0, 0);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t row = m0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = n0 + wn * 64 + n * 32 + lr;
        C[(size_t)row * N + col] = acc[m][n][r] + accx[m][n][r] * (1.0f / 2048.0f);
      }
}

static void run(int M, int N, int K) {
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  srand(1);
  auto fill = [&](std::vector<float>& v, int rows) {
    for (int r = 0; r < rows; ++r) {
      double s = 0;
      for (int k = 0; k < K; ++k) { float x = fabsf((float)rand() / RAND_MAX - 0.3f); v[(size_t)r * K + k] = x; s += (double)x * x; }
      const float inv = (float)(1.0 / sqrt(s));
      for (int k = 0; k < K; ++k) v[(size_t)r * K + k] *= inv;   // unit rows, non-negative-ish like ReID features
    }
  };
  fill(hA, M); fill(hB, N);
  float *A, *B, *C;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(N / BN, M / BM);
  float best = 1e9f;
  for (int rep = 0; rep < 10; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_f16split, grid, dim3(256), 0, 0, A, B, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  std::vector<float> hC((size_t)M * N);
  hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int t = 0; t < 4096; ++t) {
    const int i = rand() % M, j = rand() % N;
    double s = 0;
    for (int k = 0; k < K; ++k) s += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
    maxerr = fmax(maxerr, fabs(s - hC[(size_t)i * N + j]));
  }
  printf("M %d N %d K %d: %.1f us, %.1f TFLOP/s algorithmic, max |err| vs f64 on 4096 cells %.2e\n", M, N, K, best * 1000.f,
         2.0 * M * N * K / (best * 1e-3) / 1e12, maxerr);
  hipFree(A); hipFree(B); hipFree(C);
}

int main() {
  run(1024, 1024, 512);     // C2 padded
  run(2048, 5120, 4096);    // C5 padded
  run(4096, 4096, 4096);
  return 0;
}
