// sa_gemm.hip — the N x (T*K) feature-distance contraction of VisualSORT on gfx950.
//
// Reference: distance::cosine / distance::euclidean (src/distance.rs:9-47) evaluated for every
// (candidate, stored observation) pair by VisualMetric::visual_metric (visual_sort/metric.rs:200-225).
//
// cosine   : dot products on the f32 matrix cores — v_mfma_f32_32x32x2_f32, exact f32 (an fmaf chain),
//            157 TF/s peak; A = candidates [N][Dp], B = track bank [T*K][Dp], both k-contiguous, so the
//            contraction is C = A * B^T.  Block tile 128x128 (big frames) or 64x64 (small frames), 4 waves
//            as 2x2 per k-group, k staged 32 floats at a time through XOR-swizzled LDS, ds_read_b128
//            fragments (each lane takes 4 consecutive k of its row; the k-slot permutation is the same for A
//            and B, so the sum is unchanged).  Small frames cannot fill 256 CUs x 4 SIMDs with 32x32 wave
//            tiles, so the 64x64 kernel splits k across KG wave groups inside the block (each group owns its
//            LDS stage and accumulators; one LDS reduction at the end) — 2-4 waves per SIMD instead of 1.
//            The epilogue fuses everything the reference does per pair after the dot product:
//            d = dot / sqrt(n1*n2) with hoisted norms, is_ok threshold, distance_to_weight (1 - d), the
//            feature_can_be_used / minimal-track-length gates, compatible(), and the running maximum that
//            BestFitVoting needs (voting/best.rs:59-76).
// euclidean: sum (a-b)^2 directly on the VALU — the GEMM expansion |a|^2+|b|^2-2ab cancels catastrophically
//            on near-identical vectors, which are exactly the true matches (SURVEY §7 hard parts).
#include "sa_engine.h"

#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BK 32

// Physical float offset of logical 16-byte chunk `kc` (0..7) of row `row` in a [rows][32] f32 LDS tile.
// XOR with (row>>1)&7: with a 128-B row stride, the 16 rows one ds_read_b128 lane group touches land on
// 16 distinct 16-B slots of the 256-B bank row (conflict-free); see MI355X_MICROARCH.md §LDS.
__device__ __forceinline__ uint32_t lds_off(uint32_t row, uint32_t kc) { return row * BK + ((kc ^ ((row >> 1) & 7u)) << 2); }

struct GemmCols {   // per-lane column metadata kept in registers through the epilogue
  float nb;
  bool ok;
  sa_geo g;
  uint64_t epoch;
};

// Dp is a multiple of 32, so a 32-float chunk is either entirely inside a row or absent; rows past the
// matrix edge are clamped to the last row (their results are never stored) — no branches around the loads.
//
// Pipeline per k-group (256 threads, 2 LDS stages of (BM+BN) x 32 floats), ONE barrier per 32-deep chunk:
//   iteration c:  ds_read the fragments of stage c&1 | ds_write chunk c+1 (already in registers) to the other
//                 stage | issue the global loads of chunk c+2 | MFMAs of chunk c | barrier.
// The LDS write and the L2/HBM loads of the next chunks sit between the MFMAs of the current one, so the matrix
// pipe only sees one LDS read latency and one barrier per chunk.
template <int BM, int BN, int KG>
__device__ __forceinline__ void gemm_mainloop(const float* __restrict__ A, const float* __restrict__ B, uint32_t M,
                                              uint32_t Ncols, uint32_t Dp, uint32_t m0, uint32_t n0, float* lds,
                                              f32x16 (&acc)[BM / 64][BN / 64]) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;  // 16-B chunks per thread per stage
  constexpr int STAGE = (BM + BN) * BK;                     // floats per LDS stage
  const uint32_t tid = threadIdx.x;
  const uint32_t kg = tid >> 8;          // k-group of this wave (0 when KG == 1)
  const uint32_t ltid = tid & 255u, lane = tid & 63u, w4 = (tid >> 6) & 3u;
  float* base = lds + kg * 2 * STAGE;
  const uint32_t wm = w4 >> 1, wn = w4 & 1u;
  const uint32_t lr = lane & 31u, lh = lane >> 5;
  const uint32_t nchunks = Dp / BK;
  const uint32_t niter = (nchunks + KG - 1) / KG;
  uint32_t ga[A_CH], gb[B_CH], sa_[A_CH], sb_[B_CH];
#pragma unroll
  for (int r = 0; r < A_CH; ++r) {
    uint32_t c = ltid + 256u * r, row = c >> 3, kc = c & 7u;
    uint32_t gr = m0 + row;
    gr = gr < M ? gr : M - 1;
    ga[r] = gr * Dp + kc * 4u;
    sa_[r] = lds_off(row, kc);
  }
#pragma unroll
  for (int r = 0; r < B_CH; ++r) {
    uint32_t c = ltid + 256u * r, row = c >> 3, kc = c & 7u;
    uint32_t gr = n0 + row;
    gr = gr < Ncols ? gr : Ncols - 1;
    gb[r] = gr * Dp + kc * 4u;
    sb_[r] = BM * BK + lds_off(row, kc);
  }
  f32x4 ra[A_CH], rb[B_CH];
  auto gload = [&](uint32_t chunk) {
    const uint32_t k0 = chunk * BK;
#pragma unroll
    for (int r = 0; r < A_CH; ++r) ra[r] = *(const f32x4*)(A + (size_t)(ga[r] + k0));
#pragma unroll
    for (int r = 0; r < B_CH; ++r) rb[r] = *(const f32x4*)(B + (size_t)(gb[r] + k0));
  };
  auto lstore = [&](float* st) {
#pragma unroll
    for (int r = 0; r < A_CH; ++r) *(f32x4*)(st + sa_[r]) = ra[r];
#pragma unroll
    for (int r = 0; r < B_CH; ++r) *(f32x4*)(st + sb_[r]) = rb[r];
  };
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

  uint32_t aoff[TM], boff[TN];
#pragma unroll
  for (int m = 0; m < TM; ++m) aoff[m] = wm * (BM / 2) + m * 32 + lr;
#pragma unroll
  for (int n = 0; n < TN; ++n) boff[n] = wn * (BN / 2) + n * 32 + lr;

  if (kg < nchunks) { gload(kg); lstore(base); }
  if (kg + KG < nchunks) gload(kg + KG);
  __syncthreads();
  for (uint32_t it = 0; it < niter; ++it) {
    const uint32_t chunk = it * KG + kg;          // uniform inside a k-group
    const float* As = base + (it & 1u) * STAGE;
    const float* Bs = As + BM * BK;
    float* nxt = base + ((it + 1u) & 1u) * STAGE;
    if (chunk < nchunks) {
      f32x4 fa[4][TM], fb[4][TN];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int m = 0; m < TM; ++m) fa[kk][m] = *(const f32x4*)(As + lds_off(aoff[m], kk * 2 + lh));
#pragma unroll
        for (int n = 0; n < TN; ++n) fb[kk][n] = *(const f32x4*)(Bs + lds_off(boff[n], kk * 2 + lh));
      }
      if (chunk + KG < nchunks) lstore(nxt);          // chunk c+1 -> the other stage
      if (chunk + 2 * KG < nchunks) gload(chunk + 2 * KG);  // chunk c+2 -> registers
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][m][e], fb[kk][n][e], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
  }
}

// k-group reductions (64x64 tile only: one 32x32 accumulator per wave).  Every group parks its 16 partial
// sums per lane in LDS as red[group][reg][thread] (conflict-free: consecutive lanes, consecutive words).
//  * kgroup_reduce_spread: group g then owns registers [g*16/KG, (g+1)*16/KG) of every wave tile and sums them
//    over the groups in group order — the epilogue work is spread over all 4*KG waves.
template <int KG>
__device__ __forceinline__ void kgroup_reduce_spread(const f32x16& acc, float* lds, float (&out)[16 / KG]) {
  constexpr int R = 16 / KG;
  if constexpr (KG == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = acc[i];
  } else {
    const uint32_t tid = threadIdx.x, kg = tid >> 8, ltid = tid & 255u;
    float* red = lds;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(kg * 16 + r) * 256 + ltid] = acc[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const uint32_t r = kg * R + i;
      float sum = red[r * 256 + ltid];
#pragma unroll
      for (int g = 1; g < KG; ++g) sum += red[(g * 16 + r) * 256 + ltid];
      out[i] = sum;
    }
    __syncthreads();
  }
}

// Row of accumulator register r for lane half lh in a 32x32 MFMA tile (C/D layout, cdna_hip_programming.md §3)
__device__ __forceinline__ uint32_t acc_row(uint32_t r, uint32_t lh) { return (r & 3u) + 8u * (r >> 2) + 4u * lh; }

// Running maximum of the present weights (BestFitVoting's max_dist): wave shuffle -> LDS -> ONE atomic per
// workgroup, spread over SA_MAXKEY_SHARDS words.  Atomics on a single word serialise at ~12 ns each on this chip
// (MI355X_MICROARCH.md "fanin"): one per wave on one word cost 45 us at 4096 waves.
__device__ __forceinline__ void block_max_key(uint32_t* shards, uint32_t kmax, float* lds) {
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t ok = __shfl_xor(kmax, o);
    kmax = ok > kmax ? ok : kmax;
  }
  uint32_t* s_k = (uint32_t*)lds;
  __syncthreads();  // every wave is past its last read of the LDS stash
  const uint32_t wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63u) == 0) s_k[wave] = kmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t m = 0;
    for (uint32_t w = 0; w < nw; ++w) m = s_k[w] > m ? s_k[w] : m;
    if (m) atomicMax(&shards[(blockIdx.x + blockIdx.y * gridDim.x) & (SA_MAXKEY_SHARDS - 1)], m);
  }
}

// Everything the reference does per (candidate, observation) pair after the dot product.
__device__ __forceinline__ float visual_cell(const SaParams& p, float dot, float na, bool us, const sa_geo& cg, uint64_t epoch,
                                             const GemmCols& col, uint32_t* kmax) {
  float out = __builtin_nanf("");
  if (us && col.ok && sa_compatible(cg, epoch, col.g, col.epoch, p.max_idle, p.cons)) {
    // divided / (f1_divisor * f2_divisor).sqrt(): v_rsq_f32 + multiply, <= 2 ulp from the reference's sqrt + divide,
    // two orders of magnitude inside the 1e-5 gate and ~25 instructions cheaper per cell
    float d = dot * __frsqrt_rn(na * col.nb);
    if (d >= p.visual_threshold) {  // VisualSortMetricType::is_ok (NaN fails)
      out = 1.0f - d;               // distance_to_weight
      uint32_t key = sa_f32_key(out);
      *kmax = key > *kmax ? key : *kmax;
    }
  }
  return out;
}

template <int BM, int BN, int KG>
__global__ __launch_bounds__(256 * KG) void k_visual_cosine(const SceneDev* __restrict__ scenes, SaParams p) {
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  const uint32_t N = S.N, TK = S.TK, K = S.K;
  const uint32_t m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= N || n0 >= TK) return;
  constexpr int TM = BM / 64, TN = BN / 64;
  static_assert(KG == 1 || (TM == 1 && TN == 1), "k-groups only with the 64x64 tile");
  __shared__ __attribute__((aligned(16))) float lds[KG * 2 * (BM + BN) * BK];
  f32x16 acc[TM][TN];
  gemm_mainloop<BM, BN, KG>(S.c_feat, S.t_feat, N, TK, S.Dp, m0, n0, lds, acc);

  const uint32_t tid = threadIdx.x, lane = tid & 63u, w4 = (tid >> 6) & 3u, kg = tid >> 8;
  const uint32_t wm = w4 >> 1, wn = w4 & 1u, lr = lane & 31u, lh = lane >> 5;
  constexpr int R = 16 / KG;
  float part[TM == 1 && TN == 1 ? R : 1];
  if constexpr (TM == 1 && TN == 1) kgroup_reduce_spread<KG>(acc[0][0], lds, part);

  // ---- fused epilogue: row metadata through LDS, column metadata in registers ----
  float* s_na = lds;                      // [BM]
  float* s_us = lds + BM;                 // [BM] 1.0 / 0.0
  sa_geo* s_g = (sa_geo*)(lds + 2 * BM);  // [BM]
  for (uint32_t r = tid; r < (uint32_t)BM; r += 256 * KG) {
    uint32_t gi = m0 + r;
    bool in = gi < N;
    s_na[r] = in ? S.c_fnorm[gi] : 0.f;
    s_us[r] = (in && S.c_usable[gi]) ? 1.f : 0.f;
    s_g[r] = in ? S.c_geo[gi] : sa_geo{0.f, 0.f, 0.f, 0.f};
  }
  GemmCols col[TN];
#pragma unroll
  for (int n = 0; n < TN; ++n) {
    uint32_t gj = n0 + wn * (BN / 2) + n * 32 + lr;
    col[n].ok = false;
    col[n].nb = 0.f;
    col[n].g = sa_geo{0.f, 0.f, 0.f, 0.f};
    col[n].epoch = 0;
    if (gj < TK) {
      uint32_t t = gj / K;
      col[n].nb = S.t_fnorm[gj];
      col[n].ok = S.t_fpresent[gj] != 0 && S.t_fcount[t] >= p.min_track_len;
      col[n].g = S.t_geo[t];
      col[n].epoch = S.t_epoch[t];
    }
  }
  __syncthreads();
  const uint64_t epoch = S.epoch;
  uint32_t kmax = 0;  // order-preserving key of the largest present weight seen by this lane
  if constexpr (TM == 1 && TN == 1) {
    const uint32_t gj = n0 + wn * 32 + lr;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const uint32_t li = wm * 32 + acc_row(kg * R + i, lh);
      const uint32_t gi = m0 + li;
      if (gi < N && gj < TK)
        S.vis[(size_t)gi * TK + gj] = visual_cell(p, part[i], s_na[li], s_us[li] != 0.f, s_g[li], epoch, col[0], &kmax);
    }
  } else {
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t li = wm * (BM / 2) + m * 32 + acc_row(r, lh);
        const uint32_t gi = m0 + li;
        if (gi >= N) continue;
        const float na = s_na[li];
        const bool us = s_us[li] != 0.f;
        const sa_geo cg = s_g[li];
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          const uint32_t gj = n0 + wn * (BN / 2) + n * 32 + lr;
          if (gj < TK) S.vis[(size_t)gi * TK + gj] = visual_cell(p, acc[m][n][r], na, us, cg, epoch, col[n], &kmax);
        }
      }
  }
  block_max_key(S.vis_max_key, kmax, lds);
}

// Direct sum (a-b)^2: 64x64 outputs per 256-thread block, 4x4 per thread, k staged through the same
// swizzled LDS tiles.  Column c of a thread is tx + 16*c so a wave's stores cover 64-B row segments.
__device__ __forceinline__ void euclid_mainloop(const float* __restrict__ A, const float* __restrict__ B, uint32_t M,
                                                uint32_t Ncols, uint32_t Dp, uint32_t m0, uint32_t n0, float* lds,
                                                float (&acc)[4][4]) {
  constexpr int BM = 64, BN = 64;
  float* As = lds;
  float* Bs = lds + BM * BK;
  const uint32_t tid = threadIdx.x, ty = tid >> 4, tx = tid & 15u;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const float* pa[2];
  const float* pb[2];
  uint32_t so[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    uint32_t c = tid + 256u * r, row = c >> 3, kc = c & 7u;
    uint32_t ga = m0 + row, gb = n0 + row;
    ga = ga < M ? ga : M - 1;
    gb = gb < Ncols ? gb : Ncols - 1;
    pa[r] = A + (size_t)ga * Dp + kc * 4u;
    pb[r] = B + (size_t)gb * Dp + kc * 4u;
    so[r] = lds_off(row, kc);
  }
  f32x4 ra[2], rb[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) { ra[r] = *(const f32x4*)pa[r]; rb[r] = *(const f32x4*)pb[r]; }
  for (uint32_t k0 = 0; k0 < Dp; k0 += BK) {
#pragma unroll
    for (int r = 0; r < 2; ++r) { *(f32x4*)(As + so[r]) = ra[r]; *(f32x4*)(Bs + so[r]) = rb[r]; }
    __syncthreads();
    if (k0 + BK < Dp) {
#pragma unroll
      for (int r = 0; r < 2; ++r) { ra[r] = *(const f32x4*)(pa[r] + k0 + BK); rb[r] = *(const f32x4*)(pb[r] + k0 + BK); }
    }
#pragma unroll
    for (uint32_t kc = 0; kc < 8; ++kc) {
      f32x4 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *(const f32x4*)(As + lds_off(ty * 4 + i, kc));
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *(const f32x4*)(Bs + lds_off(tx + 16 * j, kc));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float df = fa[i][e] - fb[j][e];
            acc[i][j] += df * df;
          }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_visual_euclid(const SceneDev* __restrict__ scenes, SaParams p) {
  constexpr int BM = 64, BN = 64;
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  const uint32_t N = S.N, TK = S.TK, K = S.K;
  const uint32_t m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= N || n0 >= TK) return;
  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * BK];
  float acc[4][4];
  euclid_mainloop(S.c_feat, S.t_feat, N, TK, S.Dp, m0, n0, lds, acc);
  const uint32_t tid = threadIdx.x, ty = tid >> 4, tx = tid & 15u;
  const float nanv = __builtin_nanf("");
  const uint64_t epoch = S.epoch;
  uint32_t kmax = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t gi = m0 + ty * 4 + i;
    if (gi >= N) continue;
    bool us = S.c_usable[gi] != 0;
    sa_geo cg = S.c_geo[gi];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t gj = n0 + tx + 16 * j;
      if (gj >= TK) continue;
      uint32_t t = gj / K;
      float out = nanv;
      if (us && S.t_fpresent[gj] && S.t_fcount[t] >= p.min_track_len &&
          sa_compatible(cg, epoch, S.t_geo[t], S.t_epoch[t], p.max_idle, p.cons)) {
        float d = sqrtf(acc[i][j]);
        if (d <= p.visual_threshold) {
          out = d;
          uint32_t key = sa_f32_key(out);
          kmax = key > kmax ? key : kmax;
        }
      }
      S.vis[(size_t)gi * TK + gj] = out;
    }
  }
  block_max_key(S.vis_max_key, kmax, lds);
}

// ---- standalone distance matrix (sa_feature_distance_matrix): no gating, plain d ----
template <int BM, int BN, int KG>
__global__ __launch_bounds__(256 * KG) void k_cosine_matrix(const float* __restrict__ A, const float* __restrict__ an,
                                                            const float* __restrict__ B, const float* __restrict__ bn,
                                                            uint32_t M, uint32_t Ncols, uint32_t Dp, float* __restrict__ out) {
  const uint32_t m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  constexpr int TM = BM / 64, TN = BN / 64;
  static_assert(KG == 1 || (TM == 1 && TN == 1), "k-groups only with the 64x64 tile");
  __shared__ __attribute__((aligned(16))) float lds[KG * 2 * (BM + BN) * BK];
  f32x16 acc[TM][TN];
  gemm_mainloop<BM, BN, KG>(A, B, M, Ncols, Dp, m0, n0, lds, acc);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w4 = (tid >> 6) & 3u, kg = tid >> 8;
  const uint32_t wm = w4 >> 1, wn = w4 & 1u, lr = lane & 31u, lh = lane >> 5;
  if constexpr (TM == 1 && TN == 1) {
    constexpr int R = 16 / KG;
    float part[R];
    kgroup_reduce_spread<KG>(acc[0][0], lds, part);
    const uint32_t gj = n0 + wn * 32 + lr;
    const float nb = bn[gj < Ncols ? gj : Ncols - 1];
    float na[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      uint32_t gi = m0 + wm * 32 + acc_row(kg * R + i, lh);
      na[i] = an[gi < M ? gi : M - 1];
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      uint32_t gi = m0 + wm * 32 + acc_row(kg * R + i, lh);
      if (gi < M && gj < Ncols) out[(size_t)gi * Ncols + gj] = part[i] / sqrtf(na[i] * nb);
    }
  } else {
    float nb[TN];
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      uint32_t gj = n0 + wn * (BN / 2) + n * 32 + lr;
      nb[n] = bn[gj < Ncols ? gj : Ncols - 1];
    }
    float na[TM][16];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        uint32_t gi = m0 + wm * (BM / 2) + m * 32 + acc_row(r, lh);
        na[m][r] = an[gi < M ? gi : M - 1];
      }
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        uint32_t gi = m0 + wm * (BM / 2) + m * 32 + acc_row(r, lh);
        if (gi >= M) continue;
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          uint32_t gj = n0 + wn * (BN / 2) + n * 32 + lr;
          if (gj < Ncols) out[(size_t)gi * Ncols + gj] = acc[m][n][r] / sqrtf(na[m][r] * nb[n]);
        }
      }
  }
}

__global__ __launch_bounds__(256) void k_euclid_matrix(const float* __restrict__ A, const float* __restrict__ B, uint32_t M,
                                                       uint32_t Ncols, uint32_t Dp, float* __restrict__ out) {
  constexpr int BM = 64, BN = 64;
  const uint32_t m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * BK];
  float acc[4][4];
  euclid_mainloop(A, B, M, Ncols, Dp, m0, n0, lds, acc);
  const uint32_t tid = threadIdx.x, ty = tid >> 4, tx = tid & 15u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t gi = m0 + ty * 4 + i;
    if (gi >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t gj = n0 + tx + 16 * j;
      if (gj < Ncols) out[(size_t)gi * Ncols + gj] = sqrtf(acc[i][j]);
    }
  }
}

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Tile choice by the number of workgroups the frame yields on 256 CUs:
//   128x128 when that alone gives >= 192 workgroups; else 64x64, with the k dimension split over 2 or 4 wave
//   groups inside each workgroup when there are too few workgroups to put more than one wave on every SIMD.
static inline int tile_plan(uint32_t M, uint32_t Ncols, uint32_t ns, uint32_t Dp) {
  if (const char* f = getenv("SA_GEMM_PLAN")) return atoi(f);  // tuning override: 0 = 128x128, 1/2/4 = 64x64 with KG groups
  if ((size_t)cdiv(M, 128) * cdiv(Ncols, 128) * ns >= 192) return 0;
  size_t b64 = (size_t)cdiv(M, 64) * cdiv(Ncols, 64) * ns;
  uint32_t nchunks = Dp / BK;
  if (b64 <= 320 && nchunks >= 8) return 4;
  if (b64 <= 768 && nchunks >= 4) return 2;
  return 1;
}

hipError_t sa_launch_visual(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxTK, const SaParams& p,
                            hipStream_t st) {
  if (!maxN || !maxTK) return hipSuccess;
  if (p.visual_kind == SA_VIS_COSINE) {
    const uint32_t Dp = p.Dp;  // one feature length per engine
    switch (tile_plan(maxN, maxTK, ns, Dp)) {
      case 0: hipLaunchKernelGGL((k_visual_cosine<128, 128, 1>), dim3(cdiv(maxTK, 128), cdiv(maxN, 128), ns), dim3(256), 0, st, scenes, p); break;
      case 4: hipLaunchKernelGGL((k_visual_cosine<64, 64, 4>), dim3(cdiv(maxTK, 64), cdiv(maxN, 64), ns), dim3(1024), 0, st, scenes, p); break;
      case 2: hipLaunchKernelGGL((k_visual_cosine<64, 64, 2>), dim3(cdiv(maxTK, 64), cdiv(maxN, 64), ns), dim3(512), 0, st, scenes, p); break;
      default: hipLaunchKernelGGL((k_visual_cosine<64, 64, 1>), dim3(cdiv(maxTK, 64), cdiv(maxN, 64), ns), dim3(256), 0, st, scenes, p); break;
    }
  } else {
    hipLaunchKernelGGL(k_visual_euclid, dim3(cdiv(maxTK, 64), cdiv(maxN, 64), ns), dim3(256), 0, st, scenes, p);
  }
  return hipGetLastError();
}

hipError_t sa_launch_distance_matrix(int kind, const float* a, const float* an, const float* b, const float* bn,
                                     uint32_t n, uint32_t t, uint32_t dp, float* out, hipStream_t st) {
  if (!n || !t) return hipSuccess;
  if (kind == SA_VIS_COSINE) {
    switch (tile_plan(n, t, 1, dp)) {
      case 0: hipLaunchKernelGGL((k_cosine_matrix<128, 128, 1>), dim3(cdiv(t, 128), cdiv(n, 128)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 4: hipLaunchKernelGGL((k_cosine_matrix<64, 64, 4>), dim3(cdiv(t, 64), cdiv(n, 64)), dim3(1024), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 2: hipLaunchKernelGGL((k_cosine_matrix<64, 64, 2>), dim3(cdiv(t, 64), cdiv(n, 64)), dim3(512), 0, st, a, an, b, bn, n, t, dp, out); break;
      default: hipLaunchKernelGGL((k_cosine_matrix<64, 64, 1>), dim3(cdiv(t, 64), cdiv(n, 64)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
    }
  } else {
    hipLaunchKernelGGL(k_euclid_matrix, dim3(cdiv(t, 64), cdiv(n, 64)), dim3(256), 0, st, a, b, n, t, dp, out);
  }
  return hipGetLastError();
}
