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
#include "sa_frame.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Pointers that reach a kernel through the SceneDev descriptor are generic ("flat") to the compiler: their loads become
// flat_load, which counts on BOTH vmcnt and lgkmcnt — every s_waitcnt lgkmcnt for an LDS fragment then also drains the
// feature loads in flight and the software pipeline collapses (C2: 20 -> 41 us).  The feature operands are therefore
// re-typed as global-address-space pointers before the main loop.
#define SA_AS1 __attribute__((address_space(1)))
typedef const SA_AS1 float* gfloat_p;
typedef const SA_AS1 f32x4* gf32x4_p;

#define BK 32

// In-kernel timeline (build with -DSA_GEMM_TRACE, run with SA_GEMM_TRACE=<launch #> to dump gpurun_out/gemm_trace.txt):
// s_memtime stamps per workgroup at  0 entry | 1 prologue done | 2 main loop done | 3 k-group reduction done |
// 4 epilogue stores issued | 5 exit.  Compiled out of the product.
#ifdef SA_GEMM_TRACE
#define SA_STAMP(tr, i) do { if ((tr) && threadIdx.x == 0) (tr)[i] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ uint64_t* g_trace_dev;
#define SA_TRACE_PTR() (g_trace_dev ? g_trace_dev + 8 * (blockIdx.x + blockIdx.y * gridDim.x) : nullptr)
#else
#define SA_STAMP(tr, i) do { } while (0)
#define SA_TRACE_PTR() nullptr
#endif

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// XCD-aware tile order.  Workgroups of a launch go round-robin over the 8 XCDs (workgroup b -> XCD b % 8), each with an L2 of its own:
// with tiles numbered row by row, every XCD's L2 ends up pulling the WHOLE candidate panel and its share of the track panel out of
// HBM (C2: 19.9 MB measured for 4.1 MB of operands).  Instead XCD x takes the x-th CONTIGUOUS chunk of a tile numbering that walks
// the grid in bands of W tile columns (row by row inside a band): its tiles form a compact block of about W x chunk / W tiles that
// shares W column panels and chunk / W row panels (C2, 16 x 16 tiles, chunk 32, W 4: 8 x 4 tiles = 1.5 MB per XCD instead of 2.25).
// MEASURED (round 4, SA_FLAG_XCD_TILES; profiles/r04_*): FETCH_SIZE of the C2 launch 19.1 -> 15.3 MB, the frame no faster (20.42 against
// 20.48 us; the launch itself 17.6 against 16.5 us between its own timestamps), c2b's stand-alone contraction 92.3 -> 91.0 us, C5's
// 634 -> 657 us (bands of 12 x 128 tracks x 16 KB overflow the 4 MB L2 that a row-by-row walk keeps one candidate panel in): the HBM
// traffic is not what bounds these launches (1.1 TB/s at C5).  Since round 4 the FUSED first phase numbers its contraction tiles this way
// by default (the same speed on every workload that takes it, a fifth fewer L2 fills; SA_FLAG_ROW_TILES: row by row); the stand-alone
// contraction stays row by row (SA_FLAG_XCD_TILES: this order there too).
// b in [0, 8 chunk) -> tile number t = (b % 8) chunk + b / 8 (t >= tiles: an idle workgroup); t -> (row, column).
struct XcdOrder { uint32_t chunk, W; };
static inline XcdOrder xcd_order(uint32_t gx, uint32_t gy, bool row_major = false) {
  XcdOrder o;
  const uint32_t tiles = gx * gy;
  if (row_major || tiles >= (1u << 24)) { o.chunk = tiles; o.W = 0; return o; }  // the default: W = 0 -> workgroup b is tile b, row by row
  o.chunk = (tiles + 7u) / 8u;
  uint32_t w = 1;
  while ((w + 1) * (w + 1) <= o.chunk) ++w;   // ~ sqrt(chunk): square-ish blocks
  o.W = w < gx ? w : gx;
  if (o.W == 0) o.W = 1;
  // (the kernels take the pair packed as chunk << 8 | W: any band width is a valid one — 255 at most; a grid beyond 2^24 chunks — half
  // a million tracks against as many detections — falls back to the row-by-row numbering, which the launchers handle with W = 0)
  if (o.W > 255u) o.W = 255u;
  if (o.chunk >= (1u << 24)) { o.chunk = tiles; o.W = 0; }
  return o;
}
__device__ __forceinline__ bool xcd_tile(uint32_t b, uint32_t gx, uint32_t gy, uint32_t chunk, uint32_t W, uint32_t* bx, uint32_t* by) {
  if (W == 0) {  // row-major numbering (chunk = tiles)
    if (b >= gx * gy) return false;
    *bx = b % gx; *by = b / gx;
    return true;
  }
  const uint32_t t = (b & 7u) * chunk + (b >> 3);
  if (t >= gx * gy || (b >> 3) >= chunk) return false;
  const uint32_t band = t / (W * gy), t2 = t - band * W * gy;
  const uint32_t wb = gx - band * W < W ? gx - band * W : W;
  *by = t2 / wb;
  *bx = band * W + t2 % wb;
  return true;
}

// Physical float offset of logical 16-byte chunk `kc` (0..7) of row `row` in a [rows][32] f32 LDS tile.
// XOR with (row>>1)&7: with a 128-B row stride, the 16 rows one ds_read_b128 lane group touches land on
// 16 distinct 16-B slots of the 256-B bank row (conflict-free); see MI355X_MICROARCH.md §LDS.
__device__ __forceinline__ uint32_t lds_off(uint32_t row, uint32_t kc) { return row * BK + ((kc ^ ((row >> 1) & 7u)) << 2); }

struct GemmCols {   // per-lane column metadata kept in registers through the epilogue
  float nb;         // squared norm of the stored feature
  bool ok;          // feature present, track long enough, epoch distance within max_idle_epochs
  float cmax;       // max_dist of the spatio-temporal constraint that applies to this column's epoch distance; < 0 = none
  sa_geo g;
};

// Dp is a multiple of 32, so a 32-float chunk is either entirely inside a row or absent; rows past the
// matrix edge are clamped to the last row (their results are never stored) — no branches around the loads.
//
// Pipeline per k-group (256 threads, 2 LDS stages of (BM+BN) x 32 floats), ONE barrier per 32-deep chunk.
// f32 MFMA is slow (64 cycles per 32x32x2 on a SIMD), so LDS and L2 bandwidth are never the limit; what costs is every
// cycle a wave spends issuing something else while its matrix pipe is free.  Iteration c is therefore laid out as four
// k-steps (8 floats of k each), each one a straight run of 4*TM*TN MFMAs with the other work of the iteration placed
// in their shadow (an MFMA takes 4 cycles to issue and 64 to execute; the wave keeps issuing meanwhile):
//     k-step kk:  MFMAs on fragment buffer kk&1
//                 | ds_read the fragments of k-step kk+1 into the other fragment buffer
//                 | a quarter of: ds_write chunk c+1 (already in registers) to the other LDS stage, then re-issue those
//                   registers' global loads for chunk c+2
// Only the fragment read of k-step 0 (right after the barrier) is exposed; the co-resident workgroup covers it.
// The body is branch-free (three specialisations: steady state / last-but-one chunk / last chunk) so that the whole
// iteration is ONE scheduling region, pinned with sched_group_barrier; measured against the same loop with the loads and
// stores hoisted to the top of the iteration (what the compiler does on its own): 110 -> see profiles/.
// NORM: also accumulate, per lane, the sum of squares of the A fragments it multiplies (its half of the k values of row
// wm*32 + lr, this k-group's chunks) into nsq[0] — the raw-feature mode of the fused frame launch has no pre-computed norms.
template <int BM, int BN, int KG, bool NORM = false>
__device__ __forceinline__ void gemm_mainloop(gfloat_p A, gfloat_p B, uint32_t M,
                                              uint32_t Ncols, uint32_t Dp, uint32_t m0, uint32_t n0, float* lds,
                                              f32x16 (&acc)[BM / 64][BN / 64], uint64_t* tr = nullptr, float* nsq = nullptr) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;  // 16-B chunks per thread per stage
  constexpr int L_CH = A_CH + B_CH;                          // 4 (64x64) .. 8 (128x128): a multiple of 4 or exactly 6
  constexpr int STAGE = (BM + BN) * BK;                     // floats per LDS stage
  const uint32_t tid = threadIdx.x;
  // k-group of this wave (0 when KG == 1).  readfirstlane: the value is wave-uniform, and telling the compiler so keeps
  // every branch of the main loop scalar — with a divergent-looking branch it parks the 16-64 accumulator registers
  // in VGPRs and copies them to and from the matrix-core register file on EVERY chunk (128 v_accvgpr moves + a
  // pipeline drain per 64 MFMAs in the 128x128 kernel).
  const uint32_t kg = KG == 1 ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 8));
  const uint32_t ltid = tid & 255u, lane = tid & 63u, w4 = (tid >> 6) & 3u;
  float* base = lds + kg * 2 * STAGE;
  const uint32_t wm = w4 >> 1, wn = w4 & 1u;
  const uint32_t lr = lane & 31u, lh = lane >> 5;
  const uint32_t nchunks = Dp / BK;
  const uint32_t niter = (nchunks + KG - 1) / KG;
  // entries 0..A_CH-1 belong to A, the rest to B
  uint32_t goff[L_CH], soff[L_CH];
#pragma unroll
  for (int r = 0; r < L_CH; ++r) {
    const bool isA = r < A_CH;
    uint32_t c = ltid + 256u * (isA ? r : r - A_CH), row = c >> 3, kc = c & 7u;
    uint32_t gr = (isA ? m0 : n0) + row;
    const uint32_t lim = isA ? M : Ncols;
    gr = gr < lim ? gr : lim - 1;
    goff[r] = gr * Dp + kc * 4u;
    soff[r] = (isA ? 0 : BM * BK) + lds_off(row, kc);
  }
  // Two register sets: the rows of chunk i+1 are written to LDS from set (i+1)&1, which is then reloaded with chunk i+3 —
  // whatever place inside the iteration the scheduler gives a load, it has at least one whole iteration to land (with
  // one set, loads that sink to the end of an iteration are consumed 5 MFMAs later: a full L2 round trip exposed per
  // chunk, 28 us instead of 15 at C2).
  f32x4 rg[2][L_CH];
  auto gload_set = [&](int set, uint32_t k0) {
#pragma unroll
    for (int r = 0; r < L_CH; ++r) rg[set][r] = *(gf32x4_p)((r < A_CH ? A : B) + (size_t)(goff[r] + k0));
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

  // chunks of this group: local index i <-> chunk i*KG + kg, i < mine
  const uint32_t mine = kg < nchunks ? (nchunks - kg + KG - 1) / KG : 0u;
  if (mine > 0) {
    gload_set(0, kg * BK);
#pragma unroll
    for (int r = 0; r < L_CH; ++r) *(f32x4*)(base + soff[r]) = rg[0][r];
  }
  if (mine > 1) gload_set(1, (kg + KG) * BK);
  if (mine > 2) gload_set(0, (kg + 2 * KG) * BK);
  __syncthreads();
  SA_STAMP(tr, 1);

  // One iteration.  SET = (i+1)&1; STORE: chunk i+1 exists (set -> other LDS stage); LOAD: chunk i+3 exists (global -> set).
  // The iteration's ONE barrier sits between its third and fourth k-step: the rows of chunk i+1 go to the other LDS stage during
  // k-steps 0-2 (which also fetch the fragments of k-steps 1-3 from this stage), every wave then meets at the barrier with the
  // MFMAs of k-step 2 still queued on its matrix pipe, and k-step 3 — running on fragments already in registers — fetches the
  // FIRST fragments of chunk i+1 from the stage that has just become visible.  With the barrier at the end of the iteration
  // that first fetch came after it, its LDS latency exposed on an idle pipe once per chunk (one wave per SIMD: 1560 cycles per
  // chunk for 1024 of MFMA work).
  f32x4 fa[2][TM], fb[2][TN];
  if (mine > 0) {
#pragma unroll
    for (int m = 0; m < TM; ++m) fa[0][m] = *(const f32x4*)(base + lds_off(aoff[m], lh));
#pragma unroll
    for (int n = 0; n < TN; ++n) fb[0][n] = *(const f32x4*)(base + BM * BK + lds_off(boff[n], lh));
  }
  auto body = [&](uint32_t it, auto set_tag, auto store_tag, auto load_tag) {
    constexpr int SET = decltype(set_tag)::value;
    constexpr bool STORE = decltype(store_tag)::value, LOAD = decltype(load_tag)::value;
    const float* As = base + (it & 1u) * STAGE;
    const float* Bs = As + BM * BK;
    float* nxt = base + ((it + 1u) & 1u) * STAGE;
    const uint32_t k3 = ((it + 3u) * KG + kg) * BK;
    auto kstep = [&](auto kk_tag) {
      constexpr int kk = decltype(kk_tag)::value;
      constexpr int cur = kk & 1, nx = cur ^ 1;
      if constexpr (kk < 3) {
#pragma unroll
        for (int m = 0; m < TM; ++m) fa[nx][m] = *(const f32x4*)(As + lds_off(aoff[m], (kk + 1) * 2 + lh));
#pragma unroll
        for (int n = 0; n < TN; ++n) fb[nx][n] = *(const f32x4*)(Bs + lds_off(boff[n], (kk + 1) * 2 + lh));
      } else if constexpr (STORE) {
#pragma unroll
        for (int m = 0; m < TM; ++m) fa[nx][m] = *(const f32x4*)(nxt + lds_off(aoff[m], lh));
#pragma unroll
        for (int n = 0; n < TN; ++n) fb[nx][n] = *(const f32x4*)(nxt + BM * BK + lds_off(boff[n], lh));
      }
      // this k-step's share of the register -> LDS -> register hand-over of the staged rows: the LDS stores in k-steps 0-2
      // (they must be visible at the barrier), each register's reload from global memory right after its store
      constexpr int lo_of[5] = {0, (L_CH + 2) / 3, (2 * L_CH + 2) / 3, L_CH, L_CH};
      constexpr int lo = lo_of[kk], hi = lo_of[kk + 1];
#pragma unroll
      for (int r = lo; r < hi; ++r) {
        if constexpr (STORE) *(f32x4*)(nxt + soff[r]) = rg[SET][r];
        if constexpr (LOAD) rg[SET][r] = *(gf32x4_p)((r < A_CH ? A : B) + (size_t)(goff[r] + k3));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m][e], fb[cur][n][e], acc[m][n], 0, 0, 0);
      if constexpr (NORM) {
#pragma unroll
        for (int e = 0; e < 4; ++e) nsq[0] += fa[cur][0][e] * fa[cur][0][e];
      }
      // pin the interleave: one side instruction in the shadow of each of the first MFMAs of the k-step
      constexpr int NM = 4 * TM * TN;
      constexpr int nrd = (kk < 3 || STORE) ? TM + TN : 0, nst = STORE ? hi - lo : 0, nld = LOAD ? hi - lo : 0;
      constexpr int used = nrd + nst + nld;
#pragma unroll
      for (int i = 0; i < nrd; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
      for (int i = 0; i < (nst > nld ? nst : nld); ++i) {
        if (i < nst) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
        if (i < nld) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
      }
      if constexpr (used < NM) __builtin_amdgcn_sched_group_barrier(0x008, NM - used, 0);
    };
    kstep(std::integral_constant<int, 0>{});
    kstep(std::integral_constant<int, 1>{});
    kstep(std::integral_constant<int, 2>{});
    __syncthreads();
    kstep(std::integral_constant<int, 3>{});
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  uint32_t it = 0;
  // steady state two iterations per trip, so that the register set is a compile-time constant (set of iteration i = (i+1)&1)
  for (; it + 4 < mine; it += 2) {
    body(it, S1{}, T_{}, T_{});
    body(it + 1, S0{}, T_{}, T_{});
  }
  for (; it < mine; ++it) {  // at most 4 iterations left; `it` is even on entry
    const bool st = it + 1 < mine, ld = it + 3 < mine;
    if (it & 1u) {
      if (ld) body(it, S0{}, T_{}, T_{});
      else if (st) body(it, S0{}, T_{}, F_{});
      else body(it, S0{}, F_{}, F_{});
    } else {
      if (ld) body(it, S1{}, T_{}, T_{});
      else if (st) body(it, S1{}, T_{}, F_{});
      else body(it, S1{}, F_{}, F_{});
    }
  }
  for (; it < niter; ++it) __syncthreads();  // a group that ran out of chunks still meets the others at the barrier
  SA_STAMP(tr, 2);
}

// Ring variant (KG == 0 in the kernel templates): ONE wave per SIMD, a 3-stage LDS ring, nothing left for a partner wave
// to cover.  Used where a frame yields about one workgroup per CU (C2: 16 x 16 tiles of 64x64 on 256 CUs), so that the
// k-group trick above would put the two waves of a SIMD behind the SAME barrier — they then stall together (measured:
// 2583 cycles per iteration against 2048 of MFMA work).  Here a wave's own instruction stream keeps its matrix pipe fed:
//     iteration c:  MFMAs of chunk c from stage c%3, fragments double-buffered in registers
//                   | ds_write chunk c+2 (registers, loaded two iterations ago) -> stage (c+2)%3
//                   | re-issue those registers' global loads for chunk c+4
//                   | during the LAST k-step: ds_read the first fragments of chunk c+1 (stage (c+1)%3, complete since the
//                     previous barrier) — so nothing but the barrier itself sits between two chunks' MFMAs.
// Hazards: stage (c+2)%3 was last read in iteration c-1, and every wave has passed that iteration's barrier.
template <int BM, int BN>
__device__ __forceinline__ void gemm_mainloop_ring(gfloat_p A, gfloat_p B, uint32_t M,
                                                   uint32_t Ncols, uint32_t Dp, uint32_t m0, uint32_t n0, float* lds,
                                                   f32x16 (&acc)[BM / 64][BN / 64], uint64_t* tr = nullptr) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256, L_CH = A_CH + B_CH;
  constexpr int STAGE = (BM + BN) * BK;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w4 = tid >> 6;
  const uint32_t wm = w4 >> 1, wn = w4 & 1u, lr = lane & 31u, lh = lane >> 5;
  const uint32_t nchunks = Dp / BK;
  uint32_t goff[L_CH], soff[L_CH];
#pragma unroll
  for (int r = 0; r < L_CH; ++r) {
    const bool isA = r < A_CH;
    uint32_t c = tid + 256u * (isA ? r : r - A_CH), row = c >> 3, kc = c & 7u;
    uint32_t gr = (isA ? m0 : n0) + row;
    const uint32_t lim = isA ? M : Ncols;
    gr = gr < lim ? gr : lim - 1;
    goff[r] = gr * Dp + kc * 4u;
    soff[r] = (isA ? 0 : BM * BK) + lds_off(row, kc);
  }
  f32x4 rg[2][L_CH];
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

  // prologue: chunks 0 and 1 -> stages 0 and 1; chunks 2 and 3 -> register sets 0 and 1
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if ((uint32_t)c < nchunks) {
#pragma unroll
      for (int r = 0; r < L_CH; ++r) rg[c & 1][r] = *(gf32x4_p)((r < A_CH ? A : B) + (size_t)(goff[r] + c * BK));
      if (c < 2) {
#pragma unroll
        for (int r = 0; r < L_CH; ++r) *(f32x4*)(lds + c * STAGE + soff[r]) = rg[c & 1][r];
      }
    }
  __syncthreads();
  SA_STAMP(tr, 1);
  f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m) fa[0][m] = *(const f32x4*)(lds + lds_off(aoff[m], lh));
#pragma unroll
  for (int n = 0; n < TN; ++n) fb[0][n] = *(const f32x4*)(lds + BM * BK + lds_off(boff[n], lh));

  // PAR: parity of the iteration (selects the register set); STORE: chunk c+2 exists; LOAD: chunk c+4 exists;
  // NEXT: chunk c+1 exists (prefetch its first fragments)
  auto body = [&](uint32_t c, uint32_t st, auto par_tag, auto store_tag, auto load_tag, auto next_tag) {
    constexpr int PAR = decltype(par_tag)::value;
    constexpr bool STORE = decltype(store_tag)::value, LOAD = decltype(load_tag)::value, NEXT = decltype(next_tag)::value;
    const uint32_t st1 = st == 2 ? 0u : st + 1u, st2 = st == 0 ? 2u : st - 1u;  // (c+1)%3, (c+2)%3
    const float* As = lds + st * STAGE;
    const float* Bs = As + BM * BK;
    const float* An = lds + st1 * STAGE;
    const float* Bn = An + BM * BK;
    float* wr = lds + st2 * STAGE;
    const uint32_t k4 = (c + 4u) * BK;
    auto kstep = [&](auto kk_tag) {
      constexpr int kk = decltype(kk_tag)::value;
      constexpr int cur = kk & 1, nx = cur ^ 1;
      if constexpr (kk < 3) {
#pragma unroll
        for (int m = 0; m < TM; ++m) fa[nx][m] = *(const f32x4*)(As + lds_off(aoff[m], (kk + 1) * 2 + lh));
#pragma unroll
        for (int n = 0; n < TN; ++n) fb[nx][n] = *(const f32x4*)(Bs + lds_off(boff[n], (kk + 1) * 2 + lh));
      } else if constexpr (NEXT) {
#pragma unroll
        for (int m = 0; m < TM; ++m) fa[nx][m] = *(const f32x4*)(An + lds_off(aoff[m], lh));
#pragma unroll
        for (int n = 0; n < TN; ++n) fb[nx][n] = *(const f32x4*)(Bn + lds_off(boff[n], lh));
      }
      constexpr int lo_of[5] = {0, (L_CH + 3) / 4, (L_CH + 1) / 2, (3 * L_CH + 3) / 4, L_CH};
      constexpr int lo = lo_of[kk], hi = lo_of[kk + 1];
#pragma unroll
      for (int r = lo; r < hi; ++r) {
        if constexpr (STORE) *(f32x4*)(wr + soff[r]) = rg[PAR][r];
        if constexpr (LOAD) rg[PAR][r] = *(gf32x4_p)((r < A_CH ? A : B) + (size_t)(goff[r] + k4));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m][e], fb[cur][n][e], acc[m][n], 0, 0, 0);
      constexpr int NM = 4 * TM * TN;
      constexpr int nrd = (kk < 3 || NEXT) ? TM + TN : 0, nst = STORE ? hi - lo : 0, nld = LOAD ? hi - lo : 0;
      constexpr int used = nrd + nst + nld;
#pragma unroll
      for (int i = 0; i < nrd; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
      for (int i = 0; i < (nst > nld ? nst : nld); ++i) {
        if (i < nst) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
        if (i < nld) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
      }
      if constexpr (used < NM) __builtin_amdgcn_sched_group_barrier(0x008, NM - used, 0);
    };
    kstep(std::integral_constant<int, 0>{});
    kstep(std::integral_constant<int, 1>{});
    kstep(std::integral_constant<int, 2>{});
    kstep(std::integral_constant<int, 3>{});
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  uint32_t c = 0, st = 0;
  auto adv = [&]() { ++c; st = st == 2 ? 0u : st + 1u; __syncthreads(); };
  // steady state, two iterations per trip so that the register-set parity is static
  while (c + 5 < nchunks) { body(c, st, P0{}, T_{}, T_{}, T_{}); adv(); body(c, st, P1{}, T_{}, T_{}, T_{}); adv(); }
  // c is even here.  Remaining chunks: at most 5.
  if (c + 4 < nchunks) { body(c, st, P0{}, T_{}, T_{}, T_{}); adv(); }          // c+4 exists
  // from here no more loads; parity alternates from (c & 1)
  while (c < nchunks) {
    const bool store = c + 2 < nchunks, next = c + 1 < nchunks;
    if (c & 1u) {
      if (store) body(c, st, P1{}, T_{}, F_{}, T_{});
      else if (next) body(c, st, P1{}, F_{}, F_{}, T_{});
      else body(c, st, P1{}, F_{}, F_{}, F_{});
    } else {
      if (store) body(c, st, P0{}, T_{}, F_{}, T_{});
      else if (next) body(c, st, P0{}, F_{}, F_{}, T_{});
      else body(c, st, P0{}, F_{}, F_{}, F_{});
    }
    adv();
  }
  SA_STAMP(tr, 2);
}

// (fragment order: sa_engine.h, sa_frag_index)
__global__ void k_frag_reorder(const float* __restrict__ src, uint32_t rows, uint32_t Dp, float* __restrict__ dst) {
  // one thread per 16-byte piece of the destination; rows past the end repeat the last row (never stored by the consumers)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)((rows + 31u) / 32u * 32u) * Dp / 4u;
  if (i >= total) return;
  const uint32_t lane = (uint32_t)(i & 63u);
  const size_t blk = i >> 6;
  const uint32_t kb = (uint32_t)(blk % (Dp >> 3)), rb = (uint32_t)(blk / (Dp >> 3));
  uint32_t r = rb * 32u + (lane & 31u);
  r = r < rows ? r : rows - 1;
  ((f32x4*)dst)[i] = *(const f32x4*)(src + (size_t)r * Dp + kb * 8u + (lane >> 5) * 4u);
}
hipError_t sa_launch_frag_reorder(const float* src, uint32_t rows, uint32_t dp, float* dst, hipStream_t st) {
  const size_t total = (size_t)((rows + 31u) / 32u * 32u) * dp / 4u;
  hipLaunchKernelGGL(k_frag_reorder, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, src, rows, dp, dst);
  return hipGetLastError();
}

template <int I, int N, typename F>
__device__ __forceinline__ void sa_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sa_static_for<I + 1, N>(f); }
}

// k-split main loop of the 64 x 64 tile (KS): NO LDS stage and NO barrier between the first load and the last matrix instruction.
// A lone wave per SIMD pays ~25 matrix-pipe cycles for every LDS or memory instruction it issues (scripts/micro/mfma_side_mix.hip);
// the staged loop above issues 16 of them per 16 matrix instructions (8 ds_read_b128, 4 ds_write_b128, 4 global loads).  Here wave
// (wm, kg) owns tile rows wm 32 .. +31, ALL 64 tile columns, and every second 8-deep k-step (kg = its parity): a 32 x 64 wave tile of
// two independent accumulators whose operands come straight from memory into registers — per k-step one dwordx4 of A (its 32 rows,
// row-major: 32 B out of 32 lines) and two of B (row-major likewise, or BFRAG: the fragment-order copy of the bank, one contiguous
// kilobyte each) for 8 matrix instructions: 6 side instructions per 16 instead of 16.  NBUF register buffers: the loads of step
// j + NBUF - 1 are issued under the matrix instructions of step j.  The two k-halves of a 32 x 32 quadrant meet through LDS once, at
// the end (each wave ships the quadrant it does not keep: 4 KB): wave (wm, x) leaves with rows wm 32.., columns x 32.. in the
// standard 32 x 32 accumulator layout — what the epilogues of this file expect from wave (wm, wn = x).
// NORM: *nsq = the squared norm of row wm 32 + lr of A (both lane halves, both k-halves), complete.
template <int NBUF, bool NORM, bool BFRAG, bool AFRAG = false>
__device__ __forceinline__ void gemm_mainloop_ks(gfloat_p A, gfloat_p B, uint32_t M, uint32_t Ncols, uint32_t Dp, uint32_t m0,
                                                 uint32_t n0, float* lds, f32x16& out, uint64_t* tr = nullptr, float* nsq = nullptr,
                                                 uint32_t yield_every = 0) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) & 3u;
  const uint32_t wm = w4 >> 1, kg = w4 & 1u, lr = lane & 31u, lh = lane >> 5;
  // Buffer loads: a resource descriptor per operand (base = the tile's first row, range = what is left of the matrix behind it), a
  // 32-bit byte offset per lane (constant), and the position along k in the instruction's SCALAR offset — no vector arithmetic and no
  // 64-bit address in the loop, and rows past the matrix edge need no clamp: they are out of range and read as zero.
  const uint32_t Mp = AFRAG ? (M + 31u) / 32u * 32u : M, Np = BFRAG ? (Ncols + 31u) / 32u * 32u : Ncols;
  const uint32_t nb0 = BFRAG ? (n0 & ~31u) : n0;
  auto left = [&](uint32_t rows) { const uint64_t b = (uint64_t)rows * Dp * 4u; return (uint32_t)(b < 0xffffffffull ? b : 0xffffffffull); };
  const __amdgpu_buffer_rsrc_t RA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * Dp), 0, (int)left(Mp - m0), 0x00020000);
  const __amdgpu_buffer_rsrc_t RB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)nb0 * Dp), 0, (int)left(Np - nb0), 0x00020000);
  const uint32_t oa = AFRAG ? (wm * 32u * Dp + lane * 4u) * 4u : ((wm * 32u + lr) * Dp + lh * 4u) * 4u;  // (m0 is a multiple of 64)
  uint32_t ob[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    // (fragment order: the tile's first column need not start a block of 32 rows — the whole-track tiles of deeper banks start anywhere;
    // a wave-load then covers two runs of a kilobyte instead of one)
    const uint32_t rr = (n0 - nb0) + n * 32u + lr;
    ob[n] = BFRAG ? ((rr >> 5) * (Dp >> 3) * 256u + lh * 128u + (rr & 31u) * 4u) * 4u : ((n * 32u + lr) * Dp + lh * 4u) * 4u;
  }
  const uint32_t mine = Dp >> 4;  // k-steps of this wave: global step s = 2 j + kg (Dp is a multiple of 32)
  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
  f32x4 fa[NBUF], fb[NBUF][2];
  float ns = 0.f;
  auto load = [&](auto buf_tag, uint32_t j) {  // (a step past the end reads the next rows' bytes or zeros: never multiplied)
    constexpr int buf = decltype(buf_tag)::value;
    const uint32_t s = 2u * j + kg;
    fa[buf] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RA, (int)oa, (int)(s * (AFRAG ? 1024u : 32u)), 0));
#pragma unroll
    for (int n = 0; n < 2; ++n)
      fb[buf][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RB, (int)ob[n], (int)(s * (BFRAG ? 1024u : 32u)), 0));
  };
  auto compute = [&](auto buf_tag) {
    constexpr int buf = decltype(buf_tag)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][e], fb[buf][0][e], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][e], fb[buf][1][e], acc1, 0, 0, 0);
    }
    if constexpr (NORM) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ns += fa[buf][e] * fa[buf][e];
    }
  };
  auto step = [&](auto b_tag, uint32_t j) {  // step j on buffer b; the loads of step j + NBUF - 1 go to the buffer step j - 1 has just left
    constexpr int b = decltype(b_tag)::value;
    load(std::integral_constant<int, (b + NBUF - 1) % NBUF>{}, j + NBUF - 1);
    compute(b_tag);
    // three loads spread over the step: a matrix instruction has to be presented well before the pipe is free for it, so only a few
    // cycles of other instructions fit between two of them for nothing
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    // (the fused launch: the waves of the positional tiles on this SIMD issue nothing while this wave presents matrix instructions)
    if (yield_every && (j % yield_every) == 0) __builtin_amdgcn_s_sleep(1);
  };
  // prologue: steps 0 .. NBUF-2 in flight
  // (in STEP order, pinned: the wait in front of a step's first matrix instruction counts the loads issued after that step's own, and
  // the loop header's count is the smaller of the two ways in — a prologue that issues buffer 0 last would drain every trip)
  sa_static_for<0, NBUF - 1>([&](auto b) { load(b, (uint32_t)decltype(b)::value); __builtin_amdgcn_sched_barrier(0); });
  SA_STAMP(tr, 1);
  uint32_t j = 0;
  for (; j + NBUF <= mine; j += NBUF) sa_static_for<0, NBUF>([&](auto b) { step(b, j + decltype(b)::value); });
  const uint32_t rem = mine - j;
  sa_static_for<0, NBUF>([&](auto b) { if ((uint32_t)decltype(b)::value < rem) step(b, j + decltype(b)::value); });
  SA_STAMP(tr, 2);
  // the two k-halves of every quadrant meet: wave (wm, x) keeps columns x 32 .. and ships the other accumulator to wave (wm, 1 - x)
  f32x4* red = (f32x4*)lds;                 // [4 waves][4][64 lanes] f32x4 = 16 KB
  float* rn = lds + 4 * 4 * 64 * 4;         // [4 waves][32] squared-norm halves
  if (kg == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) red[(w4 * 4 + g) * 64 + lane] = f32x4{acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) red[(w4 * 4 + g) * 64 + lane] = f32x4{acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
  }
  if constexpr (NORM) {
    ns += __shfl_xor(ns, 32);
    if (lh == 0) rn[w4 * 32 + lr] = ns;
  }
  __syncthreads();
  const uint32_t pw = w4 ^ 1u;
  // (a + b is the same f32 whichever wave adds: both orders give the k-half 0 + k-half 1 sum)
  if (kg == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 o = red[(pw * 4 + g) * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) out[4 * g + c] = acc0[4 * g + c] + o[c];
    }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 o = red[(pw * 4 + g) * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) out[4 * g + c] = o[c] + acc1[4 * g + c];
    }
  }
  if constexpr (NORM) *nsq = rn[(wm * 2) * 32 + lr] + rn[(wm * 2 + 1) * 32 + lr];
  __syncthreads();  // the epilogue reuses this LDS
}

// The k-split loop for a 64 x 96 tile (the fused first phase of frames whose 64 x 64 tiles would be one and a half rounds of the chip:
// sa_launch_frame_visual): wave (wm, kg) owns tile rows wm 32 .. +31, ALL 96 columns (three accumulators) and every second 8-deep k-step —
// one gather of A and three fragment-order loads of B per 12 matrix instructions.  Wave (wm, x) leaves with columns x 32 .. +31 in the
// standard 32 x 32 accumulator (`out`: what wave (wm, wn = x) of a 64 x 64 tile holds) and with HALF of the third column block: accumulator
// registers 8 x .. 8 x + 7 of columns 64 .. 95 (`out2`; register r <-> tile row wm 32 + acc_row(r, lane half)).  Every cell is
// (k-half 0) + (k-half 1), each half accumulated over the same k-steps in the same order as gemm_mainloop_ks: the 64 x 64 tiling's bits.
// LDS: [4 waves][6][64] f32x4 of exchange (24 KB) + [4][32] squared-norm halves.
constexpr uint32_t SA_KS96_RED = 4u * 6u * 64u * 4u;
constexpr uint32_t SA_KS96_LDS = SA_KS96_RED + 4u * 32u;
template <int NBUF, bool NORM>
__device__ __forceinline__ void gemm_mainloop_ks96(gfloat_p A, gfloat_p B, uint32_t M, uint32_t Ncols, uint32_t Dp, uint32_t m0, uint32_t n0,
                                                   float* lds, f32x16& out, float (&out2)[8], uint64_t* tr = nullptr, float* nsq = nullptr,
                                                   uint32_t yield_every = 0) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) & 3u;
  const uint32_t wm = w4 >> 1, kg = w4 & 1u, lr = lane & 31u, lh = lane >> 5;
  const uint32_t Np = (Ncols + 31u) / 32u * 32u;   // (n0 is a multiple of 96: it starts a block of 32 rows of the twin)
  auto left = [&](uint32_t rows) { const uint64_t b = (uint64_t)rows * Dp * 4u; return (uint32_t)(b < 0xffffffffull ? b : 0xffffffffull); };
  const __amdgpu_buffer_rsrc_t RA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * Dp), 0, (int)left(M - m0), 0x00020000);
  const __amdgpu_buffer_rsrc_t RB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)n0 * Dp), 0, (int)left(Np - n0), 0x00020000);
  const uint32_t oa = ((wm * 32u + lr) * Dp + lh * 4u) * 4u;
  uint32_t ob[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) ob[n] = ((uint32_t)n * (Dp >> 3) * 256u + lh * 128u + lr * 4u) * 4u;
  const uint32_t mine = Dp >> 4;
  f32x16 acc0, acc1, acc2;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; acc2[e] = 0.f; }
  f32x4 fa[NBUF], fb[NBUF][3];
  float ns = 0.f;
  auto load = [&](auto buf_tag, uint32_t j) {
    constexpr int buf = decltype(buf_tag)::value;
    const uint32_t s = 2u * j + kg;
    fa[buf] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RA, (int)oa, (int)(s * 32u), 0));
#pragma unroll
    for (int n = 0; n < 3; ++n) fb[buf][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RB, (int)ob[n], (int)(s * 1024u), 0));
  };
  auto compute = [&](auto buf_tag) {
    constexpr int buf = decltype(buf_tag)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][e], fb[buf][0][e], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][e], fb[buf][1][e], acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][e], fb[buf][2][e], acc2, 0, 0, 0);
    }
    if constexpr (NORM) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ns += fa[buf][e] * fa[buf][e];
    }
  };
  auto step = [&](auto b_tag, uint32_t j) {
    constexpr int b = decltype(b_tag)::value;
    load(std::integral_constant<int, (b + NBUF - 1) % NBUF>{}, j + NBUF - 1);
    compute(b_tag);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    if (yield_every) {
      const uint32_t ev = yield_every & 255u, amt = yield_every >> 8;
      if ((j % ev) == 0) { if (amt <= 1) __builtin_amdgcn_s_sleep(1); else if (amt == 2) __builtin_amdgcn_s_sleep(2); else if (amt == 3) __builtin_amdgcn_s_sleep(3); else __builtin_amdgcn_s_sleep(4); }   // (yield_every: period | naps of 64 cycles << 8)
    }
  };
  sa_static_for<0, NBUF - 1>([&](auto b) { load(b, (uint32_t)decltype(b)::value); __builtin_amdgcn_sched_barrier(0); });
  SA_STAMP(tr, 1);
  uint32_t j = 0;
  for (; j + NBUF <= mine; j += NBUF) sa_static_for<0, NBUF>([&](auto b) { step(b, j + decltype(b)::value); });
  const uint32_t rem = mine - j;
  sa_static_for<0, NBUF>([&](auto b) { if ((uint32_t)decltype(b)::value < rem) step(b, j + decltype(b)::value); });
  SA_STAMP(tr, 2);
  f32x4* red = (f32x4*)lds;           // [4 waves][6][64 lanes]
  float* rn = lds + SA_KS96_RED;      // [4 waves][32]
  if (kg == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) red[(w4 * 6 + g) * 64 + lane] = f32x4{acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
#pragma unroll
    for (int g = 0; g < 2; ++g) red[(w4 * 6 + 4 + g) * 64 + lane] = f32x4{acc2[8 + 4 * g], acc2[9 + 4 * g], acc2[10 + 4 * g], acc2[11 + 4 * g]};
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) red[(w4 * 6 + g) * 64 + lane] = f32x4{acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
#pragma unroll
    for (int g = 0; g < 2; ++g) red[(w4 * 6 + 4 + g) * 64 + lane] = f32x4{acc2[4 * g], acc2[4 * g + 1], acc2[4 * g + 2], acc2[4 * g + 3]};
  }
  if constexpr (NORM) {
    ns += __shfl_xor(ns, 32);
    if (lh == 0) rn[w4 * 32 + lr] = ns;
  }
  __syncthreads();
  const uint32_t pw = w4 ^ 1u;
  if (kg == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 o = red[(pw * 6 + g) * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) out[4 * g + c] = acc0[4 * g + c] + o[c];
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 o = red[(pw * 6 + 4 + g) * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) out2[4 * g + c] = acc2[4 * g + c] + o[c];
    }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 o = red[(pw * 6 + g) * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) out[4 * g + c] = o[c] + acc1[4 * g + c];
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 o = red[(pw * 6 + 4 + g) * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) out2[4 * g + c] = o[c] + acc2[8 + 4 * g + c];
    }
  }
  if constexpr (NORM) *nsq = rn[(wm * 2) * 32 + lr] + rn[(wm * 2 + 1) * 32 + lr];
  __syncthreads();  // the epilogue reuses this LDS
}

// The k-split loop for the 64 x 128 tile: wave (wm, kg) owns tile rows wm 32 .. +31, ALL 128 columns (four accumulators) and every second
// 8-deep k-step — one row-major gather of A and four fragment-order loads of B per 16 matrix instructions (the direct loop's 32 x 64 wave
// tiles: 6 per 16, and every fragment loaded by two waves), nothing loaded twice.  Wave (wm, x) keeps columns x 64 .. +63 and ships its other
// two accumulators to wave (wm, 1 - x) through LDS (8 KB per wave): it leaves with acc[0][0..1] as the 2 x 2 wave layout of the epilogues has them.
// Two blocks per CU (150 VGPRs + 64 AGPRs as the compiler allots them).  Asked for three waves per SIMD (__launch_bounds__(256, 3): 162
// registers, nothing spilled) the kernel is SLOWER — C5 645 us per frame against 616, c2b on this tile 119 against 107: like the direct
// loop's tiles (three resident beat four), fewer co-resident operand streams keep their lines in L1.
template <int NBUF>
__device__ __forceinline__ void gemm_mainloop_ks128(gfloat_p A, gfloat_p B, uint32_t M, uint32_t Ncols, uint32_t Dp, uint32_t m0,
                                                    uint32_t n0, float* lds, f32x16 (&out)[1][2], uint64_t* tr = nullptr) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) & 3u;
  const uint32_t wm = w4 >> 1, kg = w4 & 1u, lr = lane & 31u, lh = lane >> 5;
  const uint32_t Np = (Ncols + 31u) / 32u * 32u, nb0 = n0 & ~31u;
  auto left = [&](uint32_t rows) { const uint64_t b = (uint64_t)rows * Dp * 4u; return (uint32_t)(b < 0xffffffffull ? b : 0xffffffffull); };
  const __amdgpu_buffer_rsrc_t RA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * Dp), 0, (int)left(M - m0), 0x00020000);
  const __amdgpu_buffer_rsrc_t RB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)nb0 * Dp), 0, (int)left(Np - nb0), 0x00020000);
  const uint32_t oa = ((wm * 32u + lr) * Dp + lh * 4u) * 4u;
  uint32_t ob[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const uint32_t rr = (n0 - nb0) + n * 32u + lr;
    ob[n] = ((rr >> 5) * (Dp >> 3) * 256u + lh * 128u + (rr & 31u) * 4u) * 4u;
  }
  const uint32_t mine = Dp >> 4;
  f32x16 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
  f32x4 fa[NBUF], fb[NBUF][4];
  auto load = [&](auto buf_tag, uint32_t j) {
    constexpr int buf = decltype(buf_tag)::value;
    const uint32_t s = 2u * j + kg;
    fa[buf] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RA, (int)oa, (int)(s * 32u), 0));
#pragma unroll
    for (int n = 0; n < 4; ++n) fb[buf][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RB, (int)ob[n], (int)(s * 1024u), 0));
  };
  auto step = [&](auto b_tag, uint32_t j) {
    constexpr int b = decltype(b_tag)::value;
    load(std::integral_constant<int, (b + NBUF - 1) % NBUF>{}, j + NBUF - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[b][e], fb[b][n][e], acc[n], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    sa_static_for<0, 4>([&](auto) { __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); });
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
  };
  sa_static_for<0, NBUF - 1>([&](auto b) { load(b, (uint32_t)decltype(b)::value); __builtin_amdgcn_sched_barrier(0); });
  SA_STAMP(tr, 1);
  uint32_t j = 0;
  for (; j + NBUF <= mine; j += NBUF) sa_static_for<0, NBUF>([&](auto b) { step(b, j + decltype(b)::value); });
  const uint32_t rem = mine - j;
  sa_static_for<0, NBUF>([&](auto b) { if ((uint32_t)decltype(b)::value < rem) step(b, j + decltype(b)::value); });
  SA_STAMP(tr, 2);
  f32x4* red = (f32x4*)lds;   // [4 waves][2 accumulators][4][64 lanes] f32x4 = 32 KB
  auto ship = [&](const f32x16& a, uint32_t slot) {
#pragma unroll
    for (int g = 0; g < 4; ++g) red[((w4 * 2u + slot) * 4u + g) * 64u + lane] = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
  };
  if (kg == 0) { ship(acc[2], 0); ship(acc[3], 1); } else { ship(acc[0], 0); ship(acc[1], 1); }
  __syncthreads();
  const uint32_t pw = w4 ^ 1u;
  auto take = [&](const f32x16& mine_, uint32_t slot, f32x16& dst, bool mine_first) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 o = red[((pw * 2u + slot) * 4u + g) * 64u + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) dst[4 * g + c] = mine_first ? mine_[4 * g + c] + o[c] : o[c] + mine_[4 * g + c];
    }
  };
  // (k-half 0 + k-half 1 whichever wave adds)
  if (kg == 0) { take(acc[0], 0, out[0][0], true); take(acc[1], 1, out[0][1], true); }
  else { take(acc[2], 0, out[0][0], false); take(acc[3], 1, out[0][1], false); }
  __syncthreads();
}

// The same idea for the wider tiles (128x128, 64x128, 128x64: 2 x 2 waves, each a (BM/2) x (BN/2) wave tile over the WHOLE k range): operands
// straight from memory — A row-major, B from the bank's fragment-order twin — TM + TN buffer loads per 8-deep k-step for 4 TM TN matrix
// instructions, no LDS stage, no barrier, no reduction.  The two waves of a wave row (column) load the same A (B) fragments: twice the
// L1 traffic of the staged loop, none of its LDS instructions.
template <int BM, int BN, int NBUF>
__device__ __forceinline__ void gemm_mainloop_direct(gfloat_p A, gfloat_p B, uint32_t M, uint32_t Ncols, uint32_t Dp, uint32_t m0, uint32_t n0,
                                                     f32x16 (&acc)[BM / 64][BN / 64], uint64_t* tr = nullptr) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) & 3u;
  const uint32_t wm = w4 >> 1, wn = w4 & 1u, lr = lane & 31u, lh = lane >> 5;
  const uint32_t Np = (Ncols + 31u) / 32u * 32u, nb0 = n0 & ~31u;
  auto left = [&](uint32_t rows) { const uint64_t b = (uint64_t)rows * Dp * 4u; return (uint32_t)(b < 0xffffffffull ? b : 0xffffffffull); };
  const __amdgpu_buffer_rsrc_t RA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * Dp), 0, (int)left(M - m0), 0x00020000);
  const __amdgpu_buffer_rsrc_t RB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)nb0 * Dp), 0, (int)left(Np - nb0), 0x00020000);
  uint32_t oa[TM], ob[TN];
#pragma unroll
  for (int m = 0; m < TM; ++m) oa[m] = ((wm * (BM / 2) + m * 32u + lr) * Dp + lh * 4u) * 4u;
#pragma unroll
  for (int n = 0; n < TN; ++n) {
    const uint32_t rr = (n0 - nb0) + wn * (BN / 2) + n * 32u + lr;
    ob[n] = ((rr >> 5) * (Dp >> 3) * 256u + lh * 128u + (rr & 31u) * 4u) * 4u;
  }
  const uint32_t steps = Dp >> 3;
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
  f32x4 fa[NBUF][TM], fb[NBUF][TN];
  auto load = [&](auto buf_tag, uint32_t s) {
    constexpr int buf = decltype(buf_tag)::value;
#pragma unroll
    for (int m = 0; m < TM; ++m) fa[buf][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RA, (int)oa[m], (int)(s * 32u), 0));
#pragma unroll
    for (int n = 0; n < TN; ++n) fb[buf][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RB, (int)ob[n], (int)(s * 1024u), 0));
  };
  auto step = [&](auto b_tag, uint32_t s) {
    constexpr int b = decltype(b_tag)::value;
    load(std::integral_constant<int, (b + NBUF - 1) % NBUF>{}, s + NBUF - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[b][m][e], fb[b][n][e], acc[m][n], 0, 0, 0);
    // the step's loads spread evenly over its 4 TM TN matrix instructions
    constexpr int NM = 4 * TM * TN, NL = TM + TN, GAP = NM / NL;
    sa_static_for<0, NL>([&](auto i) {
      __builtin_amdgcn_sched_group_barrier(0x008, decltype(i)::value == 0 ? 1 : GAP, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    });
    __builtin_amdgcn_sched_group_barrier(0x008, NM - 1 - (NL - 1) * GAP, 0);
  };
  sa_static_for<0, NBUF - 1>([&](auto b) { load(b, (uint32_t)decltype(b)::value); __builtin_amdgcn_sched_barrier(0); });
  SA_STAMP(tr, 1);
  uint32_t s = 0;
  for (; s + NBUF <= steps; s += NBUF) sa_static_for<0, NBUF>([&](auto b) { step(b, s + decltype(b)::value); });
  const uint32_t rem = steps - s;
  sa_static_for<0, NBUF>([&](auto b) { if ((uint32_t)decltype(b)::value < rem) step(b, s + decltype(b)::value); });
  SA_STAMP(tr, 2);
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

// Running maximum of the present weights (BestFitVoting's max_dist, voting/best.rs:59-76) WITHOUT atomics: a register-level
// wave reduction (DPP / ds_swizzle), the wave maxima through one LDS word each and one barrier, then every workgroup stores
// its maximum into its own slot of S.vis_max_key (slot = tile index inside the scene; the host puts the tile grid into the
// descriptor).  k_bestfit_tile folds the slots.  Device-scope atomics on shared words serialise on this chip
// (MI355X_MICROARCH.md "fanin"): one per wave on one word cost 45 us at 4096 waves; sharded 64 ways, one per wave (2048)
// still +8.7 us and one per workgroup (256) +1.1 us on the C2 kernel.  Key 0 = "no weight" (reads back as -1.0).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  v = mx(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true));    // xor 1
  v = mx(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true));    // xor 2
  v = mx(v, (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (4 << 10) | 0x1F));     // xor 4
  v = mx(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true));   // row_ror:8 = xor 8
  v = mx(v, (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (16 << 10) | 0x1F));    // xor 16
  return mx((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 32));
}
__device__ __forceinline__ void block_max_key(uint32_t SA_G* slots, uint32_t slot, uint32_t kmax) {
  __shared__ uint32_t s_wmax[16];
  const uint32_t m = wave_max_u32(kmax);
  const uint32_t wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63u) == 0) s_wmax[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t b = 0;
    for (uint32_t w = 0; w < nw; ++w) b = s_wmax[w] > b ? s_wmax[w] : b;
    slots[slot] = b;
  }
}

// Everything the reference does per (candidate, observation) pair after the dot product.  compatible() (sort.rs:250-270)
// compares the candidate's epoch — the same for every row of a scene-frame — with the track's, so its epoch part and the
// choice of the constraint (spatio_temporal_constraints.rs:48-59) are per-COLUMN facts, folded into GemmCols before the
// main loop; what is left per cell is dist_in_2r <= max_dist when a constraint applies.
// Bit c of the result: cell c of this lane fails the spatio-temporal constraint of its column (dist_in_2r > max_dist).  One
// pass, entered only when some lane of the wave has a constrained column at all.
template <int CELLS, typename RowGeo>
__device__ __forceinline__ uint32_t constraint_mask(const GemmCols& col, RowGeo row_geo) {
  uint32_t failed = 0;
  if (__ballot(col.cmax >= 0.0f) != 0ull) {
    if (col.cmax >= 0.0f) {
#pragma unroll
      for (int c = 0; c < CELLS; ++c) failed |= (sa_dist_in_2r(*row_geo(c), col.g) <= col.cmax) ? 0u : (1u << c);
    }
  }
  return failed;
}
// A candidate's row operands in raw-feature mode (the fused first phase: nothing the preparation blocks write may be read): what
// frame_prep_block derives for it (visual_sort/simple_api.rs:130-170, metric.rs:227-249).  FETCH issues the loads — which of them by the
// scene's flags alone, none behind a branch on a loaded value — and the values are turned into geometry and the feature_can_be_used gate
// BEHIND the main loop: with the gate's short-circuit chain in front of it (present? -> quality -> own area) the tile's first wave
// started its loop a memory round trip late, and the other three waited for it at the exchange.
struct RawRow { float xc, yc, aspect, height, q, oa; uint32_t fp; };
__device__ __forceinline__ RawRow raw_row_fetch(const SceneDev& S, uint32_t gi) {
  RawRow r;
  const BoxRaw br = sa_ldg(S.c_raw + gi);
  r.xc = br.box.xc; r.yc = br.box.yc; r.aspect = br.box.aspect; r.height = br.box.height;
  r.fp = (S.flags & SCN_HAS_FPRESENT) ? (uint32_t)S.c_fpresent_in[gi] : 1u;
  r.q = (S.flags & SCN_HAS_QUALITY) ? S.c_quality[gi] : 1.0f;
  r.oa = (S.flags & SCN_HAS_OWN) ? S.c_own[gi] : __builtin_nanf("");
  return r;
}
__device__ __forceinline__ bool raw_row_usable(const SceneDev& S, const SaParams& p, const RawRow& r, sa_geo* g) {
  g->xc = r.xc; g->yc = r.yc; g->r = sa_radius(r.aspect, r.height); g->hha = r.height * r.height * r.aspect;
  const bool perc_ok = !(r.oa == r.oa) || r.oa >= p.visual_minimal_own_area_use;
  return (S.flags & SCN_HAS_FEATS) != 0 && r.fp != 0 && sa_area(r.aspect, r.height) >= p.visual_minimal_area && r.q >= p.visual_minimal_quality_use && perc_ok;
}
// EU: distance::euclidean (distance.rs:9-19) on the matrix cores.  sqrt(|a|^2 + |b|^2 - 2 a.b) cancels on near-identical vectors —
// exactly the true matches — so a cell whose expansion is not trustworthy to 1e-5 relative, d^2 < rho (|a|^2 + |b|^2) with
// rho = 5e-3 sqrt(Dp) (twice the largest error of the f32 expansion seen over 10^6 pairs, scripts/euclid_error_model.py), is FLAGGED
// instead of evaluated: the tile recomputes it afterwards as the direct sum of (a - b)^2 (euclid_fixup).  In tracking frames that is
// about one cell per candidate; everything else rides the contraction at the cosine kernel's speed.
template <bool EU>
__device__ __forceinline__ float visual_cell(const SaParams& p, float dot, float na, bool cons_failed, const GemmCols& col,
                                             uint32_t* kmax, bool* flagged) {
  // Straight-line on purpose: with a short-circuit chain (usable? column ok? ...) the compiler sinks every operand load into the
  // branch that first needs it, and the row operands come from LDS — each cell then pays two or three LDS round trips one after
  // the other.  na = squared norm of the candidate's feature, NaN when feature_can_be_used() says no (the distance is then NaN
  // and fails is_ok like any NaN).  The spatio-temporal constraint, rare and expensive, is evaluated by the caller in a pass of
  // its own (constraint_mask) and arrives as one bit: a branch per cell made every cell a basic block of its own, and a lone
  // wave then walks 16 dependent chains (multiply, v_rsq, multiply, compare, key ...) one after the other — 4 k cycles.
  const bool ok0 = col.ok && !cons_failed;
  bool ok = ok0;
  float w;
  if constexpr (EU) {
    const float s = na + col.nb;
    const float d2 = s - 2.0f * dot;
    const bool f = ok0 & (d2 < p.eu_rho * s);        // NaN norms (unusable candidate): never flagged, never present
    *flagged = f;
    const float d = __fsqrt_rn(d2 < 0.0f ? 0.0f : d2);  // NaN stays NaN
    ok = ok & !f & (d <= p.visual_threshold);         // VisualSortMetricType::is_ok for Euclidean (NaN fails)
    w = d;                                            // distance_to_weight: the distance itself
  } else {
    // divided / (f1_divisor * f2_divisor).sqrt(): v_rsq_f32 + multiply, <= 2 ulp from the reference's sqrt + divide,
    // two orders of magnitude inside the 1e-5 gate and ~25 instructions cheaper per cell
    const float d = dot * __frsqrt_rn(na * col.nb);
    ok = ok & (d >= p.visual_threshold);  // VisualSortMetricType::is_ok (NaN fails)
    w = 1.0f - d;                         // distance_to_weight
    *flagged = false;
  }
  const uint32_t key = ok ? sa_f32_key(w) : 0u;
  *kmax = key > *kmax ? key : *kmax;
  return ok ? w : __builtin_nanf("");
}

// One tile of the fused visual cost kernel.  RAW (the heterogeneous frame launch, k_frame_visual): the frame-preparation
// blocks run BESIDE this tile, not before it, so nothing they produce may be read — the candidate features come straight
// from the uploaded rows (their length is a multiple of 32: no padding needed), their squared norms are accumulated from
// the A fragments inside the main loop, and the row's geometry / feature_can_be_used gate are derived from the raw box.
//
// PART (bank depth 1 only): instead of the N x T weight matrix the tile emits what k_bestfit_tile would derive from it — per row
// the lightest weight over the tile's columns (lowest column on ties), per column the lightest over its rows (lowest row) — as
// the BestFit partials k_bestfit_resolve folds.  BestFit ranks groups by W = sum_k f64(max_dist - w_k); with one observation per
// track that is decreasing in w, so the heaviest group of a row or column is its lightest weight and max_dist (known only when
// every tile is done) is needed neither here nor in k_bestfit_resolve, which folds the partials on the weights.  Where two
// DIFFERENT weights round to the same f32 difference from max_dist the reference would fall back on the index order; they
// differ by < 6e-8, four hundred times below the 1e-5 the feature distances themselves are good for.  One launch
// (k_bestfit_tile) and the write + re-read of the matrix disappear; the parity taps re-run the contraction with PART = false.
// LDS floats a contraction tile needs.  KGT: 1 / 2 / 4 = staged loop with that many k-groups (two stages each), 0 = ring (three stages),
// 9 = k-split loop (64 x 64: the quadrant exchange), 15 = direct loop (wider tiles: no LDS in the main loop), 17 = k-split loop of the 64 x 128
// tile (32 KB exchange) — and, for every loop, what the
// fused epilogue lays out in the same buffer afterwards: row operands, PART / EU: column minima + a 64-row key tile, EU: flag words + list.
constexpr uint32_t gemm_lds_floats(int BM, int BN, int KGT, bool PART, bool EU) {
  const int KG = (KGT == 9 || KGT == 15 || KGT == 17) ? 1 : KGT ? KGT : 1;
  const uint32_t loop = KGT == 15 ? 0u : KGT == 17 ? 8192u : KGT == 9 ? 4u * 4u * 64u * 4u + 4u * 32u : (uint32_t)((KGT ? KG * 2 : 3) * (BM + BN) * BK);
  const uint32_t epi = (uint32_t)((6 + KG) * BM + 2 * BN + ((PART || EU) ? 64 * (BN + 4) : 0) + (EU ? 64 * (BN / 32) + 256 : 0));
  uint32_t m = loop > epi ? loop : epi;
  // the direct loop's 64 x 128 tile (85 VGPRs: four blocks per CU by registers) is held to THREE blocks per CU by its LDS footprint: measured at
  // C5 (5000 x 2000 x 4096), four resident tiles 714 us, three 631, two 655 (the staged loop: 657) — a fourth tile's operand streams thrash the L1
  if (KGT == 15 && BM == 64 && BN == 128 && m < 12288u) m = 12288u;
  return (m + 63u) & ~63u;
}
template <int BM, int BN, int KGT, bool RAW, bool PART, bool EU = false>
__device__ __forceinline__ void visual_cosine_tile(const SceneDev& S, const SaParams& p, uint32_t bx, uint32_t by, float* lds) {
  constexpr bool KSPLIT = KGT == 9;      // KGT == 9: the k-split main loop (gemm_mainloop_ks: the bank read in fragment order, no LDS stage)
  constexpr bool DIRECT = KGT == 15;     // KGT == 15: the direct main loop of the wider tiles (gemm_mainloop_direct)
  constexpr bool KS128 = KGT == 17;      // KGT == 17: the k-split main loop of the 64 x 128 tile (gemm_mainloop_ks128)
  static_assert(!KS128 || (BM == 64 && BN == 128 && !RAW), "k-split 64 x 128");
  constexpr int KG = (KSPLIT || DIRECT || KS128) ? 1 : KGT ? KGT : 1;  // KGT == 0: ring main loop (one k-group, 3 LDS stages)
  uint64_t* tr = SA_TRACE_PTR();
  SA_STAMP(tr, 0);
  const uint32_t N = S.N, TK = S.TK, K = S.K;
  const uint32_t m0 = by * BM, n0 = bx * BN;
  if (m0 >= N || n0 >= TK) return;
  const uint32_t key_slot = by * ((TK + BN - 1) / BN) + bx;  // < S.nkeys = tiles of THIS scene
  constexpr int TM = BM / 64, TN = BN / 64;
  static_assert(KG == 1 || (TM == 1 && TN == 1), "k-groups only with the 64x64 tile");
  static_assert(!RAW || (TM == 1 && TN == 1 && KGT != 0), "raw mode: 64x64 tiles with k-groups");
  static_assert(!KSPLIT || (TM == 1 && TN == 1), "k-split: 64x64 tiles");
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w4 = (tid >> 6) & 3u, kg = (KSPLIT || DIRECT || KS128) ? 0u : tid >> 8;
  const uint32_t wm = w4 >> 1, wn = w4 & 1u, lr = lane & 31u, lh = lane >> 5;
  // The epilogue's per-row / per-column operands are fetched BEFORE the contraction: their L2/HBM latency (a chain of
  // dependent loads that used to sit, fully exposed, between the last MFMA and the first store: ~2 us of a 18 us kernel at
  // C2) disappears behind the main loop.  Row operands: thread r < BM holds row r (goes through LDS afterwards).
  float pre_na = 0.f, pre_us = 0.f;
  sa_geo pre_g{0.f, 0.f, 0.f, 0.f};
  RawRow pre_raw{0.f, 0.f, 1.f, 1.f, 0.f, 0.f, 0u};
  static_assert(BM <= 256, "one thread per tile row");
  const bool pre_in = tid < (uint32_t)BM && m0 + tid < N;
  if (pre_in) {
    const uint32_t gi = m0 + tid;
    if constexpr (RAW) pre_raw = raw_row_fetch(S, gi);   // (loads only: raw_row_usable behind the main loop)
    else {
      pre_na = S.c_fnorm[gi];
      pre_us = S.c_usable[gi] ? 1.f : 0.f;
      pre_g = sa_ldg(S.c_geo + gi);
    }
  }
  GemmCols col[TN];
#pragma unroll
  for (int n = 0; n < TN; ++n) {
    uint32_t gj = n0 + wn * (BN / 2) + n * 32 + lr;
    col[n].ok = false;
    col[n].nb = 0.f;
    col[n].cmax = -1.0f;
    col[n].g = sa_geo{0.f, 0.f, 0.f, 0.f};
    if (gj < TK) {
      // independent loads, no short-circuit: one round trip instead of a chain of three
      const uint32_t t = gj / K;
      const float nb = S.t_fnorm[gj];
      const uint8_t pres = S.t_fpresent[gj];
      const uint32_t cnt = S.t_fcount[t];
      const uint64_t te = S.t_epoch[t];
      col[n].g = sa_ldg(S.t_geo + t);
      col[n].nb = nb;
      const uint64_t delta = S.epoch > te ? S.epoch - te : te - S.epoch;
      col[n].ok = (pres != 0) & (cnt >= p.min_track_len) & (p.max_idle >= delta);
      // (the FIRST constraint whose window holds the track's age — as a chain of selects from the last one down: a `break` on the loaded
      // epoch would hold every wave of the tile in front of its main loop for the load)
      for (uint32_t i = p.cons.n; i-- > 0;) col[n].cmax = p.cons.delta[i] >= delta ? p.cons.max_dist[i] : col[n].cmax;
    }
  }

  f32x16 acc[TM][TN];
  float nsq = 0.f;
  if constexpr (KSPLIT) gemm_mainloop_ks<4, RAW, true>((gfloat_p)(RAW ? S.c_feat_raw : (const float SA_G*)S.c_feat), (gfloat_p)S.t_ffrag, N, TK, S.Dp, m0, n0, lds, acc[0][0], tr, &nsq, RAW ? p.ks_yield : 0u);
  else if constexpr (KS128) gemm_mainloop_ks128<3>((gfloat_p)S.c_feat, (gfloat_p)S.t_ffrag, N, TK, S.Dp, m0, n0, lds, acc, tr);
  else if constexpr (DIRECT) gemm_mainloop_direct<BM, BN, 4>((gfloat_p)S.c_feat, (gfloat_p)S.t_ffrag, N, TK, S.Dp, m0, n0, acc, tr);
  else if constexpr (KGT == 0) gemm_mainloop_ring<BM, BN>((gfloat_p)S.c_feat, (gfloat_p)S.t_feat, N, TK, S.Dp, m0, n0, lds, acc, tr);
  else if constexpr (RAW) gemm_mainloop<BM, BN, KG, true>((gfloat_p)S.c_feat_raw, (gfloat_p)S.t_feat, N, TK, S.Dp, m0, n0, lds, acc, tr, &nsq);
  else gemm_mainloop<BM, BN, KG>((gfloat_p)S.c_feat, (gfloat_p)S.t_feat, N, TK, S.Dp, m0, n0, lds, acc, tr);

  if constexpr (RAW) { if (pre_in) pre_us = raw_row_usable(S, p, pre_raw, &pre_g) ? 1.f : 0.f; }
  constexpr int R = 16 / KG;
  float part[TM == 1 && TN == 1 ? R : 1];
  if constexpr (TM == 1 && TN == 1) kgroup_reduce_spread<KG>(acc[0][0], lds, part);
  SA_STAMP(tr, 3);

  // ---- fused epilogue: row metadata through LDS, column metadata in registers ----
  float* s_na = lds;                      // [BM]
  sa_geo* s_g = (sa_geo*)(lds + 2 * BM);  // [BM]
  float* s_np = lds + 6 * BM;             // [KG][BM] raw mode: squared-norm partials of the k-groups
  unsigned long long* s_ck = (unsigned long long*)(lds + (6 + KG) * BM);  // [BN] PART: (weight key << 32) | row, minimum per column
  constexpr uint32_t KS = BN + 4;                                          // row stride of the key tile (words)
  uint32_t* s_key = (uint32_t*)(lds + (6 + KG) * BM + 2 * BN);             // [64][KS] PART: order-preserving keys of 64 tile rows
  constexpr uint32_t FW = BN / 32;                                         // EU: words of flag bits per tile row
  uint32_t* s_flag = (uint32_t*)(lds + (6 + KG) * BM + 2 * BN + 64 * KS);  // [64][FW] EU: cells of the current 64-row pass to recompute directly
  constexpr uint32_t FL_CAP = 255;                                         // EU: the same cells as a list (a tracking frame has a handful per tile); [FL_CAP] = their count
  uint32_t* s_flist = s_flag + 64 * FW;
  static_assert(!EU || ((6 + KG) * BM + 2 * BN + 64 * (BN + 4) + 64 * (BN / 32) + 256) <= gemm_lds_floats(BM, BN, KGT, PART, EU), "flag words and list must fit the tile's LDS");
  if constexpr (EU) {
    for (uint32_t i = tid; i < 64u * FW; i += blockDim.x) s_flag[i] = 0u;
    if (tid == 0) s_flist[FL_CAP] = 0u;
  }
  if (tid < (uint32_t)BM) {
    s_na[tid] = pre_us != 0.f ? pre_na : __builtin_nanf("");  // the feature_can_be_used gate rides in the norm
    s_g[tid] = pre_g;
  }
  if constexpr (PART) {
    for (uint32_t i = tid; i < (uint32_t)BN; i += blockDim.x) s_ck[i] = ~0ull;
  }
  if constexpr (RAW) {
    // the two halves of a row's k values sit in lanes lr and lr + 32; the waves wn = 0 / 1 of a group hold the same rows
    // (k-split: the main loop has already folded lane halves and k-halves)
    if constexpr (!KSPLIT) nsq += __shfl_xor(nsq, 32);
    if (wn == 0 && lh == 0) s_np[kg * BM + wm * 32 + lr] = nsq;
  }
  __syncthreads();
  if constexpr (RAW) {
    if (tid < (uint32_t)BM) {
      float s = s_np[tid];
#pragma unroll
      for (int g2 = 1; g2 < KG; ++g2) s += s_np[g2 * BM + tid];
      s_na[tid] = pre_us != 0.f ? s : __builtin_nanf("");
    }
    __syncthreads();
  }
  uint32_t kmax = 0;  // order-preserving key of the largest present weight seen by this lane
  // PART: the order-preserving keys of 64 tile rows at a time go to an LDS tile; after a barrier every row is scanned by
  // blockDim / 64 threads (contiguous column segments, running minimum with the lowest column on ties, then a few DPP exchanges
  // between the threads of a row) and its partial goes straight to memory.  Column minima: in-lane over the rows a lane holds,
  // the two lane halves by one exchange, the waves stacked on each other by a 64-bit LDS minimum.  (A first version reduced every
  // accumulator register across the wave with DPP + ballot: 30 instructions per cell, 10.7 k cycles of epilogue for the
  // one-k-group tile where each lane holds 16 cells.)
  static_assert(!PART || ((6 + KG) * BM + 2 * BN + 64 * (BN + 4)) <= gemm_lds_floats(BM, BN, KGT, PART, EU), "key tile must fit the tile's LDS");
  auto rows_to_partials = [&](uint32_t m) {
    __syncthreads();  // the key tile (and, in the last pass, the column minima) complete
    const uint32_t nthr = blockDim.x, TPR = nthr >> 6, CPT = BN / TPR;  // threads per row, columns per thread
    const uint32_t rr = tid / TPR, seg = tid % TPR;
    const uint32_t* kp = s_key + rr * KS + seg * CPT;
    uint32_t bk = 0xffffffffu, bc = 0;
    for (uint32_t c4 = 0; c4 < CPT; c4 += 4) {
      const uint4 v = *(const uint4*)(kp + c4);
      if (v.x < bk) { bk = v.x; bc = c4; }
      if (v.y < bk) { bk = v.y; bc = c4 + 1; }
      if (v.z < bk) { bk = v.z; bc = c4 + 2; }
      if (v.w < bk) { bk = v.w; bc = c4 + 3; }
    }
    bc += seg * CPT;
    auto take = [&](uint32_t ok, uint32_t oc) { if (ok < bk || (ok == bk && oc < bc)) { bk = ok; bc = oc; } };
    take((uint32_t)__builtin_amdgcn_mov_dpp((int)bk, 0xB1, 0xF, 0xF, true), (uint32_t)__builtin_amdgcn_mov_dpp((int)bc, 0xB1, 0xF, 0xF, true));
    take((uint32_t)__builtin_amdgcn_mov_dpp((int)bk, 0x4E, 0xF, 0xF, true), (uint32_t)__builtin_amdgcn_mov_dpp((int)bc, 0x4E, 0xF, 0xF, true));
    if (TPR == 8) take(__shfl_xor(bk, 4), __shfl_xor(bc, 4));  // the other quad of the 8-thread group
    const uint32_t gi = m0 + (rr >> 5) * (BM / 2) + m * 32 + (rr & 31u);
    if (seg == 0 && gi < N && p.vote_words) {
      if (bk != 0xffffffffu)
        __hip_atomic_fetch_min(S.row_best + gi, ((unsigned long long)bk << 32) | (n0 + bc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (seg == 0 && gi < N) {
      const bool has = bk != 0xffffffffu;
      S.row_part_w[(size_t)bx * S.N + gi] = has ? (double)sa_key_f32(bk) : -1.0;
      S.row_part_t[(size_t)bx * S.N + gi] = has ? (int32_t)(n0 + bc) : -1;
    }
  };
  // EU: the flagged cells of the 64-row pass `m`, recomputed as the direct sum of (a - b)^2: a wave per tile row, the lanes along k
  // (16-byte loads of both rows, L2-resident: the tile has just streamed them), a wave reduction, one result per cell — into the
  // key tile and the column minima (PART) or the weight matrix.  A tile with more than 64 such cells reports the frame as
  // ill-conditioned for the expansion (S.stats[0], read by the host after the frame: it switches the scene's engine to the
  // vector-pipe kernel for a while); the answers of THIS frame are exact either way.
  auto euclid_fixup = [&](uint32_t m) {
    if constexpr (EU) {
      __syncthreads();  // flag words / list (and the keys / matrix cells of the pass) complete
      const uint32_t wave = tid >> 6, nw = blockDim.x >> 6;
      const float SA_G* Ab = RAW ? S.c_feat_raw : (const float SA_G*)S.c_feat;
      auto recompute = [&](uint32_t lrow, uint32_t lc) {
        const uint32_t li = (lrow >> 5) * (BM / 2) + m * 32 + (lrow & 31u), gi = m0 + li, gj = n0 + lc;
        const float SA_G* a = Ab + (size_t)gi * S.Dp;
        // (k-split / direct loops: the track's row out of the fragment-order twin the tile has just streamed — L2-resident; the row-major
        // bank is cold there, a trip to memory per flagged cell)
        constexpr bool TWIN = KSPLIT || DIRECT || KS128;
        const float SA_G* b = TWIN ? S.t_ffrag + sa_frag_index(gj, 0, S.Dp) : S.t_feat + (size_t)gj * S.Dp;
        float acc2 = 0.f;
        for (uint32_t k = lane * 4u; k < S.Dp; k += 256u) {
          const f32x4 x = *(const f32x4 SA_G*)(a + k), y = *(const f32x4 SA_G*)(b + (TWIN ? (size_t)(k >> 3) * 256u + ((k >> 2) & 1u) * 128u : (size_t)k));
          const float d0 = x[0] - y[0], d1 = x[1] - y[1], d2 = x[2] - y[2], d3 = x[3] - y[3];
          acc2 += d0 * d0; acc2 += d1 * d1; acc2 += d2 * d2; acc2 += d3 * d3;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc2 += __shfl_xor(acc2, o);
        if (lane == 0) {
          const float d = __fsqrt_rn(acc2);
          const bool ok = d <= p.visual_threshold;
          const uint32_t key = ok ? sa_f32_key(d) : 0u;
          if constexpr (PART) {
            const uint32_t k2 = ok ? key : 0xffffffffu;
            s_key[lrow * KS + lc] = k2;
            if (ok) atomicMin(&s_ck[lc], ((unsigned long long)k2 << 32) | gi);
          } else {
            S.vis[(size_t)gi * TK + gj] = ok ? d : __builtin_nanf("");
            kmax = key > kmax ? key : kmax;
          }
        }
      };
      const uint32_t nf = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_flist[FL_CAP]);
      if (nf <= FL_CAP) {
        // the usual case: a handful of cells, dealt round-robin to the waves straight from the list
        for (uint32_t i = wave; i < nf; i += nw) {
          const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_flist[i]);
          recompute(ent >> 8, ent & 255u);
        }
      } else {
        // more than the list holds (an ill-conditioned frame): every wave walks the flag words of its rows
        for (uint32_t lrow = wave; lrow < 64u; lrow += nw)
          for (uint32_t wd = 0; wd < FW; ++wd) {
            uint32_t bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_flag[lrow * FW + wd]);
            while (bits) {
              const uint32_t lc = wd * 32u + (uint32_t)__builtin_ctz(bits);
              bits &= bits - 1u;
              recompute(lrow, lc);
            }
          }
      }
      if (nf > 64u && tid == 0) S.stats[0] = 1u;  // more than 1.5 % of a tile's cells: the frame is ill-conditioned for the expansion
      if (m + 1 < (uint32_t)TM) {  // the next pass reuses the flag words and the list
        __syncthreads();
        for (uint32_t i = tid; i < 64u * FW; i += blockDim.x) s_flag[i] = 0u;
        if (tid == 0) s_flist[FL_CAP] = 0u;
        __syncthreads();
      }
    }
  };
  // The row operands of a lane's cells: accumulator registers 4g .. 4g+3 hold four consecutive tile rows (acc_row), so one
  // 16-byte LDS read per group of four cells, all issued before the first cell is evaluated.
  if constexpr (TM == 1 && TN == 1) {
    const uint32_t lc = wn * 32 + lr, gj = n0 + lc;
    constexpr int G = R / 4;
    static_assert(R % 4 == 0, "a k-group owns whole groups of four accumulator registers");
    f32x4 nav[G];
    uint32_t rbase[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      rbase[g] = wm * 32 + 8u * (kg * (R / 4) + g) + 4u * lh;
      nav[g] = *(const f32x4*)(s_na + rbase[g]);
    }
    uint32_t ckey = 0xffffffffu, crow = 0;
    const uint32_t cfail = constraint_mask<R>(col[0], [&](int c) { return s_g + rbase[c >> 2] + (c & 3); });
    uint32_t fmask = 0;  // EU: bit i = cell i of this lane goes to the direct recompute
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const uint32_t li = rbase[i >> 2] + (i & 3);
      const uint32_t gi = m0 + li;
      bool flagged;
      const float w = visual_cell<EU>(p, part[i], nav[i >> 2][i & 3], (cfail >> i) & 1u, col[0], &kmax, &flagged);  // rows / columns past the edge: ok = false
      // (the flagged cells are REGISTERED behind the loop: with the LDS atomics inside it every cell was a basic block of its own and the
      // lone wave walked 16 dependent chains one after the other — the euclidean cells 6.2 k cycles against the cosine ones' 3.2 k)
      if constexpr (EU) fmask |= (flagged && gi < N) ? (1u << i) : 0u;
      if constexpr (PART) {
        const uint32_t key = (w == w && gi < N) ? sa_f32_key(w) : 0xffffffffu;
        s_key[li * KS + lc] = key;
        if (key < ckey) { ckey = key; crow = gi; }  // rows ascend with i: the lowest row wins ties
      } else if (gi < N && gj < TK) {
        S.vis[(size_t)gi * TK + gj] = w;
      }
    }
    if constexpr (EU) {
      if (__ballot(fmask != 0u) != 0ull) {  // (a tracking frame: about one cell per candidate — most waves skip this)
        while (fmask) {
          const uint32_t i = (uint32_t)__builtin_ctz(fmask);
          fmask &= fmask - 1u;
          const uint32_t li = wm * 32 + 8u * (kg * (R / 4) + (i >> 2)) + 4u * lh + (i & 3u);  // BM = 64: the tile row is the pass row
          atomicOr(&s_flag[li * FW + (lc >> 5)], 1u << (lc & 31u));
          const uint32_t pos = atomicAdd(&s_flist[FL_CAP], 1u);
          if (pos < FL_CAP) s_flist[pos] = (li << 8) | lc;
        }
      }
    }
    if constexpr (PART) {
      unsigned long long cb = ((unsigned long long)ckey << 32) | crow;
      const unsigned long long ob = __shfl_xor(cb, 32);
      cb = ob < cb ? ob : cb;
      if (lh == 0 && (uint32_t)(cb >> 32) != 0xffffffffu) atomicMin(&s_ck[lc], cb);
      SA_STAMP(tr, 6);
      euclid_fixup(0);
      rows_to_partials(0);
      SA_STAMP(tr, 7);
    } else euclid_fixup(0);
  } else {
    uint32_t ckey[TN], crow[TN];
#pragma unroll
    for (int n = 0; n < TN; ++n) { ckey[n] = 0xffffffffu; crow[n] = 0; }
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      f32x4 nav[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) nav[g] = *(const f32x4*)(s_na + wm * (BM / 2) + m * 32 + 8 * g + 4 * lh);
      uint32_t cfail[TN];
#pragma unroll
      for (int n = 0; n < TN; ++n)
        cfail[n] = constraint_mask<16>(col[n], [&](int c) { return s_g + wm * (BM / 2) + m * 32 + acc_row(c, lh); });
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t lrow = wm * 32 + acc_row(r, lh);          // row of the 64-row key tile of this pass
        const uint32_t li = wm * (BM / 2) + m * 32 + acc_row(r, lh);
        const uint32_t gi = m0 + li;
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          const uint32_t lc = wn * (BN / 2) + n * 32 + lr, gj = n0 + lc;
          bool flagged;
          const float w = visual_cell<EU>(p, acc[m][n][r], nav[r >> 2][r & 3], (cfail[n] >> r) & 1u, col[n], &kmax, &flagged);
          if constexpr (EU) {
            if (flagged && gi < N) {
              atomicOr(&s_flag[lrow * FW + (lc >> 5)], 1u << (lc & 31u));
              const uint32_t pos = atomicAdd(&s_flist[FL_CAP], 1u);
              if (pos < FL_CAP) s_flist[pos] = (lrow << 8) | lc;
            }
          }
          if constexpr (PART) {
            const uint32_t key = (w == w && gi < N) ? sa_f32_key(w) : 0xffffffffu;
            s_key[lrow * KS + lc] = key;
            if (key < ckey[n]) { ckey[n] = key; crow[n] = gi; }
          } else {
            if (gi < N && gj < TK) S.vis[(size_t)gi * TK + gj] = w;
          }
        }
      }
      if constexpr (PART) {
        if (m + 1 == TM) {
#pragma unroll
          for (int n = 0; n < TN; ++n) {
            unsigned long long b2 = ((unsigned long long)ckey[n] << 32) | crow[n];
            const unsigned long long ob = __shfl_xor(b2, 32);
            b2 = ob < b2 ? ob : b2;
            if (lh == 0 && (uint32_t)(b2 >> 32) != 0xffffffffu) atomicMin(&s_ck[wn * (BN / 2) + n * 32 + lr], b2);
          }
        }
        euclid_fixup(m);
        rows_to_partials(m);
        if (m + 1 < TM) __syncthreads();  // the next pass overwrites the key tile
      } else euclid_fixup(m);
    }
  }
  if constexpr (PART) {
    // column partials, in k_bestfit_tile's layout with this plan's tile grid (S.CT = column tiles, S.RT = row tiles): complete
    // since the barrier of the last row pass.  (No max_dist slot in this mode: k_bestfit_resolve compares the weights themselves.)
    for (uint32_t i = tid; i < (uint32_t)BN; i += blockDim.x) {
      const uint32_t gj = n0 + i;
      if (gj >= TK) continue;
      const unsigned long long k2 = s_ck[i];
      const bool has = k2 != ~0ull;
      if (p.vote_words) {
        if (has) __hip_atomic_fetch_min(S.col_best + gj, k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      S.col_part_w[(size_t)by * TK + gj] = has ? (double)sa_key_f32((uint32_t)(k2 >> 32)) : -1.0;
      S.col_part_q[(size_t)by * TK + gj] = has ? (uint32_t)k2 : SA_NONE;
    }
  }
  SA_STAMP(tr, 4);
  if constexpr (!PART) block_max_key(S.vis_max_key, key_slot, kmax);
  SA_STAMP(tr, 5);
}

// A 64 x 96 tile of the fused first phase: cosine, one observation per track, the BestFit vote reduced into the vote words (raw candidate
// rows: see visual_cosine_tile's RAW / PART mode, whose cells, keys and tie rules these are).  For frames whose 64 x 64 tiles would be one and
// a half rounds of the chip — 1000 detections against 1100 .. 1536 tracks: 272 .. 384 tiles, every CU that gets two of them runs twice as
// long as the others — 256 tiles of one and a half times the work each leave nothing unbalanced (sa_launch_frame_visual picks the form).
// Wave (wm, x) evaluates its 32 x 32 block of columns x 32 .. (16 cells per lane) and eight rows' worth of columns 64 .. 95 (8 cells).
__device__ __forceinline__ void visual_tile96(const SceneDev& S, const SaParams& p, uint32_t bx, uint32_t by, float* lds) {
  constexpr int BM = 64, BN = 96;
  uint64_t* tr = SA_TRACE_PTR();
  SA_STAMP(tr, 0);
  const uint32_t N = S.N, TK = S.TK, K = S.K;
  const uint32_t m0 = by * BM, n0 = bx * BN;
  if (m0 >= N || n0 >= TK) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w4 = (tid >> 6) & 3u;
  const uint32_t wm = w4 >> 1, wn = w4 & 1u, lr = lane & 31u, lh = lane >> 5;
  // row operands, fetched before the contraction (thread r < 64 holds row r): what frame_prep_block derives for the candidate
  float pre_us = 0.f;
  sa_geo pre_g{0.f, 0.f, 0.f, 0.f};
  RawRow pre_raw{0.f, 0.f, 1.f, 1.f, 0.f, 0.f, 0u};
  bool pre_in = false;
  if (tid < (uint32_t)BM && m0 + tid < N) {
    const uint32_t gi = m0 + tid;
    pre_raw = raw_row_fetch(S, gi);   // (loads only: raw_row_usable behind the main loop)
    pre_in = true;
  }
  GemmCols col[2];   // [0]: column x 32 + lr of the tile, [1]: column 64 + lr
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const uint32_t gj = n0 + (n == 0 ? wn * 32u : 64u) + lr;
    col[n].ok = false;
    col[n].nb = 0.f;
    col[n].cmax = -1.0f;
    col[n].g = sa_geo{0.f, 0.f, 0.f, 0.f};
    if (gj < TK) {
      const uint32_t t = gj / K;
      const float nb = S.t_fnorm[gj];
      const uint8_t pres = S.t_fpresent[gj];
      const uint32_t cnt = S.t_fcount[t];
      const uint64_t te = S.t_epoch[t];
      col[n].g = sa_ldg(S.t_geo + t);
      col[n].nb = nb;
      const uint64_t delta = S.epoch > te ? S.epoch - te : te - S.epoch;
      col[n].ok = (pres != 0) & (cnt >= p.min_track_len) & (p.max_idle >= delta);
      // (the FIRST constraint whose window holds the track's age — as a chain of selects from the last one down: a `break` on the loaded
      // epoch would hold every wave of the tile in front of its main loop for the load)
      for (uint32_t i = p.cons.n; i-- > 0;) col[n].cmax = p.cons.delta[i] >= delta ? p.cons.max_dist[i] : col[n].cmax;
    }
  }
  f32x16 acc;
  float acc2[8];
  float nsq = 0.f;
  gemm_mainloop_ks96<3, true>((gfloat_p)S.c_feat_raw, (gfloat_p)S.t_ffrag, N, TK, S.Dp, m0, n0, lds, acc, acc2, tr, &nsq, p.ks_yield);
  if (pre_in) pre_us = raw_row_usable(S, p, pre_raw, &pre_g) ? 1.f : 0.f;
  SA_STAMP(tr, 3);
  float* s_na = lds;                      // [BM]
  sa_geo* s_g = (sa_geo*)(lds + 2 * BM);  // [BM]
  float* s_np = lds + 6 * BM;             // [BM] squared norms of the candidates' rows
  unsigned long long* s_ck = (unsigned long long*)(lds + 7 * BM);  // [BN] (weight key << 32) | row, minimum per column
  constexpr uint32_t KS = BN + 4;
  uint32_t* s_key = (uint32_t*)(lds + 7 * BM + 2 * BN);            // [64][KS] order-preserving keys of the tile's cells
  if (tid < (uint32_t)BM) s_g[tid] = pre_g;
  for (uint32_t i = tid; i < (uint32_t)BN; i += blockDim.x) s_ck[i] = ~0ull;
  if (wn == 0 && lh == 0) s_np[wm * 32 + lr] = nsq;
  __syncthreads();
  if (tid < (uint32_t)BM) s_na[tid] = pre_us != 0.f ? s_np[tid] : __builtin_nanf("");  // the feature_can_be_used gate rides in the norm
  __syncthreads();
  uint32_t kmax = 0;
  // ---- the 32 x 32 block: registers 4g .. 4g+3 = tile rows wm 32 + 8g + 4 lh .. +3 ----
  {
    const uint32_t lc = wn * 32 + lr;
    f32x4 nav[4];
    uint32_t rbase[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      rbase[g] = wm * 32 + 8u * g + 4u * lh;
      nav[g] = *(const f32x4*)(s_na + rbase[g]);
    }
    uint32_t ckey = 0xffffffffu, crow = 0;
    const uint32_t cfail = constraint_mask<16>(col[0], [&](int c) { return s_g + rbase[c >> 2] + (c & 3); });
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t li = rbase[i >> 2] + (i & 3);
      const uint32_t gi = m0 + li;
      bool flagged;
      const float w = visual_cell<false>(p, acc[i], nav[i >> 2][i & 3], (cfail >> i) & 1u, col[0], &kmax, &flagged);
      const uint32_t key = (w == w && gi < N) ? sa_f32_key(w) : 0xffffffffu;
      s_key[li * KS + lc] = key;
      if (key < ckey) { ckey = key; crow = gi; }  // rows ascend with i: the lowest row wins ties
    }
    unsigned long long cb = ((unsigned long long)ckey << 32) | crow;
    const unsigned long long ob = __shfl_xor(cb, 32);
    cb = ob < cb ? ob : cb;
    if (lh == 0 && (uint32_t)(cb >> 32) != 0xffffffffu) atomicMin(&s_ck[lc], cb);
  }
  // ---- this wave's half of columns 64 .. 95: accumulator registers 8 wn .. 8 wn + 7 ----
  {
    const uint32_t lc = 64 + lr;
    f32x4 nav[2];
    uint32_t rbase[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      rbase[g] = wm * 32 + 8u * (2u * wn + g) + 4u * lh;
      nav[g] = *(const f32x4*)(s_na + rbase[g]);
    }
    uint32_t ckey = 0xffffffffu, crow = 0;
    const uint32_t cfail = constraint_mask<8>(col[1], [&](int c) { return s_g + rbase[c >> 2] + (c & 3); });
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t li = rbase[i >> 2] + (i & 3);
      const uint32_t gi = m0 + li;
      bool flagged;
      const float w = visual_cell<false>(p, acc2[i], nav[i >> 2][i & 3], (cfail >> i) & 1u, col[1], &kmax, &flagged);
      const uint32_t key = (w == w && gi < N) ? sa_f32_key(w) : 0xffffffffu;
      s_key[li * KS + lc] = key;
      if (key < ckey) { ckey = key; crow = gi; }
    }
    unsigned long long cb = ((unsigned long long)ckey << 32) | crow;
    const unsigned long long ob = __shfl_xor(cb, 32);
    cb = ob < cb ? ob : cb;
    if (lh == 0 && (uint32_t)(cb >> 32) != 0xffffffffu) atomicMin(&s_ck[lc], cb);   // (the column's other rows: wave (wm, 1 - wn) and the other wave row)
  }
  SA_STAMP(tr, 6);
  __syncthreads();  // the key tile and the column minima complete
  {
    // a row's lightest weight over the tile's columns (lowest column on ties): four threads per row, 24 columns each
    const uint32_t rr = tid >> 2, seg = tid & 3u;
    const uint32_t* kp = s_key + rr * KS + seg * 24u;
    uint32_t bk = 0xffffffffu, bc = 0;
#pragma unroll
    for (uint32_t c4 = 0; c4 < 24u; c4 += 4) {
      const uint4 v = *(const uint4*)(kp + c4);
      if (v.x < bk) { bk = v.x; bc = c4; }
      if (v.y < bk) { bk = v.y; bc = c4 + 1; }
      if (v.z < bk) { bk = v.z; bc = c4 + 2; }
      if (v.w < bk) { bk = v.w; bc = c4 + 3; }
    }
    bc += seg * 24u;
    auto take = [&](uint32_t ok, uint32_t oc) { if (ok < bk || (ok == bk && oc < bc)) { bk = ok; bc = oc; } };
    take((uint32_t)__builtin_amdgcn_mov_dpp((int)bk, 0xB1, 0xF, 0xF, true), (uint32_t)__builtin_amdgcn_mov_dpp((int)bc, 0xB1, 0xF, 0xF, true));
    take((uint32_t)__builtin_amdgcn_mov_dpp((int)bk, 0x4E, 0xF, 0xF, true), (uint32_t)__builtin_amdgcn_mov_dpp((int)bc, 0x4E, 0xF, 0xF, true));
    const uint32_t gi = m0 + rr;
    if (seg == 0 && gi < N && bk != 0xffffffffu)
      __hip_atomic_fetch_min(S.row_best + gi, ((unsigned long long)bk << 32) | (n0 + bc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  SA_STAMP(tr, 7);
  for (uint32_t i = tid; i < (uint32_t)BN; i += blockDim.x) {
    const uint32_t gj = n0 + i;
    if (gj >= TK) continue;
    const unsigned long long k2 = s_ck[i];
    if (k2 != ~0ull) __hip_atomic_fetch_min(S.col_best + gj, k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  SA_STAMP(tr, 4);
  SA_STAMP(tr, 5);
}

// Deeper banks (K = 2 .. SA_CLS_MAXK observations per track) WITHOUT the N x T x K weight matrix and without k_bestfit_tile: the
// whole-track tile of the fused frame launch.  A 64-column tile of the contraction holds the observations of floor(64 / K) WHOLE
// tracks (its first column is bank row bx floor(64 / K) K: the B operand stays one contiguous run of rows; the 64 mod K columns left
// over are computed and ignored — 1 of 64 at K = 3, 4 at K = 5), so every BestFit group (candidate, track) of the tile is complete
// inside it: the cells' weights go to an LDS tile instead of memory, and one thread per group adds the present ones up (f64, k
// ascending like the reference's sum) and counts them.  BestFit ranks a group by W = sum_k f64(max_dist - w_k) and max_dist is known
// only when every tile is done — but among groups with the SAME count c the heaviest is the one with the smallest sum.  So the tile
// reduces its groups per count class: 64-bit LDS minima of (order-preserving key of the f32 sum << 32 | index: lowest index on ties,
// as the reference orders exact ties) per row and per track, folded into the class words S.row_cls / S.col_cls with 64-bit atomic
// minima; the one-workgroup tail, which folds max_dist anyway, compares a row's (track's) class winners by W = c max_dist - sum.
// (W differs from the reference's sum of f32 differences by <= c/2 ulp of max_dist, 2e-7: the distances themselves are good to 1e-5.)
// The tiles stay as many and as independent as before (three per CU in flight at C2's size — a version that kept one 64 x 64 tile of
// (candidate, track) pairs per workgroup and ran the main loop K times lost exactly that: 44.7 us against 38.6 for the first phase).
// RAW rows only (the fused launch); EU: the flagged cells are recomputed directly as in euclid_fixup, into the LDS tile.
template <bool EU, bool KSL = false>
__device__ __forceinline__ void visual_ktile(const SceneDev& S, const SaParams& p, uint32_t bx, uint32_t by, float* lds) {
  constexpr int BM = 64, BN = 64, R = 16, G = 4;
  const uint32_t N = S.N, T = S.T, K = S.K, TK = S.TK;
  const uint32_t TPT = 64u / K, used = TPT * K;       // whole tracks per tile, columns they occupy
  const uint32_t m0 = by * BM, t0 = bx * TPT, n0 = t0 * K;
  if (m0 >= N || t0 >= T) return;
  const uint32_t key_slot = by * ((T + TPT - 1) / TPT) + bx;  // < S.nkeys (tiles of this scene in this mode)
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w4 = (tid >> 6) & 3u;
  const uint32_t wm = w4 >> 1, wn = w4 & 1u, lr = lane & 31u, lh = lane >> 5;
  float pre_us = 0.f;
  sa_geo pre_g{0.f, 0.f, 0.f, 0.f};
  RawRow pre_raw{0.f, 0.f, 1.f, 1.f, 0.f, 0.f, 0u};
  bool pre_in = false;
  if (tid < (uint32_t)BM && m0 + tid < N) {
    const uint32_t gi = m0 + tid;
    pre_raw = raw_row_fetch(S, gi);   // (loads only: raw_row_usable behind the main loop)
    pre_in = true;
  }
  const uint32_t lc = wn * 32 + lr, gj = n0 + lc;
  GemmCols col;
  col.ok = false; col.nb = 0.f; col.cmax = -1.0f; col.g = sa_geo{0.f, 0.f, 0.f, 0.f};
  if (lc < used && gj < TK) {
    const uint32_t t = gj / K;
    const float nb = S.t_fnorm[gj];
    const uint8_t pres = S.t_fpresent[gj];
    const uint32_t cnt = S.t_fcount[t];
    const uint64_t te = S.t_epoch[t];
    col.g = sa_ldg(S.t_geo + t);
    col.nb = nb;
    const uint64_t delta = S.epoch > te ? S.epoch - te : te - S.epoch;
    col.ok = (pres != 0) & (cnt >= p.min_track_len) & (p.max_idle >= delta);
    for (uint32_t i = p.cons.n; i-- > 0;) col.cmax = p.cons.delta[i] >= delta ? p.cons.max_dist[i] : col.cmax;   // (selects, no break: see visual_cosine_tile)
  }
  f32x16 acc[1][1];
  float nsq = 0.f;
  if constexpr (KSL) gemm_mainloop_ks<4, true, true>((gfloat_p)S.c_feat_raw, (gfloat_p)S.t_ffrag, N, TK, S.Dp, m0, n0, lds, acc[0][0], nullptr, &nsq);
  else gemm_mainloop<BM, BN, 1, true>((gfloat_p)S.c_feat_raw, (gfloat_p)S.t_feat, N, TK, S.Dp, m0, n0, lds, acc, nullptr, &nsq);
  if (pre_in) pre_us = raw_row_usable(S, p, pre_raw, &pre_g) ? 1.f : 0.f;
  float* s_na = lds;                      // [BM]
  sa_geo* s_g = (sa_geo*)(lds + 2 * BM);  // [BM]
  float* s_np = lds + 6 * BM;             // [BM] squared norms of the candidates' rows (raw mode)
  constexpr uint32_t KS = BN + 4;
  uint32_t* s_w = (uint32_t*)(lds + 7 * BM);                       // [64][KS] the cells' weights (f32 bits, NaN = absent)
  constexpr uint32_t FW = BN / 32;
  uint32_t* s_flag = (uint32_t*)(lds + 7 * BM + 64 * KS);
  constexpr uint32_t FL_CAP = 255;
  uint32_t* s_flist = s_flag + 64 * FW;
  unsigned long long* s_rc = (unsigned long long*)(s_flist + FL_CAP + 1);  // [64][K] row class words of the tile
  unsigned long long* s_cc = s_rc + 64 * SA_CLS_MAXK;                      // [TPT][K] track class words of the tile
  static_assert((7 * BM + 64 * (BN + 4) + 64 * (BN / 32) + 256 + 2 * 64 * SA_CLS_MAXK + 2 * 64) <= 2 * (BM + BN) * BK, "the epilogue's tables must fit the stages");
  if constexpr (EU) {
    for (uint32_t i = tid; i < 64u * FW; i += blockDim.x) s_flag[i] = 0u;
    if (tid == 0) s_flist[FL_CAP] = 0u;
  }
  for (uint32_t i = tid; i < 64u * K; i += blockDim.x) s_rc[i] = ~0ull;
  if (tid < used) s_cc[tid] = ~0ull;
  if (tid < (uint32_t)BM) s_g[tid] = pre_g;
  if constexpr (!KSL) nsq += __shfl_xor(nsq, 32);
  if (wn == 0 && lh == 0) s_np[wm * 32 + lr] = nsq;
  __syncthreads();
  if (tid < (uint32_t)BM) s_na[tid] = pre_us != 0.f ? s_np[tid] : __builtin_nanf("");  // the feature_can_be_used gate rides in the norm
  __syncthreads();
  uint32_t rbase[G];
  f32x4 nav[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    rbase[g] = wm * 32 + 8u * g + 4u * lh;
    nav[g] = *(const f32x4*)(s_na + rbase[g]);
  }
  uint32_t kmax = 0;
  const uint32_t cfail = constraint_mask<R>(col, [&](int c) { return s_g + rbase[c >> 2] + (c & 3); });
  uint32_t fmask = 0;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const uint32_t li = rbase[i >> 2] + (i & 3);
    const uint32_t gi = m0 + li;
    bool flagged;
    float w = visual_cell<EU>(p, acc[0][0][i], nav[i >> 2][i & 3], (cfail >> i) & 1u, col, &kmax, &flagged);
    if (gi >= N) w = __builtin_nanf("");  // (columns past the tile's tracks: col.ok = false)
    if constexpr (EU) fmask |= (flagged && gi < N) ? (1u << i) : 0u;   // (registered behind the loop: see visual_cosine_tile)
    s_w[li * KS + lc] = __float_as_uint(w);
  }
  if constexpr (EU) {
    if (__ballot(fmask != 0u) != 0ull) {
      while (fmask) {
        const uint32_t i = (uint32_t)__builtin_ctz(fmask);
        fmask &= fmask - 1u;
        const uint32_t li = wm * 32 + 8u * (i >> 2) + 4u * lh + (i & 3u);
        atomicOr(&s_flag[li * FW + (lc >> 5)], 1u << (lc & 31u));
        const uint32_t pos = atomicAdd(&s_flist[FL_CAP], 1u);
        if (pos < FL_CAP) s_flist[pos] = (li << 8) | lc;
      }
    }
    __syncthreads();  // flag words / list and the weight tile complete
    const uint32_t wave = tid >> 6, nw = blockDim.x >> 6;
    auto recompute = [&](uint32_t li, uint32_t lc2) {
      const uint32_t gi = m0 + li;
      const float SA_G* a = S.c_feat_raw + (size_t)gi * S.Dp;
      const float SA_G* b = KSL ? S.t_ffrag + sa_frag_index(n0 + lc2, 0, S.Dp) : S.t_feat + (size_t)(n0 + lc2) * S.Dp;   // (the twin: L2-resident)
      float acc2 = 0.f;
      for (uint32_t kk = lane * 4u; kk < S.Dp; kk += 256u) {
        const f32x4 x = *(const f32x4 SA_G*)(a + kk), y = *(const f32x4 SA_G*)(b + (KSL ? (size_t)(kk >> 3) * 256u + ((kk >> 2) & 1u) * 128u : (size_t)kk));
        const float d0 = x[0] - y[0], d1 = x[1] - y[1], d2 = x[2] - y[2], d3 = x[3] - y[3];
        acc2 += d0 * d0; acc2 += d1 * d1; acc2 += d2 * d2; acc2 += d3 * d3;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc2 += __shfl_xor(acc2, o);
      if (lane == 0) {
        const float d = __fsqrt_rn(acc2);
        const bool ok = d <= p.visual_threshold;
        const uint32_t key = ok ? sa_f32_key(d) : 0u;
        s_w[li * KS + lc2] = __float_as_uint(ok ? d : __builtin_nanf(""));
        kmax = key > kmax ? key : kmax;
      }
    };
    const uint32_t nf = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_flist[FL_CAP]);
    if (nf <= FL_CAP) {
      for (uint32_t i = wave; i < nf; i += nw) {
        const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_flist[i]);
        recompute(ent >> 8, ent & 255u);
      }
    } else {
      for (uint32_t lrow = wave; lrow < 64u; lrow += nw)
        for (uint32_t wd = 0; wd < FW; ++wd) {
          uint32_t bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_flag[lrow * FW + wd]);
          while (bits) {
            const uint32_t c2 = wd * 32u + (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1u;
            recompute(lrow, c2);
          }
        }
    }
    if (nf > 64u && tid == 0) S.stats[0] = 1u;  // ill-conditioned for the expansion (see euclid_fixup)
  }
  __syncthreads();  // the weight tile (and the reset class words) complete
  // one thread per group (row-major over the tile's tracks: a wave's LDS reads fall K words apart, conflict-free)
  const uint32_t mv = p.min_votes > 1u ? p.min_votes : 1u;
  for (uint32_t g2 = tid; g2 < 64u * TPT; g2 += blockDim.x) {
    const uint32_t row = g2 / TPT, tr = g2 % TPT;
    const uint32_t gi = m0 + row, gt = t0 + tr;
    if (gi >= N || gt >= T) continue;
    const uint32_t* wp = s_w + row * KS + tr * K;
    double sum = 0.0;
    uint32_t cnt = 0;
    for (uint32_t k = 0; k < K; ++k) {
      const float w = __uint_as_float(wp[k]);
      if (w == w) { sum += (double)w; ++cnt; }
    }
    if (cnt >= mv) {
      const unsigned long long key = (unsigned long long)sa_f32_key((float)sum) << 32;
      atomicMin(&s_rc[row * K + cnt - 1u], key | gt);
      atomicMin(&s_cc[tr * K + cnt - 1u], key | gi);
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < 64u * K; i += blockDim.x) {
    const unsigned long long v = s_rc[i];
    const uint32_t gi = m0 + i / K;
    if (v != ~0ull && gi < N) __hip_atomic_fetch_min(S.row_cls + (size_t)gi * K + i % K, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < used) {
    const unsigned long long v = s_cc[tid];
    const uint32_t gt = t0 + tid / K;
    if (v != ~0ull && gt < T) __hip_atomic_fetch_min(S.col_cls + (size_t)gt * K + tid % K, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  block_max_key(S.vis_max_key, key_slot, kmax);
}

// (Tile order: row by row.  An XCD-aware band order — each XCD's L2 keeping one set of candidate panels — was measured at C5 with bands
// of 1, 2 and 4 tile rows: no difference, the 114 MB working set sits in the 256 MB Infinity Cache and the fabric keeps up.)
template <int BM, int BN, int KGT, bool PART = false, bool EU = false>
__global__ __launch_bounds__(256 * ((KGT == 9 || KGT == 15 || KGT == 17) ? 1 : KGT ? KGT : 1), (KGT == 15 && BM == 128 && BN == 128) ? 2 : 1) void k_visual_cosine(const SceneDev* __restrict__ scenes, SaParams p, uint32_t gx, uint32_t gy,
                                                                          uint32_t xo_) {
  __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats(BM, BN, KGT, PART, EU)];
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  uint32_t bx, by;  // XCD-aware tile order (xcd_order): a 1-D grid of 8 chunk workgroups per scene
  if (!xcd_tile(blockIdx.x, gx, gy, xo_ >> 8, xo_ & 255u, &bx, &by)) return;
  visual_cosine_tile<BM, BN, KGT, false, PART, EU>(S, p, bx, by, lds);
}
// (the tiles of the stand-alone contraction in XCD-aware order)
template <int BM, int BN, int KGT, bool PART, bool EU>
static void launch_cosine(uint32_t maxTK, uint32_t maxN, uint32_t ns, hipStream_t st, const SceneDev* scenes, const SaParams& p) {
  const uint32_t gx = cdiv(maxTK, BN), gy = cdiv(maxN, BM);
  const XcdOrder xo = xcd_order(gx, gy, p.row_major_tiles != 0);
  SA_LAUNCH((k_visual_cosine<BM, BN, KGT, PART, EU>), dim3(xo.W ? 8u * xo.chunk : xo.chunk, 1, ns), dim3(256 * ((KGT == 9 || KGT == 15 || KGT == 17) ? 1 : KGT ? KGT : 1)), 0, st, scenes, p, gx, gy, (xo.chunk << 8) | xo.W);
}

// The whole first phase of a VisualSORT frame in ONE heterogeneous launch: blockIdx.x <
//   n_gemm            : a 64x64 tile of the feature contraction (matrix cores; raw-feature mode, see visual_cosine_tile)
//   n_gemm + n_prep   : a frame-preparation block (padded features + norms for the upkeep and the taps, vote-state reset)
//   ...               : a 16x256 positional tile (f64 VALU + LDS: pair pre-filter, disjointness proofs, polygon clipping, edges)
// The three kinds are independent of each other, so the positional tiles and the preparation blocks fill the issue slots and
// the LDS the MFMA-bound contraction leaves idle on every CU instead of costing two more dependent launches.  Tiles are
// dispatched in blockIdx order: the contraction's (longest) first.  All kinds share ONE static LDS buffer (a kernel's
// static LDS is the sum of its arrays: separate arrays would cut the residency to one block per CU and serialise the kinds).
template <int KG, bool PART, bool EU = false, bool KP = false, bool KSL = false, bool W96 = false>
__global__ __launch_bounds__(256 * KG, W96 ? 4 : 1) void k_frame_visual(const SceneDev* __restrict__ scenes, SaParams p, uint32_t gx, uint32_t gy,
                                                           uint32_t px, uint32_t py, uint32_t nprep_, uint32_t xo_) {
  // nprep_: preparation blocks of the launch; bit 31: they run their RESET half only, bit 30: the positional tiles also feed the
  // many-workgroup tail (row-major edge lists, row duals, union-find: UNION) — frames beyond the one-workgroup tail's 1024 x 1024
  // bit 29: the positional tiles are 16 x 256 (launches whose blocks would not all be resident with 16 x 128 tiles: sa_launch_frame_visual)
  const uint32_t nprep = nprep_ & 0x1fffffffu;
  const bool prep_light = (nprep_ >> 31) != 0, uni = ((nprep_ >> 30) & 1u) != 0, wide = ((nprep_ >> 29) & 1u) != 0;
  using FusedPos = PosSmem<2, 64>;  // the wide, proof-filtered positional tile of this launch (sa_frame.h)
  using FusedPosW = PosSmem<4, 64>; // ... and its 16 x 256 form
  static_assert(sizeof(FusedPos) <= sizeof(float) * 2 * 128 * BK, "the positional tile must fit one k-group's stages");
  constexpr uint32_t POS_LDS = (sizeof(FusedPosW) + 15u) & ~15u;
  constexpr uint32_t LDSF = (KG * POS_LDS + 3u) / 4u > (uint32_t)(KG * 2 * (64 + 64) * BK) ? (KG * POS_LDS + 3u) / 4u : (uint32_t)(KG * 2 * (64 + 64) * BK);
  __shared__ __attribute__((aligned(16))) float lds[LDSF];   // (35.6 KB with the 16 x 256 positional tile: four blocks per CU as before)
  static_assert(gemm_lds_floats(64, 64, KSL ? 9 : KG, PART, EU) <= LDSF, "the contraction tile must fit the launch's LDS");
  static_assert(!W96 || (KSL && PART && !EU && !KP && KG == 1 && SA_KS96_LDS <= LDSF && 7 * 64 + 2 * 96 + 64 * 100 <= LDSF), "the 64 x 96 tile: cosine vote-word frames on the k-split loop");
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  // Contraction tiles first in blockIdx order: the dispatcher hands blocks out in that order, breadth-first over the CUs, so
  // every CU starts with (at most) one contraction tile and fills its remaining slots with the other kinds.  Interleaving the
  // kinds (one contraction tile every k blocks) was measured: 32-50 us instead of 22.6; the other kinds FIRST (so that the positional tiles of a
  // frame with several rounds of contraction tiles do not queue behind them): C2 16.8 -> 17.5 us, three observations per track 40.9 -> 48.1.
  // xo_ = XcdOrder: chunk << 8 | W — the contraction's tiles in XCD-aware order (8 chunk workgroup slots, the last few possibly idle)
  uint32_t b = blockIdx.x;
  const uint32_t xchunk = xo_ >> 8, xW = xo_ & 255u;
  if (b < (xW ? 8u * xchunk : xchunk)) {
    uint32_t tbx, tby;
    if (!xcd_tile(b, gx, gy, xchunk, xW, &tbx, &tby)) return;
    if constexpr (W96) visual_tile96(S, p, tbx, tby, lds);         // (gx counts tiles of 96 columns here)
    else if constexpr (KP) visual_ktile<EU, KSL>(S, p, tbx, tby, lds);  // (gx counts tiles of floor(64 / K) whole tracks here)
    else visual_cosine_tile<64, 64, KSL ? 9 : KG, true, PART, EU>(S, p, tbx, tby, lds);
    return;
  }
  b -= xW ? 8u * xchunk : xchunk;
  // the other two kinds are 256-thread units: a block of KG * 256 threads runs KG of them side by side, each in its own part of the
  // LDS buffer.  Their barriers are the block's; units pair up barrier for barrier (same kind: same count), and a unit that has
  // nothing to do, or none, simply ends — ended waves do not take part in s_barrier.
  const uint32_t unit = b * KG + (threadIdx.x >> 8), tid = threadIdx.x & 255u;
#ifdef SA_GEMM_TRACE
  uint64_t* tr2 = g_trace_dev && blockIdx.x < 65536 ? g_trace_dev + 8 * blockIdx.x : nullptr;  // entry / exit of the other kinds
  if (tr2 && threadIdx.x == 0) { tr2[0] = __builtin_amdgcn_s_memtime(); tr2[1] = unit < nprep ? 2 : 1; }
#endif
  // preparation blocks (short) before the positional tiles (long): the tiles alone fill every slot the contraction leaves, and
  // preparation blocks queued behind them started only when the first tiles retired — the last thing to finish in the launch
  if (unit < nprep) frame_prep_block(S, p, unit, tid, prep_light);
  else if (unit - nprep < px * py) {
    unsigned char* pl = (unsigned char*)lds + (threadIdx.x >> 8) * POS_LDS;
    if (wide) positional_tile<false, true, 4, false, true, 64, true>(S, p, (unit - nprep) % px, (unit - nprep) / px, pl, tid);   // (never with the many-workgroup tail's extras)
    else if (uni) positional_tile<false, true, 2, true, true, 64, true>(S, p, (unit - nprep) % px, (unit - nprep) / px, pl, tid);
    else positional_tile<false, true, 2, false, true, 64, true>(S, p, (unit - nprep) % px, (unit - nprep) / px, pl, tid);
  }
#ifdef SA_GEMM_TRACE
  if (tr2 && threadIdx.x == 0) tr2[5] = __builtin_amdgcn_s_memtime();
#endif
}

// Direct sum (a-b)^2 on the vector pipe.  A 512-thread block computes 32 rows x 128 columns: wave w owns rows 4w .. 4w+3, lane l
// the columns l and l + 64 — 8 cells per thread, two partial sums per cell (even / odd k) so that every step is one packed
// subtract and one packed fused multiply-add (v_pk_add_f32 / v_pk_fma_f32: two f32 per lane per issue, 4-5 cycles per wave
// instruction measured, scripts/micro/pk_fma_rate.hip; the pipe's 157 TF/s is quoted on these).  What bounds such a kernel is
// operand delivery, not arithmetic: the first version (64 x 64 tile, 4 x 4 cells per thread, both operands through LDS) needs
// 8 ds_read_b128 per 64 packed instructions — exactly the 128 B/clk the LDS has when the vector pipe runs at full rate, so
// neither did (61.7 us at C2's size with scalar arithmetic, 38.7 packed; two LDS stages with fragment prefetch: 49.5).  Here the
// A operand never touches LDS or a vector register: a wave's 4 rows are the same for all its lanes, so their k-runs come through
// the SCALAR cache (s_load_dwordx8 from the constant address space) and enter the packed subtract as SGPR pairs (inline asm:
// left to itself the compiler subtracts the halves separately); only B goes through LDS, four ds_read_b128 per 64 packed
// instructions.  (Eight rows per wave halve that again but need 128 SGPRs for the two operand buffers: spilled through
// v_writelane.)  Eight waves per block: two per SIMD even when the frame is one block per CU.  Column stores of a wave cover
// 256-byte row segments.  35.3 us at C2's size (0.28 of the vector peak); with the scalar loads or the LDS reads taken out
// (wrong answers) 25.5 either way — what is left over the ~16 us of issue time is the two operand streams' round trips, which
// share one counter (lgkmcnt) and can only be awaited together; one column per lane (four waves per SIMD) 37.0, 4-float steps 35.7.
constexpr int EU_C = 2;                                          // columns per lane
constexpr int EU_BM = 32, EU_BN = 64 * EU_C, EU_R = 4, EU_THREADS = 512;  // EU_R rows per wave, EU_BM / EU_R waves
constexpr int EU_LDS_FLOATS = 2 * EU_BN * BK;
typedef const f32x4 __attribute__((address_space(4)))* kf32x4_p;
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef const f32x8 __attribute__((address_space(4)))* kf32x8_p;
__device__ __forceinline__ void euclid_mainloop(gfloat_p A, gfloat_p B, uint32_t M,
                                                uint32_t Ncols, uint32_t Dp, uint32_t m0, uint32_t n0, float* lds,
                                                float (&acc)[EU_R][EU_C]) {
  constexpr int STAGE = EU_BN * BK;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  f32x2 acc2[EU_R][EU_C];
#pragma unroll
  for (int i = 0; i < EU_R; ++i)
#pragma unroll
    for (int j = 0; j < EU_C; ++j) acc2[i][j] = f32x2{0.f, 0.f};
  // B rows of this block: 128 columns x 32 k per chunk = 1024 16-byte pieces, two per thread
  constexpr int NL = EU_BN * 8 / EU_THREADS;  // 16-byte pieces per thread per chunk
  gfloat_p pb[NL];
  uint32_t so[NL];
#pragma unroll
  for (int r = 0; r < NL; ++r) {
    const uint32_t c = tid + 512u * r, row = c >> 3, kc = c & 7u;
    uint32_t gb = n0 + row;
    gb = gb < Ncols ? gb : Ncols - 1;
    pb[r] = B + (size_t)gb * Dp + kc * 4u;
    so[r] = lds_off(row, kc);
  }
  // A rows of this wave (rows past the edge are clamped; their results are never stored)
  kf32x4_p ka[EU_R];
#pragma unroll
  for (int i = 0; i < EU_R; ++i) {
    uint32_t ga = m0 + (uint32_t)EU_R * w + i;
    ga = ga < M ? ga : M - 1;
    ka[i] = (kf32x4_p)(uintptr_t)(A + (size_t)ga * Dp);
  }
  const uint32_t nchunks = Dp / BK;
  f32x4 rb[NL];
#pragma unroll
  for (int r = 0; r < NL; ++r) rb[r] = *(gf32x4_p)pb[r];
#pragma unroll
  for (int r = 0; r < NL; ++r) *(f32x4*)(lds + so[r]) = rb[r];
  if (nchunks > 1) {
#pragma unroll
    for (int r = 0; r < NL; ++r) rb[r] = *(gf32x4_p)(pb[r] + BK);
  }
  __syncthreads();
  // one step = 8 k: four rows' runs as s_load_dwordx8 (32 SGPRs per buffer), two ds_read_b128 per column, 16 * EU_C packed
  // subtract / multiply-add pairs — long enough (~300 cycles) to cover the LDS and scalar-cache round trips of the next step's
  // operands, which share one counter (lgkmcnt) and can only be awaited together
  f32x4 fb[2][EU_C][2];
  f32x8 sa[2][EU_R];
  auto fetch = [&](auto buf_tag, const float* Bs, uint32_t k8) {  // k8 = index of the 8-float run along k (global)
    constexpr int buf = decltype(buf_tag)::value;
#pragma unroll
    for (int j = 0; j < EU_C; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) fb[buf][j][h] = *(const f32x4*)(Bs + lds_off(lane + 64u * j, (2u * k8 + h) & 7u));
#pragma unroll
    for (int i = 0; i < EU_R; ++i) sa[buf][i] = ((kf32x8_p)ka[i])[k8];
  };
  auto compute = [&](auto cur_tag) {
    constexpr int cur = decltype(cur_tag)::value;
#pragma unroll
    for (int e = 0; e < 8; e += 2)
#pragma unroll
      for (int i = 0; i < EU_R; ++i)
#pragma unroll
        for (int j = 0; j < EU_C; ++j) {
          // a - b as ONE packed instruction with the SGPR pair as a source (left to itself the compiler subtracts the two
          // halves separately: 96 instructions per step instead of 64)
          const f32x2 av = f32x2{sa[cur][i][e], sa[cur][i][e + 1]};
          const f32x2 bv = f32x2{fb[cur][j][e >> 2][e & 3], fb[cur][j][e >> 2][(e & 3) + 1]};
          f32x2 df;
          asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(df) : "s"(av), "v"(bv));
          acc2[i][j] = __builtin_elementwise_fma(df, df, acc2[i][j]);
        }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  fetch(B0{}, lds, 0);
  for (uint32_t c = 0; c < nchunks; ++c) {
    const float* Bs = lds + (c & 1u) * STAGE;
    float* nxt = lds + ((c + 1u) & 1u) * STAGE;
    const bool store = c + 1 < nchunks, load = c + 2 < nchunks;
    // two steps per trip of a ROLLED loop: the operand buffers stay compile-time constants, and the scheduler cannot hoist a
    // whole chunk's reads to the top
#pragma unroll 1
    for (uint32_t ks = 0; ks < 4; ks += 2) {
      fetch(B1{}, Bs, c * 4u + ks + 1);
      compute(B0{});
      if (ks == 0 && store) {
#pragma unroll
        for (int r = 0; r < NL; ++r) *(f32x4*)(nxt + so[r]) = rb[r];
        if (load) {
#pragma unroll
          for (int r = 0; r < NL; ++r) rb[r] = *(gf32x4_p)(pb[r] + (c + 2) * BK);
        }
      }
      if (ks + 2 < 4) fetch(B0{}, Bs, c * 4u + ks + 2);
      compute(B1{});
    }
    __syncthreads();  // the other stage is complete, and everyone is done reading this one
    if (store) fetch(B0{}, nxt, (c + 1) * 4u);
  }
#pragma unroll
  for (int i = 0; i < EU_R; ++i)
#pragma unroll
    for (int j = 0; j < EU_C; ++j) acc[i][j] = acc2[i][j][0] + acc2[i][j][1];
}

__global__ __launch_bounds__(EU_THREADS) void k_visual_euclid(const SceneDev* __restrict__ scenes, SaParams p) {
  constexpr int BM = EU_BM, BN = EU_BN;
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  const uint32_t N = S.N, TK = S.TK, K = S.K;
  const uint32_t m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= N || n0 >= TK) return;
  __shared__ __attribute__((aligned(16))) float lds[EU_LDS_FLOATS];
  float acc[EU_R][EU_C];
  euclid_mainloop((gfloat_p)S.c_feat, (gfloat_p)S.t_feat, N, TK, S.Dp, m0, n0, lds, acc);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const float nanv = __builtin_nanf("");
  const uint64_t epoch = S.epoch;
  uint32_t kmax = 0;
  // vote words (SaParams::vote_words; one observation per track, frame up to 1024 x 1024): no weight matrix — every row's and every
  // column's lightest weight, (order-preserving key << 32) | index, goes into S.row_best / S.col_best by 64-bit atomic minima, as
  // from the cosine contraction's epilogue.  Rows: a wave holds whole row segments (64 lanes x EU_C columns), so one wave
  // reduction per row; columns: in-lane over the wave's rows, then the eight waves meet in LDS (the stages are free now).
  const bool words = p.vote_words != 0;
  unsigned long long* s_col = (unsigned long long*)lds;
  unsigned long long cbest[EU_C];
#pragma unroll
  for (int j = 0; j < EU_C; ++j) cbest[j] = ~0ull;
  if (words) {
    __syncthreads();  // every wave is done with the stages
    for (uint32_t i = tid; i < (uint32_t)EU_BN; i += EU_THREADS) s_col[i] = ~0ull;
    __syncthreads();
  }
  // the two columns of this lane: their gates once, not once per row
  bool col_ok[EU_C];
  sa_geo tg[EU_C];
  uint64_t tep[EU_C];
#pragma unroll
  for (int j = 0; j < EU_C; ++j) {
    const uint32_t gj = n0 + lane + 64u * j;
    col_ok[j] = false;
    if (gj < TK) {
      const uint32_t t = gj / K;
      col_ok[j] = S.t_fpresent[gj] && S.t_fcount[t] >= p.min_track_len;
      tg[j] = sa_ldg(S.t_geo + t);
      tep[j] = S.t_epoch[t];
    }
  }
#pragma unroll
  for (int i = 0; i < EU_R; ++i) {
    const uint32_t gi = m0 + (uint32_t)EU_R * w + i;
    if (gi >= N) continue;
    const bool us = S.c_usable[gi] != 0;
    const sa_geo cg = sa_ldg(S.c_geo + gi);
#pragma unroll
    for (int j = 0; j < EU_C; ++j) {
      const uint32_t gj = n0 + lane + 64u * j;
      if (gj >= TK) continue;
      float out = nanv;
      if (us && col_ok[j] && sa_compatible(cg, epoch, tg[j], tep[j], p.max_idle, p.cons)) {
        const float d = sqrtf(acc[i][j]);
        if (d <= p.visual_threshold) {
          out = d;
          const uint32_t key = sa_f32_key(out);
          kmax = key > kmax ? key : kmax;
        }
      }
      if (!words) S.vis[(size_t)gi * TK + gj] = out;
      else acc[i][j] = out;
    }
  }
  if (!words) {
    block_max_key(S.vis_max_key, blockIdx.y * ((TK + BN - 1) / BN) + blockIdx.x, kmax);
    return;
  }
#pragma unroll
  for (int i = 0; i < EU_R; ++i) {
    const uint32_t gi = m0 + (uint32_t)EU_R * w + i;  // wave-uniform
    unsigned long long rbest = ~0ull;
    if (gi < N) {
#pragma unroll
      for (int j = 0; j < EU_C; ++j) {
        const uint32_t gj = n0 + lane + 64u * j;
        const float o = acc[i][j];
        if (gj < TK && o == o) {
          const unsigned long long k = (unsigned long long)sa_f32_key(o) << 32;
          rbest = (k | gj) < rbest ? (k | gj) : rbest;        // lowest column on ties
          cbest[j] = (k | gi) < cbest[j] ? (k | gi) : cbest[j];  // rows ascend with i: lowest row on ties
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(rbest, o);
      rbest = other < rbest ? other : rbest;
    }
    if (lane == 0 && rbest != ~0ull) __hip_atomic_fetch_min(S.row_best + gi, rbest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int j = 0; j < EU_C; ++j)
    if (cbest[j] != ~0ull) atomicMin(&s_col[lane + 64u * j], cbest[j]);
  __syncthreads();
  for (uint32_t i = tid; i < (uint32_t)EU_BN; i += EU_THREADS) {
    const unsigned long long v = s_col[i];
    if (v != ~0ull && n0 + i < TK) __hip_atomic_fetch_min(S.col_best + n0 + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- standalone distance matrix (sa_feature_distance_matrix): no gating, plain d ----
template <int BM, int BN, int KGT>
__global__ __launch_bounds__(256 * (KGT >= 9 ? 1 : KGT ? KGT : 1)) void k_cosine_matrix(const float* __restrict__ A, const float* __restrict__ an,
                                                            const float* __restrict__ B, const float* __restrict__ bn,
                                                            uint32_t M, uint32_t Ncols, uint32_t Dp, float* __restrict__ out) {
  const uint32_t m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  uint64_t* tr = SA_TRACE_PTR();
  SA_STAMP(tr, 0);
  constexpr int TM = BM / 64, TN = BN / 64;
  // KGT >= 9: the k-split main loop (gemm_mainloop_ks): 9 / 10 = B row-major / in fragment order, 13 = A in fragment order as well (the
  // stand-alone matrix entry point's measurement plans: sa_feature_distance_matrix reorders the operands it is asked to)
  constexpr bool KS = KGT >= 9 && KGT != 15 && KGT != 17;   // 15: the direct loop of the wider tiles (gemm_mainloop_direct); 17: the 64 x 128 tile's k-split loop
  constexpr int KG = KGT >= 9 ? 1 : KGT ? KGT : 1;
  static_assert(KG == 1 || (TM == 1 && TN == 1), "k-groups only with the 64x64 tile");
  static_assert(!KS || (TM == 1 && TN == 1), "k-split only with the 64x64 tile");
  __shared__ __attribute__((aligned(16))) float lds[KGT == 15 ? 64 : KGT == 17 ? 8192 : (KGT ? KG * 2 : 3) * (BM + BN) * BK];
  f32x16 acc[TM][TN];
  if constexpr (KGT == 17) gemm_mainloop_ks128<3>((gfloat_p)A, (gfloat_p)B, M, Ncols, Dp, m0, n0, lds, acc, tr);
  else if constexpr (KGT == 15) gemm_mainloop_direct<BM, BN, 4>((gfloat_p)A, (gfloat_p)B, M, Ncols, Dp, m0, n0, acc, tr);
  else if constexpr (KS) gemm_mainloop_ks<4, false, (KGT == 10 || KGT == 13), (KGT == 13)>((gfloat_p)A, (gfloat_p)B, M, Ncols, Dp, m0, n0, lds, acc[0][0], tr);
  else if constexpr (KGT == 0) gemm_mainloop_ring<BM, BN>((gfloat_p)A, (gfloat_p)B, M, Ncols, Dp, m0, n0, lds, acc, tr);
  else gemm_mainloop<BM, BN, KG>((gfloat_p)A, (gfloat_p)B, M, Ncols, Dp, m0, n0, lds, acc, tr);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w4 = (tid >> 6) & 3u, kg = tid >> 8;
  const uint32_t wm = w4 >> 1, wn = w4 & 1u, lr = lane & 31u, lh = lane >> 5;
  if constexpr (TM == 1 && TN == 1) {
    constexpr int R = 16 / KG;
    float part[R];
    kgroup_reduce_spread<KG>(acc[0][0], lds, part);
    SA_STAMP(tr, 3);
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
  SA_STAMP(tr, 4);
}

__global__ __launch_bounds__(EU_THREADS) void k_euclid_matrix(const float* __restrict__ A, const float* __restrict__ B, uint32_t M,
                                                       uint32_t Ncols, uint32_t Dp, float* __restrict__ out) {
  const uint32_t m0 = blockIdx.y * EU_BM, n0 = blockIdx.x * EU_BN;
  __shared__ __attribute__((aligned(16))) float lds[EU_LDS_FLOATS];
  float acc[EU_R][EU_C];
  euclid_mainloop((gfloat_p)A, (gfloat_p)B, M, Ncols, Dp, m0, n0, lds, acc);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
#pragma unroll
  for (int i = 0; i < EU_R; ++i) {
    const uint32_t gi = m0 + (uint32_t)EU_R * w + i;
    if (gi >= M) continue;
#pragma unroll
    for (int j = 0; j < EU_C; ++j) {
      const uint32_t gj = n0 + lane + 64u * j;
      if (gj < Ncols) out[(size_t)gi * Ncols + gj] = sqrtf(acc[i][j]);
    }
  }
}

#ifdef SA_GEMM_TRACE
// dumps the stamps of the PREVIOUS launches when the call counter reaches $SA_GEMM_TRACE
static void sa_trace_hook(hipStream_t st, uint32_t nb) {
  static uint64_t* buf = nullptr;
  static int calls = 0;
  const char* env = getenv("SA_GEMM_TRACE");
  if (!env) return;
  if (!buf) {
    hipMalloc(&buf, 8 * 8 * 65536);
    hipMemset(buf, 0, 8 * 8 * 65536);
    hipMemcpyToSymbol(HIP_SYMBOL(g_trace_dev), &buf, sizeof(void*));
  }
  if (++calls != atoi(env)) return;
  hipStreamSynchronize(st);
  if (nb > 65536) nb = 65536;
  uint64_t* h = (uint64_t*)malloc((size_t)nb * 64);
  hipMemcpy(h, buf, (size_t)nb * 64, hipMemcpyDeviceToHost);
  if (FILE* f = fopen("gpurun_out/gemm_trace.txt", "w")) {
    for (uint32_t i = 0; i < nb; ++i)
      if (h[i * 8])
        fprintf(f, "%u %llu %llu %llu %llu %llu %llu %llu %llu\n", i, (unsigned long long)h[i * 8], (unsigned long long)h[i * 8 + 1],
                (unsigned long long)h[i * 8 + 2], (unsigned long long)h[i * 8 + 3], (unsigned long long)h[i * 8 + 4],
                (unsigned long long)h[i * 8 + 5], (unsigned long long)h[i * 8 + 6], (unsigned long long)h[i * 8 + 7]);
    fclose(f);
  }
  free(h);
}
#else
static inline void sa_trace_hook(hipStream_t, uint32_t) {}
#endif

// Tile plans: 0 = 128x128, 5 = 64x128, 6 = 128x64 (4 waves, one k-group), 1/2/4 = 64x64 with 1/2/4 k-groups.
// The contraction is matrix-core bound once every SIMD holds >= 2 waves, so a CU's time is (tiles it receives) x (tile
// area); the plan minimises ceil(tiles / 256 CUs) x area x (1 + 16/BM + 16/BN) — the last factor is the measured cost of
// the shorter MFMA runs between barriers on narrower tiles.  C5 (2000 x 5000): 128x128 gives 640 tiles = 2.5 per CU
// (3 rounds of 16384 cells), 64x128 gives 1280 = 5 per CU (5 rounds of 8192 cells) — 17 % less work on the critical CU.
// Frames that fit in one round of 64x64 tiles split k over 2 or 4 wave groups inside each workgroup so that every SIMD
// still holds 2-4 waves.
static inline int tile_plan(uint32_t M, uint32_t Ncols, uint32_t ns, uint32_t Dp, int32_t plan_override = -1) {
  if (plan_override == 19) return 9;             // (19: the fused first phase's 64 x 96 tiles pinned — everything else sees the 64 x 64 k-split plan)
  if (plan_override >= 0) return plan_override;  // sa_config.gemm_plan: tuning / tests
  struct Cand { int plan, bm, bn; };
  const Cand cands[4] = {{0, 128, 128}, {5, 64, 128}, {6, 128, 64}, {1, 64, 64}};
  int best = 1;
  double best_cost = 1e300;
  for (const Cand& c : cands) {
    const size_t tiles = (size_t)cdiv(M, c.bm) * cdiv(Ncols, c.bn) * ns;
    const double rounds = (double)((tiles + 255) / 256);
    const double cost = rounds * c.bm * c.bn * (1.0 + 16.0 / c.bm + 16.0 / c.bn);
    if (cost < best_cost) { best_cost = cost; best = c.plan; }
  }
  if (best != 1) return best;
  const size_t b64 = (size_t)cdiv(M, 64) * cdiv(Ncols, 64) * ns;
  const uint32_t nchunks = Dp / BK;
  // two k-groups per tile only while a CU holds ONE tile (a lone wave per SIMD loses a third of the matrix pipe to its own LDS and memory
  // instructions, scripts/micro/mfma_side_mix.hip); from two co-resident tiles on, the second wave is there anyway and the split only adds the
  // reduction: 512 tiles (1000 x 2000 columns) 25.5 us with one group against 27.1 with two, 752 tiles (1000 x 3000) 33.0 against 40.3
  if (b64 <= 320 && nchunks >= 8) return 2;
  return 1;
}

// Tile extents the visual cost kernel will use for a batch with these maxima (the host needs them for the per-scene number of
// max-key slots, SceneDev::nkeys).
void sa_visual_tile(int visual_kind, bool eu_mfma, uint32_t maxN, uint32_t maxTK, uint32_t ns, uint32_t Dp, int32_t plan_override, uint32_t* bm, uint32_t* bn) {
  *bm = 64; *bn = 64;
  if (visual_kind == SA_VIS_EUCLIDEAN && !eu_mfma) { *bm = EU_BM; *bn = EU_BN; return; }  // k_visual_euclid's block tile (vis_max_key slots)
  if ((visual_kind != SA_VIS_COSINE && visual_kind != SA_VIS_EUCLIDEAN) || !maxN || !maxTK) return;
  switch (tile_plan(maxN, maxTK, ns, Dp, plan_override)) {
    case 0: case 8: case 15: *bm = 128; *bn = 128; break;
    case 5: case 16: case 18: *bm = 64; *bn = 128; break;
    case 6: *bm = 128; *bn = 64; break;
    default: break;
  }
}

// The fused first phase (k_frame_visual) applies when the contraction runs as 64 x 64 tiles — frames of up to two tiles per compute
// unit, where the other two kinds of work are a sizeable part of the frame and a dependent launch (~4 us) is a sizeable part of
// either — and the feature length needs no padding; any N, T: with more than 1024 detections or tracks the positional tiles of the
// launch feed the many-workgroup tail (UNION) instead of the one-workgroup one.  Returns hipErrorNotSupported when it does not
// apply: the caller falls back to k_frame + k_visual_cost.
// (and with it, for banks of 2 .. SA_CLS_MAXK observations, the whole-track tiles and their class words)
bool sa_frame_visual_ok(uint32_t ns, uint32_t maxN, uint32_t maxT, uint32_t K, uint32_t D, const SaParams& p, bool class_words) {
  const uint32_t maxTK = maxT * K;
  const bool eu = p.visual_kind == SA_VIS_EUCLIDEAN && p.eu_mfma;
  if ((p.visual_kind != SA_VIS_COSINE && !eu) || !maxN || !maxTK || D != p.Dp) return false;
  const int plan = tile_plan(maxN, maxTK, ns, p.Dp, p.gemm_plan);
  // every plan of the 64 x 64 family: the launch runs one-k-group 64 x 64 tiles whatever the stand-alone kernel would do (frames of
  // several rounds of tiles — deeper banks: 1000 x 5000 columns at five observations per track — gain as well: 107.2 -> 102.5 us)
  if (plan == 1 || plan == 2 || plan == 4 || plan == 7 || plan == 9) return true;
  // deeper banks with class words: the whole-track tiles (64 x 64) replace THREE launches of the other family's path (positional
  // tiles, the contraction on wider tiles, k_bestfit_tile) — C2's frame with two observations per track: 42.9 us there.  (Only
  // with class words: the matrix mode's per-tile slots are laid out by the engine for the plan's own tile grid.)
  if (class_words && K >= 2 && K <= SA_CLS_MAXK) return true;
  // a frame of at most two 64 x 64 tiles per compute unit that the plan would give wider tiles (1000 detections x 1500 tracks: 64 x 128):
  // one launch less and the positional tiles beside the contraction weigh more than the wider tile's shorter main loop.  Vote-word
  // frames only (p.vote_words: the matrix mode's per-tile slots follow the plan's own grid).
  return p.gemm_plan < 0 && p.vote_words && K == 1 && (size_t)cdiv(maxN, 64) * cdiv(maxTK, 64) * ns <= 512;
}
hipError_t sa_launch_frame_visual(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxT, uint32_t K, uint32_t D,
                                  const SaParams& p_in, hipStream_t st, bool partials, int prep, bool kpass, bool general_tail) {
  // The matrix waves of a one-observation cosine frame sleep 64 cycles after every k-step (SA_FLAG_NO_YIELD: never): while a wave presents
  // matrix instructions back to back, the positional tiles' waves on its SIMD issue no vector instruction at all (NOTES), and with the
  // k-split loop the contraction's tile retires ~3-6 k cycles BEFORE the positional tiles that end the launch — 32 naps hand them ~2 k
  // cycles of the port at the same cost to the tile that can afford it: C2 first phase -1.5 .. -5 % over three visits (c2n -1.5 %); euclidean
  // frames (a longer epilogue: +2.5 %) and deeper banks (several matrix waves per SIMD: no difference) do not take it.
  SaParams p_ = p_in;
  p_.ks_yield = (p_in.no_yield || kpass || (p_in.visual_kind == SA_VIS_EUCLIDEAN && p_in.eu_mfma)) ? 0u : 1u;

  const SaParams& p = p_;
  const bool eu = p.visual_kind == SA_VIS_EUCLIDEAN && p.eu_mfma;
  if (!sa_frame_visual_ok(ns, maxN, maxT, K, D, p, kpass)) return hipErrorNotSupported;
  const uint32_t maxTK = maxT * K;
  const uint32_t gy = cdiv(maxN, 64), py = cdiv(maxN, POS_TI);
  // Tiles of 64 x 96 where they take fewer rounds of the chip's 256 CUs than 64 x 64 ones cost (one and a half times the work each): the
  // frames between one and one and a half rounds of 64 x 64 tiles — c2t, 1000 x 1500: 384 tiles, half the CUs with two; 256 of 64 x 96.
  // Cosine frames that vote through the vote words (visual_tile96); sa_config.gemm_plan = 19 + 1 pins the form (tests).
  bool w96 = false;
  if (!kpass && !eu && partials && p.vote_words && K == 1 && !p.staged_loop) {
    const size_t t64 = (size_t)cdiv(maxTK, 64) * gy * ns, t96 = (size_t)cdiv(maxTK, 96) * gy * ns;
    w96 = p.gemm_plan == 19 || (p.gemm_plan < 0 && ((t96 + 255) / 256) * 3 < ((t64 + 255) / 256) * 2);
  }
  // (their matrix waves nap 128 cycles per k-step of 12 matrix instructions — such a frame brings three positional tiles per CU, which
  // end the launch: c2t first phase 24.4-25.0 us without naps, 23.9-24.3 with 64 cycles, 22.7-23.1 with 128, 23.0-23.2 with 192, 24.6 with
  // 256.  As compiled: with the nap's length chosen by a two-way branch instead of the loop's four-way one the 128-cycle form read 24.0 —
  // the scalar instructions between two k-steps are part of the gap the positional waves get)
  if (w96 && p_.ks_yield) p_.ks_yield = 1u | (2u << 8);
  const uint32_t gx = kpass ? cdiv(maxT, 64u / K) : w96 ? cdiv(maxTK, 96) : cdiv(maxTK, 64);
  uint32_t px = cdiv(maxT, 128);
  // preparation blocks: 1 = all of them (one wave per feature row: N / 4), 3 = the reset half only (one thread per row / column),
  // 0 = none (a lean frame on the one-workgroup tail: nothing on its path reads what they write, enqueue_frame)
  uint32_t prep_blocks = cdiv(maxN + maxT + 1, 256);
  if (prep == 1 && cdiv(maxN, 4) > prep_blocks) prep_blocks = cdiv(maxN, 4);
  if (prep == 0) prep_blocks = 0;
  // (the fused launch numbers its contraction tiles in XCD-aware order by default: the same speed on every workload that takes it —
  // C2 20.3, c2t 36.1 / 36.4, c2k3 46.7 / 46.6, c2d 80.9, c2e 23.4 us either way — for a fifth fewer L2 fills; SA_FLAG_ROW_TILES: row by row)
  const XcdOrder xo = xcd_order(gx, gy, p.row_major_tiles == 2u);
  const uint32_t xo_ = (xo.chunk << 8) | xo.W, n_gemm = xo.W ? 8u * xo.chunk : xo.chunk;
  // Positional tiles of 16 x 256 where the launch's blocks would not all be resident with 16 x 128 (four blocks per CU: 1024) — deeper
  // banks, request sets of several scenes: a tile that has to wait for a place starts when the first contraction tiles retire.  Measured,
  // first phase: c2k3 37.1 -> 35.5 us, c2bk3 262 -> 253, c2d 69.3 -> 68.5.  Not where the tiles also feed the many-workgroup tail (row
  // duals, union-find: a 16 x 256 tile of THAT kind is a longer chain than two rounds of 16 x 128 — c2t 27.0 -> 29.2 us).
  const bool wide_pos = !general_tail && (size_t)(n_gemm + px * py + prep_blocks) * ns > 1024u;
  if (wide_pos) px = cdiv(maxT, 256);
  sa_trace_hook(st, n_gemm + px * py + prep_blocks);
  // One k-group (256-thread blocks, 32 KB of LDS for every kind of block: five blocks per CU).  With two k-groups the
  // contraction alone is faster (15 vs 20 us) but every block of the launch then owns 512 threads and 64 KB — a kernel's LDS is
  // per launch, not per block — so only two blocks fit a CU, and even with two positional / preparation units side by side in
  // each 512-thread block the latency-bound tiles, which want four or five blocks in flight per CU, queue: 30 us for the launch
  // against 22.7 (raising the contraction's wave priority changes nothing).
  const dim3 grid(n_gemm + px * py + prep_blocks, 1, ns);
  const uint32_t np = prep_blocks | (prep == 3 ? 0x80000000u : 0u) | (general_tail ? 0x40000000u : 0u) | (wide_pos ? 0x20000000u : 0u);
  // the contraction tiles' main loop: k-split over the bank's fragment-order twin by default, the LDS-staged loop with SA_FLAG_STAGED_LOOP
#define SA_FV(PART_, EU_, KP_) do { if (p.staged_loop) SA_LAUNCH((k_frame_visual<1, PART_, EU_, KP_, false>), grid, dim3(256), 0, st, scenes, p, gx, gy, px, py, np, xo_); \
                                    else SA_LAUNCH((k_frame_visual<1, PART_, EU_, KP_, true>), grid, dim3(256), 0, st, scenes, p, gx, gy, px, py, np, xo_); } while (0)
  if (kpass) {
    if (eu) SA_FV(false, true, true);
    else SA_FV(false, false, true);
    return hipGetLastError();
  }
  if (eu) {
    if (partials) SA_FV(true, true, false);
    else SA_FV(false, true, false);
  } else if (w96) SA_LAUNCH((k_frame_visual<1, true, false, false, true, true>), grid, dim3(256), 0, st, scenes, p, gx, gy, px, py, np, xo_);
  else if (partials) SA_FV(true, false, false);
  else SA_FV(false, false, false);
#undef SA_FV
  return hipGetLastError();
}

// The stand-alone contraction.  Plans (sa_config.gemm_plan - 1 pins one): 0 / 5 / 6 = 128x128 / 64x128 / 128x64 on the LDS-staged loop, 1 / 2 / 4 =
// 64x64 with that many k-groups, 7 / 8 = ring variants; 9 = 64x64 on the k-split loop, 15 / 16 = 128x128 / 64x128 on the direct loop (both read
// the bank's fragment-order twin: gemm_mainloop_ks / gemm_mainloop_direct), 18 = 64x128 on ITS k-split loop (gemm_mainloop_ks128).  tile_plan()
// chooses among the tile SIZES; unless a plan is pinned or SA_FLAG_STAGED_LOOP is set, 128x128 / 64x128 / 64x64 then run the direct / k-split /
// k-split loops (measured on the stand-alone contraction, 4096 x 2048 x 512: 93.6 -> 74.6 us; 1000 x 1000 x 512: 15.0 -> 12.9; C5's frame with the
// 64x128 tile staged / direct / k-split: 657 / 632-642 / 617-619 us; 128x64 stays staged: two row-major gathers per fragment-order load are what
// the direct loop is worst at).
static inline int loop_plan(int plan, const SaParams& p) {
  if (p.gemm_plan >= 0 || p.staged_loop) return plan;
  return plan == 0 ? 15 : plan == 5 ? 18 : (plan == 1 || plan == 2) ? 9 : plan;
}
hipError_t sa_launch_visual(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxTK, const SaParams& p,
                            hipStream_t st, bool partials) {
  if (!maxN || !maxTK) return hipSuccess;
  sa_trace_hook(st, cdiv(maxTK, 64) * cdiv(maxN, 64));
  if (p.visual_kind == SA_VIS_EUCLIDEAN && p.eu_mfma) {
    // euclidean distances through the contraction: the one-k-group plans of every tile size (the k-group and ring plans are cosine tuning)
    int plan = tile_plan(maxN, maxTK, ns, p.Dp, p.gemm_plan);
    plan = (plan == 0 || plan == 8) ? 0 : (plan == 5 || plan == 6 || plan == 9 || plan == 15 || plan == 16 || plan == 18) ? plan : 1;
    plan = loop_plan(plan, p);
#define SA_EU_LAUNCH(BM_, BN_, KGT_) do { if (partials) launch_cosine<BM_, BN_, KGT_, true, true>(maxTK, maxN, ns, st, scenes, p); \
                                          else launch_cosine<BM_, BN_, KGT_, false, true>(maxTK, maxN, ns, st, scenes, p); } while (0)
    switch (plan) {
      case 0: SA_EU_LAUNCH(128, 128, 1); break;
      case 5: SA_EU_LAUNCH(64, 128, 1); break;
      case 6: SA_EU_LAUNCH(128, 64, 1); break;
      case 9: SA_EU_LAUNCH(64, 64, 9); break;
      case 15: SA_EU_LAUNCH(128, 128, 15); break;
      case 16: SA_EU_LAUNCH(64, 128, 15); break;
      case 18: SA_EU_LAUNCH(64, 128, 17); break;
      default: SA_EU_LAUNCH(64, 64, 1); break;
    }
#undef SA_EU_LAUNCH
    return hipGetLastError();
  }
  if (p.visual_kind == SA_VIS_COSINE) {
    const uint32_t Dp = p.Dp;  // one feature length per engine
    int plan = loop_plan(tile_plan(maxN, maxTK, ns, Dp, p.gemm_plan), p);
    if (partials) {
      plan = plan == 4 ? 2 : plan == 7 ? 1 : plan == 8 ? 0 : plan;
      switch (plan) {
        case 0: launch_cosine<128, 128, 1, true, false>(maxTK, maxN, ns, st, scenes, p); break;
        case 5: launch_cosine<64, 128, 1, true, false>(maxTK, maxN, ns, st, scenes, p); break;
        case 6: launch_cosine<128, 64, 1, true, false>(maxTK, maxN, ns, st, scenes, p); break;
        case 2: launch_cosine<64, 64, 2, true, false>(maxTK, maxN, ns, st, scenes, p); break;
        case 9: launch_cosine<64, 64, 9, true, false>(maxTK, maxN, ns, st, scenes, p); break;
        case 15: launch_cosine<128, 128, 15, true, false>(maxTK, maxN, ns, st, scenes, p); break;
        case 16: launch_cosine<64, 128, 15, true, false>(maxTK, maxN, ns, st, scenes, p); break;
        case 18: launch_cosine<64, 128, 17, true, false>(maxTK, maxN, ns, st, scenes, p); break;
        default: launch_cosine<64, 64, 1, true, false>(maxTK, maxN, ns, st, scenes, p); break;
      }
      return hipGetLastError();
    }
    switch (plan) {
      case 0: launch_cosine<128, 128, 1, false, false>(maxTK, maxN, ns, st, scenes, p); break;
      case 5: launch_cosine<64, 128, 1, false, false>(maxTK, maxN, ns, st, scenes, p); break;
      case 7: launch_cosine<64, 64, 0, false, false>(maxTK, maxN, ns, st, scenes, p); break;
      case 8: launch_cosine<128, 128, 0, false, false>(maxTK, maxN, ns, st, scenes, p); break;
      case 6: launch_cosine<128, 64, 1, false, false>(maxTK, maxN, ns, st, scenes, p); break;
      case 4: launch_cosine<64, 64, 4, false, false>(maxTK, maxN, ns, st, scenes, p); break;
      case 2: launch_cosine<64, 64, 2, false, false>(maxTK, maxN, ns, st, scenes, p); break;
      case 9: launch_cosine<64, 64, 9, false, false>(maxTK, maxN, ns, st, scenes, p); break;
      case 15: launch_cosine<128, 128, 15, false, false>(maxTK, maxN, ns, st, scenes, p); break;
      case 16: launch_cosine<64, 128, 15, false, false>(maxTK, maxN, ns, st, scenes, p); break;
      case 18: launch_cosine<64, 128, 17, false, false>(maxTK, maxN, ns, st, scenes, p); break;
      default: launch_cosine<64, 64, 1, false, false>(maxTK, maxN, ns, st, scenes, p); break;
    }
  } else {
    SA_LAUNCH(k_visual_euclid, dim3(cdiv(maxTK, EU_BN), cdiv(maxN, EU_BM), ns), dim3(EU_THREADS), 0, st, scenes, p);
  }
  return hipGetLastError();
}

hipError_t sa_launch_distance_matrix(int kind, const float* a, const float* an, const float* b, const float* bn,
                                     uint32_t n, uint32_t t, uint32_t dp, float* out, hipStream_t st, int32_t plan_override) {
  if (!n || !t) return hipSuccess;
  sa_trace_hook(st, cdiv(t, 64) * cdiv(n, 64));
  if (kind == SA_VIS_COSINE) {
    switch (tile_plan(n, t, 1, dp, plan_override)) {
      case 0: hipLaunchKernelGGL((k_cosine_matrix<128, 128, 1>), dim3(cdiv(t, 128), cdiv(n, 128)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 5: hipLaunchKernelGGL((k_cosine_matrix<64, 128, 1>), dim3(cdiv(t, 128), cdiv(n, 64)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 7: hipLaunchKernelGGL((k_cosine_matrix<64, 64, 0>), dim3(cdiv(t, 64), cdiv(n, 64)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 8: hipLaunchKernelGGL((k_cosine_matrix<128, 128, 0>), dim3(cdiv(t, 128), cdiv(n, 128)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 6: hipLaunchKernelGGL((k_cosine_matrix<128, 64, 1>), dim3(cdiv(t, 64), cdiv(n, 128)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 4: hipLaunchKernelGGL((k_cosine_matrix<64, 64, 4>), dim3(cdiv(t, 64), cdiv(n, 64)), dim3(1024), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 2: hipLaunchKernelGGL((k_cosine_matrix<64, 64, 2>), dim3(cdiv(t, 64), cdiv(n, 64)), dim3(512), 0, st, a, an, b, bn, n, t, dp, out); break;
      // k-split plans (b = the bank in fragment order for 10 / 13: sa_launch_frag_reorder; 13: a as well)
      case 9: hipLaunchKernelGGL((k_cosine_matrix<64, 64, 9>), dim3(cdiv(t, 64), cdiv(n, 64)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 10: hipLaunchKernelGGL((k_cosine_matrix<64, 64, 10>), dim3(cdiv(t, 64), cdiv(n, 64)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 15: hipLaunchKernelGGL((k_cosine_matrix<128, 128, 15>), dim3(cdiv(t, 128), cdiv(n, 128)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 16: hipLaunchKernelGGL((k_cosine_matrix<64, 128, 15>), dim3(cdiv(t, 128), cdiv(n, 64)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 17: hipLaunchKernelGGL((k_cosine_matrix<128, 64, 15>), dim3(cdiv(t, 64), cdiv(n, 128)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 18: hipLaunchKernelGGL((k_cosine_matrix<64, 128, 17>), dim3(cdiv(t, 128), cdiv(n, 64)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      case 13: hipLaunchKernelGGL((k_cosine_matrix<64, 64, 13>), dim3(cdiv(t, 64), cdiv(n, 64)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
      default: hipLaunchKernelGGL((k_cosine_matrix<64, 64, 1>), dim3(cdiv(t, 64), cdiv(n, 64)), dim3(256), 0, st, a, an, b, bn, n, t, dp, out); break;
    }
  } else {
    hipLaunchKernelGGL(k_euclid_matrix, dim3(cdiv(t, EU_BN), cdiv(n, EU_BM)), dim3(EU_THREADS), 0, st, a, b, n, t, dp, out);
  }
  return hipGetLastError();
}
