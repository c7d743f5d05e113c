// sa_kernels.hip — gfx950 kernels of the association path other than the feature contraction:
// frame preparation, pair pre-filter + IoU / Mahalanobis cost cells, BestFit vote, and the positional
// assignment (edge compaction, connected components, exact per-component solve).
//
// These are HBM/latency-bound integer / f64 / f32 elementwise kernels: the design rules are coalesced
// 256-B row segments per wave, LDS staging of the per-tile operands, wave ballots for compaction, no
// global atomics on the hot cells, few launches (each dependent launch costs ~2-4 us on this chip), and
// grid.z = scene so a whole batch of scenes goes through one launch.  No MFMA here on purpose.
#include "sa_engine.h"
#include "sa_frame.h"
#include "sa_dense.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

#define WAVE 64

thread_local hipEvent_t sa_prof_start = nullptr, sa_prof_stop = nullptr;
thread_local hipEvent_t sa_done_event = nullptr;

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// =====================================================================================================
// Preparation
// =====================================================================================================
__global__ void k_pad_features(const float* __restrict__ src, uint32_t rows, uint32_t D, uint32_t Dp, uint32_t K,
                               const uint32_t* __restrict__ slots, const uint8_t* __restrict__ present,
                               float* __restrict__ dst, float* __restrict__ norms, uint8_t* __restrict__ dst_present,
                               float* __restrict__ dst_frag) {
  uint32_t row = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  uint32_t lane = threadIdx.x % WAVE;
  if (row >= rows) return;
  uint32_t drow = slots ? slots[row / K] * K + row % K : row;
  bool pres = present ? present[row] != 0 : true;
  float nrm;
  pad_feature_row(src + (size_t)row * D, dst + (size_t)drow * Dp, D, Dp, pres && src, lane, &nrm, dst_frag, drow);
  if (lane == 0) {
    norms[drow] = nrm;
    if (dst_present) dst_present[drow] = pres ? 1 : 0;
  }
}
// visual_features_collected_count = observations that carry a feature (visual_sort/metric.rs:368-371)
__global__ void k_feat_count(const uint32_t* __restrict__ slots, uint32_t n, uint32_t K,
                             const uint8_t* __restrict__ fpresent, uint32_t* __restrict__ fcount) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s = slots[i];
  uint32_t c = 0;
  for (uint32_t k = 0; k < K; ++k) c += fpresent[(size_t)s * K + k] ? 1u : 0u;
  fcount[s] = c;
}

// Stored tracks touched by an upsert: scatter to their table rows; Kalman projection + Cholesky once per
// track (kalman_2d_box.rs:104-120,167) instead of once per pair.
__global__ void k_prep_tracks(PrepTrackArgs a, SaParams p) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  uint32_t s = a.slots[i];
  BoxRaw r = a.raw[i];
  prep_box_common(r, &a.geo[s], &a.verts[(size_t)s * 8]);
  a.ext[s] = sa_box_ext(r.box.aspect, r.box.height, r.box.has_angle && r.box.angle != 0.0f);
  a.t_epoch[s] = a.epochs[i];
  a.t_ids[s] = a.ids[i];
  if (a.kf_mean && a.kf_cov) sa_maha_prepare(p.kf_position_weight, a.kf_mean + (size_t)i * 5, a.kf_cov + (size_t)i * 25, a.maha + (size_t)s * 20);
}

// Ingest of one request set: the shader engines pull the pinned host blocks (the staging arena, and the feature blocks callers keep
// pinned themselves) over PCIe through their device mapping and store them to HBM.  One launch moves every segment; a few dozen
// workgroups keep enough 16-byte reads in flight to fill the link (scripts/micro/h2d_rate.hip: 2 MB in 40 us = 52 GB/s with 32-64
// workgroups, against 57 us for one hipMemcpyAsync on one SDMA engine), and it shares a stream with nothing but other ingests, so
// it runs beside the previous set's kernels.  Segments are 16-byte aligned on both sides (the launcher checks).
__global__ __launch_bounds__(256) void k_ingest(SaCopySegs segs) {
  for (uint32_t k = 0; k < segs.n; ++k) {
    const uint4* __restrict__ src = (const uint4*)segs.s[k].src;
    uint4* __restrict__ dst = (uint4*)segs.s[k].dst;
    const size_t n16 = segs.s[k].bytes >> 4;
    // (four reads in flight per lane and fewer workgroups: measured slower beside the contraction, 59-61 us against 54-55)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    const uint32_t tail = (uint32_t)(segs.s[k].bytes & 15u);
    if (blockIdx.x == 0 && threadIdx.x < tail) ((uint8_t*)dst)[(n16 << 4) + threadIdx.x] = ((const uint8_t*)src)[(n16 << 4) + threadIdx.x];
  }
}
// `done` (optional): signalled by the dispatch's own completion — no marker packet of its own behind the kernel (a separate
// hipEventRecord costs the copy stream ~4 us per request set: the next set's ingest is queued right behind)
hipError_t sa_launch_ingest(const SaCopySegs& segs, uint32_t blocks, hipStream_t st, hipEvent_t done) {
  if (!segs.n) return hipSuccess;
  if (done) hipExtLaunchKernelGGL(k_ingest, dim3(blocks), dim3(256), 0, st, nullptr, done, 0, segs);
  else hipLaunchKernelGGL(k_ingest, dim3(blocks), dim3(256), 0, st, segs);
  return hipGetLastError();
}

__global__ void k_gather_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                              const uint32_t* __restrict__ index, uint32_t rows, uint32_t row_bytes) {
  uint32_t row = blockIdx.x;
  if (row >= rows) return;
  const uint8_t* s = src + (size_t)index[row] * row_bytes;
  uint8_t* d = dst + (size_t)row * row_bytes;
  if ((row_bytes & 3u) == 0) {
    for (uint32_t k = threadIdx.x; k < row_bytes / 4; k += blockDim.x) ((uint32_t*)d)[k] = ((const uint32_t*)s)[k];
  } else {
    for (uint32_t k = threadIdx.x; k < row_bytes; k += blockDim.x) d[k] = s[k];
  }
}

// sa_tracks_remove: row r of every array of a scene's track table := its old row index[r], all arrays in ONE launch (one block per kept row)
__device__ __forceinline__ void gather_table_row(const SaGatherTable& g, uint32_t row) {
  const uint32_t from = g.index[row];
  for (uint32_t a = 0; a < g.n_arrays; ++a) {
    const uint32_t rb = g.row_bytes[a];
    const uint8_t* s = (const uint8_t*)g.src[a] + (size_t)from * rb;
    uint8_t* d = (uint8_t*)g.dst[a] + (size_t)row * rb;
    if ((rb & 15u) == 0) {
      for (uint32_t k = threadIdx.x; k < rb / 16; k += 256) ((uint4*)d)[k] = ((const uint4*)s)[k];
    } else if ((rb & 3u) == 0) {
      for (uint32_t k = threadIdx.x; k < rb / 4; k += 256) ((uint32_t*)d)[k] = ((const uint32_t*)s)[k];
    } else {
      for (uint32_t k = threadIdx.x; k < rb; k += 256) d[k] = s[k];
    }
    if (g.frag && a == g.frag_array) {
      // the bank's fragment-order twin: the row's K observations, 16 bytes (four consecutive k) per store, from the row-major SOURCE
      const uint32_t q = g.frag_Dp >> 2;  // 16-byte pieces per observation
      for (uint32_t k = threadIdx.x; k < g.frag_K * q; k += 256) {
        const uint32_t ob = k / q, kk = (k % q) * 4u;
        *(uint4*)(g.frag + sa_frag_index(row * g.frag_K + ob, kk, g.frag_Dp)) = ((const uint4*)s)[k];
      }
    }
  }
}
__global__ __launch_bounds__(256) void k_gather_table(SaGatherTable g) {
  if (blockIdx.x >= g.rows) return;
  gather_table_row(g, blockIdx.x);
}
__global__ __launch_bounds__(256) void k_gather_tables(SaGatherTables set) {
  const SaGatherTable& g = set.t[blockIdx.y];
  if (blockIdx.x >= g.rows) return;
  gather_table_row(g, blockIdx.x);
}
hipError_t sa_launch_gather_tables(const SaGatherTables& set, hipStream_t st, hipEvent_t done) {
  if (!set.n) return hipSuccess;
  if (set.n == 1) return sa_launch_gather_table(set.t[0], st, done);
  uint32_t rows = 0;
  for (uint32_t i = 0; i < set.n; ++i) rows = set.t[i].rows > rows ? set.t[i].rows : rows;
  if (!rows) return hipSuccess;
  if (done) hipExtLaunchKernelGGL(k_gather_tables, dim3(rows, set.n), dim3(256), 0, st, nullptr, done, 0, set);
  else hipLaunchKernelGGL(k_gather_tables, dim3(rows, set.n), dim3(256), 0, st, set);
  return hipGetLastError();
}
hipError_t sa_launch_gather_table(const SaGatherTable& g, hipStream_t st, hipEvent_t done) {
  if (!g.rows || !g.n_arrays) return hipSuccess;
  if (done) hipExtLaunchKernelGGL(k_gather_table, dim3(g.rows), dim3(256), 0, st, nullptr, done, 0, g);   // (carries its own completion signal)
  else hipLaunchKernelGGL(k_gather_table, dim3(g.rows), dim3(256), 0, st, g);
  return hipGetLastError();
}

// Once per (re)allocation of a slot's assignment state: what the tail kernels afterwards leave behind every frame.
__global__ void k_slot_init(uint32_t* e_cnt, int64_t* u, uint32_t n_rows, uint32_t* parent, uint32_t n_vertices) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_rows) { e_cnt[i] = 0; u[i] = 0; }
  if (i < n_vertices) parent[i] = i;
}

#ifdef SA_POS_TRACE
// per-tile timeline of the positional launch (positional_tile, sa_frame.h): 8 words per block, dumped by sa_pos_trace_hook
__device__ uint64_t* g_pos_trace;
static void sa_pos_trace_hook(hipStream_t st, uint32_t nb) {
  static uint64_t* buf = nullptr;
  static int calls = 0;
  const char* env = getenv("SA_POS_TRACE");
  if (!env) return;
  if (!buf) {
    hipMalloc(&buf, 64 * 262144);
    hipMemset(buf, 0, 64 * 262144);
    hipMemcpyToSymbol(HIP_SYMBOL(g_pos_trace), &buf, sizeof(void*));
  }
  if (++calls != atoi(env)) return;   // dumps what the PREVIOUS launches left
  hipStreamSynchronize(st);
  if (nb > 262144) nb = 262144;
  uint64_t* h = (uint64_t*)malloc((size_t)nb * 64);
  hipMemcpy(h, buf, (size_t)nb * 64, hipMemcpyDeviceToHost);
  if (FILE* f = fopen("gpurun_out/pos_trace.txt", "w")) {
    for (uint32_t i = 0; i < nb; ++i)
      if (h[i * 8])
        fprintf(f, "%u %llu %llu %llu %llu %llu %llu %llu %llu\n", i, (unsigned long long)h[i * 8], (unsigned long long)h[i * 8 + 1],
                (unsigned long long)h[i * 8 + 2], (unsigned long long)h[i * 8 + 3], (unsigned long long)h[i * 8 + 4],
                (unsigned long long)h[i * 8 + 5], (unsigned long long)h[i * 8 + 6], (unsigned long long)h[i * 8 + 7]);
    fclose(f);
  }
  free(h);
}
#define SA_POS_TRACE_PTR() (g_pos_trace && (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) < 262144u ? g_pos_trace + 8 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) : nullptr)
#else
static inline void sa_pos_trace_hook(hipStream_t, uint32_t) {}
#define SA_POS_TRACE_PTR() nullptr
#endif
// The first launch of a frame: blockIdx.y < pos_rows -> a positional tile; the rows above carry the frame-preparation blocks.
// PW: groups of clipping lanes whose vertex lists the tile keeps in LDS (the bulk of it: 24 KB at 64).  32 of them make the tile 23.6 KB —
// six blocks per CU instead of four — at the price of a second clip round for tiles with more than 32 surviving pairs: taken by launches
// of more than 1024 blocks (sa_launch_frame), where a tile that waits for a place costs more than a round (c2b: 2016 tiles, launch
// 25.8 -> 22.9 us); C3's 512 tiles, all resident either way, would pay 7.7 -> 9.7.
template <int NSUB, bool UNION, int PW = 64>
__global__ __launch_bounds__(256) void k_frame(const SceneDev* __restrict__ scenes, SaParams p, uint32_t pos_rows_) {
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(PosSmem<NSUB, PW>)];
  const uint32_t pos_rows = pos_rows_ & 0x7fffffffu;  // (bit 31: the preparation blocks run their reset half only)
  if (blockIdx.y < pos_rows) positional_tile<false, true, NSUB, UNION, true, PW, true>(S, p, blockIdx.x, blockIdx.y, smem, threadIdx.x, SA_POS_TRACE_PTR());
  else frame_prep_block(S, p, (blockIdx.y - pos_rows) * gridDim.x + blockIdx.x, threadIdx.x, (pos_rows_ >> 31) != 0);
}
// Parity taps: the dense f32 cost matrix, no side effects.
__global__ __launch_bounds__(256) void k_positional_dense(const SceneDev* __restrict__ scenes, SaParams p) {
  const SceneDev S = scenes[blockIdx.z];
  __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(PosSmem<1>)];
  positional_tile<true, false, 1, false>(S, p, blockIdx.x, blockIdx.y, smem, threadIdx.x);
}

// Debug tap: (w * 1e6f) as i64 of every positional cell, 0 where absent (sort/voting.rs:59).
__global__ void k_quant_tap(const SceneDev* __restrict__ scenes) {
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  size_t n = (size_t)S.N * S.T;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
    float w = S.pos[c];
    S.quant[c] = sa_quantise(w == w ? w : 0.0f);
  }
}

// =====================================================================================================
// BestFit vote (track/voting/best.rs:52-128) without the sort (these two kernels: without atomics; frames of one observation per
// track and at most 1024 x 1024 do not come here at all — vote words, k_assign_small<.., WORDS>).  Candidate q wins track
// t*(q) — its heaviest group — iff (q, t*) is the first group of column t* in (weight desc, q asc, t asc)
// order; that is the order the reference's stable sort gives a canonically ordered list.
//   k_bestfit_tile   : 64 x 64 cells per block, lane = column, wave = 16 rows.  W[q,t] = sum_k f64(max_dist -
//                      w_k) over present k (needs count >= min_votes).  Emits, per (row, column tile), the row's
//                      best (W, t) and, per (row tile, column), the column's best (W, lowest q).
//   k_bestfit_resolve: one thread per candidate folds its CT row partials, then the RT column partials of
//                      the winning column, and decides; winners mark excluded_tracks (visual_sort/voting.rs:62-71).
// =====================================================================================================
// Deeper banks on frames of at most 1024 x 1024: the tile's row / column winners go straight into the vote words as well (no
// resolve launch).  A group weight W = sum_k f64(max_dist - w_k) >= 0 does not fit beside an index in 32 bits, so the word is
// (2^54 - 1 - key54(W)) << 10 | index with key54 = the leading 54 bits of the f64 pattern + 1: the 64-bit minimum is the heaviest
// group, lowest index among groups whose weights agree to 2^-43 relative — the reference breaks EXACT ties by index; weights that
// close differ by 1e-13 of themselves, eight orders of magnitude below what the f32 distances carry.  Never all ones (= no group).
__device__ __forceinline__ unsigned long long sa_vote_word10(double W, uint32_t index) {
  const unsigned long long key = ((unsigned long long)__double_as_longlong(W) >> 9) + 1ull;
  return ((((1ull << 54) - 1ull) - key) << 10) | (unsigned long long)(index & 1023u);
}
__global__ __launch_bounds__(256) void k_bestfit_tile(const SceneDev* __restrict__ scenes, SaParams p) {
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  const uint32_t N = S.N, T = S.T, K = S.K;
  const uint32_t ct = blockIdx.x, rt = blockIdx.y;
  if (ct >= S.CT || rt >= S.RT) return;
  __shared__ double s_w[4][WAVE];
  __shared__ uint32_t s_q[4][WAVE];
  const uint32_t wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  const uint32_t t = ct * 64 + lane;
  const uint32_t q0 = rt * 64 + wave * 16;
  // BestFit max_dist = the largest present weight of the scene-frame, -1.0 when there is none (voting/best.rs:59): fold the
  // per-workgroup slots the cost kernel left (every one of them is rewritten each frame; key 0 = none) — all 256 threads
  // share the loads, the four wave maxima meet in LDS
  __shared__ uint32_t s_mk[4];
  uint32_t mk = 0;
  for (uint32_t i = threadIdx.x; i < S.nkeys; i += 256) {
    const uint32_t v = S.vis_max_key[i];
    mk = v > mk ? v : mk;
  }
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t ok = __shfl_xor(mk, o);
    mk = ok > mk ? ok : mk;
  }
  if (lane == 0) s_mk[wave] = mk;
  __syncthreads();
  mk = s_mk[0] > s_mk[1] ? s_mk[0] : s_mk[1];
  mk = s_mk[2] > mk ? s_mk[2] : mk;
  mk = s_mk[3] > mk ? s_mk[3] : mk;
  const float max_dist = mk ? sa_key_f32(mk) : -1.0f;
  // phase 1: the 16 group weights of this lane's column — all loads issued before any reduction
  double Wr[16];
  if (K == 1) {
    float x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t q = q0 + r;
      x[r] = (q < N && t < T) ? S.vis[(size_t)q * T + t] : __builtin_nanf("");
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double w = 0.0 + (double)(max_dist - x[r]);  // the reference's sum starts from 0.0
      Wr[r] = (x[r] == x[r] && 1u >= p.min_votes) ? w : -1.0;
    }
  } else if (K <= 4) {
    // shallow banks (the reference's benches use 3): all 16 x K loads of the lane in flight before the first sum
    float x[16][4];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t q = q0 + r;
      const bool in = q < N && t < T;
      const float SA_G* v = S.vis + ((size_t)(in ? q : 0) * T + (in ? t : 0)) * K;
#pragma unroll
      for (int k = 0; k < 4; ++k) x[r][k] = (in && (uint32_t)k < K) ? v[k] : __builtin_nanf("");
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      uint32_t cnt = 0;
      double w = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (x[r][k] == x[r][k]) { ++cnt; w += (double)(max_dist - x[r][k]); }  // k ascending: the reference's summation order
      Wr[r] = (cnt >= 1 && cnt >= p.min_votes) ? w : -1.0;
    }
  } else if (K <= 8) {
    // the reference's default depth is 5 (visual_sort/options.rs:194-205): the same, eight rows at a time (64 loads in flight per lane)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float x[8][8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t q = q0 + h * 8 + r;
        const bool in = q < N && t < T;
        const float SA_G* v = S.vis + ((size_t)(in ? q : 0) * T + (in ? t : 0)) * K;
#pragma unroll
        for (int k = 0; k < 8; ++k) x[r][k] = (in && (uint32_t)k < K) ? v[k] : __builtin_nanf("");
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        uint32_t cnt = 0;
        double w = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (x[r][k] == x[r][k]) { ++cnt; w += (double)(max_dist - x[r][k]); }  // k ascending: the reference's summation order
        Wr[h * 8 + r] = (cnt >= 1 && cnt >= p.min_votes) ? w : -1.0;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t q = q0 + r;
      double W = -1.0;
      if (q < N && t < T) {
        const float SA_G* v = S.vis + ((size_t)q * T + t) * K;
        uint32_t cnt = 0;
        double w = 0.0;
        for (uint32_t k = 0; k < K; ++k) {
          float xv = v[k];
          if (xv == xv) { ++cnt; w += (double)(max_dist - xv); }
        }
        if (cnt >= 1 && cnt >= p.min_votes) W = w;
      }
      Wr[r] = W;
    }
  }
  // phase 2a: column best over the wave's 16 rows (q ascends: strict > keeps the lowest q) — registers only;
  // the weights also go to LDS, transposed use below
  __shared__ double s_tile[64][65];  // [row][col], +1 padding
  double cw = -1.0;
  uint32_t cq = SA_NONE;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const double W = Wr[r];
    if (W > cw) { cw = W; cq = q0 + r; }
    s_tile[wave * 16 + r][lane] = W;
  }
  s_w[wave][lane] = cw;
  s_q[wave][lane] = cq;
  __syncthreads();
  // phase 2b: row argmax (W desc, t asc): 4 threads per row scan 16 columns each, then two shuffle steps
  {
    const uint32_t row = threadIdx.x >> 2, quarter = threadIdx.x & 3u;
    double bw = -1.0;
    uint32_t bt = SA_NONE;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t c = quarter * 16 + j;
      const double W = s_tile[row][c];
      if (W > bw) { bw = W; bt = ct * 64 + c; }  // columns ascend: strict > keeps the lowest t
    }
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
      double ow = __shfl_xor(bw, o);
      uint32_t ot = __shfl_xor(bt, o);
      if (ow > bw || (ow == bw && ot < bt)) { bw = ow; bt = ot; }
    }
    const uint32_t q = rt * 64 + row;
    if (quarter == 0 && q < N && p.vote_words) {
      if (bw >= 0.0) __hip_atomic_fetch_min(S.row_best + q, sa_vote_word10(bw, bt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (quarter == 0 && q < N) {
      S.row_part_w[(size_t)ct * S.N + q] = bw;
      S.row_part_t[(size_t)ct * S.N + q] = bw >= 0.0 ? (int32_t)bt : -1;
    }
  }
  if (wave == 0 && t < T) {
    double bw = s_w[0][lane];
    uint32_t bq = s_q[0][lane];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) {
      double ow = s_w[w2][lane];
      if (ow > bw) { bw = ow; bq = s_q[w2][lane]; }  // wave index ascends with q
    }
    if (p.vote_words) {
      if (bw >= 0.0) __hip_atomic_fetch_min(S.col_best + t, sa_vote_word10(bw, bq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      S.col_part_w[(size_t)rt * T + t] = bw;
      S.col_part_q[(size_t)rt * T + t] = bq;
    }
  }
}

// One wave per candidate: lanes fold the CT row partials, then the RT column partials of the winning column.
// RAW_W: the partials come from the contraction's own epilogue (visual_cosine_tile PART, bank depth 1) and hold the lightest
// WEIGHT of a row / column inside a tile.  The vote compares W = 0.0 + f64(max_dist - w): with one observation per group that
// is decreasing in w, so the heaviest group is the lightest weight and max_dist never has to be formed — the same reading of the
// rounding ties as inside the tiles (see visual_cosine_tile): weights that differ by less than an ulp of the difference.
template <bool RAW_W>
__global__ __launch_bounds__(256) void k_bestfit_resolve(const SceneDev* __restrict__ scenes) {
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  const uint32_t q = blockIdx.x * 4 + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (q >= S.N) return;
  // score: greater is better
  auto score = [&](double part) { return RAW_W ? -part : part; };
  const double none = RAW_W ? -__builtin_huge_val() : -1.0;
  double bw = none;
  uint32_t bt = SA_NONE;
  for (uint32_t ct = lane; ct < S.CT; ct += WAVE) {
    const double part = S.row_part_w[(size_t)ct * S.N + q];
    const int32_t tt = S.row_part_t[(size_t)ct * S.N + q];
    const double w = score(part);
    if (tt >= 0 && (bt == SA_NONE || w > bw)) { bw = w; bt = (uint32_t)tt; }  // a lane's tiles ascend with t
  }
  for (int o = 32; o > 0; o >>= 1) {
    double ow = __shfl_xor(bw, o);
    uint32_t ot = __shfl_xor(bt, o);
    if (ot != SA_NONE && (bt == SA_NONE || ow > bw || (ow == bw && ot < bt))) { bw = ow; bt = ot; }
  }
  if (bt == SA_NONE) return;  // no group at all: the candidate goes to the positional vote (wave-uniform)
  double cw = none;
  uint32_t cq = SA_NONE;
  for (uint32_t rt = lane; rt < S.RT; rt += WAVE) {
    const double part = S.col_part_w[(size_t)rt * S.T + bt];
    const uint32_t qq = S.col_part_q[(size_t)rt * S.T + bt];
    const double w = score(part);
    if (qq != SA_NONE && (cq == SA_NONE || w > cw)) { cw = w; cq = qq; }  // a lane's tiles ascend with q
  }
  for (int o = 32; o > 0; o >>= 1) {
    double ow = __shfl_xor(cw, o);
    uint32_t oq = __shfl_xor(cq, o);
    if (oq != SA_NONE && (cq == SA_NONE || ow > cw || (ow == cw && oq < cq))) { cw = ow; cq = oq; }
  }
  if (lane == 0) {
    S.row_has[q] = 1;  // feature_winners.contains_key(q)
    if (cq == q) {
      S.vis_winner[q] = (int32_t)bt;
      S.col_excluded[bt] = 1;
    }
  }
}

// =====================================================================================================
// Positional assignment = SortVoting::winners (sort/voting.rs:30-100) as an exact sparse solve.
// The edges (cells whose quantised weight beats the new-track threshold), the row duals and the union-find forest come
// straight out of the positional tiles, which run beside the visual vote; the visual vote's verdicts are applied here, lazily:
// rows that already hold a visual decision take no part (feature_winners.contains_key(from), visual_sort/voting.rs:77) and
// columns won visually are skipped while relaxing (excluded_tracks, :62-71).  A component may therefore be larger than
// strictly needed — harmless, it is still solved exactly.
//   N, T <= SA_SMALL_T (or N <= SA_SMALL_N, T <= 2 SA_SMALL_T; sa_small_tail_ok): k_assign_small / k_assign_small2 — ONE workgroup per scene: edges -> LDS, components, greedy start, group-cooperative
//            shortest augmenting paths for the rows the start left over (duals, matches and per-row minima in LDS), results,
//   else: k_assign_label (component root per row, rows pushed onto their root's list and counted), k_assign_solve (a component
//            of one or two rows: from the root thread's registers; up to 64 rows and 256 columns: one wavefront on a renumbered
//            matrix in LDS; larger: a whole workgroup with the dense solver of sa_dense.h — the last two handed out through
//            per-scene queues that every workgroup of the launch serves).
// =====================================================================================================
// In-kernel timeline of the one-workgroup tail (build with -DSA_TAIL_TRACE, run with SA_TAIL_TRACE=<launch #>): s_memtime of
// thread 0 at  0 entry | 1 counts scanned | 2 edges packed + components united | 3 labels | 4 sorted | 5 linked | 6 solved | 7 exit.
#ifdef SA_TAIL_TRACE
__device__ unsigned long long* g_tail_buf;
#define TAIL_STAMP(k) do { if (threadIdx.x == 0 && g_tail_buf) g_tail_buf[blockIdx.z * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
static void sa_tail_trace_hook(hipStream_t st, uint32_t ns) {
  static int calls = 0;
  static unsigned long long* buf = nullptr;
  const char* env = getenv("SA_TAIL_TRACE");
  if (!env) return;
  ++calls;
  const int at = atoi(env);
  if (calls == at - 1) {
    hipStreamSynchronize(st);
    hipMalloc(&buf, 64 * 1024);
    hipMemset(buf, 0, 64 * 1024);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tail_buf), &buf, sizeof buf);
    hipDeviceSynchronize();
  }
  if (calls != at) return;
  hipStreamSynchronize(st);
  unsigned long long* nul = nullptr;
  hipMemcpyToSymbol(HIP_SYMBOL(g_tail_buf), &nul, sizeof nul);
  unsigned long long h[8 * 64];
  hipMemcpy(h, buf, sizeof h, hipMemcpyDeviceToHost);
  if (FILE* f = fopen("gpurun_out/tail_trace.txt", "a")) {
    for (uint32_t z = 0; z < ns && z < 64; ++z) {
      const unsigned long long* t = h + z * 8;
      fprintf(f, "scene %u: scan %llu pack+union %llu label %llu sort %llu link %llu solve %llu out %llu  total %llu\n", z, t[1] - t[0],
              t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6], t[7] - t[0]);
    }
    fclose(f);
  }
}
// ... and of the many-workgroup tail's second kernel (SA_SOLVE_TRACE=<launch #>): per workgroup of scene 0, the 100 MHz clock every XCD
// shares at  0 entry | 1 own rows done | 2 mid-sized components served (own entries) | 3 big components served | 5 exit ;
// [6] = big << 32 | mid components of the scene ; [7] = big << 32 | mid components THIS workgroup took.
__device__ unsigned long long* g_solve_buf;
#define SOLVE_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.z == 0 && g_solve_buf && blockIdx.x < 512) g_solve_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SOLVE_NOTE(k, v) do { if (threadIdx.x == 0 && blockIdx.z == 0 && g_solve_buf && blockIdx.x < 512) g_solve_buf[blockIdx.x * 8 + (k)] = (v); } while (0)
// (the workgroup's FIRST mid-sized component, second half of the buffer: 0 taken | 1 rows listed | 2 columns hashed | 3 ranked | 4 matrix filled |
//  5 searched | 6 results out | [7] = rows << 32 | columns)
#define MID_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.z == 0 && g_solve_buf && blockIdx.x < 512 && g_solve_buf[4096 + blockIdx.x * 8 + (k)] == 0) g_solve_buf[4096 + blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MID_NOTE(k, v) do { if (threadIdx.x == 0 && blockIdx.z == 0 && g_solve_buf && blockIdx.x < 512 && g_solve_buf[4096 + blockIdx.x * 8 + (k)] == 0) g_solve_buf[4096 + blockIdx.x * 8 + (k)] = (v); } while (0)
static unsigned long long* g_solve_host = nullptr;
static int g_solve_calls = 0;
static void sa_solve_trace_before(hipStream_t st) {
  const char* env = getenv("SA_SOLVE_TRACE");
  if (!env) return;
  if (++g_solve_calls != atoi(env)) return;
  hipStreamSynchronize(st);
  hipMalloc(&g_solve_host, 64 * 1024);
  hipMemset(g_solve_host, 0, 64 * 1024);
  hipMemcpyToSymbol(HIP_SYMBOL(g_solve_buf), &g_solve_host, sizeof g_solve_host);
  hipDeviceSynchronize();
}
static void sa_solve_trace_after(hipStream_t st, uint32_t wgs) {
  const char* env = getenv("SA_SOLVE_TRACE");
  if (!env || g_solve_calls != atoi(env) || !g_solve_host) return;
  hipStreamSynchronize(st);
  unsigned long long* nul = nullptr;
  hipMemcpyToSymbol(HIP_SYMBOL(g_solve_buf), &nul, sizeof nul);
  std::vector<unsigned long long> h(8 * 1024);
  hipMemcpy(h.data(), g_solve_host, 64 * 1024, hipMemcpyDeviceToHost);
  g_solve_host = nullptr;
  if (wgs > 512) wgs = 512;
  if (FILE* f = fopen("gpurun_out/solve_trace.txt", "a")) {
    unsigned long long t0 = ~0ull;
    for (uint32_t w = 0; w < wgs; ++w) if (h[w * 8] && h[w * 8] < t0) t0 = h[w * 8];
    fprintf(f, "== k_assign_solve: %u workgroups, scene 0: %llu big + %llu mid-sized components; us after the first workgroup's entry (10 ns ticks), percentiles 10 / 50 / 90 / max\n",
            wgs, h[6] >> 32, h[6] & 0xffffffffull);
    const char* names[6] = {"entry", "own rows done", "mid-sized served (own entries)", "big components served", "", "exit"};
    for (int k = 0; k < 6; ++k) {
      std::vector<double> v;
      for (uint32_t w = 0; w < wgs; ++w) if (h[w * 8 + k]) v.push_back((double)(h[w * 8 + k] - t0) / 100.0);
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      fprintf(f, "   %-28s %7.2f / %7.2f / %7.2f / %7.2f   (%zu workgroups)\n", names[k], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back(), v.size());
    }
    {
      const char* mn[7] = {"taken", "rows listed", "columns hashed", "ranked", "matrix filled", "searched", "results out"};
      fprintf(f, "   a workgroup's first mid-sized component, us since it was taken, percentiles 10 / 50 / 90 / max:\n");
      for (int k = 1; k < 7; ++k) {
        std::vector<double> v;
        for (uint32_t w = 0; w < wgs; ++w) if (h[4096 + w * 8 + k] && h[4096 + w * 8]) v.push_back((double)(h[4096 + w * 8 + k] - h[4096 + w * 8]) / 100.0);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        fprintf(f, "      %-16s %7.2f / %7.2f / %7.2f / %7.2f   (%zu)\n", mn[k], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back(), v.size());
      }
      std::vector<double> rr, cc;
      for (uint32_t w = 0; w < wgs; ++w) if (h[4096 + w * 8 + 7]) { rr.push_back((double)(h[4096 + w * 8 + 7] >> 32)); cc.push_back((double)(h[4096 + w * 8 + 7] & 0xffffffffull)); }
      if (!rr.empty()) { std::sort(rr.begin(), rr.end()); std::sort(cc.begin(), cc.end());
        fprintf(f, "      rows %g / %g / %g / %g   columns %g / %g / %g / %g\n", rr[rr.size() / 10], rr[rr.size() / 2], rr[rr.size() * 9 / 10], rr.back(), cc[cc.size() / 10], cc[cc.size() / 2], cc[cc.size() * 9 / 10], cc.back()); }
    }
    unsigned long long mx_mid = 0, mx_big = 0;
    for (uint32_t w = 0; w < wgs; ++w) { mx_mid = std::max(mx_mid, h[w * 8 + 7] & 0xffffffffull); mx_big = std::max(mx_big, h[w * 8 + 7] >> 32); }
    fprintf(f, "   most components one workgroup took: %llu big, %llu mid-sized\n", mx_big, mx_mid);
    fclose(f);
  }
}
#else
#define TAIL_STAMP(k) do { } while (0)
#define SOLVE_STAMP(k) do { } while (0)
#define SOLVE_NOTE(k, v) do { } while (0)
#define MID_STAMP(k) do { } while (0)
#define MID_NOTE(k, v) do { } while (0)
static inline void sa_tail_trace_hook(hipStream_t, uint32_t) {}
static inline void sa_solve_trace_before(hipStream_t) {}
static inline void sa_solve_trace_after(hipStream_t, uint32_t) {}
#endif

// One 1024-thread workgroup per scene: edge lists -> LDS, components, solve, results.  The solver's duals / matches / search
// scratch live in LDS whenever the scene has at most 1024 tracks: every step of the shortest-path search is a chain of
// dependent accesses, ~10x cheaper in LDS than in L2.  VISUAL = the engine has a visual vote whose verdicts (row_has,
// vis_winner, col_excluded) must be honoured; plain SORT skips those loads altogether.
// Timeline at C3 (500 x 500 IoU, -DSA_TAIL_TRACE) before / after this version: scan 2.9 k cycles | pack + unite 8.1 k -> edge
// loads batched four at a time instead of one dependent round trip per edge | order rows inside components: 1024-key bitonic
// sort 6.8 k + link 0.7 k -> each row pushes itself on its root's LDS list, the solver thread orders the (short) list |
// solve 9.8 k | results 3.2 k.
// Workgroup barrier for phases that hand over LDS data only: __syncthreads() carries a workgroup-scope release, which on gfx9
// drains vmcnt as well (loads and stores share the counter) — every global load in flight would have to land before the
// barrier.  Here the long-latency loads (edge lists and track ids written by other XCDs: a trip to memory) are meant to stay in
// flight across the LDS phases, so only the LDS counter is drained.
__device__ __forceinline__ void sa_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// WORDS: the contraction's tiles reduced the BestFit vote into one 64-bit word per candidate and per track (SaParams::vote_words,
// visual_cosine_tile): thread q reads candidate q's word and track q's word, re-arms both, and the verdicts stay in registers
// (has / winner) and two LDS tables instead of going through k_bestfit_resolve's row_has / vis_winner / col_excluded — one
// dependent launch less per frame.
//
// The solve ("batched Jonker-Volgenant with per-row minima in LDS"):
//   1. greedy start, one thread per row, all rows at once: a row bids for the column of its heaviest usable edge (lowest column
//      on ties), a column goes to the lowest row that bids for it.  Under the duals u = -(heaviest gain), v = 0 those edges are
//      tight, so this is a feasible primal-dual start (what the shortest-path search would do for a row whose nearest column is
//      free, for every such row in ONE step).  In tracking frames almost every row keeps its bid.
//   2. the rows that lost their bid are the search roots of their connected component; every component that has any goes onto a
//      work queue;
//   3. the workgroup's 1024 / G groups of G lanes take components off the queue; a group orders the component's roots
//      (ascending: the canonical augmentation order) and runs sa_assign_component_coop<G> (sa_device.h): the search's two inner
//      loops — nearest labelled column, relax a row's edges — spread over the lanes, minima by lane reductions.
// A component of hundreds of rows (a crowd under a low IoU threshold) is then a few hundred microseconds of group work instead
// of seconds of one lane's dependent LDS chain; the usual one- and two-row components never reach step 3.
// Needs N <= SA_SMALL_N and T <= SA_SMALL_N (launcher; TC = 2: T <= SA_SMALL_T; wider frames: k_assign_small2 below): rows, columns and
// the usable edges (up to POOL of them; more stay in the HBM lists and are read from there) live in LDS.
// the dense solver (sa_dense.h): SA_DENSE_NT threads, each owning T / SA_DENSE_NT columns; components with at least SA_DENSE_MIN_ROOTS
// search roots on at least SA_DENSE_MIN_COLS columns (or whose edge lists stayed in HBM) go to it
#define SA_DENSE_NT 256
#define SA_DENSE_MIN_ROOTS 8u
#define SA_DENSE_MIN_COLS 128u
// The end of a scene's results, reported by the workgroup itself (done_seq != 0): every wave, when it has issued its last result, waits
// for its stores to be acknowledged (s_waitcnt vmcnt(0); the results are SYSTEM-scope stores — SA_OUT in k_assign_small —, acknowledged
// from the host's side of the link: a system-scope release per workgroup instead, i.e. an L2 write-back each, cost a 64-scene set
// 25 us) and counts itself in LDS; the wave that completes the count stores the launch's sequence number to the scene's completion word
// (a cache line of its own in the same block) — every result was acknowledged before that store was issued.  The host polls the word.
// Waves leave k_assign_small at three places; each of them reports.
__device__ __forceinline__ void sa_report_done(const SceneDev& S, uint64_t done_seq, uint32_t* s_done) {
  if (!done_seq) return;
  // (a workgroup-scope release alone is NOT that wait: the waves of a workgroup share their CU's memory pipeline, so the compiler
  // emits no s_waitcnt for it — measured: the word overtook the results)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) (expcnt / lgkmcnt untouched): this wave's stores have been acknowledged
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    const uint32_t before = atomicAdd(s_done, 1u);
    if (before + 1u == blockDim.x / WAVE) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // (the count seen: the other waves' acknowledgements are behind us)
      __hip_atomic_store(S.out_done, (unsigned long long)done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// TC: columns per thread — 1: T <= 1024; 2: T <= 2048 (a tracker loop's table once idle tracks linger: more tracks than detections is its
// normal state): every per-column array twice as long, the LDS edge pool given up for them (rows that lose their bid walk the HBM lists:
// rare in tracking frames); class words (SCN_WORDSK) with a register set per column — what keeps them out of k_assign_small2.
template <bool VISUAL, bool WORDS, int G, int TC = 1>
__global__ __launch_bounds__(SA_SMALL_N) void k_assign_small(const SceneDev* __restrict__ scenes, uint64_t done_seq) {
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  const uint32_t N = S.N, T = S.T;
  const uint32_t q = threadIdx.x;
  constexpr uint32_t TCAP = (uint32_t)TC * SA_SMALL_N;   // columns this instantiation holds
  constexpr int DNT = SA_DENSE_NT * TC;                  // threads of the dense solver: four columns each (eight per thread cost the two-column form spilled registers)
  __shared__ uint32_t s_done;   // waves that have reported (sa_report_done)
  // a result on its way to the host's mapped block: with completion words as a SYSTEM-scope store — such a store is acknowledged when it
  // has reached the host's memory, a plain one when the L2 has taken it (measured: behind plain stores the completion word overtook the
  // results it announces, s_waitcnt vmcnt(0) or not); otherwise plain (the dispatch's own end-of-kernel release covers it)
#define SA_OUT(ptr, val)                                                                                             \
  do {                                                                                                               \
    if (done_seq) __hip_atomic_store((ptr), (std::remove_cv_t<std::remove_reference_t<decltype(*(ptr))>>)(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
    else *(ptr) = (val);                                                                                             \
  } while (0)
  __shared__ uint32_t s_head[SA_SMALL_N];  // per component root: the rows that lost their greedy bid (pushed in any order)
  __shared__ uint32_t s_next[SA_SMALL_N];
  __shared__ int64_t s_u[SA_SMALL_N], s_v[TCAP], s_dist[TCAP];
  __shared__ int32_t s_rmatch[SA_SMALL_N], s_cmatch[TCAP], s_pred[TCAP];
  __shared__ uint32_t s_cstamp[TCAP], s_cscan[TCAP];
  __shared__ uint32_t s_lab[SA_SMALL_N];     // component root of a row with usable edges
  __shared__ uint32_t s_cwin[TCAP];          // per column: lowest row bidding for it
  __shared__ uint32_t s_rcount[SA_SMALL_N];  // per component root: search roots
  __shared__ uint32_t s_ccount[SA_SMALL_N];  // per component root: columns
  __shared__ uint32_t s_clist[TCAP];         // labelled columns of the running searches, one segment per component
  __shared__ uint32_t s_rlist[SA_SMALL_N];   // search roots in ascending order, one segment per component
  __shared__ uint32_t s_queue[SA_SMALL_N];   // components waiting for a group
  __shared__ uint32_t s_ctr[8];              // queue length | next queue entry | top of s_clist | top of s_rlist | dense queue length
  __shared__ unsigned long long s_part[2 * (DNT / 64)];  // the dense solver's per-wave minima (sa_wg_min_u64)
  // The edge lists the positional tiles left behind live in HBM, one strided row per candidate: every access from here on would be
  // a dependent, uncoalesced round trip (the solve is a chain of them).  They are packed ONCE into an LDS pool — an
  // exclusive scan of the row counts gives the offsets — and the row duals, the connected components of the usable graph
  // (rows without a visual verdict) and the solve itself then run out of LDS.  A scene whose lists do not fit (dense
  // Mahalanobis frames, crowds under a low threshold) keeps the HBM lists as the solver's edge storage.
  constexpr uint32_t POOL = TC == 1 ? 3072 : 0;   // (two columns per thread: the pool's 36 KB are the second half of the column arrays)
  __shared__ uint32_t s_parent[SA_SMALL_N + TCAP];
  __shared__ uint32_t s_ecnt[SA_SMALL_N], s_eoff[SA_SMALL_N], s_wsum[SA_SMALL_N / WAVE];
  __shared__ uint32_t s_ecol[POOL ? POOL : 1];
  __shared__ int64_t s_egain[POOL ? POOL : 1];
  TAIL_STAMP(0);
  const uint32_t rawcnt = q < N ? S.e_cnt[q] : 0u;
  if (q == 0) {  // what the first phase raised goes out with the results; re-armed for the next frame
    SA_OUT(S.out_stats + 0, S.stats[0]);
    SA_OUT(S.out_stats + 1, 0u);   // (k_assign_solve: a bounded wait ran out — the host refuses the frame's results)
    S.stats[0] = 0u;
    s_done = 0u;           // (barriers follow before any wave can leave)
  }
  __shared__ uint8_t s_cexcl[WORDS ? TCAP : 4];         // excluded_tracks as bytes, for the solver's HBM-list variant
  __shared__ uint32_t s_bt[WORDS ? SA_SMALL_N : 1];     // candidate -> its best column (SA_NONE: no group at all)
  __shared__ uint32_t s_cq[WORDS ? TCAP : 1];           // column -> its best candidate (SA_NONE: no group at all)
  bool has_verdict;
  int32_t vw0 = -1;
  uint32_t bt = SA_NONE;
  // SCN_WORDSK (deeper banks through the whole-track tiles of the contraction, sa_gemm.hip): K words per candidate and per track, one per count class —
  // (key of the f32 sum of the group's weights << 32 | index).  Whether a candidate has ANY group is known at once (it decides
  // whether the row takes part in the positional vote); WHICH group wins needs the frame's max_dist: W = c max_dist - sum, heaviest
  // wins, lowest index among equals — folded from the first phase's per-tile slots by this workgroup (one slot per thread, the
  // wave maxima through LDS at the barrier that is there anyway), then one more barrier for the two tables.
  __shared__ uint32_t s_wmk[WORDS ? SA_SMALL_N / WAVE : 1];
  unsigned long long rcls[WORDS ? SA_CLS_MAXK : 1], ccls[TC][WORDS ? SA_CLS_MAXK : 1];
  bool clsmode = false;
  if constexpr (WORDS) clsmode = (S.flags & SCN_WORDSK) != 0;
  if constexpr (WORDS) if (clsmode) {
    const uint32_t K = S.K;
    bool any = false;
    // (the first phase's max-key slots are requested FIRST, so that they travel with the class words: behind the words' processing the
    // loop below was a second trip to memory — one of the ~3 us the class-word tail took over the single-word one)
    uint32_t mk = 0, mk0 = q < S.nkeys ? S.vis_max_key[q] : 0u;
#pragma unroll
    for (uint32_t c = 0; c < SA_CLS_MAXK; ++c) {
      // all 2 x SA_CLS_MAXK loads issued together, whatever K is (the clamped index re-reads a word that is needed anyway): with the
      // load inside `c < K ? ... : ~0` every class became a scalar branch with its own load + s_waitcnt vmcnt(0) — K round trips
      // to memory one after the other, ~1 us per class
      rcls[c] = S.row_cls[(size_t)(q < N ? q : 0u) * K + (c < K ? c : K - 1u)];
#pragma unroll
      for (int cc = 0; cc < TC; ++cc) {
        const uint32_t j = q + (uint32_t)cc * SA_SMALL_N;
        ccls[cc][c] = S.col_cls[(size_t)(j < T ? j : 0u) * K + (c < K ? c : K - 1u)];
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (uint32_t c = 0; c < SA_CLS_MAXK; ++c) {
      rcls[c] = (c < K && q < N) ? rcls[c] : ~0ull;
#pragma unroll
      for (int cc = 0; cc < TC; ++cc) ccls[cc][c] = (c < K && q + (uint32_t)cc * SA_SMALL_N < T) ? ccls[cc][c] : ~0ull;
    }
#pragma unroll
    for (uint32_t c = 0; c < SA_CLS_MAXK; ++c) {
      if (rcls[c] != ~0ull) S.row_cls[(size_t)q * K + c] = ~0ull;  // re-armed (most classes of a row are empty: nothing to store)
#pragma unroll
      for (int cc = 0; cc < TC; ++cc) {
        const uint32_t j = q + (uint32_t)cc * SA_SMALL_N;
        if (ccls[cc][c] != ~0ull) S.col_cls[(size_t)j * K + c] = ~0ull;
      }
      if (S.tap_row_best && c < K) {  // SA_FLAG_TAP: the class words as the first phase left them ([N K] then [T K])
        if (q < N) S.tap_row_best[(size_t)q * K + c] = rcls[c];
#pragma unroll
        for (int cc = 0; cc < TC; ++cc) {
          const uint32_t j = q + (uint32_t)cc * SA_SMALL_N;
          if (j < T) S.tap_col_best[(size_t)j * K + c] = ccls[cc][c];
        }
      }
      any = any || rcls[c] != ~0ull;
    }
    mk = mk0;
    for (uint32_t i = q + SA_SMALL_N; i < S.nkeys; i += SA_SMALL_N) {   // (more than 1024 tiles: frames beyond this tail's reach today)
      const uint32_t v = S.vis_max_key[i];
      mk = v > mk ? v : mk;
    }
    for (int o = WAVE / 2; o > 0; o >>= 1) {
      const uint32_t ok = __shfl_xor(mk, o);
      mk = ok > mk ? ok : mk;
    }
    if (q % WAVE == 0) s_wmk[q / WAVE] = mk;
    has_verdict = any;  // feature_winners.contains_key(q)
    s_bt[q] = SA_NONE;  // (rows / columns beyond N / T)
#pragma unroll
    for (int cc = 0; cc < TC; ++cc) s_cq[q + (uint32_t)cc * SA_SMALL_N] = SA_NONE;
  }
  if constexpr (WORDS) {
   if (!clsmode) {
    // (weight key << 32 | index), all ones = no group at all; lowest weight wins, lowest index on ties — k_bestfit_resolve's order
    const unsigned long long rb = q < N ? S.row_best[q] : ~0ull;
    unsigned long long cb[TC];
#pragma unroll
    for (int cc = 0; cc < TC; ++cc) { const uint32_t j = q + (uint32_t)cc * SA_SMALL_N; cb[cc] = j < T ? S.col_best[j] : ~0ull; }
    if (q < N) S.row_best[q] = ~0ull;
#pragma unroll
    for (int cc = 0; cc < TC; ++cc) { const uint32_t j = q + (uint32_t)cc * SA_SMALL_N; if (j < T) S.col_best[j] = ~0ull; }
    if (S.tap_row_best) {  // SA_FLAG_TAP: the words as the first phase left them
      if (q < N) S.tap_row_best[q] = rb;
#pragma unroll
      for (int cc = 0; cc < TC; ++cc) { const uint32_t j = q + (uint32_t)cc * SA_SMALL_N; if (j < T) S.tap_col_best[j] = cb[cc]; }
    }
    const uint32_t imask = (S.flags & SCN_WORDS10) ? 1023u : 0xffffffffu;  // deeper banks: (inverted weight key << 10) | index, k_bestfit_tile
    bt = rb != ~0ull ? ((uint32_t)rb & imask) : SA_NONE;
    has_verdict = bt != SA_NONE;  // feature_winners.contains_key(q)
    s_bt[q] = bt;
#pragma unroll
    for (int cc = 0; cc < TC; ++cc) s_cq[q + (uint32_t)cc * SA_SMALL_N] = cb[cc] != ~0ull ? ((uint32_t)cb[cc] & imask) : SA_NONE;
   }
  } else {
    has_verdict = VISUAL && q < N && S.row_has[q];
    vw0 = (VISUAL && q < N) ? S.vis_winner[q] : -1;  // with the first round trip, not after the scan
  }
  // excluded_tracks: column j was won by the candidate that is best in it iff that candidate's own best column is j
  auto excluded = [&](uint32_t j) -> bool {
    if constexpr (WORDS) {
      const uint32_t c = s_cq[j];
      return c != SA_NONE && s_bt[c] == j;
    } else return S.col_excluded[j] != 0;
  };
  // Plain SORT (with a visual vote most rows arrive decided and their lists are never read): the first four edges of the row
  // are fetched before their count is known (what lies beyond the count is stale but
  // addressable), so that this round trip — the lists were written by other XCDs a moment ago, it goes to memory — overlaps the
  // count's.  What depends on them — the excluded-column flags and the track ids the results will need — is requested right after
  // the scan; the ids are not awaited before the results are written (sa_lds_barrier).  Tracking frames rarely have more than
  // four edges in a row.
  uint32_t sj[4];
  int64_t sg[4];
  {
    const SaEdge SA_G* row = S.e_edge + (q < N ? q : 0);  // slot-major: edge k of row q at [k * N + q]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool in = !VISUAL && q < N && (uint32_t)k < T;  // T == 0: the lists have no capacity at all
      const SaEdge ed = in ? sa_ldg(row + (size_t)k * N) : SaEdge{0, 0u, 0u};
      sj[k] = ed.col;
      sg[k] = ed.gain;
    }
  }
  if (q < N) S.e_cnt[q] = 0;  // left clean for the next frame's positional tiles (nothing below reads the global counter)
  if (S.tap_ecnt && q < N) S.tap_ecnt[q] = rawcnt;  // SA_FLAG_TAP: how many edge records the positional tiles appended to this row
  const uint32_t mycnt = (q < N && !has_verdict) ? rawcnt : 0u;
  s_rmatch[q] = -1;
  s_ecnt[q] = mycnt;
  s_head[q] = SA_NONE;
  s_next[q] = SA_NONE;
#pragma unroll
  for (int cc = 0; cc <= TC; ++cc) s_parent[q + (uint32_t)cc * SA_SMALL_N] = q + (uint32_t)cc * SA_SMALL_N;
#pragma unroll
  for (int cc = 0; cc < TC; ++cc) {
    const uint32_t j = q + (uint32_t)cc * SA_SMALL_N;
    s_v[j] = 0; s_cmatch[j] = -1; s_cstamp[j] = 0; s_cscan[j] = 0; s_cwin[j] = SA_NONE;
  }
  s_rcount[q] = 0; s_ccount[q] = 0; s_lab[q] = SA_NONE;
  if (q < 8) s_ctr[q] = 0;
  // exclusive scan of mycnt over the 1024 threads: wave scan, then the 16 wave totals
  uint32_t incl = mycnt;
  {
    const uint32_t lane = q % WAVE;
    for (int o = 1; o < WAVE; o <<= 1) {
      uint32_t up = __shfl_up(incl, o);
      if (lane >= (uint32_t)o) incl += up;
    }
    if (lane == WAVE - 1) s_wsum[q / WAVE] = incl;
  }
  sa_lds_barrier();
  if constexpr (WORDS) if (clsmode) {
    uint32_t mk = 0;
#pragma unroll
    for (uint32_t w2 = 0; w2 < SA_SMALL_N / WAVE; ++w2) mk = s_wmk[w2] > mk ? s_wmk[w2] : mk;
    const double max_dist = mk ? (double)sa_key_f32(mk) : -1.0;
    auto best_of = [&](const unsigned long long* cls) -> uint32_t {
      double bw = 0.0;
      uint32_t bi = SA_NONE;
#pragma unroll
      for (uint32_t c = 0; c < SA_CLS_MAXK; ++c) {
        if (cls[c] == ~0ull) continue;
        const double w = (double)(c + 1u) * max_dist - (double)sa_key_f32((uint32_t)(cls[c] >> 32));
        const uint32_t i = (uint32_t)cls[c];
        if (bi == SA_NONE || w > bw || (w == bw && i < bi)) { bw = w; bi = i; }
      }
      return bi;
    };
    bt = best_of(rcls);
    s_bt[q] = bt;
#pragma unroll
    for (int cc = 0; cc < TC; ++cc) s_cq[q + (uint32_t)cc * SA_SMALL_N] = best_of(ccls[cc]);
    sa_lds_barrier();
  }
  if constexpr (WORDS) {
    if (has_verdict && s_cq[bt] == q) vw0 = (int32_t)bt;  // the candidate that is best in its own best column wins it
#pragma unroll
    for (int cc = 0; cc < TC; ++cc) { const uint32_t j = q + (uint32_t)cc * SA_SMALL_N; s_cexcl[j] = j < T && excluded(j); }
  }
  uint32_t woff = 0, total = 0;
  for (uint32_t w2 = 0; w2 < SA_SMALL_N / WAVE; ++w2) {
    const uint32_t v = s_wsum[w2];
    if (w2 < q / WAVE) woff += v;
    total += v;
  }
  if (total == 0) {  // nothing left for the positional vote (every row decided visually, or no edge at all)
    if (q < N) {
      uint64_t id = 0;
      uint8_t vt = SA_VOTE_NONE;
      const int32_t vw = vw0;
      if (vw >= 0) { id = S.t_ids[vw]; vt = SA_VOTE_VISUAL; }
      SA_OUT(S.out_track_id + q, id);
      SA_OUT(S.out_vote + q, vt);
      S.win_col[q] = vw >= 0 ? vw : -1;
      SA_OUT(S.out_win + q, vw >= 0 ? vw : -1);
    }
    sa_report_done(S, done_seq, &s_done);
    return;
  }
  TAIL_STAMP(1);
  if (VISUAL) {
    const SaEdge SA_G* row = S.e_edge + (q < N ? q : 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const SaEdge ed = (uint32_t)k < mycnt ? sa_ldg(row + (size_t)k * N) : SaEdge{0, 0u, 0u};
      sj[k] = ed.col;
      sg[k] = ed.gain;
    }
  }
  bool sx[4];
  uint64_t sid[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = (uint32_t)k < mycnt;
    sx[k] = VISUAL && in && excluded(sj[k]);
    sid[k] = in ? S.t_ids[sj[k]] : 0ull;
  }
  const bool pool = total <= POOL;
  const uint32_t myoff = woff + incl - mycnt;
  s_eoff[q] = myoff;
  int64_t maxg = 0;
  uint32_t bcol = SA_NONE;  // the column of the heaviest usable edge, lowest column on ties: this row's bid
  uint32_t usable = 0;
  if (mycnt) {
    const SaEdge SA_G* row = S.e_edge + q;
    // four edges per step: all their loads (and, with a visual vote, the dependent excluded-column flags) are in flight together
    for (uint32_t e0 = 0; e0 < mycnt; e0 += 4) {
      uint32_t jj[4];
      int64_t gg[4];
      bool skip[4];
      if (e0 == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { jj[k] = sj[k]; gg[k] = sg[k]; skip[k] = !((uint32_t)k < mycnt) || sx[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool in = e0 + k < mycnt;
          const SaEdge ed = in ? sa_ldg(row + (size_t)(e0 + k) * N) : SaEdge{0, 0u, 0u};
          jj[k] = ed.col;
          gg[k] = ed.gain;
          skip[k] = !in;
        }
        if (VISUAL) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (!skip[k]) skip[k] = excluded(jj[k]);  // excluded_tracks (visual_sort/voting.rs:62-71): dropped while packing
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (skip[k]) continue;
        const uint32_t j = jj[k];
        if (pool) { s_ecol[myoff + usable] = j; s_egain[myoff + usable] = gg[k]; }
        ++usable;
        if (gg[k] > maxg || (gg[k] == maxg && j < bcol)) { maxg = gg[k]; bcol = j; }
        sa_uf_union(s_parent, q, N + j);
      }
    }
    if (pool) s_ecnt[q] = usable;  // the packed list holds usable edges only (the HBM list keeps them all: the solver skips there)
  }
  s_u[q] = -maxg;
  if (usable) atomicMin(&s_cwin[bcol], q);  // the bid (bcol is set whenever a usable edge exists: gains are > 0)
  sa_lds_barrier();
  TAIL_STAMP(2);
  // component label = root of the union-find tree = the lowest vertex = the component's first row.  Columns count themselves
  // into their component (the room a search's list of labelled columns can need); rows learn whether their bid held.
  uint32_t lab = SA_NONE;
  if (usable) {
    lab = sa_uf_find(s_parent, q);
    s_lab[q] = lab;
    if (s_cwin[bcol] == q) { s_rmatch[q] = (int32_t)bcol; s_cmatch[bcol] = (int32_t)q; }
    else {
      s_next[q] = atomicExch(&s_head[lab], q);
      atomicAdd(&s_rcount[lab], 1u);
    }
  }
#pragma unroll
  for (int cc = 0; cc < TC; ++cc) {
    const uint32_t j = q + (uint32_t)cc * SA_SMALL_N;
    if (j < T) {
      const uint32_t r = sa_uf_find(s_parent, N + j);
      if (r < N) atomicAdd(&s_ccount[r], 1u);  // a column without usable edges is its own root (>= N)
    }
  }
  sa_lds_barrier();
  TAIL_STAMP(3);
  // A component with search roots goes onto one of two queues: the wavefronts' (bottom of s_queue) or — many roots on many columns,
  // or edge lists that did not fit the LDS pool: every relax step would walk HBM — the dense solver's (top of s_queue; the mark
  // in s_ccount tells its rows that their results come later).  sa_dense.h has the why.
  if (lab == q && s_head[q] != SA_NONE) {
    const bool dense = s_rcount[q] >= SA_DENSE_MIN_ROOTS && (s_ccount[q] >= SA_DENSE_MIN_COLS || !pool);
    if (dense) {
      s_queue[SA_SMALL_N - 1u - atomicAdd(&s_ctr[4], 1u)] = q;
      s_ccount[q] |= 0x80000000u;
    } else s_queue[atomicAdd(&s_ctr[0], 1u)] = q;
  }
  sa_lds_barrier();
  TAIL_STAMP(4);
  // groups of G lanes take components off the queue
  {
    const uint32_t lane = q % G;
    const uint32_t nq = s_ctr[0];
    sa_coop_ws w;
    w.e_cnt = s_ecnt;
    w.u = s_u; w.v = s_v; w.rmatch = s_rmatch; w.cmatch = s_cmatch; w.dist = s_dist; w.pred = s_pred; w.cstamp = s_cstamp; w.cscan = s_cscan;
    for (;;) {
      uint32_t take[1], seg[2];
      if (lane == 0) take[0] = atomicAdd(&s_ctr[1], 1u);
      const uint32_t k = sa_coop_bcast<G>(take);
      if (k >= nq) break;
      const uint32_t root = s_queue[k];
      const uint32_t R = s_rcount[root], C = s_ccount[root];
      if (lane == 0) { seg[0] = atomicAdd(&s_ctr[2], C); seg[1] = atomicAdd(&s_ctr[3], R); }
      const uint32_t cbase = sa_coop_bcast<G>(seg), rbase = sa_coop_bcast<G>(seg + 1);
      uint32_t* roots = s_rlist + rbase;
      if (R <= (uint32_t)G) {
        // a short list: every lane walks it, lane l keeps element l, ranks by comparison, one store each
        uint32_t cur = s_head[root], mine = SA_NONE;
        for (uint32_t st = 0; st < R; ++st) {
          if (st == lane) mine = cur;
          cur = s_next[cur];
        }
        uint32_t rank = 0;
        for (uint32_t st = 0; st < R; ++st) {
          const uint32_t other = __shfl(mine, st, G);
          rank += other < mine ? 1u : 0u;
        }
        if (lane < R) roots[rank] = mine;
      } else {
        // a long list: compact the scene's rows (lab == root, bid lost) in row order, G rows per step
        uint32_t cnt = 0;
        for (uint32_t r0 = 0; r0 < N; r0 += G) {
          const uint32_t row = r0 + lane;
          bool f[1];
          f[0] = row < N && s_lab[row] == root && s_rmatch[row] < 0;
          uint32_t tot;
          const uint32_t rk = sa_coop_rank<G>(f, lane, &tot);
          if (f[0]) roots[cnt + rk] = row;
          cnt += tot;
        }
      }
      sa_coop_sync<G>();
      w.clist = s_clist + cbase;
      // one call site per address space of the edge storage (LDS pool or the HBM lists), so that every pointer of the work set has
      // ONE known address space after inlining — "LDS or global, decided at run time" compiles to flat_* accesses
      if (pool) {
        w.e_col = s_ecol; w.e_gain = s_egain; w.ecs = 1; w.egs = 1; w.rcs = 1; w.rgs = 1; w.estride = 0; w.e_off = s_eoff; w.excluded = nullptr;
        sa_assign_component_coop<G>(w, roots, R);
      } else {
        // slot-major lists: row r starts at record r, consecutive edges are N records apart
        w.e_col = (const uint32_t*)S.e_edge + 2; w.e_gain = (const int64_t*)S.e_edge; w.ecs = 4 * N; w.egs = 2 * N; w.rcs = 4; w.rgs = 2; w.e_off = nullptr; w.estride = 1;
        if constexpr (WORDS) w.excluded = s_cexcl;
        else w.excluded = VISUAL ? (const uint8_t*)S.col_excluded : nullptr;
        sa_assign_component_coop<G>(w, roots, R);
      }
    }
  }
  TAIL_STAMP(5);
  sa_lds_barrier();  // rmatch is in LDS
  TAIL_STAMP(6);
  const uint32_t nd = s_ctr[4];  // components waiting for the dense solver (tracking frames: none)
  const bool mine_later = nd && usable && (s_ccount[lab] & 0x80000000u);
  if (q < N && !mine_later) {
    uint64_t id = 0;
    uint8_t vt = SA_VOTE_NONE;
    int32_t win = -1;
    const int32_t vw = vw0;
    if (vw >= 0) { id = S.t_ids[vw]; vt = SA_VOTE_VISUAL; win = vw; }
    else if (!has_verdict) {
      int32_t c = s_rmatch[q];
      if (c >= 0) {
        bool found = false;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((uint32_t)k < mycnt && sj[k] == (uint32_t)c) { id = sid[k]; found = true; }
        if (!found) id = S.t_ids[c];
        vt = SA_VOTE_POSITIONAL;
        win = c;
      }
    }
    SA_OUT(S.out_track_id + q, id);
    SA_OUT(S.out_vote + q, vt);
    S.win_col[q] = win;
    SA_OUT(S.out_win + q, win);
  }
  if (nd) {
    // The dense solver runs on SA_DENSE_NT threads (one wave per SIMD: a search step is a chain of dependent instructions, more
    // waves per SIMD only stretch it): the other waves are done — a barrier waits for the surviving waves only.
    if (q >= (uint32_t)DNT) { sa_report_done(S, done_seq, &s_done); return; }
    uint32_t rtop = s_ctr[3];
    for (uint32_t k = 0; k < nd; ++k) {
      const uint32_t root = s_queue[SA_SMALL_N - 1u - k];
      const uint32_t R = s_rcount[root];
      uint32_t* roots = s_rlist + rtop;
      rtop += R;
      // the search roots, ascending (wave 0: ballot compaction of the scene's rows), and the component's gains into the dense matrix
      if (q < WAVE) {
        uint32_t cnt = 0;
        for (uint32_t r0 = 0; r0 < N; r0 += WAVE) {
          const uint32_t row = r0 + q;
          const bool f = row < N && s_lab[row] == root && s_rmatch[row] < 0;
          const unsigned long long m = __ballot(f);
          if (f) roots[cnt + (uint32_t)__popcll(m & ((1ull << q) - 1ull))] = row;
          cnt += (uint32_t)__popcll(m);
        }
      }
      if (q == 0) s_ctr[5] = 0;  // the component's heaviest gain (32-bit variant of the solver when it is small enough)
      sa_lds_barrier();
      uint32_t mg = 0;
      for (uint32_t row = q; row < N; row += (uint32_t)DNT) {
        if (s_lab[row] != root) continue;
        const int64_t heaviest = -s_u[row];
        const uint32_t h32 = heaviest > 0x7fffffffll ? 0x7fffffffu : (uint32_t)heaviest;
        mg = h32 > mg ? h32 : mg;
        int64_t SA_G* drow = S.dense + (size_t)row * T;
        const uint32_t cnt = s_ecnt[row];
        if (pool) {
          const uint32_t off = s_eoff[row];
          for (uint32_t e = 0; e < cnt; ++e) drow[s_ecol[off + e]] = s_egain[off + e];
        } else {
          const SaEdge SA_G* ep = S.e_edge + row;  // slot-major lists, excluded columns still inside; four records per round trip
          for (uint32_t e0 = 0; e0 < cnt; e0 += 4) {
            SaEdge ed[4];
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) ed[k2] = e0 + k2 < cnt ? sa_ldg(ep + (size_t)(e0 + k2) * N) : SaEdge{0, 0u, 0u};
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2)
              if (e0 + k2 < cnt && !(VISUAL && excluded(ed[k2].col))) drow[ed[k2].col] = ed[k2].gain;
          }
        }
      }
      if (mg) atomicMax(&s_ctr[5], mg);
      __syncthreads();
      {
        sa_dense_ws w;
        w.gain = (const int64_t*)S.dense; w.ld = T; w.T = T;
        w.u = s_u; w.rmatch = s_rmatch; w.cmatch = s_cmatch; w.pred = s_pred; w.part = s_part;
        if (s_ctr[5] <= (uint32_t)SA_DENSE_K32_MAXGAIN) sa_assign_component_dense<DNT, TCAP / DNT, true>(w, roots, R);
        else sa_assign_component_dense<DNT, TCAP / DNT, false>(w, roots, R);
      }
      // results of the component's rows (none of them holds a visual verdict), and the matrix left clean for the next frame
      for (uint32_t row = q; row < N; row += (uint32_t)DNT) {
        if (s_lab[row] != root) continue;
        const int32_t c = s_rmatch[row];
        SA_OUT(S.out_track_id + row, c >= 0 ? S.t_ids[c] : 0ull);
        SA_OUT(S.out_vote + row, c >= 0 ? SA_VOTE_POSITIONAL : SA_VOTE_NONE);
        S.win_col[row] = c;
        SA_OUT(S.out_win + row, c);
        int64_t SA_G* drow = S.dense + (size_t)row * T;
        const uint32_t cnt = s_ecnt[row];
        if (pool) {
          const uint32_t off = s_eoff[row];
          for (uint32_t e = 0; e < cnt; ++e) drow[s_ecol[off + e]] = 0;
        } else {
          const SaEdge SA_G* ep = S.e_edge + row;
          for (uint32_t e0 = 0; e0 < cnt; e0 += 4) {
            uint32_t cj[4];
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) cj[k2] = e0 + k2 < cnt ? sa_ldg(ep + (size_t)(e0 + k2) * N).col : 0u;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2)
              if (e0 + k2 < cnt) drow[cj[k2]] = 0;
          }
        }
      }
      __syncthreads();
    }
  }
  sa_report_done(S, done_seq, &s_done);
  TAIL_STAMP(7);
}

// The same tail for wider frames — RC rows and TC columns per thread: 2 x 2 = up to SA_SMALL_T detections AND tracks (C4: 2000 x 2000
// oriented boxes), 1 x 4 = up to 1024 detections against 4096 tracks (a crowd's tracker loop: `sdt`, 1000 x 2500).
// What the 1024-row form keeps in LDS does not fit twice: the edge pool is given up (searches walk the HBM lists: rows that lose their
// bid are rare in tracking frames), a running search's labels and distances (cstamp / cscan / dist / pred / its column list) live in
// the scene's HBM arrays (sa_mem_wg: relaxed workgroup-scope accesses — one wave works on a component, on its CU's L1), row lists and
// labels are 16-bit.  No class words (the host keeps such frames on the many-workgroup tail).  Every decision — bids, roots in ascending
// order, the solvers — is k_assign_small's: the same ids.
template <bool VISUAL, bool WORDS, int G, int RC = 2, int TC = 2>
__global__ __launch_bounds__(SA_SMALL_N) void k_assign_small2(const SceneDev* __restrict__ scenes, uint64_t done_seq) {
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  const uint32_t N = S.N, T = S.T;
  const uint32_t q = threadIdx.x;
  constexpr uint32_t RCAP = (uint32_t)RC * SA_SMALL_N, TCAP = (uint32_t)TC * SA_SMALL_N;   // rows / columns per thread: q, q + 1024, ...
  constexpr int DNT = TC <= 2 ? 512 : 1024;   // threads of the dense solver, four columns each
  static_assert(RCAP <= 2048u && (RC + TC) <= 5, "16-bit row links; LDS");
  constexpr uint32_t NONE16 = 0xffffu;
  __shared__ uint32_t s_done;
  __shared__ uint32_t s_head[RCAP];            // per component root: the rows that lost their greedy bid
  __shared__ uint16_t s_next[RCAP];
  __shared__ int64_t s_u[RCAP], s_v[TCAP];
  __shared__ int32_t s_rmatch[RCAP], s_cmatch[TCAP];
  __shared__ uint16_t s_lab[RCAP];             // component root of a row with usable edges (NONE16: none)
  __shared__ uint32_t s_cwin[TCAP];            // per column: lowest row bidding for it
  __shared__ uint32_t s_rcount[RCAP];          // per component root: search roots
  __shared__ uint32_t s_ccount[RCAP];          // per component root: columns
  __shared__ uint32_t s_rlist[RCAP];           // search roots in ascending order, one segment per component
  __shared__ uint16_t s_queue[RCAP];           // components waiting for a group (bottom) / for the dense solver (top)
  __shared__ uint32_t s_ctr[8];
  __shared__ unsigned long long s_part[2 * (DNT / 64)];
  __shared__ uint32_t s_parent[RCAP + TCAP];
  __shared__ uint32_t s_ecnt[RCAP], s_wsum[SA_SMALL_N / WAVE];
  __shared__ uint8_t s_cexcl[WORDS ? TCAP : 4];
  __shared__ uint32_t s_bt[WORDS ? RCAP : 1];
  __shared__ uint32_t s_cq[WORDS ? TCAP : 1];
  __shared__ uint32_t s_wmk[WORDS ? SA_SMALL_N / WAVE : 1];   // class words: the wave maxima of the first phase's max-key slots
  TAIL_STAMP(0);
  uint32_t rawcnt[RC];
#pragma unroll
  for (int rr = 0; rr < RC; ++rr) { const uint32_t row = q + (uint32_t)rr * SA_SMALL_N; rawcnt[rr] = row < N ? S.e_cnt[row] : 0u; }
  if (q == 0) {
    SA_OUT(S.out_stats + 0, S.stats[0]);
    SA_OUT(S.out_stats + 1, 0u);
    S.stats[0] = 0u;
    s_done = 0u;
  }
  bool has_verdict[RC];
  int32_t vw0[RC];
  uint32_t bt[RC];
#pragma unroll
  for (int rr = 0; rr < RC; ++rr) { vw0[rr] = -1; bt[rr] = SA_NONE; has_verdict[rr] = false; }
  // Class words (SCN_WORDSK: banks of 2..8 observations, k_assign_small has the why): which group wins needs the frame's max_dist, known
  // behind the first barrier — here the K words of a row / column are read TWICE (whether a row has any group at all now, the winners
  // behind the barrier, one row or column at a time) instead of being held in registers across it: two rows and two to four columns
  // per thread would be 48 .. 80 registers of words.
  bool clsmode = false;
  if constexpr (WORDS) clsmode = (S.flags & SCN_WORDSK) != 0;
  if constexpr (WORDS) if (clsmode) {
    const uint32_t K = S.K;
    uint32_t mk = q < S.nkeys ? S.vis_max_key[q] : 0u;
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      const uint32_t row = q + (uint32_t)rr * SA_SMALL_N;
      bool any = false;
#pragma unroll
      for (uint32_t c = 0; c < SA_CLS_MAXK; ++c) {
        const unsigned long long w = S.row_cls[(size_t)(row < N ? row : 0u) * K + (c < K ? c : K - 1u)];
        any = any || (c < K && row < N && w != ~0ull);
      }
      has_verdict[rr] = any;
      s_bt[row] = SA_NONE;
    }
#pragma unroll
    for (int cc = 0; cc < TC; ++cc) s_cq[q + (uint32_t)cc * SA_SMALL_N] = SA_NONE;
    for (uint32_t i = q + SA_SMALL_N; i < S.nkeys; i += SA_SMALL_N) {
      const uint32_t v = S.vis_max_key[i];
      mk = v > mk ? v : mk;
    }
    for (int o = WAVE / 2; o > 0; o >>= 1) {
      const uint32_t ok = __shfl_xor(mk, o);
      mk = ok > mk ? ok : mk;
    }
    if (q % WAVE == 0) s_wmk[q / WAVE] = mk;
  }
  if constexpr (WORDS) {
   if (!clsmode) {
    unsigned long long rb[RC], cb[TC];
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) { const uint32_t i = q + (uint32_t)rr * SA_SMALL_N; rb[rr] = i < N ? S.row_best[i] : ~0ull; }
#pragma unroll
    for (int cc = 0; cc < TC; ++cc) { const uint32_t j = q + (uint32_t)cc * SA_SMALL_N; cb[cc] = j < T ? S.col_best[j] : ~0ull; }
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      const uint32_t i = q + (uint32_t)rr * SA_SMALL_N;
      if (i < N) S.row_best[i] = ~0ull;
      if (S.tap_row_best && i < N) S.tap_row_best[i] = rb[rr];
      bt[rr] = rb[rr] != ~0ull ? (uint32_t)rb[rr] : SA_NONE;
      has_verdict[rr] = bt[rr] != SA_NONE;
      s_bt[i] = bt[rr];
    }
#pragma unroll
    for (int cc = 0; cc < TC; ++cc) {
      const uint32_t j = q + (uint32_t)cc * SA_SMALL_N;
      if (j < T) S.col_best[j] = ~0ull;
      if (S.tap_row_best && j < T) S.tap_col_best[j] = cb[cc];
      s_cq[j] = cb[cc] != ~0ull ? (uint32_t)cb[cc] : SA_NONE;
    }
   }
  } else {
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      const uint32_t row = q + (uint32_t)rr * SA_SMALL_N;
      has_verdict[rr] = VISUAL && row < N && S.row_has[row];
      vw0[rr] = (VISUAL && row < N) ? S.vis_winner[row] : -1;
    }
  }
  auto excluded = [&](uint32_t j) -> bool {
    if constexpr (WORDS) {
      const uint32_t c = s_cq[j];
      return c != SA_NONE && s_bt[c] == j;
    } else return S.col_excluded[j] != 0;
  };
  uint32_t sj[RC][4];
  int64_t sg[RC][4];
#pragma unroll
  for (int rr = 0; rr < RC; ++rr) {
    const uint32_t row = q + (uint32_t)rr * SA_SMALL_N;
    const SaEdge SA_G* rp = S.e_edge + (row < N ? row : 0);  // slot-major: edge k of row r at [k * N + r]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool in = !VISUAL && row < N && (uint32_t)k < T;
      const SaEdge ed = in ? sa_ldg(rp + (size_t)k * N) : SaEdge{0, 0u, 0u};
      sj[rr][k] = ed.col;
      sg[rr][k] = ed.gain;
    }
  }
  uint32_t mycnt[RC];
#pragma unroll
  for (int rr = 0; rr < RC; ++rr) {
    const uint32_t row = q + (uint32_t)rr * SA_SMALL_N;
    if (row < N) S.e_cnt[row] = 0;
    if (S.tap_ecnt && row < N) S.tap_ecnt[row] = rawcnt[rr];
    mycnt[rr] = (row < N && !has_verdict[rr]) ? rawcnt[rr] : 0u;
    s_rmatch[row] = -1;
    s_ecnt[row] = mycnt[rr];
    s_head[row] = SA_NONE;
    s_next[row] = (uint16_t)NONE16;
    s_rcount[row] = 0; s_ccount[row] = 0; s_lab[row] = (uint16_t)NONE16;
  }
#pragma unroll
  for (int cc = 0; cc < TC; ++cc) {   // this thread's columns
    const uint32_t j = q + (uint32_t)cc * SA_SMALL_N;
    s_v[j] = 0; s_cmatch[j] = -1; s_cwin[j] = SA_NONE;
    if (j < T) { S.cstamp[j] = 0u; S.cscan[j] = 0u; }   // (a search's labels: in HBM here — read again only behind the __syncthreads in front of the searches)
  }
#pragma unroll
  for (int k = 0; k < RC + TC; ++k) s_parent[q + (uint32_t)k * SA_SMALL_N] = q + (uint32_t)k * SA_SMALL_N;
  if (q < 8) s_ctr[q] = 0;
  {
    uint32_t tsum = 0;
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) tsum += mycnt[rr];
    for (int o = WAVE / 2; o > 0; o >>= 1) tsum += __shfl_xor(tsum, o);
    if (q % WAVE == 0) s_wsum[q / WAVE] = tsum;
  }
  sa_lds_barrier();
  if constexpr (WORDS) if (clsmode) {
    const uint32_t K = S.K;
    uint32_t mk = 0;
#pragma unroll
    for (uint32_t w2 = 0; w2 < SA_SMALL_N / WAVE; ++w2) mk = s_wmk[w2] > mk ? s_wmk[w2] : mk;
    const double max_dist = mk ? (double)sa_key_f32(mk) : -1.0;
    // W = c max_dist - sum, heaviest wins, lowest index among equals (k_assign_small's best_of); the words re-armed, the taps fed
    auto settle = [&](unsigned long long SA_G* words, unsigned long long SA_G* tap, uint32_t i, bool in) -> uint32_t {
      unsigned long long cls[SA_CLS_MAXK];
#pragma unroll
      for (uint32_t c = 0; c < SA_CLS_MAXK; ++c) cls[c] = words[(size_t)(in ? i : 0u) * K + (c < K ? c : K - 1u)];
      double bw = 0.0;
      uint32_t bi = SA_NONE;
#pragma unroll
      for (uint32_t c = 0; c < SA_CLS_MAXK; ++c) {
        const unsigned long long v = (c < K && in) ? cls[c] : ~0ull;
        if (tap && c < K && in) tap[(size_t)i * K + c] = v;
        if (v == ~0ull) continue;
        words[(size_t)i * K + c] = ~0ull;
        const double w = (double)(c + 1u) * max_dist - (double)sa_key_f32((uint32_t)(v >> 32));
        const uint32_t idx = (uint32_t)v;
        if (bi == SA_NONE || w > bw || (w == bw && idx < bi)) { bw = w; bi = idx; }
      }
      return bi;
    };
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      const uint32_t row = q + (uint32_t)rr * SA_SMALL_N;
      bt[rr] = settle(S.row_cls, S.tap_row_best, row, row < N);
      s_bt[row] = bt[rr];
    }
#pragma unroll
    for (int cc = 0; cc < TC; ++cc) {
      const uint32_t j = q + (uint32_t)cc * SA_SMALL_N;
      s_cq[j] = settle(S.col_cls, S.tap_col_best, j, j < T);
    }
    sa_lds_barrier();
  }
  if constexpr (WORDS) {
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      const uint32_t i = q + (uint32_t)rr * SA_SMALL_N;
      if (has_verdict[rr] && s_cq[bt[rr]] == i) vw0[rr] = (int32_t)bt[rr];  // the candidate that is best in its own best column wins it
    }
#pragma unroll
    for (int cc = 0; cc < TC; ++cc) { const uint32_t j = q + (uint32_t)cc * SA_SMALL_N; s_cexcl[j] = j < T && excluded(j); }
  }
  uint32_t total = 0;
  for (uint32_t w2 = 0; w2 < SA_SMALL_N / WAVE; ++w2) total += s_wsum[w2];
  if (total == 0) {  // nothing left for the positional vote
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      const uint32_t row = q + (uint32_t)rr * SA_SMALL_N;
      if (row < N) {
        uint64_t id = 0;
        uint8_t vt = SA_VOTE_NONE;
        const int32_t vw = vw0[rr];
        if (vw >= 0) { id = S.t_ids[vw]; vt = SA_VOTE_VISUAL; }
        SA_OUT(S.out_track_id + row, id);
        SA_OUT(S.out_vote + row, vt);
        S.win_col[row] = vw >= 0 ? vw : -1;
        SA_OUT(S.out_win + row, vw >= 0 ? vw : -1);
      }
    }
    sa_report_done(S, done_seq, &s_done);
    return;
  }
  TAIL_STAMP(1);
  if (VISUAL) {
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      const uint32_t row = q + (uint32_t)rr * SA_SMALL_N;
      const SaEdge SA_G* rp = S.e_edge + (row < N ? row : 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const SaEdge ed = (uint32_t)k < mycnt[rr] ? sa_ldg(rp + (size_t)k * N) : SaEdge{0, 0u, 0u};
        sj[rr][k] = ed.col;
        sg[rr][k] = ed.gain;
      }
    }
  }
  bool sx[RC][4];
  uint64_t sid[RC][4];
#pragma unroll
  for (int rr = 0; rr < RC; ++rr)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool in = (uint32_t)k < mycnt[rr];
      sx[rr][k] = VISUAL && in && excluded(sj[rr][k]);
      sid[rr][k] = in ? S.t_ids[sj[rr][k]] : 0ull;
    }
  uint32_t bcol[RC], usable[RC];
#pragma unroll
  for (int rr = 0; rr < RC; ++rr) {
    const uint32_t row = q + (uint32_t)rr * SA_SMALL_N;
    int64_t maxg = 0;
    bcol[rr] = SA_NONE;  // the column of the heaviest usable edge, lowest column on ties: this row's bid
    usable[rr] = 0;
    if (mycnt[rr]) {
      const SaEdge SA_G* rp = S.e_edge + row;
      for (uint32_t e0 = 0; e0 < mycnt[rr]; e0 += 4) {
        uint32_t jj[4];
        int64_t gg[4];
        bool skip[4];
        if (e0 == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { jj[k] = sj[rr][k]; gg[k] = sg[rr][k]; skip[k] = !((uint32_t)k < mycnt[rr]) || sx[rr][k]; }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool in = e0 + k < mycnt[rr];
            const SaEdge ed = in ? sa_ldg(rp + (size_t)(e0 + k) * N) : SaEdge{0, 0u, 0u};
            jj[k] = ed.col;
            gg[k] = ed.gain;
            skip[k] = !in;
          }
          if (VISUAL) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (!skip[k]) skip[k] = excluded(jj[k]);
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (skip[k]) continue;
          const uint32_t j = jj[k];
          ++usable[rr];
          if (gg[k] > maxg || (gg[k] == maxg && j < bcol[rr])) { maxg = gg[k]; bcol[rr] = j; }
          sa_uf_union(s_parent, row, N + j);
        }
      }
    }
    s_u[row] = -maxg;
    if (usable[rr]) atomicMin(&s_cwin[bcol[rr]], row);
  }
  sa_lds_barrier();
  TAIL_STAMP(2);
  uint32_t lab[RC];
#pragma unroll
  for (int rr = 0; rr < RC; ++rr) {
    const uint32_t row = q + (uint32_t)rr * SA_SMALL_N;
    lab[rr] = SA_NONE;
    if (usable[rr]) {
      lab[rr] = sa_uf_find(s_parent, row);
      s_lab[row] = (uint16_t)lab[rr];
      if (s_cwin[bcol[rr]] == row) { s_rmatch[row] = (int32_t)bcol[rr]; s_cmatch[bcol[rr]] = (int32_t)row; }
      else {
        s_next[row] = (uint16_t)atomicExch(&s_head[lab[rr]], row);
        atomicAdd(&s_rcount[lab[rr]], 1u);
      }
    }
  }
#pragma unroll
  for (int cc = 0; cc < TC; ++cc) {
    const uint32_t j = q + (uint32_t)cc * SA_SMALL_N;
    if (j < T) {
      const uint32_t r = sa_uf_find(s_parent, N + j);
      if (r < N) atomicAdd(&s_ccount[r], 1u);
    }
  }
  sa_lds_barrier();
  TAIL_STAMP(3);
#pragma unroll
  for (int rr = 0; rr < RC; ++rr) {
    const uint32_t row = q + (uint32_t)rr * SA_SMALL_N;
    if (lab[rr] == row && s_head[row] != SA_NONE) {
      // (no LDS pool here: a component with many search roots goes to the dense solver whatever its width)
      if (s_rcount[row] >= SA_DENSE_MIN_ROOTS) {
        s_queue[RCAP - 1u - atomicAdd(&s_ctr[4], 1u)] = (uint16_t)row;
        s_ccount[row] |= 0x80000000u;
      } else s_queue[atomicAdd(&s_ctr[0], 1u)] = (uint16_t)row;
    }
  }
  __syncthreads();   // (also: the zeroed labels in HBM have left this CU's waves before any search reads them)
  TAIL_STAMP(4);
  {
    const uint32_t lane = q % G;
    const uint32_t nq = s_ctr[0];
    sa_coop_ws w;
    w.e_cnt = s_ecnt;
    w.u = s_u; w.v = s_v; w.rmatch = s_rmatch; w.cmatch = s_cmatch;
    w.dist = (int64_t*)S.dist; w.pred = (int32_t*)S.pred; w.cstamp = (uint32_t*)S.cstamp; w.cscan = (uint32_t*)S.cscan;
    for (;;) {
      uint32_t take[1], seg[2];
      if (lane == 0) take[0] = atomicAdd(&s_ctr[1], 1u);
      const uint32_t k = sa_coop_bcast<G>(take);
      if (k >= nq) break;
      const uint32_t root = s_queue[k];
      const uint32_t R = s_rcount[root], C = s_ccount[root];
      if (lane == 0) { seg[0] = atomicAdd(&s_ctr[2], C); seg[1] = atomicAdd(&s_ctr[3], R); }
      const uint32_t cbase = sa_coop_bcast<G>(seg), rbase = sa_coop_bcast<G>(seg + 1);
      uint32_t* roots = s_rlist + rbase;
      if (R <= (uint32_t)G) {
        uint32_t cur = s_head[root], mine = SA_NONE;
        for (uint32_t st = 0; st < R; ++st) {
          if (st == lane) mine = cur;
          cur = s_next[cur & (RCAP - 1u)];
        }
        uint32_t rank = 0;
        for (uint32_t st = 0; st < R; ++st) {
          const uint32_t other = __shfl(mine, st, G);
          rank += other < mine ? 1u : 0u;
        }
        if (lane < R) roots[rank] = mine;
      } else {
        uint32_t cnt = 0;
        for (uint32_t r0 = 0; r0 < N; r0 += G) {
          const uint32_t row = r0 + lane;
          bool f[1];
          f[0] = row < N && s_lab[row] == (uint16_t)root && s_rmatch[row] < 0;
          uint32_t tot;
          const uint32_t rk = sa_coop_rank<G>(f, lane, &tot);
          if (f[0]) roots[cnt + rk] = row;
          cnt += tot;
        }
      }
      sa_coop_sync<G>();
      w.clist = (uint32_t*)S.cnext + cbase;
      // slot-major HBM lists: row r starts at record r, consecutive edges are N records apart
      w.e_col = (const uint32_t*)S.e_edge + 2; w.e_gain = (const int64_t*)S.e_edge; w.ecs = 4 * N; w.egs = 2 * N; w.rcs = 4; w.rgs = 2; w.e_off = nullptr; w.estride = 1;
      if constexpr (WORDS) w.excluded = s_cexcl;
      else w.excluded = VISUAL ? (const uint8_t*)S.col_excluded : nullptr;
      sa_assign_component_coop<G, sa_mem_wg<G>>(w, roots, R);
    }
  }
  TAIL_STAMP(5);
  sa_lds_barrier();  // rmatch is in LDS
  TAIL_STAMP(6);
  const uint32_t nd = s_ctr[4];
#pragma unroll
  for (int rr = 0; rr < RC; ++rr) {
    const uint32_t row = q + (uint32_t)rr * SA_SMALL_N;
    const bool mine_later = nd && usable[rr] && (s_ccount[lab[rr]] & 0x80000000u);
    if (row < N && !mine_later) {
      uint64_t id = 0;
      uint8_t vt = SA_VOTE_NONE;
      int32_t win = -1;
      const int32_t vw = vw0[rr];
      if (vw >= 0) { id = S.t_ids[vw]; vt = SA_VOTE_VISUAL; win = vw; }
      else if (!has_verdict[rr]) {
        const int32_t c = s_rmatch[row];
        if (c >= 0) {
          bool found = false;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if ((uint32_t)k < mycnt[rr] && sj[rr][k] == (uint32_t)c) { id = sid[rr][k]; found = true; }
          if (!found) id = S.t_ids[c];
          vt = SA_VOTE_POSITIONAL;
          win = c;
        }
      }
      SA_OUT(S.out_track_id + row, id);
      SA_OUT(S.out_vote + row, vt);
      S.win_col[row] = win;
      SA_OUT(S.out_win + row, win);
    }
  }
  if (nd) {
    if (q >= (uint32_t)DNT) { sa_report_done(S, done_seq, &s_done); return; }
    uint32_t rtop = s_ctr[3];
    for (uint32_t k = 0; k < nd; ++k) {
      const uint32_t root = s_queue[RCAP - 1u - k];
      const uint32_t R = s_rcount[root];
      uint32_t* roots = s_rlist + rtop;
      rtop += R;
      if (q < WAVE) {
        uint32_t cnt = 0;
        for (uint32_t r0 = 0; r0 < N; r0 += WAVE) {
          const uint32_t row = r0 + q;
          const bool f = row < N && s_lab[row] == (uint16_t)root && s_rmatch[row] < 0;
          const unsigned long long m = __ballot(f);
          if (f) roots[cnt + (uint32_t)__popcll(m & ((1ull << q) - 1ull))] = row;
          cnt += (uint32_t)__popcll(m);
        }
      }
      if (q == 0) s_ctr[5] = 0;
      sa_lds_barrier();
      uint32_t mg = 0;
      for (uint32_t row = q; row < N; row += (uint32_t)DNT) {
        if (s_lab[row] != (uint16_t)root) continue;
        const int64_t heaviest = -s_u[row];
        const uint32_t h32 = heaviest > 0x7fffffffll ? 0x7fffffffu : (uint32_t)heaviest;
        mg = h32 > mg ? h32 : mg;
        int64_t SA_G* drow = S.dense + (size_t)row * T;
        const uint32_t cnt = s_ecnt[row];
        const SaEdge SA_G* ep = S.e_edge + row;
        for (uint32_t e0 = 0; e0 < cnt; e0 += 4) {
          SaEdge ed[4];
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) ed[k2] = e0 + k2 < cnt ? sa_ldg(ep + (size_t)(e0 + k2) * N) : SaEdge{0, 0u, 0u};
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2)
            if (e0 + k2 < cnt && !(VISUAL && excluded(ed[k2].col))) drow[ed[k2].col] = ed[k2].gain;
        }
      }
      if (mg) atomicMax(&s_ctr[5], mg);
      __syncthreads();
      {
        sa_dense_ws w;
        w.gain = (const int64_t*)S.dense; w.ld = T; w.T = T;
        w.u = s_u; w.rmatch = s_rmatch; w.cmatch = s_cmatch; w.pred = (int32_t*)S.pred; w.part = s_part;
        bool k32 = false;
        if constexpr (TCAP <= SA_DENSE_K32_MAXT) k32 = s_ctr[5] <= (uint32_t)SA_DENSE_K32_MAXGAIN;
        if constexpr (TCAP <= SA_DENSE_K32_MAXT) { if (k32) sa_assign_component_dense<DNT, TCAP / DNT, true>(w, roots, R); }
        if (!k32) sa_assign_component_dense<DNT, TCAP / DNT, false>(w, roots, R);
      }
      for (uint32_t row = q; row < N; row += (uint32_t)DNT) {
        if (s_lab[row] != (uint16_t)root) continue;
        const int32_t c = s_rmatch[row];
        SA_OUT(S.out_track_id + row, c >= 0 ? S.t_ids[c] : 0ull);
        SA_OUT(S.out_vote + row, c >= 0 ? SA_VOTE_POSITIONAL : SA_VOTE_NONE);
        S.win_col[row] = c;
        SA_OUT(S.out_win + row, c);
        int64_t SA_G* drow = S.dense + (size_t)row * T;
        const uint32_t cnt = s_ecnt[row];
        const SaEdge SA_G* ep = S.e_edge + row;
        for (uint32_t e0 = 0; e0 < cnt; e0 += 4) {
          uint32_t cj[4];
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) cj[k2] = e0 + k2 < cnt ? sa_ldg(ep + (size_t)(e0 + k2) * N).col : 0u;
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2)
            if (e0 + k2 < cnt) drow[cj[k2]] = 0;
        }
      }
      __syncthreads();
    }
  }
  sa_report_done(S, done_seq, &s_done);
  TAIL_STAMP(7);
}
#undef SA_OUT

// General tail, kernel 1 of 2: every participating row finds its component (root = minimum vertex, always a row) and
// pushes itself onto that root's list — S.label[root] is the list head, S.next_row the links.  No O(N^2) scan for "the
// next row of my component"; the push order is arbitrary and is put right by the solver thread.
// The general tail's queue words live on their own 128-byte line of the scene's stats block: they are touched ONLY by agent-scope
// atomics while k_assign_solve runs (workgroups on different XCDs), and stats[0] next door is read and written with plain
// accesses by that kernel's first thread — a line held in one XCD's L2 by plain accesses and updated by other XCDs' atomics is
// not something to rely on.
// The label step (k_assign_label's body).
// Thread q handles ROW q (q < N) and the COLUMNS q, q + cstride, ... (< T).
// words: the visual vote arrives as vote words (one per candidate and per track, or one per count class of each: SCN_WORDSK) instead of
// k_bestfit_resolve's verdict arrays; they are turned into those arrays here for the solver (row_has / vis_winner for every candidate,
// col_excluded for every track: all of them written, none needs a reset) exactly the way k_assign_small<.., WORDS> decides: candidate q
// wins its best column iff that column's word names q; column j is excluded iff the best column of the candidate its word names is j.
// One dependent load each.  The words are re-armed by the solver, after every reader.
// s_mk: nthreads / 64 words of LDS (class words: the frame's max_dist is folded by the whole workgroup — every thread of it calls this).
__device__ __forceinline__ void sa_label_phase(const SceneDev& S, bool words, uint32_t q, uint32_t cstride, uint32_t* s_mk, uint32_t nthreads) {
  bool has_verdict = false;
  const uint32_t N = S.N, T = S.T, K = S.K;
  if (words) {
    if (S.flags & SCN_WORDSK) {
      // class words: W = c max_dist - sum decides between a row's (column's) class winners; max_dist from the first phase's per-tile slots
      uint32_t mk = 0;
      for (uint32_t i = threadIdx.x; i < S.nkeys; i += nthreads) {
        const uint32_t v = S.vis_max_key[i];
        mk = v > mk ? v : mk;
      }
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t ok = __shfl_xor(mk, o);
        mk = ok > mk ? ok : mk;
      }
      if ((threadIdx.x & 63u) == 0) s_mk[threadIdx.x >> 6] = mk;
      __syncthreads();
      mk = 0;
      for (uint32_t w2 = 0; w2 < nthreads / 64u; ++w2) mk = s_mk[w2] > mk ? s_mk[w2] : mk;
      const double max_dist = mk ? (double)sa_key_f32(mk) : -1.0;
      auto best_of = [&](const unsigned long long SA_G* cls, bool in, bool* any) -> uint32_t {
        unsigned long long w[SA_CLS_MAXK];
#pragma unroll
        for (uint32_t c = 0; c < SA_CLS_MAXK; ++c) w[c] = cls[c < K ? c : K - 1u];  // all loads together (see k_assign_small)
        double bw = 0.0;
        uint32_t bi = SA_NONE;
#pragma unroll
        for (uint32_t c = 0; c < SA_CLS_MAXK; ++c) {
          if (!in || c >= K || w[c] == ~0ull) continue;
          const double wt = (double)(c + 1u) * max_dist - (double)sa_key_f32((uint32_t)(w[c] >> 32));
          const uint32_t i = (uint32_t)w[c];
          if (bi == SA_NONE || wt > bw || (wt == bw && i < bi)) { bw = wt; bi = i; }
        }
        if (any) *any = bi != SA_NONE;
        return bi;
      };
      if (S.tap_row_best) {  // SA_FLAG_TAP: the class words as the first phase left them ([N K] then [T K])
        for (uint32_t c = 0; c < K; ++c) {
          if (q < N) S.tap_row_best[(size_t)q * K + c] = S.row_cls[(size_t)q * K + c];
          for (uint32_t j = q; j < T; j += cstride) S.tap_col_best[(size_t)j * K + c] = S.col_cls[(size_t)j * K + c];
        }
      }
      const uint32_t bt = best_of(S.row_cls + (size_t)(q < N ? q : 0u) * K, q < N, &has_verdict);
      // (second round trip: the column my row prefers; below, for my columns: the row each of them prefers)
      const uint32_t bt_cq = best_of(S.col_cls + (size_t)(bt != SA_NONE ? bt : 0u) * K, bt != SA_NONE, nullptr);
      if (q < N) {
        S.row_has[q] = has_verdict ? 1 : 0;
        S.vis_winner[q] = (has_verdict && bt_cq == q) ? (int32_t)bt : -1;
      }
      for (uint32_t j = q; j < T; j += cstride) {
        const uint32_t cq = best_of(S.col_cls + (size_t)j * K, true, nullptr);
        const uint32_t cq_bt = best_of(S.row_cls + (size_t)(cq != SA_NONE ? cq : 0u) * K, cq != SA_NONE, nullptr);
        S.col_excluded[j] = (cq != SA_NONE && cq_bt == j) ? 1 : 0;
      }
    } else {
      const unsigned long long rb = q < N ? S.row_best[q] : ~0ull;
      const unsigned long long cb0 = q < T ? S.col_best[q] : ~0ull;   // (this thread's first column, requested beside the row word)
      if (S.tap_row_best && q < N) S.tap_row_best[q] = rb;  // SA_FLAG_TAP: the words as the first phase left them
      const uint32_t bt = rb != ~0ull ? (uint32_t)rb : SA_NONE;
      has_verdict = bt != SA_NONE;
      const unsigned long long cb_bt = bt != SA_NONE ? S.col_best[bt] : ~0ull;  // (bt < T, cq < N: written by this frame's tiles)
      if (q < N) {
        S.row_has[q] = has_verdict ? 1 : 0;
        S.vis_winner[q] = (has_verdict && cb_bt != ~0ull && (uint32_t)cb_bt == q) ? (int32_t)bt : -1;
      }
      for (uint32_t j = q; j < T; j += cstride) {
        const unsigned long long cb = j == q ? cb0 : S.col_best[j];
        if (S.tap_row_best) S.tap_col_best[j] = cb;
        const uint32_t cq = cb != ~0ull ? (uint32_t)cb : SA_NONE;
        const unsigned long long rb_cq = cq != SA_NONE ? S.row_best[cq] : ~0ull;
        S.col_excluded[j] = (cq != SA_NONE && rb_cq != ~0ull && (uint32_t)rb_cq == j) ? 1 : 0;
      }
    }
  }
  if (q >= N) return;
  // move this row's edge count and dual to the solver's copies and leave the accumulators clean for the next frame
  const uint32_t cnt = S.e_cnt[q];
  S.e_use[q] = cnt;
  S.e_cnt[q] = 0;
  if (S.tap_ecnt) S.tap_ecnt[q] = cnt;  // SA_FLAG_TAP
  S.u_use[q] = S.u[q];
  S.u[q] = 0;
  if (!cnt || (words ? has_verdict : S.row_has[q] != 0)) { S.lab[q] = SA_NONE; return; }
  const uint32_t root = sa_uf_find((uint32_t*)S.parent, q);
  S.lab[q] = root;
  const uint32_t pos = atomicAdd((uint32_t*)(S.rnext + root), 1u);  // rows in the component (rnext is zeroed by the preparation blocks)
  if (pos < SA_CROW) S.crow[(size_t)root * SA_CROW + pos] = make_uint2(q, cnt);   // (k_assign_solve's middle tier: the component from one load)
  S.next_row[q] = atomicExch((uint32_t*)(S.label + root), q);
}
template <bool WORDS>
__global__ __launch_bounds__(256) void k_assign_label(const SceneDev* __restrict__ scenes) {
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  __shared__ uint32_t s_mk[4];
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q == 0) { S.stats[SA_QW_TOP] = 0u; S.stats[SA_QW_LEN] = 0u; S.stats[SA_QW_TICKET] = 0u; S.stats[SA_QW_DONE] = 0u; S.stats[SA_QW_MLEN] = 0u; S.stats[SA_QW_MTICKET] = 0u; }  // the solver's queues: top of its row lists | big queue: length, next ticket | row workgroups done | mid-sized queue: length (SA_QW_MTICKET: unused since the entries are shared out statically)
  if (q < S.N) ((uint32_t SA_G*)S.dq)[S.N + q] = SA_NONE;  // the queue of mid-sized components: an entry that is not SA_NONE has LANDED (k_assign_solve serves it at once)
  sa_label_phase(S, WORDS, q, gridDim.x * blockDim.x, s_mk, 256u);
}

// Kernel 2 of 2: thread `root` owns the component rooted at row `root`: orders its rows ascending (components are a handful of
// rows in tracking workloads; the order only has to be deterministic), solves it, and writes the results of all its rows.  Rows
// that take no part write their own result.
// The shortest-path search is a chain of dependent reads and writes of per-row / per-column state; in HBM every link of that
// chain is an L2 (or, for what other XCDs just wrote, a memory) round trip — ~10 of them even for a one-row component, 14 us
// at C4.  A component that fits a small private block of LDS (SL_R rows, SL_C distinct columns, SL_E usable edges) is therefore
// gathered once — rows sorted, columns renumbered in ascending order so that every index comparison the solver makes keeps
// its outcome — solved there by the same sa_assign_component, and scattered back; larger components use the HBM work set.
#define SL_R 8
#define SL_C 12
#define SL_E 24
struct SolveLocal {
  int64_t u[SL_R], rdist[SL_R], v[SL_C], dist[SL_C], e_gain[SL_E];
  uint32_t e_cnt[SL_R], e_off[SL_R], next_row[SL_R], rows[SL_R];
  int32_t rmatch[SL_R], rnext[SL_R], cmatch[SL_C], pred[SL_C], cnext[SL_C];
  uint32_t cstamp[SL_C], cscan[SL_C], colmap[SL_C], e_col[SL_E];
  uint32_t bank_pad[2];  // 1088 -> 1096 bytes: lane i's block starts 274 i words into LDS — 32 distinct banks for a wave's 64 lanes instead
                         // of 4 (272 i: every access of the gather, the solve and the scatter was a 16-way bank conflict)
};
static_assert(sizeof(SolveLocal) == 1096, "SolveLocal: one lane's block must not start a multiple of 16 words after its neighbour's");
template <bool VISUAL>
__device__ __forceinline__ void finalize_row_with(const SceneDev& S, uint32_t q, int32_t c) {
  uint64_t id = 0;
  uint8_t vt = SA_VOTE_NONE;
  int32_t win = -1;
  const int32_t vw = VISUAL ? S.vis_winner[q] : -1;
  if (vw >= 0) { id = S.t_ids[vw]; vt = SA_VOTE_VISUAL; win = vw; }
  else if (c >= 0 && !(VISUAL && S.row_has[q])) { id = S.t_ids[c]; vt = SA_VOTE_POSITIONAL; win = c; }
  S.out_track_id[q] = id;
  S.out_vote[q] = vt;
  S.win_col[q] = win;
  S.out_win[q] = win;
}

// The general tail's MIDDLE tier: a component of up to ML_R rows (a knot of a crowd) is solved by ONE wavefront on a private block
// of LDS.  The dense solver of the big tier reads a row of T gains from HBM and crosses a workgroup barrier per search step (~0.9 us
// at 2000 tracks) however few columns the component has; here the component's distinct columns are renumbered in ascending track
// order (a hash table in LDS finds them, a rank pass orders them — every index comparison of the solver keeps its outcome), its
// gains become a [rows][128] (or, for at most 32 rows, [rows][256]) matrix of 32-bit cells in LDS and sa_assign_component_dense runs
// with NT = 64 and 2 (4) columns per lane: a search step is an LDS read, a handful of VALU operations and a wave minimum.
// Refused (-> the workgroup finishes it with the big tier's solver): more distinct usable columns than that, or a gain beyond the
// 32-bit variant's bound.
#define ML_R 64
#define ML_WAVES 1u
#define ML_C 256
#define ML_CELLS 8192
#define ML_H 512
struct MidLocal {
  int32_t gain[ML_CELLS];
  int64_t u[ML_R];
  int32_t rmatch[ML_R], cmatch[ML_C], pred[ML_C];  // (pred doubles as the bid words of the greedy start)
  uint32_t rows[ML_R], cols[ML_C], hkey[ML_H], hval[ML_H], roots[ML_R];
  uint32_t ckey[ML_C], cslot[ML_C];  // the occupied slots of the hash table, compacted: key | slot
  uint32_t ncols, fail;
};
static_assert(sizeof(MidLocal) * ML_WAVES <= sizeof(SolveLocal) * 40, "the wavefronts' blocks lie over the pool of the small tier");
__device__ __forceinline__ void sa_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// A big component's root onto the scene's queue.  The entry must have LANDED before this workgroup reports its rows done: a plain
// (even atomic) store may still be in flight when the workgroup's barrier lets thread 0 signal — a returning exchange has been
// performed at the coherence point when its result arrives, and using the result makes the wave wait for it.
// A mid-sized entry carries the component's row count with the root ((R << 24) | root, R <= ML_R): its taker needs no second trip for it.
__device__ __forceinline__ void sa_queue_push(const SceneDev& S, uint32_t root, bool mid, uint32_t R) {
  const uint32_t at = atomicAdd((uint32_t*)(S.stats + (mid ? SA_QW_MLEN : SA_QW_LEN)), 1u) + (mid ? S.N : 0u);  // (dq: [N] big | [N] mid-sized)
  const uint32_t old = __hip_atomic_exchange((uint32_t*)S.dq + at, mid ? ((R << 24) | root) : root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::"v"(old));
}
template <bool VISUAL>
__device__ __forceinline__ bool mid_solve_component(const SceneDev& S, uint32_t root, uint32_t R, MidLocal& M) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t N = S.N;
  MID_STAMP(0);
  // rows, ascending: the labels from the root on (it is the component's lowest row), sixteen loads in flight per lane (a frame of up to
  // 1024 detections: one trip)
  // ... and their usable-edge counts in the SAME trip (a second one otherwise: what the label kernel wrote lies in memory, ~1.5 us away).
  // A component of at most SA_CROW rows: from the list the label kernel left at its root — one load per lane instead of 32; the rows
  // arrive in any order, a rank by counting puts them in ascending order.
  uint32_t cnt = 0;
  const bool listed = R <= SA_CROW;   // (wave-uniform: R comes with the queue entry)
  if (listed) {
    const uint2 rc = lane < R ? S.crow[(size_t)root * SA_CROW + lane] : make_uint2(SA_NONE, 0u);
    uint32_t rank = 0;
    for (uint32_t j = 0; j < R; ++j) rank += (uint32_t)__shfl((int)rc.x, (int)j) < rc.x ? 1u : 0u;
    if (lane < R) { M.rows[rank] = rc.x; M.roots[rank] = rc.y; }
    cnt = R;
  } else {
    for (uint32_t r0 = root; r0 < N && cnt < R; r0 += 1024) {
      uint32_t lb8[16], ne8[16];
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        const uint32_t row = r0 + (uint32_t)k2 * 64u + lane;
        lb8[k2] = row < N ? S.lab[row] : SA_NONE;
        ne8[k2] = row < N ? S.e_use[row] : 0u;
      }
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        const uint32_t row = r0 + (uint32_t)k2 * 64u + lane;
        const bool f = row < N && lb8[k2] == root;
        const unsigned long long m = __ballot(f);
        const uint32_t at = cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (f && at < ML_R) { M.rows[at] = row; M.roots[at] = ne8[k2]; }   // (roots: free until the greedy start)
        cnt += (uint32_t)__popcll(m);
      }
    }
  }
  MID_STAMP(1);
  if (cnt != R || R > ML_R) return false;  // (labels and count disagree: not this tier's to sort out — wave-uniform, both from ballots / SGPRs)
#pragma unroll
  for (int h = 0; h < ML_H / 64; ++h) M.hkey[lane + 64u * (uint32_t)h] = SA_NONE;
#pragma unroll
  for (int h = 0; h < ML_C / 64; ++h) { M.cmatch[lane + 64u * (uint32_t)h] = -1; M.pred[lane + 64u * (uint32_t)h] = (int32_t)0x7fffffff; }
  if (lane == 0) { M.ncols = 0; M.fail = 0; }
  sa_wave_sync();
  const uint32_t row = lane < R ? M.rows[lane] : 0u;
  const uint32_t ne = lane < R ? M.roots[lane] : 0u;
  const SaEdge SA_G* ep = S.e_edge + (size_t)row * S.estride;
  // every edge of the component: offsets by a wave scan of the rows' counts
  uint32_t incl = ne;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= (uint32_t)o) incl += up;
  }
  const uint32_t eoff = incl - ne, total = (uint32_t)__shfl((int)incl, 63);
  const bool fast = listed && total <= 64u;   // (wave-uniform)
  uint32_t C = 0, ldc = 128u;
  int32_t maxg = 0, bj = -1;
  if (fast) {
    // A knot of a crowd — a handful of rows, a few dozen edges, what the tracker loop's crowd frames consist of: ONE EDGE PER LANE.  The
    // records come in one trip (not a serial walk per row, and not twice), the distinct columns are ranked by two passes over the
    // (at most 64) column words in LDS instead of a hash table + compaction + rank + second walk, the matrix is [R][64].  (In-kernel
    // timeline of the hash path on such knots: columns hashed 2.1 us, ranked 1.4, matrix 1.1 — for a 0.5 us search.)
    uint32_t* const erow = M.ckey; uint32_t* const ek = M.cslot;
    uint32_t* const ecol = M.hkey; uint32_t* const ecid = M.hkey + 64; uint32_t* const efirst = M.hkey + 128;
    int32_t* const egain = (int32_t*)M.hval;
    for (uint32_t k = 0; k < ne; ++k) { erow[eoff + k] = lane; ek[eoff + k] = k; }
    sa_wave_sync();
    const bool valid = lane < total;
    const uint32_t er = valid ? erow[lane] : 0u;
    const SaEdge ed = valid ? sa_ldg(S.e_edge + (size_t)M.rows[er] * S.estride + ek[lane]) : SaEdge{0, 0u, 0u};
    const bool use = valid && !(VISUAL && S.col_excluded[ed.col]);
    if (__ballot(use && ed.gain > (int64_t)SA_DENSE_K32_MAXGAIN)) return false;
    const uint32_t key = use ? ed.col : SA_NONE;
    ecol[lane] = key;
    sa_wave_sync();
    // (four column words per LDS read; a lane beyond the last edge holds SA_NONE, which equals no key and lies below none)
    const uint32_t quads = (total + 3u) / 4u;
    bool first = use;
    for (uint32_t x = 0; x < quads; ++x) {
      const uint4 o = ((const uint4*)ecol)[x];
      const uint32_t j = 4u * x;
      first = first && !((j < lane && o.x == key) || (j + 1u < lane && o.y == key) || (j + 2u < lane && o.z == key) || (j + 3u < lane && o.w == key));
    }
    efirst[lane] = first ? key : SA_NONE;   // the DISTINCT columns: a repeated one leaves SA_NONE
    sa_wave_sync();
    uint32_t rank = 0;
    for (uint32_t x = 0; x < quads; ++x) {
      const uint4 o = ((const uint4*)efirst)[x];
      rank += (o.x < key) + (o.y < key) + (o.z < key) + (o.w < key);
    }
    C = (uint32_t)__popcll(__ballot(first));
    MID_STAMP(2);
    MID_NOTE(7, ((unsigned long long)R << 32) | C);
    ldc = 64u;
    if (first) M.cols[rank] = key;
    for (uint32_t i = lane; i < R * 16u; i += 64) ((uint4*)M.gain)[i] = make_uint4(0u, 0u, 0u, 0u);
    egain[lane] = use ? (int32_t)ed.gain : 0;
    ecid[lane] = rank;
    sa_wave_sync();
    MID_STAMP(3);
    if (use) M.gain[er * 64u + rank] = (int32_t)ed.gain;
    for (uint32_t k = 0; k < ne; ++k) {
      const int32_t g = egain[eoff + k], j = (int32_t)ecid[eoff + k];
      if (g > 0 && (g > maxg || (g == maxg && j < bj))) { maxg = g; bj = j; }
    }
    sa_wave_sync();
  } else {
    // pass 1 over the edges (lane = row): the distinct usable columns into the hash table
    for (uint32_t e0 = 0; e0 < ne; e0 += 4) {
      SaEdge ed[4];
      bool use[4];
  #pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) ed[k2] = e0 + k2 < ne ? sa_ldg(ep + e0 + k2) : SaEdge{0, 0u, 0u};
  #pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) use[k2] = e0 + k2 < ne && !(VISUAL && S.col_excluded[ed[k2].col]);
  #pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        if (!use[k2]) continue;
        if (ed[k2].gain > (int64_t)SA_DENSE_K32_MAXGAIN) { M.fail = 1; continue; }
        const uint32_t col = ed[k2].col;
        uint32_t slot = (col * 2654435761u) >> 23;
        bool placed = false;
        for (uint32_t probe = 0; probe < ML_H; ++probe) {
          const uint32_t old = atomicCAS(&M.hkey[slot], SA_NONE, col);
          if (old == SA_NONE) { if (atomicAdd(&M.ncols, 1u) >= ML_C) M.fail = 1; placed = true; break; }
          if (old == col) { placed = true; break; }
          slot = (slot + 1u) & (ML_H - 1u);
        }
        if (!placed) M.fail = 1;
      }
      if (*(volatile uint32_t*)&M.fail) break;
    }
    sa_wave_sync();
    if (__builtin_amdgcn_readfirstlane((int)M.fail)) return false;  // (one word, every lane reads the same: uniform for the compiler too)
    C = (uint32_t)__builtin_amdgcn_readfirstlane((int)M.ncols);
    MID_STAMP(2);
    MID_NOTE(7, ((unsigned long long)R << 32) | C);
    ldc = C <= 128u ? 128u : 256u;   // the matrix: [R][128], or [R][256] when the rows allow it
    if (R * ldc > ML_CELLS) return false;
    // the columns in ascending order of their track index: the occupied slots compacted, every key ranked among them
    {
      uint32_t base = 0;
  #pragma unroll
      for (int h = 0; h < ML_H / 64; ++h) {
        const uint32_t sl = lane + 64u * (uint32_t)h;
        const uint32_t key = M.hkey[sl];
        const unsigned long long m = __ballot(key != SA_NONE);
        if (key != SA_NONE) {
          const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          M.ckey[at] = key;
          M.cslot[at] = sl;
        }
        base += (uint32_t)__popcll(m);
      }
    }
    for (uint32_t i = lane; i < R * (ldc / 4); i += 64) ((uint4*)M.gain)[i] = make_uint4(0u, 0u, 0u, 0u);
    for (uint32_t i = lane; i < ML_C; i += 64)
      if (i >= C) M.ckey[i] = SA_NONE;   // (padding of the rank loop's vector reads: the largest word, never below a key)
    sa_wave_sync();
    for (uint32_t i = lane; i < C; i += 64) {
      const uint32_t key = M.ckey[i];
      uint32_t rank = 0;
      for (uint32_t x = 0; x < C; x += 4) {
        const uint4 o = *(const uint4*)&M.ckey[x];
        rank += (o.x < key) + (o.y < key) + (o.z < key) + (o.w < key);
      }
      M.hval[M.cslot[i]] = rank;
      M.cols[rank] = key;
    }
    sa_wave_sync();
    MID_STAMP(3);
    // pass 2: gains into the matrix, the row's dual and its bid (heaviest usable edge, lowest column on ties)
    for (uint32_t e0 = 0; e0 < ne; e0 += 4) {
      SaEdge ed[4];
      bool use[4];
  #pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) ed[k2] = e0 + k2 < ne ? sa_ldg(ep + e0 + k2) : SaEdge{0, 0u, 0u};
  #pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) use[k2] = e0 + k2 < ne && !(VISUAL && S.col_excluded[ed[k2].col]);
  #pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        if (!use[k2]) continue;
        const uint32_t col = ed[k2].col;
        uint32_t slot = (col * 2654435761u) >> 23;
        while (M.hkey[slot] != col) slot = (slot + 1u) & (ML_H - 1u);  // (present: pass 1 placed it)
        const int32_t j = (int32_t)M.hval[slot];
        const int32_t g = (int32_t)ed[k2].gain;
        M.gain[lane * ldc + (uint32_t)j] = g;
        if (g > maxg || (g == maxg && j < bj)) { maxg = g; bj = j; }
      }
    }
  }
  if (lane < R) {
    M.u[lane] = -(int64_t)maxg;
    M.rmatch[lane] = -1;
    if (bj >= 0) atomicMin((uint32_t*)&M.pred[bj], lane);
  }
  sa_wave_sync();
  // greedy start: the lowest row bidding for a column has it; the others are the search roots, ascending
  bool pend = false;
  if (lane < R && bj >= 0) {
    if ((uint32_t)M.pred[bj] == lane) { M.rmatch[lane] = bj; M.cmatch[bj] = (int32_t)lane; }
    else pend = true;
  }
  const unsigned long long pm = __ballot(pend);
  if (pend) M.roots[__popcll(pm & ((1ull << lane) - 1ull))] = lane;
  const uint32_t n_roots = (uint32_t)__popcll(pm);
  MID_STAMP(4);
  // the ids of the component's columns, requested now: they arrive while the search runs
  uint64_t cid_r[ML_C / 64];
#pragma unroll
  for (int h = 0; h < ML_C / 64; ++h) {
    const uint32_t j = lane + 64u * (uint32_t)h;
    cid_r[h] = j < C ? S.t_ids[M.cols[j]] : 0ull;
  }
  sa_wave_sync();
  if (n_roots) {
    sa_dense_ws w;
    w.gain = (const int64_t*)M.gain; w.ld = ldc; w.T = ldc;
    w.u = M.u; w.rmatch = M.rmatch; w.cmatch = M.cmatch; w.pred = M.pred; w.part = nullptr;
    if (ldc == 64u) sa_assign_component_dense<64, 1, true, true>(w, M.roots, n_roots);
    else if (ldc == 128u) sa_assign_component_dense<64, 2, true, true>(w, M.roots, n_roots);
    else sa_assign_component_dense<64, 4, true, true>(w, M.roots, n_roots);
  }
  MID_STAMP(5);
  // results: a row of a component carries no visual verdict (k_assign_label leaves those out of the lists), so its answer is its column
  // — whose id was requested before the search (cid_r) and went through LDS, over the hash table that pass 2 was the last to read
  uint64_t* const cid = (uint64_t*)M.hkey;   // [ML_C] ids over hkey | hval (2 x ML_H words)
  static_assert(ML_C * 8 <= 2 * ML_H * 4, "the columns' ids lie over the hash table");
#pragma unroll
  for (int h = 0; h < ML_C / 64; ++h)
    if (lane + 64u * (uint32_t)h < C) cid[lane + 64u * (uint32_t)h] = cid_r[h];
  sa_wave_sync();
  if (lane < R) {
    const int32_t c = M.rmatch[lane];
    const int32_t win = c >= 0 ? (int32_t)M.cols[c] : -1;
    S.out_track_id[row] = c >= 0 ? cid[c] : 0ull;
    S.out_vote[row] = c >= 0 ? SA_VOTE_POSITIONAL : SA_VOTE_NONE;
    S.win_col[row] = win;
    S.out_win[row] = win;
  }
  sa_wave_sync();
  MID_STAMP(6);
  return true;
}
// One big component by the dense solver of sa_dense.h, all NT threads of the workgroup on it.  rows ascending (ballot compaction of
// the scene's row labels) -> greedy start: every row bids for the column of its heaviest usable edge (global atomic minimum on cwin,
// SA_NONE between frames), its gains go into the dense matrix, u = -(heaviest gain) -> rows that lost their bid are the search roots
// (ascending, compacted in place) -> sa_assign_component_dense -> results, matrix and bids wiped.
template <bool VISUAL, int NT, int CPT, bool LDS_STATE>
__device__ __forceinline__ void dense_solve_component(const SceneDev& S, uint32_t root, int64_t* u, int32_t* rmatch, int32_t* cmatch, int32_t* pred,
                                                      unsigned long long* s_part, uint32_t* s_word) {
  const uint32_t N = S.N, T = S.T;
  const uint32_t q = threadIdx.x, lane = q & 63u;
  const uint8_t SA_G* excl = VISUAL ? S.col_excluded : nullptr;
  const uint32_t R = (uint32_t)S.rnext[root];  // rows of the component (k_assign_label)
  if (q == 0) { s_word[0] = atomicAdd((uint32_t*)(S.stats + SA_QW_TOP), R); s_word[2] = 0; }  // its segment of the row lists | heaviest gain
  __syncthreads();
  uint32_t* rows = (uint32_t*)S.big_rows + s_word[0];   // the component's rows, then (in place of the matched ones) its search roots
  if (q < 64) {  // eight label loads in flight per lane: the scan is a chain of L2 round trips otherwise
    uint32_t cnt = 0;
    for (uint32_t r0 = 0; r0 < N; r0 += 512) {
      uint32_t lb8[8];
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        const uint32_t row = r0 + (uint32_t)k2 * 64u + lane;
        lb8[k2] = row < N ? S.lab[row] : SA_NONE;
      }
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        const uint32_t row = r0 + (uint32_t)k2 * 64u + lane;
        const bool f = row < N && lb8[k2] == root;
        const unsigned long long m = __ballot(f);
        if (f) rows[cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = row;
        cnt += (uint32_t)__popcll(m);
      }
    }
  }
  __syncthreads();
  // greedy start: bids, duals, gains into the dense matrix
  for (uint32_t i = q; i < R; i += NT) {
    const uint32_t row = rows[i];
    const uint32_t ne = S.e_use[row];
    const SaEdge SA_G* ep = S.e_edge + (size_t)row * S.estride;
    int64_t SA_G* drow = S.dense + (size_t)row * T;
    int64_t maxg = 0;
    uint32_t bcol = SA_NONE;
    for (uint32_t e0 = 0; e0 < ne; e0 += 4) {  // four records (and their exclusion flags) per round trip
      SaEdge ed[4];
      bool use[4];
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) ed[k2] = e0 + k2 < ne ? sa_ldg(ep + e0 + k2) : SaEdge{0, 0u, 0u};
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) use[k2] = e0 + k2 < ne && !(excl && excl[ed[k2].col]);
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        if (!use[k2]) continue;
        drow[ed[k2].col] = ed[k2].gain;
        if (ed[k2].gain > maxg || (ed[k2].gain == maxg && ed[k2].col < bcol)) { maxg = ed[k2].gain; bcol = ed[k2].col; }
      }
    }
    u[row] = -maxg;
    S.big_bcol[row] = bcol;
    if (bcol != SA_NONE) atomicMin((uint32_t*)(S.cwin + bcol), row);
    if (maxg > 0) atomicMax(&s_word[2], maxg > 0x7fffffffll ? 0x7fffffffu : (uint32_t)maxg);
  }
  __syncthreads();
  if (q < 64) {  // matched rows keep their bid; the others become the search roots, ascending, compacted in place
    uint32_t nroots = 0;
    for (uint32_t i0 = 0; i0 < R; i0 += 64) {
      const uint32_t i = i0 + lane;
      bool pend = false;
      uint32_t row = 0;
      if (i < R) {
        row = rows[i];
        const uint32_t bc = S.big_bcol[row];
        if (bc != SA_NONE) {
          if (__hip_atomic_load((uint32_t*)(S.cwin + bc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == row) { rmatch[row] = (int32_t)bc; cmatch[bc] = (int32_t)row; }
          else pend = true;
        }
      }
      const unsigned long long m = __ballot(pend);
      // (position nroots + rank <= i: a root never overwrites an entry that has not been read yet; all lanes of this step have read theirs)
      if (pend) rows[nroots + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = row;
      nroots += (uint32_t)__popcll(m);
    }
    if (lane == 0) s_word[1] = nroots;
  }
  __syncthreads();
  {
    sa_dense_ws w;
    w.gain = (const int64_t*)S.dense; w.ld = T; w.T = T;
    w.u = u; w.rmatch = rmatch; w.cmatch = cmatch; w.pred = pred; w.part = s_part;
    bool k32 = false;
    if constexpr ((uint32_t)NT * CPT <= SA_DENSE_K32_MAXT) k32 = s_word[2] <= (uint32_t)SA_DENSE_K32_MAXGAIN;
    if constexpr ((uint32_t)NT * CPT <= SA_DENSE_K32_MAXT) {
      if (k32) sa_assign_component_dense<NT, CPT, true>(w, rows, s_word[1]);
    }
    if (!k32) sa_assign_component_dense<NT, CPT, false>(w, rows, s_word[1]);
  }
  // results; matrix, bids and (LDS) matches wiped for the next component / frame.  The rows list was overwritten by the roots:
  // walk the scene's labels again.
  for (uint32_t row = q; row < N; row += NT) {
    if (S.lab[row] != root) continue;
    const int32_t c = rmatch[row];
    finalize_row_with<VISUAL>(S, row, c);
    const uint32_t ne = S.e_use[row];
    const SaEdge SA_G* ep = S.e_edge + (size_t)row * S.estride;
    int64_t SA_G* drow = S.dense + (size_t)row * T;
    for (uint32_t e0 = 0; e0 < ne; e0 += 4) {
      uint32_t cj[4];
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) cj[k2] = e0 + k2 < ne ? sa_ldg(ep + e0 + k2).col : 0u;
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2)
        if (e0 + k2 < ne) drow[cj[k2]] = 0;
    }
    const uint32_t bc = S.big_bcol[row];
    if (bc != SA_NONE) S.cwin[bc] = SA_NONE;
    if (LDS_STATE) {
      if (c >= 0) cmatch[c] = -1;
      rmatch[row] = -1;
    }
  }
  __syncthreads();
}

// General tail, kernel 2 of 2.  NT threads per workgroup, thread = row.  The thread of a component's root (= its lowest row) owns it:
//   * ONE row (most components of a tracking frame): its heaviest usable edge (lowest column on ties) — what the shortest-path
//     search does for the first row of a component, without any search state — from records requested at the top of the kernel;
//   * TWO rows with at most four records each: from registers too, three round trips (see the pair path below);
//   * up to ML_R rows and ML_C distinct columns: onto the scene's queue of mid-sized components, one WAVEFRONT each
//     (mid_solve_component);
//   * anything larger: onto the scene's queue of big components, one WORKGROUP each (dense_solve_component).
//   (Mahalanobis engines, whose gains do not fit the middle tier's 32-bit cells: components of up to 8 rows, 12 columns and 24
//   usable edges are gathered by the root's lane into a private block of LDS — a pool of SL_POOL per workgroup —, solved there by
//   the serial sa_assign_component and scattered back; the rest goes to the big queue.)
//   The helper workgroups launched behind the scene's row workgroups serve the queues: helper b the mid-sized entries b, b + G, ... as
//   they land, then, once every row workgroup has said it is through with its rows, the big ones by ticket.  No third launch, and a
//   crowd's dozens of knots are solved side by side.  (Row workgroups never wait: forward progress does not depend on residency.)
// Per-row duals / matches and per-column matches / predecessors of the dense solver: dynamic LDS when 12 N + 8 T bytes fit
// (LDS_STATE), else the scene's arrays in HBM — the workgroup's own L1 keeps them coherent between its waves.
#define SL_POOL 40
template <bool VISUAL, int NT, int CPT, bool LDS_STATE>
__global__ __launch_bounds__(NT) void k_assign_solve(const SceneDev* __restrict__ scenes, uint32_t row_wgs_) {
  const uint32_t row_wgs = row_wgs_ & 0x0fffffffu;
  const bool no_mid = (row_wgs_ >> 31) != 0;  // (Mahalanobis gains are beyond the middle tier's 32-bit cells)
  const bool rearm_words = ((row_wgs_ >> 30) & 1u) != 0;  // the frame's visual vote came as vote words: re-armed here
  // One lane gathering a component into its pool block is a chain of dependent trips to L2 / memory — the list walk, then every
  // row's count and records: ~12 us for two rows, ~60 us for eight, and the launch lasts as long as its slowest lane (a tracker
  // loop's crowd frame: 65 us before the last row workgroup was through).  With the middle tier behind it the pool is not used at
  // all (pairs come from registers, everything else goes to a wavefront: 1000 x 2500 crowd frame 37 -> 31 us against a pool of
  // pairs); without it (Mahalanobis) it takes everything that fits.
  const uint32_t pool_r = no_mid ? SL_R : 0u;
  const SceneDev S = scenes[blockIdx.z];  // by value: wave-uniform SGPRs, cannot alias the stores below
  const bool row_wg = blockIdx.x < row_wgs;  // (the workgroups behind them only take big components off the queue)
  const uint32_t q = row_wg ? blockIdx.x * NT + threadIdx.x : 0xffffffffu;
  __shared__ union { SolveLocal pool[SL_POOL]; MidLocal mid[ML_WAVES]; } s_sh;  // (the middle tier runs after the small one, over its pool)
  SolveLocal* const s_local = s_sh.pool;
  __shared__ unsigned long long s_part[2 * (NT / 64)];
  __shared__ uint32_t s_pool_top, s_word[6], s_fail[NT], s_nfail[ML_WAVES], s_resume;
  extern __shared__ unsigned char s_dyn[];
  SOLVE_STAMP(0);
  if (threadIdx.x == 0) { s_pool_top = 0; s_resume = 0xffffffffu; }
  if (threadIdx.x < ML_WAVES) s_nfail[threadIdx.x] = 0;
  // the forest has done its job (k_assign_label): back to the identity for the next frame's unions
  if (row_wg)
    for (uint32_t i = q; i < S.N + S.T; i += row_wgs * NT) S.parent[i] = i;
  if (q == 0) {  // what the first phase raised goes out with the results; re-armed for the next frame
    S.out_stats[0] = S.stats[0];
    S.out_stats[1] = 0u;   // (k_assign_solve: a bounded wait ran out — the host refuses the frame's results)
    S.stats[0] = 0u;
  }
  if (VISUAL && rearm_words) {
    // the vote words have been read (k_assign_label<WORDS>, the launch before this one): all ones again for the next frame's tiles
    const uint32_t g = blockIdx.x * NT + threadIdx.x, stride = gridDim.x * NT;
    if (S.flags & SCN_WORDSK) {
      for (uint32_t i = g; i < S.N * S.K; i += stride) S.row_cls[i] = ~0ull;
      for (uint32_t i = g; i < S.T * S.K; i += stride) S.col_cls[i] = ~0ull;
    } else {
      for (uint32_t i = g; i < S.N; i += stride) S.row_best[i] = ~0ull;
      for (uint32_t i = g; i < S.T; i += stride) S.col_best[i] = ~0ull;
    }
  }
  __syncthreads();
  bool big = false, mid = false;
  uint32_t mid_R = 0;
  if (q < S.N) {
    // Everything a one-row component needs, requested TOGETHER: what the previous kernels wrote lies in other XCDs' L2s, so each
    // dependent load of this thread is a trip to memory (~1.5 us); the row's first four edge records are fetched before their count
    // is known (what lies beyond the count is stale but addressable) and the track ids / exclusion flags they point at right after.
    const uint32_t head = S.label[q];
    const uint32_t my_edges = S.e_use[q];
    const uint32_t Rq = (uint32_t)S.rnext[q];
    const uint32_t nx_q = S.next_row[q];  // (the other row of a two-row component when this one heads the list; stale otherwise)
    const uint8_t has_q = VISUAL ? S.row_has[q] : (uint8_t)0;
    const int32_t vw_q = VISUAL ? S.vis_winner[q] : -1;
    SaEdge pe[4];
    {
      const SaEdge SA_G* ep = S.e_edge + (size_t)q * S.estride;
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) pe[k2] = sa_ldg(ep + ((uint32_t)k2 < S.estride ? (uint32_t)k2 : S.estride - 1u));
    }
    uint64_t pid[4];
    bool pex[4];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
      const uint32_t c = pe[k2].col < S.T ? pe[k2].col : 0u;  // (a stale record may name a column of a larger, earlier table)
      pid[k2] = S.T ? S.t_ids[c] : 0ull;  // (an empty table has no id array)
      pex[k2] = VISUAL && S.col_excluded[c] != 0;
    }
    if (!my_edges || has_q) {
      // no edge, or decided by the visual vote: this row's result is known (finalize_row_with, with the values already here)
      const bool vis = vw_q >= 0;
      S.out_track_id[q] = vis ? S.t_ids[vw_q] : 0ull;
      S.out_vote[q] = vis ? SA_VOTE_VISUAL : SA_VOTE_NONE;
      S.win_col[q] = vis ? vw_q : -1;
      S.out_win[q] = vis ? vw_q : -1;
    }
    if (head != SA_NONE) {
      const uint32_t R = Rq;  // rows in the component rooted here (k_assign_label)
      if (R == 1 && head == q && my_edges <= 4u) {
        // the usual component of a tracking frame: this row alone with a handful of edges, all of them in registers already
        int64_t bg = 0;
        int32_t bj = -1;
        uint64_t bid = 0;
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2)
          if ((uint32_t)k2 < my_edges && !pex[k2] && (bj < 0 || pe[k2].gain > bg || (pe[k2].gain == bg && (int32_t)pe[k2].col < bj))) {
            bg = pe[k2].gain; bj = (int32_t)pe[k2].col; bid = pid[k2];
          }
        // (a row in a list has no visual verdict: k_assign_label leaves those out)
        S.out_track_id[q] = bj >= 0 ? bid : 0ull;
        S.out_vote[q] = bj >= 0 ? SA_VOTE_POSITIONAL : SA_VOTE_NONE;
        S.win_col[q] = bj;
        S.out_win[q] = bj;
      } else
      if (R == 1) {
        // ONE row takes part (the list's only entry — not necessarily this thread's own row: the forest also holds the rows the
        // visual vote decided, and one of those may be the root): its heaviest usable edge wins (lowest column on ties), or none
        const uint32_t row = head;
        const uint32_t ne = S.e_use[row];
        const SaEdge SA_G* ep = S.e_edge + (size_t)row * S.estride;
        int64_t bg = 0;
        int32_t bj = -1;
        for (uint32_t e0 = 0; e0 < ne; e0 += 4) {
          SaEdge ed[4];
          bool use[4];
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) ed[k2] = e0 + k2 < ne ? sa_ldg(ep + e0 + k2) : SaEdge{0, 0u, 0u};
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) use[k2] = e0 + k2 < ne && !(VISUAL && S.col_excluded[ed[k2].col]);
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2)
            if (use[k2] && (bj < 0 || ed[k2].gain > bg || (ed[k2].gain == bg && (int32_t)ed[k2].col < bj))) { bg = ed[k2].gain; bj = (int32_t)ed[k2].col; }
        }
        finalize_row_with<VISUAL>(S, row, bj);
      } else {
        // TWO rows, this one (the root = the lowest row, in its own list unless the visual vote decided it) and one other, four usable
        // records each at most: three round trips instead of the pool's ten — the other row is known from what is here already (the
        // list's head, or this row's link), its count and first four records come together, their ids / flags right after.  The
        // answer is the greedy start's when the two bids differ (each row its heaviest usable edge, lowest column on ties); when they
        // collide, the better of (this row keeps the column, the other takes its runner-up) and the converse — the root keeps it on
        // a tie, as it does in the solvers' greedy start.
        bool pair_done = false;
        if (R == 2 && !has_q && my_edges - 1u < 4u) {
          const uint32_t other = head != q ? head : nx_q;
          const uint32_t ne_o = other < S.N ? S.e_use[other] : 5u;
          SaEdge po[4];
          {
            const SaEdge SA_G* epo = S.e_edge + (size_t)(other < S.N ? other : 0u) * S.estride;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) po[k2] = sa_ldg(epo + ((uint32_t)k2 < S.estride ? (uint32_t)k2 : S.estride - 1u));
          }
          if (ne_o - 1u < 4u) {
            uint64_t oid[4];
            bool oex[4];
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
              const uint32_t c = po[k2].col < S.T ? po[k2].col : 0u;
              oid[k2] = S.t_ids[c];
              oex[k2] = VISUAL && S.col_excluded[c] != 0;
            }
            // per row: the best usable record and the best one on another column than the rival's bid
            auto best_of = [](const SaEdge* e, const bool* ex, uint32_t n, int32_t not_col, int& at) {
              int64_t bg = 0;
              at = -1;
#pragma unroll
              for (int k2 = 0; k2 < 4; ++k2)
                if ((uint32_t)k2 < n && !ex[k2] && (int32_t)e[k2].col != not_col &&
                    (at < 0 || e[k2].gain > bg || (e[k2].gain == bg && e[k2].col < e[at < 0 ? 0 : at].col))) { bg = e[k2].gain; at = k2; }
              return at < 0 ? (int64_t)0 : bg;
            };
            int ia, ib;
            const int64_t ga = best_of(pe, pex, my_edges, -1, ia), gb = best_of(po, oex, ne_o, -1, ib);
            if (ia >= 0 && ib >= 0 && pe[ia].col == po[ib].col) {
              int ia2, ib2;
              const int64_t ga2 = best_of(pe, pex, my_edges, (int32_t)pe[ia].col, ia2), gb2 = best_of(po, oex, ne_o, (int32_t)po[ib].col, ib2);
              if (ga + gb2 >= ga2 + gb) ib = ib2;
              else ia = ia2;
            }
            int32_t ca = -1, cb = -1;
            uint64_t ida = 0, idb = 0;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
              if (k2 == ia) { ca = (int32_t)pe[k2].col; ida = pid[k2]; }
              if (k2 == ib) { cb = (int32_t)po[k2].col; idb = oid[k2]; }
            }
            S.out_track_id[q] = ida;
            S.out_vote[q] = ca >= 0 ? SA_VOTE_POSITIONAL : SA_VOTE_NONE;
            S.win_col[q] = ca;
            S.out_win[q] = ca;
            S.out_track_id[other] = idb;
            S.out_vote[other] = cb >= 0 ? SA_VOTE_POSITIONAL : SA_VOTE_NONE;
            S.win_col[other] = cb;
            S.out_win[other] = cb;
            pair_done = true;
          }
        }
        bool fits = !pair_done && R <= pool_r;
        uint32_t blk = SA_NONE;
        if (fits) {
          blk = atomicAdd(&s_pool_top, 1u);
          fits = blk < SL_POOL;
        }
        uint32_t E = 0, C = 0;
        if (fits) {
          SolveLocal& L = s_local[blk];
          // rows of the component, ascending
          uint32_t n = 0;
          for (uint32_t cur = head; cur != SA_NONE && n < SL_R; cur = S.next_row[cur]) {
            uint32_t k = n;
            while (k > 0 && L.rows[k - 1] > cur) { L.rows[k] = L.rows[k - 1]; --k; }
            L.rows[k] = cur;
            ++n;
          }
          for (uint32_t r = 0; r < R && fits; ++r) {
            const uint32_t row = L.rows[r];
            const uint32_t cnt = S.e_use[row];
            const SaEdge SA_G* ep = S.e_edge + (size_t)row * S.estride;
            L.e_off[r] = E;
            uint32_t n2 = 0;
            int64_t maxg = 0;
            for (uint32_t e = 0; e < cnt; ++e) {
              const SaEdge ed = sa_ldg(ep + e);
              if (VISUAL && S.col_excluded[ed.col]) continue;  // excluded_tracks (visual_sort/voting.rs:62-71)
              uint32_t c = 0;
              while (c < C && L.colmap[c] != ed.col) ++c;
              if (E >= SL_E || (c == C && C >= SL_C)) { fits = false; break; }
              if (c == C) L.colmap[C++] = ed.col;
              L.e_col[E] = c;
              L.e_gain[E] = ed.gain;
              ++E; ++n2;
              maxg = ed.gain > maxg ? ed.gain : maxg;
            }
            L.e_cnt[r] = n2;
            L.u[r] = -maxg;
            L.rmatch[r] = -1;
            L.next_row[r] = r + 1 < R ? r + 1 : SA_NONE;
          }
        }
        if (fits) {
          SolveLocal& L = s_local[blk];
          // columns in ascending order of their track index: rank, permute, renumber the edges
          for (uint32_t c = 0; c < C; ++c) {
            uint32_t rank = 0;
            for (uint32_t d = 0; d < C; ++d) rank += L.colmap[d] < L.colmap[c];
            L.cnext[c] = (int32_t)rank;
          }
          for (uint32_t c = 0; c < C; ++c) L.pred[L.cnext[c]] = (int32_t)L.colmap[c];
          for (uint32_t e = 0; e < E; ++e) L.e_col[e] = (uint32_t)L.cnext[L.e_col[e]];
          for (uint32_t c = 0; c < C; ++c) { L.colmap[c] = (uint32_t)L.pred[c]; L.v[c] = 0; L.cmatch[c] = -1; L.cstamp[c] = 0; L.cscan[c] = 0; }
          sa_assign_ws w;
          w.e_cnt = L.e_cnt; w.e_col = L.e_col; w.e_gain = L.e_gain; w.ecs = 1; w.egs = 1; w.rcs = 1; w.rgs = 1; w.estride = 0; w.e_off = L.e_off;
          w.excluded = nullptr;
          w.next_row = L.next_row;
          w.u = L.u; w.v = L.v; w.rmatch = L.rmatch; w.cmatch = L.cmatch; w.dist = L.dist; w.pred = L.pred;
          w.cstamp = L.cstamp; w.cscan = L.cscan; w.cnext = L.cnext; w.rdist = L.rdist; w.rnext = L.rnext;
          sa_assign_component(w, 0);
          for (uint32_t r = 0; r < R; ++r) {
            const int32_t c = L.rmatch[r];
            finalize_row_with<VISUAL>(S, L.rows[r], c >= 0 ? (int32_t)L.colmap[c] : -1);
          }
        } else if (pair_done) {
        } else if (R <= ML_R && !no_mid) { mid = true; mid_R = R; }
        else big = true;
      }
    }
  }
  // Components beyond the pool go onto one of the scene's two queues: mid-sized ones (up to ML_R rows) for single wavefronts, the rest
  // for whole workgroups.  A workgroup that is through with its rows says so; its first wavefront serves its share of the mid-sized
  // queue (below), and once every row workgroup has reported EVERY workgroup of the scene takes big components by ticket (all its
  // threads on one component) and finally finishes what its own wavefront had to refuse.  (The waits are for workgroups dispatched
  // EARLIER in the same grid and are bounded.)
  SOLVE_STAMP(1);
  if (row_wg) {
    if (big) sa_queue_push(S, q, false, 0u);
    if (mid) sa_queue_push(S, q, true, mid_R);
    // (queue entries and counters are agent-scope atomics: no cache maintenance — an agent-scope fence by every thread here costs
    // 10 us at C4; everything else the takers read was written by earlier launches)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add((uint32_t*)(S.stats + SA_QW_DONE), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // FORWARD PROGRESS: a row workgroup never waits — it leaves here.  Everything below waits for "every row workgroup of the scene has
    // reported", and the only workgroups that wait are the HELPERS launched behind the scene's row workgroups: on every XCD a helper is
    // dispatched after the row workgroups of its scene that share the XCD (workgroups are handed out in index order), and those never
    // block, so the report count always gets there — whatever the device's residency: a frame of 10^5 detections, a CU mask, a
    // partitioned device, another kernel (the ReID model that fills the registered feature block) holding most of the CUs.  Before,
    // the row workgroups themselves served the queues behind the same wait, which needed ALL of them resident at once: a frame
    // beyond that (or a masked device) ran into the bounded wait and failed.  Served by helpers alone the queues lose 4 of 128
    // wavefronts at 1000 detections.
    SOLVE_STAMP(5);
    return;
  }
  const uint32_t helper = blockIdx.x - row_wgs, n_helpers = gridDim.x - row_wgs;   // (the launcher always adds helpers: launch_solve)
  // Middle tier first, AS THE ENTRIES ARRIVE: the first wavefront of every workgroup serves its entries of the mid-sized queue the
  // moment they have landed (k_assign_label left SA_NONE in every slot) — a crowd's knots are being solved while the slowest row
  // workgroup is still on its pairs (in-kernel timeline of a tracker loop's crowd frame: the last row workgroup was through 11 us after
  // the first one entered, and the wait for it was time no knot was worked on).  An entry beyond the queue's end is known as such once
  // every row workgroup has reported (SA_QW_DONE) and the slot still holds SA_NONE: the pushes of a workgroup are performed (returning
  // exchanges) before its report.
  // Everything in this loop is wave-uniform BY CONSTRUCTION — indices go through readfirstlane, a refusal is recorded by every lane
  // writing the same word: a lane-divergent branch before the back edge lets the compiler keep the two groups of lanes apart across
  // iterations (seen with a ticket taken by lane 0 only: a refusal pushed by `if (lane == 0)` inside such a loop hung the workgroup).
  const uint32_t N = S.N, T = S.T;
  [[maybe_unused]] uint32_t took_big = 0, took_mid = 0;
  uint32_t nbig_seen = SA_NONE;  // wave 0: the length of the big queue, read once every row workgroup had reported (SA_NONE: not seen yet)
  if (!no_mid) {
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t nref = 0;
    if (wv < ML_WAVES) {
      // No tickets: helper b serves the entries b, b + (helpers), ... of the queue — entries are numbered in the order they were
      // pushed, workgroups in the order they were dispatched, so the first knots go to the workgroups that are there first, and a
      // frame WITHOUT a mid-sized component (most tracking frames) leaves here at the price of the old wait: one poll of its slot and of
      // the report count, then the slot again + the big queue's length once every row workgroup has reported.
      uint32_t k = helper;
      while (nref < NT / ML_WAVES) {  // (a full refusal list: this wavefront serves no more — its later entries would be lost, so it is sized for every row it could get)
        if (k >= N) break;  // (at most one mid-sized component per row)
        uint32_t ent = SA_NONE;
        for (uint32_t spin = 0; spin < (1u << 22); ++spin) {  // (bounded: ~1 s; a slot that never fills and a report that never comes is a bug, not a hang)
          const uint32_t v = __hip_atomic_load((uint32_t*)S.dq + N + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t d = __hip_atomic_load((uint32_t*)(S.stats + SA_QW_DONE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
          if (ent != SA_NONE) break;
          if ((uint32_t)__builtin_amdgcn_readfirstlane((int)d) >= row_wgs) {  // every push has been performed: what the slot holds now is final
            const uint32_t v2 = __hip_atomic_load((uint32_t*)S.dq + N + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t l2 = __hip_atomic_load((uint32_t*)(S.stats + SA_QW_LEN), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (same trip)
            ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)v2);
            nbig_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)l2);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          if (spin + 1u == (1u << 22)) S.out_stats[1] = 1u;   // gave up: a row workgroup that never reported (a partitioned or shared device?)
        }
        k += n_helpers;
        if (ent == SA_NONE) break;
        ++took_mid;
        const uint32_t root = ent & 0xffffffu, R = ent >> 24;
        const bool ok = mid_solve_component<VISUAL>(S, root, R, s_sh.mid[wv]);
        s_fail[wv * (NT / ML_WAVES) + nref] = root;
        nref += ok ? 0u : 1u;
      }
      if ((threadIdx.x & 63u) == 0) { s_nfail[wv] = nref; s_resume = k; }   // (k: the first of this workgroup's entries the wavefront did not look at)
    }
  }
  SOLVE_STAMP(2);
  // Then the big components, one WORKGROUP each, by ticket — their queue is complete once every row workgroup has reported (the
  // wavefront above left its loop on that condition unless its refusal list filled first; the wait is for workgroups dispatched EARLIER
  // in the same grid, which never wait themselves).
  if (threadIdx.x == 0) {
    if (nbig_seen == SA_NONE) {
      for (uint32_t spin = 0; spin < (1u << 22) && __hip_atomic_load((uint32_t*)(S.stats + SA_QW_DONE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < row_wgs; ++spin) {
        __builtin_amdgcn_s_sleep(4);
        if (spin + 1u == (1u << 22)) S.out_stats[1] = 1u;
      }
      nbig_seen = __hip_atomic_load((uint32_t*)(S.stats + SA_QW_LEN), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_word[3] = nbig_seen;
  }
  __syncthreads();  // (also: the middle tier's blocks — they lie over the pool of the small tier — are free)
  const uint32_t nbig = s_word[3];
  SOLVE_NOTE(6, ((unsigned long long)nbig << 32) | __hip_atomic_load((uint32_t*)(S.stats + SA_QW_MLEN), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  uint32_t nrefused = 0;
  for (uint32_t w2 = 0; w2 < ML_WAVES; ++w2) nrefused += s_nfail[w2];
  // (a wavefront whose refusal list filled up left entries of this workgroup unserved: the workgroup takes them below, one by one)
  const bool leftover = nrefused >= NT / ML_WAVES && s_resume < N;
  if (nbig == 0 && nrefused == 0) { SOLVE_STAMP(5); return; }
  int64_t* u = LDS_STATE ? (int64_t*)s_dyn : (int64_t*)S.u_use;
  int32_t* rmatch = LDS_STATE ? (int32_t*)(s_dyn + (size_t)N * 8) : (int32_t*)S.rmatch;
  int32_t* cmatch = LDS_STATE ? (int32_t*)(s_dyn + (size_t)N * 12) : (int32_t*)S.cmatch;
  int32_t* pred = LDS_STATE ? (int32_t*)(s_dyn + (size_t)N * 12 + (size_t)T * 4) : (int32_t*)S.pred;
  bool fresh = true;
  auto dense_one = [&](uint32_t root) {  // (every thread of the workgroup, uniformly)
    if (LDS_STATE && fresh) {  // (the HBM arrays were reset by the frame's preparation blocks)
      for (uint32_t i = threadIdx.x; i < N; i += NT) rmatch[i] = -1;
      for (uint32_t i = threadIdx.x; i < T; i += NT) cmatch[i] = -1;
      fresh = false;
      __syncthreads();
    }
    dense_solve_component<VISUAL, NT, CPT, LDS_STATE>(S, root, u, rmatch, cmatch, pred, s_part, s_word);
  };
  if (nbig)
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) s_word[3] = atomicAdd((uint32_t*)(S.stats + SA_QW_TICKET), 1u);
      __syncthreads();
      const uint32_t k = s_word[3];
      if (k >= nbig) break;
      ++took_big;
      dense_one(__hip_atomic_load((uint32_t*)S.dq + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  SOLVE_STAMP(3);
  SOLVE_NOTE(7, ((unsigned long long)took_big << 32) | took_mid);
  // ... and what the workgroup's own wavefronts had to refuse (more distinct columns than the middle tier's matrix holds)
  for (uint32_t w2 = 0; w2 < ML_WAVES; ++w2) {
    const uint32_t nf = s_nfail[w2];
    for (uint32_t i = 0; i < nf; ++i) dense_one(s_fail[w2 * (NT / ML_WAVES) + i]);
  }
  if (leftover)
    for (uint32_t k = s_resume; k < N; k += n_helpers) {   // (every row workgroup has reported: the slots are final)
      const uint32_t ent = __hip_atomic_load((uint32_t*)S.dq + N + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ent == SA_NONE) break;
      dense_one(ent & 0xffffffu);
    }
  SOLVE_STAMP(5);
}

// =====================================================================================================
// Launchers
// =====================================================================================================
hipError_t sa_launch_prep_tracks(const PrepTrackArgs& a, const SaParams& p, hipStream_t st) {
  if (!a.n) return hipSuccess;
  hipLaunchKernelGGL(k_prep_tracks, dim3(cdiv(a.n, 256)), dim3(256), 0, st, a, p);
  return hipGetLastError();
}
hipError_t sa_launch_pad_features(const float* src, uint32_t rows, uint32_t D, uint32_t Dp, uint32_t K,
                                  const uint32_t* slots, const uint8_t* present, float* dst, float* norms,
                                  uint8_t* dst_present, uint32_t* fcount, hipStream_t st, float* dst_frag) {
  if (!rows) return hipSuccess;
  hipLaunchKernelGGL(k_pad_features, dim3(cdiv(rows, 4)), dim3(256), 0, st, src, rows, D, Dp, K, slots, present, dst,
                     norms, dst_present, dst_frag);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (fcount && slots) {
    uint32_t n = rows / K;
    hipLaunchKernelGGL(k_feat_count, dim3(cdiv(n, 256)), dim3(256), 0, st, slots, n, K, (const uint8_t*)dst_present, fcount);
    e = hipGetLastError();
  }
  return e;
}
hipError_t sa_launch_gather_rows(const void* src, void* dst, const uint32_t* index, uint32_t rows, uint32_t row_bytes,
                                 hipStream_t st) {
  if (!rows || !row_bytes) return hipSuccess;
  hipLaunchKernelGGL(k_gather_rows, dim3(rows), dim3(row_bytes >= 1024 ? 256 : 64), 0, st, (const uint8_t*)src,
                     (uint8_t*)dst, index, rows, row_bytes);
  return hipGetLastError();
}
hipError_t sa_launch_slot_init(uint32_t* e_cnt, int64_t* u, uint32_t n_rows, uint32_t* parent, uint32_t n_vertices, hipStream_t st) {
  const uint32_t n = n_rows > n_vertices ? n_rows : n_vertices;
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_slot_init, dim3(cdiv(n, 256)), dim3(256), 0, st, e_cnt, u, n_rows, parent, n_vertices);
  return hipGetLastError();
}
// First launch of a frame: positional tiles + frame-preparation blocks (see k_frame).
hipError_t sa_launch_frame(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxT, int visual, const SaParams& p,
                           hipStream_t st, int prep) {
  const bool force_general = p.force_general != 0;
  // wide (16 x 256) positional tiles when the frame still gives at least one block per CU that way
  const bool wide = (size_t)cdiv(maxT, 256) * cdiv(maxN, POS_TI) * ns >= 256;
  const bool uni = maxN > SA_SMALL_T || maxT > 2u * SA_SMALL_T || force_general;  // the one-workgroup tail builds duals and components itself (enqueue_frame sets force_general for every frame it sends to the other tail)
  const uint32_t gx = maxT ? cdiv(maxT, wide ? 256 : 64) : 1u;
  const uint32_t pos_rows = (maxN && maxT && prep != 2) ? cdiv(maxN, POS_TI) : 0u;
  uint32_t prep_blocks = cdiv(maxN + maxT + 1, 256);
  if (visual && prep != 3 && cdiv(maxN, 4) > prep_blocks) prep_blocks = cdiv(maxN, 4);
  if (prep == 0) prep_blocks = 0;
  if (!pos_rows && !prep_blocks) return hipSuccess;
  const dim3 grid(gx, pos_rows + cdiv(prep_blocks, gx), ns);
  sa_pos_trace_hook(st, grid.x * grid.y * grid.z);
  const uint32_t pr = pos_rows | (prep == 3 ? 0x80000000u : 0u);
  const bool crowded = (size_t)grid.x * grid.y * grid.z > 1024u;   // (more blocks than four per CU hold: the lighter tile, six per CU)
  if (wide && uni && crowded) SA_LAUNCH((k_frame<4, true, 32>), grid, dim3(256), 0, st, scenes, p, pr);
  else if (wide && uni) SA_LAUNCH((k_frame<4, true>), grid, dim3(256), 0, st, scenes, p, pr);
  else if (wide && crowded) SA_LAUNCH((k_frame<4, false, 32>), grid, dim3(256), 0, st, scenes, p, pr);
  else if (wide) SA_LAUNCH((k_frame<4, false>), grid, dim3(256), 0, st, scenes, p, pr);
  else if (uni) SA_LAUNCH((k_frame<1, true>), grid, dim3(256), 0, st, scenes, p, pr);
  else SA_LAUNCH((k_frame<1, false>), grid, dim3(256), 0, st, scenes, p, pr);
  return hipGetLastError();
}
// parity taps: the dense f32 cost matrix of the staged scenes, no side effects on the assignment state
hipError_t sa_launch_positional_dense(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxT, const SaParams& p,
                                      hipStream_t st) {
  if (!maxN || !maxT) return hipSuccess;
  hipLaunchKernelGGL(k_positional_dense, dim3(cdiv(maxT, 64), cdiv(maxN, POS_TI), ns), dim3(256), 0, st, scenes, p);
  return hipGetLastError();
}
hipError_t sa_launch_quant_tap(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxT, hipStream_t st) {
  if (!maxN || !maxT) return hipSuccess;
  size_t b = ((size_t)maxN * maxT + 255) / 256;
  uint32_t blocks = b > 2048 ? 2048u : (uint32_t)b;
  hipLaunchKernelGGL(k_quant_tap, dim3(blocks, 1, ns), dim3(256), 0, st, scenes);
  return hipGetLastError();
}
hipError_t sa_launch_bestfit(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxT, const SaParams& p,
                             hipStream_t st, int stage) {
  if (!maxN || !maxT) return hipSuccess;
  if (stage == 0) SA_LAUNCH(k_bestfit_tile, dim3(cdiv(maxT, 64), cdiv(maxN, 64), ns), dim3(256), 0, st, scenes, p);
  else if (stage == 2) SA_LAUNCH(k_bestfit_resolve<true>, dim3(cdiv(maxN, 4), 1, ns), dim3(256), 0, st, scenes);  // partials from the contraction
  else SA_LAUNCH(k_bestfit_resolve<false>, dim3(cdiv(maxN, 4), 1, ns), dim3(256), 0, st, scenes);
  return hipGetLastError();
}
// one instantiation of the general tail's solver.  Above 64 KB the dynamic LDS limit has to be asked for, and the attribute belongs to
// the (function, DEVICE) pair: one engine thread per GPU (sa_cluster) launches the same instantiation on different devices, so the
// size already granted is remembered per device (an atomic per ordinal: the threads do not share a lock on the launch path).
#include <atomic>
#define SA_MAX_DEVICES 64
template <bool VIS, int NT, int CPT, bool LDS_STATE>
static hipError_t launch_solve_one(dim3 grid, uint32_t row_wgs, size_t lds, hipStream_t st, const SceneDev* scenes) {
  if (LDS_STATE && lds > 64u * 1024u) {
    static std::atomic<size_t> allowed[SA_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
    const bool cached = dev >= 0 && dev < SA_MAX_DEVICES;
    if (!cached || lds > allowed[dev].load(std::memory_order_relaxed)) {
      const hipError_t ae = hipFuncSetAttribute((const void*)k_assign_solve<VIS, NT, CPT, LDS_STATE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (ae != hipSuccess) return ae;
      if (cached) {
        size_t cur = allowed[dev].load(std::memory_order_relaxed);
        while (cur < lds && !allowed[dev].compare_exchange_weak(cur, lds, std::memory_order_relaxed)) {}
      }
    }
  }
  sa_solve_trace_before(st);
  SA_LAUNCH((k_assign_solve<VIS, NT, CPT, LDS_STATE>), grid, dim3(NT), LDS_STATE ? lds : 0, st, scenes, row_wgs);
  sa_solve_trace_after(st, grid.x);
  return hipSuccess;
}
template <int NT, int CPT>
static hipError_t launch_solve(bool vis, bool in_lds, bool no_mid, bool words, uint32_t maxN, uint32_t ns, size_t lds, hipStream_t st, const SceneDev* scenes) {
  // the row workgroups, and behind them helpers that only take components off the scene's queues (a crowd has dozens of knots, one
  // wavefront of a workgroup each: 64 workgroups per scene left the 1000 x 2500 crowd frame two rounds of them, 30 us; 128: 24 us) —
  // fewer per scene in a wide batch
  const uint32_t rows = cdiv(maxN, NT);
  const uint32_t want = ns >= 16 ? 16u : ns >= 4 ? 32u : 128u;
  // (the row workgroups leave once their rows are through — they never wait, see the kernel — so every scene gets helpers of its own:
  // what is left of `want`, at least half of it)
  const uint32_t helpers = want > rows + want / 2 ? want - rows : want / 2;
  const dim3 grid(rows + helpers, 1, ns);
  const uint32_t rw = rows | (no_mid ? 0x80000000u : 0u) | (words ? 0x40000000u : 0u);  // (bit 31: no middle tier; bit 30: re-arm the vote words)
  if (vis && in_lds) return launch_solve_one<true, NT, CPT, true>(grid, rw, lds, st, scenes);
  if (vis) return launch_solve_one<true, NT, CPT, false>(grid, rw, lds, st, scenes);
  if (in_lds) return launch_solve_one<false, NT, CPT, true>(grid, rw, lds, st, scenes);
  return launch_solve_one<false, NT, CPT, false>(grid, rw, lds, st, scenes);
}
hipError_t sa_launch_assign(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxT, const SaParams& p,
                            hipStream_t st, int stage, uint64_t done_seq) {
  if (!maxN) return hipSuccess;
  switch (stage) {
    case 1: SA_LAUNCH(k_assign_label<false>, dim3(cdiv(maxN, 256), 1, ns), dim3(256), 0, st, scenes); break;
    case 2: SA_LAUNCH(k_assign_label<true>, dim3(cdiv(maxN > maxT ? maxN : maxT, 256), 1, ns), dim3(256), 0, st, scenes); break;  // (one thread per row AND per column)
    case 3: case 4: {
      // columns per thread of the dense solver by the widest scene; its per-row / per-column state in dynamic LDS when it fits beside
      // the pool of private blocks, else in the scene's HBM arrays
      const bool vis = p.visual_kind != SA_VIS_NONE;
      const size_t lds = (size_t)maxN * 12 + (size_t)maxT * 8;
      const bool in_lds = lds <= 96u * 1024u;
      const bool no_mid = p.positional_kind == SA_POS_MAHALANOBIS;  // gains of 1e8: beyond the middle tier's 32-bit cells
      hipError_t se;
      if (maxT <= 256u * 4u) se = launch_solve<256, 4>(vis, in_lds, no_mid, stage == 4, maxN, ns, lds, st, scenes);
      else if (maxT <= 256u * 8u) se = launch_solve<256, 8>(vis, in_lds, no_mid, stage == 4, maxN, ns, lds, st, scenes);
      else if (maxT <= 256u * 16u) se = launch_solve<256, 16>(vis, in_lds, no_mid, stage == 4, maxN, ns, lds, st, scenes);
      else if (maxT <= 256u * 32u) se = launch_solve<256, 32>(vis, in_lds, no_mid, stage == 4, maxN, ns, lds, st, scenes);
      else if (maxT <= 1024u * 32u) se = launch_solve<1024, 32>(vis, in_lds, no_mid, stage == 4, maxN, ns, lds, st, scenes);
      else return hipErrorInvalidValue;  // more than 32768 tracks in one scene (refused earlier, in bank_prepare)
      if (se != hipSuccess) return se;
      break;
    }
    default:
      sa_tail_trace_hook(st, ns);
      if (maxN > SA_SMALL_N) {   // two rows and two columns per thread (N, T <= SA_SMALL_T)
        if (stage == 8) SA_LAUNCH((k_assign_small2<true, true, 64>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
        else if (p.visual_kind != SA_VIS_NONE) SA_LAUNCH((k_assign_small2<true, false, 64>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
        else SA_LAUNCH((k_assign_small2<false, false, 64>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
      } else if (maxT > SA_SMALL_T) {   // four columns per thread (N <= SA_SMALL_N, T <= 2 SA_SMALL_T)
        if (stage == 8) SA_LAUNCH((k_assign_small2<true, true, 64, 1, 4>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
        else if (p.visual_kind != SA_VIS_NONE) SA_LAUNCH((k_assign_small2<true, false, 64, 1, 4>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
        else SA_LAUNCH((k_assign_small2<false, false, 64, 1, 4>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
      } else if (maxT > SA_SMALL_N) {   // two columns per thread (T <= SA_SMALL_T)
        if (stage == 8) SA_LAUNCH((k_assign_small<true, true, 64, 2>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
        else if (p.visual_kind != SA_VIS_NONE) SA_LAUNCH((k_assign_small<true, false, 64, 2>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
        else SA_LAUNCH((k_assign_small<false, false, 64, 2>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
      } else if (stage == 8) SA_LAUNCH((k_assign_small<true, true, 64>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
      else if (p.visual_kind != SA_VIS_NONE) SA_LAUNCH((k_assign_small<true, false, 64>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
      else SA_LAUNCH((k_assign_small<false, false, 64>), dim3(1, 1, ns), dim3(SA_SMALL_N), 0, st, scenes, done_seq);
      break;
  }
  return hipGetLastError();
}
