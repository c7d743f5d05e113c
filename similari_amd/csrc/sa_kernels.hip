// sa_kernels.hip — gfx950 kernels of the association path other than the feature contraction:
// box / track preparation, pair pre-filter + IoU / Mahalanobis cost cells, BestFit vote, and the
// positional assignment (edge compaction, connected components, exact per-component solve).
//
// These are HBM/latency-bound integer / f64 / f32 elementwise kernels: the design rules are coalesced
// 256-B row segments per wave, LDS staging of the per-tile operands, wave ballots for compaction, and
// grid.z = scene so a whole batch of scenes goes through one launch.  No MFMA here on purpose.
#include "sa_engine.h"

#define WAVE 64

// =====================================================================================================
// Preparation
// =====================================================================================================
__device__ __forceinline__ void prep_box_common(const BoxRaw& r, sa_geo* geo, double* verts) {
  const sa_box& b = r.box;
  sa_geo g;
  g.xc = b.xc;
  g.yc = b.yc;
  g.r = sa_radius(b.aspect, b.height);
  g.hha = b.height * b.height * b.aspect;
  *geo = g;
  sa_vertices(b.xc, b.yc, b.aspect, b.height, r.c, r.s, verts);
}

// Candidates of one frame (visual_sort/simple_api.rs:130-170): geometry, vertices, Mahalanobis measurement
// (angle.unwrap_or(0), kalman_2d_box.rs:159), clamped confidence (sort/metric.rs:43-47) and the
// feature_can_be_used gate (visual_sort/metric.rs:227-249).
__global__ void k_prep_cands(PrepCandArgs a, SaParams p) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  BoxRaw r = a.raw[i];
  prep_box_common(r, &a.geo[i], &a.verts[(size_t)i * 8]);
  const sa_box& b = r.box;
  float* z = &a.z[(size_t)i * 5];
  z[0] = b.xc; z[1] = b.yc; z[2] = b.has_angle ? b.angle : 0.0f; z[3] = b.aspect; z[4] = b.height;
  a.conf[i] = b.confidence < p.min_confidence ? p.min_confidence : b.confidence;
  bool usable = false;
  if (a.has_feats && (!a.feat_present || a.feat_present[i])) {
    float q = a.quality ? a.quality[i] : 1.0f;
    bool quality_ok = q >= p.visual_minimal_quality_use;
    bool perc_ok = true;
    if (a.own_area) {
      float oa = a.own_area[i];
      if (oa == oa) perc_ok = oa >= p.visual_minimal_own_area_use;
    }
    bool bbox_ok = sa_area(b.aspect, b.height) >= p.visual_minimal_area;
    usable = bbox_ok && quality_ok && perc_ok;
  }
  a.usable[i] = usable ? 1 : 0;
}

// Stored tracks touched by an upsert: scatter to their table rows; Kalman projection + Cholesky once per
// track (kalman_2d_box.rs:104-120,167) instead of once per pair.
__global__ void k_prep_tracks(PrepTrackArgs a, SaParams p) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  uint32_t s = a.slots[i];
  BoxRaw r = a.raw[i];
  prep_box_common(r, &a.geo[s], &a.verts[(size_t)s * 8]);
  a.t_epoch[s] = a.epochs[i];
  a.t_ids[s] = a.ids[i];
  if (a.kf_mean && a.kf_cov) sa_maha_prepare(p.kf_position_weight, a.kf_mean + (size_t)i * 5, a.kf_cov + (size_t)i * 25, a.maha + (size_t)s * 20);
}

// One wave per feature row: zero-pad D -> D8 (Feature::from_vec, track/utils.rs:45-71), scatter, squared norm
// (the per-pair norms of distance.rs:36-44 hoisted to once per vector).
__global__ void k_pad_features(const float* __restrict__ src, uint32_t rows, uint32_t D, uint32_t D8, uint32_t K,
                               const uint32_t* __restrict__ slots, const uint8_t* __restrict__ present,
                               float* __restrict__ dst, float* __restrict__ norms, uint8_t* __restrict__ dst_present) {
  uint32_t row = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  uint32_t lane = threadIdx.x % WAVE;
  if (row >= rows) return;
  uint32_t drow = slots ? slots[row / K] * K + row % K : row;
  bool pres = present ? present[row] != 0 : true;
  const float* s = src + (size_t)row * D;
  float* d = dst + (size_t)drow * D8;
  float acc = 0.0f;
  for (uint32_t k = lane; k < D8; k += WAVE) {
    float x = (pres && k < D) ? s[k] : 0.0f;
    d[k] = x;
    acc += x * x;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    norms[drow] = acc;
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

// =====================================================================================================
// Per-frame state reset (one thread per vertex of the bipartite graph)
// =====================================================================================================
__global__ void k_frame_init(const SceneDev* __restrict__ scenes) {
  const SceneDev& S = scenes[blockIdx.z];
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t N = S.N, T = S.T;
  if (i == 0) *S.vis_max_key = sa_f32_key(-1.0f);  // BestFit max_dist starts at -1.0 (voting/best.rs:59)
  if (i < N + T) S.parent[i] = i;
  if (i < N) {
    S.vis_winner[i] = -1;
    S.row_has[i] = 0;
    S.rmatch[i] = -1;
    S.e_cnt[i] = 0;
    S.label[i] = SA_NONE;
    S.next_row[i] = SA_NONE;
  }
  if (i < T) {
    S.col_max_w[i] = 0ull;
    S.col_min_q[i] = SA_NONE;
    S.col_excluded[i] = 0;
    S.v[i] = 0;
    S.cmatch[i] = -1;
    S.cstamp[i] = 0;
    S.cscan[i] = 0;
  }
}

// =====================================================================================================
// Positional cost cells: pair pre-filter -> (survivors only) IoU by f64 Sutherland–Hodgman / Mahalanobis.
// Tile = 16 candidates x 64 tracks per 256-thread block; each wave owns whole 256-B row segments of `pos`.
// Phase 1 tests every cell against compatible() and too_far() from LDS-staged geometry and writes NaN
// for the dead ones; the few survivors are compacted into an LDS list so that phase 2 runs the expensive
// clip with full lanes instead of 1-2 live lanes per wave.
// =====================================================================================================
#define POS_TI 16
#define POS_TJ 64
__global__ __launch_bounds__(256) void k_positional(const SceneDev* __restrict__ scenes, SaParams p) {
  const SceneDev& S = scenes[blockIdx.z];
  const uint32_t N = S.N, T = S.T;
  const uint32_t i0 = blockIdx.y * POS_TI, j0 = blockIdx.x * POS_TJ;
  if (i0 >= N || j0 >= T) return;
  __shared__ sa_geo s_cg[POS_TI];
  __shared__ sa_geo s_tg[POS_TJ];
  __shared__ uint64_t s_te[POS_TJ];
  __shared__ uint16_t s_list[POS_TI * POS_TJ];
  __shared__ uint32_t s_cnt;
  const uint32_t tid = threadIdx.x;
  if (tid < POS_TI) {
    uint32_t i = i0 + tid;
    s_cg[tid] = i < N ? S.c_geo[i] : sa_geo{0.f, 0.f, 0.f, 0.f};
  } else if (tid >= 64 && tid < 64 + POS_TJ) {
    uint32_t lj = tid - 64, j = j0 + lj;
    s_tg[lj] = j < T ? S.t_geo[j] : sa_geo{0.f, 0.f, 0.f, 0.f};
    s_te[lj] = j < T ? S.t_epoch[j] : 0ull;
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  const float nanv = __builtin_nanf("");
  const uint64_t epoch = S.epoch;
#pragma unroll
  for (int it = 0; it < (POS_TI * POS_TJ) / 256; ++it) {
    uint32_t c = it * 256 + tid;
    uint32_t li = c / POS_TJ, lj = c % POS_TJ;
    uint32_t i = i0 + li, j = j0 + lj;
    bool live = false;
    if (i < N && j < T) {
      const sa_geo cg = s_cg[li], tg = s_tg[lj];
      live = sa_compatible(cg, epoch, tg, s_te[lj], p.max_idle, p.cons) && !sa_too_far(cg, tg);
      if (!live) S.pos[(size_t)i * T + j] = nanv;
    }
    if (live) {
      uint32_t slot = atomicAdd(&s_cnt, 1u);
      s_list[slot] = (uint16_t)c;
    }
  }
  __syncthreads();
  const uint32_t cnt = s_cnt;
  for (uint32_t sidx = tid; sidx < cnt; sidx += 256) {
    uint32_t c = s_list[sidx];
    uint32_t li = c / POS_TJ, lj = c % POS_TJ;
    uint32_t i = i0 + li, j = j0 + lj;
    float conf = S.c_conf[i];
    float out;
    if (p.positional_kind == SA_POS_MAHALANOBIS) {
      float m20[20], z5[5];
      const float* mp = S.t_maha + (size_t)j * 20;
#pragma unroll
      for (int k = 0; k < 20; ++k) m20[k] = mp[k];
      const float* zp = S.c_z + (size_t)i * 5;
#pragma unroll
      for (int k = 0; k < 5; ++k) z5[k] = zp[k];
      out = sa_maha_cell(m20, z5, conf);
    } else {
      double cv[8], tv[8];
      const double* cp = S.c_verts + (size_t)i * 8;
      const double* tp = S.t_verts + (size_t)j * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) { cv[k] = cp[k]; tv[k] = tp[k]; }
      float iou;
      out = nanv;
      if (sa_iou_cell(cv, tv, s_cg[li].hha, s_tg[lj].hha, &iou)) {
        float e = iou * conf;
        if (e >= p.positional_threshold) out = e;
      }
    }
    S.pos[(size_t)i * T + j] = out;
  }
}

// Debug tap: (w * 1e6f) as i64 of every positional cell, 0 where absent (sort/voting.rs:59).
__global__ void k_quant_tap(const SceneDev* __restrict__ scenes) {
  const SceneDev& S = scenes[blockIdx.z];
  size_t n = (size_t)S.N * S.T;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
    float w = S.pos[c];
    S.quant[c] = sa_quantise(w == w ? w : 0.0f);
  }
}

// =====================================================================================================
// BestFit vote (track/voting/best.rs:52-128) without the sort: candidate q wins track t*(q) iff its group
// (q, t*) is the first group of column t* in (weight desc, q asc) order — SURVEY Appendix A3.
//   stage 0: one wave per candidate row: W[q,t] = sum_k f64(max_dist - w_k) over present k (count >= votes),
//            row argmax (W desc, t asc), column max via 64-bit atomicMax on the f64 bit pattern (W >= 0).
//   stage 1: second sweep: lowest q among the cells that reach the column max.
//   stage 2: per candidate decision + excluded_tracks (visual_sort/voting.rs:62-71).
// =====================================================================================================
__device__ __forceinline__ bool bestfit_cell(const SceneDev& S, const SaParams& p, uint32_t q, uint32_t t, float max_dist,
                                             double* W) {
  const float* v = S.vis + ((size_t)q * S.T + t) * S.K;
  uint32_t cnt = 0;
  double w = 0.0;
  for (uint32_t k = 0; k < S.K; ++k) {
    float x = v[k];
    if (x == x) { ++cnt; w += (double)(max_dist - x); }
  }
  *W = w;
  return cnt >= 1 && cnt >= p.min_votes;
}

__global__ __launch_bounds__(256) void k_bestfit_rows(const SceneDev* __restrict__ scenes, SaParams p) {
  const SceneDev& S = scenes[blockIdx.z];
  const uint32_t q = blockIdx.x * 4 + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (q >= S.N) return;
  const float max_dist = sa_key_f32(*S.vis_max_key);
  double bw = -1.0;
  int32_t bt = -1;
  for (uint32_t base = 0; base < S.T; base += WAVE) {
    uint32_t t = base + lane;
    if (t < S.T) {
      double W;
      if (bestfit_cell(S, p, q, t, max_dist, &W)) {
        atomicMax(&S.col_max_w[t], (unsigned long long)__double_as_longlong(W));
        if (W > bw) { bw = W; bt = (int32_t)t; }  // per lane t ascends, so strict > keeps the lowest t
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    double ow = __shfl_xor(bw, o);
    int32_t ot = __shfl_xor(bt, o);
    if (ot >= 0 && (bt < 0 || ow > bw || (ow == bw && ot < bt))) { bw = ow; bt = ot; }
  }
  if (lane == 0) {
    S.row_has[q] = bt >= 0 ? 1 : 0;
    S.row_best_t[q] = bt;
    S.row_best_w[q] = bw;
  }
}

__global__ __launch_bounds__(256) void k_bestfit_ties(const SceneDev* __restrict__ scenes, SaParams p) {
  const SceneDev& S = scenes[blockIdx.z];
  const uint32_t q = blockIdx.x * 4 + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (q >= S.N) return;
  if (!S.row_has[q]) return;
  const float max_dist = sa_key_f32(*S.vis_max_key);
  for (uint32_t base = 0; base < S.T; base += WAVE) {
    uint32_t t = base + lane;
    if (t < S.T) {
      double W;
      if (bestfit_cell(S, p, q, t, max_dist, &W) &&
          (unsigned long long)__double_as_longlong(W) == S.col_max_w[t])
        atomicMin(&S.col_min_q[t], q);
    }
  }
}

__global__ void k_bestfit_resolve(const SceneDev* __restrict__ scenes) {
  const SceneDev& S = scenes[blockIdx.z];
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= S.N) return;
  if (!S.row_has[q]) return;
  int32_t t = S.row_best_t[q];
  bool win = (unsigned long long)__double_as_longlong(S.row_best_w[q]) == S.col_max_w[t] && S.col_min_q[t] == q;
  if (win) {
    S.vis_winner[q] = t;
    S.col_excluded[t] = 1;
  }
}

// =====================================================================================================
// Positional assignment = SortVoting::winners (sort/voting.rs:30-100) as an exact sparse solve.
//   stage 0  k_assign_edges : one wave per candidate row scans pos[q][*] (the HBM-bound read of the cost
//            matrix), quantises, keeps cells whose gain = w_q - threshold_q > 0 in column order (wave
//            ballot + prefix popcount), records the row dual, and unions row and column in the
//            lock-free forest.  Rows that already hold a visual decision and excluded columns are
//            skipped (visual_sort/voting.rs:73-79).
//   stage 1  k_assign_label : label[q] = component representative (minimum row of the component).
//   stage 2  k_assign_next  : next_row[q] = next row of the same component, one wave per row, 64 labels
//            per probe — gives each component its rows in ascending order without a sort.
//   stage 3  k_assign_solve : one thread per component runs sa_assign_component.
//   stage 4  k_finalize     : winners -> (track id, VotingType).
// =====================================================================================================
__global__ __launch_bounds__(256) void k_assign_edges(const SceneDev* __restrict__ scenes, SaParams p) {
  const SceneDev& S = scenes[blockIdx.z];
  const uint32_t q = blockIdx.x * 4 + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (q >= S.N) return;
  if (S.row_has[q]) return;  // feature_winners.contains_key(from)
  const float* prow = S.pos + (size_t)q * S.T;
  uint32_t* ecol = S.e_col + (size_t)q * S.estride;
  int64_t* egain = S.e_gain + (size_t)q * S.estride;
  uint32_t cnt = 0;
  int64_t maxg = 0;
  for (uint32_t base = 0; base < S.T; base += WAVE) {
    uint32_t t = base + lane;
    int64_t gain = 0;
    if (t < S.T && !S.col_excluded[t]) {
      float w = prow[t];
      if (w == w) gain = sa_quantise(w) - p.threshold_q;
    }
    bool has = gain > 0;
    unsigned long long m = __ballot(has);
    if (has) {
      uint32_t off = cnt + __popcll(m & ((1ull << lane) - 1ull));
      ecol[off] = t;
      egain[off] = gain;
      if (gain > maxg) maxg = gain;
      sa_uf_union(S.parent, q, S.N + t);
    }
    cnt += __popcll(m);
  }
  for (int o = 32; o > 0; o >>= 1) {
    int64_t og = __shfl_xor(maxg, o);
    if (og > maxg) maxg = og;
  }
  if (lane == 0) {
    S.e_cnt[q] = cnt;
    S.u[q] = -maxg;
  }
}

__global__ void k_assign_label(const SceneDev* __restrict__ scenes) {
  const SceneDev& S = scenes[blockIdx.z];
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= S.N) return;
  S.label[q] = S.e_cnt[q] ? sa_uf_find(S.parent, q) : SA_NONE;
}

__global__ __launch_bounds__(256) void k_assign_next(const SceneDev* __restrict__ scenes) {
  const SceneDev& S = scenes[blockIdx.z];
  const uint32_t q = blockIdx.x * 4 + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (q >= S.N) return;
  const uint32_t lab = S.label[q];
  if (lab == SA_NONE) return;
  uint32_t found = SA_NONE;
  for (uint32_t base = q + 1; base < S.N; base += WAVE) {
    uint32_t r = base + lane;
    bool hit = r < S.N && S.label[r] == lab;
    unsigned long long m = __ballot(hit);
    if (m) { found = base + (uint32_t)__ffsll((long long)m) - 1u; break; }
  }
  if (lane == 0) S.next_row[q] = found;
}

__global__ void k_assign_solve(const SceneDev* __restrict__ scenes) {
  const SceneDev& S = scenes[blockIdx.z];
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= S.N) return;
  if (S.label[q] != q) return;  // only the representative (minimum row) of a component works
  sa_assign_ws w;
  w.e_cnt = S.e_cnt; w.e_col = S.e_col; w.e_gain = S.e_gain; w.estride = S.estride;
  w.next_row = S.next_row;
  w.u = S.u; w.v = S.v; w.rmatch = S.rmatch; w.cmatch = S.cmatch;
  w.dist = S.dist; w.pred = S.pred; w.cstamp = S.cstamp; w.cscan = S.cscan; w.cnext = S.cnext;
  w.rdist = S.rdist; w.rnext = S.rnext;
  sa_assign_component(w, q);
}

__global__ void k_finalize(const SceneDev* __restrict__ scenes) {
  const SceneDev& S = scenes[blockIdx.z];
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= S.N) return;
  uint64_t id = 0;
  uint8_t vt = SA_VOTE_NONE;
  int32_t vw = S.vis_winner[q];
  if (vw >= 0) { id = S.t_ids[vw]; vt = SA_VOTE_VISUAL; }
  else if (!S.row_has[q]) {
    int32_t c = S.rmatch[q];
    if (c >= 0) { id = S.t_ids[c]; vt = SA_VOTE_POSITIONAL; }
  }
  S.out_track_id[q] = id;
  S.out_vote[q] = vt;
}

// =====================================================================================================
// Launchers
// =====================================================================================================
static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

hipError_t sa_launch_prep_cands(const PrepCandArgs& a, const SaParams& p, hipStream_t st) {
  if (!a.n) return hipSuccess;
  hipLaunchKernelGGL(k_prep_cands, dim3(cdiv(a.n, 256)), dim3(256), 0, st, a, p);
  return hipGetLastError();
}
hipError_t sa_launch_prep_tracks(const PrepTrackArgs& a, const SaParams& p, hipStream_t st) {
  if (!a.n) return hipSuccess;
  hipLaunchKernelGGL(k_prep_tracks, dim3(cdiv(a.n, 256)), dim3(256), 0, st, a, p);
  return hipGetLastError();
}
hipError_t sa_launch_pad_features(const float* src, uint32_t rows, uint32_t D, uint32_t D8, uint32_t K,
                                  const uint32_t* slots, const uint8_t* present, float* dst, float* norms,
                                  uint8_t* dst_present, uint32_t* fcount, hipStream_t st) {
  if (!rows) return hipSuccess;
  hipLaunchKernelGGL(k_pad_features, dim3(cdiv(rows, 4)), dim3(256), 0, st, src, rows, D, D8, K, slots, present, dst,
                     norms, dst_present);
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
hipError_t sa_launch_frame_init(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxT, const SaParams&,
                                hipStream_t st) {
  hipLaunchKernelGGL(k_frame_init, dim3(cdiv(maxN + maxT + 1, 256), 1, ns), dim3(256), 0, st, scenes);
  return hipGetLastError();
}
hipError_t sa_launch_positional(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxT, const SaParams& p,
                                hipStream_t st) {
  if (!maxN || !maxT) return hipSuccess;
  hipLaunchKernelGGL(k_positional, dim3(cdiv(maxT, POS_TJ), cdiv(maxN, POS_TI), ns), dim3(256), 0, st, scenes, p);
  return hipGetLastError();
}
hipError_t sa_launch_quant_tap(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxT, hipStream_t st) {
  if (!maxN || !maxT) return hipSuccess;
  uint32_t blocks = cdiv((uint32_t)(((size_t)maxN * maxT + 255) / 256), 1);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_quant_tap, dim3(blocks, 1, ns), dim3(256), 0, st, scenes);
  return hipGetLastError();
}
hipError_t sa_launch_bestfit(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxT, const SaParams& p,
                             hipStream_t st, int stage) {
  if (!maxN || !maxT) return hipSuccess;
  switch (stage) {
    case 0: hipLaunchKernelGGL(k_bestfit_rows, dim3(cdiv(maxN, 4), 1, ns), dim3(256), 0, st, scenes, p); break;
    case 1: hipLaunchKernelGGL(k_bestfit_ties, dim3(cdiv(maxN, 4), 1, ns), dim3(256), 0, st, scenes, p); break;
    default: hipLaunchKernelGGL(k_bestfit_resolve, dim3(cdiv(maxN, 256), 1, ns), dim3(256), 0, st, scenes); break;
  }
  return hipGetLastError();
}
hipError_t sa_launch_assign(const SceneDev* scenes, uint32_t ns, uint32_t maxN, uint32_t maxT, const SaParams& p,
                            hipStream_t st, int stage) {
  if (!maxN) return hipSuccess;
  switch (stage) {
    case 0: if (maxT) hipLaunchKernelGGL(k_assign_edges, dim3(cdiv(maxN, 4), 1, ns), dim3(256), 0, st, scenes, p); break;
    case 1: hipLaunchKernelGGL(k_assign_label, dim3(cdiv(maxN, 256), 1, ns), dim3(256), 0, st, scenes); break;
    case 2: hipLaunchKernelGGL(k_assign_next, dim3(cdiv(maxN, 4), 1, ns), dim3(256), 0, st, scenes); break;
    case 3: hipLaunchKernelGGL(k_assign_solve, dim3(cdiv(maxN, 64), 1, ns), dim3(64), 0, st, scenes); break;
    default: hipLaunchKernelGGL(k_finalize, dim3(cdiv(maxN, 256), 1, ns), dim3(256), 0, st, scenes); break;
  }
  return hipGetLastError();
}
