// sa_upkeep.hip — device-side track upkeep: the step either side of the association.
//
// After the votes, Sort / VisualSort::predict turn every candidate into a new track or merge it into its winner
// (sort/simple_api.rs:164-190, visual_sort/simple_api.rs:189-226 -> Track::merge -> SortMetric / VisualMetric::optimize):
//   * Kalman step of the destination track with the candidate's box   kalman_prediction.rs:13-32, kalman_2d_box.rs:58-148
//   * the track's new predicted box becomes its row of the cost tables (geometry, polygon, Mahalanobis projection, epoch)
//   * VisualSORT: feature-bank policy optimize_observations             visual_sort/metric.rs:129-154, 297-374
// With host upkeep (sa_tracks_upsert) that state crosses PCIe twice per frame; here it never leaves HBM: the winners are
// already on the device (SceneDev::win_col), the candidates' boxes and padded features too.  Two small kernels, O(N).
// The arithmetic is the shared header sa_kalman.h — the same source the host facade compiles — so the Kalman state is
// bit-identical to the host path.  The polygon of an ORIENTED box needs cos / sin of its angle from the HOST's libm (the reference's
// f64::cos / sin resolve to it, and the device's math library differs from it in the last bit for about one angle in a thousand,
// which the bit-exact IoU gate would see): k_apply_kalman writes the polygon of boxes without an angle, and sa_tracks_apply — which
// has the predicted boxes on the host anyway — sends (row, cos, sin) of the others to k_apply_polygons, queued behind it.
#include "sa_engine.h"
#include "sa_frame.h"
#include "sa_kalman.h"

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Rank of candidate i among the candidates WITHOUT a winner (those that start a track), in candidate order: the row / id it takes when
// the engine draws them on the device (ApplyArgs::new_row == nullptr: sa_batch_run_apply).  One wave, a strided count over win_col[0, i).
__device__ __forceinline__ uint32_t sa_rank_of_new(const int32_t* __restrict__ win_col, uint32_t i, uint32_t lane) {
  uint32_t cnt = 0;
  for (uint32_t j = lane; j < i; j += 64u) cnt += win_col[j] < 0 ? 1u : 0u;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  return cnt;
}
#define SA_KF_SYNC()                                       \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
// One WAVEFRONT per candidate: the destination track's Kalman step (make_prediction: [initiate,] predict, update — sa_kalman.h) with the
// state in LDS and every stage's elements spread over the lanes.  Each ELEMENT is still computed by one lane with exactly the
// operations, and in exactly the order, of the serial code (sa_kf_predict / sa_kf_update: the loops below are those loops with the outer
// index replaced by the lane), so the state stays bit-identical to the host facade's and the oracle's — but a step is a dozen short
// stages instead of ~4 k dependent instructions of one thread on a 110-float state in scratch memory (1000 candidates: 16 wavefronts of
// single threads took ~25 us — the longest link of the facade's predict() after the association; in-kernel ids: no k_apply_ids launch).
struct KfLds {
  float mean[10], sd[10], innov[5], nm[10];
  float cov[100], mc[100], P[25], G[50], gtp[50];
};
__device__ __forceinline__ void kalman_block(const ApplyArgs& a, const SaParams& p, uint32_t blk, KfLds* s_kf) {
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const uint32_t i = blk * 4u + w;   // wave-uniform
  if (i >= a.n) return;
  KfLds& L = s_kf[w];
  const BoxRaw r = a.c_raw[i];
  const sa_box& cbox = r.box;
  const int32_t col = a.win_col[i];
  const bool merged = col >= 0;
  uint32_t row;
  uint64_t new_id = 0;
  if (merged) row = (uint32_t)col;
  else if (a.new_row) { row = a.new_row[i]; new_id = a.new_ids[i]; }
  else {
    const uint32_t rk = sa_rank_of_new(a.win_col, i, lane);
    row = a.T0 + rk;
    new_id = a.id_base + 1ull + (a.id_per_candidate ? (uint64_t)i : (uint64_t)rk);
  }
  float* st = a.kf + (size_t)row * 110;
  const float pw = p.kf_position_weight, vw = p.kf_velocity_weight;
  // ---- state in: the stored one, or sa_kf_initiate
  if (merged) {
    for (uint32_t e = lane; e < 110u; e += 64u) {
      const float v = st[e];
      if (e < 10u) L.mean[e] = v; else L.cov[e - 10u] = v;
    }
  } else {
    const float h = cbox.height;
    const float vp = 2.0f * pw * h, vv = 10.0f * vw * h;   // sa_kf_std_diag: k * w * p
    if (lane < 10u) {
      const float m0[5] = {cbox.xc, cbox.yc, sa_kf_opt_angle(cbox), cbox.aspect, cbox.height};
      L.mean[lane] = lane < 5u ? m0[lane] : 0.0f;
    }
    for (uint32_t e = lane; e < 100u; e += 64u) {
      const uint32_t ii = e / 10u, jj = e % 10u;
      const float sdv = ii < 5u ? (ii == 3u ? 1e-2f : vp) : (ii == 8u ? 1e-5f : vv);
      L.cov[e] = ii == jj ? sdv * sdv : 0.0f;
    }
  }
  SA_KF_SYNC();
  // ---- sa_kf_predict
  if (lane < 10u) {
    const float m4 = L.mean[4];
    const float vp = 1.0f * pw * m4, vv = 1.0f * vw * m4;
    L.sd[lane] = lane < 5u ? (lane == 3u ? 1e-2f : vp) : (lane == 8u ? 1e-5f : vv);
  }
  SA_KF_SYNC();
  if (lane < 5u) L.mean[lane] = L.mean[lane] + L.mean[lane + 5u];
  for (uint32_t e = lane; e < 100u; e += 64u) {
    const uint32_t ii = e / 10u;
    L.mc[e] = ii < 5u ? L.cov[e] + L.cov[e + 50u] : L.cov[e];
  }
  SA_KF_SYNC();
  for (uint32_t e = lane; e < 100u; e += 64u) {
    const uint32_t ii = e / 10u, jj = e % 10u;
    const float v = jj < 5u ? L.mc[e] + L.mc[e + 5u] : L.mc[e];
    L.cov[e] = v + (ii == jj ? L.sd[ii] * L.sd[ii] : 0.0f);
  }
  SA_KF_SYNC();
  // ---- sa_kf_update
  if (lane < 5u) {
    const float vp = 1.0f * pw * L.mean[4];
    L.sd[lane] = lane == 3u ? 1e-1f : vp;
  }
  SA_KF_SYNC();
  if (lane < 25u) {
    const uint32_t ii = lane / 5u, jj = lane % 5u;
    L.P[lane] = L.cov[ii * 10u + jj] + (ii == jj ? L.sd[ii] * L.sd[ii] : 0.0f);
  }
  if (lane < 50u) {
    const uint32_t rr = lane / 10u, cc = lane % 10u;
    L.G[lane] = L.cov[cc * 10u + rr];
  }
  if (lane < 5u) {
    const float z[5] = {cbox.xc, cbox.yc, sa_kf_opt_angle(cbox), cbox.aspect, cbox.height};
    L.innov[lane] = z[lane] - L.mean[lane];
  }
  SA_KF_SYNC();
  if (lane < 10u) {  // solve_lower_triangular on the un-factorised covariance, one column per lane
    const uint32_t c = lane;
    for (int ii = 0; ii < 5; ++ii) {
      const float coeff = L.G[ii * 10 + c] / L.P[ii * 5 + ii];
      L.G[ii * 10 + c] = coeff;
      const float nc = -coeff;
      for (int rr = ii + 1; rr < 5; ++rr) L.G[rr * 10 + c] = nc * L.P[rr * 5 + ii] + L.G[rr * 10 + c];
    }
  }
  SA_KF_SYNC();
  if (lane < 10u) {
    const uint32_t c = lane;
    float acc = L.innov[0] * L.G[c];
    for (int rr = 1; rr < 5; ++rr) acc = L.innov[rr] * L.G[rr * 10 + c] + acc;
    L.nm[c] = L.mean[c] + acc;
  }
  if (lane < 50u) {
    const uint32_t ii = lane / 5u, jj = lane % 5u;
    float acc = L.G[ii] * L.P[jj];
    for (int k = 1; k < 5; ++k) acc = L.G[k * 10 + ii] * L.P[k * 5 + jj] + acc;
    L.gtp[lane] = acc;
  }
  SA_KF_SYNC();
  for (uint32_t e = lane; e < 100u; e += 64u) {
    const uint32_t ii = e / 10u, jj = e % 10u;
    float acc = L.gtp[ii * 5u] * L.G[jj];
    for (int k = 1; k < 5; ++k) acc = L.gtp[ii * 5u + k] * L.G[k * 10 + jj] + acc;
    L.mc[e] = L.cov[e] - acc;   // (the new covariance)
  }
  SA_KF_SYNC();
  // ---- state out
  for (uint32_t e = lane; e < 110u; e += 64u) st[e] = e < 10u ? L.nm[e] : L.mc[e - 10u];
  if (lane != 0) return;
  // sa_kf_state_box + the observation's confidence; the track's row of the cost tables, exactly what k_prep_tracks derives from an upserted box
  sa_box pred;
  pred.xc = L.nm[0]; pred.yc = L.nm[1];
  pred.has_angle = L.nm[2] == 0.0f ? 0 : 1;
  pred.angle = L.nm[2];
  pred.aspect = L.nm[3]; pred.height = L.nm[4];
  pred.confidence = cbox.confidence;
  pred.reserved = 0;
  sa_geo g;
  g.xc = pred.xc;
  g.yc = pred.yc;
  g.r = sa_radius(pred.aspect, pred.height);
  g.hha = pred.height * pred.height * pred.aspect;
  a.geo[row] = g;
  a.ext[row] = sa_box_ext(pred.aspect, pred.height, pred.has_angle && pred.angle != 0.0f);
  // (a box with an angle: its polygon follows from the host's cos / sin, k_apply_polygons; until then — nothing reads the table in
  // between — the row holds the axis-aligned one)
  sa_vertices(pred.xc, pred.yc, pred.aspect, pred.height, 1.0, 0.0, a.verts + (size_t)row * 8);
  a.t_epoch[row] = a.epoch;
  if (!merged) a.t_ids[row] = new_id;
  float mean5[5], cov25[25];
  for (int x = 0; x < 5; ++x) {
    mean5[x] = L.nm[x];
    for (int y = 0; y < 5; ++y) cov25[x * 5 + y] = L.mc[x * 10 + y];
  }
  sa_maha_prepare(p.kf_position_weight, mean5, cov25, a.maha + (size_t)row * 20);
  a.out_pred[i] = pred;
}
__global__ __launch_bounds__(256) void k_apply_kalman(ApplyArgs a, SaParams p, uint32_t kf_blocks) {
  __shared__ KfLds s_kf[4];
  if (blockIdx.x < kf_blocks) { kalman_block(a, p, blockIdx.x, s_kf); return; }
  // (ApplyArgs::copy_src: one candidate's feature row out of the caller's block; rows are 16-byte aligned multiples of 32 floats)
  const uint32_t i = blockIdx.x - kf_blocks;
  const float4* src = (const float4*)(a.copy_src + (size_t)i * a.copy_row_floats);
  float4* dst = (float4*)(a.copy_dst + (size_t)i * a.copy_row_floats);
  for (uint32_t x = threadIdx.x; x < a.copy_row_floats / 4u; x += 256u) dst[x] = src[x];
}

// Polygons of the oriented boxes among the rows k_apply_kalman refreshed: cos / sin from the host's libm (sa_tracks_apply), geometry
// from the row itself (height and aspect recovered would not be exact: the host sends the box).
__global__ void k_apply_polygons(const SaPolyFix* __restrict__ fix, uint32_t n, double* __restrict__ verts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SaPolyFix f = fix[i];
  sa_vertices(f.xc, f.yc, f.aspect, f.height, f.c, f.s, verts + (size_t)f.row * 8);
}
hipError_t sa_launch_apply_polygons(const SaPolyFix* fix, uint32_t n, double* verts, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_apply_polygons, dim3(cdiv(n, 256)), dim3(256), 0, st, fix, n, verts);
  return hipGetLastError();
}

// One workgroup per candidate: the destination track's feature bank after optimize_observations.  The policy (which stored row ends up
// in which slot) is a few dozen scalar operations of thread 0; the rows then move IN REGISTERS: thread x holds element x of all K stored
// rows (K loads in flight together) and writes every slot's new content — one pass over the bank, no scratch copy and no second barrier
// (the first version went row by row through a per-candidate scratch buffer: two dependent passes, ~20 us at 1000 candidates x 3 x 512).
struct BankLds {
  uint8_t src[SA_MAX_BANK];
  float q[SA_MAX_BANK + 1], nrm[SA_MAX_BANK];  // q[K] = the new observation's quality (K may be SA_MAX_BANK)
  uint32_t newfeat;
};
template <int KMAX>
__device__ __forceinline__ void bank_block(const BankArgs& a, uint32_t i, BankLds& B) {
  if (i >= a.n) return;
  auto& s_src = B.src; auto& s_q = B.q; auto& s_nrm = B.nrm; uint32_t& s_newfeat = B.newfeat;
  const uint32_t K = a.K, Dp = a.Dp, tid = threadIdx.x;
  const int32_t col = a.win_col[i];
  const bool merged = col >= 0;
  uint32_t row;
  if (merged) row = (uint32_t)col;
  else if (a.new_row) row = a.new_row[i];
  else {  // (rows drawn on the device: table rows T0 + rank among the candidates that start a track — every wave counts the same)
    row = a.T0 + sa_rank_of_new(a.win_col, i, tid & 63u);
  }
  float* bank = a.t_feat + (size_t)row * K * Dp;
  if (tid == 0) {
    const sa_box& b = a.c_raw[i].box;
    const bool has = a.c_feat && (!a.c_fpresent_in || a.c_fpresent_in[i]);
    const float q = a.c_quality ? a.c_quality[i] : 1.0f;
    float own = 0.0f;
    bool has_own = false;
    if (a.c_own) { own = a.c_own[i]; has_own = own == own; }
    // a merge keeps the feature only when it may be COLLECTED (visual_sort/metric.rs:337-349); a new track keeps its
    // observation as it came (is_merge == false)
    bool keep = has;
    if (merged) keep = has && sa_feature_can_be_used(a.minimal_area, b, q, a.q_collect, has_own, own, a.own_collect);
    s_newfeat = keep ? 1u : 0u;
    uint8_t present[SA_MAX_BANK];
    float quality[SA_MAX_BANK];
    for (uint32_t k = 0; k < K; ++k) {
      present[k] = merged ? a.t_fpresent[(size_t)row * K + k] : 0;
      quality[k] = merged ? a.t_fquality[(size_t)row * K + k] : 0.0f;
      s_q[k] = quality[k];
      s_nrm[k] = merged ? a.t_fnorm[(size_t)row * K + k] : 0.0f;
    }
    if (merged) sa_bank_policy(K, K, present, quality, s_src);
    else {
      s_src[0] = SA_BANK_NEW;
      for (uint32_t k = 1; k < K; ++k) s_src[k] = SA_BANK_NONE;
    }
    s_q[K] = q;  // quality of the new observation
  }
  __syncthreads();
  const bool new_pres = s_newfeat != 0;
  const float* cf = a.c_feat ? a.c_feat + (size_t)i * Dp : nullptr;
  // The squared norm of the new row.  A frame that carried its preparation blocks left it in c_fnorm; a LEAN frame did not (its rows
  // are the uploaded ones, D == Dp, and nothing else of the candidates' half is read here): the block's first wave forms it the way the
  // preparation block would have — pad_feature_row, same lanes, same order: bit for bit the value the next frame's cells are scaled by.
  float nrm_new = 0.0f;
  if (a.c_fnorm) nrm_new = new_pres ? a.c_fnorm[i] : 0.0f;
  else if (new_pres && cf && tid < WAVE) pad_feature_row(cf, const_cast<float*>(cf), Dp, Dp, true, tid, &nrm_new);
  uint32_t srcs[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) srcs[k] = (uint32_t)k < K ? s_src[k] : SA_BANK_NONE;
  for (uint32_t x = tid; x < Dp; x += 256) {
    float v[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) v[k] = (merged && (uint32_t)k < K) ? bank[(size_t)k * Dp + x] : 0.0f;   // (a new track's row: nothing stored yet)
    const float nv = (new_pres && cf) ? cf[x] : 0.0f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if ((uint32_t)k >= K) continue;
      const uint32_t src = srcs[k];
      if (src == (uint32_t)k) continue;   // the row stays where it is
      float out = 0.0f;                   // SA_BANK_NONE: an empty slot
      if (src == SA_BANK_NEW) out = nv;
      else {
#pragma unroll
        for (int j = 0; j < KMAX; ++j) out = src == (uint32_t)j ? v[j] : out;
      }
      bank[(size_t)k * Dp + x] = out;
      if (a.t_ffrag) a.t_ffrag[sa_frag_index(row * K + (uint32_t)k, x, Dp)] = out;   // the bank's fragment-order twin
    }
  }
  if (tid == 0) {
    uint32_t count = 0;
    for (uint32_t k = 0; k < K; ++k) {
      const uint32_t src = s_src[k];
      bool pres = false;
      float q = 0.0f, nrm = 0.0f;
      if (src == SA_BANK_NEW) { pres = new_pres; q = s_q[K]; nrm = pres ? nrm_new : 0.0f; }
      else if (src < SA_BANK_NEW) { pres = true; q = s_q[src]; nrm = s_nrm[src]; }
      a.t_fpresent[(size_t)row * K + k] = pres ? 1 : 0;
      a.t_fquality[(size_t)row * K + k] = q;
      a.t_fnorm[(size_t)row * K + k] = nrm;
      count += pres ? 1u : 0u;
    }
    a.t_fcount[row] = count;
  }
}
// The whole upkeep of a VisualSORT frame in ONE launch: blocks [0, kf_blocks) take the Kalman steps (four candidates each, a wavefront
// per candidate), the n blocks behind them the feature banks.  The two halves share nothing but the winners they read (the Kalman half
// writes states, boxes, table rows; the bank half feature rows and their bookkeeping), so they run side by side — one dependent launch
// (~4 us) and a serialisation less.  (Round 3 measured the merge with the one-thread-per-track Kalman step and dropped it: that code's
// 110-float state in scratch made every block of the merged kernel slow; the wave-parallel step keeps its state in 1.4 KB of LDS.)
template <int KMAX>
__global__ __launch_bounds__(256) void k_apply_visual(ApplyArgs a, BankArgs b, SaParams p, uint32_t kf_blocks) {
  __shared__ union { KfLds kf[4]; BankLds bank; } s_u;
  if (blockIdx.x < kf_blocks) kalman_block(a, p, blockIdx.x, s_u.kf);
  else bank_block<KMAX>(b, blockIdx.x - kf_blocks, s_u.bank);
}
template <int KMAX>
static void launch_visual_apply(const ApplyArgs& a, const BankArgs& b, const SaParams& p, hipStream_t st, hipEvent_t done) {
  const uint32_t kfb = cdiv(a.n, 4);
  if (done) hipExtLaunchKernelGGL(k_apply_visual<KMAX>, dim3(kfb + b.n), dim3(256), 0, st, nullptr, done, 0, a, b, p, kfb);
  else hipLaunchKernelGGL(k_apply_visual<KMAX>, dim3(kfb + b.n), dim3(256), 0, st, a, b, p, kfb);
}

// done (optional): the step's LAST dispatch carries it as its own completion signal — a caller that waits for the step waits for the
// event instead of synchronising the stream (a marker packet and its round trip through the command processor: ~10 us)
// part: 0 = the whole step (VisualSORT: Kalman and bank blocks in ONE launch); 1 = the Kalman half alone, 2 = the bank half alone — a
// caller that hands the predicted boxes out as soon as they exist (the tracker facade) waits for the first and lets the second run on
hipError_t sa_launch_apply(const ApplyArgs& a, const BankArgs* b, const SaParams& p, hipStream_t st, hipEvent_t done, int part) {
  if (!a.n) return hipSuccess;
  if (part == 2) {
    if (!b) return hipSuccess;
    ApplyArgs none = a;
    if (b->K <= 4) { if (done) hipExtLaunchKernelGGL(k_apply_visual<4>, dim3(b->n), dim3(256), 0, st, nullptr, done, 0, none, *b, p, 0u); else hipLaunchKernelGGL(k_apply_visual<4>, dim3(b->n), dim3(256), 0, st, none, *b, p, 0u); }
    else if (b->K <= 8) { if (done) hipExtLaunchKernelGGL(k_apply_visual<8>, dim3(b->n), dim3(256), 0, st, nullptr, done, 0, none, *b, p, 0u); else hipLaunchKernelGGL(k_apply_visual<8>, dim3(b->n), dim3(256), 0, st, none, *b, p, 0u); }
    else { if (done) hipExtLaunchKernelGGL(k_apply_visual<SA_MAX_BANK>, dim3(b->n), dim3(256), 0, st, nullptr, done, 0, none, *b, p, 0u); else hipLaunchKernelGGL(k_apply_visual<SA_MAX_BANK>, dim3(b->n), dim3(256), 0, st, none, *b, p, 0u); }
    return hipGetLastError();
  }
  if (!b || part == 1) {
    const uint32_t kfb = cdiv(a.n, 4), grid = kfb + (a.copy_src ? a.n : 0u);
    if (done) hipExtLaunchKernelGGL(k_apply_kalman, dim3(grid), dim3(256), 0, st, nullptr, done, 0, a, p, kfb);
    else hipLaunchKernelGGL(k_apply_kalman, dim3(grid), dim3(256), 0, st, a, p, kfb);
    return hipGetLastError();
  }
  if (b->K <= 4) launch_visual_apply<4>(a, *b, p, st, done);
  else if (b->K <= 8) launch_visual_apply<8>(a, *b, p, st, done);
  else launch_visual_apply<SA_MAX_BANK>(a, *b, p, st, done);
  return hipGetLastError();
}

// ---- the same two halves for a whole request set: blockIdx.y = scene, the scene's arguments read from the set's array (wave-uniform
// address: scalar loads).  Scenes of different sizes share the grid; a block beyond its scene's count leaves at once.
__global__ __launch_bounds__(256) void k_apply_kalman_set(const ApplyScene* __restrict__ scenes, SaParams p) {
  __shared__ KfLds s_kf[4];
  const ApplyArgs a = scenes[blockIdx.y].a;
  const uint32_t kfb = (a.n + 3u) / 4u;
  if (blockIdx.x < kfb) { kalman_block(a, p, blockIdx.x, s_kf); return; }
  const uint32_t i = blockIdx.x - kfb;
  if (!a.copy_src || i >= a.n) return;
  const float4* src = (const float4*)(a.copy_src + (size_t)i * a.copy_row_floats);
  float4* dst = (float4*)(a.copy_dst + (size_t)i * a.copy_row_floats);
  for (uint32_t x = threadIdx.x; x < a.copy_row_floats / 4u; x += 256u) dst[x] = src[x];
}
template <int KMAX>
__global__ __launch_bounds__(256) void k_apply_bank_set(const ApplyScene* __restrict__ scenes) {
  __shared__ BankLds s_bank;
  const BankArgs b = scenes[blockIdx.y].b;
  bank_block<KMAX>(b, blockIdx.x, s_bank);
}
hipError_t sa_launch_apply_set(const ApplyScene* scenes, uint32_t n_scenes, uint32_t max_blocks, uint32_t K, const SaParams& p, hipStream_t st,
                               hipEvent_t done, int part) {
  if (!n_scenes || !max_blocks) return hipSuccess;
  const dim3 grid(max_blocks, n_scenes), block(256);
#define SA_SET_LAUNCH(kern, ...)                                                                     \
  do {                                                                                               \
    if (done) hipExtLaunchKernelGGL(kern, grid, block, 0, st, nullptr, done, 0, __VA_ARGS__);        \
    else hipLaunchKernelGGL(kern, grid, block, 0, st, __VA_ARGS__);                                  \
  } while (0)
  if (part == 2) {
    if (K <= 4) SA_SET_LAUNCH(k_apply_bank_set<4>, scenes);
    else if (K <= 8) SA_SET_LAUNCH(k_apply_bank_set<8>, scenes);
    else SA_SET_LAUNCH(k_apply_bank_set<SA_MAX_BANK>, scenes);
  } else SA_SET_LAUNCH(k_apply_kalman_set, scenes, p);
#undef SA_SET_LAUNCH
  return hipGetLastError();
}

// =====================================================================================================
// Non-maximum suppression (src/utils/nms.rs:32-72) on the clip machinery of the positional tiles.
// The host filters and rank-sorts the boxes (O(N log N), and it owns libm's cos/sin); the O(N^2) part runs here:
//   k_nms_mask  : bit (i, j), i < j in rank order, = intersection(box_i, box_j) as f32 / area(box_j) > threshold.  One block =
//                 16 rows x 64 columns = exactly one 64-bit word of 16 mask rows, so no global atomics: pairs are pruned with
//                 too_far() (bbox.rs:452-462, what Universal2DBox::intersection does first), the survivors are compacted in
//                 LDS and clipped by 64 worker lanes (f64 Sutherland–Hodgman + shoelace, clipping.rs:12-91).
//   k_nms_sweep : the greedy pass of the reference — a box that is still alive suppresses the boxes its row marks — by one
//                 wave that keeps the "removed" bitmap in registers and streams the mask rows 16 at a time.
// =====================================================================================================
__global__ __launch_bounds__(256) void k_nms_mask(const BoxRaw* __restrict__ raw, uint32_t n, uint32_t W, float thr,
                                                  uint64_t* __restrict__ mask) {
  const uint32_t i0 = blockIdx.y * 16, j0 = blockIdx.x * 64;
  if (i0 >= n || j0 >= n) return;
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  __shared__ unsigned long long s_bits[16];
  if (j0 + 63 <= i0) {  // the whole word lies on or below the diagonal: no pair with i < j
    if (tid < 16 && i0 + tid < n) mask[(size_t)(i0 + tid) * W + blockIdx.x] = 0ull;
    return;
  }
  __shared__ sa_geo s_rg[16], s_cg[64];
  __shared__ double s_rv[16][8], s_cv[64][8];
  __shared__ float s_carea[64];
  __shared__ uint16_t s_list[16 * 64];
  __shared__ uint32_t s_cnt;
  __shared__ double s_poly[4 * SA_POLY_CAP * 64];
  if (tid < 16) {
    s_bits[tid] = 0ull;
    const uint32_t i = i0 + tid;
    if (i < n) {
      const BoxRaw r = raw[i];
      sa_geo g;
      g.xc = r.box.xc; g.yc = r.box.yc; g.r = sa_radius(r.box.aspect, r.box.height); g.hha = 0.f;
      s_rg[tid] = g;
      sa_vertices(r.box.xc, r.box.yc, r.box.aspect, r.box.height, r.c, r.s, s_rv[tid]);
    }
  } else if (tid >= 64 && tid < 128) {
    const uint32_t lj = tid - 64, j = j0 + lj;
    if (j < n) {
      const BoxRaw r = raw[j];
      sa_geo g;
      g.xc = r.box.xc; g.yc = r.box.yc; g.r = sa_radius(r.box.aspect, r.box.height); g.hha = 0.f;
      s_cg[lj] = g;
      sa_vertices(r.box.xc, r.box.yc, r.box.aspect, r.box.height, r.c, r.s, s_cv[lj]);
      s_carea[lj] = sa_area(r.box.aspect, r.box.height);
    }
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t li = wave * 4 + r, i = i0 + li, j = j0 + lane;
    if (i < n && j < n && i < j) {
      if (!sa_too_far(s_rg[li], s_cg[lane])) {
        const uint32_t slot = atomicAdd(&s_cnt, 1u);
        s_list[slot] = (uint16_t)((li << 8) | lane);
      } else if (0.0f > thr) {  // intersection() is 0.0 for far boxes: 0 / area still beats a negative threshold
        atomicOr(&s_bits[li], 1ull << lane);
      }
    }
  }
  __syncthreads();
  const uint32_t cnt = s_cnt;
  if (tid < 64) {
    double* ws = s_poly + tid;
    for (uint32_t sidx = tid; sidx < cnt; sidx += 64) {
      const uint32_t c = s_list[sidx], li = c >> 8, lj = c & 255u;
      double sv[8], cv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { sv[k] = s_rv[li][k]; cv[k] = s_cv[lj][k]; }
      const double inter = sa_clip_area_ws(sv, cv, ws, ws + SA_POLY_CAP * 64, ws + 2 * SA_POLY_CAP * 64, ws + 3 * SA_POLY_CAP * 64, 64);
      const float metric = (float)inter / s_carea[lj];
      if (metric > thr) atomicOr(&s_bits[li], 1ull << lj);
    }
  }
  __syncthreads();
  if (tid < 16 && i0 + tid < n) mask[(size_t)(i0 + tid) * W + blockIdx.x] = s_bits[tid];
}

template <int S>  // S = words of the removed-bitmap per lane: up to 64 * 64 * S boxes
__global__ __launch_bounds__(64) void k_nms_sweep(const uint64_t* __restrict__ mask, uint32_t n, uint32_t W, uint8_t* __restrict__ keep) {
  const uint32_t lane = threadIdx.x;
  uint64_t rw[S];
#pragma unroll
  for (int s = 0; s < S; ++s) rw[s] = 0ull;
  constexpr int B = 16;
  for (uint32_t i0 = 0; i0 < n; i0 += B) {
    uint64_t rows[B][S];
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const uint32_t w = s * 64 + lane;
        rows[r][s] = (i0 + r < n && w < W) ? mask[(size_t)(i0 + r) * W + w] : 0ull;
      }
#pragma unroll
    for (int r = 0; r < B; ++r) {
      const uint32_t i = i0 + r;
      if (i >= n) break;
      const uint32_t word = i >> 6, owner = word & 63u, slot = word >> 6, bit = i & 63u;
      uint64_t mine = rw[0];
#pragma unroll
      for (int s = 1; s < S; ++s) mine = slot == (uint32_t)s ? rw[s] : mine;
      const uint64_t cur = __shfl(mine, (int)owner);
      const bool removed = (cur >> bit) & 1ull;
      if (!removed) {
#pragma unroll
        for (int s = 0; s < S; ++s) rw[s] |= rows[r][s];
      }
      if (lane == 0) keep[i] = removed ? 0 : 1;
    }
  }
}

// =====================================================================================================
// Exclusively owned areas  (src/utils/clipping/bbox_own_areas.rs:8-46 — SURVEY §8f rank 4)
//   share_i = area(box_i minus every other box that is not too_far()) / (area_i + EPS), clamped to 1 — what VisualSORT's
//   own-area gates read (visual_sort/simple_api.rs:111-127).  One wave per box:
//     1. the lanes scan all boxes: too_far() (the reference's neighbour rule), then a separating-axis test that drops the
//        circle-neighbours that do not actually overlap; the survivors' polygons go to LDS, relative to the box's centre;
//     2. one lane per polygon edge (4 per neighbour + 4 own) integrates its share of the owned region's boundary
//        (sa_own_edge, sa_device.h); a wave reduction and the reference's f32 epilogue finish the box.
//   status: bit 0 = a box overlaps more than SA_OWN_MAXNB others, bit 1 = more than SA_OWN_CAP disjoint stretches of one edge are covered.
// =====================================================================================================
#define SA_OWN_MAXNB 127
#define SA_OWN_CAP 24
__global__ __launch_bounds__(64) void k_own_area(const BoxRaw* __restrict__ raw, uint32_t n, float* __restrict__ share,
                                                 uint32_t* __restrict__ status) {
  const uint32_t i = blockIdx.x, lane = threadIdx.x;
  __shared__ double s_poly[(SA_OWN_MAXNB + 1) * 8];
  __shared__ double s_iva[SA_OWN_CAP * 64], s_ivb[SA_OWN_CAP * 64];
  __shared__ uint32_t s_cnt;
  const BoxRaw me = raw[i];
  sa_geo mg;
  mg.xc = me.box.xc; mg.yc = me.box.yc; mg.r = sa_radius(me.box.aspect, me.box.height); mg.hha = 0.f;
  double mv[8];
  sa_vertices(me.box.xc, me.box.yc, me.box.aspect, me.box.height, me.c, me.s, mv);
  const double ox = (double)me.box.xc, oy = (double)me.box.yc;
  if (lane < 8) s_poly[lane] = mv[lane] - ((lane & 1u) ? oy : ox);
  if (lane == 0) s_cnt = 1;
  __syncthreads();
  bool overflow = false;
  for (uint32_t j0 = 0; j0 < n; j0 += 64) {
    const uint32_t j = j0 + lane;
    if (j < n && j != i) {
      const BoxRaw r = raw[j];
      sa_geo g;
      g.xc = r.box.xc; g.yc = r.box.yc; g.r = sa_radius(r.box.aspect, r.box.height); g.hha = 0.f;
      if (!sa_too_far(mg, g)) {
        double v[8];
        sa_vertices(r.box.xc, r.box.yc, r.box.aspect, r.box.height, r.c, r.s, v);
        if (!sa_quads_separated(mv, v)) {
          const uint32_t slot = atomicAdd(&s_cnt, 1u);
          if (slot <= SA_OWN_MAXNB) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s_poly[slot * 8 + k] = v[k] - ((k & 1) ? oy : ox);
          } else {
            overflow = true;
          }
        }
      }
    }
  }
  __syncthreads();
  if (__any(overflow)) {
    if (lane == 0) { share[i] = NAN; atomicOr(status, 1u); }
    return;
  }
  const uint32_t m1 = s_cnt;
  double acc = 0.0;
  for (uint32_t e = lane; e < m1 * 4; e += 64)
    acc += sa_own_edge(s_poly, m1, e >> 2, e & 3u, s_iva + lane, s_ivb + lane, 64, SA_OWN_CAP);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    if (acc != acc) atomicOr(status, 2u);
    const double own = fabs(acc) * 0.5;
    const float e = (float)(own / (double)(sa_area(me.box.aspect, me.box.height) + SA_EPS));
    share[i] = acc != acc ? NAN : (e >= 1.0f ? 1.0f : e);  // NaN = "not done here": the spill path takes the box
  }
}

// Spill path for the boxes k_own_area gave up on (more than SA_OWN_MAXNB overlapping neighbours, or more than SA_OWN_CAP disjoint
// covered stretches on one edge: a box in the middle of a dense crowd).  The reference has no such limit
// (bbox_own_areas.rs:8-46), so neither does the engine: the same boundary integral with the neighbour polygons and the interval
// storage in HBM scratch — one wave per listed box, room for every other box of the frame as a neighbour and SA_OWN_BIGCAP disjoint
// stretches per edge (status bit 2 beyond that).  Slower per box (the lists live in L2, not LDS), and rare.
#define SA_OWN_BIGCAP 512
__global__ __launch_bounds__(64) void k_own_area_big(const BoxRaw* __restrict__ raw, uint32_t n, const uint32_t* __restrict__ list,
                                                     float* __restrict__ share, uint32_t* __restrict__ status, double* __restrict__ polys,
                                                     double* __restrict__ iv) {
  const uint32_t i = list[blockIdx.x], lane = threadIdx.x;
  double* s_poly = polys + (size_t)blockIdx.x * (size_t)(n + 1) * 8;
  double* s_iva = iv + (size_t)blockIdx.x * 2 * SA_OWN_BIGCAP * 64;
  double* s_ivb = s_iva + (size_t)SA_OWN_BIGCAP * 64;
  __shared__ uint32_t s_cnt;
  const BoxRaw me = raw[i];
  sa_geo mg;
  mg.xc = me.box.xc; mg.yc = me.box.yc; mg.r = sa_radius(me.box.aspect, me.box.height); mg.hha = 0.f;
  double mv[8];
  sa_vertices(me.box.xc, me.box.yc, me.box.aspect, me.box.height, me.c, me.s, mv);
  const double ox = (double)me.box.xc, oy = (double)me.box.yc;
  if (lane < 8) s_poly[lane] = mv[lane] - ((lane & 1u) ? oy : ox);
  if (lane == 0) s_cnt = 1;
  __syncthreads();
  for (uint32_t j0 = 0; j0 < n; j0 += 64) {
    const uint32_t j = j0 + lane;
    if (j < n && j != i) {
      const BoxRaw r = raw[j];
      sa_geo g;
      g.xc = r.box.xc; g.yc = r.box.yc; g.r = sa_radius(r.box.aspect, r.box.height); g.hha = 0.f;
      if (!sa_too_far(mg, g)) {
        double v[8];
        sa_vertices(r.box.xc, r.box.yc, r.box.aspect, r.box.height, r.c, r.s, v);
        if (!sa_quads_separated(mv, v)) {
          const uint32_t slot = atomicAdd(&s_cnt, 1u);  // at most n - 1 neighbours: always room
#pragma unroll
          for (int k = 0; k < 8; ++k) s_poly[(size_t)slot * 8 + k] = v[k] - ((k & 1) ? oy : ox);
        }
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  const uint32_t m1 = s_cnt;
  double acc = 0.0;
  for (uint32_t e = lane; e < m1 * 4; e += 64)
    acc += sa_own_edge(s_poly, m1, e >> 2, e & 3u, s_iva + lane, s_ivb + lane, 64, SA_OWN_BIGCAP);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    if (acc != acc) atomicOr(status, 4u);
    const double own = fabs(acc) * 0.5;
    const float e = (float)(own / (double)(sa_area(me.box.aspect, me.box.height) + SA_EPS));
    share[i] = acc != acc ? NAN : (e >= 1.0f ? 1.0f : e);
  }
}
size_t sa_own_big_scratch_bytes(uint32_t n, uint32_t batch) {
  return (size_t)batch * ((size_t)(n + 1) * 8 + 2 * (size_t)SA_OWN_BIGCAP * 64) * sizeof(double);
}
hipError_t sa_launch_own_areas_big(const BoxRaw* raw, uint32_t n, const uint32_t* list, uint32_t count, float* share, uint32_t* status,
                                   void* scratch, hipStream_t st) {
  if (!count) return hipSuccess;
  double* polys = (double*)scratch;
  double* iv = polys + (size_t)count * (size_t)(n + 1) * 8;
  hipLaunchKernelGGL(k_own_area_big, dim3(count), dim3(64), 0, st, raw, n, list, share, status, polys, iv);
  return hipGetLastError();
}

hipError_t sa_launch_own_areas(const BoxRaw* raw, uint32_t n, float* share, uint32_t* status, hipStream_t st) {
  if (!n) return hipSuccess;
  hipError_t e = hipMemsetAsync(status, 0, 4, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_own_area, dim3(n), dim3(64), 0, st, raw, n, share, status);
  return hipGetLastError();
}

hipError_t sa_launch_nms(const BoxRaw* raw, uint32_t n, float thr, uint64_t* mask, uint8_t* keep, hipStream_t st) {
  if (!n) return hipSuccess;
  const uint32_t W = cdiv(n, 64);
  hipLaunchKernelGGL(k_nms_mask, dim3(W, cdiv(n, 16)), dim3(256), 0, st, raw, n, W, thr, mask);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const uint32_t S = cdiv(W, 64);
  if (S <= 1) hipLaunchKernelGGL(k_nms_sweep<1>, dim3(1), dim3(64), 0, st, mask, n, W, keep);
  else if (S <= 2) hipLaunchKernelGGL(k_nms_sweep<2>, dim3(1), dim3(64), 0, st, mask, n, W, keep);
  else if (S <= 4) hipLaunchKernelGGL(k_nms_sweep<4>, dim3(1), dim3(64), 0, st, mask, n, W, keep);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
