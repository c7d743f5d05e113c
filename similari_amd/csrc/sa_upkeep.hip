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
// bit-identical to the host path; the one exception is the polygon of an ORIENTED box, whose cos/sin come from the device's
// math library instead of the host's libm (<= 1 ulp apart; axis-aligned boxes are unaffected).
#include "sa_engine.h"
#include "sa_kalman.h"

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

__global__ __launch_bounds__(64) void k_apply_kalman(ApplyArgs a, SaParams p) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const BoxRaw r = a.c_raw[i];
  const sa_box& cbox = r.box;
  const int32_t col = a.win_col[i];
  const bool merged = col >= 0;
  const uint32_t row = merged ? (uint32_t)col : a.new_row[i];
  sa_kf s;
  float* st = a.kf + (size_t)row * 110;
  if (merged) {
    for (int k = 0; k < 10; ++k) s.mean[k] = st[k];
    for (int k = 0; k < 100; ++k) s.cov[k] = st[10 + k];
  }
  const sa_box pred = sa_kf_make_prediction(p.kf_position_weight, p.kf_velocity_weight, merged, s, cbox);
  for (int k = 0; k < 10; ++k) st[k] = s.mean[k];
  for (int k = 0; k < 100; ++k) st[10 + k] = s.cov[k];
  // the track's row of the cost tables, exactly what k_prep_tracks derives from an upserted box
  sa_geo g;
  g.xc = pred.xc;
  g.yc = pred.yc;
  g.r = sa_radius(pred.aspect, pred.height);
  g.hha = pred.height * pred.height * pred.aspect;
  a.geo[row] = g;
  double c = 1.0, sn = 0.0;
  const double ang = (double)(pred.has_angle ? pred.angle : 0.0f);
  if (ang != 0.0) sincos(ang, &sn, &c);
  sa_vertices(pred.xc, pred.yc, pred.aspect, pred.height, c, sn, a.verts + (size_t)row * 8);
  a.t_epoch[row] = a.epoch;
  if (!merged) a.t_ids[row] = a.new_ids[i];
  float mean5[5], cov25[25];
  for (int x = 0; x < 5; ++x) {
    mean5[x] = s.mean[x];
    for (int y = 0; y < 5; ++y) cov25[x * 5 + y] = s.cov[x * 10 + y];
  }
  sa_maha_prepare(p.kf_position_weight, mean5, cov25, a.maha + (size_t)row * 20);
  a.out_pred[i] = pred;
}

// One workgroup per candidate: the destination track's feature bank after optimize_observations.
__global__ __launch_bounds__(256) void k_apply_bank(BankArgs a) {
  const uint32_t i = blockIdx.x;
  if (i >= a.n) return;
  __shared__ uint8_t s_src[SA_MAX_BANK];
  __shared__ float s_q[SA_MAX_BANK], s_nrm[SA_MAX_BANK];
  __shared__ uint32_t s_newfeat;
  const uint32_t K = a.K, Dp = a.Dp, tid = threadIdx.x;
  const int32_t col = a.win_col[i];
  const bool merged = col >= 0;
  const uint32_t row = merged ? (uint32_t)col : a.new_row[i];
  float* bank = a.t_feat + (size_t)row * K * Dp;
  float* tmp = a.tmp + (size_t)i * K * Dp;
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
  // stored rows that move: through this candidate's scratch (the permutation is in place)
  for (uint32_t k = 0; k < K; ++k) {
    const uint32_t src = s_src[k];
    if (src < SA_BANK_NEW && src != k)
      for (uint32_t x = tid; x < Dp; x += 256) tmp[(size_t)k * Dp + x] = bank[(size_t)src * Dp + x];
  }
  __syncthreads();
  uint32_t count = 0;
  for (uint32_t k = 0; k < K; ++k) {
    const uint32_t src = s_src[k];
    float* dst = bank + (size_t)k * Dp;
    bool pres = false;
    float q = 0.0f, nrm = 0.0f;
    if (src == SA_BANK_NEW) {
      pres = s_newfeat != 0;
      q = s_q[K];
      nrm = pres ? a.c_fnorm[i] : 0.0f;
      const float* cf = a.c_feat + (size_t)i * Dp;
      for (uint32_t x = tid; x < Dp; x += 256) dst[x] = pres ? cf[x] : 0.0f;
    } else if (src < SA_BANK_NEW) {
      pres = true;
      q = s_q[src];
      nrm = s_nrm[src];
      if (src != k)
        for (uint32_t x = tid; x < Dp; x += 256) dst[x] = tmp[(size_t)k * Dp + x];
    } else {
      for (uint32_t x = tid; x < Dp; x += 256) dst[x] = 0.0f;
    }
    if (tid == 0) {
      a.t_fpresent[(size_t)row * K + k] = pres ? 1 : 0;
      a.t_fquality[(size_t)row * K + k] = q;
      a.t_fnorm[(size_t)row * K + k] = nrm;
    }
    count += pres ? 1u : 0u;
  }
  if (tid == 0) a.t_fcount[row] = count;
}

hipError_t sa_launch_apply(const ApplyArgs& a, const BankArgs* b, const SaParams& p, hipStream_t st) {
  if (!a.n) return hipSuccess;
  hipLaunchKernelGGL(k_apply_kalman, dim3(cdiv(a.n, 64)), dim3(64), 0, st, a, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !b) return e;
  hipLaunchKernelGGL(k_apply_bank, dim3(a.n), dim3(256), 0, st, *b);
  return hipGetLastError();
}
