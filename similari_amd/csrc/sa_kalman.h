// sa_kalman.h — the 10-state / 5-measurement box Kalman filter of the reference, written once for the host facade
// (sa_tracker.cpp) and for the device-side track upkeep (sa_upkeep.hip).
//
//   Universal2DBoxKalmanFilter::{initiate, predict, update}     src/utils/kalman/kalman_2d_box.rs:58-148
//   TryFrom<KalmanState> for Universal2DBox                      src/utils/kalman.rs:72-92
//   make_prediction                                              src/trackers/kalman_prediction.rs:13-32
//
// Written against the filter's structure (motion = I + shift, update matrix = [I 0]): skipping the multiplications by the
// constant 0 / 1 entries of those matrices leaves every f32 result unchanged, and every remaining operation keeps the
// reference's order (nalgebra's column-major gemm / solve order), so host and device produce the same bits as the oracle's
// dense restatement.  Build with -ffp-contract=off (rustc never fuses a*b+c).
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/similari_assoc.h"
#include "sa_device.h"

struct sa_kf {
  float mean[10];
  float cov[100];
};

SA_HD float sa_kf_opt_angle(const sa_box& b) { return b.has_angle ? b.angle : 0.0f; }

SA_HD void sa_kf_std_diag(float w, float k, float cnst, float p, float* out5) {  // std_position / std_velocity, squared later
  float v = k * w * p;
  out5[0] = v; out5[1] = v; out5[2] = v; out5[3] = cnst; out5[4] = v;
}

SA_HD void sa_kf_initiate(float pw, float vw, const sa_box& b, sa_kf& s) {  // kalman_2d_box.rs:58-83
  s.mean[0] = b.xc; s.mean[1] = b.yc; s.mean[2] = sa_kf_opt_angle(b); s.mean[3] = b.aspect; s.mean[4] = b.height;
  for (int i = 5; i < 10; ++i) s.mean[i] = 0.0f;
  float sd[10];
  sa_kf_std_diag(pw, 2.0f, 1e-2f, b.height, sd);
  sa_kf_std_diag(vw, 10.0f, 1e-5f, b.height, sd + 5);
  for (int i = 0; i < 100; ++i) s.cov[i] = 0.0f;
  for (int i = 0; i < 10; ++i) s.cov[i * 10 + i] = sd[i] * sd[i];
}

SA_HD void sa_kf_predict(float pw, float vw, sa_kf& s) {  // kalman_2d_box.rs:87-102
  float sd[10];
  sa_kf_std_diag(pw, 1.0f, 1e-2f, s.mean[4], sd);
  sa_kf_std_diag(vw, 1.0f, 1e-5f, s.mean[4], sd + 5);
  for (int i = 0; i < 5; ++i) s.mean[i] = s.mean[i] + s.mean[i + 5];
  float mc[100];
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 10; ++j) mc[i * 10 + j] = i < 5 ? s.cov[i * 10 + j] + s.cov[(i + 5) * 10 + j] : s.cov[i * 10 + j];
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 10; ++j) {
      float v = j < 5 ? mc[i * 10 + j] + mc[i * 10 + j + 5] : mc[i * 10 + j];
      s.cov[i * 10 + j] = v + (i == j ? sd[i] * sd[i] : 0.0f);
    }
}

SA_HD void sa_kf_update(float pw, sa_kf& s, const sa_box& z) {  // kalman_2d_box.rs:122-148
  float sd[5];
  sa_kf_std_diag(pw, 1.0f, 1e-1f, s.mean[4], sd);
  float P[25];
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) P[i * 5 + j] = s.cov[i * 10 + j] + (i == j ? sd[i] * sd[i] : 0.0f);
  // kalman_gain = projected_cov.solve_lower_triangular(B), B[r][c] = cov[c][r]: the UN-factorised covariance is
  // used as the triangular matrix — the reference's formula, kept as is
  float G[50];
  for (int r = 0; r < 5; ++r)
    for (int c = 0; c < 10; ++c) G[r * 10 + c] = s.cov[c * 10 + r];
  for (int c = 0; c < 10; ++c)
    for (int i = 0; i < 5; ++i) {
      float coeff = G[i * 10 + c] / P[i * 5 + i];
      G[i * 10 + c] = coeff;
      float nc = -coeff;
      for (int r = i + 1; r < 5; ++r) G[r * 10 + c] = nc * P[r * 5 + i] + G[r * 10 + c];
    }
  float innov[5] = {z.xc - s.mean[0], z.yc - s.mean[1], sa_kf_opt_angle(z) - s.mean[2], z.aspect - s.mean[3], z.height - s.mean[4]};
  float nm[10];
  for (int c = 0; c < 10; ++c) {
    float acc = innov[0] * G[c];
    for (int r = 1; r < 5; ++r) acc = innov[r] * G[r * 10 + c] + acc;
    nm[c] = s.mean[c] + acc;
  }
  float gtp[50];  // 10 x 5
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 5; ++j) {
      float acc = G[i] * P[j];
      for (int k = 1; k < 5; ++k) acc = G[k * 10 + i] * P[k * 5 + j] + acc;
      gtp[i * 5 + j] = acc;
    }
  float nc[100];
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 10; ++j) {
      float acc = gtp[i * 5] * G[j];
      for (int k = 1; k < 5; ++k) acc = gtp[i * 5 + k] * G[k * 10 + j] + acc;
      nc[i * 10 + j] = s.cov[i * 10 + j] - acc;
    }
  for (int i = 0; i < 10; ++i) s.mean[i] = nm[i];
  for (int i = 0; i < 100; ++i) s.cov[i] = nc[i];
}

SA_HD sa_box sa_kf_state_box(const sa_kf& s) {  // TryFrom<KalmanState> for Universal2DBox  kalman.rs:72-92
  sa_box b;
  b.xc = s.mean[0]; b.yc = s.mean[1];
  b.has_angle = s.mean[2] == 0.0f ? 0 : 1;
  b.angle = s.mean[2];
  b.aspect = s.mean[3]; b.height = s.mean[4];
  b.confidence = 1.0f;
  b.reserved = 0;
  return b;
}

// make_prediction  kalman_prediction.rs:13-32 : (initiate when the track has no state yet,) predict, update; the box of the
// new state with the observation's confidence
SA_HD sa_box sa_kf_make_prediction(float pw, float vw, bool has_state, sa_kf& s, const sa_box& obs) {
  if (!has_state) sa_kf_initiate(pw, vw, obs, s);
  sa_kf_predict(pw, vw, s);
  sa_kf_update(pw, s, obs);
  sa_box r = sa_kf_state_box(s);
  r.confidence = obs.confidence;
  return r;
}

// VisualMetric::feature_can_be_used  visual_sort/metric.rs:227-249
SA_HD bool sa_feature_can_be_used(float minimal_area, const sa_box& b, float q, float min_q, bool has_own, float own, float min_own) {
  bool quality_ok = q >= min_q;
  bool perc_ok = has_own ? own >= min_own : true;
  float w = b.height * b.aspect;
  bool bbox_ok = w * b.height >= minimal_area;
  return bbox_ok && quality_ok && perc_ok;
}

// optimize_observations  visual_sort/metric.rs:129-154 on the bank's bookkeeping (K <= SA_MAX_BANK slots): which stored slot
// (or the new observation) ends up in which slot.  src[k] = old slot index, SA_BANK_NEW for the new observation,
// SA_BANK_NONE for an empty slot.  Returns the number of occupied slots.
//   existing observations: keep only those with features, stable sort by quality descending, if len >= max_obs drop the
//   lowest, push the new one, swap front and back (the newest observation sits at index 0).
#define SA_MAX_BANK 16
#define SA_BANK_NEW 0xfeu
#define SA_BANK_NONE 0xffu
SA_HD uint32_t sa_bank_policy(uint32_t K, uint32_t n_obs, const uint8_t* present, const float* quality, uint8_t* src) {
  uint8_t kept[SA_MAX_BANK];
  uint32_t m = 0;
  for (uint32_t k = 0; k < n_obs && k < K; ++k)
    if (present[k]) kept[m++] = (uint8_t)k;
  for (uint32_t a = 1; a < m; ++a) {  // stable insertion sort, quality descending
    uint8_t x = kept[a];
    uint32_t b = a;
    while (b > 0 && quality[kept[b - 1]] < quality[x]) { kept[b] = kept[b - 1]; --b; }
    kept[b] = x;
  }
  if (m >= K && m > 0) --m;
  kept[m++] = SA_BANK_NEW;
  { uint8_t t = kept[0]; kept[0] = kept[m - 1]; kept[m - 1] = t; }
  for (uint32_t k = 0; k < K; ++k) src[k] = k < m ? kept[k] : SA_BANK_NONE;
  return m;
}
