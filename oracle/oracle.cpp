// oracle.cpp — CPU restatement of Similari's association hot path.  TEST INFRASTRUCTURE ONLY
// (see oracle.h).  Every function cites the reference file:line it follows; all paths are
// relative to /root/reference.  Numeric types are the reference's: f32 features / Kalman,
// f64 polygon clipping, i64 x1e6 quantised weights.  Build with -ffp-contract=off.
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <unordered_map>
#include <thread>
#include <unordered_set>
#include <vector>

namespace {

constexpr float EPS = 0.00001f;                 // src/lib.rs:80
constexpr float F32_U64_MULT = 1000000.0f;      // src/trackers/sort/voting.rs:9
constexpr float CHI2INV95_4 = 11.070f;          // src/utils/kalman.rs:18-20  (CHI2INV95[4])
constexpr float CHI2_UPPER_BOUND = 100.0f;      // src/utils/kalman.rs:16
constexpr float MAHALANOBIS_NEW_TRACK_THRESHOLD = 1.0f;  // src/trackers/sort.rs:379

// wide::f32x8::reduce_add (AVX path, target-cpu=x86-64-v3 per .cargo/config.toml:1-2):
// lo+hi quads, then movehl, then lane 0+1.  PARITY-UNPINNED below 1e-6 relative.
inline float reduce_add8(const float v[8]) {
  float q0 = v[0] + v[4], q1 = v[1] + v[5], q2 = v[2] + v[6], q3 = v[3] + v[7];
  float d0 = q0 + q2, d1 = q1 + q3;
  return d0 + d1;
}

inline float opt_angle(const sa_box* b) { return b->has_angle ? b->angle : 0.0f; }

// ---- tiny dense helpers in nalgebra's evaluation order (blas gemv/axcpy: k ascending) ----
// C[m x n] = A[m x k] * B[k x n], row-major storage here; result element = sequential sum over k.
void matmul(const float* A, const float* B, float* C, int m, int k, int n) {
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) {
      float acc = A[i * k + 0] * B[0 * n + j];
      for (int kk = 1; kk < k; ++kk) acc = A[i * k + kk] * B[kk * n + j] + acc;
      C[i * n + j] = acc;
    }
}
void transpose(const float* A, float* At, int m, int n) {
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) At[j * m + i] = A[i * n + j];
}
// nalgebra solve_lower_triangular_vector_mut (column-oriented forward substitution);
// uses only the lower triangle of L (dim x dim, row-major); b is a strided vector.
bool solve_lower(const float* L, int dim, float* b, int stride) {
  for (int i = 0; i < dim; ++i) {
    float diag = L[i * dim + i];
    if (diag == 0.0f) return false;
    float coeff = b[i * stride] / diag;
    b[i * stride] = coeff;
    float nc = -coeff;
    for (int r = i + 1; r < dim; ++r) b[r * stride] = nc * L[r * dim + i] + b[r * stride];
  }
  return true;
}
// nalgebra Cholesky::new + l(): in-place column algorithm, then zero the strict upper triangle.
bool cholesky_l(float* M, int n) {
  for (int j = 0; j < n; ++j) {
    for (int k = 0; k < j; ++k) {
      float factor = -M[j * n + k];
      for (int r = j; r < n; ++r) M[r * n + j] = factor * M[r * n + k] + M[r * n + j];
    }
    float diag = M[j * n + j];
    if (diag == 0.0f || !(diag >= 0.0f)) return false;
    float denom = std::sqrt(diag);
    M[j * n + j] = denom;
    for (int r = j + 1; r < n; ++r) M[r * n + j] = M[r * n + j] / denom;
  }
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) M[i * n + j] = 0.0f;
  return true;
}

// Universal2DBoxKalmanFilter::std_position / std_velocity  kalman_2d_box.rs:47-55
void std_position(float w, float k, float cnst, float p, float out[5]) {
  float pw = k * w * p;
  out[0] = pw; out[1] = pw; out[2] = pw; out[3] = cnst; out[4] = pw;
}

// project()  kalman_2d_box.rs:104-120 on the full 10-state
void kf_project(float pw, const float mean[10], const float cov[100], float pmean[5], float pcov[25]) {
  float sd[5];
  std_position(pw, 1.0f, 1e-1f, mean[4], sd);
  float U[50];  // update_matrix: 5 x 10 identity
  std::memset(U, 0, sizeof U);
  for (int i = 0; i < 5; ++i) U[i * 10 + i] = 1.0f;
  matmul(U, mean, pmean, 5, 10, 1);
  float UC[50], Ut[50];
  matmul(U, cov, UC, 5, 10, 10);
  transpose(U, Ut, 5, 10);
  matmul(UC, Ut, pcov, 5, 10, 5);
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) pcov[i * 5 + j] = pcov[i * 5 + j] + (i == j ? sd[i] * sd[i] : 0.0f);
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------
// src/track/utils.rs:45-71  Feature::from_vec : zero-pad to f32x8 blocks; empty vec -> one zero block
uint32_t or_feature_blocks(uint32_t len) { return len == 0 ? 1u : (len + 7u) / 8u; }
uint32_t or_feature_pad(const float* v, uint32_t len, float* out) {
  uint32_t blocks = or_feature_blocks(len);
  for (uint32_t i = 0; i < blocks * 8u; ++i) out[i] = i < len ? v[i] : 0.0f;
  return blocks;
}

// src/distance.rs:9-19
float or_euclidean(const float* a, uint32_t ba, const float* b, uint32_t bb) {
  float acc = 0.0f;
  uint32_t len = std::min(ba, bb);
  for (uint32_t i = 0; i < len; ++i) {
    float blk[8];
    for (int l = 0; l < 8; ++l) {
      float d = a[i * 8 + l] - b[i * 8 + l];
      blk[l] = d * d;
    }
    acc += reduce_add8(blk);
  }
  return std::sqrt(acc);
}

// src/distance.rs:26-47  (norms recomputed per pair, as the reference does)
float or_cosine(const float* a, uint32_t ba, const float* b, uint32_t bb) {
  float divided = 0.0f;
  uint32_t len = std::min(ba, bb);
  for (uint32_t i = 0; i < len; ++i) {
    float blk[8];
    for (int l = 0; l < 8; ++l) blk[l] = a[i * 8 + l] * b[i * 8 + l];
    divided += reduce_add8(blk);
  }
  float n1 = 0.0f, n2 = 0.0f;
  for (uint32_t i = 0; i < len; ++i) {
    float blk[8];
    for (int l = 0; l < 8; ++l) blk[l] = a[i * 8 + l] * a[i * 8 + l];
    n1 = n1 + reduce_add8(blk);
  }
  for (uint32_t i = 0; i < len; ++i) {
    float blk[8];
    for (int l = 0; l < 8; ++l) blk[l] = b[i * 8 + l] * b[i * 8 + l];
    n2 = n2 + reduce_add8(blk);
  }
  return divided / std::sqrt(n1 * n2);
}

// ------------------------------------------------------------------------------------------------
// src/utils/bbox.rs:157-166
float or_radius(const sa_box* b) {
  float hw = b->aspect * b->height / 2.0f;
  float hh = b->height / 2.0f;
  return std::sqrt(hw * hw + hh * hh);
}
float or_area(const sa_box* b) {
  float w = b->height * b->aspect;
  return w * b->height;
}
// src/utils/bbox.rs:452-462
int or_too_far(const sa_box* l, const sa_box* r) {
  float max_distance = or_radius(l) + or_radius(r);
  float x = l->xc - r->xc, y = l->yc - r->yc;
  return x * x + y * y > max_distance * max_distance;
}
// src/utils/bbox.rs:464-474
float or_dist_in_2r(const sa_box* l, const sa_box* r) {
  float radial = or_radius(l) + or_radius(r);
  float x = l->xc - r->xc, y = l->yc - r->yc;
  return std::sqrt(x * x + y * y) / std::sqrt(radial * radial + EPS);
}
// src/utils/bbox.rs:287-330  Polygon::from(&Universal2DBox)
void or_vertices(const sa_box* b, double o[8]) {
  double angle = (double)opt_angle(b);
  double height = (double)b->height;
  double aspect = (double)b->aspect;
  // angle.cos() and angle.sin() of ONE value: LLVM lowers the pair to a single sincos libcall on x86_64-unknown-linux-gnu (and gcc
  // does the same merge here at -O2 and above); spelled out so that the restatement does not depend on an optimisation pass —
  // glibc's sincos and its separate cos / sin differ in the last bit for about one angle in a thousand
  double c, s;
  ::sincos(angle, &s, &c);
  double half_width = height * aspect / 2.0;
  double half_height = height / 2.0;
  double r1x = -half_width * c - half_height * s;
  double r1y = -half_width * s + half_height * c;
  double r2x = half_width * c - half_height * s;
  double r2y = half_width * s + half_height * c;
  double x = (double)b->xc, y = (double)b->yc;
  o[0] = x + r1x; o[1] = y + r1y;
  o[2] = x + r2x; o[3] = y + r2y;
  o[4] = x - r1x; o[5] = y - r1y;
  o[6] = x - r2x; o[7] = y - r2y;
}

// src/utils/clipping.rs:12-91  (open rings in, open ring out; at most ns + nc vertices)
uint32_t or_sh_clip(const double* subj, uint32_t ns, const double* clip, uint32_t nc, double* out) {
  std::vector<double> fin(subj, subj + 2 * ns);
  for (uint32_t i = 0; i < nc; ++i) {
    std::vector<double> next;
    next.swap(fin);
    uint32_t ii = i == 0 ? nc - 1 : i - 1;
    double csx = clip[2 * ii], csy = clip[2 * ii + 1];
    double cex = clip[2 * i], cey = clip[2 * i + 1];
    uint32_t nn = (uint32_t)next.size() / 2;
    auto inside = [&](double qx, double qy) {
      double r = (cex - csx) * (qy - csy) - (cey - csy) * (qx - csx);
      return r <= 0.0;
    };
    auto intersect = [&](double sx, double sy, double ex, double ey, double& ox, double& oy) {
      // compute_intersection(cp1 = s_edge_start, cp2 = s_edge_end, s = c_edge_start, e = c_edge_end)
      double dcx = sx - ex, dcy = sy - ey;
      double dpx = csx - cex, dpy = csy - cey;
      double n1 = sx * ey - sy * ex;
      double n2 = csx * cey - csy * cex;
      double n3 = 1.0 / (dcx * dpy - dcy * dpx);
      ox = (n1 * dpx - n2 * dcx) * n3;
      oy = (n1 * dpy - n2 * dcy) * n3;
    };
    for (uint32_t j = 0; j < nn; ++j) {
      uint32_t ji = j == 0 ? nn - 1 : j - 1;
      double ssx = next[2 * ji], ssy = next[2 * ji + 1];
      double sex = next[2 * j], sey = next[2 * j + 1];
      if (inside(sex, sey)) {
        if (!inside(ssx, ssy)) {
          double ox, oy;
          intersect(ssx, ssy, sex, sey, ox, oy);
          fin.push_back(ox); fin.push_back(oy);
        }
        fin.push_back(sex); fin.push_back(sey);
      } else if (inside(ssx, ssy)) {
        double ox, oy;
        intersect(ssx, ssy, sex, sey, ox, oy);
        fin.push_back(ox); fin.push_back(oy);
      }
    }
  }
  uint32_t n = (uint32_t)fin.size() / 2;
  for (uint32_t i = 0; i < 2 * n && i < 32; ++i) out[i] = fin[i];
  return n;
}

// geo 0.27 Area::unsigned_area for a Polygon without holes (NOT under /root/reference):
// Polygon::new closes the ring when first != last; twice_signed_ring_area = sum of line
// determinants with every coordinate shifted by ring[0]; < 3 coords -> 0.
double or_polygon_area(const double* xy, uint32_t n) {
  if (n == 0) return 0.0;
  std::vector<double> ring(xy, xy + 2 * n);
  if (ring[0] != ring[2 * n - 2] || ring[1] != ring[2 * n - 1]) {
    ring.push_back(xy[0]); ring.push_back(xy[1]);
  }
  uint32_t m = (uint32_t)ring.size() / 2;
  if (m < 3) return 0.0;
  double shx = ring[0], shy = ring[1];
  double tmp = 0.0;
  for (uint32_t i = 0; i + 1 < m; ++i) {
    double ax = ring[2 * i] - shx, ay = ring[2 * i + 1] - shy;
    double bx = ring[2 * i + 2] - shx, by = ring[2 * i + 3] - shy;
    tmp = tmp + (ax * by - ay * bx);
  }
  double area = tmp / (1.0 + 1.0);
  return std::fabs(area);
}

// src/utils/bbox.rs:476-510
double or_intersection(const sa_box* l, const sa_box* r) {
  if (or_too_far(l, r)) return 0.0;
  double p1[8], p2[8], out[32];
  or_vertices(l, p1);
  or_vertices(r, p2);
  uint32_t n = or_sh_clip(p1, 4, p2, 4, out);
  return or_polygon_area(out, n);
}
// src/utils/bbox.rs:512-535
int or_iou(const sa_box* l, const sa_box* r, float* out) {
  double inter = or_intersection(l, r);
  if (inter == 0.0) return 0;
  double uni = (double)(l->height * l->height * l->aspect + r->height * r->height * r->aspect) - inter;
  *out = (float)(inter / uni);
  return 1;
}

// ------------------------------------------------------------------------------------------------
// kalman_2d_box.rs:58-83
void or_kf_initiate(float pw, float vw, const sa_box* b, float mean[10], float cov[100]) {
  mean[0] = b->xc; mean[1] = b->yc; mean[2] = opt_angle(b); mean[3] = b->aspect; mean[4] = b->height;
  for (int i = 5; i < 10; ++i) mean[i] = 0.0f;
  float sd[10];
  std_position(pw, 2.0f, 1e-2f, b->height, sd);
  std_position(vw, 10.0f, 1e-5f, b->height, sd + 5);
  std::memset(cov, 0, 100 * sizeof(float));
  for (int i = 0; i < 10; ++i) cov[i * 10 + i] = sd[i] * sd[i];
}
// kalman_2d_box.rs:87-102
void or_kf_predict(float pw, float vw, const float mean[10], const float cov[100], float om[10], float oc[100]) {
  float sd[10];
  std_position(pw, 1.0f, 1e-2f, mean[4], sd);
  std_position(vw, 1.0f, 1e-5f, mean[4], sd + 5);
  float M[100], Mt[100], MC[100], MCM[100], m2[10];
  std::memset(M, 0, sizeof M);
  for (int i = 0; i < 10; ++i) M[i * 10 + i] = 1.0f;
  for (int i = 0; i < 5; ++i) M[i * 10 + 5 + i] = 1.0f;  // DT = 1
  matmul(M, mean, m2, 10, 10, 1);
  matmul(M, cov, MC, 10, 10, 10);
  transpose(M, Mt, 10, 10);
  matmul(MC, Mt, MCM, 10, 10, 10);
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 10; ++j) oc[i * 10 + j] = MCM[i * 10 + j] + (i == j ? sd[i] * sd[i] : 0.0f);
  std::memcpy(om, m2, sizeof m2);
}
// kalman_2d_box.rs:122-148  (gain = projected_cov.solve_lower_triangular(B) on the UN-factorised
// covariance — the reference's quirk, copied)
void or_kf_update(float pw, float /*vw*/, const float mean[10], const float cov[100], const sa_box* z,
                  float om[10], float oc[100]) {
  float pmean[5], pcov[25];
  kf_project(pw, mean, cov, pmean, pcov);
  float Ut[50];  // 10 x 5
  std::memset(Ut, 0, sizeof Ut);
  for (int i = 0; i < 5; ++i) Ut[i * 5 + i] = 1.0f;
  float CUt[50], B[50];  // CUt 10x5, B = its transpose 5x10
  matmul(cov, Ut, CUt, 10, 10, 5);
  transpose(CUt, B, 10, 5);
  float G[50];  // kalman_gain 5 x 10
  std::memcpy(G, B, sizeof G);
  for (int c = 0; c < 10; ++c) solve_lower(pcov, 5, G + c, 10);
  float innov[5] = {z->xc - pmean[0], z->yc - pmean[1], opt_angle(z) - pmean[2], z->aspect - pmean[3],
                    z->height - pmean[4]};
  float ig[10];
  matmul(innov, G, ig, 1, 5, 10);
  float nm[10];
  for (int i = 0; i < 10; ++i) nm[i] = mean[i] + ig[i];
  float Gt[50], GtP[50], GtPG[100];
  transpose(G, Gt, 5, 10);       // 10 x 5
  matmul(Gt, pcov, GtP, 10, 5, 5);
  matmul(GtP, G, GtPG, 10, 5, 10);
  for (int i = 0; i < 100; ++i) oc[i] = cov[i] - GtPG[i];
  std::memcpy(om, nm, sizeof nm);
}
// kalman_2d_box.rs:150-170
float or_kf_distance(float pw, float /*vw*/, const float mean[10], const float cov[100], const sa_box* z) {
  float pmean[5], pcov[25];
  kf_project(pw, mean, cov, pmean, pcov);
  float r[5] = {z->xc, z->yc, opt_angle(z), z->aspect, z->height};
  for (int i = 0; i < 5; ++i) r[i] = r[i] - pmean[i];
  if (!cholesky_l(pcov, 5)) return std::numeric_limits<float>::quiet_NaN();
  solve_lower(pcov, 5, r, 1);
  float s = 0.0f;
  for (int i = 0; i < 5; ++i) s = s + r[i] * r[i];
  return s;
}
// Same, from the 5-mean / 5x5 top-left block the C ABI carries (project() reads nothing else:
// update_matrix is the 5x10 identity, so U*mean = mean[0..5] and U*C*U^T = C[0..5,0..5] exactly).
float or_kf_distance5(float pw, const float mean5[5], const float cov25[25], const sa_box* z) {
  float mean[10], cov[100];
  std::memset(mean, 0, sizeof mean);
  std::memset(cov, 0, sizeof cov);
  for (int i = 0; i < 5; ++i) {
    mean[i] = mean5[i];
    for (int j = 0; j < 5; ++j) cov[i * 10 + j] = cov25[i * 5 + j];
  }
  return or_kf_distance(pw, 0.0f, mean, cov, z);
}
// kalman_2d_box.rs:172-184
float or_kf_cost(float d, int inverted) {
  if (!inverted) return d > CHI2INV95_4 ? CHI2_UPPER_BOUND : d;
  return d > CHI2INV95_4 ? 0.0f : CHI2_UPPER_BOUND - d;
}
// src/utils/kalman.rs:72-92
void or_kf_state_box(const float mean[10], sa_box* o) {
  o->xc = mean[0]; o->yc = mean[1];
  o->has_angle = mean[2] == 0.0f ? 0 : 1;
  o->angle = mean[2];
  o->aspect = mean[3]; o->height = mean[4];
  o->confidence = 1.0f; o->reserved = 0;
}
// src/trackers/kalman_prediction.rs:13-32
void or_make_prediction(float pw, float vw, int has_state, float mean[10], float cov[100], const sa_box* obs,
                        sa_box* out) {
  float m0[10], c0[100], m1[10], c1[100];
  if (has_state) { std::memcpy(m0, mean, sizeof m0); std::memcpy(c0, cov, sizeof c0); }
  else or_kf_initiate(pw, vw, obs, m0, c0);
  or_kf_predict(pw, vw, m0, c0, m1, c1);
  or_kf_update(pw, vw, m1, c1, obs, mean, cov);
  or_kf_state_box(mean, out);
  out->confidence = obs->confidence;
}

// ------------------------------------------------------------------------------------------------
// src/trackers/spatio_temporal_constraints.rs:48-59 (constraints sorted by delta, deduped)
int or_constraints_validate(uint32_t n, const uint64_t* deltas, const float* max_dists, uint64_t epoch_delta,
                            float dist) {
  for (uint32_t i = 0; i < n; ++i)
    if (deltas[i] >= epoch_delta) return dist <= max_dists[i];
  return 1;
}
// src/trackers/sort.rs:250-270, visual_sort/track_attributes.rs:188-208 (same scene assumed)
int or_compatible(const sa_config* cfg, const sa_box* cand, uint64_t ce, const sa_box* track, uint64_t te) {
  uint64_t delta = ce > te ? ce - te : te - ce;
  float center_dist = or_dist_in_2r(cand, track);
  return cfg->max_idle_epochs >= delta &&
         or_constraints_validate(cfg->n_constraints, cfg->constraint_epoch_delta, cfg->constraint_max_dist, delta,
                                 center_dist);
}

// src/trackers/sort/metric.rs:38-77 ; src/trackers/visual_sort/metric.rs:156-198
int or_positional_metric(const sa_config* cfg, const sa_box* cand, const sa_box* track, const float* mean5,
                         const float* cov25, float* out) {
  float conf = cand->confidence < cfg->positional_min_confidence ? cfg->positional_min_confidence : cand->confidence;
  if (or_too_far(cand, track)) return 0;
  if (cfg->positional_kind == SA_POS_MAHALANOBIS) {
    float dist = or_kf_distance5(cfg->kf_position_weight, mean5, cov25, cand);
    *out = or_kf_cost(dist, 1) / conf;
    return 1;
  }
  float iou;
  if (!or_iou(cand, track, &iou)) return 0;
  float e = iou * conf;
  if (!(e >= cfg->positional_threshold)) return 0;
  *out = e;
  return 1;
}

// ------------------------------------------------------------------------------------------------
// (w * F32_U64_MULT) as i64  — Rust `as`: truncation toward zero, saturating, NaN -> 0
int64_t or_quantise(float w) {
  float v = w * F32_U64_MULT;
  if (v != v) return 0;
  if (v >= 9223372036854775808.0f) return std::numeric_limits<int64_t>::max();
  if (v <= -9223372036854775808.0f) return std::numeric_limits<int64_t>::min();
  return (int64_t)v;
}

// pathfinding 4.8 kuhn_munkres::kuhn_munkres (NOT under /root/reference; call site
// src/trackers/sort/voting.rs:86).  Restated from the crate's published algorithm: lx = row max,
// ly = 0; one alternating tree per root row in index order; the next column is the LOWEST-index
// column outside the tree with the smallest slack (strict <); labels move by delta; augment.
int or_kuhn_munkres(uint32_t nx, uint32_t ny, const int64_t* w, int64_t* out_total, uint32_t* out_assign) {
  if (nx > ny) return -1;
  const uint32_t NONE = 0xffffffffu;
  std::vector<uint32_t> xy(nx, NONE), yx(ny, NONE);
  std::vector<int64_t> lx(nx), ly(ny, 0);
  for (uint32_t r = 0; r < nx; ++r) {
    int64_t m = std::numeric_limits<int64_t>::min();
    for (uint32_t c = 0; c < ny; ++c) m = std::max(m, w[(size_t)r * ny + c]);
    lx[r] = m;
  }
  std::vector<uint8_t> s(nx);
  std::vector<uint32_t> s_list;
  std::vector<uint32_t> alternating(ny), slackx(ny);
  std::vector<int64_t> slack(ny);
  for (uint32_t root = 0; root < nx; ++root) {
    std::fill(alternating.begin(), alternating.end(), NONE);
    std::fill(s.begin(), s.end(), 0);
    s_list.clear();
    s[root] = 1; s_list.push_back(root);
    for (uint32_t y = 0; y < ny; ++y) slack[y] = lx[root] + ly[y] - w[(size_t)root * ny + y];
    std::fill(slackx.begin(), slackx.end(), root);
    uint32_t yend;
    for (;;) {
      int64_t delta = std::numeric_limits<int64_t>::max();
      uint32_t x = 0, y = 0;
      for (uint32_t yy = 0; yy < ny; ++yy)
        if (alternating[yy] == NONE && slack[yy] < delta) { delta = slack[yy]; x = slackx[yy]; y = yy; }
      if (delta > 0) {
        for (uint32_t xx : s_list) lx[xx] -= delta;   // s.ones(): set semantics, order irrelevant
        for (uint32_t yy = 0; yy < ny; ++yy) {
          if (alternating[yy] != NONE) ly[yy] += delta;
          else slack[yy] -= delta;
        }
      }
      alternating[y] = x;
      if (yx[y] == NONE) { yend = y; break; }
      uint32_t x2 = yx[y];
      s[x2] = 1; s_list.push_back(x2);
      for (uint32_t yy = 0; yy < ny; ++yy)
        if (alternating[yy] == NONE) {
          int64_t alt = lx[x2] + ly[yy] - w[(size_t)x2 * ny + yy];
          if (slack[yy] > alt) { slack[yy] = alt; slackx[yy] = x2; }
        }
    }
    uint32_t y = yend;
    while (y != NONE) {
      uint32_t x = alternating[y];
      uint32_t prec = xy[x];
      yx[y] = x;
      xy[x] = y;
      y = prec;
    }
  }
  int64_t total = 0;
  for (uint32_t r = 0; r < nx; ++r) total += lx[r];
  for (uint32_t c = 0; c < ny; ++c) total += ly[c];
  if (out_total) *out_total = total;
  for (uint32_t r = 0; r < nx; ++r) out_assign[r] = xy[r];
  return 0;
}

}  // extern "C"

namespace {

struct Dist { uint64_t from, to; float pos, vis; };  // NaN = None
inline bool some(float v) { return v == v; }

// src/trackers/sort/voting.rs:30-100  SortVoting::winners over a canonical-order distance list.
std::unordered_map<uint64_t, uint64_t> sort_voting(float threshold_f, size_t candidate_num, size_t track_num,
                                                   const std::vector<Dist>& distances, int64_t* total_out) {
  std::unordered_map<uint64_t, uint64_t> res;
  if (total_out) *total_out = 0;
  int64_t threshold = or_quantise(threshold_f);
  if (track_num == 0) return res;
  size_t candidates_index = 0;
  std::vector<uint64_t> tracks_index(candidate_num, 0);
  std::unordered_map<uint64_t, size_t> tracks_r_index;
  size_t cols = candidate_num + track_num;
  std::vector<int64_t> cost(candidate_num * cols, 0);
  for (const Dist& d : distances) {
    int64_t weight = or_quantise(some(d.pos) ? d.pos : 0.0f);
    size_t row;
    auto it = tracks_r_index.find(d.from);
    if (it != tracks_r_index.end()) row = it->second;
    else { row = candidates_index++; tracks_index[row] = d.from; tracks_r_index[d.from] = row; }
    size_t col;
    it = tracks_r_index.find(d.to);
    if (it != tracks_r_index.end()) col = it->second;
    else { col = tracks_index.size(); tracks_index.push_back(d.to); tracks_r_index[d.to] = col; }
    cost[row * cols + col] = weight;
  }
  for (size_t i = 0; i < candidate_num; ++i) cost[i * cols + i] = threshold;
  if (candidate_num == 0) return res;
  std::vector<uint32_t> sol(candidate_num);
  or_kuhn_munkres((uint32_t)candidate_num, (uint32_t)cols, cost.data(), total_out, sol.data());
  for (size_t i = 0; i < candidate_num; ++i) {
    uint64_t from = tracks_index[i];
    uint64_t to = sol[i] < tracks_index.size() ? tracks_index[sol[i]] : 0;
    if (from > 0 && to > 0) res[from] = to;
  }
  return res;
}

struct BestFitElt { uint64_t query, winner; double weight; };

// src/track/voting/best.rs:52-128  BestFitVoting::winners.  The reference groups through a HashMap
// (iteration order arbitrary) and stable-sorts by weight; canonical order here = first appearance
// of the (from, to) group in the canonical distance list.
std::unordered_map<uint64_t, std::vector<BestFitElt>> bestfit_voting(float max_distance, size_t min_votes,
                                                                      const std::vector<Dist>& distances) {
  float max_dist = -1.0f;
  std::vector<std::pair<uint64_t, uint64_t>> order;
  std::map<std::pair<uint64_t, uint64_t>, std::vector<float>> groups;
  for (const Dist& d : distances) {
    if (!some(d.vis)) continue;
    if (max_dist < d.vis) max_dist = d.vis;
    if (!(d.vis <= max_distance)) continue;
    auto key = std::make_pair(d.from, d.to);
    auto it = groups.find(key);
    if (it == groups.end()) { order.push_back(key); groups[key] = {d.vis}; }
    else it->second.push_back(d.vis);
  }
  std::vector<BestFitElt> cands;
  for (auto& key : order) {
    auto& v = groups[key];
    if (v.size() < min_votes) continue;
    double weight = 0.0;
    for (float d : v) weight += (double)(max_dist - d);
    cands.push_back({key.first, key.second, weight});
  }
  std::stable_sort(cands.begin(), cands.end(), [](const BestFitElt& a, const BestFitElt& b) { return a.weight > b.weight; });
  std::unordered_set<uint64_t> results;
  for (auto& c : cands) {
    if (results.count(c.winner)) c.winner = c.query;
    else results.insert(c.winner);
  }
  std::unordered_map<uint64_t, std::vector<BestFitElt>> res;
  for (auto& c : cands) res[c.query].push_back(c);
  return res;
}

struct VisualWinner { uint64_t to; uint8_t type; };

// src/trackers/visual_sort/voting.rs:48-100
std::unordered_map<uint64_t, VisualWinner> visual_voting(float positional_threshold, float max_feature_distance,
                                                         size_t min_votes, const std::vector<Dist>& distances,
                                                         int64_t* total_out) {
  auto fw = bestfit_voting(max_feature_distance, min_votes, distances);
  std::unordered_set<uint64_t> excluded;
  std::unordered_map<uint64_t, VisualWinner> winners;
  for (auto& kv : fw) {
    uint64_t wt = kv.second[0].winner;
    excluded.insert(wt);
    winners[kv.first] = {wt, (uint8_t)SA_VOTE_VISUAL};
  }
  std::unordered_set<uint64_t> rem_c, rem_t;
  std::vector<Dist> remaining;
  for (const Dist& d : distances) {
    if (winners.count(d.from) || excluded.count(d.to)) continue;
    if (!some(d.pos)) continue;
    rem_c.insert(d.from);
    rem_t.insert(d.to);
    remaining.push_back(d);
  }
  auto pw = sort_voting(positional_threshold, rem_c.size(), rem_t.size(), remaining, total_out);
  for (auto& kv : pw) winners[kv.first] = {kv.second, (uint8_t)SA_VOTE_POSITIONAL};
  return winners;
}

std::vector<Dist> make_dists(uint32_t n, const uint64_t* from, const uint64_t* to, const float* pos, const float* vis) {
  std::vector<Dist> v(n);
  const float nan = std::numeric_limits<float>::quiet_NaN();
  for (uint32_t i = 0; i < n; ++i) v[i] = {from[i], to[i], pos ? pos[i] : nan, vis ? vis[i] : nan};
  return v;
}

}  // namespace

extern "C" {

int or_sort_voting(float threshold, uint32_t n_cand, uint32_t n_tracks, uint32_t n, const uint64_t* from,
                   const uint64_t* to, const float* positional, uint32_t n_ids, const uint64_t* cand_ids,
                   uint64_t* out_to, int64_t* out_total) {
  auto d = make_dists(n, from, to, positional, nullptr);
  auto w = sort_voting(threshold, n_cand, n_tracks, d, out_total);
  for (uint32_t i = 0; i < n_ids; ++i) {
    auto it = w.find(cand_ids[i]);
    out_to[i] = it == w.end() ? 0 : it->second;
  }
  return 0;
}

int or_bestfit_voting(float max_distance, uint32_t min_votes, uint32_t n, const uint64_t* from, const uint64_t* to,
                      const float* visual, uint32_t n_ids, const uint64_t* cand_ids, uint64_t* out_to,
                      double* out_weight) {
  auto d = make_dists(n, from, to, nullptr, visual);
  auto w = bestfit_voting(max_distance, min_votes, d);
  for (uint32_t i = 0; i < n_ids; ++i) {
    auto it = w.find(cand_ids[i]);
    out_to[i] = it == w.end() ? 0 : it->second[0].winner;
    if (out_weight) out_weight[i] = it == w.end() ? 0.0 : it->second[0].weight;
  }
  return 0;
}

int or_visual_voting(float positional_threshold, float max_feature_distance, uint32_t min_votes, uint32_t n,
                     const uint64_t* from, const uint64_t* to, const float* positional, const float* visual,
                     uint32_t n_ids, const uint64_t* cand_ids, uint64_t* out_to, uint8_t* out_type) {
  auto d = make_dists(n, from, to, positional, visual);
  auto w = visual_voting(positional_threshold, max_feature_distance, min_votes, d, nullptr);
  for (uint32_t i = 0; i < n_ids; ++i) {
    auto it = w.find(cand_ids[i]);
    out_to[i] = it == w.end() ? 0 : it->second.to;
    out_type[i] = it == w.end() ? (uint8_t)SA_VOTE_NONE : it->second.type;
  }
  return 0;
}

// One scene-frame: foreign_track_distances (store.rs:429-460 -> track.rs:604-652 -> metric) followed by
// winners(), in canonical order: candidates in input order, tracks in table order, observations in bank order.
// Orchestration follows sort/simple_api.rs:110-196 and visual_sort/simple_api.rs:99-230.
int or_associate(const sa_config* cfg, uint32_t total_tracks_in_store, const sa_tracks* tracks, uint64_t epoch,
                 const sa_detections* det, or_frame_out* out) {
  return or_associate_sharded(cfg, total_tracks_in_store, tracks, epoch, det, out, 1);
}

// The same with the distance stage spread over `shards` host threads the way the reference's TrackStore spreads it
// (store.rs:490-493: a track lives in shard id % shards; store.rs:199-241: every shard's worker computes the distances of ALL
// candidates against ITS tracks), followed by ONE vote over the merged distances (sort/simple_api.rs:147-162).  The merged list
// is put back into the canonical order (candidates in input order, tracks in table order, observations in bank order), so the
// result is the single-thread result, cell for cell and id for id.
int or_associate_sharded(const sa_config* cfg, uint32_t total_tracks_in_store, const sa_tracks* tracks, uint64_t epoch,
                         const sa_detections* det, or_frame_out* out, uint32_t shards) {
  const uint32_t N = det->n, T = tracks->n;
  const bool visual = cfg->visual_kind != SA_VIS_NONE;
  const uint32_t K = visual ? std::max(1u, cfg->max_observations) : 1u;
  const uint32_t D = cfg->feature_len;
  const uint32_t blocks = or_feature_blocks(D);
  const float nan = std::numeric_limits<float>::quiet_NaN();
  const uint64_t CAND = 0x8000000000000000ull;

  if (out->positional) std::fill(out->positional, out->positional + (size_t)N * T, nan);
  if (out->visual) std::fill(out->visual, out->visual + (size_t)N * T * K, nan);
  if (out->quantised) std::fill(out->quantised, out->quantised + (size_t)N * T, (int64_t)0);
  if (out->compatible) std::fill(out->compatible, out->compatible + (size_t)N * T, (uint8_t)0);

  // Feature::from_vec for every candidate (simple_api.rs:156-158) and every stored observation.
  std::vector<float> cfeat, tfeat;
  if (visual && det->feats) {
    cfeat.resize((size_t)N * blocks * 8);
    for (uint32_t i = 0; i < N; ++i) or_feature_pad(det->feats + (size_t)i * D, D, &cfeat[(size_t)i * blocks * 8]);
  }
  if (visual && tracks->feats) {
    tfeat.resize((size_t)T * K * blocks * 8);
    for (size_t r = 0; r < (size_t)T * K; ++r) or_feature_pad(tracks->feats + r * D, D, &tfeat[r * blocks * 8]);
  }
  auto cand_has_feat = [&](uint32_t i) { return visual && det->feats && (!det->feat_present || det->feat_present[i]); };
  auto track_has_feat = [&](uint32_t j, uint32_t k) {
    return visual && tracks->feats && (!tracks->feat_present || tracks->feat_present[(size_t)j * K + k]);
  };

  std::vector<Dist> dists;
  std::vector<uint64_t> cand_ids(N);
  for (uint32_t i = 0; i < N; ++i) cand_ids[i] = CAND | (uint64_t)(i + 1);
  if (shards < 1) shards = 1;
  struct Keyed { uint64_t key; Dist d; };
  // one shard's share of foreign_track_distances: every candidate against the tracks with id % shards == shard
  auto shard_work = [&](uint32_t shard, std::vector<Keyed>& sink) {
  for (uint32_t i = 0; i < N; ++i) {
    const sa_box* cb = &det->boxes[i];
    bool can_use = false;
    if (visual) {
      // feature_can_be_used  visual_sort/metric.rs:227-249
      float q = det->feat_quality ? det->feat_quality[i] : 1.0f;
      bool quality_ok = q >= cfg->visual_minimal_quality_use;
      bool perc_ok = true;
      if (det->own_area && det->own_area[i] == det->own_area[i])
        perc_ok = det->own_area[i] >= cfg->visual_minimal_own_area_percentage_use;
      bool bbox_ok = or_area(cb) >= cfg->visual_minimal_area;
      can_use = bbox_ok && quality_ok && perc_ok;
    }
    for (uint32_t j = 0; j < T; ++j) {
      if (shards > 1 && tracks->ids[j] % shards != shard) continue;
      const sa_box* tb = &tracks->boxes[j];
      if (!or_compatible(cfg, cb, epoch, tb, tracks->epochs[j])) continue;
      if (out->compatible) out->compatible[(size_t)i * T + j] = 1;
      uint32_t collected = 0;
      for (uint32_t k = 0; k < K; ++k) collected += track_has_feat(j, k) ? 1u : 0u;
      for (uint32_t k = 0; k < K; ++k) {
        if (k > 0 && !track_has_feat(j, k)) continue;  // older observations survive only with a feature
        float pos = nan, vis = nan;
        if (k == 0) {
          float v;
          const float* m5 = tracks->kf_mean ? tracks->kf_mean + (size_t)j * 5 : nullptr;
          const float* c25 = tracks->kf_cov ? tracks->kf_cov + (size_t)j * 25 : nullptr;
          if (or_positional_metric(cfg, cb, tb, m5, c25, &v)) pos = v;
        }
        if (visual && can_use && cand_has_feat(i) && track_has_feat(j, k) &&
            collected >= cfg->visual_minimal_track_length) {
          const float* a = &cfeat[(size_t)i * blocks * 8];
          const float* b = &tfeat[((size_t)j * K + k) * blocks * 8];
          float d = cfg->visual_kind == SA_VIS_COSINE ? or_cosine(a, blocks, b, blocks) : or_euclidean(a, blocks, b, blocks);
          bool ok = cfg->visual_kind == SA_VIS_COSINE ? d >= cfg->visual_threshold : d <= cfg->visual_threshold;
          if (ok) vis = cfg->visual_kind == SA_VIS_COSINE ? 1.0f - d : d;
        }
        if (out->positional && k == 0) out->positional[(size_t)i * T + j] = pos;
        if (out->quantised && k == 0) out->quantised[(size_t)i * T + j] = or_quantise(some(pos) ? pos : 0.0f);
        if (out->visual) out->visual[((size_t)i * T + j) * K + k] = vis;
        bool keep = visual ? (some(pos) || some(vis)) : some(pos);
        if (keep) sink.push_back({((uint64_t)i * T + j) * K + k, {cand_ids[i], tracks->ids[j], pos, vis}});
      }
    }
  }
  };
  if (shards == 1) {
    std::vector<Keyed> all;
    shard_work(0, all);
    dists.reserve(all.size());
    for (const Keyed& kd : all) dists.push_back(kd.d);
  } else {
    std::vector<std::vector<Keyed>> part(shards);
    std::vector<std::thread> th;
    for (uint32_t sh = 0; sh < shards; ++sh) th.emplace_back([&, sh] { shard_work(sh, part[sh]); });
    for (auto& t : th) t.join();
    std::vector<Keyed> all;
    size_t total = 0;
    for (auto& p : part) total += p.size();
    all.reserve(total);
    for (auto& p : part) { all.insert(all.end(), p.begin(), p.end()); std::vector<Keyed>().swap(p); }
    std::sort(all.begin(), all.end(), [](const Keyed& a, const Keyed& b) { return a.key < b.key; });  // keys are unique
    dists.reserve(all.size());
    for (const Keyed& kd : all) dists.push_back(kd.d);
  }
  out->n_distances = dists.size();
  out->total_weight = 0;

  float thr = cfg->positional_kind == SA_POS_MAHALANOBIS ? MAHALANOBIS_NEW_TRACK_THRESHOLD : cfg->positional_threshold;
  for (uint32_t i = 0; i < N; ++i) {
    if (out->track_id) out->track_id[i] = 0;
    if (out->voting_type) out->voting_type[i] = SA_VOTE_NONE;
  }
  if (!visual) {
    auto w = sort_voting(thr, N, total_tracks_in_store, dists, &out->total_weight);
    for (uint32_t i = 0; i < N; ++i) {
      auto it = w.find(cand_ids[i]);
      if (it != w.end() && it->second != cand_ids[i] && !(it->second & CAND)) {
        if (out->track_id) out->track_id[i] = it->second;
        if (out->voting_type) out->voting_type[i] = SA_VOTE_POSITIONAL;
      }
    }
  } else {
    auto w = visual_voting(thr, std::numeric_limits<float>::max(), cfg->visual_min_votes, dists, &out->total_weight);
    for (uint32_t i = 0; i < N; ++i) {
      auto it = w.find(cand_ids[i]);
      if (it != w.end() && it->second.to != cand_ids[i] && !(it->second.to & CAND)) {
        if (out->track_id) out->track_id[i] = it->second.to;
        if (out->voting_type) out->voting_type[i] = it->second.type;
      }
    }
  }
  return 0;
}

}  // extern "C"

// ---- src/utils/nms.rs:32-72 -------------------------------------------------------------------------------------------
// scores: NULL or NaN entries = Option::None (rank = height, passes every score threshold); score_threshold NaN = None (f32::MIN).
// out_keep: indices into `boxes` of the surviving boxes, in the reference's output order (rank descending, stable).
extern "C" int or_nms(uint32_t n, const sa_box* boxes, const float* scores, float nms_threshold, float score_threshold,
                      uint32_t* out_keep, uint32_t* out_n) {
  const float thr = score_threshold == score_threshold ? score_threshold : -3.4028234663852886e38f;
  struct Cand { uint32_t src; float rank; };
  std::vector<Cand> c;
  for (uint32_t i = 0; i < n; ++i) {
    const bool has = scores && scores[i] == scores[i];
    const float sc = has ? scores[i] : 3.4028234663852886e38f;   // score.unwrap_or(f32::MAX) > score_threshold
    if (!(sc > thr && boxes[i].height > 0.0f && boxes[i].aspect > 0.0f)) continue;
    c.push_back({i, has ? scores[i] : boxes[i].height});           // rank.unwrap_or(bbox.height)
  }
  std::stable_sort(c.begin(), c.end(), [](const Cand& a, const Cand& b) { return a.rank > b.rank; });  // sorted_by(b.rank cmp a.rank)
  std::vector<uint8_t> excluded(c.size(), 0);
  for (size_t i = 0; i < c.size(); ++i) {
    if (excluded[i]) continue;
    for (size_t j = i + 1; j < c.size(); ++j) {
      if (excluded[j]) continue;
      const sa_box* cb = &boxes[c[i].src];
      const sa_box* ob = &boxes[c[j].src];
      float metric = (float)or_intersection(cb, ob) / or_area(ob);
      if (metric > nms_threshold) excluded[j] = 1;
    }
  }
  uint32_t k = 0;
  for (size_t i = 0; i < c.size(); ++i)
    if (!excluded[i]) out_keep[k++] = c[i].src;
  *out_n = k;
  return 0;
}

// ---- src/utils/clipping/bbox_own_areas.rs:8-46 -----------------------------------------------------------
// exclusively_owned_areas(): own_poly = Polygon(box i) minus, one after another, the polygon of every other box that is not
// too_far() (geo 0.27 BooleanOps::difference — NOT under /root/reference); ..._normalized_shares(): (unsigned_area(own_poly) /
// (area_i + EPS) as f64) as f32, clamped to 1.0.  Only the share reaches the tracker (visual_sort/simple_api.rs:111-127), so the
// restatement computes the AREA of the difference, not its outline: the set difference of a convex piece and a convex quad is
// the disjoint union of "piece ∩ inside(f_0..f_{k-1}) ∩ outside(f_k)" over the quad's edges f_k, each of which is convex again
// (half-plane clipping), so the owned region is carried as a list of convex pieces and its area is the sum of shoelace areas.
// PARITY UNPINNED below f64 rounding: geo's sweep-line produces the same region with differently rounded intersection points;
// the reference's own test (bbox_own_areas.rs:58-82) checks to EPS = 1e-5 and so do ours.
namespace {
typedef std::vector<std::pair<double, double>> ConvexPiece;
// keep the part of a convex polygon on one side of the directed line a->b: side(p) = cross(b-a, p-a); keep_le: side <= 0
// (the interior side of the reference's clockwise box polygons, clipping.rs:12-15), else side > 0.
ConvexPiece halfplane_clip(const ConvexPiece& in, double ax, double ay, double bx, double by, bool keep_le) {
  ConvexPiece out;
  const size_t n = in.size();
  for (size_t i = 0; i < n; ++i) {
    const auto& p = in[i];
    const auto& q = in[(i + 1) % n];
    const double sp = (bx - ax) * (p.second - ay) - (by - ay) * (p.first - ax);
    const double sq = (bx - ax) * (q.second - ay) - (by - ay) * (q.first - ax);
    const bool ip = keep_le ? sp <= 0.0 : sp > 0.0;
    const bool iq = keep_le ? sq <= 0.0 : sq > 0.0;
    if (ip) out.push_back(p);
    if (ip != iq) {
      const double t = sp / (sp - sq);
      out.emplace_back(p.first + t * (q.first - p.first), p.second + t * (q.second - p.second));
    }
  }
  return out;
}
double piece_area(const ConvexPiece& p) {
  if (p.size() < 3) return 0.0;
  double s = 0.0;
  for (size_t i = 1; i + 1 < p.size(); ++i)
    s += (p[i].first - p[0].first) * (p[i + 1].second - p[0].second) - (p[i + 1].first - p[0].first) * (p[i].second - p[0].second);
  return std::fabs(s) * 0.5;
}
}  // namespace

extern "C" int or_own_area_shares(uint32_t n, const sa_box* boxes, float* out_share) {
  std::vector<double> verts((size_t)n * 8);
  for (uint32_t i = 0; i < n; ++i) or_vertices(&boxes[i], &verts[(size_t)i * 8]);
  for (uint32_t i = 0; i < n; ++i) {
    std::vector<ConvexPiece> pieces(1);
    for (int v = 0; v < 4; ++v) pieces[0].emplace_back(verts[(size_t)i * 8 + 2 * v], verts[(size_t)i * 8 + 2 * v + 1]);
    for (uint32_t j = 0; j < n && !pieces.empty(); ++j) {
      if (j == i || or_too_far(&boxes[i], &boxes[j])) continue;
      const double* q = &verts[(size_t)j * 8];
      std::vector<ConvexPiece> next;
      for (ConvexPiece rem : pieces) {
        for (int k = 0; k < 4 && rem.size() >= 3; ++k) {
          const double ax = q[2 * k], ay = q[2 * k + 1], bx = q[2 * ((k + 1) & 3)], by = q[2 * ((k + 1) & 3) + 1];
          ConvexPiece outside = halfplane_clip(rem, ax, ay, bx, by, false);
          if (piece_area(outside) > 0.0) next.push_back(std::move(outside));
          rem = halfplane_clip(rem, ax, ay, bx, by, true);
        }
      }
      pieces.swap(next);
    }
    double own = 0.0;
    for (const auto& p : pieces) own += piece_area(p);
    float e = (float)(own / (double)(or_area(&boxes[i]) + 1e-5f));
    out_share[i] = e >= 1.0f ? 1.0f : e;
  }
  return 0;
}
