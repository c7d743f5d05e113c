// sa_device.h — per-cell and per-component device logic of the association engine (gfx950).
//
// Everything here is scalar, branch-for-branch faithful to the reference's arithmetic (f32 boxes and
// Kalman, f64 polygon clipping, i64 quantised weights) and marked SA_HD so that tests/emu can compile
// the very same source with g++ and check it against the oracle on the CPU (test infrastructure; the
// product never runs these on the host).  Wave-level / MFMA code lives in the .hip files.
//
// Build with -ffp-contract=off: rustc never fuses a*b+c, and the bit-exact gates (IoU quantised
// matrix, assignment indices) rely on the same rounding sequence.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SA_HD __host__ __device__ __forceinline__
#else
#define SA_HD inline
#endif

#define SA_EPS 0.00001f                 // src/lib.rs:80
#define SA_F32_U64_MULT 1000000.0f      // src/trackers/sort/voting.rs:9
#define SA_CHI2INV95_4 11.070f          // src/utils/kalman.rs:18-20
#define SA_CHI2_UPPER_BOUND 100.0f      // src/utils/kalman.rs:16
#define SA_MAX_CONSTRAINTS 16
#define SA_NONE 0xffffffffu

// Per-box derived geometry, 16 B: centre, bounding-circle radius (bbox.rs:157-161), f32 area term
// h*h*aspect exactly as calculate_metric_object forms it (bbox.rs:523).
struct sa_geo {
  float xc, yc, r, hha;
};

// Half extents of a box for the axis-aligned quick reject (sa_aa_quick_reject): hw = aspect * height / 2, hh = height / 2; hw < 0 marks
// a box with a non-zero angle (its polygon is not its bounding rectangle: no quick reject).
struct sa_ext {
  float hw, hh;
};
SA_HD sa_ext sa_box_ext(float aspect, float height, bool oriented) {
  sa_ext e;
  e.hw = aspect * height / 2.0f;
  e.hh = height / 2.0f;
  if (oriented) e.hw = -e.hw;
  return e;
}

struct sa_constraints {
  uint32_t n;
  uint32_t pad;
  uint64_t delta[SA_MAX_CONSTRAINTS];
  float max_dist[SA_MAX_CONSTRAINTS];
};

// ---- box preparation (bbox.rs:157-166, 287-330) -------------------------------------------------------
SA_HD float sa_radius(float aspect, float height) {
  float hw = aspect * height / 2.0f;
  float hh = height / 2.0f;
  return sqrtf(hw * hw + hh * hh);
}
SA_HD float sa_area(float aspect, float height) {  // Universal2DBox::area  bbox.rs:163-166
  float w = height * aspect;
  return w * height;
}
// Polygon::from(&Universal2DBox) with cos/sin of (angle as f64) supplied by the host's libm
// (c = 1, s = 0 for boxes without an angle).
SA_HD void sa_vertices(float xc, float yc, float aspect_f, float height_f, double c, double s, double* o) {
  double height = (double)height_f;
  double aspect = (double)aspect_f;
  double half_width = height * aspect / 2.0;
  double half_height = height / 2.0;
  double r1x = -half_width * c - half_height * s;
  double r1y = -half_width * s + half_height * c;
  double r2x = half_width * c - half_height * s;
  double r2y = half_width * s + half_height * c;
  double x = (double)xc, y = (double)yc;
  o[0] = x + r1x; o[1] = y + r1y;
  o[2] = x + r2x; o[3] = y + r2y;
  o[4] = x - r1x; o[5] = y - r1y;
  o[6] = x - r2x; o[7] = y - r2y;
}

// ---- pair pre-filter (sort.rs:250-270, spatio_temporal_constraints.rs:48-59, bbox.rs:452-474) ----------
SA_HD bool sa_too_far(const sa_geo& l, const sa_geo& r) {
  float max_distance = l.r + r.r;
  float x = l.xc - r.xc, y = l.yc - r.yc;
  return x * x + y * y > max_distance * max_distance;
}
SA_HD float sa_dist_in_2r(const sa_geo& l, const sa_geo& r) {
  float radial = l.r + r.r;
  float x = l.xc - r.xc, y = l.yc - r.yc;
  return sqrtf(x * x + y * y) / sqrtf(radial * radial + SA_EPS);
}
SA_HD bool sa_compatible(const sa_geo& c, uint64_t ce, const sa_geo& t, uint64_t te, uint64_t max_idle,
                         const sa_constraints& cons) {
  uint64_t delta = ce > te ? ce - te : te - ce;
  if (!(max_idle >= delta)) return false;
  // validate(): the first constraint with delta_i >= epoch_delta decides; none -> true.  dist_in_2r (a sqrt and a
  // division) is only evaluated when a constraint actually applies — the reference computes it unconditionally
  // but reads it nowhere else.
  for (uint32_t i = 0; i < cons.n; ++i)
    if (cons.delta[i] >= delta) return sa_dist_in_2r(c, t) <= cons.max_dist[i];
  return true;
}

// Two boxes WITHOUT an angle are axis-aligned rectangles; the cell the reference computes for them — Sutherland–Hodgman, shoelace,
// IoU, x confidence, threshold (bbox.rs:512-535, sort/metric.rs:65-77) — is absent whenever
//   * the rectangles do not overlap (intersection exactly 0.0: every subject vertex lies outside one clip edge, the clipper emits
//     nothing), or
//   * an UPPER bound of intersection / union x confidence stays below the threshold.
// Both are decided here in f32 from centres and half extents, with margins (1e-4 of the extents on the overlap, 1e-4 relative on the
// bound) that are orders of magnitude above the f32 rounding of these few operations and of the f64 clip itself: a pair rejected
// here is absent in the reference too; a pair within the margins is NOT rejected and goes through the exact clip.  Bounding-circle
// neighbours that do not overlap, or overlap too little, are the bulk of the surviving pairs of a crowded frame (C2: 19 in 20).
// IoU engines only (the Mahalanobis cell has no such bound).  Returns true = the cell is absent.
SA_HD bool sa_aa_quick_reject(const sa_geo& c, const sa_ext& ce, const sa_geo& t, const sa_ext& te, float conf, float threshold) {
  if (ce.hw < 0.0f || te.hw < 0.0f) return false;  // an oriented box: no shortcut
  const float dx = fabsf(c.xc - t.xc), dy = fabsf(c.yc - t.yc);
  const float sx = ce.hw + te.hw, sy = ce.hh + te.hh;
  const float m = 1e-4f * (sx + sy);
  const float ox = sx - dx, oy = sy - dy;
  if (ox < -m || oy < -m) return true;  // disjoint, with room to spare
  const float wx = 2.0f * (ce.hw < te.hw ? ce.hw : te.hw), wy = 2.0f * (ce.hh < te.hh ? ce.hh : te.hh);
  float ix = (ox > 0.0f ? ox : 0.0f) + m, iy = (oy > 0.0f ? oy : 0.0f) + m;
  ix = ix < wx ? ix : wx;
  iy = iy < wy ? iy : wy;
  const float inter_ub = ix * iy;
  const float uni_lb = (c.hha + t.hha) - inter_ub;
  if (!(uni_lb > 0.0f)) return false;
  return inter_ub / uni_lb * conf * 1.0001f < threshold;
}

// ---- Sutherland–Hodgman + shoelace (clipping.rs:12-91, geo Area) ------------------------------------------
#define SA_POLY_CAP 12
// Clips the open ring `subj` (4 vertices) by the open ring `clip` (4 vertices); returns the unsigned area
// of the result exactly as Polygon::new(...).unsigned_area() evaluates it.  The two ping-pong vertex lists live
// in caller-provided storage (element v of a list at [v * stride]): on the GPU that is an LDS slice per worker
// lane — dynamically indexed per-lane arrays would otherwise be spilled to scratch memory.
SA_HD double sa_clip_area_ws(const double* subj, const double* clip, double* ax, double* ay, double* bx, double* by, int stride) {
  int n = 4;
  for (int i = 0; i < 4; ++i) { ax[i * stride] = subj[2 * i]; ay[i * stride] = subj[2 * i + 1]; }
  double* px = ax; double* py = ay; double* qx = bx; double* qy = by;
  for (int i = 0; i < 4; ++i) {
    int ii = i == 0 ? 3 : i - 1;
    double csx = clip[2 * ii], csy = clip[2 * ii + 1];
    double cex = clip[2 * i], cey = clip[2 * i + 1];
    int m = 0;
    // Two subject vertices per step.  Everything a vertex needs — its own coordinates and its predecessor's — comes from the
    // input list of the pass, so the two chains of dependent f64 operations (side test, crossing point with its division) are
    // independent and overlap; only the output position is carried.  is_inside(s_edge_start) is the previous vertex's
    // is_inside(s_edge_end) — the same expression on the same values — and is carried instead of recomputed.
    double ssx = n ? px[(n - 1) * stride] : 0.0, ssy = n ? py[(n - 1) * stride] : 0.0;  // s_edge_start of j = 0
    bool in_s = ((cex - csx) * (ssy - csy) - (cey - csy) * (ssx - csx)) <= 0.0;
    const double dpx = csx - cex, dpy = csy - cey;
    const double n2 = csx * cey - csy * cex;
    for (int j = 0; j < n; j += 2) {
      const bool has_b = j + 1 < n;
      const double ax_ = px[j * stride], ay_ = py[j * stride];
      const double bx_ = has_b ? px[(j + 1) * stride] : ax_, by_ = has_b ? py[(j + 1) * stride] : ay_;
      const bool in_a = ((cex - csx) * (ay_ - csy) - (cey - csy) * (ax_ - csx)) <= 0.0;
      const bool in_b = ((cex - csx) * (by_ - csy) - (cey - csy) * (bx_ - csx)) <= 0.0;
      // compute_intersection(cp1 = s_edge_start, cp2 = s_edge_end, s = c_edge_start, e = c_edge_end)  clipping.rs:17-38
      const double dcxa = ssx - ax_, dcya = ssy - ay_;
      const double n1a = ssx * ay_ - ssy * ax_;
      const double n3a = 1.0 / (dcxa * dpy - dcya * dpx);
      const double dcxb = ax_ - bx_, dcyb = ay_ - by_;
      const double n1b = ax_ * by_ - ay_ * bx_;
      const double n3b = 1.0 / (dcxb * dpy - dcyb * dpx);
      if (in_a != in_s && m < SA_POLY_CAP) { qx[m * stride] = (n1a * dpx - n2 * dcxa) * n3a; qy[m * stride] = (n1a * dpy - n2 * dcya) * n3a; ++m; }
      if (in_a && m < SA_POLY_CAP) { qx[m * stride] = ax_; qy[m * stride] = ay_; ++m; }
      if (has_b) {
        if (in_b != in_a && m < SA_POLY_CAP) { qx[m * stride] = (n1b * dpx - n2 * dcxb) * n3b; qy[m * stride] = (n1b * dpy - n2 * dcyb) * n3b; ++m; }
        if (in_b && m < SA_POLY_CAP) { qx[m * stride] = bx_; qy[m * stride] = by_; ++m; }
        ssx = bx_; ssy = by_; in_s = in_b;
      } else {
        ssx = ax_; ssy = ay_; in_s = in_a;
      }
    }
    double* t = px; px = qx; qx = t;
    t = py; py = qy; qy = t;
    n = m;
  }
  if (n == 0) return 0.0;
  // Polygon::new closes the ring unless first == last; < 3 coordinates -> 0
  double shx = px[0], shy = py[0];
  bool closed = shx == px[(n - 1) * stride] && shy == py[(n - 1) * stride];
  int m = closed ? n : n + 1;
  if (m < 3) return 0.0;
  double tmp = 0.0;
  double x0 = 0.0, y0 = 0.0;  // ring[0] - shift
  for (int i = 0; i + 1 < m; ++i) {
    int i1 = (i + 1 == n) ? 0 : i + 1;  // the appended closing coordinate is ring[0]
    double x1 = px[i1 * stride] - shx, y1 = py[i1 * stride] - shy;
    tmp = tmp + (x0 * y1 - y0 * x1);
    x0 = x1; y0 = y1;
  }
  double area = tmp / (1.0 + 1.0);
  return fabs(area);
}
// Cheap proof that sa_clip_area_ws(subj, clip) is exactly 0.0, so that the clip need not run: a separating edge, with a margin.
//  * All four subject vertices outside ONE clip edge — the very expression Sutherland–Hodgman evaluates (clipping.rs:12-15:
//    inside = r <= 0): pass i of the clipper then sees only outside vertices and emits nothing; the list stays empty, area 0.0.
//    Passes before i may already have replaced vertices by crossing points; those lie on segments between subject vertices up to
//    ~1e-12 of rounding, hence the margin: the subject must clear the edge's line by more than 1e-6 of the edge length.
//  * All four clip vertices outside one SUBJECT edge: the exact intersection is empty, so some pass k meets a (convex, exact)
//    working polygon that lies wholly outside its half-plane and empties the list; floating point can only disagree when a
//    vertex of that working polygon comes within rounding (~1e-12) of line k, or — in either case — when a crossing is
//    computed for a subject edge that straddles a clip line while parallel to it within ~1e-9 rad without being bit-identical
//    to it.  Neither is a configuration float32 boxes produce other than by construction (and same-angle or axis-aligned boxes
//    are always decided by the first rule, where the argument is exact); in them the reference itself returns rounding noise.
// Bounding-circle neighbours that do not overlap are the bulk of the surviving pairs in a crowded frame (C2: ~80 %).
// (statically indexed and without pointer selects, so that register-resident polygons stay in registers on the device)
SA_HD bool sa_quad_clears_an_edge(const double* f, const double* g) {  // is g wholly outside one edge of f, with the margin
  bool any = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < 4; ++i) {
    const int ii = i == 0 ? 3 : i - 1;
    const double csx = f[2 * ii], csy = f[2 * ii + 1];
    const double ex = f[2 * i] - csx, ey = f[2 * i + 1] - csy;
    const double l1 = fabs(ex) + fabs(ey);
    const double margin = 1e-6 * l1 * l1;
    bool all_out = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int v = 0; v < 4; ++v) all_out = all_out && (ex * (g[2 * v + 1] - csy) - ey * (g[2 * v] - csx)) > margin;
    any = any || all_out;
  }
  return any;
}
SA_HD bool sa_clip_is_empty(const double* subj, const double* clip) {
  return sa_quad_clears_an_edge(clip, subj) || sa_quad_clears_an_edge(subj, clip);
}
SA_HD double sa_clip_area(const double* subj, const double* clip) {
  double ax[SA_POLY_CAP], ay[SA_POLY_CAP], bx[SA_POLY_CAP], by[SA_POLY_CAP];
  return sa_clip_area_ws(subj, clip, ax, ay, bx, by, 1);
}

// ---- exclusively owned area (clipping/bbox_own_areas.rs:8-46) -----------------------------------------------
// The reference subtracts, one after another, every near box from box i with geo's BooleanOps and takes the area of what is
// left; only that area reaches the tracker.  The device never builds the outline: by Green's theorem the area of
// R = Q_0 \ (Q_1 ∪ … ∪ Q_m) is half the sum of cross(A, B) · (length fraction) over the pieces of polygon edges A→B that
// separate R from its complement —
//   * an edge of Q_0 (the box itself), where it is not inside any Q_k;
//   * an edge of Q_j (j ≥ 1), traversed backwards, where it is inside Q_0 and not inside any other Q_k.
// A segment meets a convex quad in ONE parameter interval (four half-plane cuts), so per edge the work is m intervals and the
// measure of what they leave uncovered.  Edges that lie exactly on the same line (duplicate detections, boxes on a grid) are
// decided by which side the interiors are on, so that every geometric piece of the boundary is emitted exactly once:
//   same_in / opp_in = does an edge of Y that is collinear with the segment, in the same / the opposite direction, count as
//   containing it.  Q_0's edges: (true, false) against every Q_k — just inside Q_0 is inside Q_k only when the interiors are
//   on the same side.  Q_j's edges: (false, false) against Q_0 (Q_0 emits the shared piece itself), (k < j, true) against Q_k —
//   of two coincident edges the lower index emits.
// All polygons are the reference's clockwise quads (bbox.rs:287-330): interior = cross(f1 - f0, p - f0) <= 0 (clipping.rs:12-15).
SA_HD bool sa_seg_in_quad(double ax, double ay, double bx, double by, const double* q, bool same_in, bool opp_in, double* t0,
                          double* t1) {
  double lo = *t0, hi = *t1;
  for (int k = 0; k < 4; ++k) {
    const int k1 = (k + 1) & 3;
    const double f0x = q[2 * k], f0y = q[2 * k + 1];
    const double dx = q[2 * k1] - f0x, dy = q[2 * k1 + 1] - f0y;
    const double ga = dx * (ay - f0y) - dy * (ax - f0x);
    const double gb = dx * (by - f0y) - dy * (bx - f0x);
    if (ga == 0.0 && gb == 0.0) {
      const bool same = dx * (bx - ax) + dy * (by - ay) > 0.0;
      if (!(same ? same_in : opp_in)) return false;
      continue;
    }
    if (ga <= 0.0 && gb <= 0.0) continue;
    if (ga > 0.0 && gb > 0.0) return false;
    const double t = ga / (ga - gb);
    if (ga > 0.0) lo = t > lo ? t : lo;
    else hi = t < hi ? t : hi;
  }
  *t0 = lo;
  *t1 = hi;
  return lo < hi;
}
// Measure of [lo, hi] minus the union of cnt intervals [a_k, b_k] ⊂ [lo, hi] (non-empty, element k at [k * stride]): every gap
// starts at lo or at some b_k; O(cnt^2), cnt is the handful of boxes that actually cross this edge.
SA_HD double sa_uncovered(double lo, double hi, uint32_t cnt, const double* a, const double* b, uint32_t stride) {
  double total = 0.0;
  for (uint32_t si = 0; si <= cnt; ++si) {
    const double s = si == 0 ? lo : b[(si - 1) * stride];
    if (!(s < hi)) continue;
    bool skip = false;
    double next = hi;
    for (uint32_t j = 0; j < cnt; ++j) {
      const double aj = a[j * stride], bj = b[j * stride];
      if (aj <= s && s < bj) { skip = true; break; }                 // s is covered
      if (si > 0 && j < si - 1 && bj == s) { skip = true; break; }   // the same gap start, counted at the lower index
      if (aj > s && aj < next) next = aj;
    }
    if (!skip) total += next - s;
  }
  return total;
}
// Signed contribution (twice the area) of edge ek of polygon xi of the list polys[m1][8] (index 0 = the box itself), using
// iva / ivb (cap elements, stride apart) as interval storage.  Returns NaN when more than cap disjoint stretches of this edge
// are covered (a full list is first fused into disjoint intervals).
SA_HD double sa_own_edge(const double* polys, uint32_t m1, uint32_t xi, uint32_t ek, double* iva, double* ivb, uint32_t stride,
                         uint32_t cap) {
  const double* X = polys + (size_t)xi * 8;
  const uint32_t e1 = (ek + 1) & 3u;
  const double ax = X[2 * ek], ay = X[2 * ek + 1], bx = X[2 * e1], by = X[2 * e1 + 1];
  double lo = 0.0, hi = 1.0;
  if (xi != 0 && !sa_seg_in_quad(ax, ay, bx, by, polys, false, false, &lo, &hi)) return 0.0;
  uint32_t cnt = 0;
  for (uint32_t yi = 1; yi < m1; ++yi) {
    if (yi == xi) continue;
    double t0 = lo, t1 = hi;
    const bool same_in = xi == 0 ? true : yi < xi;
    const bool opp_in = xi != 0;
    if (!sa_seg_in_quad(ax, ay, bx, by, polys + (size_t)yi * 8, same_in, opp_in, &t0, &t1)) continue;
    if (t0 <= lo && t1 >= hi) return 0.0;  // wholly covered
    if (cnt >= cap) {  // storage full: fuse overlapping intervals into disjoint ones (only very crowded edges get here)
      for (uint32_t i = 0; i < cnt; ++i) {
        bool changed = true;
        while (changed) {
          changed = false;
          for (uint32_t j = i + 1; j < cnt; ++j) {
            const double ai = iva[i * stride], bi = ivb[i * stride], aj = iva[j * stride], bj = ivb[j * stride];
            if (aj <= bi && bj >= ai) {
              iva[i * stride] = aj < ai ? aj : ai;
              ivb[i * stride] = bj > bi ? bj : bi;
              --cnt;
              iva[j * stride] = iva[cnt * stride];
              ivb[j * stride] = ivb[cnt * stride];
              changed = true;
              --j;
            }
          }
        }
      }
      if (cnt >= cap) return NAN;  // more than cap DISJOINT covered stretches on one edge
    }
    iva[cnt * stride] = t0;
    ivb[cnt * stride] = t1;
    ++cnt;
  }
  const double mu = sa_uncovered(lo, hi, cnt, iva, ivb, stride);
  const double cr = ax * by - ay * bx;
  return xi == 0 ? cr * mu : -cr * mu;
}
// Conservative "do these two quads overlap" (never false for quads that share area): a separating axis among the 8 edges.
SA_HD bool sa_quads_separated(const double* p, const double* q) {
  for (int pass = 0; pass < 2; ++pass) {
    const double* f = pass ? q : p;
    const double* g = pass ? p : q;
    for (int k = 0; k < 4; ++k) {
      const int k1 = (k + 1) & 3;
      const double f0x = f[2 * k], f0y = f[2 * k + 1], dx = f[2 * k1] - f0x, dy = f[2 * k1 + 1] - f0y;
      bool all_out = true;
      for (int v = 0; v < 4; ++v) all_out = all_out && (dx * (g[2 * v + 1] - f0y) - dy * (g[2 * v] - f0x) > 0.0);
      if (all_out) return true;
    }
  }
  return false;
}

// Universal2DBox::calculate_metric_object (bbox.rs:512-535) for a pair that is not too_far.
SA_HD bool sa_iou_from_area(double inter, float c_hha, float t_hha, float* out) {
  if (inter == 0.0) return false;
  double uni = (double)(c_hha + t_hha) - inter;
  *out = (float)(inter / uni);
  return true;
}
SA_HD bool sa_iou_cell(const double* cand_verts, const double* track_verts, float c_hha, float t_hha, float* out) {
  double inter = sa_clip_area(cand_verts, track_verts);
  if (inter == 0.0) return false;
  double uni = (double)(c_hha + t_hha) - inter;
  *out = (float)(inter / uni);
  return true;
}

// ---- Mahalanobis (kalman_2d_box.rs:104-120, 150-184) -------------------------------------------------------
// Per track, once: projected mean = mean[0..5]; projected cov = cov[0..5,0..5] + diag(std^2);
// L = nalgebra Cholesky::new(cov).l().  out20 = mean5 | L as 15 packed lower-triangular values
// (row-major: L00, L10 L11, L20 L21 L22, ...).  A non-positive pivot poisons L with NaN (the
// reference unwrap()-panics there).
SA_HD void sa_maha_prepare(float pw, const float* mean5, const float* cov25, float* out20) {
  float M[25];
  float sw = 1.0f * pw * mean5[4];
  float sd[5] = {sw, sw, sw, 1e-1f, sw};
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) M[i * 5 + j] = cov25[i * 5 + j] + (i == j ? sd[i] * sd[i] : 0.0f);
  bool ok = true;
  for (int j = 0; j < 5; ++j) {
    for (int k = 0; k < j; ++k) {
      float factor = -M[j * 5 + k];
      for (int r = j; r < 5; ++r) M[r * 5 + j] = factor * M[r * 5 + k] + M[r * 5 + j];
    }
    float diag = M[j * 5 + j];
    if (diag == 0.0f || !(diag >= 0.0f)) { ok = false; break; }
    float denom = sqrtf(diag);
    M[j * 5 + j] = denom;
    for (int r = j + 1; r < 5; ++r) M[r * 5 + j] = M[r * 5 + j] / denom;
  }
  for (int i = 0; i < 5; ++i) out20[i] = mean5[i];
  int o = 5;
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j <= i; ++j) out20[o++] = ok ? M[i * 5 + j] : NAN;
}
// Per cell: d2 = || L^-1 (z - mean) ||^2 with nalgebra's column-oriented forward substitution,
// then calculate_cost(d2, inverted = true) / conf  (sort/metric.rs:54-63).
SA_HD float sa_maha_cell(const float* m20, const float* z5, float conf) {
  float r[5];
  for (int i = 0; i < 5; ++i) r[i] = z5[i] - m20[i];
  const float* L = m20 + 5;
  for (int i = 0; i < 5; ++i) {
    float diag = L[i * (i + 1) / 2 + i];
    float coeff = r[i] / diag;
    r[i] = coeff;
    float nc = -coeff;
    for (int k = i + 1; k < 5; ++k) r[k] = nc * L[k * (k + 1) / 2 + i] + r[k];
  }
  float s = 0.0f;
  for (int i = 0; i < 5; ++i) s = s + r[i] * r[i];
  float cost = s > SA_CHI2INV95_4 ? 0.0f : SA_CHI2_UPPER_BOUND - s;
  return cost / conf;
}

// ---- quantisation (sort/voting.rs:20,59): (w * 1e6f) as i64 — trunc, saturating, NaN -> 0 ---------------------
SA_HD int64_t sa_quantise(float w) {
  float v = w * SA_F32_U64_MULT;
  if (v != v) return 0;
  if (v >= 9223372036854775808.0f) return INT64_MAX;
  if (v <= -9223372036854775808.0f) return INT64_MIN;
  return (int64_t)v;
}

// ---- order-preserving float <-> uint key (for atomicMax over possibly negative floats) ---------------------
SA_HD uint32_t sa_f32_key(float f) {
  union { float f; uint32_t u; } c;
  c.f = f;
  return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
SA_HD float sa_key_f32(uint32_t k) {
  union { float f; uint32_t u; } c;
  c.u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return c.f;
}

// ---- memory access shims: agent-scope relaxed atomics on the device, plain ops in the host emulation ------
#if defined(__HIP_DEVICE_COMPILE__)
SA_HD uint32_t sa_ld_u32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SA_HD void sa_st_u32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SA_HD uint32_t sa_cas_u32(uint32_t* p, uint32_t expect, uint32_t desired) {
  __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return expect;
}
#else
SA_HD uint32_t sa_ld_u32(const uint32_t* p) { return *p; }
SA_HD void sa_st_u32(uint32_t* p, uint32_t v) { *p = v; }
SA_HD uint32_t sa_cas_u32(uint32_t* p, uint32_t expect, uint32_t desired) {
  uint32_t old = *p;
  if (old == expect) *p = desired;
  return old;
}
#endif

// ---- union-find over rows [0,N) and columns [N,N+T): hook the larger root under the smaller, so a
// component's representative is its minimum vertex — always a candidate row.  Lock-free (ECL-CC style). ---
SA_HD uint32_t sa_uf_find(uint32_t* parent, uint32_t v) {
  uint32_t p = sa_ld_u32(parent + v);
  while (p != v) {
    uint32_t gp = sa_ld_u32(parent + p);
    if (gp != p) sa_st_u32(parent + v, gp);  // path halving: any ancestor is a valid parent
    v = p;
    p = gp;
  }
  return v;
}
SA_HD void sa_uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
  for (;;) {
    a = sa_uf_find(parent, a);
    b = sa_uf_find(parent, b);
    if (a == b) return;
    if (a < b) { uint32_t t = a; a = b; b = t; }
    if (sa_cas_u32(parent + a, a, b) == a) return;
  }
}
// The same with the two vertices' parents ALREADY LOADED by the caller (pa = parent[a], pb = parent[b]): a caller that has other
// requests to make for the same edge issues those two loads beside them, and the usual case — both vertices still roots, or one hop from
// one — costs one round trip (the loads) plus the compare-and-swap instead of three dependent ones.
SA_HD uint32_t sa_uf_find_from(uint32_t* parent, uint32_t v, uint32_t p) {
  while (p != v) {
    uint32_t gp = sa_ld_u32(parent + p);
    if (gp != p) sa_st_u32(parent + v, gp);
    v = p;
    p = gp;
  }
  return v;
}
SA_HD void sa_uf_union_from(uint32_t* parent, uint32_t a, uint32_t pa, uint32_t b, uint32_t pb) {
  a = sa_uf_find_from(parent, a, pa);
  b = sa_uf_find_from(parent, b, pb);
  for (;;) {
    if (a == b) return;
    if (a < b) { uint32_t t = a; a = b; b = t; }
    if (sa_cas_u32(parent + a, a, b) == a) return;
    a = sa_uf_find(parent, a);
    b = sa_uf_find(parent, b);
  }
}

// ---- exact sparse assignment of one connected component -----------------------------------------------------
// SortVoting::winners (sort/voting.rs:30-100) maximises sum of quantised weights over an N x (N+T) matrix
// whose diagonal "self" columns carry the threshold.  With threshold > 0 that is the same optimum as:
// maximise sum(gain) over a matching that uses only edges with gain = w - threshold > 0, every unmatched
// row falling back to its private self column.  The thresholded graph splits into
// connected components that are solved independently; this routine solves one, by successive shortest
// augmenting paths (Jonker–Volgenant style Dijkstra on reduced costs c = -gain, all i64), walking only real
// edges.  Every tree row offers its free self column as a terminal, so searches stay local.
// Edge order inside a row is irrelevant (the kernels append edges with atomics): every choice below is made on
// (distance, column index), never on list position.  Row duals may start from ANY value <= -max gain of the row's usable
// edges (here: -max over all its edges, excluded columns included) — reduced costs stay non-negative.
struct sa_assign_ws {
  // edges, row-major with row stride `estride`
  const uint32_t* e_cnt;   // [N]
  const uint32_t* e_col;   // edge e of a row at e_col[e * ecs] / e_gain[e * egs]: two packed arrays (ecs = egs = 1: the LDS pool) or
  const int64_t* e_gain;   // both pointing into one array of 16-byte {gain, col} records (ecs = 4, egs = 2: the lists in HBM)
  uint32_t ecs, egs;       // distance between consecutive edges of a row, in u32 / i64 words
  uint32_t rcs, rgs;       // words per record (1 / 1: packed arrays, 4 / 2: SaEdge): where a row's first edge is, first * rcs / first * rgs
  uint32_t estride;        // records per row
  const uint32_t* e_off;   // nullptr: row r starts at r * estride; else at e_off[r] (edge lists packed into an LDS pool)
  const uint8_t* excluded;   // [T] or nullptr: columns already taken by the visual vote (visual_sort/voting.rs:62-79);
                             // their edges stay in the lists (the edge pass runs beside the visual vote) and are skipped here
  const uint32_t* next_row;  // [N] next row of the same component (ascending), SA_NONE at the end
  int64_t* u;       // [N] row duals   (initialised to -max gain of the row)
  int64_t* v;       // [T] column duals (initialised to 0)
  int32_t* rmatch;  // [N] matched column or -1 (= self)
  int32_t* cmatch;  // [T] matched row or -1
  int64_t* dist;    // [T]
  int32_t* pred;    // [T] tree row that labelled the column
  uint32_t* cstamp; // [T] search id that labelled the column   (0 = never)
  uint32_t* cscan;  // [T] search id that scanned the column
  int32_t* cnext;   // [T] linked list of labelled columns of the current search
  int64_t* rdist;   // [N] distance at which a row entered the tree
  int32_t* rnext;   // [N] linked list of tree rows of the current search
};

SA_HD void sa_assign_relax_row(const sa_assign_ws& w, uint32_t row, int64_t base, uint32_t stamp, int32_t* list_head) {
  uint32_t cnt = w.e_cnt[row];
  const size_t first = w.e_off ? (size_t)w.e_off[row] : (size_t)row * w.estride;
  const uint32_t* cols = w.e_col + first * w.rcs;
  const int64_t* gains = w.e_gain + first * w.rgs;
  int64_t ur = w.u[row];
  for (uint32_t e = 0; e < cnt; ++e) {
    uint32_t j = cols[(size_t)e * w.ecs];
    if (w.excluded && w.excluded[j]) continue;
    if (w.cscan[j] == stamp) continue;
    int64_t d = base + (-gains[(size_t)e * w.egs] - ur - w.v[j]);
    if (w.cstamp[j] != stamp) {
      w.cstamp[j] = stamp;
      w.dist[j] = d;
      w.pred[j] = (int32_t)row;
      w.cnext[j] = *list_head;
      *list_head = (int32_t)j;
    } else if (d < w.dist[j]) {
      w.dist[j] = d;
      w.pred[j] = (int32_t)row;
    }
  }
}

// The first row of a component finds nothing matched and every column dual at 0: its search is "relax the root, take the
// nearest column, which is free" — the row's heaviest usable edge (lowest column on ties), the root dual moving by that
// column's distance, no column dual moving at all.  Doing exactly that from the edge list alone replaces ~40 dependent accesses
// of the general search (lists, stamps, predecessors) by one pass over the row's edges; most components of a tracking frame
// are this one row.
SA_HD void sa_assign_first_row(const sa_assign_ws& w, uint32_t root) {
  const uint32_t cnt = w.e_cnt[root];
  const size_t first = w.e_off ? (size_t)w.e_off[root] : (size_t)root * w.estride;
  const uint32_t* cols = w.e_col + first * w.rcs;
  const int64_t* gains = w.e_gain + first * w.rgs;
  int32_t bj = -1;
  int64_t bg = 0;
  for (uint32_t e = 0; e < cnt; ++e) {
    const uint32_t j = cols[(size_t)e * w.ecs];
    const int64_t g = gains[(size_t)e * w.egs];
    if (w.excluded && w.excluded[j]) continue;
    if (bj < 0 || g > bg || (g == bg && (int32_t)j < bj)) { bj = (int32_t)j; bg = g; }
  }
  if (bj < 0 || bg <= 0) { w.u[root] = 0; return; }  // the self column ends the path at distance -u: u += -u
  w.u[root] = -bg;                                    // u += (-gain - u) - 0
  w.rmatch[root] = bj;
  w.cmatch[bj] = (int32_t)root;
}

SA_HD void sa_assign_component(const sa_assign_ws& w, uint32_t first_row) {
  sa_assign_first_row(w, first_row);
  for (uint32_t root = w.next_row[first_row]; root != SA_NONE; root = w.next_row[root]) {
    const uint32_t stamp = root + 1u;
    int32_t col_list = -1;   // labelled columns of this search
    int32_t row_list = (int32_t)root;
    w.rnext[root] = -1;
    w.rdist[root] = 0;
    int64_t best_term = -w.u[root];  // reduced cost of the root's own self column
    int32_t term_row = (int32_t)root;
    sa_assign_relax_row(w, root, 0, stamp, &col_list);
    int32_t end_col = -1;
    int64_t delta;
    for (;;) {
      // smallest labelled, unscanned column (ties: lowest column index)
      int32_t bj = -1;
      int64_t bd = 0;
      for (int32_t j = col_list; j >= 0; j = w.cnext[j]) {
        if (w.cscan[j] == stamp) continue;
        int64_t d = w.dist[j];
        if (bj < 0 || d < bd || (d == bd && j < bj)) { bj = j; bd = d; }
      }
      if (bj < 0 || bd >= best_term) { delta = best_term; break; }  // a self column ends the path
      w.cscan[bj] = stamp;
      int32_t i = w.cmatch[bj];
      if (i < 0) { end_col = bj; delta = bd; break; }               // free real column
      w.rdist[i] = bd;
      w.rnext[i] = row_list;
      row_list = i;
      int64_t t = bd + (-w.u[i]);
      if (t < best_term) { best_term = t; term_row = i; }
      sa_assign_relax_row(w, (uint32_t)i, bd, stamp, &col_list);
    }
    // dual update: tree rows u += delta - rdist ; scanned columns v += dist - delta
    for (int32_t i = row_list; i >= 0; i = w.rnext[i]) w.u[i] += delta - w.rdist[i];
    for (int32_t j = col_list; j >= 0; j = w.cnext[j])
      if (w.cscan[j] == stamp) w.v[j] += w.dist[j] - delta;
    // augment
    int32_t j;
    if (end_col >= 0) j = end_col;
    else {
      if (term_row == (int32_t)root) continue;  // root keeps its self column
      j = w.rmatch[term_row];                  // term_row falls back to self and frees its column
      w.rmatch[term_row] = -1;
    }
    for (;;) {
      int32_t i = w.pred[j];
      int32_t prev = w.rmatch[i];
      w.rmatch[i] = j;
      w.cmatch[j] = i;
      if (i == (int32_t)root) break;
      j = prev;
    }
  }
}

// =====================================================================================================================
// Group-cooperative solve: the same shortest-augmenting-path search as sa_assign_component, with its two inner loops — "nearest
// labelled, unscanned column" and "relax the edges of the row that just entered the tree" — spread over the G lanes of a group
// (a whole wavefront, or a 16-lane quarter of one), per-row minima by lane reductions, all state in LDS.  One group solves one
// connected component at a time; a component of hundreds of rows (a dense crowd under a low IoU threshold) costs
// O(path steps x (columns + edges) / G) instead of one lane's chain of dependent LDS round trips.
//
// The source below is written once for both worlds: on the device a "lane loop" runs its body once, for this lane; in the host
// emulation (tests/emu) it runs G times, so the very same statements are checked against the dense kuhn_munkres on a machine
// without a GPU.  Per-lane values that cross a reduction sit in arrays of SA_COOP_SLOTS(G) elements (1 on the device).
// Choices are made on (distance, column index) exactly like the serial search, rows are relaxed one at a time in the order they
// enter the tree, so the result is the serial solver's, whatever G is.
// =====================================================================================================================
#if defined(__HIPCC__)  // both passes of hipcc: device functions (the host pass only parses them)
#define SA_COOP_FN __device__ __forceinline__
#define SA_COOP_SLOTS(G) 1
#define SA_COOP_SLOT(l) 0
#define SA_COOP_FOR(G, l) for (uint32_t l = (uint32_t)(__lane_id() & ((G)-1)), _sa_once = 1; _sa_once; _sa_once = 0)
template <int G>
__device__ __forceinline__ void sa_coop_sync() {  // LDS traffic of one wave is processed in order: only the compiler has to be held
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// lexicographic minimum of (d, j) over the group; every lane receives it
template <int G>
__device__ __forceinline__ void sa_coop_min(const int64_t* d, const int32_t* j, int64_t* od, int32_t* oj) {
  int64_t bd = d[0];
  int32_t bj = j[0];
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    const int64_t xd = __shfl_xor(bd, o, G);
    const int32_t xj = __shfl_xor(bj, o, G);
    const bool take = xj >= 0 && (bj < 0 || xd < bd || (xd == bd && xj < bj));
    bd = take ? xd : bd;
    bj = take ? xj : bj;
  }
  *od = bd;
  *oj = bj;
}
// rank of this lane among the lanes of its group that raise `flag`, and how many do
template <int G>
__device__ __forceinline__ uint32_t sa_coop_rank(const bool* flag, uint32_t l, uint32_t* total) {
  const unsigned long long m = __ballot(flag[0]);
  const uint32_t base = (uint32_t)__lane_id() & ~(uint32_t)(G - 1);
  const unsigned long long gm = G == 64 ? m : (m >> base) & ((1ull << (G & 63)) - 1ull);
  *total = (uint32_t)__popcll(gm);
  return (uint32_t)__popcll(gm & ((1ull << l) - 1ull));
}
template <int G, class T>
__device__ __forceinline__ T sa_coop_bcast(const T* v) {  // lane 0's value to the whole group
  return __shfl(v[0], 0, G);
}
#else
#define SA_COOP_FN inline
#define SA_COOP_SLOTS(G) (G)
#define SA_COOP_SLOT(l) (l)
#define SA_COOP_FOR(G, l) for (uint32_t l = 0; l < (uint32_t)(G); ++l)
template <int G>
inline void sa_coop_sync() {}
template <int G>
inline void sa_coop_min(const int64_t* d, const int32_t* j, int64_t* od, int32_t* oj) {
  int64_t bd = 0;
  int32_t bj = -1;
  for (int l = 0; l < G; ++l)
    if (j[l] >= 0 && (bj < 0 || d[l] < bd || (d[l] == bd && j[l] < bj))) { bd = d[l]; bj = j[l]; }
  *od = bd;
  *oj = bj;
}
template <int G>
inline uint32_t sa_coop_rank(const bool* flag, uint32_t l, uint32_t* total) {
  uint32_t r = 0, t = 0;
  for (uint32_t k = 0; k < (uint32_t)G; ++k) {
    if (flag[k]) { if (k < l) ++r; ++t; }
  }
  *total = t;
  return r;
}
template <int G, class T>
inline T sa_coop_bcast(const T* v) { return v[0]; }
#endif

struct sa_coop_ws {
  // usable edges of the rows: as in sa_assign_ws (packed LDS pool, or the 16-byte records the positional tiles left in HBM)
  const uint32_t* e_cnt;
  const uint32_t* e_col;
  const int64_t* e_gain;
  uint32_t ecs, egs, rcs, rgs, estride;
  const uint32_t* e_off;
  const uint8_t* excluded;
  int64_t* u;        // [N] row duals: -(heaviest usable gain) on entry
  int64_t* v;        // [T] column duals: 0 on entry
  int32_t* rmatch;   // [N] -1, or the column the greedy start gave the row
  int32_t* cmatch;   // [T]
  int64_t* dist;     // [T]
  int32_t* pred;     // [T]
  uint32_t* cstamp;  // [T] 0 on entry
  uint32_t* cscan;   // [T] 0 on entry
  uint32_t* clist;   // room for every column of the component: the labelled columns of the running search
};

// Where the solver's state lives.  sa_mem_plain: LDS (or host memory in the emulation) — a wave's LDS traffic is processed in order,
// only the compiler has to be held.  sa_mem_agent (device only): HBM shared by the lanes of ONE wave — every access goes to L2
// (relaxed, agent scope: never a stale line of the CU's L1) and a sync point waits for the wave's outstanding stores.
template <int G>
struct sa_mem_plain {
  template <class T> static SA_COOP_FN T ld(const T* p) { return *p; }
  template <class T> static SA_COOP_FN void st(T* p, T v) { *p = v; }
  static SA_COOP_FN void sync() { sa_coop_sync<G>(); }
};
#if defined(__HIPCC__)
// sa_mem_wg (device only): part of the state in HBM, touched by the waves of ONE workgroup (k_assign_small2: a search's labels and
// distances live there, the duals and matches in LDS) — relaxed workgroup-scope accesses: the waves of a workgroup share their CU's vector
// memory pipeline and its L1, which takes their accesses in order, so nothing is flushed and nothing is waited for beyond the compiler's order.
template <int G>
struct sa_mem_wg {
  template <class T> static __device__ __forceinline__ T ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  template <class T> static __device__ __forceinline__ void st(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  static __device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
};
template <int G>
struct sa_mem_agent {
  template <class T> static __device__ __forceinline__ T ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  template <class T> static __device__ __forceinline__ void st(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  static __device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    __builtin_amdgcn_wave_barrier();
  }
};
#endif

// Relax the usable edges of `row` (which entered the tree at distance `base`): G edges per step.  Columns labelled for the first
// time in this search are appended to clist (ballot + prefix: append order = edge order, irrelevant to any decision).
template <int G, class M = sa_mem_plain<G>>
SA_COOP_FN void sa_coop_relax(const sa_coop_ws& w, uint32_t row, int64_t base, uint32_t stamp, uint32_t* len) {
  const uint32_t cnt = w.e_cnt[row];
  const size_t first = w.e_off ? (size_t)w.e_off[row] : (size_t)row * w.estride;
  const uint32_t* cols = w.e_col + first * w.rcs;
  const int64_t* gains = w.e_gain + first * w.rgs;
  const int64_t ur = M::ld(w.u + row);
  for (uint32_t e0 = 0; e0 < cnt; e0 += G) {
    bool fresh[SA_COOP_SLOTS(G)];
    uint32_t col[SA_COOP_SLOTS(G)];
    SA_COOP_FOR(G, l) {
      const uint32_t e = e0 + l;
      bool f = false;
      uint32_t j = 0;
      if (e < cnt) {
        j = cols[(size_t)e * w.ecs];
        if (!(w.excluded && w.excluded[j]) && M::ld(w.cscan + j) != stamp) {
          const int64_t d = base + (-gains[(size_t)e * w.egs] - ur - M::ld(w.v + j));
          if (M::ld(w.cstamp + j) != stamp) {
            M::st(w.cstamp + j, stamp);
            M::st(w.dist + j, d);
            M::st(w.pred + j, (int32_t)row);
            f = true;
          } else if (d < M::ld(w.dist + j)) {
            M::st(w.dist + j, d);
            M::st(w.pred + j, (int32_t)row);
          }
        }
      }
      fresh[SA_COOP_SLOT(l)] = f;
      col[SA_COOP_SLOT(l)] = j;
    }
    uint32_t added = 0;
    SA_COOP_FOR(G, l) {
      uint32_t tot;
      const uint32_t r = sa_coop_rank<G>(fresh, l, &tot);
      if (fresh[SA_COOP_SLOT(l)]) M::st(w.clist + (*len + r), col[SA_COOP_SLOT(l)]);
      added = tot;
    }
    *len += added;
    M::sync();
  }
}

// Solves one component: `roots` = its rows the greedy start left unmatched (ascending), n_roots of them.  Rows matched by the
// greedy start hold their heaviest usable edge (tight under u = -max gain, v = 0), so the duals are feasible on entry.
template <int G, class M = sa_mem_plain<G>>
SA_COOP_FN void sa_assign_component_coop(const sa_coop_ws& w, const uint32_t* roots, uint32_t n_roots) {
  for (uint32_t ri = 0; ri < n_roots; ++ri) {
    const uint32_t root = roots[ri];
    const uint32_t stamp = root + 1u;
    uint32_t len = 0;
    int64_t best_term = -M::ld(w.u + root);  // reduced cost of the root's own self column
    int32_t term_row = (int32_t)root;
    int32_t end_col = -1;
    int64_t delta = best_term;
    sa_coop_relax<G, M>(w, root, 0, stamp, &len);
    // (every pass of the loop scans one more column of the component, every step of the augmentation walks one more tree row: the
    // caps below can only bite if the state were corrupted — then the kernel ends with a wrong answer the tests catch instead of
    // spinning on a GPU box)
    for (uint32_t guard = 0; guard < 65536u; ++guard) {
      // nearest labelled, unscanned column (ties: lowest column index): one strided pass + a lane reduction
      int64_t pd[SA_COOP_SLOTS(G)];
      int32_t pj[SA_COOP_SLOTS(G)];
      SA_COOP_FOR(G, l) {
        int64_t bd = 0;
        int32_t bj = -1;
        for (uint32_t k = l; k < len; k += G) {
          const int32_t j = (int32_t)M::ld(w.clist + k);
          if (M::ld(w.cscan + j) == stamp) continue;
          const int64_t d = M::ld(w.dist + j);
          if (bj < 0 || d < bd || (d == bd && j < bj)) { bd = d; bj = j; }
        }
        pd[SA_COOP_SLOT(l)] = bd;
        pj[SA_COOP_SLOT(l)] = bj;
      }
      int64_t bd;
      int32_t bj;
      sa_coop_min<G>(pd, pj, &bd, &bj);
      if (bj < 0 || bd >= best_term) { delta = best_term; break; }  // a self column ends the path
      SA_COOP_FOR(G, l) { if (l == 0) M::st(w.cscan + bj, stamp); }
      const int32_t i = M::ld(w.cmatch + bj);
      if (i < 0) { end_col = bj; delta = bd; break; }               // free real column
      const int64_t t = bd + (-M::ld(w.u + i));
      if (t < best_term) { best_term = t; term_row = i; }
      M::sync();
      sa_coop_relax<G, M>(w, (uint32_t)i, bd, stamp, &len);
    }
    M::sync();
    // dual update.  A tree row other than the root entered through the scanned column it is matched to, at that column's
    // distance: u[cmatch[j]] += delta - dist[j], v[j] += dist[j] - delta over the scanned columns; the root moves by delta.
    SA_COOP_FOR(G, l) {
      for (uint32_t k = l; k < len; k += G) {
        const uint32_t j = M::ld(w.clist + k);
        if (M::ld(w.cscan + j) != stamp) continue;
        const int64_t dj = M::ld(w.dist + j);
        M::st(w.v + j, M::ld(w.v + j) + (dj - delta));
        const int32_t i = M::ld(w.cmatch + j);
        if (i >= 0) M::st(w.u + i, M::ld(w.u + i) + (delta - dj));
      }
      if (l == 0) M::st(w.u + root, M::ld(w.u + root) + delta);
    }
    M::sync();
    // augment (a short dependent chain: every lane walks it, lane 0 writes)
    int32_t j;
    if (end_col >= 0) j = end_col;
    else {
      if (term_row == (int32_t)root) continue;  // root keeps its self column
      j = M::ld(w.rmatch + term_row);           // term_row falls back to self and frees its column
      M::sync();
      SA_COOP_FOR(G, l) { if (l == 0) M::st(w.rmatch + term_row, (int32_t)-1); }
    }
    for (uint32_t guard = 0; guard < 65536u; ++guard) {
      const int32_t i = M::ld(w.pred + j);
      const int32_t prev = M::ld(w.rmatch + i);
      M::sync();
      SA_COOP_FOR(G, l) { if (l == 0) { M::st(w.rmatch + i, j); M::st(w.cmatch + j, i); } }
      if (i == (int32_t)root || prev < 0) break;
      j = prev;
    }
    M::sync();
  }
}
