// emu.cpp — TEST INFRASTRUCTURE.  Compiles the product's scalar device logic (similari_amd/csrc/sa_device.h,
// the exact source the HIP kernels call) with g++ so its arithmetic and the sparse assignment solver can be
// checked against the oracle on a machine without a GPU.  The product never links this file.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/similari_assoc.h"
#include "../../similari_amd/csrc/sa_device.h"
#include "../../similari_amd/csrc/sa_dense.h"

namespace {
void prep(const sa_box& b, sa_geo* g, double* verts) {
  g->xc = b.xc;
  g->yc = b.yc;
  g->r = sa_radius(b.aspect, b.height);
  g->hha = b.height * b.height * b.aspect;
  double a = (double)(b.has_angle ? b.angle : 0.0f);
  double c = 1.0, s = 0.0;
  if (a != 0.0) ::sincos(a, &s, &c);  // as the engine's fill_raw and the oracle's or_vertices
  sa_vertices(b.xc, b.yc, b.aspect, b.height, c, s, verts);
}
sa_constraints make_cons(const sa_config* cfg) {
  sa_constraints c;
  std::memset(&c, 0, sizeof c);
  c.n = cfg->n_constraints;
  for (uint32_t i = 0; i < c.n; ++i) {
    c.delta[i] = cfg->constraint_epoch_delta[i];
    c.max_dist[i] = cfg->constraint_max_dist[i];
  }
  return c;
}
template <int G>
void coop_solve_all(const sa_coop_ws& w, const std::vector<std::vector<uint32_t>>& comps) {
  for (const auto& roots : comps)
    if (!roots.empty()) sa_assign_component_coop<G>(w, roots.data(), (uint32_t)roots.size());
}
template <int NT, int CPT>
void dense_solve_all(const sa_dense_ws& w, const sa_coop_ws& cw, const std::vector<std::vector<uint32_t>>& comps, uint32_t min_roots, bool k32) {
  for (const auto& roots : comps) {
    if (roots.empty()) continue;
    if (roots.size() >= min_roots) {
      if (k32) sa_assign_component_dense<NT, CPT, true>(w, roots.data(), (uint32_t)roots.size());
      else sa_assign_component_dense<NT, CPT, false>(w, roots.data(), (uint32_t)roots.size());
    } else sa_assign_component_coop<64>(cw, roots.data(), (uint32_t)roots.size());
  }
}
}  // namespace

extern "C" {

// k_positional for one cell, exactly as the kernel sequences it. Returns 1 and *out when present.
int emu_positional_cell(const sa_config* cfg, const sa_box* cand, uint64_t cand_epoch, const sa_box* track,
                        uint64_t track_epoch, const float* mean5, const float* cov25, float* out, int* compatible) {
  sa_geo cg, tg;
  double cv[8], tv[8];
  prep(*cand, &cg, cv);
  prep(*track, &tg, tv);
  sa_constraints cons = make_cons(cfg);
  bool comp = sa_compatible(cg, cand_epoch, tg, track_epoch, cfg->max_idle_epochs, cons);
  if (compatible) *compatible = comp;
  if (!comp || sa_too_far(cg, tg)) return 0;
  float conf = cand->confidence < cfg->positional_min_confidence ? cfg->positional_min_confidence : cand->confidence;
  if (cfg->positional_kind != SA_POS_MAHALANOBIS) {
    // the positional tiles' axis-aligned quick reject (sa_frame.h, phase 1): conservative — whatever it rejects the exact path rejects too
    const sa_ext ce = sa_box_ext(cand->aspect, cand->height, cand->has_angle && cand->angle != 0.0f);
    const sa_ext te = sa_box_ext(track->aspect, track->height, track->has_angle && track->angle != 0.0f);
    if (sa_aa_quick_reject(cg, ce, tg, te, conf, cfg->positional_threshold)) return 0;
  }
  if (cfg->positional_kind == SA_POS_MAHALANOBIS) {
    float m20[20];
    sa_maha_prepare(cfg->kf_position_weight, mean5, cov25, m20);
    float z5[5] = {cand->xc, cand->yc, cand->has_angle ? cand->angle : 0.0f, cand->aspect, cand->height};
    *out = sa_maha_cell(m20, z5, conf);
    return 1;
  }
  float iou;
  double ws[4 * SA_POLY_CAP * 3];  // element v of list k at ws[k * 36 + v * 3]: the strided layout the kernel uses in LDS
  double inter = sa_clip_area_ws(cv, tv, ws, ws + 36, ws + 72, ws + 108, 3);
  if (inter != sa_clip_area(cv, tv)) return -1;
  if (!sa_iou_from_area(inter, cg.hha, tg.hha, &iou)) return 0;
  float e = iou * conf;
  if (!(e >= cfg->positional_threshold)) return 0;
  *out = e;
  return 1;
}

int64_t emu_quantise(float w) { return sa_quantise(w); }
uint32_t emu_f32_key(float f) { return sa_f32_key(f); }
float emu_key_f32(uint32_t k) { return sa_key_f32(k); }

// Runs the assignment stages sequentially on a dense positional matrix pos[N][T] (NaN = absent), the way the kernels
// sequence them: k_positional turns EVERY cell that beats the threshold into an edge (rows and columns the visual vote has
// taken included — it runs beside that vote), appending in no particular order (atomics; mimicked here by a deterministic
// shuffle); label / next / solve then ignore rows in row_skip and skip columns in col_skip while relaxing.
// rmatch[N] = column or -1.
int emu_assign(uint32_t N, uint32_t T, const float* pos, int64_t threshold_q, const uint8_t* row_skip,
               const uint8_t* col_skip, int32_t* rmatch_out, int64_t* total_gain) {
  std::vector<uint32_t> parent(N + T), label(N, SA_NONE), next_row(N, SA_NONE), e_cnt(N, 0);
  std::vector<uint8_t> not_first(N, 0);
  const uint32_t estride = T ? T : 1;
  std::vector<uint32_t> e_col((size_t)N * estride);
  std::vector<int64_t> e_gain((size_t)N * estride);
  std::vector<int64_t> u(N, 0), v(T, 0), dist(T, 0), rdist(N, 0);
  std::vector<int32_t> rmatch(N, -1), cmatch(T, -1), pred(T, 0), cnext(T, 0), rnext(N, 0);
  std::vector<uint32_t> cstamp(T, 0), cscan(T, 0);
  for (uint32_t i = 0; i < N + T; ++i) parent[i] = i;
  uint64_t lcg = 0x9e3779b97f4a7c15ull;
  for (uint32_t q = 0; q < N; ++q) {
    uint32_t cnt = 0;
    int64_t maxg = 0;
    for (uint32_t t = 0; t < T; ++t) {
      float w = pos[(size_t)q * T + t];
      if (!(w == w)) continue;
      int64_t gain = sa_quantise(w) - threshold_q;
      if (gain > 0) {
        e_col[(size_t)q * estride + cnt] = t;
        e_gain[(size_t)q * estride + cnt] = gain;
        ++cnt;
        if (gain > maxg) maxg = gain;
        sa_uf_union(parent.data(), q, N + t);
      }
    }
    for (uint32_t a = cnt; a > 1; --a) {  // Fisher–Yates: arbitrary append order
      lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
      uint32_t b = (uint32_t)((lcg >> 33) % a);
      std::swap(e_col[(size_t)q * estride + a - 1], e_col[(size_t)q * estride + b]);
      std::swap(e_gain[(size_t)q * estride + a - 1], e_gain[(size_t)q * estride + b]);
    }
    e_cnt[q] = cnt;
    u[q] = -maxg;
  }
  for (uint32_t q = 0; q < N; ++q) label[q] = (e_cnt[q] && !(row_skip && row_skip[q])) ? sa_uf_find(parent.data(), q) : SA_NONE;
  for (uint32_t q = 0; q < N; ++q) {
    if (label[q] == SA_NONE) continue;
    for (uint32_t r = q + 1; r < N; ++r)
      if (label[r] == label[q]) { next_row[q] = r; not_first[r] = 1; break; }
  }
  sa_assign_ws w;
  w.e_cnt = e_cnt.data(); w.e_col = e_col.data(); w.e_gain = e_gain.data(); w.ecs = 1; w.egs = 1; w.rcs = 1; w.rgs = 1; w.estride = estride; w.e_off = nullptr;
  w.excluded = col_skip;
  w.next_row = next_row.data();
  w.u = u.data(); w.v = v.data(); w.rmatch = rmatch.data(); w.cmatch = cmatch.data();
  w.dist = dist.data(); w.pred = pred.data(); w.cstamp = cstamp.data(); w.cscan = cscan.data(); w.cnext = cnext.data();
  w.rdist = rdist.data(); w.rnext = rnext.data();
  for (uint32_t q = 0; q < N; ++q)
    if (label[q] != SA_NONE && !not_first[q]) sa_assign_component(w, q);
  int64_t tot = 0;
  for (uint32_t q = 0; q < N; ++q) {
    rmatch_out[q] = rmatch[q];
    if (rmatch[q] >= 0) {
      if (row_skip && row_skip[q]) return -7;                 // a visually decided row took part
      if (col_skip && col_skip[rmatch[q]]) return -8;         // an excluded column was used
      for (uint32_t e = 0; e < e_cnt[q]; ++e)
        if ((int32_t)e_col[(size_t)q * estride + e] == rmatch[q]) tot += e_gain[(size_t)q * estride + e];
      if (cmatch[rmatch[q]] != (int32_t)q) return -1;  // inconsistent matching
    }
  }
  // dual feasibility + complementary slackness over the usable graph = proof of optimality
  for (uint32_t q = 0; q < N; ++q) {
    if (row_skip && row_skip[q]) continue;
    if (u[q] > 0) return -2;
    bool usable = false;
    for (uint32_t e = 0; e < e_cnt[q]; ++e) {
      uint32_t t = e_col[(size_t)q * estride + e];
      if (col_skip && col_skip[t]) continue;
      usable = true;
      int64_t rc = -e_gain[(size_t)q * estride + e] - u[q] - v[t];
      if (rc < 0) return -3;
      if (rmatch[q] == (int32_t)t && rc != 0) return -4;
    }
    if (rmatch[q] < 0 && usable && u[q] != 0) return -5;  // self column must be tight when used
  }
  for (uint32_t t = 0; t < T; ++t)
    if (cmatch[t] < 0 && v[t] != 0) return -6;  // free columns keep their initial dual
  if (total_gain) *total_gain = tot;
  return 0;
}

// The one-workgroup tail with the group-cooperative solver, as k_assign_small sequences it: usable edges (rows the visual vote
// decided take no part; edges to excluded columns are dropped while packing, or — hbm_lists — stay in the lists and are skipped
// through w.excluded), union-find, duals u = -(heaviest usable gain), GREEDY START (every row bids for the column of its heaviest
// usable edge, lowest column on ties; a column goes to the lowest row that bids for it), the rows left unmatched become the search
// roots of their component (ascending), and sa_assign_component_coop<G> runs once per component that has any.
int emu_assign_coop(uint32_t N, uint32_t T, const float* pos, int64_t threshold_q, const uint8_t* row_skip, const uint8_t* col_skip,
                    int G, int hbm_lists, int32_t* rmatch_out, int64_t* total_gain) {
  std::vector<uint32_t> parent(N + T), e_cnt(N, 0), e_off(N, 0);
  std::vector<uint32_t> e_col;
  std::vector<int64_t> e_gain;
  std::vector<int64_t> u(N, 0), v(T, 0), dist(T, 0);
  std::vector<int32_t> rmatch(N, -1), cmatch(T, -1), pred(T, 0), bcol(N, -1);
  std::vector<uint32_t> cstamp(T, 0), cscan(T, 0), clist(T ? T : 1, 0), cwin(T, SA_NONE);
  for (uint32_t i = 0; i < N + T; ++i) parent[i] = i;
  uint64_t lcg = 0x2545f4914f6cdd1dull;
  for (uint32_t q = 0; q < N; ++q) {
    e_off[q] = (uint32_t)e_col.size();
    if (row_skip && row_skip[q]) continue;
    int64_t maxg = 0;
    for (uint32_t t = 0; t < T; ++t) {
      float wv = pos[(size_t)q * T + t];
      if (!(wv == wv)) continue;
      int64_t gain = sa_quantise(wv) - threshold_q;
      if (gain <= 0) continue;
      const bool excl = col_skip && col_skip[t];
      if (excl && !hbm_lists) continue;       // dropped while packing the LDS pool
      e_col.push_back(t);
      e_gain.push_back(gain);
      if (excl) continue;                      // stays in the list, takes no part
      if (gain > maxg || (gain == maxg && (bcol[q] < 0 || (int32_t)t < bcol[q]))) { maxg = gain; bcol[q] = (int32_t)t; }
      sa_uf_union(parent.data(), q, N + t);
    }
    const uint32_t cnt = (uint32_t)e_col.size() - e_off[q];
    for (uint32_t a = cnt; a > 1; --a) {  // arbitrary append order
      lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
      uint32_t b = (uint32_t)((lcg >> 33) % a);
      std::swap(e_col[e_off[q] + a - 1], e_col[e_off[q] + b]);
      std::swap(e_gain[e_off[q] + a - 1], e_gain[e_off[q] + b]);
    }
    e_cnt[q] = cnt;
    u[q] = -maxg;
  }
  // greedy start
  for (uint32_t q = 0; q < N; ++q)
    if (bcol[q] >= 0 && q < cwin[bcol[q]]) cwin[bcol[q]] = q;
  std::vector<std::vector<uint32_t>> comps(N);
  for (uint32_t q = 0; q < N; ++q) {
    if (bcol[q] < 0) continue;
    if (cwin[bcol[q]] == q) { rmatch[q] = bcol[q]; cmatch[bcol[q]] = (int32_t)q; }
    else comps[sa_uf_find(parent.data(), q)].push_back(q);  // ascending: q runs upwards
  }
  sa_coop_ws w;
  w.e_cnt = e_cnt.data(); w.e_col = e_col.data(); w.e_gain = e_gain.data(); w.ecs = 1; w.egs = 1; w.rcs = 1; w.rgs = 1; w.estride = 0;
  w.e_off = e_off.data();
  w.excluded = hbm_lists ? col_skip : nullptr;
  w.u = u.data(); w.v = v.data(); w.rmatch = rmatch.data(); w.cmatch = cmatch.data(); w.dist = dist.data(); w.pred = pred.data();
  w.cstamp = cstamp.data(); w.cscan = cscan.data(); w.clist = clist.data();  // one search at a time here: the whole array is its segment
  switch (G) {
    case 4: coop_solve_all<4>(w, comps); break;
    case 16: coop_solve_all<16>(w, comps); break;
    case 64: coop_solve_all<64>(w, comps); break;
    default: return -20;
  }
  int64_t tot = 0;
  for (uint32_t q = 0; q < N; ++q) {
    rmatch_out[q] = rmatch[q];
    if (rmatch[q] >= 0) {
      if (row_skip && row_skip[q]) return -7;
      if (col_skip && col_skip[rmatch[q]]) return -8;
      bool found = false;
      for (uint32_t e = 0; e < e_cnt[q]; ++e)
        if ((int32_t)e_col[e_off[q] + e] == rmatch[q]) { tot += e_gain[e_off[q] + e]; found = true; }
      if (!found) return -9;
      if (cmatch[rmatch[q]] != (int32_t)q) return -1;
    }
  }
  for (uint32_t q = 0; q < N; ++q) {  // dual feasibility + complementary slackness over the usable graph = proof of optimality
    if (row_skip && row_skip[q]) continue;
    if (u[q] > 0) return -2;
    bool usable = false;
    for (uint32_t e = 0; e < e_cnt[q]; ++e) {
      uint32_t t = e_col[e_off[q] + e];
      if (col_skip && col_skip[t]) continue;
      usable = true;
      int64_t rc = -e_gain[e_off[q] + e] - u[q] - v[t];
      if (rc < 0) return -3;
      if (rmatch[q] == (int32_t)t && rc != 0) return -4;
    }
    if (rmatch[q] < 0 && usable && u[q] != 0) return -5;
  }
  for (uint32_t t = 0; t < T; ++t)
    if (cmatch[t] < 0 && v[t] != 0) return -6;
  if (total_gain) *total_gain = tot;
  return 0;
}


// The workgroup-cooperative DENSE solver (sa_dense.h), sequenced the way the assignment tails do for a component they hand to it:
// usable edges, union-find, duals u = -(heaviest usable gain), greedy start, and then — for EVERY component with search roots
// (min_roots = 1) or only those with at least min_roots of them (the others go to the wavefront-cooperative solver, as on the
// device) — the component's gains scattered into a dense N x T matrix (excluded columns never written), its roots ascending,
// sa_assign_component_dense<NT, CPT, K32>, and the matrix wiped again.  nt_cpt: 1 = 256 threads x 4 columns, 2 = 16 x 8, 3 = 256 x 8, 4 = 64 x 2 / 5 = 64 x 4 on 32-bit cells (T <= 128 / 256: the one-wavefront form);
// + 16 = the 64-bit variant even where the 32-bit one applies.
int emu_assign_dense(uint32_t N, uint32_t T, const float* pos, int64_t threshold_q, const uint8_t* row_skip, const uint8_t* col_skip,
                     int nt_cpt, uint32_t min_roots, int32_t* rmatch_out, int64_t* total_gain) {
  std::vector<uint32_t> parent(N + T), e_cnt(N, 0), e_off(N, 0);
  std::vector<uint32_t> e_col;
  std::vector<int64_t> e_gain;
  std::vector<int64_t> u(N, 0), v(T, 0), dist(T, 0);
  std::vector<int32_t> rmatch(N, -1), cmatch(T, -1), pred(T, 0), bcol(N, -1);
  std::vector<uint32_t> cstamp(T, 0), cscan(T, 0), clist(T ? T : 1, 0), cwin(T, SA_NONE);
  std::vector<int64_t> dense((size_t)N * (T ? T : 1), 0);
  for (uint32_t i = 0; i < N + T; ++i) parent[i] = i;
  for (uint32_t q = 0; q < N; ++q) {
    e_off[q] = (uint32_t)e_col.size();
    if (row_skip && row_skip[q]) continue;
    int64_t maxg = 0;
    for (uint32_t t = 0; t < T; ++t) {
      float wv = pos[(size_t)q * T + t];
      if (!(wv == wv)) continue;
      int64_t gain = sa_quantise(wv) - threshold_q;
      if (gain <= 0 || (col_skip && col_skip[t])) continue;
      e_col.push_back(t);
      e_gain.push_back(gain);
      dense[(size_t)q * T + t] = gain;
      if (gain > maxg || (gain == maxg && (bcol[q] < 0 || (int32_t)t < bcol[q]))) { maxg = gain; bcol[q] = (int32_t)t; }
      sa_uf_union(parent.data(), q, N + t);
    }
    e_cnt[q] = (uint32_t)e_col.size() - e_off[q];
    u[q] = -maxg;
  }
  for (uint32_t q = 0; q < N; ++q)
    if (bcol[q] >= 0 && q < cwin[bcol[q]]) cwin[bcol[q]] = q;
  std::vector<std::vector<uint32_t>> comps(N);
  for (uint32_t q = 0; q < N; ++q) {
    if (bcol[q] < 0) continue;
    if (cwin[bcol[q]] == q) { rmatch[q] = bcol[q]; cmatch[bcol[q]] = (int32_t)q; }
    else comps[sa_uf_find(parent.data(), q)].push_back(q);
  }
  sa_dense_ws w;
  w.gain = dense.data(); w.ld = T; w.T = T;
  w.u = u.data(); w.rmatch = rmatch.data(); w.cmatch = cmatch.data(); w.pred = pred.data(); w.part = nullptr;
  sa_coop_ws cw;
  cw.e_cnt = e_cnt.data(); cw.e_col = e_col.data(); cw.e_gain = e_gain.data(); cw.ecs = 1; cw.egs = 1; cw.rcs = 1; cw.rgs = 1; cw.estride = 0;
  cw.e_off = e_off.data(); cw.excluded = nullptr;
  cw.u = u.data(); cw.v = v.data(); cw.rmatch = rmatch.data(); cw.cmatch = cmatch.data(); cw.dist = dist.data(); cw.pred = pred.data();
  cw.cstamp = cstamp.data(); cw.cscan = cscan.data(); cw.clist = clist.data();
  // the 32-bit variant wherever the device would take it (every gain below SA_DENSE_K32_MAXGAIN, at most 2048 tracks), unless
  // nt_cpt asks for the 64-bit one regardless (+ 16)
  int64_t maxgain = 0;
  for (int64_t g : e_gain) maxgain = g > maxgain ? g : maxgain;
  const bool k32 = !(nt_cpt & 16) && maxgain <= SA_DENSE_K32_MAXGAIN && T <= SA_DENSE_K32_MAXT;
  switch (nt_cpt & 15) {
    case 1: if (T > 256 * 4) return -21; dense_solve_all<256, 4>(w, cw, comps, min_roots, k32); break;
    case 2: if (T > 16 * 8) return -21; dense_solve_all<16, 8>(w, cw, comps, min_roots, k32); break;
    case 3: if (T > 256 * 8) return -21; dense_solve_all<256, 8>(w, cw, comps, min_roots, k32); break;
    case 4: case 5: {  // the one-wavefront form of the general tail's middle tier: 128 (256) columns, 32-bit cells
      const uint32_t ldc = (nt_cpt & 15) == 4 ? 128u : 256u;
      if (T > ldc || !k32) return -21;
      std::vector<int32_t> g32((size_t)N * ldc, 0), cm(ldc, -1), pr(ldc, 0);
      for (uint32_t q = 0; q < N; ++q)
        for (uint32_t t = 0; t < T; ++t) g32[(size_t)q * ldc + t] = (int32_t)dense[(size_t)q * T + t];
      for (uint32_t t = 0; t < T; ++t) cm[t] = cmatch[t];
      sa_dense_ws w2 = w;
      w2.gain = (const int64_t*)g32.data(); w2.ld = ldc; w2.T = ldc; w2.cmatch = cm.data(); w2.pred = pr.data();
      for (const auto& roots : comps) {
        if (roots.empty()) continue;
        if (ldc == 128u) sa_assign_component_dense<64, 2, true, true>(w2, roots.data(), (uint32_t)roots.size());
        else sa_assign_component_dense<64, 4, true, true>(w2, roots.data(), (uint32_t)roots.size());
      }
      for (uint32_t t = 0; t < T; ++t) cmatch[t] = cm[t];
      break;
    }
    default: return -20;
  }
  int64_t tot = 0;
  for (uint32_t q = 0; q < N; ++q) {
    rmatch_out[q] = rmatch[q];
    if (rmatch[q] >= 0) {
      if (row_skip && row_skip[q]) return -7;
      if (col_skip && col_skip[rmatch[q]]) return -8;
      const int64_t g = dense[(size_t)q * T + rmatch[q]];
      if (g <= 0) return -9;
      tot += g;
      if (cmatch[rmatch[q]] != (int32_t)q) return -1;
    }
  }
  // primal feasibility is checked above; the duals of the dense solver's columns live in its threads' registers, so optimality is
  // checked by the caller against the dense kuhn_munkres (total and, on unique optima, the matching itself)
  if (total_gain) *total_gain = tot;
  return 0;
}

// emulation statistics of the dense solver since the last call: search steps, searches (roots), components
void emu_dense_stats(uint64_t* out3) {
  out3[0] = sa_dense_emu_steps; out3[1] = sa_dense_emu_searches; out3[2] = sa_dense_emu_comps;
  sa_dense_emu_steps = sa_dense_emu_searches = sa_dense_emu_comps = 0;
}

// The positional tiles' pre-filter (heterogeneous launch): 1 when sa_clip_is_empty proves the clip of (cand, track) empty.
int emu_clip_is_empty(const sa_box* cand, const sa_box* track) {
  double cv[8], tv[8];
  const sa_box* bs[2] = {cand, track};
  double* vs[2] = {cv, tv};
  for (int k = 0; k < 2; ++k) {
    double a = (double)(bs[k]->has_angle ? bs[k]->angle : 0.0f);
    double c = 1.0, s = 0.0;
    if (a != 0.0) ::sincos(a, &s, &c);
    sa_vertices(bs[k]->xc, bs[k]->yc, bs[k]->aspect, bs[k]->height, c, s, vs[k]);
  }
  return sa_clip_is_empty(cv, tv) ? 1 : 0;
}

// k_own_area for every box of a frame, as the kernel sequences it (neighbour scan, relative coordinates, one call of
// sa_own_edge per polygon edge, the reference's f32 epilogue).  Returns the status bits the kernel would raise.
int emu_own_areas(uint32_t n, const sa_box* boxes, float* out_share, uint32_t max_nb, uint32_t cap) {
  std::vector<double> verts((size_t)n * 8);
  std::vector<sa_geo> geo(n);
  for (uint32_t i = 0; i < n; ++i) {
    const sa_box& b = boxes[i];
    double a = (double)(b.has_angle ? b.angle : 0.0f);
    double c = 1.0, s = 0.0;
    if (a != 0.0) ::sincos(a, &s, &c);
    sa_vertices(b.xc, b.yc, b.aspect, b.height, c, s, &verts[(size_t)i * 8]);
    geo[i].xc = b.xc; geo[i].yc = b.yc; geo[i].r = sa_radius(b.aspect, b.height); geo[i].hha = 0.f;
  }
  int status = 0;
  std::vector<double> polys, iva(cap), ivb(cap);
  for (uint32_t i = 0; i < n; ++i) {
    const double ox = (double)boxes[i].xc, oy = (double)boxes[i].yc;
    polys.clear();
    for (int k = 0; k < 8; ++k) polys.push_back(verts[(size_t)i * 8 + k] - ((k & 1) ? oy : ox));
    for (uint32_t j = 0; j < n; ++j) {
      if (j == i || sa_too_far(geo[i], geo[j]) || sa_quads_separated(&verts[(size_t)i * 8], &verts[(size_t)j * 8])) continue;
      for (int k = 0; k < 8; ++k) polys.push_back(verts[(size_t)j * 8 + k] - ((k & 1) ? oy : ox));
    }
    const uint32_t m1 = (uint32_t)(polys.size() / 8);
    if (m1 - 1 > max_nb) { status |= 1; out_share[i] = NAN; continue; }
    double acc = 0.0;
    for (uint32_t e = 0; e < m1 * 4; ++e) acc += sa_own_edge(polys.data(), m1, e >> 2, e & 3u, iva.data(), ivb.data(), 1, cap);
    if (acc != acc) status |= 2;
    const double own = fabs(acc) * 0.5;
    const float e = (float)(own / (double)(sa_area(boxes[i].aspect, boxes[i].height) + SA_EPS));
    out_share[i] = e >= 1.0f ? 1.0f : e;
  }
  return status;
}

}  // extern "C"
