// sa_frame.h — device code of the first launch of a frame, shared by sa_kernels.hip (k_frame: positional tiles + frame
// preparation blocks) and sa_gemm.hip (k_frame_visual: the same two block kinds riding beside the contraction's tiles).
#pragma once
#include "sa_engine.h"
#include <type_traits>

#ifndef WAVE
#define WAVE 64
#endif

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

// One wave per feature row: zero-pad D -> Dp (Feature::from_vec, track/utils.rs:45-71; the extra zero lanes
// add +0.0 to every sum), scatter, squared norm (the per-pair norms of distance.rs:36-44 hoisted to once
// per vector).
// (fr / frow: the destination's fragment-order twin and the row's index in it — the track bank, sa_frag_index — or nullptr)
__device__ __forceinline__ void pad_feature_row(const float* s, float* d, uint32_t D, uint32_t Dp,
                                                bool pres, uint32_t lane, float* norm_out, float* fr = nullptr, uint32_t frow = 0) {
  float acc = 0.0f;
  const bool alias = d == s;  // D == Dp: the uploaded rows ARE the padded rows (the engine points c_feat at them): norms only
  if (pres && (D & 3u) == 0 && ((uintptr_t)s & 15u) == 0) {
    for (uint32_t k = lane * 4; k < Dp; k += WAVE * 4) {
      float4 x = k < D ? *(const float4*)(s + k) : float4{0.f, 0.f, 0.f, 0.f};
      if (!alias) *(float4*)(d + k) = x;
      if (fr) *(float4*)(fr + sa_frag_index(frow, k, Dp)) = x;   // four consecutive k of a row are contiguous in fragment order too
      acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
  } else {
    for (uint32_t k = lane; k < Dp; k += WAVE) {
      float x = (pres && k < D) ? s[k] : 0.0f;
      if (!alias || !pres) d[k] = x;  // a row without a feature is zeroed in place (nothing reads its uploaded content)
      if (fr) fr[sa_frag_index(frow, k, Dp)] = x;
      acc += x * x;
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  *norm_out = acc;
}


// =====================================================================================================
// Frame preparation blocks: everything that is O(N + T) at the start of a frame and that the kernels AFTER the first launch
// consume —
//   * reset of the vote / assignment state (one thread per vertex of the bipartite graph),
//   * candidate preparation (visual_sort/simple_api.rs:130-170): geometry, f64 vertices, Mahalanobis measurement
//     (angle.unwrap_or(0), kalman_2d_box.rs:159), clamped confidence (sort/metric.rs:43-47), the feature_can_be_used gate
//     (visual_sort/metric.rs:227-249) — for the contraction's epilogue and the parity taps,
//   * candidate feature padding + squared norms (one wave per row).
// They ride in the SAME launch as the positional tiles (k_frame below), which re-derive the few candidates they need from
// the raw boxes instead of waiting for these arrays: one dependent launch (~4.7 us) less per frame.  State the positional
// tiles themselves write (edge counters, and for the many-workgroup tail the row duals and the union-find forest) cannot be
// reset beside them; the assignment tail leaves it clean for the next frame instead (k_slot_init establishes it once).
// =====================================================================================================
// (tid = thread index inside the 256-thread unit: a 512-thread block of the fused launch runs two units side by side)
// light: the RESET half only (a lean frame on the many-workgroup tail: nothing reads the candidates' derived arrays)
__device__ __forceinline__ void frame_prep_block(const SceneDev& S, const SaParams& p, uint32_t blk, uint32_t tid, bool light = false) {
  const uint32_t N = S.N, T = S.T;
  const uint32_t i = blk * 256 + tid;
  if (i < T) {
    S.col_excluded[i] = 0;
    S.v[i] = 0;
    S.cmatch[i] = -1;
    S.cstamp[i] = 0;
    S.cscan[i] = 0;
    S.cwin[i] = SA_NONE;   // greedy bids of the big-component solver (k_assign_solve)
  }
  if (i < N) {
    S.vis_winner[i] = -1;
    S.row_has[i] = 0;
    S.rmatch[i] = -1;
    S.label[i] = SA_NONE;
    S.next_row[i] = SA_NONE;
    S.rnext[i] = 0;        // rows per component, counted by k_assign_label
  }
  if (light) return;
  if (i < N) {
    BoxRaw r = sa_ldg(S.c_raw + i);
    prep_box_common(r, (sa_geo*)(S.c_geo + i), (double*)(S.c_verts + (size_t)i * 8));
    const sa_box& b = r.box;
    float SA_G* z = S.c_z + (size_t)i * 5;
    z[0] = b.xc; z[1] = b.yc; z[2] = b.has_angle ? b.angle : 0.0f; z[3] = b.aspect; z[4] = b.height;
    S.c_conf[i] = b.confidence < p.min_confidence ? p.min_confidence : b.confidence;
    bool usable = false;
    if ((S.flags & SCN_HAS_FEATS) && (!(S.flags & SCN_HAS_FPRESENT) || S.c_fpresent_in[i])) {
      float q = (S.flags & SCN_HAS_QUALITY) ? S.c_quality[i] : 1.0f;
      bool quality_ok = q >= p.visual_minimal_quality_use;
      bool perc_ok = true;
      if (S.flags & SCN_HAS_OWN) {
        float oa = S.c_own[i];
        if (oa == oa) perc_ok = oa >= p.visual_minimal_own_area_use;
      }
      bool bbox_ok = sa_area(b.aspect, b.height) >= p.visual_minimal_area;
      usable = bbox_ok && quality_ok && perc_ok;
    }
    S.c_usable[i] = usable ? 1 : 0;
  }
  if (S.flags & SCN_HAS_FEATS) {
    const uint32_t row = blk * 4 + tid / WAVE, lane = tid % WAVE;
    if (row < N) {
      bool pres = !(S.flags & SCN_HAS_FPRESENT) || S.c_fpresent_in[row] != 0;
      float nrm;
      pad_feature_row(S.c_feat_raw + (size_t)row * S.D, S.c_feat + (size_t)row * S.Dp, S.D, S.Dp, pres, lane, &nrm);
      if (lane == 0) S.c_fnorm[row] = nrm;
    }
  }
}


// =====================================================================================================
// Positional cost cells: pair pre-filter -> (survivors only) IoU by f64 Sutherland–Hodgman / Mahalanobis -> the sparse
// input of the positional vote.
// Tile = 16 candidates x 256 tracks per 256-thread block: lane = track, each wave owns 4 candidate rows, a thread tests
// 16 cells.  Phase 1 tests every cell: too_far() first (no sqrt), then compatible() — whose dist_in_2r costs a sqrt and a
// division — only for the cells that pass; the few survivors are compacted into an LDS list so that phase 2 runs the
// expensive clip with full lanes instead of 1-2 live lanes per wave.
// EDGES (the product path): a surviving cell whose quantised weight beats the new-track threshold becomes an edge of the
//   assignment graph right here — appended to its row's list (order inside a row is irrelevant to the solver), folded
//   into the row dual and the union-find forest.  The dense N x T matrix of SortVoting (sort/voting.rs:44-84) and even the
//   dense f32 cost matrix are never written: what this kernel moves is 80 B per box in and ~20 B per edge out.
// DENSE (parity taps only): the f32 cost matrix, NaN = absent.
// =====================================================================================================
// NSUB = 64-track sub-tiles per block: 4 (16 x 256 cells, 128 clip lanes) when the frame still yields several blocks per
// CU that way — fewer, longer-lived blocks amortise the fixed cost of a block (C4: 62 500 blocks of 16 x 64 took 25 us, 1000
// blocks of 16 x 256 take 15) — else 1 (16 x 64 cells, 64 clip lanes), which keeps every CU busy on small or dense frames
// where the clip rounds, not the pre-filter, set the time (C2: ~60 surviving pairs per candidate).
// LDS written by some lanes of a wave and read by others of the SAME wave: program order is enough for the hardware (one
// wave's DS operations complete in order); the fences keep the compiler from reordering across the hand-over.
#define SA_WAVE_LDS_SYNC()                                   \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

// sa_clip_area_ws (sa_device.h) spread over L lanes per pair: the same Sutherland–Hodgman passes and the same shoelace sum, value
// for value and in the same order, but L vertices of a pass are tested (and their crossings computed) side by side, one per lane.
// A lone wave issues a dependent instruction every 8-9 cycles, so the serial clip's time is its step count (ten two-vertex steps
// of ~100 instructions for an axis-aligned pair); with four lanes per pair the 64 pairs of a crowded tile occupy all four waves
// and a pass is one or two steps.  Output positions come from two ballots (which lanes emit a crossing, which keep their
// vertex): vertex j writes [crossing,] [itself] after everything lanes < j emit — the sequential order.  subj is indexed by lane
// (keep it in LDS, not in registers); clip is indexed statically.  ws = this group's LDS: two ping-pong lists of SA_POLY_CAP
// (x, y) pairs.  l = lane within the group, gshift = position of the group's lane 0 within the wave.  Every lane returns the area.
template <int L>
__device__ __forceinline__ double clip_area_lanes(const double* subj, const double* clip, double* ws, uint32_t l, uint32_t gshift) {
  double* px = ws;
  double* py = ws + SA_POLY_CAP;
  double* qx = ws + 2 * SA_POLY_CAP;
  double* qy = ws + 3 * SA_POLY_CAP;
  for (uint32_t v = l; v < 4; v += L) { px[v] = subj[2 * v]; py[v] = subj[2 * v + 1]; }
  SA_WAVE_LDS_SYNC();
  uint32_t n = 4;
  const uint32_t below = (1u << l) - 1u, gmask = (1u << L) - 1u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ii = i == 0 ? 3 : i - 1;
    const double csx = clip[2 * ii], csy = clip[2 * ii + 1];
    const double cex = clip[2 * i], cey = clip[2 * i + 1];
    const double dpx = csx - cex, dpy = csy - cey;
    const double n2 = csx * cey - csy * cex;
    uint32_t m = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += L) {
      const uint32_t j = c0 + l;
      const bool act = j < n;
      const uint32_t jp = j == 0 ? n - 1 : j - 1;
      double sex = 0.0, sey = 0.0, ssx = 0.0, ssy = 0.0;
      if (act) { sex = px[j]; sey = py[j]; ssx = px[jp]; ssy = py[jp]; }
      const bool in_e = act && ((cex - csx) * (sey - csy) - (cey - csy) * (sex - csx)) <= 0.0;
      const bool in_s = act && ((cex - csx) * (ssy - csy) - (cey - csy) * (ssx - csx)) <= 0.0;
      const bool cross = in_e != in_s;
      const uint32_t gc = (uint32_t)(__ballot(cross) >> gshift) & gmask;
      const uint32_t ge = (uint32_t)(__ballot(in_e) >> gshift) & gmask;
      uint32_t off = m + __popc(gc & below) + __popc(ge & below);
      // compute_intersection(cp1 = s_edge_start, cp2 = s_edge_end, s = c_edge_start, e = c_edge_end)  clipping.rs:17-38
      const double dcx = ssx - sex, dcy = ssy - sey;
      const double n1 = ssx * sey - ssy * sex;
      const double n3 = 1.0 / (dcx * dpy - dcy * dpx);
      const double ix = (n1 * dpx - n2 * dcx) * n3, iy = (n1 * dpy - n2 * dcy) * n3;
      if (cross) {
        if (off < SA_POLY_CAP) { qx[off] = ix; qy[off] = iy; }
        ++off;
      }
      if (in_e && off < SA_POLY_CAP) { qx[off] = sex; qy[off] = sey; }
      m += __popc(gc) + __popc(ge);
    }
    n = m < SA_POLY_CAP ? m : SA_POLY_CAP;
    double* t = px; px = qx; qx = t;
    t = py; py = qy; qy = t;
    SA_WAVE_LDS_SYNC();
  }
  if (n == 0) return 0.0;
  // Polygon::new closes the ring unless first == last; < 3 coordinates -> 0; shoelace with the first vertex as the shift
  const double shx = px[0], shy = py[0];
  const bool closed = shx == px[n - 1] && shy == py[n - 1];
  const uint32_t mm = closed ? n : n + 1;
  if (mm < 3) return 0.0;
  // the terms side by side (qx is free now), the sum in the reference's order by every lane
  for (uint32_t c0 = 0; c0 + 1 < mm; c0 += L) {
    const uint32_t i = c0 + l;
    if (i + 1 < mm) {
      const uint32_t i1 = (i + 1 == n) ? 0 : i + 1;
      const double x0 = px[i] - shx, y0 = py[i] - shy;
      const double x1 = px[i1] - shx, y1 = py[i1] - shy;
      qx[i] = x0 * y1 - y0 * x1;
    }
  }
  SA_WAVE_LDS_SYNC();
  // all SA_POLY_CAP slots are fetched at once (one LDS latency instead of one per term); a slot past the last term adds +0.0,
  // which leaves every partial sum as it was except -0.0 -> +0.0, and the fabs below does not see that
  double term[SA_POLY_CAP];
#pragma unroll
  for (int i = 0; i < SA_POLY_CAP; ++i) term[i] = qx[i];
  double tmp = 0.0;
#pragma unroll
  for (int i = 0; i < SA_POLY_CAP; ++i) tmp = tmp + ((uint32_t)i + 1 < mm ? term[i] : 0.0);
  SA_WAVE_LDS_SYNC();  // the lists are reused by the group's next clip
  const double area = tmp / (1.0 + 1.0);
  return fabs(area);
}

#define POS_TI 16
// UNION: also fold each edge into the row dual and the global union-find forest (needed by the many-workgroup assignment
// tail).  When the whole scene is solved by ONE workgroup (k_assign_small) that workgroup builds both from the edge lists in
// LDS instead — the chain of dependent global atomics per edge would otherwise sit at the tail of every block here.
template <int NSUB, int WORKERS = 64>
struct PosSmem {
  double poly[4 * SA_POLY_CAP * WORKERS];   // 24 KB at 64 workers: Sutherland–Hodgman ping-pong lists, [list][vertex][worker lane]
  double cv[POS_TI][8];                // candidate polygons, derived here from the raw boxes (see frame_prep_block)
  sa_geo cg[POS_TI];
  sa_ext ce[POS_TI];
  float cconf[POS_TI], cz[POS_TI][5];
  float thha[64 * NSUB];
  uint32_t cnt, cnt2;
  uint16_t list[POS_TI * 64 * NSUB];   // (li << 8) | lj
};
// PROOF: drop the pairs whose clip is provably empty (sa_clip_is_empty) before the clipper runs; WORKERS: lanes of the clipping
// wave (their vertex lists are the bulk of the tile's LDS).  As a kernel of its own a tile is a chain of latencies and the extra
// phase only lengthens it (measured: C2 k_frame 9.5 -> 10.0 us).  In the heterogeneous launch the tiles have slack, but every f64
// wave instruction they issue is time the matrix-core waves on the same SIMD do not get (3.7 us of a 22 us launch at C2, where
// ~80 % of the bounding-circle neighbours do not overlap) — and a clip round costs the same instructions for 12 live lanes as
// for 60, so proofs alone make it worse (24.6 us).  There the tiles are 16 x 128 (~120 surviving pairs): the threads of two
// waves prove what they can and the ~25 pairs that are left take ONE clip round where two 16 x 64 tiles took two: 21.2 us.
// (16 x 192 and 16 x 256 tiles: 25 us — the tile itself then outlasts the contraction.)
// In-kernel timeline of a tile (build with -DSA_POS_TRACE, run with SA_POS_TRACE=<launch #>: scripts/pos_trace.sh): s_memtime of thread 0 at
//   0 entry | 1 candidate boxes in LDS, this thread's track loads landed | 2 every cell screened (too_far + compatible), survivors listed |
//   3 disjointness proofs done (= 2 when the stage is skipped) | 4 clip rounds done, edges appended | [6] survivors, [7] pairs clipped |
//   [5] = (s_memrealtime at entry << 32) | (s_memrealtime at exit & 0xffffffff): the 100 MHz clock every XCD shares (s_memtime's base differs per XCD)
#ifdef SA_POS_TRACE
#define POS_STAMP(tr, i) do { if ((tr) && tid == 0) (tr)[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define POS_NOTE(tr, i, v) do { if ((tr) && tid == 0) (tr)[i] = (uint64_t)(v); } while (0)
#define POS_RT() __builtin_amdgcn_s_memrealtime()
#else
#define POS_RT() 0ull
#define POS_STAMP(tr, i) do { } while (0)
#define POS_NOTE(tr, i, v) do { } while (0)
#endif
template <bool DENSE, bool EDGES, int NSUB, bool UNION, bool PROOF = false, int WORKERS = 64, bool COOP = false>
__device__ __forceinline__ void positional_tile(const SceneDev& S, const SaParams& p, uint32_t bx, uint32_t by, void* smem, uint32_t tid,
                                                uint64_t* tr = nullptr) {
  constexpr uint32_t POS_TJ = 64u * NSUB, POS_WORKERS = (uint32_t)WORKERS;
  const uint32_t N = S.N, T = S.T;
  const uint32_t i0 = by * POS_TI, j0 = bx * POS_TJ;
  if (i0 >= N || j0 >= T) return;
#if defined(SA_POS_SKIP) && SA_POS_SKIP >= 4   // (measurement only, scripts/pos_skip_probe.sh: what each phase of a tile costs its launch; the answers are wrong)
  return;
#endif
  POS_STAMP(tr, 0);
  [[maybe_unused]] const uint64_t pos_rt0 = POS_RT();
  // LDS comes from the caller (one raw buffer per kernel): in the fused VisualSORT launch the tiles share their kernel's
  // static LDS with the contraction's stages instead of adding to it
  PosSmem<NSUB, WORKERS>& sm = *reinterpret_cast<PosSmem<NSUB, WORKERS>*>(smem);
  auto& s_cg = sm.cg; auto& s_cv = sm.cv; auto& s_cconf = sm.cconf; auto& s_cz = sm.cz; auto& s_list = sm.list;
  uint32_t& s_cnt = sm.cnt; auto& s_poly = sm.poly; auto& s_thha = sm.thha;
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  if (tid < POS_TI) {
    const uint32_t i = i0 + tid;
    if (i < N) {
      const BoxRaw r = sa_ldg(S.c_raw + i);
      prep_box_common(r, &s_cg[tid], s_cv[tid]);
      const sa_box& b = r.box;
      sm.ce[tid] = sa_box_ext(b.aspect, b.height, b.has_angle && b.angle != 0.0f);
      s_cconf[tid] = b.confidence < p.min_confidence ? p.min_confidence : b.confidence;
      s_cz[tid][0] = b.xc; s_cz[tid][1] = b.yc; s_cz[tid][2] = b.has_angle ? b.angle : 0.0f; s_cz[tid][3] = b.aspect; s_cz[tid][4] = b.height;
    } else {
      s_cg[tid] = sa_geo{0.f, 0.f, 0.f, 0.f};
      sm.ce[tid] = sa_ext{-1.f, 0.f};
    }
  }
  if (tid == 0) { s_cnt = 0; sm.cnt2 = 0; }
  // this thread's 4 tracks (one per 64-wide sub-tile): loads in flight while the candidate tile lands in LDS
  sa_geo tg[NSUB];
  uint64_t te[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
    const uint32_t j = j0 + s * 64 + lane;
    tg[s] = j < T ? sa_ldg(S.t_geo + j) : sa_geo{0.f, 0.f, 0.f, 0.f};
    te[s] = j < T ? S.t_epoch[j] : 0ull;
    if (wave == 0) s_thha[s * 64 + lane] = tg[s].hha;
  }
  __syncthreads();
  POS_STAMP(tr, 1);
  const float nanv = __builtin_nanf("");
  const uint64_t epoch = S.epoch;
  // The screen: 16 cells per thread (4 candidate rows of this wave x NSUB tracks of this lane).  It is VALU THROUGHPUT, not latency:
  // every co-resident tile of a CU is in this phase at the same time (in-kernel timeline, scripts/pos_trace.sh: 4.0 k cycles of a C4
  // tile's 18 k — 4 M cells x ~15 instructions over the chip's 1024 SIMDs — whatever the survivors' path costs: appending them per wave
  // with one ballot and a scalar branch per step instead of this divergent branch was measured at 4.8 k).  So the cells are made
  // cheaper: too_far() (bbox.rs:452-462) of TWO candidate rows against the lane's track in packed f32 arithmetic (v_pk_add / v_pk_mul:
  // two cells per instruction; every element is still subtract, multiply, multiply, add, compare in the reference's order — no fused
  // multiply-add — so each cell's verdict is bit for bit the scalar one), and compatible()'s epoch test once per TRACK.
  typedef float pf2 __attribute__((ext_vector_type(2)));
  pf2 cgx[2], cgy[2], cgr[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const sa_geo g0 = s_cg[wave * 4 + 2 * h], g1 = s_cg[wave * 4 + 2 * h + 1];
    cgx[h] = pf2{g0.xc, g1.xc}; cgy[h] = pf2{g0.yc, g1.yc}; cgr[h] = pf2{g0.r, g1.r};
  }
#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
    const uint32_t lj = s * 64 + lane, j = j0 + lj;
    const pf2 tx = pf2{tg[s].xc, tg[s].xc}, ty = pf2{tg[s].yc, tg[s].yc}, tr = pf2{tg[s].r, tg[s].r};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // sa_too_far: max_distance = l.r + r.r; x = l.xc - r.xc, y = l.yc - r.yc; x * x + y * y > max_distance * max_distance
      const pf2 md = cgr[h] + tr, x = cgx[h] - tx, y = cgy[h] - ty;
      const pf2 d2 = x * x + y * y, m2 = md * md;
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int r = 2 * h + q2;
        const uint32_t li = wave * 4 + r, i = i0 + li;
        bool live = false;
        if (i < N && j < T) {
          live = !(d2[q2] > m2[q2]) && sa_compatible(s_cg[li], epoch, tg[s], te[s], p.max_idle, p.cons);
          if (DENSE && !live) S.pos[(size_t)i * T + j] = nanv;
        }
        if (live) {
          uint32_t slot = atomicAdd(&s_cnt, 1u);
          s_list[slot] = (uint16_t)((li << 8) | lj);
        }
      }
    }
  }
  __syncthreads();
  uint32_t cnt = s_cnt;
#if defined(SA_POS_SKIP) && SA_POS_SKIP >= 3
  cnt = 0;
#endif
  POS_STAMP(tr, 2);
  POS_NOTE(tr, 6, cnt);
  // (only where it saves a clip round: up to 64 surviving pairs are one round of the four-lane clipper whatever their number)
  if (PROOF && p.positional_kind != SA_POS_MAHALANOBIS && cnt > 64u) {
    // one thread per surviving pair proves, where it can, that the polygons do not overlap; the rest are compacted in place
    // (every thread holds its entries in registers while the list is rewritten)
    uint32_t keep = 0;
    uint16_t mine[4 * NSUB];
#pragma unroll
    for (int k = 0; k < 4 * NSUB; ++k) {
      const uint32_t sidx = tid + 256u * k;
      mine[k] = 0;
      if (sidx < cnt) {
        const uint32_t c = s_list[sidx];
        const uint32_t li = c >> 8, lj = c & 255u;
        mine[k] = (uint16_t)c;
        // two boxes without an angle: rectangles that miss each other, or overlap too little to reach the threshold, decided in f32
        // from centres and half extents (sa_aa_quick_reject: conservative; 8 bytes of the track instead of its 64-byte polygon and
        // ~25 f32 operations instead of ~200 f64 ones); otherwise the separating-edge proof on the polygons
        const sa_ext cx = sm.ce[li];
        const sa_ext tx = sa_ldg(S.t_ext + j0 + lj);
        bool drop;
        if (cx.hw >= 0.0f && tx.hw >= 0.0f) drop = sa_aa_quick_reject(s_cg[li], cx, sa_ldg(S.t_geo + j0 + lj), tx, s_cconf[li], p.positional_threshold);
        else {
          double cv[8], tv[8];
          const double SA_G* tp = S.t_verts + (size_t)(j0 + lj) * 8;
#pragma unroll
          for (int q = 0; q < 8; ++q) { cv[q] = s_cv[li][q]; tv[q] = tp[q]; }
          drop = sa_clip_is_empty(cv, tv);
        }
        if (!drop) keep |= 1u << k;
        else if (DENSE) S.pos[(size_t)(i0 + li) * T + j0 + lj] = nanv;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4 * NSUB; ++k)
      if ((keep >> k) & 1u) s_list[atomicAdd(&sm.cnt2, 1u)] = mine[k];
    __syncthreads();
    cnt = sm.cnt2;
  }
#if defined(SA_POS_SKIP) && SA_POS_SKIP >= 2
  cnt = 0;
#endif
  POS_STAMP(tr, 3);
  POS_NOTE(tr, 7, cnt);
  // one surviving cell -> (optionally) the dense matrix, and its edge
  auto emit = [&](uint32_t i, uint32_t j, float w, bool present) {
    if (DENSE) S.pos[(size_t)i * T + j] = present ? w : nanv;
    if (EDGES && present) {
      const int64_t gain = sa_quantise(w) - p.threshold_q;  // (w * 1e6f) as i64 vs the diagonal of SortVoting's matrix
      if (gain > 0) {
        // every request of the edge that does not depend on another one goes out together: the slot in the row's list, and (UNION)
        // the two union-find parents — one round trip to L2 instead of three in a row at the tail of the tile (pos_trace: the edge
        // append was ~6 k of a C4 tile's 18 k cycles)
        const uint32_t slot = atomicAdd((uint32_t*)(S.e_cnt + i), 1u);
        uint32_t pa = 0, pb = 0;
        if (UNION) {
          pa = sa_ld_u32((const uint32_t*)S.parent + i);
          pb = sa_ld_u32((const uint32_t*)S.parent + N + j);
          __hip_atomic_fetch_min((int64_t*)(S.u + i), -gain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // row dual = -max gain
        }
        // general tail (UNION): row-major lists, a short row is one cache line for the thread that gathers its component;
        // one-workgroup tail: SLOT-major — its 1024 threads fetch "edge k of my row" side by side, 64 lanes = one contiguous
        // kilobyte (row-major: 64 lines per wave-load through ONE compute unit's address path, 4 k cycles of the tail)
        sa_stg(S.e_edge + (UNION ? (size_t)i * S.estride + slot : (size_t)slot * N + i), SaEdge{gain, j, 0u});
        if (UNION) sa_uf_union_from((uint32_t*)S.parent, i, pa, N + j, pb);
      }
    }
  };
  if (p.positional_kind == SA_POS_MAHALANOBIS) {
    for (uint32_t sidx = tid; sidx < cnt; sidx += 256) {
      const uint32_t c = s_list[sidx];
      const uint32_t li = c >> 8, lj = c & 255u;
      const uint32_t i = i0 + li, j = j0 + lj;
      float m20[20], z5[5];
      const float SA_G* mp = S.t_maha + (size_t)j * 20;
#pragma unroll
      for (int k = 0; k < 20; ++k) m20[k] = mp[k];
#pragma unroll
      for (int k = 0; k < 5; ++k) z5[k] = s_cz[li][k];
      emit(i, j, sa_maha_cell(m20, z5, s_cconf[li]), true);
    }
  } else if (COOP) {
    // L lanes per pair (clip_area_lanes): four — 64 pairs at a time over the four waves, the vertex lists are the same 24 KB — or, when
    // the tile's survivors fit 32 groups (cnt is uniform over the block), EIGHT: a pass over a list of up to eight vertices is then ONE step
    // of the dependent chain instead of two (a quad clipped by a quad: 4 -> at most 8 vertices), and a tile's clip phase is that chain
    auto clip_pairs = [&](auto LTAG) {
      // (the groups' vertex lists are the bulk of the tile's LDS — POS_WORKERS of them: with fewer than 256 / L the surplus groups idle)
      constexpr uint32_t L = decltype(LTAG)::value, GROUPS = (256u / L) < POS_WORKERS ? (256u / L) : POS_WORKERS;
      const uint32_t grp = tid / L, gl = tid & (L - 1u), gshift = lane & (64u - L);
      double* ws = s_poly + (grp < GROUPS ? grp : 0u) * (4 * SA_POLY_CAP);
      for (uint32_t sidx = grp < GROUPS ? grp : cnt; sidx < cnt; sidx += GROUPS) {
        const uint32_t c = s_list[sidx];
        const uint32_t li = c >> 8, lj = c & 255u;
        const uint32_t i = i0 + li, j = j0 + lj;
        double tv[8];
        const double SA_G* tp = S.t_verts + (size_t)j * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) tv[k] = tp[k];
        const double inter = clip_area_lanes<(int)L>(s_cv[li], tv, ws, gl, gshift);
        if (gl == 0) {
          float iou, out = nanv;
          bool present = false;
          if (sa_iou_from_area(inter, s_cg[li].hha, s_thha[lj], &iou)) {
            const float e = iou * s_cconf[li];
            if (e >= p.positional_threshold) { out = e; present = true; }
          }
          emit(i, j, out, present);
        }
      }
    };
    // (four lanes for every count — half the waves of a 25-pair tile in the clipper — was measured in the fused launch, where vector
    // issue is what the tiles take from the contraction: C2 first phase 15.2-15.3 us against 15.0, also with the active waves rotated from
    // tile to tile: what the launch waits for is the tiles' chain, not their instruction count)
    if (cnt <= 32u) clip_pairs(std::integral_constant<uint32_t, 8u>{});
    else clip_pairs(std::integral_constant<uint32_t, 4u>{});
  } else if (tid < POS_WORKERS) {
    // Sutherland–Hodgman vertex lists: 4 lists x 12 vertices per worker lane, [list][vertex][lane] in LDS
    double* ws = s_poly + tid;
    for (uint32_t sidx = tid; sidx < cnt; sidx += POS_WORKERS) {
      const uint32_t c = s_list[sidx];
      const uint32_t li = c >> 8, lj = c & 255u;
      const uint32_t i = i0 + li, j = j0 + lj;
      double cv[8], tv[8];
      const double SA_G* tp = S.t_verts + (size_t)j * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) { cv[k] = s_cv[li][k]; tv[k] = tp[k]; }
      const float t_hha = s_thha[lj];
      double inter = sa_clip_area_ws(cv, tv, ws, ws + SA_POLY_CAP * POS_WORKERS, ws + 2 * SA_POLY_CAP * POS_WORKERS,
                                     ws + 3 * SA_POLY_CAP * POS_WORKERS, POS_WORKERS);
      float iou, out = nanv;
      bool present = false;
      if (sa_iou_from_area(inter, s_cg[li].hha, t_hha, &iou)) {
        float e = iou * s_cconf[li];
        if (e >= p.positional_threshold) { out = e; present = true; }
      }
      emit(i, j, out, present);
    }
  }
  POS_STAMP(tr, 4);
  POS_NOTE(tr, 5, (pos_rt0 << 32) | (POS_RT() & 0xffffffffull));
}

