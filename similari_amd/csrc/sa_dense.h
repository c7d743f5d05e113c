// sa_dense.h — exact assignment of one LARGE connected component by a whole workgroup (gfx950), shared with tests/emu.
//
// The cooperative solver of sa_device.h (sa_assign_component_coop) gives a component to ONE wavefront and walks sparse edge
// lists: right for the handful-of-rows components of a tracking frame, but a crowd under a low IoU threshold is one component of
// hundreds of rows and hundreds of thousands of edges, and every step of its shortest-path searches then costs a strided scan of
// the labelled columns plus a chase through the row's edge list in HBM (round 2: 11.8 k cycles per step, 29 ms for 640 rows —
// slower than the reference's kuhn_munkres on one host core, sort/voting.rs:86).
//
// Here the same Jonker–Volgenant search (same duals, same (distance, column) tie-breaks, same order of the roots: the answer is
// the serial solver's, bit for bit) runs on a DENSE row store with one thread per column strip:
//   * the component's gains lie in a dense matrix in HBM, gain[row][col] (0 = no edge): "relax the row that just entered the
//     tree" is ONE coalesced row read — every thread fetches the cells of its own CPT columns — instead of a list walk;
//   * a column's search state (distance, labelled / scanned flags, its dual v) lives in the REGISTERS of the thread that owns it;
//   * "nearest labelled, unscanned column" is a workgroup minimum of packed (distance << 16 | column) keys: 32-bit DPP
//     reductions inside a wave, one LDS slot per wave, one barrier per search step;
//   * per-row state (dual u, match) and the per-column match / predecessor live in LDS (k_assign_small) or in HBM behind the
//     workgroup's own L1 (k_assign_dense on frames beyond the LDS budget).
// One search step = one barrier + one L2 round trip (~1.5 k cycles) whatever the density; 640 rows / 140 k edges: ~6 k steps.
//
// Written once for both worlds like the cooperative solver: on the device a "thread loop" runs its body once, for this thread;
// in the host emulation it runs NT times, per-thread values live in arrays of SA_WG_SLOTS(NT) elements.
#pragma once
#include "sa_device.h"

#if defined(__HIPCC__)
#define SA_WG_FN __device__ __forceinline__
#define SA_WG_SLOTS(NT) 1
#define SA_WG_SLOT(t) 0
#define SA_WG_FOR(NT, t) for (uint32_t t [[maybe_unused]] = threadIdx.x, _sa_wg_once = 1; _sa_wg_once; _sa_wg_once = 0)
#else
#define SA_WG_FN inline
#define SA_WG_SLOTS(NT) (NT)
#define SA_WG_SLOT(t) (t)
#define SA_WG_FOR(NT, t) for (uint32_t t = 0; t < (uint32_t)(NT); ++t)
#endif

// A search key: (distance, column) packed so that the unsigned 64-bit minimum is the lexicographic minimum the serial solver takes.
// Distances of columns that can still be scanned are < best_term <= the root's heaviest gain < 2^47 (quantised weights: 1e6 x a
// weight; a Mahalanobis cost / confidence stays below 1.4e14 down to confidences of 1e-6), larger ones saturate — a saturated key
// can only win when every candidate is beyond best_term, and then the search stops anyway.  Columns: T <= 65535.
#define SA_DENSE_KEY_NONE (~0ull)
#define SA_DENSE_DIST_SAT ((1ll << 47) - 1)
SA_HD unsigned long long sa_dense_key(int64_t d, uint32_t j) {
  const int64_t c = d > SA_DENSE_DIST_SAT ? SA_DENSE_DIST_SAT : d;
  return ((unsigned long long)c << 16) | (unsigned long long)(j & 0xffffu);
}

#if defined(__HIPCC__)
// minimum of a 32-bit value over the 16 lanes of a DPP row, result in every lane of the row (xor-butterfly by quad permutes and
// row mirrors: 4 v_min_u32 with a DPP operand)
__device__ __forceinline__ uint32_t sa_row_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));  // row_half_mirror
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));  // row_mirror
  return v;
}
__device__ __forceinline__ uint32_t sa_wave_min_u32(uint32_t v) {
  v = sa_row_min_u32(v);
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}
// 64-bit minimum over the wave as two 32-bit ones: the smallest high word, then the smallest low word among its holders
__device__ __forceinline__ unsigned long long sa_wave_min_u64(unsigned long long k) {
  const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
  const uint32_t mh = sa_wave_min_u32(hi);
  const uint32_t ml = sa_wave_min_u32(hi == mh ? lo : 0xffffffffu);
  return ((unsigned long long)mh << 32) | ml;
}
// Workgroup minimum, every thread receives it.  part: [2][NT / 64] slots in LDS, alternated by `parity` so that ONE barrier per
// call is enough (a wave that races ahead into the next call writes the other half).
template <int NT>
__device__ __forceinline__ unsigned long long sa_wg_min_u64(const unsigned long long* key, unsigned long long* part, uint32_t parity) {
  constexpr int W = NT / 64;
  const unsigned long long k = sa_wave_min_u64(key[0]);
  if constexpr (W == 1) return k;
  if ((threadIdx.x & 63u) == 0) part[parity * W + (threadIdx.x >> 6)] = k;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  unsigned long long m = part[parity * W];
#pragma unroll
  for (int w = 1; w < W; ++w) {
    const unsigned long long o = part[parity * W + w];
    m = o < m ? o : m;
  }
  return m;
}
template <int NT>
__device__ __forceinline__ void sa_wg_sync() { __syncthreads(); }
#else
template <int NT>
inline unsigned long long sa_wg_min_u64(const unsigned long long* key, unsigned long long*, uint32_t) {
  unsigned long long m = SA_DENSE_KEY_NONE;
  for (int t = 0; t < NT; ++t) m = key[t] < m ? key[t] : m;
  return m;
}
template <int NT>
inline void sa_wg_sync() {}
#endif

struct sa_dense_ws {
  const int64_t* gain;     // dense gains, row r at gain[r * ld + j]; 0 = no usable edge (excluded columns are never written)
  size_t ld;
  uint32_t T;              // columns of the scene
  int64_t* u;              // [N] row duals: -(heaviest usable gain) on entry
  int32_t* rmatch;         // [N] -1, or the column the greedy start gave the row
  int32_t* cmatch;         // [T] -1, or the row the greedy start gave the column
  int32_t* pred;           // [T] scratch: the tree row that labelled the column
  unsigned long long* part;  // device: [2][NT / 64] reduction slots in LDS
};

// Solves one component: `roots` = its rows the greedy start left unmatched (ascending), n_roots of them.  NT threads, thread t owns
// the columns t, t + NT, ... (CPT of them: T <= NT * CPT).  Every thread must call it; control flow is uniform.
template <int NT, int CPT>
SA_WG_FN void sa_assign_component_dense(const sa_dense_ws& w, const uint32_t* roots, uint32_t n_roots) {
  int64_t v[SA_WG_SLOTS(NT)][CPT];      // column duals of this thread's columns (0 on entry: a component's columns are untouched)
  int64_t dist[SA_WG_SLOTS(NT)][CPT];
  uint32_t lab[SA_WG_SLOTS(NT)], scn[SA_WG_SLOTS(NT)];  // bit c: column c of this thread is labelled / scanned in the running search
  SA_WG_FOR(NT, t) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int c = 0; c < CPT; ++c) { v[SA_WG_SLOT(t)][c] = 0; dist[SA_WG_SLOT(t)][c] = 0; }
  }
  uint32_t parity = 0;
  for (uint32_t ri = 0; ri < n_roots; ++ri) {
    const uint32_t root = roots[ri];
    int64_t best_term = -w.u[root];  // reduced cost of the root's own self column
    int32_t term_row = (int32_t)root;
    int32_t end_col = -1;
    int64_t delta = best_term;
    SA_WG_FOR(NT, t) { lab[SA_WG_SLOT(t)] = 0; scn[SA_WG_SLOT(t)] = 0; }
    uint32_t row = root;
    int64_t base = 0;
    // (every pass scans one more column of the component: the cap can only bite on corrupted state — a wrong answer the tests catch
    // instead of a kernel spinning on a GPU box)
    for (uint32_t guard = 0; guard < 65536u; ++guard) {
      // relax `row` (entered the tree at distance `base`) and form this thread's best key
      const int64_t ur = w.u[row];
      unsigned long long key[SA_WG_SLOTS(NT)];
      SA_WG_FOR(NT, t) {
        int64_t g[CPT];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < CPT; ++c) {
          const uint32_t j = t + (uint32_t)c * NT;
          g[c] = j < w.T ? w.gain[(size_t)row * w.ld + j] : 0;
        }
        unsigned long long k = SA_DENSE_KEY_NONE;
        uint32_t lb = lab[SA_WG_SLOT(t)];
        const uint32_t sc = scn[SA_WG_SLOT(t)];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < CPT; ++c) {
          const uint32_t j = t + (uint32_t)c * NT;
          const uint32_t bit = 1u << c;
          if (g[c] > 0 && !(sc & bit)) {
            const int64_t d = base + (-g[c] - ur - v[SA_WG_SLOT(t)][c]);
            if (!(lb & bit) || d < dist[SA_WG_SLOT(t)][c]) {
              dist[SA_WG_SLOT(t)][c] = d;
              w.pred[j] = (int32_t)row;
              lb |= bit;
            }
          }
          if ((lb & bit) && !(sc & bit)) {
            const unsigned long long kc = sa_dense_key(dist[SA_WG_SLOT(t)][c], j);
            k = kc < k ? kc : k;
          }
        }
        lab[SA_WG_SLOT(t)] = lb;
        key[SA_WG_SLOT(t)] = k;
      }
      const unsigned long long m = sa_wg_min_u64<NT>(key, w.part, parity);
      parity ^= 1u;
      const int64_t bd = (int64_t)(m >> 16);
      if (m == SA_DENSE_KEY_NONE || bd >= best_term) { delta = best_term; break; }  // a self column ends the path
      const uint32_t bj = (uint32_t)(m & 0xffffu);
      SA_WG_FOR(NT, t) { if (t == bj % NT) scn[SA_WG_SLOT(t)] |= 1u << (bj / NT); }
      const int32_t i = w.cmatch[bj];
      if (i < 0) { end_col = (int32_t)bj; delta = bd; break; }               // free real column
      const int64_t tt = bd + (-w.u[i]);
      if (tt < best_term) { best_term = tt; term_row = i; }
      row = (uint32_t)i;
      base = bd;
    }
    sa_wg_sync<NT>();  // every thread has read u[] for this search before the dual update rewrites it
    // dual update.  A tree row other than the root entered through the scanned column it is matched to, at that column's
    // distance: u[cmatch[j]] += delta - dist[j], v[j] += dist[j] - delta over the scanned columns; the root moves by delta.
    SA_WG_FOR(NT, t) {
      const uint32_t sc = scn[SA_WG_SLOT(t)];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int c = 0; c < CPT; ++c) {
        if (!(sc & (1u << c))) continue;
        const uint32_t j = t + (uint32_t)c * NT;
        const int64_t dj = dist[SA_WG_SLOT(t)][c];
        v[SA_WG_SLOT(t)][c] += dj - delta;
        const int32_t i = w.cmatch[j];
        if (i >= 0) w.u[i] += delta - dj;
      }
      if (t == 0) w.u[root] += delta;
    }
    sa_wg_sync<NT>();
    // augment (a short dependent chain: thread 0 walks it; nobody else reads the matches before the barrier below)
    if (end_col >= 0 || term_row != (int32_t)root) {  // else: the root keeps its self column
      SA_WG_FOR(NT, t) {
        if (t == 0) {
          int32_t j = end_col;
          if (end_col < 0) {  // term_row falls back to self and frees its column
            j = w.rmatch[term_row];
            w.rmatch[term_row] = -1;
          }
          for (uint32_t guard = 0; guard < 65536u; ++guard) {
            const int32_t i = w.pred[j];
            const int32_t prev = w.rmatch[i];
            w.rmatch[i] = j;
            w.cmatch[j] = i;
            if (i == (int32_t)root || prev < 0) break;
            j = prev;
          }
        }
      }
    }
    sa_wg_sync<NT>();
  }
}
