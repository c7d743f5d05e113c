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
//   * per-row state (dual u, match) and the per-column match / predecessor live in LDS (k_assign_small; k_assign_solve when
//     12 N + 8 T bytes fit) or in HBM behind the workgroup's own L1 (k_assign_solve on frames beyond that).
// One search step = one barrier + one L2 round trip (~1.5 k cycles) whatever the density; 640 rows / 140 k edges: ~6 k steps.
//
// Written once for both worlds like the cooperative solver: on the device a "thread loop" runs its body once, for this thread;
// in the host emulation it runs NT times, per-thread values live in arrays of SA_WG_SLOTS(NT) elements.
#pragma once
#include "sa_device.h"
#include <type_traits>

#if defined(__HIPCC__)
#define SA_WG_FN __device__ __forceinline__
#define SA_WG_SLOTS(NT) 1
#define SA_WG_SLOT(t) 0
#define SA_WG_FOR(NT, t) for (uint32_t t [[maybe_unused]] = threadIdx.x & (uint32_t)((NT) - 1), _sa_wg_once = 1; _sa_wg_once; _sa_wg_once = 0)
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

// The 32-bit variant (K32): when every gain of the component is below 2^21 - 1 (IoU weights: at most 1e6) and the scene has at most
// 2048 tracks, every quantity of the search — duals, reduced costs, distances — stays below 2^24 and a key is (distance << 11 |
// column) in ONE 32-bit word: half the vector instructions in the relax step and a one-pass minimum.
#define SA_DENSE_K32_MAXGAIN ((1 << 21) - 2)
#define SA_DENSE_K32_MAXT 2048u
#define SA_DENSE_KEY32_NONE 0xffffffffu
SA_HD uint32_t sa_dense_key32(int32_t d, uint32_t j) {
  const uint32_t c = d > (1 << 21) - 1 ? (1u << 21) - 1u : (uint32_t)d;
  return (c << 11) | (j & 2047u);
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
// Workgroup minimum, every thread receives it.  part: [2][NT / 64] 64-bit slots in LDS, alternated by `parity` so that ONE barrier
// per call is enough (a wave that races ahead into the next call writes the other half).
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
__device__ __forceinline__ uint32_t sa_wg_min_u32(const uint32_t* key, unsigned long long* part, uint32_t parity) {
  constexpr int W = NT / 64;
  const uint32_t k = sa_wave_min_u32(key[0]);
  if constexpr (W == 1) return k;
  uint32_t* p32 = (uint32_t*)(part + parity * W);
  if ((threadIdx.x & 63u) == 0) p32[threadIdx.x >> 6] = k;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  uint32_t m = p32[0];
#pragma unroll
  for (int w = 1; w < W; ++w) {
    const uint32_t o = p32[w];
    m = o < m ? o : m;
  }
  return m;
}
// (NT == 64: ONE wavefront of a larger workgroup runs the solver on its own — its LDS accesses execute in program order, only the
// compiler has to be kept from moving them across)
template <int NT>
__device__ __forceinline__ void sa_wg_sync() {
  if constexpr (NT == 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else __syncthreads();
}
#else
template <int NT>
inline unsigned long long sa_wg_min_u64(const unsigned long long* key, unsigned long long*, uint32_t) {
  unsigned long long m = SA_DENSE_KEY_NONE;
  for (int t = 0; t < NT; ++t) m = key[t] < m ? key[t] : m;
  return m;
}
template <int NT>
inline uint32_t sa_wg_min_u32(const uint32_t* key, unsigned long long*, uint32_t) {
  uint32_t m = SA_DENSE_KEY32_NONE;
  for (int t = 0; t < NT; ++t) m = key[t] < m ? key[t] : m;
  return m;
}
template <int NT>
inline void sa_wg_sync() {}
#endif

#if !defined(__HIPCC__)
static uint64_t sa_dense_emu_steps = 0, sa_dense_emu_searches = 0, sa_dense_emu_comps = 0;  // emulation statistics (tests/emu)
#endif

struct sa_dense_ws {
  const int64_t* gain;     // dense gains, row r at gain[r * ld + j]; 0 = no usable edge (excluded columns are never written)
  size_t ld;
  uint32_t T;              // columns of the scene
  int64_t* u;              // [N] row duals: -(heaviest usable gain) on entry
  int32_t* rmatch;         // [N] -1, or the column the greedy start gave the row
  int32_t* cmatch;         // [T] -1, or the row the greedy start gave the column
  int32_t* pred;           // [T] scratch: the tree row through which a SCANNED column was reached (published when the column is scanned)
  unsigned long long* part;  // device: [2][NT / 64] reduction slots in LDS
};

// Solves one component: `roots` = its rows the greedy start left unmatched (ascending), n_roots of them.  NT threads, thread t owns
// the columns t, t + NT, ... (CPT of them: T <= NT * CPT).  Every thread must call it; control flow is uniform.  K32: the 32-bit
// variant (the caller has checked SA_DENSE_K32_MAXGAIN and SA_DENSE_K32_MAXT).  GAIN32 (with K32): w.gain points at int32_t cells
// (the one-wavefront solver of the general tail keeps a component's matrix in LDS: NT = 64, CPT = 1, columns renumbered 0..63).
// wave-uniform values the compiler cannot prove uniform (they come out of LDS): into a scalar register, so that the row's address and the
// comparisons against it are scalar arithmetic
#if defined(__HIP_DEVICE_COMPILE__)
#define SA_WG_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define SA_WG_UNIFORM(x) (x)
#endif

template <int NT, int CPT, bool K32, bool GAIN32 = false>
SA_WG_FN void sa_assign_component_dense(const sa_dense_ws& w, const uint32_t* roots, uint32_t n_roots) {
  using D = typename std::conditional<K32, int32_t, int64_t>::type;              // duals, distances
  using KEY = typename std::conditional<K32, uint32_t, unsigned long long>::type;
  // "not labelled yet" is a distance no relaxation can reach (every real one is below 2^24 / 2^48 by the bounds above); its key
  // saturates, so it can only win the minimum when nothing real is left, and then bd >= best_term ends the search
  const D INF = K32 ? (D)(1 << 28) : (D)(1ll << 56);
  const KEY NONE = K32 ? (KEY)SA_DENSE_KEY32_NONE : (KEY)SA_DENSE_KEY_NONE;
  D v[SA_WG_SLOTS(NT)][CPT];      // column duals of this thread's columns (0 on entry: a component's columns are untouched)
  D dist[SA_WG_SLOTS(NT)][CPT];
  int32_t pred[SA_WG_SLOTS(NT)][CPT];
  uint32_t scn[SA_WG_SLOTS(NT)];  // bit c: column c of this thread is scanned in the running search
  SA_WG_FOR(NT, t) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int c = 0; c < CPT; ++c) { v[SA_WG_SLOT(t)][c] = 0; dist[SA_WG_SLOT(t)][c] = INF; pred[SA_WG_SLOT(t)][c] = -1; }
  }
  uint32_t parity = 0;
#if !defined(__HIPCC__)
  sa_dense_emu_comps += 1;
  sa_dense_emu_searches += n_roots;
#endif
  for (uint32_t ri = 0; ri < n_roots; ++ri) {
    const uint32_t root = roots[ri];
    D best_term = (D)(-w.u[root]);  // reduced cost of the root's own self column
    int32_t term_row = (int32_t)root;
    int32_t end_col = -1;
    D delta = best_term;
    SA_WG_FOR(NT, t) {
      scn[SA_WG_SLOT(t)] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int c = 0; c < CPT; ++c) dist[SA_WG_SLOT(t)][c] = INF;
    }
    uint32_t row = root;
    D base = 0;
    D ur = (D)w.u[root];
    // (every pass scans one more column of the component: the cap can only bite on corrupted state — a wrong answer the tests catch
    // instead of a kernel spinning on a GPU box)
    for (uint32_t guard = 0; guard < 65536u; ++guard) {
#if !defined(__HIPCC__)
      sa_dense_emu_steps += 1;
#endif
      // relax `row` (entered the tree at distance `base`, dual ur) and form this thread's best key — straight-line code
      KEY key[SA_WG_SLOTS(NT)];
      SA_WG_FOR(NT, t) {
        D g[CPT];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < CPT; ++c) {
          const uint32_t j = t + (uint32_t)c * NT;
          const size_t at = (size_t)row * w.ld + (j < w.T ? j : 0u);
          // (K32, little endian: the low word of the i64 cell; GAIN32: the matrix itself holds 32-bit cells)
          const D gv = GAIN32 ? (D)((const int32_t*)w.gain)[at] : K32 ? (D)((const int32_t*)w.gain)[2 * at] : (D)w.gain[at];
          g[c] = j < w.T ? gv : (D)0;
        }
        KEY k = NONE;
        const uint32_t sc = scn[SA_WG_SLOT(t)];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < CPT; ++c) {
          const uint32_t j = t + (uint32_t)c * NT;
          const bool open = !(sc & (1u << c));
          const D d = base + (-g[c] - ur - v[SA_WG_SLOT(t)][c]);
          const bool better = g[c] > 0 && open && d < dist[SA_WG_SLOT(t)][c];
          dist[SA_WG_SLOT(t)][c] = better ? d : dist[SA_WG_SLOT(t)][c];
          pred[SA_WG_SLOT(t)][c] = better ? (int32_t)row : pred[SA_WG_SLOT(t)][c];
          KEY kc;
          if constexpr (K32) kc = sa_dense_key32((int32_t)dist[SA_WG_SLOT(t)][c], j);
          else kc = sa_dense_key((int64_t)dist[SA_WG_SLOT(t)][c], j);
          kc = open ? kc : NONE;
          k = kc < k ? kc : k;
        }
        key[SA_WG_SLOT(t)] = k;
      }
      D bd;
      uint32_t bj;
      if constexpr (K32) {
        const uint32_t m = sa_wg_min_u32<NT>(key, w.part, parity);
        bd = (D)(m >> 11);
        bj = m & 2047u;
      } else {
        const unsigned long long m = sa_wg_min_u64<NT>(key, w.part, parity);
        bd = (D)(m >> 16);
        bj = (uint32_t)(m & 0xffffu);
      }
      parity ^= 1u;
      if (bd >= best_term) { delta = best_term; break; }  // a self column ends the path (also: nothing labelled is left)
      // the column's owner marks it scanned and publishes the tree row it was reached through (the augmentation walks these)
      SA_WG_FOR(NT, t) {
        if (t == bj % NT) {
          int32_t pr = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
          for (int c = 0; c < CPT; ++c) pr = (uint32_t)c == bj / NT ? pred[SA_WG_SLOT(t)][c] : pr;
          scn[SA_WG_SLOT(t)] |= 1u << (bj / NT);
          w.pred[bj] = pr;
        }
      }
      const int32_t i = SA_WG_UNIFORM(w.cmatch[bj]);
      if (i < 0) { end_col = (int32_t)bj; delta = bd; break; }               // free real column
      ur = (D)w.u[i];
      const D tt = bd - ur;
      if (tt < best_term) { best_term = tt; term_row = i; }
      row = (uint32_t)i;
      base = bd;
    }
    sa_wg_sync<NT>();  // every thread has read u[] for this search before the dual update rewrites it; pred[] is published
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
        const D dj = dist[SA_WG_SLOT(t)][c];
        v[SA_WG_SLOT(t)][c] += dj - delta;
        const int32_t i = w.cmatch[j];
        if (i >= 0) w.u[i] += (int64_t)(delta - dj);
      }
      if (t == 0) w.u[root] += (int64_t)delta;
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
