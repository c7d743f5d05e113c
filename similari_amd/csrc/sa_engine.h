// sa_engine.h — internal types shared by the host engine and the HIP kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/similari_assoc.h"
#include "sa_device.h"

// Fragment order of a [rows][Dp] f32 matrix (rows padded to a multiple of 32; Dp a multiple of 32): blocks of 32 rows x 8 k, each one
// contiguous kilobyte laid out [k / 4][row][k % 4] — exactly what the 64 lanes of a wave hold as the operand of four consecutive
// v_mfma_f32_32x32x2_f32 (lane l: row l & 31, k-slot l >> 5, four consecutive k), so a wave-load is ONE contiguous dwordx4 per lane: 8
// full cache lines instead of 32 B out of each of 32.  The engine keeps the track bank in this order NEXT TO the row-major bank
// (SceneDev::t_ffrag; every kernel that writes bank rows writes both), for the k-split contraction (sa_gemm.hip: gemm_mainloop_ks).
__host__ __device__ __forceinline__ size_t sa_frag_index(uint32_t r, uint32_t k, uint32_t Dp) {
  return ((size_t)(r >> 5) * (Dp >> 3) + (k >> 3)) * 256u + ((k >> 2) & 1u) * 128u + (r & 31u) * 4u + (k & 3u);
}
static inline size_t sa_frag_bytes(size_t rows, uint32_t Dp) { return (rows + 31u) / 32u * 32u * (size_t)Dp * 4u; }

// Device-visible description of one scene of the current batch (blockIdx.z selects it).
// HBM layout (all row-major, structure-of-arrays; T tracks, K bank slots, Dp = feature length padded to 32):
//   t_geo[T] 16 B | t_verts[T][8] f64 | t_epoch[T] | t_maha[T][20] (mean5 + packed Cholesky L15)
//   t_feat[T][K][Dp] f32 | t_fnorm[T][K] | t_fpresent[T][K] | t_fcount[T] | t_ids[T]
//   c_* the same for the N candidates of the frame; pos[N][T], vis[N][T][K] f32 with NaN = absent.
// Raw per-box staging record uploaded by the host: the caller's sa_box plus libm cos/sin of the angle.
// One edge of the positional vote's graph: (candidate row ->) track column, gain = quantised weight - new-track threshold.
struct alignas(16) SaEdge {
  int64_t gain;
  uint32_t col;
  uint32_t pad;
};

struct BoxRaw {
  sa_box box;
  double c, s;
};

// Every array a kernel reaches through this descriptor lives in HBM.  Pointers loaded from a struct are "generic" to the
// compiler, and generic loads/stores become flat_* instructions, which tick BOTH vmcnt and lgkmcnt (an LDS wait then also
// waits for memory) and must resolve the aperture per access: measured on the fused contraction, the flat metadata
// loads alone cost 30 us of serialisation at C2.  On the device side the members are therefore global-address-space
// pointers (same size and layout as on the host, where the qualifier is empty).
#if defined(__HIP_DEVICE_COMPILE__)
#define SA_G __attribute__((address_space(1)))
#else
#define SA_G
#endif
// whole-struct load / store through a global pointer (the implicit copy operations only bind generic references)
template <class T>
__device__ __forceinline__ T sa_ldg(const T SA_G* p) {
  T v;
  __builtin_memcpy(&v, (const T*)p, sizeof(T));
  return v;
}
template <class T>
__device__ __forceinline__ void sa_stg(T SA_G* p, const T& v) {
  __builtin_memcpy((T*)p, &v, sizeof(T));
}

struct SceneDev {
  uint32_t N, T, K, Dp;      // candidates, tracks, bank depth, feature row stride (D rounded up to 32)
  uint32_t TK, estride;      // T*K ; row stride of the edge lists
  uint32_t D, flags;         // un-padded feature length ; SCN_* bits
  uint32_t CT, RT;           // BestFit tiles: ceil(T/64), ceil(N/64)
  uint32_t nkeys, pad0;      // max-key slots = tiles of the visual cost kernel inside this scene
  uint64_t epoch;
  // stored tracks (persist across frames)
  const sa_geo SA_G* t_geo;
  const sa_ext SA_G* t_ext;   // half extents (negative hw = oriented): the axis-aligned quick reject of the positional tiles
  const double SA_G* t_verts;
  const uint64_t SA_G* t_epoch;
  const float SA_G* t_maha;
  const float SA_G* t_feat;
  const float SA_G* t_ffrag;  // the same bank in FRAGMENT order (sa_frag_index): what the k-split contraction's wave-loads read
  const float SA_G* t_fnorm;
  const uint8_t SA_G* t_fpresent;
  const uint32_t SA_G* t_fcount;
  const uint64_t SA_G* t_ids;
  // raw candidate inputs of this frame (as uploaded)
  const BoxRaw SA_G* c_raw;
  const float SA_G* c_quality;
  const float SA_G* c_own;
  const uint8_t SA_G* c_fpresent_in;
  const float SA_G* c_feat_raw;
  // derived candidate arrays (written by k_frame_prep)
  sa_geo SA_G* c_geo;
  double SA_G* c_verts;
  float SA_G* c_z;
  float SA_G* c_conf;
  float SA_G* c_feat;
  float SA_G* c_fnorm;
  uint8_t SA_G* c_usable;
  // cost matrices
  float SA_G* pos;
  float SA_G* vis;
  // BestFit vote
  uint32_t SA_G* vis_max_key;   // [nkeys] order-preserving key of the largest present weight per cost-kernel workgroup (0 = none)
  double SA_G* row_part_w;   // [CT][N] best weight of the row inside column tile ct (-1 = none)
  int32_t SA_G* row_part_t;  // [CT][N]
  double SA_G* col_part_w;   // [RT][T] best weight of the column inside row tile rt
  uint32_t SA_G* col_part_q; // [RT][T] lowest row attaining it
  unsigned long long SA_G* row_best;  // [N] vote words (SaParams::vote_words): min over the row of (weight key << 32 | column); all ones = none
  unsigned long long SA_G* col_best;  // [T] min over the column of (weight key << 32 | row)
  // Deeper banks without the weight matrix (SCN_WORDSK, the contraction's whole-track tiles): one word per candidate (track) and COUNT CLASS —
  // groups with c present observations, c = 1 .. K at [q * K + c - 1] — holding min over the row (column) of (key of the f32 sum of
  // the group's weights << 32 | column (row)): inside a class the heaviest BestFit group is the one with the smallest sum; the
  // tail, which knows the frame's max_dist, compares the classes' winners by W = c max_dist - sum.
  unsigned long long SA_G* row_cls;   // [N][K]
  unsigned long long SA_G* col_cls;   // [T][K]
  uint8_t SA_G* row_has;
  int32_t SA_G* vis_winner;
  uint8_t SA_G* col_excluded;
  // positional assignment
  uint32_t SA_G* parent;
  uint32_t SA_G* label;      // [N] general tail: head of the row list of the component rooted at this row
  uint32_t SA_G* next_row;
  uint32_t SA_G* e_cnt;      // [N] edges appended by the positional tiles; zero between frames (the tail leaves it clean)
  uint32_t SA_G* e_use;      // [N] many-workgroup tail: the counts the solver works on
  SaEdge SA_G* e_edge;       // edge records: [N][estride] (general tail) or [estride][N] (one-workgroup tail), see positional_tile
  int64_t SA_G* u;           // [N] -max gain per row, folded by the positional tiles (UNION); zero between frames
  int64_t SA_G* u_use;       // [N] many-workgroup tail: the solver's row duals
  int64_t SA_G* v;
  int32_t SA_G* rmatch;
  int32_t SA_G* cmatch;
  int64_t SA_G* dist;
  int32_t SA_G* pred;
  uint32_t SA_G* cstamp;
  uint32_t SA_G* cscan;
  int32_t SA_G* cnext;
  int64_t SA_G* rdist;
  int32_t SA_G* rnext;       // [N] general tail: rows in the component rooted at this row
  uint32_t SA_G* lab;        // [N] general tail: component root of the row (SA_NONE: takes no part)
  uint2 SA_G* crow;          // [N][SA_CROW] general tail: the first SA_CROW rows of the component rooted at a row, (row, usable-edge count), in the
                             // order they arrived (k_assign_label) — a mid-sized component of up to SA_CROW rows is gathered from ONE load
  uint32_t SA_G* cwin;       // [T] general tail, big components: lowest row bidding for the column (SA_NONE between frames)
  uint32_t SA_G* big_rows;   // [N] rows, then search roots, of the big components: one ascending segment each
  uint32_t SA_G* big_bcol;   // [N] the column a row bids for
  uint32_t SA_G* dq;         // [2 N] the general tail's queues of this frame: roots of the big components, from N on those of the mid-sized ones
  int64_t SA_G* dense;       // [N][T] gains of the components the dense solver (sa_dense.h) is working on; all zero between frames
  // results: out_track_id[N] followed by out_vote[N] in one allocation (one D2H copy)
  uint64_t SA_G* out_track_id;
  uint8_t SA_G* out_vote;
  int32_t SA_G* win_col;     // [N] winning track as a column of the table, -1 = none: what the device-side upkeep consumes
  int32_t SA_G* out_win;     // [N] the same, next to the results in mapped host memory (sa_batch_fetch_cols: a host that keeps its tracks in
                             // table order finds the winner without a hash lookup per candidate)
  uint32_t SA_G* stats;      // [8] device words: [1] [3] [4] [5] the general tail's list top / queue length / ticket / row workgroups done; [0] = 1 when the frame was ill-conditioned for the euclidean expansion
  uint32_t SA_G* out_stats;  // [4] the same, moved next to the results (mapped host memory) and re-armed by the assignment tail
  unsigned long long SA_G* out_done;  // one word on a cache line of its own behind the results (mapped host memory): the one-workgroup tail
                                      // stores its launch's sequence number here once every result of the scene has left (k_assign_small, done_seq)
  int64_t SA_G* quant;  // optional N x T tap
  // SA_FLAG_TAP (parity tests): what the timed launches themselves produced, copied out by the assignment tail before it re-arms /
  // consumes it — the vote words as the first phase left them and the edge counts of the positional tiles (the edge records stay in
  // e_edge).  Null otherwise: the tails test one wave-uniform pointer.
  unsigned long long SA_G* tap_row_best;  // [N]
  unsigned long long SA_G* tap_col_best;  // [T]
  uint32_t SA_G* tap_ecnt;                // [N]
};
// The general tail's queue words live on their own 128-byte line of the scene's stats block: they are touched ONLY by agent-scope atomics
// while k_assign_solve runs (workgroups on different XCDs), and stats[0] next door is read and written with plain accesses by that
// kernel's first thread — a line held in one XCD's L2 by plain accesses and updated by other XCDs' atomics is not something to rely on.
// Zeroed by k_assign_label's first thread.
#define SA_CROW 32u
#define SA_QW_TOP 32       // top of the dense solver's row lists
#define SA_QW_LEN 33       // queue length
#define SA_QW_TICKET 34    // next ticket
#define SA_QW_DONE 35      // row workgroups through with their rows
#define SA_QW_MLEN 36      // the same two for the queue of mid-sized components
#define SA_QW_MTICKET 37
#define SCN_HAS_FEATS 1u
#define SCN_HAS_QUALITY 2u
#define SCN_HAS_OWN 4u
#define SCN_HAS_FPRESENT 8u
#define SCN_WORDSK 32u    // the vote words of this frame are per count class (row_cls / col_cls), the contraction's whole-track tiles'
#define SA_CLS_MAXK 8u    // deepest bank the class words serve
#define SCN_WORDS10 16u   // the vote words of this frame carry a 10-bit index below a 54-bit weight key (k_bestfit_tile, deeper banks)

// Engine-wide constants, passed to kernels by value.
struct SaParams {
  int32_t positional_kind;
  int32_t visual_kind;
  float positional_threshold;   // IoU threshold as given (cells below it are absent)
  float visual_threshold;
  int64_t threshold_q;          // (new-track threshold * 1e6f) as i64
  uint32_t min_votes;
  uint32_t min_track_len;
  float min_confidence;
  float visual_minimal_area;
  float visual_minimal_quality_use;
  float visual_minimal_own_area_use;
  float kf_position_weight;
  float kf_velocity_weight;
  uint32_t Dp;                  // feature row stride of this engine (D rounded up to 32)
  uint64_t max_idle;
  sa_constraints cons;
  uint32_t vote_words;          // per launch: the contraction's BestFit epilogue reduces into row_best / col_best (64-bit atomic minima)
                                // instead of writing per-tile partials; the one-workgroup tail reads and re-arms them
  uint32_t eu_mfma;             // per launch: euclidean distances through the matrix-core contraction (expansion + flagged direct recompute)
  float eu_rho;                 // a cell with d^2 < eu_rho (|a|^2 + |b|^2) is recomputed directly
  uint32_t force_general;       // SA_FLAG_GENERAL_TAIL: the many-workgroup assignment tail (and the launches that feed it) whatever the frame size
  uint32_t row_major_tiles;     // order of the contraction's tiles: 1 = the default (stand-alone contraction row by row, fused first phase XCD-aware),
                                // 0 = XCD-aware everywhere (SA_FLAG_XCD_TILES), 2 = row by row everywhere (SA_FLAG_ROW_TILES); sa_gemm.hip
  uint32_t no_yield;            // SA_FLAG_NO_YIELD
  uint32_t ks_yield;            // k-split loops of the fused first phase: the matrix waves nap after every (ks_yield & 255)-th k-step (0: never), (ks_yield >> 8) x 64 cycles (0: 64)
  uint32_t staged_loop;         // SA_FLAG_STAGED_LOOP: the fused first phase's contraction tiles on the LDS-staged main loop (row-major bank)
  int32_t gemm_plan;            // sa_config.gemm_plan - 1: the contraction's tile plan pinned (tuning / tests), -1 = tile_plan()'s own choice
};

// Profile mode (SA_FLAG_PROFILE): while sa_prof_start is set, the per-frame launches go through hipExtLaunchKernelGGL,
// which stamps the two events with the dispatch's OWN begin / end timestamps — the same clock rocprofv3's kernel trace
// reads — instead of bracketing the launch with hipEventRecord (that adds ~2 us of command-processor time per kernel).
extern thread_local hipEvent_t sa_prof_start, sa_prof_stop;
// sa_done_event: set by the caller of a launcher for ONE launch — the dispatch carries the event as its own completion signal (the
// frame's last kernel signals "results are in" itself, instead of a marker packet queued behind it: one command-processor round trip
// less per pipelined frame).  Consumed (reset to nullptr) by the launch that takes it; ignored while profiling.
extern thread_local hipEvent_t sa_done_event;
#define SA_LAUNCH(kern, grid, block, shmem, st, ...)                                                           \
  do {                                                                                                         \
    if (sa_prof_start) hipExtLaunchKernelGGL(kern, grid, block, shmem, st, sa_prof_start, sa_prof_stop, 0, __VA_ARGS__); \
    else if (sa_done_event) {                                                                                  \
      hipExtLaunchKernelGGL(kern, grid, block, shmem, st, nullptr, sa_done_event, 0, __VA_ARGS__);             \
      sa_done_event = nullptr;                                                                                 \
    } else hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);                                      \
  } while (0)
// ---- launchers (sa_kernels.hip / sa_gemm.hip).  All enqueue on `st` and return the launch status. ----

struct PrepTrackArgs {
  const BoxRaw* raw;        // [n] compact
  const uint32_t* slots;    // [n] destination rows
  const uint64_t* epochs;   // [n]
  const uint64_t* ids;      // [n]
  const float* kf_mean;     // [n][5] or nullptr
  const float* kf_cov;      // [n][25] or nullptr
  uint32_t n;
  sa_geo* geo;
  sa_ext* ext;
  double* verts;
  uint64_t* t_epoch;
  uint64_t* t_ids;
  float* maha;
};
hipError_t sa_launch_prep_tracks(const PrepTrackArgs& a, const SaParams& p, hipStream_t st);

// Pads `rows` feature rows of length D to Dp, scatters row r to dst[(slots ? slots[r / K] * K + r % K : r)],
// and stores the squared norm; `present` (or nullptr) zeroes absent rows.
hipError_t sa_launch_pad_features(const float* src, uint32_t rows, uint32_t D, uint32_t Dp, uint32_t K,
                                  const uint32_t* slots, const uint8_t* present, float* dst, float* norms,
                                  uint8_t* dst_present, uint32_t* fcount, hipStream_t st, float* dst_frag = nullptr);
// one launch of the ingest kernel moves up to SA_COPY_SEGS pinned-host -> HBM segments (device-visible source addresses)
#define SA_COPY_SEGS 12
struct SaCopySegs {
  struct Seg { const void* src; void* dst; size_t bytes; } s[SA_COPY_SEGS];
  uint32_t n;
};
hipError_t sa_launch_ingest(const SaCopySegs& segs, uint32_t blocks, hipStream_t st, hipEvent_t done = nullptr);
hipError_t sa_launch_gather_rows(const void* src, void* dst, const uint32_t* index, uint32_t rows, uint32_t row_bytes,
                                 hipStream_t st);
// all arrays of a track table compacted in one launch (sa_tracks_remove): dst[a][r] = src[a][index[r]]
#define SA_TABLE_ARRAYS 12
struct SaGatherTable {
  const void* src[SA_TABLE_ARRAYS];
  void* dst[SA_TABLE_ARRAYS];
  uint32_t row_bytes[SA_TABLE_ARRAYS];
  uint32_t n_arrays, rows;
  const uint32_t* index;   // [rows] device-visible (mapped pinned memory)
  // the feature bank's fragment-order twin follows the rows of array `frag_array` (row-major source -> the twin, in place: nothing
  // reads the twin while the gather runs); frag = nullptr: no bank
  float* frag;
  uint32_t frag_array, frag_K, frag_Dp;
};
hipError_t sa_launch_gather_table(const SaGatherTable& g, hipStream_t st, hipEvent_t done = nullptr);
// ... of several scenes in one launch (blockIdx.y = scene; the arguments travel by value: 12 x 256 bytes fit a dispatch's 4 KB)
#define SA_GATHER_SET 12
struct SaGatherTables {
  SaGatherTable t[SA_GATHER_SET];
  uint32_t n;
};
hipError_t sa_launch_gather_tables(const SaGatherTables& g, hipStream_t st, hipEvent_t done = nullptr);

// first launch of a frame: positional tiles + frame-preparation blocks
// prep: 1 = positional tiles + preparation blocks, 0 = positional tiles only (a lean frame on the one-workgroup tail), 2 = preparation
// blocks only (what a lean frame left out, on demand: sa_tracks_apply, the visual tap), 3 = positional tiles + the preparation blocks'
// RESET half only (a lean frame on the many-workgroup tail, whose per-row / per-column state lives in HBM)
hipError_t sa_launch_frame(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxT, int visual, const SaParams& p,
                           hipStream_t st, int prep = 1);
hipError_t sa_launch_slot_init(uint32_t* e_cnt, int64_t* u, uint32_t n_rows, uint32_t* parent, uint32_t n_vertices, hipStream_t st);
hipError_t sa_launch_positional_dense(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxT,
                                      const SaParams& p, hipStream_t st);
hipError_t sa_launch_visual(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxTK,
                            const SaParams& p, hipStream_t st, bool partials);
// heterogeneous first phase of a VisualSORT frame (contraction tiles + positional tiles + preparation blocks in one launch);
// hipErrorNotSupported = not applicable, use sa_launch_frame + sa_launch_visual
hipError_t sa_launch_frame_visual(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxT, uint32_t K, uint32_t D,
                                  const SaParams& p, hipStream_t st, bool partials, int prep = 1, bool kpass = false, bool general_tail = false);
bool sa_frame_visual_ok(uint32_t n_scenes, uint32_t maxN, uint32_t maxT, uint32_t K, uint32_t D, const SaParams& p, bool class_words);
void sa_visual_tile(int visual_kind, bool eu_mfma, uint32_t maxN, uint32_t maxTK, uint32_t ns, uint32_t Dp, int32_t plan_override, uint32_t* bm, uint32_t* bn);
hipError_t sa_launch_bestfit(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxT,
                             const SaParams& p, hipStream_t st, int stage);
// stage 1 label + push, 3 solve + results (2 / 4: the same with the visual vote read from the vote words — the label kernel turns
// them into verdicts, the solver re-arms them); stage 5 = the whole tail in ONE workgroup per scene (requires sa_small_tail_ok(maxN, maxT, ..);
// 8: with vote words)
#define SA_SMALL_N 1024
// ... up to SA_SMALL_T tracks (two columns per thread of that workgroup: k_assign_small<.., TC = 2>), and up to SA_SMALL_T detections as
// well (k_assign_small2: two rows per thread too; or one row and four columns: 1024 x 4096) — never with the 10-bit index words
#define SA_SMALL_T 2048
static inline bool sa_small_tail_ok(uint32_t maxN, uint32_t maxT, uint32_t words) {
  if (maxN <= SA_SMALL_N && maxT <= SA_SMALL_N) return true;
  if (words == 2u) return false;
  if (maxT > SA_SMALL_T) return maxN <= SA_SMALL_N && maxT <= 2u * SA_SMALL_T;   // (k_assign_small2<.., 1, 4>: 1024 x 4096)
  return maxN <= SA_SMALL_T;   // (k_assign_small<.., TC = 2> / k_assign_small2)
}
// done_seq != 0 (stages 5 / 8): every scene's workgroup reports the end of its results itself, by storing done_seq to SceneDev::out_done —
// the host polls that word instead of waiting for a completion signal of the dispatch (a dispatch that carries one holds the NEXT
// dispatch of its queue back by ~4.6 us on this stack: scripts/gpu_ab_timeline.sh, NOTES section 0a)
hipError_t sa_launch_assign(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxT,
                            const SaParams& p, hipStream_t st, int stage, uint64_t done_seq = 0);
hipError_t sa_launch_quant_tap(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxT,
                               hipStream_t st);
hipError_t sa_launch_frag_reorder(const float* src, uint32_t rows, uint32_t dp, float* dst, hipStream_t st);
// Standalone contraction for sa_feature_distance_matrix: out[n][t] = cosine / euclid distance (no gating).
hipError_t sa_launch_distance_matrix(int kind, const float* a, const float* an, const float* b, const float* bn,
                                     uint32_t n, uint32_t t, uint32_t dp, float* out, hipStream_t st, int32_t plan_override = -1);

// ---- device-side track upkeep (sa_upkeep.hip) ----
struct ApplyArgs {
  const BoxRaw* c_raw;       // [n] candidates of the slot
  const int32_t* win_col;    // [n] winning column or -1
  const uint32_t* new_row;   // [n] table row for a candidate that starts a track; nullptr: drawn on the device — T0 + its rank among the
  const uint64_t* new_ids;   // [n] its id                                                  candidates without a winner, in candidate order,
  uint32_t n;                //                                                              id = id_base + 1 + (id_per_candidate ? index : rank)
  uint32_t T0;
  uint64_t id_base;
  int32_t id_per_candidate;
  uint64_t epoch;
  float* kf;                 // [T][110] Kalman mean(10) + covariance(100)
  sa_geo* geo;
  sa_ext* ext;
  double* verts;
  uint64_t* t_epoch;
  uint64_t* t_ids;
  float* maha;
  sa_box* out_pred;          // [n] device view of pinned host memory
  // The Kalman dispatch of a VisualSORT upkeep queued behind the association also takes the candidates' feature rows OUT OF THE CALLER'S
  // MEMORY (a registered device block read in place) into the slot's own buffer — n more blocks, one row each — so that the bank
  // dispatch behind it reads engine memory only: when this dispatch has retired, nothing of the engine reads caller memory any more.
  const float* copy_src;     // nullptr: nothing to copy
  float* copy_dst;
  uint32_t copy_row_floats;
};
struct BankArgs {
  const BoxRaw* c_raw;
  const int32_t* win_col;
  const uint32_t* new_row;   // (nullptr: T0 + rank, as in ApplyArgs)
  uint32_t T0;
  uint32_t n, K, Dp;
  const float* c_feat;       // [n][Dp] padded candidate features (nullptr: the frame carried none)
  const float* c_fnorm;       // (nullptr: the frame ran lean — the step forms the new rows' norms itself; only with D == Dp)
  const uint8_t* c_fpresent_in;
  const float* c_quality;
  const float* c_own;
  float* t_feat;
  float* t_ffrag;            // the bank's fragment-order twin (every row written to t_feat is written here too)
  float* t_fnorm;
  uint8_t* t_fpresent;
  float* t_fquality;
  uint32_t* t_fcount;
  float minimal_area, q_collect, own_collect;
};
hipError_t sa_launch_apply(const ApplyArgs& a, const BankArgs* b, const SaParams& p, hipStream_t st, hipEvent_t done = nullptr, int part = 0);
// The upkeep of EVERY scene of a request set in one launch per half (Batch*::predict: 64 scenes are two dispatches, not 128): the
// per-scene arguments travel as an array in the bank's arena (they ride in the set's one upload), blockIdx.y selects the scene.
// part 1 = the Kalman halves (+ the row copies out of a caller's device block), 2 = the bank halves.  max_blocks = the largest
// per-scene block count of that half (sa_apply_set_blocks).
struct ApplyScene {
  ApplyArgs a;
  BankArgs b;
};
static inline uint32_t sa_apply_set_blocks(uint32_t n, bool copies, int part) { return part == 2 ? n : (n + 3u) / 4u + (copies ? n : 0u); }
hipError_t sa_launch_apply_set(const ApplyScene* scenes, uint32_t n_scenes, uint32_t max_blocks, uint32_t K, const SaParams& p, hipStream_t st,
                               hipEvent_t done, int part);
// oriented boxes after sa_launch_apply: the host's libm cos / sin of a refreshed row's angle -> its polygon
struct SaPolyFix {
  uint32_t row, pad;
  float xc, yc, aspect, height;
  double c, s;
};
hipError_t sa_launch_apply_polygons(const SaPolyFix* fix, uint32_t n, double* verts, hipStream_t st);
// NMS: mask[n][ceil(n/64)] words + keep[n] flags for rank-sorted boxes (at most SA_NMS_MAX)
#define SA_NMS_MAX 16384u
hipError_t sa_launch_nms(const BoxRaw* raw, uint32_t n, float thr, uint64_t* mask, uint8_t* keep, hipStream_t st);
hipError_t sa_launch_own_areas(const BoxRaw* raw, uint32_t n, float* share, uint32_t* status, hipStream_t st);
// spill path: the `count` boxes listed (indices into raw) with neighbour polygons and interval lists in HBM scratch
size_t sa_own_big_scratch_bytes(uint32_t n, uint32_t batch);
hipError_t sa_launch_own_areas_big(const BoxRaw* raw, uint32_t n, const uint32_t* list, uint32_t count, float* share, uint32_t* status,
                                   void* scratch, hipStream_t st);

const char* sa_kernel_name(int id);
