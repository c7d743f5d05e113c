// sa_engine.h — internal types shared by the host engine and the HIP kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/similari_assoc.h"
#include "sa_device.h"

// Device-visible description of one scene of the current batch (blockIdx.z selects it).
// HBM layout (all row-major, structure-of-arrays; T tracks, K bank slots, Dp = feature length padded to 32):
//   t_geo[T] 16 B | t_verts[T][8] f64 | t_epoch[T] | t_maha[T][20] (mean5 + packed Cholesky L15)
//   t_feat[T][K][Dp] f32 | t_fnorm[T][K] | t_fpresent[T][K] | t_fcount[T] | t_ids[T]
//   c_* the same for the N candidates of the frame; pos[N][T], vis[N][T][K] f32 with NaN = absent.
// Raw per-box staging record uploaded by the host: the caller's sa_box plus libm cos/sin of the angle.
struct BoxRaw {
  sa_box box;
  double c, s;
};

struct SceneDev {
  uint32_t N, T, K, Dp;      // candidates, tracks, bank depth, feature row stride (D rounded up to 32)
  uint32_t TK, estride;      // T*K ; row stride of the edge lists
  uint32_t D, flags;         // un-padded feature length ; SCN_* bits
  uint32_t CT, RT;           // BestFit tiles: ceil(T/64), ceil(N/64)
  uint64_t epoch;
  // stored tracks (persist across frames)
  const sa_geo* t_geo;
  const double* t_verts;
  const uint64_t* t_epoch;
  const float* t_maha;
  const float* t_feat;
  const float* t_fnorm;
  const uint8_t* t_fpresent;
  const uint32_t* t_fcount;
  const uint64_t* t_ids;
  // raw candidate inputs of this frame (as uploaded)
  const BoxRaw* c_raw;
  const float* c_quality;
  const float* c_own;
  const uint8_t* c_fpresent_in;
  const float* c_feat_raw;
  // derived candidate arrays (written by k_frame_prep)
  sa_geo* c_geo;
  double* c_verts;
  float* c_z;
  float* c_conf;
  float* c_feat;
  float* c_fnorm;
  uint8_t* c_usable;
  // cost matrices
  float* pos;
  float* vis;
  // BestFit vote
  uint32_t* vis_max_key;   // [SA_MAXKEY_SHARDS] order-preserving keys; max over the shards = BestFit max_dist
  double* row_part_w;   // [N][CT] best weight of the row inside column tile ct (-1 = none)
  int32_t* row_part_t;  // [N][CT]
  double* col_part_w;   // [RT][T] best weight of the column inside row tile rt
  uint32_t* col_part_q; // [RT][T] lowest row attaining it
  uint8_t* row_has;
  int32_t* vis_winner;
  uint8_t* col_excluded;
  // positional assignment
  uint32_t* parent;
  uint32_t* label;
  uint32_t* next_row;
  uint32_t* e_cnt;
  uint32_t* e_col;
  int64_t* e_gain;
  int64_t* u;
  int64_t* v;
  int32_t* rmatch;
  int32_t* cmatch;
  int64_t* dist;
  int32_t* pred;
  uint32_t* cstamp;
  uint32_t* cscan;
  int32_t* cnext;
  int64_t* rdist;
  int32_t* rnext;
  // results: out_track_id[N] followed by out_vote[N] in one allocation (one D2H copy)
  uint64_t* out_track_id;
  uint8_t* out_vote;
  int64_t* quant;  // optional N x T tap
};
#define SA_MAXKEY_SHARDS 64
#define SCN_HAS_FEATS 1u
#define SCN_HAS_QUALITY 2u
#define SCN_HAS_OWN 4u
#define SCN_HAS_FPRESENT 8u

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
  uint32_t Dp;                  // feature row stride of this engine (D rounded up to 32)
  uint64_t max_idle;
  sa_constraints cons;
};

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
                                  uint8_t* dst_present, uint32_t* fcount, hipStream_t st);
hipError_t sa_launch_gather_rows(const void* src, void* dst, const uint32_t* index, uint32_t rows, uint32_t row_bytes,
                                 hipStream_t st);

hipError_t sa_launch_positional(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxT,
                                const SaParams& p, hipStream_t st);
hipError_t sa_launch_visual(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxTK,
                            const SaParams& p, hipStream_t st);
// init of the per-frame state + candidate preparation + candidate feature padding/norms, one launch
hipError_t sa_launch_frame_prep(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxT, int visual,
                                const SaParams& p, hipStream_t st);
hipError_t sa_launch_bestfit(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxT,
                             const SaParams& p, hipStream_t st, int stage);
// stage 0 edges+union, 1 label, 2 next, 3 solve, 4 finalize; stage 5 = stages 1-4 fused in ONE workgroup per
// scene (requires maxN <= SA_SMALL_N)
#define SA_SMALL_N 1024
hipError_t sa_launch_assign(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxT,
                            const SaParams& p, hipStream_t st, int stage);
hipError_t sa_launch_quant_tap(const SceneDev* scenes, uint32_t n_scenes, uint32_t maxN, uint32_t maxT,
                               hipStream_t st);
// Standalone contraction for sa_feature_distance_matrix: out[n][t] = cosine / euclid distance (no gating).
hipError_t sa_launch_distance_matrix(int kind, const float* a, const float* an, const float* b, const float* bn,
                                     uint32_t n, uint32_t t, uint32_t dp, float* out, hipStream_t st);

const char* sa_kernel_name(int id);
