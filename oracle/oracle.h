/*
 * oracle.h — CPU restatement of Similari's association hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (similari_amd/, libsimilari_assoc.so) never links, imports or calls it.
 *
 * The reference (Rust crate similari-trackers-rs 0.26.12 at /root/reference) cannot be built in
 * this image: no cargo/rustc, no vendored crates, no network.  oracle/_ref/ therefore stays
 * empty and parity is pinned on the reference's own known-answer tests (tests/test_oracle_kat.py,
 * SURVEY.md §4 / Appendix C).  Third-party arithmetic that is not under /root/reference
 * (Cargo.toml:23-37, semver ranges only — there is no Cargo.lock):
 *   pathfinding 4.8  kuhn_munkres      restated from the published algorithm   (oracle.cpp: or_kuhn_munkres)
 *   geo 0.27         Area::unsigned_area (shoelace with first-vertex shift)    (or_polygon_area)
 *   nalgebra 0.32    SMatrix mul / cholesky / solve_lower_triangular           (kf_* helpers)
 *   wide (via ultraviolet 0.9) f32x8::reduce_add lane order                    (reduce_add8)
 *   geo 0.27         BooleanOps::difference (own areas only)  restated as convex-piece decomposition (or_own_area_shares)
 * Tie-breaking of kuhn_munkres on non-unique optima and the f32x8 lane order below 1e-6
 * relative are PARITY-UNPINNED: no reference test exercises them.
 *
 * Build: `make -C oracle` (g++ -O2 -ffp-contract=off; rustc never contracts a*b+c into an FMA).
 */
#ifndef SIMILARI_ORACLE_H
#define SIMILARI_ORACLE_H

#include "../include/similari_assoc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/distance.rs, src/track/utils.rs -------------------------------------------------- */
uint32_t or_feature_blocks(uint32_t len);                                   /* from_vec: ceil(len/8), >= 1 when len==0? see .cpp */
uint32_t or_feature_pad(const float* v, uint32_t len, float* out);          /* returns number of f32x8 blocks written */
float or_euclidean(const float* a, uint32_t blocks_a, const float* b, uint32_t blocks_b);
float or_cosine(const float* a, uint32_t blocks_a, const float* b, uint32_t blocks_b);

/* ---- src/utils/bbox.rs, src/utils/clipping.rs --------------------------------------------- */
float or_radius(const sa_box* b);
float or_area(const sa_box* b);
int or_too_far(const sa_box* l, const sa_box* r);
float or_dist_in_2r(const sa_box* l, const sa_box* r);
void or_vertices(const sa_box* b, double out_xy[8]);
uint32_t or_sh_clip(const double* subject_xy, uint32_t ns, const double* clip_xy, uint32_t nc, double* out_xy /* cap 2*16 */);
double or_polygon_area(const double* xy, uint32_t n);
double or_intersection(const sa_box* l, const sa_box* r);
int or_iou(const sa_box* l, const sa_box* r, float* out); /* 1 = Some(out), 0 = None */

/* ---- src/utils/kalman/kalman_2d_box.rs, src/trackers/kalman_prediction.rs ----------------- */
void or_kf_initiate(float pw, float vw, const sa_box* b, float mean[10], float cov[100]);
void or_kf_predict(float pw, float vw, const float mean[10], const float cov[100], float out_mean[10], float out_cov[100]);
void or_kf_update(float pw, float vw, const float mean[10], const float cov[100], const sa_box* z, float out_mean[10], float out_cov[100]);
float or_kf_distance(float pw, float vw, const float mean[10], const float cov[100], const sa_box* z);
float or_kf_distance5(float pw, const float mean5[5], const float cov25[25], const sa_box* z);
float or_kf_cost(float d2, int inverted);
void or_kf_state_box(const float mean[10], sa_box* out);
/* make_prediction: state==NULL -> initiate; predict; update; returns the box (confidence copied). */
void or_make_prediction(float pw, float vw, int has_state, float mean[10], float cov[100], const sa_box* obs, sa_box* out_box);

/* ---- src/trackers/spatio_temporal_constraints.rs, sort.rs compatible() --------------------- */
int or_constraints_validate(uint32_t n, const uint64_t* deltas, const float* max_dists, uint64_t epoch_delta, float dist);
int or_compatible(const sa_config* cfg, const sa_box* cand, uint64_t cand_epoch, const sa_box* track, uint64_t track_epoch);

/* ---- src/trackers/sort/metric.rs, visual_sort/metric.rs ------------------------------------ */
int or_positional_metric(const sa_config* cfg, const sa_box* cand, const sa_box* track,
                         const float* mean5, const float* cov25, float* out);

/* ---- voting ---------------------------------------------------------------------------------
 * pathfinding::kuhn_munkres on a dense rows x cols i64 matrix (rows <= cols), maximising. */
int or_kuhn_munkres(uint32_t rows, uint32_t cols, const int64_t* w, int64_t* out_total, uint32_t* out_assign);
int64_t or_quantise(float w); /* (w * 1e6f) as i64 */

/* distances: n entries (from, to, positional or NaN, visual or NaN) in CANONICAL order.
 * winners: for every distinct `from` in order of first appearance in `froms_unique` (caller passes the
 * candidate id list): out_to[i] = winner id (== from[i] means self) or 0 = no entry. */
int or_sort_voting(float threshold, uint32_t n_cand, uint32_t n_tracks, uint32_t n,
                   const uint64_t* from, const uint64_t* to, const float* positional,
                   uint32_t n_ids, const uint64_t* cand_ids, uint64_t* out_to, int64_t* out_total);
int or_bestfit_voting(float max_distance, uint32_t min_votes, uint32_t n,
                      const uint64_t* from, const uint64_t* to, const float* visual,
                      uint32_t n_ids, const uint64_t* cand_ids, uint64_t* out_to, double* out_weight);
int or_visual_voting(float positional_threshold, float max_feature_distance, uint32_t min_votes, uint32_t n,
                     const uint64_t* from, const uint64_t* to, const float* positional, const float* visual,
                     uint32_t n_ids, const uint64_t* cand_ids, uint64_t* out_to, uint8_t* out_type);

/* ---- src/utils/nms.rs:32-72 ---------------------------------------------- */
int or_nms(uint32_t n, const sa_box* boxes, const float* scores /* NULL / NaN = None */, float nms_threshold,
           float score_threshold /* NaN = None */, uint32_t* out_keep, uint32_t* out_n);

/* ---- src/utils/clipping/bbox_own_areas.rs:8-46 ----------------------------
 * share[i] = area(box i minus every other box that is not too_far) / (area_i + EPS), clamped to 1. */
int or_own_area_shares(uint32_t n, const sa_box* boxes, float* out_share);

/* ---- one scene-frame, end to end (the thing sa_associate replaces) --------------------------
 * total_tracks_in_store: SortVoting's track_num for plain SORT (= store size over all scenes,
 * sort/simple_api.rs:160); pass tracks->n for a single-scene store.
 * Any of the out_* matrices may be NULL. */
typedef struct or_frame_out {
  float* positional;   /* N x T,     NaN = absent */
  float* visual;       /* N x T x K, NaN = absent */
  int64_t* quantised;  /* N x T */
  uint8_t* compatible; /* N x T */
  uint64_t* track_id;  /* N, 0 = new track */
  uint8_t* voting_type;/* N */
  int64_t total_weight;/* kuhn_munkres objective of the positional stage (0 if not run) */
  uint64_t n_distances;/* entries produced by foreign_track_distances after postprocess */
} or_frame_out;
int or_associate(const sa_config* cfg, uint32_t total_tracks_in_store, const sa_tracks* tracks,
                 uint64_t epoch, const sa_detections* det, or_frame_out* out);
/* The same, with the distance stage on `shards` host threads partitioned track id % shards like the reference's TrackStore
 * (store.rs:490-493) and one vote after the shards (sort/simple_api.rs:147-162).  Same results as or_associate. */
int or_associate_sharded(const sa_config* cfg, uint32_t total_tracks_in_store, const sa_tracks* tracks,
                         uint64_t epoch, const sa_detections* det, or_frame_out* out, uint32_t shards);

#ifdef __cplusplus
}
#endif
#endif
