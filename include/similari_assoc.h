/*
 * similari_assoc.h — C ABI of the MI355X-native association engine.
 *
 * This is the drop-in boundary for ONE hot path of insight-platform/Similari:
 * the per-frame  N_candidates x T_tracks  cost matrices (cosine / euclidean /
 * IoU / Mahalanobis, pair pre-filter) and the assignment solve (BestFit visual
 * vote + max-weight "Hungarian" vote).  In the reference that path is the pair
 * of statements
 *
 *     store.foreign_track_distances(tracks, 0, false);  voting.winners(dists)
 *
 * inside  Sort::predict_with_scene          src/trackers/sort/simple_api.rs:147-162
 *         VisualSort::predict_with_scene    src/trackers/visual_sort/simple_api.rs:172-187
 *         BatchSort::predict / voting_thread        src/trackers/sort/batch_api.rs:269-288, 83-98
 *         BatchVisualSort::predict / voting_thread  src/trackers/visual_sort/batch_api.rs:296-315, 92-100
 *
 * Everything here is plain C: pointers, sizes, POD structs, int status codes.
 * A Rust `extern "C"` block (or cgo / ctypes) binds it 1:1 — see INTEGRATION.md.
 *
 * Ownership: the caller owns every buffer passed in or out.  The engine copies
 * inputs to device-resident structure-of-arrays storage; track state (boxes,
 * Kalman projection, feature banks) persists on the device across frames, the
 * way the reference keeps it inside TrackStore.  Handles are freed only by the
 * matching *_destroy.  No callbacks.
 *
 * Threading: calls on one handle are not re-entrant (the reference's predict
 * takes &mut self).  Distinct handles (one per GPU / process) are independent.
 *
 * Errors: every function returns an sa_status (0 = ok, < 0 = error) and never
 * aborts; sa_last_error() returns a human-readable message for the last error
 * on that handle.  The reference's assert!s (aspect/height > 0 bbox.rs:453-456,
 * confidence in [0,1] bbox.rs:123-126, ids > 0 sort/voting.rs:57) become
 * SA_ERR_BAD_ARG.
 */
#ifndef SIMILARI_ASSOC_H
#define SIMILARI_ASSOC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_API_VERSION 2u

typedef enum sa_status {
  SA_OK = 0,
  SA_ERR_BAD_ARG = -1,     /* null pointer, bad size, reference assert! violated */
  SA_ERR_OOM = -2,         /* host or device allocation failed                   */
  SA_ERR_HIP = -3,         /* a HIP runtime call failed (message has the code)   */
  SA_ERR_UNSUPPORTED = -4, /* valid request this build cannot serve              */
  SA_ERR_NOT_FOUND = -5,   /* unknown scene / track id                           */
  SA_ERR_STATE = -6,       /* call sequence violated (e.g. fetch before run)     */
  SA_ERR_NO_DEVICE = -7    /* no gfx950 device visible: the engine has NO CPU fallback */
} sa_status;

/* PositionalMetricType  src/trackers/sort.rs:366-371 */
typedef enum sa_positional_kind { SA_POS_IOU = 0, SA_POS_MAHALANOBIS = 1 } sa_positional_kind;
/* VisualSortMetricType  src/trackers/visual_sort/metric.rs:20-24 */
typedef enum sa_visual_kind { SA_VIS_NONE = 0, SA_VIS_COSINE = 1, SA_VIS_EUCLIDEAN = 2 } sa_visual_kind;
/* VotingType  src/trackers/sort.rs:360-364 ; 0 = no winner (candidate starts a new track) */
typedef enum sa_voting_type { SA_VOTE_NONE = 0, SA_VOTE_VISUAL = 1, SA_VOTE_POSITIONAL = 2 } sa_voting_type;

/* Universal2DBox  src/utils/bbox.rs:78-87  (angle: Option<f32> -> has_angle + angle). */
typedef struct sa_box {
  float xc, yc;
  float angle;       /* radians; ignored when has_angle == 0                 */
  float aspect;      /* width / height, must be > 0                          */
  float height;      /* must be > 0                                          */
  float confidence;  /* must lie in [0, 1]                                   */
  int32_t has_angle; /* Option::is_some                                      */
  int32_t reserved;  /* must be 0                                            */
} sa_box;            /* 32 bytes */

/* Engine configuration = the option structs the reference's trackers are built from:
 * Sort::new args (sort/simple_api.rs:41-50), VisualSortOptions (visual_sort/options.rs:10-205),
 * VisualMetricBuilder defaults (visual_sort/metric/builder.rs:26-42). */
typedef struct sa_config {
  uint32_t struct_size;  /* = sizeof(sa_config), for ABI evolution */
  int32_t device;        /* HIP device ordinal, -1 = current device */
  void* stream;          /* hipStream_t to launch on; NULL = engine creates its own */

  int32_t positional_kind;          /* sa_positional_kind */
  float positional_threshold;       /* IoU(t); ignored for Mahalanobis (new-track threshold is 1.0, sort.rs:379) */
  float positional_min_confidence;  /* SortMetric min_confidence / VisualMetric positional_min_confidence */

  int32_t visual_kind;              /* sa_visual_kind; SA_VIS_NONE = plain SORT */
  float visual_threshold;           /* Cosine(t): keep d >= t ; Euclidean(t): keep d <= t */
  uint32_t feature_len;             /* D, floats per feature vector (un-padded) */
  uint32_t max_observations;        /* K = visual_max_observations (feature bank depth per track) */
  uint32_t visual_min_votes;
  uint32_t visual_minimal_track_length;
  float visual_minimal_area;
  float visual_minimal_quality_use;
  float visual_minimal_own_area_percentage_use;

  uint64_t max_idle_epochs;
  /* SpatioTemporalConstraints (spatio_temporal_constraints.rs:15-59): pairs (epoch_delta, max_dist). */
  uint32_t n_constraints;
  const uint64_t* constraint_epoch_delta;
  const float* constraint_max_dist;

  float kf_position_weight;         /* 1/20  by default (kalman_2d_box.rs:26)  */
  float kf_velocity_weight;         /* 1/160 by default                        */

  uint32_t flags;                   /* SA_FLAG_* */

  /* device-side track upkeep (sa_tracks_apply): the COLLECT gates of VisualMetric::optimize, visual_sort/metric.rs:337-349 */
  float visual_minimal_quality_collect;
  float visual_minimal_own_area_percentage_collect;

  /* tuning / test knobs (0 = the engine's own choice).  Per ENGINE: two engines of one process may differ. */
  int32_t gemm_plan;               /* n + 1 pins tile plan n of the contraction (sa_gemm.hip: 0 = 128x128, 5 = 64x128, 6 = 128x64, 1/2/4 = 64x64
                                      with 1/2/4 k-groups, 7/8 = the ring variants — all on the LDS-staged main loop; 9 = 64x64 on the k-split loop,
                                      15 / 16 = 128x128 / 64x128 on the direct loop, 18 = 64x128 on its k-split loop: 9 / 15 / 18 are what the engine
                                      runs by default, reading the track bank's fragment-order twin; 19 = plan 9 with the fused first phase's
                                      tiles 64x96 wherever that form applies — cosine frames of one observation per track — which the engine
                                      picks by itself for frames of 1.0 .. 1.5 rounds of 64x64 tiles) */
  uint32_t euclid_backoff_frames;  /* euclidean engines: after a frame that reported itself ill-conditioned for the matrix-core expansion, that
                                      SCENE's next frames run on the vector-pipe kernel, this many of them (0 = 256), before another try */
  int32_t poll_spin_us;            /* how long a host thread may poll a request set's completion words (mapped host memory, stored by the
                                      assignment tail: no completion signal on the dispatch) before it falls back to a BLOCKING wait on the
                                      engine's stream: 0 = the default (2000 us: a frame is there within tens of microseconds; a set
                                      queued behind somebody else's long kernel on a shared GPU blocks instead of burning a core),
                                      -1 = never poll (block at once: no host CPU while the GPU works, ~5 us later per frame) */
} sa_config;

#define SA_FLAG_PROFILE 0x2u        /* stamp every kernel with its dispatch begin / end (implies eager launches) */
/* First phase of a VisualSORT frame.  By default the engine puts the contraction's tiles, the positional tiles and the preparation
 * blocks into ONE heterogeneous launch whenever that applies (cosine, or euclidean through the matrix-core expansion; frames of any
 * size whose contraction runs as 64x64 tiles — up to two tiles per compute unit, or banks of 2..8 observations through whole-track
 * tiles; feature length a multiple of 32) — one dependent launch less per frame — and otherwise runs them as two launches. */
#define SA_FLAG_FUSED_FRAME 0x10u     /* ask for the heterogeneous launch explicitly (same as the default) */
#define SA_FLAG_SEPARATE_FRAME 0x20u  /* always two launches: the contraction runs as a kernel of its own (per-kernel measurements) */
#define SA_FLAG_GRAPH 0x8u          /* capture the per-frame launches into a hipGraph and replay it while the staged set is unchanged */
#define SA_FLAG_TAP 0x80u           /* parity tests: the assignment tail copies out what the frame's OWN launches produced — the BestFit vote
                                       words of the first phase and the edge counts of the positional tiles — before it consumes them, for
                                       sa_tap_votes / sa_tap_edges.  Same kernels, same launches; three more stores per candidate. */

/* Path switches (parity tests, measurements): each pins ONE decision the engine otherwise takes by frame size / metric.  They live in
 * the config, not in the environment: two engines of one process may run different paths. */
#define SA_FLAG_GENERAL_TAIL 0x100u     /* the many-workgroup assignment tail (k_assign_label + k_assign_solve) also on frames the one-workgroup tail would take */
#define SA_FLAG_NEVER_LEAN 0x200u       /* the frame-preparation blocks ride in every first phase (by default frames whose path does not read them leave them out) */
#define SA_FLAG_SEPARATE_RESOLVE 0x400u /* no vote words: per-tile partials + k_bestfit_resolve as a launch of its own */
#define SA_FLAG_EUCLID_VALU 0x800u      /* euclidean engines: always the vector-pipe kernel (direct sums of squares) */
#define SA_FLAG_EUCLID_MFMA 0x1000u     /* euclidean engines: always the matrix-core expansion + flagged recompute, also after an ill-conditioned frame */
#define SA_FLAG_XCD_TILES 0x4000u       /* the contraction's tiles in XCD-aware order (each XCD's L2 takes a compact block of tiles) ALSO where the contraction is
                                           a launch of its own — measured: c2b no faster, C5 4 % slower (the fused first phase uses that order by default:
                                           C2 20.0 -> 16.1 MB of L2 fills per launch at the same speed); A/B measurements */
#define SA_FLAG_ROW_TILES 0x8000u       /* the contraction's tiles row by row everywhere, the fused first phase included; A/B measurements */
#define SA_FLAG_SIGNAL_COMPLETION 0x10000u /* the host always waits for the completion SIGNAL of a frame's last dispatch; by default, where that dispatch is the
                                             one-workgroup tail, each scene's workgroup stores a completion WORD behind its results (mapped host memory) and
                                             the host polls it: a dispatch that carries a signal holds the next dispatch of its queue back by ~4.6 us */
#define SA_FLAG_STAGED_LOOP 0x20000u    /* the fused first phase's contraction tiles run the LDS-staged main loop over the row-major bank (the round-5 loop) instead of
                                           the k-split loop over the bank's fragment-order twin; A/B measurements and parity tests */
#define SA_FLAG_NO_YIELD 0x40000u       /* the fused first phase's matrix waves never nap between k-steps (by default one-observation cosine frames hand the positional
                                           tiles' waves 64 cycles of the vector port per k-step); A/B measurements */
#define SA_FLAG_BESTFIT_TILE 0x2000u    /* the weight matrix + k_bestfit_tile also where the contraction could vote itself (exact reference weights for deeper banks) */

/* Fill *cfg with the reference's defaults: IoU(0.3) (sort.rs:31), min confidence 0.05 (sort/metric.rs:11), no visual part,
 * one observation per track, max_idle_epochs 5, Kalman weights 1/20 and 1/160 (kalman_2d_box.rs:26), device -1. */
void sa_config_default(sa_config* cfg);

typedef struct sa_engine sa_engine;

int sa_engine_create(const sa_config* cfg, sa_engine** out);
void sa_engine_destroy(sa_engine* e);
const char* sa_last_error(const sa_engine* e); /* e may be NULL: last create() error of this thread */
uint32_t sa_api_version(void);

/* ---- track state (what TrackStore holds for this path) ------------------------------------
 * One row per stored track.  `boxes` = the track's last predicted box, i.e. the bbox of
 * observation[0] and predicted_boxes.back() (sort.rs:252-253).  `kf_mean`/`kf_cov` =
 * rows 0..5 of the Kalman state mean and the top-left 5x5 block of its covariance, row-major
 * (all `distance` reads through the identity update matrix, kalman_2d_box.rs:104-120,150-170);
 * may be NULL for IoU engines.  `feats` = n x K x D feature bank in observation order
 * (slot 0 = newest observation), `feat_present` = n x K flags; may be NULL for SORT engines.
 * Upserting an id that exists replaces its row in place; new ids are appended, so a caller that
 * hands out increasing ids (gen_track_id) keeps columns in ascending-id order. */
typedef struct sa_tracks {
  uint32_t n;
  const uint64_t* ids;        /* > 0 */
  const sa_box* boxes;
  const uint64_t* epochs;     /* last_updated_epoch */
  const float* kf_mean;       /* n x 5  or NULL */
  const float* kf_cov;        /* n x 25 or NULL */
  const float* feats;         /* n x K x D or NULL */
  const uint8_t* feat_present;/* n x K or NULL (NULL with feats != NULL means all present) */
} sa_tracks;

int sa_tracks_upsert(sa_engine* e, uint64_t scene_id, const sa_tracks* t);
int sa_tracks_remove(sa_engine* e, uint64_t scene_id, uint32_t n, const uint64_t* ids);
/* sa_tracks_remove on several scenes at once (a batch tracker taking expired tracks out of many of its scenes' tables in one predict()):
 * ids[i] lists counts[i] tracks of scene scene_ids[i]; one gather launch per dozen scenes; either every table changes or none. */
int sa_tracks_remove_many(sa_engine* e, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts, const uint64_t* const* ids);
/* The same in two steps, so that the per-scene host work can run on several threads: sa_tracks_remove_stage works out which rows of ONE
 * scene's table stay (different scenes may be staged by different threads at once; SA_ERR_STATE: this scene needs the serial call —
 * an upkeep step is still to be collected, or the engine would have to be drained first — and nothing has changed);
 * sa_tracks_remove_commit, on the calling thread, queues the gathers of every staged scene (sa_tracks_remove_many commits what is staged
 * along with its own scenes).  Nothing else may be called on the engine between a stage and its commit. */
int sa_tracks_remove_stage(sa_engine* e, uint64_t scene_id, uint32_t n, const uint64_t* ids);
int sa_tracks_remove_commit(sa_engine* e);
int sa_tracks_remove_abort(sa_engine* e);  /* forget every staged removal: no table changes (error paths between stage and commit) */
int sa_tracks_count(sa_engine* e, uint64_t scene_id, uint32_t* out_n);
/* Column order of the scene's track table (= column order of every matrix tap below). */
int sa_tracks_order(sa_engine* e, uint64_t scene_id, uint64_t* out_ids, uint32_t cap, uint32_t* out_n);

/* ---- one frame of detections of one scene ------------------------------------------------- */
typedef struct sa_detections {
  uint32_t n;
  const sa_box* boxes;
  const float* feats;          /* n x D or NULL (no features at all)                         */
  const uint8_t* feat_present; /* n or NULL (= all present when feats != NULL)                */
  const float* feat_quality;   /* n or NULL (= 1.0, visual_sort/simple_api.rs:146)            */
  const float* own_area;       /* n or NULL; NaN = None (own-area percentage of the detection) */
} sa_detections;

/* Association of one scene-frame: replaces foreign_track_distances + winners.
 * out_track_id[i] = id of the winning stored track, or 0 when candidate i starts a new track
 * (winner == self or no winner).  out_voting_type[i] = sa_voting_type.  Host buffers in, host
 * buffers out; the call stages, launches, synchronises and copies back. */
int sa_associate(sa_engine* e, uint64_t scene_id, uint64_t epoch, const sa_detections* d,
                 uint64_t* out_track_id, uint8_t* out_voting_type);

/* Batched scenes (BatchSort / BatchVisualSort): all scenes of one request go through ONE set
 * of kernel launches (grid.z = scene).  Split form, so a caller can keep inputs resident and
 * time only the device work:
 *   sa_batch_begin  — waits for the previous batch (the reference's "busy monitor",
 *                     sort/batch_api.rs:233-241), clears the staged list
 *   sa_batch_add    — stage one scene's detections in the pinned staging arena; returns its slot in *out_slot
 *   sa_batch_run    — one H2D copy of the staged set (first run only), then the whole pipeline on the engine's stream; does not
 *                     synchronise
 *   sa_batch_sync   — wait for the stream
 *   sa_batch_fetch  — copy one scene's result back (synchronises if needed)
 * sa_batch_run may be called repeatedly on the same staged inputs (benchmark loop). */
int sa_batch_begin(sa_engine* e);
int sa_batch_add(sa_engine* e, uint64_t scene_id, uint64_t epoch, const sa_detections* d, uint32_t* out_slot);
/* sa_batch_add with the feature rows given one pointer per detection (feat_rows[i] == NULL: detection i has no feature; d->feats is
 * ignored) — the shape the reference hands them over in, VisualSortObservation.feature: Option<&[f32]>
 * (visual_sort/simple_api.rs:130-170): the rows go straight into the pinned staging block, no N x D assembly on the caller's side. */
int sa_batch_add_rows(sa_engine* e, uint64_t scene_id, uint64_t epoch, const sa_detections* d, const float* const* feat_rows,
                      uint32_t* out_slot);
/* Staging a request set of many scenes on several threads (Batch*::predict): sa_batch_add_deferred lays the scene out in the arena and
 * remembers the caller's arrays (which must stay valid until the fill); sa_batch_fill(slot) validates the boxes and copies them in.
 * Fills of DIFFERENT slots may run on different threads at once (never beside an add); every slot must be filled before sa_batch_run*. */
int sa_batch_add_deferred(sa_engine* e, uint64_t scene_id, uint64_t epoch, const sa_detections* d, const float* const* feat_rows,
                          uint32_t* out_slot);
int sa_batch_fill(sa_engine* e, uint32_t slot);
int sa_batch_run(sa_engine* e);
int sa_batch_sync(sa_engine* e);
int sa_batch_fetch(sa_engine* e, uint32_t slot, uint64_t* out_track_id, uint8_t* out_voting_type);
/* The same winners as COLUMNS of the scene's track table (the order of sa_tracks_order), -1 = none: a host that keeps its tracks in
 * table order — rows are appended in upsert / apply order, sa_tracks_remove closes the gaps — finds the winner without a lookup by id. */
int sa_batch_fetch_cols(sa_engine* e, uint32_t slot, int32_t* out_cols);
/* sa_batch_fetch + sa_batch_fetch_cols without the copies: pointers to the slot's results where the assignment tail wrote them (pinned
 * host memory), valid until the next sa_batch_begin.  Any output may be NULL.  Waits like sa_batch_fetch; after sa_batch_run_apply it may
 * be called for different slots from different threads at once. */
int sa_batch_results(sa_engine* e, uint32_t slot, const uint64_t** out_track_id, const uint8_t** out_voting_type, const int32_t** out_cols);

/* ---- device-side track upkeep: the step either side of the association ------------
 * Applies the result of the last run of batch slot `slot` to that scene's device-resident track table, the way
 * Sort / VisualSort::predict do after voting (sort/simple_api.rs:164-190, visual_sort/simple_api.rs:189-226 ->
 * Track::merge -> SortMetric / VisualMetric::optimize), without a host round trip of boxes, Kalman state or features:
 *   candidate i won track w    -> w takes one Kalman predict + update with the candidate's box (make_prediction,
 *                                 kalman_prediction.rs:13-32), its epoch becomes the frame's, its feature bank follows
 *                                 optimize_observations (visual_sort/metric.rs:129-154; COLLECT gates from sa_config)
 *   candidate i has no winner  -> a new track with id new_ids[i] is appended: initiate -> predict -> update, bank = [its
 *                                 observation]
 * out_predicted[i] = the box of the destination track after the step (SortTrack::predicted_bbox).  new_ids[i] must be > 0
 * and unused in the scene where candidate i has no winner (ignored elsewhere).  Tracks must have been created by this call
 * (or given a full state with sa_tracks_set_state): a track upserted with the 5 x 5 projection only cannot be stepped. */
int sa_tracks_apply(sa_engine* e, uint32_t slot, const uint64_t* new_ids, sa_box* out_predicted);
/* The same in two halves, for a caller with host work of its own between them (the tracker facade does its per-track bookkeeping
 * there — what the reference does around Track::merge): _begin validates, appends the new rows and QUEUES the device-side step;
 * _end waits for it and hands out the predicted boxes.  The candidates' feature rows must stay untouched until _end has returned.
 * Several slots may be between the two halves at once; an entry point that needs the finished table first (the next request set, an
 * upsert / remove / set_state, a table tap) finishes what is pending — _end then only copies the boxes. */
int sa_tracks_apply_begin(sa_engine* e, uint32_t slot, const uint64_t* new_ids);
int sa_tracks_apply_end(sa_engine* e, uint32_t slot, sa_box* out_predicted);
/* The same upkeep QUEUED BEHIND the association on the device, no host round trip in between (the tracker facade's path):
 * sa_batch_run_apply = sa_batch_run of the staged scenes + for every one of them the step sa_tracks_apply would take, with the ids of
 * the tracks that START drawn on the device the way the reference draws them — from a counter, in candidate order (sort/simple_api.rs:
 * 165-187): candidate i without a winner gets id_base[slot] + 1 + r, r = its rank among the slot's new tracks (id_per_candidate != 0:
 * id_base[slot] + 1 + i — Batch* trackers draw one id per candidate, batch_api.rs:102-106).  The caller waits ONCE (sa_batch_sync /
 * sa_batch_fetch), then sa_tracks_apply_collect(slot) updates the host side of the table and hands out, per candidate, the id it
 * started a track with (0 where it merged) and the destination track's predicted box.  Either output may be NULL.  A pending slot is
 * collected by any entry point that needs the finished table.  Synchronous batches only.
 * sa_tracks_apply_collect returns as soon as the boxes and the table's rows are out: a VisualSORT engine may still be moving feature
 * rows inside its bank (kernels queued behind the Kalman step) — every later call on the engine is ordered behind them; the candidates'
 * feature rows (device blocks, pinned blocks) must stay untouched until the next call on the engine has returned. */
int sa_batch_run_apply(sa_engine* e, const uint64_t* id_base /* one per staged scene */, int id_per_candidate);
int sa_tracks_apply_collect(sa_engine* e, uint32_t slot, uint64_t* out_new_ids, sa_box* out_predicted);
/* The same for a whole request set in three steps, so that the per-scene host work can be spread over threads (Batch*::predict over
 * dozens of scenes): _begin waits ONCE for the set's upkeep (one Kalman dispatch for every scene of the set); _slot does one scene's host
 * side and may run for DIFFERENT slots on different threads at once; _end, on the calling thread again, queues what is left (the
 * polygons of refreshed oriented rows). */
/* The host side of one scene's table for the queued upkeep step (the ids of the tracks that start; the winners checked) needs the
 * association's results only: with those in (sa_batch_results) it may be done per slot, on any thread, WHILE the Kalman dispatch runs —
 * sa_tracks_apply_collect_slot then only hands out the predicted boxes.  Optional: the _slot call does it when nobody has. */
int sa_tracks_apply_collect_table(sa_engine* e, uint32_t slot, uint64_t* out_new_ids);
int sa_tracks_apply_collect_begin(sa_engine* e);
int sa_tracks_apply_collect_slot(sa_engine* e, uint32_t slot, uint64_t* out_new_ids, sa_box* out_predicted);
int sa_tracks_apply_collect_end(sa_engine* e);
/* Full per-track state for the device-side upkeep (debug / parity / seeding): Kalman mean[10] + cov[100] row-major, and per
 * bank slot the feature quality[K].  Any of the output pointers may be NULL. */
int sa_tracks_get_state(sa_engine* e, uint64_t scene_id, uint64_t id, float* mean10, float* cov100, float* quality,
                        uint8_t* present, float* feats /* K x D */);
int sa_tracks_set_state(sa_engine* e, uint64_t scene_id, uint64_t id, const float* mean10, const float* cov100,
                        const float* quality /* K or NULL */);

/* ---- non-maximum suppression (src/utils/nms.rs:32-72) --------------------------------
 * nms(detections, nms_threshold, score_threshold): boxes with score <= score_threshold (or height / aspect <= 0) are dropped,
 * the rest are visited in descending rank (score, or the box height where the score is None; stable) and a visited box
 * suppresses every later one with  intersection(visited, later) as f32 / area(later) > nms_threshold.
 * scores: NULL or NaN entries = None.  score_threshold: NaN = None (f32::MIN).  out_keep[*out_n] = indices into `boxes` of the
 * survivors in the reference's output order; capacity n.  The pair tests run on the GPU (same f64 clip as the IoU cells). */
int sa_nms(sa_engine* e, uint32_t n, const sa_box* boxes, const float* scores, float nms_threshold, float score_threshold,
           uint32_t* out_keep, uint32_t* out_n);

/* ---- exclusively owned areas (src/utils/clipping/bbox_own_areas.rs:8-46) --------------
 * exclusively_owned_areas_normalized_shares(boxes, exclusively_owned_areas(boxes)): out_share[i] = area of box i not covered by
 * any other box of the frame / (area_i + EPS), clamped to 1.0 — the value VisualSORT's own-area gates read
 * (visual_sort/simple_api.rs:111-127); feed it to sa_detections.own_area.  The reference builds the difference polygons with
 * geo's BooleanOps; the device integrates the boundary of the same region in f64 (agreement to the reference test's EPS = 1e-5).
 * A box in the middle of a crowd (more than 127 overlapping neighbours) takes a slower path with its lists in HBM; the only limit left
 * is 512 DISJOINT covered stretches on one edge of one box (SA_ERR_UNSUPPORTED). */
int sa_own_areas(sa_engine* e, uint32_t n, const sa_box* boxes, float* out_share);

typedef struct sa_scene_request {
  uint64_t scene_id;
  uint64_t epoch;
  sa_detections detections;
} sa_scene_request;
typedef struct sa_scene_result {
  uint64_t* out_track_id;   /* detections.n entries */
  uint8_t* out_voting_type; /* detections.n entries */
} sa_scene_result;
int sa_associate_batch(sa_engine* e, uint32_t n_scenes, const sa_scene_request* req, const sa_scene_result* res);

/* ---- pipelined request sets: H2D of frame n+1 beside the kernels of frame n ------------------------------------
 * The reference's predict() takes its detections (boxes + 512..4096-d features) from host memory every frame
 * (visual_sort/simple_api.rs:130-170).  sa_associate* does stage -> DMA -> kernels -> results one after the other; the calls
 * below keep up to THREE request sets in flight on two streams, so that the DMA of the next set (2 MB at 1000 x 512-d) runs beside
 * the kernels of the current one — and the one after is already queued — and a stream of frames costs max(DMA, kernels) instead of
 * their sum:
 *   sa_pipe_stage   lays the request set out in a pinned staging arena (features inside a sa_host_alloc block are not copied:
 *                   the DMA reads them in place) and queues ONE host-to-device copy on the copy stream; returns a ticket
 *   sa_pipe_launch  queues the kernels behind that copy on the compute stream; the track tables are read as they are at THIS
 *                   call, so sa_tracks_upsert / sa_tracks_apply of the previous frame may sit between stage and launch
 *   sa_pipe_submit  = stage + launch
 *   sa_pipe_wait    blocks until the ticket's kernels have retired and copies its results out (res[i] belongs to req[i]);
 *                   afterwards slot numbers of sa_tracks_apply / sa_batch_fetch / the taps refer to this ticket's scenes
 * At most three tickets are outstanding (SA_ERR_STATE otherwise).  Buffers handed to sa_pipe_stage may be reused as soon as it
 * returns, except feature blocks from sa_host_alloc, which must stay untouched until the ticket has been waited for.
 * A tracker loop with device-side upkeep:  stage(n+1); wait(n); apply(n); launch(n+1). */
int sa_pipe_stage(sa_engine* e, uint32_t n_scenes, const sa_scene_request* req, uint64_t* out_ticket);
int sa_pipe_launch(sa_engine* e, uint64_t ticket);
int sa_pipe_submit(sa_engine* e, uint32_t n_scenes, const sa_scene_request* req, uint64_t* out_ticket);
int sa_pipe_wait(sa_engine* e, uint64_t ticket, const sa_scene_result* res);

/* ---- one process, several GPUs: scenes sharded over one engine per device -------------------------------------
 * BatchSort / BatchVisualSort fan the scenes of a request out to voting threads of one process (sort/batch_api.rs:197-207,
 * 278-288; visual_sort/batch_api.rs:296-315); scenes never interact (compatible() is false across scene ids, sort.rs:251).
 * A cluster owns one engine per device and one worker thread per engine and routes by  scene_id % n_shards  (sticky: a scene's
 * track table stays resident on its GPU).  sa_cluster_associate_batch = scatter the request set by shard, every shard runs ONE
 * sa_associate_batch on its GPU concurrently with the others, gather — res[i] belongs to req[i].  No collective, no copy
 * between GPUs.  devices: n_shards HIP ordinals, or NULL for 0..n_shards-1 (an ordinal may repeat: several shards on one GPU).
 * sa_cluster_last_ms(shard): wall time the shard's worker spent inside its last call (for load-balance / scaling reports).
 * sa_cluster_engine(shard): the shard's engine, for the per-engine entry points (taps, sa_tracks_apply, ...) — not while a
 * cluster call is running. */
typedef struct sa_cluster sa_cluster;
int sa_cluster_create(const sa_config* cfg, uint32_t n_shards, const int32_t* devices, sa_cluster** out);
void sa_cluster_destroy(sa_cluster* c);
const char* sa_cluster_last_error(const sa_cluster* c); /* c may be NULL: last create() error of this thread */
uint32_t sa_cluster_size(const sa_cluster* c);
uint32_t sa_cluster_shard_of(const sa_cluster* c, uint64_t scene_id);
sa_engine* sa_cluster_engine(sa_cluster* c, uint32_t shard);
int sa_cluster_tracks_upsert(sa_cluster* c, uint64_t scene_id, const sa_tracks* t);
int sa_cluster_tracks_remove(sa_cluster* c, uint64_t scene_id, uint32_t n, const uint64_t* ids);
int sa_cluster_associate_batch(sa_cluster* c, uint32_t n_scenes, const sa_scene_request* req, const sa_scene_result* res);
double sa_cluster_last_ms(const sa_cluster* c, uint32_t shard);

/* ---- parity taps (debug / test): the matrices of the last run of batch slot `slot` --------
 * Row = candidate in input order, column = track in sa_tracks_order() order.
 *   positional: N x T f32, NaN = absent (SortMetric::metric / VisualMetric::positional_metric)
 *   visual    : N x T x K f32, NaN = absent (VisualMetric::visual_metric after distance_to_weight)
 *   quantised : N x T i64 = (w * 1e6f) as i64 of the positional value, 0 where absent
 *               (SortVoting::winners sort/voting.rs:59; the diagonal/self columns are implicit) */
int sa_tap_dims(sa_engine* e, uint32_t slot, uint32_t* n, uint32_t* t, uint32_t* k);
int sa_tap_positional(sa_engine* e, uint32_t slot, float* out);
int sa_tap_visual(sa_engine* e, uint32_t slot, float* out);
int sa_tap_quantised(sa_engine* e, uint32_t slot, int64_t* out);
/* The polygons the IoU cells clip against: 4 f64 vertices (x, y) per row of the scene's track table, in sa_tracks_order order
 * (Polygon::from(&Universal2DBox), bbox.rs:287-330).  *out_rows = rows of the table; written when cap_rows >= that. */
int sa_tap_track_polygons(sa_engine* e, uint64_t scene_id, double* out, uint32_t cap_rows, uint32_t* out_rows);
/* What the frame's own (timed) launches produced, as opposed to the matrices above, which the taps recompute on demand.  Engines created
 * with SA_FLAG_TAP only (SA_ERR_STATE otherwise).
 *   sa_tap_votes: the BestFit vote (track/voting/best.rs:52-128) as the first phase reduced it — per candidate its best track
 *     (row_idx, -1 = the row has no group at all) and per track its best candidate (col_idx), with the weight that won:
 *       *kind == 1  bank depth 1: the LIGHTEST visual weight of the row / column (the heaviest group is the lightest weight,
 *                   VisualMetric::visual_metric after distance_to_weight, visual_sort/metric.rs:200-225), exactly as the kernel formed it (f32)
 *       *kind == 2  deeper banks: the HEAVIEST group weight W = sum_k f64(max_dist - w_k) of the row / column, truncated to the 54 leading
 *                   bits the vote word carries (relative error < 2^-43)
 *     Frames up to 1024 x 1024 decode the vote words, larger ones fold the per-tile partials the resolve kernel reads.
 *   sa_tap_edges: the input of the positional vote (SortVoting::winners, sort/voting.rs:30-100) as the positional tiles emitted it:
 *     counts[i] edges for candidate i, then cols / gains in CSR order (row after row, arbitrary order inside a row), gain = quantised
 *     weight - quantised new-track threshold > 0.  cap = capacity of cols / gains in edges; *out_total = edges in all (call with
 *     cap = 0 to size the arrays). */
int sa_tap_votes(sa_engine* e, uint32_t slot, double* row_w, int32_t* row_idx, double* col_w, int32_t* col_idx, int32_t* kind);
int sa_tap_edges(sa_engine* e, uint32_t slot, uint32_t* counts, uint32_t cap, uint32_t* cols, int64_t* gains, uint32_t* out_total);

/* ---- pinned host blocks (optional) --------------------------------------------------------
 * sa_detections.feats (N x D f32: 2 MB at 1000 x 512) is normally copied into the engine's own pinned staging buffer before it
 * goes to the device.  A block obtained from sa_host_alloc is pinned already: sa_batch_add / sa_associate* recognise a feats
 * pointer inside one and the DMA reads it in place.  The caller must leave the block untouched until that frame's results have
 * been fetched (sa_associate* return after that point; with sa_batch_add: until sa_batch_sync).  A reference-side binding keeps
 * one block per tracker and writes the candidates' features straight into it — in place of the per-frame Vec<f32> -> Feature
 * copies of sort-family predict (visual_sort/simple_api.rs:156-158).  Process-wide, thread-safe, independent of any engine. */
void* sa_host_alloc(uint64_t bytes); /* NULL: no device, or out of memory */
void sa_host_free(void* block);      /* a pointer returned by sa_host_alloc, or NULL */

/* ---- detection features that are already in device memory (optional) ----------------------
 * In the reference every observation carries its feature as a host slice (VisualSortObservation.feature: Option<&[f32]>,
 * visual_sort/simple_api.rs:130-170, visual_sort.rs:34-56) — whatever produced it.  When the producer is a ReID model on the SAME
 * GPU, the rows are in HBM already and the round trip device -> host -> device (2 MB per frame at 1000 x 512-d, the largest part of
 * an ingested frame's cost) is pure overhead: register the producer's output buffer once, then pass pointers INTO it as
 * sa_detections.feats (N x D f32, row-major, as on the host).  sa_batch_add / sa_associate* / sa_pipe_stage recognise such a pointer
 * and the kernels read the rows where they lie (a base that is not 16-byte aligned costs one device-to-device copy); boxes,
 * qualities and the other per-detection arrays still come from host memory.  The rows must be FINAL when the call is made
 * (synchronise the producing stream first: the engine's streams do not know about it) and must stay untouched until that frame's
 * results have been fetched — and, when the engine also maintains the feature banks (sa_tracks_apply), until that call has returned:
 * it takes the winners' rows from the same place.  A producer that writes frame n+1 while frame n is in flight alternates two regions.  `device` = the HIP device the block lives on (< 0: the calling thread's current device); a request that reaches an engine on another device is
 * refused (SA_ERR_BAD_ARG).  Process-wide, thread-safe, independent of any engine; registering a base pointer again updates its size. */
int sa_device_block_register(const void* dev_ptr, uint64_t bytes, int device);
void sa_device_block_unregister(const void* dev_ptr);

/* ---- measurement -------------------------------------------------------------------------- */
typedef struct sa_kernel_stat {
  char name[48];
  uint64_t launches;
  double total_ms;   /* sum of hipEvent-bracketed durations on the engine's stream */
} sa_kernel_stat;
/* Per-kernel dispatch timing on an engine that was not created with SA_FLAG_PROFILE (e.g. the one inside a tracker facade,
 * sa_tracker_engine): on != 0 stamps every following kernel with its dispatch begin / end (and disables hipGraph replay), 0 stops. */
int sa_profile_enable(sa_engine* e, int on);
int sa_profile_reset(sa_engine* e);
int sa_profile_read(sa_engine* e, sa_kernel_stat* out, uint32_t cap, uint32_t* out_n);
/* hipEvent-timed wall time of `iters` back-to-back sa_batch_run()s on the engine's stream. */
int sa_batch_time(sa_engine* e, uint32_t iters, double* out_ms_total);

/* Standalone cost-matrix entry point (config C5 / MFMA roofline): out[n x t] f32 of cosine (kind 1) or euclidean (kind 2)
 * distance (src/distance.rs:9-47) between the rows of a[n x d] and b[t x d], both HOST pointers.  The call copies them to the
 * device, runs the contraction kernel once untimed and `iters` times between two hipEvents (the kernel alone), and copies the
 * matrix back when out != NULL. */
int sa_feature_distance_matrix(sa_engine* e, int32_t visual_kind, uint32_t n, uint32_t t, uint32_t d,
                               const float* a, const float* b, float* out, uint32_t iters, double* out_ms_total);

#ifdef __cplusplus
}
#endif
#endif /* SIMILARI_ASSOC_H */
