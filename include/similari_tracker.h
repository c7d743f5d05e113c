/*
 * similari_tracker.h — C ABI of the tracker facade that sits on top of the association engine.
 *
 * It preserves the reference's public tracker surface for the accelerated path, so that examples, benches and
 * tests written against it translate 1:1:
 *
 *   Sort::new / predict / predict_with_scene / idle_tracks_with_scene        src/trackers/sort/simple_api.rs:41-215
 *   BatchSort::new / predict(PredictionBatchRequest)                         src/trackers/sort/batch_api.rs:157-290
 *   VisualSort::new / predict / predict_with_scene                           src/trackers/visual_sort/simple_api.rs:45-230
 *   BatchVisualSort::new / predict                                           src/trackers/visual_sort/batch_api.rs:161-317
 *   TrackerAPI::{skip_epochs_for_scene, current_epoch_with_scene, wasted, clear_wasted, active_shard_stats}
 *                                                                            src/trackers/tracker_api.rs:9-118
 *
 * What runs where: the per-frame N x T cost matrices and the assignment run on the GPU through
 * similari_assoc.h (sa_associate_batch).  The O(N) upkeep either side of it — Kalman predict/update of the merged
 * tracks (kalman_prediction.rs:13-32) and the feature-bank policy (visual_sort/metric.rs:129-154, 297-374) — runs
 * either on the host exactly as the reference keeps it (device_upkeep = 0: the refreshed rows are uploaded with
 * sa_tracks_upsert every frame) or on the GPU (device_upkeep = 1: sa_tracks_apply, nothing but the predicted boxes
 * comes back).  Epoch counters, history deques, track ids and the wasted-track lifecycle stay on the host.
 *
 * exclusively_owned_areas (clipping/bbox_own_areas.rs:8-46): when either own-area threshold is > 0 the facade computes the
 * shares of the frame's boxes on the GPU (sa_own_areas) exactly where the reference computes them
 * (visual_sort/simple_api.rs:111-127); a caller may override a detection's share through sa_observation.own_area.
 */
#ifndef SIMILARI_TRACKER_H
#define SIMILARI_TRACKER_H

#include "similari_assoc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* SortTrack  src/trackers/sort.rs:286-311 */
typedef struct sa_sort_track {
  uint64_t id;
  uint64_t epoch;
  sa_box predicted_bbox;
  sa_box observed_bbox;
  uint64_t scene_id;
  uint64_t length;
  int32_t voting_type;           /* sa_voting_type: SA_VOTE_VISUAL / SA_VOTE_POSITIONAL */
  int32_t has_custom_object_id;  /* Option<i64>::is_some */
  int64_t custom_object_id;
} sa_sort_track;

/* VisualSortObservation  src/trackers/visual_sort.rs:34-56  (plain SORT uses bbox + custom id only) */
typedef struct sa_observation {
  sa_box bbox;
  const float* feature;          /* feature_len floats or NULL (Option<&[f32]>) */
  float feature_quality;         /* NaN = None -> 1.0 (visual_sort/simple_api.rs:146) */
  float own_area;                /* NaN = computed by the tracker when the own-area gates are armed, else None */
  int32_t has_custom_object_id;
  int32_t reserved;
  int64_t custom_object_id;
} sa_observation;

/* Sort::new arguments / VisualSortOptions (+ VisualMetricBuilder) in one POD. */
typedef struct sa_tracker_options {
  uint32_t struct_size;
  int32_t device;                       /* HIP device ordinal, -1 = current */
  int32_t visual;                       /* 0 = Sort / BatchSort, 1 = VisualSort / BatchVisualSort */
  int32_t batch_ids;                    /* 1 = Batch* id policy: a track id is drawn per candidate even when it merges
                                           (sort/batch_api.rs:102-106) */
  uint32_t history_length;              /* bbox_history / kept_history_length, must be > 0 */
  uint32_t auto_waste_periodicity;      /* DEFAULT_AUTO_WASTE_PERIODICITY = 100 (sort.rs:378) */
  uint64_t max_idle_epochs;
  int32_t positional_kind;              /* sa_positional_kind */
  float positional_threshold;           /* IoU(t) */
  float positional_min_confidence;      /* Sort: min_confidence ; VisualSort: positional_min_confidence (0.1) */
  uint32_t n_constraints;
  const uint64_t* constraint_epoch_delta;
  const float* constraint_max_dist;
  float kalman_position_weight;
  float kalman_velocity_weight;
  /* VisualSort only */
  int32_t visual_kind;                  /* sa_visual_kind */
  float visual_threshold;
  uint32_t feature_len;
  uint32_t visual_max_observations;
  uint32_t visual_min_votes;
  uint32_t visual_minimal_track_length;
  float visual_minimal_area;
  float visual_minimal_quality_use;
  float visual_minimal_quality_collect;
  float visual_minimal_own_area_percentage_use;
  float visual_minimal_own_area_percentage_collect;
  int32_t device_upkeep;                /* 1 = Kalman step, table refresh and feature-bank policy run on the GPU (sa_tracks_apply):
                                           no per-frame upload of boxes / Kalman state / features; 0 = host upkeep + sa_tracks_upsert */
  int32_t workers;                      /* threads working on the scenes of one request set (Batch*: the reference's voting_shards,
                                           sort/batch_api.rs:197-207), the calling thread included; 0 = the facade's choice, 1 = none.
                                           n > 1: bound to the CPUs next to the one the first batch call runs on that no other tracker of
                                           the process holds (a scene's records stay in one cache complex); -n: n threads left to the scheduler */
  /* Batch*::predict over the GPUs of one node (sort/batch_api.rs:157-207, visual_sort/batch_api.rs:161-317: ONE tracker object fans a
   * request set out).  n_devices > 1: the tracker owns one engine per entry of devices[] (HIP ordinals; the same ordinal may appear more
   * than once — several engines on one GPU) and every scene lives on ONE of them for good: shard = scene_id % n_devices.  A request
   * set is split by shard, every shard's share runs on its device through the same fused path as a single-device tracker (association
   * + device upkeep queued behind it, that shard's own worker threads and result driver), all shards at once, and the result handle
   * delivers the scenes of every shard as they finish.  Track ids, epochs, idle / wasted sets and the auto-waste cadence are those of ONE
   * tracker (the id counter and the waste counter are the group's; Batch* ids are a function of the request alone, so no shard waits for
   * another).  `workers` is per shard; workers = 0 shares ONE tracker's default thread budget out among the shards.  Feature rows handed over in a registered DEVICE block (sa_device_block_register) must lie on the device of their scene's shard
   * (devices[scene_id % n_devices]): a row on another GPU is refused, not fetched.  n_devices <= 1: one engine on `device` (devices is ignored). */
  uint32_t n_devices;
  const int32_t* devices;
  /* Host manners.  spin_us: how long a thread of this tracker may busy-poll — the result driver for the next request set, the worker
   * pool for its next job, a caller inside sa_batch_result_get for the next scene — before it goes to sleep on a condition variable
   * (-1 = the defaults: 300 us pool, 200 us driver, 500 us handle; 0 = sleep at once: no idle CPU use, +5-20 us per hand-over). */
  int32_t spin_us;
} sa_tracker_options;

typedef struct sa_tracker sa_tracker;

/* Defaults of Sort (visual = 0: IoU(0.3), min_confidence 0.05) or of VisualSortOptions/VisualMetricBuilder
 * (visual = 1: Euclidean(f32::MAX), IoU(0.3), minimal track length 3, 5 observations, 1 vote, max_idle_epochs 2,
 * history 10, positional_min_confidence 0.1). */
void sa_tracker_options_default(sa_tracker_options* o, int visual);
int sa_tracker_create(const sa_tracker_options* o, sa_tracker** out);
void sa_tracker_destroy(sa_tracker* t);
const char* sa_tracker_last_error(const sa_tracker* t);

/* predict_with_scene: out[n] in candidate order. */
int sa_tracker_predict(sa_tracker* t, uint64_t scene_id, uint32_t n, const sa_observation* obs, sa_sort_track* out);
/* Batch*::predict: scenes are processed in request order; every scene goes through ONE set of kernel launches. */
int sa_tracker_predict_batch(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts,
                             const sa_observation* const* obs, sa_sort_track* const* out);

/* Batch*::predict as the reference shapes it (sort/batch_api.rs:222-290, visual_sort/batch_api.rs:213-317): the call returns once the
 * request set is on the device, with a handle that delivers the scenes' tracks as they become final — PredictionBatchResult,
 * trackers/batch.rs:19-38:
 *   sa_batch_result_size   batch_size(): scenes in the request set
 *   sa_batch_result_ready  ready(): 1 when sa_batch_result_get would not block
 *   sa_batch_result_take   get() as the reference has it: the next finished scene's tracks handed over in place (no copy)
 *   sa_batch_result_get    get(): the next finished scene — its id, its tracks in candidate order (cap < *out_n: SA_ERR_BAD_ARG, nothing is
 *                          taken, *out_n says how many there are); blocks until one is there; SA_ERR_STATE once every scene was taken
 * The request is taken by value like the reference's: the observation arrays (and feature rows in host memory) may be reused as soon as
 * _begin has returned; feature rows in a registered DEVICE block must stay untouched until the last scene has been delivered.  One
 * request set is in flight at a time: every other call on the tracker first waits for it (the reference's busy monitor,
 * sort/batch_api.rs:233-241).  Trackers without device upkeep do all the work inside _begin (every scene is ready when it returns).
 * The handle is the caller's: sa_batch_result_free (at any time). */
typedef struct sa_batch_result sa_batch_result;
int sa_tracker_predict_batch_begin(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts,
                                   const sa_observation* const* obs, sa_batch_result** out_result);
uint32_t sa_batch_result_size(const sa_batch_result* r);
int sa_batch_result_ready(sa_batch_result* r);
int sa_batch_result_get(sa_batch_result* r, uint64_t* out_scene_id, sa_sort_track* out, uint32_t cap, uint32_t* out_n);
/* get() without the copy (the reference's get() MOVES a scene's Vec<SortTrack> out of the channel): *out_tracks points at the scene's
 * tracks inside the handle, valid until sa_batch_result_free.  (64 scenes x 500 tracks are 2.3 MB: copied out one scene at a time they
 * cost a C++ host 100 us of a 450 us call, scripts/micro/batch_handle_bench.cpp.) */
int sa_batch_result_take(sa_batch_result* r, uint64_t* out_scene_id, const sa_sort_track** out_tracks, uint32_t* out_n);
void sa_batch_result_free(sa_batch_result* r);

int sa_tracker_idle_tracks(sa_tracker* t, uint64_t scene_id, sa_sort_track* out, uint32_t cap, uint32_t* out_n);
int sa_tracker_skip_epochs(sa_tracker* t, uint64_t scene_id, uint64_t n);
int sa_tracker_current_epoch(sa_tracker* t, uint64_t scene_id, uint64_t* out);
/* wasted(): runs auto_waste, then drains the wasted store into out (cap entries at most; *out_n = total). */
int sa_tracker_wasted(sa_tracker* t, sa_sort_track* out, uint32_t cap, uint32_t* out_n);
int sa_tracker_clear_wasted(sa_tracker* t);
int sa_tracker_active_tracks(sa_tracker* t, uint64_t* out_n); /* sum of active_shard_stats() */
/* Kalman state of a stored track (debug / parity): mean[10], cov[100] row-major. */
int sa_tracker_track_state(sa_tracker* t, uint64_t track_id, float* mean10, float* cov100);
/* {visual_features_collected_count, stored observations, observed_boxes.len(), track_length} of a stored track. */
int sa_tracker_track_info(sa_tracker* t, uint64_t track_id, uint64_t out4[4]);
sa_engine* sa_tracker_engine(sa_tracker* t);

#ifdef __cplusplus
}
#endif
#endif /* SIMILARI_TRACKER_H */
