// oracle_tracker.cpp — CPU restatement of the reference's tracker loops around the hot path.  TEST INFRASTRUCTURE
// ONLY (see oracle.h): it exists to check the product's facade (similari_amd/csrc/sa_tracker.cpp) frame by frame.
//
// Follows, statement by statement:
//   Sort::predict_with_scene            src/trackers/sort/simple_api.rs:110-196
//   VisualSort::predict_with_scene      src/trackers/visual_sort/simple_api.rs:99-230
//   BatchSort / BatchVisualSort voting  src/trackers/sort/batch_api.rs:68-153, visual_sort/batch_api.rs:77-158
//   Track::add_observation / merge      src/track.rs:447-588
//   SortMetric::optimize                src/trackers/sort/metric.rs:79-105
//   VisualMetric::optimize(+_observations)  src/trackers/visual_sort/metric.rs:129-154, 297-374
//   TrackerAPI / EpochDb                src/trackers/tracker_api.rs, src/trackers/epoch_db.rs
// Every numeric step goes through the or_* functions of oracle.cpp (dense nalgebra-order Kalman filter, per-pair
// metrics, dense kuhn_munkres).  Canonical orders: scenes in request order, candidates in input order, stored
// tracks by ascending id.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <map>
#include <vector>

#include "../include/similari_tracker.h"
#include "oracle.h"

namespace {

struct Observation {  // Observation<VisualObservationAttributes>: (attrs: quality, bbox?, own_area?), feature?
  bool has_bbox = false;
  sa_box bbox{};
  float quality = 1.0f;
  bool has_own = false;
  float own = 0.0f;
  bool has_feature = false;
  std::vector<float> feature;
};

struct TrackO {
  uint64_t id = 0;
  // SortAttributes / VisualAttributes
  std::deque<sa_box> predicted_boxes, observed_boxes;
  uint64_t last_updated_epoch = 0, track_length = 0, scene_id = 0;
  size_t visual_features_collected_count = 0;
  bool has_custom = false;
  int64_t custom = 0;
  int voting_type = -1;  // Option<VotingType>
  bool has_state = false;
  float mean[10]{}, cov[100]{};
  std::vector<Observation> observations;  // feature class 0
};

}  // namespace

struct or_tracker {
  sa_tracker_options o{};
  std::vector<uint64_t> cd;
  std::vector<float> cm;
  sa_config cfg{};
  std::map<uint64_t, TrackO> store, wasted;
  std::map<uint64_t, uint64_t> epoch_db;
  uint64_t track_id = 0;
  size_t aw_counter = 0;
};

namespace {

bool can_use(const or_tracker* t, const sa_box& b, float q, float min_q, bool has_own, float own, float min_own) {
  bool quality_is_ok = q >= min_q;
  bool percentage_is_ok = has_own ? own >= min_own : true;
  bool bbox_is_ok = or_area(&b) >= t->o.visual_minimal_area;
  return bbox_is_ok && quality_is_ok && percentage_is_ok;
}

void update_history(const or_tracker* t, TrackO& a, const sa_box& obs, const sa_box& pred) {
  a.track_length += 1;
  a.observed_boxes.push_back(obs);
  a.predicted_boxes.push_back(pred);
  if (t->o.history_length > 0 && a.observed_boxes.size() > t->o.history_length) {
    a.observed_boxes.pop_front();
    a.predicted_boxes.pop_front();
  }
}

// ObservationMetric::optimize for the freshly pushed last observation
void optimize(or_tracker* t, TrackO& a, bool is_merge) {
  Observation observation = a.observations.back();
  a.observations.pop_back();
  const sa_box observation_bbox = observation.bbox;
  sa_box predicted;
  or_make_prediction(t->o.kalman_position_weight, t->o.kalman_velocity_weight, a.has_state ? 1 : 0, a.mean, a.cov,
                     &observation_bbox, &predicted);
  a.has_state = true;
  update_history(t, a, observation_bbox, predicted);
  if (!t->o.visual) {
    a.observations.clear();  // SortMetric keeps exactly one observation
    observation.bbox = predicted;
    a.observations.push_back(observation);
    return;
  }
  if (is_merge && !can_use(t, observation_bbox, observation.quality, t->o.visual_minimal_quality_collect, observation.has_own,
                           observation.own, t->o.visual_minimal_own_area_percentage_collect)) {
    observation.has_feature = false;
    observation.feature.clear();
  }
  observation.bbox = predicted;
  observation.has_bbox = true;
  // optimize_observations
  auto& obs = a.observations;
  obs.erase(std::remove_if(obs.begin(), obs.end(), [](const Observation& e) { return !e.has_feature; }), obs.end());
  for (auto& e : obs) e.has_bbox = false;
  std::stable_sort(obs.begin(), obs.end(), [](const Observation& e1, const Observation& e2) { return e2.quality < e1.quality; });
  if (obs.size() >= t->o.visual_max_observations && !obs.empty()) obs.resize(obs.size() - 1);
  obs.push_back(observation);
  std::swap(obs[0], obs[obs.size() - 1]);
  a.visual_features_collected_count = 0;
  for (auto& e : obs) a.visual_features_collected_count += e.has_feature ? 1 : 0;
}

uint64_t current_epoch(const or_tracker* t, uint64_t scene) {
  auto it = t->epoch_db.find(scene);
  return it == t->epoch_db.end() ? 0 : it->second;
}

void auto_waste(or_tracker* t) {
  std::vector<uint64_t> w;
  for (auto& kv : t->store)
    if (kv.second.last_updated_epoch + t->o.max_idle_epochs < current_epoch(t, kv.second.scene_id)) w.push_back(kv.first);
  for (uint64_t id : w) {
    t->wasted[id] = t->store[id];
    t->store.erase(id);
  }
}

sa_sort_track sort_track(const or_tracker* t, const TrackO& a) {
  sa_sort_track s;
  std::memset(&s, 0, sizeof s);
  s.id = a.id;
  s.epoch = a.last_updated_epoch;
  s.predicted_bbox = a.predicted_boxes.back();
  s.observed_bbox = a.observed_boxes.back();
  s.scene_id = a.scene_id;
  s.length = a.track_length;
  s.voting_type = (t->o.visual && a.voting_type >= 0) ? a.voting_type : SA_VOTE_POSITIONAL;
  s.has_custom_object_id = a.has_custom;
  s.custom_object_id = a.custom;
  return s;
}

}  // namespace

extern "C" {

or_tracker* or_tracker_create(const sa_tracker_options* o) {
  or_tracker* t = new or_tracker();
  t->o = *o;
  t->cd.assign(o->constraint_epoch_delta, o->constraint_epoch_delta + o->n_constraints);
  t->cm.assign(o->constraint_max_dist, o->constraint_max_dist + o->n_constraints);
  t->aw_counter = o->auto_waste_periodicity;
  sa_config& c = t->cfg;
  std::memset(&c, 0, sizeof c);
  c.struct_size = sizeof c;
  c.positional_kind = o->positional_kind;
  c.positional_threshold = o->positional_threshold;
  c.positional_min_confidence = o->positional_min_confidence;
  c.visual_kind = o->visual ? o->visual_kind : SA_VIS_NONE;
  c.visual_threshold = o->visual_threshold;
  c.feature_len = o->feature_len;
  c.max_observations = o->visual ? o->visual_max_observations : 1;
  c.visual_min_votes = o->visual_min_votes;
  c.visual_minimal_track_length = o->visual_minimal_track_length;
  c.visual_minimal_area = o->visual_minimal_area;
  c.visual_minimal_quality_use = o->visual_minimal_quality_use;
  c.visual_minimal_own_area_percentage_use = o->visual_minimal_own_area_percentage_use;
  c.max_idle_epochs = o->max_idle_epochs;
  c.n_constraints = o->n_constraints;
  c.constraint_epoch_delta = t->cd.data();
  c.constraint_max_dist = t->cm.data();
  c.kf_position_weight = o->kalman_position_weight;
  c.kf_velocity_weight = o->kalman_velocity_weight;
  return t;
}

void or_tracker_destroy(or_tracker* t) { delete t; }

int or_tracker_predict_batch(or_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts,
                             const sa_observation* const* obs, sa_sort_track* const* out) {
  if (t->aw_counter == 0) { auto_waste(t); t->aw_counter = t->o.auto_waste_periodicity; }
  else t->aw_counter -= 1;
  const uint32_t K = t->o.visual ? t->o.visual_max_observations : 1, D = t->o.feature_len;
  // Batch*: the main thread builds candidates and requests distances scene by scene, the votes follow; the canonical
  // (deterministic) equivalent is scene-by-scene processing in request order against the store as it stands —
  // other scenes' tracks never interact (compatible() is false across scene ids).
  for (uint32_t s = 0; s < n_scenes; ++s) {
    const uint64_t scene = scene_ids[s];
    const uint32_t n = counts[s];
    const uint64_t epoch = ++t->epoch_db[scene];
    // own-area shares of the frame's boxes when either gate is armed (visual_sort/simple_api.rs:111-127)
    std::vector<float> shares;
    if (t->o.visual && n &&
        t->o.visual_minimal_own_area_percentage_collect + t->o.visual_minimal_own_area_percentage_use > 0.0f) {
      std::vector<sa_box> frame(n);
      for (uint32_t i = 0; i < n; ++i) frame[i] = obs[s][i].bbox;
      shares.resize(n);
      or_own_area_shares(n, frame.data(), shares.data());
    }
    // candidate tracks: one observation each, optimised (Kalman no-op step)
    std::vector<TrackO> cand(n);
    for (uint32_t i = 0; i < n; ++i) {
      const sa_observation& ob = obs[s][i];
      TrackO& c = cand[i];
      c.last_updated_epoch = epoch;
      c.scene_id = scene;
      c.has_custom = ob.has_custom_object_id != 0;
      c.custom = ob.custom_object_id;
      Observation o1;
      o1.has_bbox = true;
      o1.bbox = ob.bbox;
      o1.quality = ob.feature_quality == ob.feature_quality ? ob.feature_quality : 1.0f;
      o1.has_own = ob.own_area == ob.own_area || !shares.empty();   // a caller-supplied share takes precedence (facade extension)
      o1.own = ob.own_area == ob.own_area ? ob.own_area : (shares.empty() ? 0.0f : shares[i]);
      o1.has_feature = t->o.visual && ob.feature != nullptr;
      if (o1.has_feature) o1.feature.assign(ob.feature, ob.feature + D);
      c.observations.push_back(o1);
      optimize(t, c, false);
    }
    // the stored tracks of this scene, ascending id (std::map order)
    std::vector<uint64_t> ids;
    std::vector<sa_box> tboxes;
    std::vector<uint64_t> tepochs;
    std::vector<float> m5, c25, feats;
    std::vector<uint8_t> pres;
    for (auto& kv : t->store) {
      const TrackO& a = kv.second;
      if (a.scene_id != scene) continue;
      ids.push_back(a.id);
      tboxes.push_back(a.predicted_boxes.back());
      tepochs.push_back(a.last_updated_epoch);
      for (int r = 0; r < 5; ++r) m5.push_back(a.mean[r]);
      for (int r = 0; r < 5; ++r)
        for (int c = 0; c < 5; ++c) c25.push_back(a.cov[r * 10 + c]);
      if (t->o.visual) {
        size_t base = feats.size();
        feats.resize(base + (size_t)K * D, 0.0f);
        size_t pb = pres.size();
        pres.resize(pb + K, 0);
        for (uint32_t k = 0; k < a.observations.size() && k < K; ++k)
          if (a.observations[k].has_feature) {
            pres[pb + k] = 1;
            std::memcpy(&feats[base + (size_t)k * D], a.observations[k].feature.data(), (size_t)D * 4);
          }
      }
    }
    sa_tracks tr;
    std::memset(&tr, 0, sizeof tr);
    tr.n = (uint32_t)ids.size();
    tr.ids = ids.data(); tr.boxes = tboxes.data(); tr.epochs = tepochs.data(); tr.kf_mean = m5.data(); tr.kf_cov = c25.data();
    if (t->o.visual) { tr.feats = feats.data(); tr.feat_present = pres.data(); }
    std::vector<sa_box> cboxes(n);
    std::vector<float> cfeat((size_t)n * (D ? D : 1), 0.0f), cq(n), cown(n);
    std::vector<uint8_t> cpres(n);
    for (uint32_t i = 0; i < n; ++i) {
      const Observation& o1 = cand[i].observations[0];
      cboxes[i] = o1.bbox;
      cq[i] = o1.quality;
      cown[i] = o1.has_own ? o1.own : NAN;
      cpres[i] = o1.has_feature;
      if (o1.has_feature) std::memcpy(&cfeat[(size_t)i * D], o1.feature.data(), (size_t)D * 4);
    }
    sa_detections det;
    std::memset(&det, 0, sizeof det);
    det.n = n;
    det.boxes = cboxes.data();
    if (t->o.visual) { det.feats = cfeat.data(); det.feat_present = cpres.data(); det.feat_quality = cq.data(); det.own_area = cown.data(); }
    std::vector<uint64_t> win(n, 0);
    std::vector<uint8_t> vt(n, 0);
    or_frame_out fo;
    std::memset(&fo, 0, sizeof fo);
    fo.track_id = win.data();
    fo.voting_type = vt.data();
    or_associate(&t->cfg, (uint32_t)t->store.size(), &tr, epoch, &det, &fo);
    for (uint32_t i = 0; i < n; ++i) {
      TrackO& c = cand[i];
      uint64_t drawn = 0;
      if (t->o.batch_ids) drawn = ++t->track_id;
      uint64_t tid;
      if (win[i] == 0) {
        tid = t->o.batch_ids ? drawn : ++t->track_id;
        c.id = tid;
        t->store[tid] = c;
      } else {
        tid = win[i];
        TrackO& d = t->store[tid];
        if (t->o.visual) c.voting_type = vt[i];  // add_observation(.., VisualAttributesUpdate::VotingType(vt))
        // Track::merge: attributes, then observations of class 0, then optimize(is_merge = true)
        d.last_updated_epoch = c.last_updated_epoch;
        d.has_custom = c.has_custom;
        d.custom = c.custom;
        if (t->o.visual) d.voting_type = c.voting_type;
        for (auto& o1 : c.observations) d.observations.push_back(o1);
        optimize(t, d, true);
      }
      out[s][i] = sort_track(t, t->store[tid]);
    }
  }
  return 0;
}

int or_tracker_skip_epochs(or_tracker* t, uint64_t scene, uint64_t n) {
  t->epoch_db[scene] += n;
  auto_waste(t);
  return 0;
}
uint64_t or_tracker_current_epoch(or_tracker* t, uint64_t scene) { return current_epoch(t, scene); }
uint64_t or_tracker_active_tracks(or_tracker* t) { return t->store.size(); }
uint32_t or_tracker_wasted(or_tracker* t, sa_sort_track* out, uint32_t cap) {
  auto_waste(t);
  uint32_t n = 0;
  for (auto& kv : t->wasted) {
    if (n < cap) out[n] = sort_track(t, kv.second);
    ++n;
  }
  if (cap >= n) t->wasted.clear();
  return n;
}
uint32_t or_tracker_idle_tracks(or_tracker* t, uint64_t scene, sa_sort_track* out, uint32_t cap) {
  uint32_t n = 0;
  for (auto& kv : t->store)
    if (kv.second.scene_id == scene && kv.second.last_updated_epoch != current_epoch(t, scene)) {
      if (n < cap) out[n] = sort_track(t, kv.second);
      ++n;
    }
  return n;
}
int or_tracker_track_state(or_tracker* t, uint64_t id, float* mean10, float* cov100) {
  auto it = t->store.find(id);
  if (it == t->store.end()) return -1;
  std::memcpy(mean10, it->second.mean, sizeof it->second.mean);
  std::memcpy(cov100, it->second.cov, sizeof it->second.cov);
  return 0;
}

int or_tracker_track_info(or_tracker* t, uint64_t id, uint64_t out4[4]) {
  auto it = t->store.find(id);
  if (it == t->store.end()) return -1;
  out4[0] = it->second.visual_features_collected_count;
  out4[1] = it->second.observations.size();
  out4[2] = it->second.observed_boxes.size();
  out4[3] = it->second.track_length;
  return 0;
}

}  // extern "C"
