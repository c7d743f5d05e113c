// sa_tracker.cpp — host facade with the reference's tracker surface (include/similari_tracker.h) on top of the
// association engine.  Association (cost matrices + votes) = sa_associate_batch on the GPU; everything in this
// file is the O(N) bookkeeping the reference also does on the host around that call:
//   Sort::predict_with_scene          src/trackers/sort/simple_api.rs:110-196
//   VisualSort::predict_with_scene    src/trackers/visual_sort/simple_api.rs:99-230
//   Batch*::predict / voting_thread   src/trackers/sort/batch_api.rs:68-153,222-290
//   SortMetric / VisualMetric::optimize (Kalman step, history, feature bank)   sort/metric.rs:79-105,
//                                     visual_sort/metric.rs:129-154,297-374
//   TrackerAPI (epochs, waste)        src/trackers/tracker_api.rs, epoch_db.rs
// The Kalman step is written against the filter's structure (motion = I + shift, update matrix = [I 0]); skipping
// the multiplications by the constant 0/1 entries leaves every f32 result unchanged (kalman_2d_box.rs:58-148).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/similari_tracker.h"
#include "sa_kalman.h"

namespace {

thread_local std::string g_err;

using KF = sa_kf;  // sa_kalman.h: the reference's box filter, shared with the device-side upkeep

// make_prediction  kalman_prediction.rs:13-32
sa_box make_prediction(float pw, float vw, bool& has_state, KF& s, const sa_box& obs) {
  sa_box r = sa_kf_make_prediction(pw, vw, has_state, s, obs);
  has_state = true;
  return r;
}

struct Obs {  // one stored observation of a VisualSORT track; index 0 carries the bbox
  float quality = 1.0f;
  bool has_own = false;
  float own = 0.0f;
  bool has_feat = false;
  std::vector<float> feat;   // host upkeep only: with device upkeep the vectors live in the device bank
};

// The last `cap` (observed, predicted) boxes of a track (SortAttributes::observed_boxes / predicted_boxes are VecDeques trimmed to the
// history length, sort.rs:160-176; the two are always pushed together): ONE fixed ring of pairs — no allocation per frame, one heap
// block per track.
struct BoxPair { sa_box observed, predicted; };
struct Ring {
  std::vector<BoxPair> v;
  uint32_t head = 0, count = 0;
  void push(const sa_box& observed, const sa_box& predicted, uint32_t cap) {
    if (v.size() != cap) { v.resize(cap); head = 0; count = 0; }
    uint32_t at;
    if (count < cap) { at = head + count++; at = at >= cap ? at - cap : at; }
    else { at = head; head = head + 1 == cap ? 0 : head + 1; }
    v[at].observed = observed;
    v[at].predicted = predicted;
  }
  const BoxPair& back() const { const uint32_t at = head + count - 1, cap = (uint32_t)v.size(); return v[at >= cap ? at - cap : at]; }
  uint32_t size() const { return count; }
};

struct Track {  // (what a frame's bookkeeping touches first; the filter state — 440 bytes the device owns under device upkeep — last)
  uint64_t id = 0, scene = 0, epoch = 0, length = 0;
  bool has_custom = false;
  int64_t custom = 0;
  int32_t voting = -1;  // VisualAttributes::voting_type: None
  bool has_state = false;
  bool in_engine = true;   // false: evicted from the engine's table (it can never match again) but not wasted yet
  uint32_t feat_count = 0;
  Ring boxes;
  std::vector<Obs> obs;
  KF kf;
};

}  // namespace

struct sa_tracker {
  sa_tracker_options o{};
  std::vector<uint64_t> cons_delta;
  std::vector<float> cons_dist;
  sa_engine* eng = nullptr;
  std::string err;
  uint64_t track_id = 0;
  std::map<uint64_t, uint64_t> epochs;            // scene -> current epoch
  std::unordered_map<uint64_t, Track> store;      // main store (node-based: a Track's address is stable while it lives)
  // scene -> its tracks in the order of the engine's table for that scene: rows are appended in creation order (ids ascend) and
  // removals close the gaps on both sides, so the winner the engine reports as a COLUMN (sa_batch_fetch_cols) is rows[column] —
  // no lookup by id on the per-candidate path
  std::map<uint64_t, std::vector<Track*>> by_scene;
  // Eviction.  A track whose last update lies more than max_idle_epochs behind its scene's epoch fails compatible() (sort.rs:250-270) for
  // every later frame — epochs only grow — but the reference keeps it in the store until the next auto_waste (every 100th predict by
  // default), and so would the engine's table: at 5 % churn a 1000-object VisualSORT loop associates against 6 700 rows instead of
  // 1 200.  The facade therefore takes such tracks out of the ENGINE's table as soon as they are 64 and a sixteenth of it (sa_tracks_remove:
  // one gather launch), and keeps them in its own store — idle_tracks / wasted see them as before.
  std::map<uint64_t, std::vector<uint64_t>> row_epoch;   // scene -> last_updated_epoch of by_scene's rows, in the same order (the scan's input)
  std::map<uint64_t, std::vector<Track*>> evicted;       // scene -> tracks out of the engine's table, still in `store`
  std::vector<Track> wasted_store;
  uint32_t waste_counter = 0;
  // predict()'s per-scene work arrays, kept between calls: a frame allocates nothing once the arrays have grown to its size
  struct SceneScratch {
    std::vector<sa_box> cboxes, dev_pred;      // the candidates' boxes after their own Kalman no-op step ; the device's predicted boxes
    std::vector<float> cq, cown, shares;
    std::vector<const float*> cfeat;           // one pointer per detection: the engine gathers the rows itself ...
    std::vector<uint8_t> cpres, votes;
    std::vector<uint64_t> winners, tids, new_ids;
    std::vector<int32_t> wcols;
    std::vector<Track*> trps;
    uint8_t contiguous = 0;                    // ... unless they already ARE one N x D block (then: no gather at all)
  };
  std::vector<SceneScratch> scratch[2];     // two sets, used in turn: the set of the previous predict() may still hold deferred work
  int cur_set = 0;
  std::vector<uint64_t> sc_epoch, sc_id_base, sc_touched;
  std::vector<sa_scene_request> sc_req;
  // Deferred bookkeeping (device upkeep queued behind the association: the fused path).  What a predict() RETURNS needs, per continued
  // track, one cache line of its record (id, length, epoch, custom id, vote); the rest of the reference's merge — the history deques
  // (sort.rs:160-176) and the observation policy (visual_sort/metric.rs:129-154) — only has to be in place before anything READS it:
  // the next predict()'s own merges, idle_tracks, wasted, track_info.  It is therefore run by the NEXT call on the tracker, and a
  // predict() runs it while its own association is on the device (the host would wait there anyway); every other entry point runs
  // it first.  All it reads lies in the scratch set of the frame that left it behind.
  bool pending = false;
  int pending_set = 0;
  uint32_t pending_scenes = 0;
  std::vector<uint32_t> pending_counts;
};

namespace {

int tfail(sa_tracker* t, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (t) t->err = buf;
  else g_err = buf;
  return code;
}

bool feature_can_be_used(const sa_tracker_options& o, const sa_box& b, float q, float min_q, bool has_own, float own, float min_own) {
  return sa_feature_can_be_used(o.visual_minimal_area, b, q, min_q, has_own, own, min_own);
}

void update_history(const sa_tracker_options& o, Track& tr, const sa_box& observed, const sa_box& predicted) {
  tr.length += 1;                                     // sort.rs:160-176, track_attributes.rs:60-78
  tr.boxes.push(observed, predicted, o.history_length);
}

// What to_sort_track would read from a track whose history had just taken (observed, predicted) — without the history
sa_sort_track to_sort_track_with(const sa_tracker_options& o, const Track& tr, const sa_box& observed, const sa_box& predicted) {
  sa_sort_track s;
  std::memset(&s, 0, sizeof s);
  s.id = tr.id;
  s.epoch = tr.epoch;
  s.predicted_bbox = predicted;
  s.observed_bbox = observed;
  s.scene_id = tr.scene;
  s.length = tr.length;
  s.voting_type = (o.visual && tr.voting >= 0) ? tr.voting : SA_VOTE_POSITIONAL;
  s.has_custom_object_id = tr.has_custom ? 1 : 0;
  s.custom_object_id = tr.custom;
  return s;
}

sa_sort_track to_sort_track(const sa_tracker_options& o, const Track& tr) {
  sa_sort_track s;
  std::memset(&s, 0, sizeof s);
  s.id = tr.id;
  s.epoch = tr.epoch;
  const BoxPair& last = tr.boxes.back();
  s.predicted_bbox = last.predicted;
  s.observed_bbox = last.observed;
  s.scene_id = tr.scene;
  s.length = tr.length;
  // Sort: always Positional (sort/simple_api.rs:260) ; VisualSort: attrs.voting_type.unwrap_or(Positional)
  s.voting_type = (o.visual && tr.voting >= 0) ? tr.voting : SA_VOTE_POSITIONAL;
  s.has_custom_object_id = tr.has_custom ? 1 : 0;
  s.custom_object_id = tr.custom;
  return s;
}

uint64_t current_epoch(sa_tracker* t, uint64_t scene) {
  auto it = t->epochs.find(scene);
  return it == t->epochs.end() ? 0 : it->second;
}

// get_main_store_wasted + auto_waste  tracker_api.rs:68-88 ; baked(): epoch_db.rs:52-66
int auto_waste(sa_tracker* t) {
  std::map<uint64_t, std::vector<uint64_t>> gone;
  for (auto& kv : t->store) {
    const Track& tr = kv.second;
    if (tr.epoch + t->o.max_idle_epochs < current_epoch(t, tr.scene)) gone[tr.scene].push_back(tr.id);
  }
  for (auto& kv : gone) {
    std::sort(kv.second.begin(), kv.second.end());
    std::vector<uint64_t> resident;   // those the engine's table still holds (the others were evicted from it earlier)
    for (uint64_t id : kv.second)
      if (t->store[id].in_engine) resident.push_back(id);
    int rc = resident.empty() ? SA_OK : sa_tracks_remove(t->eng, kv.first, (uint32_t)resident.size(), resident.data());
    if (rc != SA_OK) return tfail(t, rc, "sa_tracks_remove: %s", sa_last_error(t->eng));
    auto& rows = t->by_scene[kv.first];
    auto& eps = t->row_epoch[kv.first];
    size_t w = 0;
    for (size_t r = 0; r < rows.size(); ++r)
      if (!std::binary_search(kv.second.begin(), kv.second.end(), rows[r]->id)) { rows[w] = rows[r]; eps[w] = eps[r]; ++w; }
    rows.resize(w);
    eps.resize(w);
    auto& ev = t->evicted[kv.first];
    ev.erase(std::remove_if(ev.begin(), ev.end(), [&](const Track* tr) { return std::binary_search(kv.second.begin(), kv.second.end(), tr->id); }), ev.end());
    for (uint64_t id : kv.second) {
      t->wasted_store.push_back(std::move(t->store[id]));
      t->store.erase(id);
    }
  }
  return SA_OK;
}

// Pushes the rows of `ids` (tracks of one scene that were created or merged this frame) to the engine.
int sync_engine(sa_tracker* t, uint64_t scene, const std::vector<uint64_t>& ids) {
  if (ids.empty()) return SA_OK;
  const uint32_t n = (uint32_t)ids.size(), K = t->o.visual ? t->o.visual_max_observations : 1, D = t->o.feature_len;
  std::vector<sa_box> boxes(n);
  std::vector<uint64_t> epochs(n);
  std::vector<float> mean(n * 5), cov(n * 25), feats;
  std::vector<uint8_t> present;
  if (t->o.visual) { feats.assign((size_t)n * K * D, 0.0f); present.assign((size_t)n * K, 0); }
  for (uint32_t i = 0; i < n; ++i) {
    const Track& tr = t->store[ids[i]];
    boxes[i] = tr.boxes.back().predicted;
    epochs[i] = tr.epoch;
    for (int a = 0; a < 5; ++a) {
      mean[i * 5 + a] = tr.kf.mean[a];
      for (int b = 0; b < 5; ++b) cov[i * 25 + a * 5 + b] = tr.kf.cov[a * 10 + b];
    }
    if (t->o.visual)
      for (uint32_t k = 0; k < tr.obs.size() && k < K; ++k)
        if (tr.obs[k].has_feat) {
          present[(size_t)i * K + k] = 1;
          std::memcpy(&feats[((size_t)i * K + k) * D], tr.obs[k].feat.data(), (size_t)D * 4);
        }
  }
  sa_tracks st;
  std::memset(&st, 0, sizeof st);
  st.n = n; st.ids = ids.data(); st.boxes = boxes.data(); st.epochs = epochs.data();
  st.kf_mean = mean.data(); st.kf_cov = cov.data();
  if (t->o.visual) { st.feats = feats.data(); st.feat_present = present.data(); }
  int rc = sa_tracks_upsert(t->eng, scene, &st);
  if (rc != SA_OK) return tfail(t, rc, "sa_tracks_upsert: %s", sa_last_error(t->eng));
  return SA_OK;
}

// optimize_observations  visual_sort/metric.rs:129-154 on a track's stored observations: keep those with a feature, stable sort by
// quality (descending), drop the worst when the bank is full, push the new one and bring it to the front.  At most SA_MAX_BANK + 1
// entries: an insertion sort in place, no allocation per merged track.
void optimize_observations(std::vector<Obs>& obs, Obs&& nw, uint32_t max_observations) {
  size_t n = 0;
  for (size_t k = 0; k < obs.size(); ++k)
    if (obs[k].has_feat) {
      if (n != k) obs[n] = std::move(obs[k]);
      ++n;
    }
  obs.resize(n);
  for (size_t a = 1; a < n; ++a)  // stable: an element moves left only past strictly smaller qualities
    for (size_t c = a; c > 0 && obs[c - 1].quality < obs[c].quality; --c) std::swap(obs[c - 1], obs[c]);
  if (n >= max_observations && n > 0) obs.pop_back();
  obs.push_back(std::move(nw));
  std::swap(obs.front(), obs.back());
}

// The heavy half of a continued track's merge under device upkeep: history (the boxes are the candidate's and the device's prediction)
// and, for VisualSort, the observation policy on the bookkeeping records (the feature rows move inside the device bank, sa_upkeep.hip).
inline void merge_heavy(const sa_tracker_options& o, Track& tr, const sa_tracker::SceneScratch& W, uint32_t i) {
  tr.boxes.push(W.cboxes[i], W.dev_pred[i], o.history_length);
  if (o.visual) {
    Obs nw;
    nw.quality = W.cq[i];
    nw.has_own = W.cown[i] == W.cown[i];
    nw.own = nw.has_own ? W.cown[i] : 0.0f;
    nw.has_feat = W.cpres[i] != 0;
    if (!feature_can_be_used(o, W.cboxes[i], nw.quality, o.visual_minimal_quality_collect, nw.has_own, nw.own, o.visual_minimal_own_area_percentage_collect))
      nw.has_feat = false;
    optimize_observations(tr.obs, std::move(nw), o.visual_max_observations);
    tr.feat_count = 0;
    for (auto& so : tr.obs) tr.feat_count += so.has_feat ? 1u : 0u;
  }
}

// Runs what the previous predict() deferred (sa_tracker::pending).  Idempotent; every entry point that reads a track's history or
// observations calls it first.
void flush_pending(sa_tracker* t) {
  if (!t->pending) return;
  t->pending = false;
  const std::vector<sa_tracker::SceneScratch>& ss = t->scratch[t->pending_set];
  for (uint32_t s = 0; s < t->pending_scenes; ++s) {
    const sa_tracker::SceneScratch& W = ss[s];
    const uint32_t n = t->pending_counts[s];
    for (uint32_t i = 0; i < n; ++i)
      if (W.winners[i] != 0) merge_heavy(t->o, *W.trps[i], W, i);
  }
}

int predict_scenes(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts,
                   const sa_observation* const* obs, sa_sort_track* const* out) {
  const sa_tracker_options& o = t->o;
  const float pw = o.kalman_position_weight, vw = o.kalman_velocity_weight;
  const uint32_t D = o.feature_len;
  for (uint32_t s = 0; s < n_scenes; ++s)
    for (uint32_t s2 = 0; s2 < s; ++s2)
      if (scene_ids[s] == scene_ids[s2]) return tfail(t, SA_ERR_BAD_ARG, "scene %llu appears twice in one batch", (unsigned long long)scene_ids[s]);
  // SA_TRACKER_TRACE=1: where a predict() spends its time, in microseconds on stderr
  static const bool trace = getenv("SA_TRACKER_TRACE") != nullptr;
  using clk = std::chrono::steady_clock;
  const auto t_entry = clk::now();
  // auto waste (simple_api.rs:115-120)
  if (t->waste_counter == 0) {
    flush_pending(t);   // (wasted tracks are read out with their histories)
    int rc = auto_waste(t);
    if (rc != SA_OK) return rc;
    t->waste_counter = o.auto_waste_periodicity;
  } else t->waste_counter -= 1;

  // (the other scratch set may hold the previous frame's deferred bookkeeping: it is run below, while this frame's association is on the device)
  const int set = t->pending ? (t->pending_set ^ 1) : t->cur_set;
  t->cur_set = set;
  if (t->scratch[set].size() < n_scenes) t->scratch[set].resize(n_scenes);
  t->sc_epoch.resize(n_scenes);
  t->sc_req.resize(n_scenes);
  std::vector<sa_tracker::SceneScratch>& ss = t->scratch[set];
  std::vector<uint64_t>& epoch = t->sc_epoch;
  std::vector<sa_scene_request>& req = t->sc_req;
  for (uint32_t s = 0; s < n_scenes; ++s) {
    const uint32_t n = counts[s];
    sa_tracker::SceneScratch& W = ss[s];
    epoch[s] = ++t->epochs[scene_ids[s]];  // next_epoch  epoch_db.rs:35-49
    W.cboxes.resize(n);
    if (o.visual) { W.cfeat.resize(n); W.cq.resize(n); W.cown.resize(n); W.cpres.resize(n); }
    for (uint32_t i = 0; i < n; ++i) {
      const sa_box& bb = obs[s][i].bbox;
      if (!(bb.aspect > 0.0f) || !(bb.height > 0.0f) || !(bb.confidence >= 0.0f && bb.confidence <= 1.0f))
        return tfail(t, SA_ERR_BAD_ARG, "observation %u of scene %llu: bad box", i, (unsigned long long)scene_ids[s]);
    }
    // exclusively_owned_areas_normalized_shares over the frame's observed boxes, when either own-area gate is armed
    // (visual_sort/simple_api.rs:111-127) — on the GPU (sa_own_areas).  A share the caller supplies takes precedence.
    std::vector<float>& shares = W.shares;
    shares.clear();
    if (o.visual && n && o.visual_minimal_own_area_percentage_collect + o.visual_minimal_own_area_percentage_use > 0.0f) {
      bool any_missing = false;
      for (uint32_t i = 0; i < n; ++i) any_missing = any_missing || obs[s][i].own_area != obs[s][i].own_area;
      if (any_missing) {
        std::vector<sa_box> frame(n);
        for (uint32_t i = 0; i < n; ++i) frame[i] = obs[s][i].bbox;
        shares.resize(n);
        int rc = sa_own_areas(t->eng, n, frame.data(), shares.data());
        if (rc != SA_OK) return tfail(t, rc, "%s", sa_last_error(t->eng));
      }
    }
    // The throw-away candidate track of a detection (simple_api.rs:125-145) is never materialised: its box goes into the request, the
    // rest of it (custom id, feature pointer) is read from the caller's observation when a track takes it over.
    // The candidate's own Kalman step (initiate -> predict -> update with the box it was initiated from, kalman_prediction.rs:13-32)
    // is the identity on the box: zero velocity, zero innovation, so the new mean is the observation plus (+-0) * gain.  All that
    // changes is TryFrom<KalmanState>: an angle of exactly 0.0 reads back as None (kalman.rs:82-86).  The 10 x 10 filter arithmetic
    // (about 1 us per detection on the host) is therefore only run for the candidates that become tracks and need the state
    // (below); the tests compare every box with the oracle, which does run the filter.
    const float* block = nullptr;  // where row 0 of an N x D block would lie, if the features form one
    bool one_block = o.visual && n > 0, all_present = true;
    sa_box* cb = W.cboxes.data();
    for (uint32_t i = 0; i < n; ++i) {
      const sa_observation& ob = obs[s][i];
      sa_box& c = cb[i];
      c = ob.bbox;
      c.angle = ob.bbox.has_angle ? ob.bbox.angle : 0.0f;
      c.has_angle = (ob.bbox.has_angle && ob.bbox.angle != 0.0f) ? 1 : 0;
      c.reserved = 0;
      if (o.visual) {
        const bool has_own = ob.own_area == ob.own_area || !shares.empty();
        const float own = ob.own_area == ob.own_area ? ob.own_area : (shares.empty() ? 0.0f : shares[i]);
        const bool has_feat = ob.feature != nullptr;
        W.cq[i] = ob.feature_quality == ob.feature_quality ? ob.feature_quality : 1.0f;
        W.cown[i] = has_own ? own : NAN;
        W.cpres[i] = has_feat ? 1 : 0;
        W.cfeat[i] = ob.feature;
        all_present = all_present && has_feat;
        if (has_feat) {
          if (!block) block = ob.feature - (size_t)i * D;
          one_block = one_block && ob.feature == block + (size_t)i * D;
        }
      }
    }
    W.winners.resize(n);
    W.votes.resize(n);
    W.wcols.resize(n);
    sa_scene_request& r = req[s];
    std::memset(&r, 0, sizeof r);
    r.scene_id = scene_ids[s];
    r.epoch = epoch[s];
    r.detections.n = n;
    r.detections.boxes = W.cboxes.data();
    W.contiguous = 0;
    if (o.visual) {
      r.detections.feat_present = W.cpres.data();
      r.detections.feat_quality = W.cq.data();
      r.detections.own_area = W.cown.data();
      // The observations' features as ONE block (a producer that writes its N x D output contiguously — a ReID head's output buffer,
      // host or device): handed over as such — read in place when the block is pinned (sa_host_alloc) or registered device memory
      // (sa_device_block_register), one memcpy otherwise — instead of one gather per row.  Rows of detections without a feature are
      // never dereferenced by the host (flagged absent).  Only when every row lies inside a block the engine knows: rows of
      // absent detections may otherwise be unmapped memory.
      if (one_block && block && all_present) {
        W.contiguous = 1;
        r.detections.feats = block;
      }
    }
  }
  // eviction (see sa_tracker::row_epoch): tracks of these scenes that no frame from now on can match leave the engine's table
  for (uint32_t s = 0; s < n_scenes; ++s) {
    auto& rows = t->by_scene[scene_ids[s]];
    auto& eps = t->row_epoch[scene_ids[s]];
    const uint64_t cur = epoch[s];
    size_t expired = 0;
    for (uint64_t ep : eps) expired += ep + o.max_idle_epochs < cur ? 1u : 0u;
    // (a removal is a drain + one gather launch, ~20 us; a row left in the table costs the frames until auto_waste ~1/64 us each: it pays
    // from about 64 rows on — with 16 a batch tracker of 8 x 500 objects spent 78 us per predict() removing a few rows from four scenes)
    if (expired < 64 || expired * 16 < rows.size()) continue;
    std::vector<uint64_t> out_ids;
    out_ids.reserve(expired);
    auto& ev = t->evicted[scene_ids[s]];
    size_t w = 0;
    for (size_t r = 0; r < rows.size(); ++r) {
      if (eps[r] + o.max_idle_epochs < cur) { out_ids.push_back(rows[r]->id); rows[r]->in_engine = false; ev.push_back(rows[r]); }
      else { rows[w] = rows[r]; eps[w] = eps[r]; ++w; }
    }
    rows.resize(w);
    eps.resize(w);
    int rce = sa_tracks_remove(t->eng, scene_ids[s], (uint32_t)out_ids.size(), out_ids.data());
    if (rce != SA_OK) return tfail(t, rce, "sa_tracks_remove: %s", sa_last_error(t->eng));
  }
  const auto t_built = clk::now();
  double us_apply = 0.0, us_new = 0.0;
  uint32_t n_new_tracks = 0;
  // ---- the hot path: foreign_track_distances + voting.winners, on the GPU ----
  int rc = sa_batch_begin(t->eng);
  const auto t_begun = clk::now();
  for (uint32_t s = 0; s < n_scenes && rc == SA_OK; ++s)
    rc = ss[s].contiguous ? sa_batch_add(t->eng, req[s].scene_id, req[s].epoch, &req[s].detections, nullptr)
                          : sa_batch_add_rows(t->eng, req[s].scene_id, req[s].epoch, &req[s].detections, o.visual ? ss[s].cfeat.data() : nullptr, nullptr);
  const auto t_added = clk::now();
  // Device upkeep: the Kalman step, the table refresh and the feature-bank policy of every scene are queued right BEHIND the association
  // on the device (sa_batch_run_apply) — the ids of the tracks that start are a function of the winners alone (a counter, in candidate
  // order), so the device draws them itself and this thread waits once, for everything.  (Several scenes under Sort / VisualSort id
  // rules — one id per NEW track across scenes — would make a scene's first id depend on the previous scenes' winners: two phases then.)
  const bool fused = o.device_upkeep && (o.batch_ids || n_scenes == 1);
  if (rc == SA_OK && fused) {
    std::vector<uint64_t>& id_base = t->sc_id_base;
    id_base.resize(n_scenes);
    uint64_t next = t->track_id;
    for (uint32_t s = 0; s < n_scenes; ++s) { id_base[s] = next; next += counts[s]; }   // (batch ids: one per candidate; a single scene: its own counter)
    rc = sa_batch_run_apply(t->eng, id_base.data(), o.batch_ids ? 1 : 0);
  } else
  if (rc == SA_OK) rc = sa_batch_run(t->eng);
  const auto t_run = clk::now();
  flush_pending(t);   // the previous frame's deferred bookkeeping: while this frame's kernels run (or, on an error path, before the return)
  if (rc == SA_OK && !fused) rc = sa_batch_sync(t->eng);  // (fused: sa_batch_fetch waits for the END OF THE ASSOCIATION only — the upkeep kernels
                                                          // queued behind it run while this thread does its bookkeeping below)
  for (uint32_t s = 0; s < n_scenes && rc == SA_OK; ++s) {
    rc = sa_batch_fetch(t->eng, s, ss[s].winners.data(), ss[s].votes.data());
    if (rc == SA_OK) rc = sa_batch_fetch_cols(t->eng, s, ss[s].wcols.data());
  }
  if (rc != SA_OK) return tfail(t, rc, "association: %s", sa_last_error(t->eng));
  const auto t_assoc = clk::now();

  // ids first, scene by scene in the order the reference draws them; with device upkeep the Kalman step, table refresh and
  // feature-bank policy of every scene are QUEUED right away (sa_tracks_apply_begin) — they run while this thread does the
  // per-track bookkeeping that does not need their result; the predicted boxes are collected afterwards (sa_tracks_apply_end)
  std::vector<uint64_t>& touched = t->sc_touched;
  for (uint32_t s = 0; s < n_scenes; ++s) {
    const uint32_t n = counts[s];
    sa_tracker::SceneScratch& W = ss[s];
    W.tids.resize(n);
    W.new_ids.resize(n);
    const uint64_t* win = W.winners.data();
    for (uint32_t i = 0; i < n; ++i) {
      const uint64_t dest = win[i];
      uint64_t drawn = 0;
      if (o.batch_ids) drawn = ++t->track_id;          // Batch*: an id per candidate (batch_api.rs:102-106)
      if (dest == 0) { W.tids[i] = o.batch_ids ? drawn : ++t->track_id; W.new_ids[i] = W.tids[i]; }
      else { W.tids[i] = dest; W.new_ids[i] = 0; }
    }
    if (o.device_upkeep && !fused) {
      const auto ta = clk::now();
      rc = sa_tracks_apply_begin(t->eng, s, W.new_ids.data());
      if (rc != SA_OK) return tfail(t, rc, "sa_tracks_apply: %s", sa_last_error(t->eng));
      us_apply += std::chrono::duration<double, std::micro>(clk::now() - ta).count();
    }
  }
  for (uint32_t s = 0; s < n_scenes; ++s) {
    const uint64_t scene = scene_ids[s];
    const uint32_t n = counts[s];
    sa_tracker::SceneScratch& W = ss[s];
    touched.clear();
    W.trps.resize(n);
    std::vector<Track*>& rows = t->by_scene[scene];
    std::vector<uint64_t>& eps = t->row_epoch[scene];
    const size_t rows_before = rows.size();  // the table the engine voted against: columns refer to these rows
    for (uint32_t i = 0; i < n; ++i) {
      const sa_observation& ob = obs[s][i];
      const sa_box& cbox = W.cboxes[i];
      const uint64_t dest = W.winners[i];
      const bool c_has_feat = o.visual && W.cpres[i] != 0;
      const float c_quality = o.visual ? W.cq[i] : 1.0f;
      const bool c_has_own = o.visual && W.cown[i] == W.cown[i];
      const float c_own = c_has_own ? W.cown[i] : 0.0f;
      Track* trp;
      if (dest == 0) {
        // winner == self or none: the candidate becomes a new track (simple_api.rs:167-187)
        const auto tn0 = trace ? clk::now() : clk::time_point();
        ++n_new_tracks;
        Track& tr = t->store[W.tids[i]];
        trp = &tr;
        tr.id = W.tids[i]; tr.scene = scene; tr.epoch = epoch[s];
        tr.has_custom = ob.has_custom_object_id != 0; tr.custom = ob.custom_object_id;
        tr.has_state = true;
        if (!o.device_upkeep) { bool hs = false; make_prediction(pw, vw, hs, tr.kf, ob.bbox); }  // with device upkeep the state is born on the GPU
        tr.length = 0;
        update_history(o, tr, ob.bbox, cbox);
        if (o.visual) {
          tr.obs.reserve(o.visual_max_observations + 1);
          tr.obs.emplace_back();                         // is_merge = false: the feature is kept as is
          Obs& nb = tr.obs.back();
          nb.quality = c_quality; nb.has_own = c_has_own; nb.own = c_own; nb.has_feat = c_has_feat;
          if (!o.device_upkeep && c_has_feat) nb.feat.assign(ob.feature, ob.feature + D);  // device upkeep: the vectors live in the device bank only
          tr.feat_count = c_has_feat ? 1 : 0;
        }
        rows.push_back(trp);
        eps.push_back(epoch[s]);
        if (trace) us_new += std::chrono::duration<double, std::micro>(clk::now() - tn0).count();
      } else {
        // the winner as a column of the table the engine voted against = a row of `rows` (checked; by id if the orders ever disagree)
        const int32_t col = W.wcols[i];
        if (col >= 0 && (size_t)col < rows_before && rows[col]->id == dest) { trp = rows[col]; eps[col] = epoch[s]; }
        else {
          auto it = t->store.find(dest);
          if (it == t->store.end()) return tfail(t, SA_ERR_STATE, "engine returned unknown track id %llu", (unsigned long long)dest);
          trp = &it->second;
          for (size_t r = 0; r < rows.size(); ++r)
            if (rows[r] == trp) { eps[r] = epoch[s]; break; }
        }
        Track& tr = *trp;
        // TrackAttributes::merge  sort.rs:272-276 / track_attributes.rs:210-215
        tr.epoch = epoch[s];
        tr.has_custom = ob.has_custom_object_id != 0; tr.custom = ob.custom_object_id;
        if (o.visual) tr.voting = W.votes[i];
        // optimize(is_merge = true): Kalman predict + update with the candidate's box, history (device upkeep: once the boxes are back)
        if (!o.device_upkeep) update_history(o, tr, cbox, make_prediction(pw, vw, tr.has_state, tr.kf, cbox));
        if (fused) tr.length += 1;   // (the rest of the merge — history, observation policy — is deferred: merge_heavy / flush_pending)
        else if (o.visual) {
          Obs nw;
          nw.quality = c_quality; nw.has_own = c_has_own; nw.own = c_own; nw.has_feat = c_has_feat;
          if (!feature_can_be_used(o, cbox, nw.quality, o.visual_minimal_quality_collect, nw.has_own, nw.own,
                                   o.visual_minimal_own_area_percentage_collect))
            nw.has_feat = false;
          if (!o.device_upkeep && nw.has_feat) nw.feat.assign(ob.feature, ob.feature + D);
          // (with device upkeep: the bookkeeping only — the same policy moves the feature rows inside the device bank, sa_upkeep.hip)
          optimize_observations(tr.obs, std::move(nw), o.visual_max_observations);
          tr.feat_count = 0;
          for (auto& so : tr.obs) tr.feat_count += so.has_feat ? 1u : 0u;
        }
      }
      W.trps[i] = trp;
      if (!o.device_upkeep) {
        touched.push_back(W.tids[i]);
        out[s][i] = to_sort_track(o, *trp);
      }
    }
    if (o.device_upkeep) continue;
    rc = sync_engine(t, scene, touched);
    if (rc != SA_OK) return rc;
  }
  if (o.device_upkeep)
    for (uint32_t s = 0; s < n_scenes; ++s) {
      const uint32_t n = counts[s];
      sa_tracker::SceneScratch& W = ss[s];
      W.dev_pred.resize(n);
      const auto ta = clk::now();
      rc = fused ? sa_tracks_apply_collect(t->eng, s, nullptr, W.dev_pred.data()) : sa_tracks_apply_end(t->eng, s, W.dev_pred.data());
      if (rc != SA_OK) return tfail(t, rc, "sa_tracks_apply: %s", sa_last_error(t->eng));
      us_apply += std::chrono::duration<double, std::micro>(clk::now() - ta).count();
      if (fused) {
        for (uint32_t i = 0; i < n; ++i)
          out[s][i] = W.winners[i] != 0 ? to_sort_track_with(o, *W.trps[i], W.cboxes[i], W.dev_pred[i]) : to_sort_track(o, *W.trps[i]);
        continue;
      }
      for (uint32_t i = 0; i < n; ++i) {
        Track& tr = *W.trps[i];
        if (W.winners[i] != 0) update_history(o, tr, W.cboxes[i], W.dev_pred[i]);
        out[s][i] = to_sort_track(o, tr);
      }
    }
  if (fused) {
    t->pending = true;
    t->pending_set = set;
    t->pending_scenes = n_scenes;
    t->pending_counts.assign(counts, counts + n_scenes);
  }
  if (trace) {
    const auto t_end = clk::now();
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "[sa_tracker] assemble %.1f  associate %.1f (begin %.1f stage %.1f enqueue %.1f wait+fetch %.1f)  apply %.1f  bookkeeping %.1f us\n",
            us(t_entry, t_built), us(t_built, t_assoc), us(t_built, t_begun), us(t_begun, t_added), us(t_added, t_run), us(t_run, t_assoc), us_apply,
            us(t_assoc, t_end) - us_apply);
    fprintf(stderr, "[sa_newtracks] %u tracks started in %.1f us (with the trace's own clock reads)\n", n_new_tracks, us_new);
  }
  return SA_OK;
}

}  // namespace

extern "C" {

void sa_tracker_options_default(sa_tracker_options* o, int visual) {
  if (!o) return;
  std::memset(o, 0, sizeof *o);
  o->struct_size = sizeof *o;
  o->device = -1;
  o->visual = visual ? 1 : 0;
  o->auto_waste_periodicity = 100;
  o->positional_kind = SA_POS_IOU;
  o->positional_threshold = 0.3f;
  o->kalman_position_weight = 1.0f / 20.0f;
  o->kalman_velocity_weight = 1.0f / 160.0f;
  if (!visual) {
    o->history_length = 1;
    o->max_idle_epochs = 5;
    o->positional_min_confidence = 0.05f;
  } else {
    o->history_length = 10;                 // VisualSortOptions::default  options.rs:194-205
    o->max_idle_epochs = 2;
    o->positional_min_confidence = 0.1f;    // VisualMetricBuilder::default  metric/builder.rs:26-42
    o->visual_kind = SA_VIS_EUCLIDEAN;
    o->visual_threshold = 3.4028234663852886e38f;
    o->visual_max_observations = 5;
    o->visual_min_votes = 1;
    o->visual_minimal_track_length = 3;
  }
}

const char* sa_tracker_last_error(const sa_tracker* t) { return t ? t->err.c_str() : g_err.c_str(); }
sa_engine* sa_tracker_engine(sa_tracker* t) { return t ? t->eng : nullptr; }

int sa_tracker_create(const sa_tracker_options* o, sa_tracker** out) {
  if (!o || !out) return tfail(nullptr, SA_ERR_BAD_ARG, "sa_tracker_create: null argument");
  *out = nullptr;
  if (o->struct_size != sizeof(sa_tracker_options)) return tfail(nullptr, SA_ERR_BAD_ARG, "sa_tracker_options.struct_size mismatch");
  if (o->history_length == 0) return tfail(nullptr, SA_ERR_BAD_ARG, "bbox_history must be > 0 (sort/simple_api.rs:51)");
  if (o->visual && (o->feature_len == 0 || o->visual_max_observations == 0))
    return tfail(nullptr, SA_ERR_BAD_ARG, "VisualSort needs feature_len and visual_max_observations");
  sa_tracker* t = new sa_tracker();
  t->o = *o;
  t->cons_delta.assign(o->constraint_epoch_delta, o->constraint_epoch_delta + o->n_constraints);
  t->cons_dist.assign(o->constraint_max_dist, o->constraint_max_dist + o->n_constraints);
  t->o.constraint_epoch_delta = t->cons_delta.data();
  t->o.constraint_max_dist = t->cons_dist.data();
  t->waste_counter = o->auto_waste_periodicity;
  sa_config c;
  sa_config_default(&c);
  c.device = o->device;
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
  c.constraint_epoch_delta = t->cons_delta.data();
  c.constraint_max_dist = t->cons_dist.data();
  c.kf_position_weight = o->kalman_position_weight;
  c.kf_velocity_weight = o->kalman_velocity_weight;
  c.visual_minimal_quality_collect = o->visual_minimal_quality_collect;
  c.visual_minimal_own_area_percentage_collect = o->visual_minimal_own_area_percentage_collect;
  int rc = sa_engine_create(&c, &t->eng);
  if (rc != SA_OK) {
    tfail(nullptr, rc, "sa_engine_create: %s", sa_last_error(nullptr));
    delete t;
    return rc;
  }
  *out = t;
  return SA_OK;
}

void sa_tracker_destroy(sa_tracker* t) {
  if (!t) return;
  if (t->eng) sa_engine_destroy(t->eng);
  delete t;
}

int sa_tracker_predict(sa_tracker* t, uint64_t scene_id, uint32_t n, const sa_observation* obs, sa_sort_track* out) {
  if (!t || (n && (!obs || !out))) return tfail(t, SA_ERR_BAD_ARG, "sa_tracker_predict: null argument");
  const sa_observation* op = obs;
  sa_sort_track* outp = out;
  return predict_scenes(t, 1, &scene_id, &n, &op, &outp);
}

int sa_tracker_predict_batch(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts,
                             const sa_observation* const* obs, sa_sort_track* const* out) {
  if (!t || (n_scenes && (!scene_ids || !counts || !obs || !out))) return tfail(t, SA_ERR_BAD_ARG, "sa_tracker_predict_batch: null argument");
  return predict_scenes(t, n_scenes, scene_ids, counts, obs, out);
}

int sa_tracker_idle_tracks(sa_tracker* t, uint64_t scene_id, sa_sort_track* out, uint32_t cap, uint32_t* out_n) {
  if (!t || !out_n) return tfail(t, SA_ERR_BAD_ARG, "sa_tracker_idle_tracks: null argument");
  flush_pending(t);
  // IdleLookup  sort.rs:213-228: same scene and last_updated_epoch != current epoch
  uint32_t n = 0;
  for (const auto* m : {&t->by_scene, &t->evicted}) {   // (evicted tracks are idle tracks like any other until they are wasted)
    auto it = m->find(scene_id);
    if (it == m->end()) continue;
    for (const Track* trp : it->second) {
      const Track& tr = *trp;
      if (tr.epoch != current_epoch(t, scene_id)) {
        if (out && n < cap) out[n] = to_sort_track(t->o, tr);
        ++n;
      }
    }
  }
  *out_n = n;
  return SA_OK;
}

int sa_tracker_skip_epochs(sa_tracker* t, uint64_t scene_id, uint64_t n) {
  if (!t) return SA_ERR_BAD_ARG;
  flush_pending(t);
  t->epochs[scene_id] += n;  // skip_epochs_for_scene  epoch_db.rs:11-20
  return auto_waste(t);      // tracker_api.rs:48-51
}

int sa_tracker_current_epoch(sa_tracker* t, uint64_t scene_id, uint64_t* out) {
  if (!t || !out) return SA_ERR_BAD_ARG;
  *out = current_epoch(t, scene_id);
  return SA_OK;
}

int sa_tracker_wasted(sa_tracker* t, sa_sort_track* out, uint32_t cap, uint32_t* out_n) {
  if (!t || !out_n) return tfail(t, SA_ERR_BAD_ARG, "sa_tracker_wasted: null argument");
  flush_pending(t);
  int rc = auto_waste(t);
  if (rc != SA_OK) return rc;
  std::sort(t->wasted_store.begin(), t->wasted_store.end(), [](const Track& a, const Track& b) { return a.id < b.id; });
  uint32_t n = (uint32_t)t->wasted_store.size();
  for (uint32_t i = 0; i < n && i < cap && out; ++i) out[i] = to_sort_track(t->o, t->wasted_store[i]);
  *out_n = n;
  if (out && cap >= n) t->wasted_store.clear();  // fetch_tracks removes them from the wasted store
  return SA_OK;
}

int sa_tracker_clear_wasted(sa_tracker* t) {
  if (!t) return SA_ERR_BAD_ARG;
  t->wasted_store.clear();
  return SA_OK;
}

int sa_tracker_active_tracks(sa_tracker* t, uint64_t* out_n) {
  if (!t || !out_n) return SA_ERR_BAD_ARG;
  *out_n = t->store.size();
  return SA_OK;
}

int sa_tracker_track_state(sa_tracker* t, uint64_t track_id, float* mean10, float* cov100) {
  if (!t) return SA_ERR_BAD_ARG;
  auto it = t->store.find(track_id);
  if (it == t->store.end()) return tfail(t, SA_ERR_NOT_FOUND, "unknown track id %llu", (unsigned long long)track_id);
  if (t->o.device_upkeep) {  // the state lives on the device
    int rc = sa_tracks_get_state(t->eng, it->second.scene, track_id, mean10, cov100, nullptr, nullptr, nullptr);
    return rc == SA_OK ? SA_OK : tfail(t, rc, "sa_tracks_get_state: %s", sa_last_error(t->eng));
  }
  if (mean10) std::memcpy(mean10, it->second.kf.mean, sizeof it->second.kf.mean);
  if (cov100) std::memcpy(cov100, it->second.kf.cov, sizeof it->second.kf.cov);
  return SA_OK;
}

int sa_tracker_track_info(sa_tracker* t, uint64_t track_id, uint64_t out4[4]) {
  if (!t || !out4) return SA_ERR_BAD_ARG;
  flush_pending(t);
  auto it = t->store.find(track_id);
  if (it == t->store.end()) return tfail(t, SA_ERR_NOT_FOUND, "unknown track id %llu", (unsigned long long)track_id);
  const Track& tr = it->second;
  out4[0] = tr.feat_count;
  out4[1] = t->o.visual ? tr.obs.size() : 1;
  out4[2] = tr.boxes.size();
  out4[3] = tr.length;
  return SA_OK;
}

}  // extern "C"
