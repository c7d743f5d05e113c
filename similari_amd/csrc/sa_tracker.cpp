// sa_tracker.cpp — host facade with the reference's tracker surface (include/similari_tracker.h) on top of the
// association engine.  Association (cost matrices + votes) = sa_associate_batch on the GPU; everything in this
// file is the O(N) bookkeeping the reference also does on the host around that call:
//   Sort::predict_with_scene          src/trackers/sort/simple_api.rs:110-196
//   VisualSort::predict_with_scene    src/trackers/visual_sort/simple_api.rs:99-230
//   Batch*::predict / voting_thread   src/trackers/sort/batch_api.rs:68-153,222-290
//   PredictionBatchResult             src/trackers/batch.rs:24-38
//   SortMetric / VisualMetric::optimize (Kalman step, history, feature bank)   sort/metric.rs:79-105,
//                                     visual_sort/metric.rs:129-154,297-374
//   TrackerAPI (epochs, waste)        src/trackers/tracker_api.rs, epoch_db.rs
// The Kalman step is written against the filter's structure (motion = I + shift, update matrix = [I 0]); skipping
// the multiplications by the constant 0/1 entries leaves every f32 result unchanged (kalman_2d_box.rs:58-148).
//
// A request set of several scenes (Batch*::predict) is worked on scene by scene by a small pool of threads (sa_pool.h) — the
// reference's voting_shards (sort/batch_api.rs:197-207) — and on the device by ONE set of launches: the association of every scene
// (grid.z = scene), one Kalman dispatch and one feature-bank dispatch for all of them (sa_batch_run_apply).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <new>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/similari_tracker.h"
#include "sa_kalman.h"
#include "sa_pool.h"

extern "C" __attribute__((visibility("hidden"))) bool sa_in_pinned_host_block(const void* p, size_t bytes);   // sa_engine.hip: inside a block from sa_host_alloc?

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
// history length, sort.rs:160-176; the two are always pushed together): ONE fixed ring of pairs, stored right behind the track's record
// in its slab cell (TrackSlab) — no allocation per frame, none per track.
struct BoxPair { sa_box observed, predicted; };
struct Ring {
  BoxPair* v = nullptr;
  uint32_t cap = 0, head = 0, count = 0;
  void push(const sa_box& observed, const sa_box& predicted) {
    uint32_t at;
    if (count < cap) { at = head + count++; at = at >= cap ? at - cap : at; }
    else { at = head; head = head + 1 == cap ? 0 : head + 1; }
    v[at].observed = observed;
    v[at].predicted = predicted;
  }
  const BoxPair& back() const { const uint32_t at = head + count - 1; return v[at >= cap ? at - cap : at]; }
  uint32_t size() const { return count; }
};

struct Track {  // (what a frame's bookkeeping touches first)
  uint64_t id = 0, scene = 0, epoch = 0, length = 0;
  bool has_custom = false;
  int64_t custom = 0;
  int32_t voting = -1;  // VisualAttributes::voting_type: None
  bool has_state = false;
  bool in_engine = true;   // false: evicted from the engine's table (it can never match again) but not wasted yet
  uint32_t feat_count = 0;
  Ring boxes;
  std::vector<Obs> obs;
  std::unique_ptr<KF> kf;  // host upkeep only (440 bytes the device owns under device upkeep)
  Track* next_free = nullptr;
};

// Where a scene's tracks live: cells of (record + history ring) carved from chunks of 256, recycled through a free list when a track is
// wasted.  A tracker at 5 % churn starts 1 600 tracks per predict() of 64 x 500 objects; from the process heap that was three allocations
// each and — measured, MI355X box — ~300 page faults per call with the heap's growth calls (mprotect: a TLB shoot-down on every thread
// of the process) in the middle of the scenes' parallel jobs, which ran 3.6 x slower than the same jobs one after the other.
struct TrackSlab {
  std::vector<char*> chunks;
  size_t cell = 0, left = 0;   // bytes per cell ; cells left in the last chunk
  uint32_t ring_cap = 0;
  Track* free_list = nullptr;
  static constexpr size_t kCells = 256;
  Track* get(uint32_t history) {
    if (!cell) { ring_cap = history; cell = ((sizeof(Track) + 15) & ~(size_t)15) + (size_t)history * sizeof(BoxPair); cell = (cell + 63) & ~(size_t)63; }
    char* at;
    if (free_list) { at = (char*)free_list; free_list = free_list->next_free; }
    else {
      if (!left) { chunks.push_back((char*)::operator new(cell * kCells, std::align_val_t(64))); left = kCells; }
      at = chunks.back() + (kCells - left) * cell;
      --left;
    }
    Track* tr = new (at) Track();
    tr->boxes.v = (BoxPair*)(at + ((sizeof(Track) + 15) & ~(size_t)15));
    tr->boxes.cap = ring_cap;
    return tr;
  }
  void put(Track* tr) {
    tr->~Track();
    tr->next_free = free_list;   // (the cell is raw memory again: only this link lives in it)
    free_list = tr;
  }
  void release() {
    for (char* c : chunks) ::operator delete(c, std::align_val_t(64));
    chunks.clear();
    free_list = nullptr;
    left = 0;
  }
};

// One scene's share of the store.  Scenes never share a track (compatible() is false across scene ids, sort.rs:251), so everything a
// frame's bookkeeping touches belongs to exactly one scene — which is what lets the scenes of a request set be worked on side by side.
// (aligned to two cache lines: neighbouring scenes are worked on by different threads, and the vectors' headers below are written — a
// push_back per track that starts — while the neighbour's are read)
struct alignas(128) SceneState {
  uint64_t id = 0, epoch = 0;   // EpochDb: the scene's current epoch (0 = never seen)
  // the scene's tracks in the order of the engine's table for that scene: rows are appended in creation order (ids ascend) and removals
  // close the gaps on both sides, so the winner the engine reports as a COLUMN (sa_batch_results) is rows[column] — no lookup by id on
  // the per-candidate path
  std::vector<Track*> rows;
  // Eviction.  A track whose last update lies more than max_idle_epochs behind its scene's epoch fails compatible() (sort.rs:250-270) for
  // every later frame — epochs only grow — but the reference keeps it in the store until the next auto_waste (every 100th predict by
  // default), and so would the engine's table: at 5 % churn a 1000-object VisualSORT loop associates against 6 700 rows instead of
  // 1 200.  The facade therefore takes such tracks out of the ENGINE's table as soon as they are 64 and a sixteenth of it (sa_tracks_remove:
  // one gather launch, queued without a drain), and keeps them here — idle_tracks / wasted see them as before.  (What is gone with the
  // row is the Kalman state of a device-upkeep tracker: sa_tracker_track_state answers SA_ERR_NOT_FOUND for an evicted track.)
  std::vector<uint64_t> row_epoch;   // last_updated_epoch of `rows`, in the same order (the scan's input)
  std::vector<Track*> evicted;       // out of the engine's table, not wasted yet
  TrackSlab slab;
};

struct ResultState;

// The tracks of a request set on their way through a result handle lie in ONE block per set, and the blocks go round: a handle that is
// freed gives its block back, the next _begin takes it (64 scenes x 500 tracks are 2.3 MB: from the heap that was an mmap, its page
// faults and a memset per call — the call returned after 204 us where the synchronous one has launched after 135).  Shared by the tracker
// and its handles: whichever goes last frees it.
struct TrackBlocks {
  std::mutex mu;
  struct Block { std::unique_ptr<sa_sort_track[]> p; size_t cap = 0; };
  std::vector<Block> spare;
  Block take(size_t n) {
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < spare.size(); ++i)
        if (spare[i].cap >= n) { Block b = std::move(spare[i]); spare.erase(spare.begin() + (long)i); return b; }
      if (!spare.empty()) spare.pop_back();   // (too small for this tracker's sets: let it go)
    }
    Block b;
    b.cap = n + n / 4 + 64;
    b.p.reset(new sa_sort_track[b.cap]);   // (default-initialised: every track of a delivered scene is written before it is handed out)
    return b;
  }
  void give(Block&& b) {
    if (!b.p) return;
    std::lock_guard<std::mutex> lk(mu);
    if (spare.size() < 4) spare.push_back(std::move(b));
  }
};

}  // namespace

// PredictionBatchResult  trackers/batch.rs:19-38
struct sa_batch_result {
  std::shared_ptr<ResultState> st;
  // the handle of a device GROUP (sa_tracker_options.n_devices > 1): one handle per shard that received scenes; a scene is delivered
  // from whichever shard has one ready
  std::vector<sa_batch_result*> kids;
  uint32_t kid_total = 0, kid_taken = 0, kid_next = 0;
  int spin_us = 500;
};

struct sa_tracker {
  sa_tracker_options o{};
  std::vector<uint64_t> cons_delta;
  std::vector<float> cons_dist;
  sa_engine* eng = nullptr;
  std::string err;
  uint64_t track_id = 0;
  // Device group (n_devices > 1): this object owns one complete tracker per device and only routes — shard = scene_id % shards.size(), for
  // good; it has no engine, no scenes and no pool of its own, and keeps the two things the shards must share: the id counter and the
  // auto-waste counter.  A shard draws the ids of a call's scenes from forced_base (one base per scene of ITS share, set by the group for
  // the duration of the call) instead of its own counter.
  std::vector<sa_tracker*> shards;
  const uint64_t* forced_base = nullptr;
  std::map<uint64_t, SceneState> scenes;   // node-based: a SceneState's address is stable
  uint64_t n_active = 0;                   // tracks in the main store (sum of active_shard_stats)
  std::vector<sa_sort_track> wasted_store;   // (what wasted() hands out of a wasted track: its SortTrack)
  uint32_t waste_counter = 0;
  bool evict_wave = false;      // the previous request set evicted: this set's scenes join from 16 expired rows on (scan_expired)
  // predict()'s per-scene work arrays, kept between calls: a frame allocates nothing once the arrays have grown to its size.  Everything
  // the work AFTER the launches reads of the caller's observations is copied here (boxes, custom ids): the reference takes its request by
  // value, and so a caller of sa_tracker_predict_batch_begin may reuse its arrays as soon as that call has returned.
  struct alignas(128) SceneScratch {   // (one thread per scene: see SceneState)
    SceneState* st = nullptr;
    uint32_t n = 0, slot = 0, n_new = 0;
    uint64_t epoch = 0, id_base = 0;
    int rc = SA_OK;
    std::string err;
    double job_us[4] = {0, 0, 0, 0};   // (SA_TRACKER_TRACE)
    std::vector<sa_box> cboxes, oboxes, dev_pred;  // the candidates' boxes after their own Kalman no-op step ; as observed ; the device's predicted boxes
    std::vector<int64_t> ccustom;
    std::vector<uint8_t> chas_custom;
    std::vector<float> cq, cown, shares;
    std::vector<const float*> cfeat;           // one pointer per detection: the engine gathers the rows itself ...
    std::vector<uint8_t> cpres, votes, merged;
    std::vector<uint64_t> winners, tids, new_ids;
    std::vector<int32_t> wcols;
    std::vector<uint64_t> evict_ids;           // rows of the scene's table that no later frame can match (scan_expired)
    uint32_t n_expired = 0;                    // how many such rows the last scan counted (evicted or not)
    int evict_rc = SA_OK;                      // ... staged in the engine by the scene's own job (sa_tracks_remove_stage)
    std::vector<Track*> trps;
    sa_detections det{};
    uint8_t contiguous = 0;                    // ... unless they already ARE one N x D block (then: no gather at all)
  };
  std::vector<SceneScratch> scratch[2];     // two sets, used in turn: the set of the previous predict() may still hold deferred work
  int cur_set = 0;
  std::vector<uint64_t> sc_touched;
  // Deferred bookkeeping (device upkeep queued behind the association: the fused path).  What a predict() RETURNS needs, per continued
  // track, one cache line of its record (id, length, epoch, custom id, vote); the rest of the reference's merge — the history deques
  // (sort.rs:160-176) and the observation policy (visual_sort/metric.rs:129-154) — only has to be in place before anything READS it:
  // the next predict()'s own merges, idle_tracks, wasted, track_info.  It is therefore run by the NEXT call on the tracker, and a
  // predict() runs it while its own association is on the device (the host would wait there anyway); every other entry point runs
  // it first.  All it reads lies in the scratch set of the frame that left it behind.
  bool pending = false;
  int pending_set = 0;
  uint32_t pending_scenes = 0;
  // the per-scene jobs of a request set (created with the first set of more than one scene)
  std::unique_ptr<SaPool> pool;
  // sa_tracker_predict_batch_begin: the work behind the launches (waiting, merges, results) runs on this thread, so that the caller gets
  // its handle back while the GPU is still busy — the reference's voting threads.  One request set at a time: every entry point first
  // waits for the set in flight (the reference's "busy monitor", sort/batch_api.rs:233-241).
  std::shared_ptr<TrackBlocks> blocks = std::make_shared<TrackBlocks>();   // (the result handles' storage)
  std::thread driver;
  std::mutex dmu;
  std::condition_variable dcv;
  bool d_work = false, d_stop = false;
  std::atomic<bool> d_busy{false};
  std::atomic<uint32_t> d_posted{0};   // request sets handed to the driver so far (the driver looks at this word for a while before it sleeps)
  struct Flight {   // the request set between its launches and its last result
    uint32_t n_scenes = 0;
    int set = 0;
    std::vector<sa_sort_track*> out;
    std::shared_ptr<ResultState> res;   // null: a synchronous predict (the caller's arrays)
  } flight;
};

namespace {

struct ResultState {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<uint64_t> scene_ids;
  std::shared_ptr<TrackBlocks> blocks;     // where `store` came from and goes back to
  TrackBlocks::Block store;                // the set's tracks, scene after scene
  std::vector<size_t> off;                 // [scenes + 1] a scene's first track in `store`
  ~ResultState() { if (blocks) blocks->give(std::move(store)); }
  std::deque<uint32_t> ready_q;   // scenes whose tracks are final, in the order they became so
  std::atomic<uint32_t> n_ready{0};   // scenes pushed so far | bit 31: finished (what get() looks at before it takes the lock and sleeps)
  std::atomic<uint32_t> taken{0};   // (written under the lock)
  int rc = SA_OK;                 // first error of the request set
  std::string err;
  bool finished = false;
};

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
  tr.boxes.push(observed, predicted);
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
  // Sort: always Positional (sort/simple_api.rs:260) ; VisualSort: attrs.voting_type.unwrap_or(Positional)
  s.voting_type = (o.visual && tr.voting >= 0) ? tr.voting : SA_VOTE_POSITIONAL;
  s.has_custom_object_id = tr.has_custom ? 1 : 0;
  s.custom_object_id = tr.custom;
  return s;
}

sa_sort_track to_sort_track(const sa_tracker_options& o, const Track& tr) {
  const BoxPair& last = tr.boxes.back();
  return to_sort_track_with(o, tr, last.observed, last.predicted);
}

SceneState& scene_of(sa_tracker* t, uint64_t scene) {
  SceneState& s = t->scenes[scene];
  s.id = scene;
  return s;
}
uint64_t current_epoch(sa_tracker* t, uint64_t scene) {
  auto it = t->scenes.find(scene);
  return it == t->scenes.end() ? 0 : it->second.epoch;
}
// a stored track by id: rows ascend by id (binary search), evicted ones are few
Track* find_track(sa_tracker* t, uint64_t id, SceneState** where = nullptr) {
  for (auto& kv : t->scenes) {
    SceneState& S = kv.second;
    auto it = std::lower_bound(S.rows.begin(), S.rows.end(), id, [](const Track* a, uint64_t v) { return a->id < v; });
    Track* hit = (it != S.rows.end() && (*it)->id == id) ? *it : nullptr;
    if (!hit)   // (tables whose ids ever arrived out of order: none of the facade's own, but cheap to be safe)
      for (Track* tr : S.rows)
        if (tr->id == id) { hit = tr; break; }
    if (!hit)
      for (Track* tr : S.evicted)
        if (tr->id == id) { hit = tr; break; }
    if (hit) { if (where) *where = &S; return hit; }
  }
  return nullptr;
}

void run_jobs(sa_tracker* t, uint32_t n, const std::function<void(uint32_t)>& fn) {
  if (n > 1 && !t->pool) {
    // workers: n > 0 = n threads bound to the CPUs next to the caller's (sa_pool.h), n < 0 = |n| threads left to the scheduler, 0 = the facade's choice
    const int32_t ow = t->o.workers;
    uint32_t w = ow ? (uint32_t)(ow < 0 ? -ow : ow) : std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 8u));
    t->pool.reset(new SaPool(w > 1 ? w - 1 : 0, ow >= 0, t->o.spin_us));   // (the calling thread is a worker too; spin_us < 0: the pool's default)
  }
  if (n > 1 && t->pool) t->pool->run(n, fn);
  else
    for (uint32_t i = 0; i < n; ++i) fn(i);
}

// get_main_store_wasted + auto_waste  tracker_api.rs:68-88 ; baked(): epoch_db.rs:52-66
int auto_waste(sa_tracker* t) {
  for (auto& kv : t->scenes) {
    SceneState& S = kv.second;
    std::vector<uint64_t> gone, resident;   // resident: those the engine's table still holds (the others were evicted from it earlier)
    for (const auto* list : {&S.rows, &S.evicted})
      for (const Track* tr : *list)
        if (tr->epoch + t->o.max_idle_epochs < S.epoch) { gone.push_back(tr->id); if (tr->in_engine) resident.push_back(tr->id); }
    if (gone.empty()) continue;
    std::sort(gone.begin(), gone.end());
    std::sort(resident.begin(), resident.end());
    int rc = resident.empty() ? SA_OK : sa_tracks_remove(t->eng, S.id, (uint32_t)resident.size(), resident.data());
    if (rc != SA_OK) return tfail(t, rc, "sa_tracks_remove: %s", sa_last_error(t->eng));
    auto is_gone = [&](const Track* tr) { return std::binary_search(gone.begin(), gone.end(), tr->id); };
    std::vector<Track*> dead;
    size_t w = 0;
    for (size_t r = 0; r < S.rows.size(); ++r) {
      if (is_gone(S.rows[r])) dead.push_back(S.rows[r]);
      else { S.rows[w] = S.rows[r]; S.row_epoch[w] = S.row_epoch[r]; ++w; }
    }
    S.rows.resize(w);
    S.row_epoch.resize(w);
    for (Track* tr : S.evicted)
      if (is_gone(tr)) dead.push_back(tr);
    S.evicted.erase(std::remove_if(S.evicted.begin(), S.evicted.end(), is_gone), S.evicted.end());
    std::sort(dead.begin(), dead.end(), [](const Track* a, const Track* b) { return a->id < b->id; });
    for (Track* tr : dead) {
      t->wasted_store.push_back(to_sort_track(t->o, *tr));
      S.slab.put(tr);
      --t->n_active;
    }
  }
  return SA_OK;
}

// Pushes the rows of `trs` (tracks of one scene that were created or merged this frame) to the engine.
int sync_engine(sa_tracker* t, uint64_t scene, const std::vector<Track*>& trs) {
  if (trs.empty()) return SA_OK;
  const uint32_t n = (uint32_t)trs.size(), K = t->o.visual ? t->o.visual_max_observations : 1, D = t->o.feature_len;
  std::vector<sa_box> boxes(n);
  std::vector<uint64_t> epochs(n), ids(n);
  std::vector<float> mean(n * 5), cov(n * 25), feats;
  std::vector<uint8_t> present;
  if (t->o.visual) { feats.assign((size_t)n * K * D, 0.0f); present.assign((size_t)n * K, 0); }
  for (uint32_t i = 0; i < n; ++i) {
    const Track& tr = *trs[i];
    ids[i] = tr.id;
    boxes[i] = tr.boxes.back().predicted;
    epochs[i] = tr.epoch;
    for (int a = 0; a < 5; ++a) {
      mean[i * 5 + a] = tr.kf->mean[a];
      for (int b = 0; b < 5; ++b) cov[i * 25 + a * 5 + b] = tr.kf->cov[a * 10 + b];
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
  tr.boxes.push(W.cboxes[i], W.dev_pred[i]);
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

// Runs what the previous predict() deferred (sa_tracker::pending), one job per scene of that set.  Idempotent; every entry point that
// reads a track's history or observations calls it first.
void flush_pending(sa_tracker* t) {
  if (!t->pending) return;
  t->pending = false;
  const std::vector<sa_tracker::SceneScratch>& ss = t->scratch[t->pending_set];
  run_jobs(t, t->pending_scenes, [&](uint32_t s) {
    const sa_tracker::SceneScratch& W = ss[s];
    for (uint32_t i = 0; i < W.n; ++i)
      if (W.merged[i]) merge_heavy(t->o, *W.trps[i], W, i);
  });
}

// The set in flight (sa_tracker_predict_batch_begin) has delivered its last scene: every entry point waits for that first.
void wait_outstanding(sa_tracker* t) {
  if (!t->d_busy.load(std::memory_order_acquire)) return;
  std::unique_lock<std::mutex> lk(t->dmu);
  t->dcv.wait(lk, [&] { return !t->d_busy.load(std::memory_order_acquire); });
}

using clk = std::chrono::steady_clock;
#include <sys/resource.h>
inline long minor_faults() { struct rusage ru; getrusage(RUSAGE_SELF, &ru); return ru.ru_minflt; }
inline double us_between(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

// One scene's observations into its work arrays: validation (the reference's assert!s), the candidates' boxes, the optional arrays of a
// VisualSORT observation, whether the feature rows form one block.  Reads the caller's memory, writes nothing but W.
int assemble_scene(const sa_tracker_options& o, sa_tracker::SceneScratch& W, uint64_t scene_id, uint32_t n, const sa_observation* obs) {
  const uint32_t D = o.feature_len;
  W.n = n;
  W.rc = SA_OK;
  for (uint32_t i = 0; i < n; ++i) {
    const sa_box& bb = obs[i].bbox;
    if (!(bb.aspect > 0.0f) || !(bb.height > 0.0f) || !(bb.confidence >= 0.0f && bb.confidence <= 1.0f)) {
      char buf[128];
      snprintf(buf, sizeof buf, "observation %u of scene %llu: bad box", i, (unsigned long long)scene_id);
      W.err = buf;
      return W.rc = SA_ERR_BAD_ARG;
    }
  }
  W.cboxes.resize(n);
  W.oboxes.resize(n);
  W.ccustom.resize(n);
  W.chas_custom.resize(n);
  if (o.visual) { W.cfeat.resize(n); W.cq.resize(n); W.cown.resize(n); W.cpres.resize(n); }
  // The throw-away candidate track of a detection (simple_api.rs:125-145) is never materialised: its box goes into the request, the
  // rest of it (custom id, feature pointer) is kept beside it for the moment a track takes it over.
  // The candidate's own Kalman step (initiate -> predict -> update with the box it was initiated from, kalman_prediction.rs:13-32)
  // is the identity on the box: zero velocity, zero innovation, so the new mean is the observation plus (+-0) * gain.  All that
  // changes is TryFrom<KalmanState>: an angle of exactly 0.0 reads back as None (kalman.rs:82-86).  The 10 x 10 filter arithmetic
  // (about 1 us per detection on the host) is therefore only run for the candidates that become tracks and need the state;
  // the tests compare every box with the oracle, which does run the filter.
  const float* block = nullptr;  // where row 0 of an N x D block would lie, if the features form one
  bool one_block = o.visual && n > 0, all_present = true;
  const std::vector<float>& shares = W.shares;
  for (uint32_t i = 0; i < n; ++i) {
    const sa_observation& ob = obs[i];
    W.oboxes[i] = ob.bbox;
    sa_box& c = W.cboxes[i];
    c = ob.bbox;
    c.angle = ob.bbox.has_angle ? ob.bbox.angle : 0.0f;
    c.has_angle = (ob.bbox.has_angle && ob.bbox.angle != 0.0f) ? 1 : 0;
    c.reserved = 0;
    W.chas_custom[i] = ob.has_custom_object_id != 0;
    W.ccustom[i] = ob.custom_object_id;
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
  sa_detections& d = W.det;
  std::memset(&d, 0, sizeof d);
  d.n = n;
  d.boxes = W.cboxes.data();
  W.contiguous = 0;
  if (o.visual) {
    d.feat_present = W.cpres.data();
    d.feat_quality = W.cq.data();
    d.own_area = W.cown.data();
    // The observations' features as ONE block (a producer that writes its N x D output contiguously — a ReID head's output buffer,
    // host or device): handed over as such — read in place when the block is pinned (sa_host_alloc) or registered device memory
    // (sa_device_block_register), one memcpy otherwise — instead of one gather per row.  Rows of detections without a feature are
    // never dereferenced by the host (flagged absent).  Only when every row lies inside a block the engine knows: rows of
    // absent detections may otherwise be unmapped memory.
    if (one_block && block && all_present) {
      W.contiguous = 1;
      d.feats = block;
    }
  }
  return SA_OK;
}

// A candidate that starts a track (simple_api.rs:167-187).  Under device upkeep the filter state is born on the GPU and the feature
// vectors live in the device bank only.
Track* start_track(const sa_tracker_options& o, SceneState& S, const sa_tracker::SceneScratch& W, uint32_t i, uint64_t id, const float* feature) {
  Track* trp = S.slab.get(o.history_length);
  Track& tr = *trp;
  tr.id = id; tr.scene = S.id; tr.epoch = W.epoch;
  tr.has_custom = W.chas_custom[i] != 0; tr.custom = W.ccustom[i];
  tr.has_state = true;
  if (!o.device_upkeep) { bool hs = false; tr.kf.reset(new KF()); make_prediction(o.kalman_position_weight, o.kalman_velocity_weight, hs, *tr.kf, W.oboxes[i]); }
  tr.length = 0;
  update_history(o, tr, W.oboxes[i], W.cboxes[i]);
  if (o.visual) {
    const bool has_feat = W.cpres[i] != 0, has_own = W.cown[i] == W.cown[i];
    tr.obs.reserve(o.visual_max_observations + 1);
    tr.obs.emplace_back();                         // is_merge = false: the feature is kept as is
    Obs& nb = tr.obs.back();
    nb.quality = W.cq[i]; nb.has_own = has_own; nb.own = has_own ? W.cown[i] : 0.0f; nb.has_feat = has_feat;
    if (!o.device_upkeep && has_feat) nb.feat.assign(feature, feature + o.feature_len);
    tr.feat_count = has_feat ? 1 : 0;
  }
  return trp;
}

// the winner as a column of the table the engine voted against = a row of `rows` (checked; by id if the orders ever disagree)
Track* winner_row(SceneState& S, size_t rows_before, int32_t col, uint64_t dest, uint64_t epoch) {
  if (col >= 0 && (size_t)col < rows_before && S.rows[col]->id == dest) { S.row_epoch[col] = epoch; return S.rows[col]; }
  for (size_t r = 0; r < rows_before; ++r)
    if (S.rows[r]->id == dest) { S.row_epoch[r] = epoch; return S.rows[r]; }
  return nullptr;
}

// eviction (see SceneState::row_epoch): tracks of the set's scenes that no frame from now on can match leave the engine's table.
// The scan is per scene (it rides in the scene's assemble job); the removals of every scene of the set are ONE call — one gather launch
// per dozen scenes, no drain.
// wave: another scene of the tracker evicted in the previous request set — this scene joins from 16 expired rows on.  The gathers of a set's
// scenes are ONE launch (per dozen scenes) whatever their number, and a set without any eviction skips that launch and its commit
// altogether (7 us of host work + a dependent launch in front of the association): scenes that reach their 64 rows out of phase made
// nearly every set of a churned 8-scene loop pay for it; after one wave they evict together, every third set or so.  When a row leaves the
// engine's table changes no result (it can match nothing any more: the tests hold the facade against the oracle tracker, which never evicts).
void scan_expired(const sa_tracker_options& o, sa_tracker::SceneScratch& W, bool wave = false) {
  W.evict_ids.clear();
  const SceneState& S = *W.st;
  const uint64_t cur = W.epoch;
  size_t expired = 0;
  for (uint64_t ep : S.row_epoch) expired += ep + o.max_idle_epochs < cur ? 1u : 0u;
  // (a removal is a gather of the whole table; a row left in it costs the frames until auto_waste ~1/64 us each: it pays from about 64 rows on)
  W.n_expired = (uint32_t)expired;
  if (wave ? expired < 16 : (expired < 64 || expired * 16 < S.rows.size())) return;
  W.evict_ids.reserve(expired);
  for (size_t r = 0; r < S.rows.size(); ++r)
    if (S.row_epoch[r] + o.max_idle_epochs < cur) W.evict_ids.push_back(S.rows[r]->id);
}
// the facade's side of a scene's eviction, once the engine's table has let the rows go
void evict_commit(const sa_tracker_options& o, sa_tracker::SceneScratch& W) {
  if (W.evict_ids.empty()) return;
  SceneState& S = *W.st;
  const uint64_t cur = W.epoch;
  size_t w = 0;
  for (size_t r = 0; r < S.rows.size(); ++r) {
    if (S.row_epoch[r] + o.max_idle_epochs < cur) { S.rows[r]->in_engine = false; S.evicted.push_back(S.rows[r]); }
    else { S.rows[w] = S.rows[r]; S.row_epoch[w] = S.row_epoch[r]; ++w; }
  }
  S.rows.resize(w);
  S.row_epoch.resize(w);
  W.evict_ids.clear();
}
// staged: the scenes' jobs have staged their removals themselves (sa_tracks_remove_stage; W.evict_rc == SA_ERR_STATE: that scene has to go
// through the serial call); commit_rows: also do the facade's side here (else: the caller does it, in the scenes' next jobs)
int evict_expired(sa_tracker* t, std::vector<sa_tracker::SceneScratch>& ss, uint32_t n_scenes, bool staged, bool commit_rows) {
  std::vector<uint64_t> scene_ids;
  std::vector<uint32_t> counts;
  std::vector<const uint64_t*> lists;
  bool any = false;
  for (uint32_t s = 0; s < n_scenes; ++s) {
    if (ss[s].evict_ids.empty()) continue;
    any = true;
    if (staged && ss[s].evict_rc == SA_OK) continue;
    if (staged && ss[s].evict_rc != SA_ERR_STATE) {
      const int rc = tfail(t, ss[s].evict_rc, "sa_tracks_remove: %s", sa_last_error(t->eng));
      sa_tracks_remove_abort(t->eng);   // (the other scenes' staged removals with it: none of them is committed, every table stays as it is)
      return rc;
    }
    scene_ids.push_back(ss[s].st->id);
    counts.push_back((uint32_t)ss[s].evict_ids.size());
    lists.push_back(ss[s].evict_ids.data());
  }
  if (!any) return SA_OK;
  int rce = scene_ids.empty() ? sa_tracks_remove_commit(t->eng)
                              : sa_tracks_remove_many(t->eng, (uint32_t)scene_ids.size(), scene_ids.data(), counts.data(), lists.data());   // (commits what was staged as well)
  if (rce != SA_OK) { const int rc = tfail(t, rce, "sa_tracks_remove: %s", sa_last_error(t->eng)); sa_tracks_remove_abort(t->eng); return rc; }
  if (commit_rows)
    for (uint32_t s = 0; s < n_scenes; ++s) evict_commit(t->o, ss[s]);
  return SA_OK;
}

// ---- the part of a predict() BEHIND its launches (device upkeep queued behind the association) -----------------------------
// 1. the previous set's deferred bookkeeping, while this set's association is on the device
// 2. per scene, side by side: the winners (one wait for the association's completion event), ids, the tracks that start, the light half
//    of the merges
// 3. one wait for the set's Kalman dispatch, then per scene: the host side of the engine's table, the predicted boxes, the caller's tracks
// A scene whose tracks are final is handed to the result handle at once (PredictionBatchResult::get, trackers/batch.rs:29-33).
int complete_flight(sa_tracker* t) {
  const sa_tracker_options& o = t->o;
  sa_tracker::Flight& F = t->flight;
  std::vector<sa_tracker::SceneScratch>& ss = t->scratch[F.set];
  const uint32_t n_scenes = F.n_scenes;
  static const bool trace = getenv("SA_TRACKER_TRACE") != nullptr;
  const auto t0 = clk::now();
  const long flt0 = trace ? minor_faults() : 0;
  flush_pending(t);
  const auto t1 = clk::now();
  {  // ONE wait for the association's completion event, here: the jobs below then read the results without a trip into the runtime each
    const uint64_t* w0 = nullptr;
    int rc0 = sa_batch_results(t->eng, ss[0].slot, &w0, nullptr, nullptr);
    if (rc0 != SA_OK) {
      t->err = std::string("association: ") + sa_last_error(t->eng);
      if (F.res) { std::lock_guard<std::mutex> lk(F.res->mu); F.res->rc = rc0; F.res->err = t->err; F.res->finished = true; F.res->n_ready.fetch_or(0x80000000u, std::memory_order_release); F.res->cv.notify_all(); }
      return rc0;
    }
  }
  const auto t1w = clk::now();
  run_jobs(t, n_scenes, [&](uint32_t s) {
    sa_tracker::SceneScratch& W = ss[s];
    SceneState& S = *W.st;
    const uint32_t n = W.n;
    const auto j0 = trace ? clk::now() : clk::time_point();
    evict_commit(o, W);   // (rows the engine's table let go in front of this set's launches: the columns below refer to the table without them)
    uint32_t n_new = 0;
    const uint64_t* win = nullptr;
    const uint8_t* votes = nullptr;
    const int32_t* wcols = nullptr;
    W.n_new = 0;
    int rc = sa_batch_results(t->eng, W.slot, &win, &votes, &wcols);
    if (rc != SA_OK) { W.rc = rc; W.err = std::string("association: ") + sa_last_error(t->eng); return; }
    W.trps.resize(n);
    W.merged.resize(n);
    const size_t rows_before = S.rows.size();  // the table the engine voted against: columns refer to these rows
    uint64_t drawn = W.id_base;
    for (uint32_t i = 0; i < n; ++i) {
      const uint64_t dest = win[i];
      Track* trp;
      if (dest == 0) {
        // Batch*: an id per candidate (batch_api.rs:102-106); Sort / VisualSort: a counter over the tracks that start (simple_api.rs:165-187)
        const uint64_t id = o.batch_ids ? W.id_base + 1 + i : ++drawn;
        trp = start_track(o, S, W, i, id, nullptr);
        S.rows.push_back(trp);
        S.row_epoch.push_back(W.epoch);
        W.merged[i] = 0;
        ++n_new;
      } else {
        trp = winner_row(S, rows_before, wcols[i], dest, W.epoch);
        if (!trp) { W.rc = SA_ERR_STATE; W.err = "engine returned an unknown track id"; W.n_new = n_new; return; }
        Track& tr = *trp;
        // TrackAttributes::merge  sort.rs:272-276 / track_attributes.rs:210-215 (the rest of the merge — history, observation policy — is
        // deferred: merge_heavy / flush_pending)
        tr.epoch = W.epoch;
        tr.has_custom = W.chas_custom[i] != 0; tr.custom = W.ccustom[i];
        if (o.visual) tr.voting = votes[i];
        tr.length += 1;
        W.merged[i] = 1;
      }
      W.trps[i] = trp;
    }
    W.n_new = n_new;
    // what of the scene's results does not wait for the Kalman dispatch — the engine's side of the table (ids of the tracks that start,
    // the winners checked) and every field of the caller's tracks but the predicted box of a continued one — is done here, while that
    // dispatch is on the device; the second set of jobs, behind the wait, only fills the boxes in
    rc = sa_tracks_apply_collect_table(t->eng, W.slot, nullptr);
    if (rc != SA_OK) { W.rc = rc; W.err = std::string("sa_tracks_apply: ") + sa_last_error(t->eng); return; }
    sa_sort_track* out = F.out[s];
    for (uint32_t i = 0; i < n; ++i)
      out[i] = W.merged[i] ? to_sort_track_with(o, *W.trps[i], W.cboxes[i], W.cboxes[i]) : to_sort_track(o, *W.trps[i]);
    if (trace) W.job_us[0] = us_between(j0, clk::now());
  });
  const auto t2 = clk::now();
  int rc = SA_OK;
  std::string first_err;
  for (uint32_t s = 0; s < n_scenes && rc == SA_OK; ++s)
    if (ss[s].rc != SA_OK) { rc = ss[s].rc; first_err = ss[s].err; }
  if (rc == SA_OK) {
    rc = sa_tracks_apply_collect_begin(t->eng);
    if (rc != SA_OK) first_err = std::string("sa_tracks_apply: ") + sa_last_error(t->eng);
  }
  const auto t3 = clk::now();
  if (rc == SA_OK) {
    run_jobs(t, n_scenes, [&](uint32_t s) {
      sa_tracker::SceneScratch& W = ss[s];
      const uint32_t n = W.n;
      const auto j0 = trace ? clk::now() : clk::time_point();
      W.dev_pred.resize(n);
      int rcs = sa_tracks_apply_collect_slot(t->eng, W.slot, nullptr, W.dev_pred.data());
      if (rcs != SA_OK) { W.rc = rcs; W.err = std::string("sa_tracks_apply: ") + sa_last_error(t->eng); return; }
      sa_sort_track* out = F.out[s];
      for (uint32_t i = 0; i < n; ++i)
        if (W.merged[i]) out[i].predicted_bbox = W.dev_pred[i];
      if (F.res) {   // the scene's tracks are final: hand them over
        std::lock_guard<std::mutex> lk(F.res->mu);
        F.res->ready_q.push_back(s);
        F.res->n_ready.fetch_add(1, std::memory_order_release);
        F.res->cv.notify_all();
      }
      if (trace) W.job_us[1] = us_between(j0, clk::now());
    });
    for (uint32_t s = 0; s < n_scenes && rc == SA_OK; ++s)
      if (ss[s].rc != SA_OK) { rc = ss[s].rc; first_err = ss[s].err; }
  }
  if (rc == SA_OK) {
    rc = sa_tracks_apply_collect_end(t->eng);
    if (rc != SA_OK) first_err = std::string("sa_tracks_apply: ") + sa_last_error(t->eng);
  }
  uint64_t started = 0;
  for (uint32_t s = 0; s < n_scenes; ++s) started += ss[s].n_new;
  t->n_active += started;
  if (!o.batch_ids) t->track_id += started;
  if (rc == SA_OK) {
    t->pending = true;
    t->pending_set = F.set;
    t->pending_scenes = n_scenes;
  } else t->err = first_err;
  if (trace) {
    double mx[2] = {0, 0}, sum[2] = {0, 0};
    for (uint32_t s = 0; s < n_scenes; ++s)
      for (int k = 0; k < 2; ++k) { mx[k] = std::max(mx[k], ss[s].job_us[k]); sum[k] += ss[s].job_us[k]; }
    fprintf(stderr, "[sa_tracker] behind the launches: deferred %.1f  wait for the association %.1f  merges %.1f (jobs: longest %.1f, all %.1f)  wait for the Kalman "
            "dispatch %.1f  tables + results %.1f (jobs: longest %.1f, all %.1f) us  minor faults %.1f\n",
            us_between(t0, t1), us_between(t1, t1w), us_between(t1w, t2), mx[0], sum[0], us_between(t2, t3), us_between(t3, clk::now()), mx[1], sum[1],
            (double)(minor_faults() - flt0));
  }
  if (F.res) {
    std::lock_guard<std::mutex> lk(F.res->mu);
    F.res->rc = rc;
    F.res->err = first_err;
    F.res->finished = true;
    F.res->n_ready.fetch_or(0x80000000u, std::memory_order_release);
    F.res->cv.notify_all();
  }
  return rc;
}

// A tracker loop calls _begin back to back: the driver finds the next set within microseconds of the last one — it looks at the posting
// counter for a while (no futex round trip: ~50 us on this host when the thread had gone to sleep) before it waits on the condition variable.
void driver_loop(sa_tracker* t) {
  uint32_t seen = 0;
  for (;;) {
    {
      const int spin_us = t->o.spin_us < 0 ? 300 : t->o.spin_us;   // (sa_tracker_options.spin_us; 0: to sleep at once)
      const auto t0 = clk::now();
      for (uint32_t spin = 1; spin_us > 0 && t->d_posted.load(std::memory_order_acquire) == seen; ++spin) {
        SA_POOL_PAUSE();
        if ((spin & 255u) == 0 && clk::now() - t0 > std::chrono::microseconds(spin_us)) break;
      }
      std::unique_lock<std::mutex> lk(t->dmu);
      t->dcv.wait(lk, [&] { return t->d_work || t->d_stop; });
      if (t->d_stop) return;
      t->d_work = false;
      seen = t->d_posted.load(std::memory_order_acquire);
    }
    try {
      complete_flight(t);
    } catch (const std::exception& ex) {   // (a job of the set threw — out of memory, most likely: the handle reports it, nothing hangs)
      sa_tracker::Flight& F = t->flight;
      t->err = std::string("request set abandoned: ") + ex.what();
      if (F.res) {
        std::lock_guard<std::mutex> lk(F.res->mu);
        F.res->rc = SA_ERR_OOM;
        F.res->err = t->err;
        F.res->finished = true;
        F.res->n_ready.fetch_or(0x80000000u, std::memory_order_release);
        F.res->cv.notify_all();
      }
    }
    {
      std::lock_guard<std::mutex> lk(t->dmu);
      t->flight.res.reset();
      t->d_busy.store(false, std::memory_order_release);
    }
    t->dcv.notify_all();
  }
}

// ---- predict, the fused path: device upkeep with ids the device can draw itself ---------------------------------------------
// Up to the launches on the calling thread; what follows (complete_flight) on the calling thread too (res == nullptr: the caller's arrays
// are filled when this returns) or on the tracker's driver thread (sa_tracker_predict_batch_begin: the handle delivers scene by scene).
int predict_fused(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts, const sa_observation* const* obs,
                  sa_sort_track* const* out, const std::shared_ptr<ResultState>& res) {
  const sa_tracker_options& o = t->o;
  static const bool trace = getenv("SA_TRACKER_TRACE") != nullptr;
  const auto t_entry = clk::now();
  // (the other scratch set may hold the previous frame's deferred bookkeeping: it is run behind the launches, while this frame's association is on the device)
  const int set = t->pending ? (t->pending_set ^ 1) : t->cur_set;
  t->cur_set = set;
  if (t->scratch[set].size() < n_scenes) t->scratch[set].resize(n_scenes);
  std::vector<sa_tracker::SceneScratch>& ss = t->scratch[set];
  // exclusively_owned_areas_normalized_shares over the frame's observed boxes, when either own-area gate is armed
  // (visual_sort/simple_api.rs:111-127) — on the GPU (sa_own_areas).  A share the caller supplies takes precedence.
  for (uint32_t s = 0; s < n_scenes; ++s) {
    std::vector<float>& shares = ss[s].shares;
    shares.clear();
    const uint32_t n = counts[s];
    if (!(o.visual && n && o.visual_minimal_own_area_percentage_collect + o.visual_minimal_own_area_percentage_use > 0.0f)) continue;
    bool any_missing = false;
    for (uint32_t i = 0; i < n; ++i) any_missing = any_missing || obs[s][i].own_area != obs[s][i].own_area;
    if (!any_missing) continue;
    std::vector<sa_box> frame(n);
    for (uint32_t i = 0; i < n; ++i) {
      frame[i] = obs[s][i].bbox;
      if (!(frame[i].aspect > 0.0f) || !(frame[i].height > 0.0f) || !(frame[i].confidence >= 0.0f && frame[i].confidence <= 1.0f))
        return tfail(t, SA_ERR_BAD_ARG, "observation %u of scene %llu: bad box", i, (unsigned long long)scene_ids[s]);
    }
    shares.resize(n);
    int rc = sa_own_areas(t->eng, n, frame.data(), shares.data());
    if (rc != SA_OK) return tfail(t, rc, "%s", sa_last_error(t->eng));
  }
  uint64_t next = t->track_id;
  for (uint32_t s = 0; s < n_scenes; ++s) {
    SceneState& S = scene_of(t, scene_ids[s]);
    ss[s].st = &S;
    ss[s].epoch = S.epoch + 1; // next_epoch  epoch_db.rs:35-49 (committed once the request has passed validation)
    ss[s].id_base = t->forced_base ? t->forced_base[s] : next;      // (batch ids: one per candidate; a single scene: its own counter; a shard of a device group: the group's)
    next += counts[s];
  }
  const auto t_pre = clk::now();
  run_jobs(t, n_scenes, [&](uint32_t s) {
    const auto j0 = trace ? clk::now() : clk::time_point();
    sa_tracker::SceneScratch& W = ss[s];
    assemble_scene(o, W, scene_ids[s], counts[s], obs[s]);
    // _begin hands the request back to the caller as soon as the launches are queued, and the ingest pulls a PINNED block's rows over the
    // link asynchronously: a caller that refills its block for the next frame would race it.  The handle's contract ("the request may be
    // reused once _begin returns") therefore costs such a block its in-place read: its rows are copied into the staging arena during the
    // call like any other host rows.  (Registered DEVICE blocks stay in place: the Kalman dispatch moves their rows into engine memory
    // before anything of the frame is handed out.)
    if (res && W.contiguous && sa_in_pinned_host_block(W.det.feats, (size_t)W.n * o.feature_len * sizeof(float))) { W.contiguous = 0; W.det.feats = nullptr; }
    if (trace) W.job_us[2] = us_between(j0, clk::now());
  });
  const auto t_asm = clk::now();
  for (uint32_t s = 0; s < n_scenes; ++s)
    if (ss[s].rc != SA_OK) return tfail(t, ss[s].rc, "%s", ss[s].err.c_str());
  for (uint32_t s = 0; s < n_scenes; ++s) ss[s].st->epoch = ss[s].epoch;
  if (o.batch_ids) t->track_id = next;
  const auto t_built = clk::now();
  // ---- the hot path: foreign_track_distances + voting.winners, on the GPU, with the upkeep of every scene queued right BEHIND the
  // association (sa_batch_run_apply) — the ids of the tracks that start are a function of the winners alone (a counter, in candidate
  // order), so the device draws them itself
  int rc = sa_batch_begin(t->eng);
  for (uint32_t s = 0; s < n_scenes && rc == SA_OK; ++s)
    rc = sa_batch_add_deferred(t->eng, scene_ids[s], ss[s].epoch, &ss[s].det, (o.visual && !ss[s].contiguous) ? ss[s].cfeat.data() : nullptr, &ss[s].slot);
  if (rc == SA_OK) {
    // per scene, side by side: which of its table's rows no later frame can match (staged in the engine: sa_tracks_remove_stage), and the
    // copy of its detections into the staging arena
    const bool evict_wave = t->evict_wave && n_scenes > 1;
    run_jobs(t, n_scenes, [&](uint32_t s) {
      sa_tracker::SceneScratch& W = ss[s];
      scan_expired(o, W, evict_wave);
      W.evict_rc = W.evict_ids.empty() ? SA_OK : sa_tracks_remove_stage(t->eng, W.st->id, (uint32_t)W.evict_ids.size(), W.evict_ids.data());
      W.rc = sa_batch_fill(t->eng, W.slot);
    });
    // (the gathers of every staged scene: one launch per dozen scenes, in front of the request set's own launches; the facade's side of an
    // eviction follows in the scene's merge job)
    {   // (the next set is a wave when a scene evicted on its own account in this one, or is about to — 40 expired rows: at a few per
        // cent of churn its 64 are a set or two away — so that the scenes go together instead of one after the other; a wave set does not
        // start another)
      bool soon = false;
      for (uint32_t s = 0; s < n_scenes; ++s) soon = soon || !ss[s].evict_ids.empty() || ss[s].n_expired >= 40u;
      t->evict_wave = soon && !evict_wave;
    }
    const int rce = evict_expired(t, ss, n_scenes, true, false);
    for (uint32_t s = 0; s < n_scenes && rc == SA_OK; ++s) rc = ss[s].rc;
    if (rce != SA_OK || rc != SA_OK) {
      if (rce == SA_OK) for (uint32_t s = 0; s < n_scenes; ++s) evict_commit(o, ss[s]);
      flush_pending(t);
      return rce != SA_OK ? rce : tfail(t, rc, "association: %s", sa_last_error(t->eng));
    }
  }
  const auto t_staged = clk::now();
  if (rc == SA_OK) {
    std::vector<uint64_t> id_base(n_scenes);
    for (uint32_t s = 0; s < n_scenes; ++s) id_base[s] = ss[s].id_base;
    rc = sa_batch_run_apply(t->eng, id_base.data(), o.batch_ids ? 1 : 0);
  }
  if (rc != SA_OK) {
    for (uint32_t s = 0; s < n_scenes; ++s) evict_commit(o, ss[s]);
    flush_pending(t);
    return tfail(t, rc, "association: %s", sa_last_error(t->eng));
  }
  if (trace) {
    double mx = 0;
    for (uint32_t s = 0; s < n_scenes; ++s) mx = std::max(mx, ss[s].job_us[2]);
    fprintf(stderr, "[sa_tracker] up to the launches: before the jobs %.1f  assemble %.1f (longest job %.1f)  epochs %.1f  stage + evict %.1f  enqueue %.1f us\n",
            us_between(t_entry, t_pre), us_between(t_pre, t_asm), mx, us_between(t_asm, t_built), us_between(t_built, t_staged), us_between(t_staged, clk::now()));
  }
  sa_tracker::Flight& F = t->flight;
  F.n_scenes = n_scenes;
  F.set = set;
  F.out.assign(out, out + n_scenes);
  F.res = res;
  if (!res) return complete_flight(t);
  if (!t->driver.joinable()) {
    t->driver = std::thread(driver_loop, t);
#if defined(__linux__)
    // behind the pool's workers (SaPool::next_cpu: the same list and position they were placed by): the driver runs the calling thread's
    // share of the jobs behind the launches — the scenes' records should not cross a socket for it.  Without a pinned pool it is left
    // to the scheduler.
    const int cpu = t->pool ? t->pool->next_cpu() : -1;
    if (cpu >= 0) {
      cpu_set_t set;
      CPU_ZERO(&set);
      CPU_SET(cpu, &set);
      pthread_setaffinity_np(t->driver.native_handle(), sizeof set, &set);
    }
#endif
  }
  {
    std::lock_guard<std::mutex> lk(t->dmu);
    t->d_busy.store(true, std::memory_order_release);
    t->d_work = true;
    t->d_posted.fetch_add(1, std::memory_order_release);
  }
  t->dcv.notify_all();
  return SA_OK;
}

// ---- predict, the general path: host upkeep (sa_tracks_upsert of the refreshed rows), or device upkeep with Sort / VisualSort id rules
// over several scenes (one id per NEW track across scenes makes a scene's first id depend on the previous scenes' winners: two phases)
int predict_general(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts, const sa_observation* const* obs,
                    sa_sort_track* const* out) {
  const sa_tracker_options& o = t->o;
  const float pw = o.kalman_position_weight, vw = o.kalman_velocity_weight;
  const uint32_t D = o.feature_len;
  flush_pending(t);
  const int set = t->cur_set;
  if (t->scratch[set].size() < n_scenes) t->scratch[set].resize(n_scenes);
  std::vector<sa_tracker::SceneScratch>& ss = t->scratch[set];
  for (uint32_t s = 0; s < n_scenes; ++s) {
    const uint32_t n = counts[s];
    sa_tracker::SceneScratch& W = ss[s];
    for (uint32_t i = 0; i < n; ++i) {
      const sa_box& bb = obs[s][i].bbox;
      if (!(bb.aspect > 0.0f) || !(bb.height > 0.0f) || !(bb.confidence >= 0.0f && bb.confidence <= 1.0f))
        return tfail(t, SA_ERR_BAD_ARG, "observation %u of scene %llu: bad box", i, (unsigned long long)scene_ids[s]);
    }
    std::vector<float>& shares = W.shares;   // (see predict_fused)
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
    int rc = assemble_scene(o, W, scene_ids[s], n, obs[s]);
    if (rc != SA_OK) return tfail(t, rc, "%s", W.err.c_str());
  }
  for (uint32_t s = 0; s < n_scenes; ++s) {
    SceneState& S = scene_of(t, scene_ids[s]);
    ss[s].st = &S;
    ss[s].epoch = ++S.epoch;
    scan_expired(o, ss[s]);
  }
  int rc = evict_expired(t, ss, n_scenes, false, true);
  if (rc != SA_OK) return rc;
  rc = sa_batch_begin(t->eng);
  for (uint32_t s = 0; s < n_scenes && rc == SA_OK; ++s)
    rc = ss[s].contiguous ? sa_batch_add(t->eng, scene_ids[s], ss[s].epoch, &ss[s].det, &ss[s].slot)
                          : sa_batch_add_rows(t->eng, scene_ids[s], ss[s].epoch, &ss[s].det, o.visual ? ss[s].cfeat.data() : nullptr, &ss[s].slot);
  if (rc == SA_OK) rc = sa_batch_run(t->eng);
  if (rc == SA_OK) rc = sa_batch_sync(t->eng);
  for (uint32_t s = 0; s < n_scenes && rc == SA_OK; ++s) {
    sa_tracker::SceneScratch& W = ss[s];
    W.winners.resize(W.n); W.votes.resize(W.n); W.wcols.resize(W.n);
    rc = sa_batch_fetch(t->eng, W.slot, W.winners.data(), W.votes.data());
    if (rc == SA_OK) rc = sa_batch_fetch_cols(t->eng, W.slot, W.wcols.data());
  }
  if (rc != SA_OK) return tfail(t, rc, "association: %s", sa_last_error(t->eng));

  // ids first, scene by scene in the order the reference draws them; with device upkeep the Kalman step, table refresh and
  // feature-bank policy of every scene are QUEUED right away (sa_tracks_apply_begin) — they run while this thread does the
  // per-track bookkeeping that does not need their result; the predicted boxes are collected afterwards (sa_tracks_apply_end)
  for (uint32_t s = 0; s < n_scenes; ++s) {
    sa_tracker::SceneScratch& W = ss[s];
    const uint32_t n = W.n;
    W.tids.resize(n);
    W.new_ids.resize(n);
    if (t->forced_base) t->track_id = t->forced_base[s];   // (a shard of a device group: the group has drawn this scene's ids)
    for (uint32_t i = 0; i < n; ++i) {
      const uint64_t dest = W.winners[i];
      uint64_t drawn = 0;
      if (o.batch_ids) drawn = ++t->track_id;          // Batch*: an id per candidate (batch_api.rs:102-106)
      if (dest == 0) { W.tids[i] = o.batch_ids ? drawn : ++t->track_id; W.new_ids[i] = W.tids[i]; }
      else { W.tids[i] = dest; W.new_ids[i] = 0; }
    }
    if (o.device_upkeep) {
      rc = sa_tracks_apply_begin(t->eng, W.slot, W.new_ids.data());
      if (rc != SA_OK) return tfail(t, rc, "sa_tracks_apply: %s", sa_last_error(t->eng));
    }
  }
  std::vector<Track*> touched;
  for (uint32_t s = 0; s < n_scenes; ++s) {
    sa_tracker::SceneScratch& W = ss[s];
    SceneState& S = *W.st;
    const uint32_t n = W.n;
    touched.clear();
    W.trps.resize(n);
    W.merged.assign(n, 0);
    const size_t rows_before = S.rows.size();  // the table the engine voted against: columns refer to these rows
    for (uint32_t i = 0; i < n; ++i) {
      const sa_box& cbox = W.cboxes[i];
      const uint64_t dest = W.winners[i];
      Track* trp;
      if (dest == 0) {
        // winner == self or none: the candidate becomes a new track (simple_api.rs:167-187)
        trp = start_track(o, S, W, i, W.tids[i], o.visual ? W.cfeat[i] : nullptr);
        S.rows.push_back(trp);
        S.row_epoch.push_back(W.epoch);
        ++t->n_active;
      } else {
        trp = winner_row(S, rows_before, W.wcols[i], dest, W.epoch);
        if (!trp) return tfail(t, SA_ERR_STATE, "engine returned unknown track id %llu", (unsigned long long)dest);
        Track& tr = *trp;
        // TrackAttributes::merge  sort.rs:272-276 / track_attributes.rs:210-215
        tr.epoch = W.epoch;
        tr.has_custom = W.chas_custom[i] != 0; tr.custom = W.ccustom[i];
        if (o.visual) tr.voting = W.votes[i];
        // optimize(is_merge = true): Kalman predict + update with the candidate's box, history (device upkeep: once the boxes are back)
        if (!o.device_upkeep) update_history(o, tr, cbox, make_prediction(pw, vw, tr.has_state, *tr.kf, cbox));
        if (o.visual) {
          Obs nw;
          nw.quality = W.cq[i];
          nw.has_own = W.cown[i] == W.cown[i];
          nw.own = nw.has_own ? W.cown[i] : 0.0f;
          nw.has_feat = W.cpres[i] != 0;
          if (!feature_can_be_used(o, cbox, nw.quality, o.visual_minimal_quality_collect, nw.has_own, nw.own,
                                   o.visual_minimal_own_area_percentage_collect))
            nw.has_feat = false;
          if (!o.device_upkeep && nw.has_feat) nw.feat.assign(W.cfeat[i], W.cfeat[i] + D);
          // (with device upkeep: the bookkeeping only — the same policy moves the feature rows inside the device bank, sa_upkeep.hip)
          optimize_observations(tr.obs, std::move(nw), o.visual_max_observations);
          tr.feat_count = 0;
          for (auto& so : tr.obs) tr.feat_count += so.has_feat ? 1u : 0u;
        }
        W.merged[i] = 1;
      }
      W.trps[i] = trp;
      if (!o.device_upkeep) {
        touched.push_back(trp);
        out[s][i] = to_sort_track(o, *trp);
      }
    }
    if (o.device_upkeep) continue;
    rc = sync_engine(t, S.id, touched);
    if (rc != SA_OK) return rc;
  }
  if (o.device_upkeep)
    for (uint32_t s = 0; s < n_scenes; ++s) {
      sa_tracker::SceneScratch& W = ss[s];
      const uint32_t n = W.n;
      W.dev_pred.resize(n);
      rc = sa_tracks_apply_end(t->eng, W.slot, W.dev_pred.data());
      if (rc != SA_OK) return tfail(t, rc, "sa_tracks_apply: %s", sa_last_error(t->eng));
      for (uint32_t i = 0; i < n; ++i) {
        Track& tr = *W.trps[i];
        if (W.merged[i]) update_history(o, tr, W.cboxes[i], W.dev_pred[i]);
        out[s][i] = to_sort_track(o, tr);
      }
    }
  return SA_OK;
}

int predict_scenes(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts,
                   const sa_observation* const* obs, sa_sort_track* const* out, const std::shared_ptr<ResultState>& res = nullptr) {
  const sa_tracker_options& o = t->o;
  wait_outstanding(t);
  for (uint32_t s = 0; s < n_scenes; ++s) {
    if (counts[s] && (!obs[s] || !out[s])) return tfail(t, SA_ERR_BAD_ARG, "scene %llu: null observations / output", (unsigned long long)scene_ids[s]);
    for (uint32_t s2 = 0; s2 < s; ++s2)
      if (scene_ids[s] == scene_ids[s2]) return tfail(t, SA_ERR_BAD_ARG, "scene %llu appears twice in one batch", (unsigned long long)scene_ids[s]);
  }
  // auto waste (simple_api.rs:115-120)
  if (t->waste_counter == 0) {
    flush_pending(t);   // (wasted tracks are read out with their histories)
    int rc = auto_waste(t);
    if (rc != SA_OK) return rc;
    t->waste_counter = o.auto_waste_periodicity;
  } else t->waste_counter -= 1;
  if (!n_scenes) { flush_pending(t); return SA_OK; }
  // Device upkeep queued right behind the association (sa_batch_run_apply) whenever the ids of the tracks that start are a function of one
  // scene's winners alone: Batch* id rules (an id per candidate), or a single scene.
  const bool fused = o.device_upkeep && (o.batch_ids || n_scenes == 1);
  if (fused) return predict_fused(t, n_scenes, scene_ids, counts, obs, out, res);
  return predict_general(t, n_scenes, scene_ids, counts, obs, out);
}


}  // namespace
int sa_begin_into(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts, const sa_observation* const* obs,
                  sa_sort_track* const* caller_out, sa_batch_result** out_result);
namespace {
// ---- device group: Batch*::predict over several GPUs behind ONE tracker object (sort/batch_api.rs:157-207, 222-290) ------------------
inline uint32_t shard_of(const sa_tracker* t, uint64_t scene) { return (uint32_t)(scene % t->shards.size()); }
inline int group_fail(sa_tracker* t, const sa_tracker* c, int rc) { t->err = c->err; g_err = c->err; return rc; }
void group_auto_waste(sa_tracker* t, const sa_tracker* skip = nullptr) {
  for (sa_tracker* c : t->shards) {
    if (c == skip) continue;
    wait_outstanding(c);
    flush_pending(c);
    (void)auto_waste(c);
  }
}
// The request set split by shard (request order kept inside a shard), every shard's share begun one after the other on the calling
// thread — a shard's _begin returns once its launches are queued (its own pool assembles and stages its scenes, its own driver thread
// finishes them), so shard k + 1 prepares while shard k's kernels run — and ONE handle over the shards' handles.
int group_begin(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts, const sa_observation* const* obs,
                sa_batch_result** out_result, sa_sort_track* const* caller_out = nullptr) {
  const sa_tracker_options& o = t->o;
  const uint32_t ns = (uint32_t)t->shards.size();
  for (uint32_t s = 0; s < n_scenes; ++s) {
    if (counts[s] && !obs[s]) return tfail(t, SA_ERR_BAD_ARG, "scene %llu: null observations", (unsigned long long)scene_ids[s]);
    for (uint32_t s2 = 0; s2 < s; ++s2)
      if (scene_ids[s] == scene_ids[s2]) return tfail(t, SA_ERR_BAD_ARG, "scene %llu appears twice in one batch", (unsigned long long)scene_ids[s]);
  }
  // every box of the request is checked BEFORE any shard is begun (the reference asserts on its input before it touches the store): a bad
  // observation in one shard's scene must not leave the other shards a frame ahead
  for (uint32_t s = 0; s < n_scenes; ++s)
    for (uint32_t i = 0; i < counts[s]; ++i) {
      const sa_box& bb = obs[s][i].bbox;
      if (!(bb.aspect > 0.0f) || !(bb.height > 0.0f) || !(bb.confidence >= 0.0f && bb.confidence <= 1.0f))
        return tfail(t, SA_ERR_BAD_ARG, "observation %u of scene %llu: bad box", i, (unsigned long long)scene_ids[s]);
    }
  // the auto-waste cadence is the GROUP's (one counter per tracker object, simple_api.rs:115-120): a shard never wastes on its own count
  if (t->waste_counter == 0) {
    group_auto_waste(t);
    t->waste_counter = o.auto_waste_periodicity;
  } else t->waste_counter -= 1;
  std::unique_ptr<sa_batch_result> r(new sa_batch_result());
  r->spin_us = o.spin_us < 0 ? 500 : o.spin_us;
  if (!n_scenes) { for (sa_tracker* c : t->shards) { wait_outstanding(c); flush_pending(c); } *out_result = r.release(); return SA_OK; }
  struct Share { std::vector<uint64_t> ids, base; std::vector<uint32_t> counts; std::vector<const sa_observation*> obs; std::vector<sa_sort_track*> out; };
  std::vector<Share> sh(ns);
  if (o.batch_ids) {
    // Batch* ids: one per candidate, in request order — a function of the request alone, so every shard gets its scenes' bases up front
    uint64_t next = t->track_id;
    for (uint32_t s = 0; s < n_scenes; ++s) {
      Share& S = sh[shard_of(t, scene_ids[s])];
      S.ids.push_back(scene_ids[s]); S.counts.push_back(counts[s]); S.obs.push_back(obs[s]); S.base.push_back(next);
      if (caller_out) S.out.push_back(caller_out[s]);
      next += counts[s];
    }
    t->track_id = next;
    // every shard's share is begun on a thread of its own (the group's pool: shard k on thread k, the calling thread takes shard 0): a begin
    // is 70-180 us of host work whatever the number of scenes (assemble, stage, evict, four launches), and two of them one after the other
    // cost a 64 x 500 set 338 us before the last launch where one engine takes 183
    std::vector<sa_batch_result*> kids(ns, nullptr);
    std::vector<int> rcs(ns, SA_OK);
    auto begin_shard = [&](uint32_t k) {
      if (sh[k].ids.empty()) return;
      sa_tracker* c = t->shards[k];
      wait_outstanding(c);
      c->forced_base = sh[k].base.data();
      try {
        rcs[k] = sa_begin_into(c, (uint32_t)sh[k].ids.size(), sh[k].ids.data(), sh[k].counts.data(), sh[k].obs.data(), caller_out ? sh[k].out.data() : nullptr, &kids[k]);
      } catch (const std::exception& ex) { rcs[k] = tfail(c, SA_ERR_OOM, "%s", ex.what()); }
      c->forced_base = nullptr;
    };
    uint32_t busy = 0;
    for (uint32_t k = 0; k < ns; ++k) busy += sh[k].ids.empty() ? 0u : 1u;
    if (busy > 1) {
      // The group's own threads neither spin nor bind.  Bound (to the CPUs next to the caller's) they leave a shard's pool, which is created
      // ON such a thread, one allowed CPU and therefore unbound workers: 64 x 500 over two shards 880-1560 us per call; unbound but spinning
      // they can sit on the CPU a shard's bound worker is given later, and the two then take turns by the scheduler's tick (8 ms per call
      // seen once).  Asleep between calls they are placed afresh on an idle CPU at every wake-up, for a futex round trip per call.
      if (!t->pool) t->pool.reset(new SaPool(ns - 1, false, 0));
      t->pool->run(ns, begin_shard);
    } else
      for (uint32_t k = 0; k < ns; ++k) begin_shard(k);
    int rc = SA_OK;
    for (uint32_t k = 0; k < ns; ++k) {
      if (rcs[k] != SA_OK && rc == SA_OK) { rc = rcs[k]; group_fail(t, t->shards[k], rc); }
      if (kids[k]) { r->kids.push_back(kids[k]); r->kid_total += (uint32_t)sh[k].ids.size(); }
    }
    if (rc != SA_OK) { for (sa_batch_result* kid : r->kids) sa_batch_result_free(kid); return rc; }
  } else {
    // Sort / VisualSort id rules: an id per track that STARTS, drawn in candidate order — the counter a scene starts from depends on the
    // scenes before it, so the scenes run one after the other, each on its shard, and hand the counter on
    for (uint32_t s = 0; s < n_scenes; ++s) {
      sa_tracker* c = t->shards[shard_of(t, scene_ids[s])];
      wait_outstanding(c);
      c->track_id = t->track_id;
      sa_batch_result* kid = nullptr;
      int rc;
      try {
        rc = sa_begin_into(c, 1, &scene_ids[s], &counts[s], &obs[s], caller_out ? &caller_out[s] : nullptr, &kid);
      } catch (const std::exception& ex) { rc = tfail(c, SA_ERR_OOM, "%s", ex.what()); }
      if (rc == SA_OK) { wait_outstanding(c); t->track_id = c->track_id; }
      if (rc != SA_OK) { for (sa_batch_result* k2 : r->kids) sa_batch_result_free(k2); return group_fail(t, c, rc); }
      r->kids.push_back(kid);
      r->kid_total += 1;
    }
  }
  *out_result = r.release();
  return SA_OK;
}
// the next finished scene of a group handle: whichever shard has one
int group_next(sa_batch_result* r, uint64_t* out_scene_id, const sa_sort_track** tracks, uint32_t* out_n, bool bounded, uint32_t cap);
int group_predict(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts, const sa_observation* const* obs,
                  sa_sort_track* const* out) {
  for (uint32_t s = 0; s < n_scenes; ++s)
    if (counts[s] && !out[s]) return tfail(t, SA_ERR_BAD_ARG, "scene %llu: null output", (unsigned long long)scene_ids[s]);
  // every shard fills the caller's arrays of ITS scenes itself; the group waits for the shards' drivers and looks at what they report
  sa_batch_result* h = nullptr;
  int rc = group_begin(t, n_scenes, scene_ids, counts, obs, &h, out);
  if (rc != SA_OK) return rc;
  for (sa_tracker* c : t->shards) wait_outstanding(c);
  for (sa_batch_result* kid : h->kids) {
    std::lock_guard<std::mutex> lk(kid->st->mu);
    if (kid->st->rc != SA_OK && rc == SA_OK) { rc = kid->st->rc; t->err = kid->st->err; g_err = kid->st->err; }
  }
  sa_batch_result_free(h);
  return rc;
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
  o->spin_us = -1;   // the facade's own polling times (0 would mean: no thread of the tracker ever polls)
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
sa_engine* sa_tracker_engine(sa_tracker* t) { return !t ? nullptr : t->shards.empty() ? t->eng : t->shards[0]->eng; }   // (a device group: its first shard's)

int sa_tracker_create(const sa_tracker_options* o, sa_tracker** out) {
  if (!o || !out) return tfail(nullptr, SA_ERR_BAD_ARG, "sa_tracker_create: null argument");
  *out = nullptr;
  if (o->struct_size != sizeof(sa_tracker_options)) return tfail(nullptr, SA_ERR_BAD_ARG, "sa_tracker_options.struct_size mismatch");
  if (o->history_length == 0) return tfail(nullptr, SA_ERR_BAD_ARG, "bbox_history must be > 0 (sort/simple_api.rs:51)");
  if (o->visual && (o->feature_len == 0 || o->visual_max_observations == 0))
    return tfail(nullptr, SA_ERR_BAD_ARG, "VisualSort needs feature_len and visual_max_observations");
  if (o->workers < -256 || o->workers > 256) return tfail(nullptr, SA_ERR_BAD_ARG, "workers must lie in [-256, 256] (0 = the facade's own choice)");
  if (o->n_devices > 64 || (o->n_devices > 1 && !o->devices)) return tfail(nullptr, SA_ERR_BAD_ARG, "n_devices must be <= 64, with devices[] given");
  if (o->n_devices > 1) {
    // a device group: one complete tracker per entry of devices[]; this object routes (sa_tracker::shards)
    sa_tracker* g = new sa_tracker();
    g->o = *o;
    g->o.constraint_epoch_delta = nullptr; g->o.constraint_max_dist = nullptr; g->o.devices = nullptr;
    g->waste_counter = o->auto_waste_periodicity;
    for (uint32_t k = 0; k < o->n_devices; ++k) {
      sa_tracker_options oc = *o;
      oc.n_devices = 0; oc.devices = nullptr;
      oc.device = o->devices[k];
      oc.auto_waste_periodicity = 0xffffffffu;   // (the cadence is the group's)
      if (oc.workers == 0) {   // the facade's own choice, shared out: the group works on a request set with as many threads as ONE tracker would
        const uint32_t all = std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 8u));
        oc.workers = (int32_t)std::max(2u, all / o->n_devices);
      }
      sa_tracker* c = nullptr;
      int rc = sa_tracker_create(&oc, &c);
      if (rc != SA_OK) { for (sa_tracker* c2 : g->shards) sa_tracker_destroy(c2); delete g; return rc; }
      g->shards.push_back(c);
    }
    *out = g;
    return SA_OK;
  }
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
  if (!t->shards.empty()) {
    t->pool.reset();
    for (sa_tracker* c : t->shards) sa_tracker_destroy(c);
    delete t;
    return;
  }
  wait_outstanding(t);
  if (t->driver.joinable()) {
    {
      std::lock_guard<std::mutex> lk(t->dmu);
      t->d_stop = true;
    }
    t->dcv.notify_all();
    t->driver.join();
  }
  t->pool.reset();
  if (t->eng) sa_engine_destroy(t->eng);
  for (auto& kv : t->scenes) {
    for (Track* tr : kv.second.rows) kv.second.slab.put(tr);
    for (Track* tr : kv.second.evicted) kv.second.slab.put(tr);
    kv.second.slab.release();
  }
  delete t;
}

int sa_tracker_predict(sa_tracker* t, uint64_t scene_id, uint32_t n, const sa_observation* obs, sa_sort_track* out) {
  if (!t || (n && (!obs || !out))) return tfail(t, SA_ERR_BAD_ARG, "sa_tracker_predict: null argument");
  const sa_observation* op = obs;
  sa_sort_track* outp = out;
  try {
    if (!t->shards.empty()) return group_predict(t, 1, &scene_id, &n, &op, &outp);
    return predict_scenes(t, 1, &scene_id, &n, &op, &outp);
  } catch (const std::exception& ex) { return tfail(t, SA_ERR_OOM, "sa_tracker_predict: %s", ex.what()); }
}

int sa_tracker_predict_batch(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts,
                             const sa_observation* const* obs, sa_sort_track* const* out) {
  if (!t || (n_scenes && (!scene_ids || !counts || !obs || !out))) return tfail(t, SA_ERR_BAD_ARG, "sa_tracker_predict_batch: null argument");
  try {
    if (!t->shards.empty()) return group_predict(t, n_scenes, scene_ids, counts, obs, out);
    return predict_scenes(t, n_scenes, scene_ids, counts, obs, out);
  } catch (const std::exception& ex) { return tfail(t, SA_ERR_OOM, "sa_tracker_predict_batch: %s", ex.what()); }
}

// Batch*::predict(PredictionBatchRequest) -> PredictionBatchResult  (sort/batch_api.rs:222-290, trackers/batch.rs:19-38)
int sa_tracker_predict_batch_begin(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts,
                                   const sa_observation* const* obs, sa_batch_result** out_result) {
  if (!t || !out_result || (n_scenes && (!scene_ids || !counts || !obs))) return tfail(t, SA_ERR_BAD_ARG, "sa_tracker_predict_batch_begin: null argument");
  *out_result = nullptr;
  if (!t->shards.empty()) return group_begin(t, n_scenes, scene_ids, counts, obs, out_result);
  return sa_begin_into(t, n_scenes, scene_ids, counts, obs, nullptr, out_result);
}
}  // extern "C"
// _begin with the tracks going either into a block of the handle's (caller_out == nullptr: sa_batch_result_get / _take hand them out) or
// straight into the caller's arrays (a device group's synchronous predict: every shard fills its scenes' arrays itself, the group only waits)
int sa_begin_into(sa_tracker* t, uint32_t n_scenes, const uint64_t* scene_ids, const uint32_t* counts, const sa_observation* const* obs,
                  sa_sort_track* const* caller_out, sa_batch_result** out_result) {
  wait_outstanding(t);
  auto st = std::make_shared<ResultState>();
  st->scene_ids.assign(scene_ids, scene_ids + n_scenes);
  st->off.resize((size_t)n_scenes + 1, 0);
  for (uint32_t s = 0; s < n_scenes; ++s) st->off[s + 1] = st->off[s] + counts[s];
  st->blocks = t->blocks;
  if (!caller_out) st->store = t->blocks->take(st->off[n_scenes] ? st->off[n_scenes] : 1);
  std::vector<sa_sort_track*> outs(n_scenes);
  for (uint32_t s = 0; s < n_scenes; ++s) outs[s] = caller_out ? caller_out[s] : st->store.p.get() + st->off[s];
  const bool fused = t->o.device_upkeep && (t->o.batch_ids || n_scenes == 1) && n_scenes;
  int rc;
  try {
    rc = predict_scenes(t, n_scenes, scene_ids, counts, obs, outs.data(), fused ? st : nullptr);
  } catch (const std::exception& ex) { rc = tfail(t, SA_ERR_OOM, "sa_tracker_predict_batch_begin: %s", ex.what()); }
  if (rc != SA_OK) return rc;
  if (!fused) {   // everything happened inside the call: every scene is ready
    std::lock_guard<std::mutex> lk(st->mu);
    for (uint32_t s = 0; s < n_scenes; ++s) st->ready_q.push_back(s);
    st->finished = true;
    st->n_ready.store(n_scenes | 0x80000000u, std::memory_order_release);
  }
  sa_batch_result* r = new sa_batch_result();
  r->st = st;
  r->spin_us = t->o.spin_us < 0 ? 500 : t->o.spin_us;
  *out_result = r;
  return SA_OK;
}
extern "C" {

uint32_t sa_batch_result_size(const sa_batch_result* r) { return !r ? 0 : r->st ? (uint32_t)r->st->scene_ids.size() : r->kid_total; }

int sa_batch_result_ready(sa_batch_result* r) {
  if (!r) return 0;
  if (!r->st) {   // a device group's handle: any shard
    for (sa_batch_result* kid : r->kids)
      if (sa_batch_result_ready(kid)) return 1;
    return 0;
  }
  std::lock_guard<std::mutex> lk(r->st->mu);
  return (!r->st->ready_q.empty() || (r->st->finished && r->st->rc != SA_OK)) ? 1 : 0;
}

// The next finished scene of the handle: waits for one (the delivery counter first, then the lock and the condition variable), takes it
// off the queue unless `cap` says the caller's array is too short.  The scene's tracks lie in the handle's block until it is freed.
static int result_next(sa_batch_result* r, uint64_t* out_scene_id, const sa_sort_track** tracks, uint32_t* out_n, bool bounded, uint32_t cap) {
  ResultState& st = *r->st;
  {
    // (the next scene is usually microseconds away: look at the delivery counter for a while before the lock and the futex; `taken` is only
    // written by get() — by this thread, or by another caller thread under the lock, in which case the wait below sorts it out)
    const uint32_t have = st.taken;
    const auto t0 = clk::now();
    for (uint32_t spin = 1; r->spin_us > 0; ++spin) {
      const uint32_t v = st.n_ready.load(std::memory_order_acquire);
      if ((v & 0x7fffffffu) > have || (v & 0x80000000u)) break;
      SA_POOL_PAUSE();
      if ((spin & 255u) == 0 && clk::now() - t0 > std::chrono::microseconds(r->spin_us)) break;
    }
  }
  std::unique_lock<std::mutex> lk(st.mu);
  if (st.taken >= st.scene_ids.size()) return SA_ERR_STATE;   // every scene has been taken (the reference's recv() would block for ever)
  st.cv.wait(lk, [&] { return !st.ready_q.empty() || st.finished; });
  if (st.ready_q.empty()) { g_err = st.err; return st.rc != SA_OK ? st.rc : SA_ERR_STATE; }
  const uint32_t s = st.ready_q.front();
  const uint32_t n = (uint32_t)(st.off[s + 1] - st.off[s]);
  if (out_scene_id) *out_scene_id = st.scene_ids[s];
  *out_n = n;
  if (bounded && n && cap < n) return SA_ERR_BAD_ARG;   // (nothing taken: call again with room for *out_n tracks)
  *tracks = st.store.p.get() + st.off[s];
  st.ready_q.pop_front();
  ++st.taken;
  return SA_OK;
}

}  // extern "C"
namespace {
// A group handle's next scene: the shards' handles are looked at in turn (a ready scene is taken at once); while none has one, the
// calling thread polls for spin_us and then naps 20 us at a time — the shards' drivers deliver under their own locks.
int group_next(sa_batch_result* r, uint64_t* out_scene_id, const sa_sort_track** tracks, uint32_t* out_n, bool bounded, uint32_t cap) {
  if (r->kid_taken >= r->kid_total) return SA_ERR_STATE;
  const auto t0 = clk::now();
  const uint32_t nk = (uint32_t)r->kids.size();
  for (;;) {
    for (uint32_t i = 0; i < nk; ++i) {
      const uint32_t k = (r->kid_next + i) % nk;
      sa_batch_result* kid = r->kids[k];
      if (kid->st->taken >= kid->st->scene_ids.size() || !sa_batch_result_ready(kid)) continue;
      int rc = result_next(kid, out_scene_id, tracks, out_n, bounded, cap);
      if (rc == SA_OK) { ++r->kid_taken; r->kid_next = (k + 1) % nk; }
      return rc;
    }
    if (clk::now() - t0 > std::chrono::microseconds(r->spin_us)) std::this_thread::sleep_for(std::chrono::microseconds(20));
    else SA_POOL_PAUSE();
  }
}
}  // namespace
extern "C" {

int sa_batch_result_get(sa_batch_result* r, uint64_t* out_scene_id, sa_sort_track* out, uint32_t cap, uint32_t* out_n) {
  if (!r || !out_n) return SA_ERR_BAD_ARG;
  const sa_sort_track* src = nullptr;
  int rc = r->st ? result_next(r, out_scene_id, &src, out_n, true, out ? cap : 0) : group_next(r, out_scene_id, &src, out_n, true, out ? cap : 0);
  if (rc != SA_OK) return rc;
  if (*out_n) std::memcpy(out, src, (size_t)*out_n * sizeof(sa_sort_track));   // (outside the handle's lock: the scenes' jobs deliver under it)
  return SA_OK;
}

int sa_batch_result_take(sa_batch_result* r, uint64_t* out_scene_id, const sa_sort_track** out_tracks, uint32_t* out_n) {
  if (!r || !out_tracks || !out_n) return SA_ERR_BAD_ARG;
  return r->st ? result_next(r, out_scene_id, out_tracks, out_n, false, 0) : group_next(r, out_scene_id, out_tracks, out_n, false, 0);
}

void sa_batch_result_free(sa_batch_result* r) {
  if (!r) return;
  for (sa_batch_result* kid : r->kids) sa_batch_result_free(kid);
  delete r;
}

int sa_tracker_idle_tracks(sa_tracker* t, uint64_t scene_id, sa_sort_track* out, uint32_t cap, uint32_t* out_n) {
  if (!t || !out_n) return tfail(t, SA_ERR_BAD_ARG, "sa_tracker_idle_tracks: null argument");
  if (!t->shards.empty()) {
    sa_tracker* c = t->shards[shard_of(t, scene_id)];
    int rc = sa_tracker_idle_tracks(c, scene_id, out, cap, out_n);
    return rc == SA_OK ? rc : group_fail(t, c, rc);
  }
  wait_outstanding(t);
  flush_pending(t);
  // IdleLookup  sort.rs:213-228: same scene and last_updated_epoch != current epoch
  uint32_t n = 0;
  auto it = t->scenes.find(scene_id);
  if (it != t->scenes.end())
    for (const auto* list : {&it->second.rows, &it->second.evicted})   // (evicted tracks are idle tracks like any other until they are wasted)
      for (const Track* trp : *list)
        if (trp->epoch != it->second.epoch) {
          if (out && n < cap) out[n] = to_sort_track(t->o, *trp);
          ++n;
        }
  *out_n = n;
  return SA_OK;
}

int sa_tracker_skip_epochs(sa_tracker* t, uint64_t scene_id, uint64_t n) {
  if (!t) return SA_ERR_BAD_ARG;
  if (!t->shards.empty()) {   // (the owner skips and wastes its own scenes; the auto-waste behind it reaches every scene of the tracker)
    sa_tracker* c = t->shards[shard_of(t, scene_id)];
    int rc = sa_tracker_skip_epochs(c, scene_id, n);
    if (rc != SA_OK) return group_fail(t, c, rc);
    group_auto_waste(t, c);
    return SA_OK;
  }
  wait_outstanding(t);
  flush_pending(t);
  scene_of(t, scene_id).epoch += n;  // skip_epochs_for_scene  epoch_db.rs:11-20
  return auto_waste(t);              // tracker_api.rs:48-51
}

int sa_tracker_current_epoch(sa_tracker* t, uint64_t scene_id, uint64_t* out) {
  if (!t || !out) return SA_ERR_BAD_ARG;
  if (!t->shards.empty()) return sa_tracker_current_epoch(t->shards[shard_of(t, scene_id)], scene_id, out);
  wait_outstanding(t);
  *out = current_epoch(t, scene_id);
  return SA_OK;
}

int sa_tracker_wasted(sa_tracker* t, sa_sort_track* out, uint32_t cap, uint32_t* out_n) {
  if (!t || !out_n) return tfail(t, SA_ERR_BAD_ARG, "sa_tracker_wasted: null argument");
  if (!t->shards.empty()) {   // every shard's wasted store, as one list by id
    std::vector<sa_sort_track> all;
    for (sa_tracker* c : t->shards) {
      uint32_t nc = 0;
      int rc = sa_tracker_wasted(c, nullptr, 0, &nc);   // (auto-wastes; nothing leaves the shard's store)
      if (rc != SA_OK) return group_fail(t, c, rc);
      all.insert(all.end(), c->wasted_store.begin(), c->wasted_store.end());
    }
    std::sort(all.begin(), all.end(), [](const sa_sort_track& a, const sa_sort_track& b) { return a.id < b.id; });
    const uint32_t n = (uint32_t)all.size();
    for (uint32_t i = 0; i < n && i < cap && out; ++i) out[i] = all[i];
    *out_n = n;
    if (out && cap >= n)
      for (sa_tracker* c : t->shards) c->wasted_store.clear();
    return SA_OK;
  }
  wait_outstanding(t);
  flush_pending(t);
  int rc = auto_waste(t);
  if (rc != SA_OK) return rc;
  std::sort(t->wasted_store.begin(), t->wasted_store.end(), [](const sa_sort_track& a, const sa_sort_track& b) { return a.id < b.id; });
  uint32_t n = (uint32_t)t->wasted_store.size();
  for (uint32_t i = 0; i < n && i < cap && out; ++i) out[i] = t->wasted_store[i];
  *out_n = n;
  if (out && cap >= n) t->wasted_store.clear();  // fetch_tracks removes them from the wasted store
  return SA_OK;
}

int sa_tracker_clear_wasted(sa_tracker* t) {
  if (!t) return SA_ERR_BAD_ARG;
  if (!t->shards.empty()) { for (sa_tracker* c : t->shards) sa_tracker_clear_wasted(c); return SA_OK; }
  wait_outstanding(t);
  t->wasted_store.clear();
  return SA_OK;
}

int sa_tracker_active_tracks(sa_tracker* t, uint64_t* out_n) {
  if (!t || !out_n) return SA_ERR_BAD_ARG;
  if (!t->shards.empty()) {
    *out_n = 0;
    for (sa_tracker* c : t->shards) { uint64_t k = 0; sa_tracker_active_tracks(c, &k); *out_n += k; }
    return SA_OK;
  }
  wait_outstanding(t);
  *out_n = t->n_active;
  return SA_OK;
}

int sa_tracker_track_state(sa_tracker* t, uint64_t track_id, float* mean10, float* cov100) {
  if (!t) return SA_ERR_BAD_ARG;
  if (!t->shards.empty()) {   // (ids are the group's: exactly one shard knows the track)
    for (sa_tracker* c : t->shards) {
      wait_outstanding(c);
      if (!find_track(c, track_id)) continue;
      int rc = sa_tracker_track_state(c, track_id, mean10, cov100);
      return rc == SA_OK ? rc : group_fail(t, c, rc);
    }
    return tfail(t, SA_ERR_NOT_FOUND, "unknown track id %llu", (unsigned long long)track_id);
  }
  wait_outstanding(t);
  Track* tr = find_track(t, track_id);
  if (!tr) return tfail(t, SA_ERR_NOT_FOUND, "unknown track id %llu", (unsigned long long)track_id);
  if (t->o.device_upkeep) {  // the state lives on the device
    if (!tr->in_engine)
      return tfail(t, SA_ERR_NOT_FOUND, "track %llu has been idle for more than max_idle_epochs: its row (and filter state) has left the device", (unsigned long long)track_id);
    int rc = sa_tracks_get_state(t->eng, tr->scene, track_id, mean10, cov100, nullptr, nullptr, nullptr);
    return rc == SA_OK ? SA_OK : tfail(t, rc, "sa_tracks_get_state: %s", sa_last_error(t->eng));
  }
  if (mean10) std::memcpy(mean10, tr->kf->mean, sizeof tr->kf->mean);
  if (cov100) std::memcpy(cov100, tr->kf->cov, sizeof tr->kf->cov);
  return SA_OK;
}

int sa_tracker_track_info(sa_tracker* t, uint64_t track_id, uint64_t out4[4]) {
  if (!t || !out4) return SA_ERR_BAD_ARG;
  if (!t->shards.empty()) {
    for (sa_tracker* c : t->shards) {
      wait_outstanding(c);
      if (!find_track(c, track_id)) continue;
      int rc = sa_tracker_track_info(c, track_id, out4);
      return rc == SA_OK ? rc : group_fail(t, c, rc);
    }
    return tfail(t, SA_ERR_NOT_FOUND, "unknown track id %llu", (unsigned long long)track_id);
  }
  wait_outstanding(t);
  flush_pending(t);
  const Track* trp = find_track(t, track_id);
  if (!trp) return tfail(t, SA_ERR_NOT_FOUND, "unknown track id %llu", (unsigned long long)track_id);
  const Track& tr = *trp;
  out4[0] = tr.feat_count;
  out4[1] = t->o.visual ? tr.obs.size() : 1;
  out4[2] = tr.boxes.size();
  out4[3] = tr.length;
  return SA_OK;
}

}  // extern "C"
