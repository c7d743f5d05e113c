// sa_cluster.cpp — one host process, one association engine per GPU, scenes sharded across them.
//
// The reference fans the scenes of one PredictionBatchRequest out to voting threads inside ONE process
// (sort/batch_api.rs:197-207, 278-288; visual_sort/batch_api.rs:296-315); scenes never interact (compatible() is false across
// scene ids, sort.rs:251), so the natural MI355X form is one engine per device and a router in front of them:
//   scene -> shard  scene_id % n_shards   (sticky: the scene's track table stays resident on that GPU, the way the reference
//                                          keeps a track in store shard  track_id % shards, store.rs:490-493)
// sa_cluster_associate_batch splits a request set by shard, hands every shard's share to that shard's worker thread (which owns
// the device context of its engine and calls sa_associate_batch there: one DMA + one set of launches per GPU), and returns when
// all of them have — the scatter and the gather of the Batch* API without leaving the process, no Python and no collective on
// the path.  Nothing here computes: the engines do.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/similari_assoc.h"

namespace {
thread_local std::string g_cluster_err;

struct Shard {
  sa_engine* eng = nullptr;
  int device = 0;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;        // the worker sleeps here between tasks
  std::condition_variable done_cv;   // callers sleep here while a task runs
  // mailbox (one task at a time per shard)
  std::atomic<int> pending{0};   // 1 = a task is waiting / running
  int kind = 0;                  // 1 associate_batch, 2 upsert, 3 remove, 9 quit
  std::vector<sa_scene_request> req;
  std::vector<sa_scene_result> res;
  uint64_t scene_id = 0;
  const sa_tracks* tracks = nullptr;
  uint32_t n_ids = 0;
  const uint64_t* ids = nullptr;
  int rc = SA_OK;
  std::string err;
  double ms = 0.0;               // wall time of the last task inside the worker
};
}  // namespace

struct sa_cluster {
  std::mutex api;  // one cluster call at a time: the shards' mailboxes (req / res / kind) are shared state
  std::vector<Shard*> shards;
  std::string err;
  std::vector<double> last_ms;
};

namespace {

int cfail(sa_cluster* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  else g_cluster_err = buf;
  return code;
}

void run_task(Shard* s) {
  const auto t0 = std::chrono::steady_clock::now();
  switch (s->kind) {
    case 1: s->rc = sa_associate_batch(s->eng, (uint32_t)s->req.size(), s->req.data(), s->res.data()); break;
    case 2: s->rc = sa_tracks_upsert(s->eng, s->scene_id, s->tracks); break;
    case 3: s->rc = sa_tracks_remove(s->eng, s->scene_id, s->n_ids, s->ids); break;
    default: s->rc = SA_OK; break;
  }
  if (s->rc != SA_OK) s->err = sa_last_error(s->eng);
  s->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// Worker: owns the shard's engine from the first task on (HIP's current device is per thread; the engine sets it itself on every
// entry point).  Between tasks it spins for a short while — batches of a video pipeline arrive back to back — then sleeps.
void worker(Shard* s) {
  for (;;) {
    int spins = 0;
    while (s->pending.load(std::memory_order_acquire) == 0) {
      if (++spins < 20000) continue;
      std::unique_lock<std::mutex> lk(s->mu);
      s->cv.wait(lk, [&] { return s->pending.load(std::memory_order_acquire) != 0; });
    }
    const bool quit = s->kind == 9;
    if (!quit) run_task(s);
    {
      std::lock_guard<std::mutex> lk(s->mu);  // (under the lock: a caller that is about to sleep on the condition cannot miss this)
      s->pending.store(0, std::memory_order_release);
    }
    s->done_cv.notify_all();
    if (quit) return;
  }
}

void post(Shard* s, int kind) {
  s->kind = kind;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->pending.store(1, std::memory_order_release);
  }
  s->cv.notify_one();
}
// The caller's side: a short spin (a shard's share of a batch is tens of microseconds), then sleep on the shard's condition — a host
// core per cluster call is not burnt for the duration of every shard's GPU work, and the workers staging data keep theirs.
void wait_done(Shard* s) {
  for (int spins = 0; spins < 4000; ++spins)
    if (s->pending.load(std::memory_order_acquire) == 0) return;
  std::unique_lock<std::mutex> lk(s->mu);
  s->done_cv.wait(lk, [&] { return s->pending.load(std::memory_order_acquire) == 0; });
}

// the share of every shard, in request order; map[i] = (shard, position inside its share)
struct Split {
  std::vector<std::pair<uint32_t, uint32_t>> map;
};
int split(sa_cluster* c, uint32_t n, const sa_scene_request* req, const sa_scene_result* res, Split* sp) {
  const uint32_t S = (uint32_t)c->shards.size();
  for (Shard* s : c->shards) { s->req.clear(); s->res.clear(); }
  sp->map.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    Shard* s = c->shards[req[i].scene_id % S];
    sp->map[i] = {(uint32_t)(req[i].scene_id % S), (uint32_t)s->req.size()};
    s->req.push_back(req[i]);
    if (res) s->res.push_back(res[i]);
  }
  return SA_OK;
}
int collect(sa_cluster* c, const char* what) {
  int rc = SA_OK;
  c->last_ms.assign(c->shards.size(), 0.0);
  for (size_t k = 0; k < c->shards.size(); ++k) {
    Shard* s = c->shards[k];
    c->last_ms[k] = s->ms;
    if (s->rc != SA_OK && rc == SA_OK) rc = cfail(c, s->rc, "%s: shard %zu (device %d): %s", what, k, s->device, s->err.c_str());
  }
  return rc;
}

}  // namespace

extern "C" {

const char* sa_cluster_last_error(const sa_cluster* c) { return c ? c->err.c_str() : g_cluster_err.c_str(); }

int sa_cluster_create(const sa_config* cfg, uint32_t n_shards, const int32_t* devices, sa_cluster** out) {
  if (!cfg || !out || !n_shards) return cfail(nullptr, SA_ERR_BAD_ARG, "sa_cluster_create: null argument or no shards");
  *out = nullptr;
  sa_cluster* c = new sa_cluster();
  for (uint32_t k = 0; k < n_shards; ++k) {
    sa_config sc = *cfg;
    sc.device = devices ? devices[k] : (int32_t)k;
    sc.stream = nullptr;  // every engine its own streams
    Shard* s = new Shard();
    s->device = sc.device;
    int rc = sa_engine_create(&sc, &s->eng);
    if (rc != SA_OK) {
      cfail(nullptr, rc, "sa_cluster_create: shard %u (device %d): %s", k, sc.device, sa_last_error(nullptr));
      delete s;
      for (Shard* t : c->shards) { post(t, 9); t->th.join(); sa_engine_destroy(t->eng); delete t; }
      delete c;
      return rc;
    }
    s->th = std::thread(worker, s);
    c->shards.push_back(s);
  }
  c->last_ms.assign(n_shards, 0.0);
  *out = c;
  return SA_OK;
}

void sa_cluster_destroy(sa_cluster* c) {
  if (!c) return;
  for (Shard* s : c->shards) {
    wait_done(s);
    post(s, 9);
    s->th.join();
    sa_engine_destroy(s->eng);
    delete s;
  }
  delete c;
}

uint32_t sa_cluster_size(const sa_cluster* c) { return c ? (uint32_t)c->shards.size() : 0; }
uint32_t sa_cluster_shard_of(const sa_cluster* c, uint64_t scene_id) { return c && !c->shards.empty() ? (uint32_t)(scene_id % c->shards.size()) : 0; }
sa_engine* sa_cluster_engine(sa_cluster* c, uint32_t shard) { return c && shard < c->shards.size() ? c->shards[shard]->eng : nullptr; }

int sa_cluster_tracks_upsert(sa_cluster* c, uint64_t scene_id, const sa_tracks* t) {
  if (!c || !t) return cfail(c, SA_ERR_BAD_ARG, "sa_cluster_tracks_upsert: null argument");
  std::lock_guard<std::mutex> guard(c->api);
  Shard* s = c->shards[scene_id % c->shards.size()];
  wait_done(s);
  s->scene_id = scene_id;
  s->tracks = t;
  post(s, 2);
  wait_done(s);
  return s->rc == SA_OK ? SA_OK : cfail(c, s->rc, "sa_tracks_upsert on device %d: %s", s->device, s->err.c_str());
}

int sa_cluster_tracks_remove(sa_cluster* c, uint64_t scene_id, uint32_t n, const uint64_t* ids) {
  if (!c || (n && !ids)) return cfail(c, SA_ERR_BAD_ARG, "sa_cluster_tracks_remove: null argument");
  std::lock_guard<std::mutex> guard(c->api);
  Shard* s = c->shards[scene_id % c->shards.size()];
  wait_done(s);
  s->scene_id = scene_id;
  s->n_ids = n;
  s->ids = ids;
  post(s, 3);
  wait_done(s);
  return s->rc == SA_OK ? SA_OK : cfail(c, s->rc, "sa_tracks_remove on device %d: %s", s->device, s->err.c_str());
}

int sa_cluster_associate_batch(sa_cluster* c, uint32_t n_scenes, const sa_scene_request* req, const sa_scene_result* res) {
  if (!c || (n_scenes && (!req || !res))) return cfail(c, SA_ERR_BAD_ARG, "sa_cluster_associate_batch: null argument");
  std::lock_guard<std::mutex> guard(c->api);
  Split sp;
  for (Shard* s : c->shards) wait_done(s);
  split(c, n_scenes, req, res, &sp);
  for (Shard* s : c->shards) { s->rc = SA_OK; s->ms = 0.0; if (!s->req.empty()) post(s, 1); }   // scatter: every GPU starts on its scenes
  for (Shard* s : c->shards) wait_done(s);                                                      // gather: results are in the caller's arrays
  return collect(c, "sa_associate_batch");
}

double sa_cluster_last_ms(const sa_cluster* c, uint32_t shard) { return c && shard < c->last_ms.size() ? c->last_ms[shard] : 0.0; }

}  // extern "C"
