// sa_pool.h — a small fork-join pool for the tracker facade's per-scene host work.
//
// BatchSort / BatchVisualSort::predict hand every scene's votes to one of `voting_shards` threads (sort/batch_api.rs:197-207,
// 278-288: `i % voting_threads.len()`); the facade does the same with the O(N) bookkeeping it keeps on the host around the GPU's
// association: scene s of a request set is one job, and job i ALWAYS runs on thread i % threads (the calling thread is thread 0) — a
// scene's tracks, work arrays and table bookkeeping stay in one core's caches from frame to frame (measured on the 2 x 64-core host of
// the MI355X box: with jobs handed out first come first served every track record a merge touched came out of another core's cache,
// ~100 ns per object; four threads were slower than one).  run(n, fn) returns when every job has finished.  Workers spin for a few
// hundred microseconds after a run (a tracker loop calls predict() back to back: the next run finds them awake) and then sleep on a
// condition variable.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#define SA_POOL_PAUSE() _mm_pause()
#else
#define SA_POOL_PAUSE() std::this_thread::yield()
#endif

class SaPool {
 public:
  explicit SaPool(uint32_t workers) {
    for (uint32_t w = 0; w < workers; ++w) th_.emplace_back([this, w] { loop(w + 1); });
  }
  ~SaPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  uint32_t threads() const { return (uint32_t)th_.size() + 1; }

  // fn(i) for i in [0, n): job i on thread i % threads().  One run at a time (the facade's entry points are serial).
  void run(uint32_t n, const std::function<void(uint32_t)>& fn) {
    if (!n) return;
    const uint32_t nt = threads();
    if (nt == 1 || n == 1) {
      for (uint32_t i = 0; i < n; ++i) fn(i);
      return;
    }
    Run r;   // everything a worker reads about this run lives here and is immutable but for the counters: a worker that wakes up late
    r.fn = &fn;   // either finds no run at all or holds a reference to the one it works on — never a mix of two
    r.n = n;
    r.nt = nt;
    {
      std::lock_guard<std::mutex> lk(mu_);
      cur_ = &r;
      gen_.fetch_add(1, std::memory_order_release);
    }
    if (sleepers_.load(std::memory_order_acquire)) cv_.notify_all();
    for (uint32_t i = 0; i < n; i += nt) fn(i);
    const uint32_t others = n - (n + nt - 1) / nt;   // jobs of the other threads
    while (r.done.load(std::memory_order_acquire) < others) SA_POOL_PAUSE();
    {
      std::lock_guard<std::mutex> lk(mu_);
      cur_ = nullptr;
    }
    while (r.refs.load(std::memory_order_acquire)) SA_POOL_PAUSE();   // (workers past their last job, about to let go)
  }

 private:
  struct Run {
    const std::function<void(uint32_t)>* fn = nullptr;
    uint32_t n = 0, nt = 1;
    std::atomic<uint32_t> done{0}, refs{0};
  };
  void loop(uint32_t me) {
    uint64_t seen = 0;
    for (;;) {
      // spin for ~300 us (a predict() of a running tracker loop is back within that), then sleep
      uint64_t g = gen_.load(std::memory_order_acquire);
      if (g == seen) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spin = 1; g == seen; ++spin) {
          SA_POOL_PAUSE();
          g = gen_.load(std::memory_order_acquire);
          if ((spin & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
        }
      }
      Run* r = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        if (gen_.load(std::memory_order_acquire) == seen) {
          sleepers_.fetch_add(1, std::memory_order_acq_rel);
          cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
          sleepers_.fetch_sub(1, std::memory_order_acq_rel);
        }
        seen = gen_.load(std::memory_order_acquire);
        if (stop_) return;
        r = cur_;
        if (r) r->refs.fetch_add(1, std::memory_order_acq_rel);
      }
      if (!r) continue;
      uint32_t mine = 0;
      for (uint32_t i = me; i < r->n; i += r->nt) { (*r->fn)(i); ++mine; }
      if (mine) r->done.fetch_add(mine, std::memory_order_acq_rel);
      r->refs.fetch_sub(1, std::memory_order_acq_rel);
    }
  }

  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0};
  std::atomic<uint32_t> sleepers_{0};
  bool stop_ = false;
  Run* cur_ = nullptr;
};
