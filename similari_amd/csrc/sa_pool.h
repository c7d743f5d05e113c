// sa_pool.h — a small fork-join pool for the tracker facade's per-scene host work.
//
// BatchSort / BatchVisualSort::predict hand every scene's votes to one of `voting_shards` threads (sort/batch_api.rs:197-207,
// 278-288: `i % voting_threads.len()`); the facade does the same with the O(N) bookkeeping it keeps on the host around the GPU's
// association: scene s of a request set is one job, and job i ALWAYS runs on thread i % threads (the calling thread is thread 0) — a
// scene's tracks, work arrays and table bookkeeping stay in one core's caches from frame to frame (measured on the 2 x 64-core host of
// the MI355X box: with jobs handed out first come first served every track record a merge touched came out of another core's cache,
// ~100 ns per object; four threads were slower than one).  run(n, fn) returns when every job has finished.  Workers spin for a few
// hundred microseconds after a run (a tracker loop calls predict() back to back: the next run finds them awake; spin_us, 0 = not at all)
// and then sleep on a condition variable.  A job that throws does not hang the run: the exception is caught where it happens, every
// worker still answers, and run() rethrows the first one on the calling thread.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#endif
#if defined(__x86_64__)
#include <immintrin.h>
#define SA_POOL_PAUSE() _mm_pause()
#else
#define SA_POOL_PAUSE() std::this_thread::yield()
#endif

class SaPool {
 public:
  // pin: worker w is bound to the (w + 1)-th CPU after the creating thread's that no other pool of the process holds, among the CPUs that
  // thread may run on — its neighbours in the same socket and cache complex on the usual numbering.  Measured on the 2 x 64-core host of the MI355X box (64 scenes x 500 objects,
  // 8 threads): left to the scheduler the workers land on the other socket and a merge job runs at 300 ns per object, every record a
  // remote miss; next to the caller at 7.
  explicit SaPool(uint32_t workers, bool pin = true, int spin_us = 300) : acks_(workers ? new Ack[workers] : nullptr), nw_(workers), spin_us_(spin_us < 0 ? 300 : spin_us) {
#if defined(__linux__)
    if (pin && workers) {
      // the CPUs the creating thread may run on, from its own onwards; the first workers + 1 of them that no other pool of this process
      // has taken (two trackers created by one thread would otherwise bind their workers to the SAME CPUs: a worker still spinning after
      // its tracker's run and a worker of the other tracker's run on one CPU take turns by the scheduler's tick — milliseconds)
      cpu_set_t allowed;
      CPU_ZERO(&allowed);
      const int here = sched_getcpu();
      std::vector<int> cpus;
      int at = -1;
      if (sched_getaffinity(0, sizeof allowed, &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; ++c)
          if (CPU_ISSET(c, &allowed)) { if (c == here) at = (int)cpus.size(); cpus.push_back(c); }
      if (at >= 0) {
        std::lock_guard<std::mutex> lk(claims_mu());
        std::vector<int>& taken = claims();
        for (size_t k = 1; k < cpus.size() && mine_.size() < (size_t)workers + 1; ++k) {
          const int c = cpus[((size_t)at + k) % cpus.size()];
          if (std::find(taken.begin(), taken.end(), c) == taken.end()) mine_.push_back(c);
        }
        if (mine_.size() == (size_t)workers + 1) taken.insert(taken.end(), mine_.begin(), mine_.end());
        else mine_.clear();   // (not enough free CPUs: this pool is left to the scheduler)
      }
      if (!mine_.empty()) next_cpu_ = mine_.back();   // (behind the workers: kept for a companion thread of the owner's, next_cpu())
    }
#endif
    for (uint32_t w = 0; w < workers; ++w) {
      th_.emplace_back([this, w] { loop(w); });
#if defined(__linux__)
      if (!mine_.empty()) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(mine_[w], &set);
        pthread_setaffinity_np(th_.back().native_handle(), sizeof set, &set);
      }
#endif
    }
  }
  ~SaPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
    delete[] acks_;
    if (!mine_.empty()) {
      std::lock_guard<std::mutex> lk(claims_mu());
      std::vector<int>& taken = claims();
      for (int c : mine_) {
        auto it = std::find(taken.begin(), taken.end(), c);
        if (it != taken.end()) taken.erase(it);
      }
    }
  }
  uint32_t threads() const { return nw_ + 1; }
  // The CPU right behind the pinned workers' (-1: the pool is not pinned): where a companion thread of the owner belongs — chosen HERE,
  // from the same list and the same position the workers were placed by (a thread that looks up "the CPUs next to the caller" by itself,
  // later, starts from wherever the caller runs THEN and can land on a worker's CPU: two spinning threads on one CPU take turns by the
  // scheduler's tick, milliseconds at a time).
  int next_cpu() const { return next_cpu_; }
  const std::vector<int>& claimed_cpus() const { return mine_; }   // (the workers', then next_cpu(); empty: not pinned)

  // fn(i) for i in [0, n): job i on thread i % threads().  One run at a time (the facade's entry points are serial).
  // No lock on the way: the run is published by one store (the workers spin on that word), every worker answers with one store to a
  // cache line of its own, and the caller returns when all of them have answered — so nobody can be late for the NEXT run either.
  void run(uint32_t n, const std::function<void(uint32_t)>& fn) {
    if (!n) return;
    const uint32_t nt = threads();
    if (nt == 1 || n == 1) {
      for (uint32_t i = 0; i < n; ++i) fn(i);
      return;
    }
    fn_ = &fn;
    n_ = n;
    const uint64_t g = gen_.load(std::memory_order_relaxed) + 1;
    gen_.store(g, std::memory_order_seq_cst);   // (Dekker with the sleepers' count: one side must see the other)
    if (sleepers_.load(std::memory_order_seq_cst)) {
      { std::lock_guard<std::mutex> lk(mu_); }   // (a worker between its last look at gen_ and its wait holds the mutex: wait for it to be in the wait)
      cv_.notify_all();
    }
    std::exception_ptr mine;
    try {
      for (uint32_t i = 0; i < n; i += nt) fn(i);
    } catch (...) { mine = std::current_exception(); }   // (the workers still read fn_: wait for them before unwinding)
    for (uint32_t w = 0; w < nw_; ++w)
      for (uint32_t spin = 1; acks_[w].gen.load(std::memory_order_acquire) != g; ++spin) {
        if ((spin & 4095u) == 0) std::this_thread::yield();   // (a worker that lost its CPU to somebody else's thread)
        else SA_POOL_PAUSE();
      }
    std::exception_ptr theirs;
    {
      std::lock_guard<std::mutex> lk(mu_);
      theirs = error_;
      error_ = nullptr;
    }
    if (mine) std::rethrow_exception(mine);
    if (theirs) std::rethrow_exception(theirs);
  }

 private:
  struct alignas(64) Ack { std::atomic<uint64_t> gen{0}; };
  void loop(uint32_t w) {
    uint64_t seen = 0;
    for (;;) {
      // spin for spin_us (300 by default: a predict() of a running tracker loop is back within that), then sleep
      uint64_t g = gen_.load(std::memory_order_acquire);
      if (g == seen && spin_us_ > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spin = 1; g == seen; ++spin) {
          SA_POOL_PAUSE();
          g = gen_.load(std::memory_order_acquire);
          if ((spin & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) break;
        }
      }
      if (g == seen) {
        std::unique_lock<std::mutex> lk(mu_);
        sleepers_.fetch_add(1, std::memory_order_seq_cst);
        cv_.wait(lk, [&] { return gen_.load(std::memory_order_seq_cst) != seen; });
        sleepers_.fetch_sub(1, std::memory_order_seq_cst);
        g = gen_.load(std::memory_order_acquire);
      }
      seen = g;
      if (stop_) return;
      const uint32_t nt = nw_ + 1;
      try {
        for (uint32_t i = w + 1; i < n_; i += nt) (*fn_)(i);
      } catch (...) {   // (recorded, answered all the same: run() must not wait for an acknowledgement that never comes)
        std::lock_guard<std::mutex> lk(mu_);
        if (!error_) error_ = std::current_exception();
      }
      acks_[w].gen.store(g, std::memory_order_release);
    }
  }

  std::vector<std::thread> th_;
  Ack* acks_;
  uint32_t nw_;
  int spin_us_ = 300;
  std::exception_ptr error_;   // the first exception a worker's job threw during the current run (under mu_)
  int next_cpu_ = -1;
  std::vector<int> mine_;   // the CPUs this pool has claimed: its workers', then next_cpu()
  static std::mutex& claims_mu() { static std::mutex m; return m; }
  static std::vector<int>& claims() { static std::vector<int> v; return v; }   // CPUs claimed by the pools of this process
  std::mutex mu_;
  std::condition_variable cv_;
  alignas(64) std::atomic<uint64_t> gen_{0};
  alignas(64) std::atomic<uint32_t> sleepers_{0};
  bool stop_ = false;
  const std::function<void(uint32_t)>* fn_ = nullptr;
  uint32_t n_ = 0;
};
