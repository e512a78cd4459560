// vio_pool.h — a small persistent host thread pool for the per-sequence / per-window host work around a batched launch
// (landmark bookkeeping, window packing, result unpacking). The reference runs ONE sequence on one thread
// (ViewController.mm:688-724); a batch of hundreds of sequences per GPU needs the same work done for all of them
// within the ~2 ms one launch takes, so it is spread over host cores. VIO_AMD_HOST_THREADS overrides the width
// (1 = everything inline on the caller's thread). A pool runs one parallel region at a time; host threads that drive
// different contexts (several estimator objects sharing a GPU) are spread over two pools on hosts with >= 128
// hardware threads, so that the host phases of one overlap those of another as well as its kernels
// (VIO_AMD_HOST_POOLS overrides the count, at most 4).
#pragma once
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vio {

class HostPool {
 public:
  static HostPool &get() {  // the calling thread's pool: assigned round-robin at its first call
    static Pools pools;
    static std::atomic<unsigned> next_caller{0};
    thread_local const unsigned mine = next_caller.fetch_add(1);
    return pools.p[mine % pools.n];
  }
  int width() const { return (int)workers_.size() + 1; }

  // fn(i) for i in [0, n), dynamically distributed; returns when all are done. One parallel region at a time: a caller
  // that finds the pool busy (another host thread driving another context) waits for its turn when the loop is long and
  // runs it inline when it is short. fn must not call parallel_for itself.
  void parallel_for(int n, const std::function<void(int)> &fn) {
    if (n <= 0) return;
    if (workers_.empty() || n < 4) {
      for (int i = 0; i < n; i++) fn(i);
      return;
    }
    std::unique_lock<std::mutex> region(region_m_, std::defer_lock);
    if (n >= 32) {
      region.lock();
    } else if (!region.try_lock()) {
      for (int i = 0; i < n; i++) fn(i);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn, n_ = n, next_.store(0), pending_ = (int)workers_.size(), generation_++;
    }
    cv_.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  struct Pools {
    int n = 1;
    HostPool *p = nullptr;
    Pools() {
      const int hw = (int)std::thread::hardware_concurrency();
      if (const char *e = getenv("VIO_AMD_HOST_POOLS")) n = atoi(e);
      else n = hw >= 128 ? 2 : 1;
      n = std::min(4, std::max(1, n));
      p = new HostPool[n];
    }
    ~Pools() { delete[] p; }
  };
  HostPool() {
    int t = 0;
    if (const char *e = getenv("VIO_AMD_HOST_THREADS")) t = atoi(e);
    if (t <= 0) t = std::min(32, std::max(1, (int)std::thread::hardware_concurrency() / 2));
    for (int i = 1; i < t; i++) workers_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (std::thread &w : workers_) w.join();
  }
  void drain() {
    const int chunk = std::max(1, n_ / (8 * width()));
    while (true) {
      const int i0 = next_.fetch_add(chunk);
      if (i0 >= n_) break;
      for (int i = i0; i < std::min(n_, i0 + chunk); i++) (*fn_)(i);
    }
  }
  void loop() {
    unsigned seen = 0;
    while (true) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      drain();
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_, region_m_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)> *fn_ = nullptr;
  std::atomic<int> next_{0};
  int n_ = 0, pending_ = 0;
  unsigned generation_ = 0;
  bool stop_ = false;
};

}  // namespace vio
