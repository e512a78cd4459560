// vio_pool.h — a small persistent host thread pool for the per-sequence / per-window host work around a batched launch
// (landmark bookkeeping, window packing, result unpacking). The reference runs ONE sequence on one thread
// (ViewController.mm:688-724); a batch of hundreds of sequences per GPU needs the same work done for all of them
// within the ~2 ms one launch takes, so it is spread over host cores. VIO_AMD_HOST_THREADS overrides the width
// (1 = everything inline on the caller's thread). A pool runs one parallel region at a time; host threads that drive
// different contexts (several estimator objects sharing a GPU) are spread over two pools on hosts with >= 128
// hardware threads, so that the host phases of one overlap those of another as well as its kernels
// (VIO_AMD_HOST_POOLS overrides the count, at most 4). On a multi-socket host the workers of pool i stay on NUMA node
// i mod nodes (the staging buffers a context fills and the landmark stores it walks then live next to the threads that
// touch them; VIO_AMD_HOST_NUMA=0 leaves the workers unbound).
#pragma once
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
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
    Pools &pools = all();
    static std::atomic<unsigned> next_caller{0};
    thread_local const unsigned mine = next_caller.fetch_add(1);
    return pools.p[mine % pools.n];
  }
  static int count() { return all().n; }
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
  // CPUs of NUMA node `node` that this process may run on; false when there is no such node or none is allowed
  static bool node_cpus(int node, cpu_set_t *out) {
    char path[96];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    CPU_ZERO(out);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) {
      fclose(f);
      return false;
    }
    int a = 0, n_set = 0;
    while (fscanf(f, "%d", &a) == 1) {  // "0-63,128-191"
      int b = a, c = fgetc(f);
      if (c == '-' && fscanf(f, "%d", &b) == 1) c = fgetc(f);
      for (int i = a; i <= b && i < CPU_SETSIZE; i++)
        if (i >= 0 && CPU_ISSET(i, &allowed)) {
          CPU_SET(i, out);
          n_set++;
        }
      if (c != ',') break;
    }
    fclose(f);
    return n_set > 0;
  }
  // CPUs' worth of time the process's cgroup grants per period (v2: cpu.max, v1: cpu.cfs_quota_us / cpu.cfs_period_us); 0: no limit known
  static int cpu_quota() {
    long long q = 0, p = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char tok[32] = {0};
      if (fscanf(f, "%31s %lld", tok, &p) == 2 && tok[0] != 'm') q = atoll(tok);
      fclose(f);
    } else {
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lld", &q) != 1) q = 0;
        fclose(g);
      }
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(g, "%lld", &p) != 1) p = 0;
        fclose(g);
      }
    }
    if (q <= 0 || p <= 0) return 0;
    return (int)std::max<long long>(1, (q + p - 1) / p);
  }
  struct Pools {
    int n = 1;
    HostPool *p = nullptr;
    Pools() {
      const int hw = (int)std::thread::hardware_concurrency();
      if (const char *e = getenv("VIO_AMD_HOST_POOLS")) n = atoi(e);
      else n = hw >= 128 ? 2 : 1;
      n = std::min(4, std::max(1, n));
      p = new HostPool[n];
      const char *numa = getenv("VIO_AMD_HOST_NUMA");
      cpu_set_t set;
      int nodes = 0;
      while (nodes < 16 && node_cpus(nodes, &set)) nodes++;
      for (int i = 0; i < n; i++) {
        const bool bind = nodes > 1 && !(numa && numa[0] == '0') && node_cpus(i % nodes, &set);
        p[i].start(bind ? &set : nullptr);
      }
    }
    ~Pools() { delete[] p; }
  };
  static Pools &all() {
    static Pools pools;
    return pools;
  }
  HostPool() {}
  void start(const cpu_set_t *cpus) {
    int t = 0;
    if (const char *e = getenv("VIO_AMD_HOST_THREADS")) t = atoi(e);
    if (t <= 0) {
      // width from the CPUs this process may run on (a container often sees all of the host's but is bound to a few).
      // Measured on a 256-thread host, 256 sequences per frame: 4 threads 12.8 ms, 16: 6.4, 32: 5.1, 64: 5.0, 96: 5.3
      cpu_set_t allowed;
      CPU_ZERO(&allowed);
      int usable = (int)std::thread::hardware_concurrency();
      if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0 && CPU_COUNT(&allowed) > 0) usable = std::min(usable > 0 ? usable : 1 << 20, CPU_COUNT(&allowed));
      t = std::min(64, std::max(1, usable >= 64 ? usable / 4 : usable));
      // A container's CPU-time quota (cgroup cpu.max, "quota period" in microseconds) is a harder limit than the CPUs it
      // may run on: a pool wider than the quota runs ahead of it for a few periods and is then stopped until the next one
      // -- measured as 40-50 ms stalls every ten frames or so of a 512-sequence pipeline (64 threads per pool on a box that
      // grants 16 CPUs: 40 k camera frames/s; 16 threads: 74 k, no stalls).
      const int quota = cpu_quota();
      if (quota > 0) t = std::min(t, quota);
      // Several ranks of one node share the CPUs and the quota (one process per GPU under torchrun / mpirun): every
      // process takes its share, not all of it -- eight ranks of full-width pools bring the throttling stalls back times
      // eight. LOCAL_WORLD_SIZE is what torchrun exports; other launchers set VIO_AMD_HOST_THREADS per rank.
      int local_world = 1;
      for (const char *name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MV2_COMM_WORLD_LOCAL_SIZE"})
        if (const char *e = getenv(name)) {
          local_world = std::max(1, atoi(e));
          break;
        }
      t = std::max(1, t / local_world);
    }
    for (int i = 1; i < t; i++) {
      workers_.emplace_back([this] { loop(); });
      if (cpus) (void)pthread_setaffinity_np(workers_.back().native_handle(), sizeof(*cpus), cpus);  // best effort
    }
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
