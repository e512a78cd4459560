// vio_device.h — a context belongs to the HIP device that was current when it was created.
//
// HIP's current device is per host thread (default 0). The reference calls readImage on the camera-callback thread and
// solve_ceres on the mainLoop thread (VINS_ios/ViewController.mm:458 vs :688-724), and a multi-GPU host runs one
// context per device: every ABI entry therefore switches the calling thread to the context's device for the duration
// of the call (and puts the thread's previous device back), so a context works from any thread.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace vio {

// The HIP runtime(s) mapped into this process: distinct files named libamdhip64* in /proc/self/maps. The library binds
// to whichever one the process loaded first (next to PyTorch that is the copy bundled with the wheel, not /opt/rocm's).
// Two different copies in one process each keep their own device state -- streams, allocations and kernels of one are
// invisible to the other -- so the contexts refuse to start in that case instead of failing in obscure ways later.
inline std::vector<std::string> hip_runtimes() {
  std::vector<std::string> out;
  FILE *f = fopen("/proc/self/maps", "r");
  if (!f) return out;
  char line[4096];
  while (fgets(line, sizeof(line), f)) {
    const char *lib = strstr(line, "libamdhip64");
    if (!lib) continue;
    const char *path = strchr(line, '/');
    if (!path) continue;
    std::string p(path);
    while (!p.empty() && (p.back() == '\n' || p.back() == ' ')) p.pop_back();
    bool seen = false;
    for (const std::string &q : out) seen = seen || q == p;
    if (!seen) out.push_back(p);
  }
  fclose(f);
  return out;
}
// The host-only translation units are built with $(HOST_ARCH) (csrc/Makefile: -mavx2 by default). On a host CPU without
// that instruction set the first such function would die with SIGILL; the *_create entries (built without the flag: every
// .hip file) refuse with VIO_ENODEV instead.
inline bool host_isa_ok() {
#if defined(VIO_HOST_NEEDS_AVX2) && !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
  static const bool ok = [] {
    __builtin_cpu_init();
    const bool have = __builtin_cpu_supports("avx2");
    if (!have) fprintf(stderr, "vio_amd: this host CPU has no AVX2 and the library's host code was built with it (rebuild with `make HOST_ARCH=`)\n");
    return have;
  }();
  return ok;
#else
  return true;
#endif
}
inline bool single_hip_runtime() {
  static const int n = [] {
    const std::vector<std::string> r = hip_runtimes();
    const char *allow = getenv("VIO_AMD_ALLOW_TWO_RUNTIMES");
    if (r.size() > 1 && allow && allow[0] == '1') return 1;  // (the caller knows what it is doing)
    if (r.size() > 1) {
      fprintf(stderr, "vio_amd: %zu different HIP runtimes are mapped into this process:\n", r.size());
      for (const std::string &p : r) fprintf(stderr, "vio_amd:   %s\n", p.c_str());
      fprintf(stderr, "vio_amd: refusing to create device contexts (load one runtime only, e.g. import torch before this library; "
                      "VIO_AMD_ALLOW_TWO_RUNTIMES=1 overrides)\n");
    }
    return (int)r.size();
  }();
  return n <= 1 && host_isa_ok();
}

inline int current_device() {
  int d = 0;
  return hipGetDevice(&d) == hipSuccess ? d : -1;
}

class DeviceScope {
 public:
  explicit DeviceScope(int device) {
    if (device < 0) return;
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
    if (prev_ != device) {
      ok_ = hipSetDevice(device) == hipSuccess;
      switched_ = ok_;
    }
  }
  ~DeviceScope() {
    if (switched_ && prev_ >= 0) (void)hipSetDevice(prev_);
  }
  bool ok() const { return ok_; }
  DeviceScope(const DeviceScope &) = delete;
  DeviceScope &operator=(const DeviceScope &) = delete;

 private:
  int prev_ = -1;
  bool switched_ = false, ok_ = true;
};

}  // namespace vio

// First statement of every ABI entry that takes a context.
#define VIO_ON_DEVICE_OF(ctx)              \
  vio::DeviceScope vio_dev_scope_((ctx)->device); \
  if (!vio_dev_scope_.ok()) return VIO_ENODEV
