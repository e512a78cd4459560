// vio_device.h — a context belongs to the HIP device that was current when it was created.
//
// HIP's current device is per host thread (default 0). The reference calls readImage on the camera-callback thread and
// solve_ceres on the mainLoop thread (VINS_ios/ViewController.mm:458 vs :688-724), and a multi-GPU host runs one
// context per device: every ABI entry therefore switches the calling thread to the context's device for the duration
// of the call (and puts the thread's previous device back), so a context works from any thread.
#pragma once
#include <hip/hip_runtime.h>

namespace vio {

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
