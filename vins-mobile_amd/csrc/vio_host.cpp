// vio_host.cpp — host-side (IMU-rate) part of the back-end ABI: IMU pre-integration.
//
// Reference: IntegrationBase::{push_back, propagate, midPointIntegration} (VINS_ios/integration_base.h:39-169),
// called per IMU sample from VINS::processIMU (VINS_ios/VINS.cpp:333-358). This runs on the caller's thread at IMU
// rate exactly like the reference's; its result (VioPreintegration) is an input of the device solve.
#include <math.h>
#include <string.h>

#include "vio_amd.h"
#include "vio_math.h"
#include "vio_preint.h"

using namespace vio;

extern "C" int vio_preintegrate(const VioConfig *cfg, const double acc_0[3], const double gyr_0[3], const double ba[3],
                                const double bg[3], int32_t n, const double *dt, const double *acc, const double *gyr,
                                VioPreintegration *out) {
  if (!cfg || !acc_0 || !gyr_0 || !ba || !bg || n < 0 || !out || (n > 0 && (!dt || !acc || !gyr))) return VIO_EINVAL;
  host::Preint ib;
  host::preint_init(ib, cfg, acc_0, gyr_0, ba, bg);
  for (int i = 0; i < n; i++) host::propagate(ib, dt[i], acc + 3 * i, gyr + 3 * i);
  host::preint_export(ib, out);
  return VIO_OK;
}
