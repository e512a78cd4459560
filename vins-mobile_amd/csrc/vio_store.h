// vio_store.h — launchers of the landmark-store kernels (vio_store.hip, store_core.h) for the back-end's resident path.
#pragma once

#include <hip/hip_runtime.h>

#include "batch.h"
#include "preint_core.h"
#include "store_core.h"

namespace vio {

// Device pointers of the resident state of one back-end context (all of its slots; slot = window index of the batch).
struct StoreDev {
  store::Dims d;
  int n_slots;
  int *fid, *start, *nobs, *flag;  // [2 banks][n_slots][Lcap]
  double *depth, *obs;             // [2][n_slots][Lcap], [2][n_slots][Lcap][P][3]
  int *ctl;                        // [n_slots][store::C_COUNT]
  double *ctld;                    // [n_slots][store::kCtlDoubles]
  // this frame's inputs (device mirror of the staging arena)
  const int *active;               // [n_slots] the slot takes part in this frame
  const int *n_obs;                // [n_slots]
  const VioObs *obs_in;            // the frames' observations, packed; slot s starts at obs_off[s]
  const int *obs_off;              // [n_slots]
  const double *Ps, *Rs;           // [n_slots][P][3], [n_slots][P][9]: the window states triangulate reads
  const double *tic, *ric;         // camera -> body
  double *preint;                  // [n_slots][W][kPreintDoubles] the pre-integration blocks of the window, kept across frames
  const int *pre_idx;              // [n_pre] slot * W + interval of every block that arrives with this frame
  const double *pre_blk;           // [n_pre][kPreintDoubles]
  int n_pre;
  long long *prof;                 // null, or [32] cycle stamps written by slot 0's workgroups
  // IMU samples to integrate on the device (preint_core.h): n_jobs records of 16 doubles
  //   slot * W + interval | 1: a new interval, 0: more samples for the one in place | samples | first sample | acc_0 gyr_0 ba bg
  // and the samples they name, 7 doubles each (dt, acc, gyr)
  // relocalization frames (store_core.h LoopIn): per slot the window frame (-1: none), the number of matched ids and where
  // they start in the packed id / observation arrays
  const int *loop_frame, *loop_n, *loop_off, *loop_ids;
  const double *loop_xy;
  const double *imu_jobs, *imu_samples;
  int n_imu_jobs;
  double *pre_side;                // [n_slots][W][preint::kSide] last sample of every interval (what a continuation starts from)
  const double *noise;             // [18] diagonal of the IMU noise (integration_base.h:30-36)
};

int store_launch_imu(const StoreDev &S, hipStream_t st);
int store_launch_ingest(const StoreDev &S, hipStream_t st);
int store_launch_pack(const StoreDev &S, const BatchPtrs &B, int chunk, hipStream_t st);
int store_launch_finish(const StoreDev &S, const BatchPtrs &B, hipStream_t st);
size_t store_pack_lds_bytes(const store::Dims &d, int Mcap);

}  // namespace vio
