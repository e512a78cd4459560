// vio_resident.h — the back-end's device-resident path (library-internal; the estimator is its only caller).
//
// A slot of a back-end context that has been `reserve`d keeps a sequence's landmark store (store_core.h), its
// pre-integration blocks and its marginalization prior in device memory from frame to frame. Per published frame the
// host sends what only it knows — the frame's observations, the window states after IMU propagation, the
// pre-integration blocks that changed — and three kernels around the window solve do what processImage / solve_ceres /
// slideWindow do with the landmark list on the host path (vio_window.cpp, vio_estimator.cpp):
//
//   begin -> stage (per slot, any thread) -> ingest        H2D, store_ingest, counts back
//         -> launch                                        layout from the counts, store_pack, window kernel, store_finish, results back
//         -> collect -> result (per slot)                  poses, statistics, keyframe / failure decisions, next prior header
//
// The slot index is the window index of the batch and the slot of the prior store (vio_backend_reserve_priors), so a
// sequence that moves between the host path (VioWindow.resident_prior = slot + 1) and this one keeps its prior where it is.
#pragma once

#include "vio_amd.h"

struct VioResidentResult {
  int32_t status;      // VIO_OK, or the error that stopped the frame for this slot (its store is then undefined: reload)
  int32_t marginalization_flag, track_num, parallax_num;
  int32_t n_features, n_factors, n_list;
  int32_t failure_reasons;  // failureDetection of the solved window (the store has cleared itself when non-zero)
  const double *pose;       // [W+1][7] solved window after new2old
  const double *speed_bias; // [W+1][9]
  int32_t n_loop_factors;   // relocalization factors of the window (0: its loop pose was not part of the solve)
  const double *raw_pose;   // [W+1][7] the solved window before new2old, and
  const double *loop_pose;  // [7] the solved loop pose: both only when some slot of the frame carried a relocalization frame
  VioSolveStats stats;
};

// How many contexts launch their window kernels on the device at the same time (default 2, the estimator's two groups):
// a launch gives every window half a CU once launches * windows exceed the CUs, a whole CU otherwise.
int vio_backend_set_peers(vio_backend_t *be, int32_t peers);
int vio_backend_resident_reserve(vio_backend_t *be, int32_t n_slots, int32_t list_cap, int32_t obs_cap, const double ex_pose[7],
                                 const double tic[3], const double ric[9]);
int vio_backend_resident_caps(const vio_backend_t *be, int32_t *list_cap, int32_t *obs_cap);
// The landmark list of a slot as vio_features_dump gives it (list order; points [sum n_obs][3]); last_P / last_R: the
// states failureDetection compares the next solve with.
int vio_backend_resident_load(vio_backend_t *be, int32_t slot, const VioFeatureInfo *info, int32_t n, const double *points,
                              const double last_P[3], const double last_R[9]);
// n slots at once (ascending slot numbers): one strided copy per array and run of consecutive slots.
int vio_backend_resident_load_batch(vio_backend_t *be, int32_t n, const int32_t *slots, const VioFeatureInfo *const *infos,
                                    const int32_t *counts, const double *const *points, const double *last_P /* [n][3] */,
                                    const double *last_R /* [n][9] */);
int vio_backend_resident_fetch(vio_backend_t *be, int32_t slot, VioFeatureInfo *info, int32_t cap, int32_t *n, double *points,
                               int32_t cap_points, int32_t *n_points);
int vio_backend_resident_begin(vio_backend_t *be);
// prior: the header of the slot's prior (n, blocks) or null; its data is in the slot of the prior store.
// loop_frame >= 0: the window frame a relocalization frame is matched to, with the matched landmark ids (ascending, at
// most 256) and their observations in the old keyframe (retrive_pose_data, VINS.cpp:571-631).
int vio_backend_resident_stage(vio_backend_t *be, int32_t slot, const VioObs *obs, int32_t n_obs, const double *Ps, const double *Rs,
                               const double *pose, const double *speed_bias, const VioPrior *prior, int32_t loop_frame,
                               const int32_t *loop_ids, const double *loop_xy, int32_t n_loop);
// An integrated block (+ the interval's last sample: what more samples would continue from).
int vio_backend_resident_stage_preint(vio_backend_t *be, int32_t slot, int32_t interval, const VioPreintegration *block,
                                      const double last_acc[3], const double last_gyr[3]);
// The raw IMU samples of an interval instead of its integrated block (integrated by a kernel, preint_core.h). fresh: a new
// interval starting from (acc_0, gyr_0) with linearization biases (ba, bg); otherwise more samples for the block in place
// (the interval that absorbed the departed frame's at a non-keyframe slide). VIO_ECAP: this frame's staging is full.
int vio_backend_resident_stage_imu(vio_backend_t *be, int32_t slot, int32_t interval, int32_t fresh, const double acc_0[3],
                                   const double gyr_0[3], const double ba[3], const double bg[3], int32_t n, const double *dt,
                                   const double *acc, const double *gyr);
int vio_backend_resident_ingest(vio_backend_t *be);
int vio_backend_resident_launch(vio_backend_t *be);
int vio_backend_resident_collect(vio_backend_t *be);
int vio_backend_resident_result(vio_backend_t *be, int32_t slot, VioResidentResult *r, VioPrior *next_prior_header);
