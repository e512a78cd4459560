/*
 * vio_amd.h — C ABI of the MI355X-native VIO hot path.
 *
 * This is the drop-in boundary behind VINS-Mobile's two per-frame entry points
 * (all citations are into /root/reference/VINS_ios unless another root is named):
 *
 *   front-end  FeatureTracker::readImage            feature_tracker.hpp:59, feature_tracker.cpp:162-310
 *   back-end   VINS::processIMU / VINS::solve_ceres  VINS.hpp:153,163-164, VINS.cpp:333-375,480-831
 *
 * Conventions
 *   - plain pointers and sizes, caller-owned host buffers, no C++/torch types;
 *   - every function returns VIO_OK (0) or a negative VIO_E* code, never throws;
 *   - one context per sequence (or per batch); contexts are thread-compatible,
 *     not thread-safe (the reference objects are not re-entrant either:
 *     static n_id feature_tracker.cpp:11, static sqrt_info projection_facor.cpp:11);
 *   - all matrices are row-major; quaternions are stored x y z w exactly like
 *     para_Pose (VINS.cpp:93-101);
 *   - the product path needs a gfx950 device: there is no CPU fallback inside
 *     this library (the CPU restatement lives in oracle/ and is test-only);
 *   - DEVICE BINDING: a context lives on the HIP device that is current on the
 *     creating thread at *_create (hipSetDevice(k) before the call; default 0).
 *     Every later call on the context may come from ANY host thread — the
 *     reference calls readImage on the camera-callback thread and solve_ceres
 *     on the mainLoop thread (ViewController.mm:458 vs :688-724) — the entry
 *     point switches the calling thread to the context's device for the call
 *     and restores the thread's previous device on return. One process per GPU
 *     or one context per GPU in one process both work; vio_*_get_device reports
 *     the binding.
 */
#ifndef VIO_AMD_H
#define VIO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIO_ABI_VERSION 1

#define VIO_OK 0
#define VIO_EINVAL (-1)   /* bad argument / inconsistent sizes            */
#define VIO_ENODEV (-2)   /* no gfx950 device / HIP runtime error         */
#define VIO_ENOMEM (-3)   /* host or device allocation failed             */
#define VIO_ECAP (-4)     /* problem exceeds the context's capacity       */
#define VIO_ESTATE (-5)   /* call order violated                          */
#define VIO_ETIMEOUT (-6) /* the workgroups of a cooperative window did not
                             meet on the device (co-residency lost): the
                             window's termination reads 2 (FAILURE), its
                             outputs and next prior must be discarded      */

#define VIO_SIZE_POSE 7       /* global_param.hpp:30 */
#define VIO_SIZE_SPEEDBIAS 9  /* global_param.hpp:31 */
#define VIO_MAX_PRIOR_BLOCKS 96

/* ------------------------------------------------------------------------- */
/* Runtime configuration (the reference's compile-time macros and per-device
 * globals: global_param.hpp:23-58, global_param.cpp:24-131,
 * feature_tracker.hpp:24-29).                                                */
typedef struct VioConfig {
  int32_t window_size;    /* WINDOW_SIZE (10); frames in window P = W+1       */
  int32_t max_features;   /* capacity of inv_depth (NUM_OF_F = 1000)          */
  int32_t max_factors;    /* capacity of the projection-factor list           */
  int32_t max_iterations; /* options.max_num_iterations (10) VINS.cpp:645     */
  int32_t image_rows;     /* ROW (640)                                        */
  int32_t image_cols;     /* COL (480)                                        */
  int32_t max_corners;    /* MAX_CNT (70)                                     */
  int32_t min_dist;       /* MIN_DIST (30)                                    */
  int32_t freq;           /* FREQ (3): publish every freq-th frame            */
  int32_t lk_win;         /* LK window (21)        feature_tracker.cpp:181    */
  int32_t lk_levels;      /* maxLevel (3)          feature_tracker.cpp:181    */
  int32_t lk_max_iters;   /* TermCriteria COUNT (30) OpenCV default           */
  double lk_eps;          /* TermCriteria EPS (0.01)                          */
  double lk_min_eig;      /* minEigThreshold (1e-4)                           */
  double quality_level;   /* goodFeaturesToTrack qualityLevel (0.01) :263     */
  double f_threshold;     /* F_THRESHOLD (1.0 px)                             */
  double f_confidence;    /* RANSAC confidence (0.99)                         */
  double fx, fy, cx, cy;  /* FOCUS_LENGTH_X/Y, PX, PY                         */
  double gravity;         /* GRAVITY 9.805                                    */
  double acc_n, acc_w, gyr_n, gyr_w; /* ACC_N 0.5, ACC_W 2e-3, GYR_N 0.2, GYR_W 4e-5 */
  double cauchy_a;        /* CauchyLoss(1.0)       VINS.cpp:485               */
} VioConfig;

/* Fills *cfg with the reference's iPhone7P values at W=10
 * (global_param.cpp:27-42, feature_tracker.hpp:24-29).                       */
void vio_config_default(VioConfig *cfg);

/* ------------------------------------------------------------------------- */
/* IMU pre-integration between two frames: the public state of
 * IntegrationBase (integration_base.h:200-221) after the last push_back.     */
typedef struct VioPreintegration {
  double sum_dt;
  double delta_p[3];
  double delta_q[4]; /* x y z w */
  double delta_v[3];
  double linearized_ba[3];
  double linearized_bg[3];
  double jacobian[225];   /* 15x15 row-major, order O_P,O_R,O_V,O_BA,O_BG     */
  double covariance[225]; /* 15x15 row-major                                  */
} VioPreintegration;

/* Linearized prior = the kept side of MarginalizationInfo
 * (marginalization_factor.hpp:66-88): r = r0 + J0*dx.                        */
#define VIO_BLOCK_POSE 0
#define VIO_BLOCK_SPEEDBIAS 1
#define VIO_BLOCK_EXPOSE 2
typedef struct VioPrior {
  int32_t n;        /* residual rows = sum of kept local sizes (info->n)      */
  int32_t n_blocks; /* keep_block_size.size()                                 */
  int32_t block_kind[VIO_MAX_PRIOR_BLOCKS];   /* VIO_BLOCK_*                  */
  int32_t block_index[VIO_MAX_PRIOR_BLOCKS];  /* frame index the block is
                          bound to in the *next* window (after addr_shift,
                          VINS.cpp:760-769); 0 for the extrinsic             */
  int32_t block_offset[VIO_MAX_PRIOR_BLOCKS]; /* keep_block_idx - m           */
  double *block_x0;              /* [n_blocks][9] keep_block_data, 7- and
                                    9-sized blocks left-aligned               */
  double *linearized_jacobians;  /* [n][n] row-major                          */
  double *linearized_residuals;  /* [n]                                       */
} VioPrior;

#define VIO_MARGIN_OLD 0        /* VINS.hpp MarginalizationFlag               */
#define VIO_MARGIN_SECOND_NEW 1
#define VIO_MARGIN_NONE 2       /* skip the marginalization step              */

/* One sliding window as solve_ceres sees it after old2new() (VINS.cpp:505).  */
typedef struct VioWindow {
  int32_t window_size; /* W */
  int32_t n_features;  /* rows of inv_depth in use = getFeatureCount()        */
  int32_t n_factors;   /* M projection factors, grouped by feature in
                          ascending feature order as VINS.cpp:528-567 emits
                          them, loop factors (target == W+1) included          */
  int32_t marginalization_flag; /* VIO_MARGIN_*                                */
  double *pose;        /* [(W+1)][7] para_Pose        in: initial, out: see below */
  double *speed_bias;  /* [(W+1)][9] para_SpeedBias                            */
  double *ex_pose;     /* [7] para_Ex_Pose[0] (constant block)                 */
  double *inv_depth;   /* [n_features] para_Feature                            */
  const int32_t *factor_host;    /* [M] imu_i                                  */
  const int32_t *factor_target;  /* [M] imu_j; W+1 selects loop_pose           */
  const int32_t *factor_feature; /* [M] feature_index                          */
  const double *factor_pts_i;    /* [M][3] */
  const double *factor_pts_j;    /* [M][3] */
  const VioPreintegration *preint; /* [W]; preint[k] links frame k -> k+1
                                      (pre_integrations[k+1], VINS.cpp:516-521) */
  const VioPrior *prior;         /* NULL when last_marginalization_info == nullptr */
  int32_t loop_frame;  /* -1: no loop constraint; else window index i whose
                          pose initialises loop_pose (VINS.cpp:590-596)         */
  double *loop_pose;   /* [7] front_pose.loop_pose, out: optimised              */
  /* new2old() gauge anchor (VINS.cpp:133-155). 0: use pose[0] as passed in.    */
  int32_t use_origin_override;
  double origin_yaw_deg; /* Utility::R2ypr(last_R_old).x()                      */
  double origin_p[3];    /* last_P_old                                          */
  /* outputs --------------------------------------------------------------- */
  /* pose / speed_bias / inv_depth are overwritten with the state after
   * new2old() re-expressed in para_* form (what the second old2new() at
   * VINS.cpp:693 produces). raw_* (optional, may be NULL) receive the arrays
   * exactly as ceres::Solve left them.                                        */
  double *raw_pose;       /* [(W+1)][7] or NULL */
  double *raw_speed_bias; /* [(W+1)][9] or NULL */
  double *raw_inv_depth;  /* [n_features] or NULL */
  VioPrior *next_prior;   /* caller-allocated buffers sized for
                             vio_prior_capacity(W); NULL to skip               */
  /* Device-resident prior chain (vio_backend_reserve_priors). 0: the prior
   * travels through host memory as described above. k >= 1: slot k-1 of the
   * back-end's prior store. Then (a) a `prior` whose three data pointers are
   * NULL names the prior the slot holds (n, n_blocks and the block_* arrays
   * still come from the struct); a `prior` with data pointers is uploaded as
   * usual; (b) the next prior is written into the slot instead of host
   * memory: `next_prior` receives n, n_blocks and the block_* arrays only
   * (its data pointers may be NULL), and the slot is advanced when n > 0.   */
  int32_t resident_prior;
} VioWindow;

#define VIO_MAX_TRACE 64
typedef struct VioSolveStats {
  double initial_cost;
  double final_cost;     /* summary.final_cost VINS.cpp:660 */
  int32_t iterations;    /* iteration records incl. iteration 0 */
  int32_t termination;   /* 0 NO_CONVERGENCE, 1 CONVERGENCE, 2 FAILURE */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  /* per-iteration trace (IterationSummary, CS/include/ceres/iteration_callback.h) */
  double it_cost[VIO_MAX_TRACE];
  double it_radius[VIO_MAX_TRACE];
  double it_step_norm[VIO_MAX_TRACE];
  double it_relative_decrease[VIO_MAX_TRACE];
  double it_gradient_max_norm[VIO_MAX_TRACE];
  int32_t it_flags[VIO_MAX_TRACE]; /* bit0 step_is_valid, bit1 step_is_successful */
} VioSolveStats;

/* Upper bound of prior dimension for a window of size W: every pose (6),
 * every speed-bias (9) and the extrinsic (6).                                */
int32_t vio_prior_capacity(int32_t window_size);

/* ------------------------------------------------------------------------- */
/* Back-end                                                                   */
typedef struct vio_backend vio_backend_t;

/* max_batch = number of independent windows one launch may carry.            */
int vio_backend_create(const VioConfig *cfg, int32_t max_batch, vio_backend_t **out);
/* HIP device ordinal the context was created on (see DEVICE BINDING above). */
int vio_backend_get_device(const vio_backend_t *be, int32_t *device);
void vio_backend_destroy(vio_backend_t *be);

/* IntegrationBase(acc_0, gyr_0, ba, bg) followed by n push_back(dt, acc, gyr)
 * (integration_base.h:20-45, VINS.cpp:333-358). Host-side, IMU-rate path.     */
int vio_preintegrate(const VioConfig *cfg, const double acc_0[3], const double gyr_0[3],
                     const double ba[3], const double bg[3], int32_t n,
                     const double *dt, const double *acc, const double *gyr,
                     VioPreintegration *out);

/* solve_ceres(buf_num) for `n` independent windows in one device launch
 * (VINS.cpp:480-831). buf_num only selected a wall-clock budget in the
 * reference (VINS.cpp:648-653); it is accepted and ignored (no time limit, so
 * results are deterministic).                                                */
int vio_backend_solve_windows(vio_backend_t *be, VioWindow *windows, int32_t n,
                              int32_t buf_num, VioSolveStats *stats /* [n] or NULL */);

/* Device-resident prior chain. The prior a window leaves behind (the output of
 * marginalize(), VINS.cpp:697-831) is the next window's
 * last_marginalization_info (VINS.cpp:523-528): a caller that chains windows
 * frame after frame only needs its header on the host. This reserves
 * `n_slots` slots of two banks each in device memory (2 x ~46 KB per slot at
 * W=10); VioWindow.resident_prior selects a slot, see there. Reserving again
 * forgets what the slots held. One slot belongs to one chain: two windows of
 * one batch must not name the same slot (VIO_EINVAL).                        */
int vio_backend_reserve_priors(vio_backend_t *be, int32_t n_slots);

/* Resident-batch API for throughput runs: pack + upload once, launch many
 * times from the same initial state, download when wanted. `stream` is a
 * hipStream_t (or NULL for the library's own stream).                        */
int vio_backend_upload(vio_backend_t *be, const VioWindow *windows, int32_t n);
int vio_backend_launch(vio_backend_t *be, void *stream);
int vio_backend_sync(vio_backend_t *be);
int vio_backend_download(vio_backend_t *be, VioWindow *windows, int32_t n, VioSolveStats *stats);
/* Average device time (ms) of the solve kernel over the launches since the
 * last call, measured with HIP events on the launch stream.                  */
int vio_backend_kernel_ms(vio_backend_t *be, double *ms_avg, int32_t *launches);

/* Per-stage device cycle counters of the solve kernel: the kernel-side counterpart
 * of the reference's TS()/TE() printf timers (global_param.hpp:85-92, VINS.cpp:657-662,
 * 753-758). Enable before vio_backend_upload; read after a launch. Stage order:
 * setup_imu, setup_prior, eval_prior, eval_imu, eval_proj, scale, schur, rhs,
 * cholesky, tri_solve, quad_form, dogleg, cost_eval, new2old, marg_build,
 * marg_chol, total (shader-clock cycles, thread 0 of the window's workgroup).
 * Asking for more than VIO_N_STAGES entries also returns the finer sub-stage
 * breakdown listed in csrc/solver_core.h (Stage enum) / vins-mobile_amd/abi.py. */
#define VIO_N_STAGES 17
int vio_backend_set_profile(vio_backend_t *be, int32_t enable);
int vio_backend_stage_cycles(vio_backend_t *be, int32_t window, int64_t *cycles, int32_t n_stages);

/* ------------------------------------------------------------------------- */
/* Front-end                                                                  */
typedef struct vio_frontend vio_frontend_t;

typedef struct VioObs {      /* one entry of image_msg (feature_tracker.cpp:300-306) */
  int32_t id;
  double x, y, z;            /* ((u-PX)/fx, (v-PY)/fy, 1) */
} VioObs;

typedef struct VioTrackViz { /* good_pts / track_len (UI only), optional */
  float *good_pts;           /* [cap][2] */
  double *track_len;         /* [cap]    */
  int32_t cap;
  int32_t n;
} VioTrackViz;

/* n_seq independent trackers (sequences) share one context and one launch.
 * VIO_EINVAL for configurations the kernels are not laid out for: lk_win != 21,
 * more than 512 corners, images below 32 x 32 or above 32767 rows / 65535
 * columns (corner candidates carry their position as y << 16 | x).
 * Environment, read here: VIO_AMD_DETECT_ALWAYS=1 (measurement aid) runs the
 * corner detector on every published frame of every sequence; by default a
 * sequence that still tracks max_corners features skips it, like the
 * reference's n_max_cnt > 0 test (feature_tracker.cpp:256-266) -- the results
 * are the same either way.                                                    */
int vio_frontend_create(const VioConfig *cfg, int32_t n_seq, vio_frontend_t **out);
int vio_frontend_get_device(const vio_frontend_t *fe, int32_t *device);
void vio_frontend_destroy(vio_frontend_t *fe);

/* readImage for sequence `seq` (feature_tracker.cpp:162-310). `publish` is the
 * caller's img_cnt == 0 (ViewController.mm:467,494). out_obs has room for
 * cfg->max_corners entries.                                                  */
int vio_frontend_read_image(vio_frontend_t *fe, int32_t seq, const uint8_t *gray,
                            int32_t rows, int32_t cols, int32_t stride, double header,
                            int32_t publish, VioObs *out_obs, int32_t *n_obs,
                            VioTrackViz *viz /* may be NULL */);

/* Batched form: one frame for every sequence in one set of launches.
 * gray = n_seq images, each rows*stride bytes, back to back.                 */
int vio_frontend_read_images(vio_frontend_t *fe, const uint8_t *gray, int32_t rows,
                             int32_t cols, int32_t stride, const double *headers,
                             int32_t publish, VioObs *out_obs /* [n_seq][max_corners] */,
                             int32_t *n_obs /* [n_seq] */);

/* The same in two halves, for callers that overlap the front-end of frame k+1
 * with the estimator of frame k (the app runs readImage and processImage on
 * two threads: ViewController.mm:458 and :688-724). submit gathers the frames,
 * queues the transfer, the kernels and the copy of the observations and returns
 * without waiting for the device; collect waits and hands the observations
 * over. One frame in flight per context: a second submit before collect is
 * VIO_ESTATE, as is collect without submit.                                    */
int vio_frontend_submit_images(vio_frontend_t *fe, const uint8_t *gray, int32_t rows,
                               int32_t cols, int32_t stride, int32_t publish);
/* vio_frontend_submit_images with the submit's own host work (gathering the frames into page-locked memory, queueing
 * transfers and kernels) on a host thread the context owns: returns at once; `gray` must stay valid and unchanged until
 * vio_frontend_collect, which also reports an error of the submit. */
int vio_frontend_submit_images_async(vio_frontend_t *fe, const uint8_t *gray, int32_t rows, int32_t cols, int32_t stride,
                                     int32_t publish);
int vio_frontend_collect(vio_frontend_t *fe, VioObs *out_obs /* [n_seq][max_corners] */,
                         int32_t *n_obs /* [n_seq] */);

/* Host frame buffers registered once (camera / decoder ring buffers, the cv::Mat
 * storage behind `_img` in FeatureTracker::readImage, feature_tracker.cpp:162):
 * the range is page-locked in place, and read_images / submit_images(_async)
 * whose `gray` lies inside a registered range (with stride == cols) send the
 * frames to the device by DMA from where they are, without the gathering pass
 * through the library's own page-locked staging. Results are the same either
 * way. Process-wide (every context and device sees the registration);
 * overlapping a registered range or unregistering an unknown pointer is
 * VIO_ESTATE, pages that cannot be locked VIO_ENOMEM. Unregister (with the
 * pointer that was registered) before the memory is freed; it waits for the
 * device first.                                                               */
int vio_host_register(void *ptr, size_t bytes);
int vio_host_unregister(void *ptr);

/* Resident form for throughput runs: frames already in HBM.                  */
int vio_frontend_upload_frames(vio_frontend_t *fe, const uint8_t *gray, int32_t n_frames,
                               int32_t rows, int32_t cols, int32_t stride);
int vio_frontend_step_resident(vio_frontend_t *fe, int32_t frame_index, int32_t publish,
                               void *stream);
int vio_frontend_sync(vio_frontend_t *fe);
int vio_frontend_kernel_ms(vio_frontend_t *fe, double *ms_avg, int32_t *launches);

/* The image pre-step of the camera callback, batched on the device
 * (ViewController.mm:432-437): cv::cvtColor(CV_RGBA2GRAY) then CLAHE with
 * clipLimit 3 on the default 8x8 grid; its output is what readImage receives.
 * channels = 4 (RGBA as UIImageToMat delivers it) or 1 (already gray).         */
typedef struct vio_preprocess vio_preprocess_t;
int vio_preprocess_create(int32_t max_frames, int32_t rows, int32_t cols, vio_preprocess_t **out);
void vio_preprocess_destroy(vio_preprocess_t *p);
int vio_preprocess_set_clahe(vio_preprocess_t *p, double clip_limit, int32_t tiles_x, int32_t tiles_y);
/* Host buffers: pixels = n_frames images of rows*stride bytes back to back;
 * equalized_out (and gray_out, may be NULL) = n_frames * rows * cols bytes.    */
int vio_preprocess_run(vio_preprocess_t *p, const uint8_t *pixels, int32_t channels, int32_t n_frames, int32_t stride,
                       uint8_t *gray_out, uint8_t *equalized_out);
/* Resident form: device pointers, asynchronous on `stream` (a hipStream_t, NULL =
 * the context's own); d_equalized is packed [n_frames][rows][cols], ready for
 * vio_frontend_step_resident-style consumers.                                  */
int vio_preprocess_run_resident(vio_preprocess_t *p, const void *d_pixels, int32_t channels, int32_t n_frames,
                                int32_t stride, void *d_equalized, void *stream);
int vio_preprocess_sync(vio_preprocess_t *p);
int vio_preprocess_kernel_ms(vio_preprocess_t *p, double *ms_avg, int32_t *launches);

/* Introspection of tracker state (cur_pts / ids / track_cnt,
 * feature_tracker.hpp:72-75) for parity tests.                               */
int vio_frontend_get_state(vio_frontend_t *fe, int32_t seq, float *cur_pts /* [cap][2] */,
                           int32_t *ids, int32_t *track_cnt, int32_t cap, int32_t *n);
/* forw_pts / ids at the point of readImage where solveVinsPnP joins them with the
 * solved landmarks (feature_tracker.cpp:207): the last frame's tracked points behind
 * the first findFundamentalMat rejection (:194-205), AHEAD of the publish-frame
 * steps rejectWithF (:235) and setMask (:255), which drop more of them.           */
int vio_frontend_get_pnp_points(vio_frontend_t *fe, int32_t seq, float *forw_pts /* [cap][2] */, int32_t *ids,
                                int32_t cap, int32_t *n);
/* Measured iteration counts of calcOpticalFlowPyrLK's inner loop (the reference passes
 * TermCriteria(COUNT + EPS, 30, 0.01), feature_tracker.cpp:181; how many iterations a
 * level takes is data dependent): enable = 1 / 0 switches the counters of the LK kernel
 * on / off (-1: leave), iterations / visits (may both be NULL) receive, per pyramid level,
 * the iterations run and the (feature, level) visits since the last read and are reset.
 * Diagnostics for the roofline's byte formula (bench.py: roofline_frontend.lk_mean_iterations);
 * with the counters on the kernel adds two atomics per (feature, level).               */
int vio_frontend_lk_iterations(vio_frontend_t *fe, int32_t enable, uint64_t *iterations, uint64_t *visits, int32_t levels_cap);
/* While a frame submitted with vio_frontend_submit_images has not been collected,
 * every other entry point that reads or changes the tracker state or the
 * observation staging (step_resident, get_state, get_pnp_points, set / update /
 * get_tracks, a second submit) returns VIO_ESTATE.                                */

/* The tracker's public fields (feature_tracker.hpp:68-80: pre_pts / cur_pts /
 * forw_pts, ids, track_cnt) of one sequence in and out, and the steps of readImage
 * between the LK call and goodFeaturesToTrack on their own
 * (feature_tracker.cpp:183-205 status && inBorder, reduceVector, F-RANSAC; on
 * publish frames also :235-255 rejectWithF, track_cnt++ and :50-87 setMask), for
 * every sequence of the context. After the update forw_pts / ids / track_cnt
 * hold what the step kept (in setMask's order on publish frames).               */
int vio_frontend_set_tracks(vio_frontend_t *fe, int32_t seq, int32_t n, const float *pre_pts,
                            const float *cur_pts, const float *forw_pts, const int32_t *ids,
                            const int32_t *track_cnt, const uint8_t *lk_status);
int vio_frontend_update_tracks(vio_frontend_t *fe, int32_t publish);
int vio_frontend_get_tracks(vio_frontend_t *fe, int32_t seq, float *forw_pts, int32_t *ids,
                            int32_t *track_cnt, int32_t cap, int32_t *n);

/* Stand-alone operators of the front-end (each is one reference call site),
 * exposed so they can be parity-tested in isolation:
 *   calcOpticalFlowPyrLK  feature_tracker.cpp:181
 *   goodFeaturesToTrack   feature_tracker.cpp:263
 *   findFundamentalMat    feature_tracker.cpp:95,198                          */
int vio_klt_track(const VioConfig *cfg, const uint8_t *prev, const uint8_t *next,
                  int32_t rows, int32_t cols, int32_t stride, const float *prev_pts,
                  int32_t n, float *next_pts, uint8_t *status, float *err);
int vio_good_features(const VioConfig *cfg, const uint8_t *img, const uint8_t *mask,
                      int32_t rows, int32_t cols, int32_t stride, int32_t max_corners,
                      float *corners /* [max_corners][2] */, int32_t *n_corners);
int vio_fundamental_ransac(const VioConfig *cfg, const float *pts1, const float *pts2,
                           int32_t n, uint8_t *inlier_mask);

/* ------------------------------------------------------------------------- */
/* Loop-closure producer, descriptor side:
 *   KeyFrame::searchByDes               loop/keyframe.cpp:161-187
 *   KeyFrame::HammingDis                loop/keyframe.cpp:368-373
 *   KeyFrame::rejectWithF               loop/keyframe.cpp:35-58
 *   KeyFrame::findConnectionWithOldFrame loop/keyframe.cpp:267-273
 * A BRIEF descriptor (BRIEF::bitset, 256 bits) is four uint64_t words, word 0 =
 * bits 0..63. The matcher context is bound to a device like every other
 * context.                                                                     */
typedef struct vio_matcher vio_matcher_t;
int vio_matcher_create(vio_matcher_t **out);
void vio_matcher_destroy(vio_matcher_t *m);
/* searchByDes for n_pairs (current keyframe, old keyframe) pairs in one launch.
 * cur_desc holds the pairs' window descriptors back to back (sum n_cur x 4
 * words), old_desc the old keyframes' descriptors (sum n_old x 4, at most 65535
 * per keyframe). Per query, in the order of cur_desc: best_index = index into
 * that pair's old list of the smallest Hamming distance (first one on ties, as
 * the reference's `dis < bestDist` scan), best_dist = that distance;
 * best_index = -1 / best_dist = 256 when nothing is closer than 256 bits
 * (empty old list).                                                            */
int vio_matcher_search_by_des(vio_matcher_t *m, int32_t n_pairs, const int32_t *n_cur,
                              const int32_t *n_old, const uint64_t *cur_desc,
                              const uint64_t *old_desc, int32_t *best_index,
                              int32_t *best_dist);
/* findConnectionWithOldFrame: searchByDes, matched_old_pts[i] = keypoint of the
 * best old descriptor (pixels), then, from 8 matches on, rejectWithF =
 * findFundamentalMat(cur_pts, matched_old_pts, FM_RANSAC, 2.0, 0.99) ->
 * status[i] (1 = kept), matched_old_norm = (pt - (cx, cy)) / (fx, fy)
 * (optional). Fewer than 8 matches keep everything.                            */
int vio_loop_find_connection(vio_matcher_t *m, const VioConfig *cfg, int32_t n_cur,
                             const uint64_t *cur_desc, const float *cur_pts /* [n_cur][2] */,
                             int32_t n_old, const uint64_t *old_desc,
                             const float *old_pts /* [n_old][2] */,
                             float *matched_old_pts /* [n_cur][2] */,
                             float *matched_old_norm /* [n_cur][2] or NULL */,
                             uint8_t *status /* [n_cur] */, int32_t *n_inliers);

/* Bag-of-words query (DBoW2 as LoopClosure::startLoopClosure drives it, loop/loop_closure.cpp:20-36 ->
 * TemplatedLoopDetector::detectLoop, loop/TemplatedLoopDetector.h:668-700):
 *   vocabulary file layout             loop/VocabularyBinary.hpp:17-50 (k, L, scoringType, weightingType, nNodes,
 *                                      nWords; Node{nodeId, parentId, weight, descriptor[4]}; Word{nodeId, wordId}),
 *                                      TemplatedVocabulary::loadBin  ThirdParty/DBoW/TemplatedVocabulary.h:1505-1554
 *   TemplatedVocabulary::transform     TemplatedVocabulary.h:1213-1253 (descriptor -> word id, weight),
 *                                      :1061-1117 (descriptors of a keyframe -> BowVector, L1-normalised)
 *   TemplatedDatabase::add / query     ThirdParty/DBoW/TemplatedDatabase.h:439-470, 603-720 (queryL1)
 * Only L1_NORM scoring (scoringType 0, the app's vocabulary) is implemented; other vocabularies are refused with
 * VIO_EINVAL. Contexts are bound to the device current at creation.                                                  */
typedef struct vio_vocabulary vio_vocabulary_t;
int vio_vocabulary_create(const void *blob, size_t bytes, vio_vocabulary_t **out);  /* the file's bytes            */
int vio_vocabulary_load(const char *path, vio_vocabulary_t **out);                  /* TemplatedVocabulary(filename) */
void vio_vocabulary_destroy(vio_vocabulary_t *v);
int vio_vocabulary_info(const vio_vocabulary_t *v, int32_t info[6]);  /* k, L, scoring, weighting, nodes, words */
int vio_vocabulary_get_device(const vio_vocabulary_t *v, int32_t *device);
/* transform for n_keyframes keyframes in one launch. desc: the keyframes' descriptors back to back (sum n_desc x 4
 * words, vio_matcher_* layout; at most 8192 per keyframe). word_id / word_weight (optional): transform(feature, id, w)
 * per descriptor. bow_*: the BowVector of keyframe f in [f * bow_stride, ...): bow_count[f] entries, ascending word id.
 * VIO_ECAP when a keyframe has more distinct words than bow_stride (bow_count[f] = -(needed)).                        */
int vio_vocabulary_transform(vio_vocabulary_t *v, int32_t n_keyframes, const int32_t *n_desc, const uint64_t *desc,
                             int32_t *word_id, double *word_weight, int32_t *bow_count, int32_t *bow_word,
                             double *bow_value, int32_t bow_stride);
typedef struct vio_bow_database vio_bow_database_t;
/* The database takes what it needs from the vocabulary (device, word count) at create and
 * owns its stream: it stays valid after vio_vocabulary_destroy, and a vocabulary and its
 * database may be used from two threads at the same time. A vocabulary file whose node
 * records do not form one tree under node 0 (an id twice, a parent cycle) is refused
 * with VIO_EINVAL by vio_vocabulary_create / _load.                                  */
int vio_bow_database_create(vio_vocabulary_t *v, int32_t max_entries, int32_t max_total_words, vio_bow_database_t **out);
void vio_bow_database_destroy(vio_bow_database_t *d);
int vio_bow_database_size(const vio_bow_database_t *d, int32_t *n_entries);
/* TemplatedDatabase::add(BowVector) -> entry id = number of entries before the call.                                  */
int vio_bow_database_add(vio_bow_database_t *d, int32_t n, const int32_t *word, const double *value, int32_t *entry_id);
/* TemplatedDatabase::query(BowVector, ret, max_results, max_id) for n_queries BowVectors in one launch: entries with
 * id < max_id[q] (all if -1) that share a word with the query, best first, at most max_results (all if <= 0);
 * score in [0, 1] = 1 - ||v - w||_1 / 2. Equal scores come out in ascending entry id.                                 */
int vio_bow_database_query(vio_bow_database_t *d, int32_t n_queries, const int32_t *bow_count, const int32_t *bow_word,
                           const double *bow_value, int32_t bow_stride, const int32_t *max_id, int32_t max_results,
                           int32_t *n_results, int32_t *entry, double *score, int32_t result_stride);

/* Keyframe descriptor extraction: BriefExtractor::operator() (loop/keyframe.cpp:395-409) =
 * cv::FAST(im, keys, 20, true); keys += window_pts; DVision::BRIEF::compute
 * (ThirdParty/DVision/BRIEF.cpp:40-105: GaussianBlur 9x9 sigma 2, then n_bits
 * intensity tests im(pt + (x1,y1)) < im(pt + (x2,y2)) per keypoint, a test with
 * an end outside the image leaves its bit 0). The pattern is the app's
 * Resources/brief_pattern.yml (BriefExtractor::BriefExtractor, :375-393).
 * A batch of keyframes per call. keypoints[f] = the FAST corners in cv::FAST's
 * order (at most max_keypoints - n_window[f] are kept: VIO_ECAP says some were
 * cut) followed by the frame's window points; descriptors [f][k][4] words, bit
 * i of a descriptor = bit (i & 63) of word i >> 6 (vio_matcher_* layout).      */
typedef struct vio_brief vio_brief_t;
int vio_brief_load_pattern(const char *yml_path, int32_t *x1, int32_t *y1, int32_t *x2, int32_t *y2, int32_t cap,
                           int32_t *n);
int vio_brief_create(int32_t rows, int32_t cols, int32_t max_frames, int32_t max_keypoints, const int32_t *x1,
                     const int32_t *y1, const int32_t *x2, const int32_t *y2, int32_t n_bits, vio_brief_t **out);
int vio_brief_get_device(const vio_brief_t *b, int32_t *device);
void vio_brief_destroy(vio_brief_t *b);
int vio_brief_extract(vio_brief_t *b, const uint8_t *gray /* [n_frames][rows*cols] */, int32_t n_frames,
                      const float *window_pts /* [n_frames][window_stride][2] */, const int32_t *n_window,
                      int32_t window_stride, int32_t fast_threshold,
                      float *keypoints /* [n_frames][max_keypoints][2] */,
                      uint64_t *descriptors /* [n_frames][max_keypoints][4] */, int32_t *n_fast,
                      int32_t *n_keypoints);

/* 4-DoF loop pose graph: KeyFrameDatabase::optimize4DoFLoopPoseGraph
 * (VINS_ios/loop/keyfame_database.cpp:140-353). Per keyframe the unknowns are
 * yaw (degrees, AngleLocalParameterization) and translation; pitch and roll of
 * the odometry pose are kept. Edges: FourDOFError (keyfame_database.h:271-313)
 * under HuberLoss(1.0) to up to five preceding kept keyframes (:232-262), and
 * FourDOFWeightError (:315-366, weight 10, no loss) for every loop (:264-285).
 * Solver: Ceres trust region, Levenberg-Marquardt, max_num_iterations = 5
 * (:154-159; DENSE_SCHUR there is an exact linear solve).
 *
 * The graph is given in resample-index order: node k = k-th keyframe from
 * earliest_loop_index on. `skip[k]` = need_resample (:176-198): the keyframe
 * keeps its parameter blocks but gets no edge, so it is not part of the solve
 * and is moved by the drift of the last kept keyframe afterwards (:316-319).   */
typedef struct VioPoseGraph {
  int32_t n_nodes;
  double *t;            /* [n][3] in: origin translation; out: optimized (kept nodes) */
  double *ypr;          /* [n][3] in: R2ypr(origin rotation), degrees; out: [k][0] = optimized yaw */
  int32_t fixed_node;   /* SetParameterBlockConstant: the earliest_loop_index keyframe (:224-228) */
  int32_t n_edges;
  const int32_t *edge_i;   /* first pair of parameter blocks: the earlier / connected keyframe */
  const int32_t *edge_j;   /* second pair: the keyframe the edge was created for */
  const uint8_t *edge_kind; /* 0: sequential (Huber), 1: loop (weighted, no loss) */
  const double *edge_meas; /* [n_edges][6]: t_x, t_y, t_z, relative_yaw, pitch_i, roll_i */
} VioPoseGraph;

/* Keyframe list as optimize4DoFLoopPoseGraph walks it (from earliest_loop_index to cur_index). */
typedef struct VioPoseGraphKeyframe {
  double origin_t[3], origin_r[9]; /* getOriginPose (VIO odometry) */
  double t[3], r[9];               /* getPose (current, drift-corrected) — read by the resampling only */
  int32_t global_index;
  int32_t has_loop, is_looped;
  int32_t loop_index;              /* global_index of the matched old keyframe (has_loop) */
  double loop_info[8];             /* relative_t (0..2), relative_q (3..6), relative_yaw (7) */
} VioPoseGraphKeyframe;

typedef struct vio_posegraph vio_posegraph_t;
/* max_nodes / max_edges bound one graph; n_graphs = graphs one call may carry (one workgroup each). */
int vio_posegraph_create(int32_t max_nodes, int32_t max_edges, int32_t n_graphs, vio_posegraph_t **out);
int vio_posegraph_get_device(const vio_posegraph_t *pg, int32_t *device);
void vio_posegraph_destroy(vio_posegraph_t *pg);
/* The ceres::Solve of :287: n graphs in one launch; t / ypr are updated in place, stats[g] carries the trace.
 * max_iterations is 5 in the reference (:159); values above VIO_MAX_TRACE - 1 are clamped to it.               */
int vio_posegraph_optimize(vio_posegraph_t *pg, VioPoseGraph *graphs, int32_t n, int32_t max_iterations,
                           VioSolveStats *stats);
/* Host side of :166-285: resampling flags and the edge list from a keyframe list (kf[0] = earliest_loop_index,
 * kf[n_kf-1] = cur_index). Arrays are caller-owned: t/ypr [n_kf][3], skip [n_kf], edges up to cap_edges.
 * total_length, max_frame_num, list_size: the database's fields (keyfame_database.cpp:16-17,34,185).          */
int vio_posegraph_build(const VioPoseGraphKeyframe *kf, int32_t n_kf, double total_length, int32_t max_frame_num,
                        int32_t list_size, double *t, double *ypr, uint8_t *skip, int32_t cap_edges,
                        int32_t *edge_i, int32_t *edge_j, uint8_t *edge_kind, double *edge_meas, int32_t *n_edges);
/* Host side of :303-339: poses after the solve (kept keyframes take the optimized pose, skipped ones the drift of
 * the last kept one) and the drift of the current keyframe (yaw_drift, r_drift, t_drift).                      */
int vio_posegraph_apply(const VioPoseGraphKeyframe *kf, int32_t n_kf, const double *t, const double *ypr,
                        const uint8_t *skip, double *out_t /* [n_kf][3] */, double *out_r /* [n_kf][9] */,
                        double *yaw_drift, double *r_drift /* [9] */, double *t_drift /* [3] */);

/* ------------------------------------------------------------------------- */
/* Window bookkeeping around the solve (host side): FeatureManager            */
/* (VINS_ios/feature_manager.hpp:71-103). It decides which landmarks and      */
/* observations become factors of a VioWindow; the window size is a run-time  */
/* parameter (global_param.hpp:28 fixes WINDOW_SIZE = 10).                    */
typedef struct vio_features vio_features_t;

typedef struct VioFeatureInfo { /* FeaturePerId (feature_manager.hpp:47-69), list order */
  int32_t id, start_frame, n_obs, used_num;
  int32_t solve_flag;            /* 0 not solved yet, 1 ok, 2 negative depth (setDepth) */
  int32_t is_outlier, fixed;
  double estimated_depth;        /* -1: not triangulated yet */
} VioFeatureInfo;

int vio_features_create(int32_t window_size, vio_features_t **out);
void vio_features_destroy(vio_features_t *fm);
int vio_features_clear(vio_features_t *fm);                       /* clearState  feature_manager.cpp:315 */
/* addFeatureCheckParallax feature_manager.cpp:103-155. obs = image_msg of one
 * published frame (unique ids; taken in ascending id like the std::map).
 * enough_parallax = its return value (true -> MARGIN_OLD, VINS.cpp:397-400).  */
int vio_features_add_check_parallax(vio_features_t *fm, int32_t frame_count, const VioObs *obs, int32_t n_obs,
                                    int32_t *enough_parallax, int32_t *parallax_num, int32_t *last_track_num);
int vio_features_count(vio_features_t *fm, int32_t *n);           /* getFeatureCount :284 */
int vio_features_get_depth_vector(vio_features_t *fm, double *inv_depth, int32_t cap, int32_t *n); /* :270 */
int vio_features_set_depth(vio_features_t *fm, const double *inv_depth, int32_t n);                /* :300 */
int vio_features_clear_depth(vio_features_t *fm, const double *inv_depth, int32_t n);              /* :176 */
/* triangulate :189-248. Ps [W+1][3], Rs [W+1][9] row-major (body -> world),
 * tic [3], ric [9] row-major (camera -> body).                               */
int vio_features_triangulate(vio_features_t *fm, const double *Ps, const double *Rs, const double tic[3],
                             const double ric[9]);
int vio_features_remove_failures(vio_features_t *fm);             /* :259 */
int vio_features_remove_back(vio_features_t *fm);                 /* :320 */
int vio_features_remove_back_shift_depth(vio_features_t *fm, const double marg_R[9], const double marg_P[3],
                                         const double new_R[9], const double new_P[3]);            /* :250 */
int vio_features_remove_front(vio_features_t *fm, int32_t frame_count);                             /* :343 */
/* The factor enumeration of solve_ceres (VINS.cpp:528-567): fills the factor
 * arrays a VioWindow points to, landmark index = row of the depth vector.     */
int vio_features_export_factors(vio_features_t *fm, int32_t cap_factors, int32_t *host, int32_t *target,
                                int32_t *feature, double *pts_i /* [cap][3] */, double *pts_j /* [cap][3] */,
                                int32_t *n_factors, int32_t *n_features);
/* As above plus the relocalization factors of solve_ceres (VINS.cpp:597-631): landmarks
 * observed in window frame `loop_frame` whose id appears in loop_ids (ascending,
 * RetriveData::features_ids) get one more factor with target W+1 (the loop pose)
 * and pts_j = (loop_xy, 1) (RetriveData::measurements); it closes the landmark's
 * group of factors. loop_frame = -1: no loop.                                  */
int vio_features_export_factors_loop(vio_features_t *fm, int32_t cap_factors, int32_t loop_frame,
                                     const int32_t *loop_ids, const double *loop_xy /* [n_loop][2] */, int32_t n_loop,
                                     int32_t *host, int32_t *target, int32_t *feature, double *pts_i, double *pts_j,
                                     int32_t *n_factors, int32_t *n_features, int32_t *n_loop_factors /* may be NULL */);
/* estimated_depth *= s for the landmarks of the solve (visualInitialAlign VINS.cpp:1079-1085). */
int vio_features_scale_depth(vio_features_t *fm, double s);
/* Introspection: per-landmark records (and, optionally, all observation points
 * [sum n_obs][3]) in list order. cap = 0: only the counts.                   */
int vio_features_dump(vio_features_t *fm, VioFeatureInfo *info, int32_t cap, int32_t *n, double *points,
                      int32_t cap_points, int32_t *n_points);
/* The inverse of vio_features_dump: the list becomes exactly these entries (list order; points [sum n_obs][3]). Used
 * when a sequence's list returns from the device-resident store (vio_estimator_set_resident). */
int vio_features_load(vio_features_t *fm, const VioFeatureInfo *info, int32_t n, const double *points);

/* failureDetection VINS.cpp:214-265 on the newest frame after a solve (the
 * failure_hand switch of the UI stays with the caller). reasons: bit mask.    */
#define VIO_FAIL_FEW_FEATURES 1   /* f_manager.last_track_num < 4              */
#define VIO_FAIL_GYR_BIAS 2       /* |Bgs[W]| > 1                              */
#define VIO_FAIL_TRANSLATION 4    /* |Ps[W] - last_P| > 1                      */
#define VIO_FAIL_Z_TRANSLATION 8  /* |Ps[W].z - last_P.z| > 0.5                */
#define VIO_FAIL_ROTATION 16      /* angle(Rs[W]^T last_R) > 40 "degrees"      */
int vio_failure_detection(int32_t last_track_num, const double Bg_newest[3], const double P_newest[3],
                          const double R_newest[9], const double last_P[3], const double last_R[9],
                          int32_t *reasons);

/* ------------------------------------------------------------------------- */
/* Estimator: class VINS after initialisation (VINS.hpp:47-200), host side, for
 * n_seq independent sequences whose window solves share ONE device launch.
 * processIMU VINS.cpp:333-375, processImage :377-478, old2new/new2old :89-212,
 * the relocalization bookkeeping :571-637,664-680, failureDetection :214-265,
 * slideWindow :1149-1273, clearState :36-81. solveInitial (:833-1145) is not
 * part of it: the caller hands the window states over instead
 * (vio_estimator_set_initial_state) and the branches after it are the
 * reference's (first solve, final_cost > 200 -> back to INITIAL).              */
typedef struct vio_estimator vio_estimator_t;

#define VIO_SOLVER_INITIAL 0      /* VINS::SolverFlag (VINS.hpp:60-64) */
#define VIO_SOLVER_NON_LINEAR 1

#define VIO_FRAME_SKIPPED 0      /* sequence not active in this call                               */
#define VIO_FRAME_FILLING 1      /* window not full yet: frame_count++            VINS.cpp:448-452 */
#define VIO_FRAME_WAIT_INIT 2    /* window full, no initial state: slid, no solve VINS.cpp:443-446 */
#define VIO_FRAME_INIT_FAILED 3  /* first solve ended with final_cost > 200       VINS.cpp:415-424 */
#define VIO_FRAME_SOLVED 4       /* solve_ceres + slideWindow                                      */
#define VIO_FRAME_FAILURE 5      /* failureDetection fired: state cleared         VINS.cpp:462-467 */
#define VIO_FRAME_RESET 6        /* track_num < 20 with a full INITIAL window     VINS.cpp:408-412 */
#define VIO_FRAME_ERROR 7        /* capacity / argument error for this sequence, see .error        */

typedef struct VioFrameResult {
  int32_t action;               /* VIO_FRAME_*                                   */
  int32_t error;                /* VIO_E* when action == VIO_FRAME_ERROR          */
  int32_t marginalization_flag; /* VIO_MARGIN_OLD = the frame is a keyframe       */
  int32_t failure_reasons;      /* VIO_FAIL_* mask                                */
  int32_t track_num;            /* f_manager.last_track_num                       */
  int32_t n_features, n_factors, n_loop_factors;
  VioSolveStats stats;          /* valid when a solve ran                         */
} VioFrameResult;

typedef struct VioEstimatorStatus {
  int32_t frame_count, solver_flag, marginalization_flag, failure_occur;
  int32_t prior_rows;           /* rows of last_marginalization_info (0: none)    */
  double final_cost;
  double r_drift[9], t_drift[3];                 /* VINS.hpp r_drift / t_drift    */
  double relative_t[3], relative_q[4], relative_yaw, loop_pose[7]; /* front_pose  */
  int32_t resident;             /* 1: the landmark list lives in the device store */
  int32_t reserved;
} VioEstimatorStatus;

/* tic [3], ric [9] row-major: the camera-to-body extrinsic (TIC_*, RIC_*).     */
int vio_estimator_create(const VioConfig *cfg, int32_t n_seq, const double tic[3], const double ric[9],
                         vio_estimator_t **out);
void vio_estimator_destroy(vio_estimator_t *est);
int vio_estimator_clear(vio_estimator_t *est, int32_t seq);                       /* clearState */
/* solveInitial (VINS.cpp:833-1145) inside process_image when the window is full and no
 * state was handed over: relative pose, global SfM, PnP of the in-between frames,
 * visual-inertial alignment. Off by default (0): the caller hands states over.
 * 1: relativePose as the reference computes it (vio_init_relative_pose_mode 0: five-point
 * RANSAC; a frame on which it draws a non-physical root fails and the next frame retries,
 * VINS.cpp:893-901); 2: the fit over all correspondences with the gyroscope's rotation as
 * tie-breaker (mode 1: succeeds on the first frame with enough parallax, planar scenes
 * included).                                                                      */
int vio_estimator_enable_initialization(vio_estimator_t *est, int32_t enable);
int vio_estimator_process_imu(vio_estimator_t *est, int32_t seq, double dt, const double acc[3],
                              const double gyr[3]);                                /* processIMU */
/* processIMU for all sequences in one call (spread over host threads): sequence q
 * gets n_samples[q] <= stride samples dt[q*stride + i], acc/gyr[(q*stride + i)*3]. */
int vio_estimator_process_imu_batch(vio_estimator_t *est, const int32_t *n_samples, int32_t stride, const double *dt,
                                    const double *acc, const double *gyr);
/* The states solveInitial would leave for the W+1 frames with these headers
 * (Ps [W+1][3], Rs [W+1][9], Vs, Bas, Bgs [W+1][3]); consumed by the next
 * process_image that finds the window full with exactly these headers.        */
int vio_estimator_set_initial_state(vio_estimator_t *est, int32_t seq, const double *headers, const double *Ps,
                                    const double *Rs, const double *Vs, const double *Bas, const double *Bgs);
/* retrive_pose_data = ... (ViewController.mm:964): the old keyframe matched to the
 * window frame with this header; ids ascending; n = 0 withdraws it.            */
int vio_estimator_set_relocalization(vio_estimator_t *est, int32_t seq, double header, const double P_old[3],
                                     const double Q_old[4] /* x y z w */, const int32_t *ids,
                                     const double *xy /* [n][2] */, int32_t n);
/* processImage of one published frame for every active sequence (active NULL =
 * all). obs of sequence q starts at obs + q*obs_stride and has n_obs[q] entries.
 * The window solves of all sequences go to the device in one launch.          */
int vio_estimator_process_images(vio_estimator_t *est, const VioObs *obs, const int32_t *n_obs, int32_t obs_stride,
                                 const double *headers, const uint8_t *active, VioFrameResult *results /* [n_seq] */);
int vio_estimator_process_image(vio_estimator_t *est, int32_t seq, const VioObs *obs, int32_t n_obs, double header,
                                VioFrameResult *result);
int vio_estimator_get_status(vio_estimator_t *est, int32_t seq, VioEstimatorStatus *st);
/* Ps/Rs/Vs/Bas/Bgs/Headers of the window (any pointer may be NULL).            */
int vio_estimator_get_window(vio_estimator_t *est, int32_t seq, double *Ps, double *Rs, double *Vs, double *Bas,
                             double *Bgs, double *headers);
/* update_loop_correction VINS.cpp:302-331: r_drift * Ps + t_drift, r_drift * Rs. */
int vio_estimator_get_corrected_window(vio_estimator_t *est, int32_t seq, double *correct_Ps, double *correct_Rs);
/* Wall time (ms) of the last process_image(s) call by phase: bookkeeping before
 * the solve, vio_backend_solve_windows (pack, upload, kernel, download), and
 * the bookkeeping after it.                                                    */
int vio_estimator_get_timing(vio_estimator_t *est, double ms[3]);
/* The sequence's landmark store (owned by the estimator), for introspection.   */
/* Device-resident landmark stores (on by default; 0 or VIO_AMD_RESIDENT=0 turn them off): a sequence that has reached the
 * NON_LINEAR state keeps its landmark list, its pre-integration blocks and its prior in device memory; per frame the host
 * sends the observations and the propagated window states (about 12 KB instead of about 125 KB per window), kernels do
 * addFeatureCheckParallax / triangulate / the factor list / setDepth / the slide (feature_manager.cpp:103-372), and the
 * results are the host path's bit for bit (relocalization factors included). Sequences use the host-side list while they
 * initialize, for a frame with more observations (or a relocalization frame with more matched ids) than a store slot takes,
 * and when vio_estimator_features() asks for the list. */
int vio_estimator_set_resident(vio_estimator_t *est, int32_t enable);
int vio_estimator_features(vio_estimator_t *est, int32_t seq, vio_features_t **fm);

/* ------------------------------------------------------------------------- */
/* Motion-only window of the front-end: vinsPnP (vins_pnp.hpp:45-91), the 30 Hz
 * pose the tracker can compute between two back-end solves
 * (FeatureTracker::solveVinsPnP feature_tracker.cpp:107-160, called from readImage
 * :207; USE_PNP is off by default, ViewController.mm:144).                       */
#define VIO_PNP_MAX_FRAMES 8   /* PNP_SIZE + 1 = 7 in the reference (global_param.hpp) */
typedef struct VioPnpWindow {  /* vinsPnP::solve_ceres vins_pnp.cpp:264-341 after old2new() */
  int32_t n_frames;            /* PNP_SIZE + 1                                         */
  double *pose;                /* [n][7] para_Pose   in: initial, out: solved          */
  double *speed;               /* [n][3] para_Speed                                    */
  const double *bias;          /* [n][6] para_Bias (Ba, Bg): constant blocks           */
  const uint8_t *fixed;        /* [n] find_solved: pose and speed constant (:279-284)  */
  const double *ex_pose;       /* [7] para_Ex_Pose[0], constant                        */
  const VioPreintegration *preint; /* [n-1]; preint[k] links frame k -> k+1            */
  const int32_t *feat_start;   /* [n+1] frame k owns factors feat_start[k..k+1)        */
  const double *observation;   /* [M][2] IMG_MSG_LOCAL::observation                    */
  const double *position;      /* [M][3] IMG_MSG_LOCAL::position (fixed 3D point)      */
  const int32_t *track_num;    /* [M]    IMG_MSG_LOCAL::track_num (weight / 10)        */
} VioPnpWindow;
typedef struct vio_pnp vio_pnp_t;
int vio_pnp_create(const VioConfig *cfg, int32_t max_batch, vio_pnp_t **out);
void vio_pnp_destroy(vio_pnp_t *p);
/* n independent windows in one device launch (one workgroup each).              */
int vio_pnp_solve_windows(vio_pnp_t *p, VioPnpWindow *windows, int32_t n, VioSolveStats *stats /* [n] or NULL */);
int vio_pnp_kernel_ms(vio_pnp_t *p, double *ms_avg, int32_t *launches);

/* The vinsPnP object around that solve (vins_pnp.cpp:16-382), n_seq sequences sharing one launch; what
 * FeatureTracker::solveVinsPnP drives inside readImage (feature_tracker.cpp:107-160).                   */
typedef struct VioPnpFeature {  /* IMG_MSG_LOCAL vins_pnp.hpp:36-41; lists ascending in id */
  int32_t id;
  double observation[2];        /* ((forw_pts.x - PX) / fx, (forw_pts.y - PY) / fy) feature_tracker.cpp:131 */
  double position[3];           /* the landmark as the back-end solved it (world frame)   */
  int32_t track_num;
} VioPnpFeature;
typedef struct VioVinsResult {  /* VINS_RESULT vins_pnp.hpp:27-34: the newest back-end state */
  double header;
  double Ba[3], Bg[3], P[3], R[9], V[3];
} VioVinsResult;
typedef struct vio_pnp_tracker vio_pnp_tracker_t;
int vio_pnp_tracker_create(const VioConfig *cfg, int32_t n_seq, int32_t pnp_size /* PNP_SIZE = 6 */, const double tic[3],
                           const double ric[9], vio_pnp_tracker_t **out);
void vio_pnp_tracker_destroy(vio_pnp_tracker_t *t);
int vio_pnp_tracker_clear(vio_pnp_tracker_t *t, int32_t seq);                                       /* clearState */
int vio_pnp_tracker_set_init(vio_pnp_tracker_t *t, int32_t seq, const VioVinsResult *r);           /* setInit    */
int vio_pnp_tracker_process_imu(vio_pnp_tracker_t *t, int32_t seq, double dt, const double acc[3], const double gyr[3]);
/* processImage(feature_msg, header, use_pnp) for every active sequence; P_out [n_seq][3] / R_out [n_seq][9] =
 * Ps / Rs[PNP_SIZE - 1] as solveVinsPnP returns them; solved[q] = 1 when a solve ran for sequence q.       */
int vio_pnp_tracker_process_images(vio_pnp_tracker_t *t, const VioPnpFeature *features, const int32_t *n_features,
                                   int32_t stride, const double *headers, int32_t use_pnp, const uint8_t *active,
                                   double *P_out, double *R_out, int32_t *solved);
/* feature_msg of solveVinsPnP (feature_tracker.cpp:121-134): solved landmarks (id, position, track_num; ascending id)
 * joined by id with the tracker's current ids / pixel positions (vio_frontend_get_state); observation =
 * ((x - PX) / fx, (y - PY) / fy).                                                                            */
int vio_pnp_match_features(const VioConfig *cfg, const int32_t *ids, const float *forw_pts /* [n_pts][2] */, int32_t n_pts,
                           const VioPnpFeature *solved, int32_t n_solved, VioPnpFeature *out, int32_t cap, int32_t *n_out);
int vio_pnp_tracker_get_window(vio_pnp_tracker_t *t, int32_t seq, double *Ps, double *Rs, double *Vs, double *headers,
                               uint8_t *find_solved, int32_t *frame_count);

/* ------------------------------------------------------------------------- */
/* Initialisation pieces (host side, one-off): VINS::solveInitial VINS.cpp:833-1145.
 * Exposed one by one so that each can be tested against its reference.         */
typedef struct VioInitFrame {  /* ImageFrame initial_aligment.hpp:24-39            */
  double header;
  double R[9];                 /* body attitude in the SfM frame (Q[i] * ric^T)   */
  double T[3];                 /* camera position in the SfM frame, unknown scale  */
  int32_t is_key_frame;
  int32_t n_samples;           /* IMU samples from the previous frame to this one  */
  const double *dt, *acc, *gyr; /* [n_samples], [n_samples][3], [n_samples][3]      */
  double acc_0[3], gyr_0[3];   /* the sample the interval starts from              */
} VioInitFrame;
/* VisualIMUAlignment initial_aligment.cpp:223-229 (solveGyroscopeBias, SolveScale,
 * RefineGravity). Bgs [(W+1)][3] in/out (+= delta_bg); g [3] out; x out =
 * [n_frames][3] body-frame velocities followed by the metric scale; ok = its
 * return value.                                                                */
int vio_visual_imu_alignment(const VioConfig *cfg, const double tic[3], const VioInitFrame *frames, int32_t n_frames,
                             int32_t window_size, double *Bgs, double g[3], double *x, int32_t *ok);

/* solveRelativeRT motion_estimator.cpp:200-236: pose of the second camera in the first
 * (R [9], unit t [3]) from n >= 9 normalized correspondences xy0/xy1 [n][2];
 * inliers = points in front of both cameras, ok = inliers > 10.                */
/* R_hint (may be NULL): roughly known rotation of the second camera in the first (e.g.
 * gyroscope), used only to pick between the two exact solutions of a planar scene. */
int vio_init_relative_pose(const double *xy0, const double *xy1, int32_t n, const double *R_hint /* [9] or NULL */,
                           double R[9], double t[3], int32_t *inliers, int32_t *ok);
/* The two routes to the relative pose. mode 0 = the reference's: cv::findEssentialMat(ll, rr)
 * (five-point minimal solver inside RANSAC, RNG((uint64)-1), threshold 1.0, confidence 0.999)
 * followed by cv::recoverPose's cheirality count, restated from OpenCV 3.0.0
 * (csrc/vio_fivepoint.cpp; parity unpinned: OpenCV is not in the tree). With these arguments the
 * threshold accepts every correspondence, so the result is the first essential matrix of the
 * first random sample -- whether it is the physical one is chance, in the reference too; its
 * caller retries on the next frame. mode 1 = vio_init_relative_pose's fit over all
 * correspondences (rotation hint for planar scenes).                                   */
int vio_init_relative_pose_mode(const double *xy0, const double *xy1, int32_t n, int32_t mode, const double *R_hint,
                                double R[9], double t[3], int32_t *inliers, int32_t *ok);
/* EMEstimatorCallback::runKernel (OpenCV 3.0.0 calib3d/five-point.cpp): the essential matrices
 * (row-major, unit Frobenius norm, x2^T E x1 = 0) of five correspondences
 * xy0 / xy1 [5][2]; E [10][9], n_models <= 10.                                          */
int vio_init_five_point(const double *xy0, const double *xy1, double *E, int32_t *n_models);
/* cv::recoverPose(E, points1, points2, R, t) with focal 1, pp (0, 0): the (R, t) of the four
 * decompositions of E (x2 ~ R x1 + t) with the most triangulated points in front of both
 * cameras and closer than 50; inliers = that count.                                     */
int vio_init_recover_pose(const double E[9], const double *xy0, const double *xy1, int32_t n, double R[9], double t[3],
                          int32_t *inliers);
/* cv::solvePnP(..., useExtrinsicGuess = true) with K = I as inital_sfm.cpp:57 and
 * VINS.cpp:982 call it: refines world->camera R [9], t [3] in place.            */
int vio_init_pnp(const double *pts3, const double *pts2, int32_t n, double R[9], double t[3], int32_t *ok);
/* GlobalSFM::triangulatePoint inital_sfm.cpp:5-21: linear two-view triangulation.
 * pose0 / pose1: 3x4 row-major [R | t], world -> camera; xy: normalized image
 * coordinates; point = the null vector of the 4x4 design matrix, dehomogenised. */
int vio_init_triangulate_point(const double pose0[12], const double pose1[12], const double xy0[2], const double xy1[2],
                               double point[3]);
/* The "full BA" that closes GlobalSFM::construct, inital_sfm.cpp:229-296: ceres
 * problem over c_rotation [frame_num][4] (w x y z, QuaternionParameterization),
 * c_translation [frame_num][3] (both world -> camera) and the points with
 * point_ok != 0; frame l's rotation and the translations of frames l and
 * frame_num-1 are constant; ReprojectionError3D residuals (inital_sfm.hpp:25-54),
 * DENSE_SCHUR, Levenberg-Marquardt with the Solver's default options. In/out:
 * c_rotation, c_translation, points. stats (may be NULL): the Solver::Summary
 * fields and the per-iteration trace. ok = CONVERGENCE || final_cost < 3e-3
 * (:279), the value construct() returns.                                        */
int vio_init_bundle_adjust(int32_t frame_num, int32_t l, double *c_rotation, double *c_translation, int32_t n_points,
                           double *points, const uint8_t *point_ok, const int32_t *feat_start, const int32_t *obs_frame,
                           const double *obs_xy, VioSolveStats *stats, int32_t *ok);
/* GlobalSFM::construct inital_sfm.cpp:117-316. Landmark j has observations
 * obs_frame/obs_xy[feat_start[j] .. feat_start[j+1]) (frame index, normalized xy).
 * Out: q [frame_num][4] (x y z w) and T [frame_num][3] = camera poses in frame l's
 * camera frame, points [n_features][3] with point_ok flags.                     */
int vio_init_sfm(int32_t frame_num, int32_t l, const double relative_R[9], const double relative_T[3], int32_t n_features,
                 const int32_t *feat_start, const int32_t *obs_frame, const double *obs_xy, double *q, double *T,
                 double *points, uint8_t *point_ok, int32_t *ok);

/* ------------------------------------------------------------------------- */
/* Replay I/O: the record / playback formats of the app and the IMU-image
 * association of its estimator thread (host side; PNG through the image's zlib). */
typedef struct VioImuMsg {     /* IMU_MSG ViewController.h:58-62 (56 bytes)      */
  double header;
  double acc[3];
  double gyr[3];
} VioImuMsg;

typedef struct VioKeyframeData { /* KEYFRAME_DATA loop/keyfame_database.h:22-27  */
  double header;
  double translation[3];
  double rotation[4];            /* Eigen::Quaterniond coefficient order x y z w */
} VioKeyframeData;

/* "IMU" file: back-to-back IMU_MSG, closed by a header == 0 record
 * (ViewController.mm:1120-1150,1505-1511,1614-1622). cap = 0: count only.      */
int vio_replay_read_imu(const char *path, VioImuMsg *out, int32_t cap, int32_t *n);
int vio_replay_write_imu(const char *path, const VioImuMsg *msgs, int32_t n);
/* "IMAGE_TIME/<index>": 8-byte timestamp; "IMAGE/<index>": PNG
 * (ViewController.mm:1634-1708). read_image returns the gray frame the camera
 * callback feeds the tracker before CLAHE: cv::cvtColor CV_RGBA2GRAY
 * (ViewController.mm:432-433); cap = bytes available in gray (VIO_ECAP with
 * rows/cols set when too small).                                               */
int vio_replay_read_image_time(const char *dir, uint64_t index, double *header);
int vio_replay_write_image_time(const char *dir, uint64_t index, double header);
int vio_replay_read_image(const char *dir, uint64_t index, uint8_t *gray, int64_t cap, int32_t *rows, int32_t *cols);
int vio_replay_decode_png_gray(const uint8_t *png, int64_t png_bytes, uint8_t *gray, int64_t cap, int32_t *rows,
                               int32_t *cols);
/* channels 1 (gray), 3 (RGB) or 4 (RGBA, what UIImagePNGRepresentation stores). */
int vio_replay_write_image(const char *dir, uint64_t index, const uint8_t *pixels, int32_t rows, int32_t cols,
                           int32_t channels);
int vio_replay_rgba_to_gray(const uint8_t *rgba, int32_t rows, int32_t cols, int32_t stride, uint8_t *gray);
/* Pose log: back-to-back KEYFRAME_DATA records.                                */
int vio_replay_read_keyframes(const char *path, VioKeyframeData *out, int32_t cap, int32_t *n);
int vio_replay_write_keyframes(const char *path, const VioKeyframeData *kf, int32_t n);

/* getMeasurements + send_imu (ViewController.mm:603-682): queue IMU samples and
 * published image_msg lists as they arrive; each call to _next hands out one
 * (IMU batch with header <= image header, image) pair in the reference's order,
 * with the dt send_imu would pass to processIMU. available = 0: wait for more.  */
typedef struct vio_measurements vio_measurements_t;
int vio_measurements_create(vio_measurements_t **out);
void vio_measurements_destroy(vio_measurements_t *q);
int vio_measurements_push_imu(vio_measurements_t *q, const VioImuMsg *msg);
int vio_measurements_push_image(vio_measurements_t *q, double header, const VioObs *obs, int32_t n_obs);
int vio_measurements_next(vio_measurements_t *q, VioImuMsg *imu, double *dt /* may be NULL */, int32_t cap_imu,
                          int32_t *n_imu, double *header, VioObs *obs, int32_t cap_obs, int32_t *n_obs,
                          int32_t *available);

const char *vio_version(void);
/* The HIP runtime(s) mapped into the calling process (files named libamdhip64*
 * in /proc/self/maps), ';'-separated, and their number. The library binds to
 * the copy the process loaded first (next to PyTorch: the one bundled with the
 * wheel). With more than one copy mapped the device contexts refuse to start
 * (VIO_ENODEV, both paths on stderr): each copy would keep its own device state. */
int vio_hip_runtime(char *path, int32_t cap, int32_t *n_runtimes);
/* Width (threads, the caller's included) of the process-wide host pool that spreads per-sequence host work
 * (window packing, IMU feeds, observation gathers: csrc/vio_pool.h) and the number of such pools. (None in the
 * reference: one sequence per phone.) Sized from the CPUs the process may run on, its cgroup CPU quota and -- one
 * process per GPU on a node -- divided by LOCAL_WORLD_SIZE; VIO_AMD_HOST_THREADS overrides. A launcher that is
 * not torchrun sets one of the two per rank. Creates the pools on first use. */
int vio_host_pool_width(int32_t *n_pools /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* VIO_AMD_H */
