// vio_estimator.cpp — the estimator state machine around the device solve (host side), for n independent sequences.
//
// Reference: class VINS (VINS_ios/VINS.hpp:47-200): processIMU (VINS.cpp:333-375), processImage (:377-478), the host
// half of solve_ceres (old2new :89-129, the factor list :528-637, the loop bookkeeping :664-680 and new2old :168-189),
// failureDetection (:214-265), slideWindow / slideWindowOld / slideWindowNew (:1149-1273), clearState (:36-81).
// One reference VINS object = one sequence here; the solve of ALL sequences that have a frame to solve goes to the
// device in ONE launch of the window kernel (vio_backend_solve_windows), everything else is the same small host
// bookkeeping the reference does on its "mainLoop" thread.
//
// What is NOT here: solveInitial (VINS.cpp:833-1145). Where the reference calls it, the estimator
// takes the window states the caller handed over with vio_estimator_set_initial_state — same branch structure after it
// (first solve, final_cost > 200 check, fall back to INITIAL).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <new>
#include <vector>

#include "vio_amd.h"
#include "vio_resident.h"
#include "vio_math.h"
#include "vio_initial.h"
#include "vio_pool.h"
#include "vio_preint.h"

using namespace vio;

namespace {

struct Relocalization {  // RetriveData (VINS.hpp:28-45), the fields the solve reads and writes
  double header = -1;
  double P_old[3] = {0, 0, 0};
  Quat Q_old{0, 0, 0, 1};
  std::vector<int32_t> ids;
  std::vector<double> xy;
  double loop_pose[7] = {0, 0, 0, 0, 0, 0, 1};
  double relative_t[3] = {0, 0, 0};
  Quat relative_q{0, 0, 0, 1};
  double relative_yaw = 0;
};

struct PriorBuf {
  VioPrior p;
  std::vector<double> x0, J, r;
  // resident: only the header lives here, the data stays in the back-end's device store (vio_backend_reserve_priors)
  void init(int cap, bool resident) {
    memset(&p, 0, sizeof(p));
    if (resident) return;
    x0.assign((size_t)VIO_MAX_PRIOR_BLOCKS * 9, 0.0), J.assign((size_t)cap * cap, 0.0), r.assign(cap, 0.0);
    p.block_x0 = x0.data(), p.linearized_jacobians = J.data(), p.linearized_residuals = r.data();
  }
};

struct Sequence {
  int index = 0;  // position in the estimator = slot of the device-resident prior store
  int frame_count = 0, solver_flag = VIO_SOLVER_INITIAL, marginalization_flag = VIO_MARGIN_OLD;
  bool first_imu = false;
  int failure_occur = 0;
  double acc_0[3] = {0, 0, 0}, gyr_0[3] = {0, 0, 0};
  std::vector<double> Ps, Rs, Vs, Bas, Bgs, Headers;  // [W+1] x {3, 9, 3, 3, 3, 1}
  std::vector<host::Preint> pre;                      // pre_integrations[i]
  std::vector<char> pre_valid;
  std::vector<std::vector<double>> dt_buf, acc_buf, gyr_buf;
  std::vector<double> lin_acc, lin_gyr;  // [W+1][3] IntegrationBase::linearized_acc / linearized_gyr (the first sample)
  vio_features_t *fm = nullptr;
  PriorBuf prior[2];
  int cur_prior = 0;
  bool has_prior = false;
  double last_R[9], last_P[3], last_R_old[9], last_P_old[3];
  double r_drift[9], t_drift[3];
  Relocalization retrive, front;
  bool loop_enable = false;
  double final_cost = 0;
  // solveInitial's bookkeeping: every published frame since the window's oldest one, with the IMU interval that leads
  // to it integrated from zero biases (all_image_frame / tmp_pre_integration, VINS.cpp:402-404)
  std::map<double, init::Frame> all_image_frame;
  init::Frame tmp;           // the interval being integrated
  bool tmp_valid = false;
  double initial_timestamp = 0;
  double g[3] = {0, 0, 0};
  // initial window handed over in place of solveInitial
  bool init_pending = false;
  std::vector<double> init_headers, init_Ps, init_Rs, init_Vs, init_Bas, init_Bgs;
  // scratch of the window being solved
  std::vector<double> pose, sb, inv_depth, raw_pose, raw_sb, pts_i, pts_j;
  std::vector<int32_t> f_host, f_target, f_feat;
  std::vector<VioPreintegration> preint;
  double ex_pose[7], loop_pose[7];
  int loop_frame = -1, n_loop_factors = 0;
  int last_track_num = 0;
  // device-resident path (vio_resident.h): the landmark list, the pre-integration blocks and the prior of this sequence live
  // in its slot of the group's back-end; `fm` is empty meanwhile
  bool on_device = false;
  std::vector<char> pre_dirty;  // [W+1] pre[i] changed since the device last saw it
  // IMU samples integrated on the device (resident_imu): pre[i] keeps its header (linearization biases, first sample) and
  // the sample buffers, its Jacobian / covariance are not propagated on the host (pre_stale) until the sequence returns to
  // the host-side list; pre_merge[i]: trailing samples of interval i that the device has not seen yet (non-keyframe slide)
  std::vector<char> pre_stale;
  std::vector<int> pre_merge;
};

const double kI3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

}  // namespace

struct vio_estimator {
  VioConfig cfg;
  int W = 10, n_seq = 0;
  double tic[3], ric[9];
  std::vector<Sequence> seq;
  // Created at the first solve: IMU propagation and window filling need no device. The sequences are split into
  // n_groups contiguous groups with one back-end context (own stream, own resident batch, own prior store) each: while
  // the kernel of group g runs, the host packs group g + 1 and unpacks group g - 1. The window kernel occupies one CU
  // per window, so n_groups launches of n_seq / n_groups windows run side by side on the device.
  static constexpr int kMaxGroups = 8;
  vio_backend_t *be[kMaxGroups] = {nullptr};
  int n_groups = 1, group_size = 1;
  std::vector<VioWindow> windows;
  std::vector<VioSolveStats> stats;
  bool enable_init = false;
  bool init_relpose_fit = false;  // relativePose by the all-correspondence fit instead of the reference's five-point RANSAC
  // the marginalization prior of every sequence stays in device memory between launches; only its header comes back
  // (VIO_AMD_HOST_PRIORS=1: carry it through host memory instead, ~45 KB per sequence and direction)
  bool resident_priors = !(getenv("VIO_AMD_HOST_PRIORS") && getenv("VIO_AMD_HOST_PRIORS")[0] == '1');
  // Sequences in the NON_LINEAR state keep their landmark list on the device and have their windows assembled there
  // (on by default; vio_estimator_set_resident / VIO_AMD_RESIDENT=0 turn it off); needs the device-resident priors.
  bool resident = !(getenv("VIO_AMD_RESIDENT") && getenv("VIO_AMD_RESIDENT")[0] == '0');
  int res_list_cap = 0, res_obs_cap = 0;
  // VIO_AMD_RESIDENT_IMU=1: the IMU samples of their intervals travel instead of the integrated blocks and a kernel integrates
  // them (preint_core.h: the host's bits). Off by default: measured at 512 sequences the kernel takes what the host pool saves
  // (124 us against ~120 us of host::propagate on 16 AVX2 threads) and sits on the frame's critical path.
  bool resident_imu = getenv("VIO_AMD_RESIDENT_IMU") && getenv("VIO_AMD_RESIDENT_IMU")[0] == '1';
  std::vector<int> res_rc;  // per sequence: outcome of staging in the current call
  std::vector<int> solving;  // sequences of the current launch
  std::vector<VioWindow> staged;   // per sequence, built in parallel, compacted into `windows`
  std::vector<char> wants_solve;
  double ms_pre = 0, ms_solve = 0, ms_post = 0;  // wall time of the last process_images call, by phase
};

namespace {

void clear_state(vio_estimator *e, Sequence &s) {  // VINS::clearState (VINS.cpp:36-81)
  const int P = e->W + 1;
  for (int i = 0; i < P; i++) {
    memcpy(&s.Rs[9 * i], kI3, sizeof(kI3));
    for (int k = 0; k < 3; k++) s.Ps[3 * i + k] = s.Vs[3 * i + k] = s.Bas[3 * i + k] = s.Bgs[3 * i + k] = 0;
    s.pre_valid[i] = 0;
    s.dt_buf[i].clear(), s.acc_buf[i].clear(), s.gyr_buf[i].clear();
  }
  s.frame_count = 0;
  s.first_imu = false;
  s.solver_flag = VIO_SOLVER_INITIAL;
  s.has_prior = false;
  s.init_pending = false;
  s.all_image_frame.clear();
  s.tmp_valid = false;
  s.initial_timestamp = 0;
  s.on_device = false;
  std::fill(s.pre_stale.begin(), s.pre_stale.end(), 0), std::fill(s.pre_merge.begin(), s.pre_merge.end(), 0);
  vio_features_clear(s.fm);
}

void new_preintegration(vio_estimator *e, Sequence &s, int i) {
  host::preint_init(s.pre[i], &e->cfg, s.acc_0, s.gyr_0, &s.Bas[3 * i], &s.Bgs[3 * i]);
  memcpy(&s.lin_acc[3 * i], s.acc_0, 24), memcpy(&s.lin_gyr[3 * i], s.gyr_0, 24);
  s.pre_valid[i] = 1;
  s.pre_stale[i] = 0, s.pre_merge[i] = 0;
}

// IntegrationBase::repropagate (integration_base.h:47-61): the same samples again from new linearization biases
void repropagate(vio_estimator *e, Sequence &s, int i, const double ba[3], const double bg[3]) {
  if (!s.pre_valid[i]) return;
  host::preint_init(s.pre[i], &e->cfg, &s.lin_acc[3 * i], &s.lin_gyr[3 * i], ba, bg);
  for (size_t k = 0; k < s.dt_buf[i].size(); k++)
    host::propagate(s.pre[i], s.dt_buf[i][k], &s.acc_buf[i][3 * k], &s.gyr_buf[i][3 * k]);
}

// slideWindow (VINS.cpp:1149-1237) + slideWindowOld / slideWindowNew (:1239-1273)
void slide_window(vio_estimator *e, Sequence &s) {
  const int W = e->W;
  if (s.frame_count != W) return;
  if (s.marginalization_flag == VIO_MARGIN_OLD) {
    double back_R0[9], back_P0[3];
    memcpy(back_R0, &s.Rs[0], sizeof(back_R0)), memcpy(back_P0, &s.Ps[0], sizeof(back_P0));
    for (int i = 0; i < W; i++) {
      for (int k = 0; k < 9; k++) std::swap(s.Rs[9 * i + k], s.Rs[9 * (i + 1) + k]);
      std::swap(s.pre[i], s.pre[i + 1]), std::swap(s.pre_valid[i], s.pre_valid[i + 1]);
      std::swap(s.pre_stale[i], s.pre_stale[i + 1]), std::swap(s.pre_merge[i], s.pre_merge[i + 1]);
      s.dt_buf[i].swap(s.dt_buf[i + 1]), s.acc_buf[i].swap(s.acc_buf[i + 1]), s.gyr_buf[i].swap(s.gyr_buf[i + 1]);
      for (int k = 0; k < 3; k++)
        std::swap(s.lin_acc[3 * i + k], s.lin_acc[3 * (i + 1) + k]), std::swap(s.lin_gyr[3 * i + k], s.lin_gyr[3 * (i + 1) + k]);
      s.Headers[i] = s.Headers[i + 1];
      for (int k = 0; k < 3; k++)
        std::swap(s.Ps[3 * i + k], s.Ps[3 * (i + 1) + k]), std::swap(s.Vs[3 * i + k], s.Vs[3 * (i + 1) + k]);
    }
    // (Bas / Bgs are not rotated by the reference's loop: only the newest slot is overwritten below — restated as is;
    // the solve rewrites all of them from para_SpeedBias every frame)
    s.Headers[W] = s.Headers[W - 1];
    memcpy(&s.Rs[9 * W], &s.Rs[9 * (W - 1)], 72);
    for (int k = 0; k < 3; k++) {
      s.Ps[3 * W + k] = s.Ps[3 * (W - 1) + k], s.Vs[3 * W + k] = s.Vs[3 * (W - 1) + k];
      s.Bas[3 * W + k] = s.Bas[3 * (W - 1) + k], s.Bgs[3 * W + k] = s.Bgs[3 * (W - 1) + k];
    }
    new_preintegration(e, s, W);
    s.dt_buf[W].clear(), s.acc_buf[W].clear(), s.gyr_buf[W].clear();
    if (s.solver_flag == VIO_SOLVER_INITIAL)  // frames older than the new oldest one leave all_image_frame (VINS.cpp:1190-1197)
      s.all_image_frame.erase(s.all_image_frame.begin(), s.all_image_frame.lower_bound(s.Headers[0]));
    // slideWindowOld: the landmarks hosted in the departed frame move to the next one (store_finish did it on the device
    // for a resident sequence, together with the shift of the pre-integration blocks)
    if (s.on_device) {
    } else if (s.solver_flag == VIO_SOLVER_NON_LINEAR) {
      double R0[9], R1[9], P0[3], P1[3], t[3];
      mat3mul(back_R0, e->ric, R0), mat3mul(&s.Rs[0], e->ric, R1);
      mat3vec(back_R0, e->tic, t);
      for (int k = 0; k < 3; k++) P0[k] = back_P0[k] + t[k];
      mat3vec(&s.Rs[0], e->tic, t);
      for (int k = 0; k < 3; k++) P1[k] = s.Ps[k] + t[k];
      vio_features_remove_back_shift_depth(s.fm, R0, P0, R1, P1);
    } else {
      vio_features_remove_back(s.fm);
    }
  } else {
    // the second-newest frame leaves; its IMU samples extend the interval of the frame before it
    const int fc = s.frame_count;
    const bool dev_imu = s.on_device && e->resident_imu;
    if (dev_imu) s.pre_stale[fc - 1] = 1, s.pre_merge[fc - 1] += (int)s.dt_buf[fc].size();
    for (size_t i = 0; i < s.dt_buf[fc].size(); i++) {
      const double dt = s.dt_buf[fc][i];
      if (s.pre_valid[fc - 1] && !dev_imu) host::propagate(s.pre[fc - 1], dt, &s.acc_buf[fc][3 * i], &s.gyr_buf[fc][3 * i]);
      s.dt_buf[fc - 1].push_back(dt);
      for (int k = 0; k < 3; k++)
        s.acc_buf[fc - 1].push_back(s.acc_buf[fc][3 * i + k]), s.gyr_buf[fc - 1].push_back(s.gyr_buf[fc][3 * i + k]);
    }
    s.Headers[fc - 1] = s.Headers[fc];
    memcpy(&s.Rs[9 * (fc - 1)], &s.Rs[9 * fc], 72);
    for (int k = 0; k < 3; k++) {
      s.Ps[3 * (fc - 1) + k] = s.Ps[3 * fc + k], s.Vs[3 * (fc - 1) + k] = s.Vs[3 * fc + k];
      s.Bas[3 * (fc - 1) + k] = s.Bas[3 * fc + k], s.Bgs[3 * (fc - 1) + k] = s.Bgs[3 * fc + k];
    }
    new_preintegration(e, s, W);
    s.dt_buf[W].clear(), s.acc_buf[W].clear(), s.gyr_buf[W].clear();
    if (s.on_device) s.pre_dirty[fc - 1] = 1;  // the interval that took over the departed frame's samples
    else vio_features_remove_front(s.fm, s.frame_count);
  }
}

void fill_pose_arrays(vio_estimator *e, Sequence &s);

// old2new + the factor list + the loop pose: everything solve_ceres hands to the solver (VINS.cpp:89-129, 505-637)
int build_window(vio_estimator *e, Sequence &s, VioWindow *w) {
  const int W = e->W;
  fill_pose_arrays(e, s);
  {
    const Quat q = RtoQ(e->ric);
    s.ex_pose[0] = e->tic[0], s.ex_pose[1] = e->tic[1], s.ex_pose[2] = e->tic[2];
    s.ex_pose[3] = q.x, s.ex_pose[4] = q.y, s.ex_pose[5] = q.z, s.ex_pose[6] = q.w;
  }
  int nf = 0;
  int rc = vio_features_get_depth_vector(s.fm, s.inv_depth.data(), e->cfg.max_features, &nf);
  if (rc != VIO_OK) return rc;
  // relocalization constraint: front_pose follows retrive_pose_data; it enters when its frame is still in the window
  if (s.front.header != s.retrive.header) s.front = s.retrive;
  s.loop_frame = -1;
  if (!s.front.ids.empty() && s.front.header >= s.Headers[0])
    for (int i = 0; i < W; i++)
      if (s.front.header == s.Headers[i]) s.loop_frame = i;
  int m = 0, nf2 = 0;
  rc = vio_features_export_factors_loop(s.fm, e->cfg.max_factors, s.loop_frame, s.front.ids.data(), s.front.xy.data(),
                                        (int)s.front.ids.size(), s.f_host.data(), s.f_target.data(), s.f_feat.data(),
                                        s.pts_i.data(), s.pts_j.data(), &m, &nf2, &s.n_loop_factors);
  if (rc != VIO_OK) return rc;
  if (nf2 != nf) return VIO_ESTATE;
  if (s.n_loop_factors > 0) s.loop_enable = true;
  if (s.loop_frame >= 0 && s.n_loop_factors == 0) s.loop_frame = -1;  // a loop pose without factors is a free block
  for (int i = 0; i < W; i++) {
    if (!s.pre_valid[i + 1]) return VIO_ESTATE;
    host::preint_export(s.pre[i + 1], &s.preint[i]);
  }
  memset(w, 0, sizeof(*w));
  w->window_size = W, w->n_features = nf, w->n_factors = m, w->marginalization_flag = s.marginalization_flag;
  w->pose = s.pose.data(), w->speed_bias = s.sb.data(), w->ex_pose = s.ex_pose, w->inv_depth = s.inv_depth.data();
  w->factor_host = s.f_host.data(), w->factor_target = s.f_target.data(), w->factor_feature = s.f_feat.data();
  w->factor_pts_i = s.pts_i.data(), w->factor_pts_j = s.pts_j.data();
  w->preint = s.preint.data();
  w->prior = s.has_prior ? &s.prior[s.cur_prior].p : nullptr;
  w->loop_frame = s.loop_frame, w->loop_pose = s.loop_pose;
  if (s.failure_occur) {  // new2old anchors the window where the failed one stood (VINS.cpp:139-144)
    double ypr[3];
    R2ypr(s.last_R_old, ypr);
    w->use_origin_override = 1, w->origin_yaw_deg = ypr[0];
    memcpy(w->origin_p, s.last_P_old, sizeof(w->origin_p));
  }
  w->raw_pose = s.raw_pose.data(), w->raw_speed_bias = s.raw_sb.data(), w->raw_inv_depth = nullptr;
  w->next_prior = &s.prior[1 - s.cur_prior].p;
  w->resident_prior = e->resident_priors ? s.index % e->group_size + 1 : 0;  // slot in the store of the sequence's group
  return VIO_OK;
}

// para_Pose / para_SpeedBias of the window (vector2double, VINS.cpp:89-129) into s.pose / s.sb.
void fill_pose_arrays(vio_estimator *e, Sequence &s) {
  const int P = e->W + 1;
  for (int i = 0; i < P; i++) {
    const Quat q = RtoQ(&s.Rs[9 * i]);
    double *p = &s.pose[7 * i], *b = &s.sb[9 * i];
    p[0] = s.Ps[3 * i], p[1] = s.Ps[3 * i + 1], p[2] = s.Ps[3 * i + 2], p[3] = q.x, p[4] = q.y, p[5] = q.z, p[6] = q.w;
    for (int k = 0; k < 3; k++) b[k] = s.Vs[3 * i + k], b[3 + k] = s.Bas[3 * i + k], b[6 + k] = s.Bgs[3 * i + k];
  }
}

// ---- device-resident sequences (vio_resident.h) --------------------------------------------------------------------------
int group_of(const vio_estimator *e, const Sequence &s) { return s.index / e->group_size; }
int slot_of(const vio_estimator *e, const Sequence &s) { return s.index % e->group_size; }

bool resident_eligible(const vio_estimator *e, const Sequence &s) {
  // (a relocalization frame with more matched ids than a store slot takes stays with the host-side list while the frame
  // is in the window, VINS.cpp:571-596)
  const bool reloc = (int)s.retrive.ids.size() > 256 && s.retrive.header >= s.Headers[0];
  return e->resident && e->resident_priors && s.solver_flag == VIO_SOLVER_NON_LINEAR && s.frame_count == e->W && !reloc && !s.failure_occur;
}

// The group's store exists (reserved at the first promotion). Main thread: HIP calls.
int ensure_store(vio_estimator *e, int g) {
  if (!e->be[g]) return VIO_ESTATE;
  int lcap = 0, ocap = 0;
  if (vio_backend_resident_caps(e->be[g], &lcap, &ocap) == VIO_OK) return VIO_OK;
  // A frame may bring twice the tracker's feature budget; the list holds a window's worth of frames whose features were
  // all new (a sequence that tracks nothing fails the failure detection long before: last_track_num < 4). A list that
  // outgrows its slot ends the frame with VIO_ECAP for that sequence, which restarts.
  e->res_obs_cap = std::min(1024, std::max(256, 2 * e->cfg.max_corners));
  // (the last term: store_pack scans its (W + 2)^2 bucket keys through a landmark-sized array, vio_backend_resident_reserve)
  e->res_list_cap = std::max({1024, (e->W + 2) * e->cfg.max_corners, e->res_obs_cap, (e->W + 2) * (e->W + 2)});
  double ex[7];
  const Quat q = RtoQ(e->ric);
  ex[0] = e->tic[0], ex[1] = e->tic[1], ex[2] = e->tic[2], ex[3] = q.x, ex[4] = q.y, ex[5] = q.z, ex[6] = q.w;
  const int rc = vio_backend_resident_reserve(e->be[g], e->group_size, e->res_list_cap, e->res_obs_cap, ex, e->tic, e->ric);
  if (rc != VIO_OK) e->resident = false;  // (a window size the store does not take, or no memory: every sequence stays on the host-side lists)
  return rc;
}

// The landmark lists of these sequences (ascending, one group) move to their slots of the group's back-end: dumped in
// parallel, uploaded together. A list that does not fit its slot stays on the host.
void promote_group(vio_estimator *e, int g, const std::vector<int> &seqs) {
  if (seqs.empty() || ensure_store(e, g) != VIO_OK) return;
  const int n = (int)seqs.size();
  std::vector<std::vector<VioFeatureInfo>> infos(n);
  std::vector<std::vector<double>> pts(n);
  std::vector<int> ok(n, 0);
  HostPool::get().parallel_for(n, [&](int k) {
    Sequence &s = e->seq[seqs[k]];
    int cnt = 0, np = 0;
    if (vio_features_dump(s.fm, nullptr, 0, &cnt, nullptr, 0, &np) != VIO_OK || cnt > e->res_list_cap) return;
    infos[k].resize(cnt + 1), pts[k].resize(3 * (size_t)np + 3);
    if (vio_features_dump(s.fm, infos[k].data(), cnt, &cnt, pts[k].data(), np, &np) != VIO_OK) return;
    infos[k].resize(cnt);
    ok[k] = 1;
  });
  std::vector<int32_t> slots, counts;
  std::vector<const VioFeatureInfo *> ip;
  std::vector<const double *> pp;
  std::vector<double> lp, lr;
  std::vector<int> who;
  for (int k = 0; k < n; k++) {
    if (!ok[k]) continue;
    Sequence &s = e->seq[seqs[k]];
    slots.push_back(slot_of(e, s)), counts.push_back((int32_t)infos[k].size()), ip.push_back(infos[k].data()), pp.push_back(pts[k].data());
    lp.insert(lp.end(), s.last_P, s.last_P + 3), lr.insert(lr.end(), s.last_R, s.last_R + 9);
    who.push_back(seqs[k]);
  }
  if (who.empty()) return;
  if (vio_backend_resident_load_batch(e->be[g], (int)who.size(), slots.data(), ip.data(), counts.data(), pp.data(), lp.data(), lr.data()) != VIO_OK)
    return;
  for (int q : who) {
    Sequence &s = e->seq[q];
    vio_features_clear(s.fm);
    s.pre_dirty.assign(e->W + 1, 1);
    s.on_device = true;
  }
}

// ... and back: the host-side list takes over again (relocalization, a caller that wants to look at the list).
int demote(vio_estimator *e, Sequence &s) {
  if (!s.on_device) return VIO_OK;
  const int g = group_of(e, s);
  int n = 0, np = 0;
  int rc = vio_backend_resident_fetch(e->be[g], slot_of(e, s), nullptr, 0, &n, nullptr, 0, &np);
  if (rc != VIO_OK) return rc;
  std::vector<VioFeatureInfo> info(n + 1);
  std::vector<double> pts(3 * (size_t)np + 3);
  rc = vio_backend_resident_fetch(e->be[g], slot_of(e, s), info.data(), n + 1, &n, pts.data(), np + 1, &np);
  if (rc != VIO_OK) return rc;
  rc = vio_features_load(s.fm, info.data(), n, pts.data());
  if (rc != VIO_OK) return rc;
  s.on_device = false;
  // the intervals the device integrated: integrate them here again from the buffered samples (the same operations in the
  // same order as the incremental propagation: the same bits)
  for (int i = 1; i <= e->W; i++)
    if (s.pre_stale[i]) {
      const double ba[3] = {s.pre[i].ba[0], s.pre[i].ba[1], s.pre[i].ba[2]}, bg[3] = {s.pre[i].bg[0], s.pre[i].bg[1], s.pre[i].bg[2]};
      repropagate(e, s, i, ba, bg);
      s.pre_stale[i] = 0, s.pre_merge[i] = 0;
    }
  return VIO_OK;
}

void take_solution(vio_estimator *e, Sequence &s, const VioWindow &w, const VioSolveStats &st);

// double2vector of a window solved on the resident path: the results take the places unpack_window gives them on the host
// path and take_solution runs as it does there (the relocalization bookkeeping included; the landmarks took their depths
// in store_finish).
void take_resident_solution(vio_estimator *e, Sequence &s, const VioResidentResult &r, VioPrior &next) {
  const int P = e->W + 1;
  memcpy(s.pose.data(), r.pose, sizeof(double) * 7 * P), memcpy(s.sb.data(), r.speed_bias, sizeof(double) * 9 * P);
  if (r.raw_pose) memcpy(s.raw_pose.data(), r.raw_pose, sizeof(double) * 7 * P);
  if (r.loop_pose && r.n_loop_factors > 0) memcpy(s.loop_pose, r.loop_pose, sizeof(s.loop_pose));
  VioWindow w;
  memset(&w, 0, sizeof(w));
  w.window_size = e->W, w.n_features = r.n_features, w.next_prior = &next;
  take_solution(e, s, w, r.stats);
}

double normalize_angle(double a) {  // Utility::normalizeAngle, degrees (utility.hpp:171-179)
  const double two_pi = 2.0 * 180;
  if (a > 0) return a - two_pi * floor((a + 180.0) / two_pi);
  return a + two_pi * floor((-a + 180.0) / two_pi);
}

// double2vector of the solved window (the device already applied new2old to the para arrays), the loop bookkeeping
// (VINS.cpp:664-680, 168-189) and the prior hand-over.
void take_solution(vio_estimator *e, Sequence &s, const VioWindow &w, const VioSolveStats &st) {
  const int W = e->W, P = W + 1;
  s.final_cost = st.final_cost;
  if (s.loop_frame >= 0) {
    const int i = s.loop_frame;
    const double *rp = &s.raw_pose[7 * i];
    double Rs_i[9], Rs_loop[9], RlT[9], d[3], ypr_i[3], ypr_l[3];
    qtoR(qnormalized(qfrom_pose(rp)), Rs_i);
    qtoR(qnormalized(qfrom_pose(s.loop_pose)), Rs_loop);
    mat3T(Rs_loop, RlT);
    for (int k = 0; k < 3; k++) d[k] = rp[k] - s.loop_pose[k];
    mat3vec(RlT, d, s.front.relative_t);
    double Rrel[9];
    mat3mul(RlT, Rs_i, Rrel);
    s.front.relative_q = RtoQ(Rrel);
    R2ypr(Rs_i, ypr_i), R2ypr(Rs_loop, ypr_l);
    s.front.relative_yaw = normalize_angle(ypr_i[0] - ypr_l[0]);
    memcpy(s.front.loop_pose, s.loop_pose, sizeof(s.loop_pose));
  }
  // the gauge the device applied: rot_diff and origin_P0 of new2old, needed once more for the loop pose
  double origin_ypr[3], origin_P0[3], raw_ypr[3], R00[9], rot_diff[9];
  if (s.failure_occur) {
    R2ypr(s.last_R_old, origin_ypr), memcpy(origin_P0, s.last_P_old, 24);
  } else {
    R2ypr(&s.Rs[0], origin_ypr), memcpy(origin_P0, &s.Ps[0], 24);
  }
  qtoR(qfrom_pose(&s.raw_pose[0]), R00);
  R2ypr(R00, raw_ypr);
  const double yd[3] = {origin_ypr[0] - raw_ypr[0], 0, 0};
  ypr2R(yd, rot_diff);
  for (int i = 0; i < P; i++) {
    const double *p = &s.pose[7 * i], *b = &s.sb[9 * i];
    qtoR(qfrom_pose(p), &s.Rs[9 * i]);
    for (int k = 0; k < 3; k++)
      s.Ps[3 * i + k] = p[k], s.Vs[3 * i + k] = b[k], s.Bas[3 * i + k] = b[3 + k], s.Bgs[3 * i + k] = b[6 + k];
  }
  if (s.loop_enable) {  // drift of the window against the old keyframe (VINS.cpp:168-189)
    s.loop_enable = false;
    if (s.loop_frame >= 0) {
      double Rl[9], Rloop[9], d[3], Pl[3], ypr_old[3], ypr_loop[3], Rold[9];
      qtoR(qnormalized(qfrom_pose(s.loop_pose)), Rl);
      mat3mul(rot_diff, Rl, Rloop);
      for (int k = 0; k < 3; k++) d[k] = s.loop_pose[k] - s.raw_pose[k];
      mat3vec(rot_diff, d, Pl);
      for (int k = 0; k < 3; k++) Pl[k] += origin_P0[k];
      qtoR(s.front.Q_old, Rold);
      R2ypr(Rold, ypr_old), R2ypr(Rloop, ypr_loop);
      const double dy[3] = {ypr_old[0] - ypr_loop[0], 0, 0};
      ypr2R(dy, s.r_drift);
      double t[3];
      mat3vec(s.r_drift, Pl, t);
      for (int k = 0; k < 3; k++) s.t_drift[k] = s.front.P_old[k] - t[k];
    }
  }
  if (!s.on_device) vio_features_set_depth(s.fm, s.inv_depth.data(), w.n_features);
  // n == -1: MARGIN_SECOND_NEW without the second-newest pose in the old prior leaves it as it was (VINS.cpp:778-779)
  if (w.next_prior && w.next_prior->n > 0) s.cur_prior = 1 - s.cur_prior, s.has_prior = true;
}

// Eigen::Quaterniond::FromTwoVectors(a, b).toRotationMatrix() for unit a, b (Geometry/Quaternion.h)
void rotation_between(const double a[3], const double b[3], double R[9]) {
  const double c = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  if (c < -1.0 + 1e-12) {  // opposite vectors: any axis orthogonal to a, angle pi
    double ax[3] = {1, 0, 0};
    if (fabs(a[0]) > 0.9) ax[0] = 0, ax[1] = 1;
    double v[3] = {a[1] * ax[2] - a[2] * ax[1], a[2] * ax[0] - a[0] * ax[2], a[0] * ax[1] - a[1] * ax[0]};
    const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    qtoR(Quat{v[0] / n, v[1] / n, v[2] / n, 0.0}, R);
    return;
  }
  const double axis[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  const double sq = sqrt((1.0 + c) * 2.0), inv = 1.0 / sq;
  qtoR(Quat{axis[0] * inv, axis[1] * inv, axis[2] * inv, sq * 0.5}, R);
}

void g2R(const double g[3], double R0[9]) {  // Utility::g2R (utility.cpp:11-21)
  const double n = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
  const double ng1[3] = {g[0] / n, g[1] / n, g[2] / n}, ng2[3] = {0, 0, 1};
  double R[9], ypr[3], Y[9], T[9];
  rotation_between(ng1, ng2, R);
  R2ypr(R, ypr);
  const double y1[3] = {-ypr[0], 0, 0};
  ypr2R(y1, Y);
  mat3mul(Y, R, T);
  const double y2[3] = {-90, 0, 0};
  ypr2R(y2, Y);
  mat3mul(Y, T, R0);
}

// VINS::solveInitial + relativePose + visualInitialAlign (VINS.cpp:833-1145).
bool solve_initial(vio_estimator *e, Sequence &s) {
  const int W = e->W, P = W + 1;
  // the landmark store as SfM features (VINS.cpp:883-899) and as correspondences (getCorresponding)
  int nfe = 0, npts = 0;
  vio_features_dump(s.fm, nullptr, 0, &nfe, nullptr, 0, &npts);
  std::vector<VioFeatureInfo> info(nfe > 0 ? nfe : 1);
  std::vector<double> pts(3 * (size_t)(npts > 0 ? npts : 1));
  if (vio_features_dump(s.fm, info.data(), nfe, &nfe, pts.data(), npts, &npts) != VIO_OK) return false;
  std::vector<init::SfmFeature> sfm_f(nfe);
  std::vector<size_t> first(nfe);
  {
    size_t off = 0;
    for (int j = 0; j < nfe; j++) {
      first[j] = off;
      sfm_f[j].id = info[j].id;
      if (info[j].start_frame < 0 || info[j].start_frame + info[j].n_obs > P) return false;  // not a window's store
      for (int k = 0; k < info[j].n_obs; k++)
        sfm_f[j].observation.push_back({info[j].start_frame + k, {pts[3 * (off + k)], pts[3 * (off + k) + 1]}});
      off += info[j].n_obs;
    }
  }
  // relativePose (VINS.cpp:1106-1145): the oldest frame with enough parallax against the newest one
  int l = -1;
  double relative_R[9], relative_T[3];
  for (int i = 0; i < W; i++) {
    std::vector<double> a, b;
    for (int j = 0; j < nfe; j++)
      if (info[j].start_frame <= i && info[j].start_frame + info[j].n_obs - 1 >= W) {
        const double *pa = &pts[3 * (first[j] + (i - info[j].start_frame))], *pb = &pts[3 * (first[j] + (W - info[j].start_frame))];
        a.push_back(pa[0]), a.push_back(pa[1]), b.push_back(pb[0]), b.push_back(pb[1]);
      }
    const int nc = (int)a.size() / 2;
    if (nc > 20) {
      double sum = 0;
      for (int k = 0; k < nc; k++) sum += sqrt((a[2 * k] - b[2 * k]) * (a[2 * k] - b[2 * k]) + (a[2 * k + 1] - b[2 * k + 1]) * (a[2 * k + 1] - b[2 * k + 1]));
      if (sum / nc * 520 < 30) return false;  // FAIL_PARALLAX
      // the gyroscope's rotation between frame i and the newest one, as a tie-breaker for planar scenes only: the product
      // of the pre-integrated delta_q of the intervals in between (body frame), moved into the camera frame
      Quat dq{0, 0, 0, 1};
      for (int k = i + 1; k <= W; k++) dq = qmul(dq, s.pre[k].dq);
      double Rb[9], Rt[9], hint[9], ricT[9];
      qtoR(qnormalized(dq), Rb);
      mat3T(e->ric, ricT);
      mat3mul(ricT, Rb, Rt), mat3mul(Rt, e->ric, hint);
      const bool got = e->init_relpose_fit ? init::solve_relative_rt(a, b, relative_R, relative_T, nullptr, hint)
                                           : init::solve_relative_rt_five_point(a, b, relative_R, relative_T, nullptr);
      if (!got) return false;  // FAIL_RELATIVE
      l = i;
      break;
    }
  }
  if (l < 0) return false;
  std::vector<double> Q(4 * (size_t)P), T(3 * (size_t)P);
  std::map<int, std::vector<double>> tracked;
  if (!init::sfm_construct(P, Q.data(), T.data(), l, relative_R, relative_T, sfm_f, tracked)) {
    s.marginalization_flag = VIO_MARGIN_OLD;  // FAIL_SFM (VINS.cpp:915-917)
    return false;
  }
  // PnP for every frame of all_image_frame (VINS.cpp:926-1003)
  double ricT[9];
  mat3T(e->ric, ricT);
  int i = 0;
  for (auto &kv : s.all_image_frame) {
    init::Frame &f = kv.second;
    double Rq[9];
    if (i < P && f.header == s.Headers[i]) {
      qtoR(Quat{Q[4 * i], Q[4 * i + 1], Q[4 * i + 2], Q[4 * i + 3]}, Rq);
      f.is_key_frame = true;
      mat3mul(Rq, ricT, f.R);
      memcpy(f.T, &T[3 * i], 24);
      i++;
      continue;
    }
    if (i < P && f.header > s.Headers[i]) i++;
    const int ii = std::min(i, P - 1);
    qtoR(Quat{Q[4 * ii], Q[4 * ii + 1], Q[4 * ii + 2], Q[4 * ii + 3]}, Rq);
    double R_init[9], P_init[3], v[3];
    mat3T(Rq, R_init);  // Q[i].inverse()
    mat3vec(R_init, &T[3 * ii], v);
    for (int k = 0; k < 3; k++) P_init[k] = -v[k];
    f.is_key_frame = false;
    std::vector<double> p3, p2;
    for (const VioObs &o : f.points) {
      auto it = tracked.find(o.id);
      if (it == tracked.end()) continue;
      p3.insert(p3.end(), it->second.begin(), it->second.end());
      p2.push_back(o.x), p2.push_back(o.y);
    }
    if (p2.size() / 2 < 6) return false;                       // "init Not enough points for solve pnp !"
    if (!init::pnp_refine(p3, p2, R_init, P_init)) return false;  // FAIL_PNP
    double R_pnp[9];
    mat3T(R_init, R_pnp);
    mat3mul(R_pnp, ricT, f.R);
    mat3vec(R_pnp, P_init, v);
    for (int k = 0; k < 3; k++) f.T[k] = -v[k];
  }
  // visualInitialAlign (VINS.cpp:1021-1104)
  std::vector<init::Frame> frames;
  frames.reserve(s.all_image_frame.size());
  for (auto &kv : s.all_image_frame) frames.push_back(kv.second);
  std::vector<double> x;
  if (!init::visual_imu_alignment(e->cfg, e->tic, frames, W, s.Bgs.data(), s.g, x)) return false;  // FAIL_ALIGN
  {
    size_t k = 0;
    for (auto &kv : s.all_image_frame) kv.second = frames[k++];  // the re-integrated intervals
  }
  for (int k = 0; k <= s.frame_count; k++) {
    init::Frame &f = s.all_image_frame[s.Headers[k]];
    memcpy(&s.Ps[3 * k], f.T, 24), memcpy(&s.Rs[9 * k], f.R, 72);
    f.is_key_frame = true;
  }
  int nfeat = 0;
  vio_features_count(s.fm, &nfeat);
  std::vector<double> minus1(nfeat > 0 ? nfeat : 1, -1.0);
  vio_features_clear_depth(s.fm, minus1.data(), nfeat);
  const double tic0[3] = {0, 0, 0};  // "triangulat on cam pose , no tic"
  vio_features_triangulate(s.fm, s.Ps.data(), s.Rs.data(), tic0, e->ric);
  const double sc = x.back();
  const double zero[3] = {0, 0, 0};
  for (int k = 0; k <= W; k++) repropagate(e, s, k, zero, &s.Bgs[3 * k]);
  {
    double r0t[3], base[3];
    mat3vec(&s.Rs[0], e->tic, r0t);
    for (int k = 0; k < 3; k++) base[k] = sc * s.Ps[k] - r0t[k];
    for (int k = s.frame_count; k >= 0; k--) {
      double rt[3];
      mat3vec(&s.Rs[9 * k], e->tic, rt);
      for (int c = 0; c < 3; c++) s.Ps[3 * k + c] = sc * s.Ps[3 * k + c] - rt[c] - base[c];
    }
  }
  {
    int kv = -1;
    for (auto &f : s.all_image_frame)
      if (f.second.is_key_frame) {
        kv++;
        if (kv <= W && 3 * (size_t)kv + 2 < x.size()) mat3vec(f.second.R, &x[3 * kv], &s.Vs[3 * kv]);  // x.segment<3>(kv * 3), as written
      }
  }
  vio_features_scale_depth(s.fm, sc);
  double R0[9], ypr[3], Yr[9], Rd[9], gn[3];
  g2R(s.g, R0);
  R2ypr(R0, ypr);
  const double yneg[3] = {-ypr[0], 0, 0};
  ypr2R(yneg, Yr);
  mat3mul(Yr, R0, Rd);
  mat3vec(Rd, s.g, gn);
  memcpy(s.g, gn, sizeof(gn));
  for (int k = 0; k <= s.frame_count; k++) {
    double v[3], M[9];
    mat3vec(Rd, &s.Ps[3 * k], v), memcpy(&s.Ps[3 * k], v, 24);
    mat3vec(Rd, &s.Vs[3 * k], v), memcpy(&s.Vs[3 * k], v, 24);
    mat3mul(Rd, &s.Rs[9 * k], M), memcpy(&s.Rs[9 * k], M, 72);
  }
  return true;
}

void remember_last(vio_estimator *e, Sequence &s) {  // VINS.cpp:431-434, 472-475
  const int W = e->W;
  memcpy(s.last_R, &s.Rs[9 * W], 72), memcpy(s.last_P, &s.Ps[3 * W], 24);
  memcpy(s.last_R_old, &s.Rs[0], 72), memcpy(s.last_P_old, &s.Ps[0], 24);
}

// The sequences are solved in n_groups contiguous groups, one back-end context each. Decided before the first solve.
void regroup(vio_estimator *e) {
  for (int g = 0; g < vio_estimator::kMaxGroups; g++)
    if (e->be[g]) return;  // (contexts exist: slots and prior stores are bound to the grouping)
  const int n_seq = e->n_seq;
  {  // VIO_AMD_EST_GROUPS overrides the number of solve groups (1 = one launch for all sequences)
    // Two groups by default: more groups only pay while every group's stream has a hardware queue of its own (HIP maps
    // streams onto 4 queues by default; in a process that holds other streams — a torch process, say — four groups
    // alias, two of the kernels serialize and the frame gets slower than with one group: 36 k instead of 48 k solves/s).
    // groups of about 256 sequences (one window-kernel launch that fills every CU's two workgroup slots half), at least two
    int ng = n_seq >= 64 ? std::max(2, (n_seq + 128) / 256) : 1;
    // Resident sequences leave the host little to overlap with the kernel, and one launch of all windows fills the CUs'
    // second workgroup slots by itself (measured, 512 sequences: 4.9 ms per frame with one group, 6.7 ms with two)
    if (e->resident && e->resident_priors) ng = 1;
    if (const char *env = getenv("VIO_AMD_EST_GROUPS")) {
      char *end = nullptr;
      const long val = strtol(env, &end, 10);
      if (end == env || *end != '\0' || val < 1) fprintf(stderr, "vio_amd: VIO_AMD_EST_GROUPS=\"%s\" is not a positive number, ignored\n", env);
      else ng = (int)std::min<long>(val, vio_estimator::kMaxGroups);
    }
    ng = std::max(1, std::min(std::min(ng, (int)vio_estimator::kMaxGroups), n_seq));
    e->group_size = (n_seq + ng - 1) / ng;
    e->n_groups = (n_seq + e->group_size - 1) / e->group_size;
  }
}

// One published frame of every resident sequence: what processImage does with it, with the landmark work on the device.
//   main thread   begin (per group)
//   pool          per sequence: para_Pose / para_SpeedBias, the pre-integration blocks that changed, the observations -> staging
//   main thread   ingest (H2D + store_ingest, per group, no wait), then launch per group (waits for the group's counts:
//                 layout, store_pack, window kernel, store_finish, results queued), then collect per group
//   pool          per sequence: double2vector, prior header, failure / slide of the host-side states
int resident_frame(vio_estimator *e, const VioObs *obs, const int32_t *n_obs, int32_t obs_stride, const double *headers,
                   const uint8_t *active, VioFrameResult *results) {
  const int W = e->W;
  std::vector<char> in_group(e->n_groups, 0);
  std::vector<int> todo;
  for (int q = 0; q < e->n_seq; q++) {
    Sequence &s = e->seq[q];
    if (!s.on_device || (active && !active[q])) continue;
    in_group[group_of(e, s)] = 1;
    todo.push_back(q);
  }
  if (todo.empty()) return VIO_OK;
  int rc = VIO_OK;
  for (int g = 0; g < e->n_groups && rc == VIO_OK; g++)
    if (in_group[g]) rc = vio_backend_resident_begin(e->be[g]);
  e->res_rc.assign(e->n_seq, VIO_OK);
  if (rc == VIO_OK) {
    HostPool::get().parallel_for((int)todo.size(), [&](int i) {
      const int q = todo[i];
      Sequence &s = e->seq[q];
      VioFrameResult &res = results[q];
      vio_backend_t *be = e->be[group_of(e, s)];
      const int slot = slot_of(e, s);
      int r = VIO_OK;
      if (n_obs[q] < 0 || (obs_stride > 0 && n_obs[q] > obs_stride)) r = VIO_EINVAL;
      s.Headers[s.frame_count] = headers[q];
      fill_pose_arrays(e, s);
      VioPreintegration blk;
      for (int k = 1; k <= W && r == VIO_OK; k++) {
        if (!s.pre_dirty[k] && k != W) continue;  // (the newest interval is new with every frame)
        if (!s.pre_valid[k]) {
          r = VIO_ESTATE;
          break;
        }
        if (s.pre_stale[k]) {
          // the device integrates: the whole of the newest interval, or the samples an older one absorbed at a non-keyframe slide
          const int ns = (int)s.dt_buf[k].size();
          const bool fresh = k == W || s.pre_merge[k] <= 0 || s.pre_merge[k] >= ns;
          const int first = fresh ? 0 : ns - s.pre_merge[k];
          int ri = VIO_ECAP;
          if (k == W || s.pre_merge[k] > 0)
            ri = vio_backend_resident_stage_imu(be, slot, k - 1, fresh ? 1 : 0, &s.lin_acc[3 * k], &s.lin_gyr[3 * k], s.pre[k].ba, s.pre[k].bg,
                                                ns - first, s.dt_buf[k].data() + first, s.acc_buf[k].data() + 3 * first,
                                                s.gyr_buf[k].data() + 3 * first);
          if (ri == VIO_ECAP) {  // (no room for the samples, or a stale interval whose new samples are not known: the block instead)
            host::Preint tmp = s.pre[k];
            host::preint_init(tmp, &e->cfg, &s.lin_acc[3 * k], &s.lin_gyr[3 * k], s.pre[k].ba, s.pre[k].bg);
            for (int i = 0; i < ns; i++) host::propagate(tmp, s.dt_buf[k][i], &s.acc_buf[k][3 * i], &s.gyr_buf[k][3 * i]);
            host::preint_export(tmp, &blk);
            ri = vio_backend_resident_stage_preint(be, slot, k - 1, &blk, tmp.acc_0, tmp.gyr_0);
          }
          r = ri;
          s.pre_merge[k] = 0;
        } else {
          host::preint_export(s.pre[k], &blk);
          r = vio_backend_resident_stage_preint(be, slot, k - 1, &blk, s.pre[k].acc_0, s.pre[k].gyr_0);
        }
        s.pre_dirty[k] = 0;
      }
      // relocalization constraint (build_window): front_pose follows retrive_pose_data; it enters while its frame is in the window
      if (s.front.header != s.retrive.header) s.front = s.retrive;
      s.loop_frame = -1;
      if (!s.front.ids.empty() && s.front.header >= s.Headers[0])
        for (int i = 0; i < W; i++)
          if (s.front.header == s.Headers[i]) s.loop_frame = i;
      if (r == VIO_OK)
        r = vio_backend_resident_stage(be, slot, obs + (size_t)q * obs_stride, n_obs[q], s.Ps.data(), s.Rs.data(), s.pose.data(), s.sb.data(),
                                       s.has_prior ? &s.prior[s.cur_prior].p : nullptr, s.loop_frame, s.front.ids.data(), s.front.xy.data(),
                                       (int)s.front.ids.size());
      e->res_rc[q] = r;
      (void)res;
    });
    for (int g = 0; g < e->n_groups && rc == VIO_OK; g++)
      if (in_group[g]) rc = vio_backend_resident_ingest(e->be[g]);
    for (int g = 0; g < e->n_groups && rc == VIO_OK; g++)
      if (in_group[g]) rc = vio_backend_resident_launch(e->be[g]);
  }
  for (int g = 0; g < e->n_groups; g++)  // (every group that began is brought back to idle, also after an error in another)
    if (in_group[g]) {
      const int rcc = vio_backend_resident_collect(e->be[g]);
      if (rc == VIO_OK && rcc != VIO_ESTATE) rc = rcc;
    }
  if (rc != VIO_OK) {  // device error: the stores may or may not have advanced; every sequence of this path restarts
    for (int q : todo) {
      clear_state(e, e->seq[q]);
      results[q].action = VIO_FRAME_ERROR, results[q].error = rc;
    }
    return rc;
  }
  int first_error = VIO_OK;
  HostPool::get().parallel_for((int)todo.size(), [&](int i) {
    const int q = todo[i];
    Sequence &s = e->seq[q];
    VioFrameResult &res = results[q];
    VioResidentResult r;
    VioPrior &next = s.prior[1 - s.cur_prior].p;
    int rr = e->res_rc[q];
    if (rr == VIO_OK) rr = vio_backend_resident_result(e->be[group_of(e, s)], slot_of(e, s), &r, &next);
    if (rr == VIO_OK) rr = r.status;
    if (rr != VIO_OK) {  // (not staged, or the store refused the frame: the slot's list is undefined now)
      clear_state(e, s);
      res.action = VIO_FRAME_ERROR, res.error = rr;
      return;
    }
    s.marginalization_flag = r.marginalization_flag, s.last_track_num = r.track_num;
    res.marginalization_flag = r.marginalization_flag, res.track_num = r.track_num;
    res.n_features = r.n_features, res.n_factors = r.n_factors, res.n_loop_factors = r.n_loop_factors;
    res.stats = r.stats;
    s.n_loop_factors = r.n_loop_factors;
    if (s.n_loop_factors > 0) s.loop_enable = true;
    if (s.loop_frame >= 0 && s.n_loop_factors == 0) s.loop_frame = -1;  // a loop pose without factors is a free block (build_window)
    take_resident_solution(e, s, r, next);
    s.failure_occur = 0;
    if (r.failure_reasons) {
      s.failure_occur = 1;
      clear_state(e, s);
      res.action = VIO_FRAME_FAILURE, res.failure_reasons = r.failure_reasons;
      return;
    }
    slide_window(e, s);
    remember_last(e, s);
    res.action = VIO_FRAME_SOLVED;
  });
  for (int q : todo)
    if (results[q].action == VIO_FRAME_ERROR && first_error == VIO_OK) first_error = results[q].error;
  return first_error;
}

}  // namespace

extern "C" {

int vio_estimator_create(const VioConfig *cfg, int32_t n_seq, const double tic[3], const double ric[9],
                         vio_estimator_t **out) {
  if (!cfg || !out || n_seq < 1 || !tic || !ric || cfg->window_size < 3 || cfg->max_features < 1 || cfg->max_factors < 1)
    return VIO_EINVAL;
  vio_estimator *e = new (std::nothrow) vio_estimator();
  if (!e) return VIO_ENOMEM;
  e->cfg = *cfg, e->W = cfg->window_size, e->n_seq = n_seq;
  regroup(e);
  memcpy(e->tic, tic, 24), memcpy(e->ric, ric, 72);
  const int W = e->W, P = W + 1, cap = vio_prior_capacity(W);
  e->seq.resize(n_seq);
  for (Sequence &s : e->seq) {
    s.Ps.assign(3 * P, 0), s.Rs.assign(9 * P, 0), s.Vs.assign(3 * P, 0), s.Bas.assign(3 * P, 0), s.Bgs.assign(3 * P, 0);
    s.Headers.assign(P, 0);
    s.pre.resize(P), s.pre_valid.assign(P, 0);
    s.dt_buf.resize(P), s.acc_buf.resize(P), s.gyr_buf.resize(P);
    s.lin_acc.assign(3 * P, 0), s.lin_gyr.assign(3 * P, 0);
    if (vio_features_create(W, &s.fm) != VIO_OK) {
      vio_estimator_destroy(e);
      return VIO_ENOMEM;
    }
    s.index = (int)(&s - e->seq.data());
    s.prior[0].init(cap, e->resident_priors), s.prior[1].init(cap, e->resident_priors);
    memcpy(s.last_R, kI3, 72), memcpy(s.last_R_old, kI3, 72), memcpy(s.r_drift, kI3, 72);
    for (int k = 0; k < 3; k++) s.last_P[k] = s.last_P_old[k] = s.t_drift[k] = 0;
    s.pose.assign(7 * P, 0), s.sb.assign(9 * P, 0), s.raw_pose.assign(7 * P, 0), s.raw_sb.assign(9 * P, 0);
    s.inv_depth.assign(cfg->max_features, 0);
    s.pts_i.assign((size_t)3 * cfg->max_factors, 0), s.pts_j.assign((size_t)3 * cfg->max_factors, 0);
    s.f_host.assign(cfg->max_factors, 0), s.f_target.assign(cfg->max_factors, 0), s.f_feat.assign(cfg->max_factors, 0);
    s.preint.resize(W);
    s.pre_dirty.assign(P, 0), s.pre_stale.assign(P, 0), s.pre_merge.assign(P, 0);
    clear_state(e, s);
  }
  e->windows.resize(n_seq), e->stats.resize(n_seq);
  *out = e;
  return VIO_OK;
}

void vio_estimator_destroy(vio_estimator_t *e) {
  if (!e) return;
  for (Sequence &s : e->seq)
    if (s.fm) vio_features_destroy(s.fm);
  for (int g = 0; g < vio_estimator::kMaxGroups; g++)
    if (e->be[g]) vio_backend_destroy(e->be[g]);
  delete e;
}

int vio_estimator_enable_initialization(vio_estimator_t *e, int32_t enable) {
  if (!e) return VIO_EINVAL;
  e->enable_init = enable != 0;
  e->init_relpose_fit = enable == 2;
  return VIO_OK;
}

int vio_estimator_set_resident(vio_estimator_t *e, int32_t enable) {
  if (!e) return VIO_EINVAL;
  if (enable && !e->resident_priors) return VIO_ESTATE;  // (VIO_AMD_HOST_PRIORS=1: the priors travel through the host)
  e->resident = enable != 0;
  regroup(e);
  if (!e->resident)
    for (Sequence &s : e->seq) {
      const int rc = demote(e, s);
      if (rc != VIO_OK) return rc;
    }
  return VIO_OK;
}

int vio_estimator_clear(vio_estimator_t *e, int32_t seq) {
  if (!e || seq < 0 || seq >= e->n_seq) return VIO_EINVAL;
  clear_state(e, e->seq[seq]);
  return VIO_OK;
}

// VINS::processIMU (VINS.cpp:333-375): extend the pre-integration of the interval that ends in the frame being filled
// and propagate that frame's state with the midpoint rule.
int vio_estimator_process_imu(vio_estimator_t *e, int32_t seq, double dt, const double acc[3], const double gyr[3]) {
  if (!e || seq < 0 || seq >= e->n_seq || !acc || !gyr) return VIO_EINVAL;
  Sequence &s = e->seq[seq];
  if (!s.first_imu) {
    s.first_imu = true;
    memcpy(s.acc_0, acc, 24), memcpy(s.gyr_0, gyr, 24);
  }
  const int j = s.frame_count;
  if (!s.pre_valid[j]) new_preintegration(e, s, j);
  if (j != 0) {
    if (s.on_device && e->resident_imu) s.pre_stale[j] = 1;  // (the device integrates this interval from the buffered samples)
    else host::propagate(s.pre[j], dt, acc, gyr);
    if (s.solver_flag != VIO_SOLVER_NON_LINEAR && s.tmp_valid) {  // tmp_pre_integration->push_back (VINS.cpp:350-351)
      host::propagate(s.tmp.pre, dt, acc, gyr);
      s.tmp.dt.push_back(dt);
      for (int k = 0; k < 3; k++) s.tmp.acc.push_back(acc[k]), s.tmp.gyr.push_back(gyr[k]);
    }
    s.dt_buf[j].push_back(dt);
    for (int k = 0; k < 3; k++) s.acc_buf[j].push_back(acc[k]), s.gyr_buf[j].push_back(gyr[k]);
    const double g[3] = {0, 0, e->cfg.gravity};
    double *R = &s.Rs[9 * j], *Pj = &s.Ps[3 * j], *V = &s.Vs[3 * j];
    const double *ba = &s.Bas[3 * j], *bg = &s.Bgs[3 * j];
    double a0[3], ua0[3], w[3], a1[3], ua1[3];
    for (int k = 0; k < 3; k++) a0[k] = s.acc_0[k] - ba[k], w[k] = 0.5 * (s.gyr_0[k] + gyr[k]) - bg[k];
    mat3vec(R, a0, ua0);
    // Rs[j] *= deltaQ(un_gyr * dt).toRotationMatrix()  (utility.hpp:22-34: a small-angle quaternion, NOT normalized)
    double dR[9], Rn[9];
    qtoR(Quat{w[0] * dt / 2, w[1] * dt / 2, w[2] * dt / 2, 1.0}, dR);
    mat3mul(R, dR, Rn);
    memcpy(R, Rn, 72);
    for (int k = 0; k < 3; k++) a1[k] = acc[k] - ba[k];
    mat3vec(R, a1, ua1);
    for (int k = 0; k < 3; k++) {
      const double ua = 0.5 * ((ua0[k] - g[k]) + (ua1[k] - g[k]));
      Pj[k] += dt * V[k] + 0.5 * dt * dt * ua;
      V[k] += dt * ua;
    }
  }
  memcpy(s.acc_0, acc, 24), memcpy(s.gyr_0, gyr, 24);
  return VIO_OK;
}

// processIMU for every sequence at once: sequence q receives n_samples[q] samples dt/acc/gyr[q*stride ...], in order.
// Sequences are independent, so they are spread over the host pool (one frame interval of 256 sequences is a few
// thousand 15x15 covariance propagations).
int vio_estimator_process_imu_batch(vio_estimator_t *e, const int32_t *n_samples, int32_t stride, const double *dt,
                                    const double *acc, const double *gyr) {
  if (!e || !n_samples || stride < 0 || !dt || !acc || !gyr) return VIO_EINVAL;
  for (int q = 0; q < e->n_seq; q++)
    if (n_samples[q] < 0 || n_samples[q] > stride) return VIO_EINVAL;
  HostPool::get().parallel_for(e->n_seq, [&](int q) {
    for (int i = 0; i < n_samples[q]; i++) {
      const size_t k = (size_t)q * stride + i;
      vio_estimator_process_imu(e, q, dt[k], acc + 3 * k, gyr + 3 * k);
    }
  });
  return VIO_OK;
}

int vio_estimator_set_initial_state(vio_estimator_t *e, int32_t seq, const double *headers, const double *Ps,
                                    const double *Rs, const double *Vs, const double *Bas, const double *Bgs) {
  if (!e || seq < 0 || seq >= e->n_seq || !headers || !Ps || !Rs || !Vs || !Bas || !Bgs) return VIO_EINVAL;
  Sequence &s = e->seq[seq];
  if (s.solver_flag != VIO_SOLVER_INITIAL) return VIO_ESTATE;
  const int P = e->W + 1;
  s.init_headers.assign(headers, headers + P);
  s.init_Ps.assign(Ps, Ps + 3 * P), s.init_Rs.assign(Rs, Rs + 9 * P), s.init_Vs.assign(Vs, Vs + 3 * P);
  s.init_Bas.assign(Bas, Bas + 3 * P), s.init_Bgs.assign(Bgs, Bgs + 3 * P);
  s.init_pending = true;
  return VIO_OK;
}

int vio_estimator_set_relocalization(vio_estimator_t *e, int32_t seq, double header, const double P_old[3],
                                     const double Q_old[4], const int32_t *ids, const double *xy, int32_t n) {
  if (!e || seq < 0 || seq >= e->n_seq || n < 0 || (n > 0 && (!ids || !xy || !P_old || !Q_old))) return VIO_EINVAL;
  Relocalization &r = e->seq[seq].retrive;
  r = Relocalization();
  r.header = header;
  if (n > 0) {
    memcpy(r.P_old, P_old, 24);
    r.Q_old = Quat{Q_old[0], Q_old[1], Q_old[2], Q_old[3]};
    r.ids.assign(ids, ids + n), r.xy.assign(xy, xy + 2 * n);
  }
  return VIO_OK;
}

// VINS::processImage for every active sequence (VINS.cpp:377-478); the solves of all of them are ONE device launch.
int vio_estimator_process_images(vio_estimator_t *e, const VioObs *obs, const int32_t *n_obs, int32_t obs_stride,
                                 const double *headers, const uint8_t *active, VioFrameResult *results) {
  if (!e || !obs || !n_obs || !headers || !results || obs_stride < 0) return VIO_EINVAL;
  const int W = e->W, P = W + 1;
  e->solving.clear();
  int first_error = VIO_OK;
  const auto t_begin = std::chrono::steady_clock::now();
  // resident sequences that cannot stay on the device for this frame go back to the host-side list first
  bool any_resident = false;
  for (int q = 0; q < e->n_seq; q++) {
    Sequence &s = e->seq[q];
    if (!s.on_device || (active && !active[q])) continue;
    if (s.front.header != s.retrive.header) s.front = s.retrive;
    if (!resident_eligible(e, s) || n_obs[q] > e->res_obs_cap) {
      const int rcd = demote(e, s);
      if (rcd != VIO_OK) {
        clear_state(e, s);
        if (first_error == VIO_OK) first_error = rcd;
      }
    }
    any_resident = any_resident || s.on_device;
  }
  // phase A, per sequence and independent of the others (spread over the host pool): landmark bookkeeping, the
  // INITIAL / NON_LINEAR branch, triangulation, the window as solve_ceres hands it to the solver
  e->staged.resize(e->n_seq);
  e->wants_solve.assign(e->n_seq, 0);
  auto phase_a = [&](int q) {
    VioFrameResult &res = results[q];
    memset(&res, 0, sizeof(res));
    res.action = VIO_FRAME_SKIPPED;
    if (active && !active[q]) return;
    Sequence &s = e->seq[q];
    if (s.on_device) return;  // its frame goes through the resident path below
    int enough = 0, parallax_num = 0;
    if (n_obs[q] < 0 || (obs_stride > 0 && n_obs[q] > obs_stride)) {
      res.action = VIO_FRAME_ERROR, res.error = VIO_EINVAL;
      return;
    }
    int rc = vio_features_add_check_parallax(s.fm, s.frame_count, obs + (size_t)q * obs_stride, n_obs[q], &enough,
                                             &parallax_num, &s.last_track_num);
    if (rc != VIO_OK) {
      // the landmark store may hold part of this frame while the window did not advance: every later frame would be
      // refused as out of order (VIO_ESTATE). Restart the sequence, as after a device error.
      clear_state(e, s);
      res.action = VIO_FRAME_ERROR, res.error = rc;
      return;
    }
    s.marginalization_flag = enough ? VIO_MARGIN_OLD : VIO_MARGIN_SECOND_NEW;
    res.marginalization_flag = s.marginalization_flag;
    res.track_num = s.last_track_num;
    vio_features_count(s.fm, &res.n_features);
    s.Headers[s.frame_count] = headers[q];
    bool solve = false;
    if (s.solver_flag == VIO_SOLVER_INITIAL) {
      {  // ImageFrame imageframe(image_msg, header); pre_integration = tmp_pre_integration; a fresh one starts (VINS.cpp:402-404)
        init::Frame f = s.tmp_valid ? s.tmp : init::Frame();
        f.header = headers[q];
        f.points.assign(obs + (size_t)q * obs_stride, obs + (size_t)q * obs_stride + n_obs[q]);
        s.all_image_frame[headers[q]] = f;
        s.tmp = init::Frame();
        memcpy(s.tmp.lin_acc, s.acc_0, 24), memcpy(s.tmp.lin_gyr, s.gyr_0, 24);
        const double zero[3] = {0, 0, 0};
        init::repropagate(e->cfg, s.tmp, zero, zero);
        s.tmp_valid = true;
      }
      if (s.frame_count == W) {
        if (s.last_track_num < 20) {
          clear_state(e, s);
          res.action = VIO_FRAME_RESET;
          return;
        }
        bool have_init = s.init_pending;
        if (have_init)
          for (int i = 0; i < P; i++) have_init = have_init && s.init_headers[i] == s.Headers[i];
        if (have_init) {  // what solveInitial leaves behind: every state of the window (VINS.cpp:1081-1143)
          s.Ps = s.init_Ps, s.Rs = s.init_Rs, s.Vs = s.init_Vs, s.Bas = s.init_Bas, s.Bgs = s.init_Bgs;
          s.init_pending = false;
          // visualInitialAlign re-integrates every interval from the biases it found (VINS.cpp:1057-1060, with ba = 0 there)
          for (int i = 1; i < P; i++) repropagate(e, s, i, &s.Bas[3 * i], &s.Bgs[3 * i]);
          // visualInitialAlign re-triangulates every landmark with the aligned poses (VINS.cpp:1094-1100)
          int nfe = 0;
          vio_features_count(s.fm, &nfe);
          std::vector<double> minus1(nfe > 0 ? nfe : 1, -1.0);
          vio_features_clear_depth(s.fm, minus1.data(), nfe);
          vio_features_triangulate(s.fm, s.Ps.data(), s.Rs.data(), e->tic, e->ric);
          solve = true;
        } else {
          bool result = false;
          if (e->enable_init && headers[q] - s.initial_timestamp > 0.3) {  // VINS.cpp:413-417
            result = solve_initial(e, s);
            s.initial_timestamp = headers[q];
          }
          if (result) {
            solve = true;
          } else {
            slide_window(e, s);
            res.action = VIO_FRAME_WAIT_INIT;
          }
        }
      } else {
        s.frame_count++;
        res.action = VIO_FRAME_FILLING;
      }
    } else {
      vio_features_triangulate(s.fm, s.Ps.data(), s.Rs.data(), e->tic, e->ric);
      solve = true;
    }
    if (solve) {
      rc = build_window(e, s, &e->staged[q]);
      if (rc != VIO_OK) {
        // (typically VIO_ECAP: more landmarks / factors than cfg.max_features / max_factors.) The frame's observations
        // are in the landmark store but the window will not slide: without a restart add_check_parallax refuses every
        // following frame and the sequence stays stuck until the caller clears it.
        clear_state(e, s);
        res.action = VIO_FRAME_ERROR, res.error = rc;
        return;
      }
      e->wants_solve[q] = 1;
    }
  };
  // phase C, per solved sequence: double2vector, loop bookkeeping, failure detection, slide
  auto phase_c = [&](int k) {
    const int q = e->solving[k];
    Sequence &s = e->seq[q];
    VioFrameResult &res = results[q];
    const VioWindow &w = e->windows[k];
    res.stats = e->stats[k];
    res.n_factors = w.n_factors, res.n_loop_factors = s.n_loop_factors, res.n_features = w.n_features;
    take_solution(e, s, w, e->stats[k]);
    if (s.solver_flag == VIO_SOLVER_INITIAL) {
      if (s.final_cost > 200) {  // initialization failed, need reinitialize (VINS.cpp:415-424)
        s.has_prior = false;
        res.action = VIO_FRAME_INIT_FAILED;
        slide_window(e, s);
      } else {
        s.failure_occur = 0;
        s.solver_flag = VIO_SOLVER_NON_LINEAR;
        slide_window(e, s);
        vio_features_remove_failures(s.fm);
        remember_last(e, s);
        res.action = VIO_FRAME_SOLVED;
      }
    } else {
      s.failure_occur = 0;
      int reasons = 0;
      vio_failure_detection(s.last_track_num, &s.Bgs[3 * W], &s.Ps[3 * W], &s.Rs[9 * W], s.last_P, s.last_R, &reasons);
      if (reasons) {
        s.failure_occur = 1;
        clear_state(e, s);
        res.action = VIO_FRAME_FAILURE, res.failure_reasons = reasons;
        return;
      }
      slide_window(e, s);
      vio_features_remove_failures(s.fm);
      remember_last(e, s);
      res.action = VIO_FRAME_SOLVED;
    }
  };
  // The sequences form n_groups contiguous groups with a back-end context each (own stream, own resident batch, own
  // prior store). Group after group: phase A of the group on the host pool, its windows packed, uploaded and launched --
  // no device wait, so the kernels of the earlier groups run while the host prepares the later ones. Then, group after
  // group again: wait, download, phase C -- while the later groups' kernels are still running.
  auto ms_between = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  double ms_a = 0, ms_c = 0;
  const bool pipelined = e->n_groups > 1 && e->group_size >= 256;
  int g0[vio_estimator::kMaxGroups + 1] = {0};
  bool launched[vio_estimator::kMaxGroups] = {};
  int rc = VIO_OK;
  for (int g = 0; g < e->n_groups; g++) {
    const int q0 = g * e->group_size, q1 = std::min(e->n_seq, q0 + e->group_size);
    const auto ta = std::chrono::steady_clock::now();
    // (small groups: one sweep of the pool over all sequences is cheaper than a sweep per group -- measured 5.15 vs 5.4 ms
    // per frame with 2 x 128 sequences, 12.1 vs 15.5 ms the other way round with 4 x 256)
    if (!pipelined && g == 0) HostPool::get().parallel_for(e->n_seq, [&](int q) { phase_a(q); });
    if (pipelined) HostPool::get().parallel_for(q1 - q0, [&](int i) { phase_a(q0 + i); });
    g0[g] = (int)e->solving.size();
    for (int q = q0; q < q1; q++) {
      if (results[q].action == VIO_FRAME_ERROR && first_error == VIO_OK) first_error = results[q].error;
      if (!e->wants_solve[q]) continue;
      e->windows[e->solving.size()] = e->staged[q];
      e->solving.push_back(q);
    }
    g0[g + 1] = (int)e->solving.size();
    ms_a += ms_between(ta, std::chrono::steady_clock::now());
    const int ng = g0[g + 1] - g0[g];
    if (ng == 0 || rc != VIO_OK) continue;
    if (!e->be[g]) {
      rc = vio_backend_create(&e->cfg, e->group_size, &e->be[g]);
      if (rc == VIO_OK) {
        // (VIO_AMD_EST_PEERS: how many launches the layout rule should assume on the device, e.g. 2 when a front-end context
        // runs its kernels under this estimator's and should find half of every CU's LDS free)
        const char *pe = getenv("VIO_AMD_EST_PEERS");
        rc = vio_backend_set_peers(e->be[g], pe && atoi(pe) > 0 ? atoi(pe) : e->n_groups);
      }
      if (rc == VIO_OK && e->resident_priors) rc = vio_backend_reserve_priors(e->be[g], e->group_size);
    }
    if (rc == VIO_OK) rc = vio_backend_upload(e->be[g], e->windows.data() + g0[g], ng);
    if (rc == VIO_OK) rc = vio_backend_launch(e->be[g], nullptr);
    launched[g] = rc == VIO_OK;
  }
  const int n = (int)e->solving.size();
  for (int g = 0; g < e->n_groups; g++) {  // (every LAUNCHED group is waited for, also after an error in another group)
    const int ng = g0[g + 1] - g0[g];
    if (ng == 0 || !launched[g]) continue;
    // (status per group: a group whose own launch and download succeeded runs its phase C also when another group failed;
    // the first error is still what the call returns, and only the sequences of the failed / unlaunched groups restart)
    const int rd = vio_backend_download(e->be[g], e->windows.data() + g0[g], ng, e->stats.data() + g0[g]);
    if (rc == VIO_OK) rc = rd;
    if (rd != VIO_OK) continue;
    const auto tc = std::chrono::steady_clock::now();
    HostPool::get().parallel_for(ng, [&](int i) { phase_c(g0[g] + i); });
    ms_c += ms_between(tc, std::chrono::steady_clock::now());
  }
  if (any_resident) {
    const int rcr = resident_frame(e, obs, n_obs, obs_stride, headers, active, results);
    if (rcr != VIO_OK && first_error == VIO_OK) first_error = rcr;
  }
  if (rc != VIO_OK) {
    // The frame is in the landmark stores of every sequence that wanted a solve. Sequences whose phase C has run (every group
    // whose own launch and download went through) have slid their windows and stay; the others did not advance: they
    // restart rather than carry an inconsistent window (and a prior slot that may or may not have advanced) into the next call.
    for (int k = 0; k < n; k++) {
      const int q = e->solving[k];
      if (results[q].action == VIO_FRAME_SOLVED || results[q].action == VIO_FRAME_FAILURE || results[q].action == VIO_FRAME_INIT_FAILED) continue;
      clear_state(e, e->seq[q]);
      results[q].action = VIO_FRAME_ERROR, results[q].error = rc;
    }
    return rc;
  }
  const auto t_end = std::chrono::steady_clock::now();
  // (phases A and C are now interleaved with the device work of the other groups: their own sums, and the rest)
  e->ms_pre = ms_a, e->ms_post = ms_c, e->ms_solve = ms_between(t_begin, t_end) - ms_a - ms_c;
  // sequences that reached the NON_LINEAR state on the host path continue on the device
  if (e->resident)
    for (int g = 0; g < e->n_groups; g++) {
      std::vector<int> go;
      for (int q = g * e->group_size; q < std::min(e->n_seq, (g + 1) * e->group_size); q++) {
        Sequence &s = e->seq[q];
        if (!s.on_device && results[q].action == VIO_FRAME_SOLVED && resident_eligible(e, s)) go.push_back(q);
      }
      promote_group(e, g, go);
    }
  return first_error;
}

int vio_estimator_get_timing(vio_estimator_t *e, double ms[3]) {
  if (!e || !ms) return VIO_EINVAL;
  ms[0] = e->ms_pre, ms[1] = e->ms_solve, ms[2] = e->ms_post;
  return VIO_OK;
}

int vio_estimator_process_image(vio_estimator_t *e, int32_t seq, const VioObs *obs, int32_t n_obs, double header,
                                VioFrameResult *result) {
  if (!e || seq < 0 || seq >= e->n_seq || !result || n_obs < 0 || (n_obs > 0 && !obs)) return VIO_EINVAL;
  std::vector<uint8_t> active(e->n_seq, 0);
  std::vector<int32_t> n(e->n_seq, 0);
  std::vector<double> hdr(e->n_seq, 0);
  std::vector<VioFrameResult> res(e->n_seq);
  active[seq] = 1, n[seq] = n_obs, hdr[seq] = header;
  // stride 0: every sequence would read the same list, only `seq` is active
  static const VioObs none = {0, 0, 0, 0};
  int rc = vio_estimator_process_images(e, n_obs > 0 ? obs : &none, n.data(), 0, hdr.data(), active.data(), res.data());
  *result = res[seq];
  return rc;
}

int vio_estimator_get_status(vio_estimator_t *e, int32_t seq, VioEstimatorStatus *st) {
  if (!e || seq < 0 || seq >= e->n_seq || !st) return VIO_EINVAL;
  const Sequence &s = e->seq[seq];
  memset(st, 0, sizeof(*st));
  st->frame_count = s.frame_count, st->solver_flag = s.solver_flag, st->marginalization_flag = s.marginalization_flag;
  st->failure_occur = s.failure_occur, st->prior_rows = s.has_prior ? s.prior[s.cur_prior].p.n : 0;
  st->final_cost = s.final_cost;
  memcpy(st->r_drift, s.r_drift, 72), memcpy(st->t_drift, s.t_drift, 24);
  memcpy(st->relative_t, s.front.relative_t, 24);
  st->relative_q[0] = s.front.relative_q.x, st->relative_q[1] = s.front.relative_q.y;
  st->relative_q[2] = s.front.relative_q.z, st->relative_q[3] = s.front.relative_q.w;
  st->relative_yaw = s.front.relative_yaw;
  memcpy(st->loop_pose, s.front.loop_pose, sizeof(st->loop_pose));
  st->resident = s.on_device ? 1 : 0;
  return VIO_OK;
}

int vio_estimator_get_window(vio_estimator_t *e, int32_t seq, double *Ps, double *Rs, double *Vs, double *Bas, double *Bgs,
                             double *headers) {
  if (!e || seq < 0 || seq >= e->n_seq) return VIO_EINVAL;
  const Sequence &s = e->seq[seq];
  const int P = e->W + 1;
  if (Ps) memcpy(Ps, s.Ps.data(), sizeof(double) * 3 * P);
  if (Rs) memcpy(Rs, s.Rs.data(), sizeof(double) * 9 * P);
  if (Vs) memcpy(Vs, s.Vs.data(), sizeof(double) * 3 * P);
  if (Bas) memcpy(Bas, s.Bas.data(), sizeof(double) * 3 * P);
  if (Bgs) memcpy(Bgs, s.Bgs.data(), sizeof(double) * 3 * P);
  if (headers) memcpy(headers, s.Headers.data(), sizeof(double) * P);
  return VIO_OK;
}

// update_loop_correction (VINS.cpp:302-331): the window poses with the loop drift applied.
int vio_estimator_get_corrected_window(vio_estimator_t *e, int32_t seq, double *correct_Ps, double *correct_Rs) {
  if (!e || seq < 0 || seq >= e->n_seq || !correct_Ps || !correct_Rs) return VIO_EINVAL;
  const Sequence &s = e->seq[seq];
  for (int i = 0; i <= e->W; i++) {
    double t[3];
    mat3vec(s.r_drift, &s.Ps[3 * i], t);
    for (int k = 0; k < 3; k++) correct_Ps[3 * i + k] = t[k] + s.t_drift[k];
    mat3mul(s.r_drift, &s.Rs[9 * i], &correct_Rs[9 * i]);
  }
  return VIO_OK;
}

int vio_estimator_features(vio_estimator_t *e, int32_t seq, vio_features_t **fm) {
  if (!e || seq < 0 || seq >= e->n_seq || !fm) return VIO_EINVAL;
  {  // a resident sequence's list comes back to the host first (it returns to the device after its next solved frame)
    const int rc = demote(e, e->seq[seq]);
    if (rc != VIO_OK) return rc;
  }
  *fm = e->seq[seq].fm;
  return VIO_OK;
}

}  // extern "C"
