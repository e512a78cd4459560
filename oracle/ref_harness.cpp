// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Drives the REAL reference back-end: vendored Ceres 1.12 + Eigen 3.3.0 and the verbatim VINS_ios
// factor sources (projection_facor.cpp, imu_factor.h, integration_base.h, marginalization_factor.cpp,
// pose_local_parameterization.cpp, utility.{hpp,cpp}) compiled from /root/reference by oracle/Makefile.
//
// VINS.cpp itself cannot be compiled here (it pulls OpenCV through feature_manager.hpp:17 and
// draw_result.hpp), so this file restates ONLY the glue of VINS::solve_ceres around those sources:
//   problem assembly           VINS.cpp:482-567   (parameter blocks, prior, IMU and projection factors)
//   loop-closure pose/factors  VINS.cpp:571-637   (as extra factors whose target is the loop pose)
//   solver options             VINS.cpp:639-659   (max_solver_time disabled => deterministic)
//   new2old gauge fix          VINS.cpp:131-212
//   marginalization            VINS.cpp:690-830
// The window comes in through the same C structs the product ABI uses (include/vio_amd.h), so tests can
// hand identical inputs to the reference, to the CPU restatement and to the HIP path.

#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#include <ceres/ceres.h>

#include "global_param.hpp"
#include "imu_factor.h"
#include "imu_factor_pnp.h"
#include "perspective_factor.hpp"
#include "integration_base.h"
#include "marginalization_factor.hpp"
#include "pose_local_parameterization.hpp"
#include "projection_facor.hpp"
#include "utility.hpp"

#include "vio_amd.h"

using namespace Eigen;

namespace {

bool config_matches_reference_macros(const VioConfig *cfg) {
  return cfg->gravity == GRAVITY && cfg->acc_n == ACC_N && cfg->acc_w == ACC_W && cfg->gyr_n == GYR_N &&
         cfg->gyr_w == GYR_W && cfg->cauchy_a == 1.0;
}

IntegrationBase *make_integration(const VioPreintegration &p) {
  Vector3d z = Vector3d::Zero();
  IntegrationBase *ib = new IntegrationBase(z, z, Vector3d(p.linearized_ba), Vector3d(p.linearized_bg));
  ib->sum_dt = p.sum_dt;
  ib->delta_p = Vector3d(p.delta_p);
  ib->delta_q = Quaterniond(p.delta_q[3], p.delta_q[0], p.delta_q[1], p.delta_q[2]);
  ib->delta_v = Vector3d(p.delta_v);
  for (int r = 0; r < 15; r++)
    for (int c = 0; c < 15; c++) {
      ib->jacobian(r, c) = p.jacobian[r * 15 + c];
      ib->covariance(r, c) = p.covariance[r * 15 + c];
    }
  return ib;
}

void export_integration(const IntegrationBase &ib, VioPreintegration *out) {
  out->sum_dt = ib.sum_dt;
  for (int k = 0; k < 3; k++) {
    out->delta_p[k] = ib.delta_p(k);
    out->delta_v[k] = ib.delta_v(k);
    out->linearized_ba[k] = ib.linearized_ba(k);
    out->linearized_bg[k] = ib.linearized_bg(k);
  }
  out->delta_q[0] = ib.delta_q.x();
  out->delta_q[1] = ib.delta_q.y();
  out->delta_q[2] = ib.delta_q.z();
  out->delta_q[3] = ib.delta_q.w();
  for (int r = 0; r < 15; r++)
    for (int c = 0; c < 15; c++) {
      out->jacobian[r * 15 + c] = ib.jacobian(r, c);
      out->covariance[r * 15 + c] = ib.covariance(r, c);
    }
}

struct ParaArrays {
  int W;
  std::vector<double> pose, sb, feat;
  double ex[7];
  double loop[7];
  double *Pose(int i) { return &pose[7 * i]; }
  double *SB(int i) { return &sb[9 * i]; }
  double *Feat(int i) { return &feat[i]; }
};

// Maps a prior block descriptor (kind,index) to the para_* address it is bound to.
double *block_addr(ParaArrays &pa, int kind, int index) {
  switch (kind) {
    case VIO_BLOCK_POSE: return pa.Pose(index);
    case VIO_BLOCK_SPEEDBIAS: return pa.SB(index);
    default: return pa.ex;
  }
}

MarginalizationInfo *make_prior(const VioPrior *p, ParaArrays &pa, std::vector<double *> *blocks) {
  MarginalizationInfo *info = new MarginalizationInfo();
  info->m = 0;
  info->n = p->n;
  for (int b = 0; b < p->n_blocks; b++) {
    int gsize = p->block_kind[b] == VIO_BLOCK_SPEEDBIAS ? 9 : 7;
    info->keep_block_size.push_back(gsize);
    info->keep_block_idx.push_back(p->block_offset[b]);  // m == 0
    double *d = new double[gsize];
    memcpy(d, p->block_x0 + 9 * b, sizeof(double) * gsize);
    info->keep_block_data.push_back(d);
    // ~MarginalizationInfo frees parameter_block_data entries; register the copy there.
    info->parameter_block_data[reinterpret_cast<long>(d)] = d;
    blocks->push_back(block_addr(pa, p->block_kind[b], p->block_index[b]));
  }
  info->linearized_jacobians.resize(p->n, p->n);
  info->linearized_residuals.resize(p->n);
  for (int r = 0; r < p->n; r++) {
    info->linearized_residuals(r) = p->linearized_residuals[r];
    for (int c = 0; c < p->n; c++) info->linearized_jacobians(r, c) = p->linearized_jacobians[r * p->n + c];
  }
  return info;
}

// old2new's Rs -> quaternion step (VINS.cpp:96-101): Quaterniond q{Rs[i]}.
void rs_to_para(const Matrix3d &R, const Vector3d &P, double *pose) {
  pose[0] = P.x();
  pose[1] = P.y();
  pose[2] = P.z();
  Quaterniond q{R};
  pose[3] = q.x();
  pose[4] = q.y();
  pose[5] = q.z();
  pose[6] = q.w();
}

}  // namespace

extern "C" {

const char *ref_describe(void) { return "reference: vendored ceres-solver 1.12.0 + eigen 3.3.0 + VINS_ios factors"; }

int ref_preintegrate(const VioConfig *cfg, const double acc_0[3], const double gyr_0[3], const double ba[3],
                     const double bg[3], int32_t n, const double *dt, const double *acc, const double *gyr,
                     VioPreintegration *out) {
  if (!config_matches_reference_macros(cfg)) return VIO_EINVAL;
  Vector3d a0(acc_0), g0(gyr_0), vba(ba), vbg(bg);
  IntegrationBase ib(a0, g0, vba, vbg);
  for (int i = 0; i < n; i++) ib.push_back(dt[i], Vector3d(acc + 3 * i), Vector3d(gyr + 3 * i));
  export_integration(ib, out);
  return VIO_OK;
}

// ProjectionFactor::Evaluate on raw blocks. jac: [2x7 | 2x7 | 2x7 | 2x1] row-major per block, back to back.
int ref_eval_projection(const VioConfig *cfg, const double *pose_i, const double *pose_j, const double *ex,
                        const double *inv_depth, const double *pts_i, const double *pts_j, double *res,
                        double *jac) {
  ProjectionFactor::sqrt_info = cfg->fx / 1.5 * Matrix2d::Identity();
  ProjectionFactor f{Vector3d(pts_i), Vector3d(pts_j)};
  const double *params[4] = {pose_i, pose_j, ex, inv_depth};
  double *jacs[4] = {jac, jac + 14, jac + 28, jac + 42};
  return f.Evaluate(params, res, jac ? jacs : NULL) ? VIO_OK : VIO_EINVAL;
}

// IMUFactor::Evaluate. jac: [15x7 | 15x9 | 15x7 | 15x9] row-major per block.
int ref_eval_imu(const VioConfig *cfg, const VioPreintegration *pre, const double *pose_i, const double *sb_i,
                 const double *pose_j, const double *sb_j, double *res, double *jac) {
  if (!config_matches_reference_macros(cfg)) return VIO_EINVAL;
  IntegrationBase *ib = make_integration(*pre);
  IMUFactor f(ib);
  const double *params[4] = {pose_i, sb_i, pose_j, sb_j};
  double *jacs[4] = {jac, jac + 105, jac + 240, jac + 345};
  bool ok = f.Evaluate(params, res, jac ? jacs : NULL);
  delete ib;
  return ok ? VIO_OK : VIO_EINVAL;
}

// The whole of VINS::solve_ceres on one window.
int ref_solve_window(const VioConfig *cfg, VioWindow *w, VioSolveStats *stats) {
  if (!config_matches_reference_macros(cfg)) return VIO_EINVAL;
  const int W = w->window_size;
  const int P = W + 1;
  FOCUS_LENGTH_X = cfg->fx;
  ProjectionFactor::sqrt_info = FOCUS_LENGTH_X / 1.5 * Matrix2d::Identity();  // VINS.cpp:29-32

  ParaArrays pa;
  pa.W = W;
  pa.pose.assign(w->pose, w->pose + 7 * P);
  pa.sb.assign(w->speed_bias, w->speed_bias + 9 * P);
  pa.feat.assign(w->inv_depth, w->inv_depth + w->n_features);
  pa.feat.resize(cfg->max_features > w->n_features ? cfg->max_features : w->n_features, 0.0);
  memcpy(pa.ex, w->ex_pose, sizeof(pa.ex));

  // State before the solve, as new2old() reads it (Rs[0], Ps[0]).
  Quaterniond q0_in(pa.Pose(0)[6], pa.Pose(0)[3], pa.Pose(0)[4], pa.Pose(0)[5]);
  Matrix3d Rs0_in = q0_in.normalized().toRotationMatrix();
  Vector3d Ps0_in(pa.Pose(0)[0], pa.Pose(0)[1], pa.Pose(0)[2]);

  ceres::Problem problem;
  ceres::LossFunction *loss_function = new ceres::CauchyLoss(1.0);
  for (int i = 0; i < P; i++) {
    problem.AddParameterBlock(pa.Pose(i), SIZE_POSE, new PoseLocalParameterization());
    problem.AddParameterBlock(pa.SB(i), SIZE_SPEEDBIAS);
  }
  problem.AddParameterBlock(pa.ex, SIZE_POSE, new PoseLocalParameterization());
  problem.SetParameterBlockConstant(pa.ex);
  for (size_t i = 0; i < pa.feat.size(); i++) problem.AddParameterBlock(pa.Feat((int)i), SIZE_FEATURE);

  MarginalizationInfo *last_info = nullptr;
  std::vector<double *> last_blocks;
  if (w->prior) {
    last_info = make_prior(w->prior, pa, &last_blocks);
    problem.AddResidualBlock(new MarginalizationFactor(last_info), NULL, last_blocks);
  }

  std::vector<IntegrationBase *> pre(P, nullptr);
  for (int i = 0; i < W; i++) {
    int j = i + 1;
    pre[j] = make_integration(w->preint[i]);
    problem.AddResidualBlock(new IMUFactor(pre[j]), NULL, pa.Pose(i), pa.SB(i), pa.Pose(j), pa.SB(j));
  }

  bool have_loop = false;
  for (int k = 0; k < w->n_factors; k++)
    if (w->factor_target[k] == P) have_loop = true;
  if (have_loop) {
    if (w->loop_frame < 0 || w->loop_frame >= W) return VIO_EINVAL;
    for (int k = 0; k < 7; k++) pa.loop[k] = pa.Pose(w->loop_frame)[k];  // VINS.cpp:590-591
    problem.AddParameterBlock(pa.loop, SIZE_POSE, new PoseLocalParameterization());
  }
  // Window factors first, loop factors afterwards: the order VINS.cpp:528-567 then :597-631 adds them.
  for (int pass = 0; pass < 2; pass++)
    for (int k = 0; k < w->n_factors; k++) {
      bool is_loop = w->factor_target[k] == P;
      if ((pass == 1) != is_loop) continue;
      ProjectionFactor *f = new ProjectionFactor(Vector3d(w->factor_pts_i + 3 * k), Vector3d(w->factor_pts_j + 3 * k));
      double *target = is_loop ? pa.loop : pa.Pose(w->factor_target[k]);
      problem.AddResidualBlock(f, loss_function, pa.Pose(w->factor_host[k]), target, pa.ex,
                               pa.Feat(w->factor_feature[k]));
    }

  ceres::Solver::Options options;
  options.linear_solver_type = ceres::DENSE_SCHUR;
  options.num_threads = 1;
  options.trust_region_strategy_type = ceres::DOGLEG;
  options.use_explicit_schur_complement = true;
  options.minimizer_progress_to_stdout = false;
  options.max_num_iterations = cfg->max_iterations;
  options.max_solver_time_in_seconds = 1e9;  // reference: SOLVER_TIME budget (VINS.cpp:648-653), non-deterministic
  ceres::Solver::Summary summary;
  ceres::Solve(options, &problem, &summary);

  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->initial_cost = summary.initial_cost;
    stats->final_cost = summary.final_cost;
    stats->iterations = (int)summary.iterations.size();
    stats->termination = summary.termination_type == ceres::CONVERGENCE ? 1
                         : summary.termination_type == ceres::NO_CONVERGENCE ? 0 : 2;
    stats->num_successful_steps = summary.num_successful_steps;
    stats->num_unsuccessful_steps = summary.num_unsuccessful_steps;
    for (int i = 0; i < stats->iterations && i < VIO_MAX_TRACE; i++) {
      const ceres::IterationSummary &it = summary.iterations[i];
      stats->it_cost[i] = it.cost;
      stats->it_radius[i] = it.trust_region_radius;
      stats->it_step_norm[i] = it.step_norm;
      stats->it_relative_decrease[i] = it.relative_decrease;
      stats->it_gradient_max_norm[i] = it.gradient_max_norm;
      stats->it_flags[i] = (it.step_is_valid ? 1 : 0) | (it.step_is_successful ? 2 : 0);
    }
  }
  if (w->raw_pose) memcpy(w->raw_pose, pa.pose.data(), sizeof(double) * 7 * P);
  if (w->raw_speed_bias) memcpy(w->raw_speed_bias, pa.sb.data(), sizeof(double) * 9 * P);
  if (w->raw_inv_depth) memcpy(w->raw_inv_depth, pa.feat.data(), sizeof(double) * w->n_features);
  if (have_loop && w->loop_pose) memcpy(w->loop_pose, pa.loop, sizeof(pa.loop));

  // ---- new2old (VINS.cpp:131-212) ------------------------------------------------------------
  Vector3d origin_R0 = Utility::R2ypr(Rs0_in);
  Vector3d origin_P0 = Ps0_in;
  if (w->use_origin_override) {
    origin_R0 = Vector3d(w->origin_yaw_deg, 0, 0);
    origin_P0 = Vector3d(w->origin_p);
  }
  Vector3d origin_R00 = Utility::R2ypr(
      Quaterniond(pa.Pose(0)[6], pa.Pose(0)[3], pa.Pose(0)[4], pa.Pose(0)[5]).toRotationMatrix());
  double y_diff = origin_R0.x() - origin_R00.x();
  Matrix3d rot_diff = Utility::ypr2R(Vector3d(y_diff, 0, 0));
  std::vector<Matrix3d> Rs(P);
  std::vector<Vector3d> Ps(P), Vs(P), Bas(P), Bgs(P);
  for (int i = 0; i < P; i++) {
    double *pp = pa.Pose(i), *sb = pa.SB(i);
    Rs[i] = rot_diff * Quaterniond(pp[6], pp[3], pp[4], pp[5]).normalized().toRotationMatrix();
    Ps[i] = rot_diff * Vector3d(pp[0] - pa.Pose(0)[0], pp[1] - pa.Pose(0)[1], pp[2] - pa.Pose(0)[2]) + origin_P0;
    Vs[i] = rot_diff * Vector3d(sb[0], sb[1], sb[2]);
    Bas[i] = Vector3d(sb[3], sb[4], sb[5]);
    Bgs[i] = Vector3d(sb[6], sb[7], sb[8]);
  }
  // f_manager.setDepth(dep) then getDepthVector(): 1/(1/x) (feature_manager.cpp:300-349)
  for (int i = 0; i < w->n_features; i++) {
    double estimated_depth = 1.0 / pa.feat[i];
    pa.feat[i] = 1. / estimated_depth;
  }
  // second old2new() (VINS.cpp:693)
  for (int i = 0; i < P; i++) {
    rs_to_para(Rs[i], Ps[i], pa.Pose(i));
    double *sb = pa.SB(i);
    for (int k = 0; k < 3; k++) {
      sb[k] = Vs[i](k);
      sb[3 + k] = Bas[i](k);
      sb[6 + k] = Bgs[i](k);
    }
  }
  memcpy(w->pose, pa.pose.data(), sizeof(double) * 7 * P);
  memcpy(w->speed_bias, pa.sb.data(), sizeof(double) * 9 * P);
  memcpy(w->inv_depth, pa.feat.data(), sizeof(double) * w->n_features);

  // ---- marginalization (VINS.cpp:690-830) -------------------------------------------------------
  MarginalizationInfo *new_info = nullptr;
  std::unordered_map<long, double *> addr_shift;
  if (w->marginalization_flag == VIO_MARGIN_OLD) {
    new_info = new MarginalizationInfo();
    if (last_info) {
      std::vector<int> drop_set;
      for (int i = 0; i < (int)last_blocks.size(); i++)
        if (last_blocks[i] == pa.Pose(0) || last_blocks[i] == pa.SB(0)) drop_set.push_back(i);
      new_info->addResidualBlockInfo(
          new ResidualBlockInfo(new MarginalizationFactor(last_info), NULL, last_blocks, drop_set));
    }
    new_info->addResidualBlockInfo(new ResidualBlockInfo(
        new IMUFactor(pre[1]), NULL, std::vector<double *>{pa.Pose(0), pa.SB(0), pa.Pose(1), pa.SB(1)},
        std::vector<int>{0, 1}));
    for (int k = 0; k < w->n_factors; k++) {
      if (w->factor_host[k] != 0 || w->factor_target[k] == P) continue;  // only window factors hosted at frame 0
      ProjectionFactor *f = new ProjectionFactor(Vector3d(w->factor_pts_i + 3 * k), Vector3d(w->factor_pts_j + 3 * k));
      new_info->addResidualBlockInfo(new ResidualBlockInfo(
          f, loss_function,
          std::vector<double *>{pa.Pose(0), pa.Pose(w->factor_target[k]), pa.ex, pa.Feat(w->factor_feature[k])},
          std::vector<int>{0, 3}));
    }
    new_info->preMarginalize();
    new_info->marginalize();
    for (int i = 1; i <= W; i++) {
      addr_shift[reinterpret_cast<long>(pa.Pose(i))] = pa.Pose(i - 1);
      addr_shift[reinterpret_cast<long>(pa.SB(i))] = pa.SB(i - 1);
    }
    addr_shift[reinterpret_cast<long>(pa.ex)] = pa.ex;
  } else if (w->marginalization_flag == VIO_MARGIN_SECOND_NEW) {
    if (last_info && std::count(last_blocks.begin(), last_blocks.end(), pa.Pose(W - 1))) {
      new_info = new MarginalizationInfo();
      std::vector<int> drop_set;
      for (int i = 0; i < (int)last_blocks.size(); i++)
        if (last_blocks[i] == pa.Pose(W - 1)) drop_set.push_back(i);
      new_info->addResidualBlockInfo(
          new ResidualBlockInfo(new MarginalizationFactor(last_info), NULL, last_blocks, drop_set));
      new_info->preMarginalize();
      new_info->marginalize();
      for (int i = 0; i <= W; i++) {
        if (i == W - 1) continue;
        if (i == W) {
          addr_shift[reinterpret_cast<long>(pa.Pose(i))] = pa.Pose(i - 1);
          addr_shift[reinterpret_cast<long>(pa.SB(i))] = pa.SB(i - 1);
        } else {
          addr_shift[reinterpret_cast<long>(pa.Pose(i))] = pa.Pose(i);
          addr_shift[reinterpret_cast<long>(pa.SB(i))] = pa.SB(i);
        }
      }
      addr_shift[reinterpret_cast<long>(pa.ex)] = pa.ex;
    }
  }

  if (w->next_prior) {
    VioPrior *np = w->next_prior;
    if (!new_info) {
      // MARGIN_SECOND_NEW without a prior touching pose[W-1] keeps the old prior untouched
      // (VINS.cpp:778-779); signalled with n = -1.
      np->n = -1;
      np->n_blocks = 0;
    } else {
      std::vector<double *> blocks = new_info->getParameterBlocks(addr_shift);
      np->n = new_info->n;
      np->n_blocks = (int)blocks.size();
      if (np->n_blocks > VIO_MAX_PRIOR_BLOCKS) return VIO_ECAP;
      for (int b = 0; b < np->n_blocks; b++) {
        double *a = blocks[b];
        int kind = -1, index = 0;
        if (a == pa.ex) kind = VIO_BLOCK_EXPOSE;
        for (int i = 0; i < P && kind < 0; i++) {
          if (a == pa.Pose(i)) kind = VIO_BLOCK_POSE, index = i;
          else if (a == pa.SB(i)) kind = VIO_BLOCK_SPEEDBIAS, index = i;
        }
        if (kind < 0) return VIO_EINVAL;
        np->block_kind[b] = kind;
        np->block_index[b] = index;
        np->block_offset[b] = new_info->keep_block_idx[b] - new_info->m;
        memset(np->block_x0 + 9 * b, 0, sizeof(double) * 9);
        memcpy(np->block_x0 + 9 * b, new_info->keep_block_data[b], sizeof(double) * new_info->keep_block_size[b]);
      }
      for (int r = 0; r < np->n; r++) {
        np->linearized_residuals[r] = new_info->linearized_residuals(r);
        for (int c = 0; c < np->n; c++) np->linearized_jacobians[r * np->n + c] = new_info->linearized_jacobians(r, c);
      }
    }
  }
  // Ownership: Ceres' Problem deletes the cost functions it was given; ResidualBlockInfo-owned cost
  // functions are deleted by ~MarginalizationInfo. The MarginalizationFactor inside new_info points at
  // last_info, so new_info goes first.
  // (IMUFactor(pre[1]) in new_info only borrows pre[1].)
  if (new_info) delete new_info;
  // `problem` still references last_info through its MarginalizationFactor until it goes out of scope;
  // residual blocks are never evaluated again, so freeing here is safe.
  if (last_info) delete last_info;
  for (auto *p : pre) delete p;
  return VIO_OK;
}


// vinsPnP::solve_ceres (vins_pnp.cpp:264-341) with the reference's own factor classes (IMUFactorPnP, PerspectiveFactor)
// and the vendored Ceres; vins_pnp.cpp itself cannot be compiled here (vins_pnp.hpp pulls in an OpenCV header), so the
// problem is assembled here line for line. The 0.01 s wall-clock limit is lifted (non-deterministic).
int ref_pnp_solve(const VioConfig *cfg, VioPnpWindow *w, VioSolveStats *stats) {
  const int n = w->n_frames;
  PerspectiveFactor::sqrt_info = cfg->fx / 1.5 * Matrix2d::Identity();  // vinsPnP::setIMUModel
  std::vector<double> pose(w->pose, w->pose + 7 * n), speed(w->speed, w->speed + 3 * n), bias(w->bias, w->bias + 6 * n);
  double ex[7];
  memcpy(ex, w->ex_pose, sizeof(ex));
  ceres::Problem problem;
  ceres::LossFunction *loss_function = new ceres::CauchyLoss(1.0);
  for (int i = 0; i < n; i++) {
    problem.AddParameterBlock(&pose[7 * i], SIZE_POSE, new PoseLocalParameterization());
    problem.AddParameterBlock(&speed[3 * i], 3);
    problem.AddParameterBlock(&bias[6 * i], 6);
    if (w->fixed[i]) {
      problem.SetParameterBlockConstant(&pose[7 * i]);
      problem.SetParameterBlockConstant(&speed[3 * i]);
    }
    problem.SetParameterBlockConstant(&bias[6 * i]);
  }
  problem.AddParameterBlock(ex, SIZE_POSE, new PoseLocalParameterization());
  problem.SetParameterBlockConstant(ex);
  std::vector<IntegrationBase *> pre(n, nullptr);
  for (int i = 0; i + 1 < n; i++) {
    int j = i + 1;
    pre[j] = make_integration(w->preint[i]);
    problem.AddResidualBlock(new IMUFactorPnP(pre[j]), NULL, &pose[7 * i], &speed[3 * i], &bias[6 * i], &pose[7 * j], &speed[3 * j],
                             &bias[6 * j]);
  }
  for (int i = 0; i < n; i++)
    for (int m = w->feat_start[i]; m < w->feat_start[i + 1]; m++) {
      PerspectiveFactor *f = new PerspectiveFactor(Vector2d(w->observation[2 * m], w->observation[2 * m + 1]),
                                                   Vector3d(w->position[3 * m], w->position[3 * m + 1], w->position[3 * m + 2]),
                                                   w->track_num[m]);
      problem.AddResidualBlock(f, loss_function, &pose[7 * i], ex);
    }
  ceres::Solver::Options options;
  options.linear_solver_type = ceres::DENSE_SCHUR;
  options.num_threads = 1;
  options.trust_region_strategy_type = ceres::DOGLEG;
  options.use_explicit_schur_complement = true;
  options.minimizer_progress_to_stdout = false;
  options.max_num_iterations = 5;
  options.max_solver_time_in_seconds = 1e9;
  ceres::Solver::Summary summary;
  ceres::Solve(options, &problem, &summary);
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->initial_cost = summary.initial_cost;
    stats->final_cost = summary.final_cost;
    stats->iterations = (int)summary.iterations.size();
    stats->termination = summary.termination_type == ceres::CONVERGENCE ? 1 : summary.termination_type == ceres::NO_CONVERGENCE ? 0 : 2;
    stats->num_successful_steps = summary.num_successful_steps;
    stats->num_unsuccessful_steps = summary.num_unsuccessful_steps;
    for (int i = 0; i < stats->iterations && i < VIO_MAX_TRACE; i++) {
      const ceres::IterationSummary &it = summary.iterations[i];
      stats->it_cost[i] = it.cost, stats->it_radius[i] = it.trust_region_radius, stats->it_step_norm[i] = it.step_norm;
      stats->it_relative_decrease[i] = it.relative_decrease, stats->it_gradient_max_norm[i] = it.gradient_max_norm;
      stats->it_flags[i] = (it.step_is_valid ? 1 : 0) | (it.step_is_successful ? 2 : 0);
    }
  }
  // new2old (vins_pnp.cpp:137-172): normalized quaternions, no gauge change
  for (int i = 0; i < n; i++) {
    Quaterniond q = Quaterniond(pose[7 * i + 6], pose[7 * i + 3], pose[7 * i + 4], pose[7 * i + 5]).normalized();
    // Rs[i] = q.toRotationMatrix(); old2new would turn it back into Quaterniond{Rs[i]}
    Quaterniond q2{q.toRotationMatrix()};
    w->pose[7 * i] = pose[7 * i], w->pose[7 * i + 1] = pose[7 * i + 1], w->pose[7 * i + 2] = pose[7 * i + 2];
    w->pose[7 * i + 3] = q2.x(), w->pose[7 * i + 4] = q2.y(), w->pose[7 * i + 5] = q2.z(), w->pose[7 * i + 6] = q2.w();
    for (int k = 0; k < 3; k++) w->speed[3 * i + k] = speed[3 * i + k];
  }
  for (IntegrationBase *p : pre) delete p;
  return VIO_OK;
}
}  // extern "C"
