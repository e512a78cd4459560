// oracle/ref_sfm_harness.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// The visual-only bundle adjustment that closes GlobalSFM::construct (VINS_ios/inital_sfm.cpp:229-296) and the linear
// two-view triangulation (inital_sfm.cpp:5-21), run on the REAL vendored Ceres 1.12 / Eigen 3.3.0.
//
// What is real and what is restated: inital_sfm.hpp itself cannot be compiled in this image -- it is `#pragma once` (no
// include guard to skip) and includes <opencv2/...> (absent; stand-in headers are not allowed), and the reference's
// inital_sfm.cpp calls cv::solvePnP. So the two pieces below are RESTATED from the reference text:
//   * the residual functor  (ReprojectionError3D, inital_sfm.hpp:25-54: QuaternionRotatePoint, + t, perspective divide)
//   * the problem set-up    (inital_sfm.cpp:231-276: QuaternionParameterization on every rotation, frame l's rotation
//                            and the translations of frames l and frame_num-1 constant, DENSE_SCHUR, 0.3 s time limit)
//   * the design matrix of triangulatePoint (inital_sfm.cpp:8-13), whose null vector Eigen's JacobiSVD delivers.
// Everything underneath -- automatic differentiation, the quaternion parameterization, the trust-region minimizer, the
// Schur-complement solver, JacobiSVD -- is the reference's own third-party code. Parity of the product's
// vio_init_bundle_adjust / vio_init_triangulate_point against this file is therefore "pinned to Ceres, functor restated".
#include <cstring>
#include <vector>

#include <ceres/ceres.h>
#include <ceres/rotation.h>
#include <eigen3/Eigen/Dense>

#include "vio_amd.h"

namespace {

struct NormalizedReprojection {  // residual of one observation of a landmark in normalized image coordinates
  double u, v;
  template <typename T>
  bool operator()(const T *const q_wxyz, const T *const t, const T *const X, T *res) const {
    T Y[3];
    ceres::QuaternionRotatePoint(q_wxyz, X, Y);
    Y[0] += t[0], Y[1] += t[1], Y[2] += t[2];
    res[0] = Y[0] / Y[2] - T(u);
    res[1] = Y[1] / Y[2] - T(v);
    return true;
  }
};

}  // namespace

// c_rotation [frame_num][4] (w x y z, world -> camera), c_translation [frame_num][3], points [n_points][3]: in/out.
// Landmark j (point_ok[j] != 0) has observations obs_frame / obs_xy [feat_start[j], feat_start[j+1]).
extern "C" int ref_sfm_bundle_adjust(int frame_num, int l, double *c_rotation, double *c_translation, int n_points, double *points,
                                     const unsigned char *point_ok, const int *feat_start, const int *obs_frame, const double *obs_xy,
                                     VioSolveStats *stats) {
  ceres::Problem problem;
  ceres::LocalParameterization *local_parameterization = new ceres::QuaternionParameterization();
  for (int i = 0; i < frame_num; i++) {
    problem.AddParameterBlock(c_rotation + 4 * i, 4, local_parameterization);
    problem.AddParameterBlock(c_translation + 3 * i, 3);
    if (i == l) problem.SetParameterBlockConstant(c_rotation + 4 * i);
    if (i == l || i == frame_num - 1) problem.SetParameterBlockConstant(c_translation + 3 * i);
  }
  for (int j = 0; j < n_points; j++) {
    if (!point_ok[j]) continue;
    for (int k = feat_start[j]; k < feat_start[j + 1]; k++) {
      ceres::CostFunction *cost = new ceres::AutoDiffCostFunction<NormalizedReprojection, 2, 4, 3, 3>(
          new NormalizedReprojection{obs_xy[2 * k], obs_xy[2 * k + 1]});
      problem.AddResidualBlock(cost, NULL, c_rotation + 4 * obs_frame[k], c_translation + 3 * obs_frame[k], points + 3 * j);
    }
  }
  ceres::Solver::Options options;
  options.linear_solver_type = ceres::DENSE_SCHUR;
  options.max_solver_time_in_seconds = 0.3;
  options.logging_type = ceres::SILENT;
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
  // 1 = the reference's acceptance test (inital_sfm.cpp:279): CONVERGENCE or final_cost < 3e-3
  return (summary.termination_type == ceres::CONVERGENCE || summary.final_cost < 3e-03) ? 1 : 0;
}

// pose0 / pose1: 3x4 row-major [R | t] (world -> camera).
extern "C" void ref_sfm_triangulate_point(const double pose0[12], const double pose1[12], const double xy0[2], const double xy1[2],
                                          double point[3]) {
  typedef Eigen::Matrix<double, 3, 4, Eigen::RowMajor> Pose;
  const Eigen::Map<const Pose> P0(pose0), P1(pose1);
  Eigen::Matrix4d D;
  D.row(0) = xy0[0] * P0.row(2) - P0.row(0);
  D.row(1) = xy0[1] * P0.row(2) - P0.row(1);
  D.row(2) = xy1[0] * P1.row(2) - P1.row(0);
  D.row(3) = xy1[1] * P1.row(2) - P1.row(1);
  const Eigen::Vector4d h = D.jacobiSvd(Eigen::ComputeFullV).matrixV().rightCols<1>();
  for (int k = 0; k < 3; k++) point[k] = h(k) / h(3);
}
