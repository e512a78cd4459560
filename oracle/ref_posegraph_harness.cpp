// oracle/ref_posegraph_harness.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// The REAL reference pose-graph solve: the cost functors FourDOFError / FourDOFWeightError, NormalizeAngle and
// AngleLocalParameterization are the reference's own (VINS_ios/loop/keyfame_database.h:62-104,271-366), compiled from
// /root/reference where they lie, and the solver is the vendored Ceres 1.12. keyfame_database.h also declares the
// KeyFrameDatabase class, whose header chain (keyframe.h) needs OpenCV and DBoW (absent here): keyframe.h is skipped
// through its own include guard (__KEY_FRAME_) and the class name is forward-declared below; nothing of that class is
// used or instantiated. What this file restates is only the glue of optimize4DoFLoopPoseGraph around those functors
// (keyfame_database.cpp:150-160 options, :216-229 parameter blocks, :250-254 and :278-283 residual blocks, :287 Solve).
#include <cstdio>
#include <list>
#include <mutex>
#include <vector>

#include <eigen3/Eigen/Dense>
using namespace Eigen;
using namespace std;
class KeyFrame;        // (only pointers to it appear in the class declaration that the header carries along)
#define __KEY_FRAME_   // keyframe.h: OpenCV / DBoW, not needed by the functors
#include "keyfame_database.h"

#include "vio_amd.h"

extern "C" int ref_posegraph_optimize(VioPoseGraph *g, int max_iterations, VioSolveStats *stats) {
  const int n = g->n_nodes;
  std::vector<double> yaw(n);
  for (int k = 0; k < n; k++) yaw[k] = g->ypr[3 * k];
  ceres::Problem problem;
  ceres::Solver::Options options;
  options.linear_solver_type = ceres::DENSE_SCHUR;
  options.max_num_iterations = max_iterations;
  options.logging_type = ceres::SILENT;
  ceres::Solver::Summary summary;
  ceres::LossFunction *loss_function = new ceres::HuberLoss(1.0);
  ceres::LocalParameterization *angle_local_parameterization = AngleLocalParameterization::Create();
  for (int k = 0; k < n; k++) {
    problem.AddParameterBlock(&yaw[k], 1, angle_local_parameterization);
    problem.AddParameterBlock(g->t + 3 * k, 3);
    if (k == g->fixed_node) {
      problem.SetParameterBlockConstant(&yaw[k]);
      problem.SetParameterBlockConstant(g->t + 3 * k);
    }
  }
  for (int e = 0; e < g->n_edges; e++) {
    const double *m = g->edge_meas + 6 * e;
    const int i = g->edge_i[e], j = g->edge_j[e];
    if (g->edge_kind[e] == 0) {
      ceres::CostFunction *cost_function = FourDOFError::Create(m[0], m[1], m[2], m[3], m[4], m[5]);
      problem.AddResidualBlock(cost_function, loss_function, &yaw[i], g->t + 3 * i, &yaw[j], g->t + 3 * j);
    } else {
      ceres::CostFunction *cost_function = FourDOFWeightError::Create(m[0], m[1], m[2], m[3], m[4], m[5]);
      problem.AddResidualBlock(cost_function, NULL, &yaw[i], g->t + 3 * i, &yaw[j], g->t + 3 * j);
    }
  }
  ceres::Solve(options, &problem, &summary);
  for (int k = 0; k < n; k++) g->ypr[3 * k] = yaw[k];
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
  return 0;
}
