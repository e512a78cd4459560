// vio_initial.h — internal interface of the initialisation (vio_initial.cpp) shared with the estimator.
#pragma once
#include <stdint.h>
#include <map>
#include <vector>

#include "vio_amd.h"
#include "vio_preint.h"

namespace vio {
namespace init {

struct Frame {  // ImageFrame (initial_aligment.hpp:24-39) + the raw samples repropagate() needs
  double header = 0;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, T[3] = {0, 0, 0};
  bool is_key_frame = false;
  std::vector<VioObs> points;  // image_msg of the frame
  host::Preint pre;            // pre_integration: from the previous frame to this one, zero linearisation biases
  double lin_acc[3] = {0, 0, 0}, lin_gyr[3] = {0, 0, 0};
  std::vector<double> dt, acc, gyr;
};

void repropagate(const VioConfig &cfg, Frame &f, const double ba[3], const double bg[3]);

// VisualIMUAlignment (initial_aligment.cpp:223-229): gyroscope bias, then velocities / gravity / scale. Bgs [W+1][3] in/out,
// g [3] out, x = [v_0 .. v_{n-1} (body frame), (tangent-plane gravity correction,) scale] out.
bool visual_imu_alignment(const VioConfig &cfg, const double tic[3], std::vector<Frame> &frames, int window_size, double *Bgs,
                          double g[3], std::vector<double> &x);

struct SfmFeature {  // SFMFeature (inital_sfm.hpp:13-21)
  bool state = false;
  int id = 0;
  std::vector<std::pair<int, std::pair<double, double>>> observation;  // (frame, normalized xy)
  double position[3] = {0, 0, 0};
};

// MotionEstimator::solveRelativeRT (motion_estimator.cpp:200-236): rotation / unit translation of the second view expressed
// in the first from >= 9 correspondences; false when fewer than 11 points end up in front of both cameras.
// R_hint (optional, 3x3): an approximate rotation of the second camera in the first, used only to choose between
// equally good solutions (planar scenes).
bool solve_relative_rt(const std::vector<double> &xy0, const std::vector<double> &xy1, double R[9], double t[3], int *inliers,
                       const double *R_hint = nullptr);

// The same as the reference computes it (vio_fivepoint.cpp): cv::findEssentialMat -- five-point minimal solver inside RANSAC
// with OpenCV's RNG((uint64)-1) stream, threshold 1.0, confidence 0.999 -- then cv::recoverPose's cheirality count.
bool solve_relative_rt_five_point(const std::vector<double> &xy0, const std::vector<double> &xy1, double R[9], double t[3],
                                  int *inliers);
int five_point_kernel(const double q1[5][2], const double q2[5][2], double E[10][9]);
bool find_essential_ransac(const double *xy0, const double *xy1, int count, double prob, double threshold, double E[9],
                           uint8_t *mask_out);
int recover_pose(const double E[9], const double *xy0, const double *xy1, int n, double R[9], double t[3]);

// GlobalSFM::construct (inital_sfm.cpp:117-316): q [frame_num][4] (x y z w), T [frame_num][3] = camera-to-frame-l poses.
bool sfm_construct(int frame_num, double *q, double *T, int l, const double relative_R[9], const double relative_T[3],
                   std::vector<SfmFeature> &sfm_f, std::map<int, std::vector<double>> &tracked_points);

// cv::solvePnP(..., useExtrinsicGuess = true, ITERATIVE) with K = I: refines R, t (world -> camera) from the guess.
bool pnp_refine(const std::vector<double> &pts3, const std::vector<double> &pts2, double R[9], double t[3]);

}  // namespace init
}  // namespace vio
