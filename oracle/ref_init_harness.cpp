// ref_init_harness.cpp — TEST-ONLY driver of the REAL reference initialisation code that builds without OpenCV:
// VisualIMUAlignment and its helpers (VINS_ios/initial_aligment.cpp), compiled where it lies (oracle/Makefile, same
// include-guard trick as feature_manager.cpp). Nothing here is product code.
#include <map>
#include <vector>

#include "initial_aligment.hpp"
#include "vio_amd.h"

bool VisualIMUAlignment(map<double, ImageFrame> &all_image_frame, Vector3d *Bgs, Vector3d &g, VectorXd &x);

extern "C" int ref_visual_imu_alignment(const double tic[3], const VioInitFrame *frames, int32_t n_frames, int32_t window_size,
                                        double *Bgs, double g_out[3], double *x_out, int32_t *ok) {
  if (window_size != WINDOW_SIZE) return VIO_EINVAL;  // compile-time constant in the reference (global_param.hpp:28)
  TIC_X = tic[0], TIC_Y = tic[1], TIC_Z = tic[2];
  map<double, ImageFrame> all;
  for (int i = 0; i < n_frames; i++) {
    const VioInitFrame &s = frames[i];
    map<int, Vector3d> none;
    ImageFrame f(none, s.header);
    f.R = Eigen::Map<const Eigen::Matrix<double, 3, 3, Eigen::RowMajor>>(s.R);
    f.T = Eigen::Map<const Vector3d>(s.T);
    f.is_key_frame = s.is_key_frame != 0;
    // tmp_pre_integration = new IntegrationBase{acc_0, gyr_0, 0, 0} + push_back per sample (VINS.cpp:333-358, 404)
    f.pre_integration = new IntegrationBase{Eigen::Map<const Vector3d>(s.acc_0), Eigen::Map<const Vector3d>(s.gyr_0),
                                            Vector3d(0, 0, 0), Vector3d(0, 0, 0)};
    for (int k = 0; k < s.n_samples; k++)
      f.pre_integration->push_back(s.dt[k], Eigen::Map<const Vector3d>(s.acc + 3 * k), Eigen::Map<const Vector3d>(s.gyr + 3 * k));
    all.insert(make_pair(s.header, f));
  }
  std::vector<Vector3d> bgs(WINDOW_SIZE + 1);
  for (int i = 0; i <= WINDOW_SIZE; i++) bgs[i] = Eigen::Map<const Vector3d>(Bgs + 3 * i);
  Vector3d g;
  g.setZero();
  VectorXd x;
  bool r = VisualIMUAlignment(all, bgs.data(), g, x);
  for (int i = 0; i <= WINDOW_SIZE; i++)
    for (int k = 0; k < 3; k++) Bgs[3 * i + k] = bgs[i](k);
  for (int k = 0; k < 3; k++) g_out[k] = g(k);
  if (x.size() >= 3 * n_frames + 1) {
    for (int k = 0; k < 3 * n_frames; k++) x_out[k] = x(k);
    x_out[3 * n_frames] = x(x.size() - 1);
  }
  *ok = r ? 1 : 0;
  for (auto &kv : all) delete kv.second.pre_integration;
  return VIO_OK;
}
