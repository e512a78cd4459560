// vio_amd_shim.hpp — source-level drop-in: the reference's two classes with their own member signatures, bodies over
// the C ABI of vio_amd.h (last row). Header-only; nothing here computes.
//
//   vio_shim::FeatureTracker<Traits>::readImage   VINS_ios/feature_tracker.hpp:59, feature_tracker.cpp:162-310
//   vio_shim::VINS<Traits>::processIMU            VINS_ios/VINS.hpp:164,            VINS.cpp:333-375
//   vio_shim::VINS<Traits>::processImage          VINS_ios/VINS.hpp:163,            VINS.cpp:377-478 (-> solve_ceres :480-831)
//
// The classes are templates over the few third-party types the reference's signatures mention, so the same text
// compiles inside the reference tree,
//     struct RefTraits { typedef cv::Mat Mat; typedef cv::Point2f Point2f;
//                        typedef Eigen::Vector3d Vector3d; typedef Eigen::Matrix3d Matrix3d; };
// and, without OpenCV / Eigen, with any types that offer the same members: Mat { data, rows, cols, step },
// Point2f(float x, float y), Vector3d / Matrix3d with operator()(i) / operator()(i, j) and a default constructor
// (tests/shim_main.cpp). Public data members keep the reference's names (image_msg, img_cnt, Ps, Rs, ...).
#ifndef VIO_AMD_SHIM_HPP
#define VIO_AMD_SHIM_HPP

#include <map>
#include <stdexcept>
#include <vector>

#include "vio_amd.h"

namespace vio_shim {

template <class Traits>
class FeatureTracker {
 public:
  typedef typename Traits::Mat Mat;
  typedef typename Traits::Point2f Point2f;
  typedef typename Traits::Vector3d Vector3d;
  typedef typename Traits::Matrix3d Matrix3d;

  // FeatureTracker::FeatureTracker() (feature_tracker.cpp:13-16) + the compile-time constants as a VioConfig
  explicit FeatureTracker(const VioConfig &cfg) : img_cnt(0), update_finished(false), cfg_(cfg), fe_(nullptr) {
    if (vio_frontend_create(&cfg_, 1, &fe_) != VIO_OK) throw std::runtime_error("vio_frontend_create failed (a gfx950 device is required)");
    obs_.resize(cfg_.max_corners);
  }
  ~FeatureTracker() { vio_frontend_destroy(fe_); }
  FeatureTracker(const FeatureTracker &) = delete;
  FeatureTracker &operator=(const FeatureTracker &) = delete;

  // feature_tracker.hpp:59. `_frame_cnt` is never read by the reference body; `result` aliases `_img`
  // (feature_tracker.cpp:165); P / R are written only when vins_normal and USE_PNP (default off, :107-160).
  void readImage(const Mat &_img, Mat &result, int _frame_cnt, std::vector<Point2f> &good_pts, std::vector<double> &track_len,
                 double header, Vector3d &P, Matrix3d &R, bool vins_normal) {
    (void)_frame_cnt, (void)P, (void)R, (void)vins_normal;
    result = _img;
    int n_obs = 0;
    VioTrackViz *viz = nullptr;
    const int publish = img_cnt == 0;  // the caller keeps img_cnt = (img_cnt + 1) % FREQ (ViewController.mm:494)
    int rc = vio_frontend_read_image(fe_, 0, _img.data, _img.rows, _img.cols, (int)_img.step, header, publish, obs_.data(), &n_obs, viz);
    good_pts.clear(), track_len.clear();
    if (rc != VIO_OK) {  // reference convention: no return code, failure = no points
      update_finished = true;
      return;
    }
    if (publish) {
      image_msg.clear();  // feature_tracker.cpp:290
      for (int i = 0; i < n_obs; i++) {
        Vector3d v;
        v(0) = obs_[i].x, v(1) = obs_[i].y, v(2) = obs_[i].z;  // :300-306
        image_msg[obs_[i].id] = v;
      }
    }
    // good_pts / track_len drive the UI overlay (:276-283): the tracked points and their track counts
    std::vector<float> pts(2 * cfg_.max_corners);
    std::vector<int32_t> ids(cfg_.max_corners), cnt(cfg_.max_corners);
    int32_t n = 0;
    if (vio_frontend_get_state(fe_, 0, pts.data(), ids.data(), cnt.data(), cfg_.max_corners, &n) == VIO_OK)
      for (int i = 0; i < n; i++) {
        good_pts.push_back(Point2f(pts[2 * i], pts[2 * i + 1]));
        track_len.push_back(cnt[i] > 20 ? 1.0 : cnt[i] / 20.0);  // std::min(1.0, 1.0 * track_cnt[i] / WINDOW_SIZE_FEATURE_TRACKER) :280
      }
    update_finished = true;  // :309
  }

  std::map<int, Vector3d> image_msg;  // feature_tracker.hpp:68
  int img_cnt;                        // :79 (advanced by the caller)
  bool update_finished;               // :67

 private:
  VioConfig cfg_;
  vio_frontend_t *fe_;
  std::vector<VioObs> obs_;
};

template <class Traits>
class VINS {
 public:
  typedef typename Traits::Vector3d Vector3d;
  typedef typename Traits::Matrix3d Matrix3d;
  enum SolverFlag { INITIAL = 0, NON_LINEAR = 1 };  // VINS.hpp:49-53

  // VINS::VINS() + setExtrinsic / setIMUModel (VINS.cpp:15-34, 267-300): the extrinsic and the IMU model come with cfg
  VINS(const VioConfig &cfg, const double tic[3], const double ric[9]) : solver_flag(INITIAL), frame_count(0), cfg_(cfg), est_(nullptr) {
    if (vio_estimator_create(&cfg_, 1, tic, ric, &est_) != VIO_OK) throw std::runtime_error("vio_estimator_create failed (a gfx950 device is required)");
    vio_estimator_enable_initialization(est_, 1);  // solveInitial runs inside processImage like in the reference
    Ps.resize(cfg_.window_size + 1), Rs.resize(cfg_.window_size + 1), Vs.resize(cfg_.window_size + 1);
    Bas.resize(cfg_.window_size + 1), Bgs.resize(cfg_.window_size + 1), Headers.resize(cfg_.window_size + 1);
  }
  ~VINS() { vio_estimator_destroy(est_); }
  VINS(const VINS &) = delete;
  VINS &operator=(const VINS &) = delete;

  // VINS.hpp:164
  void processIMU(double dt, const Vector3d &linear_acceleration, const Vector3d &angular_velocity) {
    const double a[3] = {linear_acceleration(0), linear_acceleration(1), linear_acceleration(2)};
    const double g[3] = {angular_velocity(0), angular_velocity(1), angular_velocity(2)};
    vio_estimator_process_imu(est_, 0, dt, a, g);
  }

  // VINS.hpp:163: everything down to solve_ceres(buf_num) and slideWindow; buf_num only selected a wall-clock budget
  // (VINS.cpp:648-653), there is none here.
  void processImage(std::map<int, Vector3d> &image_msg, double header, int buf_num) {
    (void)buf_num;
    obs_.clear();
    for (typename std::map<int, Vector3d>::const_iterator it = image_msg.begin(); it != image_msg.end(); ++it) {
      VioObs o;
      o.id = it->first, o.x = it->second(0), o.y = it->second(1), o.z = it->second(2);
      obs_.push_back(o);
    }
    vio_estimator_process_image(est_, 0, obs_.empty() ? nullptr : obs_.data(), (int)obs_.size(), header, &last_result);
    refresh();
  }

  // hand-over of the first window in place of solveInitial (what visualInitialAlign leaves, VINS.cpp:1081-1143)
  void setInitialState(const std::vector<double> &headers, const std::vector<Vector3d> &P, const std::vector<Matrix3d> &R,
                       const std::vector<Vector3d> &V, const Vector3d &ba, const Vector3d &bg) {
    const int n = cfg_.window_size + 1;
    std::vector<double> p(3 * n), r(9 * n), v(3 * n), a(3 * n), g(3 * n);
    for (int i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        p[3 * i + k] = P[i](k), v[3 * i + k] = V[i](k), a[3 * i + k] = ba(k), g[3 * i + k] = bg(k);
        for (int c = 0; c < 3; c++) r[9 * i + 3 * k + c] = R[i](k, c);
      }
    vio_estimator_set_initial_state(est_, 0, headers.data(), p.data(), r.data(), v.data(), a.data(), g.data());
  }

  // VINS.hpp:60-75 (sized WINDOW_SIZE + 1)
  std::vector<Vector3d> Ps, Vs, Bas, Bgs;
  std::vector<Matrix3d> Rs;
  std::vector<double> Headers;
  int solver_flag, frame_count;
  VioFrameResult last_result;   // what the last processImage did (action, marginalization flag, solve statistics)

 private:
  void refresh() {
    const int n = cfg_.window_size + 1;
    std::vector<double> p(3 * n), r(9 * n), v(3 * n), a(3 * n), g(3 * n), h(n);
    if (vio_estimator_get_window(est_, 0, p.data(), r.data(), v.data(), a.data(), g.data(), h.data()) != VIO_OK) return;
    for (int i = 0; i < n; i++) {
      Headers[i] = h[i];
      for (int k = 0; k < 3; k++) {
        Ps[i](k) = p[3 * i + k], Vs[i](k) = v[3 * i + k], Bas[i](k) = a[3 * i + k], Bgs[i](k) = g[3 * i + k];
        for (int c = 0; c < 3; c++) Rs[i](k, c) = r[9 * i + 3 * k + c];
      }
    }
    VioEstimatorStatus st;
    if (vio_estimator_get_status(est_, 0, &st) == VIO_OK) solver_flag = st.solver_flag, frame_count = st.frame_count;
  }
  VioConfig cfg_;
  vio_estimator_t *est_;
  std::vector<VioObs> obs_;
};

}  // namespace vio_shim

#endif
