// vio_amd_shim.hpp — source-level drop-in: the reference's two classes with their own member signatures, bodies over
// the C ABI of vio_amd.h (last row). Header-only; nothing here computes.
//
//   vio_shim::FeatureTracker<Traits>::readImage   VINS_ios/feature_tracker.hpp:59, feature_tracker.cpp:162-310
//   vio_shim::FeatureTracker<Traits>::solveVinsPnP VINS_ios/feature_tracker.hpp:58, feature_tracker.cpp:107-160
//   vio_shim::VINS<Traits>::processIMU            VINS_ios/VINS.hpp:164,            VINS.cpp:333-375
//   vio_shim::VINS<Traits>::processImage          VINS_ios/VINS.hpp:163,            VINS.cpp:377-478 (-> solve_ceres :480-831)
//
// The classes are templates over the few third-party types the reference's signatures mention, so the same text
// compiles inside the reference tree,
//     struct RefTraits { typedef cv::Mat Mat; typedef cv::Point2f Point2f; typedef Eigen::Vector2d Vector2d;
//                        typedef Eigen::Vector3d Vector3d; typedef Eigen::Matrix3d Matrix3d; };
// and, without OpenCV / Eigen, with any types that offer the same members: Mat { data, rows, cols, step },
// Point2f(float x, float y), Vector2d / Vector3d / Matrix3d with operator()(i) / operator()(i, j) and a default constructor
// (tests/shim_main.cpp). Public data members keep the reference's names (image_msg, img_cnt, Ps, Rs, ...).
#ifndef VIO_AMD_SHIM_HPP
#define VIO_AMD_SHIM_HPP

#include <list>
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
  typedef typename Traits::Vector2d Vector2d;
  typedef typename Traits::Vector3d Vector3d;
  typedef typename Traits::Matrix3d Matrix3d;
  struct IMU_MSG_LOCAL {  // feature_tracker.hpp:46-50
    double header;
    Vector3d acc, gyr;
  };
  struct IMG_MSG_LOCAL {  // vins_pnp.hpp:36-41
    int id;
    Vector2d observation;
    Vector3d position;
    int track_num;
  };
  struct VINS_RESULT {  // vins_pnp.hpp:27-34
    double header;
    Vector3d Ba, Bg, P;
    Matrix3d R;
    Vector3d V;
  };

  // FeatureTracker::FeatureTracker() (feature_tracker.cpp:13-16) + the compile-time constants as a VioConfig.
  // Without an extrinsic there is no vinsPnP member: solveVinsPnP then returns false like with vins_normal == false.
  explicit FeatureTracker(const VioConfig &cfg)
      : img_cnt(0), current_time(-1.0), use_pnp(false), update_finished(false), cfg_(cfg), fe_(nullptr), pnp_(nullptr), had_points_(false) {
    if (vio_frontend_create(&cfg_, 1, &fe_) != VIO_OK) throw std::runtime_error("vio_frontend_create failed (a gfx950 device is required)");
    obs_.resize(cfg_.max_corners);
  }
  // ... and with the camera-IMU extrinsic the app hands to vins_pnp.setExtrinsic / setIMUModel (ViewController.mm:318-319): the vinsPnP
  // member (PNP_SIZE = 6, global_param.hpp:30) exists and readImage runs solveVinsPnP when vins_normal is set.
  FeatureTracker(const VioConfig &cfg, const double tic[3], const double ric[9]) : FeatureTracker(cfg) {
    // (the delegated-to constructor has completed: the destructor runs for this throw and releases fe_)
    if (vio_pnp_tracker_create(&cfg_, 1, 6, tic, ric, &pnp_) != VIO_OK) throw std::runtime_error("vio_pnp_tracker_create failed");
  }
  ~FeatureTracker() {
    if (pnp_) vio_pnp_tracker_destroy(pnp_);
    vio_frontend_destroy(fe_);
  }
  FeatureTracker(const FeatureTracker &) = delete;
  FeatureTracker &operator=(const FeatureTracker &) = delete;

  // feature_tracker.hpp:58, feature_tracker.cpp:107-160: the landmarks the back-end has solved (solved_features, ascending
  // id) joined with the tracker's current points, setInit(solved_vins), the IMU samples since the last frame, then
  // vinsPnP::processImage(feature_msg, header, use_pnp); P / R = Ps / Rs[PNP_SIZE - 1].
  // The join uses the point list as it stands at :207 -- behind the first F-RANSAC, ahead of rejectWithF / setMask --
  // which the front-end keeps aside for this purpose (vio_frontend_get_pnp_points).
  bool solveVinsPnP(double header, Vector3d &P, Matrix3d &R, bool vins_normal) {
    if (!vins_normal || !pnp_) return false;
    const int cap = cfg_.max_corners;
    std::vector<float> pts(2 * (size_t)cap);
    std::vector<int32_t> ids(cap);
    int32_t n = 0;
    if (vio_frontend_get_pnp_points(fe_, 0, pts.data(), ids.data(), cap, &n) != VIO_OK) return false;
    std::vector<VioPnpFeature> solved, msg((size_t)cap + 1);
    for (typename std::list<IMG_MSG_LOCAL>::const_iterator it = solved_features.begin(); it != solved_features.end(); ++it) {
      VioPnpFeature f;
      f.id = it->id, f.track_num = it->track_num;
      f.observation[0] = f.observation[1] = 0.0;
      for (int k = 0; k < 3; k++) f.position[k] = it->position(k);
      solved.push_back(f);
    }
    int32_t n_msg = 0;
    if (vio_pnp_match_features(&cfg_, ids.data(), pts.data(), n, solved.empty() ? nullptr : solved.data(), (int32_t)solved.size(),
                               msg.data(), cap, &n_msg) != VIO_OK)
      return false;
    VioVinsResult r;
    r.header = solved_vins.header;
    for (int k = 0; k < 3; k++) {
      r.Ba[k] = solved_vins.Ba(k), r.Bg[k] = solved_vins.Bg(k), r.P[k] = solved_vins.P(k), r.V[k] = solved_vins.V(k);
      for (int c = 0; c < 3; c++) r.R[3 * k + c] = solved_vins.R(k, c);
    }
    if (vio_pnp_tracker_set_init(pnp_, 0, &r) != VIO_OK) return false;
    for (size_t i = 0; i < imu_msgs.size(); i++) {
      const double t = imu_msgs[i].header;
      if (current_time < 0) current_time = t;
      const double dt = t - current_time;
      current_time = t;
      const double a[3] = {imu_msgs[i].acc(0), imu_msgs[i].acc(1), imu_msgs[i].acc(2)};
      const double g[3] = {imu_msgs[i].gyr(0), imu_msgs[i].gyr(1), imu_msgs[i].gyr(2)};
      if (vio_pnp_tracker_process_imu(pnp_, 0, dt, a, g) != VIO_OK) return false;
    }
    double Pout[3], Rout[9];
    int32_t did_solve = 0;
    if (vio_pnp_tracker_process_images(pnp_, msg.data(), &n_msg, cap + 1, &header, use_pnp ? 1 : 0, nullptr, Pout, Rout, &did_solve) != VIO_OK)
      return false;
    for (int k = 0; k < 3; k++) {
      P(k) = Pout[k];
      for (int c = 0; c < 3; c++) R(k, c) = Rout[3 * k + c];
    }
    return true;
  }

  // feature_tracker.hpp:59. `_frame_cnt` is never read by the reference body; `result` aliases `_img`
  // (feature_tracker.cpp:165); P / R are written by solveVinsPnP (:207), which runs once the tracker holds points.
  void readImage(const Mat &_img, Mat &result, int _frame_cnt, std::vector<Point2f> &good_pts, std::vector<double> &track_len,
                 double header, Vector3d &P, Matrix3d &R, bool vins_normal) {
    (void)_frame_cnt;
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
    if (had_points_) solveVinsPnP(header, P, R, vins_normal);  // inside `if (cur_pts.size() > 0)` (:175-207)
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
    had_points_ = n > 0;
    update_finished = true;  // :309
  }

  int img_cnt;                             // feature_tracker.hpp:81 (advanced by the caller)
  double current_time;                     // :82
  bool use_pnp;                            // :84
  std::map<int, Vector3d> image_msg;       // :89
  bool update_finished;                    // :90
  std::list<IMG_MSG_LOCAL> solved_features;  // :91 (copied in by the caller before readImage, ViewController.mm:446-451, 734-755)
  VINS_RESULT solved_vins;                 // :92
  std::vector<IMU_MSG_LOCAL> imu_msgs;     // :93

 private:
  VioConfig cfg_;
  vio_frontend_t *fe_;
  vio_pnp_tracker_t *pnp_;
  bool had_points_;
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
