// ref_fm_harness.cpp — TEST-ONLY driver of the REAL reference FeatureManager (VINS_ios/feature_manager.cpp compiled where
// it lies, see oracle/Makefile) for the window-bookkeeping parity tests. The only restated piece is the factor
// enumeration loop of VINS::solve_ceres (VINS.cpp:528-567), because VINS.cpp itself needs OpenCV headers.
#include <map>
#include <vector>

#include "feature_manager.hpp"
#include "vio_amd.h"

namespace {
struct RefFm {
  Matrix3d Rs[WINDOW_SIZE + 1];
  FeatureManager fm;
  RefFm() : fm(Rs) {
    for (auto &R : Rs) R.setIdentity();
    fm.clearState();
  }
};
Matrix3d rm(const double *M) {
  Matrix3d R;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R(i, j) = M[3 * i + j];
  return R;
}
}  // namespace

extern "C" {

int ref_fm_window_size() { return WINDOW_SIZE; }
void *ref_fm_create() { return new RefFm(); }
void ref_fm_destroy(void *h) { delete static_cast<RefFm *>(h); }

int ref_fm_add(void *h, int frame_count, const VioObs *obs, int n, int *parallax_num, int *last_track_num) {
  RefFm *r = static_cast<RefFm *>(h);
  std::map<int, Vector3d> msg;
  for (int i = 0; i < n; i++) msg[obs[i].id] = Vector3d(obs[i].x, obs[i].y, obs[i].z);
  int pn = 0;
  const size_t before = r->fm.feature.size();
  bool ret = r->fm.addFeatureCheckParallax(frame_count, msg, pn);
  // FeaturePerId's constructor leaves solve_flag uninitialised (feature_manager.hpp:63-66) and removeFailures reads
  // it (UB: a recycled list node can carry a stale 2). New features are appended at the back: give them the value
  // the comment at feature_manager.hpp:60 assigns to "haven't solved yet" so that the checker is deterministic.
  size_t idx = 0;
  for (auto &f : r->fm.feature)
    if (idx++ >= before) f.solve_flag = 0;
  *parallax_num = pn, *last_track_num = r->fm.last_track_num;
  return ret ? 1 : 0;
}

void ref_fm_triangulate(void *h, const double *Ps, const double *Rs, const double *tic, const double *ric) {
  RefFm *r = static_cast<RefFm *>(h);
  Vector3d P[WINDOW_SIZE + 1];
  for (int i = 0; i <= WINDOW_SIZE; i++) {
    r->Rs[i] = rm(Rs + 9 * i);
    P[i] = Vector3d(Ps[3 * i], Ps[3 * i + 1], Ps[3 * i + 2]);
  }
  r->fm.triangulate(P, Vector3d(tic[0], tic[1], tic[2]), rm(ric), true);
}

int ref_fm_count(void *h) { return static_cast<RefFm *>(h)->fm.getFeatureCount(); }
int ref_fm_get_depth(void *h, double *out) {
  VectorXd d = static_cast<RefFm *>(h)->fm.getDepthVector();
  for (int i = 0; i < d.size(); i++) out[i] = d(i);
  return (int)d.size();
}
void ref_fm_set_depth(void *h, const double *x, int n) {
  VectorXd d(n);
  for (int i = 0; i < n; i++) d(i) = x[i];
  static_cast<RefFm *>(h)->fm.setDepth(d);
}
void ref_fm_clear_depth(void *h, const double *x, int n) {
  VectorXd d(n);
  for (int i = 0; i < n; i++) d(i) = x[i];
  static_cast<RefFm *>(h)->fm.clearDepth(d);
}
void ref_fm_remove_failures(void *h) { static_cast<RefFm *>(h)->fm.removeFailures(); }
void ref_fm_remove_back(void *h) { static_cast<RefFm *>(h)->fm.removeBack(); }
void ref_fm_remove_back_shift_depth(void *h, const double *mR, const double *mP, const double *nR, const double *nP) {
  static_cast<RefFm *>(h)->fm.removeBackShiftDepth(rm(mR), Vector3d(mP[0], mP[1], mP[2]), rm(nR), Vector3d(nP[0], nP[1], nP[2]));
}
void ref_fm_remove_front(void *h, int frame_count) { static_cast<RefFm *>(h)->fm.removeFront(frame_count); }

// VINS.cpp:528-567 restated (the loop that turns the feature list into ProjectionFactor residual blocks)
int ref_fm_export(void *h, int cap, int *host, int *target, int *feature, double *pts_i, double *pts_j, int *n_features) {
  RefFm *r = static_cast<RefFm *>(h);
  int m = 0, feature_index = -1;
  for (auto &it_per_id : r->fm.feature) {
    it_per_id.used_num = it_per_id.feature_per_frame.size();
    if (!(it_per_id.used_num >= 2 && it_per_id.start_frame < WINDOW_SIZE - 2)) continue;
    ++feature_index;
    int imu_i = it_per_id.start_frame, imu_j = imu_i - 1;
    Vector3d pi = it_per_id.feature_per_frame[0].point;
    for (auto &it_per_frame : it_per_id.feature_per_frame) {
      imu_j++;
      if (imu_i == imu_j) continue;
      if (m >= cap) return -1;
      host[m] = imu_i, target[m] = imu_j, feature[m] = feature_index;
      for (int k = 0; k < 3; k++) pts_i[3 * m + k] = pi(k), pts_j[3 * m + k] = it_per_frame.point(k);
      m++;
    }
  }
  *n_features = feature_index + 1;
  return m;
}

int ref_fm_dump(void *h, VioFeatureInfo *info, int cap, double *points, int cap_points, int *n_points) {
  RefFm *r = static_cast<RefFm *>(h);
  int i = 0, p = 0;
  for (auto &f : r->fm.feature) {
    if (i >= cap) return -1;
    VioFeatureInfo &o = info[i++];
    o.id = f.feature_id, o.start_frame = f.start_frame, o.n_obs = (int)f.feature_per_frame.size(), o.used_num = f.used_num;
    o.solve_flag = f.solve_flag, o.is_outlier = f.is_outlier, o.fixed = f.fixed, o.estimated_depth = f.estimated_depth;
    if (points)
      for (auto &ob : f.feature_per_frame) {
        if (p >= cap_points) return -1;
        for (int k = 0; k < 3; k++) points[3 * p + k] = ob.point(k);
        p++;
      }
  }
  if (n_points) *n_points = p;
  return i;
}

}  // extern "C"
