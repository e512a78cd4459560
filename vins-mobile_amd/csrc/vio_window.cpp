// vio_window.cpp — window bookkeeping around the solve (host side): the feature store that decides which landmarks and
// observations become factors, keyframe selection by parallax, triangulation of new landmarks and the removal rules
// of the two marginalization modes.
//
// Reference: FeatureManager (VINS_ios/feature_manager.hpp:71-103, feature_manager.cpp:11-407), called from
// VINS::processImage / solve_ceres / slideWindow (VINS_ios/VINS.cpp:379-478, 528-567, 1149-1273). Like the reference
// this is host code next to the solver: lists of a few hundred landmarks, no device work. The window size is a
// run-time parameter here (WINDOW_SIZE is a compile-time 10 in global_param.hpp:28).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <list>
#include <vector>

#include "vio_amd.h"
#include "vio_math.h"

using namespace vio;

namespace {

struct Obs {        // FeaturePerFrame (feature_manager.hpp:32-45)
  double point[3];  // _point / z
  double z;
};

struct Feature {  // FeaturePerId (feature_manager.hpp:47-69)
  int feature_id, start_frame;
  std::vector<Obs> obs;  // feature_per_frame
  int used_num = 0;
  bool is_outlier = false, fixed = false;
  double estimated_depth = -1.0;
  int solve_flag = 0;  // (left uninitialised by the reference's constructor; 0 = "haven't solved yet" is its stated meaning)
  Feature(int id, int start) : feature_id(id), start_frame(start) {}
  int end_frame() const { return (int)(start_frame + obs.size() - 1); }
};

constexpr double kMinParallax = 10.0 / 549;  // MIN_PARALLAX (feature_manager.hpp:23)
constexpr double kInitDepth = 5.0;           // INIT_DEPTH (feature_manager.hpp:24)

// Right singular vector of the smallest singular value of A (rows x 4): one-sided Jacobi (Hestenes) on the columns,
// the same family as the Eigen::JacobiSVD the reference calls (feature_manager.cpp:240).
void smallest_right_singular_vector(std::vector<double> &A, int rows, double v_out[4]) {
  double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < rows; i++) {
          const double ap = A[i * 4 + p], aq = A[i * 4 + q];
          alpha += ap * ap, beta += aq * aq, gamma += ap * aq;
        }
        if (gamma == 0.0) continue;
        off = std::max(off, fabs(gamma) / sqrt(alpha * beta + 1e-300));
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < rows; i++) {
          const double ap = A[i * 4 + p], aq = A[i * 4 + q];
          A[i * 4 + p] = c * ap - s * aq, A[i * 4 + q] = s * ap + c * aq;
        }
        for (int i = 0; i < 4; i++) {
          const double vp = V[i * 4 + p], vq = V[i * 4 + q];
          V[i * 4 + p] = c * vp - s * vq, V[i * 4 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  int best = 0;
  double bn = 1e300;
  for (int j = 0; j < 4; j++) {
    double nrm = 0;
    for (int i = 0; i < rows; i++) nrm += A[i * 4 + j] * A[i * 4 + j];
    if (nrm < bn) bn = nrm, best = j;
  }
  for (int i = 0; i < 4; i++) v_out[i] = V[i * 4 + best];
}

}  // namespace

struct vio_features {
  int window_size = 10;
  std::list<Feature> feature;
  int last_track_num = 0;
  // the predicate that defines feature_index <-> para_Feature row (feature_manager.cpp:181,199,274,290,305; VINS.cpp:531)
  bool solved_in_window(const Feature &f) const { return f.used_num >= 2 && f.start_frame < window_size - 2; }
};

extern "C" {

int vio_features_create(int32_t window_size, vio_features_t **out) {
  if (!out || window_size < 2) return VIO_EINVAL;
  vio_features *fm = new vio_features();
  fm->window_size = window_size;
  *out = fm;
  return VIO_OK;
}

void vio_features_destroy(vio_features_t *fm) { delete fm; }

int vio_features_clear(vio_features_t *fm) {  // clearState (feature_manager.cpp:315-318)
  if (!fm) return VIO_EINVAL;
  fm->feature.clear();
  return VIO_OK;
}

// addFeatureCheckParallax (feature_manager.cpp:103-155). image_msg is a std::map: observations are taken in ascending id.
int vio_features_add_check_parallax(vio_features_t *fm, int32_t frame_count, const VioObs *obs, int32_t n_obs,
                                    int32_t *enough_parallax, int32_t *parallax_num, int32_t *last_track_num) {
  if (!fm || n_obs < 0 || (n_obs > 0 && !obs) || !enough_parallax) return VIO_EINVAL;
  std::vector<int> order(n_obs);
  for (int i = 0; i < n_obs; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return obs[a].id < obs[b].id; });
  for (int i = 1; i < n_obs; i++)
    if (obs[order[i]].id == obs[order[i - 1]].id) return VIO_EINVAL;  // a map has unique keys
  // every index the other calls derive from a landmark (start_frame + observation number) must stay inside the window:
  // a frame index outside [0, W] or a second message for a frame a landmark already has is a call-order error
  if (frame_count < 0 || frame_count > fm->window_size) return VIO_EINVAL;
  {
    std::vector<int> ids(n_obs);
    for (int i = 0; i < n_obs; i++) ids[i] = obs[order[i]].id;  // ascending
    for (const Feature &x : fm->feature)
      if (x.start_frame + (int)x.obs.size() > frame_count && std::binary_search(ids.begin(), ids.end(), x.feature_id))
        return VIO_ESTATE;
  }
  double parallax_sum = 0;
  int pnum = 0;
  fm->last_track_num = 0;
  for (int oi = 0; oi < n_obs; oi++) {
    const VioObs &o = obs[order[oi]];
    Obs f;
    f.z = o.z;
    f.point[0] = o.x / o.z, f.point[1] = o.y / o.z, f.point[2] = o.z / o.z;
    auto it = std::find_if(fm->feature.begin(), fm->feature.end(), [&](const Feature &x) { return x.feature_id == o.id; });
    if (it == fm->feature.end()) {
      fm->feature.emplace_back(o.id, frame_count);
      fm->feature.back().obs.push_back(f);
    } else {
      it->obs.push_back(f);
      fm->last_track_num++;
    }
  }
  if (last_track_num) *last_track_num = fm->last_track_num;
  if (parallax_num) *parallax_num = 0;
  if (frame_count < 2 || fm->last_track_num < 20) {
    *enough_parallax = 1;
    return VIO_OK;
  }
  for (const Feature &f : fm->feature) {
    if (f.start_frame <= frame_count - 2 && f.start_frame + (int)f.obs.size() - 1 >= frame_count - 1) {
      // compensatedParallax2 (feature_manager.cpp:64-95): the rotation compensation is commented out in the reference,
      // p_i_comp == p_i, so both candidates of the min() are the same number
      const Obs &fi = f.obs[frame_count - 2 - f.start_frame], &fj = f.obs[frame_count - 1 - f.start_frame];
      const double u_j = fj.point[0], v_j = fj.point[1];
      const double dep_i = fi.point[2];
      const double u_i = fi.point[0] / dep_i, v_i = fi.point[1] / dep_i;
      const double du = u_i - u_j, dv = v_i - v_j;
      const double du_comp = u_i - u_j, dv_comp = v_i - v_j;
      parallax_sum += std::max(0.0, sqrt(std::min(du * du + dv * dv, du_comp * du_comp + dv_comp * dv_comp)));
      pnum++;
    }
  }
  if (parallax_num) *parallax_num = pnum;
  *enough_parallax = pnum == 0 ? 1 : (parallax_sum / pnum >= kMinParallax ? 1 : 0);
  return VIO_OK;
}

int vio_features_count(vio_features_t *fm, int32_t *n) {  // getFeatureCount (feature_manager.cpp:284-298)
  if (!fm || !n) return VIO_EINVAL;
  int sum = 0;
  for (Feature &f : fm->feature) {
    f.used_num = (int)f.obs.size();
    if (fm->solved_in_window(f)) sum++;
  }
  *n = sum;
  return VIO_OK;
}

int vio_features_get_depth_vector(vio_features_t *fm, double *inv_depth, int32_t cap, int32_t *n) {  // :270-282
  if (!fm || !n || (cap > 0 && !inv_depth)) return VIO_EINVAL;
  int idx = 0;
  for (Feature &f : fm->feature) {
    f.used_num = (int)f.obs.size();
    if (!fm->solved_in_window(f)) continue;
    if (idx >= cap) return VIO_ECAP;
    inv_depth[idx++] = 1. / f.estimated_depth;
  }
  *n = idx;
  return VIO_OK;
}

static int assign_depths(vio_features *fm, const double *x, int n, bool flags) {  // setDepth :300-313 / clearDepth :176-187
  int idx = 0;
  for (Feature &f : fm->feature) {
    f.used_num = (int)f.obs.size();
    if (!fm->solved_in_window(f)) continue;
    if (idx >= n) return VIO_ECAP;
    f.estimated_depth = 1.0 / x[idx++];
    if (flags) f.solve_flag = f.estimated_depth < 0 ? 2 : 1;
  }
  return idx == n ? VIO_OK : VIO_EINVAL;
}
int vio_features_set_depth(vio_features_t *fm, const double *inv_depth, int32_t n) {
  if (!fm || n < 0 || (n > 0 && !inv_depth)) return VIO_EINVAL;
  return assign_depths(fm, inv_depth, n, true);
}
int vio_features_clear_depth(vio_features_t *fm, const double *inv_depth, int32_t n) {
  if (!fm || n < 0 || (n > 0 && !inv_depth)) return VIO_EINVAL;
  return assign_depths(fm, inv_depth, n, false);
}

// triangulate (feature_manager.cpp:189-248). Ps [W+1][3], Rs [W+1][9] row-major (body -> world), tic, ric (camera -> body).
int vio_features_triangulate(vio_features_t *fm, const double *Ps, const double *Rs, const double tic[3], const double ric[9]) {
  if (!fm || !Ps || !Rs || !tic || !ric) return VIO_EINVAL;
  const int W = fm->window_size;
  for (Feature &f : fm->feature) {
    if ((int)f.obs.size() >= W) f.fixed = true;
    f.used_num = (int)f.obs.size();
    if (!fm->solved_in_window(f)) continue;
    if (f.estimated_depth > 0) continue;
    f.is_outlier = false;
    const int imu_i = f.start_frame;
    if (imu_i + (int)f.obs.size() - 1 > W) return VIO_ESTATE;  // observation beyond the window: caller forgot to slide
    int imu_j = imu_i - 1;
    std::vector<double> A(2 * f.obs.size() * 4);
    double t0[3], R0[9], tmp[3];
    mat3vec(Rs + 9 * imu_i, tic, tmp);
    for (int k = 0; k < 3; k++) t0[k] = Ps[3 * imu_i + k] + tmp[k];
    mat3mul(Rs + 9 * imu_i, ric, R0);
    int row = 0;
    for (const Obs &o : f.obs) {
      imu_j++;
      double t1[3], R1[9], R0T[9], d[3], t[3], R[9], RT[9], mt[3];
      mat3vec(Rs + 9 * imu_j, tic, tmp);
      for (int k = 0; k < 3; k++) t1[k] = Ps[3 * imu_j + k] + tmp[k];
      mat3mul(Rs + 9 * imu_j, ric, R1);
      mat3T(R0, R0T);
      for (int k = 0; k < 3; k++) d[k] = t1[k] - t0[k];
      mat3vec(R0T, d, t);
      mat3mul(R0T, R1, R);
      mat3T(R, RT);
      mat3vec(RT, t, mt);
      double P[12];  // [R^T | -R^T t]
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) P[i * 4 + j] = RT[i * 3 + j];
        P[i * 4 + 3] = -mt[i];
      }
      const double nrm = sqrt(o.point[0] * o.point[0] + o.point[1] * o.point[1] + o.point[2] * o.point[2]);
      const double fx = o.point[0] / nrm, fy = o.point[1] / nrm, fz = o.point[2] / nrm;
      for (int j = 0; j < 4; j++) A[row * 4 + j] = fx * P[2 * 4 + j] - fz * P[0 * 4 + j];
      row++;
      for (int j = 0; j < 4; j++) A[row * 4 + j] = fy * P[2 * 4 + j] - fz * P[1 * 4 + j];
      row++;
    }
    double v[4];
    smallest_right_singular_vector(A, row, v);
    f.estimated_depth = v[2] / v[3];
    if (f.estimated_depth < 0.1) {
      f.is_outlier = true;
      f.estimated_depth = kInitDepth;
    }
  }
  return VIO_OK;
}

int vio_features_remove_failures(vio_features_t *fm) {  // :259-268
  if (!fm) return VIO_EINVAL;
  fm->feature.remove_if([](const Feature &f) { return f.solve_flag == 2; });
  return VIO_OK;
}

int vio_features_remove_back(vio_features_t *fm) {  // :320-341 (used while the estimator is not initialised yet)
  if (!fm) return VIO_EINVAL;
  for (auto it = fm->feature.begin(); it != fm->feature.end();) {
    auto cur = it++;
    if (cur->start_frame != 0) {
      cur->start_frame--;
    } else {
      cur->obs.erase(cur->obs.begin());
      if (cur->obs.empty()) fm->feature.erase(cur);
    }
  }
  return VIO_OK;
}

// removeBackShiftDepth (:250-257...): the oldest frame leaves; landmarks hosted there move their depth to the new host.
int vio_features_remove_back_shift_depth(vio_features_t *fm, const double marg_R[9], const double marg_P[3],
                                         const double new_R[9], const double new_P[3]) {
  if (!fm || !marg_R || !marg_P || !new_R || !new_P) return VIO_EINVAL;
  for (auto it = fm->feature.begin(); it != fm->feature.end();) {
    auto cur = it++;
    if (cur->start_frame != 0) {
      cur->start_frame--;
      continue;
    }
    const double uv[3] = {cur->obs[0].point[0], cur->obs[0].point[1], cur->obs[0].point[2]};
    cur->obs.erase(cur->obs.begin());
    if (cur->obs.size() < 2) {
      fm->feature.erase(cur);
      continue;
    }
    double pts_i[3], w[3], nRT[9], d[3], pts_j[3];
    for (int k = 0; k < 3; k++) pts_i[k] = uv[k] * cur->estimated_depth;
    mat3vec(marg_R, pts_i, w);
    for (int k = 0; k < 3; k++) d[k] = w[k] + marg_P[k] - new_P[k];
    mat3T(new_R, nRT);
    mat3vec(nRT, d, pts_j);
    cur->estimated_depth = pts_j[2] > 0 ? pts_j[2] : kInitDepth;
  }
  return VIO_OK;
}

int vio_features_remove_front(vio_features_t *fm, int32_t frame_count) {  // :343-372 (MARGIN_SECOND_NEW)
  if (!fm) return VIO_EINVAL;
  for (auto it = fm->feature.begin(); it != fm->feature.end();) {
    auto cur = it++;
    if (cur->start_frame == frame_count) {
      cur->start_frame--;
    } else {
      const int j = fm->window_size - 1 - cur->start_frame;
      if (cur->end_frame() < frame_count - 1) continue;
      if (j < 0 || j >= (int)cur->obs.size()) return VIO_ESTATE;
      cur->obs.erase(cur->obs.begin() + j);
      if (cur->obs.empty()) fm->feature.erase(cur);
    }
  }
  return VIO_OK;
}

// The factor enumeration of solve_ceres (VINS.cpp:528-567): landmark f of the depth vector, hosted at its start frame,
// one factor per later observation. Fills the arrays a VioWindow points to. With a relocalization frame (loop_frame >= 0)
// the landmarks seen in that frame and matched in the old keyframe get one more factor whose target is the loop pose
// (index W+1), VINS.cpp:597-631; it closes the landmark's group (the device solve wants factors grouped by landmark, the
// order inside a group is the reference's: window factors first).
static int export_factors(vio_features_t *fm, int32_t cap_factors, int32_t loop_frame, const int32_t *loop_ids,
                          const double *loop_xy, int32_t n_loop, int32_t *host, int32_t *target, int32_t *feature,
                          double *pts_i, double *pts_j, int32_t *n_factors, int32_t *n_features, int32_t *n_loop_factors) {
  int m = 0, fi = -1, r = 0, nl = 0;
  for (Feature &f : fm->feature) {
    f.used_num = (int)f.obs.size();
    if (!fm->solved_in_window(f)) continue;
    ++fi;
    const int imu_i = f.start_frame;
    int imu_j = imu_i - 1;
    for (const Obs &o : f.obs) {
      imu_j++;
      if (imu_i == imu_j) continue;
      if (m >= cap_factors) return VIO_ECAP;
      host[m] = imu_i, target[m] = imu_j, feature[m] = fi;
      for (int k = 0; k < 3; k++) pts_i[3 * m + k] = f.obs[0].point[k], pts_j[3 * m + k] = o.point[k];
      m++;
    }
    if (loop_frame >= 0 && f.start_frame <= loop_frame && f.end_frame() >= loop_frame) {
      while (r < n_loop && loop_ids[r] < f.feature_id) r++;  // (the reference walks its id list without the bound)
      if (r < n_loop && loop_ids[r] == f.feature_id) {
        if (m >= cap_factors) return VIO_ECAP;
        host[m] = imu_i, target[m] = fm->window_size + 1, feature[m] = fi;
        for (int k = 0; k < 3; k++) pts_i[3 * m + k] = f.obs[0].point[k];
        pts_j[3 * m] = loop_xy[2 * r], pts_j[3 * m + 1] = loop_xy[2 * r + 1], pts_j[3 * m + 2] = 1.0;
        m++, r++, nl++;
      }
    }
  }
  *n_factors = m, *n_features = fi + 1;
  if (n_loop_factors) *n_loop_factors = nl;
  return VIO_OK;
}

int vio_features_export_factors(vio_features_t *fm, int32_t cap_factors, int32_t *host, int32_t *target, int32_t *feature,
                                double *pts_i, double *pts_j, int32_t *n_factors, int32_t *n_features) {
  if (!fm || !n_factors || !n_features || (cap_factors > 0 && (!host || !target || !feature || !pts_i || !pts_j)))
    return VIO_EINVAL;
  return export_factors(fm, cap_factors, -1, nullptr, nullptr, 0, host, target, feature, pts_i, pts_j, n_factors, n_features,
                        nullptr);
}

int vio_features_export_factors_loop(vio_features_t *fm, int32_t cap_factors, int32_t loop_frame, const int32_t *loop_ids,
                                     const double *loop_xy, int32_t n_loop, int32_t *host, int32_t *target, int32_t *feature,
                                     double *pts_i, double *pts_j, int32_t *n_factors, int32_t *n_features,
                                     int32_t *n_loop_factors) {
  if (!fm || !n_factors || !n_features || (cap_factors > 0 && (!host || !target || !feature || !pts_i || !pts_j)))
    return VIO_EINVAL;
  if (n_loop < 0 || (n_loop > 0 && (!loop_ids || !loop_xy)) || loop_frame >= fm->window_size) return VIO_EINVAL;
  return export_factors(fm, cap_factors, loop_frame, loop_ids, loop_xy, n_loop, host, target, feature, pts_i, pts_j,
                        n_factors, n_features, n_loop_factors);
}

// visualInitialAlign rescales the landmarks that take part in the solve once the metric scale is known
// (VINS.cpp:1079-1085).
int vio_features_scale_depth(vio_features_t *fm, double s) {
  if (!fm) return VIO_EINVAL;
  for (Feature &f : fm->feature) {
    f.used_num = (int)f.obs.size();
    if (!fm->solved_in_window(f)) continue;
    f.estimated_depth *= s;
  }
  return VIO_OK;
}

// failureDetection (VINS.cpp:214-265): the checks on the newest frame after a solve. Returns a bit mask (0 = healthy).
int vio_failure_detection(int32_t last_track_num, const double Bg_newest[3], const double P_newest[3],
                          const double R_newest[9], const double last_P[3], const double last_R[9], int32_t *reasons) {
  if (!Bg_newest || !P_newest || !R_newest || !last_P || !last_R || !reasons) return VIO_EINVAL;
  int r = 0;
  if (last_track_num < 4) r |= VIO_FAIL_FEW_FEATURES;
  if (sqrt(Bg_newest[0] * Bg_newest[0] + Bg_newest[1] * Bg_newest[1] + Bg_newest[2] * Bg_newest[2]) > 1) r |= VIO_FAIL_GYR_BIAS;
  const double d[3] = {P_newest[0] - last_P[0], P_newest[1] - last_P[1], P_newest[2] - last_P[2]};
  if (sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > 1) r |= VIO_FAIL_TRANSLATION;
  if (fabs(P_newest[2] - last_P[2]) > 0.5) r |= VIO_FAIL_Z_TRANSLATION;
  double RT[9], dR[9];
  mat3T(R_newest, RT);
  mat3mul(RT, last_R, dR);  // tmp_R^T last_R
  const Quat dq = RtoQ(dR);  // Eigen's matrix -> quaternion (w may come out negative on the trace <= 0 branch)
  const double delta_angle = acos(dq.w) * 2.0 / 3.14 * 180.0;  // (3.14, as written in the reference)
  if (delta_angle > 40) r |= VIO_FAIL_ROTATION;
  *reasons = r;
  return VIO_OK;
}

int vio_features_load(vio_features_t *fm, const VioFeatureInfo *info, int32_t n, const double *points) {
  if (!fm || n < 0 || (n > 0 && (!info || !points))) return VIO_EINVAL;
  for (int i = 0; i < n; i++)
    if (info[i].n_obs < 1 || info[i].start_frame < 0 || info[i].start_frame + info[i].n_obs - 1 > fm->window_size) return VIO_EINVAL;
  fm->feature.clear();
  size_t p = 0;
  for (int i = 0; i < n; i++) {
    fm->feature.emplace_back(info[i].id, info[i].start_frame);
    Feature &f = fm->feature.back();
    f.obs.resize(info[i].n_obs);
    for (int j = 0; j < info[i].n_obs; j++, p++) {
      for (int k = 0; k < 3; k++) f.obs[j].point[k] = points[3 * p + k];
      f.obs[j].z = points[3 * p + 2];
    }
    f.used_num = info[i].used_num, f.solve_flag = info[i].solve_flag, f.is_outlier = info[i].is_outlier != 0, f.fixed = info[i].fixed != 0;
    f.estimated_depth = info[i].estimated_depth;
  }
  return VIO_OK;
}

int vio_features_dump(vio_features_t *fm, VioFeatureInfo *info, int32_t cap, int32_t *n, double *points, int32_t cap_points,
                      int32_t *n_points) {
  if (!fm || !n || (cap > 0 && !info)) return VIO_EINVAL;
  int i = 0, p = 0;
  if (cap == 0) {  // count only
    for (const Feature &f : fm->feature) i++, p += (int)f.obs.size();
    *n = i;
    if (n_points) *n_points = p;
    return VIO_OK;
  }
  for (const Feature &f : fm->feature) {
    if (i >= cap) return VIO_ECAP;
    VioFeatureInfo &o = info[i++];
    o.id = f.feature_id, o.start_frame = f.start_frame, o.n_obs = (int)f.obs.size(), o.used_num = f.used_num;
    o.solve_flag = f.solve_flag, o.is_outlier = f.is_outlier, o.fixed = f.fixed, o.estimated_depth = f.estimated_depth;
    if (points)
      for (const Obs &ob : f.obs) {
        if (p >= cap_points) return VIO_ECAP;
        for (int k = 0; k < 3; k++) points[3 * p + k] = ob.point[k];
        p++;
      }
  }
  *n = i;
  if (n_points) *n_points = p;
  return VIO_OK;
}

}  // extern "C"
