// vio_posegraph_host.cpp — host bookkeeping around the pose graph solve, the parts of
// KeyFrameDatabase::optimize4DoFLoopPoseGraph (VINS_ios/loop/keyfame_database.cpp) that are list walking, not arithmetic:
//   vio_posegraph_build   :166-285  resampling flags (need_resample), parameter values, sequential and loop edges
//   vio_posegraph_apply   :303-339  poses after the solve; drift of the current keyframe
// The solve between the two is vio_posegraph_optimize (vio_posegraph.hip).
#include <math.h>
#include <string.h>

#include "vio_amd.h"

namespace {

const double kPi = 3.14159265358979323846;

struct Mat3 {
  double m[9];
  double operator()(int r, int c) const { return m[3 * r + c]; }
};

// Utility::R2ypr (VINS_ios/utility.hpp:76-91): degrees
void rot_to_ypr(const double *R, double *ypr) {
  const double y = atan2(R[3], R[0]);
  const double p = atan2(-R[6], R[0] * cos(y) + R[3] * sin(y));
  const double r = atan2(R[2] * sin(y) - R[5] * cos(y), -R[1] * sin(y) + R[4] * cos(y));
  ypr[0] = y / kPi * 180.0, ypr[1] = p / kPi * 180.0, ypr[2] = r / kPi * 180.0;
}

Mat3 mul(const Mat3 &A, const Mat3 &B) {
  Mat3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C.m[3 * i + j] = A(i, 0) * B(0, j) + A(i, 1) * B(1, j) + A(i, 2) * B(2, j);
  return C;
}

// Utility::ypr2R (utility.hpp:93-121): Rz(y) Ry(p) Rx(r), degrees in
Mat3 ypr_to_rot(const double *ypr) {
  const double y = ypr[0] / 180.0 * kPi, p = ypr[1] / 180.0 * kPi, r = ypr[2] / 180.0 * kPi;
  const Mat3 Rz = {{cos(y), -sin(y), 0, sin(y), cos(y), 0, 0, 0, 1}};
  const Mat3 Ry = {{cos(p), 0., sin(p), 0., 1., 0., -sin(p), 0., cos(p)}};
  const Mat3 Rx = {{1., 0., 0., 0., cos(r), -sin(r), 0., sin(r), cos(r)}};
  return mul(mul(Rz, Ry), Rx);
}

// R^T v
void rot_t_vec(const double *R, const double *v, double *out) {
  for (int k = 0; k < 3; k++) out[k] = R[k] * v[0] + R[3 + k] * v[1] + R[6 + k] * v[2];
}

}  // namespace

extern "C" {

int vio_posegraph_build(const VioPoseGraphKeyframe *kf, int32_t n_kf, double total_length, int32_t max_frame_num,
                        int32_t list_size, double *t, double *ypr, uint8_t *skip, int32_t cap_edges, int32_t *edge_i,
                        int32_t *edge_j, uint8_t *edge_kind, double *edge_meas, int32_t *n_edges) {
  if (!kf || n_kf < 1 || max_frame_num < 1 || !t || !ypr || !skip || !n_edges || cap_edges < 0 ||
      (cap_edges > 0 && (!edge_i || !edge_j || !edge_kind || !edge_meas)))
    return VIO_EINVAL;
  // need_resample (:170-198): a keyframe stays in the graph when it starts the graph, when enough path has accumulated
  // since the last kept one, when it takes part in a loop, or while the list is still short
  const double min_dis = total_length / (1.0 * max_frame_num);
  double travelled = 0, prev[3] = {0, 0, 0};
  for (int k = 0; k < n_kf; k++) {
    const double *p = kf[k].t;
    travelled += sqrt((p[0] - prev[0]) * (p[0] - prev[0]) + (p[1] - prev[1]) * (p[1] - prev[1]) + (p[2] - prev[2]) * (p[2] - prev[2]));
    const bool keep = k == 0 || travelled > min_dis || kf[k].has_loop || kf[k].is_looped || list_size < max_frame_num;
    if (keep) travelled = 0;
    skip[k] = keep ? 0 : 1;
    memcpy(prev, p, sizeof(prev));
  }
  int ne = 0;
  for (int i = 0; i < n_kf; i++) {
    memcpy(t + 3 * i, kf[i].origin_t, 3 * sizeof(double));
    rot_to_ypr(kf[i].origin_r, ypr + 3 * i);
    if (skip[i]) continue;
    // up to five kept predecessors (:232-262): relative translation in the predecessor's frame, yaw difference, the
    // predecessor's pitch and roll
    int linked = 0;
    for (int c = i - 1; c >= 0 && linked < 5; c--) {
      if (skip[c]) continue;
      linked++;
      if (ne >= cap_edges) return VIO_ECAP;
      const double d[3] = {t[3 * i] - t[3 * c], t[3 * i + 1] - t[3 * c + 1], t[3 * i + 2] - t[3 * c + 2]};
      double *m = edge_meas + 6 * ne;
      rot_t_vec(kf[c].origin_r, d, m);
      m[3] = ypr[3 * i] - ypr[3 * c], m[4] = ypr[3 * c + 1], m[5] = ypr[3 * c + 2];
      edge_i[ne] = c, edge_j[ne] = i, edge_kind[ne] = 0;
      ne++;
    }
    if (kf[i].has_loop) {  // (:264-285)
      int c = -1;
      for (int k = 0; k < n_kf && c < 0; k++)
        if (kf[k].global_index == kf[i].loop_index) c = k;
      if (c < 0) return VIO_EINVAL;  // loop_index before earliest_loop_index (the reference asserts)
      if (ne >= cap_edges) return VIO_ECAP;
      double yc[3];
      rot_to_ypr(kf[c].origin_r, yc);
      double *m = edge_meas + 6 * ne;
      m[0] = kf[i].loop_info[0], m[1] = kf[i].loop_info[1], m[2] = kf[i].loop_info[2], m[3] = kf[i].loop_info[7];
      m[4] = yc[1], m[5] = yc[2];
      edge_i[ne] = c, edge_j[ne] = i, edge_kind[ne] = 1;
      ne++;
    }
  }
  *n_edges = ne;
  return VIO_OK;
}

int vio_posegraph_apply(const VioPoseGraphKeyframe *kf, int32_t n_kf, const double *t, const double *ypr, const uint8_t *skip,
                        double *out_t, double *out_r, double *yaw_drift, double *r_drift, double *t_drift) {
  if (!kf || n_kf < 1 || !t || !ypr || !skip || !out_t || !out_r) return VIO_EINVAL;
  Mat3 drift_r = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  double drift_t[3] = {0, 0, 0};
  for (int i = 0; i < n_kf; i++) {
    const Mat3 R = ypr_to_rot(ypr + 3 * i);
    const double *p = t + 3 * i;
    if (skip[i]) {  // moved with the drift of the last kept keyframe (:316-319)
      const Mat3 Rn = mul(drift_r, R);
      memcpy(out_r + 9 * i, Rn.m, sizeof(Rn.m));
      for (int a = 0; a < 3; a++) out_t[3 * i + a] = drift_r(a, 0) * p[0] + drift_r(a, 1) * p[1] + drift_r(a, 2) * p[2] + drift_t[a];
    } else {        // takes the optimized pose and defines the drift from here on (:320-328)
      Mat3 Ro_t;    // origin_r^T
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) Ro_t.m[3 * a + b] = kf[i].origin_r[3 * b + a];
      drift_r = mul(R, Ro_t);
      const double *po = kf[i].origin_t;
      for (int a = 0; a < 3; a++) drift_t[a] = p[a] - (drift_r(a, 0) * po[0] + drift_r(a, 1) * po[1] + drift_r(a, 2) * po[2]);
      memcpy(out_r + 9 * i, R.m, sizeof(R.m));
      memcpy(out_t + 3 * i, p, 3 * sizeof(double));
    }
  }
  // yaw_drift / r_drift / t_drift of the current keyframe (:333-339)
  const int c = n_kf - 1;
  double cur[3], org[3];
  rot_to_ypr(out_r + 9 * c, cur), rot_to_ypr(kf[c].origin_r, org);
  const double yd[3] = {cur[0] - org[0], 0, 0};
  const Mat3 Rd = ypr_to_rot(yd);
  if (yaw_drift) *yaw_drift = yd[0];
  if (r_drift) memcpy(r_drift, Rd.m, sizeof(Rd.m));
  if (t_drift) {
    const double *po = kf[c].origin_t;
    for (int a = 0; a < 3; a++) t_drift[a] = out_t[3 * c + a] - (Rd(a, 0) * po[0] + Rd(a, 1) * po[1] + Rd(a, 2) * po[2]);
  }
  return VIO_OK;
}

}  // extern "C"
