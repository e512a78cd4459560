// vio_pnp_tracker.cpp — the state machine around the motion-only window solve (host side), for n sequences sharing one
// device launch.
//
// Reference: class vinsPnP (VINS_ios/vins_pnp.hpp:45-91, vins_pnp.cpp:16-382): setInit (:53-72), updateFeatures (:74-92),
// old2new/new2old (:94-172), processIMU (:174-215), processImage (:217-238), slideWindow (:343-382), as driven from
// FeatureTracker::solveVinsPnP (feature_tracker.cpp:107-160) inside readImage. The window solve itself is
// vio_pnp_solve_windows (pnp_core.h).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "vio_amd.h"
#include "vio_math.h"
#include "vio_preint.h"

using namespace vio;

namespace {

struct Seq {
  int n = 0;  // PNP_SIZE
  int frame_count = 0;
  bool first_imu = false;
  double acc_0[3] = {0, 0, 0}, gyr_0[3] = {0, 0, 0};
  std::vector<double> Ps, Rs, Vs, Bas, Bgs, Headers;  // [n+1]
  std::vector<host::Preint> pre;
  std::vector<char> pre_valid, find_solved;
  std::vector<std::vector<VioPnpFeature>> features;
  // scratch of the window being solved
  std::vector<double> pose, speed, bias, obs, pos;
  std::vector<uint8_t> fixed;
  std::vector<int32_t> feat_start, track;
  std::vector<VioPreintegration> preint;
};

const double kI3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

}  // namespace

struct vio_pnp_tracker {
  VioConfig cfg;
  int n_seq = 0, size = 6;  // PNP_SIZE (global_param.hpp:29)
  double ex_pose[7];
  std::vector<Seq> seq;
  vio_pnp_t *solver = nullptr;  // created at the first solve
  std::vector<VioPnpWindow> windows;
  std::vector<int> solving;
};

namespace {

void clear_state(vio_pnp_tracker *t, Seq &s) {  // vinsPnP::clearState (:16-43)
  const int P = t->size + 1;
  s.n = t->size;
  s.Ps.assign(3 * P, 0), s.Vs.assign(3 * P, 0), s.Bas.assign(3 * P, 0), s.Bgs.assign(3 * P, 0), s.Headers.assign(P, 0);
  s.Rs.assign(9 * P, 0);
  for (int i = 0; i < P; i++) memcpy(&s.Rs[9 * i], kI3, sizeof(kI3));
  s.pre.resize(P), s.pre_valid.assign(P, 0), s.find_solved.assign(P, 0);
  s.features.assign(P, {});
  s.frame_count = 0, s.first_imu = false;
}

void new_pre(vio_pnp_tracker *t, Seq &s, int i) {
  host::preint_init(s.pre[i], &t->cfg, s.acc_0, s.gyr_0, &s.Bas[3 * i], &s.Bgs[3 * i]);
  s.pre_valid[i] = 1;
}

void slide_window(vio_pnp_tracker *t, Seq &s) {  // vinsPnP::slideWindow (:343-382)
  const int N = t->size;
  if (s.frame_count != N) return;
  for (int i = 0; i < N; i++) {
    for (int k = 0; k < 9; k++) std::swap(s.Rs[9 * i + k], s.Rs[9 * (i + 1) + k]);
    std::swap(s.pre[i], s.pre[i + 1]), std::swap(s.pre_valid[i], s.pre_valid[i + 1]);
    s.Headers[i] = s.Headers[i + 1];
    for (int k = 0; k < 3; k++) std::swap(s.Ps[3 * i + k], s.Ps[3 * (i + 1) + k]), std::swap(s.Vs[3 * i + k], s.Vs[3 * (i + 1) + k]);
    s.features[i].swap(s.features[i + 1]);
    s.find_solved[i] = s.find_solved[i + 1];
  }
  s.Headers[N] = s.Headers[N - 1];
  memcpy(&s.Rs[9 * N], &s.Rs[9 * (N - 1)], 72);
  for (int k = 0; k < 3; k++) {
    s.Ps[3 * N + k] = s.Ps[3 * (N - 1) + k], s.Vs[3 * N + k] = s.Vs[3 * (N - 1) + k];
    s.Bas[3 * N + k] = s.Bas[3 * (N - 1) + k], s.Bgs[3 * N + k] = s.Bgs[3 * (N - 1) + k];
  }
  s.find_solved[N] = 0;
  new_pre(t, s, N);
  s.features[N].clear();
}

// updateFeatures (:74-92): the newest message refreshes position / track_num of the same landmarks in the older frames.
// Both lists are ascending in id; the reference walks the older list without a bound, this one stops at its end.
void update_features(Seq &s, const std::vector<VioPnpFeature> &msg) {
  for (int i = 0; i < s.frame_count; i++) {
    std::vector<VioPnpFeature> &old = s.features[i];
    size_t j = 0;
    for (const VioPnpFeature &it : msg) {
      while (j < old.size() && old[j].id < it.id) j++;
      if (j < old.size() && old[j].id == it.id) {
        memcpy(old[j].position, it.position, sizeof(it.position));
        old[j].track_num = it.track_num;
      }
    }
  }
}

void build_window(vio_pnp_tracker *t, Seq &s, VioPnpWindow *w) {  // old2new (:94-135) + the factor lists of solve_ceres
  const int P = t->size + 1;
  s.pose.resize(7 * P), s.speed.resize(3 * P), s.bias.resize(6 * P), s.fixed.resize(P), s.feat_start.assign(P + 1, 0);
  s.preint.resize(P - 1);
  s.obs.clear(), s.pos.clear(), s.track.clear();
  for (int i = 0; i < P; i++) {
    const Quat q = RtoQ(&s.Rs[9 * i]);
    double *p = &s.pose[7 * i];
    p[0] = s.Ps[3 * i], p[1] = s.Ps[3 * i + 1], p[2] = s.Ps[3 * i + 2], p[3] = q.x, p[4] = q.y, p[5] = q.z, p[6] = q.w;
    for (int k = 0; k < 3; k++) s.speed[3 * i + k] = s.Vs[3 * i + k], s.bias[6 * i + k] = s.Bas[3 * i + k], s.bias[6 * i + 3 + k] = s.Bgs[3 * i + k];
    s.fixed[i] = s.find_solved[i] ? 1 : 0;
    for (const VioPnpFeature &f : s.features[i]) {
      s.obs.push_back(f.observation[0]), s.obs.push_back(f.observation[1]);
      s.pos.insert(s.pos.end(), f.position, f.position + 3);
      s.track.push_back(f.track_num);
    }
    s.feat_start[i + 1] = (int32_t)s.track.size();
    if (i > 0) host::preint_export(s.pre[i], &s.preint[i - 1]);
  }
  memset(w, 0, sizeof(*w));
  w->n_frames = P, w->pose = s.pose.data(), w->speed = s.speed.data(), w->bias = s.bias.data(), w->fixed = s.fixed.data();
  w->ex_pose = t->ex_pose, w->preint = s.preint.data(), w->feat_start = s.feat_start.data();
  w->observation = s.obs.data(), w->position = s.pos.data(), w->track_num = s.track.data();
}

}  // namespace

extern "C" {

int vio_pnp_tracker_create(const VioConfig *cfg, int32_t n_seq, int32_t pnp_size, const double tic[3], const double ric[9],
                           vio_pnp_tracker_t **out) {
  if (!cfg || !out || n_seq < 1 || !tic || !ric || pnp_size < 1 || pnp_size + 1 > VIO_PNP_MAX_FRAMES) return VIO_EINVAL;
  vio_pnp_tracker *t = new (std::nothrow) vio_pnp_tracker();
  if (!t) return VIO_ENOMEM;
  t->cfg = *cfg, t->n_seq = n_seq, t->size = pnp_size;
  const Quat q = RtoQ(ric);
  t->ex_pose[0] = tic[0], t->ex_pose[1] = tic[1], t->ex_pose[2] = tic[2];
  t->ex_pose[3] = q.x, t->ex_pose[4] = q.y, t->ex_pose[5] = q.z, t->ex_pose[6] = q.w;
  t->seq.resize(n_seq);
  for (Seq &s : t->seq) clear_state(t, s);
  t->windows.resize(n_seq);
  *out = t;
  return VIO_OK;
}

void vio_pnp_tracker_destroy(vio_pnp_tracker_t *t) {
  if (!t) return;
  if (t->solver) vio_pnp_destroy(t->solver);
  delete t;
}

int vio_pnp_tracker_clear(vio_pnp_tracker_t *t, int32_t seq) {
  if (!t || seq < 0 || seq >= t->n_seq) return VIO_EINVAL;
  clear_state(t, t->seq[seq]);
  return VIO_OK;
}

// vinsPnP::setInit (:53-72): the newest back-end result — biases for every frame, and pose / speed of the window frame
// with the same header, which becomes a constant of the next solves.
int vio_pnp_tracker_set_init(vio_pnp_tracker_t *t, int32_t seq, const VioVinsResult *r) {
  if (!t || seq < 0 || seq >= t->n_seq || !r) return VIO_EINVAL;
  Seq &s = t->seq[seq];
  for (int i = 0; i <= t->size; i++) {
    memcpy(&s.Bas[3 * i], r->Ba, 24), memcpy(&s.Bgs[3 * i], r->Bg, 24);
    if (s.Headers[i] == r->header) {
      s.find_solved[i] = 1;
      memcpy(&s.Ps[3 * i], r->P, 24), memcpy(&s.Rs[9 * i], r->R, 72), memcpy(&s.Vs[3 * i], r->V, 24);
    }
  }
  return VIO_OK;
}

int vio_pnp_tracker_process_imu(vio_pnp_tracker_t *t, int32_t seq, double dt, const double acc[3], const double gyr[3]) {
  if (!t || seq < 0 || seq >= t->n_seq || !acc || !gyr) return VIO_EINVAL;  // vinsPnP::processIMU (:174-215)
  Seq &s = t->seq[seq];
  if (!s.first_imu) {
    s.first_imu = true;
    memcpy(s.acc_0, acc, 24), memcpy(s.gyr_0, gyr, 24);
  }
  const int j = s.frame_count;
  if (!s.pre_valid[j]) new_pre(t, s, j);
  if (j != 0) {
    host::propagate(s.pre[j], dt, acc, gyr);
    const double g[3] = {0, 0, t->cfg.gravity};
    double *R = &s.Rs[9 * j], *P = &s.Ps[3 * j], *V = &s.Vs[3 * j];
    const double *ba = &s.Bas[3 * j], *bg = &s.Bgs[3 * j];
    double a0[3], ua0[3], w[3], a1[3], ua1[3], dR[9], Rn[9];
    for (int k = 0; k < 3; k++) a0[k] = s.acc_0[k] - ba[k], w[k] = 0.5 * (s.gyr_0[k] + gyr[k]) - bg[k];
    mat3vec(R, a0, ua0);
    qtoR(Quat{w[0] * dt / 2, w[1] * dt / 2, w[2] * dt / 2, 1.0}, dR);  // Utility::deltaQ, unnormalized
    mat3mul(R, dR, Rn);
    memcpy(R, Rn, 72);
    for (int k = 0; k < 3; k++) a1[k] = acc[k] - ba[k];
    mat3vec(R, a1, ua1);
    for (int k = 0; k < 3; k++) {
      const double ua = 0.5 * ((ua0[k] - g[k]) + (ua1[k] - g[k]));
      P[k] += dt * V[k] + 0.5 * dt * dt * ua;
      V[k] += dt * ua;
    }
  }
  memcpy(s.acc_0, acc, 24), memcpy(s.gyr_0, gyr, 24);
  return VIO_OK;
}

// vinsPnP::processImage (:217-238) for every active sequence; the solves share one launch. P_out / R_out receive
// Ps[PNP_SIZE - 1] / Rs[PNP_SIZE - 1] after the slide, what solveVinsPnP hands back (feature_tracker.cpp:155-156).
int vio_pnp_tracker_process_images(vio_pnp_tracker_t *t, const VioPnpFeature *features, const int32_t *n_features, int32_t stride,
                                   const double *headers, int32_t use_pnp, const uint8_t *active, double *P_out, double *R_out,
                                   int32_t *solved) {
  if (!t || !n_features || !headers || stride < 0) return VIO_EINVAL;
  const int N = t->size;
  t->solving.clear();
  for (int q = 0; q < t->n_seq; q++) {
    if (solved) solved[q] = 0;
    if (active && !active[q]) continue;
    if (n_features[q] < 0 || n_features[q] > stride || (n_features[q] > 0 && !features)) return VIO_EINVAL;
    Seq &s = t->seq[q];
    std::vector<VioPnpFeature> msg(features + (size_t)q * stride, features + (size_t)q * stride + n_features[q]);
    s.features[s.frame_count] = msg;
    s.Headers[s.frame_count] = headers[q];
    update_features(s, msg);
    if (s.frame_count < N) {
      s.frame_count++;
      continue;
    }
    if (use_pnp) {
      for (int i = 1; i <= N; i++)
        if (!s.pre_valid[i]) return VIO_ESTATE;
      build_window(t, s, &t->windows[t->solving.size()]);
      t->solving.push_back(q);
    } else {
      slide_window(t, s);
    }
  }
  const int n = (int)t->solving.size();
  if (n > 0) {
    if (!t->solver) {
      int rc = vio_pnp_create(&t->cfg, t->n_seq, &t->solver);
      if (rc != VIO_OK) return rc;
    }
    int rc = vio_pnp_solve_windows(t->solver, t->windows.data(), n, nullptr);
    if (rc != VIO_OK) return rc;
    for (int k = 0; k < n; k++) {
      Seq &s = t->seq[t->solving[k]];
      for (int i = 0; i <= N; i++) {  // new2old (:137-172)
        const double *p = &s.pose[7 * i];
        qtoR(Quat{p[3], p[4], p[5], p[6]}, &s.Rs[9 * i]);
        for (int c = 0; c < 3; c++) s.Ps[3 * i + c] = p[c], s.Vs[3 * i + c] = s.speed[3 * i + c];
      }
      slide_window(t, s);
      if (solved) solved[t->solving[k]] = 1;
    }
  }
  for (int q = 0; q < t->n_seq; q++) {
    if (active && !active[q]) continue;
    const Seq &s = t->seq[q];
    if (P_out) memcpy(P_out + 3 * (size_t)q, &s.Ps[3 * (N - 1)], 24);
    if (R_out) memcpy(R_out + 9 * (size_t)q, &s.Rs[9 * (N - 1)], 72);
  }
  return VIO_OK;
}

// The join at the top of solveVinsPnP (feature_tracker.cpp:121-134): the landmarks the back-end has solved (ascending id)
// against the tracker's current ids / points (ascending id too: ids grow with n_id and the vectors are only ever
// compacted); a match takes the back-end's position and track count and the tracker's current normalized observation.
int vio_pnp_match_features(const VioConfig *cfg, const int32_t *ids, const float *forw_pts, int32_t n_pts,
                           const VioPnpFeature *solved, int32_t n_solved, VioPnpFeature *out, int32_t cap, int32_t *n_out) {
  if (!cfg || !n_out || n_pts < 0 || n_solved < 0 || (n_pts > 0 && (!ids || !forw_pts)) || (n_solved > 0 && !solved) ||
      (cap > 0 && !out))
    return VIO_EINVAL;
  int i = 0, m = 0;
  for (int k = 0; k < n_solved; k++) {
    while (i < n_pts && ids[i] < solved[k].id) i++;  // (the reference walks ids[] without the bound)
    if (i < n_pts && ids[i] == solved[k].id) {
      if (m >= cap) return VIO_ECAP;
      out[m] = solved[k];
      out[m].observation[0] = ((double)forw_pts[2 * i] - cfg->cx) / cfg->fx;
      out[m].observation[1] = ((double)forw_pts[2 * i + 1] - cfg->cy) / cfg->fy;
      m++;
    }
  }
  *n_out = m;
  return VIO_OK;
}

int vio_pnp_tracker_get_window(vio_pnp_tracker_t *t, int32_t seq, double *Ps, double *Rs, double *Vs, double *headers,
                               uint8_t *find_solved, int32_t *frame_count) {
  if (!t || seq < 0 || seq >= t->n_seq) return VIO_EINVAL;
  const Seq &s = t->seq[seq];
  const int P = t->size + 1;
  if (Ps) memcpy(Ps, s.Ps.data(), sizeof(double) * 3 * P);
  if (Rs) memcpy(Rs, s.Rs.data(), sizeof(double) * 9 * P);
  if (Vs) memcpy(Vs, s.Vs.data(), sizeof(double) * 3 * P);
  if (headers) memcpy(headers, s.Headers.data(), sizeof(double) * P);
  if (find_solved)
    for (int i = 0; i < P; i++) find_solved[i] = s.find_solved[i] ? 1 : 0;
  if (frame_count) *frame_count = s.frame_count;
  return VIO_OK;
}

}  // extern "C"
