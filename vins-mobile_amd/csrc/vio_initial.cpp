// vio_initial.cpp — the estimator's one-off initialisation (host side): visual-inertial alignment, relative pose, global
// structure from motion. Runs once per (re)start on at most a few dozen frames and a few hundred landmarks; like the
// reference it is CPU code next to the solver.
//
// Reference: initial_aligment.cpp:10-229 (solveGyroscopeBias, TangentBasis, RefineGravity, SolveScale,
// VisualIMUAlignment), motion_estimator.cpp:200-236 (solveRelativeRT), inital_sfm.cpp:5-316 (GlobalSFM), driven by
// VINS::solveInitial / visualInitialAlign / relativePose (VINS.cpp:833-1145).
#include "vio_initial.h"

#include <math.h>
#include <string.h>

#include <algorithm>

#include "vio_dense.h"
#include "vio_math.h"

namespace vio {
namespace init {

namespace {
inline void matT_vec(const double R[9], const double v[3], double o[3]) {  // R^T v
  for (int i = 0; i < 3; i++) o[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
}
inline void matT_mat(const double A[9], const double B[9], double C[9]) {  // A^T B
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
}  // namespace

void repropagate(const VioConfig &cfg, Frame &f, const double ba[3], const double bg[3]) {  // integration_base.h:47-61
  host::preint_init(f.pre, &cfg, f.lin_acc, f.lin_gyr, ba, bg);
  for (size_t k = 0; k < f.dt.size(); k++) host::propagate(f.pre, f.dt[k], &f.acc[3 * k], &f.gyr[3 * k]);
}

// solveGyroscopeBias (initial_aligment.cpp:10-45): the rotation the camera saw between consecutive frames against the
// pre-integrated one, linearised in the gyroscope bias.
static void solve_gyroscope_bias(const VioConfig &cfg, std::vector<Frame> &frames, int window_size, double *Bgs) {
  std::vector<double> A(9, 0.0), b(3, 0.0), dbg;
  for (size_t i = 0; i + 1 < frames.size(); i++) {
    const Frame &fi = frames[i];
    const Frame &fj = frames[i + 1];
    double Rij[9];
    matT_mat(fi.R, fj.R, Rij);
    const Quat q_ij = RtoQ(Rij);
    double tA[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) tA[r * 3 + c] = fj.pre.J[(3 + r) * 15 + 12 + c];  // block<3,3>(O_R, O_BG)
    const Quat e = qmul(qinv(fj.pre.dq), q_ij);
    const double tb[3] = {2 * e.x, 2 * e.y, 2 * e.z};
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) A[r * 3 + c] += tA[r] * tA[c] + tA[3 + r] * tA[3 + c] + tA[6 + r] * tA[6 + c];
      b[r] += tA[r] * tb[0] + tA[3 + r] * tb[1] + tA[6 + r] * tb[2];
    }
  }
  if (!dense::ldlt_solve(A, b, 3, dbg)) dbg.assign(3, 0.0);
  for (int i = 0; i <= window_size; i++)
    for (int k = 0; k < 3; k++) Bgs[3 * i + k] += dbg[k];
  const double zero[3] = {0, 0, 0};
  for (size_t i = 0; i + 1 < frames.size(); i++) repropagate(cfg, frames[i + 1], zero, Bgs);
}

static void tangent_basis(const double g0[3], double bc[6] /* 3x2 row-major */) {  // initial_aligment.cpp:48-61
  const double n = sqrt(g0[0] * g0[0] + g0[1] * g0[1] + g0[2] * g0[2]);
  const double a[3] = {g0[0] / n, g0[1] / n, g0[2] / n};
  double tmp[3] = {0, 0, 1};
  if (a[0] == tmp[0] && a[1] == tmp[1] && a[2] == tmp[2]) tmp[0] = 1, tmp[2] = 0;
  const double d = a[0] * tmp[0] + a[1] * tmp[1] + a[2] * tmp[2];
  double b[3] = {tmp[0] - a[0] * d, tmp[1] - a[1] * d, tmp[2] - a[2] * d};
  const double bn = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
  for (int k = 0; k < 3; k++) b[k] /= bn;
  const double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  for (int k = 0; k < 3; k++) bc[2 * k] = b[k], bc[2 * k + 1] = c[k];
}

// The 6 x (6 + ng + 1) block one frame pair contributes to the alignment system and its right-hand side; ng = 3 (SolveScale:
// free gravity vector) or 2 (RefineGravity: correction in the tangent plane of g0, whose known part moves to the rhs).
static void pair_block(const Frame &fi, const Frame &fj, const double tic[3], int ng, const double *lxly, const double *g0,
                       double *tA /* 6 x (7+ng) */, double tb[6]) {
  const int nc = 6 + ng + 1;
  memset(tA, 0, sizeof(double) * 6 * nc), memset(tb, 0, sizeof(double) * 6);
  const double dt = fj.pre.sum_dt;
  double RiT[9], RiTRj[9], d[3], v[3];
  mat3T(fi.R, RiT);
  mat3mul(RiT, fj.R, RiTRj);
  for (int k = 0; k < 3; k++) tA[k * nc + k] = -dt, tA[(3 + k) * nc + k] = -1.0;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) tA[(3 + r) * nc + 3 + c] = RiTRj[r * 3 + c];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < ng; c++) {
      double gp = 0, gv = 0;  // R_i^T * (dt^2/2 | dt) * [I | lxly]
      if (ng == 3) {
        gp = RiT[r * 3 + c] * dt * dt / 2, gv = RiT[r * 3 + c] * dt;
      } else {
        for (int k = 0; k < 3; k++) gp += RiT[r * 3 + k] * dt * dt / 2 * lxly[2 * k + c], gv += RiT[r * 3 + k] * dt * lxly[2 * k + c];
      }
      tA[r * nc + 6 + c] = gp, tA[(3 + r) * nc + 6 + c] = gv;
    }
  for (int k = 0; k < 3; k++) d[k] = fj.T[k] - fi.T[k];
  mat3vec(RiT, d, v);
  for (int r = 0; r < 3; r++) tA[r * nc + 6 + ng] = v[r] / 100.0;
  mat3vec(RiTRj, tic, v);
  for (int r = 0; r < 3; r++) tb[r] = fj.pre.dp[r] + v[r] - tic[r], tb[3 + r] = fj.pre.dv[r];
  if (ng == 2) {
    double a[3], w[3];
    for (int k = 0; k < 3; k++) a[k] = dt * dt / 2 * g0[k], w[k] = dt * g0[k];
    double ra[3], rw[3];
    mat3vec(RiT, a, ra), mat3vec(RiT, w, rw);
    for (int r = 0; r < 3; r++) tb[r] -= ra[r], tb[3 + r] -= rw[r];
  }
}

// Accumulates r_A = tA^T tA, r_b = tA^T tb into the arrow-shaped system (initial_aligment.cpp:111-123, 183-194).
static void scatter(const double *tA, const double tb[6], int nc, int i, int n_state, std::vector<double> &A,
                    std::vector<double> &b) {
  const int tail = nc - 6;
  auto idx = [&](int c) { return c < 6 ? i * 3 + c : n_state - tail + (c - 6); };
  for (int p = 0; p < nc; p++) {
    double rb = 0;
    for (int r = 0; r < 6; r++) rb += tA[r * nc + p] * tb[r];
    b[idx(p)] += rb;
    for (int q = 0; q < nc; q++) {
      double ra = 0;
      for (int r = 0; r < 6; r++) ra += tA[r * nc + p] * tA[r * nc + q];
      A[(size_t)idx(p) * n_state + idx(q)] += ra;
    }
  }
}

// RefineGravity (initial_aligment.cpp:63-133). A and b are NOT cleared between the four passes in the reference: every
// pass adds its blocks to the (already 1000x scaled) system of the previous ones — restated as written.
static bool refine_gravity(const std::vector<Frame> &frames, const double tic[3], double gnorm, double g[3], std::vector<double> &x) {
  const double n0 = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
  double g0[3] = {g[0] / n0 * gnorm, g[1] / n0 * gnorm, g[2] / n0 * gnorm};
  const int n = (int)frames.size(), n_state = n * 3 + 2 + 1;
  std::vector<double> A((size_t)n_state * n_state, 0.0), b(n_state, 0.0);
  for (int k = 0; k < 4; k++) {
    double lxly[6];
    tangent_basis(g0, lxly);
    for (int i = 0; i + 1 < n; i++) {
      double tA[6 * 9], tb[6];
      pair_block(frames[i], frames[i + 1], tic, 2, lxly, g0, tA, tb);
      scatter(tA, tb, 9, i, n_state, A, b);
    }
    for (double &v : A) v *= 1000.0;
    for (double &v : b) v *= 1000.0;
    std::vector<double> Ac(A), bc(b);
    if (!dense::ldlt_solve(Ac, bc, n_state, x)) return false;
    const double dg0 = x[n_state - 3], dg1 = x[n_state - 2];
    double t[3];
    for (int r = 0; r < 3; r++) t[r] = g0[r] + lxly[2 * r] * dg0 + lxly[2 * r + 1] * dg1;
    const double tn = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    for (int r = 0; r < 3; r++) g0[r] = t[r] / tn * gnorm;
  }
  memcpy(g, g0, sizeof(g0));
  return true;
}

// SolveScale (initial_aligment.cpp:135-221).
static bool solve_scale(const std::vector<Frame> &frames, const double tic[3], double gnorm, double g[3], std::vector<double> &x) {
  const int n = (int)frames.size(), n_state = n * 3 + 3 + 1;
  std::vector<double> A((size_t)n_state * n_state, 0.0), b(n_state, 0.0);
  for (int i = 0; i + 1 < n; i++) {
    double tA[6 * 10], tb[6];
    pair_block(frames[i], frames[i + 1], tic, 3, nullptr, nullptr, tA, tb);
    scatter(tA, tb, 10, i, n_state, A, b);
  }
  for (double &v : A) v *= 1000.0;
  for (double &v : b) v *= 1000.0;
  if (!dense::ldlt_solve(A, b, n_state, x)) return false;
  double s = x[n_state - 1] / 100.0;
  for (int k = 0; k < 3; k++) g[k] = x[n_state - 4 + k];
  const double gn = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
  if (fabs(gn - gnorm) > 3.0 /* G_THRESHOLD global_param.hpp:49 */ || s < 0) return false;
  if (!refine_gravity(frames, tic, gnorm, g, x)) return false;
  s = x.back() / 100.0;
  x.back() = s;
  return s > 0.0;
}

bool visual_imu_alignment(const VioConfig &cfg, const double tic[3], std::vector<Frame> &frames, int window_size, double *Bgs,
                          double g[3], std::vector<double> &x) {
  if (frames.size() < 2) return false;
  solve_gyroscope_bias(cfg, frames, window_size, Bgs);
  return solve_scale(frames, tic, cfg.gravity, g, x);
}

}  // namespace init
}  // namespace vio

using namespace vio;

extern "C" int vio_visual_imu_alignment(const VioConfig *cfg, const double tic[3], const VioInitFrame *frames, int32_t n_frames,
                                        int32_t window_size, double *Bgs, double g[3], double *x, int32_t *ok) {
  if (!cfg || !tic || !frames || n_frames < 2 || window_size < 1 || !Bgs || !g || !x || !ok) return VIO_EINVAL;
  std::vector<init::Frame> fr(n_frames);
  const double zero[3] = {0, 0, 0};
  for (int i = 0; i < n_frames; i++) {
    const VioInitFrame &s = frames[i];
    if (s.n_samples < 0 || (s.n_samples > 0 && (!s.dt || !s.acc || !s.gyr))) return VIO_EINVAL;
    init::Frame &f = fr[i];
    f.header = s.header, f.is_key_frame = s.is_key_frame != 0;
    memcpy(f.R, s.R, sizeof(f.R)), memcpy(f.T, s.T, sizeof(f.T));
    memcpy(f.lin_acc, s.acc_0, 24), memcpy(f.lin_gyr, s.gyr_0, 24);
    f.dt.assign(s.dt, s.dt + s.n_samples), f.acc.assign(s.acc, s.acc + 3 * s.n_samples), f.gyr.assign(s.gyr, s.gyr + 3 * s.n_samples);
    init::repropagate(*cfg, f, zero, zero);  // tmp_pre_integration starts from zero biases (VINS.cpp:404)
  }
  std::vector<double> xs;
  *ok = init::visual_imu_alignment(*cfg, tic, fr, window_size, Bgs, g, xs) ? 1 : 0;
  if (xs.size() >= (size_t)3 * n_frames + 1) {
    // x = [velocities | gravity part | scale]: hand back the velocities and the scale
    memcpy(x, xs.data(), sizeof(double) * 3 * n_frames);
    x[3 * n_frames] = xs.back();
  }
  return VIO_OK;
}
