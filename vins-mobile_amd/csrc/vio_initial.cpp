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


// =====================================================================================================================
// Relative pose, PnP, global SfM
// =====================================================================================================================
namespace {

void exp_so3(const double w[3], double R[9]) {  // Rodrigues
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
  double a, b;
  if (th < 1e-8) {
    a = 1.0 - th2 / 6.0, b = 0.5 - th2 / 24.0;
  } else {
    a = sin(th) / th, b = (1.0 - cos(th)) / th2;
  }
  double K[9], K2[9];
  skew3(w, K);
  mat3mul(K, K, K2);
  for (int i = 0; i < 9; i++) R[i] = a * K[i] + b * K2[i];
  R[0] += 1, R[4] += 1, R[8] += 1;
}

void right_update(double R[9], const double w[3]) {  // R <- R exp(w), re-orthonormalised through the quaternion
  double E[9], Rn[9];
  exp_so3(w, E);
  mat3mul(R, E, Rn);
  qtoR(qnormalized(RtoQ(Rn)), R);
}

// Linear triangulation of one point seen in two views (GlobalSFM::triangulatePoint, inital_sfm.cpp:5-21): poses are 3x4
// world -> camera matrices [R | t].
void triangulate_point(const double P0[12], const double P1[12], const double x0[2], const double x1[2], double X[3]) {
  std::vector<double> D(16);
  for (int c = 0; c < 4; c++) {
    D[c] = x0[0] * P0[8 + c] - P0[c], D[4 + c] = x0[1] * P0[8 + c] - P0[4 + c];
    D[8 + c] = x1[0] * P1[8 + c] - P1[c], D[12 + c] = x1[1] * P1[8 + c] - P1[4 + c];
  }
  double v[4];
  dense::null_vector(D, 4, 4, v);
  for (int k = 0; k < 3; k++) X[k] = v[k] / v[3];
}

void make_pose(const double R[9], const double t[3], double P[12]) {
  for (int r = 0; r < 3; r++) P[4 * r] = R[3 * r], P[4 * r + 1] = R[3 * r + 1], P[4 * r + 2] = R[3 * r + 2], P[4 * r + 3] = t[r];
}

// Sum of squared Sampson distances of the correspondences to E = [t]x R (the error cv::findEssentialMat scores with).
double sampson_cost(const std::vector<double> &a, const std::vector<double> &b, const double R[9], const double t[3]) {
  double tx[9], E[9];
  skew3(t, tx);
  mat3mul(tx, R, E);
  double cost = 0;
  const size_t n = a.size() / 2;
  for (size_t i = 0; i < n; i++) {
    const double x1[3] = {a[2 * i], a[2 * i + 1], 1.0}, x2[3] = {b[2 * i], b[2 * i + 1], 1.0};
    double Ex1[3], Etx2[3];
    mat3vec(E, x1, Ex1);
    for (int k = 0; k < 3; k++) Etx2[k] = E[k] * x2[0] + E[3 + k] * x2[1] + E[6 + k] * x2[2];
    const double e = x2[0] * Ex1[0] + x2[1] * Ex1[1] + x2[2] * Ex1[2];
    const double d = Ex1[0] * Ex1[0] + Ex1[1] * Ex1[1] + Etx2[0] * Etx2[0] + Etx2[1] * Etx2[1];
    cost += e * e / (d + 1e-300);
  }
  return cost;
}

}  // namespace

// solveRelativeRT (motion_estimator.cpp:200-236) = cv::findEssentialMat(ll, rr) + cv::recoverPose(E, ll, rr, rot, trans).
// findEssentialMat runs a five-point RANSAC whose default threshold (1.0, in NORMALIZED image units here because the
// reference passes focal = 1) accepts every correspondence, i.e. it returns the essential matrix of its first random
// minimal sample. OpenCV is not available (and its random sample cannot be reproduced without it), so the rotation and
// the translation direction are instead fitted to ALL correspondences: Levenberg-Marquardt on the Sampson error over the
// 5 degrees of freedom of (R, t/|t|), started from R = I and six translation directions (the window spans about a second:
// the rotation between its frames is small), which also holds on planar scenes where a linear eight-point fit is
// degenerate. recoverPose's part — choosing the sign of t by the points in front of both cameras (closer than 50) and
// counting them — follows the published algorithm.
//
// One addition over the reference: a scene that is (nearly) a plane admits two exact two-view solutions, and nothing in
// the correspondences tells them apart (OpenCV's sampler picks one by chance). When the caller knows roughly how the
// camera rotated (R_hint: second camera in the first, e.g. from the gyroscope), the local minima whose cost is within a
// factor of the best are compared against it and the closest rotation wins.
bool solve_relative_rt(const std::vector<double> &xy0, const std::vector<double> &xy1, double Rout[9], double tout[3],
                       int *inliers, const double *R_hint) {
  const size_t n = xy0.size() / 2;
  if (inliers) *inliers = 0;
  if (n < 9 || xy1.size() != xy0.size()) return false;
  double best_cost = 1e300, bestR[9], bestt[3];
  double cand_cost[6], candR[6][9], candt[6][3];
  const double dirs[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  for (int start = 0; start < 6; start++) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {dirs[start][0], dirs[start][1], dirs[start][2]};
    double cost = sampson_cost(xy0, xy1, R, t), lambda = 1e-3;
    for (int it = 0; it < 60; it++) {
      // tangent basis of the unit sphere at t
      double bc[6];
      tangent_basis(t, bc);
      // Jacobian of the n Sampson residuals by forward differences in the 5 local parameters (3 rotation, 2 on the sphere)
      auto residuals = [&](const double Rr[9], const double tt[3], std::vector<double> &r) {
        double tx[9], E[9];
        skew3(tt, tx);
        mat3mul(tx, Rr, E);
        r.resize(n);
        for (size_t i = 0; i < n; i++) {
          const double x1[3] = {xy0[2 * i], xy0[2 * i + 1], 1.0}, x2[3] = {xy1[2 * i], xy1[2 * i + 1], 1.0};
          double Ex1[3], Etx2[3];
          mat3vec(E, x1, Ex1);
          for (int k = 0; k < 3; k++) Etx2[k] = E[k] * x2[0] + E[3 + k] * x2[1] + E[6 + k] * x2[2];
          const double e = x2[0] * Ex1[0] + x2[1] * Ex1[1] + x2[2] * Ex1[2];
          const double d = Ex1[0] * Ex1[0] + Ex1[1] * Ex1[1] + Etx2[0] * Etx2[0] + Etx2[1] * Etx2[1];
          r[i] = e / sqrt(d + 1e-300);
        }
      };
      auto perturbed = [&](const double p[5], double Rp[9], double tp[3]) {
        memcpy(Rp, R, sizeof(double) * 9);
        right_update(Rp, p);
        for (int k = 0; k < 3; k++) tp[k] = t[k] + bc[2 * k] * p[3] + bc[2 * k + 1] * p[4];
        const double tn = sqrt(tp[0] * tp[0] + tp[1] * tp[1] + tp[2] * tp[2]);
        for (int k = 0; k < 3; k++) tp[k] /= tn;
      };
      std::vector<double> r0, r1, J(n * 5);
      residuals(R, t, r0);
      const double h = 1e-6;
      for (int c = 0; c < 5; c++) {
        double p[5] = {0, 0, 0, 0, 0}, Rp[9], tp[3];
        p[c] = h;
        perturbed(p, Rp, tp);
        residuals(Rp, tp, r1);
        for (size_t i = 0; i < n; i++) J[i * 5 + c] = (r1[i] - r0[i]) / h;
      }
      std::vector<double> H(25, 0.0), g(5, 0.0);
      for (size_t i = 0; i < n; i++)
        for (int a = 0; a < 5; a++) {
          g[a] -= J[i * 5 + a] * r0[i];
          for (int b = 0; b < 5; b++) H[a * 5 + b] += J[i * 5 + a] * J[i * 5 + b];
        }
      bool improved = false;
      for (int tries = 0; tries < 8 && !improved; tries++) {
        std::vector<double> Hd(H), gd(g), dx;
        for (int a = 0; a < 5; a++) Hd[a * 5 + a] += lambda * (H[a * 5 + a] + 1e-12);
        if (!dense::ldlt_solve(Hd, gd, 5, dx)) break;
        double Rp[9], tp[3];
        perturbed(dx.data(), Rp, tp);
        const double c2 = sampson_cost(xy0, xy1, Rp, tp);
        if (c2 < cost) {
          const double rel = (cost - c2) / (cost + 1e-300);
          memcpy(R, Rp, sizeof(Rp)), memcpy(t, tp, sizeof(tp));
          cost = c2, lambda = std::max(lambda * 0.3, 1e-9), improved = true;
          if (rel < 1e-10) it = 1000;
        } else {
          lambda *= 10;
        }
      }
      if (!improved) break;
    }
    cand_cost[start] = cost, memcpy(candR[start], R, sizeof(R)), memcpy(candt[start], t, sizeof(t));
    if (cost < best_cost) best_cost = cost, memcpy(bestR, R, sizeof(R)), memcpy(bestt, t, sizeof(t));
  }
  if (R_hint) {  // x2 ~ R x1 + t: R is the first camera seen from the second, the hint is the other way round
    double best_angle = 1e300;
    for (int c = 0; c < 6; c++) {
      if (!(cand_cost[c] <= 3.0 * best_cost + 1e-12 * n)) continue;
      double M[9];
      mat3mul(candR[c], R_hint, M);  // R * R_hint = I when they agree
      const double tr = std::min(3.0, std::max(-1.0, M[0] + M[4] + M[8]));
      const double angle = acos((tr - 1.0) / 2.0);
      if (angle < best_angle) best_angle = angle, memcpy(bestR, candR[c], sizeof(bestR)), memcpy(bestt, candt[c], sizeof(bestt));
    }
  }
  // recoverPose: x2 ~ R x1 + t; the sign of t with more points in front of both cameras within distance 50
  int best_good = -1;
  double sign = 1;
  for (int sgn = 0; sgn < 2; sgn++) {
    const double tt[3] = {sgn ? -bestt[0] : bestt[0], sgn ? -bestt[1] : bestt[1], sgn ? -bestt[2] : bestt[2]};
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero[3] = {0, 0, 0};
    double P0[12], P1[12];
    make_pose(I3, zero, P0), make_pose(bestR, tt, P1);
    int good = 0;
    for (size_t i = 0; i < n; i++) {
      double X[3], X2[3];
      triangulate_point(P0, P1, &xy0[2 * i], &xy1[2 * i], X);
      mat3vec(bestR, X, X2);
      const double z2 = X2[2] + tt[2];
      if (X[2] > 0 && X[2] < 50 && z2 > 0 && z2 < 50) good++;
    }
    if (good > best_good) best_good = good, sign = sgn ? -1 : 1;
  }
  // the reference returns the pose of the second camera in the first: Rotation = R^T, Translation = -R^T t
  mat3T(bestR, Rout);
  const double ts[3] = {sign * bestt[0], sign * bestt[1], sign * bestt[2]};
  double v[3];
  mat3vec(Rout, ts, v);
  for (int k = 0; k < 3; k++) tout[k] = -v[k];
  if (inliers) *inliers = best_good;
  return best_good > 10;
}

// cv::solvePnP(pts3, pts2, K = I, D = none, rvec, t, useExtrinsicGuess = true) (ITERATIVE): Levenberg-Marquardt on the
// reprojection error from the guess (OpenCV: CvLevMarq, at most 20 iterations).
bool pnp_refine(const std::vector<double> &pts3, const std::vector<double> &pts2, double R[9], double t[3]) {
  const size_t n = pts3.size() / 3;
  if (n < 4 || pts2.size() != 2 * n) return false;
  auto cost_of = [&](const double Rr[9], const double tt[3]) {
    double c = 0;
    for (size_t i = 0; i < n; i++) {
      double X[3];
      mat3vec(Rr, &pts3[3 * i], X);
      for (int k = 0; k < 3; k++) X[k] += tt[k];
      const double ex = X[0] / X[2] - pts2[2 * i], ey = X[1] / X[2] - pts2[2 * i + 1];
      c += ex * ex + ey * ey;
    }
    return c;
  };
  double cost = cost_of(R, t), lambda = 1e-3;
  for (int it = 0; it < 20; it++) {
    std::vector<double> H(36, 0.0), g(6, 0.0);
    for (size_t i = 0; i < n; i++) {
      const double *Xw = &pts3[3 * i];
      double X[3];
      mat3vec(R, Xw, X);
      for (int k = 0; k < 3; k++) X[k] += t[k];
      const double iz = 1.0 / X[2], r[2] = {X[0] * iz - pts2[2 * i], X[1] * iz - pts2[2 * i + 1]};
      const double Jp[6] = {iz, 0, -X[0] * iz * iz, 0, iz, -X[1] * iz * iz};
      double Sx[9], RS[9];
      skew3(Xw, Sx);
      mat3mul(R, Sx, RS);  // dXc/dw = -R [Xw]x
      double J[12];
      for (int r2 = 0; r2 < 2; r2++)
        for (int c = 0; c < 3; c++) {
          J[r2 * 6 + c] = -(Jp[r2 * 3] * RS[c] + Jp[r2 * 3 + 1] * RS[3 + c] + Jp[r2 * 3 + 2] * RS[6 + c]);
          J[r2 * 6 + 3 + c] = Jp[r2 * 3 + c];
        }
      for (int a = 0; a < 6; a++) {
        g[a] -= J[a] * r[0] + J[6 + a] * r[1];
        for (int b = 0; b < 6; b++) H[a * 6 + b] += J[a] * J[b] + J[6 + a] * J[6 + b];
      }
    }
    bool improved = false;
    for (int tries = 0; tries < 8 && !improved; tries++) {
      std::vector<double> Hd(H), gd(g), dx;
      for (int a = 0; a < 6; a++) Hd[a * 6 + a] += lambda * (H[a * 6 + a] + 1e-12);
      if (!dense::ldlt_solve(Hd, gd, 6, dx)) return false;
      double Rn[9], tn[3];
      memcpy(Rn, R, sizeof(Rn));
      right_update(Rn, dx.data());
      for (int k = 0; k < 3; k++) tn[k] = t[k] + dx[3 + k];
      const double c2 = cost_of(Rn, tn);
      if (std::isfinite(c2) && c2 < cost) {
        const double rel = (cost - c2) / (cost + 1e-300);
        memcpy(R, Rn, sizeof(Rn)), memcpy(t, tn, sizeof(tn));
        cost = c2, lambda = std::max(lambda * 0.3, 1e-9), improved = true;
        if (rel < 1e-12) it = 1000;
      } else {
        lambda *= 10;
      }
    }
    if (!improved) break;
  }
  return std::isfinite(cost);
}

namespace {

bool solve_frame_by_pnp(double R[9], double t[3], int i, const std::vector<SfmFeature> &sfm_f) {  // inital_sfm.cpp:24-74
  std::vector<double> p2, p3;
  for (const SfmFeature &f : sfm_f) {
    if (!f.state) continue;
    for (const auto &o : f.observation)
      if (o.first == i) {
        p2.push_back(o.second.first), p2.push_back(o.second.second);
        p3.insert(p3.end(), f.position, f.position + 3);
        break;
      }
  }
  if ((int)p2.size() / 2 < 15) return false;  // "feature tracking not enough, please slowly move you device!"
  return pnp_refine(p3, p2, R, t);
}

void triangulate_two_frames(int f0, const double P0[12], int f1, const double P1[12], std::vector<SfmFeature> &sfm_f) {
  for (SfmFeature &f : sfm_f) {  // inital_sfm.cpp:76-115
    if (f.state) continue;
    bool has0 = false, has1 = false;
    double x0[2], x1[2];
    for (const auto &o : f.observation) {
      if (o.first == f0) x0[0] = o.second.first, x0[1] = o.second.second, has0 = true;
      if (o.first == f1) x1[0] = o.second.first, x1[1] = o.second.second, has1 = true;
    }
    if (has0 && has1) {
      triangulate_point(P0, P1, x0, x1, f.position);
      f.state = true;
    }
  }
}

// Full bundle adjustment of GlobalSFM::construct (inital_sfm.cpp:229-296): reprojection error in normalized image
// coordinates over all rotations (but frame l's), all translations (but frame l's and the last frame's) and all
// triangulated points, no robust loss. Ceres' default trust-region Levenberg-Marquardt with a dense Schur complement is
// restated as a plain LM with the same elimination order (points first); the optimum, not the iterate path, is what the
// initialisation consumes. Returns the final cost (sum of squares / 2) and whether the iteration converged.
bool bundle_adjust(int frame_num, int l, std::vector<double> &Rc, std::vector<double> &tc, std::vector<SfmFeature> &sfm_f,
                   double *final_cost) {
  struct Obs {
    int frame, point;
    double u, v;
  };
  std::vector<int> pidx;
  std::vector<Obs> obs;
  for (size_t j = 0; j < sfm_f.size(); j++) {
    if (!sfm_f[j].state) continue;
    for (const auto &o : sfm_f[j].observation) obs.push_back({o.first, (int)pidx.size(), o.second.first, o.second.second});
    pidx.push_back((int)j);
  }
  const int np = (int)pidx.size();
  // camera parameter offsets in the reduced system
  std::vector<int> off_r(frame_num, -1), off_t(frame_num, -1);
  int nc = 0;
  for (int i = 0; i < frame_num; i++) {
    if (i != l) off_r[i] = nc, nc += 3;
    if (i != l && i != frame_num - 1) off_t[i] = nc, nc += 3;
  }
  std::vector<double> X(3 * np);
  for (int p = 0; p < np; p++) memcpy(&X[3 * p], sfm_f[pidx[p]].position, 24);
  auto cost_of = [&](const std::vector<double> &R, const std::vector<double> &t, const std::vector<double> &Xp) {
    double c = 0;
    for (const Obs &o : obs) {
      double Y[3];
      mat3vec(&R[9 * o.frame], &Xp[3 * o.point], Y);
      for (int k = 0; k < 3; k++) Y[k] += t[3 * o.frame + k];
      const double ex = Y[0] / Y[2] - o.u, ey = Y[1] / Y[2] - o.v;
      c += ex * ex + ey * ey;
    }
    return 0.5 * c;
  };
  double cost = cost_of(Rc, tc, X), lambda = 1e-4;
  bool converged = false;
  for (int it = 0; it < 50 && !converged; it++) {
    std::vector<double> Hcc((size_t)nc * nc, 0.0), gc(nc, 0.0), Hpp(9 * (size_t)np, 0.0), gp(3 * (size_t)np, 0.0);
    std::vector<double> Hcp((size_t)nc * 3 * np, 0.0);  // dense: at most 63 x ~900
    for (const Obs &o : obs) {
      const double *R = &Rc[9 * o.frame], *Xw = &X[3 * o.point];
      double Y[3];
      mat3vec(R, Xw, Y);
      for (int k = 0; k < 3; k++) Y[k] += tc[3 * o.frame + k];
      const double iz = 1.0 / Y[2], r[2] = {Y[0] * iz - o.u, Y[1] * iz - o.v};
      const double Jp[6] = {iz, 0, -Y[0] * iz * iz, 0, iz, -Y[1] * iz * iz};
      double Sx[9], RS[9];
      skew3(Xw, Sx);
      mat3mul(R, Sx, RS);
      double Jr[6], Jt[6], JX[6];
      for (int a = 0; a < 2; a++)
        for (int c = 0; c < 3; c++) {
          Jr[a * 3 + c] = -(Jp[a * 3] * RS[c] + Jp[a * 3 + 1] * RS[3 + c] + Jp[a * 3 + 2] * RS[6 + c]);
          Jt[a * 3 + c] = Jp[a * 3 + c];
          JX[a * 3 + c] = Jp[a * 3] * R[c] + Jp[a * 3 + 1] * R[3 + c] + Jp[a * 3 + 2] * R[6 + c];
        }
      // camera columns of this observation: up to 6
      int cols[6], ncol = 0;
      double Jc[12];
      if (off_r[o.frame] >= 0)
        for (int c = 0; c < 3; c++) cols[ncol] = off_r[o.frame] + c, Jc[ncol] = Jr[c], Jc[6 + ncol] = Jr[3 + c], ncol++;
      if (off_t[o.frame] >= 0)
        for (int c = 0; c < 3; c++) cols[ncol] = off_t[o.frame] + c, Jc[ncol] = Jt[c], Jc[6 + ncol] = Jt[3 + c], ncol++;
      for (int a = 0; a < ncol; a++) {
        gc[cols[a]] -= Jc[a] * r[0] + Jc[6 + a] * r[1];
        for (int b = 0; b < ncol; b++) Hcc[(size_t)cols[a] * nc + cols[b]] += Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
        for (int c = 0; c < 3; c++) Hcp[(size_t)cols[a] * 3 * np + 3 * o.point + c] += Jc[a] * JX[c] + Jc[6 + a] * JX[3 + c];
      }
      for (int a = 0; a < 3; a++) {
        gp[3 * o.point + a] -= JX[a] * r[0] + JX[3 + a] * r[1];
        for (int b = 0; b < 3; b++) Hpp[9 * (size_t)o.point + a * 3 + b] += JX[a] * JX[b] + JX[3 + a] * JX[3 + b];
      }
    }
    bool improved = false;
    for (int tries = 0; tries < 10 && !improved; tries++) {
      // Schur complement on the points: S = Hcc' - Hcp Hpp'^-1 Hpc, rhs = gc - Hcp Hpp'^-1 gp (damped diagonals)
      std::vector<double> S(Hcc), rhs(gc), Hinv(9 * (size_t)np);
      for (int a = 0; a < nc; a++) S[(size_t)a * nc + a] += lambda * (Hcc[(size_t)a * nc + a] + 1e-12);
      for (int p = 0; p < np; p++) {
        double M[9];
        memcpy(M, &Hpp[9 * (size_t)p], sizeof(M));
        for (int a = 0; a < 3; a++) M[a * 3 + a] += lambda * (M[a * 3 + a] + 1e-12);
        const double det = dense::det3(M);
        double *I = &Hinv[9 * (size_t)p];
        I[0] = (M[4] * M[8] - M[5] * M[7]) / det, I[1] = (M[2] * M[7] - M[1] * M[8]) / det, I[2] = (M[1] * M[5] - M[2] * M[4]) / det;
        I[3] = (M[5] * M[6] - M[3] * M[8]) / det, I[4] = (M[0] * M[8] - M[2] * M[6]) / det, I[5] = (M[2] * M[3] - M[0] * M[5]) / det;
        I[6] = (M[3] * M[7] - M[4] * M[6]) / det, I[7] = (M[1] * M[6] - M[0] * M[7]) / det, I[8] = (M[0] * M[4] - M[1] * M[3]) / det;
      }
      std::vector<double> W((size_t)nc * 3 * np);  // Hcp Hpp^-1
      for (int a = 0; a < nc; a++)
        for (int p = 0; p < np; p++) {
          const double *h = &Hcp[(size_t)a * 3 * np + 3 * p], *I = &Hinv[9 * (size_t)p];
          double *w = &W[(size_t)a * 3 * np + 3 * p];
          for (int c = 0; c < 3; c++) w[c] = h[0] * I[c] + h[1] * I[3 + c] + h[2] * I[6 + c];
        }
      for (int a = 0; a < nc; a++) {
        double acc = 0;
        for (int k = 0; k < 3 * np; k++) acc += W[(size_t)a * 3 * np + k] * gp[k];
        rhs[a] -= acc;
        for (int b = 0; b <= a; b++) {
          double sum = 0;
          const double *wa = &W[(size_t)a * 3 * np], *hb = &Hcp[(size_t)b * 3 * np];
          for (int k = 0; k < 3 * np; k++) sum += wa[k] * hb[k];
          S[(size_t)a * nc + b] -= sum;
          if (b != a) S[(size_t)b * nc + a] -= sum;
        }
      }
      std::vector<double> dc;
      if (!dense::ldlt_solve(S, rhs, nc, dc)) {
        lambda *= 10;
        continue;
      }
      std::vector<double> Rn(Rc), tn(tc), Xn(X);
      for (int i = 0; i < frame_num; i++) {
        if (off_r[i] >= 0) right_update(&Rn[9 * i], &dc[off_r[i]]);
        if (off_t[i] >= 0)
          for (int k = 0; k < 3; k++) tn[3 * i + k] += dc[off_t[i] + k];
      }
      for (int p = 0; p < np; p++) {
        double v[3];
        for (int a = 0; a < 3; a++) {
          v[a] = gp[3 * p + a];
          for (int c = 0; c < nc; c++) v[a] -= Hcp[(size_t)c * 3 * np + 3 * p + a] * dc[c];
        }
        const double *I = &Hinv[9 * (size_t)p];
        for (int a = 0; a < 3; a++) Xn[3 * p + a] += I[a * 3] * v[0] + I[a * 3 + 1] * v[1] + I[a * 3 + 2] * v[2];
      }
      const double c2 = cost_of(Rn, tn, Xn);
      if (std::isfinite(c2) && c2 < cost) {
        const double rel = (cost - c2) / (cost + 1e-300);
        Rc.swap(Rn), tc.swap(tn), X.swap(Xn);
        cost = c2, lambda = std::max(lambda / 3, 1e-10), improved = true;
        if (rel < 1e-6) converged = true;  // function_tolerance of ceres::Solver::Options
      } else {
        lambda *= 10;
      }
    }
    if (!improved) {
      converged = true;  // no further decrease possible: the trust region collapsed at a minimum
      break;
    }
  }
  for (int p = 0; p < np; p++) memcpy(sfm_f[pidx[p]].position, &X[3 * p], 24);
  *final_cost = cost;
  return converged;
}

}  // namespace

bool sfm_construct(int frame_num, double *q, double *T, int l, const double relative_R[9], const double relative_T[3],
                   std::vector<SfmFeature> &sfm_f, std::map<int, std::vector<double>> &tracked_points) {
  if (frame_num < 2 || l < 0 || l >= frame_num - 1) return false;
  const int last = frame_num - 1;
  // camera -> reference poses (q, T) and their inverses, world -> camera (c_Rotation, c_Translation)
  std::vector<double> Rc(9 * (size_t)frame_num, 0.0), tc(3 * (size_t)frame_num, 0.0), P(12 * (size_t)frame_num, 0.0);
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(&Rc[9 * l], I3, sizeof(I3));
  mat3T(relative_R, &Rc[9 * last]);
  {
    double v[3];
    mat3vec(&Rc[9 * last], relative_T, v);
    for (int k = 0; k < 3; k++) tc[3 * last + k] = -v[k];
  }
  make_pose(&Rc[9 * l], &tc[3 * l], &P[12 * l]), make_pose(&Rc[9 * last], &tc[3 * last], &P[12 * last]);
  // 1: l .. last-1 against the last frame
  for (int i = l; i < last; i++) {
    if (i > l) {
      memcpy(&Rc[9 * i], &Rc[9 * (i - 1)], 72), memcpy(&tc[3 * i], &tc[3 * (i - 1)], 24);
      if (!solve_frame_by_pnp(&Rc[9 * i], &tc[3 * i], i, sfm_f)) return false;
      make_pose(&Rc[9 * i], &tc[3 * i], &P[12 * i]);
    }
    triangulate_two_frames(i, &P[12 * i], last, &P[12 * last], sfm_f);
  }
  // 2: l+1 .. last-1 against l
  for (int i = l + 1; i < last; i++) triangulate_two_frames(l, &P[12 * l], i, &P[12 * i], sfm_f);
  // 3: l-1 .. 0 (the reference ignores a failing PnP here and keeps the guess)
  for (int i = l - 1; i >= 0; i--) {
    memcpy(&Rc[9 * i], &Rc[9 * (i + 1)], 72), memcpy(&tc[3 * i], &tc[3 * (i + 1)], 24);
    double Rk[9], tk[3];
    memcpy(Rk, &Rc[9 * i], 72), memcpy(tk, &tc[3 * i], 24);
    if (solve_frame_by_pnp(Rk, tk, i, sfm_f)) memcpy(&Rc[9 * i], Rk, 72), memcpy(&tc[3 * i], tk, 24);
    make_pose(&Rc[9 * i], &tc[3 * i], &P[12 * i]);
    triangulate_two_frames(i, &P[12 * i], l, &P[12 * l], sfm_f);
  }
  // 4: everything else seen at least twice, from its first and last observation
  for (SfmFeature &f : sfm_f) {
    if (f.state || f.observation.size() < 2) continue;
    const auto &a = f.observation.front();
    const auto &b = f.observation.back();
    const double x0[2] = {a.second.first, a.second.second}, x1[2] = {b.second.first, b.second.second};
    triangulate_point(&P[12 * a.first], &P[12 * b.first], x0, x1, f.position);
    f.state = true;
  }
  // 5: full BA
  double final_cost = 0;
  const bool converged = bundle_adjust(frame_num, l, Rc, tc, sfm_f, &final_cost);
  if (!(converged || final_cost < 3e-03)) return false;  // "vision only BA not converge"
  for (int i = 0; i < frame_num; i++) {
    double Rt[9], v[3];
    mat3T(&Rc[9 * i], Rt);  // q[i] = c_rotation^-1
    const Quat qi = RtoQ(Rt);
    q[4 * i] = qi.x, q[4 * i + 1] = qi.y, q[4 * i + 2] = qi.z, q[4 * i + 3] = qi.w;
    mat3vec(Rt, &tc[3 * i], v);
    for (int k = 0; k < 3; k++) T[3 * i + k] = -v[k];
  }
  for (const SfmFeature &f : sfm_f)
    if (f.state) tracked_points[f.id] = std::vector<double>(f.position, f.position + 3);
  return true;
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

extern "C" int vio_init_relative_pose(const double *xy0, const double *xy1, int32_t n, const double *R_hint, double R[9],
                                      double t[3], int32_t *inliers, int32_t *ok) {
  if (!xy0 || !xy1 || n < 0 || !R || !t || !ok) return VIO_EINVAL;
  std::vector<double> a(xy0, xy0 + 2 * (size_t)n), b(xy1, xy1 + 2 * (size_t)n);
  int in = 0;
  *ok = init::solve_relative_rt(a, b, R, t, &in, R_hint) ? 1 : 0;
  if (inliers) *inliers = in;
  return VIO_OK;
}

extern "C" int vio_init_pnp(const double *pts3, const double *pts2, int32_t n, double R[9], double t[3], int32_t *ok) {
  if (!pts3 || !pts2 || n < 0 || !R || !t || !ok) return VIO_EINVAL;
  std::vector<double> p3(pts3, pts3 + 3 * (size_t)n), p2(pts2, pts2 + 2 * (size_t)n);
  *ok = init::pnp_refine(p3, p2, R, t) ? 1 : 0;
  return VIO_OK;
}

extern "C" int vio_init_sfm(int32_t frame_num, int32_t l, const double relative_R[9], const double relative_T[3],
                            int32_t n_features, const int32_t *feat_start, const int32_t *obs_frame, const double *obs_xy,
                            double *q, double *T, double *points, uint8_t *point_ok, int32_t *ok) {
  if (frame_num < 2 || !relative_R || !relative_T || n_features < 0 || !feat_start || !obs_frame || !obs_xy || !q || !T || !ok)
    return VIO_EINVAL;
  std::vector<init::SfmFeature> f(n_features);
  for (int j = 0; j < n_features; j++) {
    f[j].id = j;
    for (int k = feat_start[j]; k < feat_start[j + 1]; k++) {
      if (obs_frame[k] < 0 || obs_frame[k] >= frame_num) return VIO_EINVAL;
      f[j].observation.push_back({obs_frame[k], {obs_xy[2 * k], obs_xy[2 * k + 1]}});
    }
  }
  std::map<int, std::vector<double>> tracked;
  *ok = init::sfm_construct(frame_num, q, T, l, relative_R, relative_T, f, tracked) ? 1 : 0;
  for (int j = 0; j < n_features; j++) {
    if (point_ok) point_ok[j] = f[j].state ? 1 : 0;
    if (points) memcpy(points + 3 * (size_t)j, f[j].position, 24);
  }
  return VIO_OK;
}
