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

// Full bundle adjustment of GlobalSFM::construct (inital_sfm.cpp:229-296) as Ceres runs it, restated step by step so
// that the iterates -- not only the optimum -- follow the reference's (oracle/ref_sfm_harness.cpp runs the same problem on
// the vendored Ceres; tests/test_initial_sfm.py compares the iteration traces):
//   * parameter blocks: c_rotation[i] (w x y z, world -> camera) on ceres::QuaternionParameterization -- Plus(q, d) =
//     [cos|d|, sin|d| d/|d|] * q (CS/internal/ceres/local_parameterization.cc:164-201) -- c_translation[i] and the
//     triangulated points; frame l's rotation and the translations of frames l and frame_num-1 are constant (:244-251);
//   * residual: ReprojectionError3D (inital_sfm.hpp:25-54), no loss; its Jacobian over the local rotation step is
//     -2 [R X]x (what the automatic derivative times the parameterization's 4x3 Jacobian evaluates to);
//   * trust region: Levenberg-Marquardt (CS/internal/ceres/levenberg_marquardt_strategy.cc:66-163) with Jacobi scaling,
//     the loop and termination tests of CS/internal/ceres/trust_region_minimizer.cc:65-123,666-704 with the Solver's
//     defaults (50 iterations, function 1e-6, gradient 1e-10, parameter 1e-8); the 0.3 s wall-clock limit is not
//     modelled (a problem of this size takes milliseconds);
//   * linear solver: DENSE_SCHUR, the points being the e-blocks (CS/internal/ceres/schur_complement_solver.cc:161-224).
struct BaObs {
  int frame, point;
  double u, v;
};

struct BaSystem {  // JtJ and Jt r of the whole problem at one iterate, local (tangent) coordinates, unscaled
  std::vector<double> Hpp, Hcp, Hcc, gp, gc;
};

struct BaLayout {
  int frame_num = 0, np = 0, nc = 0;
  std::vector<int> off_q, off_t;  // column of each camera block in the reduced system, -1 = constant
  std::vector<BaObs> obs;
};

double ba_evaluate(const BaLayout &L, const std::vector<double> &cq, const std::vector<double> &ct, const std::vector<double> &X,
                   BaSystem *sys) {
  const int np = L.np, nc = L.nc;
  if (sys) {
    sys->Hpp.assign(9 * (size_t)np, 0.0), sys->Hcp.assign((size_t)nc * 3 * np, 0.0), sys->Hcc.assign((size_t)nc * nc, 0.0);
    sys->gp.assign(3 * (size_t)np, 0.0), sys->gc.assign(nc, 0.0);
  }
  std::vector<double> R(9 * (size_t)L.frame_num);
  for (int i = 0; i < L.frame_num; i++)  // QuaternionRotatePoint normalizes its quaternion (CS/include/ceres/rotation.h:553-571)
    qtoR(qnormalized(Quat{cq[4 * i + 1], cq[4 * i + 2], cq[4 * i + 3], cq[4 * i]}), &R[9 * (size_t)i]);
  double cost = 0.0;
  for (const BaObs &o : L.obs) {
    const double *Ri = &R[9 * (size_t)o.frame], *Xw = &X[3 * (size_t)o.point];
    double RX[3], Y[3];
    mat3vec(Ri, Xw, RX);
    for (int k = 0; k < 3; k++) Y[k] = RX[k] + ct[3 * o.frame + k];
    const double iz = 1.0 / Y[2], r[2] = {Y[0] * iz - o.u, Y[1] * iz - o.v};
    cost += r[0] * r[0] + r[1] * r[1];
    if (!sys) continue;
    const double Jp[6] = {iz, 0, -Y[0] * iz * iz, 0, iz, -Y[1] * iz * iz};
    double S[9], Jq[6], JX[6];
    skew3(RX, S);
    for (int a = 0; a < 2; a++)
      for (int c = 0; c < 3; c++) {
        Jq[a * 3 + c] = -2.0 * (Jp[a * 3] * S[c] + Jp[a * 3 + 1] * S[3 + c] + Jp[a * 3 + 2] * S[6 + c]);
        JX[a * 3 + c] = Jp[a * 3] * Ri[c] + Jp[a * 3 + 1] * Ri[3 + c] + Jp[a * 3 + 2] * Ri[6 + c];
      }
    int cols[6], ncol = 0;
    double Jc[12];
    if (L.off_q[o.frame] >= 0)
      for (int c = 0; c < 3; c++) cols[ncol] = L.off_q[o.frame] + c, Jc[ncol] = Jq[c], Jc[6 + ncol] = Jq[3 + c], ncol++;
    if (L.off_t[o.frame] >= 0)
      for (int c = 0; c < 3; c++) cols[ncol] = L.off_t[o.frame] + c, Jc[ncol] = Jp[c], Jc[6 + ncol] = Jp[3 + c], ncol++;
    for (int a = 0; a < ncol; a++) {
      sys->gc[cols[a]] += Jc[a] * r[0] + Jc[6 + a] * r[1];
      for (int b = 0; b < ncol; b++) sys->Hcc[(size_t)cols[a] * nc + cols[b]] += Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
      for (int c = 0; c < 3; c++) sys->Hcp[(size_t)cols[a] * 3 * np + 3 * o.point + c] += Jc[a] * JX[c] + Jc[6 + a] * JX[3 + c];
    }
    for (int a = 0; a < 3; a++) {
      sys->gp[3 * (size_t)o.point + a] += JX[a] * r[0] + JX[3 + a] * r[1];
      for (int b = 0; b < 3; b++) sys->Hpp[9 * (size_t)o.point + a * 3 + b] += JX[a] * JX[b] + JX[3 + a] * JX[3 + b];
    }
  }
  return 0.5 * cost;
}

// QuaternionParameterization::Plus on (w x y z)
void ba_quat_plus(const double q[4], const double d[3], double out[4]) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nd > 0.0) {
    const double k = sin(nd) / nd;
    const Quat dq{k * d[0], k * d[1], k * d[2], cos(nd)}, r = qmul(dq, Quat{q[1], q[2], q[3], q[0]});
    out[0] = r.w, out[1] = r.x, out[2] = r.y, out[3] = r.z;
  } else {
    memcpy(out, q, 32);
  }
}

// (Js^T Js + D^2) y = gs by elimination of the points (3x3 blocks), Cholesky on the camera system: y = [yp | yc].
// Hs = S H S with the Jacobi column scaling S; d2 = D^2 (both in [points | cameras] order).
bool ba_solve(const BaLayout &L, const BaSystem &sys, const std::vector<double> &scale, const std::vector<double> &d2,
              std::vector<double> &y) {
  const int np = L.np, nc = L.nc, n3 = 3 * np;
  const double *sp = scale.data(), *sc = scale.data() + n3;
  std::vector<double> Einv(9 * (size_t)np), S((size_t)nc * nc), rhs(nc);
  for (int p = 0; p < np; p++) {
    double M[9];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) M[a * 3 + b] = sp[3 * p + a] * sys.Hpp[9 * (size_t)p + a * 3 + b] * sp[3 * p + b];
    for (int a = 0; a < 3; a++) M[a * 3 + a] += d2[3 * p + a];
    // inverse through the Cholesky factor of the 3x3 block (schur_eliminator_impl.h:258-262 InvertPSDMatrix)
    const double l00 = sqrt(M[0]), l10 = M[3] / l00, l20 = M[6] / l00;
    const double l11 = sqrt(M[4] - l10 * l10), l21 = (M[7] - l20 * l10) / l11, l22 = sqrt(M[8] - l20 * l20 - l21 * l21);
    if (!(l00 > 0.0) || !(l11 > 0.0) || !(l22 > 0.0)) return false;
    const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
    const double i10 = -l10 * i00 * i11, i21 = -l21 * i11 * i22, i20 = -(l20 * i00 + l21 * i10) * i22;
    double *I = &Einv[9 * (size_t)p];  // L^-T L^-1
    I[0] = i00 * i00 + i10 * i10 + i20 * i20, I[1] = I[3] = i10 * i11 + i20 * i21, I[2] = I[6] = i20 * i22;
    I[4] = i11 * i11 + i21 * i21, I[5] = I[7] = i21 * i22, I[8] = i22 * i22;
  }
  std::vector<double> W((size_t)nc * n3), Hs((size_t)nc * n3);  // scaled Hcp and Hcp E^-1
  for (int a = 0; a < nc; a++)
    for (int k = 0; k < n3; k++) Hs[(size_t)a * n3 + k] = sc[a] * sys.Hcp[(size_t)a * n3 + k] * sp[k];
  for (int a = 0; a < nc; a++)
    for (int p = 0; p < np; p++) {
      const double *h = &Hs[(size_t)a * n3 + 3 * p], *I = &Einv[9 * (size_t)p];
      double *w = &W[(size_t)a * n3 + 3 * p];
      for (int c = 0; c < 3; c++) w[c] = h[0] * I[c] + h[1] * I[3 + c] + h[2] * I[6 + c];
    }
  for (int a = 0; a < nc; a++) {
    double acc = sc[a] * sys.gc[a];
    for (int k = 0; k < n3; k++) acc -= W[(size_t)a * n3 + k] * (sp[k] * sys.gp[k]);
    rhs[a] = acc;
    for (int b = 0; b <= a; b++) {
      double sum = sc[a] * sys.Hcc[(size_t)a * nc + b] * sc[b];
      const double *wa = &W[(size_t)a * n3], *hb = &Hs[(size_t)b * n3];
      for (int k = 0; k < n3; k++) sum -= wa[k] * hb[k];
      S[(size_t)a * nc + b] = S[(size_t)b * nc + a] = sum;
    }
    S[(size_t)a * nc + a] += d2[n3 + a];
  }
  // Eigen::LLT of the reduced camera matrix (schur_complement_solver.cc:201-213)
  for (int j = 0; j < nc; j++) {
    double d = S[(size_t)j * nc + j];
    for (int k = 0; k < j; k++) d -= S[(size_t)j * nc + k] * S[(size_t)j * nc + k];
    if (!(d > 0.0)) return false;
    d = sqrt(d), S[(size_t)j * nc + j] = d;
    for (int i = j + 1; i < nc; i++) {
      double v = S[(size_t)i * nc + j];
      for (int k = 0; k < j; k++) v -= S[(size_t)i * nc + k] * S[(size_t)j * nc + k];
      S[(size_t)i * nc + j] = v / d;
    }
  }
  y.assign((size_t)n3 + nc, 0.0);
  double *yc = y.data() + n3;
  for (int i = 0; i < nc; i++) {
    double v = rhs[i];
    for (int k = 0; k < i; k++) v -= S[(size_t)i * nc + k] * yc[k];
    yc[i] = v / S[(size_t)i * nc + i];
  }
  for (int i = nc - 1; i >= 0; i--) {
    double v = yc[i];
    for (int k = i + 1; k < nc; k++) v -= S[(size_t)k * nc + i] * yc[k];
    yc[i] = v / S[(size_t)i * nc + i];
  }
  for (int p = 0; p < np; p++) {  // back-substitution: yp = E^-1 (gs_p - Hs_pc yc)
    double v[3];
    for (int a = 0; a < 3; a++) {
      v[a] = sp[3 * p + a] * sys.gp[3 * (size_t)p + a];
      for (int c = 0; c < nc; c++) v[a] -= Hs[(size_t)c * n3 + 3 * p + a] * yc[c];
    }
    const double *I = &Einv[9 * (size_t)p];
    for (int a = 0; a < 3; a++) y[3 * (size_t)p + a] = I[a * 3] * v[0] + I[a * 3 + 1] * v[1] + I[a * 3 + 2] * v[2];
  }
  for (double v : y)
    if (!std::isfinite(v)) return false;
  return true;
}

}  // namespace

// cq [frame_num][4] (w x y z) / ct [frame_num][3]: world -> camera, in/out; the triangulated points of sfm_f in/out.
// Returns the reference's acceptance test (inital_sfm.cpp:279): CONVERGENCE or final cost < 3e-3.
bool bundle_adjust(int frame_num, int l, std::vector<double> &cq, std::vector<double> &ct, std::vector<SfmFeature> &sfm_f,
                   VioSolveStats *stats) {
  BaLayout L;
  L.frame_num = frame_num;
  std::vector<int> pidx;
  for (size_t j = 0; j < sfm_f.size(); j++) {
    if (!sfm_f[j].state) continue;
    for (const auto &o : sfm_f[j].observation) L.obs.push_back({o.first, (int)pidx.size(), o.second.first, o.second.second});
    pidx.push_back((int)j);
  }
  const int np = L.np = (int)pidx.size(), n3 = 3 * np;
  L.off_q.assign(frame_num, -1), L.off_t.assign(frame_num, -1);
  for (int i = 0; i < frame_num; i++) {
    if (i != l) L.off_q[i] = L.nc, L.nc += 3;
    if (i != l && i != frame_num - 1) L.off_t[i] = L.nc, L.nc += 3;
  }
  const int nc = L.nc, N = n3 + nc;
  std::vector<double> X(n3);
  for (int p = 0; p < np; p++) memcpy(&X[3 * (size_t)p], sfm_f[pidx[p]].position, 24);
  VioSolveStats st;
  memset(&st, 0, sizeof(st));
  auto record = [&](int it, double cost, double radius, double step_norm, double rho, double gmax, bool valid, bool ok) {
    st.iterations = it + 1;
    if (it < VIO_MAX_TRACE) {
      st.it_cost[it] = cost, st.it_radius[it] = radius, st.it_step_norm[it] = step_norm, st.it_relative_decrease[it] = rho;
      st.it_gradient_max_norm[it] = gmax, st.it_flags[it] = (valid ? 1 : 0) | (ok ? 2 : 0);
    }
  };
  BaSystem sys;
  auto x_norm_of = [&](const std::vector<double> &q, const std::vector<double> &t, const std::vector<double> &Xp) {
    double n2 = 0.0;
    for (double v : Xp) n2 += v * v;
    for (int i = 0; i < frame_num; i++) {
      if (L.off_q[i] >= 0)
        for (int k = 0; k < 4; k++) n2 += q[4 * i + k] * q[4 * i + k];
      if (L.off_t[i] >= 0)
        for (int k = 0; k < 3; k++) n2 += t[3 * i + k] * t[3 * i + k];
    }
    return sqrt(n2);
  };
  auto grad_max = [&]() {  // |Plus(x, -g) - x|_inf (trust_region_minimizer.cc:270-284)
    double m = 0.0;
    for (double v : sys.gp) m = std::max(m, fabs(v));
    for (int i = 0; i < frame_num; i++) {
      if (L.off_t[i] >= 0)
        for (int k = 0; k < 3; k++) m = std::max(m, fabs(sys.gc[L.off_t[i] + k]));
      if (L.off_q[i] >= 0) {
        const double d[3] = {-sys.gc[L.off_q[i]], -sys.gc[L.off_q[i] + 1], -sys.gc[L.off_q[i] + 2]};
        double qn[4];
        ba_quat_plus(&cq[4 * i], d, qn);
        for (int k = 0; k < 4; k++) m = std::max(m, fabs(qn[k] - cq[4 * i + k]));
      }
    }
    return m;
  };
  double x_cost = ba_evaluate(L, cq, ct, X, &sys), x_norm = -1.0;
  st.initial_cost = x_cost;
  std::vector<double> scale(N), diag(N), d2(N), y;
  for (int k = 0; k < n3; k++) scale[k] = 1.0 / (1.0 + sqrt(sys.Hpp[9 * (size_t)(k / 3) + 4 * (k % 3)]));  // Jacobi scaling (:239-254)
  for (int a = 0; a < nc; a++) scale[n3 + a] = 1.0 / (1.0 + sqrt(sys.Hcc[(size_t)a * nc + a]));
  double gmax = grad_max(), radius = 1e4, decrease_factor = 2.0;
  bool last_ok = true, reuse_diagonal = false;
  int termination = 0, invalid_run = 0, it = 0;
  st.num_successful_steps = 1;
  record(0, x_cost, radius, 0, 0, gmax, true, true);
  while (N > 0) {
    if (it >= 50) break;
    if (last_ok && gmax <= 1e-10) { termination = 1; break; }
    if (radius <= 1e-32) { termination = 1; break; }
    it++;
    if (!reuse_diagonal) {  // LevenbergMarquardtStrategy::ComputeStep (:79-89)
      for (int k = 0; k < n3; k++) diag[k] = std::min(std::max(scale[k] * scale[k] * sys.Hpp[9 * (size_t)(k / 3) + 4 * (k % 3)], 1e-6), 1e32);
      for (int a = 0; a < nc; a++) diag[n3 + a] = std::min(std::max(scale[n3 + a] * scale[n3 + a] * sys.Hcc[(size_t)a * nc + a], 1e-6), 1e32);
    }
    reuse_diagonal = true;
    for (int k = 0; k < N; k++) d2[k] = diag[k] / radius;  // lm_diagonal_^2 (:91)
    bool solver_ok = ba_solve(L, sys, scale, d2, y);
    double a = 0.0, b = 0.0;
    if (solver_ok)
      for (int k = 0; k < N; k++) {
        const double g = k < n3 ? sys.gp[k] : sys.gc[k - n3];
        a += y[k] * (scale[k] * g), b += d2[k] * y[k] * y[k];
      }
    const double model_cost_change = 0.5 * (a + b);  // step = -y; -step^T (gs + Hs step / 2) with (Hs + D^2) y = gs
    if (!(solver_ok && model_cost_change > 0.0)) {
      if (++invalid_run >= 5) { termination = 2; break; }
      radius = radius / decrease_factor, decrease_factor *= 2.0, reuse_diagonal = true;  // StepRejected (:155-159)
      last_ok = false, st.num_unsuccessful_steps++;
      record(it, x_cost, radius, 0, 0, gmax, false, false);
      continue;
    }
    invalid_run = 0;
    std::vector<double> cqn(cq), ctn(ct), Xn(X);
    double sn = 0.0;
    for (int k = 0; k < n3; k++) {
      const double d = -y[k] * scale[k];
      Xn[k] = X[k] + d, sn += (X[k] - Xn[k]) * (X[k] - Xn[k]);
    }
    for (int i = 0; i < frame_num; i++) {
      if (L.off_q[i] >= 0) {
        double d[3];
        for (int k = 0; k < 3; k++) d[k] = -y[n3 + L.off_q[i] + k] * scale[n3 + L.off_q[i] + k];
        ba_quat_plus(&cq[4 * i], d, &cqn[4 * i]);
        for (int k = 0; k < 4; k++) sn += (cq[4 * i + k] - cqn[4 * i + k]) * (cq[4 * i + k] - cqn[4 * i + k]);
      }
      if (L.off_t[i] >= 0)
        for (int k = 0; k < 3; k++) {
          const double d = -y[n3 + L.off_t[i] + k] * scale[n3 + L.off_t[i] + k];
          ctn[3 * i + k] = ct[3 * i + k] + d, sn += (ct[3 * i + k] - ctn[3 * i + k]) * (ct[3 * i + k] - ctn[3 * i + k]);
        }
    }
    double cand_cost = ba_evaluate(L, cqn, ctn, Xn, nullptr);
    if (!std::isfinite(cand_cost)) cand_cost = 1.7976931348623157e308;
    const double step_norm = sqrt(sn);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { termination = 1; break; }       // ParameterToleranceReached (:666-685)
    if (fabs(x_cost - cand_cost) <= 1e-6 * x_cost) { termination = 1; break; }  // FunctionToleranceReached (:687-704)
    const double rho = (x_cost - cand_cost) / model_cost_change;  // monotonic steps: the step evaluator's reference is x_cost
    if (rho > 1e-3) {
      cq.swap(cqn), ct.swap(ctn), X.swap(Xn);
      x_norm = x_norm_of(cq, ct, X);
      x_cost = ba_evaluate(L, cq, ct, X, &sys);
      gmax = grad_max();
      const double q3 = 2.0 * rho - 1.0;
      radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - q3 * q3 * q3));  // StepAccepted (:146-153)
      decrease_factor = 2.0, reuse_diagonal = false, last_ok = true, st.num_successful_steps++;
      record(it, x_cost, radius, step_norm, rho, gmax, true, true);
    } else {
      radius = radius / decrease_factor, decrease_factor *= 2.0, reuse_diagonal = true;  // StepRejected (:155-159)
      last_ok = false, st.num_unsuccessful_steps++;
      record(it, cand_cost, radius, step_norm, rho, 0.0, true, false);
    }
  }
  for (int p = 0; p < np; p++) memcpy(sfm_f[pidx[p]].position, &X[3 * (size_t)p], 24);
  st.final_cost = x_cost, st.termination = termination;
  if (stats) *stats = st;
  return termination == 1 || x_cost < 3e-03;
}

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
  // 5: full BA over (c_rotation = Quaterniond(c_Rotation[i]) as w x y z, c_translation)
  std::vector<double> cq(4 * (size_t)frame_num);
  for (int i = 0; i < frame_num; i++) {
    const Quat qi = RtoQ(&Rc[9 * i]);
    cq[4 * i] = qi.w, cq[4 * i + 1] = qi.x, cq[4 * i + 2] = qi.y, cq[4 * i + 3] = qi.z;
  }
  if (!bundle_adjust(frame_num, l, cq, tc, sfm_f, nullptr)) return false;  // "vision only BA not converge"
  for (int i = 0; i < frame_num; i++) {
    const Quat qi = qinv(Quat{cq[4 * i + 1], cq[4 * i + 2], cq[4 * i + 3], cq[4 * i]});  // q[i] = c_rotation^-1 (:287-295)
    double v[3];
    q[4 * i] = qi.x, q[4 * i + 1] = qi.y, q[4 * i + 2] = qi.z, q[4 * i + 3] = qi.w;
    qrot(qi, &tc[3 * i], v);
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

// mode 0: cv::findEssentialMat + cv::recoverPose restated (five-point RANSAC; the reference's semantics); mode 1: the fit
// over all correspondences (vio_init_relative_pose with its rotation hint).
extern "C" int vio_init_relative_pose_mode(const double *xy0, const double *xy1, int32_t n, int32_t mode, const double *R_hint,
                                           double R[9], double t[3], int32_t *inliers, int32_t *ok) {
  if (!xy0 || !xy1 || n < 0 || !R || !t || !ok || mode < 0 || mode > 1) return VIO_EINVAL;
  std::vector<double> a(xy0, xy0 + 2 * (size_t)n), b(xy1, xy1 + 2 * (size_t)n);
  int in = 0;
  *ok = (mode == 0 ? init::solve_relative_rt_five_point(a, b, R, t, &in) : init::solve_relative_rt(a, b, R, t, &in, R_hint)) ? 1 : 0;
  if (inliers) *inliers = in;
  return VIO_OK;
}

extern "C" int vio_init_recover_pose(const double E[9], const double *xy0, const double *xy1, int32_t n, double R[9], double t[3],
                                     int32_t *inliers) {
  if (!E || !xy0 || !xy1 || n < 0 || !R || !t || !inliers) return VIO_EINVAL;
  *inliers = init::recover_pose(E, xy0, xy1, n, R, t);
  return VIO_OK;
}

extern "C" int vio_init_five_point(const double *xy0, const double *xy1, double *E, int32_t *n_models) {
  if (!xy0 || !xy1 || !E || !n_models) return VIO_EINVAL;
  double q1[5][2], q2[5][2], Em[10][9];
  for (int i = 0; i < 5; i++) q1[i][0] = xy0[2 * i], q1[i][1] = xy0[2 * i + 1], q2[i][0] = xy1[2 * i], q2[i][1] = xy1[2 * i + 1];
  *n_models = init::five_point_kernel(q1, q2, Em);
  memcpy(E, Em, sizeof(double) * 9 * (size_t)*n_models);
  return VIO_OK;
}

extern "C" int vio_init_pnp(const double *pts3, const double *pts2, int32_t n, double R[9], double t[3], int32_t *ok) {
  if (!pts3 || !pts2 || n < 0 || !R || !t || !ok) return VIO_EINVAL;
  std::vector<double> p3(pts3, pts3 + 3 * (size_t)n), p2(pts2, pts2 + 2 * (size_t)n);
  *ok = init::pnp_refine(p3, p2, R, t) ? 1 : 0;
  return VIO_OK;
}

extern "C" int vio_init_triangulate_point(const double pose0[12], const double pose1[12], const double xy0[2], const double xy1[2],
                                          double point[3]) {
  if (!pose0 || !pose1 || !xy0 || !xy1 || !point) return VIO_EINVAL;
  init::triangulate_point(pose0, pose1, xy0, xy1, point);
  return VIO_OK;
}

extern "C" int vio_init_bundle_adjust(int32_t frame_num, int32_t l, double *c_rotation, double *c_translation, int32_t n_points,
                                      double *points, const uint8_t *point_ok, const int32_t *feat_start, const int32_t *obs_frame,
                                      const double *obs_xy, VioSolveStats *stats, int32_t *ok) {
  if (frame_num < 2 || l < 0 || l >= frame_num || !c_rotation || !c_translation || n_points < 0 || !points || !point_ok ||
      !feat_start || !obs_frame || !obs_xy || !ok)
    return VIO_EINVAL;
  std::vector<init::SfmFeature> f(n_points);
  for (int j = 0; j < n_points; j++) {
    f[j].id = j, f[j].state = point_ok[j] != 0;
    memcpy(f[j].position, points + 3 * (size_t)j, 24);
    for (int k = feat_start[j]; k < feat_start[j + 1]; k++) {
      if (obs_frame[k] < 0 || obs_frame[k] >= frame_num) return VIO_EINVAL;
      f[j].observation.push_back({obs_frame[k], {obs_xy[2 * k], obs_xy[2 * k + 1]}});
    }
  }
  std::vector<double> cq(c_rotation, c_rotation + 4 * (size_t)frame_num), ct(c_translation, c_translation + 3 * (size_t)frame_num);
  *ok = init::bundle_adjust(frame_num, l, cq, ct, f, stats) ? 1 : 0;
  memcpy(c_rotation, cq.data(), sizeof(double) * cq.size()), memcpy(c_translation, ct.data(), sizeof(double) * ct.size());
  for (int j = 0; j < n_points; j++) memcpy(points + 3 * (size_t)j, f[j].position, 24);
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
