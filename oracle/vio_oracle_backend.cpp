// oracle/vio_oracle_backend.cpp — TEST INFRASTRUCTURE ONLY (see vio_oracle.h).
//
// CPU restatement of VINS::solve_ceres (VINS_ios/VINS.cpp:480-831) in plain C++:
//   factors      ProjectionFactor::Evaluate            VINS_ios/projection_facor.cpp:16-99
//                IMUFactor::Evaluate                   VINS_ios/imu_factor.h:27-184
//                IntegrationBase                       VINS_ios/integration_base.h:63-198
//                MarginalizationFactor::Evaluate       VINS_ios/marginalization_factor.cpp:336-384
//                PoseLocalParameterization::Plus       VINS_ios/pose_local_parameterization.cpp:11-27
//   robust loss  CauchyLoss + Corrector                CSI/loss_function.cc:72-79, CSI/corrector.cc:40-158
//   minimizer    TrustRegionMinimizer                  CSI/trust_region_minimizer.cc:66-786
//                DoglegStrategy (TRADITIONAL)          CSI/dogleg_strategy.cc:77-255,515-635
//                DENSE_SCHUR + Eigen LLT               CSI/schur_complement_solver.cc:123-213,
//                                                      CSI/schur_eliminator_impl.h:173-365
//   gauge fix    VINS::new2old                         VINS_ios/VINS.cpp:131-212
//   prior        MarginalizationInfo::marginalize      VINS_ios/marginalization_factor.cpp:182-300,
//                call site                             VINS_ios/VINS.cpp:690-830
// (CSI = /root/reference/VINS_ThirdPartyLib/ceres-solver/internal/ceres)
//
// Design notes. Ceres works on an explicit scaled Jacobian J_s = J*diag(scale). Everything the minimizer
// needs from it is a function of H = J^T J, g = J^T r and the residual cost, so this restatement (and the HIP
// kernel that mirrors it) accumulates H and g directly from the factors and never stores J:
//   ||J_s[:,c]||^2 = scale_c^2 H_cc,  J_s^T r = scale*g,  ||J_s v||^2 = v^T (S H S) v,
//   model_cost_change = -(J_s d)^T (r + J_s d/2) = -d^T (S g) - d^T (S H S) d / 2.
// The elimination set is "all features" (Ceres additionally eliminates an independent set of speed-bias
// blocks, CSI/parameter_block_ordering.cc:50-78); any exact solve of the same regularized normal equations
// gives the same step up to rounding.

#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "vio_oracle.h"

namespace {

typedef std::vector<double> Vec;

// ---------------------------------------------------------------------------------------------------
// small fixed-size math (quaternions stored x y z w like para_Pose)
struct Q {
  double x, y, z, w;
};
inline Q qmul(const Q &a, const Q &b) {
  return Q{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
           a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Q qinv(const Q &q) {  // Eigen QuaternionBase::inverse(): conjugate / squaredNorm
  double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  return Q{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
}
inline Q qnormalized(const Q &q) {
  double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Q{q.x / n, q.y / n, q.z / n, q.w / n};
}
inline void qrot(const Q &q, const double v[3], double out[3]) {  // Eigen _transformVector
  double ux = 2 * (q.y * v[2] - q.z * v[1]), uy = 2 * (q.z * v[0] - q.x * v[2]), uz = 2 * (q.x * v[1] - q.y * v[0]);
  out[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
  out[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
  out[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}
inline void qtoR(const Q &q, double R[9]) {  // Eigen toRotationMatrix
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
  R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
  R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
}
inline Q RtoQ(const double R[9]) {  // Eigen quaternion-from-matrix (Quaternion.h, Shoemake)
  Q q;
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R[7] - R[5]) * t, q.y = (R[2] - R[6]) * t, q.z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    q.x = v[0], q.y = v[1], q.z = v[2];
  }
  return q;
}
inline Q qfrom(const double *p) { return Q{p[3], p[4], p[5], p[6]}; }  // pose block -> quaternion
inline void mat3mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
inline void mat3T(const double A[9], double T[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[i * 3 + j] = A[j * 3 + i];
}
inline void mat3vec(const double A[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; i++) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
inline void skew(const double v[3], double S[9]) {
  S[0] = 0, S[1] = -v[2], S[2] = v[1], S[3] = v[2], S[4] = 0, S[5] = -v[0], S[6] = -v[1], S[7] = v[0], S[8] = 0;
}
// bottom-right 3x3 of Utility::Qleft(q) / Qright(q) (utility.hpp:57-74)
inline void qleft33(const Q &q, double M[9]) {
  double v[3] = {q.x, q.y, q.z};
  skew(v, M);
  M[0] += q.w, M[4] += q.w, M[8] += q.w;
}
// bottom-right 3x3 of Qleft(a)*Qright(b)
inline void qleft_qright33(const Q &a, const Q &b, double M[9]) {
  double va[3] = {a.x, a.y, a.z}, vb[3] = {b.x, b.y, b.z};
  double L[9], Rm[9], S[9];
  skew(va, L);
  L[0] += a.w, L[4] += a.w, L[8] += a.w;
  skew(vb, S);
  for (int i = 0; i < 9; i++) Rm[i] = -S[i];
  Rm[0] += b.w, Rm[4] += b.w, Rm[8] += b.w;
  mat3mul(L, Rm, M);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[i * 3 + j] += va[i] * (-vb[j]);
}

// Utility::R2ypr / ypr2R (utility.hpp:76-118), degrees
inline void R2ypr(const double R[9], double ypr[3]) {
  double n[3] = {R[0], R[3], R[6]}, o[3] = {R[1], R[4], R[7]}, a[3] = {R[2], R[5], R[8]};
  double y = atan2(n[1], n[0]);
  double p = atan2(-n[2], n[0] * cos(y) + n[1] * sin(y));
  double r = atan2(a[0] * sin(y) - a[1] * cos(y), -o[0] * sin(y) + o[1] * cos(y));
  ypr[0] = y / M_PI * 180.0, ypr[1] = p / M_PI * 180.0, ypr[2] = r / M_PI * 180.0;
}
inline void ypr2R(const double ypr[3], double R[9]) {
  double y = ypr[0] / 180.0 * M_PI, p = ypr[1] / 180.0 * M_PI, r = ypr[2] / 180.0 * M_PI;
  double Rz[9] = {cos(y), -sin(y), 0, sin(y), cos(y), 0, 0, 0, 1};
  double Ry[9] = {cos(p), 0., sin(p), 0., 1., 0., -sin(p), 0., cos(p)};
  double Rx[9] = {1., 0., 0., 0., cos(r), -sin(r), 0., sin(r), cos(r)};
  double T[9];
  mat3mul(Rz, Ry, T);
  mat3mul(T, Rx, R);
}

// ---------------------------------------------------------------------------------------------------
// dense helpers (row-major)
// Cholesky A = L L^T in place (lower). Returns false when a pivot is <= 0 (Eigen LLT NumericalIssue,
// EIG/Eigen/src/Cholesky/LLT.h:300-312).
bool cholesky_lower(double *A, int n) {
  for (int k = 0; k < n; k++) {
    double x = A[k * n + k];
    for (int p = 0; p < k; p++) x -= A[k * n + p] * A[k * n + p];
    if (!(x > 0.0)) return false;
    x = std::sqrt(x);
    A[k * n + k] = x;
    for (int i = k + 1; i < n; i++) {
      double s = A[i * n + k];
      for (int p = 0; p < k; p++) s -= A[i * n + p] * A[k * n + p];
      A[i * n + k] = s / x;
    }
  }
  return true;
}
void chol_solve(const double *L, int n, double *b) {
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int p = 0; p < i; p++) s -= L[i * n + p] * b[p];
    b[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int p = i + 1; p < n; p++) s -= L[p * n + i] * b[p];
    b[i] = s / L[i * n + i];
  }
}
// inverse by Gauss-Jordan with partial pivoting (stands in for Eigen's PartialPivLU-based inverse())
bool invert(const double *A, int n, double *Ainv) {
  Vec M(n * 2 * n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) M[i * 2 * n + j] = A[i * n + j], M[i * 2 * n + n + j] = (i == j);
  for (int c = 0; c < n; c++) {
    int piv = c;
    for (int r = c + 1; r < n; r++)
      if (fabs(M[r * 2 * n + c]) > fabs(M[piv * 2 * n + c])) piv = r;
    if (M[piv * 2 * n + c] == 0.0) return false;
    if (piv != c)
      for (int j = 0; j < 2 * n; j++) std::swap(M[c * 2 * n + j], M[piv * 2 * n + j]);
    double d = M[c * 2 * n + c];
    for (int j = 0; j < 2 * n; j++) M[c * 2 * n + j] /= d;
    for (int r = 0; r < n; r++)
      if (r != c) {
        double f = M[r * 2 * n + c];
        if (f != 0.0)
          for (int j = 0; j < 2 * n; j++) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
      }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Ainv[i * n + j] = M[i * 2 * n + n + j];
  return true;
}
// Symmetric eigendecomposition by cyclic Jacobi: A = V diag(w) V^T, V columns = eigenvectors.
// (stands in for Eigen::SelfAdjointEigenSolver, marginalization_factor.cpp:268,286)
void jacobi_eig(const double *Ain, int n, double *w, double *V) {
  Vec A(Ain, Ain + n * n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) (i == j ? diag : off) += A[i * n + j] * A[i * n + j];
    if (off <= 1e-30 * diag || off == 0.0) break;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        double apq = A[p * n + q];
        if (apq == 0.0) continue;
        double app = A[p * n + p], aqq = A[q * n + q];
        double theta = (aqq - app) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; k++) {
          double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}

// ---------------------------------------------------------------------------------------------------
// IntegrationBase (integration_base.h)
struct Integration {
  double acc_0[3], gyr_0[3];
  double ba[3], bg[3];
  double sum_dt;
  double dp[3], dv[3];
  Q dq;
  double J[225], C[225];
  double noise[18];  // diagonal of the 18x18 noise matrix
};

void integration_init(Integration &ib, const VioConfig *cfg, const double *a0, const double *g0, const double *ba,
                      const double *bg) {
  memcpy(ib.acc_0, a0, 24), memcpy(ib.gyr_0, g0, 24), memcpy(ib.ba, ba, 24), memcpy(ib.bg, bg, 24);
  ib.sum_dt = 0;
  for (int k = 0; k < 3; k++) ib.dp[k] = ib.dv[k] = 0;
  ib.dq = Q{0, 0, 0, 1};
  for (int i = 0; i < 225; i++) ib.J[i] = (i % 16 == 0), ib.C[i] = 0;
  double an = cfg->acc_n * cfg->acc_n, gn = cfg->gyr_n * cfg->gyr_n, aw = cfg->acc_w * cfg->acc_w,
         gw = cfg->gyr_w * cfg->gyr_w;
  for (int k = 0; k < 3; k++)
    ib.noise[k] = an, ib.noise[3 + k] = gn, ib.noise[6 + k] = an, ib.noise[9 + k] = gn, ib.noise[12 + k] = aw,
    ib.noise[15 + k] = gw;
}

inline void set33(double *M, int ld, int r0, int c0, const double B[9], double s = 1.0) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[(r0 + i) * ld + c0 + j] = s * B[i * 3 + j];
}

// propagate() = midPointIntegration + state update (integration_base.h:63-169)
void integration_propagate(Integration &ib, double dt, const double *acc_1, const double *gyr_1) {
  double a0b[3], a1b[3], un_gyr[3];
  for (int k = 0; k < 3; k++) {
    a0b[k] = ib.acc_0[k] - ib.ba[k];
    a1b[k] = acc_1[k] - ib.ba[k];
    un_gyr[k] = 0.5 * (ib.gyr_0[k] + gyr_1[k]) - ib.bg[k];
  }
  double un_acc_0[3], un_acc_1[3];
  qrot(ib.dq, a0b, un_acc_0);
  Q rq = qmul(ib.dq, Q{un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2, 1.0});
  qrot(rq, a1b, un_acc_1);
  double un_acc[3], rp[3], rv[3];
  for (int k = 0; k < 3; k++) {
    un_acc[k] = 0.5 * (un_acc_0[k] + un_acc_1[k]);
    rp[k] = ib.dp[k] + ib.dv[k] * dt + 0.5 * un_acc[k] * dt * dt;
    rv[k] = ib.dv[k] + un_acc[k] * dt;
  }
  // jacobian / covariance update
  double Rw[9], Ra0[9], Ra1[9], R0[9], R1[9];
  skew(un_gyr, Rw), skew(a0b, Ra0), skew(a1b, Ra1);
  qtoR(ib.dq, R0), qtoR(rq, R1);
  double ImRw[9];  // I - R_w_x * dt
  for (int i = 0; i < 9; i++) ImRw[i] = (i % 4 == 0) - Rw[i] * dt;
  double R0Ra0[9], R1Ra1[9], R1Ra1I[9];
  mat3mul(R0, Ra0, R0Ra0), mat3mul(R1, Ra1, R1Ra1), mat3mul(R1Ra1, ImRw, R1Ra1I);
  double F[225] = {0}, V[15 * 18] = {0};
  double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, T[9];
  set33(F, 15, 0, 0, I3);
  for (int i = 0; i < 9; i++) T[i] = -0.25 * R0Ra0[i] * dt * dt + -0.25 * R1Ra1I[i] * dt * dt;
  set33(F, 15, 0, 3, T);
  set33(F, 15, 0, 6, I3, dt);
  for (int i = 0; i < 9; i++) T[i] = -0.25 * (R0[i] + R1[i]) * dt * dt;
  set33(F, 15, 0, 9, T);
  for (int i = 0; i < 9; i++) T[i] = -0.25 * R1Ra1[i] * dt * dt * -dt;
  set33(F, 15, 0, 12, T);
  set33(F, 15, 3, 3, ImRw);
  set33(F, 15, 3, 12, I3, -1.0 * dt);
  for (int i = 0; i < 9; i++) T[i] = -0.5 * R0Ra0[i] * dt + -0.5 * R1Ra1I[i] * dt;
  set33(F, 15, 6, 3, T);
  set33(F, 15, 6, 6, I3);
  for (int i = 0; i < 9; i++) T[i] = -0.5 * (R0[i] + R1[i]) * dt;
  set33(F, 15, 6, 9, T);
  for (int i = 0; i < 9; i++) T[i] = -0.5 * R1Ra1[i] * dt * -dt;
  set33(F, 15, 6, 12, T);
  set33(F, 15, 9, 9, I3);
  set33(F, 15, 12, 12, I3);

  set33(V, 18, 0, 0, R0, 0.25 * dt * dt);
  for (int i = 0; i < 9; i++) T[i] = 0.25 * -R1Ra1[i] * dt * dt * 0.5 * dt;
  set33(V, 18, 0, 3, T);
  set33(V, 18, 0, 9, T);
  set33(V, 18, 0, 6, R1, 0.25 * dt * dt);
  set33(V, 18, 3, 3, I3, 0.5 * dt);
  set33(V, 18, 3, 9, I3, 0.5 * dt);
  set33(V, 18, 6, 0, R0, 0.5 * dt);
  for (int i = 0; i < 9; i++) T[i] = 0.5 * -R1Ra1[i] * dt * 0.5 * dt;
  set33(V, 18, 6, 3, T);
  set33(V, 18, 6, 9, T);
  set33(V, 18, 6, 6, R1, 0.5 * dt);
  set33(V, 18, 9, 12, I3, dt);
  set33(V, 18, 12, 15, I3, dt);

  double Jn[225], FC[225], Cn[225];
  for (int i = 0; i < 15; i++)
    for (int j = 0; j < 15; j++) {
      double s = 0, t = 0;
      for (int k = 0; k < 15; k++) s += F[i * 15 + k] * ib.J[k * 15 + j], t += F[i * 15 + k] * ib.C[k * 15 + j];
      Jn[i * 15 + j] = s, FC[i * 15 + j] = t;
    }
  for (int i = 0; i < 15; i++)
    for (int j = 0; j < 15; j++) {
      double s = 0;
      for (int k = 0; k < 15; k++) s += FC[i * 15 + k] * F[j * 15 + k];
      double t = 0;
      for (int k = 0; k < 18; k++) t += V[i * 18 + k] * ib.noise[k] * V[j * 18 + k];
      Cn[i * 15 + j] = s + t;
    }
  memcpy(ib.J, Jn, sizeof(Jn)), memcpy(ib.C, Cn, sizeof(Cn));
  for (int k = 0; k < 3; k++) ib.dp[k] = rp[k], ib.dv[k] = rv[k];
  ib.dq = qnormalized(rq);
  ib.sum_dt += dt;
  memcpy(ib.acc_0, acc_1, 24), memcpy(ib.gyr_0, gyr_1, 24);
}

// ---------------------------------------------------------------------------------------------------
// factors. Jacobians are returned in LOCAL coordinates (pose 6, speed-bias 9, feature 1): the 7th column of every
// pose block is zero and PoseLocalParameterization::ComputeJacobian is [I6;0]
// (pose_local_parameterization.cpp:28-35), so J_local = first 6 columns.

// r[2]; Ji[2x6], Jj[2x6], Jex[2x6] (optional), Jl[2]
void projection_eval(double s_info, const double *pose_i, const double *pose_j, const double *ex, double inv_dep,
                     const double *pts_i, const double *pts_j, double *r, double *Ji, double *Jj, double *Jex,
                     double *Jl) {
  Q Qi = qfrom(pose_i), Qj = qfrom(pose_j), qic = qfrom(ex);
  double pc_i[3] = {pts_i[0] / inv_dep, pts_i[1] / inv_dep, pts_i[2] / inv_dep};
  double t[3], p_imu_i[3], p_w[3], p_imu_j[3], p_c_j[3];
  qrot(qic, pc_i, t);
  for (int k = 0; k < 3; k++) p_imu_i[k] = t[k] + ex[k];
  qrot(Qi, p_imu_i, t);
  for (int k = 0; k < 3; k++) p_w[k] = t[k] + pose_i[k];
  double d[3] = {p_w[0] - pose_j[0], p_w[1] - pose_j[1], p_w[2] - pose_j[2]};
  qrot(qinv(Qj), d, p_imu_j);
  double e[3] = {p_imu_j[0] - ex[0], p_imu_j[1] - ex[1], p_imu_j[2] - ex[2]};
  qrot(qinv(qic), e, p_c_j);
  double dep_j = p_c_j[2];
  r[0] = s_info * (p_c_j[0] / dep_j - pts_j[0]);
  r[1] = s_info * (p_c_j[1] / dep_j - pts_j[1]);
  if (!Ji) return;
  double Ri[9], Rj[9], ric[9], ricT[9], RjT[9];
  qtoR(Qi, Ri), qtoR(Qj, Rj), qtoR(qic, ric);
  mat3T(ric, ricT), mat3T(Rj, RjT);
  double red[6] = {s_info * (1. / dep_j), 0, s_info * (-p_c_j[0] / (dep_j * dep_j)),
                   0, s_info * (1. / dep_j), s_info * (-p_c_j[1] / (dep_j * dep_j))};
  double A[9], B[9], S[9], T9[9], jac3x6[18];
  mat3mul(ricT, RjT, A);  // ric^T Rj^T
  // pose i
  mat3mul(A, Ri, B);  // ric^T Rj^T Ri
  skew(p_imu_i, S);
  mat3mul(B, S, T9);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) jac3x6[i * 6 + j] = A[i * 3 + j], jac3x6[i * 6 + 3 + j] = -T9[i * 3 + j];
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 6; j++)
      Ji[i * 6 + j] = red[i * 3] * jac3x6[j] + red[i * 3 + 1] * jac3x6[6 + j] + red[i * 3 + 2] * jac3x6[12 + j];
  // pose j
  skew(p_imu_j, S);
  mat3mul(ricT, S, T9);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) jac3x6[i * 6 + j] = -A[i * 3 + j], jac3x6[i * 6 + 3 + j] = T9[i * 3 + j];
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 6; j++)
      Jj[i * 6 + j] = red[i * 3] * jac3x6[j] + red[i * 3 + 1] * jac3x6[6 + j] + red[i * 3 + 2] * jac3x6[12 + j];
  // extrinsic (projection_facor.cpp:76-86); needed by marginalization where ex_pose is a kept block
  if (Jex) {
    double RjTRi[9], C1[9], tmp_r[9];
    mat3mul(RjT, Ri, RjTRi);
    double M1[9];
    for (int i = 0; i < 9; i++) M1[i] = RjTRi[i] - (i % 4 == 0);
    mat3mul(ricT, M1, C1);
    mat3mul(B, ric, tmp_r);
    double Spc[9], T1[9], v[3], Sv[9], u[3], w3[3], Sw[9];
    skew(pc_i, Spc);
    mat3mul(tmp_r, Spc, T1);
    mat3vec(tmp_r, pc_i, v);
    skew(v, Sv);
    // ric^T (Rj^T (Ri tic + Pi - Pj) - tic)
    mat3vec(Ri, ex, u);
    for (int k = 0; k < 3; k++) u[k] += pose_i[k] - pose_j[k];
    mat3vec(RjT, u, w3);
    for (int k = 0; k < 3; k++) w3[k] -= ex[k];
    mat3vec(ricT, w3, u);
    skew(u, Sw);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        jac3x6[i * 6 + j] = C1[i * 3 + j], jac3x6[i * 6 + 3 + j] = -T1[i * 3 + j] + Sv[i * 3 + j] + Sw[i * 3 + j];
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 6; j++)
        Jex[i * 6 + j] = red[i * 3] * jac3x6[j] + red[i * 3 + 1] * jac3x6[6 + j] + red[i * 3 + 2] * jac3x6[12 + j];
  }
  // feature
  double Bric[9], v3[3];
  mat3mul(B, ric, Bric);
  mat3vec(Bric, pts_i, v3);
  for (int i = 0; i < 2; i++)
    Jl[i] = (red[i * 3] * v3[0] + red[i * 3 + 1] * v3[1] + red[i * 3 + 2] * v3[2]) * -1.0 / (inv_dep * inv_dep);
}

// sqrt_info = LLT(covariance.inverse()).matrixL().transpose()  (imu_factor.h:72) -> upper triangular U, U^T U = cov^-1
bool imu_sqrt_info(const double *cov, double *U) {
  double inv[225];
  if (!invert(cov, 15, inv)) return false;
  double L[225];
  memcpy(L, inv, sizeof(L));
  if (!cholesky_lower(L, 15)) return false;
  for (int i = 0; i < 15; i++)
    for (int j = 0; j < 15; j++) U[i * 15 + j] = (j >= i) ? L[j * 15 + i] : 0.0;
  return true;
}

// r[15]; Jpi[15x6], Jsi[15x9], Jpj[15x6], Jsj[15x9] (all whitened)
void imu_eval(const VioConfig *cfg, const VioPreintegration *pre, const double *U, const double *pose_i,
              const double *sb_i, const double *pose_j, const double *sb_j, double *r, double *Jpi, double *Jsi,
              double *Jpj, double *Jsj) {
  const double *Pi = pose_i, *Pj = pose_j, *Vi = sb_i, *Bai = sb_i + 3, *Bgi = sb_i + 6, *Vj = sb_j, *Baj = sb_j + 3,
               *Bgj = sb_j + 6;
  Q Qi = qfrom(pose_i), Qj = qfrom(pose_j);
  Q dq{pre->delta_q[0], pre->delta_q[1], pre->delta_q[2], pre->delta_q[3]};
  const double *J = pre->jacobian;
  double dp_dba[9], dp_dbg[9], dq_dbg[9], dv_dba[9], dv_dbg[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      dp_dba[i * 3 + j] = J[(0 + i) * 15 + 9 + j];
      dp_dbg[i * 3 + j] = J[(0 + i) * 15 + 12 + j];
      dq_dbg[i * 3 + j] = J[(3 + i) * 15 + 12 + j];
      dv_dba[i * 3 + j] = J[(6 + i) * 15 + 9 + j];
      dv_dbg[i * 3 + j] = J[(6 + i) * 15 + 12 + j];
    }
  double dba[3], dbg[3];
  for (int k = 0; k < 3; k++) dba[k] = Bai[k] - pre->linearized_ba[k], dbg[k] = Bgi[k] - pre->linearized_bg[k];
  double th[3];
  mat3vec(dq_dbg, dbg, th);
  Q cdq = qmul(dq, Q{th[0] / 2.0, th[1] / 2.0, th[2] / 2.0, 1.0});  // Utility::deltaQ: un-normalized
  double t1[3], t2[3], cdv[3], cdp[3];
  mat3vec(dv_dba, dba, t1), mat3vec(dv_dbg, dbg, t2);
  for (int k = 0; k < 3; k++) cdv[k] = pre->delta_v[k] + t1[k] + t2[k];
  mat3vec(dp_dba, dba, t1), mat3vec(dp_dbg, dbg, t2);
  for (int k = 0; k < 3; k++) cdp[k] = pre->delta_p[k] + t1[k] + t2[k];
  const double T = pre->sum_dt;
  const double G[3] = {0, 0, cfg->gravity};
  Q Qi_inv = qinv(Qi);
  double a[3], b[3], ra[3], rb[3];
  for (int k = 0; k < 3; k++) {
    a[k] = 0.5 * G[k] * T * T + Pj[k] - Pi[k] - Vi[k] * T;
    b[k] = G[k] * T + Vj[k] - Vi[k];
  }
  qrot(Qi_inv, a, ra), qrot(Qi_inv, b, rb);
  double res[15];
  Q qr = qmul(qinv(cdq), qmul(Qi_inv, Qj));
  for (int k = 0; k < 3; k++) {
    res[0 + k] = ra[k] - cdp[k];
    res[6 + k] = rb[k] - cdv[k];
    res[9 + k] = Baj[k] - Bai[k];
    res[12 + k] = Bgj[k] - Bgi[k];
  }
  res[3] = 2 * qr.x, res[4] = 2 * qr.y, res[5] = 2 * qr.z;
  for (int i = 0; i < 15; i++) {
    double s = 0;
    for (int k = i; k < 15; k++) s += U[i * 15 + k] * res[k];
    r[i] = s;
  }
  if (!Jpi) return;
  double Ri_invR[9];
  qtoR(Qi_inv, Ri_invR);
  double jpi[15 * 6] = {0}, jsi[15 * 9] = {0}, jpj[15 * 6] = {0}, jsj[15 * 9] = {0}, M[9], S[9];
  // pose i
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) jpi[(0 + i) * 6 + j] = -Ri_invR[i * 3 + j];
  skew(ra, S);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) jpi[(0 + i) * 6 + 3 + j] = S[i * 3 + j];
  qleft_qright33(qmul(qinv(Qj), Qi), cdq, M);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) jpi[(3 + i) * 6 + 3 + j] = -M[i * 3 + j];
  skew(rb, S);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) jpi[(6 + i) * 6 + 3 + j] = S[i * 3 + j];
  // speed-bias i
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      jsi[(0 + i) * 9 + 0 + j] = -Ri_invR[i * 3 + j] * T;
      jsi[(0 + i) * 9 + 3 + j] = -dp_dba[i * 3 + j];
      jsi[(0 + i) * 9 + 6 + j] = -dp_dbg[i * 3 + j];
      jsi[(6 + i) * 9 + 0 + j] = -Ri_invR[i * 3 + j];
      jsi[(6 + i) * 9 + 3 + j] = -dv_dba[i * 3 + j];
      jsi[(6 + i) * 9 + 6 + j] = -dv_dbg[i * 3 + j];
      jsi[(9 + i) * 9 + 3 + j] = -(i == j);
      jsi[(12 + i) * 9 + 6 + j] = -(i == j);
    }
  qleft33(qmul(qmul(qinv(Qj), Qi), cdq), M);
  double Mq[9];
  mat3mul(M, dq_dbg, Mq);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) jsi[(3 + i) * 9 + 6 + j] = -Mq[i * 3 + j];
  // pose j
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) jpj[(0 + i) * 6 + j] = Ri_invR[i * 3 + j];
  qleft33(qmul(qmul(qinv(cdq), Qi_inv), Qj), M);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) jpj[(3 + i) * 6 + 3 + j] = M[i * 3 + j];
  // speed-bias j
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      jsj[(6 + i) * 9 + 0 + j] = Ri_invR[i * 3 + j];
      jsj[(9 + i) * 9 + 3 + j] = (i == j);
      jsj[(12 + i) * 9 + 6 + j] = (i == j);
    }
  auto whiten = [&](const double *in, int cols, double *out) {
    for (int i = 0; i < 15; i++)
      for (int c = 0; c < cols; c++) {
        double s = 0;
        for (int k = i; k < 15; k++) s += U[i * 15 + k] * in[k * cols + c];
        out[i * cols + c] = s;
      }
  };
  whiten(jpi, 6, Jpi), whiten(jsi, 9, Jsi), whiten(jpj, 6, Jpj), whiten(jsj, 9, Jsj);
}

// dx of one prior block (marginalization_factor.cpp:349-367)
void prior_block_dx(int gsize, const double *x, const double *x0, double *dx) {
  if (gsize != 7) {
    for (int k = 0; k < gsize; k++) dx[k] = x[k] - x0[k];
    return;
  }
  for (int k = 0; k < 3; k++) dx[k] = x[k] - x0[k];
  Q q = qmul(qinv(qfrom(x0)), qfrom(x));
  double sgn = (q.w >= 0) ? 1.0 : -1.0;
  dx[3] = sgn * 2.0 * q.x, dx[4] = sgn * 2.0 * q.y, dx[5] = sgn * 2.0 * q.z;
}

// ---------------------------------------------------------------------------------------------------
// The window problem in reduced (local) coordinates.
//   pose-side ("f-block") layout: frame i -> [15 i, 15 i + 6) pose, [15 i + 6, 15 i + 15) speed-bias;
//   loop pose (if any) -> [15 P, 15 P + 6).  Features are the eliminated ("e") blocks.
struct Problem {
  const VioConfig *cfg;
  const VioWindow *w;
  int W, P, F, M;
  bool has_loop;
  int np;  // pose-side dimension
  double s_info;
  std::vector<double> U;  // W x 225 IMU sqrt_info (constant during the solve)
};
struct State {
  Vec pose;  // (P+1) x 7; row P = loop pose
  Vec sb;    // P x 9
  Vec feat;  // F
  double ex[7];
};
struct Lin {  // unscaled normal equations
  Vec Hpp;    // np x np
  Vec Hpf;    // F x np  (row f = w_f)
  Vec Hff;    // F
  Vec gp, gf;
};

inline int off_pose(const Problem &pb, int i) { return i == pb.P ? 15 * pb.P : 15 * i; }
inline int off_sb(int i) { return 15 * i + 6; }

inline void add_block(Vec &H, int n, int r0, int c0, const double *A, int lda, const double *B, int ldb, int rows,
                      int ra, int cb) {
  // H[r0:r0+ra, c0:c0+cb] += A^T B  (A rows x ra, B rows x cb)
  for (int i = 0; i < ra; i++)
    for (int j = 0; j < cb; j++) {
      double s = 0;
      for (int k = 0; k < rows; k++) s += A[k * lda + i] * B[k * ldb + j];
      H[(r0 + i) * n + c0 + j] += s;
    }
}

// Evaluates cost and, if lin != NULL, H and g.
double evaluate(const Problem &pb, const State &x, Lin *lin) {
  const VioWindow *w = pb.w;
  const int np = pb.np;
  double cost = 0;
  if (lin) {
    lin->Hpp.assign(np * np, 0.0), lin->Hpf.assign(pb.F * np, 0.0), lin->Hff.assign(pb.F, 0.0);
    lin->gp.assign(np, 0.0), lin->gf.assign(pb.F, 0.0);
  }
  // prior (MarginalizationFactor, no loss)
  if (w->prior) {
    const VioPrior *pr = w->prior;
    const int n = pr->n;
    Vec dx(n, 0.0), r(n);
    std::vector<int> col2par(n, -1);  // prior column -> reduced parameter index (-1: constant block)
    for (int b = 0; b < pr->n_blocks; b++) {
      int kind = pr->block_kind[b], idx = pr->block_index[b], o = pr->block_offset[b];
      const double *x0 = pr->block_x0 + 9 * b;
      if (kind == VIO_BLOCK_POSE) {
        prior_block_dx(7, &x.pose[7 * idx], x0, &dx[o]);
        for (int k = 0; k < 6; k++) col2par[o + k] = off_pose(pb, idx) + k;
      } else if (kind == VIO_BLOCK_SPEEDBIAS) {
        prior_block_dx(9, &x.sb[9 * idx], x0, &dx[o]);
        for (int k = 0; k < 9; k++) col2par[o + k] = off_sb(idx) + k;
      } else {
        prior_block_dx(7, x.ex, x0, &dx[o]);
      }
    }
    for (int i = 0; i < n; i++) {
      double s = pr->linearized_residuals[i];
      for (int j = 0; j < n; j++) s += pr->linearized_jacobians[i * n + j] * dx[j];
      r[i] = s;
      cost += 0.5 * s * s;
    }
    if (lin) {
      const double *J0 = pr->linearized_jacobians;
      for (int a = 0; a < n; a++) {
        if (col2par[a] < 0) continue;
        double g = 0;
        for (int k = 0; k < n; k++) g += J0[k * n + a] * r[k];
        lin->gp[col2par[a]] += g;
        for (int b = 0; b < n; b++) {
          if (col2par[b] < 0) continue;
          double s = 0;
          for (int k = 0; k < n; k++) s += J0[k * n + a] * J0[k * n + b];
          lin->Hpp[col2par[a] * np + col2par[b]] += s;
        }
      }
    }
  }
  // IMU factors (no loss)
  for (int i = 0; i < pb.W; i++) {
    int j = i + 1;
    double r[15], Jpi[90], Jsi[135], Jpj[90], Jsj[135];
    imu_eval(pb.cfg, &w->preint[i], &pb.U[225 * i], &x.pose[7 * i], &x.sb[9 * i], &x.pose[7 * j], &x.sb[9 * j], r,
             lin ? Jpi : NULL, Jsi, Jpj, Jsj);
    for (int k = 0; k < 15; k++) cost += 0.5 * r[k] * r[k];
    if (lin) {
      const double *Js[4] = {Jpi, Jsi, Jpj, Jsj};
      int offs[4] = {off_pose(pb, i), off_sb(i), off_pose(pb, j), off_sb(j)};
      int sz[4] = {6, 9, 6, 9};
      for (int a = 0; a < 4; a++) {
        for (int c = 0; c < sz[a]; c++) {
          double g = 0;
          for (int k = 0; k < 15; k++) g += Js[a][k * sz[a] + c] * r[k];
          lin->gp[offs[a] + c] += g;
        }
        for (int b = 0; b < 4; b++) add_block(lin->Hpp, np, offs[a], offs[b], Js[a], sz[a], Js[b], sz[b], 15, sz[a], sz[b]);
      }
    }
  }
  // projection factors with CauchyLoss(a): rho = b log(1 + s/b), b = a^2 (CSI/loss_function.cc:72-79)
  const double bb = pb.cfg->cauchy_a * pb.cfg->cauchy_a, cc = 1.0 / bb;
  for (int k = 0; k < pb.M; k++) {
    int h = w->factor_host[k], t = w->factor_target[k], f = w->factor_feature[k];
    double r[2], Ji[12], Jj[12], Jl[2];
    projection_eval(pb.s_info, &x.pose[7 * h], &x.pose[7 * t], x.ex, x.feat[f], w->factor_pts_i + 3 * k,
                    w->factor_pts_j + 3 * k, r, lin ? Ji : NULL, Jj, NULL, Jl);
    double sq = r[0] * r[0] + r[1] * r[1];
    double sum = 1.0 + sq * cc;
    double inv = 1.0 / sum;
    double rho0 = bb * std::log(sum), rho1 = inv > std::numeric_limits<double>::min() ? inv : std::numeric_limits<double>::min();
    cost += 0.5 * rho0;
    if (lin) {
      // rho'' = -c inv^2 < 0 always => Corrector: residual and Jacobian scaled by sqrt(rho') (CSI/corrector.cc:81-85)
      double sr = std::sqrt(rho1);
      for (int q = 0; q < 12; q++) Ji[q] *= sr, Jj[q] *= sr;
      Jl[0] *= sr, Jl[1] *= sr, r[0] *= sr, r[1] *= sr;
      int oi = off_pose(pb, h), oj = off_pose(pb, t);
      add_block(lin->Hpp, np, oi, oi, Ji, 6, Ji, 6, 2, 6, 6);
      add_block(lin->Hpp, np, oj, oj, Jj, 6, Jj, 6, 2, 6, 6);
      add_block(lin->Hpp, np, oi, oj, Ji, 6, Jj, 6, 2, 6, 6);
      add_block(lin->Hpp, np, oj, oi, Jj, 6, Ji, 6, 2, 6, 6);
      for (int c = 0; c < 6; c++) {
        lin->gp[oi + c] += Ji[c] * r[0] + Ji[6 + c] * r[1];
        lin->gp[oj + c] += Jj[c] * r[0] + Jj[6 + c] * r[1];
        lin->Hpf[f * np + oi + c] += Ji[c] * Jl[0] + Ji[6 + c] * Jl[1];
        lin->Hpf[f * np + oj + c] += Jj[c] * Jl[0] + Jj[6 + c] * Jl[1];
      }
      lin->Hff[f] += Jl[0] * Jl[0] + Jl[1] * Jl[1];
      lin->gf[f] += Jl[0] * r[0] + Jl[1] * r[1];
    }
  }
  return cost;
}

// PoseLocalParameterization::Plus on every block; delta in reduced layout [pose-side np | features F]
void plus(const Problem &pb, const State &x, const double *dp, const double *df, State &out) {
  out = x;
  auto pose_plus = [&](int i, const double *d) {
    double *p = &out.pose[7 * i];
    const double *p0 = &x.pose[7 * i];
    for (int k = 0; k < 3; k++) p[k] = p0[k] + d[k];
    Q q = qnormalized(qmul(qfrom(p0), Q{d[3] / 2.0, d[4] / 2.0, d[5] / 2.0, 1.0}));
    p[3] = q.x, p[4] = q.y, p[5] = q.z, p[6] = q.w;
  };
  for (int i = 0; i < pb.P; i++) {
    pose_plus(i, dp + off_pose(pb, i));
    for (int k = 0; k < 9; k++) out.sb[9 * i + k] = x.sb[9 * i + k] + dp[off_sb(i) + k];
  }
  if (pb.has_loop) pose_plus(pb.P, dp + off_pose(pb, pb.P));
  for (int f = 0; f < pb.F; f++) out.feat[f] = x.feat[f] + df[f];
}

double state_norm(const Problem &pb, const State &x) {  // x_.norm() over the reduced program's parameters
  double s = 0;
  int npose = pb.P + (pb.has_loop ? 1 : 0);
  for (int i = 0; i < 7 * npose; i++) s += x.pose[i] * x.pose[i];
  for (double v : x.sb) s += v * v;
  for (double v : x.feat) s += v * v;
  return std::sqrt(s);
}
double state_diff_norm(const Problem &pb, const State &a, const State &b, bool inf) {
  double s = 0, m = 0;
  auto acc = [&](double d) { s += d * d, m = std::fmax(m, std::fabs(d)); };
  int npose = pb.P + (pb.has_loop ? 1 : 0);
  for (int i = 0; i < 7 * npose; i++) acc(a.pose[i] - b.pose[i]);
  for (size_t i = 0; i < a.sb.size(); i++) acc(a.sb[i] - b.sb[i]);
  for (size_t i = 0; i < a.feat.size(); i++) acc(a.feat[i] - b.feat[i]);
  return inf ? m : std::sqrt(s);
}

// y = (S H S) v for v = [vp; vf]
void scaled_Hv(const Problem &pb, const Lin &L, const Vec &sp, const Vec &sf, const Vec &vp, const Vec &vf, Vec &yp,
               Vec &yf) {
  const int np = pb.np, F = pb.F;
  Vec tp(np), tf(F);
  for (int i = 0; i < np; i++) tp[i] = sp[i] * vp[i];
  for (int f = 0; f < F; f++) tf[f] = sf[f] * vf[f];
  yp.assign(np, 0.0), yf.assign(F, 0.0);
  for (int i = 0; i < np; i++) {
    double s = 0;
    for (int j = 0; j < np; j++) s += L.Hpp[i * np + j] * tp[j];
    yp[i] = s;
  }
  for (int f = 0; f < F; f++) {
    double s = L.Hff[f] * tf[f];
    for (int j = 0; j < np; j++) {
      double wv = L.Hpf[f * np + j];
      if (wv != 0.0) s += wv * tp[j], yp[j] += wv * tf[f];
    }
    yf[f] = s;
  }
  for (int i = 0; i < np; i++) yp[i] *= sp[i];
  for (int f = 0; f < F; f++) yf[f] *= sf[f];
}

// Solves (S H S + diag(D^2)) y = S g with the features eliminated (SchurEliminator + LLT + BackSubstitute).
bool schur_solve(const Problem &pb, const Lin &L, const Vec &sp, const Vec &sf, const Vec &Dp, const Vec &Df, Vec &yp,
                 Vec &yf) {
  const int np = pb.np, F = pb.F;
  Vec S(np * np), rhs(np), e(F), gsf(F);
  for (int i = 0; i < np; i++) {
    for (int j = 0; j < np; j++) S[i * np + j] = sp[i] * L.Hpp[i * np + j] * sp[j];
    S[i * np + i] += Dp[i] * Dp[i];
    rhs[i] = sp[i] * L.gp[i];
  }
  Vec wrow(np);
  for (int f = 0; f < F; f++) {
    e[f] = sf[f] * L.Hff[f] * sf[f] + Df[f] * Df[f];
    gsf[f] = sf[f] * L.gf[f];
    if (!(e[f] > 0.0)) return false;
    std::vector<int> nz;
    for (int j = 0; j < np; j++) {
      wrow[j] = L.Hpf[f * np + j] * sp[j] * sf[f];
      if (L.Hpf[f * np + j] != 0.0) nz.push_back(j);
    }
    double einv = 1.0 / e[f];
    for (int a : nz) {
      rhs[a] -= wrow[a] * einv * gsf[f];
      for (int b : nz) S[a * np + b] -= wrow[a] * einv * wrow[b];
    }
  }
  if (!cholesky_lower(S.data(), np)) return false;
  chol_solve(S.data(), np, rhs.data());
  yp = rhs;
  yf.assign(F, 0.0);
  for (int f = 0; f < F; f++) {
    double s = gsf[f];
    for (int j = 0; j < np; j++) s -= L.Hpf[f * np + j] * sp[j] * sf[f] * yp[j];
    yf[f] = s / e[f];
  }
  for (int i = 0; i < np; i++)
    if (!std::isfinite(yp[i])) return false;
  for (int f = 0; f < F; f++)
    if (!std::isfinite(yf[f])) return false;
  return true;
}

// ---------------------------------------------------------------------------------------------------
// TrustRegionMinimizer + DoglegStrategy
void minimize(const Problem &pb, State &x, VioSolveStats *st) {
  const int np = pb.np, F = pb.F;
  const int max_it = pb.cfg->max_iterations;
  Lin L;
  double x_cost = evaluate(pb, x, &L);
  double x_norm = -1.0;  // "Invalid value", trust_region_minimizer.cc:168
  Vec sp(np), sf(F);
  for (int i = 0; i < np; i++) sp[i] = 1.0 / (1.0 + std::sqrt(L.Hpp[i * np + i]));  // :239-254
  for (int f = 0; f < F; f++) sf[f] = 1.0 / (1.0 + std::sqrt(L.Hff[f]));
  auto grad_max_norm = [&](const State &xs, const Lin &LL) {  // |x - Plus(x, -g)|_inf, :270-284
    Vec ngp(np), ngf(F);
    for (int i = 0; i < np; i++) ngp[i] = -LL.gp[i];
    for (int f = 0; f < F; f++) ngf[f] = -LL.gf[f];
    State t;
    plus(pb, xs, ngp.data(), ngf.data(), t);
    return state_diff_norm(pb, xs, t, true);
  };
  double radius = 1e4, mu = 1e-8;  // solver.h initial_trust_region_radius; dogleg_strategy.cc:48-49
  const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
  bool reuse = false;
  double dogleg_step_norm = 0;
  Vec dgp(np), dgf(F), gd_p(np), gd_f(F), gn_p(np), gn_f(F);  // diagonal, gradient (d-scaled), GN step (d-scaled)
  double alpha = 0;
  int it = 0, n_ok = 0, n_bad = 0, invalid_run = 0;
  int termination = 0;
  // TrustRegionStepEvaluator with max_consecutive_nonmonotonic_steps = 0
  double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost, ev_acc_ref = 0, ev_acc_cand = 0;
  int recorded = 0;
  double min_recorded_cost = std::numeric_limits<double>::max();
  auto record = [&](int i, double cost, double step_norm, double rel, double gmax, bool valid, bool ok) {
    recorded = i + 1;
    min_recorded_cost = std::fmin(min_recorded_cost, cost);  // SetSummaryFinalCost, CSI/solver_utils.h:48-56
    if (st && i < VIO_MAX_TRACE) {
      st->it_cost[i] = cost, st->it_radius[i] = radius, st->it_step_norm[i] = step_norm;
      st->it_relative_decrease[i] = rel, st->it_gradient_max_norm[i] = gmax;
      st->it_flags[i] = (valid ? 1 : 0) | (ok ? 2 : 0);
    }
  };
  double gmax = grad_max_norm(x, L);
  bool last_ok = true;
  n_ok++;
  record(0, x_cost, 0, 0, gmax, true, true);
  if (st) st->initial_cost = x_cost;
  Vec step_p(np), step_f(F), del_p(np), del_f(F), yp, yf;
  State cand;
  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue (:291-340)
    if (it >= max_it) break;
    if (last_ok && gmax <= 1e-10) { termination = 1; break; }
    if (radius <= 1e-32) { termination = 1; break; }
    it++;
    // ---- DoglegStrategy::ComputeStep (dogleg_strategy.cc:77-163)
    bool solver_ok = true;
    if (!reuse) {
      reuse = true;
      double gsq = 0;
      for (int i = 0; i < np; i++) {
        double c = sp[i] * sp[i] * L.Hpp[i * np + i];
        dgp[i] = std::sqrt(std::fmin(std::fmax(c, 1e-6), 1e32));
        gd_p[i] = sp[i] * L.gp[i] / dgp[i];
        gsq += gd_p[i] * gd_p[i];
      }
      for (int f = 0; f < F; f++) {
        double c = sf[f] * sf[f] * L.Hff[f];
        dgf[f] = std::sqrt(std::fmin(std::fmax(c, 1e-6), 1e32));
        gd_f[f] = sf[f] * L.gf[f] / dgf[f];
        gsq += gd_f[f] * gd_f[f];
      }
      // Cauchy point: alpha = |g|^2 / |J (g/d)|^2 (:172-192)
      Vec vp(np), vf(F), hp, hf;
      for (int i = 0; i < np; i++) vp[i] = gd_p[i] / dgp[i];
      for (int f = 0; f < F; f++) vf[f] = gd_f[f] / dgf[f];
      scaled_Hv(pb, L, sp, sf, vp, vf, hp, hf);
      double jg = 0;
      for (int i = 0; i < np; i++) jg += vp[i] * hp[i];
      for (int f = 0; f < F; f++) jg += vf[f] * hf[f];
      alpha = gsq / jg;
      // Gauss-Newton step with D = diag * sqrt(mu) (:515-612)
      solver_ok = false;
      while (mu < max_mu) {
        Vec Dp(np), Df(F);
        double sm = std::sqrt(mu);
        for (int i = 0; i < np; i++) Dp[i] = dgp[i] * sm;
        for (int f = 0; f < F; f++) Df[f] = dgf[f] * sm;
        if (schur_solve(pb, L, sp, sf, Dp, Df, yp, yf)) { solver_ok = true; break; }
        mu *= mu_inc;
      }
      if (solver_ok) {
        for (int i = 0; i < np; i++) gn_p[i] = yp[i] * -dgp[i];
        for (int f = 0; f < F; f++) gn_f[f] = yf[f] * -dgf[f];
      }
    }
    bool step_valid = false;
    double model_cost_change = 0;
    if (solver_ok) {
      // ComputeTraditionalDoglegStep (:199-255)
      double gnorm2 = 0, gnn2 = 0, gdot = 0;
      for (int i = 0; i < np; i++) gnorm2 += gd_p[i] * gd_p[i], gnn2 += gn_p[i] * gn_p[i], gdot += gd_p[i] * gn_p[i];
      for (int f = 0; f < F; f++) gnorm2 += gd_f[f] * gd_f[f], gnn2 += gn_f[f] * gn_f[f], gdot += gd_f[f] * gn_f[f];
      double gradient_norm = std::sqrt(gnorm2), gauss_newton_norm = std::sqrt(gnn2);
      double ca = 0, cb = 0;  // step = ca * gradient + cb * gauss_newton
      if (gauss_newton_norm <= radius) {
        ca = 0, cb = 1, dogleg_step_norm = gauss_newton_norm;
      } else if (gradient_norm * alpha >= radius) {
        ca = -(radius / gradient_norm), cb = 0, dogleg_step_norm = radius;
      } else {
        double b_dot_a = -alpha * gdot;
        double a_squared_norm = std::pow(alpha * gradient_norm, 2.0);
        double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + std::pow(gauss_newton_norm, 2);
        double c = b_dot_a - a_squared_norm;
        double d = std::sqrt(c * c + b_minus_a_squared_norm * (std::pow(radius, 2.0) - a_squared_norm));
        double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + c);
        ca = -alpha * (1.0 - beta), cb = beta;
        dogleg_step_norm = -1;  // norm of the combination, below
      }
      double n2 = 0;
      for (int i = 0; i < np; i++) {
        double s = ca * gd_p[i] + cb * gn_p[i];
        n2 += s * s;
        step_p[i] = s / dgp[i];
      }
      for (int f = 0; f < F; f++) {
        double s = ca * gd_f[f] + cb * gn_f[f];
        n2 += s * s;
        step_f[f] = s / dgf[f];
      }
      if (dogleg_step_norm < 0) dogleg_step_norm = std::sqrt(n2);
      // model_cost_change = -(J step)^T (r + J step / 2) (trust_region_minimizer.cc:402-416)
      Vec hp, hf;
      scaled_Hv(pb, L, sp, sf, step_p, step_f, hp, hf);
      double sg = 0, shs = 0;
      for (int i = 0; i < np; i++) sg += step_p[i] * sp[i] * L.gp[i], shs += step_p[i] * hp[i];
      for (int f = 0; f < F; f++) sg += step_f[f] * sf[f] * L.gf[f], shs += step_f[f] * hf[f];
      model_cost_change = -sg - 0.5 * shs;
      step_valid = model_cost_change > 0.0;
    }
    if (!step_valid) {
      // HandleInvalidStep (:429-462) + DoglegStrategy::StepIsInvalid
      if (++invalid_run >= 5) { termination = 2; break; }  // FAILURE: iteration not recorded (:437-446)
      mu *= mu_inc;
      reuse = false;
      last_ok = false;
      n_bad++;
      record(it, x_cost, 0, 0, gmax, false, false);
      continue;
    }
    invalid_run = 0;
    for (int i = 0; i < np; i++) del_p[i] = step_p[i] * sp[i];
    for (int f = 0; f < F; f++) del_f[f] = step_f[f] * sf[f];
    plus(pb, x, del_p.data(), del_f.data(), cand);
    double cand_cost = evaluate(pb, cand, NULL);
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    // ParameterToleranceReached (:667-686)
    double step_norm = state_diff_norm(pb, x, cand, false);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { termination = 1; break; }
    // FunctionToleranceReached (:689-705)
    double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= 1e-6 * x_cost) { termination = 1; break; }
    // IsStepSuccessful: StepQuality (trust_region_step_evaluator.cc:52-60)
    double rel = (ev_cur - cand_cost) / model_cost_change;
    double hist = (ev_ref - cand_cost) / (ev_acc_ref + model_cost_change);
    double rho = std::fmax(rel, hist);
    if (rho > 1e-3) {
      x = cand;
      x_norm = state_norm(pb, x);
      x_cost = evaluate(pb, x, &L);
      gmax = grad_max_norm(x, L);
      // DoglegStrategy::StepAccepted (:614-629)
      if (rho < 0.25) radius *= 0.5;
      if (rho > 0.75) radius = std::fmax(radius, 3.0 * dogleg_step_norm);
      mu = std::fmax(min_mu, 2.0 * mu / mu_inc);
      reuse = false;
      // TrustRegionStepEvaluator::StepAccepted
      ev_cur = cand_cost, ev_acc_cand += model_cost_change, ev_acc_ref += model_cost_change;
      if (ev_cur < ev_min) ev_min = ev_cur, ev_cand = ev_cur, ev_acc_cand = 0;
      else if (ev_cur > ev_cand) ev_cand = ev_cur, ev_acc_cand = 0;
      ev_ref = ev_cand, ev_acc_ref = ev_acc_cand;
      last_ok = true;
      n_ok++;
      record(it, x_cost, step_norm, rho, gmax, true, true);
    } else {
      radius *= 0.5;  // StepRejected (:631-634)
      reuse = true;
      last_ok = false;
      n_bad++;
      record(it, cand_cost, step_norm, rho, 0.0, true, false);  // IterationSummary() default gradient_max_norm
    }
  }
  if (st) {
    st->final_cost = min_recorded_cost;
    st->iterations = recorded;  // tolerance exits return before the iteration is pushed (:104-110)
    st->termination = termination;
    st->num_successful_steps = n_ok;
    st->num_unsuccessful_steps = n_bad;
  }
}

// ---------------------------------------------------------------------------------------------------
// marginalization (marginalization_factor.cpp:182-300 at the VINS.cpp:690-830 call sites)
struct MBlock {
  int kind, index, gsize, lsize, off;  // kind: VIO_BLOCK_* or 3 = feature
  bool drop;
};

int find_block(std::vector<MBlock> &bl, int kind, int index, int gsize, bool drop) {
  for (size_t i = 0; i < bl.size(); i++)
    if (bl[i].kind == kind && bl[i].index == index) {
      if (drop) bl[i].drop = true;
      return (int)i;
    }
  bl.push_back(MBlock{kind, index, gsize, gsize == 7 ? 6 : gsize, 0, drop});
  return (int)bl.size() - 1;
}

struct MFactor {
  int rows;
  std::vector<int> blocks;
  std::vector<Vec> J;  // rows x lsize each
  Vec r;
};

void marginalize(const Problem &pb, const State &x, int flag, VioPrior *out) {
  const VioWindow *w = pb.w;
  const int W = pb.W;
  std::vector<MBlock> bl;
  std::vector<MFactor> fs;
  const VioPrior *pr = w->prior;
  bool use_prior = pr != NULL;
  if (flag == VIO_MARGIN_SECOND_NEW) {
    bool touches = false;
    if (pr)
      for (int b = 0; b < pr->n_blocks; b++)
        if (pr->block_kind[b] == VIO_BLOCK_POSE && pr->block_index[b] == W - 1) touches = true;
    if (!touches) { out->n = -1, out->n_blocks = 0; return; }
  }
  // prior as a factor
  if (use_prior) {
    MFactor f;
    f.rows = pr->n;
    Vec dx(pr->n, 0.0);
    for (int b = 0; b < pr->n_blocks; b++) {
      int kind = pr->block_kind[b], idx = pr->block_index[b], o = pr->block_offset[b];
      bool drop = (flag == VIO_MARGIN_OLD) ? (idx == 0 && kind != VIO_BLOCK_EXPOSE)
                                           : (kind == VIO_BLOCK_POSE && idx == W - 1);
      int gs = kind == VIO_BLOCK_SPEEDBIAS ? 9 : 7;
      const double *xv = kind == VIO_BLOCK_POSE ? &x.pose[7 * idx] : kind == VIO_BLOCK_SPEEDBIAS ? &x.sb[9 * idx] : x.ex;
      prior_block_dx(gs, xv, pr->block_x0 + 9 * b, &dx[o]);
      int bi = find_block(bl, kind, idx, gs, drop);
      f.blocks.push_back(bi);
      int ls = bl[bi].lsize;
      Vec J(pr->n * ls);
      for (int r = 0; r < pr->n; r++)
        for (int c = 0; c < ls; c++) J[r * ls + c] = pr->linearized_jacobians[r * pr->n + o + c];
      f.J.push_back(J);
    }
    f.r.resize(pr->n);
    for (int i = 0; i < pr->n; i++) {
      double s = pr->linearized_residuals[i];
      for (int j = 0; j < pr->n; j++) s += pr->linearized_jacobians[i * pr->n + j] * dx[j];
      f.r[i] = s;
    }
    fs.push_back(f);
  }
  if (flag == VIO_MARGIN_OLD) {
    {  // IMUFactor(pre_integrations[1]) drop {pose0, sb0}
      MFactor f;
      f.rows = 15;
      f.r.resize(15);
      Vec Jpi(90), Jsi(135), Jpj(90), Jsj(135);
      imu_eval(pb.cfg, &w->preint[0], &pb.U[0], &x.pose[0], &x.sb[0], &x.pose[7], &x.sb[9], f.r.data(), Jpi.data(),
               Jsi.data(), Jpj.data(), Jsj.data());
      f.blocks = {find_block(bl, VIO_BLOCK_POSE, 0, 7, true), find_block(bl, VIO_BLOCK_SPEEDBIAS, 0, 9, true),
                  find_block(bl, VIO_BLOCK_POSE, 1, 7, false), find_block(bl, VIO_BLOCK_SPEEDBIAS, 1, 9, false)};
      f.J = {Jpi, Jsi, Jpj, Jsj};
      fs.push_back(f);
    }
    const double bb = pb.cfg->cauchy_a * pb.cfg->cauchy_a, cc = 1.0 / bb;
    for (int k = 0; k < pb.M; k++) {
      if (w->factor_host[k] != 0 || w->factor_target[k] == pb.P) continue;
      int t = w->factor_target[k], fi = w->factor_feature[k];
      MFactor f;
      f.rows = 2;
      f.r.resize(2);
      Vec Ji(12), Jj(12), Jex(12), Jl(2);
      projection_eval(pb.s_info, &x.pose[0], &x.pose[7 * t], x.ex, x.feat[fi], w->factor_pts_i + 3 * k,
                      w->factor_pts_j + 3 * k, f.r.data(), Ji.data(), Jj.data(), Jex.data(), Jl.data());
      // ResidualBlockInfo::Evaluate loss correction (marginalization_factor.cpp:45-76); rho'' < 0 branch
      double sq = f.r[0] * f.r[0] + f.r[1] * f.r[1];
      double rho1 = 1.0 / (1.0 + sq * cc);
      double sr = std::sqrt(rho1);
      for (int q = 0; q < 12; q++) Ji[q] *= sr, Jj[q] *= sr, Jex[q] *= sr;
      Jl[0] *= sr, Jl[1] *= sr, f.r[0] *= sr, f.r[1] *= sr;
      f.blocks = {find_block(bl, VIO_BLOCK_POSE, 0, 7, true), find_block(bl, VIO_BLOCK_POSE, t, 7, false),
                  find_block(bl, VIO_BLOCK_EXPOSE, 0, 7, false), find_block(bl, 3, fi, 1, true)};
      f.J = {Ji, Jj, Jex, Jl};
      fs.push_back(f);
    }
  }
  // order: dropped blocks first (m), kept after (n). Kept order: poses by index, speed-biases by index, extrinsic.
  std::vector<int> order;
  for (size_t i = 0; i < bl.size(); i++)
    if (bl[i].drop) order.push_back((int)i);
  int m = 0;
  for (int i : order) bl[i].off = m, m += bl[i].lsize;
  std::vector<int> kept;
  for (int kind = 0; kind < 3; kind++)
    for (int idx = 0; idx <= W; idx++)
      for (size_t i = 0; i < bl.size(); i++)
        if (!bl[i].drop && bl[i].kind == kind && bl[i].index == idx) kept.push_back((int)i);
  int pos = m;
  for (int i : kept) bl[i].off = pos, pos += bl[i].lsize;
  const int n = pos - m;
  Vec A(pos * pos, 0.0), b(pos, 0.0);
  for (const MFactor &f : fs)
    for (size_t a = 0; a < f.blocks.size(); a++) {
      const MBlock &ba = bl[f.blocks[a]];
      for (int c = 0; c < ba.lsize; c++) {
        double g = 0;
        for (int k = 0; k < f.rows; k++) g += f.J[a][k * ba.lsize + c] * f.r[k];
        b[ba.off + c] += g;
      }
      for (size_t b2 = 0; b2 < f.blocks.size(); b2++) {
        const MBlock &bb2 = bl[f.blocks[b2]];
        add_block(A, pos, ba.off, bb2.off, f.J[a].data(), ba.lsize, f.J[b2].data(), bb2.lsize, f.rows, ba.lsize,
                  bb2.lsize);
      }
    }
  // Amm^+ by eigen-decomposition with the eps = 1e-8 cut (marginalization_factor.cpp:267-275)
  const double eps = 1e-8;
  Vec Amm(m * m), wv(m > 0 ? m : 1), V(m * m), Ainv(m * m, 0.0);
  for (int i = 0; i < m; i++)
    for (int j = 0; j < m; j++) Amm[i * m + j] = 0.5 * (A[i * pos + j] + A[j * pos + i]);
  if (m > 0) jacobi_eig(Amm.data(), m, wv.data(), V.data());
  for (int i = 0; i < m; i++)
    for (int j = 0; j < m; j++) {
      double s = 0;
      for (int k = 0; k < m; k++)
        if (wv[k] > eps) s += V[i * m + k] * (1.0 / wv[k]) * V[j * m + k];
      Ainv[i * m + j] = s;
    }
  // A' = Arr - Arm Amm^+ Amr ; b' = brr - Arm Amm^+ bmm
  Vec T(n * m, 0.0), Ar(n * n), br(n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < m; j++) {
      double s = 0;
      for (int k = 0; k < m; k++) s += A[(m + i) * pos + k] * Ainv[k * m + j];
      T[i * m + j] = s;
    }
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) {
      double s = A[(m + i) * pos + m + j];
      for (int k = 0; k < m; k++) s -= T[i * m + k] * A[k * pos + m + j];
      Ar[i * n + j] = s;
    }
    double s = b[m + i];
    for (int k = 0; k < m; k++) s -= T[i * m + k] * b[k];
    br[i] = s;
  }
  // SelfAdjointEigenSolver reads the lower triangle only; symmetrize the same way
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) Ar[i * n + j] = Ar[j * n + i];
  Vec w2(n > 0 ? n : 1), V2(n * n);
  if (n > 0) jacobi_eig(Ar.data(), n, w2.data(), V2.data());
  out->n = n;
  out->n_blocks = (int)kept.size();
  for (int r = 0; r < n; r++) {
    double S = w2[r] > eps ? w2[r] : 0.0, Sinv = w2[r] > eps ? 1.0 / w2[r] : 0.0;
    double ssq = std::sqrt(S), sisq = std::sqrt(Sinv);
    double vb = 0;
    for (int c = 0; c < n; c++) {
      out->linearized_jacobians[r * n + c] = ssq * V2[c * n + r];
      vb += V2[c * n + r] * br[c];
    }
    out->linearized_residuals[r] = sisq * vb;
  }
  for (size_t k = 0; k < kept.size(); k++) {
    const MBlock &B = bl[kept[k]];
    int new_index = B.index;
    // addr_shift (VINS.cpp:760-769 / :804-823)
    if (B.kind != VIO_BLOCK_EXPOSE) {
      if (flag == VIO_MARGIN_OLD) new_index = B.index - 1;
      else if (B.index == W) new_index = W - 1;
    }
    out->block_kind[k] = B.kind, out->block_index[k] = new_index, out->block_offset[k] = B.off - m;
    double *x0 = out->block_x0 + 9 * k;
    for (int q = 0; q < 9; q++) x0[q] = 0;
    const double *xv = B.kind == VIO_BLOCK_POSE ? &x.pose[7 * B.index] : B.kind == VIO_BLOCK_SPEEDBIAS ? &x.sb[9 * B.index] : x.ex;
    memcpy(x0, xv, sizeof(double) * B.gsize);
  }
}

}  // namespace

// ===================================================================================================
extern "C" {

int oracle_preintegrate(const VioConfig *cfg, const double acc_0[3], const double gyr_0[3], const double ba[3],
                        const double bg[3], int32_t n, const double *dt, const double *acc, const double *gyr,
                        VioPreintegration *out) {
  Integration ib;
  integration_init(ib, cfg, acc_0, gyr_0, ba, bg);
  for (int i = 0; i < n; i++) integration_propagate(ib, dt[i], acc + 3 * i, gyr + 3 * i);
  out->sum_dt = ib.sum_dt;
  for (int k = 0; k < 3; k++)
    out->delta_p[k] = ib.dp[k], out->delta_v[k] = ib.dv[k], out->linearized_ba[k] = ib.ba[k], out->linearized_bg[k] = ib.bg[k];
  out->delta_q[0] = ib.dq.x, out->delta_q[1] = ib.dq.y, out->delta_q[2] = ib.dq.z, out->delta_q[3] = ib.dq.w;
  memcpy(out->jacobian, ib.J, sizeof(ib.J)), memcpy(out->covariance, ib.C, sizeof(ib.C));
  return VIO_OK;
}

int oracle_eval_projection(const VioConfig *cfg, const double *pose_i, const double *pose_j, const double *ex,
                           const double *inv_depth, const double *pts_i, const double *pts_j, double *res,
                           double *jac) {
  double Ji[12], Jj[12], Jex[12], Jl[2];
  projection_eval(cfg->fx / 1.5, pose_i, pose_j, ex, inv_depth[0], pts_i, pts_j, res, jac ? Ji : NULL, Jj, Jex, Jl);
  if (jac) {
    memset(jac, 0, sizeof(double) * 44);
    for (int r = 0; r < 2; r++)
      for (int c = 0; c < 6; c++)
        jac[r * 7 + c] = Ji[r * 6 + c], jac[14 + r * 7 + c] = Jj[r * 6 + c], jac[28 + r * 7 + c] = Jex[r * 6 + c];
    jac[42] = Jl[0], jac[43] = Jl[1];
  }
  return VIO_OK;
}

int oracle_eval_imu(const VioConfig *cfg, const VioPreintegration *pre, const double *pose_i, const double *sb_i,
                    const double *pose_j, const double *sb_j, double *res, double *jac) {
  double U[225];
  if (!imu_sqrt_info(pre->covariance, U)) return VIO_EINVAL;
  double Jpi[90], Jsi[135], Jpj[90], Jsj[135];
  imu_eval(cfg, pre, U, pose_i, sb_i, pose_j, sb_j, res, jac ? Jpi : NULL, Jsi, Jpj, Jsj);
  if (jac) {
    memset(jac, 0, sizeof(double) * 480);
    for (int r = 0; r < 15; r++) {
      for (int c = 0; c < 6; c++) jac[r * 7 + c] = Jpi[r * 6 + c], jac[240 + r * 7 + c] = Jpj[r * 6 + c];
      for (int c = 0; c < 9; c++) jac[105 + r * 9 + c] = Jsi[r * 9 + c], jac[345 + r * 9 + c] = Jsj[r * 9 + c];
    }
  }
  return VIO_OK;
}

int oracle_solve_window(const VioConfig *cfg, VioWindow *w, VioSolveStats *stats) {
  Problem pb;
  pb.cfg = cfg, pb.w = w;
  pb.W = w->window_size, pb.P = pb.W + 1, pb.F = w->n_features, pb.M = w->n_factors;
  pb.has_loop = false;
  for (int k = 0; k < pb.M; k++) {
    if (w->factor_target[k] == pb.P) pb.has_loop = true;
    if (w->factor_feature[k] < 0 || w->factor_feature[k] >= pb.F) return VIO_EINVAL;
    if (w->factor_host[k] < 0 || w->factor_host[k] >= pb.P || w->factor_target[k] < 0 || w->factor_target[k] > pb.P)
      return VIO_EINVAL;
  }
  if (pb.has_loop && (w->loop_frame < 0 || w->loop_frame >= pb.W)) return VIO_EINVAL;
  pb.np = 15 * pb.P + (pb.has_loop ? 6 : 0);
  pb.s_info = cfg->fx / 1.5;  // ProjectionFactor::sqrt_info, VINS.cpp:31
  pb.U.resize(225 * pb.W);
  for (int i = 0; i < pb.W; i++)
    if (!imu_sqrt_info(w->preint[i].covariance, &pb.U[225 * i])) return VIO_EINVAL;
  State x;
  x.pose.assign(7 * (pb.P + 1), 0.0);
  memcpy(x.pose.data(), w->pose, sizeof(double) * 7 * pb.P);
  if (pb.has_loop) memcpy(&x.pose[7 * pb.P], &x.pose[7 * w->loop_frame], sizeof(double) * 7);  // VINS.cpp:590-591
  x.sb.assign(w->speed_bias, w->speed_bias + 9 * pb.P);
  x.feat.assign(w->inv_depth, w->inv_depth + pb.F);
  memcpy(x.ex, w->ex_pose, sizeof(x.ex));

  // Rs[0], Ps[0] as new2old() sees them
  double R0_in[9], ypr0[3];
  qtoR(qnormalized(qfrom(&x.pose[0])), R0_in);
  R2ypr(R0_in, ypr0);
  double P0_in[3] = {x.pose[0], x.pose[1], x.pose[2]};

  if (stats) memset(stats, 0, sizeof(*stats));
  minimize(pb, x, stats);

  if (w->raw_pose) memcpy(w->raw_pose, x.pose.data(), sizeof(double) * 7 * pb.P);
  if (w->raw_speed_bias) memcpy(w->raw_speed_bias, x.sb.data(), sizeof(double) * 9 * pb.P);
  if (w->raw_inv_depth) memcpy(w->raw_inv_depth, x.feat.data(), sizeof(double) * pb.F);
  if (pb.has_loop && w->loop_pose) memcpy(w->loop_pose, &x.pose[7 * pb.P], sizeof(double) * 7);

  // ---- new2old (VINS.cpp:131-212) then old2new (VINS.cpp:89-129)
  double origin_yaw = ypr0[0];
  double origin_P0[3] = {P0_in[0], P0_in[1], P0_in[2]};
  if (w->use_origin_override) {
    origin_yaw = w->origin_yaw_deg;
    memcpy(origin_P0, w->origin_p, sizeof(origin_P0));
  }
  double R00[9], ypr00[3];
  qtoR(qfrom(&x.pose[0]), R00);
  R2ypr(R00, ypr00);
  double yd[3] = {origin_yaw - ypr00[0], 0, 0}, rot_diff[9];
  ypr2R(yd, rot_diff);
  double p0[3] = {x.pose[0], x.pose[1], x.pose[2]};
  for (int i = 0; i < pb.P; i++) {
    double *pp = &x.pose[7 * i], *sb = &x.sb[9 * i];
    double Rq[9], Rs[9], d[3] = {pp[0] - p0[0], pp[1] - p0[1], pp[2] - p0[2]}, Ps[3], Vs[3];
    qtoR(qnormalized(qfrom(pp)), Rq);
    mat3mul(rot_diff, Rq, Rs);
    mat3vec(rot_diff, d, Ps);
    mat3vec(rot_diff, sb, Vs);
    Q q = RtoQ(Rs);
    for (int k = 0; k < 3; k++) pp[k] = Ps[k] + origin_P0[k], sb[k] = Vs[k];
    pp[3] = q.x, pp[4] = q.y, pp[5] = q.z, pp[6] = q.w;
  }
  for (int f = 0; f < pb.F; f++) {  // setDepth / getDepthVector round trip (feature_manager.cpp:300-349)
    double estimated_depth = 1.0 / x.feat[f];
    x.feat[f] = 1. / estimated_depth;
  }
  memcpy(w->pose, x.pose.data(), sizeof(double) * 7 * pb.P);
  memcpy(w->speed_bias, x.sb.data(), sizeof(double) * 9 * pb.P);
  memcpy(w->inv_depth, x.feat.data(), sizeof(double) * pb.F);

  if (w->next_prior) {
    if (w->marginalization_flag == VIO_MARGIN_NONE) w->next_prior->n = -1, w->next_prior->n_blocks = 0;
    else marginalize(pb, x, w->marginalization_flag, w->next_prior);
  }
  return VIO_OK;
}

}  // extern "C"
