// vio_math.h — small fixed-size fp64 math for the back-end kernels (quaternions x y z w like para_Pose).
// Formulas follow Eigen 3.3.0's Quaternion (EIG/Eigen/src/Geometry/Quaternion.h) so that results track the
// reference's factor code to rounding level.
#pragma once
#include <math.h>

#ifndef VIO_HD
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VIO_HD __host__ __device__ __forceinline__
#else
#define VIO_HD inline
#endif
#endif

namespace vio {

// Reciprocal, square root and reciprocal square root for the solver kernels. On the device an IEEE f64 division costs
// ~85 cycles of dependent latency and sqrt ~120 (div_scale / div_fmas / div_fixup sequences); the hardware seeds
// (v_rcp_f64, v_rsq_f64: ~26 bits) with two Newton steps cost ~50 and are good to an ulp or two, which is far inside
// the solver's 1e-6 bar. No special-case handling beyond what is noted: callers pass finite, non-zero (rcp) or
// positive (rsqrt) arguments; sqrt_f maps 0 to 0. Host builds (and the host emulation of the kernels) use the exact
// operations.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VIO_EMUL)
#define VIO_FAST_F64 1
#endif
VIO_HD double rcp_f(double x) {
#ifdef VIO_FAST_F64
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
VIO_HD double rsqrt_f(double x) {
#ifdef VIO_FAST_F64
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}
VIO_HD double sqrt_f(double x) {
#ifdef VIO_FAST_F64
  const double y = rsqrt_f(x);
  double s = x * y;
  s = fma(0.5 * y, fma(-s, s, x), s);
  return x == 0.0 ? 0.0 : s;  // (negative / NaN arguments keep the NaN)
#else
  return sqrt(x);
#endif
}

struct Quat {
  double x, y, z, w;
};

VIO_HD Quat qmul(const Quat &a, const Quat &b) {
  return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
VIO_HD Quat qinv(const Quat &q) {  // conjugate / squaredNorm
  double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
#ifdef VIO_FAST_F64
  const double in = rcp_f(n2);
  return Quat{-q.x * in, -q.y * in, -q.z * in, q.w * in};
#else
  return Quat{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
#endif
}
VIO_HD Quat qnormalized(const Quat &q) {
#ifdef VIO_FAST_F64
  const double in = rsqrt_f(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Quat{q.x * in, q.y * in, q.z * in, q.w * in};
#else
  double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Quat{q.x / n, q.y / n, q.z / n, q.w / n};
#endif
}
VIO_HD void qrot(const Quat &q, const double v[3], double out[3]) {
  double ux = 2 * (q.y * v[2] - q.z * v[1]), uy = 2 * (q.z * v[0] - q.x * v[2]), uz = 2 * (q.x * v[1] - q.y * v[0]);
  out[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
  out[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
  out[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}
VIO_HD void qtoR(const Quat &q, double R[9]) {
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
  R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
  R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
}
VIO_HD Quat RtoQ(const double R[9]) {
  Quat q;
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R[7] - R[5]) * t, q.y = (R[2] - R[6]) * t, q.z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
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
template <class P>
VIO_HD Quat qfrom_pose(P p) {
  return Quat{p[3], p[4], p[5], p[6]};
}

VIO_HD void mat3mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
VIO_HD void mat3T(const double A[9], double T[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[i * 3 + j] = A[j * 3 + i];
}
VIO_HD void mat3vec(const double A[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; i++) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
VIO_HD void skew3(const double v[3], double S[9]) {
  S[0] = 0, S[1] = -v[2], S[2] = v[1], S[3] = v[2], S[4] = 0, S[5] = -v[0], S[6] = -v[1], S[7] = v[0], S[8] = 0;
}
// bottom-right 3x3 of Utility::Qleft(q) (utility.hpp:57-65)
VIO_HD void qleft33(const Quat &q, double M[9]) {
  double v[3] = {q.x, q.y, q.z};
  skew3(v, M);
  M[0] += q.w, M[4] += q.w, M[8] += q.w;
}
// bottom-right 3x3 of Qleft(a) * Qright(b) (utility.hpp:57-74)
VIO_HD void qleft_qright33(const Quat &a, const Quat &b, double M[9]) {
  double va[3] = {a.x, a.y, a.z}, vb[3] = {b.x, b.y, b.z};
  double L[9], Rm[9], S[9];
  skew3(va, L);
  L[0] += a.w, L[4] += a.w, L[8] += a.w;
  skew3(vb, S);
  for (int i = 0; i < 9; i++) Rm[i] = -S[i];
  Rm[0] += b.w, Rm[4] += b.w, Rm[8] += b.w;
  mat3mul(L, Rm, M);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[i * 3 + j] += va[i] * (-vb[j]);
}
// Utility::R2ypr / ypr2R, degrees (utility.hpp:76-118)
VIO_HD void R2ypr(const double R[9], double ypr[3]) {
  const double kPi = 3.14159265358979323846;
  double n0 = R[0], n1 = R[3], n2 = R[6], o0 = R[1], o1 = R[4], a0 = R[2], a1 = R[5];
  double y = atan2(n1, n0);
  double p = atan2(-n2, n0 * cos(y) + n1 * sin(y));
  double r = atan2(a0 * sin(y) - a1 * cos(y), -o0 * sin(y) + o1 * cos(y));
  ypr[0] = y / kPi * 180.0, ypr[1] = p / kPi * 180.0, ypr[2] = r / kPi * 180.0;
}
VIO_HD void ypr2R(const double ypr[3], double R[9]) {
  const double kPi = 3.14159265358979323846;
  double y = ypr[0] / 180.0 * kPi, p = ypr[1] / 180.0 * kPi, r = ypr[2] / 180.0 * kPi;
  double Rz[9] = {cos(y), -sin(y), 0, sin(y), cos(y), 0, 0, 0, 1};
  double Ry[9] = {cos(p), 0., sin(p), 0., 1., 0., -sin(p), 0., cos(p)};
  double Rx[9] = {1., 0., 0., 0., cos(r), -sin(r), 0., sin(r), cos(r)};
  double T[9];
  mat3mul(Rz, Ry, T);
  mat3mul(T, Rx, R);
}

}  // namespace vio
