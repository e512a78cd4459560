// vio_fivepoint.cpp — MotionEstimator::solveRelativeRT as the reference computes it (VINS_ios/motion_estimator.cpp:200-236,
// driven by VINS::relativePose, VINS_ios/VINS.cpp:1104-1145):
//     E = cv::findEssentialMat(ll, rr)            focal 1, pp (0, 0), RANSAC, prob 0.999, threshold 1.0
//     inlier_cnt = cv::recoverPose(E, ll, rr, rot, trans);  Rotation = R^T, Translation = -R^T T, ok = inlier_cnt > 10
// OpenCV ("customized 3.0.0", VINS_ThirdPartyLib/opencv2.version:1) is a binary of the reference that is not in the tree:
// the functions below restate the published 3.0.0 algorithms (calib3d/five-point.cpp: EMEstimatorCallback::runKernel /
// computeError, recoverPose, decomposeEssentialMat; calib3d/ptsetreg.cpp: RANSACPointSetRegistrator::run with
// RNG((uint64)-1); core/mathfuncs.cpp: solvePoly; calib3d/triangulate.cpp) -- PARITY UNPINNED. Known, stated deviations:
//   * the basis of the 4-dimensional null space comes from this file's SVD, OpenCV's comes from its own Jacobi SVD: the SET
//     of essential matrices a minimal sample yields is the same, the ORDER in which they are tried need not be;
//   * the 10 x 20 constraint matrix is built by polynomial arithmetic on the fly and the degree-10 polynomial as the
//     determinant of the 3 x 3 polynomial matrix by polynomial products, where OpenCV carries machine-generated expanded
//     expressions: same coefficients up to rounding.
// With the reference's arguments the threshold (1.0 in NORMALIZED image units, Sampson distance squared) accepts every
// correspondence, so RANSAC stops after its first sample and returns the first essential matrix that sample yields: whether
// solveRelativeRT succeeds on a given frame is, in the reference too, a matter of which root comes first; the caller retries
// on the next frame (VINS.cpp:893-901).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "vio_dense.h"
#include "vio_initial.h"
#include "vio_math.h"

namespace vio {
namespace init {

namespace {

struct CvRng {  // cv::RNG: multiply-with-carry (core/operations.hpp)
  uint64_t state;
  explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
  unsigned next() {
    state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

int cv_round(double v) { return (int)lrint(v); }

int ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {  // ptsetreg.cpp RANSACUpdateNumIters
  p = std::max(p, 0.), p = std::min(p, 1.), ep = std::max(ep, 0.), ep = std::min(ep, 1.);
  double num = std::max(1. - p, 2.2250738585072014e-308), denom = 1. - pow(1. - ep, model_points);
  if (denom < 2.2250738585072014e-308) return 0;
  num = log(num), denom = log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round(num / denom);
}

// ---- polynomials in (x, y, z) of total degree <= 3, dense over exponents 0..3 per variable ----------------------------------
struct Poly {
  double c[4][4][4];
  Poly() { memset(c, 0, sizeof(c)); }
};
Poly lin(double cx, double cy, double cz, double c1) {
  Poly p;
  p.c[1][0][0] = cx, p.c[0][1][0] = cy, p.c[0][0][1] = cz, p.c[0][0][0] = c1;
  return p;
}
Poly add(const Poly &a, const Poly &b, double sb = 1.0) {
  Poly r;
  for (int i = 0; i < 64; i++) (&r.c[0][0][0])[i] = (&a.c[0][0][0])[i] + sb * (&b.c[0][0][0])[i];
  return r;
}
Poly mul(const Poly &a, const Poly &b) {  // (degrees add up to <= 3 wherever this is called)
  Poly r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j + i < 4; j++)
      for (int k = 0; k + i + j < 4; k++) {
        const double av = a.c[i][j][k];
        if (av == 0.0) continue;
        for (int l = 0; l + i < 4; l++)
          for (int m = 0; m + j < 4; m++)
            for (int n = 0; n + k < 4; n++)
              if (i + j + k + l + m + n <= 3) r.c[i + l][j + m][k + n] += av * b.c[l][m][n];
      }
  return r;
}
Poly scale(const Poly &a, double s) {
  Poly r;
  for (int i = 0; i < 64; i++) (&r.c[0][0][0])[i] = s * (&a.c[0][0][0])[i];
  return r;
}

// Monomial order of the elimination (Nister, "An efficient solution to the five-point relative pose problem", 2004):
// x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1
const int kMono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                          {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};

// ---- polynomials in z (ascending coefficients) -------------------------------------------------------------------------------
typedef std::vector<double> ZP;
ZP zmul(const ZP &a, const ZP &b) {
  ZP r(a.size() + b.size() - 1, 0.0);
  for (size_t i = 0; i < a.size(); i++)
    for (size_t j = 0; j < b.size(); j++) r[i + j] += a[i] * b[j];
  return r;
}
ZP zsub(const ZP &a, const ZP &b) {
  ZP r(std::max(a.size(), b.size()), 0.0);
  for (size_t i = 0; i < a.size(); i++) r[i] += a[i];
  for (size_t i = 0; i < b.size(); i++) r[i] -= b[i];
  return r;
}

struct Cx {
  double re, im;
};
Cx cmul(Cx a, Cx b) { return Cx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
Cx cdiv(Cx a, Cx b) {  // cv::Complex operator/
  const double t = 1. / (b.re * b.re + b.im * b.im);
  return Cx{(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
}

// cv::solvePoly (Durand-Kerner from the powers of 1 + i, at most 300 sweeps, stops when no root moved): coeffs ascending.
int solve_poly(const double *coeffs, int deg, Cx *roots) {
  int n = deg;
  for (; n > 1; n--)
    if (fabs(coeffs[n]) > 2.220446049250313e-16) break;
  Cx p{1, 0};
  const Cx r{1, 1};
  for (int i = 0; i < n; i++) roots[i] = p, p = cmul(p, r);
  for (int iter = 0; iter < 300; iter++) {
    double max_diff = 0;
    for (int i = 0; i < n; i++) {
      p = roots[i];
      Cx num{coeffs[n], 0}, denom{coeffs[n], 0};
      for (int j = 0; j < n; j++) {
        num = cmul(num, p), num.re += coeffs[n - j - 1];
        if (j != i) denom = cmul(denom, Cx{p.re - roots[j].re, p.im - roots[j].im});
      }
      num = cdiv(num, denom);
      roots[i] = Cx{p.re - num.re, p.im - num.im};
      max_diff = std::max(max_diff, sqrt(num.re * num.re + num.im * num.im));
    }
    if (max_diff <= 0) break;
  }
  for (int i = 0; i < n; i++)
    if (fabs(roots[i].im) < 1e-100) roots[i].im = 0;
  return n;
}

}  // namespace

// EMEstimatorCallback::runKernel: the essential matrices (row-major, unit Frobenius norm, x2^T E x1 = 0) consistent with five
// correspondences; at most 10.
int five_point_kernel(const double q1[5][2], const double q2[5][2], double E[10][9]) {
  // x2^T E x1 = 0 is linear in the 9 entries of E (row-major): null space of the 5 x 9 design matrix
  std::vector<double> Q(5 * 9), V, s;
  for (int i = 0; i < 5; i++) {
    const double x1 = q1[i][0], y1 = q1[i][1], x2 = q2[i][0], y2 = q2[i][1];
    const double row[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
    memcpy(&Q[9 * i], row, sizeof(row));
  }
  dense::jacobi_svd(Q, 5, 9, V, s);
  int order[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
  std::stable_sort(order, order + 9, [&](int a, int b) { return s[a] > s[b]; });
  double EE[4][9];  // the four right singular vectors of the vanishing singular values
  for (int k = 0; k < 4; k++)
    for (int i = 0; i < 9; i++) EE[k][i] = V[(size_t)i * 9 + order[5 + k]];
  // E(x, y, z) = x EE0 + y EE1 + z EE2 + EE3; constraints det E = 0 and 2 E E^T E - tr(E E^T) E = 0: ten cubics
  Poly e[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) e[r][c] = lin(EE[0][3 * r + c], EE[1][3 * r + c], EE[2][3 * r + c], EE[3][3 * r + c]);
  Poly eq[10];
  eq[0] = add(add(mul(e[0][0], add(mul(e[1][1], e[2][2]), mul(e[1][2], e[2][1]), -1.0)),
                  mul(e[0][1], add(mul(e[1][0], e[2][2]), mul(e[1][2], e[2][0]), -1.0)), -1.0),
              mul(e[0][2], add(mul(e[1][0], e[2][1]), mul(e[1][1], e[2][0]), -1.0)));
  Poly eet[3][3], tr;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      eet[r][c] = add(add(mul(e[r][0], e[c][0]), mul(e[r][1], e[c][1])), mul(e[r][2], e[c][2]));
      if (r == c) tr = add(tr, eet[r][c]);
    }
  for (int r = 0; r < 3; r++) eet[r][r] = add(eet[r][r], tr, -0.5);  // E E^T - tr(E E^T) / 2
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      eq[1 + 3 * r + c] = add(add(mul(eet[r][0], e[0][c]), mul(eet[r][1], e[1][c])), mul(eet[r][2], e[2][c]));
  double A[10][20];
  for (int r = 0; r < 10; r++)
    for (int m = 0; m < 20; m++) A[r][m] = eq[r].c[kMono[m][0]][kMono[m][1]][kMono[m][2]];
  // A <- A[:, :10]^-1 A[:, 10:] (Gauss-Jordan with partial pivoting)
  for (int c = 0; c < 10; c++) {
    int piv = c;
    for (int r = c + 1; r < 10; r++)
      if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
    if (A[piv][c] == 0.0) return 0;
    if (piv != c)
      for (int m = 0; m < 20; m++) std::swap(A[c][m], A[piv][m]);
    const double d = 1.0 / A[c][c];
    for (int m = 0; m < 20; m++) A[c][m] *= d;
    for (int r = 0; r < 10; r++) {
      if (r == c) continue;
      const double f = A[r][c];
      if (f != 0.0)
        for (int m = 0; m < 20; m++) A[r][m] -= f * A[c][m];
    }
  }
  // rows x^2z, x^2 / y^2z, y^2 / xyz, xy: <row> - z <next row> has no monomial of the eliminated set left:
  // B(z) [x y 1]^T = 0 with B 3 x 3, entries cubic, cubic, quartic in z. b: [x z^3..1 | y z^3..1 | z^4..1]
  double b[3][13];
  for (int i = 0; i < 3; i++) {
    const double *a1 = &A[2 * i + 4][10], *a2 = &A[2 * i + 5][10];
    double row1[13] = {0}, row2[13] = {0};
    for (int k = 0; k < 3; k++) row1[1 + k] = a1[k], row1[5 + k] = a1[3 + k], row2[k] = a2[k], row2[4 + k] = a2[3 + k];
    for (int k = 0; k < 4; k++) row1[9 + k] = a1[6 + k], row2[8 + k] = a2[6 + k];
    for (int k = 0; k < 13; k++) b[i][k] = row1[k] - row2[k];
  }
  auto pz = [&](int j, int part) {  // ascending coefficients of entry (j, part)
    ZP p;
    if (part < 2) p = {b[j][4 * part + 3], b[j][4 * part + 2], b[j][4 * part + 1], b[j][4 * part]};
    else p = {b[j][12], b[j][11], b[j][10], b[j][9], b[j][8]};
    return p;
  };
  const ZP m0 = zsub(zmul(pz(1, 1), pz(2, 2)), zmul(pz(2, 1), pz(1, 2)));
  const ZP m1 = zsub(zmul(pz(1, 0), pz(2, 2)), zmul(pz(2, 0), pz(1, 2)));
  const ZP m2 = zsub(zmul(pz(1, 0), pz(2, 1)), zmul(pz(2, 0), pz(1, 1)));
  ZP det = zsub(zmul(pz(0, 0), m0), zmul(pz(0, 1), m1));
  const ZP last = zmul(pz(0, 2), m2);
  det.resize(11, 0.0);
  for (size_t k = 0; k < last.size() && k < 11; k++) det[k] += last[k];
  Cx roots[10];
  const int nroots = solve_poly(det.data(), 10, roots);
  int count = 0;
  for (int i = 0; i < nroots && count < 10; i++) {
    if (fabs(roots[i].im) > 1e-10) continue;
    const double z1 = roots[i].re, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
    std::vector<double> Bz(9);
    for (int j = 0; j < 3; j++) {
      Bz[3 * j] = b[j][0] * z3 + b[j][1] * z2 + b[j][2] * z1 + b[j][3];
      Bz[3 * j + 1] = b[j][4] * z3 + b[j][5] * z2 + b[j][6] * z1 + b[j][7];
      Bz[3 * j + 2] = b[j][8] * z4 + b[j][9] * z3 + b[j][10] * z2 + b[j][11] * z1 + b[j][12];
    }
    double xy1[3];
    dense::null_vector(Bz, 3, 3, xy1);  // SVD::solveZ
    if (fabs(xy1[2]) < 1e-10) continue;
    const double x = xy1[0] / xy1[2], y = xy1[1] / xy1[2];
    double nrm = 0;
    for (int k = 0; k < 9; k++) {
      E[count][k] = EE[0][k] * x + EE[1][k] * y + EE[2][k] * z1 + EE[3][k];
      nrm += E[count][k] * E[count][k];
    }
    nrm = sqrt(nrm);
    if (!(nrm > 0.0) || !std::isfinite(nrm)) continue;
    for (int k = 0; k < 9; k++) E[count][k] /= nrm;
    count++;
  }
  return count;
}

// EMEstimatorCallback::computeError: Sampson distance squared (the error Mat is CV_32F)
static float essential_error(const double E[9], const double p1[2], const double p2[2]) {
  const double x1[3] = {p1[0], p1[1], 1.0}, x2[3] = {p2[0], p2[1], 1.0};
  double Ex1[3], Etx2[3];
  mat3vec(E, x1, Ex1);
  for (int k = 0; k < 3; k++) Etx2[k] = E[k] * x2[0] + E[3 + k] * x2[1] + E[6 + k] * x2[2];
  const double x2tEx1 = x2[0] * Ex1[0] + x2[1] * Ex1[1] + x2[2] * Ex1[2];
  const double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
  return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
}

// cv::findEssentialMat(points1, points2, focal = 1, pp = (0, 0), RANSAC, prob, threshold) through
// RANSACPointSetRegistrator::run (modelPoints 5, maxIters 1000, RNG((uint64)-1)). xy0 / xy1 [n][2]. mask (optional) [n].
bool find_essential_ransac(const double *xy0, const double *xy1, int count, double prob, double threshold, double Ebest[9],
                           uint8_t *mask_out) {
  const int model_points = 5, max_iters = 1000;
  if (count < model_points) return false;
  std::vector<uint8_t> mask(count), best_mask(count, 0);
  const float t = (float)(threshold * threshold);
  auto find_inliers = [&](const double E[9]) {
    int good = 0;
    for (int i = 0; i < count; i++) {
      const int f = essential_error(E, xy0 + 2 * i, xy1 + 2 * i) <= t;
      mask[i] = (uint8_t)f, good += f;
    }
    return good;
  };
  double models[10][9];
  if (count == model_points) {
    double q1[5][2], q2[5][2];
    for (int i = 0; i < 5; i++) q1[i][0] = xy0[2 * i], q1[i][1] = xy0[2 * i + 1], q2[i][0] = xy1[2 * i], q2[i][1] = xy1[2 * i + 1];
    if (five_point_kernel(q1, q2, models) <= 0) return false;
    memcpy(Ebest, models[0], sizeof(double) * 9);
    if (mask_out) memset(mask_out, 1, count);
    return true;
  }
  CvRng rng((uint64_t)-1);
  int niters = max_iters, max_good = 0;
  bool have = false;
  for (int iter = 0; iter < niters; iter++) {
    // getSubset(..., maxAttempts = 10000); EMEstimatorCallback has no checkSubset of its own
    int idx[5];
    for (int i = 0; i < model_points; i++) {
      for (;;) {
        const int idx_i = idx[i] = rng.uniform(0, count);
        int j;
        for (j = 0; j < i; j++)
          if (idx_i == idx[j]) break;
        if (j == i) break;
      }
    }
    double q1[5][2], q2[5][2];
    for (int i = 0; i < 5; i++)
      q1[i][0] = xy0[2 * idx[i]], q1[i][1] = xy0[2 * idx[i] + 1], q2[i][0] = xy1[2 * idx[i]], q2[i][1] = xy1[2 * idx[i] + 1];
    const int nmodels = five_point_kernel(q1, q2, models);
    if (nmodels <= 0) continue;
    for (int k = 0; k < nmodels; k++) {
      const int good = find_inliers(models[k]);
      if (good > std::max(max_good, model_points - 1)) {
        std::swap(mask, best_mask);
        memcpy(Ebest, models[k], sizeof(double) * 9);
        max_good = good, have = true;
        niters = ransac_update_num_iters(prob, (double)(count - good) / count, model_points, niters);
      }
    }
  }
  if (have && mask_out) memcpy(mask_out, best_mask.data(), count);
  return have;
}

// cv::recoverPose(E, points1, points2, R, t) with focal 1, pp (0, 0): decomposeEssentialMat, the four (R, +-t) candidates,
// linear triangulation of every point, cheirality with the distance cut at 50. x2 ~ R x1 + t. Returns the inlier count.
int recover_pose(const double E[9], const double *xy0, const double *xy1, int n, double R[9], double t[3]) {
  double U[9], s[3], V[9];
  dense::svd3(E, U, s, V);
  if (dense::det3(U) < 0)
    for (int k = 0; k < 9; k++) U[k] = -U[k];
  if (dense::det3(V) < 0)  // det(Vt) = det(V)
    for (int k = 0; k < 9; k++) V[k] = -V[k];
  const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
  double Vt[9], UW[9], R1[9], R2[9];
  mat3T(V, Vt);
  mat3mul(U, W, UW), mat3mul(UW, Vt, R1);
  mat3mul(U, Wt, UW), mat3mul(UW, Vt, R2);
  const double tt[3] = {U[2], U[5], U[8]};
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero[3] = {0, 0, 0};
  const double *Rs[4] = {R1, R2, R1, R2};
  int good[4];
  for (int c = 0; c < 4; c++) {
    const double sg = c < 2 ? 1.0 : -1.0, tc[3] = {sg * tt[0], sg * tt[1], sg * tt[2]};
    // cvTriangulatePoints: rows x P[2] - P[0], y P[2] - P[1] of both views, the right singular vector of the smallest
    // singular value; homogeneous result Q
    good[c] = 0;
    for (int i = 0; i < n; i++) {
      const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
      double P1[12];
      for (int r = 0; r < 3; r++) P1[4 * r] = Rs[c][3 * r], P1[4 * r + 1] = Rs[c][3 * r + 1], P1[4 * r + 2] = Rs[c][3 * r + 2], P1[4 * r + 3] = tc[r];
      std::vector<double> A(16);
      const double x0 = xy0[2 * i], y0 = xy0[2 * i + 1], x1 = xy1[2 * i], y1 = xy1[2 * i + 1];
      for (int k = 0; k < 4; k++) {
        A[k] = x0 * P0[8 + k] - P0[k], A[4 + k] = y0 * P0[8 + k] - P0[4 + k];
        A[8 + k] = x1 * P1[8 + k] - P1[k], A[12 + k] = y1 * P1[8 + k] - P1[4 + k];
      }
      double Qh[4];
      dense::null_vector(A, 4, 4, Qh);
      bool ok = Qh[2] * Qh[3] > 0;
      const double X[3] = {Qh[0] / Qh[3], Qh[1] / Qh[3], Qh[2] / Qh[3]};
      ok = ok && X[2] < 50.0;
      const double z2 = P1[8] * X[0] + P1[9] * X[1] + P1[10] * X[2] + P1[11];
      ok = ok && z2 > 0 && z2 < 50.0;
      good[c] += ok ? 1 : 0;
    }
  }
  (void)I3, (void)zero;
  // the order of the reference's four tests: (R1, t), (R2, t), (R1, -t), (R2, -t), each "at least as good as the others"
  int pick = 3;
  if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) pick = 0;
  else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) pick = 1;
  else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) pick = 2;
  memcpy(R, Rs[pick], sizeof(double) * 9);
  const double sg = pick < 2 ? 1.0 : -1.0;
  for (int k = 0; k < 3; k++) t[k] = sg * tt[k];
  return good[pick];
}

// MotionEstimator::solveRelativeRT (motion_estimator.cpp:200-236). The correspondences pass through cv::Point2f in the
// reference: they are rounded to float here as well.
bool solve_relative_rt_five_point(const std::vector<double> &xy0, const std::vector<double> &xy1, double Rout[9], double tout[3],
                                  int *inliers) {
  const size_t n = xy0.size() / 2;
  if (inliers) *inliers = 0;
  if (n < 9 || xy1.size() != xy0.size()) return false;
  std::vector<double> a(2 * n), b(2 * n);
  for (size_t i = 0; i < 2 * n; i++) a[i] = (double)(float)xy0[i], b[i] = (double)(float)xy1[i];
  double E[9], R[9], t[3];
  if (!find_essential_ransac(a.data(), b.data(), (int)n, 0.999, 1.0, E, nullptr)) return false;
  const int cnt = recover_pose(E, a.data(), b.data(), (int)n, R, t);
  mat3T(R, Rout);  // Rotation = R^T, Translation = -R^T T
  double v[3];
  mat3vec(Rout, t, v);
  for (int k = 0; k < 3; k++) tout[k] = -v[k];
  if (inliers) *inliers = cnt;
  return cnt > 10;
}

}  // namespace init
}  // namespace vio
