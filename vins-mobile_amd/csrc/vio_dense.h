// vio_dense.h — small dense linear algebra for the host-side, one-off parts (initialisation): the handful of Eigen
// calls the reference makes there (ldlt().solve, jacobiSvd, 3x3 determinants), on plain row-major arrays.
#pragma once
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace vio {
namespace dense {

// Solves A x = b for symmetric A (n x n, row-major, destroyed) by LDL^T with symmetric pivoting on the largest remaining
// diagonal entry — the strategy of Eigen::LDLT (Cholesky/LDLT.h), which the reference calls on its normal equations.
inline bool ldlt_solve(std::vector<double> &A, std::vector<double> &b, int n, std::vector<double> &x) {
  std::vector<int> perm(n);
  for (int i = 0; i < n; i++) perm[i] = i;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = fabs(A[(size_t)k * n + k]);
    for (int i = k + 1; i < n; i++)
      if (fabs(A[(size_t)i * n + i]) > best) best = fabs(A[(size_t)i * n + i]), p = i;
    if (p != k) {  // symmetric row/column swap
      for (int j = 0; j < n; j++) std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]);
      for (int i = 0; i < n; i++) std::swap(A[(size_t)i * n + k], A[(size_t)i * n + p]);
      std::swap(b[k], b[p]), std::swap(perm[k], perm[p]);
    }
    const double d = A[(size_t)k * n + k];
    if (d == 0.0 || !std::isfinite(d)) return false;
    for (int i = k + 1; i < n; i++) {
      const double aik = A[(size_t)i * n + k];  // column k still holds a_jk for every j
      if (aik != 0.0) {
        const double l = aik / d;
        for (int j = k + 1; j <= i; j++) A[(size_t)i * n + j] -= l * A[(size_t)j * n + k];
      }
    }
    for (int i = k + 1; i < n; i++) {
      A[(size_t)i * n + k] /= d;  // l_ik
      A[(size_t)k * n + i] = A[(size_t)i * n + k];
      for (int j = i + 1; j < n; j++) A[(size_t)i * n + j] = A[(size_t)j * n + i];  // keep it symmetric for the pivot search / swaps
    }
  }
  // L y = b, D z = y, L^T w = z
  std::vector<double> y(b);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) y[i] -= A[(size_t)i * n + j] * y[j];
  for (int i = 0; i < n; i++) y[i] /= A[(size_t)i * n + i];
  for (int i = n - 1; i >= 0; i--)
    for (int j = i + 1; j < n; j++) y[i] -= A[(size_t)j * n + i] * y[j];
  x.assign(n, 0.0);
  for (int i = 0; i < n; i++) x[perm[i]] = y[i];
  return true;
}

// One-sided Jacobi (Hestenes) SVD of A (m x n, row-major, m >= n or not): on return the columns of A are U*S, V (n x n)
// holds the right singular vectors, s the singular values (unsorted). Accurate to rounding for the 3x3 .. (2k)x9 systems
// of the initialisation, the family Eigen::JacobiSVD belongs to.
inline void jacobi_svd(std::vector<double> &A, int m, int n, std::vector<double> &V, std::vector<double> &s) {
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 80; sweep++) {
    double off = 0;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < m; i++) {
          const double ap = A[(size_t)i * n + p], aq = A[(size_t)i * n + q];
          alpha += ap * ap, beta += aq * aq, gamma += ap * aq;
        }
        if (gamma == 0.0) continue;
        off = std::max(off, fabs(gamma) / sqrt(alpha * beta + 1e-300));
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < m; i++) {
          const double ap = A[(size_t)i * n + p], aq = A[(size_t)i * n + q];
          A[(size_t)i * n + p] = c * ap - sn * aq, A[(size_t)i * n + q] = sn * ap + c * aq;
        }
        for (int i = 0; i < n; i++) {
          const double vp = V[(size_t)i * n + p], vq = V[(size_t)i * n + q];
          V[(size_t)i * n + p] = c * vp - sn * vq, V[(size_t)i * n + q] = sn * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  s.assign(n, 0.0);
  for (int j = 0; j < n; j++) {
    double t = 0;
    for (int i = 0; i < m; i++) t += A[(size_t)i * n + j] * A[(size_t)i * n + j];
    s[j] = sqrt(t);
  }
}

// Full SVD of a 3x3 matrix: M = U diag(s) V^T with s sorted descending (U, V orthogonal, possibly improper).
inline void svd3(const double M[9], double U[9], double s[3], double V[9]) {
  std::vector<double> A(M, M + 9), Vv, sv;
  jacobi_svd(A, 3, 3, Vv, sv);
  int order[3] = {0, 1, 2};
  std::sort(order, order + 3, [&](int a, int b) { return sv[a] > sv[b]; });
  for (int k = 0; k < 3; k++) {
    const int j = order[k];
    s[k] = sv[j];
    for (int i = 0; i < 3; i++) V[i * 3 + k] = Vv[i * 3 + j];
    if (sv[j] > 1e-300)
      for (int i = 0; i < 3; i++) U[i * 3 + k] = A[i * 3 + j] / sv[j];
  }
  // complete U when a singular value vanished (rank-2 essential matrices): third column = cross of the first two
  if (s[2] <= 1e-12 * s[0]) {
    U[2] = U[3] * U[7] - U[6] * U[4], U[5] = U[6] * U[1] - U[0] * U[7], U[8] = U[0] * U[4] - U[3] * U[1];
  }
}

inline double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// Right singular vector of the smallest singular value of A (m x n).
inline void null_vector(std::vector<double> A, int m, int n, double *v) {
  std::vector<double> V, s;
  jacobi_svd(A, m, n, V, s);
  int j = 0;
  for (int k = 1; k < n; k++)
    if (s[k] < s[j]) j = k;
  for (int i = 0; i < n; i++) v[i] = V[(size_t)i * n + j];
}

}  // namespace dense
}  // namespace vio
