// pnp_core.h — the motion-only window solve of the front-end (vinsPnP::solve_ceres, VINS_ios/vins_pnp.cpp:264-341),
// single source for the gfx950 kernel (vio_pnp.hip) and, with -DVIO_EMUL, the host emulation used by the CPU tests.
//
// The problem (reference lines in vins_pnp.cpp unless another file is named): PNP_SIZE + 1 = 7 frames, each with a pose
// (PoseLocalParameterization, 6 local dof) and a speed (3) block; the bias blocks and the camera extrinsic are constant,
// and so are pose and speed of the frames the back-end already solved (find_solved, :279-284). Factors: IMUFactorPnP
// between consecutive frames (imu_factor_pnp.h:20-216 — the back-end's IMU factor with speed and bias in separate
// blocks, no loss) and one PerspectiveFactor per tracked landmark with a FIXED 3D position (perspective_factor.cpp:16-67,
// weight track_num / 10, CauchyLoss(1)). Options (:317-326): DENSE_SCHUR, DOGLEG, max_num_iterations 5; the 0.01 s
// wall-clock limit is not reproduced (results would depend on the machine).
//
// One workgroup per window. The normal equations have at most 7 x 9 = 63 unknowns: the matrix lives in LDS as one dense
// block, factors are evaluated by all lanes and accumulated with LDS atomics, the factorization is a right-looking
// Cholesky with one lane per row. The trust-region loop is the one of the window solver (solver_core.h::minimize, i.e.
// Ceres' TrustRegionMinimizer + DoglegStrategy, CSI/trust_region_minimizer.cc, CSI/dogleg_strategy.cc) without
// landmark blocks.
#pragma once
#include "solver_core.h"

namespace vio {
namespace pnp {

constexpr int kMaxFrames = 8;
constexpr int kDof = 9;  // pose 6 + speed 3 per free frame
constexpr int kMaxDim = kMaxFrames * kDof;
constexpr int kPreDoubles = 17 + 225 + 225;  // VioPreintegration as doubles

struct View {  // one window, all pointers into global memory
  int n, M;
  const int *fixed;       // [n]
  const int *feat_start;  // [n+1]
  const double *pose0, *speed0, *bias, *ex;
  const double *preint;   // [n-1][kPreDoubles]
  const double *obs;      // [M][2]
  const double *pos;      // [M][3]
  const int *track;       // [M]
  double *out_pose, *out_speed;
  double *stats_d;
  int *stats_i;
  double *Jraw;           // [n-1][15*30] raw IMU Jacobians (scratch)
  double s_info, gravity, cauchy_b;
  int max_iter;
};

template <class P>
struct Work {
  P xp, xs, cp, cs;  // iterate and candidate: pose [n][7], speed [n][3]
  P H, Lf;           // normal matrix and its factor, dim x dim row-major (lower triangle valid)
  P g, sc, dg, gd, gn, step, t1, t2, del;
  P Ul;  // [n-1][225] whitening matrices (upper triangular)
  P Jw;  // [n-1][18][15] whitened IMU Jacobian columns of the current linearization (also setup scratch)
  P rl;  // [n-1][15] whitened IMU residuals
  VIO_AS3 int *tab;  // lower-triangle entry list of damped_solve, (i << 8) | k
  int dim;
  int off[kMaxFrames];  // first column of frame k, or -1 when the frame is constant
};

// ---- small dense helpers (one thread) -----------------------------------------------------------------------------------
template <class PM>
VIO_DEV bool chol15(PM A) {  // in place lower Cholesky of a 15x15 SPD matrix
  for (int j = 0; j < 15; j++) {
    double d = A[j * 15 + j];
    for (int k = 0; k < j; k++) d -= A[j * 15 + k] * A[j * 15 + k];
    if (!(d > 0)) return false;
    d = sqrt(d);
    A[j * 15 + j] = d;
    for (int i = j + 1; i < 15; i++) {
      double s = A[i * 15 + j];
      for (int k = 0; k < j; k++) s -= A[i * 15 + k] * A[j * 15 + k];
      A[i * 15 + j] = s / d;
    }
  }
  return true;
}

// sqrt_info = LLT(covariance^-1).matrixL().transpose() (imu_factor_pnp.h:72): U upper triangular with U^T U = cov^-1.
template <class PM>
VIO_DEV bool imu_sqrt_info(const double *cov, PM U /* 225 */, PM tmp /* 225 */) {
  // cov = C C^T  ->  cov^-1 = C^-T C^-1
  for (int i = 0; i < 225; i++) tmp[i] = cov[i];
  if (!chol15(tmp)) return false;
  // Cinv (lower) into U's storage
  for (int i = 0; i < 15; i++)
    for (int j = 0; j < 15; j++) U[i * 15 + j] = 0.0;
  for (int c = 0; c < 15; c++) {
    U[c * 15 + c] = 1.0 / tmp[c * 15 + c];
    for (int r = c + 1; r < 15; r++) {
      double s = 0;
      for (int k = c; k < r; k++) s -= tmp[r * 15 + k] * U[k * 15 + c];
      U[r * 15 + c] = s / tmp[r * 15 + r];
    }
  }
  // inv = Cinv^T Cinv into tmp (full), then its Cholesky L, U = L^T
  for (int i = 0; i < 15; i++)
    for (int j = 0; j <= i; j++) {
      double s = 0;
      for (int k = i; k < 15; k++) s += U[k * 15 + i] * U[k * 15 + j];
      tmp[i * 15 + j] = s, tmp[j * 15 + i] = s;
    }
  if (!chol15(tmp)) return false;
  for (int i = 0; i < 15; i++)
    for (int j = 0; j < 15; j++) U[i * 15 + j] = (j >= i) ? tmp[j * 15 + i] : 0.0;
  return true;
}

// PerspectiveFactor::Evaluate (perspective_factor.cpp:16-67): residual and the 2x6 Jacobian w.r.t. the pose's local
// coordinates (position, rotation); the extrinsic block is constant.
template <class PA>
VIO_DEV void perspective_eval(double s_info, PA pose, const double *ex, const double *obs, const double *pt, int track_num,
                              double r[2], double *J /* 2x6 or null */) {
  const Quat Qi{pose[3], pose[4], pose[5], pose[6]}, qic{ex[3], ex[4], ex[5], ex[6]};
  const double d[3] = {pt[0] - pose[0], pt[1] - pose[1], pt[2] - pose[2]};
  double pi[3], pc[3], t[3];
  qrot(qinv(Qi), d, pi);
  for (int k = 0; k < 3; k++) t[k] = pi[k] - ex[k];
  qrot(qinv(qic), t, pc);
  const double dep = pc[2], wgt = s_info * (double)track_num / 10.0;
  r[0] = wgt * (pc[0] / dep - obs[0]), r[1] = wgt * (pc[1] / dep - obs[1]);
  if (!J) return;
  double Ri[9], Rc[9], RcT[9], RiT[9], A[9], S[9], B[9];
  qtoR(Qi, Ri), qtoR(qic, Rc);
  mat3T(Rc, RcT), mat3T(Ri, RiT);
  mat3mul(RcT, RiT, A);  // -A = d pc / d P
  skew3(pi, S);
  mat3mul(RcT, S, B);    // d pc / d theta
  const double red[6] = {wgt / dep, 0, -wgt * pc[0] / (dep * dep), 0, wgt / dep, -wgt * pc[1] / (dep * dep)};
  for (int a = 0; a < 2; a++)
    for (int c = 0; c < 3; c++) {
      J[a * 6 + c] = -(red[a * 3] * A[c] + red[a * 3 + 1] * A[3 + c] + red[a * 3 + 2] * A[6 + c]);
      J[a * 6 + 3 + c] = red[a * 3] * B[c] + red[a * 3 + 1] * B[3 + c] + red[a * 3 + 2] * B[6 + c];
    }
}

// ---- one evaluation: cost, and with want_lin the normal equations H, g ------------------------------------------------------
template <class P>
VIO_DEV double evaluate(const Ctx &cx, const View &v, Work<P> &w, P pose, P speed, bool want_lin) {
  const int dim = w.dim;
  if (want_lin) {
    VIO_PARFOR(i, dim * dim) w.H[i] = 0.0;
    VIO_PARFOR(i, dim) w.g[i] = 0.0;
    VIO_SYNC();
  }
  double part = 0;
  // IMU factors. Phase 1, one lane per factor: raw residual / Jacobian (imu_factor_pnp.h:68-71), whitened residual.
  VIO_PARFOR(k, v.n - 1) {
    double sbi[9], sbj[9], pi7[7], pj7[7], res[15];
    for (int c = 0; c < 7; c++) pi7[c] = pose[7 * k + c], pj7[c] = pose[7 * (k + 1) + c];
    for (int c = 0; c < 3; c++) sbi[c] = speed[3 * k + c], sbj[c] = speed[3 * (k + 1) + c];
    for (int c = 0; c < 6; c++) sbi[3 + c] = v.bias[6 * k + c], sbj[3 + c] = v.bias[6 * (k + 1) + c];
    imu_eval_raw(v.gravity, v.preint + (size_t)k * kPreDoubles, pi7, sbi, pj7, sbj, res, want_lin ? v.Jraw + (size_t)k * 450 : nullptr);
    for (int a = 0; a < 15; a++) {
      double s = 0;
      for (int b = a; b < 15; b++) s += w.Ul[k * 225 + a * 15 + b] * res[b];
      if (want_lin) w.rl[k * 15 + a] = s;
      part += 0.5 * s * s;
    }
  }
  if (want_lin) {
    VIO_SYNC();
    // Phase 2, one lane per (factor, column): the whitened column. Columns of a factor in the order pose_i, speed_i,
    // pose_j, speed_j; those of constant frames are skipped (col < 0).
    auto column_of = [&](int k, int a, int *src) {
      const int fr = a < 9 ? k : k + 1, c = a < 9 ? a : a - 9;
      *src = (a < 9 ? 0 : 15) + (c < 6 ? c : c);  // pose 0..5, speed = first 3 of the speed-bias part (6..8)
      return w.off[fr] < 0 ? -1 : w.off[fr] + c;
    };
    VIO_PARFOR(item, (v.n - 1) * 18) {
      const int k = item / 18, a = item - k * 18;
      int src;
      if (column_of(k, a, &src) < 0) continue;
      const double *Jraw = v.Jraw + (size_t)k * 450;
      double ga = 0;
      for (int i = 0; i < 15; i++) {
        double s2 = 0;
        for (int b = i; b < 15; b++) s2 += w.Ul[k * 225 + i * 15 + b] * Jraw[b * 30 + src];
        w.Jw[(k * 18 + a) * 15 + i] = s2;
        ga += s2 * w.rl[k * 15 + i];
      }
      VIO_ATOMIC_ADD(&w.g[column_of(k, a, &src)], ga);
    }
    VIO_SYNC();
    // Phase 3, one lane per (factor, column pair): J^T J into the lower triangle of H.
    VIO_PARFOR(item, (v.n - 1) * 171) {
      const int k = item / 171, pr = item - k * 171;
      int a = 0;
      while ((a + 1) * (a + 2) / 2 <= pr) a++;
      const int b = pr - a * (a + 1) / 2;
      int sa, sb2;
      const int ca = column_of(k, a, &sa), cb = column_of(k, b, &sb2);
      if (ca < 0 || cb < 0) continue;
      double s2 = 0;
      for (int i = 0; i < 15; i++) s2 += w.Jw[(k * 18 + a) * 15 + i] * w.Jw[(k * 18 + b) * 15 + i];
      const int hi = ca > cb ? ca : cb, lo = ca > cb ? cb : ca;
      VIO_ATOMIC_ADD(&w.H[hi * dim + lo], s2);
    }
  }
  // perspective factors: all lanes; frame of factor m by a short search in feat_start
  VIO_PARFOR(m, v.M) {
    int k = 0;
    while (k + 1 < v.n && m >= v.feat_start[k + 1]) k++;
    double r[2], J[12];
    const bool lin = want_lin && w.off[k] >= 0;
    perspective_eval(v.s_info, pose + 7 * k, v.ex, v.obs + 2 * (size_t)m, v.pos + 3 * (size_t)m, v.track[m], r, lin ? J : nullptr);
    // CauchyLoss(a): rho = b log(1 + s/b) (CSI/loss_function.cc:72-79); rho'' < 0 always -> the Corrector scales residual
    // and Jacobian by sqrt(rho') (CSI/corrector.cc:81-85)
    const double sq = r[0] * r[0] + r[1] * r[1], sum = 1.0 + sq / v.cauchy_b, inv = 1.0 / sum;
    part += 0.5 * v.cauchy_b * log(sum);
    if (lin) {
      const double rho1 = inv > 2.2250738585072014e-308 ? inv : 2.2250738585072014e-308, sr = sqrt(rho1);
      for (int q = 0; q < 12; q++) J[q] *= sr;
      r[0] *= sr, r[1] *= sr;
      const int o = w.off[k];
      for (int a = 0; a < 6; a++) {
        VIO_ATOMIC_ADD(&w.g[o + a], J[a] * r[0] + J[6 + a] * r[1]);
        for (int b = 0; b <= a; b++) VIO_ATOMIC_ADD(&w.H[(o + a) * dim + o + b], J[a] * J[b] + J[6 + a] * J[6 + b]);
      }
    }
  }
  const double cost = block_sum(cx, part);
  VIO_SYNC();
  return cost;
}

// y = S H S v for the symmetric H stored in its lower triangle.
template <class P>
VIO_DEV void scaled_Hv(const Ctx &cx, Work<P> &w, P vin, P y) {
  const int dim = w.dim;
  VIO_PARFOR(i, dim) {
    double s = 0;
    for (int j = 0; j < dim; j++) s += (j <= i ? w.H[i * dim + j] : w.H[j * dim + i]) * w.sc[j] * vin[j];
    y[i] = w.sc[i] * s;
  }
  VIO_SYNC();
}

// Solves (S H S + diag(D^2) mu) y = S g; y into w.t1. Returns false when a pivot is not positive (the condition under
// which Ceres' dense Cholesky fails). Right-looking LDL^T with the column left unscaled (no square roots) and the
// right-hand side carried as one more row, so the forward substitution is part of the elimination; the lower-triangle
// entries (i, k) are listed once in w.tab, columns last to first: the entries step j updates (k > j) are a prefix of that
// list and all lanes share them evenly. The back substitution runs column by column.
template <class P>
VIO_DEV bool damped_solve(const Ctx &cx, Work<P> &w, double mu) {
  const int dim = w.dim;
  VIO_PARFOR(e, dim * dim) {
    const int i = e / dim, j = e - i * dim;
    if (j > i) continue;
    double a = w.sc[i] * w.H[i * dim + j] * w.sc[j];
    if (i == j) a += w.dg[i] * w.dg[i] * mu;
    w.Lf[i * dim + j] = a;
  }
  VIO_PARFOR(i, dim) w.t1[i] = w.sc[i] * w.g[i];
  VIO_SYNC();
  bool ok = true;
  for (int j = 0; j < dim; j++) {
    const double piv = w.Lf[j * dim + j];
    if (!(piv > 0) || !(piv < 1e300)) {
      ok = false;
      break;
    }
    const double ip = 1.0 / piv, bj = w.t1[j];
    const int rem = dim - j;  // rows below j plus the rhs row
    VIO_PARFOR(q, rem * (rem + 1) / 2 - 1) {
      const int ik = w.tab[q], i = ik >> 8, k = ik & 0xff;
      const double akj = w.Lf[k * dim + j] * ip;
      if (i < dim) w.Lf[i * dim + k] -= w.Lf[i * dim + j] * akj;
      else w.t1[k] -= bj * akj;
    }
    VIO_SYNC();
  }
  if (!ok) return false;
  for (int j = dim - 1; j >= 0; j--) {  // x_j = b_j / piv_j, then b_i -= a_ji x_j for the rows above
    const double xj = w.t1[j] / w.Lf[j * dim + j];
    VIO_SYNC();
    VIO_PARFOR(i, j + 1) {
      if (i == j) w.t1[j] = xj;
      else w.t1[i] -= w.Lf[j * dim + i] * xj;
    }
    VIO_SYNC();
  }
  return true;
}

// cand = Plus(x, delta) on the free blocks (pose: PoseLocalParameterization, pose_local_parameterization.cpp:11-27).
template <class P>
VIO_DEV void plus(const Ctx &cx, const View &v, Work<P> &w, P delta) {
  VIO_PARFOR(k, v.n) {
    for (int c = 0; c < 7; c++) w.cp[7 * k + c] = w.xp[7 * k + c];
    for (int c = 0; c < 3; c++) w.cs[3 * k + c] = w.xs[3 * k + c];
    const int o = w.off[k];
    if (o < 0) continue;
    for (int c = 0; c < 3; c++) w.cp[7 * k + c] += delta[o + c], w.cs[3 * k + c] += delta[o + 6 + c];
    const Quat q{w.xp[7 * k + 3], w.xp[7 * k + 4], w.xp[7 * k + 5], w.xp[7 * k + 6]};
    const Quat dq{delta[o + 3] / 2.0, delta[o + 4] / 2.0, delta[o + 5] / 2.0, 1.0};
    const Quat qn = qnormalized(qmul(q, dq));
    w.cp[7 * k + 3] = qn.x, w.cp[7 * k + 4] = qn.y, w.cp[7 * k + 5] = qn.z, w.cp[7 * k + 6] = qn.w;
  }
  VIO_SYNC();
}

// |x|, |x - cand|_2 and |x - cand|_inf over the free blocks in their ambient coordinates (thread 0 would do; the sums
// are tiny, every lane computes them redundantly from LDS).
template <class P>
VIO_DEV void norms(const View &v, const Work<P> &w, double *xnorm, double *d2, double *dinf) {
  double a = 0, b = 0, c = 0;
  for (int k = 0; k < v.n; k++) {
    if (w.off[k] < 0) continue;
    for (int q = 0; q < 7; q++) {
      const double x = w.xp[7 * k + q], d = x - w.cp[7 * k + q];
      a += x * x, b += d * d, c = fmax(c, fabs(d));
    }
    for (int q = 0; q < 3; q++) {
      const double x = w.xs[3 * k + q], d = x - w.cs[3 * k + q];
      a += x * x, b += d * d, c = fmax(c, fabs(d));
    }
  }
  if (xnorm) *xnorm = sqrt(a);
  if (d2) *d2 = sqrt(b);
  if (dinf) *dinf = c;
}

template <class P>
VIO_DEV void solve(const Ctx &cx, const View &v, Work<P> &w) {
  // ---- setup: layout of the free blocks, iterate, whitening matrices
  int dim = 0;
  for (int k = 0; k < v.n; k++) w.off[k] = v.fixed[k] ? -1 : (dim += kDof) - kDof;
  w.dim = dim;
  // entry list of damped_solve: column k = dim-1 .. 1, rows i = k .. dim (row dim = the right-hand side)
  VIO_PARFOR(k, dim) {
    if (k < 1) continue;
    const int off = (dim - k) * (dim - k + 1) / 2 - 1;
    for (int i = k; i <= dim; i++) w.tab[off + i - k] = (i << 8) | k;
  }
  VIO_PARFOR(i, 7 * v.n) w.xp[i] = v.pose0[i];
  VIO_PARFOR(i, 3 * v.n) w.xs[i] = v.speed0[i];
  VIO_PARFOR(k, v.n - 1) {
    if (!imu_sqrt_info(v.preint + (size_t)k * kPreDoubles + 17 + 225, w.Ul + k * 225, w.Jw + k * 270))
      for (int i = 0; i < 225; i++) w.Ul[k * 225 + i] = (i % 16 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 450; i++) v.Jraw[(size_t)k * 450 + i] = 0.0;
  }
  VIO_SYNC();
  double *sd = v.stats_d;
  int *si = v.stats_i;
  auto record = [&](int i, double cost, double radius, double step_norm, double rel, double gmax, bool valid, bool ok) {
    if (cx.tid == 0 && i < kMaxTrace) {
      sd[4 + i] = cost, sd[4 + kMaxTrace + i] = radius, sd[4 + 2 * kMaxTrace + i] = step_norm;
      sd[4 + 3 * kMaxTrace + i] = rel, sd[4 + 4 * kMaxTrace + i] = gmax;
      si[4 + i] = (valid ? 1 : 0) | (ok ? 2 : 0);
    }
  };
  auto grad_max_norm = [&]() {  // |x - Plus(x, -g)|_inf (trust_region_minimizer.cc:270-284)
    VIO_PARFOR(i, dim) w.t2[i] = -w.g[i];
    VIO_SYNC();
    plus(cx, v, w, w.t2);
    double linf;
    norms(v, w, nullptr, nullptr, &linf);
    VIO_SYNC();
    return linf;
  };
  double x_cost = evaluate(cx, v, w, w.xp, w.xs, dim > 0);
  int it = 0, n_ok = 1, n_bad = 0, invalid_run = 0, termination = 0, recorded = 1;
  double min_rec = x_cost;
  if (cx.tid == 0) sd[0] = x_cost;
  if (dim == 0) {  // every block constant: Ceres returns before its first iteration record
    termination = 1, recorded = 0, n_ok = 0;
  } else {
    double x_norm = -1.0;  // "Invalid value", trust_region_minimizer.cc:168
    VIO_PARFOR(i, dim) w.sc[i] = 1.0 / (1.0 + sqrt(w.H[i * dim + i]));  // Jacobi scaling, :239-254
    VIO_SYNC();
    double gmax = grad_max_norm();
    double radius = 1e4, mu = 1e-8;
    const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
    bool reuse = false, last_ok = true;
    double dogleg_step_norm = 0, alpha = 0;
    double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost, ev_acc_ref = 0, ev_acc_cand = 0;
    record(0, x_cost, radius, 0, 0, gmax, true, true);
    while (true) {
      if (it >= v.max_iter) break;
      if (last_ok && gmax <= 1e-10) { termination = 1; break; }
      if (radius <= 1e-32) { termination = 1; break; }
      it++;
      bool solver_ok = true;
      if (!reuse) {  // DoglegStrategy::ComputeStep (dogleg_strategy.cc:77-163)
        reuse = true;
        double gsq = 0;
        VIO_PARFOR(i, dim) {
          const double c = w.sc[i] * w.sc[i] * w.H[i * dim + i];
          const double d = sqrt(fmin(fmax(c, 1e-6), 1e32));
          w.dg[i] = d;
          const double gg = w.sc[i] * w.g[i] / d;
          w.gd[i] = gg;
          w.t2[i] = gg / d;
          gsq += gg * gg;
        }
        gsq = block_sum(cx, gsq);
        VIO_SYNC();
        scaled_Hv(cx, w, w.t2, w.step);
        double jg = 0;
        VIO_PARFOR(i, dim) jg += w.t2[i] * w.step[i];
        jg = block_sum(cx, jg);
        VIO_SYNC();
        alpha = gsq / jg;  // Cauchy point (:172-192)
        solver_ok = false;
        while (mu < max_mu) {  // Gauss-Newton step with D = diag * sqrt(mu) (:515-612)
          if (damped_solve(cx, w, mu)) { solver_ok = true; break; }
          mu *= mu_inc;
        }
        if (solver_ok) {
          VIO_PARFOR(i, dim) w.gn[i] = w.t1[i] * -w.dg[i];
          VIO_SYNC();
        }
      }
      bool step_valid = false;
      double model_cost_change = 0;
      if (solver_ok) {  // ComputeTraditionalDoglegStep (:199-255)
        double a = 0, b = 0, c = 0;
        VIO_PARFOR(i, dim) a += w.gd[i] * w.gd[i], b += w.gn[i] * w.gn[i], c += w.gd[i] * w.gn[i];
#ifndef VIO_EMUL
        block_sum3(cx, a, b, c);
        VIO_SYNC();
#endif
        const double gradient_norm = sqrt(a), gauss_newton_norm = sqrt(b), gdot = c;
        double ca = 0, cb = 0;
        if (gauss_newton_norm <= radius) {
          ca = 0, cb = 1, dogleg_step_norm = gauss_newton_norm;
        } else if (gradient_norm * alpha >= radius) {
          ca = -(radius / gradient_norm), cb = 0, dogleg_step_norm = radius;
        } else {
          const double b_dot_a = -alpha * gdot;
          const double a_sq = pow(alpha * gradient_norm, 2.0);
          const double bma_sq = a_sq - 2 * b_dot_a + pow(gauss_newton_norm, 2);
          const double cc = b_dot_a - a_sq;
          const double dd = sqrt(cc * cc + bma_sq * (pow(radius, 2.0) - a_sq));
          const double beta = (cc <= 0) ? (dd - cc) / bma_sq : (radius * radius - a_sq) / (dd + cc);
          ca = -alpha * (1.0 - beta), cb = beta;
          dogleg_step_norm = -1;
        }
        double n2 = 0;
        VIO_PARFOR(i, dim) {
          const double s = ca * w.gd[i] + cb * w.gn[i];
          n2 += s * s;
          w.step[i] = s / w.dg[i];
        }
        n2 = block_sum(cx, n2);
        VIO_SYNC();
        if (dogleg_step_norm < 0) dogleg_step_norm = sqrt(n2);
        // model_cost_change = -(J step)^T (r + J step / 2) (trust_region_minimizer.cc:402-416)
        scaled_Hv(cx, w, w.step, w.t2);
        double sg = 0, shs = 0, dummy = 0;
        VIO_PARFOR(i, dim) sg += w.step[i] * w.sc[i] * w.g[i], shs += w.step[i] * w.t2[i];
#ifndef VIO_EMUL
        block_sum3(cx, sg, shs, dummy);
        VIO_SYNC();
#endif
        (void)dummy;
        model_cost_change = -sg - 0.5 * shs;
        step_valid = model_cost_change > 0.0;
      }
      if (!step_valid) {  // HandleInvalidStep (:429-462)
        if (++invalid_run >= 5) { termination = 2; break; }
        mu *= mu_inc;
        reuse = false, last_ok = false;
        n_bad++;
        record(it, x_cost, radius, 0, 0, gmax, false, false);
        recorded = it + 1;
        continue;
      }
      invalid_run = 0;
      VIO_PARFOR(i, dim) w.del[i] = w.step[i] * w.sc[i];
      VIO_SYNC();
      plus(cx, v, w, w.del);
      double cand_cost = evaluate(cx, v, w, w.cp, w.cs, false);
      if (!(fabs(cand_cost) < 1.7e308)) cand_cost = 1.7976931348623157e308;
      double step_norm;
      norms(v, w, nullptr, &step_norm, nullptr);
      if (step_norm <= 1e-8 * (x_norm + 1e-8)) { termination = 1; break; }       // ParameterToleranceReached
      const double cost_change = x_cost - cand_cost;
      if (fabs(cost_change) <= 1e-6 * x_cost) { termination = 1; break; }          // FunctionToleranceReached
      const double rel = (ev_cur - cand_cost) / model_cost_change;
      const double hist = (ev_ref - cand_cost) / (ev_acc_ref + model_cost_change);
      const double rho = fmax(rel, hist);
      if (rho > 1e-3) {
        VIO_SYNC();
        VIO_PARFOR(i, 7 * v.n) w.xp[i] = w.cp[i];
        VIO_PARFOR(i, 3 * v.n) w.xs[i] = w.cs[i];
        VIO_SYNC();
        // x_.norm() of the accepted point
        {
          double a = 0;
          for (int k = 0; k < v.n; k++) {
            if (w.off[k] < 0) continue;
            for (int q = 0; q < 7; q++) a += w.xp[7 * k + q] * w.xp[7 * k + q];
            for (int q = 0; q < 3; q++) a += w.xs[3 * k + q] * w.xs[3 * k + q];
          }
          x_norm = sqrt(a);
        }
        x_cost = evaluate(cx, v, w, w.xp, w.xs, true);
        gmax = grad_max_norm();
        if (rho < 0.25) radius *= 0.5;  // DoglegStrategy::StepAccepted (:614-629)
        if (rho > 0.75) radius = fmax(radius, 3.0 * dogleg_step_norm);
        mu = fmax(min_mu, 2.0 * mu / mu_inc);
        reuse = false;
        ev_cur = cand_cost, ev_acc_cand += model_cost_change, ev_acc_ref += model_cost_change;
        if (ev_cur < ev_min) ev_min = ev_cur, ev_cand = ev_cur, ev_acc_cand = 0;
        else if (ev_cur > ev_cand) ev_cand = ev_cur, ev_acc_cand = 0;
        ev_ref = ev_cand, ev_acc_ref = ev_acc_cand;
        last_ok = true;
        n_ok++;
        record(it, x_cost, radius, step_norm, rho, gmax, true, true);
        recorded = it + 1;
        min_rec = fmin(min_rec, x_cost);
      } else {
        radius *= 0.5;  // StepRejected (:631-634)
        reuse = true, last_ok = false;
        n_bad++;
        record(it, cand_cost, radius, step_norm, rho, 0.0, true, false);
        recorded = it + 1;
        min_rec = fmin(min_rec, cand_cost);
      }
    }
  }
  // new2old (:137-172): no gauge correction in the PnP window, the states are taken as solved
  VIO_SYNC();
  VIO_PARFOR(k, v.n) {
    const Quat q = qnormalized(Quat{w.xp[7 * k + 3], w.xp[7 * k + 4], w.xp[7 * k + 5], w.xp[7 * k + 6]});
    for (int c = 0; c < 3; c++) v.out_pose[7 * k + c] = w.xp[7 * k + c], v.out_speed[3 * k + c] = w.xs[3 * k + c];
    v.out_pose[7 * k + 3] = q.x, v.out_pose[7 * k + 4] = q.y, v.out_pose[7 * k + 5] = q.z, v.out_pose[7 * k + 6] = q.w;
  }
  if (cx.tid == 0) {
    sd[1] = min_rec;
    si[0] = recorded, si[1] = termination, si[2] = n_ok, si[3] = n_bad;
  }
}

// LDS / workspace carving: doubles needed for n frames.
template <class P>
VIO_HD size_t carve(int n, int nthreads, P base, Work<P> *w, Ctx *cx) {
  size_t o = 0;
  auto take = [&](size_t cnt) {
    P p = base + o;
    o += (cnt + 1) & ~(size_t)1;
    return p;
  };
  const size_t dim = (size_t)n * kDof;
  P red = take(6 * ((size_t)nthreads / 64) + 2);
  P xp = take(7 * (size_t)n), xs = take(3 * (size_t)n), cp = take(7 * (size_t)n), cs = take(3 * (size_t)n);
  P H = take(dim * dim), Lf = take(dim * dim);
  P g = take(dim), sc = take(dim), dg = take(dim), gd = take(dim), gn = take(dim), step = take(dim), t1 = take(dim), t2 = take(dim),
    del = take(dim);
  P Ul = take(225 * (size_t)(n - 1)), Jw = take(270 * (size_t)(n - 1)), rl = take(15 * (size_t)(n - 1));
  P tab = take(((dim + 1) * (dim + 2) / 2 + 1) / 2 + 1);
  if (w) {
    w->Ul = Ul, w->Jw = Jw, w->rl = rl;
    w->tab = reinterpret_cast<VIO_AS3 int *>(tab);
    w->xp = xp, w->xs = xs, w->cp = cp, w->cs = cs, w->H = H, w->Lf = Lf;
    w->g = g, w->sc = sc, w->dg = dg, w->gd = gd, w->gn = gn, w->step = step, w->t1 = t1, w->t2 = t2, w->del = del;
  }
  if (cx) cx->red = red;
  return o * sizeof(double);
}

}  // namespace pnp
}  // namespace vio
