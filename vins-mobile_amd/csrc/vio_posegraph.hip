// vio_posegraph.hip — the 4-DoF loop pose graph solve (second half):
// the ceres::Solve of KeyFrameDatabase::optimize4DoFLoopPoseGraph (VINS_ios/loop/keyfame_database.cpp:140-300) as a gfx950
// kernel, one workgroup per pose graph, many graphs (sequences) per launch.
//
// What the reference builds (keyfame_database.cpp:150-285) and Ceres 1.12 then runs:
//   unknowns   per kept keyframe: yaw in degrees (AngleLocalParameterization, keyfame_database.h:74-90) and translation;
//              the earliest_loop_index keyframe is constant; keyframes without an edge drop out of the program
//   residuals  FourDOFError (keyfame_database.h:271-313) under HuberLoss(1.0) to up to five predecessors,
//              FourDOFWeightError (:315-366) without loss for every loop closure
//   solver     trust region, Levenberg-Marquardt (CSI/levenberg_marquardt_strategy.cc:66-163), Jacobi scaling,
//              max_num_iterations = 5, DENSE_SCHUR = an exact solve of (J^T J + D^T D) y = J^T r
// Here: H = J^T J (analytic Jacobians, loss-corrected rows) is accumulated once per linearization in a dense row-major
// matrix in global memory, of which only the ENVELOPE is ever touched: in keyframe order the sequential edges give a band
// of 5 keyframes = 20 unknowns, loop edges give single long rows. The LM system S H S + D^2 is factored in 16 x 16 tiles on
// the matrix cores (the tile kernels of the marginalization, marg_core.h) with tile-level envelope skipping: panel k only
// visits the tile rows whose envelope reaches column k. Vectors (iterate, candidate, gradient, scaling, right-hand side)
// live in LDS. The trust-region loop is the one of solver_core.h minimize() with the LM strategy in place of dogleg.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "vio_amd.h"
#include "vio_device.h"
#include "solver_core.h"
#include "batch.h"
#include "marg_core.h"

using namespace vio;

namespace {

#define HIP_OK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "vio_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VIO_ENODEV;                                                                   \
    }                                                                                      \
  } while (0)

constexpr int kPgThreads = 512;
constexpr int kPgHdr = 4;  // n_nodes, n_edges, N, max_iterations

struct PgPtrs {
  int ld;          // leading dimension of the dense matrices = 4 * max_nodes rounded up to 16
  int node_cap, edge_cap;
  const int *hdr;           // [g][kPgHdr]
  const int *col;           // [g][node_cap]: node -> first of its 4 unknowns, -1 constant / not in the program
  const double *node0;      // [g][node_cap][4]: yaw, t of every node as given
  const int *edge_i, *edge_j, *edge_kind;  // [g][edge_cap]
  const double *meas;       // [g][edge_cap][6]
  double *H, *A;            // [g][ld * ld]
  double *xout;             // [g][ld]
  double *stats_d;          // [g][kStatsDoubles]
  int *stats_i;             // [g][kStatsInts]
};

struct PgView {
  int n_nodes, n_edges, N, nt, ld, max_iter;
  const int *col, *ei, *ej, *ek;
  const double *node0, *meas;
  double *H, *A;
  ldsd x, xc, g, scale, diag, bm, ldinv;
  ldsi ft, list, cnt;
};

__device__ __forceinline__ double pg_normalize_angle(double a) {  // keyfame_database.h:62-72
  const double two_pi = 2.0 * 180;
  if (a > 0) return a - two_pi * floor((a + 180.0) / two_pi);
  return a + two_pi * floor((-a + 180.0) / two_pi);
}

// One residual block at the state xs: loss-corrected r[4] and (jac) J[4][8] over [yaw_i, t_i, yaw_j, t_j]; returns rho / 2.
__device__ __forceinline__ double pg_edge(const PgView &v, cldsd xs, int e, bool jac, double r[4], double J[4][8]) {
  const double kPi = 3.14159265358979323846;
  const int i = v.ei[e], j = v.ej[e], ci = v.col[i], cj = v.col[j];
  const double *m = v.meas + 6 * e;
  double yi, yj, ti[3], tj[3];
  if (ci >= 0) yi = xs[ci], ti[0] = xs[ci + 1], ti[1] = xs[ci + 2], ti[2] = xs[ci + 3];
  else yi = v.node0[4 * i], ti[0] = v.node0[4 * i + 1], ti[1] = v.node0[4 * i + 2], ti[2] = v.node0[4 * i + 3];
  if (cj >= 0) yj = xs[cj], tj[0] = xs[cj + 1], tj[1] = xs[cj + 2], tj[2] = xs[cj + 3];
  else yj = v.node0[4 * j], tj[0] = v.node0[4 * j + 1], tj[1] = v.node0[4 * j + 2], tj[2] = v.node0[4 * j + 3];
  // YawPitchRollToRotationMatrix (keyfame_database.h:228-246)
  const double y = yi / 180.0 * kPi, p = m[4] / 180.0 * kPi, rr = m[5] / 180.0 * kPi;
  const double cy = cos(y), sy = sin(y), cp = cos(p), sp = sin(p), cr = cos(rr), sr = sin(rr);
  double R[9];
  R[0] = cy * cp, R[1] = -sy * cr + cy * sp * sr, R[2] = sy * sr + cy * sp * cr;
  R[3] = sy * cp, R[4] = cy * cr + sy * sp * sr, R[5] = -cy * sr + sy * sp * cr;
  R[6] = -sp, R[7] = cp * sr, R[8] = cp * cr;
  const double d[3] = {tj[0] - ti[0], tj[1] - ti[1], tj[2] - ti[2]};
#pragma unroll
  for (int k = 0; k < 3; k++) r[k] = R[0 + k] * d[0] + R[3 + k] * d[1] + R[6 + k] * d[2] - m[k];
  r[3] = pg_normalize_angle(yj - yi - m[3]);
  if (jac) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      J[k][0] = (-R[3 + k] * d[0] + R[0 + k] * d[1]) * (kPi / 180.0);
      J[k][4] = 0.0;
#pragma unroll
      for (int c = 0; c < 3; c++) J[k][1 + c] = -R[3 * c + k], J[k][5 + c] = R[3 * c + k];
    }
#pragma unroll
    for (int c = 0; c < 8; c++) J[3][c] = 0.0;
    J[3][0] = -1.0, J[3][4] = 1.0;
  }
  double w = 1.0, cost;
  if (v.ek[e] == 1) {  // FourDOFWeightError: weight 10 on the translation rows, weight / 10 = 1 on the yaw row; no loss
#pragma unroll
    for (int k = 0; k < 3; k++) {
      r[k] *= 10.0;
      if (jac)
#pragma unroll
        for (int c = 0; c < 8; c++) J[k][c] *= 10.0;
    }
    cost = 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  } else {  // HuberLoss(1.0) + Corrector with rho'' <= 0 (CSI/loss_function.cc:47-61, CSI/corrector.cc:48-113)
    const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
    if (s > 1.0) {
      const double rt = sqrt(s);
      cost = 0.5 * (2.0 * rt - 1.0);
      w = sqrt(fmax(2.2250738585072014e-308, 1.0 / rt));
#pragma unroll
      for (int k = 0; k < 4; k++) {
        r[k] *= w;
        if (jac)
#pragma unroll
          for (int c = 0; c < 8; c++) J[k][c] *= w;
      }
    } else {
      cost = 0.5 * s;
    }
  }
  return cost;
}

// cost at xs; jac: also H (lower triangle, envelope zeroed first) and g
__device__ double pg_evaluate(const Ctx &cx, const PgView &v, cldsd xs, bool jac) {
  const int N = v.N, ld = v.ld;
  if (jac) {
    VIO_PARFOR(c, N) v.g[c] = 0.0;
    const int tid_ = VIO_TID(cx), wave = tid_ >> 6, lane = tid_ & 63, nw = cx.nt >> 6;
    for (int r = wave; r < N; r += nw)
      for (int c = 16 * v.ft[r >> 4] + lane; c <= r; c += 64) v.H[(size_t)r * ld + c] = 0.0;
    VIO_SYNC();
  }
  double cost = 0.0;
  VIO_PARFOR(e, v.n_edges) {
    const int ci = v.col[v.ei[e]], cj = v.col[v.ej[e]];
    if (ci < 0 && cj < 0) continue;  // constants only: not part of the reduced program
    double r[4], J[4][8];
    cost += pg_edge(v, xs, e, jac, r, J);
    if (!jac) continue;
#pragma unroll
    for (int a = 0; a < 8; a++) {
      const int ca = a < 4 ? ci : cj;
      if (ca < 0) continue;
      const int ra = ca + (a & 3);
      VIO_ATOMIC_ADD(v.g + ra, J[0][a] * r[0] + J[1][a] * r[1] + J[2][a] * r[2] + J[3][a] * r[3]);
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const int cb = b < 4 ? ci : cj;
        if (cb < 0) continue;
        const int rb = cb + (b & 3);
        if (rb > ra) continue;
        const double s = J[0][a] * J[0][b] + J[1][a] * J[1][b] + J[2][a] * J[2][b] + J[3][a] * J[3][b];
        if (s != 0.0) VIO_ATOMIC_ADD(v.H + (size_t)ra * ld + rb, s);
      }
    }
  }
  return block_sum(cx, cost);  // (barrier inside: H and g are complete for every thread afterwards)
}

// A = S H S + diag(D2) on the envelope, bm = S g; factorization in 16 x 16 tiles, forward and back substitution.
// On return bm = (A)^-1 S g. false: a pivot <= 0.
__device__ bool pg_solve(const Ctx &cx, const PgView &v, double radius) {
  const int N = v.N, ld = v.ld, nt = v.nt;
  const int tid_ = VIO_TID(cx), wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), lane = tid_ & 63, nw = cx.nt >> 6;
  for (int r = wave; r < N; r += nw) {
    const double sr = v.scale[r];
    for (int c = 16 * v.ft[r >> 4] + lane; c <= r; c += 64) {
      double a = sr * v.H[(size_t)r * ld + c] * v.scale[c];
      if (c == r) {
        const double lm = sqrt(v.diag[r] / radius);  // lm_diagonal_ (levenberg_marquardt_strategy.cc:91)
        a += lm * lm;
      }
      v.A[(size_t)r * ld + c] = a;
    }
  }
  VIO_PARFOR(c, 16 * nt) v.bm[c] = c < N ? v.scale[c] * v.g[c] : 0.0, v.ldinv[c] = 0.0;
  VIO_SYNC();
  auto tile = [&](int ti, int tj) { return v.A + (size_t)(16 * ti) * ld + 16 * tj; };
  auto rows_of = [&](int ti) { return N - 16 * ti < 16 ? N - 16 * ti : 16; };
  for (int k = 0; k < nt; k++) {
    if (wave == 0) {
      potrf16_cut_wave(tile(k, k), tile(k, k), ld, rows_of(k), false, 0.0, v.ldinv + 16 * k, lane);
    } else if (wave == 1) {  // tile rows below k whose envelope reaches column k, in ascending order
      int n = 0;
      for (int i0 = k + 1; i0 < nt; i0 += 64) {
        const int i = i0 + lane;
        const bool act = i < nt && v.ft[i] <= k;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(act);
        if (act) v.list[n + __builtin_popcountll(m & ((1ull << lane) - 1))] = i;
        n += __builtin_popcountll(m);
      }
      if (lane == 0) v.cnt[0] = n;
    }
    VIO_SYNC();
    const int nact = v.cnt[0];
    for (int a = wave; a < nact; a += nw) {
      const int i = v.list[a];
      dtile_trsm(tile(i, k), tile(k, k), v.ldinv + 16 * k, ld, rows_of(i), lane);
    }
    if (wave == nw - 1) dtile_forward_diag(tile(k, k), v.ldinv + 16 * k, v.bm + 16 * k, ld, rows_of(k), lane);
    VIO_SYNC();
    for (int a = wave; a < nact; a += nw) {
      const int i = v.list[a];
      dtile_rhs_update(tile(i, k), v.bm + 16 * i, v.bm + 16 * k, ld, rows_of(i), lane);
    }
    const int npairs = nact * (nact + 1) / 2;
    for (int pr = wave; pr < npairs; pr += 2 * nw) {
      const int pr1 = pr + nw;
      const bool second = pr1 < npairs;
      auto pair_ij = [&](int p, int &ti, int &tj) {
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= p) a++;
        ti = v.list[a], tj = v.list[p - a * (a + 1) / 2];
      };
      int i0, j0, i1, j1;
      pair_ij(pr, i0, j0), pair_ij(second ? pr1 : pr, i1, j1);
      dtile_update2(tile(i0, j0), tile(i0, k), tile(j0, k), rows_of(i0), rows_of(j0), tile(i1, j1), tile(i1, k), tile(j1, k),
                    second ? rows_of(i1) : 0, rows_of(j1), ld, lane);
    }
    VIO_SYNC();
  }
  // a cut pivot (tol 0: pivot <= 0 or NaN) left 1 / L_cc = 0
  double bad = 0.0;
  VIO_PARFOR(c, N) bad = fmax(bad, v.ldinv[c] > 0.0 ? 0.0 : 1.0);
  if (block_max(cx, bad) > 0.0) return false;
  // back substitution L^T x = y, tile rows from the last: x_i = L_ii^-T y_i, then y_k -= L_ik^T x_i over the row's envelope
  for (int i = nt - 1; i >= 0; i--) {
    const int rows = rows_of(i);
    if (wave == 0) {
      const int c = lane & 15;
      auto D = tile(i, i);
      ldsd b = v.bm + 16 * i;
      double s = c < rows ? v.ldinv[16 * i + c] * b[c] : 0.0;
#pragma unroll
      for (int n = 1; n < 16; n++) {  // Linv[n][c] (n > c) sits above the diagonal at D[c][n]
        const bool ok = n > c && n < rows;
        const double l = D[(size_t)(ok ? c : 0) * ld + (ok ? n : 0)], bn = b[ok ? n : 0];
        s = fma(ok ? l : 0.0, ok ? bn : 0.0, s);
      }
      if (lane < 16 && c < rows) b[c] = s;
    }
    VIO_SYNC();
    for (int k = v.ft[i] + wave; k < i; k += nw) {  // lane = 4 c + p: column c of L_ik, the quad splits the 16 rows
      const int c = lane >> 2, p = lane & 3;
      auto Lik = tile(i, k);
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = p + 4 * q;
        const bool ok = r < rows;
        const double l = Lik[(size_t)(ok ? r : 0) * ld + c], xr = v.bm[16 * i + (ok ? r : 0)];
        s = fma(ok ? l : 0.0, ok ? xr : 0.0, s);
      }
      s = quad_sum_f64(s);
      if (p == 0) v.bm[16 * k + c] -= s;
    }
    VIO_SYNC();
  }
  return true;
}

__global__ __launch_bounds__(kPgThreads) void posegraph_kernel(PgPtrs P) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int gidx = blockIdx.x;
  const int *hdr = P.hdr + (size_t)gidx * kPgHdr;
  PgView v;
  v.n_nodes = hdr[0], v.n_edges = hdr[1], v.N = hdr[2], v.max_iter = hdr[3];
  v.nt = (v.N + 15) >> 4, v.ld = P.ld;
  v.col = P.col + (size_t)gidx * P.node_cap, v.node0 = P.node0 + (size_t)gidx * P.node_cap * 4;
  v.ei = P.edge_i + (size_t)gidx * P.edge_cap, v.ej = P.edge_j + (size_t)gidx * P.edge_cap, v.ek = P.edge_kind + (size_t)gidx * P.edge_cap;
  v.meas = P.meas + (size_t)gidx * P.edge_cap * 6;
  v.H = P.H + (size_t)gidx * P.ld * P.ld, v.A = P.A + (size_t)gidx * P.ld * P.ld;
  ldsd lds = (ldsd)smem;
  const int ld = P.ld;
  v.x = lds, v.xc = lds + ld, v.g = lds + 2 * ld, v.scale = lds + 3 * ld, v.diag = lds + 4 * ld, v.bm = lds + 5 * ld, v.ldinv = lds + 6 * ld;
  Ctx cx;
  cx.wave64 = __builtin_amdgcn_readfirstlane((int)threadIdx.x & ~63);
  cx.tid = threadIdx.x, cx.nt = blockDim.x, cx.prof = nullptr, cx.lprof = nullptr;
  cx.red = lds + 7 * ld;
  ldsi ints = reinterpret_cast<ldsi>(lds + 7 * ld + 6 * (kPgThreads / 64) + 2);
  v.ft = ints, v.list = ints + ld / 16, v.cnt = ints + 2 * (ld / 16);
  const int N = v.N;
  double *sd = P.stats_d + (size_t)gidx * kStatsDoubles;
  int *si = P.stats_i + (size_t)gidx * kStatsInts;

  // envelope at tile granularity: first tile column of every tile row
  VIO_PARFOR(t, v.nt) v.ft[t] = t;
  VIO_PARFOR(n, v.n_nodes) {
    const int c = v.col[n];
    if (c >= 0) v.x[c] = v.node0[4 * n], v.x[c + 1] = v.node0[4 * n + 1], v.x[c + 2] = v.node0[4 * n + 2], v.x[c + 3] = v.node0[4 * n + 3];
  }
  VIO_SYNC();
  VIO_PARFOR(e, v.n_edges) {
    const int ci = v.col[v.ei[e]], cj = v.col[v.ej[e]];
    if (ci < 0 || cj < 0) continue;
    const int hi = ci > cj ? ci : cj, lo = ci > cj ? cj : ci;
    __hip_atomic_fetch_min(v.ft + (hi >> 4), lo >> 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  VIO_SYNC();

  int recorded = 0, n_ok = 0, n_bad = 0, termination = 0;
  double min_recorded = 1.7976931348623157e308;
  double radius = 1e4, decrease_factor = 2.0;
  auto record = [&](int i, double cost, double step_norm, double rel, double gmax, bool valid, bool ok) {
    recorded = i + 1;
    min_recorded = fmin(min_recorded, cost);
    if (cx.tid == 0 && i < kMaxTrace) {
      sd[4 + i] = cost, sd[4 + kMaxTrace + i] = radius, sd[4 + 2 * kMaxTrace + i] = step_norm;
      sd[4 + 3 * kMaxTrace + i] = rel, sd[4 + 4 * kMaxTrace + i] = gmax;
      si[4 + i] = (valid ? 1 : 0) | (ok ? 2 : 0);
    }
  };
  auto grad_max = [&]() {  // |x - Plus(x, -g)|_inf (CSI/trust_region_minimizer.cc:270-284)
    double m = 0.0;
    VIO_PARFOR(c, N) {
      const double xv = v.x[c], gv = v.g[c];
      const double xp = (c & 3) == 0 ? pg_normalize_angle(xv - gv) : xv - gv;
      m = fmax(m, fabs(xv - xp));
    }
    return block_max(cx, m);
  };
  double x_cost = pg_evaluate(cx, v, v.x, true);
  double x_norm = -1.0;
  VIO_PARFOR(c, N) v.scale[c] = 1.0 / (1.0 + sqrt(v.H[(size_t)c * ld + c]));  // Jacobi scaling (:239-254)
  VIO_SYNC();
  double gmax = grad_max();
  bool last_ok = true, reuse_diagonal = false;
  n_ok++;
  record(0, x_cost, 0, 0, gmax, true, true);
  if (cx.tid == 0) sd[0] = x_cost;
  double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost, ev_acc_ref = 0, ev_acc_cand = 0;
  int it = 0, invalid_run = 0;
  while (N > 0) {
    if (it >= v.max_iter) break;
    if (last_ok && gmax <= 1e-10) { termination = 1; break; }
    if (radius <= 1e-32) { termination = 1; break; }
    it++;
    // ---- LevenbergMarquardtStrategy::ComputeStep
    if (!reuse_diagonal) {
      VIO_PARFOR(c, N) {
        const double s = v.scale[c];
        v.diag[c] = fmin(fmax(s * s * v.H[(size_t)c * ld + c], 1e-6), 1e32);
      }
      VIO_SYNC();
    }
    reuse_diagonal = true;
    bool solver_ok = pg_solve(cx, v, radius);
    // step = -y. model_cost_change = -step^T (gs + Hs step / 2); with (Hs + D^2) step = -gs this is
    // (-step^T gs + sum D^2 step^2) / 2: no product with the matrix
    double a = 0.0, b = 0.0, fin = 0.0;
    VIO_PARFOR(c, N) {
      const double y = v.bm[c], lm = sqrt(v.diag[c] / radius);
      a += y * (v.scale[c] * v.g[c]), b += lm * lm * y * y;
      fin = fmax(fin, isfinite(y) ? 0.0 : 1.0);
    }
    block_sum3(cx, a, b, fin);
    if (fin > 0.0) solver_ok = false;
    const double model_cost_change = 0.5 * (a + b);
    const bool step_valid = solver_ok && model_cost_change > 0.0;
    if (!step_valid) {
      if (++invalid_run >= 5) { termination = 2; break; }
      radius = radius / decrease_factor, decrease_factor *= 2.0, reuse_diagonal = true;  // StepIsInvalid -> StepRejected(0)
      last_ok = false;
      n_bad++;
      record(it, x_cost, 0, 0, gmax, false, false);
      continue;
    }
    invalid_run = 0;
    double sn = 0.0;
    VIO_PARFOR(c, N) {
      const double d = -v.bm[c] * v.scale[c], xv = v.x[c];
      const double xn = (c & 3) == 0 ? pg_normalize_angle(xv + d) : xv + d;
      v.xc[c] = xn;
      sn += (xv - xn) * (xv - xn);
    }
    sn = block_sum(cx, sn);
    double cand_cost = pg_evaluate(cx, v, v.xc, false);
    if (!isfinite(cand_cost)) cand_cost = 1.7976931348623157e308;
    const double step_norm = sqrt(sn);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { termination = 1; break; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= 1e-6 * x_cost) { termination = 1; break; }
    const double rel = (ev_cur - cand_cost) / model_cost_change;
    const double hist = (ev_ref - cand_cost) / (ev_acc_ref + model_cost_change);
    const double rho = fmax(rel, hist);
    if (rho > 1e-3) {
      double xn2 = 0.0;
      VIO_PARFOR(c, N) {
        const double xv = v.xc[c];
        v.x[c] = xv, xn2 += xv * xv;
      }
      x_norm = sqrt(block_sum(cx, xn2));
      x_cost = pg_evaluate(cx, v, v.x, true);
      gmax = grad_max();
      const double q = 2.0 * rho - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - q * q * q);  // StepAccepted (:146-153)
      radius = fmin(1e16, radius);
      decrease_factor = 2.0, reuse_diagonal = false;
      ev_cur = cand_cost, ev_acc_cand += model_cost_change, ev_acc_ref += model_cost_change;
      if (ev_cur < ev_min) ev_min = ev_cur, ev_cand = ev_cur, ev_acc_cand = 0;
      else if (ev_cur > ev_cand) ev_cand = ev_cur, ev_acc_cand = 0;
      ev_ref = ev_cand, ev_acc_ref = ev_acc_cand;
      last_ok = true;
      n_ok++;
      record(it, x_cost, step_norm, rho, gmax, true, true);
    } else {
      radius = radius / decrease_factor, decrease_factor *= 2.0, reuse_diagonal = true;  // StepRejected (:155-159)
      last_ok = false;
      n_bad++;
      record(it, cand_cost, step_norm, rho, 0.0, true, false);
    }
  }
  VIO_SYNC();
  VIO_PARFOR(c, N) P.xout[(size_t)gidx * ld + c] = v.x[c];
  if (cx.tid == 0) {
    sd[1] = min_recorded;
    si[0] = recorded, si[1] = termination, si[2] = n_ok, si[3] = n_bad;
  }
}

template <class T>
struct PgBuf {
  T *d = nullptr, *h = nullptr;
  size_t n = 0;
  int ensure(size_t count) {
    if (count <= n) return VIO_OK;
    release();
    if (hipMalloc(&d, count * sizeof(T)) != hipSuccess) return VIO_ENOMEM;
    if (hipHostMalloc(&h, count * sizeof(T), hipHostMallocDefault) != hipSuccess) return VIO_ENOMEM;
    n = count;
    return VIO_OK;
  }
  void release() {
    if (d) (void)hipFree(d);
    if (h) (void)hipHostFree(h);
    d = nullptr, h = nullptr, n = 0;
  }
};

}  // namespace

struct vio_posegraph {
  int device = 0;
  int max_nodes = 0, max_edges = 0, n_graphs = 0, ld = 0;
  size_t lds_bytes = 0;
  hipStream_t stream = nullptr;
  PgBuf<int> hdr, col, ei, ej, ek, stats_i;
  PgBuf<double> node0, meas, xout, stats_d;
  double *H = nullptr, *A = nullptr;
};

extern "C" {

int vio_posegraph_create(int32_t max_nodes, int32_t max_edges, int32_t n_graphs, vio_posegraph_t **out) {
  if (!out || max_nodes < 2 || max_edges < 1 || n_graphs < 1) return VIO_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    fprintf(stderr, "vio_amd: no HIP device visible; the pose graph solver has no CPU fallback\n");
    return VIO_ENODEV;
  }
  vio_posegraph *pg = new (std::nothrow) vio_posegraph();
  if (!pg) return VIO_ENOMEM;
  pg->device = vio::current_device();
  pg->max_nodes = max_nodes, pg->max_edges = max_edges, pg->n_graphs = n_graphs;
  pg->ld = (4 * max_nodes + 15) / 16 * 16;
  pg->lds_bytes = ((size_t)7 * pg->ld + 6 * (kPgThreads / 64) + 2) * sizeof(double) + ((size_t)2 * (pg->ld / 16) + 4) * sizeof(int);
  if (pg->lds_bytes > vio::kLdsBytes) {  // seven N-vectors in LDS: at most 160 KB / 56 B = ~2900 unknowns = ~730 keyframes
    delete pg;
    return VIO_ECAP;
  }
  const size_t G = n_graphs, mat = (size_t)pg->ld * pg->ld;
  bool ok = hipStreamCreateWithFlags(&pg->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && pg->hdr.ensure(G * kPgHdr) == VIO_OK && pg->col.ensure(G * max_nodes) == VIO_OK && pg->ei.ensure(G * max_edges) == VIO_OK &&
       pg->ej.ensure(G * max_edges) == VIO_OK && pg->ek.ensure(G * max_edges) == VIO_OK && pg->stats_i.ensure(G * kStatsInts) == VIO_OK &&
       pg->node0.ensure(G * max_nodes * 4) == VIO_OK && pg->meas.ensure(G * max_edges * 6) == VIO_OK &&
       pg->xout.ensure(G * pg->ld) == VIO_OK && pg->stats_d.ensure(G * kStatsDoubles) == VIO_OK;
  if (!ok) {  // (stream creation is the only non-allocation step above)
    const bool no_stream = pg->stream == nullptr;
    vio_posegraph_destroy(pg);
    return no_stream ? VIO_ENODEV : VIO_ENOMEM;
  }
  ok = hipMalloc(&pg->H, G * mat * sizeof(double)) == hipSuccess && hipMalloc(&pg->A, G * mat * sizeof(double)) == hipSuccess;
  if (!ok) {
    vio_posegraph_destroy(pg);
    return VIO_ENOMEM;
  }
  // the factorization reads whole 16 x 16 tiles, including the rows past N of the last tile row: no uninitialised
  // memory may reach the matrix cores (their rows are independent, but a NaN pattern need not stay that way)
  ok = hipMemsetAsync(pg->H, 0, G * mat * sizeof(double), pg->stream) == hipSuccess &&
       hipMemsetAsync(pg->A, 0, G * mat * sizeof(double), pg->stream) == hipSuccess &&
       hipStreamSynchronize(pg->stream) == hipSuccess &&
       hipFuncSetAttribute((const void *)posegraph_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vio::kLdsBytes) == hipSuccess;
  if (!ok) {
    vio_posegraph_destroy(pg);
    return VIO_ENODEV;
  }
  *out = pg;
  return VIO_OK;
}

int vio_posegraph_get_device(const vio_posegraph_t *pg, int32_t *device) {
  if (!pg || !device) return VIO_EINVAL;
  *device = pg->device;
  return VIO_OK;
}

void vio_posegraph_destroy(vio_posegraph_t *pg) {
  if (!pg) return;
  vio::DeviceScope scope(pg->device);
  if (pg->stream) (void)hipStreamSynchronize(pg->stream), (void)hipStreamDestroy(pg->stream);
  pg->hdr.release(), pg->col.release(), pg->ei.release(), pg->ej.release(), pg->ek.release(), pg->stats_i.release();
  pg->node0.release(), pg->meas.release(), pg->xout.release(), pg->stats_d.release();
  if (pg->H) (void)hipFree(pg->H);
  if (pg->A) (void)hipFree(pg->A);
  delete pg;
}

int vio_posegraph_optimize(vio_posegraph_t *pg, VioPoseGraph *graphs, int32_t n, int32_t max_iterations, VioSolveStats *stats) {
  if (!pg || !graphs || n < 1 || max_iterations < 0) return VIO_EINVAL;
  if (n > pg->n_graphs) return VIO_ECAP;
  VIO_ON_DEVICE_OF(pg);
  const int NC = pg->max_nodes, EC = pg->max_edges;
  for (int g = 0; g < n; g++) {
    const VioPoseGraph &G = graphs[g];
    if (G.n_nodes < 1 || G.n_edges < 0 || !G.t || !G.ypr || (G.n_edges > 0 && (!G.edge_i || !G.edge_j || !G.edge_kind || !G.edge_meas)))
      return VIO_EINVAL;
    if (G.n_nodes > NC || G.n_edges > EC) return VIO_ECAP;
    std::vector<char> used(G.n_nodes, 0);
    for (int e = 0; e < G.n_edges; e++) {
      if (G.edge_i[e] < 0 || G.edge_i[e] >= G.n_nodes || G.edge_j[e] < 0 || G.edge_j[e] >= G.n_nodes || G.edge_kind[e] > 1)
        return VIO_EINVAL;
      used[G.edge_i[e]] = used[G.edge_j[e]] = 1;
    }
    int *col = pg->col.h + (size_t)g * NC;
    double *node0 = pg->node0.h + (size_t)g * NC * 4;
    int nv = 0;
    for (int k = 0; k < G.n_nodes; k++) {
      col[k] = (used[k] && k != G.fixed_node) ? 4 * nv++ : -1;
      node0[4 * k] = G.ypr[3 * k], node0[4 * k + 1] = G.t[3 * k], node0[4 * k + 2] = G.t[3 * k + 1], node0[4 * k + 3] = G.t[3 * k + 2];
    }
    int *hdr = pg->hdr.h + (size_t)g * kPgHdr;
    hdr[0] = G.n_nodes, hdr[1] = G.n_edges, hdr[2] = 4 * nv, hdr[3] = std::min<int>(max_iterations, kMaxTrace - 1);
    for (int e = 0; e < G.n_edges; e++) {
      pg->ei.h[(size_t)g * EC + e] = G.edge_i[e], pg->ej.h[(size_t)g * EC + e] = G.edge_j[e], pg->ek.h[(size_t)g * EC + e] = G.edge_kind[e];
      memcpy(pg->meas.h + ((size_t)g * EC + e) * 6, G.edge_meas + 6 * e, 48);
    }
  }
  hipStream_t st = pg->stream;
#define UP(b, cnt) HIP_OK(hipMemcpyAsync((b).d, (b).h, (size_t)(cnt) * sizeof(*(b).h), hipMemcpyHostToDevice, st))
  UP(pg->hdr, (size_t)n * kPgHdr);
  UP(pg->col, (size_t)n * NC);
  UP(pg->node0, (size_t)n * NC * 4);
  UP(pg->ei, (size_t)n * EC);
  UP(pg->ej, (size_t)n * EC);
  UP(pg->ek, (size_t)n * EC);
  UP(pg->meas, (size_t)n * EC * 6);
#undef UP
  HIP_OK(hipMemsetAsync(pg->stats_d.d, 0, (size_t)n * kStatsDoubles * sizeof(double), st));
  HIP_OK(hipMemsetAsync(pg->stats_i.d, 0, (size_t)n * kStatsInts * sizeof(int), st));
  PgPtrs P;
  P.ld = pg->ld, P.node_cap = NC, P.edge_cap = EC;
  P.hdr = pg->hdr.d, P.col = pg->col.d, P.node0 = pg->node0.d, P.edge_i = pg->ei.d, P.edge_j = pg->ej.d, P.edge_kind = pg->ek.d;
  P.meas = pg->meas.d, P.H = pg->H, P.A = pg->A, P.xout = pg->xout.d, P.stats_d = pg->stats_d.d, P.stats_i = pg->stats_i.d;
  hipLaunchKernelGGL(posegraph_kernel, dim3(n), dim3(kPgThreads), pg->lds_bytes, st, P);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(pg->xout.h, pg->xout.d, (size_t)n * pg->ld * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(pg->stats_d.h, pg->stats_d.d, (size_t)n * kStatsDoubles * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(pg->stats_i.h, pg->stats_i.d, (size_t)n * kStatsInts * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  for (int g = 0; g < n; g++) {
    VioPoseGraph &G = graphs[g];
    const int *col = pg->col.h + (size_t)g * NC;
    const double *x = pg->xout.h + (size_t)g * pg->ld;
    for (int k = 0; k < G.n_nodes; k++)
      if (col[k] >= 0) G.ypr[3 * k] = x[col[k]], G.t[3 * k] = x[col[k] + 1], G.t[3 * k + 1] = x[col[k] + 2], G.t[3 * k + 2] = x[col[k] + 3];
    if (stats) {
      VioSolveStats &s = stats[g];
      memset(&s, 0, sizeof(s));
      const double *sd = pg->stats_d.h + (size_t)g * kStatsDoubles;
      const int *si = pg->stats_i.h + (size_t)g * kStatsInts;
      s.initial_cost = sd[0], s.final_cost = sd[1];
      s.iterations = si[0], s.termination = si[1], s.num_successful_steps = si[2], s.num_unsuccessful_steps = si[3];
      for (int i = 0; i < s.iterations && i < VIO_MAX_TRACE && i < kMaxTrace; i++) {
        s.it_cost[i] = sd[4 + i], s.it_radius[i] = sd[4 + kMaxTrace + i], s.it_step_norm[i] = sd[4 + 2 * kMaxTrace + i];
        s.it_relative_decrease[i] = sd[4 + 3 * kMaxTrace + i], s.it_gradient_max_norm[i] = sd[4 + 4 * kMaxTrace + i];
        s.it_flags[i] = si[4 + i];
      }
    }
  }
  return VIO_OK;
}

}  // extern "C"
