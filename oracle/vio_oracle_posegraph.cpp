// oracle/vio_oracle_posegraph.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Plain C++ restatement of the 4-DoF loop pose graph, KeyFrameDatabase::optimize4DoFLoopPoseGraph
// (VINS_ios/loop/keyfame_database.cpp:140-353):
//   oracle_posegraph_build     resampling flags and edge list            :166-285
//   oracle_posegraph_optimize  the ceres::Solve of :287                  functors keyfame_database.h:62-104,271-366
//   oracle_posegraph_apply     poses after the solve, drift of cur_kf    :303-339
// The solve restates Ceres 1.12's TrustRegionMinimizer (CSI = VINS_ThirdPartyLib/ceres-solver/internal/ceres;
// CSI/trust_region_minimizer.cc) with LevenbergMarquardtStrategy (CSI/levenberg_marquardt_strategy.cc:66-163), Jacobi
// scaling, HuberLoss through the Corrector's rho'' <= 0 branch (CSI/loss_function.cc:47-61, CSI/corrector.cc:48-113) and
// an exact linear solve (DENSE_SCHUR in the reference is one: CSI/schur_complement_solver.cc). Jacobians are analytic
// (the reference uses AutoDiffCostFunction: same values up to rounding).
// PINNED against oracle/_ref (ref_posegraph_harness.cpp: the reference's own functors + vendored Ceres) through
// tests/golden/posegraph.npz (tests/golden/make_posegraph_golden.py) and directly in tests/test_posegraph.py.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <vector>

#include "vio_amd.h"
#include "vio_oracle.h"

namespace {

typedef std::vector<double> Vec;
const double kPi = 3.14159265358979323846;

// NormalizeAngle (keyfame_database.h:62-72), degrees
double normalize_angle(double a) {
  const double two_pi = 2.0 * 180;
  if (a > 0) return a - two_pi * floor((a + 180.0) / two_pi);
  return a + two_pi * floor((-a + 180.0) / two_pi);
}
// YawPitchRollToRotationMatrix (keyfame_database.h:228-246)
void ypr_to_R(double yaw, double pitch, double roll, double R[9]) {
  const double y = yaw / 180.0 * kPi, p = pitch / 180.0 * kPi, r = roll / 180.0 * kPi;
  R[0] = cos(y) * cos(p);
  R[1] = -sin(y) * cos(r) + cos(y) * sin(p) * sin(r);
  R[2] = sin(y) * sin(r) + cos(y) * sin(p) * cos(r);
  R[3] = sin(y) * cos(p);
  R[4] = cos(y) * cos(r) + sin(y) * sin(p) * sin(r);
  R[5] = -cos(y) * sin(r) + sin(y) * sin(p) * cos(r);
  R[6] = -sin(p);
  R[7] = cos(p) * sin(r);
  R[8] = cos(p) * cos(r);
}
// Utility::R2ypr (utility.hpp:76-91), degrees
void R_to_ypr(const double R[9], double ypr[3]) {
  const double n0 = R[0], n1 = R[3], n2 = R[6], o0 = R[1], o1 = R[4], a0 = R[2], a1 = R[5];
  const double y = atan2(n1, n0);
  const double p = atan2(-n2, n0 * cos(y) + n1 * sin(y));
  const double r = atan2(a0 * sin(y) - a1 * cos(y), -o0 * sin(y) + o1 * cos(y));
  ypr[0] = y / kPi * 180.0, ypr[1] = p / kPi * 180.0, ypr[2] = r / kPi * 180.0;
}
// Utility::ypr2R (utility.hpp:93-121): Rz Ry Rx
void ypr_to_R_utility(const double ypr[3], double R[9]) {
  const double y = ypr[0] / 180.0 * kPi, p = ypr[1] / 180.0 * kPi, r = ypr[2] / 180.0 * kPi;
  const double Rz[9] = {cos(y), -sin(y), 0, sin(y), cos(y), 0, 0, 0, 1};
  const double Ry[9] = {cos(p), 0., sin(p), 0., 1., 0., -sin(p), 0., cos(p)};
  const double Rx[9] = {1., 0., 0., 0., cos(r), -sin(r), 0., sin(r), cos(r)};
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += Rz[i * 3 + k] * Ry[k * 3 + j];
      T[i * 3 + j] = s;
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += T[i * 3 + k] * Rx[k * 3 + j];
      R[i * 3 + j] = s;
    }
}

struct Graph {
  const VioPoseGraph *g;
  int na;                 // variable nodes
  std::vector<int> col;   // node -> first column of its 4 unknowns [yaw, t] or -1 (constant / without edges)
  std::vector<int> node;  // variable index -> node
  int N;
};

struct Lin {
  Vec H;   // N x N, lower triangle used
  Vec g;   // J^T r
  std::vector<int> first;  // first non-zero column of every row (envelope)
};

// cost = sum rho(|r|^2) / 2; with lin: H = J^T J and g = J^T r of the loss-corrected residual blocks
double evaluate(const Graph &G, const Vec &x, Lin *lin) {
  const VioPoseGraph &g = *G.g;
  const int N = G.N;
  if (lin) {
    lin->H.assign((size_t)N * N, 0.0), lin->g.assign(N, 0.0);
    lin->first.resize(N);
    for (int r = 0; r < N; r++) lin->first[r] = r - r % 4;
  }
  auto yaw_of = [&](int k) { return G.col[k] >= 0 ? x[G.col[k]] : g.ypr[3 * k]; };
  auto t_of = [&](int k, int c) { return G.col[k] >= 0 ? x[G.col[k] + 1 + c] : g.t[3 * k + c]; };
  double cost = 0;
  for (int e = 0; e < g.n_edges; e++) {
    const int i = g.edge_i[e], j = g.edge_j[e];
    if (G.col[i] < 0 && G.col[j] < 0) continue;  // depends on constants only: removed from the program
    const double *m = g.edge_meas + 6 * e;
    const double yi = yaw_of(i), yj = yaw_of(j);
    double R[9];
    ypr_to_R(yi, m[4], m[5], R);
    const double d[3] = {t_of(j, 0) - t_of(i, 0), t_of(j, 1) - t_of(i, 1), t_of(j, 2) - t_of(i, 2)};
    double r[4];
    for (int k = 0; k < 3; k++) r[k] = R[0 + k] * d[0] + R[3 + k] * d[1] + R[6 + k] * d[2] - m[k];  // R^T d
    r[3] = normalize_angle(yj - yi - m[3]);
    // Jacobian rows over [yaw_i, t_i(3), yaw_j, t_j(3)]
    double J[4][8];
    memset(J, 0, sizeof(J));
    for (int k = 0; k < 3; k++) {
      J[k][0] = (-R[3 + k] * d[0] + R[0 + k] * d[1]) * (kPi / 180.0);  // d(R^T d)/dyaw_i, yaw in degrees
      for (int c = 0; c < 3; c++) J[k][1 + c] = -R[3 * c + k], J[k][5 + c] = R[3 * c + k];
    }
    J[3][0] = -1.0, J[3][4] = 1.0;
    if (g.edge_kind[e] == 1) {  // FourDOFWeightError: weight 10 on the translation rows, weight / 10 on the yaw row
      for (int k = 0; k < 3; k++) {
        r[k] *= 10.0;
        for (int c = 0; c < 8; c++) J[k][c] *= 10.0;
      }
      cost += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    } else {  // HuberLoss(1.0)
      const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
      if (s > 1.0) {
        const double rt = sqrt(s);
        cost += 0.5 * (2.0 * rt - 1.0);
        const double rho1 = std::max(std::numeric_limits<double>::min(), 1.0 / rt);
        const double sc = sqrt(rho1);
        for (int k = 0; k < 4; k++) {
          r[k] *= sc;
          for (int c = 0; c < 8; c++) J[k][c] *= sc;
        }
      } else {
        cost += 0.5 * s;
      }
    }
    if (!lin) continue;
    const int cols[2] = {G.col[i], G.col[j]};
    for (int a = 0; a < 8; a++) {
      const int ca = cols[a / 4];
      if (ca < 0) continue;
      const int ra = ca + a % 4;
      double gs = 0;
      for (int k = 0; k < 4; k++) gs += J[k][a] * r[k];
      lin->g[ra] += gs;
      for (int b = 0; b < 8; b++) {
        const int cb = cols[b / 4];
        if (cb < 0) continue;
        const int rb = cb + b % 4;
        if (rb > ra) continue;
        double s = 0;
        for (int k = 0; k < 4; k++) s += J[k][a] * J[k][b];
        lin->H[(size_t)ra * N + rb] += s;
        lin->first[ra] = std::min(lin->first[ra], cb);
      }
    }
  }
  return cost;
}

// (H + diag(D2)) y = b by an envelope Cholesky (rows only reach back to their first non-zero column). false: not PD.
bool solve_envelope(int N, const Vec &Hs, const Vec &D2, const std::vector<int> &first, const Vec &b, Vec &y) {
  Vec L((size_t)N * N, 0.0);
  for (int r = 0; r < N; r++) {
    for (int c = first[r]; c <= r; c++) {
      double s = Hs[(size_t)r * N + c] + (c == r ? D2[r] : 0.0);
      for (int k = std::max(first[r], first[c]); k < c; k++) s -= L[(size_t)r * N + k] * L[(size_t)c * N + k];
      if (c == r) {
        if (!(s > 0.0)) return false;
        L[(size_t)r * N + r] = sqrt(s);
      } else {
        L[(size_t)r * N + c] = s / L[(size_t)c * N + c];
      }
    }
  }
  y = b;
  for (int r = 0; r < N; r++) {
    double s = y[r];
    for (int k = first[r]; k < r; k++) s -= L[(size_t)r * N + k] * y[k];
    y[r] = s / L[(size_t)r * N + r];
  }
  for (int r = N - 1; r >= 0; r--) {
    y[r] /= L[(size_t)r * N + r];
    for (int k = first[r]; k < r; k++) y[k] -= L[(size_t)r * N + k] * y[r];
  }
  return true;
}

void plus(const Graph &G, const Vec &x, const Vec &delta, Vec &out) {
  out = x;
  for (int v = 0; v < G.na; v++) {
    out[4 * v] = normalize_angle(x[4 * v] + delta[4 * v]);  // AngleLocalParameterization (keyfame_database.h:74-90)
    for (int c = 1; c < 4; c++) out[4 * v + c] = x[4 * v + c] + delta[4 * v + c];
  }
}

}  // namespace

extern "C" int oracle_posegraph_optimize(VioPoseGraph *g, int max_iterations, VioSolveStats *st) {
  if (!g || g->n_nodes < 1 || !g->t || !g->ypr) return VIO_EINVAL;
  Graph G;
  G.g = g;
  const int n = g->n_nodes;
  std::vector<char> used(n, 0);
  for (int e = 0; e < g->n_edges; e++) {
    if (g->edge_i[e] < 0 || g->edge_i[e] >= n || g->edge_j[e] < 0 || g->edge_j[e] >= n) return VIO_EINVAL;
    used[g->edge_i[e]] = used[g->edge_j[e]] = 1;
  }
  G.col.assign(n, -1);
  for (int k = 0; k < n; k++)
    if (used[k] && k != g->fixed_node) G.col[k] = 4 * (int)G.node.size(), G.node.push_back(k);
  G.na = (int)G.node.size(), G.N = 4 * G.na;
  const int N = G.N;
  if (st) memset(st, 0, sizeof(*st));
  Vec x(N);
  for (int v = 0; v < G.na; v++) {
    x[4 * v] = g->ypr[3 * G.node[v]];
    for (int c = 0; c < 3; c++) x[4 * v + 1 + c] = g->t[3 * G.node[v] + c];
  }
  Lin L;
  double x_cost = evaluate(G, x, &L);
  double x_norm = -1.0;
  Vec scale(N);
  for (int c = 0; c < N; c++) scale[c] = 1.0 / (1.0 + sqrt(L.H[(size_t)c * N + c]));  // trust_region_minimizer.cc:239-254
  auto grad_max_norm = [&](const Vec &xs, const Lin &LL) {
    Vec ng(N), tmp;
    for (int c = 0; c < N; c++) ng[c] = -LL.g[c];
    plus(G, xs, ng, tmp);
    double m = 0;
    for (int c = 0; c < N; c++) m = std::max(m, fabs(xs[c] - tmp[c]));
    return m;
  };
  double radius = 1e4, decrease_factor = 2.0;  // levenberg_marquardt_strategy.cc:48-56
  bool reuse_diagonal = false;
  Vec diagonal(N), D2(N), Hs, gs(N), y, step(N), delta(N), cand;
  int it = 0, n_ok = 0, n_bad = 0, invalid_run = 0, termination = 0, recorded = 0;
  double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost, ev_acc_ref = 0, ev_acc_cand = 0;
  double min_recorded_cost = std::numeric_limits<double>::max();
  auto record = [&](int i, double cost, double step_norm, double rel, double gmax, bool valid, bool ok) {
    recorded = i + 1;
    min_recorded_cost = std::min(min_recorded_cost, cost);
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
  auto scaled_system = [&]() {
    Hs.assign((size_t)N * N, 0.0);
    for (int r = 0; r < N; r++) {
      for (int c = L.first[r]; c <= r; c++) Hs[(size_t)r * N + c] = scale[r] * L.H[(size_t)r * N + c] * scale[c];
      gs[r] = scale[r] * L.g[r];
    }
  };
  scaled_system();
  while (N > 0) {
    if (it >= max_iterations) break;
    if (last_ok && gmax <= 1e-10) { termination = 1; break; }
    if (radius <= 1e-32) { termination = 1; break; }
    it++;
    // LevenbergMarquardtStrategy::ComputeStep
    if (!reuse_diagonal)
      for (int c = 0; c < N; c++) diagonal[c] = std::min(std::max(Hs[(size_t)c * N + c], 1e-6), 1e32);
    for (int c = 0; c < N; c++) {
      const double lm = sqrt(diagonal[c] / radius);
      D2[c] = lm * lm;
    }
    bool solver_ok = solve_envelope(N, Hs, D2, L.first, gs, y);
    if (solver_ok)
      for (int c = 0; c < N; c++)
        if (!std::isfinite(y[c])) solver_ok = false;
    reuse_diagonal = true;
    bool step_valid = false;
    double model_cost_change = 0;
    if (solver_ok) {
      for (int c = 0; c < N; c++) step[c] = -y[c];
      // model_cost_change = -(J step)^T (r + J step / 2)   (trust_region_minimizer.cc:402-416)
      Vec hv(N, 0.0);
      for (int r = 0; r < N; r++)
        for (int c = L.first[r]; c <= r; c++) {
          const double h = Hs[(size_t)r * N + c];
          hv[r] += h * step[c];
          if (c != r) hv[c] += h * step[r];
        }
      double sg = 0, shs = 0;
      for (int r = 0; r < N; r++) sg += step[r] * gs[r], shs += step[r] * hv[r];
      model_cost_change = -sg - 0.5 * shs;
      step_valid = model_cost_change > 0.0;
    }
    if (!step_valid) {
      if (++invalid_run >= 5) { termination = 2; break; }
      radius = radius / decrease_factor, decrease_factor *= 2.0, reuse_diagonal = true;  // StepIsInvalid -> StepRejected(0)
      last_ok = false;
      n_bad++;
      record(it, x_cost, 0, 0, gmax, false, false);
      continue;
    }
    invalid_run = 0;
    for (int c = 0; c < N; c++) delta[c] = step[c] * scale[c];
    plus(G, x, delta, cand);
    double cand_cost = evaluate(G, cand, NULL);
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    double sn = 0;
    for (int c = 0; c < N; c++) sn += (x[c] - cand[c]) * (x[c] - cand[c]);
    const double step_norm = sqrt(sn);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { termination = 1; break; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= 1e-6 * x_cost) { termination = 1; break; }
    const double rel = (ev_cur - cand_cost) / model_cost_change;
    const double hist = (ev_ref - cand_cost) / (ev_acc_ref + model_cost_change);
    const double rho = std::max(rel, hist);
    if (rho > 1e-3) {
      x = cand;
      double xn = 0;
      for (int c = 0; c < N; c++) xn += x[c] * x[c];
      x_norm = sqrt(xn);
      x_cost = evaluate(G, x, &L);
      scaled_system();
      gmax = grad_max_norm(x, L);
      // LevenbergMarquardtStrategy::StepAccepted (:146-153)
      radius = radius / std::max(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3));
      radius = std::min(1e16, radius);
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
  for (int v = 0; v < G.na; v++) {
    g->ypr[3 * G.node[v]] = x[4 * v];
    for (int c = 0; c < 3; c++) g->t[3 * G.node[v] + c] = x[4 * v + 1 + c];
  }
  if (st) {
    st->final_cost = min_recorded_cost;
    st->iterations = recorded;
    st->termination = termination;
    st->num_successful_steps = n_ok;
    st->num_unsuccessful_steps = n_bad;
  }
  return VIO_OK;
}

// keyfame_database.cpp:166-285. kf[0] is the earliest_loop_index keyframe, kf[n_kf-1] the current one.
extern "C" int oracle_posegraph_build(const VioPoseGraphKeyframe *kf, int n_kf, double total_length, int max_frame_num,
                                      int list_size, double *t, double *ypr, unsigned char *skip, int cap_edges, int *edge_i,
                                      int *edge_j, unsigned char *edge_kind, double *edge_meas, int *n_edges) {
  if (!kf || n_kf < 1 || !t || !ypr || !skip || !n_edges) return VIO_EINVAL;
  const double min_dis = total_length / (1.0 * max_frame_num);
  double last_P[3] = {0, 0, 0}, dis = 0;
  for (int k = 0; k < n_kf; k++) {  // :176-198
    const double d0 = kf[k].t[0] - last_P[0], d1 = kf[k].t[1] - last_P[1], d2 = kf[k].t[2] - last_P[2];
    dis += sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    if (k == 0 || dis > min_dis || kf[k].has_loop || kf[k].is_looped || list_size < max_frame_num) dis = 0, skip[k] = 0;
    else skip[k] = 1;
    memcpy(last_P, kf[k].t, 24);
  }
  int ne = 0;
  for (int i = 0; i < n_kf; i++) {
    memcpy(t + 3 * i, kf[i].origin_t, 24);
    // (tmp_q = tmp_r; R2ypr(tmp_q.toRotationMatrix()): through a quaternion and back, a rotation matrix is unchanged
    // up to rounding; taken as is)
    R_to_ypr(kf[i].origin_r, ypr + 3 * i);
    if (skip[i]) continue;
    int j = 1, cnt = 0;
    while (cnt < 5) {  // :232-262
      if (i - j < 0) break;
      if (skip[i - j]) {
        j++;
        continue;
      }
      cnt++;
      const int c = i - j;
      if (ne >= cap_edges) return VIO_ECAP;
      const double d[3] = {t[3 * i] - t[3 * c], t[3 * i + 1] - t[3 * c + 1], t[3 * i + 2] - t[3 * c + 2]};
      double *m = edge_meas + 6 * ne;
      const double *R = kf[c].origin_r;
      for (int k = 0; k < 3; k++) m[k] = R[0 + k] * d[0] + R[3 + k] * d[1] + R[6 + k] * d[2];  // q^-1 * relative_t
      m[3] = ypr[3 * i] - ypr[3 * c], m[4] = ypr[3 * c + 1], m[5] = ypr[3 * c + 2];
      edge_i[ne] = c, edge_j[ne] = i, edge_kind[ne] = 0, ne++;
      j++;
    }
    if (kf[i].has_loop) {  // :264-285
      int c = -1;
      for (int k = 0; k < n_kf; k++)
        if (kf[k].global_index == kf[i].loop_index) c = k;
      if (c < 0) return VIO_EINVAL;  // (loop_index < earliest_loop_index: the reference asserts)
      if (ne >= cap_edges) return VIO_ECAP;
      double *m = edge_meas + 6 * ne;
      double yc[3];
      R_to_ypr(kf[c].origin_r, yc);
      m[0] = kf[i].loop_info[0], m[1] = kf[i].loop_info[1], m[2] = kf[i].loop_info[2], m[3] = kf[i].loop_info[7];
      m[4] = yc[1], m[5] = yc[2];
      edge_i[ne] = c, edge_j[ne] = i, edge_kind[ne] = 1, ne++;
    }
  }
  *n_edges = ne;
  return VIO_OK;
}

// keyfame_database.cpp:303-339
extern "C" int oracle_posegraph_apply(const VioPoseGraphKeyframe *kf, int n_kf, const double *t, const double *ypr,
                                      const unsigned char *skip, double *out_t, double *out_r, double *yaw_drift,
                                      double *r_drift, double *t_drift) {
  if (!kf || n_kf < 1 || !t || !ypr || !skip || !out_t || !out_r) return VIO_EINVAL;
  double td[3] = {0, 0, 0}, rd[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < n_kf; i++) {
    double R[9];
    ypr_to_R_utility(ypr + 3 * i, R);
    const double *tt = t + 3 * i;
    if (skip[i]) {  // r_drift_it * tmp_t + t_drift_it, r_drift_it * tmp_r
      for (int a = 0; a < 3; a++) {
        out_t[3 * i + a] = rd[3 * a] * tt[0] + rd[3 * a + 1] * tt[1] + rd[3 * a + 2] * tt[2] + td[a];
        for (int b = 0; b < 3; b++) out_r[9 * i + 3 * a + b] = rd[3 * a] * R[b] + rd[3 * a + 1] * R[3 + b] + rd[3 * a + 2] * R[6 + b];
      }
    } else {
      const double *Ro = kf[i].origin_r, *to = kf[i].origin_t;
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) rd[3 * a + b] = R[3 * a] * Ro[3 * b] + R[3 * a + 1] * Ro[3 * b + 1] + R[3 * a + 2] * Ro[3 * b + 2];
      for (int a = 0; a < 3; a++) td[a] = tt[a] - (rd[3 * a] * to[0] + rd[3 * a + 1] * to[1] + rd[3 * a + 2] * to[2]);
      memcpy(out_t + 3 * i, tt, 24), memcpy(out_r + 9 * i, R, 72);
    }
  }
  // drift of the current keyframe (:333-339)
  const int c = n_kf - 1;
  double a1[3], a2[3];
  R_to_ypr(out_r + 9 * c, a1), R_to_ypr(kf[c].origin_r, a2);
  const double yd = a1[0] - a2[0];
  const double e[3] = {yd, 0, 0};
  double Rd[9];
  ypr_to_R_utility(e, Rd);
  if (yaw_drift) *yaw_drift = yd;
  if (r_drift) memcpy(r_drift, Rd, 72);
  if (t_drift)
    for (int a = 0; a < 3; a++)
      t_drift[a] = out_t[3 * c + a] - (Rd[3 * a] * kf[c].origin_t[0] + Rd[3 * a + 1] * kf[c].origin_t[1] + Rd[3 * a + 2] * kf[c].origin_t[2]);
  return VIO_OK;
}
