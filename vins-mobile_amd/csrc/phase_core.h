// phase_core.h — the sliding-window solve (VINS::solve_ceres, VINS_ios/VINS.cpp:480-831) as a SEQUENCE OF LAUNCHES.
//
// solver_core.h runs the whole of solve_ceres in one workgroup per window: the factor-parallel phases (projection / IMU /
// prior evaluation, the Gram products) then execute at the occupancy, register budget and LDS footprint that the serial
// phases (band + pose factorization, dogleg) dictate. Here the two kinds of work are separate kernels:
//
//   setup      once per solve: slot records, cov^-1 of the IMU factors, the prior's H0 / b0 in the reduced system's layout
//   linearize  evaluation of every factor at ONE point (cost + Jacobians): a workgroup whose waves never meet at a barrier
//              while they walk the projection factors -- each wave evaluates 64 factors, stages their robustified Jacobian
//              rows in its OWN strip of LDS and forms the (host, target) Gram products of that strip on the matrix cores --,
//              IMU raw evaluation and the prior's H0 dx ride along. Output: a linearization buffer in global memory.
//   step       TrustRegionMinimizer's loop body between two evaluations (CSI/trust_region_minimizer.cc:66-786): step
//              acceptance for the candidate just evaluated, reduced system, band + pose factorization, dogleg, next
//              candidate. The loop-carried state lives in a PhaseRec in global memory; the kernel is re-entered once per
//              evaluation.
//   finish     new2old, outputs, marginalization (marg_core.h)
//
// The linearization at a candidate is computed SPECULATIVELY, cost and Jacobians together, before the step is accepted:
// Ceres evaluates the cost at the candidate and, after acceptance, residuals + Jacobians at the same point again
// (trust_region_minimizer.cc:428-640); an accepted step (the common case) therefore costs one evaluation here instead of
// two, a rejected one wastes the Jacobian half of one. Two linearization buffers alternate: `cur` belongs to the accepted
// iterate, the other receives the candidate's. Results are those of solver_core.h (same factor code, same linear algebra).
#pragma once

#include "batch.h"

namespace vio {

#ifndef VIO_EMUL

enum { PH_FIRST = 0, PH_CAND = 1, PH_DONE = 2 };

// Loop-carried scalars of minimize() (solver_core.h) between two launches of the step kernel.
struct PhaseRec {
  double x_cost, x_norm, gmax, radius, mu, mu_used, dogleg_step_norm, alpha, gd_sq, qf_cauchy;
  double ev_min, ev_cur, ev_ref, ev_cand, ev_acc_ref, ev_acc_cand, min_rec, model_cost_change;
  int phase, cur, it, n_ok, n_bad, invalid_run, termination, recorded, reuse, last_ok;
};
static_assert(sizeof(PhaseRec) <= kRecDoubles * sizeof(double), "PhaseRec outgrew its slot");

// One base pointer and offsets: every buffer address is base + offset arithmetic, so that the compiler keeps the accesses in
// the global address space (a pointer picked from a two-element pointer array by the run-time buffer index decays to a
// generic pointer, and flat loads / stores / atomics count against the LDS wait counter as well: every later ds_read
// then waits for them).
struct PhaseView {
  double *base;
  int o_rec, o_prcol, o_fh, o_x[2], o_h[2], o_sp, o_sf, o_gpf, o_dp, o_gnp, o_gnf;
  int x_sb, x_feat, h_gp, h_gf, h_hff, h_App, h_imuJ, h_imur, h_WTf;
  VIO_HD PhaseRec *rec_ptr() const { return reinterpret_cast<PhaseRec *>(base + o_rec); }
  VIO_HD int *prcol() const { return reinterpret_cast<int *>(base + o_prcol); }
  VIO_HD int *fh() const { return reinterpret_cast<int *>(base + o_fh); }
  VIO_HD double *xb(int i) const { return base + (i ? o_x[1] : o_x[0]); }
  VIO_HD double *hb(int i) const { return base + (i ? o_h[1] : o_h[0]); }
  VIO_HD double *sp() const { return base + o_sp; }
  VIO_HD double *sf() const { return base + o_sf; }
  VIO_HD double *gpf() const { return base + o_gpf; }
  VIO_HD double *dp() const { return base + o_dp; }
  VIO_HD double *gnp() const { return base + o_gnp; }
  VIO_HD double *gnf() const { return base + o_gnf; }
};

VIO_HD PhaseView make_phase_view(const BatchPtrs &B, int b) {
  const PhaseLayout &L = B.PL;
  PhaseView p;
  p.base = B.phase + (size_t)b * L.total;
  p.o_rec = (int)L.rec, p.o_prcol = (int)L.prcol, p.o_fh = (int)L.fh, p.o_x[0] = (int)L.x[0], p.o_x[1] = (int)L.x[1], p.o_h[0] = (int)L.h[0], p.o_h[1] = (int)L.h[1];
  p.o_sp = (int)L.sp, p.o_sf = (int)L.sf, p.o_gpf = (int)L.gpf, p.o_dp = (int)L.dp, p.o_gnp = (int)L.gnp, p.o_gnf = (int)L.gnf;
  p.x_sb = (int)L.x_sb, p.x_feat = (int)L.x_feat;
  p.h_gp = (int)L.h_gp, p.h_gf = (int)L.h_gf, p.h_hff = (int)L.h_hff, p.h_App = (int)L.h_App, p.h_imuJ = (int)L.h_imuJ;
  p.h_imur = (int)L.h_imur, p.h_WTf = (int)L.h_WTf;
  return p;
}

// =====================================================================================================
// setup
// =====================================================================================================
// LDS of the setup kernel: the prior's column map and one staging area (J0 of the prior / the mailboxes of the IMU
// covariance inversion).
struct SetupWork {
  ldsd App;     // staging area (named like the member of WorkT that setup_imu_info / setup_prior use)
  int nstage;
  ldsd stage;   // (= App: the name setup_prior uses)
  ldsi prcol;
};
VIO_HD size_t carve_setup(const BatchDims &d, ldsd base, SetupWork *w) {
  // (a prior too large for the staging area is read from global memory by setup_prior: 6144 doubles hold the 75 x 75 prior
  // of a W = 10 window and leave room for three workgroups per CU)
  const size_t want = ((size_t)d.Ncap * d.Ncap + 1) & ~(size_t)1, nst = want < 6144 ? (want < 1024 ? 1024 : want) : 6144, npr = (((size_t)d.Ncap + 1) / 2 + 1 + 1) & ~(size_t)1;
  if (w) w->App = base, w->stage = base, w->nstage = (int)nst, w->prcol = reinterpret_cast<ldsi>(base + nst);
  return (nst + npr) * sizeof(double);
}

template <class SW>
VIO_DEV void phase_setup(const Ctx &cx, WinView &v, const PhaseView &pv, SW &w) {
  const int P = v.P, F = v.F;
  double *X = pv.xb(0);
  VIO_PARFOR(q, P * 7) X[q] = v.pose0[q];
  if (v.has_loop) VIO_PARFOR(q, 7) X[7 * P + q] = v.pose0[7 * v.loop_frame + q];  // VINS.cpp:590-591
  VIO_PARFOR(q, P * 9) X[pv.x_sb + q] = v.sb0[q];
  VIO_PARFOR(q, F) X[pv.x_feat + q] = v.feat0[q];
  VIO_PARFOR(q, v.nslots) v.sfact[q] = -1, v.srec_i[q] = -1;
  VIO_SYNC();
  VIO_PARFOR(k, v.M) {
    const int sl = v.fslot[k];
    v.sfact[sl] = k;
    v.srec_i[sl] = v.fhost[k] | (v.ftarget[k] << 8) | (v.ffeat[k] << 16);
    double *d = v.srec_d + 6 * (size_t)sl;
    for (int c = 0; c < 3; c++) d[c] = v.pts_i[3 * k + c], d[3 + c] = v.pts_j[3 * k + c];
  }
  // entries of W and of the raw IMU Jacobians that no evaluation writes stay zero: zeroed once, in both buffers
  for (int bf = 0; bf < 2; bf++) {
    double *H = pv.hb(bf);
    VIO_PARFOR(q, F * v.n6cap) H[pv.h_WTf + q] = 0.0;
    VIO_PARFOR(q, v.W * 450) H[pv.h_imuJ + q] = 0.0;
  }
  v.imu_J = pv.hb(0) + pv.h_imuJ;  // (setup_imu_info zeroes v.imu_J: done above for both buffers, harmless to repeat)
  setup_imu_info(cx, v, w.App);
  setup_prior(cx, v, w);
  VIO_PARFOR(a, v.prior_n) pv.prcol()[a] = w.prcol[a];
  VIO_PARFOR(f, F) pv.fh()[f] = v.fstart[f + 1] > v.fstart[f] ? v.fhost[v.fstart[f]] : -1;
  if (cx.tid == 0) {
    PhaseRec r;
    r.x_cost = 0, r.x_norm = -1.0, r.gmax = 0, r.radius = 1e4, r.mu = 1e-8, r.mu_used = 1e-8, r.dogleg_step_norm = 0, r.alpha = 0;
    r.gd_sq = 0, r.qf_cauchy = 0, r.ev_min = r.ev_cur = r.ev_ref = r.ev_cand = 0, r.ev_acc_ref = r.ev_acc_cand = 0, r.min_rec = 0;
    r.model_cost_change = 0;
    r.phase = PH_FIRST, r.cur = 1, r.it = 0, r.n_ok = 1, r.n_bad = 0, r.invalid_run = 0, r.termination = 0, r.recorded = 1;
    r.reuse = 0, r.last_ok = 1;
    *pv.rec_ptr() = r;
  }
}

// =====================================================================================================
// linearize
// =====================================================================================================
constexpr int kLinStage = 64;  // factors a wave stages at once (its strip of LDS: kLinStage * kGSlot doubles)
constexpr int kLinWaves = 3;   // projection waves of the linearize workgroup; one more wave takes the IMU factors and the prior
                               // (the raw IMU evaluation -- one lane per factor, ~1500 dependent f64 operations -- takes as long as five strips)
constexpr int kLinThreads = 64 * (kLinWaves + 1);

struct LinWork {
  ldsd pose, ex, rot, feat;  // the evaluation point: 7 (P + 1), 7, 9 (P + 2), F
  ldsd ppd, gp;              // diagonal pose blocks of the projection Gram products, gradient (projections + prior)
  ldsd hff, gf;              // per landmark
  ldsd wh;                   // [6][F]: host-frame coupling of every landmark
  ldsd prdx, prr;            // prior
  ldsd red;
  ldsd lprof;                // stage clock (ST_COUNT long longs)
  ldsd stage;                // [kLinWaves][kLinStage * kGSlot]
  int Fld;
};
VIO_HD size_t carve_lin(const BatchDims &d, ldsd base, LinWork *w) {
  size_t o = 0;
  auto take = [&](size_t n) {
    ldsd p = base + o;
    o += (n + 1) & ~(size_t)1;
    return p;
  };
  LinWork t;
  const size_t F = d.Flds, npc = (size_t)d.nblk_cap * kBS;
  t.Fld = (int)((F + 1) & ~(size_t)1);
  t.pose = take(7 * (size_t)(d.Pcap + 1)), t.ex = take(8), t.rot = take(9 * (size_t)(d.Pcap + 2)), t.feat = take(F);
  t.ppd = take(36 * (size_t)(d.Pcap + 1)), t.gp = take(npc);
  t.hff = take(F), t.gf = take(F), t.wh = take(6 * (size_t)t.Fld);
  t.prdx = take(d.Ncap), t.prr = take(d.Ncap);
  t.red = take(6 * (size_t)(kLinWaves + 1) + 2);
  t.lprof = take(ST_COUNT);
  t.stage = take((size_t)kLinWaves * kLinStage * kGSlot);
  if (w) *w = t;
  return o * sizeof(double);
}

// One evaluation of every factor of the window at the point xb[1 - cur]: cost and Jacobians -> hb[1 - cur]. The pose matrix
// leaves WITHOUT the prior's H0 (the step kernel adds it while it copies the matrix into LDS): it is zero-filled here and
// only ever accumulated into.
template <class LW>
VIO_DEV void phase_linearize(const Ctx &cx, const WinView &v, const PhaseView &pv, LW &w) {
  const int phase = pv.rec_ptr()->phase;
  if (phase == PH_DONE) return;
  const int tgt = 1 - pv.rec_ptr()->cur;
  const double *X = pv.xb(tgt);
  double *H = pv.hb(tgt);
  const int P = v.P, F = v.F, np = v.np, nF = v.P + v.has_loop, n = v.prior_n;
  const int napp = (int)tri_doubles(v.nrows);
  double *Happ = H + pv.h_App, *HW = H + pv.h_WTf;
  double cost = 0.0;
  const int tid_ = VIO_TID(cx), wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), lane = tid_ & 63;
  const int NT = (int)cx.nt;

  // ---- A: the evaluation point into LDS (every load of a lane issued before its first store), accumulators zeroed ------
  {
    const int npv = 7 * nF;
    const double a0 = X[tid_ < npv ? tid_ : 0], a1 = X[pv.x_feat + (tid_ < F ? tid_ : 0)], a2 = v.ex[tid_ < 7 ? tid_ : 0];
    double a3 = 0.0, a4 = 0.0;
    if (n > 0) a3 = v.prb0[tid_ < n ? tid_ : 0];
    if (tid_ + NT < F) a4 = X[pv.x_feat + tid_ + NT];
    if (n > 0 && tid_ < v.prior_nb) {  // dx of one prior block (marginalization_factor.cpp:349-367)
      const int kind = v.pr_kind[tid_], idx = v.pr_index[tid_], o = v.pr_offset[tid_];
      const double *x0 = v.pr_x0 + 9 * tid_;
      if (kind == 0) prior_block_dx(7, X + 7 * idx, x0, w.prdx + o);
      else if (kind == 1) prior_block_dx(9, X + pv.x_sb + 9 * idx, x0, w.prdx + o);
      else prior_block_dx(7, v.ex, x0, w.prdx + o);
    }
    if (tid_ >= NT - (nF + 1)) {  // rotation matrices of the poses under evaluation, then r_ic (straight from global memory)
      const int i = tid_ - (NT - (nF + 1));
      const bool is_ex = i == nF;
      const double *q = is_ex ? v.ex + 3 : X + 7 * i + 3;
      double R[9];
      qtoR(Quat{q[0], q[1], q[2], q[3]}, R);
      auto dst = w.rot + 9 * (is_ex ? v.P + 1 : i);
      for (int k = 0; k < 9; k++) dst[k] = R[k];
    }
    if (tid_ < npv) w.pose[tid_] = a0;
    if (tid_ < F) w.feat[tid_] = a1;
    if (tid_ < 7) w.ex[tid_] = a2;
    if (n > 0 && tid_ < n) w.prr[tid_] = a3;
    if (tid_ + NT < F) w.feat[tid_ + NT] = a4;
    for (int f = tid_ + 2 * NT; f < F; f += NT) w.feat[f] = X[pv.x_feat + f];
    for (int i = tid_ + NT; i < n; i += NT) w.prr[i] = v.prb0[i];
  }
  VIO_PARFOR(q, 36 * nF) w.ppd[q] = 0.0;
  VIO_PARFOR(q, np) w.gp[q] = 0.0;
  VIO_PARFOR(f, F) {
    w.hff[f] = 0.0, w.gf[f] = 0.0;
    for (int c = 0; c < 6; c++) w.wh[c * w.Fld + f] = 0.0;
  }
  VIO_PARFOR(q, napp) Happ[q] = 0.0;
  VIO_SYNC();  // (also orders the zero fill of the pose matrix ahead of the accumulation into it)
  stamp(cx, ST_SETUP_PRIOR);

  // ---- B: no workgroup barrier from here to the end of the factor walk ----------------------------------------------
  const int li = lane & 15, kq = lane >> 4;
  double *HJ = H + pv.h_imuJ, *Hr = H + pv.h_imur;
  if (wave == kLinWaves) {
    // The spare wave: IMU factors -- raw residual + Jacobian by one lane each (the whitened Gram products are formed where
    // the reduced system is assembled: phase_step), Mr = cov^-1 r for the cost -- and the prior's H0 dx.
    for (int f = lane; f < v.W; f += 64)
      imu_eval_raw(v.gravity, v.preint + f * kPreintDoubles, X + 7 * f, X + pv.x_sb + 9 * f, X + 7 * (f + 1), X + pv.x_sb + 9 * (f + 1),
                   Hr + f * 15, HJ + f * 450);
    stamp(cx, ST_IMU_RAW);
    if (n > 0) {
      Ctx sub = cx;
      sub.tid = lane, sub.nt = 64;
      dense_matvec_cols(sub, v.prH0, n, w.prdx, [&](int i, double sacc) { VIO_ATOMIC_ADD(w.prr + i, sacc); });  // prr = b0 + H0 dx = J^T r
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int q = lane; q < v.W * 15; q += 64) {
      const int f = q / 15, r = q - f * 15;
      const double *info = v.imu_info + f * 225 + r * 15, *rr = Hr + f * 15;
      double iv[15], rv[15], s = 0;
#pragma unroll
      for (int k = 0; k < 15; k++) iv[k] = info[k], rv[k] = rr[k];
#pragma unroll
      for (int k = 0; k < 15; k++) s += iv[k] * rv[k];
      cost += 0.5 * s * rv[r];
    }
    stamp(cx, ST_M_PRIOR);
  } else {
    const double bb = v.cauchy_b, cc = 1.0 / bb;
    // where this lane's four accumulator elements of a bucket's Gram matrix go (a property of the lane, not of the bucket)
    int f_kind[4], f_off[4];  // 0 nothing, 1 LDS base + host * mul, 2 LDS base + target * mul, 3 off-diagonal block (global)
    ldsd f_base[4];
    int f_mul[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
      const int row = kq + 4 * r4, col = li;
      f_kind[r4] = 0, f_off[r4] = 0, f_mul[r4] = 36, f_base[r4] = w.ppd;
      if (row < 6) {
        if (col <= row) f_kind[r4] = 1, f_base[r4] = w.ppd + row * 6 + col;
      } else if (row < 12) {
        if (col < 6) f_kind[r4] = 3, f_off[r4] = ((row - 6) << 8) | col;
        else if (col < 12 && col <= row) f_kind[r4] = 2, f_base[r4] = w.ppd + (row - 6) * 6 + (col - 6);
      } else if (row == 12) {
        if (col < 6) f_kind[r4] = 1, f_mul[r4] = kBS, f_base[r4] = w.gp + col;
        else if (col < 12) f_kind[r4] = 2, f_mul[r4] = kBS, f_base[r4] = w.gp + col - 6;
      }
    }
    auto flush = [&](v4d acc, int ht) {
      const int h = ht >> 16, t = ht & 0xffff;
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        if (f_kind[r4] == 1) VIO_ATOMIC_ADD(f_base[r4] + h * f_mul[r4], acc[r4]);
        else if (f_kind[r4] == 2) VIO_ATOMIC_ADD(f_base[r4] + t * f_mul[r4], acc[r4]);
        else if (f_kind[r4] == 3) {
          // J_t^T J_h, element (a, c): row 6 t + a, column 6 h + c of the lower triangle (transposed when the target precedes
          // its host). Several waves may hold parts of one bucket, reversed pairs share a block: always accumulated.
          const int a = f_off[r4] >> 8, c = f_off[r4] & 255;
          const int R = t > h ? 6 * t + a : 6 * h + c, Cc = t > h ? 6 * h + c : 6 * t + a;
          VIO_ATOMIC_ADD(Happ + tri_at(R, Cc), acc[r4]);
        }
      }
    };
    // this wave's strip of the slot order
    const int per = (((v.nslots + kLinWaves - 1) / kLinWaves) + kLinStage - 1) / kLinStage * kLinStage;
    const int ws0 = wave * per, ws1 = ws0 + per < v.nslots ? ws0 + per : v.nslots;
    ldsd G = w.stage + wave * (kLinStage * kGSlot);
    // bucket descriptors: a window of 64 of them in the lanes
    int tb = 0, m_s0 = 0, m_s1 = 0, m_ht = 0;
    auto load_table = [&](int base) {
      tb = base;
      const int pl = base + lane;
      const bool pv_ = pl < v.npairs;
      m_s0 = pv_ ? v.pair_s0[pl] : 0x7fffffff, m_s1 = pv_ ? v.pair_s1[pl] : 0x7fffffff;
      m_ht = pv_ ? (v.pair_h[pl] << 16) | v.pair_t[pl] : 0;
    };
    // the first strip's factor records are on their way while the bucket table is searched
    int rec_n = ws0 + lane < ws1 ? v.srec_i[ws0 + lane] : -1;
    double pn[6];
#pragma unroll
    for (int c = 0; c < 6; c++) pn[c] = v.srec_d[6 * (size_t)(ws0 + lane < ws1 ? ws0 + lane : 0) + c];
    int b = 0;  // first bucket that reaches into the strip
    for (int base = 0; base < v.npairs; base += 64) {
      load_table(base);
      const unsigned long long before = __builtin_amdgcn_ballot_w64(base + lane < v.npairs && m_s1 <= ws0);
      const int c = __builtin_popcountll(before);
      b += c;
      if (c < 64) break;
    }
    if (ws0 < ws1 && (b < tb || b >= tb + 64)) load_table(b & ~63);
    v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
    bool pending = false;
    int pend_ht = 0;
    const bool lv = li < 13;
    const int src = li < 6 ? li : li < 9 ? li - 6 : li < 12 ? li - 3 : 9;
    const double sg = (li >= 6 && li < 9) ? -1.0 : (lv ? 1.0 : 0.0);
    for (int p0 = ws0; p0 < ws1; p0 += kLinStage) {
      const int rec = rec_n;
      double pij[6];
#pragma unroll
      for (int c = 0; c < 6; c++) pij[c] = pn[c];
      {  // the next strip's records (a global round trip) travel while this strip is evaluated
        const int nx = p0 + kLinStage + lane;
        rec_n = nx < ws1 ? v.srec_i[nx] : -1;
#pragma unroll
        for (int c = 0; c < 6; c++) pn[c] = v.srec_d[6 * (size_t)(nx < ws1 ? nx : 0) + c];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the Gram products of the previous strip have read their operands
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (rec >= 0) {
        const int h = rec & 255, t = (rec >> 8) & 255, f = rec >> 16;
        double r[2], Ji[12], Jj[12], Jl[2];
        projection_eval_rot(v.s_info, w.rot + 9 * h, w.pose + 7 * h, w.rot + 9 * t, w.pose + 7 * t, w.rot + 9 * (v.P + 1), w.ex, w.feat[f],
                            pij, pij + 3, true, r, Ji, Jj, Jl);
        const double sq = r[0] * r[0] + r[1] * r[1];
        const double sum = 1.0 + sq * cc;
        cost += 0.5 * bb * log(sum);
        const double sr = rsqrt_f(sum);  // Corrector: rho'' < 0 => scale by sqrt(rho') = 1 / sqrt(1 + s / b)   (sum >= 1)
        auto g = G + lane * kGSlot;
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
#pragma unroll
          for (int c = 0; c < 6; c++) g[rr * kGRow + c] = Ji[rr * 6 + c] * sr;
#pragma unroll
          for (int c = 3; c < 6; c++) g[rr * kGRow + 3 + c] = Jj[rr * 6 + c] * sr;
          g[rr * kGRow + 9] = r[rr] * sr;
        }
        const double s2 = sr * sr;
#pragma unroll
        for (int c = 0; c < 6; c++) HW[(size_t)f * v.n6cap + 6 * t + c] = (Jj[c] * Jl[0] + Jj[6 + c] * Jl[1]) * s2;  // one writer per (landmark, frame)
        VIO_ATOMIC_ADD(w.hff + f, (Jl[0] * Jl[0] + Jl[1] * Jl[1]) * s2);
        VIO_ATOMIC_ADD(w.gf + f, (Jl[0] * r[0] + Jl[1] * r[1]) * s2);
#pragma unroll
        for (int c = 0; c < 6; c++) VIO_ATOMIC_ADD(w.wh + c * w.Fld + f, (Ji[c] * Jl[0] + Ji[6 + c] * Jl[1]) * s2);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      stamp(cx, ST_P_FACT);
      // Gram products of the buckets that reach into [p0, q1): one v_mfma per two factors, A and B operand the same register
      const int q1 = p0 + kLinStage < ws1 ? p0 + kLinStage : ws1;
      while (b < v.npairs) {
        if (b < tb || b >= tb + 64) load_table(b & ~63);
        const int bl = __builtin_amdgcn_readfirstlane(b - tb);
        const int b_s0 = __builtin_amdgcn_readlane(m_s0, bl), b_s1 = __builtin_amdgcn_readlane(m_s1, bl);
        const int b_ht = __builtin_amdgcn_readlane(m_ht, bl);
        if (b_s0 >= q1) break;
        const int s_lo = b_s0 > p0 ? b_s0 : p0, s_hi = b_s1 < q1 ? b_s1 : q1;
        if (s_lo < s_hi) {
          auto g = G + (s_lo - p0 + (kq >> 1)) * kGSlot + (kq & 1) * kGRow + (lv ? src : 0);
          int steps = (s_hi - s_lo) >> 1;  // full two-factor steps; an odd last factor is a masked half step
          for (; steps > 0; steps -= 8, g += 16 * kGSlot) {
            double a[8];
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = g[(j < steps ? 2 * j : 0) * kGSlot];
            VIO_SCHED_FENCE();
#pragma unroll
            for (int j = 0; j < 8; j++) {
              if (j < steps) {  // (uniform)
                const double x = a[j] * sg;
                if (j & 1) acc2 = mfma_f64(x, x, acc2);
                else acc = mfma_f64(x, x, acc);
              }
            }
          }
          g += 2 * kGSlot * steps;  // (steps <= 0: back to the slot behind the last full step)
          if ((s_hi - s_lo) & 1) {  // lanes kq >= 2 would fetch the slot behind the bucket: never read, operand zero
            const bool half = lv && kq < 2;
            double a = G[half ? (int)(g - G) : 0];
            a = half ? a * sg : 0.0;
            acc = mfma_f64(a, a, acc);
          }
          pending = true, pend_ht = b_ht;
        }
        if (b_s1 > q1) break;  // the bucket goes on in the next strip: its sums stay in the accumulators
        if (pending) {
          acc += acc2;
          flush(acc, pend_ht);
          acc = v4d{0.0, 0.0, 0.0, 0.0}, acc2 = v4d{0.0, 0.0, 0.0, 0.0}, pending = false;
        }
        b++;
      }
      stamp(cx, ST_P_GRAM);
    }
    if (pending) {  // the strip ended inside a bucket: the wave of the next strip adds the rest
      acc += acc2;
      flush(acc, pend_ht);
    }
    stamp(cx, ST_M_GRAM);
  }
  VIO_SYNC();
  stamp(cx, ST_M_IMU);

  // ---- C: prior cost / gradient, host-frame coupling rows, diagonal pose blocks, vectors out ------------------------------
  if (n > 0) {
    const int i = tid_ < n ? tid_ : 0;
    const double r0 = v.pr_r[i], b0 = v.prb0[i];
    const int pa = pv.prcol()[i];
    if (tid_ < n) {
      cost += 0.5 * r0 * r0 + 0.5 * w.prdx[i] * (w.prr[i] + b0);  // |r0|^2 / 2 + b0 . dx + dx . H0 dx / 2
      if (pa >= 0) VIO_ATOMIC_ADD(w.gp + kBS * (pa >> 8) + (pa & 255), w.prr[i]);
    }
    for (int k = tid_ + NT; k < n; k += NT) {
      const double r0k = v.pr_r[k], b0k = v.prb0[k];
      cost += 0.5 * r0k * r0k + 0.5 * w.prdx[k] * (w.prr[k] + b0k);
      const int pk = pv.prcol()[k];
      if (pk >= 0) VIO_ATOMIC_ADD(w.gp + kBS * (pk >> 8) + (pk & 255), w.prr[k]);
    }
  }
  VIO_PARFOR(f, F) {
    const int h = pv.fh()[f];
    if (h >= 0)
      for (int c = 0; c < 6; c++) HW[(size_t)f * v.n6cap + 6 * h + c] = w.wh[c * w.Fld + f];
    H[pv.h_gf + f] = w.gf[f], H[pv.h_hff + f] = w.hff[f];
  }
  VIO_PARFOR(q, nF * 36) {
    const int a = q / 36, e = q - a * 36, r = e / 6, c = e - r * 6;
    if (r >= c) VIO_ATOMIC_ADD(Happ + tri_at(6 * a + r, 6 * a + c), w.ppd[q]);
  }
  const double total = block_sum(cx, cost);  // (its barrier also closes the gradient)
  VIO_PARFOR(i, np) H[pv.h_gp + i] = w.gp[i];
  if (cx.tid == 0) H[0] = total;
  stamp(cx, ST_P_FEAT);
}

// =====================================================================================================
// step
// =====================================================================================================
// minimize() of solver_core.h, re-entered once per evaluation. REGS / NW / WK as there.
template <bool REGS, int NW, class WK>
VIO_DEV void phase_step(const Ctx &cx, WinView &v, const PhaseView &pv, WK &w) {
  PhaseRec R = *pv.rec_ptr();
  if (R.phase == PH_DONE) return;
  const int np = v.np, F = v.F, P = v.P, nposes = v.P + v.has_loop;
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
    VIO_PARFOR(i, np) w.t2[i] = -w.gp[i];
    VIO_PARFOR(f, F) w.tf[f] = -w.gf[f];
    VIO_SYNC();
    apply_plus(cx, v, w, w.t2, w.tf);
    double l2, linf;
    state_norms(cx, v, w.xpose, w.xsb, w.xfeat, w.cpose, w.csb, w.cfeat, &l2, &linf);
    return linf;
  };
  // The linearization in buffer `buf` becomes the one the linear algebra works on: vectors and the pose matrix into LDS, the
  // speed-bias band started as the prior's, the IMU factors' whitened Gram products added, Jacobi scaling (first time) and
  // the trust-region diagonal. This is the tail of evaluate(jac = true) in solver_core.h.
  auto adopt = [&](int buf, bool have_scale) {
    const double *H = pv.hb(buf);
    v.WTf = pv.hb(buf) + pv.h_WTf, v.imu_J = pv.hb(buf) + pv.h_imuJ, v.imu_r = pv.hb(buf) + pv.h_imur;
    const int napp = (int)tri_doubles(v.nrows), nband = 2 * v.P * kSS;
    {
      constexpr int kU = 10;
      const double *Ha = H + pv.h_App;
      const int t = VIO_TID(cx), NT = (int)cx.nt;
      const double g0 = H[pv.h_gp + (t < np ? t : 0)], g1 = H[pv.h_gf + (t < F ? t : 0)], g2 = H[pv.h_hff + (t < F ? t : 0)];
      for (int q0 = t; q0 < napp + nband; q0 += kU * (int)cx.nt) {
        double x[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
          const int q = q0 + u * (int)cx.nt;
          const double pr = v.prior_n > 0 ? v.AppPr[q < napp + nband ? q : 0] : 0.0;  // the prior's H0 in the matrix layout (setup_prior)
          x[u] = q < napp ? Ha[q] + pr : pr;
        }
        VIO_SCHED_FENCE();
#pragma unroll
        for (int u = 0; u < kU; u++) {
          const int q = q0 + u * (int)cx.nt;
          if (q < napp) w.App[q] = x[u];
          else if (q < napp + nband) w.Dss[q - napp] = x[u];
        }
      }
      if (t < np) w.gp[t] = g0;
      if (t < F) w.gf[t] = g1, w.hff[t] = g2;
      for (int f = t + NT; f < F; f += NT) w.gf[f] = H[pv.h_gf + f], w.hff[f] = H[pv.h_hff + f];
    }
    VIO_PARFOR(q, v.P * kAS) w.AspI[q] = 0.0;
    VIO_SYNC();
    stamp(cx, ST_EVAL_PRIOR);
    {
      // One wave per IMU factor on the matrix cores (solver_core.h evaluate()): T = info [Jraw | r], G = [Jraw | r]^T T
      const int tid_ = VIO_TID(cx), wave = tid_ >> 6, nw = cx.nt >> 6, lane = tid_ & 63;
      const int n = lane & 15, kq = lane >> 4;
      // the operands of a wave's next factor are fetched before the products of the current one are formed
      double avs[4], bvs[2][4];
      auto fetch = [&](int f) {
        const double *info = v.imu_info + f * 225, *Jr = v.imu_J + f * 450, *rr = v.imu_r + f * 15;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
          const int k = 4 * s4 + kq, kc = k < 15 ? k : 0;
          avs[s4] = info[(n < 15 ? n : 0) * 15 + kc];
          bvs[0][s4] = Jr[kc * 30 + n];
          bvs[1][s4] = (n < 14) ? Jr[kc * 30 + 16 + (n < 14 ? n : 0)] : rr[kc];
        }
      };
      if (wave < v.W) fetch(wave);
      for (int f = wave; f < v.W; f += nw) {
        double av[4], bv[2][4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
          const bool kok = 4 * s4 + kq < 15;
          av[s4] = (kok && n < 15) ? avs[s4] : 0.0;
          bv[0][s4] = kok ? bvs[0][s4] : 0.0;
          bv[1][s4] = (kok && n < 15) ? bvs[1][s4] : 0.0;
        }
        if (f + nw < v.W) fetch(f + nw);
        v4d T0 = {0, 0, 0, 0}, T1 = {0, 0, 0, 0};
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) T0 = mfma_f64(av[s4], bv[0][s4], T0), T1 = mfma_f64(av[s4], bv[1][s4], T1);
        v4d G00 = {0, 0, 0, 0}, G10 = {0, 0, 0, 0}, G11 = {0, 0, 0, 0};
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
          G00 = mfma_f64(bv[0][s4], T0[s4], G00);
          G10 = mfma_f64(bv[1][s4], T0[s4], G10);
          G11 = mfma_f64(bv[1][s4], T1[s4], G11);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
          const int row = kq + 4 * r4;
          if (row >= n) red_put(v, w, f + (row >= 15), row >= 15 ? row - 15 : row, f + (n >= 15), n >= 15 ? n - 15 : n, G00[r4], true);
          const int Rr = 16 + row;
          if (Rr < 30) {
            red_put(v, w, f + 1, Rr - 15, f + (n >= 15), n >= 15 ? n - 15 : n, G10[r4], true);
            if (n < 14 && Rr >= 16 + n) red_put(v, w, f + 1, Rr - 15, f + 1, n + 1, G11[r4], true);
          } else if (Rr == 30) {
            VIO_ATOMIC_ADD(w.gp + 15 * f + n, G10[r4]);
            if (n < 14) VIO_ATOMIC_ADD(w.gp + 15 * f + 16 + n, G11[r4]);
          }
        }
      }
    }
    VIO_SYNC();
    VIO_PARFOR(i, np) {
      const int f = i / kBS, c = i - f * kBS;
      const double h = c < 6 ? w.App[tri_at(6 * f + c, 6 * f + c)] : w.Dss[f * kSS + (c - 6) * (kSB + 1)];
      if (!have_scale) w.sp[i] = rcp_f(1.0 + sqrt_f(h)), pv.sp()[i] = w.sp[i];
      const double sc = w.sp[i];
      w.dp[i] = sqrt_f(fmin(fmax(sc * sc * h, 1e-6), 1e32));
      pv.dp()[i] = w.dp[i], pv.gpf()[i] = w.gp[i];
    }
    VIO_SYNC();
    stamp(cx, ST_EVAL_IMU);
  };

  // ---- entry: the iterate and what the first decision needs -------------------------------------------------------------
  int cur = R.cur;
  {
    // every load of a lane is issued before its first store: one global round trip for the whole state instead of one per array
    const double *Xc = pv.xb(R.phase == PH_FIRST ? 0 : cur), *Xn = pv.xb(1 - cur);
    const int t = VIO_TID(cx), NT = (int)cx.nt, n7 = nposes * 7, n9 = P * 9;
    const bool cand = R.phase == PH_CAND;
    const double a0 = Xc[t < n7 ? t : 0], a1 = Xc[pv.x_sb + (t < n9 ? t : 0)], a2 = Xc[pv.x_feat + (t < F ? t : 0)], a3 = v.ex[t < 7 ? t : 0];
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    if (cand) {
      c0 = Xn[t < n7 ? t : 0], c1 = Xn[pv.x_sb + (t < n9 ? t : 0)], c2 = Xn[pv.x_feat + (t < F ? t : 0)];
      c3 = pv.sp()[t < np ? t : 0], c4 = pv.sf()[t < F ? t : 0];
    }
    VIO_SCHED_FENCE();
    if (t < n7) w.xpose[t] = a0;
    if (t < n9) w.xsb[t] = a1;
    if (t < F) w.xfeat[t] = a2;
    if (t < 7) w.ex[t] = a3;
    if (cand) {
      if (t < n7) w.cpose[t] = c0;
      if (t < n9) w.csb[t] = c1;
      if (t < F) w.cfeat[t] = c2;
      if (t < np) w.sp[t] = c3;
      if (t < F) w.sf[t] = c4;
    }
    for (int f = t + NT; f < F; f += NT) {  // (windows with more landmarks than work-items)
      w.xfeat[f] = Xc[pv.x_feat + f];
      if (cand) w.cfeat[f] = Xn[pv.x_feat + f], w.sf[f] = pv.sf()[f];
    }
    VIO_PARFOR(q, v.nblk * kBS) w.t1[q] = 0.0, w.t2[q] = 0.0;
    if (cx.tid == 0) w.flag[0] = w.flag[1] = w.flag[2] = w.flag[3] = 0;
    VIO_PARFOR(k, P) {
      int lo = 6 * (k > 0 ? k - 1 : 0), pr = 0;
      for (int b = 0; b < v.prior_nb; b++)
        if (v.pr_kind[b] == 1 && v.pr_index[b] == k) lo = 0, pr = 1;
      w.sbr[2 * k] = lo, w.sbr[2 * k + 1] = pr;
    }
    VIO_SYNC();
  }
  stamp(cx, ST_SETUP_IMU);
  double x_cost = R.x_cost, x_norm = R.x_norm, gmax = R.gmax, radius = R.radius, mu = R.mu, mu_used = R.mu_used;
  double dogleg_step_norm = R.dogleg_step_norm, alpha = R.alpha, gd_sq = R.gd_sq, qf_cauchy = R.qf_cauchy;
  double ev_min = R.ev_min, ev_cur = R.ev_cur, ev_ref = R.ev_ref, ev_cand = R.ev_cand, ev_acc_ref = R.ev_acc_ref, ev_acc_cand = R.ev_acc_cand;
  double min_rec = R.min_rec, model_cost_change = R.model_cost_change;
  int it = R.it, n_ok = R.n_ok, n_bad = R.n_bad, invalid_run = R.invalid_run, termination = R.termination, recorded = R.recorded;
  bool reuse = R.reuse != 0, last_ok = R.last_ok != 0;
  const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
  bool done = false;

  if (R.phase == PH_FIRST) {
    cur = 0;
    adopt(0, false);
    x_cost = pv.hb(0)[0];
    VIO_PARFOR(f, F) w.sf[f] = rcp_f(1.0 + sqrt_f(w.hff[f])), pv.sf()[f] = w.sf[f];  // Jacobi scaling, :239-254
    VIO_SYNC();
    gmax = grad_max_norm();
    ev_min = ev_cur = ev_ref = ev_cand = x_cost, ev_acc_ref = ev_acc_cand = 0, min_rec = x_cost;
    record(0, x_cost, radius, 0, 0, gmax, true, true);
    if (cx.tid == 0) sd[0] = x_cost;
  } else {
    // the candidate written by the previous launch has been evaluated: step acceptance (trust_region_minimizer.cc:428-640)
    double cand_cost = pv.hb(1 - cur)[0];
    if (!isfinite(cand_cost)) cand_cost = 1.7976931348623157e308;
    double step_norm, dummy;
    state_norms(cx, v, w.xpose, w.xsb, w.xfeat, w.cpose, w.csb, w.cfeat, &step_norm, &dummy);
    const double cost_change = x_cost - cand_cost;
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) termination = 1, done = true;          // ParameterToleranceReached
    else if (fabs(cost_change) <= 1e-6 * x_cost) termination = 1, done = true;     // FunctionToleranceReached
    if (!done) {
      const double rel = (ev_cur - cand_cost) / model_cost_change;                  // StepQuality
      const double hist = (ev_ref - cand_cost) / (ev_acc_ref + model_cost_change);
      const double rho = fmax(rel, hist);
      if (rho > 1e-3) {
        VIO_PARFOR(q, nposes * 7) w.xpose[q] = w.cpose[q];
        VIO_PARFOR(q, P * 9) w.xsb[q] = w.csb[q];
        VIO_PARFOR(q, F) w.xfeat[q] = w.cfeat[q];
        VIO_SYNC();
        cur ^= 1;  // the candidate's buffers are the iterate's from here on
        state_norms(cx, v, w.xpose, w.xsb, w.xfeat, nullptr, nullptr, nullptr, &x_norm, nullptr);
        adopt(cur, true);
        x_cost = cand_cost;  // (the linearization was evaluated together with the cost, at the same point)
        gmax = grad_max_norm();
        if (rho < 0.25) radius *= 0.5;                                              // StepAccepted
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
        recorded = it + 1, min_rec = fmin(min_rec, x_cost);
      } else {
        radius *= 0.5;                                                              // StepRejected
        reuse = true;
        last_ok = false;
        n_bad++;
        record(it, cand_cost, radius, step_norm, rho, 0.0, true, false);
        recorded = it + 1, min_rec = fmin(min_rec, cand_cost);
        // the accepted linearization's vectors for the next dogleg step (the matrix is only needed again if that step turns
        // out invalid: adopt() then)
        const double *H = pv.hb(cur);
        v.WTf = pv.hb(cur) + pv.h_WTf;
        {
          const int t = VIO_TID(cx), NT = (int)cx.nt, ip = t < np ? t : 0, jf = t < F ? t : 0;
          const double b0 = pv.gpf()[ip], b1 = pv.dp()[ip], b2 = pv.gnp()[ip], b3 = H[pv.h_gf + jf], b4 = H[pv.h_hff + jf], b5 = pv.gnf()[jf];
          VIO_SCHED_FENCE();
          if (t < np) w.gp[t] = b0, w.dp[t] = b1, w.gnp[t] = b2;
          if (t < F) w.gf[t] = b3, w.hff[t] = b4, w.gnf[t] = b5;
          for (int f = t + NT; f < F; f += NT) w.gf[f] = H[pv.h_gf + f], w.hff[f] = H[pv.h_hff + f], w.gnf[f] = pv.gnf()[f];
        }
        VIO_SYNC();
      }
    }
  }

  stamp(cx, ST_COST_EVAL);
  while (!done) {
    if (it >= v.max_iter) break;
    if (last_ok && gmax <= 1e-10) { termination = 1; break; }
    if (radius <= 1e-32) { termination = 1; break; }
    it++;
    bool solver_ok = true;
    if (!reuse) {
      reuse = true;
      // The loop-carried scalars that the linear solve does not touch leave the registers for its duration (they are the
      // same in every lane, but values that come out of LDS reductions live in VGPRs: ~35 registers per lane that the panel
      // steps of the factorization are short of).
      ldsd park = w.park;
      if (cx.tid == 0) {
        park[0] = x_cost, park[1] = x_norm, park[2] = gmax, park[3] = radius, park[4] = dogleg_step_norm, park[5] = ev_min;
        park[6] = ev_cur, park[7] = ev_ref, park[8] = ev_cand, park[9] = ev_acc_ref, park[10] = ev_acc_cand, park[11] = min_rec;
        ldsi pi = reinterpret_cast<ldsi>(park + 12);
        pi[0] = it, pi[1] = n_ok, pi[2] = n_bad, pi[3] = invalid_run, pi[4] = termination, pi[5] = recorded, pi[6] = last_ok ? 1 : 0;
      }
      double part = 0;
      VIO_PARFOR(i, np) {
        const double g = pose_gd(w, i);
        part += g * g;
      }
      VIO_PARFOR(f, F) {
        const double g = feat_gd(w, f);
        part += g * g;
      }
      gd_sq = block_sum(cx, part);
      auto cauchy_direction = [&]() {  // a = D^-2 S g -> t2 (poses), stf (landmarks)
        VIO_PARFOR(i, np) w.t2[i] = pose_gd(w, i) * rcp_f(w.dp[i]);
        VIO_PARFOR(f, F) w.stf[f] = w.sf[f] * w.gf[f] * rcp_f(feat_d2(w, f));
        VIO_SYNC();
      };
      cauchy_direction();
      stamp(cx, ST_DOGLEG);
      const double qf_h = quad_form_H(cx, v, w, w.t2, w.stf);
      stamp(cx, ST_QUADFORM);
      solver_ok = false;
      bool first_try = true;
      while (mu < max_mu) {
        if (!first_try) {
          adopt(cur, true);    // retry with a larger mu: the in-place system was consumed, the linearization is still in its buffer
          cauchy_direction();
        }
        first_try = false;
        if (cx.tid == 0) w.flag[0] = 0, w.flag[1] = 0, w.flag[2] = 0, w.flag[3] = 0;
        VIO_SYNC();
        bool ok = build_reduced_system(cx, v, w, mu);
        if (ok) {
          if constexpr (REGS) ok = factor_band_regs<kPanelTiles, NW>(cx, v, w);
          else ok = factor_band_lds(cx, v, w);
        }
        if (ok) ok = factor_poses(cx, v, w);
        stamp(cx, ST_CHOL);
        if (ok) {
          backsolve(cx, v, w);  // z -> t1, w_f^T z_p -> gnf
          double bad = 0;
          VIO_PARFOR(f, F) {
            double y = (w.tf[f] - w.gnf[f] * w.einv[f]) * rcp_f(w.sf[f]);
            w.gnf[f] = -feat_d(w, f) * y;
            if (!isfinite(y)) bad = 1;
          }
          VIO_PARFOR(i, np) {
            double y = w.t1[i] * rcp_f(w.sp[i]);
            w.gnp[i] = -w.dp[i] * y;
            if (!isfinite(y)) bad = 1;
          }
          if (block_max(cx, bad) > 0) ok = false;
          stamp(cx, ST_TRISOLVE);
        }
        if (ok) { solver_ok = true; mu_used = mu; break; }
        mu *= mu_inc;
      }
      {
        VIO_SYNC();
        x_cost = park[0], x_norm = park[1], gmax = park[2], radius = park[3], dogleg_step_norm = park[4], ev_min = park[5];
        ev_cur = park[6], ev_ref = park[7], ev_cand = park[8], ev_acc_ref = park[9], ev_acc_cand = park[10], min_rec = park[11];
        ldsi pi = reinterpret_cast<ldsi>(park + 12);
        it = pi[0], n_ok = pi[1], n_bad = pi[2], invalid_run = pi[3], termination = pi[4], recorded = pi[5], last_ok = pi[6] != 0;
      }
      if (solver_ok) {
        double part2 = 0;
        VIO_PARFOR(i, np) part2 += mu_used * w.dp[i] * w.dp[i] * w.t2[i] * w.t2[i];
        VIO_PARFOR(f, F) part2 += mu_used * feat_d2(w, f) * w.stf[f] * w.stf[f];
        const double reg = block_sum(cx, part2);
        qf_cauchy = qf_h + reg;
        alpha = gd_sq / qf_h;
        VIO_PARFOR(i, np) pv.gnp()[i] = w.gnp[i];
        VIO_PARFOR(f, F) pv.gnf()[f] = w.gnf[f];
      }
    }
    bool step_valid = false;
    model_cost_change = 0;
    if (solver_ok) {
      // ComputeTraditionalDoglegStep (dogleg_strategy.cc:199-255)
      double p1 = 0, p2 = 0;
      VIO_PARFOR(i, np) p1 += w.gnp[i] * w.gnp[i], p2 += pose_gd(w, i) * w.gnp[i];
      VIO_PARFOR(f, F) p1 += w.gnf[f] * w.gnf[f], p2 += feat_gd(w, f) * w.gnf[f];
      double pdummy = 0;
      block_sum3(cx, p1, p2, pdummy);
      const double gnn2 = p1, gdot = p2;
      const double gradient_norm = sqrt(gd_sq), gauss_newton_norm = sqrt(gnn2);
      double ca, cb;
      bool need_norm = false;
      if (gauss_newton_norm <= radius) {
        ca = 0, cb = 1, dogleg_step_norm = gauss_newton_norm;
      } else if (gradient_norm * alpha >= radius) {
        ca = -(radius / gradient_norm), cb = 0, dogleg_step_norm = radius;
      } else {
        double b_dot_a = -alpha * gdot;
        double a_squared_norm = pow(alpha * gradient_norm, 2.0);
        double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + pow(gauss_newton_norm, 2);
        double c = b_dot_a - a_squared_norm;
        double d = sqrt(c * c + b_minus_a_squared_norm * (pow(radius, 2.0) - a_squared_norm));
        double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + c);
        ca = -alpha * (1.0 - beta), cb = beta;
        need_norm = true;
      }
      double pn = 0, psg = 0, preg = 0;
      VIO_PARFOR(i, np) {
        double s = ca * pose_gd(w, i) + cb * w.gnp[i];
        pn += s * s;
        double st = s * rcp_f(w.dp[i]);
        w.stp[i] = st;
        psg += st * w.sp[i] * w.gp[i];
        preg += mu_used * w.dp[i] * w.dp[i] * st * st;
      }
      VIO_PARFOR(f, F) {
        const double d2 = feat_d2(w, f), id = rsqrt_f(d2);
        double s = ca * (w.sf[f] * w.gf[f] * id) + cb * w.gnf[f];
        pn += s * s;
        double st = s * id;
        w.stf[f] = st;
        psg += st * w.sf[f] * w.gf[f];
        preg += mu_used * d2 * st * st;
      }
      VIO_SYNC();
      block_sum3(cx, pn, psg, preg);
      const double n2 = pn, sg = psg, reg = preg;
      if (need_norm) dogleg_step_norm = sqrt(n2);
      const double shs = ca * ca * qf_cauchy - 2.0 * ca * cb * gd_sq - cb * cb * gdot - reg;
      model_cost_change = -sg - 0.5 * shs;
      step_valid = model_cost_change > 0.0;
    }
    if (!step_valid) {
      if (++invalid_run >= 5) { termination = 2; break; }
      mu *= mu_inc;
      reuse = false;
      last_ok = false;
      n_bad++;
      record(it, x_cost, radius, 0, 0, gmax, false, false);
      recorded = it + 1, min_rec = fmin(min_rec, x_cost);
      adopt(cur, true);  // the matrix buffer holds a factorization: the linearization is taken from its buffer again
      continue;
    }
    invalid_run = 0;
    VIO_PARFOR(i, np) w.t2[i] = w.stp[i] * w.sp[i];  // delta = step * scale
    VIO_PARFOR(f, F) w.tf[f] = w.stf[f] * w.sf[f];
    VIO_SYNC();
    apply_plus(cx, v, w, w.t2, w.tf);
    stamp(cx, ST_DOGLEG);
    // the candidate leaves for its evaluation (linearize kernel); this kernel is entered again with its cost
    {
      double *Xn = pv.xb(1 - cur);
      VIO_PARFOR(q, nposes * 7) Xn[q] = w.cpose[q];
      VIO_PARFOR(q, P * 9) Xn[pv.x_sb + q] = w.csb[q];
      VIO_PARFOR(q, F) Xn[pv.x_feat + q] = w.cfeat[q];
    }
    if (cx.tid == 0) {
      PhaseRec S;
      S.x_cost = x_cost, S.x_norm = x_norm, S.gmax = gmax, S.radius = radius, S.mu = mu, S.mu_used = mu_used;
      S.dogleg_step_norm = dogleg_step_norm, S.alpha = alpha, S.gd_sq = gd_sq, S.qf_cauchy = qf_cauchy;
      S.ev_min = ev_min, S.ev_cur = ev_cur, S.ev_ref = ev_ref, S.ev_cand = ev_cand, S.ev_acc_ref = ev_acc_ref, S.ev_acc_cand = ev_acc_cand;
      S.min_rec = min_rec, S.model_cost_change = model_cost_change;
      S.phase = PH_CAND, S.cur = cur, S.it = it, S.n_ok = n_ok, S.n_bad = n_bad, S.invalid_run = invalid_run, S.termination = termination;
      S.recorded = recorded, S.reuse = reuse ? 1 : 0, S.last_ok = last_ok ? 1 : 0;
      *pv.rec_ptr() = S;
    }
    stamp(cx, ST_NEW2OLD);
    return;
  }
  // the minimizer has returned
  if (cx.tid == 0) {
    sd[1] = min_rec;
    si[0] = recorded, si[1] = termination, si[2] = n_ok, si[3] = n_bad;
    PhaseRec S = R;
    S.phase = PH_DONE, S.cur = cur, S.it = it;
    *pv.rec_ptr() = S;
  }
}

// =====================================================================================================
// finish: the iterate back into LDS; raw outputs, new2old (VINS.cpp:131-212) + old2new, outputs. The marginalization
// (marg_core.h) follows in the same kernel on the gauge-fixed state, as in the single-launch path.
// =====================================================================================================
template <class WK>
VIO_DEV void phase_finish(const Ctx &cx, WinView &v, const PhaseView &pv, WK &w) {
  const int P = v.P, F = v.F, nposes = v.P + v.has_loop;
  const int cur = pv.rec_ptr()->cur;
  const double *Xc = pv.xb(cur);
  v.WTf = pv.hb(cur) + pv.h_WTf, v.imu_J = pv.hb(cur) + pv.h_imuJ, v.imu_r = pv.hb(cur) + pv.h_imur;
  VIO_PARFOR(q, nposes * 7) w.xpose[q] = Xc[q];
  VIO_PARFOR(q, P * 9) w.xsb[q] = Xc[pv.x_sb + q];
  VIO_PARFOR(q, F) w.xfeat[q] = Xc[pv.x_feat + q];
  VIO_PARFOR(q, 7) w.ex[q] = v.ex[q];
  VIO_SYNC();
  VIO_PARFOR(q, P * 7) v.raw_pose[q] = w.xpose[q];
  VIO_PARFOR(q, P * 9) v.raw_sb[q] = w.xsb[q];
  VIO_PARFOR(q, F) v.raw_feat[q] = w.xfeat[q];
  if (v.has_loop) VIO_PARFOR(q, 7) v.out_loop[q] = w.xpose[7 * P + q];
  VIO_SYNC();
  double R0in[9], ypr0[3], R00[9], ypr00[3], rot_diff[9];
  qtoR(qnormalized(qfrom_pose(v.pose0)), R0in);
  R2ypr(R0in, ypr0);
  double origin_yaw = v.use_origin ? v.origin_yaw : ypr0[0];
  double op[3] = {v.use_origin ? v.origin_p[0] : v.pose0[0], v.use_origin ? v.origin_p[1] : v.pose0[1],
                  v.use_origin ? v.origin_p[2] : v.pose0[2]};
  qtoR(qfrom_pose(w.xpose), R00);
  R2ypr(R00, ypr00);
  double yd[3] = {origin_yaw - ypr00[0], 0, 0};
  ypr2R(yd, rot_diff);
  double p0[3] = {w.xpose[0], w.xpose[1], w.xpose[2]};
  VIO_SYNC();
  VIO_PARFOR(i, P) {
    auto pp = w.xpose + 7 * i, sbv = w.xsb + 9 * i;
    double Rq[9], Rs[9], d[3] = {pp[0] - p0[0], pp[1] - p0[1], pp[2] - p0[2]}, Ps[3], Vs[3];
    qtoR(qnormalized(qfrom_pose(pp)), Rq);
    mat3mul(rot_diff, Rq, Rs);
    mat3vec(rot_diff, d, Ps);
    double vv[3] = {sbv[0], sbv[1], sbv[2]};
    mat3vec(rot_diff, vv, Vs);
    Quat q = RtoQ(Rs);
    auto po = w.cpose + 7 * i, so = w.csb + 9 * i;
    for (int k = 0; k < 3; k++) po[k] = Ps[k] + op[k], so[k] = Vs[k];
    po[3] = q.x, po[4] = q.y, po[5] = q.z, po[6] = q.w;
    for (int k = 3; k < 9; k++) so[k] = sbv[k];
  }
  VIO_PARFOR(f, F) {  // setDepth / getDepthVector round trip (feature_manager.cpp:300-349)
    double estimated_depth = 1.0 / w.xfeat[f];
    w.cfeat[f] = 1. / estimated_depth;
  }
  VIO_SYNC();
  VIO_PARFOR(q, P * 7) v.out_pose[q] = w.cpose[q], w.xpose[q] = w.cpose[q];
  VIO_PARFOR(q, P * 9) v.out_sb[q] = w.csb[q], w.xsb[q] = w.csb[q];
  VIO_PARFOR(q, F) v.out_feat[q] = w.cfeat[q], w.xfeat[q] = w.cfeat[q];
  VIO_SYNC();
}

#endif  // !VIO_EMUL

}  // namespace vio
