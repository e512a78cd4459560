// marg_core.h — building the next linearized prior on the device (SPMD phase code, same conventions as
// solver_core.h). Runs in the same launch right after the solve + new2old, on the gauge-fixed state.
//
// Reference: MarginalizationInfo::{preMarginalize, marginalize} (VINS_ios/marginalization_factor.cpp:118-300) at the
// two call sites in VINS::solve_ceres (VINS_ios/VINS.cpp:690-774 MARGIN_OLD, :776-830 MARGIN_SECOND_NEW).
//
// Route. The reference forms A = sum J^T J, b = sum J^T r over {prior, IMU(0,1), projections hosted at frame 0},
// orders the dropped blocks first and computes
//     A' = Arr - Arm Amm^+ Amr,  b' = br - Arm Amm^+ bm,  J0 = sqrt(S) V^T,  r0 = sqrt(S)^-1 V^T b'   (A' = V S V^T)
// with two dense symmetric eigendecompositions whose eigenvalues <= 1e-8 are cut. The next solve consumes the prior
// only through J0^T J0 = A'_+ , J0^T r0 = b'_+ and |r0|^2. Any factor A'_+ = J0^T J0 serves, so the device does ONE
// Cholesky-type factorization of A with the dropped variables first and b carried along:
//     A = L L^T  =>  the trailing block of L is L' with L' L'^T = A',  forward substitution gives r0 = L'^-1 b',
//     J0 = L'^T.
// Landmarks hosted at frame 0 (1x1 blocks) are eliminated analytically first, exactly like in the solver. The
// eigenvalue cut becomes a pivot cut: a pivot <= max(1e-8, 1e-12 * original diagonal) marks a direction without
// information (gauge directions, blocks that no factor touches); its row of L is zeroed and skipped, which is what
// the pseudo-inverse / zeroed eigenvalue does in the reference up to rounding noise.
#pragma once

#include "solver_core.h"

namespace vio {

struct MargOut {
  int *n;        // [4]: n, n_blocks, m (dropped dims), pos
  int *kind, *index, *offset;  // [kMaxPriorBlocks]
  double *x0;    // [kMaxPriorBlocks][9]
  double *J;     // [n][n] row-major
  double *r;     // [n]
  double *scratch;  // global: dense matrix when it does not fit LDS
  int ncap;         // capacity of J (ncap x ncap) and r
};

template <class MP>
struct MargWorkT {
  MP Am;         // pos x pos, lower triangle by 16-row tiles (tri_at, solver_core.h); LDS when it fits, else global
  int ld;        // capacity: largest pos the buffer holds
  ldsd bm;       // pos
  ldsd tol;      // pos
  ldsd ldinv;    // pos: 1 / L_jj (0 for a cut pivot)
  ldsd hff, gf, einv;  // F
  ldsd prdx, prr;      // prior_n
  ldsi col_pose;  // [P+1]: first dense column of pose i (-1: not involved); entry P unused
  ldsi col_sb;    // [P]
  ldsi col_ex;    // [1]
  ldsi pcol;      // prior column -> dense column: prior_n
  ldsi meta;      // [4]: pos, m, n, nblocks
  ldsd stage;     // staging of robustified Jacobian rows for the Gram products: LDS in both variants (whatever the phase's
  int stage_slots;  // vectors -- and the matrix, when it is in LDS -- leave of the workgroup's allocation)
};

// int pointer in the address space of a double pointer type (LDS or global)
template <class P>
struct IntPtrOf;
template <>
struct IntPtrOf<double *> {
  typedef int *type;
};
#ifndef VIO_HOST_BUILD
template <>
struct IntPtrOf<ldsd> {
  typedef ldsi type;
};
#endif

constexpr int kMargRowX = 6;                                   // extrinsic Jacobian row
constexpr int kMargSlot = kSlotStride + 2 * kMargRowX + 1;     // [Ji Jj r Jl] x2 + pad, [Jex] x2 + pad = 42

VIO_HD constexpr int kMargMaxPos(int W) { return 15 + 6 * W + 15; }

VIO_HD size_t marg_scratch_doubles(const int Wcap) {
  size_t p = (size_t)kMargMaxPos(Wcap);
  return tri_doubles((int)p) + 8;  // packed matrix (global-matrix variant)
}

// LDS carve for the marginalization phase. The solver's iterate (xpose, xsb, xfeat, ex) sits at the front of LDS and
// is preserved; everything behind it is re-used. Returns bytes used (base may be null to just measure).
// By value (the kernel must not take the address of its MargWorkT, see carve_all in batch.h).
template <class MP>
struct CarvedMarg {
  MargWorkT<MP> m;
  size_t bytes;
};
template <class MP, class Dims>
VIO_HD CarvedMarg<MP> carve_marg_all(const Dims &d, bool lds_matrix, ldsd base_after_state, double *am_global,
                                     size_t avail_doubles = 0) {
  CarvedMarg<MP> c;
  size_t o = 0;
  auto take = [&](size_t n) {
    ldsd p = base_after_state + o;
    o += (n + 1) & ~(size_t)1;
    return p;
  };
  const size_t pos = (size_t)kMargMaxPos(d.Wcap), F = d.Flds;
  const size_t nam = tri_doubles((int)pos);
  ldsd Am = lds_matrix ? take(nam) : nullptr;
  MargWorkT<MP> &m = c.m;
  m.bm = take(pos + 16), m.tol = take(pos + 16), m.ldinv = take(pos + 16);  // (+16: whole 16-wide tiles)
  m.hff = take(F), m.gf = take(F), m.einv = take(F);
  m.prdx = take(d.Ncap), m.prr = take(d.Ncap);
  ldsd ints = take(((size_t)(2 * d.Pcap + 2 + d.Ncap + 4) + 1) / 2 + 1);
  // whatever LDS is left (the solver's footprint is larger than the marginalization core) stages Jacobian rows -- also when
  // the matrix lives in global scratch (the staging area used to follow it there: two L2 round trips per operand batch)
  size_t stage_slots = 0;
  if (avail_doubles > o + kMargSlot * 2) stage_slots = ((avail_doubles - o) / kMargSlot) & ~(size_t)1;
  if (stage_slots > 1024) stage_slots = 1024;
  ldsd stage = take(stage_slots * kMargSlot);
  m.stage = stage, m.stage_slots = (int)stage_slots;
  m.Am = MatPick<MP>::get(lds_matrix, Am, am_global), m.ld = (int)pos;
  ldsi ip = reinterpret_cast<ldsi>(ints);
  m.col_pose = ip, m.col_sb = ip + d.Pcap + 1, m.col_ex = m.col_sb + d.Pcap;
  m.pcol = m.col_ex + 1, m.meta = m.pcol + d.Ncap;
  c.bytes = o * sizeof(double);
  return c;
}
// Pointer form for host code. Returns bytes used (base may be null to just measure).
template <class MP, class Dims>
VIO_HD size_t carve_marg(const Dims &d, bool lds_matrix, ldsd base_after_state, double *am_global, MargWorkT<MP> *m,
                         size_t avail_doubles = 0) {
  const CarvedMarg<MP> c = carve_marg_all<MP>(d, lds_matrix, base_after_state, am_global, avail_doubles);
  if (m) *m = c.m;
  return c.bytes;
}

#ifndef VIO_EMUL
// ---- tiled factorization of the dense marginalization matrix on the matrix cores ----------------------------------
// 16 x 16 tiles of the row-major pos x pos matrix (leading dimension ld), the structure of cholesky_blocks
// (solver_core.h): one wave factors the diagonal tile and produces its inverse, TRSM and the trailing update are MFMA
// products, wave 0 looks ahead. Differences: a pivot <= tol is CUT (its column of L, its row of L^-1 and its entry of
// the carried right-hand side become 0 -- the pseudo-inverse of the reference), and the last tile may be partial.
// Per-lane offsets: operand X[i][kq + 4 s] at i * ld + kq + 4 s, accumulator C[kq + 4 r][i] at (kq + 4 r) * ld + i.
template <class MP>
VIO_DEV void dtile_load_op(MP X, int op, double out[4]) {
  out[0] = X[op], out[1] = X[op + 4], out[2] = X[op + 8], out[3] = X[op + 12];
}
template <class MP>
VIO_DEV v4d dtile_load_acc(MP C, int acc, int ld) {
  v4d a;
  a[0] = C[acc], a[1] = C[acc + 4 * ld], a[2] = C[acc + 8 * ld], a[3] = C[acc + 12 * ld];
  return a;
}
template <class MP>
VIO_DEV void dtile_store_acc(MP C, int acc, int ld, v4d a, int kq, int rows, bool col_ok = true) {  // rows: valid rows of the tile
#pragma unroll
  for (int r = 0; r < 4; r++)
    if (col_ok && kq + 4 * r < rows) C[acc + 4 * r * ld] = a[r];
}

// Diagonal tile: L (lower, with diagonal) in place, the strict lower part of L^-1 transposed above the diagonal,
// 1 / L_cc (0: cut) in ldinv_k. nvalid: rows / columns of the tile inside the matrix; tolv: lane n holds tol of column n.
template <class MP, class LP>
VIO_DEV void potrf16_cut_wave(MP D, LP Lprev, int ld, int nvalid, bool with_update, double tolv, ldsd ldinv_k, int lane) {
  const int n = lane & 15, kq = lane >> 4;
  v4d A, E;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int mm = kq + 4 * r;
    const bool ok = mm < nvalid && n < nvalid;
    const int hi = mm > n ? mm : n, lo = mm > n ? n : mm;
    const double x = D[ok ? hi * ld + lo : 0];
    A[r] = ok ? x : 0.0;
    E[r] = (mm == n) ? 1.0 : 0.0;
  }
  if (with_update) {  // look-ahead: D -= Lprev Lprev^T (the panel tile left of D)
    double l[4];
    dtile_load_op(Lprev, (n < nvalid ? n : 0) * ld + kq, l);
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const double ls = n < nvalid ? l[s] : 0.0;
      A = mfma_f64(-ls, ls, A);
    }
  }
  // (round 6: four pivots per step like potrf16_wave of solver_core.h -- M = the inverse Cholesky factor of the 4 x 4 diagonal block,
  // V = M R and T -= V^T V one matrix instruction each; a cut pivot has y = 0: its row of M, hence its column of L, its row of L^-1
  // and its reciprocal are zero, exactly what the rank-1 form left)
  double keep[4] = {0.0, 0.0, 0.0, 0.0}, myinv = 0.0;
  const int mi = (n < 4 && kq <= n) ? n * (n + 1) / 2 + kq : -1;
  auto rsq_cut = [](double d, double tol) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return (d > tol) ? y : 0.0;  // cut pivot (also NaN): the direction carries no information
  };
#pragma unroll
  for (int cb = 0; cb < 4; cb++) {
    double b[10], tol[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      tol[i] = lane_bcast(tolv, 4 * cb + i);
#pragma unroll
      for (int j = 0; j <= i; j++) b[i * (i + 1) / 2 + j] = lane_bcast(A[cb], 16 * i + 4 * cb + j);
    }
    const double y0 = rsq_cut(b[0], tol[0]);
    const double l10 = b[1] * y0, l20 = b[3] * y0, l30 = b[6] * y0;
    const double y1 = rsq_cut(fma(-l10, l10, b[2]), tol[1]);
    const double l21 = fma(-l20, l10, b[4]) * y1, l31 = fma(-l30, l10, b[7]) * y1;
    const double y2 = rsq_cut(fma(-l21, l21, fma(-l20, l20, b[5])), tol[2]);
    const double l32 = fma(-l31, l21, fma(-l30, l20, b[8])) * y2;
    const double y3 = rsq_cut(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, b[9]))), tol[3]);
    double M[10];
    M[0] = y0;
    M[1] = -(l10 * y0) * y1, M[2] = y1;
    M[3] = -fma(l21, M[1], l20 * y0) * y2, M[4] = -(l21 * M[2]) * y2, M[5] = y2;
    M[6] = -fma(l32, M[3], fma(l31, M[1], l30 * y0)) * y3, M[7] = -fma(l32, M[4], l31 * M[2]) * y3, M[8] = -(l32 * M[5]) * y3, M[9] = y3;
    double mop = 0.0;
#pragma unroll
    for (int q = 0; q < 10; q++) mop = mi == q ? M[q] : mop;
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    const v4d V = mfma_f64(mop, A[cb], z), VE = mfma_f64(mop, E[cb], z);
    const double vv = V[0], ve = VE[0];
    A = mfma_f64(-vv, vv, A);
    E = mfma_f64(-vv, ve, E);
    keep[cb] = n >= 4 * cb + kq ? vv : ve;
    const double yn = (n & 3) == 0 ? y0 : (n & 3) == 1 ? y1 : (n & 3) == 2 ? y2 : y3;
    myinv = (n >> 2) == cb ? yn : myinv;
  }
  if (n < nvalid) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int c = kq + 4 * j;
      if (c < nvalid) D[n * ld + c] = keep[j];
    }
    if (kq == 0) ldinv_k[n] = myinv;
  }
}

// A_ik <- A_ik L_kk^-T (rows: valid rows of the tile)
template <class MP>
VIO_DEV void dtile_trsm(MP Aik, MP Dkk, cldsd ldinv_k, int ld, int rows, int lane) {
  const int n = lane & 15, kq = lane >> 4;
  double a[4];
  dtile_load_op(Aik, n * ld + kq, a);
  const v4d b = dtile_load_acc(Dkk, kq * ld + n, ld);  // Dkk[kq + 4 s][n] = Linv[n][kq + 4 s] for kq + 4 s < n
  const double dg = ldinv_k[n];
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s4 = 0; s4 < 4; s4++) {
    const int kk = 4 * s4 + kq;
    const double bb = kk < n ? b[s4] : (kk == n ? dg : 0.0);
    acc = mfma_f64(a[s4], bb, acc);
  }
  dtile_store_acc(Aik, kq * ld + n, ld, acc, kq, rows);
}
// y_k = L_kk^-1 b_k in place by 16 lanes of one wave
template <class MP>
VIO_DEV void dtile_forward_diag(MP Dkk, cldsd ldinv_k, ldsd bk, int ld, int nvalid, int lane) {
  const int c = lane & 15;
  double s = ldinv_k[c] * bk[c];
#pragma unroll
  for (int n = 0; n < 15; n++) {
    const double x = Dkk[n * ld + c], bn = bk[n];
    s = fma(n < c ? x : 0.0, n < c ? bn : 0.0, s);
  }
  __builtin_amdgcn_wave_barrier();  // every load of the wave precedes the stores (compiler-level ordering)
  if (lane < 16 && c < nvalid) bk[c] = s;
}
// b_i -= L_ik y_k: lane = 4 r + p, the four lanes of a quad split the 16 terms
template <class MP>
VIO_DEV void dtile_rhs_update(MP Lik, ldsd bi, cldsd yk, int ld, int rows, int lane) {
  const int r = lane >> 2, p = lane & 3;
  auto Lr = Lik + r * ld + p;
  double s = Lr[0] * yk[p];
  s = fma(Lr[4], yk[p + 4], s);
  s = fma(Lr[8], yk[p + 8], s);
  s = fma(Lr[12], yk[p + 12], s);
  s = quad_sum_f64(s);
  if (p == 0 && r < rows) bi[r] -= s;
}
// two independent tile updates C -= A B^T by one wave
template <class MP>
VIO_DEV void dtile_update2(MP C0, MP A0, MP B0, int rows0, int cols0, MP C1, MP A1, MP B1, int rows1, int cols1, int ld, int lane) {
  const int n = lane & 15, kq = lane >> 4, op = n * ld + kq, acc = kq * ld + n;
  double a0[4], b0[4], a1[4], b1[4];
  dtile_load_op(A0, op, a0), dtile_load_op(B0, op, b0), dtile_load_op(A1, op, a1), dtile_load_op(B1, op, b1);
  v4d c0 = dtile_load_acc(C0, acc, ld), c1 = dtile_load_acc(C1, acc, ld);
#pragma unroll
  for (int s4 = 0; s4 < 4; s4++) {
    c0 = mfma_f64(-a0[s4], b0[s4], c0);
    c1 = mfma_f64(-a1[s4], b1[s4], c1);
  }
  // (a partial last tile: with ld == pos its columns past the matrix ARE the next row's first entries)
  dtile_store_acc(C0, acc, ld, c0, kq, rows0, n < cols0);
  dtile_store_acc(C1, acc, ld, c1, kq, rows1, n < cols1);  // (rows1 = 0: no second tile)
}

// forward substitution pieces on the packed layout (tile rows past `rows` do not exist in memory)
template <class MP>
VIO_DEV void mtile_forward_diag(MP Dkk, cldsd ldinv_k, ldsd bk, int ld, int nvalid, int lane) {
  const int c = lane & 15;
  const bool cok = c < nvalid;
  double s = cok ? ldinv_k[c] * bk[c] : 0.0;
#pragma unroll
  for (int n = 0; n < 15; n++) {
    const bool in = cok && n < c;  // Linv[c][n] sits above the diagonal at Dkk[n][c]
    const double x = Dkk[(in ? n : 0) * ld + (in ? c : 0)], bn = bk[in ? n : 0];
    s = fma(in ? x : 0.0, bn, s);
  }
  __builtin_amdgcn_wave_barrier();  // every load of the wave precedes the stores (compiler-level ordering)
  if (lane < 16 && cok) bk[c] = s;
}
template <class MP>
VIO_DEV void mtile_rhs_update(MP Lik, ldsd bi, cldsd yk, int ld, int rows, int lane) {
  const int r = lane >> 2, p = lane & 3;
  const bool ok = r < rows;
  auto Lr = Lik + (ok ? r : 0) * ld + p;
  double s = Lr[0] * yk[p];
  s = fma(Lr[4], yk[p + 4], s);
  s = fma(Lr[8], yk[p + 8], s);
  s = fma(Lr[12], yk[p + 12], s);
  s = quad_sum_f64(s);
  if (p == 0 && ok) bi[r] -= s;
}

// In place: the lower triangle of Am becomes L (cut columns zero), m.bm becomes L^-1 b. m.tol holds the cut thresholds.
template <class MW>
VIO_DEV void marg_cholesky_tiles(const Ctx &cx, MW &m, int pos) {
  const int nt = (pos + 15) >> 4;
  const int tid_ = VIO_TID(cx), wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), nw = cx.nt >> 6, lane = tid_ & 63;
  const int li = lane & 15, kq = lane >> 4;
  auto tile = [&](int ti, int tj) { return m.Am + tri_off(ti) + 16 * tj; };
  auto rows_of = [&](int ti) { return pos - 16 * ti < 16 ? pos - 16 * ti : 16; };
  auto tol_of = [&](int ti) {
    const int j = 16 * ti + (lane & 15);
    const double t = m.tol[j];  // (padded: readable past pos)
    return j < pos ? t : 1.0;
  };
  if (wave == 0) potrf16_cut_wave(tile(0, 0), tile(0, 0), tri_ld(0), rows_of(0), false, tol_of(0), m.ldinv, lane);
  VIO_SYNC();
  for (int k = 0; k < nt; k++) {
    const int ntb = nt - k - 1;
    for (int bi = wave; bi < ntb; bi += nw)
      tile_trsm(tile(k + 1 + bi, k), tri_ld(k + 1 + bi), rows_of(k + 1 + bi), tile(k, k), tri_ld(k), m.ldinv + 16 * k, li, kq);
    if (wave == nw - 1) mtile_forward_diag(tile(k, k), m.ldinv + 16 * k, m.bm + 16 * k, tri_ld(k), rows_of(k), lane);
    VIO_SYNC();
    if (wave == 0) {
      if (ntb > 0) potrf16_cut_wave(tile(k + 1, k + 1), tile(k + 1, k), tri_ld(k + 1), rows_of(k + 1), true, tol_of(k + 1), m.ldinv + 16 * (k + 1), lane);
    } else {
      const int stride = nw - 1;
      for (int bi = wave - 1; bi < ntb; bi += stride)
        mtile_rhs_update(tile(k + 1 + bi, k), m.bm + 16 * (k + 1 + bi), m.bm + 16 * k, tri_ld(k + 1 + bi), rows_of(k + 1 + bi), lane);
      const int npairs = ntb * (ntb + 1) / 2;  // pair 0 = the look-ahead tile
      for (int pr = wave; pr < npairs; pr += stride) {
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= pr) a++;
        const int i0 = k + 1 + a, j0 = k + 1 + pr - a * (a + 1) / 2;
        tile_update(tile(i0, j0), tri_ld(i0), rows_of(i0), tile(i0, k), tile(j0, k), tri_ld(j0), rows_of(j0), li, kq);
      }
    }
    VIO_SYNC();
  }
}
#endif

template <class MW>
VIO_DEV void marginalize_window_impl(const Ctx &cx, const WinView &v, cldsd xpose, cldsd xsb, cldsd xfeat, cldsd ex,
                                     MW &m, const MargOut &out) {
  const int W = v.W, P = v.P, F = v.F;
  const int flag = v.marg_flag;
  const int pn = v.prior_n;
  // ---- which variant runs (uniform across the block) ---------------------------------------------------
  bool run = (flag == 0);
  if (flag == 1) {
    // MARGIN_SECOND_NEW only when the prior references para_Pose[W-1] (VINS.cpp:778-779)
    bool touches = false;
    for (int b = 0; b < v.prior_nb; b++)
      if (v.pr_kind[b] == 0 && v.pr_index[b] == W - 1) touches = true;
    run = pn > 0 && touches;
  }
  if (!run) {
    if (cx.tid == 0) out.n[0] = -1, out.n[1] = 0, out.n[2] = 0, out.n[3] = 0;
    return;
  }
  // ---- involved blocks and their dense columns: dropped first, then kept in (pose.., speed-bias.., extrinsic) order
  if (cx.tid == 0) {
    for (int i = 0; i <= P; i++) m.col_pose[i] = -1;
    for (int i = 0; i < P; i++) m.col_sb[i] = -1;
    m.col_ex[0] = -1;
    // mark involvement with -2
    for (int b = 0; b < v.prior_nb; b++) {
      int kind = v.pr_kind[b], idx = v.pr_index[b];
      if (kind == 0) m.col_pose[idx] = -2;
      else if (kind == 1) m.col_sb[idx] = -2;
      else m.col_ex[0] = -2;
    }
    if (flag == 0) m.col_pose[0] = m.col_sb[0] = m.col_pose[1] = m.col_sb[1] = -2;
  }
  VIO_SYNC();
  if (flag == 0) {
    // frames seen from frame 0: the (host 0, target t) buckets of the factor list (all writers store the same value)
    VIO_PARFOR(p, v.npairs)
      if (v.pair_h[p] == 0 && v.pair_t[p] != P) m.col_pose[v.pair_t[p]] = -2, m.col_ex[0] = -2;
  }
  VIO_SYNC();
  if (cx.tid == 0) {
    int pos = 0, nblocks = 0;
    // dropped
    if (flag == 0) {
      m.col_pose[0] = pos, pos += 6;
      m.col_sb[0] = pos, pos += 9;
    } else {
      m.col_pose[W - 1] = pos, pos += 6;
    }
    const int mdrop = pos;
    auto keep = [&](int kind, int idx, int col, int new_index) {
      out.kind[nblocks] = kind, out.index[nblocks] = new_index, out.offset[nblocks] = col - mdrop;
      double *x0 = out.x0 + 9 * nblocks;
      for (int q = 0; q < 9; q++) x0[q] = 0.0;
      cldsd src = kind == 0 ? xpose + 7 * idx : kind == 1 ? xsb + 9 * idx : ex;
      int gs = kind == 1 ? 9 : 7;
      for (int q = 0; q < gs; q++) x0[q] = src[q];
      nblocks++;
    };
    // addr_shift: MARGIN_OLD i -> i-1 (VINS.cpp:760-769); SECOND_NEW W -> W-1, others unchanged (:804-823)
    for (int i = 0; i < P; i++)
      if (m.col_pose[i] == -2) {
        m.col_pose[i] = pos;
        keep(0, i, pos, flag == 0 ? i - 1 : (i == W ? W - 1 : i));
        pos += 6;
      }
    for (int i = 0; i < P; i++)
      if (m.col_sb[i] == -2) {
        m.col_sb[i] = pos;
        keep(1, i, pos, flag == 0 ? i - 1 : (i == W ? W - 1 : i));
        pos += 9;
      }
    if (m.col_ex[0] == -2) {
      m.col_ex[0] = pos;
      keep(2, 0, pos, 0);
      pos += 6;
    }
    m.meta[0] = pos, m.meta[1] = mdrop, m.meta[2] = pos - mdrop, m.meta[3] = nblocks;
  }
  VIO_SYNC();
  const int pos = m.meta[0], mdrop = m.meta[1], n = m.meta[2], nblocks = m.meta[3];
  const int ld = m.ld;
  if (pos > ld || n > out.ncap) {  // more kept speed-bias blocks than any reference-made prior carries
    if (cx.tid == 0) out.n[0] = -2, out.n[1] = 0, out.n[2] = mdrop, out.n[3] = pos;
    return;
  }
  VIO_PARFOR(q, (int)tri_doubles(pos)) m.Am[q] = 0.0;
  VIO_PARFOR(q, pos) m.bm[q] = 0.0;
  VIO_PARFOR(f, F) m.hff[f] = 0.0, m.gf[f] = 0.0;
  // The landmark coupling of this phase lives in the solver's feature-major W (v.WTf [F][n6cap], column groups: pose i at
  // 6 i, the extrinsic at 6 P where the solver keeps the relocalization pose): every factor of a landmark hosted at frame 0
  // is re-evaluated below and overwrites its (landmark, target frame) entries, the host and extrinsic groups accumulate and
  // are zeroed here; rows of landmarks hosted elsewhere keep the solver's (finite) values and drop out through 1 / E_f = 0.
  if (flag == 0) VIO_PARFOR(q, 12 * F) {
    const int f = q / 12, c = q - 12 * f;
    v.WTf[(size_t)f * v.n6cap + (c < 6 ? c : 6 * P + c - 6)] = 0.0;
  }
  VIO_SYNC();
  // ---- prior as a factor (MarginalizationFactor evaluated at the current state) ------------------------
  if (pn > 0) {
    VIO_PARFOR(b, v.prior_nb) {
      int kind = v.pr_kind[b], idx = v.pr_index[b], o = v.pr_offset[b];
      const double *x0 = v.pr_x0 + 9 * b;
      int base = kind == 0 ? m.col_pose[idx] : kind == 1 ? m.col_sb[idx] : m.col_ex[0];
      int ls = kind == 1 ? 9 : 6;
      for (int k = 0; k < ls; k++) m.pcol[o + k] = base + k;
      if (kind == 0) prior_block_dx(7, xpose + 7 * idx, x0, m.prdx + o);
      else if (kind == 1) prior_block_dx(9, xsb + 9 * idx, x0, m.prdx + o);
      else prior_block_dx(7, ex, x0, m.prdx + o);
    }
    // J^T r = b0 + H0 dx and J^T J = H0 (setup_prior in solver_core.h keeps both for the whole launch)
    VIO_PARFOR(i, pn) m.prr[i] = v.prb0[i];
    VIO_SYNC();
    dense_matvec_cols(cx, v.prH0, pn, m.prdx, [&](int i, double sacc) { VIO_ATOMIC_ADD(m.prr + i, sacc); });
    {
      // H0 into the dense matrix through the column map: (row, strip of columns) items, every load of an item in flight
      // before its first store (a rolled element loop pays an L2 round trip per element)
      constexpr int kU = 12;
      const int nst = (pn + kU - 1) / kU;
      VIO_PARFOR(q, pn * nst) {
        const int a = q / nst, b0 = kU * (q - a * nst);
        const int ca = m.pcol[a];
        double x[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) x[u] = v.prH0[a * pn + (b0 + u < pn ? b0 + u : b0)];
        VIO_SCHED_FENCE();
#pragma unroll
        for (int u = 0; u < kU; u++) {
          if (b0 + u >= pn) continue;
          const int cb = m.pcol[b0 + u];
          if (ca >= cb) m.Am[tri_at(ca, cb)] = x[u];
        }
      }
    }
    VIO_SYNC();
    // (bm was zeroed above and every prior column owns its dense column)
    VIO_PARFOR(a, pn) m.bm[m.pcol[a]] += m.prr[a];
    VIO_SYNC();
  }
  stamp(cx, ST_M_PRIOR);
  if (flag == 0) {
    // ---- IMUFactor(pre_integrations[1]) on (pose0, sb0, pose1, sb1) ------------------------------------
    if (cx.tid == 0)
      imu_eval_raw(v.gravity, v.preint, xpose, xsb, xpose + 7, xsb + 9, v.imu_r, v.imu_J);
    VIO_SYNC();
    VIO_PARFOR(r, 15) {
      const double *info = v.imu_info + r * 15;
      double s = 0;
      for (int k = 0; k < 15; k++) s += info[k] * v.imu_r[k];
      v.imu_Mr[r] = s;
    }
    VIO_PARFOR(q, 450) {
      int r = q / 30, c = q % 30;
      const double *info = v.imu_info + r * 15;
      double s = 0;
      for (int k = 0; k < 15; k++) s += info[k] * v.imu_J[k * 30 + c];
      v.imu_M[q] = s;
    }
    VIO_SYNC();
    // local column a in [0,30): pose0 | sb0 | pose1 | sb1
    VIO_PARFOR(q, 900) {
      int a = q / 30, b = q % 30;
      int ca = a < 6 ? m.col_pose[0] + a : a < 15 ? m.col_sb[0] + a - 6 : a < 21 ? m.col_pose[1] + a - 15 : m.col_sb[1] + a - 21;
      int cb = b < 6 ? m.col_pose[0] + b : b < 15 ? m.col_sb[0] + b - 6 : b < 21 ? m.col_pose[1] + b - 15 : m.col_sb[1] + b - 21;
      double s = 0;
      for (int k = 0; k < 15; k++) s += v.imu_J[k * 30 + a] * v.imu_M[k * 30 + b];
      if (ca >= cb) VIO_ATOMIC_ADD(m.Am + tri_at(ca, cb), s);
    }
    VIO_PARFOR(a, 30) {
      int ca = a < 6 ? m.col_pose[0] + a : a < 15 ? m.col_sb[0] + a - 6 : a < 21 ? m.col_pose[1] + a - 15 : m.col_sb[1] + a - 21;
      double s = 0;
      for (int k = 0; k < 15; k++) s += v.imu_J[k * 30 + a] * v.imu_Mr[k];
      VIO_ATOMIC_ADD(m.bm + ca, s);
    }
    stamp(cx, ST_M_IMU);
    // ---- projections hosted at frame 0: blocks (pose0, pose_t, extrinsic, feature), Cauchy-corrected
    //      (ResidualBlockInfo::Evaluate, marginalization_factor.cpp:45-76, rho'' < 0 branch).
    //      Same scheme as the solver: rows staged in (0,t)-bucket order, one Gram product per bucket on the matrix
    //      cores: [Ji Jj r Jl]^T [Ji Jj r Jl], Jex^T [Ji Jj r Jl] and Jex^T Jex.
    const double cc = 1.0 / v.cauchy_b;
    int S0 = 0, nb0 = 0;  // buckets (0, t < P) come first in the (host, target) order
    for (int p = 0; p < v.npairs && v.pair_h[p] == 0; p++)
      if (v.pair_t[p] != P) S0 = (v.pair_s1[p] + 1) & ~1, nb0 = p + 1;
    auto G = m.stage;
    const int CH = m.stage_slots;
    if (CH < 2) {  // launcher guarantees staging space; never loop forever on a bad carve
      if (cx.tid == 0) out.n[0] = -3, out.n[1] = 0;
      return;
    }
    // dense-matrix add, lower triangle only
    auto add_lower = [&](int ra, int ca, double val) {
      if (ra >= ca) VIO_ATOMIC_ADD(m.Am + tri_at(ra, ca), val);
      else VIO_ATOMIC_ADD(m.Am + tri_at(ca, ra), val);
    };
    // bucket descriptors (0, t): one per lane, fetched once (a dependent global round trip per Gram round otherwise)
    const int lane_d = VIO_TID(cx) & 63;
    const bool dv = lane_d < nb0;
    const int d_s0 = dv ? v.pair_s0[lane_d] : 0, d_s1 = dv ? v.pair_s1[lane_d] : 0;
    const int d_t = dv ? ((v.pair_h[lane_d] != 0 || v.pair_t[lane_d] == P) ? -1 : v.pair_t[lane_d]) : -1;
    for (int c0 = 0; c0 < S0; c0 += CH) {
      // The factors hosted at frame 0 occupy the staging slots [0, S0) of the solver's slot order: the slot records built
      // at the start of the solve (srec_i / srec_d) give host | target | landmark and the observation pair in ONE global
      // round trip per pass (factor index -> target / landmark / points were three dependent ones). The per-landmark sums
      // (H_ff, g_f in LDS; the host and extrinsic coupling in the window's W) are gathered by the factor threads
      // themselves with atomics, like in the solver: no second pass over the landmarks' factor lists.
      VIO_PARFOR(slot, (S0 - c0 < CH ? S0 - c0 : CH)) {
        const int rec = v.srec_i[c0 + slot];
        double pij[6];
#pragma unroll
        for (int c = 0; c < 6; c++) pij[c] = v.srec_d[6 * (size_t)(c0 + slot) + c];
        if (rec < 0) continue;
        const int t = (rec >> 8) & 255, f = rec >> 16;
        if ((rec & 255) != 0 || t == P) continue;
        double r[2], Ji[12], Jj[12], Jex[12], Jl[2];
        projection_eval(v.s_info, xpose, xpose + 7 * t, ex, xfeat[f], pij, pij + 3, true, r, Ji, Jj, Jex, Jl);
        double sq = r[0] * r[0] + r[1] * r[1];
        double sr = sqrt(1.0 / (1.0 + sq * cc));
        auto g = G + slot * kMargSlot; auto gx = g + kSlotStride;
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
#pragma unroll
          for (int c = 0; c < 6; c++) {
            g[rr * kRowLen + c] = Ji[rr * 6 + c] * sr, g[rr * kRowLen + 6 + c] = Jj[rr * 6 + c] * sr;
            gx[rr * kMargRowX + c] = Jex[rr * 6 + c] * sr;
          }
          g[rr * kRowLen + 12] = r[rr] * sr, g[rr * kRowLen + 13] = Jl[rr] * sr;
        }
        const double s2 = sr * sr;
        double *wf = v.WTf + (size_t)f * v.n6cap;
#pragma unroll
        for (int c = 0; c < 6; c++) {
          wf[6 * t + c] = (Jj[c] * Jl[0] + Jj[6 + c] * Jl[1]) * s2;  // target-frame coupling: one writer per (feature, frame)
          VIO_ATOMIC_ADD(wf + c, (Ji[c] * Jl[0] + Ji[6 + c] * Jl[1]) * s2);
          VIO_ATOMIC_ADD(wf + 6 * P + c, (Jex[c] * Jl[0] + Jex[6 + c] * Jl[1]) * s2);
        }
        VIO_ATOMIC_ADD(m.hff + f, (Jl[0] * Jl[0] + Jl[1] * Jl[1]) * s2);
        VIO_ATOMIC_ADD(m.gf + f, (Jl[0] * r[0] + Jl[1] * r[1]) * s2);
      }
      VIO_SYNC();
      stamp(cx, ST_M_FACT);
      // one element of the three Gram matrices of bucket (0, t)
      auto flush1 = [&](int t, int row, int col, double val) {  // G1^T G1
        const int c0p = m.col_pose[0], ctp = m.col_pose[t];
        if (row < 6) {
          if (col <= row) VIO_ATOMIC_ADD(m.Am + tri_at(c0p + row, c0p + col), val);
        } else if (row < 12) {
          if (col < 6) add_lower(ctp + row - 6, c0p + col, val);
          else if (col < 12 && col <= row) VIO_ATOMIC_ADD(m.Am + tri_at(ctp + row - 6, ctp + col - 6), val);
        } else if (row == 12) {
          if (col < 6) VIO_ATOMIC_ADD(m.bm + c0p + col, val);
          else if (col < 12) VIO_ATOMIC_ADD(m.bm + ctp + col - 6, val);
        }
      };
      auto flush2 = [&](int t, int row, int col, double val) {  // Gx^T G1: rows = extrinsic components
        if (row >= 6) return;
        const int cx0 = m.col_ex[0];
        if (col < 6) add_lower(cx0 + row, m.col_pose[0] + col, val);
        else if (col < 12) add_lower(cx0 + row, m.col_pose[t] + col - 6, val);
        else if (col == 12) VIO_ATOMIC_ADD(m.bm + cx0 + row, val);
      };
      auto flush3 = [&](int row, int col, double val) {  // Gx^T Gx
        if (row < 6 && col <= row) VIO_ATOMIC_ADD(m.Am + tri_at(m.col_ex[0] + row, m.col_ex[0] + col), val);
      };
      {
        const int tid_ = VIO_TID(cx), wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), nw = cx.nt >> 6, lane = tid_ & 63;
        const int li = lane & 15, kq = lane >> 4;
        for (int p = wave; p < nb0 && p < 64; p += nw) {
          const int b_s0 = __builtin_amdgcn_readlane(d_s0, p), b_s1 = __builtin_amdgcn_readlane(d_s1, p);
          const int t = __builtin_amdgcn_readlane(d_t, p);
          if (t < 0) continue;
          const int s_lo = b_s0 > c0 ? b_s0 : c0, s_hi = b_s1 < c0 + CH ? b_s1 : c0 + CH;
          if (s_lo >= s_hi) continue;
          v4d a1 = {0.0, 0.0, 0.0, 0.0}, a2 = a1, a3 = a1;
          const bool lv = li < kRowLen, lx = li < kMargRowX;
          const int go = (s_lo - c0 + (kq >> 1)) * kMargSlot + (kq & 1) * kRowLen + (lv ? li : 0);
          const int gxo = (s_lo - c0 + (kq >> 1)) * kMargSlot + kSlotStride + (kq & 1) * kMargRowX + (lx ? li : 0);
          const int nsteps = (s_hi - s_lo + 1) >> 1;
          constexpr int kB = 6;  // two-factor steps whose operands are fetched together
          for (int st0 = 0; st0 < nsteps; st0 += kB) {
            double av[kB], xv[kB];
#pragma unroll
            for (int j = 0; j < kB; j++) {
              const bool in = st0 + j < nsteps && s_lo + 2 * (st0 + j) + (kq >> 1) < s_hi;  // odd tail: no second factor
              av[j] = G[in ? go + 2 * (st0 + j) * kMargSlot : 0], xv[j] = G[in ? gxo + 2 * (st0 + j) * kMargSlot : 0];
            }
            VIO_SCHED_FENCE();
#pragma unroll
            for (int j = 0; j < kB; j++) {
              const bool in = st0 + j < nsteps && s_lo + 2 * (st0 + j) + (kq >> 1) < s_hi;
              const double a = (lv && in) ? av[j] : 0.0, x = (lx && in) ? xv[j] : 0.0;
              if (st0 + j < nsteps) a1 = mfma_f64(a, a, a1), a2 = mfma_f64(x, a, a2), a3 = mfma_f64(x, x, a3);  // (uniform)
            }
          }
#pragma unroll
          for (int r4 = 0; r4 < 4; r4++) {
            flush1(t, kq + 4 * r4, li, a1[r4]), flush2(t, kq + 4 * r4, li, a2[r4]), flush3(kq + 4 * r4, li, a3[r4]);
          }
        }
      }
      VIO_SYNC();
      stamp(cx, ST_M_GRAM);
    }
    // ---- eliminate the landmarks hosted at frame 0 (pseudo-inverse: e <= eps contributes nothing) --------
    VIO_PARFOR(f, F) {
      const double ei = m.hff[f] > 1e-8 ? 1.0 / m.hff[f] : 0.0;
      m.einv[f] = ei, m.gf[f] *= ei;  // g_f / E_f
    }
    VIO_SYNC();
    // pose-type groups: g in [0, P] -> (W column base 6 g, dense column base). (W E^-1) W^T over that index space on the
    // matrix cores, the solver's K-split product (solver_core.h); its tiles scatter into the dense matrix by group.
    const int ng = P + 1, n6m = 6 * ng;
    auto dense_col = [&](int a) {
      const int ga = a / 6, ca = ga == P ? m.col_ex[0] : m.col_pose[ga];
      return ca < 0 ? -1 : ca + a - 6 * ga;
    };
    if (n6m <= 80) {
      schur_ksplit5(cx, v.WTf, v.n6cap, n6m, F, m.einv, m.gf,
                    [&](int arow, int bcol, double val) {
                      const int rr = dense_col(arow), cc2 = dense_col(bcol);
                      if (rr >= 0 && cc2 >= 0) add_lower(rr, cc2, -val);
                    },
                    [&](int a, double val) {
                      const int rr = dense_col(a);
                      if (rr >= 0) VIO_ATOMIC_ADD(m.bm + rr, -val);
                    });
    } else {
      // (the solver's blocked K-split product, solver_core.h schur_blocks; every element through the atomic add_lower)
      schur_blocks(
          cx, v.WTf, v.n6cap, n6m, F, m.einv, m.gf, 0, 1,
          [&](int arow, int bcol, double val, bool) {
            const int rr = dense_col(arow), cc2 = dense_col(bcol);
            if (rr >= 0 && cc2 >= 0) add_lower(rr, cc2, -val);
          },
          [&](int a, double val) {  // b -= W (g_f / E_f)
            const int rr = dense_col(a);
            if (rr >= 0) VIO_ATOMIC_ADD(m.bm + rr, -val);
          });
    }
    VIO_SYNC();
  }
  stamp(cx, ST_MARG_BUILD);
  // ---- Cholesky with pivot cut, b carried along (forward substitution) --------------------------------
  VIO_PARFOR(j, pos) m.tol[j] = fmax(1e-8, 1e-12 * m.Am[tri_at(j, j)]);
  VIO_SYNC();
  // Right-looking, ONE barrier per column: column j stays unscaled in place while the trailing update uses
  // A_ij A_kj / piv; the scaling L_ij = A_ij / sqrt(piv) happens once at the end. A cut pivot zeroes its column.
  // The lower-triangle entries (i, k) are listed once, columns from last to first, in the staging area the build phase
  // no longer needs: the entries step j touches (k > j) are a PREFIX of that list, so every lane has an element and no
  // index arithmetic beyond one table read.
  marg_cholesky_tiles(cx, m, pos);
  // ---- outputs: J0 = L'^T (upper triangular), r0 = y' ---------------------------------------------------
  VIO_PARFOR(q, n * n) {
    int r = q / n, c = q % n;
    out.J[q] = c >= r ? m.Am[tri_at(mdrop + c, mdrop + r)] : 0.0;
  }
  VIO_PARFOR(i, n) out.r[i] = m.bm[mdrop + i];
  if (cx.tid == 0) out.n[0] = n, out.n[1] = nblocks, out.n[2] = mdrop, out.n[3] = pos;
  VIO_SYNC();
  stamp(cx, ST_MARG_CHOL);
}

// data = false: header only (the data stays in the device-resident store; mo.x0 / J / r are not read).
inline void unpack_prior(const MargOut &mo, VioPrior &p, bool data = true) {
  p.n = mo.n[0];
  p.n_blocks = mo.n[1];
  if (p.n <= 0) {
    p.n_blocks = 0;
    return;
  }
  for (int b = 0; b < p.n_blocks; b++)
    p.block_kind[b] = mo.kind[b], p.block_index[b] = mo.index[b], p.block_offset[b] = mo.offset[b];
  if (!data) return;
  memcpy(p.block_x0, mo.x0, sizeof(double) * 9 * p.n_blocks);
  memcpy(p.linearized_jacobians, mo.J, sizeof(double) * p.n * p.n);
  memcpy(p.linearized_residuals, mo.r, sizeof(double) * p.n);
}

}  // namespace vio
