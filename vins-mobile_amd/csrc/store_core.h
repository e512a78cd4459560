// store_core.h — the landmark store of a sequence on the device: FeatureManager's list (feature_manager.cpp:11-407) as flat
// arrays in HBM, one workgroup per sequence, and the window assembly of solve_ceres (VINS.cpp:528-567) written straight
// into the solver's batch arrays (batch.h) instead of being packed on the host and uploaded.
//
// Three device passes per published frame and sequence, all of them list work (integer / byte traffic, a few hundred
// landmarks): nothing here is shaped for the matrix cores.
//   store_ingest   addFeatureCheckParallax (:103-155) -> keyframe decision, triangulate (:189-248), landmark / factor counts
//   store_pack     para_Feature + the factor list + the (host, target) buckets of pack_window (batch.h) for window b
//   store_finish   setDepth (:300-313), failureDetection (VINS.cpp:214-265), removeBackShiftDepth / removeFront
//                  (:250-257, :343-372), removeFailures (:259-268), compacted into the other bank of the store
//
// The host-side list (vio_window.cpp) stays the restatement the estimator uses while a sequence initializes; every
// function here performs the same floating-point operations in the same order (this file is compiled with
// -ffp-contract=off like vio_window.cpp), so a sequence gives the same bits on either path. tests/emul runs these
// functions on the host through the SIMT emulator against vio_features_* on random frame streams.
#pragma once

#include "solver_core.h"
#include "vio_amd.h"

namespace vio {
namespace store {

constexpr double kMinParallax = 10.0 / 549;  // MIN_PARALLAX (feature_manager.hpp:23)
constexpr double kInitDepth = 5.0;           // INIT_DEPTH (feature_manager.hpp:24)
constexpr int kMaxP = 32;                    // observation columns a landmark can have (window_size + 1 <= kMaxP)
constexpr int kThreads = 256;
constexpr int kTri = 32;                     // landmarks triangulated side by side (store_ingest)

// control block of a sequence (ints)
enum {
  C_N = 0,       // list length
  C_BANK,        // bank the list lives in
  C_STATUS,      // VIO_OK or the error that stopped this frame (the host restarts the sequence)
  C_MARG,        // marginalization flag of this frame
  C_TRACK,       // last_track_num
  C_PNUM,        // parallax_num
  C_F,           // landmarks of the window (para_Feature rows)
  C_M,           // projection factors
  C_FAIL,        // failureDetection reasons of this frame's solve
  C_NPAIRS, C_NSLOTS,
  C_NLOOP,       // relocalization factors of this window (VINS.cpp:597-631)
  C_COUNT = 16
};
constexpr int kCtlDoubles = 12;  // last_P[3], last_R[9]

struct Dims {
  int W, Lcap, Ocap;  // window size; list capacity; observations per frame
};

// One bank of one sequence.
struct Bank {
  int *fid, *start, *nobs, *flag;  // [Lcap] feature_id, start_frame, feature_per_frame.size(), solve_flag
  double *depth;                   // [Lcap] estimated_depth
  double *obs;                     // [Lcap][P][3] FeaturePerFrame::point
};

struct Lds {
  ldsi ids;    // [Ocap] observation ids, ascending
  ldsi perm;   // [Ocap] sorted position -> index into the frame's observations
  ldsi aux;    // [Ocap + 1]
  ldsi scanA;  // [Lcap + 1]
  ldsi scanB;  // [Lcap + 1]
  ldsi part;   // [threads + 1 + threads / 16]
  ldsd term;   // [Lcap]
  ldsd tri;    // [8 (W + 1)][kTri] row matrices of the landmarks being triangulated
  ldsd cam;    // [12 (W + 1) + 12] Rs | Ps | ric | tic for the triangulation
  ldsi misc;   // [8]
  long long *prof;  // null, or [32] cycle stamps of one workgroup's passes (VIO_AMD_STORE_PROF)
};
VIO_HD size_t lds_bytes(const Dims &d) {
  return sizeof(int) * (3 * (size_t)d.Ocap + 1 + 2 * ((size_t)d.Lcap + 1) + kThreads + 1 + kThreads / 16 + 8 + 8) + sizeof(double) * ((size_t)d.Lcap + 8 * ((size_t)d.W + 1) * kTri + 12 * ((size_t)d.W + 1) + 12) + 64;
}
template <class PI, class PD>
VIO_DEV Lds carve_lds(const Dims &d, PI ibase, PD *dbase_out) {
  // ints first, then the doubles on an 8-byte boundary
  Lds l;
  PI p = ibase;
  l.ids = p, p += d.Ocap;
  l.perm = p, p += d.Ocap;
  l.aux = p, p += d.Ocap + 1;
  l.scanA = p, p += d.Lcap + 1;
  l.scanB = p, p += d.Lcap + 1;
  l.part = p, p += kThreads + 1 + kThreads / 16;
  l.misc = p, p += 8;
  size_t ints = (size_t)(p - ibase);
  ints = (ints + 1) & ~(size_t)1;
  *dbase_out = (PD)(ibase + ints);
  l.term = *dbase_out;
  l.tri = l.term + d.Lcap;
  l.cam = l.tri + 8 * (d.W + 1) * kTri;
  l.prof = nullptr;
  return l;
}

struct Cx {
  int tid, nt;
  int wave64 = 0;  // (VIO_TID of solver_core.h)
};

#ifdef VIO_HOST_BUILD
#define STORE_STAMP(l, k) ((void)0)
#else
#define STORE_STAMP(l, k)                                   \
  do {                                                      \
    if ((l).prof && cx.tid == 0) (l).prof[k] = clock64();   \
  } while (0)
#endif

// out[i] = sum of val(k) for k < i, out[n] = the total (returned). Every work-item calls it.
template <class F>
VIO_DEV int block_scan(const Cx &cx, int n, ldsi out, ldsi part, F val) {
  const int t = VIO_TID(cx);
  const int per = (n + cx.nt - 1) / cx.nt;
  const int b = t * per < n ? t * per : n, e = b + per < n ? b + per : n;
  int s = 0;
  for (int i = b; i < e; i++) {
    const int v = val(i);
    out[i] = s;
    s += v;
  }
  part[t] = s;
  VIO_SYNC();
  // the work-items' partial sums: runs of 16 scanned by 16 work-items, then their totals by one (two short serial chains
  // instead of one of nt links: an LDS round trip per link)
  const int runs = (cx.nt + 15) >> 4;
  if (t < runs) {
    int acc = 0;
    const int k1 = (t << 4) + 16 < cx.nt ? (t << 4) + 16 : cx.nt;
    for (int k = t << 4; k < k1; k++) {
      const int v = part[k];
      part[k] = acc;
      acc += v;
    }
    part[cx.nt + 1 + t] = acc;
  }
  VIO_SYNC();
  if (t == 0) {
    int acc = 0;
    for (int k = 0; k < runs; k++) {
      const int v = part[cx.nt + 1 + k];
      part[cx.nt + 1 + k] = acc;
      acc += v;
    }
    part[cx.nt] = acc;
  }
  VIO_SYNC();
  const int off = part[t] + part[cx.nt + 1 + (t >> 4)];
  for (int i = b; i < e; i++) out[i] += off;
  if (t == 0) out[n] = part[cx.nt];
  VIO_SYNC();
  return out[n];
}

// out[i] = max of val(k) for k < i (0 for i = 0), values >= 0. Every work-item calls it. Same shape as block_scan.
template <class F>
VIO_DEV void block_prefix_max(const Cx &cx, int n, ldsi out, ldsi part, F val) {
  const int t = VIO_TID(cx);
  const int per = (n + cx.nt - 1) / cx.nt;
  const int b = t * per < n ? t * per : n, e = b + per < n ? b + per : n;
  int m = 0;
  for (int i = b; i < e; i++) {
    const int v = val(i);
    out[i] = m;
    m = v > m ? v : m;
  }
  part[t] = m;
  VIO_SYNC();
  if (t == 0) {
    int acc = 0;
    for (int k = 0; k < cx.nt; k++) {
      const int v = part[k];
      part[k] = acc;
      acc = v > acc ? v : acc;
    }
  }
  VIO_SYNC();
  const int off = part[t];
  for (int i = b; i < e; i++) out[i] = out[i] > off ? out[i] : off;
  VIO_SYNC();
}

VIO_DEV bool solved_in_window(int nobs, int start, int W) { return nobs >= 2 && start < W - 2; }

// Right singular vector of the smallest singular value of A (rows x 4): the one-sided Jacobi of vio_window.cpp, same
// operations in the same order.
struct TriRows {         // element k of a work-item's row matrix: LDS, kTri apart
  ldsd p;
  VIO_DEV VIO_AS3 double &operator()(int k) const { return p[k * kTri]; }
};

template <class Rows>
VIO_DEV void smallest_right_singular_vector(const Rows &A, int rows, double v_out[4]) {
  double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < rows; i++) {
          const double ap = A(i * 4 + p), aq = A(i * 4 + q);
          alpha += ap * ap, beta += aq * aq, gamma += ap * aq;
        }
        if (gamma == 0.0) continue;
        const double ratio = fabs(gamma) / sqrt(alpha * beta + 1e-300);
        off = off < ratio ? ratio : off;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < rows; i++) {
          const double ap = A(i * 4 + p), aq = A(i * 4 + q);
          A(i * 4 + p) = c * ap - s * aq, A(i * 4 + q) = s * ap + c * aq;
        }
        for (int i = 0; i < 4; i++) {
          const double vp = V[i * 4 + p], vq = V[i * 4 + q];
          V[i * 4 + p] = c * vp - s * vq, V[i * 4 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  int best = 0;
  double bn = 1e300;
  for (int j = 0; j < 4; j++) {
    double nrm = 0;
    for (int i = 0; i < rows; i++) nrm += A(i * 4 + j) * A(i * 4 + j);
    if (nrm < bn) bn = nrm, best = j;
  }
  for (int i = 0; i < 4; i++) v_out[i] = V[i * 4 + best];
}

// ---- pass 1 -----------------------------------------------------------------------------------------------------------
// obs: this frame's image_msg (any order, unique ids). Ps [P][3], Rs [P][9]: the window states triangulate reads (the host's
// Ps / Rs after processIMU). The frame index of the new observations is W: a resident sequence has a full window.
VIO_DEV void store_ingest(const Cx &cx, const Dims &d, const Bank &bk, int *ctl, const Lds &l, const VioObs *obs, int n_obs,
                          const double *Ps, const double *Rs, const double *tic, const double *ric) {
  const int W = d.W, P = W + 1, fc = W;
  const int t = VIO_TID(cx);
  STORE_STAMP(l, 0);
  if (t == 0) l.misc[0] = VIO_OK, l.misc[1] = 0, l.misc[2] = 0, l.misc[3] = 0;
  VIO_SYNC();
  int n = ctl[C_N];
  if (n_obs < 0 || n_obs > d.Ocap) {
    if (t == 0) ctl[C_STATUS] = n_obs < 0 ? VIO_EINVAL : VIO_ECAP;
    return;
  }
  // image_msg is a std::map: ascending ids (a stable rank sort; equal ids are an argument error). The ids pass through LDS
  // first: the rank loop reads every id once per observation, from global memory that is a cache line per id.
  VIO_PARFOR(j, n_obs) l.scanA[j] = obs[j].id;
  VIO_SYNC();
  VIO_PARFOR(j, n_obs) {
    const int id = l.scanA[j];
    int r = 0;
    for (int k = 0; k < n_obs; k++) {
      const int ik = l.scanA[k];
      r += (ik < id) || (ik == id && k < j);
    }
    l.ids[r] = id, l.perm[r] = j, l.aux[r] = 0;
  }
  VIO_SYNC();
  VIO_PARFOR(j, n_obs)
    if (j > 0 && l.ids[j] == l.ids[j - 1]) l.misc[0] = VIO_EINVAL;
  VIO_SYNC();
  if (l.misc[0] != VIO_OK) {
    if (t == 0) ctl[C_STATUS] = l.misc[0];
    return;
  }
  STORE_STAMP(l, 1);
  // known landmarks take their observation
  VIO_PARFOR(i, n) {
    const int id = bk.fid[i];
    int lo = 0, hi = n_obs;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (l.ids[mid] < id) lo = mid + 1;
      else hi = mid;
    }
    if (lo < n_obs && l.ids[lo] == id) {
      const int no = bk.nobs[i];
      if (bk.start[i] + no > fc) {
        l.misc[0] = VIO_ESTATE;  // a second message for a frame the landmark already has
      } else {
        const VioObs &o = obs[l.perm[lo]];
        double *pt = bk.obs + ((size_t)i * P + no) * 3;
        pt[0] = o.x / o.z, pt[1] = o.y / o.z, pt[2] = o.z / o.z;
        bk.nobs[i] = no + 1;
        l.aux[lo] = 1;
      }
    }
  }
  VIO_SYNC();
  if (l.misc[0] != VIO_OK) {
    if (t == 0) ctl[C_STATUS] = l.misc[0];
    return;
  }
  STORE_STAMP(l, 2);
  // new landmarks, in ascending id
  const int n_new = block_scan(cx, n_obs, l.scanA, l.part, [&](int j) { return l.aux[j] ? 0 : 1; });
  const int track = n_obs - n_new;
  if (n + n_new > d.Lcap) {
    if (t == 0) ctl[C_STATUS] = VIO_ECAP;
    return;
  }
  VIO_PARFOR(j, n_obs) {
    if (l.aux[j]) continue;
    const int i = n + l.scanA[j];
    const VioObs &o = obs[l.perm[j]];
    bk.fid[i] = o.id, bk.start[i] = fc, bk.nobs[i] = 1, bk.flag[i] = 0, bk.depth[i] = -1.0;
    double *pt = bk.obs + (size_t)i * P * 3;
    pt[0] = o.x / o.z, pt[1] = o.y / o.z, pt[2] = o.z / o.z;
  }
  const int n_old = n;
  n += n_new;
  VIO_SYNC();
  STORE_STAMP(l, 3);
  // compensatedParallax2 over the landmarks seen in the two frames before this one, summed in list order
  int enough = 1, pnum = 0;
  if (!(fc < 2 || track < 20)) {
    VIO_PARFOR(i, n_old) {
      const int s = bk.start[i], no = bk.nobs[i];
      int has = 0;
      double term = 0;
      if (s <= fc - 2 && s + no - 1 >= fc - 1) {
        const double *fi = bk.obs + ((size_t)i * P + (fc - 2 - s)) * 3, *fj = bk.obs + ((size_t)i * P + (fc - 1 - s)) * 3;
        const double u_j = fj[0], v_j = fj[1];
        const double dep_i = fi[2];
        const double u_i = fi[0] / dep_i, v_i = fi[1] / dep_i;
        const double du = u_i - u_j, dv = v_i - v_j;
        const double du_comp = u_i - u_j, dv_comp = v_i - v_j;
        const double a = du * du + dv * dv, b = du_comp * du_comp + dv_comp * dv_comp;
        const double r = sqrt(b < a ? b : a);
        term = 0.0 < r ? r : 0.0;
        has = 1;
      }
      l.term[i] = term;
      if (has) __hip_atomic_fetch_add(&l.misc[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    VIO_SYNC();
    if (t == 0) {
      const int cnt = l.misc[3];
      // in list order, like the host loop; the landmarks that do not take part contribute +0.0, which leaves the sum's bits
      double sum = 0;
      for (int i = 0; i < n_old; i++) sum += l.term[i];
      l.misc[1] = cnt;
      l.misc[2] = cnt == 0 ? 1 : (sum / cnt >= kMinParallax ? 1 : 0);
    }
    VIO_SYNC();
    pnum = l.misc[1], enough = l.misc[2];
  }
  STORE_STAMP(l, 4);
  // triangulate: landmarks of the window that have no depth yet. They are few (the ones that entered the window's solved
  // range with this frame): listed first, then kTri at a time, one work-item each, the row matrix of the SVD in LDS
  // (element-major across the work-items: conflict-free) instead of a per-thread array in scratch memory.
  const int n_tri = block_scan(cx, n, l.scanA, l.part, [&](int i) {
    return solved_in_window(bk.nobs[i], bk.start[i], W) && !(bk.depth[i] > 0) ? 1 : 0;
  });
  VIO_PARFOR(i, n)
    if (l.scanA[i + 1] != l.scanA[i]) l.scanB[l.scanA[i]] = i;
  VIO_SYNC();
  VIO_PARFOR(k, 12 * P + 12) {
    const int nr = 9 * P, np = 3 * P;
    l.cam[k] = k < nr ? Rs[k] : k < nr + np ? Ps[k - nr] : k < nr + np + 9 ? ric[k - nr - np] : tic[k - nr - np - 9];
  }
  VIO_SYNC();
  for (int base = 0; base < n_tri; base += kTri) {
    if (t < kTri && base + t < n_tri) {
      const int i = l.scanB[base + t];
      const int no = bk.nobs[i], s = bk.start[i];
      if (s + no - 1 > W) {
        l.misc[0] = VIO_ESTATE;
      } else {
        TriRows A{l.tri + t};
        double ricl[9], ticl[3], Rf[9], Pf[3];
        for (int k = 0; k < 9; k++) ricl[k] = l.cam[12 * P + k];
        for (int k = 0; k < 3; k++) ticl[k] = l.cam[12 * P + 9 + k];
        for (int k = 0; k < 9; k++) Rf[k] = l.cam[9 * s + k];
        for (int k = 0; k < 3; k++) Pf[k] = l.cam[9 * P + 3 * s + k];
        double t0[3], R0[9], tmp[3];
        mat3vec(Rf, ticl, tmp);
        for (int k = 0; k < 3; k++) t0[k] = Pf[k] + tmp[k];
        mat3mul(Rf, ricl, R0);
        int row = 0;
        for (int jo = 0; jo < no; jo++) {
          const int imu_j = s + jo;
          const double *pt = bk.obs + ((size_t)i * P + jo) * 3;
          double t1[3], R1[9], R0T[9], dd[3], tt[3], R[9], RT[9], mt[3];
          for (int k = 0; k < 9; k++) Rf[k] = l.cam[9 * imu_j + k];
          for (int k = 0; k < 3; k++) Pf[k] = l.cam[9 * P + 3 * imu_j + k];
          mat3vec(Rf, ticl, tmp);
          for (int k = 0; k < 3; k++) t1[k] = Pf[k] + tmp[k];
          mat3mul(Rf, ricl, R1);
          mat3T(R0, R0T);
          for (int k = 0; k < 3; k++) dd[k] = t1[k] - t0[k];
          mat3vec(R0T, dd, tt);
          mat3mul(R0T, R1, R);
          mat3T(R, RT);
          mat3vec(RT, tt, mt);
          double Pm[12];  // [R^T | -R^T t]
          for (int a = 0; a < 3; a++) {
            for (int b = 0; b < 3; b++) Pm[a * 4 + b] = RT[a * 3 + b];
            Pm[a * 4 + 3] = -mt[a];
          }
          const double nrm = sqrt(pt[0] * pt[0] + pt[1] * pt[1] + pt[2] * pt[2]);
          const double fx = pt[0] / nrm, fy = pt[1] / nrm, fz = pt[2] / nrm;
          for (int b = 0; b < 4; b++) A(row * 4 + b) = fx * Pm[2 * 4 + b] - fz * Pm[0 * 4 + b];
          row++;
          for (int b = 0; b < 4; b++) A(row * 4 + b) = fy * Pm[2 * 4 + b] - fz * Pm[1 * 4 + b];
          row++;
        }
        double v[4];
        smallest_right_singular_vector(A, row, v);
        double dep = v[2] / v[3];
        if (dep < 0.1) dep = kInitDepth;
        bk.depth[i] = dep;
      }
    }
  }
  VIO_SYNC();
  STORE_STAMP(l, 5);
  // para_Feature rows and factors of this window
  const int F = block_scan(cx, n, l.scanA, l.part, [&](int i) { return solved_in_window(bk.nobs[i], bk.start[i], W) ? 1 : 0; });
  const int M = block_scan(cx, n, l.scanB, l.part,
                           [&](int i) { return solved_in_window(bk.nobs[i], bk.start[i], W) ? bk.nobs[i] - 1 : 0; });
  if (t == 0) {
    ctl[C_N] = n, ctl[C_STATUS] = l.misc[0];
    ctl[C_MARG] = enough ? VIO_MARGIN_OLD : VIO_MARGIN_SECOND_NEW;
    ctl[C_TRACK] = track, ctl[C_PNUM] = pnum, ctl[C_F] = F, ctl[C_M] = M;
  }
  STORE_STAMP(l, 6);
}

// ---- pass 2 -----------------------------------------------------------------------------------------------------------
// What build_window + pack_window leave in the batch arrays for the landmark side of window b: para_Feature, the factor
// list (grouped by landmark, host = start frame, one factor per later observation), fstart, and the (host, target) buckets
// with their even-padded, chunk-aligned staging slots. keys / own: unsigned short scratch in LDS.
// The relocalization frame of a window (VINS.cpp:571-631): the window frame the old keyframe was matched to and the
// matched landmarks' ids (ascending) with their observations in the old keyframe. frame < 0: none.
struct LoopIn {
  int frame, n;
  const int *ids;
  const double *xy;  // [n][2]
};

struct PackOut {
  int *hdr;  // [kHdrInts] of window b: H_F, H_M, H_MARG, H_NPAIRS, H_NSLOTS, H_NREV, H_HAS_LOOP, H_LOOP_FRAME are written here
  double *feat;
  int *fhost, *ftarget, *ffeat, *fslot, *fstart, *pair_h, *pair_t, *pair_s0, *pair_s1;
  double *pts_i, *pts_j;
  int Fcap, Mcap, pair_cap;
  size_t slot_cap;
};
enum { PH_F = 1, PH_M = 2, PH_HAS_LOOP = 3, PH_LOOP_FRAME = 4, PH_MARG = 5, PH_NPAIRS = 9, PH_NSLOTS = 10, PH_NREV = 11 };  // = batch.h H_*

VIO_DEV void store_pack(const Cx &cx, const Dims &d, const Bank &bk, int *ctl, const Lds &l, const PackOut &o, int chunk,
                        VIO_AS3 unsigned short *keys /* [Mcap + 8] */, VIO_AS3 unsigned short *own /* [Mcap + 8] */,
                        ldsi bins /* [3 (P+1)^2] */, const LoopIn &loop) {
  const int W = d.W, P = W + 1, np1 = P + 1, nkeys = np1 * np1;
  const int t = VIO_TID(cx);
  const int n = ctl[C_N];
  if (ctl[C_STATUS] != VIO_OK) return;
  STORE_STAMP(l, 8);
  // start frame and observation count of every landmark of the window, two bytes each (0xffff: not in the window; bit 7:
  // it has a relocalization factor), and behind them the index of its match in the old keyframe (-1: none)
  VIO_AS3 unsigned short *ent = (VIO_AS3 unsigned short *)l.term;
  ldsi lpi = (ldsi)l.term + d.Lcap;
  // The reference pairs the landmarks with the old keyframe's ids by ONE forward walk over both lists (VINS.cpp:597-631,
  // vio_window.cpp export_factors): the id pointer r only advances, so a landmark matches when its id is found at or
  // behind where the landmarks ahead of it in the list left r. r ahead of landmark i = the largest (lower bound + 1 if
  // found) among the considered landmarks before it: a prefix maximum.
  VIO_PARFOR(i, n) {
    const int no = bk.nobs[i], s = bk.start[i];
    int enc = 0;  // 0: not considered, else 1 + (lower bound << 1 | found)
    if (loop.frame >= 0 && solved_in_window(no, s, W) && s <= loop.frame && s + no - 1 >= loop.frame) {
      const int id = bk.fid[i];
      int lo = 0, hi = loop.n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (loop.ids[mid] < id) lo = mid + 1;
        else hi = mid;
      }
      enc = 1 + (lo << 1 | ((lo < loop.n && loop.ids[lo] == id) ? 1 : 0));
    }
    lpi[i] = enc;
  }
  VIO_SYNC();
  block_prefix_max(cx, n, l.scanA, l.part, [&](int i) {
    const int enc = lpi[i];
    return enc ? ((enc - 1) >> 1) + ((enc - 1) & 1) : 0;
  });
  VIO_PARFOR(i, n) {
    const int enc = lpi[i];
    const int lb = (enc - 1) >> 1;
    lpi[i] = (enc && ((enc - 1) & 1) && lb >= l.scanA[i]) ? lb : -1;
  }
  if (t == 0) l.misc[7] = 0;
  VIO_SYNC();
  const int F = block_scan(cx, n, l.scanA, l.part, [&](int i) { return solved_in_window(bk.nobs[i], bk.start[i], W) ? 1 : 0; });
  const int M = block_scan(cx, n, l.scanB, l.part, [&](int i) {
    return solved_in_window(bk.nobs[i], bk.start[i], W) ? bk.nobs[i] - 1 + (lpi[i] >= 0 ? 1 : 0) : 0;
  });
  STORE_STAMP(l, 9);
  if (F > o.Fcap || M > o.Mcap) {
    if (t == 0) ctl[C_STATUS] = VIO_ECAP;
    return;
  }
  ldsi cnt = bins, start = bins + nkeys, pkey = bins + 2 * nkeys;
  VIO_PARFOR(k, nkeys) cnt[k] = 0;
  // per landmark: its para_Feature row, its factor range, and for each of its factors who owns it and which bucket it is in
  // (the relocalization factor closes the landmark's group; its target is the loop pose, index P)
  VIO_PARFOR(i, n) {
    const int no = bk.nobs[i], s = bk.start[i];
    if (!solved_in_window(no, s, W)) {
      ent[i] = 0xffff;
      continue;
    }
    const bool lp = lpi[i] >= 0;
    ent[i] = (unsigned short)(s << 8 | no | (lp ? 0x80 : 0));
    const int fi = l.scanA[i], k0 = l.scanB[i];
    o.feat[fi] = 1. / bk.depth[i];
    o.fstart[fi] = k0;
    for (int j = 1; j < no; j++) keys[k0 + j - 1] = (unsigned short)(s * np1 + s + j), own[k0 + j - 1] = (unsigned short)i;
    if (lp) {
      keys[k0 + no - 1] = (unsigned short)(s * np1 + P), own[k0 + no - 1] = (unsigned short)i;
      __hip_atomic_fetch_add(&l.misc[7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  VIO_PARFOR(i, 4) ent[n + i] = 0xffff;  // (the slot pass reads four landmarks at a time)
  if (t == 0) o.fstart[F] = M;
  VIO_SYNC();
  // per factor: host, target, landmark row, the two observations
  VIO_PARFOR(k, M) {
    const int i = own[k], key = keys[k];
    const int s = key / np1, tg = key - s * np1;
    const double *p0 = bk.obs + (size_t)i * P * 3;
    const double a0 = p0[0], a1 = p0[1], a2 = p0[2];
    double b0, b1, b2;
    if (tg == P) {
      const double *q = loop.xy + 2 * lpi[i];
      b0 = q[0], b1 = q[1], b2 = 1.0;
    } else {
      const double *pj = p0 + 3 * (tg - s);
      b0 = pj[0], b1 = pj[1], b2 = pj[2];
    }
    o.fhost[k] = s, o.ftarget[k] = tg, o.ffeat[k] = l.scanA[i];
    o.pts_i[3 * k] = a0, o.pts_i[3 * k + 1] = a1, o.pts_i[3 * k + 2] = a2;
    o.pts_j[3 * k] = b0, o.pts_j[3 * k + 1] = b1, o.pts_j[3 * k + 2] = b2;
    __hip_atomic_fetch_add(&cnt[key], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  VIO_SYNC();
  STORE_STAMP(l, 10);
  // buckets in (host, target) order; every bucket starts on an even slot and does not straddle a staging chunk. The
  // occupied buckets are listed first (a scan), the slot recurrence then walks that short list.
  const int npairs = block_scan(cx, nkeys, l.scanA, l.part, [&](int key) { return cnt[key] ? 1 : 0; });
  VIO_PARFOR(key, nkeys)
    if (cnt[key]) pkey[l.scanA[key]] = key;
  VIO_SYNC();
  if (t == 0) {
    int slot = 0, rc = npairs > o.pair_cap ? VIO_ECAP : VIO_OK;
    for (int pi = 0; pi < npairs && rc == VIO_OK; pi++) {
      const int key = pkey[pi];
      const int c = cnt[key];
      const int cpad = (c + 1) & ~1;
      if (chunk > 0 && cpad <= chunk && slot % chunk + cpad > chunk) slot = (slot / chunk + 1) * chunk;
      if ((size_t)slot + cpad > o.slot_cap) {
        rc = VIO_ECAP;
        break;
      }
      start[key] = slot;
      slot += cpad;
    }
    l.misc[5] = slot, l.misc[6] = rc;
  }
  VIO_SYNC();
  STORE_STAMP(l, 11);
  if (l.misc[6] != VIO_OK) {
    if (t == 0) ctl[C_STATUS] = l.misc[6];
    return;
  }
  // One work-item per bucket: its header, and the slot of every factor in it = the bucket's start + the factor's rank in
  // factor order. A landmark has at most one factor per bucket and the factors follow the landmarks' order, so the rank is
  // the number of earlier landmarks hosted in the bucket's host frame that reach its target frame.
  VIO_PARFOR(pi, npairs) {
    const int key = pkey[pi];
    const int h = key / np1, tg = key - h * np1, dt = tg - h;
    const bool to_loop = tg == P;
    int s = start[key];
    o.pair_h[pi] = h, o.pair_t[pi] = tg, o.pair_s0[pi] = s, o.pair_s1[pi] = s + cnt[key];
    for (int i = 0; i < n; i += 4) {
      const unsigned long long q = *(const VIO_AS3 unsigned long long *)(ent + i);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int e = (int)((q >> (16 * c)) & 0xffff);
        if ((e >> 8) != h) continue;
        const int no = e & 0x7f;
        if (to_loop) {
          if (e & 0x80) o.fslot[l.scanB[i + c] + no - 1] = s++;
        } else if (no > dt) {
          o.fslot[l.scanB[i + c] + dt - 1] = s++;
        }
      }
    }
  }
  STORE_STAMP(l, 12);
  if (t == 0) {
    const int nl = l.misc[7];
    o.hdr[PH_F] = F, o.hdr[PH_M] = M, o.hdr[PH_HAS_LOOP] = nl > 0 ? 1 : 0, o.hdr[PH_LOOP_FRAME] = nl > 0 ? loop.frame : -1, o.hdr[PH_MARG] = ctl[C_MARG];
    ctl[C_M] = M, ctl[C_NLOOP] = nl;
    o.hdr[PH_NPAIRS] = npairs, o.hdr[PH_NSLOTS] = l.misc[5], o.hdr[PH_NREV] = 0;
    ctl[C_NPAIRS] = npairs, ctl[C_NSLOTS] = l.misc[5];
  }
}

// ---- pass 3 -----------------------------------------------------------------------------------------------------------
// failureDetection (VINS.cpp:214-265) on the newest frame of the solved window.
VIO_DEV int failure_reasons(int last_track_num, const double Bg[3], const double Pn[3], const double Rn[9], const double last_P[3],
                            const double last_R[9]) {
  int r = 0;
  if (last_track_num < 4) r |= VIO_FAIL_FEW_FEATURES;
  if (sqrt(Bg[0] * Bg[0] + Bg[1] * Bg[1] + Bg[2] * Bg[2]) > 1) r |= VIO_FAIL_GYR_BIAS;
  const double dd[3] = {Pn[0] - last_P[0], Pn[1] - last_P[1], Pn[2] - last_P[2]};
  if (sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]) > 1) r |= VIO_FAIL_TRANSLATION;
  if (fabs(Pn[2] - last_P[2]) > 0.5) r |= VIO_FAIL_Z_TRANSLATION;
  double RT[9], dR[9];
  mat3T(Rn, RT);
  mat3mul(RT, last_R, dR);
  const Quat dq = RtoQ(dR);
  const double delta_angle = acos(dq.w) * 2.0 / 3.14 * 180.0;
  if (delta_angle > 40) r |= VIO_FAIL_ROTATION;
  return r;
}

// x: the solved para_Feature (after new2old), pose [P][7] / sb [P][9]: the solved window. The list moves from bank `bk`
// to bank `nb`, with the slide of the frame's marginalization flag applied. ctld: last_P, last_R of the sequence.
VIO_DEV void store_finish(const Cx &cx, const Dims &d, const Bank &bk, const Bank &nb, int *ctl, double *ctld, const Lds &l,
                          const double *x, const double *pose, const double *sb, const double *tic, const double *ric) {
  const int W = d.W, P = W + 1;
  const int t = VIO_TID(cx);
  const int n = ctl[C_N];
  if (ctl[C_STATUS] != VIO_OK) return;
  const int marg = ctl[C_MARG];
  STORE_STAMP(l, 16);
  block_scan(cx, n, l.scanA, l.part, [&](int i) { return solved_in_window(bk.nobs[i], bk.start[i], W) ? 1 : 0; });
  if (t == 0) {
    double Rn[9];
    qtoR(qfrom_pose(pose + 7 * W), Rn);
    const int r = failure_reasons(ctl[C_TRACK], sb + 9 * W + 6, pose + 7 * W, Rn, ctld, ctld + 3);
    l.misc[0] = r;
    ctl[C_FAIL] = r;
    if (r) {
      ctl[C_N] = 0;  // clearState: the host restarts the sequence
    } else {
      for (int k = 0; k < 3; k++) ctld[k] = pose[7 * W + k];
      for (int k = 0; k < 9; k++) ctld[3 + k] = Rn[k];
    }
  }
  VIO_SYNC();
  STORE_STAMP(l, 17);
  if (l.misc[0]) return;
  double mR[9], mP[3], nR[9], nP[3];
  if (marg == VIO_MARGIN_OLD) {
    double R0[9], R1[9], tt[3];
    qtoR(qfrom_pose(pose), R0), qtoR(qfrom_pose(pose + 7), R1);
    mat3mul(R0, ric, mR), mat3mul(R1, ric, nR);
    mat3vec(R0, tic, tt);
    for (int k = 0; k < 3; k++) mP[k] = pose[k] + tt[k];
    mat3vec(R1, tic, tt);
    for (int k = 0; k < 3; k++) nP[k] = pose[7 + k] + tt[k];
  }
  // new state of every landmark; scanB = it survives
  VIO_PARFOR(i, n) {
    int s = bk.start[i], no = bk.nobs[i], fl = bk.flag[i];
    double dep = bk.depth[i];
    if (solved_in_window(no, s, W)) {  // setDepth
      dep = 1.0 / x[l.scanA[i]];
      fl = dep < 0 ? 2 : 1;
    }
    int alive = 1, drop = -1;  // drop: the observation column that leaves
    if (marg == VIO_MARGIN_OLD) {
      if (s != 0) {
        s--;
      } else {
        const double *uv = bk.obs + (size_t)i * P * 3;
        drop = 0;
        if (no - 1 < 2) {
          alive = 0;
        } else {
          double pts_i[3], w[3], nRT[9], dd[3], pts_j[3];
          for (int k = 0; k < 3; k++) pts_i[k] = uv[k] * dep;
          mat3vec(mR, pts_i, w);
          for (int k = 0; k < 3; k++) dd[k] = w[k] + mP[k] - nP[k];
          mat3T(nR, nRT);
          mat3vec(nRT, dd, pts_j);
          dep = pts_j[2] > 0 ? pts_j[2] : kInitDepth;
        }
      }
    } else {
      if (s == W) {
        s--;
      } else if (s + no - 1 >= W - 1) {
        drop = W - 1 - s;
        if (no - 1 == 0) alive = 0;
      }
    }
    if (fl == 2) alive = 0;  // removeFailures
    l.term[i] = dep;
    l.scanB[i] = alive;
    // start / count / flag / dropped column of the survivor, packed: start (8 bits) | nobs (8) | flag (2) | drop + 1 (8)
    bk.flag[i] = (s & 0xff) | ((drop >= 0 ? no - 1 : no) & 0xff) << 8 | (fl & 3) << 16 | ((drop + 1) & 0xff) << 18;
  }
  VIO_SYNC();
  STORE_STAMP(l, 18);
  // (scanA is read above through x[...]: the survivors' positions go to a scan of their own)
  const int n_alive = block_scan(cx, n, l.scanA, l.part, [&](int i) { return l.scanB[i]; });
  VIO_PARFOR(i, n) {
    if (!l.scanB[i]) continue;
    const int j = l.scanA[i];
    const int pk = bk.flag[i];
    const int s = pk & 0xff, no = (pk >> 8) & 0xff, fl = (pk >> 16) & 3;
    nb.fid[j] = bk.fid[i], nb.start[j] = s, nb.nobs[j] = no, nb.flag[j] = fl, nb.depth[j] = l.term[i];
  }
  // the observations move column by column, one work-item per (landmark, column), four columns in flight per work-item
  // (a landmark-by-landmark copy is a chain of dependent round trips to HBM)
  {
    const int total = n * P;
    for (int q0 = VIO_TID(cx); q0 < total; q0 += 4 * cx.nt) {
      double v[4][3];
      long dsti[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int q = q0 + u * cx.nt;
        dsti[u] = -1;
        if (q >= total) continue;
        const int i = q / P, c = q - i * P;
        if (!l.scanB[i]) continue;
        const int pk = bk.flag[i];
        const int no = (pk >> 8) & 0xff, drop = ((pk >> 18) & 0xff) - 1;
        if (c == drop) continue;
        const int cc = c - (drop >= 0 && c > drop ? 1 : 0);
        if (cc >= no) continue;
        const double *src = bk.obs + ((size_t)i * P + c) * 3;
        v[u][0] = src[0], v[u][1] = src[1], v[u][2] = src[2];
        dsti[u] = ((long)l.scanA[i] * P + cc) * 3;
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (dsti[u] >= 0) nb.obs[dsti[u]] = v[u][0], nb.obs[dsti[u] + 1] = v[u][1], nb.obs[dsti[u] + 2] = v[u][2];
    }
  }
  STORE_STAMP(l, 19);
  if (t == 0) ctl[C_N] = n_alive, ctl[C_BANK] ^= 1;
}

}  // namespace store
}  // namespace vio
