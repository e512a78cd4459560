// batch.h — HBM layout of a batch of windows, the per-window view the kernel builds from it, the LDS working-set
// layout, and the host-side pack / unpack between the C ABI structs (include/vio_amd.h) and the flat arrays.
//
// Layout: structure-of-arrays across the batch with fixed per-window capacities (Wcap, Fcap, Mcap, Ncap) so that
// window b of every array sits at base + b * stride: one launch covers the whole batch, blockIdx.x = window.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "solver_core.h"
#include "vio_amd.h"

namespace vio {

constexpr int kHdrInts = 12;
enum { H_W = 0, H_F, H_M, H_HAS_LOOP, H_LOOP_FRAME, H_MARG, H_PRIOR_N, H_PRIOR_NB, H_USE_ORIGIN, H_NPAIRS, H_NSLOTS, H_NREV };
constexpr int kHdrDoubles = 4;
constexpr int kMaxPriorBlocks = VIO_MAX_PRIOR_BLOCKS;

struct BatchDims {
  int Wcap, Pcap, Fcap, Mcap, Ncap, Fpad, n6cap, nblk_cap, pair_cap;
  int Flds;  // landmarks the LDS layout is carved for (<= Fcap, which sizes the global strides): every landmark costs LDS
  int lds_asp;  // LDS pose matrix: the IMU speed-bias x pose coupling (AspI, 14 KB at W = 10) sits in LDS too (1, faster)
                // or in global scratch (0: leaves the room to windows with many landmarks, two of which then share a CU)
  int prof_stages;  // LDS slots of the stage clock: ST_COUNT for the launches of vio_backend_set_profile, 0 for the product's
  int max_iter;
  double s_info, gravity, cauchy_b;
};

// Wcap/Fcap/Mcap: largest window / feature / factor count in the batch; any_loop: some window carries a loop pose
// (one extra 6-dof block).
inline BatchDims make_dims(const VioConfig &cfg, int Wcap, int Fcap, int Mcap, bool any_loop) {
  BatchDims d;
  d.Wcap = Wcap, d.Pcap = Wcap + 1, d.Fcap = Fcap > 0 ? Fcap : 1, d.Mcap = Mcap > 0 ? Mcap : 1;
  d.Ncap = 15 * d.Pcap + 6;
  d.Fpad = (d.Fcap + 7) / 8 * 8;
  d.Flds = d.Fcap;
  d.lds_asp = 1;
  d.prof_stages = 0;
  d.n6cap = 6 * (d.Pcap + 1);  // pose groups 0..P-1 plus one more: loop pose (solve) / extrinsic (marginalization)
  d.nblk_cap = d.Pcap + (any_loop ? 1 : 0);
  d.pair_cap = (d.Pcap + 1) * (d.Pcap + 2) / 2;  // distinct (host, target) pairs incl. the loop pose
  d.max_iter = cfg.max_iterations;
  d.s_info = cfg.fx / 1.5;  // ProjectionFactor::sqrt_info = FOCUS_LENGTH_X / 1.5 (VINS.cpp:31)
  d.gravity = cfg.gravity;
  d.cauchy_b = cfg.cauchy_a * cfg.cauchy_a;
  return d;
}

// Pose matrix of the reduced system: 6 (P + loop pose) unknowns + the carried right-hand side row; leading dimension of
// the speed-bias x pose rows (whole tiles).
inline int pose_rows(const BatchDims &d) { return 6 * d.nblk_cap + 1; }
inline int pose_jp(const BatchDims &d) { return (pose_rows(d) + 15) / 16 * 16; }

// Staging slots of a window: its factors, one padding slot per (host, target) bucket (buckets start on even slots), and the
// slots skipped so that no bucket straddles a staging chunk (pack_window): bounded by the factors once more.
// Cooperative windows (several workgroups per window, solver_core.h): flags and payload of one window in its scratch.
inline size_t coop_doubles(const BatchDims &d) { return CoopLayout::make(d.Pcap, d.Fcap, d.nblk_cap).total; }
inline size_t slot_capacity(const BatchDims &d) { return 2 * (size_t)d.Mcap + d.pair_cap + 2; }
// Gram pieces (solver_core.h): a (host, target) bucket is cut at staging-chunk boundaries and into runs of kGramPiece slots
inline size_t gram_piece_capacity(const BatchDims &d) { return slot_capacity(d) / kGramPiece + 2 * (size_t)d.pair_cap + 64; }
inline size_t gram_chunk_capacity(const BatchDims &d) { return slot_capacity(d) / kGramMinChunk + 8; }

// Element counts per window of every array (strides).
struct BatchStrides {
  size_t pose, sb, ex, feat, fint, pts, preint, pr_int, pr_x0, pr_J, pr_r, fstart, pair;
  size_t scratch, hm;
  size_t out_pose, out_sb, out_feat, out_loop, stats_d, stats_i;
  // offsets inside the per-window scratch block (doubles)
  size_t s_info, s_aug, s_J, s_M, s_r, s_Mr, s_prJT, s_prH0, s_WTf, s_PP, s_sfact, s_Asp, s_AspG, s_AppPr, s_srec_i, s_srec_d, s_stash, s_coop, s_gpiece, s_gstart;
};

// The arrays of the per-window scratch block whose size depends on the window size alone (and on the prior capacity, itself a
// function of the window size in every batch the product builds) come FIRST: with the window size a compile-time constant
// (vio_window_kernel.inc, WS > 0) their offsets are constants and the view of a window is one base address. Returns the end.
VIO_HD size_t scratch_fixed_offsets(int Wcap, int Ncap, int nblk_cap, BatchStrides &s) {
  const int Pcap = Wcap + 1, prows = 6 * nblk_cap + 1, jp = (prows + 15) / 16 * 16;
  size_t o = 0;
  s.s_info = o, o += (size_t)Wcap * 225;
  s.s_aug = o, o += (size_t)Wcap * 450;
  s.s_J = o, o += (size_t)Wcap * 450;
  s.s_M = o, o += (size_t)Wcap * 450;
  s.s_r = o, o += (size_t)Wcap * 15;
  s.s_Mr = o, o += (size_t)Wcap * 15;
  s.s_prJT = o, o += (size_t)Ncap;  // b0 = J0^T r0
  s.s_prH0 = o, o += (size_t)Ncap * Ncap;
  s.s_PP = o, o += tri_doubles(prows);  // (layout of App)
  s.s_Asp = o, o += (size_t)kSB * jp;  // the prior's speed-bias x pose block
  s.s_AspG = o, o += (size_t)Pcap * kAS;
  s.s_AppPr = o, o += tri_doubles(prows) + 2 * (size_t)Pcap * kSS;
  return o;
}

inline BatchStrides make_strides(const BatchDims &d) {
  BatchStrides s;
  s.pose = 7 * (size_t)d.Pcap, s.sb = 9 * (size_t)d.Pcap, s.ex = 7, s.feat = d.Fcap;
  s.fint = d.Mcap, s.pts = 3 * (size_t)d.Mcap, s.preint = (size_t)d.Wcap * kPreintDoubles;
  s.fstart = (size_t)d.Fcap + 1, s.pair = d.pair_cap;
  s.pr_int = kMaxPriorBlocks, s.pr_x0 = 9 * (size_t)kMaxPriorBlocks, s.pr_J = (size_t)d.Ncap * d.Ncap, s.pr_r = d.Ncap;
  const size_t fixed_end = scratch_fixed_offsets(d.Wcap, d.Ncap, d.nblk_cap, s);
  size_t o = fixed_end;
  // ---- arrays sized by the landmark / factor capacities of the batch
  s.s_WTf = o, o += (size_t)d.n6cap * d.Fpad;
  s.s_sfact = o, o += (slot_capacity(d) + 1) / 2;  // ints: staging slot -> factor
  const size_t nslots_cap = slot_capacity(d);
  s.s_srec_i = o, o += (nslots_cap + 1) / 2;
  s.s_srec_d = o, o += 6 * nslots_cap;
  s.s_stash = o, o += 7 * (size_t)(d.Pcap + 1) + 9 * (size_t)d.Pcap + 4 * (size_t)d.Fcap + 3 * (size_t)(d.Pcap + 1) * kBS;
  s.s_coop = o, o += coop_doubles(d);  // cooperative windows: flags + payload (solver_core.h, CoopLayout)
  s.s_gpiece = o, o += (gram_piece_capacity(d) + 1) / 2;  // ints: the Gram pieces of the window (solver_core.h, build_gram_pieces)
  s.s_gstart = o, o += (gram_chunk_capacity(d) + 1) / 2;  // ints: first piece of every staging chunk
  s.scratch = (o + 7) / 8 * 8;
  // the pose matrix when it lives in global memory, and behind it the fill panels V_k of the band (factor_band_lds):
  // [Pcap][tile columns][3][64] in operand layout, consumed by ONE rank-9 P product per factorization
  s.hm = tri_doubles(6 * d.nblk_cap + 1) + 16 + (size_t)d.Pcap * ((6 * (size_t)d.nblk_cap + 1 + 15) / 16) * 192;
  s.out_pose = s.pose, s.out_sb = s.sb, s.out_feat = s.feat, s.out_loop = 7;
  s.stats_d = kStatsDoubles, s.stats_i = kStatsInts;
  return s;
}

// Device-resident prior chain: where window b reads its prior data from (null J: the strided pr_* arrays) and where it
// writes the next one (null mJ: the strided MargPtrs arrays).
struct PriorTab {
  const double *x0, *J, *r;
  double *mx0, *mJ, *mr;
  int ncap;  // capacity of mJ (ncap x ncap) and mr
};

// Base pointers (device or host, depending on who fills it).
struct BatchPtrs {
  int n;
  BatchDims d;
  BatchStrides s;
  const int *hdr;         // [n][kHdrInts]
  const double *hdr_d;    // [n][kHdrDoubles]
  const double *pose, *sb, *ex, *feat;
  const int *fhost, *ftarget, *ffeat;
  const int *fslot, *fstart, *pair_h, *pair_t, *pair_s0, *pair_s1;
  const double *pts_i, *pts_j, *preint;
  const int *pr_kind, *pr_index, *pr_offset;
  const double *pr_x0, *pr_J, *pr_r;
  const PriorTab *ptab;  // [n] or null
  double *scratch;   // [n][s.scratch]
  double *hm;        // [n][s.hm] (only used when the matrix does not fit LDS)
  const int *order;  // launch-local block index -> window (null: identity); a batch may be split into two launches
  int coop;          // workgroups per window of this launch (1: each window is solved by one workgroup)
  int n_launch;      // windows of this launch (cooperative launches pad their grid to whole groups of eight windows)
  unsigned coop_spin;  // polls before a wait between the workgroups of a cooperative window gives up (kCoopSpinLimit; tests lower it)
  int coop_fault;      // test hook (VIO_AMD_COOP_FAULT=1): the helper workgroups leave at once, so that the owner's first wait times out
  double *out_pose, *out_sb, *out_feat, *raw_pose, *raw_sb, *raw_feat, *out_loop, *stats_d;
  int *stats_i;
};

// Where the marginalization phase of the window kernel writes the next prior (strided arrays; a window that names a slot of
// the resident prior store writes through PriorTab instead), and the stage clock's switches.
struct MargPtrs {
  int *ints;        // [n][4 + 3 * kMaxPriorBlocks]
  double *x0;       // [n][9 * kMaxPriorBlocks]
  double *J;        // [n][Ncap * Ncap]
  double *r;        // [n][Ncap]
  double *scratch;  // [n][marg_scratch] (global matrix variant only)
  size_t s_ints, s_x0, s_J, s_r, s_scratch;
  long long *prof;  // [n][ST_COUNT] or null
  int prof_tid;     // work-item that keeps the stage clock (VIO_AMD_PROF_TID, default 0)
  int wrot;         // wave-role rotation: -1 = from the hardware wave slot (default), else forced (VIO_AMD_WAVE_ROT)
};

VIO_HD WinView make_view(const BatchPtrs &B, int b) {
  WinView v;
  const int *h = B.hdr + (size_t)b * kHdrInts;
  const double *hd = B.hdr_d + (size_t)b * kHdrDoubles;
  v.W = h[H_W], v.P = v.W + 1, v.F = h[H_F], v.M = h[H_M];
  v.has_loop = h[H_HAS_LOOP], v.loop_frame = h[H_LOOP_FRAME], v.marg_flag = h[H_MARG];
  v.prior_n = h[H_PRIOR_N], v.prior_nb = h[H_PRIOR_NB];
  v.np = kBS * v.P + (v.has_loop ? 6 : 0);
  v.nblk = v.P + (v.has_loop ? 1 : 0);
  v.max_iter = B.d.max_iter;
  v.Pcap = B.d.Pcap, v.Fcap = B.d.Fcap, v.nblk_cap = B.d.nblk_cap;
  v.Fpad = B.d.Fpad;
  v.npose6 = 6 * (v.P + (v.has_loop ? 1 : 0));
  v.n6 = v.npose6, v.nrows = v.n6 + 1, v.nT = (v.nrows + 15) >> 4, v.jp = (6 * B.d.nblk_cap + 1 + 15) / 16 * 16;
  v.s_info = B.d.s_info, v.gravity = B.d.gravity, v.cauchy_b = B.d.cauchy_b;
  v.pose0 = B.pose + b * B.s.pose, v.sb0 = B.sb + b * B.s.sb, v.ex = B.ex + b * B.s.ex, v.feat0 = B.feat + b * B.s.feat;
  v.fhost = B.fhost + b * B.s.fint, v.ftarget = B.ftarget + b * B.s.fint, v.ffeat = B.ffeat + b * B.s.fint;
  v.fslot = B.fslot + b * B.s.fint, v.fstart = B.fstart + b * B.s.fstart;
  v.pair_h = B.pair_h + b * B.s.pair, v.pair_t = B.pair_t + b * B.s.pair;
  v.pair_s0 = B.pair_s0 + b * B.s.pair, v.pair_s1 = B.pair_s1 + b * B.s.pair;
  v.npairs = h[H_NPAIRS], v.nslots = h[H_NSLOTS], v.n6cap = B.d.n6cap, v.nrev = h[H_NREV];
  v.pts_i = B.pts_i + b * B.s.pts, v.pts_j = B.pts_j + b * B.s.pts;
  v.preint = B.preint + b * B.s.preint;
  v.pr_kind = B.pr_kind + b * B.s.pr_int, v.pr_index = B.pr_index + b * B.s.pr_int;
  v.pr_offset = B.pr_offset + b * B.s.pr_int;
  v.pr_x0 = B.pr_x0 + b * B.s.pr_x0, v.pr_J = B.pr_J + b * B.s.pr_J, v.pr_r = B.pr_r + b * B.s.pr_r;
  if (B.ptab && B.ptab[b].J) v.pr_x0 = B.ptab[b].x0, v.pr_J = B.ptab[b].J, v.pr_r = B.ptab[b].r;
  v.use_origin = h[H_USE_ORIGIN];
  v.origin_yaw = hd[0], v.origin_p[0] = hd[1], v.origin_p[1] = hd[2], v.origin_p[2] = hd[3];
  double *sc = B.scratch + b * B.s.scratch;
  v.imu_info = sc + B.s.s_info, v.imu_aug = sc + B.s.s_aug, v.imu_J = sc + B.s.s_J, v.imu_M = sc + B.s.s_M;
  v.imu_r = sc + B.s.s_r, v.imu_Mr = sc + B.s.s_Mr, v.prb0 = sc + B.s.s_prJT, v.prH0 = sc + B.s.s_prH0;
  v.WTf = sc + B.s.s_WTf, v.PP = sc + B.s.s_PP, v.Apri = sc + B.s.s_Asp, v.AspG = sc + B.s.s_AspG, v.AppPr = sc + B.s.s_AppPr;
  v.srec_i = reinterpret_cast<int *>(sc + B.s.s_srec_i), v.srec_d = sc + B.s.s_srec_d;
  v.gpiece = reinterpret_cast<int *>(sc + B.s.s_gpiece), v.gstart = reinterpret_cast<int *>(sc + B.s.s_gstart);
  v.sfact = reinterpret_cast<int *>(sc + B.s.s_sfact);
  v.stash = sc + B.s.s_stash;
  v.coop = sc + B.s.s_coop;
  v.out_pose = B.out_pose + b * B.s.out_pose, v.out_sb = B.out_sb + b * B.s.out_sb;
  v.out_feat = B.out_feat + b * B.s.out_feat;
  v.raw_pose = B.raw_pose + b * B.s.out_pose, v.raw_sb = B.raw_sb + b * B.s.out_sb;
  v.raw_feat = B.raw_feat + b * B.s.out_feat, v.out_loop = B.out_loop + b * B.s.out_loop;
  v.stats_d = B.stats_d + b * B.s.stats_d, v.stats_i = B.stats_i + b * B.s.stats_i;
  return v;
}

constexpr size_t kLdsBytes = 160 * 1024;  // LDS of one CU (gfx950), the budget of one window

// ---- LDS working set ------------------------------------------------------------------------------------
// Carves `base` (LDS, 16-byte aligned) into the Work arrays for capacities d. When lds_matrix is false the matrix
// lives in global memory (hm_global). Returns the number of bytes used.
template <class MP>
struct MatPick;
template <>
struct MatPick<double *> {
  static VIO_HD double *get(bool lds_matrix, ldsd l, double *g) { return lds_matrix ? (double *)l : g; }
};
#ifndef VIO_HOST_BUILD
template <>
struct MatPick<ldsd> {
  static VIO_HD ldsd get(bool, ldsd l, double *) { return l; }
};
#endif

// Everything carve_work lays out, returned BY VALUE: the kernel must not take the address of its WorkT / Ctx objects.
// (A null test of a pointer to a private-memory object is not folded by LLVM -- in the private address space 0 is a
// valid address -- and one such compare is enough to keep the whole struct out of registers: the first version of the
// kernel read every w.field and cx.field back from scratch memory, ~1100 scratch_load sites.)
template <class MP, class AP = MP>
struct Carved {
  WorkT<MP, AP> w;
  ldsd red;
  VIO_AS3 long long *lprof;
  size_t bytes, state_end_doubles;
  size_t tail_doubles;  // the landmarks of the iterate sit at the END of the workgroup's LDS (see below)
};

// Layout (round 6): the fixed-size part of the iterate (poses, speed-bias, extrinsic) and the reduction scratch come first --
// the marginalization phase re-carves everything behind them (marg_core.h) --, then every array whose size depends on the
// window size alone, then the per-landmark arrays, and the landmarks of the ITERATE (xfeat) at the very end of the
// workgroup's LDS (total_doubles; 0 = measuring: directly behind the rest). With the window size a compile-time constant
// (vio_window_kernel.inc, WS > 0) every address of the first two groups is then a constant -- an immediate offset of a
// ds_read / ds_write instead of a scalar register --, and the per-landmark arrays are one base plus multiples of one stride.
template <class MP, class AP = MP>
VIO_HD Carved<MP, AP> carve_all(const BatchDims &d, bool lds_matrix, int nthreads, ldsd base, double *hm_global, double *asp_global = nullptr,
                                size_t total_doubles = 0) {
  Carved<MP, AP> c;
  size_t o = 0;
  const size_t npc = (size_t)d.nblk_cap * kBS;  // pose-side vector length (frame-major; the loop pose uses 6 of its 15)
  const size_t F = d.Flds, Fe = (F + 1) & ~(size_t)1;
  auto take = [&](size_t n) {
    ldsd p = base + o;  // (a null base only measures; the pointers are then never used)
    o += (n + 1) & ~(size_t)1;  // keep 16-byte alignment
    return p;
  };
  WorkT<MP, AP> &w = c.w;
  // the iterate comes first: the marginalization phase re-carves everything behind it (marg_core.h) up to the landmarks at the end
  w.xpose = take(7 * (size_t)(d.Pcap + 1)), w.xsb = take(9 * (size_t)d.Pcap);
  w.ex = take(8);
  c.red = take(6 * ((size_t)nthreads / 64) + 2);
  c.lprof = reinterpret_cast<VIO_AS3 long long *>(take((size_t)d.prof_stages));
  c.state_end_doubles = o;
  c.tail_doubles = Fe;
  // the reduced matrix: pose matrix (LDS or global) and the speed-bias band (always LDS), contiguous when both are in
  // LDS so that the Jacobian rows of the projection factors can be staged across them
  const size_t napp = tri_doubles(6 * (size_t)d.nblk_cap + 1);
  const size_t o_mat = o;
  ldsd app = nullptr;
  if (lds_matrix) app = take(napp);
  w.App = MatPick<MP>::get(lds_matrix, app, hm_global);
  w.VG = lds_matrix || !hm_global ? nullptr : hm_global + napp + 16;
  // (pose matrix in global scratch: the band, the fill-tile buffer and what LDS is left over sit behind the vectors,
  // contiguous, and stage the Jacobian rows -- see below)
  if (lds_matrix) w.Dss = take(2 * (size_t)d.Pcap * kSS), w.Css = w.Dss + (size_t)d.Pcap * kSS;
  const bool lds_asp = lds_matrix && d.lds_asp != 0;
  ldsd aspi = lds_asp ? take((size_t)d.Pcap * kAS + 16) : nullptr;
  w.AspI = MatPick<AP>::get(lds_asp, aspi, asp_global);
  w.stage = app, w.nstage = lds_matrix ? (int)(o - o_mat) : 0;
  w.asp_ring = lds_matrix && !lds_asp;
  w.asp_lds = lds_asp;
  w.aspring = w.asp_ring ? take(2 * (size_t)kAS + 4) : nullptr;
  w.gp = take(npc), w.sp = take(npc), w.dp = take(npc);
  // Aliases. While the Jacobians are evaluated the candidate pose / speed-bias and the Gauss-Newton step are dead: the
  // sixth accumulator of the landmarks' host-frame coupling (ef) uses their place (when it is large enough). t1 (the
  // solution of the linear solve), t2, the pose-index work vector and the pivot reciprocals are dead then as well: the
  // diagonal pose blocks of the projection Gram products (ppd, 36 per frame) accumulate there. The trust-region step
  // is formed after the solution has been consumed: it shares t1.
  const size_t o_ef = o;
  w.cpose = take(7 * (size_t)(d.Pcap + 1)), w.csb = take(9 * (size_t)d.Pcap), w.gnp = take(npc);
  const bool ef_alias = o - o_ef >= F;
  const size_t jp = (6 * (size_t)d.nblk_cap + 1 + 15) / 16 * 16;
  const size_t o_ppd = o;
  w.t1 = take(npc), w.t2 = take(npc), w.xt = take(jp), w.ldinv = take((size_t)d.Pcap * kSB + jp);
  w.stp = w.t1;
  w.ppd = w.t1;
  if (o - o_ppd < 36 * (size_t)(d.Pcap + 1)) (void)take(36 * (size_t)(d.Pcap + 1) - (o - o_ppd));
  w.prdx = take(d.Ncap), w.prr = take(d.Ncap);
  w.prcol = reinterpret_cast<ldsi>(take(((size_t)d.Ncap + 1) / 2 + 1));
  w.sbr = reinterpret_cast<ldsi>(take((size_t)d.Pcap + 1));
  w.flag = reinterpret_cast<ldsi>(take(2));
  w.ready = reinterpret_cast<ldsi>(take(((size_t)d.Pcap + 1) / 2 + 1));
  w.park = take(24);
  w.rot = take(9 * (size_t)(d.Pcap + 2));
  // pose matrices of more than kPanelTiles tile rows: the fill tiles of the band go through LDS
  w.vbuf = nullptr;
  if (lds_matrix && jp > 16 * (size_t)kPanelTiles) w.vbuf = take((jp / 16) * 192);
  // ---- per-landmark arrays: one base, multiples of Fe
  w.cfeat = take(F);
  w.gf = take(F), w.sf = take(F), w.gnf = take(F), w.stf = take(F), w.hff = take(F), w.einv = take(F), w.tf = take(F);
  w.ef = ef_alias ? w.cpose : take(F);
  w.fh = reinterpret_cast<ldsi>(take((F + 1) / 2 + 1));
  if (!lds_matrix) {
    // Pose matrix in global scratch (W > 12): while the Jacobians are evaluated the band, the fill-tile buffer and whatever
    // LDS the layout leaves free (a workgroup of this variant has the CU to itself) stage the Jacobian rows of the projection
    // factors, a chunk of up to one slot per work-item -- the first three versions staged them in the global matrix buffer,
    // two L2 round trips per Gram operand batch.
    const size_t o_st = o;
    w.Dss = take(2 * (size_t)d.Pcap * kSS), w.Css = w.Dss + (size_t)d.Pcap * kSS;
    w.vbuf = take((jp / 16) * 192);
    const size_t want = (size_t)nthreads * kGSlot, have = o - o_st, cap = kLdsBytes / sizeof(double);
    const size_t room = cap > o + Fe ? cap - o - Fe : 0;
    if (want > have) (void)take(want - have < room ? want - have : (room & ~(size_t)1));
    w.stage = w.Dss, w.nstage = (int)(o - o_st);
  }
  // the landmarks of the iterate: the last Fe doubles of the workgroup's LDS
  w.xfeat = base + (total_doubles ? total_doubles - Fe : o);
  o += Fe;
  c.bytes = o * sizeof(double);
  return c;
}

// Pointer form for host code (sizing: all outputs optional). Returns the number of bytes used. tail_doubles: what the
// marginalization phase must leave alone at the end of the LDS (the iterate's landmarks).
template <class MP>
VIO_HD size_t carve_work(const BatchDims &d, bool lds_matrix, int nthreads, ldsd base, double *hm_global,
                         WorkT<MP> *w, Ctx *cx, size_t *state_end_doubles = nullptr, size_t *tail_doubles = nullptr) {
  const Carved<MP> c = carve_all<MP>(d, lds_matrix, nthreads, base, hm_global);
  if (w) *w = c.w;
  if (cx) cx->red = c.red, cx->lprof = c.lprof;
  if (state_end_doubles) *state_end_doubles = c.state_end_doubles;
  if (tail_doubles) *tail_doubles = c.tail_doubles;
  return c.bytes;
}

// Staging slots per pass of the Jacobian evaluation (projections_jac computes the same from the carved layout).
inline int stage_chunk_slots(const BatchDims &d, bool lds_matrix, int nthreads) {
  const Carved<double *> c = carve_all<double *>(d, lds_matrix, nthreads, nullptr, nullptr);
  int ch = (c.w.nstage / kGSlot) & ~1;
  if (ch >= nthreads) ch -= ch % nthreads;
  return ch;
}

// ---- host-side staging ---------------------------------------------------------------------------------
// Staging vectors live in page-locked memory in the device build (hipMemcpyAsync then runs at link speed and really is
// asynchronous); the host emulation of the kernel (tests/emul, -DVIO_EMUL) uses plain vectors.
#if defined(__HIPCC__) && !defined(VIO_EMUL)
template <class T>
struct PinnedAllocator {
  typedef T value_type;
  PinnedAllocator() = default;
  template <class U>
  PinnedAllocator(const PinnedAllocator<U> &) {}
  T *allocate(size_t n) {
    void *p = nullptr;
    if (hipHostMalloc(&p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault) != hipSuccess) throw std::bad_alloc();
    return static_cast<T *>(p);
  }
  void deallocate(T *p, size_t) { (void)hipHostFree(p); }
  template <class U>
  bool operator==(const PinnedAllocator<U> &) const { return true; }
  template <class U>
  bool operator!=(const PinnedAllocator<U> &) const { return false; }
};
template <class T>
using HostVec = std::vector<T, PinnedAllocator<T>>;
#else
template <class T>
using HostVec = std::vector<T>;
#endif

// View of one array inside the staging arena (the std::vector members it replaces: same accessors).
template <class T>
struct Span {
  T *p = nullptr;
  size_t n = 0;
  T *data() const { return p; }
  size_t size() const { return n; }
  T &operator[](size_t i) const { return p[i]; }
  T *begin() const { return p; }
  T *end() const { return p + n; }
};

// All inputs of a batch in ONE page-locked arena, laid out exactly like the device-side input arena: the upload is one
// hipMemcpyAsync of [0, in_bytes) (plus the prior data behind it when priors travel through the host) instead of one
// call per array (~20 us each, more than the transfer itself for small batches).
struct HostBatch {
  BatchDims d;
  BatchStrides s;
  int n = 0;
  bool sized = false;
  HostVec<unsigned char> arena;
  size_t in_bytes = 0, total_bytes = 0;  // [0, in_bytes): everything but the prior data; [in_bytes, total_bytes): pr_x0, pr_J, pr_r
  Span<int> hdr, fhost, ftarget, ffeat, pr_kind, pr_index, pr_offset, fslot, fstart, pair_h, pair_t, pair_s0, pair_s1;
  Span<double> hdr_d, pose, sb, ex, feat, pts_i, pts_j, preint, pr_x0, pr_J, pr_r;
  // byte offset of every array in the arena, in the order of `arrays()` below (the device arena uses the same table)
  static constexpr int kArrays = 24;
  size_t off[kArrays];
  // poison: fill every double with NaN first (VIO_AMD_POISON) so that a kernel reading staging padding is caught.
  // Same shape as the previous batch: nothing is refilled, pack_window rewrites every field the kernel reads and the
  // padding keeps finite values of earlier windows.
  void resize(const BatchDims &dims, int n_, bool poison = false) {
    BatchDims a = d, b = dims;
    a.Flds = b.Flds = 0, a.lds_asp = b.lds_asp = 0, a.prof_stages = b.prof_stages = 0;  // the LDS carve does not change the staging layout
    if (sized && !poison && n == n_ && memcmp(&a, &b, sizeof(BatchDims)) == 0) {
      d.Flds = dims.Flds, d.lds_asp = dims.lds_asp, d.prof_stages = dims.prof_stages;
      return;
    }
    d = dims, s = make_strides(dims), n = n_, sized = true;
    const size_t N = (size_t)n;
    Span<int> *iv[] = {&hdr, &fhost, &ftarget, &ffeat, &fslot, &fstart, &pair_h, &pair_t, &pair_s0, &pair_s1, &pr_kind, &pr_index, &pr_offset};
    const size_t ic[] = {N * kHdrInts, N * s.fint, N * s.fint, N * s.fint, N * s.fint, N * s.fstart, N * s.pair, N * s.pair, N * s.pair,
                         N * s.pair, N * s.pr_int, N * s.pr_int, N * s.pr_int};
    Span<double> *dv[] = {&hdr_d, &pose, &sb, &ex, &feat, &pts_i, &pts_j, &preint, &pr_x0, &pr_J, &pr_r};
    const size_t dc[] = {N * kHdrDoubles, N * s.pose, N * s.sb, N * s.ex, N * s.feat, N * s.pts, N * s.pts, N * s.preint,
                         N * s.pr_x0, N * s.pr_J, N * s.pr_r};
    size_t o = 0;
    int k = 0;
    auto place = [&](size_t bytes) {
      const size_t at = o;
      o = (o + bytes + 255) & ~(size_t)255;
      return at;
    };
    for (int i = 0; i < 13; i++) off[k++] = place(ic[i] * sizeof(int));
    for (int i = 0; i < 8; i++) off[k++] = place(dc[i] * sizeof(double));
    in_bytes = o;
    for (int i = 8; i < 11; i++) off[k++] = place(dc[i] * sizeof(double));
    total_bytes = o;
    if (arena.size() < total_bytes) arena.assign(total_bytes + total_bytes / 8, 0);
    else memset(arena.data(), 0, total_bytes);
    k = 0;
    for (int i = 0; i < 13; i++) iv[i]->p = reinterpret_cast<int *>(arena.data() + off[k++]), iv[i]->n = ic[i];
    for (int i = 0; i < 11; i++) dv[i]->p = reinterpret_cast<double *>(arena.data() + off[k++]), dv[i]->n = dc[i];
    std::fill(feat.begin(), feat.end(), 1.0);
    if (poison) {
      double nan;
      const unsigned long long bits = 0x7ff8dead0000beefULL;
      memcpy(&nan, &bits, sizeof(nan));
      for (Span<double> *v : dv) std::fill(v->begin(), v->end(), nan);
    }
  }
};

// Validates one window against the capacities and writes it into slot b. Returns VIO_OK / VIO_EINVAL / VIO_ECAP.
// store_ok: the window names a valid slot of a reserved prior store, so a prior without data pointers is acceptable.
// chunk: staging slots per pass of the device's Jacobian evaluation (stage_chunk_slots), 0 = unknown. A bucket that would
// straddle a multiple of it starts at the next one: every bucket's Gram product then ends in ONE plain store of its
// off-diagonal pose block instead of a read-modify-write across two passes (a global round trip on the wave's path).
// straddles: set when some bucket of the window crosses a chunk boundary after all (chunk unknown, or a bucket larger than a chunk):
// cooperative launches need every bucket inside one chunk (one writer per off-diagonal block).
inline int pack_window(HostBatch &hb, int b, const VioWindow &w, bool store_ok = false, int chunk = 0, bool *straddles = nullptr) {
  const BatchDims &d = hb.d;
  const BatchStrides &s = hb.s;
  const int W = w.window_size, P = W + 1, F = w.n_features, M = w.n_factors;
  if (W < 1 || F < 0 || M < 0 || !w.pose || !w.speed_bias || !w.ex_pose || !w.preint) return VIO_EINVAL;
  if (W > d.Wcap || F > d.Fcap || M > d.Mcap) return VIO_ECAP;
  if ((F > 0 && !w.inv_depth) || (M > 0 && (!w.factor_host || !w.factor_target || !w.factor_feature ||
                                            !w.factor_pts_i || !w.factor_pts_j)))
    return VIO_EINVAL;
  int has_loop = 0, nrev = 0;
  for (int k = 0; k < M; k++) {
    int h = w.factor_host[k], t = w.factor_target[k], f = w.factor_feature[k];
    if (f < 0 || f >= F || h < 0 || h >= P || t < 0 || t > P || t == h) return VIO_EINVAL;
    if (k > 0 && f < w.factor_feature[k - 1]) return VIO_EINVAL;  // factors come grouped by ascending feature
    if (t == P) has_loop = 1;
    if (t < h) nrev++;
  }
  {
    // Device-side accumulation layout: factors bucketed by (host, target) pair, each bucket starting on an even slot
    // (one MFMA step consumes two factors = four Jacobian rows; an odd tail is masked, never read), plus the factor range of every feature.
    const int np1 = P + 1;
    std::vector<int> cnt((size_t)np1 * np1, 0), start((size_t)np1 * np1, 0), fill((size_t)np1 * np1, 0);
    for (int k = 0; k < M; k++) cnt[(size_t)w.factor_host[k] * np1 + w.factor_target[k]]++;
    int npairs = 0, slot = 0;
    for (int hh = 0; hh < np1; hh++)
      for (int tt = 0; tt < np1; tt++) {
        int c = cnt[(size_t)hh * np1 + tt];
        if (!c) continue;
        if (npairs >= d.pair_cap) return VIO_ECAP;
        const int cpad = (c + 1) & ~1;
        if (chunk > 0 && cpad <= chunk && slot % chunk + cpad > chunk) slot = (slot / chunk + 1) * chunk;
        if (straddles && (chunk <= 0 || cpad > chunk)) *straddles = true;
        if ((size_t)slot + cpad > slot_capacity(d)) return VIO_ECAP;
        start[(size_t)hh * np1 + tt] = slot;
        hb.pair_h[b * s.pair + npairs] = hh, hb.pair_t[b * s.pair + npairs] = tt;
        hb.pair_s0[b * s.pair + npairs] = slot;
        hb.pair_s1[b * s.pair + npairs] = slot + c;  // real end; the next bucket starts on an even slot
        slot += (c + 1) & ~1;
        npairs++;
      }
    for (int k = 0; k < M; k++) {
      size_t key = (size_t)w.factor_host[k] * np1 + w.factor_target[k];
      hb.fslot[b * s.fint + k] = start[key] + fill[key]++;
    }
    int *fs = &hb.fstart[b * s.fstart];
    for (int f = 0; f <= F; f++) fs[f] = 0;
    for (int k = 0; k < M; k++) fs[w.factor_feature[k] + 1]++;
    for (int f = 0; f < F; f++) fs[f + 1] += fs[f];
    hb.hdr[(size_t)b * kHdrInts + H_NPAIRS] = npairs, hb.hdr[(size_t)b * kHdrInts + H_NSLOTS] = slot;
  }
  if (has_loop && (w.loop_frame < 0 || w.loop_frame >= W)) return VIO_EINVAL;
  int *h = &hb.hdr[(size_t)b * kHdrInts];
  h[H_W] = W, h[H_F] = F, h[H_M] = M, h[H_HAS_LOOP] = has_loop, h[H_LOOP_FRAME] = w.loop_frame;
  h[H_MARG] = w.marginalization_flag, h[H_USE_ORIGIN] = w.use_origin_override, h[H_NREV] = nrev;
  double *hd = &hb.hdr_d[(size_t)b * kHdrDoubles];
  hd[0] = w.origin_yaw_deg, hd[1] = w.origin_p[0], hd[2] = w.origin_p[1], hd[3] = w.origin_p[2];
  memcpy(&hb.pose[b * s.pose], w.pose, sizeof(double) * 7 * P);
  memcpy(&hb.sb[b * s.sb], w.speed_bias, sizeof(double) * 9 * P);
  memcpy(&hb.ex[b * s.ex], w.ex_pose, sizeof(double) * 7);
  if (F) memcpy(&hb.feat[b * s.feat], w.inv_depth, sizeof(double) * F);
  if (M) {
    memcpy(&hb.fhost[b * s.fint], w.factor_host, sizeof(int) * M);
    memcpy(&hb.ftarget[b * s.fint], w.factor_target, sizeof(int) * M);
    memcpy(&hb.ffeat[b * s.fint], w.factor_feature, sizeof(int) * M);
    memcpy(&hb.pts_i[b * s.pts], w.factor_pts_i, sizeof(double) * 3 * M);
    memcpy(&hb.pts_j[b * s.pts], w.factor_pts_j, sizeof(double) * 3 * M);
  }
  static_assert(sizeof(VioPreintegration) == kPreintDoubles * sizeof(double), "VioPreintegration layout");
  memcpy(&hb.preint[b * s.preint], w.preint, sizeof(VioPreintegration) * W);
  h[H_PRIOR_N] = 0, h[H_PRIOR_NB] = 0;
  if (w.prior && w.prior->n > 0) {
    const VioPrior *p = w.prior;
    if (p->n > d.Ncap || p->n_blocks > kMaxPriorBlocks) return VIO_ECAP;
    const bool in_store = !p->block_x0 && !p->linearized_jacobians && !p->linearized_residuals;
    if (in_store ? !store_ok : (!p->block_x0 || !p->linearized_jacobians || !p->linearized_residuals)) return VIO_EINVAL;
    for (int k = 0; k < p->n_blocks; k++) {
      int kind = p->block_kind[k], idx = p->block_index[k], o = p->block_offset[k];
      int ls = kind == VIO_BLOCK_SPEEDBIAS ? 9 : 6;
      if (kind < 0 || kind > 2 || idx < 0 || idx >= P || o < 0 || o + ls > p->n) return VIO_EINVAL;
      hb.pr_kind[b * s.pr_int + k] = kind, hb.pr_index[b * s.pr_int + k] = idx, hb.pr_offset[b * s.pr_int + k] = o;
    }
    // the reduced system keeps the speed-bias x pose coupling of the IMU chain in a compact band and ONE dense row block
    // for the prior (solver_core.h): a prior with more than one speed-bias block has no slot there (no prior made by
    // marginalize() has: it keeps the speed-bias of the new oldest frame only, marginalization_factor.cpp:182-300)
    {
      int nsb = 0;
      for (int k = 0; k < p->n_blocks; k++) nsb += p->block_kind[k] == VIO_BLOCK_SPEEDBIAS;
      if (nsb > 1) return VIO_EINVAL;
    }
    h[H_PRIOR_N] = p->n, h[H_PRIOR_NB] = p->n_blocks;
    if (!in_store) {
      memcpy(&hb.pr_x0[b * s.pr_x0], p->block_x0, sizeof(double) * 9 * p->n_blocks);
      memcpy(&hb.pr_J[b * s.pr_J], p->linearized_jacobians, sizeof(double) * p->n * p->n);
      memcpy(&hb.pr_r[b * s.pr_r], p->linearized_residuals, sizeof(double) * p->n);
    }
  }
  return VIO_OK;
}

// Copies one window's results (host copies of the output arrays) back into the caller's structs.
inline void unpack_window(const BatchStrides &s, int b, const double *out_pose, const double *out_sb,
                          const double *out_feat, const double *raw_pose, const double *raw_sb,
                          const double *raw_feat, const double *out_loop, const double *stats_d, const int *stats_i,
                          VioWindow &w, VioSolveStats *st) {
  const int P = w.window_size + 1, F = w.n_features;
  memcpy(w.pose, out_pose + b * s.out_pose, sizeof(double) * 7 * P);
  memcpy(w.speed_bias, out_sb + b * s.out_sb, sizeof(double) * 9 * P);
  if (F) memcpy(w.inv_depth, out_feat + b * s.out_feat, sizeof(double) * F);
  if (w.raw_pose) memcpy(w.raw_pose, raw_pose + b * s.out_pose, sizeof(double) * 7 * P);
  if (w.raw_speed_bias) memcpy(w.raw_speed_bias, raw_sb + b * s.out_sb, sizeof(double) * 9 * P);
  if (w.raw_inv_depth && F) memcpy(w.raw_inv_depth, raw_feat + b * s.out_feat, sizeof(double) * F);
  bool has_loop = false;
  for (int k = 0; k < w.n_factors; k++)
    if (w.factor_target[k] == P) has_loop = true;
  if (has_loop && w.loop_pose) memcpy(w.loop_pose, out_loop + b * s.out_loop, sizeof(double) * 7);
  if (st) {
    const double *sd = stats_d + b * s.stats_d;
    const int *si = stats_i + b * s.stats_i;
    memset(st, 0, sizeof(*st));
    st->initial_cost = sd[0], st->final_cost = sd[1];
    st->iterations = si[0], st->termination = si[1], st->num_successful_steps = si[2];
    st->num_unsuccessful_steps = si[3];
    for (int i = 0; i < kMaxTrace && i < VIO_MAX_TRACE; i++) {
      st->it_cost[i] = sd[4 + i], st->it_radius[i] = sd[4 + kMaxTrace + i];
      st->it_step_norm[i] = sd[4 + 2 * kMaxTrace + i], st->it_relative_decrease[i] = sd[4 + 3 * kMaxTrace + i];
      st->it_gradient_max_norm[i] = sd[4 + 4 * kMaxTrace + i], st->it_flags[i] = si[4 + i];
    }
  }
}

}  // namespace vio
