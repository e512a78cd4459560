// solver_core.h — the sliding-window solve as single-source SPMD "phase" code.
//
// One workgroup solves one window (VINS::solve_ceres, VINS_ios/VINS.cpp:480-831): factor evaluation, normal
// equations, landmark Schur complement, dense Cholesky of the reduced system, Ceres' trust-region/dogleg loop and
// the new2old gauge fix all run inside one launch with no host round trip. The batch dimension (independent
// sequences) is the grid.
//
// The code is written as barrier-separated parallel-for phases (VIO_PARFOR / VIO_SYNC). Under hipcc it is the
// body of the gfx950 kernel in vio_backend.hip; with -DVIO_EMUL (tests only, never in the product library) the same
// source runs with one "thread" on the host so that index maths and control flow can be debugged without a GPU.
//
// Algorithm notes (what differs from the reference's *route*, not its result):
//  * Ceres materialises a scaled Jacobian J_s = J diag(scale). Everything the minimizer needs is a function of
//    H = J^T J, g = J^T r and the cost, so H and g are accumulated straight from the factors (LDS atomics) and J is
//    never stored: ||J_s[:,c]||^2 = scale_c^2 H_cc, J_s^T r = scale g, |J_s v|^2 = v^T (S H S) v.
//  * Elimination set = all features (1x1 e-blocks); the reduced system is 15(W+1) [+6 loop pose] and lives in LDS
//    as a block-lower matrix of 15x15 blocks (frame i -> [pose 6 | speed-bias 9]).
//  * The Jacobi-scaled system (S H S + mu D^2) y = S g is solved as (H + mu C) z = g, C = D^2 / S^2, y = z / s: no
//    scaling pass over the matrix or the landmark coupling (build_reduced_system).
//  * After H + mu C = L L^T the quadratic forms v^T (S H S + mu D^2) v that the Cauchy point and model_cost_change
//    need are evaluated with u = S v as sum_f E_f (u_f + w_f^T u_p / E_f)^2 + |L^T u_p|^2, so the un-factored matrix
//    is not kept.
//  * IMUFactor's sqrt_info = LLT(cov^-1).L^T (imu_factor.h:72) is recomputed by the reference on every
//    Evaluate; it is constant during a solve, so cov^-1 is formed once and H += J^T cov^-1 J, g += J^T cov^-1 r,
//    cost += r^T cov^-1 r / 2 are used (identical to whitening by any square root of cov^-1).
//  * MarginalizationFactor's Jacobian J0 is constant: H0 = J0^T J0 is formed once per solve.
#pragma once

#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "vio_math.h"

// Three builds of this file:
//   hipcc (product)   the gfx950 kernel body;
//   -DVIO_EMUL        tests only: ONE emulated thread, no barriers, scalar stand-ins for the wave-level sections;
//   -DVIO_SIMT        tests only: the DEVICE sections themselves on the host, every work-item a fiber and every
//                     wave-level instruction (v_mfma, v_readlane, DPP, ballot, s_barrier) evaluated with the hardware's
//                     lane semantics by tests/emul/simt.h (included by the test harness before this file).
#if defined(VIO_EMUL) || defined(VIO_SIMT)
#define VIO_HOST_BUILD 1
#endif
#ifdef VIO_HOST_BUILD
#define VIO_DEV inline
#define VIO_ATOMIC_ADD(p, v) ::vio::atomic_add((p), (v))
#else
#define VIO_DEV __device__ __forceinline__
#define VIO_ATOMIC_ADD(p, v) ::vio::atomic_add((p), (v))
#endif
#if defined(VIO_EMUL)
#define VIO_SYNC() ((void)0)
#define VIO_SYNC_LDS() ((void)0)
#elif defined(VIO_SIMT)
#define VIO_SYNC() __syncthreads()
#define VIO_SYNC_LDS() __syncthreads()
#else
#define VIO_SYNC() __syncthreads()
// Workgroup barrier that only orders LDS traffic: __syncthreads() waits for every outstanding memory operation of the
// wave (s_waitcnt vmcnt(0)), which puts an L2 round trip behind each barrier a global prefetch is meant to cross.
// ONLY where the waves exchange nothing through global memory across the barrier.
#define VIO_SYNC_LDS()                                                      \
  do {                                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");         \
    __builtin_amdgcn_s_barrier();                                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");         \
  } while (0)
#endif
// The thread index every phase starts from passes through an empty asm: LLVM otherwise hoists the per-lane address
// arithmetic of ALL phases out of the trust-region loop (loop-invariant), keeps hundreds of values alive across the whole
// kernel and spills them to scratch -- a scratch_load (L2 latency) where two integer instructions would do.
#ifdef VIO_HOST_BUILD
#define VIO_TID(cx) ((int)(cx).tid)
#else
// (round 6: the index is REBUILT where it is asked for -- lane count of the wave + the wave's first index from a scalar register, two
// vector instructions -- instead of passing the kernel's v0 through the empty asm: the register allocator kept that one value in
// scratch and every phase began with a scratch_load of it, 111 of them in the W = 10 variant)
#define VIO_TID(cx) ::vio::hw_tid((cx).wave64)
#endif
#define VIO_PARFOR(i, n) for (int i = VIO_TID(cx); i < (int)(n); i += (int)cx.nt)
// The same for any per-lane value inside a loop: per-lane addresses that do not depend on the loop counter (the 60 tile
// offsets of a panel step, say) are otherwise computed once, kept alive around the loop and come back as scratch loads.
#ifdef VIO_HOST_BUILD
#define VIO_OPAQUE(x) (x)
#else
#define VIO_OPAQUE(x) ::vio::opaque_tid(x)
#endif

// LDS pointers carry their address space in the type: generic pointers make hipcc emit flat_load/flat_store for every
// LDS access (no ds_read/ds_write at all in the first version of this kernel), which is several times slower.
#ifdef VIO_HOST_BUILD
#define VIO_AS3
#else
#define VIO_AS3 __attribute__((address_space(3)))
#endif

namespace vio {

#ifndef VIO_HOST_BUILD
__device__ __forceinline__ int opaque_tid(int t) {
  asm volatile("" : "+v"(t));
  return t;
}
__device__ __forceinline__ int hw_tid(int wave64) {
  int t;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %1\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(t) : "s"(wave64));
  return t;
}
#endif

typedef VIO_AS3 double *ldsd;
typedef const VIO_AS3 double *cldsd;
typedef VIO_AS3 int *ldsi;

#ifdef VIO_HOST_BUILD
inline void atomic_add(double *p, double v) { *p += v; }
inline void atomic_add_noret(double *p, double v) { *p += v; }
#else
// Global memory that several workgroups of ONE window may add to (cooperative windows): device scope. No return value: the
// instruction is fire-and-forget (global_atomic_add_f64 at the L2).
__device__ __forceinline__ void atomic_add_noret(double *p, double v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_add(ldsd p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void atomic_add(double *p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // only this workgroup touches its scratch
}
#endif

constexpr int kBS = 15;          // unknowns per frame in the pose-side vectors: pose 6 + speed-bias 9 (frame-major)
constexpr int kSB = 9;           // speed-bias block
constexpr int kSS = kSB * kSB;   // 81
constexpr int kAW = 18;          // pose columns an IMU chain couples a speed-bias block to: frames k-1, k, k+1
constexpr int kAS = kSB * kAW;   // 162
constexpr int kPreintDoubles = 467;
constexpr int kMaxTrace = 64;
constexpr int kStatsDoubles = 4 + 5 * kMaxTrace;  // initial, final, (it_cost, radius, step_norm, rel, gmax)[64]
constexpr int kStatsInts = 4 + kMaxTrace;         // iterations, termination, n_ok, n_bad, flags[64]

// Per-stage cycle counters (the kernel-side counterpart of the reference's TS/TE timers, global_param.hpp:85-92).
enum Stage {
  ST_SETUP_IMU = 0, ST_SETUP_PRIOR, ST_EVAL_PRIOR, ST_EVAL_IMU, ST_EVAL_PROJ, ST_SCALE, ST_SCHUR, ST_RHS, ST_CHOL,
  ST_TRISOLVE, ST_QUADFORM, ST_DOGLEG, ST_COST_EVAL, ST_NEW2OLD, ST_MARG_BUILD, ST_MARG_CHOL, ST_TOTAL,
  // finer attribution (sub-stages; their cycles are NOT included in the stages above)
  ST_P_ZERO, ST_P_FACT, ST_P_GRAM, ST_P_FEAT,      // projection factors: zeroing, evaluation+staging, Gram, per-feature
  ST_C_POTRF, ST_C_TRSM,                           // Cholesky: first diagonal block, panel
  ST_IMU_RAW,                                      // raw IMU residual/Jacobian (one thread per factor)
  ST_C_WAIT, ST_Q_W, ST_BACKSOLVE, ST_C_AHEAD,     // trailing-update wait, W part of quad_form, back-substitution, wave-0 look-ahead
  ST_TR_VEC,                                       // trust-region vector phase before the linear solve
  ST_M_PRIOR, ST_M_IMU, ST_M_FACT, ST_M_GRAM,      // marginalization: prior/setup, IMU factor, factor staging, Gram
  ST_D0, ST_D1, ST_D2, ST_D3, ST_D4, ST_D5,         // free slots for timing experiments (VIO_AMD_PROF_TID picks the clock's lane)
  // round 6: the O(n) vector phases of a trust-region iteration, one slot per barrier interval
  ST_V_GD,                                         // |g_d|^2, Cauchy direction, quad_form vectors (one pass + reduction)
  ST_V_DOT, ST_V_STEP,                             // dogleg: the two inner products (+ reduction); step -> t2 / tf (+ barrier)
  ST_V_PLUS,                                       // Plus in place, stash of the iterate, norms (+ reduction)
  ST_V_GMAX,                                       // gradient max norm + the iteration record
  ST_V_REST,                                       // restore after a rejected step
  ST_B_INIT, ST_B_POSE, ST_B_ASP, ST_B_BAND, ST_B_GN,  // backsolve: copies; pose tiles; A_sp z_p; band chains (+ landmarks); GN step
  ST_E_HEAD, ST_E_COPY, ST_E_H0DX, ST_E_TAIL,      // evaluate(jac): rotations + zeroing; AppPr + PP copy (+ raw IMU); H0 dx + diagonal blocks; diag / scaling tail
  ST_X0, ST_X1, ST_X2, ST_X3,
  ST_COUNT = 64                                    // (last slot = time of the previous stamp)
};

struct Ctx {
  int tid, nt;
  int wave64 = 0;     // index of the wave's first work-item (device: a scalar register, VIO_TID builds the index from it)
  ldsd red;           // LDS scratch for block reductions: two halves of [3 nt/64]
  mutable int red_phase = 0;  // which half the next reduction writes (uniform across the block)
  long long *prof;    // global [ST_COUNT] cycle accumulators of this window, or null; prof[ST_COUNT-1] = last stamp
  int wrot = 0;       // rotation of the wave ROLES in the serial phases: role = (wave + wrot) mod waves, role 0 runs the pivot
                      // chains. Two workgroups share a CU and the hardware puts wave w of both on the same SIMD: with the
                      // same roles the two chains would fight over one SIMD's matrix pipe while three stand idle.
  int prof_tid = 0;   // the work-item that keeps the stage clock (0; another wave's first lane to time what wave 0 does not run)
  VIO_AS3 long long *lprof;  // the same counters while the kernel runs (LDS); copied to prof at the end
  // cooperative windows (round 5): `coop` workgroups serve one window; member 0 owns the solve, the others wait for commands
  int coop = 1, member = 0;
  unsigned coop_spin = 1u << 24;  // polls before a wait between the workgroups of a window gives up (BatchPtrs::coop_spin)
  mutable unsigned coop_seq = 0;  // commands issued (owner) / served (helper) so far: the same value in every work-item
};

// Charges the cycles since the previous stamp to `stage` (thread 0 only; call between barriers).
VIO_DEV void stamp(const Ctx &cx, int stage) {
#if !defined(VIO_EMUL) && !defined(VIO_NO_STAMPS)
  if (cx.prof && cx.tid == cx.prof_tid) {  // accumulators live in LDS (a global read-modify-write per stamp costs ~3k cycles)
    long long t = clock64();
    cx.lprof[stage] += t - cx.lprof[ST_COUNT - 1];
    cx.lprof[ST_COUNT - 1] = t;
  }
#else
  (void)cx, (void)stage;
#endif
}

// Per-window view of the packed batch (all pointers device-global unless noted).
struct WinView {
  int W, P, F, M, np, nblk, has_loop, loop_frame, marg_flag, max_iter;
  int prior_n, prior_nb;
  int Fpad;      // F rounded up to a multiple of 8
  int npose6;    // 6 * (P + has_loop)
  double s_info, gravity, cauchy_b;
  const double *pose0, *sb0, *ex, *feat0;
  const int *fhost, *ftarget, *ffeat;
  const int *fslot;                     // factor -> slot in the (host,target)-bucketed, even-padded staging order
  const int *fstart;                    // [F+1] factor range of every feature
  const int *pair_h, *pair_t, *pair_s0, *pair_s1;  // [npairs] bucket -> frames and slot range
  int npairs, nslots, n6cap;
  int nrev;      // number of factors whose target frame precedes their host (never produced by the reference's factor list)
  const double *pts_i, *pts_j;
  const double *preint;
  const int *pr_kind, *pr_index, *pr_offset;
  const double *pr_x0, *pr_J, *pr_r;
  int use_origin;
  double origin_yaw, origin_p[3];
  int Pcap, Fcap, nblk_cap;  // capacities of the batch (the layout of the cooperative payload depends on them)
  // scratch
  double *imu_info;  // [W][225]  sym(cov^-1)
  double *imu_aug;   // [W][15*30] Gauss-Jordan work area
  double *imu_J;     // [W][15*30]
  double *imu_M;     // [W][15*30]
  double *imu_r;     // [W][15]
  double *imu_Mr;    // [W][15]
  double *prb0;      // [n]    b0 = J0^T r0
  double *prH0;      // [n*n]  J0^T J0
  double *Apri;      // [9][jp]  the prior's speed-bias x pose block A(s_kpr[c], pose index j): constant during a solve (H0),
                     //          written once by setup_prior; the IMU part of that coupling lives in LDS (WorkT::AspI)
  int n6, nrows, nT, jp;  // pose unknowns 6 (P + has_loop); rows of the pose matrix (n6 + the carried right-hand side);
                          // its 16-row tiles; leading dimension of Asp rows (16 nT)
  double *WTf;       // [F][n6cap]      H_fp feature-major: row = feature, col = 6*frame + c (the marginalization phase reuses it
                     //                 with the extrinsic in the column group of the relocalization pose, marg_core.h)
  int *sfact;        // staging slot -> factor index (-1: unused tail slot of an odd bucket), built once per solve
  int *srec_i;       // staging slot -> host | target << 8 | landmark << 16 (-1: unused slot), built once per solve: the factor
  int *gpiece;       // Gram pieces in slot order, built once per solve: slot offset in its chunk | slots << 10 | host << 15 | target << 21
  int *gstart;       // [chunks + 1] first piece of every staging chunk
  double *srec_d;    // data in SLOT order ([slot][6] = pts_i, pts_j), one global round trip per evaluation pass instead of two
  double *PP;        // off-diagonal pose-pose blocks of the projection Gram products IN THE LAYOUT OF App (tri_at; zero where no
                     // (host, target) bucket writes: zeroed once per solve), so that a linearization starts the pose matrix as
                     // AppPr + PP with two independent loads per element and no bucket descriptors
  double *AppPr;     // the prior's H0 scattered into the layout of App | Dss | Css once per solve (setup_prior): every
                     // linearization starts the reduced matrix as a straight copy of it instead of an element-wise scatter
  double *AspG;      // [P][9][18] the IMU part of the speed-bias x pose coupling when the pose matrix is global (WorkT::AspI)
  double *coop;      // cooperative windows: flags + payload shared by the workgroups of this window (CoopLayout)
  double *stash;     // iterate and the vectors of its linearization while the candidate is evaluated in their place (minimize):
                     // pose 7 (P + 1) | sb 9 P | feat F | gp dp gnp (nblk 15 each) | gf hff gnf (F each); read back after a rejected step
  // outputs
  double *out_pose, *out_sb, *out_feat, *raw_pose, *raw_sb, *raw_feat, *out_loop;
  double *stats_d;
  int *stats_i;
};

// =====================================================================================================
// Storage of the reduced system
// =====================================================================================================
// With the landmarks eliminated the unknowns are the poses (6 each, plus the relocalization pose) and one speed-bias
// block (9) per frame. Pose x pose is dense (landmark Schur complement, prior), but a speed-bias block only couples to
// its own and the neighbouring frame (IMU chain) and, for the one the prior keeps, to the prior's poses. The first two
// versions of this kernel stored the reduced system as a dense 15(W+1) matrix: 118.8 KB of LDS at W = 10, one workgroup
// per CU. Here the speed-bias blocks are ordered ahead of the poses and eliminated from the newest frame to the oldest
// (Ceres itself moves speed-bias blocks into its e-set, CSI/reorder_program.cc:446-541): a block-tridiagonal band whose
// fill into the pose columns is consumed as it is produced and never stored.
//   App   pose x pose, nrows = n6 + 1: row n6 carries the right-hand side through the factorization (its row of L is the
//         forward-substituted y_p). Lower triangle by 16-row tiles: tile row I = rows [16 I, 16 I + 16), 16 (I + 1) columns
//         each, tile rows one after the other (21.9 KB at W = 10 instead of 34.8 KB for the square).
//   Dss   [P][9][9]  diagonal speed-bias blocks (lower triangle read); factored: L below / on, L^-1 transposed above the
//         diagonal, 1 / L_cc in ldinv
//   Css   [P][9][9]  Css[k] = A(s_{k-1}, s_k), k >= 1; factored: E_k = L(s_{k-1}, s_k)
//   AspI  [P][9][18]  A(s_k[c], pose index alo_k + jj), alo_k = 6 max(k - 1, 0), of the UNFACTORED system: what the IMU
//         chain couples a speed-bias block to (frames k-1, k, k+1). The speed-bias block the prior keeps also couples to
//         every pose of the prior: that block is constant during a solve and stays in global memory (WinView::Apri).
// Pose index a = 6 frame + c; the pose-side VECTORS stay frame-major (15 frame + c, speed-bias at + 6).
// (Row lengths are ODD, 16 (I + 1) + 1: the operand fetch of a matrix instruction reads 16 rows at the same column, and
// with a row length that is a multiple of 16 doubles all of them fall into two LDS banks.)
VIO_HD int tri_ld(int I) { return 16 * (I + 1) + 1; }
VIO_HD int tri_off(int I) { return 128 * I * (I + 1) + 16 * I; }  // sum_{J < I} 16 (16 (J + 1) + 1)
VIO_HD int tri_at(int r, int c) {  // c < 16 ((r >> 4) + 1)
  const int I = r >> 4;
  return tri_off(I) + (r & 15) * tri_ld(I) + c;
}
VIO_HD size_t tri_doubles(int nrows) {
  const int nT = (nrows + 15) >> 4;
  return (size_t)tri_off(nT - 1) + (size_t)(nrows - 16 * (nT - 1)) * tri_ld(nT - 1);
}

// LDS (or emulated) working set; all arrays sized by the launcher from the dims. MP is the pointer type of the pose
// matrix: LDS when it fits (ldsd), else global (double *); AP that of the IMU part of the speed-bias x pose coupling,
// which follows the matrix into LDS unless the landmark arrays need the room (BatchDims::lds_asp).
template <class MP, class AP = MP>
struct WorkT {
  typedef MP mat_ptr;
  MP App;         // pose x pose (+ the carried right-hand side row), tile-row packed lower triangle
  ldsd stage;     // LDS staging area of the Jacobian rows (and of the set-up's J0 / pivot mailboxes): App itself when the matrix is
  int nstage;     // in LDS (doubles from App on that are free whenever the reduced matrix is not assembled), else the band + the
                  // fill-tile buffer + spare LDS (carve_all)
  ldsd Dss, Css;  // speed-bias band: P blocks of 81 each, Css = Dss + 81 P (contiguous with App when App is in LDS)
  AP AspI;        // [P][9][18]: behind Css in LDS, or in the window's global scratch (WinView::AspG)
  ldsd aspring;   // AspI in global scratch next to an LDS pose matrix: two blocks of it in LDS, refilled by the chain wave
  bool asp_ring;  // one slot ahead of the panel waves (factor_band_regs)
  bool asp_lds;   // AspI itself is in LDS (16 doubles of padding behind it: the zero the panel steps read through clamped addresses)
  ldsd xpose, xsb, xfeat;   // current iterate: (P+1)*7, P*9, F
  ldsd cpose, csb, cfeat;   // candidate
  ldsd ex;                  // 7
  ldsd gp, gf;              // unscaled gradient J^T r: np, F
  ldsd sp, sf;              // Jacobi scaling
  ldsd dp;                  // dogleg diagonal (poses; landmarks: feat_d); the scaled gradient g s / d is recomputed
  ldsd gnp, gnf;            // Gauss-Newton step in d-scaled space
  ldsd stp, stf;            // trust-region step (J_s coordinates), later delta
  ldsd hff;                 // H_ff (unscaled)
  ldsd ef, einv;            // landmark scratch / 1 / E_f
  ldsd ldinv;               // 1 / L_cc: speed-bias blocks [9 P], then the pose matrix [16 nT]
  ldsd t1, t2;              // np temporaries
  ldsd xt;                  // [16 nT] pose-index work vector of the back-substitution
  ldsd tf;                  // F temporary
  ldsd prdx, prr;           // prior dx / residual: prior_n each
  ldsi prcol;               // prior column -> (frame << 8 | component 0..14) of the reduced system (-1 constant): prior_n
  ldsi sbr;                 // [2 P]: first pose column speed-bias block k couples to; 1 if it is the block the prior keeps
  ldsi flag;                // [4] block-uniform flags
  ldsi ready;               // [P] band block k is factored (L_k, L_k^-1, E_k in place): set by the chain wave, polled by the panel waves
  ldsd park;                // [24] loop-carried scalars of the minimizer while the linear solve runs
  ldsi fh;                  // F: host frame of every feature (-1: it has no factor)
  ldsd rot;                 // (P+2) x 9: rotation matrices of the poses under evaluation, then r_ic
  ldsd ppd;                 // (P+1) x 36: diagonal pose-pose blocks of the projection Gram products
  ldsd vbuf;                // general panel path only: [nT][3][64] fill tiles V_k^T (else null)
  double *VG;               // pose matrix in global scratch only: the fill panels of all band blocks, [P][nT][3][64] (batch.h, s.hm)
};

VIO_DEV int off_pose(const WinView &v, int i) { return kBS * i; }  // loop pose: i == P -> 15 P
VIO_DEV int off_sb(int i) { return kBS * i + 6; }

// Element of the UNFACTORED reduced system: row (frame fr, component cr in 0..14), column (fc, cc), row >= column in
// frame-major order. add: accumulate atomically, else plain store. Speed-bias blocks more than one frame apart have no
// slot (no factor of the reference couples them; pack_window refuses priors that would).
template <class WK>
VIO_DEV void red_put(const WinView &v, WK &w, int fr, int cr, int fc, int cc, double val, bool add) {
  if (cr < 6 && cc < 6) {
    auto p = w.App + tri_at(6 * fr + cr, 6 * fc + cc);
    if (add) VIO_ATOMIC_ADD(p, val);
    else *p = val;
  } else if (cr >= 6 && cc >= 6) {
    auto p = fr == fc ? w.Dss + fr * kSS + (cr - 6) * kSB + (cc - 6) : w.Css + fr * kSS + (cc - 6) * kSB + (cr - 6);
    if (fr - fc > 1) return;
    if (add) VIO_ATOMIC_ADD(p, val);
    else *p = val;
  } else {  // speed-bias x pose: the IMU chain only (the prior's part is constant: WinView::Apri)
    const int k = cr >= 6 ? fr : fc, c = (cr >= 6 ? cr : cc) - 6, j = cr >= 6 ? 6 * fc + cc : 6 * fr + cr;
    auto p = w.AspI + (k * kSB + c) * kAW + j - 6 * (k > 0 ? k - 1 : 0);
    if (add) VIO_ATOMIC_ADD(p, val);
    else *p = val;
  }
}

// =====================================================================================================
// Cooperative windows: several workgroups per window (round 5)
// =====================================================================================================
// A launch of few large windows (W = 20 / 30: one 512-thread workgroup per window and CU) leaves most of the chip idle while
// each window spends half its time in phases that are parallel over factors or tiles. With `coop` > 1 a window gets that
// many workgroups (all resident at once: the launcher only asks for it when windows x coop fits the CUs, and maps the
// members of a window to one XCD so that they meet in one L2). Member 0 (the owner) runs the solve as before; at a parallel
// phase it publishes the phase's inputs in the window's scratch, posts a command, takes its own share, waits for the others
// and merges. The helpers idle on the command word between phases (s_sleep). Shared phases: the projection factors of a
// linearization (Jacobian-row chunks round-robin over the members; partial sums merged by the owner) and the landmark Schur
// complement of the general path (tile pairs round-robin; every tile of App -- global scratch in this variant -- has one
// writer). Everything serial (band / pose factorization, trust-region logic, marginalization) stays with the owner.
// Ordering: payload stores, workgroup barrier, device-scope fence, flag store by one lane | flag load by one lane, barrier,
// device-scope fence by EVERY wave (their vector L1 may hold the previous round's lines), payload loads.
enum CoopCmd { COOP_EXIT = 1, COOP_EVAL = 2, COOP_SCHUR = 3, COOP_SYRK = 4 };
constexpr unsigned kCoopSpinLimit = 1u << 24;  // polls (s_sleep 8 between them) before a wait gives up: seconds (Ctx::coop_spin)
constexpr int kCoopMax = 4;
struct CoopLayout {
  size_t o_pose, o_feat, o_ex, o_einv, o_tf, o_part, part, total;  // (doubles; [0, 8) are the flag words)
  size_t p_gp, p_ppd, p_f, p_cost;                                 // inside one helper's partial record
  static VIO_HD CoopLayout make(int Pcap, int Fcap, int nblk_cap) {
    CoopLayout L;
    auto up = [](size_t n) { return (n + 7) & ~(size_t)7; };
    size_t o = 8;
    L.o_pose = o, o += up(7 * (size_t)(Pcap + 1));
    L.o_feat = o, o += up((size_t)Fcap);
    L.o_ex = o, o += 8;
    L.o_einv = o, o += up((size_t)Fcap);
    L.o_tf = o, o += up((size_t)Fcap);
    size_t q = 0;
    L.p_gp = q, q += up((size_t)nblk_cap * kBS);
    L.p_ppd = q, q += up(36 * (size_t)(Pcap + 1));
    L.p_f = q, q += 8 * up((size_t)Fcap);  // hff, gf, six host-coupling accumulators
    L.p_cost = q, q += 8;
    L.part = q, L.o_part = o, o += (kCoopMax - 1) * q;
    L.total = o;
    return L;
  }
};
#ifndef VIO_HOST_BUILD
__device__ __forceinline__ unsigned *coop_flags(const WinView &v) { return reinterpret_cast<unsigned *>(v.coop); }
// (a wait that never ends would hang the device: after ~2^24 polls the error word is set and everybody moves on -- the solve
// is then wrong and says so in its termination code)
// (once the error word is set every later wait of the window returns at once: one timeout, not one per phase)
__device__ __forceinline__ void coop_spin(unsigned *word, unsigned want_at_least, unsigned *err, bool exact_seq, unsigned limit) {
  unsigned n = 0;
  for (;;) {
    const unsigned x = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (exact_seq ? (x >> 4) >= want_at_least : x >= want_at_least) break;
    if ((n & 63u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    __builtin_amdgcn_s_sleep(8);
    if (++n > limit) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}
// owner: every work-item has stored its part of the payload -> post command `code`
VIO_DEV void coop_post(const Ctx &cx, const WinView &v, int code) {
  __syncthreads();
  cx.coop_seq++;
  if (cx.tid == 0) {
    __threadfence();
    __hip_atomic_store(coop_flags(v), (cx.coop_seq << 4) | (unsigned)code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// owner: until every helper has completed the commands posted so far; then their stores are visible to every work-item
VIO_DEV void coop_wait_helpers(const Ctx &cx, const WinView &v) {
  if (cx.tid == 0) coop_spin(coop_flags(v) + 1, cx.coop_seq * (unsigned)(cx.coop - 1), coop_flags(v) + 2, false, cx.coop_spin);
  __syncthreads();
  __threadfence();
}
// helper: the next command (its inputs are visible to every work-item on return)
VIO_DEV int coop_wait_cmd(const Ctx &cx, const WinView &v) {
  cx.coop_seq++;
  if (cx.tid == 0) coop_spin(coop_flags(v), cx.coop_seq, coop_flags(v) + 2, true, cx.coop_spin);
  __syncthreads();
  __threadfence();
  // (a wait that gave up -- here or in another workgroup of the window -- reads as COOP_EXIT: the helper goes home)
  if (__hip_atomic_load(coop_flags(v) + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return COOP_EXIT;
  return (int)(__hip_atomic_load(coop_flags(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 15u);
}
// helper: every work-item has stored its results -> count this workgroup as done with the command
VIO_DEV void coop_done(const Ctx &cx, const WinView &v) {
  __syncthreads();
  if (cx.tid == 0) {
    __threadfence();
    __hip_atomic_fetch_add(coop_flags(v) + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
#endif

// ---- block reductions: every thread receives the same value ------------------------------------------------------
// Wave level on the DPP network (row_shr 1/2/4/8, row_bcast 15/31; total in lane 63, broadcast through SGPRs), block
// level through cx.red. Two alternating halves of cx.red make ONE barrier per reduction enough: a half is rewritten
// two calls later, and the barrier of the call in between orders that against its last readers.
#ifndef VIO_EMUL
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_move_f64(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_move_f64<0x111, 0xf>(v);
  v += dpp_move_f64<0x112, 0xf>(v);
  v += dpp_move_f64<0x114, 0xf>(v);
  v += dpp_move_f64<0x118, 0xf>(v);
  v += dpp_move_f64<0x142, 0xa>(v);
  v += dpp_move_f64<0x143, 0xc>(v);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_max_f64(double v) {  // operands >= 0 (norms, flags): a shifted-in 0 is neutral
  v = fmax(v, dpp_move_f64<0x111, 0xf>(v));
  v = fmax(v, dpp_move_f64<0x112, 0xf>(v));
  v = fmax(v, dpp_move_f64<0x114, 0xf>(v));
  v = fmax(v, dpp_move_f64<0x118, 0xf>(v));
  v = fmax(v, dpp_move_f64<0x142, 0xa>(v));
  v = fmax(v, dpp_move_f64<0x143, 0xc>(v));
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
#endif
VIO_DEV double block_sum(const Ctx &cx, double v) {
#ifdef VIO_EMUL
  return v;
#else
  const int nw = cx.nt >> 6;
  ldsd red = cx.red + (cx.red_phase ^= 1) * 3 * nw;
  v = wave_sum_f64(v);
  if ((cx.tid & 63) == 0) red[cx.tid >> 6] = v;
  VIO_SYNC();
  double s = 0;
  for (int w = 0; w < nw; w++) s += red[w];
  return s;
#endif
}
// Three sums with one barrier.
VIO_DEV void block_sum3(const Ctx &cx, double &a, double &b, double &c) {
#ifndef VIO_EMUL
  const int nw = cx.nt >> 6;
  ldsd red = cx.red + (cx.red_phase ^= 1) * 3 * nw;
  a = wave_sum_f64(a), b = wave_sum_f64(b), c = wave_sum_f64(c);
  if ((cx.tid & 63) == 0) red[cx.tid >> 6] = a, red[nw + (cx.tid >> 6)] = b, red[2 * nw + (cx.tid >> 6)] = c;
  VIO_SYNC();
  a = b = c = 0;
  for (int w = 0; w < nw; w++) a += red[w], b += red[nw + w], c += red[2 * nw + w];
#else
  (void)cx, (void)a, (void)b, (void)c;
#endif
}
// Maximum of non-negative values.
VIO_DEV double block_max(const Ctx &cx, double v) {
#ifdef VIO_EMUL
  return v;
#else
  const int nw = cx.nt >> 6;
  ldsd red = cx.red + (cx.red_phase ^= 1) * 3 * nw;
  v = wave_max_f64(v);
  if ((cx.tid & 63) == 0) red[cx.tid >> 6] = v;
  VIO_SYNC();
  double s = red[0];
  for (int w = 1; w < nw; w++) s = fmax(s, red[w]);
  return s;
#endif
}

// Issue priority of the wave that walks a serial chain (pivots, band substitutions): with two windows per CU its SIMD is shared
// with a wave of the other window, and every cycle that wave wins the issue slot is a cycle on this window's critical path.
#if defined(VIO_HOST_BUILD)
#define VIO_PRIO(n) ((void)0)
#else
#define VIO_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
// A flag in LDS that one wave posts and others poll (factor_band_regs)
#if defined(VIO_HOST_BUILD)
#define VIO_FLAG_STORE(p, x) (*(volatile int *)(p) = (x))
#define VIO_FLAG_LOAD(p) (*(volatile int *)(p))
#else
#define VIO_FLAG_STORE(p, x) __hip_atomic_store((p), (x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define VIO_FLAG_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif
#if defined(VIO_HOST_BUILD)
#define VIO_SCHED_FENCE() ((void)0)
#else
#define VIO_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// W lives in global memory (L2): a dependent load costs ~3k cycles here, so the mat-vecs with W fetch a whole strip
// (up to kWStrip entries, predicated) before the first multiply. Strided form: column f of WT (stride Fpad).
constexpr int kWStrip = 24;
VIO_DEV void wt_strip_load(const double *p, size_t stride, int n, double x[kWStrip]) {
#pragma unroll
  for (int j = 0; j < kWStrip; j++) x[j] = p[(size_t)(j < n ? j : 0) * stride];
#ifndef VIO_EMUL
  __builtin_amdgcn_sched_barrier(0);
#endif
}
// Splits n6 rows over the threads available per feature: (feature, part) items, <= kWStrip rows per part.
VIO_DEV void wt_parts(int nt, int F, int n6, int &nparts, int &per) {
  nparts = nt / (F > 0 ? F : 1);
  const int need = (n6 + kWStrip - 1) / kWStrip;
  nparts = nparts < need ? need : (nparts > 6 ? 6 : nparts);
  per = (n6 + nparts - 1) / nparts;
}

// y[c] += sum_k M[k][c] x[k] for a dense n x n matrix in global memory (the prior's J0 / J0^T): (column, part) items with
// neighbouring lanes on neighbouring columns (coalesced), a strip of k fetched at once, partial sums through add(c, s).
template <class XP, class ADD>
VIO_DEV void dense_matvec_cols(const Ctx &cx, const double *M, int n, XP x, ADD add) {
  constexpr int kStrip = 32;  // rows an item fetches at once: 75 prior rows over 3 parts are ONE round trip per item
  int nparts = (int)cx.nt / (n > 0 ? n : 1);
  const int need = (n + 4 * kStrip - 1) / (4 * kStrip);
  nparts = nparts < need ? need : (nparts > 8 ? 8 : nparts);
  const int per = (n + nparts - 1) / nparts;
  VIO_PARFOR(q, n * nparts) {
    const int part = q / n, c = q - part * n;
    const int k0 = part * per, k1 = k0 + per < n ? k0 + per : n;
    double xs[kStrip], s = 0;
    for (int b0 = k0; b0 < k1; b0 += kStrip) {
      const int nb = k1 - b0 < kStrip ? k1 - b0 : kStrip;
#pragma unroll
      for (int j = 0; j < kStrip; j++) xs[j] = M[(size_t)(b0 + (j < nb ? j : 0)) * n + c];
      VIO_SCHED_FENCE();
#pragma unroll
      for (int j = 0; j < kStrip; j++) s += (j < nb ? xs[j] : 0.0) * x[b0 + (j < nb ? j : 0)];
    }
    if (k1 > k0) add(c, s);
  }
}

// ProjectionFactor::Evaluate (projection_facor.cpp:16-99) in local coordinates. Jex optional.
template <class PA, class PE>
VIO_DEV void projection_eval(double s_info, PA pose_i, PA pose_j, PE ex, double inv_dep, const double *pts_i,
                             const double *pts_j, bool jac, double *r, double *Ji, double *Jj, double *Jex, double *Jl) {
  Quat Qi{pose_i[3], pose_i[4], pose_i[5], pose_i[6]}, Qj{pose_j[3], pose_j[4], pose_j[5], pose_j[6]},
      qic{ex[3], ex[4], ex[5], ex[6]};
  double pc_i[3] = {pts_i[0] / inv_dep, pts_i[1] / inv_dep, pts_i[2] / inv_dep};
  double t[3], p_imu_i[3], p_w[3], p_imu_j[3], p_c_j[3];
  qrot(qic, pc_i, t);
  for (int k = 0; k < 3; k++) p_imu_i[k] = t[k] + ex[k];
  qrot(Qi, p_imu_i, t);
  for (int k = 0; k < 3; k++) p_w[k] = t[k] + pose_i[k];
  double d[3] = {p_w[0] - pose_j[0], p_w[1] - pose_j[1], p_w[2] - pose_j[2]};
  qrot(qinv(Qj), d, p_imu_j);
  double e[3] = {p_imu_j[0] - ex[0], p_imu_j[1] - ex[1], p_imu_j[2] - ex[2]};
  qrot(qinv(qic), e, p_c_j);
  double dep_j = p_c_j[2];
  r[0] = s_info * (p_c_j[0] / dep_j - pts_j[0]);
  r[1] = s_info * (p_c_j[1] / dep_j - pts_j[1]);
  if (!jac) return;
  double Ri[9], Rj[9], ric[9], ricT[9], RjT[9];
  qtoR(Qi, Ri), qtoR(Qj, Rj), qtoR(qic, ric);
  mat3T(ric, ricT), mat3T(Rj, RjT);
  double red[6] = {s_info * (1. / dep_j), 0, s_info * (-p_c_j[0] / (dep_j * dep_j)),
                   0, s_info * (1. / dep_j), s_info * (-p_c_j[1] / (dep_j * dep_j))};
  double A[9], B[9], S[9], T9[9];
  mat3mul(ricT, RjT, A);
  mat3mul(A, Ri, B);
  skew3(p_imu_i, S);
  mat3mul(B, S, T9);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++) {
      Ji[i * 6 + j] = red[i * 3] * A[j] + red[i * 3 + 1] * A[3 + j] + red[i * 3 + 2] * A[6 + j];
      Ji[i * 6 + 3 + j] = -(red[i * 3] * T9[j] + red[i * 3 + 1] * T9[3 + j] + red[i * 3 + 2] * T9[6 + j]);
    }
  skew3(p_imu_j, S);
  mat3mul(ricT, S, T9);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++) {
      Jj[i * 6 + j] = -(red[i * 3] * A[j] + red[i * 3 + 1] * A[3 + j] + red[i * 3 + 2] * A[6 + j]);
      Jj[i * 6 + 3 + j] = red[i * 3] * T9[j] + red[i * 3 + 1] * T9[3 + j] + red[i * 3 + 2] * T9[6 + j];
    }
  double Bric[9], v3[3];
  mat3mul(B, ric, Bric);
  if (Jex) {  // projection_facor.cpp:76-86 (the extrinsic is a kept block of the marginalization prior)
    double RjTRi[9], M1[9], C1[9], Spc[9], T1[9], v[3], Sv[9], u[3], w3[3], Sw[9];
    mat3mul(RjT, Ri, RjTRi);
    for (int i = 0; i < 9; i++) M1[i] = RjTRi[i] - (i % 4 == 0);
    mat3mul(ricT, M1, C1);
    skew3(pc_i, Spc);
    mat3mul(Bric, Spc, T1);
    mat3vec(Bric, pc_i, v);
    skew3(v, Sv);
    double exv[3] = {ex[0], ex[1], ex[2]};
    mat3vec(Ri, exv, u);
    for (int k = 0; k < 3; k++) u[k] += pose_i[k] - pose_j[k];
    mat3vec(RjT, u, w3);
    for (int k = 0; k < 3; k++) w3[k] -= ex[k];
    mat3vec(ricT, w3, u);
    skew3(u, Sw);
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) {
        Jex[i * 6 + j] = red[i * 3] * C1[j] + red[i * 3 + 1] * C1[3 + j] + red[i * 3 + 2] * C1[6 + j];
        double c0 = -T1[j] + Sv[j] + Sw[j], c1 = -T1[3 + j] + Sv[3 + j] + Sw[3 + j],
               c2 = -T1[6 + j] + Sv[6 + j] + Sw[6 + j];
        Jex[i * 6 + 3 + j] = red[i * 3] * c0 + red[i * 3 + 1] * c1 + red[i * 3 + 2] * c2;
      }
  }
  mat3vec(Bric, pts_i, v3);
  for (int i = 0; i < 2; i++)
    Jl[i] = (red[i * 3] * v3[0] + red[i * 3 + 1] * v3[1] + red[i * 3 + 2] * v3[2]) * -1.0 / (inv_dep * inv_dep);
}

// The same factor from rotation MATRICES prepared once per evaluation (w.rot: one per frame, then r_ic), for the
// solver's inner loops: four mat-vecs for the residual and, with jac, (reduce A) chained through R_i and r_ic instead
// of five dense 3x3 products. Equal to projection_eval up to rounding (different association of the same products).
template <class PR, class PA, class PE>
VIO_DEV void projection_eval_rot(double s_info, PR Ri_, PA pose_i, PR Rj_, PA pose_j, PR ric_, PE ex, double inv_dep,
                                 const double *pts_i, const double *pts_j, bool jac, double *r, double *Ji, double *Jj,
                                 double *Jl) {
  double Ri[9], Rj[9], ric[9];
#pragma unroll
  for (int k = 0; k < 9; k++) Ri[k] = Ri_[k], Rj[k] = Rj_[k], ric[k] = ric_[k];
  const double tic[3] = {ex[0], ex[1], ex[2]};
  const double pi3[3] = {pts_i[0], pts_i[1], pts_i[2]};
  const double dep_i = rcp_f(inv_dep);
  const double pc_i[3] = {pi3[0] * dep_i, pi3[1] * dep_i, pi3[2] * dep_i};
  double p_imu_i[3], p_w[3], p_imu_j[3], p_c_j[3], d[3], e[3];
  for (int a = 0; a < 3; a++) p_imu_i[a] = ric[3 * a] * pc_i[0] + ric[3 * a + 1] * pc_i[1] + ric[3 * a + 2] * pc_i[2] + tic[a];
  for (int a = 0; a < 3; a++) p_w[a] = Ri[3 * a] * p_imu_i[0] + Ri[3 * a + 1] * p_imu_i[1] + Ri[3 * a + 2] * p_imu_i[2] + pose_i[a];
  for (int a = 0; a < 3; a++) d[a] = p_w[a] - pose_j[a];
  for (int a = 0; a < 3; a++) p_imu_j[a] = Rj[a] * d[0] + Rj[3 + a] * d[1] + Rj[6 + a] * d[2];  // R_j^T d
  for (int a = 0; a < 3; a++) e[a] = p_imu_j[a] - tic[a];
  for (int a = 0; a < 3; a++) p_c_j[a] = ric[a] * e[0] + ric[3 + a] * e[1] + ric[6 + a] * e[2];  // r_ic^T e
  const double idep_j = rcp_f(p_c_j[2]);
  r[0] = s_info * (p_c_j[0] * idep_j - pts_j[0]);
  r[1] = s_info * (p_c_j[1] * idep_j - pts_j[1]);
  if (!jac) return;
  // reduce (2x3, projection_facor.cpp:46-50) has the sparsity [s/z 0 -s x/z^2 ; 0 s/z -s y/z^2]
  const double rz = s_info * idep_j, rx = -(rz * p_c_j[0]) * idep_j, ry = -(rz * p_c_j[1]) * idep_j;
  double A[9];  // r_ic^T R_j^T
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) A[3 * a + b] = ric[a] * Rj[3 * b] + ric[3 + a] * Rj[3 * b + 1] + ric[6 + a] * Rj[3 * b + 2];
  double RA[6], RB[6], RC[6], RT[6];
  for (int b = 0; b < 3; b++) RA[b] = rz * A[b] + rx * A[6 + b], RA[3 + b] = rz * A[3 + b] + ry * A[6 + b];  // reduce A
  for (int i = 0; i < 2; i++)
    for (int b = 0; b < 3; b++) {
      RB[3 * i + b] = RA[3 * i] * Ri[b] + RA[3 * i + 1] * Ri[3 + b] + RA[3 * i + 2] * Ri[6 + b];        // reduce A R_i
      RT[3 * i + b] = (i == 0 ? rz * ric[3 * b] : rz * ric[3 * b + 1]) + (i == 0 ? rx : ry) * ric[3 * b + 2];  // reduce r_ic^T
    }
  for (int i = 0; i < 2; i++)
    for (int b = 0; b < 3; b++)
      RC[3 * i + b] = RB[3 * i] * ric[b] + RB[3 * i + 1] * ric[3 + b] + RB[3 * i + 2] * ric[6 + b];     // reduce A R_i r_ic
  for (int i = 0; i < 2; i++) {
    const double *u = RB + 3 * i, *t = RT + 3 * i;
    // u skew(p) = (u1 p2 - u2 p1, u2 p0 - u0 p2, u0 p1 - u1 p0)
    Ji[i * 6 + 0] = RA[3 * i], Ji[i * 6 + 1] = RA[3 * i + 1], Ji[i * 6 + 2] = RA[3 * i + 2];
    Ji[i * 6 + 3] = -(u[1] * p_imu_i[2] - u[2] * p_imu_i[1]);
    Ji[i * 6 + 4] = -(u[2] * p_imu_i[0] - u[0] * p_imu_i[2]);
    Ji[i * 6 + 5] = -(u[0] * p_imu_i[1] - u[1] * p_imu_i[0]);
    Jj[i * 6 + 0] = -RA[3 * i], Jj[i * 6 + 1] = -RA[3 * i + 1], Jj[i * 6 + 2] = -RA[3 * i + 2];
    Jj[i * 6 + 3] = t[1] * p_imu_j[2] - t[2] * p_imu_j[1];
    Jj[i * 6 + 4] = t[2] * p_imu_j[0] - t[0] * p_imu_j[2];
    Jj[i * 6 + 5] = t[0] * p_imu_j[1] - t[1] * p_imu_j[0];
    Jl[i] = -(RC[3 * i] * pi3[0] + RC[3 * i + 1] * pi3[1] + RC[3 * i + 2] * pi3[2]) * (dep_i * dep_i);
  }
}

// Raw (un-whitened) IMU residual and, if Jraw != NULL, the dense 15x30 Jacobian [pose_i 6 | sb_i 9 | pose_j 6 | sb_j 9]
// (imu_factor.h:68-180 before the sqrt_info multiplication; integration_base.h:171-198).
template <class PA>
VIO_DEV void imu_eval_raw(double gravity, const double *pre, PA pose_i, PA sb_i, PA pose_j, PA sb_j, double *res,
                          double *Jraw) {
  const double sum_dt = pre[0];
  const double *delta_p = pre + 1, *delta_q = pre + 4, *delta_v = pre + 8, *lin_ba = pre + 11, *lin_bg = pre + 14;
  const double *J = pre + 17;
  PA Pi = pose_i, Pj = pose_j, Vi = sb_i, Bai = sb_i + 3, Bgi = sb_i + 6, Vj = sb_j, Baj = sb_j + 3, Bgj = sb_j + 6;
  Quat Qi{pose_i[3], pose_i[4], pose_i[5], pose_i[6]}, Qj{pose_j[3], pose_j[4], pose_j[5], pose_j[6]};
  Quat dq{delta_q[0], delta_q[1], delta_q[2], delta_q[3]};
  double dp_dba[9], dp_dbg[9], dq_dbg[9], dv_dba[9], dv_dbg[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      dp_dba[i * 3 + j] = J[(0 + i) * 15 + 9 + j];
      dp_dbg[i * 3 + j] = J[(0 + i) * 15 + 12 + j];
      dq_dbg[i * 3 + j] = J[(3 + i) * 15 + 12 + j];
      dv_dba[i * 3 + j] = J[(6 + i) * 15 + 9 + j];
      dv_dbg[i * 3 + j] = J[(6 + i) * 15 + 12 + j];
    }
  double dba[3], dbg[3], th[3], t1[3], t2[3], cdv[3], cdp[3];
  for (int k = 0; k < 3; k++) dba[k] = Bai[k] - lin_ba[k], dbg[k] = Bgi[k] - lin_bg[k];
  mat3vec(dq_dbg, dbg, th);
  Quat cdq = qmul(dq, Quat{th[0] / 2.0, th[1] / 2.0, th[2] / 2.0, 1.0});
  mat3vec(dv_dba, dba, t1), mat3vec(dv_dbg, dbg, t2);
  for (int k = 0; k < 3; k++) cdv[k] = delta_v[k] + t1[k] + t2[k];
  mat3vec(dp_dba, dba, t1), mat3vec(dp_dbg, dbg, t2);
  for (int k = 0; k < 3; k++) cdp[k] = delta_p[k] + t1[k] + t2[k];
  const double T = sum_dt;
  const double G[3] = {0, 0, gravity};
  Quat Qi_inv = qinv(Qi);
  double a[3], b[3], ra[3], rb[3];
  for (int k = 0; k < 3; k++) {
    a[k] = 0.5 * G[k] * T * T + Pj[k] - Pi[k] - Vi[k] * T;
    b[k] = G[k] * T + Vj[k] - Vi[k];
  }
  qrot(Qi_inv, a, ra), qrot(Qi_inv, b, rb);
  Quat qr = qmul(qinv(cdq), qmul(Qi_inv, Qj));
  for (int k = 0; k < 3; k++) {
    res[0 + k] = ra[k] - cdp[k];
    res[6 + k] = rb[k] - cdv[k];
    res[9 + k] = Baj[k] - Bai[k];
    res[12 + k] = Bgj[k] - Bgi[k];
  }
  res[3] = 2 * qr.x, res[4] = 2 * qr.y, res[5] = 2 * qr.z;
  if (!Jraw) return;
  // Jraw was zeroed once per solve (setup_imu_info); every evaluation overwrites the same non-zero 3x3 blocks
  double Rinv[9], M[9], S[9], Mq[9];
  qtoR(Qi_inv, Rinv);
  // columns: pose_i [0,6), sb_i [6,15), pose_j [15,21), sb_j [21,30)
  const int ld = 30, cpi = 0, csi = 6, cpj = 15, csj = 21;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Jraw[(0 + i) * ld + cpi + j] = -Rinv[i * 3 + j];
  skew3(ra, S);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Jraw[(0 + i) * ld + cpi + 3 + j] = S[i * 3 + j];
  qleft_qright33(qmul(qinv(Qj), Qi), cdq, M);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Jraw[(3 + i) * ld + cpi + 3 + j] = -M[i * 3 + j];
  skew3(rb, S);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Jraw[(6 + i) * ld + cpi + 3 + j] = S[i * 3 + j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Jraw[(0 + i) * ld + csi + 0 + j] = -Rinv[i * 3 + j] * T;
      Jraw[(0 + i) * ld + csi + 3 + j] = -dp_dba[i * 3 + j];
      Jraw[(0 + i) * ld + csi + 6 + j] = -dp_dbg[i * 3 + j];
      Jraw[(6 + i) * ld + csi + 0 + j] = -Rinv[i * 3 + j];
      Jraw[(6 + i) * ld + csi + 3 + j] = -dv_dba[i * 3 + j];
      Jraw[(6 + i) * ld + csi + 6 + j] = -dv_dbg[i * 3 + j];
      Jraw[(9 + i) * ld + csi + 3 + j] = -(double)(i == j);
      Jraw[(12 + i) * ld + csi + 6 + j] = -(double)(i == j);
    }
  qleft33(qmul(qmul(qinv(Qj), Qi), cdq), M);
  mat3mul(M, dq_dbg, Mq);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Jraw[(3 + i) * ld + csi + 6 + j] = -Mq[i * 3 + j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Jraw[(0 + i) * ld + cpj + j] = Rinv[i * 3 + j];
  qleft33(qmul(qmul(qinv(cdq), Qi_inv), Qj), M);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Jraw[(3 + i) * ld + cpj + 3 + j] = M[i * 3 + j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Jraw[(6 + i) * ld + csj + 0 + j] = Rinv[i * 3 + j];
      Jraw[(9 + i) * ld + csj + 3 + j] = (double)(i == j);
      Jraw[(12 + i) * ld + csj + 6 + j] = (double)(i == j);
    }
}

// dx of one prior block (marginalization_factor.cpp:349-367)
template <class PX, class PD>
VIO_DEV void prior_block_dx(int gsize, PX x, const double *x0, PD dx) {
  if (gsize != 7) {
    for (int k = 0; k < gsize; k++) dx[k] = x[k] - x0[k];
    return;
  }
  for (int k = 0; k < 3; k++) dx[k] = x[k] - x0[k];
  Quat q = qmul(qinv(qfrom_pose(x0)), Quat{x[3], x[4], x[5], x[6]});
  double sgn = (q.w >= 0) ? 1.0 : -1.0;
  dx[3] = sgn * 2.0 * q.x, dx[4] = sgn * 2.0 * q.y, dx[5] = sgn * 2.0 * q.z;
}

// =====================================================================================================
// One-off setup per solve
// =====================================================================================================

// cov^-1 for every IMU factor: Gauss-Jordan with partial pivoting on [cov | I] (the same elimination order as the
// CPU restatement, so the two agree bit for bit), then the lower triangle is mirrored (Eigen's LLT only reads it).
#ifdef VIO_EMUL
template <class MP>
VIO_DEV void setup_imu_info(const Ctx &cx, const WinView &v, MP) {
  const int W = v.W;
  VIO_PARFOR(q, W * 450) {
    int f = q / 450, e = q % 450, r = e / 30, c = e % 30;
    v.imu_aug[q] = c < 15 ? v.preint[f * kPreintDoubles + 242 + r * 15 + c] : (double)(c - 15 == r);
  }
  VIO_SYNC();
  for (int c = 0; c < 15; c++) {
    // pivot search + row swap: one thread per factor (15 compares), then normalisation and elimination in parallel
    VIO_PARFOR(f, W) {
      double *A = v.imu_aug + f * 450;
      int piv = c;
      for (int r = c + 1; r < 15; r++)
        if (fabs(A[r * 30 + c]) > fabs(A[piv * 30 + c])) piv = r;
      if (piv != c)
        for (int j = 0; j < 30; j++) {
          double t = A[c * 30 + j];
          A[c * 30 + j] = A[piv * 30 + j];
          A[piv * 30 + j] = t;
        }
      double d = A[c * 30 + c];
      for (int j = 0; j < 30; j++) A[c * 30 + j] /= d;
      // stash the multipliers of this column (they are overwritten during elimination)
      for (int r = 0; r < 15; r++) v.imu_Mr[f * 15 + r] = A[r * 30 + c];
    }
    VIO_SYNC();
    VIO_PARFOR(q, W * 450) {
      int f = q / 450, e = q % 450, r = e / 30, j = e % 30;
      if (r != c) {
        double fac = v.imu_Mr[f * 15 + r];
        if (fac != 0.0) v.imu_aug[q] -= fac * v.imu_aug[f * 450 + c * 30 + j];
      }
    }
    VIO_SYNC();
  }
  VIO_PARFOR(q, W * 225) {
    int f = q / 225, e = q % 225, r = e / 15, c = e % 15;
    int rr = r >= c ? r : c, cc = r >= c ? c : r;  // lower triangle mirrored
    v.imu_info[q] = v.imu_aug[f * 450 + rr * 30 + 15 + cc];
  }
  VIO_PARFOR(q, W * 450) v.imu_J[q] = 0.0;
  VIO_SYNC();
}
#else
// Device form of the Gauss-Jordan elimination: 32 lanes per factor, lane j keeps column j of [cov | I] in registers through all
// 15 pivots; the pivot column travels through a 16-double mailbox per factor (in the still unused matrix buffer). Same
// arithmetic, same order as the loop above. Since round 6 the fallback of setup_imu_info below (a covariance that is not
// positive definite to working precision).
template <class MP>
VIO_DEV void setup_imu_info_gj(const Ctx &cx, const WinView &v, MP mbox) {
  const int W = v.W;
  const int tid_ = VIO_TID(cx), j = tid_ & 31, slot = tid_ >> 5, nslots = cx.nt >> 5;
  for (int f0 = 0; f0 < W; f0 += nslots) {
    const int f = f0 + slot;
    const bool act = f < W && j < 30;
    const int fc = f < W ? f : 0;
    auto mb = mbox + slot * 32;
    double x[15];
#pragma unroll
    for (int r = 0; r < 15; r++)
      x[r] = j < 15 ? v.preint[fc * kPreintDoubles + 242 + r * 15 + (j < 15 ? j : 0)] : (double)(j - 15 == r);
#pragma unroll
    for (int c = 0; c < 15; c++) {
      if (j == c) {  // owner of the pivot column: partial pivoting (first maximum), then publish the column
        int piv = c;
        double best = fabs(x[c]);
#pragma unroll
        for (int r = c + 1; r < 15; r++)
          if (fabs(x[r]) > best) best = fabs(x[r]), piv = r;
#pragma unroll
        for (int r = 0; r < 15; r++) mb[r] = x[r];
        mb[15] = (double)piv;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int piv = (int)mb[15];
      double m[15];
#pragma unroll
      for (int r = 0; r < 15; r++) m[r] = mb[r];  // column c before the swap
      __builtin_amdgcn_wave_barrier();
      // swap rows c <-> piv in my column and in the multiplier column
      double xp = x[c], mp = m[c];
#pragma unroll
      for (int r = 0; r < 15; r++) {
        if (r == piv) xp = x[r], mp = m[r];
      }
#pragma unroll
      for (int r = 0; r < 15; r++) {
        if (r == piv && r != c) x[r] = x[c], m[r] = m[c];
      }
      const double d = mp;  // A[c][c] after the swap
      x[c] = xp / d;
      // multipliers are read after the normalisation of row c: A[r][c] for r != c is unchanged by it
#pragma unroll
      for (int r = 0; r < 15; r++) {
        if (r != c) {
          const double fac = m[r];
          if (fac != 0.0) x[r] -= fac * x[c];
        }
      }
    }
    if (act && j >= 15) {
      const int cc = j - 15;
#pragma unroll
      for (int rr = 0; rr < 15; rr++)
        if (rr >= cc) v.imu_info[f * 225 + rr * 15 + cc] = x[rr], v.imu_info[f * 225 + cc * 15 + rr] = x[rr];
    }
  }
  VIO_SYNC();
}
#endif

#ifndef VIO_EMUL
#ifdef VIO_SIMT
typedef ::simt::v4d v4d;
#else
typedef double v4d __attribute__((ext_vector_type(4)));
#endif

// v_mfma_f64_16x16x4_f64: D = A(16x4) B(4x16) + C. Lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; it receives
// D[(l>>4) + 4 r][l&15] in element r (the f64 C/D map differs from the f32 one, cdna_hip_programming.md §3).
VIO_DEV v4d mfma_f64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
#endif

// Prior constants. MarginalizationFactor evaluates r = r0 + J0 dx with a constant J0 (marginalization_factor.cpp:
// 336-384); everything the solver takes from it is a function of  H0 = J0^T J0  and  b0 = J0^T r0:
//     J^T r = b0 + H0 dx,    cost = |r0|^2 / 2 + b0 . dx + dx . (H0 dx) / 2,    J^T J = H0,
// so one symmetric mat-vec per evaluation serves the cost-only and the Jacobian evaluations alike (the first version
// did r0 + J0 dx and J0^T r: two passes over two copies of J0). Here: the column map, H0 and b0, once per solve.
template <class WK>
VIO_DEV void setup_prior(const Ctx &cx, const WinView &v, WK &w) {
  const int n = v.prior_n;
  if (n <= 0) return;
  VIO_PARFOR(a, n) w.prcol[a] = -1;
  VIO_SYNC();
  VIO_PARFOR(b, v.prior_nb) {
    int kind = v.pr_kind[b], idx = v.pr_index[b], o = v.pr_offset[b];
    if (kind == 0)
      for (int k = 0; k < 6; k++) w.prcol[o + k] = (idx << 8) | k;
    else if (kind == 1)
      for (int k = 0; k < 9; k++) w.prcol[o + k] = (idx << 8) | (6 + k);
  }
  // J0 goes through the (not yet assembled) matrix buffer when it fits: the n^2 dot products then read LDS
  const bool stage = (size_t)n * n <= (size_t)w.nstage;
  auto Js = w.stage;
  if (stage) {
    VIO_PARFOR(q, n * n) Js[q] = v.pr_J[q];
    VIO_SYNC();
  }
#ifndef VIO_EMUL
  // H0 = J0^T J0 on the matrix cores: one 16 x 16 tile of the lower triangle per wave visit (mirrored on store), the
  // k loop in chunks of 8 steps whose 16 operand fetches (4 row segments of 128 B each) are issued together
  {
    const int tid_ = VIO_TID(cx), wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), nw = cx.nt >> 6, lane = tid_ & 63;
    const int li = lane & 15, kq = lane >> 4;
    const int nt16 = (n + 15) >> 4, ntiles = nt16 * (nt16 + 1) / 2, ksteps = (n + 3) >> 2;
    for (int t = wave; t < ntiles; t += nw) {
      int ti = 0;
      while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
      const int tj = t - ti * (ti + 1) / 2;
      const int ca = 16 * ti + li, cb = 16 * tj + li;
      const bool va = ca < n, vb = cb < n;
      const int oa = va ? ca : 0, ob = vb ? cb : 0;
      v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
      constexpr int kChunk = 8;
      for (int s0 = 0; s0 < ksteps; s0 += kChunk) {
        double av[kChunk], bv[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; j++) {
          const int k = 4 * (s0 + j) + kq, kc = k < n ? k : 0;
          if (stage) av[j] = Js[kc * n + oa], bv[j] = Js[kc * n + ob];
          else av[j] = v.pr_J[kc * n + oa], bv[j] = v.pr_J[kc * n + ob];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < kChunk; j++) {
          const bool vk = 4 * (s0 + j) + kq < n;
          av[j] = (va && vk) ? av[j] : 0.0, bv[j] = (vb && vk) ? bv[j] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < kChunk; j += 2) acc0 = mfma_f64(av[j], bv[j], acc0), acc1 = mfma_f64(av[j + 1], bv[j + 1], acc1);
      }
      acc0 += acc1;
#pragma unroll
      for (int r = 0; r < 4; r++) {  // element r: row 16 ti + kq + 4 r, column 16 tj + li
        const int a = 16 * ti + kq + 4 * r, b = cb;
        if (a < n && b < n) {
          v.prH0[a * n + b] = acc0[r];
          if (ti != tj) v.prH0[b * n + a] = acc0[r];
        }
      }
    }
  }
  if (false) {
    const int tid_ = 0, kLanes = 1, lane = 0, nwv = 1;
#else
  // rows of the symmetric H0 by wave, columns by lane: no integer division per element
  {
    const int tid_ = VIO_TID(cx), kLanes = (int)cx.nt < 64 ? (int)cx.nt : 64, lane = tid_ % kLanes, nwv = (int)cx.nt / kLanes;  // (host emulation: one thread)
#endif
    for (int a = tid_ / kLanes; a < n; a += nwv)
      for (int b = lane; b < n; b += kLanes) {
        double s = 0;
        if (stage) {
#pragma unroll 5
          for (int k = 0; k < n; k++) s = fma(Js[k * n + a], Js[k * n + b], s);
        } else {
          for (int k = 0; k < n; k++) s = fma(v.pr_J[k * n + a], v.pr_J[k * n + b], s);
        }
        v.prH0[a * n + b] = s;
      }
  }
  VIO_PARFOR(a, n) {
    double s = 0;
    if (stage) {
      for (int k = 0; k < n; k++) s = fma(Js[k * n + a], v.pr_r[k], s);
    } else {
      for (int k = 0; k < n; k++) s = fma(v.pr_J[k * n + a], v.pr_r[k], s);
    }
    v.prb0[a] = s;
  }
  const int napp = (int)tri_doubles(v.nrows), nband = 2 * v.P * kSS;
  VIO_PARFOR(q, kSB * v.jp) v.Apri[q] = 0.0;
  VIO_PARFOR(q, napp + nband) v.AppPr[q] = 0.0;
  VIO_SYNC();
  // H0 in the layout of the reduced matrix, constant during the solve: pose x pose -> App, speed-bias x speed-bias ->
  // Dss / Css (copied into place by every linearization), speed-bias x pose -> Apri (read where it is needed)
  VIO_PARFOR(q, n * n) {
    const int a = q / n, b = q - a * n, pa = w.prcol[a], pb = w.prcol[b];
    if (pa < 0 || pb < 0 || pa < pb) continue;
    const int fr = pa >> 8, cr = pa & 255, fc = pb >> 8, cc = pb & 255;
    const double x = v.prH0[q];
    if (cr < 6 && cc < 6) v.AppPr[tri_at(6 * fr + cr, 6 * fc + cc)] = x;
    else if (cr >= 6 && cc >= 6) {
      if (fr == fc) v.AppPr[napp + fr * kSS + (cr - 6) * kSB + (cc - 6)] = x;
      else if (fr - fc == 1) v.AppPr[napp + v.P * kSS + fr * kSS + (cc - 6) * kSB + (cr - 6)] = x;
    } else if (cr >= 6) v.Apri[(cr - 6) * v.jp + 6 * fc + cc] = x;
    else v.Apri[(cc - 6) * v.jp + 6 * fr + cr] = x;
  }
  VIO_SYNC();
}

// ---- wave-level device helpers (matrix cores, v_readlane) ------------------------------------------
#ifndef VIO_EMUL

// Value of x held by `lane` (compile-time constant) broadcast to the whole wave through SGPRs.
VIO_DEV double lane_bcast(double x, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}
// Sum over the four lanes of a quad (DPP quad_perm [1,0,3,2] then [2,3,0,1]); every lane of the quad gets the total.
VIO_DEV double quad_sum_f64(double v) {
  v += dpp_move_f64<0xB1, 0xf>(v);
  v += dpp_move_f64<0x4E, 0xf>(v);
  return v;
}

// Layouts of v_mfma_f64_16x16x4 (lane l: li = l & 15, kq = l >> 4):
//   operand layout of a tile X: k-step s takes X[li][4 s + kq] as A operand (rows of the product) and, for products
//   with X^T on the right, as B operand;  accumulator layout: element r is [kq + 4 r][li].
// The accumulator layout of a tile T is at the same time the B-operand layout of T over four k-steps (element r = k-step
// r), so chains of products M1 (M2 T) never leave the registers.

// Under register pressure the scheduler sinks every LDS read to its first use: a ds_read, s_waitcnt lgkmcnt(0), the
// select that masks it, the next ds_read... -- one LDS latency (~120 cycles) per VALUE on the serial chains of the
// factorization. The fetch helpers below therefore read raw (clamped addresses, nothing consumes the value), then a
// scheduling fence, then mask: one latency per batch.

// ---- 9 x 9 speed-bias blocks (row-major, ld 9) in 16 x 16 register tiles --------------------------------------------
// X[li][4 s + kq], zero outside the block (k-steps 0..2 cover k < 12)
VIO_DEV void load_op9_raw(cldsd X, int li, int kq, double out[3]) {
  cldsd p = X + (li < kSB ? li : 0) * kSB + kq;
  out[0] = p[0], out[1] = p[4], out[2] = p[kq == 0 ? 8 : 0];
}
VIO_DEV void mask_op9(int li, int kq, double out[3]) {
  const bool iok = li < kSB;
  out[0] = iok ? out[0] : 0.0, out[1] = iok ? out[1] : 0.0, out[2] = (iok && kq == 0) ? out[2] : 0.0;
}
VIO_DEV void load_op9(cldsd X, int li, int kq, double out[3]) {
  load_op9_raw(X, li, kq, out);
  VIO_SCHED_FENCE();
  mask_op9(li, kq, out);
}
// Linv[li][4 s + kq] of a factored diagonal block (potrf9_inv_wave): strict lower part of L^-1 transposed above the
// diagonal (D[kk][n] = Linv[n][kk], kk < n), 1 / L_nn in ldinv_k. As A operand: Linv (.) ; as B operand: (.) L^-T.
VIO_DEV void load_linv9_raw(cldsd D, cldsd ldinv_k, int li, int kq, double out[4]) {
  const int n = li < kSB ? li : 0;
  out[3] = ldinv_k[n];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    const int kk = 4 * s + kq;
    out[s] = D[(kk < n ? kk : 0) * kSB + n];
  }
}
VIO_DEV void mask_linv9(int li, int kq, double out[4]) {
  const bool iok = li < kSB;
  const int n = iok ? li : 0;
#pragma unroll
  for (int s = 0; s < 3; s++) {
    const int kk = 4 * s + kq;
    out[s] = (iok && kk < n) ? out[s] : ((iok && kk == n) ? out[3] : 0.0);
  }
}
VIO_DEV void load_linv9(cldsd D, cldsd ldinv_k, int li, int kq, double out[4]) {
  load_linv9_raw(D, ldinv_k, li, kq, out);
  VIO_SCHED_FENCE();
  mask_linv9(li, kq, out);
}

// Cholesky of one 9 x 9 diagonal block AND the inverse of its factor by one wave, entirely on the matrix cores.
// The block lives in the f64 accumulator layout as a full symmetric matrix; pivot c: row c sits in the 16 lanes
// kq == (c & 3), element c >> 2, which is exactly where the A and the B operand of k-slot (c & 3) are fetched from, so the
// rank-1 update D -= l l^T is ONE v_mfma with a = b = l and no data movement. ET (initially I) receives the same
// eliminations, ET -= l e^T with e = ET[c][:] / L_cc, which leaves e = row c of L^-1. with_update: D -= E E^T first
// (E = the factored coupling block to the frame eliminated before, operand layout from LDS) -- the look-ahead of the
// band. Stored: L in the lower triangle (with diagonal), L^-1's strict lower part TRANSPOSED in the strict upper
// triangle, 1 / L_cc in ldinv_k. Returns false if a pivot is <= 0.
// (Four pivots per step like potrf16_wave -- 4 + 4 + 1 -- was built for this block as well: 2 196 against 2 124 cycles alone, 2 473
// against 2 440 with the E update, window kernel +0.5 %: with ONE matrix instruction per pivot and the next pivot's reciprocal root in
// its shadow the rank-1 form is already at the cost of the blocked one. Not kept.)
VIO_DEV bool potrf9_inv_wave(ldsd D, cldsd Eprev, bool with_update, ldsd ldinv_k, int lane) {
  // Round 6: ONE matrix instruction per pivot. The block only fills 9 x 9 of the 16 x 16 tile; the inverse rides in the rest of
  // the SAME tile as a symmetric border:  T = [ A  E ; E^T  * ],  E = columns 0..6 of the running inverse (rows 0..8, tile
  // columns 9..15; initially the identity). The rank-1 update T -= v v^T with v = (row c of T) / L_cc = [l | e] then applies
  // A -= l l^T and E -= l e^T at once -- rounds 3-5 kept the inverse in a second accumulator and paid a second 64-cycle
  // instruction per pivot (an f64 matrix instruction is 16 passes of the vector unit's own double-precision pipe on this chip).
  // Column 7 of the inverse has ONE entry below the diagonal, L^-1[8][7] = -L[8][7] / (L_77 L_88): formed by hand; column 8 none.
  const int n = lane & 15, kq = lane >> 4;
  v4d A;
  double l[3];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int m = kq + 4 * r;
    const bool ok = m < kSB && n < kSB;
    const int hi = m > n ? m : n, lo = m > n ? n : m;
    A[r] = D[ok ? hi * kSB + lo : 0];
  }
  load_op9_raw(Eprev, n, kq, l);
  VIO_SCHED_FENCE();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int m = kq + 4 * r;
    // (the border: E[m][n - 9] = [m == n - 9] right of the block, its transpose below it)
    A[r] = (m < kSB && n < kSB) ? A[r] : ((m + 9 == n || n + 9 == m) ? 1.0 : 0.0);
  }
  if (with_update) {
    mask_op9(n, kq, l);
#pragma unroll
    for (int s = 0; s < 3; s++) A = mfma_f64(-l[s], l[s], A);
  }
  // Everything in this loop is on the critical path of the solve and nothing in a wave overlaps its own matrix
  // instructions, so the loop carries no bookkeeping: a pivot <= 0 shows up as a NaN / inf reciprocal root and is tested
  // once at the end.
  double keep[3] = {0.0, 0.0, 0.0}, myinv = 0.0, l87 = 0.0, y7 = 0.0, linv87 = 0.0;
  double dcc = lane_bcast(A[0], 0);
#pragma unroll
  for (int c = 0; c < kSB; c++) {
    double y = __builtin_amdgcn_rsq(dcc);
    const double h = 0.5 * dcc;
    y = y * fma(-h * y, y, 1.5);  // (v_rsq_f64 is specified to 2^29 ulp, ~2^-23 relative: one Newton step leaves ~1.5 e^2 = ~2^-45
                                  //  in every pivot, nine orders below the 1e-6 bar against the reference
                                  //  (tests/test_backend_gpu.py::test_step_quality_traces_on_badly_conditioned_windows holds low-parallax windows at
                                  //  the smallest mu to it); the second step cost 3 dependent operations on the pivot chain)
    const bool sel = kq == (c & 3);
    const double a = sel ? A[c >> 2] * y : 0.0;  // v: l[n] = L[n][c] for n < 9 (n == c: dcc / sqrt(dcc)), e[n - 9] = L^-1[c][n - 9] right of it
    keep[c >> 2] = sel ? a : keep[c >> 2];
    myinv = (n == c) ? y : myinv;
    if (c == 7) l87 = lane_bcast(a, 16 * 3 + 8), y7 = y;
    if (c == 8) linv87 = -(l87 * y7) * y;
    if (c + 1 < kSB) {
      // the next pivot D[c+1][c+1] - l[c+1]^2 is formed ahead of the matrix instruction, so its rsqrt chain runs in
      // its shadow instead of behind it
      const double lnext = lane_bcast(a, 16 * (c & 3) + c + 1);
      const double dold = lane_bcast(A[(c + 1) >> 2], 16 * ((c + 1) & 3) + c + 1);
      dcc = fma(-lnext, lnext, dold);
    }
    A = mfma_f64(-a, a, A);
  }
  // lane (n, kq) captured, at pivot c = kq + 4 j, L[n][c] (n >= c) or -- in the border -- L^-1[c][n - 9] (n - 9 < c), which goes
  // TRANSPOSED above the diagonal of the stored block
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int c = kq + 4 * j;
    if (c < kSB) {
      if (n < kSB && n >= c) D[n * kSB + c] = keep[j];
      if (n >= kSB && n - kSB < c) D[(n - kSB) * kSB + c] = keep[j];
    }
  }
  if (lane == 0) D[7 * kSB + 8] = linv87;
  if (kq == 0 && n < kSB) ldinv_k[n] = myinv;
  // rsq of a pivot <= 0 (or NaN) is NaN / inf, and every later pivot inherits it
  const bool bad = n < kSB && !(myinv > 0.0 && myinv < 1.7976931348623157e308);
  return __builtin_amdgcn_ballot_w64(bad) == 0;
}

// ---- 16 x 16 tiles of a row-major matrix (leading dimension per tile row) -------------------------------------------
// operand X[li][kq + 4 s]; rows >= rows are zero (the lane reads row 0 instead)
template <class MP>
VIO_DEV void tile_load_op_raw(MP X, int ld, int rows, int li, int kq, double out[4]) {
  auto p = X + (li < rows ? li : 0) * ld + kq;
  out[0] = p[0], out[1] = p[4], out[2] = p[8], out[3] = p[12];
}
VIO_DEV void tile_mask_op(int rows, int li, double out[4]) {
  const bool ok = li < rows;
  out[0] = ok ? out[0] : 0.0, out[1] = ok ? out[1] : 0.0, out[2] = ok ? out[2] : 0.0, out[3] = ok ? out[3] : 0.0;
}
template <class MP>
VIO_DEV void tile_load_op(MP X, int ld, int rows, int li, int kq, double out[4]) {
  tile_load_op_raw(X, ld, rows, li, kq, out);
  VIO_SCHED_FENCE();
  tile_mask_op(rows, li, out);
}
template <class MP>
VIO_DEV v4d tile_load_acc_raw(MP C, int ld, int rows, int li, int kq) {
  v4d a;
#pragma unroll
  for (int r = 0; r < 4; r++) a[r] = C[(kq + 4 * r < rows ? kq + 4 * r : 0) * ld + li];
  return a;
}
VIO_DEV v4d tile_mask_acc(v4d a, int rows, int kq) {
#pragma unroll
  for (int r = 0; r < 4; r++) a[r] = kq + 4 * r < rows ? a[r] : 0.0;
  return a;
}
template <class MP>
VIO_DEV v4d tile_load_acc(MP C, int ld, int rows, int li, int kq) {
  v4d a = tile_load_acc_raw(C, ld, rows, li, kq);
  VIO_SCHED_FENCE();
  return tile_mask_acc(a, rows, kq);
}
template <class MP>
VIO_DEV void tile_store_acc(MP C, int ld, int rows, int li, int kq, v4d a) {
  if (rows >= 16) {  // (uniform: four plain stores instead of four exec-masked regions)
    auto p = C + kq * ld + li;
    p[0] = a[0], p[4 * ld] = a[1], p[8 * ld] = a[2], p[12 * ld] = a[3];
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; r++)
    if (kq + 4 * r < rows) C[(kq + 4 * r) * ld + li] = a[r];
}
// Diagonal tile of the pose matrix: like potrf9_inv_wave on 16 x 16. nvalid rows / columns of the tile exist, the first
// npiv of them are pivots; rows past npiv (the carried right-hand side) are eliminated along and end up as their row of L.
// with_update: D -= Lprev Lprev^T first (the panel tile left of D, same tile row).
// Round 6: FOUR pivots per step. A rank-1 update used one of the four k-slots of the matrix instruction (a = b = the scaled pivot
// row in the lanes kq == c & 3, zeros elsewhere) and every pivot paid the chain matrix instruction -> vector read -> two broadcasts
// -> reciprocal root: ~306 cycles x 16. The rows 4 cb .. 4 cb + 3 of the running matrix are accumulator element cb of ALL lanes, i.e.
// exactly the B operand of one k-step: with M = L44^-1, the inverse Cholesky factor of the 4 x 4 diagonal block (ten broadcasts, then
// formed redundantly by every lane: four reciprocal roots, ~40 multiply-adds),
//     V = M R          one matrix instruction (A operand: M in the lanes li < 4; B operand: accumulator element cb)
//     T -= V^T V       one matrix instruction with a = b = V -- its four rows land in element 0 of the lanes kq = row: operand layout
// eliminates four pivots; the same M applied to the rows of the inverse's accumulator gives its four rows. 4 matrix instructions per
// four pivots instead of 8, and one broadcast / reciprocal-root round per block instead of four.
template <class MP>
VIO_DEV bool potrf16_wave(MP D, MP Lprev, int ld, int nvalid, int npiv, bool with_update, ldsd ldinv_k, int lane) {
  const int n = lane & 15, kq = lane >> 4;
  v4d A, E;
  double l[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int m = kq + 4 * r;
    const bool ok = m < nvalid && n < nvalid;
    const int hi = m > n ? m : n, lo = m > n ? n : m;
    A[r] = D[ok ? hi * ld + lo : 0];
  }
  tile_load_op_raw(Lprev, ld, nvalid, n, kq, l);
  VIO_SCHED_FENCE();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int m = kq + 4 * r;
    A[r] = (m < nvalid && n < nvalid) ? A[r] : 0.0;
    E[r] = (m == n) ? 1.0 : 0.0;
  }
  if (with_update) {
    tile_mask_op(nvalid, n, l);
#pragma unroll
    for (int s = 0; s < 4; s++) A = mfma_f64(-l[s], l[s], A);
  }
  double keep[4] = {0.0, 0.0, 0.0, 0.0}, myinv = 0.0;
  const int mi = (n < 4 && kq <= n) ? n * (n + 1) / 2 + kq : -1;  // which entry of M this lane feeds into the A operand
#pragma unroll
  for (int cb = 0; cb < 4; cb++) {
    if (4 * cb < npiv) {  // (uniform)
      // the diagonal block: element (row 4 cb + i, column 4 cb + j) is accumulator element cb of lane (n = 4 cb + j, kq = i)
      double b[10];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) b[i * (i + 1) / 2 + j] = lane_bcast(A[cb], 16 * i + 4 * cb + j);
      const bool p1 = 4 * cb + 1 < npiv, p2 = 4 * cb + 2 < npiv, p3 = 4 * cb + 3 < npiv;  // (pivots past npiv: their row of M is zero)
      auto rsq1 = [](double d) {
        double y = __builtin_amdgcn_rsq(d);
        return y * fma(-(0.5 * d) * y, y, 1.5);  // (one Newton step: potrf9_inv_wave)
      };
      const double y0 = rsq1(b[0]);
      const double l10 = b[1] * y0, l20 = b[3] * y0, l30 = b[6] * y0;
      const double y1 = rsq1(fma(-l10, l10, b[2]));
      const double l21 = fma(-l20, l10, b[4]) * y1, l31 = fma(-l30, l10, b[7]) * y1;
      const double y2 = rsq1(fma(-l21, l21, fma(-l20, l20, b[5])));
      const double l32 = fma(-l31, l21, fma(-l30, l20, b[8])) * y2;
      const double y3 = rsq1(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, b[9]))));
      // M = L44^-1 (lower), row i scaled by y_i
      double M[10];
      M[0] = y0;
      M[1] = p1 ? -(l10 * y0) * y1 : 0.0, M[2] = p1 ? y1 : 0.0;
      M[3] = p2 ? -fma(l21, M[1], l20 * y0) * y2 : 0.0, M[4] = p2 ? -(l21 * M[2]) * y2 : 0.0, M[5] = p2 ? y2 : 0.0;
      M[6] = p3 ? -fma(l32, M[3], fma(l31, M[1], l30 * y0)) * y3 : 0.0, M[7] = p3 ? -fma(l32, M[4], l31 * M[2]) * y3 : 0.0;
      M[8] = p3 ? -(l32 * M[5]) * y3 : 0.0, M[9] = p3 ? y3 : 0.0;
      double mop = 0.0;
#pragma unroll
      for (int q = 0; q < 10; q++) mop = mi == q ? M[q] : mop;
      const v4d z = {0.0, 0.0, 0.0, 0.0};
      const v4d V = mfma_f64(mop, A[cb], z), VE = mfma_f64(mop, E[cb], z);
      const double vv = V[0], ve = VE[0];
      A = mfma_f64(-vv, vv, A);
      E = mfma_f64(-vv, ve, E);
      const int c = 4 * cb + kq;  // this lane's pivot of the block: column c of L below the diagonal, row c of L^-1 left of it
      keep[cb] = n >= c ? vv : ve;
      const double yn = (n & 3) == 0 ? y0 : (n & 3) == 1 ? y1 : (n & 3) == 2 ? y2 : y3;  // (1 / L_nn in every lane of column n)
      myinv = (n >> 2) == cb ? yn : myinv;
    }
  }
  if (n < nvalid) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int c = kq + 4 * j;
      if (c < npiv) D[n * ld + c] = keep[j];
    }
    if (kq == 0 && n < npiv) ldinv_k[n] = myinv;
  }
  const bool bad = n < npiv && !(myinv > 0.0 && myinv < 1.7976931348623157e308);
  return __builtin_amdgcn_ballot_w64(bad) == 0;
}
// (the rank-1 form of rounds 2-5: one pivot per step; tools/microbench/band_bench.hip times the two side by side)
template <class MP>
VIO_DEV bool potrf16_wave_rank1(MP D, MP Lprev, int ld, int nvalid, int npiv, bool with_update, ldsd ldinv_k, int lane) {
  const int n = lane & 15, kq = lane >> 4;
  v4d A, E;
  double l[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int m = kq + 4 * r;
    const bool ok = m < nvalid && n < nvalid;
    const int hi = m > n ? m : n, lo = m > n ? n : m;
    A[r] = D[ok ? hi * ld + lo : 0];
  }
  tile_load_op_raw(Lprev, ld, nvalid, n, kq, l);
  VIO_SCHED_FENCE();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int m = kq + 4 * r;
    A[r] = (m < nvalid && n < nvalid) ? A[r] : 0.0;
    E[r] = (m == n) ? 1.0 : 0.0;
  }
  if (with_update) {
    tile_mask_op(nvalid, n, l);
#pragma unroll
    for (int s = 0; s < 4; s++) A = mfma_f64(-l[s], l[s], A);
  }
  double keep[4] = {0.0, 0.0, 0.0, 0.0}, myinv = 0.0;
  double dcc = lane_bcast(A[0], 0);
#pragma unroll
  for (int c = 0; c < 16; c++) {  // (constant trip count: c is a compile-time constant in every copy of the body)
    if (c < npiv) {
      double y = __builtin_amdgcn_rsq(dcc);
      const double h = 0.5 * dcc;
      y = y * fma(-h * y, y, 1.5);  // (one Newton step: potrf9_inv_wave)
      const bool sel = kq == (c & 3);
      const double a = sel ? A[c >> 2] * y : 0.0;
      const double e = sel ? E[c >> 2] * y : 0.0;
      keep[c >> 2] = sel ? (n >= c ? a : e) : keep[c >> 2];
      myinv = (n == c) ? y : myinv;
      if (c + 1 < 16) {
        const double lnext = lane_bcast(a, 16 * (c & 3) + c + 1);
        const double dold = lane_bcast(A[(c + 1) >> 2], 16 * ((c + 1) & 3) + c + 1);
        dcc = fma(-lnext, lnext, dold);
      }
      A = mfma_f64(-a, a, A);
      E = mfma_f64(-a, e, E);
    }
  }
  if (n < nvalid) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int c = kq + 4 * j;
      if (c < npiv) D[n * ld + c] = keep[j];
    }
    if (kq == 0 && n < npiv) ldinv_k[n] = myinv;
  }
  const bool bad = n < npiv && !(myinv > 0.0 && myinv < 1.7976931348623157e308);
  return __builtin_amdgcn_ballot_w64(bad) == 0;
}
// A_ik <- A_ik L_kk^-T (panel tile below a factored FULL diagonal tile: 16 pivots)
template <class MP>
VIO_DEV void tile_trsm(MP Aik, int ld_i, int rows, MP Dkk, int ld_k, cldsd ldinv_k, int li, int kq) {
  double a[4], x[4];
  tile_load_op_raw(Aik, ld_i, rows, li, kq, a);
  const double dg = ldinv_k[li];
#pragma unroll
  for (int s = 0; s < 4; s++) x[s] = Dkk[(4 * s + kq) * ld_k + li];  // B[kk][li] = Linv[li][kk]: above the diagonal of Dkk for kk < li
  VIO_SCHED_FENCE();
  tile_mask_op(rows, li, a);
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int kk = 4 * s + kq;
    const double bb = kk < li ? x[s] : (kk == li ? dg : 0.0);
    acc = mfma_f64(a[s], bb, acc);
  }
  tile_store_acc(Aik, ld_i, rows, li, kq, acc);
}
// C_ij -= A_ik A_jk^T
template <class MP>
VIO_DEV void tile_update(MP C, int ld_i, int rows_i, MP A, MP B, int ld_j, int rows_j, int li, int kq) {
  double a[4], b[4];
  tile_load_op_raw(A, ld_i, rows_i, li, kq, a), tile_load_op_raw(B, ld_j, rows_j, li, kq, b);
  v4d acc = tile_load_acc_raw(C, ld_i, rows_i, li, kq);
  VIO_SCHED_FENCE();
  tile_mask_op(rows_i, li, a), tile_mask_op(rows_j, li, b);
  acc = tile_mask_acc(acc, rows_i, kq);
#pragma unroll
  for (int s = 0; s < 4; s++) acc = mfma_f64(-a[s], b[s], acc);
  tile_store_acc(C, ld_i, rows_i, li, kq, acc);
}
// cov^-1 of every IMU factor (round 6): the covariance of a pre-integration is symmetric positive definite, so one wave
// factors it on the matrix cores -- potrf16_wave on a 15-pivot tile, which leaves L and L^-1 -- and forms cov^-1 = L^-T L^-1
// as ONE Gram product of the inverse factor: 15 pivots + 4 matrix instructions per factor, 10 factors on 4 waves. The
// Gauss-Jordan elimination with partial pivoting of rounds 1-5 (the CPU restatement's order, 32 lanes per factor, a mailbox
// round trip per pivot) took 75 k cycles per solve, 120 k with two windows per CU; it stays as the fallback for a covariance
// whose factorization meets a non-positive pivot. The reference inverts with Eigen's partial-pivoting LU (imu_factor.h:72 via
// integration_base.h): any backward-stable inverse agrees with it to cond(cov) eps, far inside the 1e-6 bar (goldens).
template <class MP>
VIO_DEV void setup_imu_info(const Ctx &cx, const WinView &v, MP mbox) {
  const int W = v.W;
  VIO_PARFOR(q, W * 450) v.imu_J[q] = 0.0;
  constexpr int kLd = 17, kTile = 16 * kLd + 16;  // one tile (odd leading dimension) + 16 pivot reciprocals per wave
  const int tid_ = VIO_TID(cx), wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), nw = cx.nt >> 6, lane = tid_ & 63;
  const int n = lane & 15, kq = lane >> 4;
  auto T = mbox + wave * kTile;
  auto ldv = T + 16 * kLd;
  bool all_good = true;
  if (cx.tid == 0) mbox[nw * kTile] = 0.0;  // (fallback flag)
  VIO_SYNC();
  for (int f = wave; f < W; f += nw) {
    const double *cov = v.preint + f * kPreintDoubles + 242;
    double c4[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {  // element (row kq + 4 r, column n): the lower triangle is what the factorization reads
      const int row = kq + 4 * r, hi = row > n ? row : n, lo = row > n ? n : row;
      c4[r] = cov[(hi < 15 ? hi : 0) * 15 + (lo < 15 ? lo : 0)];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = kq + 4 * r;
      T[row * kLd + n] = (row < 15 && n < 15) ? c4[r] : (row == n ? 1.0 : 0.0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool good = potrf16_wave(T, T, kLd, 15, 15, false, ldv, lane);
    all_good = all_good && good;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // X = L^-1 (lower): X[k][i] sits above the diagonal of the stored tile at T[i][k] for k > i, 1 / L_ii on the diagonal.
    // cov^-1 = X^T X: A operand (row i, k-slot 4 s + kq) and B operand (k-slot, column i) are the same register.
    double x[4];
    const double dg = ldv[n < 15 ? n : 0];
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) x[s4] = T[(n < 15 ? n : 0) * kLd + (4 * s4 + kq)];
    VIO_SCHED_FENCE();
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      const int k = 4 * s4 + kq;
      const double xv = (n < 15 && k < 15) ? (k > n ? x[s4] : (k == n ? dg : 0.0)) : 0.0;
      acc = mfma_f64(xv, xv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = kq + 4 * r;
      if (row < 15 && n < 15) v.imu_info[f * 225 + row * 15 + n] = acc[r];
    }
    __builtin_amdgcn_wave_barrier();  // (the tile is rewritten by the wave's next factor)
  }
  if (__builtin_amdgcn_ballot_w64(!all_good) != 0 && lane == 0) mbox[nw * kTile] = 1.0;
  VIO_SYNC();
  const bool fallback = mbox[nw * kTile] == 1.0;
  VIO_SYNC();
  if (fallback) setup_imu_info_gj(cx, v, mbox);
}
#endif  // !VIO_EMUL

#ifndef VIO_EMUL  // (the host emulation build only takes the factor evaluations and reductions above: pnp_core.h)
// =====================================================================================================
// Evaluation: cost, and (jac) H -> App / Dss / Css / Asp (unfactored reduced system), WTf, hff, gp, gf, dp
// =====================================================================================================
constexpr int kPanelTiles = 5;   // pose matrices of up to 80 rows (W <= 12) keep the fill tiles of the band in registers
constexpr int kRowLen = 14;      // marginalization phase: staged Jacobian row Ji(6) Jj(6) r Jl
constexpr int kSlotStride = 29;  // marginalization phase: two rows of 14 + 1 pad (odd stride: conflict-free LDS writes)
// Solver: a staged row is [Ji(6) | Jj(3..5) | r]: the translation part of the target frame's Jacobian is the negated
// translation part of the host's (projection_facor.cpp:52-73: both are reduce * r_ic^T R_j^T up to the sign) and is
// rebuilt by the operand fetch of the Gram product; the landmark column only feeds per-landmark sums that the factor
// threads form themselves. 21 doubles per factor instead of 29: fewer, larger staging chunks.
constexpr int kGRow = 10;
constexpr int kGSlot = 2 * kGRow + 1;
constexpr int kGramPiece = 30;     // slots (15 matrix instructions) of one Gram piece at most: two operand batches, one flush
constexpr int kGramMinChunk = 32;  // a staging chunk never holds fewer slots than this (gram_chunk_slots)

// Staging slots per pass of the Jacobian evaluation (batch.h stage_chunk_slots computes the same on the host).
template <class WK>
VIO_DEV int gram_chunk_slots(const Ctx &cx, const WK &w) {
  int CH = (w.nstage / kGSlot) & ~1;
  if (CH >= (int)cx.nt) CH -= CH % (int)cx.nt;  // whole rounds of the workgroup: no chunk ends in a nearly empty pass
  return CH;
}

// The Gram pieces of a window, once per solve (the bucket layout does not change during it). A (host, target) bucket is cut at
// the staging-chunk boundaries and into runs of kGramPiece slots: piece = slot offset inside its chunk | slots << 10 |
// host << 15 | target << 21, in slot order; gstart[c] = first piece of chunk c. The waves of a workgroup take the pieces of a
// chunk round-robin (projections_jac): the buckets themselves are very uneven -- a dozen pairs of neighbouring frames hold 30 to
// 90 factors each, forty others 3 to 16 -- and with a bucket per wave (rounds 3-5) the wave with the 87-factor bucket ran 44
// matrix instructions while the others waited at the barrier.
template <class WK>
VIO_DEV void build_gram_pieces(const Ctx &cx, const WinView &v, WK &w) {
  const int CH = gram_chunk_slots(cx, w), nchunks = (v.nslots + CH - 1) / CH;
  ldsi cnt = reinterpret_cast<ldsi>(w.stage);  // [npairs] pieces of every bucket, then [nchunks + 1] pieces per chunk
  ldsi cc = cnt + v.npairs;
  VIO_PARFOR(c, nchunks + 1) cc[c] = 0;
  VIO_SYNC();
  VIO_PARFOR(b, v.npairs) {
    const int s0 = v.pair_s0[b], s1 = v.pair_s1[b];
    int n = 0;
    for (int c = s0 / CH; c * CH < s1; c++) {
      const int lo = s0 > c * CH ? s0 : c * CH, hi = s1 < (c + 1) * CH ? s1 : (c + 1) * CH;
      const int np_ = (hi - lo + kGramPiece - 1) / kGramPiece;
      n += np_;
#ifdef VIO_HOST_BUILD
      cc[c + 1] += np_;
#else
      __hip_atomic_fetch_add(cc + c + 1, np_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    }
    cnt[b] = n;
  }
  VIO_SYNC();
  VIO_PARFOR(b, v.npairs) {
    int idx = 0;
    for (int q = 0; q < b; q++) idx += cnt[q];
    const int s0 = v.pair_s0[b], s1 = v.pair_s1[b], ht = (v.pair_h[b] << 15) | (v.pair_t[b] << 21);
    for (int c = s0 / CH; c * CH < s1; c++) {
      const int lo = s0 > c * CH ? s0 : c * CH, hi = s1 < (c + 1) * CH ? s1 : (c + 1) * CH;
      for (int p0 = lo; p0 < hi; p0 += kGramPiece)
        v.gpiece[idx++] = (p0 - c * CH) | ((hi - p0 < kGramPiece ? hi - p0 : kGramPiece) << 10) | ht;
    }
  }
  VIO_PARFOR(c, nchunks + 1) {
    int sacc = 0;
    for (int q = 0; q <= c; q++) sacc += cc[q];
    v.gstart[c] = sacc;
  }
  VIO_SYNC();
  // Balance: the waves take the pieces of a chunk round-robin and meet at a barrier, so a chunk's Gram phase lasts as long as its
  // busiest wave -- in slot order one wave regularly drew the long pieces (wave 0: 242 k against 188 k cycles of matrix instructions
  // per solve). The pieces of every chunk are put in descending order of their length and dealt out in a snake (rounds of nw pieces:
  // wave 0 .. nw - 1, then nw - 1 .. 0): longest-first, the classic greedy for equal finishing times. Once per solve, in LDS.
  {
    int total = 0;
    for (int c = 0; c <= nchunks; c++) total += cc[c];
    ldsi pd = cc + nchunks + 2, pc = pd + total;  // descriptors and chunk of every piece (the staging area has room for thousands)
    if ((int)(pc - reinterpret_cast<ldsi>(w.stage)) + total <= 2 * w.nstage) {
      VIO_PARFOR(i, total) {
        pd[i] = v.gpiece[i];
        int c = 0, lo = 0;
        while (c < nchunks && lo + cc[c + 1] <= i) lo += cc[c + 1], c++;
        pc[i] = c | (lo << 8);
      }
      VIO_SYNC();
      const int nw = (int)cx.nt >> 6;
      VIO_PARFOR(i, total) {
        const int c = pc[i] & 255, lo = pc[i] >> 8, cnt = cc[c + 1];
        const int len = (pd[i] >> 10) & 31;
        int rank = 0;
        for (int q = lo; q < lo + cnt; q++) {
          const int lq = (pd[q] >> 10) & 31;
          rank += (lq > len || (lq == len && q < i)) ? 1 : 0;
        }
        const int g = rank / nw, k = rank - g * nw;
        const bool whole = (g + 1) * nw <= cnt;  // (the last, partial round keeps its order: its positions must stay below cnt)
        const int pos = g * nw + ((whole && (g & 1)) ? nw - 1 - k : k);
        v.gpiece[lo + pos] = pd[i];
      }
      VIO_SYNC();
    }
  }
}

// Projection factors with Jacobians. The (not yet assembled) matrix buffer is used as a staging area: every factor
// writes its two robustified Jacobian rows into its slot of the (host,target)-bucketed order; each bucket's
// J^T J / J^T r is then ONE Gram product on the matrix cores (the operand fetch of an MFMA step is 64 consecutive
// doubles of the staging area, and A and B are the same registers), instead of ~90 scattered atomics per factor.
// Per-feature sums (host coupling w_h, H_ff, g_f) are gathered by one thread per feature. Returns the cost partial.
template <class WK>
VIO_DEV double projections_jac(const Ctx &cx, const WinView &v, WK &w, cldsd pose, cldsd feat,
                               bool /*later_eval*/, int share = 0, int nshare = 1, bool host_tail = true) {
  const double bb = v.cauchy_b, cc = 1.0 / bb;
  double cost = 0.0;
  auto G = w.stage;  // (the reduced matrix is assembled after the last chunk: its buffer stages the Jacobian rows)
  const int CH = gram_chunk_slots(cx, w);
  // Per-feature sums (host coupling w_h = sum Ji^T Jl, H_ff = sum Jl^T Jl, g_f = sum Jl^T r over the feature's factors)
  // are gathered with LDS atomics by the factor threads themselves. The six components of w_h use six F-vectors that
  // are dead whenever Jacobians are evaluated (the candidate, the step, the Gauss-Newton step and the e / 1/e / g/e
  // scratch of the previous linear solve are all recomputed before their next use): see evaluate().
  ldsd whv[6] = {w.cfeat, w.stf, w.gnf, w.tf, w.ef, w.einv};
  // (cooperative windows: chunk share, share + nshare, ... -- the host packer keeps a bucket inside one chunk, so every
  // bucket's off-diagonal block still has one writer)
  // first piece of every chunk, one chunk per lane (read ahead of the chunk loop: a dependent global round trip at the head of
  // every chunk's Gram phase otherwise); chunks past the 63rd are read where they are needed
  const int g_lane = VIO_TID(cx) & 63, g_nch = (v.nslots + CH - 1) / CH;
  const int g_start = v.gstart[g_lane <= g_nch ? g_lane : 0];
  // The slot records of a chunk (one global round trip) are fetched one chunk ahead: the fetch of chunk c + 1 travels behind the
  // arithmetic of chunk c and its Gram phase (one slot per work-item and chunk whenever the chunk is the workgroup's size).
  const bool one_pass = CH <= (int)cx.nt;
  int rec_n = -1;
  double pij_n[6] = {0, 0, 0, 0, 0, 0};
  if (one_pass) {
    const int sl0 = share * CH + VIO_TID(cx);
    if (sl0 < v.nslots && VIO_TID(cx) < CH) {
      rec_n = v.srec_i[sl0];
#pragma unroll
      for (int c = 0; c < 6; c++) pij_n[c] = v.srec_d[6 * (size_t)sl0 + c];
    }
  }
  for (int c0 = share * CH; c0 < v.nslots; c0 += nshare * CH) {
    stamp(cx, ST_P_ZERO);
    const int nsl = v.nslots - c0 < CH ? v.nslots - c0 : CH;
    // this chunk's Gram pieces: the descriptors of this wave's first round (piece p_lo + wave + nw lane in lane `lane`) are fetched
    // HERE, ahead of the factor arithmetic -- fetched behind the barrier they were a dependent global round trip (~1.2 k cycles) at
    // the head of every chunk's Gram phase
    const int g_ci = c0 / CH;
    int p_lo, p_hi;
    if (g_ci < 63) p_lo = __builtin_amdgcn_readlane(g_start, g_ci), p_hi = __builtin_amdgcn_readlane(g_start, g_ci + 1);
    else p_lo = v.gstart[g_ci], p_hi = v.gstart[g_ci + 1];
    const int g_wave = __builtin_amdgcn_readfirstlane(VIO_TID(cx) >> 6), g_nw = (int)cx.nt >> 6;
    const int g_pl0 = p_lo + g_wave + g_lane * g_nw;
    const int m_desc0 = g_pl0 < p_hi ? v.gpiece[g_pl0] : 0;
    VIO_PARFOR(slot, nsl) {  // slot order: every lane of every wave has a factor (bar the odd tails)
      int rec;
      double pij[6];
      if (one_pass) {
        rec = rec_n;
#pragma unroll
        for (int c = 0; c < 6; c++) pij[c] = pij_n[c];
        const int sln = c0 + nshare * CH + slot;  // this work-item's slot of the next chunk
        rec_n = -1;
        if (sln < v.nslots) {
          rec_n = v.srec_i[sln];
#pragma unroll
          for (int c = 0; c < 6; c++) pij_n[c] = v.srec_d[6 * (size_t)sln + c];
        }
      } else {
        rec = v.srec_i[c0 + slot];
#pragma unroll
        for (int c = 0; c < 6; c++) pij[c] = v.srec_d[6 * (size_t)(c0 + slot) + c];
      }
      if (rec < 0) continue;
      const int h = rec & 255, t = (rec >> 8) & 255, f = rec >> 16;
      double r[2], Ji[12], Jj[12], Jl[2];
      projection_eval_rot(v.s_info, w.rot + 9 * h, pose + 7 * h, w.rot + 9 * t, pose + 7 * t, w.rot + 9 * (v.P + 1), w.ex,
                          feat[f], pij, pij + 3, true, r, Ji, Jj, Jl);
      double sq = r[0] * r[0] + r[1] * r[1];
      double sum = 1.0 + sq * cc;
      cost += 0.5 * bb * log(sum);
      double sr = rsqrt_f(sum);  // Corrector: rho'' < 0 => scale by sqrt(rho') = 1 / sqrt(1 + s / b)   (sum >= 1)
      auto g = G + slot * kGSlot;
#pragma unroll
      for (int rr = 0; rr < 2; rr++) {
#pragma unroll
        for (int c = 0; c < 6; c++) g[rr * kGRow + c] = Ji[rr * 6 + c] * sr;
#pragma unroll
        for (int c = 3; c < 6; c++) g[rr * kGRow + 3 + c] = Jj[rr * 6 + c] * sr;
        g[rr * kGRow + 9] = r[rr] * sr;
      }
      // target-frame coupling w_t = Jj^T Jl: one writer per (feature, frame)
      const double s2 = sr * sr;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double val = (Jj[c] * Jl[0] + Jj[6 + c] * Jl[1]) * s2;
        v.WTf[f * v.n6cap + 6 * t + c] = val;
      }
      VIO_ATOMIC_ADD(w.hff + f, (Jl[0] * Jl[0] + Jl[1] * Jl[1]) * s2);
      VIO_ATOMIC_ADD(w.gf + f, (Jl[0] * r[0] + Jl[1] * r[1]) * s2);
#pragma unroll
      for (int c = 0; c < 6; c++) VIO_ATOMIC_ADD(whv[c] + f, (Ji[c] * Jl[0] + Ji[6 + c] * Jl[1]) * s2);
    }
    VIO_SYNC();
    stamp(cx, ST_P_FACT);
    {
      const int tid_ = VIO_TID(cx), wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), nw = cx.nt >> 6, lane = tid_ & 63;
      const int li = lane & 15, kq = lane >> 4;
      // Where this lane's four accumulator elements (rows kq + 4 r4, column li of G^T G) go is a property of the lane,
      // not of the bucket: one LDS target  base + (h or t) * mul  (diagonal pose blocks in ppd, J^T r in gp) or one
      // entry of the bucket's off-diagonal block in PP. Worked out once, a flush is 4 LDS atomics + 4 stores under
      // two predicates each instead of a dozen divergent branches per element.
      ldsd f_base[4];
      int f_mul[4], f_goff[4];
      bool f_t[4], f_lds[4], f_glb[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        const int row = kq + 4 * r4, col = li;
        f_lds[r4] = false, f_glb[r4] = false, f_t[r4] = false, f_mul[r4] = 36, f_goff[r4] = 0, f_base[r4] = w.ppd;
        if (row < 6) {
          if (col <= row) f_lds[r4] = true, f_base[r4] = w.ppd + row * 6 + col;
        } else if (row < 12) {
          if (col < 6) f_glb[r4] = true, f_goff[r4] = (row - 6) * 6 + col;
          else if (col < 12 && col <= row) f_lds[r4] = true, f_t[r4] = true, f_base[r4] = w.ppd + (row - 6) * 6 + (col - 6);
        } else if (row == 12) {
          if (col < 6) f_lds[r4] = true, f_mul[r4] = kBS, f_base[r4] = w.gp + col;
          else if (col < 12) f_lds[r4] = true, f_t[r4] = true, f_mul[r4] = kBS, f_base[r4] = w.gp + col - 6;
        }
      }
      // Pieces (round 6, build_gram_pieces): the waves take the pieces of this chunk round-robin; a piece is at most 8 matrix
      // instructions on ONE batch of operand reads, the next piece's reads are in flight while this one multiplies. Pieces of one
      // bucket meet through atomics -- LDS for the diagonal blocks and the gradient, no-return global atomics for the off-diagonal
      // block in PP, which every linearization finds zeroed (evaluate()).
      // column li of G = [Ji | Jj | r]: staged entry; its sign (the translation part of the target's Jacobian is the negated
      // host's) is applied to the PRODUCT: element (row, col) of G^T G carries sign(row) sign(col)
      const int src = li < 6 ? li : li < 9 ? li - 6 : li < 12 ? li - 3 : li < 13 ? 9 : 0;
      double sgn[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        const int row = kq + 4 * r4;
        sgn[r4] = ((row >= 6 && row < 9) != (li >= 6 && li < 9)) ? -1.0 : 1.0;
      }
      auto gl = G + (kq >> 1) * kGSlot + (kq & 1) * kGRow + src;  // this lane's entry of the chunk's first two factors
      // issue: the operand reads of a piece's first eight steps (raw: steps past the piece read its last step again, nothing
      // consumes them)
      auto issue = [&](int desc, double (&a)[8]) {
        const int off = desc & 1023, n = (desc >> 10) & 31, last = ((n + 1) >> 1) - 1;
        auto g = gl + off * kGSlot;
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = g[(j < last ? j : last) * 2 * kGSlot];
      };
      // consume: the products of a piece and its flush
      auto consume = [&](int desc, double (&a)[8]) {
        const int n = (desc >> 10) & 31, h = (desc >> 15) & 63, t = (desc >> 21) & 63;
        const int total = (n + 1) >> 1;  // matrix instructions: two factors each, an odd last factor is a masked half step
        const int full = n >> 1;         // ... of which `full` take both factors: nothing to mask inside the loops
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        // the odd last factor: the step's operand (every batch repeats the piece's last step in its unused entries: entry 7 of the
        // batch that holds it) with the lanes kq >= 2 -- they read the padding slot behind the bucket -- zeroed, once per piece
        double xo = 0.0;
        if (n & 1) xo = kq < 2 ? a[full < 8 ? 7 : 0] : 0.0;  // (full >= 8: taken from the second batch below)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if (j < full) acc = mfma_f64(a[j], a[j], acc);  // (uniform)
        }
        if (kGramPiece > 16 && total > 8) {  // (the second half of a long piece: its own batch of reads)
          const int off = desc & 1023, last = total - 1;
          auto g = gl + off * kGSlot;
          double a2[8];
#pragma unroll
          for (int j = 0; j < 8; j++) a2[j] = g[(8 + j < last ? 8 + j : last) * 2 * kGSlot];
          VIO_SCHED_FENCE();
          if (n & 1) xo = kq < 2 ? a2[7] : 0.0;
#pragma unroll
          for (int j = 0; j < 8; j++) {
            if (8 + j < full) acc = mfma_f64(a2[j], a2[j], acc);
          }
        }
        if (n & 1) acc = mfma_f64(xo, xo, acc);
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
          const double val = acc[r4] * sgn[r4];
          if (f_lds[r4]) VIO_ATOMIC_ADD(f_base[r4] + (f_t[r4] ? t : h) * f_mul[r4], val);
          if (f_glb[r4]) {
            const int rt = 6 * t + kq + 4 * r4 - 6, ch = 6 * h + li;
            atomic_add_noret(v.PP + (t > h ? tri_at(rt, ch) : tri_at(ch, rt)), val);
          }
        }
      };
      for (int pb = p_lo + wave; pb < p_hi; pb += 64 * nw) {  // (rounds of 64 pieces per wave: one is enough below 64 nw pieces per chunk)
        const int pl = pb + lane * nw;
        const int m_desc = pb == p_lo + wave ? m_desc0 : (pl < p_hi ? v.gpiece[pl] : 0);  // (uniform choice; later rounds: > 64 nw pieces in a chunk)
        const int cnt = (p_hi - pb + nw - 1) / nw < 64 ? (p_hi - pb + nw - 1) / nw : 64;
        double A[8], Bq[8];
        issue(__builtin_amdgcn_readlane(m_desc, 0), A);
        for (int k = 0; k < cnt; k += 2) {
          const int d0 = __builtin_amdgcn_readlane(m_desc, k);
          const int d1 = k + 1 < cnt ? __builtin_amdgcn_readlane(m_desc, k + 1) : 0;
          if (k + 1 < cnt) issue(d1, Bq);
          VIO_SCHED_FENCE();
          consume(d0, A);
          if (k + 1 < cnt) {
            if (k + 2 < cnt) issue(__builtin_amdgcn_readlane(m_desc, k + 2), A);
            VIO_SCHED_FENCE();
            consume(d1, Bq);
          }
        }
      }
    }
    VIO_SYNC();
    stamp(cx, ST_P_GRAM);
    stamp(cx, ST_P_FEAT);
  }
  if (!host_tail) return cost;  // (cooperative windows: the owner writes it after the partial sums are merged)
  // host-frame coupling of every feature: LDS sums -> both layouts of W
  VIO_PARFOR(f, v.F) {
    const int h = w.fh[f];
    if (h >= 0)
      for (int c = 0; c < 6; c++) {
        const double val = whv[c][f];
        v.WTf[(size_t)f * v.n6cap + 6 * h + c] = val;
      }
  }
  return cost;
}

// keep_aux (cost-only evaluation of a candidate): also leave the raw IMU Jacobians behind; reuse_aux (the Jacobian
// evaluation at a point whose cost-only evaluation just ran, i.e. an accepted step): the prior's dx and J^T r (prdx, prr)
// and the raw IMU residuals / Jacobians (imu_r, imu_J) of that evaluation are still valid and are not formed again --
// two serial few-thread phases and a mat-vec less per accepted iteration.
template <class WK>
VIO_DEV double evaluate(const Ctx &cx, const WinView &v, WK &w, cldsd pose, cldsd sb,
                        cldsd feat, bool jac, bool have_scale = false, bool reuse_aux = false, bool keep_aux = false) {
  const int np = v.np;
  double cost = 0.0;  // per-thread partial, reduced at the end
  VIO_PARFOR(i, v.P + v.has_loop + 1) {  // rotation matrices for the projection factors (consumed after a barrier)
    const bool is_ex = i == v.P + v.has_loop;
    double R[9];
    if (is_ex) qtoR(Quat{w.ex[3], w.ex[4], w.ex[5], w.ex[6]}, R);
    else qtoR(Quat{pose[7 * i + 3], pose[7 * i + 4], pose[7 * i + 5], pose[7 * i + 6]}, R);
    auto dst = w.rot + 9 * (is_ex ? v.P + 1 : i);
    for (int k = 0; k < 9; k++) dst[k] = R[k];
  }
  if (jac) {
    const int nF = v.P + v.has_loop;
    VIO_PARFOR(q, np) w.gp[q] = 0.0;
    VIO_PARFOR(q, v.F) {
      w.gf[q] = 0.0, w.hff[q] = 0.0;
      w.cfeat[q] = w.stf[q] = w.gnf[q] = w.tf[q] = w.ef[q] = w.einv[q] = 0.0;  // w_h accumulators (projections_jac)
    }
    if (!have_scale) {
      // the (feature, frame) entries every evaluation writes are the same; they are all assigned (not accumulated)
      // when each feature has its own thread, so zeroing is needed once (first evaluation of the solve)
      VIO_PARFOR(q, v.F * v.n6cap) v.WTf[q] = 0.0;
    }
    VIO_PARFOR(q, nF * 36) w.ppd[q] = 0.0;
    VIO_SYNC();
#ifndef VIO_HOST_BUILD
    if (cx.coop > 1) {
      // cooperative window: the evaluation point goes to the window's scratch, every workgroup stages and multiplies the
      // Jacobian-row chunks of its share; the owner then adds the helpers' partial sums (gradient, diagonal pose blocks,
      // per-landmark sums, cost) to its own and writes the landmarks' host-frame coupling, which needs the complete sums
      const CoopLayout L = CoopLayout::make(v.Pcap, v.Fcap, v.nblk_cap);
      const int nFr = v.P + v.has_loop;
      VIO_PARFOR(q, 7 * nFr) v.coop[L.o_pose + q] = pose[q];
      VIO_PARFOR(q, v.F) v.coop[L.o_feat + q] = feat[q];
      VIO_PARFOR(q, 7) v.coop[L.o_ex + q] = w.ex[q];
      coop_post(cx, v, COOP_EVAL);
      cost += projections_jac(cx, v, w, pose, feat, have_scale, 0, cx.coop, false);
      coop_wait_helpers(cx, v);
      ldsd whv[6] = {w.cfeat, w.stf, w.gnf, w.tf, w.ef, w.einv};
      const size_t fs = ((size_t)v.Fcap + 7) & ~(size_t)7;
      for (int m = 1; m < cx.coop; m++) {
        const double *pr = v.coop + L.o_part + (size_t)(m - 1) * L.part;
        VIO_PARFOR(q, np) w.gp[q] += pr[L.p_gp + q];
        VIO_PARFOR(q, nFr * 36) w.ppd[q] += pr[L.p_ppd + q];
        VIO_PARFOR(f, v.F) {
          w.hff[f] += pr[L.p_f + f], w.gf[f] += pr[L.p_f + fs + f];
#pragma unroll
          for (int c = 0; c < 6; c++) whv[c][f] += pr[L.p_f + (2 + c) * fs + f];
        }
        if (cx.tid == 0) cost += pr[L.p_cost];
      }
      VIO_SYNC();
      VIO_PARFOR(f, v.F) {
        const int h = w.fh[f];
        if (h >= 0)
          for (int c = 0; c < 6; c++) v.WTf[(size_t)f * v.n6cap + 6 * h + c] = whv[c][f];
      }
    } else
#endif
    {
      stamp(cx, ST_E_HEAD);
      cost += projections_jac(cx, v, w, pose, feat, have_scale);
    }
    stamp(cx, ST_EVAL_PROJ);
    // ---- the rest of the linearization in three barrier intervals, every global fetch of an interval in flight at once
    //      (each dependent round trip costs ~3.5 k cycles here; the first versions took six of them, one phase at a time):
    //  1  reduced matrix <- the prior's H0 + the projection factors' off-diagonal pose blocks; prior dx; raw IMU residuals / Jacobians on the LAST W work-items
    //     (one lane per factor, ~1.5 k dependent operations: they run beside the copy instead of in a phase of their own)
    //  2  prr = b0 + H0 dx (MarginalizationFactor::Evaluate as J^T r); diagonal pose blocks of the projection factors -> App
    //  3  prior cost and gradient; IMU factors on the matrix cores (Gram products, gradient, cost)
    const int napp = (int)tri_doubles(v.nrows), nband = 2 * v.P * kSS;
    const int n = v.prior_n, nt_ = (int)cx.nt;
    const bool do_prior = n > 0 && !reuse_aux;
    {
      const int tid_ = VIO_TID(cx);
      {
        // the pose matrix starts as the prior's H0 (constant during the solve, laid out once by setup_prior) plus the
        // off-diagonal blocks of the projection factors (PP, same layout), the band as the prior's part alone: straight
        // sums / copies, two dozen loads in flight per lane, no zeroing pass, no scatter with index decoding
        constexpr int kU = 12;
        const bool pr = n > 0;
        for (int q0 = tid_; q0 < napp; q0 += kU * nt_) {
          double x[kU], y[kU];
#pragma unroll
          for (int u = 0; u < kU; u++) {
            const int q = q0 + u * nt_, qc = q < napp ? q : 0;
            x[u] = pr ? v.AppPr[qc] : 0.0, y[u] = v.PP[qc];
          }
          VIO_SCHED_FENCE();
#pragma unroll
          for (int u = 0; u < kU; u++) {
            const int q = q0 + u * nt_;
            if (q < napp) w.App[q] = x[u] + y[u], v.PP[q] = 0.0;  // (PP is added to atomically: every linearization finds it zeroed)
          }
        }
        for (int q0 = tid_; q0 < nband; q0 += kU * nt_) {
          double x[kU];
#pragma unroll
          for (int u = 0; u < kU; u++) {
            const int q = q0 + u * nt_;
            x[u] = pr ? v.AppPr[napp + (q < nband ? q : 0)] : 0.0;
          }
          VIO_SCHED_FENCE();
#pragma unroll
          for (int u = 0; u < kU; u++) {
            const int q = q0 + u * nt_;
            if (q < nband) w.Dss[q] = x[u];  // (Css follows Dss)
          }
        }
      }
      VIO_PARFOR(q, v.P * kAS) w.AspI[q] = 0.0;
      // (the zero the panel steps read where a tile has no coupling entry: the pad behind the LDS copy of the coupling, panel_step_static)
      if (w.asp_ring) {
        if (tid_ < 4) w.aspring[2 * kAS + tid_] = 0.0;
      } else if (w.asp_lds && tid_ < 16) {
        w.AspI[v.Pcap * kAS + tid_] = 0.0;
      }
      if (do_prior) {
        VIO_PARFOR(b, v.prior_nb) {
          int kind = v.pr_kind[b], idx = v.pr_index[b], o = v.pr_offset[b];
          const double *x0 = v.pr_x0 + 9 * b;
          if (kind == 0) prior_block_dx(7, pose + 7 * idx, x0, w.prdx + o);
          else if (kind == 1) prior_block_dx(9, sb + 9 * idx, x0, w.prdx + o);
          else prior_block_dx(7, w.ex, x0, w.prdx + o);
        }
        VIO_PARFOR(i, n) w.prr[i] = v.prb0[i];
      }
      if (!reuse_aux) {
        const int f = nt_ - 1 - tid_;
        if (f < v.W)
          imu_eval_raw(v.gravity, v.preint + f * kPreintDoubles, pose + 7 * f, sb + 9 * f, pose + 7 * (f + 1), sb + 9 * (f + 1),
                       v.imu_r + f * 15, v.imu_J + f * 450);
      }
    }
    VIO_SYNC();
    stamp(cx, ST_IMU_RAW);
    {
      // H0 dx: (column, part) items like dense_matvec_cols, ONE strip per item when the prior fits (75 rows: 225 items)
      const int tid_ = VIO_TID(cx);
      constexpr int kStrip = 25;
      int nparts = do_prior ? nt_ / n : 1;
      const int need = (n + 4 * kStrip - 1) / (4 * kStrip);
      nparts = nparts < need ? need : (nparts > 8 ? 8 : nparts);
      const int per = do_prior ? (n + nparts - 1) / nparts : 0;
      const bool one_strip = do_prior && n * nparts <= nt_ && per <= kStrip;
      if (one_strip && tid_ < n * nparts) {
        const int hpart = tid_ / n, hc = tid_ - hpart * n;
        const int hk0 = hpart * per, hnb = hk0 + per < n ? per : n - hk0;
        if (hnb > 0) {
          double hs[kStrip], sacc = 0;
#pragma unroll
          for (int j = 0; j < kStrip; j++) hs[j] = v.prH0[(size_t)(hk0 + (j < hnb ? j : 0)) * n + hc];
          VIO_SCHED_FENCE();
#pragma unroll
          for (int j = 0; j < kStrip; j++) sacc += (j < hnb ? hs[j] : 0.0) * w.prdx[hk0 + (j < hnb ? j : 0)];
          VIO_ATOMIC_ADD(w.prr + hc, sacc);
        }
      }
      if (do_prior && !one_strip)
        dense_matvec_cols(cx, v.prH0, n, w.prdx, [&](int i, double sacc) { VIO_ATOMIC_ADD(w.prr + i, sacc); });  // prr = b0 + H0 dx = J^T r
      const int nF = v.P + v.has_loop;
      VIO_PARFOR(q, nF * 36) {
        const int a = q / 36, e = q - a * 36, r = e / 6, c = e - r * 6;
        if (r >= c) VIO_ATOMIC_ADD(w.App + tri_at(6 * a + r, 6 * a + c), w.ppd[q]);
      }
    }
    VIO_SYNC();
    stamp(cx, ST_EVAL_PRIOR);
    if (n > 0) VIO_PARFOR(i, n) {
      const double r0 = v.pr_r[i], b0 = v.prb0[i];
      cost += 0.5 * r0 * r0 + 0.5 * w.prdx[i] * (w.prr[i] + b0);  // |r0|^2 / 2 + b0 . dx + dx . H0 dx / 2
      const int pa = w.prcol[i];
      if (pa >= 0) VIO_ATOMIC_ADD(w.gp + kBS * (pa >> 8) + (pa & 255), w.prr[i]);
    }
    {
      // One wave per IMU factor on the matrix cores. B = [Jraw | r | 0] (15 x 32, k padded to 16):
      //   T = info B            (2 column tiles x 4 k-steps)
      //   G = B^T T             (lower tiles (0,0), (1,0), (1,1)): G[a][b] = (J^T info J)_ab, G[30][b] = (J^T info r)_b,
      //                         G[30][30] = r^T info r (twice the factor's cost)
      // The f64 accumulator layout of T (lane l, element r <-> T[(l>>4)+4r][l&15]) IS the B-operand layout of k-step r,
      // so T never leaves the registers.
      const int tid_ = VIO_TID(cx), wave = tid_ >> 6, nw = cx.nt >> 6, lane = tid_ & 63;
      const int n = lane & 15, kq = lane >> 4;
      // (the operands of a wave's NEXT factor are fetched -- raw, clamped addresses -- before this one's products: one global round
      // trip per evaluation on the wave's path instead of one per factor)
      struct ImuOps {
        double a[4], b0[4], b1[4];
      };
      auto imu_fetch = [&](int f, ImuOps &o) {
        const double *info = v.imu_info + f * 225, *Jr = v.imu_J + f * 450, *rr = v.imu_r + f * 15;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
          const int k = 4 * s4 + kq, kc = k < 15 ? k : 0;
          o.a[s4] = info[(n < 15 ? n : 0) * 15 + kc];
          o.b0[s4] = Jr[kc * 30 + n];
          o.b1[s4] = n == 14 ? rr[kc] : Jr[kc * 30 + 16 + (n < 14 ? n : 0)];
        }
      };
      ImuOps nxt;
      if (wave < v.W) imu_fetch(wave, nxt);
      for (int f = wave; f < v.W; f += nw) {
        const ImuOps cur = nxt;
        if (f + nw < v.W) imu_fetch(f + nw, nxt);
        VIO_SCHED_FENCE();
        double av[4], bv[2][4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
          const bool kok = 4 * s4 + kq < 15;
          av[s4] = (kok && n < 15) ? cur.a[s4] : 0.0;
          bv[0][s4] = kok ? cur.b0[s4] : 0.0;
          bv[1][s4] = (kok && n < 15) ? cur.b1[s4] : 0.0;
        }
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
          const int row = kq + 4 * r4;  // within the tile
          // local index a in [0, 30): frame f components 0..14, then frame f + 1; tile (0,0): rows / cols 0..15
          if (row >= n) red_put(v, w, f + (row >= 15), row >= 15 ? row - 15 : row, f + (n >= 15), n >= 15 ? n - 15 : n, G00[r4], true);
          // tiles (1,x): rows 16..31 -> 16..29 are Jacobian columns, 30 is the residual row
          const int R = 16 + row;
          if (R < 30) {
            red_put(v, w, f + 1, R - 15, f + (n >= 15), n >= 15 ? n - 15 : n, G10[r4], true);
            if (n < 14 && R >= 16 + n) red_put(v, w, f + 1, R - 15, f + 1, n + 1, G11[r4], true);
          } else if (R == 30) {
            VIO_ATOMIC_ADD(w.gp + 15 * f + n, G10[r4]);
            if (n < 14) VIO_ATOMIC_ADD(w.gp + 15 * f + 16 + n, G11[r4]);
            if (n == 14) cost += 0.5 * G11[r4];
          }
        }
      }
    }
    VIO_SYNC();
    stamp(cx, ST_EVAL_IMU);
  } else {
    // ---- cost only (a candidate after a rejected step): prior, IMU factors, projection factors -----------------------
    const int n = v.prior_n;
    if (n > 0 && !reuse_aux) {
      VIO_PARFOR(b, v.prior_nb) {
        int kind = v.pr_kind[b], idx = v.pr_index[b], o = v.pr_offset[b];
        const double *x0 = v.pr_x0 + 9 * b;
        if (kind == 0) prior_block_dx(7, pose + 7 * idx, x0, w.prdx + o);
        else if (kind == 1) prior_block_dx(9, sb + 9 * idx, x0, w.prdx + o);
        else prior_block_dx(7, w.ex, x0, w.prdx + o);
      }
      VIO_PARFOR(i, n) w.prr[i] = v.prb0[i];
      VIO_SYNC();
      dense_matvec_cols(cx, v.prH0, n, w.prdx, [&](int i, double sacc) { VIO_ATOMIC_ADD(w.prr + i, sacc); });  // prr = b0 + H0 dx = J^T r
    }
    VIO_SYNC();
    if (n > 0) VIO_PARFOR(i, n) {
      const double r0 = v.pr_r[i], b0 = v.prb0[i];
      cost += 0.5 * r0 * r0 + 0.5 * w.prdx[i] * (w.prr[i] + b0);  // |r0|^2 / 2 + b0 . dx + dx . H0 dx / 2
    }
    stamp(cx, ST_COST_EVAL);
    if (!reuse_aux) {
      VIO_PARFOR(f, v.W) {
        imu_eval_raw(v.gravity, v.preint + f * kPreintDoubles, pose + 7 * f, sb + 9 * f, pose + 7 * (f + 1),
                     sb + 9 * (f + 1), v.imu_r + f * 15, keep_aux ? v.imu_J + f * 450 : nullptr);
      }
      VIO_SYNC();
    }
    stamp(cx, ST_D0);
    VIO_PARFOR(q, v.W * 15) {  // cost += r^T info r / 2
      int f = q / 15, r = q % 15;
      const double *info = v.imu_info + f * 225 + r * 15;
      const double *rr = v.imu_r + f * 15;
      double s = 0;
      for (int k = 0; k < 15; k++) s += info[k] * rr[k];
      cost += 0.5 * s * rr[r];
    }
    stamp(cx, ST_D1);
    // ---- projection factors, cost only: CauchyLoss rho = b log(1 + s / b) (CSI/loss_function.cc:72-79) ----------
    const double bb = v.cauchy_b, cc = 1.0 / bb;
    // slot order (the records built once per solve), the fetch of the next pass in flight while this one is evaluated:
    // a pass is otherwise one global round trip (thousands of cycles) followed by a few hundred cycles of arithmetic
    int sl = VIO_TID(cx);
    int rec = sl < v.nslots ? v.srec_i[sl] : -1;
    double pij[6];
#pragma unroll
    for (int c = 0; c < 6; c++) pij[c] = v.srec_d[6 * (size_t)(sl < v.nslots ? sl : 0) + c];
    while (sl < v.nslots) {
      const int nx = sl + (int)cx.nt;
      const int rec_n = nx < v.nslots ? v.srec_i[nx] : -1;
      double pn[6];
#pragma unroll
      for (int c = 0; c < 6; c++) pn[c] = v.srec_d[6 * (size_t)(nx < v.nslots ? nx : 0) + c];
      if (rec >= 0) {
        const int h = rec & 255, t = (rec >> 8) & 255, f = rec >> 16;
        double r[2];
        projection_eval_rot(v.s_info, w.rot + 9 * h, pose + 7 * h, w.rot + 9 * t, pose + 7 * t, w.rot + 9 * (v.P + 1), w.ex,
                            feat[f], pij, pij + 3, false, r, nullptr, nullptr, nullptr);
        cost += 0.5 * bb * log(1.0 + (r[0] * r[0] + r[1] * r[1]) * cc);
      }
      sl = nx, rec = rec_n;
#pragma unroll
      for (int c = 0; c < 6; c++) pij[c] = pn[c];
    }
    stamp(cx, ST_D2);
  }
  double total = block_sum(cx, cost);  // contains barriers
  if (jac) {
    // diag(H) -> Jacobi scaling (fixed at the first evaluation, trust_region_minimizer.cc:239-254) and the trust-region
    // diagonal D = sqrt(clamp(diag(J_s^T J_s))) of THIS linearization (dogleg_strategy.cc:98-115)
    VIO_PARFOR(i, np) {
      const int f = i / kBS, c = i - f * kBS;
      const double h = c < 6 ? w.App[tri_at(6 * f + c, 6 * f + c)] : w.Dss[f * kSS + (c - 6) * (kSB + 1)];
      if (!have_scale) w.sp[i] = rcp_f(1.0 + sqrt_f(h));
      const double sc = w.sp[i];
      w.dp[i] = sqrt_f(fmin(fmax(sc * sc * h, 1e-6), 1e32));
    }
    VIO_SYNC();
  }
  stamp(cx, ST_COST_EVAL);
  return total;
}

// =====================================================================================================
// Linear algebra on the block-lower matrix
// =====================================================================================================

// In place: Hm <- S Hm S + diag(Dp^2) on the pose side, then subtracts the landmark Schur term
// sum_f ws_f ws_f^T / e_f (ws = WT scaled by sp, sf). Also builds rhs (-> w.t1) = sp gp - sum_f ws_f gs_f / e_f.
// Returns false if some e_f <= 0.
// Ceres' trust-region diagonal of a landmark, D_f = sqrt(clamp(diag(J_s^T J_s))) = sqrt(clamp(s_f^2 H_ff)), and the scaled
// gradient in those units, g_f s_f / D_f (dogleg_strategy.cc:98-115). Both are functions of arrays that only change at a
// new linearization, so they are recomputed where needed instead of living in LDS (16 bytes per landmark that decide
// whether a window's matrix still fits next to its vectors).
template <class WK>
VIO_DEV double feat_d2(const WK &w, int f) {  // D_f^2 (clamped: always in [1e-6, 1e32])
  return fmin(fmax(w.sf[f] * w.sf[f] * w.hff[f], 1e-6), 1e32);
}
template <class WK>
VIO_DEV double feat_d(const WK &w, int f) {
  return sqrt_f(feat_d2(w, f));
}
template <class WK>
VIO_DEV double feat_gd(const WK &w, int f) {
  return w.sf[f] * w.gf[f] * rsqrt_f(feat_d2(w, f));
}

// pose-side scaled gradient g s / d (dogleg_strategy.cc:98-115), recomputed where it is used like feat_gd
template <class WK>
VIO_DEV double pose_gd(const WK &w, int i) {
  return w.sp[i] * w.gp[i] * rcp_f(w.dp[i]);
}

// u^T H u for u = S a (a in Ceres' scaled coordinates: vp poses, vf landmarks) on the UNFACTORED system left behind by
// evaluate(): u_p^T H_pp u_p + 2 u_f^T W^T u_p + sum_f H_ff u_f^2. This is |J_s a|^2 of the Cauchy-point formula
// (dogleg_strategy.cc:172-192); the first two versions evaluated it after the factorization through L, which forced the
// fill of the factor to be kept. Uses w.t1 (u_p, frame-major), w.xt (u_p by pose index) and w.tf as scratch.
template <class WK>
VIO_DEV double quad_form_H(const Ctx &cx, const WinView &v, WK &w, cldsd vp, cldsd vf, bool prepared = false, bool w_part = true) {
  const int np = v.np, F = v.F, n6 = v.n6, P = v.P;
  if (!prepared) {  // (prepared: the caller's pass that formed vp also left t1 / xt / tf = 0 behind, barrier included)
    VIO_PARFOR(f, F) w.tf[f] = 0.0;
    VIO_PARFOR(i, np) {
      const int f = i / kBS, c = i - f * kBS;
      const double u = w.sp[i] * vp[i];
      w.t1[i] = u;
      if (c < 6) w.xt[6 * f + c] = u;
    }
    VIO_SYNC();
  }
  int nparts, per;
  wt_parts((int)cx.nt, F, n6, nparts, per);
  // (w_part = false: the caller takes 2 u_f^T W^T u_p from the Schur sweep that follows, schur_ksplit5; tf stays zero here)
  if (w_part) VIO_PARFOR(q, F * nparts) {  // (W^T u_p)_f += sum_{a in part} W[f][a] u_p[a]
    const int part = q / F, f = q - part * F;
    const int a0 = part * per, a1 = a0 + per < n6 ? a0 + per : n6;
    double x[kWStrip], sacc = 0;
    for (int b0 = a0; b0 < a1; b0 += kWStrip) {
      const int nb = a1 - b0 < kWStrip ? a1 - b0 : kWStrip;
      wt_strip_load(v.WTf + (size_t)f * v.n6cap + b0, 1, nb, x);  // the lane's own contiguous strip of the feature-major W
#pragma unroll
      for (int j = 0; j < kWStrip; j++) sacc += (j < nb ? x[j] : 0.0) * w.xt[b0 + (j < nb ? j : 0)];
    }
    VIO_ATOMIC_ADD(w.tf + f, sacc);
  }
  stamp(cx, ST_Q_W);
  double acc = 0;
  // pose x pose (lower triangle, off-diagonal entries count twice): (row, quarter) items
  VIO_PARFOR(q, 4 * n6) {
    const int r = q >> 2, part = q & 3;
    auto row = w.App + tri_at(r, 0);
    double sacc = 0;
    for (int c = part; c < r; c += 4) sacc = fma(row[c], w.xt[c], sacc);
    sacc *= 2.0;
    if (part == 0) sacc = fma(row[r], w.xt[r], sacc);
    acc = fma(w.xt[r], sacc, acc);
  }
  // speed-bias band: D_k (lower triangle) and the coupling Css[k] = A(s_{k-1}, s_k)
  VIO_PARFOR(q, P * kSB) {
    const int k = q / kSB, r = q - k * kSB;
    cldsd D = w.Dss + k * kSS + r * kSB;
    const double ur = w.t1[kBS * k + 6 + r];
    double sacc = 0;
    for (int c = 0; c < r; c++) sacc = fma(D[c], w.t1[kBS * k + 6 + c], sacc);
    sacc = 2.0 * sacc + D[r] * ur;
    if (k >= 1) {  // row r of Css[k] belongs to s_{k-1}[r]
      cldsd C = w.Css + k * kSS + r * kSB;
      double s2 = 0;
      for (int c = 0; c < kSB; c++) s2 = fma(C[c], w.t1[kBS * k + 6 + c], s2);
      acc = fma(2.0 * w.t1[kBS * (k - 1) + 6 + r], s2, acc);
    }
    // speed-bias x pose: the IMU chain (LDS) and, for the block the prior keeps, the prior's row (global)
    const int alo = 6 * (k > 0 ? k - 1 : 0), aw = n6 - alo < kAW ? n6 - alo : kAW;
    auto A = w.AspI + (k * kSB + r) * kAW;
    double s3 = 0, ar[kAW];
    // (one batch of loads: the row may live in global scratch, where a rolled loop pays an L2 round trip per element)
#pragma unroll
    for (int jj = 0; jj < kAW; jj++) ar[jj] = A[jj < aw ? jj : 0];
    VIO_SCHED_FENCE();
#pragma unroll
    for (int jj = 0; jj < kAW; jj++) s3 = fma(jj < aw ? ar[jj] : 0.0, w.xt[alo + (jj < aw ? jj : 0)], s3);
    acc = fma(ur, sacc + 2.0 * s3, acc);
  }
  // the prior's speed-bias x pose block (global): (component, strip of 16 columns) items, one fetch batch per item
  VIO_PARFOR(q, kSB * v.nT) {
    const int c = q / v.nT, j0 = 16 * (q - c * v.nT);
    int kpr = -1;
    for (int k = 0; k < P; k++)
      if (w.sbr[2 * k + 1]) kpr = k;
    if (kpr < 0) continue;
    double xs[16], s3 = 0;
    const double *Ap = v.Apri + (size_t)c * v.jp + j0;
#pragma unroll
    for (int j = 0; j < 16; j++) xs[j] = Ap[j];  // (rows are jp = 16 nT long and zero outside the prior's poses)
    VIO_SCHED_FENCE();
#pragma unroll
    for (int j = 0; j < 16; j++) s3 = fma(xs[j], j0 + j < n6 ? w.xt[j0 + j] : 0.0, s3);
    acc = fma(2.0 * w.t1[kBS * kpr + 6 + c], s3, acc);
  }
  if (w_part) VIO_SYNC();  // tf complete
  VIO_PARFOR(f, F) {
    const double u = w.sf[f] * vf[f];
    acc = fma(u, fma(w.hff[f], u, 2.0 * w.tf[f]), acc);
  }
  return block_sum(cx, acc);
}

// C = (W E^-1) W^T (lower 16 x 16 tiles) and c = W (g / E) for a feature-major coupling matrix Wf [F][ldw] with n6 <= 80
// pose-type indices, K-split on the matrix cores: every wave owns a slice of the features and forms ALL lower tiles from it.
// A and B operands are the same 5 loads per k-step (A = W e^-1, B = W), so a chunk of 5 k-steps is 25 global loads (one
// latency) feeding 75 matrix instructions; the partial results of the waves meet through the callers' atomic adds:
// flush_tile(row, col <= row, value), flush_rhs(index, value). ge = g_f / E_f.
// up / tq (optional): the same sweep also leaves tq[f] = sum_a W[f][a] up[a] (the landmark coupling applied to a pose-index
// vector: the W part of the Cauchy point's quadratic form) -- five multiply-adds and a 16-lane DPP sum per k-step in the
// shadow of its 15 matrix instructions instead of another pass over W in global memory.
template <class FT, class FR>
VIO_DEV void schur_ksplit5(const Ctx &cx, const double *Wf, int ldw, int n6, int F, cldsd einv, cldsd ge, FT flush_tile,
                           FR flush_rhs, cldsd up = nullptr, ldsd tq = nullptr) {
  constexpr int kT = 5;  // row tiles the K-split form holds in registers (15 accumulators)
  const int tid_ = VIO_TID(cx), wave = tid_ >> 6, lane = tid_ & 63, nw = cx.nt >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int ksteps = (F + 3) / 4;
  const int ksw = (ksteps + nw - 1) / nw;
  const int s_begin = wave * ksw, s_end = s_begin + ksw < ksteps ? s_begin + ksw : ksteps;
  v4d acc[kT * (kT + 1) / 2];
  double rp[kT];  // the same fetch also yields this slice's part of W (g_f / E_f)
#pragma unroll
  for (int q = 0; q < kT * (kT + 1) / 2; q++) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < kT; t++) rp[t] = 0.0;
  double u5[kT];
#pragma unroll
  for (int t = 0; t < kT; t++) u5[t] = (tq && 16 * t + li < n6) ? up[16 * t + li] : 0.0;
  for (int s0 = s_begin; s0 < s_end; s0 += 5) {
    double wv[5][kT], ev[5], gv[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int f = 4 * (s0 + j) + kq;
      const int fc = (s0 + j < s_end && f < F) ? f : 0;
      ev[j] = einv[fc], gv[j] = ge[fc];
#pragma unroll
      for (int t = 0; t < kT; t++) {
        const int col = 16 * t + li;
        wv[j][t] = Wf[(size_t)fc * ldw + (col < n6 ? col : 0)];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int f = 4 * (s0 + j) + kq;
      const bool vf = s0 + j < s_end && f < F;
#pragma unroll
      for (int t = 0; t < kT; t++) {
        wv[j][t] = (vf && 16 * t + li < n6) ? wv[j][t] : 0.0;
        rp[t] = fma(wv[j][t], gv[j], rp[t]);
      }
      if (tq) {  // (uniform) row f of W against up: this lane's five columns, then the 16 lanes of the row (total in lane 15)
        double pq = wv[j][0] * u5[0];
#pragma unroll
        for (int t = 1; t < kT; t++) pq = fma(wv[j][t], u5[t], pq);
        pq += dpp_move_f64<0x111, 0xf>(pq);
        pq += dpp_move_f64<0x112, 0xf>(pq);
        pq += dpp_move_f64<0x114, 0xf>(pq);
        pq += dpp_move_f64<0x118, 0xf>(pq);
        if (li == 15 && vf) tq[f] = pq;
      }
#pragma unroll
      for (int ti = 0; ti < kT; ti++) {
        const double a = wv[j][ti] * ev[j];
#pragma unroll
        for (int tj = 0; tj <= ti; tj++) acc[ti * (ti + 1) / 2 + tj] = mfma_f64(a, wv[j][tj], acc[ti * (ti + 1) / 2 + tj]);
      }
    }
  }
#pragma unroll
  for (int ti = 0; ti < kT; ti++)
#pragma unroll
    for (int tj = 0; tj <= ti; tj++) {
      const int bcol = 16 * tj + li;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int arow = 16 * ti + kq + 4 * r;
        if (arow < n6 && bcol <= arow) flush_tile(arow, bcol, acc[ti * (ti + 1) / 2 + tj][r]);
      }
    }
#pragma unroll
  for (int t = 0; t < kT; t++) {
    const int a = 16 * t + li;
    if (a < n6 && rp[t] != 0.0) flush_rhs(a, rp[t]);
  }
}

// (App of the general path is global scratch on the device -- several workgroups may add to a tile: device scope, no return --, a
// plain array under the host emulation)
VIO_DEV void schur_add(double *p, double v) { atomic_add_noret(p, v); }
#ifndef VIO_HOST_BUILD
VIO_DEV void schur_add(ldsd p, double v) { atomic_add(p, v); }
#endif

// C = (W E^-1) W^T over the lower 16 x 16 tiles of an index space of any size (more than the five tile rows schur_ksplit5 keeps in
// registers), W feature-major [F][ldw], on the matrix cores in BLOCKS of 2 x 2 tiles: a k-step of a block is FOUR loads (two strips
// of W as A operands, two as B) for four matrix instructions (three on the diagonal, whose upper tile is not needed). With one tile
// per wave (rounds 3-6) a matrix instruction took two loads of its own and the CU's address unit -- ~16 cycles per load instruction,
// eight waves -- was busy twice as long as the matrix pipes. K-split: with fewer blocks than waves (W = 20: ten blocks for the 32
// waves of a window on four workgroups) or a ragged last round the feature range of a block is cut into ksplit parts (whole fetch
// chunks) that meet through atomic adds: the split that minimises rounds x chunks per part, ceil(nblocks k / waves) x ceil(chunks / k)
// over k = 1 .. 4. share / nshare: this workgroup's units of a cooperative window.
// flush(row, col <= row, value, parts): parts = true -> the element has several writers, add atomically.
// ge / flush_rhs: the same fetch also yields c = W (g_f / E_f) -- the diagonal blocks, which have a matrix instruction to spare, multiply
// the two strips they load by ge[f] on the way; flush_rhs(index, partial) is called by every lane group and part (add atomically).
// up / tq (optional): the diagonal blocks also leave tq[f] += sum_a W[f][a] up[a] over their 32 columns (the W part of the Cauchy point's
// quadratic form, as in schur_ksplit5) -- tq zeroed by the caller, LDS atomics: every feature meets every diagonal block once.
template <class FT, class FR>
VIO_DEV void schur_blocks(const Ctx &cx, const double *Wf, int ldw, int n6, int F, cldsd einv, cldsd ge, int share, int nshare, FT flush_el,
                          FR flush_rhs, cldsd up = nullptr, ldsd tq = nullptr) {
  const int T = (n6 + 15) / 16, NB = (T + 1) / 2, nblocks = NB * (NB + 1) / 2;
  const int tid_ = VIO_TID(cx), wave = tid_ >> 6, lane = tid_ & 63, nw = cx.nt >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int ksteps = (F + 3) / 4;
  constexpr int kChunk = 6;  // k-steps whose operands are fetched together: 24 global loads in flight per lane
  const int nchunks = (ksteps + kChunk - 1) / kChunk;
  int ksplit = 1;
  {
    const int waves = nshare * nw;
    int best = ((nblocks + waves - 1) / waves) * nchunks;  // cost in chunks: rounds x chunks per part
    for (int k = 2; k <= 4; k++) {
      const int cost = ((nblocks * k + waves - 1) / waves) * ((nchunks + k - 1) / k);
      if (cost < best) best = cost, ksplit = k;
    }
  }
  const int ks_part = (nchunks + ksplit - 1) / ksplit * kChunk;  // whole chunks per part
  for (int u = share * nw + wave; u < nblocks * ksplit; u += nshare * nw) {
    const int p = u / ksplit, part = u - p * ksplit;
    const int s_lo = part * ks_part, s_hi = s_lo + ks_part < ksteps ? s_lo + ks_part : ksteps;
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= p) I++;
    const int J = p - I * (I + 1) / 2;
    const bool diag = I == J;
    const int ta[2] = {2 * I, 2 * I + 1}, tb[2] = {2 * J, 2 * J + 1};
    bool va[2], vb[2];
    const double *pa[2], *pb[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      va[q] = ta[q] < T && 16 * ta[q] + li < n6, vb[q] = tb[q] < T && 16 * tb[q] + li < n6;
      pa[q] = Wf + (va[q] ? 16 * ta[q] + li : 0), pb[q] = Wf + (vb[q] ? 16 * tb[q] + li : 0);  // feature-major: 16 lanes = 128 B
    }
    v4d c00 = {0.0, 0.0, 0.0, 0.0}, c01 = c00, c10 = c00, c11 = c00;  // c_xy: row tile ta[x], column tile tb[y]
    double rp[2] = {0.0, 0.0};
    const bool do_tq = diag && tq != nullptr;
    double u2[2] = {0.0, 0.0};
    if (do_tq) {
#pragma unroll
      for (int q = 0; q < 2; q++) u2[q] = va[q] ? up[16 * ta[q] + li] : 0.0;
    }
    for (int s0 = s_lo; s0 < s_hi; s0 += kChunk) {
      double av[2][kChunk], bv[2][kChunk], ev[kChunk], gv[kChunk];
#pragma unroll
      for (int j = 0; j < kChunk; j++) {  // issue every load of the chunk before anything consumes one
        const int f = 4 * (s0 + j) + kq;
        const size_t fo = (size_t)((f < F && s0 + j < s_hi) ? f : 0) * ldw;
        av[0][j] = pa[0][fo], av[1][j] = pa[1][fo], bv[0][j] = pb[0][fo], bv[1][j] = pb[1][fo];
        ev[j] = einv[(f < F && s0 + j < s_hi) ? f : 0];
        gv[j] = diag ? ge[(f < F && s0 + j < s_hi) ? f : 0] : 0.0;
      }
      VIO_SCHED_FENCE();
#pragma unroll
      for (int j = 0; j < kChunk; j++) {
        const int f = 4 * (s0 + j) + kq;
        const bool vf = f < F && s0 + j < s_hi;
        double pq = 0.0;
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const double a_raw = (va[q] && vf) ? av[q][j] : 0.0;
          if (diag) rp[q] = fma(a_raw, gv[j], rp[q]);  // (uniform)
          if (do_tq) pq = fma(a_raw, u2[q], pq);
          av[q][j] = a_raw * ev[j], bv[q][j] = (vb[q] && vf) ? bv[q][j] : 0.0;
        }
        if (do_tq) {  // (uniform) this block's part of row f against up: the 16 lanes of the row (total in lane 15)
          pq += dpp_move_f64<0x111, 0xf>(pq);
          pq += dpp_move_f64<0x112, 0xf>(pq);
          pq += dpp_move_f64<0x114, 0xf>(pq);
          pq += dpp_move_f64<0x118, 0xf>(pq);
          if (li == 15 && vf) VIO_ATOMIC_ADD(tq + f, pq);
        }
      }
#pragma unroll
      for (int j = 0; j < kChunk; j++) {
        c00 = mfma_f64(av[0][j], bv[0][j], c00), c10 = mfma_f64(av[1][j], bv[0][j], c10), c11 = mfma_f64(av[1][j], bv[1][j], c11);
        if (!diag) c01 = mfma_f64(av[0][j], bv[1][j], c01);  // (uniform; on the diagonal that tile lies above it)
      }
    }
    auto flush = [&](int tr, int tc, const v4d &acc) {
      const int bcol = 16 * tc + li;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int arow = 16 * tr + kq + 4 * r;
        if (tr < T && tc < T && arow < n6 && bcol < n6 && bcol <= arow) flush_el(arow, bcol, acc[r], ksplit > 1);
      }
    };
    flush(ta[0], tb[0], c00), flush(ta[1], tb[0], c10), flush(ta[1], tb[1], c11);
    if (!diag) flush(ta[0], tb[1], c01);
    if (diag) {
#pragma unroll
      for (int q = 0; q < 2; q++)
        if (va[q] && rp[q] != 0.0) flush_rhs(16 * ta[q] + li, rp[q]);
    }
  }
}

// The landmark Schur term of the general path (more than 5 tile rows) and the right-hand side row: App -= (W E^-1) W^T,
// App[n6][:] -= W (g_f / E_f). share / nshare: the units and right-hand side items of THIS workgroup of a cooperative window (App
// is global scratch in this variant).
// tq (optional, zeroed by the caller): += W^T u_p of this workgroup's diagonal blocks, u_p by pose index in w.xt.
template <class WK>
VIO_DEV void schur_general(const Ctx &cx, const WinView &v, WK &w, int share, int nshare, ldsd tq = nullptr) {
  const int n6 = v.n6, F = v.F;
  schur_blocks(
      cx, v.WTf, v.n6cap, n6, F, w.einv, w.tf, share, nshare,
      [&](int arow, int bcol, double val, bool parts) {
        if (parts) schur_add(w.App + tri_at(arow, bcol), -val);
        else w.App[tri_at(arow, bcol)] -= val;  // (one writer per tile: plain read-modify-write)
      },
      [&](int a, double val) { schur_add(w.App + tri_at(n6, a), -val); },  // rhs_p -= sum_f W_f (g_f / E_f)
      w.xt, tq);
  stamp(cx, ST_SCHUR);
}

// In place: (H + mu C) on the diagonals, then the landmark Schur term  App -= (W E^-1) W^T  and the right-hand side row
// App[n6][:] = g_p - W (g_f / E_f). Returns false if some E_f <= 0.
template <class WK>
VIO_DEV bool build_reduced_system(const Ctx &cx, const WinView &v, WK &w, double mu, ldsd tq = nullptr) {
  const int np = v.np, F = v.F;
  stamp(cx, ST_TR_VEC);
  // Ceres solves (S H S + mu D^2) y = S g with the Jacobi scaling S and D^2 = clamp(diag(S H S)). The same system in
  // unscaled unknowns z = S y is (H + mu C) z = g with C = D^2 / S^2 (diagonal): no scaling pass over the matrix or
  // over the landmark coupling W is needed, and y = z / s at the end. (Cholesky is invariant under this diagonal
  // congruence up to rounding; a non-positive pivot appears in both forms or in neither.)
  VIO_PARFOR(f, F) {
    const double isf = rcp_f(w.sf[f]);
    const double e = w.hff[f] + mu * (feat_d2(w, f) * isf * isf);  // E_f = H_ff + mu (D_f / s_f)^2
    const double ei = rcp_f(e);
    w.einv[f] = ei;
    w.tf[f] = w.gf[f] * ei;  // g_f / E_f
    if (!(e > 0.0)) w.flag[0] = 1;
  }
  const bool tq_general = tq != nullptr && ((v.n6 + 15) >> 4) > 5;  // (the general path accumulates tq: zeroed here, ahead of the barrier)
  if (tq_general) VIO_PARFOR(f, F) tq[f] = 0.0;
  VIO_PARFOR(i, np) {
    const int f = i / kBS, c = i - f * kBS;
    const double cc = w.dp[i] * rcp_f(w.sp[i]);
    if (c < 6) {
      w.App[tri_at(6 * f + c, 6 * f + c)] += mu * cc * cc;
      w.App[tri_at(v.n6, 6 * f + c)] = w.gp[i];  // the carried right-hand side (speed-bias rows: taken from gp by the panel)
    } else {
      w.Dss[f * kSS + (c - 6) * (kSB + 1)] += mu * cc * cc;
    }
  }
  VIO_SYNC();
  stamp(cx, ST_SCALE);
  const int n6 = v.n6;
  {
    // Landmark Schur complement as a GEMM on the matrix cores: C(n6 x n6, lower tiles) = (W E^-1) W^T, K = F. Rows and
    // columns are pose indices: the product tiles ARE the tiles of App.
    const int T = (n6 + 15) / 16, npairs = T * (T + 1) / 2;
    const int tid_ = VIO_TID(cx), wave = tid_ >> 6, lane = tid_ & 63, nw = cx.nt >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int ksteps = (F + 3) / 4;
    constexpr int kT = 5;  // row tiles the K-split form holds in registers (15 accumulators)
    if (T <= kT) {
      schur_ksplit5(cx, v.WTf, v.n6cap, n6, F, w.einv, w.tf,
                    [&](int arow, int bcol, double val) { VIO_ATOMIC_ADD(w.App + tri_at(arow, bcol), -val); },
                    [&](int a, double val) { VIO_ATOMIC_ADD(w.App + tri_at(n6, a), -val); }, w.xt, tq);
    } else {
#ifndef VIO_HOST_BUILD
      if (cx.coop > 1) {
        // cooperative window: the helpers need 1 / E_f and g_f / E_f (the matrix and W are global); every workgroup takes its
        // share of the tile pairs, the owner goes on when all of them are done
        const CoopLayout L = CoopLayout::make(v.Pcap, v.Fcap, v.nblk_cap);
        VIO_PARFOR(f, F) v.coop[L.o_einv + f] = w.einv[f], v.coop[L.o_tf + f] = w.tf[f];
        if (tq) VIO_PARFOR(q, 16 * T) v.coop[L.o_pose + q] = w.xt[q];  // u_p for the helpers' part of W^T u_p (the pose slot is free here)
        coop_post(cx, v, COOP_SCHUR);
        schur_general(cx, v, w, 0, cx.coop, tq);
        coop_wait_helpers(cx, v);
        if (tq) {  // the helpers' parts of W^T u_p (left in the per-landmark slot of their partial records)
          VIO_SYNC();
          for (int m = 1; m < cx.coop; m++) {
            const double *pr = v.coop + L.o_part + (size_t)(m - 1) * L.part;
            VIO_PARFOR(f, F) tq[f] += pr[L.p_f + f];
          }
        }
      } else
#endif
        schur_general(cx, v, w, 0, 1, tq);
    }
  }
  VIO_SYNC();
  stamp(cx, ST_RHS);
  (void)np;
  return w.flag[0] == 0;
}

// =====================================================================================================
// Factorization of the reduced system
// =====================================================================================================
// Elimination order: speed-bias blocks s_W, s_{W-1}, ..., s_0, then the poses.
//   band   (wave 0, a chain of 9-pivot steps)   L_k = chol(D_k - E_{k+1} E_{k+1}^T),  E_k = C_k L_k^-T
//   panel  (the other waves, one step behind)   V_k^T = L_k^-1 (Asp_k - E_{k+1} V_{k+1}^T)   9 x (n6 + 1): the fill of
//          the eliminated block into the pose columns and, in column n6, the forward-substituted right-hand side;
//          App -= V_k V_k^T at once (which also carries App's right-hand side row along), then V_k is forgotten
//   poses  tiled right-looking Cholesky of App (16 x 16 tiles, look-ahead on wave 0); the row n6 of L is y_p
// Every product is a chain of v_mfma_f64_16x16x4 whose intermediate tiles stay in registers: the accumulator layout of
// a tile is the B-operand layout of the next product (V_{k+1}^T -> E V^T, T -> L^-1 T) and V_k^T in accumulator layout
// is V_k in operand layout for the rank-9 update of App.

// Band step k on one wave.
template <class WK>
VIO_DEV void band_step(const Ctx &cx, const WinView &v, WK &w, int k, int fail_flag, int lane_) {
  const int lane = VIO_OPAQUE(lane_), li = lane & 15, kq = lane >> 4;
  ldsd D = w.Dss + k * kSS;
  ldsd ldk = w.ldinv + kSB * k;
  const bool good = potrf9_inv_wave(D, w.Css + (k + 1 <= v.W ? k + 1 : k) * kSS, k < v.W, ldk, lane);
  if (!good && lane == 0) w.flag[fail_flag] = 1;
  stamp(cx, ST_D3);
  if (k >= 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the factor just stored is read back in another lane mapping
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    ldsd C = w.Css + k * kSS;
    double a[3], b[4];
    load_op9_raw(C, li, kq, a), load_linv9_raw(D, ldk, li, kq, b);
    VIO_SCHED_FENCE();
    mask_op9(li, kq, a), mask_linv9(li, kq, b);
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 3; s++) acc = mfma_f64(a[s], b[s], acc);
    __builtin_amdgcn_wave_barrier();  // (in place: every operand load of the wave precedes the stores)
    if (li < kSB) {
#pragma unroll
      for (int r = 0; r < 3; r++)
        if (kq + 4 * r < kSB) C[(kq + 4 * r) * kSB + li] = acc[r];
    }
  }
}

// Asp_k^T tile t in accumulator layout: element r = A(s_k[kq + 4 r], pose index 16 t + li); column n6 = the gradient of
// s_k (the right-hand side rides along as one more pose column). Raw fetch (clamped addresses, nothing consumes the
// values) and the masking at the point of use are separate so that a caller can put every read of a step in one batch.
struct PanelRaw {
  double x[3];   // AspI (LDS)
  double p[3];   // Apri (global): only fetched for the speed-bias block the prior keeps
};
template <class WK>
VIO_DEV PanelRaw panel_fetch(const WinView &v, WK &w, int k, bool is_pr, int t, int li, int kq) {
  const int j = 16 * t + li, alo = 6 * (k > 0 ? k - 1 : 0);
  const bool inr = j >= alo && j < alo + kAW && j < v.n6;
  const int ao = inr ? j - alo : 0;
  PanelRaw p;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int c = kq + 4 * r < kSB ? kq + 4 * r : 0;
    if (w.asp_ring) p.x[r] = w.aspring[(k & 1) * kAS + ao + c * kAW];
    else p.x[r] = w.AspI[k * kAS + ao + c * kAW];
    p.p[r] = is_pr ? v.Apri[c * v.jp + (j < v.n6 ? j : 0)] : 0.0;
  }
  return p;
}
// gr[r]: gradient of s_k[kq + 4 r] (raw LDS values, fetched by the caller in its batch)
VIO_DEV v4d panel_tile(int n6, int k, bool is_pr, int t, int li, int kq, const PanelRaw &p, const double gr[3]) {
  const int j = 16 * t + li, alo = 6 * (k > 0 ? k - 1 : 0);
  const bool inr = j >= alo && j < alo + kAW && j < n6, inp = is_pr && j < n6, isr = j == n6;
  v4d T = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const bool ok = kq + 4 * r < kSB;
    const double x = (inr ? p.x[r] : 0.0) + (inp ? p.p[r] : 0.0);
    T[r] = ok ? (isr ? gr[r] : x) : 0.0;
  }
  return T;
}
template <class WK>
VIO_DEV v4d panel_load_tile(const WinView &v, WK &w, int k, int t, int li, int kq) {
  const bool is_pr = w.sbr[2 * k + 1] != 0;
  const PanelRaw p = panel_fetch(v, w, k, is_pr, t, li, kq);
  double gr[3];
#pragma unroll
  for (int r = 0; r < 3; r++) gr[r] = w.gp[kBS * k + 6 + (kq + 4 * r < kSB ? kq + 4 * r : 0)];
  return panel_tile(v.n6, k, is_pr, t, li, kq, p, gr);
}

// Panel step k with the fill tiles in registers (every panel wave computes all of V_k: 6 matrix instructions per tile
// cost less than a hand-over through LDS and its barrier), then this wave's share of the rank-9 update of App.
// V: three doubles per tile (element 3 of the accumulator layout is sb component 12 + kq: always zero). Tiles below tlo
// are structurally zero at this step (the fill of s_k only reaches the poses the frames k.. couple to) and are skipped.
// NPW panel waves: tile q = I (I + 1) / 2 + J of the update belongs to wave q % NPW, which keeps it in accumulator
// q / NPW. Every read of the step -- Asp_k (global), operands, accumulators, gradient (LDS) -- is issued in ONE batch
// ahead of the first matrix instruction, every store follows the last. (The Asp fetch is not carried over from the
// previous step: 30 more registers per lane put scratch reloads into this loop, which cost more than the L2 round trip
// that hides behind wave 0's longer band step anyway.)
struct VTile {
  double x[3];
};
template <int NT, int NPW, class WK>
VIO_DEV void panel_step_regs(const Ctx &cx, const WinView &v, WK &w, int k, int tlo, VTile (&V)[NT], int lane_, int pw) {
  const int lane = VIO_OPAQUE(lane_), li = lane & 15, kq = lane >> 4;
  const int nT = v.nT;
  double e[3] = {0.0, 0.0, 0.0}, linv[4], gr[3];
  constexpr int kTiles = NT * (NT + 1) / 2, kAcc = (kTiles + NPW - 1) / NPW;
  v4d acc[kAcc];
  PanelRaw X[NT];
  const bool is_pr = __builtin_amdgcn_readfirstlane(w.sbr[2 * k + 1]) != 0;
#pragma unroll
  for (int t = 0; t < NT; t++)
    if (t >= tlo && t < nT) X[t] = panel_fetch(v, w, k, is_pr, t, li, kq);
  if (k < v.W) load_op9_raw(w.Css + (k + 1) * kSS, li, kq, e);
  load_linv9_raw(w.Dss + k * kSS, w.ldinv + kSB * k, li, kq, linv);
#pragma unroll
  for (int r = 0; r < 3; r++) gr[r] = w.gp[kBS * k + 6 + (kq + 4 * r < kSB ? kq + 4 * r : 0)];
#pragma unroll
  for (int I = 0; I < NT; I++)
#pragma unroll
    for (int J = 0; J <= I; J++) {
      const int q = I * (I + 1) / 2 + J;
      if (I < nT && J >= tlo && q % NPW == pw)
        acc[q / NPW] = tile_load_acc_raw(w.App + tri_off(I) + 16 * J, tri_ld(I), v.nrows - 16 * I, li, kq);
    }
  VIO_SCHED_FENCE();
  if (k < v.W) mask_op9(li, kq, e);
  mask_linv9(li, kq, linv);
  if (cx.prof) {
    double probe = linv[0] + e[0] + gr[0] + X[NT - 1].x[0] + X[NT - 1].p[0] + acc[0][0];  // (timing run only: the batch has landed)
    asm volatile("" ::"v"(probe));
    stamp(cx, ST_D4);
  }
#pragma unroll
  for (int t = 0; t < NT; t++) {
    if (t < tlo || t >= nT) continue;
    v4d Tt = panel_tile(v.n6, k, is_pr, t, li, kq, X[t], gr);
    if (k < v.W) {
#pragma unroll
      for (int s = 0; s < 3; s++) Tt = mfma_f64(-e[s], V[t].x[s], Tt);
    }
    v4d Vn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 3; s++) Vn = mfma_f64(linv[s], Tt[s], Vn);
    V[t].x[0] = Vn[0], V[t].x[1] = Vn[1], V[t].x[2] = Vn[2];
  }
  stamp(cx, ST_D0);
#pragma unroll
  for (int I = 0; I < NT; I++)
#pragma unroll
    for (int J = 0; J <= I; J++) {
      const int q = I * (I + 1) / 2 + J;
      if (I < nT && J >= tlo && q % NPW == pw) {
        v4d c = tile_mask_acc(acc[q / NPW], v.nrows - 16 * I, kq);
#pragma unroll
        for (int s = 0; s < 3; s++) c = mfma_f64(-V[I].x[s], V[J].x[s], c);
        acc[q / NPW] = c;
      }
    }
  if (cx.prof) {
    double probe = acc[0][0] + acc[kAcc - 1][0];
    asm volatile("" ::"v"(probe));
    stamp(cx, ST_D1);
  }
#pragma unroll
  for (int I = 0; I < NT; I++)
#pragma unroll
    for (int J = 0; J <= I; J++) {
      const int q = I * (I + 1) / 2 + J;
      if (I < nT && J >= tlo && q % NPW == pw)
        tile_store_acc(w.App + tri_off(I) + 16 * J, tri_ld(I), v.nrows - 16 * I, li, kq, acc[q / NPW]);
    }
}

// Which tiles of the update a panel wave owns and which fill tiles V_t it therefore needs. Default: tile q = I (I + 1) / 2 + J
// goes to wave q % NPW, which needs (nearly) every V_t. For 5 tile rows on 3 panel waves the tiles are grouped by the
// indices they touch instead -- {0, 1, 4}, {2, 3, 4} and the 2 x 2 block {2, 3} x {0, 1} -- so that a wave forms 3 or 4 of
// the 5 fill tiles (6 matrix instructions each) for its 6 / 5 / 4 update tiles: 36 / 33 / 36 matrix instructions per step
// instead of 45 on a pipe the wave shares with a wave of the CU's other window.
template <int NT, int NPW, int PW>
struct PanelMap {
  static constexpr bool owns(int I, int J) { return (I * (I + 1) / 2 + J) % NPW == PW; }
  static constexpr bool needs(int) { return true; }
};
template <int PW>
struct PanelMap<5, 3, PW> {
  static constexpr bool in_a(int t) { return t == 0 || t == 1 || t == 4; }
  static constexpr bool in_b(int t) { return t == 2 || t == 3 || t == 4; }
  static constexpr bool owns(int I, int J) {
    const bool a = in_a(I) && in_a(J), b = in_b(I) && in_b(J) && !a;  // (tile (4, 4) belongs to the first set)
    return PW == 0 ? a : PW == 1 ? b : !a && !b;
  }
  static constexpr bool needs(int t) { return PW == 0 ? in_a(t) : PW == 1 ? in_b(t) : t < 4; }
};
template <int NT, int NPW, int PW>
constexpr int panel_slot(int I, int J) {  // position of tile (I, J) in the wave's accumulator list
  int n = 0;
  for (int i = 0; i < NT; i++)
    for (int j = 0; j <= i; j++) {
      if (i == I && j == J) return n;
      if (PanelMap<NT, NPW, PW>::owns(i, j)) n++;
    }
  return n;
}
template <int NT, int NPW, int PW>
constexpr int panel_count() { return panel_slot<NT, NPW, PW>(NT, 0); }

// The same step for a pose matrix of exactly NT tile rows, specialised for panel wave PW of NPW: the tiles of the wave
// are a compile-time list, so every accumulator access is one ds_read / ds_write at an immediate offset from one of NT
// per-lane row bases (the generic form spends ~600 instructions per step on predicates and address arithmetic).
// (ISPR: block k is the speed-bias block the prior keeps -- one step of a factorization --: its coupling has the prior's dense
// row in global memory on top of the IMU chain's 18 columns. The common instantiation carries neither the loads nor the selects.)
// (TLO: first tile column the fill reaches, a compile-time value since round 6 -- it only takes the values 0 .. NT - 2 along the
// chain --: every "tile >= tlo" test of the step folds, nothing branches around a tile)
template <int NT, int NPW, int PW, bool ISPR, int TLO, class WK>
VIO_DEV void panel_step_static(const Ctx &cx, const WinView &v, WK &w, int k, VTile (&V)[NT], int lane_) {
  constexpr int tlo = TLO;
  const int lane = VIO_OPAQUE(lane_), li = lane & 15, kq = lane >> 4;
  double e[3] = {0.0, 0.0, 0.0}, linv[4], gr[3];
  typedef PanelMap<NT, NPW, PW> Map;
  constexpr int kAcc = panel_count<NT, NPW, PW>();
  v4d acc[kAcc];
  double X[NT][3], Xp[NT][3];
  const int rows_last = v.nrows - 16 * (NT - 1);  // rows of the last tile row (the others are full)
  const int n6 = v.n6, alo = 6 * (k > 0 ? k - 1 : 0);
  // The IMU chain couples s_k to the 18 pose columns from alo on: two of the five tiles. A tile outside them (and outside the
  // prior's row, ISPR) is zero except for the right-hand side column (last tile): nothing to fetch, nothing to select.
  auto touches = [&](int t) { return ISPR || (16 * t + 15 >= alo && 16 * t < alo + kAW); };  // (uniform)
  // Round 6: a lane reads its entry of Asp_k^T tile t THROUGH ITS ADDRESS -- the coupling where the tile has one, the gradient of
  // s_k in the right-hand side column, a zero (the pad behind the coupling's LDS copy) everywhere else -- instead of reading a
  // clamped address and selecting on the 64-bit value afterwards: one 32-bit select per element where there were 2-3 pairs.
  constexpr bool kLdsAsp = std::is_same<decltype(w.AspI), ldsd>::value;
  constexpr bool kByAddress = !ISPR;
  cldsd asp_src = nullptr, zero_src = nullptr;
  if constexpr (kLdsAsp) {
    if (w.asp_lds) asp_src = w.AspI + k * kAS, zero_src = w.AspI + v.Pcap * kAS;
  }
  if (w.asp_ring) asp_src = w.aspring + (k & 1) * kAS, zero_src = w.aspring + 2 * kAS;
#pragma unroll
  for (int t = 0; t < NT; t++) {
    if (!(Map::needs(t) && t >= tlo)) continue;
    if (kByAddress && ((kLdsAsp && w.asp_lds) || w.asp_ring)) {
      if (!touches(t) && t != NT - 1) continue;
      const int j = 16 * t + li;
      const bool inr = j >= alo && j < alo + kAW && j < n6, isr = t == NT - 1 && j == n6;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const int c = kq + 4 * r;
        const bool rowok = r < 2 || kq == 0;  // (rows kq + 4 r >= 9 only exist for r = 2, kq >= 1: zero)
        cldsd p = zero_src;
        p = (inr && rowok) ? asp_src + (j - alo) + c * kAW : p;
        p = (isr && rowok) ? w.gp + (kBS * k + 6 + c) : p;
        X[t][r] = *p;
      }
      continue;
    }
    if (!touches(t)) continue;
    const int j = 16 * t + li;
    const bool inr = j >= alo && j < alo + kAW && j < n6;
    const int ao = inr ? j - alo : 0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int c = kq + 4 * r < kSB ? kq + 4 * r : 0;
      if (w.asp_ring) X[t][r] = w.aspring[(k & 1) * kAS + ao + c * kAW];
      else X[t][r] = w.AspI[k * kAS + ao + c * kAW];
      if (ISPR) Xp[t][r] = v.Apri[c * v.jp + (j < n6 ? j : 0)];
    }
  }
  // (E_{k+1} and L_k^-1 through clamped addresses + value selects: fetching them "by address" like the tiles above traded 220
  // selects for as many scalar / move instructions and measured 1 % slower)
  if (k < v.W) load_op9_raw(w.Css + (k + 1) * kSS, li, kq, e);
  load_linv9_raw(w.Dss + k * kSS, w.ldinv + kSB * k, li, kq, linv);
#pragma unroll
  for (int r = 0; r < 3; r++) gr[r] = w.gp[kBS * k + 6 + (kq + 4 * r < kSB ? kq + 4 * r : 0)];
#pragma unroll
  for (int I = 0; I < NT; I++) {
    auto rowbase = w.App + tri_off(I) + kq * tri_ld(I) + li;  // element r of tile (I, J): rowbase[4 r ld + 16 J]
#pragma unroll
    for (int J = 0; J <= I; J++) {
      if (Map::owns(I, J) && J >= tlo) {
        v4d a;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (I < NT - 1) a[r] = rowbase[4 * r * tri_ld(I) + 16 * J];
          else if (4 * r >= rows_last) a[r] = 0.0;  // (no lane holds a row of this element: W = 10 without a loop pose has 3 rows here -- r = 0 only)
          else a[r] = rowbase[(kq + 4 * r < rows_last ? 4 * r * tri_ld(I) : -kq * tri_ld(I)) + 16 * J];
        }
        acc[panel_slot<NT, NPW, PW>(I, J)] = a;
      }
    }
  }
  VIO_SCHED_FENCE();
  if (k < v.W) mask_op9(li, kq, e);
  mask_linv9(li, kq, linv);
  if (cx.prof) {
    double probe = linv[0] + e[0] + gr[0] + acc[0][0];  // (timing run only: the batch has landed)
    asm volatile("" ::"v"(probe));
    stamp(cx, ST_D4);
  }
#pragma unroll
  for (int t = 0; t < NT; t++) {
    if (!Map::needs(t) || t < tlo) continue;
    // Asp_k^T tile t in accumulator layout: element r = A(s_k[kq + 4 r], pose index 16 t + li); column n6 (last tile) = the
    // gradient of s_k. Rows kq + 4 r >= 9 only exist for r = 2, kq >= 1.
    v4d Tt = {0.0, 0.0, 0.0, 0.0};
    if (kByAddress && ((kLdsAsp && w.asp_lds) || w.asp_ring)) {
      if (touches(t) || t == NT - 1) Tt[0] = X[t][0], Tt[1] = X[t][1], Tt[2] = X[t][2];
    } else {
      if (touches(t)) {
        const int j = 16 * t + li;
        const bool inr = j >= alo && j < alo + kAW && j < n6;
#pragma unroll
        for (int r = 0; r < 3; r++) {
          double x = inr ? X[t][r] : 0.0;
          if (ISPR) x += j < n6 ? Xp[t][r] : 0.0;
          Tt[r] = x;
        }
      }
      if (t == NT - 1) {  // (the right-hand side column: n6 = 66 or 72 sits in the last tile)
        const bool isr = 16 * t + li == n6;
#pragma unroll
        for (int r = 0; r < 3; r++) Tt[r] = isr ? gr[r] : Tt[r];
      }
      Tt[2] = kq == 0 ? Tt[2] : 0.0;
    }
    if (k < v.W) {
#pragma unroll
      for (int s = 0; s < 3; s++) Tt = mfma_f64(-e[s], V[t].x[s], Tt);
    }
    v4d Vn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 3; s++) Vn = mfma_f64(linv[s], Tt[s], Vn);
    V[t].x[0] = Vn[0], V[t].x[1] = Vn[1], V[t].x[2] = Vn[2];
  }
  stamp(cx, ST_D0);
#pragma unroll
  for (int I = 0; I < NT; I++)
#pragma unroll
    for (int J = 0; J <= I; J++) {
      if (Map::owns(I, J) && J >= tlo) {
        // (rows of the last tile row past the matrix hold whatever the clamped load fetched: a row of the product only depends on
        // the same row of the accumulator, and those rows are stored to the padding element -- no masking)
        v4d c = acc[panel_slot<NT, NPW, PW>(I, J)];
#pragma unroll
        for (int s = 0; s < 3; s++) c = mfma_f64(-V[I].x[s], V[J].x[s], c);
        acc[panel_slot<NT, NPW, PW>(I, J)] = c;
      }
    }
  if (cx.prof) {
    double probe = acc[0][0] + acc[kAcc - 1][0];
    asm volatile("" ::"v"(probe));
    stamp(cx, ST_D1);
  }
#pragma unroll
  for (int I = 0; I < NT; I++) {
    auto rowbase = w.App + tri_off(I) + kq * tri_ld(I) + li;
#pragma unroll
    for (int J = 0; J <= I; J++) {
      if (Map::owns(I, J) && J >= tlo) {
        const v4d c = acc[panel_slot<NT, NPW, PW>(I, J)];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (I < NT - 1) rowbase[4 * r * tri_ld(I) + 16 * J] = c[r];
          else if (4 * r >= rows_last) continue;
          else {
            // (rows past the matrix: the store goes to the padding element behind row 0 of this tile row -- every row is
            // 16 (I + 1) + 1 long and its last element is never read -- instead of an exec-masked region per store)
            const int off = kq + 4 * r < rows_last ? 4 * r * tri_ld(I) + 16 * J : 16 * (I + 1) - li - kq * tri_ld(I);
            rowbase[off] = c[r];
          }
        }
      }
    }
  }
}
template <int NT, int NPW, class WK>
VIO_DEV void panel_step_dispatch(const Ctx &cx, const WinView &v, WK &w, int k, int tlo, VTile (&V)[NT], int lane, int pw) {
  if (v.nT != NT) {
    panel_step_regs<NT, NPW>(cx, v, w, k, tlo, V, lane, pw);
    return;
  }
  if constexpr (NPW == 3) {
    const bool is_pr = __builtin_amdgcn_readfirstlane(w.sbr[2 * k + 1]) != 0;
    auto run = [&](auto PWc) {
      constexpr int PW = decltype(PWc)::value;
      if (is_pr || tlo <= 0) {  // (the prior's block reaches every column)
        if (is_pr) panel_step_static<NT, 3, PW, true, 0>(cx, v, w, k, V, lane);
        else panel_step_static<NT, 3, PW, false, 0>(cx, v, w, k, V, lane);
      } else if (tlo == 1) panel_step_static<NT, 3, PW, false, 1>(cx, v, w, k, V, lane);
      else if (tlo == 2) panel_step_static<NT, 3, PW, false, 2>(cx, v, w, k, V, lane);
      else if (tlo == 3) panel_step_static<NT, 3, PW, false, 3>(cx, v, w, k, V, lane);
      else panel_step_static<NT, 3, PW, false, NT - 1>(cx, v, w, k, V, lane);
    };
    if (pw == 0) run(std::integral_constant<int, 0>{});
    else if (pw == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
  } else {
    panel_step_regs<NT, NPW>(cx, v, w, k, tlo, V, lane, pw);
  }
}

// Band + panel for pose matrices of up to NT tile rows, NW waves. false: a pivot of the band was <= 0.
template <int NT, int NW, class WK>
VIO_DEV bool factor_band_regs(const Ctx &cx, const WinView &v, WK &w) {
  const int tid_ = VIO_TID(cx), wave = (__builtin_amdgcn_readfirstlane(tid_ >> 6) + cx.wrot) & (NW - 1), lane = tid_ & 63;
  const int W = v.W;
  VTile V[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) V[t].x[0] = V[t].x[1] = V[t].x[2] = 0.0;
  int flo = v.n6;  // first pose column the fill of the steps so far reaches
  if (!w.asp_ring) {
    // Decoupled (round 6). The chain wave needs nothing from the panel waves while it walks the band (they only write the pose
    // matrix), so it runs ahead at its own pace and posts a flag per finished block; a panel wave waits for the flag of ITS block
    // only. With a workgroup barrier per slot (rounds 3-5) a slot took the longer of the two -- the early panel steps touch two
    // fill tiles, the late ones five, a band block always costs the same -- and the band phase the sum of those maxima.
    if (wave == 0) {
      VIO_PRIO(3);
      for (int kb = W; kb >= 0; kb--) {
        band_step(cx, v, w, kb, 2, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();  // (every lane's stores of the block precede the flag)
        if (lane == 0) VIO_FLAG_STORE(w.ready + kb, 1);
        stamp(cx, ST_C_AHEAD);
      }
      VIO_PRIO(0);
    } else {
      for (int kp = W; kp >= 0; kp--) {
        while (__builtin_amdgcn_readfirstlane(VIO_FLAG_LOAD(w.ready + kp)) == 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        flo = flo < w.sbr[2 * kp] ? flo : w.sbr[2 * kp];  // (0 for the block the prior keeps)
        panel_step_dispatch<NT, NW - 1>(cx, v, w, kp, flo >> 4, V, lane, wave - 1);
        stamp(cx, ST_D2);
      }
    }
    VIO_SYNC_LDS();
    stamp(cx, ST_C_WAIT);
    return w.flag[2] == 0;
  }
  for (int slot = 0; slot <= W + 1; slot++) {
    const int kb = W - slot, kp = kb + 1, ff = 2 + (slot & 1);
    if (wave == 0) {
      // AspI in global scratch: the coupling rows of block kb -- what the panel waves need in the NEXT slot -- are fetched
      // here, travel behind the band step and land in the LDS ring before the slot's barrier (an L2 round trip at the
      // head of every panel step otherwise: ~3.5 k of a slot's ~10 k cycles)
      double pf[3] = {0.0, 0.0, 0.0};
      const bool ring = w.asp_ring && kb >= 0;
      if (ring) {
#pragma unroll
        for (int r = 0; r < 3; r++) pf[r] = w.AspI[kb * kAS + (lane + 64 * r < kAS ? lane + 64 * r : 0)];
      }
      if (kb >= 0) band_step(cx, v, w, kb, ff, lane);
      if (ring) {
#pragma unroll
        for (int r = 0; r < 3; r++)
          if (lane + 64 * r < kAS) w.aspring[(kb & 1) * kAS + lane + 64 * r] = pf[r];
      }
      stamp(cx, ST_C_AHEAD);
    } else if (kp <= W) {
      flo = flo < w.sbr[2 * kp] ? flo : w.sbr[2 * kp];  // (0 for the block the prior keeps)
      panel_step_dispatch<NT, NW - 1>(cx, v, w, kp, flo >> 4, V, lane, wave - 1);
      stamp(cx, ST_D2);  // (clock on a panel wave: the rank-9 updates)
    }
    VIO_SYNC_LDS();  // (band blocks, pose matrix and flags are all LDS)
    stamp(cx, ST_C_WAIT);
    // (the flag of this slot is not written again before every wave has passed the next barrier)
    if (w.flag[ff]) return false;
  }
  return true;
}

// The same for pose matrices of any size that live in GLOBAL scratch (W > 12; round 6, the cooperative factorization).
// Rounds 3-5 walked the band slot by slot: band block on wave 0, barrier, the fill tiles V_k through LDS, barrier, then a
// read-modify-write of EVERY tile of the pose matrix in global memory -- W + 1 times per factorization, three barriers and two
// dependent L2 round trips per slot (c_wait + c_trsm: 25 % of a W = 30 solve, on the owner alone in a cooperative window).
// Now:
//   (A) the chain wave runs the band blocks at its own pace and posts a flag per block (like factor_band_regs). Column tile t of
//       the fill, V_k^T[t] = L_k^-1 (Asp_k^T[t] - E_{k+1} V_{k+1}^T[t]), depends on tile t of the previous step ONLY: every
//       panel wave owns whole tile columns, keeps the running tile in registers (accumulator layout = B-operand layout) down
//       the chain and stores each V_k^T[t] once, in operand layout, into the window's global panel buffer w.VG. No barrier.
//   (B) App -= sum_k V_k V_k^T as ONE product: every lower tile (I, J) has one writer, which loads it once, runs over the
//       blocks k that reach tile column J (the fill of s_k starts at pose column w.sbr[2 k]: about half of all (tile, k) pairs
//       are structurally zero) and stores it once. The tiles are dealt out heaviest first over the waves -- of ALL workgroups of
//       a cooperative window (command COOP_SYRK: the helpers read w.VG and the matrix from the XCD's L2).
// Reference: the reduced system's factorization, CSI/schur_complement_solver.cc:161-224 (result, not route).
#ifdef VIO_SIMT
inline int &simt_syrk_shares() {  // (test hook of the SIMT emulator build, tests/emul/simt_backend.cpp)
  static int n = 1;
  return n;
}
#endif
template <class WK>
VIO_DEV int band_tile_reach(const WinView &v, const WK &w, int t) {  // largest k whose fill reaches tile column t
  int flo = v.n6;
  for (int k = v.W; k > 0; k--) {
    flo = flo < w.sbr[2 * k] ? flo : w.sbr[2 * k];
    if (t >= (flo >> 4)) return k;
  }
  return 0;
}
template <class WK>
VIO_DEV void band_syrk_share(const Ctx &cx, const WinView &v, WK &w, int share, int nshare) {
  const int tid_ = VIO_TID(cx), wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), nw = cx.nt >> 6, lane = tid_ & 63;
  const int li = lane & 15, kq = lane >> 4, nT = v.nT;
  const int ntiles = nT * (nT + 1) / 2, nwk = nshare * nw, me = share * nw + wave;
  const double *VG = w.VG;
  // (lane t walks the reach table for tile column t once: a walk per tile was 30 dependent LDS reads ahead of its first fetch)
  const int my_reach = band_tile_reach(v, w, lane < nT ? lane : nT - 1);
  for (int r = 0;; r++) {
    // (heaviest first -- tile column nT - 1 is reached by every block, column 0 by the last few --, dealt out in snake order)
    const int q = r * nwk + ((r & 1) ? nwk - 1 - me : me);
    if (r * nwk >= ntiles) break;
    if (q >= ntiles) continue;
    int c = 0;
    while ((c + 1) * (c + 2) / 2 <= q) c++;
    const int J = nT - 1 - c, I = J + (q - c * (c + 1) / 2);
    auto C = w.App + tri_off(I) + 16 * J;
    const int ld = tri_ld(I), rows = v.nrows - 16 * I;
    v4d acc = tile_load_acc_raw(C, ld, rows, li, kq);  // (masked below, behind the first operand fetch)
    const int kmax = __builtin_amdgcn_readlane(my_reach, J);
    // operand batches of KB blocks, the next batch's loads in flight behind this one's matrix instructions
    constexpr int KB = 3;
    double a[2][KB][3], b[2][KB][3];
    auto fetch = [&](int k0, double (&aa)[KB][3], double (&bb)[KB][3]) {
#pragma unroll
      for (int u = 0; u < KB; u++) {
        const int k = k0 - u >= 0 ? k0 - u : 0;
        const double *pa = VG + ((size_t)(k * nT + I) * 3) * 64 + lane, *pb = VG + ((size_t)(k * nT + J) * 3) * 64 + lane;
#pragma unroll
        for (int s3 = 0; s3 < 3; s3++) aa[u][s3] = pa[64 * s3], bb[u][s3] = pb[64 * s3];
      }
    };
    auto multiply = [&](int k0, const double (&aa)[KB][3], const double (&bb)[KB][3]) {
#pragma unroll
      for (int u = 0; u < KB; u++) {
        const bool on = k0 - u >= 0;
#pragma unroll
        for (int s3 = 0; s3 < 3; s3++) acc = mfma_f64(on ? -aa[u][s3] : 0.0, bb[u][s3], acc);
      }
    };
    fetch(kmax, a[0], b[0]);
    VIO_SCHED_FENCE();
    acc = tile_mask_acc(acc, rows, kq);
    for (int k0 = kmax; k0 >= 0; k0 -= 2 * KB) {
      if (k0 - KB >= 0) fetch(k0 - KB, a[1], b[1]);
      VIO_SCHED_FENCE();
      multiply(k0, a[0], b[0]);
      if (k0 - KB < 0) break;
      if (k0 - 2 * KB >= 0) fetch(k0 - 2 * KB, a[0], b[0]);
      VIO_SCHED_FENCE();
      multiply(k0 - KB, a[1], b[1]);
    }
    tile_store_acc(C, ld, rows, li, kq, acc);
  }
}
template <class WK>
VIO_DEV bool factor_band_lds(const Ctx &cx, const WinView &v, WK &w) {
  const int tid_ = VIO_TID(cx), wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), nw = cx.nt >> 6, lane = tid_ & 63;
  const int li = lane & 15, kq = lane >> 4;
  const int W = v.W, nT = v.nT, n6 = v.n6;
  double *VG = w.VG;
  if (wave == 0) {
    VIO_PRIO(3);
    for (int kb = W; kb >= 0; kb--) {
      band_step(cx, v, w, kb, 2, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_wave_barrier();  // (every lane's stores of the block precede the flag)
      if (lane == 0) VIO_FLAG_STORE(w.ready + kb, 1);
      stamp(cx, ST_C_AHEAD);
    }
    VIO_PRIO(0);
  } else {
    // (two tile columns of a wave walk down the chain together: one flag wait and one batch of LDS operands per block)
    constexpr int NTW = 2;
    for (int t0 = wave - 1; t0 < nT; t0 += NTW * (nw - 1)) {
      double V[NTW][3];
      bool started[NTW];
#pragma unroll
      for (int j = 0; j < NTW; j++) V[j][0] = V[j][1] = V[j][2] = 0.0, started[j] = false;
      // (a wave joins the chain at the first block whose fill reaches one of its tile columns: until then it sleeps in long
      // intervals instead of polling every block's flag beside the chain wave)
      const int t1 = t0 + (nw - 1) < nT ? t0 + (nw - 1) : t0;
      const int kstart = __builtin_amdgcn_readfirstlane(band_tile_reach(v, w, t1));
      int flo = n6;
      for (int k = W; k > kstart; k--) flo = flo < w.sbr[2 * k] ? flo : w.sbr[2 * k];
      while (__builtin_amdgcn_readfirstlane(VIO_FLAG_LOAD(w.ready + kstart)) == 0) __builtin_amdgcn_s_sleep(8);
      for (int k = kstart; k >= 0; k--) {
        while (__builtin_amdgcn_readfirstlane(VIO_FLAG_LOAD(w.ready + k)) == 0) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        flo = flo < w.sbr[2 * k] ? flo : w.sbr[2 * k];  // (0 for the block the prior keeps)
        const bool is_pr = w.sbr[2 * k + 1] != 0;
        const int alo = 6 * (k > 0 ? k - 1 : 0);
        bool act[NTW];
        v4d Tt[NTW];
#pragma unroll
        for (int j = 0; j < NTW; j++) {
          const int t = t0 + j * (nw - 1);
          act[j] = t < nT && t >= (flo >> 4);  // (else: the fill has not reached this tile column yet)
          Tt[j] = v4d{0.0, 0.0, 0.0, 0.0};
          if (!act[j]) continue;
          // Asp_k^T tile t (+ the gradient of s_k in column n6): the coupling rows are global in this variant -- fetched only where
          // the 18 IMU columns of block k (or the prior's dense row) fall into the tile
          if (is_pr || (16 * t + 15 >= alo && 16 * t < alo + kAW)) {
            Tt[j] = panel_load_tile(v, w, k, t, li, kq);
          } else if (16 * t + li == n6) {
#pragma unroll
            for (int r = 0; r < 3; r++)
              if (kq + 4 * r < kSB) Tt[j][r] = w.gp[kBS * k + 6 + kq + 4 * r];
          }
        }
        if (!act[0] && !act[1]) continue;
        double e[3] = {0.0, 0.0, 0.0}, linv[4];
        if (k < W) load_op9(w.Css + (k + 1) * kSS, li, kq, e);
        load_linv9(w.Dss + k * kSS, w.ldinv + kSB * k, li, kq, linv);
#pragma unroll
        for (int j = 0; j < NTW; j++) {
          if (!act[j]) continue;
          if (started[j]) {
#pragma unroll
            for (int s3 = 0; s3 < 3; s3++) Tt[j] = mfma_f64(-e[s3], V[j][s3], Tt[j]);
          }
          v4d Vn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s3 = 0; s3 < 3; s3++) Vn = mfma_f64(linv[s3], Tt[j][s3], Vn);
          double *dst = VG + ((size_t)(k * nT + t0 + j * (nw - 1)) * 3) * 64 + lane;
#pragma unroll
          for (int s3 = 0; s3 < 3; s3++) V[j][s3] = Vn[s3], dst[64 * s3] = Vn[s3];
          started[j] = true;
        }
      }
      stamp(cx, ST_D2);
    }
  }
  VIO_SYNC();  // (the panels are global stores: visible to the whole workgroup behind this barrier)
  stamp(cx, ST_C_TRSM);
  if (w.flag[2]) return false;
#ifndef VIO_HOST_BUILD
  if (cx.coop > 1) {
    coop_post(cx, v, COOP_SYRK);
    band_syrk_share(cx, v, w, 0, cx.coop);
    coop_wait_helpers(cx, v);
  } else
#endif
#ifdef VIO_SIMT
  // (tests: the shares of a cooperative window one after the other on the emulated workgroup -- every tile must be written exactly once)
  for (int sh = 1; sh < simt_syrk_shares(); sh++) band_syrk_share(cx, v, w, sh, simt_syrk_shares());
  band_syrk_share(cx, v, w, 0, simt_syrk_shares());
#else
    band_syrk_share(cx, v, w, 0, 1);
#endif
  VIO_SYNC();
  stamp(cx, ST_C_WAIT);
  return true;
}

// Tiled right-looking Cholesky of App with the right-hand side row carried along. false: a pivot was <= 0.
template <class WK>
VIO_DEV bool factor_poses(const Ctx &cx, const WinView &v, WK &w) {
  const int tid_ = VIO_TID(cx), nw = cx.nt >> 6, wave = (__builtin_amdgcn_readfirstlane(tid_ >> 6) + cx.wrot) & (nw - 1), lane0 = tid_ & 63;
  const int nT = v.nT, nrows = v.nrows, n6 = v.n6;
  ldsd ldp = w.ldinv + kSB * v.P;
  auto tile = [&](int I, int J) { return w.App + tri_off(I) + 16 * J; };
  auto rows_of = [&](int I) { return nrows - 16 * I < 16 ? nrows - 16 * I : 16; };
  auto piv_of = [&](int I) { return n6 - 16 * I < 16 ? (n6 - 16 * I > 0 ? n6 - 16 * I : 0) : 16; };
  if (wave == 0) {
    VIO_PRIO(3);
    const bool good = potrf16_wave(tile(0, 0), tile(0, 0), tri_ld(0), rows_of(0), piv_of(0), false, ldp, lane0);
    if (!good && lane0 == 0) w.flag[1] = 1;
    VIO_PRIO(0);
  }
  VIO_SYNC();
  stamp(cx, ST_C_POTRF);
  for (int K = 0; K < nT; K++) {
    if (w.flag[1]) return false;
    const int ntb = nT - K - 1;
    const int lane = VIO_OPAQUE(lane0), li = lane & 15, kq = lane >> 4;
    for (int bi = wave; bi < ntb; bi += nw)
      tile_trsm(tile(K + 1 + bi, K), tri_ld(K + 1 + bi), rows_of(K + 1 + bi), tile(K, K), tri_ld(K), ldp + 16 * K, li, kq);
    VIO_SYNC();
    stamp(cx, ST_C_TRSM);
    // trailing update with look-ahead: wave 0 updates the next diagonal tile in registers and factors it at once while
    // the other waves update the remaining tiles
    if (wave == 0) {
      if (ntb > 0) {
        VIO_PRIO(3);
        const bool good = potrf16_wave(tile(K + 1, K + 1), tile(K + 1, K), tri_ld(K + 1), rows_of(K + 1), piv_of(K + 1), true,
                                       ldp + 16 * (K + 1), lane);
        if (!good && lane == 0) w.flag[1] = 1;
        VIO_PRIO(0);
      }
      stamp(cx, ST_C_AHEAD);
    } else {
      const int npairs = ntb * (ntb + 1) / 2;  // pair 0 = the look-ahead tile
      for (int pr = wave; pr < npairs; pr += nw - 1) {
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= pr) a++;
        const int I = K + 1 + a, J = K + 1 + pr - a * (a + 1) / 2;
        tile_update(tile(I, J), tri_ld(I), rows_of(I), tile(I, K), tile(J, K), tri_ld(J), rows_of(J), li, kq);
      }
    }
    VIO_SYNC();
    stamp(cx, ST_C_WAIT);
  }
  return w.flag[1] == 0;
}

// z <- (H + mu C)^-1 g from the factorization: w.t1 receives z (frame-major). Poses: backward substitution through the
// tiles of App (one barrier per tile column, wave 0 runs the chain); speed-bias: A_ss z_s = g_s - A_sp z_p with the
// UNFACTORED coupling (the fill was never stored) through the band factor, forward (newest to oldest) and backward.
// While wave 0 walks the band, the other waves accumulate w_f^T z_p of the landmark back-substitution into w.gnf.
template <class WK>
VIO_DEV void backsolve(const Ctx &cx, const WinView &v, WK &w) {
  const int tid_ = VIO_TID(cx), nw = cx.nt >> 6, wave = (__builtin_amdgcn_readfirstlane(tid_ >> 6) + cx.wrot) & (nw - 1), lane = tid_ & 63;
  const int nT = v.nT, n6 = v.n6, P = v.P, W = v.W, F = v.F;
  cldsd ldp = w.ldinv + kSB * P;
  ldsd x = w.xt;
  auto tile = [&](int I, int J) { return w.App + tri_off(I) + 16 * J; };
  auto piv_of = [&](int I) { return n6 - 16 * I < 16 ? n6 - 16 * I : 16; };
  VIO_PARFOR(a, 16 * nT) x[a] = a < n6 ? w.App[tri_at(n6, a)] : 0.0;
  VIO_PARFOR(f, F) w.gnf[f] = 0.0;  // accumulates w_f^T z_p below
  VIO_PARFOR(q, P * kSB) w.t1[kBS * (q / kSB) + 6 + q % kSB] = w.gp[kBS * (q / kSB) + 6 + q % kSB];  // t_s = g_s - A_sp z_p below
  VIO_SYNC();
  stamp(cx, ST_B_INIT);
  int qc = lane >> 2, qp = lane & 3;  // lane = 4 c + p: the four lanes of a quad split a 16-term dot product
  // x_K <- L_KK^-T x_K: (L^-T x)[c] = x[c] / L_cc + sum_{r > c} Linv[r][c] x[r], Linv[r][c] at D[c][r]
  auto solve_diag = [&](int K) {
    auto D = tile(K, K);
    const int ld = tri_ld(K), np_ = piv_of(K);
    const bool ok = qc < np_;
    const int c = ok ? qc : 0;
    double sacc = qp == 0 ? ldp[16 * K + c] * x[16 * K + c] : 0.0;
#pragma unroll
    for (int t4 = 0; t4 < 4; t4++) {
      const int r = c + 1 + qp + 4 * t4;
      const bool in = ok && r < np_;
      const double lv = D[c * ld + (in ? r : c)], xv = x[16 * K + (in ? r : c)];
      sacc = fma(in ? lv : 0.0, xv, sacc);
    }
    sacc = quad_sum_f64(sacc);
    __builtin_amdgcn_wave_barrier();
    if (ok && qp == 0) x[16 * K + qc] = sacc;
  };
  // x_J[c] -= (L_KJ^T x_K)[c], rows of tile (K, J) that are pivots
  auto apply = [&](int K, int J) {
    auto Lt = tile(K, J);
    const int ld = tri_ld(K), np_ = piv_of(K);
    double sacc = 0.0;
#pragma unroll
    for (int t4 = 0; t4 < 4; t4++) {
      const int m = qp + 4 * t4;
      const bool in = m < np_;
      const double lv = Lt[(in ? m : 0) * ld + qc], xv = x[16 * K + (in ? m : 0)];
      sacc = fma(in ? lv : 0.0, xv, sacc);
    }
    sacc = quad_sum_f64(sacc);
    if (qp == 0) x[16 * J + qc] -= sacc;
  };
  if (wave == 0) solve_diag(nT - 1);
  VIO_SYNC();
  for (int K = nT - 1; K >= 1; K--) {
    qc = VIO_OPAQUE(qc), qp = VIO_OPAQUE(qp);
    if (wave == 0) {
      apply(K, K - 1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      solve_diag(K - 1);
    } else {
      for (int J = wave - 1; J < K - 1; J += nw - 1) apply(K, J);
    }
    VIO_SYNC();
  }
  stamp(cx, ST_B_POSE);
  // z_p -> t1 (pose components), t_s = g_s - A_sp z_p -> t1 (speed-bias components)
  VIO_PARFOR(a, n6) w.t1[kBS * (a / 6) + a % 6] = x[a];
  VIO_PARFOR(q, P * kSB + kSB * nT) {
    if (q < P * kSB) {  // the IMU chain's coupling (LDS)
      const int k = q / kSB, c = q - k * kSB;
      const int alo = 6 * (k > 0 ? k - 1 : 0), aw = n6 - alo < kAW ? n6 - alo : kAW;
      auto A = w.AspI + (k * kSB + c) * kAW;
      double sacc = 0.0, ar[kAW];
#pragma unroll
      for (int jj = 0; jj < kAW; jj++) ar[jj] = A[jj < aw ? jj : 0];
      VIO_SCHED_FENCE();
#pragma unroll
      for (int jj = 0; jj < kAW; jj++) sacc = fma(jj < aw ? ar[jj] : 0.0, x[alo + (jj < aw ? jj : 0)], sacc);
      VIO_ATOMIC_ADD(w.t1 + kBS * k + 6 + c, -sacc);
    } else {  // the prior's block (global): (component, strip of 16 columns) items, one fetch batch each
      const int qq = q - P * kSB, c = qq / nT, j0 = 16 * (qq - c * nT);
      int kpr = -1;
      for (int k = 0; k < P; k++)
        if (w.sbr[2 * k + 1]) kpr = k;
      if (kpr < 0) continue;
      double xs[16], sacc = 0;
      const double *Ap = v.Apri + (size_t)c * v.jp + j0;
#pragma unroll
      for (int j = 0; j < 16; j++) xs[j] = Ap[j];
      VIO_SCHED_FENCE();
#pragma unroll
      for (int j = 0; j < 16; j++) sacc = fma(xs[j], x[j0 + j], sacc);  // (x is zero past n6)
      VIO_ATOMIC_ADD(w.t1 + kBS * kpr + 6 + c, -sacc);
    }
  }
  VIO_SYNC();
  stamp(cx, ST_BACKSOLVE);
  if (wave == 0) {
    // band: u_k = L_k^-1 (t_k - E_{k+1} u_{k+1}), k = W..0; then z_k = L_k^-T (u_k - E_k^T z_{k-1}), k = 0..W: two chains of 9 x 9
    // mat-vecs on the matrix cores. The running vector is a B operand replicated over the 16 columns: the accumulator layout of a
    // product (lane (li, kq), element r = component kq + 4 r, the same in every column li) IS the B-operand layout of the next
    // one, so a step is 3 + 3 dependent v_mfma and nothing else on the chain; the blocks of the next step are fetched while this
    // one multiplies. (Rounds 3-5 ran these chains on v_readlane broadcasts: 18 dependent multiply-adds and 36 broadcasts per
    // step, ~1.0 k cycles of the ~22 k the two chains took per solve.)
    const int li = lane & 15, kq = lane >> 4;
    struct Blk {
      double e[3], l[4], t[3];
    };
    // forward operands of step k: E_{k+1} (rows s_k, columns s_{k+1}) and L_k^-1 as A operands X[li][4 s + kq]; t_k in accumulator layout
    auto fetch_fwd = [&](int k, Blk &b) {
      load_op9_raw(w.Css + (k + 1 <= W ? k + 1 : k) * kSS, li, kq, b.e);
      load_linv9_raw(w.Dss + k * kSS, w.ldinv + kSB * k, li, kq, b.l);
#pragma unroll
      for (int r = 0; r < 3; r++) b.t[r] = w.t1[kBS * k + 6 + (kq + 4 * r < kSB ? kq + 4 * r : 0)];
    };
    // backward operands of step k: E_k^T (X[li][j] = E_k[j][li]) and L_k^-T (X[li][j] = Linv[j][li]: above the diagonal of the
    // stored block for j > li, 1 / L_ii on it, zero below)
    auto fetch_bwd = [&](int k, Blk &b) {
      const int n = li < kSB ? li : 0;
      cldsd E = w.Css + k * kSS + n, D = w.Dss + k * kSS + n * kSB;
      b.l[3] = w.ldinv[kSB * k + n];
#pragma unroll
      for (int s4 = 0; s4 < 3; s4++) {
        const int j = 4 * s4 + kq, jc = j < kSB ? j : 0;
        b.e[s4] = E[jc * kSB];
        b.l[s4] = D[jc > n ? jc : n];
      }
#pragma unroll
      for (int r = 0; r < 3; r++) b.t[r] = w.t1[kBS * k + 6 + (kq + 4 * r < kSB ? kq + 4 * r : 0)];
    };
    auto mask_bwd = [&](Blk &b) {
      const bool iok = li < kSB;
#pragma unroll
      for (int s4 = 0; s4 < 3; s4++) {
        const int j = 4 * s4 + kq;
        const bool jok = iok && j < kSB;
        b.e[s4] = jok ? b.e[s4] : 0.0;
        b.l[s4] = (jok && j > li) ? b.l[s4] : ((jok && j == li) ? b.l[3] : 0.0);
      }
    };
    auto acc_of = [&](const Blk &b) {
      v4d T = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 3; r++) T[r] = kq + 4 * r < kSB ? b.t[r] : 0.0;
      return T;
    };
    auto put = [&](int k, v4d U) {
      if (li == 0) {
#pragma unroll
        for (int r = 0; r < 3; r++)
          if (kq + 4 * r < kSB) w.t1[kBS * k + 6 + kq + 4 * r] = U[r];
      }
    };
    VIO_PRIO(3);
    Blk cur, nxt;
    v4d U = {0.0, 0.0, 0.0, 0.0};
    fetch_fwd(W, cur);
    for (int k = W; k >= 0; k--) {
      if (k >= 1) fetch_fwd(k - 1, nxt);
      VIO_SCHED_FENCE();
      mask_op9(li, kq, cur.e), mask_linv9(li, kq, cur.l);
      v4d T = acc_of(cur);
      if (k < W) {
#pragma unroll
        for (int s4 = 0; s4 < 3; s4++) T = mfma_f64(-cur.e[s4], U[s4], T);
      }
      v4d Un = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 3; s4++) Un = mfma_f64(cur.l[s4], T[s4], Un);
      U = Un;
      put(k, U);
      cur = nxt;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    U = v4d{0.0, 0.0, 0.0, 0.0};
    fetch_bwd(0, cur);
    for (int k = 0; k <= W; k++) {
      if (k < W) fetch_bwd(k + 1, nxt);
      VIO_SCHED_FENCE();
      mask_bwd(cur);
      v4d T = acc_of(cur);
      if (k >= 1) {
#pragma unroll
        for (int s4 = 0; s4 < 3; s4++) T = mfma_f64(-cur.e[s4], U[s4], T);
      }
      v4d Zn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 3; s4++) Zn = mfma_f64(cur.l[s4], T[s4], Zn);
      U = Zn;
      put(k, U);
      cur = nxt;
    }
    VIO_PRIO(0);
    stamp(cx, ST_X0);  // (the two band chains; ST_B_BAND behind the barrier is then the wait for the landmark part)
  } else {
    // landmark back-substitution, first half: w_f^T z_p by the waves that do not walk the band. Pose matrices of up to five tile
    // rows: a row of the feature-major W per 16 lanes (lane li takes columns li, li + 16, ...: 128 contiguous bytes per load and
    // row), four rows per wave and step, the 16-lane sum on the DPP network -- the (feature, part) items below read a strip of 22
    // consecutive doubles per LANE, 64 cache lines per load instruction, and took longer than the band chains beside them.
    if (nT <= 5) {
      constexpr int kT = 5, kIt = 7;  // rows in flight per lane: kIt steps of four rows, five columns each
      const int li = lane & 15, kq = lane >> 4, pw = wave - 1, npw = nw - 1;
      double u5[kT];
#pragma unroll
      for (int t = 0; t < kT; t++) u5[t] = 16 * t + li < n6 ? x[16 * t + li] : 0.0;
      const int steps = (F + 3) >> 2, per_w = (steps + npw - 1) / npw;
      const int s_begin = pw * per_w, s_end = s_begin + per_w < steps ? s_begin + per_w : steps;
      for (int s0 = s_begin; s0 < s_end; s0 += kIt) {
        double wv[kIt][kT];
#pragma unroll
        for (int j = 0; j < kIt; j++) {
          const int f = 4 * (s0 + j) + kq, fc = (s0 + j < s_end && f < F) ? f : 0;
#pragma unroll
          for (int t = 0; t < kT; t++) wv[j][t] = v.WTf[(size_t)fc * v.n6cap + (16 * t + li < n6 ? 16 * t + li : 0)];
        }
        VIO_SCHED_FENCE();
#pragma unroll
        for (int j = 0; j < kIt; j++) {
          const int f = 4 * (s0 + j) + kq;
          double pq = 0.0;
#pragma unroll
          for (int t = 0; t < kT; t++) pq = fma(16 * t + li < n6 ? wv[j][t] : 0.0, u5[t], pq);
          pq += dpp_move_f64<0x111, 0xf>(pq);
          pq += dpp_move_f64<0x112, 0xf>(pq);
          pq += dpp_move_f64<0x114, 0xf>(pq);
          pq += dpp_move_f64<0x118, 0xf>(pq);
          if (li == 15 && s0 + j < s_end && f < F) w.gnf[f] = pq;  // (one writer per landmark)
        }
      }
    } else {
    int nparts, per;
    wt_parts((int)cx.nt - 64, F, n6, nparts, per);
    for (int q = 64 * (wave - 1) + lane; q < F * nparts; q += (int)cx.nt - 64) {
      const int part = q / F, f = q - part * F;
      const int a0 = part * per, a1 = a0 + per < n6 ? a0 + per : n6;
      double xs[kWStrip], sacc = 0;
      for (int b0 = a0; b0 < a1; b0 += kWStrip) {
        const int nb = a1 - b0 < kWStrip ? a1 - b0 : kWStrip;
        wt_strip_load(v.WTf + (size_t)f * v.n6cap + b0, 1, nb, xs);  // the lane's own contiguous strip of the feature-major W
#pragma unroll
        for (int j = 0; j < kWStrip; j++) sacc += (j < nb ? xs[j] : 0.0) * x[b0 + (j < nb ? j : 0)];
      }
      VIO_ATOMIC_ADD(w.gnf + f, sacc);
    }
    }
  }
  VIO_SYNC();
  stamp(cx, ST_B_BAND);
}

// PoseLocalParameterization::Plus on all blocks: c = x [+] (delta_p, delta_f)
template <class WK>
VIO_DEV void apply_plus(const Ctx &cx, const WinView &v, WK &w, cldsd dpv, cldsd dfv) {
  const int npose = v.P + v.has_loop;
  VIO_PARFOR(i, npose) {
    auto p0 = w.xpose + 7 * i;
    auto d = dpv + off_pose(v, i);
    auto p = w.cpose + 7 * i;
    for (int k = 0; k < 3; k++) p[k] = p0[k] + d[k];
    Quat q = qnormalized(qmul(qfrom_pose(p0), Quat{d[3] / 2.0, d[4] / 2.0, d[5] / 2.0, 1.0}));
    p[3] = q.x, p[4] = q.y, p[5] = q.z, p[6] = q.w;
  }
  VIO_PARFOR(q, v.P * 9) w.csb[q] = w.xsb[q] + dpv[off_sb(q / 9) + q % 9];
  VIO_PARFOR(f, v.F) w.cfeat[f] = w.xfeat[f] + dfv[f];
  VIO_SYNC();
}

// norms over the reduced program's (global-size) parameters: |a - b|_2 / |a - b|_inf / |a|_2
VIO_DEV void state_norms(const Ctx &cx, const WinView &v, cldsd apose, cldsd asb, cldsd afeat, cldsd bpose, cldsd bsb,
                         cldsd bfeat, double *l2, double *linf) {
  double s = 0, m = 0;
  const int npose = v.P + v.has_loop;
  VIO_PARFOR(q, npose * 7) {
    double d = apose[q] - (bpose ? bpose[q] : 0.0);
    s += d * d, m = fmax(m, fabs(d));
  }
  VIO_PARFOR(q, v.P * 9) {
    double d = asb[q] - (bsb ? bsb[q] : 0.0);
    s += d * d, m = fmax(m, fabs(d));
  }
  VIO_PARFOR(q, v.F) {
    double d = afeat[q] - (bfeat ? bfeat[q] : 0.0);
    s += d * d, m = fmax(m, fabs(d));
  }
  *l2 = sqrt(block_sum(cx, s));
  if (linf) *linf = block_max(cx, m);
}

// The O(n) vector work of one trust-region iteration used to be a dozen barrier-separated passes of one element per
// work-item each (Plus, three norms, the step, its three inner products, the parking of the iterate ...): every pass costs
// its LDS round trips and a block reduction whatever it computes. The two fused passes below replace eight of them.
//
// x <- x [+] (delta_p, delta_f) IN PLACE (PoseLocalParameterization::Plus on every block), the old iterate parked in
// `stash` (global: pose 7 (P + 1) | sb at o_sb | feat at o_f). Partial sums: d2 += |x_new - x_old|^2, n2 += |x_new|^2 over
// the global-size parameters (what state_norms sums).
template <class WK>
VIO_DEV void plus_in_place(const Ctx &cx, const WinView &v, WK &w, cldsd dpv, cldsd dfv, double *stash, int o_sb, int o_f,
                           double &d2, double &n2) {
  const int npose = v.P + v.has_loop;
  VIO_PARFOR(i, npose) {
    auto p0 = w.xpose + 7 * i;
    auto d = dpv + off_pose(v, i);
    double o[7], c[7];
#pragma unroll
    for (int k = 0; k < 7; k++) o[k] = p0[k];
    for (int k = 0; k < 3; k++) c[k] = o[k] + d[k];
    Quat q = qnormalized(qmul(Quat{o[3], o[4], o[5], o[6]}, Quat{d[3] / 2.0, d[4] / 2.0, d[5] / 2.0, 1.0}));
    c[3] = q.x, c[4] = q.y, c[5] = q.z, c[6] = q.w;
#pragma unroll
    for (int k = 0; k < 7; k++) {
      const double e = c[k] - o[k];
      d2 = fma(e, e, d2), n2 = fma(c[k], c[k], n2);
      stash[7 * i + k] = o[k], p0[k] = c[k];
    }
  }
  VIO_PARFOR(q, v.P * 9) {
    const double o = w.xsb[q], c = o + dpv[off_sb(q / 9) + q % 9], e = c - o;
    d2 = fma(e, e, d2), n2 = fma(c, c, n2);
    stash[o_sb + q] = o, w.xsb[q] = c;
  }
  VIO_PARFOR(f, v.F) {
    const double o = w.xfeat[f], c = o + dfv[f], e = c - o;
    d2 = fma(e, e, d2), n2 = fma(c, c, n2);
    stash[o_f + f] = o, w.xfeat[f] = c;
  }
}

// |x - Plus(x, -g)|_inf (trust_region_minimizer.cc:270-284) in one pass and one reduction: the candidate is not stored.
template <class WK>
VIO_DEV double gradient_max_norm(const Ctx &cx, const WinView &v, WK &w) {
  const int npose = v.P + v.has_loop;
  double m = 0.0;
  VIO_PARFOR(i, npose) {
    auto p0 = w.xpose + 7 * i;
    auto g = w.gp + off_pose(v, i);
    for (int k = 0; k < 3; k++) m = fmax(m, fabs(p0[k] - (p0[k] - g[k])));
    const Quat q0{p0[3], p0[4], p0[5], p0[6]};
    const Quat q = qnormalized(qmul(q0, Quat{-g[3] / 2.0, -g[4] / 2.0, -g[5] / 2.0, 1.0}));
    m = fmax(fmax(m, fabs(q0.x - q.x)), fmax(fabs(q0.y - q.y), fmax(fabs(q0.z - q.z), fabs(q0.w - q.w))));
  }
  VIO_PARFOR(q, v.P * 9) {
    const double o = w.xsb[q];
    m = fmax(m, fabs(o - (o - w.gp[off_sb(q / 9) + q % 9])));
  }
  VIO_PARFOR(f, v.F) {
    const double o = w.xfeat[f];
    m = fmax(m, fabs(o - (o - w.gf[f])));
  }
  return block_max(cx, m);
}

// =====================================================================================================
// TrustRegionMinimizer + DoglegStrategy (CSI/trust_region_minimizer.cc, CSI/dogleg_strategy.cc)
// =====================================================================================================
// REGS: the pose matrix has at most kPanelTiles tile rows (the launcher's LDS variant): fill tiles in registers
// The per-window view handed to a phase. A phase reads a few of WinView's ~50 pointers; carried through the whole kernel
// they are all live everywhere and most of them sit in spilled scalar registers (v_readlane to get one back). `fresh()`
// derives the view again from the kernel's arguments at the phase's entry: the fields the phase does not use fall away,
// the ones it uses cost a few scalar instructions. SameView: the stored view as is (host builds).
struct SameView {
  const WinView &v;
  VIO_DEV const WinView &operator()() const { return v; }
};
template <class WK>
struct SameWork {
  const WK &w;
  VIO_DEV WK operator()() const { return w; }
};

template <bool REGS, int NW, class WK, class VP, class WP>
VIO_DEV void minimize(const Ctx &cx, const WinView &v, WK &w_whole, const VP &fresh, const WP &fresh_work) {
  WK &w = w_whole;
  const int np = v.np, F = v.F;
  double *sd = v.stats_d;
  int *si = v.stats_i;
  auto record = [&](int i, double cost, double radius, double step_norm, double rel, double gmax, bool valid,
                    bool ok) {
    if (cx.tid == 0 && i < kMaxTrace) {
      sd[4 + i] = cost, sd[4 + kMaxTrace + i] = radius, sd[4 + 2 * kMaxTrace + i] = step_norm;
      sd[4 + 3 * kMaxTrace + i] = rel, sd[4 + 4 * kMaxTrace + i] = gmax;
      si[4 + i] = (valid ? 1 : 0) | (ok ? 2 : 0);
    }
  };
  auto grad_max_norm = [&]() { return gradient_max_norm(cx, fresh(), w); };

  double x_cost = evaluate(cx, fresh(), w, w.xpose, w.xsb, w.xfeat, true, false);
  double x_norm = -1.0;  // "Invalid value", trust_region_minimizer.cc:168
  VIO_PARFOR(f, F) w.sf[f] = rcp_f(1.0 + sqrt_f(w.hff[f]));  // Jacobi scaling, :239-254 (poses: at the end of evaluate)
  VIO_SYNC();
  double gmax = grad_max_norm();
  double radius = 1e4, mu = 1e-8;
  const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
  bool reuse = false, last_ok = true, have_factor = false;
  double dogleg_step_norm = 0, alpha = 0;
  int it = 0, n_ok = 1, n_bad = 0, invalid_run = 0, termination = 0, recorded = 1;
  double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost, ev_acc_ref = 0, ev_acc_cand = 0;
  double min_rec = x_cost;
  record(0, x_cost, radius, 0, 0, gmax, true, true);
  if (cx.tid == 0) sd[0] = x_cost;
  double gd_sq = 0, mu_used = mu, qf_cauchy = 0, qf_part = 0;
  bool qf_pending = false;  // the W part of the Cauchy point's quadratic form is still to be added (first dogleg step after a solve)
  // a linearization at x is due at the head of the next iteration: after a step that was accepted on a cost-only evaluation
  // (relin_reuse: the raw IMU Jacobians and the prior's dx of that evaluation are still valid; rec_*: the iteration record
  // waits for the gradient norm) or after an invalid step (the matrix buffer holds a factorization)
  bool relin = false, relin_reuse = false, rec_pending = false;
  int rec_it = 0;
  double rec_step_norm = 0, rec_rho = 0;

  while (true) {
    // (the LDS layout, like the view: derived again per iteration, so that its ~40 addresses are not carried -- and spilled --
    // across the whole loop)
    WK w_iter = fresh_work();
    WK &w = w_iter;
    if (relin) {
      evaluate(cx, fresh(), w, w.xpose, w.xsb, w.xfeat, true, true, relin_reuse);
      if (rec_pending) {
        gmax = grad_max_norm();
        record(rec_it, x_cost, radius, rec_step_norm, rec_rho, gmax, true, true);
        stamp(cx, ST_V_GMAX);
      }
      relin = relin_reuse = rec_pending = false;
    }
    if (it >= v.max_iter) break;
    if (last_ok && gmax <= 1e-10) { termination = 1; break; }
    if (radius <= 1e-32) { termination = 1; break; }
    it++;
    bool solver_ok = true;
    if (!reuse) {
      reuse = true;
      // The loop-carried scalars that the linear solve does not touch leave the registers for its duration (the same in every
      // lane, but values that come out of LDS reductions live in VGPRs: ~35 registers per lane that the panel steps of the
      // factorization are short of). (w.park: a slot of its own -- an evaluation inside the retry loop uses every other scratch vector.)
      ldsd park = w.park;
      if (cx.tid == 0) {
        park[0] = x_cost, park[1] = x_norm, park[2] = gmax, park[3] = radius, park[4] = dogleg_step_norm, park[5] = ev_min;
        park[6] = ev_cur, park[7] = ev_ref, park[8] = ev_cand, park[9] = ev_acc_ref, park[10] = ev_acc_cand, park[11] = min_rec;
        ldsi pi = reinterpret_cast<ldsi>(park + 12);
        pi[0] = it, pi[1] = n_ok, pi[2] = n_bad, pi[3] = invalid_run, pi[4] = termination, pi[5] = recorded, pi[6] = last_ok ? 1 : 0;
      }
      // |g_d|^2, the Cauchy direction a = D^-2 S g (-> t2 poses, stf landmarks) and the vectors quad_form_H starts from
      // (u = S a frame-major -> t1, by pose index -> xt, its landmark accumulator tf = 0) in ONE pass; the reduction's barrier
      // publishes them. Cauchy point: alpha = |g_d|^2 / |J_s (g_d / d)|^2 (dogleg_strategy.cc:172-192); |J_s a|^2 = u^T H u
      // is taken from the unfactored system, i.e. before the linear solve consumes it (it does not depend on mu).
      {
        double part = 0;
        VIO_PARFOR(i, np) {
          const int f = i / kBS, c = i - f * kBS;
          const double g = pose_gd(w, i), a = g * rcp_f(w.dp[i]), u = w.sp[i] * a;
          part += g * g;
          w.t2[i] = a, w.t1[i] = u;
          if (c < 6) w.xt[6 * f + c] = u;
        }
        VIO_PARFOR(f, F) {
          const double d2 = feat_d2(w, f), sg = w.sf[f] * w.gf[f], g = sg * rsqrt_f(d2);
          part += g * g;
          w.stf[f] = sg * rcp_f(d2), w.tf[f] = 0.0;
        }
        gd_sq = block_sum(cx, part);
      }
      stamp(cx, ST_V_GD);
      // (pose matrices of up to five tile rows: the W part of the form, 2 u_f^T W^T u_p, comes out of the Schur sweep of
      // build_reduced_system -- tq -> cfeat, dead between a linearization and the next Plus -- and joins the reductions of the
      // dogleg step below: one pass over W in global memory and one barrier less per iteration)
      const bool fuse_qw = true;  // (round 6: the general path's blocked product accumulates it as well, schur_blocks)
      const double qf_h0 = quad_form_H(cx, fresh(), w, w.t2, w.stf, /*prepared=*/true, /*w_part=*/!fuse_qw);
      stamp(cx, ST_QUADFORM);
      // Gauss-Newton step: (S H S + mu D^2) y = S g, features eliminated (dogleg_strategy.cc:515-612)
      solver_ok = false;
      bool first_try = true;
      while (mu < max_mu) {
        if (!first_try) {
          // retry with a larger mu: the in-place system was consumed, rebuild H from the factors (rare path)
          evaluate(cx, fresh(), w, w.xpose, w.xsb, w.xfeat, true, true);
          // (the Schur sweep below takes u_p = S D^-2 S g from xt for the W part of the Cauchy form: a first try that got as far
          // as the back-substitution has left z there)
          if (fuse_qw) {
            VIO_PARFOR(i, np) {
              const int f = i / kBS, c = i - f * kBS;
              const double u = w.sp[i] * (pose_gd(w, i) * rcp_f(w.dp[i]));
              w.t1[i] = u;
              if (c < 6) w.xt[6 * f + c] = u;
            }
          }
        }
        first_try = false;
        if (cx.tid == 0) w.flag[0] = 0, w.flag[1] = 0, w.flag[2] = 0, w.flag[3] = 0;
        VIO_PARFOR(k, v.P) w.ready[k] = 0;
        VIO_SYNC();
        bool ok = build_reduced_system(cx, fresh(), w, mu, fuse_qw ? w.cfeat : nullptr);
        if (ok) {
          if constexpr (REGS) ok = factor_band_regs<kPanelTiles, NW>(cx, fresh(), w);
          else ok = factor_band_lds(cx, fresh(), w);
        }
        if (ok) ok = factor_poses(cx, fresh(), w);
        stamp(cx, ST_CHOL);
        if (ok) {
          backsolve(cx, fresh(), w);  // z -> t1, w_f^T z_p -> gnf
          // back-substitute features: z_f = (g_f - w_f^T z_p) / E_f ; y = z / s ; GN = -d * y
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
        if (ok) { solver_ok = true; mu_used = mu; have_factor = true; break; }
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
        // mu |D a|^2 of the Cauchy direction a = D^-2 S g for the mu the solve ended with: D a = g_d, so it is mu |g_d|^2
        qf_part = qf_h0, qf_pending = fuse_qw;
        if (!fuse_qw) {
          qf_cauchy = qf_h0 + mu_used * gd_sq;
          alpha = gd_sq / qf_h0;
        }
        stamp(cx, ST_DOGLEG);
      }
    }
    (void)have_factor;
    bool step_valid = false;
    double model_cost_change = 0;
    if (solver_ok) {
      // ComputeTraditionalDoglegStep (dogleg_strategy.cc:199-255)
      double p1 = 0, p2 = 0, p3 = 0;
      VIO_PARFOR(i, np) p1 += w.gnp[i] * w.gnp[i], p2 += pose_gd(w, i) * w.gnp[i];
      VIO_PARFOR(f, F) {
        p1 += w.gnf[f] * w.gnf[f], p2 += feat_gd(w, f) * w.gnf[f];
        // (first step after a solve: u_f (W^T u_p)_f of the Cauchy direction, u_f = s_f^2 g_f / D_f^2, W^T u_p from the Schur sweep)
        if (qf_pending) p3 += 2.0 * (w.sf[f] * w.sf[f] * w.gf[f] * rcp_f(feat_d2(w, f))) * w.cfeat[f];
      }
      block_sum3(cx, p1, p2, p3);
      stamp(cx, ST_V_DOT);
      if (qf_pending) {
        const double qf_h = qf_part + p3;
        qf_cauchy = qf_h + mu_used * gd_sq;
        alpha = gd_sq / qf_h;
        qf_pending = false;
      }
      double gnn2 = p1, gdot = p2;
      double gradient_norm = sqrt(gd_sq), gauss_newton_norm = sqrt(gnn2);
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
      // The step  s = ca g_d + cb gn  (d-scaled coordinates) is a combination of two vectors whose inner products are reduced
      // already: |s|^2 = ca^2 |g_d|^2 + 2 ca cb g_d.gn + cb^2 |gn|^2, (step)^T S g = s . g_d = ca |g_d|^2 + cb g_d.gn,
      // mu |D step|^2 = mu |s|^2 -- no reduction pass over the step. What is stored is delta = S D^-1 s, the argument of Plus.
      VIO_PARFOR(i, np) {
        const double s_ = ca * pose_gd(w, i) + cb * w.gnp[i];
        w.t2[i] = s_ * rcp_f(w.dp[i]) * w.sp[i];
      }
      VIO_PARFOR(f, F) {
        const double id = rsqrt_f(feat_d2(w, f));
        const double s_ = ca * (w.sf[f] * w.gf[f] * id) + cb * w.gnf[f];
        w.tf[f] = s_ * id * w.sf[f];
      }
      const double n2 = ca * ca * gd_sq + 2.0 * ca * cb * gdot + cb * cb * gnn2, sg = ca * gd_sq + cb * gdot, reg = mu_used * n2;
      if (need_norm) dogleg_step_norm = sqrt(n2);
      VIO_SYNC();
      // model_cost_change = -(J step)^T (r + J step / 2) (trust_region_minimizer.cc:402-416). The dogleg step is a
      // combination  v = ca a - cb y  of the Cauchy direction a = D^-2 S g and the Gauss-Newton solution y of
      // M y = S g, M = S H S + mu D^2, so its quadratic form needs no third pass over the factor and the landmark
      // coupling:  v^T M v = ca^2 a^T M a - 2 ca cb a^T (S g) + cb^2 y^T (S g), with a^T M a the Cauchy form above,
      // a^T S g = |g_d|^2 and y^T S g = -g_d . gn, all reduced already. (Exact up to the residual of the linear solve.)
      stamp(cx, ST_V_STEP);
      double shs = ca * ca * qf_cauchy - 2.0 * ca * cb * gd_sq - cb * cb * gdot - reg;
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
      relin = true;  // the matrix buffer holds a factorization: H is rebuilt before the next build_reduced_system
      continue;
    }
    invalid_run = 0;
    stamp(cx, ST_DOGLEG);
    // After an ACCEPTED step the next candidate is linearized SPECULATIVELY -- cost and Jacobians in one evaluation, before
    // its step is accepted. Ceres evaluates the cost at the candidate and, once the step is accepted, residuals + Jacobians at
    // the same point (trust_region_minimizer.cc:428-640): two passes over the factors per accepted step, one here. The
    // evaluation works in place, so the iterate and the vectors of its linearization that a rejected step still needs wait
    // in global scratch (stores only; read back after a rejection). After a REJECTED step the next candidate takes Ceres'
    // route -- cost only, linearization once it is accepted --: rejections come in runs (a window at its noise floor), and a
    // rejected speculative evaluation wastes its Jacobian half, which at W = 30 is three quarters of it.
    const bool speculate = last_ok;
    const int npose7 = (v.P + v.has_loop) * 7, o_sb = 7 * (v.P + 1), o_f = o_sb + 9 * v.P, o_v = o_f + F, nv = v.nblk * kBS;
    // (either way the candidate takes the iterate's place for its evaluation -- one code path, fixed LDS addresses -- and the
    // iterate waits in the stash.) Plus, the step norm |x_cand - x|, the candidate's norm (x_norm once the step is accepted)
    // and the parking of the iterate are ONE pass and one reduction.
    double step_norm, cand_norm;
    {
      double d2 = 0, c2 = 0, dummy3 = 0;
      plus_in_place(cx, fresh(), w, w.t2, w.tf, v.stash, o_sb, o_f, d2, c2);
      VIO_PARFOR(q, F)
        v.stash[o_v + 3 * nv + q] = w.gf[q], v.stash[o_v + 3 * nv + F + q] = w.hff[q], v.stash[o_v + 3 * nv + 2 * F + q] = w.gnf[q];
      VIO_PARFOR(i, np) v.stash[o_v + i] = w.gp[i], v.stash[o_v + nv + i] = w.dp[i], v.stash[o_v + 2 * nv + i] = w.gnp[i];
      block_sum3(cx, d2, c2, dummy3);
      step_norm = sqrt(d2), cand_norm = sqrt(c2);
      stamp(cx, ST_V_PLUS);
    }
    double cand_cost = evaluate(cx, fresh(), w, w.xpose, w.xsb, w.xfeat, /*jac=*/speculate, true, false, /*keep_aux=*/!speculate);
    if (!isfinite(cand_cost)) cand_cost = 1.7976931348623157e308;
    auto restore_iterate = [&]() {  // x <- the iterate the candidate replaced
      VIO_PARFOR(q, npose7) w.xpose[q] = v.stash[q];
      VIO_PARFOR(q, v.P * 9) w.xsb[q] = v.stash[o_sb + q];
      VIO_PARFOR(q, F) w.xfeat[q] = v.stash[o_f + q];
    };
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) {                                // ParameterToleranceReached
      restore_iterate();
      termination = 1;
      break;
    }
    double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= 1e-6 * x_cost) {                                 // FunctionToleranceReached
      restore_iterate();
      termination = 1;
      break;
    }
    double rel = (ev_cur - cand_cost) / model_cost_change;                    // StepQuality
    double hist = (ev_ref - cand_cost) / (ev_acc_ref + model_cost_change);
    double rho = fmax(rel, hist);
    if (rho > 1e-3) {
      x_norm = cand_norm;
      x_cost = cand_cost;  // (x is the candidate)
      if (rho < 0.25) radius *= 0.5;                                          // StepAccepted
      if (rho > 0.75) radius = fmax(radius, 3.0 * dogleg_step_norm);
      mu = fmax(min_mu, 2.0 * mu / mu_inc);
      reuse = false;
      ev_cur = cand_cost, ev_acc_cand += model_cost_change, ev_acc_ref += model_cost_change;
      if (ev_cur < ev_min) ev_min = ev_cur, ev_cand = ev_cur, ev_acc_cand = 0;
      else if (ev_cur > ev_cand) ev_cand = ev_cur, ev_acc_cand = 0;
      ev_ref = ev_cand, ev_acc_ref = ev_acc_cand;
      last_ok = true;
      n_ok++;
      recorded = it + 1, min_rec = fmin(min_rec, x_cost);
      if (speculate) {  // the linearization is in place
        gmax = grad_max_norm();
        record(it, x_cost, radius, step_norm, rho, gmax, true, true);
        stamp(cx, ST_V_GMAX);
      } else {          // it follows at the head of the next iteration, the record with it
        relin = true, relin_reuse = true, rec_pending = true;
        rec_it = it, rec_step_norm = step_norm, rec_rho = rho;
      }
    } else {
      radius *= 0.5;                                                          // StepRejected
      reuse = true;
      last_ok = false;
      n_bad++;
      record(it, cand_cost, radius, step_norm, rho, 0.0, true, false);
      recorded = it + 1, min_rec = fmin(min_rec, cand_cost);
      // back to the iterate and, after a speculative evaluation, to the vectors of ITS linearization: the next dogleg step is
      // formed from them (the matrix buffer then holds the candidate's linearization, which nothing reads: an invalid next
      // step re-evaluates at x)
      restore_iterate();
      if (speculate) {
        VIO_PARFOR(q, F) w.gf[q] = v.stash[o_v + 3 * nv + q], w.hff[q] = v.stash[o_v + 3 * nv + F + q], w.gnf[q] = v.stash[o_v + 3 * nv + 2 * F + q];
        VIO_PARFOR(i, np) w.gp[i] = v.stash[o_v + i], w.dp[i] = v.stash[o_v + nv + i], w.gnp[i] = v.stash[o_v + 2 * nv + i];
      }
      VIO_SYNC();
      stamp(cx, ST_V_REST);
    }
  }
  if (cx.tid == 0) {
    sd[1] = min_rec;
    si[0] = recorded, si[1] = termination, si[2] = n_ok, si[3] = n_bad;
  }
  VIO_SYNC();
}

// first pose column speed-bias block k couples to: the IMU chain reaches the frame and both neighbours, the block the
// prior keeps reaches every pose of the prior
template <class WK>
VIO_DEV void init_band_reach(const Ctx &cx, const WinView &v, WK &w) {
  VIO_PARFOR(k, v.P) {
    int lo = 6 * (k > 0 ? k - 1 : 0), pr = 0;
    for (int b = 0; b < v.prior_nb; b++)
      if (v.pr_kind[b] == 1 && v.pr_index[b] == k) lo = 0, pr = 1;
    w.sbr[2 * k] = lo, w.sbr[2 * k + 1] = pr;
  }
}

#ifndef VIO_HOST_BUILD
// A helper workgroup of a cooperative window (member >= 1): serves the owner's commands until COOP_EXIT. Its LDS has the
// owner's layout; it only ever touches the staging area, the evaluation point and the accumulators of its partial sums.
template <class WK>
VIO_DEV void coop_helper(const Ctx &cx, const WinView &v, WK &w) {
  const CoopLayout L = CoopLayout::make(v.Pcap, v.Fcap, v.nblk_cap);
  const int np = v.np, F = v.F, nFr = v.P + v.has_loop;
  const size_t fs = ((size_t)v.Fcap + 7) & ~(size_t)7;
  double *part = v.coop + L.o_part + (size_t)(cx.member - 1) * L.part;
  init_band_reach(cx, v, w);  // (the band's trailing product, COOP_SYRK, skips the tiles a block's fill does not reach)
  VIO_SYNC();
  for (;;) {
    const int cmd = coop_wait_cmd(cx, v);
    if (cmd == COOP_EVAL) {
      VIO_PARFOR(q, 7 * nFr) w.xpose[q] = v.coop[L.o_pose + q];
      VIO_PARFOR(q, F) w.xfeat[q] = v.coop[L.o_feat + q];
      VIO_PARFOR(q, 7) w.ex[q] = v.coop[L.o_ex + q];
      VIO_PARFOR(q, np) w.gp[q] = 0.0;
      VIO_PARFOR(q, F) {
        w.gf[q] = 0.0, w.hff[q] = 0.0;
        w.cfeat[q] = w.stf[q] = w.gnf[q] = w.tf[q] = w.ef[q] = w.einv[q] = 0.0;
      }
      VIO_PARFOR(q, nFr * 36) w.ppd[q] = 0.0;
      VIO_SYNC();
      VIO_PARFOR(i, nFr + 1) {  // rotation matrices (evaluate())
        const bool is_ex = i == nFr;
        double R[9];
        if (is_ex) qtoR(Quat{w.ex[3], w.ex[4], w.ex[5], w.ex[6]}, R);
        else qtoR(Quat{w.xpose[7 * i + 3], w.xpose[7 * i + 4], w.xpose[7 * i + 5], w.xpose[7 * i + 6]}, R);
        auto dst = w.rot + 9 * (is_ex ? v.P + 1 : i);
        for (int k = 0; k < 9; k++) dst[k] = R[k];
      }
      VIO_SYNC();
      const double cost = block_sum(cx, projections_jac(cx, v, w, w.xpose, w.xfeat, true, cx.member, cx.coop, false));
      ldsd whv[6] = {w.cfeat, w.stf, w.gnf, w.tf, w.ef, w.einv};
      VIO_PARFOR(q, np) part[L.p_gp + q] = w.gp[q];
      VIO_PARFOR(q, nFr * 36) part[L.p_ppd + q] = w.ppd[q];
      VIO_PARFOR(f, F) {
        part[L.p_f + f] = w.hff[f], part[L.p_f + fs + f] = w.gf[f];
#pragma unroll
        for (int c = 0; c < 6; c++) part[L.p_f + (2 + c) * fs + f] = whv[c][f];
      }
      if (cx.tid == 0) part[L.p_cost] = cost;
    } else if (cmd == COOP_SCHUR) {
      VIO_PARFOR(f, F) w.einv[f] = v.coop[L.o_einv + f], w.tf[f] = v.coop[L.o_tf + f], w.cfeat[f] = 0.0;
      VIO_PARFOR(q, 16 * ((v.n6 + 15) >> 4)) w.xt[q] = v.coop[L.o_pose + q];  // u_p (only meaningful when the owner fuses W^T u_p: read either way)
      VIO_SYNC();
      schur_general(cx, v, w, cx.member, cx.coop, w.cfeat);
      VIO_SYNC();
      VIO_PARFOR(f, F) part[L.p_f + f] = w.cfeat[f];
    } else if (cmd == COOP_SYRK) {
      band_syrk_share(cx, v, w, cx.member, cx.coop);
    } else {
      break;  // COOP_EXIT (or a timed-out wait)
    }
    coop_done(cx, v);
  }
}
#endif

// =====================================================================================================
// Whole solve for one window: load, setup, minimize, raw outputs, new2old, outputs
// =====================================================================================================
// NW: waves of the workgroup (blockDim.x / 64)
template <bool REGS, int NW, class WK, class VP, class WP>
VIO_DEV void solve_window(const Ctx &cx, const WinView &v, WK &w, const VP &fresh, const WP &fresh_work) {
  const int P = v.P, F = v.F;
  VIO_PARFOR(q, P * 7) w.xpose[q] = v.pose0[q];
  VIO_PARFOR(q, P * 9) w.xsb[q] = v.sb0[q];
  VIO_PARFOR(q, F) w.xfeat[q] = v.feat0[q];
  VIO_PARFOR(q, 7) w.ex[q] = v.ex[q];
  if (v.has_loop) VIO_PARFOR(q, 7) w.xpose[7 * P + q] = v.pose0[7 * v.loop_frame + q];  // VINS.cpp:590-591
  VIO_PARFOR(q, v.nblk * kBS) w.t1[q] = 0.0, w.t2[q] = 0.0;
  if (cx.tid == 0) w.flag[0] = w.flag[1] = w.flag[2] = w.flag[3] = 0;
  init_band_reach(cx, v, w);
  VIO_SYNC();
#ifndef VIO_EMUL
  if (cx.prof && cx.tid == cx.prof_tid) {
    for (int q = 0; q < ST_COUNT; q++) cx.lprof[q] = 0;
    cx.lprof[ST_COUNT - 1] = clock64();
    cx.lprof[ST_TOTAL] = -cx.lprof[ST_COUNT - 1];
  }
#endif
  VIO_PARFOR(q, v.nslots) v.sfact[q] = -1;
  VIO_SYNC();
  VIO_PARFOR(q, v.nslots) v.srec_i[q] = -1;
  VIO_SYNC();
  VIO_PARFOR(k, v.M) {
    const int sl = v.fslot[k];
    v.sfact[sl] = k;
    v.srec_i[sl] = v.fhost[k] | (v.ftarget[k] << 8) | (v.ffeat[k] << 16);
    double *d = v.srec_d + 6 * (size_t)sl;
    for (int c = 0; c < 3; c++) d[c] = v.pts_i[3 * k + c], d[3 + c] = v.pts_j[3 * k + c];
  }
  build_gram_pieces(cx, v, w);
  VIO_PARFOR(f, F) w.fh[f] = v.fstart[f + 1] > v.fstart[f] ? v.fhost[v.fstart[f]] : -1;
  VIO_PARFOR(q, (int)tri_doubles(v.nrows)) v.PP[q] = 0.0;  // (blocks without a (host, target) bucket stay zero for the whole solve)
  setup_imu_info(cx, fresh(), w.stage);
  stamp(cx, ST_SETUP_IMU);
  setup_prior(cx, fresh(), w);
  stamp(cx, ST_SETUP_PRIOR);

  minimize<REGS, NW>(cx, v, w, fresh, fresh_work);

  VIO_PARFOR(q, P * 7) v.raw_pose[q] = w.xpose[q];
  VIO_PARFOR(q, P * 9) v.raw_sb[q] = w.xsb[q];
  VIO_PARFOR(q, F) v.raw_feat[q] = w.xfeat[q];
  if (v.has_loop) VIO_PARFOR(q, 7) v.out_loop[q] = w.xpose[7 * P + q];
  VIO_SYNC();
  // ---- new2old (VINS.cpp:131-212) followed by old2new (VINS.cpp:89-129) ----------------------------
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
  stamp(cx, ST_NEW2OLD);
}

#endif  // !VIO_EMUL

}  // namespace vio
