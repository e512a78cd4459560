// vio_phase.hip — gfx950 kernels of the phase path: VINS::solve_ceres (VINS_ios/VINS.cpp:480-831) as a fixed sequence of
// launches with no host round trip (phase_core.h): factor-parallel linearization kernels alternating with per-window
// trust-region step kernels.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "batch.h"
#include "marg_core.h"
#include "phase_core.h"
#include "vio_phase.h"
#include "vio_amd.h"

using namespace vio;

namespace {

constexpr int kThreadsLds = 256;
constexpr size_t kLdsLimit = vio::kLdsBytes;

constexpr int kThreadsLin = vio::kLinThreads;

__global__ __launch_bounds__(256, 3) void vio_phase_setup_kernel(BatchPtrs B) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = B.order ? B.order[blockIdx.x] : (int)blockIdx.x;
  WinView v = make_view(B, b);
  const PhaseView pv = make_phase_view(B, b);
  SetupWork sw;
  carve_setup(B.d, (ldsd)smem, &sw);
  Ctx cx;
  cx.tid = threadIdx.x, cx.nt = blockDim.x, cx.prof = nullptr, cx.red = nullptr, cx.lprof = nullptr;
  phase_setup(cx, v, pv, sw);
}

// Stage clock of a phase kernel (vio_backend_set_profile): the counters of one launch are added to the window's row, so
// that a whole launch sequence accumulates like the single-launch kernel's counters do.
__device__ __forceinline__ void prof_begin(const Ctx &cx) {
  if (cx.prof && cx.tid == cx.prof_tid) {
    for (int q = 0; q < ST_COUNT; q++) cx.lprof[q] = 0;
    cx.lprof[ST_COUNT - 1] = clock64();
    cx.lprof[ST_TOTAL] = -cx.lprof[ST_COUNT - 1];
  }
}
__device__ __forceinline__ void prof_end(const Ctx &cx, int total_stage) {
  if (cx.prof && cx.tid == cx.prof_tid) {
    cx.lprof[ST_TOTAL] += clock64();
    const long long t = cx.lprof[ST_TOTAL];
    cx.lprof[ST_TOTAL] = 0;
    for (int q = 0; q < ST_COUNT - 1; q++) cx.prof[q] += cx.lprof[q];
    cx.prof[total_stage] += t, cx.prof[ST_TOTAL] += t;  // (ST_TOTAL: every kernel of the sequence; total_stage: this kind)
  }
}

__global__ __launch_bounds__(kThreadsLin, 2) void vio_phase_lin_kernel(BatchPtrs B, long long *prof, int prof_tid) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = B.order ? B.order[blockIdx.x] : (int)blockIdx.x;
  const WinView v = make_view(B, b);
  const PhaseView pv = make_phase_view(B, b);
  LinWork lw;
  carve_lin(B.d, (ldsd)smem, &lw);
  Ctx cx;
  cx.tid = threadIdx.x, cx.nt = blockDim.x, cx.red = lw.red;
  cx.prof = prof ? prof + (size_t)b * ST_COUNT : nullptr, cx.prof_tid = prof_tid, cx.lprof = reinterpret_cast<VIO_AS3 long long *>(lw.lprof);
  prof_begin(cx);
  phase_linearize(cx, v, pv, lw);
  prof_end(cx, ST_MARG_BUILD);  // (linearize kernels: their share of the total in the slot of a stage they do not run)
}

template <bool LDS_ASP>
__global__ __launch_bounds__(kThreadsLds, 2) void vio_phase_step_kernel(BatchPtrs B, int wrot_forced, long long *prof, int prof_tid) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = B.order ? B.order[blockIdx.x] : (int)blockIdx.x;
  WinView v = make_view(B, b);
  const PhaseView pv = make_phase_view(B, b);
  typedef typename std::conditional<LDS_ASP, ldsd, double *>::type AspP;
  ldsd lds = (ldsd)smem;
  BatchDims dims = B.d;
  dims.lds_asp = LDS_ASP ? 1 : 0;
  const Carved<ldsd, AspP> cw = carve_all<ldsd, AspP>(dims, true, blockDim.x, lds, nullptr, v.AspG);
  WorkT<ldsd, AspP> w = cw.w;
  Ctx cx;
  cx.tid = threadIdx.x, cx.nt = blockDim.x;
  cx.prof = prof ? prof + (size_t)b * ST_COUNT : nullptr;
  {
    if (threadIdx.x == 0) cw.w.flag[0] = (int)__builtin_amdgcn_s_getreg(0x1C04) & 3;
    __syncthreads();
    cx.wrot = wrot_forced >= 0 ? wrot_forced : cw.w.flag[0];
    __syncthreads();
  }
  // (the stage clock follows the chain wave, like in the single-launch kernel)
  if (prof_tid == 0) cx.prof_tid = ((kThreadsLds / 64 - cx.wrot) & (kThreadsLds / 64 - 1)) * 64;
  else cx.prof_tid = (((prof_tid >> 6) - cx.wrot) & (kThreadsLds / 64 - 1)) * 64;
  cx.red = cw.red, cx.lprof = cw.lprof;
  prof_begin(cx);
  phase_step<true, kThreadsLds / 64>(cx, v, pv, w);
  prof_end(cx, ST_MARG_CHOL);  // (step kernels: likewise)
}

template <bool LDS_ASP>
__global__ __launch_bounds__(kThreadsLds, 2) void vio_phase_finish_kernel(BatchPtrs B, MargPtrs MP, int lds_doubles) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = B.order ? B.order[blockIdx.x] : (int)blockIdx.x;
  WinView v = make_view(B, b);
  const PhaseView pv = make_phase_view(B, b);
  typedef typename std::conditional<LDS_ASP, ldsd, double *>::type AspP;
  ldsd lds = (ldsd)smem;
  BatchDims dims = B.d;
  dims.lds_asp = LDS_ASP ? 1 : 0;
  const Carved<ldsd, AspP> cw = carve_all<ldsd, AspP>(dims, true, blockDim.x, lds, nullptr, v.AspG);
  WorkT<ldsd, AspP> w = cw.w;
  Ctx cx;
  cx.tid = threadIdx.x, cx.nt = blockDim.x, cx.prof = nullptr, cx.prof_tid = 0, cx.wrot = 0;
  cx.red = cw.red, cx.lprof = cw.lprof;
  const size_t state_end = cw.state_end_doubles;
  phase_finish(cx, v, pv, w);
  MargOut mo;
  int *mi = MP.ints + (size_t)b * MP.s_ints;
  mo.n = mi, mo.kind = mi + 4, mo.index = mo.kind + kMaxPriorBlocks, mo.offset = mo.index + kMaxPriorBlocks;
  mo.x0 = MP.x0 + (size_t)b * MP.s_x0, mo.J = MP.J + (size_t)b * MP.s_J, mo.r = MP.r + (size_t)b * MP.s_r;
  mo.scratch = nullptr;
  mo.ncap = B.d.Ncap;
  if (B.ptab && B.ptab[b].mJ) mo.x0 = B.ptab[b].mx0, mo.J = B.ptab[b].mJ, mo.r = B.ptab[b].mr, mo.ncap = B.ptab[b].ncap;
  MargWorkT<ldsd> mw = carve_marg_all<ldsd>(B.d, true, lds + state_end, nullptr, (size_t)lds_doubles - state_end).m;
  __syncthreads();
  marginalize_window_impl(cx, v, w.xpose, w.xsb, w.xfeat, w.ex, mw, mo);
}


}  // namespace

namespace vio {

void phase_lds_need(const BatchDims &d, size_t *setup_bytes, size_t *lin_bytes) {
  *setup_bytes = carve_setup(d, nullptr, nullptr);
  *lin_bytes = carve_lin(d, nullptr, nullptr);
}

int phase_prepare() {
  const int lim = (int)kLdsLimit;
  const bool ok = hipFuncSetAttribute((const void *)vio_phase_setup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess &&
                  hipFuncSetAttribute((const void *)vio_phase_lin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess &&
                  hipFuncSetAttribute((const void *)vio_phase_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess &&
                  hipFuncSetAttribute((const void *)vio_phase_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess &&
                  hipFuncSetAttribute((const void *)vio_phase_finish_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess &&
                  hipFuncSetAttribute((const void *)vio_phase_finish_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess;
  return ok ? VIO_OK : VIO_ENODEV;
}

void phase_launch(const BatchPtrs &Bl, const MargPtrs &MP, int n, size_t lds_setup, size_t lds_lin, size_t lds_step, hipStream_t st) {
  const dim3 grid(n);
  const int ldsd_n = (int)(lds_step / sizeof(double));
  const bool asp = Bl.d.lds_asp != 0;
  hipLaunchKernelGGL(vio_phase_setup_kernel, grid, dim3(256), lds_setup, st, Bl);
  hipLaunchKernelGGL(vio_phase_lin_kernel, grid, dim3(kThreadsLin), lds_lin, st, Bl, MP.prof, MP.prof_tid);
  for (int k = 0; k <= Bl.d.max_iter; k++) {
    if (asp) hipLaunchKernelGGL(vio_phase_step_kernel<true>, grid, dim3(kThreadsLds), lds_step, st, Bl, MP.wrot, MP.prof, MP.prof_tid);
    else hipLaunchKernelGGL(vio_phase_step_kernel<false>, grid, dim3(kThreadsLds), lds_step, st, Bl, MP.wrot, MP.prof, MP.prof_tid);
    if (k < Bl.d.max_iter) hipLaunchKernelGGL(vio_phase_lin_kernel, grid, dim3(kThreadsLin), lds_lin, st, Bl, MP.prof, MP.prof_tid);
  }
  if (asp) hipLaunchKernelGGL(vio_phase_finish_kernel<true>, grid, dim3(kThreadsLds), lds_step, st, Bl, MP, ldsd_n);
  else hipLaunchKernelGGL(vio_phase_finish_kernel<false>, grid, dim3(kThreadsLds), lds_step, st, Bl, MP, ldsd_n);
}

}  // namespace vio
