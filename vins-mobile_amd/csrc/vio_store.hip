// vio_store.hip — kernels of the device-resident landmark store (store_core.h): one workgroup per sequence slot.
// Built with -ffp-contract=off: the passes give the bits of the host-side list (vio_window.cpp).
#include "vio_store.h"

#include <stdio.h>
#include <stdlib.h>

using namespace vio;
namespace st = vio::store;

static_assert(st::PH_F == H_F && st::PH_M == H_M && st::PH_HAS_LOOP == H_HAS_LOOP && st::PH_LOOP_FRAME == H_LOOP_FRAME &&
                  st::PH_MARG == H_MARG && st::PH_NPAIRS == H_NPAIRS && st::PH_NSLOTS == H_NSLOTS && st::PH_NREV == H_NREV,
              "store_core.h writes the header slots of batch.h");

namespace {

__device__ __forceinline__ st::Bank bank_of(const StoreDev &S, int slot, int b) {
  const size_t e = ((size_t)b * S.n_slots + slot) * S.d.Lcap;
  return st::Bank{S.fid + e, S.start + e, S.nobs + e, S.flag + e, S.depth + e, S.obs + e * (S.d.W + 1) * 3};
}

// One wave per job: the pre-integration block of an interval from its raw samples (a new interval, or more samples for one
// that absorbed its neighbour's at a non-keyframe slide).
__global__ __launch_bounds__(256) void preint_jobs_kernel(StoreDev S) {
  extern __shared__ double lds_pre[];  // (half a CU like every store kernel, see lds_block; four waves use 44 KB of it)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave;
  if (j >= S.n_imu_jobs) return;
  const double *job = S.imu_jobs + (size_t)j * 16;
  const int idx = (int)job[0], fresh = (int)job[1], n = (int)job[2], first = (int)job[3];
  double *blk = S.preint + (size_t)idx * kPreintDoubles, *side = S.pre_side + (size_t)idx * preint::kSide;
  ldsd lds = (ldsd)lds_pre + wave * preint::kLdsDoubles;
  preint::State s;
  if (fresh) preint::init_wave(s, lds, lane, job + 4, job + 7, job + 10, job + 13);
  else preint::load_wave(s, lds, lane, blk, side);
  for (int i = 0; i < n; i++) {
    const double *smp = S.imu_samples + (size_t)(first + i) * 7;
    preint::propagate_wave(s, lds, lane, smp[0], smp + 1, smp + 4, S.noise);
  }
  preint::store_wave(s, lds, lane, blk, side);
}

__global__ __launch_bounds__(st::kThreads) void store_ingest_kernel(StoreDev S) {
  extern __shared__ int lds_raw[];
  const int slot = blockIdx.x;
  // pre-integration blocks that arrive with this frame (any slot's: the blocks are spread over the grid)
  for (int k = blockIdx.x; k < S.n_pre; k += gridDim.x) {
    const double *src = S.pre_blk + (size_t)k * (kPreintDoubles + 6);
    double *dst = S.preint + (size_t)S.pre_idx[k] * kPreintDoubles, *side = S.pre_side + (size_t)S.pre_idx[k] * preint::kSide;
    for (int i = threadIdx.x; i < kPreintDoubles; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x < 6) side[threadIdx.x] = src[kPreintDoubles + threadIdx.x];
  }
  if (!S.active[slot]) return;
  st::Cx cx{(int)threadIdx.x, (int)blockDim.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x & ~63)};
  ldsd dbase;
  st::Lds l = st::carve_lds<ldsi, ldsd>(S.d, (ldsi)lds_raw, &dbase);
  if (slot == 0) l.prof = S.prof;
  int *ctl = S.ctl + (size_t)slot * st::C_COUNT;
  const int P = S.d.W + 1;
  st::store_ingest(cx, S.d, bank_of(S, slot, ctl[st::C_BANK]), ctl, l, S.obs_in + S.obs_off[slot], S.n_obs[slot],
                   S.Ps + (size_t)slot * 3 * P, S.Rs + (size_t)slot * 9 * P, S.tic, S.ric);
}

__global__ __launch_bounds__(st::kThreads) void store_pack_kernel(StoreDev S, BatchPtrs B, int chunk) {
  extern __shared__ int lds_raw[];
  const int slot = blockIdx.x;
  if (!S.active[slot]) return;
  st::Cx cx{(int)threadIdx.x, (int)blockDim.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x & ~63)};
  ldsd dbase;
  st::Lds l = st::carve_lds<ldsi, ldsd>(S.d, (ldsi)lds_raw, &dbase);
  if (slot == 0) l.prof = S.prof;
  // behind the store's own scratch: bucket counters and the factor keys
  const int np1 = S.d.W + 2;
  ldsi bins = (ldsi)(l.cam + 12 * (S.d.W + 1) + 12);
  VIO_AS3 unsigned short *keys = (VIO_AS3 unsigned short *)(bins + 3 * np1 * np1 + (np1 & 1));
  VIO_AS3 unsigned short *own = keys + ((B.d.Mcap + 8 + 3) & ~3);
  int *ctl = S.ctl + (size_t)slot * st::C_COUNT;
  const BatchStrides &s = B.s;
  const size_t b = slot;
  st::PackOut o;
  o.hdr = const_cast<int *>(B.hdr) + b * kHdrInts;
  o.feat = const_cast<double *>(B.feat) + b * s.feat;
  o.fhost = const_cast<int *>(B.fhost) + b * s.fint, o.ftarget = const_cast<int *>(B.ftarget) + b * s.fint;
  o.ffeat = const_cast<int *>(B.ffeat) + b * s.fint, o.fslot = const_cast<int *>(B.fslot) + b * s.fint;
  o.fstart = const_cast<int *>(B.fstart) + b * s.fstart;
  o.pair_h = const_cast<int *>(B.pair_h) + b * s.pair, o.pair_t = const_cast<int *>(B.pair_t) + b * s.pair;
  o.pair_s0 = const_cast<int *>(B.pair_s0) + b * s.pair, o.pair_s1 = const_cast<int *>(B.pair_s1) + b * s.pair;
  o.pts_i = const_cast<double *>(B.pts_i) + b * s.pts, o.pts_j = const_cast<double *>(B.pts_j) + b * s.pts;
  o.Fcap = B.d.Fcap, o.Mcap = B.d.Mcap, o.pair_cap = B.d.pair_cap, o.slot_cap = 2 * (size_t)B.d.Mcap + B.d.pair_cap + 2;  // = slot_capacity(B.d) (batch.h, host function)
  st::LoopIn lp;
  lp.frame = S.loop_frame[slot], lp.n = S.loop_n[slot], lp.ids = S.loop_ids + S.loop_off[slot], lp.xy = S.loop_xy + 2 * (size_t)S.loop_off[slot];
  st::store_pack(cx, S.d, bank_of(S, slot, ctl[st::C_BANK]), ctl, l, o, chunk, keys, own, bins, lp);
}

__global__ __launch_bounds__(st::kThreads) void store_finish_kernel(StoreDev S, BatchPtrs B) {
  extern __shared__ int lds_raw[];
  const int slot = blockIdx.x;
  if (!S.active[slot]) return;
  st::Cx cx{(int)threadIdx.x, (int)blockDim.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x & ~63)};
  ldsd dbase;
  st::Lds l = st::carve_lds<ldsi, ldsd>(S.d, (ldsi)lds_raw, &dbase);
  if (slot == 0) l.prof = S.prof;
  int *ctl = S.ctl + (size_t)slot * st::C_COUNT;
  const BatchStrides &s = B.s;
  const int bank = ctl[st::C_BANK];
  st::store_finish(cx, S.d, bank_of(S, slot, bank), bank_of(S, slot, 1 - bank), ctl, S.ctld + (size_t)slot * st::kCtlDoubles, l,
                   B.out_feat + (size_t)slot * s.out_feat, B.out_pose + (size_t)slot * s.out_pose, B.out_sb + (size_t)slot * s.out_sb,
                   S.tic, S.ric);
  __syncthreads();
  // slideWindow, MARGIN_OLD: the pre-integration of interval i + 1 becomes that of interval i (VINS.cpp:1160-1187); the
  // newest interval arrives with the next frame
  if (ctl[st::C_STATUS] == VIO_OK && ctl[st::C_FAIL] == 0 && ctl[st::C_MARG] == VIO_MARGIN_OLD) {
    // (every work-item reads what it moves before anything is written: one barrier, not one per interval)
    {
      double *sd = S.pre_side + (size_t)slot * S.d.W * preint::kSide;
      const int ns = (S.d.W - 1) * preint::kSide;
      double keep = 0;
      if ((int)threadIdx.x < ns) keep = sd[preint::kSide + threadIdx.x];
      __syncthreads();
      if ((int)threadIdx.x < ns) sd[threadIdx.x] = keep;
    }
    double *pre = S.preint + (size_t)slot * S.d.W * kPreintDoubles;
    const int total = (S.d.W - 1) * kPreintDoubles;
    constexpr int kPer = 24;  // doubles per work-item: enough for W <= 13 at 256 work-items
    if (total <= kPer * (int)blockDim.x) {
      double v[kPer];
#pragma unroll
      for (int j = 0; j < kPer; j++) {
        const int k = threadIdx.x + j * blockDim.x;
        if (k < total) v[j] = pre[kPreintDoubles + k];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < kPer; j++) {
        const int k = threadIdx.x + j * blockDim.x;
        if (k < total) pre[k] = v[j];
      }
    } else {
      for (int i = 0; i + 1 < S.d.W; i++) {
        for (int k = threadIdx.x; k < kPreintDoubles; k += blockDim.x) pre[(size_t)i * kPreintDoubles + k] = pre[(size_t)(i + 1) * kPreintDoubles + k];
        __syncthreads();
      }
    }
  }
}

}  // namespace

namespace vio {

size_t store_pack_lds_bytes(const store::Dims &d, int Mcap) {
  const int np1 = d.W + 2;
  return st::lds_bytes(d) + sizeof(int) * (3 * np1 * np1 + 1) + 2 * sizeof(unsigned short) * ((size_t)Mcap + 12);
}

// Every store kernel asks for exactly half (or all) of a CU's LDS: a workgroup whose block sits somewhere in the middle
// of the CU's LDS while a window-kernel workgroup of another context is placed next to it leaves, when it ends, two free
// pieces neither of which takes the next window-kernel workgroup (half a CU each) -- measured as two contexts' window
// kernels running one after the other. Blocks of the window kernel's own size keep the halves whole.
static size_t lds_block(size_t need) {
  static const bool exact = getenv("VIO_AMD_STORE_LDS_EXACT") && getenv("VIO_AMD_STORE_LDS_EXACT")[0] == '1';
  if (exact) return need;
  return need <= kLdsBytes / 2 ? kLdsBytes / 2 : kLdsBytes;
}
static int raise_lds(const void *fn, size_t lds) {
  if (lds > kLdsBytes) return VIO_ECAP;
  if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VIO_ENODEV;
  return VIO_OK;
}

#define STORE_HIP_OK(expr)                                                                                      \
  do {                                                                                                          \
    hipError_t e_ = (expr);                                                                                     \
    if (e_ != hipSuccess) {                                                                                     \
      fprintf(stderr, "vio_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__);    \
      return VIO_ENODEV;                                                                                        \
    }                                                                                                           \
  } while (0)

int store_launch_imu(const StoreDev &S, hipStream_t stream) {
  if (S.n_imu_jobs <= 0) return VIO_OK;
  const size_t lds = lds_block(4 * preint::kLdsDoubles * sizeof(double));
  const int rc = raise_lds((const void *)preint_jobs_kernel, lds);
  if (rc != VIO_OK) return rc;
  hipLaunchKernelGGL(preint_jobs_kernel, dim3((S.n_imu_jobs + 3) / 4), dim3(256), lds, stream, S);
  STORE_HIP_OK(hipGetLastError());
  return VIO_OK;
}

int store_launch_ingest(const StoreDev &S, hipStream_t stream) {
  const size_t lds = lds_block(st::lds_bytes(S.d));
  const int rc = raise_lds((const void *)store_ingest_kernel, lds);
  if (rc != VIO_OK) return rc;
  hipLaunchKernelGGL(store_ingest_kernel, dim3(S.n_slots), dim3(st::kThreads), lds, stream, S);
  STORE_HIP_OK(hipGetLastError());
  return VIO_OK;
}

int store_launch_pack(const StoreDev &S, const BatchPtrs &B, int chunk, hipStream_t stream) {
  const size_t lds = lds_block(store_pack_lds_bytes(S.d, B.d.Mcap));
  const int rc = raise_lds((const void *)store_pack_kernel, lds);
  if (rc != VIO_OK) return rc;
  hipLaunchKernelGGL(store_pack_kernel, dim3(S.n_slots), dim3(st::kThreads), lds, stream, S, B, chunk);
  STORE_HIP_OK(hipGetLastError());
  return VIO_OK;
}

int store_launch_finish(const StoreDev &S, const BatchPtrs &B, hipStream_t stream) {
  const size_t lds = lds_block(st::lds_bytes(S.d));
  const int rc = raise_lds((const void *)store_finish_kernel, lds);
  if (rc != VIO_OK) return rc;
  hipLaunchKernelGGL(store_finish_kernel, dim3(S.n_slots), dim3(st::kThreads), lds, stream, S, B);
  STORE_HIP_OK(hipGetLastError());
  return VIO_OK;
}

}  // namespace vio
