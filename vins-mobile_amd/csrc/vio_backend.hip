// vio_backend.hip — gfx950 kernels and C ABI of the sliding-window back-end (include/vio_amd.h).
//
// One workgroup owns one window for the whole of VINS::solve_ceres (VINS_ios/VINS.cpp:480-831): solve, new2old,
// marginalization. At W <= 12 the reduced system (pose matrix 22 KB + speed-bias band 14 KB, solver_core.h) and every
// vector of the solve fit in 80 KB of LDS: 256 threads (4 wave64) per window, TWO windows resident per CU, so that one
// window's serial pivot chains run in the shadow of the other's parallel phases. The batch of independent sequences
// is the grid: a launch of >= 512 windows fills the chip and each XCD's L2 only ever sees its own windows' scratch.
// Larger windows keep the pose matrix in global scratch and take a CU each (512 threads).
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include <chrono>

#include "batch.h"
#include "vio_pool.h"
#include "vio_device.h"
#include "marg_core.h"
#include "vio_window_variants.h"
#include "vio_amd.h"

using namespace vio;

namespace {

using vio_wk::kThreadsGlb;
using vio_wk::kThreadsLds;
constexpr size_t kLdsLimit = vio::kLdsBytes;      // one CU
constexpr size_t kLdsHalf = vio::kLdsBytes / 2;   // two resident workgroups per CU


// Debug aid (VIO_AMD_POISON=1): every launch is preceded by NaN patterns in the whole LDS of every CU and in all device
// scratch / output buffers, so that a read of something the kernel did not write itself cannot go unnoticed.
__global__ __launch_bounds__(1024) void poison_lds_kernel(int n_doubles) {
  extern __shared__ __attribute__((aligned(16))) double poison_smem[];
  for (int i = threadIdx.x; i < n_doubles; i += blockDim.x) poison_smem[i] = __longlong_as_double(0x7ff8dead0000beefLL);
  __syncthreads();
  if (poison_smem[(threadIdx.x * 7) % n_doubles] == 1.0) poison_smem[0] = 2.0;  // keep the stores alive
}

#define HIP_OK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "vio_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VIO_ENODEV;                                                                   \
    }                                                                                      \
  } while (0)

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  int ensure(size_t count) {
    if (count <= n && p) return VIO_OK;
    {
      static const bool log = getenv("VIO_AMD_HOST_TIMING") && getenv("VIO_AMD_HOST_TIMING")[0] == '1';
      if (log) fprintf(stderr, "vio_amd: device buffer grows %zu -> %zu elements of %zu bytes\n", n, count, sizeof(T));
    }
    if (p) (void)hipFree(p);
    p = nullptr, n = 0;
    if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return VIO_ENOMEM;
    n = count;
    return VIO_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr, n = 0;
  }
};

}  // namespace

// The instantiations of the window kernel, one translation unit each (vio_wk_unit.hip, csrc/Makefile).
#define VIO_WK_DECL(v) vio_wk::VariantFns vio_wk_variant_##v##_0(); vio_wk::VariantFns vio_wk_variant_##v##_1();
VIO_WK_DECL(0) VIO_WK_DECL(1) VIO_WK_DECL(2) VIO_WK_DECL(3) VIO_WK_DECL(4) VIO_WK_DECL(5)
#undef VIO_WK_DECL
const vio_wk::VariantFns &vio_wk::variant(int v, bool prof) {
  static const VariantFns tab[kVariants][2] = {{vio_wk_variant_0_0(), vio_wk_variant_0_1()}, {vio_wk_variant_1_0(), vio_wk_variant_1_1()},
                                               {vio_wk_variant_2_0(), vio_wk_variant_2_1()}, {vio_wk_variant_3_0(), vio_wk_variant_3_1()},
                                               {vio_wk_variant_4_0(), vio_wk_variant_4_1()}, {vio_wk_variant_5_0(), vio_wk_variant_5_1()}};
  return tab[v][prof ? 1 : 0];
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static bool host_timing() {
  static const bool on = getenv("VIO_AMD_HOST_TIMING") && getenv("VIO_AMD_HOST_TIMING")[0] == '1';
  return on;
}

struct vio_resident;
static void resident_destroy(vio_resident *r);

struct vio_backend {
  VioConfig cfg;
  vio_resident *res = nullptr;  // device-resident path (vio_resident.h), null until reserved
  int peers = 2;  // contexts whose window kernels share the device at the same time (vio_backend_set_peers)
  int device = -1;  // HIP device the context lives on (current device at create)
  int max_batch = 0;
  hipStream_t stream = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  // current batch
  int n = 0;
  bool uploaded = false;
  hipStream_t last_stream = nullptr;  // where the last launch went: what sync / download wait for
  hipEvent_t upload_done = nullptr;   // recorded behind the upload's copies on `stream`: launches on another stream wait for it
  bool lds_matrix = true;  // every window of the batch runs the LDS variant
  // a batch is split by variant: windows whose matrix and vectors fit the CU's LDS, and the rest (relocalization pose,
  // very many landmarks) with the matrix in global scratch
  int n_lds = 0, n_glb = 0;
  BatchDims d_lds, d_glb;
  size_t lds_bytes_glb = 0;
  DevBuf<int> d_order;
  HostVec<int> h_order;
  bool profile = false;
  size_t lds_bytes = 0;
  int threads_lds = kThreadsLds;
  int n_cus = 256;  // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  bool static_w = false;  // every window of the LDS launch has W = vio_wk::kStaticW frames and the batch has that window's capacities
  bool coop_ok = false;  // the global-matrix windows of the uploaded batch may run as cooperative windows (host-packed, no bucket across chunks)
  DevBuf<long long> d_prof;
  HostBatch hb;
  BatchPtrs B;
  MargPtrs MP;
  DevBuf<unsigned char> d_in;  // every input array, laid out like the host staging arena (HostBatch::off)
  DevBuf<int> d_stats_i, d_m_ints;
  DevBuf<double> d_scratch,
      d_hm, d_out_pose, d_out_sb, d_out_feat, d_raw_pose, d_raw_sb, d_raw_feat, d_out_loop, d_stats_d, d_m_x0, d_m_J,
      d_m_r, d_m_scratch;
  // device-resident prior chain (vio_backend_reserve_priors): st_n slots x 2 banks; st_bank[k] = the bank slot k's
  // current prior lives in; the kernel writes the next one into the other bank
  int st_n = 0, st_ncap = 0;
  DevBuf<double> d_st_x0[2], d_st_J[2], d_st_r[2];
  std::vector<unsigned char> st_bank;
  std::vector<int> slot_of;  // [n] slot of window b in this upload or -1
  DevBuf<PriorTab> d_ptab;
  HostVec<PriorTab> h_ptab;
  bool host_prior_in = true, host_prior_out = true, slots_advanced = false;  // any window of this upload moves prior data through the host
  // host copies of the outputs
  HostVec<double> h_out_pose, h_out_sb, h_out_feat, h_raw_pose, h_raw_sb, h_raw_feat, h_out_loop, h_stats_d, h_m_x0,
      h_m_J, h_m_r;
  HostVec<int> h_stats_i, h_m_ints;
};

extern "C" {

const char *vio_version(void) { return "vio_amd 0.1 (gfx950)"; }

int vio_host_pool_width(int32_t *n_pools) {
  if (n_pools) *n_pools = vio::HostPool::count();
  return vio::HostPool::get().width();
}

int vio_hip_runtime(char *path, int32_t cap, int32_t *n_runtimes) {
  const std::vector<std::string> r = vio::hip_runtimes();
  if (n_runtimes) *n_runtimes = (int32_t)r.size();
  if (path && cap > 0) {
    std::string all;
    for (size_t i = 0; i < r.size(); i++) all += (i ? ";" : "") + r[i];
    strncpy(path, all.c_str(), (size_t)cap - 1);
    path[cap - 1] = 0;
  }
  return VIO_OK;
}

void vio_config_default(VioConfig *c) {
  // iPhone7P, global_param.cpp:27-42; feature_tracker.hpp:24-29; global_param.hpp:28-58
  c->window_size = 10, c->max_features = 1000, c->max_factors = 8192, c->max_iterations = 10;
  c->image_rows = 640, c->image_cols = 480, c->max_corners = 70, c->min_dist = 30, c->freq = 3;
  c->lk_win = 21, c->lk_levels = 3, c->lk_max_iters = 30, c->lk_eps = 0.01, c->lk_min_eig = 1e-4;
  c->quality_level = 0.01, c->f_threshold = 1.0, c->f_confidence = 0.99;
  c->fx = 526.600, c->fy = 526.678, c->cx = 243.481, c->cy = 315.280;
  c->gravity = 9.805, c->acc_n = 0.5, c->acc_w = 0.002, c->gyr_n = 0.2, c->gyr_w = 4.0e-5, c->cauchy_a = 1.0;
}

int32_t vio_prior_capacity(int32_t window_size) { return 15 * (window_size + 1) + 6; }

int vio_backend_create(const VioConfig *cfg, int32_t max_batch, vio_backend_t **out) {
  if (!cfg || !out || max_batch < 1) return VIO_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    fprintf(stderr, "vio_amd: no HIP device visible; the back-end has no CPU fallback\n");
    return VIO_ENODEV;
  }
  if (!vio::single_hip_runtime()) return VIO_ENODEV;
  vio_backend *be = new (std::nothrow) vio_backend();
  if (!be) return VIO_ENOMEM;
  be->cfg = *cfg;
  be->max_batch = max_batch;
  be->device = vio::current_device();
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, be->device) == hipSuccess && prop.multiProcessorCount > 0) be->n_cus = prop.multiProcessorCount;
  }
  // the dynamic-LDS ceiling is a property of the FUNCTION, not of a launch: raised once to the CU's whole LDS for both
  // variants (several contexts on several host threads launch these kernels; a per-launch value could be lowered by
  // another thread between this thread's set and its launch)
  for (int v = 0; v < vio_wk::kVariants; v++)
    for (int p = 0; p < 2; p++)
      if (hipFuncSetAttribute(vio_wk::variant(v, p != 0).fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit) != hipSuccess) {
        delete be;
        return VIO_ENODEV;
      }
  if (hipStreamCreateWithFlags(&be->stream, hipStreamNonBlocking) != hipSuccess) {
    delete be;
    return VIO_ENODEV;
  }
  *out = be;
  return VIO_OK;
}

int vio_backend_get_device(const vio_backend_t *be, int32_t *device) {
  if (!be || !device) return VIO_EINVAL;
  *device = be->device;
  return VIO_OK;
}

void vio_backend_destroy(vio_backend_t *be) {
  if (!be) return;
  vio::DeviceScope scope(be->device);
  (void)hipStreamSynchronize(be->stream);
  for (auto &e : be->events) (void)hipEventDestroy(e.first), (void)hipEventDestroy(e.second);
  if (be->upload_done) (void)hipEventDestroy(be->upload_done);
  be->d_in.release(), be->d_stats_i.release(), be->d_m_ints.release();
  DevBuf<double> *db[] = {&be->d_scratch, &be->d_hm,
                          &be->d_out_pose, &be->d_out_sb, &be->d_out_feat, &be->d_raw_pose, &be->d_raw_sb,
                          &be->d_raw_feat, &be->d_out_loop, &be->d_stats_d, &be->d_m_x0, &be->d_m_J, &be->d_m_r,
                          &be->d_m_scratch};
  for (auto *b : db) b->release();
  be->d_prof.release();
  for (int k = 0; k < 2; k++) be->d_st_x0[k].release(), be->d_st_J[k].release(), be->d_st_r[k].release();
  be->d_ptab.release();
  resident_destroy(be->res);
  (void)hipStreamDestroy(be->stream);
  delete be;
}

int vio_backend_reserve_priors(vio_backend_t *be, int32_t n_slots) {
  if (!be || n_slots < 0) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(be);
  HIP_OK(hipStreamSynchronize(be->last_stream ? be->last_stream : be->stream));
  be->uploaded = false;
  be->st_n = 0;
  const int ncap = 6 * be->cfg.window_size + 15;  // what marginalize() can leave behind: W poses, one speed-bias, the extrinsic
  for (int k = 0; k < 2; k++) {
    int rc = be->d_st_x0[k].ensure((size_t)n_slots * 9 * kMaxPriorBlocks);
    if (rc == VIO_OK) rc = be->d_st_J[k].ensure((size_t)n_slots * ncap * ncap);
    if (rc == VIO_OK) rc = be->d_st_r[k].ensure((size_t)n_slots * ncap);
    if (rc != VIO_OK) return rc;
  }
  be->st_n = n_slots, be->st_ncap = ncap;
  be->st_bank.assign(n_slots, 0);
  return VIO_OK;
}

// LDS layout and launch split of a batch whose dims are d: which windows run the LDS variant (be->n_lds, be->d_lds,
// be->lds_bytes), which the global-matrix one (be->n_glb, be->d_glb), in which order (`order`: launch-local block ->
// window). nfeat[b]: landmarks of window b; Fmax: their maximum.
static int plan_layout(vio_backend *be, BatchDims &d, int n, const int *nfeat, int Fmax, std::vector<int> &order,
                       const std::vector<int> *active = nullptr) {  // active: the windows that take part (null: all n)
  d.Flds = std::max(Fmax, 1);
  d.prof_stages = be->profile ? ST_COUNT : 0;  // (the stage clock's accumulators live in LDS: only the profiling launches pay for them)
  // LDS or global pose matrix, per window. The LDS variant keeps the fill tiles of the speed-bias band in registers (pose
  // matrices of at most kPanelTiles tile rows: W <= 12) and needs both phases inside the CU's LDS. Eligible windows are
  // ordered by landmark count; the largest prefix whose layout (carved for its own landmark maximum) fits runs the LDS
  // variant in one launch -- with half a CU's LDS per workgroup whenever that is enough, so that two windows share a
  // CU --, everything else the global-matrix variant in a second one. (Decided before packing: the staging chunk the
  // buckets are aligned to depends on the layout.)
  static const int threads_lds = (getenv("VIO_AMD_WINDOW_THREADS") && atoi(getenv("VIO_AMD_WINDOW_THREADS")) == 512) ? kThreadsGlb : kThreadsLds;
  be->threads_lds = threads_lds;
  auto need_lds = [&](BatchDims dd, int asp) {
    dd.lds_asp = asp;
    size_t se = 0, tail = 0;
    const size_t bs = carve_work<ldsd>(dd, true, threads_lds, nullptr, nullptr, nullptr, nullptr, &se, &tail);
    const size_t bm = (se + tail) * sizeof(double) + carve_marg<ldsd>(dd, true, nullptr, nullptr, nullptr, 0);
    // the marginalization phase additionally wants >= 64 staging slots behind its matrix
    return std::max(bs, bm + 64 * kMargSlot * sizeof(double));
  };
  // (eligibility: the lean layout, IMU coupling in global memory, inside one CU; the 512-thread build of the LDS variant
  // -- an experiment switch -- only exists with the coupling in LDS)
  const int lean_asp = threads_lds == kThreadsLds ? 0 : 1;
  auto fits_lds = [&](const BatchDims &dd) { return pose_jp(dd) <= 16 * kPanelTiles && need_lds(dd, lean_asp) <= kLdsLimit; };
  std::vector<int> cand;
  order.clear();
  if (active) cand = *active;
  else
    for (int b = 0; b < n; b++) cand.push_back(b);
  const std::vector<int> all = cand;
  std::stable_sort(cand.begin(), cand.end(), [&](int a, int b2) { return nfeat[a] < nfeat[b2]; });
  BatchDims dl = d;
  int n_lds = (int)cand.size();
  while (n_lds > 0) {
    dl.Flds = std::max(1, nfeat[cand[n_lds - 1]]);
    if (fits_lds(dl)) break;
    n_lds--;
  }
  std::vector<char> in_lds(n, 0);
  for (int i = 0; i < n_lds; i++) in_lds[cand[i]] = 1, order.push_back(cand[i]);
  for (int b : all)
    if (!in_lds[b]) order.push_back(b);
  be->n_lds = n_lds, be->n_glb = (int)all.size() - n_lds, be->lds_matrix = be->n_glb == 0;
  // Two workgroups per CU (half its LDS each) beat everything else; inside that, the IMU speed-bias x pose coupling is
  // better off in LDS. Windows with many landmarks give its 14 KB up (global scratch, L2-resident) to stay two per CU.
  // The marginalization phase stages Jacobian rows in whatever LDS is left behind its matrix.
  static const bool one_per_cu = getenv("VIO_AMD_WINDOW_ONE_PER_CU") && getenv("VIO_AMD_WINDOW_ONE_PER_CU")[0] == '1';
  static const int force_asp = getenv("VIO_AMD_LDS_ASP") ? atoi(getenv("VIO_AMD_LDS_ASP")) : -1;
  be->lds_bytes = kLdsLimit;
  if (n_lds > 0) {
    const size_t fat = need_lds(dl, 1), lean = need_lds(dl, lean_asp);
    // (the lean layout costs a window ~10 % of its latency -- a few more round trips to L2 per Gauss-Newton step --: it only
    // pays when the launch has enough windows to fill the second workgroup slot of the CUs. Measured with closed-loop
    // windows of ~190 landmarks: 2 x 128 windows are faster with the fat layout on whole CUs, 2 x 256 windows take 7.3
    // instead of 9.4 ms per frame with the lean one.)
    const bool crowded = n_lds * std::max(1, be->peers) > be->n_cus;  // (with the other contexts' launches: more windows than CUs)
    if (!one_per_cu && fat <= kLdsHalf) dl.lds_asp = 1, be->lds_bytes = kLdsHalf;
    else if (!one_per_cu && crowded && lean <= kLdsHalf) dl.lds_asp = lean_asp, be->lds_bytes = kLdsHalf;
    else dl.lds_asp = fat <= kLdsLimit ? 1 : lean_asp;
    if ((force_asp == 0 && lean_asp == 0) || (force_asp == 1 && fat <= kLdsLimit)) {
      dl.lds_asp = force_asp;
      be->lds_bytes = (!one_per_cu && need_lds(dl, force_asp) <= kLdsHalf) ? kLdsHalf : kLdsLimit;
    }
  }
  be->d_lds = dl;
  if (be->n_glb > 0) {
    BatchDims dg = d;
    int fg = 1;
    for (size_t i = n_lds; i < order.size(); i++) fg = std::max(fg, nfeat[order[i]]);
    dg.Flds = fg;
    size_t se = 0, tail = 0;
    const size_t bs = carve_work<double *>(dg, false, kThreadsGlb, nullptr, nullptr, nullptr, nullptr, &se, &tail);
    const size_t bm = (se + tail) * sizeof(double) + carve_marg<double *>(dg, false, nullptr, nullptr, nullptr, 0);
    // (the marginalization phase stages its Jacobian rows in LDS in this variant too: at least 64 slots behind its vectors)
    const size_t bmm = bm + 64 * kMargSlot * sizeof(double);
    if (std::max(bs, bmm) > kLdsLimit) return VIO_ECAP;
    be->d_glb = dg, be->lds_bytes_glb = std::max(bs, bmm);
  }
  return VIO_OK;
}

// May the LDS launch of this batch take the instantiations with the window size at compile time (vio_window_variants.h)? Every
// window has W = kStaticW frames, no window carries a relocalization pose (one more 6-dof block: other strides, another LDS
// layout) and the prior capacity is what marginalize() can leave at that size. VIO_AMD_STATIC_W=0 keeps the run-time variants (A/B).
static bool static_launch_ok(const vio_backend *be, const BatchDims &d, bool all_windows_static_w) {
  static const bool off = getenv("VIO_AMD_STATIC_W") && getenv("VIO_AMD_STATIC_W")[0] == '0';
  constexpr int W = vio_wk::kStaticW;
  return !off && all_windows_static_w && be->threads_lds == kThreadsLds && d.Wcap == W && d.Pcap == W + 1 && d.nblk_cap == W + 1 &&
         d.n6cap == 6 * (W + 2) && d.Ncap == 6 * W + 15 && d.pair_cap == (W + 2) * (W + 3) / 2;
}

// Work and output buffers of a batch of N windows with dims d (sticky: they only grow).
static int ensure_work_buffers(vio_backend *be, const BatchDims &d, const BatchStrides &s, size_t N) {
  const size_t m_ints = 4 + 3 * kMaxPriorBlocks, m_scr = be->lds_matrix ? 0 : marg_scratch_doubles(d.Wcap);
#define ENSURE(buf, count)                 \
  do {                                     \
    int rc_ = (buf).ensure(count);         \
    if (rc_ != VIO_OK) return rc_;         \
  } while (0)
  ENSURE(be->d_scratch, N * s.scratch);
  ENSURE(be->d_hm, be->lds_matrix ? 1 : N * s.hm);  // (indexed by window: sized for the whole batch when any window needs it)
  ENSURE(be->d_out_pose, N * s.out_pose);
  ENSURE(be->d_out_sb, N * s.out_sb);
  ENSURE(be->d_out_feat, N * s.out_feat);
  ENSURE(be->d_raw_pose, N * s.out_pose);
  ENSURE(be->d_raw_sb, N * s.out_sb);
  ENSURE(be->d_raw_feat, N * s.out_feat);
  ENSURE(be->d_out_loop, N * s.out_loop);
  ENSURE(be->d_stats_d, N * s.stats_d);
  ENSURE(be->d_stats_i, N * s.stats_i);
  ENSURE(be->d_m_ints, N * m_ints);
  ENSURE(be->d_m_x0, N * 9 * kMaxPriorBlocks);
  ENSURE(be->d_m_J, N * (size_t)d.Ncap * d.Ncap);
  ENSURE(be->d_m_r, N * (size_t)d.Ncap);
  ENSURE(be->d_m_scratch, m_scr ? N * m_scr : 1);
#undef ENSURE
  return VIO_OK;
}

// The work / output side of be->B and be->MP (the input arrays and B.ptab are bound by the caller).
static int bind_work_buffers(vio_backend *be, const BatchDims &d, const BatchStrides &s, size_t N) {
  (void)s;
  const size_t m_ints = 4 + 3 * kMaxPriorBlocks, m_scr = be->lds_matrix ? 0 : marg_scratch_doubles(d.Wcap);
  BatchPtrs &B = be->B;
  B.scratch = be->d_scratch.p, B.hm = be->d_hm.p, B.order = nullptr, B.coop = 1, B.n_launch = 0;
  B.coop_spin = vio::kCoopSpinLimit, B.coop_fault = 0;
  B.out_pose = be->d_out_pose.p, B.out_sb = be->d_out_sb.p, B.out_feat = be->d_out_feat.p;
  B.raw_pose = be->d_raw_pose.p, B.raw_sb = be->d_raw_sb.p, B.raw_feat = be->d_raw_feat.p;
  B.out_loop = be->d_out_loop.p, B.stats_d = be->d_stats_d.p, B.stats_i = be->d_stats_i.p;
  MargPtrs &MP = be->MP;
  MP.ints = be->d_m_ints.p, MP.x0 = be->d_m_x0.p, MP.J = be->d_m_J.p, MP.r = be->d_m_r.p;
  MP.scratch = m_scr ? be->d_m_scratch.p : nullptr;
  MP.s_ints = m_ints, MP.s_x0 = 9 * kMaxPriorBlocks, MP.s_J = (size_t)d.Ncap * d.Ncap, MP.s_r = d.Ncap;
  MP.s_scratch = m_scr;
  MP.prof = nullptr;
  MP.prof_tid = getenv("VIO_AMD_PROF_TID") ? atoi(getenv("VIO_AMD_PROF_TID")) : 0;
  MP.wrot = getenv("VIO_AMD_WAVE_ROT") ? atoi(getenv("VIO_AMD_WAVE_ROT")) : -1;
  if (be->profile) {
    int rcp = be->d_prof.ensure(N * ST_COUNT);
    if (rcp != VIO_OK) return rcp;
    MP.prof = be->d_prof.p;
  }
  return VIO_OK;
}

static int backend_upload_impl(vio_backend_t *be, const VioWindow *windows, int32_t n);

int vio_backend_upload(vio_backend_t *be, const VioWindow *windows, int32_t n) {
  if (!be || !windows || n < 1) return VIO_EINVAL;
  if (n > be->max_batch) return VIO_ECAP;
  VIO_ON_DEVICE_OF(be);
  be->uploaded = false;  // a failed upload leaves nothing to launch or download
  try {  // page-locked staging vectors and the per-window packers allocate: no exception crosses the ABI
    return backend_upload_impl(be, windows, n);
  } catch (const std::bad_alloc &) {
    return VIO_ENOMEM;
  }
}

static int backend_upload_impl(vio_backend_t *be, const VioWindow *windows, int32_t n) {
  be->coop_ok = false;  // (set again below once every global-matrix window is known to be packed for a cooperative launch)
  // an earlier launch may still be reading the device inputs this upload overwrites (it may have gone to a caller's
  // stream): wait for it first
  if (be->last_stream && be->last_stream != be->stream) HIP_OK(hipStreamSynchronize(be->last_stream));
  // device-resident prior chain: which windows name a slot, and whether any prior data crosses the host at all
  be->slot_of.assign(n, -1);
  be->host_prior_in = false, be->host_prior_out = false;
  bool any_slot = false;
  {
    std::vector<char> taken(be->st_n, 0);
    for (int b = 0; b < n; b++) {
      const VioWindow &w = windows[b];
      if (w.resident_prior < 0 || w.resident_prior > be->st_n) return VIO_EINVAL;
      const int slot = w.resident_prior - 1;
      if (slot >= 0) {
        if (taken[slot]) return VIO_EINVAL;
        taken[slot] = 1, any_slot = true;
        if (w.prior && w.prior->n > be->st_ncap) return VIO_ECAP;
      }
      be->slot_of[b] = slot;
      if (w.prior && w.prior->n > 0 && w.prior->linearized_jacobians) be->host_prior_in = true;
      if (w.next_prior && slot < 0) be->host_prior_out = true;
    }
  }
  int Wmax = 1, Fmax = 1, Mmax = 1, Nmax = 0;
  bool any_loop = false;
  for (int b = 0; b < n; b++) {
    const VioWindow &w = windows[b];
    if (w.window_size < 1 || w.n_features < 0 || w.n_factors < 0) return VIO_EINVAL;
    if (w.window_size > be->cfg.window_size || w.n_features > be->cfg.max_features ||
        w.n_factors > be->cfg.max_factors)
      return VIO_ECAP;
    Wmax = std::max(Wmax, w.window_size), Fmax = std::max(Fmax, w.n_features), Mmax = std::max(Mmax, w.n_factors);
    if (w.prior) Nmax = std::max(Nmax, w.prior->n);
    if (w.factor_target)
      for (int k = 0; k < w.n_factors; k++)
        if (w.factor_target[k] == w.window_size + 1) any_loop = true;
  }
  // Capacities (global strides, allocations) are sticky while the batch keeps its size and grow in generous steps: a
  // batch that fits the previous layout re-stages and reallocates nothing (hipFree synchronizes the device and
  // page-locked reallocation costs milliseconds, which stalls every other context on the GPU). The LDS layout is carved
  // for the exact landmark maximum of THIS batch (d.Flds): every landmark costs LDS.
  BatchDims d;
  const BatchDims &pd = be->hb.d;
  if (be->hb.sized && n == be->hb.n && Wmax == pd.Wcap && Fmax <= pd.Fcap && Mmax <= pd.Mcap && Nmax <= pd.Ncap &&
      (!any_loop || pd.nblk_cap == pd.Pcap + 1)) {
    d = pd;
  } else {
    // headroom (25 % landmarks, 12.5 % factors: every factor slot is ~60 bytes of upload per window), then rounded: landmark and factor counts of a running batch drift by a few percent from frame to
    // frame, and every growth is a round of page-locked and device reallocations (tens of milliseconds)
    const int Fr = std::min(be->cfg.max_features, (Fmax + Fmax / 4 + 63) / 64 * 64),
              Mr = std::min(be->cfg.max_factors, (Mmax + Mmax / 8 + 127) / 128 * 128);
    d = make_dims(be->cfg, Wmax, std::max(Fr, Fmax), std::max(Mr, Mmax), any_loop);
    d.Ncap = std::max(6 * Wmax + 15, Nmax);
  }
  std::vector<int> order;
  {
    std::vector<int> nfeat(n);
    for (int b = 0; b < n; b++) nfeat[b] = windows[b].n_features;
    const int rcl = plan_layout(be, d, n, nfeat.data(), Fmax, order);
    if (rcl != VIO_OK) return rcl;
  }
  {
    bool all_w = true;
    for (int b = 0; b < n; b++) all_w = all_w && windows[b].window_size == vio_wk::kStaticW;
    be->static_w = static_launch_ok(be, d, all_w);
  }
  const int threads_lds = be->threads_lds, n_lds = be->n_lds;
  const BatchDims dl = be->d_lds;
  (void)n_lds;
  static const bool poison_staging = getenv("VIO_AMD_POISON") && getenv("VIO_AMD_POISON")[0] == '1';
  // the previous upload's copy may still be reading the staging arena (uploads do not wait for their own transfer)
  HIP_OK(hipStreamSynchronize(be->stream));
  const double t0 = now_ms();
  be->hb.resize(d, n, poison_staging);
  {
    std::vector<int> rcs(n, VIO_OK);
    const bool lds_shape = pose_jp(d) <= 16 * kPanelTiles;
    const int chunk = stage_chunk_slots(dl, lds_shape, lds_shape ? threads_lds : kThreadsGlb);
    // (windows of the global-matrix launch have their own layout and chunk; a cooperative launch needs every bucket inside a chunk)
    const int chunk_glb = be->n_glb > 0 ? stage_chunk_slots(be->d_glb, false, kThreadsGlb) : 0;
    std::vector<char> glb(n, 0), strad(n, 0);
    for (size_t i = (size_t)n_lds; i < order.size(); i++) glb[order[i]] = 1;
    vio::HostPool::get().parallel_for(n, [&](int b) {
      try {
        bool st_ = false;
        rcs[b] = pack_window(be->hb, b, windows[b], be->slot_of[b] >= 0, glb[b] ? chunk_glb : chunk, &st_);
        strad[b] = st_ ? 1 : 0;
      } catch (const std::bad_alloc &) {  // (worker thread: must not unwind out of the pool)
        rcs[b] = VIO_ENOMEM;
      }
    });
    for (int b = 0; b < n; b++)
      if (rcs[b] != VIO_OK) return rcs[b];
    be->coop_ok = be->n_glb > 0;
    for (int b = 0; b < n; b++)
      if (glb[b] && strad[b]) be->coop_ok = false;
  }
  const double t1 = now_ms();
  const BatchStrides &s = be->hb.s;
  if (be->d_order.ensure(n) != VIO_OK) return VIO_ENOMEM;
  be->h_order.assign(order.begin(), order.end());  // page-locked: goes up with the other staging copies below

  const size_t N = (size_t)n;
  {
    int rce = be->d_in.ensure(be->hb.total_bytes);
    if (rce == VIO_OK) rce = ensure_work_buffers(be, d, s, N);
    if (rce != VIO_OK) return rce;
  }
  hipStream_t st = be->stream;
#define H2D(dst, src) HIP_OK(hipMemcpyAsync((dst).p, (src).data(), (src).size() * sizeof((src)[0]), hipMemcpyHostToDevice, st))
  H2D(be->d_order, be->h_order);
  // one copy for every input array; the prior data ([in_bytes, total_bytes): the bulk, ~45 KB per window at W=10) only
  // when some prior of this upload travels through the host
  HIP_OK(hipMemcpyAsync(be->d_in.p, be->hb.arena.data(), be->host_prior_in ? be->hb.total_bytes : be->hb.in_bytes,
                        hipMemcpyHostToDevice, st));
  if (any_slot) {
    if (be->d_ptab.ensure(n) != VIO_OK) return VIO_ENOMEM;
    be->h_ptab.resize(n);
    const size_t sx = 9 * kMaxPriorBlocks, sJ = (size_t)be->st_ncap * be->st_ncap, sr = be->st_ncap;
    for (int b = 0; b < n; b++) {
      PriorTab t = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
      const int k = be->slot_of[b];
      if (k >= 0) {
        const int cur = be->st_bank[k], nxt = 1 - cur;
        const VioPrior *p = windows[b].prior;
        if (p && p->n > 0 && !p->linearized_jacobians)
          t.x0 = be->d_st_x0[cur].p + k * sx, t.J = be->d_st_J[cur].p + k * sJ, t.r = be->d_st_r[cur].p + k * sr;
        t.mx0 = be->d_st_x0[nxt].p + k * sx, t.mJ = be->d_st_J[nxt].p + k * sJ, t.mr = be->d_st_r[nxt].p + k * sr;
        t.ncap = std::min(be->st_ncap, d.Ncap);
      }
      be->h_ptab[b] = t;
    }
    H2D(be->d_ptab, be->h_ptab);
  }
#undef H2D
  if (!be->upload_done) HIP_OK(hipEventCreateWithFlags(&be->upload_done, hipEventDisableTiming));
  HIP_OK(hipEventRecord(be->upload_done, st));
  // (no wait here: the launch follows on the same stream, and the staging arena is not touched again before the wait at
  // the top of the next upload)
  if (host_timing()) fprintf(stderr, "vio_backend_upload: pack %.2f ms, alloc+H2D calls %.2f ms (n=%d)\n", t1 - t0, now_ms() - t1, n);

  BatchPtrs &B = be->B;
  B.n = n, B.d = d, B.s = s;
  {
    const size_t *o = be->hb.off;
    unsigned char *base = be->d_in.p;
    auto ip = [&](int k) { return reinterpret_cast<int *>(base + o[k]); };
    auto dp = [&](int k) { return reinterpret_cast<double *>(base + o[k]); };
    B.hdr = ip(0), B.fhost = ip(1), B.ftarget = ip(2), B.ffeat = ip(3), B.fslot = ip(4), B.fstart = ip(5);
    B.pair_h = ip(6), B.pair_t = ip(7), B.pair_s0 = ip(8), B.pair_s1 = ip(9);
    B.pr_kind = ip(10), B.pr_index = ip(11), B.pr_offset = ip(12);
    B.hdr_d = dp(13), B.pose = dp(14), B.sb = dp(15), B.ex = dp(16), B.feat = dp(17), B.pts_i = dp(18), B.pts_j = dp(19);
    B.preint = dp(20), B.pr_x0 = dp(21), B.pr_J = dp(22), B.pr_r = dp(23);
  }
  B.ptab = any_slot ? be->d_ptab.p : nullptr;
  {
    const int rcb = bind_work_buffers(be, d, s, N);
    if (rcb != VIO_OK) return rcb;
  }
  be->n = n;
  be->uploaded = true, be->slots_advanced = false;
  return VIO_OK;
}

int vio_backend_launch(vio_backend_t *be, void *stream) {
  if (!be) return VIO_EINVAL;
  if (!be->uploaded) return VIO_ESTATE;
  VIO_ON_DEVICE_OF(be);
  hipStream_t st = stream ? (hipStream_t)stream : be->stream;
  if (st != be->stream && be->upload_done) HIP_OK(hipStreamWaitEvent(st, be->upload_done, 0));
  be->last_stream = st;
  if (be->events_used == be->events.size()) {
    if (be->events.size() >= 4096) {  // recycle: fold what is pending into nothing (caller did not ask for it)
      be->events_used = 0;
    } else {
      hipEvent_t a, b;
      HIP_OK(hipEventCreate(&a));
      HIP_OK(hipEventCreate(&b));
      be->events.push_back({a, b});
    }
  }
  auto &ev = be->events[be->events_used++];
  static const bool poison = getenv("VIO_AMD_POISON") && getenv("VIO_AMD_POISON")[0] == '1';
  if (poison) {
    HIP_OK(hipMemsetAsync(be->d_scratch.p, 0xff, be->d_scratch.n * sizeof(double), st));
    HIP_OK(hipMemsetAsync(be->d_hm.p, 0xff, be->d_hm.n * sizeof(double), st));
    HIP_OK(hipMemsetAsync(be->d_out_pose.p, 0xff, be->d_out_pose.n * sizeof(double), st));
    HIP_OK(hipMemsetAsync(be->d_stats_d.p, 0xff, be->d_stats_d.n * sizeof(double), st));
    HIP_OK(hipFuncSetAttribute((const void *)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(1024), kLdsLimit, st, (int)(kLdsLimit / sizeof(double)));
  }
  HIP_OK(hipEventRecord(ev.first, st));
  if (be->n_lds > 0) {
    BatchPtrs Bl = be->B;
    Bl.d = be->d_lds, Bl.order = be->d_order.p;
    int variant = be->threads_lds == kThreadsLds ? (be->d_lds.lds_asp ? 0 : 1) : 2;
    if (variant < 2 && be->static_w) variant += 4;  // (every window of the launch has the compile-time window size: vio_window_variants.h)
    vio_wk::variant(variant, be->MP.prof != nullptr).launch(be->n_lds, be->lds_bytes, st, Bl, be->MP);
  }
  if (be->n_glb > 0) {
    BatchPtrs Bg = be->B;
    Bg.d = be->d_glb, Bg.order = be->d_order.p + be->n_lds;
    // Cooperative windows (solver_core.h): few large windows leave most CUs idle -- a window gets 2 or 4 workgroups when all of
    // them fit the chip at once (one per CU: this variant's workgroup takes more than half a CU's LDS). VIO_AMD_COOP = 1 / 2 / 4
    // forces the width (1: off). The profiling clock follows one workgroup per window: off while it runs.
    int coop = 1;
    if (be->coop_ok && !be->MP.prof && be->lds_bytes_glb > kLdsHalf) {
      const char *fe_ = getenv("VIO_AMD_COOP");  // (read per launch: tests switch it within one process)
      const int forced = fe_ ? atoi(fe_) : 0;
      const int cus = be->n_cus / std::max(1, be->peers);
      const int groups = (be->n_glb + 7) / 8;  // (grids are padded to whole groups of eight windows: the XCD mapping)
      // (measured, profiles/r05_*_large_windows.txt: four members pay at W = 30 up to a full chip, at W = 20 only while half the
      // CUs stay free -- with every CU busy the shared phases of 64 windows hit the memory system together)
      const bool four = groups * 8 * 4 <= cus && (be->d_glb.Wcap >= 24 || groups * 8 * 8 <= cus);
      coop = four ? 4 : groups * 8 * 2 <= cus ? 2 : 1;
      if (forced >= 1 && forced <= vio::kCoopMax && groups * 8 * forced <= cus) coop = forced;  // (a forced width obeys the peers bound too)
      if (coop > 1) {
        // every workgroup of the launch has to be resident at once (the members of a window wait for each other): ask the runtime
        // what the device holds of this kernel at this LDS size instead of trusting the arithmetic above alone
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vio_wk::variant(3, false).fn,
                                                         kThreadsGlb, be->lds_bytes_glb) != hipSuccess ||
            (long long)per_cu * cus < (long long)groups * 8 * coop)
          coop = 1;
      }
    }
    Bg.coop = coop, Bg.n_launch = be->n_glb;
    {  // test hooks of the timeout path (tests/test_backend_gpu.py): read per launch
      const char *sp_ = getenv("VIO_AMD_COOP_SPIN"), *ft_ = getenv("VIO_AMD_COOP_FAULT");
      Bg.coop_spin = sp_ && atoi(sp_) > 0 ? (unsigned)atoi(sp_) : vio::kCoopSpinLimit;
      Bg.coop_fault = ft_ && ft_[0] == '1' ? 1 : 0;
    }
    int grid = be->n_glb;
    if (coop > 1) {
      grid = (be->n_glb + 7) / 8 * 8 * coop;
      // flag words of every window of the batch: command, completions, error (the first 32 bytes of the cooperative area)
      HIP_OK(hipMemset2DAsync(be->d_scratch.p + be->B.s.s_coop, be->B.s.scratch * sizeof(double), 0, 32, (size_t)be->B.n, st));
    }
    vio_wk::variant(3, be->MP.prof != nullptr).launch(grid, be->lds_bytes_glb, st, Bg, be->MP);
  }
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(ev.second, st));
  return VIO_OK;
}

int vio_backend_sync(vio_backend_t *be) {
  if (!be) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(be);
  HIP_OK(hipDeviceSynchronize());
  return VIO_OK;
}

int vio_backend_kernel_ms(vio_backend_t *be, double *ms_avg, int32_t *launches) {
  if (!be || !ms_avg || !launches) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(be);
  HIP_OK(hipDeviceSynchronize());
  double sum = 0;
  for (size_t i = 0; i < be->events_used; i++) {
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, be->events[i].first, be->events[i].second));
    sum += ms;
  }
  *launches = (int32_t)be->events_used;
  *ms_avg = be->events_used ? sum / be->events_used : 0.0;
  be->events_used = 0;
  return VIO_OK;
}

int vio_backend_set_profile(vio_backend_t *be, int32_t enable) {
  if (!be) return VIO_EINVAL;
  be->profile = enable != 0;
  be->uploaded = false;  // takes effect at the next upload
  return VIO_OK;
}

int vio_backend_stage_cycles(vio_backend_t *be, int32_t window, int64_t *cycles, int32_t n_stages) {
  if (!be || !cycles || n_stages < 1) return VIO_EINVAL;
  if (!be->uploaded || !be->profile || window < 0 || window >= be->n) return VIO_ESTATE;
  VIO_ON_DEVICE_OF(be);
  HIP_OK(hipDeviceSynchronize());
  long long tmp[ST_COUNT];
  HIP_OK(hipMemcpy(tmp, be->d_prof.p + (size_t)window * ST_COUNT, sizeof(tmp), hipMemcpyDeviceToHost));
  for (int i = 0; i < n_stages; i++) cycles[i] = i < ST_COUNT ? (int64_t)tmp[i] : 0;
  return VIO_OK;
}

static int backend_download_impl(vio_backend_t *be, VioWindow *windows, int32_t n, VioSolveStats *stats);

int vio_backend_download(vio_backend_t *be, VioWindow *windows, int32_t n, VioSolveStats *stats) {
  if (!be || !windows) return VIO_EINVAL;
  if (!be->uploaded || n != be->n) return VIO_ESTATE;
  VIO_ON_DEVICE_OF(be);
  try {  // the page-locked host copies are (re)sized here
    return backend_download_impl(be, windows, n, stats);
  } catch (const std::bad_alloc &) {
    return VIO_ENOMEM;
  }
}

static int backend_download_impl(vio_backend_t *be, VioWindow *windows, int32_t n, VioSolveStats *stats) {
  const double t0 = now_ms();
  // only this context's launch: other contexts (other host threads) keep the device busy meanwhile
  HIP_OK(hipStreamSynchronize(be->last_stream ? be->last_stream : be->stream));
  const double t1 = now_ms();
  hipStream_t st = be->stream;
#define D2H(dst, src)                                                                                          \
  do {                                                                                                         \
    if ((dst).size() != (src).n) (dst).resize((src).n);                                                        \
    HIP_OK(hipMemcpyAsync((dst).data(), (src).p, (src).n * sizeof((dst)[0]), hipMemcpyDeviceToHost, st));      \
  } while (0)
  D2H(be->h_out_pose, be->d_out_pose);
  D2H(be->h_out_sb, be->d_out_sb);
  D2H(be->h_out_feat, be->d_out_feat);
  D2H(be->h_raw_pose, be->d_raw_pose);
  D2H(be->h_raw_sb, be->d_raw_sb);
  D2H(be->h_raw_feat, be->d_raw_feat);
  D2H(be->h_out_loop, be->d_out_loop);
  D2H(be->h_stats_d, be->d_stats_d);
  D2H(be->h_stats_i, be->d_stats_i);
  D2H(be->h_m_ints, be->d_m_ints);
  if (be->host_prior_out) {
    D2H(be->h_m_x0, be->d_m_x0);
    D2H(be->h_m_J, be->d_m_J);
    D2H(be->h_m_r, be->d_m_r);
  }
#undef D2H
  HIP_OK(hipStreamSynchronize(st));
  const double t2 = now_ms();
  const BatchStrides &s = be->hb.s;
  vio::HostPool::get().parallel_for(n, [&](int b) {
    unpack_window(s, b, be->h_out_pose.data(), be->h_out_sb.data(), be->h_out_feat.data(), be->h_raw_pose.data(),
                  be->h_raw_sb.data(), be->h_raw_feat.data(), be->h_out_loop.data(), be->h_stats_d.data(),
                  be->h_stats_i.data(), windows[b], stats ? stats + b : nullptr);
    if (windows[b].next_prior) {
      MargOut mo;
      int *mi = be->h_m_ints.data() + (size_t)b * be->MP.s_ints;
      mo.n = mi, mo.kind = mi + 4, mo.index = mo.kind + kMaxPriorBlocks, mo.offset = mo.index + kMaxPriorBlocks;
      mo.x0 = nullptr, mo.J = nullptr, mo.r = nullptr;
      mo.scratch = nullptr, mo.ncap = be->B.d.Ncap;
      const int k = be->slot_of[b];
      if (k < 0) mo.x0 = be->h_m_x0.data() + (size_t)b * be->MP.s_x0, mo.J = be->h_m_J.data() + (size_t)b * be->MP.s_J,
                 mo.r = be->h_m_r.data() + (size_t)b * be->MP.s_r;
      unpack_prior(mo, *windows[b].next_prior, k < 0);
    }
  });
  // a cooperative window whose workgroups did not meet (solver_core.h, coop_spin): FAILURE, and the call says so
  bool timed_out = false;
  for (int b = 0; b < n; b++)
    if (be->h_stats_i[(size_t)b * s.stats_i + 1] == -9) {
      timed_out = true;
      if (stats) stats[b].termination = 2;
      if (windows[b].next_prior) windows[b].next_prior->n = 0, windows[b].next_prior->n_blocks = 0;
      be->h_m_ints[(size_t)b * be->MP.s_ints] = 0;  // (its slot keeps the previous prior)
    }
  // advance the slots whose window produced a new prior; once per upload (a second download of the same launch, or a
  // re-launch of the same upload, reads and writes the same banks again)
  if (!be->slots_advanced) {
    for (int b = 0; b < n; b++) {
      const int k = be->slot_of[b];
      if (k >= 0 && be->h_m_ints[(size_t)b * be->MP.s_ints] > 0) be->st_bank[k] ^= 1;
    }
    be->slots_advanced = true;
  }
  if (host_timing())
    fprintf(stderr, "vio_backend_download: wait for kernel %.2f ms, D2H %.2f ms, unpack %.2f ms\n", t1 - t0, t2 - t1, now_ms() - t2);
  return timed_out ? VIO_ETIMEOUT : VIO_OK;
}

int vio_backend_solve_windows(vio_backend_t *be, VioWindow *windows, int32_t n, int32_t buf_num, VioSolveStats *stats) {
  (void)buf_num;  // wall-clock budget selector in the reference (VINS.cpp:648-653); no time limit here
  const double t0 = now_ms();
  int rc = vio_backend_upload(be, windows, n);
  if (rc != VIO_OK) return rc;
  const double t1 = now_ms();
  rc = vio_backend_launch(be, nullptr);
  if (rc != VIO_OK) return rc;
  const double t2 = now_ms();
  rc = vio_backend_download(be, windows, n, stats);
  if (host_timing()) fprintf(stderr, "vio_backend_solve_windows: upload %.2f ms, launch call %.2f ms, download %.2f ms\n", t1 - t0, t2 - t1, now_ms() - t2);
  return rc;
}

}  // extern "C"

#include "vio_backend_resident.inc"
