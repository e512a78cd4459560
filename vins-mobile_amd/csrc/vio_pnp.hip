// vio_pnp.hip — device launch of the motion-only window solve (pnp_core.h) for a batch of independent windows, and the
// host packing around it. Reference: vinsPnP::solve_ceres (VINS_ios/vins_pnp.cpp:264-341).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

#include "pnp_core.h"
#include "vio_amd.h"
#include "vio_device.h"

using namespace vio;

namespace {

constexpr int kThreads = 256;
constexpr int kStatsD = 4 + 5 * kMaxTrace, kStatsI = 4 + kMaxTrace;

struct PnpBatch {  // device pointers, per-window strides in elements
  int n_windows, max_frames, max_factors;
  const int *hdr;  // [N][4 + 2*max_frames + 1]: n, M, max_iter, -, fixed[max_frames], feat_start[max_frames+1]
  const double *pose, *speed, *bias, *ex, *preint, *obs, *pos;
  const int *track;
  double *out_pose, *out_speed, *stats_d, *U, *Jraw;
  int *stats_i;
  double s_info, gravity, cauchy_b;
};

__global__ __launch_bounds__(kThreads) void pnp_window_kernel(PnpBatch B) {
  extern __shared__ __attribute__((aligned(16))) double pnp_smem[];
  const int b = blockIdx.x, F = B.max_frames;
  const int *h = B.hdr + (size_t)b * (4 + 2 * F + 1);
  pnp::View v;
  v.n = h[0], v.M = h[1], v.max_iter = h[2];
  v.fixed = h + 4, v.feat_start = h + 4 + F;
  v.pose0 = B.pose + (size_t)b * 7 * F, v.speed0 = B.speed + (size_t)b * 3 * F, v.bias = B.bias + (size_t)b * 6 * F;
  v.ex = B.ex + (size_t)b * 7;
  v.preint = B.preint + (size_t)b * (F - 1) * pnp::kPreDoubles;
  v.obs = B.obs + (size_t)b * 2 * B.max_factors, v.pos = B.pos + (size_t)b * 3 * B.max_factors;
  v.track = B.track + (size_t)b * B.max_factors;
  v.out_pose = B.out_pose + (size_t)b * 7 * F, v.out_speed = B.out_speed + (size_t)b * 3 * F;
  v.stats_d = B.stats_d + (size_t)b * kStatsD, v.stats_i = B.stats_i + (size_t)b * kStatsI;
  v.Jraw = B.Jraw + (size_t)b * (F - 1) * 450;
  v.s_info = B.s_info, v.gravity = B.gravity, v.cauchy_b = B.cauchy_b;
  Ctx cx;
  cx.wave64 = __builtin_amdgcn_readfirstlane((int)threadIdx.x & ~63);
  cx.tid = threadIdx.x, cx.nt = blockDim.x, cx.prof = nullptr, cx.lprof = nullptr;
  pnp::Work<ldsd> w;
  pnp::carve<ldsd>(v.n, kThreads, (ldsd)pnp_smem, &w, &cx);
  pnp::solve(cx, v, w);
}

template <class T>
struct Dev {
  T *p = nullptr;
  size_t n = 0;
  bool ensure(size_t count) {
    if (count <= n && p) return true;
    if (p) (void)hipFree(p);
    p = nullptr, n = 0;
    if (hipMalloc(&p, (count ? count : 1) * sizeof(T)) != hipSuccess) return false;
    n = count;
    return true;
  }
  ~Dev() {
    if (p) (void)hipFree(p);
  }
};

}  // namespace

struct vio_pnp {
  int device = -1;  // HIP device the context lives on (current device at create)
  VioConfig cfg;
  int max_batch = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double ms_sum = 0;
  int launches = 0;
  Dev<int> d_hdr, d_track, d_stats_i;
  Dev<double> d_pose, d_speed, d_bias, d_ex, d_preint, d_obs, d_pos, d_out_pose, d_out_speed, d_stats_d, d_U, d_Jraw;
};

extern "C" {

int vio_pnp_create(const VioConfig *cfg, int32_t max_batch, vio_pnp_t **out) {
  if (!cfg || !out || max_batch < 1) return VIO_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    fprintf(stderr, "vio_amd: no HIP device visible; the PnP window solve has no CPU fallback\n");
    return VIO_ENODEV;
  }
  vio_pnp *p = new (std::nothrow) vio_pnp();
  if (p) p->device = vio::current_device();
  if (!p) return VIO_ENOMEM;
  p->cfg = *cfg, p->max_batch = max_batch;
  if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&p->ev0) != hipSuccess ||
      hipEventCreate(&p->ev1) != hipSuccess) {
    vio_pnp_destroy(p);
    return VIO_ENODEV;
  }
  *out = p;
  return VIO_OK;
}

void vio_pnp_destroy(vio_pnp_t *p) {
  if (!p) return;
  vio::DeviceScope scope(p->device);
  if (p->ev0) (void)hipEventDestroy(p->ev0);
  if (p->ev1) (void)hipEventDestroy(p->ev1);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

int vio_pnp_solve_windows(vio_pnp_t *p, VioPnpWindow *windows, int32_t n, VioSolveStats *stats) {
  if (!p || !windows || n < 1) return VIO_EINVAL;
  if (n > p->max_batch) return VIO_ECAP;
  VIO_ON_DEVICE_OF(p);
  int F = 2, Mmax = 1;
  for (int b = 0; b < n; b++) {
    const VioPnpWindow &w = windows[b];
    if (w.n_frames < 2 || w.n_frames > VIO_PNP_MAX_FRAMES || !w.pose || !w.speed || !w.bias || !w.fixed || !w.ex_pose ||
        !w.preint || !w.feat_start)
      return VIO_EINVAL;
    const int M = w.feat_start[w.n_frames];
    if (w.feat_start[0] != 0 || M < 0 || (M > 0 && (!w.observation || !w.position || !w.track_num))) return VIO_EINVAL;
    for (int k = 0; k < w.n_frames; k++)
      if (w.feat_start[k + 1] < w.feat_start[k]) return VIO_EINVAL;
    F = std::max(F, w.n_frames), Mmax = std::max(Mmax, M);
  }
  const size_t N = n, hs = 4 + 2 * (size_t)F + 1;
  std::vector<int> hdr(N * hs, 0), track(N * Mmax, 1);
  std::vector<double> pose(N * 7 * F, 0.0), speed(N * 3 * F, 0.0), bias(N * 6 * F, 0.0), ex(N * 7, 0.0),
      pre(N * (F - 1) * pnp::kPreDoubles, 0.0), obs(N * 2 * Mmax, 0.0), pos(N * 3 * Mmax, 1.0);
  static_assert(sizeof(VioPreintegration) == pnp::kPreDoubles * sizeof(double), "VioPreintegration layout");
  for (int b = 0; b < n; b++) {
    const VioPnpWindow &w = windows[b];
    const int nf = w.n_frames, M = w.feat_start[nf];
    int *h = &hdr[b * hs];
    h[0] = nf, h[1] = M, h[2] = 5;  // options.max_num_iterations = 5 (vins_pnp.cpp:324)
    for (int k = 0; k < nf; k++) h[4 + k] = w.fixed[k] ? 1 : 0;
    for (int k = 0; k <= nf; k++) h[4 + F + k] = w.feat_start[k];
    memcpy(&pose[(size_t)b * 7 * F], w.pose, sizeof(double) * 7 * nf);
    memcpy(&speed[(size_t)b * 3 * F], w.speed, sizeof(double) * 3 * nf);
    memcpy(&bias[(size_t)b * 6 * F], w.bias, sizeof(double) * 6 * nf);
    memcpy(&ex[(size_t)b * 7], w.ex_pose, sizeof(double) * 7);
    memcpy(&pre[(size_t)b * (F - 1) * pnp::kPreDoubles], w.preint, sizeof(VioPreintegration) * (nf - 1));
    if (M) {
      memcpy(&obs[(size_t)b * 2 * Mmax], w.observation, sizeof(double) * 2 * M);
      memcpy(&pos[(size_t)b * 3 * Mmax], w.position, sizeof(double) * 3 * M);
      memcpy(&track[(size_t)b * Mmax], w.track_num, sizeof(int) * M);
    }
  }
  bool ok = p->d_hdr.ensure(hdr.size()) && p->d_track.ensure(track.size()) && p->d_pose.ensure(pose.size()) &&
            p->d_speed.ensure(speed.size()) && p->d_bias.ensure(bias.size()) && p->d_ex.ensure(ex.size()) &&
            p->d_preint.ensure(pre.size()) && p->d_obs.ensure(obs.size()) && p->d_pos.ensure(pos.size()) &&
            p->d_out_pose.ensure(pose.size()) && p->d_out_speed.ensure(speed.size()) && p->d_stats_d.ensure(N * kStatsD) &&
            p->d_stats_i.ensure(N * kStatsI) && p->d_U.ensure(N * (F - 1) * 225) && p->d_Jraw.ensure(N * (F - 1) * 450);
  if (!ok) return VIO_ENOMEM;
  hipStream_t st = p->stream;
#define H2D(d, h) \
  if (hipMemcpyAsync((d).p, (h).data(), (h).size() * sizeof((h)[0]), hipMemcpyHostToDevice, st) != hipSuccess) return VIO_ENODEV
  H2D(p->d_hdr, hdr);
  H2D(p->d_track, track);
  H2D(p->d_pose, pose);
  H2D(p->d_speed, speed);
  H2D(p->d_bias, bias);
  H2D(p->d_ex, ex);
  H2D(p->d_preint, pre);
  H2D(p->d_obs, obs);
  H2D(p->d_pos, pos);
#undef H2D
  PnpBatch B;
  B.n_windows = n, B.max_frames = F, B.max_factors = Mmax;
  B.hdr = p->d_hdr.p, B.pose = p->d_pose.p, B.speed = p->d_speed.p, B.bias = p->d_bias.p, B.ex = p->d_ex.p;
  B.preint = p->d_preint.p, B.obs = p->d_obs.p, B.pos = p->d_pos.p, B.track = p->d_track.p;
  B.out_pose = p->d_out_pose.p, B.out_speed = p->d_out_speed.p, B.stats_d = p->d_stats_d.p, B.stats_i = p->d_stats_i.p;
  B.U = p->d_U.p, B.Jraw = p->d_Jraw.p;
  B.s_info = p->cfg.fx / 1.5;  // PerspectiveFactor::sqrt_info = FOCUS_LENGTH_X / 1.5 (vins_pnp.cpp:19)
  B.gravity = p->cfg.gravity, B.cauchy_b = p->cfg.cauchy_a * p->cfg.cauchy_a;
  const size_t lds = pnp::carve<ldsd>(F, kThreads, nullptr, nullptr, nullptr);
  if (hipFuncSetAttribute((const void *)pnp_window_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VIO_ENODEV;
  (void)hipEventRecord(p->ev0, st);
  hipLaunchKernelGGL(pnp_window_kernel, dim3(n), dim3(kThreads), lds, st, B);
  (void)hipEventRecord(p->ev1, st);
  if (hipGetLastError() != hipSuccess) return VIO_ENODEV;
  std::vector<double> o_pose(pose.size()), o_speed(speed.size()), sd(N * kStatsD);
  std::vector<int> si(N * kStatsI);
  if (hipMemcpyAsync(o_pose.data(), p->d_out_pose.p, o_pose.size() * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(o_speed.data(), p->d_out_speed.p, o_speed.size() * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(sd.data(), p->d_stats_d.p, sd.size() * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(si.data(), p->d_stats_i.p, si.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)
    return VIO_ENODEV;
  float ms = 0;
  if (hipEventElapsedTime(&ms, p->ev0, p->ev1) == hipSuccess) p->ms_sum += ms, p->launches++;
  for (int b = 0; b < n; b++) {
    VioPnpWindow &w = windows[b];
    memcpy(w.pose, &o_pose[(size_t)b * 7 * F], sizeof(double) * 7 * w.n_frames);
    memcpy(w.speed, &o_speed[(size_t)b * 3 * F], sizeof(double) * 3 * w.n_frames);
    if (stats) {
      VioSolveStats &s = stats[b];
      memset(&s, 0, sizeof(s));
      const double *d = &sd[(size_t)b * kStatsD];
      const int *i = &si[(size_t)b * kStatsI];
      s.initial_cost = d[0], s.final_cost = d[1];
      s.iterations = i[0], s.termination = i[1], s.num_successful_steps = i[2], s.num_unsuccessful_steps = i[3];
      for (int k = 0; k < kMaxTrace && k < VIO_MAX_TRACE; k++) {
        s.it_cost[k] = d[4 + k], s.it_radius[k] = d[4 + kMaxTrace + k], s.it_step_norm[k] = d[4 + 2 * kMaxTrace + k];
        s.it_relative_decrease[k] = d[4 + 3 * kMaxTrace + k], s.it_gradient_max_norm[k] = d[4 + 4 * kMaxTrace + k];
        s.it_flags[k] = i[4 + k];
      }
    }
  }
  return VIO_OK;
}

int vio_pnp_kernel_ms(vio_pnp_t *p, double *ms_avg, int32_t *launches) {
  if (!p || !ms_avg || !launches) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(p);
  *launches = p->launches, *ms_avg = p->launches ? p->ms_sum / p->launches : 0.0;
  p->ms_sum = 0, p->launches = 0;
  return VIO_OK;
}

}  // extern "C"
