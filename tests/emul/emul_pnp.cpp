// tests/emul/emul_pnp.cpp — TEST-ONLY host emulation of the PnP window kernel: compiles vins-mobile_amd/csrc/pnp_core.h
// with -DVIO_EMUL (one "thread", no barriers). Not part of the product library, not a CPU fallback.
#include <string.h>

#include <limits>
#include <vector>

#include "vio_amd.h"
#include "pnp_core.h"

using namespace vio;

extern "C" int emul_pnp_solve(const VioConfig *cfg, VioPnpWindow *win, VioSolveStats *stats) {
  const int n = win->n_frames, M = win->feat_start[n];
  const double kNaN = std::numeric_limits<double>::quiet_NaN();
  std::vector<int> fixed(n), si(kStatsInts, 0);
  for (int k = 0; k < n; k++) fixed[k] = win->fixed[k] ? 1 : 0;
  std::vector<double> out_pose(7 * n, kNaN), out_speed(3 * n, kNaN), sd(kStatsDoubles, 0.0), U(225 * (n - 1), kNaN), Jraw(450 * (n - 1), kNaN);
  pnp::View v;
  v.n = n, v.M = M, v.max_iter = 5;
  v.fixed = fixed.data(), v.feat_start = win->feat_start;
  v.pose0 = win->pose, v.speed0 = win->speed, v.bias = win->bias, v.ex = win->ex_pose;
  v.preint = reinterpret_cast<const double *>(win->preint);
  v.obs = win->observation, v.pos = win->position, v.track = win->track_num;
  v.out_pose = out_pose.data(), v.out_speed = out_speed.data(), v.stats_d = sd.data(), v.stats_i = si.data();
  v.Jraw = Jraw.data();
  v.s_info = cfg->fx / 1.5, v.gravity = cfg->gravity, v.cauchy_b = cfg->cauchy_a * cfg->cauchy_a;
  Ctx cx;
  cx.tid = 0, cx.nt = 1, cx.prof = nullptr, cx.lprof = nullptr;
  std::vector<double> work(pnp::carve<double *>(n, 64, nullptr, nullptr, nullptr) / sizeof(double) + 8, kNaN);
  pnp::Work<double *> w;
  pnp::carve<double *>(n, 64, work.data(), &w, &cx);
  pnp::solve(cx, v, w);
  for (int i = 0; i < 7 * n; i++) win->pose[i] = out_pose[i];
  for (int i = 0; i < 3 * n; i++) win->speed[i] = out_speed[i];
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->initial_cost = sd[0], stats->final_cost = sd[1];
    stats->iterations = si[0], stats->termination = si[1], stats->num_successful_steps = si[2], stats->num_unsuccessful_steps = si[3];
    for (int k = 0; k < kMaxTrace && k < VIO_MAX_TRACE; k++) {
      stats->it_cost[k] = sd[4 + k], stats->it_radius[k] = sd[4 + kMaxTrace + k], stats->it_step_norm[k] = sd[4 + 2 * kMaxTrace + k];
      stats->it_relative_decrease[k] = sd[4 + 3 * kMaxTrace + k], stats->it_gradient_max_norm[k] = sd[4 + 4 * kMaxTrace + k];
      stats->it_flags[k] = si[4 + k];
    }
  }
  return VIO_OK;
}
