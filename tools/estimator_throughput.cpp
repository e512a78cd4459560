// estimator_throughput — end-to-end rate of the batched estimator from host buffers, driven from C++ (no Python in the
// loop): E estimator objects of S sequences each, one host thread per object, replaying a data set written by
// tools/estimator_dataset.py. Build: g++ -O2 -std=c++17 -Iinclude tools/estimator_throughput.cpp -Lvins-mobile_amd/csrc
//   -lvio_amd -Wl,-rpath,$PWD/vins-mobile_amd/csrc -lpthread -o tools/estimator_throughput
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <array>
#include <chrono>
#include <thread>
#include <vector>

#include "vio_amd.h"

struct Frame {
  double header;
  std::vector<double> dt, acc, gyr;
  std::vector<VioObs> obs;
  double P[3], R[9], V[3];
};
struct World {
  double tic[3], ric[9], ba[3], bg[3];
  std::vector<Frame> frames;
};

static bool rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }

int main(int argc, char **argv) {
  if (argc < 4) return fprintf(stderr, "usage: estimator_throughput <dataset> <sequences per estimator> <estimators>\n"), 2;
  const int S = atoi(argv[2]), E = atoi(argv[3]);
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t nw = 0, nf = 0;
  rd(f, &nw, 4), rd(f, &nf, 4);
  std::vector<World> worlds(nw);
  for (World &w : worlds) {
    rd(f, w.tic, 24), rd(f, w.ric, 72), rd(f, w.ba, 24), rd(f, w.bg, 24);
    w.frames.resize(nf);
    for (Frame &fr : w.frames) {
      int32_t ni = 0, no = 0;
      rd(f, &fr.header, 8), rd(f, fr.P, 24), rd(f, fr.R, 72), rd(f, fr.V, 24), rd(f, &ni, 4);
      fr.dt.resize(ni), fr.acc.resize(3 * ni), fr.gyr.resize(3 * ni);
      rd(f, fr.dt.data(), 8 * ni), rd(f, fr.acc.data(), 24 * ni), rd(f, fr.gyr.data(), 24 * ni);
      rd(f, &no, 4);
      fr.obs.resize(no);
      for (VioObs &o : fr.obs) rd(f, &o.id, 4), rd(f, &o.x, 8), rd(f, &o.y, 8), rd(f, &o.z, 8);
    }
  }
  fclose(f);
  VioConfig cfg;
  vio_config_default(&cfg);
  cfg.max_corners = 150;
  const int W = cfg.window_size, stride = 160, imu_stride = 16;
  std::vector<double> t_begin(E), t_end(E);
  std::vector<long> solved(E, 0);
  std::vector<std::array<double, 3>> phase(E, {0, 0, 0});
  std::vector<int> frames_timed(E, 0);
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto run = [&](int e) {
    vio_estimator_t *est = nullptr;
    if (vio_estimator_create(&cfg, S, worlds[0].tic, worlds[0].ric, &est) != VIO_OK) return;
    std::vector<VioObs> obs((size_t)S * stride);
    std::vector<int32_t> n_obs(S), n_imu(S);
    std::vector<double> hdr(S), dt((size_t)S * imu_stride), acc((size_t)S * imu_stride * 3), gyr((size_t)S * imu_stride * 3);
    std::vector<VioFrameResult> res(S);
    for (int k = 0; k < nf; k++) {
      for (int q = 0; q < S; q++) {
        const World &w = worlds[(q + e) % nw];
        const Frame &fr = w.frames[k];
        n_imu[q] = (int32_t)fr.dt.size();
        memcpy(&dt[(size_t)q * imu_stride], fr.dt.data(), 8 * fr.dt.size());
        memcpy(&acc[(size_t)q * imu_stride * 3], fr.acc.data(), 24 * fr.dt.size());
        memcpy(&gyr[(size_t)q * imu_stride * 3], fr.gyr.data(), 24 * fr.dt.size());
        n_obs[q] = (int32_t)fr.obs.size(), hdr[q] = fr.header;
        memcpy(&obs[(size_t)q * stride], fr.obs.data(), sizeof(VioObs) * fr.obs.size());
        if (k == W) {
          std::vector<double> H(W + 1), Ps(3 * (W + 1)), Rs(9 * (W + 1)), Vs(3 * (W + 1)), Ba(3 * (W + 1)), Bg(3 * (W + 1));
          for (int i = 0; i <= W; i++) {
            H[i] = w.frames[i].header;
            memcpy(&Ps[3 * i], w.frames[i].P, 24), memcpy(&Rs[9 * i], w.frames[i].R, 72), memcpy(&Vs[3 * i], w.frames[i].V, 24);
            memcpy(&Ba[3 * i], w.ba, 24), memcpy(&Bg[3 * i], w.bg, 24);
          }
          vio_estimator_set_initial_state(est, q, H.data(), Ps.data(), Rs.data(), Vs.data(), Ba.data(), Bg.data());
        }
      }
      if (k == W + 2) t_begin[e] = now();
      vio_estimator_process_imu_batch(est, n_imu.data(), imu_stride, dt.data(), acc.data(), gyr.data());
      if (vio_estimator_process_images(est, obs.data(), n_obs.data(), stride, hdr.data(), nullptr, res.data()) != VIO_OK) break;
      if (k >= W + 2) {
        for (int q = 0; q < S; q++) solved[e] += res[q].action == VIO_FRAME_SOLVED;
        double ms[3];
        vio_estimator_get_timing(est, ms);
        for (int i = 0; i < 3; i++) phase[e][i] += ms[i];
        frames_timed[e]++;
      }
    }
    t_end[e] = now();
    vio_estimator_destroy(est);
  };
  std::vector<std::thread> th;
  for (int e = 0; e < E; e++) th.emplace_back(run, e);
  for (auto &t : th) t.join();
  double b = t_begin[0], en = t_end[0];
  long tot = 0;
  for (int e = 0; e < E; e++) b = std::min(b, t_begin[e]), en = std::max(en, t_end[e]), tot += solved[e];
  for (int e = 0; e < E; e++)
    printf("  estimator %d: mean ms per frame: before-solve %.2f, solve_windows %.2f, after-solve %.2f\n", e, phase[e][0] / frames_timed[e],
           phase[e][1] / frames_timed[e], phase[e][2] / frames_timed[e]);
  printf("%d estimator(s) x %d sequences: %ld window solves in %.1f ms -> %.0f solves/s end to end (host buffers in, host states out)\n", E, S,
         tot, (en - b) * 1e3, tot / (en - b));
  return 0;
}
