// tests/fuzz/host_sanity.cpp — TEST-ONLY driver that walks the host-side C++ of the library (landmark store, estimator state
// machine incl. solveInitial, initialisation pieces, PnP tracker bookkeeping, measurement queue) under
// -fsanitize=address,undefined (built and run by tests/test_abi_cpu.py). The device entry points those files call are
// stubbed HERE to return VIO_ENODEV: this binary never solves anything, it only has to finish without a sanitizer report.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "vio_amd.h"
#include "vio_resident.h"

extern "C" {  // stubs of the device side (this is not the product library)
int vio_backend_create(const VioConfig *, int32_t, vio_backend_t **) { return VIO_ENODEV; }
void vio_backend_destroy(vio_backend_t *) {}
int vio_backend_reserve_priors(vio_backend_t *, int32_t) { return VIO_ENODEV; }
int vio_backend_solve_windows(vio_backend_t *, VioWindow *, int32_t, int32_t, VioSolveStats *) { return VIO_ENODEV; }
int vio_backend_upload(vio_backend_t *, const VioWindow *, int32_t) { return VIO_ENODEV; }
int vio_backend_launch(vio_backend_t *, void *) { return VIO_ENODEV; }
int vio_backend_download(vio_backend_t *, VioWindow *, int32_t, VioSolveStats *) { return VIO_ENODEV; }
int vio_pnp_create(const VioConfig *, int32_t, vio_pnp_t **) { return VIO_ENODEV; }
void vio_pnp_destroy(vio_pnp_t *) {}
int vio_pnp_solve_windows(vio_pnp_t *, VioPnpWindow *, int32_t, VioSolveStats *) { return VIO_ENODEV; }
int32_t vio_prior_capacity(int32_t W) { return 15 * (W + 1) + 6; }
void vio_config_default(VioConfig *c) {
  memset(c, 0, sizeof(*c));
  c->window_size = 10, c->max_features = 1000, c->max_factors = 20000, c->max_iterations = 10, c->image_rows = 640, c->image_cols = 480;
  c->max_corners = 150, c->min_dist = 30, c->freq = 3, c->fx = c->fy = 460, c->cx = 240, c->cy = 320, c->gravity = 9.805;
  c->acc_n = 0.5, c->acc_w = 2e-3, c->gyr_n = 0.2, c->gyr_w = 4e-5, c->cauchy_a = 1.0;
}
}
// the device-resident path of the back-end (vio_resident.h): no device, no resident sequences
int vio_backend_set_peers(vio_backend_t *, int32_t) { return VIO_ENODEV; }
int vio_backend_resident_reserve(vio_backend_t *, int32_t, int32_t, int32_t, const double *, const double *, const double *) { return VIO_ENODEV; }
int vio_backend_resident_caps(const vio_backend_t *, int32_t *, int32_t *) { return VIO_ENODEV; }
int vio_backend_resident_load(vio_backend_t *, int32_t, const VioFeatureInfo *, int32_t, const double *, const double *, const double *) { return VIO_ENODEV; }
int vio_backend_resident_fetch(vio_backend_t *, int32_t, VioFeatureInfo *, int32_t, int32_t *, double *, int32_t, int32_t *) { return VIO_ENODEV; }
int vio_backend_resident_load_batch(vio_backend_t *, int32_t, const int32_t *, const VioFeatureInfo *const *, const int32_t *, const double *const *,
                                    const double *, const double *) { return VIO_ENODEV; }
int vio_backend_resident_begin(vio_backend_t *) { return VIO_ENODEV; }
int vio_backend_resident_stage(vio_backend_t *, int32_t, const VioObs *, int32_t, const double *, const double *, const double *, const double *,
                               const VioPrior *, int32_t, const int32_t *, const double *, int32_t) { return VIO_ENODEV; }
int vio_backend_resident_stage_preint(vio_backend_t *, int32_t, int32_t, const VioPreintegration *, const double *, const double *) { return VIO_ENODEV; }
int vio_backend_resident_stage_imu(vio_backend_t *, int32_t, int32_t, int32_t, const double *, const double *, const double *, const double *, int32_t,
                                   const double *, const double *, const double *) { return VIO_ENODEV; }
int vio_backend_resident_ingest(vio_backend_t *) { return VIO_ENODEV; }
int vio_backend_resident_launch(vio_backend_t *) { return VIO_ENODEV; }
int vio_backend_resident_collect(vio_backend_t *) { return VIO_ENODEV; }
int vio_backend_resident_result(vio_backend_t *, int32_t, VioResidentResult *, VioPrior *) { return VIO_ENODEV; }

static unsigned long long st = 0x9E3779B97F4A7C15ULL;
static double urand() { st ^= st << 13, st ^= st >> 7, st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; }
static double nrand() { return sqrt(-2 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

struct Scene {  // camera looking along +z of the body (ric = I), moving mostly sideways, slowly yawing
  std::vector<double> lm;
  Scene() {
    for (int i = 0; i < 400; i++) lm.push_back(-6 + 12 * urand()), lm.push_back(-4 + 8 * urand()), lm.push_back(4 + 6 * urand());
  }
  void pose(double t, double P[3], double R[9]) const {
    P[0] = 0.6 * t + 0.1 * sin(2 * t), P[1] = 0.15 * sin(1.3 * t), P[2] = 0.1 * cos(0.7 * t);
    const double y = 0.08 * sin(0.9 * t);
    const double Rr[9] = {cos(y), 0, sin(y), 0, 1, 0, -sin(y), 0, cos(y)};
    memcpy(R, Rr, sizeof(Rr));
  }
  std::vector<VioObs> observe(double t, int cap) const {
    double P[3], R[9];
    pose(t, P, R);
    std::vector<VioObs> o;
    for (size_t i = 0; i < lm.size() / 3 && (int)o.size() < cap; i++) {
      const double d[3] = {lm[3 * i] - P[0], lm[3 * i + 1] - P[1], lm[3 * i + 2] - P[2]};
      const double c[3] = {R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2], R[2] * d[0] + R[5] * d[1] + R[8] * d[2]};
      if (c[2] < 0.5 || fabs(c[0] / c[2]) > 0.45 || fabs(c[1] / c[2]) > 0.6) continue;
      o.push_back(VioObs{(int32_t)i, c[0] / c[2] + 0.001 * nrand(), c[1] / c[2] + 0.001 * nrand(), 1.0});
    }
    return o;
  }
};

#define REQUIRE(x)                                               \
  do {                                                           \
    if (!(x)) {                                                  \
      fprintf(stderr, "host_sanity: %s failed (line %d)\n", #x, __LINE__); \
      return 1;                                                  \
    }                                                            \
  } while (0)

int main(int argc, char **argv) {
  const int NS = argc > 1 ? atoi(argv[1]) : 3;  // sequences in the estimator (>= 32 engages the host thread pool)
  VioConfig cfg;
  vio_config_default(&cfg);
  cfg.window_size = 6;
  const int W = cfg.window_size;
  const double tic[3] = {0, 0.05, 0.01}, ric[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  Scene sc;
  // ---- estimator: three sequences at different phases, own initialisation switched on for sequence 0's sake
  vio_estimator_t *est = nullptr;
  REQUIRE(NS >= 3 && vio_estimator_create(&cfg, NS, tic, ric, &est) == VIO_OK);
  REQUIRE(vio_estimator_enable_initialization(est, 1) == VIO_OK);
  std::vector<VioObs> obs((size_t)NS * 160);
  std::vector<VioFrameResult> res(NS);
  int solve_attempts = 0;
  for (int k = 0; k < 40; k++) {
    const double t = 0.1 * k;
    std::vector<int32_t> n_imu(NS);
    for (int q = 0; q < NS; q++) n_imu[q] = q % 3 == 2 ? (k % 2 ? 10 : 0) : 10;
    std::vector<double> dt((size_t)NS * 10, 0.01), acc((size_t)NS * 30), gyr((size_t)NS * 30);
    for (size_t i = 0; i < acc.size(); i++) acc[i] = (i % 3 == 2 ? 9.8 : 0.0) + 0.05 * nrand(), gyr[i] = 0.01 * nrand();
    REQUIRE(vio_estimator_process_imu_batch(est, n_imu.data(), 10, dt.data(), acc.data(), gyr.data()) == VIO_OK);
    std::vector<int32_t> n_obs(NS);
    std::vector<double> hdr(NS);
    std::vector<uint8_t> active(NS);
    for (int q = 0; q < NS; q++) {
      hdr[q] = t + 100 * (q % 3);
      active[q] = q % 3 == 0 ? 1 : (q % 3 == 1 ? (uint8_t)(k >= 5) : (uint8_t)(k % 3 != 1));
      std::vector<VioObs> o = sc.observe(t + 0.03 * (q % 3), q % 3 == 2 && k > 20 ? 10 : 150);  // every third sequence starves -> RESET branch
      n_obs[q] = (int32_t)o.size();
      memcpy(&obs[160 * q], o.data(), sizeof(VioObs) * o.size());
    }
    if (k == 12) {
      const int32_t ids[4] = {3, 9, 40, 77};
      const double xy[8] = {0, 0, 0.1, 0.1, -0.1, 0.2, 0.3, -0.2}, P_old[3] = {1, 2, 3}, Q_old[4] = {0, 0, 0, 1};
      double headers[16];
      REQUIRE(vio_estimator_get_window(est, 1, nullptr, nullptr, nullptr, nullptr, nullptr, headers) == VIO_OK);
      REQUIRE(vio_estimator_set_relocalization(est, 1, headers[2], P_old, Q_old, ids, xy, 4) == VIO_OK);
    }
    const int rc = vio_estimator_process_images(est, obs.data(), n_obs.data(), 160, hdr.data(), active.data(), res.data());
    if (rc == VIO_ENODEV) solve_attempts++;  // solveInitial went through and wanted the device: stubbed here
    else REQUIRE(rc == VIO_OK || rc == VIO_ESTATE);  // (sequence 2 is fed IMU irregularly: a window without an interval)
    VioEstimatorStatus stt;
    REQUIRE(vio_estimator_get_status(est, 0, &stt) == VIO_OK);
    if (k == 30) REQUIRE(vio_estimator_clear(est, 1) == VIO_OK);
  }
  printf("host_sanity: estimator walked, %d frames reached the (stubbed) solve after solveInitial\n", solve_attempts);
  double cP[3 * 16], cR[9 * 16];
  REQUIRE(vio_estimator_get_corrected_window(est, 0, cP, cR) == VIO_OK);
  vio_estimator_destroy(est);

  // ---- initialisation pieces on the same scene
  {
    std::vector<double> a, b;
    std::vector<VioObs> o0 = sc.observe(0.0, 400), o1 = sc.observe(0.9, 400);
    for (const VioObs &p : o0)
      for (const VioObs &q : o1)
        if (p.id == q.id) a.push_back(p.x), a.push_back(p.y), b.push_back(q.x), b.push_back(q.y);
    double R[9], t[3];
    int32_t inl = 0, ok = 0;
    REQUIRE(vio_init_relative_pose(a.data(), b.data(), (int32_t)a.size() / 2, nullptr, R, t, &inl, &ok) == VIO_OK);
    REQUIRE(ok == 1);
    const int F = 8;
    std::vector<int32_t> start(1, 0), fr;
    std::vector<double> xy;
    std::vector<std::vector<VioObs>> per(F);
    for (int k = 0; k < F; k++) per[k] = sc.observe(0.9 * k / (F - 1), 400);
    for (int id = 0; id < 400; id++) {
      for (int k = 0; k < F; k++)
        for (const VioObs &p : per[k])
          if (p.id == id) fr.push_back(k), xy.push_back(p.x), xy.push_back(p.y);
      start.push_back((int32_t)fr.size());
    }
    std::vector<double> q(4 * F), T(3 * F), pts(3 * 400);
    std::vector<uint8_t> pok(400);
    int32_t oks = 0;
    REQUIRE(vio_init_sfm(F, 0, R, t, 400, start.data(), fr.data(), xy.data(), q.data(), T.data(), pts.data(), pok.data(), &oks) == VIO_OK);
    printf("host_sanity: relative pose inliers %d, sfm ok %d\n", inl, oks);
    // alignment on made-up frames (fails or not: only has to be clean)
    std::vector<VioInitFrame> frames(F);
    std::vector<std::vector<double>> dts(F), accs(F), gyrs(F);
    for (int k = 0; k < F; k++) {
      VioInitFrame &f = frames[k];
      memset(&f, 0, sizeof(f));
      f.header = 0.1 * k;
      double P[3];
      sc.pose(0.1 * k, P, f.R);
      memcpy(f.T, P, sizeof(P));
      f.n_samples = k ? 10 : 0;
      dts[k].assign(10, 0.01), accs[k].assign(30, 0.0), gyrs[k].assign(30, 0.001);
      for (int i = 0; i < 10; i++) accs[k][3 * i + 2] = 9.8;
      f.dt = dts[k].data(), f.acc = accs[k].data(), f.gyr = gyrs[k].data();
      f.acc_0[2] = 9.8;
    }
    std::vector<double> Bgs(3 * (W + 1), 0.0), x(3 * F + 1);
    double g[3];
    int32_t oka = 0;
    REQUIRE(vio_visual_imu_alignment(&cfg, tic, frames.data(), F, W, Bgs.data(), g, x.data(), &oka) == VIO_OK);
  }
  // ---- PnP tracker bookkeeping and the measurement queue
  {
    vio_pnp_tracker_t *tr = nullptr;
    REQUIRE(vio_pnp_tracker_create(&cfg, 2, 6, tic, ric, &tr) == VIO_OK);
    for (int k = 0; k < 12; k++) {
      for (int q = 0; q < 2; q++) {
        const double a[3] = {0, 0, 9.8}, w[3] = {0, 0, 0.01};
        for (int i = 0; i < 3; i++) REQUIRE(vio_pnp_tracker_process_imu(tr, q, 0.01, a, w) == VIO_OK);
      }
      std::vector<VioPnpFeature> f(2 * 64);
      int32_t nf[2] = {40 + k, k % 5};
      for (int q = 0; q < 2; q++)
        for (int i = 0; i < nf[q]; i++) {
          VioPnpFeature &x = f[64 * q + i];
          x.id = 2 * i + k, x.track_num = 3 + i, x.observation[0] = 0.01 * i, x.observation[1] = -0.01 * i;
          x.position[0] = i, x.position[1] = 1, x.position[2] = 5;
        }
      double hdr[2] = {0.033 * k, 5 + 0.033 * k}, P[6], R[18];
      int32_t solved[2];
      if (k == 6) {
        VioVinsResult r;
        memset(&r, 0, sizeof(r));
        r.header = 0.033 * 4, r.R[0] = r.R[4] = r.R[8] = 1;
        REQUIRE(vio_pnp_tracker_set_init(tr, 0, &r) == VIO_OK);
      }
      REQUIRE(vio_pnp_tracker_process_images(tr, f.data(), nf, 64, hdr, 0, nullptr, P, R, solved) == VIO_OK);
    }
    vio_pnp_tracker_destroy(tr);
    vio_measurements_t *mq = nullptr;
    REQUIRE(vio_measurements_create(&mq) == VIO_OK);
    std::vector<VioObs> o = sc.observe(0, 50);
    for (int i = 0; i < 300; i++) {
      VioImuMsg m = {1.0 + 0.01 * i, {0, 0, 9.8}, {0, 0, 0}};
      REQUIRE(vio_measurements_push_imu(mq, &m) == VIO_OK);
      if (i % 10 == 3) REQUIRE(vio_measurements_push_image(mq, 0.95 + 0.01 * i, o.data(), (int32_t)o.size()) == VIO_OK);
      VioImuMsg out[64];
      double dt[64], h;
      VioObs oo[64];
      int32_t ni, no, av;
      while (vio_measurements_next(mq, out, dt, 64, &ni, &h, oo, 64, &no, &av) == VIO_OK && av) {
      }
    }
    vio_measurements_destroy(mq);
  }
  printf("host_sanity: done\n");
  return 0;
}
