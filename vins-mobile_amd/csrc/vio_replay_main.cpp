// vio_replay — headless playback of a recording made by the reference app's record mode, through the C ABI only:
//
//   <dir>/IMU              IMU_MSG stream                      (ViewController.mm:1120-1150, 1614-1622)
//   <dir>/IMAGE/<i>        PNG frames, <dir>/IMAGE_TIME/<i>    (ViewController.mm:1634-1708)
//   <dir>/INIT  (optional) window states in place of solveInitial: records of 22 doubles
//                          {header, P[3], R[9] row-major, V[3], Ba[3], Bg[3]} (this tool's own format); without it the
//                          estimator initialises itself (relative pose, SfM, visual-inertial alignment)
//   -> <out>               KEYFRAME_DATA per solved frame: header, position, attitude (x y z w) of the newest frame
//
// It strings the calls in the order the app's two threads make them: camera callback (cvtColor + CLAHE + readImage,
// publish every FREQ-th frame, ViewController.mm:364-494) and estimator loop (getMeasurements -> send_imu ->
// processImage, ViewController.mm:603-724). Everything it computes goes through vio_amd.h.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "vio_amd.h"
#include "vio_math.h"

namespace {

struct InitState {
  double P[3], R[9], V[3], Ba[3], Bg[3];
};

int usage() {
  fprintf(stderr,
          "usage: vio_replay <recording dir> <pose log out> [--no-clahe] [--max-corners N] [--min-dist N] [--window W]\n"
          "                  [--tic x y z] [--ric-ypr yaw pitch roll] [--fx f --fy f --cx c --cy c]\n");
  return 2;
}

#define CHECK(call)                                                  \
  do {                                                               \
    int rc_ = (call);                                                \
    if (rc_ != VIO_OK) {                                             \
      fprintf(stderr, "vio_replay: %s failed: %d\n", #call, rc_);    \
      return 1;                                                      \
    }                                                                \
  } while (0)

}  // namespace

int main(int argc, char **argv) {
  if (argc < 3) return usage();
  const std::string dir = argv[1], out_path = argv[2];
  VioConfig cfg;
  vio_config_default(&cfg);
  bool clahe = true;
  double tic[3] = {0.0, 0.092, 0.01};            // TIC_X/Y/Z of iPhone7P (global_param.cpp:37-39)
  double ypr[3] = {0.0, 0.0, 180.0};             // RIC_y, RIC_p, RIC_r (global_param.cpp:16-18)
  for (int i = 3; i < argc; i++) {
    const std::string a = argv[i];
    auto need = [&](int n) { return i + n < argc; };
    if (a == "--no-clahe") clahe = false;
    else if (a == "--max-corners" && need(1)) cfg.max_corners = atoi(argv[++i]);
    else if (a == "--min-dist" && need(1)) cfg.min_dist = atoi(argv[++i]);
    else if (a == "--window" && need(1)) cfg.window_size = atoi(argv[++i]);
    else if (a == "--fx" && need(1)) cfg.fx = atof(argv[++i]);
    else if (a == "--fy" && need(1)) cfg.fy = atof(argv[++i]);
    else if (a == "--cx" && need(1)) cfg.cx = atof(argv[++i]);
    else if (a == "--cy" && need(1)) cfg.cy = atof(argv[++i]);
    else if (a == "--tic" && need(3)) { for (int k = 0; k < 3; k++) tic[k] = atof(argv[++i]); }
    else if (a == "--ric-ypr" && need(3)) { for (int k = 0; k < 3; k++) ypr[k] = atof(argv[++i]); }
    else return usage();
  }
  double ric[9];
  vio::ypr2R(ypr, ric);
  const int W = cfg.window_size, P = W + 1;

  // ---- the recording ----
  int32_t n_imu = 0;
  CHECK(vio_replay_read_imu((dir + "/IMU").c_str(), nullptr, 0, &n_imu));
  std::vector<VioImuMsg> imu(n_imu > 0 ? n_imu : 1);
  CHECK(vio_replay_read_imu((dir + "/IMU").c_str(), imu.data(), n_imu, &n_imu));
  std::map<double, InitState> init;
  if (FILE *f = fopen((dir + "/INIT").c_str(), "rb")) {
    double rec[22];
    while (fread(rec, sizeof(double), 22, f) == 22) {
      InitState s;
      memcpy(s.P, rec + 1, 24), memcpy(s.R, rec + 4, 72), memcpy(s.V, rec + 13, 24), memcpy(s.Ba, rec + 16, 24), memcpy(s.Bg, rec + 19, 24);
      init[rec[0]] = s;
    }
    fclose(f);
  }
  int32_t rows = 0, cols = 0;
  if (vio_replay_read_image((dir + "/IMAGE").c_str(), 0, nullptr, 0, &rows, &cols) != VIO_ECAP) {
    fprintf(stderr, "vio_replay: %s/IMAGE/0 is not a readable PNG\n", dir.c_str());
    return 1;
  }
  cfg.image_rows = rows, cfg.image_cols = cols;
  printf("vio_replay: %d IMU samples, frames %dx%d, %zu initial states, window %d\n", n_imu, rows, cols, init.size(), W);

  vio_frontend_t *fe = nullptr;
  vio_estimator_t *est = nullptr;
  vio_preprocess_t *pp = nullptr;
  vio_measurements_t *mq = nullptr;
  CHECK(vio_frontend_create(&cfg, 1, &fe));
  CHECK(vio_estimator_create(&cfg, 1, tic, ric, &est));
  // no INIT file: the estimator's own solveInitial; relativePose by the fit over all correspondences (2) unless the
  // reference's five-point route is asked for (VIO_REPLAY_RELPOSE=reference: a frame that draws a wrong root retries)
  if (init.empty()) {
    const char *rp = getenv("VIO_REPLAY_RELPOSE");
    CHECK(vio_estimator_enable_initialization(est, rp && !strcmp(rp, "reference") ? 1 : 2));
  }
  CHECK(vio_measurements_create(&mq));
  if (clahe) CHECK(vio_preprocess_create(1, rows, cols, &pp));

  std::vector<uint8_t> gray((size_t)rows * cols), equ((size_t)rows * cols);
  std::vector<VioObs> obs(cfg.max_corners), mobs(cfg.max_corners);
  std::vector<VioImuMsg> batch(4096);
  std::vector<double> dts(4096), hdr(P), Ps(3 * P), Rs(9 * P), Vs(3 * P), Bas(3 * P), Bgs(3 * P);
  std::vector<VioKeyframeData> log;
  int img_cnt = 0, next_imu = 0, solved = 0, failures = 0;
  for (uint64_t index = 0;; index++) {
    double t = 0;
    if (vio_replay_read_image_time((dir + "/IMAGE_TIME").c_str(), index, &t) != VIO_OK) break;  // still_play == false
    int32_t r = 0, c = 0;
    if (vio_replay_read_image((dir + "/IMAGE").c_str(), index, gray.data(), (int64_t)gray.size(), &r, &c) != VIO_OK || r != rows || c != cols) break;
    // camera callback: pre-step, readImage, publish every FREQ-th frame
    const uint8_t *frame = gray.data();
    if (clahe) {
      CHECK(vio_preprocess_run(pp, gray.data(), 1, 1, cols, nullptr, equ.data()));
      frame = equ.data();
    }
    const int publish = img_cnt == 0;
    int32_t n_obs = 0;
    CHECK(vio_frontend_read_image(fe, 0, frame, rows, cols, cols, t, publish, obs.data(), &n_obs, nullptr));
    img_cnt = (img_cnt + 1) % cfg.freq;
    // IMU callback: everything recorded up to the first sample past this frame has arrived by now
    while (next_imu < n_imu && (next_imu == 0 || imu[next_imu - 1].header <= t)) CHECK(vio_measurements_push_imu(mq, &imu[next_imu++]));
    if (publish) CHECK(vio_measurements_push_image(mq, t, obs.data(), n_obs));
    // estimator loop
    while (true) {
      int32_t nb = 0, no = 0, avail = 0;
      double header = 0;
      CHECK(vio_measurements_next(mq, batch.data(), dts.data(), (int32_t)batch.size(), &nb, &header, mobs.data(),
                                  (int32_t)mobs.size(), &no, &avail));
      if (!avail) break;
      for (int i = 0; i < nb; i++) CHECK(vio_estimator_process_imu(est, 0, dts[i], batch[i].acc, batch[i].gyr));
      VioEstimatorStatus st;
      CHECK(vio_estimator_get_status(est, 0, &st));
      if (st.solver_flag == VIO_SOLVER_INITIAL && st.frame_count == W && !init.empty()) {
        CHECK(vio_estimator_get_window(est, 0, nullptr, nullptr, nullptr, nullptr, nullptr, hdr.data()));
        hdr[W] = header;
        bool all = true;
        for (int i = 0; i < P && all; i++) {
          auto it = init.find(hdr[i]);
          if (it == init.end()) { all = false; break; }
          memcpy(&Ps[3 * i], it->second.P, 24), memcpy(&Rs[9 * i], it->second.R, 72), memcpy(&Vs[3 * i], it->second.V, 24);
          memcpy(&Bas[3 * i], it->second.Ba, 24), memcpy(&Bgs[3 * i], it->second.Bg, 24);
        }
        if (all) CHECK(vio_estimator_set_initial_state(est, 0, hdr.data(), Ps.data(), Rs.data(), Vs.data(), Bas.data(), Bgs.data()));
      }
      VioFrameResult res;
      CHECK(vio_estimator_process_image(est, 0, mobs.data(), no, header, &res));
      if (res.action == VIO_FRAME_FAILURE) failures++;
      if (res.action == VIO_FRAME_SOLVED) {
        CHECK(vio_estimator_get_window(est, 0, Ps.data(), Rs.data(), nullptr, nullptr, nullptr, hdr.data()));
        VioKeyframeData k;
        k.header = hdr[W];
        memcpy(k.translation, &Ps[3 * W], 24);
        const vio::Quat q = vio::RtoQ(&Rs[9 * W]);
        k.rotation[0] = q.x, k.rotation[1] = q.y, k.rotation[2] = q.z, k.rotation[3] = q.w;
        log.push_back(k);
        solved++;
      }
    }
  }
  CHECK(vio_replay_write_keyframes(out_path.c_str(), log.data(), (int32_t)log.size()));
  printf("vio_replay: %d solved frames, %d failures -> %s\n", solved, failures, out_path.c_str());
  if (pp) vio_preprocess_destroy(pp);
  vio_measurements_destroy(mq);
  vio_estimator_destroy(est);
  vio_frontend_destroy(fe);
  return 0;
}
