// shim_main.cpp — compiles include/vio_amd_shim.hpp (the reference's member signatures over the C ABI) WITHOUT OpenCV /
// Eigen: the traits below carry just the members the shim touches. Driven by tests/test_shim_gpu.py:
//   shim_main frames.bin rows cols n_frames freq obs_out.bin        FeatureTracker::readImage on raw gray frames
//   shim_main --vins data.bin out.bin                               VINS::processIMU / processImage on a recorded sequence
//   shim_main --pnp frames.bin rows cols n_frames pnp.bin out.bin   readImage with vins_normal: the solveVinsPnP branch
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <list>
#include <vector>

#include "vio_amd_shim.hpp"

struct Mat {   // the members of cv::Mat the shim reads
  unsigned char *data;
  int rows, cols;
  size_t step;
};
struct Point2f {
  float x, y;
  Point2f(float x_ = 0, float y_ = 0) : x(x_), y(y_) {}
};
struct Vec3 {
  double v[3];
  Vec3() { v[0] = v[1] = v[2] = 0; }
  double &operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
};
struct Vec2 {
  double v[2];
  Vec2() { v[0] = v[1] = 0; }
  double &operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
};
struct Mat3 {
  double m[9];
  Mat3() { memset(m, 0, sizeof(m)); }
  double &operator()(int i, int j) { return m[3 * i + j]; }
  double operator()(int i, int j) const { return m[3 * i + j]; }
};
struct Traits {
  typedef ::Mat Mat;
  typedef ::Point2f Point2f;
  typedef Vec2 Vector2d;
  typedef Vec3 Vector3d;
  typedef Mat3 Matrix3d;
};

static std::vector<unsigned char> slurp(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> b(n);
  if (fread(b.data(), 1, n, f) != (size_t)n) exit(2);
  fclose(f);
  return b;
}

static int run_tracker(int argc, char **argv) {
  if (argc < 7) return 2;
  const int rows = atoi(argv[2]), cols = atoi(argv[3]), n_frames = atoi(argv[4]), freq = atoi(argv[5]);
  std::vector<unsigned char> frames = slurp(argv[1]);
  VioConfig cfg;
  vio_config_default(&cfg);
  cfg.image_rows = rows, cfg.image_cols = cols, cfg.max_corners = 60, cfg.min_dist = 25;
  vio_shim::FeatureTracker<Traits> tracker(cfg);
  FILE *out = fopen(argv[6], "wb");
  for (int f = 0; f < n_frames; f++) {
    Mat img = {frames.data() + (size_t)f * rows * cols, rows, cols, (size_t)cols}, result = {nullptr, 0, 0, 0};
    std::vector<Point2f> good_pts;
    std::vector<double> track_len;
    Vec3 P;
    Mat3 R;
    tracker.readImage(img, result, f, good_pts, track_len, 0.1 * f, P, R, false);
    if (result.data != img.data) return 3;
    const int published = tracker.img_cnt == 0;
    int32_t hdr[3] = {f, published, published ? (int32_t)tracker.image_msg.size() : 0};
    fwrite(hdr, sizeof(hdr), 1, out);
    if (published)
      for (auto &kv : tracker.image_msg) {
        double rec[4] = {(double)kv.first, kv.second(0), kv.second(1), kv.second(2)};
        fwrite(rec, sizeof(rec), 1, out);
      }
    int32_t ng = (int32_t)good_pts.size();
    fwrite(&ng, sizeof(ng), 1, out);
    tracker.img_cnt = (tracker.img_cnt + 1) % freq;  // ViewController.mm:494
  }
  fclose(out);
  return 0;
}

// pnp.bin: int32 first_frame (solved_* are set from that frame on), n_solved, imu_per_frame; n_solved x (int32 id, int32
// track_num, double position[3]); VINS_RESULT (header, Ba, Bg, P, R[9], V); then per frame imu_per_frame x (header, acc[3],
// gyr[3]). out.bin: per frame (returned-by-readImage P[3], R[9]).
static int run_pnp(char **argv) {
  const int rows = atoi(argv[3]), cols = atoi(argv[4]), n_frames = atoi(argv[5]);
  std::vector<unsigned char> frames = slurp(argv[2]), d = slurp(argv[6]);
  const unsigned char *p = d.data();
  auto rd = [&](void *dst, size_t n) { memcpy(dst, p, n), p += n; };
  typedef vio_shim::FeatureTracker<Traits> Tracker;
  VioConfig cfg;
  vio_config_default(&cfg);
  cfg.image_rows = rows, cfg.image_cols = cols, cfg.max_corners = 60, cfg.min_dist = 25;
  const double tic[3] = {0, 0, 0}, ric[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  Tracker tracker(cfg, tic, ric);
  tracker.use_pnp = true;  // featuretracker.use_pnp = USE_PNP (ViewController.mm:457)
  int32_t first, n_solved, ipf;
  rd(&first, 4), rd(&n_solved, 4), rd(&ipf, 4);
  std::list<Tracker::IMG_MSG_LOCAL> solved;
  for (int i = 0; i < n_solved; i++) {
    Tracker::IMG_MSG_LOCAL m;
    int32_t id, tn;
    rd(&id, 4), rd(&tn, 4), rd(m.position.v, 24);
    m.id = id, m.track_num = tn;
    solved.push_back(m);
  }
  Tracker::VINS_RESULT res;
  rd(&res.header, 8), rd(res.Ba.v, 24), rd(res.Bg.v, 24), rd(res.P.v, 24), rd(res.R.m, 72), rd(res.V.v, 24);
  FILE *out = fopen(argv[7], "wb");
  for (int f = 0; f < n_frames; f++) {
    tracker.imu_msgs.clear();
    for (int s = 0; s < ipf; s++) {
      Tracker::IMU_MSG_LOCAL m;
      rd(&m.header, 8), rd(m.acc.v, 24), rd(m.gyr.v, 24);
      tracker.imu_msgs.push_back(m);
    }
    if (f >= first) tracker.solved_features = solved, tracker.solved_vins = res;
    Mat img = {frames.data() + (size_t)f * rows * cols, rows, cols, (size_t)cols}, result = {nullptr, 0, 0, 0};
    std::vector<Point2f> good_pts;
    std::vector<double> track_len;
    Vec3 P;
    Mat3 R;
    tracker.readImage(img, result, f, good_pts, track_len, 0.1 * f, P, R, f >= first);
    fwrite(P.v, sizeof(P.v), 1, out), fwrite(R.m, sizeof(R.m), 1, out);
    tracker.img_cnt = (tracker.img_cnt + 1) % 3;
  }
  fclose(out);
  return 0;
}

// data.bin: int32 W, n_frames, imu_per_frame; double tic[3], ric[9]; then per frame: double header; per IMU sample of the
// interval: dt, acc[3], gyr[3]; int32 n_obs; n_obs x (int32 id, double x, y, z). After frame W: the hand-over block
// (W+1) x (header, P[3], R[9], V[3]) + ba[3] + bg[3] precedes that frame's processImage.
static int run_vins(char **argv) {
  std::vector<unsigned char> d = slurp(argv[2]);
  const unsigned char *p = d.data();
  auto rd = [&](void *dst, size_t n) { memcpy(dst, p, n), p += n; };
  int32_t W, n_frames, ipf;
  rd(&W, 4), rd(&n_frames, 4), rd(&ipf, 4);
  double tic[3], ric[9];
  rd(tic, 24), rd(ric, 72);
  VioConfig cfg;
  vio_config_default(&cfg);
  cfg.window_size = W;
  vio_shim::VINS<Traits> vins(cfg, tic, ric);
  FILE *out = fopen(argv[3], "wb");
  for (int f = 0; f < n_frames; f++) {
    double header;
    rd(&header, 8);
    const int ns = f == 0 ? 1 : ipf;
    for (int s = 0; s < ns; s++) {
      double rec[7];
      rd(rec, 56);
      Vec3 a, g;
      for (int k = 0; k < 3; k++) a(k) = rec[1 + k], g(k) = rec[4 + k];
      vins.processIMU(rec[0], a, g);
    }
    int32_t n_obs;
    rd(&n_obs, 4);
    std::map<int, Vec3> image_msg;
    for (int i = 0; i < n_obs; i++) {
      int32_t id;
      double xyz[3];
      rd(&id, 4), rd(xyz, 24);
      Vec3 v;
      v(0) = xyz[0], v(1) = xyz[1], v(2) = xyz[2];
      image_msg[id] = v;
    }
    if (f == W) {
      std::vector<double> hs(W + 1);
      std::vector<Vec3> P(W + 1), V(W + 1);
      std::vector<Mat3> R(W + 1);
      for (int i = 0; i <= W; i++) {
        rd(&hs[i], 8), rd(P[i].v, 24), rd(R[i].m, 72), rd(V[i].v, 24);
      }
      Vec3 ba, bg;
      rd(ba.v, 24), rd(bg.v, 24);
      vins.setInitialState(hs, P, R, V, ba, bg);
    }
    vins.processImage(image_msg, header, 0);
    double rec[8] = {(double)f, (double)vins.last_result.action, (double)vins.solver_flag, vins.Ps[W](0), vins.Ps[W](1), vins.Ps[W](2),
                     (double)vins.last_result.stats.iterations, vins.last_result.stats.final_cost};
    fwrite(rec, sizeof(rec), 1, out);
  }
  fclose(out);
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 4 && !strcmp(argv[1], "--vins")) return run_vins(argv);
  if (argc >= 8 && !strcmp(argv[1], "--pnp")) return run_pnp(argv);
  return run_tracker(argc, argv);
}
