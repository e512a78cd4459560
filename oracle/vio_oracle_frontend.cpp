// oracle/vio_oracle_frontend.cpp — TEST INFRASTRUCTURE ONLY (see vio_oracle.h). PARITY UNPINNED.
//
// CPU restatement of FeatureTracker::readImage (VINS_ios/feature_tracker.cpp:162-321) and of the OpenCV routines it
// calls. OpenCV ("customized 3.0.0", VINS_ThirdPartyLib/opencv2.version:1) is neither in /root/reference nor
// installed, so these follow the PUBLISHED OpenCV 3.0.0 algorithms; nothing here could be checked against a build
// of the real library. Call sites and upstream files:
//   calcOpticalFlowPyrLK(cur, forw, pts, ..., Size(21,21), 3)   feature_tracker.cpp:181   video/src/lkpyramid.cpp
//   goodFeaturesToTrack(img, n_pts, n, 0.01, 30, mask)          feature_tracker.cpp:263   imgproc/src/featureselect.cpp,
//                                                                                          corner.cpp, deriv.cpp
//   findFundamentalMat(p1, p2, FM_RANSAC, 1.0, 0.99, status)    feature_tracker.cpp:95,198 calib3d/src/fundam.cpp,
//                                                                                          ptsetreg.cpp, core mathfuncs
//   cv::circle(mask, p, 30, 0, -1)                              feature_tracker.cpp:80    imgproc/src/drawing.cpp
//
// Stated deviations (places where OpenCV's own result depends on build flavour or is unspecified):
//  * LK sums A11/A12/A22/b1/b2: OpenCV accumulates float per pixel in raster order (scalar path) or in 4-lane
//    partial sums (SSE2/NEON paths). Here the integer products are summed EXACTLY and rounded to float once; every
//    OpenCV path approximates that value. (mode 1 below reproduces the scalar raster-order accumulation.)
//  * box sum of the 3x3 derivative products: OpenCV keeps running row/column sums across the image; here each
//    window is summed directly in a fixed order.
//  * corner candidates of equal strength: std::sort order is unspecified in 3.0.0; ties break by raster index.
//  * setMask's std::sort on track_cnt is likewise unspecified for ties; a stable sort is used.
//  * 7-point null space: OpenCV takes the last two right singular vectors of the 7x9 system; here the null space
//    comes from Gauss-Jordan elimination with complete pivoting. The set of candidate F matrices is basis-invariant.
//  * fewer than 15 correspondences: OpenCV 3.0.0 switches to LMedS (fundam.cpp); restated below from the published
//    3.0.0 algorithm (ptsetreg.cpp LMeDSPointSetRegistrator::run), like the RANSAC path.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "vio_oracle.h"

namespace {

int g_lk_accum_mode = 0;  // 0: exact integer sums (default), 1: OpenCV scalar-path float accumulation order

inline int cv_floor(float v) { return (int)floorf(v); }
inline int cv_round(double v) { return (int)lrint(v); }  // round-half-to-even like SSE2 cvRound
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}

struct Image {
  int rows, cols;
  std::vector<uint8_t> px;
  uint8_t at(int y, int x) const { return px[(size_t)reflect101(y, rows) * cols + reflect101(x, cols)]; }
};

// pyrDown for 8-bit images: separable [1 4 6 4 1]/16, BORDER_REFLECT_101, (v + 128) >> 8 (imgproc/src/pyramids.cpp)
void pyr_down(const Image &s, Image &d) {
  d.rows = (s.rows + 1) / 2, d.cols = (s.cols + 1) / 2;
  d.px.resize((size_t)d.rows * d.cols);
  std::vector<int> hrow((size_t)5 * d.cols);
  for (int y = 0; y < d.rows; y++) {
    for (int k = -2; k <= 2; k++) {
      int sy = reflect101(2 * y + k, s.rows);
      const uint8_t *row = &s.px[(size_t)sy * s.cols];
      int *h = &hrow[(size_t)(k + 2) * d.cols];
      for (int x = 0; x < d.cols; x++) {
        int c = 2 * x;
        h[x] = row[reflect101(c - 2, s.cols)] + 4 * row[reflect101(c - 1, s.cols)] + 6 * row[reflect101(c, s.cols)] +
               4 * row[reflect101(c + 1, s.cols)] + row[reflect101(c + 2, s.cols)];
      }
    }
    for (int x = 0; x < d.cols; x++) {
      int v = hrow[x] + 4 * hrow[d.cols + x] + 6 * hrow[2 * d.cols + x] + 4 * hrow[3 * d.cols + x] + hrow[4 * d.cols + x];
      d.px[(size_t)y * d.cols + x] = (uint8_t)((v + 128) >> 8);
    }
  }
}

// Scharr derivatives (calcSharrDeriv, lkpyramid.cpp): smoothing 3-10-3, difference -1 0 1, un-normalized int16;
// BORDER_REFLECT_101 inside the image, zero outside (copyMakeBorder BORDER_CONSTANT).
inline void scharr_at(const Image &im, int y, int x, int &dx, int &dy) {
  if (x < 0 || x >= im.cols || y < 0 || y >= im.rows) {
    dx = dy = 0;
    return;
  }
  int p00 = im.at(y - 1, x - 1), p01 = im.at(y - 1, x), p02 = im.at(y - 1, x + 1);
  int p10 = im.at(y, x - 1), p12 = im.at(y, x + 1);
  int p20 = im.at(y + 1, x - 1), p21 = im.at(y + 1, x), p22 = im.at(y + 1, x + 1);
  dx = 3 * (p02 - p00) + 10 * (p12 - p10) + 3 * (p22 - p20);
  dy = 3 * (p20 - p00) + 10 * (p21 - p01) + 3 * (p22 - p02);
}

struct FloatAcc {  // exact (mode 0) or float raster-order (mode 1) accumulation of integer terms
  double exact = 0;
  float seq = 0;
  void add(int v) {
    exact += (double)v;
    seq += (float)v;
  }
  void add64(long long v) {
    exact += (double)v;
    seq += (float)v;
  }
  float value() const { return g_lk_accum_mode == 0 ? (float)exact : seq; }
};

void klt_track(const VioConfig *cfg, const Image &prev0, const Image &next0, const float *prev_pts, int n,
               float *next_pts, uint8_t *status, float *err) {
  const int win = cfg->lk_win;
  int max_level = cfg->lk_levels;
  std::vector<Image> pp(1, prev0), np(1, next0);
  for (int l = 1; l <= max_level; l++) {  // buildOpticalFlowPyramid: stop when a level would not hold the window
    Image a, b;
    pyr_down(pp[l - 1], a), pyr_down(np[l - 1], b);
    if (a.cols <= win || a.rows <= win) {
      max_level = l - 1;
      break;
    }
    pp.push_back(a), np.push_back(b);
  }
  const float half = (win - 1) * 0.5f;
  const int W_BITS = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float FLT_EPS = 1.1920929e-07f;
  int max_count = std::min(std::max(cfg->lk_max_iters, 0), 100);
  double epsilon = std::min(std::max(cfg->lk_eps, 0.), 10.);
  epsilon *= epsilon;
  std::vector<short> Ibuf((size_t)win * win), dIbuf((size_t)win * win * 2);
  for (int i = 0; i < n; i++) status[i] = 1, err[i] = 0;
  std::vector<float> nx(n), ny(n);
  for (int level = max_level; level >= 0; level--) {
    const Image &I = pp[level], &J = np[level];
    for (int pt = 0; pt < n; pt++) {
      float scale = (float)(1. / (1 << level));
      float px = prev_pts[2 * pt] * scale, py = prev_pts[2 * pt + 1] * scale;
      float qx, qy;
      if (level == max_level) qx = px, qy = py;
      else qx = nx[pt] * 2.f, qy = ny[pt] * 2.f;
      nx[pt] = qx, ny[pt] = qy;
      px -= half, py -= half;
      int ipx = cv_floor(px), ipy = cv_floor(py);
      if (ipx < -win || ipx >= I.cols || ipy < -win || ipy >= I.rows) {
        if (level == 0) status[pt] = 0, err[pt] = 0;
        continue;
      }
      float a = px - ipx, b = py - ipy;
      int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
      int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
      int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
      int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      FloatAcc s11, s12, s22;
      for (int y = 0; y < win; y++)
        for (int x = 0; x < win; x++) {
          int X = ipx + x, Y = ipy + y;
          int ival = descale(I.at(Y, X) * iw00 + I.at(Y, X + 1) * iw01 + I.at(Y + 1, X) * iw10 + I.at(Y + 1, X + 1) * iw11,
                             W_BITS - 5);
          int dx00, dy00, dx01, dy01, dx10, dy10, dx11, dy11;
          scharr_at(I, Y, X, dx00, dy00), scharr_at(I, Y, X + 1, dx01, dy01);
          scharr_at(I, Y + 1, X, dx10, dy10), scharr_at(I, Y + 1, X + 1, dx11, dy11);
          int ixval = descale(dx00 * iw00 + dx01 * iw01 + dx10 * iw10 + dx11 * iw11, W_BITS);
          int iyval = descale(dy00 * iw00 + dy01 * iw01 + dy10 * iw10 + dy11 * iw11, W_BITS);
          Ibuf[y * win + x] = (short)ival;
          dIbuf[2 * (y * win + x)] = (short)ixval, dIbuf[2 * (y * win + x) + 1] = (short)iyval;
          s11.add(ixval * ixval), s12.add(ixval * iyval), s22.add(iyval * iyval);
        }
      float A11 = s11.value() * FLT_SCALE, A12 = s12.value() * FLT_SCALE, A22 = s22.value() * FLT_SCALE;
      float D = A11 * A22 - A12 * A12;
      float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
      if (minEig < (float)cfg->lk_min_eig || D < FLT_EPS) {
        if (level == 0) status[pt] = 0;
        continue;
      }
      D = 1.f / D;
      qx -= half, qy -= half;
      float pdx = 0, pdy = 0;
      for (int j = 0; j < max_count; j++) {
        int iqx = cv_floor(qx), iqy = cv_floor(qy);
        if (iqx < -win || iqx >= J.cols || iqy < -win || iqy >= J.rows) {
          if (level == 0) status[pt] = 0;
          break;
        }
        a = qx - iqx, b = qy - iqy;
        iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        FloatAcc sb1, sb2;
        for (int y = 0; y < win; y++)
          for (int x = 0; x < win; x++) {
            int X = iqx + x, Y = iqy + y;
            int diff = descale(J.at(Y, X) * iw00 + J.at(Y, X + 1) * iw01 + J.at(Y + 1, X) * iw10 + J.at(Y + 1, X + 1) * iw11,
                               W_BITS - 5) - Ibuf[y * win + x];
            sb1.add(diff * dIbuf[2 * (y * win + x)]), sb2.add(diff * dIbuf[2 * (y * win + x) + 1]);
          }
        float b1 = sb1.value() * FLT_SCALE, b2 = sb2.value() * FLT_SCALE;
        float ddx = (float)((A12 * b2 - A22 * b1) * D), ddy = (float)((A12 * b1 - A11 * b2) * D);
        qx += ddx, qy += ddy;
        nx[pt] = qx + half, ny[pt] = qy + half;
        if ((double)ddx * ddx + (double)ddy * ddy <= epsilon) break;
        if (j > 0 && fabs((double)(ddx + pdx)) < 0.01 && fabs((double)(ddy + pdy)) < 0.01) {
          nx[pt] -= ddx * 0.5f, ny[pt] -= ddy * 0.5f;
          break;
        }
        pdx = ddx, pdy = ddy;
      }
      if (status[pt] && level == 0) {  // err = mean |I - J| / 32 over the final window
        float ex = nx[pt] - half, ey = ny[pt] - half;
        int iex = cv_floor(ex), iey = cv_floor(ey);
        if (iex < -win || iex >= J.cols || iey < -win || iey >= J.rows) {
          status[pt] = 0;
          continue;
        }
        float aa = ex - iex, bb = ey - iey;
        iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
        iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
        iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        FloatAcc se;
        for (int y = 0; y < win; y++)
          for (int x = 0; x < win; x++) {
            int X = iex + x, Y = iey + y;
            int diff = descale(J.at(Y, X) * iw00 + J.at(Y, X + 1) * iw01 + J.at(Y + 1, X) * iw10 + J.at(Y + 1, X + 1) * iw11,
                               W_BITS - 5) - Ibuf[y * win + x];
            se.add(abs(diff));
          }
        err[pt] = se.value() * 1.f / (32 * win * win);
      }
    }
  }
  for (int i = 0; i < n; i++) next_pts[2 * i] = nx[i], next_pts[2 * i + 1] = ny[i];
}

// cornerMinEigenVal(img, eig, blockSize 3, ksize 3): Sobel scaled by 1/(4*3*255) -> products -> 3x3 box sum ->
// (a + c) - sqrt((a - c)^2 + b^2) with a = 0.5 sum dx^2, b = sum dx dy, c = 0.5 sum dy^2 (imgproc/src/corner.cpp)
void min_eigen_map(const Image &im, std::vector<float> &eig) {
  const int R = im.rows, C = im.cols;
  const float s = (float)(1.0 / (4.0 * 3.0 * 255.0));
  const float s2 = s * 2.f;  // kernel [s 2s s]: cv::Sobel scales the smoothing kernel (deriv.cpp)
  std::vector<float> dxx((size_t)R * C), dxy((size_t)R * C), dyy((size_t)R * C);
  for (int y = 0; y < R; y++)
    for (int x = 0; x < C; x++) {
      // dx: row filter [-1 0 1] (exact ints), then symmetric column filter f*c + k1*(u + d)
      float r0 = (float)(im.at(y - 1, x + 1) - im.at(y - 1, x - 1));
      float r1 = (float)(im.at(y, x + 1) - im.at(y, x - 1));
      float r2 = (float)(im.at(y + 1, x + 1) - im.at(y + 1, x - 1));
      float dx = s2 * r1 + s * (r0 + r2);
      // dy: symmetric row filter s2*c + s*(l + r), then column filter [-1 0 1]
      float t0 = s2 * (float)im.at(y - 1, x) + s * ((float)im.at(y - 1, x - 1) + (float)im.at(y - 1, x + 1));
      float t2 = s2 * (float)im.at(y + 1, x) + s * ((float)im.at(y + 1, x - 1) + (float)im.at(y + 1, x + 1));
      float dy = t2 - t0;
      dxx[(size_t)y * C + x] = dx * dx, dxy[(size_t)y * C + x] = dx * dy, dyy[(size_t)y * C + x] = dy * dy;
    }
  eig.assign((size_t)R * C, 0.f);
  auto box = [&](const std::vector<float> &m, int y, int x) {
    float rs[3];
    for (int k = -1; k <= 1; k++) {
      int yy = reflect101(y + k, R);
      const float *row = &m[(size_t)yy * C];
      rs[k + 1] = (row[reflect101(x - 1, C)] + row[x]) + row[reflect101(x + 1, C)];
    }
    return (rs[0] + rs[1]) + rs[2];
  };
  for (int y = 0; y < R; y++)
    for (int x = 0; x < C; x++) {
      float a = box(dxx, y, x) * 0.5f, b = box(dxy, y, x), c = box(dyy, y, x) * 0.5f;
      eig[(size_t)y * C + x] = (a + c) - sqrtf((a - c) * (a - c) + b * b);
    }
}

// goodFeaturesToTrack (imgproc/src/featureselect.cpp), blockSize 3, Shi-Tomasi
int good_features(const Image &im, const uint8_t *mask, int max_corners, double quality, double min_dist, float *corners) {
  const int R = im.rows, C = im.cols;
  std::vector<float> eig;
  min_eigen_map(im, eig);
  float maxv = 0.f;
  bool any = false;
  for (int i = 0; i < R * C; i++)
    if (!mask || mask[i]) {
      if (!any || eig[i] > maxv) maxv = eig[i], any = true;
    }
  if (!any) return 0;
  float thr = (float)((double)maxv * quality);
  for (int i = 0; i < R * C; i++)
    if (!(eig[i] > thr)) eig[i] = 0.f;  // THRESH_TOZERO
  std::vector<int> cand;
  for (int y = 1; y < R - 1; y++)
    for (int x = 1; x < C - 1; x++) {
      float v = eig[(size_t)y * C + x];
      if (v == 0.f || (mask && !mask[(size_t)y * C + x])) continue;
      float m = v;  // 3x3 dilate; outside pixels do not exist here since 1 <= x,y <= n-2
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) m = std::max(m, eig[(size_t)(y + dy) * C + x + dx]);
      if (v == m) cand.push_back(y * C + x);
    }
  std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return eig[a] > eig[b]; });
  int ncorners = 0;
  if (min_dist >= 1) {
    const int cell = cv_round(min_dist);
    const int gw = (C + cell - 1) / cell, gh = (R + cell - 1) / cell;
    std::vector<std::vector<int>> grid((size_t)gw * gh);
    const double md2 = min_dist * min_dist;
    for (int idx : cand) {
      int y = idx / C, x = idx % C;
      bool good = true;
      int xc = x / cell, yc = y / cell;
      int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1), x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
      for (int yy = y1; yy <= y2 && good; yy++)
        for (int xx = x1; xx <= x2 && good; xx++)
          for (int p : grid[(size_t)yy * gw + xx]) {
            float dx = (float)(x - p % C), dy = (float)(y - p / C);
            if (dx * dx + dy * dy < md2) {
              good = false;
              break;
            }
          }
      if (good) {
        grid[(size_t)yc * gw + xc].push_back(idx);
        corners[2 * ncorners] = (float)x, corners[2 * ncorners + 1] = (float)y;
        if (++ncorners == max_corners && max_corners > 0) break;
      }
    }
  } else {
    for (int idx : cand) {
      corners[2 * ncorners] = (float)(idx % C), corners[2 * ncorners + 1] = (float)(idx / C);
      if (++ncorners == max_corners && max_corners > 0) break;
    }
  }
  return ncorners;
}

// ---- findFundamentalMat(FM_RANSAC) --------------------------------------------------------------------
struct CvRng {  // cv::RNG (core/include/opencv2/core/operations.hpp): multiply-with-carry
  uint64_t state;
  explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
  unsigned next() {
    state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

bool have_collinear(const float *p, int count) {  // haveCollinearPoints: checks the LAST point against all pairs
  int i = count - 1;
  for (int j = 0; j < i; j++) {
    double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];
    for (int k = 0; k < j; k++) {
      double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
      if (fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920929e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
    }
  }
  return false;
}

int solve_cubic(const double c[4], double r[3]) {  // cv::solveCubic (core/src/mathfuncs.cpp)
  double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
  double x0 = 0, x1 = 0, x2 = 0;
  int n = 0;
  if (a0 == 0) {
    if (a1 == 0) {
      if (a2 == 0) n = a3 == 0 ? -1 : 0;
      else x0 = -a3 / a2, n = 1;
    } else {
      double d = a2 * a2 - 4 * a1 * a3;
      if (d >= 0) {
        d = sqrt(d);
        double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
        if (fabs(q1) > fabs(q2)) x0 = q1 / a1, x1 = a3 / q1;
        else x0 = q2 / a1, x1 = a3 / q2;
        n = d > 0 ? 2 : 1;
      }
    }
  } else {
    a0 = 1. / a0, a1 *= a0, a2 *= a0, a3 *= a0;
    double Q = (a1 * a1 - 3 * a2) * (1. / 9), R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
    double Qcubed = Q * Q * Q, d = Qcubed - R * R;
    if (d >= 0) {
      double theta = acos(R / sqrt(Qcubed)), sqrtQ = sqrt(Q);
      double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
      x0 = t0 * cos(t1) - t2, x1 = t0 * cos(t1 + (2. * M_PI / 3)) - t2, x2 = t0 * cos(t1 + (4. * M_PI / 3)) - t2;
      n = 3;
    } else {
      d = sqrt(-d);
      double e = pow(d + fabs(R), 0.333333333333);
      if (R > 0) e = -e;
      x0 = (e + Q / e) - a1 * (1. / 3);
      n = 1;
    }
  }
  r[0] = x0, r[1] = x1, r[2] = x2;
  return n;
}

// Null space (dimension 2) of the 7x9 epipolar system by Gauss-Jordan with complete pivoting.
void null_space_7x9(double a[63], double f1[9], double f2[9]) {
  int colperm[9];
  for (int j = 0; j < 9; j++) colperm[j] = j;
  for (int k = 0; k < 7; k++) {
    int pr = k, pc = k;
    double best = -1;
    for (int i = k; i < 7; i++)
      for (int j = k; j < 9; j++)
        if (fabs(a[i * 9 + j]) > best) best = fabs(a[i * 9 + j]), pr = i, pc = j;
    if (pr != k)
      for (int j = 0; j < 9; j++) std::swap(a[k * 9 + j], a[pr * 9 + j]);
    if (pc != k) {
      for (int i = 0; i < 7; i++) std::swap(a[i * 9 + k], a[i * 9 + pc]);
      std::swap(colperm[k], colperm[pc]);
    }
    double d = a[k * 9 + k];
    if (d == 0.0) continue;
    for (int j = 0; j < 9; j++) a[k * 9 + j] /= d;
    for (int i = 0; i < 7; i++)
      if (i != k) {
        double f = a[i * 9 + k];
        if (f != 0.0)
          for (int j = 0; j < 9; j++) a[i * 9 + j] -= f * a[k * 9 + j];
      }
  }
  double *out[2] = {f1, f2};
  for (int q = 0; q < 2; q++) {
    double v[9];
    for (int k = 0; k < 7; k++) v[k] = -a[k * 9 + 7 + q];
    v[7] = q == 0 ? 1.0 : 0.0, v[8] = q == 1 ? 1.0 : 0.0;
    for (int j = 0; j < 9; j++) out[q][colperm[j]] = v[j];
  }
}

int run7point(const float *m1, const float *m2, double *fmatrix) {  // fundam.cpp run7Point
  double a[63], f1[9], f2[9], c[4], r[3];
  for (int i = 0; i < 7; i++) {
    double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
    double *row = a + i * 9;
    row[0] = x1 * x0, row[1] = x1 * y0, row[2] = x1, row[3] = y1 * x0, row[4] = y1 * y0, row[5] = y1, row[6] = x0,
    row[7] = y0, row[8] = 1;
  }
  null_space_7x9(a, f1, f2);
  for (int i = 0; i < 9; i++) f1[i] -= f2[i];
  double t0 = f2[4] * f2[8] - f2[5] * f2[7], t1 = f2[3] * f2[8] - f2[5] * f2[6], t2 = f2[3] * f2[7] - f2[4] * f2[6];
  c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
  c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
         f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
         f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
         f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
  t0 = f1[4] * f1[8] - f1[5] * f1[7], t1 = f1[3] * f1[8] - f1[5] * f1[6], t2 = f1[3] * f1[7] - f1[4] * f1[6];
  c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
         f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
         f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
         f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
  c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
  int n = solve_cubic(c, r);
  if (n < 1 || n > 3) return n;
  for (int k = 0; k < n; k++, fmatrix += 9) {
    double lambda = r[k], mu = 1., s = f1[8] * r[k] + f2[8];
    if (fabs(s) > 2.220446049250313e-16) mu = 1. / s, lambda *= mu, fmatrix[8] = 1.;
    else fmatrix[8] = 0.;
    for (int i = 0; i < 8; i++) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
  }
  return n;
}

int find_inliers(const float *m1, const float *m2, int count, const double *F, double thresh, uint8_t *mask) {
  float t = (float)(thresh * thresh);
  int nz = 0;
  for (int i = 0; i < count; i++) {  // FMEstimatorCallback::computeError
    double a, b, c, d1, d2, s1, s2;
    double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
    a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
    s2 = 1. / (a * a + b * b);
    d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6], b = F[1] * x2 + F[4] * y2 + F[7], c = F[2] * x2 + F[5] * y2 + F[8];
    s1 = 1. / (a * a + b * b);
    d1 = x1 * a + y1 * b + c;
    float e = (float)std::max(d1 * d1 * s1, d2 * d2 * s2);
    int f = e <= t;
    mask[i] = (uint8_t)f, nz += f;
  }
  return nz;
}

int ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
  p = std::max(p, 0.), p = std::min(p, 1.), ep = std::max(ep, 0.), ep = std::min(ep, 1.);
  double num = std::max(1. - p, 2.2250738585072014e-308), denom = 1. - pow(1. - ep, model_points);
  if (denom < 2.2250738585072014e-308) return 0;
  num = log(num), denom = log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round(num / denom);
}

// FMEstimatorCallback::computeError for one correspondence (float, as the error Mat is CV_32F)
float epipolar_error(const double *F, float fx1, float fy1, float fx2, float fy2) {
  double a, b, c, d1, d2, s1, s2;
  double x1 = fx1, y1 = fy1, x2 = fx2, y2 = fy2;
  a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
  s2 = 1. / (a * a + b * b);
  d2 = x2 * a + y2 * b + c;
  a = F[0] * x2 + F[3] * y2 + F[6], b = F[1] * x2 + F[4] * y2 + F[7], c = F[2] * x2 + F[5] * y2 + F[8];
  s1 = 1. / (a * a + b * b);
  d1 = x1 * a + y1 * b + c;
  return (float)std::max(d1 * d1 * s1, d2 * d2 * s2);
}

// getSubset(..., maxAttempts = 10000), checkPartialSubsets == false (ptsetreg.cpp), shared by both registrators
bool get_subset(CvRng &rng, const float *m1, const float *m2, int count, float ms1[14], float ms2[14]) {
  const int model_points = 7, max_attempts = 10000;
  int idx[7], i = 0, iters = 0;
  for (; iters < max_attempts; iters++) {
    for (i = 0; i < model_points && iters < max_attempts;) {
      int idx_i = 0;
      for (;;) {
        idx_i = idx[i] = rng.uniform(0, count);
        int j;
        for (j = 0; j < i; j++)
          if (idx_i == idx[j]) break;
        if (j == i) break;
      }
      ms1[2 * i] = m1[2 * idx_i], ms1[2 * i + 1] = m1[2 * idx_i + 1];
      ms2[2 * i] = m2[2 * idx_i], ms2[2 * i + 1] = m2[2 * idx_i + 1];
      i++;
    }
    if (i == model_points && (have_collinear(ms1, i) || have_collinear(ms2, i))) continue;
    break;
  }
  return i == model_points && iters < max_attempts;
}

// LMeDSPointSetRegistrator::run (ptsetreg.cpp, OpenCV 3.0.0): outlierRatio 0.45, niters fixed up front (300 for
// confidence 0.99 and 7-point models), the model with the smallest median error wins (strictly smaller, so the first
// of equals), inliers = error <= sigma^2 with sigma = 2.5 * 1.4826 * (1 + 5 / (count - 7)) * sqrt(median).
bool fundamental_lmeds(const float *m1, const float *m2, int count, double confidence, uint8_t *out_mask) {
  const int model_points = 7, max_iters = 1000;
  const int niters = ransac_update_num_iters(confidence, 0.45, model_points, max_iters);
  CvRng rng((uint64_t)-1);
  double min_median = 1.7976931348623157e308, best[9];
  float ms1[14], ms2[14];
  std::vector<float> err(count);
  for (int iter = 0; iter < niters; iter++) {
    if (!get_subset(rng, m1, m2, count, ms1, ms2)) {
      if (iter == 0) {
        for (int q = 0; q < count; q++) out_mask[q] = 1;
        return false;
      }
      break;
    }
    double F[27];
    int nmodels = run7point(ms1, ms2, F);
    if (nmodels <= 0) continue;
    for (int k = 0; k < nmodels; k++) {
      for (int i = 0; i < count; i++) err[i] = epipolar_error(F + 9 * k, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]);
      // std::sort(errf.ptr<int>(), ...): the float bit patterns are ordered as integers
      std::vector<int32_t> bits(count);
      memcpy(bits.data(), err.data(), sizeof(float) * count);
      std::sort(bits.begin(), bits.end());
      memcpy(err.data(), bits.data(), sizeof(float) * count);
      double median = count % 2 != 0 ? err[count / 2] : (err[count / 2 - 1] + err[count / 2]) * 0.5;
      if (median < min_median) {
        min_median = median;
        memcpy(best, F + 9 * k, sizeof(best));
      }
    }
  }
  if (min_median < 1.7976931348623157e308) {
    double sigma = 2.5 * 1.4826 * (1 + 5. / (count - model_points)) * std::sqrt(min_median);
    sigma = std::max(sigma, 0.001);
    int good = find_inliers(m1, m2, count, best, sigma, out_mask);  // the mask is copied out before `result` is formed
    return good >= model_points;
  }
  for (int q = 0; q < count; q++) out_mask[q] = 1;  // no model at all: the reference would read an empty status vector
  return false;
}

// RANSACPointSetRegistrator::run with FMEstimatorCallback, modelPoints 7, maxIters 1000 (ptsetreg.cpp)
bool fundamental_ransac(const float *m1, const float *m2, int count, double threshold, double confidence, uint8_t *out_mask) {
  const int model_points = 7, max_iters = 1000;
  if (count < 15) {
    // findFundamentalMat (fundam.cpp, 3.0.0): "(method & ~3) == FM_RANSAC && npoints >= 15" else LMedS. The tracker only
    // calls with >= 8 points (feature_tracker.cpp:92,196); fewer than 8 keeps everything (7 would run the plain solver).
    if (count >= 8) return fundamental_lmeds(m1, m2, count, confidence, out_mask);
    for (int i = 0; i < count; i++) out_mask[i] = 1;
    return false;
  }
  CvRng rng((uint64_t)-1);
  int niters = max_iters, max_good = 0;
  std::vector<uint8_t> mask(count), best(count, 0);
  float ms1[14], ms2[14];
  for (int iter = 0; iter < niters; iter++) {
    // getSubset(..., maxAttempts = 10000), checkPartialSubsets == false
    int idx[7], i = 0, iters = 0;
    const int max_attempts = 10000;
    for (; iters < max_attempts; iters++) {
      for (i = 0; i < model_points && iters < max_attempts;) {
        int idx_i = 0;
        for (;;) {
          idx_i = idx[i] = rng.uniform(0, count);
          int j;
          for (j = 0; j < i; j++)
            if (idx_i == idx[j]) break;
          if (j == i) break;
        }
        ms1[2 * i] = m1[2 * idx_i], ms1[2 * i + 1] = m1[2 * idx_i + 1];
        ms2[2 * i] = m2[2 * idx_i], ms2[2 * i + 1] = m2[2 * idx_i + 1];
        i++;
      }
      if (i == model_points && (have_collinear(ms1, i) || have_collinear(ms2, i))) continue;
      break;
    }
    bool found = i == model_points && iters < max_attempts;
    if (!found) {
      if (iter == 0) {
        for (int q = 0; q < count; q++) out_mask[q] = 1;
        return false;
      }
      break;
    }
    double F[27];
    int nmodels = run7point(ms1, ms2, F);
    if (nmodels <= 0) continue;
    for (int k = 0; k < nmodels; k++) {
      int good = find_inliers(m1, m2, count, F + 9 * k, threshold, mask.data());
      if (good > std::max(max_good, model_points - 1)) {
        std::swap(mask, best);
        max_good = good;
        niters = ransac_update_num_iters(confidence, (double)(count - good) / count, model_points, niters);
      }
    }
  }
  if (max_good > 0) {
    memcpy(out_mask, best.data(), count);
    return true;
  }
  for (int q = 0; q < count; q++) out_mask[q] = 1;  // reference would read an empty status vector (UB); keep all
  return false;
}

// cv::circle(mask, center, radius, 0, -1): filled midpoint circle (imgproc/src/drawing.cpp Circle()), returned as
// the half-width of every scanline dy in [-r, r].
void circle_halfwidths(int radius, std::vector<int> &hw) {
  hw.assign(2 * radius + 1, -1);
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    // scanlines y +- dy span x +- dx ; scanlines y +- dx span x +- dy
    hw[radius + dy] = std::max(hw[radius + dy], dx), hw[radius - dy] = std::max(hw[radius - dy], dx);
    hw[radius + dx] = std::max(hw[radius + dx], dy), hw[radius - dx] = std::max(hw[radius - dx], dy);
    dy++;
    err += plus;
    plus += 2;
    int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

}  // namespace

struct oracle_tracker {
  VioConfig cfg;
  Image cur, pre, forw;
  bool have_img = false;
  std::vector<float> cur_pts, pre_pts, forw_pts;  // x y interleaved
  std::vector<int> ids, track_cnt;
  int n_id = 0;
  std::vector<int> hw;
};

namespace {

template <class T>
void reduce_vec(std::vector<T> &v, const std::vector<uint8_t> &st, int stride) {
  size_t j = 0;
  for (size_t i = 0; i < st.size(); i++)
    if (st[i]) {
      for (int k = 0; k < stride; k++) v[j * stride + k] = v[i * stride + k];
      j++;
    }
  v.resize(j * stride);
}

bool in_border(const VioConfig &cfg, float x, float y) {  // feature_tracker.cpp:18-24 (COL = cols, ROW = rows)
  int ix = cv_round(x), iy = cv_round(y);
  return 1 <= ix && ix < cfg.image_cols - 1 && 1 <= iy && iy < cfg.image_rows - 1;
}

}  // namespace

// The steps of readImage between the LK call and goodFeaturesToTrack (feature_tracker.cpp:183-205, and on publish
// frames :235-255 with setMask :50-87): status && inBorder, reduceVector, F-RANSAC on (cur, forw), rejectWithF on
// (pre, forw), track_cnt++, setMask. `mask` is filled on publish frames. Also the body of the isolated operator tests
// (oracle_tracker_update_tracks).
static void tracker_update_tracks(oracle_tracker_t *t, std::vector<uint8_t> &status, bool publish, int rows, int cols,
                                  std::vector<uint8_t> &mask) {
  const VioConfig &cfg = t->cfg;
  if (!t->cur_pts.empty()) {
    int n = (int)t->cur_pts.size() / 2;
    for (int i = 0; i < n; i++)
      if (status[i] && !in_border(cfg, t->forw_pts[2 * i], t->forw_pts[2 * i + 1])) status[i] = 0;
    reduce_vec(t->pre_pts, status, 2), reduce_vec(t->cur_pts, status, 2), reduce_vec(t->forw_pts, status, 2);
    reduce_vec(t->ids, status, 1), reduce_vec(t->track_cnt, status, 1);
    if (t->forw_pts.size() / 2 >= 8) {
      int m = (int)t->forw_pts.size() / 2;
      std::vector<uint8_t> st(m);
      fundamental_ransac(t->cur_pts.data(), t->forw_pts.data(), m, cfg.f_threshold, cfg.f_confidence, st.data());
      reduce_vec(t->cur_pts, st, 2), reduce_vec(t->pre_pts, st, 2), reduce_vec(t->forw_pts, st, 2);
      reduce_vec(t->ids, st, 1), reduce_vec(t->track_cnt, st, 1);
    }
  }
  if (!publish) return;
  // rejectWithF over the publish baseline (feature_tracker.cpp:89-103)
  if (t->forw_pts.size() / 2 >= 8) {
    int m = (int)t->forw_pts.size() / 2;
    std::vector<uint8_t> st(m);
    fundamental_ransac(t->pre_pts.data(), t->forw_pts.data(), m, cfg.f_threshold, cfg.f_confidence, st.data());
    reduce_vec(t->pre_pts, st, 2), reduce_vec(t->cur_pts, st, 2), reduce_vec(t->forw_pts, st, 2);
    reduce_vec(t->ids, st, 1), reduce_vec(t->track_cnt, st, 1);
  }
  for (auto &c : t->track_cnt) c++;
  // setMask (feature_tracker.cpp:50-87)
  mask.assign((size_t)rows * cols, 255);
  int n = (int)t->ids.size();
  std::vector<int> order(n);
  for (int i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return t->track_cnt[a] > t->track_cnt[b]; });
  std::vector<float> kp;
  std::vector<int> kid, kcnt;
  const int r = cfg.min_dist;
  for (int i : order) {
    float x = t->forw_pts[2 * i], y = t->forw_pts[2 * i + 1];
    int ix = cv_round(x), iy = cv_round(y);  // mask.at<uchar>(Point2f) rounds
    if (mask[(size_t)iy * cols + ix] == 255) {
      kp.push_back(x), kp.push_back(y), kid.push_back(t->ids[i]), kcnt.push_back(t->track_cnt[i]);
      for (int dy = -r; dy <= r; dy++) {  // cv::circle(mask, pt, MIN_DIST, 0, -1); center = Point(Point2f) rounds
        int yy = iy + dy;
        if (yy < 0 || yy >= rows) continue;
        int h = t->hw[r + dy];
        for (int xx = std::max(0, ix - h); xx <= std::min(cols - 1, ix + h); xx++) mask[(size_t)yy * cols + xx] = 0;
      }
    }
  }
  t->forw_pts = kp, t->ids = kid, t->track_cnt = kcnt;
}


extern "C" {

void oracle_set_lk_accum_mode(int mode) { g_lk_accum_mode = mode; }

static Image wrap(const uint8_t *p, int rows, int cols, int stride) {
  Image im;
  im.rows = rows, im.cols = cols;
  im.px.resize((size_t)rows * cols);
  for (int y = 0; y < rows; y++) memcpy(&im.px[(size_t)y * cols], p + (size_t)y * stride, cols);
  return im;
}

int oracle_pyr_down(const uint8_t *src, int32_t rows, int32_t cols, int32_t stride, uint8_t *dst) {
  Image s = wrap(src, rows, cols, stride), d;
  pyr_down(s, d);
  memcpy(dst, d.px.data(), d.px.size());
  return VIO_OK;
}

int oracle_klt_track(const VioConfig *cfg, const uint8_t *prev, const uint8_t *next, int32_t rows, int32_t cols,
                     int32_t stride, const float *prev_pts, int32_t n, float *next_pts, uint8_t *status, float *err) {
  Image a = wrap(prev, rows, cols, stride), b = wrap(next, rows, cols, stride);
  klt_track(cfg, a, b, prev_pts, n, next_pts, status, err);
  return VIO_OK;
}

int oracle_min_eigen_map(const uint8_t *img, int32_t rows, int32_t cols, int32_t stride, float *eig) {
  Image im = wrap(img, rows, cols, stride);
  std::vector<float> e;
  min_eigen_map(im, e);
  memcpy(eig, e.data(), e.size() * sizeof(float));
  return VIO_OK;
}

int oracle_good_features(const VioConfig *cfg, const uint8_t *img, const uint8_t *mask, int32_t rows, int32_t cols,
                         int32_t stride, int32_t max_corners, float *corners, int32_t *n_corners) {
  Image im = wrap(img, rows, cols, stride);
  *n_corners = good_features(im, mask, max_corners, cfg->quality_level, (double)cfg->min_dist, corners);
  return VIO_OK;
}

int oracle_fundamental_ransac(const VioConfig *cfg, const float *pts1, const float *pts2, int32_t n, uint8_t *inlier_mask) {
  fundamental_ransac(pts1, pts2, n, cfg->f_threshold, cfg->f_confidence, inlier_mask);
  return VIO_OK;
}

oracle_tracker_t *oracle_tracker_create(const VioConfig *cfg) {
  oracle_tracker *t = new oracle_tracker();
  t->cfg = *cfg;
  circle_halfwidths(cfg->min_dist, t->hw);
  return t;
}
void oracle_tracker_destroy(oracle_tracker_t *t) { delete t; }

// FeatureTracker::readImage (feature_tracker.cpp:162-310), without the UI outputs and the (default-off) vinsPnP.
int oracle_tracker_read_image(oracle_tracker_t *t, const uint8_t *gray, int32_t rows, int32_t cols, int32_t stride,
                              double header, int32_t publish, VioObs *out_obs, int32_t *n_obs) {
  (void)header;
  const VioConfig &cfg = t->cfg;
  if (rows != cfg.image_rows || cols != cfg.image_cols) return VIO_EINVAL;
  Image img = wrap(gray, rows, cols, stride);
  if (!t->have_img) t->pre = t->cur = t->forw = img, t->have_img = true;
  else t->forw = img;
  t->forw_pts.clear();
  std::vector<uint8_t> mask;
  if (!t->cur_pts.empty()) {
    int n = (int)t->cur_pts.size() / 2;
    std::vector<uint8_t> status(n);
    std::vector<float> err(n);
    t->forw_pts.resize(2 * n);
    klt_track(&cfg, t->cur, t->forw, t->cur_pts.data(), n, t->forw_pts.data(), status.data(), err.data());
    tracker_update_tracks(t, status, publish != 0, rows, cols, mask);
  } else if (publish) {
    std::vector<uint8_t> none;
    tracker_update_tracks(t, none, true, rows, cols, mask);
  }
  if (publish) {
    int n_max = cfg.max_corners - (int)t->ids.size();
    std::vector<float> npts;
    if (n_max > 0) {
      npts.resize(2 * n_max);
      int nc = good_features(t->forw, mask.data(), n_max, cfg.quality_level, (double)cfg.min_dist, npts.data());
      npts.resize(2 * nc);
    }
    for (size_t i = 0; i < npts.size() / 2; i++) {  // addPoints
      t->forw_pts.push_back(npts[2 * i]), t->forw_pts.push_back(npts[2 * i + 1]);
      t->ids.push_back(-1), t->track_cnt.push_back(1);
    }
    t->pre = t->forw;
    t->pre_pts = t->forw_pts;
  }
  t->cur = t->forw;
  t->cur_pts = t->forw_pts;
  *n_obs = 0;
  if (publish) {
    for (size_t i = 0; i < t->ids.size(); i++)
      if (t->ids[i] == -1) t->ids[i] = t->n_id++;  // updateID
    for (size_t i = 0; i < t->ids.size(); i++) {
      out_obs[i].id = t->ids[i];
      out_obs[i].x = (t->cur_pts[2 * i] - cfg.cx) / cfg.fx;
      out_obs[i].y = (t->cur_pts[2 * i + 1] - cfg.cy) / cfg.fy;
      out_obs[i].z = 1.0;
    }
    *n_obs = (int)t->ids.size();
  }
  return VIO_OK;
}

int oracle_tracker_get_state(oracle_tracker_t *t, float *cur_pts, int32_t *ids, int32_t *track_cnt, int32_t cap, int32_t *n) {
  int m = (int)t->ids.size();
  *n = m;
  if (m > cap) return VIO_ECAP;
  memcpy(cur_pts, t->cur_pts.data(), sizeof(float) * 2 * m);
  memcpy(ids, t->ids.data(), sizeof(int) * m), memcpy(track_cnt, t->track_cnt.data(), sizeof(int) * m);
  return VIO_OK;
}

// Public fields of the tracker (feature_tracker.hpp:68-80) in / out, and the update step on its own: the isolated
// operator tests of rejectWithF (F6) and setMask (F7).
int oracle_tracker_set_tracks(oracle_tracker_t *t, int32_t n, const float *pre_pts, const float *cur_pts, const float *forw_pts,
                              const int32_t *ids, const int32_t *track_cnt) {
  t->pre_pts.assign(pre_pts, pre_pts + 2 * n), t->cur_pts.assign(cur_pts, cur_pts + 2 * n);
  t->forw_pts.assign(forw_pts, forw_pts + 2 * n);
  t->ids.assign(ids, ids + n), t->track_cnt.assign(track_cnt, track_cnt + n);
  return VIO_OK;
}
int oracle_tracker_update_tracks(oracle_tracker_t *t, const uint8_t *lk_status, int32_t publish) {
  std::vector<uint8_t> status(lk_status, lk_status + t->cur_pts.size() / 2), mask;
  tracker_update_tracks(t, status, publish != 0, t->cfg.image_rows, t->cfg.image_cols, mask);
  return VIO_OK;
}
int oracle_tracker_get_tracks(oracle_tracker_t *t, float *forw_pts, int32_t *ids, int32_t *track_cnt, int32_t cap, int32_t *n) {
  int m = (int)t->ids.size();
  *n = m;
  if (m > cap) return VIO_ECAP;
  memcpy(forw_pts, t->forw_pts.data(), sizeof(float) * 2 * m);
  memcpy(ids, t->ids.data(), sizeof(int) * m), memcpy(track_cnt, t->track_cnt.data(), sizeof(int) * m);
  return VIO_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// The image pre-step of the camera callback: cv::cvtColor(CV_RGBA2GRAY) + cv::CLAHE(clipLimit 3, 8x8)
// (VINS_ios/ViewController.mm:432-437). OpenCV is a third-party binary the reference links (opencv2.framework, 3.x), not
// in /root/reference: this restates the PUBLISHED algorithm (imgproc color.cpp RGB2Gray<uchar>; imgproc clahe.cpp
// CLAHE_CalcLut_Body / CLAHE_Interpolation_Body) with plain loops over whole images — parity unpinned (no OpenCV here).
#include <vector>
extern "C" int oracle_preprocess(const uint8_t *pixels, int32_t channels, int32_t rows, int32_t cols, int32_t stride,
                                 double clip_limit, int32_t tiles_x, int32_t tiles_y, uint8_t *gray_out,
                                 uint8_t *equalized_out) {
  if (!pixels || !equalized_out || !(channels == 1 || channels == 4) || rows < 1 || cols < 1 || tiles_x < 1 || tiles_y < 1)
    return VIO_EINVAL;
  std::vector<uint8_t> gray((size_t)rows * cols);
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      const uint8_t *p = pixels + (size_t)y * stride + (size_t)channels * x;
      gray[(size_t)y * cols + x] = channels == 1 ? p[0] : (uint8_t)((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + (1 << 13)) >> 14);
    }
  if (gray_out) memcpy(gray_out, gray.data(), gray.size());
  // the image the histograms are taken from: the source, or its reflect-101 extension to a multiple of the grid
  int ext_r = rows, ext_c = cols;
  if (cols % tiles_x != 0 || rows % tiles_y != 0) ext_c = cols + (tiles_x - cols % tiles_x), ext_r = rows + (tiles_y - rows % tiles_y);
  std::vector<uint8_t> ext((size_t)ext_r * ext_c);
  for (int y = 0; y < ext_r; y++)
    for (int x = 0; x < ext_c; x++) {
      const int sy = y < rows ? y : 2 * rows - 2 - y, sx = x < cols ? x : 2 * cols - 2 - x;
      if (sy < 0 || sx < 0) return VIO_EINVAL;
      ext[(size_t)y * ext_c + x] = gray[(size_t)sy * cols + sx];
    }
  const int tw = ext_c / tiles_x, th = ext_r / tiles_y, area = tw * th;
  const float lut_scale = (float)(256 - 1) / area;
  int clip = 0;
  if (clip_limit > 0.0) {
    clip = (int)(clip_limit * area / 256);
    if (clip < 1) clip = 1;
  }
  std::vector<uint8_t> lut((size_t)tiles_x * tiles_y * 256);
  for (int k = 0; k < tiles_x * tiles_y; k++) {
    const int ty = k / tiles_x, tx = k % tiles_x;
    int hist[256] = {0};
    for (int y = 0; y < th; y++)
      for (int x = 0; x < tw; x++) hist[ext[(size_t)(ty * th + y) * ext_c + tx * tw + x]]++;
    if (clip > 0) {
      int clipped = 0;
      for (int i = 0; i < 256; i++)
        if (hist[i] > clip) clipped += hist[i] - clip, hist[i] = clip;
      const int batch = clipped / 256;
      int residual = clipped - batch * 256;
      for (int i = 0; i < 256; i++) hist[i] += batch;
      if (residual != 0) {
        const int step = 256 / residual > 1 ? 256 / residual : 1;
        for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
      }
    }
    int sum = 0;
    for (int i = 0; i < 256; i++) {
      sum += hist[i];
      long r = lrintf(sum * lut_scale);
      lut[(size_t)k * 256 + i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
  }
  const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
  for (int y = 0; y < rows; y++) {
    const float tyf = y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    const float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = ty1 < 0 ? 0 : ty1, ty2 = ty2 > tiles_y - 1 ? tiles_y - 1 : ty2;
    const uint8_t *p1 = &lut[(size_t)ty1 * tiles_x * 256], *p2 = &lut[(size_t)ty2 * tiles_x * 256];
    for (int x = 0; x < cols; x++) {
      const float txf = x * inv_tw - 0.5f;
      int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
      const float xa = txf - tx1, xa1 = 1.0f - xa;
      tx1 = tx1 < 0 ? 0 : tx1, tx2 = tx2 > tiles_x - 1 ? tiles_x - 1 : tx2;
      const int v = gray[(size_t)y * cols + x], i1 = tx1 * 256 + v, i2 = tx2 * 256 + v;
      const float res = (p1[i1] * xa1 + p1[i2] * xa) * ya1 + (p2[i1] * xa1 + p2[i2] * xa) * ya;
      long r = lrintf(res);
      equalized_out[(size_t)y * cols + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
  }
  return VIO_OK;
}

// ---- loop-closure producer, descriptor side (test infrastructure like everything in oracle/) ----------------------
// KeyFrame::HammingDis (VINS_ios/loop/keyframe.cpp:368-373): popcount of the xor of two 256-bit strings.
static int hamming256(const uint64_t *a, const uint64_t *b) {
  int d = 0;
  for (int w = 0; w < 4; w++) {
    uint64_t x = a[w] ^ b[w];
    while (x) d++, x &= x - 1;
  }
  return d;
}

extern "C" int oracle_search_by_des(const uint64_t *cur_desc, int32_t n_cur, const uint64_t *old_desc, int32_t n_old,
                                    int32_t *best_index, int32_t *best_dist) {
  // keyframe.cpp:167-187: for every window descriptor the first old descriptor of minimum distance
  for (int i = 0; i < n_cur; i++) {
    int bestDist = 256, bestIndex = -1;
    for (int j = 0; j < n_old; j++) {
      int dis = hamming256(cur_desc + 4 * (size_t)i, old_desc + 4 * (size_t)j);
      if (dis < bestDist) bestDist = dis, bestIndex = j;
    }
    best_index[i] = bestDist < 256 ? bestIndex : -1;
    best_dist[i] = bestDist;
  }
  return 0;
}

extern "C" int oracle_loop_find_connection(const VioConfig *cfg, int32_t n_cur, const uint64_t *cur_desc, const float *cur_pts,
                                           int32_t n_old, const uint64_t *old_desc, const float *old_pts,
                                           float *matched_old_pts, float *matched_old_norm, uint8_t *status,
                                           int32_t *n_inliers) {
  *n_inliers = 0;
  if (n_cur == 0) return 0;
  std::vector<int32_t> idx(n_cur), dist(n_cur);
  oracle_search_by_des(cur_desc, n_cur, old_desc, n_old, idx.data(), dist.data());
  for (int i = 0; i < n_cur; i++) {
    if (idx[i] < 0) {
      for (int k = 0; k < n_cur; k++) status[k] = 0;
      return 0;
    }
    matched_old_pts[2 * i] = old_pts[2 * idx[i]], matched_old_pts[2 * i + 1] = old_pts[2 * idx[i] + 1];
  }
  if (n_cur >= 8) {  // rejectWithF, keyframe.cpp:35-58
    if (matched_old_norm)
      for (int i = 0; i < n_cur; i++) {
        matched_old_norm[2 * i] = (float)((matched_old_pts[2 * i] - (float)cfg->cx) / (float)cfg->fx);
        matched_old_norm[2 * i + 1] = (float)((matched_old_pts[2 * i + 1] - (float)cfg->cy) / (float)cfg->fy);
      }
    VioConfig c = *cfg;
    c.f_threshold = 2.0, c.f_confidence = 0.99;
    int rc = oracle_fundamental_ransac(&c, cur_pts, matched_old_pts, n_cur, status);
    if (rc != 0) return rc;
  } else {
    for (int i = 0; i < n_cur; i++) status[i] = 1;
  }
  int k = 0;
  for (int i = 0; i < n_cur; i++) k += status[i] ? 1 : 0;
  *n_inliers = k;
  return 0;
}
