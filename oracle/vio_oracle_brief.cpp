// oracle/vio_oracle_brief.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Plain C++ restatement of the keyframe descriptor extraction of the loop-closure producer,
// BriefExtractor::operator() (VINS_ios/loop/keyframe.cpp:395-409):
//     cv::FAST(im, keys, 20, true);  keys += window_pts;  m_brief.compute(im, keys, descriptors);
//   * DVision::BRIEF::compute (ThirdParty/DVision/BRIEF.cpp:40-105, in the tree): cv::GaussianBlur(9x9, sigma 2) and 256
//     pairwise intensity tests at (int)(pt + offset), skipped (bit stays 0) when either end leaves the image; the test
//     pattern is the app's Resources/brief_pattern.yml (keyframe.cpp:375-393).
//   * cv::FAST and cv::GaussianBlur are OpenCV ("customized 3.0.0", a binary that is not in the tree). PARITY UNPINNED:
//     they follow the published OpenCV 3.0.0 algorithms — features2d/fast.cpp FAST_t<16> with fast_score.cpp
//     cornerScore<16> and the 3x3 non-maximum rule; imgproc/smooth.cpp GaussianBlur -> the 8-bit fixed-point separable
//     filter of filter.cpp (kernel getGaussianKernel(9, 2) in float, scaled by 2^8 and rounded per tap, row pass in
//     int, column pass (sum + 2^15) >> 16, BORDER_REFLECT_101).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "vio_amd.h"
#include "vio_oracle.h"

namespace {

int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// getGaussianKernel(9, 2.0, CV_32F) -> convertTo(CV_32S, 256): cvRound of each tap
void gauss9_taps(int taps[9]) {
  const int n = 9;
  const double sigma = 2.0, scale2X = -0.5 / (sigma * sigma);
  float cf[9];
  double sum = 0;
  for (int i = 0; i < n; i++) {
    const double x = i - (n - 1) * 0.5;
    const double t = exp(scale2X * x * x);
    cf[i] = (float)t;
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; i++) {
    cf[i] = (float)(cf[i] * sum);
    taps[i] = (int)lrint((double)cf[i] * 256.0);  // saturate_cast<int>(double): round half to even
  }
}

const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// fast_score.cpp cornerScore<16>
int corner_score16(const uint8_t *img, int cols, int x, int y, int threshold) {
  const int K = 8, N = K * 3 + 1;
  const int v = img[y * cols + x];
  int d[N];
  for (int k = 0; k < N; k++) d[k] = v - img[(y + kCircle[k & 15][1]) * cols + x + kCircle[k & 15][0]];
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min(d[k + 1], d[k + 2]);
    a = std::min(a, d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, d[k + 4]), a = std::min(a, d[k + 5]), a = std::min(a, d[k + 6]), a = std::min(a, d[k + 7]);
    a = std::min(a, d[k + 8]);
    a0 = std::max(a0, std::min(a, d[k]));
    a0 = std::max(a0, std::min(a, d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max(d[k + 1], d[k + 2]);
    b = std::max(b, d[k + 3]), b = std::max(b, d[k + 4]), b = std::max(b, d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, d[k + 6]), b = std::max(b, d[k + 7]), b = std::max(b, d[k + 8]);
    b0 = std::min(b0, std::max(b, d[k]));
    b0 = std::min(b0, std::max(b, d[k + 9]));
  }
  return -b0 - 1;
}

}  // namespace

extern "C" {

// cv::GaussianBlur(src, dst, Size(9, 9), 2, 2) on 8-bit gray
int oracle_gaussian_blur9(const uint8_t *src, int rows, int cols, uint8_t *dst) {
  if (!src || !dst || rows < 1 || cols < 1) return VIO_EINVAL;
  int taps[9];
  gauss9_taps(taps);
  std::vector<int> tmp((size_t)rows * cols);
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      int s = 0;
      for (int k = 0; k < 9; k++) s += taps[k] * src[y * cols + reflect101(x + k - 4, cols)];
      tmp[(size_t)y * cols + x] = s;
    }
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      int s = 0;
      for (int k = 0; k < 9; k++) s += taps[k] * tmp[(size_t)reflect101(y + k - 4, rows) * cols + x];
      s = (s + (1 << 15)) >> 16;
      dst[(size_t)y * cols + x] = (uint8_t)std::min(std::max(s, 0), 255);
    }
  return VIO_OK;
}

// cv::FAST(img, keys, threshold, nonmaxSuppression = true), TYPE_9_16. keypoints: (x, y) in emission order (rows, then
// columns); returns the number found in *n (all of them are counted, at most cap are stored).
int oracle_fast9_16(const uint8_t *img, int rows, int cols, int threshold, float *keypoints, int cap, int *n) {
  if (!img || !n || rows < 7 || cols < 7) return VIO_EINVAL;
  threshold = std::min(std::max(threshold, 0), 255);
  std::vector<uint8_t> score((size_t)rows * cols, 0);
  const int K = 8, N = 25;
  for (int i = 3; i < rows - 3; i++)
    for (int j = 3; j < cols - 3; j++) {
      const int v = img[i * cols + j];
      bool corner = false;
      for (int pass = 0; pass < 2 && !corner; pass++) {  // darker arc (x < v - t), then brighter arc (x > v + t)
        int count = 0;
        for (int k = 0; k < N; k++) {
          const int x = img[(i + kCircle[k & 15][1]) * cols + j + kCircle[k & 15][0]];
          const bool hit = pass == 0 ? x < v - threshold : x > v + threshold;
          if (hit) {
            if (++count > K) {
              corner = true;
              break;
            }
          } else {
            count = 0;
          }
        }
      }
      if (corner) score[(size_t)i * cols + j] = (uint8_t)corner_score16(img, cols, j, i, threshold);
    }
  int c = 0;
  for (int i = 3; i < rows - 3; i++)
    for (int j = 3; j < cols - 3; j++) {
      const int s = score[(size_t)i * cols + j];
      if (!s) continue;
      const uint8_t *p = &score[(size_t)i * cols + j];
      if (s > p[1] && s > p[-1] && s > p[-cols - 1] && s > p[-cols] && s > p[-cols + 1] && s > p[cols - 1] && s > p[cols] &&
          s > p[cols + 1]) {
        if (c < cap && keypoints) keypoints[2 * c] = (float)j, keypoints[2 * c + 1] = (float)i;
        c++;
      }
    }
  *n = c;
  return VIO_OK;
}

// DVision::BRIEF::compute on an already blurred image (treat_image = false part): descriptors [n][4] words, bit i of the
// descriptor = bit (i & 63) of word i >> 6
int oracle_brief_compute(const uint8_t *blurred, int rows, int cols, const float *pts, int n, const int32_t *x1, const int32_t *y1,
                         const int32_t *x2, const int32_t *y2, int n_bits, uint64_t *desc) {
  if (!blurred || (n > 0 && (!pts || !desc)) || !x1 || !y1 || !x2 || !y2 || n_bits < 1 || n_bits > 256) return VIO_EINVAL;
  for (int p = 0; p < n; p++) {
    uint64_t w[4] = {0, 0, 0, 0};
    for (int i = 0; i < n_bits; i++) {
      const int ax = (int)(pts[2 * p] + x1[i]), ay = (int)(pts[2 * p + 1] + y1[i]);
      const int bx = (int)(pts[2 * p] + x2[i]), by = (int)(pts[2 * p + 1] + y2[i]);
      if (ax >= 0 && ax < cols && ay >= 0 && ay < rows && bx >= 0 && bx < cols && by >= 0 && by < rows)
        if (blurred[(size_t)ay * cols + ax] < blurred[(size_t)by * cols + bx]) w[i >> 6] |= 1ull << (i & 63);
    }
    memcpy(desc + 4 * (size_t)p, w, sizeof(w));
  }
  return VIO_OK;
}

// BriefExtractor::operator(): FAST keypoints, then the window points; descriptors of all of them on the blurred image
int oracle_brief_extract(const uint8_t *gray, int rows, int cols, const float *window_pts, int n_window, int fast_threshold,
                         const int32_t *x1, const int32_t *y1, const int32_t *x2, const int32_t *y2, int n_bits, int cap,
                         float *keypoints, uint64_t *desc, int *n_fast, int *n_keypoints) {
  if (!gray || !keypoints || !desc || !n_fast || !n_keypoints || n_window < 0 || cap < n_window) return VIO_EINVAL;
  int nf = 0;
  int rc = oracle_fast9_16(gray, rows, cols, fast_threshold, keypoints, cap - n_window, &nf);
  if (rc != VIO_OK) return rc;
  const int stored = std::min(nf, cap - n_window);
  for (int i = 0; i < n_window; i++) keypoints[2 * (stored + i)] = window_pts[2 * i], keypoints[2 * (stored + i) + 1] = window_pts[2 * i + 1];
  std::vector<uint8_t> blur((size_t)rows * cols);
  oracle_gaussian_blur9(gray, rows, cols, blur.data());
  rc = oracle_brief_compute(blur.data(), rows, cols, keypoints, stored + n_window, x1, y1, x2, y2, n_bits, desc);
  *n_fast = nf, *n_keypoints = stored + n_window;
  return rc != VIO_OK ? rc : (nf > stored ? VIO_ECAP : VIO_OK);
}

}  // extern "C"
