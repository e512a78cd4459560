// vio_brief.hip — keyframe descriptor extraction of the loop-closure producer:
// BriefExtractor::operator() (VINS_ios/loop/keyframe.cpp:395-409) for a batch of keyframes,
//     cv::FAST(im, keys, 20, true);  keys += window_pts;  m_brief.compute(im, keys, descriptors);
// where DVision::BRIEF::compute (ThirdParty/DVision/BRIEF.cpp:40-105) blurs the image (GaussianBlur 9x9, sigma 2) and
// makes 256 intensity comparisons per keypoint with the app's test pattern (Resources/brief_pattern.yml).
// All of it is integer work on bytes — HBM-bound streaming kernels, bit-exact against the restatement in oracle/:
//   blur9_kernel       64x16 output tile per workgroup; the 72x24 source patch goes through LDS once, row pass in int,
//                      column pass (sum + 2^15) >> 16 — OpenCV's 8-bit fixed-point separable filter (taps scaled by 2^8)
//   fast_score_kernel  FAST-9/16 segment test + cornerScore per pixel (two opposite-pixel quick rejections first,
//                      like fast.cpp): one byte per pixel, 0 = no corner
//   fast_collect_kernel one workgroup per frame walks the rows in order, 16 rows (one per wave) at a time: 3x3
//                      non-maximum test, ballot + popcount compaction -> keypoints in cv::FAST's emission order; the
//                      window points are appended behind them
//   brief_kernel       one wave per keypoint: lane l makes tests l, l + 64, l + 128, l + 192; four ballots are the
//                      four descriptor words
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

#include "vio_amd.h"
#include "vio_device.h"

namespace {

#define HIP_OK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "vio_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VIO_ENODEV;                                                                   \
    }                                                                                      \
  } while (0)

struct Taps9 {
  int t[9];
};

__device__ __forceinline__ int reflect101_d(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

constexpr int kBlW = 64, kBlH = 16;
__global__ __launch_bounds__(256) void blur9_kernel(const uint8_t *src, uint8_t *dst, int rows, int cols, Taps9 T) {
  __shared__ uint8_t s_src[kBlH + 8][kBlW + 8];
  __shared__ int s_row[kBlH + 8][kBlW];
  const size_t frame = (size_t)blockIdx.z * rows * cols;
  const int x0 = blockIdx.x * kBlW, y0 = blockIdx.y * kBlH, tid = threadIdx.x;
  for (int q = tid; q < (kBlH + 8) * (kBlW + 8); q += 256) {
    const int ly = q / (kBlW + 8), lx = q - ly * (kBlW + 8);
    // (rows / columns of the tile that lie past the image are never used by a stored output; clamp their index)
    const int y = reflect101_d(min(y0 + ly - 4, 2 * rows - 2), rows), x = reflect101_d(min(x0 + lx - 4, 2 * cols - 2), cols);
    s_src[ly][lx] = src[frame + (size_t)y * cols + x];
  }
  __syncthreads();
  for (int q = tid; q < (kBlH + 8) * kBlW; q += 256) {
    const int ly = q / kBlW, lx = q - ly * kBlW;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) s += T.t[k] * s_src[ly][lx + k];
    s_row[ly][lx] = s;
  }
  __syncthreads();
  for (int q = tid; q < kBlH * kBlW; q += 256) {
    const int ly = q / kBlW, lx = q - ly * kBlW, x = x0 + lx, y = y0 + ly;
    if (x >= cols || y >= rows) continue;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) s += T.t[k] * s_row[ly + k][lx];
    s = (s + (1 << 15)) >> 16;
    dst[frame + (size_t)y * cols + x] = (uint8_t)min(max(s, 0), 255);
  }
}

__constant__ int c_circle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

__global__ __launch_bounds__(256) void fast_score_kernel(const uint8_t *img, uint8_t *score, int rows, int cols, int threshold) {
  const size_t frame = (size_t)blockIdx.z * rows * cols;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= rows || j >= cols) return;
  uint8_t out = 0;
  if (i >= 3 && i < rows - 3 && j >= 3 && j < cols - 3) {
    const uint8_t *p = img + frame + (size_t)i * cols + j;
    const int v = p[0];
    int px[16];
#pragma unroll
    for (int k = 0; k < 16; k++) px[k] = p[c_circle[k][1] * cols + c_circle[k][0]];
    // a run of 9 of the 16 contains one pixel of every opposite pair: quick rejection on the pairs (0,8) and (4,12)
    const int lo = v - threshold, hi = v + threshold;
    auto cls = [&](int x) { return x < lo ? 1 : (x > hi ? 2 : 0); };
    int d = (cls(px[0]) | cls(px[8])) & (cls(px[4]) | cls(px[12]));
    if (d) {
      bool corner = false;
#pragma unroll
      for (int pass = 0; pass < 2; pass++) {
        int count = 0;
        bool found = false;
#pragma unroll
        for (int k = 0; k < 25; k++) {
          const int x = px[k & 15];
          const bool hit = pass == 0 ? x < lo : x > hi;
          count = hit ? count + 1 : 0;
          found |= count > 8;
        }
        corner |= found;
      }
      if (corner) {  // fast_score.cpp cornerScore<16>
        int dd[25];
#pragma unroll
        for (int k = 0; k < 25; k++) dd[k] = v - px[k & 15];
        int a0 = threshold;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          int a = min(min(dd[k + 1], dd[k + 2]), dd[k + 3]);
          if (a <= a0) continue;
          a = min(a, dd[k + 4]), a = min(a, dd[k + 5]), a = min(a, dd[k + 6]), a = min(a, dd[k + 7]), a = min(a, dd[k + 8]);
          a0 = max(a0, min(a, dd[k]));
          a0 = max(a0, min(a, dd[k + 9]));
        }
        int b0 = -a0;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          int b = max(max(dd[k + 1], dd[k + 2]), dd[k + 3]);
          b = max(b, dd[k + 4]), b = max(b, dd[k + 5]);
          if (b >= b0) continue;
          b = max(b, dd[k + 6]), b = max(b, dd[k + 7]), b = max(b, dd[k + 8]);
          b0 = min(b0, max(b, dd[k]));
          b0 = min(b0, max(b, dd[k + 9]));
        }
        out = (uint8_t)(-b0 - 1);
      }
    }
  }
  score[frame + (size_t)i * cols + j] = out;
}

constexpr int kColThreads = 1024;
constexpr int kBriefMaxCols = 2048;  // widest image (the per-row corner lists of the collector live in LDS)
// One workgroup per frame. Rows are taken 16 at a time (one per wave); a wave compacts the kept corners of its row into
// its LDS list (ballot + popcount: column order), then the 16 lists are copied out behind each other: the emission order
// of cv::FAST (row by row, columns ascending). n_fast counts every corner, stored are at most cap - n_window.
__global__ __launch_bounds__(kColThreads) void fast_collect_kernel(const uint8_t *score, int rows, int cols, const float *window_pts,
                                                                   const int *n_window, int window_stride, int cap, float *keypoints,
                                                                   int *n_fast, int *n_keypoints) {
  constexpr int kWaves = kColThreads / 64, kRowCap = kBriefMaxCols / 2;  // 3x3 non-maxima: at most every other pixel of a row
  __shared__ unsigned short s_cols[kWaves][kRowCap];
  __shared__ int s_cnt[kWaves];
  const int f = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint8_t *S = score + (size_t)f * rows * cols;
  const int nw = n_window[f], room = cap - nw;
  float *kp = keypoints + (size_t)f * cap * 2;
  int base = 0;
  for (int r0 = 3; r0 < rows - 3; r0 += kWaves) {
    const int i = r0 + wave;
    int cnt = 0;
    if (i < rows - 3) {
      for (int j0 = 3; j0 < cols - 3; j0 += 64) {
        const int j = j0 + lane;
        bool keep = false;
        if (j < cols - 3) {
          const uint8_t *p = S + (size_t)i * cols + j;
          const int s = p[0];
          keep = s && s > p[1] && s > p[-1] && s > p[-cols - 1] && s > p[-cols] && s > p[-cols + 1] && s > p[cols - 1] && s > p[cols] &&
                 s > p[cols + 1];
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) {
          const int at = cnt + __builtin_popcountll(m & ((1ull << lane) - 1));
          if (at < kRowCap) s_cols[wave][at] = (unsigned short)j;
        }
        cnt += __builtin_popcountll(m);
      }
    }
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    int off = base, total = 0;
    for (int w = 0; w < kWaves; w++) {
      if (w < wave) off += s_cnt[w];
      total += s_cnt[w];
    }
    for (int q = lane; q < min(cnt, kRowCap); q += 64) {
      const int at = off + q;
      if (at < room) kp[2 * at] = (float)s_cols[wave][q], kp[2 * at + 1] = (float)i;
    }
    base += total;
    __syncthreads();
  }
  const int stored = min(base, max(room, 0));
  for (int q = tid; q < nw; q += kColThreads) {
    kp[2 * (stored + q)] = window_pts[((size_t)f * window_stride + q) * 2];
    kp[2 * (stored + q) + 1] = window_pts[((size_t)f * window_stride + q) * 2 + 1];
  }
  if (tid == 0) n_fast[f] = base, n_keypoints[f] = stored + nw;
}

struct Pattern {
  const int *x1, *y1, *x2, *y2;
  int n_bits;
};

__global__ __launch_bounds__(256) void brief_kernel(const uint8_t *blurred, int rows, int cols, const float *keypoints, const int *n_keypoints,
                                                    int cap, Pattern P, unsigned long long *desc) {
  const int f = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + wave;
  if (k >= n_keypoints[f]) return;
  const uint8_t *im = blurred + (size_t)f * rows * cols;
  const float px = keypoints[((size_t)f * cap + k) * 2], py = keypoints[((size_t)f * cap + k) * 2 + 1];
  unsigned long long w[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = 64 * q + lane;
    bool bit = false;
    if (i < P.n_bits) {
      const int ax = (int)(px + (float)P.x1[i]), ay = (int)(py + (float)P.y1[i]);
      const int bx = (int)(px + (float)P.x2[i]), by = (int)(py + (float)P.y2[i]);
      if (ax >= 0 && ax < cols && ay >= 0 && ay < rows && bx >= 0 && bx < cols && by >= 0 && by < rows)
        bit = im[(size_t)ay * cols + ax] < im[(size_t)by * cols + bx];
    }
    w[q] = __builtin_amdgcn_ballot_w64(bit);
  }
  if (lane < 4) desc[((size_t)f * cap + k) * 4 + lane] = w[lane];
}

}  // namespace

struct vio_brief {
  int device = 0;
  int rows = 0, cols = 0, max_frames = 0, cap = 0, n_bits = 0;
  Taps9 taps;
  hipStream_t stream = nullptr;
  uint8_t *d_img = nullptr, *d_blur = nullptr, *d_score = nullptr;
  float *d_kp = nullptr, *d_wpts = nullptr;
  unsigned long long *d_desc = nullptr;
  int *d_nw = nullptr, *d_nfast = nullptr, *d_nkp = nullptr, *d_pat = nullptr;
};

extern "C" {

// OpenCV FileStorage YAML 1.0 as written for Resources/brief_pattern.yml: top-level keys x1, y1, x2, y2, each a block
// sequence of integers ("  - 12") or a flow sequence ("[ 1, 2 ]").
int vio_brief_load_pattern(const char *yml_path, int32_t *x1, int32_t *y1, int32_t *x2, int32_t *y2, int32_t cap, int32_t *n) {
  if (!yml_path || !x1 || !y1 || !x2 || !y2 || !n || cap < 1) return VIO_EINVAL;
  FILE *fp = fopen(yml_path, "r");
  if (!fp) return VIO_EINVAL;
  int32_t *dst = nullptr;
  int cnt[4] = {0, 0, 0, 0}, which = -1;
  char line[4096];
  int rc = VIO_OK;
  while (fgets(line, sizeof(line), fp)) {
    const char *s = line;
    if (line[0] != ' ' && line[0] != '-' && line[0] != '\t') {  // a key line
      which = -1, dst = nullptr;
      if (!strncmp(line, "x1:", 3)) which = 0, dst = x1;
      else if (!strncmp(line, "y1:", 3)) which = 1, dst = y1;
      else if (!strncmp(line, "x2:", 3)) which = 2, dst = x2;
      else if (!strncmp(line, "y2:", 3)) which = 3, dst = y2;
      if (which < 0) continue;
      s = line + 3;
    }
    if (which < 0) continue;
    while (*s) {  // every integer on the line
      while (*s && !(*s == '-' || (*s >= '0' && *s <= '9'))) s++;
      if (!*s) break;
      if (*s == '-' && !(s[1] >= '0' && s[1] <= '9')) {  // the "- " of a block sequence entry
        s++;
        continue;
      }
      char *end = nullptr;
      const long v = strtol(s, &end, 10);
      if (end == s) break;
      if (cnt[which] >= cap) {
        rc = VIO_ECAP;
        break;
      }
      dst[cnt[which]++] = (int32_t)v;
      s = end;
    }
    if (rc != VIO_OK) break;
  }
  fclose(fp);
  if (rc != VIO_OK) return rc;
  if (cnt[0] < 1 || cnt[0] != cnt[1] || cnt[0] != cnt[2] || cnt[0] != cnt[3]) return VIO_EINVAL;
  *n = cnt[0];
  return VIO_OK;
}

int vio_brief_create(int32_t rows, int32_t cols, int32_t max_frames, int32_t max_keypoints, const int32_t *x1, const int32_t *y1,
                     const int32_t *x2, const int32_t *y2, int32_t n_bits, vio_brief_t **out) {
  if (!out || rows < 7 || cols < 7 || cols > kBriefMaxCols || max_frames < 1 || max_keypoints < 1 || !x1 || !y1 || !x2 || !y2 || n_bits < 1 ||
      n_bits > 256)
    return VIO_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    fprintf(stderr, "vio_amd: no HIP device visible; the descriptor extraction has no CPU fallback\n");
    return VIO_ENODEV;
  }
  vio_brief *b = new (std::nothrow) vio_brief();
  if (!b) return VIO_ENOMEM;
  b->device = vio::current_device();
  b->rows = rows, b->cols = cols, b->max_frames = max_frames, b->cap = max_keypoints, b->n_bits = n_bits;
  {  // getGaussianKernel(9, 2, CV_32F) scaled by 2^8 and rounded per tap (filter.cpp, 8-bit fixed-point path)
    const double sigma = 2.0, scale2X = -0.5 / (sigma * sigma);
    float cf[9];
    double sum = 0;
    for (int i = 0; i < 9; i++) cf[i] = (float)exp(scale2X * (i - 4.0) * (i - 4.0)), sum += cf[i];
    sum = 1. / sum;
    for (int i = 0; i < 9; i++) cf[i] = (float)(cf[i] * sum), b->taps.t[i] = (int)lrint((double)cf[i] * 256.0);
  }
  const size_t px = (size_t)max_frames * rows * cols, kp = (size_t)max_frames * max_keypoints;
  bool ok = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipMalloc(&b->d_img, px) == hipSuccess && hipMalloc(&b->d_blur, px) == hipSuccess && hipMalloc(&b->d_score, px) == hipSuccess;
  ok = ok && hipMalloc(&b->d_kp, kp * 2 * sizeof(float)) == hipSuccess && hipMalloc(&b->d_wpts, kp * 2 * sizeof(float)) == hipSuccess;
  ok = ok && hipMalloc(&b->d_desc, kp * 4 * sizeof(unsigned long long)) == hipSuccess;
  ok = ok && hipMalloc(&b->d_nw, max_frames * sizeof(int)) == hipSuccess && hipMalloc(&b->d_nfast, max_frames * sizeof(int)) == hipSuccess &&
       hipMalloc(&b->d_nkp, max_frames * sizeof(int)) == hipSuccess && hipMalloc(&b->d_pat, 4 * 256 * sizeof(int)) == hipSuccess;
  if (ok) {
    int pat[4 * 256];
    memset(pat, 0, sizeof(pat));
    memcpy(pat, x1, n_bits * sizeof(int)), memcpy(pat + 256, y1, n_bits * sizeof(int));
    memcpy(pat + 512, x2, n_bits * sizeof(int)), memcpy(pat + 768, y2, n_bits * sizeof(int));
    ok = hipMemcpy(b->d_pat, pat, sizeof(pat), hipMemcpyHostToDevice) == hipSuccess;
  }
  if (!ok) {
    vio_brief_destroy(b);
    return VIO_ENOMEM;
  }
  *out = b;
  return VIO_OK;
}

int vio_brief_get_device(const vio_brief_t *b, int32_t *device) {
  if (!b || !device) return VIO_EINVAL;
  *device = b->device;
  return VIO_OK;
}

void vio_brief_destroy(vio_brief_t *b) {
  if (!b) return;
  vio::DeviceScope scope(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream), (void)hipStreamDestroy(b->stream);
  void *ptrs[] = {b->d_img, b->d_blur, b->d_score, b->d_kp, b->d_wpts, b->d_desc, b->d_nw, b->d_nfast, b->d_nkp, b->d_pat};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  delete b;
}

int vio_brief_extract(vio_brief_t *b, const uint8_t *gray, int32_t n_frames, const float *window_pts, const int32_t *n_window,
                      int32_t window_stride, int32_t fast_threshold, float *keypoints, uint64_t *descriptors, int32_t *n_fast,
                      int32_t *n_keypoints) {
  if (!b || !gray || n_frames < 1 || !n_window || !keypoints || !descriptors || !n_fast || !n_keypoints || window_stride < 0)
    return VIO_EINVAL;
  if (n_frames > b->max_frames) return VIO_ECAP;
  int max_w = 0;
  for (int f = 0; f < n_frames; f++) {
    if (n_window[f] < 0 || n_window[f] > window_stride || (n_window[f] > 0 && !window_pts)) return VIO_EINVAL;
    if (n_window[f] > b->cap) return VIO_ECAP;
    max_w = n_window[f] > max_w ? n_window[f] : max_w;
  }
  if (max_w > 0 && window_stride > b->cap) return VIO_ECAP;  // (the caller's stride is kept on the device)
  VIO_ON_DEVICE_OF(b);
  hipStream_t st = b->stream;
  const int rows = b->rows, cols = b->cols, cap = b->cap;
  const size_t px = (size_t)rows * cols;
  HIP_OK(hipMemcpyAsync(b->d_img, gray, px * n_frames, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync(b->d_nw, n_window, n_frames * sizeof(int), hipMemcpyHostToDevice, st));
  if (max_w > 0) {
    HIP_OK(hipMemcpyAsync(b->d_wpts, window_pts, (size_t)n_frames * window_stride * 2 * sizeof(float), hipMemcpyHostToDevice, st));
  }
  const int thr = fast_threshold < 0 ? 0 : (fast_threshold > 255 ? 255 : fast_threshold);
  hipLaunchKernelGGL(blur9_kernel, dim3((cols + kBlW - 1) / kBlW, (rows + kBlH - 1) / kBlH, n_frames), dim3(256), 0, st, b->d_img, b->d_blur,
                     rows, cols, b->taps);
  hipLaunchKernelGGL(fast_score_kernel, dim3((cols + 63) / 64, (rows + 3) / 4, n_frames), dim3(256), 0, st, b->d_img, b->d_score, rows, cols, thr);
  hipLaunchKernelGGL(fast_collect_kernel, dim3(n_frames), dim3(kColThreads), 0, st, b->d_score, rows, cols, b->d_wpts, b->d_nw,
                     window_stride, cap, b->d_kp, b->d_nfast, b->d_nkp);
  Pattern P = {b->d_pat, b->d_pat + 256, b->d_pat + 512, b->d_pat + 768, b->n_bits};
  hipLaunchKernelGGL(brief_kernel, dim3((cap + 3) / 4, n_frames), dim3(256), 0, st, b->d_blur, rows, cols, b->d_kp, b->d_nkp, cap, P,
                     b->d_desc);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(n_fast, b->d_nfast, n_frames * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(n_keypoints, b->d_nkp, n_frames * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(keypoints, b->d_kp, (size_t)n_frames * cap * 2 * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(descriptors, b->d_desc, (size_t)n_frames * cap * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  for (int f = 0; f < n_frames; f++)
    if (n_fast[f] > cap - n_window[f]) return VIO_ECAP;  // (what fitted is valid; the FAST list is cut at the capacity)
  return VIO_OK;
}

}  // extern "C"
