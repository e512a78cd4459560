// vio_loop.hip — descriptor matching of the loop-closure producer (second half):
// KeyFrame::searchByDes (VINS_ios/loop/keyframe.cpp:161-187) as a gfx950 kernel, and findConnectionWithOldFrame
// (:267-273) = searchByDes + rejectWithF (:35-58, findFundamentalMat RANSAC with a 2.0 px threshold) on top of the
// tracker's block-cooperative RANSAC.
//
// A BRIEF descriptor is 256 bits (BRIEF::bitset, loop/keyframe.h), here four 64-bit words, word 0 = bits 0..63. For
// every window descriptor of the current keyframe the best (smallest Hamming distance, first index on ties — the
// reference scans j upwards with `dis < bestDist`) descriptor of the old keyframe is found: integer work, bit-exact
// against the plain restatement in oracle/. Many (current, old) keyframe pairs share one launch: one wave per query
// descriptor, the old keyframe's descriptors staged through LDS once per workgroup.
#include <hip/hip_runtime.h>

#include <stdio.h>

#include <new>
#include <vector>

#include "vio_amd.h"
#include "vio_device.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kStage = 512;  // old descriptors staged per round: 16 KB of LDS

#define HIP_OK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "vio_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VIO_ENODEV;                                                                   \
    }                                                                                      \
  } while (0)

// grid (ceil(max_cur / 4), n_pairs); pair p: queries cur[cur_off[p] .. +n_cur[p]), candidates old[old_off[p] .. +n_old[p])
__global__ __launch_bounds__(64 * kWavesPerBlock) void search_by_des_kernel(const unsigned long long *cur,
                                                                            const unsigned long long *old, const int *n_cur,
                                                                            const int *n_old, const int *cur_off,
                                                                            const int *old_off, int *best_index, int *best_dist) {
  __shared__ unsigned long long stage[kStage][4];
  const int p = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nc = n_cur[p], no = n_old[p];
  if ((int)blockIdx.x * kWavesPerBlock >= nc) return;  // (uniform per workgroup)
  const int q = blockIdx.x * kWavesPerBlock + wave;
  const bool have = q < nc;
  unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  if (have) {
    const unsigned long long *c = cur + 4 * (size_t)(cur_off[p] + q);
    a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
  }
  // key = distance << 16 | index: the minimum key is the smallest distance and, among equals, the first index
  unsigned best = 256u << 16;  // bestDist = 256, bestIndex = -1 (keyframe.cpp:169-170)
  const unsigned long long *ob = old + 4 * (size_t)old_off[p];
  for (int j0 = 0; j0 < no; j0 += kStage) {
    const int nj = min(kStage, no - j0);
    __syncthreads();
    for (int e = threadIdx.x; e < nj * 4; e += blockDim.x) (&stage[0][0])[e] = ob[4 * (size_t)j0 + e];
    __syncthreads();
    if (have)
      for (int j = lane; j < nj; j += 64) {
        const unsigned d = __popcll(a0 ^ stage[j][0]) + __popcll(a1 ^ stage[j][1]) + __popcll(a2 ^ stage[j][2]) +
                           __popcll(a3 ^ stage[j][3]);
        const unsigned key = (d << 16) | (unsigned)(j0 + j);
        best = key < best ? key : best;
      }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = __shfl_xor(best, o, 64);
    best = t < best ? t : best;
  }
  if (have && lane == 0) {
    const int d = (int)(best >> 16);
    best_dist[cur_off[p] + q] = d;
    best_index[cur_off[p] + q] = d < 256 ? (int)(best & 0xffff) : -1;   // `if (bestDist < 256)` (:182)
  }
}

template <class T>
struct Buf {
  T *p = nullptr;
  size_t n = 0;
  int ensure(size_t count) {
    if (count <= n && p) return VIO_OK;
    if (p) (void)hipFree(p);
    p = nullptr, n = 0;
    if (hipMalloc((void **)&p, (count ? count : 1) * sizeof(T)) != hipSuccess) return VIO_ENOMEM;
    n = count;
    return VIO_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr, n = 0;
  }
};

}  // namespace

struct vio_matcher {
  int device = -1;
  hipStream_t stream = nullptr;
  Buf<unsigned long long> d_cur, d_old;
  Buf<int> d_meta, d_idx, d_dist;
};

extern "C" {

int vio_matcher_create(vio_matcher_t **out) {
  if (!out) return VIO_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    fprintf(stderr, "vio_amd: no HIP device visible; the descriptor matcher has no CPU fallback\n");
    return VIO_ENODEV;
  }
  vio_matcher *m = new (std::nothrow) vio_matcher();
  if (!m) return VIO_ENOMEM;
  m->device = vio::current_device();
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
    delete m;
    return VIO_ENODEV;
  }
  *out = m;
  return VIO_OK;
}

void vio_matcher_destroy(vio_matcher_t *m) {
  if (!m) return;
  vio::DeviceScope scope(m->device);
  (void)hipStreamSynchronize(m->stream);
  m->d_cur.release(), m->d_old.release(), m->d_meta.release(), m->d_idx.release(), m->d_dist.release();
  (void)hipStreamDestroy(m->stream);
  delete m;
}

int vio_matcher_search_by_des(vio_matcher_t *m, int32_t n_pairs, const int32_t *n_cur, const int32_t *n_old,
                              const uint64_t *cur_desc, const uint64_t *old_desc, int32_t *best_index, int32_t *best_dist) {
  if (!m || n_pairs < 0 || (n_pairs > 0 && (!n_cur || !n_old || !best_index || !best_dist))) return VIO_EINVAL;
  if (n_pairs == 0) return VIO_OK;
  if (n_pairs > 65535) return VIO_ECAP;  // (grid.y: one block row per pair)
  VIO_ON_DEVICE_OF(m);
  std::vector<int> meta;
  long long tc = 0, to = 0;
  int max_cur = 0;
  try {
    meta.resize((size_t)4 * n_pairs);
    for (int p = 0; p < n_pairs; p++) {
      if (n_cur[p] < 0 || n_old[p] < 0 || n_old[p] > 65535) return n_old[p] > 65535 ? VIO_ECAP : VIO_EINVAL;
      meta[p] = n_cur[p], meta[n_pairs + p] = n_old[p], meta[2 * n_pairs + p] = (int)tc, meta[3 * n_pairs + p] = (int)to;
      tc += n_cur[p], to += n_old[p];
      max_cur = n_cur[p] > max_cur ? n_cur[p] : max_cur;
    }
  } catch (const std::bad_alloc &) {
    return VIO_ENOMEM;
  }
  if (tc == 0) return VIO_OK;
  if (!cur_desc || (to > 0 && !old_desc) || tc > 0x7fffffff || to > 0x7fffffff) return VIO_EINVAL;
  int rc = m->d_cur.ensure(4 * (size_t)tc);
  if (rc == VIO_OK) rc = m->d_old.ensure(4 * (size_t)(to ? to : 1));
  if (rc == VIO_OK) rc = m->d_meta.ensure(meta.size());
  if (rc == VIO_OK) rc = m->d_idx.ensure((size_t)tc);
  if (rc == VIO_OK) rc = m->d_dist.ensure((size_t)tc);
  if (rc != VIO_OK) return rc;
  hipStream_t st = m->stream;
  HIP_OK(hipMemcpyAsync(m->d_cur.p, cur_desc, 32 * (size_t)tc, hipMemcpyHostToDevice, st));
  if (to) HIP_OK(hipMemcpyAsync(m->d_old.p, old_desc, 32 * (size_t)to, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync(m->d_meta.p, meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice, st));
  dim3 grid((max_cur + kWavesPerBlock - 1) / kWavesPerBlock, n_pairs);
  hipLaunchKernelGGL(search_by_des_kernel, grid, dim3(64 * kWavesPerBlock), 0, st, m->d_cur.p, m->d_old.p, m->d_meta.p,
                     m->d_meta.p + n_pairs, m->d_meta.p + 2 * n_pairs, m->d_meta.p + 3 * n_pairs, m->d_idx.p, m->d_dist.p);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(best_index, m->d_idx.p, sizeof(int) * (size_t)tc, hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(best_dist, m->d_dist.p, sizeof(int) * (size_t)tc, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  return VIO_OK;
}

// KeyFrame::findConnectionWithOldFrame (keyframe.cpp:267-273): searchByDes, then rejectWithF on (measurements of the
// current keyframe, matched keypoints of the old one) when at least 8 matches exist.
int vio_loop_find_connection(vio_matcher_t *m, const VioConfig *cfg, int32_t n_cur, const uint64_t *cur_desc,
                             const float *cur_pts, int32_t n_old, const uint64_t *old_desc, const float *old_pts,
                             float *matched_old_pts, float *matched_old_norm, uint8_t *status, int32_t *n_inliers) {
  if (!m || !cfg || n_cur < 0 || n_old < 0 || !status || !n_inliers || (n_cur > 0 && (!cur_desc || !cur_pts || !matched_old_pts)))
    return VIO_EINVAL;
  if (n_old > 0 && (!old_desc || !old_pts)) return VIO_EINVAL;
  *n_inliers = 0;
  if (n_cur == 0) return VIO_OK;
  VIO_ON_DEVICE_OF(m);  // (the RANSAC below runs on the matcher's device as well: vio_amd.h, DEVICE BINDING)
  std::vector<int> idx, dist;
  try {
    idx.resize(n_cur), dist.resize(n_cur);
  } catch (const std::bad_alloc &) {
    return VIO_ENOMEM;
  }
  int rc = vio_matcher_search_by_des(m, 1, &n_cur, &n_old, cur_desc, old_desc, idx.data(), dist.data());
  if (rc != VIO_OK) return rc;
  // (with a non-empty old keyframe every query has a match: the distance of two 256-bit strings is at most 256 only when
  // all bits differ; the reference then drops the query and its arrays go out of step — restated as "no connection")
  for (int i = 0; i < n_cur; i++) {
    if (idx[i] < 0) {
      for (int k = 0; k < n_cur; k++) status[k] = 0;
      return VIO_OK;
    }
    matched_old_pts[2 * i] = old_pts[2 * idx[i]], matched_old_pts[2 * i + 1] = old_pts[2 * idx[i] + 1];
  }
  if (n_cur >= 8) {  // rejectWithF (keyframe.cpp:35-58)
    if (matched_old_norm)
      for (int i = 0; i < n_cur; i++) {
        matched_old_norm[2 * i] = (float)((matched_old_pts[2 * i] - (float)cfg->cx) / (float)cfg->fx);
        matched_old_norm[2 * i + 1] = (float)((matched_old_pts[2 * i + 1] - (float)cfg->cy) / (float)cfg->fy);
      }
    VioConfig c = *cfg;
    c.f_threshold = 2.0, c.f_confidence = 0.99;  // cv::findFundamentalMat(measurements, measurements_old, FM_RANSAC, 2.0, 0.99)
    rc = vio_fundamental_ransac(&c, cur_pts, matched_old_pts, n_cur, status);
    if (rc != VIO_OK) return rc;
  } else {
    for (int i = 0; i < n_cur; i++) status[i] = 1;
  }
  int k = 0;
  for (int i = 0; i < n_cur; i++) k += status[i] ? 1 : 0;
  *n_inliers = k;
  return VIO_OK;
}

}  // extern "C"
