// vio_preprocess.hip — the image pre-step of the camera callback on the device, batched over frames:
//   cv::cvtColor(input_frame, gray, CV_RGBA2GRAY); clahe = cv::createCLAHE(); clahe->setClipLimit(3); clahe->apply(gray, img_equa)
// (VINS_ios/ViewController.mm:432-437), i.e. what every frame goes through right before FeatureTracker::readImage.
//
// Algorithm restated from the published OpenCV implementation the reference links (third-party, not in /root/reference:
// opencv2.framework for iOS, 3.x; modules/imgproc/src/color.cpp RGB2Gray<uchar> and modules/imgproc/src/clahe.cpp):
//   gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14
//   per tile (8 x 8 grid; image extended by BORDER_REFLECT_101 to a multiple of the grid when needed):
//     histogram -> clip at max(int(clipLimit * tileArea / 256), 1) -> the clipped mass is handed back, clipped/256 to
//     every bin and the remainder one count each to every (256/remainder)-th bin -> lut[i] = round(cdf[i] * 255/tileArea)
//   per pixel: bilinear blend (fp32, the order of operations of CLAHE_Interpolation_Body) of the four surrounding
//     tiles' lut[gray], rounded to nearest-even and saturated.
// Integer / fp32 work with a fixed operation order: the device result is bit-exact against the CPU oracle
// (oracle/vio_oracle_frontend.cpp::vio_oracle_preprocess); there is no OpenCV in this image to pin the oracle itself.
//
// Two kernels per batch, both HBM-bound streaming passes:
//   clahe_lut_kernel    one workgroup per (tile, frame): RGBA -> gray (written once), LDS histogram (u32 LDS atomics),
//                       clip + redistribution + 256-wide scan in the same workgroup, 256-byte LUT out
//   clahe_apply_kernel  one workgroup per band of 8 rows: the <= 3 tile rows of LUTs it needs staged in LDS as fp32
//                       (aligned dword reads), 4 pixels per lane (one dword load, one dword store)
// Algorithmic bytes per RGBA frame: 4*R*C (read) + R*C (gray out) + R*C (gray in) + R*C (equalized out) = 7*R*C.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include <algorithm>

#include "vio_amd.h"
#include "vio_device.h"
#include "vio_pool.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxTiles = 16;  // per axis
constexpr int kBandRows = 8;

struct PreGeom {
  int rows, cols, tiles_x, tiles_y, tw, th;  // tw/th: tile size of the (virtually) extended image
  int clip;                                  // 0: no clipping
  float lut_scale;
};

__device__ __forceinline__ int reflect101(int i, int n) { return i < n ? i : 2 * n - 2 - i; }

__device__ __forceinline__ uint32_t rgba_gray(uint32_t px) {  // little endian: R in the low byte
  const uint32_t r = px & 0xff, g = (px >> 8) & 0xff, b = (px >> 16) & 0xff;
  return (r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14;
}

template <int CH>
__global__ __launch_bounds__(kThreads) void clahe_lut_kernel(const uint8_t *__restrict__ src, size_t frame_stride,
                                                              int row_stride, uint8_t *__restrict__ gray,
                                                              uint8_t *__restrict__ lut, PreGeom G) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t scan[256];
  __shared__ uint32_t wsum[kThreads / 64];
  const int tid = threadIdx.x, tile = blockIdx.x, frame = blockIdx.y;
  const int tx = tile % G.tiles_x, ty = tile / G.tiles_x;
  hist[tid] = 0;
  __syncthreads();
  const uint8_t *img = src + (size_t)frame * frame_stride;
  uint8_t *gout = gray + (size_t)frame * G.rows * G.cols;
  const int area = G.tw * G.th;
  for (int p = tid; p < area; p += kThreads) {
    const int yy = p / G.tw, xx = p - yy * G.tw;
    const int ey = ty * G.th + yy, ex = tx * G.tw + xx;
    const int sy = reflect101(ey, G.rows), sx = reflect101(ex, G.cols);
    uint32_t v;
    if (CH == 4)
      v = rgba_gray(*reinterpret_cast<const uint32_t *>(img + (size_t)sy * row_stride + 4 * sx));
    else
      v = img[(size_t)sy * row_stride + sx];
    if (ey < G.rows && ex < G.cols) gout[(size_t)ey * G.cols + ex] = (uint8_t)v;  // every image pixel is in one tile
    atomicAdd(&hist[v], 1u);
  }
  __syncthreads();
  uint32_t h = hist[tid];
  if (G.clip > 0) {
    uint32_t excess = h > (uint32_t)G.clip ? h - G.clip : 0;
    if (excess) h = G.clip;
    // block sum of the clipped mass
    uint32_t s = excess;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = s;
    __syncthreads();
    uint32_t clipped = 0;
    for (int w = 0; w < kThreads / 64; w++) clipped += wsum[w];
    const uint32_t batch = clipped / 256u;
    uint32_t residual = clipped - batch * 256u;
    h += batch;
    if (residual != 0) {
      const uint32_t step = 256u / residual > 1u ? 256u / residual : 1u;  // MAX(histSize / residual, 1)
      if ((uint32_t)tid % step == 0 && (uint32_t)tid / step < residual) h += 1;
    }
  }
  // inclusive scan over the 256 bins
  scan[tid] = h;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t add = tid >= o ? scan[tid - o] : 0;
    __syncthreads();
    scan[tid] += add;
    __syncthreads();
  }
  const float f = (float)(int)scan[tid] * G.lut_scale;
  int r = __float2int_rn(f);  // cvRound
  r = r < 0 ? 0 : (r > 255 ? 255 : r);
  lut[((size_t)frame * G.tiles_x * G.tiles_y + tile) * 256 + tid] = (uint8_t)r;
}

__device__ __forceinline__ uint32_t blend_px(const float *__restrict__ L1, const float *__restrict__ L2, int i1, int i2,
                                              uint32_t v, float xa, float xa1, float ya, float ya1) {
  const float res = (L1[i1 + v] * xa1 + L1[i2 + v] * xa) * ya1 + (L2[i1 + v] * xa1 + L2[i2 + v] * xa) * ya;
  int r = __float2int_rn(res);
  r = r < 0 ? 0 : (r > 255 ? 255 : r);
  return (uint32_t)r;
}

template <bool VEC4>
__global__ __launch_bounds__(kThreads) void clahe_apply_kernel(const uint8_t *__restrict__ gray,
                                                                const uint8_t *__restrict__ lut, uint8_t *__restrict__ dst,
                                                                size_t dst_frame_stride, int dst_row_stride, PreGeom G) {
  __shared__ float L[3 * kMaxTiles * 256];
  const int tid = threadIdx.x, frame = blockIdx.y;
  const int y0 = blockIdx.x * kBandRows, y1 = min(y0 + kBandRows, G.rows);
  const float inv_th = 1.0f / G.th, inv_tw = 1.0f / G.tw;
  int t_lo = (int)floorf(y0 * inv_th - 0.5f);
  t_lo = t_lo < 0 ? 0 : t_lo;
  int t_hi = (int)floorf((y1 - 1) * inv_th - 0.5f) + 1;
  t_hi = t_hi > G.tiles_y - 1 ? G.tiles_y - 1 : t_hi;
  const int n_lut = (t_hi - t_lo + 1) * G.tiles_x * 256;  // <= 3 tile rows (kBandRows <= th is checked on the host)
  const uint8_t *fl = lut + ((size_t)frame * G.tiles_y + t_lo) * G.tiles_x * 256;
  for (int i = tid * 4; i < n_lut; i += kThreads * 4) {
    const uint32_t q = *reinterpret_cast<const uint32_t *>(fl + i);
    L[i] = (float)(q & 0xff), L[i + 1] = (float)((q >> 8) & 0xff), L[i + 2] = (float)((q >> 16) & 0xff), L[i + 3] = (float)(q >> 24);
  }
  __syncthreads();
  const uint8_t *gin = gray + (size_t)frame * G.rows * G.cols;
  uint8_t *out = dst + (size_t)frame * dst_frame_stride;
  const int groups = VEC4 ? G.cols / 4 : G.cols;
  for (int p = tid; p < (y1 - y0) * groups; p += kThreads) {
    const int y = y0 + p / groups, gx = p % groups;
    const float tyf = y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf);
    int ty2 = ty1 + 1;
    const float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = ty1 < 0 ? 0 : ty1;
    ty2 = ty2 > G.tiles_y - 1 ? G.tiles_y - 1 : ty2;
    const float *L1 = L + (ty1 - t_lo) * G.tiles_x * 256, *L2 = L + (ty2 - t_lo) * G.tiles_x * 256;
    if (VEC4) {
      const uint32_t q = *reinterpret_cast<const uint32_t *>(gin + (size_t)y * G.cols + 4 * gx);
      uint32_t o = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int x = 4 * gx + k;
        const float txf = x * inv_tw - 0.5f;
        int tx1 = (int)floorf(txf);
        int tx2 = tx1 + 1;
        const float xa = txf - tx1, xa1 = 1.0f - xa;
        tx1 = tx1 < 0 ? 0 : tx1;
        tx2 = tx2 > G.tiles_x - 1 ? G.tiles_x - 1 : tx2;
        o |= blend_px(L1, L2, tx1 * 256, tx2 * 256, (q >> (8 * k)) & 0xff, xa, xa1, ya, ya1) << (8 * k);
      }
      *reinterpret_cast<uint32_t *>(out + (size_t)y * dst_row_stride + 4 * gx) = o;
    } else {
      const int x = gx;
      const float txf = x * inv_tw - 0.5f;
      int tx1 = (int)floorf(txf);
      int tx2 = tx1 + 1;
      const float xa = txf - tx1, xa1 = 1.0f - xa;
      tx1 = tx1 < 0 ? 0 : tx1;
      tx2 = tx2 > G.tiles_x - 1 ? G.tiles_x - 1 : tx2;
      out[(size_t)y * dst_row_stride + x] =
          (uint8_t)blend_px(L1, L2, tx1 * 256, tx2 * 256, gin[(size_t)y * G.cols + x], xa, xa1, ya, ya1);
    }
  }
}

}  // namespace

struct vio_preprocess {
  int device = -1;  // HIP device the context lives on (current device at create)
  int max_frames = 0, rows = 0, cols = 0;
  double clip_limit = 3.0;  // clahe->setClipLimit(3) ViewController.mm:436
  int tiles_x = 8, tiles_y = 8;  // cv::createCLAHE() default tileGridSize
  uint8_t *d_src = nullptr, *d_gray = nullptr, *d_lut = nullptr, *d_out = nullptr;
  uint8_t *h_in = nullptr, *h_out = nullptr;  // page-locked staging of the host-buffer entry point
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double ms_sum = 0;
  int launches = 0;
};

namespace {

PreGeom geometry(const vio_preprocess *p) {
  PreGeom G;
  G.rows = p->rows, G.cols = p->cols, G.tiles_x = p->tiles_x, G.tiles_y = p->tiles_y;
  // an image that is not a multiple of the grid in BOTH directions is extended on the right by tiles_x - cols % tiles_x
  // and at the bottom by tiles_y - rows % tiles_y (a whole extra `tiles` in a direction that did divide: as published)
  const bool fits = p->cols % p->tiles_x == 0 && p->rows % p->tiles_y == 0;
  const int ext_c = fits ? p->cols : p->cols + (p->tiles_x - p->cols % p->tiles_x);
  const int ext_r = fits ? p->rows : p->rows + (p->tiles_y - p->rows % p->tiles_y);
  G.tw = ext_c / p->tiles_x, G.th = ext_r / p->tiles_y;
  const int area = G.tw * G.th;
  G.lut_scale = (float)(256 - 1) / area;
  G.clip = 0;
  if (p->clip_limit > 0.0) {
    G.clip = (int)(p->clip_limit * area / 256);
    if (G.clip < 1) G.clip = 1;
  }
  return G;
}

int launch(vio_preprocess *p, const uint8_t *d_src, int channels, int n_frames, size_t frame_stride, int row_stride,
           uint8_t *d_out, size_t out_frame_stride, int out_row_stride, hipStream_t st) {
  const PreGeom G = geometry(p);
  const dim3 g1(G.tiles_x * G.tiles_y, n_frames), g2((G.rows + kBandRows - 1) / kBandRows, n_frames);
  (void)hipEventRecord(p->ev0, st);
  if (channels == 4)
    hipLaunchKernelGGL(clahe_lut_kernel<4>, g1, dim3(kThreads), 0, st, d_src, frame_stride, row_stride, p->d_gray, p->d_lut, G);
  else
    hipLaunchKernelGGL(clahe_lut_kernel<1>, g1, dim3(kThreads), 0, st, d_src, frame_stride, row_stride, p->d_gray, p->d_lut, G);
  const bool vec4 = G.cols % 4 == 0 && out_row_stride % 4 == 0 && out_frame_stride % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(d_out) & 3) == 0;
  if (vec4)
    hipLaunchKernelGGL(clahe_apply_kernel<true>, g2, dim3(kThreads), 0, st, p->d_gray, p->d_lut, d_out, out_frame_stride,
                       out_row_stride, G);
  else
    hipLaunchKernelGGL(clahe_apply_kernel<false>, g2, dim3(kThreads), 0, st, p->d_gray, p->d_lut, d_out, out_frame_stride,
                       out_row_stride, G);
  (void)hipEventRecord(p->ev1, st);
  return hipGetLastError() == hipSuccess ? VIO_OK : VIO_ENODEV;
}

void account(vio_preprocess *p) {
  float ms = 0;
  if (hipEventElapsedTime(&ms, p->ev0, p->ev1) == hipSuccess) p->ms_sum += ms, p->launches++;
}

}  // namespace

extern "C" {

int vio_preprocess_create(int32_t max_frames, int32_t rows, int32_t cols, vio_preprocess_t **out) {
  if (!out || max_frames < 1 || rows < 16 || cols < 16 || rows > 16384 || cols > 16384) return VIO_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    fprintf(stderr, "vio_amd: no HIP device visible; the image pre-step has no CPU fallback\n");
    return VIO_ENODEV;
  }
  vio_preprocess *p = new (std::nothrow) vio_preprocess();
  if (p) p->device = vio::current_device();
  if (!p) return VIO_ENOMEM;
  p->max_frames = max_frames, p->rows = rows, p->cols = cols;
  const size_t px = (size_t)rows * cols;
  bool ok = hipMalloc((void **)&p->d_src, px * 4 * max_frames) == hipSuccess &&
            hipMalloc((void **)&p->d_gray, px * max_frames) == hipSuccess &&
            hipMalloc((void **)&p->d_out, px * max_frames) == hipSuccess &&
            hipMalloc((void **)&p->d_lut, (size_t)kMaxTiles * kMaxTiles * 256 * max_frames) == hipSuccess &&
            hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreate(&p->ev0) == hipSuccess && hipEventCreate(&p->ev1) == hipSuccess;
  if (!ok) {
    vio_preprocess_destroy(p);
    return VIO_ENOMEM;
  }
  if (vio_preprocess_set_clahe(p, 3.0, 8, 8) != VIO_OK) {  // the reference's setting must fit the frame size
    vio_preprocess_destroy(p);
    return VIO_EINVAL;
  }
  *out = p;
  return VIO_OK;
}

void vio_preprocess_destroy(vio_preprocess_t *p) {
  if (!p) return;
  vio::DeviceScope scope(p->device);
  if (p->d_src) (void)hipFree(p->d_src);
  if (p->d_gray) (void)hipFree(p->d_gray);
  if (p->d_out) (void)hipFree(p->d_out);
  if (p->d_lut) (void)hipFree(p->d_lut);
  if (p->h_in) (void)hipHostFree(p->h_in);
  if (p->h_out) (void)hipHostFree(p->h_out);
  if (p->ev0) (void)hipEventDestroy(p->ev0);
  if (p->ev1) (void)hipEventDestroy(p->ev1);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

int vio_preprocess_set_clahe(vio_preprocess_t *p, double clip_limit, int32_t tiles_x, int32_t tiles_y) {
  if (!p || tiles_x < 1 || tiles_y < 1 || tiles_x > kMaxTiles || tiles_y > kMaxTiles) return VIO_EINVAL;
  vio_preprocess q = *p;
  q.clip_limit = clip_limit, q.tiles_x = tiles_x, q.tiles_y = tiles_y;
  const PreGeom G = geometry(&q);
  // a band of kBandRows rows must not span more than 3 tile rows; the reflected extension must stay inside the image
  if (G.th < kBandRows || G.tw < 1 || G.th * tiles_y - p->rows >= p->rows || G.tw * tiles_x - p->cols >= p->cols) return VIO_EINVAL;
  p->clip_limit = clip_limit, p->tiles_x = tiles_x, p->tiles_y = tiles_y;
  return VIO_OK;
}

int vio_preprocess_run(vio_preprocess_t *p, const uint8_t *pixels, int32_t channels, int32_t n_frames, int32_t stride,
                       uint8_t *gray_out, uint8_t *equalized_out) {
  if (!p || !pixels || !equalized_out || !(channels == 1 || channels == 4) || n_frames < 1 || stride < channels * p->cols)
    return VIO_EINVAL;
  if (n_frames > p->max_frames) return VIO_ECAP;
  VIO_ON_DEVICE_OF(p);
  const size_t px = (size_t)p->rows * p->cols, row_bytes = (size_t)channels * p->cols, fb = row_bytes * p->rows;
  hipStream_t st = p->stream;
  // caller memory is pageable: gather into / scatter from page-locked staging with the host pool, a few frames per chunk,
  // so that the DMA of one chunk overlaps the host copy of the next (a direct pageable copy runs at ~5 GB/s)
  const size_t N = (size_t)n_frames;
  if (!p->h_in && hipHostMalloc((void **)&p->h_in, (size_t)p->max_frames * px * 4, hipHostMallocDefault) != hipSuccess) return VIO_ENOMEM;
  if (!p->h_out && hipHostMalloc((void **)&p->h_out, (size_t)p->max_frames * px * 2, hipHostMallocDefault) != hipSuccess) return VIO_ENOMEM;
  const size_t n_chunks = N >= 16 ? 8 : 1, per = (N + n_chunks - 1) / n_chunks;
  for (size_t c0 = 0; c0 < N; c0 += per) {
    const size_t c1 = std::min(N, c0 + per);
    vio::HostPool::get().parallel_for((int)(c1 - c0), [&](int i) {
      const size_t f = c0 + i;
      const uint8_t *src = pixels + f * (size_t)p->rows * stride;
      uint8_t *dst = p->h_in + f * fb;
      if ((size_t)stride == row_bytes) memcpy(dst, src, fb);
      else
        for (int r = 0; r < p->rows; r++) memcpy(dst + (size_t)r * row_bytes, src + (size_t)r * stride, row_bytes);
    });
    if (hipMemcpyAsync(p->d_src + c0 * fb, p->h_in + c0 * fb, (c1 - c0) * fb, hipMemcpyHostToDevice, st) != hipSuccess) return VIO_ENODEV;
  }
  int rc = launch(p, p->d_src, channels, n_frames, fb, (int)row_bytes, p->d_out, px, p->cols, st);
  if (rc != VIO_OK) return rc;
  if (hipMemcpyAsync(p->h_out, p->d_out, px * N, hipMemcpyDeviceToHost, st) != hipSuccess) return VIO_ENODEV;
  if (gray_out && hipMemcpyAsync(p->h_out + (size_t)p->max_frames * px, p->d_gray, px * N, hipMemcpyDeviceToHost, st) != hipSuccess)
    return VIO_ENODEV;
  if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return VIO_ENODEV;
  vio::HostPool::get().parallel_for(n_frames, [&](int f) {
    memcpy(equalized_out + (size_t)f * px, p->h_out + (size_t)f * px, px);
    if (gray_out) memcpy(gray_out + (size_t)f * px, p->h_out + ((size_t)p->max_frames + f) * px, px);
  });
  account(p);
  return VIO_OK;
}

int vio_preprocess_run_resident(vio_preprocess_t *p, const void *d_pixels, int32_t channels, int32_t n_frames, int32_t stride,
                                void *d_equalized, void *stream) {
  if (!p || !d_pixels || !d_equalized || !(channels == 1 || channels == 4) || n_frames < 1 || stride < channels * p->cols)
    return VIO_EINVAL;
  if (n_frames > p->max_frames) return VIO_ECAP;
  if (channels == 4 && ((reinterpret_cast<uintptr_t>(d_pixels) | (uintptr_t)stride) & 3)) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(p);
  hipStream_t st = stream ? (hipStream_t)stream : p->stream;
  return launch(p, (const uint8_t *)d_pixels, channels, n_frames, (size_t)stride * p->rows, stride, (uint8_t *)d_equalized,
                (size_t)p->rows * p->cols, p->cols, st);
}

int vio_preprocess_sync(vio_preprocess_t *p) {
  if (!p) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(p);
  if (hipEventSynchronize(p->ev1) != hipSuccess || hipGetLastError() != hipSuccess) return VIO_ENODEV;
  account(p);
  return VIO_OK;
}

int vio_preprocess_kernel_ms(vio_preprocess_t *p, double *ms_avg, int32_t *launches) {
  if (!p || !ms_avg || !launches) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(p);
  *launches = p->launches;
  *ms_avg = p->launches ? p->ms_sum / p->launches : 0.0;
  p->ms_sum = 0, p->launches = 0;
  return VIO_OK;
}

}  // extern "C"
