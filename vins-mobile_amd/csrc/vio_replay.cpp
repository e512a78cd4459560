// vio_replay.cpp — the recording formats of the reference's record / playback mode and the IMU-image association of
// its estimator thread (host side, I/O only).
//
// Reference (VINS_ios/): IMU_MSG and the "IMU" file = back-to-back structs closed by a header == 0 record
// (ViewController.h:58-62, ViewController.mm:1120-1150, 1505-1511, 1614-1622); "IMAGE/<index>" = one PNG per frame and
// "IMAGE_TIME/<index>" = its 8-byte timestamp (ViewController.mm:1634-1708); the RGBA -> gray step of the camera
// callback (cv::cvtColor CV_RGBA2GRAY, ViewController.mm:432-433); KEYFRAME_DATA, the pose record of the keyframe
// database (loop/keyfame_database.h:22-27, keyfame_database.cpp:383-387); getMeasurements / send_imu
// (ViewController.mm:603-682). PNG is decoded with zlib's inflate (the image's libz); libpng headers are not
// available and nothing else is needed for 8/16-bit non-interlaced files.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <deque>
#include <new>
#include <string>
#include <vector>

#include "vio_amd.h"

namespace {

bool read_file(const char *path, std::vector<uint8_t> &out) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  size_t got = n > 0 ? fread(out.data(), 1, (size_t)n, f) : 0;
  fclose(f);
  return got == out.size();
}

bool write_file(const char *path, const void *data, size_t n) {
  FILE *f = fopen(path, "wb");
  if (!f) return false;
  size_t put = n ? fwrite(data, 1, n, f) : 0;
  return fclose(f) == 0 && put == n;
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put_be32(uint8_t *p, uint32_t v) { p[0] = v >> 24, p[1] = v >> 16, p[2] = v >> 8, p[3] = v; }

const uint8_t kPngSig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};

// cv::cvtColor RGB(A) -> gray, 8-bit: fixed point with 14 fractional bits, coefficients 0.299 / 0.587 / 0.114
// (OpenCV imgproc color conversion, RGB2Gray<uchar>: R2Y 4899, G2Y 9617, B2Y 1868, descale by 14 with rounding).
inline uint8_t rgb_to_gray(int r, int g, int b) { return (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14); }

int paeth(int a, int b, int c) {
  int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// PNG (ISO/IEC 15948) -> 8-bit gray. Colour types 0/2/3/4/6, bit depth 8 or 16 (1/2/4 for gray and palette),
// non-interlaced. Alpha is ignored like CV_RGBA2GRAY ignores it.
int decode_png_gray_impl(const std::vector<uint8_t> &file, uint8_t *gray, int64_t cap, int32_t *rows, int32_t *cols) {
  if (file.size() < 8 + 25 || memcmp(file.data(), kPngSig, 8) != 0) return VIO_EINVAL;
  size_t pos = 8;
  uint32_t width = 0, height = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte;
  bool have_ihdr = false, done = false;
  while (!done && pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const uint8_t *type = &file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) return VIO_EINVAL;
    const uint8_t *data = &file[pos + 8];
    if (be32(data + len) != (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4)) return VIO_EINVAL;
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) return VIO_EINVAL;
      width = be32(data), height = be32(data + 4), depth = data[8], ctype = data[9], interlace = data[12];
      have_ihdr = true;
    } else if (!memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      done = true;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || idat.empty() || width == 0 || height == 0 || width > 16384 || height > 16384 ||
      (uint64_t)width * height > (1u << 26))  // 64 Mpixel: far beyond any camera frame, keeps the buffers bounded
    return VIO_EINVAL;
  if (interlace != 0) return VIO_EINVAL;
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return VIO_EINVAL;
  }
  if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return VIO_EINVAL;
  if (ctype == 3 && depth == 16) return VIO_EINVAL;
  *rows = (int32_t)height, *cols = (int32_t)width;
  if ((int64_t)width * height > cap) return VIO_ECAP;
  const size_t bpp_bits = (size_t)channels * depth, stride = (width * bpp_bits + 7) / 8, bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
  std::vector<uint8_t> raw((stride + 1) * height);
  uLongf raw_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return VIO_EINVAL;
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  for (uint32_t y = 0; y < height; y++) {
    const uint8_t *src = &raw[(stride + 1) * y];
    const int filter = src[0];
    if (filter > 4) return VIO_EINVAL;
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int v = src[1 + i];
      switch (filter) {
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: break;
      }
      cur[i] = (uint8_t)v;
    }
    uint8_t *dst = gray + (size_t)y * width;
    const int step = depth == 16 ? 2 : 1;  // 16-bit samples: the high byte (what an 8-bit conversion keeps)
    for (uint32_t x = 0; x < width; x++) {
      if (depth < 8) {
        const int per = 8 / depth, sh = (per - 1 - (int)(x % per)) * depth;
        const int v = (cur[x / per] >> sh) & ((1 << depth) - 1);
        if (ctype == 3) {
          if ((size_t)v * 3 + 2 >= plte.size()) return VIO_EINVAL;
          dst[x] = rgb_to_gray(plte[v * 3], plte[v * 3 + 1], plte[v * 3 + 2]);
        } else {
          dst[x] = (uint8_t)(v * 255 / ((1 << depth) - 1));
        }
        continue;
      }
      const uint8_t *p = &cur[(size_t)x * channels * step];
      switch (ctype) {
        case 0:
        case 4: dst[x] = p[0]; break;
        case 2:
        case 6: dst[x] = rgb_to_gray(p[0], p[step], p[2 * step]); break;
        case 3:
          if ((size_t)p[0] * 3 + 2 >= plte.size()) return VIO_EINVAL;
          dst[x] = rgb_to_gray(plte[p[0] * 3], plte[p[0] * 3 + 1], plte[p[0] * 3 + 2]);
          break;
      }
    }
    prev.swap(cur);
  }
  return VIO_OK;
}

// The C ABI never throws: allocation failures on hostile sizes come back as VIO_ENOMEM.
int decode_png_gray(const std::vector<uint8_t> &file, uint8_t *gray, int64_t cap, int32_t *rows, int32_t *cols) {
  try {
    return decode_png_gray_impl(file, gray, cap, rows, cols);
  } catch (const std::bad_alloc &) {
    return VIO_ENOMEM;
  }
}

void png_chunk(std::vector<uint8_t> &out, const char *type, const uint8_t *data, uint32_t len) {
  uint8_t hdr[8];
  put_be32(hdr, len), memcpy(hdr + 4, type, 4);
  out.insert(out.end(), hdr, hdr + 8);
  if (len) out.insert(out.end(), data, data + len);
  uLong c = crc32(crc32(0L, Z_NULL, 0), (const Bytef *)type, 4);
  if (len) c = crc32(c, data, len);
  uint8_t crc[4];
  put_be32(crc, (uint32_t)c);
  out.insert(out.end(), crc, crc + 4);
}

std::string index_path(const char *dir, uint64_t index) {
  char name[32];
  snprintf(name, sizeof(name), "/%llu", (unsigned long long)index);  // [NSString stringWithFormat:@"%lu", index]
  return std::string(dir) + name;
}

struct ImgMsg {
  double header;
  std::vector<VioObs> obs;
};

}  // namespace

struct vio_measurements {  // imu_msg_buf / img_msg_buf (ViewController.mm:343-346) + current_time of send_imu (:663-667)
  std::deque<VioImuMsg> imu;
  std::deque<ImgMsg> img;
  double current_time = -1;
};

extern "C" {

int vio_replay_read_imu(const char *path, VioImuMsg *out, int32_t cap, int32_t *n) {
  if (!path || !n || cap < 0 || (cap > 0 && !out)) return VIO_EINVAL;
  std::vector<uint8_t> file;
  if (!read_file(path, file)) return VIO_EINVAL;
  const size_t total = file.size() / sizeof(VioImuMsg);
  int32_t k = 0;
  for (size_t i = 0; i < total; i++) {
    VioImuMsg m;
    memcpy(&m, &file[i * sizeof(VioImuMsg)], sizeof(m));
    if (m.header == 0) break;  // the ending marker (ViewController.mm:1127, 1508)
    if (k < cap) out[k] = m;
    k++;
  }
  *n = k;
  return k <= cap || cap == 0 ? VIO_OK : VIO_ECAP;
}

int vio_replay_write_imu(const char *path, const VioImuMsg *msgs, int32_t n) {
  if (!path || n < 0 || (n > 0 && !msgs)) return VIO_EINVAL;
  std::vector<VioImuMsg> all(msgs, msgs + n);
  VioImuMsg end;
  memset(&end, 0, sizeof(end));
  all.push_back(end);
  return write_file(path, all.data(), all.size() * sizeof(VioImuMsg)) ? VIO_OK : VIO_EINVAL;
}

int vio_replay_read_image_time(const char *dir, uint64_t index, double *header) {
  if (!dir || !header) return VIO_EINVAL;
  std::vector<uint8_t> file;
  if (!read_file(index_path(dir, index).c_str(), file) || file.size() < sizeof(double)) return VIO_EINVAL;
  memcpy(header, file.data(), sizeof(double));
  return VIO_OK;
}

int vio_replay_write_image_time(const char *dir, uint64_t index, double header) {
  if (!dir) return VIO_EINVAL;
  return write_file(index_path(dir, index).c_str(), &header, sizeof(header)) ? VIO_OK : VIO_EINVAL;
}

int vio_replay_decode_png_gray(const uint8_t *png, int64_t png_bytes, uint8_t *gray, int64_t cap, int32_t *rows,
                               int32_t *cols) {
  if (!png || png_bytes <= 0 || !rows || !cols || cap < 0 || (cap > 0 && !gray)) return VIO_EINVAL;
  std::vector<uint8_t> file(png, png + png_bytes);
  return decode_png_gray(file, gray, cap, rows, cols);
}

int vio_replay_read_image(const char *dir, uint64_t index, uint8_t *gray, int64_t cap, int32_t *rows, int32_t *cols) {
  if (!dir || !rows || !cols || cap < 0 || (cap > 0 && !gray)) return VIO_EINVAL;
  std::vector<uint8_t> file;
  if (!read_file(index_path(dir, index).c_str(), file)) return VIO_EINVAL;
  return decode_png_gray(file, gray, cap, rows, cols);
}

int vio_replay_write_image(const char *dir, uint64_t index, const uint8_t *pixels, int32_t rows, int32_t cols,
                           int32_t channels) {
  if (!dir || !pixels || rows < 1 || cols < 1 || !(channels == 1 || channels == 3 || channels == 4)) return VIO_EINVAL;
  const size_t stride = (size_t)cols * channels;
  std::vector<uint8_t> raw((stride + 1) * rows);
  for (int y = 0; y < rows; y++) {
    raw[(stride + 1) * y] = 0;  // filter type None
    memcpy(&raw[(stride + 1) * y + 1], pixels + stride * y, stride);
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 1) != Z_OK) return VIO_ENOMEM;
  std::vector<uint8_t> out(kPngSig, kPngSig + 8);
  uint8_t ihdr[13];
  put_be32(ihdr, (uint32_t)cols), put_be32(ihdr + 4, (uint32_t)rows);
  ihdr[8] = 8, ihdr[9] = channels == 1 ? 0 : (channels == 3 ? 2 : 6), ihdr[10] = 0, ihdr[11] = 0, ihdr[12] = 0;
  png_chunk(out, "IHDR", ihdr, 13);
  png_chunk(out, "IDAT", z.data(), (uint32_t)zlen);
  png_chunk(out, "IEND", nullptr, 0);
  return write_file(index_path(dir, index).c_str(), out.data(), out.size()) ? VIO_OK : VIO_EINVAL;
}

int vio_replay_rgba_to_gray(const uint8_t *rgba, int32_t rows, int32_t cols, int32_t stride, uint8_t *gray) {
  if (!rgba || !gray || rows < 1 || cols < 1 || stride < 4 * cols) return VIO_EINVAL;
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      const uint8_t *p = rgba + (size_t)y * stride + 4 * x;
      gray[(size_t)y * cols + x] = rgb_to_gray(p[0], p[1], p[2]);
    }
  return VIO_OK;
}

int vio_replay_read_keyframes(const char *path, VioKeyframeData *out, int32_t cap, int32_t *n) {
  if (!path || !n || cap < 0 || (cap > 0 && !out)) return VIO_EINVAL;
  std::vector<uint8_t> file;
  if (!read_file(path, file) || file.size() % sizeof(VioKeyframeData) != 0) return VIO_EINVAL;
  const size_t total = file.size() / sizeof(VioKeyframeData);
  if (cap > 0) memcpy(out, file.data(), std::min(total, (size_t)cap) * sizeof(VioKeyframeData));
  *n = (int32_t)total;
  return (int64_t)total <= cap || cap == 0 ? VIO_OK : VIO_ECAP;
}

int vio_replay_write_keyframes(const char *path, const VioKeyframeData *kf, int32_t n) {
  if (!path || n < 0 || (n > 0 && !kf)) return VIO_EINVAL;
  return write_file(path, kf, (size_t)n * sizeof(VioKeyframeData)) ? VIO_OK : VIO_EINVAL;
}

// ---- getMeasurements (ViewController.mm:603-638) -------------------------------------------------------------------
int vio_measurements_create(vio_measurements_t **out) {
  if (!out) return VIO_EINVAL;
  *out = new (std::nothrow) vio_measurements();
  return *out ? VIO_OK : VIO_ENOMEM;
}

void vio_measurements_destroy(vio_measurements_t *q) { delete q; }

int vio_measurements_push_imu(vio_measurements_t *q, const VioImuMsg *msg) {
  if (!q || !msg) return VIO_EINVAL;
  q->imu.push_back(*msg);
  return VIO_OK;
}

int vio_measurements_push_image(vio_measurements_t *q, double header, const VioObs *obs, int32_t n_obs) {
  if (!q || n_obs < 0 || (n_obs > 0 && !obs)) return VIO_EINVAL;
  ImgMsg m;
  m.header = header;
  m.obs.assign(obs, obs + n_obs);
  q->img.push_back(std::move(m));
  return VIO_OK;
}

// One (IMU batch, image) pair per call, in the order the reference's loop emits them. *available = 0: nothing
// complete yet ("wait for imu"). dt[i] is what send_imu hands to processIMU for sample i (0 for the very first one).
int vio_measurements_next(vio_measurements_t *q, VioImuMsg *imu, double *dt, int32_t cap_imu, int32_t *n_imu, double *header,
                          VioObs *obs, int32_t cap_obs, int32_t *n_obs, int32_t *available) {
  if (!q || !n_imu || !header || !n_obs || !available || cap_imu < 0 || cap_obs < 0 || (cap_imu > 0 && !imu) ||
      (cap_obs > 0 && !obs))
    return VIO_EINVAL;
  *available = 0, *n_imu = 0, *n_obs = 0;
  while (true) {
    if (q->imu.empty() || q->img.empty()) return VIO_OK;
    if (!(q->imu.back().header > q->img.front().header)) return VIO_OK;  // wait for imu
    if (!(q->imu.front().header < q->img.front().header)) {             // throw img
      q->img.pop_front();
      continue;
    }
    break;
  }
  const ImgMsg &im = q->img.front();
  int32_t k = 0;
  for (const VioImuMsg &m : q->imu) {
    if (!(m.header <= im.header)) break;
    k++;
  }
  if (k > cap_imu || (int32_t)im.obs.size() > cap_obs) {
    *n_imu = k, *n_obs = (int32_t)im.obs.size();
    return VIO_ECAP;  // nothing consumed
  }
  for (int32_t i = 0; i < k; i++) {
    imu[i] = q->imu.front();
    q->imu.pop_front();
    if (q->current_time < 0) q->current_time = imu[i].header;
    if (dt) dt[i] = imu[i].header - q->current_time;
    q->current_time = imu[i].header;
  }
  *n_imu = k, *header = im.header, *n_obs = (int32_t)im.obs.size();
  if (*n_obs) memcpy(obs, im.obs.data(), sizeof(VioObs) * im.obs.size());
  q->img.pop_front();
  *available = 1;
  return VIO_OK;
}

}  // extern "C"
