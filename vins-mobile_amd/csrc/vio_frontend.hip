// vio_frontend.hip — gfx950 kernels and C ABI of the KLT front-end (include/vio_amd.h).
//
// Drop-in for FeatureTracker::readImage (VINS_ios/feature_tracker.cpp:162-321) over a batch of independent sequences:
// every kernel takes (sequence, item) grids so one set of launches advances all trackers by one frame.
//   pyr_down_kernel        cv::pyrDown levels of calcOpticalFlowPyrLK                  (feature_tracker.cpp:181)
//   lk_track_kernel        calcOpticalFlowPyrLK, one wave64 per feature, all levels    (feature_tracker.cpp:181)
//   track_update_kernel    inBorder + reduceVector + findFundamentalMat(RANSAC) [+ rejectWithF, track_cnt++, setMask]
//                                                                                      (feature_tracker.cpp:183-205,235-255)
//   detect_kernel          goodFeaturesToTrack front half, fused: cornerMinEigenVal, masked maximum, 3x3 non-max and
//                          the setMask discs evaluated analytically (cv::circle(mask, pt, MIN_DIST, 0, -1))
//                                                                                      (feature_tracker.cpp:80,263)
//   corner_select_kernel   quality threshold, sorted greedy min-distance pick, addPoints, updateID, image_msg
//                                                                                      (feature_tracker.cpp:263-307)
//   copy_frames_kernel     forw_img = _img                                              (feature_tracker.cpp:165-170)
// Arithmetic follows the OpenCV 3.0 integer/float sequences restated in oracle/vio_oracle_frontend.cpp so the two
// agree bit for bit; sums that OpenCV accumulates in float (LK's A and b) are accumulated exactly.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "vio_amd.h"
#include "vio_device.h"
#include "vio_pool.h"

namespace {

constexpr int kMaxLevels = 4;   // level 0..3
constexpr int kMaxCap = 512;    // max tracked features per sequence supported by the per-sequence kernels
constexpr int kWin = 21;        // LK window (the kernel is specialised for 21x21; cfg->lk_win must match)
constexpr int kWBits = 14;
constexpr int kMaxRadius = 128;  // largest MIN_DIST (setMask circle radius) the tracker's LDS tables are sized for
#ifndef VIO_LK_FPW
#define VIO_LK_FPW 1
#endif
constexpr int kLkFpw = VIO_LK_FPW;  // features per wave of lk_track_kernel (1 or 2)

#define HIP_OK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "vio_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VIO_ENODEV;                                                                   \
    }                                                                                      \
  } while (0)

struct LevelDims {
  int rows[kMaxLevels], cols[kMaxLevels];
  size_t off[kMaxLevels];  // byte offset of the level inside one pyramid
  size_t pyr_bytes;
  int levels;              // number of levels actually used (maxLevel + 1)
};

__device__ __forceinline__ int reflect101(int p, int len) {
  // valid for |overshoot| < len (callers stay within a window of the image)
  p = p < 0 ? -p : p;
  p = p >= len ? 2 * len - 2 - p : p;
  return p;
}
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// ---- pyrDown: separable [1 4 6 4 1], BORDER_REFLECT_101, (v + 128) >> 8 -------------------------------
// One workgroup = 64x8 output pixels: the (2*64+4) x (2*8+4) source patch goes through LDS once, the horizontal pass
// leaves 20 rows of integer sums in LDS, the vertical pass writes the tile.
constexpr int kPdW = 64, kPdH = 8;
__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t *src_base, uint8_t *dst_base, size_t seq_stride,
                                                       int srows, int scols, int drows, int dcols) {
  constexpr int SW = 2 * kPdW + 4, SH = 2 * kPdH + 4;
  __shared__ int patch[SH][SW + 1];  // one dword per pixel: sub-dword LDS accesses are slow on this hardware
  __shared__ int hsum[SH][kPdW + 1];
  const uint8_t *src = src_base + (size_t)blockIdx.z * seq_stride;
  uint8_t *dst = dst_base + (size_t)blockIdx.z * seq_stride;
  const int ox = blockIdx.x * kPdW, oy = blockIdx.y * kPdH;
  const int tid = threadIdx.x;
  for (int e = tid; e < SH * SW; e += 256) {
    int ly = e / SW, lx = e - ly * SW;
    int y = reflect101(min(2 * oy + ly - 2, 2 * srows - 2), srows), x = reflect101(min(2 * ox + lx - 2, 2 * scols - 2), scols);
    patch[ly][lx] = src[(size_t)y * scols + x];
  }
  __syncthreads();
  for (int e = tid; e < SH * kPdW; e += 256) {
    int ly = e / kPdW, lx = e - ly * kPdW;
    const int *r = &patch[ly][2 * lx];
    hsum[ly][lx] = r[0] + 4 * r[1] + 6 * r[2] + 4 * r[3] + r[4];
  }
  __syncthreads();
  for (int e = tid; e < kPdH * kPdW; e += 256) {
    int ly = e / kPdW, lx = e - ly * kPdW;
    int x = ox + lx, y = oy + ly;
    if (x < dcols && y < drows) {
      int v = hsum[2 * ly][lx] + 4 * hsum[2 * ly + 1][lx] + 6 * hsum[2 * ly + 2][lx] + 4 * hsum[2 * ly + 3][lx] + hsum[2 * ly + 4][lx];
      dst[(size_t)y * dcols + x] = (uint8_t)((v + 128) >> 8);
    }
  }
}

// ---- pyramidal LK -------------------------------------------------------------------------------------
// Smallest float threshold such that fl(dx*dx + dy*dy) > threshold implies dx^2 + dy^2 > eps_sq in exact arithmetic (the float
// sum is within 2^-22 relative of the exact one; the margin is 1e-5).
inline float lk_eps_screen(double eps_sq) { return nextafterf((float)(eps_sq * (1.0 + 1e-5)), INFINITY); }
struct LkParams {
  LevelDims ld;
  int cap;             // feature slots per sequence
  int max_count;       // criteria.maxCount
  float epsilon_sq_f;  // screen of the convergence test: a float sum of squares above it cannot pass the double compare
  double epsilon_sq;
  float min_eig;
  // level 0 of either pyramid outside its pyramid buffer (resident frames are tracked where they lie: `forw_img = _img` is a
  // reference to the caller's pixels in the reference too, feature_tracker.cpp:169); null: level 0 is the pyramid's own copy
  const uint8_t *prev0, *next0;
  size_t stride0;  // bytes between the sequences' frames there
};

// Wave-wide integer sum on the DPP network (row_shr 1/2/4/8, row_bcast 15/31): no LDS round trips. The total lands
// in lane 63 and is broadcast through an SGPR.
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8   -> lane 15 of each row = row sum
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}
// Exact sum over the wave of per-lane partials |p| < 2^30: split into 16-bit low part and high part so that both
// 32-bit reductions cannot overflow; returned as a double holding the exact integer.
__device__ __forceinline__ double wave_sum_exact(int p) {
  int lo = p & 0xffff, hi = p >> 16;
  int slo = wave_sum_i32(lo), shi = wave_sum_i32(hi);
  return (double)shi * 65536.0 + (double)slo;
}

__device__ __forceinline__ void lk_weights(float a, float b, int &iw00, int &iw01, int &iw10, int &iw11) {
  iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << kWBits));
  iw01 = __float2int_rn(a * (1.f - b) * (1 << kWBits));
  iw10 = __float2int_rn((1.f - a) * b * (1 << kWBits));
  iw11 = (1 << kWBits) - iw00 - iw01 - iw10;
}

// LDS staging of one wave: the 24x24 template neighbourhood of I, its 22x22 Scharr derivative image and a
// (21+1+2*kJMargin)^2 region of J around the current estimate. Waves of a workgroup are independent (one feature
// each); LDS traffic of a wave is ordered, so a wave-level fence is all the synchronisation staging needs.
constexpr int kIP = kWin + 3;              // 24: window + 1 (bilinear) + 2 (Scharr halo)
constexpr int kDP = kWin + 1;              // 22: derivative positions
constexpr int kJMargin = 3;
constexpr int kJP = kWin + 1 + 2 * kJMargin;  // 28
constexpr int kJS = 43;  // row stride of the staged J region (dwords): the three lane groups of an iteration read rows 0 / 1 / 2 of 21 columns each -- at 43 (and 2 x 43 = 22 mod 64) their bank ranges [0, 21), [43, 64), [22, 43) do not meet (29: 15 two-way conflicts per read; lk_track 539 -> 532 us)
constexpr int kIS = kIP + 1;                  // row stride of the staged I patch (dwords)
// Every staged pixel is ONE ALIGNED DWORD holding the pair (v[x] | v[x+1] << 16): sub-dword and unaligned LDS accesses
// crawl on this hardware (the byte-array version of this kernel spent 40 % of its wave cycles in LDS issue stalls).
// A pair is also exactly one v_dot2_u32_u16 operand, so a bilinear sample is two reads and two dot instructions
// (weights < 2^15, products < 2^22).
// The template patch and its derivatives are consumed (into registers) before the first J region of a level is staged,
// so the two share their LDS: 4.3 KB per feature instead of 7.6 KB.
struct LkWaveLds {
  union {
    struct {
      uint32_t I[kIP][kIS];
      short2 dI[kDP][kDP];
    };
    uint32_t J[kJP][kJS];
  };
};
typedef short lk_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short lk_us2 __attribute__((ext_vector_type(2)));

// Four consecutive pixels from an arbitrary byte address (global memory takes unaligned dword loads on this hardware).
struct __attribute__((packed)) LkU32 {
  uint32_t v;
};
__device__ __forceinline__ uint32_t lk_load4(const uint8_t *p) { return reinterpret_cast<const LkU32 *>(p)->v; }
// Pair words (v[x] | v[x+1] << 16) of the four pixels of `cur`; the fifth pixel is byte 0 of `next`.
__device__ __forceinline__ void lk_pairs(uint32_t cur, uint32_t next, uint32_t w[4]) {
  // v_perm_b32 selects bytes of {src0 (bytes 4..7), src1 (bytes 0..3)}; selector 0x0c yields 0x00
  w[0] = __builtin_amdgcn_perm(0u, cur, 0x0c010c00u);
  w[1] = __builtin_amdgcn_perm(0u, cur, 0x0c020c01u);
  w[2] = __builtin_amdgcn_perm(0u, cur, 0x0c030c02u);
  w[3] = __builtin_amdgcn_perm(next, cur, 0x0c040c03u);
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sums over the lanes of one feature. FPW = 1: the whole wave (wave_sum_i32). FPW = 2: each half of the wave is a feature;
// the row-level DPP steps never leave a row of 16 lanes and row_bcast:15 only feeds rows 1 and 3, so the two halves do
// not mix: the totals land in lanes 31 and 63.
template <int FPW>
__device__ __forceinline__ int feat_sum_i32(int v, int sub) {
  if (FPW == 1) return wave_sum_i32(v);
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  const int lo = __builtin_amdgcn_readlane(v, 31), hi = __builtin_amdgcn_readlane(v, 63);
  return sub ? hi : lo;
}
template <int FPW>
__device__ __forceinline__ double feat_sum_exact(int p, int sub) {
  int lo = p & 0xffff, hi = p >> 16;
  int slo = feat_sum_i32<FPW>(lo, sub), shi = feat_sum_i32<FPW>(hi, sub);
  return (double)shi * 65536.0 + (double)slo;
}

// Two exact sums at once (one feature per wave): the four 32-bit chains advance in lockstep, so that every DPP step finds
// its operand written three instructions earlier (a chain on its own waits two issue slots after every step).
// (every partial below 2^24 in magnitude -- what a window of ordinary contrast gives: the bounds are 2^28 -- means the 64-lane sums fit
// 32 bits unsplit: half the chains. The split form stays for the rest; both give the exact integer, rounded once.)
__device__ __forceinline__ bool wave_all_small(int p, int q, int r = 0) {
  const bool small = (unsigned)(p + (1 << 24)) < (1u << 25) && (unsigned)(q + (1 << 24)) < (1u << 25) && (unsigned)(r + (1 << 24)) < (1u << 25);
  return __builtin_amdgcn_ballot_w64(!small) == 0ull;
}
__device__ __forceinline__ void wave_sum_exact2(int p, int q, float &sp, float &sq) {
  if (wave_all_small(p, q)) {  // (uniform)
    int u[2] = {p, q};
#define VIO_DPPS(ctrl, rmask)                                                  \
  {                                                                            \
    const int t0 = __builtin_amdgcn_update_dpp(0, u[0], ctrl, rmask, 0xf, false); \
    const int t1 = __builtin_amdgcn_update_dpp(0, u[1], ctrl, rmask, 0xf, false); \
    u[0] += t0, u[1] += t1;                                                    \
  }
    VIO_DPPS(0x111, 0xf) VIO_DPPS(0x112, 0xf) VIO_DPPS(0x114, 0xf) VIO_DPPS(0x118, 0xf) VIO_DPPS(0x142, 0xa) VIO_DPPS(0x143, 0xc)
#undef VIO_DPPS
    sp = (float)__builtin_amdgcn_readlane(u[0], 63), sq = (float)__builtin_amdgcn_readlane(u[1], 63);
    return;
  }
  int a = p & 0xffff, b = q & 0xffff, c = p >> 16, d = q >> 16;
#define VIO_DPP4(ctrl, rmask)                                              \
  {                                                                        \
    const int ta = __builtin_amdgcn_update_dpp(0, a, ctrl, rmask, 0xf, false); \
    const int tb = __builtin_amdgcn_update_dpp(0, b, ctrl, rmask, 0xf, false); \
    const int tc = __builtin_amdgcn_update_dpp(0, c, ctrl, rmask, 0xf, false); \
    const int td = __builtin_amdgcn_update_dpp(0, d, ctrl, rmask, 0xf, false); \
    a += ta, b += tb, c += tc, d += td;                                    \
  }
  VIO_DPP4(0x111, 0xf) VIO_DPP4(0x112, 0xf) VIO_DPP4(0x114, 0xf) VIO_DPP4(0x118, 0xf) VIO_DPP4(0x142, 0xa) VIO_DPP4(0x143, 0xc)
#undef VIO_DPP4
  const int sa = __builtin_amdgcn_readlane(a, 63), sb = __builtin_amdgcn_readlane(b, 63), sc = __builtin_amdgcn_readlane(c, 63),
            sd = __builtin_amdgcn_readlane(d, 63);
  // |high sum| < 2^20 and low sum < 2^22 are exact floats, the product by 2^16 is exact: the one rounding is that of the add,
  // i.e. the result is the correctly rounded float of the exact integer (= (float) of the exact double)
  sp = (float)sc * 65536.f + (float)sa, sq = (float)sd * 65536.f + (float)sb;
}

// Three at once (the 2 x 2 system's A11, A12, A22; per-lane partials |p| <= 7 * 4080^2): the sum of a ROW of 16 lanes still
// fits 32 bits (< 1.87e9), so the four in-row steps run on the three unsplit values; only the two cross-row steps need the
// 16-bit halves (six chains).
__device__ __forceinline__ void wave_sum_exact3(int p, int q, int r, float &sp, float &sq, float &sr) {
  if (wave_all_small(p, q, r)) {  // (uniform)
    int w[3] = {p, q, r};
#define VIO_DPPS3(ctrl, rmask)                                                                                       \
  {                                                                                                                   \
    int t[3];                                                                                                         \
    _Pragma("unroll") for (int k = 0; k < 3; k++) t[k] = __builtin_amdgcn_update_dpp(0, w[k], ctrl, rmask, 0xf, false); \
    _Pragma("unroll") for (int k = 0; k < 3; k++) w[k] += t[k];                                                       \
  }
    VIO_DPPS3(0x111, 0xf) VIO_DPPS3(0x112, 0xf) VIO_DPPS3(0x114, 0xf) VIO_DPPS3(0x118, 0xf) VIO_DPPS3(0x142, 0xa) VIO_DPPS3(0x143, 0xc)
#undef VIO_DPPS3
    sp = (float)__builtin_amdgcn_readlane(w[0], 63), sq = (float)__builtin_amdgcn_readlane(w[1], 63), sr = (float)__builtin_amdgcn_readlane(w[2], 63);
    return;
  }
  int u[3] = {p, q, r};
#define VIO_DPPN(N, arr, ctrl, rmask)                                               \
  {                                                                                 \
    int t[N];                                                                       \
    _Pragma("unroll") for (int k = 0; k < N; k++) t[k] = __builtin_amdgcn_update_dpp(0, arr[k], ctrl, rmask, 0xf, false); \
    _Pragma("unroll") for (int k = 0; k < N; k++) arr[k] += t[k];                   \
  }
  VIO_DPPN(3, u, 0x111, 0xf) VIO_DPPN(3, u, 0x112, 0xf) VIO_DPPN(3, u, 0x114, 0xf) VIO_DPPN(3, u, 0x118, 0xf)
  int v[6] = {u[0] & 0xffff, u[1] & 0xffff, u[2] & 0xffff, u[0] >> 16, u[1] >> 16, u[2] >> 16};
  VIO_DPPN(6, v, 0x142, 0xa) VIO_DPPN(6, v, 0x143, 0xc)
#undef VIO_DPPN
  int s[6];
#pragma unroll
  for (int k = 0; k < 6; k++) s[k] = __builtin_amdgcn_readlane(v[k], 63);
  sp = (float)s[3] * 65536.f + (float)s[0], sq = (float)s[4] * 65536.f + (float)s[1], sr = (float)s[5] * 65536.f + (float)s[2];
}

// FPW features per wave (64 / FPW lanes each). prev/next pyramids: per sequence `pyr_bytes` apart. pts arrays:
// [seq][cap][2]. Two features per wave share every wave-uniform instruction (the reductions, the 2x2 solve, the
// convergence tests, the bilinear weights): the kernel is VALU-issue-bound and those are half of an LK iteration.
// Occupancy: the kernel is latency-bound on its dependent chains (LDS round trips, DPP reductions), not on VALU issue —
// two features per wave (FPW = 2: half the wave-uniform instructions per feature, but 154 VGPRs = 3 waves per SIMD) is
// SLOWER (1.36 vs 1.28 ms per front-end step), more resident waves are faster: with the LDS per feature down to 4.3 KB the
// register count is what limits residency, so the kernel is compiled for 6 waves per SIMD (80 VGPRs, no spills; 95 -> 5
// waves before): front-end step 1.28 -> 1.17 ms. (8 waves per SIMD = 64 VGPRs spills 17 registers and gains another 0.5 %.)
template <int FPW, bool STATS = false>
__global__ __launch_bounds__(256, FPW == 1 ? 6 : 1) void lk_track_kernel(const uint8_t *prev_pyr, const uint8_t *next_pyr, LkParams P,
                                                       const int *n_pts, const float *prev_pts, float *next_pts,
                                                       uint8_t *status, float *err, unsigned long long *stats) {
  constexpr int LPF = 64 / FPW;  // lanes per feature
  __shared__ LkWaveLds lds_all[4 * FPW];
  const int seq = blockIdx.y;
  const int wave = threadIdx.x >> 6, wlane = threadIdx.x & 63;
  const int sub = wlane / LPF, lane = wlane & (LPF - 1);  // feature within the wave, lane within the feature
  const int pt = (blockIdx.x * (blockDim.x >> 6) + wave) * FPW + sub;
  if (pt >= n_pts[seq]) return;
  LkWaveLds &L = lds_all[wave * FPW + sub];
  const uint8_t *pp = prev_pyr + (size_t)seq * P.ld.pyr_bytes, *np = next_pyr + (size_t)seq * P.ld.pyr_bytes;
  const size_t pidx = ((size_t)seq * P.cap + pt) * 2;
  const float ptx = prev_pts[pidx], pty = prev_pts[pidx + 1];
  const float half = (kWin - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (1 << 20);
  constexpr int NPX = (kWin * kWin + LPF - 1) / LPF;  // window pixels per lane: 7 (14 with two features per wave)
  // This lane's window pixels. One feature per wave: lane l owns column l % 21 of rows l / 21 + 3 q (q = 0..6; 63 lanes) --
  // the q-th pixel sits a CONSTANT 3 q rows below the first, so the LDS reads of an iteration share one address register
  // (+ immediate offsets) instead of one address computation per pixel. Two features per wave: pixel lane + 32 q of the
  // row-major window. (The sums over the window are exact integers: which lane owns which pixel does not reach the result.)
  constexpr bool COLS = FPW == 1;
  const int wy0 = min(lane / kWin, 2), wx0 = lane % kWin;  // (lane 63 owns nothing: it shadows lane 62's rows, masked below)
  auto pix_ok = [&](int q) { return COLS ? lane < 3 * kWin : lane + LPF * q < kWin * kWin; };
  auto pix_y = [&](int q) { return COLS ? wy0 + 3 * q : min(lane + LPF * q, kWin * kWin - 1) / kWin; };  // (always inside the window)
  auto pix_x = [&](int q) { return COLS ? wx0 : min(lane + LPF * q, kWin * kWin - 1) % kWin; };
  int joff[COLS ? 1 : NPX];  // element offsets into the staged J region (COLS: of the first pixel)
#pragma unroll
  for (int q = 0; q < (COLS ? 1 : NPX); q++) joff[q] = pix_y(q) * kJS + pix_x(q);
  bool st = true;
  float er = 0.f;
  float nxx = 0.f, nxy = 0.f;
  const int max_level = P.ld.levels - 1;
  for (int level = max_level; level >= 0; level--) {
    const int rows = P.ld.rows[level], cols = P.ld.cols[level];
    const uint8_t *I = pp + P.ld.off[level], *J = np + P.ld.off[level];
    if (level == 0) {  // (scalar selects)
      if (P.prev0) I = P.prev0 + (size_t)seq * P.stride0;
      if (P.next0) J = P.next0 + (size_t)seq * P.stride0;
    }
    const float scale = __int_as_float((127 - level) << 23);  // (float)(1. / (1 << level)) = 2^-level, without the double division
    float px = ptx * scale, py = pty * scale;
    float qx, qy;
    if (level == max_level) qx = px, qy = py;
    else qx = nxx * 2.f, qy = nxy * 2.f;
    nxx = qx, nxy = qy;
    px -= half, py -= half;
    int ipx = (int)floorf(px), ipy = (int)floorf(py);
    if (ipx < -kWin || ipx >= cols || ipy < -kWin || ipy >= rows) {
      if (level == 0) st = false, er = 0.f;
      continue;
    }
    float a = px - ipx, b = py - ipy;
    int iw00, iw01, iw10, iw11;
    lk_weights(a, b, iw00, iw01, iw10, iw11);
    // stage I on [ipx-1, ipx+23) x [ipy-1, ipy+23) (BORDER_REFLECT_101), then its Scharr derivatives on
    // [ipx, ipx+22) x [ipy, ipy+22): zero outside the image (copyMakeBorder BORDER_CONSTANT of derivI)
    wave_lds_fence();  // previous level's readers are done
    if (FPW == 1 && ipx >= 1 && ipx + kIP - 1 <= cols && ipy >= 1 && ipy + kIP - 1 <= rows) {
      // the patch lies inside the image (all but the features within a window of the border): FOUR pixels per lane and
      // load, 6 lanes per row, 10 rows per trip -- 3 trips instead of 12 (the kernel is VALU-issue-bound: staging the two
      // patches byte by byte was a quarter of a level's instructions)
      const int r = lane / 6, d = lane - 6 * r;
      constexpr int TI = (kIP + 9) / 10;
      uint32_t curs[TI];
#pragma unroll
      for (int t = 0; t < TI; t++) {  // (the loads of all trips in flight together: one global round trip per patch, not one per trip)
        const int ly = r + 10 * t;
        curs[t] = lk_load4(I + (unsigned)(__mul24(ipy - 1 + (ly < kIP ? ly : 0), cols) + ipx - 1 + 4 * d));
      }
#pragma unroll
      for (int t = 0; t < TI; t++) {  // (uniform trip count: the DPP below needs every lane)
        const int ly = r + 10 * t;
        const bool ok = lane < 60 && ly < kIP;
        const uint32_t cur = curs[t];
        uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur, 0x130, 0xf, 0xf, false);  // dword of lane + 1
        if (d == 5) next = cur >> 24;  // (the pair behind the last pixel repeats it, like the byte path's clamped lane)
        uint32_t w4[4];
        lk_pairs(cur, next, w4);
        if (ok) {
#pragma unroll
          for (int k = 0; k < 4; k++) L.I[ly][4 * d + k] = w4[k];
        }
      }
    } else {  // 32 lanes per row (24 used), two rows per trip: the column index is reflected once per lane; the right-hand
       // neighbour of every pixel comes from the next lane (DPP wave_shl) so that a pair can be stored as one dword
      const int lx = lane & 31;
      const int x = reflect101(ipx - 1 + min(lx, kIP - 1), cols);
      // (a third of the features take this path at the coarse levels, where the patch is a quarter of the image: all loads
      // of the patch are issued before the first is consumed -- one memory round trip instead of one per trip)
      constexpr int RPT = LPF / 32, TB = kIP / RPT;
      static_assert(kIP % RPT == 0, "whole trips");
      int vs[TB];
#pragma unroll
      for (int t = 0; t < TB; t++) {
        if constexpr (FPW == 1) {
          // (one feature per wave: the patch origin is the same in every lane, so the reflected ROW of a trip is one of two scalars
          // -- even / odd half of the wave -- and comes off the scalar unit: 2 vector instructions per trip instead of ~16; the
          // border path is not rare: 58 % of the features take it at level 3, 32 % at level 2)
          const int s_y0 = __builtin_amdgcn_readfirstlane(ipy) - 1 + RPT * t;
          const int ro_a = reflect101(s_y0, rows) * cols, ro_b = reflect101(s_y0 + 1, rows) * cols;
          vs[t] = I[(unsigned)(((lane >> 5) ? ro_b : ro_a) + x)];
        } else {
          vs[t] = I[(unsigned)(__mul24(reflect101(ipy - 1 + (lane >> 5) + RPT * t, rows), cols) + x)];
        }
      }
#pragma unroll
      for (int t = 0; t < TB; t++) {
        const int ly = (lane >> 5) + RPT * t, v = vs[t];
        const int vr = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false);  // value of lane + 1
        if (lx < kIP) L.I[ly][lx] = (uint32_t)v | ((uint32_t)vr << 16);
      }
    }
    wave_lds_fence();
    if (FPW == 1) {
      // Scharr derivatives by COLUMN-PAIR WALK on packed 16-bit arithmetic: lanes 0..54 = 11 column pairs x five segments of five
      // output rows (the last one starts at row 17 and repeats three rows of its neighbour). A staged dword is a pixel pair
      // (I[x] | I[x+1] << 16), so the three pair words at columns c, c + 1, c + 2 are the left / centre / right taps of the derivative
      // columns c AND c + 1 at once: per input row h = P2 - P0 and v = 3 (P0 + P2) + 10 P1, per output row dx = 3 (h0 + h2) + 10 h1 and
      // dy = v2 - v0, each ONE v_pk_* instruction for both columns (|values| <= 16 x 255: exact in 16 bits) -- the same integers as the
      // 3 x 3 sums of calcSharrDeriv. (The one-column walk on 32-bit values, rounds 4-6: 168 vector instructions per level on 44 lanes,
      // a seventh of the kernel's; this form: ~75 on 55 lanes.)
      constexpr int SEG = 5, NSEG = 5, NCP = kDP / 2;
      static_assert(kDP % 2 == 0 && NCP * NSEG <= 64 && SEG * NSEG >= kDP && kDP >= SEG, "column pairs x segments");
      const bool inside = ipx >= 0 && ipx + kDP <= cols && ipy >= 0 && ipy + kDP <= rows;  // every derivative position is in the image
      if (lane < NCP * NSEG) {
        const int seg = lane / NCP, cp = lane - NCP * seg, c = 2 * cp;
        const int r0 = seg * SEG < kDP - SEG ? seg * SEG : kDP - SEG;
        const uint32_t *Ip = &L.I[0][0] + (__mul24(r0, kIS) + c);
        uint32_t *Dp = reinterpret_cast<uint32_t *>(&L.dI[0][0]) + (__mul24(r0, kDP) + c);
        const lk_s2 k3 = {3, 3}, k10 = {10, 10};
        auto walk = [&](auto checked) {  // (checked: the patch leaves the image -- derivI's BORDER_CONSTANT zeros)
          const unsigned ok0 = (unsigned)(ipx + c) < (unsigned)cols, ok1 = (unsigned)(ipx + c + 1) < (unsigned)cols;
          lk_s2 P0[SEG + 2], P1[SEG + 2], P2[SEG + 2];  // pair words of input row r0 + r: the staged patch holds reflected values at the image border
#pragma unroll
          for (int r = 0; r < SEG + 2; r++)  // (all reads ahead of the first write: one wait instead of one per row)
            __builtin_memcpy(&P0[r], Ip + r * kIS, 4), __builtin_memcpy(&P1[r], Ip + r * kIS + 1, 4), __builtin_memcpy(&P2[r], Ip + r * kIS + 2, 4);
          lk_s2 h[SEG + 2], v[SEG + 2];
#pragma unroll
          for (int r = 0; r < SEG + 2; r++) h[r] = P2[r] - P0[r], v[r] = (P0[r] + P2[r]) * k3 + P1[r] * k10;
#pragma unroll
          for (int r = 2; r < SEG + 2; r++) {
            const lk_s2 dx = (h[r - 2] + h[r]) * k3 + h[r - 1] * k10, dy = v[r] - v[r - 2];
            uint32_t ux, uy;
            __builtin_memcpy(&ux, &dx, 4), __builtin_memcpy(&uy, &dy, 4);
            uint32_t d0 = __builtin_amdgcn_perm(uy, ux, 0x05040100u), d1 = __builtin_amdgcn_perm(uy, ux, 0x07060302u);  // short2 {dx, dy} of columns c, c + 1
            if (decltype(checked)::value) {
              const unsigned rok = (unsigned)(ipy + r0 + r - 2) < (unsigned)rows;
              d0 &= 0u - (ok0 & rok), d1 &= 0u - (ok1 & rok);
            }
            Dp[(r - 2) * kDP] = d0, Dp[(r - 2) * kDP + 1] = d1;
          }
        };
        if (inside) walk(std::false_type{});
        else walk(std::true_type{});
      }
    } else {
      {
        // element e = lane + LPF t of the 22 x 22 derivative patch, (ly, lx) advanced by (LPF / 22, LPF % 22) per trip (an
        // integer division and the 32-bit multiplies of the Scharr sums are quarter-rate instructions: none are left here)
        int e = lane, ly = lane / kDP, lx = lane - kDP * ly;
        const bool inside = ipx >= 0 && ipx + kDP <= cols && ipy >= 0 && ipy + kDP <= rows;  // every derivative position is in the image
#pragma unroll 1
        for (int t = 0; t < (kDP * kDP + LPF - 1) / LPF; t++) {
          if (ly < kDP) {
            uint32_t d = 0;
            if (inside || (ipx + lx >= 0 && ipx + lx < cols && ipy + ly >= 0 && ipy + ly < rows)) {
              // the staged patch holds reflected values, i.e. exactly what calcSharrDeriv reads at the image border
              const uint32_t *Ip = &L.I[0][0] + (e + __mul24(kIS - kDP, ly));  // = &L.I[ly][lx]
              const uint32_t a0 = Ip[0], a1 = Ip[1], b0 = Ip[kIS], b1 = Ip[kIS + 1], c0 = Ip[2 * kIS], c1 = Ip[2 * kIS + 1];
              const int p00 = a0 & 0xffff, p01 = a0 >> 16, p02 = a1 >> 16;
              const int p10 = b0 & 0xffff, p12 = b1 >> 16;
              const int p20 = c0 & 0xffff, p21 = c0 >> 16, p22 = c1 >> 16;
              const int dx = __mul24(3, (p02 - p00) + (p22 - p20)) + __mul24(10, p12 - p10);
              const int dy = __mul24(3, (p20 - p00) + (p22 - p02)) + __mul24(10, p21 - p01);
              d = __builtin_amdgcn_perm((uint32_t)dy, (uint32_t)dx, 0x05040100u);  // short2 {dx, dy}
            }
            __builtin_memcpy(&L.dI[0][0] + e, &d, 4);  // = L.dI[ly][lx]
          }
          e += LPF, lx += LPF % kDP, ly += LPF / kDP;
          if (lx >= kDP) lx -= kDP, ly++;
        }
      }
    }
    wave_lds_fence();
    // template patch + derivatives for this lane's window pixels, kept in registers across the iterations
    // (the template value enters the iterations as the accumulator seed of the bilinear sample: rounding constant minus
    // Iv << n, so that the arithmetic shift that ends the sample yields J - I directly -- (a + r - (Iv << n)) >> n ==
    // ((a + r) >> n) - Iv exactly)
    // Ix, Iy of pixels 2h and 2h + 1 share a register: the iterations multiply-accumulate them pairwise (v_dot2_i32_i16).
    constexpr int NPP = (NPX + 1) / 2;
    const lk_s2 sw_top = {(short)iw00, (short)iw01}, sw_bot = {(short)iw10, (short)iw11};
    lk_s2 IxP[NPP], IyP[NPP];
    int Ic[NPX];
    int p11 = 0, p12 = 0, p22 = 0;  // per-lane partials: <= 14 products of |v| <= 4080^2 fit 32 bits
#pragma unroll
    for (int h = 0; h < NPP; h++) IxP[h] = IyP[h] = lk_s2{0, 0};
#pragma unroll
    for (int q = 0; q < NPX; q++) {
      const int y = pix_y(q), x = pix_x(q);
      // bilinear taps as dot products of packed pairs (a 32-bit integer multiply is a quarter-rate instruction here; the
      // element-wise form of this block was 12 of them per pixel): the I patch is staged as pairs already; the derivative
      // pairs (d[x], d[x+1]) are cut out of two (dx, dy) words with v_perm_b32. Weights <= 2^14 fit int16, and the SIGNED
      // dot keeps iw11 = 2^14 - iw00 - iw01 - iw10 right when the three roundings push it to -1.
      lk_s2 it, ib;
      __builtin_memcpy(&it, &L.I[y + 1][x + 1], 4);  // (I[x+1], I[x+2]) of the two rows
      __builtin_memcpy(&ib, &L.I[y + 2][x + 1], 4);
      const int ival = __builtin_amdgcn_sdot2(it, sw_top, __builtin_amdgcn_sdot2(ib, sw_bot, 1 << (kWBits - 5 - 1), false), false) >> (kWBits - 5);
      uint32_t d00, d01, d10, d11;
      __builtin_memcpy(&d00, &L.dI[y][x], 4), __builtin_memcpy(&d01, &L.dI[y][x + 1], 4);
      __builtin_memcpy(&d10, &L.dI[y + 1][x], 4), __builtin_memcpy(&d11, &L.dI[y + 1][x + 1], 4);
      const uint32_t xt = __builtin_amdgcn_perm(d01, d00, 0x05040100u), yt = __builtin_amdgcn_perm(d01, d00, 0x07060302u);
      const uint32_t xb = __builtin_amdgcn_perm(d11, d10, 0x05040100u), yb = __builtin_amdgcn_perm(d11, d10, 0x07060302u);
      lk_s2 sxt, syt, sxb, syb;
      __builtin_memcpy(&sxt, &xt, 4), __builtin_memcpy(&syt, &yt, 4), __builtin_memcpy(&sxb, &xb, 4), __builtin_memcpy(&syb, &yb, 4);
      int ixval = __builtin_amdgcn_sdot2(sxt, sw_top, __builtin_amdgcn_sdot2(sxb, sw_bot, 1 << (kWBits - 1), false), false) >> kWBits;
      int iyval = __builtin_amdgcn_sdot2(syt, sw_top, __builtin_amdgcn_sdot2(syb, sw_bot, 1 << (kWBits - 1), false), false) >> kWBits;
      if (!pix_ok(q)) ixval = iyval = 0;  // a pixel this lane does not own: weight zero in every sum below
      Ic[q] = (1 << (kWBits - 5 - 1)) - (int)((unsigned)ival << (kWBits - 5));
      if (q & 1) IxP[q >> 1].y = (short)ixval, IyP[q >> 1].y = (short)iyval;
      else IxP[q >> 1].x = (short)ixval, IyP[q >> 1].x = (short)iyval;
      p11 += ixval * ixval, p12 += ixval * iyval, p22 += iyval * iyval;
    }
    float f11, f12, f22;  // exact integer sums, rounded once to float
    if (FPW == 1) wave_sum_exact3(p11, p12, p22, f11, f12, f22);
    else f11 = (float)feat_sum_exact<FPW>(p11, sub), f12 = (float)feat_sum_exact<FPW>(p12, sub), f22 = (float)feat_sum_exact<FPW>(p22, sub);
    float A11 = f11 * FLT_SCALE, A12 = f12 * FLT_SCALE, A22 = f22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * kWin * kWin);
    if (minEig < P.min_eig || D < 1.1920929e-07f) {
      if (level == 0) st = false;
      continue;
    }
    D = 1.f / D;
    qx -= half, qy -= half;
    float pdx = 0.f, pdy = 0.f;
    int jox = 0, joy = 0;      // origin of the staged J region
    bool j_staged = false;
    auto stage_j = [&](int iqx, int iqy) {
      jox = iqx - kJMargin, joy = iqy - kJMargin;
      wave_lds_fence();
      if (FPW == 1 && jox >= 0 && jox + kJP <= cols && joy >= 0 && joy + kJP <= rows) {
        // inside the image: 4 pixels per lane and load, 7 lanes per row, 9 rows per trip -- 4 trips instead of 14
        const int r = lane / 7, d = lane - 7 * r;
        constexpr int TJ = (kJP + 8) / 9;
        uint32_t curs[TJ];
#pragma unroll
        for (int t = 0; t < TJ; t++) {
          const int ly = r + 9 * t;
          curs[t] = lk_load4(J + (unsigned)(__mul24(joy + (ly < kJP ? ly : 0), cols) + jox + 4 * d));
        }
#pragma unroll
        for (int t = 0; t < TJ; t++) {
          const int ly = r + 9 * t;
          const bool ok = lane < 63 && ly < kJP;
          const uint32_t cur = curs[t];
          uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur, 0x130, 0xf, 0xf, false);
          if (d == 6) next = cur >> 24;
          uint32_t w4[4];
          lk_pairs(cur, next, w4);
          if (ok) {
#pragma unroll
            for (int k = 0; k < 4; k++) L.J[ly][4 * d + k] = w4[k];
          }
        }
      } else {  // 32 lanes per row (28 used), two rows per trip; pairs (J[x] | J[x+1] << 16) like the template patch
        const int lx = lane & 31;
        const int x = reflect101(min(max(jox + min(lx, kJP - 1), -cols + 1), 2 * cols - 2), cols);
        constexpr int RPT = LPF / 32, TB = kJP / RPT;
        static_assert(kJP % RPT == 0, "whole trips");
        int vs[TB];
#pragma unroll
        for (int t = 0; t < TB; t++) {
          if constexpr (FPW == 1) {  // (rows on the scalar unit, as in the template patch's border path)
            const int s_y0 = __builtin_amdgcn_readfirstlane(joy) + RPT * t;
            const int ro_a = reflect101(min(max(s_y0, -rows + 1), 2 * rows - 2), rows) * cols;
            const int ro_b = reflect101(min(max(s_y0 + 1, -rows + 1), 2 * rows - 2), rows) * cols;
            vs[t] = J[(unsigned)(((lane >> 5) ? ro_b : ro_a) + x)];
          } else {
            const int y = reflect101(min(max(joy + (lane >> 5) + RPT * t, -rows + 1), 2 * rows - 2), rows);
            vs[t] = J[(unsigned)(__mul24(y, cols) + x)];
          }
        }
#pragma unroll
        for (int t = 0; t < TB; t++) {
          const int ly = (lane >> 5) + RPT * t, v = vs[t];
          const int vr = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false);  // value of lane + 1
          if (lx < kJP) L.J[ly][lx] = (uint32_t)v | ((uint32_t)vr << 16);
        }
      }
      wave_lds_fence();
      j_staged = true;
    };
    const uint32_t *Jl = &L.J[0][0];
    // (the top-row weights are never negative: their dot is the unsigned three-operand instruction, seeded with Ic without
    // a copy; the bottom row, whose iw11 can be -1, goes through the signed accumulate-in-place one)
    auto diff_j = [&](int base, int q, lk_us2 wtop, lk_s2 wbot) {  // bilinear J at this lane's q-th window pixel, minus I there
      lk_us2 top;
      lk_s2 bot;
      const int o = base + (COLS ? joff[0] + 3 * q * kJS : joff[COLS ? 0 : q]);  // (q is a constant after unrolling)
      const uint32_t t32 = Jl[o], b32 = Jl[o + kJS];
      __builtin_memcpy(&top, &t32, 4);
      __builtin_memcpy(&bot, &b32, 4);
      return __builtin_amdgcn_sdot2(bot, wbot, (int)__builtin_amdgcn_udot2(top, wtop, (unsigned)Ic[q], false), false) >> (kWBits - 5);  // (mod 2^32)
    };
    int nit = 0;  // (STATS only)
    for (int j = 0; j < P.max_count; j++) {
      int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
      if (iqx < -kWin || iqx >= cols || iqy < -kWin || iqy >= rows) {
        if (level == 0) st = false;
        break;
      }
      if (STATS) nit++;
      if (!j_staged || iqx < jox || iqx > jox + 2 * kJMargin || iqy < joy || iqy > joy + 2 * kJMargin) stage_j(iqx, iqy);
      a = qx - iqx, b = qy - iqy;
      lk_weights(a, b, iw00, iw01, iw10, iw11);
      int pb1 = 0, pb2 = 0;  // |diff * dI| <= 16320 * 4080 per pixel, <= 14 pixels per lane: 9.3e8 fits 32 bits
      const lk_us2 wtop = {(unsigned short)iw00, (unsigned short)iw01};
      const lk_s2 wbot = {(short)iw10, (short)iw11};
      const int jbase = __mul24(iqy - joy, kJS) + (iqx - jox);
#pragma unroll
      for (int h = 0; h < NPP; h++) {  // (a pixel the lane does not own has Ix = Iy = 0)
        const int d0 = diff_j(jbase, 2 * h, wtop, wbot), d1 = 2 * h + 1 < NPX ? diff_j(jbase, 2 * h + 1, wtop, wbot) : 0;
        const unsigned pk = __builtin_amdgcn_perm((unsigned)d1, (unsigned)d0, 0x05040100u);  // |J - I| <= 8160: (d0, d1) as two int16
        lk_s2 dp;
        __builtin_memcpy(&dp, &pk, 4);
        pb1 = __builtin_amdgcn_sdot2(dp, IxP[h], pb1, false), pb2 = __builtin_amdgcn_sdot2(dp, IyP[h], pb2, false);
      }
      float fb1, fb2;  // the exact integer sums, rounded once to float (what (float) of the exact double gave)
      if (FPW == 1) wave_sum_exact2(pb1, pb2, fb1, fb2);
      else fb1 = (float)feat_sum_exact<FPW>(pb1, sub), fb2 = (float)feat_sum_exact<FPW>(pb2, sub);
      float b1 = fb1 * FLT_SCALE, b2 = fb2 * FLT_SCALE;
      float ddx = (A12 * b2 - A22 * b1) * D, ddy = (A12 * b1 - A11 * b2) * D;
      qx += ddx, qy += ddy;
      nxx = qx + half, nxy = qy + half;
      // delta.ddot(delta) <= epsilon in double, behind a float screen that only lets candidates through (conversions to and
      // from double are slow instructions; all but the last iteration of a level stop at the screen)
      if (!(ddx * ddx + ddy * ddy > P.epsilon_sq_f) && (double)ddx * ddx + (double)ddy * ddy <= P.epsilon_sq) break;
      // |x| < 0.01 for a float x: 0.01 (double) lies strictly between 0.01f and the next float up, so the float compare against
      // that next float decides the same
      constexpr float kCentiUp = 0.010000000707805157f;
      static_assert((double)kCentiUp > 0.01 && (double)0.01f < 0.01, "float neighbours of 0.01");
      if (j > 0 && fabsf(ddx + pdx) < kCentiUp && fabsf(ddy + pdy) < kCentiUp) {
        nxx -= ddx * 0.5f, nxy -= ddy * 0.5f;
        break;
      }
      pdx = ddx, pdy = ddy;
    }
    if (STATS && lane == 0) {  // (64 copies of the counters, picked by block: one address would serialize every wave of the launch)
      unsigned long long *sl = stats + ((blockIdx.x + 7 * blockIdx.y) & 63) * 16;
      atomicAdd(sl + level, (unsigned long long)nit);
      atomicAdd(sl + P.ld.levels + level, 1ull);
    }
    if (st && level == 0) {
      float ex = nxx - half, ey = nxy - half;
      int iex = (int)floorf(ex), iey = (int)floorf(ey);
      if (iex < -kWin || iex >= cols || iey < -kWin || iey >= rows) {
        st = false;
        continue;
      }
      if (!j_staged || iex < jox || iex > jox + 2 * kJMargin || iey < joy || iey > joy + 2 * kJMargin) stage_j(iex, iey);
      float aa = ex - iex, bb = ey - iey;
      lk_weights(aa, bb, iw00, iw01, iw10, iw11);
      int pe = 0;
      const lk_us2 wtop = {(unsigned short)iw00, (unsigned short)iw01};
      const lk_s2 wbot = {(short)iw10, (short)iw11};
      const int jbase = __mul24(iey - joy, kJS) + (iex - jox);
#pragma unroll
      for (int q = 0; q < NPX; q++) {
        const int d = abs(diff_j(jbase, q, wtop, wbot));
        pe += pix_ok(q) ? d : 0;
      }
      const int se = feat_sum_i32<FPW>(pe, sub);  // <= 441 * 16320 < 2^23
      er = (float)se * 1.f / (32 * kWin * kWin);
    }
  }
  if (lane == 0) {
    next_pts[pidx] = nxx, next_pts[pidx + 1] = nxy;
    status[(size_t)seq * P.cap + pt] = st ? 1 : 0;
    err[(size_t)seq * P.cap + pt] = er;
  }
}

// ---- findFundamentalMat(FM_RANSAC): device version of calib3d fundam.cpp / ptsetreg.cpp -------------------------
__device__ int solve_cubic_dev(const double c[4], double r[3]) {
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
    const double kPi = 3.14159265358979323846;
    if (d >= 0) {
      double theta = acos(R / sqrt(Qcubed)), sqrtQ = sqrt(Q);
      double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
      x0 = t0 * cos(t1) - t2, x1 = t0 * cos(t1 + (2. * kPi / 3)) - t2, x2 = t0 * cos(t1 + (4. * kPi / 3)) - t2;
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

// 7-point models for the subset (ms1, ms2); F[27]; returns the number of models.
// Work area of one 7-point solve. The elimination indexes its 7x9 system, the two null vectors and the column
// permutation dynamically: as local arrays they live in scratch memory (an L2 round trip per access, ~2000 of them on
// the one thread that owns the hypothesis); here they sit in LDS next to the hypothesis' points (track_update_kernel
// 0.30 -> 0.17 ms per 256 sequences).
struct SevenPointWork {
  double a[63], f1[9], f2[9];
  int colperm[10];
};

// The 7-point solve by the 16 lanes of one DPP row (lane l of the group). One thread alone spends ~130 k cycles in it: the
// elimination is ~1300 DEPENDENT LDS accesses (dynamic indexing keeps the system out of registers), and the whole RANSAC
// call waits for it. Here the pivot search (4 candidates per lane, then a 4-step butterfly on (|value|, position) keys
// with the serial scan's tie rule: first position wins), the swaps, the pivot-row division and the elimination (54
// elements, every lane reads its factors and pivot-row entries before anyone writes) run across the lanes; every element
// sees exactly the operations of the serial restatement (oracle run7point), so the result is bit-identical. The cubic and the model assembly
// (~300 flops) stay on lane 0. Returns the number of models on every lane of the group.
__device__ __forceinline__ int run7point_group(const float *ms1, const float *ms2, double *fmatrix, SevenPointWork &wk, int l) {
  double *a = wk.a, *f1 = wk.f1, *f2 = wk.f2;
  int *colperm = wk.colperm;
  auto group_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  if (l < 7) {
    double x0 = ms1[2 * l], y0 = ms1[2 * l + 1], x1 = ms2[2 * l], y1 = ms2[2 * l + 1];
    double *row = a + l * 9;
    row[0] = x1 * x0, row[1] = x1 * y0, row[2] = x1, row[3] = y1 * x0, row[4] = y1 * y0, row[5] = y1, row[6] = x0,
    row[7] = y0, row[8] = 1;
  }
  if (l < 9) colperm[l] = l;
  group_sync();
  for (int k = 0; k < 7; k++) {
    // pivot: largest |a[i][j]| over i >= k, j >= k, the first one in row-major order among equals
    double best = -1;
    int bpos = k * 9 + k;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int e = l + 16 * q, i = e / 9, j = e - 9 * i;
      if (e < 63 && i >= k && j >= k) {
        const double v = fabs(a[e]);
        if (v > best) best = v, bpos = e;
      }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
      const double ob = __shfl_xor(best, m, 16);
      const int op = __shfl_xor(bpos, m, 16);
      if (ob > best || (ob == best && op < bpos)) best = ob, bpos = op;
    }
    const int pr = bpos / 9, pc = bpos - 9 * pr;
    if (pr != k && l < 9) {
      const double t = a[k * 9 + l];
      a[k * 9 + l] = a[pr * 9 + l], a[pr * 9 + l] = t;
    }
    group_sync();
    if (pc != k) {
      if (l < 7) {
        const double t = a[l * 9 + k];
        a[l * 9 + k] = a[l * 9 + pc], a[l * 9 + pc] = t;
      }
      if (l == 15) {
        const int t = colperm[k];
        colperm[k] = colperm[pc], colperm[pc] = t;
      }
    }
    group_sync();
    const double d = a[k * 9 + k];
    group_sync();
    if (d == 0.0) continue;
    if (l < 9) a[k * 9 + l] /= d;
    group_sync();
    double fv[4], pv[4], av[4];
    int ev[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {  // element q of this lane: rows i != k in order, 9 columns each
      const int e = l + 16 * q, ii = e / 9, j = e - 9 * ii, i = ii < k ? ii : ii + 1;
      const bool on = e < 54;
      ev[q] = on ? i * 9 + j : -1;
      fv[q] = on ? a[i * 9 + k] : 0.0, pv[q] = on ? a[k * 9 + j] : 0.0, av[q] = on ? a[i * 9 + j] : 0.0;
    }
    group_sync();
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (ev[q] >= 0 && fv[q] != 0.0) a[ev[q]] = av[q] - fv[q] * pv[q];
    group_sync();
  }
  if (l < 9) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      double *out = q == 0 ? f1 : f2;
      const double vj = l < 7 ? -a[l * 9 + 7 + q] : (l - 7 == q ? 1.0 : 0.0);
      out[colperm[l]] = vj;
    }
  }
  group_sync();
  if (l < 9) f1[l] -= f2[l];
  group_sync();
  int n = 0;
  if (l == 0) {
    double c[4], r[3];
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
    n = solve_cubic_dev(c, r);
    if (n >= 1 && n <= 3)
      for (int k = 0; k < n; k++, fmatrix += 9) {
        double lambda = r[k], mu = 1., s = f1[8] * r[k] + f2[8];
        if (fabs(s) > 2.220446049250313e-16) mu = 1. / s, lambda *= mu, fmatrix[8] = 1.;
        else fmatrix[8] = 0.;
        for (int i = 0; i < 8; i++) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
      }
  }
  return __shfl(n, 0, 16);
}

__device__ __forceinline__ float epipolar_error(const double *F, float x1f, float y1f, float x2f, float y2f) {
  double x1 = x1f, y1 = y1f, x2 = x2f, y2 = y2f;
  double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
  double s2 = 1. / (a * a + b * b);
  double d2 = x2 * a + y2 * b + c;
  a = F[0] * x2 + F[3] * y2 + F[6], b = F[1] * x2 + F[4] * y2 + F[7], c = F[2] * x2 + F[5] * y2 + F[8];
  double s1 = 1. / (a * a + b * b);
  double d1 = x1 * a + y1 * b + c;
  return (float)fmax(d1 * d1 * s1, d2 * d2 * s2);
}
__device__ __forceinline__ bool epipolar_inlier(const double *F, float x1f, float y1f, float x2f, float y2f, float t) {
  return epipolar_error(F, x1f, y1f, x2f, y2f) <= t;
}

__device__ bool have_collinear_dev(const float *p, int count) {
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

__device__ int ransac_update_iters(double p, double ep, int model_points, int max_iters) {
  p = fmax(p, 0.), p = fmin(p, 1.), ep = fmax(ep, 0.), ep = fmin(ep, 1.);
  double num = fmax(1. - p, 2.2250738585072014e-308), denom = 1. - pow(1. - ep, (double)model_points);
  if (denom < 2.2250738585072014e-308) return 0;
  num = log(num), denom = log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

constexpr int kHypBatch = 16;  // hypotheses generated / evaluated per round

struct RansacShared {
  SevenPointWork work[kHypBatch];
  float ms1[kHypBatch][14], ms2[kHypBatch][14];
  double F[kHypBatch][27];
  int nmodels[kHypBatch];
  int good[kHypBatch][3];
  int idx[kHypBatch][7];  // drawn point indices of the batch
  int valid[kHypBatch];  // subset found
  int coll[kHypBatch];   // speculative subset failed checkSubset
  unsigned long long rng_before[kHypBatch];
  unsigned long long rng_state;
  int niters, max_good, best_h, best_k, iter, done, failed_first;
  double bestF[9];
  double med[kHypBatch][3], min_median;  // LMedS (fewer than 15 correspondences)
};

// Block-cooperative RANSAC over `count` correspondences (m1, m2 in LDS or global). Writes mask[count] (1 = inlier).
// Exactly reproduces the sequential loop of RANSACPointSetRegistrator::run: subsets are drawn in order from one RNG
// stream; hypotheses are evaluated a batch at a time and then scanned in order with the adaptive iteration bound.
template <class MaskT>
__device__ void fundamental_ransac_block(RansacShared &S, const float *m1, const float *m2, int count, float thresh,
                                         double confidence, MaskT *mask, long long *prof = nullptr) {
  const int tid = threadIdx.x, nt = blockDim.x;
  long long pt = prof ? clock64() : 0;
#define RS_STAMP(k)                                                    \
  do {                                                                 \
    if (prof && tid == 0) {                                            \
      const long long now = clock64();                                 \
      prof[k] += now - pt, pt = now;                                   \
    }                                                                  \
  } while (0)
  const int model_points = 7, max_iters = 1000;
  // findFundamentalMat (fundam.cpp, 3.0.0) runs RANSAC only from 15 points on, LMedS below; the tracker calls with
  // >= 8 points. LMedS shares the subset stream and the 7-point solver: a fixed number of iterations (300 at
  // confidence 0.99), the model with the smallest median error wins, inliers are cut at sigma^2 of that median.
  const bool lmeds = count < 15;
  if (count < 8) {
    for (int i = tid; i < count; i += nt) mask[i] = 1;
    __syncthreads();
    return;
  }
  if (tid == 0) {
    S.rng_state = 0xffffffffffffffffULL;
    S.niters = lmeds ? ransac_update_iters(confidence, 0.45, model_points, max_iters) : max_iters;
    S.max_good = 0, S.best_h = -1, S.best_k = 0, S.iter = 0, S.done = 0, S.failed_first = 0;
    S.min_median = 1.7976931348623157e308;
  }
  __syncthreads();
  const float t = thresh * thresh;
  while (true) {
    // hypotheses of this round. The adaptive bound usually ends a RANSAC call within its first handful of hypotheses (inlier
    // ratio 0.9: 7 iterations), and every hypothesis of a round is drawn, solved and scored whether the scan reaches it or
    // not: the first round takes 8, later rounds (and LMedS, whose iteration count is fixed) kHypBatch.
    const int nb = (!lmeds && S.iter == 0) ? 8 : kHypBatch;
    // ---- phase 1: draw the next nb subsets sequentially (one RNG stream). The draws are cheap integer
    //      work; checkSubset (collinearity, 2 x 15 cross products in double) is evaluated in parallel afterwards.
    //      Subsets are drawn speculatively as if every check passed, which is the same RNG consumption; if one fails
    //      (rare) the tail from that hypothesis is redrawn with the full serial semantics.
    auto draw = [&](unsigned long long &st, int h, bool with_check) {
      int idx[7], i = 0, iters = 0;
      const int max_attempts = 10000;
      for (; iters < max_attempts; iters++) {
        for (i = 0; i < model_points && iters < max_attempts;) {
          int idx_i = 0;
          for (;;) {
            st = (unsigned long long)(unsigned)st * 4164903690U + (unsigned)(st >> 32);
            idx_i = idx[i] = (int)((unsigned)st % (unsigned)count);
            int j;
            for (j = 0; j < i; j++)
              if (idx_i == idx[j]) break;
            if (j == i) break;
          }
          S.ms1[h][2 * i] = m1[2 * idx_i], S.ms1[h][2 * i + 1] = m1[2 * idx_i + 1];
          S.ms2[h][2 * i] = m2[2 * idx_i], S.ms2[h][2 * i + 1] = m2[2 * idx_i + 1];
          i++;
        }
        if (with_check && i == model_points && (have_collinear_dev(S.ms1[h], i) || have_collinear_dev(S.ms2[h], i))) continue;
        break;
      }
      return (i == model_points && iters < max_attempts) ? 1 : 0;
    };
    if (tid == 0) {
      // the serial part is the RNG stream only: indices go to LDS, the points are gathered by all threads afterwards.
      // x % count without an integer division: q = umulhi(x, floor((2^32 - 1) / count)) underestimates the quotient
      // by at most 2. (Pinning the stream to scalar registers -- the 112 dependent steps on the scalar unit -- measured the
      // same 21 k cycles per call.)
      const unsigned ucount = (unsigned)count, magic = 0xffffffffu / ucount;
      unsigned long long st = S.rng_state;
      for (int h = 0; h < nb; h++) {
        S.rng_before[h] = st;
        int idx[7];
#pragma unroll
        for (int i = 0; i < 7; i++) {
          for (;;) {
            st = (unsigned long long)(unsigned)st * 4164903690U + (unsigned)(st >> 32);
            const unsigned x = (unsigned)st;
            unsigned r = x - __umulhi(x, magic) * ucount;
            r = r >= ucount ? r - ucount : r;
            r = r >= ucount ? r - ucount : r;
            bool dup = false;
#pragma unroll
            for (int j = 0; j < i; j++) dup |= (int)r == idx[j];
            if (!dup) { idx[i] = (int)r; break; }
          }
          S.idx[h][i] = idx[i];
        }
        S.valid[h] = 1;  // (count >= 8 distinct points exist: the attempt cap of getSubset cannot trigger without checks)
      }
      S.rng_state = st;
    }
    __syncthreads();
    RS_STAMP(0);
    for (int it = tid; it < nb * 7; it += nt) {
      const int h = it / 7, i = it - 7 * h, id = S.idx[h][i];
      S.ms1[h][2 * i] = m1[2 * id], S.ms1[h][2 * i + 1] = m1[2 * id + 1];
      S.ms2[h][2 * i] = m2[2 * id], S.ms2[h][2 * i + 1] = m2[2 * id + 1];
    }
    __syncthreads();
    if (tid < nb) S.coll[tid] = (have_collinear_dev(S.ms1[tid], 7) || have_collinear_dev(S.ms2[tid], 7)) ? 1 : 0;
    __syncthreads();
    if (tid == 0) {
      int first = -1;
      for (int h = 0; h < nb && first < 0; h++)
        if (S.coll[h]) first = h;
      if (first >= 0) {
        unsigned long long st = S.rng_before[first];
        for (int h = first; h < nb; h++) {
          S.valid[h] = draw(st, h, true);
          if (!S.valid[h]) {
            for (int hh = h + 1; hh < nb; hh++) S.valid[hh] = 0;
            break;
          }
        }
        S.rng_state = st;
      }
    }
    __syncthreads();
    RS_STAMP(1);
    // ---- phase 2: 7-point models, 16 lanes per hypothesis
    for (int h = tid >> 4; h < nb; h += nt >> 4) {
      int n = 0;
      if (S.valid[h]) n = run7point_group(S.ms1[h], S.ms2[h], S.F[h], S.work[h], tid & 15);
      if ((tid & 15) == 0) {
        S.nmodels[h] = n < 0 ? 0 : n;
        S.good[h][0] = S.good[h][1] = S.good[h][2] = 0;
      }
    }
    __syncthreads();
    RS_STAMP(2);
    // ---- phase 3: inlier counts for every (hypothesis, model) [RANSAC] / median error of every model [LMedS]
    if (lmeds) {
      for (int hk = tid; hk < nb * 3; hk += nt) {
        const int h = hk / 3, k = hk - 3 * h;
        if (k >= S.nmodels[h]) continue;
        int bits[14];  // count <= 14; std::sort on the float bit patterns as ints (ptsetreg.cpp)
        for (int i = 0; i < count; i++) {
          const int b = __float_as_int(epipolar_error(S.F[h] + 9 * k, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]));
          int j = i;
          for (; j > 0 && bits[j - 1] > b; j--) bits[j] = bits[j - 1];
          bits[j] = b;
        }
        S.med[h][k] = (count & 1) ? (double)__int_as_float(bits[count / 2])
                                  : (double)(__int_as_float(bits[count / 2 - 1]) + __int_as_float(bits[count / 2])) * 0.5;
      }
    } else
    for (int item = tid; item < nb * 3 * count; item += nt) {
      int hk = item / count, i = item - hk * count;
      int h = hk / 3, k = hk - 3 * h;
      if (k < S.nmodels[h] && epipolar_inlier(S.F[h] + 9 * k, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1], t))
        atomicAdd(&S.good[h][k], 1);
    }
    __syncthreads();
    RS_STAMP(3);
    // ---- phase 4: sequential scan with the adaptive bound
    if (tid == 0) {
      for (int h = 0; h < nb; h++) {
        if (S.iter >= S.niters) { S.done = 1; break; }
        if (!S.valid[h]) {
          if (S.iter == 0) S.failed_first = 1;
          S.done = 1;
          break;
        }
        for (int k = 0; k < S.nmodels[h]; k++) {
          if (lmeds) {
            if (S.med[h][k] < S.min_median) {
              S.min_median = S.med[h][k];
              for (int q = 0; q < 9; q++) S.bestF[q] = S.F[h][9 * k + q];
            }
            continue;
          }
          int good = S.good[h][k];
          if (good > max(S.max_good, model_points - 1)) {
            S.max_good = good;
            for (int q = 0; q < 9; q++) S.bestF[q] = S.F[h][9 * k + q];
            S.best_h = h;
            S.niters = ransac_update_iters(confidence, (double)(count - good) / count, model_points, S.niters);
          }
        }
        S.iter++;
      }
      if (S.iter >= S.niters) S.done = 1;
    }
    __syncthreads();
    RS_STAMP(4);
    if (prof && tid == 0) prof[6] += 1;
    if (S.done) break;
  }
  if (lmeds && S.min_median < 1.7976931348623157e308 && !S.failed_first) {
    double sigma = 2.5 * 1.4826 * (1 + 5. / (count - model_points)) * sqrt(S.min_median);
    sigma = fmax(sigma, 0.001);
    const float ts = (float)(sigma * sigma);
    for (int i = tid; i < count; i += nt)
      mask[i] = epipolar_inlier(S.bestF, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1], ts) ? 1 : 0;
  } else if (!lmeds && S.max_good > 0 && !S.failed_first) {
    for (int i = tid; i < count; i += nt)
      mask[i] = epipolar_inlier(S.bestF, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1], t) ? 1 : 0;
  } else {
    for (int i = tid; i < count; i += nt) mask[i] = 1;
  }
  __syncthreads();
  RS_STAMP(5);
#undef RS_STAMP
}

// ---- per-sequence tracker update -------------------------------------------------------------------------------
struct TrackerArrays {
  int cap;
  int rows, cols;
  float *cur_pts, *pre_pts, *forw_pts;  // [seq][cap][2]
  int *ids, *track_cnt;                  // [seq][cap]
  int *n_pts;                            // [seq] number of tracked points (cur)
  int *n_forw;                           // [seq] after this kernel: points kept
  int *n_id;                             // [seq] next id
  uint8_t *lk_status;                    // [seq][cap]
  int *kept_xy;                          // [seq][cap][2] rounded centres of kept features (for the mask painter)
  int *n_kept;                           // [seq]
  float *pnp_pts;                        // [seq][cap][2] forw_pts / ids as solveVinsPnP sees them (feature_tracker.cpp:207:
  int *pnp_ids, *n_pnp;                  // behind the first F-RANSAC, ahead of rejectWithF / setMask)
  const int *hw;                         // [2 r + 1] half-widths of the filled circle
  int radius;
  float f_thresh;
  double f_conf;
  long long *prof;  // null, or cycle stamps of sequence 0's workgroup (VIO_AMD_TU_PROF, tools/tu_prof.sh)
};

// Stable compaction of the five per-feature arrays by `keep` flags; all arrays staged in LDS.
// CAP: capacity the LDS arrays are laid out for. The launcher picks the smallest instantiation that holds the tracker's
// feature slots: with the arrays of the 512-slot layout (95 KB with the RANSAC state) only ONE workgroup fits a CU and the
// 512 sequences of the bench ran as two rounds; the 256-slot layout is 47 KB.
template <int CAP>
struct TrackShared {
  float pre[CAP][2], cur[CAP][2], forw[CAP][2];
  int ids[CAP], cnt[CAP];
  int keep[CAP];  // (dword flags: sub-dword LDS accesses are slow)
  int pos[CAP];
  float t_pre[CAP][2], t_cur[CAP][2], t_forw[CAP][2];
  int t_ids[CAP], t_cnt[CAP];
  int n;
  int order[CAP];
  int ixy[CAP][2];
  unsigned long long inside[CAP][CAP / 64];
  int hw[2 * kMaxRadius + 1];  // half-widths of the filled circle (setMask), staged from global memory
};

template <int CAP>
__device__ void compact_block(TrackShared<CAP> &T) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n = T.n;
  __syncthreads();  // everyone has read n
  // exclusive prefix sum of keep (n <= kMaxCap): a ballot per wave and chunk of blockDim items, the waves' counts through LDS
  // (one work-item walking the flags was a chain of ~n LDS round trips, three times per frame)
  {
    __shared__ int s_wcnt[16];
    const int wave = tid >> 6, lane = tid & 63, nwv = nt >> 6;
    int running = 0;
    for (int base = 0; base < n; base += nt) {
      const int i = base + tid;
      const bool k = i < n && T.keep[i];
      const unsigned long long b = __builtin_amdgcn_ballot_w64(k);
      if (lane == 0) s_wcnt[wave] = __builtin_popcountll(b);
      __syncthreads();
      int off = running, total = 0;
      for (int w = 0; w < nwv; w++) {
        const int c = s_wcnt[w];
        if (w < wave) off += c;
        total += c;
      }
      if (i < n) T.pos[i] = off + __builtin_popcountll(b & ((1ull << lane) - 1ull));
      running += total;
      __syncthreads();
    }
    if (tid == 0) T.n = running;
  }
  __syncthreads();
  for (int i = tid; i < n; i += nt)
    if (T.keep[i]) {
      int p = T.pos[i];
      T.t_pre[p][0] = T.pre[i][0], T.t_pre[p][1] = T.pre[i][1];
      T.t_cur[p][0] = T.cur[i][0], T.t_cur[p][1] = T.cur[i][1];
      T.t_forw[p][0] = T.forw[i][0], T.t_forw[p][1] = T.forw[i][1];
      T.t_ids[p] = T.ids[i], T.t_cnt[p] = T.cnt[i];
    }
  __syncthreads();
  for (int i = tid; i < T.n; i += nt) {
    T.pre[i][0] = T.t_pre[i][0], T.pre[i][1] = T.t_pre[i][1];
    T.cur[i][0] = T.t_cur[i][0], T.cur[i][1] = T.t_cur[i][1];
    T.forw[i][0] = T.t_forw[i][0], T.forw[i][1] = T.t_forw[i][1];
    T.ids[i] = T.t_ids[i], T.cnt[i] = T.t_cnt[i];
  }
  __syncthreads();
}

// One workgroup per sequence: everything between the LK call and goodFeaturesToTrack.
// (two waves per SIMD: left to itself the compiler takes 289 registers per work-item for the double-precision 7-point code --
// ONE wave per SIMD, one workgroup per CU, and the 512 sequences of the bench ran as two rounds of 256 workgroups)
template <int CAP>
__global__ __launch_bounds__(256, 2) void track_update_kernel(TrackerArrays A, int publish) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  TrackShared<CAP> &T = *reinterpret_cast<TrackShared<CAP> *>(smem_raw);
  RansacShared &R = *reinterpret_cast<RansacShared *>(smem_raw + ((sizeof(TrackShared<CAP>) + 15) & ~(size_t)15));
  const int seq = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const size_t base = (size_t)seq * A.cap;
  const int n0 = A.n_pts[seq];
#define TU_STAMP(k)                                                  \
  do {                                                               \
    if (A.prof && seq == 0 && tid == 0) A.prof[k] = clock64();       \
  } while (0)
  TU_STAMP(0);
  if (tid == 0) T.n = n0;
  for (int i = tid; i < n0; i += nt) {
    T.pre[i][0] = A.pre_pts[(base + i) * 2], T.pre[i][1] = A.pre_pts[(base + i) * 2 + 1];
    T.cur[i][0] = A.cur_pts[(base + i) * 2], T.cur[i][1] = A.cur_pts[(base + i) * 2 + 1];
    T.forw[i][0] = A.forw_pts[(base + i) * 2], T.forw[i][1] = A.forw_pts[(base + i) * 2 + 1];
    T.ids[i] = A.ids[base + i], T.cnt[i] = A.track_cnt[base + i];
    // status && inBorder (feature_tracker.cpp:183-185, :18-24)
    int ix = __float2int_rn(T.forw[i][0]), iy = __float2int_rn(T.forw[i][1]);
    bool inb = 1 <= ix && ix < A.cols - 1 && 1 <= iy && iy < A.rows - 1;
    T.keep[i] = (A.lk_status[base + i] && inb) ? 1 : 0;
  }
  __syncthreads();
  TU_STAMP(1);
  if (n0 > 0) {
    compact_block(T);
    TU_STAMP(2);
    if (T.n >= 8) {  // findFundamentalMat(cur_pts, forw_pts, FM_RANSAC, F_THRESHOLD, 0.99) :194-205
      fundamental_ransac_block(R, &T.cur[0][0], &T.forw[0][0], T.n, A.f_thresh, A.f_conf, T.keep, A.prof && seq == 0 ? A.prof + 16 : nullptr);
      TU_STAMP(3);
      compact_block(T);
    }
  }
  TU_STAMP(4);
  // the point list solveVinsPnP joins with the solved landmarks (:207): behind the first rejection, ahead of the
  // publish-frame steps (rejectWithF :235, setMask :255) that drop more of it
  for (int i = tid; i < T.n; i += nt) {
    A.pnp_pts[(base + i) * 2] = T.forw[i][0], A.pnp_pts[(base + i) * 2 + 1] = T.forw[i][1];
    A.pnp_ids[base + i] = T.ids[i];
  }
  if (tid == 0) A.n_pnp[seq] = T.n;
  if (publish) {
    if (T.n >= 8) {  // rejectWithF: (pre_pts, forw_pts) :89-103
      TU_STAMP(5);
      fundamental_ransac_block(R, &T.pre[0][0], &T.forw[0][0], T.n, A.f_thresh, A.f_conf, T.keep, A.prof && seq == 0 ? A.prof + 24 : nullptr);
      TU_STAMP(6);
      compact_block(T);
    }
    TU_STAMP(7);
    const int n = T.n;
    for (int i = tid; i < n; i += nt) {
      T.cnt[i] += 1;  // for (auto &n : track_cnt) n++ :252-253
      T.ixy[i][0] = __float2int_rn(T.forw[i][0]), T.ixy[i][1] = __float2int_rn(T.forw[i][1]);
    }
    __syncthreads();
    // setMask :50-87 — stable order by track_cnt desc
    for (int i = tid; i < n; i += nt) {
      int rank = 0, ci = T.cnt[i];
      for (int j = 0; j < n; j++) rank += (T.cnt[j] > ci || (T.cnt[j] == ci && j < i)) ? 1 : 0;
      T.order[rank] = i;
    }
    const int words = (n + 63) / 64;
    __syncthreads();
    TU_STAMP(8);
    // inside[i][w] bit j: pixel of i lies in the filled circle painted at j (i, j in ORIGINAL indices). One word per wave
    // instruction: the wave takes (i, w), lane b tests j = 64 w + b, the ballot IS the word. (The per-thread loop over
    // 64 j with the half-width table read from global memory inside it was 57 k cycles; the table now sits in LDS.)
    for (int q = tid; q < 2 * A.radius + 1; q += nt) T.hw[q] = A.hw[q];
    __syncthreads();
    {
      // lane b keeps the centres j = 64 w + b of the (at most kW) words in registers and walks i: per (i, w) two compares, one
      // table read and a ballot -- the item loop that re-read both centres per (i, w) was a chain of three dependent LDS round
      // trips per item (566 cycles each, 18 % of the kernel)
      const int wave = tid >> 6, lane = tid & 63, nwv = nt >> 6;
      constexpr int kW = CAP / 64;
      int jx[kW], jy[kW];
#pragma unroll
      for (int w = 0; w < kW; w++) {
        const int j = min(64 * w + lane, n - 1);
        jx[w] = T.ixy[j][0], jy[w] = T.ixy[j][1];
      }
      const int rad = A.radius;
      for (int i = wave; i < n; i += nwv) {
        const int ix = T.ixy[i][0], iy = T.ixy[i][1];
#pragma unroll
        for (int w = 0; w < kW; w++) {
          if (w < words) {
            const int dy = iy - jy[w], dx = ix - jx[w];
            const int h = T.hw[rad + min(max(dy, -rad), rad)];
            const bool in = 64 * w + lane < n && dy >= -rad && dy <= rad && dx >= -h && dx <= h;
            const unsigned long long bits = __builtin_amdgcn_ballot_w64(in);
            if (lane == 0) T.inside[i][w] = bits;
          }
        }
      }
    }
    __syncthreads();
    TU_STAMP(9);
    // greedy in sorted order (:73-83): a feature is kept unless it lies in the circle of an earlier kept one. One wave:
    // lane w owns word w of the kept set in a register, the hit test is a ballot; the index list and the bit rows do not
    // depend on the decisions, so the compiler is free to fetch them ahead. Kept features get their output position
    // (forw_pts.push_back order) on the way. (One thread with the kept set in a dynamically indexed array: 105 k cycles.)
    if (tid < 64) {
      const int lane = tid;
      unsigned long long mine = 0;  // word `lane` of the kept set
      int c = 0;
      for (int r0 = 0; r0 < n; r0 += 64) {
        const int my_i = r0 + lane < n ? T.order[r0 + lane] : 0;
        const int cnt = n - r0 < 64 ? n - r0 : 64;
        unsigned long long kept = 0;  // bit q: candidate r0 + q is kept (uniform)
        for (int q0 = 0; q0 < cnt; q0 += 8) {  // the bit rows of eight candidates are fetched together, then decided in order
          int is[8];
          unsigned long long rows[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            is[u] = __builtin_amdgcn_readlane(my_i, (q0 + u) & 63);
            rows[u] = lane < words ? T.inside[is[u]][lane] : 0ull;
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {  // (no LDS traffic and no lane-0 branches inside the chain: the decisions go to a bit mask)
            const int i = is[u];
            const bool keep_it = q0 + u < cnt && __builtin_amdgcn_ballot_w64((rows[u] & mine) != 0) == 0;
            const unsigned long long bit = keep_it ? 1ULL << (i & 63) : 0ull;
            mine |= lane == (i >> 6) ? bit : 0ull;
            kept |= (unsigned long long)keep_it << ((q0 + u) & 63);
          }
        }
        if (lane < cnt) {  // candidate r0 + lane: its decision and, if kept, its output position (forw_pts.push_back order)
          const bool k = (kept >> lane) & 1ull;
          T.keep[my_i] = k ? 1 : 0;
          if (k) T.pos[my_i] = c + __builtin_popcountll(kept & ((1ull << lane) - 1ull));
        }
        c += __builtin_popcountll(kept);
      }
      if (lane == 0) T.n = c;
    }
    __syncthreads();
    TU_STAMP(10);
    for (int i = tid; i < n; i += nt)
      if (T.keep[i]) {
        int p = T.pos[i];
        T.t_forw[p][0] = T.forw[i][0], T.t_forw[p][1] = T.forw[i][1];
        T.t_ids[p] = T.ids[i], T.t_cnt[p] = T.cnt[i];
        A.kept_xy[(base + p) * 2] = T.ixy[i][0], A.kept_xy[(base + p) * 2 + 1] = T.ixy[i][1];
      }
    __syncthreads();
    for (int i = tid; i < T.n; i += nt) {
      T.forw[i][0] = T.t_forw[i][0], T.forw[i][1] = T.t_forw[i][1];
      T.ids[i] = T.t_ids[i], T.cnt[i] = T.t_cnt[i];
      // pre_pts / cur_pts are overwritten with forw_pts at the end of a publish frame (:274, :285)
    }
    if (tid == 0) A.n_kept[seq] = T.n;
    __syncthreads();
  }
  // write back
  const int n = T.n;
  for (int i = tid; i < n; i += nt) {
    A.forw_pts[(base + i) * 2] = T.forw[i][0], A.forw_pts[(base + i) * 2 + 1] = T.forw[i][1];
    A.ids[base + i] = T.ids[i], A.track_cnt[base + i] = T.cnt[i];
    if (!publish) {
      A.pre_pts[(base + i) * 2] = T.pre[i][0], A.pre_pts[(base + i) * 2 + 1] = T.pre[i][1];
      // cur_pts = forw_pts (:285)
      A.cur_pts[(base + i) * 2] = T.forw[i][0], A.cur_pts[(base + i) * 2 + 1] = T.forw[i][1];
    }
  }
  if (tid == 0) {
    A.n_forw[seq] = n;
    if (!publish) A.n_pts[seq] = n;
  }
  TU_STAMP(11);
#undef TU_STAMP
}

// ---- goodFeaturesToTrack -----------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ordered_bits(float v) {  // monotone float -> uint map (atomicMax on it)
  unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// goodFeaturesToTrack front half, fused: cornerMinEigenVal (Sobel/1/3060 -> products -> 3x3 box -> min eigenvalue),
// the masked maximum (minMaxLoc) and the candidate test "equal to its 3x3 maximum, inside the mask, 1-px border
// excluded". Nothing but the candidates goes to memory and nothing is staged in LDS:
//   * a wave owns 64 consecutive columns (60 outputs + 2 halo each side) and walks down a strip of kDetR rows;
//   * each lane forms the derivative products of ITS column for one row from nine image bytes (row addresses are
//     scalar, a product position outside the image takes the value of its reflected position: boxFilter's
//     BORDER_REFLECT_101 acts on the product images);
//   * horizontal neighbours travel on the DPP network (wave_shr / wave_shl), vertical neighbours are the previous two
//     rows kept in registers: box sums, eigenvalues and the 3x3 maximum all slide down the strip;
//   * the mask is either an image (stand-alone operator) or the discs setMask painted (feature_tracker.cpp:80),
//     rasterised once per workgroup into one 64-bit word per (wave, row).
// The quality threshold needs the global maximum, so it is applied later by the selection kernel.
// (Round 6 built the two-pixels-per-lane form with v_pk_add / v_pk_mul / v_pk_fma_f32 for everything that takes no DPP operand --
// bit-exact, 124 outputs per wave -- and measured it SLOWER, 500 vs 406 us per 512 frames: 228 M against 209 M vector instructions
// per launch. What packs is a third of a row's instructions; the DPP neighbour sums, the square-root selects, the 3x3 maximum and
// the candidate tests are per pixel either way, and building the operand pairs costs moves. commit 3f94e20 has the kernel.)
constexpr int kDetW = 60;      // output columns per wave
constexpr int kDetWaves = 4;   // waves per workgroup, side by side
constexpr int kDetR = 32;      // output rows per strip (64 halves the halo rows but its 31 KB candidate buffer halves the resident
                               // workgroups: 548 instead of 413 us per 512 frames)

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));  // (bound_ctrl: an edge lane reads 0 without a zero-initialised destination)
}
#define LANE_LEFT(v) dpp_f32<0x138>(v)   /* wave_shr:1 -> value of lane - 1 */
#define LANE_RIGHT(v) dpp_f32<0x130>(v)  /* wave_shl:1 -> value of lane + 1 */

// sqrtf, correctly rounded, for arguments that are zero or normal numbers (what cornerMinEigenVal feeds it here: the
// derivative products are multiples of (1 / 3060)^2 rounded to float, so (a - c)^2 + b^2 is either 0 or above 1e-30 -- never a
// denormal, which is the one case the library expansion spends five more instructions on): the hardware estimate, then the
// neighbour whose residual changes sign (the same correction step the library uses).
__device__ __forceinline__ float sqrt_rn_normal(float x) {
  const float r = __builtin_amdgcn_sqrtf(x);
  const float rm = __int_as_float(__float_as_int(r) - 1), rp = __int_as_float(__float_as_int(r) + 1);
  const float em = __builtin_fmaf(-rm, r, x), ep = __builtin_fmaf(-rp, r, x);
  float y = em <= 0.f ? rm : r;
  y = ep > 0.f ? rp : y;
  // x == 0 needs no case of its own: r = 0, rm is a NaN pattern (0 - 1 ulp), so em is NaN and `em <= 0` is false; rp is the smallest
  // denormal, ep = fma(-rp, 0, 0) = 0 and `ep > 0` is false: y = r = 0. (Infinity cannot occur: the products are bounded by 255^2.)
  return y;
}

template <bool IMG_MASK>
__global__ __launch_bounds__(256) void detect_kernel(const uint8_t *img_base, size_t img_stride, const uint8_t *mask_base,
                                                     size_t mask_stride, const int *kept_xy, const int *n_kept, int cap,
                                                     const int *hw, int radius, unsigned *max_bits, int rows, int cols,
                                                     unsigned long long *cand_base, int seg_cap, int *n_cand, const int *n_have,
                                                     int max_corners) {
  // n_max_cnt = MAX_CNT - forw_pts.size() <= 0: the reference does not call goodFeaturesToTrack at all (feature_tracker.cpp:256-266);
  // the sequence's candidate counters and maxima stay zero, which is "no corners" to the selection kernel
  if (n_have && n_have[blockIdx.z] >= max_corners) return;
  constexpr int kCandLds = kDetWaves * kDetW * kDetR / 4 + 64;  // a 3x3 maximum occupies at most one pixel in four
  __shared__ unsigned long long s_mask[kDetWaves][kDetR];
  __shared__ unsigned long long s_cand[kCandLds];
  __shared__ unsigned smax;
  __shared__ int s_ncand, s_base, s_ndisc;
  __shared__ int2 s_disc[kMaxCap];
  const int seq = blockIdx.z, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint8_t *img = img_base + (size_t)seq * img_stride;
  const int XB = blockIdx.x * (kDetWaves * kDetW), Y0 = blockIdx.y * kDetR;
  const int X0 = XB + wave * kDetW;  // first output column of this wave; lane l <-> extended column X0 - 2 + l
  if (tid == 0) smax = 0, s_ncand = 0, s_ndisc = 0;
  if (tid < kDetWaves * kDetR) s_mask[tid / kDetR][tid % kDetR] = 0ull;
  __syncthreads();
  if (!IMG_MASK) {
    // disc rows that cross this workgroup's strip -> bits of the (wave, row) words
    // (1) every thread tests one kept feature against the strip, the few that touch it are listed in LDS;
    // (2) one item per (listed disc, strip row, wave) ORs the lanes the disc covers on that row.
    const int nk = n_kept[seq];
    for (int k = tid; k < nk; k += 256) {
      const int cx = kept_xy[((size_t)seq * cap + k) * 2], cy = kept_xy[((size_t)seq * cap + k) * 2 + 1];
      if (cy + radius >= Y0 && cy - radius < Y0 + kDetR && cx + radius >= XB - 2 && cx - radius < XB + kDetWaves * kDetW + 2) {
        const int slot = atomicAdd(&s_ndisc, 1);
        s_disc[slot] = make_int2(cx, cy);  // (at most cap entries: every feature at most once)
      }
    }
    __syncthreads();
    const int nd = s_ndisc;
    for (int it = tid; it < nd * kDetR * kDetWaves; it += 256) {
      const int wv = it % kDetWaves, r = (it / kDetWaves) % kDetR, dsc = it / (kDetWaves * kDetR);
      const int cx = s_disc[dsc].x, dy = Y0 + r - s_disc[dsc].y;
      if (dy < -radius || dy > radius) continue;
      const int h = hw[radius + dy];
      const int l0 = max(cx - h - (XB + wv * kDetW - 2), 0), l1 = min(cx + h - (XB + wv * kDetW - 2), 63);
      if (l0 > l1) continue;
      const unsigned long long bits = (l1 - l0 == 63 ? ~0ull : ((1ull << (l1 - l0 + 1)) - 1ull)) << l0;
      atomicOr(&s_mask[wv][r], bits);
    }
    __syncthreads();
  }
  const float s = (float)(1.0 / (4.0 * 3.0 * 255.0));
  const float s2 = s * 2.f;
  const __amdgpu_buffer_rsrc_t img_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, rows * cols, 0x00020000);  // (raw bytes, bounds = the frame)
  // this lane's column (reflected like the product image) and its two neighbours for the Sobel taps
  const int xe = X0 - 2 + lane;
  const int xr = reflect101(min(max(xe, -cols + 1), 2 * cols - 2), cols);
  const int xm = reflect101(xr - 1, cols), xp = reflect101(xr + 1, cols);
  // Everything that slides down the strip has period three -- the (d, t) pairs of the three image rows in flight, the horizontal sums of
  // three product rows, three rows of eigenvalues --: each lives in a ring of three registers and the row loop is unrolled by three, so
  // that the slot of "the row that enters at this step" is a compile-time constant and NOTHING is moved when the window slides (with
  // named variables shifted at the end of a step the compiler kept ~11 register moves per row, a seventh of the loop's vector instructions:
  // the reload path of the border rows turned the shifted values into phi nodes it could not coalesce).
  float D[3] = {0.f, 0.f, 0.f}, T[3] = {0.f, 0.f, 0.f};                                  // slot s % 3: the image row that enters at step s
  float Hxx[3] = {0.f, 0.f, 0.f}, Hxy[3] = {0.f, 0.f, 0.f}, Hyy[3] = {0.f, 0.f, 0.f};    // slot s % 3: horizontal sums of step s's product row
  float E[3] = {0.f, 0.f, 0.f};                                                           // slot s % 3: eigenvalues formed at step s
  float my_max = -INFINITY;  // running masked maximum of this lane's column (one v_max per row; mapped to ordered bits once)
  int prev_yr = -1, prev_yp = -1;               // rows of the previous step (uniform)
  const bool out_lane = lane >= 2 && lane < 2 + kDetW && xe < cols;
  static_assert((kDetR + 4) % 3 == 0, "the row loop is unrolled by the rotation period");
  // The three bytes of the row that ENTERS at step s + 1 are fetched during step s (ring RB, slot s % 3): a step never waits for its own
  // loads. (Fetched where they were used, every row began with a memory round trip that seven resident waves per SIMD did not cover:
  // dropping 15 % of the row's vector instructions in a timing experiment bought 3.5 %.)
  unsigned RB[3][3] = {{0u, 0u, 0u}, {0u, 0u, 0u}, {0u, 0u, 0u}};
  // The strip as a function of INNER (a compile-time flag): a strip whose rows y - 1 .. y + 1 all lie inside the image (13 of the 15
  // strips of a 480-row frame) needs no reflected row indices, never reloads rows and has no row bounds to test on its outputs. Those
  // were ~20 of the ~45 SCALAR instructions of a step -- and the CU's one scalar unit, shared by the 28 resident waves, was what the
  // kernel was bound by once the loads were out of the way (the row arithmetic without reflection, as a timing experiment: 324 -> 264 us).
  const bool inner_rows = Y0 - 3 >= 0 && Y0 + kDetR + 2 <= rows - 1;  // (uniform)
  auto strip = [&](auto inner_tag) {
  constexpr bool INNER = decltype(inner_tag)::value;
  auto entering_row = [&](int step) {  // image row that enters at `step`: y + 1 of extended product row Y0 - 2 + step
    if (INNER) return Y0 - 1 + step;
    const int yr_ = reflect101(min(max(Y0 - 2 + step, -rows + 1), 2 * rows - 2), rows);
    return reflect101(yr_ + 1, rows);
  };
  auto fetch_raw = [&](int y, unsigned (&b)[3]) {
    const int ro = __builtin_amdgcn_readfirstlane(y * cols);
    b[0] = __builtin_amdgcn_raw_buffer_load_b8(img_rsrc, xm, ro, 0), b[1] = __builtin_amdgcn_raw_buffer_load_b8(img_rsrc, xr, ro, 0),
    b[2] = __builtin_amdgcn_raw_buffer_load_b8(img_rsrc, xp, ro, 0);
  };
  fetch_raw(entering_row(0), RB[2]);
  // (three steps; HEAD: the first six of a strip, which still test whether an eigenvalue / an output row exists -- the loop behind them
  // does not: two scalar compares and branches per step less)
  auto three_steps = [&](int s0, auto head_tag) {
    constexpr bool HEAD = decltype(head_tag)::value;
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int step = s0 + u;
      const int jn = u, jm = (u + 1) % 3, jr = (u + 2) % 3;  // slots of the entering row (y + 1), of row y - 1 and of row y
      const int ye = Y0 - 2 + step;  // extended product row (uniform)
      const int yr = INNER ? ye : reflect101(min(max(ye, -rows + 1), 2 * rows - 2), rows);
      const int ym = INNER ? ye - 1 : reflect101(yr - 1, rows), yp = INNER ? ye + 1 : reflect101(yr + 1, rows);
      // Per image row the Sobel pair needs two numbers per column: the horizontal difference d = a(x+1) - a(x-1) and the
      // horizontal smoothing t = 2s a(x) + s (a(x-1) + a(x+1)); dx = 2s d(y) + s (d(y-1) + d(y+1)), dy = t(y+1) - t(y-1). Inside
      // the image the rows (y-1, y) of this step are the rows (y, y+1) of the previous one: only the entering row is loaded.
      auto row_dt = [&](int y, float &d, float &t) {
        // buffer loads: the row offset rides in the scalar offset operand, the column in the lane offset -- no address arithmetic
        // on the vector unit (three 64-bit adds per row with flat pointers)
        const int ro = __builtin_amdgcn_readfirstlane(y * cols);
        const float a0 = (float)__builtin_amdgcn_raw_buffer_load_b8(img_rsrc, xm, ro, 0), a1 = (float)__builtin_amdgcn_raw_buffer_load_b8(img_rsrc, xr, ro, 0),
                    a2 = (float)__builtin_amdgcn_raw_buffer_load_b8(img_rsrc, xp, ro, 0);
        d = a2 - a0;
        t = s2 * a1 + s * (a0 + a2);
      };
      if (INNER ? step == 0 : !(step > 0 && ym == prev_yr && yr == prev_yp)) row_dt(ym, D[jm], T[jm]), row_dt(yr, D[jr], T[jr]);  // (first step; image border: reflected rows)
      if (step + 1 < kDetR + 4) fetch_raw(entering_row(step + 1), RB[jn]);  // (uniform) next step's row, on its way under this step's arithmetic
      {
        const float a0 = (float)RB[jr][0], a1 = (float)RB[jr][1], a2 = (float)RB[jr][2];  // this step's row yp: fetched during the previous step
        D[jn] = a2 - a0;
        T[jn] = s2 * a1 + s * (a0 + a2);
      }
      prev_yr = yr, prev_yp = yp;
      const float dx = s2 * D[jr] + s * (D[jm] + D[jn]);
      const float dy = T[jn] - T[jm];
      const float pxx = dx * dx, pxy = dx * dy, pyy = dy * dy;
      // 3-tap horizontal sums (left + centre) + right, then the 3-row vertical sums (top + middle) + bottom
      Hxx[jn] = (LANE_LEFT(pxx) + pxx) + LANE_RIGHT(pxx);
      Hxy[jn] = (LANE_LEFT(pxy) + pxy) + LANE_RIGHT(pxy);
      Hyy[jn] = (LANE_LEFT(pyy) + pyy) + LANE_RIGHT(pyy);
      float e2 = 0.f;
      if (!HEAD || step >= 2) {  // eigenvalue of extended row ye - 1
        const float a = ((Hxx[jm] + Hxx[jr]) + Hxx[jn]) * 0.5f;
        const float b = (Hxy[jm] + Hxy[jr]) + Hxy[jn];
        const float c = ((Hyy[jm] + Hyy[jr]) + Hyy[jn]) * 0.5f;
        e2 = (a + c) - sqrt_rn_normal((a - c) * (a - c) + b * b);
      }
      E[jn] = e2;
      if (!HEAD || step >= 4) {  // output row ye - 2: centre E[jr], neighbours E[jm] / E[jn] and the lanes left and right
        const int y = ye - 2, r = step - 4;
        float m = fmaxf(fmaxf(E[jm], E[jr]), E[jn]);
        m = fmaxf(m, fmaxf(LANE_LEFT(m), LANE_RIGHT(m)));
        if (out_lane && (INNER || y < rows)) {
          const float v = E[jr];
          bool unmasked;
          if (IMG_MASK) unmasked = mask_base[(size_t)seq * mask_stride + (size_t)y * cols + xe] != 0;
          else unmasked = ((s_mask[wave][r] >> lane) & 1ull) == 0ull;
          if (unmasked) {
            my_max = fmaxf(my_max, v);
            if (xe >= 1 && xe < cols - 1 && (INNER || (y >= 1 && y < rows - 1)) && v > 0.f && v == m) {
              const int slot = atomicAdd(&s_ncand, 1);
              const unsigned idx = (unsigned)y << 16 | (unsigned)xe;  // (row-major order like y * cols + x, and no division to take it apart)
              if (slot < kCandLds) s_cand[slot] = ((unsigned long long)__float_as_uint(v) << 32) | (0xffffffffu - idx);
            }
          }
        }
      }
    }
  };  // three_steps
  static_assert(kDetR + 4 > 6 && (kDetR + 4) % 6 == 0, "two head rounds, then rounds of six steps");
  three_steps(0, std::true_type{}), three_steps(3, std::true_type{});
#pragma unroll 1
  for (int s0 = 6; s0 < kDetR + 4; s0 += 6) three_steps(s0, std::false_type{}), three_steps(s0 + 3, std::false_type{});
  };  // strip
  if (inner_rows) strip(std::true_type{});
  else strip(std::false_type{});
  // masked maximum: wave max on the DPP/shuffle network, then one LDS atomic per wave
  {
    unsigned m = my_max == -INFINITY ? 0u : ordered_bits(my_max);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if (lane == 0 && m) atomicMax(&smax, m);
  }
  __syncthreads();
  // every strip of a sequence owns a segment of the candidate list and a counter / maximum slot, so the global
  // atomics of one address come from the few workgroups of that strip only
  const int nseg = gridDim.y;
  const size_t segi = (size_t)seq * nseg + blockIdx.y;
  const int nc = min(s_ncand, kCandLds);
  if (tid == 0 && nc) s_base = atomicAdd(&n_cand[segi], nc);
  __syncthreads();
  for (int i = tid; i < nc; i += 256) {
    const int slot = s_base + i;
    if (slot < seg_cap) cand_base[segi * seg_cap + slot] = s_cand[i];
  }
  if (tid == 0 && smax) atomicMax(&max_bits[segi], smax);
}

#undef LANE_LEFT
#undef LANE_RIGHT

struct SelectParams {
  int cap, rows, cols, max_corners;
  float min_dist;
  double fx, fy, cx, cy;
};

constexpr int kSelThreads = 512;
constexpr int kSelMaxSeg = 256;  // strips of kDetR rows: images up to 8192 rows
constexpr int kSelLds = 8192;  // candidate keys kept in LDS (64 KB); larger lists are processed in place in HBM

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

// goodFeaturesToTrack back half: quality threshold, then greedy pick in sorted order with the min-distance rule
// == repeat { take the best alive candidate; kill everything closer than min_dist }. Then addPoints / updateID /
// image_msg and the end-of-publish copies (feature_tracker.cpp:271-307).
__global__ __launch_bounds__(kSelThreads) void corner_select_kernel(unsigned long long *cand_base, int seg_cap, int nseg,
                                                                    int *n_cand, unsigned *max_bits,
                                                                    double quality, SelectParams P, float *forw_pts,
                                                                    float *cur_pts, float *pre_pts, int *ids, int *track_cnt,
                                                                    int *n_forw, int *n_pts, int *n_id, VioObs *obs,
                                                                    int *n_obs) {
  __shared__ unsigned long long keys[kSelLds];
  __shared__ unsigned long long red[kSelThreads / 64];
  __shared__ int s_n, s_cnt;
  __shared__ int s_off[kSelMaxSeg + 1];
  const int seq = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  unsigned long long *cand = cand_base + (size_t)seq * nseg * seg_cap;  // nseg segments of seg_cap keys
  const size_t base = (size_t)seq * P.cap;
  int n = n_forw[seq];
  const int want = P.max_corners - n;
  const float md2 = P.min_dist * P.min_dist;
  unsigned mb = 0;
  for (int g = 0; g < nseg; g++) mb = max(mb, max_bits[(size_t)seq * nseg + g]);
  const float thr = mb ? (float)((double)from_ordered_bits(mb) * quality) : 3.4e38f;
  if (tid == 0) s_n = n, s_cnt = 0;
  __syncthreads();
  // threshold(eig, maxVal * qualityLevel, THRESH_TOZERO): keep v > thr. Only the slots the detector really filled are
  // visited (prefix of the per-segment counts), survivors go to LDS and NOTHING is written back to the candidate list:
  // the first version zeroed every rejected and unused slot in HBM (0.6 MB per sequence and publish frame, twice the
  // frame itself) although only the rare in-place path below reads the list again.
  if (tid == 0) {
    int o = 0;
    for (int g = 0; g < nseg; g++) s_off[g] = o, o += min(n_cand[(size_t)seq * nseg + g], seg_cap);
    s_off[nseg] = o;
  }
  __syncthreads();
  const int real = s_off[nseg];
  for (int i = tid; i < real; i += nt) {
    int g = 0;
    while (s_off[g + 1] <= i) g++;
    const unsigned long long k = cand[(size_t)g * seg_cap + (i - s_off[g])];
    if (k && __uint_as_float((unsigned)(k >> 32)) > thr) {
      int slot = atomicAdd(&s_cnt, 1);
      if (slot < kSelLds) keys[slot] = k;
    }
  }
  __syncthreads();
  // this kernel is the only consumer of the per-segment counters: leave them zeroed for the next detection pass
  for (int g = tid; g < nseg; g += nt) n_cand[(size_t)seq * nseg + g] = 0, max_bits[(size_t)seq * nseg + g] = 0;
  if (s_cnt > kSelLds) {
    // More survivors than LDS holds (a cold start at 1280 x 720 / 1920 x 1080: ~10^5 candidates for 300 / 500 corners). The
    // first versions ran the round loop below over the WHOLE candidate range in HBM -- want x nseg x seg_cap visits, 43 ms /
    // 170 ms per frame at those sizes. The candidates are already bucketed: segment g holds the local maxima of strip g
    // (kDetR rows). So: one live maximum per strip in LDS, the global best is the maximum of those, and taking a corner only
    // touches the strips within min_dist of its row -- their candidates are suppressed and their maxima recomputed, the
    // others are not visited. Same picks in the same order (greedy in sorted order).
    __shared__ unsigned long long smax[kSelMaxSeg];
    __shared__ unsigned long long part[kSelThreads / 64][4];
    const int wv = tid >> 6, nwv = nt >> 6;
    // strips [g0, g0 + ng) (ng <= 4): suppress around (bx, by) when md2s >= 0 (else apply the quality threshold: first visit),
    // new maxima -> smax
    auto rescan = [&](int g0, int ng, int bx, int by, bool first) {
      unsigned long long m[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (q >= ng) continue;
        unsigned long long *seg = cand + (size_t)(g0 + q) * seg_cap;
        const int cnt = s_off[g0 + q + 1] - s_off[g0 + q];
        for (int j = tid; j < cnt; j += nt) {
          const unsigned long long c = seg[j];
          if (!c) continue;
          bool dead;
          if (first) {
            dead = !(__uint_as_float((unsigned)(c >> 32)) > thr);
          } else if (P.min_dist >= 1.f) {
            const unsigned idx = 0xffffffffu - (unsigned)(c & 0xffffffffu);
            const float dx = (float)((int)(idx & 0xffffu) - bx), dy = (float)((int)(idx >> 16) - by);
            dead = dx * dx + dy * dy < md2;
          } else {
            const unsigned idx = 0xffffffffu - (unsigned)(c & 0xffffffffu);
            dead = (int)(idx & 0xffffu) == bx && (int)(idx >> 16) == by;
          }
          if (dead) seg[j] = 0;
          else m[q] = c > m[q] ? c : m[q];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const unsigned long long x = wave_max_u64(m[q]);
        if ((tid & 63) == 0) part[wv][q] = x;
      }
      __syncthreads();
      if (tid < ng) {
        unsigned long long x = part[0][tid];
        for (int w2 = 1; w2 < nwv; w2++) x = part[w2][tid] > x ? part[w2][tid] : x;
        smax[g0 + tid] = x;
      }
      __syncthreads();
    };
    for (int g0 = 0; g0 < nseg; g0 += 4) rescan(g0, min(4, nseg - g0), 0, 0, true);
    const int reach = P.min_dist >= 1.f ? (int)ceilf(P.min_dist) : 0;
    for (int round = 0; round < want; round++) {
      unsigned long long best = 0;  // (every wave takes the maximum over the strips for itself: no barrier)
      for (int g = tid & 63; g < nseg; g += 64) best = smax[g] > best ? smax[g] : best;
      best = wave_max_u64(best);
      if (best == 0) break;
      const unsigned bidx = 0xffffffffu - (unsigned)(best & 0xffffffffu);
      const int bx = bidx & 0xffffu, by = bidx >> 16;
      if (tid == 0) {
        int p = s_n++;
        forw_pts[(base + p) * 2] = (float)bx, forw_pts[(base + p) * 2 + 1] = (float)by;
        ids[base + p] = -1, track_cnt[base + p] = 1;  // addPoints :36-48
      }
      const int glo = max(0, (by - reach) / kDetR), ghi = min(nseg - 1, (by + reach) / kDetR);
      for (int g0 = glo; g0 <= ghi; g0 += 4) rescan(g0, min(4, ghi - g0 + 1), bx, by, false);
    }
  } else {
  unsigned long long *K = keys;
  const int nc = s_cnt;
  // One pass over the candidates per round: it suppresses around the corner just taken AND collects the maximum of what
  // survives (the next corner) -- the first version walked the list twice per round with a barrier in between.
  unsigned long long mine = 0;  // this thread's largest live key
  for (int i = tid; i < nc; i += nt) {
    const unsigned long long k = K[i];
    mine = k > mine ? k : mine;
  }
  for (int round = 0; round < want; round++) {
    unsigned long long best = wave_max_u64(mine);
    if ((tid & 63) == 0) red[tid >> 6] = best;
    __syncthreads();
    best = red[0];
    for (int w = 1; w < nt / 64; w++) best = red[w] > best ? red[w] : best;
    __syncthreads();
    if (best == 0) break;
    const unsigned bidx = 0xffffffffu - (unsigned)(best & 0xffffffffu);
    const int bx = bidx & 0xffffu, by = bidx >> 16;
    if (tid == 0) {
      int p = s_n++;
      forw_pts[(base + p) * 2] = (float)bx, forw_pts[(base + p) * 2 + 1] = (float)by;
      ids[base + p] = -1, track_cnt[base + p] = 1;  // addPoints :36-48
    }
    mine = 0;
    if (P.min_dist >= 1.f) {
      for (int i = tid; i < nc; i += nt) {
        const unsigned long long c = K[i];
        if (c) {
          const unsigned idx = 0xffffffffu - (unsigned)(c & 0xffffffffu);
          const float dx = (float)((int)(idx & 0xffffu) - bx), dy = (float)((int)(idx >> 16) - by);
          if (dx * dx + dy * dy < md2) K[i] = 0;
          else mine = c > mine ? c : mine;
        }
      }
    } else {
      for (int i = tid; i < nc; i += nt) {
        const unsigned long long c = K[i];
        if (c == best) K[i] = 0;
        else mine = c > mine ? c : mine;
      }
    }
  }
  }
  __syncthreads();
  n = s_n;
  // updateID (:311-321), image_msg (:297-306), pre_pts = cur_pts = forw_pts (:274, :285)
  if (tid == 0) {
    int nid = n_id[seq];
    for (int i = 0; i < n; i++)
      if (ids[base + i] == -1) ids[base + i] = nid++;
    n_id[seq] = nid;
    n_forw[seq] = n, n_pts[seq] = n, n_obs[seq] = n;
  }
  __syncthreads();
  for (int i = tid; i < n; i += nt) {
    float x = forw_pts[(base + i) * 2], y = forw_pts[(base + i) * 2 + 1];
    cur_pts[(base + i) * 2] = x, cur_pts[(base + i) * 2 + 1] = y;
    pre_pts[(base + i) * 2] = x, pre_pts[(base + i) * 2 + 1] = y;
    VioObs o;
    o.id = ids[base + i];
    o.x = ((double)x - P.cx) / P.fx, o.y = ((double)y - P.cy) / P.fy, o.z = 1.0;
    obs[base + i] = o;
  }
}

void circle_halfwidths(int radius, std::vector<int> &hw) {  // cv::circle filled midpoint raster (drawing.cpp Circle())
  hw.assign(2 * radius + 1, -1);
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
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

template <class T>
int dev_alloc(T **p, size_t count) {
  return hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)) == hipSuccess ? VIO_OK : VIO_ENOMEM;
}

}  // namespace

struct vio_frontend {
  int device = -1;  // HIP device the context lives on (current device at create)
  VioConfig cfg;
  int n_seq = 0, cap = 0;
  LevelDims ld;
  hipStream_t stream = nullptr;
  // device state
  uint8_t *pyr[2] = {nullptr, nullptr};  // [n_seq][pyr_bytes]; cur = pyr[cur_idx], forw = pyr[1 - cur_idx]
  int cur_idx = 0;
  bool have_img = false;
  // level 0 of pyr[k] when it is NOT the pyramid's own copy: the frame inside the resident ring (vio_frontend_upload_frames) the
  // pyramid was built from -- the ring belongs to the context and outlives the two steps that read the frame
  const uint8_t *lvl0[2] = {nullptr, nullptr};
  uint8_t *mask = nullptr;      // [rows*cols], only the stand-alone vio_good_features uses a mask image
  unsigned *max_bits = nullptr;
  unsigned long long *cand = nullptr;
  int nseg = 0, seg_cap = 0;  // candidate list: one segment per strip of kDetR rows
  int *n_cand = nullptr;
  float *cur_pts = nullptr, *pre_pts = nullptr, *forw_pts = nullptr, *lk_err = nullptr;
  unsigned long long *lk_stats = nullptr;  // [64][16] iteration counters of lk_track_kernel (vio_frontend_lk_iterations)
  bool lk_stats_on = false;
  int *ids = nullptr, *track_cnt = nullptr, *n_pts = nullptr, *n_forw = nullptr, *n_id = nullptr, *kept_xy = nullptr,
      *n_kept = nullptr, *hw = nullptr, *n_obs = nullptr, *pnp_ids = nullptr, *n_pnp = nullptr;
  float *pnp_pts = nullptr;
  uint8_t *lk_status = nullptr;
  VioObs *obs = nullptr;
  bool attr_set = false;
  bool detect_always = false;  // VIO_AMD_DETECT_ALWAYS=1 (measurement aid): detect_kernel also runs for sequences that need no new corner
  // resident frames (throughput runs)
  uint8_t *frames = nullptr;
  int n_frames = 0;
  // timing
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  // host-buffer path (vio_frontend_read_images): device staging for the frames, pinned memory for the observations
  uint8_t *d_stage = nullptr;
  VioObs *p_obs = nullptr;
  uint8_t *p_frames = nullptr;  // page-locked gathering buffer of read_images
  bool pending = false, pending_publish = false;  // a submitted frame waits for vio_frontend_collect
  bool pending_async = false;                     // ... and it was queued by vio_frontend_submit_images_async (Async::rc is its status)
  // vio_frontend_submit_images_async: the submit itself (gather + transfers + launches) runs on this context's own host
  // thread, the caller returns at once; collect waits for it first
  struct Async {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    bool has_job = false, busy = false, quit = false;
    const uint8_t *gray = nullptr;
    int rows = 0, cols = 0, stride = 0, publish = 0, rc = VIO_OK;
  } *async = nullptr;
  int *p_nobs = nullptr;
  // host staging
  std::vector<VioObs> h_obs;
  std::vector<int> h_nobs;
};

namespace {

// The same for source rows that are whole dwords (scols % 4 == 0, dword-aligned level offsets): the patch arrives as 4
// pixels per lane and per load instead of 1, the output leaves as 4 pixels per store. One workgroup = 64 x 16 output
// pixels from a (2*64+8) x (2*16+4) source patch [2 ox - 4, 2 ox + 132) x [2 oy - 2, 2 oy + 34). Rows are reflected as
// whole rows; the (at most two) reflected columns on either side of the image are patched into the staged dwords.
// Arithmetic and rounding are those of pyr_down_kernel (bit-identical output).
constexpr int kPd4H = 16, kPd4Dw = (2 * kPdW + 8) / 4;  // 34 dwords per staged row
// COPY: the source is the caller's frame (forw_img = _img, feature_tracker.cpp:169): the workgroup also writes the core of
// its staged patch to level 0 of the pyramid, so the frame is read ONCE for the copy and for level 1 (a separate copy kernel
// plus the level-1 kernel read it twice and the copy wrote what the level-1 kernel read again).
// A workgroup walks kPd4T tiles down its column of the image: the loads of the NEXT tile's patch are issued (into registers) before
// the barriers and the arithmetic of this one, so a workgroup always has a patch in flight -- with one tile per workgroup the load
// phase was a third of a workgroup's life and the kernel ran at a third of the memory system's rate (2.2 TB/s on level 1).
#ifndef VIO_PD4T
#define VIO_PD4T 3
#endif
constexpr int kPd4T = VIO_PD4T;
template <bool COPY>
__global__ __launch_bounds__(256) void pyr_down4_kernel(const uint8_t *src_base, size_t src_stride, uint8_t *dst_base, size_t seq_stride,
                                                        int srows, int scols, int drows, int dcols, uint8_t *copy_base) {
  constexpr int SH = 2 * kPd4H + 4;
  constexpr int NE = (SH * kPd4Dw + 255) / 256;  // staged dwords per work-item: 5 (the last one only for some)
  __shared__ uint32_t raw[SH][kPd4Dw + 1];
  __shared__ uint32_t hsum[SH / 2][kPdW + 1];
  const uint8_t *src = src_base + (size_t)blockIdx.z * src_stride;
  uint8_t *dst = dst_base + (size_t)blockIdx.z * seq_stride;
  const int ox = blockIdx.x * kPdW;
  const int tid = threadIdx.x;
  const int ndw = scols >> 2, dw0 = (2 * ox - 4) >> 2;  // dwords per source row; dword index of the patch's first column
  // this work-item's elements of a patch: the same (row, dword) in every tile
  int e_ly[NE], e_d[NE];
  bool e_ok[NE];
#pragma unroll
  for (int i = 0; i < NE; i++) {
    const int e = tid + 256 * i;
    e_ok[i] = e < SH * kPd4Dw;
    const int ec = e_ok[i] ? e : 0;
    e_ly[i] = ec / kPd4Dw;
    e_d[i] = min(max(dw0 + ec - e_ly[i] * kPd4Dw, 0), ndw - 1);
  }
  const int tile0 = blockIdx.y * kPd4T;
  auto fetch = [&](int oy, uint32_t (&v)[NE]) {
#pragma unroll
    for (int i = 0; i < NE; i++) {
      const int y = reflect101(min(2 * oy + e_ly[i] - 2, 2 * srows - 2), srows);
      v[i] = reinterpret_cast<const uint32_t *>(src + (size_t)y * scols)[e_d[i]];
    }
  };
  uint32_t cur[NE], nxt[NE];
  fetch(tile0 * kPd4H, cur);
#pragma unroll 1
  for (int t = 0; t < kPd4T; t++) {
    const int oy = (tile0 + t) * kPd4H;
    if (oy >= drows) break;  // (uniform)
    const bool more = t + 1 < kPd4T && oy + kPd4H < drows;
    if (more) fetch(oy + kPd4H, nxt);
#pragma unroll
    for (int i = 0; i < NE; i++) {
      if (!e_ok[i]) continue;
      const int ly = e_ly[i], lx = tid + 256 * i - ly * kPd4Dw;
      raw[ly][lx] = cur[i];
      if (COPY) {  // core of the patch: rows 2 oy .. 2 oy + 2 kPd4H - 1, columns 2 ox .. 2 ox + 2 kPdW - 1 (staged dwords 1 .. 32)
        const int yy = 2 * oy + ly - 2, dd = dw0 + lx;
        if (ly >= 2 && ly < SH - 2 && lx >= 1 && lx <= (2 * kPdW) / 4 && yy < srows && dd < ndw)
          reinterpret_cast<uint32_t *>(copy_base + (size_t)blockIdx.z * seq_stride + (size_t)yy * scols)[dd] = cur[i];
      }
    }
    __syncthreads();
    // columns -2, -1 (left-most tile) and scols, scols + 1 (the tile that holds them): BORDER_REFLECT_101
    if (tid < SH) {
      if (ox == 0) {  // dword 0 = columns -4 .. -1, dword 1 = columns 0 .. 3
        const uint32_t d1 = raw[tid][1];
        raw[tid][0] = ((d1 >> 16) & 0xffu) << 16 | ((d1 >> 8) & 0xffu) << 24;  // col -2 <- col 2, col -1 <- col 1
      }
      const int lr = ndw - dw0;  // staged dword that starts at column scols
      if (lr >= 1 && lr < kPd4Dw) {
        const uint32_t dl = raw[tid][lr - 1];  // columns scols - 4 .. scols - 1
        raw[tid][lr] = ((dl >> 16) & 0xffu) | ((dl >> 8) & 0xffu) << 8;  // col scols <- col scols - 2, col scols + 1 <- col scols - 3
      }
    }
    __syncthreads();
    // horizontal pass, two source rows per item: output column ox + lx reads columns 2 lx - 2 .. 2 lx + 2 of the tile = staged
    // bytes 2 lx + 2 .. 2 lx + 6: four of them cut out of a dword pair with one funnel shift and weighted by ONE byte dot product
    // (1 4 6 4), the fifth is its accumulator. The sums (<= 16 * 255) of rows 2 p and 2 p + 1 share a dword, so that the vertical
    // pass below is two 16-bit dot products per output.
    for (int e = tid; e < (SH / 2) * kPdW; e += 256) {
      const int p = e / kPdW, lx = e - p * kPdW;
      const int b = 2 * lx + 2, dwi = b >> 2, sh = (b & 3) * 8;
      auto hrow = [&](int ly) {
        const uint32_t lo = raw[ly][dwi], hi = raw[ly][dwi + 1];
        return __builtin_amdgcn_udot4(__builtin_amdgcn_alignbit(hi, lo, sh), 0x04060401u, __builtin_amdgcn_ubfe(hi, sh, 8), false);
      };
      hsum[p][lx] = hrow(2 * p) | hrow(2 * p + 1) << 16;
    }
    __syncthreads();
    {
      const int ly = tid >> 4, lx = (tid & 15) * 4;
      const int x = ox + lx, y = oy + ly;
      if (x < dcols && y < drows) {
        uint32_t out = 0;
        const lk_us2 w14 = {1, 4}, w64 = {6, 4};
#pragma unroll
        for (int i = 0; i < 4; i++) {  // source rows 2 ly .. 2 ly + 4 = pairs ly, ly + 1 and the low half of pair ly + 2
          lk_us2 p0, p1;
          __builtin_memcpy(&p0, &hsum[ly][lx + i], 4), __builtin_memcpy(&p1, &hsum[ly + 1][lx + i], 4);
          const uint32_t v = __builtin_amdgcn_udot2(p0, w14, __builtin_amdgcn_udot2(p1, w64, hsum[ly + 2][lx + i] & 0xffffu, false), false);
          out |= ((v + 128) >> 8) << (8 * i);
        }
        *reinterpret_cast<uint32_t *>(dst + (size_t)y * dcols + x) = out;  // (dcols % 4 == 0: x + 3 < dcols)
      }
    }
    if (!more) break;
    // (the next tile's patch overwrites `raw` behind this tile's horizontal pass, which the barrier above closed; `hsum` is rewritten
    // behind the next tile's two barriers)
#pragma unroll
    for (int i = 0; i < NE; i++) cur[i] = nxt[i];
  }
}

// forw_img = _img for every sequence: packed frames -> level 0 of each sequence's pyramid (16 B per thread when the
// layout allows it; the runtime's rectangular copy reaches ~140 GB/s on this shape).
__global__ __launch_bounds__(256) void copy_frames_kernel(const uint8_t *src, size_t src_stride, uint8_t *dst,
                                                          size_t dst_stride, size_t bytes, int vec_ok) {
  const int seq = blockIdx.y;
  const uint8_t *sp = src + (size_t)seq * src_stride;
  uint8_t *dp = dst + (size_t)seq * dst_stride;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec_ok) {
    if (i * 16 < bytes) reinterpret_cast<uint4 *>(dp)[i] = reinterpret_cast<const uint4 *>(sp)[i];
  } else {
    const size_t b0 = i * 16, b1 = b0 + 16 < bytes ? b0 + 16 : bytes;
    for (size_t b = b0; b < b1; b++) dp[b] = sp[b];
  }
}

// feature_tracker.cpp:183-205 (+ :235-255, :50-87 on publish frames) for every sequence: one workgroup each
int launch_track_update(vio_frontend *fe, int publish, hipStream_t st) {
  TrackerArrays A;
  A.cap = fe->cap, A.rows = fe->cfg.image_rows, A.cols = fe->cfg.image_cols;
  A.cur_pts = fe->cur_pts, A.pre_pts = fe->pre_pts, A.forw_pts = fe->forw_pts, A.ids = fe->ids, A.track_cnt = fe->track_cnt;
  A.n_pts = fe->n_pts, A.n_forw = fe->n_forw, A.n_id = fe->n_id, A.lk_status = fe->lk_status, A.kept_xy = fe->kept_xy;
  A.n_kept = fe->n_kept, A.hw = fe->hw, A.radius = fe->cfg.min_dist, A.f_thresh = (float)fe->cfg.f_threshold;
  A.pnp_pts = fe->pnp_pts, A.pnp_ids = fe->pnp_ids, A.n_pnp = fe->n_pnp;
  A.f_conf = fe->cfg.f_confidence;
  static const bool tu_prof = getenv("VIO_AMD_TU_PROF") && getenv("VIO_AMD_TU_PROF")[0] == '1';
  static long long *d_prof = nullptr;
  if (tu_prof && !d_prof) (void)hipMalloc(&d_prof, 32 * sizeof(long long));
  A.prof = tu_prof ? d_prof : nullptr;
  // the smallest LDS layout that holds this tracker's feature slots (fe->cap <= kMaxCap is checked at create)
  const bool small = fe->cap <= 256;
  const size_t shm = ((small ? sizeof(TrackShared<256>) : sizeof(TrackShared<kMaxCap>)) + 15 & ~(size_t)15) + sizeof(RansacShared);
  const void *kern = small ? (const void *)track_update_kernel<256> : (const void *)track_update_kernel<kMaxCap>;
  if (!fe->attr_set) {
    HIP_OK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    fe->attr_set = true;
  }
  if (A.prof) (void)hipMemsetAsync(d_prof, 0, 32 * sizeof(long long), st);
  if (small) hipLaunchKernelGGL(track_update_kernel<256>, dim3(fe->n_seq), dim3(256), shm, st, A, publish);
  else hipLaunchKernelGGL(track_update_kernel<kMaxCap>, dim3(fe->n_seq), dim3(256), shm, st, A, publish);
  if (A.prof) {  // (debug only: synchronises)
    long long h[32];
    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
      static const char *name[11] = {"load", "compact", "ransac(cur,forw)", "compact", "pnp list", "ransac(pre,forw)", "compact", "counts+rank",
                                     "inside bits", "greedy", "outputs"};
      fprintf(stderr, "track_update cycles (sequence 0, publish %d):", publish);
      const int map[11][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {4, 5}, {5, 6}, {6, 7}, {7, 8}, {8, 9}, {9, 10}, {10, 11}};
      for (int k = 0; k < 11; k++) fprintf(stderr, " %s %lld,", name[k], h[map[k][1]] - h[map[k][0]]);
      fprintf(stderr, " total %lld\n", h[11] - h[0]);
      for (int c = 0; c < 2; c++)
        fprintf(stderr, "   ransac call %d: draw %lld, gather+collinear %lld, 7-point %lld, scoring %lld, scan %lld, mask %lld, rounds %lld\n", c,
                h[16 + 8 * c], h[17 + 8 * c], h[18 + 8 * c], h[19 + 8 * c], h[20 + 8 * c], h[21 + 8 * c], h[22 + 8 * c]);
    }
  }
  return VIO_OK;
}

// keep_frames: d_frames stays valid and unchanged until the frame after the next one has been tracked (the context's resident ring):
// level 0 is then read where it lies instead of being copied into the pyramid.
int fe_step(vio_frontend *fe, const uint8_t *d_frames /* [n_seq][rows*cols] on device */, int publish, hipStream_t st, bool keep_frames = false) {
  const int S = fe->n_seq, rows = fe->cfg.image_rows, cols = fe->cfg.image_cols, cap = fe->cap;
  const size_t img_bytes = (size_t)rows * cols;
  // forw_img = _img : level 0 of the forw pyramid
  const int fidx = fe->have_img ? 1 - fe->cur_idx : fe->cur_idx;
  uint8_t *forw = fe->pyr[fidx];
  // level 0 and level 1 from ONE read of the frame when the rows of both are whole dwords (640x480, 720p, 1080p)
  const bool fused0 = fe->ld.levels >= 2 && cols % 4 == 0 && fe->ld.cols[1] % 4 == 0 && cols >= 8 && fe->ld.pyr_bytes % 4 == 0 &&
                      img_bytes % 4 == 0 && (uintptr_t)d_frames % 4 == 0 && (uintptr_t)forw % 4 == 0 && fe->ld.off[1] % 4 == 0 &&
                      !(getenv("VIO_AMD_PYR_UNFUSED") && getenv("VIO_AMD_PYR_UNFUSED")[0] == '1');
  const bool alias0 = keep_frames && fused0 && !(getenv("VIO_AMD_COPY_LEVEL0") && getenv("VIO_AMD_COPY_LEVEL0")[0] == '1');
  fe->lvl0[fidx] = alias0 ? d_frames : nullptr;
  if (!fused0) {
    const int vec_ok = img_bytes % 16 == 0 && fe->ld.pyr_bytes % 16 == 0 && (uintptr_t)d_frames % 16 == 0 &&
                       (uintptr_t)forw % 16 == 0;
    dim3 grd((unsigned)((img_bytes + 16 * 256 - 1) / (16 * 256)), S);
    hipLaunchKernelGGL(copy_frames_kernel, grd, dim3(256), 0, st, d_frames, img_bytes, forw, fe->ld.pyr_bytes, img_bytes,
                       vec_ok);
  }
  for (int l = 1; l < fe->ld.levels; l++) {
    const uint8_t *sp = forw + fe->ld.off[l - 1];
    uint8_t *dp = forw + fe->ld.off[l];
    const int sc = fe->ld.cols[l - 1], dc = fe->ld.cols[l];
    if (l == 1 && fused0) {
      dim3 blk(256), grd((dc + kPdW - 1) / kPdW, (fe->ld.rows[1] + kPd4H * kPd4T - 1) / (kPd4H * kPd4T), S);
      if (alias0)
        hipLaunchKernelGGL(pyr_down4_kernel<false>, grd, blk, 0, st, d_frames, img_bytes, dp, fe->ld.pyr_bytes, rows, cols, fe->ld.rows[1], dc,
                           (uint8_t *)nullptr);
      else
        hipLaunchKernelGGL(pyr_down4_kernel<true>, grd, blk, 0, st, d_frames, img_bytes, dp, fe->ld.pyr_bytes, rows, cols, fe->ld.rows[1], dc, forw);
      continue;
    }
    // whole-dword rows on both sides (and at least two staged dwords of image): 4 pixels per load and per store
    const bool dwords = sc % 4 == 0 && dc % 4 == 0 && sc >= 8 && fe->ld.pyr_bytes % 4 == 0 && (uintptr_t)sp % 4 == 0 && (uintptr_t)dp % 4 == 0;
    if (dwords) {
      dim3 blk(256), grd((dc + kPdW - 1) / kPdW, (fe->ld.rows[l] + kPd4H * kPd4T - 1) / (kPd4H * kPd4T), S);
      hipLaunchKernelGGL(pyr_down4_kernel<false>, grd, blk, 0, st, sp, fe->ld.pyr_bytes, dp, fe->ld.pyr_bytes, fe->ld.rows[l - 1], sc, fe->ld.rows[l], dc,
                         (uint8_t *)nullptr);
    } else {
      dim3 blk(256), grd((dc + kPdW - 1) / kPdW, (fe->ld.rows[l] + kPdH - 1) / kPdH, S);
      hipLaunchKernelGGL(pyr_down_kernel, grd, blk, 0, st, sp, dp, fe->ld.pyr_bytes, fe->ld.rows[l - 1], sc, fe->ld.rows[l], dc);
    }
  }
  if (!fe->have_img) {  // pre_img = cur_img = forw_img = _img (:166-167); no points to track yet
    fe->have_img = true;
    fe->cur_idx = fidx;
  } else {
    LkParams P;
    P.ld = fe->ld, P.cap = cap;
    P.max_count = std::min(std::max(fe->cfg.lk_max_iters, 0), 100);
    double eps = std::min(std::max(fe->cfg.lk_eps, 0.), 10.);
    P.epsilon_sq = eps * eps, P.epsilon_sq_f = lk_eps_screen(eps * eps), P.min_eig = (float)fe->cfg.lk_min_eig;
    P.prev0 = fe->lvl0[fe->cur_idx], P.next0 = fe->lvl0[fidx], P.stride0 = img_bytes;
    dim3 grd((cap + 4 * kLkFpw - 1) / (4 * kLkFpw), S);
    // (the counting variant is a kernel of its own: STATS = [2 * levels] iterations run / (feature, level) visits of this launch,
    // vio_frontend_lk_iterations; the product kernel sits at 80 registers for six waves per SIMD and has none to spare)
    if (fe->lk_stats_on && fe->lk_stats)
      hipLaunchKernelGGL((lk_track_kernel<kLkFpw, true>), grd, dim3(256), 0, st, fe->pyr[fe->cur_idx], forw, P, fe->n_pts, fe->cur_pts,
                         fe->forw_pts, fe->lk_status, fe->lk_err, fe->lk_stats);
    else
      hipLaunchKernelGGL((lk_track_kernel<kLkFpw, false>), grd, dim3(256), 0, st, fe->pyr[fe->cur_idx], forw, P, fe->n_pts, fe->cur_pts,
                         fe->forw_pts, fe->lk_status, fe->lk_err, (unsigned long long *)nullptr);
  }
  int rcu = launch_track_update(fe, publish, st);
  if (rcu != VIO_OK) return rcu;
  if (publish) {
    dim3 tb(256), tg((cols + kDetWaves * kDetW - 1) / (kDetWaves * kDetW), fe->nseg, S);
    hipLaunchKernelGGL(detect_kernel<false>, tg, tb, 0, st, alias0 ? d_frames : forw, alias0 ? img_bytes : fe->ld.pyr_bytes, (const uint8_t *)nullptr, (size_t)0,
                       fe->kept_xy, fe->n_kept, cap, fe->hw, fe->cfg.min_dist, fe->max_bits, rows, cols, fe->cand,
                       fe->seg_cap, fe->n_cand, fe->detect_always ? (const int *)nullptr : fe->n_forw, fe->cfg.max_corners);
    SelectParams SP;
    SP.cap = cap, SP.rows = rows, SP.cols = cols, SP.max_corners = fe->cfg.max_corners, SP.min_dist = (float)fe->cfg.min_dist;
    SP.fx = fe->cfg.fx, SP.fy = fe->cfg.fy, SP.cx = fe->cfg.cx, SP.cy = fe->cfg.cy;
    hipLaunchKernelGGL(corner_select_kernel, dim3(S), dim3(kSelThreads), 0, st, fe->cand, fe->seg_cap, fe->nseg, fe->n_cand,
                       fe->max_bits, fe->cfg.quality_level, SP, fe->forw_pts, fe->cur_pts, fe->pre_pts, fe->ids,
                       fe->track_cnt, fe->n_forw, fe->n_pts, fe->n_id, fe->obs, fe->n_obs);
  }
  HIP_OK(hipGetLastError());
  fe->cur_idx = fidx;  // cur_img = forw_img (:284)
  return VIO_OK;
}

}  // namespace

extern "C" {

int vio_frontend_create(const VioConfig *cfg, int32_t n_seq, vio_frontend_t **out) {
  if (!cfg || !out || n_seq < 1) return VIO_EINVAL;
  if (cfg->lk_win != kWin || cfg->lk_levels < 0 || cfg->lk_levels >= kMaxLevels || cfg->max_corners < 1 ||
      cfg->max_corners > kMaxCap || cfg->image_rows < 32 || cfg->image_cols < 32 || cfg->min_dist < 0 || cfg->min_dist > kMaxRadius ||
      cfg->image_rows > 32767 || cfg->image_cols > 65535)  // (corner candidates carry their position as y << 16 | x)
    return VIO_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    fprintf(stderr, "vio_amd: no HIP device visible; the front-end has no CPU fallback\n");
    return VIO_ENODEV;
  }
  if (!vio::single_hip_runtime()) return VIO_ENODEV;
  vio_frontend *fe = new vio_frontend();
  fe->device = vio::current_device();
  fe->cfg = *cfg, fe->n_seq = n_seq, fe->cap = cfg->max_corners;
  fe->detect_always = getenv("VIO_AMD_DETECT_ALWAYS") && getenv("VIO_AMD_DETECT_ALWAYS")[0] == '1';
  // buildOpticalFlowPyramid: levels stop when one would not hold the window
  LevelDims &ld = fe->ld;
  ld.rows[0] = cfg->image_rows, ld.cols[0] = cfg->image_cols, ld.off[0] = 0, ld.levels = 1;
  size_t off = (size_t)ld.rows[0] * ld.cols[0];
  for (int l = 1; l <= cfg->lk_levels; l++) {
    int r = (ld.rows[l - 1] + 1) / 2, c = (ld.cols[l - 1] + 1) / 2;
    if (c <= kWin || r <= kWin) break;
    ld.rows[l] = r, ld.cols[l] = c, ld.off[l] = off, off += (size_t)r * c, ld.levels = l + 1;
  }
  ld.pyr_bytes = (off + 255) & ~(size_t)255;
  const size_t S = n_seq, cap = fe->cap;
  fe->nseg = (cfg->image_rows + kDetR - 1) / kDetR;
  if (fe->nseg > kSelMaxSeg) {
    delete fe;
    return VIO_ECAP;
  }
  fe->seg_cap = kDetR * cfg->image_cols / 4;  // a 3x3 local maximum can occupy at most one pixel in four
  int rc = VIO_OK;
  if (hipStreamCreateWithFlags(&fe->stream, hipStreamNonBlocking) != hipSuccess) rc = VIO_ENODEV;
#define ALLOC(ptr, count) \
  if (rc == VIO_OK) rc = dev_alloc(&(ptr), (count))
  ALLOC(fe->pyr[0], S * ld.pyr_bytes);
  ALLOC(fe->pyr[1], S * ld.pyr_bytes);
  ALLOC(fe->max_bits, S * fe->nseg);
  ALLOC(fe->cand, S * fe->nseg * fe->seg_cap);
  ALLOC(fe->n_cand, S * fe->nseg);
  ALLOC(fe->cur_pts, S * cap * 2);
  ALLOC(fe->pre_pts, S * cap * 2);
  ALLOC(fe->forw_pts, S * cap * 2);
  ALLOC(fe->lk_err, S * cap);
  ALLOC(fe->ids, S * cap);
  ALLOC(fe->track_cnt, S * cap);
  ALLOC(fe->n_pts, S);
  ALLOC(fe->n_forw, S);
  ALLOC(fe->n_id, S);
  ALLOC(fe->kept_xy, S * cap * 2);
  ALLOC(fe->n_kept, S);
  ALLOC(fe->pnp_pts, S * cap * 2);
  ALLOC(fe->pnp_ids, S * cap);
  ALLOC(fe->n_pnp, S);
  ALLOC(fe->n_obs, S);
  ALLOC(fe->lk_status, S * cap);
  ALLOC(fe->obs, S * cap);
  ALLOC(fe->hw, (size_t)2 * cfg->min_dist + 1);
#undef ALLOC
  if (rc == VIO_OK) {
    std::vector<int> hw;
    circle_halfwidths(cfg->min_dist, hw);
    bool ok = hipMemcpy(fe->hw, hw.data(), hw.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(fe->max_bits, 0, sizeof(unsigned) * S * fe->nseg) == hipSuccess;
    ok = ok && hipMemset(fe->n_cand, 0, sizeof(int) * S * fe->nseg) == hipSuccess;
    ok = ok && hipMemset(fe->n_pts, 0, sizeof(int) * S) == hipSuccess;
    ok = ok && hipMemset(fe->n_forw, 0, sizeof(int) * S) == hipSuccess;
    ok = ok && hipMemset(fe->n_id, 0, sizeof(int) * S) == hipSuccess;
    ok = ok && hipMemset(fe->n_kept, 0, sizeof(int) * S) == hipSuccess;
    ok = ok && hipMemset(fe->n_pnp, 0, sizeof(int) * S) == hipSuccess;
    ok = ok && hipMemset(fe->n_obs, 0, sizeof(int) * S) == hipSuccess;
    if (!ok) rc = VIO_ENODEV;
  }
  if (rc != VIO_OK) {
    vio_frontend_destroy(fe);
    return rc;
  }
  *out = fe;
  return VIO_OK;
}

int vio_frontend_get_device(const vio_frontend_t *fe, int32_t *device) {
  if (!fe || !device) return VIO_EINVAL;
  *device = fe->device;
  return VIO_OK;
}

void vio_frontend_destroy(vio_frontend_t *fe) {
  if (!fe) return;
  if (fe->async) {
    {
      std::lock_guard<std::mutex> lk(fe->async->m);
      fe->async->quit = true;
    }
    fe->async->cv.notify_all();
    if (fe->async->th.joinable()) fe->async->th.join();
    delete fe->async;
  }
  vio::DeviceScope scope(fe->device);
  (void)hipDeviceSynchronize();
  void *ptrs[] = {fe->pyr[0], fe->pyr[1], fe->mask, fe->max_bits, fe->cand, fe->n_cand, fe->cur_pts, fe->pre_pts,
                  fe->forw_pts, fe->lk_err, fe->ids, fe->track_cnt, fe->n_pts, fe->n_forw, fe->n_id, fe->kept_xy, fe->n_kept,
                  fe->n_obs, fe->lk_status, fe->obs, fe->hw, fe->frames, fe->pnp_pts, fe->pnp_ids, fe->n_pnp};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  for (auto &e : fe->events) (void)hipEventDestroy(e.first), (void)hipEventDestroy(e.second);
  if (fe->d_stage) (void)hipFree(fe->d_stage);
  if (fe->lk_stats) (void)hipFree(fe->lk_stats);
  if (fe->p_obs) (void)hipHostFree(fe->p_obs);
  if (fe->p_frames) (void)hipHostFree(fe->p_frames);
  if (fe->stream) (void)hipStreamDestroy(fe->stream);
  delete fe;
}

int vio_frontend_upload_frames(vio_frontend_t *fe, const uint8_t *gray, int32_t n_frames, int32_t rows, int32_t cols,
                               int32_t stride) {
  if (!fe || !gray || n_frames < 1) return VIO_EINVAL;
  if (rows != fe->cfg.image_rows || cols != fe->cfg.image_cols || stride < cols) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(fe);
  const size_t px = (size_t)rows * cols, total = (size_t)n_frames * fe->n_seq;
  if (fe->pending) return VIO_ESTATE;
  // the current image may still be read in place from the ring that is about to go: it moves into its pyramid first
  HIP_OK(hipDeviceSynchronize());
  for (int k = 0; k < 2; k++) {
    if (fe->lvl0[k])
      HIP_OK(hipMemcpy2D(fe->pyr[k], fe->ld.pyr_bytes, fe->lvl0[k], px, px, fe->n_seq, hipMemcpyDeviceToDevice));
    fe->lvl0[k] = nullptr;
  }
  if (fe->frames) (void)hipFree(fe->frames), fe->frames = nullptr;
  if (dev_alloc(&fe->frames, total * px) != VIO_OK) return VIO_ENOMEM;
  HIP_OK(hipMemcpy2D(fe->frames, cols, gray, stride, cols, total * rows, hipMemcpyHostToDevice));
  fe->n_frames = n_frames;
  return VIO_OK;
}

int vio_frontend_step_resident(vio_frontend_t *fe, int32_t frame_index, int32_t publish, void *stream) {
  if (!fe) return VIO_EINVAL;
  if (!fe->frames || frame_index < 0 || frame_index >= fe->n_frames) return VIO_ESTATE;
  if (fe->pending) return VIO_ESTATE;  // a submitted frame owns the tracker state and the observation staging until it is collected
  VIO_ON_DEVICE_OF(fe);
  hipStream_t st = stream ? (hipStream_t)stream : fe->stream;
  if (fe->events_used == fe->events.size()) {
    if (fe->events.size() >= 4096) fe->events_used = 0;
    else {
      hipEvent_t a, b;
      HIP_OK(hipEventCreate(&a));
      HIP_OK(hipEventCreate(&b));
      fe->events.push_back({a, b});
    }
  }
  auto &ev = fe->events[fe->events_used++];
  HIP_OK(hipEventRecord(ev.first, st));
  const size_t px = (size_t)fe->cfg.image_rows * fe->cfg.image_cols;
  int rc = fe_step(fe, fe->frames + (size_t)frame_index * fe->n_seq * px, publish, st, true);
  if (rc != VIO_OK) return rc;
  HIP_OK(hipEventRecord(ev.second, st));
  return VIO_OK;
}

int vio_frontend_sync(vio_frontend_t *fe) {
  if (!fe) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(fe);
  HIP_OK(hipDeviceSynchronize());
  return VIO_OK;
}

int vio_frontend_kernel_ms(vio_frontend_t *fe, double *ms_avg, int32_t *launches) {
  if (!fe || !ms_avg || !launches) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(fe);
  HIP_OK(hipDeviceSynchronize());
  double sum = 0;
  for (size_t i = 0; i < fe->events_used; i++) {
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, fe->events[i].first, fe->events[i].second));
    sum += ms;
  }
  *launches = (int32_t)fe->events_used;
  *ms_avg = fe->events_used ? sum / fe->events_used : 0.0;
  fe->events_used = 0;
  return VIO_OK;
}

// Host buffers the caller registered (vio_host_register): page-locked in place, so their frames go to the device by DMA straight
// from where the camera / decoder wrote them -- no gathering pass over them on the host pool.
namespace {
struct HostRange {
  const uint8_t *p;
  size_t n;
};
std::mutex g_host_reg_m;
std::vector<HostRange> g_host_reg;
bool host_range_registered(const void *ptr, size_t bytes) {
  const uint8_t *b = static_cast<const uint8_t *>(ptr);
  std::lock_guard<std::mutex> lk(g_host_reg_m);
  for (const HostRange &r : g_host_reg)
    if (b >= r.p && b + bytes <= r.p + r.n) return true;
  return false;
}
}  // namespace

int vio_host_register(void *ptr, size_t bytes) {
  if (!ptr || bytes == 0) return VIO_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || !vio::single_hip_runtime()) return VIO_ENODEV;
  {
    std::lock_guard<std::mutex> lk(g_host_reg_m);
    const uint8_t *b = static_cast<const uint8_t *>(ptr);
    for (const HostRange &r : g_host_reg)
      if (b < r.p + r.n && r.p < b + bytes) return VIO_ESTATE;  // overlaps a registered range: unregister that one first
  }
  if (hipHostRegister(ptr, bytes, hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    return VIO_ENOMEM;  // (the pages could not be locked: RLIMIT_MEMLOCK, or not the caller's memory)
  }
  std::lock_guard<std::mutex> lk(g_host_reg_m);
  g_host_reg.push_back({static_cast<const uint8_t *>(ptr), bytes});
  return VIO_OK;
}

int vio_host_unregister(void *ptr) {
  if (!ptr) return VIO_EINVAL;
  {
    std::lock_guard<std::mutex> lk(g_host_reg_m);
    size_t i = 0;
    for (; i < g_host_reg.size(); i++)
      if (g_host_reg[i].p == static_cast<const uint8_t *>(ptr)) break;
    if (i == g_host_reg.size()) return VIO_ESTATE;
    g_host_reg.erase(g_host_reg.begin() + (long)i);
  }
  // (a transfer still queued from this range must not lose its pages under it)
  if (hipDeviceSynchronize() != hipSuccess || hipHostUnregister(ptr) != hipSuccess) {
    (void)hipGetLastError();
    return VIO_ENODEV;
  }
  return VIO_OK;
}

// The two halves of read_images. submit: gathers the caller's (pageable) frames into page-locked memory, queues their
// transfer, every kernel of the frame and the copy of the published observations on the context's stream and returns
// without waiting for the device; collect: waits and hands the observations over. A caller that submits frame k+1 before
// it runs the estimator on frame k overlaps the front-end's transfers and kernels with the estimator's host phases.
static int submit_body(vio_frontend_t *fe, const uint8_t *gray, int32_t rows, int32_t cols, int32_t stride, int32_t publish) {
  VIO_ON_DEVICE_OF(fe);
  const size_t px = (size_t)rows * cols, S = fe->n_seq;
  // host frames land in a device staging buffer kept for the life of the context; observations come back through
  // pinned host memory (no allocation, no pageable bounce on the return path)
  if (!fe->d_stage && dev_alloc(&fe->d_stage, S * px) != VIO_OK) return VIO_ENOMEM;
  if (!fe->p_obs) {
    if (hipHostMalloc((void **)&fe->p_obs, sizeof(VioObs) * S * fe->cap + sizeof(int) * S, hipHostMallocDefault) != hipSuccess)
      return VIO_ENOMEM;
    fe->p_nobs = reinterpret_cast<int *>(fe->p_obs + S * fe->cap);
  }
  hipStream_t st = fe->stream;
  // The caller's frames are pageable memory: a direct copy bounces through the runtime's staging at ~5 GB/s. They are
  // gathered into page-locked memory by the host pool (one sequence per task, rows packed) in a few chunks, each chunk
  // going to the device as soon as it is complete, so the DMA of one overlaps the gathering of the next.
  if (stride == cols && host_range_registered(gray, S * px)) {
    // frames in a buffer the caller registered: one DMA from where they are
    HIP_OK(hipMemcpyAsync(fe->d_stage, gray, S * px, hipMemcpyHostToDevice, st));
  } else {
    if (!fe->p_frames && hipHostMalloc((void **)&fe->p_frames, S * px, hipHostMallocDefault) != hipSuccess) return VIO_ENOMEM;
    const size_t n_chunks = S >= 32 ? 8 : 1, per = (S + n_chunks - 1) / n_chunks;
    for (size_t c0 = 0; c0 < S; c0 += per) {
      const size_t c1 = std::min(S, c0 + per);
      vio::HostPool::get().parallel_for((int)(c1 - c0), [&](int i) {
        const size_t s = c0 + i;
        const uint8_t *src = gray + s * (size_t)rows * stride;
        uint8_t *dst = fe->p_frames + s * px;
        if (stride == cols) memcpy(dst, src, px);
        else
          for (int r = 0; r < rows; r++) memcpy(dst + (size_t)r * cols, src + (size_t)r * stride, cols);
      });
      HIP_OK(hipMemcpyAsync(fe->d_stage + c0 * px, fe->p_frames + c0 * px, (c1 - c0) * px, hipMemcpyHostToDevice, st));
    }
  }
  int rc = fe_step(fe, fe->d_stage, publish, st);
  if (rc != VIO_OK) return rc;
  if (publish) {
    HIP_OK(hipMemcpyAsync(fe->p_obs, fe->obs, sizeof(VioObs) * S * fe->cap, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(fe->p_nobs, fe->n_obs, sizeof(int) * S, hipMemcpyDeviceToHost, st));
  }
  return VIO_OK;
}

int vio_frontend_submit_images(vio_frontend_t *fe, const uint8_t *gray, int32_t rows, int32_t cols, int32_t stride, int32_t publish) {
  if (!fe || !gray) return VIO_EINVAL;
  if (rows != fe->cfg.image_rows || cols != fe->cfg.image_cols || stride < cols) return VIO_EINVAL;
  if (fe->pending) return VIO_ESTATE;  // one frame in flight per context: collect it first
  const int rc = submit_body(fe, gray, rows, cols, stride, publish);
  if (rc != VIO_OK) return rc;
  fe->pending = true, fe->pending_publish = publish != 0, fe->pending_async = false;
  return VIO_OK;
}

// The same with the submit's host work (gathering the frames into page-locked memory, queueing the transfers and the
// kernels) on the context's own host thread: the call returns at once, `gray` must stay valid and unchanged until
// vio_frontend_collect, which reports what the submit returned.
int vio_frontend_submit_images_async(vio_frontend_t *fe, const uint8_t *gray, int32_t rows, int32_t cols, int32_t stride, int32_t publish) {
  if (!fe || !gray) return VIO_EINVAL;
  if (rows != fe->cfg.image_rows || cols != fe->cfg.image_cols || stride < cols) return VIO_EINVAL;
  if (fe->pending) return VIO_ESTATE;
  try {
    if (!fe->async) {
      fe->async = new vio_frontend::Async();
      vio_frontend::Async *a = fe->async;
      a->th = std::thread([fe, a] {
        for (;;) {
          std::unique_lock<std::mutex> lk(a->m);
          a->cv.wait(lk, [&] { return a->quit || a->has_job; });
          if (a->quit) return;
          a->has_job = false;
          lk.unlock();
          const int rc = submit_body(fe, a->gray, a->rows, a->cols, a->stride, a->publish);
          lk.lock();
          a->rc = rc, a->busy = false;
          lk.unlock();
          a->cv.notify_all();
        }
      });
    }
  } catch (...) {
    return VIO_ENOMEM;
  }
  vio_frontend::Async *a = fe->async;
  {
    std::lock_guard<std::mutex> lk(a->m);
    a->gray = gray, a->rows = rows, a->cols = cols, a->stride = stride, a->publish = publish;
    a->has_job = true, a->busy = true, a->rc = VIO_OK;
  }
  a->cv.notify_all();
  fe->pending = true, fe->pending_publish = publish != 0, fe->pending_async = true;
  return VIO_OK;
}

int vio_frontend_collect(vio_frontend_t *fe, VioObs *out_obs, int32_t *n_obs) {
  if (!fe || !n_obs) return VIO_EINVAL;
  if (!fe->pending) return VIO_ESTATE;
  if (fe->pending_publish && !out_obs) return VIO_EINVAL;
  if (fe->async && fe->pending_async) {  // an asynchronous submit finishes queueing first (a->rc belongs to THAT submit only)
    vio_frontend::Async *a = fe->async;
    std::unique_lock<std::mutex> lk(a->m);
    a->cv.wait(lk, [&] { return !a->busy; });
    if (a->rc != VIO_OK) {
      fe->pending = false;
      return a->rc;
    }
  }
  VIO_ON_DEVICE_OF(fe);
  fe->pending = false;
  HIP_OK(hipStreamSynchronize(fe->stream));
  const size_t S = fe->n_seq;
  for (size_t s = 0; s < S; s++) n_obs[s] = 0;
  if (fe->pending_publish)
    for (size_t s = 0; s < S; s++) {
      n_obs[s] = fe->p_nobs[s];
      memcpy(out_obs + s * fe->cap, fe->p_obs + s * fe->cap, sizeof(VioObs) * fe->p_nobs[s]);
    }
  return VIO_OK;
}

int vio_frontend_read_images(vio_frontend_t *fe, const uint8_t *gray, int32_t rows, int32_t cols, int32_t stride,
                             const double *headers, int32_t publish, VioObs *out_obs, int32_t *n_obs) {
  (void)headers;  // the reference only forwards the header to the (default-off) vinsPnP branch
  if (!fe || !gray || !n_obs || (publish && !out_obs)) return VIO_EINVAL;
  int rc = vio_frontend_submit_images(fe, gray, rows, cols, stride, publish);
  if (rc != VIO_OK) return rc;
  return vio_frontend_collect(fe, out_obs, n_obs);
}

int vio_frontend_read_image(vio_frontend_t *fe, int32_t seq, const uint8_t *gray, int32_t rows, int32_t cols, int32_t stride,
                            double header, int32_t publish, VioObs *out_obs, int32_t *n_obs, VioTrackViz *viz) {
  (void)seq, (void)viz;
  if (!fe || fe->n_seq != 1 || seq != 0) return VIO_EINVAL;  // single-sequence contexts only; batches use read_images
  return vio_frontend_read_images(fe, gray, rows, cols, stride, &header, publish, out_obs, n_obs);
}

int vio_frontend_get_state(vio_frontend_t *fe, int32_t seq, float *cur_pts, int32_t *ids, int32_t *track_cnt, int32_t cap,
                           int32_t *n) {
  if (!fe || seq < 0 || seq >= fe->n_seq || !n) return VIO_EINVAL;
  if (fe->pending) return VIO_ESTATE;
  VIO_ON_DEVICE_OF(fe);
  HIP_OK(hipDeviceSynchronize());
  int m = 0;
  HIP_OK(hipMemcpy(&m, fe->n_pts + seq, sizeof(int), hipMemcpyDeviceToHost));
  *n = m;
  if (m > cap) return VIO_ECAP;
  const size_t base = (size_t)seq * fe->cap;
  if (m > 0) {
    HIP_OK(hipMemcpy(cur_pts, fe->cur_pts + base * 2, sizeof(float) * 2 * m, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ids, fe->ids + base, sizeof(int) * m, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(track_cnt, fe->track_cnt + base, sizeof(int) * m, hipMemcpyDeviceToHost));
  }
  return VIO_OK;
}

// forw_pts / ids at the point of readImage where solveVinsPnP runs (feature_tracker.cpp:207): the tracked points behind
// the first F-RANSAC of the last frame, ahead of rejectWithF / setMask. A frame that tracked nothing (the first one)
// leaves the list empty.
int vio_frontend_lk_iterations(vio_frontend_t *fe, int32_t enable, uint64_t *iterations, uint64_t *visits, int32_t levels_cap) {
  if (!fe || (iterations && (!visits || levels_cap < 1))) return VIO_EINVAL;
  if (fe->pending) return VIO_ESTATE;  // (a submitted frame's worker reads lk_stats_on / lk_stats while it queues the step)
  VIO_ON_DEVICE_OF(fe);
  constexpr int kSlots = 16 * 64;  // 64 copies of [iterations (levels) | visits (levels)], summed here
  if (!fe->lk_stats) {
    if (dev_alloc(&fe->lk_stats, (size_t)kSlots) != VIO_OK) return VIO_ENOMEM;
    HIP_OK(hipMemset(fe->lk_stats, 0, sizeof(unsigned long long) * kSlots));
  }
  HIP_OK(hipStreamSynchronize(fe->stream));
  if (iterations) {
    unsigned long long h[kSlots];
    HIP_OK(hipMemcpy(h, fe->lk_stats, sizeof(h), hipMemcpyDeviceToHost));
    const int L = fe->ld.levels;
    for (int l = 0; l < levels_cap; l++) {
      iterations[l] = visits[l] = 0;
      for (int c = 0; c < 64 && l < L; c++) iterations[l] += h[16 * c + l], visits[l] += h[16 * c + L + l];
    }
    HIP_OK(hipMemset(fe->lk_stats, 0, sizeof(h)));
  }
  if (enable >= 0) fe->lk_stats_on = enable != 0;
  return VIO_OK;
}

int vio_frontend_get_pnp_points(vio_frontend_t *fe, int32_t seq, float *forw_pts, int32_t *ids, int32_t cap, int32_t *n) {
  if (!fe || seq < 0 || seq >= fe->n_seq || !n) return VIO_EINVAL;
  if (fe->pending) return VIO_ESTATE;
  VIO_ON_DEVICE_OF(fe);
  HIP_OK(hipDeviceSynchronize());
  int m = 0;
  HIP_OK(hipMemcpy(&m, fe->n_pnp + seq, sizeof(int), hipMemcpyDeviceToHost));
  *n = m;
  if (m > cap) return VIO_ECAP;
  const size_t base = (size_t)seq * fe->cap;
  if (m > 0) {
    if (!forw_pts || !ids) return VIO_EINVAL;
    HIP_OK(hipMemcpy(forw_pts, fe->pnp_pts + base * 2, sizeof(float) * 2 * m, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ids, fe->pnp_ids + base, sizeof(int) * m, hipMemcpyDeviceToHost));
  }
  return VIO_OK;
}

// ---- the tracker's public fields in / out and the update step on its own (isolated tests of F3/F4/F6/F7) ---------------
int vio_frontend_set_tracks(vio_frontend_t *fe, int32_t seq, int32_t n, const float *pre_pts, const float *cur_pts,
                            const float *forw_pts, const int32_t *ids, const int32_t *track_cnt, const uint8_t *lk_status) {
  if (!fe || seq < 0 || seq >= fe->n_seq || n < 0 || (n > 0 && (!pre_pts || !cur_pts || !forw_pts || !ids || !track_cnt || !lk_status)))
    return VIO_EINVAL;
  if (n > fe->cap) return VIO_ECAP;
  if (fe->pending) return VIO_ESTATE;
  VIO_ON_DEVICE_OF(fe);
  HIP_OK(hipDeviceSynchronize());
  const size_t base = (size_t)seq * fe->cap;
  if (n > 0) {
    HIP_OK(hipMemcpy(fe->pre_pts + base * 2, pre_pts, sizeof(float) * 2 * n, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(fe->cur_pts + base * 2, cur_pts, sizeof(float) * 2 * n, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(fe->forw_pts + base * 2, forw_pts, sizeof(float) * 2 * n, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(fe->ids + base, ids, sizeof(int) * n, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(fe->track_cnt + base, track_cnt, sizeof(int) * n, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(fe->lk_status + base, lk_status, n, hipMemcpyHostToDevice));
  }
  HIP_OK(hipMemcpy(fe->n_pts + seq, &n, sizeof(int), hipMemcpyHostToDevice));
  return VIO_OK;
}

int vio_frontend_update_tracks(vio_frontend_t *fe, int32_t publish) {
  if (!fe) return VIO_EINVAL;
  if (fe->pending) return VIO_ESTATE;
  VIO_ON_DEVICE_OF(fe);
  int rc = launch_track_update(fe, publish, fe->stream);
  if (rc != VIO_OK) return rc;
  HIP_OK(hipStreamSynchronize(fe->stream));
  HIP_OK(hipGetLastError());
  return VIO_OK;
}

int vio_frontend_get_tracks(vio_frontend_t *fe, int32_t seq, float *forw_pts, int32_t *ids, int32_t *track_cnt, int32_t cap,
                            int32_t *n) {
  if (!fe || seq < 0 || seq >= fe->n_seq || !n) return VIO_EINVAL;
  if (fe->pending) return VIO_ESTATE;
  VIO_ON_DEVICE_OF(fe);
  HIP_OK(hipDeviceSynchronize());
  int m = 0;
  HIP_OK(hipMemcpy(&m, fe->n_forw + seq, sizeof(int), hipMemcpyDeviceToHost));
  *n = m;
  if (m > cap) return VIO_ECAP;
  const size_t base = (size_t)seq * fe->cap;
  if (m > 0) {
    HIP_OK(hipMemcpy(forw_pts, fe->forw_pts + base * 2, sizeof(float) * 2 * m, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ids, fe->ids + base, sizeof(int) * m, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(track_cnt, fe->track_cnt + base, sizeof(int) * m, hipMemcpyDeviceToHost));
  }
  return VIO_OK;
}

// ---- stand-alone operators (one reference call site each), for isolated parity tests --------------------------------
int vio_klt_track(const VioConfig *cfg, const uint8_t *prev, const uint8_t *next, int32_t rows, int32_t cols, int32_t stride,
                  const float *prev_pts, int32_t n, float *next_pts, uint8_t *status, float *err) {
  if (!cfg || !prev || !next || n < 0 || (n > 0 && (!prev_pts || !next_pts || !status || !err))) return VIO_EINVAL;
  if (n == 0) return VIO_OK;
  VioConfig c = *cfg;
  c.image_rows = rows, c.image_cols = cols, c.max_corners = std::min(std::max(n, 1), kMaxCap);
  if (n > kMaxCap) return VIO_ECAP;
  vio_frontend *fe = nullptr;
  int rc = vio_frontend_create(&c, 1, &fe);
  if (rc != VIO_OK) return rc;
  const size_t px = (size_t)rows * cols;
  uint8_t *d = nullptr;
  rc = dev_alloc(&d, 2 * px);
  auto fail = [&](int code) {
    if (d) (void)hipFree(d);
    vio_frontend_destroy(fe);
    return code;
  };
  if (rc != VIO_OK) return fail(rc);
  if (hipMemcpy2D(d, cols, prev, stride, cols, rows, hipMemcpyHostToDevice) != hipSuccess) return fail(VIO_ENODEV);
  if (hipMemcpy2D(d + px, cols, next, stride, cols, rows, hipMemcpyHostToDevice) != hipSuccess) return fail(VIO_ENODEV);
  hipStream_t st = fe->stream;
  for (int k = 0; k < 2; k++) {
    uint8_t *dst = fe->pyr[k];
    if (hipMemcpyAsync(dst, d + k * px, px, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(VIO_ENODEV);
    for (int l = 1; l < fe->ld.levels; l++) {
      dim3 blk(256), grd((fe->ld.cols[l] + kPdW - 1) / kPdW, (fe->ld.rows[l] + kPdH - 1) / kPdH, 1);
      hipLaunchKernelGGL(pyr_down_kernel, grd, blk, 0, st, dst + fe->ld.off[l - 1], dst + fe->ld.off[l], fe->ld.pyr_bytes,
                         fe->ld.rows[l - 1], fe->ld.cols[l - 1], fe->ld.rows[l], fe->ld.cols[l]);
    }
  }
  if (hipMemcpyAsync(fe->cur_pts, prev_pts, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st) != hipSuccess) return fail(VIO_ENODEV);
  if (hipMemcpyAsync(fe->n_pts, &n, sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) return fail(VIO_ENODEV);
  LkParams P;
  P.ld = fe->ld, P.cap = fe->cap, P.max_count = std::min(std::max(c.lk_max_iters, 0), 100);
  double eps = std::min(std::max(c.lk_eps, 0.), 10.);
  P.epsilon_sq = eps * eps, P.epsilon_sq_f = lk_eps_screen(eps * eps), P.min_eig = (float)c.lk_min_eig;
  P.prev0 = P.next0 = nullptr, P.stride0 = 0;
  hipLaunchKernelGGL((lk_track_kernel<kLkFpw, false>), dim3((fe->cap + 4 * kLkFpw - 1) / (4 * kLkFpw), 1), dim3(256), 0, st, fe->pyr[0], fe->pyr[1], P, fe->n_pts,
                     fe->cur_pts, fe->forw_pts, fe->lk_status, fe->lk_err, (unsigned long long *)nullptr);
  if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return fail(VIO_ENODEV);
  if (hipMemcpy(next_pts, fe->forw_pts, sizeof(float) * 2 * n, hipMemcpyDeviceToHost) != hipSuccess) return fail(VIO_ENODEV);
  if (hipMemcpy(status, fe->lk_status, n, hipMemcpyDeviceToHost) != hipSuccess) return fail(VIO_ENODEV);
  if (hipMemcpy(err, fe->lk_err, sizeof(float) * n, hipMemcpyDeviceToHost) != hipSuccess) return fail(VIO_ENODEV);
  return fail(VIO_OK);
}

int vio_good_features(const VioConfig *cfg, const uint8_t *img, const uint8_t *mask, int32_t rows, int32_t cols, int32_t stride,
                      int32_t max_corners, float *corners, int32_t *n_corners) {
  if (!cfg || !img || !corners || !n_corners || max_corners < 1) return VIO_EINVAL;
  if (max_corners > kMaxCap) return VIO_ECAP;
  VioConfig c = *cfg;
  c.image_rows = rows, c.image_cols = cols, c.max_corners = max_corners;
  vio_frontend *fe = nullptr;
  int rc = vio_frontend_create(&c, 1, &fe);
  if (rc != VIO_OK) return rc;
  auto fail = [&](int code) {
    vio_frontend_destroy(fe);
    return code;
  };
  const size_t px = (size_t)rows * cols;
  hipStream_t st = fe->stream;
  if (hipMemcpy2D(fe->pyr[0], cols, img, stride, cols, rows, hipMemcpyHostToDevice) != hipSuccess) return fail(VIO_ENODEV);
  if (dev_alloc(&fe->mask, px) != VIO_OK) return fail(VIO_ENOMEM);
  if (mask) {
    if (hipMemcpy2D(fe->mask, cols, mask, stride, cols, rows, hipMemcpyHostToDevice) != hipSuccess) return fail(VIO_ENODEV);
  } else if (hipMemset(fe->mask, 255, px) != hipSuccess) {
    return fail(VIO_ENODEV);
  }
  if (hipMemset(fe->max_bits, 0, sizeof(unsigned) * fe->nseg) != hipSuccess ||
      hipMemset(fe->n_cand, 0, sizeof(int) * fe->nseg) != hipSuccess)
    return fail(VIO_ENODEV);
  dim3 tb(256), tg((cols + kDetWaves * kDetW - 1) / (kDetWaves * kDetW), fe->nseg, 1);
  hipLaunchKernelGGL(detect_kernel<true>, tg, tb, 0, st, fe->pyr[0], fe->ld.pyr_bytes, fe->mask, px, fe->kept_xy, fe->n_kept,
                     fe->cap, fe->hw, c.min_dist, fe->max_bits, rows, cols, fe->cand, fe->seg_cap, fe->n_cand, (const int *)nullptr, 0);
  SelectParams SP;
  SP.cap = fe->cap, SP.rows = rows, SP.cols = cols, SP.max_corners = max_corners, SP.min_dist = (float)c.min_dist;
  SP.fx = c.fx, SP.fy = c.fy, SP.cx = c.cx, SP.cy = c.cy;
  hipLaunchKernelGGL(corner_select_kernel, dim3(1), dim3(kSelThreads), 0, st, fe->cand, fe->seg_cap, fe->nseg, fe->n_cand, fe->max_bits,
                     c.quality_level, SP, fe->forw_pts, fe->cur_pts, fe->pre_pts, fe->ids, fe->track_cnt, fe->n_forw, fe->n_pts,
                     fe->n_id, fe->obs, fe->n_obs);
  if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return fail(VIO_ENODEV);
  int n = 0;
  if (hipMemcpy(&n, fe->n_pts, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(VIO_ENODEV);
  *n_corners = n;
  if (n > 0 && hipMemcpy(corners, fe->forw_pts, sizeof(float) * 2 * n, hipMemcpyDeviceToHost) != hipSuccess) return fail(VIO_ENODEV);
  return fail(VIO_OK);
}

__global__ __launch_bounds__(256) void ransac_only_kernel(const float *p1, const float *p2, int n, float thresh, double conf,
                                                          uint8_t *mask) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  RansacShared &R = *reinterpret_cast<RansacShared *>(smem_raw);
  fundamental_ransac_block(R, p1, p2, n, thresh, conf, mask);
}

int vio_fundamental_ransac(const VioConfig *cfg, const float *pts1, const float *pts2, int32_t n, uint8_t *inlier_mask) {
  if (!cfg || !pts1 || !pts2 || !inlier_mask || n < 0) return VIO_EINVAL;
  if (n == 0) return VIO_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return VIO_ENODEV;
  float *d1 = nullptr, *d2 = nullptr;
  uint8_t *dm = nullptr;
  int rc = dev_alloc(&d1, (size_t)2 * n);
  if (rc == VIO_OK) rc = dev_alloc(&d2, (size_t)2 * n);
  if (rc == VIO_OK) rc = dev_alloc(&dm, (size_t)n);
  auto done = [&](int code) {
    if (d1) (void)hipFree(d1);
    if (d2) (void)hipFree(d2);
    if (dm) (void)hipFree(dm);
    return code;
  };
  if (rc != VIO_OK) return done(rc);
  if (hipMemcpy(d1, pts1, sizeof(float) * 2 * n, hipMemcpyHostToDevice) != hipSuccess) return done(VIO_ENODEV);
  if (hipMemcpy(d2, pts2, sizeof(float) * 2 * n, hipMemcpyHostToDevice) != hipSuccess) return done(VIO_ENODEV);
  hipLaunchKernelGGL(ransac_only_kernel, dim3(1), dim3(256), sizeof(RansacShared), 0, d1, d2, n, (float)cfg->f_threshold,
                     cfg->f_confidence, dm);
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return done(VIO_ENODEV);
  if (hipMemcpy(inlier_mask, dm, n, hipMemcpyDeviceToHost) != hipSuccess) return done(VIO_ENODEV);
  return done(VIO_OK);
}

}  // extern "C"
