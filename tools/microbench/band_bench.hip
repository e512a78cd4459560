// band_bench.hip — the wave-level pieces of the structured factorization (csrc/solver_core.h) in isolation (tools
// only, not product): s_memtime cycles per call for one workgroup of 256 threads on an otherwise idle CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I../../include -I../../vins-mobile_amd/csrc -o bin/band_bench band_bench.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "solver_core.h"

using namespace vio;

// experimental variants of potrf9_inv_wave: what does a pivot cost without the inverse accumulation / the Newton steps?
template <bool WITH_E, bool NEWTON>
__device__ __forceinline__ bool potrf9_variant(ldsd D, ldsd ldinv_k, int lane) {
  const int n = lane & 15, kq = lane >> 4;
  v4d A, E;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int m = kq + 4 * r;
    const bool ok = m < kSB && n < kSB;
    const int hi = m > n ? m : n, lo = m > n ? n : m;
    const double x = D[ok ? hi * kSB + lo : 0];
    A[r] = ok ? x : 0.0;
    E[r] = (m == n) ? 1.0 : 0.0;
  }
  double keep[3] = {0.0, 0.0, 0.0}, myinv = 0.0;
  double dcc = lane_bcast(A[0], 0);
#pragma unroll
  for (int c = 0; c < kSB; c++) {
    double y = __builtin_amdgcn_rsq(dcc);
    if (NEWTON) {
      const double h = 0.5 * dcc;
      y = y * fma(-h * y, y, 1.5);
      y = y * fma(-h * y, y, 1.5);
    }
    const bool sel = kq == (c & 3);
    const double a = sel ? A[c >> 2] * y : 0.0;
    const double e = sel ? E[c >> 2] * y : 0.0;
    keep[c >> 2] = sel ? (n >= c ? a : e) : keep[c >> 2];
    myinv = (n == c) ? y : myinv;
    if (c + 1 < kSB) {
      const double lnext = lane_bcast(a, 16 * (c & 3) + c + 1);
      const double dold = lane_bcast(A[(c + 1) >> 2], 16 * ((c + 1) & 3) + c + 1);
      dcc = fma(-lnext, lnext, dold);
    }
    A = mfma_f64(-a, a, A);
    if (WITH_E) E = mfma_f64(-a, e, E);
  }
  if (n < kSB) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int c = kq + 4 * j;
      if (c < kSB) D[n * kSB + c] = keep[j];
    }
    if (kq == 0) ldinv_k[n] = myinv;
  }
  return __builtin_amdgcn_ballot_w64(n < kSB && !(myinv > 0.0)) == 0;
}

enum { M_V_NOE = 100, M_V_NONEWTON, M_V_NEITHER, M_POTRF9 = 0, M_POTRF9_UPD, M_TRSM9, M_MFMA_CHAIN15, M_MFMA_DEP15, M_TILE_RMW5, M_POTRF16, M_READLANE_MV, M_LDS_RT, M_COUNT };
static const char *kNames[M_COUNT] = {"potrf9 (no update)", "potrf9 + E update", "9x9 trsm (loads, 3 mfma, store)", "15 mfma, 5 accumulators x 3",
                                      "15 mfma, one accumulator", "5 tiles: acc load, 3 mfma, store", "potrf16 (16 pivots)",
                                      "9x9 mat-vec x2 by v_readlane", "dependent ds_read round trip"};

template <int mode>
__global__ __launch_bounds__(256, 2) void bench_kernel(const double *gD, double *gout, long long *cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  ldsd D = (ldsd)smem, E = D + 96, C = E + 96, ldinv = C + 96, T = ldinv + 32;  // T: 5 tiles with ld 81
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
  long long total = 0;
  double sink = 0.0;
  for (int rep = 0; rep < reps; rep++) {
    for (int i = tid; i < 81; i += 256) D[i] = gD[i], E[i] = 0.01 * gD[80 - i], C[i] = 0.02 * gD[i];
    for (int i = tid; i < 16 * 81 * 5; i += 256) T[i] = (i % 82 == 0) ? 40.0 : 0.01 * (i % 7);
    if (tid < 16) ldinv[tid] = 0.5;
    __syncthreads();
    if (wave == 0) {
      const long long t0 = clock64();
      if (mode == M_V_NOE) potrf9_variant<false, true>(D, ldinv, lane);
      if (mode == M_V_NONEWTON) potrf9_variant<true, false>(D, ldinv, lane);
      if (mode == M_V_NEITHER) potrf9_variant<false, false>(D, ldinv, lane);
      if (mode == M_POTRF9) potrf9_inv_wave(D, E, false, ldinv, lane);
      if (mode == M_POTRF9_UPD) potrf9_inv_wave(D, E, true, ldinv, lane);
      if (mode == M_TRSM9) {
        double a[3], b[4];
        load_op9_raw(C, li, kq, a), load_linv9_raw(D, ldinv, li, kq, b);
        VIO_SCHED_FENCE();
        mask_op9(li, kq, a), mask_linv9(li, kq, b);
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 3; s++) acc = mfma_f64(a[s], b[s], acc);
        if (li < kSB)
#pragma unroll
          for (int r = 0; r < 3; r++)
            if (kq + 4 * r < kSB) C[(kq + 4 * r) * kSB + li] = acc[r];
      }
      if (mode == M_MFMA_CHAIN15 || mode == M_MFMA_DEP15) {
        v4d acc[5];
        double x = D[lane], y = E[lane];
#pragma unroll
        for (int q = 0; q < 5; q++) acc[q] = v4d{x, y, x, y};
        if (mode == M_MFMA_CHAIN15) {
#pragma unroll
          for (int s = 0; s < 3; s++)
#pragma unroll
            for (int q = 0; q < 5; q++) acc[q] = mfma_f64(x, y, acc[q]);
        } else {
#pragma unroll
          for (int s = 0; s < 15; s++) acc[0] = mfma_f64(x, y, acc[0]);
        }
#pragma unroll
        for (int q = 0; q < 5; q++) sink += acc[q][0] + acc[q][3];
      }
      if (mode == M_TILE_RMW5) {
        v4d acc[5];
        double x = D[lane], y = E[lane];
#pragma unroll
        for (int q = 0; q < 5; q++) acc[q] = tile_load_acc_raw(T + q * 16 * 81, 81, 16, li, kq);
        VIO_SCHED_FENCE();
#pragma unroll
        for (int q = 0; q < 5; q++) {
          v4d c = tile_mask_acc(acc[q], 16, kq);
#pragma unroll
          for (int s = 0; s < 3; s++) c = mfma_f64(x, y, c);
          acc[q] = c;
        }
#pragma unroll
        for (int q = 0; q < 5; q++) tile_store_acc(T + q * 16 * 81, 81, 16, li, kq, acc[q]);
      }
      if (mode == M_POTRF16) potrf16_wave(T, T, 81, 16, 16, false, ldinv + 16, lane);
      if (mode == M_READLANE_MV) {
        double er[kSB], lr[kSB], prev = D[lane % 9], val = E[lane % 9];
#pragma unroll
        for (int m = 0; m < kSB; m++) er[m] = D[(lane % 9) * 9 + m], lr[m] = E[(lane % 9) * 9 + m];
        VIO_SCHED_FENCE();
#pragma unroll
        for (int m = 0; m < kSB; m++) val = fma(-er[m], lane_bcast(prev, m), val);
        double u = 0.0;
#pragma unroll
        for (int m = 0; m < kSB; m++) u = fma(lr[m], lane_bcast(val, m), u);
        sink += u;
      }
      if (mode == M_LDS_RT) {
        int idx = lane;
#pragma unroll
        for (int q = 0; q < 10; q++) idx = (int)D[idx & 63] & 63;
        sink += idx;
      }
      const long long t1 = clock64();
      if (rep > 0) total += t1 - t0;
    }
    __syncthreads();
  }
  if (tid == 0) cyc[blockIdx.x] = total / (reps - 1);
  gout[blockIdx.x * 256 + tid] = sink + D[tid % 81] + T[tid];
}

template <int mode>
static void run(const double *dD, double *dout, long long *dcyc) {
  hipLaunchKernelGGL(bench_kernel<mode>, dim3(1), dim3(256), 60000, 0, dD, dout, dcyc, 21);
  long long c = 0;
  hipDeviceSynchronize();
  hipMemcpy(&c, dcyc, sizeof(c), hipMemcpyDeviceToHost);
  printf("%-40s %8lld cycles%s\n", mode >= 100 ? (mode == M_V_NOE ? "potrf9 without the L^-1 mfma" : mode == M_V_NONEWTON ? "potrf9 without Newton steps" : "potrf9 without either") : kNames[mode], c, mode == M_LDS_RT ? " per 10" : "");
}

int main() {
  std::vector<double> D(81);
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < 9; j++) D[i * 9 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
  double *dD, *dout;
  long long *dcyc;
  hipMalloc(&dD, 81 * 8), hipMalloc(&dout, 256 * 8), hipMalloc(&dcyc, 8);
  hipMemcpy(dD, D.data(), 81 * 8, hipMemcpyHostToDevice);
  run<M_V_NOE>(dD, dout, dcyc), run<M_V_NONEWTON>(dD, dout, dcyc), run<M_V_NEITHER>(dD, dout, dcyc);
  run<M_POTRF9>(dD, dout, dcyc), run<M_POTRF9_UPD>(dD, dout, dcyc), run<M_TRSM9>(dD, dout, dcyc), run<M_MFMA_CHAIN15>(dD, dout, dcyc);
  run<M_MFMA_DEP15>(dD, dout, dcyc), run<M_TILE_RMW5>(dD, dout, dcyc), run<M_POTRF16>(dD, dout, dcyc), run<M_READLANE_MV>(dD, dout, dcyc);
  run<M_LDS_RT>(dD, dout, dcyc);
  return 0;
}
