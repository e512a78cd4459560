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

// potrf9 on the VECTOR unit (round 6, asked for by the round-4 and round-5 reviews): lane m < 9 keeps row m of the block and row m
// of L^-1 in registers; the pivot column / pivot row reach the other rows as scalars (v_readlane of RAW registers: the scale
// 1 / L_cc is folded into the multiplier, so the broadcasts are off the pivot chain). Same storage as potrf9_inv_wave.
// Measured (profiles/r06_microbench.txt): 2.44 k cycles per block -- 18 v_readlane + 9 multiply-adds + the reciprocal-root chain
// per pivot, ~270 cycles -- against 2.65 k for the two-accumulator matrix-core form of rounds 3-5 and ~2.1 k for the
// one-instruction-per-pivot form the product uses since round 6 (the inverse in the border of the pivot tile). A variant with
// the square-root-free elimination on the chain (only 1 / d_c ahead of the next pivot, 1 / sqrt(d_c) beside it) measured 2.78 k:
// the block is bound by instruction issue, not by the length of the chain. Not adopted (the mark was <= 1.3 k).
__device__ __forceinline__ bool potrf9_rows_wave(ldsd D, ldsd ldinv_k, int lane) {
  const bool act = lane < kSB;
  const int m = act ? lane : 0;
  double A[kSB], E[kSB];
#pragma unroll
  for (int n = 0; n < kSB; n++) A[n] = D[m * kSB + n];  // (entries right of the diagonal are never used)
  VIO_SCHED_FENCE();
#pragma unroll
  for (int n = 0; n < kSB; n++) E[n] = n == m ? 1.0 : 0.0;
  double myinv = 0.0;
  double dcc = lane_bcast(A[0], 0);
#pragma unroll
  for (int c = 0; c < kSB; c++) {
    double an[kSB], en[kSB];
#pragma unroll
    for (int n = c + 1; n < kSB; n++) an[n] = lane_bcast(A[c], n);  // A[n][c], raw
#pragma unroll
    for (int n = 0; n <= c; n++) en[n] = lane_bcast(E[n], c);       // row c of the inverse so far, raw
    double y = __builtin_amdgcn_rsq(dcc);
    const double h = 0.5 * dcc;
    y = y * fma(-h * y, y, 1.5);
    const double l = A[c] * y;    // L[m][c]
    const double lzy = (m > c ? l : 0.0) * y;
    A[c] = l;
#pragma unroll
    for (int n = c + 1; n < kSB; n++) A[n] = fma(-lzy, an[n], A[n]);
    if (c + 1 < kSB) dcc = lane_bcast(A[c + 1], c + 1);
#pragma unroll
    for (int n = 0; n <= c; n++) E[n] = fma(-lzy, en[n], E[n]);
    myinv = m == c ? y : myinv;
  }
  if (act) {
#pragma unroll
    for (int n = 0; n < kSB; n++) {
      if (n <= m) D[m * kSB + n] = A[n];
      if (n < m) D[n * kSB + m] = E[n] * myinv;  // L^-1[m][n], transposed into the upper triangle
    }
    ldinv_k[m] = myinv;
  }
  const bool bad = act && !(myinv > 0.0 && myinv < 1.7976931348623157e308);
  return __builtin_amdgcn_ballot_w64(bad) == 0;
}

// experimental variants of the ROUND 3-5 potrf9_inv_wave (two accumulators): what does a pivot cost without the inverse accumulation / the Newton steps?
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

enum { M_V_NOE = 100, M_V_NONEWTON, M_V_NEITHER, M_P9_ROWS, M_P9_ROWS_CHECK, M_V_OLD, M_POTRF9 = 0, M_POTRF9_UPD, M_TRSM9, M_MFMA_CHAIN15, M_MFMA_DEP15, M_TILE_RMW5, M_POTRF16, M_READLANE_MV, M_LDS_RT, M_BAND_BESIDE_PANELS, M_POTRF16_RANK1, M_COUNT };
static const char *kNames[M_COUNT] = {"potrf9 (product: inverse in the tile border)", "potrf9 (product) + E update", "9x9 trsm (loads, 3 mfma, store)", "15 mfma, 5 accumulators x 3",
                                      "15 mfma, one accumulator", "5 tiles: acc load, 3 mfma, store", "potrf16 (16 pivots, four per step)",
                                      "9x9 mat-vec x2 by v_readlane", "dependent ds_read round trip",
                                      "potrf9 + E update beside 3 waves of panel-like LDS / matrix traffic",
                                      "potrf16, one pivot per step (rounds 2-5)"};

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
    if (mode == M_BAND_BESIDE_PANELS && wave > 0) {
      // the three panel waves of the window: tile loads, rank-9 updates, tile stores, for as long as the band block takes
      double x = D[lane] * 1e-3, y = E[lane] * 1e-3;
      for (int it = 0; it < 3; it++) {
        v4d acc[5];
#pragma unroll
        for (int q = 0; q < 5; q++) acc[q] = tile_load_acc_raw(T + q * 16 * 81, 81, 16, li, kq);
        VIO_SCHED_FENCE();
#pragma unroll
        for (int q = 0; q < 5; q++) {
          v4d c = acc[q];
#pragma unroll
          for (int s3 = 0; s3 < 3; s3++) c = mfma_f64(x, y, c);
          acc[q] = c;
        }
#pragma unroll
        for (int q = 0; q < 5; q++) tile_store_acc(T + q * 16 * 81, 81, 16, li, kq, acc[q]);
      }
    }
    if (wave == 0) {
      const long long t0 = clock64();
      if (mode == M_BAND_BESIDE_PANELS) {
        potrf9_inv_wave(D, E, true, ldinv, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double a[3], b[4];
        load_op9_raw(C, li, kq, a), load_linv9_raw(D, ldinv, li, kq, b);
        VIO_SCHED_FENCE();
        mask_op9(li, kq, a), mask_linv9(li, kq, b);
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s3 = 0; s3 < 3; s3++) acc = mfma_f64(a[s3], b[s3], acc);
        if (li < kSB)
#pragma unroll
          for (int r = 0; r < 3; r++)
            if (kq + 4 * r < kSB) C[(kq + 4 * r) * kSB + li] = acc[r];
      }
      if (mode == M_V_NOE) potrf9_variant<false, true>(D, ldinv, lane);
      if (mode == M_V_NONEWTON) potrf9_variant<true, false>(D, ldinv, lane);
      if (mode == M_V_NEITHER) potrf9_variant<false, false>(D, ldinv, lane);
      if (mode == M_V_OLD) potrf9_variant<true, true>(D, ldinv, lane);
      if (mode == M_P9_ROWS) potrf9_rows_wave(D, ldinv, lane);
      if (mode == M_P9_ROWS_CHECK) {  // (correctness: the vector-unit form against the matrix-core form on the same block)
        potrf9_rows_wave(D, ldinv, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double mine = lane < 81 ? D[lane] : 0.0, mine2 = lane + 64 < 81 ? D[lane + 64] : 0.0, li_ = lane < 9 ? ldinv[lane] : 0.0;
        for (int i = lane; i < 81; i += 64) D[i] = gD[i];
        __builtin_amdgcn_wave_barrier();
        potrf9_inv_wave(D, E, false, ldinv, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double d = fabs(mine - (lane < 81 ? D[lane] : 0.0)) + fabs(mine2 - (lane + 64 < 81 ? D[lane + 64] : 0.0)) + fabs(li_ - (lane < 9 ? ldinv[lane] : 0.0));
        d = wave_max_f64(d);
        if (lane == 0) gout[300] = d;
      }
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
      if (mode == M_POTRF16_RANK1) potrf16_wave_rank1(T, T, 81, 16, 16, false, ldinv + 16, lane);
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
  if (mode == M_P9_ROWS_CHECK) {
    double d = -1;
    hipMemcpy(&d, dout + 300, sizeof(d), hipMemcpyDeviceToHost);
    printf("potrf9 on the vector unit vs the matrix-core form: max |difference| of L, L^-1, 1 / L_cc = %.3e\n", d);
    return;
  }
  printf("%-40s %8lld cycles%s\n", mode >= 100 ? (mode == M_V_NOE ? "potrf9, 2 accumulators, without the L^-1 mfma" : mode == M_V_NONEWTON ? "potrf9, 2 accumulators, 2 Newton steps -> 0" : mode == M_P9_ROWS ? "potrf9 on the vector unit (rows in lanes)" : mode == M_V_OLD ? "potrf9, 2 accumulators (rounds 3-5, 2 Newton steps)" : "potrf9, 2 accumulators, without either") : kNames[mode], c, mode == M_LDS_RT ? " per 10" : "");
}

int main() {
  std::vector<double> D(81);
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < 9; j++) D[i * 9 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
  double *dD, *dout;
  long long *dcyc;
  hipMalloc(&dD, 81 * 8), hipMalloc(&dout, 512 * 8), hipMalloc(&dcyc, 8);
  hipMemcpy(dD, D.data(), 81 * 8, hipMemcpyHostToDevice);
  run<M_V_NOE>(dD, dout, dcyc), run<M_V_NONEWTON>(dD, dout, dcyc), run<M_V_NEITHER>(dD, dout, dcyc);
  run<M_V_OLD>(dD, dout, dcyc), run<M_P9_ROWS>(dD, dout, dcyc), run<M_P9_ROWS_CHECK>(dD, dout, dcyc);
  run<M_POTRF9>(dD, dout, dcyc), run<M_POTRF9_UPD>(dD, dout, dcyc), run<M_TRSM9>(dD, dout, dcyc), run<M_MFMA_CHAIN15>(dD, dout, dcyc);
  run<M_MFMA_DEP15>(dD, dout, dcyc), run<M_TILE_RMW5>(dD, dout, dcyc), run<M_POTRF16>(dD, dout, dcyc), run<M_READLANE_MV>(dD, dout, dcyc);
  run<M_LDS_RT>(dD, dout, dcyc), run<M_BAND_BESIDE_PANELS>(dD, dout, dcyc), run<M_POTRF16_RANK1>(dD, dout, dcyc);
  return 0;
}
