// chol_bench.hip — the reduced-system Cholesky of the window solver in isolation (tools only, not product).
// Runs cholesky_blocks / cholesky_backsolve of csrc/solver_core.h on a 165x165 SPD matrix held in LDS exactly as the
// solver holds it (one workgroup of 512 threads per CU), checks the solve against a host factorization and reports
// shader cycles (s_memtime) of the whole and of its pieces.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../vins-mobile_amd/csrc -o bin/chol_bench chol_bench.hip
//   bin/chol_bench [workgroups]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "solver_core.h"

using namespace vio;

constexpr int NB = 11, NP = NB * kBS, NBLK = NB * (NB + 1) / 2;

struct Layout {
  ldsd Hm, ldinv, rhs, red;
  ldsi flag, blk_ij;
};
__device__ Layout carve(ldsd base) {
  Layout L;
  L.Hm = base;
  L.ldinv = base + NBLK * kBB + 16;
  L.rhs = L.ldinv + 176;
  L.red = L.rhs + 176;
  L.flag = (ldsi)(L.red + 64);
  L.blk_ij = L.flag + 8;
  return L;
}
constexpr size_t kLdsBytes = (NBLK * kBB + 16 + 176 * 2 + 64 + 8 + 40) * 8;

enum { M_FULL = 0, M_POTRF, M_UPDATE1, M_UPDATE7, M_TRSM1, M_BACKSOLVE_ONLY, M_COUNT };

template <int mode>
__global__ __launch_bounds__(512) void chol_kernel(const double *gH, double *gout, long long *cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  Layout L = carve((ldsd)smem);
  Ctx cx;
  cx.tid = threadIdx.x, cx.nt = 512, cx.red = L.red, cx.prof = nullptr, cx.lprof = nullptr;
  WinView v;
  v.nblk = NB, v.np = NP;
  WorkT<ldsd> w;
  w.Hm = L.Hm, w.ldinv = L.ldinv, w.flag = L.flag, w.blk_ij = L.blk_ij;
  const int wave = cx.tid >> 6, lane = cx.tid & 63;
  const LaneMap m = lane_map(lane);
  long long tsum[2] = {0, 0};
  for (int q = cx.tid; q < NBLK; q += 512) {
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= q) bi++;
    L.blk_ij[q] = (bi << 8) | (q - bi * (bi + 1) / 2);
  }
  for (int rep = 0; rep < reps; rep++) {
    for (int i = cx.tid; i < NBLK * kBB; i += 512) L.Hm[i] = gH[i];
    for (int i = cx.tid; i < NP; i += 512) L.rhs[i] = 1.0 + 0.01 * i;
    if (cx.tid == 0) L.flag[0] = L.flag[1] = 0;
    __syncthreads();
    long long t0 = clock64();
    if (mode == M_FULL) {
      bool ok = cholesky_blocks(cx, v, w, L.rhs);
      long long t1 = clock64();
      cholesky_backsolve(cx, v, w, L.rhs);
      tsum[0] += t1 - t0, tsum[1] += clock64() - t1;
      if (!ok && cx.tid == 0) gout[NBLK * kBB + NP] = -1;
    } else if (mode == M_POTRF) {  // wave 0 factors the 11 diagonal blocks as they are, the others wait
      if (wave == 0)
        for (int k = 0; k < NB; k++) potrf15_inv_wave(L.Hm + blk_off(k, k), L.Hm, false, L.ldinv + k * kBS, lane);
      tsum[0] += clock64() - t0;
    } else if (mode == M_UPDATE1 || mode == M_UPDATE7) {  // 20 double block updates by wave 1 alone / by waves 1..7
      if (mode == M_UPDATE1 ? wave == 1 : wave >= 1)
        for (int q = 0; q < 20; q++) {
          const int b0 = 1 + ((wave * 5 + q) % 9), b1 = 1 + ((wave * 7 + q + 3) % 9);
          block_update2(L.Hm + blk_off(10, b0), L.Hm + blk_off(9, b0 - 1), L.Hm + blk_off(8, b1 % 8),
                        L.Hm + blk_off(10, b1), L.Hm + blk_off(9, b1 - 1), L.Hm + blk_off(8, b0 % 8), true, m);
        }
      long long t1 = clock64();
      if (wave == 1 && lane == 0) L.flag[3] = (int)(t1 - t0);
      __syncthreads();
      tsum[0] += L.flag[3];
    } else if (mode == M_TRSM1) {  // 20 panel blocks by wave 1 alone (the diagonal block holds whatever is there)
      if (wave == 1)
        for (int q = 0; q < 20; q++) block_trsm(L.Hm + blk_off(10, q % 9), L.Hm + blk_off(q % 7, q % 7), L.ldinv, m, lane);
      long long t1 = clock64();
      if (wave == 1 && lane == 0) L.flag[3] = (int)(t1 - t0);
      __syncthreads();
      tsum[0] += L.flag[3];
    }
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    if (cx.tid == 0) cyc[0] = tsum[0] / reps, cyc[1] = tsum[1] / reps;
    for (int i = cx.tid; i < NBLK * kBB; i += 512) gout[i] = L.Hm[i];
    for (int i = cx.tid; i < NP; i += 512) gout[NBLK * kBB + i] = L.rhs[i];
  }
}

typedef void (*kern_t)(const double *, double *, long long *, int);

int main(int argc, char **argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256;
  kern_t kern[M_COUNT] = {chol_kernel<0>, chol_kernel<1>, chol_kernel<2>, chol_kernel<3>, chol_kernel<4>, chol_kernel<5>};
  const char *names[M_COUNT] = {"cholesky_blocks | cholesky_backsolve", "wave 0: 11 x potrf15_inv_wave", "wave 1 alone: 20 x block_update2",
                                "waves 1-7: 20 x block_update2 each", "wave 1 alone: 20 x block_trsm", "(unused)"};
  // SPD test matrix, like a damped reduced Hessian
  std::vector<double> A((size_t)NP * NP), M((size_t)NP * NP);
  srand(7);
  for (auto &x : M) x = (rand() / (double)RAND_MAX - 0.5);
  for (int i = 0; i < NP; i++)
    for (int j = 0; j < NP; j++) {
      double s = 0;
      for (int k = 0; k < NP; k++) s += M[i * NP + k] * M[j * NP + k];
      A[i * NP + j] = s / NP * 50 + (i == j ? 3.0 : 0.0);
    }
  std::vector<double> hH((size_t)NBLK * kBB);
  for (int bi = 0; bi < NB; bi++)
    for (int bj = 0; bj <= bi; bj++)
      for (int r = 0; r < kBS; r++)
        for (int c = 0; c < kBS; c++) hH[(bi * (bi + 1) / 2 + bj) * kBB + r * kBS + c] = A[(bi * kBS + r) * NP + bj * kBS + c];
  std::vector<double> Lh = A, xh(NP);
  for (int j = 0; j < NP; j++) {
    for (int k = 0; k < j; k++)
      for (int i = j; i < NP; i++) Lh[i * NP + j] -= Lh[i * NP + k] * Lh[j * NP + k];
    double d = sqrt(Lh[j * NP + j]);
    for (int i = j; i < NP; i++) Lh[i * NP + j] /= d;
  }
  for (int i = 0; i < NP; i++) xh[i] = 1.0 + 0.01 * i;
  for (int i = 0; i < NP; i++) {
    for (int k = 0; k < i; k++) xh[i] -= Lh[i * NP + k] * xh[k];
    xh[i] /= Lh[i * NP + i];
  }
  for (int i = NP - 1; i >= 0; i--) {
    for (int k = i + 1; k < NP; k++) xh[i] -= Lh[k * NP + i] * xh[k];
    xh[i] /= Lh[i * NP + i];
  }
  double *dH, *dout;
  long long *dc;
  (void)hipMalloc(&dH, hH.size() * 8), (void)hipMalloc(&dout, (hH.size() + NP + 8) * 8), (void)hipMalloc(&dc, 64);
  (void)hipMemcpy(dH, hH.data(), hH.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemset(dout, 0, (hH.size() + NP + 8) * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  for (int mode = 0; mode < M_COUNT - 1; mode++) {
    const int reps = 8;
    (void)hipFuncSetAttribute((const void *)kern[mode], hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    hipLaunchKernelGGL(kern[mode], dim3(nwg), dim3(512), kLdsBytes, 0, dH, dout, dc, reps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern[mode], dim3(nwg), dim3(512), kLdsBytes, 0, dH, dout, dc, reps);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c[2];
    (void)hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-40s: %8lld | %8lld cycles per rep; kernel %.3f ms for %d reps x %d workgroups\n", names[mode], c[0], c[1], ms, reps, nwg);
    if (mode == M_FULL) {
      std::vector<double> out(hH.size() + NP + 8);
      (void)hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
      double err = 0, nrm = 0;
      for (int i = 0; i < NP; i++) err = fmax(err, fabs(out[hH.size() + i] - xh[i])), nrm = fmax(nrm, fabs(xh[i]));
      printf("        solve error vs host: %.3e (max |x| %.3e), failure flag %g\n", err, nrm, out[hH.size() + NP]);
    }
  }
  return 0;
}
