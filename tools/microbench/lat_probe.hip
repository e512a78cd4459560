// lat_probe.hip — gfx950 latency probes behind the back-end's design decisions (tools only, not product):
// dependent-chain cycles of the f64 VALU / transcendental / MFMA / LDS / L2 / barrier primitives the window solver is
// built from, measured with s_memtime on one workgroup of 512 threads (the solver's shape).
//   hipcc --offload-arch=gfx950 -O3 -o lat_probe lat_probe.hip && ./lat_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
#define N 256

__global__ __launch_bounds__(512) void probe(double *out, long long *cyc, const double *gbuf, int *gidx) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16384; i += 512) lds[i] = 1.0 + 1e-9 * i;
  __shared__ int chase[1024];
  for (int i = tid; i < 1024; i += 512) chase[i] = (i * 37 + 11) & 1023;
  __syncthreads();
  long long t0, t1;
  double x = 1.0 + 1e-12 * tid, y = 0.999999, acc = 0;
  int k = 0;
#define TIC() __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#define TOC(slot) __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0); if (tid == 0) cyc[slot] = t1 - t0;
  // 0: dependent fma f64
  if (wave == 0) {
    TIC();
#pragma unroll
    for (int i = 0; i < N; i++) x = fma(x, y, 1e-9);
    TOC(0);
    // 1: dependent mul f64
    TIC();
#pragma unroll
    for (int i = 0; i < N; i++) x = x * y;
    TOC(1);
    // 2: independent fma f64 (4 chains)
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    TIC();
#pragma unroll
    for (int i = 0; i < N / 4; i++) a0 = fma(a0, y, 1e-9), a1 = fma(a1, y, 1e-9), a2 = fma(a2, y, 1e-9), a3 = fma(a3, y, 1e-9);
    TOC(2);
    x = a0 + a1 + a2 + a3;
    // 3: dependent rsq f64
    x = fabs(x) + 1.0;
    TIC();
#pragma unroll
    for (int i = 0; i < N; i++) x = __builtin_amdgcn_rsq(x) + 1.0;
    TOC(3);
    // 4: dependent rcp f64
    TIC();
#pragma unroll
    for (int i = 0; i < N; i++) x = __builtin_amdgcn_rcp(x) + 1.0;
    TOC(4);
    // 5: IEEE division chain
    TIC();
#pragma unroll
    for (int i = 0; i < 64; i++) x = 1.0 / x + 1.0;
    TOC(5);
    // 6: IEEE sqrt chain
    TIC();
#pragma unroll
    for (int i = 0; i < 64; i++) x = sqrt(x) + 1.0;
    TOC(6);
    // 7: readlane -> fma chain (VALU -> SGPR -> VALU)
    TIC();
#pragma unroll
    for (int i = 0; i < N; i++) {
      int lo = __builtin_amdgcn_readlane(__double2loint(x), 17), hi = __builtin_amdgcn_readlane(__double2hiint(x), 17);
      x = fma(__hiloint2double(hi, lo), y, x);
    }
    TOC(7);
    // 8: dependent mfma f64 16x16x4 (accumulator chain)
    v4d c = {x, x, x, x};
    TIC();
#pragma unroll
    for (int i = 0; i < 64; i++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c, 0, 0, 0);
    TOC(8);
    // 9: independent mfma f64 (4 accumulators)
    v4d c1 = c, c2 = c, c3 = c;
    TIC();
#pragma unroll
    for (int i = 0; i < 16; i++) {
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c, 0, 0, 0), c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c2, 0, 0, 0), c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c3, 0, 0, 0);
    }
    TOC(9);
    x = c[0] + c1[1] + c2[2] + c3[3];
    // 10: mfma -> VALU read -> mfma (rank-1 pattern: a = acc * y; acc = mfma(a, a, acc))
    TIC();
#pragma unroll
    for (int i = 0; i < 64; i++) {
      double a = c[0] * y;
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c, 0, 0, 0);
    }
    TOC(10);
    x += c[0];
    // 11: dependent LDS read chain (ds_read_b32 pointer chase)
    k = lane;
    TIC();
#pragma unroll
    for (int i = 0; i < N; i++) k = chase[k];
    TOC(11);
    // 12: dependent LDS f64 read-modify chain: x = lds[idx(x)]
    TIC();
#pragma unroll
    for (int i = 0; i < 64; i++) { k = (k + 7) & 1023; x += lds[k + (int)(x * 1e-30)]; }
    TOC(12);
    // 13: LDS atomic add f64 throughput (64 lanes, distinct addresses), 64 in a row
    TIC();
#pragma unroll
    for (int i = 0; i < 64; i++) __hip_atomic_fetch_add(&lds[2048 + ((lane * 17 + i * 64) & 4095)], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    TOC(13);
    // 14: dependent global load chain (L2 hits after first touch): pointer chase through gidx
    k = lane & 7;
    for (int i = 0; i < 64; i++) k = gidx[k];
    TIC();
#pragma unroll
    for (int i = 0; i < 64; i++) k = gidx[k];
    TOC(14);
    // 15: global atomic add f64 (returnless) x 64, then wait
    TIC();
#pragma unroll
    for (int i = 0; i < 16; i++) __hip_atomic_fetch_add((double *)gbuf + 4096 + lane + 64 * i, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    TOC(15);
    // 16: v_rsq + 2 newton (the potrf pivot chain) dependent
    x = fabs(x) + 2.0;
    TIC();
#pragma unroll
    for (int i = 0; i < 64; i++) {
      double r = __builtin_amdgcn_rsq(x);
      const double h = 0.5 * x;
      r = r * fma(-h * r, r, 1.5);
      r = r * fma(-h * r, r, 1.5);
      x = r + 2.0;
    }
    TOC(16);
  }
  __syncthreads();
  // 17: 64 barriers, all 8 waves
  TIC();
#pragma unroll
  for (int i = 0; i < 64; i++) __syncthreads();
  TOC(17);
  // 18: 64 x (trivial LDS phase + barrier): each thread writes one double, barrier, reads neighbour
  TIC();
#pragma unroll 8
  for (int i = 0; i < 64; i++) {
    lds[tid] = x;
    __syncthreads();
    x += lds[(tid + 64) & 511];
  }
  TOC(18);
  // 19: phase with a global (L2) round trip: store, barrier, load neighbour
  double *g = (double *)gbuf + 8192;
  TIC();
#pragma unroll 8
  for (int i = 0; i < 64; i++) {
    g[tid] = x;
    __syncthreads();
    x += g[(tid + 64) & 511];
  }
  TOC(19);
  // 20: two waves on one SIMD issuing MFMA (wave 0 and 4 presumably share SIMD 0): wave 0 timed, dependent chain
  {
    v4d c = {x, x, x, x};
    __syncthreads();
    TIC();
    if (wave == 0 || wave == 4) {
#pragma unroll
      for (int i = 0; i < 64; i++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c, 0, 0, 0);
    }
    TOC(20);
    x += c[0];
    __syncthreads();
    TIC();
    if (wave == 0 || wave == 1) {
#pragma unroll
      for (int i = 0; i < 64; i++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c, 0, 0, 0);
    }
    TOC(21);
    x += c[0];
    __syncthreads();
    TIC();  // all 8 waves MFMA
#pragma unroll
    for (int i = 0; i < 64; i++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c, 0, 0, 0);
    TOC(22);
    x += c[0];
  }
  out[tid] = x + acc + k;
}

int main() {
  double *out, *gbuf;
  long long *cyc;
  int *gidx;
  hipMalloc(&out, 512 * 8), hipMalloc(&gbuf, 65536 * 8), hipMalloc(&cyc, 64 * 8), hipMalloc(&gidx, 4096 * 4);
  hipMemset(gbuf, 0, 65536 * 8);
  std::vector<int> h(4096);
  for (int i = 0; i < 4096; i++) h[i] = (i * 613 + 7) & 4095;
  hipMemcpy(gidx, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(512), 140 * 1024, 0, out, cyc, gbuf, gidx);
    hipDeviceSynchronize();
  }
  long long c[64];
  hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  const char *names[] = {"dep fma f64 x256", "dep mul f64 x256", "4 indep fma chains x256 total", "dep rsq f64(+add) x256", "dep rcp f64(+add) x256",
                         "dep IEEE div(+add) x64", "dep IEEE sqrt(+add) x64", "readlane x2 -> fma x256", "dep mfma f64 16x16x4 x64",
                         "4 indep mfma f64 x64 total", "acc*y -> mfma(a,a,acc) x64", "dep ds_read_b32 chase x256", "dep ds_read_b64 + add x64",
                         "ds_add_f64 x64 (64 lanes)", "dep global load (L2) x64", "global atomic add f64 x16", "rsq + 2 newton x64",
                         "s_barrier x64 (8 waves)", "lds write + barrier + read x64", "global write + barrier + read x64",
                         "dep mfma x64, waves 0+4", "dep mfma x64, waves 0+1", "dep mfma x64, all 8 waves"};
  const int cnt[] = {256, 256, 256, 256, 256, 64, 64, 256, 64, 64, 64, 256, 64, 64, 64, 16, 64, 64, 64, 64, 64, 64, 64};
  for (int i = 0; i < 23; i++) printf("%-40s %8lld cycles  %7.1f / op\n", names[i], c[i], (double)c[i] / cnt[i]);
  return 0;
}
