// mfma_share.hip — does v_mfma_f64_16x16x4 issued from several waves of one CU (one wave per SIMD) run concurrently?
// One workgroup of 256 threads; `active` waves each run a chain of N dependent-free f64 MFMAs (4 accumulators);
// reports the cycles wave 0 needed. (tools only)
//   hipcc --offload-arch=gfx950 -O3 -o bin/mfma_share mfma_share.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int kind>
__global__ __launch_bounds__(256, 2) void k(double *out, long long *cyc, int active, int n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-3, b = 0.5 - lane * 1e-3;
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  if (wave < active) {
    for (int i = 0; i < n; i++) {
      if (kind == 0) {  // independent accumulators
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      } else if (kind == 1) {  // one accumulator (dependent chain)
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      } else {  // f64 VALU fma chain, 4 independent
        c0 += a * b; c1 += a * b; c2 += a * b; c3 += a * b;
        c0 = c0 * a + b; c1 = c1 * a + b; c2 = c2 * a + b; c3 = c3 * a + b;
      }
    }
  }
  double s = c0[0] + c1[1] + c2[2] + c3[3] + c0[3];
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(s));
  long long t1 = __builtin_readcyclecounter() + (lo & 0);
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = s;
}
int main() {
  double *out; long long *cyc;
  hipMalloc(&out, 256 * 8), hipMalloc(&cyc, 8);
  const int n = 64;
  for (int kind = 0; kind < 3; kind++)
    for (int active = 1; active <= 4; active++) {
      for (int rep = 0; rep < 2; rep++) {
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, out, cyc, active, n);
        if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, out, cyc, active, n);
        if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, out, cyc, active, n);
        hipDeviceSynchronize();
      }
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("%s, %d active wave(s): %lld cycles for %d x 4 -> %.1f cycles per instruction (wave 0)\n",
             kind == 0 ? "mfma f64, 4 accumulators" : kind == 1 ? "mfma f64, 1 accumulator " : "f64 VALU fma (16 per iter)", active, c, n, (double)c / (n * (kind == 2 ? 16 : 4)));
    }
  return 0;
}
