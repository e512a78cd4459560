// icache_probe.hip — what straight-line code costs on gfx950 once it no longer fits the instruction cache.
// The window kernel is one body of ~66 k instructions (~450 KB) that every wave walks once per trust-region iteration; the
// shader instruction cache is 64 KB per pair of CUs. Here: a loop whose body is KB kilobytes of independent v_fma_f64
// (8 bytes each, 4 accumulators in rotation: issue-bound at 4 cycles per instruction when every fetch hits), run by
// 1 / 4 / 8 waves per CU on every CU; reported: cycles per instruction by body size.
// build (one binary per body size: the assembler's branch relaxation does not see through .rept across many kernels of one file):
//   for kb in 8 32 48 64 96 192 384; do hipcc --offload-arch=gfx950 -O3 -DBODY_KB=$kb -o bin/icache_probe_$kb icache_probe.hip; done
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <utility>
#ifndef BODY_KB
#define BODY_KB 64
#endif

#define FMA4 "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"

// one piece of code of KB kilobytes (<= 96: a branch reaches 128 KB); ID makes every instance its own copy
template <int KB, int ID>
__device__ __attribute__((noinline)) void piece(double &a, double &b, double &c, double &d, double y, double z) {
  asm volatile(".rept %c6\n" FMA4 ".endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y), "v"(z), "n"(KB * 32));
}
template <int... ID>
__device__ __forceinline__ void call_pieces(std::integer_sequence<int, ID...>, double &a, double &b, double &c, double &d, double y, double z) {
  (piece<16, ID>(a, b, c, d, y, z), ...);
}
// KB kilobytes of code per trip: 128 instructions (32 x FMA4) = 1 KB; bodies past 96 KB are chains of 64 KB pieces (calls)
template <int KB>
__global__ __launch_bounds__(512) void stream_code(double *out, long long *cyc, int trips) {
  double a = 1.0 + 1e-9 * threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  const double y = 0.9999999, z = 1e-9;
  unsigned long long t0, t1;
  // one untimed trip warms whatever can be warmed
  for (int t = -1; t < trips; t++) {
    if (t == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if constexpr (KB < 16) piece<KB, 0>(a, b, c, d, y, z);
    else call_pieces(std::make_integer_sequence<int, KB / 16>(), a, b, c, d, y, z);
  }
  asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (long long)(t1 - t0);
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

template <int KB>
static void run(int waves_per_block, int blocks, double *out, long long *cyc) {
  const int trips = 16;
  hipLaunchKernelGGL(stream_code<KB>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, trips);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * waves_per_block);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double s = 0;
  long long mx = 0;
  for (auto x : h) s += (double)x, mx = x > mx ? x : mx;
  const double instr = (double)trips * KB * 128;
  // s_memtime counts a constant 100 MHz clock on this part: convert with the shader clock the caller passes through clock rate
  printf("code %4d KB  blocks %4d x %d waves: mean %8.3f ticks/instr   max %8.3f ticks/instr\n", KB, blocks, waves_per_block,
         s / h.size() / instr, (double)mx / instr);
}

int main() {
  double *out;
  long long *cyc;
  hipMalloc(&out, 4096 * 512 * sizeof(double));
  hipMalloc(&cyc, 4096 * 8 * sizeof(long long));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s  CUs %d  clock %d kHz  (s_memtime ticks: compare rows with each other; 8 KB row = all-hit reference)\n", p.name,
         p.multiProcessorCount, p.clockRate);
  const int cus = p.multiProcessorCount;
  for (int wpb : {1, 4, 8}) {
    for (int blocks : {1, cus, 2 * cus}) {
      if (wpb == 8 && blocks == 2 * cus) continue;
      run<BODY_KB>(wpb, blocks, out, cyc);
    }
  }
  return 0;
}
