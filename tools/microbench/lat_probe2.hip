// lat_probe2.hip — dependent-issue latencies of the f64 VALU ops on gfx950 (inline asm, so nothing is folded).
#include <hip/hip_runtime.h>
#include <stdio.h>

#define TIC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define TOC(slot) asm volatile("s_nop 7\n s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); if (threadIdx.x == 0) cyc[slot] = t1 - t0
#define REP64(X) _Pragma("unroll") for (int i_ = 0; i_ < 64; i_++) { X; }

__global__ __launch_bounds__(512) void probe(double *out, long long *cyc) {
  unsigned long long t0, t1;
  double x = 1.0 + 1e-9 * threadIdx.x, y = 0.9999999, z = 1e-9;
  const int wave = threadIdx.x >> 6;
  float xf = 1.5f;
  if (wave == 0) {
    TIC(); REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))); TOC(0);
    TIC(); REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y))); TOC(1);
    TIC(); REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(z))); TOC(2);
    double a = x, b = x + 1, c = x + 2, d = x + 3;
    TIC(); REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y), "v"(z))); TOC(3);
    x = a + b + c + d;
    x = fabs(x) + 1.5;
    TIC(); REP64(asm volatile("v_rsq_f64 %0, %0" : "+v"(x))); TOC(4);
    x = fabs(x) + 1.5;
    TIC(); REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(x))); TOC(5);
    x = fabs(x) + 1.5;
    TIC(); REP64(asm volatile("v_sqrt_f64 %0, %0" : "+v"(x))); TOC(6);
    // f32 rsq for comparison
    TIC(); REP64(asm volatile("v_rsq_f32 %0, %0" : "+v"(xf))); TOC(7);
    x += xf;
    // cndmask b32 dependent pair (lo/hi) as used for selecting
    int lo = __double2loint(x), hi = __double2hiint(x);
    TIC(); REP64(asm volatile("v_cndmask_b32 %0, 0, %0, vcc\n v_cndmask_b32 %1, 0, %1, vcc" : "+v"(lo), "+v"(hi))); TOC(8);
    x += __hiloint2double(hi, lo);
    // readlane -> v_mov from sgpr -> (dependent)
    TIC(); REP64(asm volatile("v_readlane_b32 s20, %0, 5\n s_nop 0\n v_mov_b32 %0, s20" : "+v"(lo) :: "s20")); TOC(9);
    // readfirstlane round trip
    TIC(); REP64(asm volatile("v_readfirstlane_b32 s20, %0\n v_add_u32 %0, s20, %0" : "+v"(lo) :: "s20")); TOC(10);
    // DPP row_shr move dependent chain (b32)
    TIC(); REP64(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(lo))); TOC(11);
    // ds_bpermute dependent chain
    int addr = (threadIdx.x * 4 + 4) & 255;
    TIC(); REP64(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(lo) : "v"(addr))); TOC(12);
    // v_permlane32_swap dependent
    TIC(); REP64(asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi))); TOC(13);
    x += lo + hi;
    // IEEE division / sqrt as the compiler emits them (data-dependent chain through volatile asm identity)
    double q = x;
    TIC();
#pragma unroll
    for (int i = 0; i < 16; i++) { q = 1.0 / q + 1.0; asm volatile("" : "+v"(q)); }
    TOC(14);
    TIC();
#pragma unroll
    for (int i = 0; i < 16; i++) { q = sqrt(q) + 1.0; asm volatile("" : "+v"(q)); }
    TOC(15);
    // rsq + 2 newton
    TIC();
#pragma unroll
    for (int i = 0; i < 16; i++) {
      double r = __builtin_amdgcn_rsq(q);
      const double h = 0.5 * q;
      r = r * fma(-h * r, r, 1.5);
      r = r * fma(-h * r, r, 1.5);
      q = r + 2.0;
      asm volatile("" : "+v"(q));
    }
    TOC(16);
    // log (Cauchy loss) chain
    TIC();
#pragma unroll
    for (int i = 0; i < 8; i++) { q = log(q + 1.5) + 2.0; asm volatile("" : "+v"(q)); }
    TOC(17);
    x += q;
    TIC(); REP64(asm volatile("s_nop 15")); TOC(18);
    TIC(); REP64(asm volatile("s_nop 0")); TOC(19);
    // rsq -> mul -> mul -> fma -> mul (one Newton step, as compiled) dependent chain
    TIC(); REP64(asm volatile("v_rsq_f64 %1, %0\n v_mul_f64 %0, %0, -0.5\n v_mul_f64 %0, %0, %1\n v_fma_f64 %0, %0, %1, %2\n v_mul_f64 %0, %1, %0\n v_add_f64 %0, %0, %2" : "+v"(x), "=&v"(y) : "v"(z))); TOC(20);
    // mfma -> (dependent VALU read of acc) with explicit s_nop padding as the compiler emits
    asm volatile("v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0" ::: "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
    a = x;
#define ACC_CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"
    TIC(); REP64(asm volatile("v_mfma_f64_16x16x4_f64 v[100:107], %0, %0, v[100:107]\n s_nop 15\n s_nop 2\n v_mul_f64 %0, v[100:101], %1" : "+v"(a) : "v"(z) : ACC_CLOB)); TOC(21);
    TIC(); REP64(asm volatile("v_mfma_f64_16x16x4_f64 v[100:107], %0, %0, v[100:107]\n v_readlane_b32 s20, %2, 5\n s_nop 15\n s_nop 1\n v_mul_f64 %0, v[100:101], %1" : "+v"(a) : "v"(z), "v"(lo) : ACC_CLOB, "s20")); TOC(22);
    TIC(); REP64(asm volatile("v_mfma_f64_16x16x4_f64 v[100:107], %0, %0, v[100:107]\n s_nop 15\n v_mul_f64 %0, v[100:101], %1" : "+v"(a) : "v"(z) : ACC_CLOB)); TOC(23);
    TIC(); REP64(asm volatile("v_mfma_f64_16x16x4_f64 v[100:107], %0, %0, v[100:107]\n s_nop 7\n v_mul_f64 %0, v[100:101], %1" : "+v"(a) : "v"(z) : ACC_CLOB)); TOC(24);
    TIC(); REP64(asm volatile("v_mfma_f64_16x16x4_f64 v[100:107], %0, %0, v[100:107]\n v_mul_f64 %0, v[100:101], %1" : "+v"(a) : "v"(z) : ACC_CLOB)); TOC(25);
    asm volatile("v_mov_b32 v110, 0\n v_mov_b32 v111, 0x3ff00000\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0xbff00000\n v_mov_b32 v100, 0\n v_mov_b32 v101, 0x40100000\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3fe00000\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0x40100000" ::: "v100","v101","v102","v103","v104","v105","v106","v107","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","s20","s21","s22","s23");
    TIC(); REP64(asm volatile("v_readlane_b32 s20, v110, 17\n v_readlane_b32 s21, v111, 17\n v_readlane_b32 s22, v100, 34\n v_readlane_b32 s23, v101, 34\n"
      "v_mfma_f64_16x16x4_f64 v[100:107], v[112:113], v[110:111], v[100:107]\n"
      "v_mov_b32 v114, s22\n v_mov_b32 v115, s23\n v_fma_f64 v[114:115], -s[20:21], s[20:21], v[114:115]\n"
      "v_rsq_f64 v[116:117], v[114:115]\n v_mul_f64 v[114:115], v[114:115], -0.5\n v_mul_f64 v[114:115], v[114:115], v[116:117]\n"
      "v_fma_f64 v[114:115], v[114:115], v[116:117], %0\n v_mul_f64 v[116:117], v[116:117], v[114:115]\n"
      "v_mul_f64 v[118:119], v[100:101], v[116:117]\n"
      "v_cndmask_b32 v111, 0, v119, vcc\n v_cndmask_b32 v110, 0, v118, vcc\n v_xor_b32 v113, 0x80000000, v111\n v_mov_b32 v112, v110\n" :: "v"(z) : "v100","v101","v102","v103","v104","v105","v106","v107","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","s20","s21","s22","s23")); TOC(26);
    TIC(); REP64(asm volatile("v_mfma_f64_16x16x4_f64 v[100:107], v[112:113], v[110:111], v[100:107]\n"
      "v_mov_b32 v114, s22\n v_mov_b32 v115, s23\n v_fma_f64 v[114:115], -s[20:21], s[20:21], v[114:115]\n"
      "v_rsq_f64 v[116:117], v[114:115]\n v_mul_f64 v[114:115], v[114:115], -0.5\n v_mul_f64 v[114:115], v[114:115], v[116:117]\n"
      "v_fma_f64 v[114:115], v[114:115], v[116:117], %0\n v_mul_f64 v[116:117], v[116:117], v[114:115]\n"
      "v_mul_f64 v[118:119], v[100:101], v[116:117]\n"
      "v_cndmask_b32 v111, 0, v119, vcc\n v_cndmask_b32 v110, 0, v118, vcc\n v_xor_b32 v113, 0x80000000, v111\n v_mov_b32 v112, v110\n" :: "v"(z) : "v100","v101","v102","v103","v104","v105","v106","v107","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","s20","s21","s22","s23")); TOC(27);
    TIC(); REP64(asm volatile("v_readlane_b32 s20, v110, 17\n v_readlane_b32 s21, v111, 17\n v_readlane_b32 s22, v100, 34\n v_readlane_b32 s23, v101, 34\n"
      "v_mov_b32 v114, s22\n v_mov_b32 v115, s23\n v_fma_f64 v[114:115], -s[20:21], s[20:21], v[114:115]\n"
      "v_rsq_f64 v[116:117], v[114:115]\n v_mul_f64 v[114:115], v[114:115], -0.5\n v_mul_f64 v[114:115], v[114:115], v[116:117]\n"
      "v_fma_f64 v[114:115], v[114:115], v[116:117], %0\n v_mul_f64 v[116:117], v[116:117], v[114:115]\n"
      "v_mul_f64 v[118:119], v[100:101], v[116:117]\n"
      "v_cndmask_b32 v111, 0, v119, vcc\n v_cndmask_b32 v110, 0, v118, vcc\n v_xor_b32 v113, 0x80000000, v111\n v_mov_b32 v112, v110\n" :: "v"(z) : "v100","v101","v102","v103","v104","v105","v106","v107","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","s20","s21","s22","s23")); TOC(28);
    TIC(); REP64(asm volatile("v_readfirstlane_b32 s20, v110\n v_readfirstlane_b32 s21, v111\n v_readfirstlane_b32 s22, v100\n v_readfirstlane_b32 s23, v101\n"
      "v_mfma_f64_16x16x4_f64 v[100:107], v[112:113], v[110:111], v[100:107]\n"
      "v_mov_b32 v114, s22\n v_mov_b32 v115, s23\n v_fma_f64 v[114:115], -s[20:21], s[20:21], v[114:115]\n"
      "v_rsq_f64 v[116:117], v[114:115]\n v_mul_f64 v[114:115], v[114:115], -0.5\n v_mul_f64 v[114:115], v[114:115], v[116:117]\n"
      "v_fma_f64 v[114:115], v[114:115], v[116:117], %0\n v_mul_f64 v[116:117], v[116:117], v[114:115]\n"
      "v_mul_f64 v[118:119], v[100:101], v[116:117]\n"
      "v_cndmask_b32 v111, 0, v119, vcc\n v_cndmask_b32 v110, 0, v118, vcc\n v_xor_b32 v113, 0x80000000, v111\n v_mov_b32 v112, v110\n" :: "v"(z) : "v100","v101","v102","v103","v104","v105","v106","v107","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","s20","s21","s22","s23")); TOC(29);
    x += a;
  }
  __syncthreads();
  // wave 0: dependent v_fma_f64 chain x256 timed; wave 4 (same SIMD) / wave 1 (other SIMD): back-to-back f64 MFMAs
  for (int cfg = 0; cfg < 4; cfg++) {
    __syncthreads();
    if (wave == 0) {
      TIC(); REP64(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))); TOC(30 + cfg);
    } else if ((cfg == 1 && wave == 4) || (cfg == 2 && wave == 1)) {
      REP64(asm volatile("v_mfma_f64_16x16x4_f64 v[100:107], %0, %0, v[100:107]" :: "v"(y) : "v100","v101","v102","v103","v104","v105","v106","v107"));
    } else if (cfg == 3 && wave == 4) {
      REP64(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0"   : "+v"(xf)));
    }
  }
  // same wave: mfma followed by 6 independent f64 VALU ops, then the dependent read
  if (wave == 0) {
    double p = x, q2 = y;
    TIC(); REP64(asm volatile("v_mfma_f64_16x16x4_f64 v[100:107], %2, %2, v[100:107]\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_mul_f64 %1, v[100:101], %3" : "+v"(p), "+v"(q2) : "v"(y), "v"(z) : "v100","v101","v102","v103","v104","v105","v106","v107")); TOC(34);
    TIC(); REP64(asm volatile("v_mfma_f64_16x16x4_f64 v[100:107], %2, %2, v[100:107]\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_mul_f64 %1, v[100:101], %3" : "+v"(xf), "+v"(q2) : "v"(y), "v"(z) : "v100","v101","v102","v103","v104","v105","v106","v107")); TOC(35);
    x += p + q2;
  }
  out[threadIdx.x] = x;
}

int main() {
  double *out;
  long long *cyc;
  hipMalloc(&out, 512 * 8), hipMalloc(&cyc, 64 * 8);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, out, cyc);
    hipDeviceSynchronize();
  }
  long long c[64];
  hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  const char *names[] = {"dep v_fma_f64", "dep v_mul_f64", "dep v_add_f64", "4 indep v_fma_f64 (per group of 4)", "dep v_rsq_f64", "dep v_rcp_f64", "dep v_sqrt_f64",
                         "dep v_rsq_f32", "dep v_cndmask pair", "readlane -> v_mov", "readfirstlane -> v_add", "dpp row_shr mov", "ds_bpermute", "v_permlane32_swap",
                         "IEEE 1/x + 1 (x16)", "IEEE sqrt + 1 (x16)", "rsq + 2 newton + add (x16)", "log + add (x8)", "s_nop 15", "s_nop 0", "rsq,mul,mul,fma,mul,add chain", "mfma; s_nop 19; v_mul(acc)", "mfma; readlane; s_nop 18; v_mul(acc)", "mfma; s_nop 16; v_mul(acc)", "mfma; s_nop 8; v_mul(acc)", "mfma; v_mul(acc) (no nop)", "pivot step (as compiled, no s_nop)", "pivot step without readlanes", "pivot step without mfma", "pivot step with readfirstlane", "wave0 4 dep fma f64 (alone)", "  .. while wave 4 (same SIMD) issues f64 mfma", "  .. while wave 1 (other SIMD) issues f64 mfma", "  .. while wave 4 issues f32 fma", "mfma; 6 indep dep-chain f64 fma; v_mul(acc)", "mfma; 6 f32 fma; v_mul(acc)"};
  const int cnt[] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 16, 16, 16, 8, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};
  for (int i = 0; i < 36; i++) printf("%-40s %8lld cycles  %7.1f / op\n", names[i], c[i], (double)c[i] / cnt[i]);
  return 0;
}
