// vio_window_variants.h -- the instantiations of the window kernel (vio_window_kernel.inc) and how the launcher reaches them.
// Every (variant, stage clock on / off) pair is a translation unit of its own (vio_wk_unit.hip compiled once per pair, csrc/Makefile):
// the kernel is ~70 k instructions per instantiation and the units build in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include <stddef.h>

namespace vio {
struct BatchPtrs;
struct MargPtrs;
}  // namespace vio

namespace vio_wk {

constexpr int kThreadsLds = 256, kThreadsGlb = 512;

// variant: pose matrix in LDS?, IMU coupling in LDS?, work-items, compile-time window size (0: run-time)
//   0  LDS, LDS, 256, 0     the fat layout, two windows per CU (F <= 160 at W = 10)
//   1  LDS, scratch, 256, 0 the lean layout (F <= 310)
//   2  LDS, LDS, 512, 0     experiment switch VIO_AMD_WINDOW_THREADS=512
//   3  scratch, scratch, 512, 0   W > 12 (cooperative windows)
//   4  = 0 with W = 10 at compile time (the reference's WINDOW_SIZE, global_param.hpp:28), no relocalization pose in the batch
//   5  = 1 with W = 10 at compile time
constexpr int kVariants = 6;
constexpr int kStaticW = 10;
struct VariantTraits {
  bool lds_matrix, lds_asp;
  int threads, ws;
};
constexpr VariantTraits kTraits[kVariants] = {{true, true, kThreadsLds, 0},  {true, false, kThreadsLds, 0},      {true, true, kThreadsGlb, 0},
                                              {false, false, kThreadsGlb, 0}, {true, true, kThreadsLds, kStaticW}, {true, false, kThreadsLds, kStaticW}};

struct VariantFns {
  const void *fn;  // the kernel (hipFuncSetAttribute, occupancy queries)
  void (*launch)(int grid, size_t lds_bytes, hipStream_t st, const vio::BatchPtrs &B, const vio::MargPtrs &MP);
};
// (vio_backend.hip) prof: the instantiation with the stage clock compiled in (vio_backend_set_profile)
const VariantFns &variant(int v, bool prof);

}  // namespace vio_wk
