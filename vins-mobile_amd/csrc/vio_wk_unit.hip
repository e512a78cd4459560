// vio_wk_unit.hip -- ONE instantiation of the window kernel: compiled once per (variant, stage clock) pair with
// -DVIO_WK_V=<variant> -DVIO_WK_P=<0|1> (csrc/Makefile; the table of variants is in vio_window_variants.h).
#include "vio_window_kernel.inc"

#ifndef VIO_WK_V
#error "compile with -DVIO_WK_V=<variant> -DVIO_WK_P=<0|1>"
#endif
#define VIO_WK_CAT2(a, b, c) a##b##_##c
#define VIO_WK_CAT(a, b, c) VIO_WK_CAT2(a, b, c)
#define VIO_WK_GETTER VIO_WK_CAT(vio_wk_variant_, VIO_WK_V, VIO_WK_P)

namespace {
constexpr vio_wk::VariantTraits T = vio_wk::kTraits[VIO_WK_V];
constexpr auto kernel = vio_wk::vio_window_kernel<T.lds_matrix, T.lds_asp, T.threads, VIO_WK_P != 0, T.ws>;
void launch(int grid, size_t lds_bytes, hipStream_t st, const vio::BatchPtrs &B, const vio::MargPtrs &MP) {
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(T.threads), lds_bytes, st, B, MP, (int)(lds_bytes / sizeof(double)));
}
}  // namespace

vio_wk::VariantFns VIO_WK_GETTER() { return vio_wk::VariantFns{(const void *)kernel, launch}; }
