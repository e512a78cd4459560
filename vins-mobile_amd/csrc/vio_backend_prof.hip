// vio_backend_prof.hip -- the window kernel with its stage clock compiled in (vio_backend_set_profile / vio_backend_stage_cycles:
// the kernel-side counterpart of the reference's TS / TE timers, global_param.hpp:85-92). A translation unit of its own: the
// product's launches take the instantiations without the clock (vio_backend.hip), and the two compile in parallel.
#include "vio_window_kernel.inc"

hipError_t vio_window_attrs_prof(int max_lds_bytes) { return vio_wk::window_kernel_attrs<true>(max_lds_bytes); }
void vio_window_launch_prof(int variant, int grid, size_t lds_bytes, hipStream_t st, const vio::BatchPtrs &B, const vio::MargPtrs &MP) {
  vio_wk::window_kernel_launch<true>(variant, grid, lds_bytes, st, B, MP);
}
