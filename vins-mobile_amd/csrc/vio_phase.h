// vio_phase.h — host-side interface between the back-end context (vio_backend.hip) and the kernels of the phase path
// (vio_phase.hip, phase_core.h): its own translation unit, so that the launch-sequence kernels and the single-launch
// kernel compile independently.
#pragma once

#include <hip/hip_runtime.h>

#include "batch.h"

namespace vio {

struct MargPtrs {
  int *ints;        // [n][4 + 3 * kMaxPriorBlocks]
  double *x0;       // [n][9 * kMaxPriorBlocks]
  double *J;        // [n][Ncap * Ncap]
  double *r;        // [n][Ncap]
  double *scratch;  // [n][marg_scratch] (global matrix variant only)
  size_t s_ints, s_x0, s_J, s_r, s_scratch;
  long long *prof;  // [n][ST_COUNT] or null
  int prof_tid;     // work-item that keeps the stage clock (VIO_AMD_PROF_TID, default 0)
  int wrot;         // wave-role rotation: -1 = from the hardware wave slot (default), else forced (VIO_AMD_WAVE_ROT)
};

// LDS bytes of the setup and the linearization kernel for the layout d.
void phase_lds_need(const BatchDims &d, size_t *setup_bytes, size_t *lin_bytes);
// Raises the dynamic-LDS ceiling of the phase kernels (once per process is enough; cheap). VIO_OK / VIO_ENODEV.
int phase_prepare();
// setup, linearize, (step, linearize) x max_iter, step, finish for the n windows B.order names, asynchronously on st.
void phase_launch(const BatchPtrs &B, const MargPtrs &MP, int n, size_t lds_setup, size_t lds_lin, size_t lds_step, hipStream_t st);

}  // namespace vio
