"""Host-side mirror of the reference's back-end entry point for batches of independent sequences.

`WindowSolver.solve(windows)` is VINS::solve_ceres (VINS_ios/VINS.cpp:480-831) for every window of the batch in one
device launch: trust-region solve, new2old gauge fix and marginalization; results are written back into the
`abi.Window` objects (pose / speed_bias / inv_depth / next_prior), like the reference writes back into
Ps/Rs/Vs/Bas/Bgs, f_manager depths and last_marginalization_info.

All compute happens in csrc/libvio_amd.so (HIP, gfx950). There is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import abi


class WindowSolver:
    def __init__(self, cfg=None, max_batch=1024):
        self.lib = abi.load_product()
        self.cfg = cfg if cfg is not None else abi.default_config()
        self.max_batch = max_batch
        self._h = C.c_void_p()
        rc = self.lib.vio_backend_create(C.byref(self.cfg), max_batch, C.byref(self._h))
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_backend_create failed rc=%d (a gfx950 device is required)" % rc)
        self._structs = None

    def close(self):
        if self._h:
            self.lib.vio_backend_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _array(self, windows):
        arr = (abi.VioWindow * len(windows))()
        for i, w in enumerate(windows):
            w.fill_struct(arr[i])
        return arr

    @staticmethod
    def _check(rc, what):
        if rc != abi.VIO_OK:
            raise RuntimeError("%s failed rc=%d" % (what, rc))

    def solve(self, windows, buf_num=0):
        """solve_ceres(buf_num) on every window; returns a list of per-window stats dicts."""
        arr = self._array(windows)
        stats = (abi.VioSolveStats * len(windows))()
        self._check(self.lib.vio_backend_solve_windows(self._h, arr, len(windows), buf_num, stats), "solve_windows")
        return [abi.stats_to_dict(s) for s in stats]

    def device(self):
        """HIP device ordinal the context is bound to (the device current on the creating thread)."""
        d = C.c_int32(-1)
        self._check(self.lib.vio_backend_get_device(self._h, C.byref(d)), "get_device")
        return d.value

    def reserve_priors(self, n_slots):
        """Device-resident prior chain: Window.resident_prior = k selects slot k-1 (vio_amd.h)."""
        self._check(self.lib.vio_backend_reserve_priors(self._h, n_slots), "reserve_priors")

    # resident-batch API (throughput runs)
    def upload(self, windows):
        self._structs = self._array(windows)
        self._check(self.lib.vio_backend_upload(self._h, self._structs, len(windows)), "upload")

    def launch(self, stream=None):
        self._check(self.lib.vio_backend_launch(self._h, C.c_void_p(stream) if stream else None), "launch")

    def sync(self):
        self._check(self.lib.vio_backend_sync(self._h), "sync")

    def download(self, windows):
        arr = self._array(windows)
        stats = (abi.VioSolveStats * len(windows))()
        self._check(self.lib.vio_backend_download(self._h, arr, len(windows), stats), "download")
        return [abi.stats_to_dict(s) for s in stats]

    def set_profile(self, enable=True):
        self._check(self.lib.vio_backend_set_profile(self._h, 1 if enable else 0), "set_profile")

    def stage_cycles(self, window=0):
        """Per-stage shader-clock cycles of one window's workgroup (see VIO_N_STAGES in include/vio_amd.h)."""
        out = (C.c_int64 * len(abi.STAGES))()
        self._check(self.lib.vio_backend_stage_cycles(self._h, window, out, len(abi.STAGES)), "stage_cycles")
        return dict(zip(abi.STAGES, [int(x) for x in out]))

    def kernel_ms(self):
        ms = C.c_double()
        n = C.c_int32()
        self._check(self.lib.vio_backend_kernel_ms(self._h, C.byref(ms), C.byref(n)), "kernel_ms")
        return ms.value, n.value


def preintegrate(cfg, acc0, gyr0, ba, bg, dt, acc, gyr):
    """IntegrationBase(acc0, gyr0, ba, bg) + push_back per sample (product host code, csrc/vio_host.cpp)."""
    return abi.preintegrate_with(abi.load_product().vio_preintegrate, cfg, acc0, gyr0, ba, bg, dt, acc, gyr)
