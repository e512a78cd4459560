"""Host-side mirror of the motion-only window interface (vinsPnP::solve_ceres, VINS_ios/vins_pnp.cpp:264-341): a numpy
container for one window and thin ctypes wrappers over vio_pnp_*. No logic lives here."""
import ctypes as C

import numpy as np

from . import abi

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class PnpWindow:
    def __init__(self, pose, speed, bias, fixed, ex_pose, preint, feat_start, observation, position, track_num):
        f = lambda a: np.ascontiguousarray(a, np.float64)
        self.pose, self.speed, self.bias, self.ex_pose = f(pose).copy(), f(speed).copy(), f(bias), f(ex_pose)
        self.fixed = np.ascontiguousarray(fixed, np.uint8)
        self.preint = f(preint)                      # [n-1, 467] packed VioPreintegration records
        self.feat_start = np.ascontiguousarray(feat_start, np.int32)
        self.observation, self.position = f(observation).reshape(-1, 2), f(position).reshape(-1, 3)
        self.track_num = np.ascontiguousarray(track_num, np.int32)
        self.n = len(self.pose)
        assert self.preint.shape == (self.n - 1, 467) and len(self.feat_start) == self.n + 1

    def copy(self):
        return PnpWindow(self.pose, self.speed, self.bias, self.fixed, self.ex_pose, self.preint, self.feat_start,
                         self.observation, self.position, self.track_num)

    def fill_struct(self, s):
        s.n_frames = self.n
        s.pose, s.speed, s.bias = (a.ctypes.data_as(_dp) for a in (self.pose, self.speed, self.bias))
        s.fixed = self.fixed.ctypes.data_as(C.POINTER(C.c_uint8))
        s.ex_pose = self.ex_pose.ctypes.data_as(_dp)
        s.preint = C.cast(self.preint.ctypes.data, C.POINTER(abi.VioPreintegration))
        s.feat_start = self.feat_start.ctypes.data_as(_ip)
        s.observation, s.position = self.observation.ctypes.data_as(_dp), self.position.ctypes.data_as(_dp)
        s.track_num = self.track_num.ctypes.data_as(_ip)


def stats_dict(st):
    n = st.iterations
    return dict(initial_cost=st.initial_cost, final_cost=st.final_cost, iterations=n, termination=st.termination,
                it_cost=np.array(st.it_cost[:n]), it_flags=np.array(st.it_flags[:n]), it_radius=np.array(st.it_radius[:n]),
                it_step_norm=np.array(st.it_step_norm[:n]), it_gradient_max_norm=np.array(st.it_gradient_max_norm[:n]))


def solve_with(fn, cfg, w):
    """fn(cfg*, VioPnpWindow*, VioSolveStats*) -> (solved copy, stats dict): for the reference / emulation entry points."""
    out = w.copy()
    s, st = abi.VioPnpWindow(), abi.VioSolveStats()
    out.fill_struct(s)
    rc = fn(C.byref(cfg), C.byref(s), C.byref(st))
    assert rc == 0, rc
    return out, stats_dict(st)


class PnpSolver:
    def __init__(self, cfg, max_batch=1, lib=None):
        self.lib = lib or abi.load_product()
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = self.lib.vio_pnp_create(C.byref(cfg), max_batch, C.byref(self._h))
        if rc != 0:
            raise RuntimeError("vio_pnp_create failed: %d" % rc)

    def close(self):
        if self._h:
            self.lib.vio_pnp_destroy(self._h)
            self._h = C.c_void_p()

    def solve(self, windows):
        """Solves the windows in place (pose, speed); returns one stats dict per window."""
        n = len(windows)
        arr, st = (abi.VioPnpWindow * n)(), (abi.VioSolveStats * n)()
        for a, w in zip(arr, windows):
            w.fill_struct(a)
        rc = self.lib.vio_pnp_solve_windows(self._h, arr, n, st)
        if rc != 0:
            raise RuntimeError("vio_pnp_solve_windows failed: %d" % rc)
        return [stats_dict(s) for s in st]

    def kernel_ms(self):
        ms, k = C.c_double(), C.c_int32()
        self.lib.vio_pnp_kernel_ms(self._h, C.byref(ms), C.byref(k))
        return ms.value, k.value
