"""Host-side mirror of the window-bookkeeping interface (FeatureManager, VINS_ios/feature_manager.hpp:71-103): thin
ctypes wrappers over vio_features_* for tests and examples. No logic lives here."""
import ctypes as C

import numpy as np

from . import abi

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _d(a):
    return np.ascontiguousarray(a, np.float64)


class FeatureManager:
    def __init__(self, window_size, lib=None, prefix="vio"):
        self.lib = lib or abi.load_product()
        self.W = window_size
        self._h = C.c_void_p()
        self._check(self.lib.vio_features_create(window_size, C.byref(self._h)), "create")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("vio_features_%s failed: %d" % (what, rc))

    def close(self):
        if self._h:
            self.lib.vio_features_destroy(self._h)
            self._h = C.c_void_p()

    def clear(self):
        self._check(self.lib.vio_features_clear(self._h), "clear")

    def add_check_parallax(self, frame_count, ids, xyz):
        """image_msg of one frame -> (enough_parallax, parallax_num, last_track_num)."""
        n = len(ids)
        obs = (abi.VioObs * max(n, 1))()
        for i in range(n):
            obs[i].id, obs[i].x, obs[i].y, obs[i].z = int(ids[i]), float(xyz[i][0]), float(xyz[i][1]), float(xyz[i][2])
        e, p, t = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.vio_features_add_check_parallax(self._h, frame_count, obs, n, C.byref(e), C.byref(p), C.byref(t)),
                    "add_check_parallax")
        return bool(e.value), p.value, t.value

    def count(self):
        n = C.c_int32()
        self._check(self.lib.vio_features_count(self._h, C.byref(n)), "count")
        return n.value

    def get_depth_vector(self):
        cap = self.count()
        out, n = np.zeros(max(cap, 1)), C.c_int32()
        self._check(self.lib.vio_features_get_depth_vector(self._h, out.ctypes.data_as(_dp), cap, C.byref(n)), "get_depth_vector")
        return out[:n.value].copy()

    def set_depth(self, x):
        x = _d(x)
        self._check(self.lib.vio_features_set_depth(self._h, x.ctypes.data_as(_dp), len(x)), "set_depth")

    def clear_depth(self, x):
        x = _d(x)
        self._check(self.lib.vio_features_clear_depth(self._h, x.ctypes.data_as(_dp), len(x)), "clear_depth")

    def triangulate(self, Ps, Rs, tic, ric):
        Ps, Rs, tic, ric = _d(Ps).reshape(-1, 3), _d(Rs).reshape(-1, 9), _d(tic), _d(ric).reshape(9)
        assert len(Ps) == self.W + 1 and len(Rs) == self.W + 1
        self._check(self.lib.vio_features_triangulate(self._h, Ps.ctypes.data_as(_dp), Rs.ctypes.data_as(_dp),
                                                      tic.ctypes.data_as(_dp), ric.ctypes.data_as(_dp)), "triangulate")

    def remove_failures(self):
        self._check(self.lib.vio_features_remove_failures(self._h), "remove_failures")

    def remove_back(self):
        self._check(self.lib.vio_features_remove_back(self._h), "remove_back")

    def remove_back_shift_depth(self, marg_R, marg_P, new_R, new_P):
        a, b, c, d = _d(marg_R).reshape(9), _d(marg_P), _d(new_R).reshape(9), _d(new_P)
        self._check(self.lib.vio_features_remove_back_shift_depth(self._h, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp),
                                                                  c.ctypes.data_as(_dp), d.ctypes.data_as(_dp)),
                    "remove_back_shift_depth")

    def remove_front(self, frame_count):
        self._check(self.lib.vio_features_remove_front(self._h, frame_count), "remove_front")

    def export_factors(self, cap=20000):
        host, target, feat = (np.zeros(cap, np.int32) for _ in range(3))
        pi, pj = np.zeros((cap, 3)), np.zeros((cap, 3))
        m, nf = C.c_int32(), C.c_int32()
        self._check(self.lib.vio_features_export_factors(self._h, cap, host.ctypes.data_as(_ip), target.ctypes.data_as(_ip),
                                                         feat.ctypes.data_as(_ip), pi.ctypes.data_as(_dp),
                                                         pj.ctypes.data_as(_dp), C.byref(m), C.byref(nf)), "export_factors")
        k = m.value
        return host[:k].copy(), target[:k].copy(), feat[:k].copy(), pi[:k].copy(), pj[:k].copy(), nf.value

    def dump(self, cap=4096, cap_points=65536):
        info = (abi.VioFeatureInfo * cap)()
        pts = np.zeros((cap_points, 3))
        n, npts = C.c_int32(), C.c_int32()
        self._check(self.lib.vio_features_dump(self._h, info, cap, C.byref(n), pts.ctypes.data_as(_dp), cap_points,
                                               C.byref(npts)), "dump")
        rec = np.array([(f.id, f.start_frame, f.n_obs, f.used_num, f.solve_flag, f.is_outlier, f.fixed, f.estimated_depth)
                        for f in info[:n.value]], dtype=np.float64).reshape(-1, 8)
        return rec, pts[:npts.value].copy()


def failure_detection(last_track_num, Bg, P, R, last_P, last_R, lib=None):
    """failureDetection (VINS.cpp:214-265) -> bit mask of VIO_FAIL_* reasons (0 = healthy)."""
    lib = lib or abi.load_product()
    a = [_d(x).ravel() for x in (Bg, P, R, last_P, last_R)]
    out = C.c_int32()
    rc = lib.vio_failure_detection(int(last_track_num), *[x.ctypes.data_as(_dp) for x in a], C.byref(out))
    if rc != 0:
        raise RuntimeError("vio_failure_detection failed: %d" % rc)
    return out.value
