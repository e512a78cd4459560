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


class PnpTracker:
    """class vinsPnP for n sequences (vio_pnp_tracker_*)."""

    def __init__(self, cfg, tic, ric, n_seq=1, pnp_size=6, lib=None):
        self.lib = lib or abi.load_product()
        self.cfg, self.n_seq, self.size = cfg, n_seq, pnp_size
        self._h = C.c_void_p()
        tic, ric = np.ascontiguousarray(tic, np.float64), np.ascontiguousarray(ric, np.float64)
        rc = self.lib.vio_pnp_tracker_create(C.byref(cfg), n_seq, pnp_size, tic.ctypes.data_as(_dp), ric.ctypes.data_as(_dp),
                                             C.byref(self._h))
        if rc != 0:
            raise RuntimeError("vio_pnp_tracker_create failed: %d" % rc)

    def close(self):
        if self._h:
            self.lib.vio_pnp_tracker_destroy(self._h)
            self._h = C.c_void_p()

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("vio_pnp_tracker_%s failed: %d" % (what, rc))

    def set_init(self, header, Ba, Bg, P, R, V, seq=0):
        r = abi.VioVinsResult()
        r.header = float(header)
        r.Ba[:], r.Bg[:], r.P[:], r.V[:] = list(Ba), list(Bg), list(P), list(V)
        r.R[:] = list(np.asarray(R, np.float64).ravel())
        self._check(self.lib.vio_pnp_tracker_set_init(self._h, seq, C.byref(r)), "set_init")

    def process_imu(self, dt, acc, gyr, seq=0):
        a, g = np.ascontiguousarray(acc, np.float64), np.ascontiguousarray(gyr, np.float64)
        self._check(self.lib.vio_pnp_tracker_process_imu(self._h, seq, float(dt), a.ctypes.data_as(_dp), g.ctypes.data_as(_dp)), "process_imu")

    def process_images(self, features_per_seq, headers, use_pnp=True, active=None):
        """features_per_seq: per sequence a list of (id, (x, y), (X, Y, Z), track_num), ascending id.
        -> (P [n_seq, 3], R [n_seq, 3, 3], solved [n_seq])."""
        stride = max(1, max(len(f) for f in features_per_seq))
        arr = (abi.VioPnpFeature * (stride * self.n_seq))()
        n = np.zeros(self.n_seq, np.int32)
        for q, feats in enumerate(features_per_seq):
            n[q] = len(feats)
            for i, (fid, ob, pos, tn) in enumerate(feats):
                f = arr[q * stride + i]
                f.id, f.track_num = int(fid), int(tn)
                f.observation[:] = [float(ob[0]), float(ob[1])]
                f.position[:] = [float(v) for v in pos]
        hdr = np.ascontiguousarray(headers, np.float64)
        P, R, solved = np.zeros((self.n_seq, 3)), np.zeros((self.n_seq, 3, 3)), np.zeros(self.n_seq, np.int32)
        act = None if active is None else np.ascontiguousarray(active, np.uint8).ctypes.data_as(C.POINTER(C.c_uint8))
        self._check(self.lib.vio_pnp_tracker_process_images(self._h, arr, n.ctypes.data_as(_ip), stride, hdr.ctypes.data_as(_dp),
                                                            int(use_pnp), act, P.ctypes.data_as(_dp), R.ctypes.data_as(_dp),
                                                            solved.ctypes.data_as(_ip)), "process_images")
        return P, R, solved

    def window(self, seq=0):
        n = self.size + 1
        Ps, Rs, Vs, hdr, fs, fc = np.zeros((n, 3)), np.zeros((n, 3, 3)), np.zeros((n, 3)), np.zeros(n), np.zeros(n, np.uint8), C.c_int32()
        self._check(self.lib.vio_pnp_tracker_get_window(self._h, seq, Ps.ctypes.data_as(_dp), Rs.ctypes.data_as(_dp), Vs.ctypes.data_as(_dp),
                                                        hdr.ctypes.data_as(_dp), fs.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(fc)),
                    "get_window")
        return dict(Ps=Ps, Rs=Rs, Vs=Vs, headers=hdr, find_solved=fs, frame_count=fc.value)
