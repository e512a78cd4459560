"""Host-side mirror of the estimator interface (class VINS, VINS_ios/VINS.hpp:47-200): thin ctypes wrappers over
vio_estimator_* for tests, the replay tool and examples. No logic lives here."""
import ctypes as C

import numpy as np

from . import abi

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


class Estimator:
    """n_seq independent VINS objects whose window solves share one device launch."""

    def __init__(self, cfg, tic, ric, n_seq=1, lib=None):
        self.lib = lib or abi.load_product()
        self.cfg, self.n_seq, self.W = cfg, n_seq, cfg.window_size
        self._h = C.c_void_p()
        tic, ric = _d(tic), _d(ric)
        self._check(self.lib.vio_estimator_create(C.byref(cfg), n_seq, _p(tic), _p(ric), C.byref(self._h)), "create")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("vio_estimator_%s failed: %d" % (what, rc))

    def close(self):
        if self._h:
            self.lib.vio_estimator_destroy(self._h)
            self._h = C.c_void_p()

    def enable_initialization(self, on=True):
        """solveInitial inside process_image. on: False / 0 off; True / 2 / "fit": relativePose by the fit over all
        correspondences; 1 / "reference": by five-point RANSAC + recoverPose as the reference computes it (a lottery over the
        roots of one minimal sample, see include/vio_amd.h)."""
        # (dispatch on type: in Python 1 == True with the same hash, a dict lookup would send 1 to the fit)
        if isinstance(on, bool):
            mode = 2 if on else 0
        elif isinstance(on, str):
            mode = {"fit": 2, "reference": 1, "off": 0}[on]
        else:
            mode = int(on)
        self._check(self.lib.vio_estimator_enable_initialization(self._h, int(mode)), "enable_initialization")

    def set_resident(self, on=True):
        """Landmark lists of the sequences in the NON_LINEAR state on the device, windows assembled there
        (vio_estimator_set_resident; include/vio_amd.h)."""
        self._check(self.lib.vio_estimator_set_resident(self._h, 1 if on else 0), "set_resident")

    def clear(self, seq=0):
        self._check(self.lib.vio_estimator_clear(self._h, seq), "clear")

    def process_imu(self, dt, acc, gyr, seq=0):
        acc, gyr = _d(acc), _d(gyr)
        self._check(self.lib.vio_estimator_process_imu(self._h, seq, float(dt), _p(acc), _p(gyr)), "process_imu")

    def process_imu_batch(self, n_samples, dt, acc, gyr):
        """dt [n_seq, stride], acc / gyr [n_seq, stride, 3]; sequence q consumes its first n_samples[q] entries."""
        n = np.ascontiguousarray(n_samples, np.int32)
        dt, acc, gyr = _d(dt), _d(acc), _d(gyr)
        assert dt.shape[0] == self.n_seq and acc.shape == dt.shape + (3,) and gyr.shape == acc.shape
        self._check(self.lib.vio_estimator_process_imu_batch(self._h, n.ctypes.data_as(_ip), dt.shape[1], _p(dt), _p(acc),
                                                             _p(gyr)), "process_imu_batch")

    def set_initial_state(self, headers, Ps, Rs, Vs, Bas, Bgs, seq=0):
        a = [_d(x) for x in (headers, Ps, Rs, Vs, Bas, Bgs)]
        P = self.W + 1
        assert a[0].size == P and a[1].size == 3 * P and a[2].size == 9 * P
        self._check(self.lib.vio_estimator_set_initial_state(self._h, seq, *[_p(x) for x in a]), "set_initial_state")

    def set_relocalization(self, header, P_old, Q_old, ids, xy, seq=0):
        ids = np.ascontiguousarray(ids, np.int32)
        xy, P_old, Q_old = _d(xy), _d(P_old), _d(Q_old)
        self._check(self.lib.vio_estimator_set_relocalization(self._h, seq, float(header), _p(P_old), _p(Q_old),
                                                              ids.ctypes.data_as(_ip), _p(xy), len(ids)), "set_relocalization")

    @staticmethod
    def _pack_obs(ids, xyz, dst, off):
        for i in range(len(ids)):
            o = dst[off + i]
            o.id, o.x, o.y, o.z = int(ids[i]), float(xyz[i][0]), float(xyz[i][1]), float(xyz[i][2])

    def process_image(self, ids, xyz, header, seq=0):
        n = len(ids)
        obs = (abi.VioObs * max(n, 1))()
        self._pack_obs(ids, xyz, obs, 0)
        res = abi.VioFrameResult()
        self._check(self.lib.vio_estimator_process_image(self._h, seq, obs, n, float(header), C.byref(res)), "process_image")
        return res

    def process_images(self, obs_per_seq, headers, active=None, strict=False):
        """obs_per_seq: list of (ids, xyz) per sequence; one launch solves every sequence that has a full window.

        Error contract (vio_estimator_process_images): bad arguments raise. A failure INSIDE the processing of a sequence
        (capacity exceeded, a device error of its solve group) does not: that sequence's result carries
        action == VIO_FRAME_ERROR and its `error` code, the library has restarted it (clearState, like the reference after
        failureDetection), every other sequence was processed normally, and the first failing code is kept in
        `last_error`. A warning is emitted for every such frame; strict=True raises instead."""
        stride = max(1, max(len(o[0]) for o in obs_per_seq))
        obs = (abi.VioObs * (stride * self.n_seq))()
        n = np.zeros(self.n_seq, np.int32)
        for q, (ids, xyz) in enumerate(obs_per_seq):
            self._pack_obs(ids, xyz, obs, q * stride)
            n[q] = len(ids)
        headers = _d(headers)
        res = (abi.VioFrameResult * self.n_seq)()
        act = None
        if active is not None:
            act = np.ascontiguousarray(active, np.uint8).ctypes.data_as(C.POINTER(C.c_uint8))
        rc = self.lib.vio_estimator_process_images(self._h, obs, n.ctypes.data_as(_ip), stride, _p(headers), act, res)
        out = list(res)
        # A non-zero code with per-sequence VIO_FRAME_ERROR results is the FIRST failing sequence's code: every other
        # sequence was processed, solved and slid, so the results are returned (state and caller stay in step) and the code
        # is kept in last_error. Only a failure before any processing (bad arguments) raises.
        self.last_error = rc
        failed = [q for q, r in enumerate(out) if r.action == abi.VIO_FRAME_ERROR]
        if rc != 0 and not failed:
            self._check(rc, "process_images")
        if failed:
            msg = "vio_estimator_process_images: rc=%d, sequences %s failed and were restarted" % (rc, failed[:8])
            if strict:
                raise RuntimeError(msg)
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
        return out

    def status(self, seq=0):
        st = abi.VioEstimatorStatus()
        self._check(self.lib.vio_estimator_get_status(self._h, seq, C.byref(st)), "get_status")
        return st

    def window(self, seq=0):
        P = self.W + 1
        Ps, Rs, Vs, Bas, Bgs, hdr = (np.zeros((P, 3)), np.zeros((P, 3, 3)), np.zeros((P, 3)), np.zeros((P, 3)),
                                     np.zeros((P, 3)), np.zeros(P))
        self._check(self.lib.vio_estimator_get_window(self._h, seq, _p(Ps), _p(Rs), _p(Vs), _p(Bas), _p(Bgs), _p(hdr)),
                    "get_window")
        return dict(Ps=Ps, Rs=Rs, Vs=Vs, Bas=Bas, Bgs=Bgs, headers=hdr)

    def corrected_window(self, seq=0):
        P = self.W + 1
        Ps, Rs = np.zeros((P, 3)), np.zeros((P, 3, 3))
        self._check(self.lib.vio_estimator_get_corrected_window(self._h, seq, _p(Ps), _p(Rs)), "get_corrected_window")
        return Ps, Rs

    def features(self, seq=0):
        """A non-owning FeatureManager view of the sequence's landmark store."""
        from . import window
        h = C.c_void_p()
        self._check(self.lib.vio_estimator_features(self._h, seq, C.byref(h)), "features")
        fm = window.FeatureManager.__new__(window.FeatureManager)
        fm.lib, fm.W, fm._h = self.lib, self.W, h
        fm.close = lambda: None
        return fm
