"""Host-side mirror of the replay I/O interface (record / playback formats and getMeasurements of the reference's app,
VINS_ios/ViewController.mm:603-682,1120-1150,1614-1708): thin ctypes wrappers over vio_replay_* / vio_measurements_*."""
import ctypes as C
import os

import numpy as np

from . import abi

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %d" % (what, rc))


def _lib(lib):
    return lib or abi.load_product()


def write_imu(path, headers, acc, gyr, lib=None):
    n = len(headers)
    msgs = (abi.VioImuMsg * max(n, 1))()
    for i in range(n):
        msgs[i].header = float(headers[i])
        msgs[i].acc[:] = [float(v) for v in acc[i]]
        msgs[i].gyr[:] = [float(v) for v in gyr[i]]
    _check(_lib(lib).vio_replay_write_imu(os.fsencode(path), msgs, n), "vio_replay_write_imu")


def read_imu(path, lib=None):
    lib = _lib(lib)
    n = C.c_int32()
    _check(lib.vio_replay_read_imu(os.fsencode(path), None, 0, C.byref(n)), "vio_replay_read_imu")
    msgs = (abi.VioImuMsg * max(n.value, 1))()
    _check(lib.vio_replay_read_imu(os.fsencode(path), msgs, n.value, C.byref(n)), "vio_replay_read_imu")
    a = np.array([[m.header, *m.acc, *m.gyr] for m in msgs[:n.value]]).reshape(-1, 7)
    return a[:, 0].copy(), a[:, 1:4].copy(), a[:, 4:7].copy()


def write_image_time(dirname, index, header, lib=None):
    _check(_lib(lib).vio_replay_write_image_time(os.fsencode(dirname), index, float(header)), "vio_replay_write_image_time")


def read_image_time(dirname, index, lib=None):
    h = C.c_double()
    rc = _lib(lib).vio_replay_read_image_time(os.fsencode(dirname), index, C.byref(h))
    return None if rc != 0 else h.value


def write_image(dirname, index, pixels, lib=None):
    pixels = np.ascontiguousarray(pixels, np.uint8)
    ch = 1 if pixels.ndim == 2 else pixels.shape[2]
    _check(_lib(lib).vio_replay_write_image(os.fsencode(dirname), index, pixels.ctypes.data_as(_u8p), pixels.shape[0],
                                            pixels.shape[1], ch), "vio_replay_write_image")


def read_image(dirname, index, lib=None):
    """IMAGE/<index> as the gray frame the tracker pipeline starts from, or None when the file does not exist."""
    lib = _lib(lib)
    r, c = C.c_int32(), C.c_int32()
    rc = lib.vio_replay_read_image(os.fsencode(dirname), index, None, 0, C.byref(r), C.byref(c))
    if rc != abi.VIO_ECAP:
        return None
    out = np.zeros((r.value, c.value), np.uint8)
    _check(lib.vio_replay_read_image(os.fsencode(dirname), index, out.ctypes.data_as(_u8p), out.size, C.byref(r), C.byref(c)),
           "vio_replay_read_image")
    return out


def decode_png_gray(data, lib=None):
    lib = _lib(lib)
    buf = np.frombuffer(data, np.uint8)
    r, c = C.c_int32(), C.c_int32()
    rc = lib.vio_replay_decode_png_gray(buf.ctypes.data_as(_u8p), buf.size, None, 0, C.byref(r), C.byref(c))
    if rc != abi.VIO_ECAP:
        raise ValueError("not a decodable PNG (%d)" % rc)
    out = np.zeros((r.value, c.value), np.uint8)
    _check(lib.vio_replay_decode_png_gray(buf.ctypes.data_as(_u8p), buf.size, out.ctypes.data_as(_u8p), out.size, C.byref(r),
                                          C.byref(c)), "vio_replay_decode_png_gray")
    return out


def rgba_to_gray(rgba, lib=None):
    rgba = np.ascontiguousarray(rgba, np.uint8)
    out = np.zeros(rgba.shape[:2], np.uint8)
    _check(_lib(lib).vio_replay_rgba_to_gray(rgba.ctypes.data_as(_u8p), rgba.shape[0], rgba.shape[1], rgba.shape[1] * 4,
                                             out.ctypes.data_as(_u8p)), "vio_replay_rgba_to_gray")
    return out


def write_keyframes(path, headers, translations, rotations_xyzw, lib=None):
    n = len(headers)
    kf = (abi.VioKeyframeData * max(n, 1))()
    for i in range(n):
        kf[i].header = float(headers[i])
        kf[i].translation[:] = [float(v) for v in translations[i]]
        kf[i].rotation[:] = [float(v) for v in rotations_xyzw[i]]
    _check(_lib(lib).vio_replay_write_keyframes(os.fsencode(path), kf, n), "vio_replay_write_keyframes")


def read_keyframes(path, lib=None):
    lib = _lib(lib)
    n = C.c_int32()
    _check(lib.vio_replay_read_keyframes(os.fsencode(path), None, 0, C.byref(n)), "vio_replay_read_keyframes")
    kf = (abi.VioKeyframeData * max(n.value, 1))()
    _check(lib.vio_replay_read_keyframes(os.fsencode(path), kf, n.value, C.byref(n)), "vio_replay_read_keyframes")
    a = np.array([[k.header, *k.translation, *k.rotation] for k in kf[:n.value]]).reshape(-1, 8)
    return a[:, 0].copy(), a[:, 1:4].copy(), a[:, 4:8].copy()


class Measurements:
    """imu_msg_buf / img_msg_buf + getMeasurements + the dt of send_imu."""

    def __init__(self, lib=None):
        self.lib = _lib(lib)
        self._h = C.c_void_p()
        _check(self.lib.vio_measurements_create(C.byref(self._h)), "vio_measurements_create")

    def close(self):
        if self._h:
            self.lib.vio_measurements_destroy(self._h)
            self._h = C.c_void_p()

    def push_imu(self, header, acc, gyr):
        m = abi.VioImuMsg()
        m.header = float(header)
        m.acc[:] = [float(v) for v in acc]
        m.gyr[:] = [float(v) for v in gyr]
        _check(self.lib.vio_measurements_push_imu(self._h, C.byref(m)), "vio_measurements_push_imu")

    def push_image(self, header, ids, xyz):
        n = len(ids)
        obs = (abi.VioObs * max(n, 1))()
        for i in range(n):
            obs[i].id, obs[i].x, obs[i].y, obs[i].z = int(ids[i]), float(xyz[i][0]), float(xyz[i][1]), float(xyz[i][2])
        _check(self.lib.vio_measurements_push_image(self._h, float(header), obs, n), "vio_measurements_push_image")

    def next(self, cap_imu=4096, cap_obs=1024):
        """-> None, or (imu [(header, dt, acc, gyr)], header, ids, xyz)."""
        imu = (abi.VioImuMsg * cap_imu)()
        dt = np.zeros(cap_imu)
        obs = (abi.VioObs * cap_obs)()
        ni, no, av, h = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
        _check(self.lib.vio_measurements_next(self._h, imu, dt.ctypes.data_as(_dp), cap_imu, C.byref(ni), C.byref(h), obs, cap_obs,
                                              C.byref(no), C.byref(av)), "vio_measurements_next")
        if not av.value:
            return None
        samples = [(imu[i].header, dt[i], np.array(imu[i].acc[:]), np.array(imu[i].gyr[:])) for i in range(ni.value)]
        return samples, h.value, [obs[i].id for i in range(no.value)], [[obs[i].x, obs[i].y, obs[i].z] for i in range(no.value)]
