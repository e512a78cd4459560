"""Host-side mirror of the reference's front-end entry point for batches of independent sequences.

`FeatureTracker.read_images(frames, publish)` is FeatureTracker::readImage (VINS_ios/feature_tracker.cpp:162-310) for
one frame of every sequence in one set of device launches; it returns the image_msg of each sequence (id, x, y, z)
when `publish` (the caller's img_cnt == 0, ViewController.mm:467,494). Tracker state (cur/pre/forw points, ids,
track_cnt, n_id) lives on the device between calls, like the reference object's fields.

All compute happens in csrc/libvio_amd.so (HIP, gfx950). There is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import abi

_u8p, _fp, _ip = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int32)


_OBS_DTYPE = np.dtype({"names": ["id", "x", "y", "z"], "formats": ["<i4", "<f8", "<f8", "<f8"],
                       "offsets": [0, 8, 16, 24], "itemsize": 32})
assert C.sizeof(abi.VioObs) == 32


class FeatureTracker:
    def __init__(self, cfg=None, n_seq=1):
        self.lib = abi.load_product()
        self.cfg = cfg if cfg is not None else abi.default_config()
        self.n_seq = n_seq
        self._h = C.c_void_p()
        rc = self.lib.vio_frontend_create(C.byref(self.cfg), n_seq, C.byref(self._h))
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_frontend_create failed rc=%d (a gfx950 device is required)" % rc)
        self._frames = None

    def close(self):
        if self._h:
            self.lib.vio_frontend_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _check(rc, what):
        if rc != abi.VIO_OK:
            raise RuntimeError("%s failed rc=%d" % (what, rc))

    def device(self):
        """HIP device the tracker's contexts live on (vio_frontend_get_device)."""
        d = C.c_int32(-1)
        self._check(self.lib.vio_frontend_get_device(self._h, C.byref(d)), "get_device")
        return d.value

    def read_images(self, frames, publish):
        """frames: uint8 [n_seq, rows, cols]. Returns a list (per sequence) of (ids int32[n], xyz float64[n,3])."""
        frames = np.ascontiguousarray(frames, np.uint8)
        assert frames.shape == (self.n_seq, self.cfg.image_rows, self.cfg.image_cols), frames.shape
        cap = self.cfg.max_corners
        obs = np.zeros(self.n_seq * cap, _OBS_DTYPE)      # VioObs records, read back without per-element ctypes access
        n_obs = np.zeros(self.n_seq, np.int32)
        self._check(self.lib.vio_frontend_read_images(self._h, frames.ctypes.data_as(_u8p), frames.shape[1], frames.shape[2],
                                                      frames.shape[2], None, 1 if publish else 0,
                                                      C.cast(obs.ctypes.data, C.POINTER(abi.VioObs)), n_obs.ctypes.data_as(_ip)),
                    "read_images")
        obs = obs.reshape(self.n_seq, cap)
        out = []
        for s in range(self.n_seq):
            o = obs[s, :int(n_obs[s])]
            out.append((o["id"].copy(), np.stack([o["x"], o["y"], o["z"]], axis=1)))
        return out

    def register_host(self, array):
        """vio_host_register on a numpy buffer of frames: read_images / submit of frames inside it skip the gathering pass."""
        assert array.flags["C_CONTIGUOUS"]
        self._check(self.lib.vio_host_register(C.c_void_p(array.ctypes.data), array.nbytes), "host_register")

    def unregister_host(self, array):
        self._check(self.lib.vio_host_unregister(C.c_void_p(array.ctypes.data)), "host_unregister")

    def submit(self, frames, publish, asynchronous=False):
        """vio_frontend_submit_images (asynchronous: ..._async, the submit's host work on the context's own thread; the frame
        buffer is kept alive here until collect)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        assert frames.shape == (self.n_seq, self.cfg.image_rows, self.cfg.image_cols), frames.shape
        self._pending_frames = frames
        fn = self.lib.vio_frontend_submit_images_async if asynchronous else self.lib.vio_frontend_submit_images
        rc = fn(self._h, frames.ctypes.data_as(_u8p), frames.shape[1], frames.shape[2], frames.shape[2], 1 if publish else 0)
        if rc != abi.VIO_OK:
            self._pending_frames = None
        self._check(rc, "submit_images")

    def collect(self):
        """vio_frontend_collect -> the list read_images returns."""
        cap = self.cfg.max_corners
        obs = np.zeros(self.n_seq * cap, _OBS_DTYPE)
        n_obs = np.zeros(self.n_seq, np.int32)
        rc = self.lib.vio_frontend_collect(self._h, C.cast(obs.ctypes.data, C.POINTER(abi.VioObs)), n_obs.ctypes.data_as(_ip))
        self._pending_frames = None
        self._check(rc, "collect")
        obs = obs.reshape(self.n_seq, cap)
        out = []
        for s in range(self.n_seq):
            o = obs[s, :int(n_obs[s])]
            out.append((o["id"].copy(), np.stack([o["x"], o["y"], o["z"]], axis=1)))
        return out

    def state(self, seq=0):
        cap = self.cfg.max_corners
        pts = np.zeros((cap, 2), np.float32)
        ids = np.zeros(cap, np.int32)
        cnt = np.zeros(cap, np.int32)
        n = C.c_int32()
        self._check(self.lib.vio_frontend_get_state(self._h, seq, pts.ctypes.data_as(_fp), ids.ctypes.data_as(_ip),
                                                    cnt.ctypes.data_as(_ip), cap, C.byref(n)), "get_state")
        return pts[: n.value].copy(), ids[: n.value].copy(), cnt[: n.value].copy()

    def pnp_points(self, seq=0):
        """forw_pts / ids where solveVinsPnP joins them (feature_tracker.cpp:207): ahead of rejectWithF / setMask."""
        cap = self.cfg.max_corners
        pts = np.zeros((cap, 2), np.float32)
        ids = np.zeros(cap, np.int32)
        n = C.c_int32()
        self._check(self.lib.vio_frontend_get_pnp_points(self._h, seq, pts.ctypes.data_as(_fp), ids.ctypes.data_as(_ip), cap,
                                                         C.byref(n)), "get_pnp_points")
        return pts[: n.value].copy(), ids[: n.value].copy()

    def lk_iterations(self, enable=None, read=True):
        """vio_frontend_lk_iterations: switches the LK kernel's iteration counters (enable True / False / None = leave) and,
        with read, returns (iterations, visits) per pyramid level since the last read."""
        it = (C.c_uint64 * 8)()
        vis = (C.c_uint64 * 8)()
        self._check(self.lib.vio_frontend_lk_iterations(self._h, -1 if enable is None else (1 if enable else 0),
                                                        it if read else None, vis if read else None, 8), "lk_iterations")
        return (np.array(it[:], np.float64), np.array(vis[:], np.float64)) if read else None

    # resident API (throughput runs): frames [n_frames, n_seq, rows, cols] uploaded once
    def upload_frames(self, frames):
        frames = np.ascontiguousarray(frames, np.uint8)
        assert frames.ndim == 4 and frames.shape[1] == self.n_seq
        self._check(self.lib.vio_frontend_upload_frames(self._h, frames.ctypes.data_as(_u8p), frames.shape[0], frames.shape[2],
                                                        frames.shape[3], frames.shape[3]), "upload_frames")

    def step(self, frame_index, publish, stream=None):
        self._check(self.lib.vio_frontend_step_resident(self._h, frame_index, 1 if publish else 0,
                                                        C.c_void_p(stream) if stream else None), "step_resident")

    def sync(self):
        self._check(self.lib.vio_frontend_sync(self._h), "sync")

    def kernel_ms(self):
        ms = C.c_double()
        n = C.c_int32()
        self._check(self.lib.vio_frontend_kernel_ms(self._h, C.byref(ms), C.byref(n)), "kernel_ms")
        return ms.value, n.value


def klt_track(cfg, prev, nxt, pts):
    """cv::calcOpticalFlowPyrLK(prev, next, pts, ..., Size(21,21), 3) (feature_tracker.cpp:181)."""
    lib = abi.load_product()
    prev, nxt = np.ascontiguousarray(prev, np.uint8), np.ascontiguousarray(nxt, np.uint8)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = len(pts)
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    rc = lib.vio_klt_track(C.byref(cfg), prev.ctypes.data_as(_u8p), nxt.ctypes.data_as(_u8p), prev.shape[0], prev.shape[1],
                           prev.shape[1], pts.ctypes.data_as(_fp), n, out.ctypes.data_as(_fp), st.ctypes.data_as(_u8p),
                           err.ctypes.data_as(_fp))
    if rc != abi.VIO_OK:
        raise RuntimeError("vio_klt_track rc=%d" % rc)
    return out, st, err


def good_features(cfg, img, mask, max_corners):
    """cv::goodFeaturesToTrack(img, pts, max_corners, 0.01, MIN_DIST, mask) (feature_tracker.cpp:263)."""
    lib = abi.load_product()
    img = np.ascontiguousarray(img, np.uint8)
    corners = np.zeros((max_corners, 2), np.float32)
    n = C.c_int32()
    mp = np.ascontiguousarray(mask, np.uint8).ctypes.data_as(_u8p) if mask is not None else None
    rc = lib.vio_good_features(C.byref(cfg), img.ctypes.data_as(_u8p), mp, img.shape[0], img.shape[1], img.shape[1],
                               max_corners, corners.ctypes.data_as(_fp), C.byref(n))
    if rc != abi.VIO_OK:
        raise RuntimeError("vio_good_features rc=%d" % rc)
    return corners[: n.value].copy()


def fundamental_ransac(cfg, p1, p2):
    """status of cv::findFundamentalMat(p1, p2, FM_RANSAC, F_THRESHOLD, 0.99, status) (feature_tracker.cpp:95,198)."""
    lib = abi.load_product()
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    m = np.zeros(len(p1), np.uint8)
    rc = lib.vio_fundamental_ransac(C.byref(cfg), p1.ctypes.data_as(_fp), p2.ctypes.data_as(_fp), len(p1), m.ctypes.data_as(_u8p))
    if rc != abi.VIO_OK:
        raise RuntimeError("vio_fundamental_ransac rc=%d" % rc)
    return m


class Preprocessor:
    """The image pre-step of the camera callback on the device (cvtColor RGBA2GRAY + CLAHE, ViewController.mm:432-437):
    ctypes wrapper over vio_preprocess_*."""

    def __init__(self, rows, cols, max_frames=1, lib=None):
        self.lib = lib or abi.load_product()
        self.rows, self.cols, self.max_frames = rows, cols, max_frames
        self._h = C.c_void_p()
        rc = self.lib.vio_preprocess_create(max_frames, rows, cols, C.byref(self._h))
        if rc != 0:
            raise RuntimeError("vio_preprocess_create failed: %d" % rc)

    def close(self):
        if self._h:
            self.lib.vio_preprocess_destroy(self._h)
            self._h = C.c_void_p()

    def set_clahe(self, clip_limit=3.0, tiles_x=8, tiles_y=8):
        rc = self.lib.vio_preprocess_set_clahe(self._h, float(clip_limit), tiles_x, tiles_y)
        if rc != 0:
            raise RuntimeError("vio_preprocess_set_clahe failed: %d" % rc)

    def run(self, frames):
        """frames: [n, rows, cols] gray or [n, rows, cols, 4] RGBA (uint8) -> (gray [n, rows, cols], equalized)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        ch = 4 if frames.ndim == 4 else 1
        n = frames.shape[0]
        assert frames.shape[1:3] == (self.rows, self.cols)
        gray, eq = np.zeros((n, self.rows, self.cols), np.uint8), np.zeros((n, self.rows, self.cols), np.uint8)
        u8p = C.POINTER(C.c_uint8)
        rc = self.lib.vio_preprocess_run(self._h, frames.ctypes.data_as(u8p), ch, n, self.cols * ch, gray.ctypes.data_as(u8p),
                                         eq.ctypes.data_as(u8p))
        if rc != 0:
            raise RuntimeError("vio_preprocess_run failed: %d" % rc)
        return gray, eq

    def run_resident(self, d_pixels, channels, n_frames, d_equalized, stream=None):
        rc = self.lib.vio_preprocess_run_resident(self._h, C.c_void_p(d_pixels), channels, n_frames, self.cols * channels,
                                                  C.c_void_p(d_equalized), C.c_void_p(stream or 0))
        if rc != 0:
            raise RuntimeError("vio_preprocess_run_resident failed: %d" % rc)

    def sync(self):
        rc = self.lib.vio_preprocess_sync(self._h)
        if rc != 0:
            raise RuntimeError("vio_preprocess_sync failed: %d" % rc)

    def kernel_ms(self):
        ms, n = C.c_double(), C.c_int32()
        self.lib.vio_preprocess_kernel_ms(self._h, C.byref(ms), C.byref(n))
        return ms.value, n.value
