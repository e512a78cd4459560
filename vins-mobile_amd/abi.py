"""ctypes mirror of include/vio_amd.h — the C ABI of the MI355X VIO hot path.

Only plumbing lives here: struct layouts, numpy <-> struct marshalling and a loader for the product
shared library (``csrc/libvio_amd.so``).  The same structs are understood by the test-only checkers
under ``oracle/`` (``libvio_oracle.so``, ``_ref/libvio_ref.so``), which is what lets tests hand
bit-identical inputs to the reference, the CPU restatement and the HIP path.

Reference interface mirrored: FeatureTracker::readImage (VINS_ios/feature_tracker.hpp:59) and
VINS::solve_ceres (VINS_ios/VINS.hpp:153).
"""
import ctypes as C
import os

import numpy as np

VIO_OK = 0
VIO_EINVAL, VIO_ENODEV, VIO_ENOMEM, VIO_ECAP, VIO_ESTATE, VIO_ETIMEOUT = -1, -2, -3, -4, -5, -6
VIO_MAX_PRIOR_BLOCKS = 96
VIO_MAX_TRACE = 64
VIO_BLOCK_POSE, VIO_BLOCK_SPEEDBIAS, VIO_BLOCK_EXPOSE = 0, 1, 2
VIO_MARGIN_OLD, VIO_MARGIN_SECOND_NEW, VIO_MARGIN_NONE = 0, 1, 2
STAGES = ["setup_imu", "setup_prior", "eval_prior", "eval_imu", "eval_proj", "scale", "schur", "rhs", "cholesky",
          "tri_solve", "quad_form", "dogleg", "cost_eval", "new2old", "marg_build", "marg_chol", "total",
          "p_zero", "p_fact", "p_gram", "p_feat", "c_potrf", "c_trsm", "imu_raw", "c_wait", "q_w", "backsolve",
          "c_ahead", "tr_vec", "m_prior", "m_imu", "m_fact", "m_gram", "d0", "d1", "d2", "d3", "d4", "d5",
          "v_gd", "v_dot", "v_step", "v_plus", "v_gmax", "v_rest", "b_init", "b_pose", "b_asp", "b_band", "b_gn",
          "e_head", "e_copy", "e_h0dx", "e_tail", "x0", "x1", "x2", "x3"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class VioConfig(C.Structure):
    _fields_ = [
        ("window_size", C.c_int32), ("max_features", C.c_int32), ("max_factors", C.c_int32),
        ("max_iterations", C.c_int32), ("image_rows", C.c_int32), ("image_cols", C.c_int32),
        ("max_corners", C.c_int32), ("min_dist", C.c_int32), ("freq", C.c_int32),
        ("lk_win", C.c_int32), ("lk_levels", C.c_int32), ("lk_max_iters", C.c_int32),
        ("lk_eps", C.c_double), ("lk_min_eig", C.c_double), ("quality_level", C.c_double),
        ("f_threshold", C.c_double), ("f_confidence", C.c_double),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("gravity", C.c_double), ("acc_n", C.c_double), ("acc_w", C.c_double),
        ("gyr_n", C.c_double), ("gyr_w", C.c_double), ("cauchy_a", C.c_double),
    ]


def hip_runtime(lib=None):
    """-> (list of libamdhip64 files mapped into this process, as the library sees them)."""
    lib = lib or load_product()
    buf, n = C.create_string_buffer(4096), C.c_int32()
    lib.vio_hip_runtime(buf, 4096, C.byref(n))
    return [p for p in buf.value.decode().split(";") if p]


def host_pool_width():
    """(threads of one host pool incl. the caller, number of pools) of this process (vio_host_pool_width; csrc/vio_pool.h)."""
    lib = load_product()
    n = C.c_int32(0)
    return int(lib.vio_host_pool_width(C.byref(n))), int(n.value)


def default_config(**kw):
    """The reference's iPhone7P values (global_param.cpp:27-42, feature_tracker.hpp:24-29)."""
    c = VioConfig(
        window_size=10, max_features=1000, max_factors=8192, max_iterations=10,
        image_rows=640, image_cols=480, max_corners=70, min_dist=30, freq=3,
        lk_win=21, lk_levels=3, lk_max_iters=30, lk_eps=0.01, lk_min_eig=1e-4,
        quality_level=0.01, f_threshold=1.0, f_confidence=0.99,
        fx=526.600, fy=526.678, cx=243.481, cy=315.280,
        gravity=9.805, acc_n=0.5, acc_w=0.002, gyr_n=0.2, gyr_w=4.0e-5, cauchy_a=1.0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class VioPreintegration(C.Structure):
    _fields_ = [
        ("sum_dt", C.c_double), ("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4),
        ("delta_v", C.c_double * 3), ("linearized_ba", C.c_double * 3),
        ("linearized_bg", C.c_double * 3), ("jacobian", C.c_double * 225),
        ("covariance", C.c_double * 225),
    ]


PREINT_DOUBLES = C.sizeof(VioPreintegration) // 8  # 467


class VioPrior(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("n_blocks", C.c_int32),
        ("block_kind", C.c_int32 * VIO_MAX_PRIOR_BLOCKS),
        ("block_index", C.c_int32 * VIO_MAX_PRIOR_BLOCKS),
        ("block_offset", C.c_int32 * VIO_MAX_PRIOR_BLOCKS),
        ("block_x0", _dp), ("linearized_jacobians", _dp), ("linearized_residuals", _dp),
    ]


class VioWindow(C.Structure):
    _fields_ = [
        ("window_size", C.c_int32), ("n_features", C.c_int32), ("n_factors", C.c_int32),
        ("marginalization_flag", C.c_int32),
        ("pose", _dp), ("speed_bias", _dp), ("ex_pose", _dp), ("inv_depth", _dp),
        ("factor_host", _ip), ("factor_target", _ip), ("factor_feature", _ip),
        ("factor_pts_i", _dp), ("factor_pts_j", _dp),
        ("preint", C.POINTER(VioPreintegration)), ("prior", C.POINTER(VioPrior)),
        ("loop_frame", C.c_int32), ("loop_pose", _dp),
        ("use_origin_override", C.c_int32), ("origin_yaw_deg", C.c_double),
        ("origin_p", C.c_double * 3),
        ("raw_pose", _dp), ("raw_speed_bias", _dp), ("raw_inv_depth", _dp),
        ("next_prior", C.POINTER(VioPrior)),
        ("resident_prior", C.c_int32),
    ]


class VioSolveStats(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("iterations", C.c_int32), ("termination", C.c_int32),
        ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
        ("it_cost", C.c_double * VIO_MAX_TRACE), ("it_radius", C.c_double * VIO_MAX_TRACE),
        ("it_step_norm", C.c_double * VIO_MAX_TRACE),
        ("it_relative_decrease", C.c_double * VIO_MAX_TRACE),
        ("it_gradient_max_norm", C.c_double * VIO_MAX_TRACE),
        ("it_flags", C.c_int32 * VIO_MAX_TRACE),
    ]


class VioObs(C.Structure):
    _fields_ = [("id", C.c_int32), ("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class VioImuMsg(C.Structure):
    _fields_ = [("header", C.c_double), ("acc", C.c_double * 3), ("gyr", C.c_double * 3)]


class VioKeyframeData(C.Structure):
    _fields_ = [("header", C.c_double), ("translation", C.c_double * 3), ("rotation", C.c_double * 4)]


class VioPnpFeature(C.Structure):
    _fields_ = [("id", C.c_int32), ("observation", C.c_double * 2), ("position", C.c_double * 3), ("track_num", C.c_int32)]


class VioVinsResult(C.Structure):
    _fields_ = [("header", C.c_double), ("Ba", C.c_double * 3), ("Bg", C.c_double * 3), ("P", C.c_double * 3),
                ("R", C.c_double * 9), ("V", C.c_double * 3)]


class VioPnpWindow(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("pose", _dp), ("speed", _dp), ("bias", _dp), ("fixed", C.POINTER(C.c_uint8)),
                ("ex_pose", _dp), ("preint", C.POINTER(VioPreintegration)), ("feat_start", _ip), ("observation", _dp),
                ("position", _dp), ("track_num", _ip)]


class VioInitFrame(C.Structure):
    _fields_ = [("header", C.c_double), ("R", C.c_double * 9), ("T", C.c_double * 3), ("is_key_frame", C.c_int32),
                ("n_samples", C.c_int32), ("dt", _dp), ("acc", _dp), ("gyr", _dp), ("acc_0", C.c_double * 3),
                ("gyr_0", C.c_double * 3)]


class VioFrameResult(C.Structure):
    _fields_ = [("action", C.c_int32), ("error", C.c_int32), ("marginalization_flag", C.c_int32),
                ("failure_reasons", C.c_int32), ("track_num", C.c_int32), ("n_features", C.c_int32),
                ("n_factors", C.c_int32), ("n_loop_factors", C.c_int32), ("stats", VioSolveStats)]


class VioEstimatorStatus(C.Structure):
    _fields_ = [("frame_count", C.c_int32), ("solver_flag", C.c_int32), ("marginalization_flag", C.c_int32),
                ("failure_occur", C.c_int32), ("prior_rows", C.c_int32), ("final_cost", C.c_double),
                ("r_drift", C.c_double * 9), ("t_drift", C.c_double * 3), ("relative_t", C.c_double * 3),
                ("relative_q", C.c_double * 4), ("relative_yaw", C.c_double), ("loop_pose", C.c_double * 7),
                ("resident", C.c_int32), ("reserved", C.c_int32)]


VIO_SOLVER_INITIAL, VIO_SOLVER_NON_LINEAR = 0, 1
VIO_FAIL_FEW_FEATURES, VIO_FAIL_GYR_BIAS, VIO_FAIL_TRANSLATION, VIO_FAIL_Z_TRANSLATION, VIO_FAIL_ROTATION = 1, 2, 4, 8, 16
(VIO_FRAME_SKIPPED, VIO_FRAME_FILLING, VIO_FRAME_WAIT_INIT, VIO_FRAME_INIT_FAILED, VIO_FRAME_SOLVED, VIO_FRAME_FAILURE,
 VIO_FRAME_RESET, VIO_FRAME_ERROR) = range(8)


class VioFeatureInfo(C.Structure):
    _fields_ = [("id", C.c_int32), ("start_frame", C.c_int32), ("n_obs", C.c_int32), ("used_num", C.c_int32),
                ("solve_flag", C.c_int32), ("is_outlier", C.c_int32), ("fixed", C.c_int32),
                ("estimated_depth", C.c_double)]


def prior_capacity(W):
    return 6 * (W + 1) + 9 * (W + 1) + 6


def _ptr(a, ty=_dp):
    return a.ctypes.data_as(ty)


class Prior:
    """numpy-backed VioPrior (kept side of MarginalizationInfo)."""

    def __init__(self, W=None, cap=None):
        cap = cap if cap is not None else prior_capacity(W)
        self.cap = cap
        self.x0 = np.zeros((VIO_MAX_PRIOR_BLOCKS, 9))
        self.J = np.zeros(cap * cap)
        self.r = np.zeros(cap)
        self.c = VioPrior()
        self.c.block_x0 = _ptr(self.x0)
        self.c.linearized_jacobians = _ptr(self.J)
        self.c.linearized_residuals = _ptr(self.r)

    @property
    def n(self):
        return self.c.n

    @property
    def n_blocks(self):
        return self.c.n_blocks

    def blocks(self):
        """[(kind, index, offset, local_size)]"""
        out = []
        for b in range(self.c.n_blocks):
            k = self.c.block_kind[b]
            out.append((k, self.c.block_index[b], self.c.block_offset[b], 9 if k == VIO_BLOCK_SPEEDBIAS else 6))
        return out

    def jac(self):
        n = self.c.n
        return self.J[: n * n].reshape(n, n)

    def res(self):
        return self.r[: self.c.n]

    def canonical(self):
        """Column-order independent form: (H, b, x0) with blocks sorted by (kind, index).

        The reference's column order depends on unordered_map iteration over pointer keys
        (marginalization_factor.cpp:182-200,302-319); H = J0^T J0 and b = J0^T r0 are what the
        next solve actually consumes, and they are invariant to that order and to eigenvector signs.
        """
        J, r = self.jac(), self.res()
        order = sorted(self.blocks(), key=lambda t: (t[0], t[1]))
        cols = np.concatenate([np.arange(o, o + s) for (_, _, o, s) in order]) if order else np.zeros(0, int)
        Jc = J[:, cols]
        x0 = {}
        for b, (k, i, o, s) in enumerate(self.blocks()):
            x0[(k, i)] = self.x0[b, : (9 if k == VIO_BLOCK_SPEEDBIAS else 7)].copy()
        return Jc.T @ Jc, Jc.T @ r, [(k, i, x0[(k, i)]) for (k, i, _, _) in order]

    def copy(self):
        p = Prior(cap=self.cap)
        p.x0[:] = self.x0
        p.J[:] = self.J
        p.r[:] = self.r
        p.c.n, p.c.n_blocks = self.c.n, self.c.n_blocks
        for b in range(VIO_MAX_PRIOR_BLOCKS):
            p.c.block_kind[b] = self.c.block_kind[b]
            p.c.block_index[b] = self.c.block_index[b]
            p.c.block_offset[b] = self.c.block_offset[b]
        return p

    def header_only(self):
        """A copy whose data pointers are NULL: names the prior a device-resident slot holds (vio_amd.h)."""
        p = self.copy()
        p.c.block_x0 = p.c.linearized_jacobians = p.c.linearized_residuals = _dp()
        return p

    def to_npz_dict(self, prefix):
        n, nb = self.c.n, self.c.n_blocks
        return {
            prefix + "n": np.int32(n), prefix + "n_blocks": np.int32(nb),
            prefix + "kind": np.array(self.c.block_kind[:nb], np.int32),
            prefix + "index": np.array(self.c.block_index[:nb], np.int32),
            prefix + "offset": np.array(self.c.block_offset[:nb], np.int32),
            prefix + "x0": self.x0[:nb].copy(), prefix + "J": self.jac().copy(), prefix + "r": self.res().copy(),
        }

    @staticmethod
    def from_npz_dict(d, prefix, W):
        p = Prior(W)
        n, nb = int(d[prefix + "n"]), int(d[prefix + "n_blocks"])
        p.c.n, p.c.n_blocks = n, nb
        for b in range(nb):
            p.c.block_kind[b] = int(d[prefix + "kind"][b])
            p.c.block_index[b] = int(d[prefix + "index"][b])
            p.c.block_offset[b] = int(d[prefix + "offset"][b])
        p.x0[:nb] = d[prefix + "x0"]
        p.J[: n * n] = np.asarray(d[prefix + "J"]).ravel()
        p.r[:n] = d[prefix + "r"]
        return p


class Window:
    """numpy-backed VioWindow: one sliding window as solve_ceres sees it after old2new()."""

    def __init__(self, W, pose, speed_bias, ex_pose, inv_depth, factor_host, factor_target,
                 factor_feature, pts_i, pts_j, preint, prior=None, marginalization_flag=VIO_MARGIN_OLD,
                 loop_frame=-1):
        P = W + 1
        self.W = W
        self.pose = np.ascontiguousarray(pose, np.float64).reshape(P, 7).copy()
        self.speed_bias = np.ascontiguousarray(speed_bias, np.float64).reshape(P, 9).copy()
        self.ex_pose = np.ascontiguousarray(ex_pose, np.float64).reshape(7).copy()
        self.inv_depth = np.ascontiguousarray(inv_depth, np.float64).ravel().copy()
        self.factor_host = np.ascontiguousarray(factor_host, np.int32).copy()
        self.factor_target = np.ascontiguousarray(factor_target, np.int32).copy()
        self.factor_feature = np.ascontiguousarray(factor_feature, np.int32).copy()
        self.pts_i = np.ascontiguousarray(pts_i, np.float64).reshape(-1, 3).copy()
        self.pts_j = np.ascontiguousarray(pts_j, np.float64).reshape(-1, 3).copy()
        # preint: float64 array [W, PREINT_DOUBLES] with the exact VioPreintegration layout
        self.preint = np.ascontiguousarray(preint, np.float64).reshape(W, PREINT_DOUBLES).copy()
        self.prior = prior
        self.marginalization_flag = marginalization_flag
        self.loop_frame = loop_frame
        self.loop_pose = np.zeros(7)
        self.use_origin_override = 0
        self.origin_yaw_deg = 0.0
        self.origin_p = np.zeros(3)
        self.resident_prior = 0  # k >= 1: slot k-1 of the back-end's device-resident prior store
        self.raw_pose = np.zeros((P, 7))
        self.raw_speed_bias = np.zeros((P, 9))
        self.raw_inv_depth = np.zeros(len(self.inv_depth))
        self.next_prior = Prior(W)

    @property
    def n_features(self):
        return len(self.inv_depth)

    @property
    def n_factors(self):
        return len(self.factor_host)

    def copy(self):
        w = Window(self.W, self.pose, self.speed_bias, self.ex_pose, self.inv_depth, self.factor_host,
                   self.factor_target, self.factor_feature, self.pts_i, self.pts_j, self.preint,
                   self.prior.copy() if self.prior is not None else None, self.marginalization_flag,
                   self.loop_frame)
        w.use_origin_override = self.use_origin_override
        w.origin_yaw_deg = self.origin_yaw_deg
        w.origin_p = self.origin_p.copy()
        return w

    def fill_struct(self, c):
        c.window_size = self.W
        c.n_features = self.n_features
        c.n_factors = self.n_factors
        c.marginalization_flag = self.marginalization_flag
        c.pose, c.speed_bias = _ptr(self.pose), _ptr(self.speed_bias)
        c.ex_pose, c.inv_depth = _ptr(self.ex_pose), _ptr(self.inv_depth)
        c.factor_host = _ptr(self.factor_host, _ip)
        c.factor_target = _ptr(self.factor_target, _ip)
        c.factor_feature = _ptr(self.factor_feature, _ip)
        c.factor_pts_i, c.factor_pts_j = _ptr(self.pts_i), _ptr(self.pts_j)
        c.preint = C.cast(self.preint.ctypes.data, C.POINTER(VioPreintegration))
        c.prior = C.pointer(self.prior.c) if self.prior is not None else C.POINTER(VioPrior)()
        c.loop_frame = self.loop_frame
        c.loop_pose = _ptr(self.loop_pose)
        c.use_origin_override = self.use_origin_override
        c.origin_yaw_deg = self.origin_yaw_deg
        for k in range(3):
            c.origin_p[k] = self.origin_p[k]
        c.raw_pose, c.raw_speed_bias = _ptr(self.raw_pose), _ptr(self.raw_speed_bias)
        c.raw_inv_depth = _ptr(self.raw_inv_depth)
        c.next_prior = C.pointer(self.next_prior.c)
        c.resident_prior = self.resident_prior
        return c

    def struct(self):
        return self.fill_struct(VioWindow())

    # ---- fixtures -------------------------------------------------------------------------------
    def to_npz_dict(self):
        d = dict(W=np.int32(self.W), pose=self.pose, speed_bias=self.speed_bias, ex_pose=self.ex_pose,
                 inv_depth=self.inv_depth, factor_host=self.factor_host, factor_target=self.factor_target,
                 factor_feature=self.factor_feature, pts_i=self.pts_i, pts_j=self.pts_j, preint=self.preint,
                 marginalization_flag=np.int32(self.marginalization_flag), loop_frame=np.int32(self.loop_frame),
                 has_prior=np.int32(self.prior is not None))
        if self.prior is not None:
            d.update(self.prior.to_npz_dict("prior_"))
        return d

    @staticmethod
    def from_npz_dict(d):
        W = int(d["W"])
        prior = Prior.from_npz_dict(d, "prior_", W) if int(d["has_prior"]) else None
        return Window(W, d["pose"], d["speed_bias"], d["ex_pose"], d["inv_depth"], d["factor_host"],
                      d["factor_target"], d["factor_feature"], d["pts_i"], d["pts_j"], d["preint"], prior,
                      int(d["marginalization_flag"]), int(d["loop_frame"]))


def stats_to_dict(s):
    n = min(s.iterations, VIO_MAX_TRACE)
    return dict(initial_cost=s.initial_cost, final_cost=s.final_cost, iterations=s.iterations,
                termination=s.termination, num_successful_steps=s.num_successful_steps,
                num_unsuccessful_steps=s.num_unsuccessful_steps,
                it_cost=np.array(s.it_cost[:n]), it_radius=np.array(s.it_radius[:n]),
                it_step_norm=np.array(s.it_step_norm[:n]),
                it_relative_decrease=np.array(s.it_relative_decrease[:n]),
                it_gradient_max_norm=np.array(s.it_gradient_max_norm[:n]),
                it_flags=np.array(s.it_flags[:n], np.int32))


# ---- library loading -----------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "csrc", "libvio_amd.so")


def bind_backend_solver(lib, prefix):
    """Declares `<prefix>_solve_window(cfg, window, stats)` / preintegrate on a checker library."""
    f = getattr(lib, prefix + "_solve_window")
    f.argtypes = [C.POINTER(VioConfig), C.POINTER(VioWindow), C.POINTER(VioSolveStats)]
    f.restype = C.c_int
    g = getattr(lib, prefix + "_preintegrate")
    g.argtypes = [C.POINTER(VioConfig), _dp, _dp, _dp, _dp, C.c_int32, _dp, _dp, _dp, C.POINTER(VioPreintegration)]
    g.restype = C.c_int
    return f, g


def preintegrate_with(fn, cfg, acc0, gyr0, ba, bg, dt, acc, gyr):
    """Runs a `*_preintegrate` entry point; returns the struct as a float64 row (PREINT_DOUBLES)."""
    out = np.zeros(PREINT_DOUBLES)
    a = [np.ascontiguousarray(x, np.float64) for x in (acc0, gyr0, ba, bg, dt, acc, gyr)]
    rc = fn(C.byref(cfg), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), len(a[4]), _ptr(a[4]), _ptr(a[5]),
            _ptr(a[6]), C.cast(out.ctypes.data, C.POINTER(VioPreintegration)))
    if rc != VIO_OK:
        raise RuntimeError("preintegrate failed rc=%d" % rc)
    return out


_product = None


def load_product():
    """Loads csrc/libvio_amd.so. Fails loudly when the HIP extension was not built: there is no
    CPU fallback for the product path."""
    global _product
    if _product is not None:
        return _product
    if not os.path.exists(PRODUCT_LIB):
        raise RuntimeError("HIP extension missing: %s (run __graft_entry__.build())" % PRODUCT_LIB)
    lib = C.CDLL(PRODUCT_LIB)
    vp = C.c_void_p
    lib.vio_version.restype = C.c_char_p
    lib.vio_config_default.argtypes = [C.POINTER(VioConfig)]
    lib.vio_prior_capacity.argtypes = [C.c_int32]
    lib.vio_prior_capacity.restype = C.c_int32
    lib.vio_backend_create.argtypes = [C.POINTER(VioConfig), C.c_int32, C.POINTER(vp)]
    lib.vio_backend_destroy.argtypes = [vp]
    lib.vio_backend_destroy.restype = None
    lib.vio_preintegrate.argtypes = [C.POINTER(VioConfig), _dp, _dp, _dp, _dp, C.c_int32, _dp, _dp, _dp,
                                     C.POINTER(VioPreintegration)]
    lib.vio_backend_solve_windows.argtypes = [vp, C.POINTER(VioWindow), C.c_int32, C.c_int32,
                                              C.POINTER(VioSolveStats)]
    lib.vio_backend_reserve_priors.argtypes = [vp, C.c_int32]
    lib.vio_backend_get_device.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.vio_frontend_get_device.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.vio_backend_upload.argtypes = [vp, C.POINTER(VioWindow), C.c_int32]
    lib.vio_backend_launch.argtypes = [vp, vp]
    lib.vio_backend_sync.argtypes = [vp]
    lib.vio_backend_download.argtypes = [vp, C.POINTER(VioWindow), C.c_int32, C.POINTER(VioSolveStats)]
    lib.vio_backend_kernel_ms.argtypes = [vp, _dp, _ip]
    u8p, fp = C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    cfgp = C.POINTER(VioConfig)
    lib.vio_frontend_create.argtypes = [cfgp, C.c_int32, C.POINTER(vp)]
    lib.vio_frontend_destroy.argtypes = [vp]
    lib.vio_frontend_destroy.restype = None
    lib.vio_frontend_read_image.argtypes = [vp, C.c_int32, u8p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                                            C.POINTER(VioObs), _ip, vp]
    lib.vio_frontend_read_images.argtypes = [vp, u8p, C.c_int32, C.c_int32, C.c_int32, _dp, C.c_int32,
                                             C.POINTER(VioObs), _ip]
    lib.vio_hip_runtime.argtypes = [C.c_char_p, C.c_int32, _ip]
    lib.vio_host_pool_width.argtypes = [_ip]
    lib.vio_frontend_submit_images.argtypes = [vp, u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.vio_frontend_submit_images_async.argtypes = [vp, u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.vio_frontend_collect.argtypes = [vp, C.POINTER(VioObs), _ip]
    lib.vio_host_register.argtypes = [vp, C.c_size_t]
    lib.vio_host_unregister.argtypes = [vp]
    lib.vio_frontend_upload_frames.argtypes = [vp, u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.vio_frontend_step_resident.argtypes = [vp, C.c_int32, C.c_int32, vp]
    lib.vio_frontend_sync.argtypes = [vp]
    lib.vio_frontend_kernel_ms.argtypes = [vp, _dp, _ip]
    lib.vio_frontend_get_state.argtypes = [vp, C.c_int32, fp, _ip, _ip, C.c_int32, _ip]
    lib.vio_frontend_get_pnp_points.argtypes = [vp, C.c_int32, fp, _ip, C.c_int32, _ip]
    lib.vio_frontend_lk_iterations.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32]
    lib.vio_klt_track.argtypes = [cfgp, u8p, u8p, C.c_int32, C.c_int32, C.c_int32, fp, C.c_int32, fp, u8p, fp]
    lib.vio_good_features.argtypes = [cfgp, u8p, u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, _ip]
    lib.vio_fundamental_ransac.argtypes = [cfgp, fp, fp, C.c_int32, u8p]
    obsp, infop = C.POINTER(VioObs), C.POINTER(VioFeatureInfo)
    lib.vio_features_create.argtypes = [C.c_int32, C.POINTER(vp)]
    lib.vio_features_destroy.argtypes = [vp]
    lib.vio_features_destroy.restype = None
    lib.vio_features_clear.argtypes = [vp]
    lib.vio_features_add_check_parallax.argtypes = [vp, C.c_int32, obsp, C.c_int32, _ip, _ip, _ip]
    lib.vio_features_count.argtypes = [vp, _ip]
    lib.vio_features_get_depth_vector.argtypes = [vp, _dp, C.c_int32, _ip]
    lib.vio_features_set_depth.argtypes = [vp, _dp, C.c_int32]
    lib.vio_features_clear_depth.argtypes = [vp, _dp, C.c_int32]
    lib.vio_features_triangulate.argtypes = [vp, _dp, _dp, _dp, _dp]
    lib.vio_features_remove_failures.argtypes = [vp]
    lib.vio_features_remove_back.argtypes = [vp]
    lib.vio_features_remove_back_shift_depth.argtypes = [vp, _dp, _dp, _dp, _dp]
    lib.vio_features_remove_front.argtypes = [vp, C.c_int32]
    lib.vio_features_export_factors.argtypes = [vp, C.c_int32, _ip, _ip, _ip, _dp, _dp, _ip, _ip]
    lib.vio_features_dump.argtypes = [vp, infop, C.c_int32, _ip, _dp, C.c_int32, _ip]
    lib.vio_failure_detection.argtypes = [C.c_int32, _dp, _dp, _dp, _dp, _dp, _ip]
    lib.vio_features_export_factors_loop.argtypes = [vp, C.c_int32, C.c_int32, _ip, _dp, C.c_int32, _ip, _ip, _ip, _dp, _dp,
                                                     _ip, _ip, _ip]
    lib.vio_preprocess_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.vio_preprocess_destroy.argtypes = [vp]
    lib.vio_preprocess_destroy.restype = None
    lib.vio_preprocess_set_clahe.argtypes = [vp, C.c_double, C.c_int32, C.c_int32]
    lib.vio_preprocess_run.argtypes = [vp, u8p, C.c_int32, C.c_int32, C.c_int32, u8p, u8p]
    lib.vio_preprocess_run_resident.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp]
    lib.vio_preprocess_sync.argtypes = [vp]
    lib.vio_preprocess_kernel_ms.argtypes = [vp, _dp, _ip]
    imup, kfp, i64 = C.POINTER(VioImuMsg), C.POINTER(VioKeyframeData), C.c_int64
    lib.vio_replay_read_imu.argtypes = [C.c_char_p, imup, C.c_int32, _ip]
    lib.vio_replay_write_imu.argtypes = [C.c_char_p, imup, C.c_int32]
    lib.vio_replay_read_image_time.argtypes = [C.c_char_p, C.c_uint64, _dp]
    lib.vio_replay_write_image_time.argtypes = [C.c_char_p, C.c_uint64, C.c_double]
    lib.vio_replay_read_image.argtypes = [C.c_char_p, C.c_uint64, u8p, i64, _ip, _ip]
    lib.vio_replay_decode_png_gray.argtypes = [u8p, i64, u8p, i64, _ip, _ip]
    lib.vio_replay_write_image.argtypes = [C.c_char_p, C.c_uint64, u8p, C.c_int32, C.c_int32, C.c_int32]
    lib.vio_replay_rgba_to_gray.argtypes = [u8p, C.c_int32, C.c_int32, C.c_int32, u8p]
    lib.vio_replay_read_keyframes.argtypes = [C.c_char_p, kfp, C.c_int32, _ip]
    lib.vio_replay_write_keyframes.argtypes = [C.c_char_p, kfp, C.c_int32]
    lib.vio_measurements_create.argtypes = [C.POINTER(vp)]
    lib.vio_measurements_destroy.argtypes = [vp]
    lib.vio_measurements_destroy.restype = None
    lib.vio_measurements_push_imu.argtypes = [vp, imup]
    lib.vio_measurements_push_image.argtypes = [vp, C.c_double, obsp, C.c_int32]
    lib.vio_measurements_next.argtypes = [vp, imup, _dp, C.c_int32, _ip, _dp, obsp, C.c_int32, _ip, _ip]
    lib.vio_pnp_create.argtypes = [cfgp, C.c_int32, C.POINTER(vp)]
    lib.vio_pnp_destroy.argtypes = [vp]
    lib.vio_pnp_destroy.restype = None
    lib.vio_pnp_solve_windows.argtypes = [vp, C.POINTER(VioPnpWindow), C.c_int32, C.POINTER(VioSolveStats)]
    lib.vio_pnp_kernel_ms.argtypes = [vp, _dp, _ip]
    lib.vio_pnp_tracker_create.argtypes = [cfgp, C.c_int32, C.c_int32, _dp, _dp, C.POINTER(vp)]
    lib.vio_pnp_tracker_destroy.argtypes = [vp]
    lib.vio_pnp_tracker_destroy.restype = None
    lib.vio_pnp_tracker_clear.argtypes = [vp, C.c_int32]
    lib.vio_pnp_tracker_set_init.argtypes = [vp, C.c_int32, C.POINTER(VioVinsResult)]
    lib.vio_pnp_tracker_process_imu.argtypes = [vp, C.c_int32, C.c_double, _dp, _dp]
    lib.vio_pnp_tracker_process_images.argtypes = [vp, C.POINTER(VioPnpFeature), _ip, C.c_int32, _dp, C.c_int32, u8p, _dp, _dp, _ip]
    lib.vio_pnp_match_features.argtypes = [cfgp, _ip, fp, C.c_int32, C.POINTER(VioPnpFeature), C.c_int32, C.POINTER(VioPnpFeature),
                                           C.c_int32, _ip]
    lib.vio_pnp_tracker_get_window.argtypes = [vp, C.c_int32, _dp, _dp, _dp, _dp, u8p, _ip]
    lib.vio_init_relative_pose.argtypes = [_dp, _dp, C.c_int32, _dp, _dp, _dp, _ip, _ip]
    lib.vio_init_relative_pose_mode.argtypes = [_dp, _dp, C.c_int32, C.c_int32, _dp, _dp, _dp, _ip, _ip]
    lib.vio_init_five_point.argtypes = [_dp, _dp, _dp, _ip]
    lib.vio_init_recover_pose.argtypes = [_dp, _dp, _dp, C.c_int32, _dp, _dp, _ip]
    lib.vio_init_pnp.argtypes = [_dp, _dp, C.c_int32, _dp, _dp, _ip]
    lib.vio_init_triangulate_point.argtypes = [_dp, _dp, _dp, _dp, _dp]
    lib.vio_init_bundle_adjust.argtypes = [C.c_int32, C.c_int32, _dp, _dp, C.c_int32, _dp, u8p, _ip, _ip, _dp, C.POINTER(VioSolveStats), _ip]
    lib.vio_init_sfm.argtypes = [C.c_int32, C.c_int32, _dp, _dp, C.c_int32, _ip, _ip, _dp, _dp, _dp, _dp, u8p, _ip]
    lib.vio_visual_imu_alignment.argtypes = [cfgp, _dp, C.POINTER(VioInitFrame), C.c_int32, C.c_int32, _dp, _dp, _dp, _ip]
    resp, stp = C.POINTER(VioFrameResult), C.POINTER(VioEstimatorStatus)
    lib.vio_estimator_create.argtypes = [cfgp, C.c_int32, _dp, _dp, C.POINTER(vp)]
    lib.vio_estimator_destroy.argtypes = [vp]
    lib.vio_estimator_destroy.restype = None
    lib.vio_estimator_clear.argtypes = [vp, C.c_int32]
    lib.vio_estimator_enable_initialization.argtypes = [vp, C.c_int32]
    lib.vio_estimator_set_resident.argtypes = [vp, C.c_int32]
    lib.vio_features_scale_depth.argtypes = [vp, C.c_double]
    lib.vio_estimator_process_imu.argtypes = [vp, C.c_int32, C.c_double, _dp, _dp]
    lib.vio_estimator_process_imu_batch.argtypes = [vp, _ip, C.c_int32, _dp, _dp, _dp]
    lib.vio_estimator_set_initial_state.argtypes = [vp, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp]
    lib.vio_estimator_set_relocalization.argtypes = [vp, C.c_int32, C.c_double, _dp, _dp, _ip, _dp, C.c_int32]
    lib.vio_estimator_process_images.argtypes = [vp, obsp, _ip, C.c_int32, _dp, u8p, resp]
    lib.vio_estimator_process_image.argtypes = [vp, C.c_int32, obsp, C.c_int32, C.c_double, resp]
    lib.vio_estimator_get_status.argtypes = [vp, C.c_int32, stp]
    lib.vio_estimator_get_window.argtypes = [vp, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp]
    lib.vio_estimator_get_corrected_window.argtypes = [vp, C.c_int32, _dp, _dp]
    lib.vio_estimator_features.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    lib.vio_estimator_get_timing.argtypes = [vp, _dp]
    lib.vio_backend_set_profile.argtypes = [vp, C.c_int32]
    lib.vio_backend_stage_cycles.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64), C.c_int32]
    _product = lib
    return lib
