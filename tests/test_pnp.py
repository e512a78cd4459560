"""The motion-only window of the front-end (vinsPnP::solve_ceres, VINS_ios/vins_pnp.cpp:264-341).

Checker: the REAL reference factor classes (IMUFactorPnP, PerspectiveFactor) under the vendored Ceres, assembled exactly
as vins_pnp.cpp does (oracle/ref_harness.cpp::ref_pnp_solve; vins_pnp.cpp itself needs an OpenCV header) — live where
oracle/_ref exists, else through tests/golden/pnp_windows.npz. CPU: the kernel's source compiled for the host
(tests/emul/emul_pnp.cpp, -DVIO_EMUL). GPU: the HIP kernel through the C ABI."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from helpers import abi, pkg, synth

GOLDEN = os.path.join(H.ROOT, "tests", "golden", "pnp_windows.npz")
EMUL_DIR = os.path.join(H.ROOT, "tests", "emul")
TOL = 1e-6

# (seed, frames, features per frame, fixed frames, perturbation scale)
CASES = [(1, 7, 60, (0,), 1.0), (2, 7, 150, (0,), 3.0), (3, 7, 25, (0, 1), 1.0), (4, 5, 80, (), 1.0), (5, 7, 8, (2,), 6.0),
         (6, 7, 100, (0, 1, 2, 3, 4, 5, 6), 1.0), (7, 3, 40, (0,), 10.0)]


def make_window(cfg, seed, n, feats, fixed, perturb):
    rng = np.random.default_rng(seed)
    traj = synth.Trajectory(rng)
    t0 = rng.uniform(0, 20)
    ex = synth.ex_pose_default()
    ric, tic = synth.quat_to_rot(ex[3:]), ex[:3]
    g = np.array([0, 0, cfg.gravity])
    ba, bg = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
    fdt, per = 1.0 / 30, 4
    dt = fdt / per

    def imu(t):
        R = traj.rot(t)
        return R.T @ (traj.acc(t) + g) + ba + rng.normal(0, 0.02, 3), traj.omega_body(t) + bg + rng.normal(0, 0.002, 3)

    pose, speed, pre = [], [], []
    last = imu(t0)
    for k in range(n):
        t = t0 + k * fdt
        P, R, V = traj.pos(t), traj.rot(t), traj.vel(t)
        if k in fixed:
            Pn, Rn, Vn = P, R, V
        else:
            Pn = P + rng.normal(0, 0.004 * perturb, 3)
            Rn = R @ synth.rotvec_to_rot(rng.normal(0, 0.003 * perturb, 3))
            Vn = V + rng.normal(0, 0.02 * perturb, 3)
        pose.append(np.concatenate([Pn, synth.rot_to_quat(Rn)])), speed.append(Vn)
        if k > 0:
            samples = [imu(t - fdt + (s + 1) * dt) for s in range(per)]
            pre.append(pkg.backend.preintegrate(cfg, last[0], last[1], ba, bg, np.full(per, dt), np.array([s[0] for s in samples]),
                                                np.array([s[1] for s in samples])))
            last = samples[-1]
    # landmarks in front of the middle camera, "solved" by the back-end to within a centimetre or two
    tm = t0 + (n // 2) * fdt
    Rm, Pm = traj.rot(tm) @ ric, traj.pos(tm) + traj.rot(tm) @ tic
    z = rng.uniform(3, 10, feats)
    Xw = np.column_stack([rng.uniform(-0.4, 0.4, feats) * z, rng.uniform(-0.5, 0.5, feats) * z, z]) @ Rm.T + Pm
    track = rng.integers(2, 40, feats)
    start, obs, pos, tn = [0], [], [], []
    for k in range(n):
        t = t0 + k * fdt
        Rc, Pc = traj.rot(t) @ ric, traj.pos(t) + traj.rot(t) @ tic
        for j in range(feats):
            if rng.random() < 0.1:
                continue
            c = Rc.T @ (Xw[j] - Pc)
            obs.append(c[:2] / c[2] + rng.normal(0, 0.7 / cfg.fx, 2))
            pos.append(Xw[j] + rng.normal(0, 0.01, 3))
            tn.append(track[j])
        start.append(len(obs))
    fx = np.zeros(n, np.uint8)
    fx[list(fixed)] = 1
    return pkg.pnp.PnpWindow(np.array(pose), np.array(speed), np.tile(np.concatenate([ba, bg]), (n, 1)), fx, ex, np.array(pre), start,
                             np.array(obs), np.array(pos), np.array(tn))


def reference(cfg, seed, w):
    lib = H.ref_lib_or_none()
    if lib is not None and hasattr(lib, "ref_pnp_solve"):
        lib.ref_pnp_solve.argtypes = None
        return pkg.pnp.solve_with(lib.ref_pnp_solve, cfg, w)
    d = np.load(GOLDEN)
    out = w.copy()
    out.pose, out.speed = d["c%d_pose" % seed], d["c%d_speed" % seed]
    st = {k: d["c%d_%s" % (seed, k)] for k in ("initial_cost", "final_cost", "iterations", "it_cost", "it_flags")}
    return out, st


def check(got, gs, ref, rs, tol=TOL):
    assert gs["iterations"] == int(rs["iterations"]) and list(gs["it_flags"]) == list(rs["it_flags"])
    assert H.relerr(np.array(gs["it_cost"]), np.array(rs["it_cost"])) < 1e-6
    assert abs(gs["initial_cost"] - float(rs["initial_cost"])) <= 1e-9 * float(rs["initial_cost"])
    assert H.pose_relerr(got.pose, ref.pose) < tol and H.relerr(got.speed, ref.speed) < tol


@pytest.fixture(scope="module")
def emul():
    so = os.path.join(EMUL_DIR, "libvio_emul_pnp.so")
    csrc = os.path.join(H.ROOT, "vins-mobile_amd", "csrc")
    srcs = glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(EMUL_DIR, "emul_pnp.cpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DVIO_EMUL", "-I" + os.path.join(H.ROOT, "include"),
                               "-I" + csrc, "-shared", "-o", so, os.path.join(EMUL_DIR, "emul_pnp.cpp")])
    lib = C.CDLL(so)
    lib.emul_pnp_solve.argtypes = None
    return lib


@pytest.mark.parametrize("seed,n,feats,fixed,perturb", CASES)
def test_kernel_source_on_host_matches_the_reference(seed, n, feats, fixed, perturb, emul):
    cfg = abi.default_config()
    w = make_window(cfg, seed, n, feats, fixed, perturb)
    ref, rs = reference(cfg, seed, w)
    got, gs = pkg.pnp.solve_with(emul.emul_pnp_solve, cfg, w)
    check(got, gs, ref, rs)
    if len(fixed) < n:
        assert rs["final_cost"] < rs["initial_cost"]
    for k in fixed:   # constant blocks come back untouched (up to the quaternion round trip)
        assert np.abs(got.pose[k] - w.pose[k]).max() < 1e-12 and np.array_equal(got.speed[k], w.speed[k])


@pytest.mark.gpu
def test_device_kernel_matches_the_reference_in_one_ragged_launch():
    cfg = abi.default_config()
    ws = [make_window(cfg, *c) for c in CASES]
    solver = pkg.pnp.PnpSolver(cfg, max_batch=len(ws))
    got = [w.copy() for w in ws]
    stats = solver.solve(got)
    for c, w, g, s in zip(CASES, ws, got, stats):
        ref, rs = reference(cfg, c[0], w)
        check(g, s, ref, rs)
    ms, k = solver.kernel_ms()
    assert k == 1 and ms > 0
    solver.close()


@pytest.mark.gpu
def test_device_kernel_full_batch_and_errors():
    cfg = abi.default_config()
    base = make_window(cfg, 2, 7, 150, (0,), 3.0)
    solver = pkg.pnp.PnpSolver(cfg, max_batch=256)
    ws = [base.copy() for _ in range(256)]
    stats = solver.solve(ws)
    ref, rs = reference(cfg, 2, base)
    for g, s in zip(ws[::37], stats[::37]):
        check(g, s, ref, rs)
    assert all(np.abs(w.pose - ws[0].pose).max() < 1e-9 for w in ws)
    ms, _ = solver.kernel_ms()
    print("256 PnP windows (7 frames, ~950 factors): %.3f ms" % ms)
    with pytest.raises(RuntimeError):
        solver.solve([base.copy() for _ in range(257)])
    solver.close()


# ---------------------------------------------------------------------------------------------------------------------
# the vinsPnP object around the solve (vio_pnp_tracker_*)
class _Scene:
    """30 Hz camera, 120 Hz IMU, a landmark cloud with known positions: what the tracker + the back-end feed solveVinsPnP."""

    def __init__(self, cfg, seed):
        self.cfg, rng = cfg, np.random.default_rng(seed)
        self.rng = rng
        self.traj = synth.Trajectory(rng)
        self.t0 = rng.uniform(0, 20)
        ex = synth.ex_pose_default()
        self.ric, self.tic = synth.quat_to_rot(ex[3:]), ex[:3]
        self.g = np.array([0, 0, cfg.gravity])
        self.ba, self.bg = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
        self.fdt, self.per = 1.0 / 30, 4
        self.lm = np.column_stack([rng.uniform(-9, 9, 3000), rng.uniform(-9, 9, 3000), rng.uniform(-11, -4, 3000)])

    def t(self, k):
        return self.t0 + k * self.fdt

    def imu(self, t):
        R = self.traj.rot(t)
        return (R.T @ (self.traj.acc(t) + self.g) + self.ba + self.rng.normal(0, 0.02, 3),
                self.traj.omega_body(t) + self.bg + self.rng.normal(0, 0.002, 3))

    def imu_interval(self, k):
        dt = self.fdt / self.per
        return [(dt,) + self.imu(self.t(k - 1) + (s + 1) * dt) for s in range(self.per)]

    def features(self, k, n=120):
        P, R = self.traj.pos(self.t(k)), self.traj.rot(self.t(k))
        Rc, Pc = R @ self.ric, P + R @ self.tic
        pc = (self.lm - Pc) @ Rc
        vis = np.flatnonzero((pc[:, 2] > 0.5) & (np.abs(pc[:, 0]) < 0.4 * pc[:, 2]) & (np.abs(pc[:, 1]) < 0.55 * pc[:, 2]))[:n]
        return [(int(i), pc[i, :2] / pc[i, 2] + self.rng.normal(0, 0.5 / self.cfg.fx, 2), self.lm[i] + self.rng.normal(0, 0.01, 3),
                 int(5 + i % 20)) for i in vis]


def test_pnp_tracker_bookkeeping_without_a_solve():
    """Window filling, updateFeatures, setInit, IMU propagation and the slide with use_pnp = false never reach the
    device (feature_tracker.cpp:151 passes use_pnp through; the default is off)."""
    cfg = abi.default_config()
    sc = _Scene(cfg, 3)
    tr = pkg.pnp.PnpTracker(cfg, sc.tic, sc.ric, pnp_size=6)
    hdrs = []
    for k in range(10):
        for dt, a, w in ([(0.0,) + sc.imu(sc.t(0))] if k == 0 else sc.imu_interval(k)):
            tr.process_imu(dt, a, w)
        if k == 4:   # the back-end's result for frame 2 arrives: it becomes the constant of the window
            P, R, V = sc.traj.pos(sc.t(2)), sc.traj.rot(sc.t(2)), sc.traj.vel(sc.t(2))
            tr.set_init(sc.t(2), sc.ba, sc.bg, P, R, V)
            w = tr.window()
            assert list(w["find_solved"]) == [0, 0, 1, 0, 0, 0, 0] and np.array_equal(w["Ps"][2], P) and np.array_equal(w["Vs"][2], V)
        _, _, solved = tr.process_images([sc.features(k)], [sc.t(k)], use_pnp=False)
        assert solved[0] == 0
        hdrs.append(sc.t(k))
        w = tr.window()
        assert w["frame_count"] == min(k + 1, 6)
    w = tr.window()
    assert list(w["headers"][:6]) == hdrs[4:10]                      # four slides: frames 4..9 remain (+ the copy in slot 6)
    assert list(w["find_solved"]) == [0] * 7                         # frame 2 (the solved one) has left the window
    assert np.array_equal(w["Ps"][6], w["Ps"][5]) and np.array_equal(w["Rs"][6], w["Rs"][5])   # the slide seeds the next frame
    tr.close()


@pytest.mark.gpu
def test_pnp_tracker_follows_the_truth_between_backend_results():
    """solveVinsPnP as readImage runs it: IMU samples since the last frame, the landmarks the back-end has solved with
    their current observations, and every third frame the back-end's newest state (two frames late) through setInit."""
    cfg = abi.default_config()
    sc = _Scene(cfg, 4)
    tr = pkg.pnp.PnpTracker(cfg, sc.tic, sc.ric, pnp_size=6)
    errs, n_solved = [], 0
    for k in range(60):
        for dt, a, w in ([(0.0,) + sc.imu(sc.t(0))] if k == 0 else sc.imu_interval(k)):
            tr.process_imu(dt, a, w)
        if k >= 2 and k % 3 == 2:
            j = k - 2
            tr.set_init(sc.t(j), sc.ba, sc.bg, sc.traj.pos(sc.t(j)) + sc.rng.normal(0, 0.005, 3), sc.traj.rot(sc.t(j)),
                        sc.traj.vel(sc.t(j)) + sc.rng.normal(0, 0.01, 3))
        P, R, solved = tr.process_images([sc.features(k)], [sc.t(k)], use_pnp=True)
        n_solved += int(solved[0])
        if solved[0]:
            # P / R = the second-newest slot after the slide = the frame just solved
            errs.append(np.linalg.norm(P[0] - sc.traj.pos(sc.t(k))))
            assert np.abs(R[0] - sc.traj.rot(sc.t(k))).max() < 0.01
    assert n_solved == 60 - 6
    errs = np.array(errs)
    assert np.sqrt((errs ** 2).mean()) < 0.03 and errs.max() < 0.08, (np.sqrt((errs ** 2).mean()), errs.max())
    tr.close()


def test_solved_features_are_joined_with_the_tracker_points_by_id():
    cfg = abi.default_config()
    lib = abi.load_product()
    ids = np.array([3, 4, 7, 9, 12, 15, 20], np.int32)                 # tracker: ascending ids
    pts = np.array([[10 * i + 0.25, 20 * i + 0.5] for i in range(7)], np.float32)
    solved = (abi.VioPnpFeature * 5)()
    for k, (fid, tn) in enumerate([(1, 5), (4, 6), (9, 7), (13, 8), (20, 9)]):   # back-end: ascending ids, some lost by the tracker
        solved[k].id, solved[k].track_num = fid, tn
        solved[k].position[:] = [fid * 1.0, fid * 2.0, fid * 3.0]
    out, n = (abi.VioPnpFeature * 8)(), C.c_int32()
    rc = lib.vio_pnp_match_features(C.byref(cfg), ids.ctypes.data_as(C.POINTER(C.c_int32)), pts.ctypes.data_as(C.POINTER(C.c_float)), 7,
                                    solved, 5, out, 8, C.byref(n))
    assert rc == 0 and n.value == 3
    assert [out[i].id for i in range(3)] == [4, 9, 20] and [out[i].track_num for i in range(3)] == [6, 7, 9]
    for o, row in zip(out[:3], (1, 3, 6)):
        assert abs(o.observation[0] - (float(pts[row, 0]) - cfg.cx) / cfg.fx) < 1e-15
        assert abs(o.observation[1] - (float(pts[row, 1]) - cfg.cy) / cfg.fy) < 1e-15
        assert list(o.position) == [o.id * 1.0, o.id * 2.0, o.id * 3.0]
    assert lib.vio_pnp_match_features(C.byref(cfg), ids.ctypes.data_as(C.POINTER(C.c_int32)), pts.ctypes.data_as(C.POINTER(C.c_float)), 7,
                                      solved, 5, out, 2, C.byref(n)) == abi.VIO_ECAP
