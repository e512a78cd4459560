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
