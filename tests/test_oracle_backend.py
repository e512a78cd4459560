"""CPU tests: the plain-C++ restatement (oracle/libvio_oracle.so) against golden vectors produced by the REAL
reference (vendored Ceres 1.12 + VINS_ios factors; tests/golden/make_golden.py). This is what pins the oracle."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from helpers import abi


def P(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_preintegration_matches_reference():
    d = np.load(H.GOLDEN + "/factors.npz")
    _, opre = H.oracle_backend()
    cfg = abi.default_config()
    for c in range(len(d["pre_n"])):
        n = int(d["pre_n"][c])
        out = abi.preintegrate_with(opre, cfg, d["pre_acc0"][c], d["pre_gyr0"][c], d["pre_ba"][c], d["pre_bg"][c],
                                    d["pre_dt"][c][:n], d["pre_acc"][c][:n], d["pre_gyr"][c][:n])
        ref = d["pre_out"][c]
        # state, jacobian, covariance compared separately (covariance entries span 1e-12..1e-2)
        assert np.allclose(out[:17], ref[:17], rtol=1e-12, atol=1e-15)
        assert H.relerr(out[17:242], ref[17:242]) < 1e-12
        assert H.relerr(out[242:], ref[242:]) < 1e-12


def test_projection_factor_matches_reference():
    d = np.load(H.GOLDEN + "/factors.npz")
    w = abi.Window.from_npz_dict({k[4:]: v for k, v in d.items() if k.startswith("win_")})
    lib = H.oracle_lib()
    cfg = abi.default_config()
    for q, k in enumerate(d["proj_idx"]):
        h, t, f = w.factor_host[k], w.factor_target[k], w.factor_feature[k]
        res, jac = np.zeros(2), np.zeros(44)
        rc = lib.oracle_eval_projection(C.byref(cfg), P(w.pose[h]), P(w.pose[t]), P(w.ex_pose),
                                        P(w.inv_depth[f:f + 1]), P(w.pts_i[k]), P(w.pts_j[k]), P(res), P(jac))
        assert rc == 0
        assert np.allclose(res, d["proj_res"][q], rtol=1e-10, atol=1e-10)
        assert H.relerr(jac, d["proj_jac"][q]) < 1e-11


def test_imu_factor_matches_reference():
    d = np.load(H.GOLDEN + "/factors.npz")
    w = abi.Window.from_npz_dict({k[4:]: v for k, v in d.items() if k.startswith("win_")})
    lib = H.oracle_lib()
    cfg = abi.default_config()
    for i in range(w.W):
        res, jac = np.zeros(15), np.zeros(480)
        rc = lib.oracle_eval_imu(C.byref(cfg), C.cast(w.preint[i].ctypes.data, C.POINTER(abi.VioPreintegration)),
                                 P(w.pose[i]), P(w.speed_bias[i]), P(w.pose[i + 1]), P(w.speed_bias[i + 1]),
                                 P(res), P(jac))
        assert rc == 0
        # sqrt_info comes from inverting an ill-conditioned covariance (cond ~1e9): LU vs Gauss-Jordan agree to ~1e-7
        assert H.relerr(res, d["imu_res"][i]) < 1e-6
        assert H.relerr(jac, d["imu_jac"][i]) < 1e-6


@pytest.mark.parametrize("name", H.golden_window_names())
def test_window_solve_matches_reference(name):
    cfg, w, d = H.load_golden_window(name)
    osolve, _ = H.oracle_backend()
    got, stats = H.solve_with(osolve, cfg, w)
    # north_star bar is 1e-4 relative; the restatement is held to 1e-6
    H.check_solution(got, stats, d, tol=1e-6, tol_prior=1e-5)


def test_oracle_rejects_bad_indices():
    cfg, w, _ = H.load_golden_window("win_small_w4")
    osolve, _ = H.oracle_backend()
    w.factor_feature[0] = w.n_features  # out of range
    st = abi.VioSolveStats()
    assert osolve(C.byref(cfg), C.byref(w.struct()), C.byref(st)) == abi.VIO_EINVAL


def test_reference_build_passes_ceres_own_unit_tests():
    """oracle/_ref is a build of the vendored Ceres made here from the reference tree; where that tree exists, a few of
    Ceres' OWN unit tests — the components the back-end restates (corrector, loss functions, dogleg strategy,
    trust-region minimizer, Schur eliminator / complement solver, residual blocks, parameter-block ordering)
    — are compiled where they lie and run against the very objects libvio_ref.so is linked from."""
    import os
    import subprocess
    if not os.path.isdir("/root/reference") or not os.path.isdir(os.path.join(H.ROOT, "oracle", "_ref", "obj")):
        pytest.skip("needs /root/reference and the object files of oracle/_ref")
    r = subprocess.run(["make", "-s", "-C", os.path.join(H.ROOT, "oracle"), "ref-selftest"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "PASSED" in r.stdout and "FAILED" not in r.stdout
    n = int(r.stdout.split("PASSED  ]")[1].split("tests")[0])
    assert n >= 39, r.stdout[-500:]
