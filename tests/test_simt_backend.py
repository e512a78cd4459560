"""CPU test of the DEVICE SECTIONS of the kernel source: vins-mobile_amd/csrc/solver_core.h + marg_core.h compiled with
-DVIO_SIMT and executed by the wave64 SIMT emulator of tests/emul/simt.h (one fiber per work-item; v_mfma_f64_16x16x4,
v_readlane, DPP moves, ballot and s_barrier evaluated with the hardware's lane semantics), through the same pack / view
/ carve / unpack code and the same kernel body as the device path, against the reference's golden outputs.

(The scalar one-thread emulation of rounds 1-2 only walked stand-ins of those sections; it is gone.) Lanes run in three
different orders between rendezvous points: a missing barrier or an undeclared reliance on lockstep execution changes
the result.
Test-only build (tests/emul/); the product library has no CPU path."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from helpers import abi, synth

EMUL_DIR = os.path.join(H.ROOT, "tests", "emul")


@pytest.fixture(scope="module")
def simt():
    so = os.path.join(EMUL_DIR, "libvio_simt.so")
    csrc = os.path.join(H.ROOT, "vins-mobile_amd", "csrc")
    srcs = glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(EMUL_DIR, "simt_backend.cpp"), os.path.join(EMUL_DIR, "simt.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-psabi", "-DVIO_SIMT",
                               "-I" + os.path.join(H.ROOT, "include"), "-I" + csrc, "-I" + EMUL_DIR, "-shared", "-o", so,
                               os.path.join(EMUL_DIR, "simt_backend.cpp")])
    lib = C.CDLL(so)
    lib.simt_solve_window.argtypes = [C.POINTER(abi.VioConfig), C.POINTER(abi.VioWindow), C.POINTER(abi.VioSolveStats),
                                      C.c_int, C.c_int, C.c_int]
    return lib


def run(simt, nthreads, variant, order):
    return lambda cfg, win, st: simt.simt_solve_window(cfg, win, st, nthreads, variant, order)


# (threads per workgroup, variant: -1 what the launcher picks / 0 matrix in global scratch / 2 matrix in LDS with the IMU
# coupling in global scratch -- the layout of windows with many landmarks --, lane order)
MODES = [(512, -1, 0), (256, -1, 1), (256, 0, 2), (256, 2, 3)]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", H.golden_window_names())
def test_device_sections_on_simt_emulator(name, mode, simt):
    cfg, w, d = H.load_golden_window(name)
    if mode[1] == 2 and cfg.window_size > 12:
        pytest.skip("the LDS-matrix layouts end at W = 12 (kPanelTiles tile rows)")
    got, stats = H.solve_with(run(simt, *mode), cfg, w)
    H.check_solution(got, stats, d, tol=1e-6, tol_prior=1e-5)


@pytest.mark.parametrize("W,F,loop,seed", H.ODD_SHAPES)
def test_device_sections_odd_shapes(W, F, loop, seed, simt):
    """Seeded windows of awkward sizes through the NaN-poisoned emulated workgroup against the CPU oracle."""
    cfg = abi.default_config(window_size=W)
    osolve, opre = H.oracle_backend()
    w = synth.make_window(cfg, lambda *a: abi.preintegrate_with(opre, cfg, *a), seed=900 + seed, n_features=F, W=W,
                          with_loop=loop)
    got, gs = H.solve_with(run(simt, 256, -1, seed % 3), cfg, w)
    ref, rs = H.solve_with(osolve, cfg, w)
    assert np.isfinite(got.pose).all() and np.isfinite(got.inv_depth).all()
    assert gs["iterations"] == rs["iterations"] and list(gs["it_flags"]) == list(rs["it_flags"])
    assert H.pose_relerr(got.pose, ref.pose) < 1e-6
    assert H.relerr(got.inv_depth, ref.inv_depth) < 1e-6
    assert got.next_prior.n == ref.next_prior.n


@pytest.mark.parametrize("shares", [2, 4])
@pytest.mark.parametrize("name", ["win_c3_w20", "win_c5_w30_b_prior_loop"])
def test_band_product_shares_cover_every_tile_once(name, shares, simt):
    """The deferred product of the band phase, App -= sum V_k V_k^T (csrc/solver_core.h, band_syrk_share), is dealt out tile by tile over
    the waves of all workgroups of a cooperative window. The emulator has one workgroup: it runs the shares one after the other -- a tile
    that two shares took, or none, changes the factorization and the golden comparison fails."""
    cfg, w, d = H.load_golden_window(name)
    simt.simt_set_syrk_shares(shares)
    try:
        got, stats = H.solve_with(run(simt, 512, -1, 0), cfg, w)
    finally:
        simt.simt_set_syrk_shares(1)
    H.check_solution(got, stats, d, tol=1e-6, tol_prior=1e-5)
