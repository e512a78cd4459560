"""GPU parity tests of the HIP back-end, through the C ABI (vins-mobile_amd/csrc/libvio_amd.so).

Bars: north_star asks for poses and inverse depths within 1e-4 relative of the reference CPU Ceres path. The golden
fixtures ARE that path's outputs (tests/golden/make_golden.py); the HIP path is held to 1e-6 against them and against
the CPU oracle on freshly seeded windows."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers as H
from helpers import abi, synth, pkg

pytestmark = pytest.mark.gpu

TOL = 1e-6
TOL_PRIOR = 1e-5


@pytest.fixture(scope="module")
def solver_cache():
    cache = {}
    yield cache
    for s in cache.values():
        s.close()


def get_solver(cache, cfg, max_batch=64):
    key = (cfg.window_size, cfg.max_features, cfg.max_factors, cfg.max_iterations, cfg.fx, max_batch)
    if key not in cache:
        cache[key] = pkg.backend.WindowSolver(cfg, max_batch=max_batch)
    return cache[key]


@pytest.mark.parametrize("name", H.golden_window_names())
def test_golden_window(name, solver_cache):
    cfg, w, d = H.load_golden_window(name)
    solver = get_solver(solver_cache, cfg)
    got = w.copy()
    stats = solver.solve([got])[0]
    H.check_solution(got, stats, d, tol=TOL, tol_prior=TOL_PRIOR)


def test_ragged_batch_matches_single(solver_cache):
    """All W=10 fixtures in ONE launch (different F, M, priors, loop / no loop) == each solved alone."""
    names = [n for n in H.golden_window_names() if "w4" not in n and "w20" not in n and "w30" not in n]
    cfg = None
    ws, ds = [], []
    for n in names:
        cfg, w, d = H.load_golden_window(n)
        ws.append(w.copy()), ds.append(d)
    solver = get_solver(solver_cache, cfg)
    stats = solver.solve(ws)
    for w, s, d in zip(ws, stats, ds):
        H.check_solution(w, s, d, tol=TOL, tol_prior=TOL_PRIOR)


def test_seeded_windows_match_oracle(solver_cache):
    cfg = abi.default_config()
    osolve, opre = H.oracle_backend()
    pre = lambda *a: pkg.backend.preintegrate(cfg, *a)
    ws = [synth.make_window(cfg, pre, seed=1000 + i, perturb_scale=[1, 1, 4, 10][i % 4]) for i in range(8)]
    # product pre-integration == oracle pre-integration
    w_or = synth.make_window(cfg, lambda *a: abi.preintegrate_with(opre, cfg, *a), seed=1000)
    assert np.abs(w_or.preint - ws[0].preint).max() <= 1e-12 * np.abs(w_or.preint).max()
    solver = get_solver(solver_cache, cfg)
    got = [w.copy() for w in ws]
    stats = solver.solve(got)
    for w, g, s in zip(ws, got, stats):
        ref, rs = H.solve_with(osolve, cfg, w)
        assert H.pose_relerr(g.pose, ref.pose) < TOL
        assert H.relerr(g.speed_bias, ref.speed_bias) < TOL
        assert H.relerr(g.inv_depth, ref.inv_depth) < TOL
        assert s["iterations"] == rs["iterations"] and list(s["it_flags"]) == list(rs["it_flags"])
        assert H.relerr(s["it_cost"], rs["it_cost"]) < 1e-6
        Hr, br, _ = ref.next_prior.canonical()
        Hg, bg, _ = g.next_prior.canonical()
        assert H.relerr(Hg, Hr) < TOL_PRIOR and H.relerr(bg, br) < TOL_PRIOR


def _cpu_solver():
    """The real reference path (vendored Ceres + the VINS factors) when oracle/_ref travelled, else the restatement."""
    ref = H.ref_lib_or_none()
    return abi.bind_backend_solver(ref, "ref")[0] if ref is not None else H.oracle_backend()[0]


def _follow_traces(cfg, w, solver, cpu, what):
    g = w.copy()
    s = solver.solve([g])[0]
    ref, rs = H.solve_with(cpu, cfg, w)
    assert s["iterations"] == rs["iterations"] and list(s["it_flags"]) == list(rs["it_flags"]), what
    assert H.relerr(s["it_cost"], rs["it_cost"]) < 1e-6, what
    assert H.relerr(s["it_radius"], rs["it_radius"]) < 1e-4, what
    # the step-quality ratio rho = cost change / model cost change (the device takes the model change from an identity
    # that assumes an exact Gauss-Newton solve: DESIGN 3.2 "one quadratic form per step")
    n = s["iterations"]
    rho_g, rho_r = np.asarray(s["it_relative_decrease"][:n]), np.asarray(rs["it_relative_decrease"][:n])
    assert np.abs(rho_g - rho_r).max() < 1e-4 * max(1.0, np.abs(rho_r).max()), (what, rho_g, rho_r)
    assert H.pose_relerr(g.pose, ref.pose) < TOL and H.relerr(g.inv_depth, ref.inv_depth) < TOL, what
    return s


def test_step_quality_traces_on_badly_conditioned_windows(solver_cache):
    """Windows on which the model cost change is delicate: (a) next to no parallax (a 20 ms frame interval: depths are
    barely observable, the dogleg mixes Cauchy and Gauss-Newton steps and rejects some), started far off; (b) a prior that
    outweighs the visual cost by 1e4 (its expanded cost form |r0|^2/2 + b0.dx + dx.H0.dx/2 cancels against |r0|^2/2). The
    accept / reject sequence, costs, radii and the rho of every iteration follow the CPU path."""
    cfg = abi.default_config()
    pre = lambda *a: pkg.backend.preintegrate(cfg, *a)
    solver, cpu = get_solver(solver_cache, cfg), _cpu_solver()
    rejected = 0
    for seed in (3100, 3101, 3102):
        w = synth.make_window(cfg, pre, seed=seed, frame_dt=0.02, imu_per_frame=4, perturb_scale=6.0)
        s = _follow_traces(cfg, w, solver, cpu, "low parallax seed %d" % seed)
        rejected += sum(1 for f in s["it_flags"] if f == 1)
    # (b): steady-state window whose prior is scaled up
    a = synth.make_window(cfg, pre, seed=3200, traj_seed=88, frame_offset=0)
    solver.solve([a])
    b = synth.make_window(cfg, pre, seed=3201, traj_seed=88, frame_offset=1)
    b.prior = a.next_prior.copy()
    n = b.prior.n
    b.prior.J[: n * n] *= 100.0
    b.prior.r[:n] *= 100.0
    _follow_traces(cfg, b, solver, cpu, "heavy prior")
    assert rejected >= 1, "none of the low-parallax windows rejected a step: pick harder ones"


def test_prior_chain_on_device(solver_cache):
    """The prior the device builds is consumed by the next device solve (MARGIN_OLD chain); compared with the
    oracle running the same chain on its own priors."""
    cfg = abi.default_config()
    osolve, _ = H.oracle_backend()
    pre = lambda *a: pkg.backend.preintegrate(cfg, *a)
    solver = get_solver(solver_cache, cfg)
    prior_g = prior_o = None
    for k in range(3):
        w = synth.make_window(cfg, pre, seed=500 + k, traj_seed=77, frame_offset=k)
        wg, wo = w.copy(), w.copy()
        wg.prior, wo.prior = prior_g, prior_o
        solver.solve([wg])
        ref, _ = H.solve_with(osolve, cfg, wo)
        assert H.pose_relerr(wg.pose, ref.pose) < 1e-5
        assert H.relerr(wg.inv_depth, ref.inv_depth) < 1e-5
        prior_g, prior_o = wg.next_prior.copy(), ref.next_prior.copy()


def test_prior_chain_resident_in_device_memory():
    """vio_backend_reserve_priors: a chain whose priors never leave the device gives what the chain through host memory
    gives. One batch carries both kinds; a SECOND_NEW step that leaves the prior alone (n = -1) keeps the slot."""
    cfg = abi.default_config()
    pre = lambda *a: pkg.backend.preintegrate(cfg, *a)
    solver = pkg.backend.WindowSolver(cfg, max_batch=4)
    solver.reserve_priors(3)
    flags = [abi.VIO_MARGIN_OLD, abi.VIO_MARGIN_OLD, abi.VIO_MARGIN_SECOND_NEW, abi.VIO_MARGIN_OLD, abi.VIO_MARGIN_OLD]
    prior_h = prior_r = None
    for k, flag in enumerate(flags):
        w = synth.make_window(cfg, pre, seed=700 + k, traj_seed=78, frame_offset=k)
        w.marginalization_flag = flag
        wh, wr = w.copy(), w.copy()
        wh.prior, wr.prior = prior_h, prior_r
        wr.resident_prior = 2 + 1  # slot 2
        stats = solver.solve([wh, wr])
        assert stats[0]["iterations"] == stats[1]["iterations"]
        # (the two chains run in different workgroups: LDS atomics sum in a different order, the chains drift apart by rounding)
        assert np.abs(wh.pose - wr.pose).max() < 1e-7 and np.abs(wh.inv_depth - wr.inv_depth).max() < 1e-7
        nh, nr = wh.next_prior.c, wr.next_prior.c
        assert nh.n == nr.n and nh.n_blocks == nr.n_blocks
        assert list(nh.block_index[: nh.n_blocks]) == list(nr.block_index[: nr.n_blocks])
        assert list(nh.block_offset[: nh.n_blocks]) == list(nr.block_offset[: nr.n_blocks])
        if flag == abi.VIO_MARGIN_SECOND_NEW and nh.n < 0:
            continue  # both chains keep their prior
        assert nh.n > 0
        assert not wr.next_prior.J.any()  # the data stayed on the device
        prior_h, prior_r = wh.next_prior.copy(), wr.next_prior.header_only()
    # errors: a slot out of range, a slot named twice, a header-only prior without a slot
    w = synth.make_window(cfg, pre, seed=1, traj_seed=78)
    a, b = w.copy(), w.copy()
    a.resident_prior = 4
    with pytest.raises(RuntimeError):
        solver.solve([a])
    a.resident_prior = b.resident_prior = 1
    with pytest.raises(RuntimeError):
        solver.solve([a, b])
    a.resident_prior, a.prior = 0, prior_r
    with pytest.raises(RuntimeError):
        solver.solve([a])
    solver.close()


def test_determinism_and_resident_api(solver_cache):
    cfg, w, d = H.load_golden_window("win_c2_easy")
    solver = get_solver(solver_cache, cfg)
    ws = [w.copy() for _ in range(4)]
    solver.kernel_ms()  # drain timings of earlier launches
    solver.upload(ws)
    solver.launch()
    solver.launch()  # relaunch from the same resident inputs
    solver.sync()
    stats = solver.download(ws)
    for x, s in zip(ws, stats):
        H.check_solution(x, s, d, tol=TOL, tol_prior=TOL_PRIOR)
        assert np.abs(x.pose - ws[0].pose).max() < 1e-9
    ms, n = solver.kernel_ms()
    assert n == 2 and ms > 0


def test_error_codes(solver_cache):
    cfg, w, _ = H.load_golden_window("win_small_w4")
    solver = get_solver(solver_cache, cfg)
    bad = w.copy()
    bad.factor_feature[0] = bad.n_features
    arr = (abi.VioWindow * 1)()
    bad.fill_struct(arr[0])
    lib = abi.load_product()
    assert lib.vio_backend_solve_windows(solver._h, arr, 1, 0, None) == abi.VIO_EINVAL
    big_cfg, big, _ = H.load_golden_window("win_c2_easy")
    arr2 = (abi.VioWindow * 1)()
    big.fill_struct(arr2[0])
    assert lib.vio_backend_solve_windows(solver._h, arr2, 1, 0, None) == abi.VIO_ECAP  # W=10 into a W=4 context
    # a failed upload leaves nothing behind that could be launched or downloaded
    assert lib.vio_backend_launch(solver._h, None) == abi.VIO_ESTATE
    assert lib.vio_backend_download(solver._h, arr2, 1, None) == abi.VIO_ESTATE


def test_poisoned_device_buffers():
    """VIO_AMD_POISON=1 fills the whole LDS of every CU and every device scratch / output buffer with NaN patterns before
    each launch: anything the kernel reads without having written it (LDS and hipMalloc'd scratch keep whatever the
    previous kernel left) turns the solve into NaNs instead of passing by luck. Own process: the switch is read once."""
    import subprocess, sys
    env = dict(os.environ, VIO_AMD_POISON="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "not poisoned and not error_codes", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_one_batch_split_over_both_kernel_variants(solver_cache):
    """Windows that fit the CU's LDS and windows that do not (260 landmarks, a relocalization pose) in ONE batch: the
    launch is split by variant, every window still matches the CPU oracle solved alone."""
    cfg = abi.default_config()
    osolve, opre = H.oracle_backend()
    pre = lambda *a: abi.preintegrate_with(opre, cfg, *a)
    shapes = [(150, 0), (260, 0), (40, 0), (150, 12), (200, 0), (255, 8), (7, 0), (236, 0)]
    ws = [synth.make_window(cfg, pre, seed=700 + i, n_features=F, W=10, with_loop=loop) for i, (F, loop) in enumerate(shapes)]
    solver = get_solver(solver_cache, cfg)
    got = [w.copy() for w in ws]
    stats = solver.solve(got)
    for w, g, s in zip(ws, got, stats):
        ref, rs = H.solve_with(osolve, cfg, w)
        assert s["iterations"] == rs["iterations"] and list(s["it_flags"]) == list(rs["it_flags"])
        assert H.pose_relerr(g.pose, ref.pose) < TOL and H.relerr(g.inv_depth, ref.inv_depth) < TOL
        assert g.next_prior.n == ref.next_prior.n


@pytest.mark.parametrize("W,F,loop,seed", H.ODD_SHAPES)
def test_odd_shapes_match_oracle(W, F, loop, seed):
    """Awkward window sizes (1..260 landmarks, W = 3..13, loop pose) through the device kernel against the CPU oracle; the
    poisoned-buffer run below repeats them with NaN-filled LDS and scratch."""
    cfg = abi.default_config(window_size=W)
    osolve, opre = H.oracle_backend()
    w = synth.make_window(cfg, lambda *a: abi.preintegrate_with(opre, cfg, *a), seed=900 + seed, n_features=F, W=W,
                          with_loop=loop)
    solver = pkg.backend.WindowSolver(cfg, max_batch=1)
    got = w.copy()
    gs = solver.solve([got])[0]
    solver.close()
    ref, rs = H.solve_with(osolve, cfg, w)
    assert np.isfinite(got.pose).all() and np.isfinite(got.inv_depth).all()
    assert gs["iterations"] == rs["iterations"] and list(gs["it_flags"]) == list(rs["it_flags"])
    assert H.pose_relerr(got.pose, ref.pose) < TOL and H.relerr(got.inv_depth, ref.inv_depth) < TOL
    assert got.next_prior.n == ref.next_prior.n
    if ref.next_prior.n > 0:
        Hr, br, _ = ref.next_prior.canonical()
        Hg, bg, _ = got.next_prior.canonical()
        # with a handful of landmarks the block that is marginalized out is close to singular (the eps cut of
        # marginalization_factor.cpp:268-276 is what keeps it finite): the prior is then determined to ~1e-4 only and the
        # order of the kernel's atomic sums shows (the solve itself is held to TOL above)
        tol = TOL_PRIOR if F >= 10 else 2e-3
        assert np.abs(Hg - Hr).max() <= tol * np.abs(Hr).max() + 1e-6
        lam, V = np.linalg.eigh(Hr)
        keep = V[:, lam > 1e-6 * lam.max()]       # b = J^T r on the well-determined directions
        assert np.abs(keep.T @ (bg - br)).max() <= tol * np.abs(br).max() + 1e-6


@pytest.mark.parametrize("name", ["win_c3_w20", "win_c5_w30_a", "win_c5_w30_b_prior_loop"])
def test_cooperative_windows_match_the_reference(name, monkeypatch):
    """Large windows (pose matrix in global scratch) solved by 1, 2 and 4 workgroups per window (csrc/solver_core.h, cooperative
    windows: projection-factor chunks and Schur tile pairs shared, inputs and partial sums through the window's scratch): every
    width against the reference's golden output, with the reference's accept / reject trace, alone and in a batch whose grid is
    padded to whole groups of eight windows."""
    cfg, w, d = H.load_golden_window(name)
    ref_pose = None
    for width in (1, 2, 4):
        monkeypatch.setenv("VIO_AMD_COOP", str(width))
        solver = pkg.backend.WindowSolver(cfg, max_batch=16)
        got = w.copy()
        stats = solver.solve([got])[0]
        H.check_solution(got, stats, d, tol=TOL, tol_prior=TOL_PRIOR)
        batch = [w.copy() for _ in range(9)]   # two groups of eight, seven padded blocks per member row
        bstats = solver.solve(batch)
        for g, s in zip(batch, bstats):
            assert s["iterations"] == stats["iterations"] and list(s["it_flags"]) == list(stats["it_flags"])
            assert H.pose_relerr(g.pose, got.pose) < 1e-9 and H.relerr(g.inv_depth, got.inv_depth) < 1e-9
        solver.close()
        if ref_pose is None:
            ref_pose = np.asarray(got.pose).copy()
        else:  # the widths differ in summation order only
            assert H.pose_relerr(got.pose, ref_pose) < 1e-8, width



def test_cooperative_window_timeout_is_reported(monkeypatch):
    """The wait between the workgroups of a cooperative window gives up instead of hanging the device (csrc/solver_core.h,
    coop_spin): with the helpers told to stay away (VIO_AMD_COOP_FAULT, a test hook of the launcher) and a short poll budget the
    owner's first wait times out, every later wait of the window returns at once, the window reads FAILURE (termination 2), its
    next prior is dropped and the call returns VIO_ETIMEOUT; the same context then solves the window correctly again."""
    cfg, w, d = H.load_golden_window("win_c3_w20")
    monkeypatch.setenv("VIO_AMD_COOP", "2")
    monkeypatch.setenv("VIO_AMD_COOP_FAULT", "1")
    monkeypatch.setenv("VIO_AMD_COOP_SPIN", "2000")
    solver = pkg.backend.WindowSolver(cfg, max_batch=4)
    got = [w.copy(), w.copy()]
    try:
        solver.solve(got)
        raised = None
    except RuntimeError as e:
        raised = str(e)
    assert raised is not None and ("rc=%d" % abi.VIO_ETIMEOUT) in raised, raised
    monkeypatch.delenv("VIO_AMD_COOP_FAULT")
    monkeypatch.delenv("VIO_AMD_COOP_SPIN")
    again = w.copy()
    stats = solver.solve([again])[0]
    H.check_solution(again, stats, d, tol=TOL, tol_prior=TOL_PRIOR)
    solver.close()
