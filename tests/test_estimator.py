"""The native estimator (vio_estimator_*, csrc/vio_estimator.cpp = class VINS after initialisation: processIMU,
processImage, the host half of solve_ceres, failureDetection, slideWindow, clearState; VINS_ios/VINS.cpp:36-478,
1149-1273).

CPU: the parts that never reach the solver (IMU propagation, window filling, the INITIAL-phase slides, resets) against
numpy restatements and the stand-alone landmark store. GPU: the full loop against the python restatement of the same
loop (tools/replay_synthetic.py::ClosedLoop) running the CPU ORACLE solver, plus batching, failure recovery and the
relocalization (loop) bookkeeping."""
import os
import sys

import numpy as np
import pytest

import helpers as H
from helpers import abi, pkg, synth

sys.path.insert(0, os.path.join(H.ROOT, "tools"))
import replay_synthetic as RS  # noqa: E402
from test_closed_loop import oracle_loop  # noqa: E402

TIC = np.array([0.0, 0.065, 0.0])
RIC = np.diag([1.0, -1.0, -1.0])


def obs_grid(n, shift=0.0, first_id=0):
    ids = list(range(first_id, first_id + n))
    xyz = [[-0.4 + 0.8 * (i % 12) / 11 + shift, -0.3 + 0.6 * (i // 12) / 11, 1.0] for i in range(n)]
    return ids, xyz


def test_process_imu_propagates_like_the_reference_formula():
    cfg = abi.default_config()
    est = pkg.estimator.Estimator(cfg, TIC, RIC)
    rng = np.random.default_rng(0)
    est.process_imu(0.01, [0.1, 0.2, 9.7], [0.01, -0.02, 0.03])     # first sample: only acc_0 / gyr_0
    assert est.process_image(*obs_grid(60), 1.0).action == abi.VIO_FRAME_FILLING
    a0, w0 = np.array([0.1, 0.2, 9.7]), np.array([0.01, -0.02, 0.03])
    P, R, V, g = np.zeros(3), np.eye(3), np.zeros(3), np.array([0, 0, cfg.gravity])
    dts, accs, gyrs = [], [], []
    for _ in range(25):
        dt = rng.uniform(0.005, 0.02)
        a1, w1 = rng.normal(0, 1, 3) + [0, 0, 9.8], rng.normal(0, 0.5, 3)
        est.process_imu(dt, a1, w1)
        ua0 = R @ a0 - g                                              # VINS.cpp:360-369
        R = R @ synth.quat_to_rot(np.array([*(0.5 * (w0 + w1) * dt / 2), 1.0]), normalize=False)
        ua = 0.5 * (ua0 + R @ a1 - g)
        P, V = P + dt * V + 0.5 * dt * dt * ua, V + dt * ua
        a0, w0 = a1, w1
        dts.append(dt), accs.append(a1), gyrs.append(w1)
    w = est.window()
    assert np.abs(w["Ps"][1] - P).max() < 1e-12 and np.abs(w["Vs"][1] - V).max() < 1e-12
    assert np.abs(w["Rs"][1] - R).max() < 1e-12
    assert np.abs(w["Ps"][0]).max() == 0 and est.status().frame_count == 1
    est.close()


def test_imu_batch_equals_per_sample_calls():
    cfg = abi.default_config(window_size=4)
    rng = np.random.default_rng(3)
    a, b = pkg.estimator.Estimator(cfg, TIC, RIC, n_seq=5), pkg.estimator.Estimator(cfg, TIC, RIC, n_seq=5)
    for frame in range(3):
        ns = rng.integers(0, 9, 5)
        ns[0] = 8
        dt, acc, gyr = rng.uniform(0.005, 0.02, (5, 8)), rng.normal(0, 2, (5, 8, 3)), rng.normal(0, 0.5, (5, 8, 3))
        a.process_imu_batch(ns, dt, acc, gyr)
        for q in range(5):
            for i in range(ns[q]):
                b.process_imu(dt[q, i], acc[q, i], gyr[q, i], seq=q)
        obs = [obs_grid(40 + q, shift=0.01 * frame) for q in range(5)]
        ra, rb = a.process_images(obs, [float(frame)] * 5), b.process_images(obs, [float(frame)] * 5)
        assert [r.action for r in ra] == [r.action for r in rb]
    for q in range(5):
        wa, wb = a.window(q), b.window(q)
        assert all(np.array_equal(wa[k], wb[k]) for k in wa)
    a.close(), b.close()


def test_window_fills_then_slides_while_waiting_for_the_initial_state():
    cfg = abi.default_config(window_size=5)
    W = cfg.window_size
    est = pkg.estimator.Estimator(cfg, TIC, RIC)
    fm = pkg.window.FeatureManager(W)                 # the stand-alone store driven the way processImage drives it
    actions, fc = [], 0
    for k in range(W + 4):
        est.process_imu(0.01, [0, 0, 9.8], [0, 0, 0.1])
        ids, xyz = obs_grid(80, shift=0.03 * k)       # large parallax: every frame is a keyframe
        res = est.process_image(ids, xyz, 10.0 + k)
        actions.append(res.action)
        enough, _, _ = fm.add_check_parallax(fc, ids, xyz)
        assert res.marginalization_flag == (abi.VIO_MARGIN_OLD if enough else abi.VIO_MARGIN_SECOND_NEW)
        if fc < W:
            fc += 1
        elif enough:
            fm.remove_back()                          # INITIAL phase: no depth shift (VINS.cpp:1255-1271)
        else:
            fm.remove_front(fc)
    assert actions == [abi.VIO_FRAME_FILLING] * W + [abi.VIO_FRAME_WAIT_INIT] * 4
    st = est.status()
    assert st.frame_count == W and st.solver_flag == abi.VIO_SOLVER_INITIAL and st.prior_rows == 0
    hdr = est.window()["headers"]
    assert list(hdr[:W]) == [10.0 + k for k in range(4, W + 4)]      # four slides dropped the four oldest frames
    a, b = est.features().dump(), fm.dump()
    assert a[0].shape == b[0].shape and len(a[0]) > 50 and np.array_equal(a[0][:, :3], b[0][:, :3])   # id, start, n_obs
    assert np.array_equal(a[1], b[1])
    est.close(), fm.close()


def test_reset_when_the_full_initial_window_tracks_too_little():
    cfg = abi.default_config(window_size=4)
    est = pkg.estimator.Estimator(cfg, TIC, RIC)
    for k in range(4):
        est.process_imu(0.01, [0, 0, 9.8], [0, 0, 0])
        assert est.process_image(*obs_grid(50, shift=0.02 * k), float(k)).action == abi.VIO_FRAME_FILLING
    res = est.process_image(*obs_grid(50, first_id=1000), 4.0)        # nothing tracked: track_num < 20
    assert res.action == abi.VIO_FRAME_RESET and res.track_num == 0
    st = est.status()
    assert st.frame_count == 0 and est.features().count() == 0
    assert np.abs(est.window()["Ps"]).max() == 0
    est.close()


def test_capacity_error_restarts_the_sequence_and_is_reported_not_raised():
    """A window with more landmarks than cfg.max_features cannot be assembled (VIO_ECAP in the host half of solve_ceres,
    before any device call): that sequence's result says FRAME_ERROR / VIO_ECAP and the library has restarted it, the other
    sequence of the call went on normally, the python mirror warns, keeps the code in last_error and only raises with
    strict=True."""
    cfg = abi.default_config(window_size=4, max_features=30)
    W = cfg.window_size
    est = pkg.estimator.Estimator(cfg, TIC, RIC, n_seq=2)
    for k in range(W):
        for q in range(2):
            est.process_imu(0.01, [0, 0, 9.8], [0, 0, 0], seq=q)
        res = est.process_images([obs_grid(50, shift=0.03 * k), obs_grid(25, shift=0.03 * k)], [float(k)] * 2, active=[1, int(k >= 2)])
        assert res[0].action == abi.VIO_FRAME_FILLING and res[1].action == (abi.VIO_FRAME_FILLING if k >= 2 else abi.VIO_FRAME_SKIPPED)
    P = W + 1
    est.set_initial_state([float(k) for k in range(P)], np.zeros((P, 3)), np.tile(np.eye(3), (P, 1, 1)), np.zeros((P, 3)),
                          np.zeros((P, 3)), np.zeros((P, 3)), seq=0)
    for q in range(2):
        est.process_imu(0.01, [0, 0, 9.8], [0, 0, 0], seq=q)
    frame = [obs_grid(50, shift=0.03 * W), obs_grid(25, shift=0.03 * W)]
    with pytest.warns(RuntimeWarning, match="restarted"):
        res = est.process_images(frame, [float(W)] * 2)
    assert res[0].action == abi.VIO_FRAME_ERROR and res[0].error == abi.VIO_ECAP
    assert res[1].action == abi.VIO_FRAME_FILLING and res[1].error == 0      # the other sequence is untouched
    assert est.last_error == abi.VIO_ECAP
    st = est.status(0)
    assert st.frame_count == 0 and st.solver_flag == abi.VIO_SOLVER_INITIAL and est.features(0).count() == 0   # clearState
    assert est.status(1).frame_count == 3
    # the restarted sequence fills its window again from the next frame on
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        res = est.process_images([obs_grid(20, shift=0.0), obs_grid(25, shift=0.03 * (W + 1))], [float(W + 1)] * 2)
    assert res[0].action == abi.VIO_FRAME_FILLING and est.status(0).frame_count == 1 and est.last_error == 0
    est.close()
    # strict=True: the same situation raises
    est = pkg.estimator.Estimator(cfg, TIC, RIC, n_seq=1)
    for k in range(W):
        est.process_imu(0.01, [0, 0, 9.8], [0, 0, 0])
        est.process_images([obs_grid(50, shift=0.03 * k)], [float(k)])
    est.set_initial_state([float(k) for k in range(P)], np.zeros((P, 3)), np.tile(np.eye(3), (P, 1, 1)), np.zeros((P, 3)),
                          np.zeros((P, 3)), np.zeros((P, 3)))
    est.process_imu(0.01, [0, 0, 9.8], [0, 0, 0])
    with pytest.raises(RuntimeError, match="restarted"):
        est.process_images([obs_grid(50, shift=0.03 * W)], [float(W)], strict=True)
    est.close()


def test_argument_errors():
    cfg = abi.default_config()
    lib = abi.load_product()
    import ctypes as C
    h = C.c_void_p()
    assert lib.vio_estimator_create(C.byref(cfg), 0, None, None, C.byref(h)) == abi.VIO_EINVAL
    est = pkg.estimator.Estimator(cfg, TIC, RIC, n_seq=2)
    assert lib.vio_estimator_clear(est._h, 2) == abi.VIO_EINVAL
    res = abi.VioFrameResult()
    assert lib.vio_estimator_process_image(est._h, 5, None, 0, 0.0, C.byref(res)) == abi.VIO_EINVAL
    est.close()


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("host_priors", [False, True])
def test_estimator_matches_the_python_loop_with_the_oracle_solver(host_priors, monkeypatch):
    """host_priors False: the marginalization priors stay in the back-end's device store from frame to frame (the
    default); True: VIO_AMD_HOST_PRIORS=1, they travel through host memory. The oracle loop always carries its own."""
    monkeypatch.setenv("VIO_AMD_HOST_PRIORS", "1" if host_priors else "0")
    cfg = abi.default_config()
    prod = RS.EstimatorLoop(cfg, seed=5, init_noise=1.0)
    ref = oracle_loop(cfg, seed=5)
    for _ in range(60):
        prod.step(), ref.step()
    assert len(prod.history) == len(ref.history) == 60 - cfg.window_size
    dp = np.array([a[1] - b[1] for a, b in zip(prod.history, ref.history)])
    assert np.abs(dp).max() < 1e-5, np.abs(dp).max()
    assert [h[3].stats.iterations for h in prod.history] == [h[3]["iterations"] for h in ref.history]
    e = prod.errors()
    assert np.sqrt((e ** 2).mean()) < 0.06 and e.max() < 0.15
    st = prod.est.status()
    assert st.solver_flag == abi.VIO_SOLVER_NON_LINEAR and st.prior_rows == ref.prior.n
    assert prod.est.features().count() == ref.fm.count()
    prod.close(), ref.close()


@pytest.mark.gpu
def test_batched_sequences_equal_single_sequences():
    """Three sequences in one estimator (their window solves share a launch, they reach the solve phase on different
    frames) give what three single-sequence estimators give."""
    cfg = abi.default_config(window_size=6)
    W = cfg.window_size
    worlds = [RS.SyntheticWorld(cfg, 20 + q) for q in range(3)]
    singles = [RS.EstimatorLoop(cfg, seed=20 + q, init_noise=1.0) for q in range(3)]
    for s in singles:
        for _ in range(24):
            s.step()
    est = pkg.estimator.Estimator(cfg, worlds[0].tic, worlds[0].ric, n_seq=3)
    feeders = [RS.EstimatorLoop(cfg, seed=20 + q, init_noise=1.0, world=worlds[q]) for q in range(3)]
    start = [0, 3, 5]                                   # sequence q starts `start[q]` calls late
    got = [[] for _ in range(3)]
    for call in range(24 + max(start)):
        obs, hdr, act = [], [], []
        for q in range(3):
            k = call - start[q]
            f = feeders[q]
            if k < 0 or k >= 24:
                obs.append(([], [])), hdr.append(0.0), act.append(0)
                continue
            f.est.close()
            f.est = _SeqView(est, q)                    # the feeder's IMU / init calls go to sequence q of the shared estimator
            obs.append(f.feed_until_image()), hdr.append(worlds[q].time(k)), act.append(1)
        res = est.process_images(obs, hdr, act)
        for q in range(3):
            if res[q].action == abi.VIO_FRAME_SOLVED:
                got[q].append(est.window(q)["Ps"][W].copy())
    for q in range(3):
        want = np.array([h[1] for h in singles[q].history])
        assert len(got[q]) == len(want) and np.abs(np.array(got[q]) - want).max() < 1e-6   # (LDS atomics: sums are not bit-reproducible)
    est.close()
    for s in singles:
        s.close()


class _SeqView:
    """Routes a feeder's single-sequence calls to sequence q of a shared estimator."""

    def __init__(self, est, q):
        self.est, self.q = est, q

    def process_imu(self, dt, a, w):
        self.est.process_imu(dt, a, w, seq=self.q)

    def set_initial_state(self, *a):
        self.est.set_initial_state(*a, seq=self.q)

    def close(self):
        pass


@pytest.mark.gpu
def test_failure_detection_clears_and_the_next_window_continues_the_track():
    cfg = abi.default_config(window_size=6)
    W = cfg.window_size
    loop = RS.EstimatorLoop(cfg, seed=9, init_noise=1.0)
    for _ in range(20):
        loop.step()
    last = loop.est.window()
    assert loop.history[-1][3].action == abi.VIO_FRAME_SOLVED
    # a frame that tracks nothing: f_manager.last_track_num < 4 -> failure -> clearState (VINS.cpp:216-220, 462-467)
    k = loop.k
    for a, w in loop.world.imu_interval(k):
        loop.est.process_imu(loop.world.dt, a, w)
    res = loop.est.process_image(*obs_grid(60, first_id=10 ** 6), loop.world.time(k))
    assert res.action == abi.VIO_FRAME_FAILURE and res.failure_reasons & abi.VIO_FAIL_FEW_FEATURES
    st = loop.est.status()
    assert st.failure_occur == 1 and st.frame_count == 0 and st.solver_flag == abi.VIO_SOLVER_INITIAL and st.prior_rows == 0
    # refill a window and hand over initial states in ANOTHER gauge (rotated about z, shifted): new2old must anchor the
    # first solved window where the failed one stood (last_P_old / yaw of last_R_old), VINS.cpp:139-144
    yaw = synth.rotvec_to_rot(np.array([0, 0, 0.7]))
    shift = np.array([3.0, -2.0, 0.0])
    world = loop.world
    init = []
    loop.world.tracked.clear()
    for j in range(W + 1):
        kk = k + 1 + j
        for a, w in world.imu_interval(kk):
            loop.est.process_imu(world.dt, a, w)
        P, R, V = world.truth(kk)
        init.append((world.time(kk), yaw @ P + shift, yaw @ R, yaw @ V))
        if j == W:
            loop.est.set_initial_state([i[0] for i in init], [i[1] for i in init], [i[2] for i in init], [i[3] for i in init],
                                       [world.ba] * (W + 1), [world.bg] * (W + 1))
        res = loop.est.process_image(*world.observe(kk), world.time(kk))
    assert res.action == abi.VIO_FRAME_SOLVED
    st = loop.est.status()
    assert st.failure_occur == 0 and st.solver_flag == abi.VIO_SOLVER_NON_LINEAR
    # the oldest frame of the solved window sat at last_P_old (it left with the slide; the frame after it is one
    # frame of motion away) and the yaw gauge is the old one: the window is back in the ORIGINAL frame, not the rotated one
    w = loop.est.window()
    P_true = np.array([world.truth(k + 2 + j)[0] for j in range(W)])
    d = w["Ps"][:W] - P_true
    assert np.abs(d - d.mean(0)).max() < 0.2            # same orientation gauge as before the failure (no 0.7 rad yaw)
    assert np.linalg.norm(w["Ps"][0] - last["Ps"][0]) < 2.0
    loop.close()


@pytest.mark.gpu
def test_relocalization_adds_loop_factors_and_reports_the_drift():
    cfg = abi.default_config()
    W = cfg.window_size
    loop = RS.EstimatorLoop(cfg, seed=5, init_noise=1.0)
    loop.est.set_resident(True)                         # (the default; said here so that the checks below hold under VIO_AMD_RESIDENT=0)
    for _ in range(25):
        loop.step()
    assert loop.history and loop.history[-1][0] == 24
    est, world = loop.est, loop.world
    win = est.window()
    i = 4                                               # the window frame the loop detector matched
    assert est.status().resident == 1                   # (a solved NON_LINEAR sequence keeps its landmark list on the device)
    info, pts = est.features().dump()                   # ... and a look at the list brings it back to the host
    assert est.status().resident == 0
    # the old keyframe saw the landmarks of frame i from the (true) pose of frame i, but its own map places that pose
    # 1.5 m away: the drift the relocalization has to report
    k_i = int(round((win["headers"][i] - world.time(0)) / world.frame_dt))
    assert abs(world.time(k_i) - win["headers"][i]) < 1e-9
    ids, xy, off = [], [], 0
    for fid, start, n_obs in info[:, :3].astype(int):
        if start <= i <= start + n_obs - 1 and n_obs >= 2 and start < W - 2:
            p = pts[off + (i - start)]
            ids.append(fid), xy.append([p[0], p[1]])
        off += n_obs
    assert len(ids) > 30
    P_i, R_i = win["Ps"][i], win["Rs"][i]
    drift = np.array([1.5, -0.5, 0.0])
    est.set_relocalization(win["headers"][i], P_i + drift, synth.rot_to_quat(R_i), ids, xy)
    res = loop.step()
    assert res.action == abi.VIO_FRAME_SOLVED and res.n_loop_factors == len(ids)
    st = est.status()
    assert st.resident == 1                             # (solved on the host-side list -- the look above had fetched it --, then back on the device)
    assert np.abs(np.array(st.r_drift).reshape(3, 3) - np.eye(3)).max() < 2e-2
    assert np.abs(np.array(st.t_drift) - drift).max() < 0.1, st.t_drift[:]
    assert np.abs(np.array(st.relative_t)).max() < 0.1 and abs(st.relative_yaw) < 1.0   # loop pose ~ frame i itself
    cP, cR = est.corrected_window()
    w2 = est.window()
    assert np.abs(cP - (w2["Ps"] @ np.array(st.r_drift).reshape(3, 3).T + np.array(st.t_drift))).max() < 1e-12
    # the constraint stays while its frame is in the window, then drops out
    n_loop = []
    for _ in range(80):
        n_loop.append(loop.step().n_loop_factors)
        if n_loop[-1] == 0:
            break
    assert n_loop[0] > 0 and n_loop[-1] == 0 and all(a >= b for a, b in zip(n_loop, n_loop[1:]))
    assert est.window()["headers"][0] > win["headers"][i]          # ... which happened because its frame left the window
    assert est.status().resident == 1                   # the loop factors of the later frames were paired on the device
    loop.close()


@pytest.mark.gpu
def test_bad_initial_state_fails_the_cost_check_and_returns_to_initial():
    """The branch after solveInitial: a first solve that ends above final_cost 200 drops the prior and goes back to
    INITIAL (VINS.cpp:415-424); a good hand-over afterwards succeeds."""
    cfg = abi.default_config(window_size=6)
    W = cfg.window_size
    loop = RS.EstimatorLoop(cfg, seed=13, init_noise=0.0)
    world, est = loop.world, loop.est
    for k in range(W + 1):
        ids, xyz = loop.feed_until_image()
        if k == W:   # overwrite the pending hand-over with garbage: identity attitudes on a straight line
            P = W + 1
            est.set_initial_state([world.time(j) for j in range(P)], [[0.3 * j, 0, 0] for j in range(P)], [np.eye(3)] * P,
                                  [[3.0, 0, 0]] * P, [[0, 0, 0]] * P, [[0, 0, 0]] * P)
        res = est.process_image(ids, xyz, world.time(k))
    assert res.action == abi.VIO_FRAME_INIT_FAILED and res.stats.final_cost > 200
    st = est.status()
    assert st.solver_flag == abi.VIO_SOLVER_INITIAL and st.prior_rows == 0 and st.frame_count == W
    # next frame: the window holds frames 1..W+1 (or 0..W-1,W+1 after a non-keyframe slide); hand over the truth
    k = loop.k
    ids, xyz = loop.feed_until_image()
    hdr = est.window()["headers"].copy()
    hdr[W] = world.time(k)
    ks = [int(round((h - world.time(0)) / world.frame_dt)) for h in hdr]
    tr = [world.truth(j) for j in ks]
    est.set_initial_state(hdr, [t[0] for t in tr], [t[1] for t in tr], [t[2] for t in tr], [world.ba] * (W + 1), [world.bg] * (W + 1))
    res = est.process_image(ids, xyz, world.time(k))
    assert res.action == abi.VIO_FRAME_SOLVED and res.stats.final_cost < 200
    assert est.status().solver_flag == abi.VIO_SOLVER_NON_LINEAR
    assert np.abs(est.window()["Ps"][W] - world.truth(k)[0]).max() < 0.1
    loop.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 3, 4])
def test_estimator_initialises_itself(seed):
    """No hand-over: solveInitial inside the estimator (relative pose, global SfM + BA, PnP, visual-inertial alignment,
    VINS.cpp:833-1104) starts the track at the first full window; positions compared after aligning the estimator's own
    gravity-aligned frame (yaw, origin) with the scene's."""
    cfg = abi.default_config()
    W = cfg.window_size
    loop = RS.EstimatorLoop(cfg, seed=seed, self_init=True)
    acts = [loop.step().action for _ in range(60)]
    assert acts[:W] == [abi.VIO_FRAME_FILLING] * W
    assert acts[W:].count(abi.VIO_FRAME_SOLVED) >= 48 and abi.VIO_FRAME_FAILURE not in acts
    e = loop.errors()
    assert np.sqrt((e ** 2).mean()) < 0.1 and e.max() < 0.2, (np.sqrt((e ** 2).mean()), e.max())
    w = loop.est.window()
    k = loop.history[-1][0]
    # metric scale and gravity direction came out right: speed and the vertical velocity component match the truth
    v_true = loop.world.truth(k)[2]
    assert abs(np.linalg.norm(w["Vs"][W]) - np.linalg.norm(v_true)) < 0.1 and abs(w["Vs"][W][2] - v_true[2]) < 0.1
    assert np.abs(w["Bgs"][W] - loop.world.bg).max() < 5e-3
    loop.close()


@pytest.mark.gpu
def test_two_estimators_driven_from_two_host_threads():
    """The deployment DESIGN §5 recommends (one estimator's host phases overlap the other's kernel): contexts are
    independent, the shared host pool serves one parallel region at a time and the other caller runs inline."""
    import threading
    cfg = abi.default_config()

    def run(seed, out):
        lp = RS.EstimatorLoop(cfg, seed=seed, init_noise=1.0)
        for _ in range(40):
            lp.step()
        out[seed] = np.array([h[1] for h in lp.history])
        lp.close()

    ref, got = {}, {}
    run(5, ref), run(6, ref)
    ts = [threading.Thread(target=run, args=(s, got)) for s in (5, 6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for s in (5, 6):
        assert got[s].shape == ref[s].shape and np.abs(got[s] - ref[s]).max() < 1e-6


# ---- device-resident landmark stores (vio_estimator_set_resident) ---------------------------------------------------------
def _dump_lists(est, seq=0):
    rec, pts = est.features(seq).dump()    # rows: id, start_frame, n_obs, used_num, solve_flag, is_outlier, fixed, depth
    return rec, pts


@pytest.mark.gpu
@pytest.mark.parametrize("W,seed,device_imu", [(10, 5, False), (6, 5, False), (10, 21, True), (5, 33, False), (8, 8, True)])
def test_resident_sequences_give_what_the_host_side_list_gives(W, seed, device_imu, monkeypatch):
    """The same replay twice: landmark list, window assembly and slides on the host (vio_window.cpp) / on the device
    (store_core.h). Same keyframe decisions, same iteration counts, same landmark list at the end (ids, start frames,
    observation counts, flags exactly; depths and positions to the reproducibility of the window kernel's sums)."""
    monkeypatch.setenv("VIO_AMD_RESIDENT_IMU", "1" if device_imu else "0")   # (read when an estimator is created)
    cfg = abi.default_config(window_size=W)
    host = RS.EstimatorLoop(cfg, seed=seed, init_noise=1.0)
    dev = RS.EstimatorLoop(cfg, seed=seed, init_noise=1.0)
    host.est.set_resident(False)
    dev.est.set_resident(True)
    n = 60
    for _ in range(n):
        a, b = host.step(), dev.step()
        assert a.action == b.action and a.marginalization_flag == b.marginalization_flag and a.track_num == b.track_num
        assert a.n_features == b.n_features and a.n_factors == b.n_factors
        if a.action == abi.VIO_FRAME_SOLVED:
            assert a.stats.iterations == b.stats.iterations
    assert len(host.history) == len(dev.history) == n - W
    dp = np.array([x[1] - y[1] for x, y in zip(host.history, dev.history)])
    assert np.abs(dp).max() < 1e-5, np.abs(dp).max()   # (the window kernel's sums are not bit-reproducible: two host-path runs differ alike)
    sa, sb = host.est.status(), dev.est.status()
    assert sa.solver_flag == sb.solver_flag == abi.VIO_SOLVER_NON_LINEAR and sa.prior_rows == sb.prior_rows
    la, pa = _dump_lists(host.est)
    lb, pb = _dump_lists(dev.est)          # (the list comes back from the device for this)
    assert la.shape == lb.shape and np.array_equal(la[:, [0, 1, 2, 4]], lb[:, [0, 1, 2, 4]])
    assert np.array_equal(pa, pb)
    da, db = la[:, 7], lb[:, 7]
    assert np.abs(da - db).max() < 1e-5 * max(1.0, np.abs(da).max())
    # ... and the replay continues after the look at the list (the sequence returns to the device with its next solve)
    for _ in range(8):
        a, b = host.step(), dev.step()
        assert a.action == b.action == abi.VIO_FRAME_SOLVED and a.n_factors == b.n_factors
    assert np.abs(host.history[-1][1] - dev.history[-1][1]).max() < 1e-5
    host.close(), dev.close()


@pytest.mark.gpu
def test_resident_batch_with_staggered_starts_failure_and_relocalization():
    """Four sequences in one estimator, resident stores on: they reach the solve phase on different frames (host-path
    and resident windows in the same call), one of them loses its track (failure detection on the device clears the
    slot), one gets a relocalization frame (its list returns to the host for the loop factors). Every sequence must
    follow the single-sequence host-path replay of its own world."""
    cfg = abi.default_config(window_size=6)
    W = cfg.window_size
    nq, steps = 4, 30
    worlds = [RS.SyntheticWorld(cfg, 40 + q) for q in range(nq)]
    singles = [RS.EstimatorLoop(cfg, seed=40 + q, init_noise=1.0) for q in range(nq)]
    for s in singles:
        for _ in range(steps):
            s.step()
    est = pkg.estimator.Estimator(cfg, worlds[0].tic, worlds[0].ric, n_seq=nq)
    est.set_resident(True)
    feeders = [RS.EstimatorLoop(cfg, seed=40 + q, init_noise=1.0, world=worlds[q]) for q in range(nq)]
    start = [0, 2, 5, 9]
    got = [[] for _ in range(nq)]
    for call in range(steps + max(start)):
        obs, hdr, act = [], [], []
        for q in range(nq):
            k = call - start[q]
            f = feeders[q]
            if k < 0 or k >= steps:
                obs.append(([], [])), hdr.append(0.0), act.append(0)
                continue
            f.est.close()
            f.est = _SeqView(est, q)
            obs.append(f.feed_until_image()), hdr.append(worlds[q].time(k)), act.append(1)
        res = est.process_images(obs, hdr, act)
        for q in range(nq):
            if res[q].action == abi.VIO_FRAME_SOLVED:
                got[q].append(est.window(q)["Ps"][W].copy())
    for q in range(nq):
        want = np.array([h[1] for h in singles[q].history])
        assert len(got[q]) == len(want) and np.abs(np.array(got[q]) - want).max() < 1e-6
    est.close()
    for s in singles:
        s.close()


@pytest.mark.gpu
def test_resident_failure_detection_clears_the_slot_and_the_sequence_restarts():
    """A frame that tracks nothing: failureDetection runs on the device for a resident sequence, the slot clears itself, the
    next window fills on the host-side list, is solved there once (anchored where the failed window stood) and moves to
    the device again. The host-path replay of the same calls is the reference."""
    cfg = abi.default_config(window_size=6)
    W = cfg.window_size
    loops = [RS.EstimatorLoop(cfg, seed=9, init_noise=1.0) for _ in range(2)]
    loops[1].est.set_resident(True)
    wins = []
    for loop in loops:
        for _ in range(20):
            loop.step()
        assert loop.history[-1][3].action == abi.VIO_FRAME_SOLVED
        k = loop.k
        for a, w in loop.world.imu_interval(k):
            loop.est.process_imu(loop.world.dt, a, w)
        res = loop.est.process_image(*obs_grid(60, first_id=10 ** 6), loop.world.time(k))
        assert res.action == abi.VIO_FRAME_FAILURE and res.failure_reasons & abi.VIO_FAIL_FEW_FEATURES
        st = loop.est.status()
        assert st.failure_occur == 1 and st.frame_count == 0 and st.solver_flag == abi.VIO_SOLVER_INITIAL and st.prior_rows == 0
        assert loop.est.features().count() == 0
        world = loop.world
        init = []
        loop.world.tracked.clear()
        for j in range(W + 6):
            kk = k + 1 + j
            for a, w in world.imu_interval(kk):
                loop.est.process_imu(world.dt, a, w)
            P, R, V = world.truth(kk)
            if j <= W:
                init.append((world.time(kk), P, R, V))
            if j == W:
                loop.est.set_initial_state([i[0] for i in init], [i[1] for i in init], [i[2] for i in init], [i[3] for i in init],
                                           [world.ba] * (W + 1), [world.bg] * (W + 1))
            res = loop.est.process_image(*world.observe(kk), world.time(kk))
            if j >= W:
                assert res.action == abi.VIO_FRAME_SOLVED
        wins.append(loop.est.window())
        loop.close()
    for key in ("Ps", "Rs", "Vs", "Bas", "Bgs"):
        assert np.abs(wins[0][key] - wins[1][key]).max() < 1e-6, key


@pytest.mark.gpu
def test_resident_sequence_moves_between_the_paths_in_mid_run():
    """set_resident(False) brings every list back to the host in the middle of a replay, set_resident(True) sends them to the
    device again with the next solved frame; a frame with more observations than a store slot takes is handled on the
    host-side list. The host-path replay of the same data is the reference throughout."""
    cfg = abi.default_config(window_size=6, max_corners=40)     # store slots take 256 observations per frame
    W = cfg.window_size
    host = RS.EstimatorLoop(cfg, seed=11, init_noise=1.0)
    dev = RS.EstimatorLoop(cfg, seed=11, init_noise=1.0)
    host.est.set_resident(False)
    dev.est.set_resident(True)
    where = []
    for k in range(44):
        if k == 20:
            dev.est.set_resident(False)
        if k == 28:
            dev.est.set_resident(True)
        if k == 36:       # a burst of 300 extra observations (new ids, seen once): both replays get the same frame
            for loop in (host, dev):
                ids, xyz = loop.feed_until_image()
                extra_ids = list(range(10 ** 6, 10 ** 6 + 300))
                extra = [[-0.4 + 0.8 * (i % 20) / 19, -0.3 + 0.6 * (i // 20) / 14, 1.0] for i in range(300)]
                res = loop.est.process_image(list(ids) + extra_ids, list(xyz) + extra, loop.world.time(loop.k - 1))
                assert res.action == abi.VIO_FRAME_SOLVED
                loop.history.append((loop.k - 1, loop.est.window()["Ps"][W].copy(), loop.world.truth(loop.k - 1)[0], res))
            where.append(dev.est.status().resident)
            continue
        a, b = host.step(), dev.step()
        assert a.action == b.action and a.n_features == b.n_features and a.n_factors == b.n_factors
        where.append(dev.est.status().resident)
    assert where[W + 1] == 1 and where[19] == 1        # on the device once the first window is solved ...
    assert where[20] == 0 and where[27] == 0           # ... on the host while resident stores are off ...
    assert where[29] == 1                              # ... back with the first solved frame after they are on again ...
    assert where[36] == 1 and where[37] == 1           # ... (the oversized frame ran on the host-side list and moved back at once)
    assert len(host.history) == len(dev.history)
    dp = np.array([x[1] - y[1] for x, y in zip(host.history, dev.history)])
    assert np.abs(dp).max() < 1e-5, np.abs(dp).max()
    host.close(), dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("events_seed,device_imu", [(2024, False), (7, True), (99, False), (5, True)])
def test_resident_stress_random_path_switches_in_a_batch(events_seed, device_imu, monkeypatch):
    """Six sequences in one estimator, 70 calls, seeded random events applied to a resident and a host-only estimator alike:
    sequences skipped in a call (they fall out of step with each other), looks at a landmark list (the list returns to the
    host and moves back), frames with more observations than a store slot takes, resident stores switched off and on for the
    whole object. Every sequence must follow the host-only estimator: actions, counts, positions."""
    # device_imu: VIO_AMD_RESIDENT_IMU=1, the IMU samples of resident sequences are integrated by a kernel (preint_core.h)
    # instead of on the host; the blocks must come out the same, so everything below must hold unchanged
    monkeypatch.setenv("VIO_AMD_RESIDENT_IMU", "1" if device_imu else "0")
    cfg = abi.default_config(window_size=6, max_corners=40)
    W, nq, calls = cfg.window_size, 6, 70
    rng = np.random.default_rng(events_seed)
    both = [[RS.SyntheticWorld(cfg, 60 + q) for q in range(nq)] for _ in range(2)]   # (a world keeps track state: one per side)
    worlds = both[0]
    ests = [pkg.estimator.Estimator(cfg, worlds[0].tic, worlds[0].ric, n_seq=nq) for _ in range(2)]
    ests[0].set_resident(False)
    ests[1].set_resident(True)
    feeders = [[RS.EstimatorLoop(cfg, seed=60 + q, init_noise=1.0, world=both[side][q]) for q in range(nq)] for side in range(2)]
    for e, fs in zip(ests, feeders):
        for q, f in enumerate(fs):
            f.est.close()
            f.est = _SeqView(e, q)
    got = [[[] for _ in range(nq)] for _ in range(2)]
    resident_calls, n_reloc, n_loopf = 0, 0, 0
    for call in range(calls):
        active = [1 if rng.random() > 0.12 else 0 for _ in range(nq)]
        burst = [bool(a and rng.random() < 0.06) for a in active]
        peek = [q for q in range(nq) if rng.random() < 0.05]
        if call in (25, 48):
            ests[1].set_resident(call == 48)
        results = []
        for side in range(2):
            obs, hdr = [], []
            for q in range(nq):
                f = feeders[side][q]
                if not active[q]:
                    obs.append(([], [])), hdr.append(0.0)
                    continue
                ids, xyz = f.feed_until_image()
                if burst[q]:
                    ids = list(ids) + list(range(10 ** 6 + 1000 * call, 10 ** 6 + 1000 * call + 300))
                    xyz = list(xyz) + [[-0.4 + 0.8 * (i % 20) / 19, -0.3 + 0.6 * (i // 20) / 14, 1.0] for i in range(300)]
                obs.append((ids, xyz)), hdr.append(worlds[q].time(f.k - 1))
            res = ests[side].process_images(obs, hdr, active)
            results.append(res)
            for q in range(nq):
                if res[q].action == abi.VIO_FRAME_SOLVED:
                    got[side][q].append(ests[side].window(q)["Ps"][W].copy())
        for q in range(nq):
            a, b = results[0][q], results[1][q]
            assert a.action == b.action and a.n_features == b.n_features and a.n_factors == b.n_factors, (call, q, a.action, b.action)
            assert a.marginalization_flag == b.marginalization_flag and a.track_num == b.track_num
        resident_calls += sum(ests[1].status(q).resident for q in range(nq))
        # now and then a sequence gets a relocalization frame: a window frame's landmarks as the host-only estimator lists them,
        # seen from a shifted old keyframe; both estimators get the same call
        for q in range(nq):
            if results[0][q].action != abi.VIO_FRAME_SOLVED or rng.random() > 0.08:
                continue
            wq = ests[0].window(q)
            i = int(rng.integers(1, W - 1))
            info, pts = ests[0].features(q).dump()
            ids, xy, off = [], [], 0
            for fid, start, n_obs in info[:, :3].astype(int):
                if start <= i <= start + n_obs - 1:
                    p = pts[off + (i - start)]
                    ids.append(int(fid)), xy.append([p[0], p[1]])
                off += n_obs
            order = np.argsort(ids)
            ids, xy = [ids[j] for j in order], [xy[j] for j in order]
            if len(ids) < 8:
                continue
            n_reloc += 1
            for e in ests:
                e.set_relocalization(wq["headers"][i], wq["Ps"][i] + np.array([0.4, -0.2, 0.0]), synth.rot_to_quat(wq["Rs"][i]), ids, xy, seq=q)
        for q in range(nq):
            assert results[0][q].n_loop_factors == results[1][q].n_loop_factors
            n_loopf += results[1][q].n_loop_factors
        for q in peek:
            la, lb = ests[0].features(q).dump(), ests[1].features(q).dump()
            assert np.array_equal(la[0][:, [0, 1, 2, 4]], lb[0][:, [0, 1, 2, 4]]) and np.array_equal(la[1], lb[1])
    assert resident_calls > nq * 20            # the sequences did spend most of the solved frames on the device
    assert n_reloc >= 3 and n_loopf > 50       # ... and relocalization factors were paired on both paths
    for q in range(nq):
        a, b = np.array(got[0][q]), np.array(got[1][q])
        assert len(a) == len(b) and len(a) > 30 and np.abs(a - b).max() < 1e-5, (q, len(a), len(b))
    for e in ests:
        e.close()


@pytest.mark.gpu
def test_resident_relocalization_factors_match_the_host_side_list():
    """The same relocalization frame handed to a host-only and a resident estimator: the loop factors are paired with the
    landmarks by store_pack on the device (the forward walk of VINS.cpp:597-631 as a prefix maximum), the loop pose is solved
    and the drift reported -- same factor counts frame by frame, same drift, same positions, and the sequence never leaves the
    device."""
    cfg = abi.default_config()
    W = cfg.window_size
    host = RS.EstimatorLoop(cfg, seed=5, init_noise=1.0)
    dev = RS.EstimatorLoop(cfg, seed=5, init_noise=1.0)
    host.est.set_resident(False), dev.est.set_resident(True)
    for _ in range(25):
        host.step(), dev.step()
    win = host.est.window()
    i = 4
    info, pts = host.est.features().dump()             # (the host-only estimator's list: looking at it moves nothing)
    ids, xy, off = [], [], 0
    for fid, start, n_obs in info[:, :3].astype(int):
        if start <= i <= start + n_obs - 1 and n_obs >= 2 and start < W - 2:
            p = pts[off + (i - start)]
            ids.append(fid), xy.append([p[0], p[1]])
        off += n_obs
    assert len(ids) > 30
    drift = np.array([1.5, -0.5, 0.0])
    for loop in (host, dev):
        w = loop.est.window()
        loop.est.set_relocalization(w["headers"][i], w["Ps"][i] + drift, synth.rot_to_quat(w["Rs"][i]), ids, xy)
    n_loop = []
    for k in range(14):
        a, b = host.step(), dev.step()
        assert a.action == b.action == abi.VIO_FRAME_SOLVED
        assert a.n_loop_factors == b.n_loop_factors and a.n_factors == b.n_factors and a.stats.iterations == b.stats.iterations
        assert dev.est.status().resident == 1
        n_loop.append(b.n_loop_factors)
        sa, sb = host.est.status(), dev.est.status()
        for key in ("r_drift", "t_drift", "relative_t", "relative_q", "loop_pose"):
            assert np.abs(np.array(getattr(sa, key)) - np.array(getattr(sb, key))).max() < 1e-5, (k, key)
        assert abs(sa.relative_yaw - sb.relative_yaw) < 1e-5
    assert n_loop[0] == len(ids) and n_loop[-1] == 0                # the constraint entered with every match and left with its frame
    dp = np.array([x[1] - y[1] for x, y in zip(host.history, dev.history)])
    assert np.abs(dp).max() < 1e-5
    host.close(), dev.close()
