"""Size-independent properties of the HIP path at BASELINE.json's full configuration (configs[1]: 640x480 frames,
150 features, window 10, ~850 projection factors, 256 sequences per GPU) — the sizes at which the CPU oracle is too
slow to be the checker for every item of a batch."""
import numpy as np
import pytest

import helpers as H
from helpers import abi, synth, pkg

pytestmark = pytest.mark.gpu


def yaw_deg(q_xyzw):
    x, y, z, w = q_xyzw
    return np.degrees(np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z)))


@pytest.fixture(scope="module")
def batch():
    cfg = abi.default_config(max_corners=150, min_dist=20)
    pre = lambda *a: pkg.backend.preintegrate(cfg, *a)
    uniq = [synth.make_window(cfg, pre, seed=300 + i, n_features=150) for i in range(8)]
    B = 256
    ws = [uniq[i % len(uniq)].copy() for i in range(B)]
    solver = pkg.backend.WindowSolver(cfg, max_batch=B)
    inp = [w.copy() for w in ws]
    stats = solver.solve(ws)
    yield cfg, solver, inp, ws, stats, len(uniq)
    solver.close()


def test_full_batch_equals_itself_and_single_solves(batch):
    cfg, solver, inp, ws, stats, nu = batch
    # copies of the same window inside one launch agree to rounding: the order of the floating-point atomics (LDS / L2) is
    # the only non-determinism. These first-solve windows carry no prior, so the scale of the window is held by the IMU
    # alone and rounding noise shows up as a common factor of ~1e-9 on all inverse depths (observed <= 1.2e-9 relative
    # over 1000 launches; north_star asks for 1e-4)
    for i in range(nu, len(ws)):
        assert np.abs(ws[i].pose - ws[i % nu].pose).max() < 1e-8
        assert np.abs(ws[i].inv_depth - ws[i % nu].inv_depth).max() < 1e-8 * np.abs(ws[i].inv_depth).max() + 1e-12
        assert stats[i]["iterations"] == stats[i % nu]["iterations"]
    # ... and equal the window solved alone
    alone = inp[3].copy()
    pkg.backend.WindowSolver(cfg, max_batch=1).solve([alone])
    assert np.abs(alone.pose - ws[3].pose).max() < 1e-8


def test_cost_trace_and_gauge(batch):
    cfg, solver, inp, ws, stats, nu = batch
    for u in range(nu):
        s = stats[u]
        n = s["iterations"]
        cost, flags = np.asarray(s["it_cost"][:n]), np.asarray(s["it_flags"][:n])
        acc = cost[(flags & 2) != 0]                       # accepted iterations
        assert (np.diff(acc) <= 1e-12 * acc[:-1]).all()    # trust region: the cost never goes up on an accepted step
        assert s["final_cost"] <= s["initial_cost"] * (1 + 1e-12)
        # new2old (VINS.cpp:131-212): frame 0 keeps the yaw and the position it had before the solve
        assert abs(yaw_deg(ws[u].pose[0, 3:]) - yaw_deg(inp[u].pose[0, 3:])) < 1e-6
        assert np.abs(ws[u].pose[0, :3] - inp[u].pose[0, :3]).max() < 1e-9
        q = ws[u].pose[:, 3:]
        assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 1e-12
        assert np.isfinite(ws[u].inv_depth).all() and np.isfinite(ws[u].speed_bias).all()


def test_resolve_keeps_descending_to_a_cost_plateau(batch):
    cfg, solver, inp, ws, stats, nu = batch
    prev, prev_cost = [ws[u].copy() for u in range(nu)], [stats[u]["final_cost"] for u in range(nu)]
    drop = None
    for rep in range(3):  # the first solve stops at max_iter = 10; re-solving from its result must keep descending
        cur = [p.copy() for p in prev]
        st = solver.solve(cur)
        for u in range(nu):
            assert st[u]["final_cost"] <= prev_cost[u] * (1 + 1e-9)
        drop = [1 - st[u]["final_cost"] / prev_cost[u] for u in range(nu)]
        prev, prev_cost = cur, [st[u]["final_cost"] for u in range(nu)]
    # (without a prior the window keeps 4 gauge directions and a weakly observed scale: the STATE may keep sliding along
    # them, the cost may not) -- after 40 iterations the cost has stopped moving
    assert max(drop) < 1e-3


def test_next_prior_is_a_valid_factor(batch):
    cfg, solver, inp, ws, stats, nu = batch
    for u in range(nu):
        p = ws[u].next_prior
        assert p.n > 0
        Hm, b, x0 = p.canonical()
        assert np.isfinite(Hm).all() and np.isfinite(b).all()
        assert np.abs(Hm - Hm.T).max() <= 1e-9 * np.abs(Hm).max()
        ev = np.linalg.eigvalsh(Hm)
        assert ev.min() >= -1e-8 * ev.max()                # J0^T J0 is positive semi-definite
        kinds = [(k, i) for k, i, _ in x0]
        assert len(set(kinds)) == len(kinds)               # every kept block appears once, already shifted (i - 1)
        assert all(0 <= i < cfg.window_size + 1 for _, i in kinds)


def test_frontend_batch_properties():
    cfg = abi.default_config(max_corners=150, min_dist=20)
    rows, cols, S, T = cfg.image_rows, cfg.image_cols, 32, 4
    streams = [synth.make_image_stream(900 + s % 4, T, rows=rows, cols=cols)[0] for s in range(S)]
    frames = np.stack([np.stack([streams[s][f] for s in range(S)]) for f in range(T)])
    trk = pkg.frontend.FeatureTracker(cfg, n_seq=S)
    seen = [set() for _ in range(S)]
    for f in range(T):
        got = trk.read_images(frames[f], True)
        for s in range(S):
            ids, xyz = got[s]
            pts, sid, cnt = trk.state(s)
            assert len(ids) == len(sid) and (np.sort(ids) == np.sort(sid)).all() and len(set(ids)) == len(ids)
            assert len(ids) <= cfg.max_corners and len(ids) > 100
            # inBorder and the min-distance rule of setMask / goodFeaturesToTrack
            x, y = np.rint(pts[:, 0]), np.rint(pts[:, 1])
            assert (x >= 1).all() and (x < cols - 1).all() and (y >= 1).all() and (y < rows - 1).all()
            d = np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=2) + np.eye(len(pts)) * 1e9
            assert d.min() >= cfg.min_dist - 1.0          # (centres are rounded before the disc test)
            assert (cnt >= 1).all() and cnt.max() <= f + 1
            seen[s] |= set(int(i) for i in ids)
            # sequences fed the same stream evolve identically
            if s >= 4:
                p0, i0, c0 = trk.state(s % 4)
                assert np.array_equal(p0, pts) and np.array_equal(c0, cnt)
    # tracking a frame against itself: nothing moves by more than float rounding, nothing is lost to LK
    n_before = [len(trk.state(s)[0]) for s in range(S)]
    pts_before = [trk.state(s)[0].copy() for s in range(S)]
    trk.read_images(frames[T - 1], False)
    for s in range(0, S, 8):
        pts, _, _ = trk.state(s)
        assert len(pts) >= n_before[s] - 2
        if len(pts) == n_before[s]:
            assert np.abs(pts - pts_before[s]).max() < 2e-3
    trk.close()
