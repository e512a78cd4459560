"""Isolated tests of the steps between the LK call and goodFeaturesToTrack, on tracker fields
set from outside (they are public members of FeatureTracker, feature_tracker.hpp:68-80):
    status && inBorder + reduceVector   feature_tracker.cpp:183-191, 18-34
    rejectWithF                         :89-103 (findFundamentalMat over the publish baseline pre_pts -> forw_pts)
    setMask                             :50-87  (sort by track_cnt, greedy keep, filled circles of MIN_DIST)
The device step is compared bit for bit with the oracle's, and the oracle's setMask with an independent numpy greedy."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from helpers import abi, pkg

pytestmark = pytest.mark.gpu
_fp, _ip, _u8p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


def bind():
    lib = abi.load_product()
    vp = C.c_void_p
    lib.vio_frontend_set_tracks.argtypes = [vp, C.c_int32, C.c_int32, _fp, _fp, _fp, _ip, _ip, _u8p]
    lib.vio_frontend_update_tracks.argtypes = [vp, C.c_int32]
    lib.vio_frontend_get_tracks.argtypes = [vp, C.c_int32, _fp, _ip, _ip, C.c_int32, _ip]
    o = H.oracle_lib()
    o.oracle_tracker_set_tracks.argtypes = [C.c_void_p, C.c_int32, _fp, _fp, _fp, _ip, _ip]
    o.oracle_tracker_update_tracks.argtypes = [C.c_void_p, _u8p, C.c_int32]
    o.oracle_tracker_get_tracks.argtypes = [C.c_void_p, _fp, _ip, _ip, C.c_int32, _ip]
    return lib, o


def run_both(cfg, pre, cur, forw, ids, cnt, status, publish):
    """-> (product (pts, ids, cnt), oracle (pts, ids, cnt)) after one update step on the same fields."""
    lib, o = bind()
    n = len(ids)
    pre, cur, forw = (np.ascontiguousarray(a, np.float32).reshape(-1, 2) for a in (pre, cur, forw))
    ids, cnt = np.ascontiguousarray(ids, np.int32), np.ascontiguousarray(cnt, np.int32)
    status = np.ascontiguousarray(status, np.uint8)
    args = [a.ctypes.data_as(_fp) for a in (pre, cur, forw)] + [ids.ctypes.data_as(_ip), cnt.ctypes.data_as(_ip)]
    trk = pkg.frontend.FeatureTracker(cfg, n_seq=2)   # sequence 1 carries the fields, sequence 0 stays empty
    assert lib.vio_frontend_set_tracks(trk._h, 1, n, *args, status.ctypes.data_as(_u8p)) == 0
    assert lib.vio_frontend_update_tracks(trk._h, 1 if publish else 0) == 0
    cap = cfg.max_corners
    gp, gi, gc, gn = np.zeros((cap, 2), np.float32), np.zeros(cap, np.int32), np.zeros(cap, np.int32), C.c_int32()
    assert lib.vio_frontend_get_tracks(trk._h, 1, gp.ctypes.data_as(_fp), gi.ctypes.data_as(_ip), gc.ctypes.data_as(_ip), cap, C.byref(gn)) == 0
    e0 = C.c_int32()
    assert lib.vio_frontend_get_tracks(trk._h, 0, gp[gn.value:].ctypes.data_as(_fp), gi[gn.value:].ctypes.data_as(_ip),
                                       gc[gn.value:].ctypes.data_as(_ip), 0, C.byref(e0)) == 0 and e0.value == 0
    trk.close()
    ot = H.OracleTracker(cfg)
    o.oracle_tracker_set_tracks(ot.h, n, *args)
    st = status.copy()
    o.oracle_tracker_update_tracks(ot.h, st.ctypes.data_as(_u8p), 1 if publish else 0)
    rp, ri, rc, rn = np.zeros((cap, 2), np.float32), np.zeros(cap, np.int32), np.zeros(cap, np.int32), C.c_int32()
    o.oracle_tracker_get_tracks(ot.h, rp.ctypes.data_as(_fp), ri.ctypes.data_as(_ip), rc.ctypes.data_as(_ip), cap, C.byref(rn))
    ot.close()
    g, r = (gp[:gn.value], gi[:gn.value], gc[:gn.value]), (rp[:rn.value], ri[:rn.value], rc[:rn.value])
    return g, r


def assert_same(g, r):
    assert len(g[1]) == len(r[1])
    assert np.array_equal(g[0], r[0]) and np.array_equal(g[1], r[1]) and np.array_equal(g[2], r[2])


def two_view_points(rng, cfg, n, shift):
    """Points of a 3D scene seen from two camera positions (a real epipolar geometry with parallax)."""
    z = rng.uniform(3, 9, n)
    x = rng.uniform(-0.35, 0.35, n) * z
    y = rng.uniform(-0.5, 0.5, n) * z
    a = np.column_stack([cfg.fx * x / z + cfg.cx, cfg.fy * y / z + cfg.cy])
    b = np.column_stack([cfg.fx * (x - shift[0]) / (z - shift[2]) + cfg.cx, cfg.fy * (y - shift[1]) / (z - shift[2]) + cfg.cy])
    return a.astype(np.float32), b.astype(np.float32)


def test_reject_with_f_removes_publish_baseline_outliers():
    """F6: the frame-to-frame motion (cur -> forw) is consistent for every track, but twelve tracks drifted over the
    publish baseline (pre -> forw): only rejectWithF can see them."""
    rng = np.random.default_rng(5)
    cfg = abi.default_config(max_corners=150, min_dist=1)   # MIN_DIST 1: setMask keeps everything that is 2 px apart
    n = 110
    pre, forw = two_view_points(rng, cfg, n, (0.25, 0.05, 0.1))
    cur = (0.6 * pre + 0.4 * forw).astype(np.float32)       # same geometry, shorter baseline: the first RANSAC keeps it
    cur, _ = two_view_points(np.random.default_rng(5), cfg, n, (0.1, 0.02, 0.04))
    bad = rng.choice(n, 12, replace=False)
    pre[bad] += rng.uniform(8, 25, (12, 2)).astype(np.float32) * rng.choice([-1, 1], (12, 2))
    ids, cnt = np.arange(100, 100 + n), rng.integers(1, 9, n)
    g, r = run_both(cfg, pre, cur, forw, ids, cnt, np.ones(n, np.uint8), publish=True)
    assert_same(g, r)
    kept = set(g[1].tolist())
    assert len(kept & set(ids[bad].tolist())) <= 1           # the drifted tracks are gone
    assert len(kept) >= n - 12 - 6                            # and (nearly) only those
    # without publish the same fields lose nothing to rejectWithF, and track_cnt is not incremented
    g2, r2 = run_both(cfg, pre, cur, forw, ids, cnt, np.ones(n, np.uint8), publish=False)
    assert_same(g2, r2)
    assert len(g2[1]) >= n - 3 and np.array_equal(g2[2], cnt[np.isin(ids, g2[1])])


def numpy_set_mask(cfg, pts, ids, cnt):
    """Independent statement of setMask: stable sort by count (descending), keep a point iff its rounded pixel is not
    inside an earlier kept point's filled circle (cv::circle's raster: half-widths from the midpoint algorithm)."""
    r = cfg.min_dist
    hw = np.full(2 * r + 1, -1)
    err, dx, dy, plus, minus = 0, r, 0, 1, (r << 1) - 1
    while dx >= dy:
        for a, b in ((dy, dx), (-dy, dx), (dx, dy), (-dx, dy)):
            hw[r + a] = max(hw[r + a], b)
        dy += 1
        err += plus
        plus += 2
        mask = (err <= 0) - 1
        err -= minus & mask
        dx += mask
        minus -= mask & 2
    order = sorted(range(len(ids)), key=lambda i: -cnt[i])
    kept = []
    for i in order:
        ix, iy = int(np.rint(pts[i, 0])), int(np.rint(pts[i, 1]))
        free = True
        for j in kept:
            jx, jy = int(np.rint(pts[j, 0])), int(np.rint(pts[j, 1]))
            d = iy - jy
            if -r <= d <= r and abs(ix - jx) <= hw[r + d]:
                free = False
                break
        if free:
            kept.append(i)
    return kept


def test_set_mask_order_ties_and_circles():
    """F7: fewer than 8 tracks (no RANSAC runs), clustered points, equal track counts, points exactly on the circle's edge."""
    cfg = abi.default_config(max_corners=150, min_dist=30)
    rng = np.random.default_rng(9)
    for trial in range(6):
        n = 7 if trial < 4 else 60
        base = rng.uniform(80, 380, (n, 2))
        if trial >= 4:   # >= 8 tracks: make both RANSACs keep everything (identical points in all three views are degenerate:
            base = None  # use a real two-view geometry instead)
        if base is not None:
            pts = base.copy()
            pts[1] = pts[0] + [29.6, 0.2]      # inside the circle of point 0 (rounded: dx = 30, row 0 half-width 30)
            pts[2] = pts[0] + [31.2, 0.0]      # just outside
            pts[3] = pts[0] + [21.0, 21.4]     # diagonal: inside the disc
            pts[4] = pts[0] + [22.0, 22.0]     # diagonal: outside (half-width at dy = 22 is 20)
            pre = cur = forw = pts.astype(np.float32)
        else:
            pre, forw = two_view_points(rng, cfg, n, (0.2, 0.04, 0.08))
            cur, _ = two_view_points(rng, cfg, n, (0.2, 0.04, 0.08))
            cur = (0.5 * pre + 0.5 * forw).astype(np.float32)
            forw[n // 2:] = forw[: n - n // 2] + rng.uniform(-20, 20, (n - n // 2, 2)).astype(np.float32)  # clusters
            pre[n // 2:] = pre[: n - n // 2] + (forw[n // 2:] - forw[: n - n // 2])
            cur[n // 2:] = cur[: n - n // 2] + (forw[n // 2:] - forw[: n - n // 2])
        ids = np.arange(n) + 10
        cnt = rng.integers(1, 4, n)              # many ties: the order among equals is the original order (stable)
        g, r = run_both(cfg, pre, cur, forw, ids, cnt, np.ones(n, np.uint8), publish=True)
        assert_same(g, r)
        if n < 8:
            want = numpy_set_mask(cfg, forw, ids, cnt)
            assert g[1].tolist() == ids[want].tolist()
            assert np.array_equal(g[2], cnt[want] + 1) and np.array_equal(g[0], forw[want])


def test_status_and_border_compaction():
    """F3 / F4: lost tracks (status 0) and tracks whose rounded position leaves [1, COL-1) x [1, ROW-1) are dropped, the
    order of the survivors is kept."""
    cfg = abi.default_config(max_corners=150, min_dist=1)
    n = 7   # below 8: no RANSAC, so exactly the status / border rule is visible
    pts = np.array([[100, 100], [0.4, 50], [0.6, 50], [cfg.image_cols - 1.4, 60], [cfg.image_cols - 1.6, 60], [200, 0.49],
                    [200, cfg.image_rows - 1.51]], np.float32)
    status = np.array([1, 1, 1, 1, 1, 1, 1], np.uint8)
    ids, cnt = np.arange(n), np.full(n, 2)
    g, r = run_both(cfg, pts, pts, pts, ids, cnt, status, publish=False)
    assert_same(g, r)
    # cvRound: x in [1, COL-1): 0.4 -> 0 out, 0.6 -> 1 in, COL-1.4 -> COL-1 out, COL-1.6 -> COL-2 in; y 0.49 -> 0 out; ROW-1.51 -> ROW-2 in
    assert g[1].tolist() == [0, 2, 4, 6]
    status[4] = 0
    g, r = run_both(cfg, pts, pts, pts, ids, cnt, status, publish=False)
    assert_same(g, r)
    assert g[1].tolist() == [0, 2, 6]
