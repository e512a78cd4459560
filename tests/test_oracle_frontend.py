"""CPU tests of the front-end oracle (OpenCV-3.0 semantics restated; PARITY UNPINNED — no OpenCV build exists here
to pin it). What can be checked without the library: primitives against scipy.ndimage, tracking against the known
synthetic motion, RANSAC against injected outliers, and the tracker's bookkeeping invariants."""
import ctypes as C

import numpy as np
import scipy.ndimage as ndi

import helpers as H
from helpers import abi, synth


def test_pyr_down_matches_scipy():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 50), dtype=np.uint8)  # odd sizes exercise the (n+1)/2 rule and the borders
    out = np.zeros(((37 + 1) // 2, (50 + 1) // 2), np.uint8)
    u8p = C.POINTER(C.c_uint8)
    H.oracle_lib().oracle_pyr_down(img.ctypes.data_as(u8p), 37, 50, 50, out.ctypes.data_as(u8p))
    k = np.array([1, 4, 6, 4, 1], np.int64)
    full = ndi.convolve1d(ndi.convolve1d(img.astype(np.int64), k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
    ref = ((full[::2, ::2] + 128) >> 8).astype(np.uint8)
    assert np.array_equal(out, ref)


def test_min_eigen_map_matches_scipy_within_float_rounding():
    frames, _ = synth.make_image_stream(1, 1, rows=96, cols=80)
    img = frames[0]
    eig = np.zeros(img.shape, np.float32)
    H.oracle_lib().oracle_min_eigen_map(img.ctypes.data_as(C.POINTER(C.c_uint8)), 96, 80, 80, eig.ctypes.data_as(C.POINTER(C.c_float)))
    f = img.astype(np.float64)
    s = 1.0 / (4 * 3 * 255)
    dx = ndi.correlate1d(ndi.correlate1d(f, [-1, 0, 1], axis=1, mode="mirror"), [1, 2, 1], axis=0, mode="mirror") * s
    dy = ndi.correlate1d(ndi.correlate1d(f, [1, 2, 1], axis=1, mode="mirror"), [-1, 0, 1], axis=0, mode="mirror") * s
    box = lambda m: ndi.uniform_filter(m, 3, mode="mirror") * 9
    a, b, c = box(dx * dx) * 0.5, box(dx * dy), box(dy * dy) * 0.5
    ref = (a + c) - np.sqrt((a - c) ** 2 + b * b)
    assert np.abs(eig - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())


def test_klt_tracks_known_motion_and_accumulation_modes_agree():
    frames, aff = synth.make_image_stream(5, 3)
    cfg = abi.default_config(max_corners=150, min_dist=20)
    pts = H.oracle_good_features(cfg, frames[0], None, 150)
    nxt, st, err = H.oracle_klt(cfg, frames[0], frames[1], pts)
    assert st.sum() >= 145
    fwd = lambda A, p: (A[:, :2] @ p.T).T + A[:, 2]
    inv = lambda A, t: (np.linalg.inv(A[:, :2]) @ (t - A[:, 2]).T).T
    gt = inv(aff[1], fwd(aff[0], pts.astype(np.float64)))
    e = np.linalg.norm(nxt - gt, axis=1)[st > 0]
    assert np.median(e) < 0.05 and e.max() < 0.3
    H.oracle_lib().oracle_set_lk_accum_mode(1)  # OpenCV scalar-path float accumulation order
    nxt1, st1, _ = H.oracle_klt(cfg, frames[0], frames[1], pts)
    H.oracle_lib().oracle_set_lk_accum_mode(0)
    assert (st1 == st).all() and np.abs(nxt1 - nxt)[st > 0].max() < 1e-3  # <= 1e-3 px between variants


def test_good_features_respects_min_distance_quality_and_mask():
    frames, _ = synth.make_image_stream(2, 1)
    cfg = abi.default_config(max_corners=200, min_dist=30)
    mask = np.full(frames[0].shape, 255, np.uint8)
    mask[:, :100] = 0
    c = H.oracle_good_features(cfg, frames[0], mask, 200)
    assert len(c) > 50 and (c[:, 0] >= 100).all()
    d = np.linalg.norm(c[:, None, :] - c[None, :, :], axis=2) + np.eye(len(c)) * 1e9
    assert d.min() >= 30.0
    assert (c[:, 0] >= 1).all() and (c[:, 0] <= 478).all() and (c[:, 1] >= 1).all() and (c[:, 1] <= 638).all()


def test_ransac_rejects_injected_outliers_only():
    frames, _ = synth.make_image_stream(5, 2)
    cfg = abi.default_config(max_corners=150, min_dist=20)
    p1 = H.oracle_good_features(cfg, frames[0], None, 150)
    p2, st, _ = H.oracle_klt(cfg, frames[0], frames[1], p1)
    p1, p2 = p1[st > 0], p2[st > 0].copy()
    clean = H.oracle_ransac(cfg, p1, p2)
    assert clean.sum() >= len(p1) - 2
    rng = np.random.default_rng(3)
    bad = rng.choice(len(p1), 12, replace=False)
    p2[bad] += (rng.uniform(8, 25, (12, 2)) * rng.choice([-1, 1], (12, 2))).astype(np.float32)
    m = H.oracle_ransac(cfg, p1, p2)
    assert (m[bad] == 0).sum() >= 9          # a gross outlier can still lie along its epipolar line
    good = np.setdiff1d(np.arange(len(p1)), bad)
    assert (m[good] == 0).sum() <= 2


def test_lmeds_branch_below_15_points():
    """8..14 correspondences take OpenCV 3.0.0's LMedS branch: a gross outlier goes, consistent points stay; below 8
    points the tracker never calls findFundamentalMat and everything is kept."""
    frames, _ = synth.make_image_stream(5, 2)
    cfg = abi.default_config(max_corners=150, min_dist=20)
    p1 = H.oracle_good_features(cfg, frames[0], None, 150)
    p2, st, _ = H.oracle_klt(cfg, frames[0], frames[1], p1)
    p1, p2 = p1[st > 0], p2[st > 0].copy()
    for n in (8, 11, 14):
        q1, q2 = p1[30:30 + n], p2[30:30 + n].copy()
        clean = H.oracle_ransac(cfg, q1, q2)
        # the cut is sigma = 2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median error): on nearly noise-free tracks it is
        # tight and drops consistent points too (that is what the 3.0.0 code does); the 7 sample points always survive
        assert set(np.unique(clean)) <= {0, 1} and clean.sum() >= 7
        q2[3] += np.float32([13.0, 19.0])
        m = H.oracle_ransac(cfg, q1, q2)
        # (with 8 points every 7-subset fits its own points exactly, outlier included: only the larger sets must drop it)
        assert m.sum() >= 7 and (n < 14 or m[3] == 0)
        assert np.array_equal(m, H.oracle_ransac(cfg, q1, q2))
    assert H.oracle_ransac(cfg, p1[:7], p2[:7]).all()


def test_tracker_bookkeeping():
    cfg = abi.default_config(max_corners=120, min_dist=25)
    frames, _ = synth.make_image_stream(11, 7)
    t = H.OracleTracker(cfg)
    seen = set()
    last_cnt = {}
    for f in range(7):
        ids, xyz = t.read_image(frames[f], f % 3 == 0)
        pts, sid, cnt = t.state()
        assert len(pts) <= 120
        if f % 3 == 0:
            assert len(ids) == len(sid) and (ids == sid).all() and (ids >= 0).all() and len(set(ids)) == len(ids)
            assert np.allclose(xyz[:, 0], (pts[:, 0].astype(np.float64) - cfg.cx) / cfg.fx)
            for i, c in zip(sid, cnt):  # track_cnt grows by one per publish for surviving ids, new ids start at 1
                assert c == last_cnt.get(i, 0) + 1
            last_cnt = dict(zip(sid.tolist(), cnt.tolist()))
            new = set(ids.tolist()) - seen
            assert all(i > max(seen, default=-1) for i in new)  # ids are handed out monotonically (n_id++)
            seen |= set(ids.tolist())
        else:
            assert len(ids) == 0
    t.close()


def test_integer_identities_behind_the_lk_kernel():
    """The arithmetic shortcuts of csrc/vio_frontend.hip's lk_track_kernel, checked on random operands (numpy, no device):
    (1) an exact 36-bit sum hi * 2^16 + lo goes to float with ONE rounding whether it passes through a double or is formed
    as float(hi) * 65536 + float(lo) (both parts and the product are exact floats); (2) the template value folded into the
    accumulator seed: (a + r - (I << n)) >> n == ((a + r) >> n) - I for the arithmetic shift; (3) |x| < 0.01 in double
    for a float x is |x| < nextafter(0.01f, 1) in float; (4) the float screen of the convergence test never rejects a
    displacement the double compare accepts."""
    rng = np.random.default_rng(11)
    hi = rng.integers(-(1 << 20) + 1, 1 << 20, 200000)
    lo = rng.integers(0, 1 << 22, 200000)
    via_double = (hi.astype(np.float64) * 65536.0 + lo.astype(np.float64)).astype(np.float32)
    direct = hi.astype(np.float32) * np.float32(65536.0) + lo.astype(np.float32)
    assert direct.dtype == np.float32 and np.array_equal(via_double, direct)
    n, r = 9, 1 << 8
    a = rng.integers(0, 255 * (1 << 14) + 1, 200000).astype(np.int64)
    iv = rng.integers(0, 8161, 200000).astype(np.int64)
    assert np.array_equal((a + r - (iv << n)) >> n, ((a + r) >> n) - iv)
    c = np.float32(0.01)
    up = np.nextafter(c, np.float32(1))
    assert float(c) < 0.01 < float(up) and abs(float(up) - 0.010000000707805157) < 1e-18
    x = np.concatenate([np.array([c, up, np.nextafter(c, np.float32(0)), np.nextafter(up, np.float32(1))], np.float32),
                        rng.uniform(0.0099, 0.0101, 10000).astype(np.float32)])
    assert np.array_equal(np.abs(x.astype(np.float64)) < 0.01, np.abs(x) < up)
    eps_sq = 0.01 * 0.01
    screen = np.nextafter(np.float32(eps_sq * (1.0 + 1e-5)), np.float32(np.inf))
    d = rng.uniform(-0.012, 0.012, (200000, 2)).astype(np.float32)
    fsum = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
    exact = d[:, 0].astype(np.float64) ** 2 + d[:, 1].astype(np.float64) ** 2
    assert not np.any((fsum > screen) & (exact <= eps_sq))
