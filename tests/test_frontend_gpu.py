"""GPU parity tests of the HIP KLT front-end against the CPU oracle (OpenCV-3.0 semantics, parity unpinned: the
oracle is its own reference). Integer and index work is compared bit-exactly; the float stages of LK and of the
min-eigen map use the same IEEE operation sequence on both sides (no FMA contraction) and are compared exactly too."""
import numpy as np
import pytest

import helpers as H
from helpers import abi, synth, pkg

pytestmark = pytest.mark.gpu
fe = pkg.frontend


@pytest.fixture(scope="module")
def stream():
    frames, aff = synth.make_image_stream(5, 7, rows=640, cols=480)
    return frames, aff


def test_klt_track_bit_exact(stream):
    frames, _ = stream
    cfg = abi.default_config(max_corners=150, min_dist=20)
    pts = H.oracle_good_features(cfg, frames[0], None, 150)
    assert len(pts) == 150
    # add sub-pixel starts, points near / outside the border and a point in a flat region
    rng = np.random.default_rng(1)
    pts = pts + rng.uniform(-0.5, 0.5, pts.shape).astype(np.float32)
    extra = np.array([[2.3, 3.1], [478.6, 638.2], [-30.0, 50.0], [500.0, 700.0], [240.0, 1.0]], np.float32)
    pts = np.vstack([pts, extra])
    got, gst, gerr = fe.klt_track(cfg, frames[0], frames[1], pts)
    ref, rst, rerr = H.oracle_klt(cfg, frames[0], frames[1], pts)
    assert (gst == rst).all()
    assert rst.sum() >= 150
    ok = rst > 0
    assert np.array_equal(got[ok], ref[ok]), np.abs(got[ok] - ref[ok]).max()
    assert np.array_equal(gerr[ok], rerr[ok])


def test_klt_extreme_contrast_takes_the_split_sums():
    """The wave sums of the LK system run unsplit while every lane's partial is below 2^24 and fall back to the 16-bit split otherwise
    (vio_frontend.hip wave_all_small). A black / white block pattern drives the derivative products far past that bound
    (7 pixels x 4080^2 per lane): the split form must give the oracle's bits too, on the same call as ordinary windows."""
    rng = np.random.default_rng(7)
    soft = synth.make_image_stream(3, 2)[0]
    rows, cols = soft[0].shape
    half = cols // 2
    blocks = (rng.integers(0, 2, (rows // 6 + 1, cols // 6 + 1)) * 255).astype(np.uint8)
    img0 = np.ascontiguousarray(np.kron(blocks, np.ones((6, 6), np.uint8))[:rows, :cols])
    img1 = np.ascontiguousarray(np.roll(img0, (1, 2), axis=(0, 1)))         # a shift LK can follow
    img0[:, half:], img1[:, half:] = soft[0][:, half:], soft[1][:, half:]   # ordinary contrast on the right half: the unsplit form
    cfg = abi.default_config(max_corners=150, min_dist=20)
    gy, gx = np.meshgrid(np.arange(40, rows - 40, 40), np.arange(40, cols - 40, 40), indexing="ij")
    pts = np.stack([gx.ravel(), gy.ravel()], axis=1).astype(np.float32) + rng.uniform(-0.5, 0.5, (gx.size, 2)).astype(np.float32)
    got, gst, gerr = fe.klt_track(cfg, img0, img1, pts)
    ref, rst, rerr = H.oracle_klt(cfg, img0, img1, pts)
    assert (gst == rst).all() and rst.sum() >= 40
    ok = rst > 0
    assert np.array_equal(got[ok], ref[ok]), np.abs(got[ok] - ref[ok]).max()
    assert np.array_equal(gerr[ok], rerr[ok])
    # the pattern really is past the bound: a 21 x 21 window of it holds derivative products above 64 x 2^24 in sum
    gxs = np.abs(np.diff(img0[:, :half].astype(np.int64), axis=1)).max()
    assert gxs == 255


def test_klt_small_image_and_levels():
    frames, _ = synth.make_image_stream(9, 2, rows=120, cols=96)  # only 2 pyramid levels hold a 21x21 window
    cfg = abi.default_config(max_corners=40, min_dist=10, image_rows=120, image_cols=96)
    pts = H.oracle_good_features(cfg, frames[0], None, 40)
    got, gst, _ = fe.klt_track(cfg, frames[0], frames[1], pts)
    ref, rst, _ = H.oracle_klt(cfg, frames[0], frames[1], pts)
    assert (gst == rst).all() and np.array_equal(got[rst > 0], ref[rst > 0])


def test_good_features_identical(stream):
    frames, _ = stream
    cfg = abi.default_config(max_corners=150, min_dist=30)
    for mask in (None, "discs"):
        m = None
        if mask:
            m = np.full(frames[0].shape, 255, np.uint8)
            m[100:300, 50:250] = 0
            m[:, 400:] = 0
        got = fe.good_features(cfg, frames[2], m, 150)
        ref = H.oracle_good_features(cfg, frames[2], m, 150)
        assert len(ref) > 50
        assert np.array_equal(got, ref)


def test_good_features_flat_image_and_empty_mask():
    cfg = abi.default_config(max_corners=50)
    flat = np.full((640, 480), 128, np.uint8)
    assert len(fe.good_features(cfg, flat, None, 50)) == 0
    frames, _ = synth.make_image_stream(3, 1)
    assert len(fe.good_features(cfg, frames[0], np.zeros((640, 480), np.uint8), 50)) == 0


def test_fundamental_ransac_identical(stream):
    frames, _ = stream
    cfg = abi.default_config(max_corners=150, min_dist=20)
    p1 = H.oracle_good_features(cfg, frames[0], None, 150)
    p2, st, _ = H.oracle_klt(cfg, frames[0], frames[1], p1)
    p1, p2 = p1[st > 0], p2[st > 0].copy()
    rng = np.random.default_rng(0)
    bad = rng.choice(len(p1), 15, replace=False)
    p2[bad] += (rng.uniform(5, 25, (15, 2)) * rng.choice([-1, 1], (15, 2))).astype(np.float32)
    got, ref = fe.fundamental_ransac(cfg, p1, p2), H.oracle_ransac(cfg, p1, p2)
    assert np.array_equal(got, ref)
    assert ref.sum() < len(p1) and ref.sum() > len(p1) - 30
    # 8..14 points: OpenCV 3.0.0 switches to LMedS (fundam.cpp); below 8 the tracker does not call at all
    rejected = 0
    for n in range(8, 15):
        for off in (0, 20, 40):
            q1, q2 = p1[off:off + n], p2[off:off + n].copy()
            if off:
                q2[n // 2] += np.float32(17.0)  # one gross outlier
            got, ref = fe.fundamental_ransac(cfg, q1, q2), H.oracle_ransac(cfg, q1, q2)
            assert np.array_equal(got, ref), (n, off)
            rejected += int((ref == 0).sum())
    assert rejected > 0
    assert fe.fundamental_ransac(cfg, p1[:7], p2[:7]).all()


def test_tracker_sequence_matches_oracle():
    """readImage over a 3-sequence batch for 10 frames (FREQ = 3 publish cadence)."""
    cfg = abi.default_config(max_corners=150, min_dist=20)
    S, T = 3, 10
    streams = [synth.make_image_stream(20 + s, T, rows=640, cols=480)[0] for s in range(S)]
    trk = fe.FeatureTracker(cfg, n_seq=S)
    oracles = [H.OracleTracker(cfg) for _ in range(S)]
    for f in range(T):
        publish = f % 3 == 0
        frames = np.stack([streams[s][f] for s in range(S)])
        got = trk.read_images(frames, publish)
        for s in range(S):
            rids, rxyz = oracles[s].read_image(streams[s][f], publish)
            gids, gxyz = got[s]
            assert np.array_equal(gids, rids), (f, s)
            assert np.array_equal(gxyz, rxyz), (f, s)
            gp, gi, gc = trk.state(s)
            rp, ri, rc = oracles[s].state()
            assert np.array_equal(gi, ri) and np.array_equal(gc, rc), (f, s)
            assert np.array_equal(gp, rp), (f, s, np.abs(gp - rp).max())
        if publish:
            assert len(got[0][0]) > 100
    trk.close()
    for o in oracles:
        o.close()


def test_no_detection_when_no_corner_is_needed(monkeypatch):
    """feature_tracker.cpp:256-266: goodFeaturesToTrack is only called while n_max_cnt = MAX_CNT - forw_pts.size() > 0. A sequence
    that shows the same frame again keeps every feature, so detect_kernel returns at once for it (the other sequence of the batch
    moves and tops up as usual); results equal the oracle's, and equal what the kernel gives when it is made to run for every
    sequence (VIO_AMD_DETECT_ALWAYS=1, the bench's setting)."""
    cfg = abi.default_config(max_corners=60, min_dist=30)
    T = 6
    moving = synth.make_image_stream(31, T, rows=640, cols=480)[0]
    still = [moving[0]] * T
    streams = [still, moving]
    outs = []
    for always in ("0", "1"):
        monkeypatch.setenv("VIO_AMD_DETECT_ALWAYS", always)
        trk = fe.FeatureTracker(cfg, n_seq=2)
        oracles = [H.OracleTracker(cfg) for _ in range(2)]
        rec = []
        for f in range(T):
            got = trk.read_images(np.stack([streams[s][f] for s in range(2)]), True)
            for s in range(2):
                rids, rxyz = oracles[s].read_image(streams[s][f], True)
                assert np.array_equal(got[s][0], rids) and np.array_equal(got[s][1], rxyz), (always, f, s)
                rec.append((got[s][0].copy(), got[s][1].copy()))
            if f >= 1:
                assert len(got[0][0]) == 60  # the still sequence tracks all of its MAX_CNT features: nothing to detect
        trk.close()
        for o in oracles:
            o.close()
        outs.append(rec)
    for (ia, xa), (ib, xb) in zip(*outs):
        assert np.array_equal(ia, ib) and np.array_equal(xa, xb)


def test_resident_stepping_equals_read_images():
    cfg = abi.default_config(max_corners=100, min_dist=25)
    S, T = 2, 4
    streams = np.stack([np.stack([synth.make_image_stream(40 + s, T)[0][f] for s in range(S)]) for f in range(T)])
    a = fe.FeatureTracker(cfg, n_seq=S)
    b = fe.FeatureTracker(cfg, n_seq=S)
    a.upload_frames(streams)
    for f in range(T):
        a.step(f, f % 3 == 0)
        b.read_images(streams[f], f % 3 == 0)
    a.sync()
    for s in range(S):
        pa, ia, ca = a.state(s)
        pb, ib, cb = b.state(s)
        assert np.array_equal(pa, pb) and np.array_equal(ia, ib) and np.array_equal(ca, cb)
    ms, n = a.kernel_ms()
    assert n == T and ms > 0
    a.close(), b.close()


def test_resident_frames_are_tracked_in_place_across_a_new_upload():
    """vio_frontend_step_resident reads level 0 of both pyramids where the frames lie in the context's ring (no copy). A second
    vio_frontend_upload_frames frees that ring while the current image still lives in it: the image must move into the pyramid's
    own storage first. Stepping through two rings, and then on through the host path, equals read_images frame by frame -- with the
    in-place reading and with VIO_AMD_COPY_LEVEL0=1."""
    import os
    cfg = abi.default_config(max_corners=100, min_dist=25)
    S, T = 2, 6
    streams = np.stack([np.stack([synth.make_image_stream(50 + s, T)[0][f] for s in range(S)]) for f in range(T)])
    want = fe.FeatureTracker(cfg, n_seq=S)
    ref_states = []
    for f in range(T):
        want.read_images(streams[f], f % 2 == 0)
        ref_states.append([want.state(s) for s in range(S)])
    want.close()
    for copy0 in ("0", "1"):
        os.environ["VIO_AMD_COPY_LEVEL0"] = copy0
        try:
            a = fe.FeatureTracker(cfg, n_seq=S)
            a.upload_frames(streams[:3])
            for f in range(3):
                a.step(f, f % 2 == 0)
            a.upload_frames(np.ascontiguousarray(streams[3:5]))   # the ring of frames 0..2 goes away; frame 2 is the current image
            for f in range(3, 5):
                a.step(f - 3, f % 2 == 0)
                a.sync()
                for s in range(S):
                    for x, y in zip(a.state(s), ref_states[f][s]):
                        assert np.array_equal(x, y), (copy0, f, s)
            a.read_images(streams[5], False)                      # host path behind resident frames: previous image still in the ring
            for s in range(S):
                for x, y in zip(a.state(s), ref_states[5][s]):
                    assert np.array_equal(x, y), (copy0, 5, s)
            a.close()
        finally:
            os.environ.pop("VIO_AMD_COPY_LEVEL0", None)


def test_submit_collect_equals_read_images():
    """The two halves of read_images (vio_frontend_submit_images / vio_frontend_collect) with host work between them
    publish what the one call publishes; a second submit or a collect out of order is VIO_ESTATE."""
    import ctypes as C
    cfg = abi.default_config(max_corners=80, min_dist=20, image_rows=240, image_cols=320)
    S, T, cap = 3, 5, 80
    streams = [synth.make_image_stream(40 + s, T, rows=240, cols=320)[0] for s in range(S)]
    a, b = fe.FeatureTracker(cfg, n_seq=S), fe.FeatureTracker(cfg, n_seq=S)
    lib = b.lib
    obs = np.zeros(S * cap, fe._OBS_DTYPE)
    obs_p, n_obs = C.cast(obs.ctypes.data, C.POINTER(abi.VioObs)), np.zeros(S, np.int32)
    u8p, ip = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    assert lib.vio_frontend_collect(b._h, obs_p, n_obs.ctypes.data_as(ip)) == abi.VIO_ESTATE
    for f in range(T):
        frames = np.ascontiguousarray(np.stack([streams[s][f] for s in range(S)]))
        want = a.read_images(frames, f % 2 == 0)
        assert lib.vio_frontend_submit_images(b._h, frames.ctypes.data_as(u8p), 240, 320, 320, int(f % 2 == 0)) == 0
        assert lib.vio_frontend_submit_images(b._h, frames.ctypes.data_as(u8p), 240, 320, 320, 1) == abi.VIO_ESTATE
        frames[:] = 0      # the caller's buffer is free again once submit has returned
        assert lib.vio_frontend_collect(b._h, obs_p, n_obs.ctypes.data_as(ip)) == 0
        o = obs.reshape(S, cap)
        for s in range(S):
            ids, xyz = want[s]
            assert n_obs[s] == len(ids)
            assert np.array_equal(o[s, :n_obs[s]]["id"], ids) and np.array_equal(o[s, :n_obs[s]]["x"], xyz[:, 0])
            for x, y in zip(a.state(s), b.state(s)):
                assert np.array_equal(x, y)
    a.close(), b.close()


@pytest.mark.parametrize("rows,cols,corners,min_dist,T", [
    (250, 333, 60, 12, 5),     # neither a multiple of the 32-row strips nor of the 60-column waves of detect_kernel
    (97, 61, 20, 8, 4),        # one partial strip row, one partial wave; 2 pyramid levels
    (720, 1280, 300, 30, 3),   # BASELINE configs[2] frame geometry and feature count
    (1080, 1920, 500, 30, 3),  # BASELINE configs[4] frame geometry and feature count (close to the 512-feature capacity)
])
def test_tracker_sequence_other_geometries(rows, cols, corners, min_dist, T):
    """Tracker state and published observations stay bit-identical to the oracle on frame sizes that leave partial
    strips / waves / tiles in every kernel."""
    cfg = abi.default_config(max_corners=corners, min_dist=min_dist, image_rows=rows, image_cols=cols)
    S = 2
    streams = [synth.make_image_stream(70 + s, T, rows=rows, cols=cols)[0] for s in range(S)]
    trk = fe.FeatureTracker(cfg, n_seq=S)
    oracles = [H.OracleTracker(cfg) for _ in range(S)]
    for f in range(T):
        publish = f % 2 == 0
        got = trk.read_images(np.stack([streams[s][f] for s in range(S)]), publish)
        for s in range(S):
            rids, rxyz = oracles[s].read_image(streams[s][f], publish)
            gids, gxyz = got[s]
            assert np.array_equal(gids, rids), (f, s)
            assert np.array_equal(gxyz, rxyz), (f, s)
            gp, gi, gc = trk.state(s)
            rp, ri, rc = oracles[s].state()
            assert np.array_equal(gi, ri) and np.array_equal(gc, rc), (f, s)
            assert np.array_equal(gp, rp), (f, s, np.abs(gp - rp).max())
        if publish:
            assert len(got[0][0]) > corners // 4
    trk.close()
    for o in oracles:
        o.close()


def test_good_features_mask_image_odd_size():
    """Stand-alone goodFeaturesToTrack with a mask IMAGE (detect_kernel<true>) on an odd-sized frame."""
    rows, cols = 203, 187
    frames, _ = synth.make_image_stream(91, 1, rows=rows, cols=cols)
    cfg = abi.default_config(max_corners=80, min_dist=9, image_rows=rows, image_cols=cols)
    rng = np.random.default_rng(3)
    m = np.full((rows, cols), 255, np.uint8)
    for _ in range(12):
        y, x = rng.integers(0, rows), rng.integers(0, cols)
        m[max(0, y - 15):y + 15, max(0, x - 20):x + 20] = 0
    for mask in (None, m):
        got = fe.good_features(cfg, frames[0], mask, 80)
        ref = H.oracle_good_features(cfg, frames[0], mask, 80)
        assert np.array_equal(got, ref)
    assert len(got) > 10


def test_lk_iteration_counters():
    """vio_frontend_lk_iterations: the counting variant of the LK kernel reports, per pyramid level, one visit per (tracked
    feature, level) and between 1 and lk_max_iters iterations per visit; the tracker's results do not depend on it."""
    cfg = abi.default_config(max_corners=80, min_dist=20, image_rows=240, image_cols=320)
    frames = synth.make_image_stream(5, 4, rows=240, cols=320)[0]
    a, b = fe.FeatureTracker(cfg, n_seq=1), fe.FeatureTracker(cfg, n_seq=1)
    a.lk_iterations(enable=True, read=True)
    n_tracked = 0
    for f in range(4):
        ga = a.read_images(frames[f:f + 1], True)[0]
        gb = b.read_images(frames[f:f + 1], True)[0]
        assert np.array_equal(ga[0], gb[0]) and np.array_equal(ga[1], gb[1])
        if f < 3:
            n_tracked += len(a.state(0)[1])   # the points the next frame's LK call starts from
    it, vis = a.lk_iterations(enable=False, read=True)
    levels = int((vis > 0).sum())
    assert levels >= 2 and np.all(vis[:levels] == vis[0]) and vis[0] == n_tracked, (vis, n_tracked)
    mean = it[:levels] / vis[:levels]
    assert np.all(mean >= 1.0) and np.all(mean <= cfg.lk_max_iters), mean
    it2, vis2 = a.lk_iterations(read=True)
    assert it2.sum() == 0 and vis2.sum() == 0   # reading resets
    a.close(), b.close()
