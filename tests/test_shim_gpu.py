"""The source-level shim (include/vio_amd_shim.hpp: the reference's member signatures readImage / processIMU /
processImage over the C ABI) compiled with plain stand-in value types (tests/shim_main.cpp: no OpenCV, no Eigen) and run
against the python mirrors of the same ABI on the same data: what goes through the shim is what the ABI produces."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
from helpers import abi, pkg

sys.path.insert(0, os.path.join(H.ROOT, "tools"))
import replay_synthetic as RS  # noqa: E402


def build_shim(tmp_path):
    exe = str(tmp_path / "shim_main")
    csrc = os.path.join(H.ROOT, "vins-mobile_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(H.ROOT, "include"),
                           os.path.join(H.ROOT, "tests", "shim_main.cpp"), "-o", exe, "-L" + csrc, "-lvio_amd",
                           "-Wl,-rpath," + csrc])
    return exe


def test_shim_compiles_without_opencv_or_eigen(tmp_path):
    """CPU: the header and its driver build with -Wall -Werror against the product library (no device needed to link)."""
    if not os.path.exists(abi.PRODUCT_LIB):
        subprocess.check_call(["make", "-C", os.path.dirname(abi.PRODUCT_LIB)])
    assert os.path.exists(build_shim(tmp_path))


@pytest.mark.gpu
def test_feature_tracker_shim_publishes_what_the_abi_publishes(tmp_path):
    exe = build_shim(tmp_path)
    rows, cols, n, freq = 240, 320, 7, 3
    frames, _ = pkg.synth.make_image_stream(9, n, rows=rows, cols=cols)
    (tmp_path / "frames.bin").write_bytes(np.ascontiguousarray(frames).tobytes())
    subprocess.check_call([exe, str(tmp_path / "frames.bin"), str(rows), str(cols), str(n), str(freq), str(tmp_path / "obs.bin")])
    raw = (tmp_path / "obs.bin").read_bytes()
    cfg = abi.default_config(max_corners=60, min_dist=25, image_rows=rows, image_cols=cols)
    trk = pkg.frontend.FeatureTracker(cfg, n_seq=1)
    off, published = 0, 0
    for f in range(n):
        fi, pub, cnt = struct.unpack_from("<3i", raw, off)
        off += 12
        assert fi == f and pub == (1 if f % freq == 0 else 0)
        ids, xyz = trk.read_images(frames[f:f + 1], bool(pub))[0]
        if pub:
            rec = np.frombuffer(raw, np.float64, 4 * cnt, off).reshape(cnt, 4)
            off += 32 * cnt
            order = np.argsort(ids)   # image_msg is a std::map: ascending id
            assert np.array_equal(rec[:, 0].astype(np.int64), ids[order]) and np.array_equal(rec[:, 1:], xyz[order])
            published += 1
        (ng,) = struct.unpack_from("<i", raw, off)
        off += 4
        assert ng > 20
    assert off == len(raw) and published == 3
    trk.close()


@pytest.mark.gpu
def test_vins_shim_follows_the_estimator(tmp_path):
    exe = build_shim(tmp_path)
    cfg = abi.default_config()
    W, n_frames = cfg.window_size, 26
    world = RS.SyntheticWorld(cfg, 4)
    blob = [struct.pack("<3i", W, n_frames, world.imu_per_frame), np.asarray(world.tic, np.float64).tobytes(),
            np.ascontiguousarray(world.ric, np.float64).tobytes()]
    loop = RS.EstimatorLoop(cfg, seed=4, init_noise=0.0, world=RS.SyntheticWorld(cfg, 4))   # same seed -> same data
    want = []
    init = []
    for k in range(n_frames):
        imu = [world.imu(world.time(0))] if k == 0 else world.imu_interval(k)
        ids, xyz = world.observe(k)
        blob.append(struct.pack("<d", world.time(k)))
        for a, g in imu:
            blob.append(struct.pack("<7d", world.dt, *a, *g))
        blob.append(struct.pack("<i", len(ids)))
        for i, p in zip(ids, xyz):
            blob.append(struct.pack("<i3d", i, *p))
        Pt, Rt, Vt = world.truth(k)
        if k <= W:
            init.append((world.time(k), Pt, Rt, Vt))
        if k == W:
            for t, P, R, V in init:
                blob.append(struct.pack("<d3d9d3d", t, *P, *np.asarray(R).ravel(), *V))
            blob.append(struct.pack("<3d3d", *world.ba, *world.bg))
        res = loop.step()
        want.append((res.action, loop.est.window()["Ps"][W].copy(), res.stats.iterations, res.stats.final_cost))
    loop.close()
    (tmp_path / "data.bin").write_bytes(b"".join(blob))
    subprocess.check_call([exe, "--vins", str(tmp_path / "data.bin"), str(tmp_path / "out.bin")])
    got = np.frombuffer((tmp_path / "out.bin").read_bytes(), np.float64).reshape(n_frames, 8)
    solved = 0
    for k in range(n_frames):
        act, P, it, fc = want[k]
        assert int(got[k, 1]) == act, (k, got[k, 1], act)
        if act == abi.VIO_FRAME_SOLVED:
            solved += 1
            # (two runs of the solver differ in the order of their LDS atomic accumulations: ~1e-9 m)
            assert np.abs(got[k, 3:6] - P).max() < 1e-7 and int(got[k, 6]) == it and abs(got[k, 7] - fc) <= 1e-7 * max(1.0, fc)
    assert solved == n_frames - W


@pytest.mark.gpu
def test_feature_tracker_shim_runs_the_vins_pnp_branch(tmp_path):
    """readImage with vins_normal (feature_tracker.cpp:207 -> :107-160): solved_features joined with the tracker's points,
    setInit(solved_vins), the frame's IMU samples, vinsPnP::processImage -- against the same sequence of ABI calls made
    from python. A static camera looking at landmarks five metres away, camera frame = body frame."""
    import ctypes as C
    exe = build_shim(tmp_path)
    rows, cols, n, first, ipf = 240, 320, 12, 3, 4
    frames, _ = pkg.synth.make_image_stream(9, n, rows=rows, cols=cols, max_shift=0.8)
    cfg = abi.default_config(max_corners=60, min_dist=25, image_rows=rows, image_cols=cols)
    lib = abi.load_product()
    trk = pkg.frontend.FeatureTracker(cfg, n_seq=1)
    pnp = pkg.pnp.PnpTracker(cfg, np.zeros(3), np.eye(3), pnp_size=6)
    acc, gyr = np.array([0.0, 0.0, cfg.gravity]), np.zeros(3)
    imu = [[(0.1 * f - 0.1 + 0.025 * (s + 1), acc, gyr) for s in range(ipf)] for f in range(n)]
    solved, want, current_time = None, [], -1.0
    _ip, _fp = C.POINTER(C.c_int32), C.POINTER(C.c_float)
    for f in range(n):
        trk.read_images(frames[f:f + 1], f % 3 == 0)
        pts, ids, _ = trk.state(0)
        ppts, pids = trk.pnp_points(0)          # the list at feature_tracker.cpp:207 (ahead of rejectWithF / setMask)
        if f % 3 != 0:                          # tracking-only frames drop nothing behind :207
            assert np.array_equal(pids, ids) and np.array_equal(ppts, pts)
        elif f > 0:                             # publishing frames: the kept tracks are a subset of it (new corners come behind)
            assert set(ids[np.isin(ids, pids)]) <= set(pids) and len(pids) >= np.isin(ids, pids).sum()
        if f == first - 1:   # the landmarks "the back-end has solved": the points tracked so far, 5 m in front of the camera
            order = np.argsort(ids)
            solved = [(int(ids[i]), 7, ((pts[i, 0] - cfg.cx) / cfg.fx * 5.0, (pts[i, 1] - cfg.cy) / cfg.fy * 5.0, 5.0)) for i in order]
        P, R = np.zeros(3), np.zeros((3, 3))
        if f >= first and f > 0:
            arr = (abi.VioPnpFeature * len(solved))()
            for k, (fid, tn, pos) in enumerate(solved):
                arr[k].id, arr[k].track_num = fid, tn
                arr[k].position[:] = [float(v) for v in pos]
            out, nm = (abi.VioPnpFeature * (cfg.max_corners + 1))(), C.c_int32()
            assert lib.vio_pnp_match_features(C.byref(cfg), pids.ctypes.data_as(_ip), ppts.ctypes.data_as(_fp), len(pids), arr, len(solved),
                                              out, cfg.max_corners, C.byref(nm)) == 0
            assert nm.value > 20
            pnp.set_init(0.1 * (first - 1), np.zeros(3), np.zeros(3), np.zeros(3), np.eye(3), np.zeros(3))
            for t, a, g in imu[f]:
                if current_time < 0:
                    current_time = t
                pnp.process_imu(t - current_time, a, g)
                current_time = t
            feats = [(out[i].id, tuple(out[i].observation), tuple(out[i].position), out[i].track_num) for i in range(nm.value)]
            Pq, Rq, _ = pnp.process_images([feats], [0.1 * f], use_pnp=True)
            P, R = Pq[0], Rq[0]
        want.append((P, R))
    trk.close(), pnp.close()
    (tmp_path / "frames.bin").write_bytes(np.ascontiguousarray(frames).tobytes())
    blob = [struct.pack("<3i", first, len(solved), ipf)]
    for fid, tn, pos in solved:
        blob.append(struct.pack("<2i3d", fid, tn, *pos))
    blob.append(struct.pack("<d3d3d3d9d3d", 0.1 * (first - 1), *np.zeros(3), *np.zeros(3), *np.zeros(3), *np.eye(3).ravel(), *np.zeros(3)))
    for f in range(n):
        for t, a, g in imu[f]:
            blob.append(struct.pack("<d3d3d", t, *a, *g))
    (tmp_path / "pnp.bin").write_bytes(b"".join(blob))
    subprocess.check_call([exe, "--pnp", str(tmp_path / "frames.bin"), str(rows), str(cols), str(n), str(tmp_path / "pnp.bin"),
                           str(tmp_path / "out.bin")])
    got = np.frombuffer((tmp_path / "out.bin").read_bytes(), np.float64).reshape(n, 12)
    moved = 0
    for f in range(n):
        P, R = want[f]
        assert np.abs(got[f, :3] - P).max() < 1e-9 and np.abs(got[f, 3:].reshape(3, 3) - R).max() < 1e-9, (f, got[f], P, R)
        moved += int(np.abs(R).max() > 0)
    assert moved == n - first                     # P / R are only written once vins_normal is set
    # the window fills after PNP_SIZE frames: from then on the solve runs and follows the view's slow drift over the texture
    # (a few pixels per frame at five metres: centimetres)
    assert np.abs(got[-1, :3]).max() < 0.3 and np.abs(got[-1, 3:].reshape(3, 3) - np.eye(3)).max() < 0.05
