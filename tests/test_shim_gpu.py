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
