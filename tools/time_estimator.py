#!/usr/bin/env python3
"""Wall time of vio_estimator_process_images for many sequences: host bookkeeping + pack/upload + ONE window-kernel
launch + download + slides. Observations and IMU are precomputed so that only library time is measured."""
import ctypes as C
import sys
import time
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import replay_synthetic as RS  # noqa: E402

pkg = RS.pkg
abi = pkg.abi


def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    n_est = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if n_est > 1:   # several estimator objects, one host thread each: their kernels overlap the others' host phases
        import threading
        t0 = time.perf_counter()
        res = [None] * n_est
        ths = [threading.Thread(target=lambda i=i: res.__setitem__(i, run(n_seq, n_frames, quiet=True))) for i in range(n_est)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        solves = sum(r[0] for r in res)
        span = max(r[2] for r in res) - min(r[1] for r in res)
        print("   (library time per estimator: %s s)" % ", ".join("%.3f" % r[3] for r in res))
        print("%d estimators x %d sequences on %d host threads: %d window solves in %.1f ms of solve phase -> %.0f solves/s aggregate"
              % (n_est, n_seq, n_est, solves, span * 1e3, solves / span))
        return
    run(n_seq, n_frames)


def run(n_seq, n_frames, quiet=False):
    n_worlds = min(n_seq, 8)     # distinct data sets, reused round-robin (python-side generation is the slow part)
    cfg = abi.default_config()
    W = cfg.window_size
    worlds = [RS.SyntheticWorld(cfg, 100 + q) for q in range(n_worlds)]
    est = pkg.estimator.Estimator(cfg, worlds[0].tic, worlds[0].ric, n_seq=n_seq)
    lib = est.lib
    stride = 160
    data = []
    for w in worlds:
        frames = []
        for k in range(n_frames):
            imu = [w.imu(w.time(0))] if k == 0 else w.imu_interval(k)
            ids, xyz = w.observe(k)
            frames.append((imu, ids, xyz, w.time(k), w.truth(k)))
        data.append(frames)
    obs = (abi.VioObs * (stride * n_seq))()
    n_obs = np.zeros(n_seq, np.int32)
    hdr = np.zeros(n_seq)
    res = (abi.VioFrameResult * n_seq)()
    t_img, t_imu, phases = [], [], []
    t_first_solve = t_last = None
    for k in range(n_frames):
        ns = np.array([len(data[q % n_worlds][k][0]) for q in range(n_seq)], np.int32)
        st = int(ns.max())
        dts = np.full((n_seq, st), worlds[0].dt)
        accs, gyrs = np.zeros((n_seq, st, 3)), np.zeros((n_seq, st, 3))
        for q in range(n_seq):
            for i, (a, g) in enumerate(data[q % n_worlds][k][0]):
                accs[q, i], gyrs[q, i] = a, g
        t0 = time.perf_counter()
        est.process_imu_batch(ns, dts, accs, gyrs)
        t_imu.append(time.perf_counter() - t0)
        for q in range(n_seq):
            _, ids, xyz, t, _ = data[q % n_worlds][k]
            for i in range(len(ids)):
                o = obs[q * stride + i]
                o.id, o.x, o.y, o.z = ids[i], xyz[i][0], xyz[i][1], xyz[i][2]
            n_obs[q], hdr[q] = len(ids), t
            if k == W:
                fr = data[q % n_worlds]
                wd = worlds[q % n_worlds]
                est.set_initial_state([f[3] for f in fr[:W + 1]], [f[4][0] for f in fr[:W + 1]], [f[4][1] for f in fr[:W + 1]],
                                      [f[4][2] for f in fr[:W + 1]], [wd.ba] * (W + 1), [wd.bg] * (W + 1), seq=q)
        t0 = time.perf_counter()
        rc = lib.vio_estimator_process_images(est._h, obs, n_obs.ctypes.data_as(C.POINTER(C.c_int32)), stride,
                                              hdr.ctypes.data_as(C.POINTER(C.c_double)), None, res)
        dt = time.perf_counter() - t0
        assert rc == 0, rc
        t_img.append(dt)
        ph = np.zeros(3)
        lib.vio_estimator_get_timing(est._h, ph.ctypes.data_as(C.POINTER(C.c_double)))
        phases.append(ph)
        if k >= W:
            assert all(r.action == abi.VIO_FRAME_SOLVED for r in res), [r.action for r in res][:8]
            if k == W + 2:
                t_first_solve = time.perf_counter()
            t_last = time.perf_counter()
    if quiet:
        est.close()
        lib_s = float(np.sum(t_img[W + 3:]) + np.sum(t_imu[W + 3:]))   # library time only (the python marshalling between calls is not the product)
        return n_seq * (n_frames - W - 3), t_first_solve, t_last, lib_s
    solve = np.array(t_img[W + 2:]) * 1e3
    fill = np.array(t_img[1:W]) * 1e3
    print("n_seq %d: process_images with a solve: %.2f ms median (%.2f min) -> %.0f window solves/s end to end; filling phase %.2f ms"
          % (n_seq, np.median(solve), solve.min(), n_seq / np.median(solve) * 1e3, np.median(fill)))
    ph = np.median(np.array(phases[W + 2:]), axis=0)
    print("   phases (median ms): before-solve bookkeeping %.2f, solve_windows (pack+H2D+kernel+D2H) %.2f, after-solve %.2f; IMU feed %.2f ms/frame"
          % (ph[0], ph[1], ph[2], np.median(t_imu[1:]) * 1e3))
    est.close()


if __name__ == "__main__":
    main()
