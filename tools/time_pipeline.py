#!/usr/bin/env python3
"""Wall time of the WHOLE pipeline through the C ABI for many sequences, per camera frame:
    pageable host frames -> vio_frontend_read_images (gather, H2D, pyramid, KLT, RANSAC, mask, corners, D2H of the
    observations) -> vio_estimator_process_imu_batch -> vio_estimator_process_images (landmark bookkeeping, window assembly,
    pack, H2D, vio_window_kernel, D2H, slide) -> host states,
in the call order of the app (ViewController.mm:458 readImage, :688-724 processIMU / processImage). The frames are
rendered beforehand from textured planes filmed along synthetic trajectories (tools/replay_synthetic.py ImageWorld), so
the observations the estimator receives are the ones the front-end publishes and IMU / images are consistent.

    tools/time_pipeline.py [n_seq] [n_frames] [overlap] [freq] [registered]
        overlap = 1: the front-end of camera frame k+1 is submitted before the estimator of frame k runs
                     (vio_frontend_submit_images / vio_frontend_collect), 0: strictly one call after the other,
                     2: the same with vio_frontend_submit_images_async (the gathering of the pageable frames and the queueing
                     run on the context's own host thread, under the estimator call)
        freq    = 1: every camera frame is published and solved (the convention of bench.py's headline number),
                  3: the app's cadence, FREQ = 3 (global_param.hpp:33, ViewController.mm:467,494): the tracker runs on every
                     camera frame, every third one is published to the estimator; n_frames counts published frames
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import replay_synthetic as RS  # noqa: E402

pkg = RS.pkg
abi = pkg.abi
_dp, _ip, _u8p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


def run(n_seq=256, n_frames=22, overlap=True, n_worlds=4, quiet=False, freq=1, registered=False):
    cfg = abi.default_config(max_corners=150, min_dist=20)
    W, cap = cfg.window_size, cfg.max_corners
    # (published frames stay 0.1 s apart: the camera runs freq times faster, the IMU freq x 10 samples per published frame)
    worlds = [RS.ImageWorld(cfg, 300 + q, frame_dt=0.1 / freq, imu_per_frame=10 // freq if freq > 1 else 10) for q in range(min(n_worlds, n_seq))]
    nw = len(worlds)
    n_cam = (n_frames - 1) * freq + 1          # camera frames; camera frame c is published when c % freq == 0
    rendered = [[w.render(c) for c in range(n_cam)] for w in worlds]
    # one pageable buffer per camera frame holding the image of every sequence (sequence q films world q % nw)
    frames = [np.ascontiguousarray(np.stack([rendered[q % nw][c] for q in range(n_seq)])) for c in range(n_cam)]
    fe = pkg.frontend.FeatureTracker(cfg, n_seq=n_seq)
    if registered:
        for f in frames:
            fe.register_host(f)
    est = pkg.estimator.Estimator(cfg, worlds[0].tic, worlds[0].ric, n_seq=n_seq)
    lib = est.lib
    have_async = overlap and hasattr(lib, "vio_frontend_submit_images")
    obs = np.zeros(n_seq * cap, pkg.frontend._OBS_DTYPE)
    obs_p = C.cast(obs.ctypes.data, C.POINTER(abi.VioObs))
    n_obs = np.zeros(n_seq, np.int32)
    hdr = np.zeros(n_seq)
    res = (abi.VioFrameResult * n_seq)()
    rows, cols = cfg.image_rows, cfg.image_cols
    imu = []     # per published frame: the samples of its freq camera intervals
    for k in range(n_frames):
        per = []
        for w in worlds:
            smp = [w.imu(w.time(0))] if k == 0 else [x for c in range((k - 1) * freq + 1, k * freq + 1) for x in w.imu_interval(c)]
            per.append(smp)
        st = max(len(p) for p in per)
        ns = np.array([len(per[q % nw]) for q in range(n_seq)], np.int32)
        accs, gyrs = np.zeros((n_seq, st, 3)), np.zeros((n_seq, st, 3))
        for q in range(n_seq):
            for i, (a, g) in enumerate(per[q % nw]):
                accs[q, i], gyrs[q, i] = a, g
        imu.append((ns, np.full((n_seq, st), worlds[0].dt), accs, gyrs))

    def fe_call(c):
        rc = lib.vio_frontend_read_images(fe._h, frames[c].ctypes.data_as(_u8p), rows, cols, cols, None, int(c % freq == 0), obs_p,
                                          n_obs.ctypes.data_as(_ip))
        assert rc == 0, rc

    def fe_submit(c):
        submit = lib.vio_frontend_submit_images_async if int(overlap) == 2 else lib.vio_frontend_submit_images
        rc = submit(fe._h, frames[c].ctypes.data_as(_u8p), rows, cols, cols, int(c % freq == 0))
        assert rc == 0, rc

    def fe_collect():
        rc = lib.vio_frontend_collect(fe._h, obs_p, n_obs.ctypes.data_as(_ip))
        assert rc == 0, rc

    t_frame, t_fe, t_est, tracked, solved = [], [], [], [], 0
    if have_async:
        fe_submit(0)
    fe_acc = 0.0
    t0 = time.perf_counter()
    for c in range(n_cam):
        ta = time.perf_counter()
        if have_async:
            fe_collect()
            if c + 1 < n_cam:
                fe_submit(c + 1)          # its gather / H2D / kernels overlap the estimator call below
        else:
            fe_call(c)
        fe_acc += time.perf_counter() - ta
        if c % freq:
            continue
        k = c // freq
        t1 = time.perf_counter()
        est.process_imu_batch(*imu[k])
        skip = 0.0
        if k == W:   # hand-over of the first window in place of solveInitial (a plane is degenerate for the SfM start)
            ts = time.perf_counter()
            for q in range(n_seq):
                w = worlds[q % nw]
                tr = [w.truth(j * freq) for j in range(W + 1)]
                est.set_initial_state([w.time(j * freq) for j in range(W + 1)], [t[0] for t in tr], [t[1] for t in tr], [t[2] for t in tr],
                                      [w.ba] * (W + 1), [w.bg] * (W + 1), seq=q)
            skip = time.perf_counter() - ts      # (python-side hand-over loop: not the product)
        for q in range(n_seq):
            hdr[q] = worlds[q % nw].time(c)
        rc = lib.vio_estimator_process_images(est._h, obs_p, n_obs.ctypes.data_as(_ip), cap, hdr.ctypes.data_as(_dp), None, res)
        assert rc == 0, rc
        t2 = time.perf_counter()
        t_frame.append(t2 - t0 - skip), t_fe.append(fe_acc), t_est.append(t2 - t1 - skip)
        t0, fe_acc = t2, 0.0
        tracked.append(float(n_obs.mean()))
        if k >= W:
            acts = [r.action for r in res]
            assert all(a == abi.VIO_FRAME_SOLVED for a in acts), (k, acts[:8])
            solved += n_seq
    # the newest position of every sequence against its trajectory: the pipeline is solving the problem it was given
    errs = []
    for q in range(min(n_seq, 2 * nw)):
        win = est.window(q)
        errs.append(float(np.linalg.norm(win["Ps"][W] - worlds[q % nw].truth(n_cam - 1)[0])))
    if registered:
        for f in frames:
            fe.unregister_host(f)
    fe.close(), est.close()
    steady = slice(W + 3, n_frames)
    per_frame = float(np.mean(t_frame[steady]))
    out = {"sequences": n_seq, "published_frames_timed": len(t_frame[steady]), "camera_frames_per_published_frame": freq,
           "ms_per_published_frame_of_all_sequences": per_frame * 1e3,
           "camera_frames_per_s": n_seq * freq / per_frame, "solves_per_s": n_seq / per_frame,
           "ms_frontend_calls": float(np.mean(t_fe[steady])) * 1e3,
           "ms_estimator_calls": float(np.mean(t_est[steady])) * 1e3, "overlap": int(overlap) if have_async else 0,
           "registered_host_frames": bool(registered),
           "mean_published_features": float(np.mean(tracked[steady])), "position_error_m_max": max(errs),
           "ms_every_published_frame": [round(t * 1e3, 2) for t in t_frame]}
    if not quiet:
        print(out)
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]) if a else 256, int(a[1]) if len(a) > 1 else 22, int(a[2]) if len(a) > 2 else 1,
        freq=int(a[3]) if len(a) > 3 else 1, registered=bool(int(a[4])) if len(a) > 4 else False)
