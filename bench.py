#!/usr/bin/env python3
"""bench.py — VIO frames/sec (KLT + window solve) on MI355X, BASELINE.json's metric.

One step = one camera frame for every resident sequence through the whole hot path:
    FeatureTracker::readImage  (pyramid, pyramidal LK, F-RANSAC, setMask, Shi-Tomasi detection)   [publish frame]
    VINS::solve_ceres          (10-iteration dogleg solve, new2old, marginalization)
Workload = BASELINE.json configs[1]: 640x480 frames, up to 150 features, window 10, ~800 projection factors.
Every frame is published (the conservative reading of "KLT + window solve" per frame; the reference publishes every
3rd frame). Inputs (frames, windows) are resident in HBM before the timed region; sequences are independent, so N
GPUs run N x the sequences with no data-path collective (weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == matrix peak (MI355X_MICROARCH.md / SURVEY §8d)
HBM_PEAK_GBS = 8000.0
# Memory-side traffic of vio_window_kernel per window from the PMC passes of this very workload
# (profiles/r01_i_pmc_hbm.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs, 256 windows per launch):
# FETCH_SIZE 1004820 KB (doubled: gfx950 reports half the bytes, MI355X_MICROARCH.md "HBM") + WRITE_SIZE 952949 KB.
# bench.py cannot collect counters itself; the figure is only attached when the profiled configuration is the one run.
PMC_TRAFFIC_BYTES_PER_WINDOW = (2 * 1007830.6 + 953780.0) * 1024.0 / 256.0
PMC_TRAFFIC_CONFIG = (10, 150)  # (window size, features) the passes were taken on
# The same passes for the front-end kernels of one publish step of 256 sequences (sum over copy_frames, 3 x pyr_down,
# lk_track, track_update, detect, corner_select): FETCH_SIZE 523110 KB (doubled) + WRITE_SIZE 272870 KB.
PMC_FE_TRAFFIC_BYTES_PER_SEQUENCE = (2 * 523110.0 + 272870.0) * 1024.0 / 256.0


def algorithmic_flops_per_solve(W, M, n_prior, iters):
    """per GN iteration J^T J of projection / IMU / prior blocks, landmark Schur outer products,
    reduced Cholesky and factor evaluation; times the iterations actually run."""
    P = W + 1
    proj = M * 2 * 13 * 13 * 2
    imu = W * 15 * 30 * 30 * 2
    prior = 2.0 * n_prior ** 3 if n_prior else 0.0
    kf = max(2.0, M / 150.0 + 1.0)
    schur = 150 * (6 * kf) ** 2 * 2
    chol = (15 * P) ** 3 / 3.0
    evalf = M * 500 + W * 3000
    return iters * (proj + imu + prior + schur + chol + evalf)


def algorithmic_bytes_per_tracked_frame(rows, cols, n_feats, levels=4, mean_iters=10, publish=True):
    """SURVEY.md §8(d) B_track (+ B_det on publish frames)."""
    px = rows * cols
    b = px + 0.328 * px * 2 + n_feats * levels * (24 * 24 + mean_iters * 22 * 22)
    if publish:
        b += 10 * px
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sequences", type=int, default=256, help="independent sequences resident per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, choices=[1, 2], default=1,
                    help="1: tracker and solver kernels in order on one HIP stream; 2: each on its own stream")
    ap.add_argument("--publish-every", type=int, default=1,
                    help="aid: publish (detect + solve) only every N-th frame like the app's FREQ = 3; with N != 1 the "
                         "reported value is NOT the benchmark metric, which publishes and solves every frame")
    ap.add_argument("--only", choices=["both", "frontend", "backend"], default="both",
                    help="profiling aid: run one half alone (the reported value is then NOT the benchmark metric)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: same code path for every world size
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("vins-mobile_amd")
    abi, synth, backend, frontend = pkg.abi, pkg.synth, pkg.backend, pkg.frontend

    S = args.sequences
    cfg = abi.default_config(max_corners=150, min_dist=20)  # 150 features need MIN_DIST 20 at 640x480
    rows, cols = cfg.image_rows, cfg.image_cols

    # ---- synthetic inputs (seed = 42 + sequence id; a few unique streams / windows tiled over the batch) ----------
    # this rank owns global sequences rank, rank + world, ... (multi.sequences_of_rank); seeds follow the ids
    n_unique, T = 4, 4
    my_ids = pkg.multi.sequences_of_rank(S * world, rank, world)
    uniq_frames = [synth.make_image_stream(pkg.multi.seed_of_sequence(my_ids[u]), T, rows=rows, cols=cols)[0]
                   for u in range(n_unique)]
    frames = np.stack([np.stack([uniq_frames[s % n_unique][f] for s in range(S)]) for f in range(T)])
    pre = lambda *a: backend.preintegrate(cfg, *a)
    uniq_w = [synth.make_window(cfg, pre, seed=pkg.multi.seed_of_sequence(my_ids[u]), n_features=150) for u in range(8)]
    windows = [uniq_w[s % len(uniq_w)].copy() for s in range(S)]

    fe = frontend.FeatureTracker(cfg, n_seq=S)
    fe.upload_frames(frames)
    be = backend.WindowSolver(cfg, max_batch=S)
    be.upload(windows)
    pingpong = list(range(T)) + list(range(T - 2, 0, -1))  # consecutive frames stay adjacent in time

    # Both halves fill the chip on their own (the solver holds every CU's LDS), so running them back to back on one
    # stream beats letting two streams interleave their kernels.
    one = torch.cuda.Stream().cuda_stream if args.streams == 1 else None

    def step(k):
        publish = k % args.publish_every == 0
        if args.only != "backend":
            fe.step(pingpong[k % len(pingpong)], publish=publish, stream=one)
        if args.only != "frontend" and publish:
            be.launch(stream=one)

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    fe.kernel_ms(), be.kernel_ms()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        dt = pkg.multi.max_over_ranks(dist, dt, device="cuda")

    fe_ms, _ = fe.kernel_ms()
    be_ms, _ = be.kernel_ms()
    fe_ms, be_ms = max(fe_ms, 1e-9), max(be_ms, 1e-9)
    stats = be.download(windows)
    iters = float(np.mean([s["iterations"] - 1 for s in stats]))
    M = float(np.mean([w.n_factors for w in windows]))

    if rank == 0:
        frames_total = S * world * args.steps
        value = frames_total / dt
        flops = algorithmic_flops_per_solve(cfg.window_size, M, 0, iters) * S
        achieved = flops / (be_ms * 1e-3) / 1e12
        fe_bytes = algorithmic_bytes_per_tracked_frame(rows, cols, 150) * S
        out = {
            "metric": "VIO frames/sec (KLT+window solve), 640x480/150 feats/W=10" +
                      ("" if args.only == "both" else " [PARTIAL: %s only, not the metric]" % args.only),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (solve) / u8+i32+f32 (KLT)", "data": "synthetic",
            "config": {"workload": "configs[1]: 640x480 stream, 150 feats, window=10, ~%d projection factors; every frame "
                                   "published: KLT track + F-RANSAC + detect + 10-iteration window solve + marginalization" % M,
                       "sequences_per_gpu": S, "publish_every": args.publish_every,
                       "preprocessing": "none (the app's CLAHE pre-step is outside readImage)", "gn_iterations": iters, "kernel_ms": {"frontend_step": fe_ms, "window_solve": be_ms}},
            "roofline": {"kernel": "vio_window_kernel (solve + new2old + marginalization, one workgroup per window)",
                         "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS,
                         "traffic": PMC_TRAFFIC_BYTES_PER_WINDOW * S if (cfg.window_size, 150) == PMC_TRAFFIC_CONFIG else None,
                         "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_i_pmc_hbm.txt)"},
            "roofline_frontend": {"kernel": "front-end step (copy + pyr_down x3 + lk_track + track_update + detect + corner_select)",
                                  "bound": "hbm", "achieved": fe_bytes / (fe_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": fe_bytes / (fe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "traffic": PMC_FE_TRAFFIC_BYTES_PER_SEQUENCE * S
                                  if (rows, cols, cfg.max_corners, args.publish_every) == (640, 480, 150, 1) else None,
                                  "traffic_unit": "bytes per step (PMC FETCH_SIZE x2 + WRITE_SIZE summed over the step's kernels, "
                                                  "profiles/r01_i_pmc_hbm.txt)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, abi, uniq_frames[0], uniq_w)
        print(json.dumps(out))
    fe.close(), be.close()
    if dist:
        dist.destroy_process_group()


def cpu_baseline(cfg, abi, stream, uniq_windows):
    """The same per-frame work on ONE host core (the reference runs Ceres with num_threads = 1, VINS.cpp:642):
    front-end = CPU restatement (OpenCV is not available anywhere: 'port'); window solve = the real reference
    (vendored Ceres 1.12 + VINS factors, oracle/_ref) when its prebuilt library travelled here, else the restatement."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    trk = H.OracleTracker(cfg)
    t_fe, n_fe = 0.0, 0
    order = [0, 1, 2, 3, 2, 1]
    trk.read_image(stream[0], True)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 6.0:
        trk.read_image(stream[order[(n_fe + 1) % len(order)]], True)
        n_fe += 1
    t_fe = (time.perf_counter() - t0) / n_fe
    trk.close()
    ref = H.ref_lib_or_none()
    kind = "reference" if ref is not None else "port"
    solve = abi.bind_backend_solver(ref, "ref")[0] if ref is not None else H.oracle_backend()[0]
    devnull = os.open(os.devnull, os.O_WRONLY)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(devnull, 1)  # the reference printf()s from marginalization
    try:
        n_s, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 6.0:
            w = uniq_windows[n_s % len(uniq_windows)].copy()
            st = abi.VioSolveStats()
            solve(C.byref(cfg), C.byref(w.struct()), C.byref(st))
            n_s += 1
        t_solve = (time.perf_counter() - t0) / n_s
    finally:
        C.CDLL(None).fflush(None)  # the reference's printf()s sit in C stdio's buffer: drain them into /dev/null too
        os.dup2(saved, 1)
        os.close(devnull)
    return {"value": 1.0 / (t_fe + t_solve), "unit": "frames/s", "cores": 1,
            "kind": "port", "solve_kind": kind,
            "sample": "%d published frames through the KLT restatement (%.1f ms each) + %d window solves incl. "
                      "marginalization through %s (%.1f ms each), one thread, %s" % (
                          n_fe, t_fe * 1e3, n_s, "vendored Ceres 1.12 + VINS factors" if ref is not None else
                          "the C++ restatement", t_solve * 1e3, cpu_model()),
            "frontend_ms": t_fe * 1e3, "solve_ms": t_solve * 1e3}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip() + " (%d cores visible)" % os.cpu_count()
    except OSError:
        pass
    return "unknown CPU"


if __name__ == "__main__":
    main()
