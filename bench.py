#!/usr/bin/env python3
"""bench.py — VIO frames/sec (KLT + window solve) on MI355X, BASELINE.json's metric.

One step = one camera frame for every resident sequence through the whole hot path:
    FeatureTracker::readImage  (pyramid, pyramidal LK, F-RANSAC, setMask, Shi-Tomasi detection)   [publish frame]
    VINS::solve_ceres          (10-iteration dogleg solve, new2old, marginalization)
Workload = BASELINE.json configs[1]: 640x480 frames, up to 150 features, window 10, ~800 projection factors, and — as
every steady-state solve_ceres call has (VINS.cpp:508-513) — the marginalization prior left by the preceding MARGIN_OLD
solve. Every frame is published (the conservative reading of "KLT + window solve" per frame; the reference publishes
every 3rd frame). Inputs (frames, windows) are resident in HBM before the timed region; sequences are independent, so
N GPUs run N x the sequences with no data-path collective (weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Besides the contract line's `roofline` (window kernel; `roofline.frontend` = the front-end step against HBM) and `cpu_baseline`, rank 0
at N = 1 runs secondary measurements (bounded, ~2 minutes in total; `--quick` skips them): `end_to_end` (the estimator path: host
observations in, host states out), `end_to_end_full` (frames in, states out), `ate` (closed-loop position error against the truth
and against the CPU path on the same data), `large_windows` (kernel time of configs[2] and configs[4]), `configs2` / `configs4`
(those configs end to end, at 64 and at 256 sequences), `loop_closure` (pose graph solve and descriptor matching, SURVEY 8f rank
4). Their full records are ONE JSON line on stderr prefixed `SECONDARY ` (and `--secondary-out FILE`), printed BEFORE the contract
line; the contract line -- the last line of stdout -- carries their numbers in `config.secondary`.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == matrix peak (MI355X_MICROARCH.md / SURVEY §8d)
HBM_PEAK_GBS = 8000.0
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc.json")   # written by tools/rocpd_pmc_summary.py from the rocprofv3 --pmc passes


def algorithmic_flops_per_solve(W, M, n_prior, iters, n_features=150):
    """per GN iteration J^T J of projection / IMU / prior blocks, landmark Schur outer products,
    reduced Cholesky and factor evaluation; times the iterations actually run."""
    P = W + 1
    proj = M * 2 * 13 * 13 * 2
    imu = W * 15 * 30 * 30 * 2
    prior = 2.0 * n_prior ** 3 if n_prior else 0.0
    kf = max(2.0, M / float(n_features) + 1.0)
    schur = n_features * (6 * kf) ** 2 * 2
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


def pmc_traffic(kernel_key, workload_key):
    """HBM bytes per launch from the tracked PMC summary (FETCH_SIZE x correction + WRITE_SIZE), or None when the file
    is absent or was taken on another workload."""
    try:
        d = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return None, None
    k = d.get(kernel_key)
    if not k or d.get("workload") != workload_key:
        return None, None
    return k.get("bytes_per_launch"), "profiles/%s: %s" % (os.path.basename(PMC_FILE), d.get("method", ""))


def steady_state_windows(cfg, pkg, pre, seeds, n_features=150, with_loop=0, imu_per_frame=10):
    """configs[1] windows as solve_ceres sees them in steady state: window A of a sequence is solved with MARGIN_OLD on
    the device and the prior it leaves is carried by window B, one frame later on the same trajectory (how
    tests/golden/make_golden.py builds win_chain_b_prior; SURVEY §8d C5 'a prior produced by a preceding MARGIN_OLD step')."""
    synth, backend = pkg.synth, pkg.backend
    first = [synth.make_window(cfg, pre, seed=1000 + s, traj_seed=s, frame_offset=0, n_features=n_features,
                               imu_per_frame=imu_per_frame) for s in seeds]
    solver = backend.WindowSolver(cfg, max_batch=len(first))
    solver.solve(first)
    solver.close()
    out = []
    for s, a in zip(seeds, first):
        b = synth.make_window(cfg, pre, seed=2000 + s, traj_seed=s, frame_offset=1, n_features=n_features,
                              imu_per_frame=imu_per_frame, with_loop=with_loop)
        assert a.next_prior.n > 0
        b.prior = a.next_prior.copy()
        out.append(b)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sequences", type=int, default=512,
                    help="independent sequences resident per GPU (512 = two windows per CU: the window kernel keeps two "
                         "256-thread workgroups resident per CU; `resident_256` reports the one-window-per-CU batch too)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="contract line only (no secondary measurements)")
    ap.add_argument("--no-prior", action="store_true", help="aid: first-solve windows without a marginalization prior")
    ap.add_argument("--streams", type=int, choices=[1, 2], default=1,
                    help="1: tracker and solver kernels in order on one HIP stream; 2: each on its own stream")
    ap.add_argument("--publish-every", type=int, default=1,
                    help="aid: publish (detect + solve) only every N-th frame like the app's FREQ = 3; with N != 1 the "
                         "reported value is NOT the benchmark metric, which publishes and solves every frame")
    ap.add_argument("--leg", choices=["configs2", "configs4"], default=None,
                    help="profiling aid: run ONE secondary leg (BASELINE configs[2] / configs[4] end to end) and print its record")
    ap.add_argument("--secondary-out", default=None, help="also write the secondary measurements (one JSON object) to this file")
    ap.add_argument("--only", choices=["both", "frontend", "backend"], default="both",
                    help="profiling aid: run one half alone (the reported value is then NOT the benchmark metric)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    torch.cuda.set_device(local_rank)   # the contexts below bind to this device (vio_amd.h, DEVICE BINDING)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: same code path for every world size
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("vins-mobile_amd")
    abi, synth, backend, frontend = pkg.abi, pkg.synth, pkg.backend, pkg.frontend
    if args.leg:
        print(json.dumps(config_leg(pkg, {"configs2": "configs[2]", "configs4": "configs[4]"}[args.leg], steps=args.steps, warmup=args.warmup,
                                    cpu_frames=0 if args.no_cpu_baseline else 50)))
        return

    S = args.sequences
    cfg = abi.default_config(max_corners=150, min_dist=20)  # 150 features need MIN_DIST 20 at 640x480
    if os.environ.get("VIO_BENCH_MAX_ITER"):  # diagnostics only (tools/profile_round.sh: instructions per trust-region iteration = the
        cfg.max_iterations = int(os.environ["VIO_BENCH_MAX_ITER"])  # difference of two PMC passes); not the metric's config
    if os.environ.get("VIO_BENCH_LK_ITERS"):  # diagnostics only (tools/fe_iters.sh): what the LK iterations cost; not the metric's config
        cfg.lk_max_iters = int(os.environ["VIO_BENCH_LK_ITERS"])
    rows, cols = cfg.image_rows, cfg.image_cols

    # ---- synthetic inputs (seed = 42 + sequence id; a few unique streams / windows tiled over the batch) ----------
    # this rank owns global sequences rank, rank + world, ... (multi.sequences_of_rank); seeds follow the ids
    n_unique, T = 4, 4
    my_ids = pkg.multi.sequences_of_rank(S * world, rank, world)
    uniq_frames = [synth.make_image_stream(pkg.multi.seed_of_sequence(my_ids[u]), T, rows=rows, cols=cols)[0]
                   for u in range(n_unique)]
    frames = np.stack([np.stack([uniq_frames[s % n_unique][f] for s in range(S)]) for f in range(T)])
    pre = lambda *a: backend.preintegrate(cfg, *a)
    seeds = [pkg.multi.seed_of_sequence(my_ids[u]) for u in range(8)]
    if args.no_prior:
        uniq_w = [synth.make_window(cfg, pre, seed=s, n_features=150) for s in seeds]
    else:
        uniq_w = steady_state_windows(cfg, pkg, pre, seeds)
    windows = [uniq_w[s % len(uniq_w)].copy() for s in range(S)]
    n_prior = int(np.mean([w.prior.n if w.prior is not None else 0 for w in uniq_w]))

    def run_resident(S_, steps, warmup, timed_barrier, detect_always=True):
        """One resident batch of S_ sequences per GPU through `steps` timed steps: (seconds, front-end ms, solver ms, stats).
        detect_always: goodFeaturesToTrack runs on every published frame of every sequence (DETECT_NOTE)."""
        fr = frames[:, :S_] if S_ <= S else None
        ws = windows[:S_]
        with detect_every_frame(detect_always):
            fe = frontend.FeatureTracker(cfg, n_seq=S_)
        fe.upload_frames(np.ascontiguousarray(fr))
        be = backend.WindowSolver(cfg, max_batch=S_)
        be.upload(ws)
        if world > 1:  # the contexts of this rank live on this rank's GPU (vio_amd.h, DEVICE BINDING)
            assert be.device() == local_rank and fe.device() == local_rank, (be.device(), fe.device(), local_rank)
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

        for k in range(warmup):
            step(k)
        torch.cuda.synchronize()
        fe.kernel_ms(), be.kernel_ms()
        if dist and timed_barrier:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(warmup + k)
        torch.cuda.synchronize()
        if dist and timed_barrier:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        fe_ms_, _ = fe.kernel_ms()
        be_ms_, _ = be.kernel_ms()
        stats_ = be.download(ws)
        if args.only != "backend" and S_ == S and not lk_stats and not args.quick:  # (--quick: profiling runs keep their kernel trace clean)
            lk_stats.extend(measured_lk_iterations(fe, lambda k: fe.step(pingpong[(warmup + steps + k) % len(pingpong)], publish=True, stream=one)))
        fe.close(), be.close()
        return dt_, max(fe_ms_, 1e-9), max(be_ms_, 1e-9), stats_

    lk_stats = []  # (mean LK iterations per (feature, level) visit, per level) of the headline batch, measured once

    dt, fe_ms, be_ms, stats = run_resident(S, args.steps, args.warmup, True)
    replicas = None
    if dist:
        # one launch of the job answers for both batch sizes: the headline (S sequences per GPU) and BASELINE.json's configs[3]
        # read literally (64 sequences over 8 GPUs = 8 per GPU); every rank's own step time rides along
        replicas = {"headline": pkg.multi.replica_report(dist, S, args.steps, dt, device="cuda")}
        dt8 = run_resident(8, args.steps, args.warmup, True)[0]
        replicas["configs3_literal"] = pkg.multi.replica_report(dist, 8, args.steps, dt8, device="cuda")
        dt = pkg.multi.max_over_ranks(dist, dt, device="cuda")
    iters = float(np.mean([s["iterations"] - 1 for s in stats]))
    M = float(np.mean([w.n_factors for w in windows]))

    multi_gpu = None
    if dist:
        multi_gpu = multi_gpu_proof(pkg, dist, cfg, uniq_w[0], my_ids, pre, rank, local_rank, world)

    if rank == 0:
        frames_total = S * world * args.steps
        value = frames_total / dt
        flops = algorithmic_flops_per_solve(cfg.window_size, M, n_prior, iters) * S
        achieved = flops / (be_ms * 1e-3) / 1e12
        fe_bytes = algorithmic_bytes_per_tracked_frame(rows, cols, 150) * S
        wkey = "configs[1] x %d sequences, prior %d" % (S, n_prior)
        be_traffic, be_src = pmc_traffic("vio_window_kernel", wkey)
        fe_traffic, fe_src = pmc_traffic("frontend_step", wkey)
        out = {
            "metric": "VIO frames/sec (KLT+window solve), 640x480/150 feats/W=10" +
                      ("" if args.only == "both" else " [PARTIAL: %s only, not the metric]" % args.only),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (solve) / u8+i32+f32 (KLT)", "data": "synthetic",
            "config": {"workload": "configs[1]: 640x480 stream, 150 feats, window=10, ~%d projection factors, marginalization "
                                   "prior of %d rows from the preceding MARGIN_OLD solve; every frame published: KLT track + "
                                   "F-RANSAC + detect + 10-iteration window solve + marginalization" % (M, n_prior),
                       "sequences_per_gpu": S, "publish_every": args.publish_every, "prior_rows": n_prior,
                       "preprocessing": "none (the app's CLAHE pre-step is outside readImage)", "gn_iterations": iters,
                       "detection": DETECT_NOTE,
                       "kernel_ms": {"frontend_step": fe_ms, "window_solve": be_ms},
                       "hip_runtime": abi.hip_runtime()},
            "roofline": {"kernel": "vio_window_kernel (solve + new2old + marginalization, one workgroup per window; candidates linearized speculatively)",
                         "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS,
                         "flops_per_solve": flops / S, "traffic": be_traffic,
                         "traffic_unit": "bytes per launch; " + (be_src or "no PMC summary for this workload under profiles/")},
        }
        out["roofline"]["frontend"] = {"kernel": "front-end step (copy_frames + pyr_down x3 + lk_track + track_update + detect + corner_select)",
                                  "bound": "hbm", "achieved": fe_bytes / (fe_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": fe_bytes / (fe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "traffic": fe_traffic,
                                  # the byte formula assumes 10 LK iterations per (feature, level); what the kernel actually ran:
                                  "lk_mean_iterations": lk_stats[0] if lk_stats else None,
                                  "lk_mean_iterations_per_level": lk_stats[1] if lk_stats else None,
                                  "frac_at_measured_lk_iterations": (algorithmic_bytes_per_tracked_frame(rows, cols, 150, mean_iters=lk_stats[0]) * S / (fe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if lk_stats and lk_stats[0] else None,
                                  "traffic_unit": "bytes per step; " + (fe_src or "no PMC summary for this workload under profiles/")}
        if multi_gpu is not None:
            multi_gpu["replicas"] = replicas
            out["multi_gpu"] = multi_gpu
        extras = world == 1 and args.only == "both" and not args.quick
        if world == 1 and not args.no_cpu_baseline:
            base = CpuBaseline(cfg, abi, uniq_frames, uniq_w)
            out["cpu_baseline"] = base.one_core(warmup_frames=20, frames=200)
        if extras:
            # Secondary measurements: the full records go to stderr (one JSON line, `--secondary-out` also writes them to a file) AHEAD of
            # the contract line; the contract line carries their numbers, compact, in config.secondary -- a reader of the driver's record
            # needs nothing else.
            sec = {}
            if not args.no_cpu_baseline:
                sec["cpu_baseline_all_cores"] = guarded(lambda: base.all_cores(seconds=6.0))
            if S != 256:
                def at_256():
                    dt2, fe2, be2, _ = run_resident(256, args.steps, args.warmup, False)
                    fl = algorithmic_flops_per_solve(cfg.window_size, M, n_prior, iters) * 256
                    return {"sequences_per_gpu": 256, "value": 256 * args.steps / dt2, "unit": "frames/s", "ms_per_step": dt2 / args.steps * 1e3,
                            "kernel_ms": {"frontend_step": fe2, "window_solve": be2},
                            "roofline_frac_window_kernel": fl / (be2 * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                            "note": "one window per CU: the second resident workgroup of every CU stays empty"}
                sec["resident_256"] = guarded(at_256)
            def as_tracked():
                dt2, fe2, be2, _ = run_resident(S, args.steps, args.warmup, False, detect_always=False)
                return {"value": S * args.steps / dt2, "unit": "frames/s", "ms_per_step": dt2 / args.steps * 1e3,
                        "kernel_ms": {"frontend_step": fe2, "window_solve": be2},
                        "note": "the headline's loop with the product's default: detect_kernel returns at once for a sequence that still tracks "
                                "MAX_CNT features (most of this stream's frames); not the metric's value"}
            sec["frontend_as_tracked"] = guarded(as_tracked)
            sec["small_batches"] = guarded(lambda: small_batches(cfg, pkg, windows))
            sec["end_to_end"] = guarded(lambda: end_to_end(S if S >= 64 and S % 2 == 0 else 512))
            sec["end_to_end_full"] = guarded(lambda: end_to_end_full(256))
            sec["ate"] = guarded(lambda: closed_loop_ate(cfg, pkg))
            sec["large_windows"] = guarded(lambda: large_windows(pkg))
            for key, name in (("configs2", "configs[2]"), ("configs4", "configs[4]")):
                sec[key] = guarded(lambda: config_leg(pkg, name))
                # (the same leg where the chip is full: 64 cooperative windows leave CUs idle through the owner's serial phases)
                sec[key + "_256"] = guarded(lambda: config_leg(pkg, name, S=256, steps=6, warmup=2, cpu_frames=0))
            sec["loop_closure"] = guarded(lambda: loop_closure(pkg))
            out["config"]["secondary"] = compact_secondary(sec)
            line = json.dumps(sec)
            print("SECONDARY " + line, file=sys.stderr, flush=True)
            if args.secondary_out:
                with open(args.secondary_out, "w") as f:
                    f.write(line + "\n")
        sys.stderr.flush()
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


DETECT_NOTE = ("goodFeaturesToTrack runs on every published frame of every sequence (VIO_AMD_DETECT_ALWAYS=1, read when the tracker is "
               "created): the bench's 4-frame ping-pong stream loses no feature, a moving camera loses some in every frame. The product "
               "skips the call for a sequence that still tracks MAX_CNT features, as feature_tracker.cpp:256-266 does (n_max_cnt <= 0); "
               "`frontend_as_tracked` is this stream with that skip")


class detect_every_frame:
    """Context: trackers created inside run detect_kernel for every sequence on every published frame (see DETECT_NOTE)."""
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("VIO_AMD_DETECT_ALWAYS")
        if self.on:
            os.environ["VIO_AMD_DETECT_ALWAYS"] = "1"
        else:
            os.environ.pop("VIO_AMD_DETECT_ALWAYS", None)

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("VIO_AMD_DETECT_ALWAYS", None)
        else:
            os.environ["VIO_AMD_DETECT_ALWAYS"] = self.old


def guarded(fn):
    try:
        return fn()
    except Exception as e:  # a secondary measurement must not take the contract line down
        return {"error": "%s: %s" % (type(e).__name__, e)}


def compact_secondary(sec):
    """Numbers only (< 1 KB): what a reader of the contract line needs from the secondary legs."""
    def g(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return round(d, 4) if isinstance(d, float) else d
    c = {}
    for key in ("configs2", "configs4", "configs2_256", "configs4_256"):
        c[key] = {"frames_per_s": g(sec, key, "value"), "window_ms": g(sec, key, "kernel_ms", "window_solve"),
                  "frontend_ms": g(sec, key, "kernel_ms", "frontend_step"), "frac": g(sec, key, "roofline", "frac"),
                  "frac_frontend": g(sec, key, "roofline", "frontend", "frac_at_measured_lk_iterations")}
    c["end_to_end_full"] = {"256": g(sec, "end_to_end_full", "value"), "512": g(sec, "end_to_end_full", "at_512_sequences", "camera_frames_per_s"),
                            "freq3_256": g(sec, "end_to_end_full", "app_cadence_freq3", "camera_frames_per_s"),
                            "512_registered_host_frames": g(sec, "end_to_end_full", "registered_host_frames_at_512_sequences", "camera_frames_per_s")}
    c["end_to_end_solves_per_s"] = g(sec, "end_to_end", "value")
    c["resident_256"] = {"frames_per_s": g(sec, "resident_256", "value"), "window_ms": g(sec, "resident_256", "kernel_ms", "window_solve"),
                         "frac": g(sec, "resident_256", "roofline_frac_window_kernel")}
    c["small_batches_ms"] = {"B1": g(sec, "small_batches", "B=1", "kernel_ms"), "B8": g(sec, "small_batches", "B=8", "kernel_ms"),
                             "B64": g(sec, "small_batches", "B=64", "kernel_ms")}
    c["frontend_as_tracked"] = {"frames_per_s": g(sec, "frontend_as_tracked", "value"), "frontend_ms": g(sec, "frontend_as_tracked", "kernel_ms", "frontend_step")}
    c["ate_m"] = {"gpu_vs_truth": g(sec, "ate", "ate_gpu_vs_truth"), "cpu_vs_truth": g(sec, "ate", "ate_cpu_vs_truth"), "gpu_vs_cpu": g(sec, "ate", "rmse_gpu_vs_cpu")}
    c["cpu_all_cores_frames_per_s"] = g(sec, "cpu_baseline_all_cores", "value")
    c["errors"] = [k for k, v in sec.items() if isinstance(v, dict) and "error" in v]
    return c


# ---- CPU path timed beside it: one thread per sequence like the reference (num_threads = 1, VINS.cpp:642) --
class CpuBaseline:
    """front-end = the CPU restatement (OpenCV is not available anywhere: 'port'); window solve = the real reference
    (vendored Ceres 1.12 + VINS factors, oracle/_ref) when its prebuilt library travelled here, else the restatement."""

    def __init__(self, cfg, abi, streams, uniq_windows):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import helpers as H
        self.H, self.cfg, self.abi, self.streams, self.windows = H, cfg, abi, streams, uniq_windows
        ref = H.ref_lib_or_none()
        self.kind = "reference" if ref is not None else "port"
        self.solve = abi.bind_backend_solver(ref, "ref")[0] if ref is not None else H.oracle_backend()[0]

    def _worker(self, idx, seconds, out, warmup_frames=0, frames=None):
        """Runs for `seconds`, or -- when `frames` is given -- for exactly that many frames after `warmup_frames` untimed ones."""
        H, abi = self.H, self.abi
        trk = H.OracleTracker(self.cfg)
        stream = self.streams[idx % len(self.streams)]
        order = [0, 1, 2, 3, 2, 1]
        trk.read_image(stream[0], True)
        n, t_fe, t_so, k = 0, 0.0, 0.0, 0
        t_end = time.perf_counter() + seconds
        while (n < frames) if frames is not None else (time.perf_counter() < t_end):
            t0 = time.perf_counter()
            trk.read_image(stream[order[(k + 1) % len(order)]], True)
            t1 = time.perf_counter()
            w = self.windows[(idx + k) % len(self.windows)].copy()
            st = abi.VioSolveStats()
            self.solve(C.byref(self.cfg), C.byref(w.struct()), C.byref(st))
            t2 = time.perf_counter()
            k += 1
            if k <= warmup_frames:
                continue
            t_fe += t1 - t0
            t_so += t2 - t1
            n += 1
        trk.close()
        out[idx] = (n, t_fe, t_so)

    def _run(self, threads, seconds, **kw):
        devnull = os.open(os.devnull, os.O_WRONLY)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(devnull, 1)  # the reference printf()s from marginalization
        out = [None] * threads
        try:
            t0 = time.perf_counter()
            ths = [threading.Thread(target=self._worker, args=(i, seconds, out), kwargs=kw) for i in range(threads)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            wall = time.perf_counter() - t0
        finally:
            C.CDLL(None).fflush(None)  # the reference's printf()s sit in C stdio's buffer: drain them into /dev/null too
            os.dup2(saved, 1)
            os.close(devnull)
        return out, wall

    def one_core(self, warmup_frames=20, frames=200):
        """frames processed / wall time over >= 200 frames after 20 warm-up frames, one thread."""
        out, _ = self._run(1, 0.0, warmup_frames=warmup_frames, frames=frames)
        n, t_fe, t_so = out[0]
        return {"value": n / (t_fe + t_so), "unit": "frames/s", "cores": 1, "kind": "port", "solve_kind": self.kind,
                "sample": "after 20 warm-up frames: %d published frames through the KLT restatement (%.1f ms each) + %d steady-state window solves incl. "
                          "prior and marginalization through %s (%.1f ms each), one thread, %s" % (
                              n, t_fe / n * 1e3, n, "vendored Ceres 1.12 + VINS factors" if self.kind == "reference" else
                              "the C++ restatement", t_so / n * 1e3, cpu_model()),
                "frontend_ms": t_fe / n * 1e3, "solve_ms": t_so / n * 1e3}

    def all_cores(self, seconds):
        cores = usable_cores()
        out, wall = self._run(cores, seconds)
        n = sum(o[0] for o in out)
        return {"value": n / wall, "unit": "frames/s", "cores": cores, "kind": "port", "solve_kind": self.kind,
                "sample": "%d threads (one sequence each: KLT restatement frame + window solve through %s), %d frames in %.1f s; "
                          "%s" % (cores, "vendored Ceres" if self.kind == "reference" else "the C++ restatement", n, wall, cpu_model()),
                "per_thread_frames_per_s": n / wall / cores}


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # container CPU quota (cgroup v2): "max 100000" or "<quota> <period>"
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip() + " (%d cores visible, %d usable)" % (os.cpu_count(), usable_cores())
    except OSError:
        pass
    return "unknown CPU"


def multi_gpu_proof(pkg, dist, cfg, first_window, my_ids, pre, rank, local_rank, world):
    """Evidence that an N-GPU run is N independent replicas on N devices: every rank's device / PCI bus id, the
    collective backend's world size, and the solved poses of each rank's first sequence gathered over RCCL and compared, on
    rank 0, with a solve of the same sequences on rank 0's own GPU (same seeds -> same windows -> same poses)."""
    import torch
    be = pkg.backend.WindowSolver(cfg, max_batch=1)
    mine = first_window.copy()   # (unsolved: the steady-state window of this rank's first sequence)
    be.solve([mine])
    props = torch.cuda.get_device_properties(local_rank)
    info = {"rank": rank, "local_rank": local_rank, "device": be.device(), "name": props.name,
            "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
            "first_sequence": int(my_ids[0]),
            # threads of this process's host pool: its share of the node's CPUs / quota (csrc/vio_pool.h: divided by LOCAL_WORLD_SIZE)
            "host_pool_width": pkg.abi.host_pool_width()[0], "local_world_size": int(os.environ.get("LOCAL_WORLD_SIZE", "1"))}
    infos = [None] * world
    dist.all_gather_object(infos, info)
    poses = pkg.multi.all_gather_array(dist, np.asarray(mine.pose, np.float64).ravel(), device="cuda")
    be.close()
    if rank != 0:
        return None
    # rank 0 rebuilds the first sequence of every rank (seed = 42 + global id) and solves them on its own GPU
    seeds = [pkg.multi.seed_of_sequence(i["first_sequence"]) for i in infos]
    again = steady_state_windows(cfg, pkg, pre, seeds)
    chk = pkg.backend.WindowSolver(cfg, max_batch=len(again))
    chk.solve(again)
    chk.close()
    err = max(float(np.abs(np.asarray(a.pose).ravel() - p).max()) for a, p in zip(again, poses))
    assert sorted(i["device"] for i in infos) == list(range(world)), infos
    assert err < 1e-7, "gathered poses differ from a single-GPU solve of the same sequences: %g" % err
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": infos,
            "gathered_first_sequence_poses_max_abs_diff_vs_rank0_solve": err,
            "ownership": "global sequence id % world == rank (vins-mobile_amd/multi.py)"}


def small_batches(cfg, pkg, windows):
    """configs[3] read literally (8 sequences per GPU) and other launches that cannot fill the chip: window-kernel time."""
    out = {}
    for B in (1, 8, 64):
        be = pkg.backend.WindowSolver(cfg, max_batch=B)
        be.upload(windows[:B])
        be.launch()
        be.sync()
        be.kernel_ms()
        for _ in range(5):
            be.launch()
        be.sync()
        ms, _ = be.kernel_ms()
        be.close()
        out["B=%d" % B] = {"kernel_ms": ms, "solves_per_s": B / (ms * 1e-3)}
    return out


# ---- secondary measurements ------------------------------------------------------------------------------------
def _tool(code, env=None):
    """Runs a tools/ measurement in a fresh interpreter WITHOUT torch and returns the JSON it prints last. The library
    binds to whichever libamdhip64 the process loaded first: next to torch that is torch's bundled ROCm 7.0 runtime, on
    which the estimator path (many small transfers and launches per frame) measures ~10 % slower than on /opt/rocm's 7.2.
    A C++ caller of the C ABI has no torch in its process; the contract line above is kernel time and unaffected."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import json, sys; sys.path.insert(0, %r); " % os.path.join(ROOT, "tools") + code],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **(env or {})))
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-400:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def cpp_driver(n_seq, n_objects):
    """tools/estimator_throughput.cpp: the same path driven from C++ (no interpreter in the loop), `n_objects` estimator objects
    of `n_seq` sequences each on a host thread of their own -- while one object's host code runs, the other's kernels do."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("g++"):
        return {"skipped": "no g++ on this machine"}
    lib = os.path.join(ROOT, "vins-mobile_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        exe, data = os.path.join(tmp, "estimator_throughput"), os.path.join(tmp, "est.bin")
        subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "estimator_throughput.cpp"),
                        "-L" + lib, "-lvio_amd", "-Wl,-rpath," + lib, "-lpthread", "-o", exe], check=True, capture_output=True, timeout=300)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "estimator_dataset.py"), data, "8", "60"], check=True, capture_output=True,
                       timeout=300, cwd=ROOT)
        r = subprocess.run([exe, data, str(n_seq), str(n_objects)], check=True, capture_output=True, text=True, timeout=300)
    line = r.stdout.strip().splitlines()[-1]
    import re
    m = re.search(r"(\d+) window solves in ([0-9.]+) ms", line)
    n, ms = int(m.group(1)), float(m.group(2))
    return {"value": n / ms * 1e3, "unit": "window solves/s", "estimator_objects": n_objects, "sequences_each": n_seq, "window_solves": n,
            "driver": "tools/estimator_throughput.cpp (C++, one host thread per object), host buffers in / host states out"}


def end_to_end(n_seq):
    """The estimator path (csrc/vio_estimator.cpp): per frame, host observations + IMU in, host states out. Sequences in the
    NON_LINEAR state keep their landmark lists on the device (vio_estimator_set_resident, the default): observations and
    propagated states up, store_ingest / store_pack / ONE window-kernel launch for all sequences / store_finish, states
    back. `host_side_lists`: the same with the lists, the window assembly and the packing on the host (VIO_AMD_RESIDENT=0)."""
    run = lambda n, frames, env=None: _tool("import time_estimator as TE; s, _, _, l = TE.run(%d, %d, quiet=True); print(json.dumps([s, l]))"
                                            % (n, frames), env)
    half, a, c = run(n_seq // 2, 40), run(n_seq, 36), run(2 * n_seq, 30)   # (enough frames behind the one-time allocations of the first solves)
    h = run(n_seq // 2, 40, {"VIO_AMD_RESIDENT": "0", "VIO_AMD_HOST_THREADS": "64"})   # (its own best pool width)
    solves, lib_s = a
    per = lambda r, n: {"value": r[0] / r[1], "ms_per_frame_of_all_sequences": r[1] / (r[0] // n) * 1e3}
    two = guarded(lambda: cpp_driver(n_seq, 2))
    return {"value": solves / lib_s, "unit": "window solves/s (= published frames/s of the back-end half)", "sequences": n_seq,
            "two_estimator_objects": two,
            "frames_timed": solves // n_seq, "path": "vio_estimator_process_imu_batch + vio_estimator_process_images, one estimator "
            "object on one host thread, host buffers in / host states out; landmark lists, pre-integration blocks and priors "
            "resident on the device, window assembly by kernels (store_core.h); time inside the two library calls "
            "(closed-loop windows: ~190 landmarks, ~1400 factors, prior); measured in a process of its own; `value` is at the "
            "headline's sequence count (config.sequences), the other counts beside it (rounds 1-3 quoted 256 here)",
            "ms_per_frame_of_all_sequences": lib_s / (solves // n_seq) * 1e3,
            "at_%d_sequences" % (n_seq // 2): per(half, n_seq // 2), "at_%d_sequences" % (2 * n_seq): per(c, 2 * n_seq),
            "host_side_lists": dict(per(h, n_seq // 2), sequences=n_seq // 2, note="VIO_AMD_RESIDENT=0: FeatureManager lists, window assembly "
                                    "and packing on the host pool (64 threads, its best width), ~125 KB of upload per window (round 3's path)")}


def end_to_end_full(n_seq):
    """The WHOLE pipeline through the C ABI (tools/time_pipeline.py): pageable host frames -> vio_frontend_submit_images_async /
    vio_frontend_collect -> observations -> vio_estimator_process_imu_batch + vio_estimator_process_images -> host states;
    frames rendered from textured planes along synthetic trajectories, so the estimator receives what the tracker
    publishes. `value`: every camera frame published and solved (the headline's convention); `app_cadence_freq3`: the
    app's FREQ = 3, every third camera frame published."""
    run = lambda n, frames, overlap, freq, env=None, registered=False: _tool(
        "import time_pipeline as TP; print(json.dumps(TP.run(%d, %d, %d, quiet=True, freq=%d, registered=%s)))" % (n, frames, overlap, freq, bool(registered)), env)
    every, twice, sync_submit, serial, app = run(n_seq, 44, 2, 1), run(2 * n_seq, 34, 2, 1), run(n_seq, 22, 1, 1), run(n_seq, 22, 0, 1), run(n_seq, 20, 2, 3)   # (the headline legs time ~30 / ~20 frames: one host hiccup of 10 ms in 15 frames moved the mean by 15 %)
    host_lists = run(n_seq, 22, 1, 1, {"VIO_AMD_RESIDENT": "0", "VIO_AMD_HOST_THREADS": "64"})
    # the frame buffers registered once with vio_host_register (a camera ring): DMA from where the frames lie, no gathering pass
    registered = guarded(lambda: run(2 * n_seq, 34, 2, 1, registered=True))   # (a box that cannot page-lock the frames must not cost the legs above)
    return {"value": every["camera_frames_per_s"], "unit": "camera frames/s, every frame published and solved", "sequences": n_seq,
            "path": "pageable frames in -> vio_frontend_submit_images_async (gather to page-locked memory, H2D, kernels and the D2H "
                    "of the observations queued by the context's own host thread) -> vio_frontend_collect -> "
                    "vio_estimator_process_imu_batch -> vio_estimator_process_images (resident landmark stores) -> host states; "
                    "frame k+1 is submitted before the estimator of frame k runs; one host thread drives both contexts; "
                    "measured in a process of its own",
            "every_frame_published": every, "at_%d_sequences" % (2 * n_seq): twice, "synchronous_submit": sync_submit,
            "one_call_after_the_other": serial, "app_cadence_freq3": app,
            "registered_host_frames_at_%d_sequences" % (2 * n_seq): registered,
            "host_side_lists_synchronous_submit": host_lists}


def se3_align_rmse(est, ref):
    """RMSE of positions after the best rigid alignment (Kabsch, no scale): ATE."""
    est, ref = np.asarray(est), np.asarray(ref)
    ce, cr = est.mean(0), ref.mean(0)
    Hm = (est - ce).T @ (ref - cr)
    U, _, Vt = np.linalg.svd(Hm)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    d = (est - ce) @ R.T - (ref - cr)
    return float(np.sqrt((d ** 2).sum(1).mean()))


def closed_loop_ate(cfg, pkg, n_frames=70):
    """The same synthetic sequence (truth known) through the closed loop twice: window solves on the device, and through
    the CPU path (reference build when present, else the restatement); everything else identical."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    import replay_synthetic as RS
    c = pkg.abi.default_config()
    pre = lambda *a: pkg.backend.preintegrate(c, *a)
    solver = pkg.backend.WindowSolver(c, max_batch=1)
    ref = H.ref_lib_or_none()
    cpu_solve = pkg.abi.bind_backend_solver(ref, "ref")[0] if ref is not None else H.oracle_backend()[0]

    def cpu(w):
        ref_w, st = H.solve_with(cpu_solve, c, w)
        w.pose[:], w.speed_bias[:], w.inv_depth[:] = ref_w.pose, ref_w.speed_bias, ref_w.inv_depth
        w.next_prior = ref_w.next_prior
        return st

    res = {}
    devnull = os.open(os.devnull, os.O_WRONLY)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(devnull, 1)
    try:
        loops = {"gpu": RS.ClosedLoop(c, lambda w: solver.solve([w])[0], pre, seed=3, init_noise=1.0),
                 "cpu": RS.ClosedLoop(c, cpu, pre, seed=3, init_noise=1.0)}
        for name, lp in loops.items():
            for _ in range(n_frames):
                lp.step()
            res[name] = (np.array([h[1] for h in lp.history]), np.array([h[2] for h in lp.history]))
            lp.close()
    finally:
        C.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(devnull)
    solver.close()
    g, truth = res["gpu"]
    cpos, _ = res["cpu"]
    return {"unit": "m", "frames": n_frames, "solves": int(len(g)),
            "ate_gpu_vs_truth": se3_align_rmse(g, truth), "ate_cpu_vs_truth": se3_align_rmse(cpos, truth),
            "rmse_gpu_vs_cpu": float(np.sqrt(((g - cpos) ** 2).sum(1).mean())),
            "cpu_path": "vendored Ceres 1.12 + VINS factors" if ref is not None else "C++ restatement",
            "note": "synthetic landmark cloud, 0.5 px observation noise, noisy biased IMU; newest-frame positions of every solve"}


def large_windows(pkg, batches=(64, 256)):
    """configs[2] (W=20, 300 feats, 200 Hz IMU) and configs[4] (W=30, 500 feats, prior + loop constraint): the reduced
    system no longer fits a CU's LDS, vio_window_kernel<false> keeps it in global memory. Solver kernel only."""
    abi, synth, backend = pkg.abi, pkg.synth, pkg.backend
    out = {}
    specs = [("configs[2]", dict(window_size=20, fx=1053.2, fy=1053.4, cx=640.0, cy=360.0), 300, 20, False),
             ("configs[4]", dict(window_size=30, fx=1579.8, fy=1580.0, cx=960.0, cy=540.0), 500, 10, True)]
    for name, kw, nf, ipf, full in specs:
        cfg = abi.default_config(**kw)
        pre = lambda *a, cfg=cfg: backend.preintegrate(cfg, *a)
        if full:
            uniq = steady_state_windows(cfg, pkg, pre, [7, 8], n_features=nf, with_loop=40, imu_per_frame=ipf)
        else:
            uniq = [synth.make_window(cfg, pre, seed=20 + s, n_features=nf, imu_per_frame=ipf) for s in range(2)]
        for batch in batches:
            ws = [uniq[i % len(uniq)].copy() for i in range(batch)]
            solver = backend.WindowSolver(cfg, max_batch=batch)
            solver.upload(ws)
            solver.launch()
            solver.sync()
            solver.kernel_ms()
            for _ in range(3):
                solver.launch()
            solver.sync()
            ms, _ = solver.kernel_ms()
            st = solver.download(ws)
            solver.close()
            iters = float(np.mean([s["iterations"] - 1 for s in st]))
            M = float(np.mean([w.n_factors for w in ws]))
            n_prior = int(np.mean([w.prior.n if w.prior is not None else 0 for w in uniq]))
            flops = algorithmic_flops_per_solve(cfg.window_size, M, n_prior, iters, nf) * batch
            rec = {"window": cfg.window_size, "features": nf, "factors": M, "prior_rows": n_prior, "loop_factors": 40 if full else 0,
                   "batch": batch, "kernel_ms": ms, "ms_per_solve_at_batch": ms / batch, "solves_per_s": batch / (ms * 1e-3),
                   "gn_iterations": iters, "achieved_tflops": flops / (ms * 1e-3) / 1e12,
                   "frac_of_fp64_peak": flops / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
            if batch == batches[0]:
                out[name] = rec
            else:
                out[name]["batch_%d" % batch] = {k: rec[k] for k in ("batch", "kernel_ms", "ms_per_solve_at_batch", "solves_per_s", "frac_of_fp64_peak")}
    return out


CONFIG_LEGS = {
    # BASELINE.json configs[2] / configs[4]: frame geometry, corners (MIN_DIST as tests/test_frontend_gpu.py), window, landmarks,
    # IMU samples per frame (200 Hz IMU at 10 Hz frames: 20), steady-state prior + 40 relocalization factors for [4]
    "configs[2]": dict(rows=720, cols=1280, corners=300, min_dist=30, imu_per_frame=20, full=False,
                       cam=dict(window_size=20, fx=1053.2, fy=1053.4, cx=640.0, cy=360.0)),
    "configs[4]": dict(rows=1080, cols=1920, corners=500, min_dist=30, imu_per_frame=10, full=True,
                       cam=dict(window_size=30, fx=1579.8, fy=1580.0, cx=960.0, cy=540.0)),
}


def measured_lk_iterations(fe, step, n_steps=4):
    """Mean LK iterations per (feature, level) visit, per pyramid level (level 0 first), over n_steps more steps with the
    kernel's counters on (vio_frontend_lk_iterations); untimed."""
    fe.lk_iterations(enable=True, read=True)
    for k in range(n_steps):
        step(k)
    fe.sync()
    it, vis = fe.lk_iterations(enable=False, read=True)
    per_level = [float(i / v) if v > 0 else None for i, v in zip(it, vis)]
    mean = float(it.sum() / vis.sum()) if vis.sum() > 0 else None
    return mean, [x for x in per_level if x is not None]


def config_leg(pkg, name, S=64, steps=10, warmup=2, cpu_frames=50):
    """One BASELINE config end to end like the headline: S resident sequences, every step = front-end step on S frames
    (track + F-RANSAC + detect, every frame published) + the window solve of S windows (incl. marginalization), inputs resident in
    HBM; frames/s, per-kernel ms, both rooflines with the image-size-scaled byte count of SURVEY 8(d), and the CPU path (KLT
    restatement frame + reference Ceres solve, one thread) on a bounded sample of the same workload."""
    import torch
    abi, synth, backend, frontend = pkg.abi, pkg.synth, pkg.backend, pkg.frontend
    sp = CONFIG_LEGS[name]
    rows, cols, nf = sp["rows"], sp["cols"], sp["corners"]
    cfg = abi.default_config(max_corners=nf, min_dist=sp["min_dist"], image_rows=rows, image_cols=cols, **sp["cam"])
    pre = lambda *a: backend.preintegrate(cfg, *a)
    T, n_unique = 4, 2
    uniq_frames = [synth.make_image_stream(300 + u, T, rows=rows, cols=cols)[0] for u in range(n_unique)]
    frames = np.stack([np.stack([uniq_frames[s % n_unique][f] for s in range(S)]) for f in range(T)])
    if sp["full"]:
        uniq_w = steady_state_windows(cfg, pkg, pre, [7, 8], n_features=nf, with_loop=40, imu_per_frame=sp["imu_per_frame"])
    else:
        uniq_w = [synth.make_window(cfg, pre, seed=20 + s, n_features=nf, imu_per_frame=sp["imu_per_frame"]) for s in range(2)]
    ws = [uniq_w[s % len(uniq_w)].copy() for s in range(S)]
    with detect_every_frame(True):  # (DETECT_NOTE)
        fe = frontend.FeatureTracker(cfg, n_seq=S)
    fe.upload_frames(np.ascontiguousarray(frames))
    be = backend.WindowSolver(cfg, max_batch=S)
    be.upload(ws)
    pingpong = list(range(T)) + list(range(T - 2, 0, -1))
    one = torch.cuda.Stream().cuda_stream

    def step(k):
        fe.step(pingpong[k % len(pingpong)], publish=True, stream=one)
        be.launch(stream=one)

    for k in range(warmup):
        step(k)
    torch.cuda.synchronize()
    fe.kernel_ms(), be.kernel_ms()
    t0 = time.perf_counter()
    for k in range(steps):
        step(warmup + k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fe_ms, _ = fe.kernel_ms()
    be_ms, _ = be.kernel_ms()
    stats = be.download(ws)
    lk_mean, lk_levels = measured_lk_iterations(fe, lambda k: fe.step(pingpong[(warmup + steps + k) % len(pingpong)], publish=True, stream=one))
    fe.close(), be.close()
    iters = float(np.mean([s_["iterations"] - 1 for s_ in stats]))
    M = float(np.mean([w.n_factors for w in ws]))
    n_prior = int(np.mean([w.prior.n if w.prior is not None else 0 for w in uniq_w]))
    flops = algorithmic_flops_per_solve(cfg.window_size, M, n_prior, iters, nf) * S
    levels = len(lk_levels) if lk_levels else 4
    fe_bytes = algorithmic_bytes_per_tracked_frame(rows, cols, nf, levels=levels) * S
    fe_bytes_measured = algorithmic_bytes_per_tracked_frame(rows, cols, nf, levels=levels, mean_iters=lk_mean or 10) * S
    out = {"workload": "%s: %dx%d stream, %d feats (MIN_DIST %d), window=%d, %d projection factors, %d IMU samples per frame%s; "
                       "every frame published and solved" % (name, cols, rows, nf, sp["min_dist"], cfg.window_size, M, sp["imu_per_frame"],
                                                             ", %d-row marginalization prior, 40 relocalization factors" % n_prior if sp["full"] else ""),
           "sequences_per_gpu": S, "steps": steps, "value": S * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3,
           "gn_iterations": iters, "kernel_ms": {"frontend_step": fe_ms, "window_solve": be_ms},
           "roofline": {"kernel": "vio_window_kernel<false> (pose matrix in global scratch)", "bound": "mfma",
                        "achieved": flops / (be_ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (be_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "flops_per_solve": flops / S, "traffic": None,
                        "frontend": {"bound": "hbm", "achieved": fe_bytes / (fe_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": fe_bytes / (fe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                 "lk_mean_iterations": lk_mean, "lk_mean_iterations_per_level": lk_levels,
                                 "frac_at_measured_lk_iterations": fe_bytes_measured / (fe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_frame": fe_bytes / S, "algorithmic_bytes_per_frame_at_measured_iterations": fe_bytes_measured / S}}}
    if cpu_frames:
        base = CpuBaseline(cfg, abi, uniq_frames, uniq_w)
        # (BASELINE.md's protocol is >= 200 frames after 20: a reference solve of this window takes 0.1 - 0.4 s, so the leg times 50
        # frames after 5 -- the sample's spread over frames is a few percent, the windows cycle through two unique ones)
        out["cpu_baseline"] = base.one_core(warmup_frames=5, frames=cpu_frames)
        out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"].replace("after 20 warm-up frames", "after 5 warm-up frames (bounded sample: BASELINE.md asks for 200 after 20; one reference solve of this window takes 0.1 - 0.4 s)")
        out["vs_cpu_one_core"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def loop_closure(pkg, n_graphs=256):
    """descriptor matching (searchByDes) and the 4-DoF pose graph solve, many sequences per launch; the
    CPU side is the reference's own functors under the vendored Ceres when oracle/_ref travelled, else the restatement."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    pg, synth = pkg.posegraph, pkg.synth
    out = {}
    # ---- pose graph: 100 keyframes, 10 loop edges (a loop closed after ~100 keyframes of the 10 Hz keyframe stream)
    plib = pg.bind_host(pkg.abi.load_product(), "vio")
    kfs, total, _ = synth.make_loop_keyframes(100, 3, n_loops=10)
    g0, _ = pg.build_with(plib, "vio", kfs, total)
    opt = pg.PoseGraphOptimizer(max_nodes=100, max_edges=len(g0.edge_i), n_graphs=n_graphs)
    try:
        opt.optimize([g0.copy() for _ in range(n_graphs)])
        gs = [g0.copy() for _ in range(n_graphs)]
        t0 = time.perf_counter()
        st = opt.optimize(gs)
        dt = time.perf_counter() - t0
    finally:
        opt.close()
    ref = H.ref_lib_or_none()
    have_ref = ref is not None and hasattr(ref, "ref_posegraph_optimize")
    cfn = pg.bind_checker(ref, "ref") if have_ref else pg.bind_checker(H.oracle_lib(), "oracle")
    c = g0.copy()
    t0 = time.perf_counter()
    pg.optimize_with(cfn, c)
    dc = time.perf_counter() - t0
    out["pose_graph"] = {"keyframes": 100, "edges": int(len(g0.edge_i)), "graphs_per_launch": n_graphs, "iterations": st[0]["iterations"] - 1,
                         "ms_per_call_host_to_host": dt * 1e3, "graphs_per_s": n_graphs / dt,
                         "cpu_ms_per_graph": dc * 1e3, "cpu_kind": "reference (vendored Ceres DENSE_SCHUR + the reference's functors)" if have_ref else "port",
                         "max_abs_dt_vs_cpu_m": float(np.abs(gs[0].t - c.t).max())}
    # ---- searchByDes: 256 keyframe pairs, 150 window descriptors against 500 old descriptors each
    rng = np.random.default_rng(0)
    m = pkg.loop.Matcher()
    try:
        cur = [rng.integers(0, 2 ** 63, (150, 4), dtype=np.int64).astype(np.uint64) for _ in range(n_graphs)]
        old = [rng.integers(0, 2 ** 63, (500, 4), dtype=np.int64).astype(np.uint64) for _ in range(n_graphs)]
        m.search_by_des(cur, old)
        t0 = time.perf_counter()
        m.search_by_des(cur, old)
        dm = time.perf_counter() - t0
    finally:
        m.close()
    # ---- keyframe descriptor extraction: 64 keyframes of 640x480, 150 window points each (FAST + blur + BRIEF)
    pat = np.load(os.path.join(ROOT, "tests", "golden", "brief_pattern.npz"))
    nkf = 64
    frames = np.stack([np.ascontiguousarray(synth.make_texture(np.random.default_rng(100 + s), 480, 640), np.uint8) for s in range(4)])
    frames = np.ascontiguousarray(frames[np.arange(nkf) % 4])
    wpts = [rng.uniform(30, [610, 450], (150, 2)).astype(np.float32) for _ in range(nkf)]
    ex = pkg.loop.BriefExtractor(480, 640, (pat["x1"], pat["y1"], pat["x2"], pat["y2"]), max_frames=nkf, max_keypoints=8192)
    try:
        ex.extract(frames, wpts, allow_cut=True)
        t0 = time.perf_counter()
        res = ex.extract(frames, wpts, allow_cut=True)
        de = time.perf_counter() - t0
    finally:
        ex.close()
    out["brief_extract"] = {"keyframes_per_call": nkf, "image": "640x480", "fast_corners_per_frame": float(np.mean([r[2] for r in res])),
                            "window_points": 150, "ms_per_call_host_to_host": de * 1e3, "keyframes_per_s": nkf / de}
    # ---- bag-of-words query (DBoW2): 64 keyframes x 1000 descriptors -> words + BowVectors, then 64 queries against a
    # database of 4096 keyframes (inverted file on the device); synthetic vocabulary in the app's file layout (k = 10, L = 5: the app's is k = 10, L = 6)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_dbow as TD
    blob, desc = TD.make_vocabulary(10, 5, seed=3, flip=20)
    voc = pkg.loop.BowVocabulary(blob=blob)
    try:
        n_inner = sum(10 ** l for l in range(5))
        leaves = np.arange(n_inner, n_inner + 10 ** 5)
        r2 = np.random.default_rng(5)
        # 16 places (disjoint sets of 4000 words), 4 views of each
        kfs = [TD.keyframe_descriptors(desc, leaves[(i % 16) * 6000:(i % 16) * 6000 + 4000], r2, 890) for i in range(64)]
        voc.transform(kfs)
        t0 = time.perf_counter()
        bows = voc.transform(kfs)
        dtf = time.perf_counter() - t0
        n_db = 4096
        db = pkg.loop.BowDatabase(voc, max_entries=n_db, max_total_words=n_db * 1024)
        try:
            for e in range(n_db - 64):
                db.add(bows[e % 64][2], bows[e % 64][3])
            t0 = time.perf_counter()
            for e in range(n_db - 64, n_db):
                db.add(bows[e % 64][2], bows[e % 64][3])
            dadd = (time.perf_counter() - t0) / 64
            q = [(b[2], b[3]) for b in bows]
            db.query(q, [n_db - 96] * 64, max_results=50)
            t0 = time.perf_counter()
            db.query(q, [n_db - 96] * 64, max_results=50)
            dq = time.perf_counter() - t0
            db.query(q[:1], [n_db - 96], max_results=50)
            t0 = time.perf_counter()
            db.query(q[:1], [n_db - 96], max_results=50)
            dq1 = time.perf_counter() - t0
        finally:
            db.close()
    finally:
        voc.close()
    out["bow_query"] = {"vocabulary": "synthetic k=10 L=5 (111110 nodes), TF_IDF / L1_NORM", "keyframes_per_call": 64,
                        "descriptors_per_keyframe": int(len(kfs[0])), "transform_ms_host_to_host": dtf * 1e3,
                        "descriptors_per_s": 64 * len(kfs[0]) / dtf, "database_entries": n_db, "queries_per_call": 64,
                        "query_ms_host_to_host": dq * 1e3, "query_pairs_per_s": 64 * (n_db - 96) / dq,
                        "one_query_ms_host_to_host": dq1 * 1e3, "add_ms_per_entry_at_full_size": dadd * 1e3,
                        "note": "16 places x 4 views, every view 64 times in the database: the inverted file hands a query the entries "
                                "of its own place (plus what descriptor noise reaches) as candidates, the rest of the database costs nothing"}
    out["search_by_des"] = {"pairs_per_launch": n_graphs, "queries": 150, "candidates": 500, "ms_per_call_host_to_host": dm * 1e3,
                            "descriptor_comparisons_per_s": n_graphs * 150 * 500 / dm}
    return out


if __name__ == "__main__":
    main()
