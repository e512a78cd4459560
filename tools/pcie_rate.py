#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: frames and windows handed over as HOST buffers on every step
(vio_frontend_read_images + vio_backend_solve_windows), for the note in DESIGN.md section 5 (never bench.py's value)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (its HIP runtime must load first)
pkg = importlib.import_module("vins-mobile_amd")
abi, synth, backend, frontend = pkg.abi, pkg.synth, pkg.backend, pkg.frontend

S = 256
cfg = abi.default_config(max_corners=150, min_dist=20)
rows, cols = cfg.image_rows, cfg.image_cols
uniq = [synth.make_image_stream(42 + u, 4, rows=rows, cols=cols)[0] for u in range(4)]
frames = np.stack([np.stack([uniq[s % 4][f] for s in range(S)]) for f in range(4)])
pre = lambda *a: backend.preintegrate(cfg, *a)
uw = [synth.make_window(cfg, pre, seed=42 + u, n_features=150) for u in range(8)]
fe = frontend.FeatureTracker(cfg, n_seq=S)
be = backend.WindowSolver(cfg, max_batch=S)
order = [0, 1, 2, 3, 2, 1]
for k in range(3):
    fe.read_images(frames[order[k % 6]], True)
    be.solve([uw[s % 8].copy() for s in range(S)])
n = 8
t0 = time.perf_counter()
for k in range(n):
    fe.read_images(frames[order[(3 + k) % 6]], True)
t_fe = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for k in range(n):
    ws = [uw[s % 8].copy() for s in range(S)]
    t1 = time.perf_counter()
    be.solve(ws)
    t_be_inner = time.perf_counter() - t1
t_be = t_be_inner
print("host-buffer path, %d sequences: read_images %.2f ms/step (upload %d MB + observations back), solve_windows %.2f ms/step "
      "(pack + upload + solve + download) -> %.0f frames/s" % (S, t_fe * 1e3, S * rows * cols // 2**20, t_be * 1e3, S / (t_fe + t_be)))
