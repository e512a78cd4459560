#!/usr/bin/env python3
"""Device time of single golden windows (which kernel variant they take and how long one solve runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import helpers as H
pkg = H.pkg
for name in sys.argv[1:] or H.golden_window_names():
    cfg, w, d = H.load_golden_window(name)
    solver = pkg.backend.WindowSolver(cfg, max_batch=1)
    if os.environ.get("PROFILE"):
        solver.set_profile(True)
        solver.upload([w.copy()]); solver.launch(); solver.sync()
        cyc = solver.stage_cycles(0)
        print("  stages:", ", ".join("%s=%d" % (k, c) for k, c in cyc.items() if c > cyc["total"] * 0.02))
        solver.set_profile(False)
    for _ in range(3):
        solver.solve([w.copy()])
    solver.kernel_ms()
    for _ in range(5):
        solver.solve([w.copy()])
    ms, n = solver.kernel_ms()
    print("%-28s W=%2d F=%3d M=%4d loop=%d prior=%d  %.3f ms/solve" % (name, w.W, len(w.inv_depth), len(w.factor_host), int(w.loop_frame >= 0),
          0 if w.prior is None else w.prior.n, ms))
    solver.close()
