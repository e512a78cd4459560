#!/usr/bin/env python3
"""Run-to-run / copy-to-copy agreement of the window kernel: copies of the same window inside one launch may only differ by
the order of floating-point atomics (aid for hunting races; tests/test_properties_gpu.py holds the 1e-9 bound)."""
import sys, os, importlib, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vins-mobile_amd")
abi, synth, backend = pkg.abi, pkg.synth, pkg.backend
cfg = abi.default_config(max_corners=150, min_dist=20)
pre = lambda *a: backend.preintegrate(cfg, *a)
uniq = [synth.make_window(cfg, pre, seed=300 + i, n_features=150) for i in range(8)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
s = backend.WindowSolver(cfg, max_batch=B)
worst = 0
for rep in range(reps):
    ws = [uniq[i % 8].copy() for i in range(B)]
    st = s.solve(ws)
    d = [np.abs(ws[i].pose - ws[i % 8].pose).max() for i in range(8, B)]
    bad = [i + 8 for i, x in enumerate(d) if x > 1e-9]
    its = sorted(set((st[i]["iterations"], st[i % 8]["iterations"]) for i in range(8, B)))
    worst = max(worst, max(d))
    if bad or len(its) > 1 and any(a != b for a, b in its):
        i = bad[0] if bad else 8
        print("rep", rep, "max diff", max(d), "n bad", len(bad), "first", i, "iters", st[i]["iterations"], st[i % 8]["iterations"],
              "costs", st[i]["final_cost"], st[i % 8]["final_cost"], "flags", list(st[i]["it_flags"])[:12], list(st[i % 8]["it_flags"])[:12])
print("worst pose diff over", reps, "launches of", B, ":", worst)
