#!/usr/bin/env python3
"""Repeats golden-window solves and reports the worst deviation from the fixture (debug aid for atomics-order noise)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import helpers as H
pkg = H.pkg
names = sys.argv[1:] or H.golden_window_names()
for name in names:
    cfg, w, d = H.load_golden_window(name)
    solver = pkg.backend.WindowSolver(cfg, max_batch=1)
    fails, worst = 0, None
    for rep in range(60):
        got = w.copy()
        st = solver.solve([got])[0]
        try:
            H.check_solution(got, st, d, tol=1e-6, tol_prior=1e-5)
        except AssertionError as e:
            fails += 1
            worst = str(e)[:300]
    print(name, "fails", fails, "/60", worst or "")
    solver.close()
