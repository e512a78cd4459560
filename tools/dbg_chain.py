"""Variance of the 42-solve reference chain on the device: python tools/dbg_chain.py [path] [runs]"""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers as H
from helpers import abi, pkg
import replay_synthetic as RS
g = np.load(os.path.join(H.GOLDEN, "chain_ref_closed_loop.npz"))
cfg = abi.default_config()
path = sys.argv[1] if len(sys.argv) > 1 else "auto"
for run in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
    solver = pkg.backend.WindowSolver(cfg, max_batch=1)
    pre = lambda *a: pkg.backend.preintegrate(cfg, *a)
    rec = []
    def solve(w):
        st = solver.solve([w])[0]
        rec.append((w.pose.copy(), st))
        return st
    loop = RS.ClosedLoop(cfg, solve, pre, seed=int(g["seed"]), init_noise=1.0)
    for _ in range(int(g["frames"])): loop.step()
    loop.close(), solver.close()
    ce = [abs(st["final_cost"] - g["final_cost"][k]) / g["final_cost"][k] for k, (p, st) in enumerate(rec)]
    dp = [np.abs(p[:, :3] - g["pose"][k][:, :3]).max() for k, (p, st) in enumerate(rec)]
    it = all(st["iterations"] == g["iters"][k] for k, (p, st) in enumerate(rec))
    print("path=%s run %d: max rel final-cost error %.2e (window %d), max |dp| %.2e m, iterations equal: %s" % (path, run, max(ce), int(np.argmax(ce)), max(dp), it))
