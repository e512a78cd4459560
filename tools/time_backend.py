#!/usr/bin/env python3
"""Quick device timing of the window-solve kernel at config-2 shape (W=10, 150 features, ~800 factors)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vins-mobile_amd")
abi, synth, backend = pkg.abi, pkg.synth, pkg.backend


def main():
    path = "single"  # (the launch-sequence path of rounds 4-5 left the library in round 6; the flag is still accepted)
    prof_batches = [1]
    for a in sys.argv[1:]:
        if a.startswith("--prof-batch="):  # stage clock of window 0 (and the last one) with that many windows in the launch
            prof_batches = [int(x) for x in a.split("=")[1].split(",")]
    args = [a for a in sys.argv[1:] if a != "--no-prior" and not a.startswith("--path=") and not a.startswith("--prof-batch=")]
    batches = [int(x) for x in args] or [1, 64, 256, 512, 1024]
    cfg = abi.default_config()
    pre = lambda *a: backend.preintegrate(cfg, *a)
    if "--no-prior" in sys.argv:
        uniq = [synth.make_window(cfg, pre, seed=42 + i) for i in range(16)]
    else:  # steady-state windows: the prior of the preceding MARGIN_OLD solve rides along (bench.py's workload)
        sys.path.insert(0, ROOT)
        import bench
        uniq = bench.steady_state_windows(cfg, pkg, pre, [42 + i for i in range(8)])
    solver = backend.WindowSolver(cfg, max_batch=max(batches + prof_batches))
    for pb in prof_batches:
        solver.set_profile(True)
        ws = [uniq[i % len(uniq)].copy() for i in range(pb)]
        solver.upload(ws)
        solver.launch()
        solver.sync()
        solver.kernel_ms()
        solver.launch()
        solver.sync()
        pms, _ = solver.kernel_ms()
        print("profiling launch of %d windows: kernel %.3f ms" % (pb, pms))
        for wi in sorted({0, pb - 1}):
            cyc = solver.stage_cycles(wi)
            tot = max(1, cyc["total"])
            print("stage cycles (window %d of %d): " % (wi, pb) + ", ".join("%s=%d(%.1f%%)" % (k, c, 100.0 * c / tot) for k, c in cyc.items()))
        solver.set_profile(False)
    for B in batches:
        ws = [uniq[i % len(uniq)].copy() for i in range(B)]
        solver.upload(ws)
        solver.launch()
        solver.sync()
        solver.kernel_ms()
        t0 = time.time()
        reps = 5
        for _ in range(reps):
            solver.launch()
        solver.sync()
        wall = (time.time() - t0) / reps
        ms, n = solver.kernel_ms()
        st = solver.download(ws)
        print("path=%s " % path + "B=%5d kernel %.3f ms (wall %.3f ms) -> %.0f solves/s, %.1f us/solve/CU-slot; iters %s final cost %.4f" % (
            B, ms, wall * 1e3, B / (ms * 1e-3), ms * 1e3 / max(1, -(-B // 256)), st[0]["iterations"], st[0]["final_cost"]))


if __name__ == "__main__":
    main()
