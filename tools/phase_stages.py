#!/usr/bin/env python3
"""Stage cycles of one window's workgroup inside a batch of B windows (either path): python tools/phase_stages.py phase 512"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vins-mobile_amd")
abi, synth, backend = pkg.abi, pkg.synth, pkg.backend
import bench


def main():
    path, B = sys.argv[1], int(sys.argv[2])
    cfg = abi.default_config()
    pre = lambda *a: backend.preintegrate(cfg, *a)
    uniq = bench.steady_state_windows(cfg, pkg, pre, [42 + i for i in range(8)])
    solver = backend.WindowSolver(cfg, max_batch=B)
    solver.set_path(path)
    solver.set_profile(True)
    ws = [uniq[i % len(uniq)].copy() for i in range(B)]
    solver.upload(ws)
    solver.launch()
    solver.sync()
    solver.launch()
    solver.sync()
    ms, _ = solver.kernel_ms()
    for win in (0, B // 2):
        cyc = solver.stage_cycles(win)
        tot = max(1, cyc["total"])
        print("path=%s B=%d (%.3f ms) window %d, prof_tid=%s: " % (path, B, ms, win, os.environ.get("VIO_AMD_PROF_TID", "0")) +
              ", ".join("%s=%d(%.1f%%)" % (k, c, 100.0 * c / tot) for k, c in cyc.items() if c))


if __name__ == "__main__":
    main()
