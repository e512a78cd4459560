#!/usr/bin/env python3
"""Stage cycles and kernel time of vio_window_kernel<false> (reduced system in global memory) at configs[2] / configs[4]."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vins-mobile_amd")
import bench  # noqa: E402


def main():
    abi, synth, backend = pkg.abi, pkg.synth, pkg.backend
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    specs = [("configs[2]", dict(window_size=20, fx=1053.2, fy=1053.4, cx=640.0, cy=360.0), 300, 20, False),
             ("configs[4]", dict(window_size=30, fx=1579.8, fy=1580.0, cx=960.0, cy=540.0), 500, 10, True)]
    for name, kw, nf, ipf, full in specs:
        cfg = abi.default_config(**kw)
        pre = lambda *a, cfg=cfg: backend.preintegrate(cfg, *a)
        if full:
            uniq = bench.steady_state_windows(cfg, pkg, pre, [7, 8], n_features=nf, with_loop=40, imu_per_frame=ipf)
        else:
            uniq = [synth.make_window(cfg, pre, seed=20 + s, n_features=nf, imu_per_frame=ipf) for s in range(2)]
        solver = backend.WindowSolver(cfg, max_batch=batch)
        solver.set_profile(True)
        solver.upload([uniq[0].copy()])
        solver.launch()
        solver.sync()
        cyc = solver.stage_cycles(0)
        tot = max(1, cyc["total"])
        print(name + " stage cycles: " + ", ".join("%s=%d(%.1f%%)" % (k, c, 100.0 * c / tot) for k, c in sorted(cyc.items(), key=lambda kv: -kv[1]) if c * 200 > tot))
        solver.set_profile(False)
        for B in (1, batch):
            ws = [uniq[i % len(uniq)].copy() for i in range(B)]
            solver.upload(ws)
            solver.launch()
            solver.sync()
            solver.kernel_ms()
            for _ in range(3):
                solver.launch()
            solver.sync()
            ms, _ = solver.kernel_ms()
            st = solver.download(ws)
            print("%s B=%d kernel %.3f ms; iters %s final cost %.4f" % (name, B, ms, st[0]["iterations"], st[0]["final_cost"]))
        solver.close()


if __name__ == "__main__":
    main()
