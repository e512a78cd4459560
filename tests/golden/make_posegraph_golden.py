#!/usr/bin/env python3
"""Pose-graph fixtures: seeded synthetic keyframe lists (vins-mobile_amd/synth.py make_loop_keyframes), the edge list
the product's host code builds from them, and what the REAL reference leaves behind for that graph — the reference's
own cost functors (VINS_ios/loop/keyfame_database.h) under the vendored Ceres 1.12, oracle/ref_posegraph_harness.cpp.
tests/test_posegraph.py compares the CPU restatement and the HIP path with these.

    python tests/golden/make_posegraph_golden.py      # needs /root/reference (oracle/_ref built by `make -C oracle ref`)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402

pkg = H.pkg
pg, synth = pkg.posegraph, pkg.synth

# name: (n keyframes, seed, loops, keyword overrides, max_frame_num, list_size, mutation)
CASES = {
    "small": (30, 1, 4, {}, 500, None, None),
    "lap80": (80, 2, 10, {}, 500, None, None),
    "lap200": (200, 3, 30, {}, 500, None, None),
    "resampled": (160, 4, 12, {}, 60, 400, None),            # list longer than max_frame_num: need_resample flags
    "heavy_drift": (120, 5, 16, {"yaw_drift_deg": 1.5, "pos_drift": 0.15}, 500, None, None),
    "bad_loop": (90, 6, 8, {}, 500, None, "bad_loop"),        # one loop measurement 3 m / 40 deg off
    "wrap": (70, 7, 8, {"laps": 1.6}, 500, None, None),       # yaw crosses +-180 several times
    "late_start": (64, 8, 6, {"first_index": 37}, 500, None, None),
    # wrong loops (last nb loop edges off by dt metres / dyaw degrees): rejected steps, radius / 2, / 4, ... in the trace
    "rejects_a": (100, 33, 20, {"yaw_drift_deg": 2.0}, 500, None, ("bad_tail", 6, 50.0, 90.0)),
    "rejects_b": (100, 32, 20, {"yaw_drift_deg": 2.0}, 500, None, ("bad_tail", 2, 200.0, 120.0)),
    "rejects_25it": (100, 35, 20, {"yaw_drift_deg": 2.0}, 500, None, ("bad_tail", 6, 50.0, 90.0)),
}
MAX_ITERATIONS = {"rejects_25it": 25}   # everything else: the reference's 5 (keyfame_database.cpp:159)


def make_case(name, lib, prefix):
    n, seed, loops, kw, mfn, ls, mut = CASES[name]
    kfs, total, truth = synth.make_loop_keyframes(n, seed, n_loops=loops, **kw)
    if mut == "bad_loop":
        k = n - 3
        kfs[k]["loop_info"][0] += 3.0
        kfs[k]["loop_info"][7] += 40.0
    elif mut is not None:
        _, nb, dt, dyaw = mut
        for k in range(n - nb, n):
            kfs[k]["loop_info"][0] += dt
            kfs[k]["loop_info"][7] += dyaw
    g, skip = pg.build_with(lib, prefix, kfs, total, max_frame_num=mfn, list_size=ls)
    return kfs, total, truth, g, skip


def main():
    ref = H.ref_lib_or_none()
    assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
    rfn = pg.bind_checker(ref, "ref")
    olib = pg.bind_host(H.oracle_lib(), "oracle")
    out = {}
    for name in CASES:
        kfs, total, truth, g, skip = make_case(name, olib, "oracle")
        r = g.copy()
        st = pg.optimize_with(rfn, r, MAX_ITERATIONS.get(name, 5))
        out[name + "_max_iterations"] = np.int32(MAX_ITERATIONS.get(name, 5))
        out.update(g.to_npz_dict(name + "_in_"))
        out[name + "_skip"] = skip
        out[name + "_ref_t"], out[name + "_ref_ypr"] = r.t, r.ypr
        for k in ("iterations", "termination", "initial_cost", "final_cost", "num_successful_steps", "num_unsuccessful_steps"):
            out[name + "_ref_" + k] = st[k]
        for k in ("it_cost", "it_radius", "it_step_norm", "it_relative_decrease", "it_flags"):
            out[name + "_ref_" + k] = np.asarray(st[k])[:st["iterations"]]
        print("%-12s nodes %3d (skipped %3d) edges %4d  iterations %d flags %s cost %.4f -> %.4f, moved %.3f m" % (
            name, len(g.t), int(skip.sum()), len(g.edge_i), st["iterations"], list(np.asarray(st["it_flags"])[:st["iterations"]]),
            st["initial_cost"], st["final_cost"], np.abs(r.t - g.t).max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "posegraph.npz"), **out)


if __name__ == "__main__":
    main()
