#!/usr/bin/env python3
"""Records what the reference's solver stack (oracle/_ref: vendored Ceres 1.12 + Eigen 3.3.0 under the restated
ReprojectionError3D functor, oracle/ref_sfm_harness.cpp) returns for the bundle adjustment and the two-view triangulation
of GlobalSFM::construct on the cases of tests/test_initial_sfm.py into tests/golden/init_sfm.npz. Run where
/root/reference exists:
    make -C oracle ref && python tests/golden/make_init_sfm_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import test_initial_sfm as T

lib = H.ref_lib_or_none()
assert lib is not None and hasattr(lib, "ref_sfm_bundle_adjust"), "build oracle/_ref first"
lib.ref_sfm_triangulate_point.restype = None
tri = T.triangulate_with(lib.ref_sfm_triangulate_point)
out = {}
tp0, tp1, tx0, tx1, tX = [], [], [], [], []
for seed in T.CASES + [T.RUNAWAY]:
    c = T.make_case(seed, tri)
    for k, v in c.items():
        if k != "P":
            out["c%d_in_%s" % (seed, k)] = np.asarray(v)
    r = T.run_ba(lib.ref_sfm_bundle_adjust, c, False)
    assert r["ok"] == 1 and (r["termination"] == 1 or r["iterations"] == 51), r["termination"]   # never the 0.3 s clock
    assert (r["termination"] == 1) == (seed != T.RUNAWAY)
    print("seed", seed, "iterations", r["iterations"], "ok/bad", r["n_ok"], r["n_bad"], "cost %.3e -> %.3e" % (r["initial_cost"], r["final_cost"]))
    for k, v in r.items():
        out["c%d_out_%s" % (seed, k)] = np.asarray(v)
    for j in range(len(c["pts"])):   # the triangulations that produced the case's points
        a, b = c["start"][j], c["start"][j + 1] - 1
        if c["ok"][j]:
            tp0.append(c["P"][c["fr"][a]]), tp1.append(c["P"][c["fr"][b]]), tx0.append(c["xy"][a]), tx1.append(c["xy"][b]), tX.append(c["pts"][j])
out.update(tri_P0=np.array(tp0), tri_P1=np.array(tp1), tri_x0=np.array(tx0), tri_x1=np.array(tx1), tri_X=np.array(tX))
np.savez_compressed(T.GOLDEN, **out)
print("wrote", T.GOLDEN, len(out), "arrays", os.path.getsize(T.GOLDEN), "bytes")
