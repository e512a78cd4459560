#!/usr/bin/env python3
"""Records what the REAL reference PnP solve (oracle/_ref::ref_pnp_solve: vendored Ceres + IMUFactorPnP +
PerspectiveFactor, built from /root/reference) returns on the cases of tests/test_pnp.py into
tests/golden/pnp_windows.npz. Run where /root/reference exists:
    make -C oracle ref && python tests/golden/make_pnp_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import test_pnp as T

lib = H.ref_lib_or_none()
assert lib is not None and hasattr(lib, "ref_pnp_solve"), "build oracle/_ref first"
cfg = H.abi.default_config()
out = {}
for c in T.CASES:
    w = T.make_window(cfg, *c)
    ref, rs = T.reference(cfg, c[0], w)
    out["c%d_pose" % c[0]], out["c%d_speed" % c[0]] = ref.pose, ref.speed
    for k in ("initial_cost", "final_cost", "iterations", "it_cost", "it_flags"):
        out["c%d_%s" % (c[0], k)] = np.asarray(rs[k])
np.savez_compressed(T.GOLDEN, **out)
print("wrote", T.GOLDEN, len(out), "arrays")
