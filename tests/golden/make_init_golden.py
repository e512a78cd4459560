#!/usr/bin/env python3
"""Records what the REAL reference VisualIMUAlignment (oracle/_ref, built from /root/reference) returns on the cases of
tests/test_initial_cpu.py into tests/golden/init_alignment.npz. Run where /root/reference exists:
    make -C oracle ref && python tests/golden/make_init_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import test_initial_cpu as T

lib = H.ref_lib_or_none()
assert lib is not None and hasattr(lib, "ref_visual_imu_alignment"), "build oracle/_ref first"
lib.ref_visual_imu_alignment.argtypes = None
out = {}
for seed, n, scale in T.CASES:
    frames, truth = T.make_frames(seed, n, scale)
    r = T.run(lib.ref_visual_imu_alignment, frames, truth["tic"])
    for k, v in r.items():
        out["c%d_%s" % (seed, k)] = np.asarray(v)
np.savez_compressed(T.GOLDEN, **out)
print("wrote", T.GOLDEN, len(out), "arrays")
