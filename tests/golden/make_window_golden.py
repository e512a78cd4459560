#!/usr/bin/env python3
"""Records what the REAL reference FeatureManager (oracle/_ref, built from /root/reference) does on the scenarios of
tests/test_window_cpu.py into tests/golden/window_scenarios.npz. Run where /root/reference exists:
    make -C oracle ref && python tests/golden/make_window_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import test_window_cpu as T

lib = H.ref_lib_or_none()
assert lib is not None and hasattr(lib, "ref_fm_create"), "build oracle/_ref first"
out = {}
for si in T.GOLDEN_SCENARIOS:
    seed, n_frames, uninit = T.SCENARIOS[si]
    ref = T.RefFm(lib)
    for k, v in T.flatten(T.run_scenario(ref, seed, n_frames, uninit)).items():
        out["s%d_%s" % (si, k)] = v
    ref.close()
np.savez_compressed(T.GOLDEN, **out)
print("wrote", T.GOLDEN, len(out), "arrays")
