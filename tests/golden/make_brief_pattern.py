#!/usr/bin/env python3
"""The BRIEF test pattern the app ships with (Resources/brief_pattern.yml, read by BriefExtractor::BriefExtractor,
VINS_ios/loop/keyframe.cpp:375-393) as four int32 arrays: a data file of the reference, needed to make descriptors that
are compatible with its vocabulary.   python tests/golden/make_brief_pattern.py   (needs /root/reference)"""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
txt = open("/root/reference/Resources/brief_pattern.yml").read()
out = {}
for key in ("x1", "y1", "x2", "y2"):
    m = re.search(r"^%s:\n((?:\s+- -?\d+\n)+)" % key, txt, re.M)
    out[key] = np.array([int(v) for v in re.findall(r"-\s+(-?\d+)", m.group(1))], np.int32)
    assert len(out[key]) == 256
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "brief_pattern.npz"), **out)
