#!/usr/bin/env python3
"""Writes the synthetic data set tools/estimator_throughput.cpp replays: n_worlds sequences of n_frames frames (IMU
samples, image_msg lists, true states for the initial window)."""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import replay_synthetic as RS  # noqa: E402


def main():
    path, n_worlds, n_frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    cfg = RS.abi.default_config()
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", n_worlds, n_frames))
        for q in range(n_worlds):
            w = RS.SyntheticWorld(cfg, 100 + q)
            f.write(struct.pack("<3d", *w.tic) + struct.pack("<9d", *w.ric.ravel()) + struct.pack("<3d", *w.ba) + struct.pack("<3d", *w.bg))
            for k in range(n_frames):
                imu = [w.imu(w.time(0))] if k == 0 else w.imu_interval(k)
                ids, xyz = w.observe(k)
                P, R, V = w.truth(k)
                f.write(struct.pack("<d", w.time(k)) + struct.pack("<3d", *P) + struct.pack("<9d", *R.ravel()) + struct.pack("<3d", *V))
                f.write(struct.pack("<i", len(imu)) + struct.pack("<%dd" % len(imu), *([w.dt] * len(imu))))
                f.write(np.array([s[0] for s in imu], np.float64).tobytes() + np.array([s[1] for s in imu], np.float64).tobytes())
                f.write(struct.pack("<i", len(ids)))
                for i, p in zip(ids, xyz):
                    f.write(struct.pack("<i3d", int(i), *p))
    print("wrote", path)


if __name__ == "__main__":
    main()
