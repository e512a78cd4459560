#!/usr/bin/env python3
"""Writes a synthetic recording in the reference app's record-mode layout (VINS_ios/ViewController.mm:1614-1708):

    <dir>/IMU             IMU_MSG stream with the header == 0 end marker, 100 Hz
    <dir>/IMAGE/<i>       RGBA PNG frames of a textured plane rendered along a known trajectory, 30 Hz
    <dir>/IMAGE_TIME/<i>  8-byte timestamps
    <dir>/INIT            (this repo's addition) true window states per frame header, in place of solveInitial
    <dir>/TRUTH.npz       ground truth for scoring

for `vins-mobile_amd/csrc/vio_replay <dir> <poses>`; see tests/test_replay_cli_gpu.py."""
import argparse
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import replay_synthetic as RS  # noqa: E402

pkg = RS.pkg
abi, replay = pkg.abi, pkg.replay


def make_recording(path, n_frames=120, seed=3, fps=30.0, imu_hz=100.0, shade=True):
    cfg = abi.default_config()
    world = RS.ImageWorld(cfg, seed, frame_dt=1.0 / fps, imu_per_frame=1)
    os.makedirs(os.path.join(path, "IMAGE"), exist_ok=True)
    os.makedirs(os.path.join(path, "IMAGE_TIME"), exist_ok=True)
    t_first, t_last = world.time(0), world.time(n_frames - 1)
    t_imu = np.arange(t_first - 0.0537, t_last + 0.06, 1.0 / imu_hz)      # off the frame grid: no header ties
    samples = [world.imu(t) for t in t_imu]
    replay.write_imu(os.path.join(path, "IMU"), t_imu, [s[0] for s in samples], [s[1] for s in samples])
    rows, cols = cfg.image_rows, cfg.image_cols
    yy, xx = np.mgrid[0:rows, 0:cols]
    light = 0.55 + 0.45 * (xx / cols) if shade else np.ones((rows, cols))   # uneven lighting: what the CLAHE pre-step is for
    truth_t, truth_P, truth_q = [], [], []
    with open(os.path.join(path, "INIT"), "wb") as f:
        for k in range(n_frames):
            g = np.clip(world.render(k) * light, 0, 255).astype(np.uint8)
            rgba = np.stack([g, g, g, np.full_like(g, 255)], axis=-1)
            replay.write_image(os.path.join(path, "IMAGE"), k, rgba)
            replay.write_image_time(os.path.join(path, "IMAGE_TIME"), k, world.time(k))
            P, R, V = world.truth(k)
            f.write(struct.pack("<22d", world.time(k), *P, *R.ravel(), *V, *world.ba, *world.bg))
            truth_t.append(world.time(k)), truth_P.append(P), truth_q.append(RS.synth.rot_to_quat(R))
    np.savez(os.path.join(path, "TRUTH.npz"), t=np.array(truth_t), P=np.array(truth_P), q=np.array(truth_q))
    return dict(t=np.array(truth_t), P=np.array(truth_P), q=np.array(truth_q), tic=world.tic, cfg=cfg)


def score(pose_log, truth):
    """Position error of the pose log against the truth at the same headers, gauge removed at the first record."""
    t, P, q = replay.read_keyframes(pose_log)
    idx = [int(np.argmin(np.abs(truth["t"] - h))) for h in t]
    assert all(abs(truth["t"][i] - h) < 1e-9 for i, h in zip(idx, t))
    d = P - truth["P"][idx]
    d = d - d[0]
    return np.sqrt((d ** 2).sum(1)), t


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--seed", type=int, default=3)
    a = ap.parse_args()
    make_recording(a.dir, a.frames, a.seed)
    print("wrote", a.dir)
