"""The headless playback tool (csrc/vio_replay, C++ over the C ABI only) on a synthetic recording written in the app's
record-mode layout: PNG + timestamp per frame at 30 Hz, raw IMU_MSG stream at 100 Hz — pre-step (CLAHE on the device),
KLT front-end publishing every 3rd frame, getMeasurements association, native estimator, window solves, pose log."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

sys.path.insert(0, os.path.join(H.ROOT, "tools"))
pytestmark = pytest.mark.gpu


def test_vio_replay_tracks_a_recorded_sequence(tmp_path):
    import make_recording as MR
    exe = os.path.join(H.ROOT, "vins-mobile_amd", "csrc", "vio_replay")
    assert os.path.exists(exe), "build it with make -C vins-mobile_amd/csrc"
    rec = str(tmp_path / "rec")
    truth = MR.make_recording(rec, n_frames=135, seed=3)
    out = str(tmp_path / "poses")
    r = subprocess.run([exe, rec, out, "--max-corners", "150", "--min-dist", "20"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    err, t = MR.score(out, truth)
    # 135 frames at 30 Hz -> 45 published frames -> window full at the 11th -> 35 solved frames, none lost
    assert len(err) == 35, (len(err), r.stdout[-500:])
    assert np.all(np.diff(t) > 0.09) and np.all(np.diff(t) < 0.11)
    assert np.sqrt((err ** 2).mean()) < 0.08 and err.max() < 0.2, (np.sqrt((err ** 2).mean()), err.max())
    # the same recording without the CLAHE pre-step still tracks (the option exists; the app always equalizes)
    r2 = subprocess.run([exe, rec, out + "2", "--no-clahe", "--max-corners", "150", "--min-dist", "20"], capture_output=True,
                        text=True, timeout=300)
    assert r2.returncode == 0
    err2, _ = MR.score(out + "2", truth)
    assert len(err2) == 35 and err2.max() < 0.3
