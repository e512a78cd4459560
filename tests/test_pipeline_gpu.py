"""The whole pipeline through the C ABI, the way bench.py's `end_to_end_full` leg drives it (tools/time_pipeline.py): rendered
frames -> vio_frontend_submit_images / collect (or read_images) -> the observations the tracker publishes -> estimator ->
states. The tool asserts that every window from the hand-over on is solved; here the result is also held against the
trajectory, and the overlapped call order against the strictly serial one."""
import os
import sys

import pytest

import helpers as H

sys.path.insert(0, os.path.join(H.ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("freq", [1, 3])
def test_frames_in_states_out(freq):
    import time_pipeline as TP
    n_frames = 15
    serial = TP.run(n_seq=6, n_frames=n_frames, overlap=False, n_worlds=2, quiet=True, freq=freq)
    overlapped = TP.run(n_seq=6, n_frames=n_frames, overlap=True, n_worlds=2, quiet=True, freq=freq)
    assert overlapped["overlap"] and not serial["overlap"]
    for r in (serial, overlapped):
        assert r["mean_published_features"] > 100            # the tracker keeps its 150 corners on the rendered planes
        assert r["position_error_m_max"] < 0.05, r           # newest position against the trajectory (a few solved windows)
        assert r["camera_frames_per_published_frame"] == freq
    # the same frames give the same observations whatever the call order: the estimate is the same to rounding noise of the
    # solver's atomic accumulation order
    assert abs(serial["position_error_m_max"] - overlapped["position_error_m_max"]) < 1e-6
