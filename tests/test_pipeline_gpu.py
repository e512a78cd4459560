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


def test_asynchronous_submit_gives_the_same_observations_and_reports_its_errors():
    """vio_frontend_submit_images_async: the submit's host work runs on the context's own thread. Same observations as
    read_images frame by frame (bit-exact: same kernels, same inputs), the pending-frame guards hold meanwhile, a second submit
    is refused, and the context can be destroyed with its thread idle."""
    import numpy as np
    from helpers import abi, pkg
    cfg = abi.default_config(max_corners=60, min_dist=25, image_rows=240, image_cols=320)
    streams = [pkg.synth.make_image_stream(30 + q, 6, rows=240, cols=320)[0] for q in range(3)]
    frames = np.stack(streams, axis=1)                                      # [frame][sequence][rows][cols]
    ref = pkg.frontend.FeatureTracker(cfg, n_seq=3)
    got = pkg.frontend.FeatureTracker(cfg, n_seq=3)
    for k in range(frames.shape[0]):
        want = ref.read_images(frames[k], True)
        got.submit(frames[k], True, asynchronous=True)
        with pytest.raises(RuntimeError):
            got.submit(frames[k], True, asynchronous=True)               # one frame in flight per context
        with pytest.raises(RuntimeError):
            got.state(0)                                                  # the tracker state is in flux until collect
        have = got.collect()
        for (ia, xa), (ib, xb) in zip(want, have):
            assert np.array_equal(ia, ib) and np.array_equal(xa, xb)
    with pytest.raises(RuntimeError):
        got.collect()                                                     # nothing pending
    # the overlapped pipeline with the asynchronous submit against the synchronous one
    import time_pipeline as TP
    a = TP.run(n_seq=6, n_frames=15, overlap=1, n_worlds=2, quiet=True, freq=1)
    b = TP.run(n_seq=6, n_frames=15, overlap=2, n_worlds=2, quiet=True, freq=1)
    assert b["overlap"] == 2 and abs(a["position_error_m_max"] - b["position_error_m_max"]) < 1e-6
    ref.close(), got.close()


def test_registered_host_frames_give_the_same_observations():
    """vio_host_register: frames inside a registered buffer go to the device by DMA from where they are (no gathering pass).
    Bit-identical observations to the pageable route, frame by frame, with the synchronous and the asynchronous submit; the
    registry refuses overlaps and unknown pointers; frames OUTSIDE the range still take the gathering route."""
    import ctypes as C
    import numpy as np
    from helpers import abi, pkg
    cfg = abi.default_config(max_corners=60, min_dist=25, image_rows=240, image_cols=320)
    streams = [pkg.synth.make_image_stream(40 + q, 6, rows=240, cols=320)[0] for q in range(3)]
    frames = np.ascontiguousarray(np.stack(streams, axis=1))              # [frame][sequence][rows][cols], ONE buffer
    ref = pkg.frontend.FeatureTracker(cfg, n_seq=3)
    got = pkg.frontend.FeatureTracker(cfg, n_seq=3)
    got.register_host(frames)
    lib = got.lib
    assert lib.vio_host_register(C.c_void_p(frames.ctypes.data + 64), 1024) == abi.VIO_ESTATE     # overlaps
    assert lib.vio_host_unregister(C.c_void_p(frames.ctypes.data + 64)) == abi.VIO_ESTATE         # not a registered base
    outside = frames[0].copy()                                            # a pageable frame next to the registered ring
    for k in range(frames.shape[0]):
        want = ref.read_images(frames[k].copy(), True)                    # (a copy: pageable, gathered)
        if k == 3:
            assert not np.shares_memory(outside, frames)
        got.submit(frames[k], True, asynchronous=bool(k & 1))             # a view into the registered buffer
        have = got.collect()
        for (ia, xa), (ib, xb) in zip(want, have):
            assert np.array_equal(ia, ib) and np.array_equal(xa, xb)
    got.unregister_host(frames)
    assert lib.vio_host_unregister(C.c_void_p(frames.ctypes.data)) == abi.VIO_ESTATE              # already gone
    want = ref.read_images(outside, True)
    have = got.read_images(outside, True)                                 # after unregistering: the gathering route again
    for (ia, xa), (ib, xb) in zip(want, have):
        assert np.array_equal(ia, ib) and np.array_equal(xa, xb)
    ref.close(), got.close()
