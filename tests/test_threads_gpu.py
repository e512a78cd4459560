"""Contexts are bound to the HIP device that was current at *_create and may be driven from any host thread (the
reference calls readImage on the camera-callback thread and solve_ceres on the mainLoop thread,
VINS_ios/ViewController.mm:458 vs :688-724). HIP's current device is per thread, default 0: without the binding a
context created on GPU k > 0 would talk to GPU 0 from every other thread."""
import ctypes as C
import threading

import numpy as np
import pytest

import helpers as H
from helpers import abi, pkg

pytestmark = pytest.mark.gpu


def _hip():
    import torch  # noqa: F401  (its HIP runtime first)
    lib = C.CDLL("libamdhip64.so")
    return lib


def _run_in_thread(fn):
    out = {}

    def body():
        try:
            out["value"] = fn()
        except BaseException as e:  # noqa: BLE001
            out["error"] = e

    t = threading.Thread(target=body)
    t.start()
    t.join()
    if "error" in out:
        raise out["error"]
    return out["value"]


def test_backend_context_created_on_one_thread_solves_on_another():
    import torch
    ndev = torch.cuda.device_count()
    dev = ndev - 1   # the last device: on a multi-GPU box this is NOT the per-thread default
    torch.cuda.set_device(dev)
    cfg, w, d = H.load_golden_window("win_chain_b_prior")
    solver = pkg.backend.WindowSolver(cfg, max_batch=2)
    assert solver.device() == dev

    def other_thread():
        hip = _hip()
        cur = C.c_int(-1)
        hip.hipGetDevice(C.byref(cur))
        got = w.copy()
        stats = solver.solve([got])[0]          # a fresh thread: its current device is 0
        after = C.c_int(-1)
        hip.hipGetDevice(C.byref(after))
        return cur.value, after.value, got, stats

    cur, after, got, stats = _run_in_thread(other_thread)
    assert cur == after                           # the call put the thread's device back
    if ndev > 1:
        assert cur != dev
    H.check_solution(got, stats, d, tol=1e-6)
    # and again from the creating thread, resident form: upload here, launch + download there
    ws = [w.copy(), w.copy()]
    solver.upload(ws)
    st2 = _run_in_thread(lambda: (solver.launch(), solver.download(ws))[1])
    for g, s in zip(ws, st2):
        H.check_solution(g, s, d, tol=1e-6)
    solver.close()
    torch.cuda.set_device(0)


def test_frontend_context_is_thread_agnostic_and_bit_exact():
    import torch
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(ndev - 1)
    cfg = abi.default_config(max_corners=60, min_dist=25, image_rows=240, image_cols=320)
    frames, _ = pkg.synth.make_image_stream(5, 4, rows=240, cols=320)
    trk = pkg.frontend.FeatureTracker(cfg, n_seq=1)
    otrk = H.OracleTracker(cfg)
    dv = C.c_int32(-1)
    assert trk.lib.vio_frontend_get_device(trk._h, C.byref(dv)) == 0 and dv.value == ndev - 1
    for f in range(4):
        # even frames from the creating thread, odd frames from a fresh one (readImage is a camera-callback call)
        call = lambda f=f: trk.read_images(frames[f:f + 1], True)[0]
        gids, gxyz = call() if f % 2 == 0 else _run_in_thread(call)
        rids, rxyz = otrk.read_image(frames[f], True)
        assert np.array_equal(gids, rids) and np.array_equal(gxyz, rxyz), f
    trk.close(), otrk.close()
    torch.cuda.set_device(0)


def test_two_contexts_two_threads_concurrently():
    """Two back-end contexts driven at the same time from two threads (the kernels' dynamic-LDS ceiling is a property of
    the function: it is raised once at create, not per launch)."""
    cfg, w, d = H.load_golden_window("win_c2_easy")
    solvers = [pkg.backend.WindowSolver(cfg, max_batch=8) for _ in range(2)]
    results = [None, None]

    def work(i):
        out = []
        for _ in range(6):
            ws = [w.copy() for _ in range(1 + 3 * i)]
            out.append((ws, solvers[i].solve(ws)))
        results[i] = out

    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for out in results:
        assert out is not None
        for ws, sts in out:
            for g, s in zip(ws, sts):
                H.check_solution(g, s, d, tol=1e-6)
    [s.close() for s in solvers]
