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


def test_loop_closure_contexts_created_on_one_thread_run_on_another():
    """The pose-graph, descriptor-extraction and matcher contexts carry the same device binding."""
    import os
    import torch
    torch.cuda.set_device(torch.cuda.device_count() - 1)
    pg, loop, synth = pkg.posegraph, pkg.loop, pkg.synth
    d = np.load(os.path.join(H.ROOT, "tests", "golden", "posegraph.npz"))
    g0 = pg.Graph.from_npz_dict(d, "lap80_in_")
    pat = np.load(os.path.join(H.ROOT, "tests", "golden", "brief_pattern.npz"))
    opt = pg.PoseGraphOptimizer(max_nodes=128, max_edges=1024, n_graphs=2)
    ex = loop.BriefExtractor(120, 160, (pat["x1"], pat["y1"], pat["x2"], pat["y2"]), max_frames=1, max_keypoints=4096)
    m = loop.Matcher()
    img = np.ascontiguousarray(synth.make_texture(np.random.default_rng(1), 120, 160), np.uint8)
    pts = np.random.default_rng(2).uniform(20, [140, 100], (50, 2)).astype(np.float32)
    try:
        assert opt.device() == torch.cuda.device_count() - 1
        here = g0.copy()
        opt.optimize([here])
        kp_here, desc_here, _ = ex.extract(img[None], [pts])[0]

        def other_thread():
            g = g0.copy()
            st = opt.optimize([g])[0]
            kp, desc, _ = ex.extract(img[None], [pts])[0]
            idx, dist = m.search_by_des([desc[-50:]], [desc_here[-50:]])[0]
            return g, st, kp, desc, idx, dist

        g, st, kp, desc, idx, dist = _run_in_thread(other_thread)
    finally:
        opt.close(), ex.close(), m.close()
    assert st["iterations"] == int(d["lap80_ref_iterations"])
    assert np.abs(g.t - here.t).max() < 1e-9
    assert np.array_equal(kp, kp_here) and np.array_equal(desc, desc_here)
    assert np.array_equal(idx, np.arange(50)) and np.all(dist == 0)
