"""4-DoF loop pose graph (KeyFrameDatabase::optimize4DoFLoopPoseGraph, VINS_ios/loop/keyfame_database.cpp:140-353).

CPU: the plain restatement (oracle/vio_oracle_posegraph.cpp) against the fixtures the REAL reference left behind
(tests/golden/posegraph.npz: the reference's own cost functors under the vendored Ceres 1.12), and, where oracle/_ref is
present, against the reference directly on fresh graphs; the product's host code (edge list, drift) against the
restatement. GPU: vio_posegraph_optimize against both."""
import os

import numpy as np
import pytest

import helpers as H

pkg = H.pkg
pg, synth = pkg.posegraph, pkg.synth
GOLD = os.path.join(H.ROOT, "tests", "golden", "posegraph.npz")
CASES = ["small", "lap80", "lap200", "resampled", "heavy_drift", "bad_loop", "wrap", "late_start", "rejects_a", "rejects_b",
         "rejects_25it"]
# Tolerances. Translations in metres against graphs of 10-150 m extent, yaw in degrees. The oracle follows the reference
# to ~1e-10 (same algorithm, analytic instead of automatic derivatives); the device differs from both by the order of its
# factorization and atomics.
TOL_ORACLE, TOL_GPU = 1e-8, 1e-6


def load_case(d, name):
    return pg.Graph.from_npz_dict(d, name + "_in_"), int(d[name + "_max_iterations"])


def check_against_golden(d, name, g, st, tol):
    scale = max(1.0, float(np.abs(d[name + "_ref_t"]).max()))
    assert np.abs(g.t - d[name + "_ref_t"]).max() < tol * scale
    dy = (g.ypr[:, 0] - d[name + "_ref_ypr"][:, 0] + 180.0) % 360.0 - 180.0
    assert np.abs(dy).max() < tol * 180.0
    assert np.array_equal(g.ypr[:, 1:], d[name + "_ref_ypr"][:, 1:])  # pitch / roll are not unknowns
    n = int(d[name + "_ref_iterations"])
    assert st["iterations"] == n
    assert st["termination"] == int(d[name + "_ref_termination"])
    assert list(np.asarray(st["it_flags"])[:n]) == list(d[name + "_ref_it_flags"])
    assert np.allclose(np.asarray(st["it_cost"])[:n], d[name + "_ref_it_cost"], rtol=max(tol, 1e-9), atol=0)
    assert np.allclose(np.asarray(st["it_radius"])[:n], d[name + "_ref_it_radius"], rtol=max(tol * 100, 1e-9), atol=0)
    assert abs(st["final_cost"] - float(d[name + "_ref_final_cost"])) <= max(tol, 1e-9) * float(d[name + "_ref_final_cost"])
    assert st["num_successful_steps"] == int(d[name + "_ref_num_successful_steps"])
    assert st["num_unsuccessful_steps"] == int(d[name + "_ref_num_unsuccessful_steps"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_follows_the_reference_fixture(name):
    d = np.load(GOLD)
    g, mi = load_case(d, name)
    fn = pg.bind_checker(H.oracle_lib(), "oracle")
    st = pg.optimize_with(fn, g, mi)
    check_against_golden(d, name, g, st, TOL_ORACLE)


def test_fixture_exercises_rejected_steps_and_resampling():
    d = np.load(GOLD)
    assert any(f == 1 for f in d["rejects_a_ref_it_flags"]) and any(f == 1 for f in d["rejects_b_ref_it_flags"])
    assert int(d["resampled_skip"].sum()) > 50
    assert int(d["rejects_25it_ref_iterations"]) == 26


def test_oracle_against_the_reference_on_fresh_graphs():
    ref = H.ref_lib_or_none()
    if ref is None or not hasattr(ref, "ref_posegraph_optimize"):
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    rfn, ofn = pg.bind_checker(ref, "ref"), pg.bind_checker(H.oracle_lib(), "oracle")
    olib = pg.bind_host(H.oracle_lib(), "oracle")
    for seed in range(50, 56):
        kfs, total, _ = synth.make_loop_keyframes(40 + 17 * (seed - 50), seed, n_loops=5 + seed % 4, yaw_drift_deg=0.2 * (seed - 49))
        g, _ = pg.build_with(olib, "oracle", kfs, total)
        a, b = g.copy(), g.copy()
        sa, sb = pg.optimize_with(rfn, a), pg.optimize_with(ofn, b)
        assert sa["iterations"] == sb["iterations"] and list(sa["it_flags"]) == list(sb["it_flags"])
        assert np.abs(a.t - b.t).max() < 1e-9 and np.abs(a.ypr - b.ypr).max() < 1e-9


def product_host():
    lib = pkg.abi.load_product()  # (no device call below: the two host entry points only)
    return pg.bind_host(lib, "vio")


@pytest.mark.parametrize("mfn,ls", [(500, None), (40, 300)])
def test_host_edge_list_and_drift_match_the_restatement(mfn, ls):
    olib, plib = pg.bind_host(H.oracle_lib(), "oracle"), product_host()
    kfs, total, _ = synth.make_loop_keyframes(120, 9, n_loops=9, first_index=12)
    go, so = pg.build_with(olib, "oracle", kfs, total, max_frame_num=mfn, list_size=ls)
    gp, sp = pg.build_with(plib, "vio", kfs, total, max_frame_num=mfn, list_size=ls)
    assert np.array_equal(so, sp)
    assert np.array_equal(go.edge_i, gp.edge_i) and np.array_equal(go.edge_j, gp.edge_j) and np.array_equal(go.edge_kind, gp.edge_kind)
    assert np.abs(go.edge_meas - gp.edge_meas).max() < 1e-12 and np.abs(go.t - gp.t).max() == 0 and np.abs(go.ypr - gp.ypr).max() < 1e-12
    if ls is not None:
        assert 0 < int(sp.sum()) < len(sp)
    # edges: at most five sequential ones per kept keyframe, all to earlier kept keyframes; one loop edge per has_loop
    seq = gp.edge_kind == 0
    assert np.all(gp.edge_i[seq] < gp.edge_j[seq]) and np.all(sp[gp.edge_i] == 0) and np.all(sp[gp.edge_j] == 0)
    assert np.bincount(gp.edge_j[seq]).max() == 5 and int((~seq).sum()) == sum(k["has_loop"] for k in kfs)
    # the solve (restatement), then the poses / drift from both host codes
    pg.optimize_with(pg.bind_checker(H.oracle_lib(), "oracle"), go)
    ao, ap = pg.apply_with(olib, "oracle", kfs, go, so), pg.apply_with(plib, "vio", kfs, go, so)
    for k in ("t", "r", "r_drift", "t_drift"):
        assert np.abs(ao[k] - ap[k]).max() < 1e-12
    assert abs(ao["yaw_drift"] - ap["yaw_drift"]) < 1e-12
    # kept keyframes carry the optimized pose, rotations stay rotations
    kept = so == 0
    assert np.abs(ap["t"][kept] - go.t[kept]).max() == 0
    assert np.abs(np.einsum("nij,nkj->nik", ap["r"], ap["r"]) - np.eye(3)).max() < 1e-12


def test_entry_points_are_exported_and_validate_arguments():
    lib = pg.bind(pkg.abi.load_product())
    assert lib.vio_posegraph_build(None, 0, 0.0, 1, 0, None, None, None, 0, None, None, None, None, None) == pkg.abi.VIO_EINVAL
    for sym in ("vio_posegraph_create", "vio_posegraph_destroy", "vio_posegraph_optimize", "vio_posegraph_get_device",
                "vio_posegraph_apply"):
        assert hasattr(lib, sym)


# ---- device ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def optimizer():
    o = pg.PoseGraphOptimizer(max_nodes=256, max_edges=2048, n_graphs=16)
    yield o
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_follows_the_reference_fixture(name, optimizer):
    d = np.load(GOLD)
    g, mi = load_case(d, name)
    st = optimizer.optimize([g], mi)[0]
    check_against_golden(d, name, g, st, TOL_GPU)


@pytest.mark.gpu
def test_device_batch_matches_single_and_oracle(optimizer):
    d = np.load(GOLD)
    names = [n for n in CASES if n != "rejects_25it"]
    batch = [load_case(d, n)[0] for n in names] + [load_case(d, "small")[0]]
    sts = optimizer.optimize(batch, 5)
    ofn = pg.bind_checker(H.oracle_lib(), "oracle")
    for n, g, st in zip(names + ["small"], batch, sts):
        o, _ = load_case(d, n)
        so = pg.optimize_with(ofn, o, 5)
        assert st["iterations"] == so["iterations"] and list(st["it_flags"]) == list(so["it_flags"])
        assert np.abs(g.t - o.t).max() < TOL_GPU * max(1.0, np.abs(o.t).max())
        assert np.abs((g.ypr[:, 0] - o.ypr[:, 0] + 180.0) % 360.0 - 180.0).max() < TOL_GPU * 180.0
    # the same graph twice in one launch: the same result up to the order of the atomic accumulation of H
    assert np.abs(batch[0].t - batch[-1].t).max() < 1e-9


@pytest.mark.gpu
def test_device_closes_the_loop_end_to_end(optimizer):
    """build (host) -> optimize (device) -> apply (host): the drift accumulated around the lap is taken out."""
    plib = product_host()
    kfs, total, truth = synth.make_loop_keyframes(150, 21, n_loops=14, yaw_drift_deg=0.4, pos_drift=0.05)
    g, skip = pg.build_with(plib, "vio", kfs, total, max_frame_num=500)
    before = np.linalg.norm(g.t - truth, axis=1).max()
    optimizer.optimize([g], 5)
    out = pg.apply_with(plib, "vio", kfs, g, skip)
    after = np.linalg.norm(out["t"] - truth, axis=1).max()
    assert after < 0.35 * before
    assert optimizer.device() >= 0


@pytest.mark.gpu
def test_device_large_graph_matches_the_restatement():
    """500 keyframes (2000 unknowns, 125 tile columns, 40 long loop rows) and 700 keyframes (the LDS limit is ~730)."""
    plib = product_host()
    ofn = pg.bind_checker(H.oracle_lib(), "oracle")
    for n, loops in ((500, 40), (700, 25)):
        kfs, total, truth = synth.make_loop_keyframes(n, 5, n_loops=loops, radius=30.0)
        g, skip = pg.build_with(plib, "vio", kfs, total, max_frame_num=1000)
        opt = pg.PoseGraphOptimizer(max_nodes=n, max_edges=len(g.edge_i), n_graphs=2)
        try:
            a, b, o = g.copy(), g.copy(), g.copy()
            sa, sb = opt.optimize([a, b])
            so = pg.optimize_with(ofn, o)
        finally:
            opt.close()
        assert sa["iterations"] == so["iterations"] and list(sa["it_flags"]) == list(so["it_flags"])
        assert np.allclose(np.asarray(sa["it_cost"])[:so["iterations"]], np.asarray(so["it_cost"])[:so["iterations"]], rtol=1e-8)
        assert np.abs(a.t - o.t).max() < TOL_GPU * np.abs(o.t).max()
        assert np.abs((a.ypr[:, 0] - o.ypr[:, 0] + 180.0) % 360.0 - 180.0).max() < TOL_GPU * 180.0
        assert np.abs(a.t - b.t).max() < 1e-8
        assert np.linalg.norm(a.t - truth, axis=1).max() < 0.9 * np.linalg.norm(g.t - truth, axis=1).max()
    with pytest.raises(RuntimeError):
        pg.PoseGraphOptimizer(max_nodes=900, max_edges=100, n_graphs=1)  # seven N-vectors no longer fit the LDS: VIO_ECAP


@pytest.mark.gpu
def test_device_properties_gauge_invariance_and_fixed_point(optimizer):
    """Size-independent properties (no oracle involved). (1) The residuals only see relative poses in the frame of the
    earlier keyframe: moving the whole graph by a yaw and a translation about the vertical axis moves the optimum with it
    (pitch / roll are per-edge constants, untouched by a yaw). (2) A consistent graph — every edge equal to what its two
    keyframes say — is a fixed point: zero cost, no step taken."""
    plib = product_host()
    kfs, total, _ = synth.make_loop_keyframes(140, 31, n_loops=12, yaw_drift_deg=0.5)
    g, _ = pg.build_with(plib, "vio", kfs, total)
    yaw0, shift = 73.0, np.array([5.0, -3.0, 1.5])
    c, s_ = np.cos(np.deg2rad(yaw0)), np.sin(np.deg2rad(yaw0))
    Rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
    moved = g.copy()
    moved.t = g.t @ Rz.T + shift
    moved.ypr = g.ypr.copy()
    moved.ypr[:, 0] = (g.ypr[:, 0] + yaw0 + 180.0) % 360.0 - 180.0
    a, b = g.copy(), moved.copy()
    sa, sb = optimizer.optimize([a, b])
    assert sa["iterations"] == sb["iterations"] and list(sa["it_flags"]) == list(sb["it_flags"])
    n = sa["iterations"]
    assert np.allclose(np.asarray(sa["it_cost"])[:n], np.asarray(sb["it_cost"])[:n], rtol=1e-7)
    assert np.abs(a.t @ Rz.T + shift - b.t).max() < 1e-6 * np.abs(b.t).max()
    assert np.abs((a.ypr[:, 0] + yaw0 - b.ypr[:, 0] + 180.0) % 360.0 - 180.0).max() < 1e-6
    # (2) make every measurement consistent with the optimized poses of `a`, then optimize again from there
    fixed = a.copy()
    R = [synth._ypr_to_R(y) for y in fixed.ypr]
    for e in range(len(fixed.edge_i)):
        i, j = fixed.edge_i[e], fixed.edge_j[e]
        Ri = synth._ypr_to_R([fixed.ypr[i, 0], fixed.edge_meas[e, 4], fixed.edge_meas[e, 5]])
        fixed.edge_meas[e, 0:3] = Ri.T @ (fixed.t[j] - fixed.t[i])
        fixed.edge_meas[e, 3] = (fixed.ypr[j, 0] - fixed.ypr[i, 0] + 180.0) % 360.0 - 180.0
    before = fixed.copy()
    st = optimizer.optimize([fixed])[0]
    assert st["initial_cost"] < 1e-18 and st["termination"] == 1 and st["iterations"] == 1
    assert np.array_equal(fixed.t, before.t) and np.array_equal(fixed.ypr, before.ypr)
    del R


@pytest.mark.gpu
def test_device_capacity_and_argument_errors(optimizer):
    d = np.load(GOLD)
    g, _ = load_case(d, "lap200")
    small = pg.PoseGraphOptimizer(max_nodes=64, max_edges=256, n_graphs=1)
    try:
        with pytest.raises(RuntimeError):
            small.optimize([g])          # 200 keyframes into a 64-keyframe context: VIO_ECAP
        with pytest.raises(RuntimeError):
            small.optimize([load_case(d, "small")[0]] * 2)  # two graphs into a one-graph context
    finally:
        small.close()
    bad = load_case(d, "small")[0]
    bad.edge_j[0] = 999
    with pytest.raises(RuntimeError):
        optimizer.optimize([bad])
