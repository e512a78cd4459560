#!/usr/bin/env python3
"""Reference chain fixture: the closed loop of tools/replay_synthetic.py (seeded synthetic sequence, product host code
for the bookkeeping) with EVERY window solved by the real reference (oracle/_ref: vendored Ceres 1.12 + the verbatim
VINS factor / marginalization sources), priors handed on from solve to solve — MARGIN_OLD and MARGIN_SECOND_NEW as the
parallax test decides. Stores what each solve left behind; tests/test_closed_loop.py replays the same sequence with
the product solver and compares per solve. Bounds the drift of the device's pivot-cut marginalization against the
reference's eigenvalue-cut route over a long chain ("only chains of <= 5 priors are compared").

    python tests/golden/make_chain_golden.py      # needs /root/reference (oracle/_ref built by `make -C oracle ref`)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers as H  # noqa: E402
import replay_synthetic as RS  # noqa: E402

abi, pkg = H.abi, H.pkg
SEED, FRAMES = 11, 52   # 42 consecutive solves


def main():
    ref = H.ref_lib_or_none()
    assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
    rsolve = abi.bind_backend_solver(ref, "ref")[0]
    cfg = abi.default_config()
    pre = lambda *a: pkg.backend.preintegrate(cfg, *a)  # product host code == reference to 1e-12 (tests/test_abi_cpu.py)
    rec = []

    def solve(w):
        out, st = H.solve_with(rsolve, cfg, w)
        w.pose[:], w.speed_bias[:], w.inv_depth[:] = out.pose, out.speed_bias, out.inv_depth
        w.next_prior = out.next_prior
        rec.append(dict(pose=out.pose.copy(), sb=out.speed_bias.copy(), n_feat=w.n_features, n_fact=w.n_factors,
                        flag=w.marginalization_flag, prior_n=w.prior.n if w.prior is not None else 0,
                        next_n=out.next_prior.n, iters=st["iterations"], final_cost=st["final_cost"],
                        initial_cost=st["initial_cost"]))
        return st

    loop = RS.ClosedLoop(cfg, solve, pre, seed=SEED, init_noise=1.0)
    for _ in range(FRAMES):
        loop.step()
    loop.close()
    out = os.path.join(ROOT, "tests", "golden", "chain_ref_closed_loop.npz")
    np.savez_compressed(out, seed=SEED, frames=FRAMES, pose=np.array([r["pose"] for r in rec]), sb=np.array([r["sb"] for r in rec]),
                        n_feat=np.array([r["n_feat"] for r in rec]), n_fact=np.array([r["n_fact"] for r in rec]),
                        flag=np.array([r["flag"] for r in rec]), prior_n=np.array([r["prior_n"] for r in rec]),
                        next_n=np.array([r["next_n"] for r in rec]), iters=np.array([r["iters"] for r in rec]),
                        final_cost=np.array([r["final_cost"] for r in rec]), initial_cost=np.array([r["initial_cost"] for r in rec]))
    flags = [r["flag"] for r in rec]
    print("%d solves, MARGIN_OLD %d / SECOND_NEW %d, prior rows %s..%s, iterations %s" % (
        len(rec), flags.count(0), flags.count(1), min(r["prior_n"] for r in rec[1:]), max(r["prior_n"] for r in rec),
        sorted(set(r["iters"] for r in rec))))


if __name__ == "__main__":
    main()
