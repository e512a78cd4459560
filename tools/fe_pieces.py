#!/usr/bin/env python3
"""Front-end step of 512 resident sequences as ONE context against P contexts of 512 / P sequences on their own streams (no join per
step): do the latency-bound kernels (track_update, corner_select) hide behind the other pieces' throughput work?
    tools/fe_pieces.py [total_sequences] [steps]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VIO_AMD_DETECT_ALWAYS", "1")
pkg = importlib.import_module("vins-mobile_amd")
abi, synth, frontend = pkg.abi, pkg.synth, pkg.frontend


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cfg = abi.default_config(max_corners=150, min_dist=20)
    T, n_unique = 4, 4
    uniq = [synth.make_image_stream(100 + u, T)[0] for u in range(n_unique)]
    pingpong = list(range(T)) + list(range(T - 2, 0, -1))
    for P in (1, 2, 4, 8):
        n = S // P
        frames = np.ascontiguousarray(np.stack([np.stack([uniq[s % n_unique][f] for s in range(n)]) for f in range(T)]))
        fes = [frontend.FeatureTracker(cfg, n_seq=n) for _ in range(P)]
        for fe in fes:
            fe.upload_frames(frames)
        for k in range(6):
            for fe in fes:
                fe.step(pingpong[k % len(pingpong)], publish=True)
        for fe in fes:
            fe.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            for fe in fes:
                fe.step(pingpong[(6 + k) % len(pingpong)], publish=True)
        for fe in fes:
            fe.sync()
        dt = (time.perf_counter() - t0) / steps
        print("%d context(s) x %d sequences: %.3f ms per step of %d sequences" % (P, n, dt * 1e3, S))
        for fe in fes:
            fe.close()


if __name__ == "__main__":
    main()
