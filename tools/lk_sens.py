import importlib, sys, os, numpy as np, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
pkg = importlib.import_module("vins-mobile_amd")
abi, synth, frontend = pkg.abi, pkg.synth, pkg.frontend
S = 256
for iters in (30, 3, 1):
    cfg = abi.default_config(max_corners=150, min_dist=20)
    cfg.lk_max_iters = iters
    rows, cols = cfg.image_rows, cfg.image_cols
    uniq = [synth.make_image_stream(42 + u, 4, rows=rows, cols=cols)[0] for u in range(4)]
    frames = np.stack([np.stack([uniq[s % 4][f] for s in range(S)]) for f in range(4)])
    fe = frontend.FeatureTracker(cfg, n_seq=S)
    fe.upload_frames(frames)
    order = [0, 1, 2, 3, 2, 1]
    for k in range(4): fe.step(order[k % 6], publish=True)
    torch.cuda.synchronize(); fe.kernel_ms()
    t0 = time.perf_counter()
    for k in range(12): fe.step(order[(4 + k) % 6], publish=True)
    torch.cuda.synchronize()
    print("lk_max_iters", iters, "ms/step", (time.perf_counter() - t0) / 12 * 1e3)
    fe.close()
