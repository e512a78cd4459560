#!/usr/bin/env python3
"""Closed-loop replay of a synthetic visual-inertial sequence through the library's pieces in the order the reference
calls them (VINS::processIMU / processImage / solve_ceres / slideWindow, VINS_ios/VINS.cpp:333-478, 480-831, 1149-1273):

    IMU samples -> vio_preintegrate + state propagation          (processIMU)
    image_msg   -> vio_features_add_check_parallax               (keyframe decision -> marginalization flag)
                   vio_features_triangulate / get_depth_vector / export_factors
                   window solve (+ new2old + marginalization)     vio_backend_solve_windows  [or an oracle, for CPU runs]
                   vio_features_set_depth / remove_failures
                   slide: remove_back_shift_depth | remove_front, state / pre-integration shifts

This file is TEST / EXAMPLE glue (numpy bookkeeping of the window arrays only); every computation of the hot path goes
through the C ABI. The observations are synthetic projections of a landmark cloud (the KLT front-end has its own
parity tests); the estimator starts from known states for the first window (the reference's SfM / visual-inertial
alignment initialisation is a later row).

    python tools/replay_synthetic.py --frames 200            # on an MI355X: product solver, prints the trajectory error
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vins-mobile_amd")
abi, synth, window = pkg.abi, pkg.synth, pkg.window


class SyntheticWorld:
    """Ground-truth trajectory, IMU samples and landmark observations."""

    def __init__(self, cfg, seed, n_landmarks=4000, frame_dt=0.1, imu_per_frame=10, pix_noise=0.5):
        self.cfg, self.rng = cfg, np.random.default_rng(seed)
        rng = self.rng
        self.traj = synth.Trajectory(rng)
        self.t0 = rng.uniform(0, 20)
        self.frame_dt, self.imu_per_frame, self.dt = frame_dt, imu_per_frame, frame_dt / imu_per_frame
        self.ex = synth.ex_pose_default()
        self.ric, self.tic = synth.quat_to_rot(self.ex[3:]), self.ex[:3]
        self.ba, self.bg = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
        self.g = np.array([0, 0, synth.GRAVITY])
        self.pix = pix_noise / cfg.fx
        # the camera looks along -z of the body (ric = Rx(180 deg)): a slab of landmarks below the path
        self.lm = np.column_stack([rng.uniform(-9, 9, n_landmarks), rng.uniform(-9, 9, n_landmarks), rng.uniform(-11, -4, n_landmarks)])
        self.half_x, self.half_y = 0.8 * cfg.cx / cfg.fx, 0.8 * cfg.cy / cfg.fy
        self.tracked = {}   # feature id -> landmark index
        self.next_id = 0

    def time(self, k):
        return self.t0 + k * self.frame_dt

    def truth(self, k):
        t = self.time(k)
        return self.traj.pos(t), self.traj.rot(t), self.traj.vel(t)

    def imu(self, t):
        R = self.traj.rot(t)
        a = R.T @ (self.traj.acc(t) + self.g) + self.ba + self.rng.normal(0, 0.02, 3)
        w = self.traj.omega_body(t) + self.bg + self.rng.normal(0, 0.002, 3)
        return a, w

    def imu_interval(self, k):
        """Samples strictly after frame k-1 up to frame k (the reference pushes one sample per IMU message)."""
        t = self.time(k - 1)
        return [self.imu(t + (s + 1) * self.dt) for s in range(self.imu_per_frame)]

    def observe(self, k, max_features=150):
        P, R, _ = self.truth(k)
        Rc, Pc = R @ self.ric, P + R @ self.tic
        pc = (self.lm - Pc) @ Rc
        z = pc[:, 2]
        vis = (z > 0.5) & (np.abs(pc[:, 0]) < self.half_x * z) & (np.abs(pc[:, 1]) < self.half_y * z)
        ids, xyz = [], []
        for fid, li in list(self.tracked.items()):
            if vis[li] and self.rng.random() < 0.97:
                ids.append(fid)
                xyz.append([pc[li, 0] / z[li] + self.rng.normal(0, self.pix), pc[li, 1] / z[li] + self.rng.normal(0, self.pix), 1.0])
            else:
                del self.tracked[fid]
        busy = set(self.tracked.values())
        for li in self.rng.permutation(np.flatnonzero(vis)):
            if len(ids) >= max_features:
                break
            if li in busy:
                continue
            self.tracked[self.next_id] = int(li)
            ids.append(self.next_id)
            xyz.append([pc[li, 0] / z[li] + self.rng.normal(0, self.pix), pc[li, 1] / z[li] + self.rng.normal(0, self.pix), 1.0])
            self.next_id += 1
        return ids, xyz


class ImageWorld(SyntheticWorld):
    """The same trajectory and IMU, but the camera films a textured plane: frames are rendered by intersecting every
    pixel's ray with the plane z = z0, and the observations come from the KLT front-end run on those frames."""

    def __init__(self, cfg, seed, z0=-6.0, px_per_m=77.0, **kw):
        super().__init__(cfg, seed, n_landmarks=1, **kw)
        self.z0, self.s = z0, px_per_m
        self.tex_half = 9.0  # metres covered by the texture around the origin
        n = int(2 * self.tex_half * px_per_m)
        self.tex = synth.make_texture(self.rng, n, n)
        rows, cols = cfg.image_rows, cfg.image_cols
        v, u = np.mgrid[0:rows, 0:cols].astype(np.float64)
        self.rays = np.stack([(u - cfg.cx) / cfg.fx, (v - cfg.cy) / cfg.fy, np.ones_like(u)], axis=-1)

    def render(self, k):
        P, R, _ = self.truth(k)
        Rc, Pc = R @ self.ric, P + R @ self.tic
        d = self.rays @ Rc.T
        lam = (self.z0 - Pc[2]) / d[..., 2]
        X, Y = Pc[0] + lam * d[..., 0], Pc[1] + lam * d[..., 1]
        img = synth._bilinear(self.tex, (X + self.tex_half) * self.s, (Y + self.tex_half) * self.s)
        img = img + self.rng.normal(0, 1.0, img.shape)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)


class ClosedLoop:
    """The estimator loop. solve(window) -> stats solves one abi.Window in place (product WindowSolver or an oracle).
    With `tracker` (a frontend.FeatureTracker for one sequence) the observations are what the KLT front-end publishes
    on frames rendered by an ImageWorld; without, they are noisy projections of a landmark cloud."""

    def __init__(self, cfg, solve, preintegrate, seed=1, init_noise=0.0, tracker=None):
        self.cfg, self.solve, self.pre = cfg, solve, preintegrate
        self.W = cfg.window_size
        self.tracker = tracker
        self.world = ImageWorld(cfg, seed) if tracker is not None else SyntheticWorld(cfg, seed)
        self.fm = window.FeatureManager(self.W)
        self.Ps, self.Rs, self.Vs, self.Bas, self.Bgs = [], [], [], [], []
        self.pre_arr, self.pre_samples = [], []   # per interval (i-1, i): packed pre-integration and its raw samples
        self.prior = None
        self.frame_count = 0
        self.k = 0
        self.last_imu = None
        self.init_noise = init_noise
        self.history = []   # (frame k, estimated position of the newest frame, true position, stats)
        self.rng = np.random.default_rng(seed + 99)

    # ---- processIMU: pre-integration of the interval and propagation of the newest state (VINS.cpp:333-375) ----
    def _integrate(self, acc0, gyr0, ba, bg, samples):
        dts = np.full(len(samples), self.world.dt)
        accs = np.array([s[0] for s in samples])
        gyrs = np.array([s[1] for s in samples])
        return self.pre(acc0, gyr0, ba, bg, dts, accs, gyrs)

    def _propagate(self, P, R, V, ba, bg, acc0, gyr0, samples):
        g, dt = self.world.g, self.world.dt
        for a1, w1 in samples:
            un_acc_0 = R @ (acc0 - ba) - g
            un_gyr = 0.5 * (gyr0 + w1) - bg
            R = R @ synth.quat_to_rot(np.array([*(un_gyr * dt / 2), 1.0]), normalize=False)  # Utility::deltaQ, unnormalized
            un_acc_1 = R @ (a1 - ba) - g
            un_acc = 0.5 * (un_acc_0 + un_acc_1)
            P = P + dt * V + 0.5 * dt * dt * un_acc
            V = V + dt * un_acc
            acc0, gyr0 = a1, w1
        return P, R, V

    def step(self):
        world, W, k = self.world, self.W, self.k
        if k == 0:
            self.last_imu = world.imu(world.time(0))
        else:
            samples = world.imu_interval(k)
            ba, bg = self.Bas[-1].copy(), self.Bgs[-1].copy()
            self.pre_arr.append(self._integrate(self.last_imu[0], self.last_imu[1], ba, bg, samples))
            self.pre_samples.append((self.last_imu, samples))
            P, R, V = self._propagate(self.Ps[-1], self.Rs[-1], self.Vs[-1], ba, bg, self.last_imu[0], self.last_imu[1], samples)
            self.last_imu = samples[-1]
            self.Bas.append(ba), self.Bgs.append(bg)
        if k <= W:  # the first window starts from (nearly) known states: initialisation is a later row
            Pt, Rt, Vt = world.truth(k)
            n = self.init_noise
            P = Pt + self.rng.normal(0, 0.01 * n, 3)
            R = Rt @ synth.rotvec_to_rot(self.rng.normal(0, 0.005 * n, 3))
            V = Vt + self.rng.normal(0, 0.02 * n, 3)
            if k == 0:
                self.Bas.append(world.ba + self.rng.normal(0, 0.01 * n, 3)), self.Bgs.append(world.bg + self.rng.normal(0, 0.001 * n, 3))
        self.Ps.append(P), self.Rs.append(R), self.Vs.append(V)
        if self.tracker is not None:
            ids, xyz = self.tracker.read_images(world.render(k)[None], True)[0]  # readImage, every frame published
        else:
            ids, xyz = world.observe(k)
        enough, _, ltn = self.fm.add_check_parallax(self.frame_count, ids, xyz)
        self.k += 1
        if self.frame_count < W:
            self.frame_count += 1
            return None
        stats = self._solve_and_slide(enough)
        self.history.append((k, self.Ps[-1].copy(), world.truth(k)[0], stats, ltn))
        return stats

    # ---- processImage after initialisation: solve_ceres + slideWindow (VINS.cpp:1119-1143, 1149-1273) ----
    def _solve_and_slide(self, enough):
        W, fm, world = self.W, self.fm, self.world
        Rrows = np.array([R.ravel() for R in self.Rs])
        fm.triangulate(np.array(self.Ps), Rrows, world.tic, world.ric)
        pose = np.array([np.concatenate([p, synth.rot_to_quat(R)]) for p, R in zip(self.Ps, self.Rs)])
        sb = np.array([np.concatenate([v, a, g]) for v, a, g in zip(self.Vs, self.Bas, self.Bgs)])
        inv_depth = fm.get_depth_vector()
        host, target, feat, pi, pj, nf = fm.export_factors()
        assert nf == len(inv_depth)
        flag = abi.VIO_MARGIN_OLD if enough else abi.VIO_MARGIN_SECOND_NEW
        w = abi.Window(W, pose, sb, world.ex, inv_depth, host, target, feat, pi, pj, np.array(self.pre_arr), prior=self.prior,
                       marginalization_flag=flag)
        stats = self.solve(w)
        for i in range(W + 1):
            self.Ps[i], self.Rs[i] = w.pose[i, :3].copy(), synth.quat_to_rot(w.pose[i, 3:])
            self.Vs[i], self.Bas[i], self.Bgs[i] = w.speed_bias[i, :3].copy(), w.speed_bias[i, 3:6].copy(), w.speed_bias[i, 6:].copy()
        fm.set_depth(w.inv_depth)
        if w.next_prior.n > 0:   # n == -1: MARGIN_SECOND_NEW left the old prior untouched (VINS.cpp:778-779)
            self.prior = w.next_prior.copy()
        if enough:               # slideWindowOld (VINS.cpp:1253-1273)
            R0, P0 = self.Rs[0] @ world.ric, self.Ps[0] + self.Rs[0] @ world.tic
            for lst in (self.Ps, self.Rs, self.Vs, self.pre_arr, self.pre_samples):
                lst.pop(0)
            self.Bas.pop(), self.Bgs.pop()  # the reference's loop does not rotate Bas / Bgs (VINS.cpp:1160-1176), restated as is
            fm.remove_back_shift_depth(R0, P0, self.Rs[0] @ world.ric, self.Ps[0] + self.Rs[0] @ world.tic)
        else:                    # slideWindowNew (VINS.cpp:1204-1251): frame W-1 leaves, its IMU samples join the last interval
            (imu0, s_a), (_, s_b) = self.pre_samples[W - 2], self.pre_samples[W - 1]
            merged = s_a + s_b
            self.pre_samples[W - 2] = (imu0, merged)
            lin = self.pre_arr[W - 2]  # push_back keeps the integrator's own linearisation biases (integration_base.h:39-45)
            self.pre_arr[W - 2] = self._integrate(imu0[0], imu0[1], lin[11:14].copy(), lin[14:17].copy(), merged)
            self.pre_samples.pop(W - 1), self.pre_arr.pop(W - 1)
            for lst in (self.Ps, self.Rs, self.Vs, self.Bas, self.Bgs):
                lst.pop(W - 1)
            fm.remove_front(self.frame_count)
        fm.remove_failures()     # after the slide (VINS.cpp:469-470)
        return stats

    def errors(self):
        est = np.array([h[1] for h in self.history])
        tru = np.array([h[2] for h in self.history])
        d = est - tru
        d = d - d[0]  # the first window fixes the gauge up to its own (small) error
        return np.sqrt((d ** 2).sum(1))

    def close(self):
        self.fm.close()


class EstimatorLoop:
    """The same replay through the native estimator (vio_estimator_*, csrc/vio_estimator.cpp): this class only feeds IMU
    samples and image_msg lists and, once, the initial window (in place of solveInitial)."""

    def __init__(self, cfg, seed=1, init_noise=0.0, tracker=None, n_seq=1, world=None, self_init=False):
        self.cfg, self.W = cfg, cfg.window_size
        self.self_init = self_init   # True: the estimator's own solveInitial instead of handing the first window over
        self.tracker = tracker
        self.world = world or (ImageWorld(cfg, seed) if tracker is not None else SyntheticWorld(cfg, seed))
        self.est = pkg.estimator.Estimator(cfg, self.world.tic, self.world.ric, n_seq=n_seq)
        if self_init:
            self.est.enable_initialization(True)
        self.k = 0
        self.init_noise = init_noise
        self.rng = np.random.default_rng(seed + 99)
        self.init = []
        self.history = []   # (frame k, estimated position of the newest frame, true position, VioFrameResult)

    def feed_until_image(self):
        """IMU samples up to frame k, the initial window when it is due; returns image_msg of frame k as (ids, xyz)."""
        world, W, k, est = self.world, self.W, self.k, self.est
        if k == 0:
            a, w = world.imu(world.time(0))
            est.process_imu(world.dt, a, w)
        else:
            for a, w in world.imu_interval(k):
                est.process_imu(world.dt, a, w)
        if k <= W and not self.self_init:
            Pt, Rt, Vt = world.truth(k)
            n = self.init_noise
            P = Pt + self.rng.normal(0, 0.01 * n, 3)
            R = Rt @ synth.rotvec_to_rot(self.rng.normal(0, 0.005 * n, 3))
            V = Vt + self.rng.normal(0, 0.02 * n, 3)
            if k == 0:
                self.ba0, self.bg0 = world.ba + self.rng.normal(0, 0.01 * n, 3), world.bg + self.rng.normal(0, 0.001 * n, 3)
            self.init.append((world.time(k), P, R, V))
            if k == W:
                est.set_initial_state([i[0] for i in self.init], [i[1] for i in self.init], [i[2] for i in self.init],
                                      [i[3] for i in self.init], [self.ba0] * (W + 1), [self.bg0] * (W + 1))
        if self.tracker is not None:
            ids, xyz = self.tracker.read_images(world.render(k)[None], True)[0]
        else:
            ids, xyz = world.observe(k)
        self.k += 1
        return ids, xyz

    def step(self):
        k, W, world, est = self.k, self.W, self.world, self.est
        ids, xyz = self.feed_until_image()
        res = est.process_image(ids, xyz, world.time(k))
        if res.action == abi.VIO_FRAME_SOLVED:
            self.history.append((k, est.window()["Ps"][W].copy(), world.truth(k)[0], res))
        return res

    def errors(self):
        est = np.array([h[1] for h in self.history])
        tru = np.array([h[2] for h in self.history])
        if self.self_init:
            # the estimator chose its own world frame (gravity-aligned, yaw and origin arbitrary): align the first solved
            # position and the yaw (horizontal Procrustes) before comparing
            a, b = est - est[0], tru - tru[0]
            num = (a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]).sum()
            den = (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]).sum()
            th = np.arctan2(num, den)
            Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
            return np.sqrt(((a @ Rz.T - b) ** 2).sum(1))
        d = est - tru
        d = d - d[0]
        return np.sqrt((d ** 2).sum(1))

    def close(self):
        self.est.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--python-loop", action="store_true", help="window bookkeeping in this file instead of the native estimator")
    ap.add_argument("--self-init", action="store_true", help="the estimator initialises itself (solveInitial) instead of taking the first window")
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--images", action="store_true", help="render frames of a textured plane and run the KLT front-end on them")
    args = ap.parse_args()
    import torch  # noqa: F401  (HIP runtime first)
    cfg = abi.default_config()
    solver = pkg.backend.WindowSolver(cfg, max_batch=1)
    pre = lambda *a: pkg.backend.preintegrate(cfg, *a)
    tracker = pkg.frontend.FeatureTracker(cfg, n_seq=1) if args.images else None
    if args.python_loop:
        loop = ClosedLoop(cfg, lambda w: solver.solve([w])[0], pre, seed=args.seed, init_noise=1.0, tracker=tracker)
    else:
        loop = EstimatorLoop(cfg, seed=args.seed, init_noise=1.0, tracker=tracker, self_init=args.self_init)
    for _ in range(args.frames):
        loop.step()
    e = loop.errors()
    flags = [h[3]["iterations"] if args.python_loop else h[3].stats.iterations for h in loop.history]
    print("frames %d, solves %d, position error of the newest frame: rms %.4f m, max %.4f m, final %.4f m; mean iterations %.1f"
          % (args.frames, len(e), np.sqrt((e ** 2).mean()), e.max(), e[-1], np.mean(flags)))
    loop.close()


if __name__ == "__main__":
    main()
