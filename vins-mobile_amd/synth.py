"""Seeded synthetic inputs for the VIO hot path: sliding windows for the back-end and
rendered image streams for the KLT front-end.  numpy only; nothing here computes the hot path itself.

Conventions follow the reference: gravity (0,0,9.805) with p_j = p_i + v_i dt - 1/2 g dt^2 + R_i dp
(VINS_ios/integration_base.h:171-198), extrinsic ric = ypr2R(0,0,180), tic = (0,0.092,0.01)
(global_param.hpp:23-25, global_param.cpp:36-41), quaternions stored x y z w (VINS.cpp:93-101).
"""
import numpy as np

from . import abi

GRAVITY = 9.805


# ---- SO(3) helpers (numpy) -------------------------------------------------------------------------
def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_to_rot(q, normalize=True):
    x, y, z, w = q / np.linalg.norm(q) if normalize else q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


def rotvec_to_rot(v):
    th = np.linalg.norm(v)
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def ypr_to_rot(y, p, r):
    """Utility::ypr2R with degrees (utility.hpp:95-118)."""
    y, p, r = np.deg2rad([y, p, r])
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    return Rz @ Ry @ Rx


class Trajectory:
    """Smooth analytic body trajectory: sums of sinusoids, <= ~0.5 m/s, <= ~20 deg/s."""

    def __init__(self, rng):
        self.pa = rng.uniform(0.15, 0.45, (3, 2))
        self.pf = rng.uniform(0.08, 0.35, (3, 2)) * 2 * np.pi
        self.pp = rng.uniform(0, 2 * np.pi, (3, 2))
        self.ra = rng.uniform(0.03, 0.12, (3, 2))
        self.rf = rng.uniform(0.08, 0.3, (3, 2)) * 2 * np.pi
        self.rp = rng.uniform(0, 2 * np.pi, (3, 2))

    def pos(self, t):
        return (self.pa * np.sin(self.pf * t + self.pp)).sum(1)

    def vel(self, t):
        return (self.pa * self.pf * np.cos(self.pf * t + self.pp)).sum(1)

    def acc(self, t):
        return (-self.pa * self.pf ** 2 * np.sin(self.pf * t + self.pp)).sum(1)

    def rot(self, t):
        return rotvec_to_rot((self.ra * np.sin(self.rf * t + self.rp)).sum(1))

    def omega_body(self, t, h=1e-5):
        R0, R1, R = self.rot(t - h), self.rot(t + h), self.rot(t)
        S = R.T @ (R1 - R0) / (2 * h)
        return np.array([S[2, 1] - S[1, 2], S[0, 2] - S[2, 0], S[1, 0] - S[0, 1]]) * 0.5


def ex_pose_default():
    ric = ypr_to_rot(0.0, 0.0, 180.0)
    q = rot_to_quat(ric)
    return np.array([0.0, 0.092, 0.01, q[0], q[1], q[2], q[3]])


def make_window(cfg, preintegrate, seed=42, n_features=150, W=None, imu_per_frame=10, frame_dt=0.1,
                pix_noise=0.5, perturb=True, with_loop=0, traj_seed=None, frame_offset=0, perturb_scale=1.0,
                long_track_prob=0.3):
    """One synthetic window.  `preintegrate(acc0, gyr0, ba, bg, dt, acc, gyr)` -> float64[PREINT_DOUBLES]
    is the IMU pre-integration entry point to use (product, oracle or reference)."""
    rng = np.random.default_rng(seed)
    # trajectory + true biases can be shared by consecutive windows of one sequence (traj_seed, frame_offset)
    rng_t = np.random.default_rng(traj_seed) if traj_seed is not None else rng
    W = W if W is not None else cfg.window_size
    P = W + 1
    traj = Trajectory(rng_t)
    t0 = rng_t.uniform(0, 20) + frame_offset * frame_dt
    ex = ex_pose_default()
    ric, tic = quat_to_rot(ex[3:]), ex[:3]
    ba_true = rng_t.normal(0, 0.02, 3)
    bg_true = rng_t.normal(0, 0.002, 3)
    ps = perturb_scale
    g = np.array([0, 0, GRAVITY])
    dt = frame_dt / imu_per_frame

    ft = t0 + frame_dt * np.arange(P)
    Rw = [traj.rot(t) for t in ft]
    Pw = [traj.pos(t) for t in ft]
    Vw = [traj.vel(t) for t in ft]

    def imu_at(t):
        R = traj.rot(t)
        a = R.T @ (traj.acc(t) + g) + ba_true + rng.normal(0, 0.02, 3)
        w = traj.omega_body(t) + bg_true + rng.normal(0, 0.002, 3)
        return a, w

    # bias estimates the pre-integrations were linearized at (Bas/Bgs when the frame arrived)
    ba_lin = ba_true + rng.normal(0, 0.01, 3) if perturb else ba_true.copy()
    bg_lin = bg_true + rng.normal(0, 0.001, 3) if perturb else bg_true.copy()
    pre = np.zeros((W, abi.PREINT_DOUBLES))
    a_prev, w_prev = imu_at(ft[0])
    for k in range(W):
        dts = np.full(imu_per_frame, dt)
        accs = np.zeros((imu_per_frame, 3))
        gyrs = np.zeros((imu_per_frame, 3))
        for s in range(imu_per_frame):
            accs[s], gyrs[s] = imu_at(ft[k] + (s + 1) * dt)
        pre[k] = preintegrate(a_prev, w_prev, ba_lin, bg_lin, dts, accs, gyrs)
        a_prev, w_prev = accs[-1], gyrs[-1]

    # feature tracks: start frame in {0..W-3}, length in {2..P-start} (VINS.cpp:531 filter holds by construction)
    starts = np.zeros(n_features, int)
    lens = np.zeros(n_features, int)
    for f in range(n_features):
        s = 0 if rng.uniform() < 0.45 else rng.integers(0, W - 2)
        lmax = P - s
        ln = lmax if rng.uniform() < long_track_prob else rng.integers(2, lmax + 1)
        starts[f], lens[f] = s, ln
    order = np.argsort(starts, kind="stable")  # f_manager.feature is ordered by first appearance
    starts, lens = starts[order], lens[order]

    half_x = 0.8 * cfg.cx / cfg.fx
    half_y = 0.8 * cfg.cy / cfg.fy
    host, target, feat, pts_i, pts_j, depth_true = [], [], [], [], [], []
    for f in range(n_features):
        s = starts[f]
        n_i = np.array([rng.uniform(-half_x, half_x), rng.uniform(-half_y, half_y), 1.0])
        d = rng.uniform(3.0, 10.0)
        p_w = Rw[s] @ (ric @ (n_i * d) + tic) + Pw[s]
        depth_true.append(d)
        obs_i = n_i.copy()
        obs_i[:2] += rng.normal(0, pix_noise / cfg.fx, 2)
        for k in range(1, lens[f]):
            j = s + k
            p_c = ric.T @ (Rw[j].T @ (p_w - Pw[j]) - tic)
            o = np.array([p_c[0] / p_c[2], p_c[1] / p_c[2], 1.0])
            o[:2] += rng.normal(0, pix_noise / cfg.fx, 2)
            host.append(s), target.append(j), feat.append(f), pts_i.append(obs_i), pts_j.append(o)
    depth_true = np.array(depth_true)

    pose = np.zeros((P, 7))
    sb = np.zeros((P, 9))
    for i in range(P):
        R, p, v = Rw[i], Pw[i].copy(), Vw[i].copy()
        ba, bg = ba_lin.copy(), bg_lin.copy()
        if perturb:
            p += rng.normal(0, 0.02 * ps, 3)
            R = R @ rotvec_to_rot(rng.normal(0, 0.01 * ps, 3))
            v += rng.normal(0, 0.05 * ps, 3)
        pose[i, :3], pose[i, 3:] = p, rot_to_quat(R)
        sb[i, :3], sb[i, 3:6], sb[i, 6:] = v, ba, bg
    inv_depth = 1.0 / depth_true
    if perturb:
        inv_depth = inv_depth * (1 + np.clip(0.1 * ps * rng.normal(0, 1, n_features), -0.8, 3.0))

    loop_frame = -1
    if with_loop:
        # config-5 style relocalization constraint (VINS.cpp:571-637): `with_loop` features observed in
        # window frame i are also matched in an "old" frame whose true pose is a perturbed copy of pose i.
        loop_frame = W // 2
        R_old = Rw[loop_frame] @ rotvec_to_rot(rng.normal(0, 0.03, 3))
        P_old = Pw[loop_frame] + rng.normal(0, 0.1, 3)
        cand = [f for f in range(n_features) if starts[f] <= loop_frame < starts[f] + lens[f]]
        rng.shuffle(cand)
        first_k = {}
        for k, f in enumerate(feat):
            first_k.setdefault(f, k)
        for f in sorted(cand[:with_loop]):
            s = starts[f]
            k0 = first_k[f]
            n_i = pts_i[k0]
            p_w = Rw[s] @ (ric @ (np.array([n_i[0], n_i[1], 1.0]) * depth_true[f]) + tic) + Pw[s]
            p_c = ric.T @ (R_old.T @ (p_w - P_old) - tic)
            o = np.array([p_c[0] / p_c[2], p_c[1] / p_c[2], 1.0])
            host.append(s), target.append(P), feat.append(f), pts_i.append(pts_i[k0]), pts_j.append(o)
        # keep factors grouped by feature (stable: window factors before the loop factor of a feature)
        idx = np.argsort(np.array(feat), kind="stable")
        host, target, feat = [host[i] for i in idx], [target[i] for i in idx], [feat[i] for i in idx]
        pts_i, pts_j = [pts_i[i] for i in idx], [pts_j[i] for i in idx]

    w = abi.Window(W, pose, sb, ex, inv_depth, host, target, feat, np.array(pts_i), np.array(pts_j), pre,
                   prior=None, marginalization_flag=abi.VIO_MARGIN_OLD, loop_frame=loop_frame)
    w.truth = dict(pose=np.array([np.r_[Pw[i], rot_to_quat(Rw[i])] for i in range(P)]), vel=np.array(Vw),
                   ba=ba_true, bg=bg_true, depth=depth_true)
    return w


# ---- images ----------------------------------------------------------------------------------------
def make_texture(rng, rows, cols):
    """Band-limited random texture with plenty of Shi-Tomasi corners (blurred dots + blocks)."""
    img = rng.uniform(60, 190, (rows // 8 + 2, cols // 8 + 2))
    img = np.kron(img, np.ones((8, 8)))[:rows, :cols]
    n_dots = rows * cols // 160
    ys, xs = rng.integers(2, rows - 2, n_dots), rng.integers(2, cols - 2, n_dots)
    amp = rng.uniform(-70, 70, n_dots)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            np.add.at(img, (ys + dy, xs + dx), amp * (0.5 if dy or dx else 1.0))
    k = np.array([1, 4, 6, 4, 1]) / 16.0
    img = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, img)
    img = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, img)
    return img


def _bilinear(img, x, y):
    h, w = img.shape
    x = np.clip(x, 0, w - 1.001)
    y = np.clip(y, 0, h - 1.001)
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    fx, fy = x - x0, y - y0
    return ((1 - fx) * (1 - fy) * img[y0, x0] + fx * (1 - fy) * img[y0, x0 + 1] +
            (1 - fx) * fy * img[y0 + 1, x0] + fx * fy * img[y0 + 1, x0 + 1])


def make_image_stream(seed, n_frames, rows=640, cols=480, max_shift=2.5, noise=1.0):
    """Frames of a large textured plane seen through a slowly translating / rotating / zooming view.
    Returns uint8 [n_frames, rows, cols] and the per-frame 2x3 affine (frame px -> texture px)."""
    rng = np.random.default_rng(seed)
    pad = 96
    tex = make_texture(rng, rows + 2 * pad, cols + 2 * pad)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
    frames = np.zeros((n_frames, rows, cols), np.uint8)
    affines = np.zeros((n_frames, 2, 3))
    vx, vy = rng.uniform(-max_shift, max_shift, 2)
    wr = rng.uniform(-0.002, 0.002)
    zs = rng.uniform(-0.001, 0.001)
    for f in range(n_frames):
        ang, zoom = wr * f, 1.0 + zs * f
        c, s = np.cos(ang) * zoom, np.sin(ang) * zoom
        cx, cy = cols / 2.0, rows / 2.0
        tx = pad + cx + vx * f + 3.0 * np.sin(0.3 * f)
        ty = pad + cy + vy * f + 3.0 * np.cos(0.23 * f)
        A = np.array([[c, -s, tx - c * cx + s * cy], [s, c, ty - s * cx - c * cy]])
        affines[f] = A
        u = A[0, 0] * xx + A[0, 1] * yy + A[0, 2]
        v = A[1, 0] * xx + A[1, 1] * yy + A[1, 2]
        img = _bilinear(tex, u, v) + rng.normal(0, noise, (rows, cols))
        frames[f] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return frames, affines


# ---- loop pose graph (keyfame_database.cpp:140-353) ------------------------------------------------------
def _ypr_to_R(ypr_deg):
    y, p, r = np.deg2rad(ypr_deg)
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    return Rz @ Ry @ Rx


def make_loop_keyframes(n=60, seed=0, n_loops=6, laps=1.15, radius=8.0, yaw_drift_deg=0.25, pos_drift=0.03, first_index=0):
    """A keyframe list as optimize4DoFLoopPoseGraph sees it: a (more than once around) circle walked by a drifting
    odometry; the last `n_loops` keyframes revisit the start and carry a loop to an early keyframe. Returns
    (keyframes as dicts for posegraph.keyframes_struct, total_length, true translations)."""
    rng = np.random.default_rng(seed)
    ang = np.linspace(0, 2 * np.pi * laps, n)
    true_t = np.stack([radius * np.cos(ang), radius * np.sin(ang), 0.3 * np.sin(3 * ang)], 1)
    true_ypr = np.stack([np.rad2deg(ang) + 90.0, 4.0 * np.sin(2 * ang), 3.0 * np.cos(ang)], 1)
    true_ypr[:, 0] = (true_ypr[:, 0] + 180.0) % 360.0 - 180.0
    true_R = [_ypr_to_R(a) for a in true_ypr]
    # odometry: relative motions with a yaw bias and translation noise, chained from the true first pose
    R, t = [true_R[0]], [true_t[0]]
    for k in range(1, n):
        dR = true_R[k - 1].T @ true_R[k]
        dt = true_R[k - 1].T @ (true_t[k] - true_t[k - 1])
        dR = _ypr_to_R([yaw_drift_deg * (1 + 0.3 * rng.standard_normal()), 0, 0]) @ dR
        dt = dt + pos_drift * rng.standard_normal(3)
        t.append(t[-1] + R[-1] @ dt)
        R.append(R[-1] @ dR)
    kfs = []
    for k in range(n):
        kfs.append({"origin_t": t[k], "origin_r": R[k], "t": t[k], "r": R[k], "global_index": first_index + k, "has_loop": 0,
                    "is_looped": 0, "loop_index": -1, "loop_info": np.zeros(8)})
    total_length = float(np.sum(np.linalg.norm(np.diff(np.array(t), axis=0), axis=1)))
    lap_len = int(round(n / laps))
    for k in range(n - n_loops, n):
        old = max(0, k - lap_len)
        rel_t = true_R[old].T @ (true_t[k] - true_t[old]) + 0.01 * rng.standard_normal(3)
        rel_yaw = true_ypr[k, 0] - true_ypr[old, 0] + 0.05 * rng.standard_normal()
        rel_yaw = (rel_yaw + 180.0) % 360.0 - 180.0
        kfs[k]["has_loop"], kfs[k]["loop_index"] = 1, first_index + old
        kfs[old]["is_looped"] = 1
        info = np.zeros(8)
        info[0:3], info[3], info[7] = rel_t, 1.0, rel_yaw
        kfs[k]["loop_info"] = info
    return kfs, total_length, true_t
