"""Keyframe descriptor extraction, BriefExtractor::operator() (VINS_ios/loop/keyframe.cpp:395-409): FAST corners + window
points -> GaussianBlur -> BRIEF tests. Integer work: the device is BIT-EXACT against the restatement (oracle/
vio_oracle_brief.cpp). The restatement itself is checked against independent formulations of the definitions (OpenCV, which
owns FAST and GaussianBlur in the reference, is a binary that is not in the tree: parity unpinned)."""
import ctypes as C
import os

import numpy as np
import pytest
from scipy import ndimage

import helpers as H

pkg = H.pkg
loop, synth = pkg.loop, pkg.synth
_u8p, _fp, _ip, _u64p, _i32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint64),
                                C.POINTER(C.c_int32))
CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1),
          (-2, 2), (-1, 3)]


def pattern():
    d = np.load(os.path.join(H.ROOT, "tests", "golden", "brief_pattern.npz"))
    return d["x1"], d["y1"], d["x2"], d["y2"]


def olib():
    lib = H.oracle_lib()
    lib.oracle_gaussian_blur9.argtypes = [_u8p, C.c_int, C.c_int, _u8p]
    lib.oracle_fast9_16.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _ip]
    lib.oracle_brief_compute.argtypes = [_u8p, C.c_int, C.c_int, _fp, C.c_int, _i32p, _i32p, _i32p, _i32p, C.c_int, _u64p]
    lib.oracle_brief_extract.argtypes = [_u8p, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, C.c_int, C.c_int, _fp, _u64p,
                                         _ip, _ip]
    return lib


def oracle_blur(img):
    out = np.zeros_like(img)
    assert olib().oracle_gaussian_blur9(img.ctypes.data_as(_u8p), img.shape[0], img.shape[1], out.ctypes.data_as(_u8p)) == 0
    return out


def oracle_fast(img, thr=20, cap=100000):
    kp, n = np.zeros((cap, 2), np.float32), C.c_int(0)
    assert olib().oracle_fast9_16(img.ctypes.data_as(_u8p), img.shape[0], img.shape[1], thr, kp.ctypes.data_as(_fp), cap, C.byref(n)) == 0
    return kp[:min(n.value, cap)].copy(), n.value


def oracle_extract(img, wpts, pat, thr=20, cap=4096):
    wpts = np.ascontiguousarray(wpts, np.float32).reshape(-1, 2)
    kp, desc = np.zeros((cap, 2), np.float32), np.zeros((cap, 4), np.uint64)
    nf, nk = C.c_int(0), C.c_int(0)
    p = [np.ascontiguousarray(a, np.int32) for a in pat]
    rc = olib().oracle_brief_extract(img.ctypes.data_as(_u8p), img.shape[0], img.shape[1], wpts.ctypes.data_as(_fp), len(wpts), thr,
                                     *[a.ctypes.data_as(_i32p) for a in p], len(p[0]), cap, kp.ctypes.data_as(_fp),
                                     desc.ctypes.data_as(_u64p), C.byref(nf), C.byref(nk))
    return rc, kp[:nk.value].copy(), desc[:nk.value].copy(), nf.value


def make_image(seed, rows=120, cols=160):
    rng = np.random.default_rng(seed)
    img = synth.make_texture(rng, rows, cols)
    img = np.clip(img + 25.0 * (rng.random((rows, cols)) < 0.01) * rng.standard_normal((rows, cols)) * 4, 0, 255)
    img[20:40, 30:60] = 230  # flat bright box: strong corners at its vertices
    img[70:75, 100:140] = 15
    return np.ascontiguousarray(img, np.uint8)


def test_pattern_fixture_shape():
    x1, y1, x2, y2 = pattern()
    assert len(x1) == len(y1) == len(x2) == len(y2) == 256
    assert max(np.abs(a).max() for a in (x1, y1, x2, y2)) <= 24  # patch size 48


def test_blur_restatement_is_the_fixed_point_separable_filter():
    """Independent formulation: float Gaussian taps * 256 rounded, scipy's 1-D correlation with mirror (= REFLECT_101)
    borders on integers, (sum + 2^15) >> 16."""
    x = np.arange(9) - 4.0
    k = np.exp(-0.5 * x * x / 4.0).astype(np.float32)
    k = (k * np.float32(1.0 / k.astype(np.float64).sum())).astype(np.float32)
    taps = np.rint(k.astype(np.float64) * 256.0).astype(np.int64)
    assert taps.sum() in (255, 256, 257) and np.array_equal(taps, taps[::-1])
    for seed, shape in ((1, (120, 160)), (2, (37, 53)), (3, (9, 200))):
        img = make_image(seed, *shape)
        rowp = ndimage.correlate1d(img.astype(np.int64), taps, axis=1, mode="mirror")
        ref = (ndimage.correlate1d(rowp, taps, axis=0, mode="mirror") + (1 << 15)) >> 16
        assert np.array_equal(oracle_blur(img), np.clip(ref, 0, 255).astype(np.uint8))


def brute_is_corner(img, i, j, t):
    v = int(img[i, j])
    ring = [int(img[i + dy, j + dx]) for dx, dy in CIRCLE]
    for sign in (-1, 1):
        hit = [(x < v - t) if sign < 0 else (x > v + t) for x in ring]
        run = 0
        for k in range(32):
            run = run + 1 if hit[k % 16] else 0
            if run >= 9:
                return True
    return False


def test_fast_restatement_matches_the_segment_test_definition():
    img = make_image(4, 48, 64)
    rows, cols = img.shape
    t = 20
    score = np.zeros((rows, cols), np.int32)
    for i in range(3, rows - 3):
        for j in range(3, cols - 3):
            if brute_is_corner(img, i, j, t):
                s = t
                while s < 255 and brute_is_corner(img, i, j, s + 1):  # the largest threshold that still passes
                    s += 1
                score[i, j] = s
    want = []
    for i in range(3, rows - 3):
        for j in range(3, cols - 3):
            s = score[i, j]
            if s and all(s > score[i + a, j + b] for a in (-1, 0, 1) for b in (-1, 0, 1) if (a, b) != (0, 0)):
                want.append((j, i))
    kp, n = oracle_fast(img, t)
    assert n == len(want) and n > 10
    assert [tuple(p) for p in kp.astype(int)] == want


def test_brief_restatement_matches_the_formula():
    img = make_image(5)
    blur = oracle_blur(img)
    x1, y1, x2, y2 = pattern()
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.uniform(-5, [165, 125], (40, 2)), [[0, 0], [159.6, 119.6], [10.5, 3.25]]]).astype(np.float32)
    desc = np.zeros((len(pts), 4), np.uint64)
    p = [np.ascontiguousarray(a, np.int32) for a in (x1, y1, x2, y2)]
    assert olib().oracle_brief_compute(blur.ctypes.data_as(_u8p), 120, 160, pts.ctypes.data_as(_fp), len(pts),
                                       *[a.ctypes.data_as(_i32p) for a in p], 256, desc.ctypes.data_as(_u64p)) == 0
    for k, (px, py) in enumerate(pts):
        for i in range(256):
            ax, ay = int(np.float32(px) + np.float32(x1[i])), int(np.float32(py) + np.float32(y1[i]))
            bx, by = int(np.float32(px) + np.float32(x2[i])), int(np.float32(py) + np.float32(y2[i]))
            inside = 0 <= ax < 160 and 0 <= ay < 120 and 0 <= bx < 160 and 0 <= by < 120
            want = bool(inside and blur[ay, ax] < blur[by, bx])
            assert bool((int(desc[k, i >> 6]) >> (i & 63)) & 1) == want


def test_pattern_loader_reads_block_and_flow_sequences(tmp_path):
    x1, y1, x2, y2 = pattern()
    f = tmp_path / "p.yml"
    with open(f, "w") as fh:
        fh.write("%YAML:1.0\n")
        for name, arr in (("x1", x1), ("y1", y1)):
            fh.write(name + ":\n" + "".join("  - %d\n" % v for v in arr))
        for name, arr in (("x2", x2), ("y2", y2)):
            fh.write(name + ": [ " + ", ".join(str(v) for v in arr) + " ]\n")
    got = loop.load_pattern(f)
    for a, b in zip(got, (x1, y1, x2, y2)):
        assert np.array_equal(a, b)
    ref = "/root/reference/Resources/brief_pattern.yml"
    if os.path.exists(ref):  # the fixture IS the app's pattern file
        for a, b in zip(loop.load_pattern(ref), (x1, y1, x2, y2)):
            assert np.array_equal(a, b)
    with pytest.raises(RuntimeError):
        loop.load_pattern(tmp_path / "missing.yml")


# ---- device ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(120, 160), (97, 131), (480, 640)])
def test_device_extraction_is_bit_exact(shape):
    rows, cols = shape
    pat = pattern()
    frames = np.stack([make_image(10 + s, rows, cols) for s in range(3)])
    rng = np.random.default_rng(7)
    wpts = [rng.uniform(0, [cols, rows], (k, 2)).astype(np.float32) for k in (40, 0, 150)]
    wpts[0][:3] = [[0.2, 0.3], [cols - 0.6, rows - 0.7], [3.5, rows - 1.0]]  # patches hanging over the border
    ex = loop.BriefExtractor(rows, cols, pat, max_frames=3, max_keypoints=20000)
    try:
        out = ex.extract(frames, wpts)
        assert ex.lib.vio_brief_get_device(ex._h, C.byref(C.c_int32())) == 0
    finally:
        ex.close()
    for f in range(3):
        rc, kp, desc, nf = oracle_extract(frames[f], wpts[f], pat, cap=20000)
        assert rc == 0 and nf > 20
        gk, gd, gnf = out[f]
        assert gnf == nf and np.array_equal(gk, kp) and np.array_equal(gd, desc)
        assert np.array_equal(gk[nf:], wpts[f])  # the window points follow the FAST corners


@pytest.mark.gpu
def test_device_capacity_cut_keeps_the_window_points():
    pat = pattern()
    img = make_image(20)
    w = np.array([[50.0, 60.0], [80.5, 20.25]], np.float32)
    ex = loop.BriefExtractor(120, 160, pat, max_frames=1, max_keypoints=12)
    try:
        with pytest.raises(RuntimeError):
            ex.extract(img[None], [w])
        kp, desc, nf = ex.extract(img[None], [w], allow_cut=True)[0]
    finally:
        ex.close()
    rc, okp, odesc, onf = oracle_extract(img, w, pat, cap=12)
    assert rc == pkg.abi.VIO_ECAP and nf == onf > 10
    assert len(kp) == 12 and np.array_equal(kp, okp) and np.array_equal(desc, odesc) and np.array_equal(kp[-2:], w)


@pytest.mark.gpu
def test_descriptors_of_a_shifted_frame_match_through_search_by_des():
    """Producer chain: the window points of a keyframe and of the same scene shifted by (3, 2) px get descriptors on the
    device; searchByDes then pairs every point with its own shifted copy."""
    pat = pattern()
    rng = np.random.default_rng(3)
    big = synth.make_texture(rng, 300, 400)
    a = np.ascontiguousarray(big[20:260, 30:350], np.uint8)
    b = np.ascontiguousarray(big[18:258, 27:347], np.uint8)   # content moved by (+3, +2)
    pts = rng.uniform(40, [280, 200], (120, 2)).astype(np.float32)
    ex = loop.BriefExtractor(240, 320, pat, max_frames=2, max_keypoints=20000)
    m = loop.Matcher()
    try:
        (ka, da, nfa), (kb, db, nfb) = ex.extract(np.stack([a, b]), [pts, pts + np.float32([3, 2])])
        idx, dist = m.search_by_des([da[nfa:]], [db[nfb:]])[0]
    finally:
        ex.close(), m.close()
    assert np.mean(idx == np.arange(120)) > 0.97 and np.median(dist) <= 2
