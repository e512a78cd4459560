"""The image pre-step of the camera callback (cvtColor RGBA2GRAY + CLAHE clipLimit 3 on 8x8 tiles,
VINS_ios/ViewController.mm:432-437). CPU: the oracle's restatement of the published OpenCV algorithm against
independent numpy formulations of its definition. GPU: the HIP kernels bit-exact against the oracle."""
import numpy as np
import pytest

import helpers as H
from helpers import pkg, synth


def _scene(rng, rows, cols, rgba=True):
    tex = synth.make_texture(rng, rows, cols).astype(np.float64)
    # uneven illumination: what CLAHE is there to even out
    yy, xx = np.mgrid[0:rows, 0:cols]
    shade = 0.35 + 0.65 * (xx / cols) * (0.5 + 0.5 * yy / rows)
    g = np.clip(tex * shade, 0, 255)
    if not rgba:
        return g.astype(np.uint8)
    out = np.stack([g * 0.9 + rng.normal(0, 4, g.shape), g + rng.normal(0, 4, g.shape), g * 0.7 + rng.normal(0, 4, g.shape),
                    np.full_like(g, 255)], axis=-1)
    return np.clip(out, 0, 255).astype(np.uint8)


def test_gray_conversion_is_the_fixed_point_formula():
    rng = np.random.default_rng(0)
    px = rng.integers(0, 256, (64, 64, 4), dtype=np.uint8)
    gray, _ = H.oracle_preprocess(px)
    r, g, b = (px[..., k].astype(np.int64) for k in range(3))
    assert np.array_equal(gray, ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8))
    assert np.abs(gray.astype(np.float64) - (0.299 * r + 0.587 * g + 0.114 * b)).max() <= 0.51
    white = np.full((64, 64, 4), 255, np.uint8)
    assert H.oracle_preprocess(white)[0].min() == 255               # the three weights sum to exactly 1 << 14


def test_without_clipping_clahe_is_tilewise_histogram_equalization():
    """clipLimit 0 switches the clipping off: each tile's LUT is round(cdf * 255 / area) and a pixel at a tile centre
    (where the bilinear weights are 1, 0, 0, 0) takes exactly its own tile's LUT value."""
    rng = np.random.default_rng(1)
    img = _scene(rng, 128, 160, rgba=False)
    _, eq = H.oracle_preprocess(img, clip_limit=0.0, tiles_x=4, tiles_y=4)
    tw, th = 40, 32
    for ty in range(4):
        for tx in range(4):
            tile = img[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            cdf = np.cumsum(np.bincount(tile.ravel(), minlength=256))
            lut = np.rint(cdf.astype(np.float32) * np.float32(255.0 / (tw * th))).astype(np.int64)
            # x*inv_tw - 0.5 is integral at x = tw/2 + k*tw: weights (1, 0)
            y, x = ty * th + th // 2, tx * tw + tw // 2
            assert eq[y, x] == lut[img[y, x]]


def test_clipping_bounds_the_slope_and_keeps_the_mapping_monotone():
    rng = np.random.default_rng(2)
    img = _scene(rng, 640, 480, rgba=False)
    _, eq = H.oracle_preprocess(img)
    # per tile centre the LUT is monotone non-decreasing in the input level, and one level never maps across more than
    # clip/area*255 + redistribution (~3/256*255 + 1) output levels: slope limited contrast
    tw, th = 60, 80
    for ty in (0, 3, 7):
        for tx in (0, 4, 7):
            probe = img.copy()
            y, x = ty * th + th // 2, tx * tw + tw // 2
            outs = []
            for v in range(0, 256, 5):
                probe[y, x] = v
                outs.append(H.oracle_preprocess(probe)[1][y, x])    # (one pixel barely changes the histogram)
            d = np.diff(np.array(outs, np.int64))
            assert d.min() >= -1 and d.max() <= 5 * 5
    # the shaded scene gains local contrast everywhere and uses the full range
    assert eq.std() > img.std() and eq.max() >= 250 and eq.min() <= 5
    dark = img[:, :120].astype(np.float64)
    assert eq[:, :120].astype(np.float64).std() > 1.5 * dark.std()


def test_non_divisible_sizes_extend_by_reflection():
    """A frame whose size is not a multiple of the grid is extended (reflect-101) for the histograms only: the output
    keeps the frame's size, and rows/columns far from the extension equal those of ... nothing simpler than itself, so
    check the definition on a case small enough to do by hand: 18 x 18, 4 x 4 tiles -> extended to 20 x 20, tile 5 x 5."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (18, 18), dtype=np.uint8)
    _, eq = H.oracle_preprocess(img, clip_limit=0.0, tiles_x=4, tiles_y=4)
    assert eq.shape == (18, 18)
    ext = np.pad(img, ((0, 2), (0, 2)), mode="reflect")              # numpy 'reflect' = BORDER_REFLECT_101
    luts = np.zeros((4, 4, 256), np.float32)
    for ty in range(4):
        for tx in range(4):
            cdf = np.cumsum(np.bincount(ext[ty * 5:ty * 5 + 5, tx * 5:tx * 5 + 5].ravel(), minlength=256))
            luts[ty, tx] = np.rint(cdf.astype(np.float32) * np.float32(255.0 / 25))
    f32 = np.float32
    for y in range(18):
        for x in range(18):
            tyf, txf = f32(y) * f32(1 / f32(5)) - f32(0.5), f32(x) * f32(1 / f32(5)) - f32(0.5)
            ty1, tx1 = int(np.floor(tyf)), int(np.floor(txf))
            ya, xa = f32(tyf - ty1), f32(txf - tx1)
            a, b = max(ty1, 0), min(ty1 + 1, 3)
            c, d = max(tx1, 0), min(tx1 + 1, 3)
            v = img[y, x]
            res = (luts[a, c, v] * (f32(1) - xa) + luts[a, d, v] * xa) * (f32(1) - ya) + (luts[b, c, v] * (f32(1) - xa) + luts[b, d, v] * xa) * ya
            assert eq[y, x] == int(np.clip(np.rint(res), 0, 255)), (y, x)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,rgba", [(640, 480, True), (640, 480, False), (480, 752, True), (243, 321, True),
                                            (243, 321, False), (136, 100, False)])
def test_device_prestep_is_bit_exact_against_the_oracle(rows, cols, rgba):
    rng = np.random.default_rng(rows + cols)
    frames = np.stack([_scene(rng, rows, cols, rgba) for _ in range(3)])
    pp = pkg.frontend.Preprocessor(rows, cols, max_frames=3)
    gray, eq = pp.run(frames)
    for k in range(3):
        g0, e0 = H.oracle_preprocess(frames[k])
        assert np.array_equal(gray[k], g0)
        assert np.array_equal(eq[k], e0), np.abs(eq[k].astype(int) - e0.astype(int)).max()
    ms, n = pp.kernel_ms()
    assert n == 1 and ms > 0
    pp.close()


@pytest.mark.gpu
def test_device_prestep_other_grids_and_limits():
    rng = np.random.default_rng(9)
    frame = _scene(rng, 480, 640, True)
    pp = pkg.frontend.Preprocessor(480, 640)
    for clip, tx, ty in [(0.0, 8, 8), (1.0, 4, 4), (40.0, 16, 16), (2.0, 5, 7)]:
        pp.set_clahe(clip, tx, ty)
        _, eq = pp.run(frame[None])
        assert np.array_equal(eq[0], H.oracle_preprocess(frame, clip, tx, ty)[1]), (clip, tx, ty)
    with pytest.raises(RuntimeError):
        pp.set_clahe(3.0, 17, 8)
    pp.close()


@pytest.mark.gpu
def test_resident_prestep_feeds_the_tracker_without_leaving_the_device():
    """RGBA frames resident in HBM -> pre-step on the tracker's stream -> vio_frontend_step_resident on the equalized
    frames: same tracker state as the host path fed with the oracle's equalized frames."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")                       # the runtime libvio_amd.so is linked against
    hip.hipMalloc.argtypes, hip.hipMemcpy.argtypes = [C.POINTER(C.c_void_p), C.c_size_t], [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    cfg = H.abi.default_config(image_rows=240, image_cols=320, max_corners=80, min_dist=15)
    gray_frames, _ = synth.make_image_stream(4, 4, rows=240, cols=320)
    assert gray_frames.shape == (4, 240, 320)
    rgba = np.stack([np.stack([f, f, f, np.full_like(f, 255)], axis=-1) for f in gray_frames])    # R=G=B -> gray == f
    pp = pkg.frontend.Preprocessor(240, 320, max_frames=4)
    d_rgba, d_eq = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_rgba), rgba.nbytes) == 0 and hip.hipMalloc(C.byref(d_eq), 4 * 240 * 320) == 0
    assert hip.hipMemcpy(d_rgba, rgba.ctypes.data, rgba.nbytes, 1) == 0           # hipMemcpyHostToDevice
    pp.run_resident(d_rgba.value, 4, 4, d_eq.value)
    pp.sync()
    eq = np.zeros((4, 240, 320), np.uint8)
    assert hip.hipMemcpy(eq.ctypes.data, d_eq, eq.nbytes, 2) == 0                  # hipMemcpyDeviceToHost
    hip.hipFree(d_rgba), hip.hipFree(d_eq)
    for k in range(4):
        assert np.array_equal(eq[k], H.oracle_preprocess(rgba[k])[1])
    a = pkg.frontend.FeatureTracker(cfg, n_seq=1)
    b = pkg.frontend.FeatureTracker(cfg, n_seq=1)
    for k in range(4):
        oa = a.read_images(eq[k][None], True)[0]
        ob = b.read_images(H.oracle_preprocess(rgba[k])[1][None], True)[0]
        assert list(oa[0]) == list(ob[0]) and np.array_equal(np.array(oa[1]), np.array(ob[1]))
    assert len(oa[0]) > 40
    a.close(), b.close(), pp.close()
