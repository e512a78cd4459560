"""Loop-closure producer, descriptor side: KeyFrame::searchByDes / findConnectionWithOldFrame
(VINS_ios/loop/keyframe.cpp:161-187, 267-273). Integer work: the HIP kernel is bit-exact against the plain restatement
in oracle/; the restatement is checked here against numpy on its own."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from helpers import abi, pkg

_u64p, _i32p, _fp, _u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_uint8)


def oracle():
    lib = H.oracle_lib()
    lib.oracle_search_by_des.argtypes = [_u64p, C.c_int32, _u64p, C.c_int32, _i32p, _i32p]
    lib.oracle_loop_find_connection.argtypes = [C.POINTER(abi.VioConfig), C.c_int32, _u64p, _fp, C.c_int32, _u64p, _fp, _fp, _fp,
                                                _u8p, _i32p]
    return lib


def oracle_search(cur, old):
    lib = oracle()
    cur, old = np.ascontiguousarray(cur, np.uint64).reshape(-1, 4), np.ascontiguousarray(old, np.uint64).reshape(-1, 4)
    idx, dist = np.zeros(max(1, len(cur)), np.int32), np.zeros(max(1, len(cur)), np.int32)
    lib.oracle_search_by_des(cur.ctypes.data_as(_u64p), len(cur), old.ctypes.data_as(_u64p), len(old), idx.ctypes.data_as(_i32p),
                             dist.ctypes.data_as(_i32p))
    return idx[:len(cur)], dist[:len(cur)]


def random_desc(rng, n):
    return rng.integers(0, 2 ** 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)


def noisy_copies(rng, base, flips):
    """Descriptors derived from `base` rows by flipping `flips` random bits each (a re-observed feature)."""
    out = base.copy()
    for i in range(len(out)):
        for b in rng.choice(256, size=flips, replace=False):
            out[i, b // 64] ^= np.uint64(1) << np.uint64(b % 64)
    return out


def popcount_matrix(cur, old):
    x = cur[:, None, :] ^ old[None, :, :]
    bits = np.unpackbits(x.view(np.uint8), axis=-1)
    return bits.reshape(len(cur), len(old), -1).sum(-1)


def test_oracle_search_by_des_matches_numpy():
    rng = np.random.default_rng(1)
    old = random_desc(rng, 300)
    cur = np.concatenate([noisy_copies(rng, old[rng.integers(0, 300, 60)], 20), random_desc(rng, 15)])
    old[17] = old[5]  # duplicated descriptor: the FIRST index wins
    cur[0] = old[5]
    idx, dist = oracle_search(cur, old)
    D = popcount_matrix(cur, old)
    assert np.array_equal(dist, D.min(1)) and np.array_equal(idx, D.argmin(1))
    assert idx[0] == 5 and dist[0] == 0
    # empty old keyframe, and the all-bits-differ case (distance 256 is never "< 256")
    i0, d0 = oracle_search(cur[:3], np.zeros((0, 4), np.uint64))
    assert list(i0) == [-1] * 3 and list(d0) == [256] * 3
    i1, d1 = oracle_search(cur[:1], ~cur[:1])
    assert i1[0] == -1 and d1[0] == 256


@pytest.mark.gpu
def test_search_by_des_bit_exact_many_pairs():
    rng = np.random.default_rng(2)
    m = pkg.loop.Matcher()
    sizes = [(150, 700), (1, 1), (70, 513), (0, 40), (33, 0), (260, 1500), (5, 64), (64, 65)]
    cur_list, old_list = [], []
    for nc, no in sizes:
        old = random_desc(rng, no)
        if no and nc:
            cur = np.concatenate([noisy_copies(rng, old[rng.integers(0, no, nc // 2)], 25), random_desc(rng, nc - nc // 2)])
            old[no // 2] = old[0]            # ties: first index
            cur[-1] = old[0]
        else:
            cur = random_desc(rng, nc)
        cur_list.append(cur), old_list.append(old)
    got = m.search_by_des(cur_list, old_list)
    for (gi, gd), cur, old in zip(got, cur_list, old_list):
        ri, rd = oracle_search(cur, old)
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd)
    m.close()


@pytest.mark.gpu
def test_find_connection_matches_oracle_and_rejects_outliers():
    rng = np.random.default_rng(3)
    cfg = abi.default_config()
    n, n_old = 120, 600
    # an "old" keyframe and the current one see the same plane under a small camera motion: the inlier matches obey one
    # fundamental matrix, 15 % of the current descriptors are re-observations matched to the wrong place
    old_pts = np.column_stack([rng.uniform(20, cfg.image_cols - 20, n_old), rng.uniform(20, cfg.image_rows - 20, n_old)]).astype(np.float32)
    old_desc = random_desc(rng, n_old)
    pick = rng.choice(n_old, n, replace=False)
    cur_desc = noisy_copies(rng, old_desc[pick], 18)
    th = 0.03
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    ctr = np.array([cfg.cx, cfg.cy])
    depth_par = rng.uniform(-6, 6, n)  # parallax along x: a real epipolar geometry, not a homography
    cur_pts = ((old_pts[pick] - ctr) @ R.T + ctr + np.column_stack([4.0 + depth_par, np.zeros(n)])).astype(np.float32)
    bad = rng.choice(n, 18, replace=False)
    cur_pts[bad] += rng.uniform(-60, 60, (18, 2)).astype(np.float32)
    m = pkg.loop.Matcher()
    mo, mn, status, k = m.find_connection(cfg, cur_desc, cur_pts, old_desc, old_pts)
    lib = oracle()
    omo, omn = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32)
    ost, ok = np.zeros(n, np.uint8), C.c_int32()
    cd, od = np.ascontiguousarray(cur_desc), np.ascontiguousarray(old_desc)
    lib.oracle_loop_find_connection(C.byref(cfg), n, cd.ctypes.data_as(_u64p), cur_pts.ctypes.data_as(_fp), n_old,
                                    od.ctypes.data_as(_u64p), old_pts.ctypes.data_as(_fp), omo.ctypes.data_as(_fp),
                                    omn.ctypes.data_as(_fp), ost.ctypes.data_as(_u8p), C.byref(ok))
    assert np.array_equal(mo, omo) and np.array_equal(mn, omn) and np.array_equal(status, ost) and k == ok.value
    assert np.array_equal(mo, old_pts[pick])                   # every descriptor found its source
    assert status[bad].sum() <= 3 and status.sum() >= n - 18 - 8  # the mismatched places are rejected, the rest kept
    # fewer than 8 matches: everything kept (keyframe.cpp:38)
    mo2, _, st2, k2 = m.find_connection(cfg, cur_desc[:5], cur_pts[:5], old_desc, old_pts)
    assert k2 == 5 and st2.all()
    m.close()
