"""CPU test of the device passes of the landmark store (vins-mobile_amd/csrc/store_core.h): store_ingest / store_pack /
store_finish run on the SIMT emulator (one fiber per work-item, three lane orders) frame by frame against the host-side
list of vio_window.cpp (the restatement of FeatureManager, feature_manager.cpp:11-407) and against pack_window (batch.h)
on seeded streams: list entries, observations, depth bits, keyframe decisions, factor arrays and bucket layout must be
identical, through keyframes, non-keyframes, negative depths, a failure-detection reset and re-promotion.
Test-only build (tests/emul/simt_store.cpp); the product library has no CPU path."""
import ctypes as C
import glob
import os
import subprocess

import pytest

import helpers as H

EMUL_DIR = os.path.join(H.ROOT, "tests", "emul")


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(EMUL_DIR, "libvio_simt_store.so")
    csrc = os.path.join(H.ROOT, "vins-mobile_amd", "csrc")
    srcs = glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "vio_window.cpp"), os.path.join(EMUL_DIR, "simt_store.cpp"),
                                                   os.path.join(EMUL_DIR, "simt.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-psabi", "-DVIO_SIMT",
                               "-I" + os.path.join(H.ROOT, "include"), "-I" + csrc, "-I" + EMUL_DIR, "-shared", "-o", so,
                               os.path.join(EMUL_DIR, "simt_store.cpp")])
    l = C.CDLL(so)
    l.simt_store_fuzz.argtypes = [C.c_int] * 5 + [C.c_char_p, C.c_int, C.POINTER(C.c_longlong)]
    return l


# (seed, window size, lane order, landmarks in the scene); seed % 3 == 0: one frame fails the failure detection,
# seed % 4 == 1: a slow camera (most frames are not keyframes)
STREAMS = [(1, 10, 0, 400), (2, 10, 1, 600), (3, 10, 2, 1500), (4, 5, 3, 300), (5, 10, 1, 900), (6, 12, 1, 2500), (9, 10, 2, 800),
           (13, 20, 0, 1200), (17, 30, 3, 800)]


@pytest.mark.parametrize("seed,W,order,n_landmarks", STREAMS)
def test_store_passes_match_the_host_list(seed, W, order, n_landmarks, lib):
    buf = C.create_string_buffer(8192)
    stats = (C.c_longlong * 8)()
    bad = lib.simt_store_fuzz(seed, 60, W, order, n_landmarks, buf, 8192, stats)
    assert bad == 0, buf.value.decode()
    frames, sum_f, sum_m, keyframes, failures = stats[0], stats[1], stats[2], stats[3], stats[4]
    assert frames == 60 and sum_f > 10 * frames and sum_m > sum_f
    if seed % 3 == 0:
        assert failures == 1
    if seed % 4 == 1:
        assert keyframes < 2 * frames // 3   # the non-keyframe slide ran often
    else:
        assert keyframes > frames // 2


@pytest.mark.parametrize("seed,order", [(1, 0), (2, 1), (3, 2), (4, 3)])
def test_wave_preintegration_matches_the_host_restatement_bit_for_bit(seed, order, lib):
    """preint_core.h (one wave64 per interval, the 15 x 15 products spread over the lanes) against host::propagate
    (vio_preint.h, the restatement of IntegrationBase::propagate): 16 intervals of 1-24 samples, every second one integrated
    in two pieces (store, load, continue: the non-keyframe merge). Every double of the block and the carried last sample
    must have the same bits."""
    lib.simt_preint_fuzz.argtypes = [C.c_int] * 3
    assert lib.simt_preint_fuzz(seed, 16, order) == 0
