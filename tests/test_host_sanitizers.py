"""The host-side C++ of the library (landmark store, estimator incl. solveInitial, initialisation pieces, PnP tracker
bookkeeping, measurement queue, recording readers) compiled with AddressSanitizer + UndefinedBehaviorSanitizer and
walked by tests/fuzz/host_sanity.cpp, with the device entry points stubbed inside that test binary."""
import os
import subprocess

import helpers as H


def test_host_code_walk_is_clean_under_asan_ubsan_and_tsan(tmp_path):
    csrc = os.path.join(H.ROOT, "vins-mobile_amd", "csrc")
    exe = str(tmp_path / "host_sanity")
    src = [os.path.join(H.ROOT, "tests", "fuzz", "host_sanity.cpp")] + [os.path.join(csrc, f + ".cpp") for f in (
        "vio_window", "vio_initial", "vio_fivepoint", "vio_host", "vio_estimator", "vio_pnp_tracker", "vio_replay")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + os.path.join(H.ROOT, "include"), "-I" + csrc] + src + ["-lz", "-lpthread", "-o", exe])
    for n_seq in ("3", "48"):      # 48 sequences engage the host thread pool
        r = subprocess.run([exe, n_seq], capture_output=True, text=True, timeout=600, env=dict(os.environ, VIO_AMD_HOST_THREADS="8"))
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
        assert "host_sanity: done" in r.stdout and "sfm ok 1" in r.stdout
        reached = int(r.stdout.split("estimator walked,")[1].split("frames")[0])
        assert reached >= 1            # solveInitial ran to the end at least once (the solve itself is stubbed)
    # the same walk under ThreadSanitizer: the per-sequence phases run on the pool's worker threads
    tsan = str(tmp_path / "host_sanity_tsan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I" + os.path.join(H.ROOT, "include"), "-I" + csrc] + src +
                          ["-lz", "-lpthread", "-o", tsan])
    r = subprocess.run([tsan, "48"], capture_output=True, text=True, timeout=900, env=dict(os.environ, VIO_AMD_HOST_THREADS="8"))
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout[-1000:] + r.stderr[-4000:]
    assert "host_sanity: done" in r.stdout


def test_kernel_sources_are_clean_under_asan_ubsan(tmp_path):
    """solver_core.h / marg_core.h compiled for the host with -DVIO_SIMT -- the DEVICE sections, one fiber per work-item
    (tests/emul/simt.h) -- and pnp_core.h with -DVIO_EMUL, both WITH sanitizers, run over 26 + 7 windows
    (tests/fuzz/run_emul_sanitized.py): index arithmetic of the kernel source against the packed batch arrays, the carved
    LDS (a heap buffer here) and the scratch arrays."""
    import sys
    csrc = os.path.join(H.ROOT, "vins-mobile_amd", "csrc")
    emul = os.path.join(H.ROOT, "tests", "emul")
    flags = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-psabi", "-fsanitize=address,undefined",
             "-fno-sanitize-recover=undefined", "-I" + os.path.join(H.ROOT, "include"), "-I" + csrc, "-I" + emul, "-shared"]
    so_b, so_p = str(tmp_path / "simt_b.so"), str(tmp_path / "emul_p.so")
    subprocess.check_call(flags + ["-DVIO_SIMT", "-o", so_b, os.path.join(emul, "simt_backend.cpp")])
    subprocess.check_call(flags + ["-DVIO_EMUL", "-o", so_p, os.path.join(emul, "emul_pnp.cpp")])
    pre = ":".join(subprocess.check_output(["g++", "-print-file-name=" + n], text=True).strip() for n in ("libasan.so", "libubsan.so"))
    # (fibers switch stacks behind the sanitizer's back: its fake-stack bookkeeping is off, heap checks are what matters)
    env = dict(os.environ, LD_PRELOAD=pre, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tests", "fuzz", "run_emul_sanitized.py"), so_b, so_p], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    assert "26 windows clean" in r.stdout and "7 windows clean" in r.stdout
