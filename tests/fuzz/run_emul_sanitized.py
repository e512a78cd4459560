"""Runs the host builds of both solver kernels (window solver: the device sections on the wave64 SIMT emulator; PnP: the
one-thread emulation; built with -fsanitize=address,undefined by tests/test_host_sanitizers.py, this process started
with the sanitizer runtimes preloaded) over the golden windows, the odd-shaped seeded windows and the PnP cases: an
out-of-range index in the kernel SOURCE shows up as a sanitizer report."""
import sys, ctypes as C, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as H
from helpers import abi, pkg, synth
lib=C.CDLL(sys.argv[1]); lib.simt_solve_window.argtypes=[C.POINTER(abi.VioConfig), C.POINTER(abi.VioWindow), C.POINTER(abi.VioSolveStats), C.c_int, C.c_int, C.c_int]
def solver(nt, variant, order):
    return lambda cfg, win, st: lib.simt_solve_window(cfg, win, st, nt, variant, order)
n=0
for name in H.golden_window_names():
    cfg,w,d=H.load_golden_window(name)
    got,stats=H.solve_with(solver(256,-1,n%3),cfg,w); H.check_solution(got,stats,d,tol=1e-6,tol_prior=1e-5); n+=1
osolve,opre=H.oracle_backend()
for (W,F,loop,seed) in H.ODD_SHAPES:
    cfg=abi.default_config(window_size=W)
    w=synth.make_window(cfg, lambda *a: abi.preintegrate_with(opre,cfg,*a), seed=900+seed, n_features=F, W=W, with_loop=loop)
    got,gs=H.solve_with(solver(512 if seed%2 else 256,-1 if seed%3 else 0,seed%3),cfg,w); n+=1
print("backend emulation under ASan/UBSan:", n, "windows clean")
import test_pnp as T
lp=C.CDLL(sys.argv[2]); lp.emul_pnp_solve.argtypes=None
cfg=abi.default_config()
for c in T.CASES:
    w=T.make_window(cfg,*c); pkg.pnp.solve_with(lp.emul_pnp_solve,cfg,w)
print("pnp emulation under ASan/UBSan:", len(T.CASES), "windows clean")
