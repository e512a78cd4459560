import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import helpers as H
from helpers import pkg
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes, hip.hipMemcpy.argtypes = [C.POINTER(C.c_void_p), C.c_size_t], [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
n=256; rows,cols=640,480
rng=np.random.default_rng(0)
rgba=rng.integers(0,256,(n,rows,cols,4),dtype=np.uint8)
pp=pkg.frontend.Preprocessor(rows,cols,max_frames=n)
d_in,d_out=C.c_void_p(),C.c_void_p()
assert hip.hipMalloc(C.byref(d_in), rgba.nbytes)==0 and hip.hipMalloc(C.byref(d_out), n*rows*cols)==0
assert hip.hipMemcpy(d_in, rgba.ctypes.data, rgba.nbytes, 1)==0
for ch in (4,1):
    for it in range(6):
        pp.run_resident(d_in.value, ch, n, d_out.value); pp.sync()
    pp.kernel_ms()
    for it in range(10):
        pp.run_resident(d_in.value, ch, n, d_out.value); pp.sync()
    ms,k=pp.kernel_ms()
    by=(ch+3)*rows*cols*n
    print("channels %d: %d frames %.3f ms -> %.0f frames/s, %.0f GB/s algorithmic" % (ch, n, ms, n/ms*1e3, by/ms/1e6))
# host-buffer entry point (pageable numpy in, pageable numpy out)
import time
for it in range(3):
    pp.run(rgba)
t0 = time.perf_counter()
for it in range(5):
    pp.run(rgba)
dt = (time.perf_counter() - t0) / 5
print("host path, %d RGBA frames (%d MB in, %d MB out): %.2f ms per call -> %.0f frames/s" % (n, rgba.nbytes >> 20, 2 * n * rows * cols >> 20, dt * 1e3, n / dt))
