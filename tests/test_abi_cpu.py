"""CPU checks of the drop-in boundary: the product library loads and exports every symbol include/vio_amd.h
declares (no compute calls: there is no GPU here), ctypes struct layouts match the C header."""
import ctypes as C
import os
import re
import subprocess

import helpers as H
from helpers import abi


def declared_symbols():
    txt = open(os.path.join(H.ROOT, "include", "vio_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vio_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(abi.PRODUCT_LIB):
        subprocess.check_call(["make", "-C", os.path.dirname(abi.PRODUCT_LIB)])
    lib = C.CDLL(abi.PRODUCT_LIB)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.vio_version.restype = C.c_char_p
    assert b"gfx950" in lib.vio_version()


def test_struct_layouts_match_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vio_amd.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(VioConfig),sizeof(VioPreintegration),sizeof(VioPrior),sizeof(VioWindow),sizeof(VioSolveStats),"
                   "offsetof(VioWindow,next_prior),offsetof(VioConfig,cauchy_a));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I" + os.path.join(H.ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(abi.VioConfig), C.sizeof(abi.VioPreintegration), C.sizeof(abi.VioPrior), C.sizeof(abi.VioWindow),
            C.sizeof(abi.VioSolveStats), abi.VioWindow.next_prior.offset,
            abi.VioConfig.cauchy_a.offset]
    assert got == want


def test_default_config_matches_library():
    lib = abi.load_product()
    c = abi.VioConfig()
    lib.vio_config_default(C.byref(c))
    d = abi.default_config()
    for k, _ in abi.VioConfig._fields_:
        assert getattr(c, k) == getattr(d, k), k


def test_product_preintegration_matches_reference_golden():
    import numpy as np
    d = np.load(H.GOLDEN + "/factors.npz")
    cfg = abi.default_config()
    lib = abi.load_product()
    for c in range(len(d["pre_n"])):
        n = int(d["pre_n"][c])
        out = abi.preintegrate_with(lib.vio_preintegrate, cfg, d["pre_acc0"][c], d["pre_gyr0"][c], d["pre_ba"][c],
                                    d["pre_bg"][c], d["pre_dt"][c][:n], d["pre_acc"][c][:n], d["pre_gyr"][c][:n])
        ref = d["pre_out"][c]
        assert np.allclose(out[:17], ref[:17], rtol=1e-12, atol=1e-15)
        assert H.relerr(out[17:242], ref[17:242]) < 1e-12
        assert H.relerr(out[242:], ref[242:]) < 1e-12


def test_hip_runtime_report_names_what_is_mapped():
    """vio_hip_runtime lists the libamdhip64 copies of this process as /proc/self/maps shows them (here: exactly one, the
    copy the library was linked against or the one PyTorch brought along, whichever was loaded first)."""
    lib = abi.load_product()
    paths = abi.hip_runtime(lib)
    mapped = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln})
    assert sorted(paths) == mapped and len(paths) == 1
