"""Host-side mirror of the 4-DoF loop pose graph, KeyFrameDatabase::optimize4DoFLoopPoseGraph
(VINS_ios/loop/keyfame_database.cpp:140-353). Plumbing over csrc/vio_posegraph.hip (the solve) and
csrc/vio_posegraph_host.cpp (resampling / edge list / drift); no compute here.  The same structs are understood by
the test-only checkers (oracle_posegraph_*, ref_posegraph_optimize)."""
import ctypes as C

import numpy as np

from . import abi

_dp, _ip, _u8p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


class VioPoseGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("t", _dp), ("ypr", _dp), ("fixed_node", C.c_int32), ("n_edges", C.c_int32),
                ("edge_i", _ip), ("edge_j", _ip), ("edge_kind", _u8p), ("edge_meas", _dp)]


class VioPoseGraphKeyframe(C.Structure):
    _fields_ = [("origin_t", C.c_double * 3), ("origin_r", C.c_double * 9), ("t", C.c_double * 3), ("r", C.c_double * 9),
                ("global_index", C.c_int32), ("has_loop", C.c_int32), ("is_looped", C.c_int32), ("loop_index", C.c_int32),
                ("loop_info", C.c_double * 8)]


class Graph:
    """One pose graph in numpy arrays (node k = k-th keyframe from earliest_loop_index on)."""

    def __init__(self, t, ypr, edge_i, edge_j, edge_kind, edge_meas, fixed_node=0):
        self.t = np.ascontiguousarray(t, np.float64).reshape(-1, 3).copy()
        self.ypr = np.ascontiguousarray(ypr, np.float64).reshape(-1, 3).copy()
        self.edge_i = np.ascontiguousarray(edge_i, np.int32).copy()
        self.edge_j = np.ascontiguousarray(edge_j, np.int32).copy()
        self.edge_kind = np.ascontiguousarray(edge_kind, np.uint8).copy()
        self.edge_meas = np.ascontiguousarray(edge_meas, np.float64).reshape(-1, 6).copy()
        self.fixed_node = int(fixed_node)

    def copy(self):
        return Graph(self.t, self.ypr, self.edge_i, self.edge_j, self.edge_kind, self.edge_meas, self.fixed_node)

    def fill_struct(self, g):
        g.n_nodes, g.fixed_node, g.n_edges = len(self.t), self.fixed_node, len(self.edge_i)
        g.t, g.ypr = self.t.ctypes.data_as(_dp), self.ypr.ctypes.data_as(_dp)
        g.edge_i, g.edge_j = self.edge_i.ctypes.data_as(_ip), self.edge_j.ctypes.data_as(_ip)
        g.edge_kind, g.edge_meas = self.edge_kind.ctypes.data_as(_u8p), self.edge_meas.ctypes.data_as(_dp)

    def to_npz_dict(self, prefix):
        return {prefix + k: getattr(self, k) for k in ("t", "ypr", "edge_i", "edge_j", "edge_kind", "edge_meas")} | {
            prefix + "fixed_node": np.int32(self.fixed_node)}

    @staticmethod
    def from_npz_dict(d, prefix):
        return Graph(*[d[prefix + k] for k in ("t", "ypr", "edge_i", "edge_j", "edge_kind", "edge_meas")],
                     fixed_node=int(d[prefix + "fixed_node"]))


def bind_checker(lib, prefix):
    """oracle_posegraph_optimize / ref_posegraph_optimize(VioPoseGraph*, max_iterations, VioSolveStats*)."""
    fn = getattr(lib, prefix + "_posegraph_optimize")
    fn.argtypes = [C.POINTER(VioPoseGraph), C.c_int32, C.POINTER(abi.VioSolveStats)]
    return fn


def optimize_with(fn, graph, max_iterations=5):
    g, st = VioPoseGraph(), abi.VioSolveStats()
    graph.fill_struct(g)
    rc = fn(C.byref(g), max_iterations, C.byref(st))
    if rc != 0:
        raise RuntimeError("posegraph optimize failed rc=%d" % rc)
    return abi.stats_to_dict(st)


def bind(lib):
    lib.vio_posegraph_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.vio_posegraph_destroy.argtypes = [C.c_void_p]
    lib.vio_posegraph_get_device.argtypes = [C.c_void_p, _ip]
    lib.vio_posegraph_optimize.argtypes = [C.c_void_p, C.POINTER(VioPoseGraph), C.c_int32, C.c_int32, C.POINTER(abi.VioSolveStats)]
    bind_host(lib, "vio")
    return lib


def bind_host(lib, prefix):
    kfp = C.POINTER(VioPoseGraphKeyframe)
    getattr(lib, prefix + "_posegraph_build").argtypes = [kfp, C.c_int32, C.c_double, C.c_int32, C.c_int32, _dp, _dp, _u8p, C.c_int32,
                                                          _ip, _ip, _u8p, _dp, _ip]
    getattr(lib, prefix + "_posegraph_apply").argtypes = [kfp, C.c_int32, _dp, _dp, _u8p, _dp, _dp, _dp, _dp, _dp]
    return lib


def keyframes_struct(kfs):
    arr = (VioPoseGraphKeyframe * len(kfs))()
    for a, k in zip(arr, kfs):
        a.origin_t[:] = list(k["origin_t"])
        a.origin_r[:] = list(np.asarray(k["origin_r"], float).reshape(9))
        a.t[:] = list(k["t"])
        a.r[:] = list(np.asarray(k["r"], float).reshape(9))
        a.global_index, a.has_loop, a.is_looped, a.loop_index = int(k["global_index"]), int(k["has_loop"]), int(k["is_looped"]), int(k["loop_index"])
        a.loop_info[:] = list(k["loop_info"])
    return arr


def build_with(lib, prefix, kfs, total_length, max_frame_num=500, list_size=None, cap_edges=None):
    """-> (Graph, skip). Host side of keyfame_database.cpp:166-285."""
    n = len(kfs)
    arr = keyframes_struct(kfs)
    cap = cap_edges if cap_edges is not None else 6 * n + 8
    t, ypr, skip = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n, np.uint8)
    ei, ej, ek, em = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.uint8), np.zeros((cap, 6))
    ne = C.c_int32(0)
    rc = getattr(lib, prefix + "_posegraph_build")(arr, n, float(total_length), max_frame_num, list_size if list_size is not None else n,
                                                  t.ctypes.data_as(_dp), ypr.ctypes.data_as(_dp), skip.ctypes.data_as(_u8p), cap,
                                                  ei.ctypes.data_as(_ip), ej.ctypes.data_as(_ip), ek.ctypes.data_as(_u8p),
                                                  em.ctypes.data_as(_dp), C.byref(ne))
    if rc != 0:
        raise RuntimeError("posegraph build failed rc=%d" % rc)
    k = ne.value
    return Graph(t, ypr, ei[:k], ej[:k], ek[:k], em[:k], fixed_node=0), skip


def apply_with(lib, prefix, kfs, graph, skip):
    """-> dict(t, r, yaw_drift, r_drift, t_drift). Host side of keyfame_database.cpp:303-339."""
    n = len(kfs)
    arr = keyframes_struct(kfs)
    out_t, out_r = np.zeros((n, 3)), np.zeros((n, 9))
    yd, rd, td = C.c_double(0), np.zeros(9), np.zeros(3)
    skip = np.ascontiguousarray(skip, np.uint8)
    rc = getattr(lib, prefix + "_posegraph_apply")(arr, n, graph.t.ctypes.data_as(_dp), graph.ypr.ctypes.data_as(_dp),
                                                  skip.ctypes.data_as(_u8p), out_t.ctypes.data_as(_dp), out_r.ctypes.data_as(_dp),
                                                  C.byref(yd), rd.ctypes.data_as(_dp), td.ctypes.data_as(_dp))
    if rc != 0:
        raise RuntimeError("posegraph apply failed rc=%d" % rc)
    return {"t": out_t, "r": out_r.reshape(n, 3, 3), "yaw_drift": yd.value, "r_drift": rd.reshape(3, 3), "t_drift": td}


class PoseGraphOptimizer:
    """n_graphs independent pose graphs per launch, one workgroup each (csrc/vio_posegraph.hip)."""

    def __init__(self, max_nodes=512, max_edges=4096, n_graphs=1):
        self.lib = bind(abi.load_product())
        self._h = C.c_void_p()
        rc = self.lib.vio_posegraph_create(max_nodes, max_edges, n_graphs, C.byref(self._h))
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_posegraph_create failed rc=%d (a gfx950 device is required)" % rc)

    def close(self):
        if self._h:
            self.lib.vio_posegraph_destroy(self._h)
            self._h = C.c_void_p()

    def device(self):
        d = C.c_int32(-1)
        self.lib.vio_posegraph_get_device(self._h, C.byref(d))
        return d.value

    def optimize(self, graphs, max_iterations=5):
        """Updates every Graph in place; returns the list of stats dicts."""
        n = len(graphs)
        arr, st = (VioPoseGraph * n)(), (abi.VioSolveStats * n)()
        for g, a in zip(graphs, arr):
            g.fill_struct(a)
        rc = self.lib.vio_posegraph_optimize(self._h, arr, n, max_iterations, st)
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_posegraph_optimize failed rc=%d" % rc)
        return [abi.stats_to_dict(s) for s in st]
