"""Host-side mirror of the loop-closure producer's descriptor side: KeyFrame::searchByDes / findConnectionWithOldFrame
(VINS_ios/loop/keyframe.cpp:161-187, 267-273). Plumbing over csrc/vio_loop.hip for tests; no compute here."""
import ctypes as C

import numpy as np

from . import abi

_u64p, _i32p, _fp, _u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_uint8)


def bind(lib, prefix="vio"):
    if prefix == "vio":
        lib.vio_matcher_create.argtypes = [C.POINTER(C.c_void_p)]
        lib.vio_matcher_destroy.argtypes = [C.c_void_p]
        lib.vio_matcher_search_by_des.argtypes = [C.c_void_p, C.c_int32, _i32p, _i32p, _u64p, _u64p, _i32p, _i32p]
        lib.vio_loop_find_connection.argtypes = [C.c_void_p, C.POINTER(abi.VioConfig), C.c_int32, _u64p, _fp, C.c_int32, _u64p,
                                                 _fp, _fp, _fp, _u8p, _i32p]
    return lib


class Matcher:
    def __init__(self):
        self.lib = bind(abi.load_product())
        self._h = C.c_void_p()
        rc = self.lib.vio_matcher_create(C.byref(self._h))
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_matcher_create failed rc=%d (a gfx950 device is required)" % rc)

    def close(self):
        if self._h:
            self.lib.vio_matcher_destroy(self._h)
            self._h = C.c_void_p()

    def search_by_des(self, cur_list, old_list):
        """cur_list / old_list: per pair uint64 arrays [n][4]. Returns per pair (best_index, best_dist)."""
        n_cur = np.array([len(c) for c in cur_list], np.int32)
        n_old = np.array([len(o) for o in old_list], np.int32)
        cur = np.ascontiguousarray(np.concatenate([np.asarray(c, np.uint64).reshape(-1, 4) for c in cur_list] + [np.zeros((0, 4), np.uint64)]))
        old = np.ascontiguousarray(np.concatenate([np.asarray(o, np.uint64).reshape(-1, 4) for o in old_list] + [np.zeros((0, 4), np.uint64)]))
        idx = np.zeros(max(1, int(n_cur.sum())), np.int32)
        dist = np.zeros_like(idx)
        rc = self.lib.vio_matcher_search_by_des(self._h, len(cur_list), n_cur.ctypes.data_as(_i32p), n_old.ctypes.data_as(_i32p),
                                                cur.ctypes.data_as(_u64p), old.ctypes.data_as(_u64p), idx.ctypes.data_as(_i32p),
                                                dist.ctypes.data_as(_i32p))
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_matcher_search_by_des failed rc=%d" % rc)
        out, o = [], 0
        for n in n_cur:
            out.append((idx[o:o + n].copy(), dist[o:o + n].copy()))
            o += n
        return out

    def find_connection(self, cfg, cur_desc, cur_pts, old_desc, old_pts):
        cur_desc = np.ascontiguousarray(cur_desc, np.uint64).reshape(-1, 4)
        old_desc = np.ascontiguousarray(old_desc, np.uint64).reshape(-1, 4)
        cur_pts = np.ascontiguousarray(cur_pts, np.float32).reshape(-1, 2)
        old_pts = np.ascontiguousarray(old_pts, np.float32).reshape(-1, 2)
        n = len(cur_desc)
        mo, mn = np.zeros((max(n, 1), 2), np.float32), np.zeros((max(n, 1), 2), np.float32)
        status, k = np.zeros(max(n, 1), np.uint8), C.c_int32()
        rc = self.lib.vio_loop_find_connection(self._h, C.byref(cfg), n, cur_desc.ctypes.data_as(_u64p), cur_pts.ctypes.data_as(_fp),
                                               len(old_desc), old_desc.ctypes.data_as(_u64p), old_pts.ctypes.data_as(_fp),
                                               mo.ctypes.data_as(_fp), mn.ctypes.data_as(_fp), status.ctypes.data_as(_u8p), C.byref(k))
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_loop_find_connection failed rc=%d" % rc)
        return mo[:n], mn[:n], status[:n], k.value


# ---- keyframe descriptor extraction: BriefExtractor::operator() (loop/keyframe.cpp:375-409) -----------------------------
def bind_brief(lib):
    lib.vio_brief_load_pattern.argtypes = [C.c_char_p, _i32p, _i32p, _i32p, _i32p, C.c_int32, _i32p]
    lib.vio_brief_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, _i32p, _i32p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.vio_brief_destroy.argtypes = [C.c_void_p]
    lib.vio_brief_get_device.argtypes = [C.c_void_p, _i32p]
    lib.vio_brief_extract.argtypes = [C.c_void_p, _u8p, C.c_int32, _fp, _i32p, C.c_int32, C.c_int32, _fp, _u64p, _i32p, _i32p]
    return lib


def load_pattern(path, cap=256):
    """-> (x1, y1, x2, y2) int32 arrays from an OpenCV FileStorage YAML file (Resources/brief_pattern.yml). Host code only."""
    lib = bind_brief(abi.load_product())
    arr = [np.zeros(cap, np.int32) for _ in range(4)]
    n = C.c_int32(0)
    rc = lib.vio_brief_load_pattern(str(path).encode(), *[a.ctypes.data_as(_i32p) for a in arr], cap, C.byref(n))
    if rc != abi.VIO_OK:
        raise RuntimeError("vio_brief_load_pattern failed rc=%d" % rc)
    return tuple(a[:n.value].copy() for a in arr)


class BriefExtractor:
    """FAST corners + window points -> BRIEF descriptors for a batch of keyframes (csrc/vio_brief.hip)."""

    def __init__(self, rows, cols, pattern, max_frames=1, max_keypoints=4096):
        self.lib = bind_brief(abi.load_product())
        self.rows, self.cols, self.max_frames, self.cap = rows, cols, max_frames, max_keypoints
        pat = [np.ascontiguousarray(p, np.int32) for p in pattern]
        self._h = C.c_void_p()
        rc = self.lib.vio_brief_create(rows, cols, max_frames, max_keypoints, *[p.ctypes.data_as(_i32p) for p in pat], len(pat[0]),
                                       C.byref(self._h))
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_brief_create failed rc=%d (a gfx950 device is required)" % rc)

    def close(self):
        if self._h:
            self.lib.vio_brief_destroy(self._h)
            self._h = C.c_void_p()

    def extract(self, frames, window_pts, fast_threshold=20, allow_cut=False):
        """frames: [n][rows][cols] uint8; window_pts: per frame float arrays [k][2]. Returns per frame
        (keypoints [m][2], descriptors [m][4] uint64, n_fast)."""
        frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, self.rows, self.cols)
        n = len(frames)
        nw = np.array([len(w) for w in window_pts], np.int32)
        stride = max(1, int(nw.max()) if n else 1)
        wp = np.zeros((n, stride, 2), np.float32)
        for f, w in enumerate(window_pts):
            if len(w):
                wp[f, :len(w)] = np.asarray(w, np.float32).reshape(-1, 2)
        kp = np.zeros((n, self.cap, 2), np.float32)
        desc = np.zeros((n, self.cap, 4), np.uint64)
        nf, nk = np.zeros(n, np.int32), np.zeros(n, np.int32)
        rc = self.lib.vio_brief_extract(self._h, frames.ctypes.data_as(_u8p), n, wp.ctypes.data_as(_fp), nw.ctypes.data_as(_i32p), stride,
                                        fast_threshold, kp.ctypes.data_as(_fp), desc.ctypes.data_as(_u64p), nf.ctypes.data_as(_i32p),
                                        nk.ctypes.data_as(_i32p))
        if rc != abi.VIO_OK and not (allow_cut and rc == abi.VIO_ECAP):
            raise RuntimeError("vio_brief_extract failed rc=%d" % rc)
        return [(kp[f, :nk[f]].copy(), desc[f, :nk[f]].copy(), int(nf[f])) for f in range(n)]


# ---- bag-of-words query (DBoW2: csrc/vio_bow.hip) ---------------------------------------------------------------------
_f64p = C.POINTER(C.c_double)


def make_vocabulary_blob(k, L, scoring, weighting, nodes, words):
    """Bytes of a vocabulary in the app's binary layout (loop/VocabularyBinary.hpp): nodes = iterable of
    (node_id, parent_id, weight, desc[4] uint64), words = iterable of (node_id, word_id)."""
    import struct
    nodes, words = list(nodes), list(words)
    out = [struct.pack("<6i", k, L, scoring, weighting, len(nodes), len(words))]
    for nid, pid, w, d in nodes:
        out.append(struct.pack("<iid4Q", nid, pid, w, *[int(x) for x in d]))
    for nid, wid in words:
        out.append(struct.pack("<ii", nid, wid))
    return b"".join(out)


def bind_bow(lib):
    vp = C.c_void_p
    lib.vio_vocabulary_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
    lib.vio_vocabulary_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.vio_vocabulary_destroy.argtypes = [vp]
    lib.vio_vocabulary_destroy.restype = None
    lib.vio_vocabulary_info.argtypes = [vp, _i32p]
    lib.vio_vocabulary_get_device.argtypes = [vp, _i32p]
    lib.vio_vocabulary_transform.argtypes = [vp, C.c_int32, _i32p, _u64p, _i32p, _f64p, _i32p, _i32p, _f64p, C.c_int32]
    lib.vio_bow_database_create.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.vio_bow_database_destroy.argtypes = [vp]
    lib.vio_bow_database_destroy.restype = None
    lib.vio_bow_database_size.argtypes = [vp, _i32p]
    lib.vio_bow_database_add.argtypes = [vp, C.c_int32, _i32p, _f64p, _i32p]
    lib.vio_bow_database_query.argtypes = [vp, C.c_int32, _i32p, _i32p, _f64p, C.c_int32, _i32p, C.c_int32, _i32p, _i32p, _f64p, C.c_int32]
    return lib


class BowVocabulary:
    """TemplatedVocabulary<FBrief> on the device: transform() for batches of keyframes."""

    def __init__(self, blob=None, path=None):
        self.lib = bind_bow(abi.load_product())
        self._h = C.c_void_p()
        rc = self.lib.vio_vocabulary_load(path.encode(), C.byref(self._h)) if path else \
            self.lib.vio_vocabulary_create(blob, len(blob), C.byref(self._h))
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_vocabulary_create failed rc=%d" % rc)

    def close(self):
        if self._h:
            self.lib.vio_vocabulary_destroy(self._h)
            self._h = C.c_void_p()

    def info(self):
        a = np.zeros(6, np.int32)
        self.lib.vio_vocabulary_info(self._h, a.ctypes.data_as(_i32p))
        return dict(zip(("k", "L", "scoring", "weighting", "nodes", "words"), [int(x) for x in a]))

    def transform(self, desc_list, bow_stride=None):
        """desc_list: per keyframe uint64 [n][4]. -> per keyframe (word_id[n], word_weight[n], bow_word[m], bow_value[m])."""
        n_desc = np.array([len(d) for d in desc_list], np.int32)
        desc = np.ascontiguousarray(np.concatenate([np.asarray(d, np.uint64).reshape(-1, 4) for d in desc_list] + [np.zeros((0, 4), np.uint64)]))
        total = int(n_desc.sum())
        stride = int(bow_stride or max(1, int(n_desc.max())))
        wid, ww = np.zeros(max(1, total), np.int32), np.zeros(max(1, total), np.float64)
        bc = np.zeros(len(desc_list), np.int32)
        bw, bv = np.zeros((len(desc_list), stride), np.int32), np.zeros((len(desc_list), stride), np.float64)
        rc = self.lib.vio_vocabulary_transform(self._h, len(desc_list), n_desc.ctypes.data_as(_i32p), desc.ctypes.data_as(_u64p),
                                               wid.ctypes.data_as(_i32p), ww.ctypes.data_as(_f64p), bc.ctypes.data_as(_i32p),
                                               bw.ctypes.data_as(_i32p), bv.ctypes.data_as(_f64p), stride)
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_vocabulary_transform failed rc=%d (bow_count %s)" % (rc, bc))
        out, o = [], 0
        for f, n in enumerate(n_desc):
            out.append((wid[o:o + n].copy(), ww[o:o + n].copy(), bw[f, :bc[f]].copy(), bv[f, :bc[f]].copy()))
            o += n
        return out


class BowDatabase:
    """TemplatedDatabase<FBrief>: add(BowVector), query(BowVector, max_results, max_id) for batches of queries."""

    def __init__(self, voc, max_entries=4096, max_total_words=1 << 22):
        self.voc, self.lib = voc, voc.lib
        self._h = C.c_void_p()
        rc = self.lib.vio_bow_database_create(voc._h, max_entries, max_total_words, C.byref(self._h))
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_bow_database_create failed rc=%d" % rc)

    def close(self):
        if self._h:
            self.lib.vio_bow_database_destroy(self._h)
            self._h = C.c_void_p()

    def add(self, word, value):
        word, value = np.ascontiguousarray(word, np.int32), np.ascontiguousarray(value, np.float64)
        e = C.c_int32(-1)
        rc = self.lib.vio_bow_database_add(self._h, len(word), word.ctypes.data_as(_i32p), value.ctypes.data_as(_f64p), C.byref(e))
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_bow_database_add failed rc=%d" % rc)
        return e.value

    def query(self, bows, max_id, max_results=0):
        """bows: per query (word[m], value[m]); max_id: per query. -> per query (entry[r], score[r]), best first."""
        nq = len(bows)
        stride = max(1, max(len(b[0]) for b in bows))
        bc = np.array([len(b[0]) for b in bows], np.int32)
        bw, bv = np.zeros((nq, stride), np.int32), np.zeros((nq, stride), np.float64)
        for q, (w_, v_) in enumerate(bows):
            bw[q, :len(w_)], bv[q, :len(v_)] = w_, v_
        mid = np.ascontiguousarray(max_id, np.int32)
        n = C.c_int32(0)
        self.lib.vio_bow_database_size(self._h, C.byref(n))
        rs = max(1, n.value)
        nr = np.zeros(nq, np.int32)
        ent, sc = np.zeros((nq, rs), np.int32), np.zeros((nq, rs), np.float64)
        rc = self.lib.vio_bow_database_query(self._h, nq, bc.ctypes.data_as(_i32p), bw.ctypes.data_as(_i32p), bv.ctypes.data_as(_f64p), stride,
                                             mid.ctypes.data_as(_i32p), max_results, nr.ctypes.data_as(_i32p), ent.ctypes.data_as(_i32p),
                                             sc.ctypes.data_as(_f64p), rs)
        if rc != abi.VIO_OK:
            raise RuntimeError("vio_bow_database_query failed rc=%d" % rc)
        return [(ent[q, :nr[q]].copy(), sc[q, :nr[q]].copy()) for q in range(nq)]
