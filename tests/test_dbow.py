"""Bag-of-words query of the loop-closure producer (DBoW2 as the app drives it): the CPU restatement
(oracle/vio_oracle_dbow.cpp) against independent numpy formulations, and the device kernels (csrc/vio_bow.hip) against the
restatement -- word ids, weights and BowVectors bit for bit, query results identical. PARITY UNPINNED: DBoW2 needs boost
and OpenCV headers that are not in the image, and the app's vocabulary file is not part of the reference tree; the
synthetic vocabularies below use the app's binary layout (loop/VocabularyBinary.hpp) with k = 10, L up to 6."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from helpers import pkg

loop = pkg.loop
_u64p, _i32p, _f64p = C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_double)


def make_vocabulary(k, L, seed, weighting=0, flip=24, stop_fraction=0.05):
    """A k-ary tree of depth L in the file layout: every child = its parent's descriptor with `flip` random bits flipped
    (so that the descent is decided by a few bits and ties do occur), leaves are words with positive idf-like weights
    (a few stopped: weight 0). Node ids breadth first, but written to the file in a shuffled order to exercise loadBin."""
    rng = np.random.default_rng(seed)
    desc = {0: rng.integers(0, 2 ** 63, 4, dtype=np.int64).astype(np.uint64)}
    nodes, words, level, nid = [], [], [0], 1
    for lev in range(1, L + 1):
        nxt = []
        for p in level:
            for _ in range(k):
                d = desc[p].copy()
                for b in rng.integers(0, 256, flip):
                    d[b >> 6] ^= np.uint64(1) << np.uint64(b & 63)
                desc[nid] = d
                leaf = lev == L
                w = 0.0 if (leaf and rng.random() < stop_fraction) else float(rng.uniform(0.5, 9.0)) if leaf else 0.0
                nodes.append((nid, p, w, d))
                if leaf:
                    words.append((nid, len(words)))
                nxt.append(nid)
                nid += 1
        level = nxt
    # the file order of a parent's children is the order loadBin pushes them: keep siblings in order, shuffle the families
    fam = {}
    for n in nodes:
        fam.setdefault(n[1], []).append(n)
    order = list(fam.keys())
    rng.shuffle(order)
    nodes = [n for p in order for n in fam[p]]
    perm = rng.permutation(len(words))
    words = [(words[i][0], int(j)) for i, j in enumerate(perm)]   # word ids are not in node order either
    return loop.make_vocabulary_blob(k, L, 0, weighting, nodes, words), desc


def bind_oracle():
    lib = H.oracle_lib()
    lib.oracle_voc_create.restype = C.c_void_p
    lib.oracle_voc_create.argtypes = [C.c_char_p, C.c_size_t]
    lib.oracle_voc_destroy.argtypes = [C.c_void_p]
    lib.oracle_voc_transform.argtypes = [C.c_void_p, _u64p, C.c_int, _i32p, _f64p]
    lib.oracle_voc_bow.argtypes = [C.c_void_p, _u64p, C.c_int, _i32p, _f64p, C.c_int]
    lib.oracle_db_create.restype = C.c_void_p
    lib.oracle_db_create.argtypes = [C.c_void_p]
    lib.oracle_db_destroy.argtypes = [C.c_void_p]
    lib.oracle_db_add.argtypes = [C.c_void_p, _i32p, _f64p, C.c_int]
    lib.oracle_db_query.argtypes = [C.c_void_p, _i32p, _f64p, C.c_int, C.c_int, C.c_int, _i32p, _f64p, C.c_int]
    return lib


class OracleVoc:
    def __init__(self, blob):
        self.lib = bind_oracle()
        self.h = self.lib.oracle_voc_create(blob, len(blob))
        assert self.h

    def transform(self, d):
        d = np.ascontiguousarray(d, np.uint64)
        w, ww = np.zeros(len(d), np.int32), np.zeros(len(d), np.float64)
        self.lib.oracle_voc_transform(self.h, d.ctypes.data_as(_u64p), len(d), w.ctypes.data_as(_i32p), ww.ctypes.data_as(_f64p))
        return w, ww

    def bow(self, d):
        d = np.ascontiguousarray(d, np.uint64)
        w, v = np.zeros(max(1, len(d)), np.int32), np.zeros(max(1, len(d)), np.float64)
        n = self.lib.oracle_voc_bow(self.h, d.ctypes.data_as(_u64p), len(d), w.ctypes.data_as(_i32p), v.ctypes.data_as(_f64p), len(w))
        assert n >= 0
        return w[:n].copy(), v[:n].copy()


def keyframe_descriptors(desc, leaves, rng, n, noise=10):
    """Descriptors near random leaves of the tree (plus a few duplicates: repeated words exercise addWeight)."""
    out = []
    for _ in range(n):
        d = desc[int(rng.choice(leaves))].copy()
        for b in rng.integers(0, 256, noise):
            d[b >> 6] ^= np.uint64(1) << np.uint64(b & 63)
        out.append(d)
    for i in rng.integers(0, n, n // 8):
        out.append(out[int(i)].copy())
    return np.array(out, np.uint64)


def popcount(x):
    return sum(bin(int(v)).count("1") for v in x)


@pytest.mark.parametrize("weighting", [0, 2])
def test_restatement_against_brute_force(weighting):
    k, L = 10, 3
    blob, desc = make_vocabulary(k, L, seed=5, weighting=weighting)
    voc = OracleVoc(blob)
    rng = np.random.default_rng(1)
    n_inner = sum(k ** l for l in range(L))               # nodes 0 .. n_inner-1 have children (breadth-first ids)
    leaves = list(range(n_inner, n_inner + k ** L))
    feats = keyframe_descriptors(desc, leaves, rng, 120)
    w, ww = voc.transform(feats)
    # independent descent on the breadth-first ids: children of node p are 1 + k p .. k p + k
    import struct
    hdr = struct.unpack("<6i", blob[:24])
    rec = {}
    for i in range(hdr[4]):
        nid, pid, wgt = struct.unpack("<iid", blob[24 + 48 * i: 24 + 48 * i + 16])
        rec[nid] = wgt
    w2n = {}
    for i in range(hdr[5]):
        nid, wid = struct.unpack("<ii", blob[24 + 48 * hdr[4] + 8 * i: 24 + 48 * hdr[4] + 8 * i + 8])
        w2n[nid] = wid
    for f, (wi, wwi) in zip(feats, zip(w, ww)):
        node = 0
        for _ in range(L):
            ch = list(range(1 + k * node, 1 + k * node + k))
            dist = [popcount(f ^ desc[c]) for c in ch]
            node = ch[int(np.argmin(dist))]                 # first minimum
        assert w2n[node] == wi and rec[node] == wwi
    # BowVector: dense formulation
    bw, bv = voc.bow(feats)
    dense = np.zeros(k ** L)
    for wi, wwi in zip(w, ww):
        if wwi > 0:
            dense[wi] = dense[wi] + wwi if weighting == 0 else wwi
    dense /= np.abs(dense).sum()
    assert list(bw) == list(np.nonzero(dense)[0]) and np.allclose(bv, dense[bw], rtol=1e-14, atol=0)
    assert abs(bv.sum() - 1.0) < 1e-12
    # query: score = 1 - ||v - w||_1 / 2 on the dense vectors, only entries sharing a word, only ids < max_id
    lib = voc.lib
    db = lib.oracle_db_create(voc.h)
    dens = []
    for e in range(12):
        fe = keyframe_descriptors(desc, leaves[:60] if e % 2 else leaves, rng, 80)
        ew, ev = voc.bow(fe)
        assert lib.oracle_db_add(db, ew.ctypes.data_as(_i32p), ev.ctypes.data_as(_f64p), len(ew)) == e
        d = np.zeros(k ** L)
        d[ew] = ev
        dens.append(d)
    ent, sc = np.zeros(32, np.int32), np.zeros(32, np.float64)
    n = lib.oracle_db_query(db, bw.ctypes.data_as(_i32p), bv.ctypes.data_as(_f64p), len(bw), 5, 9, ent.ctypes.data_as(_i32p), sc.ctypes.data_as(_f64p), 32)
    want = sorted(((1 - 0.5 * np.abs(dense - dens[e]).sum(), e) for e in range(9) if (dense * dens[e]).any()), key=lambda t: (-t[0], t[1]))[:5]
    assert n == len(want)
    assert [int(x) for x in ent[:n]] == [e for _, e in want]
    assert np.allclose(sc[:n], [s for s, _ in want], rtol=0, atol=1e-12)
    lib.oracle_db_destroy(db)
    lib.oracle_voc_destroy(voc.h)


def test_symbols_exported():
    lib = pkg.abi.load_product()
    for name in ("vio_vocabulary_create", "vio_vocabulary_load", "vio_vocabulary_destroy", "vio_vocabulary_info", "vio_vocabulary_get_device",
                 "vio_vocabulary_transform", "vio_bow_database_create", "vio_bow_database_destroy", "vio_bow_database_size",
                 "vio_bow_database_add", "vio_bow_database_query"):
        assert hasattr(lib, name)


@pytest.mark.gpu
@pytest.mark.parametrize("k,L,weighting,seed", [(10, 3, 0, 7), (10, 6, 0, 8), (9, 4, 2, 9), (20, 2, 1, 10)])
def test_device_matches_restatement(k, L, weighting, seed, tmp_path):
    blob, desc = make_vocabulary(k, L, seed=seed, weighting=weighting, flip=24 if L < 6 else 16)
    ovoc = OracleVoc(blob)
    path = tmp_path / "voc.bin"
    path.write_bytes(blob)
    voc = loop.BowVocabulary(path=str(path)) if seed % 2 else loop.BowVocabulary(blob=blob)
    try:
        info = voc.info()
        assert (info["k"], info["L"], info["weighting"], info["words"]) == (k, L, weighting, k ** L)
        rng = np.random.default_rng(seed)
        n_inner = sum(k ** l for l in range(L))
        leaves = np.arange(n_inner, n_inner + k ** L)
        # keyframes of ragged sizes, one empty, one made of a single repeated descriptor
        kfs = [keyframe_descriptors(desc, leaves[: max(50, len(leaves) // 50)], rng, n) for n in (300, 17, 1200, 64)]
        kfs.append(np.zeros((0, 4), np.uint64))
        kfs.append(np.repeat(kfs[0][:1], 40, axis=0))
        got = voc.transform(kfs)
        bows = []
        for d, (w, ww, bw, bv) in zip(kfs, got):
            rw, rww = ovoc.transform(d)
            assert np.array_equal(w, rw) and np.array_equal(ww, rww)          # bit-exact
            rbw, rbv = ovoc.bow(d)
            assert np.array_equal(bw, rbw) and np.array_equal(bv, rbv)         # bit-exact
            bows.append((bw, bv))
        # database: 40 entries, queries with max_id windows as detectLoop uses them (entry_id - dislocal)
        lib = ovoc.lib
        odb = lib.oracle_db_create(ovoc.h)
        db = loop.BowDatabase(voc, max_entries=64, max_total_words=1 << 16)
        try:
            ents = [keyframe_descriptors(desc, leaves[: max(50, len(leaves) // 50)], rng, int(n)) for n in rng.integers(30, 400, 40)]
            ebows = voc.transform(ents)
            for e, (_, _, bw, bv) in enumerate(ebows):
                assert db.add(bw, bv) == e
                assert lib.oracle_db_add(odb, bw.ctypes.data_as(_i32p), bv.ctypes.data_as(_f64p), len(bw)) == e
            queries = [b for b in bows if len(b[0])] + [(ebows[3][2], ebows[3][3])]
            max_ids = [40, 25, -1, 10, 5, 40][:len(queries)]
            res = db.query(queries, max_ids, max_results=7)
            for (qw, qv), mid, (ent, sc) in zip(queries, max_ids, res):
                re_, rs_ = np.zeros(64, np.int32), np.zeros(64, np.float64)
                n = lib.oracle_db_query(odb, np.ascontiguousarray(qw).ctypes.data_as(_i32p), np.ascontiguousarray(qv).ctypes.data_as(_f64p),
                                        len(qw), 7, mid, re_.ctypes.data_as(_i32p), rs_.ctypes.data_as(_f64p), 64)
                assert n == len(ent)
                assert np.array_equal(ent, re_[:n]) and np.array_equal(sc, rs_[:n])    # same order of additions: bit-exact
            # an entry queried against the database finds itself with score 1
            ent, sc = db.query([(ebows[3][2], ebows[3][3])], [-1], max_results=1)[0]
            assert ent[0] == 3 and abs(sc[0] - 1.0) < 1e-12
        finally:
            db.close()
            lib.oracle_db_destroy(odb)
    finally:
        voc.close()
        ovoc.lib.oracle_voc_destroy(ovoc.h)


@pytest.mark.gpu
@pytest.mark.parametrize("k,L,n_entries,seed", [(10, 4, 3000, 21), (10, 6, 1500, 22), (4, 3, 700, 23)])
def test_database_of_a_session(k, L, n_entries, seed):
    """Thousands of keyframes: the inverted file is merged entry by entry, queries (with detectLoop's max_id windows, with and
    without a cut) come back as TemplatedDatabase::queryL1 returns them -- same entries, same order, same bits. The three
    vocabularies put a query's words in very long (4^3 words), medium and mostly single-posting (10^6 words) runs."""
    blob, desc = make_vocabulary(k, L, seed=seed, flip=20 if L < 6 else 16)
    ovoc = OracleVoc(blob)
    voc = loop.BowVocabulary(blob=blob)
    lib = ovoc.lib
    odb = lib.oracle_db_create(ovoc.h)
    db = loop.BowDatabase(voc, max_entries=n_entries, max_total_words=n_entries * 260)
    try:
        rng = np.random.default_rng(seed)
        n_inner = sum(k ** l for l in range(L))
        leaves = np.arange(n_inner, n_inner + k ** L)
        pool = leaves[: max(40, min(len(leaves), 20000))]
        # 48 distinct keyframes (one of them without descriptors), visited again and again like places of a session
        kfs = [keyframe_descriptors(desc, pool, rng, int(n)) for n in rng.integers(20, 250, 47)] + [np.zeros((0, 4), np.uint64)]
        bows = [(b[2], b[3]) for b in voc.transform(kfs)]
        visit = rng.integers(0, len(bows), n_entries)
        for e, v in enumerate(visit):
            bw, bv = bows[v]
            assert db.add(bw, bv) == e
            assert lib.oracle_db_add(odb, np.ascontiguousarray(bw).ctypes.data_as(_i32p), np.ascontiguousarray(bv).ctypes.data_as(_f64p), len(bw)) == e
        fresh = [(b[2], b[3]) for b in voc.transform([keyframe_descriptors(desc, pool, rng, 180) for _ in range(5)])]
        queries = [bows[i] for i in (0, 5, 11, 46)] + fresh + [bows[47]]   # (the last one is the empty BowVector)
        max_ids = [n_entries, n_entries - 50, -1, 1, n_entries // 2, 0, 17, -1, n_entries, -1]
        for cut in (0, 4):
            res = db.query(queries, max_ids, max_results=cut)
            for (qw, qv), mid, (ent, sc) in zip(queries, max_ids, res):
                re_, rs_ = np.zeros(n_entries, np.int32), np.zeros(n_entries, np.float64)
                n = lib.oracle_db_query(odb, np.ascontiguousarray(qw).ctypes.data_as(_i32p), np.ascontiguousarray(qv).ctypes.data_as(_f64p),
                                        len(qw), cut, mid, re_.ctypes.data_as(_i32p), rs_.ctypes.data_as(_f64p), n_entries)
                assert n == len(ent)
                assert np.array_equal(ent, re_[:n]) and np.array_equal(sc, rs_[:n])
        hit = db.query([bows[5]], [-1], max_results=0)[0]
        assert set(np.flatnonzero(visit == 5)) <= set(hit[0][np.abs(hit[1] - 1.0) < 1e-12])   # every visit of the place scores 1
    finally:
        db.close()
        lib.oracle_db_destroy(odb)
        voc.close()
        lib.oracle_voc_destroy(ovoc.h)


@pytest.mark.gpu
def test_refuses_what_it_does_not_implement():
    blob, _ = make_vocabulary(4, 2, seed=1)
    import struct
    bad = struct.pack("<6i", 4, 2, 1, 0, *struct.unpack("<2i", blob[16:24])) + blob[24:]   # L2_NORM scoring
    with pytest.raises(RuntimeError):
        loop.BowVocabulary(blob=bad)
    with pytest.raises(RuntimeError):
        loop.BowVocabulary(blob=blob[:100])                                                 # truncated file
    voc = loop.BowVocabulary(blob=blob)
    try:
        with pytest.raises(RuntimeError):
            voc.transform([np.zeros((9000, 4), np.uint64)])                                 # more than 8192 descriptors
    finally:
        voc.close()


@pytest.mark.gpu
def test_refuses_corrupt_vocabularies_and_outlives_its_vocabulary():
    """A node id that appears twice, a parent cycle and a node that is its own parent are refused at load (a duplicate used
    to overrun the child table and leave the root as its own child: an endless descent on the device). A database keeps
    working, and can be destroyed, after its vocabulary is gone."""
    import struct
    blob, desc = make_vocabulary(4, 2, seed=3)
    n_nodes = struct.unpack("<i", blob[16:20])[0]
    rec = lambda i: 24 + 48 * i

    def with_node(i, nid=None, pid=None):
        b = bytearray(blob)
        o_nid, o_pid = struct.unpack("<2i", blob[rec(i):rec(i) + 8])
        b[rec(i):rec(i) + 8] = struct.pack("<2i", o_nid if nid is None else nid, o_pid if pid is None else pid)
        return bytes(b), o_nid, o_pid

    ids = [struct.unpack("<2i", blob[rec(i):rec(i) + 8]) for i in range(n_nodes)]
    dup, _, _ = with_node(1, nid=ids[0][0])                       # two records with the same node id
    with pytest.raises(RuntimeError):
        loop.BowVocabulary(blob=dup)
    inner = [i for i, (nid, pid) in enumerate(ids) if pid == 0][0]       # a child of the root ...
    kid = [i for i, (nid, pid) in enumerate(ids) if pid == ids[inner][0]][0]
    cyc, _, _ = with_node(inner, pid=ids[kid][0])                 # ... made the child of its own child: a cycle off the tree
    with pytest.raises(RuntimeError):
        loop.BowVocabulary(blob=cyc)
    selfp, _, _ = with_node(2, pid=ids[2][0])
    with pytest.raises(RuntimeError):
        loop.BowVocabulary(blob=selfp)
    voc = loop.BowVocabulary(blob=blob)
    db = loop.BowDatabase(voc, max_entries=4, max_total_words=256)
    rng = np.random.default_rng(5)
    leaves = np.arange(1 + 4, 1 + 4 + 16)
    (_, _, bw, bv), = voc.transform([keyframe_descriptors(desc, leaves, rng, 60)])
    assert db.add(bw, bv) == 0
    voc.close()                                                   # the vocabulary goes first
    assert db.add(bw, bv) == 1
    ent, sc = db.query([(bw, bv)], [-1], max_results=2)[0]
    assert sorted(ent.tolist()) == [0, 1] and abs(sc[0] - 1.0) < 1e-12
    db.close()
