"""CPU: ptBlock sort/merge (BED side outputs).  The reference's own known-answer vectors
(tests/golden/ptblock_kats.json, from programs/src/secphase_test.c:30-231) pin BOTH the oracle's literal
restatement and the product's sweep-line implementation; random inputs cross-check the two."""
import ctypes as C
import json
import os

import numpy as np

from oracle import orc
from secphase_amd import api

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ptblock_kats.json")))


def _arr(x):
    return (C.c_int * max(len(x), 1))(*x)


def oracle_merge_v2(blocks, counts):
    L = orc.lib()
    n = len(blocks)
    s, e = _arr([b[0] for b in blocks]), _arr([b[1] for b in blocks])
    c = _arr(counts) if counts is not None else None
    L.orc_blocks_sort(n, s, e, c)
    os_, oe, oc = (C.c_int * (2 * n + 2))(), (C.c_int * (2 * n + 2))(), (C.c_int * (2 * n + 2))()
    m = L.orc_blocks_merge_v2(n, s, e, c, os_, oe, oc if counts is not None else None)
    return [[os_[i], oe[i], oc[i] if counts is not None else 0] for i in range(m)]


def product_merge(blocks, counts):
    n = len(blocks)
    i32 = lambda x: np.ascontiguousarray(np.array(x, np.int32))
    s, e = i32([b[0] for b in blocks]), i32([b[1] for b in blocks])
    c = i32(counts) if counts is not None else None
    cap = 2 * n + 2
    os_, oe, oc = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None
    L = api.lib()
    L.spx_merge_blocks_count.argtypes = [C.c_int32] + [C.POINTER(C.c_int32)] * 6 + [C.c_int32]
    m = L.spx_merge_blocks_count(n, p(s), p(e), p(c), p(os_), p(oe), p(oc), cap)
    assert m >= 0
    return [[int(os_[i]), int(oe[i]), int(oc[i])] for i in range(m)]


def test_reference_kats_sort(built):
    L = orc.lib()
    for v in KATS["sort_by_rfs"]:
        b = v["blocks"]
        s, e = _arr([x[0] for x in b]), _arr([x[1] for x in b])
        L.orc_blocks_sort(len(b), s, e, None)
        assert list(s)[:len(b)] == v["sorted_starts"]


def test_reference_kats_merge_v1(built):
    L = orc.lib()
    for v in KATS["merge_v1"]:
        b = v["blocks"]
        n = len(b)
        s, e = _arr([x[0] for x in b]), _arr([x[1] for x in b])
        L.orc_blocks_sort(n, s, e, None)
        os_, oe = (C.c_int * (n + 1))(), (C.c_int * (n + 1))()
        m = L.orc_blocks_merge(n, s, e, None, os_, oe, None)
        assert [[os_[i], oe[i]] for i in range(m)] == v["merged"]


def test_reference_kats_merge_v2_with_count(built):
    for v in KATS["merge_v2_count"]:
        ones = [1] * len(v["blocks"])
        assert oracle_merge_v2(v["blocks"], ones) == v["merged"]
        assert product_merge(v["blocks"], ones) == v["merged"]


def test_product_merge_equals_oracle_on_random_blocks(built):
    rng = np.random.default_rng(11)
    for trial in range(200):
        n = int(rng.integers(1, 40))
        starts = rng.integers(0, 200, n)
        lens = rng.integers(0, 60, n)
        blocks = [[int(s), int(s + l)] for s, l in zip(starts, lens)]
        counts = [int(c) for c in rng.integers(1, 4, n)]
        assert product_merge(blocks, counts) == oracle_merge_v2(blocks, counts), blocks
        assert product_merge(blocks, None) == oracle_merge_v2(blocks, None), blocks


def test_bed_save_of_one_large_contig_on_several_threads(built, tmp_path):
    """few contigs, many blocks: the save sorts one contig's blocks on several threads (runs + merges) -- the same text as the
    single-threaded merge of spx_merge_blocks_count, with counts and for bare marker positions"""
    L = api.lib()
    rng = np.random.default_rng(5)
    n = 150000
    starts = rng.integers(0, 3_000_000, n).astype(np.int32)
    lens = rng.integers(0, 400, n).astype(np.int32)
    cnts = rng.integers(1, 4, n).astype(np.int32)
    for with_count in (1, 0):
        h = C.c_void_p()
        assert L.spx_bedset_create(C.byref(h)) == 0
        if with_count:
            for s, l, c in zip(starts.tolist(), lens.tolist(), cnts.tolist()):
                L.spx_bedset_add(h, b"ctg", s, s + l, c)
            want = product_merge([[s, s + l] for s, l in zip(starts.tolist(), lens.tolist())], cnts.tolist())
        else:
            L.spx_bedset_add_points.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32), C.c_int32]
            L.spx_bedset_add_points(h, b"ctg", starts.ctypes.data_as(C.POINTER(C.c_int32)), n)  # dense: one bit per position
            far = np.array([2_000_000_000, 5, 1_999_999_999], np.int32)  # ... and a second contig that is sparse: sorted
            L.spx_bedset_add_points(h, b"sparse", np.concatenate([starts[:70000] * 300, far]).astype(np.int32).ctypes.data_as(C.POINTER(C.c_int32)), 70003)
            want = product_merge([[s, s] for s in starts.tolist()], None)
            want_sparse = product_merge([[s, s] for s in (starts[:70000].astype(np.int64) * 300).tolist() + far.tolist()], None)
        path = str(tmp_path / f"x{with_count}.bed")
        assert L.spx_bedset_save(h, path.encode(), with_count) == 0
        L.spx_bedset_free(h)
        got = [l.split("\t") for l in open(path).read().splitlines()]
        if not with_count:
            sp = [g for g in got if g[0] == "sparse"]
            got = [g for g in got if g[0] == "ctg"]
            assert [(int(g[1]), int(g[2])) for g in sp] == [(w[0], w[1] + 1) for w in want_sparse]
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g[0] == "ctg" and int(g[1]) == w[0] and int(g[2]) == w[1] + 1 and (not with_count or int(g[3]) == w[2])


def test_bed_point_sets_and_blocks_on_one_contig(built, tmp_path):
    """marker positions are kept as a set of bits while they arrive; blocks on the same contig go through the general merge together
    with them -- with and without counts the text equals the merge of everything as plain blocks"""
    L = api.lib()
    L.spx_bedset_add_points.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32), C.c_int32]
    rng = np.random.default_rng(9)
    pts = np.concatenate([rng.integers(0, 300_000, 5000), rng.integers(2_000_000, 2_000_400, 300), [0, 65535, 65536, 131071]]).astype(np.int32)
    blocks = [[int(s), int(s + l)] for s, l in zip(rng.integers(0, 300_000, 400), rng.integers(0, 3000, 400))]
    cnts = [int(c) for c in rng.integers(1, 4, 400)]
    for with_count in (0, 1):
        h = C.c_void_p()
        assert L.spx_bedset_create(C.byref(h)) == 0
        for k in range(0, len(pts), 97):  # in several calls, like one alignment at a time
            part = np.ascontiguousarray(pts[k:k + 97])
            L.spx_bedset_add_points(h, b"both", part.ctypes.data_as(C.POINTER(C.c_int32)), len(part))
        L.spx_bedset_add_points(h, b"only_points", pts.ctypes.data_as(C.POINTER(C.c_int32)), len(pts))
        for (s, e), c in zip(blocks, cnts):
            L.spx_bedset_add(h, b"both", s, e, c)
        assert L.spx_bedset_size(h) == 2 * len(pts) + len(blocks)
        path = str(tmp_path / f"m{with_count}.bed")
        assert L.spx_bedset_save(h, path.encode(), with_count) == 0
        L.spx_bedset_free(h)
        got = [l.split("\t") for l in open(path).read().splitlines()]
        as_blocks = [[int(p), int(p)] for p in pts.tolist()]
        want_both = product_merge(as_blocks + blocks, ([0] * len(as_blocks) + cnts) if with_count else None)
        want_pts = product_merge(as_blocks, ([0] * len(as_blocks)) if with_count else None)
        for name, want in (("both", want_both), ("only_points", want_pts)):
            g = [x for x in got if x[0] == name]
            assert [(int(x[1]), int(x[2])) for x in g] == [(w[0], w[1] + 1) for w in want], name
            if with_count:
                assert [int(x[3]) for x in g] == [w[2] for w in want], name


def test_bed_write_errors_are_reported_not_fatal(built):
    """a full device (ENOSPC on every write) must come back as an error code of spx_bedset_save -- not as SIGBUS out of a
    mapped file, and not as success with a truncated BED (the command line prints 'could not write the BED outputs')"""
    import ctypes as C
    import os
    if not os.path.exists("/dev/full"):
        pytest.skip("no /dev/full")
    L = api.lib()
    h = C.c_void_p()
    assert L.spx_bedset_create(C.byref(h)) == 0
    L.spx_bedset_add.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32]
    for k in range(1000):
        L.spx_bedset_add(h, b"ctg", 100 * k, 100 * k + 50, 1)
    assert L.spx_bedset_save(h, b"/dev/full", 1) != 0
    assert L.spx_bedset_save(h, b"/nonexistent-dir/x.bed", 1) != 0
    L.spx_bedset_free(h)
