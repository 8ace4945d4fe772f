"""CPU: the product's host logic (CIGAR/cs scan, markers, consensus blocks, BAQ window list,
marker table) against the oracle.  The DP itself is supplied by the oracle here (no GPU), the
device-side write-back / scoring rules are applied in numpy (tests/common.py emulate_plan)."""
import ctypes as C

import numpy as np
import pytest

from common import (HandBatch, HandRef, batch_qual_copy, emulate_plan, emulate_rows, replay_qual_edits,
                    small_genome)
from oracle import orc
from secphase_amd import api, records, synth


def _compare(genome_or_ref, batch, params):
    ref = genome_or_ref
    nre, res = orc.run_batch(batch, ref, params, threads=2, seed=1)
    plan = api.Plan(ref, batch, params)
    v = plan.view
    assert v.n_problems == sum(r.n_baq_calls for r in res if r.n_aln > 0)
    em = emulate_plan(plan, ref, params)
    n = batch.contents.n_groups
    n_disp = 0
    for g in range(n):
        if orc.lib().orc_group_is_dispatched(batch, g):
            n_disp += 1
            assert api.lib().spx_group_is_dispatched(batch, g) == 1
            sc, prim, mx, tie, ok = em[g]
            o = res[g]
            assert [o.score[a] for a in range(o.n_aln)] == sc, g
            assert o.prim_idx == prim
            assert [v.rfe[10 * list(v.grp_index[:v.n_groups]).index(g) + a] for a in range(o.n_aln)] == \
                   [o.rfe[a] for a in range(o.n_aln)]
        else:
            assert api.lib().spx_group_is_dispatched(batch, g) == 0
            assert g not in em
    assert n_disp == v.n_groups
    return plan, res


def test_plan_hifi(built):
    g = small_genome(synth.HIFI)
    r = g.reads(0, 40)
    _compare(g.ref, r.batch, records.preset("hifi"))


def test_plan_ont(built):
    g = small_genome(synth.ONT, n_paralogs=3)
    r = g.reads(0, 10)
    _compare(g.ref, r.batch, records.preset("ont", bandwidth=50))


def test_plan_edge_cases(built):
    g = small_genome(synth.HIFI, hardclip_frac=0.5, softclip_frac=0.5, shuffle_records=1, inverted_paralogs=1,
                     n_paralogs=3, max_secondaries=4, n_base_frac=0.002, read_len=5000)
    r = g.reads(100, 40)
    _compare(g.ref, r.batch, records.preset("hifi"))


def test_plan_md_only_records(built):
    g = small_genome(synth.HIFI, tag_mode=1, read_len=5000, max_secondaries=3, n_paralogs=2, hardclip_frac=0.3,
                     softclip_frac=0.3)
    r = g.reads(0, 30)
    assert all(r.batch.contents.cs_off[a] < 0 for a in range(r.batch.contents.n_alns))
    _compare(g.ref, r.batch, records.preset("hifi"))


def test_plan_mixed_lengths(built):
    g = small_genome(synth.MIXED, n_paralogs=7, contig_len=250000, max_read_len=40000)
    r = g.reads(7, 12)
    _compare(g.ref, r.batch, records.preset("hifi"))


def _compare_quals(ref, batch, params):
    """-w/--writeBam: the record qualities after calc_local_baq (all-rows work list) against the oracle's"""
    import copy
    p_all = copy.copy(params)
    p_all.flags = 1
    want, res = orc.run_batch_quals(batch, ref, params, batch_qual_copy(batch), threads=2)
    plan = api.Plan(ref, batch, p_all)
    got = replay_qual_edits(plan, emulate_rows(plan, ref, p_all), batch, p_all)
    assert plan.view.n_qedits > 0
    assert np.array_equal(got, want)
    assert not np.array_equal(want, batch_qual_copy(batch))  # BAQ did change something
    # the scores do not depend on the mode
    base = emulate_plan(api.Plan(ref, batch, params), ref, params)
    assert emulate_plan(plan, ref, p_all) == base


def test_plan_all_rows_quals_hifi(built):
    g = small_genome(synth.HIFI, read_len=4000, hardclip_frac=0.3, softclip_frac=0.4, max_secondaries=3, n_paralogs=2)
    r = g.reads(0, 12)
    _compare_quals(g.ref, r.batch, records.preset("hifi"))


def test_plan_all_rows_quals_ont(built):
    g = small_genome(synth.ONT, n_paralogs=3, read_len=3000)
    r = g.reads(3, 4)
    _compare_quals(g.ref, r.batch, records.preset("ont"))


def test_plan_no_baq_no_consensus(built):
    g = small_genome(synth.HIFI, read_len=3000)
    r = g.reads(0, 20)
    p = records.preset("hifi")
    p.baq_flag = 0
    plan, _ = _compare(g.ref, r.batch, p)
    assert plan.view.n_problems == 0
    p = records.preset("hifi")
    p.consensus = 0  # BAQ over whole confident blocks (thousands of rows per problem)
    g2 = small_genome(synth.HIFI, read_len=1500, max_secondaries=1)
    r2 = g2.reads(0, 6)
    _compare(g2.ref, r2.batch, p)


def test_plan_rejects_undefined_cigar_ops(built):
    seq = "ACGT" * 10
    hr = HandRef([("c0", seq * 4)])
    hb = HandBatch([("r", [(0, 0, 0, "20M5N20M", seq, 30, ":20:20"), (256, 0, 3, "40M", seq, 30, ":40")])])
    plan = api.Plan(hr.ref, hb.batch, records.preset("hifi"))
    assert plan.view.grp_error[0] == api.EUNSUPPORTED and plan.view.n_groups == 0
    o = orc.GroupResult()
    st = orc.Rand()
    orc.lib().orc_srand(C.byref(st), 1)
    assert orc.lib().orc_score_group(hb.batch, hr.ref, 0, C.byref(records.preset("hifi")), C.byref(st), C.byref(o), None,
                                     None, 0) == -2


def test_plan_rejects_alignment_without_aligned_bases(built):
    """U6: a record made of clips only (no aligner writes one; the reference would index an empty SEQ)"""
    seq = "ACGT" * 10
    hr = HandRef([("c0", seq * 4)])
    hb = HandBatch([("r", [(0, 0, 0, "40M", seq, 30, ":40"), (256, 0, 3, "40H", "", 30, "")]),
                    ("s", [(0, 0, 0, "40M", seq, 30, ":40"), (256, 0, 3, "38M", seq[:38], 30, ":38")])])
    plan = api.Plan(hr.ref, hb.batch, records.preset("hifi"))
    assert plan.view.grp_error[0] == api.EUNSUPPORTED and plan.view.grp_error[1] == 0 and plan.view.n_groups == 1
    _, res = orc.run_batch(hb.batch, hr.ref, records.preset("hifi"), threads=1, seed=1)
    assert res[0].n_aln < 0 and res[1].n_aln == 2


def test_plan_reports_records_without_cs_and_md(built):
    """cigar_it.c:64-67: the reference prints "At least one of the MD or CS tags should be present!" and exits"""
    seq = "ACGT" * 10
    hr = HandRef([("c0", seq * 4)])
    hb = HandBatch([("r", [(0, 0, 0, "40M", seq, 30, ":40"), (256, 0, 3, "38M", seq[:38], 30, None)])])
    plan = api.Plan(hr.ref, hb.batch, records.preset("hifi"))
    assert plan.view.grp_error[0] == api.ENOTAG and plan.view.n_groups == 0
    assert b"MD or CS" in api.lib().spx_strerror(api.ENOTAG)
    _, res = orc.run_batch(hb.batch, hr.ref, records.preset("hifi"), threads=1, seed=1)
    assert res[0].n_aln < 0


def test_host_tables_match_libm(built):
    import math
    thr = (C.c_double * 102)()
    mt = (C.c_double * 256)()
    ms = (C.c_double * 256)()
    api.lib().spx_host_tables(thr, mt, ms)
    f = orc.lib().orc_phred_from_posterior
    for k in range(1, 102):
        x = thr[k]
        # thr[k] is the largest x with phred(x) >= k: check both sides of the boundary on the double lattice
        assert int(-4.343 * math.log(x) + .499) >= k
        assert int(-4.343 * math.log(np.nextafter(x, 2.0)) + .499) < k
        for ulps in range(1, 40):
            x = np.nextafter(x, 0.0)
            assert int(-4.343 * math.log(x) + .499) >= k
    assert mt[0] == -93.0 and mt[93] == 0.0 and ms[10] == -10 - 10 * math.log(3)


def test_secondaries_that_repeat_the_primary_stay_on_the_host(built):
    """what crosses PCIe: a secondary whose SEQ / QUAL repeat the primary's (same strand, reverse complement, minus hard
    clips) is aliased; one changed quality or base and the record is transferred like any other"""
    import ctypes as C
    from common import small_genome
    from secphase_amd import api, records, synth
    L = api.lib()
    L.spx_stage_transfer_stats.argtypes = [C.POINTER(C.POINTER(records.SpxBatch)), C.c_int32, C.c_int, C.POINTER(C.c_int64)]
    g = small_genome(synth.HIFI, read_len=3000, max_secondaries=3, min_secondaries=1, n_paralogs=3, hardclip_frac=0.5, softclip_frac=0.5,
                     inverted_paralogs=1, shuffle_records=1)
    r = g.reads(0, 120)
    b = r.batch.contents

    def stats():
        arr = (C.POINTER(records.SpxBatch) * 1)(r.batch)
        out = (C.c_int64 * 4)()
        assert L.spx_stage_transfer_stats(arr, 1, 2, out) == 0
        return list(out)

    n_slots, n_alias, all_bytes, sent = stats()
    n_sec = sum(1 for a in range(b.n_alns) if b.flag[a] & 256)
    rev = sum(1 for a in range(b.n_alns) if (b.flag[a] & 256) and (b.flag[a] & 16))
    # a group transfers ONE record, the one holding most of the read (rarely two: hard clips at opposite ends); both strands occur
    assert n_slots == b.n_alns and b.n_alns - b.n_groups - 3 <= n_alias <= b.n_alns - b.n_groups and rev > 5
    n_sec = n_alias
    assert sent < 0.62 * all_bytes
    # one quality of one secondary changed: that record (only) is transferred
    hard = lambda a: (b.cigar[b.cigar_off[a]] & 15) == 5 or (b.cigar[b.cigar_off[a] + b.n_cigar[a] - 1] & 15) == 5
    a = next(a for a in range(b.n_alns) if (b.flag[a] & 256) and (b.flag[a] & 16) and hard(a))
    old = b.qual[b.qual_off[a] + 57]
    b.qual[b.qual_off[a] + 57] = (old + 1) % 60
    assert stats()[1] == n_sec - 1
    b.qual[b.qual_off[a] + 57] = old
    # one base changed (high nibble of a SEQ byte)
    a2 = next(a for a in range(b.n_alns) if (b.flag[a] & 256) and not (b.flag[a] & 16) and hard(a))
    o2 = b.seq4[b.seq_off[a2] + 20]
    b.seq4[b.seq_off[a2] + 20] = o2 ^ 0x30
    assert stats()[1] == n_sec - 1
    b.seq4[b.seq_off[a2] + 20] = o2
    assert stats()[1] == n_sec
    # any single base / quality of any record, first and last positions included: never missed by the word-wise compare
    import numpy as np
    rng = np.random.default_rng(4)
    missed = 0
    longest = set()
    for gi in range(b.n_groups):
        members = range(b.grp_first[gi], b.grp_first[gi + 1] if gi + 1 < b.n_groups else b.n_alns)
        longest.add(max(members, key=lambda x: (b.l_qseq[x], not (b.flag[x] & 256), -x)))   # ties: the primary, else the first
    rest = [a for a in range(b.n_alns) if a not in longest]     # (a change in a source only counts where a copy covers it)
    for a in rng.choice(rest, 40, replace=False):
        a = int(a)
        lq = b.l_qseq[a]
        for pos in (0, lq - 1, max(0, lq - 17), int(rng.integers(0, lq)), int(rng.integers(0, lq))):
            if rng.random() < 0.5:
                at, v = b.qual_off[a] + pos, b.qual[b.qual_off[a] + pos]
                b.qual[at] = (v + 7) % 60
                dropped = n_sec - stats()[1]
                b.qual[at] = v
            else:
                at, v = b.seq_off[a] + pos // 2, b.seq4[b.seq_off[a] + pos // 2]
                b.seq4[at] = v ^ (0x10 if pos % 2 == 0 else 0x01)
                dropped = n_sec - stats()[1]
                b.seq4[at] = v
            missed += dropped != 1
    assert missed <= 5   # (at most one of the 40 records is neither a source nor aliased: its 5 changes alter nothing)
    assert stats()[1] == n_sec


_SHARE_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from common import small_genome
from secphase_amd import api, records, synth
h = hashlib.sha256()
for kw, first, n, par in (
        (dict(platform=synth.MIXED, n_paralogs=7, contig_len=250000, max_read_len=40000), 7, 10, records.preset("hifi")),
        (dict(platform=synth.ONT, n_paralogs=3, read_len=6000), 0, 8, records.preset("ont", bandwidth=50)),
        (dict(platform=synth.HIFI, hardclip_frac=0.5, softclip_frac=0.5, shuffle_records=1, inverted_paralogs=1, n_paralogs=3, max_secondaries=4,
              n_base_frac=0.002, read_len=5000), 100, 30, records.preset("hifi")),
        (dict(platform=synth.HIFI, tag_mode=1, read_len=5000, max_secondaries=3, n_paralogs=2, hardclip_frac=0.3, softclip_frac=0.3), 0, 20,
         records.preset("hifi"))):
    plat = kw.pop("platform")
    g = small_genome(plat, **kw)
    r = g.reads(first, n)
    plan = api.Plan(g.ref, r.batch, par)
    v = plan.view
    h.update(np.array([v.n_problems, v.n_rows, v.n_groups, v.n_markers, v.n_qedits], np.int64).tobytes())
    for name, cnt in (("L", v.n_problems), ("R", v.n_problems), ("bw", v.n_problems), ("ref_rfs", v.n_problems), ("qry_nib", v.n_problems),
                      ("row_off", v.n_problems), ("n_rows_of", v.n_problems), ("rows", v.n_rows), ("row_expect", v.n_rows), ("row_rawq", v.n_rows),
                      ("mk_first", v.n_groups + 1), ("mk_row", v.n_markers), ("mk_qfix", v.n_markers), ("mk_is_match", v.n_markers),
                      ("mk_aln", v.n_markers), ("n_aln", v.n_groups), ("rfe", v.n_groups * 10), ("grp_error", n)):
        if cnt:
            h.update(np.ctypeslib.as_array(getattr(v, name), shape=(cnt,)).tobytes())
    assert v.n_problems > 0
print(h.hexdigest())
"""


def test_plan_does_not_depend_on_how_the_walks_are_shared(built):
    """Round 5: on the device the 64 lanes of a wave share ONE heavy alignment -- contiguous ranges of its marker columns (aln_fill_range /
    aln_filter_range) and of its BAQ blocks (plan_baq_range), cursors from binary searches, output offsets from the counting pass's
    per-range totals.  SPX_PLAN_PARTS=n makes the host plan walk the same ranges one after the other (last range first): every array of
    the work list must come out as with one range (which test_plan_* above check against the oracle)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for parts in ("1", "2", "7", "64"):
        p = subprocess.run([sys.executable, "-c", _SHARE_SCRIPT, root], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, SPX_PLAN_PARTS=parts))
        assert p.returncode == 0, p.stderr[-800:]
        digests[parts] = p.stdout.strip().splitlines()[-1]
    assert len(set(digests.values())) == 1, digests


def test_plan_does_not_depend_on_how_the_round_windows_are_found(built):
    """Round 5: the flanking windows of a consensus round come from a one-byte-per-column table of break rounds scanned eight columns at a
    time (spx_logic.h flank_break_rounds / flank_blocks_by_round) for groups of >= 1024 columns and the first 16 rounds, from the walk over
    the positions otherwise.  SPX_WINDOW_TABLE=min_cols,rounds (host plan only) moves both limits: the table for EVERY group, a table of 3
    rounds or of 1 that hands over to the walk in mid-group, no table at all -- every array of the work list must come out the same (and
    test_plan_* above check the default against the oracle)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for table in ("", "2,16", "2,3", "2,1", "2,0", "300,5"):
        env = dict(os.environ)
        env.pop("SPX_WINDOW_TABLE", None)
        if table:
            env["SPX_WINDOW_TABLE"] = table
        p = subprocess.run([sys.executable, "-c", _SHARE_SCRIPT, root], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-800:]
        digests[table] = p.stdout.strip().splitlines()[-1]
    assert len(set(digests.values())) == 1, digests
