"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle, bit for bit."""
import ctypes as C
import filecmp
import os

import numpy as np
import pytest

from common import oracle_probaln, small_genome
from oracle import orc
from secphase_amd import api, records, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(built):
    c = api.Context(0)
    yield c
    c.close()


def _rand_problem(rng, L, indel_rate, sub_rate, n_frac=0.0):
    ref = rng.integers(0, 4, size=L + 40).astype(np.uint8)
    q = []
    i = 0
    while i < len(ref) and len(q) < L:
        u = rng.random()
        if u < indel_rate / 2:
            i += 1  # deletion
            continue
        if u < indel_rate:
            q.append(rng.integers(0, 4))  # insertion
            continue
        b = ref[i]
        if rng.random() < sub_rate:
            b = (b + 1 + rng.integers(0, 3)) % 4
        q.append(b)
        i += 1
    qry = np.array(q, np.uint8)
    ref = ref[:max(1, i)]
    if n_frac > 0:
        ref = ref.copy()
        ref[rng.random(len(ref)) < n_frac] = 4
        qry[rng.random(len(qry)) < n_frac] = 4
    return ref, qry


def _check(ctx, probs, set_q, pars):
    refs = [p[0] for p in probs]
    qrys = [p[1] for p in probs]
    st, qq, ms = ctx.probaln_batch(refs, qrys, set_q, pars)
    for i, (r, q) in enumerate(probs):
        _, est, eq = oracle_probaln(r, q, set_q[i], pars[i][0], pars[i][1], pars[i][2])
        assert np.array_equal(st[i], est), f"state differs, problem {i} L={len(q)} R={len(r)} bw={pars[i][2]}"
        assert np.array_equal(qq[i], eq), f"q differs, problem {i} L={len(q)} R={len(r)} bw={pars[i][2]}"


def test_probaln_hifi_windows(ctx):
    rng = np.random.default_rng(7)
    probs = [_rand_problem(rng, int(rng.integers(600, 1001)), 0.002, 0.003) for _ in range(96)]
    pars = [(1e-4, 0.1, abs(len(r) - len(q)) + 20) for r, q in probs]
    _check(ctx, probs, [40] * len(probs), pars)


def test_probaln_ont_windows(ctx):
    rng = np.random.default_rng(8)
    probs = [_rand_problem(rng, int(rng.integers(300, 900)), 0.04, 0.02) for _ in range(48)]
    pars = [(1e-3, 0.1, abs(len(r) - len(q)) + 50) for r, q in probs]
    _check(ctx, probs, [20] * len(probs), pars)


def test_probaln_small_and_degenerate(ctx):
    rng = np.random.default_rng(9)
    probs, pars = [], []
    for L in (1, 2, 3, 5, 8, 13, 21, 22, 40, 41, 42, 60):
        for bw in (1, 3, 20):
            r, q = _rand_problem(rng, L, 0.05, 0.05)
            probs.append((r, q))
            pars.append((1e-4, 0.1, abs(len(r) - len(q)) + bw))
    # ref much shorter / longer than query
    probs.append((np.array([0, 1, 2], np.uint8), rng.integers(0, 4, 30).astype(np.uint8)))
    pars.append((1e-3, 0.1, 27 + 5))
    probs.append((rng.integers(0, 4, 90).astype(np.uint8), rng.integers(0, 4, 25).astype(np.uint8)))
    pars.append((1e-3, 0.1, 65 + 2))
    _check(ctx, probs, [30] * len(probs), pars)


def test_probaln_band_wider_than_query_and_reference(ctx):
    """the regime in which two readings of probaln.c's terminal guard differ (l_query <= bw, 2*bw+1 > l_ref; reached by
    `--ont -b 50` on blocks of <= 50 bases): the 240 problems tools/pin_htslib/make_problems.py sets aside for it -- states,
    qualities and, for a sample, every posterior product equal the oracle's (which follows the `u >= bw2*3+3` reading)"""
    import importlib.util
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pin_htslib", "make_problems.py")
    spec = importlib.util.spec_from_file_location("make_problems", tool)
    mp_ = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mp_)
    block = mp_.problems()[-240:]
    probs = [(np.array(r, np.uint8), np.array(q, np.uint8)) for r, q, *_ in block]
    pars = [(d, e, bw) for _, _, bw, d, e, _ in block]
    sq = [v[5] for v in block]
    _check(ctx, probs, sq, pars)
    for k in range(0, 240, 12):
        _posteriors_equal(ctx, probs[k][0], probs[k][1], sq[k], pars[k])


def test_probaln_ambiguous_bases(ctx):
    rng = np.random.default_rng(10)
    probs = [_rand_problem(rng, int(rng.integers(100, 400)), 0.01, 0.01, n_frac=0.03) for _ in range(32)]
    pars = [(1e-4, 0.1, abs(len(r) - len(q)) + 20) for r, q in probs]
    _check(ctx, probs, [40] * len(probs), pars)


def test_probaln_overflowed_backward_values(ctx):
    """found by tools/fuzz.py gpu (seed 777): unrelated sequences, 1 294 x 1 281, band 268.  The backward values overflow
    to inf; the reference multiplies the D values of row 1 by y = 0 (inf * 0 = NaN, which reaches M(1,k) and the MAP state
    of row 1) where the kernel used to SELECT 0.  States, qualities, 1/s and every z = f*b must equal the oracle's, NaNs
    included."""
    import json
    c = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "probaln_overflow_case.json")))
    ref = np.array([int(x) for x in c["ref"]], np.uint8)
    q = np.array([int(x) for x in c["qry"]], np.uint8)
    par = (c["d"], c["e"], c["bw"])
    st, qq, _ = ctx.probaln_batch([ref], [q], [c["set_q"]], [par])
    _, est, eq = oracle_probaln(ref, q, c["set_q"], *par)
    assert np.array_equal(st[0], est) and np.array_equal(qq[0], eq)
    sc, zM, zI = ctx.probaln_posteriors([ref], [q], [c["set_q"]], [par], which=0)
    s, oM, oI = orc.probaln_posteriors(ref, q, c["set_q"], *par)
    L = len(q)
    assert np.isnan(np.asarray(oM[0])).any()      # the case still exercises what it was kept for
    with np.errstate(divide="ignore", invalid="ignore"):
        assert np.array_equal(sc[1:L], 1.0 / s[1:L], equal_nan=True)
    assert np.array_equal(zM, oM, equal_nan=True) and np.array_equal(zI, oI, equal_nan=True)


def test_probaln_wide_bands(ctx):
    rng = np.random.default_rng(11)
    probs, pars = [], []
    for bw in (70, 130, 260):
        r, q = _rand_problem(rng, 700, 0.02, 0.02)
        probs.append((r, q))
        pars.append((1e-3, 0.1, abs(len(r) - len(q)) + bw))
    _check(ctx, probs, [20] * len(probs), pars)


def _posteriors_equal(ctx, ref, qry, set_q, par):
    """bit-exact comparison of the kernels' own numbers -- the per-row scaling factors and every posterior product
    z = f*b -- with the oracle's; state[]/q[] alone are too coarse (a phred and an arg-max hide 1e-6 errors)"""
    sc, zM, zI = ctx.probaln_posteriors(ref, qry, set_q, par)
    s, oM, oI = orc.probaln_posteriors(ref, qry, set_q, *par)
    L = len(qry)
    what = f"L={L} R={len(ref)} set_q={set_q} d,e,bw={par}"
    with np.errstate(divide="ignore"):
        assert np.array_equal(sc[1:L], 1.0 / s[1:L]), "1/s differs: " + what
    assert sc[L] == s[L] and sc[L + 1] == s[L + 1], "s[L], s[L+1] differ: " + what
    assert np.array_equal(zM, oM), "z (M state) differs: " + what
    assert np.array_equal(zI, oI), "z (I state) differs: " + what


def test_probaln_posteriors_bit_exact(ctx):
    """every band class, R != L, bands wider than the query, homopolymers (exact ties), ambiguous bases, odd
    HMM parameters: scaling factors and posterior products identical to the oracle's, bit for bit"""
    rng = np.random.default_rng(11)
    shapes = [(40, 70, 60), (200, 230, 130), (300, 330, 130), (387, 507, 143), (100, 100, 20), (262, 260, 21),
              (500, 497, 22), (481, 487, 23), (300, 300, 52), (150, 160, 55), (200, 190, 60), (120, 150, 130),
              (64, 64, 30), (9, 12, 5), (1, 1, 1), (3, 40, 37), (40, 3, 37), (700, 690, 300), (260, 262, 600), (90, 95, 45),
              (260, 262, 55), (240, 240, 58), (230, 233, 59), (250, 250, 62), (180, 170, 25)]  # W = 115, 117, 125, 125, 71
    for (L, R, bw) in shapes:
        for kind in ("homopolymer", "related", "unrelated"):
            if kind == "homopolymer":
                ref = np.full(R, 2, np.uint8)
                qry = np.full(L, 2, np.uint8)
                qry[L // 2] = 1
            elif kind == "related":
                ref = rng.integers(0, 4, R).astype(np.uint8)
                qry = np.resize(ref, L).copy()
                m = rng.random(L) < 0.02
                qry[m] = (qry[m] + 1) % 4
            else:
                ref = rng.integers(0, 4, R).astype(np.uint8)
                qry = rng.integers(0, 4, L).astype(np.uint8)
            for (d, e, sq) in ((1e-4, 0.1, 40), (1e-3, 0.1, 20), (0.01, 0.3, 27)):
                _posteriors_equal(ctx, ref, qry, sq, (d, e, abs(R - L) + bw if bw < 600 else bw))
    ref = rng.integers(0, 4, 300).astype(np.uint8)
    qry = ref[:290].copy()
    ref[rng.random(300) < 0.03] = 4
    qry[rng.random(290) < 0.03] = 4
    for bw in (20, 52, 130):
        _posteriors_equal(ctx, ref, qry, 40, (1e-4, 0.1, 10 + bw))


def test_probaln_posteriors_in_mixed_waves(ctx):
    """the same comparison for problems that share their wavefronts with others (sorted into waves by band width
    and length: a wave's interior-row range is the minimum over its problems)"""
    rng = np.random.default_rng(12)
    probs, sq, pars = [], [], []
    for _ in range(160):
        L = int(rng.integers(30, 420))
        R = max(1, L + int(rng.choice([0, 0, 1, -1, 2, -3, 5, 12, -20, 30])))
        ref = rng.integers(0, 4, R).astype(np.uint8) if rng.random() < 0.7 else np.full(R, 1, np.uint8)
        qry = np.resize(ref, L).copy()
        m = rng.random(L) < 0.02
        qry[m] = (qry[m] + 1) % 4
        probs.append((ref, qry))
        sq.append(40)
        pars.append((1e-4, 0.1, abs(R - L) + int(rng.choice([20, 20, 20, 21, 22, 50, 52, 60, 100, 130]))))
    refs, qrys = [p[0] for p in probs], [p[1] for p in probs]
    for which in rng.choice(len(probs), size=40, replace=False):
        sc, zM, zI = ctx.probaln_posteriors(refs, qrys, sq, pars, which=int(which))
        s, oM, oI = orc.probaln_posteriors(refs[which], qrys[which], sq[which], *pars[which])
        L = len(qrys[which])
        what = f"problem {which}: L={L} R={len(refs[which])} par={pars[which]}"
        assert np.array_equal(sc[1:L], 1.0 / s[1:L]) and sc[L] == s[L] and sc[L + 1] == s[L + 1], "s differs, " + what
        assert np.array_equal(zM, oM) and np.array_equal(zI, oI), "z differs, " + what


def _plans_equal(dev, host, n_input_groups):
    """every array of the work list the DEVICE built against the host plan (same spx_logic.h source run on the CPU,
    itself checked against the oracle in tests/test_host_plan.py), bit for bit"""
    a, b = dev.view, host.view
    for f in ("n_problems", "n_rows", "n_groups", "n_markers", "n_qedits"):
        assert getattr(a, f) == getattr(b, f), (f, getattr(a, f), getattr(b, f))

    def arr(v, name, n, dt=None):
        p = getattr(v, name)
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0)

    np_, nr, ng, nm, nq = a.n_problems, a.n_rows, a.n_groups, a.n_markers, a.n_qedits
    for name, n in (("L", np_), ("R", np_), ("bw", np_), ("ref_tid", np_), ("ref_rfs", np_), ("qry_nib", np_), ("row_off", np_),
                    ("n_rows_of", np_), ("rows", nr), ("row_expect", nr), ("row_rawq", nr), ("grp_index", ng),
                    ("mk_first", ng + 1), ("mk_row", nm), ("mk_qfix", nm), ("mk_is_match", nm), ("mk_aln", nm),
                    ("mk_first_of_pos", nm), ("n_aln", ng), ("sec_mask", ng), ("rfe", ng * 10), ("grp_error", n_input_groups),
                    ("qe_rec", nq), ("qe_pos", nq), ("qe_len", nq), ("qe_row0", nq)):
        x, y = arr(a, name, n), arr(b, name, n)
        assert np.array_equal(x, y), (name, np.flatnonzero(x != y)[:5])
    ha, hb_ = arr(a, "hmm", np_ * 16), arr(b, "hmm", np_ * 16)
    assert np.array_equal(ha.view(np.uint64), hb_.view(np.uint64)), "HMM constants differ"
    from common import nibbles
    for p in range(0, np_, max(1, np_ // 200)):
        assert np.array_equal(nibbles(a.qry4, a.qry_nib[p], a.L[p]), nibbles(b.qry4, b.qry_nib[p], b.L[p])), p


def test_device_work_list_equals_host_plan(ctx):
    """the preparation kernels (spx_prep_kernels.hip) against the host plan: HiFi, ONT, clips + shuffled records +
    inverted paralogs + ambiguous bases, MD-only records, mixed lengths, all-rows lists, odd parameters"""
    import copy
    cases = [
        (small_genome(synth.HIFI), 96, records.preset("hifi")),
        (small_genome(synth.ONT, n_paralogs=3), 24, records.preset("ont", bandwidth=50)),
        (small_genome(synth.HIFI, hardclip_frac=0.5, softclip_frac=0.5, shuffle_records=1, inverted_paralogs=1, n_paralogs=3,
                      max_secondaries=4, n_base_frac=0.002, read_len=6000), 64, records.preset("hifi")),
        (small_genome(synth.HIFI, tag_mode=1, read_len=6000, max_secondaries=3, n_paralogs=2, hardclip_frac=0.3, softclip_frac=0.3),
         48, records.preset("hifi")),
        (small_genome(synth.MIXED, n_paralogs=7, contig_len=250000, max_read_len=40000), 24, records.preset("hifi")),
    ]
    odd = records.preset("hifi")
    odd.conf_d, odd.conf_e, odd.conf_b, odd.set_q, odd.min_q, odd.indel_threshold = 3e-3, 0.25, 33.0, 27, 5, 4
    odd.flank_margin = 300
    cases.append((small_genome(synth.HIFI, read_len=6000, max_secondaries=2, hardclip_frac=0.2, softclip_frac=0.2), 24, odd))
    noc = records.preset("hifi")
    noc.consensus = 0
    cases.append((small_genome(synth.HIFI, read_len=3000, max_secondaries=2), 12, noc))
    for g, n, par in cases:
        r = g.reads(0, n)
        ctx.set_reference(g.ref)
        for flags in (0, 1):
            p2 = copy.copy(par)
            p2.flags = flags
            w = ctx.prepare(r.batch, p2)
            dev = w.export_plan()
            host = api.Plan(g.ref, r.batch, p2)
            _plans_equal(dev, host, n)
            dev.close()
            w.free()



def test_md_tagged_records_on_a_fresh_context(built):
    """Found by tools/fuzz.py gpu (seed 20261003): the op pool of the preparation kernels is sized from the tag lengths
    before any payload byte is read -- one op per two tag characters, which holds for cs but not for MD ("10A5C3": a token
    per character).  On a context whose pools have not grown yet the last alignments of an MD-tagged batch overflowed it;
    the overflow was reported as 'group scratch too small', only the group passes were repeated, and the groups of the
    unbuilt alignments came back unscored (all scores 0).  Now the device reports pool overflows separately and the host
    repeats the per-alignment phase with pools of the exact size.  Fresh context: the module-wide one has large pools."""
    import copy
    kw = dict(seed=831971708757, n_contigs=2, contig_len=60000, n_paralogs=5, softclip_frac=0.0, hardclip_frac=1.0, shuffle_records=1,
              inverted_paralogs=1, n_base_frac=0.001, snv_rate=0.0002, indel_rate=2e-05, paralog_snv_rate=0.04, tag_mode=1, read_len=1000,
              max_secondaries=5, min_secondaries=5)
    par = records.preset("hifi")
    par.prim_margin_score, par.conf_d, par.conf_e, par.conf_b, par.flank_margin = 40.0, 1e-4, 0.1, 20.0, 500
    g = synth.Genome(synth.default_cfg(synth.MIXED, **kw))
    for first, n in ((2364, 26), (2389, 1), (2364, 40)):
        c = api.Context(0)
        try:
            c.set_reference(g.ref)
            r = g.reads(first, n)
            _, res = orc.run_batch(r.batch, g.ref, par, threads=2, seed=1)
            for flags in (0, 1):
                p2 = copy.copy(par)
                p2.flags = flags
                w = c.prepare(r.batch, p2)
                dev = w.export_plan()
                host = api.Plan(g.ref, r.batch, p2)
                _plans_equal(dev, host, n)
                dev.close()
                w.launch()
                out = w.collect(finalize_seed=1)
                for k in range(n):
                    assert out[k].n_aln == res[k].n_aln
                    assert [out[k].score[a] for a in range(max(res[k].n_aln, 0))] == [res[k].score[a] for a in range(max(res[k].n_aln, 0))], (first, n, k)
                w.free()
        finally:
            c.close()



def test_preparation_table_bounds_and_their_fallback(built, monkeypatch):
    """Phase 1 of the preparation sizes the op / block / mismatch tables from the record lengths (spxl::aln_caps) and
    parses every tag once.  SPX_PREP_TIGHT shrinks the bounds so that alignments outgrow them: the device must report it
    and the host must fall back to the exact counting pass; SPX_PREP_EXACT takes that path from the start.  All three
    give the host plan's work list, cs and MD tags, clips, long and short reads."""
    import copy
    cases = [
        (small_genome(synth.HIFI, read_len=6000, max_secondaries=3, hardclip_frac=0.3, softclip_frac=0.3), 32, records.preset("hifi")),
        (small_genome(synth.HIFI, tag_mode=1, read_len=3000, max_secondaries=3, n_paralogs=3, paralog_snv_rate=0.04), 32, records.preset("hifi")),
        (small_genome(synth.ONT, n_paralogs=3), 12, records.preset("ont", bandwidth=50)),
    ]
    for env in ({}, {"SPX_PREP_TIGHT": "1"}, {"SPX_PREP_EXACT": "1"}):
        for k in ("SPX_PREP_TIGHT", "SPX_PREP_EXACT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = api.Context(0)
        try:
            for g, n, par in cases:
                r = g.reads(0, n)
                c.set_reference(g.ref)
                w = c.prepare(r.batch, par)
                dev = w.export_plan()
                host = api.Plan(g.ref, r.batch, par)
                _plans_equal(dev, host, n)
                dev.close()
                w.free()
        finally:
            c.close()



def test_trim_between_batches(built):
    """spx_trim hands the cached arenas and pools back to the driver; the next work list allocates afresh and gives the
    same result"""
    g = small_genome(synth.HIFI, read_len=4000, max_secondaries=2)
    par = records.preset("hifi")
    r = g.reads(0, 48)
    c = api.Context(0)
    try:
        c.set_reference(g.ref)
        out1, _ = c.score_batch(r.batch, par, finalize_seed=1)
        s1 = [[out1[k].score[a] for a in range(max(out1[k].n_aln, 0))] for k in range(48)]
        api._chk(api.lib().spx_trim(c.h), "spx_trim")
        out2, _ = c.score_batch(r.batch, par, finalize_seed=1)
        s2 = [[out2[k].score[a] for a in range(max(out2[k].n_aln, 0))] for k in range(48)]
        assert s1 == s2
        _, res = orc.run_batch(r.batch, g.ref, par, threads=2, seed=1)
        assert s2 == [[res[k].score[a] for a in range(max(res[k].n_aln, 0))] for k in range(48)]
    finally:
        c.close()


def _batch_parity(ctx, genome, reads, params, tmp_path, tag):
    ctx.set_reference(genome.ref)
    out, st = ctx.score_batch(reads.batch, params, finalize_seed=1)
    log_o = str(tmp_path / f"{tag}.oracle.log")
    log_g = str(tmp_path / f"{tag}.gpu.log")
    nre, res = orc.run_batch(reads.batch, genome.ref, params, threads=2, seed=1, log_path=log_o)
    api.write_relabel_log(log_g, reads.batch, genome.ref, out)
    n = reads.batch.contents.n_groups
    assert st.n_problems == sum(r.n_baq_calls for r in res)
    assert st.dp_cells == sum(r.dp_cells for r in res)
    for i in range(n):
        o, e = out[i], res[i]
        assert o.n_aln == e.n_aln, i
        for a in range(max(e.n_aln, 0)):
            assert o.score[a] == e.score[a], (i, a, o.score[a], e.score[a])
        assert o.best_idx == e.best_idx, i
        assert bool(o.relabel) == bool(e.relabel), i
    assert filecmp.cmp(log_o, log_g, shallow=False)
    assert sum(o.relabel for o in out) == nre
    return st


def test_batch_hifi(ctx, tmp_path):
    g = small_genome(synth.HIFI)
    r = g.reads(0, 96)
    _batch_parity(ctx, g, r, records.preset("hifi"), tmp_path, "hifi")


def test_batch_ont(ctx, tmp_path):
    g = small_genome(synth.ONT, n_paralogs=3)
    r = g.reads(0, 24)
    _batch_parity(ctx, g, r, records.preset("ont", bandwidth=50), tmp_path, "ont")


def test_batch_edge_cases(ctx, tmp_path):
    g = small_genome(synth.HIFI, hardclip_frac=0.5, softclip_frac=0.5, shuffle_records=1, inverted_paralogs=1,
                     n_paralogs=3, max_secondaries=4, n_base_frac=0.002, read_len=6000)
    r = g.reads(100, 64)
    _batch_parity(ctx, g, r, records.preset("hifi"), tmp_path, "edge")


def test_batch_md_only_records(ctx, tmp_path):
    g = small_genome(synth.HIFI, tag_mode=1, read_len=6000, max_secondaries=3, n_paralogs=2, hardclip_frac=0.3,
                     softclip_frac=0.3)
    r = g.reads(0, 48)
    _batch_parity(ctx, g, r, records.preset("hifi"), tmp_path, "md")


def test_batch_option_combinations(ctx, tmp_path):
    """-q without -c (BAQ over whole confident blocks: thousands of rows per problem), -c without -q, neither, and
    non-preset gap / band / quality parameters (src/secphase.c:420-449)"""
    g = small_genome(synth.HIFI, read_len=6000, max_secondaries=2, hardclip_frac=0.2, softclip_frac=0.2)
    r = g.reads(0, 24)
    p = records.preset("hifi")
    p.consensus = 0
    st = _batch_parity(ctx, g, r, p, tmp_path, "noc")
    assert st.n_rows > 0 and st.dp_cells // max(st.n_problems, 1) > 41 * 1500
    p = records.preset("hifi")
    p.baq_flag = 0
    st = _batch_parity(ctx, g, r, p, tmp_path, "noq")
    assert st.n_problems == 0
    p.consensus = 0
    _batch_parity(ctx, g, r, p, tmp_path, "none")
    p = records.preset("hifi")
    p.conf_d, p.conf_e, p.conf_b, p.set_q, p.min_q, p.indel_threshold = 3e-3, 0.25, 33.0, 27, 5, 4
    p.prim_margin_score, p.prim_margin_random, p.min_score, p.flank_margin = 5.0, 3.0, -40, 300
    _batch_parity(ctx, g, r, p, tmp_path, "odd")


def test_no_cpu_fallback():
    assert api.lib().spx_device_count() >= 1


def test_command_line_drop_in(ctx, tmp_path):
    """secphase --hifi -i X.bam -f asm.fa -o DIR -P P : relabel list and both marker-mode BEDs byte-identical to
    the oracle's, all six output files present, several GPU batches sharing one rand() stream."""
    import subprocess
    from bamio import write_bam, write_fasta
    g = small_genome(synth.HIFI, max_secondaries=4, n_paralogs=3, hardclip_frac=0.2, softclip_frac=0.3)
    r = g.reads(0, 150)
    fa, bam, outd = str(tmp_path / "asm.fa"), str(tmp_path / "reads.bam"), str(tmp_path / "out")
    write_fasta(fa, g.ref)
    write_bam(bam, r.batch, g.ref)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "secphase_amd", "bin", "secphase")
    p = subprocess.run([exe, "--hifi", "-@", "4", "-i", bam, "-f", fa, "--outDir", outd, "--prefix", "t",
                        "--groupsPerBatch", "64"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    log_o, bm_o, bk_o = (str(tmp_path / n) for n in ("o.log", "o.mod.bed", "o.mk.bed"))
    nre, res = orc.run_batch(r.batch, g.ref, records.preset("hifi"), threads=2, seed=1, log_path=log_o,
                             bed_modified=bm_o, bed_markers=bk_o)
    assert nre > 5
    assert filecmp.cmp(log_o, os.path.join(outd, "t.out.log"), shallow=False)
    assert filecmp.cmp(bm_o, os.path.join(outd, "t.modified_read_blocks.markers.bed"), shallow=False)
    assert filecmp.cmp(bk_o, os.path.join(outd, "t.marker_blocks.bed"), shallow=False)
    for sfx in ("initial_variant_blocks.bed", "modified_read_blocks.variants.bed", "variant_blocks.bed"):
        assert os.path.getsize(os.path.join(outd, "t." + sfx)) == 0
    assert f"Number of reads modified by marker score = {nre}" in p.stderr
    # row N2: the list the HIP path wrote, consumed the way correct_bam consumes it (correct_bam.c:32-91), and applied to the BAM
    # (flag swap correct_bam.c:352-358): the table names, for every relabelled read, the promoted secondary's contig + 0-based start
    import subprocess as sp
    from test_correct_bam import EXE as CORRECT_BAM, expected_records, load_table, parse_list_like_correct_bam
    from bamio import read_bam
    cli_log = os.path.join(outd, "t.out.log")
    table = load_table(cli_log)
    b, rf = r.batch.contents, g.ref.contents
    expect = {}
    for i, e in enumerate(res):
        if e.relabel:
            a = b.grp_first[i] + e.best_idx
            expect[C.string_at(b.qnames + b.qname_off[i]).decode()] = (C.string_at(rf.names + rf.name_off[b.tid[a]]).decode(), b.pos[a])
    assert table == expect == parse_list_like_correct_bam(cli_log) and len(table) == nre
    fixed = str(tmp_path / "corrected.bam")
    q = sp.run([CORRECT_BAM, "-i", bam, "-o", fixed, "-P", cli_log, "-m", "1000", "-a", "500"], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stderr
    got = read_bam(fixed)[2]
    assert [x["raw"] for x in got] == expected_records(bam, table, min_read=1000, min_aln=500)
    prim = {}
    for x in got:
        if not x["flag"] & 256:
            prim.setdefault(x["name"], []).append((read_bam(fixed)[1][x["tid"]][0], x["pos"]))
    assert all(prim.get(n) == [loc] for n, loc in table.items())


def test_aliased_and_altered_secondaries(ctx, tmp_path):
    """staging leaves out SEQ / QUAL of secondaries that repeat the primary's and the device rebuilds them (both strands,
    hard-clipped records): scores and the list equal the oracle's; with the qualities / bases of some secondaries ALTERED
    those records are transferred as they are -- and the oracle, which reads every record's own bytes, still agrees"""
    g = small_genome(synth.HIFI, read_len=4000, max_secondaries=3, min_secondaries=1, n_paralogs=3, hardclip_frac=0.5, softclip_frac=0.5,
                     inverted_paralogs=1, shuffle_records=1)
    r = g.reads(0, 150)
    b = r.batch.contents
    p = records.preset("hifi")
    ctx.set_reference(g.ref)
    _batch_parity(ctx, g, r, p, tmp_path, "alias")
    rng = np.random.default_rng(9)
    changed = 0
    for a in range(b.n_alns):
        if (b.flag[a] & 256) and rng.random() < 0.4:
            lq = b.l_qseq[a]
            lo = int(rng.integers(0, max(1, lq - 400)))
            for k in range(lo, min(lq, lo + 400)):   # a stretch of low qualities: markers there fall under min_q
                b.qual[b.qual_off[a] + k] = 3
            b.seq4[b.seq_off[a] + lo // 2] ^= 0x33
            changed += 1
    assert changed > 10
    _batch_parity(ctx, g, r, p, tmp_path, "alias")


def test_command_line_several_devices(ctx, tmp_path):
    """secphase --devices 0,0,0 : three scoring contexts (here on one GPU), batches dealt round-robin, results taken in file
    order by ONE finalizer -- the relabel list (tie groups included: their rand() draws are replayed in file order), both
    BEDs and the counters are byte-identical to the oracle's single stream (src/secphase.c:194-217 at -@1)."""
    import subprocess
    g = small_genome(synth.HIFI, max_secondaries=4, n_paralogs=3, read_len=4000, min_secondaries=0, paralog_snv_rate=0.0002)
    chunks = [g.reads(i * 50, 50) for i in range(6)]
    whole = g.reads(0, 300)
    fa, bam = str(tmp_path / "asm.fa"), str(tmp_path / "reads.bam")
    synth.write_fasta(fa, g.ref)
    synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=2)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "secphase_amd", "bin", "secphase")
    par = records.preset("hifi")
    par.prim_margin_score = 5.0
    log_o, bm_o, bk_o = (str(tmp_path / n) for n in ("o.log", "o.mod.bed", "o.mk.bed"))
    nre, res = orc.run_batch(whole.batch, g.ref, par, threads=2, seed=1, log_path=log_o, bed_modified=bm_o, bed_markers=bk_o)
    ties = 0
    for e in res:
        if e.n_aln >= 2:
            sec = [a for a in range(e.n_aln) if a != e.prim_idx]
            mxs = max(e.score[a] for a in sec)
            ties += sum(1 for a in sec if e.score[a] >= mxs) > 1
    assert ties > 0 and nre > 5
    # (eight contexts and eight input pipelines on the one GPU of the box, segments of 64 KB: groups -- tie groups among them --
    # straddle every cut between pipelines; what `--devices 0-7` does on an 8-GPU node, minus the peers)
    for devs, batch, extra, env in (("0,0,0", "17", [], {}), ("0-0", "40", [], {}), ("0,0", "1000", [], {"SPX_DIN_SEG_KB": "128"}),
                                    ("0,0,0", "17", ["--hostInput"], {}), ("0,0", "1000", ["--hostInput"], {}),
                                    ("0,0,0,0,0,0,0,0", "9", [], {"SPX_DIN_SEG_KB": "64"}), ("0,0,0,0,0,0,0,0", "1000", [], {"SPX_DIN_SEG_KB": "64"}),
                                    ("0,0,0,0,0,0,0,0", "11", ["--hostInput"], {})):
        outd = str(tmp_path / f"out_{batch}_{len(devs)}_{len(extra)}_{len(env)}")
        p = subprocess.run([exe, "--hifi", "-p", "5", "-@", "4", "-i", bam, "-f", fa, "--outDir", outd, "--prefix", "t", "--devices", devs,
                            "--groupsPerBatch", batch] + extra, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr
        assert filecmp.cmp(log_o, os.path.join(outd, "t.out.log"), shallow=False), devs
        assert filecmp.cmp(bm_o, os.path.join(outd, "t.modified_read_blocks.markers.bed"), shallow=False)
        assert filecmp.cmp(bk_o, os.path.join(outd, "t.marker_blocks.bed"), shallow=False)
        assert f"Number of reads modified by marker score = {nre}" in p.stderr
    p = subprocess.run([exe, "--hifi", "-i", bam, "-f", fa, "--outDir", str(tmp_path / "bad"), "--devices", "0,99"], capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "device" in p.stderr


def test_command_line_with_device_inflate(ctx, tmp_path):
    """every chunk of the BAM inflated by bgzf_inflate_kernel (the reader is held back until the device workers are
    attached, 64 KB chunks): outputs identical to the oracle's; a flipped byte in the file ends the run with an error,
    on the device path as on the host path"""
    import subprocess
    g = small_genome(synth.HIFI, max_secondaries=3, n_paralogs=3, read_len=6000)
    chunks = [g.reads(i * 40, 40) for i in range(5)]
    whole = g.reads(0, 200)
    fa, bam = str(tmp_path / "asm.fa"), str(tmp_path / "reads.bam")
    synth.write_fasta(fa, g.ref)
    synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=2)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "secphase_amd", "bin", "secphase")
    log_o, bm_o, bk_o = (str(tmp_path / n) for n in ("o.log", "o.mod.bed", "o.mk.bed"))
    nre, _ = orc.run_batch(whole.batch, g.ref, records.preset("hifi"), threads=2, seed=1, log_path=log_o, bed_modified=bm_o, bed_markers=bk_o)
    assert nre > 3
    env = dict(os.environ, SPX_GPU_INFLATE_FIRST="1", SPX_BAM_DEVICE_ALL="1", SPX_BAM_CHUNK_KB="64", SPX_TIMING="1")
    outd = str(tmp_path / "out")
    p = subprocess.run([exe, "--hifi", "-@", "4", "-i", bam, "-f", fa, "--outDir", outd, "--prefix", "t", "--groupsPerBatch", "37",
                        "--gpuInflate", "3", "--hostInput"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr
    import re
    m = re.search(r"inflate chunks: (\d+) on the host pool, (\d+) on the device", p.stderr)
    assert m and int(m.group(2)) > 10 and int(m.group(1)) == 0, p.stderr[-400:]
    assert filecmp.cmp(log_o, os.path.join(outd, "t.out.log"), shallow=False)
    assert filecmp.cmp(bm_o, os.path.join(outd, "t.modified_read_blocks.markers.bed"), shallow=False)
    assert filecmp.cmp(bk_o, os.path.join(outd, "t.marker_blocks.bed"), shallow=False)
    blob = bytearray(open(bam, "rb").read())
    blob[len(blob) // 2] ^= 0x20
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(blob))
    for e, extra in ((env, ["--hostInput"]), (dict(os.environ), ["--hostInput"]), (dict(os.environ), [])):
        p = subprocess.run([exe, "--hifi", "-@", "4", "-i", bad, "-f", fa, "--outDir", str(tmp_path / "outbad"), "--prefix", "t",
                            "--groupsPerBatch", "37"] + extra, capture_output=True, text=True, timeout=600, env=e)
        assert p.returncode != 0 and "BAM read error" in p.stderr, p.stderr[-300:]


def _cli_parity(tmp_path, g, n_groups, chunk, flags, par, tag, env=None, min_relabelled=1):
    """the command-line drop-in on a BAM of n_groups synthetic groups (written by the C writer, htslib block policy) with
    `flags`, against the oracle with the matching parameters: out.log and both BEDs byte for byte, the counters"""
    import subprocess
    chunks = [g.reads(k, min(chunk, n_groups - k)) for k in range(0, n_groups, chunk)]
    whole = g.reads(0, n_groups)
    fa, bam, outd = str(tmp_path / f"{tag}.fa"), str(tmp_path / f"{tag}.bam"), str(tmp_path / f"{tag}_out")
    synth.write_fasta(fa, g.ref)
    synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=4)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "secphase_amd", "bin", "secphase")
    log_o, bm_o, bk_o = (str(tmp_path / f"{tag}.{n}") for n in ("o.log", "o.mod.bed", "o.mk.bed"))
    nre, _ = orc.run_batch(whole.batch, g.ref, par, threads=8, seed=1, log_path=log_o, bed_modified=bm_o, bed_markers=bk_o)
    assert nre >= min_relabelled, nre
    # both input sides: device-resident (the default: inflate, record chain, fields, dispatch filter, staging as kernels) and the host reader
    for extra in ([], ["--hostInput"]):
        import shutil
        shutil.rmtree(outd, ignore_errors=True)
        p = subprocess.run([exe] + flags + extra + ["-@", "8", "-i", bam, "-f", fa, "--outDir", outd, "--prefix", "t"], capture_output=True, text=True,
                           timeout=1200, env=dict(os.environ, **(env or {})))
        assert p.returncode == 0, (extra, p.stderr[-600:])
        assert filecmp.cmp(log_o, os.path.join(outd, "t.out.log"), shallow=False), (tag, extra)
        assert filecmp.cmp(bm_o, os.path.join(outd, "t.modified_read_blocks.markers.bed"), shallow=False), (tag, extra)
        assert filecmp.cmp(bk_o, os.path.join(outd, "t.marker_blocks.bed"), shallow=False), (tag, extra)
        assert f"Number of reads modified by marker score = {nre}" in p.stderr
    return p


def test_command_line_ont_preset_with_band_50(ctx, tmp_path):
    """BASELINE config 3 through the command line: `secphase --ont -b 50` (src/secphase.c:491-504 + -b) on ONT-shaped groups
    (30 kb reads, <= 4 secondaries): the wide band classes reached from a BAM file, not only through the library API"""
    g = small_genome(synth.ONT, n_paralogs=3, contig_len=400000)
    _cli_parity(tmp_path, g, 96, 32, ["--ont", "-b", "50"], records.preset("ont", bandwidth=50), "ont50")
    # the preset alone (band 20) is a different parameter set and a different list
    _cli_parity(tmp_path, g, 48, 48, ["--ont"], records.preset("ont"), "ont20")


def test_command_line_mixed_workload_at_the_default_batch_size(ctx, tmp_path):
    """BASELINE config 5's mix (HiFi + ONT, power-law lengths 2-100 kb, <= 8 secondaries) run as --hifi with the DEFAULT
    --groupsPerBatch (nothing passed): records of 100 kb reads span several BGZF blocks"""
    cfg = synth.default_cfg(synth.MIXED, n_contigs=2, contig_len=600000)
    g = synth.Genome(cfg)
    _cli_parity(tmp_path, g, 400, 100, ["--hifi"], records.preset("hifi"), "mixed")


def test_command_line_non_preset_scoring_flags(ctx, tmp_path):
    """every scoring flag given by hand, none of them a preset value (src/secphase.c:506-560): -q -c -d -e -b -t -s -m -p -r -n
    --flankMargin"""
    g = small_genome(synth.HIFI, read_len=6000, max_secondaries=3, n_paralogs=3, hardclip_frac=0.2, softclip_frac=0.2, paralog_snv_rate=0.002)
    p = records.preset("hifi")
    p.conf_d, p.conf_e, p.conf_b, p.set_q, p.min_q, p.indel_threshold = 3e-3, 0.25, 33.0, 27, 5, 4
    p.prim_margin_score, p.prim_margin_random, p.min_score, p.flank_margin = 5.0, 3.0, -40, 300
    flags = ["-q", "-c", "-d", "3e-3", "-e", "0.25", "-b", "33", "-t", "4", "-s", "27", "-m", "5", "-p", "5", "-r", "3", "-n", "-40",
             "--flankMargin", "300", "--groupsPerBatch", "53"]
    _cli_parity(tmp_path, g, 200, 50, flags, p, "odd")
    # -q without -c, and neither: the defaults of src/secphase.c:420-449 otherwise
    p2 = records.preset("hifi")
    p2.consensus = 0
    _cli_parity(tmp_path, g, 60, 60, ["-q", "--groupsPerBatch", "31"], p2, "noc", min_relabelled=0)
    p3 = records.preset("hifi")
    p3.consensus, p3.baq_flag = 0, 0
    _cli_parity(tmp_path, g, 60, 60, [], p3, "none", min_relabelled=0)


def _quals_parity(ctx, genome, reads, params):
    """all-rows work list (-w/--writeBam): record qualities after BAQ equal the oracle's, scores unchanged"""
    import copy
    from common import batch_qual_copy
    ctx.set_reference(genome.ref)
    p_all = copy.copy(params)
    p_all.flags = 1
    want, res = orc.run_batch_quals(reads.batch, genome.ref, params, batch_qual_copy(reads.batch), threads=2)
    w = ctx.prepare(reads.batch, p_all)
    w.launch()
    out = w.collect(finalize_seed=1)
    got = w.apply_quals(reads.batch, batch_qual_copy(reads.batch))
    st = w.stats()
    w.free()
    assert np.array_equal(got, want)
    assert not np.array_equal(want, batch_qual_copy(reads.batch))
    for i in range(reads.batch.contents.n_groups):
        for a in range(max(res[i].n_aln, 0)):
            assert out[i].score[a] == res[i].score[a], (i, a)
    return st


def test_quality_modified_records_hifi(ctx):
    g = small_genome(synth.HIFI, hardclip_frac=0.3, softclip_frac=0.4, max_secondaries=3, n_paralogs=2, n_base_frac=0.001)
    st = _quals_parity(ctx, g, g.reads(0, 48), records.preset("hifi"))
    assert st.n_rows > 20 * st.n_problems  # every base of the windows, not only markers


def test_quality_modified_records_ont(ctx):
    g = small_genome(synth.ONT, n_paralogs=3)
    _quals_parity(ctx, g, g.reads(0, 12), records.preset("ont", bandwidth=50))


def test_quality_modified_records_in_a_merged_work_list(ctx):
    """spx_prepare_many over two record blocks: each block's qualities come back through its own batch index"""
    import copy
    from common import batch_qual_copy
    g = small_genome(synth.HIFI, hardclip_frac=0.3, softclip_frac=0.4, max_secondaries=3, n_paralogs=2)
    ra, rb = g.reads(0, 20), g.reads(20, 33)
    par = records.preset("hifi")
    p_all = copy.copy(par)
    p_all.flags = 1
    ctx.set_reference(g.ref)
    w = ctx.prepare([ra.batch, rb.batch], p_all)
    w.launch()
    w.collect(finalize_seed=None)
    for k, r in enumerate((ra, rb)):
        want, _ = orc.run_batch_quals(r.batch, g.ref, par, batch_qual_copy(r.batch), threads=2)
        got = w.apply_quals(r.batch, batch_qual_copy(r.batch), batch_index=k)
        assert np.array_equal(got, want), k
    w.free()


def test_command_line_write_bam(ctx, tmp_path):
    """secphase --hifi -w: <prefix>.quality_modified.out.bam is what the reference writes -- SAM text (sam_open "w",
    src/secphase.c:643-652) with the header and, for every dispatched group in file order (= -@1), all records with
    the qualities calc_local_baq left (src/secphase.c:182-189); the other outputs do not change."""
    import subprocess
    from bamio import sam_text, write_bam, write_fasta
    from common import batch_qual_copy
    g = small_genome(synth.HIFI, max_secondaries=4, n_paralogs=3, hardclip_frac=0.2, softclip_frac=0.3, read_len=6000)
    r = g.reads(0, 90)
    fa, bam, outd = str(tmp_path / "asm.fa"), str(tmp_path / "reads.bam"), str(tmp_path / "out")
    write_fasta(fa, g.ref)
    write_bam(bam, r.batch, g.ref)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "secphase_amd", "bin", "secphase")
    p = subprocess.run([exe, "--hifi", "-w", "-@", "4", "-i", bam, "-f", fa, "--outDir", outd, "--prefix", "t",
                        "--groupsPerBatch", "32"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    par = records.preset("hifi")
    log_o = str(tmp_path / "o.log")
    nre, _ = orc.run_batch(r.batch, g.ref, par, threads=2, seed=1, log_path=log_o)
    assert filecmp.cmp(log_o, os.path.join(outd, "t.out.log"), shallow=False)
    want_q, _ = orc.run_batch_quals(r.batch, g.ref, par, batch_qual_copy(r.batch), threads=2)
    disp = [k for k in range(r.batch.contents.n_groups) if orc.lib().orc_group_is_dispatched(r.batch, k)]
    assert 0 < len(disp) < r.batch.contents.n_groups or len(disp) == r.batch.contents.n_groups
    want = sam_text(r.batch, g.ref, qual=want_q, groups=disp)
    assert open(os.path.join(outd, "t.quality_modified.out.bam")).read() == want


def _full_size_properties(ctx, platform, par, n, chunk, n_shards, min_problems, min_cells, sample):
    """A workload of bench.py's shape (a quarter of its default step): the oracle would need minutes for all of it, so the whole batch is checked
    through size-independent properties -- the same groups scored as one work list, as shards, in another order and
    several times in a row give bit-identical scores and decisions -- and a random sample of groups is compared with
    the oracle."""
    import threading
    cfg = synth.default_cfg(platform)
    g = synth.Genome(cfg)
    parts = [None] * (n // chunk)

    def gen(k0):
        for k in range(k0, len(parts), 8):
            parts[k] = g.reads(k * chunk, chunk)
    th = [threading.Thread(target=gen, args=(k,)) for k in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    ctx.set_reference(g.ref)

    def score(batches, relaunch=False):
        w = ctx.prepare(batches, par)
        w.launch()
        if relaunch:
            for _ in range(4):
                w.launch()
        out = w.collect(finalize_seed=None)
        st = w.stats()
        res = [(o.n_aln, tuple(o.score[a] for a in range(max(o.n_aln, 0))), o.prim_idx, o.max_idx, o.tie_mask, o.pass_)
               for o in out]
        w.free()
        return res, st

    whole, st = score([p.batch for p in parts])
    assert st.n_dispatched == n and st.n_problems > min_problems and st.dp_cells > min_cells
    # shards (what rank r of a multi-GPU run would score) -- union equals the whole
    per = len(parts) // n_shards
    sharded = []
    for r in range(n_shards):
        res, _ = score([p.batch for p in parts[per * r:per * r + per]])
        sharded += res
    assert sharded == whole
    # another order of the same groups: results follow the groups
    perm = list(range(len(parts)))[::-1]
    res, _ = score([parts[k].batch for k in perm])
    back = [None] * len(parts)
    for pos, k in enumerate(perm):
        back[k] = res[pos * chunk:(pos + 1) * chunk]
    assert [x for blk in back for x in blk] == whole
    # idempotence: launching the same work list five times back to back leaves the same results
    again, _ = score([p.batch for p in parts], relaunch=True)
    assert again == whole
    # a sample against the oracle
    rng = np.random.default_rng(3)
    for k in rng.choice(n, size=sample, replace=False):
        sub = g.reads(int(k), 1)
        _, ores = orc.run_batch(sub.batch, g.ref, par, threads=1, seed=1)
        e, o = ores[0], whole[int(k)]
        assert o[0] == e.n_aln and o[1] == tuple(e.score[a] for a in range(e.n_aln)) and o[2] == e.prim_idx, int(k)
    return st


def test_work_list_of_131072_groups_in_a_pipeline_of_depth_3(ctx):
    """The size bench.py's default step has: ONE work list of 131 072 HiFi groups (config 2, the 100 Mbp assembly), fed from
    host memory through the in-order pipeline at depth 3 with two 65 536-group lists (its two halves) in flight beside it --
    three preparations on their lanes at once, ~90 GB of work lists.  The halves must reproduce the whole bit for bit, the
    whole must be in input order, and 48 random groups must equal the oracle's scores."""
    import threading
    n, chunk = 131072, 1024
    g = synth.Genome(synth.default_cfg(synth.HIFI))
    parts = [None] * (n // chunk)

    def gen(k0):
        for k in range(k0, len(parts), 8):
            parts[k] = g.reads(k * chunk, chunk)
    th = [threading.Thread(target=gen, args=(k,)) for k in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    ctx.set_reference(g.ref)
    par = records.preset("hifi")
    ptr = [p.batch for p in parts]
    half = len(ptr) // 2
    pipe = api.Pipe(ctx, par, depth=3, host_threads=0)
    try:
        for sub in (ptr, ptr[:half], ptr[half:], ptr):
            pipe.submit(batch=sub)
        key = lambda o: (o.n_aln, tuple(o.score[a] for a in range(max(o.n_aln, 0))), o.prim_idx, o.max_idx, o.tie_mask, o.pass_, o.n_problems, o.dp_cells)
        got = []
        for want in (n, n // 2, n // 2, n):
            out, m = pipe.next()
            assert m == want
            got.append([key(out[k]) for k in range(m)])
    finally:
        pipe.close()
    whole = got[0]
    assert got[1] + got[2] == whole and got[3] == whole
    assert sum(1 for x in whole if x[0] >= 2) == n and sum(x[6] for x in whole) > 8000000 and sum(x[7] for x in whole) > 8 * 10 ** 10
    rng = np.random.default_rng(5)
    for k in rng.choice(n, size=48, replace=False):
        sub = g.reads(int(k), 1)
        _, ores = orc.run_batch(sub.batch, g.ref, par, threads=1, seed=1)
        e, o = ores[0], whole[int(k)]
        assert o[0] == e.n_aln and o[1] == tuple(e.score[a] for a in range(e.n_aln)) and o[2] == e.prim_idx and o[7] == e.dp_cells, int(k)


def test_dp_slices_share_one_scratch_area_and_change_nothing(ctx, tmp_path, monkeypatch):
    """Work-list scratch per slice (1/s of every DP row + the saved forward rows are the bulk of a list's memory): forward ->
    backward -> MAP run over K ranges of consecutive groups that re-use ONE scratch area.  Whatever K, every result is
    bit-identical to the unsliced list's (markers whose rows fall into different slices included), equals the oracle's, and
    the all-rows lists of -w give the same qualities; the scratch shrinks with K."""
    cases = [(small_genome(synth.HIFI, read_len=6000, max_secondaries=3, n_paralogs=3, hardclip_frac=0.2, softclip_frac=0.2), 1500, records.preset("hifi")),
             (small_genome(synth.ONT, n_paralogs=3, contig_len=300000), 600, records.preset("ont", bandwidth=50)),
             (small_genome(synth.MIXED, n_paralogs=7, contig_len=250000, max_read_len=40000), 700, records.preset("hifi"))]
    key = lambda o: (o.n_aln, tuple(o.score[a] for a in range(max(o.n_aln, 0))), o.prim_idx, o.max_idx, o.tie_mask, o.pass_, o.n_problems, o.n_markers, o.dp_cells)
    for g, n, par in cases:
        r = g.reads(0, n)
        ctx.set_reference(g.ref)
        res = {}
        for K in (1, 2, 5):
            monkeypatch.setenv("SPX_DP_SLICES", str(K))
            w = ctx.stage(r.batch, par)
            w.prepare_staged()
            w.launch()
            w.launch()  # a second launch over the same scratch
            out = w.collect(finalize_seed=None)
            st = w.stats()
            res[K] = ([key(out[k]) for k in range(n)], st.main_fwd_ms, st.n_problems, w.device_bytes())
            w.free()
        assert res[2][0] == res[1][0] and res[5][0] == res[1][0]
        assert res[5][1] > 0 and res[5][2] == res[1][2]
        k5 = min(5, max(1, n // 256))  # (a slice holds at least 256 groups)
        assert res[1][3][1] == 1 and res[2][3][1] == 2 and res[5][3][1] == k5 and res[5][3][0] <= res[2][3][0] < res[1][3][0]
        if res_first_case_holder[0] is None:
            res_first_case_holder[0] = res[1][0]
        _, ores = orc.run_batch(r.batch, g.ref, par, threads=8, seed=1)
        for k in range(0, n, 7):
            e = ores[k]
            assert res[5][0][k][0] == e.n_aln and res[5][0][k][1] == tuple(e.score[a] for a in range(max(e.n_aln, 0))), k
    # all-rows lists (-w): the qualities written back by sliced lists
    g, n, par = cases[0][0], 300, copy_preset(cases[0][2])
    par.flags |= 1
    r = g.reads(0, n)
    ctx.set_reference(g.ref)
    quals = {}
    for K in (1, 3):
        monkeypatch.setenv("SPX_DP_SLICES", str(K))
        w = ctx.prepare(r.batch, par)
        w.launch()
        w.collect(finalize_seed=None)
        from common import batch_qual_copy
        quals[K] = w.apply_quals(r.batch, batch_qual_copy(r.batch)).copy()
        w.free()
    assert np.array_equal(quals[1], quals[3])
    monkeypatch.delenv("SPX_DP_SLICES")
    # the automatic choice: a budget per slice
    monkeypatch.setenv("SPX_DP_SLICE_GB", "0.05")
    g, n, par = cases[0]
    r = g.reads(0, n)
    ctx.set_reference(g.ref)
    w = ctx.stage(r.batch, par)
    w.prepare_staged()
    assert w.device_bytes()[1] > 1
    w.launch()
    out = w.collect(finalize_seed=None)
    w.free()
    assert [key(out[k]) for k in range(n)] == res_first_case_holder[0]


def copy_preset(p):
    import copy
    return copy.copy(p)


res_first_case_holder = [None]


def test_full_size_workload_properties(ctx):
    """BASELINE config 2 at a quarter of the size bench.py times per step (test_work_list_of_131072_groups... has the full
    size): 32 768 HiFi groups, 15 kb reads, the 100 Mbp assembly"""
    _full_size_properties(ctx, synth.HIFI, records.preset("hifi"), 32768, 1024, 4, 2000000, 2 * 10 ** 10, 48)


def test_full_size_workload_properties_ont(ctx):
    """BASELINE config 3 at a quarter of the size bench.py --platform ont times per step: 4 096 ONT groups, 30 kb reads, <= 4 secondaries,
    band 50 on the 100 Mbp assembly -- the band classes (4,26)/(4,28)/(4,30)/(8,16)+(4,32) on real work lists"""
    st = _full_size_properties(ctx, synth.ONT, records.preset("ont", bandwidth=50), 4096, 256, 4, 1000000, 2 * 10 ** 10, 48)
    wide = sum(st.problems_per_class[c] for c in (6, 7, 12, 13))
    assert wide > 0.9 * st.n_problems  # the ONT widths W = 101..127


def test_batch_mixed_config5(ctx, tmp_path):
    """BASELINE config 5 (load-balance stress): mixed HiFi + ONT reads, power-law lengths 2..100 kb, up to 8
    secondaries, run as --hifi over the whole mix: scores, decisions and out.log byte-identical to the
    oracle's over 256 groups"""
    cfg = synth.default_cfg(synth.MIXED)
    assert cfg.max_secondaries == 8 and cfg.max_read_len == 100000
    g = synth.Genome(cfg)
    r = g.reads(0, 256)
    b = r.batch.contents
    lens = [b.l_qseq[b.grp_first[k]] for k in range(b.n_groups)]
    assert min(lens) < 4000 and max(lens) > 30000
    assert max(b.grp_first[k + 1] - b.grp_first[k] for k in range(b.n_groups)) >= 8
    st = _batch_parity(ctx, g, r, records.preset("hifi"), tmp_path, "mixed")
    assert sum(1 for c in range(16) if st.problems_per_class[c] > 0) >= 4  # HiFi-like and ONT-like band widths side by side


def test_probaln_glocal_symbol_on_device(built):
    """spx_probaln_glocal -- the htslib-signature entry point (ptMarker.c:755-757) -- called directly: state[], q[] AND
    the returned phred-scaled likelihood equal the oracle's probaln_glocal"""
    rng = np.random.default_rng(21)
    L = api.lib()
    u8 = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint8))
    for (n, sub, ind, bw, d, sq) in ((801, 0.003, 0.002, 20, 1e-4, 40), (640, 0.02, 0.04, 50, 1e-3, 20), (57, 0.05, 0.05, 7, 1e-2, 30),
                                     (1, 0.0, 0.0, 3, 1e-4, 40), (300, 0.3, 0.1, 130, 1e-3, 13)):
        ref, qry = _rand_problem(rng, n, ind, sub)
        par = api.ProbalnPar(d, 0.1, abs(len(ref) - len(qry)) + bw)
        iq = np.full(len(qry), sq, np.uint8)
        st = np.zeros(len(qry), np.int32)
        q = np.zeros(len(qry), np.uint8)
        pr = L.spx_probaln_glocal(u8(ref), len(ref), u8(qry), len(qry), u8(iq), C.byref(par),
                                  st.ctypes.data_as(C.POINTER(C.c_int)), u8(q))
        epr, est, eq = oracle_probaln(ref, qry, sq, d, 0.1, par.bw)
        assert pr == epr and pr != -2 ** 31, (n, pr, epr)
        assert np.array_equal(st, est) and np.array_equal(q, eq), n
    # iqual == NULL means Q30 for every base
    ref, qry = _rand_problem(rng, 200, 0.01, 0.01)
    par = api.ProbalnPar(1e-4, 0.1, 25)
    st = np.zeros(len(qry), np.int32)
    q = np.zeros(len(qry), np.uint8)
    pr = L.spx_probaln_glocal(u8(ref), len(ref), u8(qry), len(qry), None, C.byref(par), st.ctypes.data_as(C.POINTER(C.c_int)), u8(q))
    epr, est, eq = oracle_probaln(ref, qry, 30, 1e-4, 0.1, 25)
    assert pr == epr and np.array_equal(st, est) and np.array_equal(q, eq)
    assert L.spx_probaln_glocal(u8(ref), 0, u8(qry), len(qry), None, C.byref(par), st.ctypes.data_as(C.POINTER(C.c_int)), u8(q)) == 0


def test_probaln_glocal_with_per_base_qualities(built):
    """htslib's contract in full: iqual[i] may differ from base to base (samtools' BAQ; secphase passes a constant).  Such
    problems take the general kernel (spx_probaln_general.hip): state[], q[] and the returned likelihood equal the oracle's
    for HiFi- and ONT-shaped windows, bands wider than the sequences, ambiguous bases, R != L, qualities from 0 to 93 -- and
    a constant array still gives what the scoring kernels give"""
    rng = np.random.default_rng(33)
    L = api.lib()
    OL = orc.lib()
    u8 = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint8))
    ip = lambda x: x.ctypes.data_as(C.POINTER(C.c_int))
    cases = [(801, 0.003, 0.002, 20, 1e-4, 0.0), (640, 0.02, 0.04, 50, 1e-3, 0.0), (57, 0.05, 0.05, 7, 1e-2, 0.0), (2, 0.0, 0.0, 3, 1e-4, 0.0),
             (300, 0.3, 0.1, 130, 1e-3, 0.02), (33, 0.05, 0.05, 50, 1e-3, 0.0), (1200, 0.01, 0.02, 35, 1e-3, 0.01)]
    for (n, sub, ind, bw, d, nfrac) in cases:
        ref, qry = _rand_problem(rng, n, ind, sub, n_frac=nfrac)
        par = api.ProbalnPar(d, 0.1, abs(len(ref) - len(qry)) + bw)
        opar = orc.ProbalnPar(d, 0.1, par.bw)
        iq = rng.integers(0, 94, len(qry)).astype(np.uint8)
        if len(qry) > 1 and iq[0] == iq[1]:
            iq[1] = (iq[0] + 7) % 94
        st, q = np.zeros(len(qry), np.int32), np.zeros(len(qry), np.uint8)
        est, eq = np.zeros(len(qry), np.int32), np.zeros(len(qry), np.uint8)
        pr = L.spx_probaln_glocal(u8(ref), len(ref), u8(qry), len(qry), u8(iq), C.byref(par), ip(st), u8(q))
        epr = OL.orc_probaln_glocal(u8(ref), len(ref), u8(qry), len(qry), u8(iq), C.byref(opar), ip(est), u8(eq))
        assert pr == epr and pr != -2 ** 31, (n, pr, epr)
        assert np.array_equal(st, est) and np.array_equal(q, eq), (n, np.flatnonzero(st != est)[:5], np.flatnonzero(q != eq)[:5])
    # the general kernel on a CONSTANT array = the scoring kernels (two implementations, one answer)
    ref, qry = _rand_problem(rng, 500, 0.01, 0.01)
    par = api.ProbalnPar(1e-4, 0.1, 24)
    iq = np.full(len(qry), 37, np.uint8)
    st0, q0 = np.zeros(len(qry), np.int32), np.zeros(len(qry), np.uint8)
    pr0 = L.spx_probaln_glocal(u8(ref), len(ref), u8(qry), len(qry), u8(iq), C.byref(par), ip(st0), u8(q0))
    iq[-1] = 36  # one base differs: the general kernel
    st1, q1 = np.zeros(len(qry), np.int32), np.zeros(len(qry), np.uint8)
    L.spx_probaln_glocal(u8(ref), len(ref), u8(qry), len(qry), u8(iq), C.byref(par), ip(st1), u8(q1))
    est, eq = np.zeros(len(qry), np.int32), np.zeros(len(qry), np.uint8)
    opar = orc.ProbalnPar(1e-4, 0.1, 24)
    OL.orc_probaln_glocal(u8(ref), len(ref), u8(qry), len(qry), u8(iq), C.byref(opar), ip(est), u8(eq))
    assert np.array_equal(st1, est) and np.array_equal(q1, eq) and pr0 != -2 ** 31
    assert np.array_equal(st0[:-40], st1[:-40])  # far from the changed base nothing moves


def test_command_line_stops_on_records_without_tags(ctx, tmp_path):
    """a dispatched group whose records carry neither cs nor MD ends the run like the reference does (cigar_it.c:64-67)"""
    import subprocess
    from bamio import write_bam, write_fasta
    from common import HandBatch, HandRef
    seq = "ACGT" * 10
    hr = HandRef([("c0", seq * 4)])
    hb = HandBatch([("r", [(0, 0, 0, "40M", seq, 30, None), (256, 0, 3, "38M", seq[:38], 30, None)])])
    fa, bam, outd = str(tmp_path / "asm.fa"), str(tmp_path / "reads.bam"), str(tmp_path / "out")
    write_fasta(fa, hr.ref)
    write_bam(bam, hb.batch, hr.ref, extra_tags=False)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "secphase_amd", "bin", "secphase")
    p = subprocess.run([exe, "--hifi", "-i", bam, "-f", fa, "--outDir", outd], capture_output=True, text=True, timeout=300)
    assert p.returncode == 1
    assert "At least one of the MD or CS tags should be present!" in p.stderr


def test_pipeline_in_order_equals_one_shot_scoring(ctx):
    """spx_pipe: batches submitted from host memory and staged work lists (records resident in HBM, prepared again for
    every submission) come back in submission order with the very results spx_score_batch gives; device-packed
    16-byte decision records equal the ones built from the collected results"""
    g = small_genome(synth.HIFI, max_secondaries=3, n_paralogs=2, hardclip_frac=0.2, softclip_frac=0.3, read_len=5000)
    par = records.preset("hifi")
    ctx.set_reference(g.ref)
    sizes = [17, 40, 3, 29, 64, 8, 31]
    reads, first = [], 0
    for n in sizes:
        reads.append(g.reads(first, n))
        first += n
    want = []
    for r in reads:
        out, _ = ctx.score_batch(r.batch, par, finalize_seed=None)
        want.append([(o.n_aln, tuple(o.score[a] for a in range(max(o.n_aln, 0))), o.prim_idx, o.max_idx, o.tie_mask, o.pass_,
                      tuple(o.rfe[a] for a in range(max(o.n_aln, 0))), o.n_problems, o.n_markers, o.dp_cells) for o in out])

    def sig(out, n):
        return [(o.n_aln, tuple(o.score[a] for a in range(max(o.n_aln, 0))), o.prim_idx, o.max_idx, o.tie_mask, o.pass_,
                 tuple(o.rfe[a] for a in range(max(o.n_aln, 0))), o.n_problems, o.n_markers, o.dp_cells) for o in out[:n]]

    pipe = api.Pipe(ctx, par, depth=3, host_threads=8)
    staged = [ctx.stage(r.batch, par) for r in reads]
    for rnd in range(2):  # the second round re-prepares the same staged records
        got = []
        sub = 0
        for k in range(len(reads)):
            while sub < len(reads) and pipe.pending() < 4:
                if (sub + rnd) % 2 == 0:
                    pipe.submit(batch=reads[sub].batch)
                else:
                    pipe.submit(staged=staged[sub])
                sub += 1
            out, n = pipe.next()
            assert n == sizes[k]
            got.append(sig(out, n))
        assert got == want, rnd
    # decision records: device pack kernel vs host conversion of the collected results
    import torch
    w = staged[4]
    w.prepare_staged()
    w.launch()
    out = w.collect(finalize_seed=None)
    dev = torch.zeros(sizes[4] * 16, dtype=torch.uint8, device="cuda")
    nd = w.pack_decisions(1000, dev.data_ptr(), sizes[4])
    host = (api.Decision * sizes[4])()
    nh = api.lib().spx_decisions_from_results(out, sizes[4], 1000, host, sizes[4])
    raw = dev.cpu().numpy().tobytes()
    drec = [api.Decision.from_buffer_copy(raw[16 * k:16 * k + 16]) for k in range(nd)]
    drec = [d for d in drec if d.n_aln >= 2]
    assert len(drec) == nh
    for a, b in zip(drec, host):
        assert (a.group, a.n_aln, a.prim_idx, a.max_idx, a.pass_, a.tie_mask, a.absdiff) == \
               (b.group, b.n_aln, b.prim_idx, b.max_idx, b.pass_, b.tie_mask, b.absdiff)
    pipe.close()
    for s in staged:
        s.free()


def test_pipeline_reports_group_errors_in_place(ctx):
    """a group the reference cannot score (N op in the CIGAR) keeps its slot and its error code inside a pipelined batch"""
    from common import HandBatch, HandRef
    seq = "ACGT" * 10
    hr = HandRef([("c0", seq * 4)])
    hb = HandBatch([("bad", [(0, 0, 0, "20M5N20M", seq, 30, ":20:20"), (256, 0, 3, "40M", seq, 30, ":40")]),
                    ("ok", [(0, 0, 0, "40M", seq, 30, ":40"), (256, 0, 3, "38M", seq[:38], 30, ":38")]),
                    ("notag", [(0, 0, 0, "40M", seq, 30, ":40"), (256, 0, 3, "38M", seq[:38], 30, None)])])
    ctx.set_reference(hr.ref)
    pipe = api.Pipe(ctx, records.preset("hifi"), depth=2, host_threads=2)
    pipe.submit(batch=hb.batch)
    out, n = pipe.next()
    assert n == 3 and out[0].n_aln == api.EUNSUPPORTED and out[1].n_aln == 2 and out[2].n_aln == api.ENOTAG
    pipe.close()


_FORCED_SHARE_SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from common import small_genome
from oracle import orc
from secphase_amd import api, records, synth
ctx = api.Context(0)
bad = 0
for kw, first, n, par in (
        (dict(platform=synth.MIXED, n_paralogs=7, contig_len=250000, max_read_len=40000), 7, 96, records.preset("hifi")),
        (dict(platform=synth.HIFI, hardclip_frac=0.5, softclip_frac=0.5, shuffle_records=1, inverted_paralogs=1, n_paralogs=3, max_secondaries=4,
              n_base_frac=0.002, read_len=5000), 100, 80, records.preset("hifi")),
        (dict(platform=synth.ONT, n_paralogs=3, read_len=6000), 0, 40, records.preset("ont", bandwidth=50))):
    plat = kw.pop("platform")
    g = small_genome(plat, **kw)
    r = g.reads(first, n)
    ctx.set_reference(g.ref)
    out, st = ctx.score_batch(r.batch, par, finalize_seed=1)
    _, res = orc.run_batch(r.batch, g.ref, par, threads=4, seed=1)
    for i in range(n):
        o, e = out[i], res[i]
        ok = o.n_aln == e.n_aln and o.best_idx == e.best_idx and all(o.score[a] == e.score[a] for a in range(max(e.n_aln, 0)))
        bad += not ok
    assert st.n_problems == sum(x.n_baq_calls for x in res)
print("differing groups:", bad)
sys.exit(1 if bad else 0)
"""


def test_extracted_alignments_on_small_batches(built):
    """The extraction of heavy alignments / groups (lone-lane waves; marker columns and BAQ blocks shared by the 64 lanes of a wave,
    spx_prep_kernels.hip) only triggers on lists with long reads.  SPX_PREP_HEAVY_MIN=1 makes every item that is four times its list's
    median 'heavy': the mixed, the clipped / shuffled / inverted and the ONT batches of this suite then go through those paths, and
    scores, decisions and problem counts must still equal the oracle's.  (A child process: the switches are read once per process.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _FORCED_SHARE_SCRIPT, root], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SPX_PREP_HEAVY_MIN="1", SPX_PREP_HEAVY="16", SPX_PREP_SHARE="4096", GPU_MAX_HW_QUEUES="14"))
    assert p.returncode == 0, (p.stdout[-300:], p.stderr[-800:])
    assert "differing groups: 0" in p.stdout
