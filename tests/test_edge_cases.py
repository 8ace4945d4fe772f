"""The edge cases of SURVEY section 7 step 2 on hand-built groups (tests/edgecases.py): every test first ASSERTS that its case occurred
(from the host plan and the oracle's per-group results), then compares -- CPU: host plan + oracle DP with the oracle; `-m gpu`: the HIP
path (work list, scores, decisions, relabel list) with the oracle.  Reference lines: edgecases.py."""
import ctypes as C
import filecmp

import numpy as np
import pytest

import edgecases as ec
from common import HandBatch, HandRef, emulate_plan
from oracle import orc
from secphase_amd import api, records


def _build(cases, preset):
    b, at = cases()
    return b, at, HandBatch(b.groups), HandRef(b.contigs), preset


def _hifi():
    return _build(ec.hifi_cases, records.preset("hifi"))


def _ont():
    return _build(ec.ont_cases, records.preset("ont", bandwidth=50))


def _group_view(plan, g):
    """(markers, zeroed markers, wanted rows [(problem, row)], problems) of input group g in a host plan"""
    v = plan.view
    idx = list(v.grp_index[:v.n_groups])
    k = idx.index(g)
    mk = range(v.mk_first[k], v.mk_first[k + 1])
    zeroed = [i for i in mk if v.mk_row[i] < 0 and v.mk_qfix[i] == 0]
    rows = sorted({int(v.mk_row[i]) for i in mk if v.mk_row[i] >= 0})
    probs = set()
    for p in range(v.n_problems):
        r0, nr = v.row_off[p], v.n_rows_of[p]
        if any(r0 <= r < r0 + nr for r in rows):
            probs.add(p)
    return list(mk), zeroed, rows, sorted(probs)


def _occurred(b, at, hb, hr, par):
    """asserts on the host plan / the oracle that every case of the batch happened; returns (plan, oracle results)"""
    built = pytest.importorskip("__graft_entry__")
    built.build_cpu_helpers()
    _, res = orc.run_batch(hb.batch, hr.ref, par, threads=2, seed=1)
    plan = api.Plan(hr.ref, hb.batch, par)
    v = plan.view
    idx = list(v.grp_index[:v.n_groups])
    for name, want in b.expected.items():
        g = at[name]
        assert g in idx, name
        mk, zeroed, rows, probs = _group_view(plan, g)
        if "positions" in want:
            assert len(mk) == want["positions"] * want["n_aln"], (name, len(mk))
        if "zeroed_per_alignment" in want:
            assert len(zeroed) == want["zeroed_per_alignment"] * want["n_aln"], (name, len(zeroed))
        if "row_first" in want:  # a wanted row at base sqs + 10 (row 11, 1-based) and one at sqe - 10 ... which is then zeroed
            assert any(v.rows[r] == want["row_first"] for r in rows), name
    if "edges" in at:
        # positions 5 (inside [sqs, sqs+10)), 1995 (inside [sqe-10, sqe]) and 1989 = sqe - 10 (inside the window the BAQ values are written for,
        # [sqs+10, sqe-10], and zeroed afterwards) are the three zeroed markers per alignment asserted above; position 10 = sqs + 10 is the wanted
        # row 11 asserted above
        g = at["edges"]
        mk, zeroed, rows, probs = _group_view(plan, g)
        assert res[g].n_baq_calls >= 2 and len(rows) >= 2
    if "tie3" in at:
        e = res[at["tie3"]]
        sec = [e.score[a] for a in range(1, e.n_aln)]
        assert e.n_aln == 4 and len(set(sec)) == 1 and sec[0] > e.score[0] + par.prim_margin_score and sec[0] >= par.min_score
        assert e.n_rand == 2 and e.relabel == 1 and e.best_idx in (1, 2, 3)      # rand() % 3, then rand() % 2
    if "ten" in at:
        assert res[at["ten"]].n_aln == 10 and at["ten"] in idx
        assert at["eleven"] not in idx and not orc.lib().orc_group_is_dispatched(hb.batch, at["eleven"])
    if "long_cs" in at:
        recs = dict(b.groups)["longcs"]
        assert recs[0][6].startswith("=") and "~" in recs[1][6]
    if "guard" in at:
        mk, zeroed, rows, probs = _group_view(plan, at["guard"])
        assert probs and all(v.L[p] <= v.bw[p] and 2 * v.bw[p] + 1 > v.R[p] for p in probs)   # the regime of the terminal guard
    return plan, res


@pytest.mark.parametrize("cases", [_hifi, _ont])
def test_cases_occur_and_host_plan_equals_oracle(cases):
    b, at, hb, hr, par = cases()
    plan, res = _occurred(b, at, hb, hr, par)
    em = emulate_plan(plan, hr.ref, par)
    assert em, "no dispatched group"
    for k, (sc, prim, mx, tie, ok) in em.items():
        assert [res[k].score[a] for a in range(res[k].n_aln)] == sc, k
        assert res[k].prim_idx == prim, k
    if "tie3" in at:
        assert bin(em[at["tie3"]][3]).count("1") == 3     # three secondaries share the greatest score


@pytest.mark.gpu
@pytest.mark.parametrize("cases", [_hifi, _ont])
@pytest.mark.parametrize("tiers", [1, 0])
def test_cases_on_the_device(cases, tiers, tmp_path, monkeypatch):
    from test_gpu_parity import _plans_equal
    b, at, hb, hr, par = cases()
    plan, res = _occurred(b, at, hb, hr, par)
    monkeypatch.setenv("SPX_FAST_MIN_SHARE", "0")
    monkeypatch.setenv("SPX_FAST_MIN_TOTAL", "0")
    old = api.get_dp_tiers()
    api.set_dp_tiers(tiers)
    ctx = api.Context(0)
    try:
        ctx.set_reference(hr.ref)
        w = ctx.prepare(hb.batch, par)
        dev = w.export_plan()
        _plans_equal(dev, plan, hb.struct.n_groups)
        dev.close()
        w.launch()
        out = w.collect(finalize_seed=1)
        w.free()
        log_o, log_g = str(tmp_path / "o.log"), str(tmp_path / "g.log")
        orc.run_batch(hb.batch, hr.ref, par, threads=2, seed=1, log_path=log_o)
        api.write_relabel_log(log_g, hb.batch, hr.ref, out)
        for g in range(hb.struct.n_groups):
            o, e = out[g], res[g]
            assert o.n_aln == e.n_aln, g
            assert [o.score[a] for a in range(max(e.n_aln, 0))] == [e.score[a] for a in range(max(e.n_aln, 0))], g
            assert o.best_idx == e.best_idx and bool(o.relabel) == bool(e.relabel), g
        assert filecmp.cmp(log_o, log_g, shallow=False)
        if "tie3" in at:
            assert bin(out[at["tie3"]].tie_mask).count("1") == 3
    finally:
        api.set_dp_tiers(old)
        ctx.close()
