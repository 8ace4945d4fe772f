"""The one line of probaln_glocal whose reading is open (include/spx.h SPX_GUARD_*): the guard of the
termination sum / backward start, `u >= bw2*3+3` (BAND, the default) or `u >= i_dim-3` (ROW).  One switch selects the reading
in the oracle and in the product; both sides are compared bit for bit under BOTH settings, so that a pin against a real
htslib 1.17 (tools/pin_htslib) is a one-constant flip.  Call site: /root/reference/programs/submodules/ptMarker/ptMarker.c:754-757."""
import importlib.util
import os

import numpy as np
import pytest

from common import oracle_probaln, small_genome
from oracle import orc
from secphase_amd import api, records, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H_TDROP = 14  # spx_device.h SPX_H_TDROP


def _regime_block():
    spec = importlib.util.spec_from_file_location("make_problems", os.path.join(ROOT, "tools", "pin_htslib", "make_problems.py"))
    mp_ = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mp_)
    return mp_.problems()[-240:]


def in_regime(L, R, bw_in):
    bw = min(max(L, R), bw_in)
    bw = max(bw, abs(R - L))
    return L <= bw and 2 * bw + 1 > R


@pytest.fixture(params=[orc.GUARD_BAND, orc.GUARD_ROW], ids=["band", "row"])
def guard(request, built):
    """sets the reading on both sides, restores the default afterwards"""
    orc.set_terminal_guard(request.param)
    api.set_terminal_guard(request.param)
    yield request.param
    orc.set_terminal_guard(orc.GUARD_BAND)
    api.set_terminal_guard(api.GUARD_BAND)


def test_switch_round_trip(built):
    assert api.get_terminal_guard() == api.GUARD_BAND and orc.get_terminal_guard() == orc.GUARD_BAND
    api.set_terminal_guard(api.GUARD_ROW)
    assert api.get_terminal_guard() == api.GUARD_ROW
    api.set_terminal_guard(api.GUARD_BAND)
    with pytest.raises(api.SpxError):
        api.set_terminal_guard(7)


def test_readings_differ_in_the_regime_and_nowhere_else(built):
    """oracle only: outside (l_query <= bw and 2*bw+1 > l_ref) the two readings give the same s[], state[], q[]; inside,
    column l_ref leaves the termination, s[l_query+1] shrinks and the posteriors of the last rows move"""
    rng = np.random.default_rng(5)
    probs = []
    for _ in range(120):
        L = int(rng.integers(1, 160))
        R = max(1, L + int(rng.integers(-12, 13)))
        bw = abs(R - L) + int(rng.choice([1, 5, 20, 50, 130]))
        ref = rng.integers(0, 4, R).astype(np.uint8)
        qry = np.resize(ref, L).copy()
        m = rng.random(L) < 0.05
        qry[m] = (qry[m] + 1) % 4
        probs.append((ref, qry, bw))
    n_in = n_out = 0
    try:
        for ref, qry, bw in probs:
            orc.set_terminal_guard(orc.GUARD_BAND)
            pa, sa, qa = oracle_probaln(ref, qry, 20, 1e-3, 0.1, bw)
            s0, zM0, zI0 = orc.probaln_posteriors(ref, qry, 20, 1e-3, 0.1, bw)
            orc.set_terminal_guard(orc.GUARD_ROW)
            pb, sb, qb = oracle_probaln(ref, qry, 20, 1e-3, 0.1, bw)
            s1, zM1, zI1 = orc.probaln_posteriors(ref, qry, 20, 1e-3, 0.1, bw)
            L = len(qry)
            assert np.array_equal(s0[:L + 1], s1[:L + 1])          # the forward pass does not depend on the guard
            if in_regime(L, len(ref), bw):
                n_in += 1
                assert s1[L + 1] <= s0[L + 1]              # (equal when the dropped cell is below one ulp of the sum)
                at = (L - 1, len(ref) - 1)
                assert zM1[at] == 0.0 and zI1[at] == 0.0 and zM0[at] + zI0[at] > 0.0
            else:
                n_out += 1
                assert pa == pb and np.array_equal(sa, sb) and np.array_equal(qa, qb) and s0[L + 1] == s1[L + 1]
                assert np.array_equal(zM0, zM1)
    finally:
        orc.set_terminal_guard(orc.GUARD_BAND)
    assert n_in >= 20 and n_out >= 20


def test_pin_problems_cover_the_regime(built):
    block = _regime_block()
    assert all(in_regime(len(q), len(r), bw) for r, q, bw, *_ in block)


def test_pin_tool_names_the_reading_of_the_library_it_ran_against(guard, tmp_path):
    """tools/pin_htslib/make_vectors.py with the ORACLE standing in for htslib, once per reading: the tool must report the reading
    its answers were produced under (the step that turns the eventual pin into a one-constant flip)"""
    import json
    import subprocess
    import sys
    tool_dir = os.path.join(ROOT, "tools", "pin_htslib")
    problems = subprocess.run([sys.executable, os.path.join(tool_dir, "make_problems.py")], capture_output=True, text=True, check=True).stdout
    lines = problems.splitlines()[:60] + problems.splitlines()[-240:]
    answers = []
    for ln in lines:
        f = ln.split()
        ref = np.array([int(c) for c in f[6]], np.uint8)
        qry = np.array([int(c) for c in f[7]], np.uint8)
        pr, st, q = oracle_probaln(ref, qry, int(f[5]), float(f[3]), float(f[4]), int(f[2]))
        answers.append(" ".join(str(x) for x in [pr] + st.tolist() + q.tolist()))
    (tmp_path / "p.txt").write_text("\n".join(lines) + "\n")
    (tmp_path / "a.txt").write_text("\n".join(answers) + "\n")
    out = tmp_path / "vectors.json"
    subprocess.run([sys.executable, os.path.join(tool_dir, "make_vectors.py"), str(tmp_path / "p.txt"), str(tmp_path / "a.txt"), "oracle-as-htslib",
                    str(out)], check=True, capture_output=True, env=dict(os.environ, SPX_TERMINAL_GUARD="band"))
    j = json.load(open(out))
    assert j["terminal_guard"] == ("row" if guard == orc.GUARD_ROW else "band")
    n = j["guard_report"]["regime_vectors"]   # (a few of the general problems fall into the regime too)
    assert n >= 240 and j["guard_report"]["row_matches" if guard == orc.GUARD_ROW else "band_matches"] == n
    assert j["guard_report"]["band_matches" if guard == orc.GUARD_ROW else "row_matches"] < n


def test_host_plan_marks_exactly_the_regime_problems(guard):
    """the product's work list carries the reading as a per-problem flag beside the HMM constants (what the kernels read)"""
    g = small_genome(synth.ONT, n_paralogs=3, read_len=6000)
    r = g.reads(0, 16)
    p = records.preset("ont", bandwidth=50)
    p.conf_b = 200.0   # a wide band puts a good share of the blocks into the regime
    plan = api.Plan(g.ref, r.batch, p)   # (keeps the arrays of the view alive)
    v = plan.view
    n = v.n_problems
    assert n > 50
    hmm = np.ctypeslib.as_array(v.hmm, shape=(n * 16,)).reshape(n, 16)
    L = np.ctypeslib.as_array(v.L, shape=(n,))
    R = np.ctypeslib.as_array(v.R, shape=(n,))
    bw = np.ctypeslib.as_array(v.bw, shape=(n,))
    reg = (L <= bw) & (2 * bw + 1 > R)
    assert reg.sum() > 0 and (~reg).sum() > 0
    want = reg.astype(float) if guard == orc.GUARD_ROW else np.zeros(n)
    assert np.array_equal(hmm[:, H_TDROP], want)


def test_host_plan_equals_oracle_under_both_readings(guard):
    """whole groups on the CPU: host plan + the oracle's DP (which follows the same switch) against the oracle's scores"""
    from test_host_plan import _compare
    g = small_genome(synth.ONT, n_paralogs=3, read_len=5000)
    r = g.reads(0, 8)
    p = records.preset("ont", bandwidth=50)
    p.conf_b = 150.0
    _compare(g.ref, r.batch, p)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx(built):
    c = api.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_regime_block_on_the_kernels(ctx, guard):
    """the 240 problems tools/pin_htslib sets aside for the regime: state[], q[] of every problem and every posterior product
    of a sample equal the oracle's under the same reading -- for both readings"""
    from test_gpu_parity import _check, _posteriors_equal
    block = _regime_block()
    probs = [(np.array(r, np.uint8), np.array(q, np.uint8)) for r, q, *_ in block]
    pars = [(d, e, bw) for _, _, bw, d, e, _ in block]
    sq = [v[5] for v in block]
    _check(ctx, probs, sq, pars)
    for k in range(0, 240, 6):
        _posteriors_equal(ctx, probs[k][0], probs[k][1], sq[k], pars[k])


@pytest.mark.gpu
def test_every_band_class_under_both_readings(ctx, guard):
    """problems inside and outside the regime for every kernel instantiation (one lane per problem, 2 / 4 / 8 / 16 / 32 / 64
    lanes): scaling factors and posterior products, bit for bit"""
    from test_gpu_parity import _posteriors_equal
    rng = np.random.default_rng(21)
    shapes = [(15, 18, 20), (20, 22, 21), (22, 20, 22), (18, 30, 23), (30, 41, 40), (40, 50, 52), (45, 60, 55), (57, 70, 58),
              (50, 60, 60), (100, 110, 120), (200, 220, 250), (300, 333, 400), (500, 480, 700), (100, 100, 20), (300, 290, 52),
              (18, 30, 24), (22, 40, 25), (60, 34, 26), (25, 52, 27), (300, 276, 24), (280, 307, 27), (90, 118, 28), (70, 40, 31)]
    n_reg = 0
    for (L, R, bw) in shapes:
        n_reg += in_regime(L, R, bw)
        for kind in ("related", "homopolymer"):
            ref = rng.integers(0, 4, R).astype(np.uint8) if kind == "related" else np.full(R, 3, np.uint8)
            qry = np.resize(ref, L).copy()
            m = rng.random(L) < 0.03
            qry[m] = (qry[m] + 1) % 4
            _posteriors_equal(ctx, ref, qry, 20, (1e-3, 0.1, bw))
    assert n_reg >= 10


@pytest.mark.gpu
def test_general_kernel_under_both_readings(built, guard):
    """spx_probaln_glocal with per-base qualities (spx_probaln_general.hip) honours the switch too"""
    import ctypes as C
    L_ = api.lib()
    rng = np.random.default_rng(3)
    for (L, R, bw) in ((20, 25, 30), (40, 38, 50), (60, 70, 20)):
        ref = rng.integers(0, 4, R).astype(np.uint8)
        qry = np.resize(ref, L).copy()
        iq = rng.integers(5, 41, L).astype(np.uint8)
        st = np.zeros(L, np.int32)
        q = np.zeros(L, np.uint8)
        par = api.ProbalnPar(1e-3, 0.1, bw)
        u8 = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint8))
        pr = L_.spx_probaln_glocal(u8(ref), R, u8(qry), L, u8(iq), C.byref(par), st.ctypes.data_as(C.POINTER(C.c_int)), u8(q))
        est = np.zeros(L, np.int32)
        eq = np.zeros(L, np.uint8)
        opar = orc.ProbalnPar(1e-3, 0.1, bw)
        epr = orc.lib().orc_probaln_glocal(u8(ref), R, u8(qry), L, u8(iq), C.byref(opar), est.ctypes.data_as(C.POINTER(C.c_int)), u8(eq))
        assert pr == epr and np.array_equal(st, est) and np.array_equal(q, eq), (L, R, bw)


@pytest.mark.gpu
def test_ont_batch_under_both_readings(ctx, guard, tmp_path):
    """whole groups through the HIP path (`--ont -b 50`, where 1.5 % of the DP problems are in the regime): scores, decisions
    and the relabel list equal the oracle's under the same reading"""
    from test_gpu_parity import _batch_parity
    g = small_genome(synth.ONT, n_paralogs=3)
    r = g.reads(0, 24)
    _batch_parity(ctx, g, r, records.preset("ont", bandwidth=50), tmp_path, f"ont_guard{guard}")
    p = records.preset("ont", bandwidth=50)
    p.conf_b = 150.0
    _batch_parity(ctx, g, r, p, tmp_path, f"ont_wide_guard{guard}")
