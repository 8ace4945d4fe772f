"""-m gpu: the two-tier DP -- certified fast kernels + exact re-run of what they do not certify.

What must hold: with the tiers ON, (state, q) of EVERY wanted row equals the CPU oracle's, whatever share of the problems the fast
tier certified; the fast tier really is what computed most of them (tier statistics); the same with nothing certified (test mode 2:
every problem takes the re-run path) and with the tiers off (the exact kernels alone); under both readings of the terminal guard."""
import numpy as np
import pytest

from common import oracle_probaln, small_genome
from oracle import orc
from secphase_amd import api, records, synth
from test_gpu_parity import _batch_parity, _rand_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(built):
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture
def tiers():
    old = api.get_dp_tiers()

    def set_(mode):
        api.set_dp_tiers(mode)
    yield set_
    api.set_dp_tiers(old)


def _check(ctx, probs, set_q, pars):
    st, qq, _ = ctx.probaln_batch([p[0] for p in probs], [p[1] for p in probs], set_q, pars)
    bad = []
    for i, (r, q) in enumerate(probs):
        _, est, eq = oracle_probaln(r, q, set_q[i], pars[i][0], pars[i][1], pars[i][2])
        if not (np.array_equal(st[i], est) and np.array_equal(qq[i], eq)):
            rows = np.nonzero((st[i] != est) | (qq[i] != eq))[0]
            bad.append((i, len(q), len(r), pars[i][2], rows[:5].tolist()))
    assert not bad, f"{len(bad)} problems differ from the oracle: {bad[:5]}"
    return api.last_tier_stats()


def _hifi_problems(rng, n):
    probs = [_rand_problem(rng, int(rng.integers(200, 1001)), 0.002, 0.003) for _ in range(n)]
    pars = [(1e-4, 0.1, abs(len(r) - len(q)) + 20) for r, q in probs]
    return probs, [40] * n, pars


def _ont_problems(rng, n, bw=50):
    probs = [_rand_problem(rng, int(rng.integers(60, 900)), 0.04, 0.02) for _ in range(n)]
    pars = [(1e-3, 0.1, abs(len(r) - len(q)) + bw) for r, q in probs]
    return probs, [20] * n, pars


def test_fast_tier_hifi_windows(ctx, tiers):
    """band widths 41..47 (one lane per problem) and the neighbouring generic classes: equal to the oracle, and certified by the fast tier"""
    tiers(1)
    rng = np.random.default_rng(61)
    probs, sq, pars = _hifi_problems(rng, 384)
    nf, by_cert, by_model, by_range, rows = _check(ctx, probs, sq, pars)
    assert nf >= 300, nf                      # (a few windows have |R - L| > 3: wider classes)
    assert by_cert + by_model + by_range <= nf // 20, (nf, by_cert, by_model, by_range)


def test_fast_tier_ont_windows(ctx, tiers):
    """four lanes per problem (band widths 101..119), the carry scan across the lanes"""
    tiers(1)
    rng = np.random.default_rng(62)
    probs, sq, pars = _ont_problems(rng, 192)
    nf, by_cert, by_model, by_range, rows = _check(ctx, probs, sq, pars)
    assert nf >= 100, nf
    assert by_cert + by_model + by_range <= nf // 4, (nf, by_cert, by_model, by_range)


@pytest.mark.parametrize("bw", [3, 10, 21, 24, 27, 30, 35, 40, 45, 51, 52, 55, 59, 60, 62, 63])
def test_fast_tier_every_band_class(ctx, tiers, bw, monkeypatch):
    """generic classes (2,24), (4,16), (4,26), (4,28), (4,30) at widths that leave 0 .. many slots beyond the band; bw 60..63: the first class
    WITHOUT a fast kernel (W = 121..128: exact kernels beside the tiers)"""
    monkeypatch.setenv("SPX_FAST_MIN_SHARE", "0")  # (every class with a fast kernel takes it, however few cells it holds)
    monkeypatch.setenv("SPX_FAST_MIN_TOTAL", "0")
    tiers(1)
    rng = np.random.default_rng(100 + bw)
    probs = [_rand_problem(rng, int(rng.integers(20, 500)), 0.01, 0.02) for _ in range(64)]
    pars = [(1e-3, 0.1, abs(len(r) - len(q)) + bw) for r, q in probs]
    _check(ctx, probs, [25] * len(probs), pars)


def test_fast_tier_small_and_degenerate(ctx, tiers, monkeypatch):
    monkeypatch.setenv("SPX_FAST_MIN_SHARE", "0")
    monkeypatch.setenv("SPX_FAST_MIN_TOTAL", "0")
    tiers(1)
    rng = np.random.default_rng(63)
    probs, pars = [], []
    for L in (1, 2, 3, 5, 8, 13, 15, 16, 17, 18, 21, 22, 31, 32, 33, 34, 40, 41, 42, 60):
        for bw in (1, 3, 20, 50):
            r, q = _rand_problem(rng, L, 0.05, 0.05)
            probs.append((r, q))
            pars.append((1e-4, 0.1, abs(len(r) - len(q)) + bw))
    probs.append((np.array([0, 1, 2], np.uint8), rng.integers(0, 4, 30).astype(np.uint8)))
    pars.append((1e-3, 0.1, 27 + 5))
    probs.append((rng.integers(0, 4, 90).astype(np.uint8), rng.integers(0, 4, 25).astype(np.uint8)))
    pars.append((1e-3, 0.1, 65 + 2))
    _check(ctx, probs, [30] * len(probs), pars)


def test_fast_tier_outside_the_model(ctx, tiers, monkeypatch):
    """ambiguous bases, unrelated sequences (dynamic range), extreme parameters: all answered by the exact re-run"""
    monkeypatch.setenv("SPX_FAST_MIN_SHARE", "0")
    monkeypatch.setenv("SPX_FAST_MIN_TOTAL", "0")
    tiers(1)
    rng = np.random.default_rng(64)
    probs = [_rand_problem(rng, int(rng.integers(100, 400)), 0.01, 0.01, n_frac=0.03) for _ in range(32)]
    pars = [(1e-4, 0.1, abs(len(r) - len(q)) + 20) for r, q in probs]
    nf, by_cert, by_model, by_range, _ = _check(ctx, probs, [40] * len(probs), pars)
    assert by_model >= 20, (nf, by_model)
    # unrelated sequences over many rows
    probs = [(rng.integers(0, 4, 700 + 3 * k).astype(np.uint8), rng.integers(0, 4, 700).astype(np.uint8)) for k in range(8)]
    pars = [(1e-4, 0.1, 3 * k + 20) for k in range(8)]
    _check(ctx, probs, [40] * 8, pars)
    # parameters far from the presets (set_q 93: the smallest factor per row is below the model's limit; tiny gap open)
    probs = [_rand_problem(rng, 300, 0.01, 0.01) for _ in range(16)]
    pars = [((1e-6, 0.01, abs(len(r) - len(q)) + 20) if k % 2 else (0.05, 0.5, abs(len(r) - len(q)) + 20)) for k, (r, q) in enumerate(probs)]
    _check(ctx, probs, [93 if k % 4 == 0 else 5 for k in range(16)], pars)


def test_nothing_certified_equals_everything_certified(ctx, tiers, monkeypatch):
    """test mode 2: the fast kernels run, every range check fails, every problem is re-run by the exact kernels"""
    monkeypatch.setenv("SPX_FAST_MIN_SHARE", "0")
    monkeypatch.setenv("SPX_FAST_MIN_TOTAL", "0")
    rng = np.random.default_rng(65)
    probs, sq, pars = _hifi_problems(rng, 96)
    p2, s2, r2 = _ont_problems(rng, 48)
    tiers(2)
    nf, by_cert, by_model, by_range, _ = _check(ctx, probs + p2, sq + s2, pars + r2)
    assert by_range + by_model + by_cert >= nf > 0, (nf, by_cert, by_model, by_range)
    tiers(0)
    assert _check(ctx, probs + p2, sq + s2, pars + r2) == (0, 0, 0, 0, 0)


@pytest.mark.parametrize("guard", [orc.GUARD_BAND, orc.GUARD_ROW])
def test_fast_tier_under_both_guard_readings(ctx, tiers, guard, monkeypatch):
    """blocks of <= 50 bases under `-b 50` (l_query <= bw, 2 bw + 1 > l_ref): the one band cell the two readings of probaln.c's terminal
    guard treat differently -- the fast tier follows the same switch"""
    monkeypatch.setenv("SPX_FAST_MIN_TOTAL", "0")
    tiers(1)
    rng = np.random.default_rng(66)
    old_a, old_o = api.get_terminal_guard(), orc.get_terminal_guard()
    try:
        api.set_terminal_guard(guard)
        orc.set_terminal_guard(guard)
        probs = [_rand_problem(rng, int(rng.integers(8, 51)), 0.03, 0.02) for _ in range(128)]
        pars = [(1e-3, 0.1, abs(len(r) - len(q)) + 50) for r, q in probs]
        nf, by_cert, by_model, by_range, _ = _check(ctx, probs, [20] * len(probs), pars)
        assert nf > 0
    finally:
        api.set_terminal_guard(old_a)
        orc.set_terminal_guard(old_o)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_batches_in_every_tier_mode(ctx, tiers, tmp_path, mode, monkeypatch):
    """whole batches (scores, decisions, relabel list byte for byte): exact kernels alone, two tiers, two tiers with nothing certified"""
    monkeypatch.setenv("SPX_FAST_MIN_TOTAL", "0")  # (also the mixed batch, whose cells spread over many classes, takes the tiers)
    tiers(mode)
    g = small_genome(synth.HIFI)
    _batch_parity(ctx, g, g.reads(0, 96), records.preset("hifi"), tmp_path, f"hifi{mode}")
    g2 = small_genome(synth.ONT)
    _batch_parity(ctx, g2, g2.reads(0, 12), records.preset("ont", bandwidth=50), tmp_path, f"ont{mode}")
    g3 = small_genome(synth.MIXED)
    _batch_parity(ctx, g3, g3.reads(0, 48), records.preset("hifi"), tmp_path, f"mixed{mode}")
    st = api.last_tier_stats()
    if mode == 0:
        assert st == (0, 0, 0, 0, 0)
    else:
        assert st[0] > 0


@pytest.mark.parametrize("guard", [orc.GUARD_BAND, orc.GUARD_ROW])
@pytest.mark.parametrize("name", ["hifi", "ont", "edge", "mixed"])
def test_fixture_problems_state_and_q_on_every_row(ctx, tiers, name, guard, monkeypatch):
    """the DP problems of the committed fixtures' batches (tests/golden: hifi, ont, edge) and of a mixed 2-100 kb batch, as the host plan
    cuts them: fast tier + certificate + exact re-run give the oracle's (state, q) on EVERY row (the wanted rows are a subset), under both
    readings of the terminal guard"""
    from common import nibbles, ref_codes
    from golden.make_golden import CASES
    monkeypatch.setenv("SPX_FAST_MIN_SHARE", "0")
    monkeypatch.setenv("SPX_FAST_MIN_TOTAL", "0")
    tiers(1)
    if name == "mixed":
        c = dict(platform=synth.MIXED, cfg=dict(n_contigs=2, contig_len=150000), first=0, n=12, preset=("hifi", None))
    else:
        c = CASES[name]
    old_a, old_o = api.get_terminal_guard(), orc.get_terminal_guard()
    try:
        api.set_terminal_guard(guard)
        orc.set_terminal_guard(guard)
        g = synth.Genome(synth.default_cfg(c["platform"], **c["cfg"]))
        r = g.reads(c["first"], c["n"])
        par = records.preset(c["preset"][0], bandwidth=c["preset"][1])
        plan = api.Plan(g.ref, r.batch, par)
        v = plan.view
        n = min(int(v.n_problems), 1500)
        assert n > 0
        probs = [(ref_codes(g.ref, v.ref_tid[p], v.ref_rfs[p], v.R[p]), nibbles(v.qry4, v.qry_nib[p], v.L[p])) for p in range(n)]
        pars = [(par.conf_d, par.conf_e, int(v.bw[p])) for p in range(n)]
        nf, by_cert, by_model, by_range, _ = _check(ctx, probs, [par.set_q] * n, pars)
        assert nf > 0 and by_cert + by_model + by_range < nf, (nf, by_cert, by_model, by_range)   # the fast tier answered problems of its own
    finally:
        api.set_terminal_guard(old_a)
        orc.set_terminal_guard(old_o)
