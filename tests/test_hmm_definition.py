"""CPU: the oracle's probaln_glocal against an INDEPENDENT evaluation of the same profile HMM, written from the model's
definition in exact rational arithmetic -- full (L+1) x (R+1) matrices, no band-index macro, no scaling, no float
rounding anywhere -- so that a formula-level misreading of htslib's probaln.c in oracle/probaln_oracle.c (index macro,
transition table, begin / end probabilities, which neighbours a state reads, which states the MAP step maximises over)
cannot hide behind the fact that oracle and kernels were restated by the same hand.

The model (Heng Li's kprobaln / htslib probaln.c, "glocal": the query is aligned end to end, the reference locally):
  states M, I, D per cell (i, k), query base i = 1..L, reference base k = 1..R, band  max(1, i-bw) <= k <= min(R, i+bw)
  begin   M(1,k) = e(1,k) * (1-d)/R        I(1,k) = EI * d/R        D(1,k) = 0 (no deletion state in the first row)
  forward M(i,k) = e(i,k) * [ mMM M(i-1,k-1) + mIM I(i-1,k-1) + mDM D(i-1,k-1) ]
          I(i,k) = EI     * [ mMI M(i-1,k)   + mII I(i-1,k) ]
          D(i,k) =          [ mMD M(i,k-1)   + mDD D(i,k-1) ]
  end     P = sum_k  M(L,k) sM + I(L,k) sI
  with  mMM = (1-2d)(1-sM), mMI = mMD = d(1-sM), mIM = (1-e)(1-sI), mII = e(1-sI), mDM = 1-e, mDD = e,
        sM = sI = 1/(2L+2), EI = 1/4, e(i,k) = 1-p on a match, p * 0.33333333333 on a mismatch, 1 if either base is N,
        p = 10^(-q/10) stored as a float, d and e stored as floats.
  posterior of query base i being in state s at column k = forward * backward / P;  the MAP step takes the maximum over
  the M and I states of the row and reports phred(1 - max / sum).
This is only a statement of the model: that htslib 1.17 computes exactly this is what tools/pin_htslib/ is for.
"""
import math
from fractions import Fraction

import numpy as np
import pytest

from common import oracle_probaln
from oracle import orc

EI = Fraction(1, 4)
EM = Fraction("0.33333333333")


def _f32(x):
    return Fraction(float(np.float32(x)))


def model_posteriors(ref, qry, set_q, d, e, bw_in, last_column_may_end=True):
    """exact posteriors zM[i][k], zI[i][k] (1-based, 0 outside the band) and the likelihood P, as Fractions.
    last_column_may_end=False: the variant of the model in which an alignment may not END in reference column R."""
    L, R = len(qry), len(ref)
    bw = max(R, L)
    bw = min(bw, bw_in)
    bw = max(bw, abs(R - L))
    # the model's CONSTANTS are single-precision numbers in htslib (probaln_par_t holds floats; 1 - d - d, 1 - e,
    # (1 - d) / l_ref and d / l_ref are float expressions): they are taken as such, everything after is exact
    df, ef = np.float32(d), np.float32(e)
    one = np.float32(1)
    d, e = Fraction(float(df)), Fraction(float(ef))
    one_2d = Fraction(float((one - df) - df))
    one_e = Fraction(float(one - ef))
    begM = Fraction(float((one - df) / np.float32(R)))
    begI = Fraction(float(df / np.float32(R)))
    p = Fraction(float(np.float32(10.0 ** (-set_q / 10.0))))
    sM = sI = Fraction(1, 2 * L + 2)
    mMM, mMI, mMD = one_2d * (1 - sM), d * (1 - sM), d * (1 - sM)
    mIM, mII = one_e * (1 - sI), e * (1 - sI)
    mDM, mDD = one_e, e
    inband = lambda i, k: 1 <= k <= R and max(1, i - bw) <= k <= min(R, i + bw)

    def emit(i, k):
        a, b = int(ref[k - 1]), int(qry[i - 1])
        if a > 3 or b > 3:
            return Fraction(1)
        return 1 - p if a == b else p * EM

    Z = Fraction(0)
    fM = [[Z] * (R + 2) for _ in range(L + 2)]
    fI = [[Z] * (R + 2) for _ in range(L + 2)]
    fD = [[Z] * (R + 2) for _ in range(L + 2)]
    for k in range(1, R + 1):
        if inband(1, k):
            fM[1][k] = emit(1, k) * begM
            fI[1][k] = EI * begI
    for i in range(2, L + 1):
        for k in range(1, R + 1):
            if not inband(i, k):
                continue
            fM[i][k] = emit(i, k) * (mMM * fM[i - 1][k - 1] + mIM * fI[i - 1][k - 1] + mDM * fD[i - 1][k - 1])
            fI[i][k] = EI * (mMI * fM[i - 1][k] + mII * fI[i - 1][k])
            fD[i][k] = mMD * fM[i][k - 1] + mDD * fD[i][k - 1]
    may_end = lambda k: inband(L, k) and (last_column_may_end or k != R)
    P = sum((fM[L][k] * sM + fI[L][k] * sI for k in range(1, R + 1) if may_end(k)), Z)
    bM = [[Z] * (R + 3) for _ in range(L + 3)]
    bI = [[Z] * (R + 3) for _ in range(L + 3)]
    bD = [[Z] * (R + 3) for _ in range(L + 3)]
    for k in range(1, R + 1):
        if may_end(k):
            bM[L][k], bI[L][k] = sM, sI
    for i in range(L - 1, 0, -1):
        for k in range(R, 0, -1):
            if not inband(i, k):
                continue
            nxt = emit(i + 1, k + 1) * bM[i + 1][k + 1] if inband(i + 1, k + 1) else Z
            ins = EI * bI[i + 1][k] if inband(i + 1, k) else Z
            dele = bD[i][k + 1] if inband(i, k + 1) else Z
            bM[i][k] = mMM * nxt + mMI * ins + mMD * dele
            bI[i][k] = mIM * nxt + mII * ins
            bD[i][k] = (mDM * nxt + mDD * dele) if i > 1 else Z  # the first row has no D state (as in the forward pass)
    zM = [[fM[i][k] * bM[i][k] / P if P else Z for k in range(R + 1)] for i in range(L + 1)]
    zI = [[fI[i][k] * bI[i][k] / P if P else Z for k in range(R + 1)] for i in range(L + 1)]
    return zM, zI, P, bw


def _problems():
    rng = np.random.default_rng(20241221)
    out = []
    for _ in range(230):
        L = int(rng.integers(1, 23))
        R = max(1, L + int(rng.integers(-3, 4)))
        ref = rng.integers(0, 4, R).astype(np.uint8)
        qry = np.resize(ref, L).copy() if rng.random() < 0.7 else rng.integers(0, 4, L).astype(np.uint8)
        m = rng.random(L) < 0.1
        qry[m] = (qry[m] + 1) % 4
        if rng.random() < 0.15:
            ref[rng.integers(0, R)] = 4
        if rng.random() < 0.15:
            qry[rng.integers(0, L)] = 4
        d = float(rng.choice([1e-4, 1e-3, 1e-2, 0.05]))
        e = float(rng.choice([0.1, 0.3, 0.02]))
        out.append((ref, qry, int(rng.choice([40, 20, 30, 13])), d, e, int(rng.integers(1, 9))))
    return out


def test_oracle_posteriors_equal_the_model_definition(built):
    checked_state = checked_q = 0
    for ref, qry, sq, d, e, bw in _problems():
        zM, zI, P, bw_eff = model_posteriors(ref, qry, sq, d, e, bw)
        s, oM, oI = orc.probaln_posteriors(ref, qry, sq, d, e, bw)
        L, R = len(qry), len(ref)
        # the oracle scales every row by s[i]: its products sum to 1/s[i], so z * s[i] is the posterior
        for i in range(1, L + 1):
            for k in range(1, R + 1):
                for exact, got in ((zM[i][k], oM[i - 1, k - 1] * s[i]), (zI[i][k], oI[i - 1, k - 1] * s[i])):
                    ex = float(exact)
                    assert abs(got - ex) <= 1e-12 * max(ex, 1e-300) + 1e-300, (i, k, got, ex, L, R, bw)
        # likelihood: product of the scaling factors
        like = 1.0
        for i in range(1, L + 2):
            like *= s[i]
        assert abs(like - float(P)) <= 1e-11 * float(P)
        # MAP state and phred, wherever the decision is not within rounding of a tie / a phred boundary
        pr, st, q = oracle_probaln(ref, qry, sq, d, e, bw)
        for i in range(1, L + 1):
            cells = [(zM[i][k], (k - 1) << 2) for k in range(1, R + 1)] + [(zI[i][k], ((k - 1) << 2) | 1) for k in range(1, R + 1)]
            tot = sum(c[0] for c in cells)
            if tot == 0:
                continue
            best = max(c[0] for c in cells)
            winners = [c for c in cells if c[0] == best]
            second = max([c[0] for c in cells if c[0] != best], default=Fraction(0))
            if len(winners) == 1 and float(second) < float(best) * (1 - 1e-9):
                assert st[i - 1] == winners[0][1], (i, st[i - 1], winners[0][1])
                checked_state += 1
            x = 1 - best / tot
            if x > 0:
                val = -4.343 * math.log(float(x)) + .499
                if abs(val - round(val)) > 1e-6:
                    kq = int(val)
                    assert q[i - 1] == (99 if kq > 100 else kq), (i, q[i - 1], kq)
                    checked_q += 1
        exp_pr = -4.343 * math.log(float(P) * R * L)
        assert abs(pr - (exp_pr + .499)) <= 1.0 + 1e-9  # (int) truncation of exp_pr + .499
    assert checked_state > 2000 and checked_q > 2000


def test_the_two_readings_of_the_terminal_guard_as_models(built):
    """The open line of probaln.c (include/spx.h SPX_GUARD_*), read as a MODEL: in the regime l_query <= bw and
    2*bw+1 > l_ref the BAND reading (the default) is the model above; the ROW reading is exactly the model in which an alignment may
    not end in the last reference column -- and where that ending is likely the two differ by far more than rounding.  Outside the
    regime both readings are the model above.  (Evidence for the default, not a pin: only a real htslib 1.17 decides.)"""
    rng = np.random.default_rng(3)
    seen = visible = 0

    def worst(zM, zI, s, oM, oI, L, R):
        return max(max(abs(float(zM[i][k]) - oM[i - 1, k - 1] * s[i]), abs(float(zI[i][k]) - oI[i - 1, k - 1] * s[i]))
                   for i in range(1, L + 1) for k in range(1, R + 1))
    try:
        for (L, R, bw_in) in ((9, 30, 25), (15, 15, 40), (1, 6, 10), (10, 14, 12), (8, 9, 30), (5, 16, 11), (7, 7, 7), (20, 24, 5), (12, 12, 3)):
            ref = rng.integers(0, 4, R).astype(np.uint8)
            qry = np.resize(ref, L).copy()
            if L <= R and (L + R) % 2 == 0:
                qry = ref[R - L:].copy()  # the query is the END of the window: its alignment ends in the last reference column
            mut = rng.random(L) < 0.1
            qry[mut] = (qry[mut] + 1) % 4
            fullM, fullI, _, bw = model_posteriors(ref, qry, 20, 1e-3, 0.1, bw_in)
            cutM, cutI, _, _ = model_posteriors(ref, qry, 20, 1e-3, 0.1, bw_in, last_column_may_end=False)
            reg = L <= bw and 2 * bw + 1 > R
            seen += reg
            orc.set_terminal_guard(orc.GUARD_BAND)
            s, oM, oI = orc.probaln_posteriors(ref, qry, 20, 1e-3, 0.1, bw_in)
            assert worst(fullM, fullI, s, oM, oI, L, R) < 1e-11, (L, R, bw_in)
            orc.set_terminal_guard(orc.GUARD_ROW)
            s, rM, rI = orc.probaln_posteriors(ref, qry, 20, 1e-3, 0.1, bw_in)
            if reg:
                assert worst(cutM, cutI, s, rM, rI, L, R) < 1e-11, (L, R, bw_in)
                end_mass = float(fullM[L][R] + fullI[L][R])  # posterior of "the last query base sits in the last reference column"
                if end_mass > 1e-3:
                    visible += 1
                    assert worst(fullM, fullI, s, rM, rI, L, R) > 1e-4, (L, R, bw_in, end_mass)
            else:
                assert worst(fullM, fullI, s, rM, rI, L, R) < 1e-11, (L, R, bw_in)
    finally:
        orc.set_terminal_guard(orc.GUARD_BAND)
    assert seen >= 5 and visible >= 2
