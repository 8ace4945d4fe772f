"""Targeted GPU fuzz of spx_probaln_glocal in the overflow regime (unrelated / half-related / shifted long sequences, wide and narrow bands): states, qualities and sampled posteriors against the oracle.  Usage: python tools/fuzz_overflow.py [seed] [seconds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import oracle_probaln
from oracle import orc
from secphase_amd import api
ctx = api.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t0 = time.time(); n_cases = 0; n_nan = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 120:
    probs, sq, pars = [], [], []
    for _ in range(24):
        L = int(rng.integers(300, 1600)); R = max(1, L + int(rng.choice([0, 1, -1, 5, -13, 30, -40, 120, -200])))
        ref = rng.integers(0, 4, size=R).astype(np.uint8)
        mode = rng.random()
        if mode < 0.5: q = rng.integers(0, 4, size=L).astype(np.uint8)                         # unrelated
        elif mode < 0.8:                                                                         # related up to a point, then unrelated
            q = np.resize(ref, L).copy(); cut = int(rng.integers(0, L)); q[cut:] = rng.integers(0, 4, size=L - cut)
        else:                                                                                    # shifted copy (the band misses the diagonal)
            sh = int(rng.integers(50, 400)); q = np.resize(np.roll(ref, sh), L).copy()
        if rng.random() < 0.2: ref[rng.random(R) < 0.02] = 4
        bw_in = abs(R - L) + int(rng.choice([1, 5, 10, 20, 21, 23, 30, 50, 51, 64, 100, 127, 255, 300]))
        probs.append((ref, q)); sq.append(int(rng.choice([1, 10, 20, 40, 60, 93])))
        pars.append((float(rng.choice([1e-6, 1e-4, 1e-3, 1e-2, 0.1])), float(rng.choice([0.01, 0.1, 0.3, 0.5])), bw_in))
    st, qq, _ = ctx.probaln_batch([p[0] for p in probs], [p[1] for p in probs], sq, pars)
    for i, (r, q) in enumerate(probs):
        _, est, eq = oracle_probaln(r, q, sq[i], *pars[i])
        what = f"L={len(q)} R={len(r)} set_q={sq[i]} d,e,bw={pars[i]}"
        assert np.array_equal(st[i], est), "state differs: " + what + f" rows {np.nonzero(np.asarray(st[i]) != np.asarray(est))[0][:8]}"
        assert np.array_equal(qq[i], eq), "q differs: " + what
        n_cases += 1
    i = int(rng.integers(0, len(probs)))
    sc, zM, zI = ctx.probaln_posteriors([p[0] for p in probs], [p[1] for p in probs], sq, pars, which=i)
    s, oM, oI = orc.probaln_posteriors(probs[i][0], probs[i][1], sq[i], *pars[i])
    n_nan += int(np.isnan(np.asarray(oM)).any())
    assert np.array_equal(zM, oM, equal_nan=True) and np.array_equal(zI, oI, equal_nan=True), "z differs"
print(f"overflow fuzz: {n_cases} problems, {n_nan} sampled posteriors with NaNs, no mismatch, {time.time() - t0:.0f} s")
