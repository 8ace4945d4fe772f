#!/usr/bin/env python3
"""problems.txt for pin_probaln: the six problems of tests/golden/probaln.json + 1 200 seeded random ones (HiFi- and
ONT-shaped windows, small and degenerate shapes, ambiguous bases, odd parameters) + 240 in the regime where two readings of
the terminal / backward-start guard differ (l_query <= bw and 2*bw+1 > l_ref).  Deterministic: the same file on every
machine."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def problems():
    out = []
    for v in json.load(open(os.path.join(ROOT, "tests", "golden", "probaln.json")))["vectors"]:
        out.append((v["ref"], v["query"], v["bw"], v["d"], v["e"], v["set_q"]))
    rng = np.random.default_rng(1717)

    def mutate(ref, L, sub, indel):
        q, i = [], 0
        while i < len(ref) and len(q) < L:
            u = rng.random()
            if u < indel / 2:
                i += 1
                continue
            if u < indel:
                q.append(int(rng.integers(0, 4)))
                continue
            b = int(ref[i])
            if rng.random() < sub:
                b = (b + 1 + int(rng.integers(0, 3))) % 4
            q.append(b)
            i += 1
        return ref[:max(1, i)], q

    for k in range(1200):
        kind = k % 4
        if kind == 0:      # HiFi window
            L, sub, indel, bw, d, sq = int(rng.integers(300, 1001)), 0.003, 0.002, 20, 1e-4, 40
        elif kind == 1:    # ONT window
            L, sub, indel, bw, d, sq = int(rng.integers(200, 900)), 0.02, 0.04, 50, 1e-3, 20
        elif kind == 2:    # small / degenerate
            L, sub, indel, bw, d, sq = int(rng.integers(1, 60)), 0.05, 0.05, int(rng.integers(1, 25)), 1e-4, 30
        else:              # odd parameters, ambiguous bases
            L, sub, indel, bw, d, sq = int(rng.integers(50, 400)), 0.1, 0.05, int(rng.choice([5, 33, 70, 130])), \
                float(rng.choice([1e-6, 1e-2, 0.05])), int(rng.choice([5, 13, 27, 60, 93]))
        ref = rng.integers(0, 4, L + 40).tolist()
        ref, q = mutate(ref, L, sub, indel)
        if kind == 3:
            ref = [4 if rng.random() < 0.02 else b for b in ref]
            q = [4 if rng.random() < 0.02 else b for b in q]
        e = 0.1 if kind < 3 else float(rng.choice([0.1, 0.3, 0.5]))
        out.append((ref, q, abs(len(ref) - len(q)) + bw, d, e, sq))
    # ---- block of its own: l_query <= bw AND 2*bw + 1 > l_ref.  There the band covers the whole reference on every row
    # (set_u's row offset is 0), and two readings of probaln.c differ: the terminal sum / backward start skip a column with
    # `u >= bw2*3+3` (what oracle/probaln_oracle.c follows: never true here, column l_ref takes part) or with
    # `u >= i_dim-3` (i_dim = 3*l_ref+6 when 2*bw+1 >= l_ref: column l_ref would be skipped).  secphase reaches the regime
    # with --ont -b 50 on consensus blocks of 21-50 bases (ptMarker.c:754: bw = |l_ref - l_query| + conf_bw).
    rng2 = np.random.default_rng(2718)
    for k in range(240):
        bw_in = int(rng2.choice([20, 50, 50, 70]))
        L = int(rng2.integers(1, bw_in + 1))               # l_query <= conf_bw <= bw
        R = max(1, L + int(rng2.integers(-min(L - 1, 6), 7)))  # l_ref within a few bases: 2*bw+1 > l_ref
        ref = rng2.integers(0, 4, R).tolist()
        q = [ref[min(i, R - 1)] if rng2.random() > 0.08 else int(rng2.integers(0, 4)) for i in range(L)]
        d, sq = (1e-3, 20) if bw_in >= 50 else (1e-4, 40)
        bw = abs(R - L) + bw_in
        assert L <= bw and 2 * bw + 1 > R
        out.append((ref, q, bw, d, 0.1, sq))
    return out


if __name__ == "__main__":
    for ref, q, bw, d, e, sq in problems():
        sys.stdout.write("%d %d %d %.10g %.10g %d %s %s\n" % (len(ref), len(q), bw, d, e, sq, "".join(map(str, ref)), "".join(map(str, q))))
