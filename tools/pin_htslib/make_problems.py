#!/usr/bin/env python3
"""problems.txt for pin_probaln: the six problems of tests/golden/probaln.json + 1 200 seeded random ones (HiFi- and
ONT-shaped windows, small and degenerate shapes, ambiguous bases, odd parameters).  Deterministic: the same file on
every machine."""
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
    return out


if __name__ == "__main__":
    for ref, q, bw, d, e, sq in problems():
        sys.stdout.write("%d %d %d %.10g %.10g %d %s %s\n" % (len(ref), len(q), bw, d, e, sq, "".join(map(str, ref)), "".join(map(str, q))))
