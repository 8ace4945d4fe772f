#!/usr/bin/env python3
"""problems.txt + answers.txt (from pin_probaln, i.e. from the real htslib) -> tests/golden/htslib_probaln_vectors.json.
Once that file is committed, tests/test_htslib_pin.py checks the oracle (CPU) and the kernels (-m gpu) against it and the
oracle stops being "parity unpinned" for probaln_glocal."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main(problems_path, answers_path, version):
    vec = []
    for pl, al in zip(open(problems_path), open(answers_path)):
        f = pl.split()
        a = [int(x) for x in al.split()]
        lq = int(f[1])
        assert len(a) == 1 + 2 * lq, "answer line does not match its problem"
        vec.append(dict(ref=[int(c) for c in f[6]], query=[int(c) for c in f[7]], bw=int(f[2]), d=float(f[3]), e=float(f[4]),
                        set_q=int(f[5]), Pr=a[0], state=a[1:1 + lq], q=a[1 + lq:]))
    out = os.path.join(ROOT, "tests", "golden", "htslib_probaln_vectors.json")
    json.dump(dict(source=f"htslib {version} probaln_glocal via tools/pin_htslib/pin_probaln.c", vectors=vec), open(out, "w"))
    print(f"wrote {out}: {len(vec)} vectors")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "unknown")
