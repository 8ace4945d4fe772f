#!/usr/bin/env python3
"""problems.txt + answers.txt (from pin_probaln, i.e. from the real htslib) -> tests/golden/htslib_probaln_vectors.json.
Once that file is committed, tests/test_htslib_pin.py checks the oracle (CPU) and the kernels (-m gpu) against it and the
oracle stops being "parity unpinned" for probaln_glocal."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main(problems_path, answers_path, version, out=None):
    vec = []
    for pl, al in zip(open(problems_path), open(answers_path)):
        f = pl.split()
        a = [int(x) for x in al.split()]
        lq = int(f[1])
        assert len(a) == 1 + 2 * lq, "answer line does not match its problem"
        vec.append(dict(ref=[int(c) for c in f[6]], query=[int(c) for c in f[7]], bw=int(f[2]), d=float(f[3]), e=float(f[4]),
                        set_q=int(f[5]), Pr=a[0], state=a[1:1 + lq], q=a[1 + lq:]))
    guard, report = which_guard(vec)
    out = out or os.path.join(ROOT, "tests", "golden", "htslib_probaln_vectors.json")
    json.dump(dict(source=f"htslib {version} probaln_glocal via tools/pin_htslib/pin_probaln.c", terminal_guard=guard,
                   guard_report=report, vectors=vec), open(out, "w"))
    print(f"wrote {out}: {len(vec)} vectors")
    print(f"terminal guard (include/spx.h SPX_GUARD_*): this htslib matches the reading '{guard}'  {report}")
    if guard == "row":
        print("  -> the repository's default is 'band': flip the default in secphase_amd/csrc/spx_prep.cpp terminal_guard() and "
              "oracle/probaln_oracle.c orc_get_terminal_guard() (one constant each), or export SPX_TERMINAL_GUARD=row")
    elif guard == "neither":
        print("  -> neither reading reproduces this htslib on the regime block: the restatement differs elsewhere; see the first "
              "failing vector of tests/test_htslib_pin.py")


def which_guard(vec):
    """runs the oracle under both readings over the vectors of the regime (l_query <= bw, 2*bw+1 > l_ref) and reports which one
    reproduces the htslib answers; needs oracle/liborc.so (make -C oracle)"""
    import ctypes as C
    sys.path.insert(0, ROOT)
    from oracle import orc
    import numpy as np
    L = orc.lib()
    match = {}
    regime = []
    for v in vec:
        lq, lr = len(v["query"]), len(v["ref"])
        bw = max(min(max(lq, lr), v["bw"]), abs(lr - lq))
        if lq <= bw and 2 * bw + 1 > lr:
            regime.append(v)
    for name, reading in (("band", 0), ("row", 1)):
        orc.set_terminal_guard(reading)
        ok = 0
        for v in regime:
            ref = np.array(v["ref"], np.uint8)
            qry = np.array(v["query"], np.uint8)
            iq = np.full(len(qry), v["set_q"], np.uint8)
            st = np.zeros(len(qry), np.int32)
            q = np.zeros(len(qry), np.uint8)
            par = orc.ProbalnPar(v["d"], v["e"], v["bw"])
            u8 = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint8))
            pr = L.orc_probaln_glocal(u8(ref), len(ref), u8(qry), len(qry), u8(iq), C.byref(par), st.ctypes.data_as(C.POINTER(C.c_int)), u8(q))
            ok += pr == v["Pr"] and st.tolist() == v["state"] and q.tolist() == v["q"]
        match[name] = ok
    orc.set_terminal_guard(0)
    n = len(regime)
    report = dict(regime_vectors=n, band_matches=match["band"], row_matches=match["row"])
    if match["band"] == n and match["row"] < n:
        return "band", report
    if match["row"] == n and match["band"] < n:
        return "row", report
    if match["band"] == n and match["row"] == n:
        return "band", report  # (the block cannot tell them apart: should not happen, every problem of it is in the regime)
    return "neither", report


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "unknown", sys.argv[4] if len(sys.argv) > 4 else None)
