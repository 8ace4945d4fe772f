"""The pin against the REAL htslib: consumes tests/golden/htslib_probaln_vectors.json when it exists (it is produced by
tools/pin_htslib/run.sh on a machine that has htslib 1.17 -- this repository's container does not) and checks
state[], q[] and the returned likelihood of every vector against the oracle (CPU) and the HIP kernels (-m gpu).
Until the file is committed both tests skip and the oracle stays "parity unpinned" for probaln_glocal."""
import json
import os

import numpy as np
import pytest

from common import oracle_probaln

VECTORS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "htslib_probaln_vectors.json")
needs_vectors = pytest.mark.skipif(not os.path.exists(VECTORS), reason="no htslib vectors yet: run tools/pin_htslib/run.sh HTSLIB_PREFIX")


def _vectors():
    return json.load(open(VECTORS))["vectors"]


def _pinned_guard():
    """the reading of the terminal guard the real htslib matched (tools/pin_htslib/make_vectors.py which_guard)"""
    return {"band": 0, "row": 1}.get(json.load(open(VECTORS)).get("terminal_guard", "band"), 0)


@pytest.fixture
def pinned_guard(built):
    from oracle import orc
    from secphase_amd import api
    g = _pinned_guard()
    orc.set_terminal_guard(g)
    api.set_terminal_guard(g)
    yield g
    orc.set_terminal_guard(0)
    api.set_terminal_guard(0)


@needs_vectors
def test_default_guard_is_the_pinned_one(built):
    """fails when the real htslib matched the OTHER reading than the repository's default: flip the default (one constant in
    secphase_amd/csrc/spx_prep.cpp terminal_guard() and in oracle/probaln_oracle.c orc_get_terminal_guard())"""
    from oracle import orc
    from secphase_amd import api
    assert json.load(open(VECTORS)).get("terminal_guard") in ("band", "row"), "neither reading reproduced htslib on the regime block"
    assert api.get_terminal_guard() == _pinned_guard() == orc.get_terminal_guard()


@needs_vectors
def test_oracle_equals_htslib(pinned_guard):
    for k, v in enumerate(_vectors()):
        pr, st, q = oracle_probaln(np.array(v["ref"], np.uint8), np.array(v["query"], np.uint8), v["set_q"], v["d"], v["e"], v["bw"])
        assert pr == v["Pr"] and st.tolist() == v["state"] and q.tolist() == v["q"], k


@needs_vectors
@pytest.mark.gpu
def test_kernels_equal_htslib(pinned_guard):
    from secphase_amd import api
    ctx = api.Context(0)
    vec = _vectors()
    for lo in range(0, len(vec), 128):
        part = vec[lo:lo + 128]
        st, qq, _ = ctx.probaln_batch([np.array(v["ref"], np.uint8) for v in part], [np.array(v["query"], np.uint8) for v in part],
                                      [v["set_q"] for v in part], [(v["d"], v["e"], v["bw"]) for v in part])
        for k, v in enumerate(part):
            assert st[k].tolist() == v["state"] and qq[k].tolist() == v["q"], lo + k
    ctx.close()


def test_problem_list_is_deterministic():
    """the problem file the harness feeds to htslib is a pure function of the repository"""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pin_htslib", "make_problems.py")
    a = subprocess.run([sys.executable, tool], capture_output=True, text=True, check=True).stdout
    b = subprocess.run([sys.executable, tool], capture_output=True, text=True, check=True).stdout
    assert a == b and len(a.splitlines()) == 1446
    f = a.splitlines()[0].split()
    assert len(f) == 8 and len(f[6]) == int(f[0]) and len(f[7]) == int(f[1])
    # the last 240: the regime in which the two readings of the terminal guard differ (oracle/probaln_oracle.c, "guard variants")
    for ln in a.splitlines()[-240:]:
        r_, l_, bw_ = (int(x) for x in ln.split()[:3])
        assert l_ <= bw_ and 2 * bw_ + 1 > r_
