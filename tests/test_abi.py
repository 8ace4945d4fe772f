"""CPU: the C-ABI library loads, exports every symbol include/spx.h declares, refuses to run
without a device (no CPU fallback), and its host-only entry points behave like the reference tail."""
import ctypes as C
import filecmp
import os
import re

import pytest
import torch

from common import small_genome
from oracle import orc
from secphase_amd import api, records, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(built):
    hdr = open(os.path.join(ROOT, "include", "spx.h")).read()
    declared = set(re.findall(r"\b(spx_[a-z_0-9]+)\s*\(", hdr))
    L = api.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(api.EXPORTS)


def test_no_device_means_error_not_fallback(built):
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = api.lib().spx_create(0, C.byref(h))
    assert rc == api.ENODEVICE and not h.value
    with pytest.raises(api.SpxError):
        api.Context(0)
    # the htslib-compatible single-problem symbol reports failure the way the reference checks for it
    import numpy as np
    r = np.zeros(10, np.uint8)
    st = np.zeros(10, np.int32)
    q = np.zeros(10, np.uint8)
    par = api.ProbalnPar(1e-4, 0.1, 5)
    u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    INT_MIN = -2 ** 31
    assert api.lib().spx_probaln_glocal(u8(r), 10, u8(r), 10, None, C.byref(par), st.ctypes.data_as(C.POINTER(C.c_int)),
                                        u8(q)) == INT_MIN


def test_finalize_and_relabel_log_reproduce_the_reference_tail(built, tmp_path):
    """feed scores (from the oracle) through spx_finalize + spx_write_relabel_log: the rand()
    replay and the text format must equal the oracle's own tail."""
    g = small_genome(synth.HIFI, max_secondaries=4, n_paralogs=3)
    r = g.reads(0, 60)
    p = records.preset("hifi")
    log_o = str(tmp_path / "o.log")
    nre, res = orc.run_batch(r.batch, g.ref, p, threads=2, seed=1, log_path=log_o)
    n = r.batch.contents.n_groups
    out = (api.GroupOut * n)()
    for i in range(n):
        e = res[i]
        o = out[i]
        o.n_aln = e.n_aln
        sec = [a for a in range(e.n_aln) if a != e.prim_idx]
        mx, mxs = -1, -1.7976931348623157e308
        for a in sec:
            if mxs < e.score[a]:
                mx, mxs = a, e.score[a]
        for a in range(e.n_aln):
            o.score[a] = e.score[a]
            o.rfe[a] = e.rfe[a]
        o.prim_idx = e.prim_idx
        o.max_idx = mx
        o.tie_mask = sum(1 << a for a in sec if mxs <= e.score[a])
        o.pass_ = int(not (mxs <= e.score[e.prim_idx] + p.prim_margin_score or mxs < p.min_score))
    assert api.lib().spx_finalize(C.byref(p), 1, out, n) == 0
    for i in range(n):
        assert out[i].best_idx == res[i].best_idx and bool(out[i].relabel) == bool(res[i].relabel)
    log_g = str(tmp_path / "g.log")
    api.write_relabel_log(log_g, r.batch, g.ref, out)
    assert filecmp.cmp(log_o, log_g, shallow=False)
    assert sum(o.relabel for o in out) == nre > 0


def test_effective_cpus_respects_affinity_and_quota(built):
    """default of every host_threads argument: online CPUs cut by the affinity mask and the cgroup CPU quota"""
    import os
    from secphase_amd import api
    n = api.lib().spx_effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert n <= len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = -(-int(q) // int(per))
    except OSError:
        pass
    if quota:
        assert n <= quota
