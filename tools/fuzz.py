#!/usr/bin/env python3
"""Randomised parity campaign (not part of the test-suite: run by hand, minutes to hours).

  python tools/fuzz.py cpu  --seconds 600 [--seed S]   host logic (libspx plan + oracle DP) vs oracle, no GPU
  python tools/fuzz.py gpu  --seconds 300 [--seed S]   HIP path vs oracle: random DP problems and random batches

Every case draws its own generator configuration (platform, read length, clipping, number of secondaries,
ambiguous bases, divergence, cs/MD tags) and its own scoring parameters (gap open/extension, band width, initial
quality, thresholds, margins, -q/-c on or off).  The first mismatch stops the run and prints the recipe."""
import argparse
import copy
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def draw_case(rng):
    from secphase_amd import records, synth
    plat = int(rng.choice([synth.HIFI, synth.ONT, synth.MIXED], p=[0.5, 0.3, 0.2]))
    kw = dict(seed=int(rng.integers(1, 2 ** 40)), n_contigs=int(rng.integers(1, 4)), contig_len=int(rng.choice([60000, 150000, 300000])),
              n_paralogs=int(rng.integers(1, 6)), softclip_frac=float(rng.choice([0, 0.2, 0.6])),
              hardclip_frac=float(rng.choice([0, 0.3, 1.0])), shuffle_records=int(rng.integers(0, 2)),
              inverted_paralogs=int(rng.integers(0, 2)), n_base_frac=float(rng.choice([0, 0, 1e-3, 1e-2])),
              snv_rate=float(rng.choice([1 / 5000, 1 / 500, 1 / 100])), indel_rate=float(rng.choice([1 / 50000, 1 / 2000, 1 / 300])),
              paralog_snv_rate=float(rng.choice([0.002, 0.01, 0.04])), tag_mode=int(rng.choice([0, 0, 1, 2])))
    if rng.random() < 0.7:
        kw["read_len"] = int(rng.choice([300, 1000, 2500, 6000, 12000]))
    else:
        kw["read_len"] = 0
        kw["max_read_len"] = int(rng.choice([5000, 20000, 40000]))
    mx = int(rng.integers(1, 7))
    kw["max_secondaries"] = mx
    kw["min_secondaries"] = int(rng.integers(0, mx + 1))
    kw["n_paralogs"] = max(kw["n_paralogs"], mx)
    par = records.preset("ont" if plat == synth.ONT else "hifi")
    if rng.random() < 0.6:
        par.conf_d = float(rng.choice([1e-6, 1e-4, 1e-3, 1e-2, 0.05]))
        par.conf_e = float(rng.choice([0.01, 0.1, 0.3, 0.5]))
        par.conf_b = float(rng.choice([1, 5, 20, 33, 50, 70]))
        par.set_q = int(rng.choice([5, 20, 27, 40, 60, 93]))
        par.min_q = int(rng.choice([0, 5, 10, 20]))
        par.indel_threshold = int(rng.choice([1, 4, 10, 20, 50]))
        par.flank_margin = int(rng.choice([30, 100, 300, 500, 1200]))
        par.prim_margin_score = float(rng.choice([0, 5, 20, 40]))
        par.prim_margin_random = float(rng.choice([0, 0, 3, 50]))
        par.min_score = int(rng.choice([-100, -10, 0]))
        par.baq_flag = int(rng.random() < 0.85)
        par.consensus = int(rng.random() < 0.8)
    return plat, kw, par


def describe(plat, kw, par, first, n):
    fields = {f[0]: getattr(par, f[0]) for f in par._fields_}
    return f"platform={plat} cfg={kw} params={fields} groups=[{first},{first + n})"


def cpu_case(rng):
    from common import batch_qual_copy, emulate_plan, emulate_rows, replay_qual_edits
    from oracle import orc
    from secphase_amd import api, synth
    plat, kw, par = draw_case(rng)
    if not par.consensus:
        kw["read_len"] = min(kw.get("read_len") or 2500, 2500)  # whole-block windows: keep the Python emulation short
        kw.pop("max_read_len", None)
    g = synth.Genome(synth.default_cfg(plat, **kw))
    first, n = int(rng.integers(0, 1000)), int(rng.integers(2, 9))
    r = g.reads(first, n)
    what = describe(plat, kw, par, first, n)
    _, res = orc.run_batch(r.batch, g.ref, par, threads=2, seed=1)
    plan = api.Plan(g.ref, r.batch, par)
    v = plan.view
    em = emulate_plan(plan, g.ref, par)
    idx = list(v.grp_index[:v.n_groups])
    for k in range(n):
        e = res[k]
        disp = orc.lib().orc_group_is_dispatched(r.batch, k)
        assert bool(disp) == bool(api.lib().spx_group_is_dispatched(r.batch, k)), what
        if e.n_aln < 0:
            assert v.grp_error[k] < 0, what
            continue
        if not disp:
            assert k not in em, what
            continue
        assert v.grp_error[k] == 0, (what, k, v.grp_error[k])
        sc, prim, mx, tie, ok = em[k]
        assert [e.score[a] for a in range(e.n_aln)] == sc, (what, k)
        assert e.prim_idx == prim, (what, k)
        assert [v.rfe[10 * idx.index(k) + a] for a in range(e.n_aln)] == [e.rfe[a] for a in range(e.n_aln)], (what, k)
    assert v.n_problems == sum(x.n_baq_calls for x in res if x.n_aln > 0), what
    if par.baq_flag and rng.random() < 0.3 and all(res[k].n_aln >= 0 for k in range(n)):
        p_all = copy.copy(par)
        p_all.flags = 1
        want, _ = orc.run_batch_quals(r.batch, g.ref, par, batch_qual_copy(r.batch), threads=2)
        plan2 = api.Plan(g.ref, r.batch, p_all)
        got = replay_qual_edits(plan2, emulate_rows(plan2, g.ref, p_all), r.batch, p_all)
        assert np.array_equal(got, want), what + " (quality-modified records)"
    return v.n_problems


def gpu_problems(ctx, rng, n):
    from common import oracle_probaln
    probs, sq, pars = [], [], []
    for _ in range(n):
        L = int(rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 40, 41, 63, 64, 65, 100, 250, 500, 999, 1000, 1200]))
        if rng.random() < 0.5:
            L = int(rng.integers(1, 1300))
        R = max(1, L + int(rng.choice([0, 0, 0, 1, -1, 2, -2, 3, -5, 9, -13, 30, -40, 120])))
        mode = rng.random()
        ref = rng.integers(0, 4, size=R).astype(np.uint8)
        if mode < 0.6:  # related sequences
            q = np.resize(ref, L).copy()
            m = rng.random(L) < rng.choice([0.001, 0.01, 0.1])
            q[m] = (q[m] + 1 + rng.integers(0, 3, size=int(m.sum()))) % 4
        elif mode < 0.8:  # unrelated
            q = rng.integers(0, 4, size=L).astype(np.uint8)
        else:  # homopolymers / repeats
            ref[:] = rng.integers(0, 4)
            q = np.full(L, ref[0], np.uint8)
            if L > 3:
                q[rng.integers(0, L)] = (ref[0] + 1) % 4
        if rng.random() < 0.15:
            ref = ref.copy()
            ref[rng.random(R) < 0.03] = 4
            q[rng.random(L) < 0.03] = 4
        bw_in = abs(R - L) + int(rng.choice([1, 2, 5, 10, 20, 21, 22, 23, 24, 30, 50, 51, 63, 64, 100, 127, 255, 300]))
        probs.append((ref, q.astype(np.uint8)))
        sq.append(int(rng.choice([1, 10, 20, 40, 60, 93])))
        pars.append((float(rng.choice([1e-6, 1e-4, 1e-3, 1e-2, 0.1])), float(rng.choice([0.01, 0.1, 0.3, 0.5])), bw_in))
    st, qq, _ = ctx.probaln_batch([p[0] for p in probs], [p[1] for p in probs], sq, pars)
    from oracle import orc
    for i in rng.choice(n, size=min(n, 24), replace=False):  # the kernels' own numbers, bit for bit
        r, q = probs[i]
        if len(r) * len(q) > 400000:
            continue
        sc, zM, zI = ctx.probaln_posteriors([p[0] for p in probs], [p[1] for p in probs], sq, pars, which=int(i))
        s, oM, oI = orc.probaln_posteriors(r, q, sq[i], *pars[i])
        L = len(q)
        what = f"L={L} R={len(r)} set_q={sq[i]} d,e,bw={pars[i]} ref={r.tolist()} qry={q.tolist()}"
        with np.errstate(divide="ignore", invalid="ignore"):
            assert np.array_equal(sc[1:L], 1.0 / s[1:L], equal_nan=True), "1/s differs: " + what
        assert np.array_equal(zM, oM, equal_nan=True) and np.array_equal(zI, oI, equal_nan=True), "z differs: " + what
    for i, (r, q) in enumerate(probs):
        _, est, eq = oracle_probaln(r, q, sq[i], pars[i][0], pars[i][1], pars[i][2])
        what = f"L={len(q)} R={len(r)} set_q={sq[i]} d,e,bw={pars[i]} ref={r.tolist()} qry={q.tolist()}"
        assert np.array_equal(st[i], est), "state differs: " + what
        assert np.array_equal(qq[i], eq), "q differs: " + what
    return n


def gpu_batch(ctx, rng):
    from common import batch_qual_copy
    from oracle import orc
    from secphase_amd import synth
    plat, kw, par = draw_case(rng)
    g = synth.Genome(synth.default_cfg(plat, **kw))
    first, n = int(rng.integers(0, 5000)), int(rng.integers(8, 64))
    r = g.reads(first, n)
    what = describe(plat, kw, par, first, n)
    ctx.set_reference(g.ref)
    allrows = par.baq_flag and rng.random() < 0.25
    p_run = copy.copy(par)
    if allrows:
        p_run.flags = 1
    w = ctx.prepare(r.batch, p_run)
    w.launch()
    out = w.collect(finalize_seed=1)
    _, res = orc.run_batch(r.batch, g.ref, par, threads=8, seed=1)
    for k in range(n):
        o, e = out[k], res[k]
        assert o.n_aln == e.n_aln, (what, k, o.n_aln, e.n_aln)
        for a in range(max(e.n_aln, 0)):
            assert o.score[a] == e.score[a], (what, k, a, o.score[a], e.score[a])
        assert o.best_idx == e.best_idx and bool(o.relabel) == bool(e.relabel), (what, k)
    if allrows and all(res[k].n_aln >= 0 for k in range(n)):
        want, _ = orc.run_batch_quals(r.batch, g.ref, par, batch_qual_copy(r.batch), threads=8)
        got = w.apply_quals(r.batch, batch_qual_copy(r.batch))
        assert np.array_equal(got, want), what + " (quality-modified records)"
    st = w.stats()
    w.free()
    return st.n_problems


def gpu_pipe(ctx, rng):
    """several batches of different sizes through the in-order pipeline (spx_pipe): work lists of different shapes share
    the context's recycled arenas and pools, submissions may hold more than one record block"""
    from oracle import orc
    from secphase_amd import api, synth
    plat, kw, par = draw_case(rng)
    kw["read_len"] = min(kw.get("read_len") or 6000, 6000)
    kw.pop("max_read_len", None)
    g = synth.Genome(synth.default_cfg(plat, **kw))
    ctx.set_reference(g.ref)
    what = describe(plat, kw, par, 0, 0)
    subs = []
    first = int(rng.integers(0, 3000))
    for _ in range(int(rng.integers(3, 8))):
        blocks = []
        for _ in range(int(rng.choice([1, 1, 2, 3]))):
            n = int(rng.choice([1, 2, 5, 17, 40, 96]))
            blocks.append(g.reads(first, n))
            first += n
        subs.append(blocks)
    depth = int(rng.integers(1, 4))
    pipe = api.Pipe(ctx, par, depth=depth, host_threads=4)
    nprob = 0
    try:
        sent = got = 0
        while got < len(subs):
            while sent < len(subs) and pipe.pending() < depth + 1:
                pipe.submit(batch=[b.batch for b in subs[sent]])
                sent += 1
            out, n = pipe.next()
            base = 0
            for b in subs[got]:
                _, res = orc.run_batch(b.batch, g.ref, par, threads=8, seed=1)
                for k in range(b.batch.contents.n_groups):
                    o, e = out[base + k], res[k]
                    assert o.n_aln == e.n_aln, (what, "pipe", got, k, o.n_aln, e.n_aln)
                    for a in range(max(e.n_aln, 0)):
                        assert o.score[a] == e.score[a], (what, "pipe", got, k, a, o.score[a], e.score[a])
                    nprob += max(e.n_baq_calls, 0) if e.n_aln > 0 else 0
                base += b.batch.contents.n_groups
            assert base == n, (what, "pipe", got, base, n)
            got += 1
    finally:
        pipe.close()
    return nprob


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["cpu", "gpu"])
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    try:
        run(args, rng)
    except AssertionError as ex:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{args.mode}_{args.seed}.txt"), "w") as f:
            f.write(repr(ex.args))
        raise


def run(args, rng):
    import __graft_entry__ as ge
    ge.build_cpu_helpers()
    t0, cases, problems = time.time(), 0, 0
    if args.mode == "cpu":
        while time.time() - t0 < args.seconds:
            problems += cpu_case(rng)
            cases += 1
    else:
        from secphase_amd import api
        ctx = api.Context(0)
        while time.time() - t0 < args.seconds:
            problems += gpu_problems(ctx, rng, 96)
            problems += gpu_batch(ctx, rng)
            cases += 2
            if cases % 8 == 0:
                problems += gpu_pipe(ctx, rng)
                cases += 1
    print(f"fuzz {args.mode}: seed {args.seed}, {cases} cases, {problems} DP problems, no mismatch, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
