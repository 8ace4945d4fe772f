#!/usr/bin/env python3
"""Two-tier DP, step A (no GPU): the fast tier's CPU model (tools/fastdp/fastdp_model.c) beside the CPU oracle on the DP problems
the host plan produces for generator groups.  DESIGN STUDY / TEST INFRASTRUCTURE -- run by hand:

  python tools/fastdp_study.py --config hifi --problems 200000 [--guard band|row] [--threads 8]
  python tools/fastdp_study.py --config ont|mixed|fuzz ...

Reports, per configuration: problems / rows examined; rows flagged (by reason); UNFLAGGED rows whose (state, q) differ from the
oracle's (must be 0 -- over ALL rows of every problem, not only the wanted ones); problems flagged through their wanted rows (= the
share the exact tier re-runs); the largest deviation of the row-normalised posteriors on a sample, absolute and in units of the
certificate's delta.  Go / no-go rule of VERDICT r05 item 1: unflagged mismatches = 0 and flagged problems < 2 %."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


class StudyIn(C.Structure):
    _fields_ = [("n_problems", C.c_int32), ("L", C.c_void_p), ("R", C.c_void_p), ("bw", C.c_void_p), ("ref_tid", C.c_void_p),
                ("ref_rfs", C.c_void_p), ("qry_nib", C.c_void_p), ("qry4", C.c_void_p), ("hmm", C.c_void_p),
                ("row_off", C.c_void_p), ("n_rows_of", C.c_void_p), ("rows", C.c_void_p), ("bases", C.c_void_p),
                ("seq_off", C.c_void_p), ("set_q", C.c_int32), ("d", C.c_float), ("e", C.c_float), ("thr", C.c_void_p),
                ("threads", C.c_int32), ("dev_every", C.c_int32)]


class StudyOut(C.Structure):
    _fields_ = [("n_problems", C.c_int64), ("n_rows_all", C.c_int64), ("n_rows_wanted", C.c_int64),
                ("rows_flagged_all", C.c_int64), ("rows_flagged_wanted", C.c_int64), ("rows_by_reason", C.c_int64 * 8),
                ("unflagged_mismatch_all", C.c_int64), ("unflagged_mismatch_wanted", C.c_int64),
                ("flagged_but_equal_all", C.c_int64), ("problems_flagged_wanted", C.c_int64),
                ("problems_flagged_all", C.c_int64), ("problems_model", C.c_int64), ("dev_rows", C.c_int64),
                ("max_dev", C.c_double), ("max_dev_over_delta", C.c_double), ("x_hist", C.c_int64 * 20),
                ("first_bad_problem", C.c_int64)]


def load():
    import subprocess
    d = os.path.join(ROOT, "tools", "fastdp")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", d])
    L = C.CDLL(os.path.join(d, "libfastdp.so"))
    L.fdp_study.argtypes = [C.POINTER(StudyIn), C.POINTER(StudyOut)]
    return L


REASONS = ["argmax", "threshold", "x_small", "range", "model"]


def study_plan(L, plan, ref, params, thr, threads, dev_every):
    v = plan.view
    a = lambda p: C.cast(p, C.c_void_p)
    si = StudyIn(v.n_problems, a(v.L), a(v.R), a(v.bw), a(v.ref_tid), a(v.ref_rfs), a(v.qry_nib), a(v.qry4), a(v.hmm),
                 a(v.row_off), a(v.n_rows_of), a(v.rows), ref.contents.bases, a(ref.contents.seq_off), params.set_q,
                 params.conf_d, params.conf_e, C.cast(thr, C.c_void_p), threads, dev_every)
    so = StudyOut()
    L.fdp_study(C.byref(si), C.byref(so))
    return so


def add(tot, so):
    for f, _ in StudyOut._fields_:
        x = getattr(so, f)
        if f in ("max_dev", "max_dev_over_delta"):
            tot[f] = max(tot.get(f, 0.0), x)
        elif f == "first_bad_problem":
            continue
        elif hasattr(x, "__len__"):
            tot[f] = [p + q for p, q in zip(tot.get(f, [0] * len(x)), list(x))]
        else:
            tot[f] = tot.get(f, 0) + x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="hifi", choices=["hifi", "ont", "mixed", "fuzz"])
    ap.add_argument("--problems", type=int, default=200000)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--guard", default="band", choices=["band", "row"])
    ap.add_argument("--dev-every", type=int, default=200, help="posterior deviation on every n-th problem (0: off)")
    ap.add_argument("--chunk", type=int, default=0, help="groups per plan")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=0, help="fuzz: stop after this long instead of --problems")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    from oracle import orc
    from secphase_amd import api, records, synth
    Ls = load()
    g = 1 if args.guard == "row" else 0
    api.set_terminal_guard(g)
    orc.set_terminal_guard(g)
    thr = (C.c_double * 102)()
    mt = (C.c_double * 256)()
    mi = (C.c_double * 256)()
    api.lib().spx_host_tables(thr, mt, mi)

    tot = {}
    t0 = time.time()
    cases = 0
    if args.config == "fuzz":
        import fuzz
        rng = np.random.default_rng(args.seed)
        while (tot.get("n_problems", 0) < args.problems) if not args.seconds else (time.time() - t0 < args.seconds):
            plat, kw, par = fuzz.draw_case(rng)
            genome = synth.Genome(synth.default_cfg(plat, **kw))
            r = genome.reads(int(rng.integers(0, 1000)), int(rng.integers(2, 9)))
            try:
                plan = api.Plan(genome.ref, r.batch, par)
            except api.SpxError:
                continue
            if plan.view.n_problems:
                so = study_plan(Ls, plan, genome.ref, par, thr, args.threads, args.dev_every)
                add(tot, so)
                if so.unflagged_mismatch_all or so.unflagged_mismatch_wanted:
                    print("MISMATCH", fuzz.describe(plat, kw, par, 0, 0), "problem", so.first_bad_problem, flush=True)
            plan.close(); r.close(); genome.close()
            cases += 1
    else:
        plat = dict(hifi=synth.HIFI, ont=synth.ONT, mixed=synth.MIXED)[args.config]
        par = records.preset("ont", bandwidth=50) if args.config == "ont" else records.preset("hifi")
        genome = synth.Genome(synth.default_cfg(plat))
        chunk = args.chunk or dict(hifi=2048, ont=128, mixed=512)[args.config]
        first = 0
        while tot.get("n_problems", 0) < args.problems:
            r = genome.reads(first, chunk)
            plan = api.Plan(genome.ref, r.batch, par)
            so = study_plan(Ls, plan, genome.ref, par, thr, args.threads, args.dev_every)
            add(tot, so)
            if so.unflagged_mismatch_all or so.unflagged_mismatch_wanted:
                print("MISMATCH groups", first, chunk, "problem", so.first_bad_problem, flush=True)
            plan.close(); r.close()
            first += chunk
            cases += 1
            print(f"  {tot['n_problems']} problems, {tot['n_rows_all']} rows, unflagged mismatches {tot['unflagged_mismatch_all']}, "
                  f"flagged problems {tot['problems_flagged_wanted']}, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    res = dict(config=args.config, guard=args.guard, cases=cases, seconds=round(time.time() - t0, 1), **tot)
    res["rows_by_reason"] = dict(zip(REASONS, tot.get("rows_by_reason", [0] * 8)[:5]))
    n = max(1, tot.get("n_problems", 0))
    res["flagged_problem_fraction_wanted_rows"] = tot.get("problems_flagged_wanted", 0) / n
    res["flagged_problem_fraction_all_rows"] = tot.get("problems_flagged_all", 0) / n
    res["flagged_row_fraction_all_rows"] = tot.get("rows_flagged_all", 0) / max(1, tot.get("n_rows_all", 0))
    res["go"] = bool(tot.get("unflagged_mismatch_all", 1) == 0 and tot.get("unflagged_mismatch_wanted", 1) == 0
                     and res["flagged_problem_fraction_wanted_rows"] < 0.02)
    s = json.dumps(res)
    print(s)
    if args.out:
        with open(args.out, "a") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
