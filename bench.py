#!/usr/bin/env python3
"""bench.py -- secphase hot path (markers -> consensus windows -> banded-HMM BAQ -> marker-consistency score ->
decision -> relabel list) on N MI355X GPUs of one node.

A "step" is one pass of the WHOLE hot path over one batch of synthetic alignment groups whose RECORDS (flag / position /
CIGAR / cs / SEQ / QUAL, the bytes a BAM reader produces) are already resident in HBM when the timed region starts:
  device:  CIGAR+cs walk, markers, consensus windows, DP work list (spx_prep_kernels.hip)  ->  banded-HMM forward /
           backward / MAP kernels  ->  marker filter, score, decision kernels  ->  one packed result record per group
  host:    one D2H copy of those records, the rand() replay of the decision in file order, the relabel list appended
           to out.log;  N>1: one RCCL gather of 16-byte decision records + one of the candidate records to rank 0,
           which replays and writes for all ranks.
Steps are pipelined (`--depth` batches in flight: the preparation of batch k+1 runs beside the DP kernels of batch k);
every step works on its own batch of groups (`distinct_batches` in the output; they repeat only when steps + warmup
exceeds it).  `value` = dispatched groups/s over all ranks.  Also reported: `pipelined_from_host` (the same steps fed
from host memory: staging + PCIe copy inside the timed region) and, with --kernel-only, the DP + scoring kernels of ONE
prepared work list replayed (the figure round 1 reported; kept for kernel A/B runs and profiles).

Contract: python bench.py --gpus N --steps K --warmup W prints ONE JSON line on rank 0.  N>1 without torchrun:
bench.py starts `python -m torch.distributed.run` itself as a child process.
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_VECTOR_TFLOPS = 78.6  # MI355X vector FP64, FMA counted as 2 flops (datasheet)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
FLOPS_PER_CELL = 45             # forward 19 + backward 20 + MAP 6 per band cell
FWD_FLOPS_PER_CELL = 19


def pmc_traffic(kernel, gps, platform, with_x2=False):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r0N_counters_<platform>.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of `bench.py --kernel-only`, KB units; tools/pmc_collect.py).
    The guide's x2 correction of FETCH_SIZE on gfx950 applies to 16 B/lane streaming reads; for the access pattern of
    these kernels the counters were calibrated on the backward kernel's known read / write volume:
    factor 1.  None when no profile matches this workload."""
    for name in (f"r06_counters_{platform}.json", f"r05_counters_{platform}.json", f"r04_counters_{platform}.json", f"r03_counters_{platform}.json", f"r02_counters_{platform}.json", "r01_counters.json"):
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", name)))
            meta = prof.get("_meta", {})
            if meta.get("platform", "hifi") != platform or not meta.get("groups_per_step"):
                continue
            k = prof.get("void " + kernel + "(spx_dev_batch)") or prof.get("void " + kernel + "(spx_dev_batch, spx_fast_consts)") or prof.get("void " + kernel) or prof.get(kernel)
            # per launch of THIS run: the kernels' traffic is proportional to the groups of a launch (same workload, same
            # per-problem volumes), the profile may have been taken at another batch size
            # (gps = groups per KERNEL launch of this run; the profile's dispatches are kernel launches too: slices of its lists)
            scale = gps / (meta["groups_per_step"] / max(1, meta.get("kernel_launches_per_step", 1)))
            b = (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024
            if with_x2:
                return int(b * scale), int((2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 * scale), name
            return int(b * scale)
        except Exception:
            continue
    return (None, None, None) if with_x2 else None


def pmc_kernel(kernel, gps, platform):
    """(duration alone on the chip in ms, VALU wave-instructions) per launch of `kernel` from the committed PMC passes, scaled to a launch
    of `gps` groups of this run (same workload: both are proportional to the groups of a launch); (None, None, None) without a profile"""
    for name in (f"r06_counters_{platform}.json", f"r05_counters_{platform}.json", f"r04_counters_{platform}.json", f"r03_counters_{platform}.json"):
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", name)))
            meta = prof.get("_meta", {})
            if meta.get("platform", "hifi") != platform or not meta.get("groups_per_step"):
                continue
            k = prof.get("void " + kernel + "(spx_dev_batch)") or prof.get("void " + kernel + "(spx_dev_batch, spx_fast_consts)") or prof.get("void " + kernel) or prof.get(kernel)
            scale = gps / (meta["groups_per_step"] / max(1, meta.get("kernel_launches_per_step", 1)))
            return k["duration_ns_alone"] * scale / 1e6, k["SQ_INSTS_VALU"] * scale, name
        except Exception:  # noqa: BLE001
            continue
    return None, None, None


def gen_parallel(genome, first, n, chunk, threads):
    """generate n groups starting at `first` in `chunk`-sized batches on `threads` threads"""
    starts = list(range(first, first + n, chunk))
    out = [None] * len(starts)
    it = iter(range(len(starts)))
    lock = threading.Lock()

    def run():
        while True:
            with lock:
                k = next(it, None)
            if k is None:
                return
            out[k] = genome.reads(starts[k], min(chunk, first + n - starts[k]))

    ths = [threading.Thread(target=run) for _ in range(max(1, threads))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return out


def from_bam_leg(args, genome, record_chunks, n_groups, ncpu, oracle_log, oracle_groups, n_devices=1):
    """BASELINE.json's 'GB BAM/sec' = SURVEY 8(d)'s metric (first byte read -> out.log closed): the command-line drop-in
    (compressed BGZF blocks to the device(s), inflate / record chain / fields / dispatch filter / staging / scoring there,
    finalizer, relabel list, BEDs on the host) on a BAM file holding `record_chunks` (generator chunks of the very workload
    that was timed), as a CHILD process (this one holds the GPU already).  The BAM is written by the C writer of synth/
    (htslib block policy, zlib level 6) onto tmpfs, so the file is in the page cache like a file that was just produced
    by an aligner.  Figures: the whole process (first byte -> out.log and BEDs closed; HIP start-up, FASTA parse,
    reference upload included) and its scoring loop alone (from the program's own timing line).  The relabel list is
    compared with the oracle's list of the first `oracle_groups` groups: one rand() stream in file order, so the
    oracle's list must be a byte prefix of the command line's."""
    import re
    import shutil
    import subprocess
    from secphase_amd import api, synth
    ont = args.platform == "ont"
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="spx_bench_bam_", dir=base)
    fa, bam, outd = os.path.join(d, "asm.fa"), os.path.join(d, "reads.bam"), os.path.join(d, "out")
    threads = min(ncpu, 64)
    # -@ of the command line = the CPUs the container may really use (16 on the MI355X boxes that show 256): more inflate
    # threads than that cost CPU time (21.5 core-s of inflate at -@64 against 16.2 at -@16 for the same file) and wall time
    cli_threads = max(1, min(threads, api.lib().spx_effective_cpus()))
    res = {"groups": n_groups, "host_threads": cli_threads}
    try:
        t0 = time.perf_counter()
        synth.write_fasta(fa, genome.ref)
        size = synth.write_bam(bam, record_chunks, genome.ref, threads=threads, level=6)
        res["bam_write_s"] = round(time.perf_counter() - t0, 2)
        res["bam_bytes"] = size
        exe = os.path.join(ROOT, "secphase_amd", "bin", "secphase")
        flags = ["--ont", "-b", "50"] if ont else ["--hifi"]
        dev_list = ",".join(str(k) for k in range(max(1, n_devices)))
        base_cmd = [exe] + flags + ["-@", str(cli_threads), "-i", bam, "-f", fa, "--outDir", outd, "--prefix", "bench", "--devices", dev_list]

        def run_cli(extra, times):
            """best of `times` runs: (wall, loop, CPU core-seconds of the child: user + system)"""
            import resource
            runs = []
            for _ in range(times):
                shutil.rmtree(outd, ignore_errors=True)
                r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
                t0 = time.perf_counter()
                p = subprocess.run(base_cmd + extra, capture_output=True, text=True)
                wall = time.perf_counter() - t0
                r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
                if p.returncode != 0:
                    return None, p
                m = re.search(r"time in the scoring loop: ([0-9.]+) s", p.stderr)
                runs.append((wall, float(m.group(1)) if m else None, (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)))
            m2 = re.search(r"time in the scoring loop: .*", p.stderr)
            res["cli_timing_line"] = m2.group(0)[:400] if m2 else None
            return runs, None

        # the default command line: DEVICE-RESIDENT input (compressed bytes up, inflate / record chain / fields / dispatch filter / staging as kernels)
        runs, bad = run_cli([], 2)  # two runs, the better one is reported (boxes of the pool differ; the first also warms the page cache of the binary)
        if runs is None:
            res["rc"] = bad.returncode
            res["stderr_tail"] = bad.stderr[-400:]
            return res
        wall, loop, cpu_s = min(runs)
        res["rc"] = 0
        res["input"] = "device-resident (spx_dbam: BGZF inflate, record chain, fields / tags, name groups, dispatch filter, staging as kernels)"
        res["devices"] = max(1, n_devices)
        res["wall_s"] = round(wall, 3)
        res["runs_wall_s"] = [round(r[0], 3) for r in runs]
        res["groups_per_s"] = round(n_groups / wall, 1)
        res["gb_bam_per_s"] = round(size / wall / 1e9, 4)
        res["host_cpu_core_s"] = round(cpu_s, 2)
        res["host_cpu_core_s_per_262144_groups"] = round(cpu_s * 262144 / max(1, n_groups), 2)
        if loop and loop > 0:
            res["loop_s"] = loop
            res["loop_groups_per_s"] = round(n_groups / loop, 1)
            res["loop_gb_bam_per_s"] = round(size / loop / 1e9, 4)
        got = open(os.path.join(outd, "bench.out.log"), "rb").read()
        if oracle_log and os.path.exists(oracle_log):
            want = open(oracle_log, "rb").read()
            res["out_log_identical_to_oracle"] = bool(got[:len(want)] == want and len(want) > 0)
            res["out_log_checked_groups"] = oracle_groups
            res["out_log_bytes"] = len(got)
        if not getattr(args, "no_host_input_leg", False):
            # the round-3 host reader on the same file, for comparison (one run)
            hr, hbad = run_cli(["--hostInput", "--groupsPerBatch", str(4096 if ont else 16384)], 1)
            if hr:
                hw, hl, hc = hr[0]
                res["host_input"] = {"wall_s": round(hw, 3), "groups_per_s": round(n_groups / hw, 1), "gb_bam_per_s": round(size / hw / 1e9, 4),
                                     "loop_groups_per_s": round(n_groups / hl, 1) if hl else None, "host_cpu_core_s": round(hc, 2),
                                     "what": "the same file with --hostInput (mmap reader, host pool + device inflate workers, host record walk / parse / staging)"}
                got_h = open(os.path.join(outd, "bench.out.log"), "rb").read()
                res["host_input"]["out_log_identical_to_device_input"] = bool(got_h == got)
        res["what"] = ("secphase_amd/bin/secphase on a synthetic BAM (tmpfs) of the same workload: whole process (exec -> all six output "
                       "files closed; HIP start-up, FASTA parse, reference upload included) and its scoring loop alone; the host side of the "
                       f"GPU box gives this container ~16 cores of CPU time (cgroup quota), which bounds inflate + staging (DESIGN.md 6b)")
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return res


def guard_exposure(genome, params, first, n_groups, ctx, ncpu):
    """PARITY-UNPINNED exposure of this workload (include/spx.h SPX_GUARD_*): how many DP problems lie in the regime
    in which the two readings of probaln.c's terminal guard differ (l_query <= bw and 2*bw+1 > l_ref), and how many records of the
    relabel list change when the reading is flipped -- from a CPU run of the ORACLE over `n_groups` groups under both readings
    (checker only; nothing here is timed), and from the HIP path over the same groups under both readings."""
    import numpy as np
    from oracle import orc
    from secphase_amd import api
    guard_api0, guard_orc0 = api.get_terminal_guard(), orc.get_terminal_guard()  # restored on the way out (a run started with SPX_TERMINAL_GUARD=row stays on it)
    sub = genome.reads(first, n_groups)
    plan = api.Plan(genome.ref, sub.batch, params)
    v = plan.view
    n = int(v.n_problems)
    L_ = np.ctypeslib.as_array(v.L, shape=(n,)) if n else np.zeros(0, np.int32)
    R_ = np.ctypeslib.as_array(v.R, shape=(n,)) if n else np.zeros(0, np.int32)
    bw = np.ctypeslib.as_array(v.bw, shape=(n,)) if n else np.zeros(0, np.int32)
    regime = int(((L_ <= bw) & (2 * bw + 1 > R_)).sum())
    res = {"groups": n_groups, "dp_problems": n, "guard_regime_problems": regime, "guard_regime_fraction": round(regime / max(1, n), 5),
           "default_reading": "band (u >= bw2*3+3)"}
    d = tempfile.mkdtemp(prefix="spx_guard_")

    def records_of(path):
        out = {}
        for rec in open(path).read().split("\n\n"):
            ln = rec.strip("\n").split("\n")
            if len(ln) >= 2 and ln[1].startswith("$"):
                out[ln[1]] = rec
        return out

    try:
        logs, scored = {}, {}
        t0 = time.perf_counter()
        for name, reading in (("band", 0), ("row", 1)):
            orc.set_terminal_guard(reading)
            logs[name] = os.path.join(d, f"oracle.{name}.out.log")
            _, r_ = orc.run_batch(sub.batch, genome.ref, params, threads=min(ncpu, 64), seed=1, reuse_scratch=True, log_path=logs[name])
            scored[name] = [[r_[g].score[a] for a in range(max(r_[g].n_aln, 0))] + [r_[g].best_idx] for g in range(n_groups)]
        # the -w / --writeBam output: every base quality calc_local_baq leaves in the records (ptMarker.c:706,759,763) under either reading
        quals = {}
        try:
            import ctypes as _C
            bb = sub.batch.contents
            end = max((bb.qual_off[a_] + bb.l_qseq[a_] for a_ in range(bb.n_alns)), default=0)
            q0 = np.frombuffer(_C.string_at(_C.addressof(bb.qual.contents), end), np.uint8).copy()
            for name, reading in (("band", 0), ("row", 1)):
                orc.set_terminal_guard(reading)
                quals[name], _ = orc.run_batch_quals(sub.batch, genome.ref, params, q0.copy(), threads=min(ncpu, 64))
            res["oracle_write_bam"] = {"qual_bytes": int(end), "qual_bytes_that_differ_between_the_readings": int((quals["band"] != quals["row"]).sum()),
                                       "qual_bytes_the_baq_changed_band": int((quals["band"] != q0).sum())}
        except Exception as e:  # noqa: BLE001
            res["oracle_write_bam"] = {"error": str(e)[:200]}
        orc.set_terminal_guard(0)
        a, b = records_of(logs["band"]), records_of(logs["row"])
        res["oracle"] = {"relabel_records_band": len(a), "relabel_records_row": len(b),
                         "relabel_records_that_differ": sum(1 for k in set(a) | set(b) if a.get(k) != b.get(k)),
                         "groups_whose_scores_or_decision_differ": sum(1 for x, y in zip(scored["band"], scored["row"]) if x != y),
                         "seconds": round(time.perf_counter() - t0, 1)}
        if ctx is not None:
            gpu = {}
            for name, reading in (("band", 0), ("row", 1)):
                api.set_terminal_guard(reading)
                out, _ = ctx.score_batch(sub.batch, params, finalize_seed=1)
                gpu[name] = [[out[g].score[a_] for a_ in range(max(out[g].n_aln, 0))] + [out[g].best_idx] for g in range(n_groups)]
            api.set_terminal_guard(0)
            if "band" in quals:
                try:
                    import copy
                    p_all = copy.copy(params)
                    p_all.flags = 1
                    diff = {}
                    for name, reading in (("band", 0), ("row", 1)):
                        api.set_terminal_guard(reading)
                        wq = ctx.prepare(sub.batch, p_all)
                        wq.launch()
                        wq.collect()
                        got = wq.apply_quals(sub.batch, q0.copy())
                        diff[name] = int((got != quals[name]).sum())
                        wq.free()
                    res["hip_write_bam"] = {"qual_bytes_that_differ_from_the_oracle_band": diff["band"], "qual_bytes_that_differ_from_the_oracle_row": diff["row"]}
                except Exception as e:  # noqa: BLE001
                    res["hip_write_bam"] = {"error": str(e)[:200]}
                api.set_terminal_guard(0)
            res["hip_path"] = {"groups_whose_scores_or_decision_differ": sum(1 for x, y in zip(gpu["band"], gpu["row"]) if x != y),
                               "equals_oracle_under_band": gpu["band"] == scored["band"], "equals_oracle_under_row": gpu["row"] == scored["row"]}
    finally:
        orc.set_terminal_guard(guard_orc0)
        api.set_terminal_guard(guard_api0)
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    return res


def also_leg(platform, steps, warmup):
    """the other BASELINE workloads on one GPU, each as a child process of its own (own context, own HBM): a short run of
    this very script; its JSON line is returned (cut down to the figures the headline has)"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--platform", platform, "--steps", str(steps), "--warmup", str(warmup),
           "--no-also", "--no-host-leg", "--no-build", "--verify", "64", "--cpu-runs", "3", "--cpu-threads", "32", "--no-host-input-leg"]
    # config 3 also end to end: `secphase --ont -b 50` on a BAM of one step's groups
    # both legs end to end too, on >= 100 k groups each: ONT 131 072 groups = 16.7 GB of BAM (8 distinct batches), mixed 131 072
    # groups (two of its batches); SPX_BENCH_ALSO_ONT_BAM / SPX_BENCH_ALSO_MIXED_BAM for other sizes (0: no leg)
    nbam = os.environ.get("SPX_BENCH_ALSO_ONT_BAM", "131072") if platform == "ont" else os.environ.get("SPX_BENCH_ALSO_MIXED_BAM", "131072")
    cmd += ["--from-bam", nbam] if int(nbam) > 0 else ["--no-from-bam"]
    # (mixed: lists of 65 536 groups, four in flight (round 6; up to round 5: 16 384 groups, eight in flight on six preparation lanes -- the single-lane
    # walks of a list's heaviest groups last the same whatever the list's size, so their cost per group falls with it: 209 k -> 230 k groups/s);
    # ONT: the preset of BASELINE config 3, 16 384 groups per step, four lists in flight (DP slices keep a list at ~30 GB) -- the parent has handed
    # its device memory back (spx_trim) before this runs; SPX_BENCH_ALSO_ONT_GPS=8192 for a shorter leg)
    ont_gps = os.environ.get("SPX_BENCH_ALSO_ONT_GPS")
    cmd += ["--distinct", "3", "--depth", "4"] if platform == "mixed" else \
        (["--distinct", "4", "--depth", "3", "--groups-per-step", ont_gps] if ont_gps else ["--distinct", "8", "--depth", "4"])
    env = dict(os.environ)
    t0 = time.perf_counter()
    p = subprocess.run(cmd, capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if p.returncode != 0 or line is None:
        return {"error": (p.stderr or p.stdout)[-300:], "rc": p.returncode}
    d = json.loads(line)
    r = d.get("roofline", {})
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
            "workload": d["config"]["workload"], "groups_per_step": d["config"]["groups_per_step_per_gpu"],
            "dp_cells_per_step": d["config"]["dp_cells_per_step"], "verified_groups_vs_oracle": d["config"]["verified_groups_vs_oracle"],
            "roofline": {"kernel": r.get("kernel"), "frac": r.get("frac"), "alone_frac": r.get("alone_frac"), "lane_instr_per_cell": r.get("lane_instr_per_cell"),
                         "achieved": r.get("achieved"), "avg_launch_ms": r.get("avg_launch_ms"), "traffic": r.get("traffic"), "phase": r.get("phase")},
            "verified_timed_groups": d["config"].get("verified_timed_groups"),
            "verified_own_relabel_list": (d["config"].get("verified_own_relabel_list") or {}).get("oracle_list_is_byte_prefix_of_this_runs_list"),
            "from_bam": d.get("from_bam"), "guard_exposure": d.get("guard_exposure"),
            "kernel_ms_per_step": d.get("kernel_ms_per_step"), "cpu_baseline": d.get("cpu_baseline"), "wall_s": round(dt, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--platform", default="hifi", choices=["hifi", "ont", "mixed"])
    ap.add_argument("--kernel-only", action="store_true", help="only the replay of one prepared work list (profiles, kernel A/B)")
    ap.add_argument("--groups-per-step", type=int, default=0, help="groups per rank per step (0: preset)")
    ap.add_argument("--depth", type=int, default=0,
                    help="batches in flight in the pipeline (their device preparations run side by side); 0: 3 (hifi), 4 (mixed), "
                         "4 (ont; round 3: 2 -- a list of 16 384 ONT groups kept ~70 GB of saved rows; with DP slices ~30 GB)")
    ap.add_argument("--distinct", type=int, default=0,
                    help="at most this many distinct batches per rank (HBM / host memory); 0: 8 for --platform hifi (8 x 131 072 = the 1 M reads of BASELINE "
                         "config 2), 4 otherwise (raised to depth + 1 where the pipeline needs it)")
    ap.add_argument("--gen-chunk", type=int, default=1024, help="groups per generator call (parallel generation)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="groups in the CPU baseline sample (0: preset)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the pipelined-from-host measurement")
    ap.add_argument("--cpu-bracket", action="store_true", help="also time the -O0 / calloc-per-call / 4-thread variants of the CPU baseline")
    ap.add_argument("--from-bam", type=int, default=-1, metavar="N",
                    help="groups in the BAM of the end-to-end leg (the command-line drop-in on a synthetic BAM of the same workload: "
                         "GB of compressed BAM per second, the second half of BASELINE.json's metric); default: one step's groups; "
                         "N = 1 GPU only")
    ap.add_argument("--no-from-bam", action="store_true", help="skip the end-to-end leg")
    ap.add_argument("--no-host-input-leg", action="store_true", help="end-to-end leg: skip the comparison run with --hostInput")
    ap.add_argument("--no-also", action="store_true", help="skip the short ONT / mixed legs that follow the headline (N = 1, --platform hifi)")
    ap.add_argument("--no-build", action="store_true", help="never build (under a profiler: no child processes)")
    ap.add_argument("--cpu-runs", type=int, default=3, help="timed runs of the CPU baseline after one warm-up; the median is reported")
    ap.add_argument("--cpu-threads", type=int, default=0, help="thread count of the CPU baseline (0: all hardware threads and 32, the better one)")
    ap.add_argument("--verify", type=int, default=256, help="groups checked against the oracle before timing")
    ap.add_argument("--guard-exposure", type=int, default=-1, metavar="N",
                    help="groups of the workload run through the oracle (CPU) and the HIP path under BOTH readings of probaln.c's terminal guard "
                         "(parity-unpinned switch, include/spx.h); -1: 4096 for --platform ont, 0 otherwise")
    ap.add_argument("--keep-log", default="", metavar="PATH", help="tests: rank 0 copies the relabel list of the whole run (set-up, warm-up and timed "
                    "steps: one rand() stream) to PATH, and every rank writes PATH.rank<r>.json = which generator ranges made up its batches "
                    "and the order the batches were run in")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: test rig for the N>1 code path on a box with fewer GPUs than ranks (ranks share devices, "
                         "records cross ranks through host memory); never a measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # one hardware queue per stream of the context (before HIP initialises)
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the one-rank-per-GPU job as a CHILD process -- nothing has touched the
        # GPU yet (torch is not even imported), and the launcher is never exec'd -- and relay its JSON line
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge
    from secphase_amd import api, records, shard, synth

    if args.no_build:
        if not os.path.exists(api.LIB_PATH):
            sys.exit("bench.py --no-build: secphase_amd/libspx.so is missing")
    else:
        if not os.path.exists(api.LIB_PATH):
            ge.build()
        ge.build_cpu_helpers()
    # The short ONT / mixed legs run FIRST, as child processes, while this process has not touched the GPU yet: a leg -- and the command line
    # its end-to-end part starts as a child of its own -- then shares the device with nobody.  Run after the headline, under a parent that
    # still held a GPU context (torch's cannot be given up), the ONT command line's kernels took 1.85 x as long (three processes with
    # contexts on one GPU are time-sliced: 19 k groups/s in the loop against 33 k on its own).
    also_results = None
    if world == 1 and not args.no_also and not args.kernel_only and args.platform == "hifi":
        also_results = {}
        for plat in ("ont", "mixed"):
            try:
                # (a timed region starts with an empty pipeline: its first list waits for a whole preparation; 8 steps = 524 288 mixed groups
                # (290 ms steps), 8 of the 195 ms ONT steps = 131 072 groups)
                also_results[plat] = also_leg(plat, 8, 2)
            except Exception as ex:  # noqa: BLE001
                also_results[plat] = {"error": str(ex)}
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the scoring path has no CPU fallback")
    if args.dist_backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    coll_dev = "cuda" if args.dist_backend == "nccl" else "cpu"  # where the collectives' tensors live
    L = api.lib()

    ont = args.platform == "ont"
    mixed = args.platform == "mixed"
    gps = args.groups_per_step or (16384 if ont else 65536 if mixed else 131072)
    if args.distinct <= 0:
        args.distinct = 8 if not (ont or mixed) else 4
    if args.depth <= 0:
        # round 4: the scratch of a work list (1/s of every DP row + the saved forward rows) exists per DP slice, not per list
        # (16 GB per slice by default), so that ONT lists (70 -> ~30 GB per 16 384 groups) and mixed lists fit deeper pipelines
        args.depth = 4 if mixed else (4 if ont else 3)
        if mixed:
            args.distinct = max(args.distinct, 3)
        if ont:
            args.distinct = max(args.distinct, 5)
    # (large batches: the preparation kernels are dependent chains -- one lane walks one alignment / one group -- whose
    # duration hardly depends on the number of groups, so their cost per group falls with the batch size)
    # config 5 (mixed HiFi+ONT, power-law lengths, <= 8 secondaries) is run as --hifi over the whole mix, SURVEY 8(d)
    params = records.preset("ont", bandwidth=50) if ont else records.preset("hifi")
    cfg = synth.default_cfg(synth.ONT if ont else synth.MIXED if mixed else synth.HIFI)
    # SURVEY section 8(d) assembly: 2 haplotypes x 10 contigs x 5 Mbp (+ paralog copies); resident in HBM as 4-bit codes
    t0 = time.time()
    genome = synth.Genome(cfg)
    t_genome = time.time() - t0
    ctx = api.Context(local_rank)
    ctx.set_reference(genome.ref)

    ncpu = os.cpu_count() or 1
    host_threads = max(1, min(64, ncpu // max(1, world)))
    # the from-host leg stages on as many threads as the container's CPU quota allows (16 on the MI355X boxes, which show
    # 256 CPUs): more only get the group throttled (measured: 555 k groups/s on 64 threads, 655 k on 16)
    stage_threads = int(os.environ.get("SPX_BENCH_STAGE_THREADS", max(1, min(64, api.lib().spx_effective_cpus() // max(1, world)))))
    n_total = args.steps + args.warmup
    D = 1 if args.kernel_only else max(2, min(max(n_total, 2), args.distinct))
    first = rank * D * gps  # every rank scores its own shard of the read-group stream (weak scaling)
    t0 = time.time()
    batches = []  # batches[i] = the record blocks (generator chunks) of distinct batch i
    batches.append(gen_parallel(genome, first, gps, args.gen_chunk, host_threads))
    batch_ranges = {0: [(first, gps)]}  # generator ranges (first group, count) of every distinct batch of this rank
    if D > 1:
        # the distinct batches stay in HOST memory for the whole run (names for the relabel list, the from-host leg) and in
        # HBM: bound them by what this rank's share of the host can hold (8 ranks x 8 batches x 7.5 GB would not fit everywhere)
        def batch_bytes(b):
            tot = 0
            for ch in b:
                bt = ch.batch.contents
                la = int(bt.n_alns) - 1  # offsets of the LAST record: the payload arrays end just behind them
                if la < 0:
                    continue
                tot += int(bt.cigar_off[la]) * 4 + int(bt.seq_off[la]) + int(bt.qual_off[la]) + 2 * int(bt.l_qseq[la]) + 64 * (la + 1)
                tot += max(0, int(bt.cs_off[la])) if bt.cs_off else 0
                tot += max(0, int(bt.md_off[la])) if bt.md_off else 0
            return tot
        try:
            per = max(1, batch_bytes(batches[0]))
            avail = int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1]) * 1024
            try:  # the container's own limit (cgroup v2): the GPU boxes show 3 TB of host memory and grant 300 GiB
                lim = open("/sys/fs/cgroup/memory.max").read().strip()
                if lim != "max":
                    avail = min(avail, int(lim) - int(open("/sys/fs/cgroup/memory.current").read()))
            except Exception:  # noqa: BLE001
                pass
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
            fit = int(0.45 * avail / max(1, local_world) // per)
            D = max(2, min(D, fit))
        except Exception:  # noqa: BLE001  (no /proc/meminfo, another record layout: keep the preset)
            pass
        if world > 1:
            dmin = torch.tensor([D], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(dmin, op=dist.ReduceOp.MIN)
            D = int(dmin.item())
    for i in range(1, D):
        batches.append(gen_parallel(genome, first + i * gps, gps, args.gen_chunk, host_threads))
        batch_ranges[i] = [(first + i * gps, gps)]
    # a staged list may be in flight once: run() keeps depth + 1 submissions in flight and rotates over S staged lists.  S >= depth + 1
    # even when host memory bounds the DISTINCT batches below that (8 ranks under one container limit: D = 2): a batch is then staged
    # into HBM more than once -- HBM has the room the host lacks -- and the pipeline keeps its depth
    S = D
    if not args.kernel_only:
        S = max(D, args.depth + 1)
        args.depth = max(1, min(args.depth, S - 1))
    # ---- BASELINE config 5 at N > 1 (load-balance stress): the step's groups are cut by COST, not by count.  Every rank has
    # generated its count-based share; the per-group costs (bases over all alignments of the group: what the preparation
    # walks and the DP realigns) are all-gathered, the same cost boundaries come out on every rank (shard_by_cost), and a
    # rank regenerates the groups of its cost-balanced range (the generator is deterministic per group index).
    shard_info = None
    if world > 1 and mixed and not args.kernel_only:
        def group_costs(b):
            out = []
            for ch in b:
                bt = ch.batch.contents
                lq = np.ctypeslib.as_array(bt.l_qseq, shape=(bt.n_alns,)).astype(np.float64)
                gf = np.ctypeslib.as_array(bt.grp_first, shape=(bt.n_groups + 1,))
                cs_ = np.concatenate([[0.0], np.cumsum(lq)])
                out.append(cs_[gf[1:]] - cs_[gf[:-1]])
            return np.concatenate(out)

        new_batches, imb_before, imb_after = [], [], []
        for i in range(D):
            mine = torch.from_numpy(group_costs(batches[i])).to(coll_dev)
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            cost = np.concatenate([p_.cpu().numpy() for p_ in parts])  # global order of the step: (rank, group)
            per_rank = [float(p_.sum()) for p_ in parts]
            imb_before.append(max(per_rank) * world / max(sum(per_rank), 1.0))
            bounds, imb = shard.shard_by_cost(cost, world)
            imb_after.append(imb)
            lo, hi = bounds[rank], bounds[rank + 1]
            pieces, ranges_i = [], []
            j = lo
            while j < hi:  # global index j = owner * gps + offset  ->  generator index owner * D * gps + i * gps + offset
                owner, off = divmod(j, gps)
                n_ = min(hi - j, gps - off)
                pieces += gen_parallel(genome, owner * D * gps + i * gps + off, n_, args.gen_chunk, host_threads)
                ranges_i.append((owner * D * gps + i * gps + off, n_))
                j += n_
            new_batches.append(pieces)
            batch_ranges[i] = ranges_i
        batches = new_batches
        shard_info = {"by": "cost (bases over the alignments of a group), secphase_amd/shard.py::shard_by_cost",
                      "imbalance_by_count": round(float(np.mean(imb_before)), 4), "imbalance_by_cost": round(float(np.mean(imb_after)), 4),
                      "groups_of_rank0_per_step": [int(sum(ch.batch.contents.n_groups for ch in b)) for b in batches]}
    t_gen = time.time() - t0
    ptrs = [[ch.batch for ch in b] for b in batches]
    gmax = max(int(sum(ch.batch.contents.n_groups for ch in b)) for b in batches)  # groups in this rank's largest batch

    # ---- parity gate on a sample of the very workload being timed (oracle = checker only) ----
    verified = 0
    if rank == 0 and args.verify > 0:
        from oracle import orc
        nchk = min(args.verify, gps)
        sub = genome.reads(first, nchk)
        out, _ = ctx.score_batch(sub.batch, params, finalize_seed=1)
        _, res = orc.run_batch(sub.batch, genome.ref, params, threads=min(ncpu, 64), seed=1)
        for i in range(nchk):
            o, e = out[i], res[i]
            ok = o.n_aln == e.n_aln and o.best_idx == e.best_idx and all(o.score[a] == e.score[a] for a in range(max(e.n_aln, 0)))
            if not ok:
                sys.exit(f"parity check failed on group {i}: GPU result differs from the oracle")
        verified = nchk
    guard_exp = None
    n_ge = args.guard_exposure if args.guard_exposure >= 0 else (4096 if ont else 0)
    if rank == 0 and world == 1 and n_ge > 0 and not args.kernel_only:
        try:
            guard_exp = guard_exposure(genome, params, first, min(n_ge, gps), ctx, ncpu)
        except Exception as ex:  # noqa: BLE001  (a side figure must not cost the line its headline)
            guard_exp = {"error": str(ex)}
        # the context keeps the work-list memory of those two unsliced 4 096-group lists for re-use (~2 x 20 GB on the ONT preset): hand it back,
        # or the timed pipeline -- five 30 GB lists in flight and the preparation pools of as many lanes -- waits for memory at the allocation
        # gate (measured: 424 instead of 325 ms per ONT step once the context had six preparation lanes)
        if not os.environ.get("SPX_BENCH_KEEP_GUARD_BLOCKS"):  # (set: leaves it to the library, which returns waiting blocks when the driver runs short)
            api._chk(L.spx_trim(ctx.h), "spx_trim")

    def ctx_sync():
        api._chk(L.spx_sync(ctx.h), "spx_sync")

    def sync_all():
        ctx_sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tmpdir = tempfile.mkdtemp(prefix="spx_bench_")
    log_path = os.path.join(tmpdir, f"bench.{rank}.out.log")

    # ------------------------------------------------------------------ kernel-only replay (round-1 figure)
    def kernel_only(steps, warmup):
        w = ctx.prepare(ptrs[0], params, host_threads=host_threads)
        for _ in range(warmup):
            w.launch()
        ctx_sync()
        w.collect(finalize_seed=None)  # closes the event window: the averages cover the timed launches only
        sync_all()
        t = time.perf_counter()
        for _ in range(steps):
            w.launch()
        sync_all()
        el = time.perf_counter() - t
        w.collect(finalize_seed=None)
        st = w.stats()
        w.free()
        return el, st

    staged, pipe, fin = [], None, C.c_void_p()
    t_stage = 0.0
    relabelled = [0]
    if not args.kernel_only:
        t0 = time.time()
        staged = [ctx.stage(ptrs[j % D], params, host_threads=host_threads) for j in range(S)]  # records -> HBM, outside the timed region
        ctx_sync()
        t_stage = time.time() - t0
        pipe = api.Pipe(ctx, params, depth=args.depth, host_threads=stage_threads)
        api._chk(L.spx_finalizer_create(1, C.byref(fin)), "spx_finalizer_create")
    outbuf = (api.GroupOut * max(gps, gmax))()

    def out_at(base):
        return C.cast(C.byref(outbuf, base * C.sizeof(api.GroupOut)), C.POINTER(api.GroupOut))

    def finish_step(step_global, i, n, work):
        """what follows the kernels: rand() replay in file order + relabel list (N=1), or the two gathers and rank 0's
        replay for all ranks (N>1)"""
        if world == 1:
            api._chk(L.spx_finalizer_apply(fin, C.byref(params), outbuf, n), "spx_finalizer_apply")
            base = 0
            for bp in ptrs[i]:
                api._chk(L.spx_write_relabel_log(log_path.encode(), b"a", bp, genome.ref, out_at(base)), "spx_write_relabel_log")
                base += bp.contents.n_groups
            relabelled[0] += sum(1 for g in range(0, n, 997) if outbuf[g].relabel)  # (sampled: keeps Python out of the timing)
            return
        # N > 1 (round 3): every rank decides its own groups -- the ranks exchange the number of draws their groups consume,
        # each moves its copy of the rand() stream over the others' draws, finalizes and formats its own fragment of the
        # list; ONE gather (RCCL) brings the fragments to rank 0, which appends them in rank order.  Rank 0 no longer replays
        # and formats every rank's records (~250 ms per 8 x 131 072 groups against a 170 ms step in round 2).
        shard.decide_locally(api, params, fin, outbuf, n, dist, torch, coll_dev)
        relabelled[0] += sum(1 for g in range(0, n, 997) if outbuf[g].relabel)
        frag = shard.relabel_text(api, ptrs[i], genome.ref, outbuf)
        parts = shard.gather_bytes(torch.from_numpy(frag).to(coll_dev), dist, torch)
        if rank == 0:
            writer_q.put(parts)  # appended on a helper thread; the timed region ends only when it has caught up

    import queue
    writer_q = queue.Queue()
    writer_err = []

    def writer_main():
        while True:
            item = writer_q.get()
            try:
                if item is None:
                    return
                if not writer_err:
                    shard.append_fragments(item, log_path)
            except Exception as ex:  # noqa: BLE001
                writer_err.append(ex)
            finally:
                writer_q.task_done()

    writer = None
    if world > 1 and rank == 0:
        writer = threading.Thread(target=writer_main, daemon=True)
        writer.start()

    def writer_drain():
        if writer is not None:
            writer_q.join()
            if writer_err:
                raise writer_err[0]

    batch_sequence = []  # batch index of every step whose list went into log_path, in order

    def run(nsteps, offset, from_host):
        submitted = received = 0
        # staged list j of S holds batch j % D (S = D unless host memory bounds D below the pipeline depth)
        if not from_host:
            batch_sequence.extend(((offset + k) % S) % D for k in range(nsteps))
        while received < nsteps:
            while submitted < nsteps and pipe.pending() < args.depth + 1:
                if from_host:
                    pipe.submit(batch=ptrs[(offset + submitted) % D])
                else:
                    pipe.submit(staged=staged[(offset + submitted) % S])
                submitted += 1
            j = (offset + received) % (D if from_host else S)
            i = j % D
            _, n = pipe.next(outbuf)
            finish_step(offset + received, i, n, staged[j] if not from_host else None)
            if not from_host:
                staged[j].release()  # the list's HBM goes back to the context for the next step's list
            received += 1

    def verify_last_step(i_last):
        """whole generator chunks of batch i_last (the records that were staged and timed) through the oracle: n_aln, primary,
        every score bit for bit; best_idx / relabel after replaying the chunk's draws from a fresh seed on both sides"""
        from oracle import orc
        chunks = ptrs[i_last]
        bases, b_ = [], 0
        for bp in chunks:
            bases.append(b_)
            b_ += bp.contents.n_groups
        want = max(256, args.verify)
        rng = np.random.default_rng(20241220 + i_last)
        order = rng.permutation(len(chunks))
        checked = 0
        for ci in order:
            if checked >= want:
                break
            bp, base = chunks[ci], bases[ci]
            n = bp.contents.n_groups
            _, res = orc.run_batch(bp, genome.ref, params, threads=min(ncpu, 64), seed=1, reuse_scratch=True)
            mine = (api.GroupOut * n)()
            C.memmove(mine, C.byref(outbuf, base * C.sizeof(api.GroupOut)), n * C.sizeof(api.GroupOut))
            api._chk(L.spx_finalize(C.byref(params), 1, mine, n), "spx_finalize")
            for g in range(n):
                o, e = mine[g], res[g]
                ok = (o.n_aln == e.n_aln and (e.n_aln <= 0 or (o.prim_idx == e.prim_idx and o.best_idx == e.best_idx and bool(o.relabel) == bool(e.relabel)))
                      and all(o.score[a] == e.score[a] for a in range(max(e.n_aln, 0))))
                if not ok:
                    sys.exit(f"parity check failed on group {base + g} of the last timed step: GPU result differs from the oracle")
            checked += n
        return checked

    verified_timed = 0
    host_leg = None
    host_leg_error = None
    if args.kernel_only:
        elapsed, st_k = kernel_only(args.steps, args.warmup)
        per_step_stats = [st_k]
    else:
        open(log_path, "w").close()
        run(args.depth + 1, 0, False)  # set-up, like the staging above: the context's HBM arenas of every pipeline slot exist
        run(args.warmup, 0, False)
        writer_drain()
        sync_all()
        t0 = time.perf_counter()
        run(args.steps, args.warmup, False)
        writer_drain()  # rank 0: the list of every timed step is on disk
        sync_all()
        elapsed = time.perf_counter() - t0
        per_step_stats = [staged[(args.warmup + k) % S].stats() for k in range(min(args.steps, S))]
        if args.keep_log:
            import shutil
            if rank == 0:
                shutil.copyfile(log_path, args.keep_log)
            with open(f"{args.keep_log}.rank{rank}.json", "w") as f_:
                json.dump({"rank": rank, "world": world, "D": D, "gps": gps, "sequence": batch_sequence,
                           "ranges": {str(k): v for k, v in batch_ranges.items() if k < D}}, f_)
        # ---- the TIMED steps' own output against the oracle (rank 0; the oracle is the checker, never the thing measured):
        # (a) scores / decision fields of whole generator chunks of the LAST timed step, straight from the buffer the step's
        # results were collected into; (b) the relabel list this run wrote: the oracle's list of the run's first groups must be
        # a byte prefix of it (one rand() stream in (step, rank, group) order, so rank 0's first groups come first at any N)
        if rank == 0 and args.verify > 0:
            verified_timed = verify_last_step(((args.warmup + args.steps - 1) % S) % D)
        if world == 1 and not args.no_host_leg:
            # the same steps fed from HOST memory: staging (host threads) + PCIe copy inside the timed region
            hs = max(2, min(args.steps, 8))
            # the resident copies of the records have served their purpose (the timed steps above): their HBM (15 GB per
            # batch) goes back to the context, or `depth` work lists of 44 GB + their images do not fit beside them
            for w_ in staged:
                w_.free()
            staged = []
            try:
                run(args.depth + 2, 0, True)  # warm-up: pinned staging buffers and device arenas of every slot exist
                sync_all()
                t1 = time.perf_counter()
                run(hs, args.depth + 2, True)
                sync_all()
                el_h = time.perf_counter() - t1
                host_leg = (hs, el_h)
            except api.SpxError as ex:  # a secondary figure must not cost the line its headline
                host_leg_error = str(ex)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the end-to-end leg (SURVEY 8(d)'s metric: first byte read -> out.log closed) runs the command line as a child process on
    # ALL the job's devices (`secphase --devices 0..N-1`: one input pipeline per GPU): every rank hands its device memory back first
    want_bam_all = not args.no_from_bam and not args.kernel_only and args.from_bam != 0
    if world > 1 and want_bam_all:
        if pipe is not None:
            pipe.close()
            pipe = None
        for w_ in staged:
            w_.free()
        staged = []
        ctx.close()  # memory AND hardware queues: see the N = 1 case below
        torch.cuda.empty_cache()
        flag_ = os.path.join(tempfile.gettempdir(), f"spx_bench_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}.done")
        if rank == 0 and os.path.exists(flag_):
            os.unlink(flag_)  # (a stale flag of an earlier job on the same port)
        dist.barrier()

    nst = len(per_step_stats)
    n_disp = sum(int(s.n_dispatched) for s in per_step_stats) / nst
    n_prob = sum(int(s.n_problems) for s in per_step_stats) / nst
    n_rows = sum(int(s.n_rows) for s in per_step_stats) / nst
    cells = sum(int(s.dp_cells) for s in per_step_stats) / nst
    bytes_in = sum(int(s.bytes_h2d) for s in per_step_stats) / nst
    if world > 1:
        tot = torch.tensor([n_disp, n_prob, cells], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tot)
        n_disp_all, n_prob_all, cells_all = [float(x) for x in tot.tolist()]
        mx = torch.tensor([cells], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        imbalance = float(mx.item()) * world / cells_all if cells_all > 0 else 1.0  # heaviest rank / mean rank, in DP cells
    else:
        n_disp_all, n_prob_all, cells_all = n_disp, n_prob, cells
        imbalance = 1.0

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_disp_all * args.steps / elapsed
        # ---- roofline of the dominant kernel: the forward kernel of the band class holding most cells, timed with HIP
        # events on the stream it is launched on (spx_launch), averaged over the timed steps.
        # FP64 vector-ALU bound (not HBM, not MFMA): 19 of the 45 flops per band cell are forward flops.
        st0 = per_step_stats[0]
        G, slots = st0.main_class_lanes, st0.main_class_slots
        tiers_on = bool(L.spx_get_dp_tiers()) and any(p.tier_fast_problems > 0 for p in per_step_stats)
        if tiers_on:
            # the fast tier's forward kernel of that class (spx_fast_kernels.hip spx_launch_fast: lanes x slots, exact width, waves per SIMD, fence)
            fast_names = {42: "fast_fwd_kernel<2, 21, 41, 4, 1>", 44: "fast_fwd_kernel<2, 22, 43, 3, 1>", 46: "fast_fwd_kernel<2, 23, 45, 3, 1>",
                          (1, 48): "fast_fwd_kernel<2, 24, 47, 3, 1>", (2, 48): "fast_fwd_kernel<2, 24, 0, 3, 1>", 64: "fast_fwd_kernel<4, 16, 0, 4, 1>",
                          104: "fast_fwd_kernel<4, 26, 0, 3, 1>", 112: "fast_fwd_kernel<4, 28, 0, 3, 1>", 120: "fast_fwd_kernel<4, 30, 0, 2, 1>"}
            kname = fast_names.get(slots) or fast_names.get((G, slots)) or (f"baq_fwd1_kernel<{slots - 1}>" if G == 1 else f"baq_fwd_kernel<{G}, {slots // G}, 0, false>")
        else:
            kname = f"baq_fwd1_kernel<{slots - 1}>" if G == 1 else f"baq_fwd_kernel<{G}, {slots // G}, 0, false>"
        # (a sliced list launches the kernel once per DP slice: the roofline is per KERNEL launch -- what a kernel trace shows)
        n_slices = max(1, int(round(sum(max(1, p.dp_slices) for p in per_step_stats) / nst)))
        fwd_ms = sum(p.main_fwd_ms for p in per_step_stats) / nst / n_slices
        bwd_ms = sum(p.main_bwd_ms for p in per_step_stats) / nst
        baq_ms = sum(p.baq_kernel_ms for p in per_step_stats) / nst
        score_ms = sum(p.score_kernel_ms for p in per_step_stats) / nst
        # round 6: the two spans separately.  dp_critical: what a list keeps the MAIN stream for (first DP kernel -> behind its last backward / MAP
        # kernel); tail_span: re-runs of uncertified problems + scoring kernels on the result stream BESIDE the next list's DP kernels (start of its
        # first kernel -> end of its last, waiting for the chip included).  Only the first is inside the step's critical path.
        crit_ms = sum(p.dp_critical_ms for p in per_step_stats) / nst
        tail_ms = sum(p.tail_span_ms for p in per_step_stats) / nst
        fast_prob = sum(p.tier_fast_problems for p in per_step_stats) / nst
        rerun_prob = sum(p.tier_rerun_certificate + p.tier_rerun_model + p.tier_rerun_range for p in per_step_stats) / nst
        cls_cells = sum(p.main_class_cells for p in per_step_stats) / nst / n_slices
        achieved_tf = FWD_FLOPS_PER_CELL * cls_cells / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
        # whole BAQ phase (forward + backward + MAP, all classes) at the algorithm's 45 flop per cell
        phase_tf = FLOPS_PER_CELL * cells / (crit_ms * 1e-3) / 1e12 if crit_ms > 0 else 0.0
        compulsory = n_prob * 1700  # ~1.7 KB of compulsory HBM bytes per DP problem
        alone_ms, valu_insts, pmc_src = pmc_kernel(kname, gps / n_slices, args.platform)
        roofline = {
            "bound": "valu_fp64",  # FP64 VECTOR ALU (no MFMA is issued; MI355X's FP64 MFMA peak equals its FP64 vector peak, so the roof is the same number)
            "compute_unit": "valu_fp64",
            # the same kernel ALONE on the chip (rocprofv3 --pmc serialises the dispatches; committed pass, scaled to this launch size):
            # what the kernel can do when nothing runs beside it -- `frac` is what it gets inside the pipelined step
            "alone_frac": round(FWD_FLOPS_PER_CELL * cls_cells / (alone_ms * 1e-3) / 1e12 / PEAK_FP64_VECTOR_TFLOPS, 4) if alone_ms else None,
            "alone_launch_ms": round(alone_ms, 3) if alone_ms else None,
            # VALU wave-instructions x 64 lanes per band cell of the launch (SQ_INSTS_VALU of the same pass): 19 algorithmic flops per cell
            "lane_instr_per_cell": round(valu_insts * 64 / cls_cells, 2) if valu_insts and cls_cells else None,
            "alone_source": pmc_src,
            "achieved": round(achieved_tf, 3),
            "peak": PEAK_FP64_VECTOR_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved_tf / PEAK_FP64_VECTOR_TFLOPS, 4),
            "traffic": pmc_traffic(kname, gps / n_slices, args.platform),
            "traffic_if_fetch_size_counts_half": pmc_traffic(kname, gps / n_slices, args.platform, with_x2=True)[1],
            "traffic_source": pmc_traffic(kname, gps, args.platform, with_x2=True)[2],
            "kernel": kname,
            "avg_launch_ms": round(fwd_ms, 4),
            "launches_averaged": int(sum(p.n_launches_averaged for p in per_step_stats)) * n_slices,
            "kernel_launches_per_step": n_slices,
            "cells_per_launch": int(cls_cells),
            "flops_per_cell": FWD_FLOPS_PER_CELL,
            "traffic_note": "HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE, every kernel alone on the chip). "
                            "`traffic` takes the counters as they read: calibrated on the backward kernel's known volume (DESIGN 3.1), these 8-16 B per lane, "
                            "partly scattered accesses are not under-reported; `traffic_if_fetch_size_counts_half` applies the guide's gfx950 correction for "
                            "16 B/lane streaming reads (FETCH_SIZE x 2) anyway, as the upper bound",
            "tier": ("fast (certified) + exact re-run of uncertified problems" if tiers_on else "exact"),
            "executed_flops_per_cell": (11 if tiers_on else 19),
            "note": "compute-bound on the FP64 VECTOR ALU, not on the matrix cores (sequential dependences inside every row). "
                    "`achieved` = ALGORITHMIC forward flops (19 per band cell of the class, SURVEY 8(d): the reference's forward pass over every "
                    "row) / the kernel's launch duration.  With the two-tier DP (DESIGN 3.4) the kernel that answers is the FAST forward kernel: same "
                    "model, fused multiply-adds, no row sums, rows behind a problem's last wanted row not walked (they have no observable "
                    "effect: 73 % of the rows of this workload are walked) -- it executes ~11 flops per cell it walks (6 FP64 instructions, 5 of "
                    "them fma) -- and the exact kernels re-run the `flagged_problem_fraction` its certificate does not cover.  peak = vector FP64 "
                    "at 2.4 GHz with FMA counted as 2.  avg_launch_ms = HIP events on the launch stream, averaged over the timed steps (the other "
                    "band classes, and the preparation kernels of the next batch, run beside it on their own streams)",
            "flagged_problem_fraction": round(rerun_prob / fast_prob, 7) if fast_prob > 0 else None,
            "phase": {"what": "forward + backward + MAP kernels, all band classes, 45 algorithmic flop per band cell, over the list's span on the main stream (dp_critical_path)",
                      "achieved": round(phase_tf, 3), "frac": round(phase_tf / PEAK_FP64_VECTOR_TFLOPS, 4),
                      "ms_per_launch": round(crit_ms, 4), "backward_kernel_ms": round(bwd_ms, 4)},
            "hbm": {"bound": "hbm", "achieved": round(compulsory / (crit_ms * 1e-3) / 1e9, 2) if crit_ms > 0 else 0.0,
                    "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(compulsory / (crit_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 6) if crit_ms > 0 else 0.0,
                    "algorithmic_bytes_per_launch": int(compulsory)},
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only: the other ranks would sit in a barrier
            from oracle import orc
            ns = args.cpu_sample or max(256 if ont else 1024, (4 if ont else 16) * ncpu)
            ns = min(ns, gps)
            sample = genome.reads(first, ns)
            # the oracle keeps the reference's memory pattern (two ~0.8 MB matrices zeroed per BAQ call), which
            # saturates the host memory system well before all hardware threads are busy: time it at all
            # threads and at 32, report the better one with the thread count actually used

            oracle_log = os.path.join(tmpdir, "oracle_prefix.out.log")

            def time_oracle(cores, reuse, variant=None, log=None, sub=None):
                sb = sub or sample
                t_ = time.perf_counter()
                _, res_ = orc.run_batch(sb.batch, genome.ref, params, threads=cores, seed=1, reuse_scratch=reuse, variant=variant, log_path=log)
                dt_ = time.perf_counter() - t_
                nd_ = sum(1 for r in res_ if r.n_aln > 0)
                return nd_ / dt_, dt_, sum(r.dp_cells for r in res_) / dt_

            # one warm-up, then timed runs, the median is reported.  The oracle keeps per-thread DP scratch
            # (the reference's calloc/free per BAQ call is in the bracket below); thread counts: all hardware threads and 32
            # -- the better one with the count actually used (the GPU boxes give a container ~16 cores of CPU time, so more
            # threads than that only add switching).
            best = None
            nrun = max(1, args.cpu_runs)
            for cores in ([min(ncpu, args.cpu_threads)] if args.cpu_threads > 0 else sorted({ncpu, min(ncpu, 32)}, reverse=True)):
                time_oracle(cores, True, log=oracle_log if world == 1 else None)  # warm-up; also writes the list the BAM leg is checked against
                runs = sorted(time_oracle(cores, True) for _ in range(nrun))
                v, dt, cps = runs[len(runs) // 2]
                cand = {"value": round(v, 2), "unit": "groups/s", "cores": cores, "threads": cores, "effective_cpus": int(L.spx_effective_cpus()),
                        "host_hw_threads": ncpu, "kind": "port",
                        "cores_note": "`cores` = `threads` = worker threads of the oracle's pool (the contract's field); the container may use "
                                      "`effective_cpus` CPUs' worth of time (cgroup quota) of the host's `host_hw_threads` hardware threads, so "
                                      "the figure does not scale to a host that grants more",
                        "sample": f"first {ns} groups of the same workload, oracle (C restatement, -O2 -ffp-contract=off, "
                                  f"pthread pool over groups, per-thread DP scratch instead of calloc/free per call), "
                                  f"median of {nrun} runs after a warm-up, {dt:.2f} s wall; host has {ncpu} hardware threads",
                        "runs": [round(r[0], 1) for r in runs],
                        "cells_per_s": round(cps, 1)}
                if best is None or cand["value"] > best["value"]:
                    best = cand
            cpu = best
            cpu["oracle_log"] = oracle_log
            cpu["oracle_groups"] = ns
            if args.cpu_bracket or (args.platform == "hifi" and not args.no_also):
                # what separates the port from the real reference build (smaller samples for the
                # slow variants, so that the default run stays within minutes)
                br = {}
                cores = best["cores"]
                v, dt, _ = time_oracle(cores, False)
                br["calloc_per_call"] = {"value": round(v, 2), "cores": cores, "what": "two calloc'd FP64 matrices per BAQ call, like htslib"}
                try:
                    v, dt, _ = time_oracle(cores, False, variant="O0")
                    br["O0_build"] = {"value": round(v, 2), "cores": cores,
                                      "what": "gcc -O0 (the reference Makefile gives no -O flag), calloc per call"}
                except Exception as e:  # noqa: BLE001
                    br["O0_build"] = {"error": str(e)}
                small = genome.reads(first, min(ns, 512))
                v1, dt1, _ = time_oracle(1, True, sub=small)
                br["one_thread"] = {"value": round(v1, 2), "cores": 1, "sample_groups": min(ns, 512)}
                mid = genome.reads(first, min(ns, 1024))
                v4, dt4, _ = time_oracle(min(4, ncpu), False, sub=mid)
                br["reference_default_threads"] = {"value": round(v4, 2), "cores": min(4, ncpu), "sample_groups": min(ns, 1024),
                                                   "what": "-@ default of the reference (4), calloc per call"}
                try:
                    orc.set_reference_overheads(True)
                    v, dt, _ = time_oracle(cores, False)
                    br["fai_reload_and_regcomp"] = {"value": round(v, 2), "cores": cores,
                                                    "what": "calloc per call + the reference's per-group fai_load (src/secphase.c:101) "
                                                            "and per-iterator regcomp / per-token regexec (cigar_it.c:50,148)"}
                except Exception as e:  # noqa: BLE001
                    br["fai_reload_and_regcomp"] = {"error": str(e)}
                finally:
                    try:
                        orc.set_reference_overheads(False)
                    except Exception:  # noqa: BLE001
                        pass
                cpu["bracket"] = br
        own_log_check = None
        if not args.kernel_only and args.verify > 0:
            from oracle import orc
            o_log, o_groups = (cpu.get("oracle_log"), cpu.get("oracle_groups")) if cpu else (None, 0)
            if not o_log or not os.path.exists(o_log):
                # (N > 1, or no CPU baseline asked for: the oracle on the first generator chunk of this rank's first batch)
                o_log = os.path.join(tmpdir, "oracle_first_chunk.out.log")
                first_chunk = ptrs[0][0]
                orc.run_batch(first_chunk, genome.ref, params, threads=min(ncpu, 64), seed=1, reuse_scratch=True, log_path=o_log)
                o_groups = int(first_chunk.contents.n_groups)
            want_b = open(o_log, "rb").read()
            got_b = open(log_path, "rb").read()
            own_log_check = {"oracle_list_is_byte_prefix_of_this_runs_list": bool(len(want_b) > 0 and got_b[:len(want_b)] == want_b),
                             "groups": int(o_groups), "oracle_list_bytes": len(want_b), "this_runs_list_bytes": len(got_b)}
            if not own_log_check["oracle_list_is_byte_prefix_of_this_runs_list"]:
                sys.exit("parity check failed: the oracle's relabel list of the first groups is not a byte prefix of the list this run wrote")
        line = {
            "metric": "reads/sec (primary+secondary groups scored)",
            "value": round(value, 2),
            "unit": "groups/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("1M ONT reads, 30 kb, <=4 secondaries, band=50, ONT gap params" if ont else
                             "Mixed HiFi+ONT reads, length 2-100 kb power-law, <=8 secondaries, run as --hifi" if mixed else
                             "1M HiFi reads, 15 kb, <=2 secondaries, band=20, BAQ window 500 bp") +
                            f" (streamed as batches of {gps} groups per GPU per step; " +
                            ("ONE prepared work list replayed: kernels only)" if args.kernel_only else
                             "records resident in HBM, whole path records -> relabel list inside the step)"),
                "groups_per_step_per_gpu": gps,
                "distinct_batches_per_gpu": D,
                "staged_lists_per_gpu": S,
                "pipeline_depth": args.depth,
                "dp_problems_per_step": int(n_prob_all),
                "dp_cells_per_step": int(cells_all),
                "wanted_rows_per_step_rank0": int(n_rows),
                "record_bytes_per_step_rank0": int(bytes_in),
                "parallelism": (f"reads sharded over {world} GPUs; every rank decides and formats its own groups (draw counts exchanged, "
                                f"rand() stream kept in global step), ONE RCCL gather of the relabel-list fragments to rank 0, which "
                                f"appends them in rank order") if world > 1 else "single GPU",
                "verified_groups_vs_oracle": verified,
                "verified_timed_groups": verified_timed,
                "verified_own_relabel_list": own_log_check,
                "distinct_groups": (f"{D} distinct batches x {gps} groups per GPU in rotation (a steady-state figure: "
                                    f"{(args.steps + args.warmup) * gps} group scorings over {D * gps} distinct groups per GPU)"),
                "rank_imbalance_dp_cells": round(imbalance, 4),
                "sharding": shard_info,
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            # what ONE list keeps the main stream for (every figure below is <= ms_per_step; the tail runs beside the NEXT list's DP kernels)
            "kernel_ms_per_step": {"dp_critical_path": round(crit_ms, 3), "main_class_forward": round(fwd_ms * n_slices, 3), "main_class_backward": round(bwd_ms, 3),
                                   "score": round(score_ms, 3)},
            "tail_span_ms": round(tail_ms, 3),
            "dp_tiers": {"on": tiers_on, "fast_class_problems_per_step": int(fast_prob), "rerun_problems_per_step": round(rerun_prob, 1),
                         "flagged_problem_fraction": round(rerun_prob / fast_prob, 7) if fast_prob > 0 else None},
            "dp_cells_per_s": round(cells_all * args.steps / elapsed, 1),
            "dp_problems_per_s": round(n_prob_all * args.steps / elapsed, 1),
            "setup_s": {"genome": round(t_genome, 2), "generate": round(t_gen, 2), "stage_records_to_hbm": round(t_stage, 2)},
        }
        if guard_exp is not None:
            line["guard_exposure"] = guard_exp
        if host_leg:
            hs, el_h = host_leg
            line["pipelined_from_host"] = {"value": round(n_disp * hs / el_h, 2), "unit": "groups/s", "steps": hs,
                                           "ms_per_step": round(el_h / hs * 1e3, 3),
                                           "what": "the same steps with the records in HOST memory: dispatch filter + staging into pinned "
                                                   f"memory on {stage_threads} host threads + PCIe copy inside the timed region",
                                           "GB_per_s_over_pcie": round(bytes_in * hs / el_h / 1e9, 2)}
        if host_leg_error:
            line["pipelined_from_host"] = {"error": host_leg_error}
        if not args.kernel_only:
            line["relabelled_sampled"] = relabelled[0]
        # uncompressed record bytes (what spx_stage hands to the device: flags, CIGAR, SEQ, QUAL, cs/MD text) through the step
        line["gb_records_per_s"] = round(bytes_in * world * args.steps / elapsed / 1e9, 2)
        want_bam = want_bam_all
        if want_bam:
            # the command line / the other workloads run as other processes on the same GPU: hand back what this one holds first
            if pipe is not None:
                pipe.close()
                pipe = None
            for w_ in staged:
                w_.free()
            staged = []
            # (the whole context, not only its memory: a context holds 16 hardware queues, and with this process, an `also` child and ITS
            # command-line child alive the GPU's queue slots are oversubscribed -- the scheduler then time-slices the processes: the ONT
            # end-to-end leg ran at 18.8 k groups/s as a grandchild of a process with a live context, at 33.4 k on its own)
            ctx.close()
            torch.cuda.empty_cache()
        if want_bam:
            # every distinct batch of this rank (524 288 HiFi groups by default): start-up (HIP initialisation, reference upload:
            # ~0.4 s) and the exit of the process are part of the metric, a larger file shows the steady state better
            nb = D * gps if args.from_bam < 0 else args.from_bam
            chunks, have = [], 0
            for b in ptrs:  # the generator chunks of the timed batches, in order
                for ch in b:
                    if have < nb:
                        chunks.append(ch)
                        have += ch.contents.n_groups
            try:
                n_devs = world if args.dist_backend == "nccl" else min(world, torch.cuda.device_count())
                olog = cpu.get("oracle_log") if cpu else None
                ogr = cpu.get("oracle_groups") if cpu else 0
                if not olog and own_log_check:
                    olog, ogr = os.path.join(tmpdir, "oracle_first_chunk.out.log"), own_log_check["groups"]
                line["from_bam"] = from_bam_leg(args, genome, chunks, have, ncpu, olog, ogr, n_devices=n_devs)
            except Exception as ex:  # noqa: BLE001  (a secondary figure must not cost the line its headline)
                line["from_bam"] = {"error": str(ex)}
            line["gb_bam_per_s"] = line["from_bam"].get("gb_bam_per_s")
            fb = line["from_bam"]
            line["metric_8d"] = {"what": "SURVEY 8(d): groups that pass the dispatch filter / wall time from the first byte read to out.log closed, and compressed "
                                         "BAM bytes / the same time: the command line as a child process (HIP start-up, FASTA parse, reference upload, exit included)",
                                 "groups_per_s": fb.get("groups_per_s"), "gb_bam_per_s": fb.get("gb_bam_per_s"), "loop_groups_per_s": fb.get("loop_groups_per_s"),
                                 "loop_over_value": round(fb["loop_groups_per_s"] / value, 3) if fb.get("loop_groups_per_s") else None,
                                 "host_cpu_core_s_per_262144_groups": fb.get("host_cpu_core_s_per_262144_groups"), "devices": fb.get("devices"),
                                 "out_log_identical_to_oracle": fb.get("out_log_identical_to_oracle")}
            if cpu and line["from_bam"].get("groups_per_s"):
                line["from_bam"]["whole_process_vs_cpu_baseline"] = round(line["from_bam"]["groups_per_s"] / cpu["value"], 1)
        if cpu:
            cpu.pop("oracle_log", None)
            cpu.pop("oracle_groups", None)
        if also_results is not None:
            line["also"] = also_results
        print(json.dumps(line), flush=True)
    if world > 1 and want_bam_all:
        # rank 0 ran the command line on EVERY device of the job: the other ranks wait on the host (a flag file), not in a collective
        # whose kernel would spin on their GPUs beside the child process's kernels
        flag = os.path.join(tempfile.gettempdir(), f"spx_bench_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}.done")
        if rank == 0:
            open(flag, "w").close()
        else:
            t_wait = time.time()
            while not os.path.exists(flag) and time.time() - t_wait < 1500:
                time.sleep(0.2)
    if writer is not None:
        writer_q.put(None)
        writer.join()
    if pipe is not None:
        pipe.close()
    if fin:
        L.spx_finalizer_free(fin)
    for w in staged:
        w.free()
    if world > 1:
        dist.barrier()
        if rank == 0 and want_bam_all:
            try:
                os.unlink(os.path.join(tempfile.gettempdir(), f"spx_bench_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}.done"))
            except OSError:
                pass
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
