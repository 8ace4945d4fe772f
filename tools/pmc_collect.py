#!/usr/bin/env python3
"""Hardware counters of the bench kernels, the way MI355X_MICROARCH.md prescribes: every counter group in its OWN
rocprofv3 pass (FETCH_SIZE and WRITE_SIZE cannot share one; --pmc is never combined with a trace option), the
program itself right after `--` (python3 bench.py ..., no env / shell hop).  Under --pmc rocprofv3 serialises the
dispatches, so durations here are those of each kernel ALONE on the chip.

  python3 tools/pmc_collect.py --platform ont --out profiles/r02_counters_ont.json [--steps 2]

Output: {kernel name: {counter: mean per dispatch over the timed launches, "duration_ns_alone", "effective_clock_GHz",
"dispatches"}, "_meta": {...}}.  FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3's unit)."""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"],
          ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES",
           "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"],
          ["SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
           "SQ_INSTS_VALU_MFMA_MOPS_F64"]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--platform", default="hifi")
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--groups-per-step", type=int, default=0)
    ap.add_argument("--scratch", default=os.path.join(ROOT, "gpurun_out", "pmc_tmp"))
    ap.add_argument("--extra", nargs="*", default=[], help="more arguments for bench.py")
    ap.add_argument("--full", action="store_true", help="the whole pipelined step (preparation kernels included) instead of --kernel-only")
    ap.add_argument("--more", action="store_true", help="extra passes: instruction-cache, LDS and scalar activity counters (a pass whose counter this chip lacks is listed under failed_passes)")
    ap.add_argument("--inflate", action="store_true", help="profile tools/inflate_bench.py (the BGZF inflate kernel alone on a synthetic BAM's blocks) instead of bench.py")
    a = ap.parse_args()
    os.environ.setdefault("TMPDIR", "/tmp")
    # build once, here: the profiled bench never builds (a process the profiler has initialised must not spawn compilers)
    subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)
    res, meta = {}, {"platform": a.platform, "steps": a.steps, "warmup": a.warmup, "passes": PASSES,
                     "note": "each pass = rocprofv3 --pmc <counters> -- python3 bench.py ...; dispatches are serialised under "
                             "--pmc, so duration_ns_alone is the kernel alone on the chip; FETCH_SIZE/WRITE_SIZE in KB"}
    bench = ["python3", os.path.join(ROOT, "bench.py"), "--platform", a.platform, "--steps", str(a.steps), "--warmup",
             str(a.warmup), "--no-cpu-baseline", "--verify", "0", "--no-build", "--no-from-bam", "--no-also"] + \
            (["--no-host-leg", "--depth", "1"] if a.full else ["--kernel-only"]) + a.extra
    if a.groups_per_step:
        bench += ["--groups-per-step", str(a.groups_per_step)]
    if a.inflate:
        bench = ["python3", os.path.join(ROOT, "tools", "inflate_bench.py"), "--groups", str(a.groups_per_step or 16384), "--platform", a.platform,
                 "--repeat", str(a.steps)] + a.extra
    passes = list(PASSES)
    if a.more:
        passes += [["SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"], ["SQ_INST_CYCLES_SALU", "SQ_THREAD_CYCLES_VALU", "SQ_IFETCH"],
                   ["SQC_ICACHE_REQ", "SQC_ICACHE_MISSES", "SQC_ICACHE_HITS"], ["SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_VMEM", "SQ_LEVEL_WAVES"]]
        meta["passes"] = passes
    for k, ctrs in enumerate(passes):
        d = os.path.join(a.scratch, f"pass{k}")
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d, exist_ok=True)
        cmd = ["rocprofv3", "--pmc"] + ctrs + ["-d", d, "-o", "run", "--output-format", "csv", "--"] + bench
        p = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp")
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            meta.setdefault("failed_passes", []).append({"counters": ctrs, "rc": p.returncode, "stderr": p.stderr[-600:]})
            continue
        bj = json.loads(line[-1])
        if a.inflate:
            meta["groups_per_step"] = bj["groups"]
            meta["blocks"], meta["inflated_bytes"], meta["compressed_bytes"] = bj["blocks"], bj["inflated_bytes"], bj["compressed_bytes"]
        else:
            meta["groups_per_step"] = bj["config"]["groups_per_step_per_gpu"]
            meta["kernel_launches_per_step"] = bj.get("roofline", {}).get("kernel_launches_per_step", 1)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        acc = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"]
                if "spx" not in name and "baq" not in name and "kernel" not in name:
                    continue
                e = acc.setdefault(name, {})
                key = (row["Dispatch_Id"])
                dd = e.setdefault(key, {"dur": int(row["End_Timestamp"]) - int(row["Start_Timestamp"])})
                dd[row["Counter_Name"]] = dd.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                for extra in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size"):
                    dd["_" + extra] = row.get(extra)
        for name, disp in acc.items():
            ids = sorted(disp, key=int)
            # the timed launches are the last `steps` dispatches of each kernel (warm-up and the verify pass come first)
            ids = ids[-a.steps:] if len(ids) >= a.steps else ids
            out = res.setdefault(name, {})
            for c in ctrs:
                vals = [disp[i].get(c) for i in ids if c in disp[i]]
                if vals:
                    out[c] = sum(vals) / len(vals)
            out["dispatches"] = len(ids)
            out["duration_ns_alone"] = sum(disp[i]["dur"] for i in ids) / len(ids)
            for extra in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size"):
                out[extra] = disp[ids[-1]].get("_" + extra)
    for name, out in res.items():
        if "GRBM_GUI_ACTIVE" in out and out.get("duration_ns_alone"):
            out["effective_clock_GHz"] = round(out["GRBM_GUI_ACTIVE"] / 8 / out["duration_ns_alone"], 3)  # the counter sums the 8 XCDs
    res["_meta"] = meta
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1, sort_keys=True)
    print(f"wrote {a.out}: {len(res) - 1} kernels, failed passes: {len(meta.get('failed_passes', []))}")


if __name__ == "__main__":
    main()
