#!/usr/bin/env python3
"""End-to-end run of the command-line drop-in on a synthetic BAM: wall time from process start to all
outputs written, groups/s and GB (compressed BAM)/s, next to the CPU oracle on the same groups; the relabel list
is checked against the oracle's on a prefix of the file (same rand() stream: the prefix of the list must be identical).
Usage: python tools/e2e_cli.py [--groups N] [--platform hifi|ont] [--threads T] [--devices 0] [--dir /dev/shm]"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def gen_chunks(g, n, chunk, threads):
    starts = list(range(0, n, chunk))
    out = [None] * len(starts)
    it = iter(range(len(starts)))
    lock = threading.Lock()

    def run():
        while True:
            with lock:
                k = next(it, None)
            if k is None:
                return
            out[k] = g.reads(starts[k], min(chunk, n - starts[k]))

    ths = [threading.Thread(target=run) for _ in range(max(1, threads))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=8192)
    ap.add_argument("--platform", default="hifi")
    ap.add_argument("--threads", type=int, default=min(64, os.cpu_count() or 8))
    ap.add_argument("--batch", type=int, default=0, help="--groupsPerBatch of the command line (0: its default)")
    ap.add_argument("--host-input", action="store_true", help="--hostInput: the round-3 host reader instead of the device-resident input")
    ap.add_argument("--runs", type=int, default=2, help="runs of the command line; the best one is reported")
    ap.add_argument("--env", action="append", default=[], help="NAME=VALUE for the command line's environment")
    ap.add_argument("--rocprof", default="", metavar="DIR", help="after the runs: the same command once under `rocprofv3 --kernel-trace --stats` "
                    "(the binary itself after `--`), its CSV files under DIR")
    ap.add_argument("--sweep", action="append", default=[], help="NAME=v1,v2,... (may be given several times; NAME may be A+B: both variables get the value pair a+b): "
                    "after the main run, the same command with each value of NAME in its environment (--runs each); their wall / loop times are "
                    "reported under `sweep`")
    ap.add_argument("--check-groups", type=int, default=4096)
    ap.add_argument("--devices", default="0")
    ap.add_argument("--dir", default="/dev/shm" if os.path.isdir("/dev/shm") else None)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--gpu-inflate", type=int, default=-1)
    ap.add_argument("--cli-threads", type=int, default=0, help="-@ of the command line (default: the CPUs the container may use)")
    ap.add_argument("--tidy", action="store_true")
    args = ap.parse_args()
    from oracle import orc
    from secphase_amd import api, records, synth
    cli_threads = args.cli_threads or max(1, min(args.threads, api.lib().spx_effective_cpus()))
    ont = args.platform == "ont"
    cfg = synth.default_cfg(synth.ONT if ont else synth.HIFI)
    g = synth.Genome(cfg)
    t0 = time.time()
    chunks = gen_chunks(g, args.groups, 1024, args.threads)
    t_gen = time.time() - t0
    d = tempfile.mkdtemp(prefix="spx_e2e_", dir=args.dir)
    fa, bam, outd = os.path.join(d, "asm.fa"), os.path.join(d, "reads.bam"), os.path.join(d, "out")
    synth.write_fasta(fa, g.ref)
    t0 = time.time()
    synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=args.threads)
    t_write = time.time() - t0
    exe = os.path.join(ROOT, "secphase_amd", "bin", "secphase")
    flags = ["--ont", "-b", "50"] if ont else ["--hifi"]
    cmd = [exe] + flags + ["-@", str(cli_threads), "-i", bam, "-f", fa, "--outDir", outd, "--prefix", "e2e", "--devices", args.devices]
    if args.batch > 0:
        cmd += ["--groupsPerBatch", str(args.batch)]
    if args.host_input:
        cmd += ["--hostInput"]
    if args.gpu_inflate >= 0:
        cmd += ["--gpuInflate", str(args.gpu_inflate)]
    env = dict(os.environ, SPX_TIMING="1", **({"SPX_TIDY_EXIT": "1"} if args.tidy else {}))
    for kv in args.env:
        k, _, v = kv.partition("=")
        env[k] = v
    best = None
    walls = []
    for _ in range(max(1, args.runs)):
        shutil.rmtree(outd, ignore_errors=True)
        t0 = time.time()
        p = subprocess.run(cmd, capture_output=True, text=True, env=env)
        t1 = time.time()
        walls.append(round(t1 - t0, 3))
        if p.returncode != 0:
            sys.exit(p.stderr[-3000:])
        if best is None or t1 - t0 < best[0]:
            best = (t1 - t0, p, t0, t1)
    wall, p, t0, t1 = best
    sweep = {}
    for sw in args.sweep:
        import re
        name, _, vals = sw.partition("=")
        for v in vals.split(","):
            ws, loops = [], []
            extra = dict(zip(name.split("+"), v.split("+")))
            for _ in range(max(1, args.runs)):
                shutil.rmtree(outd, ignore_errors=True)
                ta = time.time()
                q = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, **extra))
                ws.append(round(time.time() - ta, 3))
                m = re.search(r"time in the scoring loop: ([0-9.]+) s", q.stderr)
                loops.append(float(m.group(1)) if m else None)
                if q.returncode != 0:
                    ws[-1] = "rc %d: %s" % (q.returncode, q.stderr[-200:])
            sweep[f"{name}={v}"] = {"wall_s": ws, "loop_s": loops}
    if args.sweep:
        shutil.rmtree(outd, ignore_errors=True)
        subprocess.run(cmd, capture_output=True, text=True, env=env)  # (the checks below read the main configuration's list)
    outside = None
    for l in p.stderr.splitlines():
        if "main() entered at" in l:
            a, b = [float(x) for x in l.replace(",", " ").split() if x.replace(".", "").isdigit() and "." in x][:2]
            outside = {"exec_to_main_s": round(a - t0, 3), "exit_to_parent_s": round(t1 - b, 3)}
    if p.returncode != 0:
        sys.exit(p.stderr[-3000:])
    if args.rocprof:
        os.makedirs(args.rocprof, exist_ok=True)
        rp = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", os.path.abspath(args.rocprof), "-o", "cli", "--output-format", "csv", "--"] + cmd[:-0 or None],
                            capture_output=True, text=True, env=dict(env, TMPDIR="/tmp", SPX_PROFILER="1", SPX_QUICK_EXIT="0"), cwd="/tmp")
        if rp.returncode != 0:
            print("rocprofv3 run failed: " + rp.stderr[-400:], file=sys.stderr)
        shutil.rmtree(outd, ignore_errors=True)
        subprocess.run(cmd, capture_output=True, text=True, env=env)
    params = records.preset("ont", bandwidth=50) if ont else records.preset("hifi")
    ncheck = min(args.check_groups, args.groups)
    ncheck = (ncheck // 1024) * 1024 or min(ncheck, 1024)
    same = None
    cpu = None
    if ncheck > 0:
        sub = g.reads(0, ncheck)
        log_o = os.path.join(d, "oracle.log")
        t0 = time.time()
        orc.run_batch(sub.batch, g.ref, params, threads=args.threads, seed=1, log_path=log_o, reuse_scratch=True)
        cpu = time.time() - t0
        want = open(log_o, "rb").read()
        got = open(os.path.join(outd, "e2e.out.log"), "rb").read()
        same = got[:len(want)] == want and (args.groups > ncheck or len(got) == len(want))
    size = os.path.getsize(bam)
    keep = ("start-up", "time in the scoring loop", "finalise", "wind-down", "inflate chunks", "reader closed", "CPU time", "device input:", "hipMalloc", "closed beside", "BED ")
    print(json.dumps({"groups": args.groups, "platform": args.platform, "devices": args.devices, "bam_bytes": size, "wall_s": round(wall, 3), "runs_wall_s": walls, "host_input": args.host_input,
                      "groups_per_s": round(args.groups / wall, 1), "GB_bam_per_s": round(size / wall / 1e9, 4),
                      "cpu_oracle_groups_per_s": round(ncheck / cpu, 1) if cpu else None, "cpu_threads": args.threads, "cli_threads": cli_threads,
                      "outside_main": outside, "sweep": sweep, "out_log_identical_to_oracle": same, "checked_groups": ncheck, "generate_s": round(t_gen, 1), "bam_write_s": round(t_write, 1),
                      "stderr_tail": [l for l in p.stderr.strip().splitlines() if any(k in l for k in keep)]}, indent=0))
    if not args.keep:
        shutil.rmtree(d, ignore_errors=True)
    if same is False:
        sys.exit("out.log differs from the oracle's")


if __name__ == "__main__":
    main()
