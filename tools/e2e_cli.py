#!/usr/bin/env python3
"""End-to-end run of the command-line drop-in on a synthetic BAM: wall time from process start to all
outputs written, groups/s and GB (compressed BAM)/s, next to the CPU oracle on the same groups.
Usage: python tools/e2e_cli.py [--groups N] [--platform hifi|ont] [--threads T]"""
import argparse
import filecmp
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=8192)
    ap.add_argument("--platform", default="hifi")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--cpu-groups", type=int, default=1024)
    args = ap.parse_args()
    import bamio
    import zlib
    from oracle import orc
    from secphase_amd import records, synth
    ont = args.platform == "ont"
    cfg = synth.default_cfg(synth.ONT if ont else synth.HIFI, n_contigs=4, contig_len=2000000)
    g = synth.Genome(cfg)
    r = g.reads(0, args.groups)
    d = tempfile.mkdtemp(prefix="spx_e2e_")
    fa, bam, outd = os.path.join(d, "asm.fa"), os.path.join(d, "reads.bam"), os.path.join(d, "out")
    bamio.write_fasta(fa, g.ref)
    t0 = time.time()
    bamio.write_bam(bam, r.batch, g.ref)
    t_write = time.time() - t0
    exe = os.path.join(ROOT, "secphase_amd", "bin", "secphase")
    flags = ["--ont", "-b", "50"] if ont else ["--hifi"]
    t0 = time.time()
    p = subprocess.run([exe] + flags + ["-@", str(args.threads), "-i", bam, "-f", fa, "--outDir", outd, "--prefix", "e2e",
                                        "--groupsPerBatch", str(args.batch)], capture_output=True, text=True, env=dict(os.environ, SPX_TIMING="1"))
    wall = time.time() - t0
    if p.returncode != 0:
        sys.exit(p.stderr)
    params = records.preset("ont", bandwidth=50) if ont else records.preset("hifi")
    sub = g.reads(0, min(args.cpu_groups, args.groups))
    t0 = time.time()
    orc.run_batch(sub.batch, g.ref, params, threads=args.threads, seed=1, reuse_scratch=True)
    cpu = time.time() - t0
    log_o = os.path.join(d, "oracle.log")
    same = None
    if args.groups <= 4096:
        orc.run_batch(r.batch, g.ref, params, threads=args.threads, seed=1, log_path=log_o, reuse_scratch=True)
        same = filecmp.cmp(log_o, os.path.join(outd, "e2e.out.log"), shallow=False)
    size = os.path.getsize(bam)
    print(json.dumps({"groups": args.groups, "platform": args.platform, "bam_bytes": size, "wall_s": round(wall, 3),
                      "groups_per_s": round(args.groups / wall, 1), "GB_bam_per_s": round(size / wall / 1e9, 4),
                      "cpu_oracle_groups_per_s": round(sub.batch.contents.n_groups / cpu, 1), "cpu_threads": args.threads,
                      "out_log_identical_to_oracle": same, "bam_write_s": round(t_write, 1),
                      "stderr_tail": [l for l in p.stderr.strip().splitlines() if "time in the scoring loop" in l or "finalise+write:" in l]}, indent=0))


if __name__ == "__main__":
    main()
