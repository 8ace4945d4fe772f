#!/usr/bin/env python3
"""Reader-only benchmark: inflate + record walk + field pass of the BAM reader (no GPU), groups/s and GB BAM/s.
Usage: python tools/read_bench.py [--groups N] [--threads T] [--batch B] [--platform hifi|ont] [--dir D]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=16384)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--platform", default="hifi")
    ap.add_argument("--dir", default="/dev/shm/spx")
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    from secphase_amd import api, records, synth
    L = api.lib()
    os.makedirs(args.dir, exist_ok=True)
    bam = os.path.join(args.dir, f"rb_{args.platform}_{args.groups}.bam")
    if not os.path.exists(bam):
        cfg = synth.default_cfg(synth.ONT if args.platform == "ont" else synth.HIFI, n_contigs=4, contig_len=2000000)
        g = synth.Genome(cfg)
        t0 = time.time()
        chunks = [g.reads(i, min(1024, args.groups - i)) for i in range(0, args.groups, 1024)]
        t1 = time.time()
        synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=args.threads)
        print(f"generated in {t1 - t0:.1f} s, written in {time.time() - t1:.1f} s", file=sys.stderr)
    size = os.path.getsize(bam)
    out = []
    for _ in range(args.repeat):
        rd = C.c_void_p()
        t0 = time.perf_counter()
        assert L.spx_bam_open(bam.encode(), args.threads, C.byref(rd)) == 0
        n = na = 0
        while True:
            bp = C.POINTER(records.SpxBatch)()
            k = L.spx_bam_next_batch(rd, args.batch, C.byref(bp))
            assert k >= 0, L.spx_io_last_error()
            if k == 0:
                break
            n += k
            na += bp.contents.n_alns
            L.spx_bam_release_batch(rd, bp)
        dt = time.perf_counter() - t0
        L.spx_bam_close(rd)
        out.append({"groups": n, "records": na, "s": round(dt, 3), "groups_per_s": round(n / dt), "gb_bam_per_s": round(size / dt / 1e9, 3)})
    print(json.dumps({"bam_bytes": size, "threads": args.threads, "batch": args.batch, "runs": out}))


if __name__ == "__main__":
    main()
