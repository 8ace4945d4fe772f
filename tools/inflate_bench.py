#!/usr/bin/env python3
"""Device BGZF inflate alone: GB/s of inflated bytes for the blocks of a synthetic BAM (kernel time from HIP events).
Usage: python tools/inflate_bench.py [--groups N] [--platform hifi|ont]"""
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
    ap.add_argument("--groups", type=int, default=8192)
    ap.add_argument("--platform", default="hifi")
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    from secphase_amd import api, synth
    L = api.lib()
    cfg = synth.default_cfg(synth.ONT if args.platform == "ont" else synth.HIFI, n_contigs=4, contig_len=2000000)
    g = synth.Genome(cfg)
    chunks = [g.reads(i, min(1024, args.groups - i)) for i in range(0, args.groups, 1024)]
    bam = f"/dev/shm/spx_inflate_bench_{os.getpid()}.bam"
    synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=16)
    blob = open(bam, "rb").read()
    os.unlink(bam)
    offs, at = [0], 0
    while at < len(blob):
        at += (blob[at + 16] | (blob[at + 17] << 8)) + 1
        offs.append(at)
    n = len(offs) - 1
    L.spx_inflate_bgzf_device.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.c_int32, C.c_void_p, C.c_int64,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.spx_inflate_bgzf_device.restype = C.c_int64
    ctx = api.Context(0)
    cap = 65536 * n
    out = C.create_string_buffer(cap)
    st = (C.c_int32 * n)()
    ms = C.c_double()
    res = []
    for _ in range(args.repeat):
        t0 = time.perf_counter()
        got = L.spx_inflate_bgzf_device(ctx.h, blob, (C.c_int64 * (n + 1))(*offs), n, out, cap, st, C.byref(ms))
        wall = time.perf_counter() - t0
        assert int(os.environ.get('SPX_INFLATE_TOK_STAGE', '3')) < 3 or (got > 0 and all(s == 0 for s in st)), (got, [s for s in st if s][:4])
        res.append({"kernel_ms": round(ms.value, 3), "inflated_GB_per_s": round(got / ms.value / 1e6, 2),
                    "compressed_GB_per_s": round(len(blob) / ms.value / 1e6, 2), "wall_s": round(wall, 3)})
    print(json.dumps({"blocks": n, "compressed_bytes": len(blob), "inflated_bytes": got, "groups": args.groups, "runs": res}))


if __name__ == "__main__":
    main()
