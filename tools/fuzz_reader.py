"""CPU fuzz of the BAM reader on DAMAGED input (it parses untrusted files): the uncompressed bytes of a small valid BAM are
mutated (bit flips, overwritten length fields, truncation, inserted / deleted bytes) and re-packed into valid BGZF blocks
with correct CRCs, so that the damage reaches the record chain, the field / tag / CIGAR parsers and the index builder.  Every
file must end in batches or in an error -- never in a crash, a hang or (under AddressSanitizer, tools/sanitize_cpu.sh) a
report.  Every batch the reader does hand out goes through spx_plan_create as well: the host plan runs the SAME source
(spx_logic.h) as the preparation kernels, so a walk that leaves its arrays on damaged CIGAR / cs / MD content shows up here, on
the CPU, under the sanitizer -- instead of as a GPU fault.  Usage: python tools/fuzz_reader.py [seed] [seconds] [gpu]
"gpu" (needs an MI355X): every damaged file ALSO goes through the device-resident input (spx_dbam: inflate, record chain,
fields / tags, groups, dispatch filter, staging as kernels) with small segments, and through scoring on both paths: the two
must agree -- both refuse the file, or both score every group identically (names, flags, positions, scores).  A kernel that
walks off its buffers on damaged bytes would end the process with a GPU memory fault."""
import ctypes as C
import gzip
import os
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bamio import write_bgzf  # noqa: E402
from common import small_genome  # noqa: E402
from secphase_amd import api, records, synth  # noqa: E402


class BamOptions(C.Structure):
    _fields_ = [("threads", C.c_int32), ("batch_groups", C.c_int32), ("ahead_batches", C.c_int32), ("flags", C.c_int32),
                ("chunk_bytes", C.c_int64), ("max_bytes", C.c_int64), ("start_voffset", C.c_int64), ("end_voffset", C.c_int64),
                ("keep_batches", C.c_int32), ("reserved", C.c_int32)]


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60
    gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"
    rng = np.random.default_rng(seed)
    L = api.lib()
    vp = C.c_void_p
    L.spx_bam_open_opts.argtypes = [C.c_char_p, C.POINTER(BamOptions), C.POINTER(vp)]
    L.spx_bam_default_options.argtypes = [C.POINTER(BamOptions)]
    L.spx_bam_next_batch.argtypes = [vp, C.c_int32, C.POINTER(C.POINTER(records.SpxBatch))]
    L.spx_bam_close.argtypes = [vp]
    L.spx_bam_close.restype = None
    L.spx_bam_bind_reference.argtypes = [vp, C.POINTER(records.SpxRef)]
    L.spx_bam_index_build.argtypes = [C.c_char_p, C.c_int, C.c_int32, C.POINTER(C.c_int64), C.c_int64]
    L.spx_bam_index_build.restype = C.c_int64
    g = small_genome(synth.HIFI, read_len=1500, max_secondaries=3, n_paralogs=3, hardclip_frac=0.3, softclip_frac=0.3)
    d = tempfile.mkdtemp(prefix="spx_fuzz_reader_")
    good = os.path.join(d, "good.bam")
    reads = g.reads(0, 60)  # (kept alive: the writer reads its memory)
    synth.write_bam(good, [reads.batch], g.ref, threads=2)
    raw = gzip.open(good).read()
    l_text = struct.unpack_from("<i", raw, 4)[0]
    n_ref = struct.unpack_from("<i", raw, 8 + l_text)[0]
    at = 12 + l_text
    for _ in range(n_ref):
        at += 8 + struct.unpack_from("<i", raw, at)[0]
    first_rec = at
    ctx = None
    if gpu:
        ctx = api.Context(0)
        ctx.set_reference(g.ref)
    dev_stats = {"both_ok": 0, "both_error": 0, "groups": 0}

    def key(o):
        return (o.n_aln, tuple(o.score[a] for a in range(max(o.n_aln, 0))), o.prim_idx, o.max_idx, o.tie_mask, o.pass_, o.n_problems, o.dp_cells)

    def names(bp):
        bt = bp.contents
        return [(C.string_at(bt.qnames + bt.qname_off[k]), tuple((bt.flag[a], bt.tid[a], bt.pos[a]) for a in range(bt.grp_first[k], bt.grp_first[k + 1])))
                for k in range(bt.n_groups)]

    def device_side(path, par):
        os.environ["SPX_DIN_SEG_KB"] = str(int(rng.choice([64, 100, 256, 4096])))
        os.environ["SPX_DIN_CARRY_KB"] = "4096"
        res, nms = [], []
        try:
            d_ = api.DeviceBam(path, [ctx], par, g.ref, max_groups=int(rng.choice([5, 95000])), host_inflate_percent=int(rng.choice([0, 50, 100])))
        except api.SpxError:
            return None, None
        try:
            while True:
                nx = d_.next()
                if nx is None:
                    break
                w, _, nb, cnt = nx
                w.prepare_staged()
                w.launch()
                out = w.collect(finalize_seed=None)
                res += [key(out[k]) for k in range(cnt)]
                nms += names(nb)
                w.free()
                d_.release(nb)
        except api.SpxError:
            res = None
        finally:
            d_.close()
        return res, nms

    t0, n, outcomes = time.time(), 0, {"records": 0, "error": 0}
    plans, plan_err = 0, 0
    presets = [records.preset("hifi"), records.preset("ont", bandwidth=50)]
    bad = os.path.join(d, "bad.bam")
    while time.time() - t0 < seconds:
        b = bytearray(raw)
        kind = rng.random()
        for _ in range(int(rng.integers(1, 6))):
            if len(b) < 8:
                break
            pos = int(rng.integers(0, len(b))) if rng.random() < 0.3 or first_rec >= len(b) else int(rng.integers(first_rec, len(b)))
            if kind < 0.4:
                b[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind < 0.6:  # a plausible place of a length field: overwrite 4 bytes
                struct.pack_into("<i", b, min(pos, len(b) - 4), int(rng.choice([-1, 0, 1, 31, 32, 255, 65535, 65536, 2 ** 31 - 1, -2 ** 31, int(rng.integers(-10 ** 6, 10 ** 6))])))
            elif kind < 0.75:
                del b[pos:pos + int(rng.integers(1, 64))]
            elif kind < 0.9:
                b[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
            else:
                del b[pos:]
        write_bgzf(bad, bytes(b), block=int(rng.choice([200, 4096, 60000])))
        o = BamOptions()
        L.spx_bam_default_options(C.byref(o))
        o.threads, o.chunk_bytes, o.batch_groups, o.ahead_batches = 3, int(rng.choice([65536, 1 << 20])), int(rng.choice([0, 7])), 2
        rd = vp()
        if L.spx_bam_open_opts(bad.encode(), C.byref(o), C.byref(rd)) != 0:
            outcomes["error"] += 1
        else:
            L.spx_bam_bind_reference(rd, g.ref)
            ok = True
            host_res, host_names = [], []
            for _ in range(200):
                bp = C.POINTER(records.SpxBatch)()
                k = L.spx_bam_next_batch(rd, 7, C.byref(bp))
                if k <= 0:
                    ok = k == 0
                    break
                bt = bp.contents  # touch what a consumer touches
                for a in range(bt.n_alns):
                    lq = bt.l_qseq[a]
                    if lq > 0:
                        _ = bt.qual[bt.qual_off[a] + lq - 1] + bt.seq4[bt.seq_off[a] + (lq - 1) // 2]
                    if bt.cs_off[a] >= 0:
                        C.string_at(bt.cs + bt.cs_off[a])
                    for c_ in range(bt.n_cigar[a]):
                        _ = bt.cigar[bt.cigar_off[a] + c_]
                if gpu:
                    out_, _ = ctx.score_batch(bp, presets[n % 2], finalize_seed=None)
                    host_res += [key(out_[q]) for q in range(k)]
                    host_names += names(bp)
                h = vp()
                rc = L.spx_plan_create(g.ref, bp, C.byref(presets[n % 2]), C.byref(h))
                plans += 1
                if rc != 0:
                    plan_err += 1
                else:
                    L.spx_plan_free(h)
            outcomes["records" if ok else "error"] += 1
            L.spx_bam_close(rd)
            if gpu:
                dres, dnames = device_side(bad, presets[n % 2])
                if ok:
                    assert dres is not None, f"seed {seed}, file {n}: the host reader accepts the file, the device input refuses it"
                    assert dres == host_res and dnames == host_names, f"seed {seed}, file {n}: device input and host reader disagree"
                    dev_stats["both_ok"] += 1
                    dev_stats["groups"] += len(dres)
                else:
                    assert dres is None, f"seed {seed}, file {n}: the host reader refuses the file, the device input accepts it"
                    dev_stats["both_error"] += 1
        if rng.random() < 0.1:
            off = (C.c_int64 * 64)()
            L.spx_bam_index_build(bad.encode(), 2, 5, off, 64)
        n += 1
    if gpu:
        print(f"device input vs host reader: {dev_stats['both_ok']} files accepted by both ({dev_stats['groups']} groups, identical results), "
              f"{dev_stats['both_error']} refused by both")
    print(f"reader fuzz: seed {seed}, {n} damaged files ({outcomes['records']} read to the end, {outcomes['error']} ended in an error); "
          f"{plans} batches through the host plan ({plan_err} rejected); no crash, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
