"""CPU: the zlib-only BAM / FASTA readers of libspx hand back exactly the records that were written."""
import ctypes as C

import numpy as np

from bamio import _bgzf_block, contigs_of, sam_text, write_bam, write_bgzf, write_fasta
from common import small_genome
from secphase_amd import api, records, synth


def _declare(L):
    vp = C.c_void_p
    L.spx_bam_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.spx_bam_next_batch.argtypes = [vp, C.c_int32, C.POINTER(C.POINTER(records.SpxBatch))]
    L.spx_bam_bind_reference.argtypes = [vp, C.POINTER(records.SpxRef)]
    L.spx_bam_close.argtypes = [vp]
    L.spx_bam_close.restype = None
    L.spx_fasta_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.spx_fasta_ref.argtypes = [vp]
    L.spx_fasta_ref.restype = C.POINTER(records.SpxRef)
    L.spx_fasta_free.argtypes = [vp]
    L.spx_fasta_free.restype = None
    L.spx_io_last_error.restype = C.c_char_p


def _records(bp):
    b = bp.contents
    out = []
    for g in range(b.n_groups):
        name = C.string_at(b.qnames + b.qname_off[g])
        recs = []
        for a in range(b.grp_first[g], b.grp_first[g + 1]):
            lq, nc = b.l_qseq[a], b.n_cigar[a]
            recs.append((b.flag[a], b.tid[a], b.pos[a], lq,
                         tuple(b.cigar[b.cigar_off[a] + k] for k in range(nc)),
                         C.string_at(C.addressof(b.seq4.contents) + b.seq_off[a], (lq + 1) // 2),
                         C.string_at(C.addressof(b.qual.contents) + b.qual_off[a], lq),
                         C.string_at(b.cs + b.cs_off[a]) if b.cs_off[a] >= 0 else None,
                         C.string_at(b.md + b.md_off[a]) if (b.md_off and b.cs_off[a] < 0 and b.md_off[a] >= 0) else None))
        out.append((name, recs))
    return out


def test_fasta_and_bam_round_trip(built, tmp_path):
    L = api.lib()
    _declare(L)
    _round_trip(tmp_path, 0)


def test_bam_round_trip_md_only(built, tmp_path):
    _round_trip(tmp_path, 1)


def _round_trip(tmp_path, tag_mode):
    L = api.lib()
    g = small_genome(synth.HIFI, read_len=3000, max_secondaries=3, n_paralogs=2, hardclip_frac=0.3, softclip_frac=0.4,
                     tag_mode=tag_mode)
    r = g.reads(0, 37)
    fa, bam = str(tmp_path / "asm.fa"), str(tmp_path / "reads.bam")
    write_fasta(fa, g.ref)
    # header lists the contigs in reverse order: the reader must map target ids by NAME
    order = list(range(g.ref.contents.n_contigs))[::-1]
    write_bam(bam, r.batch, g.ref, contig_order=order)
    fh = C.c_void_p()
    assert L.spx_fasta_load(fa.encode(), C.byref(fh)) == 0, L.spx_io_last_error()
    ref2 = L.spx_fasta_ref(fh)
    assert contigs_of(ref2) == contigs_of(g.ref)
    rd = C.c_void_p()
    assert L.spx_bam_open(bam.encode(), 3, C.byref(rd)) == 0, L.spx_io_last_error()
    assert L.spx_bam_bind_reference(rd, ref2) == 0
    got = []
    while True:
        bp = C.POINTER(records.SpxBatch)()
        n = L.spx_bam_next_batch(rd, 10, C.byref(bp))
        assert n >= 0, L.spx_io_last_error()
        if n == 0:
            break
        assert n <= 10
        got += _records(bp)
    L.spx_bam_close(rd)
    L.spx_fasta_free(fh)
    assert got == _records(r.batch)



def test_bam_round_trip_large_batches_on_threads(built, tmp_path):
    """batches of thousands of records: the field / tag / copy passes of the reader run on threads (they are serial below
    2 048 records per thread), batches end on group boundaries, short reads keep the file small"""
    L = api.lib()
    _declare(L)
    g = small_genome(synth.HIFI, read_len=300, max_secondaries=3, min_secondaries=1, n_paralogs=3, hardclip_frac=0.3, softclip_frac=0.3)
    r = g.reads(0, 5000)
    fa, bam = str(tmp_path / "asm.fa"), str(tmp_path / "reads.bam")
    write_fasta(fa, g.ref)
    write_bam(bam, r.batch, g.ref)
    fh = C.c_void_p()
    assert L.spx_fasta_load(fa.encode(), C.byref(fh)) == 0, L.spx_io_last_error()
    ref2 = L.spx_fasta_ref(fh)
    rd = C.c_void_p()
    assert L.spx_bam_open(bam.encode(), 8, C.byref(rd)) == 0, L.spx_io_last_error()
    assert L.spx_bam_bind_reference(rd, ref2) == 0
    got = []
    sizes = []
    while True:
        bp = C.POINTER(records.SpxBatch)()
        n = L.spx_bam_next_batch(rd, 3000, C.byref(bp))
        assert n >= 0, L.spx_io_last_error()
        if n == 0:
            break
        sizes.append((n, bp.contents.n_alns))
        got += _records(bp)
    L.spx_bam_close(rd)
    L.spx_fasta_free(fh)
    assert sizes[0][0] == 3000 and sizes[0][1] >= 4096  # enough records for more than one thread
    assert got == _records(r.batch)


def _sam_decl(L):
    vp = C.c_void_p
    L.spx_sam_open.argtypes = [C.c_char_p, vp, C.POINTER(vp)]
    L.spx_sam_write_group.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_uint8)]
    L.spx_sam_close.argtypes = [vp]


def test_sam_writer_matches_spec_formatter(built, tmp_path):
    """-w/--writeBam output format: sam_open(path, "w") = SAM text (src/secphase.c:643-652); every group, own qualities"""
    L = api.lib()
    _declare(L)
    _sam_decl(L)
    g = small_genome(synth.HIFI, read_len=2000, max_secondaries=3, n_paralogs=2, hardclip_frac=0.3, softclip_frac=0.4,
                     tag_mode=2)
    r = g.reads(0, 23)
    bam, sam = str(tmp_path / "reads.bam"), str(tmp_path / "out.sam")
    order = list(range(g.ref.contents.n_contigs))[::-1]
    write_bam(bam, r.batch, g.ref, contig_order=order)
    rd, wr = C.c_void_p(), C.c_void_p()
    assert L.spx_bam_open(bam.encode(), 2, C.byref(rd)) == 0, L.spx_io_last_error()
    assert L.spx_sam_open(sam.encode(), rd, C.byref(wr)) == 0, L.spx_io_last_error()
    while True:
        bp = C.POINTER(records.SpxBatch)()
        n = L.spx_bam_next_batch(rd, 7, C.byref(bp))
        assert n >= 0
        if n == 0:
            break
        for k in range(n):
            assert L.spx_sam_write_group(wr, rd, k, None) == bp.contents.grp_first[k + 1] - bp.contents.grp_first[k]
    assert L.spx_sam_close(wr) == 0
    L.spx_bam_close(rd)
    assert open(sam).read() == sam_text(r.batch, g.ref, contig_order=order)


def test_sam_writer_aux_types_and_missing_fields(built, tmp_path):
    """every aux type of the SAM specification, mate fields, a record without SEQ, a header without @SQ lines"""
    import struct
    L = api.lib()
    _declare(L)
    _sam_decl(L)
    text = b"@HD\tVN:1.6\n@PG\tID:x"           # no @SQ line, no trailing newline
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 2)
    for nm, ln in ((b"c1", 1000), (b"c2", 2000)):
        hdr += struct.pack("<i", len(nm) + 1) + nm + b"\0" + struct.pack("<i", ln)

    def rec(name, flag, tid, pos, mapq, cig, seq, qual, mtid, mpos, tlen, aux):
        nt = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
        sq = bytearray((len(seq) + 1) // 2)
        for k, c in enumerate(seq):
            sq[k >> 1] |= nt[c] << (4 if k % 2 == 0 else 0)
        cg = b"".join(struct.pack("<I", (n << 4) | "MIDNSHP=X".index(o)) for n, o in cig)
        qn = name + b"\0"
        core = struct.pack("<iiBBHHHiiii", tid, pos, len(qn), mapq, 0, len(cig), flag, len(seq), mtid, mpos, tlen)
        body = core + qn + cg + bytes(sq) + bytes(qual) + aux
        return struct.pack("<i", len(body)) + body

    aux = (b"XAAq" + b"Xcc" + struct.pack("<b", -5) + b"XCC" + struct.pack("<B", 200) + b"Xss" + struct.pack("<h", -300) +
           b"XSS" + struct.pack("<H", 60000) + b"Xii" + struct.pack("<i", -70000) + b"XII" + struct.pack("<I", 4000000000) +
           b"Xff" + struct.pack("<f", 0.5) + b"Xgf" + struct.pack("<f", 1e-7) + b"XZZhello world\0" + b"XHH1AE3\0" +
           b"B1Bc" + struct.pack("<ibb", 2, -1, 2) + b"B2BC" + struct.pack("<iB", 1, 255) + b"B3BS" + struct.pack("<iH", 1, 65535) +
           b"B4Bi" + struct.pack("<ii", 1, -9) + b"B5BI" + struct.pack("<iI", 1, 9) + b"B6Bf" + struct.pack("<iff", 2, 1.5, -2.25) +
           b"B7Bs" + struct.pack("<i", 0))
    body = (rec(b"r1", 0, 0, 99, 37, [(3, "S"), (4, "M"), (1, "I"), (2, "D"), (2, "=")], "ACGTNACGTA", [10, 20, 30, 40, 0, 1, 2, 3, 93, 50],
                1, 499, -321, aux) +
            rec(b"r1", 256, 1, 0, 0, [(5, "M")], "ACGTA", [0xff] * 5, 1, 7, 0, b"") +
            rec(b"r2", 16, 1, 5, 255, [(2, "H"), (3, "X")], "", [], -1, -1, 0, b"NMC" + struct.pack("<B", 0)))
    bam, sam = str(tmp_path / "x.bam"), str(tmp_path / "x.sam")
    write_bgzf(bam, hdr + body)
    rd, wr = C.c_void_p(), C.c_void_p()
    assert L.spx_bam_open(bam.encode(), 1, C.byref(rd)) == 0, L.spx_io_last_error()
    assert L.spx_sam_open(sam.encode(), rd, C.byref(wr)) == 0
    bp = C.POINTER(records.SpxBatch)()
    assert L.spx_bam_next_batch(rd, 10, C.byref(bp)) == 2
    assert L.spx_sam_write_group(wr, rd, 0, None) == 2
    assert L.spx_sam_write_group(wr, rd, 1, None) == 1
    assert L.spx_sam_write_group(wr, rd, 2, None) < 0
    assert L.spx_sam_close(wr) == 0
    L.spx_bam_close(rd)
    want = ("@HD\tVN:1.6\n@PG\tID:x\n@SQ\tSN:c1\tLN:1000\n@SQ\tSN:c2\tLN:2000\n"
            "r1\t0\tc1\t100\t37\t3S4M1I2D2=\tc2\t500\t-321\tACGTNACGTA\t+5?I!\"#$~S\t"
            "XA:A:q\tXc:i:-5\tXC:i:200\tXs:i:-300\tXS:i:60000\tXi:i:-70000\tXI:i:4000000000\tXf:f:0.5\tXg:f:1e-07\t"
            "XZ:Z:hello world\tXH:H:1AE3\tB1:B:c,-1,2\tB2:B:C,255\tB3:B:S,65535\tB4:B:i,-9\tB5:B:I,9\tB6:B:f,1.5,-2.25\tB7:B:s\n"
            "r1\t256\tc2\t1\t0\t5M\t=\t8\t0\tACGTA\t*\n"
            "r2\t16\tc2\t6\t255\t2H3X\t*\t0\t0\t*\t*\tNM:i:0\n")
    assert open(sam).read() == want


def _raw_bam(path, records_bytes, contigs=((b"c0", 1000),)):
    import struct
    text = b"@HD\tVN:1.6\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % c for c in contigs)
    out = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(contigs)))
    for nm, ln in contigs:
        out += struct.pack("<i", len(nm) + 1) + nm + b"\0" + struct.pack("<i", ln)
    for rec in records_bytes:
        out += struct.pack("<i", len(rec)) + rec
    write_bgzf(path, bytes(out))


def _raw_record(name, flag, pos, cigar, seq_len, aux=b"", l_name=None, n_cigar=None, l_seq=None):
    import struct
    qn = name + b"\0"
    cig = b"".join(struct.pack("<I", (ln << 4) | op) for ln, op in cigar)
    core = struct.pack("<iiBBHHHiiii", 0, pos, len(qn) if l_name is None else l_name, 60, 4680,
                       len(cigar) if n_cigar is None else n_cigar, flag, seq_len if l_seq is None else l_seq, -1, -1, 0)
    return core + qn + cig + bytes((seq_len + 1) // 2) + bytes([30]) * seq_len + aux


def test_bam_long_cigar_is_taken_from_the_cg_tag(built, tmp_path):
    """> 65535 CIGAR operations: BAM stores <l_seq>S<ref_len>N and the real CIGAR in CG:B,I; htslib's sam_read1
    restores it before secphase's loop (src/secphase.c:268) sees the record"""
    import struct
    L = api.lib()
    _declare(L)
    real = [(5, 4), (20, 0), (2, 1), (13, 0)]  # 5S20M2I13M
    cg = b"CGBI" + struct.pack("<i", len(real)) + b"".join(struct.pack("<I", (ln << 4) | op) for ln, op in real)
    rec = _raw_record(b"r1", 0, 7, [(40, 4), (33, 3)], 40, aux=b"NMi" + struct.pack("<i", 1) + cg + b"csZ:20+ac:13\0")
    plain = _raw_record(b"r1", 256, 9, [(40, 0)], 40, aux=b"csZ:40\0")
    bam = str(tmp_path / "cg.bam")
    _raw_bam(bam, [rec, plain])
    h = C.c_void_p()
    assert L.spx_bam_open(bam.encode(), 1, C.byref(h)) == 0
    bp = C.POINTER(records.SpxBatch)()
    assert L.spx_bam_next_batch(h, 8, C.byref(bp)) == 1
    got = _records(bp)
    assert got[0][1][0][4] == tuple((ln << 4) | op for ln, op in real)
    assert got[0][1][1][4] == ((40 << 4),)
    assert got[0][1][0][7] == b":20+ac:13"
    L.spx_bam_close(h)


def test_bam_records_with_impossible_lengths_are_rejected(built, tmp_path):
    """n_cigar / l_seq / l_read_name reaching past block_size, or a name without its NUL: SPX_EINVAL, no wild reads"""
    L = api.lib()
    _declare(L)
    good = _raw_record(b"ok", 0, 1, [(10, 0)], 10, aux=b"csZ:10\0")
    for k, bad in enumerate((_raw_record(b"a", 0, 1, [(10, 0)], 10, n_cigar=60000),
                             _raw_record(b"b", 0, 1, [(10, 0)], 10, l_seq=1 << 20),
                             _raw_record(b"c", 0, 1, [(10, 0)], 10, l_name=200),
                             _raw_record(b"dddd", 0, 1, [(10, 0)], 10, l_name=3))):
        bam = str(tmp_path / f"bad{k}.bam")
        _raw_bam(bam, [good, bad])
        h = C.c_void_p()
        assert L.spx_bam_open(bam.encode(), 1, C.byref(h)) == 0
        bp = C.POINTER(records.SpxBatch)()
        assert L.spx_bam_next_batch(h, 8, C.byref(bp)) == api.EINVAL, k
        assert b"corrupt" in L.spx_io_last_error()
        L.spx_bam_close(h)


# ---------------------------------------------------------------- round 3: chunked reader, shards, index
class BamOptions(C.Structure):
    _fields_ = [("threads", C.c_int32), ("batch_groups", C.c_int32), ("ahead_batches", C.c_int32), ("flags", C.c_int32),
                ("chunk_bytes", C.c_int64), ("max_bytes", C.c_int64), ("start_voffset", C.c_int64), ("end_voffset", C.c_int64),
                ("keep_batches", C.c_int32), ("reserved", C.c_int32)]


def _declare_opts(L):
    _declare(L)
    vp = C.c_void_p
    L.spx_bam_default_options.argtypes = [C.POINTER(BamOptions)]
    L.spx_bam_default_options.restype = None
    L.spx_bam_open_opts.argtypes = [C.c_char_p, C.POINTER(BamOptions), C.POINTER(vp)]
    L.spx_bam_release_batch.argtypes = [vp, C.POINTER(records.SpxBatch)]
    L.spx_bam_index_build.argtypes = [C.c_char_p, C.c_int, C.c_int32, C.POINTER(C.c_int64), C.c_int64]
    L.spx_bam_index_build.restype = C.c_int64
    L.spx_bam_index_save.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.c_int64]
    L.spx_bam_index_load.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.c_int64]
    L.spx_bam_index_load.restype = C.c_int64


def _read_all(L, bam, batch, release=False, **kw):
    o = BamOptions()
    L.spx_bam_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    rd = C.c_void_p()
    assert L.spx_bam_open_opts(bam.encode(), C.byref(o), C.byref(rd)) == 0, L.spx_io_last_error()
    got, sizes = [], []
    while True:
        bp = C.POINTER(records.SpxBatch)()
        n = L.spx_bam_next_batch(rd, batch, C.byref(bp))
        assert n >= 0, L.spx_io_last_error()
        if n == 0:
            break
        assert n <= batch
        sizes.append(n)
        got += _records(bp)
        if release:
            assert L.spx_bam_release_batch(rd, bp) == 0
    L.spx_bam_close(rd)
    return got, sizes


def test_records_straddling_inflate_chunks(built, tmp_path):
    """64 KB inflate chunks against 4.5 KB records: hundreds of records start in one chunk and end in the next (their
    front part is copied into the next slot's head room); both writers' block policies; batches released early, so
    slots are recycled while later batches still point into their neighbours"""
    L = api.lib()
    _declare_opts(L)
    g = small_genome(synth.HIFI, read_len=3000, max_secondaries=3, n_paralogs=2, hardclip_frac=0.2, softclip_frac=0.3)
    chunks = [g.reads(i * 100, 100) for i in range(6)]
    want = []
    for ch in chunks:
        want += _records(ch.batch)
    a, b = str(tmp_path / "c.bam"), str(tmp_path / "py.bam")
    synth.write_bam(a, [c.batch for c in chunks], g.ref, threads=3)
    whole = g.reads(0, 600)
    write_bam(b, whole.batch, g.ref)
    for path in (a, b):
        for batch, rel in ((7, False), (64, True), (1000, False)):
            got, sizes = _read_all(L, path, batch, release=rel, threads=3, chunk_bytes=65536, ahead_batches=3)
            assert got == want, (path, batch)
            assert all(s == batch for s in sizes[:-1])
    # the same through a reader that starts cutting batches at open (the command line does: start-up overlap)
    got, _ = _read_all(L, a, 50, threads=2, chunk_bytes=65536, batch_groups=50)
    assert got == want


def test_index_round_trip_and_shards(built, tmp_path):
    """group-start index in the reference's format (int64 count + int64 BGZF virtual offsets, src/secphase_index.c:76-119);
    readers opened on [a[i], a[j]) return exactly the groups [i*step, j*step), whatever the chunk size"""
    L = api.lib()
    _declare_opts(L)
    g = small_genome(synth.HIFI, read_len=2000, max_secondaries=2, n_paralogs=2)
    chunks = [g.reads(i * 64, 64) for i in range(5)]
    want = []
    for ch in chunks:
        want += _records(ch.batch)
    whole = g.reads(0, 320)  # (kept alive: the batch pointer does not own the records)
    for writer in ("c", "py"):
        bam = str(tmp_path / f"{writer}.bam")
        if writer == "c":
            synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=2)
        else:
            write_bam(bam, whole.batch, g.ref)
        step = 10
        n = L.spx_bam_index_build(bam.encode(), 3, step, None, 0)
        assert n == 320 // step + 1
        off = (C.c_int64 * n)()
        assert L.spx_bam_index_build(bam.encode(), 3, step, off, n) == n
        idx = bam + ".secphase.index"
        assert L.spx_bam_index_save(idx.encode(), off, n) == 0
        raw = np.fromfile(idx, dtype="<i8")
        assert raw[0] == n and list(raw[1:]) == list(off)
        back = (C.c_int64 * n)()
        assert L.spx_bam_index_load(idx.encode(), back, n) == n and list(back) == list(off)
        assert all(off[i] < off[i + 1] for i in range(n - 1))
        for chunk_bytes in (65536, 1 << 20):
            for i, j in ((0, n - 1), (0, 3), (3, 17), (17, n - 1), (5, 6), (n - 2, n - 1)):
                got, _ = _read_all(L, bam, 23, threads=2, chunk_bytes=chunk_bytes, start_voffset=off[i], end_voffset=off[j])
                assert got == want[i * step:j * step], (writer, chunk_bytes, i, j)
        # an empty shard
        got, _ = _read_all(L, bam, 23, threads=2, start_voffset=off[4], end_voffset=off[4])
        assert got == []


def test_damaged_files_end_with_an_error(built, tmp_path):
    """a flipped payload byte (CRC32 of the block), a file cut inside a block, a file cut inside a record"""
    L = api.lib()
    _declare_opts(L)
    g = small_genome(synth.HIFI, read_len=2000, max_secondaries=2, n_paralogs=2)
    r = g.reads(0, 200)
    bam = str(tmp_path / "ok.bam")
    synth.write_bam(bam, [r.batch], g.ref, threads=2)
    data = bytearray(open(bam, "rb").read())

    def outcome(blob, **kw):
        p = str(tmp_path / "damaged.bam")
        open(p, "wb").write(bytes(blob))
        o = BamOptions()
        L.spx_bam_default_options(C.byref(o))
        o.threads = 2
        o.chunk_bytes = 1 << 18
        for k, v in kw.items():
            setattr(o, k, v)
        rd = C.c_void_p()
        if L.spx_bam_open_opts(p.encode(), C.byref(o), C.byref(rd)) != 0:
            return "open", L.spx_io_last_error()
        total = 0
        while True:
            bp = C.POINTER(records.SpxBatch)()
            n = L.spx_bam_next_batch(rd, 64, C.byref(bp))
            if n < 0:
                err = L.spx_io_last_error()
                # the error is sticky
                assert L.spx_bam_next_batch(rd, 64, C.byref(bp)) == n
                L.spx_bam_close(rd)
                return "error", err
            if n == 0:
                L.spx_bam_close(rd)
                return "eof", total
            total += n

    assert outcome(data) == ("eof", 200)
    flipped = bytearray(data)
    flipped[len(data) // 2] ^= 0x55
    kind, err = outcome(flipped)
    assert kind == "error" and (b"CRC" in err or b"inflate" in err or b"BGZF" in err)
    kind, err = outcome(data[: len(data) // 2])
    assert kind == "error" and b"truncated" in err
    # cut on a block boundary but inside a record: write the first half of the record stream as its own valid BGZF file
    import gzip
    payload = gzip.open(bam).read()
    p2 = str(tmp_path / "cutrec.bam")
    write_bgzf(p2, payload[: len(payload) // 2 + 3])
    kind, err = outcome(open(p2, "rb").read())
    assert kind == "error" and b"truncated BAM record" in err
    # without the EOF marker block the file still reads (htslib only warns)
    assert outcome(data[:-28]) == ("eof", 200)


def test_c_writer_equals_python_writer_records(built, tmp_path):
    """the bench-side C writer (synth/spx_bamwrite.c) and tests/bamio.py produce the same record stream"""
    import gzip
    g = small_genome(synth.HIFI, read_len=1500, max_secondaries=3, n_paralogs=2, tag_mode=2)
    parts = [g.reads(i * 40, 40) for i in range(3)]
    whole = g.reads(0, 120)
    order = list(range(g.ref.contents.n_contigs))[::-1]
    a, b = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    synth.write_bam(a, [p.batch for p in parts], g.ref, contig_order=order, threads=2)
    write_bam(b, whole.batch, g.ref, contig_order=order)
    assert gzip.open(a).read() == gzip.open(b).read()
    fa, fb = str(tmp_path / "a.fa"), str(tmp_path / "b.fa")
    synth.write_fasta(fa, g.ref)
    write_fasta(fb, g.ref)
    assert open(fa, "rb").read() == open(fb, "rb").read()


def test_readers_closed_early_and_side_by_side(built, tmp_path):
    """a reader closed while it is still inflating and cutting batches ahead (at open, in the middle, after one batch), and
    four readers on one file at once from four threads: no hang, no crash, the same records"""
    import threading
    L = api.lib()
    _declare_opts(L)
    g = small_genome(synth.HIFI, read_len=2500, max_secondaries=2, n_paralogs=2)
    chunks = [g.reads(i * 100, 100) for i in range(4)]
    want = []
    for ch in chunks:
        want += _records(ch.batch)
    bam = str(tmp_path / "r.bam")
    synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=2)
    for take in (0, 1, 3):
        for pre in (0, 25):
            o = BamOptions()
            L.spx_bam_default_options(C.byref(o))
            o.threads, o.chunk_bytes, o.batch_groups, o.ahead_batches = 3, 65536, pre, 4
            rd = C.c_void_p()
            assert L.spx_bam_open_opts(bam.encode(), C.byref(o), C.byref(rd)) == 0
            for _ in range(take):
                bp = C.POINTER(records.SpxBatch)()
                assert L.spx_bam_next_batch(rd, 25, C.byref(bp)) == 25
            L.spx_bam_close(rd)
    got = [None] * 4

    def run(k):
        got[k], _ = _read_all(L, bam, 31 + k, threads=2, chunk_bytes=65536 * (k + 1))

    ths = [threading.Thread(target=run, args=(k,)) for k in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert all(x == want for x in got)


def test_pages_dropped_before_close(built, tmp_path):
    """spx_bam_drop_pages (what the command line calls before it exits without closing): the reader's pages go back to the
    kernel on its pool; the reader can still be closed afterwards, also in the middle of a file with batches handed out"""
    L = api.lib()
    _declare_opts(L)
    L.spx_bam_drop_pages.argtypes = [C.c_void_p]
    L.spx_bam_drop_pages.restype = None
    g = small_genome(synth.HIFI, read_len=2500, max_secondaries=2, n_paralogs=2)
    chunks = [g.reads(i * 100, 100) for i in range(3)]
    bam = str(tmp_path / "d.bam")
    synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=2)
    for take in (1, 12):  # in the middle / at the end of the file
        o = BamOptions()
        L.spx_bam_default_options(C.byref(o))
        o.threads, o.chunk_bytes, o.batch_groups, o.ahead_batches = 3, 65536, 25, 2
        rd = C.c_void_p()
        assert L.spx_bam_open_opts(bam.encode(), C.byref(o), C.byref(rd)) == 0
        n = 0
        for _ in range(take):
            bp = C.POINTER(records.SpxBatch)()
            n += L.spx_bam_next_batch(rd, 25, C.byref(bp))
        assert n == 25 * take
        L.spx_bam_drop_pages(rd)
        L.spx_bam_drop_pages(None)
        L.spx_bam_close(rd)


def test_device_input_handle_without_a_device(built, tmp_path):
    """spx_dbam_open only maps the file and parses the header (no HIP call): targets, reference binding and close work on a
    machine without a GPU; a file that is not a BAM is refused; the header-only reader refuses to cut batches"""
    L = api.lib()
    _declare_opts(L)
    g = small_genome(synth.HIFI, read_len=1500)
    r = g.reads(0, 20)
    bam = str(tmp_path / "d.bam")
    synth.write_bam(bam, [r.batch], g.ref, threads=1)
    h = C.c_void_p()
    o = api.DbamOptions()
    L.spx_dbam_default_options(C.byref(o))
    assert o.max_groups == 95000 and o.host_inflate_percent == -1 and o.start_voffset == -1
    assert L.spx_dbam_open(bam.encode(), C.byref(o), C.byref(h)) == 0
    hdr = L.spx_dbam_header(h)
    assert L.spx_bam_n_targets(hdr) == g.ref.contents.n_contigs
    assert L.spx_bam_bind_reference(hdr, g.ref) == 0
    bp = C.POINTER(records.SpxBatch)()
    assert L.spx_bam_next_batch(hdr, 8, C.byref(bp)) == api.EINVAL
    L.spx_dbam_close(h)
    junk = str(tmp_path / "junk.bam")
    open(junk, "wb").write(b"not a bam at all" * 10)
    assert L.spx_dbam_open(junk.encode(), None, C.byref(h)) != 0
    # SPX_BAM_HEADER_ONLY through the reader's own options
    ob = BamOptions()
    L.spx_bam_default_options(C.byref(ob))
    ob.flags = 4
    rd = C.c_void_p()
    assert L.spx_bam_open_opts(bam.encode(), C.byref(ob), C.byref(rd)) == 0
    assert L.spx_bam_n_targets(rd) == g.ref.contents.n_contigs
    L.spx_bam_close(rd)
