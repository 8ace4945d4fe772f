"""Test-side BAM / FASTA writers (pure Python + zlib): turn a flat record batch into the files the
command-line program reads.  BGZF blocks of <= 60 KB, standard EOF marker."""
import ctypes as C
import struct
import zlib

import numpy as np

_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(data):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def write_bgzf(path, payload, block=60000):
    with open(path, "wb") as f:
        for o in range(0, len(payload), block):
            f.write(_bgzf_block(payload[o:o + block]))
        f.write(_EOF)


def _cstr(base, off):
    return C.string_at(base + off)


def contigs_of(ref):
    r = ref.contents
    out = []
    for i in range(r.n_contigs):
        name = _cstr(r.names, r.name_off[i]).decode()
        seq = C.string_at(r.bases + r.seq_off[i], r.seq_off[i + 1] - r.seq_off[i]).decode()
        out.append((name, seq))
    return out


def write_fasta(path, ref, width=80):
    with open(path, "w") as f:
        for name, seq in contigs_of(ref):
            f.write(f">{name} synthetic\n")
            for o in range(0, len(seq), width):
                f.write(seq[o:o + width] + "\n")


def write_bam(path, batch, ref, contig_order=None, extra_tags=True, de_of=None, mapq_of=None):
    """contig_order: permutation of contig indices for the BAM header (default identity); de_of(a) / mapq_of(a): per-alignment `de:f`
    tag (None: no tag) and MAPQ (default 60) -- what the correct_bam tests filter on"""
    b = batch.contents
    r = ref.contents
    names = [(_cstr(r.names, r.name_off[i]), r.seq_off[i + 1] - r.seq_off[i]) for i in range(r.n_contigs)]
    order = list(contig_order) if contig_order is not None else list(range(r.n_contigs))
    tid_of = {c: i for i, c in enumerate(order)}
    text = b"@HD\tVN:1.6\tSO:queryname\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (names[c][0], names[c][1]) for c in order)
    out = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(order)))
    for c in order:
        nm = names[c][0] + b"\0"
        out += struct.pack("<i", len(nm)) + nm + struct.pack("<i", names[c][1])
    for g in range(b.n_groups):
        qn = _cstr(b.qnames, b.qname_off[g]) + b"\0"
        for a in range(b.grp_first[g], b.grp_first[g + 1]):
            lq, nc = b.l_qseq[a], b.n_cigar[a]
            cig = bytes(np.ctypeslib.as_array(b.cigar, shape=(b.cigar_off[a] + nc,))[b.cigar_off[a]:].astype("<u4").tobytes())
            seq = C.string_at(C.addressof(b.seq4.contents) + b.seq_off[a], (lq + 1) // 2)
            qual = C.string_at(C.addressof(b.qual.contents) + b.qual_off[a], lq)
            aux = b""
            if extra_tags:
                aux += b"NMi" + struct.pack("<i", 3) + b"tpAP"
            if b.cs_off[a] >= 0:
                aux += b"csZ" + _cstr(b.cs, b.cs_off[a]) + b"\0"
            if b.md_off and b.md_off[a] >= 0:
                aux += b"MDZ" + _cstr(b.md, b.md_off[a]) + b"\0"
            if extra_tags:
                aux += b"zzBs" + struct.pack("<ihh", 2, -1, 7)
            if de_of is not None and de_of(a) is not None:
                aux += b"def" + struct.pack("<f", de_of(a))
            core = struct.pack("<iiBBHHHiiii", tid_of[b.tid[a]] if b.tid[a] >= 0 else -1, b.pos[a], len(qn), 60 if mapq_of is None else mapq_of(a), 4680, nc,
                               b.flag[a], lq, -1, -1, 0)
            rec = core + qn + cig + seq + qual + aux
            out += struct.pack("<i", len(rec)) + rec
    write_bgzf(path, bytes(out))


def _fmt_g(x):
    return "%g" % x


def sam_text(batch, ref, qual=None, contig_order=None, extra_tags=True, groups=None):
    """The SAM text (header + records) of the BAM write_bam() produces, written independently of the product's
    formatter from the SAM specification: 11 mandatory fields, then the aux fields in file order.
    qual: optional replacement for the batch's qual[] (same layout); groups: indices to include (default all);
    unmapped records are skipped like the reference's reader does (src/secphase.c:340)."""
    b = batch.contents
    r = ref.contents
    names = [(_cstr(r.names, r.name_off[i]).decode(), r.seq_off[i + 1] - r.seq_off[i]) for i in range(r.n_contigs)]
    order = list(contig_order) if contig_order is not None else list(range(r.n_contigs))
    out = ["@HD\tVN:1.6\tSO:queryname\n"] + ["@SQ\tSN:%s\tLN:%d\n" % names[c] for c in order]
    for g in (range(b.n_groups) if groups is None else groups):
        qn = _cstr(b.qnames, b.qname_off[g]).decode()
        for a in range(b.grp_first[g], b.grp_first[g + 1]):
            if b.flag[a] & 4:
                continue
            lq, nc = b.l_qseq[a], b.n_cigar[a]
            cig = "".join("%d%s" % (b.cigar[b.cigar_off[a] + k] >> 4, "MIDNSHP=XB"[b.cigar[b.cigar_off[a] + k] & 15])
                          for k in range(nc)) or "*"
            sq = C.string_at(C.addressof(b.seq4.contents) + b.seq_off[a], (lq + 1) // 2)
            seq = "".join("=ACMGRSVTWYHKDBN"[(sq[k >> 1] >> (4 if k % 2 == 0 else 0)) & 15] for k in range(lq)) or "*"
            if qual is None:
                ql = C.string_at(C.addressof(b.qual.contents) + b.qual_off[a], lq)
            else:
                ql = bytes(qual[b.qual_off[a]:b.qual_off[a] + lq])
            qs = "".join(chr(x + 33) for x in ql) if lq else "*"
            f = [qn, str(b.flag[a]), names[b.tid[a]][0], str(b.pos[a] + 1), "60", cig, "*", "0", "0", seq, qs]
            if extra_tags:
                f += ["NM:i:3", "tp:A:P"]
            if b.cs_off[a] >= 0:
                f.append("cs:Z:" + _cstr(b.cs, b.cs_off[a]).decode())
            if b.md_off and b.md_off[a] >= 0:
                f.append("MD:Z:" + _cstr(b.md, b.md_off[a]).decode())
            if extra_tags:
                f.append("zz:B:s,-1,7")
            out.append("\t".join(f) + "\n")
    return "".join(out)


def read_bam(path):
    """(header text, [(name, length)], [record]) of a BAM file, read independently of the product (gzip members + struct); a record is a
    dict with the raw bytes (without block_size) and the fields the correct_bam tests look at"""
    import gzip
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", raw, 4)
    text = raw[8:8 + l_text]
    at = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, at)
    at += 4
    refs = []
    for _ in range(n_ref):
        ln, = struct.unpack_from("<i", raw, at)
        nm = raw[at + 4:at + 4 + ln - 1].decode()
        tl, = struct.unpack_from("<i", raw, at + 4 + ln)
        refs.append((nm, tl))
        at += 8 + ln
    recs = []
    while at < len(raw):
        bs, = struct.unpack_from("<i", raw, at)
        r = raw[at + 4:at + 4 + bs]
        at += 4 + bs
        tid, pos, l_name, mapq, _bin, ncig, flag, lseq = struct.unpack_from("<iiBBHHHi", r, 0)
        name = r[32:32 + l_name - 1].decode()
        cig = struct.unpack_from("<%dI" % ncig, r, 32 + l_name)
        aux_at = 32 + l_name + 4 * ncig + (lseq + 1) // 2 + lseq
        recs.append(dict(raw=r, name=name, tid=tid, pos=pos, mapq=mapq, flag=flag, cigar=[(c & 15, c >> 4) for c in cig], aux=r[aux_at:], aux_at=aux_at))
    return text, refs, recs
