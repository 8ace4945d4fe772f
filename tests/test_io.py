"""CPU: the zlib-only BAM / FASTA readers of libspx hand back exactly the records that were written."""
import ctypes as C

import numpy as np

from bamio import contigs_of, write_bam, write_fasta
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
