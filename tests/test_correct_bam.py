"""the consumer of the relabel list.  spx_relabel_table_* parses `<prefix>.out.log` the way
/root/reference/programs/src/correct_bam.c:32-91 does, spx_correct_bam / bin/correct_bam apply it (flag swap :352-358, filters :359-367).
CPU tests: the list comes from the oracle; the expected output is an independent Python restatement of correct_bam's record loop
over the input BAM read with gzip + struct.  The -m gpu test feeds the list the HIP path wrote (test_gpu_parity.py)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from bamio import read_bam, write_bam, write_fasta
from common import small_genome
from oracle import orc
from secphase_amd import api, records, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "secphase_amd", "bin", "correct_bam")


def parse_list_like_correct_bam(path):
    """get_phased_read_table (correct_bam.c:32-91), restated"""
    table = {}
    name = new = old = None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith("$"):
            name, new, old = line.split("\t")[1], None, None
        elif line.startswith("@"):
            f = line.split("\t")
            new = (f[2], int(f[3]))
            if old is not None and old != new:
                table[name] = new
        elif line.startswith("*"):
            f = line.split("\t")
            old = (f[2], int(f[3]))
            if new is not None and old != new:
                table[name] = new
    return table


def load_table(path):
    L = api.lib()
    L.spx_relabel_table_size.restype = C.c_int64
    L.spx_relabel_table_get.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int32)]
    L.spx_relabel_table_size.argtypes = [C.c_void_p]
    L.spx_relabel_table_free.argtypes = [C.c_void_p]
    L.spx_relabel_table_free.restype = None
    h = C.c_void_p()
    api._chk(L.spx_relabel_table_load(path.encode() if path else None, C.byref(h)), "spx_relabel_table_load")
    out = {}
    for i in range(L.spx_relabel_table_size(h)):
        q, c, s = C.c_char_p(), C.c_char_p(), C.c_int32()
        assert L.spx_relabel_table_get(h, i, C.byref(q), C.byref(c), C.byref(s)) == 0
        out[q.value.decode()] = (c.value.decode(), s.value)
    L.spx_relabel_table_free(h)
    return out


def expected_records(in_bam, table, primary_only=False, no_tag=False, min_read=5000, min_aln=5000, max_mapq=100, max_div=0.12, mapq_table=None,
                     exclude=()):
    """correct_bam.c:347-376, restated over the records of in_bam"""
    _, refs, recs = read_bam(in_bam)
    out = []
    for r in recs:
        if r["flag"] & 4 or r["name"] in exclude:
            continue
        contig = refs[r["tid"]][0]
        if r["name"] in table:
            prim = table[r["name"]] == (contig, r["pos"])
        else:
            prim = not (r["flag"] & 256)
        flag = r["flag"]
        if prim:
            flag &= ~256
        else:
            if primary_only:
                continue
            flag |= 256
        read_len = sum(n for op, n in r["cigar"] if op in (0, 7, 8, 1, 4, 5))
        aln_len = sum(n for op, n in r["cigar"] if op in (0, 7, 8))
        if read_len < min_read or aln_len < min_aln:
            continue
        mapq = r["mapq"]
        for (c_, s_, m_) in (mapq_table or {}).get(r["name"], []):
            if c_ == contig and s_ == r["pos"]:
                mapq = m_ & 255
                break
        if max_mapq < mapq:
            continue
        de = 0.0
        k = r["aux"].find(b"def")
        if k >= 0:
            de, = struct.unpack_from("<f", r["aux"], k + 3)
        if max_div < de:
            continue
        raw = bytearray(r["raw"][:r["aux_at"]] if no_tag else r["raw"])
        raw[9] = mapq
        raw[14:16] = struct.pack("<H", flag)
        out.append(bytes(raw))
    return out


@pytest.fixture(scope="module")
def fixture(built, tmp_path_factory):
    """tie groups, clipped records, a `de` tag per record, an unmapped record's worth of variety in MAPQ"""
    d = tmp_path_factory.mktemp("correct")
    g = small_genome(synth.HIFI, max_secondaries=4, n_paralogs=3, read_len=6000, min_secondaries=0, paralog_snv_rate=0.0002, hardclip_frac=0.2,
                     softclip_frac=0.3)
    p = records.preset("hifi")
    p.prim_margin_score = 5.0
    r = g.reads(0, 120)
    bam, log = str(d / "in.bam"), str(d / "o.out.log")
    rng = np.random.default_rng(3)
    de = rng.choice([0.001, 0.01, 0.05, 0.2], size=r.batch.contents.n_alns, p=[0.4, 0.3, 0.2, 0.1])
    mq = rng.choice([0, 1, 30, 60], size=r.batch.contents.n_alns)
    write_bam(bam, r.batch, g.ref, de_of=lambda a: None if a % 17 == 0 else float(de[a]), mapq_of=lambda a: int(mq[a]))
    nre, res = orc.run_batch(r.batch, g.ref, p, threads=2, seed=1, log_path=log)
    ties = 0
    for e in res:
        if e.n_aln >= 2:
            sec = [a for a in range(e.n_aln) if a != e.prim_idx]
            mxs = max(e.score[a] for a in sec)
            ties += sum(1 for a in sec if e.score[a] >= mxs) > 1
    assert nre > 5 and ties > 0
    return dict(dir=d, genome=g, reads=r, bam=bam, log=log, res=res, nre=nre)


def test_table_is_what_correct_bam_reads(fixture):
    t = load_table(fixture["log"])
    assert t == parse_list_like_correct_bam(fixture["log"])
    b, rf = fixture["reads"].batch.contents, fixture["genome"].ref.contents
    expect = {}
    for i, e in enumerate(fixture["res"]):
        if e.relabel:
            a = b.grp_first[i] + e.best_idx
            expect[C.string_at(b.qnames + b.qname_off[i]).decode()] = (C.string_at(rf.names + rf.name_off[b.tid[a]]).decode(), b.pos[a])
    assert t == expect and len(t) == fixture["nre"]
    assert load_table(None) == {}


def test_table_ignores_records_whose_locations_coincide(built, tmp_path):
    """correct_bam.c:64-68,77-82; either order of `*` and `@`; a repeated read name keeps the later record"""
    p = tmp_path / "x.out.log"
    p.write_text("#MARKER SCORE\n$\tr1\n*\t-3.00\tctgA\t100\t900\n@\t-1.00\tctgB\t50\t800\n!\t-9.00\tctgC\t7\t70\n\n"
                 "#MARKER SCORE\n$\tr2\n@\t-1.00\tctgA\t100\t800\n*\t-3.00\tctgA\t100\t900\n\n"
                 "#MARKER SCORE\n$\tr3\n@\t-1.00\tctgB\t0\t800\n*\t-3.00\tctgA\t0\t900\n\n"
                 "#MARKER SCORE\n$\tr1\n*\t-3.00\tctgA\t100\t900\n@\t-1.00\tctgD\t5\t800\n\n")
    t = load_table(str(p))
    assert t == {"r1": ("ctgD", 5), "r3": ("ctgB", 0)} == parse_list_like_correct_bam(str(p))


@pytest.mark.parametrize("opts,kw", [
    ([], {}),
    (["-m", "1000", "-a", "500"], dict(min_read=1000, min_aln=500)),
    (["-m", "1000", "-a", "500", "-p"], dict(min_read=1000, min_aln=500, primary_only=True)),
    (["-m", "1000", "-a", "500", "-t", "-x", "30", "-d", "0.03"], dict(min_read=1000, min_aln=500, no_tag=True, max_mapq=30, max_div=0.03)),
], ids=["defaults", "short-reads-kept", "primary-only", "noTag-maxMapq-maxDiv"])
def test_correct_bam_equals_the_restated_record_loop(fixture, tmp_path, opts, kw):
    out = str(tmp_path / "out.bam")
    p = subprocess.run([EXE, "-i", fixture["bam"], "-o", out, "-P", fixture["log"], "-n", "3"] + opts, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    table = parse_list_like_correct_bam(fixture["log"])
    want = expected_records(fixture["bam"], table, **kw)
    text_in, refs_in, _ = read_bam(fixture["bam"])
    text, refs, got = read_bam(out)
    assert text == text_in and refs == refs_in
    assert [r["raw"] for r in got] == want
    if not opts:
        assert len(want) > 100   # reads of 6 kb pass the 5 kb defaults
    if "-p" in opts:
        # exactly one primary per read that kept one, and for a relabelled read it is the record the list names
        names = {}
        for r in got:
            assert not (r["flag"] & 256)
            names.setdefault(r["name"], []).append((refs[r["tid"]][0], r["pos"]))
        hits = [n for n in table if n in names]
        assert hits and all(names[n] == [table[n]] for n in hits)


def test_mapq_table_exclude_list_and_sam_text(fixture, tmp_path):
    _, refs, recs = read_bam(fixture["bam"])
    some = [r for r in recs if not r["flag"] & 4][::7]
    mt = tmp_path / "mapq.tsv"
    mt.write_text("".join(f"{r['name']}\t{refs[r['tid']][0]}\t{r['pos'] + 1}\t{(k * 37) % 300}\n" for k, r in enumerate(some)))
    ex = tmp_path / "exclude.txt"
    ex_names = sorted({r["name"] for r in recs})[:5]
    ex.write_text("".join(n + "\n" for n in ex_names))
    mapq_table = {}
    for k, r in enumerate(some):
        mapq_table.setdefault(r["name"], []).append((refs[r["tid"]][0], r["pos"], (k * 37) % 300))
    out = str(tmp_path / "out.bam")
    p = subprocess.run([EXE, "-i", fixture["bam"], "-o", out, "-P", fixture["log"], "-M", str(mt), "-e", str(ex), "-m", "1000", "-a", "500", "-x", "255"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    table = parse_list_like_correct_bam(fixture["log"])
    want = expected_records(fixture["bam"], table, min_read=1000, min_aln=500, max_mapq=255, mapq_table=mapq_table, exclude=set(ex_names))
    _, _, got = read_bam(out)
    assert [r["raw"] for r in got] == want
    assert not any(r["name"] in ex_names for r in got)
    # the same records as SAM text through the library's formatter: flags and MAPQs of the text equal the BAM's
    sam = str(tmp_path / "out.sam")
    p = subprocess.run([EXE, "-i", fixture["bam"], "-o", sam, "-P", fixture["log"], "-M", str(mt), "-e", str(ex), "-m", "1000", "-a", "500", "-x", "255", "--samText"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = [ln.split("\t") for ln in open(sam) if not ln.startswith("@")]
    assert [(f[0], int(f[1]), f[2], int(f[3]) - 1, int(f[4])) for f in lines] == [(r["name"], r["flag"], refs[r["tid"]][0], r["pos"], r["mapq"]) for r in got]


def test_correct_bam_reports_errors(built, tmp_path):
    p = subprocess.run([EXE, "-i", str(tmp_path / "missing.bam"), "-o", str(tmp_path / "o.bam")], capture_output=True, text=True)
    assert p.returncode != 0 and "missing.bam" in p.stderr
    p = subprocess.run([EXE, "-o", str(tmp_path / "o.bam")], capture_output=True, text=True)
    assert p.returncode != 0
