"""-m gpu: the device-resident BAM input (spx_dbam_*: BGZF inflate, record chain, fields / tags, name groups, dispatch filter
and the gather into the staged layout as kernels) against the host reader, the host plan and the oracle."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from common import small_genome
from oracle import orc
from secphase_amd import api, records, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(built):
    c = api.Context(0)
    yield c
    c.close()


def _key(o):
    return (o.n_aln, tuple(o.score[a] for a in range(max(o.n_aln, 0))), tuple(o.rfe[a] for a in range(max(o.n_aln, 0))), o.prim_idx, o.max_idx,
            o.tie_mask, o.pass_, o.n_problems, o.n_markers, o.dp_cells)


def _names(bp):
    b = bp.contents
    out = []
    for g in range(b.n_groups):
        nm = C.string_at(b.qnames + b.qname_off[g])
        out.append((nm, tuple((b.flag[a], b.tid[a], b.pos[a]) for a in range(b.grp_first[g], b.grp_first[g + 1]))))
    return out


def _device_run(bam, ctx, par, ref, env=None, compare_plans=None, **opts):
    """every work list of the device input: results per group, names per group, list sizes; compare_plans(work, first group, n)
    may look at the staged list before it is freed"""
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        d = api.DeviceBam(bam, [ctx], par, ref, **opts)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    res, names, sizes = [], [], []
    try:
        while True:
            nx = d.next()
            if nx is None:
                break
            w, lane, nb, n = nx
            assert lane == 0 and nb.contents.n_groups == n
            w.prepare_staged()
            if compare_plans:
                compare_plans(w, len(res), n)
            w.launch()
            out = w.collect(finalize_seed=None)
            res += [_key(out[k]) for k in range(n)]
            names += _names(nb)
            sizes.append(n)
            w.free()
            d.release(nb)
    finally:
        d.close()
    return res, names, sizes


def _expected(ctx, whole, par):
    out, _ = ctx.score_batch(whole.batch, par, finalize_seed=None)
    n = whole.batch.contents.n_groups
    return [_key(out[k]) for k in range(n)], _names(whole.batch)


CASES = {
    "hifi": (dict(platform=synth.HIFI, max_secondaries=3, n_paralogs=3, read_len=6000, hardclip_frac=0.2, softclip_frac=0.3), 300, "hifi"),
    "edge": (dict(platform=synth.HIFI, hardclip_frac=0.5, softclip_frac=0.5, shuffle_records=1, inverted_paralogs=1, n_paralogs=3, max_secondaries=4,
                  n_base_frac=0.002, read_len=5000, min_secondaries=0), 240, "hifi"),
    "md": (dict(platform=synth.HIFI, tag_mode=1, read_len=5000, max_secondaries=3, n_paralogs=2, hardclip_frac=0.3, softclip_frac=0.3), 160, "hifi"),
    "long": (dict(platform=synth.MIXED, n_paralogs=7, contig_len=250000, max_read_len=90000), 120, "hifi"),  # records longer than a BGZF block
    "ont": (dict(platform=synth.ONT, n_paralogs=3, contig_len=300000), 64, "ont"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_input_equals_host_path(ctx, tmp_path, name):
    """the same file through the device input -- as ONE segment, as many small segments (carry across segments, records and
    groups straddling them) and with work lists split by a group cap -- gives, group for group, the results, names, flags,
    targets and positions of the host path on the generator's own records; and the work list the device builds from the
    device-staged records equals the host plan array for array"""
    kw, n, preset = CASES[name]
    kw = dict(kw)
    plat = kw.pop("platform")
    g = small_genome(plat, **kw)
    par = records.preset("ont", bandwidth=50) if preset == "ont" else records.preset("hifi")
    chunk = 40
    chunks = [g.reads(k, min(chunk, n - k)) for k in range(0, n, chunk)]
    whole = g.reads(0, n)
    bam = str(tmp_path / f"{name}.bam")
    synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=2)
    ctx.set_reference(g.ref)
    want, want_names = _expected(ctx, whole, par)
    assert sum(1 for x in want if x[0] >= 2) > n // 2

    from test_gpu_parity import _plans_equal

    def cmp_plans(w, first, cnt):
        sub = g.reads(first, cnt)
        host = api.Plan(g.ref, sub.batch, par)
        dev = w.export_plan()
        _plans_equal(dev, host, cnt)
        dev.close()
        host.close()

    res, names, sizes = _device_run(bam, ctx, par, g.ref, compare_plans=cmp_plans)
    assert sizes == [n] and res == want and names == want_names
    # (by default a share of every segment is inflated by the host pool and uploaded raw; 0 / 100: the inflate kernel / the host alone)
    res, names, sizes = _device_run(bam, ctx, par, g.ref, env={"SPX_DIN_SEG_KB": "192"}, compare_plans=cmp_plans if name in ("hifi", "long") else None,
                                    host_inflate_percent=0)
    assert len(sizes) > 3 and sum(sizes) == n and res == want and names == want_names
    res, names, sizes = _device_run(bam, ctx, par, g.ref, env={"SPX_DIN_SEG_KB": "1024"}, max_groups=17, host_inflate_percent=100)
    assert max(sizes) <= 17 and sum(sizes) == n and res == want and names == want_names
    res, names, sizes = _device_run(bam, ctx, par, g.ref, env={"SPX_DIN_SEG_KB": "2048"}, host_inflate_percent=50)
    assert sum(sizes) == n and res == want and names == want_names


def _raw_bam(path, records_bytes, contigs, level=6, block=0xff00):
    """a BAM from raw record bytes with htslib's block policy: a record that does not fit the current block starts a new one"""
    import zlib
    text = b"@HD\tVN:1.6\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % c for c in contigs)
    hdr = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(contigs)))
    for nm, ln in contigs:
        hdr += struct.pack("<i", len(nm) + 1) + nm + b"\0" + struct.pack("<i", ln)

    def bgzf(data):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))
    out = bytearray()
    cur = bytearray(hdr)
    for rec in records_bytes:
        piece = struct.pack("<i", len(rec)) + rec
        if len(cur) + len(piece) > block and cur:
            out += bgzf(bytes(cur))
            cur = bytearray()
        while len(piece) > block:  # a record longer than a block is written in pieces
            room = block - len(cur)
            cur += piece[:room]
            piece = piece[room:]
            out += bgzf(bytes(cur))
            cur = bytearray()
        cur += piece
    if cur:
        out += bgzf(bytes(cur))
    out += bgzf(b"")
    open(path, "wb").write(bytes(out))


def _rec(name, flag, tid, pos, cigar, seq, qual, aux=b"", n_cigar=None, l_seq=None, l_name=None):
    qn = name + b"\0"
    while len(qn) % 4:  # htslib pads the name so that the CIGAR is word-aligned inside the record
        qn += b"\0"
    nt = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}
    sq = bytearray()
    for i in range(0, len(seq), 2):
        sq.append(nt[seq[i]] << 4 | (nt[seq[i + 1]] if i + 1 < len(seq) else 0))
    cig = b"".join(struct.pack("<I", (ln << 4) | op) for ln, op in cigar)
    core = struct.pack("<iiBBHHHiiii", tid, pos, len(qn) if l_name is None else l_name, 60, 4680, len(cigar) if n_cigar is None else n_cigar, flag,
                       len(seq) if l_seq is None else l_seq, -1, -1, 0)
    return core + qn + cig + bytes(sq) + bytes(qual) + aux


def _host_reader_run(bam, ctx, par, ref):
    """the same file through the host reader (spx_bam_next_batch) and spx_score_batch"""
    L = api.lib()
    h = C.c_void_p()
    assert L.spx_bam_open(bam.encode(), 2, C.byref(h)) == 0, L.spx_io_last_error()
    L.spx_bam_bind_reference(h, ref)
    res, names = [], []
    while True:
        bp = C.POINTER(records.SpxBatch)()
        n = L.spx_bam_next_batch(h, 4096, C.byref(bp))
        assert n >= 0, L.spx_io_last_error()
        if n == 0:
            break
        out, _ = ctx.score_batch(bp, par, finalize_seed=None)
        res += [_key(out[k]) for k in range(n)]
        names += _names(bp)
    L.spx_bam_close(h)
    return res, names


def test_device_input_record_shapes_of_the_format(ctx, tmp_path):
    """hand-made records: a CIGAR that lives in the CG:B,I tag, unmapped records inside a group, a group of more than ten
    records, a supplementary record, a group of one, equal names that are not adjacent, MD beside cs, aux fields of every
    type in front of the tags, a target the FASTA lacks -- the device input and the host reader must agree on every group"""
    from common import HandRef
    rng = np.random.default_rng(4)
    ref_seq = "".join("ACGT"[i] for i in rng.integers(0, 4, 4000))
    ref2 = "".join("ACGT"[i] for i in rng.integers(0, 4, 3000))
    href = HandRef([("c0", ref_seq), ("c1", ref2)])
    contigs = ((b"c0", 4000), (b"ghost", 500), (b"c1", 3000))  # BAM target 1 is not in the FASTA; c1 is target 2 -> contig 1

    def aln(name, flag, tid, pos, length, muts=(), extra_aux=b"", md=False, qual=40, seqsrc=None):
        src = (ref_seq if tid == 0 else ref2)[pos:pos + length]
        s = list(src)
        cs = ""
        last = 0
        for m in muts:
            s[m] = "ACGT"[("ACGT".index(s[m]) + 1) % 4]
            if m > last:
                cs += f":{m - last}"
            cs += f"*{src[m].lower()}{s[m].lower()}"
            last = m + 1
        if length > last:
            cs += f":{length - last}"
        aux = extra_aux + (b"MDZ" + str(length).encode() + b"\0" if md else b"") + b"csZ" + cs.encode() + b"\0"
        return _rec(name, flag, tid, pos, [(length, 0)], "".join(s), [qual] * length, aux=aux)

    every = (b"XAAx" + b"Xcc\x05" + b"XCC\x05" + b"Xss\x05\0" + b"XSS\x05\0" + b"Xii\x05\0\0\0" + b"XII\x05\0\0\0" + b"Xff\0\0\x80?" +
             b"XZZhello\0" + b"XHH1AE3\0" + b"XBBc\x03\0\0\0abc" + b"XDBS\x02\0\0\0\x01\0\x02\0" + b"XEBf\x01\0\0\0\0\0\x80?")
    recs = []
    # group 1: plain pair with markers
    recs += [aln(b"g1", 0, 0, 100, 600, muts=(50, 300)), aln(b"g1", 256, 0, 1200, 600, muts=(50, 420), extra_aux=every)]
    # group 2: CIGAR in the CG tag (placeholder <l_seq>S<ref_len>N in the record)
    real = [(5, 4), (300, 0), (2, 1), (293, 0)]
    seq2 = "ACGTA" + ref_seq[2000:2300] + "GG" + ref_seq[2300:2593]
    cg = b"CGBI" + struct.pack("<i", len(real)) + b"".join(struct.pack("<I", (ln << 4) | op) for ln, op in real)
    recs += [_rec(b"g2", 0, 0, 2000, [(600, 4), (593, 3)], seq2, [35] * 600, aux=b"NMi" + struct.pack("<i", 2) + cg + b"csZ:300+gg:293\0"),
             aln(b"g2", 256, 2, 500, 600, muts=(10,))]
    # group 3: unmapped record in the middle, secondary on the target the FASTA lacks
    recs += [aln(b"g3", 0, 0, 300, 500, muts=(100,)), _rec(b"g3", 4, -1, -1, [], "ACGT", [10] * 4), aln(b"g3", 256, 2, 100, 500, muts=(100, 200))]
    recs += [aln(b"g3b", 0, 0, 300, 500, muts=(100,)), aln(b"g3b", 256, 1, 10, 400)]
    # group 4: twelve records (rejected: more than ten), group 5: a supplementary record, group 6: one record
    recs += [aln(b"g4", 0 if k == 0 else 256, 0, 50 * k, 400, muts=(7,)) for k in range(12)]
    recs += [aln(b"g5", 0, 0, 10, 400, muts=(7,)), aln(b"g5", 2048, 0, 900, 400, muts=(9,)), aln(b"g5", 256, 2, 900, 400)]
    recs += [aln(b"g6", 0, 0, 10, 400)]
    # the name g1 again: a NEW group (groups are runs of equal names, src/secphase.c:273-279)
    recs += [aln(b"g1", 0, 2, 100, 700, muts=(350,), md=True), aln(b"g1", 256 | 16, 0, 2500, 700, muts=(20, 600), md=True)]
    # MD only
    recs += [_rec(b"g7", 0, 0, 700, [(300, 0)], ref_seq[700:1000], [40] * 300, aux=b"MDZ300\0"),
             _rec(b"g7", 256, 2, 700, [(300, 0)], ref2[700:850] + "A" + ref2[851:1000], [40] * 300, aux=b"MDZ150" + ref2[850].encode() + b"149\0")]
    # no tag at all in a dispatched group: the group reports SPX_ENOTAG on both paths
    recs += [_rec(b"g8", 0, 0, 700, [(300, 0)], ref_seq[700:1000], [40] * 300), _rec(b"g8", 256, 2, 700, [(300, 0)], ref2[700:1000], [40] * 300)]
    bam = str(tmp_path / "shapes.bam")
    _raw_bam(bam, recs, contigs)
    par = records.preset("hifi")
    ctx.set_reference(href.ref)
    want, want_names = _host_reader_run(bam, ctx, par, href.ref)
    assert len(want) == 10 and [x[0] >= 2 for x in want] == [True, True, True, False, False, False, False, True, True, False]
    assert want[9][0] == api.ENOTAG and want[3][0] < 0  # no tag; a kept record on a target the FASTA lacks
    for env, opts in (({}, {}), ({"SPX_DIN_SEG_KB": "64"}, {}), ({}, {"max_groups": 3})):
        res, names, sizes = _device_run(bam, ctx, par, href.ref, env=env, **opts)
        assert res == want and names == want_names, (env, opts)


def test_device_input_records_longer_than_blocks_and_segments(ctx, tmp_path):
    """records of 150-400 KB (each spans several BGZF blocks, the next record starts in the MIDDLE of a block), segments of
    64 KB (a record spans several segments: it travels in the carry until it is complete)"""
    from common import HandRef
    rng = np.random.default_rng(6)
    ref_seq = "".join("ACGT"[i] for i in rng.integers(0, 4, 400000))
    href = HandRef([("c0", ref_seq)])
    recs = []
    for k, ln in enumerate((100000, 260000, 3000, 150000, 70000, 500)):
        muts = sorted(set(int(x) for x in rng.integers(10, ln - 10, 6)))
        for j, (flag, pos) in enumerate(((0, 1000 + 17 * k), (256, 50000 + 31 * k))):
            if pos + ln > len(ref_seq):
                pos = len(ref_seq) - ln - 1
            src = ref_seq[pos:pos + ln]
            s = list(src)
            cs, last = "", 0
            for m in (muts if j else muts[:3]):
                s[m] = "ACGT"[("ACGT".index(s[m]) + 1) % 4]
                cs += (f":{m - last}" if m > last else "") + f"*{src[m].lower()}{s[m].lower()}"
                last = m + 1
            cs += f":{ln - last}" if ln > last else ""
            recs.append(_rec(b"read%d" % k, flag, 0, pos, [(ln, 0)], "".join(s), [40] * ln, aux=b"csZ" + cs.encode() + b"\0"))
    bam = str(tmp_path / "longrec.bam")
    _raw_bam(bam, recs, ((b"c0", 400000),))
    par = records.preset("hifi")
    ctx.set_reference(href.ref)
    want, want_names = _host_reader_run(bam, ctx, par, href.ref)
    assert len(want) == 6 and all(x[0] == 2 for x in want)
    for env in ({}, {"SPX_DIN_SEG_KB": "64"}, {"SPX_DIN_SEG_KB": "200"}):
        res, names, sizes = _device_run(bam, ctx, par, href.ref, env=env)
        assert res == want and names == want_names, env
    # a record that does not fit the carry buffer is refused with a message, not walked past
    with pytest.raises(api.SpxError) as ei:
        _device_run(bam, ctx, par, href.ref, env={"SPX_DIN_SEG_KB": "64", "SPX_DIN_CARRY_KB": "128"})
    assert "carry" in str(ei.value) or "corrupt" in str(ei.value)


def test_device_input_refuses_damaged_files(ctx, tmp_path):
    """damage behind valid BGZF framing (record lengths, field lengths, names without NUL), damaged DEFLATE data, a wrong
    CRC, a truncated file: an error from spx_dbam_next, never a fault; the host reader refuses the same files"""
    import zlib
    g = small_genome(synth.HIFI, read_len=3000, max_secondaries=2)
    r = g.reads(0, 60)
    bam = str(tmp_path / "ok.bam")
    synth.write_bam(bam, [r.batch], g.ref, threads=1)
    par = records.preset("hifi")
    ctx.set_reference(g.ref)
    blob = open(bam, "rb").read()

    def blocks(b):
        out, at = [], 0
        while at < len(b):
            bsize = struct.unpack_from("<H", b, at + 16)[0] + 1
            out.append((at, bsize))
            at += bsize
        return out
    bl = blocks(blob)
    payload = [zlib.decompress(blob[a + 18:a + n - 8], -15) for a, n in bl]

    def repack(pl):
        out = bytearray()
        for data in pl:
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = co.compress(data) + co.flush()
            out += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                    struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))
        return bytes(out)
    assert _device_run(str(_w(tmp_path / "re.bam", repack(payload))), ctx, par, g.ref)[2] == [60]
    # where the first record starts: behind the header
    hdr_len = 12 + struct.unpack_from("<i", payload[0], 4)[0]
    n_ref = struct.unpack_from("<i", payload[0], hdr_len - 4)[0]
    at = hdr_len
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", payload[0], at)[0]
        at += 8 + ln
    rec0 = at
    damaged = []
    for what in ("block_size", "l_seq", "n_cigar", "name", "huge"):
        pl = [bytearray(p) for p in payload]
        k, o = (0, rec0) if rec0 < len(pl[0]) else (1, 0)
        if what == "block_size":
            struct.pack_into("<i", pl[k], o, 7)
        elif what == "huge":
            struct.pack_into("<i", pl[k], o, 0x7fffff00)
        elif what == "l_seq":
            struct.pack_into("<i", pl[k], o + 4 + 16, 1 << 28)
        elif what == "n_cigar":
            struct.pack_into("<H", pl[k], o + 4 + 12, 65000)
        else:
            ln = pl[k][o + 4 + 8]
            pl[k][o + 4 + 32 + ln - 1] = 65
        damaged.append((what, repack([bytes(p) for p in pl])))
    mid = bl[len(bl) // 2]
    b2 = bytearray(blob)
    b2[mid[0] + 18 + 40] ^= 0x55
    damaged.append(("deflate", bytes(b2)))
    b3 = bytearray(blob)
    b3[mid[0] + mid[1] - 8] ^= 1
    damaged.append(("crc", bytes(b3)))
    damaged.append(("truncated", blob[:bl[-3][0] + 100]))
    L = api.lib()
    for what, data in damaged:
        path = str(_w(tmp_path / "bad.bam", data))
        for env in ({}, {"SPX_DIN_SEG_KB": "64"}):
            with pytest.raises(api.SpxError):
                _device_run(path, ctx, par, g.ref, env=env)
        h = C.c_void_p()
        if L.spx_bam_open(path.encode(), 2, C.byref(h)) == 0:
            rc = 1
            while rc > 0:
                bp = C.POINTER(records.SpxBatch)()
                rc = L.spx_bam_next_batch(h, 16, C.byref(bp))
            assert rc < 0, what
            L.spx_bam_close(h)
    # the context is still usable afterwards
    assert _device_run(bam, ctx, par, g.ref)[2] == [60]


def _w(path, data):
    open(path, "wb").write(data)
    return path


def test_device_input_on_several_lanes(ctx, tmp_path):
    """two input pipelines (here two contexts on the one GPU of the box): segments are dealt to whichever lane is free, the
    carry of a segment crosses to the other lane's buffer, the work lists still come out in file order"""
    g = small_genome(synth.HIFI, max_secondaries=3, n_paralogs=3, read_len=5000)
    n = 400
    chunks = [g.reads(k, 50) for k in range(0, n, 50)]
    whole = g.reads(0, n)
    bam = str(tmp_path / "lanes.bam")
    synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=2)
    par = records.preset("hifi")
    ctx.set_reference(g.ref)
    want, want_names = _expected(ctx, whole, par)
    c2 = api.Context(0)
    c2.set_reference(g.ref)
    os.environ["SPX_DIN_SEG_KB"] = "256"
    try:
        d = api.DeviceBam(bam, [ctx, c2], par, g.ref)
    finally:
        os.environ.pop("SPX_DIN_SEG_KB")
    res, names, lanes = [], [], set()
    while True:
        nx = d.next()
        if nx is None:
            break
        w, lane, nb, cnt = nx
        lanes.add(lane)
        w.prepare_staged()
        w.launch()
        out = w.collect(finalize_seed=None)
        res += [_key(out[k]) for k in range(cnt)]
        names += _names(nb)
        w.free()
        d.release(nb)
    d.close()
    c2.close()
    assert lanes == {0, 1} and res == want and names == want_names


def test_device_input_on_shards_of_the_file(ctx, tmp_path):
    """start_voffset / end_voffset (the group-start index in the reference's format, src/secphase_index.c:76-119): the device input
    on consecutive shards of a file yields exactly the file's groups -- shard starts and ends in the middle of BGZF blocks,
    with small segments so that a shard has several"""
    g = small_genome(synth.HIFI, max_secondaries=3, n_paralogs=3, read_len=4000)
    n = 300
    chunks = [g.reads(k, 50) for k in range(0, n, 50)]
    whole = g.reads(0, n)
    bam = str(tmp_path / "shards.bam")
    synth.write_bam(bam, [c.batch for c in chunks], g.ref, threads=2)
    par = records.preset("hifi")
    ctx.set_reference(g.ref)
    want, want_names = _expected(ctx, whole, par)
    L = api.lib()
    L.spx_bam_index_build.argtypes = [C.c_char_p, C.c_int, C.c_int32, C.POINTER(C.c_int64), C.c_int64]
    L.spx_bam_index_build.restype = C.c_int64
    off = (C.c_int64 * 64)()
    k = L.spx_bam_index_build(bam.encode(), 2, 37, off, 64)
    assert k == (n + 36) // 37 + 1
    assert any(off[i] & 0xffff for i in range(k)), "no shard boundary inside a block: the test would not test much"
    res, names = [], []
    for cut in ((0, 3), (3, 4), (4, k - 1)):
        r, nm, sizes = _device_run(bam, ctx, par, g.ref, env={"SPX_DIN_SEG_KB": "256"}, start_voffset=off[cut[0]], end_voffset=off[cut[1]])
        assert sum(sizes) == min(n, cut[1] * 37) - cut[0] * 37
        res += r
        names += nm
    assert res == want and names == want_names
