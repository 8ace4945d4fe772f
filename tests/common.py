"""Shared helpers of the test-suite (oracle access, synthetic inputs, emulation of the
device kernels with the oracle's DP so that the HOST logic can be checked without a GPU)."""
import ctypes as C

import numpy as np

from oracle import orc
from secphase_amd import api, records, synth

NT16_TABLE = np.full(256, 15, np.uint8)
for ch, v in zip("=ACMGRSVTWYHKDBN", range(16)):
    NT16_TABLE[ord(ch)] = v
    NT16_TABLE[ord(ch.lower())] = v
for ch, v in zip("0123", (1, 2, 4, 8)):
    NT16_TABLE[ord(ch)] = v
NT16_INT = np.array([4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4], np.uint8)


def small_genome(platform, **kw):
    base = dict(n_contigs=2, contig_len=150000)
    base.update(kw)
    cfg = synth.default_cfg(platform, **base)
    return synth.Genome(cfg)


def ref_codes(ref, tid, start, n):
    r = ref.contents
    off = r.seq_off[tid] + start
    raw = np.frombuffer(C.string_at(r.bases + off, n), np.uint8)
    return NT16_INT[NT16_TABLE[raw]]


def nibbles(buf_ptr, nib_off, n):
    """decode n 4-bit codes starting at nibble nib_off (low nibble first)"""
    b0 = nib_off // 2
    nb = (nib_off + n + 1) // 2 - b0
    raw = np.ctypeslib.as_array(buf_ptr, shape=(b0 + nb,))[b0:b0 + nb]
    out = np.empty(nb * 2, np.uint8)
    out[0::2] = raw & 0xf
    out[1::2] = raw >> 4
    s = nib_off - 2 * b0
    return out[s:s + n].copy()


def oracle_probaln(ref, qry, set_q, d, e, bw):
    L = orc.lib()
    ref = np.ascontiguousarray(ref, np.uint8)
    qry = np.ascontiguousarray(qry, np.uint8)
    iq = np.full(len(qry), set_q, np.uint8)
    st = np.zeros(len(qry), np.int32)
    q = np.zeros(len(qry), np.uint8)
    par = orc.ProbalnPar(d, e, bw)
    u8 = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint8))
    pr = L.orc_probaln_glocal(u8(ref), len(ref), u8(qry), len(qry), u8(iq), C.byref(par),
                              st.ctypes.data_as(C.POINTER(C.c_int)), u8(q))
    return pr, st, q


def emulate_rows(plan, ref, params):
    """The oracle DP on every problem of a host plan + the device-side write-back rule: BAQ value per wanted row."""
    v = plan.view
    bq = np.zeros(max(v.n_rows, 1), np.int64)
    for p in range(v.n_problems):
        L_, R_, bw = v.L[p], v.R[p], v.bw[p]
        r = ref_codes(ref, v.ref_tid[p], v.ref_rfs[p], R_)
        q = nibbles(v.qry4, v.qry_nib[p], L_)
        _, st, qq = oracle_probaln(r, q, params.set_q, params.conf_d, params.conf_e, bw)
        for w in range(v.n_rows_of[p]):
            ri = v.row_off[p] + w
            t = v.rows[ri] - 1
            if (st[t] & 3) != 0 or (st[t] >> 2) != v.row_expect[ri]:
                b = 0
            else:
                b = min(int(v.row_rawq[ri]), int(qq[t]))
            bq[ri] = min(b, 93)
    return bq


def batch_qual_copy(batch):
    """writable copy of a batch's qual[] (its extent is not stored: derived from the offsets)"""
    b = batch.contents
    end = max((b.qual_off[a] + b.l_qseq[a] for a in range(b.n_alns)), default=0)
    return np.frombuffer(C.string_at(C.addressof(b.qual.contents), end), np.uint8).copy()


def replay_qual_edits(plan, bq, batch, params):
    """what spx_apply_quals does, from the host plan view (SPX_PAR_ALL_ROWS)"""
    v, b = plan.view, batch.contents
    qual = batch_qual_copy(batch)
    keep = min(params.set_q, 93)
    for k in range(v.n_qedits):
        at = b.qual_off[v.qe_rec[k]] + v.qe_pos[k]
        if v.qe_len[k] == 0:
            qual[at] = 0
            continue
        for t in range(v.qe_len[k]):
            row = v.qe_row0[k] + t
            qual[at + t] = bq[row] if v.row_expect[row] >= 0 else keep
    return qual


def emulate_plan(plan, ref, params):
    """Run the oracle DP on every problem of a host plan and apply the device-side
    write-back + scoring rules in numpy: returns {input group index: (scores, prim, max, tie, pass)}."""
    v = plan.view
    bq = emulate_rows(plan, ref, params)
    match_tbl = (C.c_double * 256)()
    mis_tbl = (C.c_double * 256)()
    thr = (C.c_double * 102)()
    api.lib().spx_host_tables(thr, match_tbl, mis_tbl)
    res = {}
    for k in range(v.n_groups):
        n = v.n_aln[k]
        sec = v.sec_mask[k]
        m0, m1 = v.mk_first[k], v.mk_first[k + 1]
        scores = [0.0] * n
        p = m0
        while p < m1:
            e = p + 1
            while e < m1 and not v.mk_first_of_pos[e]:
                e += 1
            qs = [int(bq[v.mk_row[i]]) if v.mk_row[i] >= 0 else int(v.mk_qfix[i]) for i in range(p, e)]
            mn = min(min(qs), 100)
            if mn > params.min_q:
                for i in range(p, e):
                    a = v.mk_aln[i]
                    scores[a] += match_tbl[mn] if v.mk_is_match[i] else mis_tbl[mn]
            p = e
        prim, mx, mxs, prs = -1, -1, -1.7976931348623157e308, -1.7976931348623157e308
        for a in range(n):
            if not (sec >> a) & 1:
                prim, prs = a, scores[a]
            elif mxs < scores[a]:
                mx, mxs = a, scores[a]
        tie = sum(1 << a for a in range(n) if (sec >> a) & 1 and mxs <= scores[a])
        ok = not (prim == -1 or mxs <= prs + params.prim_margin_score or mxs < params.min_score)
        res[v.grp_index[k]] = (scores, prim, mx, tie, ok)
    return res


# ---------------------------------------------------------------------------
# hand-built batches (SAM-like tuples) for targeted tests
_CIG = {c: i for i, c in enumerate("MIDNSHP=X")}
_NT = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}


class HandBatch:
    """groups = [(qname, [(flag, tid, pos, cigar_str, seq, qual_list_or_int, cs_or_None), ...]), ...]"""

    def __init__(self, groups):
        import re
        self.keep = []
        grp_first, qname_off, qnames = [0], [], b""
        flag, tid, pos, lq, ncig, cig_off, seq_off, qual_off, cs_off, md_off = [], [], [], [], [], [], [], [], [], []
        cig, seq4, qual, cs, md = [], bytearray(), bytearray(), bytearray(), bytearray()
        for name, recs in groups:
            qname_off.append(len(qnames))
            qnames += name.encode() + b"\0"
            for rec in recs:
                (f, t, p, cg, sq, ql, c), m = rec[:7], (rec[7] if len(rec) > 7 else None)
                flag.append(f); tid.append(t); pos.append(p); lq.append(len(sq))
                ops = re.findall(r"(\d+)([MIDNSHP=X])", cg)
                ncig.append(len(ops)); cig_off.append(len(cig))
                cig += [(int(n) << 4) | _CIG[o] for n, o in ops]
                seq_off.append(len(seq4))
                for i in range(0, len(sq), 2):
                    hi = _NT.get(sq[i], 15)
                    lo = _NT.get(sq[i + 1], 15) if i + 1 < len(sq) else 0
                    seq4.append(hi << 4 | lo)
                qual_off.append(len(qual))
                qual += bytes([ql] * len(sq)) if isinstance(ql, int) else bytes(ql)
                if c is None:
                    cs_off.append(-1)
                else:
                    cs_off.append(len(cs)); cs += c.encode() + b"\0"
                if m is None:
                    md_off.append(-1)
                else:
                    md_off.append(len(md)); md += m.encode() + b"\0"
            grp_first.append(len(flag))
        A = lambda x, dt: np.ascontiguousarray(np.array(x, dt))
        self.arr = dict(grp_first=A(grp_first, np.int32), qname_off=A(qname_off, np.int64), flag=A(flag, np.uint16),
                        tid=A(tid, np.int32), pos=A(pos, np.int32), l_qseq=A(lq, np.int32), n_cigar=A(ncig, np.int32),
                        cigar_off=A(cig_off, np.int64), seq_off=A(seq_off, np.int64), qual_off=A(qual_off, np.int64),
                        cs_off=A(cs_off, np.int64), md_off=A(md_off, np.int64), cigar=A(cig if cig else [0], np.uint32),
                        seq4=A(list(seq4) + [0], np.uint8), qual=A(list(qual) + [0], np.uint8))
        self.qnames = C.create_string_buffer(bytes(qnames) + b"\0")
        self.cs = C.create_string_buffer(bytes(cs) + b"\0")
        self.md = C.create_string_buffer(bytes(md) + b"\0")
        b = records.SpxBatch()
        b.n_groups = len(groups)
        b.n_alns = len(flag)
        for k, v in self.arr.items():
            setattr(b, k, v.ctypes.data_as(dict(records.SpxBatch._fields_)[k]))
        b.qnames = C.cast(self.qnames, C.c_void_p)
        b.cs = C.cast(self.cs, C.c_void_p)
        b.md = C.cast(self.md, C.c_void_p)
        self.struct = b
        self.batch = C.pointer(b)


class HandRef:
    def __init__(self, contigs):
        """contigs = [(name, sequence_str), ...]"""
        names, name_off, seq_off, bases = b"", [], [0], b""
        for n, s in contigs:
            name_off.append(len(names)); names += n.encode() + b"\0"
            bases += s.encode(); seq_off.append(len(bases))
        self.names = C.create_string_buffer(names + b"\0")
        self.bases = C.create_string_buffer(bases + b"\0")
        self.name_off = np.array(name_off, np.int64)
        self.seq_off = np.array(seq_off, np.int64)
        r = records.SpxRef()
        r.n_contigs = len(contigs)
        r.name_off = self.name_off.ctypes.data_as(records.c_i64p)
        r.names = C.cast(self.names, C.c_void_p)
        r.seq_off = self.seq_off.ctypes.data_as(records.c_i64p)
        r.bases = C.cast(self.bases, C.c_void_p)
        self.struct = r
        self.ref = C.pointer(r)


def walk(batch, a):
    """oracle CIGAR/cs iterator states of alignment a: list of dicts"""
    ops = C.POINTER(orc.Op)()
    n = orc.lib().orc_walk_cigar(batch, a, C.byref(ops))
    assert n > 0, n
    out = [{f: getattr(ops[i], f) for f, _ in orc.Op._fields_} for i in range(n)]
    C.CDLL(None).free(ops)
    return out
