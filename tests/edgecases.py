"""Hand-built alignment groups (no generator) for the edge cases SURVEY section 7 step 2 lists -- each group is made so that ONE
of them must occur, and `expected` says what the host plan / the oracle must then show, so that a test can assert the case really
happened before it compares the HIP path with the oracle:

  edges        marker at base sqs+10 (first row with a BAQ value) and at sqe-10 (written, then zeroed), markers inside the 10-base margins
               (quality 0)                                                       ptMarker.c:709-717,786,797-803
  in_insertion a mismatch marker of one alignment inside ANOTHER alignment's insertion -> dropped       ptMarker.c:178-182
  all_mismatch a read position that mismatches in every alignment -> removed (read error)                ptMarker.c:219-233
  clips        markers inside another alignment's soft clip / hard clip; a reverse-strand record with a leading hard clip
                                                                                 ptMarker.c:84-104,168,188; cigar_it.c:277-300
  long_cs      long-form cs (`=ACGT`) and `~` characters, which the reference's un-anchored regex skips   cigar_it.c:148,209
  tie3         three secondaries with the same greatest score -> rand() % 3                              ptAlignment.c:165-171
  ten / eleven a group of 10 records is scored, one of 11 is not dispatched                              secphase.c:285-288,337
  guard        (ONT, -b 50) a window of <= 50 bases: l_query <= bw and 2 bw + 1 > l_ref, the regime of probaln.c's terminal guard

Every record is written as a list of edit operations against the READ; the reference contigs are then made to fit (the read is the
given, as for an aligner)."""
import numpy as np

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
_B = "ACGT"


def revcomp(s):
    return "".join(_COMP[c] for c in reversed(s))


def other(b, k=1):
    return _B[(_B.index(b) + k) % 4]


class Aln:
    """one record: ops = [("=", n) | ("X", n) | ("I", n) | ("D", n) | ("S", n) | ("H", n)] over the read in RECORD orientation
    (reverse records are given the reverse complement of the read).  Produces SEQ, CIGAR, cs and the reference segment."""

    def __init__(self, read, ops, reverse=False, long_cs=False, intron_char=False):
        self.reverse = reverse
        r = revcomp(read) if reverse else read
        seq, ref, cig, cs = [], [], [], []
        i = 0  # position in r
        rng = np.random.default_rng(len(read) * 7 + len(ops))
        for op, n in ops:
            if op == "=":
                seg = r[i:i + n]
                seq.append(seg); ref.append(seg)
                cig.append((n, "M"))
                cs.append("=" + seg if long_cs else f":{n}")
                i += n
            elif op == "X":
                seg = r[i:i + n]
                rf = "".join(other(c) for c in seg)
                seq.append(seg); ref.append(rf)
                cig.append((n, "M"))
                cs.append("".join(f"*{a.lower()}{b.lower()}" for a, b in zip(rf, seg)))
                i += n
            elif op == "I":
                seg = r[i:i + n]
                seq.append(seg)
                cig.append((n, "I"))
                cs.append("+" + seg.lower())
                i += n
            elif op == "D":
                rf = "".join(_B[int(x)] for x in rng.integers(0, 4, n))
                ref.append(rf)
                cig.append((n, "D"))
                cs.append("-" + rf.lower())
            elif op == "S":
                seq.append(r[i:i + n])
                cig.append((n, "S"))
                i += n
            elif op == "H":
                cig.append((n, "H"))
                i += n
            else:
                raise ValueError(op)
        assert i == len(r), (i, len(r))
        merged = []
        for n, o in cig:
            if merged and merged[-1][1] == o:
                merged[-1] = (merged[-1][0] + n, o)
            else:
                merged.append((n, o))
        self.seq = "".join(seq)
        self.ref = "".join(ref)
        self.cigar = "".join(f"{n}{o}" for n, o in merged)
        self.cs = "".join(cs)
        if intron_char:  # a character no short-form token starts with, in front of a token: the un-anchored search skips it
            self.cs = self.cs.replace("*", "~*", 1)


def _rand_seq(rng, n):
    return "".join(_B[int(x)] for x in rng.integers(0, 4, n))


class Builder:
    """collects groups; every record's reference segment becomes part of a contig of its own"""

    def __init__(self, seed=1):
        self.rng = np.random.default_rng(seed)
        self.contigs, self.groups, self.expected = [], [], {}

    def place(self, aln):
        pad = _rand_seq(self.rng, 700)
        name = f"c{len(self.contigs)}"
        self.contigs.append((name, pad + aln.ref + _rand_seq(self.rng, 700)))
        return len(self.contigs) - 1, len(pad)

    def group(self, name, read, recs, qual=40):
        """recs = [(is_secondary, Aln)]"""
        out = []
        for sec, a in recs:
            tid, pos = self.place(a)
            flag = (256 if sec else 0) | (16 if a.reverse else 0)
            q = qual if isinstance(qual, int) else list(qual[:len(a.seq)])
            out.append((flag, tid, pos, a.cigar, a.seq, q, a.cs))
        self.groups.append((name, out))
        return len(self.groups) - 1


def hifi_cases(seed=11):
    """the HiFi-preset cases; returns (Builder, {case: group index})"""
    b = Builder(seed)
    rng = b.rng
    at = {}
    # ---- edges / in_insertion / all_mismatch: one group, read of 2000 bases, two forward records
    T = 2000
    read = _rand_seq(rng, T)

    def ops_with_x(T, xs, extra=None):
        """'=' everywhere, single-base 'X' at the read positions xs; extra = {pos: (op, n)} replaces the base(s) starting there"""
        extra = extra or {}
        ops, i = [], 0
        marks = sorted(set(xs) | set(extra))
        for p in marks:
            if p > i:
                ops.append(("=", p - i))
            if p in extra:
                op, n = extra[p]
                ops.append((op, n))
                i = p + (n if op in ("I", "X", "S", "H") else 0)
            else:
                ops.append(("X", 1))
                i = p + 1
        if i < T:
            ops.append(("=", T - i))
        return ops
    a_prim = Aln(read, ops_with_x(T, [5, 10, 700, 1200, 1989, 1995]))
    a_sec = Aln(read, ops_with_x(T, [300, 1200], extra={698: ("I", 4)}))
    at["edges"] = b.group("edges", read, [(False, a_prim), (True, a_sec)])
    # surviving marker positions: 5, 10, 300, 1989, 1995 (700 lies in the secondary's insertion, 1200 mismatches everywhere)
    b.expected["edges"] = {"positions": 5, "n_aln": 2, "row_first": 11, "zeroed_per_alignment": 3}
    # ---- clips: forward primary; reverse secondary with a LEADING hard clip (= the read's end); forward secondary with a soft clip
    T2 = 1500
    read2 = _rand_seq(rng, T2)
    c_prim = Aln(read2, ops_with_x(T2, [100, 500, 800, 1450]))
    # reverse record: its orientation is the reverse complement; leading H100 hides read positions 1400..1499 (forward coordinates)
    c_rev = Aln(read2, [("H", 100)] + ops_with_x(T2 - 100, [T2 - 1 - 900 - 100]), reverse=True)  # one X at record position 599 = read position 900
    c_soft = Aln(read2, [("S", 200)] + ops_with_x(T2 - 200, [600 - 200]))  # X at forward position 600
    at["clips"] = b.group("clips", read2, [(False, c_prim), (True, c_rev), (True, c_soft)])
    # primary markers at 100 (in the soft clip of c_soft) and 1450 (in the hard clip of c_rev) are dropped; 500, 800 stay; 900 (reverse record), 600 stay
    b.expected["clips"] = {"positions": 4, "n_aln": 3}
    # ---- long-form cs and '~'
    T3 = 900
    read3 = _rand_seq(rng, T3)
    l_prim = Aln(read3, ops_with_x(T3, [200, 450]), long_cs=True)
    l_sec = Aln(read3, ops_with_x(T3, [300, 600]), intron_char=True)
    at["long_cs"] = b.group("longcs", read3, [(False, l_prim), (True, l_sec)])
    # ---- tie3: the primary mismatches at five places where the three (identical) secondaries match: they tie above it
    T4 = 1200
    read4 = _rand_seq(rng, T4)
    t_prim = Aln(read4, ops_with_x(T4, [150, 350, 550, 750, 950]))
    t_secs = [Aln(read4, ops_with_x(T4, [])) for _ in range(3)]
    at["tie3"] = b.group("tie3", read4, [(False, t_prim)] + [(True, s) for s in t_secs])
    b.expected["tie3"] = {"tie_bits": 3}
    # ---- ten / eleven records
    T5 = 600
    read5 = _rand_seq(rng, T5)
    at["ten"] = b.group("ten", read5, [(False, Aln(read5, ops_with_x(T5, [100, 300])))] + [(True, Aln(read5, ops_with_x(T5, [200 + 7 * k]))) for k in range(9)])
    at["eleven"] = b.group("eleven", read5, [(False, Aln(read5, ops_with_x(T5, [100, 300])))] + [(True, Aln(read5, ops_with_x(T5, [200 + 7 * k]))) for k in range(10)])
    return b, at


def ont_cases(seed=12):
    """the `--ont -b 50` case: a read of 45 bases, one window of 45 <= bw = 50 bases (2 bw + 1 = 101 > l_ref)"""
    b = Builder(seed)
    T = 45
    read = _rand_seq(b.rng, T)
    g_prim = Aln(read, [("=", 20), ("X", 1), ("=", 24)])
    g_sec = Aln(read, [("=", 30), ("X", 1), ("=", 14)])
    at = {"guard": b.group("guard", read, [(False, g_prim), (True, g_sec)], qual=30)}
    b.expected["guard"] = {"positions": 2, "n_aln": 2}
    return b, at
