/*
 * spx_logic.h -- the integer part of the secphase marker path (SURVEY.md section 8 rows A1-A8 and the control flow of
 * A10), written ONCE for both sides: hipcc compiles these functions for gfx950, where one thread owns one alignment
 * (op table, extents, confident blocks, mismatch list) or one read group (marker columns, consensus windows, BAQ work
 * list), and for the host, where the very same code produces the host-only plan view the CPU tests check against the
 * oracle.  No std:: containers, no allocation: every function works on caller-provided arrays with stated capacities
 * and reports SPX_ENOMEM when one would overflow (the runtime then grows its pools and repeats the batch).
 *
 * Behavioural contract, by reference line (/root/reference/programs):
 *   op table            submodules/cigar_it/cigar_it.c:14-69,72-141,145-211,213-308
 *   aligned extents     submodules/ptAlignment/ptAlignment.c:42-95
 *   markers             submodules/ptMarker/ptMarker.c:42-107,156-295
 *   blocks              submodules/ptMarker/ptMarker.c:328-667, src/secphase.c:162-170
 *   BAQ windows / rows  submodules/ptMarker/ptMarker.c:670-831
 *   HMM constants       htslib-1.17 probaln.c initialisation
 */
#ifndef SPX_LOGIC_H
#define SPX_LOGIC_H

#include <stdint.h>

#include "../../include/spx.h"
#include "spx_device.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SPX_HD __host__ __device__ inline
#else
#define SPX_HD inline
#endif

namespace spxl {

/* ---- records as the device sees them: one Rec per alignment of a dispatched group, pools of packed payload ---- */
struct Rec {
    int32_t rec;      /* record index in its input batch */
    int32_t batch;    /* input batch (spx_prepare_many) */
    int32_t grp;      /* dispatched-group index inside the work list */
    int32_t flag, tid, pos, l_qseq, n_cigar;
    int32_t cs_len;   /* -1: no cs tag */
    int32_t md_len;   /* -1: no MD tag (looked at only without cs) */
    int64_t cigar_off; /* uint32 units into the cigar pool */
    int64_t seq_off;   /* BYTE offset into the sequence pools (raw BAM nibbles / recoded 0..4 codes), multiple of 4 */
    int64_t qual_off;  /* BYTE offset into the quality pool */
    int64_t tag_off;   /* BYTE offset of the cs (or MD) string in the text pool, NUL-terminated */
    /* what crosses PCIe (round 3): SEQ / QUAL of a secondary that merely repeat the primary's -- the same bases, or their
     * reverse complement, minus hard clips: verified byte for byte on the staging threads -- are NOT transferred; the device
     * rebuilds them from the primary's (seqqual_alias_kernel).  pk_*: offsets in the packed transfer buffers. */
    int64_t pk_seq_off, pk_qual_off;
    int32_t alias_slot;  /* -1: own bytes were transferred; else the slot (same work list) whose SEQ / QUAL this one repeats */
    int32_t alias_shift; /* base i of this record = base (alias_rev ? alias_shift - i : alias_shift + i) of that slot */
    int32_t alias_rev;   /* reverse complement (qualities reversed) */
    int32_t pad_;
};

struct Op {
    int32_t op, len; /* len doubles as the iterator's return value (0 ends a while(next) loop) */
    int32_t sqs, sqe, rfs, rfe, rds, rde;
};
struct Blk {
    int32_t rfs, rfe, sqs, sqe, rds, rde;
};
struct Iv { /* interval in forward-read coordinates, inclusive */
    int32_t s, e;
};
struct MM { /* mismatch marker of one alignment, in op order */
    int32_t pos, base_idx, q, ref_pos;
};
struct Mk { /* one cell of the marker table: (read position, alignment) */
    int32_t base_idx, ref_pos, row;
    uint8_t q, is_match, pad0, pad1;
};

struct AlnState {
    int32_t lclip, rclip, rfs, rfe, rds, rde;
    int32_t n_ops, n_visit, rest;
    int32_t n_conf, n_mm;
    int32_t err;
    int32_t mm_cap, conf_cap; /* bounds from the counting pass */
    int32_t has_n, pad;       /* SEQ holds a base other than ACGT */
    int64_t ops_off, conf_off, mm_off;
};

struct Pools {
    const uint32_t *cigar;
    const uint8_t *qual;
    const char *text;
    const uint8_t *code4; /* SEQ recoded to 0..4, low nibble first, same byte offsets as the raw pool, behind a lead pad */
    int64_t code_lead_bytes;
    Op *ops;
    Blk *conf;
    MM *mm;
};

/* reference side: contig table of the 4-bit pool + ambiguous-base index */
struct RefView {
    int32_t n_contigs;
    const int64_t *nib_off; /* first nibble of each contig in the device pool */
    const int64_t *len;
    const int64_t *npos_off; /* [n_contigs+1] into npos */
    const int32_t *npos;     /* ascending positions of non-ACGT bases per contig */
};

struct Params {
    int32_t baq_flag, consensus, indel_threshold, min_q, set_q, flank_margin, all_rows;
    int32_t term_guard; /* SPX_GUARD_BAND (default) or SPX_GUARD_ROW: see terminal_drop() */
    double conf_b;
    float d, e;
    float qf; /* (float)pow(10, -set_q/10.), host libm */
    int32_t table_min_cols, table_rounds; /* flank_break_rounds(): groups with at least so many columns, the first so many rounds (<= kTableRounds) */
    int32_t row_mult; /* doubles a wanted row takes in the DP scratch, in units of its class' slots: 2 (M, I rows: the exact tier) or 4 (two-tier DP: the fast
                       * tier's forward rows U, V and backward rows Bm, Bi side by side; the exact tier uses the first half of a problem's share) */
};

/* ---------------- band classes = DP kernel instantiations (keep in step with spx_launch_baq) ---------------- */
SPX_HD int class_slots(int cls)
{
    switch (cls) {
    case 0: return 42; case 1: return 44; case 2: return 46; case 3: return 48; case 4: return 48; case 5: return 64;
    case 6: return 104; case 7: return 128; case 8: return 256; case 9: return 512; case 10: return 1024; case 11: return 2048;
    case 12: return 112; default: return 120;
    }
}
SPX_HD int class_lanes(int cls)
{
    switch (cls) {
    case 0: case 1: case 2: case 3: return 1;
    case 4: return 2; case 5: case 6: case 12: case 13: return 4;
    case 7: return 8; case 8: return 16; case 9: return 32; default: return 64;
    }
}
SPX_HD int class_lanes_bwd(int cls)
{
    switch (cls) {
    case 0: case 1: case 2: case 3: case 4: return 2;
    case 5: case 6: case 7: case 12: case 13: return 4;
    case 8: return 16; case 9: return 32; default: return 64;
    }
}
SPX_HD int band_class(int W)
{
    if (W == 41) return 0;
    if (W == 43) return 1;
    if (W == 45) return 2;
    if (W == 47) return 3;
    if (W <= 48) return 4;
    if (W <= 64) return 5;
    if (W <= 104) return 6;
    if (W <= 112) return 12;
    if (W <= 120) return 13;
    if (W <= 128) return 7;
    if (W <= 256) return 8;
    if (W <= 512) return 9;
    if (W <= 1024) return 10;
    if (W <= 2048) return 11;
    return -1;
}

SPX_HD int effective_bw(int l_ref, int l_query, int bw_in)
{
    int bw = l_ref > l_query ? l_ref : l_query;
    if (bw > bw_in) bw = bw_in;
    const int diff = l_ref > l_query ? l_ref - l_query : l_query - l_ref;
    if (bw < diff) bw = diff;
    return bw;
}

/* The one line of probaln_glocal this repository cannot check against a real htslib 1.17 (PARITY UNPINNED):
 * the guard of the termination sum s[l_query+1] and of the backward start, `if (u < 3 || u >= LIMIT) continue;`.
 *   SPX_GUARD_BAND (0, default): LIMIT = bw2*3+3  -- kprobaln's test: column k of row l_query is skipped iff it lies outside the band.
 *   SPX_GUARD_ROW  (1):          LIMIT = i_dim-3 with i_dim = min(bw2, l_ref)*3+6 (the row length of the htslib releases that shrank
 *                                the matrices).  With u = (k - max(l_query-bw, 0) + 1)*3 the two differ for exactly one cell: column
 *                                k = l_ref of row l_query when l_query <= bw and 2*bw+1 > l_ref -- a real band cell that this reading
 *                                leaves out of s[l_query+1] and of the backward start (b = 0 there).
 * Everything else of the recursion is the same under both.  One switch (spx_set_terminal_guard / SPX_TERMINAL_GUARD, and the oracle's
 * orc_set_terminal_guard) selects the reading for the oracle, the scoring kernels and the general kernel. */
SPX_HD int terminal_drop(int guard, int l_query, int l_ref, int bw) { return guard == 1 && l_query <= bw && 2 * bw + 1 > l_ref; }

SPX_HD int64_t band_cells(int L, int R, int bw)
{
    /* sum over rows i = 1..L of (min(R, i+bw) - max(1, i-bw) + 1), closed form; effective_bw() guarantees bw >= |R - L| */
    const int64_t l = L, r = R, w = bw;
    int64_t a = r - w < l ? r - w : l;
    if (a < 0) a = 0;
    const int64_t hi = a * (a + 1) / 2 + a * w + (l - a) * r;
    const int64_t b = l < w + 1 ? l : w + 1;
    const int64_t lo = b + (l * (l + 1) / 2 - b * (b + 1) / 2) - (l - b) * w;
    return hi - lo + l;
}

/* the float/double mix below is the one of probaln_glocal's initialisation: probaln_par_t holds floats,
 * 1 - c->d - c->d and (1 - c->d) / l_ref are float expressions.  qf = (float)pow(10, -set_q/10.) comes from the host. */
SPX_HD void hmm_constants(int l_ref, int l_query, float d, float e, float qf, double *h)
{
    const double sM = 1. / (2 * l_query + 2), sI = sM;
    h[SPX_H_M0] = (double)((1 - d) - d) * (1 - sM);
    h[SPX_H_M1] = (double)d * (1 - sM);
    h[SPX_H_M2] = h[SPX_H_M1];
    h[SPX_H_M3] = (double)(1 - e) * (1 - sI);
    h[SPX_H_M4] = (double)e * (1 - sI);
    h[SPX_H_M6] = (double)(1 - e);
    h[SPX_H_M8] = (double)e;
    h[SPX_H_BM] = (double)((1 - d) / l_ref);
    h[SPX_H_BI] = (double)(d / l_ref);
    h[SPX_H_SM] = sM;
    h[SPX_H_SI] = sI;
    h[SPX_H_EMATCH] = 1. - (double)qf;
    h[SPX_H_EMIS] = (double)qf * .33333333333;
    h[SPX_H_PAD0] = h[SPX_H_TDROP] = h[SPX_H_PAD2] = 0.;
}

SPX_HD bool window_has_n(const RefView &rv, int tid, int64_t start, int64_t n)
{
    if (!rv.npos_off || tid < 0 || tid >= rv.n_contigs) return false;
    int64_t lo = rv.npos_off[tid], hi = rv.npos_off[tid + 1];
    const int64_t end = hi;
    while (lo < hi) {
        const int64_t m = (lo + hi) / 2;
        if (rv.npos[m] < start) lo = m + 1; else hi = m;
    }
    return lo < end && rv.npos[lo] < start + n;
}

/* ---------------- tokenisers ---------------- */
SPX_HD bool lower_c(char c) { return c >= 'a' && c <= 'z'; }
SPX_HD bool upper_c(char c) { return c >= 'A' && c <= 'Z'; }
SPX_HD bool digit_c(char c) { return c >= '0' && c <= '9'; }

/* A tag string as the walks read it.  On the device a character comes out of an 8-byte window that is re-loaded when the cursor leaves
 * it: ONE load per eight characters instead of one or more per character -- the tokenizers are chains of dependent loads, and a lone
 * lane walking a 35 KB cs string spends its time waiting for them (round 5).  The window is an ALIGNED 8-byte word: up to seven bytes in
 * front of the string and behind its NUL are read and never looked at; the text pool starts on a 256-byte boundary and ends with 16
 * spare bytes (spx_prep.cpp stage_layout).  On the host the same code reads byte by byte. */
struct TextView {
    const char *s;
    uint64_t w;
    uintptr_t wa;
    SPX_HD explicit TextView(const char *p) : s(p), w(0), wa(1) {} /* (no aligned address equals 1) */
    SPX_HD char operator[](int i)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const uintptr_t a = (uintptr_t)(s + i), al = a & ~(uintptr_t)7;
        if (al != wa) { w = *reinterpret_cast<const uint64_t *>(al); wa = al; }
        return (char)(w >> ((a & 7) * 8));
#else
        return s[i];
#endif
    }
};

/* first short-form cs token at or after t[at] (what an un-anchored POSIX search of
 * (:[0-9]+)|([+-][a-z]+)|((\*[a-z]+)+) returns); false if none.  so / eo are relative to `at`; num = the decimal value of a ':' token
 * (atoi of its digits: wraps like 32-bit arithmetic on absurd input) */
SPX_HD bool next_cs_token(TextView &t, int at, int &so, int &eo, int &num)
{
    for (int p = 0; t[at + p]; ++p) {
        const char c = t[at + p];
        if (c == ':') {
            if (!digit_c(t[at + p + 1])) continue;
            int e = p + 1;
            uint32_t v = 0;
            for (char d = t[at + e]; digit_c(d); d = t[at + ++e]) v = v * 10u + (uint32_t)(d - '0');
            so = p; eo = e; num = (int)v;
            return true;
        }
        if (c == '+' || c == '-') {
            if (!lower_c(t[at + p + 1])) continue;
            int e = p + 1;
            while (lower_c(t[at + e])) ++e;
            so = p; eo = e;
            return true;
        }
        if (c == '*') {
            if (!lower_c(t[at + p + 1])) continue;
            int e = p;
            while (t[at + e] == '*' && lower_c(t[at + e + 1])) {
                ++e;
                while (lower_c(t[at + e])) ++e;
            }
            so = p; eo = e;
            return true;
        }
    }
    return false;
}

/* first MD token at or after t[at]: a mismatch run X(0X)*, a match count, or a deletion ^XXX (cigar_it.h:10) */
SPX_HD bool next_md_token(TextView &t, int at, int &so, int &eo)
{
    for (int p = 0; t[at + p]; ++p) {
        const char c = t[at + p];
        if (upper_c(c)) {
            int e = p + 1;
            while (t[at + e] == '0' && upper_c(t[at + e + 1])) e += 2;
            so = p; eo = e;
            return true;
        }
        if (digit_c(c)) {
            int e = p + 1;
            while (digit_c(t[at + e])) ++e;
            so = p; eo = e;
            return true;
        }
        if (c == '^' && upper_c(t[at + p + 1])) {
            int e = p + 1;
            while (upper_c(t[at + e])) ++e;
            so = p; eo = e;
            return true;
        }
    }
    return false;
}

/* one MD step (cigar_it.c:72-141): a lone "0" separates two mismatches and is skipped */
SPX_HD int md_step(TextView &md, int &at, Op &cur)
{
    for (;;) {
        int so, eo;
        if (!next_md_token(md, at, so, eo)) return 0;
        const char c = md[at + so];
        if (c == '0') { cur.op = SPX_CDIFF; cur.len = 0; }
        else if (c <= '9') {
            /* sic: the reference copies eo - so characters from the START of the shifted string (not from so) and
             * atoi()s them; at most 19 digits reach its buffer */
            int n = eo - so;
            if (n > 19) n = 19;
            uint32_t v = 0;
            for (int k = 0; k < n && digit_c(md[at + k]); ++k) v = v * 10u + (uint32_t)(md[at + k] - '0');
            cur.op = SPX_CEQUAL;
            cur.len = (int)v;
        } else if (c < 90) { cur.op = SPX_CDIFF; cur.len = 1 + (eo - so - 1) / 2; }
        else if (c == '^') { cur.op = SPX_CDEL; cur.len = eo - so - 1; }
        at += eo;
        if (cur.len != 0) return cur.len;
    }
}

SPX_HD bool mx(int op) { return op == SPX_CMATCH || op == SPX_CEQUAL || op == SPX_CDIFF; }

/* ---------------- per-alignment pass ----------------
 * EMIT = false: counts the op table (st.n_ops) and bounds the mismatch list and the confident blocks (st.mm_cap,
 * st.conf_cap); EMIT = true: writes ops[0..n_ops) (ops[0] = state before the first step).  Both fill lclip/rclip,
 * n_visit, rest and return 0 or SPX_E*. */
/* Upper bounds of what build_ops / finish_alignment produce, from the record's lengths alone (no payload byte read):
 * an op per CIGAR op and per tag token (a cs token takes two characters or more, an MD token one), a confident block per
 * CIGAR op, a mismatch per three cs characters / per MD character.  The device sizes its tables with these and parses
 * every tag ONCE; an alignment that outgrows them (malformed tags: the reference's tokenizer then re-uses the previous
 * token) sends the whole batch through the exact counting pass instead. */
SPX_HD void aln_caps(const Rec &r, int32_t &ops_cap, int32_t &conf_cap, int32_t &mm_cap)
{
    const int64_t nc = r.n_cigar > 0 ? r.n_cigar : 0;
    const int64_t tl = r.cs_len >= 0 ? r.cs_len : (r.md_len >= 0 ? r.md_len : 0);
    const int64_t tokens = r.cs_len >= 0 ? tl / 2 : tl;
    int64_t mm = r.cs_len >= 0 ? tl / 3 : tl;
    const int64_t lq = r.l_qseq > 0 ? r.l_qseq : 0;
    if (mm > lq) mm = lq;
    const int64_t oc = nc + tokens + 4;
    ops_cap = (int32_t)(oc > 0x3fffffff ? 0x3fffffff : oc);
    conf_cap = (int32_t)(nc + 2 > 0x3fffffff ? 0x3fffffff : nc + 2);
    mm_cap = (int32_t)(mm + 1);
}

/* cap: room in ops[] (EMIT only; 0 = the table was sized by the counting pass) */
template <bool EMIT>
SPX_HD int build_ops(const Rec &r, const Pools &P, int min_q, int indel_thr, AlnState &st, Op *ops, int32_t cap = 0)
{
    if (r.n_cigar <= 0) return SPX_EINVAL;
    const uint32_t *cigar = P.cigar + r.cigar_off;
    const uint8_t *qual = P.qual + r.qual_off;
    st.lclip = ((cigar[0] & 0xf) == SPX_CHARD_CLIP) ? (int32_t)(cigar[0] >> 4) : 0;
    st.rclip = ((cigar[r.n_cigar - 1] & 0xf) == SPX_CHARD_CLIP) ? (int32_t)(cigar[r.n_cigar - 1] >> 4) : 0;
    const bool use_cs = r.cs_len >= 0, use_md = !use_cs && r.md_len >= 0;
    if (!use_cs && !use_md) return SPX_ENOTAG; /* neither cs nor MD: the reference exits (cigar_it.c:64-67) */
    TextView tag(P.text + r.tag_off);
    const bool rev = (r.flag & SPX_FREVERSE) != 0;
    int md_at = 0;
    Op cur;
    cur.op = 255; cur.len = 0;
    cur.sqs = 0; cur.sqe = -1;
    cur.rfs = r.pos; cur.rfe = r.pos - 1;
    const int32_t T = st.lclip + st.rclip + r.l_qseq;
    cur.rds = rev ? T : 0;
    cur.rde = rev ? T - 1 : -1;
    int32_t n = 0;
    if (EMIT) ops[0] = cur;
    n = 1;
    int idx = -1, remain = 0, cs_at = 0;
    bool aligned = false, visiting = true;
    int32_t n_visit = -1, mm = 0, cf = 0;
    while (idx != r.n_cigar - 1) {
        ++idx;
        const int op = (int)(cigar[idx] & 0xf), len = (int)(cigar[idx] >> 4);
        int rd, sq, rf;
        const bool mtype = op == SPX_CMATCH || op == SPX_CEQUAL || op == SPX_CDIFF;
        if (use_cs && (mtype || op == SPX_CINS || op == SPX_CDEL)) {
            int so, eo, num = 0;
            if (next_cs_token(tag, cs_at, so, eo, num)) {
                const char c = tag[cs_at + so];
                if (c == ':') { cur.op = SPX_CEQUAL; cur.len = num; }
                else if (c == '*') { cur.op = SPX_CDIFF; cur.len = (eo - so + 1) / 3; }
                else if (c == '+') { cur.op = SPX_CINS; cur.len = eo - so - 1; }
                else { cur.op = SPX_CDEL; cur.len = eo - so - 1; }
                cs_at += eo;
            }
        }
        if (mtype) {
            if (remain == 0) remain = len;
            if (use_cs) {
                remain -= cur.len;
                if (remain > 0) --idx; /* stay on this CIGAR op until cs has covered it */
            } else {
                /* MD knows nothing about insertions: a match run may reach into the following M ops (remain < 0) */
                if (remain >= 0) md_step(tag, md_at, cur);
                if (remain < 0) {
                    cur.op = SPX_CEQUAL;
                    cur.len = len < -remain ? len : -remain;
                    remain += len;
                } else {
                    const int md_len = cur.len;
                    cur.len = cur.len < remain ? cur.len : remain;
                    remain -= md_len;
                }
                if (remain > 0) --idx;
            }
            rd = sq = rf = cur.len;
        } else if (op == SPX_CINS) {
            cur.len = len; cur.op = op;
            rd = sq = len; rf = 0;
        } else if (op == SPX_CDEL) {
            if (use_md) md_step(tag, md_at, cur);
            rd = sq = 0; rf = len;
        } else if (op == SPX_CSOFT_CLIP) {
            cur.len = len; cur.op = op;
            rd = sq = len; rf = 0;
        } else if (op == SPX_CHARD_CLIP) {
            cur.len = len; cur.op = op;
            rd = len; sq = 0; rf = 0;
        } else {
            return SPX_EUNSUPPORTED; /* N / P / B: undefined in the reference (cigar_it.c:225-291) */
        }
        if (rev) { cur.rde = cur.rds - 1; cur.rds -= rd; }
        else { cur.rds = cur.rde + 1; cur.rde += rd; }
        cur.sqs = cur.sqe + 1; cur.sqe += sq;
        cur.rfs = cur.rfe + 1; cur.rfe += rf;
        if (EMIT) {
            if (cap && n >= cap) return SPX_ENOMEM;
            ops[n] = cur;
        }
        aligned = aligned || (mx(cur.op) && cur.len > 0);
        if (visiting && cur.len == 0) { visiting = false; n_visit = n; }
        if (!EMIT && visiting) {
            if (cur.op == SPX_CDIFF) {
                for (int j = 0; j < cur.len; ++j) {
                    const int b = cur.sqs + j;
                    if (b >= 0 && b < r.l_qseq && (int)qual[b] >= min_q) ++mm;
                }
            }
            const bool indel = cur.op == SPX_CINS || cur.op == SPX_CDEL, clip = cur.op == SPX_CSOFT_CLIP || cur.op == SPX_CHARD_CLIP;
            if (clip || (indel && cur.len > indel_thr)) ++cf;
        }
        ++n;
        if (n > 40000000) return SPX_EINVAL;
    }
    /* U6: no aligned base at all (e.g. a CIGAR of clips only): undefined in the reference, rejected like U3 */
    if (!aligned || r.l_qseq <= 0) return SPX_EUNSUPPORTED;
    st.n_ops = n;
    if (n_visit < 0) { st.n_visit = n; st.rest = n - 1; }
    else { st.n_visit = n_visit; st.rest = n_visit; }
    if (!EMIT) { st.mm_cap = mm; st.conf_cap = cf + 1; }
    return 0;
}

/* aligned extents (ptAlignment_init_coordinates), confident blocks (find_confident_blocks) and the mismatch
 * markers with raw quality >= min_q (ptMarker_get_initial_markers), all from the op table */
SPX_HD int finish_alignment(const Rec &r, const Pools &P, int min_q, int thr, AlnState &st, const Op *ops, Blk *conf, MM *mmv)
{
    const bool rev = (r.flag & SPX_FREVERSE) != 0;
    const uint8_t *qual = P.qual + r.qual_off;
    st.rfs = st.rfe = st.rds = st.rde = -1;
    /* ONE pass over the visited ops for the three lists (round 5; they were three passes: every op was loaded three times by a lane that
     * does nothing but wait for its loads), four ops fetched at a time so that their loads are in flight together.  Each list is
     * appended in op order, as before. */
    int nc = 0, nm = 0;
    int c_sqs = 0, c_rfs = r.pos;
    int c_rd = rev ? ops[0].rde : ops[0].rds;
    for (int t0 = 1; t0 < st.n_visit; t0 += 4) {
        Op quad[4];
        const int m = st.n_visit - t0 < 4 ? st.n_visit - t0 : 4;
        for (int k = 0; k < 4; ++k)
            if (k < m) quad[k] = ops[t0 + k];
        for (int k = 0; k < 4; ++k) {
            if (k >= m) break;
            const Op o = quad[k];
            /* aligned extents (ptAlignment_init_coordinates) */
            if (st.rfs == -1 && mx(o.op)) {
                st.rfs = o.rfs;
                if (rev) st.rde = o.rde; else st.rds = o.rds;
            }
            if (st.rfe == -1 && st.rfs != -1 && (o.op == SPX_CHARD_CLIP || o.op == SPX_CSOFT_CLIP)) {
                st.rfe = o.rfe;
                if (rev) st.rds = o.rde + 1; else st.rde = o.rds - 1;
            }
            /* confident blocks (find_confident_blocks) */
            const bool indel = o.op == SPX_CINS || o.op == SPX_CDEL;
            const bool clip = o.op == SPX_CSOFT_CLIP || o.op == SPX_CHARD_CLIP;
            if ((indel && o.len > thr) || clip) {
                if (c_sqs < o.sqs && c_rfs < o.rfs) {
                    if (nc >= st.conf_cap) return SPX_ENOMEM;
                    Blk b;
                    b.rfs = c_rfs; b.rfe = o.rfs - 1; b.sqs = c_sqs; b.sqe = o.sqs - 1;
                    if (rev) { b.rds = o.rde + 1; b.rde = c_rd; } else { b.rds = c_rd; b.rde = o.rds - 1; }
                    conf[nc++] = b;
                }
                c_sqs = o.sqe + 1;
                c_rfs = o.rfe + 1;
                c_rd = rev ? o.rds - 1 : o.rde + 1;
            }
            /* mismatch bases with raw quality >= min_q, in op order (ascending read position on the forward strand,
             * descending on the reverse strand) */
            if (o.op == SPX_CDIFF) {
                for (int j = 0; j < o.len; ++j) {
                    const int b = o.sqs + j;
                    if (b < 0 || b >= r.l_qseq) continue; /* cs longer than SEQ: the reference would read past the record */
                    const int q = qual[b];
                    if (q < min_q) continue;
                    if (nm >= st.mm_cap) return SPX_ENOMEM;
                    MM mrec;
                    mrec.base_idx = b; mrec.pos = rev ? o.rde - j : o.rds + j; mrec.q = q; mrec.ref_pos = o.rfs + j;
                    mmv[nm++] = mrec;
                }
            }
        }
    }
    {
        const Op o = ops[st.rest];
        if (st.rfe == -1 && mx(o.op)) {
            st.rfe = o.rfe;
            if (rev) st.rds = o.rds; else st.rde = o.rde;
        }
        if (c_sqs <= o.sqe) {
            if (nc >= st.conf_cap) return SPX_ENOMEM;
            Blk b;
            b.rfs = c_rfs; b.rfe = o.rfe; b.sqs = c_sqs; b.sqe = o.sqe;
            if (rev) { b.rds = o.rds; b.rde = c_rd; } else { b.rds = c_rd; b.rde = o.rde; }
            conf[nc++] = b;
        }
    }
    st.n_conf = nc;
    st.n_mm = nm;
    return 0;
}

/* ---------------- per-group pass ---------------- */
struct GroupView {
    int32_t n;             /* alignments */
    const Rec *rec;        /* [n] */
    AlnState *st;          /* [n] */
};

/* what one group needs in its scratch arena (bytes), from the per-alignment counts */
struct GroupArena {
    int32_t P_cap, blk_cap, rows_cap;
    int64_t o_pos, o_mk, o_keep, o_flank, o_cur, o_nxt, o_proj, o_nproj, o_state, o_rowsmk, bytes;
};
SPX_HD GroupArena group_arena_layout(const GroupView &G, bool all_rows, int slack)
{
    GroupArena A;
    int64_t P = 0, C = 0, lq = 0;
    for (int i = 0; i < G.n; ++i) {
        P += G.st[i].n_mm;
        C += G.st[i].n_conf;
        if (G.rec[i].l_qseq > lq) lq = G.rec[i].l_qseq;
    }
    A.P_cap = (int32_t)P;
    /* interval lists: the intersection of k sorted lists has at most the sum of their lengths; flanking windows are at
     * most one per marker cell in the degenerate margin-0 case, normally about one per position.  `slack` multiplies
     * the estimate after an overflow. */
    int64_t bc = C + 2 * P + 16;
    if (slack > 1) bc = C + (int64_t)slack * P * (G.n > 0 ? G.n : 1) + 16;
    A.blk_cap = (int32_t)(bc > 0x3fffffff ? 0x3fffffff : bc);
    A.rows_cap = (int32_t)(all_rows ? lq + 8 : P + 8);
    int64_t o = 0;
    auto take = [&](int64_t bytes) { const int64_t at = o; o += (bytes + 15) & ~(int64_t)15; return at; };
    A.o_pos = take(4 * (P + 1));
    A.o_mk = take((int64_t)sizeof(Mk) * (P * G.n + 1));
    A.o_keep = take(P + 1);
    A.o_flank = take((int64_t)sizeof(Iv) * A.blk_cap);
    A.o_cur = take((int64_t)sizeof(Iv) * A.blk_cap);
    A.o_nxt = take((int64_t)sizeof(Iv) * A.blk_cap);
    A.o_proj = take((int64_t)sizeof(Blk) * A.blk_cap * G.n);
    A.o_nproj = take(4 * 32); /* [0..10): block counts per alignment, [16..26): their too-long flags */
    A.o_state = take(64);
    A.o_rowsmk = take(4 * (int64_t)A.rows_cap * (G.n > 0 ? G.n : 1)); /* one list per alignment: their passes run side by side */
    A.bytes = o;
    return A;
}

struct GroupCount {
    int32_t err;      /* 0 or SPX_E* (the group is then reported with that code and contributes nothing) */
    int32_t scored;
    int32_t n_cols;   /* marker columns kept (markers = n_cols * n) */
    int32_t n_prob, n_rows, n_qe;
    int64_t cells, s_need, f_need;
    int32_t cls_prob[SPX_N_CLASSES];
    int64_t cls_cells[SPX_N_CLASSES];
};

SPX_HD Mk match_marker(const Rec &r, const AlnState &st, const uint8_t *qual, int pos)
{
    Mk m;
    const bool rev = (r.flag & SPX_FREVERSE) != 0;
    m.is_match = 1; m.row = -1; m.ref_pos = -1; m.pad0 = m.pad1 = 0;
    m.base_idx = rev ? r.l_qseq + st.rclip - pos - 1 : pos - st.lclip;
    m.q = (m.base_idx >= 0 && m.base_idx < r.l_qseq) ? qual[m.base_idx] : 0;
    return m;
}

/* Marker columns of a group, in three steps so that the middle one can run one alignment per thread:
 * group_merge   k-way merge of the alignments' mismatch lists by read position, positions where every alignment
 *               mismatches dropped (remove_all_mismatch_markers); returns the number of columns
 * aln_fill      one alignment's cell of every column: its mismatch, or a match marker (sort_and_fill_markers)
 * aln_filter    one alignment's walk over its ops and the columns (filter_ins_markers): columns inside an insertion /
 *               clip are marked for removal, the reference positions of its match markers inside '=' ops filled in
 * aln_compact / group_compact remove the marked columns (cells per alignment, positions per group; returns the number kept) */
SPX_HD int group_merge(const GroupView &G, const Pools &P, int32_t *pos, uint8_t *keep)
{
    /* positions only: the cells of a column (16 bytes per alignment) are written by aln_fill, one alignment per thread */
    const int n = G.n;
    int head[10], left[10], hpos[10]; /* hpos: read position of each list's head (0x7fffffff: exhausted) */
    for (int i = 0; i < n; ++i) {
        const bool rev = (G.rec[i].flag & SPX_FREVERSE) != 0;
        left[i] = G.st[i].n_mm;
        head[i] = rev ? G.st[i].n_mm - 1 : 0;
        hpos[i] = left[i] > 0 ? P.mm[G.st[i].mm_off + head[i]].pos : 0x7fffffff;
    }
    int ncol = 0;
    for (;;) {
        int best = 0x7fffffff, cnt = 0;
        for (int i = 0; i < n; ++i)
            if (hpos[i] < best) best = hpos[i];
        if (best == 0x7fffffff) break;
        for (int i = 0; i < n; ++i) {
            if (hpos[i] != best) continue;
            ++cnt;
            const bool rev = (G.rec[i].flag & SPX_FREVERSE) != 0;
            head[i] += rev ? -1 : 1;
            left[i]--;
            hpos[i] = left[i] > 0 ? P.mm[G.st[i].mm_off + head[i]].pos : 0x7fffffff;
        }
        if (cnt != n) { pos[ncol] = best; keep[ncol] = 1; ++ncol; }
    }
    return ncol;
}

/* this alignment's cell of every column: its mismatch at that position, or a match marker.  Its mismatches at positions
 * that are no column (every alignment mismatches there) are stepped over. */
/* columns [c_lo, c_hi) only: what a lane of a wave that shares ONE heavy alignment does (round 5); the mismatch cursor starts where the
 * walk over all columns would stand on reaching column c_lo -- behind every mismatch in front of that column's position (binary search: the
 * list is ascending in read position in the order the cursor takes it) */
SPX_HD void aln_fill_range(const GroupView &G, int i, const Pools &P, const int32_t *pos, Mk *mk, int ncol, int c_lo, int c_hi)
{
    const int n = G.n;
    const Rec &r = G.rec[i];
    const AlnState &st = G.st[i];
    const bool rev = (r.flag & SPX_FREVERSE) != 0;
    const int step = rev ? -1 : 1;
    const uint8_t *qual = P.qual + r.qual_off;
    int left = st.n_mm, head = rev ? st.n_mm - 1 : 0;
    if (c_hi > ncol) c_hi = ncol;
    if (c_lo > 0 && c_lo < c_hi && left > 0) {
        const int p0 = pos[c_lo];
        int lo = 0, hi = st.n_mm; /* lo = mismatches, in cursor order, in front of p0 */
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (P.mm[st.mm_off + (rev ? st.n_mm - 1 - mid : mid)].pos < p0) lo = mid + 1; else hi = mid;
        }
        head = rev ? st.n_mm - 1 - lo : lo;
        left = st.n_mm - lo;
    }
    MM s;
    s.pos = 0x7fffffff; s.base_idx = 0; s.q = 0; s.ref_pos = 0;
    if (left > 0) s = P.mm[st.mm_off + head];
    /* four columns at a time (round 5): their positions, and the quality a MATCH marker of this alignment would carry at each (what
     * match_marker reads: one scattered byte per column, a cache miss each on a long read), are fetched together -- four loads in flight
     * instead of one per column on the critical path; a column that turns out to hold this alignment's mismatch does not use its byte */
    for (int c0 = c_lo; c0 < c_hi; c0 += 4) {
        int pp[4], bi[4];
        uint8_t qq[4];
        const int m4 = c_hi - c0 < 4 ? c_hi - c0 : 4;
        for (int k = 0; k < 4; ++k) pp[k] = k < m4 ? pos[c0 + k] : 0;
        for (int k = 0; k < 4; ++k) {
            bi[k] = rev ? r.l_qseq + st.rclip - pp[k] - 1 : pp[k] - st.lclip;
            qq[k] = (k < m4 && bi[k] >= 0 && bi[k] < r.l_qseq) ? qual[bi[k]] : 0;
        }
        for (int k = 0; k < 4; ++k) {
            if (k >= m4) break;
            const int p = pp[k];
            while (left > 0 && s.pos < p) {
                head += step;
                if (--left > 0) s = P.mm[st.mm_off + head];
            }
            Mk m;
            if (left > 0 && s.pos == p) {
                m.base_idx = s.base_idx; m.ref_pos = s.ref_pos; m.row = -1; m.q = (uint8_t)s.q; m.is_match = 0;
                m.pad0 = m.pad1 = 0;
                head += step;
                if (--left > 0) s = P.mm[st.mm_off + head];
            } else { /* = match_marker(r, st, qual, p) */
                m.is_match = 1; m.row = -1; m.ref_pos = -1; m.pad0 = m.pad1 = 0;
                m.base_idx = bi[k];
                m.q = qq[k];
            }
            mk[(int64_t)(c0 + k) * n + i] = m;
        }
    }
}

SPX_HD void aln_fill(const GroupView &G, int i, const Pools &P, const int32_t *pos, Mk *mk, int ncol) { aln_fill_range(G, i, P, pos, mk, ncol, 0, ncol); }

/* columns [c_lo, c_hi) only (see aln_fill_range).  The op cursor starts at, or one op in front of, the op whose read interval holds the
 * first column's position (binary search over the ops' interval starts, which are monotone along the table): the ops in between are
 * stepped over exactly as the walk over all columns steps over ops that hold no column. */
SPX_HD void aln_filter_range(const GroupView &G, int i, const Pools &P, const int32_t *pos, Mk *mk, uint8_t *keep, int ncol, int c_lo, int c_hi)
{
    /* positions inside an insertion / clip of any alignment are not comparable: the column goes */
    const int n = G.n;
    const Rec &r = G.rec[i];
    const AlnState &st = G.st[i];
    const bool rev = (r.flag & SPX_FREVERSE) != 0;
    const Op *ops = P.ops + st.ops_off;
    if (c_hi > ncol) c_hi = ncol;
    if (c_lo >= c_hi) return;
    int col = rev ? c_hi - 1 : c_lo;
    const int step = rev ? -1 : 1;
    int p = pos[col];
    int t_first = 1;
    if (!(c_lo == 0 && c_hi == ncol) && st.n_visit > 2) {
        /* forward strand: interval starts ascend with t -- the LAST op with rds <= p; reverse strand: they descend -- the FIRST such op */
        int lo = 1, hi = st.n_visit - 1;
        if (!rev) {
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ops[mid].rds <= p) lo = mid; else hi = mid - 1; }
        } else {
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (ops[mid].rds <= p) hi = mid; else lo = mid + 1; }
        }
        t_first = lo > 1 ? lo - 1 : 1;
    }
    /* (four ops fetched at a time: their loads are in flight together) */
    for (int t0 = t_first; t0 < st.n_visit && col >= c_lo && col < c_hi; t0 += 4) {
        Op quad[4];
        const int m4 = st.n_visit - t0 < 4 ? st.n_visit - t0 : 4;
        for (int k = 0; k < 4; ++k)
            if (k < m4) quad[k] = ops[t0 + k];
        for (int k = 0; k < 4; ++k) {
            if (k >= m4 || !(col >= c_lo && col < c_hi)) break;
            const Op o = quad[k];
            while (col >= c_lo && col < c_hi) {
                if (!(o.rds <= p && p <= o.rde)) break;
                /* (every alignment writes the same 0; a relaxed atomic store, so that the host plan's threads -- one per alignment,
                 * like the lanes of the kernel -- do it without a data race: ThreadSanitizer run of the CPU suite) */
                if (o.op == SPX_CINS || o.op == SPX_CSOFT_CLIP || o.op == SPX_CHARD_CLIP) __atomic_store_n(&keep[col], (uint8_t)0, __ATOMIC_RELAXED);
                if (o.op == SPX_CEQUAL) /* ptMarker.c:184-187: reference position of this alignment's marker */
                    mk[(int64_t)col * n + i].ref_pos = rev ? o.rfs + o.rde - p : o.rfs + p - o.rds;
                col += step;
                if (col >= c_lo && col < c_hi) p = pos[col];
            }
        }
    }
}
SPX_HD void aln_filter(const GroupView &G, int i, const Pools &P, const int32_t *pos, Mk *mk, uint8_t *keep, int ncol)
{
    aln_filter_range(G, i, P, pos, mk, keep, ncol, 0, ncol);
}

/* (the columns' marker cells are moved by aln_compact, one alignment per thread: a column of a 9-alignment group is
 * 144 bytes, and on the device one lane moving the 45 000 columns of a 100 kb group was most of the pass) */
SPX_HD int group_compact(const GroupView &G, int32_t *pos, const uint8_t *keep, int ncol)
{
    (void)G;
    int w = 0;
    for (int c = 0; c < ncol; ++c) {
        if (!keep[c]) continue;
        if (w != c) pos[w] = pos[c];
        ++w;
    }
    return w;
}
/* this alignment's cell of every kept column moves to the column's new place (cells of different alignments never
 * share an address, so the alignments of a group can do this side by side) */
SPX_HD void aln_compact(const GroupView &G, int i, Mk *mk, const uint8_t *keep, int ncol)
{
    const int n = G.n;
    int w = 0;
    for (int c = 0; c < ncol; ++c) {
        if (!keep[c]) continue;
        if (w != c) mk[(int64_t)w * n + i] = mk[(int64_t)c * n + i];
        ++w;
    }
}

/* ascending by key; inputs are monotone (ascending, or descending on the reverse strand), so: reverse when
 * descending, then an insertion pass that is linear on sorted data */
template <int KEY> /* 0: rds, 1: sqs */
SPX_HD void sort_blocks(Blk *b, int n)
{
    auto key = [](const Blk &x) { return KEY == 0 ? x.rds : x.sqs; };
    if (n > 1 && key(b[0]) > key(b[n - 1]))
        for (int i = 0, j = n - 1; i < j; ++i, --j) { const Blk t = b[i]; b[i] = b[j]; b[j] = t; }
    for (int i = 1; i < n; ++i) {
        const Blk t = b[i];
        int j = i - 1;
        while (j >= 0 && key(b[j]) > key(t)) { b[j + 1] = b[j]; --j; }
        b[j + 1] = t;
    }
}

/* flanking windows of one alignment around every marker cell (find_flanking_blocks).  The reference walks the
 * marker LIST, which holds every position n times (one cell per alignment): the repeats are replayed from a register. */
SPX_HD int flank_blocks(const AlnState &st, const int32_t *pos, int ncol, int n, int margin, Iv *out, int cap)
{
    int cnt = 0;
    int start = 0, end = 0;
    for (int col = 0; col < ncol; ++col) {
        const int p = pos[col];
        const int v0 = p - margin, v1 = p + margin;
        const int cs = st.rds > v0 ? st.rds : v0, ce = st.rde < v1 ? st.rde : v1;
        /* after the first cell of a position the window is [.., ce]; its n - 1 repeats take the "cs < end" branch and set
         * end = ce again -- unless the window is empty (cs >= ce), where every repeat opens a new block */
        const int reps = cs < ce ? 1 : n;
        for (int rep = 0; rep < reps; ++rep) {
            if (col == 0 && rep == 0) { start = cs; end = ce; continue; }
            if (cs < end) end = ce;
            else {
                if (cnt >= cap) return -1;
                Iv b = {start, end};
                out[cnt++] = b;
                start = cs; end = ce;
            }
        }
    }
    if (cnt >= cap) return -1;
    Iv b = {start, end};
    out[cnt++] = b;
    return cnt;
}

/* The windows of the consensus rounds WITHOUT clamping (blocks_rounds' shared case: margin > 0, every position inside every alignment's
 * extent) depend on the positions through their gaps only: with cs = p - margin, ce = p + margin and cs < ce the walk above opens a new
 * window at column c iff pos[c] - pos[c-1] >= 2 * margin, and a window is {first position - margin, last position + margin}.  The margins of
 * the rounds are a fixed sequence (margin <- (int)(margin * 0.8), from the flank margin), so every column has ONE round from which on it
 * starts a window: break_round[c] = the first round r with 2 * margin_r <= gap(c) = the number of rounds whose doubled margin exceeds the
 * gap, kept in one byte per column (the `keep` flags of the column filter are dead by then).  A round then finds its windows by scanning
 * the bytes eight at a time instead of walking the positions: a 100 kb read's group has ~45 000 columns and ~12 rounds, and this walk was
 * the bulk of the group pass (3.3 M of its 3.6 M loop steps per 1 024 mixed groups).  The table covers the first kTableRounds rounds (the
 * doubled margins sit in registers: one pass of compares, no look-ups); later rounds and groups of few columns take the walk. */
constexpr int kNoBreak = 127;     /* break_round of column 0 (it starts the first window by itself) */
constexpr int kTableRounds = 16;      /* at most (Params.table_rounds; SPX_WINDOW_TABLE=min_cols,rounds on the host, for CPU tests) */
constexpr int kTableMinCols = 1024;   /* default of Params.table_min_cols */
SPX_HD void flank_break_rounds(const int32_t *pos, int ncol, int flank_margin, int rounds, uint8_t *br)
{
    uint32_t thr[kTableRounds]; /* 2 * margin of round r (0 once the margin is used up: such rounds do not take this path) */
    {
        int m = flank_margin;
#pragma unroll
        for (int r = 0; r < kTableRounds; ++r) {
            m = (int)(m * 0.8);
            thr[r] = (m > 0 && r < rounds) ? 2u * (uint32_t)m : 0u;
        }
    }
    if (ncol > 0) br[0] = (uint8_t)kNoBreak;
    int prev = ncol > 0 ? pos[0] : 0;
    for (int c = 1; c < ncol; ++c) {
        const int p = pos[c];
        const int gap = p - prev;
        prev = p;
        const uint32_t g = gap < 0 ? 0u : (uint32_t)gap;
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < kTableRounds; ++r) cnt += thr[r] > g ? 1 : 0; /* thr is non-increasing: the count is the first r with thr[r] <= g */
        br[c] = (uint8_t)cnt; /* = the number of table rounds: no window starts here in a round of the table */
    }
}
/* the windows of round `round` (margin = its margin, > 0): same list, same overflow behaviour as flank_blocks() over unclamped extents */
SPX_HD int flank_blocks_by_round(const int32_t *pos, const uint8_t *br, int ncol, int round, int margin, Iv *out, int cap)
{
    int cnt = 0, first = 0;
    auto open_at = [&](int c) -> bool { /* the window [first, c) ends; the next one starts at column c */
        if (cnt >= cap) return false;
        Iv b = {pos[first] - margin, pos[c - 1] + margin};
        out[cnt++] = b;
        first = c;
        return true;
    };
    int c = 1;
    for (; c < ncol && (((uintptr_t)(br + c)) & 7u); ++c)
        if (br[c] <= round && !open_at(c)) return -1;
    const uint64_t ones = 0x0101010101010101ull, high = 0x8080808080808080ull;
    for (; c + 8 <= ncol; c += 8) {
        uint64_t w;
        __builtin_memcpy(&w, __builtin_assume_aligned(br + c, 8), 8);
        /* every byte is <= 127: (b | 0x80) - (round + 1) borrows from no neighbour and keeps its top bit iff b > round */
        uint64_t hit = ~((w | high) - (uint64_t)(round + 1) * ones) & high;
        while (hit) {
            const int k = __builtin_ctzll(hit) >> 3;
            if (!open_at(c + k)) return -1;
            hit &= hit - 1;
        }
    }
    for (; c < ncol; ++c)
        if (br[c] <= round && !open_at(c)) return -1;
    if (cnt >= cap) return -1;
    Iv b = {pos[first] - margin, pos[ncol - 1] + margin};
    out[cnt++] = b;
    return cnt;
}

/* ascending by start; the lists are ascending already except in degenerate cases */
SPX_HD void sort_intervals(Iv *b, int n)
{
    for (int i = 1; i < n; ++i) {
        const Iv t = b[i];
        int j = i - 1;
        if (b[j].s <= t.s) continue;
        while (j >= 0 && b[j].s > t.s) { b[j + 1] = b[j]; --j; }
        b[j + 1] = t;
    }
}

/* intersect_by_rd_f (ptMarker.c:398-435): strict '<' overlap test, windows that merely touch do not intersect.
 * GetY(j) yields interval j of the second list (an Iv list, or the read coordinates of a block list). */
template <class GetY>
SPX_HD int intersect(const Iv *x, int nx, GetY gety, int ny, Iv *out, int cap, bool *all_positive = nullptr)
{
    if (all_positive) *all_positive = true;
    if (nx == 0 || ny == 0) return 0;
    int cnt = 0, j = 0;
    Iv y = gety(0);
    for (int i = 0; i < nx; ++i) {
        const Iv xi = x[i];
        while (j < ny && y.e < xi.s) { ++j; if (j < ny) y = gety(j); }
        while (j < ny && y.s < xi.e) {
            if (cnt >= cap) return -1;
            Iv b = {xi.s > y.s ? xi.s : y.s, xi.e < y.e ? xi.e : y.e};
            out[cnt++] = b;
            if (all_positive && !(b.s < b.e)) *all_positive = false;
            if (y.e <= xi.e) { ++j; if (j < ny) y = gety(j); } else break;
        }
    }
    return cnt;
}

/* project the consensus intervals cur[0..nb) (read coordinates) onto one alignment (correct_conf_blocks) */
SPX_HD int project_blocks(const Rec &r, const AlnState &st, const Op *ops, const Iv *cur, int nb, int thr, Blk *out, int cap)
{
    const bool rev = (r.flag & SPX_FREVERSE) != 0;
    int cnt = 0;
    int j = rev ? nb - 1 : 0;
    bool have = true, del_flag = false;
    int bs = rev ? -cur[j].e : cur[j].s, be = rev ? -cur[j].s : cur[j].e;
    int rfs = -1, rfe = -1, sqs = -1, sqe = -1;
    for (int t = 1; t < st.n_visit; ++t) {
        const Op o = ops[t];
        const int cs = rev ? -o.rde : o.rds, ce = rev ? -o.rds : o.rde;
        if (mx(o.op) || o.op == SPX_CINS) {
            const bool ins = o.op == SPX_CINS;
            while (have && be <= ce) {
                if (cs <= bs && !(del_flag && cs == bs)) {
                    rfs = ins ? o.rfs : o.rfs + (bs - cs);
                    sqs = o.sqs + (bs - cs);
                }
                rfe = ins ? o.rfe : o.rfs + (be - cs);
                sqe = o.sqs + (be - cs);
                if (cnt >= cap) return -1;
                Blk b = {rfs, rfe, sqs, sqe, cur[j].s, cur[j].e};
                out[cnt++] = b;
                if (rev && j > 0) { --j; bs = -cur[j].e; be = -cur[j].s; }
                else if (!rev && j < nb - 1) { ++j; bs = cur[j].s; be = cur[j].e; }
                else have = false;
            }
            if (!have) break;
            if (cs <= bs && bs <= ce && !(del_flag && cs == bs)) {
                rfs = ins ? o.rfs : o.rfs + (bs - cs);
                sqs = o.sqs + (bs - cs);
            }
            del_flag = false;
        } else if (o.op == SPX_CDEL) {
            if (have && bs == cs && o.len <= thr) {
                del_flag = true;
                rfs = o.rfs;
                sqs = o.sqs;
            }
        }
    }
    sort_blocks<1>(out, cnt);
    return cnt;
}

/* the consensus rounds of a group can be interrupted where the windows have to be projected onto the alignments (a
 * walk over every alignment's ops: one thread per alignment does that on the device) and resumed afterwards */
struct BlocksState {
    int32_t margin, iter, nblk, same, np, nc, too_long;
    int32_t phase;      /* 0: rounds running, 1: projections of cur[0..nc) wanted, 2: finished */
    int32_t pi, ci, ni; /* which of the three interval buffers is prev / cur / nxt */
    int32_t result;     /* at phase 2: 1 scored, 0 not scored, < 0 SPX_E* */
    int32_t ncol, pad[3];
};

/* the scratch arrays of one group */
struct GroupScratch {
    int32_t *pos;
    Mk *mk;
    uint8_t *keep;
    Iv *flank, *cur, *nxt; /* interval lists of the consensus rounds: read coordinates only */
    Blk *proj;       /* [n][blk_cap]: each alignment's current block list (confident blocks, then projected windows) */
    int32_t *nproj;  /* [n] */
    struct BlocksState *bstate;
    int32_t *rows_mk;
    int32_t blk_cap, rows_cap;
};
SPX_HD GroupScratch group_scratch(const GroupArena &A, char *base)
{
    GroupScratch S;
    S.pos = (int32_t *)(base + A.o_pos);
    S.mk = (Mk *)(base + A.o_mk);
    S.keep = (uint8_t *)(base + A.o_keep);
    S.flank = (Iv *)(base + A.o_flank);
    S.cur = (Iv *)(base + A.o_cur);
    S.nxt = (Iv *)(base + A.o_nxt);
    S.proj = (Blk *)(base + A.o_proj);
    S.nproj = (int32_t *)(base + A.o_nproj);
    S.bstate = (struct BlocksState *)(base + A.o_state);
    S.rows_mk = (int32_t *)(base + A.o_rowsmk);
    S.blk_cap = A.blk_cap;
    S.rows_cap = A.rows_cap;
    return S;
}

SPX_HD Iv *blocks_buf(const GroupScratch &S, int k) { return k == 0 ? S.flank : k == 1 ? S.cur : S.nxt; }

/* rounds of the consensus loop (secphase.c:162-170 + ptMarker.c:398-667) until the windows must be projected onto the
 * alignments (phase 1) or the loop ends (phase 2) */
SPX_HD void blocks_rounds(const GroupView &G, const Pools &P, const Params &par, GroupScratch &S, BlocksState &B)
{
    const int n = G.n, cap = S.blk_cap, ncol = B.ncol;
    Iv *prev = blocks_buf(S, B.pi), *cur = blocks_buf(S, B.ci), *nxt = blocks_buf(S, B.ni);
    int pi = B.pi, ci = B.ci, ni = B.ni;
    auto swap_cn = [&]() { Iv *t = cur; cur = nxt; nxt = t; const int q = ci; ci = ni; ni = q; };
    auto fail = [&](int code) { B.phase = 2; B.result = code; };
    while (par.consensus && B.too_long) {
        B.margin = (int)(B.margin * 0.8);
        const int margin = B.margin;
        const int round = B.pad[0]++; /* margin = the margin of round `round` of flank_break_rounds() */
        /* intersect every alignment's blocks, then every alignment's flanking windows, in read coordinates.  Once a
         * list has been intersected with one copy of another list, every interval lies inside ONE interval of that
         * list; intersecting it with the same list again changes nothing as long as no interval is empty (strict '<'
         * drops those) -- such repeats are skipped. */
        int nc;
        bool positive = false;
        if (B.same) {
            nc = B.np;
            for (int k = 0; k < nc; ++k) cur[k] = prev[k];
            for (int i = 1; i < n; ++i) {
                if (i > 1 && positive) continue;
                const Iv *pv = prev;
                const int m = intersect(cur, nc, [&](int j) { return pv[j]; }, B.np, nxt, cap, &positive);
                if (m < 0) return fail(SPX_ENOMEM);
                nc = m;
                swap_cn();
            }
        } else {
            sort_blocks<0>(S.proj, S.nproj[0]);
            nc = S.nproj[0];
            for (int k = 0; k < nc; ++k) { Iv v = {S.proj[k].rds, S.proj[k].rde}; cur[k] = v; }
            for (int i = 1; i < n; ++i) {
                Blk *bi = S.proj + (int64_t)i * cap;
                sort_blocks<0>(bi, S.nproj[i]);
                const int m = intersect(cur, nc, [&](int j) { Iv v = {bi[j].rds, bi[j].rde}; return v; }, S.nproj[i], nxt, cap);
                if (m < 0) return fail(SPX_ENOMEM);
                nc = m;
                swap_cn();
            }
        }
        /* prev is free now: it takes the flanking windows.  With a positive margin and every marker position inside
         * the aligned extent [rds, rde] of every alignment, the merge decisions (cs < end) do not depend on the extent:
         * clamping only moves the start of the first window and the end of the last one.  The windows are then built
         * ONCE without clamping and every alignment reads them through its own clamp. */
        Iv *fl = prev;
        bool shared = margin > 0;
        for (int i = 0; i < n && shared; ++i)
            shared = G.st[i].rde > G.st[i].rds && S.pos[0] >= G.st[i].rds && S.pos[ncol - 1] <= G.st[i].rde;
        int nu = 0;
        if (shared) {
            const int table_rounds = par.table_rounds < kTableRounds ? par.table_rounds : kTableRounds;
            if (round < table_rounds && ncol >= par.table_min_cols) {
                if (!B.pad[1]) { flank_break_rounds(S.pos, ncol, par.flank_margin, table_rounds, S.keep); B.pad[1] = 1; } /* once per group, at its first shared round */
                nu = flank_blocks_by_round(S.pos, S.keep, ncol, round, margin, fl, cap);
            } else {
                AlnState wide = G.st[0];
                wide.rds = -0x3fffffff; wide.rde = 0x3fffffff;
                nu = flank_blocks(wide, S.pos, ncol, n, margin, fl, cap);
            }
            if (nu < 0) return fail(SPX_ENOMEM);
        }
        positive = false;
        for (int i = 0; i < n; ++i) {
            int nf, m;
            if (shared) {
                const int lo = G.st[i].rds, hi = G.st[i].rde, lastj = nu - 1;
                /* after the first alignment: same windows, another clamp -- nothing to do when the clamp cuts nothing */
                if (i > 0 && positive && nc > 0 && lo <= cur[0].s && hi >= cur[nc - 1].e) continue;
                nf = nu;
                const Iv *fv = fl;
                m = intersect(cur, nc, [&](int j) {
                    Iv v = fv[j];
                    if (j == 0 && v.s < lo) v.s = lo;
                    if (j == lastj && v.e > hi) v.e = hi;
                    return v;
                }, nf, nxt, cap, &positive);
            } else {
                nf = flank_blocks(G.st[i], S.pos, ncol, n, margin, fl, cap);
                if (nf < 0) return fail(SPX_ENOMEM);
                sort_intervals(fl, nf);
                const Iv *fv = fl;
                m = intersect(cur, nc, [&](int j) { return fv[j]; }, nf, nxt, cap);
            }
            if (m < 0) return fail(SPX_ENOMEM);
            nc = m;
            swap_cn();
        }
        if (nc == 0) {
            for (int i = 0; i < n; ++i) S.nproj[i] = 0;
            B.nblk = 0;
            break;
        }
        ++B.iter;
        /* A window longer than 1000 bases in READ coordinates is longer than 1000 in SEQ coordinates on every
         * alignment (inside an alignment SEQ index and read position move together), so the loop is certain to go
         * round again and this round's projection onto the alignments -- a walk over all their ops -- would only
         * feed the next round's intersection, which reads nothing but the read coordinates: every alignment gets
         * every window (windows lie inside confident blocks of all alignments), so each list is the window list
         * itself.  The last permitted round always projects for real. */
        bool long_window = false;
        for (int k = 0; k < nc; ++k)
            if (cur[k].e - cur[k].s > 1000) long_window = true;
        B.nblk = nc;
        if (long_window && B.iter < 64) {
            B.same = 1;
            { Iv *t = prev; prev = cur; cur = t; const int q = pi; pi = ci; ci = q; } /* this round's windows become `prev` */
            B.np = nc;
            B.too_long = 1;
        } else {
            B.same = 0;
            B.nc = nc;
            B.pi = pi; B.ci = ci; B.ni = ni;
            B.phase = 1; /* project cur[0..nc) onto every alignment, then blocks_after_projection */
            return;
        }
    }
    B.pi = pi; B.ci = ci; B.ni = ni;
    B.phase = 2;
    B.result = (B.nblk > 0 || !par.consensus) ? 1 : 0;
}

SPX_HD void blocks_begin(const GroupView &G, const Pools &P, const Params &par, GroupScratch &S, int ncol, BlocksState &B)
{
    const int n = G.n, cap = S.blk_cap;
    B.margin = par.flank_margin; B.iter = 0; B.nblk = 1 /* DESIGN.md U1 */; B.same = 0; B.np = 0; B.nc = 0;
    B.phase = 0; B.pi = 0; B.ci = 1; B.ni = 2; B.result = 0; B.ncol = ncol; B.pad[0] = B.pad[1] = B.pad[2] = 0;
    bool too_long = false; /* needs_to_find_blocks: a block longer than 1000 (SEQ or reference), or an alignment without blocks */
    for (int i = 0; i < n; ++i) {
        const AlnState &st = G.st[i];
        if (st.n_conf > cap) { B.phase = 2; B.result = SPX_ENOMEM; return; }
        Blk *dst = S.proj + (int64_t)i * cap;
        for (int k = 0; k < st.n_conf; ++k) {
            const Blk b = P.conf[st.conf_off + k];
            dst[k] = b;
            if ((b.sqe - b.sqs) > 1000 || (b.rfe - b.rfs) > 1000) too_long = true;
        }
        S.nproj[i] = st.n_conf;
        if (st.n_conf == 0) too_long = true;
    }
    B.too_long = too_long;
    blocks_rounds(G, P, par, S, B);
}

/* phase 1, one alignment: its blocks for the windows cur[0..nc) */
SPX_HD void blocks_project(const GroupView &G, int i, const Pools &P, const Params &par, GroupScratch &S, const BlocksState &B)
{
    if (B.phase != 1) return;
    const int cap = S.blk_cap;
    Blk *dst = S.proj + (int64_t)i * cap;
    const int m = project_blocks(G.rec[i], G.st[i], P.ops + G.st[i].ops_off, blocks_buf(S, B.ci), B.nc, par.indel_threshold, dst, cap);
    S.nproj[i] = m;
    bool tl = m <= 0;
    for (int k = 0; k < m; ++k)
        if ((dst[k].sqe - dst[k].sqs) > 1000 || (dst[k].rfe - dst[k].rfs) > 1000) tl = true;
    S.nproj[16 + i] = tl ? 1 : 0;
}

/* phase 1 -> after every alignment has its blocks: goes on with the rounds or ends */
SPX_HD void blocks_after_projection(const GroupView &G, const Pools &P, const Params &par, GroupScratch &S, BlocksState &B)
{
    if (B.phase != 1) return;
    bool too_long = false;
    for (int i = 0; i < G.n; ++i) {
        if (S.nproj[i] < 0) { B.phase = 2; B.result = SPX_ENOMEM; return; }
        if (S.nproj[16 + i]) too_long = true;
    }
    B.too_long = too_long;
    B.phase = 0;
    if (B.iter >= 64) { B.phase = 2; B.result = (B.nblk > 0 || !par.consensus) ? 1 : 0; return; }
    blocks_rounds(G, P, par, S, B);
}

/* the whole loop in one go (host plan; on the device the tail of groups that need more projection rounds than the
 * kernel sequence provides).  On return S.proj / S.nproj hold, per alignment, the blocks plan_baq walks.  Returns 1 if
 * the group is scored, 0 if not, < 0 on SPX_E*. */
SPX_HD int blocks_finish(const GroupView &G, const Pools &P, const Params &par, GroupScratch &S, BlocksState &B)
{
    while (B.phase == 1) {
        for (int i = 0; i < G.n; ++i) blocks_project(G, i, P, par, S, B);
        blocks_after_projection(G, P, par, S, B);
    }
    return B.result;
}
SPX_HD int group_blocks(const GroupView &G, const Pools &P, const Params &par, GroupScratch &S, int ncol)
{
    BlocksState B;
    blocks_begin(G, P, par, S, ncol, B);
    return blocks_finish(G, P, par, S, B);
}

struct RowRec;
/* where the emitting pass writes */
struct PlanOut {
    /* per problem */
    int64_t *ref_nib, *qry_nib;
    int32_t *ref_tid, *ref_rfs; /* may be NULL */
    int32_t *L, *R, *bw, *row_off, *n_rows, *prob_slots;
    uint8_t *has_n; /* window or query holds a base other than ACGT (selects the general emission path) */
    int64_t *s_off, *fsave_off;
    /* per wanted row */
    int32_t *rows, *row_expect, *row_prob;
    uint8_t *row_rawq;
    /* device: the emitting pass writes the four row fields as ONE 16-byte record per row (a lane walks its alignment and
     * writes its rows one after the other: one open cache line per lane instead of four; rows_unpack_kernel then spreads
     * the records over the arrays above, coalesced).  NULL: the arrays are written directly (host plan). */
    struct RowRec *rr;
    /* quality edits (all_rows) */
    int32_t *qe_rec, *qe_pos, *qe_len, *qe_row0, *qe_batch;
};
struct RowRec { int32_t row, expect, prob, rawq; };
struct PlanBase { /* this group's first problem / row / edit and scratch offsets */
    int64_t prob, row, qe, s_off, f_off;
};

/* BAQ windows of one alignment (calc_local_baq's control flow).  EMIT = false: counts into gc; EMIT = true: writes
 * the problems, the wanted rows and the marker updates.  Columns are visited through this alignment's cell only:
 * the reference's loops step over every marker and skip the other alignments' cells. */
/* Round 5: blocks [bi_lo, bi_hi) only (plan_baq_range): the blocks of an alignment are independent once the two cursors and the output
 * offsets at the first block of a range are known, so the lanes of a wave that shares ONE heavy alignment take contiguous ranges of its
 * blocks.  The cursors come from binary searches -- the column cursor stands on the first column (in the order it walks them) whose base
 * index reaches the block's start, the op cursor on, or one op in front of, the first op that reaches the block's start in both
 * coordinates; both tables are monotone along the walk -- and the offsets from the counting pass's per-range totals.  Allowed only when
 * plan_can_split() holds: no op of length 0 in front of the table's end (the walk pauses on such an op -- `adv() == 0` -- and a search would
 * jump over it) and not in the all-rows mode (its row scratch is per alignment). */
SPX_HD bool plan_can_split(const Params &par, const AlnState &st) { return !par.all_rows && st.n_visit == st.n_ops; }

template <bool EMIT>
SPX_HD int plan_baq_range(const GroupView &G, int ai, const Pools &P, const RefView &rv, const Params &par, GroupScratch &S, int ncol,
                          GroupCount &gc, PlanBase &at, const PlanOut &out, int bi_lo, int bi_hi)
{
    const int n = G.n;
    const Rec &r = G.rec[ai];
    const AlnState &st = G.st[ai];
    const Op *ops = P.ops + st.ops_off;
    const uint8_t *qual = P.qual + r.qual_off;
    const bool rev = (r.flag & SPX_FREVERSE) != 0;
    const int step = rev ? -1 : 1;
    int c = rev ? ncol - 1 : 0;
    int ci = 0;
    const int margin = 10;
    const int last = st.n_ops - 1;
    /* the op and the marker under the two cursors live in registers: every re-read would be a dependent global load */
    Op co = ops[0];
    auto adv = [&]() -> int { if (ci < last) { ++ci; co = ops[ci]; return co.len; } return 0; };
    auto own = [&](int col) -> Mk & { return S.mk[(int64_t)col * n + ai]; };
    int cb = (c >= 0 && c < ncol) ? own(c).base_idx : 0; /* base_idx of this alignment's marker in column c */
    auto step_c = [&]() { c += step; if (c >= 0 && c < ncol) cb = own(c).base_idx; };
    auto zero_edit = [&](int base) {
        if (!par.all_rows) return;
        if (EMIT) {
            out.qe_rec[at.qe] = r.rec; out.qe_pos[at.qe] = base; out.qe_len[at.qe] = 0; out.qe_row0[at.qe] = 0;
            out.qe_batch[at.qe] = r.batch;
        }
        at.qe++;
        if (!EMIT) gc.n_qe++;
    };
    /* row fields: one record per row on the device, four arrays on the host */
    auto row_get = [&](int64_t i) -> int { return out.rr ? out.rr[i].row : out.rows[i]; };
    auto row_set = [&](int64_t i, int v) { if (out.rr) out.rr[i].row = v; else out.rows[i] = v; };
    auto exp_get = [&](int64_t i) -> int { return out.rr ? out.rr[i].expect : out.row_expect[i]; };
    auto exp_set = [&](int64_t i, int v) { if (out.rr) out.rr[i].expect = v; else out.row_expect[i] = v; };
    const Blk *blocks = S.proj + (int64_t)ai * S.blk_cap;
    const int nblocks = S.nproj[ai];
    int32_t *rows_mk = S.rows_mk + (int64_t)ai * S.rows_cap; /* (all-rows mode only) */
    if (bi_hi > nblocks) bi_hi = nblocks;
    if (bi_lo > 0 && bi_lo < bi_hi) {
        const Blk b0 = blocks[bi_lo];
        int lo = 0, hi = ncol; /* lo = columns, in walking order, whose base index lies in front of the block */
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (own(rev ? ncol - 1 - mid : mid).base_idx < b0.sqs) lo = mid + 1; else hi = mid;
        }
        c = rev ? ncol - 1 - lo : lo;
        cb = (c >= 0 && c < ncol) ? own(c).base_idx : 0;
        lo = 0; hi = last; /* first op that reaches the block's start in SEQ and in reference coordinates */
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ops[mid].sqe >= b0.sqs && ops[mid].rfe >= b0.rfs) hi = mid; else lo = mid + 1;
        }
        ci = lo > 0 ? lo - 1 : 0;
        co = ops[ci];
    }
    for (int bi = bi_lo; bi < bi_hi; ++bi) {
        const Blk b = blocks[bi];
        while (co.sqe < b.sqs || co.rfe < b.rfs)
            if (adv() == 0) break;
        /* markers of this alignment in the leading margin lose their quality */
        while (c >= 0 && c < ncol && cb < b.sqs + margin) {
            if (b.sqs <= cb) {
                if (EMIT) { own(c).q = 0; own(c).row = -1; }
                zero_edit(cb);
            }
            step_c();
        }
        if (c >= 0 && c < ncol && cb <= b.sqe - margin && b.sqs + margin <= cb) {
            const int L = b.sqe - b.sqs + 1, R = b.rfe - b.rfs + 1;
            if (L <= 0 || R <= 0) return SPX_EINVAL;
            if (r.tid < 0 || r.tid >= rv.n_contigs || b.rfs < 0 || b.rfe >= rv.len[r.tid]) return SPX_EINVAL;
            if (b.sqs < 0 || b.sqe >= r.l_qseq) return SPX_EINVAL;
            const int diff = R > L ? R - L : L - R;
            const int bw_in = (int)(diff + par.conf_b);
            const int bw = effective_bw(R, L, bw_in);
            const int cls = band_class(2 * bw + 1);
            if (cls < 0) return SPX_EUNSUPPORTED;
            const int slots = class_slots(cls);
            /* wanted rows: this alignment's markers in [sqs+margin, sqe-margin) keep a BAQ value; with all_rows every
             * base of that range is wanted (the write-back at ptMarker.c:786) */
            int nrows = 0;
            if (par.all_rows) {
                nrows = L - 2 * margin > 0 ? L - 2 * margin : 0;
                if (nrows > S.rows_cap) return SPX_ENOMEM;
                if (EMIT) for (int t = 0; t < nrows; ++t) rows_mk[t] = -1;
            }
            {
                int kb = cb;
                for (int k = c; k >= 0 && k < ncol;) {
                    if (kb > b.sqe) break;
                    const int t = kb - b.sqs;
                    if (t >= margin && t < L - margin) {
                        if (par.all_rows) { if (EMIT) rows_mk[t - margin] = k; }
                        else { /* (the wanted rows are the consecutive columns c, c + step, ..: their columns need no table) */
                            if (nrows >= S.rows_cap) return SPX_ENOMEM;
                            if (EMIT) row_set(at.row + nrows, t + 1);
                            ++nrows;
                        }
                    }
                    k += step;
                    if (k >= 0 && k < ncol) kb = own(k).base_idx;
                }
            }
            const int64_t row0 = at.row;
            if (EMIT) {
                for (int w = 0; w < nrows; ++w) {
                    const int t = par.all_rows ? margin + w : row_get(row0 + w) - 1;
                    if (out.rr) {
                        RowRec rec = {t + 1, -1, (int32_t)at.prob, (int32_t)qual[b.sqs + t]};
                        out.rr[row0 + w] = rec;
                    } else {
                        out.rows[row0 + w] = t + 1;
                        out.row_expect[row0 + w] = -1;
                        out.row_rawq[row0 + w] = qual[b.sqs + t];
                        out.row_prob[row0 + w] = (int32_t)at.prob;
                    }
                }
            }
            /* expected reference index of every wanted base, from the CIGAR walk of the write-back loop; the wanted
             * rows ascend and the M/=/X pieces of a block are disjoint and ascending, so one cursor serves */
            int w = 0;
            while (co.sqs <= b.sqe || co.rfs <= b.rfe) {
                const Op o = co;
                int x = o.rfs - b.rfs, y = o.sqs - b.sqs;
                if (x < 0) x = 0;
                if (y < 0) y = 0;
                if (EMIT && mx(o.op)) {
                    const int e1 = o.sqe < b.sqe ? o.sqe : b.sqe, s1 = o.sqs > b.sqs ? o.sqs : b.sqs;
                    int len = e1 - s1 + 1;
                    if (o.len < len) len = o.len;
                    while (w < nrows && row_get(row0 + w) - 1 < y) ++w;
                    while (w < nrows && row_get(row0 + w) - 1 < y + len) {
                        exp_set(row0 + w, x + (row_get(row0 + w) - 1 - y));
                        ++w;
                    }
                }
                if (o.sqe <= b.sqe || o.rfe <= b.rfe) { if (adv() == 0) break; }
                else break;
            }
            if (par.all_rows) {
                if (EMIT) {
                    out.qe_rec[at.qe] = r.rec; out.qe_pos[at.qe] = b.sqs + margin; out.qe_len[at.qe] = nrows;
                    out.qe_row0[at.qe] = (int32_t)row0; out.qe_batch[at.qe] = r.batch;
                }
                at.qe++;
                if (!EMIT) gc.n_qe++;
            }
            if (EMIT) {
                for (int w2 = 0; w2 < nrows; ++w2) {
                    const int k = par.all_rows ? rows_mk[w2] : c + w2 * step;
                    if (k < 0) continue;
                    Mk &m = own(k);
                    if (exp_get(row0 + w2) >= 0) m.row = (int32_t)(row0 + w2);
                    else { m.row = -1; m.q = (uint8_t)(par.set_q < 94 ? par.set_q : 93); } /* base not under an M op: keeps set_q */
                }
                const int64_t p = at.prob;
                out.ref_nib[p] = rv.nib_off[r.tid] + b.rfs;
                if (out.ref_tid) { out.ref_tid[p] = r.tid; out.ref_rfs[p] = b.rfs; }
                out.qry_nib[p] = (P.code_lead_bytes + r.seq_off) * 2 + b.sqs;
                out.L[p] = L; out.R[p] = R; out.bw[p] = bw;
                out.row_off[p] = (int32_t)row0;
                out.n_rows[p] = nrows;
                out.prob_slots[p] = slots;
                out.s_off[p] = at.s_off + 8;
                out.fsave_off[p] = at.f_off;
                bool has_n = window_has_n(rv, r.tid, b.rfs, R);
                if (!has_n && st.has_n) {
                    const uint8_t *code = P.code4 + P.code_lead_bytes + r.seq_off;
                    for (int k = 0; k < L && !has_n; ++k) {
                        const int q = b.sqs + k;
                        has_n = ((code[q >> 1] >> ((q & 1) << 2)) & 0xf) > 3;
                    }
                }
                out.has_n[p] = has_n ? 1 : 0;
            }
            const int64_t cells = band_cells(L, R, bw);
            if (!EMIT) {
                gc.n_prob++;
                gc.n_rows += nrows;
                gc.cells += cells;
                gc.cls_prob[cls]++;
                gc.cls_cells[cls] += cells;
                gc.s_need += 8 + ((L + 2 + 7) & ~7);
                gc.f_need += (int64_t)nrows * (par.row_mult > 2 ? par.row_mult : 2) * slots;
            }
            at.prob++;
            at.row += nrows;
            /* whole 64-byte lines behind a lead pad of one line: the one-lane forward kernel writes 1/s[] eight rows at a time */
            at.s_off += 8 + ((L + 2 + 7) & ~7);
            at.f_off += (int64_t)nrows * (par.row_mult > 2 ? par.row_mult : 2) * slots;
        }
        /* markers in the trailing margin lose their quality */
        while (c >= 0 && c < ncol && cb <= b.sqe) {
            if (b.sqe - margin <= cb) {
                if (EMIT) { own(c).q = 0; own(c).row = -1; }
                zero_edit(cb);
            }
            step_c();
        }
    }
    return 0;
}

template <bool EMIT>
SPX_HD int plan_baq(const GroupView &G, int ai, const Pools &P, const RefView &rv, const Params &par, GroupScratch &S, int ncol,
                    GroupCount &gc, PlanBase &at, const PlanOut &out)
{
    return plan_baq_range<EMIT>(G, ai, P, rv, par, S, ncol, gc, at, out, 0, 0x7fffffff);
}

/* the HMM constants of problem p (one thread per problem on the device: coalesced, unlike the group passes) */
SPX_HD void problem_constants(const Params &par, int L, int R, uint8_t has_n, double *h)
{
    hmm_constants(R, L, par.d, par.e, par.qf, h);
    h[SPX_H_PAD0] = has_n ? 1.0 : 0.0;
    const int diff = R > L ? R - L : L - R;
    h[SPX_H_TDROP] = terminal_drop(par.term_guard, L, R, effective_bw(R, L, (int)(diff + par.conf_b))) ? 1.0 : 0.0; /* bw as in plan_baq */
}

/* ---- the passes of a read group.  G* run one group per thread, A* one alignment per thread (the walks over ops and
 * markers are per alignment and make up most of the work); the marker table and the block lists stay in the group's
 * scratch in between.  The host plan calls them in the same order. ---- */
SPX_HD void count_clear(GroupCount &gc)
{
    gc.err = 0; gc.scored = 0; gc.n_cols = 0; gc.n_prob = 0; gc.n_rows = 0; gc.n_qe = 0;
    gc.cells = 0; gc.s_need = 0; gc.f_need = 0;
    for (int k = 0; k < SPX_N_CLASSES; ++k) { gc.cls_prob[k] = 0; gc.cls_cells[k] = 0; }
}

/* G1: error checks + marker columns before the insertion / clip filter (gc.n_cols = their number) */
SPX_HD void group_pass_merge(const GroupView &G, const Pools &P, const RefView &rv, GroupScratch &S, GroupCount &gc)
{
    count_clear(gc);
    for (int i = 0; i < G.n; ++i) S.nproj[i] = 0;
    for (int i = 0; i < G.n; ++i)
        if (G.st[i].err) { gc.err = G.st[i].err; return; }
    for (int i = 0; i < G.n; ++i)
        if (G.rec[i].tid < 0 || G.rec[i].tid >= rv.n_contigs) { gc.err = SPX_EINVAL; return; }
    gc.n_cols = group_merge(G, P, S.pos, S.keep);
}

/* A1 */
SPX_HD void aln_pass_filter(const GroupView &G, int i, const Pools &P, GroupScratch &S, const GroupCount &gc, int part = 0, int nparts = 1)
{
    if (gc.err || gc.n_cols == 0) return;
    /* part / nparts: the columns are dealt in contiguous shares to the lanes of a wave that holds ONE heavy alignment; a share is filled and
     * then filtered by its lane (the filter writes into the cells the fill has just written) */
    const int chunk = (gc.n_cols + nparts - 1) / nparts, c_lo = part * chunk, c_hi = c_lo + chunk < gc.n_cols ? c_lo + chunk : gc.n_cols;
    if (c_lo >= c_hi) return;
    aln_fill_range(G, i, P, S.pos, S.mk, gc.n_cols, c_lo, c_hi);
    aln_filter_range(G, i, P, S.pos, S.mk, S.keep, gc.n_cols, c_lo, c_hi);
}
/* A1b: after EVERY alignment of the group has been through aln_pass_filter (keep[] is final), before the group pass that
 * compacts the positions */
SPX_HD void aln_pass_compact(const GroupView &G, int i, GroupScratch &S, const GroupCount &gc)
{
    if (gc.err || gc.n_cols == 0) return;
    aln_compact(G, i, S.mk, S.keep, gc.n_cols);
}

/* G2: filtered columns, consensus windows */
SPX_HD void group_pass_blocks(const GroupView &G, const Pools &P, const Params &par, GroupScratch &S, GroupCount &gc)
{
    if (gc.err || gc.n_cols == 0) return;
    const int ncol = group_compact(G, S.pos, S.keep, gc.n_cols);
    gc.n_cols = ncol;
    if (ncol == 0) return;
    const int sc = group_blocks(G, P, par, S, ncol);
    if (sc < 0) { gc.err = sc; gc.n_cols = 0; return; }
    gc.scored = sc;
    if (!sc) gc.n_cols = 0;
}

/* G2 on the device, split at the projections: begin (columns, rounds up to the first projection), per-alignment
 * projections and resume (a fixed number of times), end (whatever is left, serially) */
SPX_HD void group_pass_blocks_begin(const GroupView &G, const Pools &P, const Params &par, GroupScratch &S, GroupCount &gc)
{
    if (gc.err || gc.n_cols == 0) return;
    const int ncol = group_compact(G, S.pos, S.keep, gc.n_cols);
    gc.n_cols = ncol;
    if (ncol == 0) return;
    blocks_begin(G, P, par, S, ncol, *S.bstate);
}
SPX_HD void group_pass_blocks_end(const GroupView &G, const Pools &P, const Params &par, GroupScratch &S, GroupCount &gc)
{
    if (gc.err || gc.n_cols == 0) return;
    const int sc = blocks_finish(G, P, par, S, *S.bstate);
    if (sc < 0) { gc.err = sc; gc.n_cols = 0; return; }
    gc.scored = sc;
    if (!sc) gc.n_cols = 0;
}

/* A2: work-list sizes of one alignment (ac.err: SPX_E* of its BAQ plan) */
SPX_HD void aln_pass_count(const GroupView &G, int i, const Pools &P, const RefView &rv, const Params &par, GroupScratch &S,
                           const GroupCount &gc, GroupCount &ac)
{
    count_clear(ac);
    if (gc.err || !gc.scored || !par.baq_flag) return;
    PlanBase at = {0, 0, 0, 0, 0};
    PlanOut none = {};
    const int rc = plan_baq<false>(G, i, P, rv, par, S, gc.n_cols, ac, at, none);
    if (rc) { count_clear(ac); ac.err = rc; }
}
/* the same for ONE share of the alignment's blocks (part of nparts, contiguous; plan_can_split must hold when nparts > 1): the share's
 * counts in ac (err = its error, nothing cleared: the caller combines the shares in order -- the first error wins, as in the walk over all
 * blocks) and what it adds to the output offsets in `add` */
SPX_HD void block_share(int nblocks, int part, int nparts, int &lo, int &hi)
{
    const int chunk = (nblocks + nparts - 1) / nparts;
    lo = part * chunk;
    hi = lo + chunk < nblocks ? lo + chunk : nblocks;
    if (lo > nblocks) lo = nblocks;
}
SPX_HD void aln_pass_count_part(const GroupView &G, int i, const Pools &P, const RefView &rv, const Params &par, GroupScratch &S,
                                const GroupCount &gc, GroupCount &ac, PlanBase &add, int part, int nparts)
{
    count_clear(ac);
    add = PlanBase{0, 0, 0, 0, 0};
    if (gc.err || !gc.scored || !par.baq_flag) return;
    int lo, hi;
    block_share(S.nproj[i], part, nparts, lo, hi);
    if (lo >= hi) return;
    PlanOut none = {};
    ac.err = plan_baq_range<false>(G, i, P, rv, par, S, gc.n_cols, ac, add, none, lo, hi);
}
SPX_HD void count_add(GroupCount &a, const GroupCount &b)
{
    a.n_prob += b.n_prob; a.n_rows += b.n_rows; a.n_qe += b.n_qe;
    a.cells += b.cells; a.s_need += b.s_need; a.f_need += b.f_need;
    for (int k = 0; k < SPX_N_CLASSES; ++k) { a.cls_prob[k] += b.cls_prob[k]; a.cls_cells[k] += b.cls_cells[k]; }
}

/* G3: sums of the group; an alignment's error (the first in alignment order) makes the whole group an error and
 * clears every count */
SPX_HD void group_pass_sum(const GroupView &G, GroupCount &gc, GroupCount *ac)
{
    if (gc.err) return;
    for (int i = 0; i < G.n; ++i)
        if (ac[i].err) {
            const int e = ac[i].err;
            count_clear(gc);
            gc.err = e;
            for (int k = 0; k < G.n; ++k) count_clear(ac[k]);
            return;
        }
    for (int i = 0; i < G.n; ++i) {
        gc.n_prob += ac[i].n_prob; gc.n_rows += ac[i].n_rows; gc.n_qe += ac[i].n_qe;
        gc.cells += ac[i].cells; gc.s_need += ac[i].s_need; gc.f_need += ac[i].f_need;
        for (int k = 0; k < SPX_N_CLASSES; ++k) { gc.cls_prob[k] += ac[i].cls_prob[k]; gc.cls_cells[k] += ac[i].cls_cells[k]; }
    }
}

/* A3: problems / rows / edits of one alignment at `at`, marker updates on its own cells */
SPX_HD int aln_pass_emit(const GroupView &G, int i, const Pools &P, const RefView &rv, const Params &par, GroupScratch &S,
                         const GroupCount &gc, PlanBase at, const PlanOut &out)
{
    if (gc.err || !gc.scored || !par.baq_flag) return 0;
    GroupCount dummy;
    return plan_baq<true>(G, i, P, rv, par, S, gc.n_cols, dummy, at, out); /* same control flow as the counting pass */
}
/* one share of the blocks, written at `at` = the alignment's offsets + what the shares in front of it add (aln_pass_count_part) */
SPX_HD int aln_pass_emit_part(const GroupView &G, int i, const Pools &P, const RefView &rv, const Params &par, GroupScratch &S,
                              const GroupCount &gc, PlanBase at, const PlanOut &out, int part, int nparts)
{
    if (gc.err || !gc.scored || !par.baq_flag) return 0;
    int lo, hi;
    block_share(S.nproj[i], part, nparts, lo, hi);
    if (lo >= hi) return 0;
    GroupCount dummy;
    return plan_baq_range<true>(G, i, P, rv, par, S, gc.n_cols, dummy, at, out, lo, hi);
}

/* G4: the group's marker table (n_cols * n entries) */
SPX_HD void group_pass_markers(const GroupView &G, GroupScratch &S, const GroupCount &gc, spx_dev_marker *mk_out, int32_t *mk_ref_pos, int part = 0,
                               int nparts = 1)
{
    if (gc.err || !gc.scored) return;
    const int n = G.n, ncol = gc.n_cols;
    /* every cell is independent: the lanes of a wave that shares one heavy group take the cells in turn (part / nparts, coalesced) */
    const int64_t cells = (int64_t)ncol * n;
    for (int64_t x = part; x < cells; x += nparts) {
        const int i = (int)(x % n);
        const Mk m = S.mk[x];
        spx_dev_marker dm;
        dm.row = m.row;
        dm.qfix = m.q;
        dm.is_match = m.is_match;
        dm.aln = (uint8_t)i;
        dm.first_of_pos = i == 0 ? (uint8_t)n : 0;
        mk_out[x] = dm;
        mk_ref_pos[x] = m.ref_pos;
    }
}

} // namespace spxl
#endif
