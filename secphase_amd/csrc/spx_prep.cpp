/*
 * spx_prep.cpp -- host-side group preparation of the MI355X path.
 *
 * Integer bookkeeping that stays on the host (SURVEY.md section 8 rows A1-A8 and the
 * control flow of A10); everything FP64 per DP cell runs on the device.  Every
 * alignment's CIGAR+cs is scanned ONCE into a flat op table (the reference
 * re-runs a regex iterator >= 7 times per alignment); markers, consensus
 * blocks and the BAQ window list are then derived from the tables.
 *
 * Behavioural contract, by reference line (/root/reference/programs):
 *   op table            submodules/cigar_it/cigar_it.c:14-69,145-211,213-308
 *   aligned extents     submodules/ptAlignment/ptAlignment.c:42-95
 *   markers             submodules/ptMarker/ptMarker.c:42-107,156-295
 *   blocks              submodules/ptMarker/ptMarker.c:328-667, src/secphase.c:162-170
 *   BAQ windows / rows  submodules/ptMarker/ptMarker.c:670-831
 *   HMM constants       htslib-1.17 probaln.c initialisation
 */
#include "spx_prep.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>

namespace spx {

static const unsigned char kNt16Int[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};

void RefIndex::index_ambiguous(const spx_ref *ref)
{
    npos.assign(ref->n_contigs, std::vector<int32_t>());
    for (int i = 0; i < ref->n_contigs; ++i) {
        const char *s = ref->bases + ref->seq_off[i];
        const int64_t n = ref->seq_off[i + 1] - ref->seq_off[i];
        for (int64_t k = 0; k < n; ++k) {
            const char c = s[k] & ~0x20;
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T') npos[i].push_back((int32_t)k);
        }
    }
}

void HostBatch::clear()
{
    ref_nib.clear(); qry_nib.clear(); ref_tid.clear(); ref_rfs.clear(); L.clear(); R.clear(); bw.clear(); row_off.clear(); n_rows.clear();
    hmm.clear(); rows.clear(); row_expect.clear(); row_rawq.clear(); qry4.clear(); qry_nibbles = 0;
    grp_index.clear(); mk_first.clear(); markers.clear(); n_aln.clear(); sec_mask.clear(); rfe.clear();
    grp_problems.clear(); grp_cells.clear(); grp_error.clear(); dp_cells = 0;
    rfs.clear(); atid.clear(); mk_ref_pos.clear();
    qe_rec.clear(); qe_pos.clear(); qe_len.clear(); qe_row0.clear(); qe_batch.clear();
}

template <class T>
static void cat(std::vector<T> &a, const std::vector<T> &b) { a.insert(a.end(), b.begin(), b.end()); }

void HostBatch::append(const HostBatch &o)
{
    const int32_t row_base = (int32_t)rows.size();
    const int64_t nib_base = qry_nibbles;
    const int32_t mk_base = (int32_t)markers.size();
    /* query windows are nibble packed: keep every appended part byte aligned */
    for (size_t i = 0; i < o.qry_nib.size(); ++i) qry_nib.push_back(o.qry_nib[i] + nib_base);
    cat(ref_nib, o.ref_nib); cat(ref_tid, o.ref_tid); cat(ref_rfs, o.ref_rfs); cat(L, o.L); cat(R, o.R); cat(bw, o.bw); cat(n_rows, o.n_rows); cat(hmm, o.hmm);
    for (size_t i = 0; i < o.row_off.size(); ++i) row_off.push_back(o.row_off[i] + row_base);
    cat(rows, o.rows); cat(row_expect, o.row_expect); cat(row_rawq, o.row_rawq);
    cat(qry4, o.qry4);
    qry_nibbles += (int64_t)o.qry4.size() * 2;
    cat(grp_index, o.grp_index);
    if (mk_first.empty()) mk_first.push_back(0);
    for (size_t i = 1; i < o.mk_first.size(); ++i) mk_first.push_back(o.mk_first[i] + mk_base);
    for (size_t i = 0; i < o.markers.size(); ++i) {
        spx_dev_marker m = o.markers[i];
        if (m.row >= 0) m.row += row_base;
        markers.push_back(m);
    }
    cat(n_aln, o.n_aln); cat(sec_mask, o.sec_mask); cat(rfe, o.rfe); cat(grp_problems, o.grp_problems);
    cat(rfs, o.rfs); cat(atid, o.atid); cat(mk_ref_pos, o.mk_ref_pos);
    cat(grp_cells, o.grp_cells);
    dp_cells += o.dp_cells;
    cat(qe_rec, o.qe_rec); cat(qe_pos, o.qe_pos); cat(qe_len, o.qe_len); cat(qe_batch, o.qe_batch);
    for (size_t i = 0; i < o.qe_row0.size(); ++i) qe_row0.push_back(o.qe_len[i] > 0 ? o.qe_row0[i] + row_base : 0);
}

void HostBatch::assign_merged(std::vector<HostBatch> &parts, int n_threads)
{
    clear();
    const size_t P = parts.size();
    /* element offsets of every part in the merged arrays */
    std::vector<size_t> o_np(P + 1, 0), o_nr(P + 1, 0), o_q4(P + 1, 0), o_ng(P + 1, 0), o_nm(P + 1, 0), o_nq(P + 1, 0), o_ge(P + 1, 0);
    for (size_t t = 0; t < P; ++t) {
        const HostBatch &o = parts[t];
        o_np[t + 1] = o_np[t] + o.L.size();
        o_nr[t + 1] = o_nr[t] + o.rows.size();
        o_q4[t + 1] = o_q4[t] + o.qry4.size();
        o_ng[t + 1] = o_ng[t] + o.grp_index.size();
        o_nm[t + 1] = o_nm[t] + o.markers.size();
        o_nq[t + 1] = o_nq[t] + o.qe_rec.size();
        o_ge[t + 1] = o_ge[t] + o.grp_error.size();
        dp_cells += o.dp_cells;
    }
    const size_t np = o_np[P], nr = o_nr[P], ng = o_ng[P], nm = o_nm[P], nq = o_nq[P];
    ref_nib.resize(np); qry_nib.resize(np); ref_tid.resize(np); ref_rfs.resize(np); L.resize(np); R.resize(np); bw.resize(np);
    row_off.resize(np); n_rows.resize(np); hmm.resize(np * SPX_H_N);
    rows.resize(nr); row_expect.resize(nr); row_rawq.resize(nr);
    qry4.resize(o_q4[P]);
    qry_nibbles = (int64_t)o_q4[P] * 2;
    grp_index.resize(ng); mk_first.resize(ng + 1); n_aln.resize(ng); sec_mask.resize(ng); rfe.resize(ng * 10); rfs.resize(ng * 10);
    atid.resize(ng * 10); grp_problems.resize(ng); grp_cells.resize(ng);
    markers.resize(nm); mk_ref_pos.resize(nm);
    qe_rec.resize(nq); qe_pos.resize(nq); qe_len.resize(nq); qe_row0.resize(nq); qe_batch.resize(nq);
    grp_error.resize(o_ge[P]);
    mk_first[0] = 0;
    auto cp = [](auto &dst, size_t at, const auto &src) {
        if (!src.empty()) memcpy(dst.data() + at, src.data(), src.size() * sizeof(src[0]));
    };
    std::atomic<size_t> next(0);
    auto work = [&]() {
        for (;;) {
            const size_t t = next.fetch_add(1);
            if (t >= P) break;
            const HostBatch &o = parts[t];
            const size_t a = o_np[t], r0 = o_nr[t], g0 = o_ng[t], m0 = o_nm[t], q0 = o_nq[t];
            const int64_t nib_base = (int64_t)o_q4[t] * 2;
            const int32_t row_base = (int32_t)r0, mk_base = (int32_t)m0;
            cp(ref_nib, a, o.ref_nib); cp(ref_tid, a, o.ref_tid); cp(ref_rfs, a, o.ref_rfs); cp(L, a, o.L); cp(R, a, o.R);
            cp(bw, a, o.bw); cp(n_rows, a, o.n_rows); cp(hmm, a * SPX_H_N, o.hmm);
            for (size_t i = 0; i < o.qry_nib.size(); ++i) qry_nib[a + i] = o.qry_nib[i] + nib_base;
            for (size_t i = 0; i < o.row_off.size(); ++i) row_off[a + i] = o.row_off[i] + row_base;
            cp(rows, r0, o.rows); cp(row_expect, r0, o.row_expect); cp(row_rawq, r0, o.row_rawq);
            cp(qry4, o_q4[t], o.qry4);
            cp(grp_index, g0, o.grp_index); cp(n_aln, g0, o.n_aln); cp(sec_mask, g0, o.sec_mask); cp(rfe, g0 * 10, o.rfe);
            cp(rfs, g0 * 10, o.rfs); cp(atid, g0 * 10, o.atid); cp(grp_problems, g0, o.grp_problems); cp(grp_cells, g0, o.grp_cells);
            for (size_t i = 1; i < o.mk_first.size(); ++i) mk_first[g0 + i] = o.mk_first[i] + mk_base;
            for (size_t i = 0; i < o.markers.size(); ++i) {
                spx_dev_marker m = o.markers[i];
                if (m.row >= 0) m.row += row_base;
                markers[m0 + i] = m;
            }
            cp(mk_ref_pos, m0, o.mk_ref_pos);
            cp(qe_rec, q0, o.qe_rec); cp(qe_pos, q0, o.qe_pos); cp(qe_len, q0, o.qe_len); cp(qe_batch, q0, o.qe_batch);
            for (size_t i = 0; i < o.qe_row0.size(); ++i) qe_row0[q0 + i] = o.qe_len[i] > 0 ? o.qe_row0[i] + row_base : 0;
            cp(grp_error, o_ge[t], o.grp_error);
            parts[t] = HostBatch(); /* give the memory back early */
        }
    };
    const int nt = std::max(1, std::min<int>(n_threads, (int)P));
    if (nt == 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work);
        for (auto &t : th) t.join();
    }
}

/* ---------------- HMM set-up (host, once per problem) ---------------- */
void hmm_constants(int l_ref, int l_query, float d, float e, int set_q, double *h)
{
    /* the float/double mix below is the one of probaln_glocal's initialisation:
     * probaln_par_t holds floats, 1 - c->d - c->d and (1 - c->d) / l_ref are float expressions */
    const double sM = 1. / (2 * l_query + 2), sI = sM;
    const float qf = (float)pow(10, -set_q / 10.);
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
    h[SPX_H_PAD0] = h[SPX_H_PAD1] = h[SPX_H_PAD2] = 0.;
}

int effective_bw(int l_ref, int l_query, int bw_in)
{
    int bw = l_ref > l_query ? l_ref : l_query;
    if (bw > bw_in) bw = bw_in;
    if (bw < abs(l_ref - l_query)) bw = abs(l_ref - l_query);
    return bw;
}

int64_t band_cells(int L, int R, int bw)
{
    /* sum over rows i = 1..L of (min(R, i+bw) - max(1, i-bw) + 1), in closed form (called once per problem on the
     * launch path).  effective_bw() guarantees bw >= |R - L|, so every row has at least one cell. */
    const int64_t l = L, r = R, w = bw;
    const int64_t a = std::max<int64_t>(0, std::min<int64_t>(l, r - w)); /* rows with i + bw <= R */
    const int64_t hi = a * (a + 1) / 2 + a * w + (l - a) * r;
    const int64_t b = std::min<int64_t>(l, w + 1);                       /* rows with i - bw <= 1 */
    const int64_t lo = b + (l * (l + 1) / 2 - b * (b + 1) / 2) - (l - b) * w;
    return hi - lo + l;
}

/* band classes = kernel instantiations (spx_launch_baq): four exact widths (one-lane forward kernel), then generic
 * ones by capacity.  Classes 12 and 13 were added for the ONT widths: four lanes per problem cost half the serial
 * passes of the (8,16) class, and 28 / 30 slots per lane is what the register file still takes (38 / 64 spilled
 * VGPRs in the forward kernel; (4,32) spills 134 and loses). */
static const int kClassSlots[SPX_N_CLASSES] = {42, 44, 46, 48, 48, 64, 104, 128, 256, 512, 1024, 2048, 112, 120};
static const int kClassLanes[SPX_N_CLASSES] = {1, 1, 1, 1, 2, 4, 4, 8, 16, 32, 64, 64, 4, 4};
static const int kClassLanesBwd[SPX_N_CLASSES] = {2, 2, 2, 2, 2, 4, 4, 4, 16, 32, 64, 64, 4, 4};
int band_class(int W)
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
    for (int c = 7; c < 12; ++c)
        if (W <= kClassSlots[c]) return c;
    return -1;
}
int class_lanes(int cls) { return kClassLanes[cls]; }
int class_lanes_bwd(int cls) { return kClassLanesBwd[cls]; }
int class_slots(int cls) { return kClassSlots[cls]; }

/* largest x in (0,1] with (int)(-4.343*log(x)+.499) >= k, by bisection on the
 * double lattice with the HOST libm -- the same log() the reference's CPU path calls */
static int phred_of(double x) { return (int)(-4.343 * log(x) + .499); }
void phred_thresholds(double *thr)
{
    thr[0] = 1.0;
    for (int k = 1; k <= 101; ++k) {
        double lo = 4.9406564584124654e-324, hi = 1.0; /* f(lo) >= k, f(hi) = 0 < k */
        uint64_t a, b;
        memcpy(&a, &lo, 8);
        memcpy(&b, &hi, 8);
        while (b - a > 1) {
            uint64_t m = a + (b - a) / 2;
            double x;
            memcpy(&x, &m, 8);
            if (phred_of(x) >= k) a = m; else b = m;
        }
        memcpy(&thr[k], &a, 8);
    }
}

void score_tables(double *match_tbl, double *mis_tbl)
{
    /* calc_alignment_score + reverse_quality, ptMarker.c:298-325 */
    for (int q = 0; q < 256; ++q) {
        double rq;
        if (q >= 93) rq = 0;
        else if (q == 0) rq = 93;
        else {
            double p = 1 - pow(10, (double)q / -10);
            rq = -10 * log(p);
        }
        match_tbl[q] = -1 * rq;
        mis_tbl[q] = -1 * q - 10 * log(3);
    }
}

/* ---------------- per-alignment op table ---------------- */
struct Op {
    int32_t op, len, ret;
    int32_t sqs, sqe, rfs, rfe, rds, rde;
};

struct Blk {
    int32_t rfs, rfe, sqs, sqe, rds, rde;
};

struct Mk {
    int32_t pos;  /* read_pos_f */
    int32_t aln;
    int32_t base_idx;
    int32_t q;
    int32_t is_match;
    int32_t row;  /* wanted-row index (batch global) or -1 */
    int32_t ref_pos; /* ptMarker.ref_pos: set for mismatches at creation, for matches inside an '=' op */
};

struct Aln {
    int32_t rec; /* record index in the batch */
    uint32_t flag;
    int32_t tid, pos, l_qseq, n_cigar;
    bool rev;
    const uint32_t *cigar;
    const uint8_t *seq4, *qual;
    const char *cs;
    const char *md;      /* consulted only when the record has no cs tag */
    std::vector<Op> ops; /* ops[0] = state before the first step */
    int32_t n_visit;     /* states a while(next) loop visits: ops[1..n_visit-1] */
    int32_t rest;        /* state the iterator rests on afterwards */
    int32_t lclip, rclip;
    int32_t rfs, rfe, rds, rde;
    std::vector<Blk> conf, flank;
    bool have_conf;
};

static inline bool lower_c(char c) { return c >= 'a' && c <= 'z'; }
static inline bool digit_c(char c) { return c >= '0' && c <= '9'; }

/* first short-form cs token at or after s (what an un-anchored POSIX search of
 * (:[0-9]+)|([+-][a-z]+)|((\*[a-z]+)+) returns); 0 if none */
static inline bool next_cs_token(const char *s, int &so, int &eo)
{
    for (int p = 0; s[p]; ++p) {
        const char c = s[p];
        if (c == ':') {
            if (!digit_c(s[p + 1])) continue;
            int e = p + 1;
            while (digit_c(s[e])) ++e;
            so = p; eo = e;
            return true;
        }
        if (c == '+' || c == '-') {
            if (!lower_c(s[p + 1])) continue;
            int e = p + 1;
            while (lower_c(s[e])) ++e;
            so = p; eo = e;
            return true;
        }
        if (c == '*') {
            if (!lower_c(s[p + 1])) continue;
            int e = p;
            while (s[e] == '*' && lower_c(s[e + 1])) {
                ++e;
                while (lower_c(s[e])) ++e;
            }
            so = p; eo = e;
            return true;
        }
    }
    return false;
}

static inline bool upper_c(char c) { return c >= 'A' && c <= 'Z'; }

/* first MD token at or after s: a mismatch run X(0X)*, a match count, or a deletion ^XXX
 * (the un-anchored POSIX search of cigar_it.h:10) */
static inline bool next_md_token(const char *s, int &so, int &eo)
{
    for (int p = 0; s[p]; ++p) {
        const char c = s[p];
        if (upper_c(c)) {
            int e = p + 1;
            while (s[e] == '0' && upper_c(s[e + 1])) e += 2;
            so = p; eo = e;
            return true;
        }
        if (digit_c(c)) {
            int e = p + 1;
            while (digit_c(s[e])) ++e;
            so = p; eo = e;
            return true;
        }
        if (c == '^' && upper_c(s[p + 1])) {
            int e = p + 1;
            while (upper_c(s[e])) ++e;
            so = p; eo = e;
            return true;
        }
    }
    return false;
}

/* one MD step (cigar_it.c:72-141): a lone "0" separates two mismatches and is skipped */
static int md_step(const char *md, int &at, Op &cur)
{
    for (;;) {
        const char *s = md + at;
        int so, eo;
        if (!next_md_token(s, so, eo)) return 0;
        const char c = s[so];
        if (c == '0') { cur.op = SPX_CDIFF; cur.len = 0; }
        else if (c <= '9') {
            char buf[24];
            int n = std::min(eo - so, 19);
            memcpy(buf, s, n); /* sic: from the start of the shifted string, like the reference */
            buf[n] = 0;
            cur.op = SPX_CEQUAL;
            cur.len = atoi(buf);
        } else if (c < 90) { cur.op = SPX_CDIFF; cur.len = 1 + (eo - so - 1) / 2; }
        else if (c == '^') { cur.op = SPX_CDEL; cur.len = eo - so - 1; }
        at += eo;
        if (cur.len != 0) return cur.len;
    }
}

static int build_ops(Aln &a)
{
    a.ops.clear();
    a.lclip = ((a.cigar[0] & 0xf) == SPX_CHARD_CLIP) ? (int32_t)(a.cigar[0] >> 4) : 0;
    a.rclip = ((a.cigar[a.n_cigar - 1] & 0xf) == SPX_CHARD_CLIP) ? (int32_t)(a.cigar[a.n_cigar - 1] >> 4) : 0;
    const bool use_cs = a.cs != nullptr, use_md = !use_cs && a.md != nullptr;
    if (!use_cs && !use_md) return SPX_ENOTAG; /* neither cs nor MD: the reference exits (cigar_it.c:64-67) */
    int md_at = 0;
    Op cur;
    cur.op = 255; cur.len = 0; cur.ret = 0;
    cur.sqs = 0; cur.sqe = -1;
    cur.rfs = a.pos; cur.rfe = a.pos - 1;
    const int32_t T = a.lclip + a.rclip + a.l_qseq;
    cur.rds = a.rev ? T : 0;
    cur.rde = a.rev ? T - 1 : -1;
    a.ops.push_back(cur);
    int idx = -1, remain = 0, cs_at = 0;
    while (idx != a.n_cigar - 1) {
        ++idx;
        const int op = a.cigar[idx] & 0xf, len = (int)(a.cigar[idx] >> 4);
        int rd, sq, rf;
        const bool mtype = op == SPX_CMATCH || op == SPX_CEQUAL || op == SPX_CDIFF;
        if (use_cs && (mtype || op == SPX_CINS || op == SPX_CDEL)) {
            int so, eo;
            const char *s = a.cs + cs_at;
            if (next_cs_token(s, so, eo)) {
                const char c = s[so];
                if (c == ':') { cur.op = SPX_CEQUAL; cur.len = atoi(s + so + 1); }
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
                if (remain >= 0) md_step(a.md, md_at, cur);
                if (remain < 0) {
                    cur.op = SPX_CEQUAL;
                    cur.len = std::min(len, -remain);
                    remain += len;
                } else {
                    const int md_len = cur.len;
                    cur.len = std::min(cur.len, remain);
                    remain -= md_len;
                }
                if (remain > 0) --idx;
            }
            rd = sq = rf = cur.len;
        } else if (op == SPX_CINS) {
            cur.len = len; cur.op = op;
            rd = sq = len; rf = 0;
        } else if (op == SPX_CDEL) {
            if (use_md) md_step(a.md, md_at, cur);
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
        if (a.rev) { cur.rde = cur.rds - 1; cur.rds -= rd; }
        else { cur.rds = cur.rde + 1; cur.rde += rd; }
        cur.sqs = cur.sqe + 1; cur.sqe += sq;
        cur.rfs = cur.rfe + 1; cur.rfe += rf;
        cur.ret = cur.len;
        a.ops.push_back(cur);
        if (a.ops.size() > 40000000u) return SPX_EINVAL;
    }
    const int n = (int)a.ops.size();
    { /* U6: no aligned base at all (e.g. a CIGAR of clips only): undefined in the reference, rejected like U3 */
        bool aligned = false;
        for (int t = 1; t < n; ++t)
            aligned |= (a.ops[t].op == SPX_CMATCH || a.ops[t].op == SPX_CEQUAL || a.ops[t].op == SPX_CDIFF) && a.ops[t].ret > 0;
        if (!aligned || a.l_qseq <= 0) return SPX_EUNSUPPORTED;
    }
    a.n_visit = n;
    a.rest = n - 1;
    for (int t = 1; t < n; ++t)
        if (a.ops[t].ret == 0) { a.n_visit = t; a.rest = t; break; }
    return 0;
}

static inline bool mx(int op) { return op == SPX_CMATCH || op == SPX_CEQUAL || op == SPX_CDIFF; }

static void aligned_extents(Aln &a)
{
    a.rfs = a.rfe = a.rds = a.rde = -1;
    for (int t = 1; t < a.n_visit; ++t) {
        const Op &o = a.ops[t];
        if (a.rfs == -1 && mx(o.op)) {
            a.rfs = o.rfs;
            if (a.rev) a.rde = o.rde; else a.rds = o.rds;
        }
        if (a.rfe == -1 && a.rfs != -1 && (o.op == SPX_CHARD_CLIP || o.op == SPX_CSOFT_CLIP)) {
            a.rfe = o.rfe;
            if (a.rev) a.rds = o.rde + 1; else a.rde = o.rds - 1;
        }
    }
    const Op &o = a.ops[a.rest];
    if (a.rfe == -1 && mx(o.op)) {
        a.rfe = o.rfe;
        if (a.rev) a.rds = o.rds; else a.rde = o.rde;
    }
}

/* ---------------- markers ---------------- */
static inline bool mk_less(const Mk &x, const Mk &y) { return x.pos != y.pos ? x.pos < y.pos : x.aln < y.aln; }

static Mk match_marker(const Aln &a, int ai, int pos)
{
    Mk m;
    m.pos = pos; m.aln = ai; m.is_match = 1; m.row = -1; m.ref_pos = -1;
    m.base_idx = a.rev ? a.l_qseq + a.rclip - pos - 1 : pos - a.lclip;
    m.q = (m.base_idx >= 0 && m.base_idx < a.l_qseq) ? a.qual[m.base_idx] : 0;
    return m;
}

static void collect_markers(std::vector<Aln> &al, int min_q, std::vector<Mk> &mk, std::vector<Mk> &tmp)
{
    const int n = (int)al.size();
    mk.clear();
    /* mismatch bases with raw quality >= min_q */
    for (int i = 0; i < n; ++i) {
        const Aln &a = al[i];
        for (int t = 1; t < a.n_visit; ++t) {
            const Op &o = a.ops[t];
            if (o.op != SPX_CDIFF) continue;
            for (int j = 0; j < o.len; ++j) {
                const int q = a.qual[o.sqs + j];
                if (q < min_q) continue;
                Mk m;
                m.aln = i; m.base_idx = o.sqs + j; m.pos = a.rev ? o.rde - j : o.rds + j;
                m.q = q; m.is_match = 0; m.row = -1; m.ref_pos = o.rfs + j;
                mk.push_back(m);
            }
        }
    }
    std::sort(mk.begin(), mk.end(), mk_less);
    /* drop positions where every alignment mismatches; give the others a full column of n markers */
    tmp.clear();
    for (size_t s = 0; s < mk.size();) {
        size_t e = s;
        while (e < mk.size() && mk[e].pos == mk[s].pos) ++e;
        if ((int)(e - s) != n) {
            size_t k = s;
            for (int ai = 0; ai < n; ++ai) {
                if (k < e && mk[k].aln == ai) tmp.push_back(mk[k++]);
                else tmp.push_back(match_marker(al[ai], ai, mk[s].pos));
            }
        }
        s = e;
    }
    mk.swap(tmp);
    /* positions inside an insertion / clip of any alignment are not comparable: drop the column */
    if (mk.empty()) return;
    const int ncol = (int)mk.size() / n;
    std::vector<char> keep(ncol, 1);
    for (int i = 0; i < n; ++i) {
        const Aln &a = al[i];
        int col = a.rev ? ncol - 1 : 0;
        const int step = a.rev ? -1 : 1;
        for (int t = 1; t < a.n_visit && col >= 0 && col < ncol; ++t) {
            const Op &o = a.ops[t];
            while (col >= 0 && col < ncol) {
                const int p = mk[(size_t)col * n].pos;
                if (!(o.rds <= p && p <= o.rde)) break;
                if (o.op == SPX_CINS || o.op == SPX_CSOFT_CLIP || o.op == SPX_CHARD_CLIP) keep[col] = 0;
                if (o.op == SPX_CEQUAL) /* ptMarker.c:184-187: reference position of this alignment's marker */
                    mk[(size_t)col * n + i].ref_pos = a.rev ? o.rfs + o.rde - p : o.rfs + p - o.rds;
                col += step;
            }
        }
    }
    tmp.clear();
    for (int c = 0; c < ncol; ++c)
        if (keep[c])
            for (int i = 0; i < n; ++i) tmp.push_back(mk[(size_t)c * n + i]);
    mk.swap(tmp);
}

/* ---------------- blocks ---------------- */
static void confident_blocks(Aln &a, int thr)
{
    a.conf.clear();
    int c_sqs = 0, c_rfs = a.pos;
    int c_rd = a.rev ? a.ops[0].rde : a.ops[0].rds;
    auto emit = [&](const Op &o) {
        Blk b;
        b.rfs = c_rfs; b.rfe = o.rfs - 1; b.sqs = c_sqs; b.sqe = o.sqs - 1;
        if (a.rev) { b.rds = o.rde + 1; b.rde = c_rd; } else { b.rds = c_rd; b.rde = o.rds - 1; }
        a.conf.push_back(b);
    };
    for (int t = 1; t < a.n_visit; ++t) {
        const Op &o = a.ops[t];
        const bool indel = o.op == SPX_CINS || o.op == SPX_CDEL;
        const bool clip = o.op == SPX_CSOFT_CLIP || o.op == SPX_CHARD_CLIP;
        if (!(indel || clip)) continue;
        if (indel && o.len <= thr) continue;
        if (c_sqs < o.sqs && c_rfs < o.rfs) emit(o);
        c_sqs = o.sqe + 1;
        c_rfs = o.rfe + 1;
        c_rd = a.rev ? o.rds - 1 : o.rde + 1;
    }
    const Op &o = a.ops[a.rest];
    if (c_sqs <= o.sqe) {
        Blk b;
        b.rfs = c_rfs; b.rfe = o.rfe; b.sqs = c_sqs; b.sqe = o.sqe;
        if (a.rev) { b.rds = o.rds; b.rde = c_rd; } else { b.rds = c_rd; b.rde = o.rde; }
        a.conf.push_back(b);
    }
    a.have_conf = true;
}

static void flank_blocks(Aln &a, const std::vector<Mk> &mk, int margin)
{
    a.flank.clear();
    int start = std::max(a.rds, mk[0].pos - margin), end = std::min(a.rde, mk[0].pos + margin);
    for (size_t i = 1; i < mk.size(); ++i) {
        const int cs = std::max(a.rds, mk[i].pos - margin), ce = std::min(a.rde, mk[i].pos + margin);
        if (cs < end) end = ce;
        else {
            Blk b = {-1, -1, -1, -1, start, end};
            a.flank.push_back(b);
            start = cs; end = ce;
        }
    }
    Blk b = {-1, -1, -1, -1, start, end};
    a.flank.push_back(b);
}

static void intersect(const std::vector<Blk> &x, const std::vector<Blk> &y, std::vector<Blk> &out)
{
    out.clear();
    if (x.empty() || y.empty()) return;
    size_t j = 0;
    for (size_t i = 0; i < x.size(); ++i) {
        while (j < y.size() && y[j].rde < x[i].rds) ++j;
        while (j < y.size() && y[j].rds < x[i].rde) {
            Blk b = {-1, -1, -1, -1, std::max(x[i].rds, y[j].rds), std::min(x[i].rde, y[j].rde)};
            out.push_back(b);
            if (y[j].rde <= x[i].rde) ++j; else break;
        }
    }
}

static inline bool by_rds(const Blk &x, const Blk &y) { return x.rds < y.rds; }
static inline bool by_sqs(const Blk &x, const Blk &y) { return x.sqs < y.sqs; }

/* consensus windows in read coordinates, then projected onto each alignment */
static int consensus_blocks(std::vector<Aln> &al, int thr, std::vector<Blk> &cur, std::vector<Blk> &nxt)
{
    const int n = (int)al.size();
    std::sort(al[0].conf.begin(), al[0].conf.end(), by_rds);
    cur = al[0].conf;
    for (int i = 1; i < n; ++i) {
        std::sort(al[i].conf.begin(), al[i].conf.end(), by_rds);
        intersect(cur, al[i].conf, nxt);
        cur.swap(nxt);
    }
    for (int i = 0; i < n; ++i) {
        std::sort(al[i].flank.begin(), al[i].flank.end(), by_rds);
        intersect(cur, al[i].flank, nxt);
        cur.swap(nxt);
    }
    if (cur.empty()) {
        for (int i = 0; i < n; ++i) al[i].conf.clear();
        return 0;
    }
    const int nb = (int)cur.size();
    for (int i = 0; i < n; ++i) {
        Aln &a = al[i];
        std::vector<Blk> &out = nxt;
        out.clear();
        const bool rev = a.rev;
        int j = rev ? nb - 1 : 0;
        bool have = true, del_flag = false;
        int bs = rev ? -cur[j].rde : cur[j].rds, be = rev ? -cur[j].rds : cur[j].rde;
        int rfs = -1, rfe = -1, sqs = -1, sqe = -1;
        for (int t = 1; t < a.n_visit; ++t) {
            const Op &o = a.ops[t];
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
                    Blk b = {rfs, rfe, sqs, sqe, cur[j].rds, cur[j].rde};
                    out.push_back(b);
                    if (rev && j > 0) { --j; bs = -cur[j].rde; be = -cur[j].rds; }
                    else if (!rev && j < nb - 1) { ++j; bs = cur[j].rds; be = cur[j].rde; }
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
        std::sort(out.begin(), out.end(), by_sqs);
        a.conf = out;
    }
    return nb;
}

static bool blocks_too_long(const std::vector<Aln> &al, int thr)
{
    bool flag = false;
    for (size_t j = 0; j < al.size(); ++j) {
        if (!al[j].have_conf || al[j].conf.empty()) return true;
        for (const Blk &b : al[j].conf)
            if ((b.sqe - b.sqs) > thr || (b.rfe - b.rfs) > thr) flag = true;
    }
    return flag;
}

/* ---------------- BAQ windows of one alignment ---------------- */
struct GroupScratch {
    std::vector<Aln> al;
    std::vector<Mk> mk, mtmp;
    std::vector<Blk> b1, b2;
    std::vector<int32_t> own;      /* indices into mk of this alignment's markers, in seq order */
    std::vector<int32_t> rows_t, rows_mk;
};

static int plan_baq(const Aln &a, int ai, const std::vector<Mk> &mkc, std::vector<Mk> &mk, const RefIndex &ref,
                    const spx_params *par, GroupScratch &S, HostBatch &out, int &n_prob, int64_t &cells)
{
    const int nm = (int)mkc.size();
    const int step = a.rev ? -1 : 1;
    int j = a.rev ? nm - 1 : 0;
    int ci = 0;
    const int margin = 10;
    const int last = (int)a.ops.size() - 1;
    auto adv = [&]() -> int { if (ci < last) { ++ci; return a.ops[ci].ret; } return 0; };
    const float d = (float)par->conf_d, e = (float)par->conf_e;
    const bool all_rows = (par->flags & SPX_PAR_ALL_ROWS) != 0;
    auto zero_edit = [&](int base) {
        if (!all_rows) return;
        out.qe_rec.push_back(a.rec); out.qe_pos.push_back(base); out.qe_len.push_back(0); out.qe_row0.push_back(0);
        out.qe_batch.push_back(0);
    };
    for (size_t bi = 0; bi < a.conf.size(); ++bi) {
        const Blk &b = a.conf[bi];
        while (a.ops[ci].sqe < b.sqs || a.ops[ci].rfe < b.rfs)
            if (adv() == 0) break;
        /* markers of this alignment in the leading margin lose their quality */
        while (j >= 0 && j < nm && (mk[j].base_idx < b.sqs + margin || mk[j].aln != ai)) {
            if (mk[j].aln == ai && b.sqs <= mk[j].base_idx) { mk[j].q = 0; mk[j].row = -1; zero_edit(mk[j].base_idx); }
            j += step;
        }
        if (j >= 0 && j < nm && mk[j].base_idx <= b.sqe - margin && b.sqs + margin <= mk[j].base_idx) {
            const int L = b.sqe - b.sqs + 1, R = b.rfe - b.rfs + 1;
            if (L <= 0 || R <= 0) return SPX_EINVAL;
            if (b.rfs < 0 || b.rfe >= ref.len[a.tid]) return SPX_EINVAL;
            const int bw_in = (int)(abs(R - L) + par->conf_b);
            const int bw = effective_bw(R, L, bw_in);
            if (band_class(2 * bw + 1) < 0) return SPX_EUNSUPPORTED;
            /* wanted rows: this alignment's markers in [sqs+margin, sqe-margin) keep a BAQ value; with
             * SPX_PAR_ALL_ROWS every base of that range is wanted (the write-back at ptMarker.c:802) */
            S.rows_t.clear(); S.rows_mk.clear();
            if (all_rows)
                for (int t = margin; t < L - margin; ++t) { S.rows_t.push_back(t); S.rows_mk.push_back(-1); }
            for (int k = j; k >= 0 && k < nm; k += step) {
                if (mk[k].aln != ai) continue;
                if (mk[k].base_idx > b.sqe) break;
                const int t = mk[k].base_idx - b.sqs;
                if (t >= margin && t < L - margin) {
                    if (all_rows) S.rows_mk[t - margin] = k;
                    else { S.rows_t.push_back(t); S.rows_mk.push_back(k); }
                }
            }
            const int32_t row0 = (int32_t)out.rows.size();
            for (size_t w = 0; w < S.rows_t.size(); ++w) {
                out.rows.push_back(S.rows_t[w] + 1);
                out.row_expect.push_back(-1);
                out.row_rawq.push_back(a.qual[b.sqs + S.rows_t[w]]);
            }
            /* expected reference index of every wanted base, from the CIGAR walk of the write-back loop */
            std::vector<char> covered(S.rows_t.size(), 0);
            while (a.ops[ci].sqs <= b.sqe || a.ops[ci].rfs <= b.rfe) {
                const Op &o = a.ops[ci];
                int x = o.rfs - b.rfs, y = o.sqs - b.sqs;
                if (x < 0) x = 0;
                if (y < 0) y = 0;
                if (mx(o.op)) {
                    const int len = std::min(o.len, std::min(o.sqe, b.sqe) - std::max(o.sqs, b.sqs) + 1);
                    for (size_t w = 0; w < S.rows_t.size(); ++w) {
                        const int t = S.rows_t[w];
                        if (t >= y && t < y + len) { out.row_expect[row0 + w] = x + (t - y); covered[w] = 1; }
                    }
                }
                if (o.sqe <= b.sqe || o.rfe <= b.rfe) { if (adv() == 0) break; }
                else break;
            }
            if (all_rows) {
                out.qe_rec.push_back(a.rec); out.qe_pos.push_back(b.sqs + margin); out.qe_len.push_back((int32_t)S.rows_t.size());
                out.qe_row0.push_back(row0); out.qe_batch.push_back(0);
            }
            for (size_t w = 0; w < S.rows_t.size(); ++w) {
                if (S.rows_mk[w] < 0) continue;
                Mk &m = mk[S.rows_mk[w]];
                if (covered[w]) m.row = row0 + (int32_t)w;
                else { m.row = -1; m.q = par->set_q < 94 ? par->set_q : 93; } /* base not under an M op: keeps set_q */
            }
            /* the problem itself */
            out.ref_nib.push_back(ref.nib_off[a.tid] + b.rfs);
            out.ref_tid.push_back(a.tid);
            out.ref_rfs.push_back(b.rfs);
            out.qry_nib.push_back(out.qry_nibbles);
            out.L.push_back(L); out.R.push_back(R); out.bw.push_back(bw);
            out.row_off.push_back(row0);
            out.n_rows.push_back((int32_t)S.rows_t.size());
            out.hmm.resize(out.hmm.size() + SPX_H_N);
            hmm_constants(R, L, d, e, par->set_q, &out.hmm[out.hmm.size() - SPX_H_N]);
            {
                const size_t nb = ((size_t)(L + 7) / 8) * 4, at = out.qry4.size(); /* whole dwords: the device reads 8 codes at a time */
                bool has_n = ref.window_has_n(a.tid, b.rfs, R);
                out.qry4.resize(at + nb, 0);
                for (int k = 0; k < L; ++k) {
                    const int p = b.sqs + k;
                    const unsigned code = kNt16Int[(a.seq4[p >> 1] >> ((~p & 1) << 2)) & 0xf];
                    has_n |= code > 3;
                    out.qry4[at + (k >> 1)] |= (uint8_t)(code << ((k & 1) << 2));
                }
                out.qry_nibbles += (int64_t)nb * 2;
                out.hmm[out.hmm.size() - SPX_H_N + SPX_H_PAD0] = has_n ? 1.0 : 0.0;
            }
            ++n_prob;
            cells += band_cells(L, R, bw);
        }
        /* markers in the trailing margin lose their quality */
        while (j >= 0 && j < nm && ((mk[j].base_idx <= b.sqe && mk[j].aln == ai) || mk[j].aln != ai)) {
            if (b.sqe - margin <= mk[j].base_idx && mk[j].aln == ai) { mk[j].q = 0; mk[j].row = -1; zero_edit(mk[j].base_idx); }
            j += step;
        }
    }
    (void)mkc;
    return 0;
}

static int group_records(const spx_batch *bt, int g, int *rec)
{
    int n = 0;
    for (int a = bt->grp_first[g]; a < bt->grp_first[g + 1]; ++a) {
        if (bt->flag[a] & SPX_FUNMAP) continue;
        if (n > 10) continue;
        rec[n++] = a;
    }
    return n;
}

static bool dispatched(const spx_batch *bt, int g)
{
    int rec[16], n = group_records(bt, g, rec), supp = 0, prim = 0;
    for (int i = 0; i < n; ++i) {
        if (bt->flag[rec[i]] & SPX_FSUPPLEMENTARY) ++supp;
        if (!(bt->flag[rec[i]] & SPX_FSECONDARY)) ++prim;
    }
    return n > 1 && n <= 10 && supp == 0 && prim == 1;
}

static int prepare_one(const spx_batch *bt, const RefIndex &ref, const spx_params *par, int g, GroupScratch &S,
                       HostBatch &out)
{
    int rec[16];
    const int n = group_records(bt, g, rec);
    S.al.resize(n);
    uint16_t sec = 0;
    for (int i = 0; i < n; ++i) {
        Aln &a = S.al[i];
        const int r = rec[i];
        a.rec = r; a.flag = bt->flag[r]; a.tid = bt->tid[r]; a.pos = bt->pos[r]; a.l_qseq = bt->l_qseq[r];
        a.n_cigar = bt->n_cigar[r];
        a.rev = (a.flag & SPX_FREVERSE) != 0;
        a.cigar = bt->cigar + bt->cigar_off[r];
        a.seq4 = bt->seq4 + bt->seq_off[r];
        a.qual = bt->qual + bt->qual_off[r];
        a.cs = bt->cs_off[r] >= 0 ? bt->cs + bt->cs_off[r] : nullptr;
        a.md = (bt->md_off && bt->md && bt->md_off[r] >= 0) ? bt->md + bt->md_off[r] : nullptr;
        a.conf.clear(); a.flank.clear(); a.have_conf = false;
        if (a.n_cigar <= 0) return SPX_EINVAL;
        if (a.tid < 0 || (size_t)a.tid >= ref.len.size()) return SPX_EINVAL;
        int rc = build_ops(a);
        if (rc) return rc;
        aligned_extents(a);
        if (a.flag & SPX_FSECONDARY) sec |= (uint16_t)(1u << i);
    }
    collect_markers(S.al, par->min_q, S.mk, S.mtmp);
    /* snapshot of the state this group appends to, so a failing group leaves nothing behind */
    const size_t s_prob = out.L.size(), s_rows = out.rows.size(), s_q = out.qry4.size(), s_hmm = out.hmm.size(),
                 s_qe = out.qe_rec.size();
    const int64_t s_nib = out.qry_nibbles;
    int n_prob = 0;
    int64_t cells = 0;
    bool scored = false;
    if (!S.mk.empty()) {
        int margin = par->flank_margin, nblk = 1 /* see DESIGN.md U1 */, iter = 0;
        for (Aln &a : S.al) confident_blocks(a, par->indel_threshold);
        while (par->consensus && blocks_too_long(S.al, 1000)) {
            margin = (int)(margin * 0.8);
            for (Aln &a : S.al) flank_blocks(a, S.mk, margin);
            nblk = consensus_blocks(S.al, par->indel_threshold, S.b1, S.b2);
            if (nblk == 0) break;
            if (++iter >= 64) break;
        }
        if (nblk > 0 || !par->consensus) {
            scored = true;
            if (par->baq_flag) {
                for (int i = 0; i < n; ++i) {
                    int rc = plan_baq(S.al[i], i, S.mk, S.mk, ref, par, S, out, n_prob, cells);
                    if (rc) {
                        out.ref_nib.resize(s_prob); out.ref_tid.resize(s_prob); out.ref_rfs.resize(s_prob);
                        out.qry_nib.resize(s_prob); out.L.resize(s_prob);
                        out.R.resize(s_prob); out.bw.resize(s_prob); out.row_off.resize(s_prob);
                        out.n_rows.resize(s_prob); out.hmm.resize(s_hmm); out.rows.resize(s_rows);
                        out.row_expect.resize(s_rows); out.row_rawq.resize(s_rows); out.qry4.resize(s_q);
                        out.qry_nibbles = s_nib;
                        /* the quality edits of the failing group go too: their row indices are about to be reused */
                        out.qe_rec.resize(s_qe); out.qe_pos.resize(s_qe); out.qe_len.resize(s_qe);
                        out.qe_row0.resize(s_qe); out.qe_batch.resize(s_qe);
                        return rc;
                    }
                }
            }
        }
    }
    /* marker table for the scoring kernel (empty when the group is not scored) */
    out.grp_index.push_back(g);
    if (out.mk_first.empty()) out.mk_first.push_back(0);
    if (scored) {
        for (size_t k = 0; k < S.mk.size(); ++k) {
            const Mk &m = S.mk[k];
            spx_dev_marker dm;
            dm.row = m.row;
            dm.qfix = (uint8_t)m.q;
            dm.is_match = (uint8_t)m.is_match;
            dm.aln = (uint8_t)m.aln;
            dm.first_of_pos = (k == 0 || S.mk[k - 1].pos != m.pos) ? (uint8_t)n : 0;
            out.markers.push_back(dm);
            out.mk_ref_pos.push_back(m.ref_pos);
        }
    }
    out.mk_first.push_back((int32_t)out.markers.size());
    out.n_aln.push_back((uint8_t)n);
    out.sec_mask.push_back(sec);
    for (int i = 0; i < 10; ++i) {
        out.rfe.push_back(i < n ? S.al[i].rfe : 0);
        out.rfs.push_back(i < n ? S.al[i].rfs : 0);
        out.atid.push_back(i < n ? S.al[i].tid : -1);
    }
    out.grp_problems.push_back(n_prob);
    out.grp_cells.push_back(cells);
    out.dp_cells += cells;
    return 0;
}

int prepare_groups(const spx_batch *bt, const RefIndex &ref, const spx_params *par, int32_t g0, int32_t g1,
                   HostBatch &out)
{
    GroupScratch S;
    out.clear();
    out.grp_error.assign(g1 - g0, 0);
    out.mk_first.push_back(0);
    for (int g = g0; g < g1; ++g) {
        if (!dispatched(bt, g)) { out.grp_error[g - g0] = 1; continue; }
        int rc = prepare_one(bt, ref, par, g, S, out);
        if (rc) out.grp_error[g - g0] = rc;
    }
    return 0;
}

} // namespace spx

extern "C" int spx_group_is_dispatched(const spx_batch *bt, int32_t g) { return spx::dispatched(bt, g) ? 1 : 0; }
