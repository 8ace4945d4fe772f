/*
 * spx_prep.cpp -- host side of the work-list preparation.
 *
 * The group logic itself (control flow of A10) lives in spx_logic.h and runs on the
 * DEVICE in the product path (spx_prep_kernels.hip).  What stays on the host:
 *   - the dispatch filter (src/secphase.c:285-288) and the staging of the dispatched groups' records into one packed,
 *     pinned buffer (one memcpy per payload array and alignment, on threads);
 *   - the host-only plan (spx_plan_create): the same spx_logic.h passes executed on the CPU, which is what the CPU
 *     tests compare with the oracle;
 *   - tables that need libm (phred thresholds, score tables, the float quality LUT entry).
 */
#include "spx_prep.h"
#include "spx_cpuacc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>

namespace spx {

static const unsigned char kNt16Int[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};

void RefIndex::build(const spx_ref *ref)
{
    const int nc = ref->n_contigs;
    nib_off.assign(nc, 0);
    len.assign(nc, 0);
    int64_t nib = kRefLeadNibbles; /* leading pad: the kernels fetch codes up to a band width before a window */
    for (int i = 0; i < nc; ++i) {
        nib_off[i] = nib;
        len[i] = ref->seq_off[i + 1] - ref->seq_off[i];
        nib += (len[i] + 1) & ~(int64_t)1; /* every contig starts on a byte boundary */
    }
    /* positions of the non-ACGT bases: the contigs are scanned in pieces on threads, the pieces' lists joined in order */
    npos_off.assign(nc + 1, 0);
    npos.clear();
    struct Piece { int contig; int64_t k0, k1; std::vector<int32_t> pos; };
    std::vector<Piece> pieces;
    const int64_t step = (int64_t)4 << 20;
    for (int i = 0; i < nc; ++i)
        for (int64_t k = 0; k < len[i] || (k == 0 && len[i] == 0); k += step) {
            pieces.push_back({i, k, std::min(len[i], k + step), {}});
            if (len[i] == 0) break;
        }
    {
        std::atomic<size_t> next(0);
        auto work = [&]() {
            static const struct Tbl { bool acgt[256]; Tbl() { for (int c = 0; c < 256; ++c) { const char u = (char)(c & ~0x20); acgt[c] = u == 'A' || u == 'C' || u == 'G' || u == 'T'; } } } T;
            for (;;) {
                const size_t q = next.fetch_add(1);
                if (q >= pieces.size()) break;
                Piece &pc = pieces[q];
                const unsigned char *s_ = (const unsigned char *)ref->bases + ref->seq_off[pc.contig];
                for (int64_t k = pc.k0; k < pc.k1; ++k)
                    if (!T.acgt[s_[k]]) pc.pos.push_back((int32_t)k);
            }
        };
        const unsigned nthr = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(32, std::thread::hardware_concurrency()), pieces.size()));
        if (nthr <= 1) work();
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nthr; ++t) th.emplace_back(work);
            for (auto &t : th) t.join();
        }
    }
    for (const Piece &pc : pieces) {
        npos.insert(npos.end(), pc.pos.begin(), pc.pos.end());
        npos_off[pc.contig + 1] = (int64_t)npos.size();
    }
    for (int i = 0; i < nc; ++i) npos_off[i + 1] = std::max(npos_off[i + 1], npos_off[i]);
}

spxl::RefView RefIndex::view() const
{
    spxl::RefView v;
    v.n_contigs = (int32_t)nib_off.size();
    v.nib_off = nib_off.data();
    v.len = len.data();
    v.npos_off = npos_off.data();
    v.npos = npos.data();
    return v;
}

void hmm_constants(int l_ref, int l_query, float d, float e, int set_q, double *h)
{
    const float qf = (float)pow(10, -set_q / 10.); /* htslib's qual2prob table entry, a float */
    spxl::hmm_constants(l_ref, l_query, d, e, qf, h);
}

/* the reading of probaln_glocal's terminal guard (spx_logic.h terminal_drop): process-wide, SPX_TERMINAL_GUARD=band|row
 * (or 0|1) at the first use, spx_set_terminal_guard() afterwards.  Work lists prepared earlier keep the reading they were built with. */
static std::atomic<int> g_term_guard{-1};
int terminal_guard()
{
    int v = g_term_guard.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("SPX_TERMINAL_GUARD");
        v = (e && (!strcmp(e, "row") || !strcmp(e, "1") || !strcmp(e, "idim"))) ? 1 : 0;
        g_term_guard.store(v, std::memory_order_relaxed);
    }
    return v;
}
void set_terminal_guard(int reading) { g_term_guard.store(reading ? 1 : 0, std::memory_order_relaxed); }

spxl::Params logic_params(const spx_params *par)
{
    spxl::Params p;
    memset(&p, 0, sizeof p);
    p.baq_flag = par->baq_flag;
    p.consensus = par->consensus;
    p.indel_threshold = par->indel_threshold;
    p.min_q = par->min_q;
    p.set_q = par->set_q;
    p.flank_margin = par->flank_margin;
    p.all_rows = (par->flags & SPX_PAR_ALL_ROWS) != 0;
    p.conf_b = par->conf_b;
    p.d = (float)par->conf_d;
    p.e = (float)par->conf_e;
    p.qf = (float)pow(10, -par->set_q / 10.);
    p.term_guard = terminal_guard();
    /* windows of the consensus rounds from the break-round table (spx_logic.h flank_break_rounds): groups of >= 1024 columns, 16 rounds;
     * SPX_WINDOW_TABLE=min_cols,rounds for the CPU tests (a short table hands over to the walk in mid-group; rounds 0 = the walk only) */
    static const std::pair<int, int> table = [] {
        int mc = spxl::kTableMinCols, r = spxl::kTableRounds;
        if (const char *e = getenv("SPX_WINDOW_TABLE")) { if (sscanf(e, "%d,%d", &mc, &r) < 2) r = spxl::kTableRounds; }
        return std::make_pair(mc < 2 ? 2 : mc, r < 0 ? 0 : (r > spxl::kTableRounds ? spxl::kTableRounds : r));
    }();
    p.row_mult = 2; /* (spx_prepare_staged raises it to 4 for the two-tier DP) */
    p.table_min_cols = table.first;
    p.table_rounds = table.second;
    return p;
}

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

/* ---------------- dispatch filter ---------------- */
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

bool group_dispatched(const spx_batch *bt, int g)
{
    int rec[16], n = group_records(bt, g, rec), supp = 0, prim = 0;
    for (int i = 0; i < n; ++i) {
        if (bt->flag[rec[i]] & SPX_FSUPPLEMENTARY) ++supp;
        if (!(bt->flag[rec[i]] & SPX_FSECONDARY)) ++prim;
    }
    return n > 1 && n <= 10 && supp == 0 && prim == 1;
}

bool recode_seq(const uint8_t *src, uint8_t *dst, int64_t n_bytes)
{
    bool has_n = false;
    for (int64_t k = 0; k < n_bytes; ++k) {
        const unsigned a = kNt16Int[src[k] >> 4], b = kNt16Int[src[k] & 0xf];
        dst[k] = (uint8_t)(a | (b << 4));
        has_n |= (a | b) > 3;
    }
    return has_n;
}

/* ---------------- staging ---------------- */
template <class F>
static void parallel_for(int64_t n, int threads, F f)
{
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n));
    if (threads == 1) { f(0, n); return; }
    std::vector<std::thread> th;
    std::atomic<int64_t> next(0);
    const int64_t chunk = std::max<int64_t>(1, n / (threads * 8));
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&]() {
            for (;;) {
                const int64_t a = next.fetch_add(chunk);
                if (a >= n) break;
                f(a, std::min(n, a + chunk));
            }
        });
    for (auto &t : th) t.join();
}

int stage_measure(const spx_batch *const *bts, int32_t n_batches, int threads, Stage &st)
{
    CpuScope cs_serial(CPU_MEASURE);
    st.batches.assign(bts, bts + n_batches);
    st.batch_base.assign(n_batches, 0);
    int64_t n_in = 0;
    for (int b = 0; b < n_batches; ++b) {
        if (!bts[b]) return SPX_EINVAL;
        st.batch_base[b] = (int32_t)n_in;
        n_in += bts[b]->n_groups;
    }
    if (n_in > 0x7fffffff) return SPX_EINVAL;
    st.grp_error.assign((size_t)n_in, 1);
    st.grp_index.clear();
    st.slot0.assign(1, 0);
    st.recs.clear();
    for (int b = 0; b < n_batches; ++b) {
        const spx_batch *bt = bts[b];
        for (int32_t g = 0; g < bt->n_groups; ++g) {
            int rec[16];
            if (!group_dispatched(bt, g)) continue;
            const int n = group_records(bt, g, rec);
            st.grp_error[(size_t)st.batch_base[b] + g] = 0;
            const int32_t k = (int32_t)st.grp_index.size();
            st.grp_index.push_back(st.batch_base[b] + g);
            for (int i = 0; i < n; ++i) {
                spxl::Rec r;
                memset(&r, 0, sizeof r);
                const int a = rec[i];
                r.rec = a; r.batch = b; r.grp = k;
                r.flag = bt->flag[a]; r.tid = bt->tid[a]; r.pos = bt->pos[a]; r.l_qseq = bt->l_qseq[a];
                r.n_cigar = bt->n_cigar[a];
                r.cs_len = -1; r.md_len = -1;
                r.alias_slot = -1;
                st.recs.push_back(r);
            }
            st.slot0.push_back((int32_t)st.recs.size());
        }
    }
    const int64_t ns = (int64_t)st.recs.size();
    /* tag lengths: the only part that touches payload bytes */
    parallel_for(ns, threads, [&](int64_t a0, int64_t a1) {
        CpuScope cs(CPU_MEASURE);
        for (int64_t s = a0; s < a1; ++s) {
            spxl::Rec &r = st.recs[(size_t)s];
            const spx_batch *bt = bts[r.batch];
            if (bt->cs_off[r.rec] >= 0) r.cs_len = (int32_t)strlen(bt->cs + bt->cs_off[r.rec]);
            else if (bt->md_off && bt->md && bt->md_off[r.rec] >= 0) r.md_len = (int32_t)strlen(bt->md + bt->md_off[r.rec]);
        }
    });
    /* SEQ / QUAL of a record that repeat those of another record of its group (ptMarker.c:48,79 read them from every record; an aligner writes
     * the read once per record): same strand = the same bytes, other strand = reverse complement / reversed qualities,
     * each minus the record's hard clips.  Verified base by base here; such a record's SEQ / QUAL stay on the host. */
    const bool alias_on = !getenv("SPX_NO_ALIAS");
    parallel_for((int64_t)st.grp_index.size(), threads, [&](int64_t k0, int64_t k1) {
        CpuScope cs(CPU_MEASURE);
        for (int64_t k = k0; k < k1 && alias_on; ++k) {
            const int32_t s0 = st.slot0[(size_t)k], s1 = st.slot0[(size_t)k + 1];
            /* the source: the record that holds most of the read (the primary unless it is the hard-clipped one) */
            int32_t p = -1;
            for (int32_t q = s0; q < s1; ++q) {
                const spxl::Rec &rq = st.recs[(size_t)q];
                if (p < 0 || rq.l_qseq > st.recs[(size_t)p].l_qseq ||
                    (rq.l_qseq == st.recs[(size_t)p].l_qseq && !(rq.flag & SPX_FSECONDARY) && (st.recs[(size_t)p].flag & SPX_FSECONDARY)))
                    p = q;
            }
            if (p < 0) continue;
            const spxl::Rec &rp = st.recs[(size_t)p];
            const spx_batch *bp = bts[rp.batch];
            auto clips = [](const spx_batch *bt, const spxl::Rec &r, int &lc, int &rc) {
                lc = rc = 0;
                if (r.n_cigar <= 0) return;
                const uint32_t *cg = bt->cigar + bt->cigar_off[r.rec];
                if ((cg[0] & 0xf) == SPX_CHARD_CLIP) lc = (int)(cg[0] >> 4);
                if (r.n_cigar > 1 && (cg[r.n_cigar - 1] & 0xf) == SPX_CHARD_CLIP) rc = (int)(cg[r.n_cigar - 1] >> 4);
            };
            int lcp, rcp;
            clips(bp, rp, lcp, rcp);
            const int64_t Tp = (int64_t)rp.l_qseq + lcp + rcp;
            if (rp.l_qseq <= 0) continue;
            const uint8_t *sp = bp->seq4 + bp->seq_off[rp.rec], *qp = bp->qual + bp->qual_off[rp.rec];
            auto base_of = [](const uint8_t *sq, int64_t i) -> unsigned { return (sq[i >> 1] >> ((~i & 1) << 2)) & 0xf; };
            static const uint8_t comp16[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15}; /* nt16 complement = the nibble's bits reversed */
            auto ld64 = [](const uint8_t *q) -> uint64_t { uint64_t v; memcpy(&v, q, 8); return v; };
            /* 16 bases from base j of a BAM SEQ as one word, base j in the top nibble (reads the byte of base j + 15) */
            auto nib16 = [&](const uint8_t *sq, int64_t j) -> uint64_t {
                const uint64_t v = __builtin_bswap64(ld64(sq + (j >> 1)));
                return (j & 1) ? (v << 4) | (uint64_t)(sq[(j >> 1) + 8] >> 4) : v;
            };
            for (int32_t q = s0; q < s1; ++q) {
                if (q == p) continue;
                spxl::Rec &rs = st.recs[(size_t)q];
                const spx_batch *bs = bts[rs.batch];
                int lcs, rcs;
                clips(bs, rs, lcs, rcs);
                const int64_t lq = rs.l_qseq;
                if (lq <= 0 || (int64_t)lq + lcs + rcs != Tp) continue;
                const bool rev = ((rs.flag ^ rp.flag) & SPX_FREVERSE) != 0;
                /* base i of the record sits at oriented read coordinate lcs + i; in the source's orientation that is the
                 * same coordinate, or T - 1 - it */
                const int64_t shift = rev ? Tp - 1 - lcs - lcp : (int64_t)lcs - lcp;
                const int64_t j0 = rev ? shift - (lq - 1) : shift, j1 = rev ? shift : shift + lq - 1;
                if (j0 < 0 || j1 >= rp.l_qseq) continue;
                const uint8_t *ss = bs->seq4 + bs->seq_off[rs.rec], *qs = bs->qual + bs->qual_off[rs.rec];
                /* whole words (no early exit: a difference is the rare case), the last few bases one by one */
                uint64_t diff = 0;
                int64_t i = 0;
                if (!rev) {
                    if (memcmp(qs, qp + shift, (size_t)lq) != 0) continue;
                    if ((shift & 1) == 0) {
                        diff = memcmp(ss, sp + (shift >> 1), (size_t)(lq >> 1)) != 0;
                        i = lq & ~(int64_t)1;
                    } else
                        for (; i + 16 <= lq; i += 16) diff |= ld64(ss + (i >> 1)) ^ __builtin_bswap64(nib16(sp, shift + i));
                    for (; i < lq && !diff; ++i) diff = base_of(ss, i) != base_of(sp, shift + i);
                } else {
                    for (; i + 8 <= lq; i += 8) diff |= __builtin_bswap64(ld64(qs + i)) ^ ld64(qp + shift - i - 7);
                    for (; i < lq; ++i) diff |= (uint64_t)(qs[i] ^ qp[shift - i]);
                    if (diff) continue;
                    /* bases i .. i+15 = the complements of the source's bases shift-i .. shift-i-15: the 16 nibbles in
                     * reverse order with the bits of each reversed = the 64-bit word bit-reversed */
                    for (i = 0; i + 16 <= lq; i += 16)
                        diff |= __builtin_bswap64(ld64(ss + (i >> 1))) ^ __builtin_bitreverse64(nib16(sp, shift - i - 15));
                    for (; i < lq && !diff; ++i) diff = base_of(ss, i) != comp16[base_of(sp, shift - i)];
                }
                if (diff) continue;
                rs.alias_slot = p;
                rs.alias_shift = (int32_t)shift;
                rs.alias_rev = rev ? 1 : 0;
            }
        }
    });
    StageLayout &L = st.lay;
    L = StageLayout();
    L.n_groups_in = n_in;
    L.n_dgroups = (int64_t)st.grp_index.size();
    L.n_slots = ns;
    int64_t cw = 0, sb = 0, qb = 0, tb = 0;
    for (int64_t s = 0; s < ns; ++s) {
        spxl::Rec &r = st.recs[(size_t)s];
        r.cigar_off = cw; cw += r.n_cigar > 0 ? r.n_cigar : 0;
        const int64_t lq = r.l_qseq > 0 ? r.l_qseq : 0;
        r.seq_off = sb; sb += (((lq + 1) / 2) + 3) & ~(int64_t)3;
        r.qual_off = qb; qb += lq;
        r.tag_off = tb; tb += (r.cs_len >= 0 ? r.cs_len : r.md_len >= 0 ? r.md_len : 0) + 1;
        r.pk_seq_off = L.pk_seq_bytes; r.pk_qual_off = L.pk_qual_bytes;
        if (r.alias_slot < 0) { L.pk_seq_bytes += (((lq + 1) / 2) + 3) & ~(int64_t)3; L.pk_qual_bytes += (lq + 3) & ~(int64_t)3; }
        else L.n_aliased++;
    }
    L.cigar_words = cw; L.seq_bytes = sb; L.qual_bytes = qb; L.text_bytes = tb;
    for (int64_t s = 0; s < ns; ++s) {
        int32_t oc, cc, mc;
        spxl::aln_caps(st.recs[(size_t)s], oc, cc, mc);
        L.ops_bound += oc; L.conf_bound += cc; L.mm_bound += mc;
    }
    stage_layout_offsets(L);
    return SPX_OK;
}

void stage_layout_offsets(StageLayout &L)
{
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
    L.o_recs = take((size_t)L.n_slots * sizeof(spxl::Rec));
    L.o_slot0 = take(((size_t)L.n_dgroups + 1) * 4);
    L.o_gidx = take((size_t)L.n_dgroups * 4);
    L.o_cigar = take((size_t)L.cigar_words * 4 + 16);
    L.o_seq = take((size_t)L.seq_bytes + 16);
    L.o_qual = take((size_t)L.qual_bytes + 16);
    L.o_text = take((size_t)L.text_bytes + 16);
    L.bytes = o;
}

void stage_copy(const Stage &st, char *dst, int threads)
{
    const StageLayout &L = st.lay;
    if (!st.recs.empty()) memcpy(dst + L.o_recs, st.recs.data(), st.recs.size() * sizeof(spxl::Rec));
    memcpy(dst + L.o_slot0, st.slot0.data(), st.slot0.size() * 4);
    if (!st.grp_index.empty()) memcpy(dst + L.o_gidx, st.grp_index.data(), st.grp_index.size() * 4);
    uint32_t *cig = (uint32_t *)(dst + L.o_cigar);
    uint8_t *seq = (uint8_t *)(dst + L.o_seq), *qual = (uint8_t *)(dst + L.o_qual);
    char *text = dst + L.o_text;
    parallel_for(L.n_slots, threads, [&](int64_t a0, int64_t a1) {
        for (int64_t s = a0; s < a1; ++s) {
            const spxl::Rec &r = st.recs[(size_t)s];
            const spx_batch *bt = st.batches[r.batch];
            if (r.n_cigar > 0) memcpy(cig + r.cigar_off, bt->cigar + bt->cigar_off[r.rec], (size_t)r.n_cigar * 4);
            const int64_t lq = r.l_qseq > 0 ? r.l_qseq : 0, nb = (lq + 1) / 2, pad = ((nb + 3) & ~(int64_t)3) - nb;
            if (nb) memcpy(seq + r.seq_off, bt->seq4 + bt->seq_off[r.rec], (size_t)nb);
            if (pad) memset(seq + r.seq_off + nb, 0, (size_t)pad);
            if (lq) memcpy(qual + r.qual_off, bt->qual + bt->qual_off[r.rec], (size_t)lq);
            if (r.cs_len >= 0) memcpy(text + r.tag_off, bt->cs + bt->cs_off[r.rec], (size_t)r.cs_len + 1);
            else if (r.md_len >= 0) memcpy(text + r.tag_off, bt->md + bt->md_off[r.rec], (size_t)r.md_len + 1);
            else text[r.tag_off] = 0;
        }
    });
}

/* ---------------- piecewise staging ---------------- */
int64_t stage_section_bytes(const Stage &st, int sec)
{
    const StageLayout &L = st.lay;
    return sec == 0 ? L.cigar_words * 4 : sec == 1 ? L.seq_bytes : sec == 2 ? L.qual_bytes : sec == 3 ? L.text_bytes : sec == 4 ? L.pk_seq_bytes : L.pk_qual_bytes;
}
size_t stage_section_offset(const Stage &st, int sec)
{
    const StageLayout &L = st.lay;
    return sec == 0 ? L.o_cigar : sec == 1 ? L.o_seq : sec == 2 ? L.o_qual : L.o_text;
}

namespace {
/* section-relative byte offset / stored length / source of record r's part */
inline int64_t sec_off(const spxl::Rec &r, int sec)
{
    return sec == 0 ? r.cigar_off * 4 : sec == 1 ? r.seq_off : sec == 2 ? r.qual_off : sec == 3 ? r.tag_off : sec == 4 ? r.pk_seq_off : r.pk_qual_off;
}
inline int64_t sec_len(const spxl::Rec &r, int sec)
{
    const int64_t lq = r.l_qseq > 0 ? r.l_qseq : 0;
    switch (sec) {
    case 0: return (int64_t)(r.n_cigar > 0 ? r.n_cigar : 0) * 4;
    case 1: return (((lq + 1) / 2) + 3) & ~(int64_t)3;
    case 2: return lq;
    case 3: return (r.cs_len >= 0 ? r.cs_len : r.md_len >= 0 ? r.md_len : 0) + 1;
    case 4: return r.alias_slot >= 0 ? 0 : (((lq + 1) / 2) + 3) & ~(int64_t)3;
    default: return r.alias_slot >= 0 ? 0 : (lq + 3) & ~(int64_t)3;
    }
}
} // namespace

void stage_fill(const Stage &st, int sec, int64_t b0, int64_t b1, char *dst,
                const std::function<void(int64_t, int64_t, const std::function<void(int64_t, int64_t)> &)> &f_parallel)
{
    const int64_t ns = (int64_t)st.recs.size();
    if (b1 <= b0 || ns == 0) return;
    /* first record whose part ends behind b0 (offsets ascend with the record index) */
    int64_t lo = 0, hi = ns;
    while (lo < hi) {
        const int64_t m = (lo + hi) / 2;
        if (sec_off(st.recs[(size_t)m], sec) + sec_len(st.recs[(size_t)m], sec) <= b0) lo = m + 1; else hi = m;
    }
    const int64_t s0 = lo;
    lo = s0; hi = ns;
    while (lo < hi) { /* first record that starts at or behind b1 */
        const int64_t m = (lo + hi) / 2;
        if (sec_off(st.recs[(size_t)m], sec) < b1) lo = m + 1; else hi = m;
    }
    const int64_t s1 = lo;
    if (s1 <= s0) return;
    const int64_t grain = std::max<int64_t>(1, (int64_t)(((int64_t)4 << 20) / std::max<int64_t>(1, (b1 - b0) / (s1 - s0)))); /* ~4 MB per piece */
    f_parallel(s1 - s0, grain, [&](int64_t k0, int64_t k1) {
        CpuScope cs(CPU_FILL);
        for (int64_t s = s0 + k0; s < s0 + k1; ++s) {
            const spxl::Rec &r = st.recs[(size_t)s];
            const spx_batch *bt = st.batches[r.batch];
            const int64_t off = sec_off(r, sec), len = sec_len(r, sec);
            const int64_t a = std::max(off, b0), e = std::min(off + len, b1); /* the part of this record inside the range */
            if (e <= a) continue;
            char *d = dst + (a - b0);
            const int64_t skip = a - off, n = e - a;
            const char *src = nullptr;
            int64_t have = 0; /* bytes the record really has (the rest of `len` is padding / the terminator) */
            switch (sec) {
            case 0: src = (const char *)(bt->cigar + bt->cigar_off[r.rec]); have = len; break;
            case 1: case 4: src = (const char *)(bt->seq4 + bt->seq_off[r.rec]); have = ((r.l_qseq > 0 ? r.l_qseq : 0) + 1) / 2; break;
            case 2: src = (const char *)(bt->qual + bt->qual_off[r.rec]); have = len; break;
            case 5: src = (const char *)(bt->qual + bt->qual_off[r.rec]); have = r.l_qseq > 0 ? r.l_qseq : 0; break;
            default:
                if (r.cs_len >= 0) { src = bt->cs + bt->cs_off[r.rec]; have = r.cs_len; }
                else if (r.md_len >= 0) { src = bt->md + bt->md_off[r.rec]; have = r.md_len; }
                break;
            }
            const int64_t c1 = std::min(skip + n, have); /* copy [skip, c1), zero the rest */
            if (c1 > skip) memcpy(d, src + skip, (size_t)(c1 - skip));
            const int64_t z0 = std::max(skip, have);
            if (skip + n > z0) memset(d + (z0 - skip), 0, (size_t)(skip + n - z0));
        }
    });
}

/* ---------------- host plan: the device passes, executed on the CPU ---------------- */
int host_plan(const spx_batch *const *bts, int32_t n_batches, const RefIndex &ref, const spx_params *par, int threads,
              HostBatch &hb)
{
    hb = HostBatch();
    Stage st;
    int rc = stage_measure(bts, n_batches, threads, st);
    if (rc) return rc;
    const StageLayout &L = st.lay;
    std::vector<char> buf(L.bytes + 64);
    stage_copy(st, buf.data(), threads);
    const spxl::Params lp = logic_params(par);
    const spxl::RefView rv = ref.view();
    const int64_t ns = L.n_slots, ng = L.n_dgroups;
    const spxl::Rec *recs = (const spxl::Rec *)(buf.data() + L.o_recs);
    /* recoded sequence pool (what the query windows point into) */
    std::vector<uint8_t> code((size_t)(kCodeLeadBytes + L.seq_bytes + kCodeTailBytes), 0);
    std::vector<spxl::AlnState> ast((size_t)ns);
    spxl::Pools P;
    memset(&P, 0, sizeof P);
    P.cigar = (const uint32_t *)(buf.data() + L.o_cigar);
    P.qual = (const uint8_t *)(buf.data() + L.o_qual);
    P.text = buf.data() + L.o_text;
    P.code4 = code.data();
    P.code_lead_bytes = kCodeLeadBytes;
    const uint8_t *raw = (const uint8_t *)(buf.data() + L.o_seq);
    /* pass A: recode + count */
    parallel_for(ns, threads, [&](int64_t a0, int64_t a1) {
        for (int64_t s = a0; s < a1; ++s) {
            const spxl::Rec &r = recs[s];
            spxl::AlnState &a = ast[(size_t)s];
            memset(&a, 0, sizeof a);
            const int64_t lq = r.l_qseq > 0 ? r.l_qseq : 0;
            recode_seq(raw + r.seq_off, code.data() + kCodeLeadBytes + r.seq_off, (lq + 1) / 2);
            bool hn = false;
            for (int64_t k = 0; k < lq && !hn; ++k)
                hn = ((code[(size_t)(kCodeLeadBytes + r.seq_off + (k >> 1))] >> ((k & 1) << 2)) & 0xf) > 3;
            a.has_n = hn;
            a.err = spxl::build_ops<false>(r, P, lp.min_q, lp.indel_threshold, a, nullptr);
            if (a.err) { a.n_ops = 0; a.mm_cap = 0; a.conf_cap = 0; }
        }
    });
    int64_t n_ops = 0, n_conf = 0, n_mm = 0;
    for (int64_t s = 0; s < ns; ++s) {
        spxl::AlnState &a = ast[(size_t)s];
        a.ops_off = n_ops; n_ops += a.n_ops;
        a.conf_off = n_conf; n_conf += a.conf_cap;
        a.mm_off = n_mm; n_mm += a.mm_cap;
    }
    std::vector<spxl::Op> ops((size_t)n_ops + 1);
    std::vector<spxl::Blk> conf((size_t)n_conf + 1);
    std::vector<spxl::MM> mm((size_t)n_mm + 1);
    /* SPX_POISON=1 (tests): the device's pools and arenas are recycled, never zeroed -- give the host plan the same
     * conditions, so that a read of something never written shows up on the CPU too */
    const bool poison = getenv("SPX_POISON") != nullptr;
    if (poison) {
        memset((void *)ops.data(), 0xA5, ops.size() * sizeof(spxl::Op));
        memset((void *)conf.data(), 0xA5, conf.size() * sizeof(spxl::Blk));
        memset((void *)mm.data(), 0xA5, mm.size() * sizeof(spxl::MM));
    }
    P.ops = ops.data(); P.conf = conf.data(); P.mm = mm.data();
    /* pass B: op tables, extents, confident blocks, mismatch lists */
    parallel_for(ns, threads, [&](int64_t a0, int64_t a1) {
        for (int64_t s = a0; s < a1; ++s) {
            const spxl::Rec &r = recs[s];
            spxl::AlnState &a = ast[(size_t)s];
            if (a.err) continue;
            a.err = spxl::build_ops<true>(r, P, lp.min_q, lp.indel_threshold, a, ops.data() + a.ops_off);
            if (!a.err) a.err = spxl::finish_alignment(r, P, lp.min_q, lp.indel_threshold, a, ops.data() + a.ops_off,
                                                      conf.data() + a.conf_off, mm.data() + a.mm_off);
        }
    });
    /* the group / alignment passes, in the order the device runs them (scratch kept between them) */
    std::vector<spxl::GroupCount> gc((size_t)ng), ac((size_t)ns);
    std::vector<spxl::GroupArena> ga((size_t)ng);
    std::vector<int64_t> ga_off((size_t)ng + 1, 0);
    std::vector<int32_t> slot_grp((size_t)ns);
    for (int64_t k = 0; k < ng; ++k)
        for (int32_t q = st.slot0[(size_t)k]; q < st.slot0[(size_t)k + 1]; ++q) slot_grp[(size_t)q] = (int32_t)k;
    auto view = [&](int64_t k) {
        spxl::GroupView G = {st.slot0[(size_t)k + 1] - st.slot0[(size_t)k], recs + st.slot0[(size_t)k], ast.data() + st.slot0[(size_t)k]};
        return G;
    };
    int slack = 1;
    static const int plan_parts = [] { const char *e = getenv("SPX_PLAN_PARTS"); return e ? std::max(1, atoi(e)) : 1; }();
    std::vector<spxl::PlanBase> part_add(plan_parts > 1 ? (size_t)ns * (size_t)plan_parts : 0);
    std::vector<char> arena;
    for (;;) {
        for (int64_t k = 0; k < ng; ++k) {
            ga[(size_t)k] = spxl::group_arena_layout(view(k), lp.all_rows != 0, slack);
            ga_off[(size_t)k + 1] = ga_off[(size_t)k] + ga[(size_t)k].bytes;
        }
        arena.assign((size_t)ga_off[(size_t)ng] + 64, poison ? (char)0xA5 : (char)0);
        if (poison) {
            memset((void *)gc.data(), 0xA5, gc.size() * sizeof(spxl::GroupCount));
            memset((void *)ac.data(), 0xA5, ac.size() * sizeof(spxl::GroupCount));
        }
        auto scratch = [&](int64_t k) { return spxl::group_scratch(ga[(size_t)k], arena.data() + ga_off[(size_t)k]); };
        std::atomic<int> overflow(0);
        parallel_for(ng, threads, [&](int64_t k0, int64_t k1) { /* G1 */
            for (int64_t k = k0; k < k1; ++k) { spxl::GroupScratch S = scratch(k); spxl::group_pass_merge(view(k), P, rv, S, gc[(size_t)k]); }
        });
        parallel_for(ns, threads, [&](int64_t q0, int64_t q1) { /* A1 */
            for (int64_t q = q0; q < q1; ++q) {
                const int64_t k = slot_grp[(size_t)q];
                spxl::GroupScratch S = scratch(k);
                /* SPX_PLAN_PARTS=n (CPU tests): the columns in n contiguous shares, one after the other -- what the lanes of a wave that
                 * shares one heavy alignment do side by side on the device; the result must not depend on n */
                static const int parts = [] { const char *e = getenv("SPX_PLAN_PARTS"); return e ? std::max(1, atoi(e)) : 1; }();
                for (int pt = parts - 1; pt >= 0; --pt) /* (descending: no share may rely on an earlier one having run) */
                    spxl::aln_pass_filter(view(k), (int)(q - st.slot0[(size_t)k]), P, S, gc[(size_t)k], pt, parts);
            }
        });
        parallel_for(ns, threads, [&](int64_t q0, int64_t q1) { /* A1b */
            for (int64_t q = q0; q < q1; ++q) {
                const int64_t k = slot_grp[(size_t)q];
                spxl::GroupScratch S = scratch(k);
                spxl::aln_pass_compact(view(k), (int)(q - st.slot0[(size_t)k]), S, gc[(size_t)k]);
            }
        });
        parallel_for(ng, threads, [&](int64_t k0, int64_t k1) { /* G2 */
            for (int64_t k = k0; k < k1; ++k) {
                spxl::GroupScratch S = scratch(k);
                spxl::group_pass_blocks(view(k), P, lp, S, gc[(size_t)k]);
                if (gc[(size_t)k].err == SPX_ENOMEM) overflow = 1;
            }
        });
        if (!overflow) {
            parallel_for(ns, threads, [&](int64_t q0, int64_t q1) { /* A2 */
                for (int64_t q = q0; q < q1; ++q) {
                    const int64_t k = slot_grp[(size_t)q];
                    spxl::GroupScratch S = scratch(k);
                    const int ai = (int)(q - st.slot0[(size_t)k]);
                    if (plan_parts > 1 && spxl::plan_can_split(lp, view(k).st[ai])) {
                        /* (CPU tests, SPX_PLAN_PARTS: the blocks in shares, last share first; combined in share order: the first error wins) */
                        std::vector<spxl::GroupCount> pc((size_t)plan_parts);
                        for (int pt = plan_parts - 1; pt >= 0; --pt)
                            spxl::aln_pass_count_part(view(k), ai, P, rv, lp, S, gc[(size_t)k], pc[(size_t)pt], part_add[(size_t)q * plan_parts + pt], pt, plan_parts);
                        spxl::GroupCount &a = ac[(size_t)q];
                        spxl::count_clear(a);
                        for (int pt = 0; pt < plan_parts; ++pt) {
                            if (pc[(size_t)pt].err) { spxl::count_clear(a); a.err = pc[(size_t)pt].err; break; }
                            spxl::count_add(a, pc[(size_t)pt]);
                        }
                    } else
                        spxl::aln_pass_count(view(k), ai, P, rv, lp, S, gc[(size_t)k], ac[(size_t)q]);
                    if (ac[(size_t)q].err == SPX_ENOMEM) overflow = 1;
                }
            });
        }
        if (!overflow || slack >= 64) break;
        slack *= 4; /* an interval list outgrew its estimate: bigger scratch, same computation */
    }
    for (int64_t k = 0; k < ng; ++k) spxl::group_pass_sum(view(k), gc[(size_t)k], ac.data() + st.slot0[(size_t)k]); /* G3 */
    if (getenv("SPX_DEBUG_GC")) /* debugging aid: the same line spx_prepare_staged prints for the device */
        for (int64_t k = 0; k < ng; ++k) {
            fprintf(stderr, "[spx debug host] group %lld: err %d scored %d n_cols %d n_prob %d n_rows %d |", (long long)k, gc[(size_t)k].err, gc[(size_t)k].scored,
                    gc[(size_t)k].n_cols, gc[(size_t)k].n_prob, gc[(size_t)k].n_rows);
            for (int32_t q = st.slot0[(size_t)k]; q < st.slot0[(size_t)k + 1]; ++q)
                fprintf(stderr, " [ops %d visit %d conf %d mm %d err %d rds %d rde %d rfs %d rfe %d]", ast[(size_t)q].n_ops, ast[(size_t)q].n_visit, ast[(size_t)q].n_conf,
                        ast[(size_t)q].n_mm, ast[(size_t)q].err, ast[(size_t)q].rds, ast[(size_t)q].rde, ast[(size_t)q].rfs, ast[(size_t)q].rfe);
            fprintf(stderr, "\n");
        }
    /* offsets: errored groups are left out of the work list (their code goes to grp_error) */
    std::vector<spxl::PlanBase> base((size_t)ns + 1);
    std::vector<int64_t> mk_base((size_t)ng + 1, 0);
    std::vector<int32_t> okidx((size_t)ng, -1);
    spxl::PlanBase tot = {0, 0, 0, 0, 0};
    int32_t n_ok = 0;
    hb.grp_error = st.grp_error;
    for (int64_t k = 0; k < ng; ++k) {
        const spxl::GroupCount &c = gc[(size_t)k];
        mk_base[(size_t)k + 1] = mk_base[(size_t)k];
        for (int32_t q = st.slot0[(size_t)k]; q < st.slot0[(size_t)k + 1]; ++q) {
            const spxl::GroupCount &a = ac[(size_t)q];
            base[(size_t)q] = tot;
            tot.prob += a.n_prob; tot.row += a.n_rows; tot.qe += a.n_qe; tot.s_off += a.s_need; tot.f_off += a.f_need;
        }
        if (c.err) { hb.grp_error[(size_t)st.grp_index[(size_t)k]] = c.err; continue; }
        okidx[(size_t)k] = n_ok++;
        const int n = st.slot0[(size_t)k + 1] - st.slot0[(size_t)k];
        mk_base[(size_t)k + 1] += (int64_t)c.n_cols * n;
        hb.dp_cells += c.cells;
    }
    const size_t np = (size_t)tot.prob, nr = (size_t)tot.row, nq = (size_t)tot.qe, nm = (size_t)mk_base[(size_t)ng];
    hb.ref_nib.resize(np); hb.qry_nib.resize(np); hb.ref_tid.resize(np); hb.ref_rfs.resize(np); hb.L.resize(np); hb.R.resize(np);
    hb.bw.resize(np); hb.row_off.resize(np); hb.n_rows.resize(np); hb.hmm.resize(np * SPX_H_N);
    hb.rows.resize(nr); hb.row_expect.resize(nr); hb.row_rawq.resize(nr);
    hb.qe_rec.resize(nq); hb.qe_pos.resize(nq); hb.qe_len.resize(nq); hb.qe_row0.resize(nq); hb.qe_batch.resize(nq);
    hb.markers.resize(nm); hb.mk_ref_pos.resize(nm);
    hb.grp_index.resize((size_t)n_ok); hb.mk_first.assign((size_t)n_ok + 1, 0); hb.n_aln.resize((size_t)n_ok); hb.sec_mask.resize((size_t)n_ok);
    hb.rfe.assign((size_t)n_ok * 10, 0); hb.rfs.assign((size_t)n_ok * 10, 0); hb.atid.assign((size_t)n_ok * 10, -1);
    hb.grp_problems.resize((size_t)n_ok); hb.grp_cells.resize((size_t)n_ok);
    std::vector<int64_t> s_off(np), f_off(np);
    std::vector<int32_t> prob_slots(np), row_prob(nr);
    std::vector<uint8_t> has_n(np);
    spxl::PlanOut out;
    memset(&out, 0, sizeof out);
    out.ref_nib = hb.ref_nib.data(); out.qry_nib = hb.qry_nib.data(); out.ref_tid = hb.ref_tid.data(); out.ref_rfs = hb.ref_rfs.data();
    out.L = hb.L.data(); out.R = hb.R.data(); out.bw = hb.bw.data(); out.row_off = hb.row_off.data(); out.n_rows = hb.n_rows.data();
    out.prob_slots = prob_slots.data(); out.has_n = has_n.data(); out.s_off = s_off.data(); out.fsave_off = f_off.data();
    out.rows = hb.rows.data(); out.row_expect = hb.row_expect.data(); out.row_prob = row_prob.data(); out.row_rawq = hb.row_rawq.data();
    out.qe_rec = hb.qe_rec.data(); out.qe_pos = hb.qe_pos.data(); out.qe_len = hb.qe_len.data(); out.qe_row0 = hb.qe_row0.data();
    out.qe_batch = hb.qe_batch.data();
    auto scratch = [&](int64_t k) { return spxl::group_scratch(ga[(size_t)k], arena.data() + ga_off[(size_t)k]); };
    parallel_for(ns, threads, [&](int64_t q0, int64_t q1) { /* A3 */
        for (int64_t q = q0; q < q1; ++q) {
            const int64_t k = slot_grp[(size_t)q];
            spxl::GroupScratch S = scratch(k);
            const int ai = (int)(q - st.slot0[(size_t)k]);
            if (plan_parts > 1 && spxl::plan_can_split(lp, view(k).st[ai])) {
                for (int pt = plan_parts - 1; pt >= 0; --pt) {
                    spxl::PlanBase at = base[(size_t)q];
                    for (int j = 0; j < pt; ++j) {
                        const spxl::PlanBase &d = part_add[(size_t)q * plan_parts + j];
                        at.prob += d.prob; at.row += d.row; at.qe += d.qe; at.s_off += d.s_off; at.f_off += d.f_off;
                    }
                    spxl::aln_pass_emit_part(view(k), ai, P, rv, lp, S, gc[(size_t)k], at, out, pt, plan_parts);
                }
            } else
                spxl::aln_pass_emit(view(k), ai, P, rv, lp, S, gc[(size_t)k], base[(size_t)q], out);
        }
    });
    parallel_for(ng, threads, [&](int64_t k0, int64_t k1) { /* G4 + the per-group arrays */
        for (int64_t k = k0; k < k1; ++k) {
            const spxl::GroupCount &c = gc[(size_t)k];
            if (c.err) continue;
            const int32_t kk = okidx[(size_t)k];
            spxl::GroupView G = view(k);
            const int n = G.n;
            spxl::GroupScratch S = scratch(k);
            spxl::group_pass_markers(G, S, c, hb.markers.data() + mk_base[(size_t)k], hb.mk_ref_pos.data() + mk_base[(size_t)k]);
            hb.grp_index[(size_t)kk] = st.grp_index[(size_t)k];
            hb.mk_first[(size_t)kk + 1] = (int32_t)mk_base[(size_t)k + 1];
            hb.n_aln[(size_t)kk] = (uint8_t)n;
            uint16_t sec = 0;
            for (int i = 0; i < n; ++i) {
                if (G.rec[i].flag & SPX_FSECONDARY) sec |= (uint16_t)(1u << i);
                hb.rfe[(size_t)kk * 10 + i] = G.st[i].rfe;
                hb.rfs[(size_t)kk * 10 + i] = G.st[i].rfs;
                hb.atid[(size_t)kk * 10 + i] = G.rec[i].tid;
            }
            hb.sec_mask[(size_t)kk] = sec;
            hb.grp_problems[(size_t)kk] = c.n_prob;
            hb.grp_cells[(size_t)kk] = c.cells;
        }
    });
    parallel_for((int64_t)np, threads, [&](int64_t p0, int64_t p1) {
        for (int64_t q = p0; q < p1; ++q)
            spxl::problem_constants(lp, hb.L[(size_t)q], hb.R[(size_t)q], has_n[(size_t)q], hb.hmm.data() + (size_t)q * SPX_H_N);
    });
    hb.qry4.swap(code);
    hb.qry_nibbles = (int64_t)hb.qry4.size() * 2;
    return SPX_OK;
}

} // namespace spx

extern "C" int spx_group_is_dispatched(const spx_batch *bt, int32_t g) { return spx::group_dispatched(bt, g) ? 1 : 0; }
