/*
 * secphase_oracle.c -- TEST INFRASTRUCTURE ONLY (see secphase_oracle.h).
 *
 * CPU restatement of the marker branch of secphase's runOneThread
 * (/root/reference/programs/src/secphase.c:156-219) and everything it calls
 * in programs/submodules/{cigar_it,ptAlignment,ptMarker}.  Written against
 * the flat record format of include/spx_records.h instead of bam1_t/stList;
 * each function names the reference lines it follows.  PARITY UNPINNED (the
 * reference has no test for this path and cannot be built here).
 *
 * Decisions taken where the reference is undefined, all
 * documented in DESIGN.md:
 *   U1 conf_blocks_length read uninitialised when the consensus loop body
 *      never runs (secphase.c:107,164-170): treated as "> 0".
 *   U2 match-marker base_idx outside SEQ when the position lies in this
 *      alignment's hard clip (ptMarker.c:91-104): base_q := 0 (the position
 *      is dropped by filter_ins_markers right after, so the value is unused).
 *   U3 CIGAR ops N/P/B leave the step variables unset (cigar_it.c:225-291):
 *      the group is rejected with error -2.
 *   U4 (int) conversion of +inf/NaN in the BAQ phred: x86 cvttsd2si result.
 *   U5 the consensus loop is capped at 64 iterations (the reference would
 *      spin for ever if blocks stayed > 1000 bp at margin 0).
 */
#define _GNU_SOURCE
#include "secphase_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- htslib nt16 tables (hts.c), used at ptMarker.c:733,744 ---- */
const unsigned char orc_nt16_table[256] = {
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 1,  2,  4,  8,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 0,  15, 15, 15, 1,  14, 2,  13, 15, 15, 4,  11, 15, 15, 12, 15, 3,
    15, 15, 15, 15, 5,  6,  8,  15, 7,  9,  15, 10, 15, 15, 15, 15, 15, 15, 15, 1,  14, 2,  13, 15, 15, 4,
    11, 15, 15, 12, 15, 3,  15, 15, 15, 15, 5,  6,  8,  15, 7,  9,  15, 10, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15};
const unsigned char orc_nt16_int[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};

static inline int imin(int a, int b) { return a < b ? a : b; } /* common.c:28-30 */
static inline int imax(int a, int b) { return b < a ? a : b; } /* common.c:32-34 */

/* ---- glibc rand() replay ------------------------------------------------ */
void orc_srand(orc_rand *st, unsigned seed)
{
    /* glibc random_r.c TYPE_3: r[i] = 16807*r[i-1] mod (2^31-1), 31 words,
     * then 310 outputs discarded */
    int32_t word;
    int i;
    if (seed == 0) seed = 1;
    st->tbl[0] = seed;
    word = (int32_t)seed;
    for (i = 1; i < 31; ++i) {
        long hi = word / 127773, lo = word % 127773;
        word = (int32_t)(16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        st->tbl[i] = (uint32_t)word;
    }
    st->f = 3;
    st->b = 0;
    for (i = 0; i < 310; ++i) (void)orc_rand_next(st);
}

int orc_rand_next(orc_rand *st)
{
    uint32_t v;
    st->tbl[st->f] += st->tbl[st->b];
    v = st->tbl[st->f] >> 1;
    if (++st->f >= 31) st->f = 0;
    if (++st->b >= 31) st->b = 0;
    return (int)v;
}

/* ---- per-alignment working copy ----------------------------------------- */
typedef struct {
    int flag, tid, pos, l_qseq, n_cigar, is_rev;
    const uint32_t *cigar;
    const uint8_t *seq4;
    uint8_t *qual; /* private, mutable */
    const char *cs;
    const char *md; /* used only when cs is absent (cigar_it.c:46-63) */
    orc_op *ops; /* ops[0] = start state */
    int n_ops;
    int lclip, rclip;
    int rfs, rfe, rds_f, rde_f; /* ptAlignment.h:26-36 */
    double score;
    orc_block *conf;
    int n_conf;
    orc_block *flank;
    int n_flank;
} waln;

/* ---- cs tokenizer: what regexec(CS_PATTERN) finds (cigar_it.h:8, cigar_it.c:145-211).
 * Un-anchored leftmost-longest search for  :[0-9]+ | [+-][a-z]+ | (\*[a-z]+)+  */
static int is_lower(char c) { return c >= 'a' && c <= 'z'; }
static int is_digit(char c) { return c >= '0' && c <= '9'; }

static int cs_next_token(const char *s, int *so, int *eo)
{
    int p;
    for (p = 0; s[p]; ++p) {
        char c = s[p];
        if (c == ':' && is_digit(s[p + 1])) {
            int e = p + 1;
            while (is_digit(s[e])) ++e;
            *so = p; *eo = e;
            return 1;
        }
        if ((c == '+' || c == '-') && is_lower(s[p + 1])) {
            int e = p + 1;
            while (is_lower(s[e])) ++e;
            *so = p; *eo = e;
            return 1;
        }
        if (c == '*' && is_lower(s[p + 1])) {
            int e = p;
            while (s[e] == '*' && is_lower(s[e + 1])) {
                e += 1;
                while (is_lower(s[e])) ++e;
            }
            *so = p; *eo = e;
            return 1;
        }
    }
    return 0;
}

/* MD tokenizer: what regexec(MD_PATTERN) finds (cigar_it.h:10): un-anchored leftmost-longest search for
 *   [A-Z]([0][A-Z])*  |  [0-9]+  |  \^[A-Z]+                                                       */
static int is_upper(char c) { return c >= 'A' && c <= 'Z'; }
static int md_next_token(const char *s, int *so, int *eo)
{
    int p;
    for (p = 0; s[p]; ++p) {
        char c = s[p];
        if (is_upper(c)) {
            int e = p + 1;
            while (s[e] == '0' && is_upper(s[e + 1])) e += 2;
            *so = p; *eo = e;
            return 1;
        }
        if (is_digit(c)) {
            int e = p + 1;
            while (is_digit(s[e])) ++e;
            *so = p; *eo = e;
            return 1;
        }
        if (c == '^' && is_upper(s[p + 1])) {
            int e = p + 1;
            while (is_upper(s[e])) ++e;
            *so = p; *eo = e;
            return 1;
        }
    }
    return 0;
}

/* ptCigarIt_next_md (cigar_it.c:72-141), including its recursion on zero-length tokens */
static int next_md(const char *md, int *off, orc_op *cur)
{
    const char *s = md + *off;
    int so, eo;
    char c, buf[24];
    if (!md_next_token(s, &so, &eo)) return 0;
    c = s[so];
    if (c == '0') { cur->op = SPX_CDIFF; cur->len = 0; }
    else if (c <= '9') {
        int n = eo - so; /* the reference copies (eo-so) bytes from the START of the shifted string */
        if (n > 19) n = 19;
        memcpy(buf, s, n);
        buf[n] = 0;
        cur->op = SPX_CEQUAL;
        cur->len = atoi(buf);
    } else if (c < 90) { cur->op = SPX_CDIFF; cur->len = 1 + (eo - so - 1) / 2; }
    else if (c == 94) { cur->op = SPX_CDEL; cur->len = eo - so - 1; }
    *off += eo;
    if (cur->len == 0) next_md(md, off, cur);
    return cur->len;
}

/* ptCigarIt_construct + repeated ptCigarIt_next (cigar_it.c:14-69,213-308),
 * materialised once: the reference re-creates the iterator >=7 times per
 * alignment and always sees the same sequence of states. */
static int is_mx(int op);

static int walk_cigar(waln *a)
{
    int cap = a->n_cigar * 2 + 16, n = 0, idx = -1, match_remain = 0, cs_off = 0;
    orc_op cur;
    orc_op *ops = malloc(sizeof(orc_op) * cap);
    a->lclip = a->rclip = 0;
    if ((a->cigar[0] & 0xf) == SPX_CHARD_CLIP) a->lclip = a->cigar[0] >> 4;
    if ((a->cigar[a->n_cigar - 1] & 0xf) == SPX_CHARD_CLIP) a->rclip = a->cigar[a->n_cigar - 1] >> 4;
    cur.op = 255; /* (uint8_t)-1 */
    cur.len = 0;
    cur.ret = 0;
    cur.sqs = 0;
    cur.sqe = -1;
    cur.rfs = a->pos;
    cur.rfe = a->pos - 1;
    cur.rds_f = a->is_rev ? a->rclip + a->lclip + a->l_qseq : 0;
    cur.rde_f = a->is_rev ? a->rclip + a->lclip + a->l_qseq - 1 : -1;
    ops[n++] = cur;
    const int use_cs = a->cs != NULL, use_md = !use_cs && a->md != NULL;
    int md_off = 0;
    if (!use_cs && !use_md) { free(ops); return -3; } /* the reference exit(1)s: neither tag present */
    while (idx != a->n_cigar - 1) {
        int op, len, rd_step, sq_step, rf_step, so, eo, have;
        idx += 1;
        op = a->cigar[idx] & 0xf;
        len = a->cigar[idx] >> 4;
        have = 0;
        if (use_cs && (op == SPX_CMATCH || op == SPX_CEQUAL || op == SPX_CDIFF || op == SPX_CINS || op == SPX_CDEL)) {
            /* ptCigarIt_next_cs */
            const char *s = a->cs + cs_off;
            have = cs_next_token(s, &so, &eo);
            if (have) {
                switch (s[so]) {
                case ':': cur.op = SPX_CEQUAL; cur.len = atoi(s + so + 1); break;
                case '*': cur.op = SPX_CDIFF; cur.len = (eo - so + 1) / 3; break;
                case '+': cur.op = SPX_CINS; cur.len = eo - so - 1; break;
                case '-': cur.op = SPX_CDEL; cur.len = eo - so - 1; break;
                }
                cs_off += eo;
            }
        }
        switch (op) {
        case SPX_CMATCH:
        case SPX_CEQUAL:
        case SPX_CDIFF:
            if (match_remain == 0) match_remain = len;
            if (use_cs) {
                match_remain -= cur.len;
                if (0 < match_remain) idx -= 1;
            } else { /* MD tokens do not see insertions, so they may span several CIGAR M ops (cigar_it.c:237-256) */
                if (0 <= match_remain) next_md(a->md, &md_off, &cur);
                if (match_remain < 0) {
                    cur.op = SPX_CEQUAL;
                    cur.len = imin(len, -1 * match_remain);
                    match_remain += len;
                } else {
                    int md_len = cur.len;
                    cur.len = imin(cur.len, match_remain);
                    match_remain -= md_len;
                }
                if (0 < match_remain) idx -= 1;
            }
            rd_step = sq_step = rf_step = cur.len;
            break;
        case SPX_CINS:
            cur.len = len; cur.op = op;
            rd_step = sq_step = len; rf_step = 0;
            break;
        case SPX_CDEL:
            if (use_md) next_md(a->md, &md_off, &cur);
            rd_step = sq_step = 0; rf_step = len;
            break;
        case SPX_CSOFT_CLIP:
            cur.len = len; cur.op = op;
            rd_step = sq_step = len; rf_step = 0;
            break;
        case SPX_CHARD_CLIP:
            cur.len = len; cur.op = op;
            rd_step = len; sq_step = 0; rf_step = 0;
            break;
        default:
            free(ops);
            return -2; /* U3 */
        }
        if (a->is_rev) {
            cur.rde_f = cur.rds_f - 1;
            cur.rds_f -= rd_step;
        } else {
            cur.rds_f = cur.rde_f + 1;
            cur.rde_f += rd_step;
        }
        cur.sqs = cur.sqe + 1;
        cur.sqe += sq_step;
        cur.rfs = cur.rfe + 1;
        cur.rfe += rf_step;
        cur.ret = cur.len;
        if (n == cap) { cap *= 2; ops = realloc(ops, sizeof(orc_op) * cap); }
        ops[n++] = cur;
        if (n > 50000000) { free(ops); return -4; }
    }
    { /* U6: an alignment without a single aligned base (all clips / insertions): the reference would index an
       * empty SEQ/QUAL; no aligner writes such a record -- the group is rejected, like U3 */
        int t, aligned = 0;
        for (t = 1; t < n; ++t) aligned |= is_mx(ops[t].op) && ops[t].ret > 0;
        if (!aligned || a->l_qseq <= 0) { free(ops); return -2; }
    }
    a->ops = ops;
    a->n_ops = n;
    return 0;
}

/* number of states a `while (ptCigarIt_next(it))` loop visits, and the state
 * the iterator rests on afterwards */
static int loop_end(const waln *a, int *rest)
{
    int t;
    for (t = 1; t < a->n_ops; ++t)
        if (a->ops[t].ret == 0) { *rest = t; return t; } /* zero-length op ends the loop */
    *rest = a->n_ops - 1;
    return a->n_ops;
}

int orc_walk_cigar(const spx_batch *bt, int ai, orc_op **ops_out)
{
    waln a;
    int rc;
    memset(&a, 0, sizeof a);
    a.flag = bt->flag[ai]; a.pos = bt->pos[ai]; a.l_qseq = bt->l_qseq[ai]; a.n_cigar = bt->n_cigar[ai];
    a.is_rev = (a.flag & SPX_FREVERSE) != 0;
    a.cigar = bt->cigar + bt->cigar_off[ai];
    a.cs = bt->cs_off[ai] >= 0 ? bt->cs + bt->cs_off[ai] : NULL;
    a.md = (bt->md_off && bt->md && bt->md_off[ai] >= 0) ? bt->md + bt->md_off[ai] : NULL;
    rc = walk_cigar(&a);
    if (rc < 0) return rc;
    *ops_out = a.ops;
    return a.n_ops;
}

static int is_mx(int op) { return op == SPX_CMATCH || op == SPX_CEQUAL || op == SPX_CDIFF; }

/* ptAlignment_init_coordinates (ptAlignment.c:42-95) */
static void init_coordinates(waln *a)
{
    int t, rest, end = loop_end(a, &rest);
    a->rfs = a->rfe = a->rde_f = a->rds_f = -1;
    for (t = 1; t < end; ++t) {
        const orc_op *o = &a->ops[t];
        if (a->rfs == -1 && is_mx(o->op)) {
            a->rfs = o->rfs;
            if (a->is_rev) a->rde_f = o->rde_f; else a->rds_f = o->rds_f;
        }
        if (a->rfe == -1 && a->rfs != -1 && (o->op == SPX_CHARD_CLIP || o->op == SPX_CSOFT_CLIP)) {
            a->rfe = o->rfe;
            if (a->is_rev) a->rds_f = o->rde_f + 1; else a->rde_f = o->rds_f - 1;
        }
    }
    {
        const orc_op *o = &a->ops[rest];
        if (a->rfe == -1 && is_mx(o->op)) {
            a->rfe = o->rfe;
            if (a->is_rev) a->rds_f = o->rds_f; else a->rde_f = o->rde_f;
        }
    }
}

/* ---- markers ------------------------------------------------------------ */
typedef struct { orc_marker *v; int n, cap; } mvec;
static void mpush(mvec *m, orc_marker x)
{
    if (m->n == m->cap) { m->cap = m->cap ? m->cap * 2 : 64; m->v = realloc(m->v, sizeof(orc_marker) * m->cap); }
    m->v[m->n++] = x;
}
static int marker_cmp(const void *pa, const void *pb) /* ptMarker_cmp, ptMarker.c:31-39 */
{
    const orc_marker *a = pa, *b = pb;
    if (a->read_pos_f == b->read_pos_f) return a->alignment_idx - b->alignment_idx;
    return a->read_pos_f - b->read_pos_f;
}

/* ptMarker_get_initial_markers (ptMarker.c:42-75) */
static void initial_markers(waln *al, int n, int min_q, mvec *mk)
{
    int i, t, j;
    for (i = 0; i < n; ++i) {
        int rest, end = loop_end(&al[i], &rest);
        for (t = 1; t < end; ++t) {
            const orc_op *o = &al[i].ops[t];
            if (o->op != SPX_CDIFF) continue;
            for (j = 0; j < o->len; ++j) {
                orc_marker m;
                if (al[i].qual[o->sqs + j] < min_q) continue;
                m.alignment_idx = i;
                m.base_idx = o->sqs + j;
                m.read_pos_f = al[i].is_rev ? o->rde_f - j : o->rds_f + j;
                m.base_q = al[i].qual[o->sqs + j];
                m.is_match = 0;
                m.ref_pos = o->rfs + j;
                mpush(mk, m);
            }
        }
    }
}

/* remove_all_mismatch_markers (ptMarker.c:209-248) */
static void remove_all_mismatch(mvec *mk, int n_aln)
{
    int i, occ = 0, n = mk->n, w = 0;
    char *keep;
    qsort(mk->v, n, sizeof(orc_marker), marker_cmp);
    if (n == 0) return;
    keep = malloc(n);
    memset(keep, 1, n);
    for (i = 0; i < n; ++i) {
        if (i > 0 && mk->v[i - 1].read_pos_f < mk->v[i].read_pos_f) {
            if (occ == n_aln) memset(keep + i - n_aln, 0, n_aln);
            occ = 0;
        }
        occ += 1;
    }
    if (occ == n_aln) memset(keep + n - n_aln, 0, n_aln);
    for (i = 0; i < n; ++i) if (keep[i]) mk->v[w++] = mk->v[i];
    mk->n = w;
    free(keep);
}

/* ptMarker_construct_match (ptMarker.c:77-107) */
static orc_marker make_match(const waln *al, int ai, int read_pos_f)
{
    const waln *a = &al[ai];
    orc_marker m;
    m.alignment_idx = ai;
    m.read_pos_f = read_pos_f;
    m.base_idx = a->is_rev ? a->l_qseq + a->rclip - read_pos_f - 1 : read_pos_f - a->lclip;
    m.base_q = (m.base_idx >= 0 && m.base_idx < a->l_qseq) ? a->qual[m.base_idx] : 0; /* U2 */
    m.is_match = 1;
    m.ref_pos = -1;
    return m;
}

/* sort_and_fill_markers (ptMarker.c:251-295) */
static void sort_and_fill(mvec *mk, const waln *al, int n_aln)
{
    int i, j, idx = 0, n0 = mk->n;
    qsort(mk->v, mk->n, sizeof(orc_marker), marker_cmp);
    for (i = 0; i < n0; ++i) {
        orc_marker cur = mk->v[i];
        if (i > 0 && mk->v[i - 1].read_pos_f < cur.read_pos_f) {
            for (j = idx; j < n_aln; ++j) mpush(mk, make_match(al, j, mk->v[i - 1].read_pos_f));
            idx = 0;
        }
        for (j = idx; j < cur.alignment_idx; ++j) mpush(mk, make_match(al, j, cur.read_pos_f));
        idx = cur.alignment_idx + 1;
    }
    if (n0 > 0)
        for (j = idx; j < n_aln; ++j) mpush(mk, make_match(al, j, mk->v[n0 - 1].read_pos_f));
    if (mk->n > 1) qsort(mk->v, mk->n, sizeof(orc_marker), marker_cmp); /* (no markers: v is NULL, which qsort must not be given) */
}

/* filter_ins_markers (ptMarker.c:156-206) */
static void filter_ins(mvec *mk, const waln *al, int n_aln)
{
    int n = mk->n, i, t, w = 0;
    char *keep;
    if (n == 0) return;
    keep = malloc(n);
    memset(keep, 1, n);
    for (i = 0; i < n_aln; ++i) {
        const waln *a = &al[i];
        int rest, end = loop_end(a, &rest);
        int j = a->is_rev ? n - 1 : 0, step = a->is_rev ? -1 : 1;
        for (t = 1; t < end; ++t) {
            const orc_op *o = &a->ops[t];
            while (j >= 0 && j < n && o->rds_f <= mk->v[j].read_pos_f && o->rde_f >= mk->v[j].read_pos_f) {
                orc_marker *m = &mk->v[j];
                if (o->op == SPX_CINS || o->op == SPX_CSOFT_CLIP || o->op == SPX_CHARD_CLIP) keep[j] = 0;
                if (o->op == SPX_CEQUAL && m->alignment_idx == i)
                    m->ref_pos = a->is_rev ? o->rfs + o->rde_f - m->read_pos_f : o->rfs + m->read_pos_f - o->rds_f;
                j += step;
            }
        }
    }
    for (i = 0; i < n; ++i) if (keep[i]) mk->v[w++] = mk->v[i];
    mk->n = w;
    free(keep);
}

/* ---- blocks ------------------------------------------------------------- */
typedef struct { orc_block *v; int n, cap; } bvec;
static void bpush(bvec *b, int rfs, int rfe, int sqs, int sqe, int rds_f, int rde_f)
{
    orc_block x = {rfs, rfe, sqs, sqe, rds_f, rde_f};
    if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 16; b->v = realloc(b->v, sizeof(orc_block) * b->cap); }
    b->v[b->n++] = x;
}
static int cmp_rds_f(const void *a, const void *b) { return ((const orc_block *)a)->rds_f - ((const orc_block *)b)->rds_f; } /* ptBlock.c:172-176 */
static int cmp_sqs(const void *a, const void *b) { return ((const orc_block *)a)->sqs - ((const orc_block *)b)->sqs; }       /* ptBlock.c:178-182 */

/* find_confident_blocks (ptMarker.c:328-395) */
static void find_confident(waln *a, int threshold)
{
    bvec out = {0};
    int t, rest, end = loop_end(a, &rest);
    int conf_sqs = 0, conf_rfs = a->pos;
    int conf_rd_f = a->is_rev ? a->ops[0].rde_f : a->ops[0].rds_f;
    for (t = 1; t < end; ++t) {
        const orc_op *o = &a->ops[t];
        if (o->op == SPX_CINS || o->op == SPX_CDEL) {
            if (o->len > threshold && conf_sqs < o->sqs && conf_rfs < o->rfs) {
                if (a->is_rev) bpush(&out, conf_rfs, o->rfs - 1, conf_sqs, o->sqs - 1, o->rde_f + 1, conf_rd_f);
                else bpush(&out, conf_rfs, o->rfs - 1, conf_sqs, o->sqs - 1, conf_rd_f, o->rds_f - 1);
            }
            if (o->len > threshold) {
                conf_sqs = o->sqe + 1;
                conf_rfs = o->rfe + 1;
                conf_rd_f = a->is_rev ? o->rds_f - 1 : o->rde_f + 1;
            }
        } else if (o->op == SPX_CSOFT_CLIP || o->op == SPX_CHARD_CLIP) {
            if (conf_sqs < o->sqs && conf_rfs < o->rfs) {
                if (a->is_rev) bpush(&out, conf_rfs, o->rfs - 1, conf_sqs, o->sqs - 1, o->rde_f + 1, conf_rd_f);
                else bpush(&out, conf_rfs, o->rfs - 1, conf_sqs, o->sqs - 1, conf_rd_f, o->rds_f - 1);
            }
            conf_sqs = o->sqe + 1;
            conf_rfs = o->rfe + 1;
            conf_rd_f = a->is_rev ? o->rds_f - 1 : o->rde_f + 1;
        }
    }
    {
        const orc_op *o = &a->ops[rest];
        if (conf_sqs <= o->sqe) {
            if (a->is_rev) bpush(&out, conf_rfs, o->rfe, conf_sqs, o->sqe, o->rds_f, conf_rd_f);
            else bpush(&out, conf_rfs, o->rfe, conf_sqs, o->sqe, conf_rd_f, o->rde_f);
        }
    }
    free(a->conf);
    a->conf = out.v;
    a->n_conf = out.n;
}

/* find_flanking_blocks (ptMarker.c:446-484) */
static void find_flanking(waln *a, const mvec *mk, int margin)
{
    bvec out = {0};
    int i, start, end;
    start = imax(a->rds_f, mk->v[0].read_pos_f - margin);
    end = imin(a->rde_f, mk->v[0].read_pos_f + margin);
    for (i = 1; i < mk->n; ++i) {
        int cs = imax(a->rds_f, mk->v[i].read_pos_f - margin);
        int ce = imin(a->rde_f, mk->v[i].read_pos_f + margin);
        if (cs < end) end = ce;
        else {
            bpush(&out, -1, -1, -1, -1, start, end);
            start = cs;
            end = ce;
        }
    }
    bpush(&out, -1, -1, -1, -1, start, end);
    free(a->flank);
    a->flank = out.v;
    a->n_flank = out.n;
}

/* intersect_by_rd_f (ptMarker.c:398-435) */
static bvec intersect_rd_f(const orc_block *b1, int n1, const orc_block *b2, int n2)
{
    bvec out = {0};
    int i, j = 0;
    if (n1 == 0 || n2 == 0) return out;
    for (i = 0; i < n1; ++i) {
        while (j < n2 && b2[j].rde_f < b1[i].rds_f) j++;
        while (j < n2 && b2[j].rds_f < b1[i].rde_f) {
            bpush(&out, -1, -1, -1, -1, imax(b1[i].rds_f, b2[j].rds_f), imin(b1[i].rde_f, b2[j].rde_f));
            if (b2[j].rde_f <= b1[i].rde_f) j++;
            else break;
        }
    }
    return out;
}

/* correct_conf_blocks (ptMarker.c:495-647) */
static int correct_conf(waln *al, int n_aln, int threshold)
{
    bvec blocks = {0}, nb;
    int i;
    qsort(al[0].conf, al[0].n_conf, sizeof(orc_block), cmp_rds_f);
    for (i = 0; i < al[0].n_conf; ++i) {
        const orc_block *s = &al[0].conf[i];
        bpush(&blocks, s->rfs, s->rfe, s->sqs, s->sqe, s->rds_f, s->rde_f);
    }
    for (i = 1; i < n_aln; ++i) {
        qsort(al[i].conf, al[i].n_conf, sizeof(orc_block), cmp_rds_f);
        nb = intersect_rd_f(blocks.v, blocks.n, al[i].conf, al[i].n_conf);
        free(blocks.v);
        blocks = nb;
    }
    for (i = 0; i < n_aln; ++i) {
        qsort(al[i].flank, al[i].n_flank, sizeof(orc_block), cmp_rds_f);
        nb = intersect_rd_f(blocks.v, blocks.n, al[i].flank, al[i].n_flank);
        free(blocks.v);
        blocks = nb;
    }
    if (blocks.n == 0) {
        for (i = 0; i < n_aln; ++i) { free(al[i].conf); al[i].conf = NULL; al[i].n_conf = 0; }
        free(blocks.v);
        return 0;
    }
    /* projection of the consensus read intervals onto each alignment (:528-643) */
    for (i = 0; i < n_aln; ++i) {
        waln *a = &al[i];
        bvec out = {0};
        int rev = a->is_rev, nblk = blocks.n;
        int j = rev ? nblk - 1 : 0, have_block = 1, del_flag = 0;
        int brs, bre, rfs = -1, rfe = -1, sqs = -1, sqe = -1, t, rest, end = loop_end(a, &rest);
        if (rev) { brs = -blocks.v[j].rde_f; bre = -blocks.v[j].rds_f; }
        else { brs = blocks.v[j].rds_f; bre = blocks.v[j].rde_f; }
        for (t = 1; t < end; ++t) {
            const orc_op *o = &a->ops[t];
            int crs = rev ? -o->rde_f : o->rds_f, cre = rev ? -o->rds_f : o->rde_f;
            if (is_mx(o->op) || o->op == SPX_CINS) {
                int ins = o->op == SPX_CINS;
                while (have_block && bre <= cre) {
                    if (crs <= brs && !(del_flag && crs == brs)) {
                        rfs = ins ? o->rfs : o->rfs + (brs - crs);
                        sqs = o->sqs + (brs - crs);
                    }
                    rfe = ins ? o->rfe : o->rfs + (bre - crs);
                    sqe = o->sqs + (bre - crs);
                    bpush(&out, rfs, rfe, sqs, sqe, blocks.v[j].rds_f, blocks.v[j].rde_f);
                    if (rev && j > 0) {
                        j--;
                        brs = -blocks.v[j].rde_f; bre = -blocks.v[j].rds_f;
                    } else if (!rev && j < nblk - 1) {
                        j++;
                        brs = blocks.v[j].rds_f; bre = blocks.v[j].rde_f;
                    } else if (j == 0 || j == nblk - 1) {
                        have_block = 0;
                    }
                }
                if (!have_block) break;
                if (crs <= brs && brs <= cre && !(del_flag && crs == brs)) {
                    rfs = ins ? o->rfs : o->rfs + (brs - crs);
                    sqs = o->sqs + (brs - crs);
                }
                del_flag = 0;
            } else if (o->op == SPX_CDEL) {
                /* :615-625 also patches rfe of the previous CONSENSUS block object,
                 * which is never read again -- no observable effect, not restated. */
                if (have_block && brs == crs && o->len <= threshold) {
                    del_flag = 1;
                    rfs = o->rfs;
                    sqs = o->sqs;
                }
            }
        }
        free(a->conf);
        qsort(out.v, out.n, sizeof(orc_block), cmp_sqs);
        a->conf = out.v;
        a->n_conf = out.n;
    }
    i = blocks.n;
    free(blocks.v);
    return i;
}

/* needs_to_find_blocks (ptMarker.c:649-667) */
static int needs_blocks(const waln *al, int n_aln, int threshold)
{
    int j, i, flag = 0;
    for (j = 0; j < n_aln; ++j) {
        if (al[j].conf == NULL) return 1;
        if (al[j].n_conf == 0) return 1;
        for (i = 0; i < al[j].n_conf; ++i) {
            const orc_block *b = &al[j].conf[i];
            if ((b->sqe - b->sqs) > threshold || (b->rfe - b->rfs) > threshold) flag = 1;
        }
    }
    return flag;
}

/* ---- BAQ driver --------------------------------------------------------- */
typedef struct {
    orc_baq_call *calls;
    int max_calls, n_calls;
    long long cells;
} baq_trace;

static long long band_cells(int L, int R, int bw_in)
{
    int bw = L > R ? L : R, i;
    long long c = 0;
    if (bw > bw_in) bw = bw_in;
    if (bw < abs(R - L)) bw = abs(R - L);
    for (i = 1; i <= L; ++i) {
        int beg = imax(1, i - bw), end = imin(R, i + bw);
        if (end >= beg) c += end - beg + 1;
    }
    return c;
}

/* calc_local_baq (ptMarker.c:670-809) */
static int local_baq(const spx_ref *ref, waln *a, int ai, mvec *mk, const spx_params *par, baq_trace *tr)
{
    orc_probaln_par conf;
    int nm = mk->n, step = a->is_rev ? -1 : 1;
    int j = a->is_rev ? nm - 1 : 0;
    int ci = 0; /* iterator state index */
    int bi;
    const int block_margin = 10;
    uint8_t *qual = a->qual;
    conf.d = (float)par->conf_d;
    conf.e = (float)par->conf_e;
    conf.bw = (int)par->conf_b;
#define IT (a->ops[ci])
#define IT_NEXT() ((ci < a->n_ops - 1) ? (ci++, a->ops[ci].ret) : 0)
#define MK(j_) ((j_) >= 0 && (j_) < nm ? &mk->v[j_] : NULL)
    for (bi = 0; bi < a->n_conf; ++bi) {
        const orc_block *blk = &a->conf[bi];
        orc_marker *m;
        while (IT.sqe < blk->sqs || IT.rfe < blk->rfs)
            if (IT_NEXT() == 0) break;
        while ((m = MK(j)) && (m->base_idx < blk->sqs + block_margin || m->alignment_idx != ai)) {
            if (m->alignment_idx == ai && blk->sqs <= m->base_idx) qual[m->base_idx] = 0;
            j += step;
        }
        m = MK(j);
        if (m && m->base_idx <= blk->sqe - block_margin && blk->sqs + block_margin <= m->base_idx) {
            int seq_len = blk->sqe - blk->sqs + 1, ref_len = blk->rfe - blk->rfs + 1, k, rc;
            uint8_t *tseq, *tref, *bq, *bqual, *q;
            int *state;
            const char *contig;
            if (seq_len <= 0 || ref_len <= 0) return -5;
            if (blk->rfs < 0 || blk->rfe >= ref->seq_off[a->tid + 1] - ref->seq_off[a->tid]) return -6;
            tseq = malloc(seq_len);
            tref = malloc(ref_len);
            bqual = malloc(seq_len);
            bq = malloc(seq_len);
            q = malloc(seq_len);
            state = malloc(sizeof(int) * seq_len);
            for (k = 0; k < seq_len; ++k) {
                int p = blk->sqs + k;
                tseq[k] = orc_nt16_int[(a->seq4[p >> 1] >> ((~p & 1) << 2)) & 0xf];
            }
            contig = ref->bases + ref->seq_off[a->tid];
            for (k = 0; k < ref_len; ++k) tref[k] = orc_nt16_int[orc_nt16_table[(unsigned char)contig[blk->rfs + k]]];
            for (k = 0; k < seq_len; ++k) bqual[k] = (uint8_t)par->set_q;
            conf.bw = (int)(abs(ref_len - seq_len) + par->conf_b);
            rc = orc_probaln_glocal(tref, ref_len, tseq, seq_len, bqual, &conf, state, q);
            if (tr) {
                if (tr->calls && tr->n_calls < tr->max_calls) {
                    orc_baq_call *c = &tr->calls[tr->n_calls];
                    c->aln = ai; c->block = bi; c->sqs = blk->sqs; c->sqe = blk->sqe;
                    c->rfs = blk->rfs; c->rfe = blk->rfe; c->bw = conf.bw;
                }
                tr->n_calls++;
                tr->cells += band_cells(seq_len, ref_len, conf.bw);
            }
            (void)rc;
            memcpy(bq, bqual, seq_len);
            while (IT.sqs <= blk->sqe || IT.rfs <= blk->rfe) {
                int x = IT.rfs - blk->rfs, y = IT.sqs - blk->sqs;
                if (x < 0) x = 0;
                if (y < 0) y = 0;
                if (is_mx(IT.op)) {
                    int len = imin(IT.len, imin(IT.sqe, blk->sqe) - imax(IT.sqs, blk->sqs) + 1), t;
                    for (t = y; t < y + len; ++t) {
                        if (t >= seq_len) { free(tseq); free(tref); free(bqual); free(bq); free(q); free(state); return -7; }
                        if ((state[t] & 3) != 0 || (state[t] >> 2) != x + (t - y)) bq[t] = 0;
                        else bq[t] = qual[blk->sqs + t] < q[t] ? qual[blk->sqs + t] : q[t];
                    }
                }
                if (IT.sqe <= blk->sqe || IT.rfe <= blk->rfe) {
                    if (IT_NEXT() == 0) break;
                } else break;
            }
            for (k = block_margin; k < seq_len - block_margin; ++k) qual[blk->sqs + k] = bq[k] < 94 ? bq[k] : 93;
            free(tseq); free(tref); free(bqual); free(bq); free(q); free(state);
        }
        while ((m = MK(j)) && ((m->base_idx <= blk->sqe && m->alignment_idx == ai) || m->alignment_idx != ai)) {
            if (blk->sqe - block_margin <= m->base_idx && m->alignment_idx == ai) qual[m->base_idx] = 0;
            j += step;
        }
    }
#undef IT
#undef IT_NEXT
#undef MK
    return 0;
}

/* filter_lowq_markers (ptMarker.c:110-153) */
static void filter_lowq(mvec *mk, int threshold)
{
    int n = mk->n, j, k, idx_s = 0, min_q = 100;
    mvec out = {0};
    if (n == 0) return;
    for (j = 0; j < n; ++j) {
        if (j > 0 && mk->v[j].read_pos_f != mk->v[j - 1].read_pos_f) {
            if (min_q > threshold)
                for (k = idx_s; k < j; ++k) { orc_marker c = mk->v[k]; c.base_q = min_q; mpush(&out, c); }
            idx_s = j;
            min_q = 100;
        }
        if (min_q > mk->v[j].base_q) min_q = mk->v[j].base_q;
    }
    if (min_q > threshold)
        for (k = idx_s; k < n; ++k) { orc_marker c = mk->v[k]; c.base_q = min_q; mpush(&out, c); }
    free(mk->v);
    *mk = out;
}

/* reverse_quality (ptMarker.c:298-304) */
static double reverse_quality(uint8_t q)
{
    double p;
    if (q >= 93) return 0;
    if (q == 0) return 93;
    p = 1 - pow(10, (double)q / -10);
    return -10 * log(p);
}

/* calc_alignment_score (ptMarker.c:307-325) */
static void alignment_scores(const mvec *mk, waln *al)
{
    int j;
    for (j = 0; j < mk->n; ++j) {
        const orc_marker *m = &mk->v[j];
        if (m->is_match) al[m->alignment_idx].score += -1 * reverse_quality((uint8_t)m->base_q);
        else al[m->alignment_idx].score += -1 * m->base_q - 10 * log(3);
    }
}

static int cvt_trunc_x86(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}

/* get_best_record_index (ptAlignment.c:137-177) */
static int best_record_index(const int *flag, const double *score, int n, double prim_margin, double min_score,
                             double prim_margin_random, orc_rand *rng, int *n_rand)
{
    double max_score = -DBL_MAX, prim_score = -DBL_MAX;
    int max_idx = -1, prim_idx = -1, i, cnt = 0, tied[16], rnd, d;
    *n_rand = 0;
    if (n == 1) return 0;
    for (i = 0; i < n; ++i) {
        if ((flag[i] & SPX_FSECONDARY) == 0) { prim_idx = i; prim_score = score[i]; }
        else if (max_score < score[i]) { max_idx = i; max_score = score[i]; }
    }
    for (i = 0; i < n; ++i)
        if ((flag[i] & SPX_FSECONDARY) != 0 && max_score <= score[i]) tied[cnt++] = i;
    if (cnt > 1) { max_idx = tied[orc_rand_next(rng) % cnt]; (*n_rand)++; }
    rnd = orc_rand_next(rng) % 2;
    (*n_rand)++;
    d = cvt_trunc_x86(max_score - prim_score); /* abs() is int abs(int) */
    if (d < 0 && d != INT_MIN) d = -d;
    if (d < prim_margin_random) return rnd == 0 ? prim_idx : max_idx;
    if (prim_idx == -1 || max_score <= (prim_score + prim_margin) || max_score < min_score) return prim_idx;
    return max_idx;
}

int orc_group_is_dispatched(const spx_batch *bt, int g)
{
    int a0 = bt->grp_first[g], a1 = bt->grp_first[g + 1], n = 0, supp = 0, prim = 0, a;
    /* unmapped records are never added; more than 11 mapped records keep n at 11 (secphase.c:336-339) */
    for (a = a0; a < a1; ++a) {
        if (bt->flag[a] & SPX_FUNMAP) continue;
        if (n > 10) continue;
        n++;
        if (bt->flag[a] & SPX_FSUPPLEMENTARY) supp++;
        if ((bt->flag[a] & SPX_FSECONDARY) == 0) prim++;
    }
    return n > 1 && n <= 10 && supp == 0 && prim == 1;
}

static void free_alns(waln *al, int n)
{
    int i;
    for (i = 0; i < n; ++i) { free(al[i].ops); free(al[i].qual); free(al[i].conf); free(al[i].flank); }
}

/* print_alignment_scores + record header (secphase.c:32-57,194-200) */
static void write_record(FILE *log, const spx_batch *bt, const spx_ref *ref, int g, const int *amap,
                         const orc_group_result *r)
{
    int i;
    fprintf(log, "#MARKER SCORE\n");
    fprintf(log, "$\t%s\n", bt->qnames + bt->qname_off[g]);
    for (i = 0; i < r->n_aln; ++i) {
        int a = amap[i];
        if ((bt->flag[a] & SPX_FSECONDARY) == 0) fprintf(log, "*\t");
        else if (i == r->best_idx) fprintf(log, "@\t");
        else fprintf(log, "!\t");
        fprintf(log, "%.2f\t%s\t%ld\t%d\n", r->score[i], ref->names + ref->name_off[bt->tid[a]], (long)bt->pos[a],
                r->rfe[i]);
    }
    fprintf(log, "\n");
}

static int group_alns(const spx_batch *bt, int g, int *amap)
{
    int a0 = bt->grp_first[g], a1 = bt->grp_first[g + 1], n = 0, a;
    for (a = a0; a < a1; ++a) {
        if (bt->flag[a] & SPX_FUNMAP) continue;
        if (n > 10) continue;
        amap[n++] = a;
    }
    return n;
}

static int g_keep_markers = 0; /* set by orc_run_batch_bed while it runs */
static uint8_t *g_qual_sink = NULL; /* set by orc_run_batch_quals while it runs: laid out like bt->qual */

/* marker branch of runOneThread up to (not including) get_best_record_index */
static int score_group(const spx_batch *bt, const spx_ref *ref, int g, const spx_params *par, orc_group_result *out,
                       orc_baq_call *calls, int max_calls)
{
    waln al[16];
    int amap[16], n, i, rc = 0, conf_blocks_length = 1 /* U1 */;
    mvec mk = {0};
    baq_trace tr = {calls, max_calls, 0, 0};
    memset(out, 0, sizeof *out);
    memset(al, 0, sizeof al);
    n = group_alns(bt, g, amap);
    out->n_aln = n;
    for (i = 0; i < n; ++i) {
        int a = amap[i];
        waln *w = &al[i];
        w->flag = bt->flag[a]; w->tid = bt->tid[a]; w->pos = bt->pos[a]; w->l_qseq = bt->l_qseq[a];
        w->n_cigar = bt->n_cigar[a];
        w->is_rev = (w->flag & SPX_FREVERSE) != 0;
        w->cigar = bt->cigar + bt->cigar_off[a];
        w->seq4 = bt->seq4 + bt->seq_off[a];
        w->qual = malloc(w->l_qseq > 0 ? w->l_qseq : 1);
        memcpy(w->qual, bt->qual + bt->qual_off[a], w->l_qseq);
        w->cs = bt->cs_off[a] >= 0 ? bt->cs + bt->cs_off[a] : NULL;
        w->md = (bt->md_off && bt->md && bt->md_off[a] >= 0) ? bt->md + bt->md_off[a] : NULL;
        rc = walk_cigar(w);
        if (rc < 0) { free_alns(al, n); return rc; }
        init_coordinates(w);
        w->score = 0.0;
    }
    initial_markers(al, n, par->min_q, &mk);
    out->n_markers_initial = mk.n;
    remove_all_mismatch(&mk, n);
    sort_and_fill(&mk, al, n);
    filter_ins(&mk, al, n);
    if (mk.n > 0) {
        int margin = par->flank_margin, iter = 0;
        for (i = 0; i < n; ++i) find_confident(&al[i], par->indel_threshold);
        while (par->consensus && needs_blocks(al, n, 1000)) {
            margin = (int)(margin * 0.8); /* flank_margin_eff *= 0.8 on an int */
            for (i = 0; i < n; ++i) find_flanking(&al[i], &mk, margin);
            conf_blocks_length = correct_conf(al, n, par->indel_threshold);
            if (conf_blocks_length == 0) break;
            if (++iter >= 64) break; /* U5 */
        }
        if (conf_blocks_length > 0 || !par->consensus) {
            out->n_blocks = al[0].n_conf;
            if (par->baq_flag) {
                for (i = 0; i < n && rc == 0; ++i) rc = local_baq(ref, &al[i], i, &mk, par, &tr);
                if (rc < 0) { free(mk.v); free_alns(al, n); return rc; }
                for (i = 0; i < mk.n; ++i) mk.v[i].base_q = al[mk.v[i].alignment_idx].qual[mk.v[i].base_idx];
            }
            filter_lowq(&mk, par->min_q);
            alignment_scores(&mk, al);
        }
    }
    out->n_markers_final = mk.n;
    out->n_baq_calls = tr.n_calls;
    out->dp_cells = tr.cells;
    out->prim_idx = -1;
    for (i = 0; i < n; ++i) {
        out->score[i] = al[i].score;
        out->rfe[i] = al[i].rfe;
        out->rfs[i] = al[i].rfs;
        if (out->prim_idx < 0 && (al[i].flag & SPX_FSECONDARY) == 0) out->prim_idx = i;
    }
    if (g_keep_markers) { out->final_markers = mk.v; out->n_final = mk.n; mk.v = NULL; }
    /* what sam_write1 would see at secphase.c:182-189: the record qualities after calc_local_baq */
    if (g_qual_sink)
        for (i = 0; i < n; ++i) memcpy(g_qual_sink + bt->qual_off[amap[i]], al[i].qual, al[i].l_qseq > 0 ? al[i].l_qseq : 0);
    free(mk.v);
    free_alns(al, n);
    return 0;
}

static void decide_group(const spx_batch *bt, int g, const spx_params *par, orc_rand *rng, orc_group_result *r)
{
    int amap[16], flag[16], n = group_alns(bt, g, amap), i;
    for (i = 0; i < n; ++i) flag[i] = bt->flag[amap[i]];
    r->best_idx = best_record_index(flag, r->score, n, par->prim_margin_score, (double)par->min_score,
                                    par->prim_margin_random, rng, &r->n_rand);
    r->relabel = r->best_idx >= 0 && (flag[r->best_idx] & SPX_FSECONDARY) != 0;
}

static int g_ref_overheads;
static void reference_overheads(const spx_batch *bt, const spx_ref *ref, int g);
int orc_score_group(const spx_batch *bt, const spx_ref *ref, int g, const spx_params *par, orc_rand *rng,
                    orc_group_result *out, FILE *log, orc_baq_call *calls, int max_calls)
{
    if (g_ref_overheads) reference_overheads(bt, ref, g);
    int rc = score_group(bt, ref, g, par, out, calls, max_calls);
    if (rc < 0) return rc;
    if (rng) {
        decide_group(bt, g, par, rng, out);
        if (log && out->relabel) {
            int amap[16];
            group_alns(bt, g, amap);
            write_record(log, bt, ref, g, amap, out);
        }
    }
    return 0;
}

/* ---- bench.py's CPU-baseline bracket: work the reference does around the algorithm and this restatement does not ----
 * (a) runOneThread calls fai_load(fastaPath) for EVERY group (src/secphase.c:101): the .fai index is opened, parsed line
 *     by line and freed again; (b) every ptCigarIt_construct compiles the cs / MD pattern with regcomp (cigar_it.c:50,59)
 *     and every token is found with regexec (cigar_it.c:75,148); the marker path constructs >= 7 iterators per alignment
 *     (initial markers, all-mismatch removal via sort_and_fill, filter_ins, confident blocks, correct_conf_blocks per
 *     round, calc_local_baq, init_coordinates).  When the switch is on, the same calls are executed beside the restated
 *     algorithm; their results are not used, so the outputs do not change. */
#include <pthread.h>
#include <regex.h>
#include <unistd.h>
static char g_fai_path[256];
static pthread_mutex_t g_fai_mu = PTHREAD_MUTEX_INITIALIZER;
void orc_set_reference_overheads(int on) { g_ref_overheads = on; }
static const char *kCsPattern = "(:([0-9]+))|(([+-])([a-z]+)|([\\*]([a-z]+))+)";      /* cigar_it.h:8 */
static const char *kMdPattern = "(([A-Z])([0][A-Z])*)|([0-9]+)|([\\^]([A-Z]+))";          /* cigar_it.h:10 */

static void overhead_fai(const spx_ref *ref)
{
    pthread_mutex_lock(&g_fai_mu);
    if (!g_fai_path[0]) { /* the index fai_load would read: name, length, offset, line bases, line width */
        snprintf(g_fai_path, sizeof g_fai_path, "/tmp/spx_oracle_%d.fai", (int)getpid());
        FILE *f = fopen(g_fai_path, "w");
        if (f) {
            long long off = 0;
            for (int c = 0; c < ref->n_contigs; ++c) {
                const char *nm = ref->names + ref->name_off[c];
                const long long ln = ref->seq_off[c + 1] - ref->seq_off[c];
                off += (long long)strlen(nm) + 12;
                fprintf(f, "%s\t%lld\t%lld\t80\t81\n", nm, ln, off);
                off += ln + (ln + 79) / 80;
            }
            fclose(f);
        }
    }
    pthread_mutex_unlock(&g_fai_mu);
    FILE *f = fopen(g_fai_path, "r");
    if (!f) return;
    char line[1024], name[512];
    int cap = 16, n = 0;
    struct ent { char *name; long long len, off; int lb, lw; } *tab = malloc(sizeof(*tab) * cap);
    while (fgets(line, sizeof line, f)) {
        long long ln, of;
        int lb, lw;
        if (sscanf(line, "%511s\t%lld\t%lld\t%d\t%d", name, &ln, &of, &lb, &lw) != 5) continue;
        if (n == cap) { cap *= 2; tab = realloc(tab, sizeof(*tab) * cap); }
        tab[n].name = strdup(name); tab[n].len = ln; tab[n].off = of; tab[n].lb = lb; tab[n].lw = lw;
        ++n;
    }
    fclose(f);
    for (int i = 0; i < n; ++i) free(tab[i].name);
    free(tab);
}

static void overhead_regex(const char *cs, const char *md)
{
    const char *text = cs ? cs : md;
    if (!text) return;
    for (int rep = 0; rep < 7; ++rep) {
        regex_t re;
        regmatch_t m[8];
        if (regcomp(&re, cs ? kCsPattern : kMdPattern, REG_EXTENDED) != 0) return;
        const char *p = text;
        while (*p && regexec(&re, p, 8, m, 0) == 0 && m[0].rm_eo > 0) p += m[0].rm_eo;
        regfree(&re);
    }
}

static void reference_overheads(const spx_batch *bt, const spx_ref *ref, int g)
{
    overhead_fai(ref);
    for (int a = bt->grp_first[g]; a < bt->grp_first[g + 1]; ++a) {
        const char *cs = bt->cs_off[a] >= 0 ? bt->cs + bt->cs_off[a] : NULL;
        const char *md = (!cs && bt->md_off && bt->md && bt->md_off[a] >= 0) ? bt->md + bt->md_off[a] : NULL;
        overhead_regex(cs, md);
    }
}

/* ---- batch driver (pool over groups like secphase.c:257,303) ------------- */
typedef struct {
    const spx_batch *bt;
    const spx_ref *ref;
    const spx_params *par;
    orc_group_result *res;
    int *rc;
    volatile int *next;
} pool_arg;

static void *pool_worker(void *p_)
{
    pool_arg *p = p_;
    for (;;) {
        int g = __sync_fetch_and_add(p->next, 1);
        if (g >= p->bt->n_groups) break;
        if (!orc_group_is_dispatched(p->bt, g)) { memset(&p->res[g], 0, sizeof p->res[g]); p->rc[g] = 1; continue; }
        p->rc[g] = score_group(p->bt, p->ref, g, p->par, &p->res[g], NULL, 0);
    }
    return NULL;
}

typedef struct { int *s, *e, *c; int n, cap; } blkacc;
static void acc_push(blkacc *a, int s, int e, int c)
{
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 64;
        a->s = realloc(a->s, sizeof(int) * a->cap); a->e = realloc(a->e, sizeof(int) * a->cap);
        a->c = realloc(a->c, sizeof(int) * a->cap);
    }
    a->s[a->n] = s; a->e[a->n] = e; a->c[a->n] = c; a->n++;
}
static const spx_ref *g_sort_ref;
static int cmp_contig(const void *a, const void *b)
{
    return strcmp(g_sort_ref->names + g_sort_ref->name_off[*(const int *)a], g_sort_ref->names + g_sort_ref->name_off[*(const int *)b]);
}
/* merge_and_save_blocks (secphase.c:59-72): sort by start, ptBlock_merge_blocks_v2, ptBlock_save_in_bed */
static void save_bed(const spx_ref *ref, blkacc *acc, const char *path, int with_count)
{
    FILE *fp = fopen(path, "w");
    int nc = ref->n_contigs, i, k, *order = malloc(sizeof(int) * (nc ? nc : 1));
    if (!fp) { free(order); return; }
    for (i = 0; i < nc; ++i) order[i] = i;
    g_sort_ref = ref;
    qsort(order, nc, sizeof(int), cmp_contig);
    for (k = 0; k < nc; ++k) {
        blkacc *a = &acc[order[k]];
        int m, *os, *oe, *oc;
        if (a->n == 0) continue;
        orc_blocks_sort(a->n, a->s, a->e, a->c);
        os = malloc(sizeof(int) * (2 * a->n + 1)); oe = malloc(sizeof(int) * (2 * a->n + 1)); oc = malloc(sizeof(int) * (2 * a->n + 1));
        m = orc_blocks_merge_v2(a->n, a->s, a->e, with_count ? a->c : NULL, os, oe, with_count ? oc : NULL);
        for (i = 0; i < m; ++i) {
            if (oe[i] < os[i]) continue;
            if (with_count) fprintf(fp, "%s\t%d\t%d\t%d\n", ref->names + ref->name_off[order[k]], os[i], oe[i] + 1, oc[i]);
            else fprintf(fp, "%s\t%d\t%d\n", ref->names + ref->name_off[order[k]], os[i], oe[i] + 1);
        }
        free(os); free(oe); free(oc);
    }
    fclose(fp);
    free(order);
}

int orc_run_batch_bed(const spx_batch *bt, const spx_ref *ref, const spx_params *par, int threads, unsigned rand_seed,
                      orc_group_result *results, const char *log_path, const char *bed_modified_path,
                      const char *bed_marker_path)
{
    pthread_t th[256];
    pool_arg pa;
    volatile int next = 0;
    int i, g, relabelled = 0, *rc = calloc(bt->n_groups > 0 ? bt->n_groups : 1, sizeof(int));
    orc_rand rng;
    FILE *log = log_path ? fopen(log_path, "w") : NULL;
    const int want_bed = bed_modified_path || bed_marker_path;
    blkacc *acc_mod = calloc(ref->n_contigs > 0 ? ref->n_contigs : 1, sizeof(blkacc));
    blkacc *acc_mk = calloc(ref->n_contigs > 0 ? ref->n_contigs : 1, sizeof(blkacc));
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    g_keep_markers = want_bed;
    pa.bt = bt; pa.ref = ref; pa.par = par; pa.res = results; pa.rc = rc; pa.next = &next;
    if (threads == 1) pool_worker(&pa);
    else {
        for (i = 0; i < threads; ++i) pthread_create(&th[i], NULL, pool_worker, &pa);
        for (i = 0; i < threads; ++i) pthread_join(th[i], NULL);
    }
    g_keep_markers = 0;
    orc_srand(&rng, rand_seed);
    for (g = 0; g < bt->n_groups; ++g) {
        if (rc[g] != 0) { results[g].best_idx = -1; results[g].relabel = 0; if (rc[g] < 0) results[g].n_aln = rc[g]; continue; }
        decide_group(bt, g, par, &rng, &results[g]);
        if (results[g].relabel) {
            int amap[16];
            group_alns(bt, g, amap);
            relabelled++;
            if (log) write_record(log, bt, ref, g, amap, &results[g]);
            if (want_bed) { /* secphase.c:201-212 */
                int pair[2], t, k;
                pair[0] = results[g].prim_idx; pair[1] = results[g].best_idx;
                for (t = 0; t < 2; ++t) {
                    int a = pair[t], tid = bt->tid[amap[a]];
                    acc_push(&acc_mod[tid], results[g].rfs[a], results[g].rfe[a], 1);
                    for (k = 0; k < results[g].n_final; ++k)
                        if (results[g].final_markers[k].alignment_idx == a)
                            acc_push(&acc_mk[tid], results[g].final_markers[k].ref_pos, results[g].final_markers[k].ref_pos, 0);
                }
            }
        }
        free(results[g].final_markers);
        results[g].final_markers = NULL;
        results[g].n_final = 0;
    }
    if (log) fclose(log);
    if (bed_modified_path) save_bed(ref, acc_mod, bed_modified_path, 1);
    if (bed_marker_path) save_bed(ref, acc_mk, bed_marker_path, 0);
    for (i = 0; i < ref->n_contigs; ++i) {
        free(acc_mod[i].s); free(acc_mod[i].e); free(acc_mod[i].c);
        free(acc_mk[i].s); free(acc_mk[i].e); free(acc_mk[i].c);
    }
    free(acc_mod); free(acc_mk);
    free(rc);
    return relabelled;
}

int orc_run_batch_quals(const spx_batch *bt, const spx_ref *ref, const spx_params *par, int threads,
                        orc_group_result *results, uint8_t *qual_out)
{
    int rc;
    g_qual_sink = qual_out;
    rc = orc_run_batch_bed(bt, ref, par, threads, 1, results, NULL, NULL, NULL);
    g_qual_sink = NULL;
    return rc;
}

int orc_run_batch(const spx_batch *bt, const spx_ref *ref, const spx_params *par, int threads, unsigned rand_seed,
                  orc_group_result *results, const char *log_path)
{
    return orc_run_batch_bed(bt, ref, par, threads, rand_seed, results, log_path, NULL, NULL);
}
