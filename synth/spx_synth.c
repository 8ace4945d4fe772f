/*
 * spx_synth.c -- deterministic synthetic assemblies + alignment groups
 * (SURVEY.md section 8(d)).  Test/bench input generator; no scoring logic here.
 */
#include "spx_synth.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------- RNG: splitmix64 -> xoshiro256** ---------------- */
typedef struct { uint64_t s[4]; } rng_t;
static uint64_t splitmix(uint64_t *x)
{
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static void rng_seed(rng_t *r, uint64_t seed, uint64_t stream)
{
    uint64_t x = seed ^ (stream * 0xd1342543de82ef95ULL + 0x632be59bd9b4e019ULL);
    int i;
    for (i = 0; i < 4; ++i) r->s[i] = splitmix(&x);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t *r)
{
    uint64_t *s = r->s, result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}
static inline double rng_u(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint32_t rng_below(rng_t *r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }
static double rng_normal(rng_t *r)
{
    double s = 0;
    int i;
    for (i = 0; i < 12; ++i) s += rng_u(r);
    return s - 6.0;
}
/* number of non-event positions before the next event of per-position rate p */
static int64_t rng_gap(rng_t *r, double p)
{
    double u;
    if (p <= 0) return (int64_t)1 << 60;
    if (p >= 1) return 0;
    u = rng_u(r);
    return (int64_t)(log(1.0 - u) / log(1.0 - p));
}
static const char ACGT[4] = {'A', 'C', 'G', 'T'};
static inline char comp(char c)
{
    switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    default: return c;
    }
}
static inline char other_base(rng_t *r, char c)
{
    int k = rng_below(r, 3), i, j = 0;
    for (i = 0; i < 4; ++i) {
        if (ACGT[i] == c) continue;
        if (j++ == k) return ACGT[i];
    }
    return ACGT[rng_below(r, 4)]; /* c was N */
}

/* ---------------- genome ---------------- */
struct spx_synth_genome {
    spx_synth_cfg cfg;
    int n_sets;        /* 2 + n_paralogs */
    int K;
    char **seq;        /* [n_sets*K] */
    int32_t *len;      /* [n_sets*K] */
    int32_t **xpos2;   /* [K] hap-1 column -> hap-2 coordinate, -1 if deleted */
    /* flat ref */
    spx_ref ref;
    int64_t *name_off, *seq_off;
    char *names, *bases;
};

void spx_synth_default_cfg(spx_synth_cfg *c, int platform)
{
    memset(c, 0, sizeof *c);
    c->seed = 20241220ULL;
    c->n_contigs = 10;
    c->contig_len = 5000000;
    c->platform = platform;
    c->read_len = platform == SPX_SYNTH_ONT ? 30000 : platform == SPX_SYNTH_HIFI ? 15000 : 0;
    c->max_read_len = 100000;
    c->min_secondaries = 1;
    c->max_secondaries = platform == SPX_SYNTH_ONT ? 4 : platform == SPX_SYNTH_HIFI ? 2 : 8;
    c->n_paralogs = c->max_secondaries - 1;
    c->softclip_frac = 0.10;
    c->hardclip_frac = 0.0;
    c->shuffle_records = 0;
    c->inverted_paralogs = 0;
    c->n_base_frac = 0.0;
    c->snv_rate = 1.0 / 5000;
    c->indel_rate = 1.0 / 50000;
    c->paralog_snv_rate = 0.01;
}

spx_synth_genome *spx_synth_genome_create(const spx_synth_cfg *cfg)
{
    spx_synth_genome *g = calloc(1, sizeof *g);
    int K = cfg->n_contigs, n = cfg->contig_len, c, p;
    int64_t tot = 0, noff = 0;
    g->cfg = *cfg;
    g->K = K;
    g->n_sets = 2 + cfg->n_paralogs;
    g->seq = calloc((size_t)g->n_sets * K, sizeof(char *));
    g->len = calloc((size_t)g->n_sets * K, sizeof(int32_t));
    g->xpos2 = calloc(K, sizeof(int32_t *));
    for (c = 0; c < K; ++c) {
        rng_t r;
        char *h1 = malloc(n), *h2 = malloc((size_t)n + n / 1000 + 64);
        int32_t *xp = malloc(sizeof(int32_t) * n);
        int64_t b, x = 0, next_ev;
        double ev_rate = cfg->snv_rate + cfg->indel_rate;
        rng_seed(&r, cfg->seed, 1 + c);
        for (b = 0; b < n; b += 32) {
            uint64_t w = rng_next(&r);
            int k;
            for (k = 0; k < 32 && b + k < n; ++k) h1[b + k] = ACGT[(w >> (2 * k)) & 3];
        }
        if (cfg->n_base_frac > 0) {
            int64_t q = rng_gap(&r, cfg->n_base_frac);
            while (q < n) { h1[q] = 'N'; q += 1 + rng_gap(&r, cfg->n_base_frac); }
        }
        /* hap 2 = hap 1 + SNVs + 1-3 bp indels, events >= 8 columns apart */
        rng_seed(&r, cfg->seed, 101 + c);
        next_ev = 8 + rng_gap(&r, ev_rate);
        for (b = 0; b < n;) {
            if (b == next_ev && b + 8 < n) {
                double u = rng_u(&r) * ev_rate;
                if (u < cfg->snv_rate) {
                    xp[b] = (int32_t)x;
                    h2[x++] = other_base(&r, h1[b]);
                    b++;
                } else {
                    int l = 1 + rng_below(&r, 3), k;
                    if (rng_next(&r) & 1) { /* deletion in hap 2 */
                        for (k = 0; k < l; ++k) xp[b + k] = -1;
                        b += l;
                    } else { /* insertion in hap 2 after column b */
                        xp[b] = (int32_t)x;
                        h2[x++] = h1[b];
                        for (k = 0; k < l; ++k) h2[x++] = ACGT[rng_below(&r, 4)];
                        b++;
                    }
                }
                next_ev = b + 8 + rng_gap(&r, ev_rate);
            } else {
                xp[b] = (int32_t)x;
                h2[x++] = h1[b];
                b++;
            }
        }
        g->seq[0 * K + c] = h1; g->len[0 * K + c] = n;
        g->seq[1 * K + c] = h2; g->len[1 * K + c] = (int32_t)x;
        g->xpos2[c] = xp;
        for (p = 0; p < cfg->n_paralogs; ++p) {
            char *q = malloc(n);
            int64_t pos;
            memcpy(q, h1, n);
            rng_seed(&r, cfg->seed, 1000 + 64 * (uint64_t)p + c);
            pos = rng_gap(&r, cfg->paralog_snv_rate);
            while (pos < n) {
                q[pos] = other_base(&r, q[pos]);
                pos += 1 + rng_gap(&r, cfg->paralog_snv_rate);
            }
            if (cfg->inverted_paralogs && (p & 1)) {
                int64_t i, j;
                for (i = 0, j = n - 1; i < j; ++i, --j) { char t = comp(q[i]); q[i] = comp(q[j]); q[j] = t; }
                if (i == j) q[i] = comp(q[i]);
            }
            g->seq[(2 + p) * K + c] = q;
            g->len[(2 + p) * K + c] = n;
        }
    }
    /* flat reference */
    for (c = 0; c < g->n_sets * K; ++c) tot += g->len[c];
    g->bases = malloc(tot > 0 ? tot : 1);
    g->seq_off = malloc(sizeof(int64_t) * (g->n_sets * K + 1));
    g->name_off = malloc(sizeof(int64_t) * g->n_sets * K);
    g->names = malloc((size_t)g->n_sets * K * 32);
    tot = 0;
    for (c = 0; c < g->n_sets * K; ++c) {
        int set = c / K, k = c % K;
        g->seq_off[c] = tot;
        memcpy(g->bases + tot, g->seq[c], g->len[c]);
        tot += g->len[c];
        g->name_off[c] = noff;
        if (set < 2) noff += 1 + sprintf(g->names + noff, "synth#%d#ctg%d", set + 1, k);
        else noff += 1 + sprintf(g->names + noff, "synth#p%d#ctg%d", set - 1, k);
    }
    g->seq_off[g->n_sets * K] = tot;
    g->ref.n_contigs = g->n_sets * K;
    g->ref.name_off = g->name_off;
    g->ref.names = g->names;
    g->ref.seq_off = g->seq_off;
    g->ref.bases = g->bases;
    return g;
}

const spx_ref *spx_synth_genome_ref(const spx_synth_genome *g) { return &g->ref; }

void spx_synth_genome_free(spx_synth_genome *g)
{
    int i;
    if (!g) return;
    for (i = 0; i < g->n_sets * g->K; ++i) free(g->seq[i]);
    for (i = 0; i < g->K; ++i) free(g->xpos2[i]);
    free(g->seq); free(g->len); free(g->xpos2);
    free(g->bases); free(g->seq_off); free(g->name_off); free(g->names);
    free(g);
}

/* ---------------- reads ---------------- */
typedef struct { void *p; size_t n, cap, esz; } vec;
static void *vec_grow(vec *v, size_t add)
{
    if (v->n + add > v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 1024;
        while (nc < v->n + add) nc *= 2;
        v->p = realloc(v->p, nc * v->esz);
        v->cap = nc;
    }
    v->n += add;
    return (char *)v->p + (v->n - add) * v->esz;
}
#define VPUSH(v, T, x) (*(T *)vec_grow(&(v), 1) = (x))

struct spx_synth_reads {
    spx_batch bt;
    vec grp_first, qname_off, qnames, flag, tid, pos, l_qseq, n_cigar, cigar_off, seq_off, qual_off, cs_off, cigar, seq4,
        qual, cs, md_off, md;
};

typedef struct { /* one source base */
    int32_t b;      /* hap-1 column */
    int8_t ins_idx; /* -1: column base, else index among bases inserted after column b */
    char base;
    int8_t fate;    /* 0 kept, 1 substituted, 2 deleted */
    int32_t rb;     /* read index of the kept base, -1 if deleted */
    int32_t ins0, nins; /* read-error insertion after it: read indices [ins0, ins0+nins) */
} sbase_t;

typedef struct {
    char type;  /* 'M','I','D','S','H' */
    char r, t;  /* upper-case bases */
    int32_t rb; /* read base index (M,I,S,H) */
    int32_t tc; /* target coordinate (M,D) */
} ev_t;

static inline int nt16(char c)
{
    switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; default: return 15; }
}
static inline char lower(char c) { return (char)(c | 0x20); }

/* state of contig set `set` at hap-1 column b */
static inline void col_get(const spx_synth_genome *g, int set, int c, int32_t b, int *present, char *base, int32_t *coord,
                           int *nins)
{
    if (set == 1) {
        const int32_t *xp = g->xpos2[c];
        int32_t x = xp[b];
        *present = x >= 0;
        *coord = x;
        *base = x >= 0 ? g->seq[g->K + c][x] : 'N';
        *nins = (x >= 0 && b + 1 < g->len[c] && xp[b + 1] >= 0) ? xp[b + 1] - x - 1 : 0;
    } else {
        int inv = g->cfg.inverted_paralogs && set >= 2 && ((set - 2) & 1);
        int32_t n = g->len[set * g->K + c];
        *present = 1;
        *coord = b; /* coordinate in the un-inverted copy */
        *base = inv ? comp(g->seq[set * g->K + c][n - 1 - b]) : g->seq[set * g->K + c][b];
        *nins = 0;
    }
}

spx_synth_reads *spx_synth_reads_create(const spx_synth_genome *g, const spx_synth_cfg *cfg, int64_t first, int32_t n)
{
    spx_synth_reads *R = calloc(1, sizeof *R);
    int64_t gi;
    vec sb = {0, 0, 0, sizeof(sbase_t)}, rseq = {0, 0, 0, 1}, rq = {0, 0, 0, 1}, ev = {0, 0, 0, sizeof(ev_t)},
        tmp = {0, 0, 0, 1};
    R->grp_first.esz = 4; R->qname_off.esz = 8; R->qnames.esz = 1; R->flag.esz = 2; R->tid.esz = 4; R->pos.esz = 4;
    R->l_qseq.esz = 4; R->n_cigar.esz = 4; R->cigar_off.esz = 8; R->seq_off.esz = 8; R->qual_off.esz = 8;
    R->cs_off.esz = 8; R->cigar.esz = 4; R->seq4.esz = 1; R->qual.esz = 1; R->cs.esz = 1; R->md_off.esz = 8; R->md.esz = 1;

    for (gi = first; gi < first + n; ++gi) {
        rng_t r;
        int ont, rlen, c, h, rev, nsec, P, a, K = g->K;
        int32_t b0, b, n1 = g->len[0 * K];
        double p_sub, p_ins, p_del;
        int64_t next_sub, next_del, next_ins, si;
        char name[64];
        rng_seed(&r, cfg->seed, 1000000ULL + (uint64_t)gi);
        ont = cfg->platform == SPX_SYNTH_ONT || (cfg->platform == SPX_SYNTH_MIXED && rng_u(&r) < 0.3);
        if (cfg->read_len > 0) rlen = cfg->read_len;
        else {
            double u = rng_u(&r), l = 2000.0 * pow(1.0 - u, -1.0 / 1.2);
            rlen = l > cfg->max_read_len ? cfg->max_read_len : (int)l;
        }
        if (rlen > n1 / 2) rlen = n1 / 2;
        c = rng_below(&r, K);
        h = rng_next(&r) & 1;
        rev = rng_next(&r) & 1;
        nsec = cfg->min_secondaries + rng_below(&r, cfg->max_secondaries - cfg->min_secondaries + 1);
        if (nsec > 1 + cfg->n_paralogs) nsec = 1 + cfg->n_paralogs;
        P = (rng_next(&r) & 1) ? 1 - h : h; /* haplotype the primary is placed on */
        b0 = 300 + rng_below(&r, (uint32_t)(n1 - rlen - rlen / 8 - 1200));
        if (ont) { p_sub = 0.015; p_ins = 0.015; p_del = 0.02; }
        else { p_sub = 0.0002; p_ins = 0.0004; p_del = 0.0004; }

        /* --- source bases of haplotype h --- */
        sb.n = 0;
        for (b = b0; (int)sb.n < rlen; ++b) {
            int pr, ni, k;
            char bs;
            int32_t co;
            col_get(g, h, c, b, &pr, &bs, &co, &ni);
            if (pr) {
                sbase_t s = {b, -1, bs, 0, -1, 0, 0};
                VPUSH(sb, sbase_t, s);
            }
            for (k = 0; k < ni && (int)sb.n < rlen; ++k) {
                sbase_t s = {b, (int8_t)k, g->seq[K + c][co + 1 + k], 0, -1, 0, 0};
                VPUSH(sb, sbase_t, s);
            }
        }
        /* --- sequencing errors + qualities (shared by every alignment of the read) --- */
        rseq.n = rq.n = 0;
        next_sub = rng_gap(&r, p_sub);
        next_del = rng_gap(&r, p_del);
        next_ins = rng_gap(&r, p_ins);
        for (si = 0; si < (int64_t)sb.n; ++si) {
            sbase_t *s = (sbase_t *)sb.p + si;
            int errq = ont ? 3 + (int)rng_below(&r, 13) : 5 + (int)rng_below(&r, 16);
            int q;
            {
                double z = rng_normal(&r);
                q = ont ? (int)floor(22 + 8 * z + 0.5) : (int)floor(45 + 12 * z + 0.5);
                if (ont) { if (q < 1) q = 1; if (q > 50) q = 50; }
                else { if (q < 2) q = 2; if (q > 93) q = 93; }
            }
            {
                int is_del = si == next_del && si > 0 && si + 1 < (int64_t)sb.n, is_sub, is_ins;
                if (next_del <= si) next_del = si + 1 + rng_gap(&r, p_del);
                is_sub = si == next_sub && !is_del;
                if (next_sub <= si) next_sub = si + 1 + rng_gap(&r, p_sub);
                is_ins = si == next_ins && si + 1 < (int64_t)sb.n;
                if (next_ins <= si) next_ins = si + 1 + rng_gap(&r, p_ins);
                if (is_del) {
                    s->fate = 2;
                    s->rb = -1;
                } else {
                    char bs = s->base;
                    if (is_sub) {
                        bs = other_base(&r, bs);
                        s->fate = 1;
                        q = errq;
                    }
                    s->rb = (int32_t)rseq.n;
                    VPUSH(rseq, char, bs);
                    VPUSH(rq, uint8_t, (uint8_t)q);
                }
                if (is_ins) {
                    int l = 1, k;
                    if (ont) while (rng_u(&r) > 0.7 && l < 12) l++;
                    s = (sbase_t *)sb.p + si;
                    s->ins0 = (int32_t)rseq.n;
                    s->nins = l;
                    for (k = 0; k < l; ++k) {
                        VPUSH(rseq, char, ACGT[rng_below(&r, 4)]);
                        VPUSH(rq, uint8_t, (uint8_t)(ont ? 3 + rng_below(&r, 13) : 5 + rng_below(&r, 16)));
                    }
                }
            }
        }

        VPUSH(R->grp_first, int32_t, (int32_t)R->flag.n);
        VPUSH(R->qname_off, int64_t, (int64_t)R->qnames.n);
        {
            int l = sprintf(name, "read%010lld", (long long)gi);
            memcpy(vec_grow(&R->qnames, l + 1), name, l + 1);
        }

        /* record order: primary first unless shuffled */
        {
            int order[16], nrec = 1 + nsec, i;
            for (i = 0; i < nrec; ++i) order[i] = i;
            if (cfg->shuffle_records) {
                int j = rng_below(&r, nrec), t = order[0];
                order[0] = order[j]; order[j] = t;
            }
            for (a = 0; a < nrec; ++a) {
                int rec = order[a]; /* 0 primary, 1 other haplotype, 2.. paralogs */
                int set = rec == 0 ? P : rec == 1 ? 1 - P : rec;
                int inv = cfg->inverted_paralogs && set >= 2 && ((set - 2) & 1);
                int32_t tlen = g->len[set * K + c];
                int64_t e, nev, fm, lm;
                int do_clip = rng_u(&r) < cfg->softclip_frac, clip_len = 50 + rng_below(&r, 451),
                    clip_left = rng_next(&r) & 1, hard = rng_u(&r) < cfg->hardclip_frac;
                uint16_t flag;
                ev_t *E;
                /* --- columns -> events --- */
                ev.n = 0;
                si = 0;
                for (b = b0; si < (int64_t)sb.n; ++b) {
                    int tp, tni, k;
                    char tb;
                    int32_t tco;
                    const sbase_t *s;
                    col_get(g, set, c, b, &tp, &tb, &tco, &tni);
                    s = (const sbase_t *)sb.p + si;
                    if (s->b == b && s->ins_idx < 0) {
                        if (s->fate != 2) {
                            ev_t x = {tp ? 'M' : 'I', ((char *)rseq.p)[s->rb], tb, s->rb, tco};
                            VPUSH(ev, ev_t, x);
                        } else if (tp) {
                            ev_t x = {'D', 0, tb, -1, tco};
                            VPUSH(ev, ev_t, x);
                        }
                        for (k = 0; k < s->nins; ++k) {
                            ev_t x = {'I', ((char *)rseq.p)[s->ins0 + k], 0, s->ins0 + k, 0};
                            VPUSH(ev, ev_t, x);
                        }
                        si++;
                    } else if (tp) {
                        ev_t x = {'D', 0, tb, -1, tco};
                        VPUSH(ev, ev_t, x);
                    }
                    /* bases the source haplotype has inserted after this column */
                    while (si < (int64_t)sb.n && ((const sbase_t *)sb.p)[si].b == b &&
                           ((const sbase_t *)sb.p)[si].ins_idx >= 0) {
                        s = (const sbase_t *)sb.p + si;
                        if (set == h) { /* same haplotype: the inserted bases exist in the target too */
                            if (s->fate != 2) {
                                ev_t x = {'M', ((char *)rseq.p)[s->rb], s->base, s->rb, tco + 1 + s->ins_idx};
                                VPUSH(ev, ev_t, x);
                            } else {
                                ev_t x = {'D', 0, s->base, -1, tco + 1 + s->ins_idx};
                                VPUSH(ev, ev_t, x);
                            }
                        } else if (s->fate != 2) {
                            ev_t x = {'I', ((char *)rseq.p)[s->rb], 0, s->rb, 0};
                            VPUSH(ev, ev_t, x);
                        }
                        for (k = 0; k < s->nins; ++k) {
                            ev_t x = {'I', ((char *)rseq.p)[s->ins0 + k], 0, s->ins0 + k, 0};
                            VPUSH(ev, ev_t, x);
                        }
                        si++;
                    }
                    if (si >= (int64_t)sb.n) break;
                    /* bases the target has inserted after this column */
                    for (k = 0; k < tni && set != h; ++k) {
                        ev_t x = {'D', 0, g->seq[K + c][tco + 1 + k], -1, tco + 1 + k};
                        VPUSH(ev, ev_t, x);
                    }
                }
                E = (ev_t *)ev.p;
                nev = (int64_t)ev.n;
                if (inv) { /* target stored reverse-complemented */
                    int64_t i, j;
                    for (i = 0, j = nev - 1; i < j; ++i, --j) { ev_t t = E[i]; E[i] = E[j]; E[j] = t; }
                    for (i = 0; i < nev; ++i) {
                        E[i].r = comp(E[i].r);
                        E[i].t = comp(E[i].t);
                        E[i].tc = tlen - 1 - E[i].tc;
                    }
                }
                /* --- optional clip of 50-500 read bases on one side --- */
                if (do_clip) {
                    int cnt = 0;
                    /* a clip never swallows the alignment: at least 100 read bases (or half of a short read) stay
                     * aligned -- no aligner writes a record made of clips only */
                    { int keep = rlen / 2 < 100 ? rlen / 2 : 100; if (clip_len > rlen - keep) clip_len = rlen - keep; }
                    if (clip_left) {
                        for (e = 0; e < nev && cnt < clip_len; ++e)
                            if (E[e].type == 'M' || E[e].type == 'I') { E[e].type = 'S'; cnt++; }
                            else E[e].type = 0;
                    } else {
                        for (e = nev - 1; e >= 0 && cnt < clip_len; --e)
                            if (E[e].type == 'M' || E[e].type == 'I') { E[e].type = 'S'; cnt++; }
                            else E[e].type = 0;
                    }
                }
                /* --- trim so the aligned part starts and ends on an M column --- */
                for (fm = 0; fm < nev && E[fm].type != 'M'; ++fm)
                    if (E[fm].type == 'I') E[fm].type = 'S';
                    else if (E[fm].type == 'D') E[fm].type = 0;
                for (lm = nev - 1; lm > fm && E[lm].type != 'M'; --lm)
                    if (E[lm].type == 'I') E[lm].type = 'S';
                    else if (E[lm].type == 'D') E[lm].type = 0;
                if (hard)
                    for (e = 0; e < nev; ++e)
                        if (E[e].type == 'S') E[e].type = 'H';
                /* --- emit record --- */
                flag = (uint16_t)(((rev ^ inv) ? SPX_FREVERSE : 0) | (rec == 0 ? 0 : SPX_FSECONDARY));
                VPUSH(R->flag, uint16_t, flag);
                VPUSH(R->tid, int32_t, set * K + c);
                VPUSH(R->pos, int32_t, fm < nev ? E[fm].tc : 0);
                VPUSH(R->cigar_off, int64_t, (int64_t)R->cigar.n);
                VPUSH(R->seq_off, int64_t, (int64_t)R->seq4.n);
                VPUSH(R->qual_off, int64_t, (int64_t)R->qual.n);
                VPUSH(R->cs_off, int64_t, cfg->tag_mode == 1 ? (int64_t)-1 : (int64_t)R->cs.n);
                VPUSH(R->md_off, int64_t, cfg->tag_mode == 0 ? (int64_t)-1 : (int64_t)R->md.n);
                {
                    int ncig = 0, lq = 0, eqrun = 0, mdrun = 0, md_after_del = 0;
                    char buf[32];
                    const int want_cs = cfg->tag_mode != 1, want_md = cfg->tag_mode != 0;
                    const size_t cs_mark = R->cs.n;
                    tmp.n = 0; /* unpacked SEQ */
                    for (e = 0; e < nev;) {
                        char ty = E[e].type;
                        int64_t e2 = e;
                        int op;
                        if (ty == 0) { e++; continue; }
                        while (e2 < nev && (E[e2].type == ty || E[e2].type == 0)) e2++;
                        /* count real events in the run */
                        {
                            int64_t k;
                            int len = 0;
                            for (k = e; k < e2; ++k) if (E[k].type == ty) len++;
                            op = ty == 'M' ? SPX_CMATCH : ty == 'I' ? SPX_CINS : ty == 'D' ? SPX_CDEL
                                 : ty == 'S' ? SPX_CSOFT_CLIP : SPX_CHARD_CLIP;
                            VPUSH(R->cigar, uint32_t, (uint32_t)len << 4 | (uint32_t)op);
                            ncig++;
                            if (ty != 'D' && ty != 'M') md_after_del = 0; /* a separate CIGAR D op gets its own MD token */
                            if (ty == 'I') VPUSH(R->cs, char, '+');
                            if (ty == 'D') VPUSH(R->cs, char, '-');
                            for (k = e; k < e2; ++k) {
                                if (E[k].type != ty) continue;
                                if (ty == 'M' && want_md) { /* MD: match counts, mismatched / deleted reference bases */
                                    if (E[k].r == E[k].t) { mdrun++; md_after_del = 0; }
                                    else {
                                        int l2 = sprintf(buf, "%d%c", mdrun, E[k].t);
                                        memcpy(vec_grow(&R->md, l2), buf, l2);
                                        mdrun = 0; md_after_del = 0;
                                    }
                                }
                                if (ty == 'D' && want_md) {
                                    if (!md_after_del) { int l2 = sprintf(buf, "%d^", mdrun); memcpy(vec_grow(&R->md, l2), buf, l2); mdrun = 0; }
                                    VPUSH(R->md, char, E[k].t);
                                    md_after_del = 1;
                                }
                                if (ty == 'M') {
                                    if (E[k].r == E[k].t) eqrun++;
                                    else {
                                        if (eqrun) { int l = sprintf(buf, ":%d", eqrun); memcpy(vec_grow(&R->cs, l), buf, l); eqrun = 0; }
                                        VPUSH(R->cs, char, '*');
                                        VPUSH(R->cs, char, lower(E[k].t));
                                        VPUSH(R->cs, char, lower(E[k].r));
                                    }
                                } else if (ty == 'I') VPUSH(R->cs, char, lower(E[k].r));
                                else if (ty == 'D') VPUSH(R->cs, char, lower(E[k].t));
                                if (ty == 'M' || ty == 'I' || ty == 'S') {
                                    VPUSH(tmp, char, E[k].r);
                                    VPUSH(R->qual, uint8_t, ((uint8_t *)rq.p)[E[k].rb]);
                                    lq++;
                                }
                            }
                            if (ty == 'M' && eqrun) { int l = sprintf(buf, ":%d", eqrun); memcpy(vec_grow(&R->cs, l), buf, l); eqrun = 0; }
                        }
                        e = e2;
                    }
                    VPUSH(R->cs, char, 0);
                    if (!want_cs) R->cs.n = cs_mark; /* MD-only records carry no cs */
                    if (want_md) { int l2 = sprintf(buf, "%d", mdrun); memcpy(vec_grow(&R->md, l2 + 1), buf, l2 + 1); }
                    VPUSH(R->n_cigar, int32_t, ncig);
                    VPUSH(R->l_qseq, int32_t, lq);
                    {
                        int i2;
                        uint8_t *dst = vec_grow(&R->seq4, (size_t)(lq + 1) / 2);
                        const char *s = tmp.p;
                        for (i2 = 0; i2 + 1 < lq; i2 += 2) dst[i2 >> 1] = (uint8_t)(nt16(s[i2]) << 4 | nt16(s[i2 + 1]));
                        if (lq & 1) dst[lq >> 1] = (uint8_t)(nt16(s[lq - 1]) << 4);
                    }
                }
            }
        }
    }
    VPUSH(R->grp_first, int32_t, (int32_t)R->flag.n);
    free(sb.p); free(rseq.p); free(rq.p); free(ev.p); free(tmp.p);
    R->bt.n_groups = n;
    R->bt.n_alns = (int32_t)R->flag.n;
    R->bt.grp_first = R->grp_first.p; R->bt.qname_off = R->qname_off.p; R->bt.qnames = R->qnames.p;
    R->bt.flag = R->flag.p; R->bt.tid = R->tid.p; R->bt.pos = R->pos.p; R->bt.l_qseq = R->l_qseq.p;
    R->bt.n_cigar = R->n_cigar.p; R->bt.cigar_off = R->cigar_off.p; R->bt.seq_off = R->seq_off.p;
    R->bt.qual_off = R->qual_off.p; R->bt.cs_off = R->cs_off.p; R->bt.cigar = R->cigar.p; R->bt.seq4 = R->seq4.p;
    R->bt.qual = R->qual.p; R->bt.cs = R->cs.p;
    R->bt.md_off = R->md_off.p; R->bt.md = R->md.p;
    return R;
}

const spx_batch *spx_synth_reads_batch(const spx_synth_reads *r) { return &r->bt; }

void spx_synth_reads_free(spx_synth_reads *R)
{
    if (!R) return;
    free(R->grp_first.p); free(R->qname_off.p); free(R->qnames.p); free(R->flag.p); free(R->tid.p); free(R->pos.p);
    free(R->l_qseq.p); free(R->n_cigar.p); free(R->cigar_off.p); free(R->seq_off.p); free(R->qual_off.p);
    free(R->cs_off.p); free(R->cigar.p); free(R->seq4.p); free(R->qual.p); free(R->cs.p); free(R->md_off.p); free(R->md.p);
    free(R);
}
