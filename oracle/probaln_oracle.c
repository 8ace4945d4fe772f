/*
 * probaln_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into
 * the product library).
 *
 * CPU restatement of htslib-1.17 `probaln_glocal` (probaln.c), the banded
 * glocal profile-HMM forward/backward + MAP that secphase calls at
 *   /root/reference/programs/submodules/ptMarker/ptMarker.c:755-757
 * with parameters built at ptMarker.c:680,754.
 *
 * PARITY UNPINNED: htslib is a third-party dependency that is NOT vendored in
 * /root/reference (Dockerfile:17-25 pins release 1.17; programs/Makefile:4
 * links -lhts) and is absent from this machine, and the reference's own tests
 * (programs/src/secphase_test.c) hold no vector for this function.  This file
 * restates the PUBLISHED algorithm (htslib probaln.c, MIT/Expat, derived from
 * Heng Li's kprobaln.c) from its documented structure:
 *   - 3-state (M/I/D) profile HMM, band half-width bw, FP64 with per-row
 *     scaling s[i]; emission EM=.33333333333, EI=.25; float qual LUT
 *     10^(-q/10); float par_t{d,e}; transition matrix m[9]; bM,bI begin
 *     probabilities computed in float; sM=sI=1/(2L+2).
 *   - forward rows 1..L (row 1 divides by its sum, rows >=2 multiply by the
 *     reciprocal), terminal s[L+1]; backward rows L..1 scaled by 1/s[i];
 *     MAP over M and I states only; q = (int)(-4.343*log(1-max/sum)+.499),
 *     >100 -> 99.
 * It must be re-checked against the real htslib 1.17 source at the first
 * opportunity; until then every parity statement in this repository reads
 * "vs. our restatement of htslib-1.17 probaln_glocal".
 *
 * Build: gcc -O2 -ffp-contract=off (no -march=native / -mfma) so every FP64
 * operation is a separately rounded IEEE op, as in a stock x86-64 build of
 * htslib.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "secphase_oracle.h"

#define ORC_EI .25
#define ORC_EM .33333333333

/* band-relative slot of column k in row i (three doubles per slot, one spare
 * slot on either side) */
static inline int slot3(int bw, int i, int k)
{
    int x = i - bw;
    if (x < 0) x = 0;
    return (k - x + 1) * 3;
}

/* (int) of a double with x86 cvttsd2si semantics: out-of-range / NaN gives
 * INT_MIN ("integer indefinite").  The reference performs this conversion on
 * -4.343*log(1-max)+.499 which is +inf when the posterior rounds to 1 and NaN
 * when a row underflowed; C leaves that undefined, an x86-64 build yields
 * INT_MIN, and we fix that choice here so GPU and CPU agree. */
static inline int cvt_trunc_x86(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}

int orc_phred_from_posterior(double max_over_sum)
{
    int k = cvt_trunc_x86(-4.343 * log(1. - max_over_sum) + .499);
    return (uint8_t)(k > 100 ? 99 : k);
}

void orc_probaln_consts(int l_ref, int l_query, float d, float e, int set_q, orc_hmm_consts *c)
{
    double sM, sI;
    float qf = (float)pow(10, -set_q / 10.);
    sM = sI = 1. / (2 * l_query + 2);
    /* (1 - d - d) and (1 - e) are float expressions promoted afterwards */
    c->m[0] = (double)((1 - d) - d) * (1 - sM);
    c->m[1] = c->m[2] = (double)d * (1 - sM);
    c->m[3] = (double)(1 - e) * (1 - sI);
    c->m[4] = (double)e * (1 - sI);
    c->m[5] = 0.;
    c->m[6] = (double)(1 - e);
    c->m[7] = 0.;
    c->m[8] = (double)e;
    c->bM = (double)((1 - d) / l_ref); /* float division */
    c->bI = (double)(d / l_ref);
    c->sM = sM;
    c->sI = sI;
    c->e_match = 1. - (double)qf;
    c->e_mis = (double)qf * ORC_EM;
}

/* The reference calloc()s two (L+1) x (3*bw2+6) double matrices per call and frees them again; with many
 * worker threads that turns into mmap/munmap + page-fault contention.  For the CPU BASELINE the worker
 * threads may keep one zeroed-on-demand scratch block each (same values, same arithmetic; only the
 * allocator traffic goes away).  Off by default = the reference's allocation pattern. */
static int g_reuse_scratch = 0;
void orc_set_scratch_reuse(int on) { g_reuse_scratch = on; }
static __thread double *t_buf = NULL;
static __thread size_t t_cap = 0;
static double *scratch_zeroed(size_t n_doubles)
{
    if (n_doubles > t_cap) {
        free(t_buf);
        t_cap = n_doubles + n_doubles / 4;
        t_buf = malloc(t_cap * sizeof(double));
        if (!t_buf) { t_cap = 0; return NULL; }
    }
    memset(t_buf, 0, n_doubles * sizeof(double));
    return t_buf;
}

/* which reading of the terminal guard the oracle follows: 0 = `u >= bw2*3+3` (default), 1 = `u >= i_dim-3`.  Process-wide; initial
 * value from SPX_TERMINAL_GUARD (band|row), like the product's switch (include/spx.h spx_set_terminal_guard). */
static _Atomic int g_term_guard = -1; /* (atomic: worker threads of orc_run_batch read it, the first of them initialises it) */
int orc_get_terminal_guard(void)
{
    int v = g_term_guard;
    if (v < 0) {
        const char *e = getenv("SPX_TERMINAL_GUARD");
        v = (e && (!strcmp(e, "row") || !strcmp(e, "1") || !strcmp(e, "idim"))) ? 1 : 0;
        g_term_guard = v;
    }
    return v;
}
void orc_set_terminal_guard(int reading) { g_term_guard = reading ? 1 : 0; }

static double *g_cap_s = NULL, *g_cap_zM = NULL, *g_cap_zI = NULL; /* set by orc_probaln_posteriors (single-threaded tests) */

int orc_probaln_glocal(const uint8_t *ref, int l_ref, const uint8_t *query, int l_query,
                       const uint8_t *iqual, const orc_probaln_par *c, int *state, uint8_t *q)
{
    double *f = NULL, *b = NULL, *s = NULL, m[9], sI, sM, bI, bM;
    float *qual = NULL;
    int bw, bw2, i, k, is_backward, Pr;
    size_t i_dim;

    if (l_ref <= 0 || l_query <= 0) return 0;

    is_backward = (state && q) ? 1 : 0;
    bw = l_ref > l_query ? l_ref : l_query;
    if (bw > c->bw) bw = c->bw;
    if (bw < abs(l_ref - l_query)) bw = abs(l_ref - l_query);
    bw2 = bw * 2 + 1;
    i_dim = bw2 < l_ref ? (size_t)bw2 * 3 + 6 : (size_t)l_ref * 3 + 6;

    /* The reference reads bi1[v11] for k = l_ref (multiplied by 0: probaln.c, `e = (k >= l_ref? 0 : ...) * bi1[v11]`); when the
     * band covers the whole reference (i_dim = 3*l_ref + 6) that slot is the first one behind the row, and for row l_query the
     * first double behind the matrix: a heap over-read in the reference, harmless there as long as the bytes behind happen to
     * be a finite double (0 * x = 0).  Here the matrices get a zeroed tail, so that the oracle's result does not depend on the
     * allocator (found when the CPU fuzz ran under ThreadSanitizer's allocator: NaN behind the block, every state -1). */
    const size_t tail = 8;
    const int reuse = g_reuse_scratch;
    if (reuse) {
        size_t nm = (size_t)(l_query + 1) * i_dim + tail;
        double *blk = scratch_zeroed(nm * (is_backward ? 2 : 1) + (size_t)l_query + 2 + ((size_t)l_query + 1) / 2 + 1);
        if (!blk) return INT_MIN;
        f = blk;
        if (is_backward) b = blk + nm;
        s = blk + nm * (is_backward ? 2 : 1);
        qual = (float *)(s + l_query + 2);
    } else {
        f = calloc((size_t)(l_query + 1) * i_dim + tail, sizeof(double));
        if (is_backward) b = calloc((size_t)(l_query + 1) * i_dim + tail, sizeof(double));
        s = calloc((size_t)l_query + 2, sizeof(double));
        qual = calloc((size_t)l_query, sizeof(float));
        if (!f || (is_backward && !b) || !s || !qual) {
            free(f); free(b); free(s); free(qual);
            return INT_MIN;
        }
    }
    for (i = 0; i < l_query; ++i) qual[i] = (float)pow(10, -(iqual ? iqual[i] : 30) / 10.);

    sM = sI = 1. / (2 * l_query + 2);
    m[0] = (double)((1 - c->d) - c->d) * (1 - sM);
    m[1] = m[2] = (double)c->d * (1 - sM);
    m[3] = (double)(1 - c->e) * (1 - sI);
    m[4] = (double)c->e * (1 - sI);
    m[5] = 0.;
    m[6] = (double)(1 - c->e);
    m[7] = 0.;
    m[8] = (double)c->e;
    bM = (double)((1 - c->d) / l_ref);
    bI = (double)(c->d / l_ref);

    /*** forward ***/
    f[slot3(bw, 0, 0)] = s[0] = 1.;
    { /* row 1 */
        double *fi = f + i_dim, sum = 0.;
        int beg = 1, end = l_ref < bw + 1 ? l_ref : bw + 1, lo, hi;
        for (k = beg; k <= end; ++k) {
            int u = slot3(bw, 1, k);
            double e = (ref[k - 1] > 3 || query[0] > 3) ? 1.
                       : ref[k - 1] == query[0]         ? 1. - qual[0]
                                                        : qual[0] * ORC_EM;
            fi[u + 0] = e * bM;
            fi[u + 1] = ORC_EI * bI;
            sum += fi[u] + fi[u + 1];
        }
        s[1] = sum;
        lo = slot3(bw, 1, beg);
        hi = slot3(bw, 1, end) + 2;
        for (k = lo; k <= hi; ++k) fi[k] /= sum;
    }
    for (i = 2; i <= l_query; ++i) {
        double *fi = f + (size_t)i * i_dim, *fp = f + (size_t)(i - 1) * i_dim, sum = 0., qli = qual[i - 1];
        int beg = 1, end = l_ref, x, lo, hi;
        uint8_t qyi = query[i - 1];
        x = i - bw; if (beg < x) beg = x;
        x = i + bw; if (end > x) end = x;
        for (k = beg; k <= end; ++k) {
            int u = slot3(bw, i, k), v11 = slot3(bw, i - 1, k - 1), v10 = slot3(bw, i - 1, k),
                v01 = slot3(bw, i, k - 1);
            double e = (ref[k - 1] > 3 || qyi > 3) ? 1. : ref[k - 1] == qyi ? 1. - qli : qli * ORC_EM;
            fi[u + 0] = e * (m[0] * fp[v11 + 0] + m[3] * fp[v11 + 1] + m[6] * fp[v11 + 2]);
            fi[u + 1] = ORC_EI * (m[1] * fp[v10 + 0] + m[4] * fp[v10 + 1]);
            fi[u + 2] = m[2] * fi[v01 + 0] + m[8] * fi[v01 + 2];
            sum += fi[u] + fi[u + 1] + fi[u + 2];
        }
        s[i] = sum;
        lo = slot3(bw, i, beg);
        hi = slot3(bw, i, end) + 2;
        sum = 1. / sum;
        for (k = lo; k <= hi; ++k) fi[k] *= sum;
    }
    /* GUARD VARIANTS (parity unpinned): by default this restatement skips a column of the termination when its slot lies
     * outside the band, `u < 3 || u >= bw2*3+3` -- kprobaln's test.  The other reading of the htslib releases that shrank the
     * row to i_dim = min(bw2, l_ref)*3+6 is `u >= i_dim-3`; it differs exactly when l_query <= bw and 2*bw+1 > l_ref (set_u's
     * row offset is 0, u = 3*(k+1), i_dim-3 = 3*l_ref+3): it drops column l_ref from this sum and from the backward start
     * below.  secphase reaches that regime (--ont -b 50, blocks of <= 50 bases).  Only a real htslib 1.17 decides
     * (tools/pin_htslib: 240 problems of that shape; the tool reports which reading matches); orc_set_terminal_guard() /
     * SPX_TERMINAL_GUARD select the reading here, and the same switch exists in the product (include/spx.h). */
    const size_t guard_limit = orc_get_terminal_guard() ? i_dim - 3 : (size_t)bw2 * 3 + 3;
    { /* terminal */
        double sum = 0.;
        for (k = 1; k <= l_ref; ++k) {
            int u = slot3(bw, l_query, k);
            if (u < 3 || (size_t)u >= guard_limit) continue;
            sum += f[(size_t)l_query * i_dim + u + 0] * sM + f[(size_t)l_query * i_dim + u + 1] * sI;
        }
        s[l_query + 1] = sum;
    }
    { /* likelihood */
        double p = 1., Pr1 = 0.;
        for (i = 0; i <= l_query + 1; ++i) {
            p *= s[i];
            if (p < 1e-100) Pr1 += -4.343 * log(p), p = 1.;
        }
        Pr1 += -4.343 * log(p * l_ref * l_query);
        Pr = (int)(Pr1 + .499);
        if (!is_backward) {
            if (!reuse) { free(f); free(s); free(qual); }
            return Pr;
        }
    }
    /*** backward ***/
    for (k = 1; k <= l_ref; ++k) {
        int u = slot3(bw, l_query, k);
        double *bi = b + (size_t)l_query * i_dim;
        if (u < 3 || (size_t)u >= guard_limit) continue; /* (same guard as the terminal sum: see GUARD VARIANTS above) */
        bi[u + 0] = sM / s[l_query] / s[l_query + 1];
        bi[u + 1] = sI / s[l_query] / s[l_query + 1];
    }
    for (i = l_query - 1; i >= 1; --i) {
        int beg = 1, end = l_ref, x, lo, hi;
        double *bi = b + (size_t)i * i_dim, *bn = b + (size_t)(i + 1) * i_dim, y = (i > 1), qli1 = qual[i];
        uint8_t qyi1 = query[i];
        x = i - bw; if (beg < x) beg = x;
        x = i + bw; if (end > x) end = x;
        for (k = end; k >= beg; --k) {
            int u = slot3(bw, i, k), v11 = slot3(bw, i + 1, k + 1), v10 = slot3(bw, i + 1, k),
                v01 = slot3(bw, i, k + 1);
            double e = (k >= l_ref                       ? 0
                        : (ref[k] > 3 || qyi1 > 3)       ? 1.
                        : ref[k] == qyi1                 ? 1. - qli1
                                                         : qli1 * ORC_EM) *
                       bn[v11];
            bi[u + 0] = e * m[0] + ORC_EI * m[1] * bn[v10 + 1] + m[2] * bi[v01 + 2];
            bi[u + 1] = e * m[3] + ORC_EI * m[4] * bn[v10 + 1];
            bi[u + 2] = (e * m[6] + m[8] * bi[v01 + 2]) * y;
        }
        lo = slot3(bw, i, beg);
        hi = slot3(bw, i, end) + 2;
        y = 1. / s[i];
        for (k = lo; k <= hi; ++k) bi[k] *= y;
    }
    if (g_cap_s) { /* orc_probaln_posteriors: s[], and z = f*b (M, I) of every in-band cell */
        memcpy(g_cap_s, s, sizeof(double) * ((size_t)l_query + 2));
        for (i = 1; i <= l_query; ++i) {
            int lo2 = i - bw > 1 ? i - bw : 1, hi2 = i + bw < l_ref ? i + bw : l_ref;
            for (k = 1; k <= l_ref; ++k) {
                size_t at = (size_t)(i - 1) * l_ref + (k - 1);
                g_cap_zM[at] = g_cap_zI[at] = 0.;
                if (k >= lo2 && k <= hi2) {
                    int u = slot3(bw, i, k);
                    g_cap_zM[at] = f[(size_t)i * i_dim + u] * b[(size_t)i * i_dim + u];
                    g_cap_zI[at] = f[(size_t)i * i_dim + u + 1] * b[(size_t)i * i_dim + u + 1];
                }
            }
        }
    }
    /*** MAP ***/
    for (i = 1; i <= l_query; ++i) {
        double sum = 0., *fi = f + (size_t)i * i_dim, *bi = b + (size_t)i * i_dim, max = 0.;
        int beg = 1, end = l_ref, x, max_k = -1;
        x = i - bw; if (beg < x) beg = x;
        x = i + bw; if (end > x) end = x;
        for (k = beg; k <= end; ++k) {
            int u = slot3(bw, i, k);
            double z;
            z = fi[u + 0] * bi[u + 0];
            if (z > max) max = z, max_k = (k - 1) << 2 | 0;
            sum += z;
            z = fi[u + 1] * bi[u + 1];
            if (z > max) max = z, max_k = (k - 1) << 2 | 1;
            sum += z;
        }
        max /= sum;
        state[i - 1] = max_k;
        q[i - 1] = (uint8_t)orc_phred_from_posterior(max);
    }
    if (!reuse) { free(f); free(b); free(s); free(qual); }
    return Pr;
}

/* test diagnostics: the scaling factors s[0..l_query+1] and the posterior products z = f*b of the M and I states,
 * row major [l_query][l_ref] (0 outside the band), with which the product's kernels are compared bit for bit */
int orc_probaln_posteriors(const uint8_t *ref, int l_ref, const uint8_t *query, int l_query, const uint8_t *iqual,
                           const orc_probaln_par *c, double *s, double *zM, double *zI)
{
    int *state = malloc(sizeof(int) * (size_t)(l_query > 0 ? l_query : 1)), rc;
    uint8_t *q = malloc((size_t)(l_query > 0 ? l_query : 1));
    g_cap_s = s; g_cap_zM = zM; g_cap_zI = zI;
    rc = orc_probaln_glocal(ref, l_ref, query, l_query, iqual, c, state, q);
    g_cap_s = g_cap_zM = g_cap_zI = NULL;
    free(state); free(q);
    return rc;
}
