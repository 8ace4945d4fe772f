/*
 * fastdp_study.c -- DESIGN STUDY / TEST INFRASTRUCTURE: runs the fast-tier model (fastdp_model.c) beside the CPU oracle
 * (oracle/probaln_oracle.c) over the DP problems of a host plan and counts flagged rows / problems, unflagged rows whose
 * (state, q) differ from the oracle's, and the deviation of the row-normalised posteriors.  Driven by tools/fastdp_study.py.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/secphase_oracle.h"
#include "fastdp_model.h"

typedef struct fdp_study_in {
    int32_t n_problems;
    const int32_t *L, *R, *bw, *ref_tid, *ref_rfs;
    const int64_t *qry_nib;
    const uint8_t *qry4;
    const double *hmm; /* [n][16] */
    const int32_t *row_off, *n_rows_of, *rows;
    const char *bases;       /* spx_ref.bases */
    const int64_t *seq_off;  /* spx_ref.seq_off */
    int32_t set_q;
    float d, e;
    const double *thr; /* 102 */
    int32_t threads, dev_every;
} fdp_study_in;

typedef struct fdp_study_out {
    int64_t n_problems, n_rows_all, n_rows_wanted;
    int64_t rows_flagged_all, rows_flagged_wanted;
    int64_t rows_by_reason[8];          /* all rows, per flag bit */
    int64_t unflagged_mismatch_all;     /* MUST be 0 */
    int64_t unflagged_mismatch_wanted;  /* MUST be 0 */
    int64_t flagged_but_equal_all;      /* flagged rows whose fast answer was right anyway */
    int64_t problems_flagged_wanted, problems_flagged_all, problems_model;
    int64_t dev_rows;
    double max_dev;      /* max over sampled rows / cells with normalised z >= 1e-30 of |zf/sum_f - ze/sum_e| / (ze/sum_e) */
    double max_dev_over_delta;
    int64_t x_hist[20];  /* wanted rows: floor(-log10(x)) of the fast tier's 1 - max/sum (19: smaller / zero) */
    int64_t first_bad_problem;
} fdp_study_out;

typedef struct {
    const fdp_study_in *in;
    fdp_study_out out;
    int t, step, serial_sampling;
} worker_t;

static void *work(void *vp)
{
    worker_t *w = vp;
    const fdp_study_in *in = w->in;
    fdp_study_out *o = &w->out;
    o->first_bad_problem = -1;
    int cap = 0;
    uint8_t *ref = NULL, *qry = NULL, *iq = NULL, *oq = NULL, *fq = NULL, *fl = NULL;
    int *ost = NULL, *fst = NULL, *allrows = NULL;
    double *xs = NULL;
    for (int p = w->t; p < in->n_problems; p += w->step) {
        const int L = in->L[p], R = in->R[p], bw = in->bw[p];
        const int need = (L > R ? L : R) + 8;
        if (need > cap) {
            cap = need * 2;
            free(ref); free(qry); free(iq); free(oq); free(fq); free(fl); free(ost); free(fst); free(allrows); free(xs);
            ref = malloc(cap); qry = malloc(cap); iq = malloc(cap); oq = malloc(cap); fq = malloc(cap); fl = malloc(cap);
            ost = malloc(sizeof(int) * cap); fst = malloc(sizeof(int) * cap); allrows = malloc(sizeof(int) * cap);
            xs = malloc(sizeof(double) * cap);
        }
        const char *b = in->bases + in->seq_off[in->ref_tid[p]] + in->ref_rfs[p];
        for (int k = 0; k < R; ++k) ref[k] = orc_nt16_int[orc_nt16_table[(unsigned char)b[k]]];
        for (int i = 0; i < L; ++i) {
            const int64_t a = in->qry_nib[p] + i;
            const uint8_t by = in->qry4[a >> 1];
            qry[i] = (a & 1) ? (by >> 4) : (by & 0xf);
            iq[i] = (uint8_t)in->set_q;
            allrows[i] = i + 1;
        }
        orc_probaln_par par = {in->d, in->e, bw};
        orc_probaln_glocal(ref, R, qry, L, iq, &par, ost, oq);
        const double *h = in->hmm + (size_t)p * 16;
        const int W = 2 * bw + 1;
        const int sample = w->serial_sampling && (size_t)L * W < (size_t)40 << 20;
        double *zf = sample ? malloc(sizeof(double) * (size_t)L * 2 * W) : NULL;
        const int any_all = fdp_glocal(ref, R, qry, L, h, bw, L, allrows, in->thr, fst, fq, fl, zf, xs);
        if (w->serial_sampling) goto sampling;
        o->n_problems++;
        o->n_rows_all += L;
        if (any_all & FDP_F_MODEL) o->problems_model++;
        if (any_all) o->problems_flagged_all++;
        for (int i = 0; i < L; ++i) {
            const int same = fst[i] == ost[i] && fq[i] == oq[i];
            if (fl[i]) {
                o->rows_flagged_all++;
                for (int bit = 0; bit < 8; ++bit) if (fl[i] >> bit & 1) o->rows_by_reason[bit]++;
                if (same) o->flagged_but_equal_all++;
            } else if (!same) {
                o->unflagged_mismatch_all++;
                if (o->first_bad_problem < 0) o->first_bad_problem = p;
            }
        }
        /* the plan's own wanted rows */
        const int nw = in->n_rows_of[p];
        const int32_t *wr = in->rows + in->row_off[p];
        int pf = 0;
        if (nw > 0) {
            int *st2 = malloc(sizeof(int) * nw);
            uint8_t *q2 = malloc(nw), *f2 = malloc(nw);
            double *x2 = malloc(sizeof(double) * nw);
            pf = fdp_glocal(ref, R, qry, L, h, bw, nw, (const int *)wr, in->thr, st2, q2, f2, NULL, x2);
            for (int k = 0; k < nw; ++k) {
                const int i = wr[k] - 1;
                o->n_rows_wanted++;
                if (f2[k]) o->rows_flagged_wanted++;
                else if (st2[k] != ost[i] || q2[k] != oq[i]) {
                    o->unflagged_mismatch_wanted++;
                    if (o->first_bad_problem < 0) o->first_bad_problem = p;
                }
                int hb = 19;
                if (x2[k] > 0) { hb = (int)floor(-log10(x2[k])); if (hb < 0) hb = 0; if (hb > 19) hb = 19; }
                o->x_hist[hb]++;
            }
            free(st2); free(q2); free(f2); free(x2);
        }
        if (pf) o->problems_flagged_wanted++;
    sampling:
        if (sample && !(any_all & (FDP_F_MODEL | FDP_F_RANGE))) {
            double *s = malloc(sizeof(double) * ((size_t)L + 2)), *zM = malloc(sizeof(double) * (size_t)L * R),
                   *zI = malloc(sizeof(double) * (size_t)L * R);
            orc_probaln_posteriors(ref, R, qry, L, iq, &par, s, zM, zI);
            const double delta = fdp_delta(L, R, W);
            for (int i = 1; i <= L; ++i) {
                double se = 0, sf = 0;
                const double *zo = zf + (size_t)(i - 1) * 2 * W;
                for (int j = 0; j < W; ++j) {
                    const int k = i - bw + j;
                    if (k < 1 || k > R) continue;
                    se += zM[(size_t)(i - 1) * R + k - 1] + zI[(size_t)(i - 1) * R + k - 1];
                    sf += zo[j] + zo[W + j];
                }
                if (!(se > 0) || !(sf > 0)) continue;
                o->dev_rows++;
                for (int j = 0; j < W; ++j) {
                    const int k = i - bw + j;
                    if (k < 1 || k > R) continue;
                    const double e0 = zM[(size_t)(i - 1) * R + k - 1] / se, e1 = zI[(size_t)(i - 1) * R + k - 1] / se;
                    const double f0 = zo[j] / sf, f1 = zo[W + j] / sf;
                    if (e0 >= 1e-30) { const double dv = fabs(f0 - e0) / e0; if (dv > o->max_dev) o->max_dev = dv; if (dv / delta > o->max_dev_over_delta) o->max_dev_over_delta = dv / delta; }
                    if (e1 >= 1e-30) { const double dv = fabs(f1 - e1) / e1; if (dv > o->max_dev) o->max_dev = dv; if (dv / delta > o->max_dev_over_delta) o->max_dev_over_delta = dv / delta; }
                }
            }
            free(s); free(zM); free(zI);
        }
        free(zf);
    }
    free(ref); free(qry); free(iq); free(oq); free(fq); free(fl); free(ost); free(fst); free(allrows); free(xs);
    return NULL;
}

int fdp_study(const fdp_study_in *in, fdp_study_out *out)
{
    const int T = in->threads > 0 ? in->threads : 1;
    fdp_study_in local = *in;
    local.threads = T;
    worker_t *ws = calloc(T, sizeof *ws);
    pthread_t *th = calloc(T, sizeof *th);
    for (int t = 0; t < T; ++t) { ws[t].in = &local; ws[t].t = t; ws[t].step = T; }
    for (int t = 0; t < T; ++t) pthread_create(&th[t], NULL, work, &ws[t]);
    for (int t = 0; t < T; ++t) pthread_join(th[t], NULL);
    /* orc_probaln_posteriors keeps its capture pointers in globals: the sampled problems run afterwards, on this thread alone */
    worker_t samp;
    memset(&samp, 0, sizeof samp);
    samp.in = &local; samp.t = 0; samp.step = local.dev_every > 0 ? local.dev_every : local.n_problems + 1; samp.serial_sampling = 1;
    if (local.dev_every > 0) work(&samp);
    memset(out, 0, sizeof *out);
    out->dev_rows = samp.out.dev_rows; out->max_dev = samp.out.max_dev; out->max_dev_over_delta = samp.out.max_dev_over_delta;
    out->first_bad_problem = -1;
    for (int t = 0; t < T; ++t) {
        const fdp_study_out *o = &ws[t].out;
        out->n_problems += o->n_problems; out->n_rows_all += o->n_rows_all; out->n_rows_wanted += o->n_rows_wanted;
        out->rows_flagged_all += o->rows_flagged_all; out->rows_flagged_wanted += o->rows_flagged_wanted;
        for (int b = 0; b < 8; ++b) out->rows_by_reason[b] += o->rows_by_reason[b];
        out->unflagged_mismatch_all += o->unflagged_mismatch_all; out->unflagged_mismatch_wanted += o->unflagged_mismatch_wanted;
        out->flagged_but_equal_all += o->flagged_but_equal_all;
        out->problems_flagged_wanted += o->problems_flagged_wanted; out->problems_flagged_all += o->problems_flagged_all;
        out->problems_model += o->problems_model;
        out->dev_rows += o->dev_rows;
        if (o->max_dev > out->max_dev) out->max_dev = o->max_dev;
        if (o->max_dev_over_delta > out->max_dev_over_delta) out->max_dev_over_delta = o->max_dev_over_delta;
        for (int b = 0; b < 20; ++b) out->x_hist[b] += o->x_hist[b];
        if (out->first_bad_problem < 0 && o->first_bad_problem >= 0) out->first_bad_problem = o->first_bad_problem;
    }
    free(ws); free(th);
    return 0;
}
