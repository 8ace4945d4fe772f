/*
 * fastdp_model.c -- DESIGN STUDY / TEST INFRASTRUCTURE (never linked into libspx.so).
 *
 * CPU model of the FAST tier of the two-tier banded-HMM DP: the same real-number model as
 * htslib-1.17 probaln_glocal as secphase calls it (/root/reference/programs/submodules/ptMarker/ptMarker.c:755-757;
 * restated in oracle/probaln_oracle.c), evaluated in a DIFFERENT arithmetic -- FMA contraction allowed, no per-row sums,
 * power-of-two block rescaling, states carried as two combined rows (U, V) -- plus the CERTIFICATE that decides, per
 * wanted row, whether the exact tier's (state, q) is implied by the fast values.  Rows that are not certified flag their
 * problem, which the product then re-runs through the exact (bit-exact) kernels.
 *
 * The HIP kernels (secphase_amd/csrc/spx_fast_kernels.hip) follow this file's formulation; tools/fastdp_study.py runs it
 * beside the oracle on generator problems and counts (a) flagged rows / problems, (b) UNFLAGGED rows whose (state, q)
 * differ from the oracle's (must be 0), (c) the largest deviation of the row-normalised posteriors.
 *
 * Formulation.  With M, I, D the forward states of probaln_glocal (row i, column k; e(i,k) the emission; m[] the
 * transition constants; EI = 0.25):
 *     M(i,k) = e(i,k) * (m0 M(i-1,k-1) + m3 I(i-1,k-1) + m6 D(i-1,k-1))
 *     I(i,k) = EI * (m1 M(i-1,k) + m4 I(i-1,k))
 *     D(i,k) = m2 M(i,k-1) + m8 D(i,k-1)
 * carry   It = I / (EI m1),  Dt = D / m2,  Ut = (m0 M + m3 I + m6 D) / (m6 m2):
 *     M(i,k)  = (e(i,k) m6 m2) * Ut(i-1,k-1)                                1 mul (+ select of the constant)
 *     It(i,k) = Vt(i-1,k),  Vt(i,k) = M(i,k) + (EI m4) It(i,k)              1 fma
 *     Dt(i,k) = M(i,k-1) + m8 Dt(i,k-1)                                     1 fma  (running value, not stored)
 *     Ut(i,k) = cU0 M(i,k) + cU1 It(i,k) + Dt(i,k)                          2 fma
 * Two rows (Ut, Vt) live across rows; per-row sums are not needed because argmax_k z and max z / sum z of a row do not
 * change when the row is multiplied by a constant -- the rows are only kept in range by a power-of-two factor every
 * FDP_RESCALE rows.  Backward, with X = e(i+1,k+1) bM(i+1,k+1), Y = bI(i+1,k):
 *     bM = m0 X + EI m1 Y + m2 bD(i,k+1),  bI = m3 X + EI m4 Y,  bD = m6 X + m8 bD(i,k+1)
 * carry   Bm = bM / m0,  Bi = bI / m3,  Dt = bD / m6:
 *     X = (e m0) Bm(i+1,k+1);  Dt = X + m8 Dt(k+1);  Bm = X + cB1 Bi(i+1,k) + cB2 Dt(k+1);  Bi = X + (EI m4) Bi(i+1,k)
 * z_M = M Bm (x m0), z_I = I Bi (x m3): the saved forward row holds M and rho It with rho = EI m1 m3 / m0.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fastdp_model.h"

#define EI 0.25
/* the rows saved at the wanted rows (forward M, I; backward Bm, Bi) are kept as FP32 (half the scratch, half the MAP kernel's traffic): each
 * posterior product takes two such roundings, (1 + 2^-24)^2 - 1 < 2^-22 = FDP_DELTA_STORE, and values below the FP32 range flush to zero
 * (a row that loses its mass that way has no largest product and is flagged) */
#if FDP_STORE_FLOAT
#define FDP_STORE(x) ((double)(float)(x))
#else
#define FDP_STORE(x) (x)
#endif
#define U53 1.1102230246251565e-16 /* 2^-53 */

enum { H_M0 = 0, H_M1, H_M2, H_M3, H_M4, H_M6, H_M8, H_BM, H_BI, H_SM, H_SI, H_EMATCH, H_EMIS, H_HASN, H_TDROP };

static inline double pow2i(int e) { return ldexp(1.0, e); }

/* exponent such that x * 2^-exp is in [1, 2); x > 0 finite */
static inline int expo(double x)
{
    int e;
    frexp(x, &e);
    return e - 1;
}

int fdp_phred(double x, const double *thr)
{
    if (!(x > 0.0)) return 0;
    int lo = 0, hi = 101;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (x <= thr[mid]) lo = mid; else hi = mid - 1;
    }
    return lo > 100 ? 99 : lo;
}

double fdp_delta(int L, int R, int W) { return FDP_DELTA_PER_STEP * ((double)L + R + W + 16) * U53 + FDP_DELTA_STORE; }

/* certificate of one row from its posterior products z[0..n) (column order: M, I per column), see fastdp_model.h */
int fdp_certify(const double *z, int n, int k0, double delta, double A, const double *thr, int *state, int *q, double *x_out)
{
    int bi = -1;
    double best = 0.0, second = 0.0;
    for (int t = 0; t < n; ++t) {
        const double v = z[t];
        if (!(v >= 0.0) || v > 1.7e308) { *state = -1; *q = 0; return FDP_F_RANGE; }
        if (v > best) { second = best; best = v; bi = t; }
        else if (v > second) second = v;
    }
    if (bi < 0) { *state = -1; *q = 0; return FDP_F_RANGE; }
    double others = 0.0;
    for (int t = 0; t < n; ++t) if (t != bi) others += z[t];
    const double x = others / (best + others);
    if (x_out) *x_out = x;
    int flag = 0;
    if (!(best * (1.0 - delta) > second * (1.0 + delta))) flag |= FDP_F_ARGMAX;
    const double x_lo = x * (1.0 - 2.0 * delta) - A, x_hi = x * (1.0 + 2.0 * delta) + A;
    const int q_hi = fdp_phred(x_lo, thr), q_lo = fdp_phred(x_hi, thr);
    if (q_hi != q_lo) flag |= (x_lo <= 0.0 ? FDP_F_XSMALL : FDP_F_THRESH);
    *state = ((k0 + (bi >> 1) - 1) << 2) | (bi & 1);
    *q = fdp_phred(x, thr);
    return flag;
}

int fdp_glocal(const uint8_t *ref, int R, const uint8_t *qry, int L, const double *h, int bw, int n_rows, const int *rows,
               const double *thr, int *state, uint8_t *q, uint8_t *flag, double *z_out, double *x_out)
{
    if (L <= 0 || R <= 0) return -1;
    const int W = 2 * bw + 1;
    const double m0 = h[H_M0], m1 = h[H_M1], m2 = h[H_M2], m3 = h[H_M3], m4 = h[H_M4], m6 = h[H_M6], m8 = h[H_M8];
    const int Rt = h[H_TDROP] != 0.0 ? R - 1 : R;
    int pflag = 0;
    /* model conditions of the fast tier: every constant strictly positive and finite (all-nonnegative arithmetic: no
     * cancellation anywhere), no ambiguous base */
    if (h[H_HASN] != 0.0) pflag |= FDP_F_MODEL;
    {
        const double cs[] = {m0, m1, m2, m3, m4, m6, m8, h[H_BM], h[H_BI], h[H_SM], h[H_SI], h[H_EMATCH], h[H_EMIS]};
        for (unsigned t = 0; t < sizeof cs / sizeof cs[0]; ++t)
            if (!(cs[t] > 1e-30 && cs[t] < 1e30)) pflag |= FDP_F_MODEL;
        const double mu = fmin(m0 * h[H_EMIS], EI * m4);
        if (!(mu >= pow2i(-FDP_MU_BITS))) pflag |= FDP_F_MODEL;
    }
    if (pflag) {
        for (int w = 0; w < n_rows; ++w) flag[w] = (uint8_t)pflag;
        return pflag;
    }
    /* derived constants (each a few roundings off its real value: counted in delta) */
    const double ups = m6 * m2;
    const double emU = h[H_EMATCH] * ups, exU = h[H_EMIS] * ups; /* M = e * (m6 m2) * Ut */
    const double c4 = EI * m4;
    const double gam = EI * m1;             /* I = gam * It */
    const double cU0 = m0 / ups, cU1 = (m3 * gam) / ups;
    const double rho = (gam * m3) / m0;     /* z_I / z_M correction */
    const double emB = h[H_EMATCH] * m0, exB = h[H_EMIS] * m0; /* X = e * m0 * Bm */
    const double cB1 = (gam * m3) / m0, cB2 = (m2 * m6) / m0;

    /* band slot j of row i <-> column k = i - bw + j */
    double *U = calloc((size_t)W + 2, sizeof(double)), *V = calloc((size_t)W + 2, sizeof(double));
    double *fs = calloc((size_t)n_rows * 2 * W + 1, sizeof(double)); /* saved forward rows: M[W], rho*It[W] */
    double *Bm = calloc((size_t)W + 2, sizeof(double)), *Bi = calloc((size_t)W + 2, sizeof(double));
    double *zrow = malloc(sizeof(double) * 2 * (size_t)W);
    if (!U || !V || !fs || !Bm || !Bi || !zrow) { free(U); free(V); free(fs); free(Bm); free(Bi); free(zrow); return -2; }
    const double range_lim = pow2i(FDP_RANGE_BITS);

    /* ---- forward ---- */
    int wnext = 0;
    /* row 1: M = e bM, I = EI bI on columns 1 .. min(R, bw+1); D = 0 */
    {
        double *sv = (wnext < n_rows && rows[wnext] == 1) ? fs + (size_t)wnext * 2 * W : NULL;
        for (int j = 0; j < W; ++j) {
            const int k = 1 - bw + j;
            double M = 0.0, It = 0.0;
            if (k >= 1 && k <= R) {
                const double e = ref[k - 1] == qry[0] ? h[H_EMATCH] : h[H_EMIS];
                M = e * h[H_BM];
                It = (EI * h[H_BI]) / gam;
            }
            if (sv) { sv[j] = FDP_STORE(M); sv[W + j] = FDP_STORE(rho * It); }
            U[j] = fma(cU0, M, cU1 * It);
            V[j] = fma(c4, It, M);
        }
        if (sv) wnext++;
    }
    for (int i = 2; i <= L; ++i) {
        /* block rescale + dynamic-range check on the rows entering row i */
        if (((i - 2) % FDP_RESCALE) == 0) {
            double mx = 0.0, mn = INFINITY;
            for (int j = 0; j < W; ++j) {
                if (U[j] > mx) mx = U[j];
                if (V[j] > mx) mx = V[j];
                if (U[j] > 0.0 && U[j] < mn) mn = U[j];
                if (V[j] > 0.0 && V[j] < mn) mn = V[j];
            }
            if (!(mx > 0.0) || !(mx < INFINITY) || mx > mn * range_lim) pflag |= FDP_F_RANGE;
            if (mx > 0.0 && mx < INFINITY) {
                const double sc = pow2i(-expo(mx));
                for (int j = 0; j < W; ++j) { U[j] *= sc; V[j] *= sc; }
            }
        }
        const int save = wnext < n_rows && rows[wnext] == i;
        double *sv = save ? fs + (size_t)wnext * 2 * W : NULL;
        double Dt = 0.0, Mprev = 0.0;
        const uint8_t qy = qry[i - 1];
        for (int j = 0; j < W; ++j) {
            const int k = i - bw + j;
            const int valid = k >= 1 && k <= R;
            const double e = valid ? (ref[k - 1] == qy ? emU : exU) : 0.0;
            const double M = e * U[j];
            const double It = j + 1 < W ? V[j + 1] : 0.0;
            Dt = (j > 0 && valid) ? fma(m8, Dt, Mprev) : 0.0;
            if (sv) { sv[j] = FDP_STORE(M); sv[W + j] = FDP_STORE(rho * It); }
            U[j] = fma(cU0, M, fma(cU1, It, Dt));
            V[j] = fma(c4, It, M);
            if (!valid) { U[j] = 0.0; V[j] = (k < 1) ? V[j] : 0.0; } /* (k < 1: zero by induction anyway) */
            Mprev = M;
        }
        if (save) wnext++;
    }
    /* ---- backward ---- */
    int wprev = n_rows - 1;
    for (int j = 0; j < W; ++j) {
        const int k = L - bw + j;
        const int valid = k >= 1 && k <= Rt;
        Bm[j] = valid ? h[H_SM] / m0 : 0.0;
        Bi[j] = valid ? h[H_SI] / m3 : 0.0;
    }
    const int stop = n_rows > 0 ? rows[0] : L + 1;
    for (int i = L; i >= stop && i >= 1; --i) {
        if (i < L) {
            if (((L - 1 - i) % FDP_RESCALE) == 0) {
                double mx = 0.0, mn = INFINITY;
                for (int j = 0; j < W; ++j) {
                    if (Bm[j] > mx) mx = Bm[j];
                    if (Bi[j] > mx) mx = Bi[j];
                    if (Bm[j] > 0.0 && Bm[j] < mn) mn = Bm[j];
                    if (Bi[j] > 0.0 && Bi[j] < mn) mn = Bi[j];
                }
                if (!(mx > 0.0) || !(mx < INFINITY) || mx > mn * range_lim) pflag |= FDP_F_RANGE;
                if (mx > 0.0 && mx < INFINITY) {
                    const double sc = pow2i(-expo(mx));
                    for (int j = 0; j < W; ++j) { Bm[j] *= sc; Bi[j] *= sc; }
                }
            }
            /* row i from row i+1: slot j of row i <-> column k = i - bw + j; (i+1,k+1) is slot j of row i+1, (i+1,k) slot j-1 */
            const uint8_t qy = qry[i]; /* query base of row i+1 */
            double Dt = 0.0;
            for (int j = W - 1; j >= 0; --j) {
                const int k = i - bw + j;
                const int valid = k >= 1 && k <= R;
                const double e = (k + 1 <= R && k + 1 >= 1) ? (ref[k] == qy ? emB : exB) : 0.0;
                const double X = e * Bm[j];
                const double Y = j > 0 ? Bi[j - 1] : 0.0;
                const double nBm = fma(cB1, Y, fma(cB2, Dt, X));
                const double nBi = fma(c4, Y, X);
                Dt = (i > 1) ? fma(m8, Dt, X) : 0.0;
                Bm[j] = valid ? nBm : 0.0;
                Bi[j] = valid ? nBi : 0.0;
                if (!valid) Dt = 0.0;
            }
        }
        if (wprev >= 0 && rows[wprev] == i) {
            const double *sv = fs + (size_t)wprev * 2 * W;
            const int j0 = bw + 1 - i > 0 ? bw + 1 - i : 0, j1 = (R - i + bw < W - 1) ? R - i + bw : W - 1;
            int n = 0;
            for (int j = j0; j <= j1; ++j) {
                zrow[n++] = sv[j] * FDP_STORE(Bm[j]);
                zrow[n++] = sv[W + j] * FDP_STORE(Bi[j]);
            }
            int st, qq;
            double x;
            const double delta = fdp_delta(L, R, W), A = (2.0 * W + 8.0) * U53;
            int fl = fdp_certify(zrow, n, i - bw + j0, delta, A, thr, &st, &qq, &x);
            state[wprev] = st;
            q[wprev] = (uint8_t)qq;
            flag[wprev] = (uint8_t)fl;
            if (x_out) x_out[wprev] = x;
            if (z_out) {
                double *zo = z_out + (size_t)wprev * 2 * W;
                for (int j = 0; j < W; ++j) { zo[j] = 0; zo[W + j] = 0; }
                for (int j = j0; j <= j1; ++j) { zo[j] = sv[j] * Bm[j]; zo[W + j] = sv[W + j] * Bi[j]; }
            }
            wprev--;
        }
    }
    if (pflag)
        for (int w = 0; w < n_rows; ++w) flag[w] |= (uint8_t)pflag;
    int any = pflag;
    for (int w = 0; w < n_rows; ++w) any |= flag[w];
    free(U); free(V); free(fs); free(Bm); free(Bi); free(zrow);
    return any;
}
