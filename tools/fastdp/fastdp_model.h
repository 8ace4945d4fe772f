/* fastdp_model.h -- DESIGN STUDY / TEST INFRASTRUCTURE: CPU model of the fast DP tier and its certificate (fastdp_model.c). */
#ifndef FASTDP_MODEL_H
#define FASTDP_MODEL_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rows between two power-of-two rescales / dynamic-range checks */
#ifndef FDP_RESCALE
#define FDP_RESCALE 16
#endif
/* a row whose largest and smallest non-zero value are more than 2^FDP_RANGE_BITS apart flags the problem: beyond it the
 * exact tier (rows normalised to sum 1) may leave the normal FP64 range before the next check, where its roundings are no
 * longer relative */
#ifndef FDP_RANGE_BITS
#define FDP_RANGE_BITS 600
#endif
/* a problem whose smallest per-row factor mu = min(m0 e_mis, EI m4) is below 2^-FDP_MU_BITS is outside the model: between two checks a
 * row's spread can grow by 1/mu per row, and FDP_RANGE_BITS + FDP_RESCALE * FDP_MU_BITS + 100 (constant factors between the carried
 * rows and the exact tier's M, I, D) must stay below 1022 */
#define FDP_MU_BITS ((1000 - 100 - FDP_RANGE_BITS) / FDP_RESCALE)
/* delta = FDP_DELTA_PER_STEP * (L + R + W + 16) * 2^-53: bound on the relative deviation between a row-normalised posterior
 * product of the fast tier and of the exact tier.  Every value of either tier is a sum of products of non-negative numbers,
 * so its relative error is at most (roundings along the deepest lattice path) * 2^-53; a path has <= L row steps and <= R
 * column steps, a row step costs <= 6 roundings in either tier, a column step <= 2, constants <= 4, z = f*b doubles it and
 * two tiers add up: 2 * 2 * (6 L + 2 R + W + 8) <= 24 (L + R + W).  48 leaves a factor 2. */
#ifndef FDP_DELTA_PER_STEP
#define FDP_DELTA_PER_STEP 48.0
#endif

/* FP32 storage of the saved rows (fastdp_model.c FDP_STORE): studied in round 6 and NOT adopted -- it halves the scratch and the MAP kernel's
 * traffic, still with 0 uncertified mismatches, but the wider delta flags 0.04 % (HiFi) / 0.6 % (ONT, mixed) of the problems through near-ties
 * of the two largest products, and a re-run problem costs its whole wave in the exact kernels (ONT: ~9 % of the waves).  -DFDP_STORE_FLOAT=1. */
#ifndef FDP_STORE_FLOAT
#define FDP_STORE_FLOAT 0
#endif
#define FDP_DELTA_STORE (FDP_STORE_FLOAT ? 2.384185791015625e-07 /* 2^-22 */ : 0.0)

/* why a row / problem is not certified */
#define FDP_F_ARGMAX 1  /* two largest z closer than delta */
#define FDP_F_THRESH 2  /* 1 - max/sum within its error interval of a phred threshold */
#define FDP_F_XSMALL 4  /* ... of zero (the exact tier's 1 - fl(max/sum) may be 0: q = 0) */
#define FDP_F_RANGE 8   /* dynamic range of a row / non-finite or non-positive values */
#define FDP_F_MODEL 16  /* outside the fast tier's model (ambiguous bases, degenerate constants) */

int fdp_phred(double x, const double *thr);
double fdp_delta(int L, int R, int W);
/* z[0..n): posterior products of one row in column order (M, I per column), first column k0 (1-based).  A = absolute error bound of
 * the exact tier's 1 - fl(max / fl(sum)).  Returns the flag bits; *state / *q are the fast tier's answer. */
int fdp_certify(const double *z, int n, int k0, double delta, double A, const double *thr, int *state, int *q, double *x_out);
/* one problem: ref[R], qry[L] codes 0..3 (4 = ambiguous: flagged FDP_F_MODEL), h = 16 HMM constants in spx_device.h's SPX_H_* order,
 * bw = effective half band width, rows[n_rows] ascending 1-based wanted rows.  Outputs per wanted row; z_out (may be NULL):
 * [n_rows][2][2*bw+1] products by band slot; x_out (may be NULL): 1 - max/sum.  Returns the OR of all flags (< 0: error). */
int fdp_glocal(const uint8_t *ref, int R, const uint8_t *qry, int L, const double *h, int bw, int n_rows, const int *rows,
               const double *thr, int *state, uint8_t *q, uint8_t *flag, double *z_out, double *x_out);

#ifdef __cplusplus
}
#endif
#endif
