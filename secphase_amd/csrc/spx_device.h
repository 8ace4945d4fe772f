/*
 * spx_device.h -- device-side batch layout shared by the host runtime
 * (spx_runtime.cpp) and the HIP kernels (spx_kernels.hip).  Internal.
 */
#ifndef SPX_DEVICE_H
#define SPX_DEVICE_H

#include <stdint.h>

/* ref/query base codes on the device: 0..3 = ACGT, 4 = N/ambiguous,
 * SPX_CODE_OUT = column outside [1,R] (no such DP cell) */
/* band classes = kernel instantiations (lanes per problem x slots per lane), see spx_launch_baq */
#define SPX_N_CLASSES 14
#define SPX_MAX_STAGE_SLOTS (1 << 20) /* alignments (and groups) per work list: the tiled device scans cover 1024 tiles of 1024 */

#define SPX_CODE_N 4
#define SPX_CODE_OUT 8

/* per-problem HMM constants, all computed on the host exactly as
 * probaln_glocal's initialisation does (float/double mix), 16 doubles */
enum {
    SPX_H_M0 = 0, /* M->M */
    SPX_H_M1,     /* M->I */
    SPX_H_M2,     /* M->D */
    SPX_H_M3,     /* I->M */
    SPX_H_M4,     /* I->I */
    SPX_H_M6,     /* D->M */
    SPX_H_M8,     /* D->D */
    SPX_H_BM,
    SPX_H_BI,
    SPX_H_SM,
    SPX_H_SI,
    SPX_H_EMATCH,
    SPX_H_EMIS,
    SPX_H_PAD0,   /* != 0: the window or the query holds an ambiguous base */
    SPX_H_TDROP,  /* != 0: the termination / backward start leave out column l_ref (terminal-guard reading "row", spx_logic.h terminal_drop) */
    SPX_H_PAD2,
    SPX_H_N
};

/* One batch of DP problems resident in HBM (structure of arrays). */
typedef struct spx_dev_batch {
    /* launch list of one band class: problem ids ordered by (W, L desc), padded
     * with -1 so that every wave holds problems of one W */
    const int32_t *order;     /* forward kernels: by (W, L descending) */
    const int32_t *order_bwd; /* backward kernels: by (W, rows actually walked = L - first wanted row, descending) */
    int32_t n_order;
    int32_t n_order_bwd;
    /* per problem */
    const int64_t *ref_nib; /* nibble index of ref window start in ref4   */
    const int64_t *qry_nib; /* nibble index of query window start in qry4 */
    const int32_t *L;       /* query length  */
    const int32_t *R;       /* ref length    */
    const int32_t *bw;      /* effective half band width */
    const double *hmm;      /* [n][SPX_H_N] */
    const int32_t *row_off; /* first wanted row of the problem */
    const int32_t *n_rows;
    const int64_t *s_off;   /* offset into sinv[] (L+2 doubles per problem) */
    /* pools */
    const uint8_t *ref4; /* reference, 4-bit codes, low nibble first */
    const uint8_t *qry4; /* query windows, same packing */
    /* wanted rows (ascending per problem) */
    const int32_t *rows;       /* 1-based query row */
    const int32_t *row_expect; /* ref index (0-based, window-relative) the CIGAR puts this base on */
    const uint8_t *row_rawq;   /* raw base quality */
    /* scratch */
    double *sinv;  /* 1/s[i] per row */
    double *s_raw; /* diagnostics (spx_probaln_glocal's return value): s[i] itself for 1 <= i < L, same offsets as sinv; NULL on the scoring path */
    double *fsave; /* wanted rows: scaled forward M,I ([row][2][slots]), replaced by f*b in the backward pass */
    const int32_t *row_prob;   /* per wanted row: its problem */
    const int32_t *prob_slots; /* per problem: band slots of its class */
    int64_t fsave_stride; /* doubles per wanted row = 2*slots of the class */
    const int64_t *fsave_off; /* per problem: offset (in doubles) of its first saved row */
    /* outputs per wanted row */
    uint8_t *out_bq;    /* min(raw, q) with the CIGAR/MAP consistency check, capped at 93 */
    int32_t *out_state; /* may be NULL */
    uint8_t *out_q;     /* may be NULL */
    const double *qthr; /* [102] phred thresholds on 1 - max/sum */
    /* DP slices (round 4): a launch may cover the wanted rows [row_base, row_base + n) of the list only (MAP kernel); the
     * forward / backward kernels of a slice get that slice's launch orders, and sinv / fsave point at scratch that is
     * shared by all slices, shifted by the slice's first offset (s_off / fsave_off stay absolute) */
    int32_t row_base;
    /* Two-tier DP (round 6).  tier[p]: SPX_TIER_EXACT = exact tier only (band class without a fast kernel; the
     * value the list's 0xff fill leaves), SPX_TIER_FAST = computed by the fast tier (written by the fast forward kernel at every launch),
     * SPX_TIER_RERUN = not certified / outside the fast tier's model: the exact kernels re-run the problem.  NULL: no tiers (every problem
     * takes the exact kernels).  The exact kernels and MAP kernel process a problem only if tier[p] == tier_want (SPX_TIER_ALL: every
     * problem). */
    int32_t tier_want;
    int32_t *tier;
    int32_t *tier_counts; /* diagnostics, each starting at -1 (the same fill): [0] problems flagged by the certificate, [1] by the model
                           * conditions at entry, [2] by the dynamic-range check, [3] rows not certified */
} spx_dev_batch;
#define SPX_TIER_ALL (-2)
#define SPX_TIER_EXACT (-1)
#define SPX_TIER_FAST 1
#define SPX_TIER_RERUN 2

/* Constants of the fast tier: the same for every problem of a launch.  The exact tier's transition constants carry a factor
 * (1 - sM), sM = 1/(2 l_query + 2), on every transition OUT of an M or I state (probaln_glocal's m[0..4]); a path that has reached row i
 * has left exactly i - 1 such states whatever its shape, so that factor is common to the whole row and cancels in the row-normalised
 * posteriors: the fast tier drops it, and its constants depend on (d, e, set_q) only -- they live in SGPRs. */
#define SPX_FAST_MAXC 32
typedef struct spx_fast_consts {
    double pw[SPX_FAST_MAXC + 1]; /* pw[c] = m8^c */
    double m8;                    /* D->D = (double)e */
    double m0h, m1h, m3h, m4h, m6; /* (double)((1-d)-d), (double)d, (double)(1-e), (double)e, (double)(1-e): m[] without (1 - sM) */
    double e_match, e_mis;
    /* derived (spx_runtime.cpp fast_constants) */
    double emU, exU, cU0, cU1, c4; /* forward:  M = (e m6 m2) U',  U = cU0 M + cU1 It + Dt,  V = M + c4 It */
    double emB, exB, cB1, cB2;     /* backward: X = (e m0) Bm',  Bm = X + cB1 Y + cB2 Dt,  Bi = X + c4 Y */
    double rho;                    /* z_I / z_M correction = EI m1 m3 / m0 */
    double ups, gam;               /* m6 m2, EI m1 */
    int32_t range_bits; /* a row spanning more than 2^range_bits flags its problem */
    int32_t mu_bits;    /* problems whose smallest per-row factor is below 2^-mu_bits are outside the model */
} spx_fast_consts;

/* marker table for the scoring kernel: one entry per (position, alignment) */
typedef struct spx_dev_marker {
    int32_t row;     /* index into out_bq, or -1: use qfix */
    uint8_t qfix;    /* quality when row < 0 (raw, or 0 at block edges) */
    uint8_t is_match;
    uint8_t aln;
    uint8_t first_of_pos; /* on the first marker of a read position: how many markers it has (= alignments), else 0 */
} spx_dev_marker;

typedef struct spx_dev_groups {
    int32_t n_groups;
    const int32_t *mk_first; /* [n_groups+1] */
    const spx_dev_marker *markers;
    const uint8_t *n_aln;     /* [n_groups] */
    const uint16_t *sec_mask; /* bit a set: alignment a is secondary */
    const uint8_t *out_bq;
    const double *match_tbl; /* [256] -1*reverse_quality(q)    */
    const double *mis_tbl;   /* [256] -1*q - 10*log(3)         */
    int32_t min_q;
    double prim_margin, min_score;
    /* outputs */
    double *score;      /* [n_groups][10] */
    uint8_t *prim_idx;  /* [n_groups] */
    uint8_t *max_idx;   /* first secondary with the greatest score */
    uint16_t *tie_mask; /* secondaries with score >= max */
    uint8_t *pass;      /* max > prim+margin && max >= min_score */
} spx_dev_groups;

#endif
