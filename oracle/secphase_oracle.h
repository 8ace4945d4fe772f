/*
 * secphase_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of secphase's per-read-group marker / BAQ /
 * marker-consistency path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, load or call anything in oracle/.
 * The product library (secphase_amd/csrc) never links or calls it.
 *
 * PARITY UNPINNED: the reference cannot be built here (htslib + sonLib are
 * absent) and its tests hold no vector for this path
 * (programs/src/secphase_test.c covers ptBlock only).  Every function below
 * cites the reference lines it restates.
 */
#ifndef SECPHASE_ORACLE_H
#define SECPHASE_ORACLE_H

#include <stdint.h>
#include <stdio.h>

#include "../include/spx_records.h"

#ifdef __cplusplus
extern "C" {
#endif

/* htslib hts.h probaln_par_t */
typedef struct orc_probaln_par {
    float d, e;
    int bw;
} orc_probaln_par;

typedef struct orc_hmm_consts {
    double m[9];
    double bM, bI, sM, sI;
    double e_match, e_mis;
} orc_hmm_consts;

int orc_probaln_glocal(const uint8_t *ref, int l_ref, const uint8_t *query, int l_query,
                       const uint8_t *iqual, const orc_probaln_par *c, int *state, uint8_t *q);
/* test diagnostics: s[0..l_query+1] and z = f*b of the M and I states, row major [l_query][l_ref] (0 outside the
 * band) -- the quantities the product's kernels are compared with bit for bit */
int orc_probaln_posteriors(const uint8_t *ref, int l_ref, const uint8_t *query, int l_query, const uint8_t *iqual,
                           const orc_probaln_par *c, double *s, double *zM, double *zI);
int orc_phred_from_posterior(double max_over_sum);
/* reading of probaln.c's terminal guard: 0 = `u >= bw2*3+3` (default), 1 = `u >= i_dim-3` (probaln_oracle.c, GUARD VARIANTS) */
void orc_set_terminal_guard(int reading);
int orc_get_terminal_guard(void);
void orc_set_scratch_reuse(int on);
/* CPU-baseline runs: per-thread scratch instead of calloc/free per call */
void orc_set_reference_overheads(int on); /* bench bracket: per-group fai_load, per-iterator regcomp / regexec */
void orc_probaln_consts(int l_ref, int l_query, float d, float e, int set_q, orc_hmm_consts *c);

/* glibc rand() (TYPE_3 additive feedback) replay, so that the tie-breaking
 * draws of ptAlignment.c:165-171 are reproducible at "-@1" semantics. */
typedef struct orc_rand {
    int32_t r[34];
    int f, b;
    uint32_t tbl[31];
} orc_rand;
void orc_srand(orc_rand *st, unsigned seed);
int orc_rand_next(orc_rand *st);

/* One CIGAR-iterator state (cigar_it.h:12-41); ret = value ptCigarIt_next
 * returned when it produced this state. */
typedef struct orc_op {
    int op, len, ret;
    int sqs, sqe, rfs, rfe, rds_f, rde_f;
} orc_op;

typedef struct orc_marker { /* ptMarker.h:34-41 */
    int32_t alignment_idx;
    int32_t read_pos_f;
    int32_t base_idx;
    int32_t base_q;
    int32_t is_match;
    int32_t ref_pos;
} orc_marker;

typedef struct orc_block { /* ptBlock.h:23-38 without the payload */
    int rfs, rfe, sqs, sqe, rds_f, rde_f;
} orc_block;

/* trace of every probaln_glocal call of a group (for kernel-level parity) */
typedef struct orc_baq_call {
    int aln, block;
    int sqs, sqe, rfs, rfe, bw;
} orc_baq_call;

typedef struct orc_group_result {
    int n_aln;
    int best_idx;      /* return of get_best_record_index */
    int prim_idx;
    int relabel;       /* best is a secondary -> a record is written */
    int n_rand;        /* rand() draws consumed (1 or 2) */
    double score[16];
    int rfe[16];       /* ptAlignment.rfe, printed in out.log */
    int n_markers_initial, n_markers_final, n_blocks;
    int n_baq_calls;
    long long dp_cells;
    int rfs[16];       /* ptAlignment.rfs */
    orc_marker *final_markers; /* markers after filter_lowq_markers (only kept on request; owned) */
    int n_final;
} orc_group_result;

/* Run the marker branch of runOneThread (secphase.c:156-219) for group g of
 * the batch.  qual bytes are copied, the batch is not modified.  If `log` is
 * non-NULL the out.log record is appended (secphase.c:194-200,32-57).
 * `calls`/`max_calls` optionally receive the trace of BAQ calls.
 * Returns 0, or <0 when the group uses a construct the reference leaves
 * undefined. */
int orc_score_group(const spx_batch *bt, const spx_ref *ref, int g, const spx_params *par,
                    orc_rand *rng, orc_group_result *out, FILE *log,
                    orc_baq_call *calls, int max_calls);

/* dispatch filter of parseAlignmentsAndScatterJobs (secphase.c:285-288) */
int orc_group_is_dispatched(const spx_batch *bt, int g);

/* Whole-batch driver used by tests and by the cpu_baseline leg: runs groups
 * in file order on `threads` worker threads, then emits records in file order
 * with the rand() replay done in file order (= the reference at -@1).
 * results[] must hold bt->n_groups entries.  Returns number of relabelled
 * groups. */
int orc_run_batch(const spx_batch *bt, const spx_ref *ref, const spx_params *par, int threads,
                  unsigned rand_seed, orc_group_result *results, const char *log_path);

/* orc_run_batch + the two marker-mode BED side outputs (src/secphase.c:201-212,719-732): blocks of the old
 * primary and of the promoted secondary of every relabelled read, merged with counts; reference positions of
 * their surviving markers, merged without counts.  Paths may be NULL. */
int orc_run_batch_bed(const spx_batch *bt, const spx_ref *ref, const spx_params *par, int threads, unsigned rand_seed,
                      orc_group_result *results, const char *log_path, const char *bed_modified_path,
                      const char *bed_marker_path);

/* orc_run_batch that also returns the record qualities as calc_local_baq leaves them (ptMarker.c:706,759,763),
 * i.e. what -w/--writeBam hands to sam_write1 (src/secphase.c:182-189).  qual_out is laid out like bt->qual and
 * must be pre-filled with a copy of it; records of groups that are not dispatched keep their bytes. */
int orc_run_batch_quals(const spx_batch *bt, const spx_ref *ref, const spx_params *par, int threads,
                        orc_group_result *results, uint8_t *qual_out);

/* ptBlock sort / merge (blocks_oracle.c) */
void orc_blocks_sort(int n, int *s, int *e, int *c);
int orc_blocks_merge(int n, const int *s, const int *e, const int *c, int *os, int *oe, int *oc);
int orc_blocks_merge_v2(int n, const int *s, const int *e, const int *c, int *os, int *oe, int *oc);

/* helpers exposed for unit tests */
int orc_walk_cigar(const spx_batch *bt, int a, orc_op **ops_out); /* returns n states incl. state 0 */
extern const unsigned char orc_nt16_table[256];
extern const unsigned char orc_nt16_int[16];

#ifdef __cplusplus
}
#endif
#endif
