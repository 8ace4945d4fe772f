/*
 * spx.h -- C-ABI of the MI355X-native secphase scoring path (libspx.so).
 *
 * secphase has no plugin/FFI layer; the seams this library replaces are
 * (all paths relative to /root/reference/programs):
 *
 *   spx_score_batch / spx_prepare+spx_launch+spx_collect
 *       = the marker branch of runOneThread for every dispatched group,
 *         src/secphase.c:156-192 (markers, consensus blocks, BAQ, marker
 *         filter, score, get_best_record_index), i.e. what
 *         tpool_add_work(tm, runOneThread, arg) at src/secphase.c:303 runs.
 *   spx_group_is_dispatched
 *       = the dispatch filter, src/secphase.c:285-288.
 *   spx_finalize
 *       = the rand() dependent tail of get_best_record_index,
 *         submodules/ptAlignment/ptAlignment.c:163-176, replayed in file order
 *         (= the reference at -@1).
 *   spx_write_relabel_log
 *       = the record writer, src/secphase.c:194-200 + print_alignment_scores
 *         src/secphase.c:32-57 (the list correct_bam.c:32-88 parses).
 *   spx_probaln_glocal
 *       = htslib-1.17 probaln_glocal as called at
 *         submodules/ptMarker/ptMarker.c:755-757 (same contract: caller owns
 *         state[l_query] and q[l_query]; INT_MIN on failure).
 *
 * Plain pointers and sizes only; no torch / HIP types.  Every entry point
 * returns 0 on success or a negative SPX_E* code; nothing calls exit().
 * There is NO CPU fallback: without a usable gfx950 device spx_create fails
 * with SPX_ENODEVICE.
 */
#ifndef SPX_H
#define SPX_H

#include <stdint.h>

#include "spx_records.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SPX_OK 0
#define SPX_ENODEVICE (-1) /* no HIP device / wrong architecture                     */
#define SPX_EHIP (-2)      /* a HIP runtime call failed (see spx_last_error)          */
#define SPX_EINVAL (-3)    /* malformed argument                                      */
#define SPX_ENOMEM (-4)
#define SPX_EUNSUPPORTED (-5) /* input uses a construct the reference leaves undefined  */
#define SPX_ENOREF (-6)    /* spx_set_reference has not been called                   */
#define SPX_ENOTAG (-7)    /* a record of a dispatched group has neither cs nor MD: the reference prints "At least
                            * one of the MD or CS tags should be present!" and exits (cigar_it.c:64-67)   */

typedef struct spx_ctx spx_ctx;
typedef struct spx_work spx_work;

/* htslib hts.h probaln_par_t */
typedef struct spx_probaln_par {
    float d, e;
    int bw;
} spx_probaln_par;

/* per-group result (fixed size so that ranks can gather it with one collective) */
typedef struct spx_group_out {
    double score[10];   /* marker-consistency score of each alignment (ptAlignment.score) */
    int32_t rfe[10];    /* ptAlignment.rfe, printed in the relabel list                     */
    int8_t n_aln;       /* alignments scored (0: group not dispatched, <0: SPX_E* for the group) */
    int8_t prim_idx;    /* the non-secondary record                                         */
    int8_t max_idx;     /* first secondary with the greatest score                          */
    int8_t pass;        /* max > prim + prim_margin && max >= min_score                     */
    uint16_t tie_mask;  /* secondaries whose score >= max                                   */
    int8_t best_idx;    /* filled by spx_finalize: return of get_best_record_index          */
    int8_t relabel;     /* filled by spx_finalize: best is a secondary                      */
    int32_t n_problems; /* banded DP problems this group produced                           */
    int32_t n_markers;  /* markers entering the filter                                      */
    int64_t dp_cells;   /* band cells of those problems                                     */
} spx_group_out;

typedef struct spx_stats {
    int64_t n_groups, n_dispatched, n_problems, n_rows, dp_cells;
    int64_t n_markers;
    int64_t bytes_h2d, bytes_d2h;
    int64_t problems_per_class[16];
    double prep_seconds, h2d_seconds, kernel_seconds, d2h_seconds;
    /* dominant kernel, measured with HIP events on the launch stream, averaged (see n_launches_averaged) */
    double baq_kernel_ms;   /* all BAQ launches of the work list (forward, backward, MAP; every band class): start -> last backward kernel on the
                             * main stream + the span of the last MAP kernel on the result stream (where it runs beside the next list's DP kernels) */
    double score_kernel_ms;
    /* the band class holding most cells: its forward / backward kernel alone */
    double main_fwd_ms, main_bwd_ms;
    int64_t main_class_cells;
    int32_t main_class, main_class_lanes, main_class_slots;
    int32_t n_launches_averaged; /* the *_ms fields are averages over this many launches: those since the previous
                                  * spx_collect on the context (at most the last 64) */
    int32_t dp_slices;           /* DP slices of the list (spx_work_device_bytes): main_fwd_ms / main_bwd_ms are the SUM over the slices'
                                  * kernels of one launch of the list, i.e. dp_slices launches of the kernel each */
    int32_t reserved_;
    /* two-tier DP (round 6): problems of band classes that have a fast tier, and how many of them the exact kernels re-ran because a
     * wanted row was not certified / the problem is outside the fast tier's model (ambiguous base, degenerate constants) / a row's values
     * spanned too many binades; rows not certified.  Per launch, averaged like the *_ms fields.  All 0 when the tiers are off. */
    int64_t tier_fast_problems, tier_rerun_certificate, tier_rerun_model, tier_rerun_range, tier_rows_uncertified;
    /* round 6: the two spans baq_kernel_ms adds up, separately.  dp_critical_ms: main stream, first DP kernel of the list -> behind its last backward
     * kernel (and, with the two-tier DP, behind the MAP kernels of every slice): what the list keeps the main stream for.  tail_span_ms: what follows on
     * the result stream BESIDE the next list's DP kernels -- last MAP kernel (tiers off), re-runs of uncertified problems, marker filter / score /
     * decision / result kernels -- from its first kernel's start to its last kernel's end (waiting for the chip included). */
    double dp_critical_ms, tail_span_ms;
} spx_stats;

const char *spx_strerror(int code);
const char *spx_last_error(void);
int spx_device_count(void);

int spx_create(int device, spx_ctx **out);
void spx_destroy(spx_ctx *ctx);

/* copy the assembly into HBM as 4-bit codes (resident for the ctx lifetime) */
int spx_set_reference(spx_ctx *ctx, const spx_ref *ref);

int spx_group_is_dispatched(const spx_batch *bt, int32_t g);

/* one call: prepare + launch + collect */
int spx_score_batch(spx_ctx *ctx, const spx_batch *bt, const spx_params *par, spx_group_out *out, spx_stats *stats);

/* split phase (what bench.py times; also lets a caller overlap host parsing of the next batch) */
int spx_prepare(spx_ctx *ctx, const spx_batch *bt, const spx_params *par, int host_threads, spx_work **work);
/* same, over several record batches that become ONE work list; group g of batch b is reported at
 * out[(groups of batches < b) + g] */
int spx_prepare_many(spx_ctx *ctx, const spx_batch *const *batches, int32_t n_batches, const spx_params *par,
                     int host_threads, spx_work **work);
/* The two halves of spx_prepare_many.  spx_stage: dispatch filter (src/secphase.c:285-288) + the dispatched groups'
 * records packed into pinned memory on host_threads threads + ONE asynchronous copy into HBM.  spx_prepare_staged:
 * the work list itself -- CIGAR/cs walk, markers, consensus windows, banded DP problems and marker table
 * (cigar_it.c, ptMarker.c:42-831, src/secphase.c:162-170) -- is built ON THE DEVICE from the staged records.  It may
 * be called again on the same staged records (records resident in HBM; bench.py times exactly that), and is
 * thread-safe: preparations of different work lists are serialised on the context's preparation stream, which runs
 * beside the DP kernels of the list launched before. */
int spx_stage(spx_ctx *ctx, const spx_batch *const *batches, int32_t n_batches, const spx_params *par, int host_threads,
              spx_work **work);
int spx_prepare_staged(spx_ctx *ctx, spx_work *work);
/* CPUs this process can really use: online CPUs cut by the affinity mask and by the container's CPU-time quota (cgroup
 * cpu.max).  The default wherever a host_threads argument is <= 0. */
int spx_effective_cpus(void);
/* Diagnostics, host only: what spx_stage would put on the wire for these batches.  A secondary whose SEQ / QUAL merely
 * repeat the primary's (same strand: the same bytes; other strand: reverse complement / reversed qualities; each minus
 * the record's hard clips -- verified base by base on the staging threads) is not transferred: the device rebuilds it.
 * out[0] alignments of dispatched groups, out[1] aliased ones, out[2] SEQ + QUAL bytes of all, out[3] bytes transferred. */
int spx_stage_transfer_stats(const spx_batch *const *batches, int32_t n_batches, int host_threads, int64_t *out);
/* drops the prepared list (its HBM goes back to the context's cache), keeps the staged records */
int spx_work_release(spx_ctx *ctx, spx_work *work);
int spx_launch(spx_ctx *ctx, spx_work *work);  /* asynchronous on the ctx stream; inputs already in HBM */
int spx_sync(spx_ctx *ctx);
/* Returns the device and pinned memory the context keeps for re-use (arenas of freed work lists, preparation pools)
 * to the driver -- before another process needs the GPU, or after an unusually large batch. */
int spx_trim(spx_ctx *ctx);
/* ---- multi-GPU: what crosses ranks (spx_gather.cpp).  Read groups shard over ranks; the relabel list is a property of
 * the whole file -- records in file order, the tie-breaking rand() stream consumed in file order (src/secphase.c:194-217,
 * ptAlignment.c:163-176 at -@1).  Every rank contributes (a) one spx_decision per dispatched group (ONE gather of
 * fixed-size records: enough to replay the draws of every group in global order) and (b) one spx_relabel_rec per
 * candidate group (those whose best alignment can be a secondary: what print_alignment_scores, src/secphase.c:32-57,
 * needs to write the record). ---- */
typedef struct spx_decision { /* 16 bytes */
    uint32_t group;    /* global group index */
    int8_t n_aln;      /* alignments scored; < 2: the group draws nothing */
    int8_t prim_idx, max_idx;
    uint8_t pass;      /* max > prim + prim_margin && max >= min_score */
    uint16_t tie_mask; /* secondaries whose score >= max */
    uint16_t reserved;
    int32_t absdiff;   /* abs((int)(max_score - prim_score)), the operand of the prim_margin_random test (ptAlignment.c:172) */
} spx_decision;
typedef struct spx_relabel_rec {
    uint32_t group;
    int8_t n_aln, prim_idx;
    int8_t pad_[2];
    double score[10];
    int32_t rfe[10], pos[10], tid[10];
    uint16_t flag[10];
    char qname[260];
} spx_relabel_rec;
/* One spx_decision per dispatched group of `work` into a caller-owned DEVICE buffer (e.g. a torch tensor that is then
 * handed to an RCCL gather); group = index in the work list's input + group_base.  Waits for the work list's kernels
 * and returns when the records are written.  Returns the number of records or SPX_E*. */
int spx_pack_decisions(spx_ctx *ctx, spx_work *work, int32_t group_base, void *device_out, int64_t capacity);
/* the same records from collected results (host side; groups that are not scored are left out) */
int spx_decisions_from_results(const spx_group_out *out, int32_t n_groups, int32_t group_base, spx_decision *dst, int32_t capacity);
/* candidate records of a collected batch (dst == NULL: only counts them) */
int spx_relabel_candidates(const spx_batch *bt, int32_t group_base, const spx_group_out *out, const spx_params *par,
                           spx_relabel_rec *dst, int32_t capacity);
/* rank 0: the draws of all groups, records sorted by group (a finalizer keeps ONE stream over the whole run) */
typedef struct spx_finalizer spx_finalizer;
int spx_finalizer_apply_decisions(spx_finalizer *f, const spx_params *par, const spx_decision *dec, int32_t n, int8_t *best_idx,
                                  int8_t *relabel);
/* appends the relabel records of the candidates whose decision (best_idx[k], from the call above) is a secondary;
 * returns how many were written */
int spx_write_relabel_records(const char *path, const char *mode, const spx_ref *ref, const spx_relabel_rec *recs, int32_t n,
                              const int8_t *best_idx);
/* Decisions made where the groups are (round 3): the rand() values a rank's groups consume are known locally, so ranks
 * exchange one count each, pass over the others' draws and finalize + format their own groups; rank 0 appends the text
 * fragments in rank order -- the same list, without a serial replay of every group on one rank. */
int64_t spx_count_draws(const spx_group_out *out, int32_t n_groups);
int spx_finalizer_skip(spx_finalizer *f, int64_t n_draws);
/* what spx_write_relabel_log would append for a finalized batch, into malloc'ed memory (spx_free_text) */
int spx_format_relabel_text(const spx_batch *bt, const spx_ref *ref, const spx_group_out *out, char **text, int64_t *len);
void spx_free_text(char *text);
int spx_collect(spx_ctx *ctx, spx_work *work, spx_group_out *out);
/* Quality arrays as the reference leaves them in the records after calc_local_baq (ptMarker.c:706,759,763),
 * for a work list prepared with params.flags & SPX_PAR_ALL_ROWS and already launched.  `qual` must hold a copy
 * of bt->qual (same qual_off layout) for batch number batch_index of spx_prepare_many; it is edited in place. */
int spx_apply_quals(spx_ctx *ctx, spx_work *w, int32_t batch_index, const spx_batch *bt, uint8_t *qual);
int spx_work_stats(const spx_work *work, spx_stats *stats);
/* diagnostics: device bytes of the prepared list (work list + scratch + outputs; the staged records not counted) and the
 * number of DP slices it runs in.  Slices (round 4): 1/s of every DP row and the saved forward rows -- the bulk of a list --
 * exist for ONE range of consecutive groups at a time: forward -> backward -> MAP run range by range over the same scratch
 * (ptMarker.c:699-806 has no such notion: it realigns one window at a time).  SPX_DP_SLICE_GB (default 16) is the scratch a
 * slice may take, SPX_DP_SLICES forces a count. */
int64_t spx_work_device_bytes(const spx_work *work, int32_t *n_slices);
void spx_work_free(spx_ctx *ctx, spx_work *work);

/* ---- in-order pipeline over one context: what replaces the reference's thread pool + output mutex
 * (src/secphase.c:230-351, :74-228).  `depth` submissions are in flight at once, each on its own worker thread:
 * staging (host threads) -> copy to HBM -> device preparation -> DP + scoring kernels -> packed results back; the copy
 * of one batch, the preparation of the next and the kernels of a third overlap on the device.  Results are handed out
 * in SUBMISSION order (the rand() replay and the relabel list need file order). ---- */
typedef struct spx_pipe spx_pipe;
int spx_pipe_create(spx_ctx *ctx, const spx_params *par, int depth, int host_threads, spx_pipe **out);
/* Either record batches (valid until the matching spx_pipe_next has returned), or -- staged != NULL -- a work list the
 * caller staged with spx_stage (records resident in HBM; it is prepared again, launched and collected; it stays the
 * caller's) with the number of input groups it covers.  Blocks while depth + 1 submissions are in flight. */
int spx_pipe_submit(spx_pipe *p, const spx_batch *const *batches, int32_t n_batches, spx_work *staged, int32_t n_groups_staged,
                    void *tag);
/* Results of the OLDEST submission: returns its number of input groups (out[0..n) filled like spx_collect does) or
 * SPX_E*.  *work (may be NULL) receives the work list -- for spx_relabel_blocks / spx_apply_quals; the caller frees it
 * with spx_work_free unless it is its own staged list; with work == NULL a list the pipe created is freed here. */
int spx_pipe_next(spx_pipe *p, spx_group_out *out, int32_t capacity, spx_work **work, void **tag);
int spx_pipe_pending(spx_pipe *p);
void spx_pipe_destroy(spx_pipe *p);

/* rand() replay + decision, in file order */
int spx_finalize(const spx_params *par, unsigned rand_seed, spx_group_out *out, int32_t n_groups);

/* the same with ONE draw stream kept across batches (a whole run = the reference at -@1) */
int spx_finalizer_create(unsigned rand_seed, spx_finalizer **out);
int spx_finalizer_apply(spx_finalizer *f, const spx_params *par, spx_group_out *out, int32_t n_groups);
void spx_finalizer_free(spx_finalizer *f);

/* append the relabel records of a finalized batch to `path` (mode "w" or "a") */
int spx_write_relabel_log(const char *path, const char *mode, const spx_batch *bt, const spx_ref *ref,
                          const spx_group_out *out);

/* single banded-HMM problem on the device (unit tests / drop-in for the htslib symbol).
 * Uses a process-wide context on device 0 created on first use.
 * iqual == NULL: Q30 for every base, as htslib.  One value for all bases -- the only way secphase calls it
 * (submodules/ptMarker/ptMarker.c:747-749 fills the array with set_q before the call at :755) -- runs on the scoring kernels,
 * which keep the two emission values of that quality in registers.  Base-by-base qualities (samtools' BAQ use of probaln_glocal)
 * take a general kernel (spx_probaln_general.hip: one lane per problem, rows in HBM, the reference's own operation order): same
 * results bit for bit, at the rate of a contract, not of the hot path. */
int spx_probaln_glocal(const uint8_t *ref, int l_ref, const uint8_t *query, int l_query, const uint8_t *iqual,
                       const spx_probaln_par *c, int *state, uint8_t *q);

/* batched form of the above on an explicit ctx: n problems, all rows wanted.
 * ref/query are concatenated 0..4 codes; *_off have n+1 entries; outputs are
 * concatenated like query. set_q is the constant base quality of every row. */
int spx_probaln_batch(spx_ctx *ctx, int32_t n, const uint8_t *ref, const int64_t *ref_off, const uint8_t *query,
                      const int64_t *qry_off, const int32_t *set_q, const spx_probaln_par *pars, int32_t *state,
                      uint8_t *q, double *kernel_ms);
/* Diagnostics for parity tests: runs the same batch as spx_probaln_batch and returns, for problem `which`, the
 * kernels' own intermediate results, so that a test can compare them with a CPU implementation bit for bit
 * instead of through the quantised state[]/q[] only.  With L = its query length, R = its reference length:
 * scale[0] = 1, scale[i] = 1/s[i] for 1 <= i < L, scale[L] = s[L], scale[L+1] = s[L+1] (s = kprobaln's per-row
 * scaling factors); zM, zI [L][R] row major: f*b of the M and I states of cell (i,k), 0 outside the band -- the
 * products probaln_glocal's MAP loop maximises and sums. */
int spx_probaln_posteriors(spx_ctx *ctx, int32_t n, const uint8_t *ref, const int64_t *ref_off, const uint8_t *query,
                           const int64_t *qry_off, const int32_t *set_q, const spx_probaln_par *pars, int32_t which,
                           double *scale, double *zM, double *zI);

/* PARITY-UNPINNED switch.  htslib's probaln.c guards the termination sum s[l_query+1] and the backward start with
 * `if (u < 3 || u >= LIMIT) continue;`.  No htslib exists on the build machine, and two recollections of LIMIT in release 1.17 disagree:
 *   SPX_GUARD_BAND (default): bw2*3+3 (kprobaln's band test);  SPX_GUARD_ROW: i_dim-3 (the shrunken row's length).
 * They differ for one cell only -- column l_ref of row l_query when l_query <= bw && 2*bw+1 > l_ref (reached by `--ont -b 50` on blocks of
 * <= 50 bases) -- which the ROW reading leaves out.  The setting is process-wide, is read when a work list is prepared (work lists already
 * prepared keep theirs) and is honoured by every DP kernel and by spx_probaln_glocal; the environment variable SPX_TERMINAL_GUARD=band|row
 * sets the initial value.  Call sites replaced: /root/reference/programs/submodules/ptMarker/ptMarker.c:754-757. */
#define SPX_GUARD_BAND 0
#define SPX_GUARD_ROW 1
int spx_set_terminal_guard(int reading);
int spx_get_terminal_guard(void);

/* Two-tier DP (round 6): a fast evaluation of the same HMM (fused multiply-adds, no row sums) whose (state, q) are
 * CERTIFIED equal to the exact tier's per wanted row; problems with a row that is not certified are re-run by the exact (bit-exact)
 * kernels on the device.  Process-wide, read when a work list is prepared; SPX_DP_TIERS=0|1 sets the initial value (default 1).  The
 * diagnostics entry points (spx_probaln_posteriors, spx_probaln_glocal's likelihood) always take the exact kernels. */
int spx_set_dp_tiers(int on); /* 0 off, 1 on, 2 test mode: the fast tier runs but certifies nothing, so that every problem is re-run */
int spx_get_dp_tiers(void);
/* diagnostics of the latest spx_collect / spx_probaln_batch of the process: problems of band classes with a fast tier; of those,
 * re-run because a row was not certified / outside the model / dynamic range; rows not certified */
int spx_last_tier_stats(int64_t *out5);

/* ---- BED side outputs (src/secphase.c:59-72,201-212,713-732; ptBlock.c:228-428,573-602) ------------ */
typedef struct spx_bedset spx_bedset;
int spx_bedset_create(spx_bedset **out);
void spx_bedset_free(spx_bedset *b);
int spx_bedset_add(spx_bedset *b, const char *contig, int32_t start, int32_t end /* inclusive */, int32_t count);
/* n single-base blocks (count 0) on one contig: the marker positions of one alignment */
int spx_bedset_add_points(spx_bedset *set, const char *contig, const int32_t *pos, int32_t n);
int64_t spx_bedset_size(const spx_bedset *b);
/* sort + ptBlock_merge_blocks_v2 per contig + ptBlock_save_in_bed; the file is created even when empty */
int spx_bedset_save(const spx_bedset *b, const char *path, int print_count);
/* the merge alone: disjoint pieces cut at every start and every end+1, each with the summed count of the
 * blocks covering it (c == NULL: no counts).  Returns the number of pieces (<= cap) or SPX_EINVAL. */
int spx_merge_blocks_count(int32_t n, const int32_t *s, const int32_t *e, const int32_t *c, int32_t *os, int32_t *oe,
                           int32_t *oc, int32_t cap);
/* after spx_collect + finalize: add the blocks of every relabelled group of `work` to the two sets
 * (either may be NULL).  Returns the number of relabelled groups. */
int spx_relabel_blocks(const spx_work *work, const spx_ref *ref, const spx_group_out *out, spx_bedset *modified_blocks,
                       spx_bedset *marker_blocks);

/* ---- input side: name-grouped BAM and FASTA readers (the reference reads through htslib on one thread:
 * sam_open/sam_read1 src/secphase.c:236-268, fai_load/fai_fetch src/secphase.c:101, ptMarker.c:739-744).  The BAM
 * reader maps the file, walks the BGZF block chain on the mapping, inflates runs of blocks on a persistent pool of
 * threads (libdeflate when the machine has it, zlib otherwise; CRC32 checked) into slots of one arena, follows the
 * record chain on one thread and hands out batches that POINT INTO the arena (SEQ / QUAL / tag text are not copied);
 * finished batches wait in a read-ahead queue.  A BAM record may not exceed the chunk size (32 MB by default). ---- */
typedef struct spx_bam_reader spx_bam_reader;
typedef struct spx_fasta spx_fasta;
#define SPX_BAM_WANT_VOFFSETS 1 /* keep the BGZF virtual offset of every group's first record (index building) */
#define SPX_BAM_NO_CRC 2        /* skip the CRC32 check of the inflated blocks */
#define SPX_BAM_HEADER_ONLY 4   /* map the file and parse the header only: no batches are cut (the reader then serves spx_dbam_open) */
typedef struct spx_bam_options {
    int32_t threads;        /* inflate / parse threads [4] */
    int32_t batch_groups;   /* > 0: batches of this many groups are cut in the background from spx_bam_open_opts on (the
                             * first spx_bam_next_batch must ask for the same number); 0: cutting starts with the first call */
    int32_t ahead_batches;  /* finished batches kept ahead of the caller [2] */
    int32_t flags;          /* SPX_BAM_* */
    int64_t chunk_bytes;    /* inflated bytes per inflate chunk [32 MB]; also the largest BAM record the reader accepts */
    int64_t max_bytes;      /* soft cap of the inflate arena [a quarter of the machine's memory] */
    int64_t start_voffset;  /* BGZF virtual offsets (compressed offset << 16 | offset in the inflated block) of a record    */
    int64_t end_voffset;    /* range [start, end): one shard of the file, cut at group starts; -1: header end / end of file */
    int32_t keep_batches;   /* batches that stay valid after they were handed out [SPX_BAM_KEEP]; a caller with more
                             * batches in flight (several devices) raises it and releases with spx_bam_release_batch */
    int32_t reserved;
} spx_bam_options;
void spx_bam_default_options(spx_bam_options *opt);
const char *spx_io_last_error(void);
int spx_bam_open(const char *path, int threads, spx_bam_reader **out);
int spx_bam_open_opts(const char *path, const spx_bam_options *opt, spx_bam_reader **out);
int32_t spx_bam_n_targets(const spx_bam_reader *r);
const char *spx_bam_target_name(const spx_bam_reader *r, int32_t i);
/* map BAM target ids to the contig indices of `ref` by name; returns the number of targets the FASTA lacks.  Applies to
 * the batches handed out from then on (the mapping happens at hand-out, the background work does not wait for it). */
int spx_bam_bind_reference(spx_bam_reader *r, const spx_ref *ref);
/* up to max_groups complete name groups (consecutive records with one read name, src/secphase.c:273-279);
 * the batch is owned by the reader and stays valid for the next SPX_BAM_KEEP calls (a pipelined caller has several
 * batches in flight) or until spx_bam_release_batch; returns groups read, 0 at EOF, <0 on error.  The bytes a batch
 * points to (SEQ, QUAL, tag text) belong to that batch alone: the caller may edit them in place until it lets the
 * batch go (the command line applies the BAQ qualities of -w that way, like calc_local_baq does in the bam1_t). */
#define SPX_BAM_KEEP 6
int spx_bam_next_batch(spx_bam_reader *r, int32_t max_groups, const spx_batch **out);
/* done with a batch: its part of the inflate arena is recycled now rather than SPX_BAM_KEEP calls later */
int spx_bam_release_batch(spx_bam_reader *r, const spx_batch *batch);
void spx_bam_close(spx_bam_reader *r);
/* For a caller about to exit without closing (every output written): returns the reader's pages -- inflate arena, file
 * mapping -- to the kernel on the reader's thread pool, in parallel.  Every batch is dead afterwards.  (Left to the
 * process exit the same pages are torn down by one thread while the parent waits: 0.5 s per 13 GB.)  spx_bam_close does
 * the same on its way out. */
void spx_bam_drop_pages(spx_bam_reader *r);
/* Device inflate beside the host pool (round 3: the GPU boxes give a container ~16 cores of CPU time; inflate is what they
 * are spent on).  Once attached, dispatched chunks (runs of BGZF blocks, ~32 MB inflated) wait in one queue: the host pool
 * claims from its front, an idle one of the n_workers device workers claims from its back while the pool is saturated, so
 * the split follows the speeds of the two sides.  fn inflates blocks[0..n) of the mapped file
 * into dst (blocks[k] goes to dst + uoff) and returns 0, 1 (corrupt DEFLATE data), 2 (CRC-32 mismatch) or a negative value
 * (the device could not do it: the reader inflates the chunk on the host).  spx_inflater_* (below) is that function on a
 * scoring context. */
typedef struct spx_bgzf_block {
    int64_t data_off;  /* offset of the block's DEFLATE data in the file */
    uint32_t clen;     /* bytes of DEFLATE data */
    uint32_t uoff;     /* where its inflated bytes go, relative to dst */
    uint32_t ulen;     /* ISIZE */
    uint32_t crc;      /* CRC-32 of the inflated bytes */
    uint32_t reserved;
} spx_bgzf_block;
typedef int (*spx_bgzf_inflate_fn)(void *user, int32_t worker, const uint8_t *file, int64_t file_bytes, const spx_bgzf_block *blocks,
                                   int32_t n_blocks, uint8_t *dst, int64_t dst_bytes, int32_t check_crc);
int spx_bam_attach_device_inflate(spx_bam_reader *r, spx_bgzf_inflate_fn fn, void *user, int32_t n_workers);
void spx_bam_inflate_counts(const spx_bam_reader *r, int64_t *chunks_host, int64_t *chunks_device);
/* the device side of the above on a scoring context: n_workers independent sets of {stream, pinned buffers, device
 * buffers}; spx_inflater_run is an spx_bgzf_inflate_fn with user = the inflater */
typedef struct spx_inflater spx_inflater;
int spx_inflater_create(spx_ctx *ctx, int32_t n_workers, spx_inflater **out);
int spx_inflater_run(void *inflater, int32_t worker, const uint8_t *file, int64_t file_bytes, const spx_bgzf_block *blocks, int32_t n_blocks,
                     uint8_t *dst, int64_t dst_bytes, int32_t check_crc);
void spx_inflater_free(spx_inflater *inf);

/* ---- device-resident BAM input (round 4; spx_devin.cpp, spx_devin_kernels.hip): what sam_read1 + the group scan + the
 * dispatch filter of src/secphase.c:230-351 do one record at a time on the reading thread happens on the device -- the host
 * sends COMPRESSED bytes (runs of BGZF blocks, "segments"), bgzf_inflate_kernel inflates them into HBM, kernels follow the
 * record chain, parse fields and cs / MD / CG tags, form the name groups, apply the dispatch filter (:285-288) and gather
 * the dispatched groups' records into the staged layout: the product is a STAGED work list per segment (hand it to
 * spx_pipe_submit as `staged`) plus a spx_batch that holds only what the relabel list prints (names, grp_first, flag,
 * tid, pos; every other array is NULL).  Work lists come out in FILE order; with several contexts (one per GPU) the
 * segments are dealt to whichever device is free, each device running its own upload / inflate / parse pipeline --
 * only the few hundred KB a segment hands to the next one (its last, still open group) cross devices, through the host. ---- */
typedef struct spx_dbam spx_dbam;
typedef struct spx_dbam_options {
    int32_t threads;        /* host threads for the copies into pinned memory [4] */
    int32_t max_groups;     /* groups per work list at most [95 000 = the cap]; a segment with more is handed out as several lists */
    int32_t ahead;          /* finished segments per device that may wait for the caller [3] */
    int32_t flags;          /* SPX_BAM_NO_CRC */
    int32_t host_inflate_percent; /* share of every segment's inflated bytes that the HOST pool inflates and uploads raw, beside the
                                   * inflate kernel; -1: from the CPUs the process may use and the number of devices */
    int32_t reserved;
    int64_t segment_bytes;  /* inflated bytes per segment [1 GB] */
    int64_t carry_bytes;    /* room for the open group + a continuing record in front of a segment [256 MB]; also the largest record */
    int64_t start_voffset, end_voffset; /* shard of the file, as in spx_bam_options; -1: all of it */
} spx_dbam_options;
void spx_dbam_default_options(spx_dbam_options *opt);
int spx_dbam_open(const char *path, const spx_dbam_options *opt, spx_dbam **out);
/* the header of the file (target names, spx_bam_bind_reference: bind BEFORE spx_dbam_start) */
spx_bam_reader *spx_dbam_header(spx_dbam *d);
/* starts the input pipelines, one per context */
int spx_dbam_start(spx_dbam *d, spx_ctx *const *ctxs, int32_t n_ctx, const spx_params *par);
/* the next work list in file order: staged on ctxs[*ctx_index]; returns its number of input groups, 0 at the end of the file,
 * SPX_E* on error (spx_last_error).  The caller frees the work list (spx_work_free on that context) and releases `names`. */
int spx_dbam_next(spx_dbam *d, spx_work **work, int32_t *ctx_index, const spx_batch **names);
int spx_dbam_release(spx_dbam *d, const spx_batch *names);
void spx_dbam_stats(const spx_dbam *d, int64_t *segments, int64_t *bytes_uploaded, double *seconds7);
void spx_dbam_close(spx_dbam *d);

/* Index of group starts in the reference's on-disk format (src/secphase_index.c:76-119 writes, get_offset_array
 * src/secphase.c:357-385 reads: int64 count, then that many int64 BGZF virtual offsets): the offset of the first record
 * of every step_groups-th read group plus the offset where the records end.  A reader opened with start_voffset = a[i]
 * and end_voffset = a[j] yields exactly the groups [i*step, j*step): N readers / N devices can share one file.
 * spx_bam_index_build returns the number of offsets (offsets == NULL: only counts) or SPX_E*. */
int64_t spx_bam_index_build(const char *path, int threads, int32_t step_groups, int64_t *offsets, int64_t capacity);
int spx_bam_index_save(const char *index_path, const int64_t *offsets, int64_t n);
int64_t spx_bam_index_load(const char *index_path, int64_t *offsets, int64_t capacity);
/* -w/--writeBam (src/secphase.c:182-189,643-657): the reference opens the output with sam_open(path, "w"), i.e.
 * SAM text despite the .bam name, writes the input header (sam_hdr_write) and then sam_write1()s every stored
 * alignment of every dispatched group with the qualities calc_local_baq left in the record.
 * spx_sam_write_group formats group g of the reader's CURRENT batch; qual is laid out like that batch's qual[]
 * (what spx_apply_quals produced) or NULL for the record's own qualities.  Returns records written. */
typedef struct spx_sam_writer spx_sam_writer;
int spx_sam_open(const char *path, const spx_bam_reader *src, spx_sam_writer **out);
int spx_sam_write_group(spx_sam_writer *w, const spx_bam_reader *src, int32_t g, const uint8_t *qual);
/* the same for group g of an EARLIER batch of the reader that is still alive (SPX_BAM_KEEP) */
int spx_sam_write_group_of(spx_sam_writer *w, const spx_bam_reader *src, const spx_batch *bt, int32_t g, const uint8_t *qual);
int spx_sam_close(spx_sam_writer *w);
int spx_fasta_load(const char *path, spx_fasta **out);
const spx_ref *spx_fasta_ref(const spx_fasta *f);
void spx_fasta_free(spx_fasta *f);

/* ---- the consumer of the relabel list: /root/reference/programs/src/correct_bam.c ---------------
 * spx_relabel_table_*: get_phased_read_table (correct_bam.c:32-91) -- the list is parsed the way correct_bam parses it: `$` read
 * name, `*` old primary, `@` promoted secondary, columns 3-4 = contig and 0-based start; records whose two locations coincide
 * are ignored (:64-68,77-82).  path == NULL: an empty table.  _get iterates in name order. */
typedef struct spx_relabel_table spx_relabel_table;
int spx_relabel_table_load(const char *out_log_path, spx_relabel_table **out);
int64_t spx_relabel_table_size(const spx_relabel_table *t);
int spx_relabel_table_get(const spx_relabel_table *t, int64_t i, const char **qname, const char **contig, int32_t *start);
int spx_relabel_table_find(const spx_relabel_table *t, const char *qname, const char **contig, int32_t *start); /* 1 found, 0 not */
void spx_relabel_table_free(spx_relabel_table *t);
/* spx_correct_bam: correct_bam's record loop (correct_bam.c:347-376) on this library's BAM reader: unmapped and excluded reads are
 * dropped; a read the table names gets BAM_FSECONDARY cleared on the record at the table's location and set on all its other
 * records (is_prim, :93-109); then --primaryOnly, the read-length / alignment-length filters, the MAPQ table, --maxMapq, the `de`
 * divergence filter and --noTag, in the reference's order.  Output: BAM (BGZF; the reference opens "wb") or, with sam_text, SAM.
 * Options mirror correct_bam's command line (:222-238); defaults from spx_correct_default_options (:248-252). */
typedef struct spx_correct_options {
    const char *phasing_log;      /* -P  <prefix>.out.log, may be NULL */
    const char *mapq_table;       /* -M  read \t contig \t 1-based start \t mapq, may be NULL */
    const char *exclude;          /* -e  read names, one per line, may be NULL */
    int32_t primary_only;         /* -p */
    int32_t no_tag;               /* -t */
    int32_t min_read_length;      /* -m [5000] */
    int32_t min_alignment_length; /* -a [5000] */
    int32_t max_mapq;             /* -x [100] */
    int32_t threads;              /* -n [2] */
    double max_divergence;        /* -d [0.12] */
    int32_t sam_text;             /* not in the reference: write SAM text instead of BAM */
    int32_t reserved;
} spx_correct_options;
typedef struct spx_correct_stats {
    int64_t records_in, records_out, made_primary, made_secondary, table_reads;
} spx_correct_stats;
void spx_correct_default_options(spx_correct_options *opt);
int spx_correct_bam(const char *in_bam, const char *out_path, const spx_correct_options *opt, spx_correct_stats *stats);

/* ---- BGZF inflate on the device (spx_inflate_kernels.hip: one wavefront per BGZF block; decoder core shared with the
 * host build in spx_inflate.h).  htslib inflates on the reading thread (bgzf_read_block under sam_read1,
 * src/secphase.c:268).  `file` + block_off[0 .. n_blocks] delimit consecutive BGZF blocks; their inflated bytes are
 * written back to back to out (host memory); status[b]: 0 ok, -1 corrupt DEFLATE data, -2/-3 size mismatch, -4 CRC-32
 * mismatch.  Returns the inflated size or SPX_E*. */
int64_t spx_inflate_bgzf_device(spx_ctx *ctx, const uint8_t *file, const int64_t *block_off, int32_t n_blocks, uint8_t *out,
                                int64_t out_cap, int32_t *status, double *kernel_ms);
/* the same decoder core compiled for the host (diagnostics: lets CPU tests check it against zlib): one raw DEFLATE
 * stream; and the kernel's striped CRC-32 */
int spx_inflate_core_host(const uint8_t *in, int64_t in_len, uint8_t *out, int64_t out_len);
uint32_t spx_crc32_core_host(const uint8_t *p, int64_t n, int32_t pieces);

/* ---- host-only view of the work list (no device needed) ------------------
 * What spx_prepare would upload: the banded DP problems and the marker table.
 * Lets CPU-only tests check the host logic against the oracle, and documents
 * the device batch layout for integrators. */
typedef struct spx_plan spx_plan;
typedef struct spx_plan_view {
    int32_t n_problems, n_rows, n_groups, n_markers;
    const int32_t *L, *R, *bw;           /* per problem; bw is the effective half band width */
    const int32_t *ref_tid, *ref_rfs;    /* reference window = contig tid, [rfs, rfs+R) */
    const int64_t *qry_nib;              /* nibble offset of the query window in qry4 */
    const uint8_t *qry4;                 /* 0..4 codes, low nibble first */
    const double *hmm;                   /* 16 doubles per problem, see spx_device.h */
    const int32_t *row_off, *n_rows_of;  /* wanted rows of each problem */
    const int32_t *rows, *row_expect;    /* 1-based query row; expected window-relative ref index */
    const uint8_t *row_rawq;
    const int32_t *grp_index;            /* dispatched group -> group of the input batch */
    const int32_t *mk_first;             /* [n_groups+1] */
    const int32_t *mk_row;               /* [n_markers] wanted-row index or -1 */
    const uint8_t *mk_qfix, *mk_is_match, *mk_aln, *mk_first_of_pos;
    const uint8_t *n_aln;
    const uint16_t *sec_mask;
    const int32_t *rfe;                  /* 10 per group */
    const int32_t *grp_error;            /* per INPUT group: 0 ok, 1 not dispatched, <0 SPX_E* */
    /* SPX_PAR_ALL_ROWS only: calc_local_baq's writes to the record qualities, in order.  len 0: qual[rec][pos]=0;
     * len>0: qual[rec][pos..pos+len) = BAQ value of wanted rows row0.. (row_expect<0: min(set_q,93)) */
    int32_t n_qedits, pad_;
    const int32_t *qe_rec, *qe_pos, *qe_len, *qe_row0;
} spx_plan_view;
int spx_plan_create(const spx_ref *ref, const spx_batch *bt, const spx_params *par, spx_plan **out);
int spx_plan_get(const spx_plan *plan, spx_plan_view *view);
/* diagnostics for parity tests: the work list the DEVICE built for `work`, copied back in the shape of a host plan
 * (free it with spx_plan_free), so that it can be compared with spx_plan_create's field by field */
int spx_work_export(spx_ctx *ctx, spx_work *work, spx_plan **out);
void spx_plan_free(spx_plan *plan);
/* the tables the kernels use: phred thresholds thr[102], match_tbl[256], mis_tbl[256] */
void spx_host_tables(double *thr, double *match_tbl, double *mis_tbl);

#ifdef __cplusplus
}
#endif
#endif
