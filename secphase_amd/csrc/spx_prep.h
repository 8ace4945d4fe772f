/*
 * spx_prep.h -- host side of the work-list preparation: staging of record batches into the packed pools the
 * device reads (spx_logic.h Rec + payload), the host-only plan (the same spx_logic.h functions run on the CPU),
 * and the tables the kernels need.  Internal.
 */
#ifndef SPX_PREP_H
#define SPX_PREP_H

#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <vector>

#include "../../include/spx.h"
#include "spx_device.h"
#include "spx_logic.h"

namespace spx {

/* a work list as the host sees it: filled completely by the host plan (spx_plan_create, spx_probaln_*), and for
 * device-prepared work lists only with the per-group arrays spx_collect / spx_relabel_blocks / spx_apply_quals pull back */
struct HostBatch {
    /* DP problems */
    std::vector<int64_t> ref_nib, qry_nib;
    std::vector<int32_t> ref_tid, ref_rfs; /* where the ref window lies (debug / host-plan view) */
    std::vector<int32_t> L, R, bw, row_off, n_rows;
    std::vector<double> hmm; /* SPX_H_N per problem */
    std::vector<int32_t> rows, row_expect;
    std::vector<uint8_t> row_rawq;
    std::vector<uint8_t> qry4; /* 4-bit codes, low nibble first */
    int64_t qry_nibbles = 0;
    /* dispatched groups */
    std::vector<int32_t> grp_index; /* index in the input batch */
    std::vector<int32_t> mk_first;  /* per dispatched group (+1) */
    std::vector<spx_dev_marker> markers;
    std::vector<uint8_t> n_aln;
    std::vector<uint16_t> sec_mask;
    std::vector<int32_t> rfe;       /* 10 per dispatched group */
    std::vector<int32_t> rfs, atid; /* 10 per dispatched group: ptAlignment.rfs, contig index */
    std::vector<int32_t> mk_ref_pos; /* per marker: reference position (BED side outputs) */
    std::vector<int32_t> grp_problems;
    std::vector<int64_t> grp_cells;
    std::vector<int32_t> grp_error; /* per input group: 0, 1 (not dispatched) or SPX_E* */
    /* SPX_PAR_ALL_ROWS only: the writes calc_local_baq makes to the record's quality array, in the order it
     * makes them (ptMarker.c:706,759,763).  len == 0: qual[rec][pos] = 0; len > 0: rows row0.. hold the values of
     * qual[rec][pos .. pos+len) (row_expect < 0: base not under an M/=/X op, keeps set_q) */
    std::vector<int32_t> qe_rec, qe_pos, qe_len, qe_row0, qe_batch;
    int64_t dp_cells = 0;
};

/* Pads of the 4-bit reference pool.  The kernels fetch codes in chunks of 16 a whole band width before / after a
 * window (columns < 1 and > R: never used, but read): up to bw + 16 <= 1039 nibbles in front of the first contig and
 * slots + 32 behind the last one (slots <= 2048). */
constexpr int64_t kRefLeadNibbles = 4096;
constexpr size_t kRefTailBytes = 2048;
/* same for the pool of recoded read sequences the query windows point into */
constexpr int64_t kCodeLeadBytes = 128;
constexpr int64_t kCodeTailBytes = 256;

/* nibble offset of every contig inside the device reference pool + the ambiguous-base index */
struct RefIndex {
    std::vector<int64_t> nib_off; /* [n_contigs] first nibble of each contig (byte aligned) */
    std::vector<int64_t> len;     /* [n_contigs] bases */
    std::vector<int64_t> npos_off; /* [n_contigs+1] */
    std::vector<int32_t> npos;     /* positions of non-ACGT bases per contig, ascending */
    void build(const spx_ref *ref);
    spxl::RefView view() const;
};

/* ---- staging: record batches -> one packed buffer (host pinned memory in the runtime, plain memory for the plan) ---- */
struct StageLayout {
    int64_t n_groups_in = 0; /* input groups over all batches */
    int64_t n_dgroups = 0;   /* groups passing the dispatch filter (src/secphase.c:285-288) */
    int64_t n_slots = 0;     /* their alignments */
    int64_t cigar_words = 0, seq_bytes = 0, qual_bytes = 0, text_bytes = 0;
    int64_t pk_seq_bytes = 0, pk_qual_bytes = 0; /* SEQ / QUAL bytes that are really transferred (aliased secondaries left out) */
    int64_t n_aliased = 0;
    int64_t ops_bound = 0, conf_bound = 0, mm_bound = 0; /* sums of spxl::aln_caps over the alignments */
    /* byte offsets inside the staged buffer */
    size_t o_recs = 0, o_slot0 = 0, o_gidx = 0, o_cigar = 0, o_seq = 0, o_qual = 0, o_text = 0, bytes = 0;
};
struct Stage {
    StageLayout lay;
    std::vector<spxl::Rec> recs;      /* offsets already final */
    std::vector<int32_t> slot0;       /* [n_dgroups+1] */
    std::vector<int32_t> grp_index;   /* [n_dgroups] input group (global over the batches) */
    std::vector<int32_t> grp_error;   /* per input group: 0 dispatched / 1 not */
    std::vector<const spx_batch *> batches;
    std::vector<int32_t> batch_base;  /* first input group of every batch */
};
/* measure: dispatch filter, per-alignment payload sizes (strlen of the tags), layout */
int stage_measure(const spx_batch *const *bts, int32_t n_batches, int threads, Stage &st);
/* byte offsets of the image's sections from its counts (n_slots, n_dgroups, cigar_words, seq / qual / text bytes) */
void stage_layout_offsets(StageLayout &L);
/* copy the payload into dst (lay.bytes bytes) on `threads` threads */
void stage_copy(const Stage &st, char *dst, int threads);
/* The same image piecewise, for staging through a ring of pinned chunks: byte range [b0, b1) of payload section `sec`
 * (0 CIGAR words, 1 SEQ, 2 QUAL, 3 tag text, 4 / 5 SEQ / QUAL in the PACKED transfer layout that leaves aliased
 * secondaries out; section-relative offsets) into dst (= the byte at b0).  Records that
 * straddle the range are copied in part.  f_parallel(n, grain, fn(k0,k1)) runs the record loop (a thread pool). */
int64_t stage_section_bytes(const Stage &st, int sec);
size_t stage_section_offset(const Stage &st, int sec);
void stage_fill(const Stage &st, int sec, int64_t b0, int64_t b1, char *dst,
                const std::function<void(int64_t, int64_t, const std::function<void(int64_t, int64_t)> &)> &f_parallel);

/* what the work list needs from spx_params, in the shape spx_logic.h wants (qf through the host libm) */
spxl::Params logic_params(const spx_params *par);
int terminal_guard();             /* 0 = SPX_GUARD_BAND (default), 1 = SPX_GUARD_ROW: spx_logic.h terminal_drop */
void set_terminal_guard(int reading);

/* host plan: the spx_logic.h passes run on the CPU; fills every array of hb */
int host_plan(const spx_batch *const *bts, int32_t n_batches, const RefIndex &ref, const spx_params *par, int threads,
              HostBatch &hb);

void hmm_constants(int l_ref, int l_query, float d, float e, int set_q, double *h /* SPX_H_N */);
inline int effective_bw(int l_ref, int l_query, int bw_in) { return spxl::effective_bw(l_ref, l_query, bw_in); }
inline int64_t band_cells(int L, int R, int bw_eff) { return spxl::band_cells(L, R, bw_eff); }
inline int band_class(int W) { return spxl::band_class(W); } /* index into the (G,C) table, -1 if too wide */
inline int class_slots(int cls) { return spxl::class_slots(cls); }
inline int class_lanes(int cls) { return spxl::class_lanes(cls); } /* lanes of a wavefront that share one problem (forward kernel) */
inline int class_lanes_bwd(int cls) { return spxl::class_lanes_bwd(cls); }
void phred_thresholds(double *thr /* 102 */);
void score_tables(double *match_tbl /*256*/, double *mis_tbl /*256*/);
bool group_dispatched(const spx_batch *bt, int g);
/* SEQ as BAM stores it (nt16, high nibble first) -> 0..4 codes, low nibble first; returns whether a code > 3 occurred */
bool recode_seq(const uint8_t *src, uint8_t *dst, int64_t n_bytes);

} // namespace spx
#endif
