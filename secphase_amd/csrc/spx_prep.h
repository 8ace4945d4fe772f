/*
 * spx_prep.h -- host-side preparation of a batch of alignment groups for the
 * device: CIGAR/cs walk, markers, consensus blocks, and the list of banded
 * DP problems + the marker table the scoring kernel consumes.  Internal.
 */
#ifndef SPX_PREP_H
#define SPX_PREP_H

#include <stdint.h>

#include <vector>

#include "../../include/spx.h"
#include "spx_device.h"

namespace spx {

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
    std::vector<int32_t> grp_error; /* per input group: 0 or SPX_E* */
    /* SPX_PAR_ALL_ROWS only: the writes calc_local_baq makes to the record's quality array, in the order it
     * makes them (ptMarker.c:706,759,763).  len == 0: qual[rec][pos] = 0; len > 0: rows row0.. hold the values of
     * qual[rec][pos .. pos+len) (row_expect < 0: base not under an M/=/X op, keeps set_q) */
    std::vector<int32_t> qe_rec, qe_pos, qe_len, qe_row0, qe_batch;
    int64_t dp_cells = 0;

    void clear();
    void append(const HostBatch &o);
    /* *this = parts[0] + parts[1] + ... (same result as appending them one by one), copied on n_threads threads */
    void assign_merged(std::vector<HostBatch> &parts, int n_threads);
};

/* Pads of the 4-bit reference pool.  The kernels fetch codes in chunks of 16 a whole band width before / after a
 * window (columns < 1 and > R: never used, but read): up to bw + 16 <= 1039 nibbles in front of the first contig and
 * slots + 32 behind the last one (slots <= 2048). */
constexpr int64_t kRefLeadNibbles = 4096;
constexpr size_t kRefTailBytes = 2048;

/* nibble offset of every contig inside the device reference pool */
struct RefIndex {
    std::vector<int64_t> nib_off; /* [n_contigs] first nibble of each contig (byte aligned) */
    std::vector<int64_t> len;     /* [n_contigs] bases */
    /* positions of ambiguous (non-ACGT) bases per contig, ascending: lets prep flag windows that need
     * the general emission path */
    std::vector<std::vector<int32_t>> npos;
    bool window_has_n(int tid, int64_t start, int64_t n) const
    {
        if ((size_t)tid >= npos.size()) return false;
        const std::vector<int32_t> &v = npos[tid];
        size_t lo = 0, hi = v.size();
        while (lo < hi) { size_t m = (lo + hi) / 2; if (v[m] < start) lo = m + 1; else hi = m; }
        return lo < v.size() && v[lo] < start + n;
    }
    void index_ambiguous(const spx_ref *ref);
};

/* prepares groups [g0,g1) of bt */
int prepare_groups(const spx_batch *bt, const RefIndex &ref, const spx_params *par, int32_t g0, int32_t g1,
                   HostBatch &out);

void hmm_constants(int l_ref, int l_query, float d, float e, int set_q, double *h /* SPX_H_N */);
int effective_bw(int l_ref, int l_query, int bw_in);
int64_t band_cells(int L, int R, int bw_eff);
int band_class(int W); /* index into the (G,C) table, -1 if too wide */
int class_slots(int cls);
int class_lanes(int cls); /* lanes of a wavefront that share one problem (forward kernel) */
int class_lanes_bwd(int cls);
void phred_thresholds(double *thr /* 102 */);
void score_tables(double *match_tbl /*256*/, double *mis_tbl /*256*/);

} // namespace spx
#endif
