/*
 * spx_prep_dev.h -- argument blocks of the preparation kernels (spx_prep_kernels.hip), shared with the runtime.
 * Internal.
 */
#ifndef SPX_PREP_DEV_H
#define SPX_PREP_DEV_H

#include <stdint.h>

#include "spx_logic.h"

/* sizes the host reads back once per work list (one small D2H copy) */
struct spx_prep_totals {
    int64_t n_ops, n_conf, n_mm;      /* pool needs of the per-alignment pass */
    int64_t arena_bytes;              /* scratch need of the per-group pass */
    int64_t n_prob, n_rows, n_qe, n_mk, s_tot, f_tot, cells;
    int64_t n_ok;                     /* dispatched groups without an error */
    int32_t overflow;                 /* bits: 1 group scratch arena too small, 2 a group's interval lists outgrew their estimate, 4 an op / block / mismatch pool too small, 8 an alignment outgrew the length bounds of its tables */
    int32_t pad;
    int64_t cls_prob[SPX_N_CLASSES], cls_cells[SPX_N_CLASSES];
};

struct spx_group_info { /* per dispatched group, copied back with the results */
    int32_t err, n_aln, n_prob, n_mk;
    int64_t cells;
};

struct spx_prep_args {
    int32_t n_slots, n_dgroups;
    const spxl::Rec *recs;
    const int32_t *slot0; /* [n_dgroups+1] */
    spxl::AlnState *ast;
    spxl::Pools P;
    uint8_t *code4_w;     /* P.code4, writable */
    int64_t ops_cap, conf_cap, mm_cap;
    spxl::RefView rv;
    spxl::Params par;
    spxl::GroupCount *gc;  /* per dispatched group */
    spxl::GroupCount *ac;  /* per alignment: its share of the work list */
    int64_t *ga_bytes, *ga_off;
    char *arena;
    int64_t arena_cap;
    int32_t slack, pad;
    int32_t exact_counts, tight_caps;  /* phase 1: table sizes from the counting pass (fallback) instead of the length bounds; tests: shrunken bounds */
    /* scratch of the prefix sums: five int64 columns of scan_stride entries, their tile totals, the grand totals */
    int64_t *scan_v, *scan_tile, *scan_grand;
    int64_t scan_stride;
    spx_prep_totals *tot;
    /* round 5: the HEAVIEST alignments / groups of a list walk alone in a wave of their own (lane 0 walks, the others idle): a lone lane does
     * not wait for 63 others at every branch of the token loop and its loads touch one cache line per instruction instead of 64 -- a list's
     * preparation lasts as long as its longest walk, and wave slots are what these latency-bound kernels have plenty of.  slot_heavy /
     * group_heavy: the n_heavy_* heaviest items, heaviest first (a radix sort of (work estimate, index)); *_flag: 1 for an item that is in
     * that list (its lane in the ordinary waves, which keep list order -- neighbours share cache lines -- then idles).  NULL: no extraction. */
    const int32_t *slot_heavy, *group_heavy;
    const uint8_t *slot_flag, *group_flag;
    int32_t n_heavy_slots, n_heavy_groups;
    /* two tiers (the flag is the tier: 2 = one of the n_heavy_* heaviest, 1 = one of the n_share_* heaviest, 0 neither): the kernels whose walk
     * can be SHARED by the 64 lanes of a wave (marker fill / filter, BAQ plan count / emit, marker table) extract many more items than those
     * where the item's wave has one working lane */
    int32_t n_share_slots, n_share_groups;
    spxl::PlanBase *heavy_plan; /* [n_share_slots][64]: what each lane's share of a heavy alignment's blocks adds to the output offsets (counting pass -> emitting pass) */
};

struct spx_emit_args {
    const spxl::PlanBase *base; /* per alignment */
    const int64_t *mk_base;
    spxl::PlanOut out;
    double *hmm;      /* [n_prob][SPX_H_N], filled by problem_constants_kernel */
    int32_t n_prob, pad;
    int64_t n_rows;   /* wanted rows of the work list (rows_unpack_kernel) */
    spx_dev_marker *markers;
    int32_t *mk_ref_pos;
    int32_t *mk_first;
    uint8_t *n_aln;
    uint16_t *sec_mask;
    int32_t *rfe, *rfs, *atid;
    spx_group_info *info;
};

struct spx_order_segs { /* where each band class' launch order lives inside the order array */
    int64_t off[SPX_N_CLASSES];
    int64_t cap[SPX_N_CLASSES];
};

#define SPX_MAX_SLICES 32 /* DP slices of a work list */
struct spx_order_args {
    int32_t n_prob;
    int32_t fwd_by_last_row; /* two-tier DP: the fast forward kernel stops at a problem's LAST wanted row: the forward order goes by that, not by the query length */
    /* round 5: ONE pair of sorts per work list instead of one per DP slice: the slice of a problem is the top field of its key */
    int32_t n_slices;
    int32_t slice_prob[SPX_MAX_SLICES + 1]; /* first problem of every slice, [n_slices] = n_prob */
    const spx_order_segs *segs_f, *segs_b;  /* [n_slices], device memory */
    const int32_t *bw, *L, *n_rows, *row_off, *rows;
    uint64_t *key_f, *key_b, *key_sorted;
    int32_t *val, *val_sorted;
    int32_t *bin_start, *bin_end, *pad_base; /* SPX_N_CLASSES * 1024 each */
    void *temp;
    size_t temp_bytes;
    int32_t *order_f, *order_b;
};

#endif
