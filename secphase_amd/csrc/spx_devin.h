/*
 * spx_devin.h -- device-resident BAM input: argument blocks shared by the host side (spx_devin.cpp) and the kernels
 * (spx_devin_kernels.hip).  Internal.
 *
 * What it replaces: sam_read1 + the group scan + the dispatch filter of the reference's main loop
 * (/root/reference/programs/src/secphase.c:230-351, filter :285-288) and, in this repository, the host reader's record
 * walk, field / tag parse and the staging copies (spx_io.cpp fill_batch, spx_prep.cpp stage_measure / stage_fill): the
 * inflated bytes of a run of BGZF blocks STAY in HBM; record chain, fields, tags, name groups, dispatch filter and the
 * gather into the staged layout the preparation kernels read (spxl::Rec + payload pools) are kernels.  Only names, flags
 * and positions come back (the relabel list prints them).
 *
 * A SEGMENT = a run of whole BGZF blocks (~1 GB inflated) of one file, in one device buffer:
 *
 *        [ ... unused ... | carry ][ inflated bytes of the segment's blocks ]
 *        0                 ^p0     ^SPX_DIN_CARRY_CAP                        ^n_end
 *
 * carry = the tail of the previous segment: every record of its last (still open) name group and the front part of a
 * record that continues in this segment.  It is placed so that it ENDS where the inflated bytes start; inflating a
 * segment therefore does not wait for the previous segment's record chain.
 */
#ifndef SPX_DEVIN_H
#define SPX_DEVIN_H

#include <stdint.h>

#include "spx_logic.h"

struct spx_din_counts { /* device -> host, a few small copies per segment */
    /* record chain */
    int64_t n_rec;       /* complete records on the chain */
    int64_t tail_start;  /* first byte behind the last complete record */
    int64_t err_at;
    int32_t err;         /* 0 ok, 1 corrupt record length, 2 chain did not end (internal), 3 corrupt record fields */
    int32_t inflate_bad; /* worst block status of the inflate kernel (0 ok, else -status) */
    /* groups */
    int64_t n_groups;    /* name groups among the n_rec records, the last (possibly open) one included */
    int64_t n_batch;     /* groups that are complete = handed on (all of them in the final segment) */
    int64_t n_batch_rec; /* their records */
    int64_t carry_start; /* where the next segment's carry begins (first record of the open group, or tail_start) */
    int64_t n_dgroups, n_slots;
    int64_t cigar_words, seq_bytes, qual_bytes, text_bytes, ops_bound, conf_bound, mm_bound, name_bytes;
};

struct spx_din_recs { /* per record of a segment */
    int64_t *off;    /* offset of the record's block_size field in the segment buffer */
    int64_t *cig_at; /* offset of its CIGAR words (inside the record, or the CG:B,I tag's payload) */
    int64_t *tag_at; /* offset of the cs (or MD) text */
    int32_t *flag, *tid, *pos, *lq, *ncig, *cs_len, *md_len, *lname;
    int32_t *isnew;  /* 1: the record opens a name group */
    int32_t *gid;    /* inclusive scan of isnew: group index + 1 */
};

struct spx_din_group_scan { /* scanned per group */
    int64_t disp, slots, name_bytes;
};
struct spx_din_slot_scan { /* scanned per slot (alignment of a dispatched group) */
    int64_t cw, sb, qb, tb, oc, cc, mc;
};

struct spx_din_args {
    const uint8_t *buf;      /* the segment buffer */
    int64_t p0, n_end;       /* record chain starts at p0, data ends at n_end */
    const int64_t *bstart;   /* [n_blocks + 1] offsets of the BGZF blocks' inflated bytes in buf (last: n_end) */
    int32_t n_blocks, is_final;
    int64_t max_rec;         /* largest block_size accepted */
    /* speculative walk from every block start */
    int64_t *land;           /* [n_blocks] where the walk from bstart[b] left the block */
    int32_t *cnt, *bflag;    /* records started in the block; 1 incomplete record at land, 2 corrupt length at land */
    int32_t *first_idx;      /* [n_blocks] index of the block's first record when the real chain passes through its start, else -1 */
    spx_din_recs R;
    int64_t rec_cap;
    const int32_t *tmap;     /* BAM target id -> contig index of the scorer's reference */
    int32_t n_targets, pad;
    /* groups of the batch */
    int32_t *grp_first;      /* [n_groups + 1] */
    spx_din_group_scan *gscan; /* [n_groups + 1] in: per group, out: exclusive prefix */
    /* slots */
    int32_t *slot_rec;       /* [n_slots] record of every slot */
    int32_t *slot_grp;       /* [n_slots] dispatched-group index */
    spx_din_slot_scan *sscan;  /* [n_slots + 1] */
    spx_din_counts *counts;
};

struct spx_din_range { /* the groups [g0, g1) of a segment = one work list: their records [r0, r1), slots [s0, s1) and the
                         * prefix sums in front of them / behind them */
    int64_t g0, g1, r0, r1, s0, s1;
    spx_din_group_scan gbase, gend;
    spx_din_slot_scan sbase;
};

struct spx_din_out { /* where the staged image goes (spx_prep.h StageLayout) */
    spxl::Rec *recs;
    int32_t *slot0, *gidx;
    uint32_t *cigar;
    uint8_t *seq, *qual;
    char *text;
    /* what the host keeps: names and per-record flag / tid / pos of every group handed on */
    char *names;
    int64_t *name_off;   /* [n_batch] */
    int32_t *h_grp_first; /* [n_batch + 1] first record of every group, relative to the range's first record */
    uint8_t *grp_disp;   /* [n_batch] 1: dispatched */
    uint16_t *h_flag;    /* [n_batch_rec] */
    int32_t *h_tid, *h_pos;
};

#endif
