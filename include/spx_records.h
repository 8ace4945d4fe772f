/*
 * spx_records.h -- neutral in-memory form of name-grouped alignment records
 * and of the reference assembly.
 *
 * This is a DATA FORMAT, not code: the same fields secphase reads from a
 * bam1_t (htslib sam.h) and from faidx, laid out as flat structure-of-arrays
 * so that (a) a BAM reader can fill it without per-record mallocs, (b) the
 * synthetic generator, the CPU oracle and the MI355X path all consume the
 * very same bytes.  Reference usage of each field:
 *   flag/tid/pos/l_qseq/n_cigar/cigar/seq/qual/cs/qname
 *        programs/submodules/cigar_it/cigar_it.c:14-69   (iterator start state)
 *        programs/submodules/ptMarker/ptMarker.c:42-107  (qual, seq access)
 *        programs/src/secphase.c:268-338                 (grouping by qname)
 *   contig names/sequences
 *        programs/submodules/ptMarker/ptMarker.c:739-744 (fai_fetch window)
 */
#ifndef SPX_RECORDS_H
#define SPX_RECORDS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* BAM flag bits used on the path (sam.h) */
#define SPX_FUNMAP 0x4
#define SPX_FREVERSE 0x10
#define SPX_FSECONDARY 0x100
#define SPX_FSUPPLEMENTARY 0x800

/* CIGAR op codes (sam.h BAM_C*) */
#define SPX_CMATCH 0
#define SPX_CINS 1
#define SPX_CDEL 2
#define SPX_CREF_SKIP 3
#define SPX_CSOFT_CLIP 4
#define SPX_CHARD_CLIP 5
#define SPX_CPAD 6
#define SPX_CEQUAL 7
#define SPX_CDIFF 8

/* A batch of name-grouped alignment records (consecutive records with equal
 * qname form one group, secphase.c:273-279).  All arrays are owned by whoever
 * built the batch.  qual is the only mutable array on the reference path
 * (calc_local_baq writes BAQ into it, ptMarker.c:716,786,802); consumers that
 * need to mutate it take a private copy. */
typedef struct spx_batch {
    int32_t n_groups;
    int32_t n_alns;
    const int32_t *grp_first; /* [n_groups+1] first alignment of each group      */
    const int64_t *qname_off; /* [n_groups]   offset of NUL-terminated qname     */
    const char *qnames;
    /* per alignment, bam1_core_t fields */
    const uint16_t *flag;
    const int32_t *tid;
    const int32_t *pos;     /* 0-based leftmost reference coordinate            */
    const int32_t *l_qseq;  /* bases stored in SEQ (hard clips excluded)        */
    const int32_t *n_cigar;
    const int64_t *cigar_off; /* index into cigar[] (uint32 units)              */
    const int64_t *seq_off;   /* BYTE offset into seq4[] (4-bit packed, BAM)    */
    const int64_t *qual_off;  /* BYTE offset into qual[]                        */
    const int64_t *cs_off;    /* BYTE offset into cs[] ; -1 when the tag is absent */
    const uint32_t *cigar;    /* len<<4 | op, as in BAM                         */
    const uint8_t *seq4;      /* nt16 codes, two bases per byte, high nibble first */
    const uint8_t *qual;      /* phred, one byte per base                       */
    const char *cs;           /* short-form cs strings (without the leading 'Z'), NUL-terminated */
    /* MD tag, used only for records without cs (cigar_it.c:46-63); both may be NULL when no record has one */
    const int64_t *md_off;    /* BYTE offset into md[] ; -1 when the tag is absent */
    const char *md;           /* MD strings, NUL-terminated */
} spx_batch;

/* The reference assembly, resident in RAM.  bases[] holds the raw FASTA
 * characters (what fai_fetch returns); codes are derived by the consumer via
 * the nt16 tables (ptMarker.c:744). */
typedef struct spx_ref {
    int32_t n_contigs;
    const int64_t *name_off; /* [n_contigs] offset of NUL-terminated contig name */
    const char *names;
    const int64_t *seq_off; /* [n_contigs+1] */
    const char *bases;
} spx_ref;

/* Scoring parameters = the subset of work_arg_t (tpool.h:26-55) the marker
 * path reads; defaults and presets are secphase.c:420-449,477-504. */
/* spx_params.flags: every base of every realigned window gets its BAQ value (the quality-modified output
 * of -w/--writeBam, secphase.c:182-189), not only the marker bases the scores need */
#define SPX_PAR_ALL_ROWS 1

typedef struct spx_params {
    int32_t baq_flag;
    int32_t consensus;
    int32_t indel_threshold;
    int32_t min_q;
    int32_t min_score;
    int32_t set_q;
    int32_t flank_margin;
    int32_t flags;     /* SPX_PAR_*; 0 on the scoring path */
    double prim_margin_score;
    double prim_margin_random;
    double conf_d;
    double conf_e;
    double conf_b;
} spx_params;

#ifdef __cplusplus
}
#endif
#endif
