/*
 * spx_synth.h -- deterministic synthetic assemblies and name-grouped
 * alignment records: diploid assembly + paralog copies,
 * HiFi / ONT / mixed reads, alignments built from ground truth (CIGAR with
 * M/I/D/S/H and short-form cs), in the flat format of include/spx_records.h.
 * Used by tests and bench.py; not part of the scoring path.
 */
#ifndef SPX_SYNTH_H
#define SPX_SYNTH_H

#include <stdint.h>

#include "../include/spx_records.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { SPX_SYNTH_HIFI = 0, SPX_SYNTH_ONT = 1, SPX_SYNTH_MIXED = 2 };

typedef struct spx_synth_cfg {
    uint64_t seed;        /* master seed (20241220) */
    int32_t n_contigs;    /* contigs per haplotype */
    int32_t contig_len;   /* bases per hap-1 contig */
    int32_t n_paralogs;   /* paralog copy sets (>= max_secondaries-1) */
    int32_t platform;     /* SPX_SYNTH_* */
    int32_t read_len;     /* fixed length; <=0: power-law 2000*(1-u)^(-1/1.2) capped at max_read_len */
    int32_t max_read_len;
    int32_t min_secondaries, max_secondaries; /* U{min..max} secondaries per read */
    double softclip_frac; /* alignments carrying a 50-500 bp soft clip */
    double hardclip_frac; /* of the clipped ones, fraction turned into hard clips */
    int32_t shuffle_records; /* 1: primary at a random place in the group */
    int32_t inverted_paralogs; /* 1: odd paralog sets are stored reverse-complemented */
    double n_base_frac;   /* fraction of assembly bases replaced by N (edge-case tests) */
    double snv_rate, indel_rate, paralog_snv_rate; /* 1/5000, 1/50000, 0.01 */
    int32_t tag_mode;     /* 0: cs only, 1: MD only, 2: both */
    int32_t reserved;
} spx_synth_cfg;

typedef struct spx_synth_genome spx_synth_genome;
typedef struct spx_synth_reads spx_synth_reads;

void spx_synth_default_cfg(spx_synth_cfg *cfg, int platform);
spx_synth_genome *spx_synth_genome_create(const spx_synth_cfg *cfg);
const spx_ref *spx_synth_genome_ref(const spx_synth_genome *g);
void spx_synth_genome_free(spx_synth_genome *g);

/* groups [first, first+n): each group has its own RNG sub-stream, so any
 * range can be generated independently (sharding across ranks). */
spx_synth_reads *spx_synth_reads_create(const spx_synth_genome *g, const spx_synth_cfg *cfg, int64_t first, int32_t n);
const spx_batch *spx_synth_reads_batch(const spx_synth_reads *r);
void spx_synth_reads_free(spx_synth_reads *r);

/* bench/test-side writers (spx_bamwrite.c): the batches, in order, as one name-grouped BAM (htslib block policy,
 * zlib `level`, compressed on `threads` threads; contig_order: BAM target order as a permutation of the contig
 * indices or NULL); returns the file size or -1.  FASTA with `width` bases per line. */
int64_t spx_synth_write_bam(const char *path, const spx_batch *const *batches, int32_t n_batches, const spx_ref *ref,
                            const int32_t *contig_order, int threads, int level, int extra_tags);
int spx_synth_write_fasta(const char *path, const spx_ref *ref, int width);

#ifdef __cplusplus
}
#endif
#endif
