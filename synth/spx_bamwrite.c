/*
 * spx_bamwrite.c -- test/bench-side writers: record batches -> a name-grouped BAM file, assembly -> FASTA.
 *
 * Not part of the scoring path.  The BAM is what samtools/htslib would write for these records: BGZF blocks of at
 * most 0xff00 payload bytes, a record that does not fit into the rest of the current block starts a new one
 * (bgzf_flush_try), records larger than a block are split; raw deflate (zlib) + crc32 + isize, the 28-byte EOF block.
 * Batches are serialised and compressed on threads (one batch per task, written in order; every batch ends its last
 * block, like a bgzf_flush between writers).  Record layout = tests/bamio.py:write_bam (mapq 60, bin 4680, tags
 * NM:i tp:A [cs:Z | MD:Z] zz:B:s), so the two writers are interchangeable in the tests.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "spx_synth.h"

#define BGZF_PAYLOAD 0xff00

typedef struct {
    uint8_t *p;
    size_t n, cap;
} buf_t;

static int buf_need(buf_t *b, size_t extra)
{
    if (b->n + extra <= b->cap) return 0;
    size_t c = b->cap ? b->cap : (1u << 20);
    while (c < b->n + extra) c += c / 2;
    uint8_t *q = (uint8_t *)realloc(b->p, c);
    if (!q) return -1;
    b->p = q;
    b->cap = c;
    return 0;
}
static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

/* one BGZF block holding src[0..n) appended to out */
static int bgzf_block(buf_t *out, const uint8_t *src, size_t n, int level, z_stream *zs)
{
    if (buf_need(out, 18 + compressBound((uLong)n) + 64 + 8)) return -1;
    uint8_t *h = out->p + out->n;
    static const uint8_t magic[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0};
    memcpy(h, magic, 16);
    if (deflateReset(zs) != Z_OK) return -1;
    zs->next_in = (Bytef *)src;
    zs->avail_in = (uInt)n;
    zs->next_out = h + 18;
    zs->avail_out = (uInt)(out->cap - out->n - 18 - 8);
    if (deflate(zs, Z_FINISH) != Z_STREAM_END) return -1;
    const size_t clen = (size_t)(zs->next_out - (h + 18));
    if (18 + clen + 8 > 65536) return -2; /* incompressible beyond the block limit: cannot happen at 0xff00 payload */
    put16(h + 16, (uint32_t)(18 + clen + 8 - 1));
    put32(h + 18 + clen, (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)n));
    put32(h + 18 + clen + 4, (uint32_t)n);
    out->n += 18 + clen + 8;
    (void)level;
    return 0;
}

/* payload bytes -> blocks; `cuts` are record boundaries (offsets of record starts, ascending, ending with n) */
static int bgzf_stream(buf_t *out, const uint8_t *src, size_t n, const size_t *cuts, size_t n_cuts, int level)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return -1;
    size_t start = 0, k = 0; /* block under construction = [start, end) */
    int rc = 0;
    while (start < n && rc == 0) {
        size_t end = start;
        /* whole records while they fit */
        while (k < n_cuts && cuts[k] <= start) ++k;
        while (k < n_cuts && cuts[k] - start <= BGZF_PAYLOAD) end = cuts[k++];
        if (end == start) { /* the next record alone exceeds a block: split it */
            end = start + BGZF_PAYLOAD < n ? start + BGZF_PAYLOAD : n;
            if (k < n_cuts && cuts[k] < end) end = cuts[k];
        }
        rc = bgzf_block(out, src + start, end - start, level, &zs);
        start = end;
    }
    deflateEnd(&zs);
    return rc;
}

typedef struct {
    const spx_batch *const *batches;
    int32_t n_batches;
    const int32_t *tid_of; /* contig index -> BAM target id */
    int level, extra_tags;
    /* scheduling */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int32_t next, written, window;
    buf_t *done;   /* [n_batches] compressed bytes */
    char *ready;   /* [n_batches] */
    int failed;
} job_t;

static int serialise_batch(const spx_batch *b, const int32_t *tid_of, int extra_tags, buf_t *raw, size_t **cuts_out, size_t *n_cuts_out)
{
    size_t n_cuts = 0, cap_cuts = (size_t)b->n_alns + 2;
    size_t *cuts = (size_t *)malloc(cap_cuts * sizeof(size_t));
    if (!cuts) return -1;
    for (int32_t g = 0; g < b->n_groups; ++g) {
        const char *qn = b->qnames + b->qname_off[g];
        const size_t lqn = strlen(qn) + 1;
        for (int32_t a = b->grp_first[g]; a < b->grp_first[g + 1]; ++a) {
            const int32_t lq = b->l_qseq[a], nc = b->n_cigar[a];
            const char *cs = b->cs_off[a] >= 0 ? b->cs + b->cs_off[a] : NULL;
            const char *md = (b->md_off && b->md && b->md_off[a] >= 0) ? b->md + b->md_off[a] : NULL;
            const size_t lcs = cs ? strlen(cs) + 1 : 0, lmd = md ? strlen(md) + 1 : 0;
            size_t aux = (cs ? 3 + lcs : 0) + (md ? 3 + lmd : 0) + (extra_tags ? (3 + 4) + (3 + 1) + (3 + 1 + 4 + 4) : 0);
            const size_t bs = 32 + lqn + 4 * (size_t)nc + ((size_t)lq + 1) / 2 + (size_t)lq + aux;
            if (buf_need(raw, bs + 4)) { free(cuts); return -1; }
            cuts[n_cuts++] = raw->n;
            uint8_t *p = raw->p + raw->n;
            put32(p, (uint32_t)bs);
            p += 4;
            put32(p, (uint32_t)(b->tid[a] >= 0 ? tid_of[b->tid[a]] : -1));
            put32(p + 4, (uint32_t)b->pos[a]);
            p[8] = (uint8_t)lqn; p[9] = 60;
            put16(p + 10, 4680);
            put16(p + 12, (uint32_t)nc);
            put16(p + 14, b->flag[a]);
            put32(p + 16, (uint32_t)lq);
            put32(p + 20, (uint32_t)-1); put32(p + 24, (uint32_t)-1); put32(p + 28, 0);
            p += 32;
            memcpy(p, qn, lqn); p += lqn;
            for (int32_t c = 0; c < nc; ++c, p += 4) put32(p, b->cigar[b->cigar_off[a] + c]);
            memcpy(p, b->seq4 + b->seq_off[a], ((size_t)lq + 1) / 2); p += ((size_t)lq + 1) / 2;
            memcpy(p, b->qual + b->qual_off[a], (size_t)lq); p += lq;
            if (extra_tags) {
                memcpy(p, "NMi", 3); put32(p + 3, 3); p += 7;
                memcpy(p, "tpAP", 4); p += 4;
            }
            if (cs) { memcpy(p, "csZ", 3); memcpy(p + 3, cs, lcs); p += 3 + lcs; }
            if (md) { memcpy(p, "MDZ", 3); memcpy(p + 3, md, lmd); p += 3 + lmd; }
            if (extra_tags) {
                memcpy(p, "zzBs", 4); put32(p + 4, 2); put16(p + 8, (uint32_t)(uint16_t)-1); put16(p + 10, 7); p += 12;
            }
            raw->n += bs + 4;
        }
    }
    cuts[n_cuts++] = raw->n;
    *cuts_out = cuts;
    *n_cuts_out = n_cuts;
    return 0;
}

static void *bam_worker(void *arg)
{
    job_t *J = (job_t *)arg;
    buf_t raw = {0, 0, 0};
    for (;;) {
        pthread_mutex_lock(&J->mu);
        while (!J->failed && J->next < J->n_batches && J->next >= J->written + J->window) pthread_cond_wait(&J->cv, &J->mu);
        if (J->failed || J->next >= J->n_batches) { pthread_mutex_unlock(&J->mu); break; }
        const int32_t k = J->next++;
        pthread_mutex_unlock(&J->mu);
        raw.n = 0;
        size_t *cuts = NULL, n_cuts = 0;
        buf_t out = {0, 0, 0};
        int rc = serialise_batch(J->batches[k], J->tid_of, J->extra_tags, &raw, &cuts, &n_cuts);
        if (!rc) rc = bgzf_stream(&out, raw.p, raw.n, cuts, n_cuts, J->level);
        free(cuts);
        pthread_mutex_lock(&J->mu);
        if (rc) J->failed = 1;
        J->done[k] = out;
        J->ready[k] = 1;
        pthread_cond_broadcast(&J->cv);
        pthread_mutex_unlock(&J->mu);
    }
    free(raw.p);
    return NULL;
}

int64_t spx_synth_write_bam(const char *path, const spx_batch *const *batches, int32_t n_batches, const spx_ref *ref,
                            const int32_t *contig_order, int threads, int level, int extra_tags)
{
    if (!path || !batches || n_batches < 0 || !ref) return -1;
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    const int32_t nc = ref->n_contigs;
    int32_t *tid_of = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nc + 1));
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nc + 1));
    for (int32_t i = 0; i < nc; ++i) order[i] = contig_order ? contig_order[i] : i;
    for (int32_t i = 0; i < nc; ++i) tid_of[order[i]] = i;
    int64_t total = 0;
    int rc = 0;
    { /* header */
        buf_t h = {0, 0, 0}, out = {0, 0, 0};
        const char *hd = "@HD\tVN:1.6\tSO:queryname\n";
        if (buf_need(&h, 4096) || !h.p) { fclose(fp); free(tid_of); free(order); return -1; }
        memcpy(h.p, "BAM\1", 4);
        h.n = 8;
        memcpy(h.p + h.n, hd, strlen(hd));
        h.n += strlen(hd);
        for (int32_t i = 0; i < nc; ++i) {
            const int32_t c = order[i];
            const char *nm = ref->names + ref->name_off[c];
            buf_need(&h, strlen(nm) + 64);
            h.n += (size_t)sprintf((char *)h.p + h.n, "@SQ\tSN:%s\tLN:%lld\n", nm, (long long)(ref->seq_off[c + 1] - ref->seq_off[c]));
        }
        put32(h.p + 4, (uint32_t)(h.n - 8));
        buf_need(&h, 4);
        put32(h.p + h.n, (uint32_t)nc);
        h.n += 4;
        for (int32_t i = 0; i < nc; ++i) {
            const int32_t c = order[i];
            const char *nm = ref->names + ref->name_off[c];
            const size_t ln = strlen(nm) + 1;
            buf_need(&h, ln + 8);
            put32(h.p + h.n, (uint32_t)ln);
            memcpy(h.p + h.n + 4, nm, ln);
            put32(h.p + h.n + 4 + ln, (uint32_t)(ref->seq_off[c + 1] - ref->seq_off[c]));
            h.n += 8 + ln;
        }
        rc = bgzf_stream(&out, h.p, h.n, NULL, 0, level);
        if (!rc && fwrite(out.p, 1, out.n, fp) != out.n) rc = -1;
        total += (int64_t)out.n;
        free(h.p);
        free(out.p);
    }
    if (!rc && n_batches > 0) {
        job_t J;
        memset(&J, 0, sizeof J);
        J.batches = batches; J.n_batches = n_batches; J.tid_of = tid_of; J.level = level; J.extra_tags = extra_tags;
        if (threads < 1) threads = 1;
        if (threads > n_batches) threads = n_batches;
        J.window = 2 * threads;
        J.done = (buf_t *)calloc((size_t)n_batches, sizeof(buf_t));
        J.ready = (char *)calloc((size_t)n_batches, 1);
        pthread_mutex_init(&J.mu, NULL);
        pthread_cond_init(&J.cv, NULL);
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
        for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, bam_worker, &J);
        for (int32_t k = 0; k < n_batches; ++k) {
            pthread_mutex_lock(&J.mu);
            while (!J.ready[k]) pthread_cond_wait(&J.cv, &J.mu);
            buf_t out = J.done[k];
            pthread_mutex_unlock(&J.mu);
            if (!rc && !J.failed && out.n && fwrite(out.p, 1, out.n, fp) != out.n) rc = -1;
            total += (int64_t)out.n;
            free(out.p);
            pthread_mutex_lock(&J.mu);
            J.written = k + 1;
            if (rc) J.failed = 1;
            pthread_cond_broadcast(&J.cv);
            pthread_mutex_unlock(&J.mu);
        }
        for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
        if (J.failed) rc = -1;
        free(th);
        free(J.done);
        free(J.ready);
        pthread_mutex_destroy(&J.mu);
        pthread_cond_destroy(&J.cv);
    }
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!rc && fwrite(eof_block, 1, 28, fp) != 28) rc = -1;
    total += 28;
    if (fclose(fp) != 0) rc = -1;
    free(tid_of);
    free(order);
    return rc ? -1 : total;
}

int spx_synth_write_fasta(const char *path, const spx_ref *ref, int width)
{
    if (!path || !ref) return -1;
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    if (width < 1) width = 80;
    size_t cap = (size_t)1 << 22;
    char *line = (char *)malloc(cap + (size_t)width + 8);
    for (int32_t c = 0; c < ref->n_contigs; ++c) {
        fprintf(fp, ">%s synthetic\n", ref->names + ref->name_off[c]);
        const char *s = ref->bases + ref->seq_off[c];
        const int64_t n = ref->seq_off[c + 1] - ref->seq_off[c];
        size_t fill = 0;
        for (int64_t o = 0; o < n; o += width) {
            const size_t w = (size_t)(n - o < width ? n - o : width);
            memcpy(line + fill, s + o, w);
            fill += w;
            line[fill++] = '\n';
            if (fill >= cap) { fwrite(line, 1, fill, fp); fill = 0; }
        }
        if (fill) fwrite(line, 1, fill, fp);
    }
    free(line);
    return fclose(fp) == 0 ? 0 : -1;
}
