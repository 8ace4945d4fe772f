/*
 * spx_io.cpp -- input side of the drop-in: name-grouped BAM -> spx_batch blocks, FASTA -> spx_ref.
 *
 * The reference reads through htslib (sam_open/sam_read1 at src/secphase.c:236-268, fai_load/fai_fetch at
 * src/secphase.c:101, submodules/ptMarker/ptMarker.c:739-744).  htslib is not available in this environment,
 * so this is a small self-contained reader on zlib: BGZF blocks are inflated in parallel, BAM records are
 * copied field by field into the flat record format (SEQ and CIGAR keep BAM's own packing).
 * Grouping follows src/secphase.c:273-279: consecutive records with the same read name form a group.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <future>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spx.h"

namespace {

thread_local std::string g_io_err;

struct Block { size_t coff, clen, uoff, ulen; };

/* byte buffer that grows WITHOUT zero-filling (a batch inflates to ~1 GB: value-initialising that much memory
 * on one thread costs more than inflating it on 64) */
struct RawBuf {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { free(p); }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
    void reserve(size_t c)
    {
        if (c <= cap) return;
        if (c >= ((size_t)32 << 20)) {
            /* batch-sized buffers: 2 MB alignment + MADV_HUGEPAGE, so that the inflate threads' first touch of a
             * gigabyte costs a few hundred page faults instead of a quarter of a million (where the kernel offers
             * transparent huge pages on request; harmless otherwise) */
            const size_t al = (size_t)2 << 20, want = (c + al - 1) & ~(al - 1);
            void *q = nullptr;
            if (posix_memalign(&q, al, want) != 0 || !q) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
            (void)madvise(q, want, MADV_HUGEPAGE);
#endif
            if (n) memcpy(q, p, n);
            free(p);
            p = (uint8_t *)q; cap = want;
            return;
        }
        uint8_t *q = (uint8_t *)realloc(p, c);
        if (!q) throw std::bad_alloc();
        p = q; cap = c;
    }
    void resize(size_t m) /* new bytes are NOT initialised */
    {
        if (m > cap) reserve(std::max(m, cap + cap / 2));
        n = m;
    }
    void clear() { n = 0; }
    void swap(RawBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
    void assign(const uint8_t *b, const uint8_t *e) { resize((size_t)(e - b)); if (e > b) memmove(p, b, (size_t)(e - b)); }
    void erase_front(size_t k) { if (k >= n) { n = 0; return; } memmove(p, p + k, n - k); n -= k; }
};

/* one batch under construction / handed out: SEQ, QUAL and cs are NOT copied, they are offsets into the
 * inflated byte stream `ubuf` (only CIGAR is copied, it needs 4-byte alignment) */
struct Slot {
    RawBuf ubuf;
    std::vector<int32_t> grp_first, tid, pos, l_qseq, n_cigar;
    std::vector<int64_t> qname_off, cigar_off, seq_off, qual_off, cs_off, md_off;
    std::vector<int64_t> rec_off; /* first byte (after block_size) of each raw BAM record in ubuf */
    std::vector<uint16_t> flag;
    std::vector<uint32_t> cigar;
    std::vector<char> qnames;
    spx_batch view;
    int ng = 0;
    int rc = 0;
    void clear()
    {
        grp_first.clear(); tid.clear(); pos.clear(); l_qseq.clear(); n_cigar.clear(); qname_off.clear(); cigar_off.clear();
        seq_off.clear(); qual_off.clear(); cs_off.clear(); md_off.clear(); flag.clear(); cigar.clear(); qnames.clear(); ng = 0; rc = 0;
        rec_off.clear();
    }
};

struct CChunk { /* compressed blocks of one chunk */
    std::vector<uint8_t> cbuf;
    std::vector<Block> blocks;
    bool eof = false;
    std::string err;
};

struct Reader {
    FILE *fp = nullptr;
    CChunk cchunk[2];              /* the chunk being inflated and the one a helper thread reads ahead */
    int ccur = 0;
    std::future<void> ahead;
    bool have_ahead = false;
    RawBuf leftover; /* inflated bytes after the last complete group of the previous batch */
    bool eof = false;
    int threads = 4;
    /* header */
    std::vector<std::string> tname;
    std::vector<int64_t> tlen;
    std::vector<int32_t> tmap; /* BAM tid -> contig index of the reference handed to the scorer (-1 unknown) */
    std::string header_text;
    /* ring of batches: the caller may keep the last SPX_BAM_SLOTS - 2 batches alive (a pipelined caller has several in
     * flight), one more is being read ahead */
    static const int NSLOT = 8;
    Slot slot[NSLOT];
    int cur = 0;
    std::future<void> pending; /* the NEXT batch is read while the caller works on the current one */
    bool have_pending = false;
    int32_t pending_max = 0;
    std::string last_name;
    bool have_last = false;
    int64_t n_records = 0, n_groups_total = 0;
    double t_inflate = 0;
    size_t last_batch_bytes = 0;
};

/* read up to 1024 BGZF blocks (~64 MB inflated), inflate them in parallel, append to `ubuf`; false at EOF / error */
bool read_chunk_impl(Reader *r, RawBuf &ubuf);
bool read_chunk(Reader *r, RawBuf &ubuf)
{
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    const bool ok = read_chunk_impl(r, ubuf);
    clock_gettime(CLOCK_MONOTONIC, &b);
    r->t_inflate += (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
    return ok;
}
/* libdeflate inflates BGZF blocks 2-3x faster than zlib.  The image ships its runtime library without headers, so the
 * three entry points are bound at run time (their C signatures are part of libdeflate's stable ABI); zlib is what
 * runs when the library is not there. */
struct Deflate {
    void *(*alloc)(void) = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    Deflate()
    {
        if (getenv("SPX_NO_LIBDEFLATE")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)(void))dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        if (!alloc || !decompress || !release) alloc = nullptr;
    }
};
static const Deflate &deflate_lib()
{
    static const Deflate d;
    return d;
}

/* compressed side of one chunk: up to kChunkBlocks BGZF blocks read from the file (serial freads).  It is fetched by a
 * helper thread while the previous chunk is being inflated on the worker threads. */
static const int kChunkBlocks = 4096; /* ~256 MB inflated */
static void read_compressed(Reader *r, CChunk &c)
{
    c.cbuf.clear();
    c.blocks.clear();
    c.eof = false;
    c.err.clear();
    size_t utot = 0;
    for (int n = 0; n < kChunkBlocks; ++n) {
        uint8_t hdr[18];
        size_t got = fread(hdr, 1, 18, r->fp);
        if (got == 0) { c.eof = true; break; }
        if (got != 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) { c.err = "not a BGZF block"; c.eof = true; return; }
        const unsigned xlen = hdr[10] | (hdr[11] << 8);
        /* the BC subfield is first in every BAM written by htslib/samtools; general case: scan the extra field */
        std::vector<uint8_t> extra(xlen);
        memcpy(extra.data(), hdr + 12, std::min<size_t>(6, xlen));
        if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, r->fp) != xlen - 6) { c.err = "truncated BGZF header"; c.eof = true; return; }
        int bsize = -1;
        for (size_t o = 0; o + 4 <= xlen;) {
            const unsigned slen = extra[o + 2] | (extra[o + 3] << 8);
            if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2) bsize = extra[o + 4] | (extra[o + 5] << 8);
            o += 4 + slen;
        }
        if (bsize < 0) { c.err = "BGZF block without BC field"; c.eof = true; return; }
        const size_t clen = (size_t)bsize + 1 - 12 - xlen; /* deflate data + crc32 + isize */
        const size_t coff = c.cbuf.size();
        c.cbuf.resize(coff + clen);
        if (fread(c.cbuf.data() + coff, 1, clen, r->fp) != clen || clen < 8) { c.err = "truncated BGZF block"; c.eof = true; return; }
        const uint8_t *t = c.cbuf.data() + coff + clen - 4;
        const size_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((size_t)t[3] << 24);
        c.blocks.push_back({coff, clen - 8, utot, isize});
        utot += isize;
    }
}

bool read_chunk_impl(Reader *r, RawBuf &ubuf)
{
    if (r->eof && !r->have_ahead) return false;
    CChunk *c = &r->cchunk[r->ccur];
    if (r->have_ahead) {
        r->ahead.get(); /* the helper thread has filled cchunk[ccur] */
        r->have_ahead = false;
    } else
        read_compressed(r, *c);
    if (!c->err.empty()) { g_io_err = c->err; r->eof = true; return false; }
    if (c->eof) r->eof = true;
    if (!r->eof) { /* fetch the next chunk's compressed bytes while this one is inflated */
        r->ccur ^= 1;
        CChunk *nx = &r->cchunk[r->ccur];
        r->ahead = std::async(std::launch::async, [r, nx]() { read_compressed(r, *nx); });
        r->have_ahead = true;
    }
    const std::vector<Block> &blocks = c->blocks;
    if (blocks.empty()) return false;
    const size_t ubase = ubuf.size();
    ubuf.resize(ubase + blocks.back().uoff + blocks.back().ulen);
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    const Deflate &DL = deflate_lib();
    auto work = [&]() {
        if (DL.alloc) {
            void *d = DL.alloc();
            if (!d) { bad = 1; return; }
            for (;;) {
                size_t k = next.fetch_add(4);
                if (k >= blocks.size()) break;
                for (size_t q = k; q < std::min(blocks.size(), k + 4); ++q) {
                    const Block &b = blocks[q];
                    if (b.ulen == 0) continue;
                    size_t got = 0;
                    if (DL.decompress(d, c->cbuf.data() + b.coff, b.clen, ubuf.data() + ubase + b.uoff, b.ulen, &got) != 0 || got != b.ulen) bad = 1;
                }
            }
            DL.release(d);
            return;
        }
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
        for (;;) {
            size_t k = next.fetch_add(4);
            if (k >= blocks.size()) break;
            for (size_t q = k; q < std::min(blocks.size(), k + 4); ++q) {
                const Block &b = blocks[q];
                if (b.ulen == 0) continue;
                if (inflateReset(&zs) != Z_OK) { bad = 1; continue; }
                zs.next_in = c->cbuf.data() + b.coff; zs.avail_in = (uInt)b.clen;
                zs.next_out = ubuf.data() + ubase + b.uoff; zs.avail_out = (uInt)b.ulen;
                int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.avail_out != 0) bad = 1;
            }
        }
        inflateEnd(&zs);
    };
    int nt = std::max(1, std::min<int>(r->threads, (int)(blocks.size() + 3) / 4));
    if (nt == 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work);
        for (auto &t : th) t.join();
    }
    if (bad) { g_io_err = "inflate failed"; r->eof = true; return false; }
    return true;
}

/* make sure n bytes are available at `at` in ubuf; false at clean EOF / error */
bool need(Reader *r, RawBuf &ubuf, size_t at, size_t n)
{
    while (ubuf.size() < at + n)
        if (!read_chunk(r, ubuf)) return ubuf.size() >= at + n;
    return true;
}

inline int32_t le32(const uint8_t *p) { return (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)); }

/* find a Z tag (cs or MD) in the aux block */
/* value of a B-array tag (sub-type byte, count, elements) or NULL */
const uint8_t *find_tag_b(const uint8_t *aux, const uint8_t *end, char k0, char k1)
{
    while (aux + 3 <= end) {
        const char t0 = (char)aux[0], t1 = (char)aux[1], ty = (char)aux[2];
        const uint8_t *v = aux + 3;
        size_t len;
        switch (ty) {
        case 'A': case 'c': case 'C': len = 1; break;
        case 's': case 'S': len = 2; break;
        case 'i': case 'I': case 'f': len = 4; break;
        case 'Z': case 'H': {
            const uint8_t *z = v;
            while (z < end && *z) ++z;
            len = (size_t)(z - v) + 1;
            break;
        }
        case 'B': {
            if (v + 5 > end) return nullptr;
            const char sub = (char)v[0];
            const uint32_t cnt = (uint32_t)le32(v + 1);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            len = 5 + es * (size_t)cnt;
            if (t0 == k0 && t1 == k1) return v + len <= end ? v : nullptr;
            break;
        }
        default: return nullptr;
        }
        aux = v + len;
    }
    return nullptr;
}

const char *find_tag(const uint8_t *aux, const uint8_t *end, char k0, char k1)
{
    while (aux + 3 <= end) {
        const char t0 = (char)aux[0], t1 = (char)aux[1], ty = (char)aux[2];
        const uint8_t *v = aux + 3;
        size_t len;
        switch (ty) {
        case 'A': case 'c': case 'C': len = 1; break;
        case 's': case 'S': len = 2; break;
        case 'i': case 'I': case 'f': len = 4; break;
        case 'Z': case 'H': {
            const uint8_t *z = v;
            while (z < end && *z) ++z;
            if (t0 == k0 && t1 == k1 && ty == 'Z') return (const char *)v;
            len = (size_t)(z - v) + 1;
            break;
        }
        case 'B': {
            if (v + 5 > end) return nullptr;
            const char sub = (char)v[0];
            const uint32_t cnt = (uint32_t)le32(v + 1);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            len = 5 + es * cnt;
            break;
        }
        default: return nullptr;
        }
        aux = v + len;
    }
    return nullptr;
}

} // namespace

struct spx_bam_reader { Reader r; };
struct spx_fasta {
    std::vector<int64_t> name_off, seq_off;
    std::vector<char> names, bases;
    spx_ref ref;
};

extern "C" const char *spx_io_last_error(void) { return g_io_err.c_str(); }

extern "C" int spx_bam_open(const char *path, int threads, spx_bam_reader **out)
{
    if (!path || !out) return SPX_EINVAL;
    *out = nullptr;
    FILE *fp = fopen(path, "rb");
    if (!fp) { g_io_err = std::string("cannot open ") + path; return SPX_EINVAL; }
    spx_bam_reader *h = new spx_bam_reader();
    Reader *r = &h->r;
    r->fp = fp;
    r->threads = threads > 0 ? threads : 4;
    RawBuf &u = r->leftover;
    size_t at = 0;
    auto bail = [&](const char *msg) { g_io_err = msg; fclose(fp); delete h; return SPX_EINVAL; };
    if (!need(r, u, at, 12) || memcmp(u.data(), "BAM\1", 4) != 0) return bail("not a BAM file");
    const int32_t l_text = le32(u.data() + 4);
    at = 8;
    if (!need(r, u, at, (size_t)l_text + 4)) return bail("truncated BAM header");
    r->header_text.assign((const char *)u.data() + at, strnlen((const char *)u.data() + at, (size_t)l_text));
    at += (size_t)l_text;
    const int32_t n_ref = le32(u.data() + at);
    at += 4;
    for (int32_t i = 0; i < n_ref; ++i) {
        if (!need(r, u, at, 4)) return bail("truncated BAM header");
        const int32_t ln = le32(u.data() + at);
        at += 4;
        if (!need(r, u, at, (size_t)ln + 4)) return bail("truncated BAM header");
        r->tname.emplace_back((const char *)u.data() + at);
        at += (size_t)ln;
        r->tlen.push_back(le32(u.data() + at));
        at += 4;
    }
    u.erase_front(at);
    r->tmap.assign(n_ref, -1);
    for (int32_t i = 0; i < n_ref; ++i) r->tmap[i] = i;
    *out = h;
    return SPX_OK;
}

extern "C" int32_t spx_bam_n_targets(const spx_bam_reader *h) { return h ? (int32_t)h->r.tname.size() : 0; }
extern "C" const char *spx_bam_target_name(const spx_bam_reader *h, int32_t i)
{
    return (h && i >= 0 && (size_t)i < h->r.tname.size()) ? h->r.tname[i].c_str() : nullptr;
}

/* BAM target ids -> contig indices of `ref` (by name); alignments on contigs the FASTA lacks get tid -1
 * and their group is rejected by the scorer with SPX_EINVAL */
extern "C" int spx_bam_bind_reference(spx_bam_reader *h, const spx_ref *ref)
{
    if (!h || !ref) return SPX_EINVAL;
    Reader *r = &h->r;
    int missing = 0;
    for (size_t i = 0; i < r->tname.size(); ++i) {
        r->tmap[i] = -1;
        for (int32_t c = 0; c < ref->n_contigs; ++c)
            if (r->tname[i] == ref->names + ref->name_off[c]) { r->tmap[i] = c; break; }
        if (r->tmap[i] < 0) ++missing;
    }
    return missing;
}

/* fill one slot with up to max_groups complete name groups */
static double io_now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* f(k0, k1) over [0, n) on up to `threads` threads */
template <class F>
static void io_parallel(int64_t n, int threads, F f)
{
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(threads, 16), n / 2048));
    if (T <= 1) { f((int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] { f(n * t / T, n * (t + 1) / T); });
    for (auto &x : th) x.join();
}

static void fill_slot(Reader *r, Slot &S, int32_t max_groups)
{
    const double t_fill0 = io_now();
    S.clear();
    S.ubuf.swap(r->leftover);
    r->leftover.clear();
    RawBuf &u = S.ubuf;
    u.reserve(r->last_batch_bytes + r->last_batch_bytes / 8 + (320u << 20)); /* one allocation per batch, not a doubling chain */
    size_t at = 0;
    bool open_group = false;
    /* Pass 1, serial (every record says where the next one starts, and the batch ends on a name change): record
     * offsets, field-length validation, group boundaries.  One cache line per record.  The fields themselves, the tag
     * search (a walk over the aux bytes of every record) and the copies are pass 2, on threads. */
    std::vector<int64_t> rec_at;     /* offset of the record body (behind block_size) */
    std::vector<uint8_t> rec_new;    /* the record opens a group */
    for (;;) {
        if (!need(r, u, at, 4)) break;
        const int32_t bs = le32(u.data() + at);
        if (bs < 32) { g_io_err = "corrupt BAM record"; S.rc = SPX_EINVAL; break; }
        if (!need(r, u, at, (size_t)bs + 4)) { g_io_err = "truncated BAM record"; S.rc = SPX_EINVAL; break; }
        const uint8_t *p = u.data() + at + 4;
        const uint32_t l_name = p[8];
        const uint32_t ncig = p[12] | (p[13] << 8);
        const int32_t lseq = le32(p + 16);
        const char *name = (const char *)p + 32;
        /* the fixed part announces the lengths of the variable part: none of them may reach past the record, and the
         * name must be NUL-terminated inside it (strlen / the CIGAR copy below would walk off the buffer otherwise) */
        if (lseq < 0 || l_name < 1 || 32 + (uint64_t)l_name + 4 * (uint64_t)ncig + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq > (uint64_t)bs ||
            name[l_name - 1] != 0) {
            g_io_err = "corrupt BAM record (field lengths exceed the record)";
            S.rc = SPX_EINVAL;
            break;
        }
        /* group boundary on a name change (src/secphase.c:273-279) */
        const bool same = r->have_last && r->last_name == name;
        bool opens = false;
        if (!same || !open_group) {
            if (open_group && S.ng == max_groups) break; /* this record opens the next batch */
            r->last_name = name;
            r->have_last = true;
            open_group = true;
            opens = true;
            ++S.ng;
        }
        rec_at.push_back((int64_t)(at + 4));
        rec_new.push_back(opens ? 1 : 0);
        at += (size_t)bs + 4;
        r->n_records++;
    }
    const double t_p1 = io_now();
    const int64_t nrec = (int64_t)rec_at.size();
    S.flag.resize((size_t)nrec); S.tid.resize((size_t)nrec); S.pos.resize((size_t)nrec); S.l_qseq.resize((size_t)nrec);
    S.n_cigar.resize((size_t)nrec); S.cigar_off.resize((size_t)nrec); S.seq_off.resize((size_t)nrec); S.qual_off.resize((size_t)nrec);
    S.rec_off.resize((size_t)nrec); S.cs_off.resize((size_t)nrec); S.md_off.resize((size_t)nrec);
    std::vector<int64_t> cg_at((size_t)nrec, -1); /* CG:B,I payload when the real CIGAR lives in the tag */
    const uint8_t *base = u.data();
    io_parallel(nrec, r->threads, [&](int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; ++k) {
            const uint8_t *p = base + rec_at[(size_t)k];
            const int32_t bs = le32(p - 4);
            const int32_t refid = le32(p), posv = le32(p + 4);
            const uint32_t l_name = p[8];
            const uint32_t ncig = p[12] | (p[13] << 8), flg = p[14] | (p[15] << 8);
            const int32_t lseq = le32(p + 16);
            const uint8_t *cig = p + 32 + l_name, *sq = cig + 4 * (size_t)ncig, *ql = sq + ((size_t)lseq + 1) / 2, *aux = ql + lseq,
                          *end = p + bs;
            S.flag[(size_t)k] = (uint16_t)flg;
            S.tid[(size_t)k] = (refid >= 0 && (size_t)refid < r->tmap.size()) ? r->tmap[refid] : -1;
            S.pos[(size_t)k] = posv;
            S.l_qseq[(size_t)k] = lseq;
            /* more than 65535 CIGAR operations: the record carries the placeholder <l_seq>S<ref_len>N and the real CIGAR
             * in the CG:B,I tag; sam_read1 puts it back before secphase sees the record, so do we */
            uint32_t cg_n = 0;
            if (ncig == 2 && aux <= end && ((uint32_t)le32(cig) & 0xf) == SPX_CSOFT_CLIP && ((uint32_t)le32(cig) >> 4) == (uint32_t)lseq &&
                ((uint32_t)le32(cig + 4) & 0xf) == SPX_CREF_SKIP) {
                const uint8_t *b = find_tag_b(aux, end, 'C', 'G');
                if (b && (b[0] == 'I' || b[0] == 'i')) {
                    cg_n = (uint32_t)le32(b + 1);
                    if (cg_n > 0 && b + 5 + 4 * (size_t)cg_n <= end) cg_at[(size_t)k] = (int64_t)(b + 5 - base); else cg_n = 0;
                }
            }
            S.n_cigar[(size_t)k] = (int32_t)(cg_at[(size_t)k] >= 0 ? cg_n : ncig);
            S.seq_off[(size_t)k] = (int64_t)(sq - base);
            S.qual_off[(size_t)k] = (int64_t)(ql - base);
            S.rec_off[(size_t)k] = (int64_t)(p - base);
            const char *csz = aux <= end ? find_tag(aux, end, 'c', 's') : nullptr;
            S.cs_off[(size_t)k] = csz ? (int64_t)((const uint8_t *)csz - base) : -1;
            const char *mdz = (!csz && aux <= end) ? find_tag(aux, end, 'M', 'D') : nullptr; /* only looked at without cs */
            S.md_off[(size_t)k] = mdz ? (int64_t)((const uint8_t *)mdz - base) : -1;
        }
    });
    const double t_p2 = io_now();
    /* offsets of the copied parts (CIGAR words, group names), then the copies */
    int64_t cw = 0, nb = 0;
    std::vector<int64_t> grp_rec; /* first record of every group */
    for (int64_t k = 0; k < nrec; ++k) {
        S.cigar_off[(size_t)k] = cw;
        cw += S.n_cigar[(size_t)k];
        if (rec_new[(size_t)k]) {
            S.grp_first.push_back((int32_t)k);
            S.qname_off.push_back(nb);
            grp_rec.push_back(k);
            nb += (int64_t)(base + rec_at[(size_t)k])[8]; /* l_read_name counts the NUL */
        }
    }
    S.cigar.resize((size_t)cw);
    S.qnames.resize((size_t)nb);
    io_parallel(nrec, r->threads, [&](int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; ++k) {
            const uint8_t *p = base + rec_at[(size_t)k];
            const uint8_t *src = cg_at[(size_t)k] >= 0 ? base + cg_at[(size_t)k] : p + 32 + p[8];
            uint32_t *dst = S.cigar.data() + S.cigar_off[(size_t)k];
            for (int32_t c = 0; c < S.n_cigar[(size_t)k]; ++c) dst[c] = (uint32_t)le32(src + 4 * (size_t)c);
        }
    });
    io_parallel((int64_t)grp_rec.size(), r->threads, [&](int64_t g0, int64_t g1) {
        for (int64_t g = g0; g < g1; ++g) {
            const uint8_t *p = base + rec_at[(size_t)grp_rec[(size_t)g]];
            memcpy(S.qnames.data() + S.qname_off[(size_t)g], p + 32, (size_t)p[8]);
        }
    });
    const double t_p3 = io_now();
    /* bytes of the records that belong to the next batch */
    r->leftover.assign(u.data() + at, u.data() + u.size());
    r->last_batch_bytes = at;
    S.grp_first.push_back((int32_t)S.flag.size());
    S.qnames.push_back(0);
    S.cigar.push_back(0);
    u.resize(at + 8); /* keep the consumed part (+ slack), drop the tail that was copied out */
    memset(u.data() + at, 0, 8);
    spx_batch &b = S.view;
    b.n_groups = S.ng;
    b.n_alns = (int32_t)S.flag.size();
    b.grp_first = S.grp_first.data(); b.qname_off = S.qname_off.data(); b.qnames = S.qnames.data();
    b.flag = S.flag.data(); b.tid = S.tid.data(); b.pos = S.pos.data(); b.l_qseq = S.l_qseq.data();
    b.n_cigar = S.n_cigar.data(); b.cigar_off = S.cigar_off.data(); b.seq_off = S.seq_off.data();
    b.qual_off = S.qual_off.data(); b.cs_off = S.cs_off.data(); b.cigar = S.cigar.data();
    b.seq4 = u.data(); b.qual = u.data(); b.cs = (const char *)u.data();
    b.md_off = S.md_off.data(); b.md = (const char *)u.data();
    r->n_groups_total += S.ng;
    if (getenv("SPX_TIMING"))
        fprintf(stderr, "[spx timing] BAM batch: %d groups, %.1f MB inflated, %.3f s (record chain incl. read+inflate %.3f [read+inflate %.3f], "
                        "fields+tags %.3f, CIGAR+names %.3f, tail %.3f)\n", S.ng, u.size() / 1e6, io_now() - t_fill0, t_p1 - t_fill0, r->t_inflate,
                t_p2 - t_p1, t_p3 - t_p2, io_now() - t_p3);
    r->t_inflate = 0;
}

/* up to max_groups complete name groups; the batch stays valid until the next call.  The following batch is
 * read and inflated in the background while the caller works on this one.  Returns the number of groups
 * (0 at end of file) or SPX_E*. */
extern "C" int spx_bam_next_batch(spx_bam_reader *h, int32_t max_groups, const spx_batch **out)
{
    if (!h || !out || max_groups <= 0) return SPX_EINVAL;
    Reader *r = &h->r;
    if (r->have_pending && r->pending_max == max_groups) {
        r->pending.get();
        r->cur = (r->cur + 1) % Reader::NSLOT; /* the slot the background task filled */
    } else {
        if (r->have_pending) r->pending.get(); /* different batch size requested: cannot happen in the CLI */
        fill_slot(r, r->slot[r->cur], max_groups);
    }
    r->have_pending = false;
    Slot &S = r->slot[r->cur];
    *out = &S.view;
    if (S.rc < 0) return S.rc;
    if (S.ng > 0) {
        Slot *nxt = &r->slot[(r->cur + 1) % Reader::NSLOT];
        r->pending_max = max_groups;
        r->pending = std::async(std::launch::async, [r, nxt, max_groups]() { fill_slot(r, *nxt, max_groups); });
        r->have_pending = true;
    }
    return S.ng;
}

extern "C" void spx_bam_close(spx_bam_reader *h)
{
    if (!h) return;
    if (h->r.have_pending) h->r.pending.get();
    if (h->r.have_ahead) h->r.ahead.get();
    if (h->r.fp) fclose(h->r.fp);
    delete h;
}

/* ---- -w/--writeBam: the reference opens `<prefix>.quality_modified.out.bam` with sam_open(path, "w")
 * (src/secphase.c:643-652), which in htslib means SAM TEXT, writes the input header and then, per dispatched
 * group, every alignment with its BAQ-modified qualities through sam_write1 (src/secphase.c:182-189).  This is
 * the same text: header, then one SAM line per record formatted from the raw BAM record. ---- */
struct spx_sam_writer { FILE *fp = nullptr; std::string line; };

extern "C" int spx_sam_open(const char *path, const spx_bam_reader *src, spx_sam_writer **out)
{
    if (!path || !src || !out) return SPX_EINVAL;
    *out = nullptr;
    FILE *fp = fopen(path, "wb");
    if (!fp) { g_io_err = std::string("cannot create ") + path; return SPX_EINVAL; }
    const Reader &r = src->r;
    const std::string &t = r.header_text;
    if (!t.empty()) {
        fwrite(t.data(), 1, t.size(), fp);
        if (t.back() != '\n') fputc('\n', fp);
    }
    /* a header without @SQ lines gets them from the target list, as sam_hdr_write does */
    bool has_sq = t.compare(0, 4, "@SQ\t") == 0 || t.find("\n@SQ\t") != std::string::npos;
    if (!has_sq)
        for (size_t i = 0; i < r.tname.size(); ++i) fprintf(fp, "@SQ\tSN:%s\tLN:%lld\n", r.tname[i].c_str(), (long long)r.tlen[i]);
    spx_sam_writer *w = new spx_sam_writer();
    w->fp = fp;
    *out = w;
    return SPX_OK;
}

static void put_int(std::string &s, long long v)
{
    char b[24];
    s.append(b, (size_t)snprintf(b, sizeof b, "%lld", v));
}
static void put_g(std::string &s, double v)
{
    char b[40];
    s.append(b, (size_t)snprintf(b, sizeof b, "%g", v));
}

/* one SAM line (sam_format1): the eleven mandatory fields, then the aux block */
static bool format_sam(const Reader &r, const uint8_t *p, int32_t bs, const uint8_t *qual, std::string &s)
{
    const int32_t refid = le32(p), posv = le32(p + 4);
    const uint32_t l_name = p[8], mapq = p[9];
    const uint32_t ncig = p[12] | (p[13] << 8), flg = p[14] | (p[15] << 8);
    const int32_t lseq = le32(p + 16), mtid = le32(p + 20), mpos = le32(p + 24), tlen = le32(p + 28);
    const uint8_t *cig = p + 32 + l_name, *sq = cig + 4 * (size_t)ncig, *ql = sq + ((size_t)lseq + 1) / 2, *aux = ql + lseq,
                  *end = p + bs;
    if (aux > end) return false;
    auto tname = [&](int32_t t) -> const char * { return (t >= 0 && (size_t)t < r.tname.size()) ? r.tname[t].c_str() : "*"; };
    s.clear();
    s += (const char *)p + 32; s += '\t';
    put_int(s, flg); s += '\t';
    s += tname(refid); s += '\t';
    put_int(s, (long long)posv + 1); s += '\t';
    put_int(s, mapq); s += '\t';
    if (ncig == 0) s += '*';
    for (uint32_t k = 0; k < ncig; ++k) {
        const uint32_t c = (uint32_t)le32(cig + 4 * k);
        put_int(s, c >> 4);
        s += "MIDNSHP=XB??????"[c & 15];
    }
    s += '\t';
    if (mtid < 0) s += '*';
    else if (mtid == refid) s += '=';
    else s += tname(mtid);
    s += '\t';
    put_int(s, (long long)mpos + 1); s += '\t';
    put_int(s, tlen); s += '\t';
    if (lseq == 0) s += "*\t*";
    else {
        for (int32_t k = 0; k < lseq; ++k) s += "=ACMGRSVTWYHKDBN"[(sq[k >> 1] >> ((~k & 1) << 2)) & 15];
        s += '\t';
        const uint8_t *q = qual ? qual : ql;
        if (q[0] == 0xff) s += '*';
        else for (int32_t k = 0; k < lseq; ++k) s += (char)(q[k] + 33);
    }
    while (aux + 3 <= end) {
        const char ty = (char)aux[2];
        s += '\t'; s += (char)aux[0]; s += (char)aux[1]; s += ':';
        const uint8_t *v = aux + 3;
        auto rd16 = [](const uint8_t *x) { return (uint32_t)(x[0] | (x[1] << 8)); };
        switch (ty) {
        case 'A': if (v + 1 > end) return false; s += "A:"; s += (char)v[0]; aux = v + 1; break;
        case 'c': if (v + 1 > end) return false; s += "i:"; put_int(s, (int8_t)v[0]); aux = v + 1; break;
        case 'C': if (v + 1 > end) return false; s += "i:"; put_int(s, v[0]); aux = v + 1; break;
        case 's': if (v + 2 > end) return false; s += "i:"; put_int(s, (int16_t)rd16(v)); aux = v + 2; break;
        case 'S': if (v + 2 > end) return false; s += "i:"; put_int(s, rd16(v)); aux = v + 2; break;
        case 'i': if (v + 4 > end) return false; s += "i:"; put_int(s, le32(v)); aux = v + 4; break;
        case 'I': if (v + 4 > end) return false; s += "i:"; put_int(s, (uint32_t)le32(v)); aux = v + 4; break;
        case 'f': { if (v + 4 > end) return false; float f; memcpy(&f, v, 4); s += "f:"; put_g(s, f); aux = v + 4; break; }
        case 'd': { if (v + 8 > end) return false; double d; memcpy(&d, v, 8); s += "d:"; put_g(s, d); aux = v + 8; break; }
        case 'Z': case 'H': {
            s += ty; s += ':';
            while (v < end && *v) s += (char)*v++;
            aux = v + 1;
            break;
        }
        case 'B': {
            if (v + 5 > end) return false;
            const char sub = (char)v[0];
            const uint32_t cnt = (uint32_t)le32(v + 1);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            if (v + 5 + es * (size_t)cnt > end) return false;
            s += "B:"; s += sub;
            const uint8_t *e = v + 5;
            for (uint32_t k = 0; k < cnt; ++k, e += es) {
                s += ',';
                switch (sub) {
                case 'c': put_int(s, (int8_t)e[0]); break;
                case 'C': put_int(s, e[0]); break;
                case 's': put_int(s, (int16_t)rd16(e)); break;
                case 'S': put_int(s, rd16(e)); break;
                case 'i': put_int(s, le32(e)); break;
                case 'I': put_int(s, (uint32_t)le32(e)); break;
                case 'f': { float f; memcpy(&f, e, 4); put_g(s, f); break; }
                default: return false;
                }
            }
            aux = e;
            break;
        }
        default: return false;
        }
    }
    s += '\n';
    return true;
}

/* group g of the reader's CURRENT batch; `qual` is laid out like that batch's qual[] (NULL: the record's own
 * qualities).  Unmapped records are skipped, as the reference never stores them (src/secphase.c:340). */
static int sam_write_group_of(spx_sam_writer *w, const Reader &r, const Slot &S, int32_t g, const uint8_t *qual)
{
    if (g < 0 || g >= S.ng) return SPX_EINVAL;
    int n = 0;
    for (int32_t a = S.grp_first[g]; a < S.grp_first[g + 1]; ++a) {
        if (S.flag[a] & SPX_FUNMAP) continue;
        if (n > 10) continue;
        ++n;
        const uint8_t *p = S.ubuf.data() + S.rec_off[a];
        const int32_t bs = le32(p - 4);
        if (!format_sam(r, p, bs, qual ? qual + S.qual_off[a] : nullptr, w->line)) { g_io_err = "corrupt aux block"; return SPX_EINVAL; }
        if (fwrite(w->line.data(), 1, w->line.size(), w->fp) != w->line.size()) { g_io_err = "write failed"; return SPX_EINVAL; }
    }
    return n;
}

extern "C" int spx_sam_write_group(spx_sam_writer *w, const spx_bam_reader *src, int32_t g, const uint8_t *qual)
{
    if (!w || !src) return SPX_EINVAL;
    return sam_write_group_of(w, src->r, src->r.slot[src->r.cur], g, qual);
}

/* the same for a batch handed out earlier and still alive (pipelined callers) */
extern "C" int spx_sam_write_group_of(spx_sam_writer *w, const spx_bam_reader *src, const spx_batch *bt, int32_t g, const uint8_t *qual)
{
    if (!w || !src || !bt) return SPX_EINVAL;
    for (int k = 0; k < Reader::NSLOT; ++k)
        if (&src->r.slot[k].view == bt) return sam_write_group_of(w, src->r, src->r.slot[k], g, qual);
    g_io_err = "batch is no longer held by the reader";
    return SPX_EINVAL;
}

extern "C" int spx_sam_close(spx_sam_writer *w)
{
    if (!w) return SPX_OK;
    int rc = fclose(w->fp) == 0 ? SPX_OK : SPX_EINVAL;
    delete w;
    return rc;
}

/* whole FASTA into RAM (the scorer keeps its own 4-bit copy in HBM; this one feeds spx_set_reference and
 * the contig names of the relabel list) */
extern "C" int spx_fasta_load(const char *path, spx_fasta **out)
{
    if (!path || !out) return SPX_EINVAL;
    *out = nullptr;
    FILE *fp = fopen(path, "rb");
    if (!fp) { g_io_err = std::string("cannot open ") + path; return SPX_EINVAL; }
    spx_fasta *f = new spx_fasta();
    std::vector<char> buf(1 << 22);
    bool in_name = false, name_done = false, bol = true;
    size_t got;
    while ((got = fread(buf.data(), 1, buf.size(), fp)) > 0) {
        for (size_t i = 0; i < got; ++i) {
            const char c = buf[i];
            if (in_name) {
                if (c == '\n') { f->names.push_back(0); in_name = false; bol = true; }
                else if (!name_done) {
                    if (c == ' ' || c == '\t' || c == '\r') name_done = true;
                    else f->names.push_back(c);
                }
                continue;
            }
            if (c == '\n') { bol = true; continue; }
            if (bol && c == '>') {
                f->name_off.push_back((int64_t)f->names.size());
                f->seq_off.push_back((int64_t)f->bases.size());
                in_name = true; name_done = false; bol = false;
                continue;
            }
            bol = false;
            if (c == '\r' || c == ' ' || c == '\t') continue;
            if (f->seq_off.empty()) continue; /* junk before the first header */
            f->bases.push_back(c);
        }
    }
    fclose(fp);
    if (in_name) f->names.push_back(0);
    f->seq_off.push_back((int64_t)f->bases.size());
    f->bases.push_back(0);
    f->ref.n_contigs = (int32_t)f->name_off.size();
    f->ref.name_off = f->name_off.data();
    f->ref.names = f->names.data();
    f->ref.seq_off = f->seq_off.data();
    f->ref.bases = f->bases.data();
    *out = f;
    return SPX_OK;
}
extern "C" const spx_ref *spx_fasta_ref(const spx_fasta *f) { return f ? &f->ref : nullptr; }
extern "C" void spx_fasta_free(spx_fasta *f) { delete f; }
