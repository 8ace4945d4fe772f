/*
 * spx_io.cpp -- input side of the drop-in: name-grouped BAM -> spx_batch blocks, FASTA -> spx_ref.
 *
 * The reference reads through htslib on ONE thread (sam_open/sam_read1 at src/secphase.c:236-268, fai_load/fai_fetch at
 * src/secphase.c:101, submodules/ptMarker/ptMarker.c:739-744).  htslib is not available in this environment, and a
 * single inflate thread could not feed the device anyway, so this is a self-contained reader (round 3 design):
 *
 *   file      mmap'ed; the BGZF block chain (BSIZE of every block header, ISIZE/CRC32 of every trailer) is walked on
 *             the mapping -- no read() copies, 2 cache lines per block;
 *   chunks    runs of blocks (~32 MB inflated) are inflated by a PERSISTENT pool of worker threads (libdeflate through
 *             dlopen when the machine has it, zlib otherwise; CRC32 checked) straight into slots of one arena
 *             (2 MB-aligned, MADV_HUGEPAGE; slots are recycled most-recently-freed first, so a long run touches a
 *             bounded set of pages);
 *   records   ONE walker thread follows the record chain over the inflated chunks in file order (block_size fields,
 *             length validation, group boundaries on a name change -- src/secphase.c:273-279); a record that straddles
 *             two chunks is made contiguous by copying the few bytes in front of it into the head room of the next slot;
 *             fields, tag search (cs / MD / CG) and the CIGAR / name copies of a finished batch run on the pool;
 *   batches   SEQ, QUAL and tag text are NOT copied: a batch points into the arena (offsets from the arena base) and
 *             holds a reference on every slot it touches; finished batches wait in a read-ahead queue.
 * Grouping follows src/secphase.c:273-279: consecutive records with the same read name form a group.
 */
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spx.h"
#include "spx_cpuacc.h"
#include "spx_pool.h"

namespace {

thread_local std::string g_io_err;

double io_now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
bool io_timing()
{
    static const bool on = getenv("SPX_TIMING") != nullptr;
    return on;
}

inline int32_t le32(const uint8_t *p) { return (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)); }

/* ---- libdeflate inflates BGZF blocks 2-3x faster than zlib and has a carry-less-multiply CRC32.  The image ships its
 * runtime library without headers, so the entry points are bound at run time (their C signatures are part of
 * libdeflate's stable ABI); zlib is what runs when the library is not there. ---- */
struct Deflate {
    void *(*alloc)(void) = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;
    Deflate()
    {
        if (getenv("SPX_NO_LIBDEFLATE")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)(void))dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        crc = (uint32_t(*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        if (!alloc || !decompress || !release) alloc = nullptr;
    }
};
const Deflate &deflate_lib()
{
    static const Deflate d;
    return d;
}

/* per-thread inflate state (lives as long as its pool thread) */
struct Inflater {
    void *ld = nullptr;
    z_stream zs;
    bool z_ok = false;
    Inflater()
    {
        const Deflate &DL = deflate_lib();
        if (DL.alloc) ld = DL.alloc();
        if (!ld) {
            memset(&zs, 0, sizeof zs);
            z_ok = inflateInit2(&zs, -15) == Z_OK;
        }
    }
    ~Inflater()
    {
        if (ld) deflate_lib().release(ld);
        if (z_ok) inflateEnd(&zs);
    }
    bool run(const uint8_t *src, size_t clen, uint8_t *dst, size_t ulen)
    {
        if (ld) {
            size_t got = 0;
            return deflate_lib().decompress(ld, src, clen, dst, ulen, &got) == 0 && got == ulen;
        }
        if (!z_ok || inflateReset(&zs) != Z_OK) return false;
        zs.next_in = (Bytef *)src; zs.avail_in = (uInt)clen;
        zs.next_out = dst; zs.avail_out = (uInt)ulen;
        return inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.avail_out == 0;
    }
};
uint32_t crc_of(const uint8_t *p, size_t n)
{
    const Deflate &DL = deflate_lib();
    if (DL.crc) return DL.crc(0, p, n);
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
}

/* ---- arena of chunk slots: one virtual reservation, slots handed out lazily, recycled LIFO ---- */
struct Arena {
    uint8_t *base = nullptr;
    size_t vbytes = 0, slot_bytes = 0;
    int n_max = 0, n_fresh = 0;
    std::vector<int> free_list;
    ~Arena()
    {
        if (base) munmap(base, vbytes);
    }
    bool init(size_t slot, size_t want_bytes)
    {
        slot_bytes = slot;
        for (size_t v = want_bytes; v >= 4 * slot; v /= 2) {
            void *p = mmap(nullptr, v + ((size_t)2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (p == MAP_FAILED) continue;
            vbytes = v + ((size_t)2 << 20);
            const uintptr_t a = ((uintptr_t)p + (((size_t)2 << 20) - 1)) & ~(uintptr_t)(((size_t)2 << 20) - 1);
            base = (uint8_t *)p; /* munmap needs the original address */
            aligned = (uint8_t *)a;
            n_max = (int)(v / slot);
#ifdef MADV_HUGEPAGE
            if (!getenv("SPX_BAM_NO_THP")) (void)madvise(aligned, (size_t)n_max * slot, MADV_HUGEPAGE);
#endif
            return true;
        }
        return false;
    }
    uint8_t *aligned = nullptr;
    uint8_t *slot_ptr(int s) const { return aligned + (size_t)s * slot_bytes; }
    int take() /* caller holds the reader's lock; -1: nothing free and the reservation is used up */
    {
        if (!free_list.empty()) { const int s = free_list.back(); free_list.pop_back(); return s; }
        if (n_fresh < n_max) return n_fresh++;
        return -1;
    }
};

struct Block { size_t boff, coff, clen; uint32_t uoff, ulen, crc; }; /* block start, deflate data, inflated range */

struct Chunk {
    int slot = -1;
    uint8_t *data = nullptr; /* inflated bytes: data[0 .. len); data[-head .. 0) is the head room */
    size_t len = 0;
    int64_t g0 = 0;          /* stream offset (inflated bytes since the reader's start) of data[0] */
    std::vector<Block> blocks;
    std::atomic<int> pending{0}, bad{0};
    int refs = 0;            /* walker + batches; under Reader::mu */
    size_t stop_at = (size_t)-1; /* the shard ends at data + stop_at (a record start) */
    bool last = false;
};

/* one batch under construction / handed out: SEQ, QUAL and tag text are NOT copied, they are offsets from the arena
 * base (only CIGAR is copied, it needs 4-byte alignment) */
struct Batch {
    std::vector<int32_t> grp_first, tid, pos, l_qseq, n_cigar, refid;
    std::vector<int64_t> qname_off, cigar_off, seq_off, qual_off, cs_off, md_off;
    std::vector<int64_t> rec_off; /* first byte (after block_size) of each raw BAM record, from the arena base */
    std::vector<int64_t> grp_voff; /* BGZF virtual offset of every group's first record (index building only) */
    std::vector<uint8_t> rec_new;
    std::vector<uint16_t> flag;
    std::vector<uint32_t> cigar;
    std::vector<char> qnames;
    std::vector<Chunk *> chunks; /* slots this batch points into (one reference each) */
    spx_batch view;
    int ng = 0;
    int rc = 0;
    std::string err;
    bool released = false;
    int64_t end_voff = -1;
};

struct Reader {
    /* file */
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t fsize = 0;
    bool map_is_malloc = false;
    /* header */
    std::vector<std::string> tname;
    std::vector<int64_t> tlen;
    std::vector<int32_t> tmap; /* BAM tid -> contig index of the reference handed to the scorer (-1 unknown) */
    std::string header_text;
    /* configuration */
    int threads = 4;
    size_t chunk_target = (size_t)32 << 20, head = 0;
    std::atomic<int> max_inflight{4}; /* raised by spx_bam_attach_device_inflate while the walker runs */
    int ahead = 2, keep = SPX_BAM_KEEP;
    bool want_voff = false, check_crc = true;
    size_t soft_cap_slots = 0;
    bool header_only = false; /* SPX_BAM_HEADER_ONLY: no arena, no walker -- the device-resident input (spx_devin.cpp) reads the mapping itself */
    size_t start_coff = 0;    /* where the records start: compressed offset of the block, offset inside its inflated bytes */
    uint32_t start_uoff = 0;
    /* block-chain walk (dispatcher state; reader thread only) */
    size_t populated = 0; /* the mapping is faulted in up to here (tasks on the pool, ahead of the walk) */
    size_t fpos = 0;
    bool index_eof = false;
    size_t end_coff = (size_t)-1;
    uint32_t end_uoff = 0;
    int64_t g_next = 0;
    /* shared state */
    std::mutex mu;
    std::condition_variable cv_chunk, cv_out, cv_room;
    Arena arena;
    std::deque<Chunk *> inflight;   /* dispatched, not yet taken by the walker (in file order) */
    std::deque<Batch *> outq;       /* finished batches */
    std::deque<Batch *> handed;     /* handed to the caller, newest last */
    std::atomic<int32_t> batch_groups{0};
    bool closing = false, consumer_waiting = false;
    std::unique_ptr<spx::Pool> pool;
    std::thread walker;
    /* Who inflates a dispatched chunk is decided when somebody has TIME for it, not when it is dispatched: the chunks wait
     * in `unclaimed` (file order); the host pool claims from the FRONT (the walker needs those first, and a chunk takes
     * the pool a few milliseconds) as long as fewer than `host_window` chunks are in work on it; a device worker
     * (spx_bam_attach_device_inflate: compressed bytes -> pinned memory -> HBM -> inflate kernel -> back, tens of
     * milliseconds per chunk, next to no CPU time) that is idle while the pool is saturated claims from the BACK.  Both
     * sides stay busy, the split follows their speeds. */
    std::deque<Chunk *> unclaimed;
    int host_active = 0, host_window = 3;
    spx_bgzf_inflate_fn dev_fn = nullptr;
    void *dev_user = nullptr;
    std::vector<std::thread> dev_workers;
    std::condition_variable cv_dev;
    bool dev_all = false; /* tests: the pool claims nothing */
    int64_t n_chunks_dev = 0, n_chunks_host = 0;
    /* walker cursor (reader thread only) */
    Chunk *cur = nullptr;
    uint8_t *at = nullptr, *end = nullptr;
    int64_t gpos = 0; /* stream offset of `at` */
    std::vector<Block> prev_blocks;
    int64_t prev_g0 = 0;
    bool stream_eof = false;
    std::string werr;
    std::string last_name;
    bool have_last = false;
    int64_t n_records = 0, n_groups_total = 0;
    double t_wait_inflate = 0, t_wait_slot = 0, t_dispatch = 0;
};

bool parse_block_header(const uint8_t *p, size_t avail, size_t *total, size_t *hdr_len, std::string &err)
{
    if (avail < 18 || p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) { err = "not a BGZF block"; return false; }
    const size_t xlen = p[10] | (p[11] << 8);
    if (12 + xlen > avail) { err = "truncated BGZF header"; return false; }
    int bsize = -1;
    /* the BC subfield is first in every BAM written by htslib/samtools; general case: scan the extra field */
    for (size_t o = 0; o + 4 <= xlen;) {
        const uint8_t *e = p + 12 + o;
        const size_t slen = e[2] | (e[3] << 8);
        if (e[0] == 'B' && e[1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = e[4] | (e[5] << 8);
        o += 4 + slen;
    }
    if (bsize < 0) { err = "BGZF block without BC field"; return false; }
    *total = (size_t)bsize + 1;
    *hdr_len = 12 + xlen;
    if (*total < *hdr_len + 8) { err = "corrupt BGZF block"; return false; }
    if (*total > avail) { err = "truncated BGZF block"; return false; }
    return true;
}

void chunk_unref_locked(Reader *r, Chunk *c)
{
    if (--c->refs > 0) return;
    if (c->slot >= 0) r->arena.free_list.push_back(c->slot);
    delete c;
    r->cv_room.notify_all();
}

void inflate_blocks(Reader *r, Chunk *c, size_t b0, size_t b1)
{
    thread_local Inflater inf;
    {
        spx::CpuScope cs(spx::CPU_INFLATE);
        for (size_t q = b0; q < b1; ++q) {
            const Block &b = c->blocks[q];
            if (!inf.run(r->map + b.coff, b.clen, c->data + b.uoff, b.ulen)) c->bad = 1;
        }
    }
    if (c->bad || !r->check_crc) return;
    spx::CpuScope cs(spx::CPU_CRC);
    for (size_t q = b0; q < b1; ++q) {
        const Block &b = c->blocks[q];
        if (crc_of(c->data + b.uoff, b.ulen) != b.crc) c->bad = 2;
    }
}

/* next run of blocks -> a slot -> inflate tasks; false: nothing dispatched (end of the block chain, error, closing) */
void populate_ahead(Reader *r)
{
#ifdef MADV_POPULATE_READ
    /* the chain walk touches two cache lines per block: without this every touch of a new page is a page fault on the
     * walker thread; with it the pool threads map the file 64 MB at a time, 256 MB ahead */
    if (r->map_is_malloc) return;
    const size_t step = (size_t)64 << 20, lead = (size_t)256 << 20;
    if (r->populated < r->fpos) r->populated = r->fpos & ~(size_t)4095;
    while (r->populated < r->fsize && r->populated < r->fpos + lead) {
        const size_t a = r->populated, n = std::min(step, r->fsize - a);
        const uint8_t *base = r->map;
        r->pool->submit([base, a, n] {
            spx::CpuScope cs(spx::CPU_POPULATE);
            (void)madvise((void *)(base + a), n, MADV_POPULATE_READ);
        });
        r->populated = a + n;
    }
#else
    (void)r;
#endif
}

/* the pool takes the front-most waiting chunks while fewer than host_window are in work on it (caller holds r->mu) */
void host_pump_locked(Reader *r)
{
    while (!r->dev_all && !r->closing && r->host_active < r->host_window && !r->unclaimed.empty()) {
        Chunk *cp = r->unclaimed.front();
        r->unclaimed.pop_front();
        ++r->host_active;
        ++r->n_chunks_host;
        const size_t nb = cp->blocks.size(), per = 16;
        cp->pending = (int)((nb + per - 1) / per);
        for (size_t b0 = 0; b0 < nb; b0 += per) {
            const size_t b1 = std::min(nb, b0 + per);
            r->pool->submit([r, cp, b0, b1] {
                inflate_blocks(r, cp, b0, b1);
                if (cp->pending.fetch_sub(1) == 1) {
                    {
                        std::lock_guard<std::mutex> lk(r->mu);
                        --r->host_active;
                        host_pump_locked(r);
                        r->cv_chunk.notify_all();
                    }
                    r->cv_dev.notify_one();
                }
            });
        }
    }
}

bool dispatch_chunk(Reader *r)
{
    if (r->index_eof) return false;
    populate_ahead(r);
    std::unique_ptr<Chunk> c(new Chunk());
    size_t u = 0;
    while (u < r->chunk_target) {
        if (r->fpos >= r->fsize) { r->index_eof = true; break; }
        if (r->fpos == r->end_coff && r->end_uoff == 0) { r->index_eof = true; break; }
        size_t total = 0, hl = 0;
        std::string err;
        if (!parse_block_header(r->map + r->fpos, r->fsize - r->fpos, &total, &hl, err)) {
            r->werr = err;
            r->index_eof = true;
            return false;
        }
        const uint8_t *t = r->map + r->fpos + total - 8;
        const uint32_t crc = (uint32_t)le32(t), isize = (uint32_t)le32(t + 4);
        if (isize > 65536) { r->werr = "corrupt BGZF block (ISIZE)"; r->index_eof = true; return false; }
        const bool stop_here = r->fpos == r->end_coff;
        if (isize > 0) c->blocks.push_back({r->fpos, r->fpos + hl, total - hl - 8, (uint32_t)u, isize, crc});
        if (stop_here) {
            c->stop_at = u + std::min<uint32_t>(r->end_uoff, isize);
            r->index_eof = true;
        }
        u += isize;
        r->fpos += total;
        if (stop_here) break;
    }
    if (r->index_eof) c->last = true;
    c->len = u;
    c->g0 = r->g_next;
    r->g_next += (int64_t)u;
    const double t0 = io_now();
    {
        std::unique_lock<std::mutex> lk(r->mu);
        for (;;) {
            if (r->closing) return false;
            const bool over = (size_t)(r->arena.n_fresh - (int)r->arena.free_list.size()) >= r->soft_cap_slots;
            /* above the soft cap: wait for a slot to come back -- but only while the caller still has finished batches
             * to take (it releases older ones as it goes); a starving caller is never kept waiting for memory */
            if (over && !r->outq.empty() && !r->consumer_waiting) { r->cv_room.wait(lk); continue; }
            c->slot = r->arena.take();
            if (c->slot < 0) {
                if (!r->outq.empty() && !r->consumer_waiting) { r->cv_room.wait(lk); continue; }
                r->werr = "BAM reader: inflate arena exhausted (too many batches held by the caller)";
                return false;
            }
            break;
        }
        c->data = r->arena.slot_ptr(c->slot) + r->head;
        c->refs = 1; /* the walker's */
        c->pending = c->blocks.empty() ? 0 : 1; /* (the claimer sets the real count) */
        r->inflight.push_back(c.get());
    }
    r->t_wait_slot += io_now() - t0;
    Chunk *cp = c.release();
    if (cp->blocks.empty()) { r->cv_chunk.notify_all(); return true; }
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->unclaimed.push_back(cp);
        host_pump_locked(r);
    }
    r->cv_dev.notify_one();
    return true;
}

/* the walker moves on to the next chunk: bytes [at, end) of the current one (the front part of a record) are copied in
 * front of the next chunk's data.  false: end of the stream or error (r->werr set). */
bool advance_chunk(Reader *r)
{
    const double td0 = io_now();
    for (;;) {
        size_t n_in;
        {
            std::lock_guard<std::mutex> lk(r->mu);
            n_in = r->inflight.size();
        }
        if ((int)n_in >= r->max_inflight.load(std::memory_order_relaxed) || !dispatch_chunk(r)) break;
    }
    r->t_dispatch += io_now() - td0;
    if (!r->werr.empty()) return false;
    Chunk *nx = nullptr;
    const double t0 = io_now();
    {
        std::unique_lock<std::mutex> lk(r->mu);
        if (r->inflight.empty()) { r->stream_eof = true; return false; }
        nx = r->inflight.front();
        r->cv_chunk.wait(lk, [&] { return nx->pending.load() == 0 || r->closing; });
        if (r->closing) return false;
        r->inflight.pop_front();
    }
    r->t_wait_inflate += io_now() - t0;
    if (nx->bad) {
        r->werr = nx->bad == 2 ? "BGZF block CRC mismatch" : "inflate failed";
        std::lock_guard<std::mutex> lk(r->mu);
        chunk_unref_locked(r, nx);
        return false;
    }
    const size_t tail = r->cur ? (size_t)(r->end - r->at) : 0;
    if (tail > r->head) {
        r->werr = "BAM record larger than the reader's chunk size";
        std::lock_guard<std::mutex> lk(r->mu);
        chunk_unref_locked(r, nx);
        return false;
    }
    if (tail) memcpy(nx->data - tail, r->at, tail);
    if (r->cur) {
        /* (a record front that already sits in this slot's head room started in an even earlier chunk: keep that one) */
        if (r->want_voff && r->at >= r->cur->data) { r->prev_blocks = r->cur->blocks; r->prev_g0 = r->cur->g0; }
        std::lock_guard<std::mutex> lk(r->mu);
        chunk_unref_locked(r, r->cur);
    }
    r->cur = nx;
    r->at = nx->data - tail;
    r->end = nx->data + nx->len;
    return true;
}

/* makes [at, at+n) contiguous inflated bytes; false at the end of the stream (or error: r->werr) */
bool ensure(Reader *r, size_t n)
{
    while (!r->cur || (size_t)(r->end - r->at) < n) {
        if (r->cur && r->cur->last) { r->stream_eof = true; return false; }
        if (!advance_chunk(r)) return false;
    }
    return true;
}

/* BGZF virtual offset of stream position g (inside the current or the previous chunk) */
int64_t voffset_of(const Reader *r, int64_t g)
{
    const std::vector<Block> *bl = nullptr;
    int64_t g0 = 0;
    if (r->cur && g >= r->cur->g0) { bl = &r->cur->blocks; g0 = r->cur->g0; }
    else { bl = &r->prev_blocks; g0 = r->prev_g0; }
    const int64_t u = g - g0;
    size_t lo = 0, hi = bl->size();
    while (hi - lo > 1) {
        const size_t m = (lo + hi) / 2;
        if ((int64_t)(*bl)[m].uoff <= u) lo = m; else hi = m;
    }
    if (bl->empty()) return -1;
    const Block &b = (*bl)[lo];
    return (int64_t)(((uint64_t)b.boff << 16) | (uint64_t)(u - b.uoff));
}

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
            const uint8_t *z = (const uint8_t *)memchr(v, 0, (size_t)(end - v));
            if (!z) return nullptr;
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
            const uint8_t *z = (const uint8_t *)memchr(v, 0, (size_t)(end - v));
            if (!z) return nullptr; /* unterminated string: the record's aux block is corrupt */
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

void batch_add_chunk(Reader *r, Batch &B, Chunk *c)
{
    if (!B.chunks.empty() && B.chunks.back() == c) return;
    B.chunks.push_back(c);
    std::lock_guard<std::mutex> lk(r->mu);
    ++c->refs;
}

/* fill B with up to max_groups complete name groups */
void fill_batch(Reader *r, Batch &B, int32_t max_groups)
{
    const double t_fill0 = io_now();
    spx::CpuScope cs_walk(spx::CPU_WALK);
    r->t_wait_inflate = r->t_wait_slot = r->t_dispatch = 0;
    const uint8_t *base = r->arena.aligned;
    bool open_group = false;
    /* Pass 1, serial (every record says where the next one starts, and the batch ends on a name change): record
     * offsets, field-length validation, group boundaries.  One cache line per record.  The fields themselves, the tag
     * search (a walk over the aux bytes of every record) and the copies are pass 2, on the pool. */
    for (;;) {
        if (r->cur && r->cur->stop_at != (size_t)-1 && r->at >= r->cur->data + r->cur->stop_at) { r->stream_eof = true; break; }
        if (!ensure(r, 4)) {
            if (!r->werr.empty()) { B.err = r->werr; B.rc = SPX_EINVAL; }
            else if (r->cur && r->end > r->at) { B.err = "truncated BAM record"; B.rc = SPX_EINVAL; }
            break;
        }
        const int32_t bs = le32(r->at);
        if (bs < 32) { B.err = "corrupt BAM record"; B.rc = SPX_EINVAL; break; }
        int64_t voff = -1;
        if (r->want_voff) voff = voffset_of(r, r->gpos); /* before the record may move into the next slot */
        if (!ensure(r, (size_t)bs + 4)) {
            B.err = r->werr.empty() ? "truncated BAM record" : r->werr;
            B.rc = SPX_EINVAL;
            break;
        }
        const uint8_t *p = r->at + 4;
        const uint32_t l_name = p[8];
        const uint32_t ncig = p[12] | (p[13] << 8);
        const int32_t lseq = le32(p + 16);
        const char *name = (const char *)p + 32;
        /* the fixed part announces the lengths of the variable part: none of them may reach past the record, and the
         * name must be NUL-terminated inside it (strlen / the CIGAR copy below would walk off the buffer otherwise) */
        if (lseq < 0 || l_name < 1 || 32 + (uint64_t)l_name + 4 * (uint64_t)ncig + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq > (uint64_t)bs ||
            name[l_name - 1] != 0) {
            B.err = "corrupt BAM record (field lengths exceed the record)";
            B.rc = SPX_EINVAL;
            break;
        }
        /* group boundary on a name change (src/secphase.c:273-279) */
        const bool same = r->have_last && r->last_name.size() + 1 == l_name && memcmp(r->last_name.data(), name, l_name - 1) == 0;
        bool opens = false;
        if (!same || !open_group) {
            if (open_group && B.ng == max_groups) break; /* this record opens the next batch */
            r->last_name.assign(name, l_name - 1);
            r->have_last = true;
            open_group = true;
            opens = true;
            ++B.ng;
            if (r->want_voff) B.grp_voff.push_back(voff);
        }
        batch_add_chunk(r, B, r->cur);
        B.rec_off.push_back((int64_t)(p - base));
        B.rec_new.push_back(opens ? 1 : 0);
        r->at += (size_t)bs + 4;
        r->gpos += (int64_t)bs + 4;
        r->n_records++;
    }
    if (r->want_voff) B.end_voff = (r->cur && r->at < r->end) ? voffset_of(r, r->gpos) : (int64_t)((uint64_t)std::min(r->fpos, r->fsize) << 16);
    const double t_p1 = io_now();
    const int64_t nrec = (int64_t)B.rec_off.size();
    const size_t n = (size_t)nrec;
    B.flag.resize(n); B.refid.resize(n); B.tid.resize(n); B.pos.resize(n); B.l_qseq.resize(n);
    B.n_cigar.resize(n); B.cigar_off.resize(n); B.seq_off.resize(n); B.qual_off.resize(n);
    B.cs_off.resize(n); B.md_off.resize(n);
    std::vector<int64_t> cg_at(n, -1); /* CG:B,I payload when the real CIGAR lives in the tag */
    r->pool->parallel_for(nrec, 1024, [&](int64_t k0, int64_t k1) {
        spx::CpuScope cs_parse(spx::CPU_PARSE);
        for (int64_t k = k0; k < k1; ++k) {
            const uint8_t *p = base + B.rec_off[(size_t)k];
            const int32_t bs = le32(p - 4);
            const int32_t refid = le32(p), posv = le32(p + 4);
            const uint32_t l_name = p[8];
            const uint32_t ncig = p[12] | (p[13] << 8), flg = p[14] | (p[15] << 8);
            const int32_t lseq = le32(p + 16);
            const uint8_t *cig = p + 32 + l_name, *sq = cig + 4 * (size_t)ncig, *ql = sq + ((size_t)lseq + 1) / 2, *aux = ql + lseq,
                          *end = p + bs;
            B.flag[(size_t)k] = (uint16_t)flg;
            B.refid[(size_t)k] = refid;
            B.pos[(size_t)k] = posv;
            B.l_qseq[(size_t)k] = lseq;
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
            B.n_cigar[(size_t)k] = (int32_t)(cg_at[(size_t)k] >= 0 ? cg_n : ncig);
            B.seq_off[(size_t)k] = (int64_t)(sq - base);
            B.qual_off[(size_t)k] = (int64_t)(ql - base);
            const char *csz = aux <= end ? find_tag(aux, end, 'c', 's') : nullptr;
            B.cs_off[(size_t)k] = csz ? (int64_t)((const uint8_t *)csz - base) : -1;
            const char *mdz = (!csz && aux <= end) ? find_tag(aux, end, 'M', 'D') : nullptr; /* only looked at without cs */
            B.md_off[(size_t)k] = mdz ? (int64_t)((const uint8_t *)mdz - base) : -1;
        }
    });
    const double t_p2 = io_now();
    /* offsets of the copied parts (CIGAR words, group names), then the copies */
    int64_t cw = 0, nb = 0;
    std::vector<int64_t> grp_rec; /* first record of every group */
    grp_rec.reserve((size_t)B.ng);
    B.grp_first.reserve((size_t)B.ng + 1);
    B.qname_off.reserve((size_t)B.ng);
    for (int64_t k = 0; k < nrec; ++k) {
        B.cigar_off[(size_t)k] = cw;
        cw += B.n_cigar[(size_t)k];
        if (B.rec_new[(size_t)k]) {
            B.grp_first.push_back((int32_t)k);
            B.qname_off.push_back(nb);
            grp_rec.push_back(k);
            nb += (int64_t)(base + B.rec_off[(size_t)k])[8]; /* l_read_name counts the NUL */
        }
    }
    B.cigar.resize((size_t)cw + 1);
    B.qnames.resize((size_t)nb + 1);
    r->pool->parallel_for(nrec, 2048, [&](int64_t k0, int64_t k1) {
        spx::CpuScope cs_parse(spx::CPU_PARSE);
        for (int64_t k = k0; k < k1; ++k) {
            const uint8_t *p = base + B.rec_off[(size_t)k];
            const uint8_t *src = cg_at[(size_t)k] >= 0 ? base + cg_at[(size_t)k] : p + 32 + p[8];
            uint32_t *dst = B.cigar.data() + B.cigar_off[(size_t)k];
            for (int32_t c = 0; c < B.n_cigar[(size_t)k]; ++c) dst[c] = (uint32_t)le32(src + 4 * (size_t)c);
        }
    });
    r->pool->parallel_for((int64_t)grp_rec.size(), 4096, [&](int64_t g0, int64_t g1) {
        spx::CpuScope cs_parse(spx::CPU_PARSE);
        for (int64_t g = g0; g < g1; ++g) {
            const uint8_t *p = base + B.rec_off[(size_t)grp_rec[(size_t)g]];
            memcpy(B.qnames.data() + B.qname_off[(size_t)g], p + 32, (size_t)p[8]);
        }
    });
    const double t_p3 = io_now();
    B.grp_first.push_back((int32_t)n);
    B.qnames[(size_t)nb] = 0;
    B.cigar[(size_t)cw] = 0;
    spx_batch &b = B.view;
    memset(&b, 0, sizeof b);
    b.n_groups = B.ng;
    b.n_alns = (int32_t)n;
    b.grp_first = B.grp_first.data(); b.qname_off = B.qname_off.data(); b.qnames = B.qnames.data();
    b.flag = B.flag.data(); b.tid = B.tid.data(); b.pos = B.pos.data(); b.l_qseq = B.l_qseq.data();
    b.n_cigar = B.n_cigar.data(); b.cigar_off = B.cigar_off.data(); b.seq_off = B.seq_off.data();
    b.qual_off = B.qual_off.data(); b.cs_off = B.cs_off.data(); b.cigar = B.cigar.data();
    b.seq4 = base; b.qual = base; b.cs = (const char *)base;
    b.md_off = B.md_off.data(); b.md = (const char *)base;
    r->n_groups_total += B.ng;
    if (io_timing())
        fprintf(stderr, "[spx timing] BAM batch: %d groups, %lld records, %.3f s (record chain %.3f [block chain + dispatch %.3f incl. waiting for a slot %.3f; "
                        "waiting for inflate %.3f], fields+tags %.3f, CIGAR+names %.3f)\n", B.ng, (long long)nrec, io_now() - t_fill0, t_p1 - t_fill0,
                r->t_dispatch, r->t_wait_slot, r->t_wait_inflate, t_p2 - t_p1, t_p3 - t_p2);
}

void batch_release_locked(Reader *r, Batch *B)
{
    if (B->released) return;
    B->released = true;
    for (Chunk *c : B->chunks) chunk_unref_locked(r, c);
    B->chunks.clear();
}

void walker_main(Reader *r)
{
    for (;;) {
        int32_t mg = 0;
        {
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv_out.wait(lk, [&] { return r->closing || (r->batch_groups.load() > 0 && (int)r->outq.size() < r->ahead); });
            if (r->closing) return;
            mg = r->batch_groups.load();
        }
        Batch *B = new Batch();
        fill_batch(r, *B, mg);
        const bool fin = B->ng == 0 || B->rc < 0;
        {
            std::lock_guard<std::mutex> lk(r->mu);
            r->outq.push_back(B);
        }
        r->cv_out.notify_all();
        r->cv_room.notify_all();
        if (fin) { /* the sentinel stays at the end of the queue; the walker gives its slots back and rests */
            std::unique_lock<std::mutex> lk(r->mu);
            if (r->cur) { chunk_unref_locked(r, r->cur); r->cur = nullptr; }
            while (!r->inflight.empty()) { /* dispatched but not walked (error / closing): wait for the tasks, then drop */
                Chunk *c = r->inflight.front();
                r->cv_chunk.wait(lk, [&] { return c->pending.load() == 0 || r->closing; });
                if (r->closing) return; /* (nobody claims chunks any more; spx_bam_close drops what is left once the pool has stopped) */
                r->inflight.pop_front();
                chunk_unref_locked(r, c);
            }
            return;
        }
    }
}

/* one device-inflate worker: chunks claimed from the back of `unclaimed` go through the caller's function; a chunk the device cannot do (error from
 * the function itself, not from the data) falls back to the host pool */
void dev_worker_main(Reader *r, int index)
{
    std::vector<spx_bgzf_block> blk;
    for (;;) {
        Chunk *c = nullptr;
        {
            std::unique_lock<std::mutex> lk(r->mu);
            /* a chunk the pool will not get to right away: the pool is saturated, or out of the game */
            r->cv_dev.wait(lk, [&] {
                return r->closing || (r->dev_all ? !r->unclaimed.empty() : (r->unclaimed.size() >= 2 && r->host_active >= r->host_window));
            });
            if (r->closing || r->unclaimed.empty()) return;
            c = r->unclaimed.back();
            r->unclaimed.pop_back();
            c->pending = 1;
            ++r->n_chunks_dev;
        }
        blk.resize(c->blocks.size());
        for (size_t k = 0; k < c->blocks.size(); ++k) {
            const Block &b = c->blocks[k];
            blk[k].data_off = (int64_t)b.coff; blk[k].clen = (uint32_t)b.clen; blk[k].uoff = b.uoff; blk[k].ulen = b.ulen; blk[k].crc = b.crc;
            blk[k].reserved = 0;
        }
        int rc;
        {
            spx::CpuScope cs(spx::CPU_DEV_CHUNK);
            rc = r->dev_fn(r->dev_user, index, r->map, (int64_t)r->fsize, blk.data(), (int32_t)blk.size(), c->data, (int64_t)c->len, r->check_crc ? 1 : 0);
        }
        if (rc == 1) c->bad = 1;       /* corrupt DEFLATE data */
        else if (rc == 2) c->bad = 2;  /* CRC mismatch */
        else if (rc != 0) {            /* the device side failed: the host does this chunk */
            Inflater inf;
            for (const Block &b : c->blocks) {
                if (!inf.run(r->map + b.coff, b.clen, c->data + b.uoff, b.ulen)) { c->bad = 1; continue; }
                if (r->check_crc && crc_of(c->data + b.uoff, b.ulen) != b.crc) c->bad = 2;
            }
        }
        {
            std::lock_guard<std::mutex> lk(r->mu);
            c->pending = 0;
            r->cv_chunk.notify_all();
        }
    }
}

/* serial inflate of the blocks at the start of the file until `need` bytes are there (header parsing) */
bool header_bytes(Reader *r, std::vector<uint8_t> &buf, std::vector<size_t> &blk_coff, std::vector<size_t> &blk_uoff, size_t &fpos, size_t need)
{
    Inflater inf;
    while (buf.size() < need) {
        if (fpos >= r->fsize) return false;
        size_t total = 0, hl = 0;
        std::string err;
        if (!parse_block_header(r->map + fpos, r->fsize - fpos, &total, &hl, err)) { g_io_err = err; return false; }
        const uint32_t isize = (uint32_t)le32(r->map + fpos + total - 4);
        if (isize > 65536) { g_io_err = "corrupt BGZF block (ISIZE)"; return false; }
        blk_coff.push_back(fpos);
        blk_uoff.push_back(buf.size());
        const size_t o = buf.size();
        buf.resize(o + isize);
        if (isize && !inf.run(r->map + fpos + hl, total - hl - 8, buf.data() + o, isize)) { g_io_err = "inflate failed"; return false; }
        fpos += total;
    }
    return true;
}

} // namespace

struct spx_bam_reader { Reader r; };
struct spx_fasta {
    std::vector<int64_t> name_off, seq_off;
    std::vector<char> names, bases;
    spx_ref ref;
};

extern "C" const char *spx_io_last_error(void) { return g_io_err.c_str(); }

extern "C" void spx_bam_default_options(spx_bam_options *o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->threads = 4;
    o->ahead_batches = 2;
    o->start_voffset = -1;
    o->end_voffset = -1;
}

extern "C" int spx_bam_open_opts(const char *path, const spx_bam_options *opt, spx_bam_reader **out)
{
    if (!path || !out) return SPX_EINVAL;
    *out = nullptr;
    spx_bam_options o;
    if (opt) o = *opt; else spx_bam_default_options(&o);
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { g_io_err = std::string("cannot open ") + path; return SPX_EINVAL; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); g_io_err = std::string("cannot stat ") + path; return SPX_EINVAL; }
    std::unique_ptr<spx_bam_reader> h(new spx_bam_reader());
    Reader *r = &h->r;
    r->fd = fd;
    auto bail = [&](const char *msg) {
        g_io_err = msg;
        if (r->map && !r->map_is_malloc) munmap((void *)r->map, r->fsize);
        if (r->map && r->map_is_malloc) free((void *)r->map);
        close(fd);
        return SPX_EINVAL;
    };
    if (S_ISREG(st.st_mode) && st.st_size > 0) {
        r->fsize = (size_t)st.st_size;
        void *m = mmap(nullptr, r->fsize, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            r->map = (const uint8_t *)m;
            (void)madvise(m, r->fsize, MADV_SEQUENTIAL);
        }
    }
    if (!r->map) { /* a pipe, or a file system without mmap: the whole input in memory */
        size_t cap = (size_t)64 << 20, n = 0;
        uint8_t *buf = (uint8_t *)malloc(cap);
        if (!buf) return bail("out of memory");
        for (;;) {
            if (n == cap) {
                cap *= 2;
                uint8_t *q = (uint8_t *)realloc(buf, cap);
                if (!q) { free(buf); return bail("out of memory"); }
                buf = q;
            }
            const ssize_t got = read(fd, buf + n, cap - n);
            if (got < 0 && errno == EINTR) continue;
            if (got < 0) { free(buf); return bail("read error"); }
            if (got == 0) break;
            n += (size_t)got;
        }
        r->map = buf;
        r->fsize = n;
        r->map_is_malloc = true;
    }
    /* ---- header: serial, from the start of the file ---- */
    std::vector<uint8_t> hb;
    std::vector<size_t> hcoff, huoff;
    size_t hpos = 0;
    if (!header_bytes(r, hb, hcoff, huoff, hpos, 12) || memcmp(hb.data(), "BAM\1", 4) != 0) return bail("not a BAM file");
    const int32_t l_text = le32(hb.data() + 4);
    size_t at = 8;
    if (l_text < 0 || !header_bytes(r, hb, hcoff, huoff, hpos, at + (size_t)l_text + 4)) return bail("truncated BAM header");
    r->header_text.assign((const char *)hb.data() + at, strnlen((const char *)hb.data() + at, (size_t)l_text));
    at += (size_t)l_text;
    const int32_t n_ref = le32(hb.data() + at);
    at += 4;
    if (n_ref < 0) return bail("corrupt BAM header");
    for (int32_t i = 0; i < n_ref; ++i) {
        if (!header_bytes(r, hb, hcoff, huoff, hpos, at + 4)) return bail("truncated BAM header");
        const int32_t ln = le32(hb.data() + at);
        at += 4;
        if (ln < 1 || !header_bytes(r, hb, hcoff, huoff, hpos, at + (size_t)ln + 4)) return bail("truncated BAM header");
        r->tname.emplace_back((const char *)hb.data() + at, strnlen((const char *)hb.data() + at, (size_t)ln));
        at += (size_t)ln;
        r->tlen.push_back(le32(hb.data() + at));
        at += 4;
    }
    r->tmap.assign((size_t)n_ref, -1);
    for (int32_t i = 0; i < n_ref; ++i) r->tmap[(size_t)i] = i;
    /* where the records start: inside the block holding header byte `at` (or at the next block) */
    size_t start_coff = hpos;
    uint32_t start_uoff = 0;
    for (size_t k = 0; k < hcoff.size(); ++k)
        if (huoff[k] <= at && (k + 1 == hcoff.size() ? at < hb.size() : at < huoff[k + 1])) { start_coff = hcoff[k]; start_uoff = (uint32_t)(at - huoff[k]); }
    if (o.start_voffset >= 0) { start_coff = (size_t)((uint64_t)o.start_voffset >> 16); start_uoff = (uint32_t)(o.start_voffset & 0xffff); }
    if (o.end_voffset >= 0) { r->end_coff = (size_t)((uint64_t)o.end_voffset >> 16); r->end_uoff = (uint32_t)(o.end_voffset & 0xffff); }
    if (start_coff > r->fsize) return bail("start offset beyond the end of the file");
    hb.clear(); hb.shrink_to_fit();
    r->start_coff = start_coff;
    r->start_uoff = start_uoff;
    if (o.flags & SPX_BAM_HEADER_ONLY) {
        r->header_only = true;
        r->threads = o.threads > 0 ? o.threads : 4;
        r->check_crc = !(o.flags & SPX_BAM_NO_CRC) && !getenv("SPX_BAM_NOCRC");
        r->pool.reset(new spx::Pool(r->threads));
        r->fpos = start_coff;
        *out = h.release();
        return SPX_OK;
    }
    /* ---- configuration ---- */
    r->threads = o.threads > 0 ? o.threads : 4;
    r->ahead = o.ahead_batches > 0 ? o.ahead_batches : 2;
    r->keep = o.keep_batches > 0 ? o.keep_batches : SPX_BAM_KEEP;
    r->want_voff = (o.flags & SPX_BAM_WANT_VOFFSETS) != 0;
    r->check_crc = !(o.flags & SPX_BAM_NO_CRC) && !getenv("SPX_BAM_NOCRC");
    size_t target = o.chunk_bytes > 0 ? (size_t)o.chunk_bytes : ((size_t)32 << 20);
    if (const char *e = getenv("SPX_BAM_CHUNK_KB")) target = (size_t)atoll(e) << 10;
    target = std::max<size_t>(target, (size_t)64 << 10);
    r->chunk_target = target;
    r->head = target + ((size_t)64 << 10);
    const size_t al = (size_t)2 << 20;
    const size_t slot = (2 * r->head + al - 1) & ~(al - 1);
    r->max_inflight = std::max(3, r->threads / 8 + 2);
    r->host_window = std::max(2, r->threads / 16 + 2);
    if (const char *e = getenv("SPX_BAM_HOST_WINDOW")) r->host_window = std::max(1, atoi(e)); /* (experiments) */
    const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
    const size_t phys = (pages > 0 && psz > 0) ? (size_t)pages * (size_t)psz : ((size_t)64 << 30);
    size_t cap_bytes = o.max_bytes > 0 ? (size_t)o.max_bytes : phys / 4;
    if (const char *e = getenv("SPX_BAM_ARENA_GB")) cap_bytes = (size_t)atoll(e) << 30;
    r->soft_cap_slots = std::max<size_t>(cap_bytes / slot, (size_t)r->max_inflight.load() + 4);
    /* the reservation is virtual (MAP_NORESERVE, touched slot by slot): room for the soft cap twice over */
    const size_t want = std::max(std::min<size_t>((size_t)4 << 40, 2 * (r->soft_cap_slots + 8) * slot), 8 * slot);
    if (!r->arena.init(slot, want)) return bail("cannot reserve the inflate arena");
    r->pool.reset(new spx::Pool(r->threads));
    r->fpos = start_coff;
    /* the walker starts start_uoff bytes into its first chunk */
    r->batch_groups = o.batch_groups > 0 ? o.batch_groups : 0;
    if (start_uoff) {
        /* position the cursor: take the first chunk synchronously */
        if (!advance_chunk(r)) {
            const std::string e = r->werr.empty() ? "start offset beyond the end of the file" : r->werr;
            r->pool.reset();
            return bail(e.c_str());
        }
        if ((size_t)start_uoff > r->cur->len) { r->pool.reset(); return bail("start offset beyond its block"); }
        r->at += start_uoff;
        r->gpos = start_uoff;
    }
    r->walker = std::thread(walker_main, r);
    *out = h.release();
    return SPX_OK;
}

extern "C" int spx_bam_open(const char *path, int threads, spx_bam_reader **out)
{
    spx_bam_options o;
    spx_bam_default_options(&o);
    o.threads = threads > 0 ? threads : 4;
    return spx_bam_open_opts(path, &o, out);
}

extern "C" int32_t spx_bam_n_targets(const spx_bam_reader *h) { return h ? (int32_t)h->r.tname.size() : 0; }
extern "C" const char *spx_bam_target_name(const spx_bam_reader *h, int32_t i)
{
    return (h && i >= 0 && (size_t)i < h->r.tname.size()) ? h->r.tname[i].c_str() : nullptr;
}

/* BAM target ids -> contig indices of `ref` (by name); alignments on contigs the FASTA lacks get tid -1
 * and their group is rejected by the scorer with SPX_EINVAL.  Applies to the batches handed out from now on (the
 * reader may already be inflating and cutting batches in the background: target ids are mapped at hand-out). */
extern "C" int spx_bam_bind_reference(spx_bam_reader *h, const spx_ref *ref)
{
    if (!h || !ref) return SPX_EINVAL;
    Reader *r = &h->r;
    std::vector<int32_t> tm(r->tname.size(), -1);
    int missing = 0;
    if (r->tname.size() > 64) { /* many targets: by sorted names instead of the quadratic scan */
        std::vector<std::pair<std::string, int32_t>> byname;
        byname.reserve((size_t)ref->n_contigs);
        for (int32_t c = ref->n_contigs - 1; c >= 0; --c) byname.emplace_back(ref->names + ref->name_off[c], c);
        std::stable_sort(byname.begin(), byname.end(), [](const std::pair<std::string, int32_t> &a, const std::pair<std::string, int32_t> &b) { return a.first < b.first; });
        for (size_t i = 0; i < r->tname.size(); ++i) {
            auto it = std::lower_bound(byname.begin(), byname.end(), r->tname[i],
                                       [](const std::pair<std::string, int32_t> &a, const std::string &k) { return a.first < k; });
            /* the FIRST contig of that name, like the linear scan */
            int32_t best = -1;
            for (; it != byname.end() && it->first == r->tname[i]; ++it) best = best < 0 ? it->second : std::min(best, it->second);
            tm[i] = best;
            if (best < 0) ++missing;
        }
    } else
        for (size_t i = 0; i < r->tname.size(); ++i) {
            for (int32_t c = 0; c < ref->n_contigs; ++c)
                if (r->tname[i] == ref->names + ref->name_off[c]) { tm[i] = c; break; }
            if (tm[i] < 0) ++missing;
        }
    std::lock_guard<std::mutex> lk(r->mu);
    r->tmap.swap(tm);
    return missing;
}

/* up to max_groups complete name groups; the batch stays valid for the next SPX_BAM_KEEP calls (or until
 * spx_bam_release_batch).  Following batches are inflated and cut in the background.  Returns the number of groups
 * (0 at end of file) or SPX_E*.  The batch size of a reader is fixed by its first call (or its options): batches that
 * were cut ahead keep the size they were cut with. */
extern "C" int spx_bam_next_batch(spx_bam_reader *h, int32_t max_groups, const spx_batch **out)
{
    if (!h || !out || max_groups <= 0) return SPX_EINVAL;
    Reader *r = &h->r;
    if (r->header_only) { g_io_err = "reader was opened with SPX_BAM_HEADER_ONLY"; return SPX_EINVAL; }
    Batch *B = nullptr;
    {
        std::unique_lock<std::mutex> lk(r->mu);
        if (r->batch_groups.load() != max_groups) r->batch_groups = max_groups;
        r->consumer_waiting = true;
        r->cv_out.notify_all();
        r->cv_room.notify_all();
        r->cv_out.wait(lk, [&] { return !r->outq.empty(); });
        r->consumer_waiting = false;
        B = r->outq.front();
        if (B->ng == 0 || B->rc < 0) { /* end of the stream: the sentinel stays */
            *out = &B->view;
            if (B->rc < 0) { g_io_err = B->err; return B->rc; }
            return 0;
        }
        r->outq.pop_front();
        for (size_t k = 0; k < B->refid.size(); ++k) {
            const int32_t t = B->refid[k];
            B->tid[k] = (t >= 0 && (size_t)t < r->tmap.size()) ? r->tmap[(size_t)t] : -1;
        }
        r->handed.push_back(B);
        while ((int)r->handed.size() > r->keep + 1) {
            Batch *old = r->handed.front();
            r->handed.pop_front();
            batch_release_locked(r, old);
            delete old;
        }
    }
    r->cv_out.notify_all();
    *out = &B->view;
    return B->ng;
}

/* the caller is done with a batch it was handed: its share of the inflate arena is recycled now instead of
 * SPX_BAM_KEEP calls later (the spx_batch itself must not be used afterwards) */
extern "C" int spx_bam_release_batch(spx_bam_reader *h, const spx_batch *bt)
{
    if (!h || !bt) return SPX_EINVAL;
    Reader *r = &h->r;
    std::lock_guard<std::mutex> lk(r->mu);
    for (Batch *B : r->handed)
        if (&B->view == bt) { batch_release_locked(r, B); return SPX_OK; }
    return SPX_EINVAL;
}

/* internal (spx_devin.cpp): one BGZF block on the calling thread (its thread's inflate state): 0 ok, 1 corrupt data, 2 CRC mismatch */
extern "C" int spx_internal_inflate_block(const uint8_t *src, size_t clen, uint8_t *dst, size_t ulen, uint32_t crc, int check_crc)
{
    thread_local Inflater inf;
    {
        spx::CpuScope cs(spx::CPU_INFLATE);
        if (!inf.run(src, clen, dst, ulen)) return 1;
    }
    if (check_crc) {
        spx::CpuScope cs(spx::CPU_CRC);
        if (crc_of(dst, ulen) != crc) return 2;
    }
    return 0;
}

/* internal (spx_devin.cpp): the mapping and where the records start / end in it */
extern "C" int spx_internal_bam_layout(spx_bam_reader *h, const uint8_t **map, int64_t *fsize, int64_t *start_coff, int32_t *start_uoff,
                                       int64_t *end_coff, int32_t *end_uoff, int32_t *check_crc)
{
    if (!h) return SPX_EINVAL;
    Reader *r = &h->r;
    *map = r->map; *fsize = (int64_t)r->fsize; *start_coff = (int64_t)r->start_coff; *start_uoff = (int32_t)r->start_uoff;
    *end_coff = r->end_coff == (size_t)-1 ? -1 : (int64_t)r->end_coff; *end_uoff = (int32_t)r->end_uoff;
    *check_crc = r->check_crc ? 1 : 0;
    return SPX_OK;
}
/* internal: the target-id map of spx_bam_bind_reference (n entries) */
extern "C" int32_t spx_internal_bam_tmap(spx_bam_reader *h, int32_t *dst, int32_t cap)
{
    if (!h) return 0;
    Reader *r = &h->r;
    std::lock_guard<std::mutex> lk(r->mu);
    const int32_t n = (int32_t)r->tmap.size();
    for (int32_t i = 0; i < n && i < cap; ++i) dst[i] = r->tmap[(size_t)i];
    return n;
}
/* internal: a task on the reader's thread pool (faulting the mapping in ahead of the device input, copies into pinned memory) */
extern "C" void spx_internal_bam_parallel(spx_bam_reader *h, int64_t n, int64_t grain, void (*fn)(void *, int64_t, int64_t), void *user)
{
    if (!h || !h->r.pool) { fn(user, 0, n); return; }
    h->r.pool->parallel_for(n, grain, [&](int64_t a, int64_t b) { fn(user, a, b); });
}
extern "C" void spx_internal_bam_submit(spx_bam_reader *h, void (*fn)(void *), void *user)
{
    if (!h || !h->r.pool) { fn(user); return; }
    h->r.pool->submit([fn, user] { fn(user); });
}

extern "C" int spx_bam_attach_device_inflate(spx_bam_reader *h, spx_bgzf_inflate_fn fn, void *user, int32_t n_workers)
{
    if (!h || !fn || n_workers < 1 || n_workers > 32) return SPX_EINVAL;
    Reader *r = &h->r;
    if (r->header_only) return SPX_EINVAL;
    std::lock_guard<std::mutex> lk(r->mu);
    if (r->dev_fn) return SPX_EINVAL;
    r->dev_user = user;
    if (getenv("SPX_BAM_DEVICE_ALL")) r->dev_all = true; /* tests: every chunk from now on goes to the device */
    /* a chunk on the device takes tens of milliseconds (one wave per block: latency, not throughput), the walker takes
     * chunks in file order: look further ahead, so that device chunks are dispatched long before they are needed */
    r->max_inflight += 3 * n_workers;
    for (int k = 0; k < n_workers; ++k) r->dev_workers.emplace_back(dev_worker_main, r, k);
    r->dev_fn = fn;
    return SPX_OK;
}

extern "C" void spx_bam_inflate_counts(const spx_bam_reader *h, int64_t *chunks_host, int64_t *chunks_device)
{
    if (!h) return;
    Reader *r = const_cast<Reader *>(&h->r);
    std::lock_guard<std::mutex> lk(r->mu);
    if (chunks_host) *chunks_host = r->n_chunks_host;
    if (chunks_device) *chunks_device = r->n_chunks_dev;
}

/* Give the reader's pages back to the kernel ON THE POOL: madvise(MADV_DONTNEED) takes the address-space lock shared, so
 * the 13 GB of a 262 144-group run (arena slots touched + the populated file mapping) go in parallel instead of in one
 * serial munmap / process exit (0.36 + 0.10 s, or 0.8 s of the parent's wait for the exit).  The batches handed out are
 * dead afterwards: called when every output is written. */
static void drop_pages(Reader *r)
{
    if (!r->pool) return;
    struct Piece { uint8_t *p; size_t n; };
    std::vector<Piece> pieces;
    const size_t step = (size_t)64 << 20;
    int n_slots;
    {
        std::lock_guard<std::mutex> lk(r->mu); /* (the walker may be taking a fresh slot right now) */
        n_slots = r->arena.n_fresh;
    }
    for (int s = 0; s < n_slots; ++s) pieces.push_back({r->arena.slot_ptr(s), r->arena.slot_bytes});
    if (r->map && !r->map_is_malloc)
        for (size_t a = 0; a < r->fsize; a += step) pieces.push_back({(uint8_t *)r->map + a, std::min(step, r->fsize - a)});
    r->pool->parallel_for((int64_t)pieces.size(), 1, [&](int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; ++k) (void)madvise(pieces[(size_t)k].p, pieces[(size_t)k].n, MADV_DONTNEED);
    });
}

extern "C" void spx_bam_drop_pages(spx_bam_reader *h)
{
    if (h) drop_pages(&h->r);
}

extern "C" void spx_bam_close(spx_bam_reader *h)
{
    if (!h) return;
    Reader *r = &h->r;
    const double tc0 = io_now();
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->closing = true;
    }
    r->cv_out.notify_all();
    r->cv_room.notify_all();
    r->cv_chunk.notify_all();
    r->cv_dev.notify_all();
    if (r->walker.joinable()) r->walker.join();
    r->cv_dev.notify_all();
    for (auto &t : r->dev_workers) if (t.joinable()) t.join(); /* (chunks still queued are finished first: their slots are live) */
    drop_pages(r);   /* (behind whatever inflate tasks are still queued; nobody reads the arena any more) */
    r->pool.reset(); /* joins the workers: no inflate task is running after this */
    {
        std::lock_guard<std::mutex> lk(r->mu);
        for (Batch *B : r->outq) { batch_release_locked(r, B); delete B; }
        r->outq.clear();
        for (Batch *B : r->handed) { batch_release_locked(r, B); delete B; }
        r->handed.clear();
        if (r->cur) { chunk_unref_locked(r, r->cur); r->cur = nullptr; }
        for (Chunk *c : r->inflight) chunk_unref_locked(r, c);
        r->inflight.clear();
    }
    const double tc1 = io_now();
    if (r->map && !r->map_is_malloc) munmap((void *)r->map, r->fsize);
    if (r->map && r->map_is_malloc) free((void *)r->map);
    if (r->fd >= 0) close(r->fd);
    const double tc2 = io_now();
    const int slots = r->arena.n_fresh;
    delete h;
    if (io_timing())
        fprintf(stderr, "[spx timing] reader closed: threads joined + batches released %.3f s, file unmapped %.3f s, arena (%d slots touched) unmapped %.3f s\n",
                tc1 - tc0, tc2 - tc1, slots, io_now() - tc2);
}

/* ---- index of group starts in the reference's on-disk format (src/secphase_index.c:76-119: int64 count, then that
 * many int64 BGZF virtual offsets; consumer get_offset_array, src/secphase.c:357-385): the virtual offset of the first
 * record of every step-th read group, and the offset where the records end.  A reader opened with
 * start_voffset = a[i], end_voffset = a[j] yields exactly the groups [i*step, j*step). ---- */
extern "C" int64_t spx_bam_index_build(const char *path, int threads, int32_t step_groups, int64_t *offsets, int64_t capacity)
{
    if (!path || step_groups < 1) return SPX_EINVAL;
    spx_bam_options o;
    spx_bam_default_options(&o);
    o.threads = threads;
    o.flags = SPX_BAM_WANT_VOFFSETS;
    o.batch_groups = 8192;
    spx_bam_reader *h = nullptr;
    int rc = spx_bam_open_opts(path, &o, &h);
    if (rc != SPX_OK) return rc;
    int64_t n = 0, g = 0, end = -1;
    for (;;) {
        const spx_batch *bt = nullptr;
        const int ng = spx_bam_next_batch(h, o.batch_groups, &bt);
        if (ng < 0) { spx_bam_close(h); return ng; }
        Batch *B = nullptr;
        {
            std::lock_guard<std::mutex> lk(h->r.mu);
            B = ng > 0 ? h->r.handed.back() : h->r.outq.front();
        }
        end = B->end_voff;
        if (ng == 0) break;
        for (int k = 0; k < ng; ++k, ++g)
            if (g % step_groups == 0) {
                if (offsets && n < capacity) offsets[n] = B->grp_voff[(size_t)k];
                ++n;
            }
        spx_bam_release_batch(h, bt);
    }
    if (offsets && n < capacity) offsets[n] = end;
    ++n;
    spx_bam_close(h);
    return n;
}

extern "C" int spx_bam_index_save(const char *index_path, const int64_t *offsets, int64_t n)
{
    if (!index_path || !offsets || n < 0) return SPX_EINVAL;
    FILE *fp = fopen(index_path, "wb");
    if (!fp) { g_io_err = std::string("cannot create ") + index_path; return SPX_EINVAL; }
    const bool ok = fwrite(&n, sizeof(int64_t), 1, fp) == 1 && (n == 0 || fwrite(offsets, sizeof(int64_t), (size_t)n, fp) == (size_t)n);
    return (fclose(fp) == 0 && ok) ? SPX_OK : SPX_EINVAL;
}

extern "C" int64_t spx_bam_index_load(const char *index_path, int64_t *offsets, int64_t capacity)
{
    if (!index_path) return SPX_EINVAL;
    FILE *fp = fopen(index_path, "rb");
    if (!fp) { g_io_err = std::string("cannot open ") + index_path; return SPX_EINVAL; }
    int64_t n = 0;
    if (fread(&n, sizeof(int64_t), 1, fp) != 1 || n < 0) { fclose(fp); g_io_err = "corrupt index"; return SPX_EINVAL; }
    if (offsets) {
        const int64_t m = std::min(n, capacity);
        if (m > 0 && fread(offsets, sizeof(int64_t), (size_t)m, fp) != (size_t)m) { fclose(fp); g_io_err = "truncated index"; return SPX_EINVAL; }
    }
    fclose(fp);
    return n;
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
static int sam_write_group_of(spx_sam_writer *w, const Reader &r, const Batch &S, int32_t g, const uint8_t *qual)
{
    if (g < 0 || g >= S.ng) return SPX_EINVAL;
    int n = 0;
    for (int32_t a = S.grp_first[g]; a < S.grp_first[g + 1]; ++a) {
        if (S.flag[a] & SPX_FUNMAP) continue;
        if (n > 10) continue;
        ++n;
        const uint8_t *p = r.arena.aligned + S.rec_off[a];
        const int32_t bs = le32(p - 4);
        if (!format_sam(r, p, bs, qual ? qual + S.qual_off[a] : nullptr, w->line)) { g_io_err = "corrupt aux block"; return SPX_EINVAL; }
        if (fwrite(w->line.data(), 1, w->line.size(), w->fp) != w->line.size()) { g_io_err = "write failed"; return SPX_EINVAL; }
    }
    return n;
}

extern "C" int spx_sam_write_group(spx_sam_writer *w, const spx_bam_reader *src, int32_t g, const uint8_t *qual)
{
    if (!w || !src) return SPX_EINVAL;
    const Batch *B = nullptr;
    {
        std::lock_guard<std::mutex> lk(const_cast<Reader &>(src->r).mu);
        if (!src->r.handed.empty()) B = src->r.handed.back();
    }
    if (!B || B->released) { g_io_err = "no current batch"; return SPX_EINVAL; }
    return sam_write_group_of(w, src->r, *B, g, qual);
}

/* the same for a batch handed out earlier and still alive (pipelined callers) */
extern "C" int spx_sam_write_group_of(spx_sam_writer *w, const spx_bam_reader *src, const spx_batch *bt, int32_t g, const uint8_t *qual)
{
    if (!w || !src || !bt) return SPX_EINVAL;
    const Batch *B = nullptr;
    {
        std::lock_guard<std::mutex> lk(const_cast<Reader &>(src->r).mu);
        for (const Batch *b : src->r.handed)
            if (&b->view == bt && !b->released) B = b;
    }
    if (B) return sam_write_group_of(w, src->r, *B, g, qual);
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
 * the contig names of the relabel list).  The file is mapped and copied line by line (memchr + memcpy); blanks inside
 * sequence lines -- legal, rare -- send a line through the character filter. */
extern "C" int spx_fasta_load(const char *path, spx_fasta **out)
{
    if (!path || !out) return SPX_EINVAL;
    *out = nullptr;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { g_io_err = std::string("cannot open ") + path; return SPX_EINVAL; }
    struct stat st;
    std::vector<char> slurp;
    const char *buf = nullptr;
    size_t n = 0;
    void *m = MAP_FAILED;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) { buf = (const char *)m; n = (size_t)st.st_size; (void)madvise(m, n, MADV_SEQUENTIAL); }
    }
    if (!buf) {
        char tmp[1 << 16];
        ssize_t got;
        while ((got = read(fd, tmp, sizeof tmp)) > 0) slurp.insert(slurp.end(), tmp, tmp + got);
        buf = slurp.data();
        n = slurp.size();
    }
    spx_fasta *f = new spx_fasta();
    f->bases.reserve(n + 1);
    const char *p = buf, *e = buf + n;
    while (p < e) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        const char *le = nl ? nl : e;
        if (*p == '>') {
            f->name_off.push_back((int64_t)f->names.size());
            f->seq_off.push_back((int64_t)f->bases.size());
            const char *q = p + 1;
            while (q < le && *q != ' ' && *q != '\t' && *q != '\r') ++q;
            f->names.insert(f->names.end(), p + 1, q);
            f->names.push_back(0);
        } else if (!f->seq_off.empty() && le > p) { /* (junk before the first header is skipped) */
            const char *l1 = le;
            if (l1[-1] == '\r') --l1;
            const size_t len = (size_t)(l1 - p);
            if (len && !memchr(p, ' ', len) && !memchr(p, '\t', len) && !memchr(p, '\r', len)) f->bases.insert(f->bases.end(), p, l1);
            else
                for (const char *q = p; q < l1; ++q)
                    if (*q != ' ' && *q != '\t' && *q != '\r') f->bases.push_back(*q);
        }
        p = nl ? nl + 1 : e;
    }
    if (m != MAP_FAILED) munmap(m, n);
    close(fd);
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

/* ---- internal accessors for spx_correct.cpp (the correct_bam counterpart re-emits raw records) ---- */
extern "C" int spx_internal_bam_record(const spx_bam_reader *src, const spx_batch *bt, int32_t a, const uint8_t **rec, int32_t *block_size)
{
    if (!src || !bt || !rec || !block_size) return SPX_EINVAL;
    const Batch *B = nullptr;
    {
        std::lock_guard<std::mutex> lk(const_cast<Reader &>(src->r).mu);
        for (const Batch *b : src->r.handed)
            if (&b->view == bt && !b->released) B = b;
    }
    if (!B || a < 0 || (size_t)a >= B->rec_off.size()) return SPX_EINVAL;
    const uint8_t *p = src->r.arena.aligned + B->rec_off[(size_t)a];
    *rec = p;
    *block_size = le32(p - 4);
    return SPX_OK;
}

extern "C" int spx_internal_bam_header(const spx_bam_reader *src, const char **text, int64_t *text_len, int32_t *n_targets)
{
    if (!src) return SPX_EINVAL;
    if (text) *text = src->r.header_text.data();
    if (text_len) *text_len = (int64_t)src->r.header_text.size();
    if (n_targets) *n_targets = (int32_t)src->r.tname.size();
    return SPX_OK;
}

extern "C" int64_t spx_internal_bam_target_len(const spx_bam_reader *src, int32_t i)
{
    return (src && i >= 0 && (size_t)i < src->r.tlen.size()) ? src->r.tlen[(size_t)i] : -1;
}

/* one SAM line of a raw record (sam_format1) appended to *line (a std::string the caller owns) */
extern "C" int spx_internal_format_sam(const spx_bam_reader *src, const uint8_t *rec, int32_t block_size, void *std_string_out)
{
    if (!src || !rec || !std_string_out) return SPX_EINVAL;
    std::string tmp;
    if (!format_sam(src->r, rec, block_size, nullptr, tmp)) { g_io_err = "corrupt aux block"; return SPX_EINVAL; }
    static_cast<std::string *>(std_string_out)->append(tmp);
    return SPX_OK;
}
