/*
 * spx_devin.cpp -- device-resident BAM input (spx_dbam_*): compressed bytes in, staged work lists out.
 *
 * The reference reads one record at a time through htslib on the main thread (sam_read1, the group scan and the dispatch
 * filter of /root/reference/programs/src/secphase.c:230-351); this repository's host reader (spx_io.cpp) inflates on a
 * thread pool and on the device, but brings the inflated bytes BACK to the host, walks / parses / stages them there and
 * sends them up again -- on the MI355X boxes (16 cores of CPU time per container) that host work is what the command line
 * waits for (the GPU idles ~80 % of an end-to-end run).  Here the host touches COMPRESSED bytes only:
 *
 *   host    BGZF block chain on the mapping (BSIZE / ISIZE / CRC fields), runs of blocks (~1 GB inflated = a SEGMENT),
 *           compressed bytes -> ring of pinned chunks -> HBM on a copy-only stream;
 *   device  bgzf_inflate_kernel (spx_inflate_kernels.hip) -> record chain, fields, cs / MD / CG tags, name groups,
 *           dispatch filter, gather into the staged layout (spx_devin_kernels.hip) -> a STAGED spx_work, which the in-order
 *           pipeline (spx_pipe.cpp) prepares, launches and collects like any other;
 *   back    per group: name, per record: flag, target, position (the relabel list prints them) -- ~60 bytes per group.
 *
 * Segments are independent up to their front: the last, possibly open, name group of segment k (and the front of a record
 * that continues) is the CARRY of segment k + 1 -- a few hundred KB that travel through a pinned host buffer, so that
 * consecutive segments may sit on DIFFERENT devices: `secphase --devices 0-7` has one input pipeline per GPU (upload,
 * inflate, parse, gather), only the carry hand-over is serial.  Every lane has an uploader thread (segments are cut and
 * sent ahead of the parsing) and a parser thread (waits for the carry, runs the kernels, hands the work lists out in
 * file order).
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spx.h"
#include "spx_cpuacc.h"
#include "spx_devin.h"

extern "C" int spx_internal_bam_layout(spx_bam_reader *h, const uint8_t **map, int64_t *fsize, int64_t *start_coff, int32_t *start_uoff,
                                       int64_t *end_coff, int32_t *end_uoff, int32_t *check_crc);
extern "C" int32_t spx_internal_bam_tmap(spx_bam_reader *h, int32_t *dst, int32_t cap);
extern "C" void spx_internal_bam_parallel(spx_bam_reader *h, int64_t n, int64_t grain, void (*fn)(void *, int64_t, int64_t), void *user);
extern "C" void spx_internal_bam_submit(spx_bam_reader *h, void (*fn)(void *), void *user);
extern "C" int spx_internal_inflate_block(const uint8_t *src, size_t clen, uint8_t *dst, size_t ulen, uint32_t crc, int check_crc);
extern "C" void spx_internal_set_error(const char *msg);
extern "C" int spx_internal_ctx_device(spx_ctx *c);
struct spx_devstage_sizes {
    int64_t n_groups_in, n_dgroups, n_slots, cigar_words, seq_bytes, qual_bytes, text_bytes, ops_bound, conf_bound, mm_bound, info_bytes;
};
extern "C" int spx_internal_devstage_begin(spx_ctx *c, const spx_params *par, const spx_devstage_sizes *sz, spx_work **out, spx_din_out *O, char **d_info);
extern "C" int spx_internal_devstage_finish(spx_ctx *c, spx_work *w, const uint8_t *grp_disp, hipStream_t st);
extern "C" size_t spx_bgzf_inflate_scratch_bytes(int32_t n_blocks);
extern "C" hipError_t spx_launch_bgzf_inflate2(const uint8_t *comp, const void *blocks, int32_t n_blocks, uint8_t *out, int32_t *status, int check_crc,
                                               void *scratch, hipStream_t st);
extern "C" hipError_t spx_launch_bgzf_inflate(const uint8_t *comp, const void *blocks, int32_t n_blocks, uint8_t *out, int32_t *status,
                                              int check_crc, hipStream_t st);
extern "C" size_t spx_din_scan_temp_bytes(int64_t n_items);
extern "C" hipError_t spx_din_chain(const spx_din_args *A, hipStream_t st);
extern "C" hipError_t spx_din_groups(const spx_din_args *A, int64_t n_rec, void *temp, size_t temp_bytes, hipStream_t st);
extern "C" hipError_t spx_din_image(const spx_din_args *A, const spx_din_out *O, const spx_din_range *Q, hipStream_t st);
extern "C" hipError_t spx_din_inflate_status(const int32_t *status, int32_t n, spx_din_counts *C, hipStream_t st);

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool timing_on()
{
    static const bool on = getenv("SPX_TIMING") != nullptr;
    return on;
}
inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct BlockDesc { /* = spx_inflate_kernels.hip BlockDesc */
    int64_t in_off, out_off;
    uint32_t clen, ulen, crc, pad;
};

constexpr int kSlots = 3;  /* segment buffers per lane: one being parsed, one inflating, one on its way up */
constexpr int kPins = 4;   /* pinned chunks of a lane's upload ring */

struct Seg {
    int64_t index = 0;
    int slot = -1;
    std::vector<BlockDesc> blocks;
    std::vector<int64_t> bstart; /* [n + 1] offsets in the segment buffer */
    int64_t c0 = 0, c1 = 0;      /* file range that holds the blocks */
    int64_t ulen = 0;            /* inflated bytes */
    int64_t p0_extra = 0;        /* first segment: the records start this far into the first block */
    int64_t stop_at = -1;        /* shard end (end_voffset): the records end this far into the inflated bytes */
    bool last = false;
    double t_cut = 0, t_up = 0;
    size_t n_dev_blocks = 0;     /* blocks [0, n_dev_blocks) are inflated on the device, the rest on the host pool */
};

struct NameBatch { /* what the host keeps of a work list's groups: a spx_batch with names, flags, targets, positions only */
    spx_batch view;
    void *pinned = nullptr;
    size_t cap = 0;
};

struct Item {
    spx_work *work = nullptr;
    int lane = 0;
    NameBatch *names = nullptr;
    int32_t n_groups = 0;
};

struct Lane;
} // namespace

struct spx_dbam {
    spx_bam_reader *hdr = nullptr;
    const uint8_t *map = nullptr;
    int64_t fsize = 0, start_coff = 0, end_coff = -1;
    int32_t start_uoff = 0, end_uoff = 0, check_crc = 1;
    spx_params par;
    int32_t max_groups = 95000, ahead = 3;
    int64_t seg_bytes = (int64_t)1 << 30, carry_cap = (int64_t)256 << 20;
    size_t carry_host_cap = 0; /* bytes of the pinned carry_host buffer */
    int64_t seg_first = 0; /* > 0: the first segment of every lane has this size, each following one twice its predecessor's up to seg_bytes */
    std::vector<Lane *> lanes;
    /* cutting segments (any uploader, under cut_mu) */
    std::mutex cut_mu;
    int64_t fpos = 0, populated = 0;
    std::atomic<int64_t> next_index{0};
    bool cut_eof = false;
    /* carry hand-over + results, in segment order */
    std::mutex mu;
    std::condition_variable cv;
    int64_t carry_for = 0;   /* the carry in carry_host belongs in front of this segment */
    int64_t carry_len = 0;
    uint8_t *carry_host = nullptr;
    std::map<int64_t, std::vector<Item>> results; /* finished segments not yet handed out */
    int64_t next_out = 0;       /* segment the consumer takes next */
    std::deque<Item> out_items; /* its work lists */
    int64_t total_segments = -1; /* known once the last segment has been cut */
    int rc = SPX_OK;
    std::string err;
    bool closing = false;
    std::vector<NameBatch *> free_names, live_names;
    int64_t bytes_up = 0, n_segments = 0, bytes_host_inflated = 0, bytes_dev_inflated = 0;
    double host_share = 0; /* fraction of every segment's inflated bytes that the host pool inflates (uploaded raw) */
    double t_parse = 0, t_wait_carry = 0, t_wait_inflate = 0, t_chain = 0, t_groups = 0, t_image = 0, t_upload = 0;
    std::vector<int32_t> tmap;
};

namespace {

#define DCHK(x)                                                                                              \
    do {                                                                                                     \
        hipError_t e_ = (x);                                                                                 \
        if (e_ != hipSuccess) return fail_here(std::string(#x) + ": " + hipGetErrorString(e_), SPX_EHIP); \
    } while (0)

struct Lane {
    spx_dbam *D = nullptr;
    spx_ctx *ctx = nullptr;
    int index = 0, device = 0;
    hipStream_t up_stream = nullptr, up2_stream = nullptr, inf_stream = nullptr, in_stream = nullptr;
    struct Slot {
        uint8_t *d_buf = nullptr, *d_comp = nullptr;
        size_t buf_cap = 0, comp_cap = 0, nb_cap = 0;
        BlockDesc *d_desc = nullptr;
        int32_t *d_status = nullptr;
        int64_t *d_bstart = nullptr;
        void *d_scratch = nullptr; /* the inflate kernels' (spx_bgzf_inflate_scratch_bytes) */
        hipEvent_t ev_inf = nullptr, ev_host = nullptr;
        bool busy = false;
    } slot[kSlots];
    /* two rings of pinned chunks: [0] compressed bytes (uploader thread), [1] host-inflated bytes (helper thread) */
    void *pin[2][kPins] = {};
    hipEvent_t pin_ev[2][kPins] = {};
    bool pin_busy[2][kPins] = {};
    size_t pin_bytes = (size_t)64 << 20;
    int pin_next[2] = {0, 0};
    /* parse-time pools (grow-only) */
    void *d_pool = nullptr, *d_blk = nullptr, *d_temp = nullptr;
    size_t pool_cap = 0, blk_cap = 0, temp_cap = 0;
    int64_t rec_cap = 0, blk_n = 0;
    spx_din_counts *d_counts = nullptr, *h_counts = nullptr;
    spx_din_group_scan *h_g = nullptr; /* pinned scratch for the range prefixes */
    spx_din_slot_scan *h_s = nullptr;
    int32_t *d_tmap = nullptr;
    int32_t n_targets = 0;
    std::thread uploader, helper, parser;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Seg *> host_q, parse_q; /* uploader -> helper (the host pool's share of the inflate) -> parser */
    bool up_done = false, help_done = false;

    int fail_here(const std::string &msg, int code)
    {
        std::lock_guard<std::mutex> lk(D->mu);
        if (D->rc == SPX_OK) { D->rc = code; D->err = msg; }
        D->cv.notify_all();
        return code;
    }
    bool stopping()
    {
        std::lock_guard<std::mutex> lk(D->mu);
        return D->closing || D->rc != SPX_OK;
    }
    int init();
    void destroy();
    int ensure_slot(Slot &S, size_t buf, size_t comp, size_t nb);
    int ensure_pools(int64_t recs, int64_t blocks);
    void carve(spx_din_args &A);
    Seg *cut_segment();
    int upload(Seg *s);
    int host_part(Seg *s);
    int pin_chunk(int ring, char **h);
    void uploader_main();
    void helper_main();
    int parse(Seg *s);
    void parser_main();
};

int Lane::init()
{
    DCHK(hipSetDevice(device));
    DCHK(hipStreamCreateWithFlags(&up_stream, hipStreamNonBlocking));
    DCHK(hipStreamCreateWithFlags(&up2_stream, hipStreamNonBlocking));
    /* the inflate grids (tens of thousands of workgroups of ~10 ms each) fill every CU; the small, latency-bound kernels of
     * the record chain / parse / gather get what frees up FIRST (high priority), the inflate kernel last (low priority:
     * the scoring kernels of the pipeline, at normal priority, go in front of it too) */
    int pr_lo = 0, pr_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi); /* lo = numerically greatest = least urgent */
    if (getenv("SPX_DIN_NO_PRIORITY")) pr_lo = pr_hi = 0;
    DCHK(hipStreamCreateWithPriority(&inf_stream, hipStreamNonBlocking, pr_lo));
    DCHK(hipStreamCreateWithPriority(&in_stream, hipStreamNonBlocking, pr_hi));
    for (int k = 0; k < kSlots; ++k) {
        DCHK(hipEventCreateWithFlags(&slot[k].ev_inf, hipEventDisableTiming | hipEventBlockingSync));
        DCHK(hipEventCreateWithFlags(&slot[k].ev_host, hipEventDisableTiming | hipEventBlockingSync));
    }
    DCHK(hipMalloc((void **)&d_counts, sizeof(spx_din_counts)));
    DCHK(hipHostMalloc((void **)&h_counts, sizeof(spx_din_counts) + 4 * sizeof(spx_din_group_scan) + 4 * sizeof(spx_din_slot_scan), hipHostMallocDefault));
    h_g = (spx_din_group_scan *)(h_counts + 1);
    h_s = (spx_din_slot_scan *)(h_g + 4);
    n_targets = (int32_t)D->tmap.size();
    DCHK(hipMalloc((void **)&d_tmap, sizeof(int32_t) * (size_t)std::max(1, n_targets)));
    if (n_targets) DCHK(hipMemcpy(d_tmap, D->tmap.data(), sizeof(int32_t) * (size_t)n_targets, hipMemcpyHostToDevice));
    return SPX_OK;
}

void Lane::destroy()
{
    (void)hipSetDevice(device);
    if (in_stream) (void)hipStreamSynchronize(in_stream);
    if (inf_stream) (void)hipStreamSynchronize(inf_stream);
    if (up_stream) (void)hipStreamSynchronize(up_stream);
    if (up2_stream) (void)hipStreamSynchronize(up2_stream);
    for (int k = 0; k < kSlots; ++k) {
        Slot &S = slot[k];
        if (S.d_buf) (void)hipFree(S.d_buf);
        if (S.d_comp) (void)hipFree(S.d_comp);
        if (S.d_desc) (void)hipFree(S.d_desc);
        if (S.d_status) (void)hipFree(S.d_status);
        if (S.d_bstart) (void)hipFree(S.d_bstart);
        if (S.d_scratch) (void)hipFree(S.d_scratch);
        if (S.ev_inf) (void)hipEventDestroy(S.ev_inf);
        if (S.ev_host) (void)hipEventDestroy(S.ev_host);
    }
    for (int r = 0; r < 2; ++r)
        for (int k = 0; k < kPins; ++k) {
            if (pin[r][k]) (void)hipHostFree(pin[r][k]);
            if (pin_ev[r][k]) (void)hipEventDestroy(pin_ev[r][k]);
        }
    if (d_pool) (void)hipFree(d_pool);
    if (d_blk) (void)hipFree(d_blk);
    if (d_temp) (void)hipFree(d_temp);
    if (d_counts) (void)hipFree(d_counts);
    if (h_counts) (void)hipHostFree(h_counts);
    if (d_tmap) (void)hipFree(d_tmap);
    if (up_stream) (void)hipStreamDestroy(up_stream);
    if (up2_stream) (void)hipStreamDestroy(up2_stream);
    if (inf_stream) (void)hipStreamDestroy(inf_stream);
    if (in_stream) (void)hipStreamDestroy(in_stream);
}

int Lane::ensure_slot(Slot &S, size_t buf, size_t comp, size_t nb)
{
    auto grow = [&](void **p, size_t *cap, size_t want) -> hipError_t {
        if (want <= *cap) return hipSuccess;
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        *cap = 0;
        const size_t take = want + want / 8 + 4096;
        hipError_t e = hipMalloc(p, take);
        if (e == hipSuccess) *cap = take;
        return e;
    };
    DCHK(grow((void **)&S.d_buf, &S.buf_cap, buf));
    DCHK(grow((void **)&S.d_comp, &S.comp_cap, comp));
    if (nb > S.nb_cap) {
        if (S.d_desc) (void)hipFree(S.d_desc);
        if (S.d_status) (void)hipFree(S.d_status);
        if (S.d_bstart) (void)hipFree(S.d_bstart);
        if (S.d_scratch) (void)hipFree(S.d_scratch);
        S.d_desc = nullptr; S.d_status = nullptr; S.d_bstart = nullptr; S.d_scratch = nullptr;
        const size_t take = nb + nb / 4 + 64;
        DCHK(hipMalloc((void **)&S.d_desc, take * sizeof(BlockDesc)));
        DCHK(hipMalloc((void **)&S.d_status, take * sizeof(int32_t)));
        DCHK(hipMalloc((void **)&S.d_bstart, (take + 1) * sizeof(int64_t)));
        DCHK(hipMalloc(&S.d_scratch, spx_bgzf_inflate_scratch_bytes((int32_t)take) + 64));
        S.nb_cap = take;
    }
    return SPX_OK;
}

/* bytes per record slot of the parse pools: the spx_din_recs arrays, group / slot tables and their scans */
constexpr size_t kPoolPerRec = 3 * 8 + 10 * 4 + 4 /* grp_first */ + sizeof(spx_din_group_scan) + 4 + 4 + sizeof(spx_din_slot_scan);

int Lane::ensure_pools(int64_t recs, int64_t blocks)
{
    if (recs > rec_cap) {
        DCHK(hipStreamSynchronize(in_stream));
        if (d_pool) (void)hipFree(d_pool);
        d_pool = nullptr;
        const int64_t want = std::max<int64_t>((int64_t)1 << 18, recs + recs / 4);
        const size_t bytes = (size_t)(want + 2) * kPoolPerRec + 20 * 256; /* carve() rounds each of its 18 arrays up to 256 bytes */
        DCHK(hipMalloc(&d_pool, bytes));
        pool_cap = bytes;
        rec_cap = want;
        const size_t tb = spx_din_scan_temp_bytes(want + 1);
        if (tb > temp_cap) {
            if (d_temp) (void)hipFree(d_temp);
            d_temp = nullptr;
            DCHK(hipMalloc(&d_temp, tb));
            temp_cap = tb;
        }
    }
    if (blocks > blk_n) {
        DCHK(hipStreamSynchronize(in_stream));
        if (d_blk) (void)hipFree(d_blk);
        d_blk = nullptr;
        const int64_t want = blocks + blocks / 4 + 64;
        DCHK(hipMalloc(&d_blk, (size_t)want * 20 + 1024));
        blk_n = want;
    }
    return SPX_OK;
}

void Lane::carve(spx_din_args &A)
{
    char *p = (char *)d_pool;
    auto take = [&](size_t bytes) { char *at = p; p += (bytes + 255) & ~(size_t)255; return at; };
    const size_t n = (size_t)rec_cap + 2;
    A.R.off = (int64_t *)take(n * 8);
    A.R.cig_at = (int64_t *)take(n * 8);
    A.R.tag_at = (int64_t *)take(n * 8);
    int32_t **i32[] = {&A.R.flag, &A.R.tid, &A.R.pos, &A.R.lq, &A.R.ncig, &A.R.cs_len, &A.R.md_len, &A.R.lname, &A.R.isnew, &A.R.gid};
    for (int32_t **q : i32) *q = (int32_t *)take(n * 4);
    A.grp_first = (int32_t *)take(n * 4);
    A.gscan = (spx_din_group_scan *)take(n * sizeof(spx_din_group_scan));
    A.slot_rec = (int32_t *)take(n * 4);
    A.slot_grp = (int32_t *)take(n * 4);
    A.sscan = (spx_din_slot_scan *)take(n * sizeof(spx_din_slot_scan));
    A.rec_cap = rec_cap;
    char *b = (char *)d_blk;
    A.land = (int64_t *)b;
    A.cnt = (int32_t *)(b + (size_t)blk_n * 8);
    A.bflag = A.cnt + blk_n;
    A.first_idx = A.bflag + blk_n;
    A.tmap = d_tmap;
    A.n_targets = n_targets;
    A.counts = d_counts;
}

/* next run of BGZF blocks (about seg_bytes inflated); nullptr at the end of the chain or on error */
Seg *Lane::cut_segment()
{
    spx_dbam *d = D;
    std::lock_guard<std::mutex> lk(d->cut_mu);
    if (d->cut_eof) return nullptr;
    std::unique_ptr<Seg> s(new Seg());
    s->index = d->next_index.load();
    s->c0 = d->fpos;
    if (s->index == 0) s->p0_extra = d->start_uoff;
    int64_t u = 0;
    /* short segments first (the pipeline behind them fills early), long ones in the steady state (the preparation kernels of a
     * work list are chains of fixed latency: their cost per group falls with the size of the list) */
    int64_t target = d->seg_bytes;
    if (d->seg_first > 0) {
        const int64_t round = s->index / (int64_t)std::max<size_t>(1, d->lanes.size());
        target = round < 20 ? std::min(d->seg_bytes, d->seg_first << round) : d->seg_bytes;
    }
    auto bad = [&](const char *msg) -> Seg * {
        d->cut_eof = true;
        fail_here(msg, SPX_EINVAL);
        return nullptr;
    };
    /* a cap on the blocks of a segment beside the byte target: a BAM written as many tiny BGZF blocks (samtools --write-index on a stream
     * of short flushes, `bgzip -b`) would otherwise grow the block table past the pinned chunk it travels in, and the inflate kernel's
     * scratch (8 KB per block) to gigabytes per slot; the carry buffer takes the records such a short segment cuts through */
    constexpr size_t kMaxSegBlocks = (size_t)1 << 18;
    while (u < target && s->blocks.size() < kMaxSegBlocks) {
        if (d->fpos >= d->fsize) { d->cut_eof = true; break; }
        if (d->fpos == d->end_coff && d->end_uoff == 0) { d->cut_eof = true; break; }
        const uint8_t *p = d->map + d->fpos;
        const int64_t avail = d->fsize - d->fpos;
        if (avail < 18 || p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return bad("not a BGZF block");
        const int64_t xlen = p[10] | (p[11] << 8);
        if (12 + xlen > avail) return bad("truncated BGZF header");
        int bsize = -1;
        for (int64_t o = 0; o + 4 <= xlen;) {
            const uint8_t *e = p + 12 + o;
            const int64_t slen = e[2] | (e[3] << 8);
            if (e[0] == 'B' && e[1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = e[4] | (e[5] << 8);
            o += 4 + slen;
        }
        if (bsize < 0) return bad("BGZF block without BC field");
        const int64_t total = (int64_t)bsize + 1, hl = 12 + xlen;
        if (total < hl + 8) return bad("corrupt BGZF block");
        if (total > avail) return bad("truncated BGZF block");
        const uint32_t crc = le32(p + total - 8), isize = le32(p + total - 4);
        if (isize > 65536) return bad("corrupt BGZF block (ISIZE)");
        const bool stop_here = d->fpos == d->end_coff;
        if (isize > 0) {
            BlockDesc b;
            b.in_off = d->fpos + hl - s->c0;
            b.out_off = u;
            b.clen = (uint32_t)(total - hl - 8);
            b.ulen = isize;
            b.crc = crc;
            b.pad = 0;
            s->blocks.push_back(b);
        }
        if (stop_here) {
            s->stop_at = u + std::min<uint32_t>((uint32_t)d->end_uoff, isize);
            d->cut_eof = true;
        }
        u += isize;
        d->fpos += total;
        if (stop_here) break;
    }
    s->c1 = d->fpos;
    s->ulen = u;
    s->last = d->cut_eof;
    s->bstart.resize(s->blocks.size() + 1);
    for (size_t k = 0; k < s->blocks.size(); ++k) s->bstart[k] = d->carry_cap + s->blocks[k].out_off;
    s->bstart[s->blocks.size()] = d->carry_cap + (s->stop_at >= 0 ? s->stop_at : u);
    /* (a shard end inside a block: the blocks behind the end are not there, the last block's tail is cut off by n_end) */
    ++d->next_index;
    if (s->last) {
        std::lock_guard<std::mutex> lk2(d->mu);
        d->total_segments = d->next_index.load();
        d->cv.notify_all();
    }
    s->t_cut = now_s();
    return s.release();
}

int Lane::upload(Seg *s)
{
    spx_dbam *d = D;
    /* a free segment buffer */
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] {
            for (int k = 0; k < kSlots; ++k) if (!slot[k].busy) return true;
            return stopping();
        });
        if (stopping()) return SPX_EINVAL;
        for (int k = 0; k < kSlots; ++k) if (!slot[k].busy) { s->slot = k; slot[k].busy = true; break; }
    }
    Slot &S = slot[s->slot];
    DCHK(hipSetDevice(device));
    const size_t nb = s->blocks.size();
    /* The host pool helps: the LAST blocks of the segment (a share of its inflated bytes) are inflated on the host into pinned
     * memory and uploaded raw; the device inflates the first ones, whose compressed bytes go up first.  (One MI355X inflates
     * ~13 GB/s beside nothing else; 16 host cores ~9 GB/s: neither alone feeds the scoring kernels.) */
    size_t nd = nb;
    if (d->host_share > 0 && nb > 1) {
        const int64_t want_host = (int64_t)(d->host_share * (double)s->ulen);
        while (nd > 0 && s->ulen - s->blocks[nd - 1].out_off <= want_host) --nd;
        if (nd == 0 && d->host_share < 1.0) nd = 1;
    }
    s->n_dev_blocks = nd;
    const size_t comp = nd == nb ? (size_t)(s->c1 - s->c0) : (nd == 0 ? 0 : (size_t)(s->blocks[nd - 1].in_off + s->blocks[nd - 1].clen + 8));
    /* sized for the LARGEST segment from the start: growing a buffer later means hipFree, which waits for the whole device */
    const size_t full = (size_t)(d->seg_bytes + 65536);
    int rc = ensure_slot(S, (size_t)d->carry_cap + std::max((size_t)s->ulen, full) + 256, std::max(comp, d->seg_first > 0 ? full / 4 * 3 : (size_t)0) + 256, nb + 1);
    if (rc) return rc;
    const double t0 = now_s();
    /* compressed bytes through the ring of pinned chunks; the copies run on the reader's pool */
    struct Cp { char *dst; const uint8_t *src; };
    for (size_t o = 0; o < comp; o += pin_bytes) {
        char *h = nullptr;
        const int k = pin_chunk(0, &h);
        if (k < 0) return fail_here("pinned memory for the compressed bytes", SPX_ENOMEM);
        const size_t n = std::min(pin_bytes, comp - o);
        Cp cp{h, d->map + s->c0 + o};
        {
            spx::CpuScope cs(spx::CPU_FILL);
            spx_internal_bam_parallel(d->hdr, (int64_t)n, (int64_t)4 << 20, [](void *u, int64_t a, int64_t b) {
                Cp *c = (Cp *)u;
                memcpy(c->dst + a, c->src + a, (size_t)(b - a));
            }, &cp);
        }
        DCHK(hipMemcpyAsync(S.d_comp + o, h, n, hipMemcpyHostToDevice, up_stream));
        DCHK(hipEventRecord(pin_ev[0][k], up_stream));
        pin_busy[0][k] = true;
    }
    { /* block table + block starts, through the ring as well */
        char *h = nullptr;
        const int k = pin_chunk(0, &h);
        if (k < 0) return fail_here("pinned memory for the block table", SPX_ENOMEM);
        const size_t b1 = nd * sizeof(BlockDesc), b2 = (nb + 1) * sizeof(int64_t);
        if (b1 + b2 > pin_bytes) return fail_here("segment has too many blocks", SPX_EINVAL);
        memcpy(h, s->blocks.data(), b1);
        memcpy(h + b1, s->bstart.data(), b2);
        if (nd) DCHK(hipMemcpyAsync(S.d_desc, h, b1, hipMemcpyHostToDevice, up_stream));
        DCHK(hipMemcpyAsync(S.d_bstart, h + b1, b2, hipMemcpyHostToDevice, up_stream));
        DCHK(hipEventRecord(pin_ev[0][k], up_stream));
        pin_busy[0][k] = true;
        /* the inflate stream goes on behind the last copy */
        DCHK(hipStreamWaitEvent(inf_stream, pin_ev[0][k], 0));
    }
    if (nd) DCHK(spx_launch_bgzf_inflate2(S.d_comp, S.d_desc, (int32_t)nd, S.d_buf + d->carry_cap, S.d_status, d->check_crc, S.d_scratch, inf_stream));
    DCHK(hipEventRecord(S.ev_inf, inf_stream));
    s->t_up = now_s();
    {
        std::lock_guard<std::mutex> lk(d->mu);
        d->bytes_up += (int64_t)comp;
        d->t_upload += s->t_up - t0;
        ++d->n_segments;
    }
    return SPX_OK;
}

int Lane::pin_chunk(int ring, char **h)
{
    const int k = pin_next[ring];
    pin_next[ring] = (k + 1) % kPins;
    if (!pin[ring][k]) {
        if (hipHostMalloc(&pin[ring][k], pin_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pin[ring][k] = nullptr; return -1; }
        if (hipEventCreateWithFlags(&pin_ev[ring][k], hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) return -1;
    }
    if (pin_busy[ring][k] && hipEventSynchronize(pin_ev[ring][k]) != hipSuccess) return -1;
    pin_busy[ring][k] = false;
    *h = (char *)pin[ring][k];
    return k;
}

/* the host's share of a segment (helper thread): runs of blocks whose inflated bytes fill a pinned chunk, inflated on the pool, uploaded raw */
int Lane::host_part(Seg *s)
{
    spx_dbam *d = D;
    Slot &S = slot[s->slot];
    DCHK(hipSetDevice(device));
    const size_t nb = s->blocks.size(), nd = s->n_dev_blocks;
    int64_t host_bytes = 0;
    for (size_t q0 = nd; q0 < nb;) {
        char *h = nullptr;
        const int k = pin_chunk(1, &h);
        if (k < 0) return fail_here("pinned memory for the host-inflated bytes", SPX_ENOMEM);
        const int64_t base = s->blocks[q0].out_off;
        size_t q1 = q0;
        while (q1 < nb && s->blocks[q1].out_off + (int64_t)s->blocks[q1].ulen - base <= (int64_t)pin_bytes) ++q1;
        if (q1 == q0) return fail_here("pinned chunk smaller than a BGZF block", SPX_EINVAL);
        struct HI { Seg *s; const uint8_t *file; char *dst; int64_t base; size_t q0; int check; std::atomic<int> bad{0}; } hi;
        hi.s = s; hi.file = d->map + s->c0; hi.dst = h; hi.base = base; hi.q0 = q0; hi.check = d->check_crc;
        spx_internal_bam_parallel(d->hdr, (int64_t)(q1 - q0), 8, [](void *u, int64_t a, int64_t b) {
            HI *x = (HI *)u;
            for (int64_t q = a; q < b; ++q) {
                const BlockDesc &bd = x->s->blocks[x->q0 + (size_t)q];
                const int r = spx_internal_inflate_block(x->file + bd.in_off, bd.clen, (uint8_t *)x->dst + (bd.out_off - x->base), bd.ulen, bd.crc, x->check);
                if (r) x->bad = r;
            }
        }, &hi);
        if (hi.bad.load()) return fail_here(hi.bad.load() == 2 ? "BGZF block CRC mismatch" : "inflate failed", SPX_EINVAL);
        const int64_t n = s->blocks[q1 - 1].out_off + (int64_t)s->blocks[q1 - 1].ulen - base;
        DCHK(hipMemcpyAsync(S.d_buf + d->carry_cap + base, h, (size_t)n, hipMemcpyHostToDevice, up2_stream));
        DCHK(hipEventRecord(pin_ev[1][k], up2_stream));
        pin_busy[1][k] = true;
        host_bytes += n;
        q0 = q1;
    }
    DCHK(hipEventRecord(S.ev_host, up2_stream));
    {
        std::lock_guard<std::mutex> lk(d->mu);
        d->bytes_up += host_bytes;
        d->bytes_host_inflated += host_bytes;
        d->bytes_dev_inflated += s->ulen - host_bytes;
    }
    return SPX_OK;
}

void Lane::helper_main()
{
    for (;;) {
        Seg *s = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !host_q.empty() || up_done; });
            if (host_q.empty()) break;
            s = host_q.front();
            host_q.pop_front();
        }
        const int rc = stopping() ? SPX_EINVAL : host_part(s);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (rc == SPX_OK) parse_q.push_back(s);
            else {
                if (s->slot >= 0) slot[s->slot].busy = false;
                delete s;
            }
        }
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        help_done = true;
    }
    cv.notify_all();
}

void Lane::uploader_main()
{
    for (;;) {
        if (stopping()) break;
        /* not more than `ahead` finished segments wait for the consumer */
        {
            std::unique_lock<std::mutex> lk(D->mu);
            D->cv.wait(lk, [&] { return D->closing || D->rc != SPX_OK || D->next_index.load() - D->next_out < (int64_t)D->ahead * (int64_t)D->lanes.size() + kSlots; });
            if (D->closing || D->rc != SPX_OK) break;
        }
        Seg *s = cut_segment();
        if (!s) break;
        const int rc = upload(s);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (rc == SPX_OK) host_q.push_back(s);
            else {
                if (s->slot >= 0) slot[s->slot].busy = false;
                delete s;
            }
        }
        cv.notify_all();
        if (rc != SPX_OK) break;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        up_done = true;
    }
    cv.notify_all();
}

NameBatch *names_get(spx_dbam *d, size_t bytes)
{
    {
        std::lock_guard<std::mutex> lk(d->mu);
        for (size_t k = 0; k < d->free_names.size(); ++k)
            if (d->free_names[k]->cap >= bytes) {
                NameBatch *n = d->free_names[k];
                d->free_names.erase(d->free_names.begin() + (long)k);
                return n;
            }
    }
    NameBatch *n = new NameBatch();
    n->cap = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&n->pinned, n->cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); delete n; return nullptr; }
    return n;
}

int Lane::parse(Seg *s)
{
    spx_dbam *d = D;
    DCHK(hipSetDevice(device));
    Slot &S = slot[s->slot];
    const double t0 = now_s();
    /* the carry of the segment in front */
    int64_t clen = 0;
    {
        std::unique_lock<std::mutex> lk(d->mu);
        d->cv.wait(lk, [&] { return d->closing || d->rc != SPX_OK || d->carry_for == s->index; });
        if (d->closing || d->rc != SPX_OK) return SPX_EINVAL;
        clen = d->carry_len;
    }
    const double t1 = now_s();
    if (clen > 0) {
        DCHK(hipMemcpyAsync(S.d_buf + d->carry_cap - clen, d->carry_host, (size_t)clen, hipMemcpyHostToDevice, in_stream));
        DCHK(hipStreamSynchronize(in_stream)); /* the buffer is written again by this very segment's hand-over */
    }
    DCHK(hipStreamWaitEvent(in_stream, S.ev_inf, 0));
    DCHK(hipStreamWaitEvent(in_stream, S.ev_host, 0));
    const int64_t nb = (int64_t)s->blocks.size();
    int rc = ensure_pools(std::max<int64_t>(rec_cap, 1), nb + 1);
    if (rc) return rc;
    spx_din_args A;
    memset(&A, 0, sizeof A);
    A.buf = S.d_buf;
    A.p0 = d->carry_cap - clen + s->p0_extra;
    A.n_end = d->carry_cap + (s->stop_at >= 0 ? s->stop_at : s->ulen);
    A.bstart = S.d_bstart;
    A.n_blocks = (int32_t)nb;
    A.is_final = s->last ? 1 : 0;
    A.max_rec = d->carry_cap;
    carve(A);
    DCHK(hipMemsetAsync(d_counts, 0, sizeof(spx_din_counts), in_stream));
    DCHK(spx_din_inflate_status(S.d_status, (int32_t)s->n_dev_blocks, d_counts, in_stream));
    spx_din_counts &C = *h_counts;
    double t_inf = 0;
    for (int attempt = 0;; ++attempt) {
        DCHK(spx_din_chain(&A, in_stream));
        DCHK(hipMemcpyAsync(h_counts, d_counts, sizeof(spx_din_counts), hipMemcpyDeviceToHost, in_stream));
        if (attempt == 0) {
            const double tw = now_s();
            DCHK(hipEventSynchronize(S.ev_inf));
            DCHK(hipEventSynchronize(S.ev_host));
            t_inf = now_s() - tw;
        }
        DCHK(hipStreamSynchronize(in_stream));
        if (C.inflate_bad) return fail_here(C.inflate_bad == 4 ? "BGZF block CRC mismatch" : "inflate failed", SPX_EINVAL);
        if (C.err == 1) return fail_here("corrupt BAM record", SPX_EINVAL);
        if (C.err) return fail_here("BAM record chain could not be followed", SPX_EINVAL);
        if (C.n_rec <= rec_cap) break;
        if (attempt >= 2) return fail_here("record table keeps overflowing", SPX_ENOMEM);
        if ((rc = ensure_pools(C.n_rec, nb + 1))) return rc;
        carve(A);
    }
    if (s->last && C.tail_start != A.n_end) return fail_here("truncated BAM record", SPX_EINVAL);
    const int64_t n_rec = C.n_rec;
    const double t2 = now_s();
    DCHK(spx_din_groups(&A, n_rec, d_temp, temp_cap, in_stream));
    DCHK(hipMemcpyAsync(h_counts, d_counts, sizeof(spx_din_counts), hipMemcpyDeviceToHost, in_stream));
    DCHK(hipStreamSynchronize(in_stream));
    if (C.err == 3) return fail_here("corrupt BAM record (field lengths exceed the record)", SPX_EINVAL);
    /* hand the open group (and the front of a record that continues) to the next segment */
    if (!s->last) {
        const int64_t cl2 = A.n_end - C.carry_start;
        if (cl2 > d->carry_cap || cl2 < 0)
            return fail_here("a read group (or a record) is larger than the carry buffer of the device input (SPX_DIN_CARRY_MB); use --hostInput", SPX_EUNSUPPORTED);
        if (cl2 > 0) {
            if ((size_t)cl2 + 64 > d->carry_host_cap) { /* (this segment has consumed its own carry: nobody reads the buffer now) */
                (void)hipHostFree(d->carry_host);
                d->carry_host = nullptr;
                d->carry_host_cap = std::min<size_t>((size_t)d->carry_cap, (size_t)cl2 * 2) + 64;
                if (hipHostMalloc((void **)&d->carry_host, d->carry_host_cap, hipHostMallocDefault) != hipSuccess) {
                    (void)hipGetLastError();
                    d->carry_host_cap = 0;
                    return fail_here("pinned memory for the carry buffer", SPX_ENOMEM);
                }
            }
            DCHK(hipMemcpyAsync(d->carry_host, S.d_buf + C.carry_start, (size_t)cl2, hipMemcpyDeviceToHost, in_stream));
            DCHK(hipStreamSynchronize(in_stream));
        }
        std::lock_guard<std::mutex> lk(d->mu);
        d->carry_len = cl2;
        d->carry_for = s->index + 1;
        d->cv.notify_all();
    } else {
        std::lock_guard<std::mutex> lk(d->mu);
        d->carry_len = 0;
        d->carry_for = s->index + 1;
        d->cv.notify_all();
    }
    const double t3 = now_s();
    /* ---- the segment's complete groups as work lists ---- */
    std::vector<Item> items;
    const int64_t n_batch = C.n_batch;
    const spx_din_counts Cs = C; /* (h_counts is reused below) */
    std::vector<int32_t> gf_host; /* only needed when the segment is split */
    /* a list that fails half-way leaves nothing behind: the lists made so far (their device memory) and the name blocks go back */
    auto drop_items = [&] {
        for (Item &x : items) {
            if (x.work) spx_work_free(ctx, x.work);
            if (x.names) {
                std::lock_guard<std::mutex> lk(d->mu);
                d->free_names.push_back(x.names);
            }
        }
        items.clear();
    };
    auto make_list = [&](int64_t g0, int64_t g1) -> int {
        spx_din_range Q;
        memset(&Q, 0, sizeof Q);
        Q.g0 = g0; Q.g1 = g1;
        if (g0 == 0 && g1 == n_batch) {
            Q.r0 = 0; Q.r1 = Cs.n_batch_rec; Q.s0 = 0; Q.s1 = Cs.n_slots;
            Q.gend.disp = Cs.n_dgroups; Q.gend.slots = Cs.n_slots; Q.gend.name_bytes = Cs.name_bytes;
        } else {
            if (gf_host.empty()) {
                gf_host.resize((size_t)n_batch + 1);
                DCHK(hipMemcpyAsync(gf_host.data(), A.grp_first, sizeof(int32_t) * ((size_t)n_batch + 1), hipMemcpyDeviceToHost, in_stream));
                DCHK(hipStreamSynchronize(in_stream));
            }
            DCHK(hipMemcpyAsync(&h_g[0], A.gscan + g0, sizeof(spx_din_group_scan), hipMemcpyDeviceToHost, in_stream));
            DCHK(hipMemcpyAsync(&h_g[1], A.gscan + g1, sizeof(spx_din_group_scan), hipMemcpyDeviceToHost, in_stream));
            DCHK(hipStreamSynchronize(in_stream));
            Q.gbase = h_g[0]; Q.gend = h_g[1];
            Q.r0 = gf_host[(size_t)g0]; Q.r1 = gf_host[(size_t)g1];
            Q.s0 = Q.gbase.slots; Q.s1 = Q.gend.slots;
            DCHK(hipMemcpyAsync(&h_s[0], A.sscan + Q.s0, sizeof(spx_din_slot_scan), hipMemcpyDeviceToHost, in_stream));
            DCHK(hipMemcpyAsync(&h_s[1], A.sscan + Q.s1, sizeof(spx_din_slot_scan), hipMemcpyDeviceToHost, in_stream));
            DCHK(hipStreamSynchronize(in_stream));
            Q.sbase = h_s[0];
        }
        spx_din_slot_scan send;
        if (g0 == 0 && g1 == n_batch) {
            send.cw = Cs.cigar_words; send.sb = Cs.seq_bytes; send.qb = Cs.qual_bytes; send.tb = Cs.text_bytes;
            send.oc = Cs.ops_bound; send.cc = Cs.conf_bound; send.mc = Cs.mm_bound;
        } else
            send = h_s[1];
        const int64_t ng = g1 - g0, nr = Q.r1 - Q.r0, name_bytes = Q.gend.name_bytes - Q.gbase.name_bytes;
        /* the host-bound block: [grp_first | name_off | tid | pos | flag | disp | names] */
        size_t o = 0;
        auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
        const size_t o_gf = take(((size_t)ng + 1) * 4), o_no = take((size_t)ng * 8), o_tid = take((size_t)nr * 4), o_pos = take((size_t)nr * 4),
                     o_flag = take((size_t)nr * 2), o_disp = take((size_t)ng), o_names = take((size_t)name_bytes + 1);
        spx_devstage_sizes sz;
        sz.n_groups_in = ng;
        sz.n_dgroups = Q.gend.disp - Q.gbase.disp;
        sz.n_slots = Q.s1 - Q.s0;
        sz.cigar_words = send.cw - Q.sbase.cw; sz.seq_bytes = send.sb - Q.sbase.sb; sz.qual_bytes = send.qb - Q.sbase.qb;
        sz.text_bytes = send.tb - Q.sbase.tb; sz.ops_bound = send.oc - Q.sbase.oc; sz.conf_bound = send.cc - Q.sbase.cc;
        sz.mm_bound = send.mc - Q.sbase.mc;
        sz.info_bytes = (int64_t)o;
        spx_work *w = nullptr;
        spx_din_out O;
        char *d_info = nullptr;
        if ((rc = spx_internal_devstage_begin(ctx, &d->par, &sz, &w, &O, &d_info)) != SPX_OK) return fail_here(spx_last_error(), rc);
        Item it;
        it.work = w;
        it.lane = index;
        it.n_groups = (int32_t)ng;
        items.push_back(it);
        O.h_grp_first = (int32_t *)(d_info + o_gf);
        O.name_off = (int64_t *)(d_info + o_no);
        O.h_tid = (int32_t *)(d_info + o_tid);
        O.h_pos = (int32_t *)(d_info + o_pos);
        O.h_flag = (uint16_t *)(d_info + o_flag);
        O.grp_disp = (uint8_t *)(d_info + o_disp);
        O.names = d_info + o_names;
        NameBatch *nbh = names_get(d, o + 16);
        if (!nbh) return fail_here("pinned memory for the group names", SPX_ENOMEM);
        items.back().names = nbh;
        hipError_t e = spx_din_image(&A, &O, &Q, in_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nbh->pinned, d_info, o, hipMemcpyDeviceToHost, in_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(in_stream);
        if (e != hipSuccess) return fail_here(std::string("device input kernels: ") + hipGetErrorString(e), SPX_EHIP);
        char *hp = (char *)nbh->pinned;
        if ((rc = spx_internal_devstage_finish(ctx, w, (const uint8_t *)(hp + o_disp), in_stream)) != SPX_OK) return fail_here(spx_last_error(), rc);
        spx_batch &b = nbh->view;
        memset(&b, 0, sizeof b);
        b.n_groups = (int32_t)ng;
        b.n_alns = (int32_t)nr;
        b.grp_first = (const int32_t *)(hp + o_gf);
        b.qname_off = (const int64_t *)(hp + o_no);
        b.qnames = hp + o_names;
        b.flag = (const uint16_t *)(hp + o_flag);
        b.tid = (const int32_t *)(hp + o_tid);
        b.pos = (const int32_t *)(hp + o_pos);
        return SPX_OK;
    };
    for (int64_t g0 = 0; g0 < n_batch;) {
        const int64_t g1 = std::min<int64_t>(n_batch, g0 + d->max_groups);
        if ((rc = make_list(g0, g1)) != SPX_OK) {
            drop_items();
            return rc;
        }
        g0 = g1;
    }
    /* the buffer may be written again once the image kernels have run (they have: every range ended with a synchronize) */
    {
        std::lock_guard<std::mutex> lk(mu);
        S.busy = false;
    }
    cv.notify_all();
    const double t4 = now_s();
    {
        std::lock_guard<std::mutex> lk(d->mu);
        for (Item &x : items) d->live_names.push_back(x.names);
        d->results[s->index] = std::move(items);
        d->t_wait_carry += t1 - t0;
        d->t_wait_inflate += t_inf;
        d->t_chain += t2 - t1 - t_inf;
        d->t_groups += t3 - t2;
        d->t_image += t4 - t3;
        d->t_parse += t4 - t0;
        d->cv.notify_all();
    }
    if (timing_on())
        fprintf(stderr, "[spx timing] device input: segment %lld on lane %d: %zu blocks, %.1f MB inflated, %lld records, %lld groups (%lld dispatched); cut->uploaded %.3f s, "
                        "waited for the carry %.3f, for inflate %.3f, chain %.3f, fields+groups %.3f, image %.3f\n",
                (long long)s->index, index, s->blocks.size(), s->ulen / 1e6, (long long)n_rec, (long long)n_batch, (long long)Cs.n_dgroups, s->t_up - s->t_cut, t1 - t0, t_inf,
                t2 - t1 - t_inf, t3 - t2, t4 - t3);
    return SPX_OK;
}

void Lane::parser_main()
{
    for (;;) {
        Seg *s = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !parse_q.empty() || help_done; });
            if (parse_q.empty()) break;
            s = parse_q.front();
            parse_q.pop_front();
        }
        int rc = stopping() ? SPX_EINVAL : parse(s);
        if (rc != SPX_OK) {
            std::lock_guard<std::mutex> lk(mu);
            if (s->slot >= 0) slot[s->slot].busy = false;
        }
        delete s;
        cv.notify_all();
        if (rc != SPX_OK) {
            /* let the uploader and everybody who waits for a carry see the error (fail_here has set it, or we are closing) */
            std::lock_guard<std::mutex> lk(D->mu);
            D->cv.notify_all();
        }
    }
}

} // namespace

extern "C" void spx_dbam_default_options(spx_dbam_options *o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->threads = 4;
    o->max_groups = 95000;
    o->ahead = 3;
    o->start_voffset = -1;
    o->end_voffset = -1;
    o->host_inflate_percent = -1;
}

extern "C" int spx_dbam_open(const char *path, const spx_dbam_options *opt, spx_dbam **out)
{
    if (!path || !out) return SPX_EINVAL;
    *out = nullptr;
    spx_dbam_options o;
    if (opt) o = *opt; else spx_dbam_default_options(&o);
    spx_bam_options bo;
    spx_bam_default_options(&bo);
    bo.threads = o.threads > 0 ? o.threads : 4;
    bo.flags = SPX_BAM_HEADER_ONLY | (o.flags & SPX_BAM_NO_CRC);
    bo.start_voffset = o.start_voffset;
    bo.end_voffset = o.end_voffset;
    spx_bam_reader *hdr = nullptr;
    int rc = spx_bam_open_opts(path, &bo, &hdr);
    if (rc != SPX_OK) { spx_internal_set_error(spx_io_last_error()); return rc; }
    spx_dbam *d = new spx_dbam();
    d->hdr = hdr;
    spx_internal_bam_layout(hdr, &d->map, &d->fsize, &d->start_coff, &d->start_uoff, &d->end_coff, &d->end_uoff, &d->check_crc);
    d->fpos = d->start_coff;
    d->max_groups = std::max(1, std::min(o.max_groups > 0 ? o.max_groups : 95000, 95000));
    d->ahead = o.ahead > 0 ? o.ahead : 3;
    if (o.segment_bytes > 0) d->seg_bytes = o.segment_bytes;
    if (const char *e = getenv("SPX_DIN_SEG_MB")) d->seg_bytes = (int64_t)atoll(e) << 20;
    if (const char *e = getenv("SPX_DIN_SEG_KB")) d->seg_bytes = (int64_t)atoll(e) << 10;
    d->seg_bytes = std::max<int64_t>(d->seg_bytes, 65536);
    if (const char *e = getenv("SPX_DIN_SEG_FIRST_MB")) d->seg_first = (int64_t)atoll(e) << 20;
    if (o.carry_bytes > 0) d->carry_cap = o.carry_bytes;
    if (const char *e = getenv("SPX_DIN_CARRY_MB")) d->carry_cap = (int64_t)atoll(e) << 20;
    if (const char *e = getenv("SPX_DIN_CARRY_KB")) d->carry_cap = (int64_t)atoll(e) << 10;
    d->carry_cap = (std::max<int64_t>(d->carry_cap, 65536) + 255) & ~(int64_t)255;
    d->host_share = o.host_inflate_percent < 0 ? -1.0 : o.host_inflate_percent / 100.0;
    if (const char *e = getenv("SPX_DIN_HOST_PCT")) d->host_share = atof(e) / 100.0;
    memset(&d->par, 0, sizeof d->par);
    *out = d;
    return SPX_OK;
}

extern "C" spx_bam_reader *spx_dbam_header(spx_dbam *d) { return d ? d->hdr : nullptr; }

extern "C" int spx_dbam_start(spx_dbam *d, spx_ctx *const *ctxs, int32_t n_ctx, const spx_params *par)
{
    if (!d || !ctxs || n_ctx < 1 || !par || !d->lanes.empty()) return SPX_EINVAL;
    d->par = *par;
    if (d->host_share < 0) {
        /* default: the host pool as a HELPER.  Measured on the MI355X boxes (16 cores of CPU time): libdeflate ~0.55 GB/s per core beside
         * the pipelines' own threads; the inflate kernels 44 GB/s alone on a device, about half of that beside the scoring kernels.  A
         * sweep of the share on 524 288 HiFi groups (loop time): 0 % 1.49 s, 15-39 % 1.32-1.47 s with nothing to choose between them
         * -- so the share is set where the host's CPU time stays under 8 core-s per 262 144 groups: host / (host + 40 GB/s per device),
         * 17 % on a 16-core quota with one device */
        const double host_rate = 0.55 * (double)std::max(1, spx_effective_cpus() - 1), dev_rate = 40.0 * n_ctx;
        d->host_share = host_rate / (host_rate + dev_rate);
    }
    d->host_share = std::min(1.0, std::max(0.0, d->host_share));
    d->tmap.resize(1 << 20);
    const int32_t nt = spx_internal_bam_tmap(d->hdr, d->tmap.data(), (int32_t)d->tmap.size());
    if (nt > (int32_t)d->tmap.size()) { d->tmap.resize((size_t)nt); spx_internal_bam_tmap(d->hdr, d->tmap.data(), nt); }
    d->tmap.resize((size_t)nt);
    /* (the carry between two segments is a few hundred KB -- one open name group; the device buffers reserve carry_cap for it, the pinned
     * host buffer it travels through starts small and grows when a hand-over needs more: pinning 256 MB up front was 0.1 s of start-up) */
    d->carry_host_cap = std::min<size_t>((size_t)d->carry_cap, (size_t)4 << 20) + 64;
    if (hipHostMalloc((void **)&d->carry_host, d->carry_host_cap, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        spx_internal_set_error("pinned memory for the carry buffer");
        return SPX_ENOMEM;
    }
    for (int32_t k = 0; k < n_ctx; ++k) {
        Lane *L = new Lane();
        L->D = d;
        L->ctx = ctxs[k];
        L->index = k;
        L->device = spx_internal_ctx_device(ctxs[k]);
        d->lanes.push_back(L);
        if (L->init() != SPX_OK) { spx_internal_set_error(d->err.c_str()); return d->rc; }
    }
    for (Lane *L : d->lanes) {
        L->uploader = std::thread([L] { L->uploader_main(); });
        L->helper = std::thread([L] { L->helper_main(); });
        L->parser = std::thread([L] { L->parser_main(); });
    }
    return SPX_OK;
}

extern "C" int spx_dbam_next(spx_dbam *d, spx_work **work, int32_t *ctx_index, const spx_batch **names)
{
    if (!d || !work || !ctx_index || !names) return SPX_EINVAL;
    *work = nullptr;
    *names = nullptr;
    std::unique_lock<std::mutex> lk(d->mu);
    for (;;) {
        if (!d->out_items.empty()) {
            Item it = d->out_items.front();
            d->out_items.pop_front();
            *work = it.work;
            *ctx_index = it.lane;
            *names = &it.names->view;
            return it.n_groups;
        }
        /* complete segments in front of a failing one are handed out first, in file order: the relabel list then ends where the host
         * reader's would on the same damaged file (the reference scores every group in front of the bad record) */
        auto f = d->results.find(d->next_out);
        if (f != d->results.end()) {
            for (Item &x : f->second) d->out_items.push_back(x);
            d->results.erase(f);
            ++d->next_out;
            d->cv.notify_all();
            continue;
        }
        if (d->rc != SPX_OK) { spx_internal_set_error(d->err.c_str()); return d->rc; }
        if (d->total_segments >= 0 && d->next_out >= d->total_segments) return 0;
        d->cv.wait(lk);
    }
}

extern "C" int spx_dbam_release(spx_dbam *d, const spx_batch *names)
{
    if (!d || !names) return SPX_EINVAL;
    std::lock_guard<std::mutex> lk(d->mu);
    for (size_t k = 0; k < d->live_names.size(); ++k)
        if (&d->live_names[k]->view == names) {
            d->free_names.push_back(d->live_names[k]);
            d->live_names.erase(d->live_names.begin() + (long)k);
            return SPX_OK;
        }
    return SPX_EINVAL;
}

extern "C" void spx_dbam_stats(const spx_dbam *dc, int64_t *segments, int64_t *bytes_uploaded, double *seconds /* 7: upload, parse, wait carry, wait inflate, chain, fields+groups, image */)
{
    spx_dbam *d = const_cast<spx_dbam *>(dc);
    if (!d) return;
    std::lock_guard<std::mutex> lk(d->mu);
    if (segments) *segments = d->n_segments;
    if (bytes_uploaded) *bytes_uploaded = d->bytes_up;
    if (segments && getenv("SPX_TIMING"))
        fprintf(stderr, "[spx timing] device input: %.2f GB inflated on the device(s), %.2f GB on the host pool (share %.0f %%)\n", d->bytes_dev_inflated / 1e9,
                d->bytes_host_inflated / 1e9, 100.0 * d->host_share);
    if (seconds) {
        seconds[0] = d->t_upload; seconds[1] = d->t_parse; seconds[2] = d->t_wait_carry; seconds[3] = d->t_wait_inflate;
        seconds[4] = d->t_chain; seconds[5] = d->t_groups; seconds[6] = d->t_image;
    }
}

extern "C" void spx_dbam_close(spx_dbam *d)
{
    if (!d) return;
    {
        std::lock_guard<std::mutex> lk(d->mu);
        d->closing = true;
    }
    d->cv.notify_all();
    for (Lane *L : d->lanes) {
        { std::lock_guard<std::mutex> lk(L->mu); }
        L->cv.notify_all();
    }
    for (Lane *L : d->lanes) {
        if (L->uploader.joinable()) L->uploader.join();
        L->cv.notify_all();
        if (L->helper.joinable()) L->helper.join();
        L->cv.notify_all();
        if (L->parser.joinable()) L->parser.join();
    }
    /* work lists nobody took */
    for (auto &kv : d->results)
        for (Item &x : kv.second) spx_work_free(d->lanes[(size_t)x.lane]->ctx, x.work);
    for (Item &x : d->out_items) spx_work_free(d->lanes[(size_t)x.lane]->ctx, x.work);
    for (Lane *L : d->lanes) {
        for (Seg *s : L->host_q) delete s;
        for (Seg *s : L->parse_q) delete s;
        L->destroy();
        delete L;
    }
    for (NameBatch *n : d->free_names) { (void)hipHostFree(n->pinned); delete n; }
    for (NameBatch *n : d->live_names) { (void)hipHostFree(n->pinned); delete n; }
    if (d->carry_host) (void)hipHostFree(d->carry_host);
    spx_bam_close(d->hdr);
    delete d;
}
