/*
 * spx_runtime.cpp -- C-ABI (include/spx.h) over the HIP kernels: device
 * context, HBM-resident reference, batch upload, launches, collection,
 * decision replay and the relabel-list writer.
 *
 * The product path has no CPU fallback: every scoring entry point needs a
 * live gfx950 device and fails with SPX_ENODEVICE / SPX_EHIP otherwise.
 */
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spx.h"
#include "spx_device.h"
#include "spx_prep.h"

struct spx_bedset;

extern "C" hipError_t spx_launch_baq(int cls, int phase, const spx_dev_batch *B, hipStream_t st);
extern "C" hipError_t spx_launch_score(const spx_dev_groups *Gd, int32_t n_markers, uint8_t *posmin, hipStream_t st);
extern "C" hipError_t spx_launch_map(const spx_dev_batch *B, int32_t n_rows_total, int wide, hipStream_t st);
extern "C" hipError_t spx_launch_pack(const spx_dev_groups *Gd, const int32_t *grp_index, int32_t group_base,
                                      unsigned long long *out, hipStream_t st);

static thread_local std::string g_err;
static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define HIPCHK(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail(SPX_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

static bool timing_on()
{
    static const bool on = getenv("SPX_TIMING") != nullptr; /* diagnostics: phase times of the host side on stderr */
    return on;
}

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static const unsigned char kNt16Table[256] = {
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 1,  2,  4,  8,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 0,  15, 15, 15, 1,  14, 2,  13, 15, 15, 4,  11, 15, 15, 12, 15, 3,
    15, 15, 15, 15, 5,  6,  8,  15, 7,  9,  15, 10, 15, 15, 15, 15, 15, 15, 15, 1,  14, 2,  13, 15, 15, 4,
    11, 15, 15, 12, 15, 3,  15, 15, 15, 15, 5,  6,  8,  15, 7,  9,  15, 10, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15};
static const unsigned char kNt16Int[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};

struct spx_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    /* HIP events around the kernels of the last SPX_EV_RING launches (0 start, 1 after MAP, 2 after scoring, 3..5 around
     * the main class' forward and backward kernels): spx_collect averages over the launches since the previous collect */
    static const int SPX_EV_RING = 64;
    hipEvent_t evr[SPX_EV_RING][6] = {};
    int64_t n_launch = 0;
    /* the band classes run concurrently: a handful of wide-band problems must not serialise behind
     * (or in front of) the bulk class */
    hipStream_t cls_stream[SPX_N_CLASSES] = {};
    hipEvent_t cls_done[SPX_N_CLASSES] = {};
    uint8_t *d_ref4 = nullptr;
    int64_t ref_bytes = 0;
    spx::RefIndex ref;
    double *d_tables = nullptr; /* qthr[102] | match[256] | mis[256] */
    /* device arenas of finished work lists are kept for the next one (hipMalloc of several GB costs ~0.2 s) */
    std::vector<std::pair<void *, size_t>> arena_cache;
    std::mutex arena_mu;
    /* a tiny private reference pool for spx_probaln_batch */
};

struct spx_work {
    spx::HostBatch hb;
    int32_t n_groups_in = 0;
    void *arena = nullptr;
    size_t arena_bytes = 0, arena_cap = 0;
    spx_dev_batch cls_batch[SPX_N_CLASSES];
    int cls_used[SPX_N_CLASSES] = {};
    int main_cls = -1;
    int64_t cls_cells[SPX_N_CLASSES] = {};
    spx_dev_groups dg;
    bool have_groups = false;
    /* device output mirrors */
    double *d_score = nullptr;
    uint8_t *d_prim = nullptr, *d_max = nullptr, *d_pass = nullptr;
    uint16_t *d_tie = nullptr;
    int32_t *d_grp_index = nullptr;
    uint8_t *d_bq = nullptr, *d_q = nullptr, *d_posmin = nullptr;
    int32_t *d_state = nullptr;
    spx_stats st;
    spx_params par;
    bool launched = false;
    std::vector<int64_t> launch_ids; /* this work list's launches since its last spx_collect (indices into the ctx event ring) */
    std::vector<uint8_t> posmin_host; /* filled by spx_collect: per first-of-position marker, min quality */
    std::vector<uint8_t> bq_host;     /* filled by spx_apply_quals: BAQ value of every wanted row */
};

extern "C" const char *spx_strerror(int code)
{
    switch (code) {
    case SPX_OK: return "ok";
    case SPX_ENODEVICE: return "no usable HIP device (gfx950 required, no CPU fallback)";
    case SPX_EHIP: return "HIP runtime error";
    case SPX_EINVAL: return "invalid argument";
    case SPX_ENOMEM: return "out of memory";
    case SPX_EUNSUPPORTED: return "construct left undefined by the reference / not supported";
    case SPX_ENOREF: return "reference not set";
    case SPX_ENOTAG: return "At least one of the MD or CS tags should be present!";
    default: return "unknown error";
    }
}
extern "C" const char *spx_last_error(void) { return g_err.c_str(); }

extern "C" int spx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int spx_create(int device, spx_ctx **out)
{
    if (!out) return fail(SPX_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SPX_ENODEVICE, "hipGetDeviceCount found no device");
    if (device < 0 || device >= n) return fail(SPX_ENODEVICE, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SPX_ENODEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    spx_ctx *c = new spx_ctx();
    c->device = device;
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int r = 0; r < spx_ctx::SPX_EV_RING; ++r)
        for (int i = 0; i < 6; ++i) HIPCHK(hipEventCreate(&c->evr[r][i]));
    for (int i = 0; i < SPX_N_CLASSES; ++i) {
        HIPCHK(hipStreamCreateWithFlags(&c->cls_stream[i], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->cls_done[i], hipEventDisableTiming));
    }
    {
        std::vector<double> t(102 + 512);
        spx::phred_thresholds(t.data());
        spx::score_tables(t.data() + 102, t.data() + 102 + 256);
        HIPCHK(hipMalloc((void **)&c->d_tables, t.size() * sizeof(double)));
        HIPCHK(hipMemcpy(c->d_tables, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    *out = c;
    return SPX_OK;
}

extern "C" void spx_destroy(spx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_ref4) (void)hipFree(c->d_ref4);
    if (c->d_tables) (void)hipFree(c->d_tables);
    for (auto &a : c->arena_cache) (void)hipFree(a.first);
    for (int i = 0; i < 6; ++i)
        for (int r = 0; r < spx_ctx::SPX_EV_RING; ++r)
            if (c->evr[r][i]) (void)hipEventDestroy(c->evr[r][i]);
    for (int i = 0; i < SPX_N_CLASSES; ++i) {
        if (c->cls_done[i]) (void)hipEventDestroy(c->cls_done[i]);
        if (c->cls_stream[i]) (void)hipStreamDestroy(c->cls_stream[i]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int spx_set_reference(spx_ctx *c, const spx_ref *ref)
{
    if (!c || !ref) return fail(SPX_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    const int nc = ref->n_contigs;
    c->ref.nib_off.assign(nc, 0);
    c->ref.len.assign(nc, 0);
    int64_t nib = spx::kRefLeadNibbles; /* leading pad: the kernels fetch codes up to a band width before a window */
    for (int i = 0; i < nc; ++i) {
        c->ref.nib_off[i] = nib;
        c->ref.len[i] = ref->seq_off[i + 1] - ref->seq_off[i];
        nib += (c->ref.len[i] + 1) & ~(int64_t)1; /* every contig starts on a byte boundary */
    }
    std::vector<uint8_t> packed((size_t)(nib / 2) + spx::kRefTailBytes, 0); /* slack: ... and past a window */
    unsigned nthr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    auto work = [&](unsigned tid) {
        for (int i = (int)tid; i < nc; i += (int)nthr) {
            const char *s = ref->bases + ref->seq_off[i];
            uint8_t *d = packed.data() + c->ref.nib_off[i] / 2;
            const int64_t len = c->ref.len[i];
            for (int64_t k = 0; k + 1 < len; k += 2)
                d[k >> 1] = (uint8_t)(kNt16Int[kNt16Table[(unsigned char)s[k]]] |
                                      (kNt16Int[kNt16Table[(unsigned char)s[k + 1]]] << 4));
            if (len & 1) d[len >> 1] = kNt16Int[kNt16Table[(unsigned char)s[len - 1]]];
        }
    };
    for (unsigned t = 0; t < nthr; ++t) th.emplace_back(work, t);
    for (auto &t : th) t.join();
    c->ref.index_ambiguous(ref);
    if (c->d_ref4) { (void)hipFree(c->d_ref4); c->d_ref4 = nullptr; }
    c->ref_bytes = (int64_t)packed.size();
    HIPCHK(hipMalloc((void **)&c->d_ref4, packed.size()));
    HIPCHK(hipMemcpy(c->d_ref4, packed.data(), packed.size(), hipMemcpyHostToDevice));
    return SPX_OK;
}

/* ------------------------------------------------------------------ */
struct Carver {
    size_t off = 0;
    template <class T>
    size_t take(size_t n)
    {
        off = (off + 255) & ~(size_t)255;
        size_t at = off;
        off += n * sizeof(T);
        return at;
    }
};

static int build_device_batch(spx_ctx *c, spx_work *w, bool want_state_q)
{
    spx::HostBatch &hb = w->hb;
    const size_t np = hb.L.size(), nr = hb.rows.size(), ng = hb.grp_index.size(), nm = hb.markers.size();
    const double tb0 = now_s();
    /* per-class launch order: (W, L desc), each W padded to whole waves */
    std::vector<int32_t> order[SPX_N_CLASSES], order_b[SPX_N_CLASSES];
    std::vector<int32_t> ids[SPX_N_CLASSES];
    for (size_t p = 0; p < np; ++p) {
        int cls = spx::band_class(2 * hb.bw[p] + 1);
        ids[cls].push_back((int32_t)p);
        w->cls_cells[cls] += spx::band_cells(hb.L[p], hb.R[p], hb.bw[p]);
    }
    /* rows the backward kernel walks: L down to the first wanted row */
    auto brows = [&](int32_t p) { return hb.n_rows[p] > 0 ? hb.L[p] - hb.rows[hb.row_off[p]] + 1 : 0; };
    /* order = (band width ascending, length descending, index ascending).  Keys are built in index order, so a stable
     * LSD radix sort of (bw, max - length) gives it in O(n); forward and backward orders of all classes are
     * independent and sorted on their own threads */
    auto sort_class = [&](int cls, int pass) {
        const std::vector<int32_t> &v = ids[cls];
        std::vector<int32_t> &dst = pass ? order_b[cls] : order[cls];
        if (v.empty()) return;
        const int ppw = 64 / (pass ? spx::class_lanes_bwd(cls) : spx::class_lanes(cls));
        const size_t n = v.size();
        std::vector<uint64_t> a(n), b(n);
        for (size_t i = 0; i < n; ++i) {
            const int32_t p = v[i];
            const uint32_t len = (uint32_t)(pass ? brows(p) : hb.L[p]);
            const uint32_t key = ((uint32_t)hb.bw[p] << 20) | (0xfffffu - (len > 0xfffffu ? 0xfffffu : len)); /* bw <= 1023 */
            a[i] = ((uint64_t)key << 32) | (uint32_t)p;
        }
        for (int shift = 32; shift < 64; shift += 11) { /* three 11-bit digits cover the 30 key bits in use */
            size_t cnt[2049] = {0};
            for (size_t i = 0; i < n; ++i) cnt[((a[i] >> shift) & 2047) + 1]++;
            for (int d = 0; d < 2048; ++d) cnt[d + 1] += cnt[d];
            for (size_t i = 0; i < n; ++i) b[cnt[(a[i] >> shift) & 2047]++] = a[i];
            a.swap(b);
        }
        dst.reserve(n + 64);
        for (size_t i = 0; i < n;) {
            const uint64_t bwkey = a[i] >> 52;
            size_t j = i;
            while (j < n && (a[j] >> 52) == bwkey) { dst.push_back((int32_t)(uint32_t)a[j]); ++j; }
            while (dst.size() % ppw) dst.push_back(-1);
            i = j;
        }
    };
    {
        std::vector<std::thread> th;
        for (int cls = 0; cls < SPX_N_CLASSES; ++cls) {
            w->st.problems_per_class[cls] = (int64_t)ids[cls].size();
            if (ids[cls].empty()) continue;
            for (int pass = 0; pass < 2; ++pass) {
                if (ids[cls].size() < 20000) sort_class(cls, pass);
                else th.emplace_back(sort_class, cls, pass);
            }
        }
        for (auto &t : th) t.join();
    }
    const double tb1 = now_s();
    /* scratch offsets */
    std::vector<int64_t> s_off(np), fsave_off(np);
    std::vector<int32_t> prob_slots(np), row_prob(nr);
    int64_t s_tot = 0, f_tot = 0;
    for (size_t p = 0; p < np; ++p) {
        /* whole 64-byte lines behind a lead pad of one line: the one-lane forward kernel writes 1/s[] eight rows at a time */
        s_off[p] = s_tot + 8;
        s_tot += 8 + ((hb.L[p] + 2 + 7) & ~7);
        fsave_off[p] = f_tot;
        prob_slots[p] = spx::class_slots(spx::band_class(2 * hb.bw[p] + 1));
        f_tot += (int64_t)hb.n_rows[p] * 2 * prob_slots[p];
        for (int32_t w2 = 0; w2 < hb.n_rows[p]; ++w2) row_prob[hb.row_off[p] + w2] = (int32_t)p;
    }
    /* arena layout */
    Carver cv;
    size_t o_ref_nib = cv.take<int64_t>(np), o_qry_nib = cv.take<int64_t>(np), o_L = cv.take<int32_t>(np),
           o_R = cv.take<int32_t>(np), o_bw = cv.take<int32_t>(np), o_hmm = cv.take<double>(np * SPX_H_N),
           o_row_off = cv.take<int32_t>(np), o_n_rows = cv.take<int32_t>(np), o_s_off = cv.take<int64_t>(np),
           o_fs_off = cv.take<int64_t>(np), o_qry4 = cv.take<uint8_t>(hb.qry4.size() + 256), /* 128-byte lead pad + slack: chunked fetches reach a few codes outside a window */
           o_rows = cv.take<int32_t>(nr), o_expect = cv.take<int32_t>(nr), o_rawq = cv.take<uint8_t>(nr),
           o_row_prob = cv.take<int32_t>(nr), o_prob_slots = cv.take<int32_t>(np);
    size_t o_order[SPX_N_CLASSES], o_order_b[SPX_N_CLASSES];
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls) {
        o_order[cls] = cv.take<int32_t>(order[cls].size());
        o_order_b[cls] = cv.take<int32_t>(order_b[cls].size());
    }
    size_t o_mk_first = cv.take<int32_t>(ng + 1), o_markers = cv.take<spx_dev_marker>(nm), o_naln = cv.take<uint8_t>(ng),
           o_sec = cv.take<uint16_t>(ng), o_gidx = cv.take<int32_t>(ng);
    const size_t in_bytes = cv.off;
    size_t o_sinv = cv.take<double>((size_t)s_tot), o_fsave = cv.take<double>((size_t)f_tot),
           o_bq = cv.take<uint8_t>(nr + 16), o_posmin = cv.take<uint8_t>(nm + 16), o_state = want_state_q ? cv.take<int32_t>(nr) : 0,
           o_q = want_state_q ? cv.take<uint8_t>(nr + 16) : 0, o_sraw = want_state_q ? cv.take<double>((size_t)s_tot) : 0,
           o_score = cv.take<double>(ng * 10),
           o_prim = cv.take<uint8_t>(ng), o_max = cv.take<uint8_t>(ng), o_pass = cv.take<uint8_t>(ng),
           o_tie = cv.take<uint16_t>(ng);
    w->arena_bytes = cv.off + 256;
    {
        std::lock_guard<std::mutex> lk(c->arena_mu);
        int best = -1;
        for (size_t i = 0; i < c->arena_cache.size(); ++i)
            if (c->arena_cache[i].second >= w->arena_bytes && (best < 0 || c->arena_cache[i].second < c->arena_cache[best].second))
                best = (int)i;
        if (best >= 0) {
            w->arena = c->arena_cache[best].first;
            w->arena_cap = c->arena_cache[best].second;
            c->arena_cache.erase(c->arena_cache.begin() + best);
        }
    }
    if (!w->arena) {
        w->arena_cap = w->arena_bytes + w->arena_bytes / 8; /* head room so that the next, slightly larger list fits */
        HIPCHK(hipMalloc(&w->arena, w->arena_cap));
    }
    char *base = (char *)w->arena;
    double t0 = now_s();
    if (timing_on()) fprintf(stderr, "[spx timing] device batch: launch orders %.3f s, offsets+arena %.3f s (%.2f GB)\n", tb1 - tb0, t0 - tb1, w->arena_bytes / 1e9);
#define UP(off, vec)                                                                                         \
    if (!(vec).empty())                                                                                      \
    HIPCHK(hipMemcpyAsync(base + (off), (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice, \
                          c->stream))
    UP(o_ref_nib, hb.ref_nib); UP(o_qry_nib, hb.qry_nib); UP(o_L, hb.L); UP(o_R, hb.R); UP(o_bw, hb.bw);
    UP(o_hmm, hb.hmm); UP(o_row_off, hb.row_off); UP(o_n_rows, hb.n_rows); UP(o_s_off, s_off); UP(o_fs_off, fsave_off);
    UP(o_qry4 + 128, hb.qry4); UP(o_rows, hb.rows); UP(o_expect, hb.row_expect); UP(o_rawq, hb.row_rawq);
    UP(o_row_prob, row_prob); UP(o_prob_slots, prob_slots);
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls) { UP(o_order[cls], order[cls]); UP(o_order_b[cls], order_b[cls]); }
    UP(o_mk_first, hb.mk_first); UP(o_markers, hb.markers); UP(o_naln, hb.n_aln); UP(o_sec, hb.sec_mask);
    UP(o_gidx, hb.grp_index);
#undef UP
    HIPCHK(hipStreamSynchronize(c->stream));
    w->st.h2d_seconds = now_s() - t0;
    w->st.bytes_h2d = (int64_t)in_bytes;
    if (timing_on()) fprintf(stderr, "[spx timing] device batch: upload %.3f s (%.1f MB)\n", w->st.h2d_seconds, in_bytes / 1e6);
    w->d_bq = (uint8_t *)(base + o_bq);
    w->d_posmin = (uint8_t *)(base + o_posmin);
    w->d_state = want_state_q ? (int32_t *)(base + o_state) : nullptr;
    w->d_q = want_state_q ? (uint8_t *)(base + o_q) : nullptr;
    w->d_score = (double *)(base + o_score);
    w->d_prim = (uint8_t *)(base + o_prim);
    w->d_max = (uint8_t *)(base + o_max);
    w->d_pass = (uint8_t *)(base + o_pass);
    w->d_tie = (uint16_t *)(base + o_tie);
    w->d_grp_index = (int32_t *)(base + o_gidx);
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls) {
        spx_dev_batch &B = w->cls_batch[cls];
        memset(&B, 0, sizeof B);
        w->cls_used[cls] = !order[cls].empty();
        if (w->cls_used[cls] && (w->main_cls < 0 || w->cls_cells[cls] > w->cls_cells[w->main_cls])) w->main_cls = cls;
        B.order = (const int32_t *)(base + o_order[cls]);
        B.order_bwd = (const int32_t *)(base + o_order_b[cls]);
        B.n_order = (int32_t)order[cls].size();
        B.n_order_bwd = (int32_t)order_b[cls].size();
        B.ref_nib = (const int64_t *)(base + o_ref_nib);
        B.qry_nib = (const int64_t *)(base + o_qry_nib);
        B.L = (const int32_t *)(base + o_L);
        B.R = (const int32_t *)(base + o_R);
        B.bw = (const int32_t *)(base + o_bw);
        B.hmm = (const double *)(base + o_hmm);
        B.row_off = (const int32_t *)(base + o_row_off);
        B.n_rows = (const int32_t *)(base + o_n_rows);
        B.s_off = (const int64_t *)(base + o_s_off);
        B.ref4 = c->d_ref4;
        B.qry4 = (const uint8_t *)(base + o_qry4 + 128);
        B.rows = (const int32_t *)(base + o_rows);
        B.row_expect = (const int32_t *)(base + o_expect);
        B.row_rawq = (const uint8_t *)(base + o_rawq);
        B.sinv = (double *)(base + o_sinv);
        B.s_raw = want_state_q ? (double *)(base + o_sraw) : nullptr;
        B.fsave = (double *)(base + o_fsave);
        B.row_prob = (const int32_t *)(base + o_row_prob);
        B.prob_slots = (const int32_t *)(base + o_prob_slots);
        B.fsave_stride = 2 * spx::class_slots(cls);
        B.fsave_off = (const int64_t *)(base + o_fs_off);
        B.out_bq = w->d_bq;
        B.out_state = w->d_state;
        B.out_q = w->d_q;
        B.qthr = c->d_tables;
    }
    spx_dev_groups &G = w->dg;
    memset(&G, 0, sizeof G);
    G.n_groups = (int32_t)ng;
    G.mk_first = (const int32_t *)(base + o_mk_first);
    G.markers = (const spx_dev_marker *)(base + o_markers);
    G.n_aln = (const uint8_t *)(base + o_naln);
    G.sec_mask = (const uint16_t *)(base + o_sec);
    G.out_bq = w->d_bq;
    G.match_tbl = c->d_tables + 102;
    G.mis_tbl = c->d_tables + 102 + 256;
    G.min_q = w->par.min_q;
    G.prim_margin = w->par.prim_margin_score;
    G.min_score = (double)w->par.min_score;
    G.score = w->d_score;
    G.prim_idx = w->d_prim;
    G.max_idx = w->d_max;
    G.tie_mask = w->d_tie;
    G.pass = w->d_pass;
    w->have_groups = ng > 0;
    return SPX_OK;
}

/* several record batches (e.g. the blocks a reader thread hands over) become ONE work list; group g of
 * batch b is reported at index (groups of batches < b) + g */
extern "C" int spx_prepare_many(spx_ctx *c, const spx_batch *const *bts, int32_t n_batches, const spx_params *par,
                                int host_threads, spx_work **out)
{
    if (!c || !bts || n_batches <= 0 || !par || !out) return fail(SPX_EINVAL, "NULL argument");
    if (!c->d_ref4) return fail(SPX_ENOREF, "spx_set_reference has not been called");
    HIPCHK(hipSetDevice(c->device));
    *out = nullptr;
    spx_work *w = new spx_work();
    memset(&w->st, 0, sizeof w->st);
    w->par = *par;
    double t0 = now_s();
    /* tasks of 16..256 groups (about four per thread, so that long reads do not leave threads idle), pulled by the
     * worker threads, merged in file order */
    struct Task { int b; int32_t g0, g1, base; };
    std::vector<Task> tasks;
    int nthr = host_threads > 0 ? host_threads : (int)std::thread::hardware_concurrency();
    nthr = std::max(1, std::min(nthr, 128));
    int64_t n_all = 0;
    for (int b = 0; b < n_batches; ++b) {
        if (!bts[b]) { delete w; return fail(SPX_EINVAL, "NULL batch"); }
        n_all += bts[b]->n_groups;
    }
    const int32_t tsize = (int32_t)std::max<int64_t>(16, std::min<int64_t>(256, n_all / (4 * (int64_t)nthr)));
    int32_t base = 0;
    for (int b = 0; b < n_batches; ++b) {
        for (int32_t g = 0; g < bts[b]->n_groups; g += tsize)
            tasks.push_back({b, g, std::min(bts[b]->n_groups, g + tsize), base});
        base += bts[b]->n_groups;
    }
    w->n_groups_in = base;
    nthr = std::min<int>(nthr, (int)std::max<size_t>(tasks.size(), 1));
    const spx::RefIndex &ri = c->ref;
    std::vector<spx::HostBatch> parts(tasks.size());
    std::atomic<size_t> next(0);
    auto run = [&]() {
        for (;;) {
            size_t t = next.fetch_add(1);
            if (t >= tasks.size()) break;
            const Task &k = tasks[t];
            spx::prepare_groups(bts[k.b], ri, par, k.g0, k.g1, parts[t]);
            for (int32_t &gi : parts[t].grp_index) gi += k.base;
            for (int32_t &qb : parts[t].qe_batch) qb = k.b;
        }
    };
    if (nthr == 1) run();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; ++t) th.emplace_back(run);
        for (auto &t : th) t.join();
    }
    const double t_par = now_s();
    w->hb.assign_merged(parts, nthr);
    w->st.prep_seconds = now_s() - t0;
    if (timing_on()) fprintf(stderr, "[spx timing] prepare: %d threads, group logic %.3f s, merge %.3f s\n", nthr, t_par - t0, now_s() - t_par);
    w->st.n_groups = w->n_groups_in;
    w->st.n_dispatched = (int64_t)w->hb.grp_index.size();
    w->st.n_problems = (int64_t)w->hb.L.size();
    w->st.n_rows = (int64_t)w->hb.rows.size();
    w->st.dp_cells = w->hb.dp_cells;
    w->st.n_markers = (int64_t)w->hb.markers.size();
    int rc = build_device_batch(c, w, false);
    if (rc) { spx_work_free(c, w); return rc; }
    *out = w;
    return SPX_OK;
}

extern "C" int spx_prepare(spx_ctx *c, const spx_batch *bt, const spx_params *par, int host_threads, spx_work **out)
{
    if (!bt) return fail(SPX_EINVAL, "NULL argument");
    return spx_prepare_many(c, &bt, 1, par, host_threads, out);
}

extern "C" int spx_launch(spx_ctx *c, spx_work *w)
{
    if (!c || !w) return fail(SPX_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    hipEvent_t *ev = c->evr[c->n_launch % spx_ctx::SPX_EV_RING];
    w->launch_ids.push_back(c->n_launch);
    if (w->launch_ids.size() > (size_t)spx_ctx::SPX_EV_RING) w->launch_ids.erase(w->launch_ids.begin());
    c->n_launch++;
    HIPCHK(hipEventRecord(ev[0], c->stream));
    /* the class holding most of the band cells runs on the main stream (its forward and backward kernels are
     * bracketed by events); the others run beside it on their own streams */
    const int mc = w->main_cls;
    static const bool serial = getenv("SPX_SERIAL") != nullptr; /* diagnostics: one class after the other */
    for (int cls = SPX_N_CLASSES - 1; cls >= 0; --cls) {
        if (!w->cls_used[cls] || cls == mc) continue;
        if (serial) { HIPCHK(spx_launch_baq(cls, 2, &w->cls_batch[cls], c->stream)); continue; }
        HIPCHK(hipStreamWaitEvent(c->cls_stream[cls], ev[0], 0));
        HIPCHK(spx_launch_baq(cls, 2, &w->cls_batch[cls], c->cls_stream[cls]));
        HIPCHK(hipEventRecord(c->cls_done[cls], c->cls_stream[cls]));
    }
    if (mc >= 0) {
        HIPCHK(hipEventRecord(ev[3], c->stream));
        HIPCHK(spx_launch_baq(mc, 0, &w->cls_batch[mc], c->stream));
        HIPCHK(hipEventRecord(ev[4], c->stream));
        HIPCHK(spx_launch_baq(mc, 1, &w->cls_batch[mc], c->stream));
        HIPCHK(hipEventRecord(ev[5], c->stream));
    }
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls)
        if (w->cls_used[cls] && cls != mc && !serial) HIPCHK(hipStreamWaitEvent(c->stream, c->cls_done[cls], 0));
    {
        int64_t narrow = 0, wide = 0;
        for (int cls = 0; cls < SPX_N_CLASSES; ++cls) (spx::class_slots(cls) <= 48 ? narrow : wide) += w->st.problems_per_class[cls];
        HIPCHK(spx_launch_map(&w->cls_batch[0], (int32_t)w->hb.rows.size(), wide > narrow, c->stream));
    }
    HIPCHK(hipEventRecord(ev[1], c->stream));
    if (w->have_groups) HIPCHK(spx_launch_score(&w->dg, (int32_t)w->hb.markers.size(), w->d_posmin, c->stream));
    HIPCHK(hipEventRecord(ev[2], c->stream));
    w->launched = true;
    return SPX_OK;
}

extern "C" int spx_pack_decisions(spx_ctx *c, spx_work *w, int32_t group_base, void *device_out, int64_t capacity)
{
    if (!c || !w || !device_out) return fail(SPX_EINVAL, "NULL argument");
    if (capacity < (int64_t)w->hb.grp_index.size()) return fail(SPX_EINVAL, "decision buffer too small");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(spx_launch_pack(&w->dg, w->d_grp_index, group_base, (unsigned long long *)device_out, c->stream));
    return (int)w->hb.grp_index.size();
}

extern "C" int spx_sync(spx_ctx *c)
{
    if (!c) return fail(SPX_EINVAL, "NULL argument");
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPX_OK;
}

extern "C" int spx_collect(spx_ctx *c, spx_work *w, spx_group_out *out)
{
    if (!c || !w || !out) return fail(SPX_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (w->launched) {
        /* averages over THIS work list's launches since its previous collect (those whose events are still in the ring);
         * other work lists launched in between have their own slots.  Timing is diagnostics: a failing event query
         * zeroes the figures instead of failing the collect */
        double baq = 0, sc = 0, fw = 0, bw = 0;
        int n = 0;
        bool ok = true;
        for (int64_t l : w->launch_ids) {
            if (c->n_launch - l > spx_ctx::SPX_EV_RING) continue; /* slot reused since */
            hipEvent_t *ev = c->evr[l % spx_ctx::SPX_EV_RING];
            float ms = 0;
            ok = ok && hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess;
            baq += ms;
            ok = ok && hipEventElapsedTime(&ms, ev[1], ev[2]) == hipSuccess;
            sc += ms;
            if (w->main_cls >= 0) {
                ok = ok && hipEventElapsedTime(&ms, ev[3], ev[4]) == hipSuccess;
                fw += ms;
                ok = ok && hipEventElapsedTime(&ms, ev[4], ev[5]) == hipSuccess;
                bw += ms;
            }
            ++n;
        }
        w->launch_ids.clear();
        if (!ok) { (void)hipGetLastError(); n = 0; }
        const double dn = n > 0 ? (double)n : 1.0;
        if (n == 0) baq = sc = fw = bw = 0;
        w->st.baq_kernel_ms = baq / dn;
        w->st.score_kernel_ms = sc / dn;
        w->st.n_launches_averaged = n;
        if (w->main_cls >= 0) {
            w->st.main_fwd_ms = fw / dn;
            w->st.main_bwd_ms = bw / dn;
            w->st.main_class = w->main_cls;
            w->st.main_class_cells = w->cls_cells[w->main_cls];
            w->st.main_class_lanes = spx::class_lanes(w->main_cls);
            w->st.main_class_slots = spx::class_slots(w->main_cls);
        }
        w->st.kernel_seconds = (w->st.baq_kernel_ms + w->st.score_kernel_ms) * 1e-3;
    }
    const size_t ng = w->hb.grp_index.size();
    std::vector<double> score(ng * 10);
    std::vector<uint8_t> prim(ng), mx(ng), pass(ng);
    std::vector<uint16_t> tie(ng);
    double t0 = now_s();
    if (ng) {
        HIPCHK(hipMemcpy(score.data(), w->d_score, ng * 10 * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(prim.data(), w->d_prim, ng, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(mx.data(), w->d_max, ng, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(pass.data(), w->d_pass, ng, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(tie.data(), w->d_tie, ng * sizeof(uint16_t), hipMemcpyDeviceToHost));
        w->posmin_host.resize(w->hb.markers.size());
        if (!w->posmin_host.empty())
            HIPCHK(hipMemcpy(w->posmin_host.data(), w->d_posmin, w->posmin_host.size(), hipMemcpyDeviceToHost));
    }
    w->st.d2h_seconds = now_s() - t0;
    w->st.bytes_d2h = (int64_t)(ng * (80 + 5));
    for (int32_t g = 0; g < w->n_groups_in; ++g) {
        memset(&out[g], 0, sizeof out[g]);
        int e = w->hb.grp_error[g];
        out[g].n_aln = (int8_t)(e < 0 ? e : 0);
        out[g].prim_idx = out[g].max_idx = out[g].best_idx = -1;
    }
    for (size_t k = 0; k < ng; ++k) {
        spx_group_out &o = out[w->hb.grp_index[k]];
        o.n_aln = (int8_t)w->hb.n_aln[k];
        for (int i = 0; i < 10; ++i) { o.score[i] = score[k * 10 + i]; o.rfe[i] = w->hb.rfe[k * 10 + i]; }
        o.prim_idx = (int8_t)prim[k];
        o.max_idx = (int8_t)mx[k];
        o.pass = (int8_t)pass[k];
        o.tie_mask = tie[k];
        o.n_problems = w->hb.grp_problems[k];
        o.n_markers = w->hb.mk_first[k + 1] - w->hb.mk_first[k];
        o.dp_cells = w->hb.grp_cells[k];
    }
    return SPX_OK;
}

/* the quality array the reference would hand to sam_write1 (secphase.c:182-189): replays calc_local_baq's
 * writes (ptMarker.c:706,759,763) with the BAQ values the kernels produced */
extern "C" int spx_apply_quals(spx_ctx *c, spx_work *w, int32_t batch_index, const spx_batch *bt, uint8_t *qual)
{
    if (!c || !w || !bt || !qual) return fail(SPX_EINVAL, "NULL argument");
    if (!(w->par.flags & SPX_PAR_ALL_ROWS)) return fail(SPX_EINVAL, "work list was not prepared with SPX_PAR_ALL_ROWS");
    if (!w->launched) return fail(SPX_EINVAL, "work list has not been launched");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const spx::HostBatch &hb = w->hb;
    if (w->bq_host.size() != hb.rows.size()) {
        w->bq_host.resize(hb.rows.size());
        if (!hb.rows.empty()) HIPCHK(hipMemcpy(w->bq_host.data(), w->d_bq, hb.rows.size(), hipMemcpyDeviceToHost));
    }
    const uint8_t keep = (uint8_t)(w->par.set_q < 94 ? w->par.set_q : 93);
    for (size_t k = 0; k < hb.qe_rec.size(); ++k) {
        if (hb.qe_batch[k] != batch_index) continue;
        const int32_t r = hb.qe_rec[k];
        if (r < 0 || r >= bt->n_alns) return fail(SPX_EINVAL, "batch does not match the work list");
        uint8_t *q = qual + bt->qual_off[r] + hb.qe_pos[k];
        if (hb.qe_len[k] == 0) { *q = 0; continue; }
        for (int32_t t = 0; t < hb.qe_len[k]; ++t) {
            const int32_t row = hb.qe_row0[k] + t;
            q[t] = hb.row_expect[row] >= 0 ? w->bq_host[row] : keep;
        }
    }
    return SPX_OK;
}

extern "C" int spx_work_stats(const spx_work *w, spx_stats *st)
{
    if (!w || !st) return fail(SPX_EINVAL, "NULL argument");
    *st = w->st;
    return SPX_OK;
}

extern "C" void spx_work_free(spx_ctx *c, spx_work *w)
{
    if (!w) return;
    if (c) (void)hipSetDevice(c->device);
    if (w->arena) {
        bool kept = false;
        if (c) {
            (void)hipStreamSynchronize(c->stream); /* nothing of this work list may still be running */
            std::lock_guard<std::mutex> lk(c->arena_mu);
            if (c->arena_cache.size() < 3) { c->arena_cache.emplace_back(w->arena, w->arena_cap); kept = true; }
        }
        if (!kept) (void)hipFree(w->arena);
    }
    delete w;
}

extern "C" int spx_score_batch(spx_ctx *c, const spx_batch *bt, const spx_params *par, spx_group_out *out,
                               spx_stats *stats)
{
    spx_work *w = nullptr;
    int rc = spx_prepare(c, bt, par, 0, &w);
    if (rc) return rc;
    rc = spx_launch(c, w);
    if (!rc) rc = spx_collect(c, w, out);
    if (!rc && stats) *stats = w->st;
    spx_work_free(c, w);
    return rc;
}

/* ------------------------------------------------------------------ */
/* get_best_record_index's rand()-dependent tail (ptAlignment.c:163-176), replayed in file order.
 * random_r with a private state is glibc's rand() algorithm without the process-global state; a finalizer
 * keeps that state across batches so that a whole run consumes ONE stream, like the reference at -@1. */
struct spx_finalizer {
    struct random_data rd;
    char statebuf[128];
};

extern "C" int spx_finalizer_create(unsigned rand_seed, spx_finalizer **out)
{
    if (!out) return fail(SPX_EINVAL, "NULL argument");
    spx_finalizer *f = new spx_finalizer();
    memset(&f->rd, 0, sizeof f->rd);
    memset(f->statebuf, 0, sizeof f->statebuf);
    initstate_r(rand_seed, f->statebuf, sizeof f->statebuf, &f->rd);
    *out = f;
    return SPX_OK;
}
extern "C" void spx_finalizer_free(spx_finalizer *f) { delete f; }

extern "C" int spx_finalizer_apply(spx_finalizer *f, const spx_params *par, spx_group_out *out, int32_t n_groups)
{
    if (!f || !par || !out) return fail(SPX_EINVAL, "NULL argument");
    for (int32_t g = 0; g < n_groups; ++g) {
        spx_group_out &o = out[g];
        o.best_idx = -1;
        o.relabel = 0;
        if (o.n_aln < 2) continue;
        int tied[16], cnt = 0, max_idx = o.max_idx;
        for (int a = 0; a < o.n_aln; ++a)
            if ((o.tie_mask >> a) & 1) tied[cnt++] = a;
        int32_t r;
        if (cnt > 1) { random_r(&f->rd, &r); max_idx = tied[r % cnt]; }
        random_r(&f->rd, &r);
        const int rnd = r % 2;
        const double max_score = o.max_idx >= 0 ? o.score[o.max_idx] : -DBL_MAX;
        const double prim_score = o.prim_idx >= 0 ? o.score[o.prim_idx] : -DBL_MAX;
        double dd = max_score - prim_score;
        int d = (dd > -2147483649.0 && dd < 2147483648.0) ? (int)dd : INT_MIN;
        if (d < 0 && d != INT_MIN) d = -d;
        int best;
        if (d < par->prim_margin_random) best = rnd == 0 ? o.prim_idx : max_idx;
        else best = o.pass ? max_idx : o.prim_idx;
        o.best_idx = (int8_t)best;
        o.relabel = (best >= 0 && best != o.prim_idx) ? 1 : 0;
    }
    return SPX_OK;
}

extern "C" int spx_finalize(const spx_params *par, unsigned rand_seed, spx_group_out *out, int32_t n_groups)
{
    spx_finalizer *f = nullptr;
    int rc = spx_finalizer_create(rand_seed, &f);
    if (rc) return rc;
    rc = spx_finalizer_apply(f, par, out, n_groups);
    spx_finalizer_free(f);
    return rc;
}

/* BED bookkeeping of relabelled reads (src/secphase.c:201-212): extents of the old primary and of the promoted
 * secondary (count 1 each), and the reference positions of their surviving markers */
extern "C" int spx_bedset_add(spx_bedset *b, const char *contig, int32_t start, int32_t end, int32_t count);
extern "C" int spx_relabel_blocks(const spx_work *w, const spx_ref *ref, const spx_group_out *out,
                                  spx_bedset *modified_blocks, spx_bedset *marker_blocks)
{
    if (!w || !ref || !out) return fail(SPX_EINVAL, "NULL argument");
    const spx::HostBatch &hb = w->hb;
    if (marker_blocks && w->posmin_host.size() != hb.markers.size()) return fail(SPX_EINVAL, "spx_collect has not run");
    int n = 0;
    for (size_t k = 0; k < hb.grp_index.size(); ++k) {
        const spx_group_out &o = out[hb.grp_index[k]];
        if (!o.relabel) continue;
        ++n;
        const int pair[2] = {o.prim_idx, o.best_idx};
        for (int t = 0; t < 2; ++t) {
            const int a = pair[t];
            const int32_t tid = hb.atid[k * 10 + a];
            const char *contig = ref->names + ref->name_off[tid];
            if (modified_blocks) spx_bedset_add(modified_blocks, contig, hb.rfs[k * 10 + a], hb.rfe[k * 10 + a], 1);
            if (!marker_blocks) continue;
            const int na = hb.n_aln[k];
            for (int32_t m = hb.mk_first[k]; m < hb.mk_first[k + 1]; m += na) {
                if (w->posmin_host[m] <= w->par.min_q) continue; /* position removed by filter_lowq_markers */
                const int32_t rp = hb.mk_ref_pos[m + a];
                spx_bedset_add(marker_blocks, contig, rp, rp, 0);
            }
        }
    }
    return n;
}

extern "C" int spx_write_relabel_log(const char *path, const char *mode, const spx_batch *bt, const spx_ref *ref,
                                     const spx_group_out *out)
{
    if (!path || !bt || !ref || !out) return fail(SPX_EINVAL, "NULL argument");
    FILE *f = fopen(path, mode && *mode ? mode : "w");
    if (!f) return fail(SPX_EINVAL, std::string("cannot open ") + path);
    for (int32_t g = 0; g < bt->n_groups; ++g) {
        const spx_group_out &o = out[g];
        if (!o.relabel) continue;
        fprintf(f, "#MARKER SCORE\n");
        fprintf(f, "$\t%s\n", bt->qnames + bt->qname_off[g]);
        int i = 0;
        for (int a = bt->grp_first[g]; a < bt->grp_first[g + 1]; ++a) {
            if (bt->flag[a] & SPX_FUNMAP) continue;
            if (i >= o.n_aln) break;
            const char *tag = !(bt->flag[a] & SPX_FSECONDARY) ? "*" : (i == o.best_idx ? "@" : "!");
            fprintf(f, "%s\t%.2f\t%s\t%ld\t%d\n", tag, o.score[i], ref->names + ref->name_off[bt->tid[a]],
                    (long)bt->pos[a], o.rfe[i]);
            ++i;
        }
        fprintf(f, "\n");
    }
    fclose(f);
    return SPX_OK;
}

/* ------------------------------------------------------------------ */
/* raw banded-HMM problems, all rows wanted */
static int probaln_run(spx_ctx *c, int32_t n, const uint8_t *ref, const int64_t *ref_off, const uint8_t *query,
                       const int64_t *qry_off, const int32_t *set_q, const spx_probaln_par *pars, int32_t *state, uint8_t *q,
                       double *kernel_ms, int32_t post_which, double *post_scale, double *post_zM, double *post_zI,
                       int32_t *pr_out = nullptr)
{
    if (!c || n < 0 || !ref || !ref_off || !query || !qry_off || !set_q || !pars || !state || !q)
        return fail(SPX_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    spx_work *w = new spx_work();
    memset(&w->st, 0, sizeof w->st);
    memset(&w->par, 0, sizeof w->par);
    spx::HostBatch &hb = w->hb;
    hb.clear();
    hb.mk_first.push_back(0);
    /* private reference pool: the problems' own ref windows */
    std::vector<uint8_t> ref4;
    int64_t rn = spx::kRefLeadNibbles; /* leading pad, as in spx_set_reference */
    for (int32_t p = 0; p < n; ++p) {
        const int R = (int)(ref_off[p + 1] - ref_off[p]), L = (int)(qry_off[p + 1] - qry_off[p]);
        if (R <= 0 || L <= 0) { delete w; return fail(SPX_EINVAL, "empty problem"); }
        const int bw = spx::effective_bw(R, L, pars[p].bw);
        if (spx::band_class(2 * bw + 1) < 0) { delete w; return fail(SPX_EUNSUPPORTED, "band wider than 2048 columns"); }
        hb.ref_nib.push_back(rn);
        ref4.resize((size_t)(rn / 2) + (size_t)(R + 1) / 2, 0);
        for (int k = 0; k < R; ++k) {
            unsigned code = ref[ref_off[p] + k] > 3 ? 4u : ref[ref_off[p] + k];
            ref4[(size_t)(rn / 2) + (k >> 1)] |= (uint8_t)(code << ((k & 1) << 2));
        }
        rn += ((R + 1) / 2) * 2;
        hb.qry_nib.push_back(hb.qry_nibbles);
        const size_t at = hb.qry4.size(), nb = ((size_t)(L + 7) / 8) * 4;
        hb.qry4.resize(at + nb, 0);
        for (int k = 0; k < L; ++k) {
            unsigned code = query[qry_off[p] + k] > 3 ? 4u : query[qry_off[p] + k];
            hb.qry4[at + (k >> 1)] |= (uint8_t)(code << ((k & 1) << 2));
        }
        hb.qry_nibbles += (int64_t)nb * 2;
        hb.L.push_back(L); hb.R.push_back(R); hb.bw.push_back(bw);
        hb.row_off.push_back((int32_t)hb.rows.size());
        hb.n_rows.push_back(L);
        for (int i = 1; i <= L; ++i) { hb.rows.push_back(i); hb.row_expect.push_back(i - 1); hb.row_rawq.push_back(93); }
        hb.hmm.resize(hb.hmm.size() + SPX_H_N);
        spx::hmm_constants(R, L, pars[p].d, pars[p].e, set_q[p], &hb.hmm[hb.hmm.size() - SPX_H_N]);
        {
            bool has_n = false;
            for (int k = 0; k < R; ++k) has_n |= ref[ref_off[p] + k] > 3;
            for (int k = 0; k < L; ++k) has_n |= query[qry_off[p] + k] > 3;
            hb.hmm[hb.hmm.size() - SPX_H_N + SPX_H_PAD0] = has_n ? 1.0 : 0.0;
        }
        hb.dp_cells += spx::band_cells(L, R, bw);
    }
    ref4.resize(ref4.size() + spx::kRefTailBytes, 0);
    uint8_t *d_ref = nullptr, *saved = c->d_ref4;
    HIPCHK(hipMalloc((void **)&d_ref, ref4.size()));
    HIPCHK(hipMemcpy(d_ref, ref4.data(), ref4.size(), hipMemcpyHostToDevice));
    c->d_ref4 = d_ref;
    int rc = build_device_batch(c, w, true);
    c->d_ref4 = saved;
    if (!rc) rc = spx_launch(c, w);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(SPX_EHIP, "kernel execution failed");
    if (!rc) {
        float ms = 0;
        hipEvent_t *ev = c->evr[(c->n_launch - 1) % spx_ctx::SPX_EV_RING];
        (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
        if (kernel_ms) *kernel_ms = ms;
        const size_t nr = hb.rows.size();
        if (post_scale) { /* spx_probaln_posteriors: 1/s[] and z = f*b of every slot of every row of one problem */
            int64_t s_at = 0, f_at = 0; /* same layout rule as build_device_batch */
            s_at = 8;
            for (int32_t p = 0; p < post_which; ++p) {
                s_at += 8 + ((hb.L[p] + 2 + 7) & ~7);
                f_at += (int64_t)hb.n_rows[p] * 2 * spx::class_slots(spx::band_class(2 * hb.bw[p] + 1));
            }
            const int cls = spx::band_class(2 * hb.bw[post_which] + 1), slots = spx::class_slots(cls), L = hb.L[post_which],
                      R = hb.R[post_which], bw = hb.bw[post_which];
            std::vector<double> zv((size_t)L * 2 * slots);
            if (hipMemcpy(post_scale, w->cls_batch[cls].sinv + s_at, ((size_t)L + 2) * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(zv.data(), w->cls_batch[cls].fsave + f_at, zv.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
                rc = fail(SPX_EHIP, "copy back failed");
            post_scale[0] = 1.0;
            for (int i = 1; i <= L && !rc; ++i)
                for (int k = 1; k <= R; ++k) {
                    const int j = k - (i - bw);
                    const bool in = k >= std::max(1, i - bw) && k <= std::min(R, i + bw);
                    post_zM[(size_t)(i - 1) * R + (k - 1)] = in ? zv[((size_t)(i - 1) * 2 + 0) * slots + j] : 0.0;
                    post_zI[(size_t)(i - 1) * R + (k - 1)] = in ? zv[((size_t)(i - 1) * 2 + 1) * slots + j] : 0.0;
                }
        }
        if (pr_out && !rc) { /* probaln_glocal's return value: phred-scaled likelihood from the scaling factors s[0..L+1] */
            int64_t s_all = 0;
            for (int32_t p = 0; p < n; ++p) s_all += 8 + ((hb.L[p] + 2 + 7) & ~7);
            std::vector<double> sraw((size_t)s_all), sfin((size_t)s_all);
            const int cls0 = spx::band_class(2 * hb.bw[0] + 1);
            if (hipMemcpy(sraw.data(), w->cls_batch[cls0].s_raw, sraw.size() * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(sfin.data(), w->cls_batch[cls0].sinv, sfin.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
                rc = fail(SPX_EHIP, "copy back failed");
            int64_t s_at = 8;
            for (int32_t p = 0; p < n && !rc; ++p) {
                const int L = hb.L[p], R = hb.R[p];
                double pp = 1., Pr1 = 0.; /* s[0] = 1 */
                for (int i = 1; i <= L + 1; ++i) {
                    pp *= i < L ? sraw[s_at + i] : sfin[s_at + i]; /* sinv[L], sinv[L+1] hold s[L], s[L+1] themselves */
                    if (pp < 1e-100) { Pr1 += -4.343 * log(pp); pp = 1.; }
                }
                Pr1 += -4.343 * log(pp * R * L);
                const double v = Pr1 + .499;
                pr_out[p] = (v > -2147483649.0 && v < 2147483648.0) ? (int)v : INT_MIN;
                s_at += 8 + ((L + 2 + 7) & ~7);
            }
        }
        if (hipMemcpy(state, w->d_state, nr * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(q, w->d_q, nr, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(SPX_EHIP, "copy back failed");
    }
    (void)hipFree(d_ref);
    spx_work_free(c, w);
    return rc;
}

extern "C" int spx_probaln_batch(spx_ctx *c, int32_t n, const uint8_t *ref, const int64_t *ref_off, const uint8_t *query,
                                 const int64_t *qry_off, const int32_t *set_q, const spx_probaln_par *pars,
                                 int32_t *state, uint8_t *q, double *kernel_ms)
{
    return probaln_run(c, n, ref, ref_off, query, qry_off, set_q, pars, state, q, kernel_ms, 0, nullptr, nullptr, nullptr);
}

extern "C" int spx_probaln_posteriors(spx_ctx *c, int32_t n, const uint8_t *ref, const int64_t *ref_off, const uint8_t *query,
                                      const int64_t *qry_off, const int32_t *set_q, const spx_probaln_par *pars, int32_t which,
                                      double *scale, double *zM, double *zI)
{
    if (!scale || !zM || !zI || !qry_off || which < 0 || which >= n) return fail(SPX_EINVAL, "invalid argument");
    std::vector<int32_t> state((size_t)qry_off[n]);
    std::vector<uint8_t> q((size_t)qry_off[n]);
    return probaln_run(c, n, ref, ref_off, query, qry_off, set_q, pars, state.data(), q.data(), nullptr, which, scale, zM, zI);
}

static std::mutex g_single_mu;
static spx_ctx *g_single = nullptr;

extern "C" int spx_probaln_glocal(const uint8_t *ref, int l_ref, const uint8_t *query, int l_query, const uint8_t *iqual,
                                  const spx_probaln_par *cpar, int *state, uint8_t *q)
{
    if (l_ref <= 0 || l_query <= 0) return 0; /* as htslib */
    if (!ref || !query || !cpar || !state || !q) return INT_MIN;
    int sq = iqual ? iqual[0] : 30;
    if (iqual)
        for (int i = 1; i < l_query; ++i)
            if (iqual[i] != sq) { fail(SPX_EUNSUPPORTED, "per-base iqual: secphase always passes a constant (ptMarker.c:747-749)"); return INT_MIN; }
    std::lock_guard<std::mutex> lk(g_single_mu);
    if (!g_single && spx_create(0, &g_single) != SPX_OK) return INT_MIN;
    int64_t ro[2] = {0, l_ref}, qo[2] = {0, l_query};
    int32_t sq32 = sq;
    std::vector<int32_t> st32(l_query);
    int32_t pr = 0;
    int rc = probaln_run(g_single, 1, ref, ro, query, qo, &sq32, cpar, st32.data(), q, nullptr, 0, nullptr, nullptr, nullptr, &pr);
    if (rc) return INT_MIN;
    for (int i = 0; i < l_query; ++i) state[i] = st32[i];
    /* phred-scaled likelihood, like htslib (secphase itself only tests the return value for INT_MIN, ptMarker.c:755-760) */
    return pr;
}

/* ------------------------------------------------------------------ */
/* host-only plan view */
struct spx_plan {
    spx::HostBatch hb;
    std::vector<int32_t> mk_row;
    std::vector<uint8_t> mk_qfix, mk_is_match, mk_aln, mk_fop;
};

extern "C" int spx_plan_create(const spx_ref *ref, const spx_batch *bt, const spx_params *par, spx_plan **out)
{
    if (!ref || !bt || !par || !out) return fail(SPX_EINVAL, "NULL argument");
    spx::RefIndex ri;
    int64_t nib = spx::kRefLeadNibbles;
    for (int i = 0; i < ref->n_contigs; ++i) {
        ri.nib_off.push_back(nib);
        ri.len.push_back(ref->seq_off[i + 1] - ref->seq_off[i]);
        nib += (ri.len[i] + 1) & ~(int64_t)1;
    }
    ri.index_ambiguous(ref);
    spx_plan *p = new spx_plan();
    spx::prepare_groups(bt, ri, par, 0, bt->n_groups, p->hb);
    for (const spx_dev_marker &m : p->hb.markers) {
        p->mk_row.push_back(m.row);
        p->mk_qfix.push_back(m.qfix);
        p->mk_is_match.push_back(m.is_match);
        p->mk_aln.push_back(m.aln);
        p->mk_fop.push_back(m.first_of_pos);
    }
    *out = p;
    return SPX_OK;
}

extern "C" int spx_plan_get(const spx_plan *p, spx_plan_view *v)
{
    if (!p || !v) return fail(SPX_EINVAL, "NULL argument");
    const spx::HostBatch &h = p->hb;
    v->n_problems = (int32_t)h.L.size();
    v->n_rows = (int32_t)h.rows.size();
    v->n_groups = (int32_t)h.grp_index.size();
    v->n_markers = (int32_t)h.markers.size();
    v->L = h.L.data(); v->R = h.R.data(); v->bw = h.bw.data();
    v->ref_tid = h.ref_tid.data(); v->ref_rfs = h.ref_rfs.data();
    v->qry_nib = h.qry_nib.data(); v->qry4 = h.qry4.data(); v->hmm = h.hmm.data();
    v->row_off = h.row_off.data(); v->n_rows_of = h.n_rows.data();
    v->rows = h.rows.data(); v->row_expect = h.row_expect.data(); v->row_rawq = h.row_rawq.data();
    v->grp_index = h.grp_index.data(); v->mk_first = h.mk_first.data();
    v->mk_row = p->mk_row.data(); v->mk_qfix = p->mk_qfix.data(); v->mk_is_match = p->mk_is_match.data();
    v->mk_aln = p->mk_aln.data(); v->mk_first_of_pos = p->mk_fop.data();
    v->n_aln = h.n_aln.data(); v->sec_mask = h.sec_mask.data(); v->rfe = h.rfe.data();
    v->grp_error = h.grp_error.data();
    v->n_qedits = (int32_t)h.qe_rec.size(); v->pad_ = 0;
    v->qe_rec = h.qe_rec.data(); v->qe_pos = h.qe_pos.data(); v->qe_len = h.qe_len.data(); v->qe_row0 = h.qe_row0.data();
    return SPX_OK;
}

extern "C" void spx_plan_free(spx_plan *p) { delete p; }

extern "C" void spx_host_tables(double *thr, double *match_tbl, double *mis_tbl)
{
    if (thr) spx::phred_thresholds(thr);
    if (match_tbl && mis_tbl) spx::score_tables(match_tbl, mis_tbl);
}
