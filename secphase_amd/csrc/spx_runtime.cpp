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
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spx.h"
#include "spx_device.h"
#include "spx_prep.h"
#include "spx_prep_dev.h"
#include "spx_devin.h"
#include "spx_cpuacc.h"
#include "spx_pool.h"

struct spx_bedset;

extern "C" hipError_t spx_launch_baq(int cls, int phase, const spx_dev_batch *B, hipStream_t st);
extern "C" hipError_t spx_launch_score(const spx_dev_groups *Gd, int32_t n_markers, uint8_t *posmin, hipStream_t st);
extern "C" hipError_t spx_launch_map(const spx_dev_batch *B, int32_t n_rows_total, int wide, hipStream_t st);
extern "C" hipError_t spx_launch_fast(int cls, int phase, const spx_dev_batch *B, const spx_fast_consts *K, hipStream_t st);
extern "C" hipError_t spx_launch_fast_map(const spx_dev_batch *B, const spx_fast_consts *K, int32_t n_rows_total, int max_slots, hipStream_t st);
extern "C" int spx_fast_class(int cls);
extern "C" hipError_t spx_launch_pack(const spx_dev_groups *Gd, const int32_t *grp_index, int32_t group_base,
                                      spx_decision *out, hipStream_t st);
extern "C" hipError_t spx_launch_results(const spx_dev_groups *Gd, const spx_group_info *info, const int32_t *rfe,
                                         spx_group_out *out, hipStream_t st);
extern "C" hipError_t spx_prep_phase1(const spx_prep_args *A, const uint32_t *raw_seq, int64_t seq_words, hipStream_t st);
extern "C" size_t spx_prep_heavy_temp_bytes(int32_t n);
extern "C" hipError_t spx_prep_heavy(const spx_prep_args *A, int32_t *keys, int32_t *vals, void *temp, size_t temp_bytes, int32_t *slot_heavy, int32_t *group_heavy,
                                     uint8_t *slot_flag, uint8_t *group_flag, int32_t min_work, hipStream_t st);
extern "C" hipError_t spx_prep_phase2(const spx_prep_args *A, spxl::PlanBase *base_out, int64_t *mk_base, hipStream_t st);
extern "C" hipError_t spx_prep_emit(const spx_prep_args *A, const spx_emit_args *E, hipStream_t st);
extern "C" size_t spx_order_temp_bytes(int32_t n_prob);
extern "C" hipError_t spx_stage_expand(const void *recs, int32_t n_slots, const uint8_t *pk_seq, const uint8_t *pk_qual, uint8_t *seq, uint8_t *qual,
                                       int has_alias, hipStream_t st);
extern "C" hipError_t spx_prep_orders(const spx_order_args *O, hipStream_t st);
extern "C" hipError_t spx_prep_slice_bounds(const int32_t *slot0, const spxl::PlanBase *base, int32_t ng, int32_t K, const spxl::PlanBase *tot,
                                            spxl::PlanBase *out, hipStream_t st);

extern "C" size_t spx_bgzf_inflate_scratch_bytes(int32_t n_blocks);
extern "C" hipError_t spx_launch_bgzf_inflate2(const uint8_t *comp, const void *blocks, int32_t n_blocks, uint8_t *out, int32_t *status, int check_crc,
                                               void *scratch, hipStream_t st);
extern "C" hipError_t spx_launch_bgzf_inflate(const uint8_t *comp, const void *blocks, int32_t n_blocks, uint8_t *out, int32_t *status,
                                              int check_crc, hipStream_t st);

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
     * the main class' forward and backward kernels, 6 behind the last backward kernels on the main stream, 7 in front of the last MAP kernel on
     * the stream it runs on): spx_collect averages over the launches since the previous collect */
    static const int SPX_EV_RING = 64;
    hipEvent_t evr[SPX_EV_RING][8] = {};
    std::atomic<int64_t> n_launch{0}; /* written under launch_mu, read by collects of other lists (ThreadSanitizer run on the GPU box) */
    /* the band classes run concurrently: a handful of wide-band problems must not serialise behind
     * (or in front of) the bulk class */
    /* (three side streams, the classes spread over them by band cells: with the main, the preparation and the copy
     * stream that is six -- one hardware queue each, see spx_create) */
    static const int SPX_N_SIDE = 8; /* at most */
    int n_side = 6;                  /* in use (SPX_SIDE_STREAMS; round 5: 3 before -- on the mixed workload nine classes queued on three side streams
                                      * and the chain 45 -> 47 -> (4,26) of small, latency-bound launches was the longest path of a slice) */
    hipStream_t side_stream[SPX_N_SIDE] = {};
    hipEvent_t side_done[SPX_N_SIDE] = {};
    /* two-tier DP: the exact re-runs of uncertified problems fan out over streams of their own (SPX_RERUN_STREAMS, default 3; 0: over the side
     * streams -- where they stood in front of the NEXT list's side classes, and the main stream waited ~20 ms per mixed list at its join) */
    static const int SPX_N_RERUN = 4;
    int n_rerun = 3;
    hipStream_t rerun_stream[SPX_N_RERUN] = {};
    /* host -> HBM copies of staged records, and NOTHING else: once a kernel or a memset has gone through a stream the
     * runtime serves its copies with a copy KERNEL instead of the DMA engines -- 18-24 GB/s beside the DP kernels instead
     * of 52 (tools/scratch/h2d_probe.hip) */
    hipStream_t copy_stream = nullptr;
    hipStream_t unpack_stream = nullptr; /* what follows the copies of a staging: unpack / rebuild kernels, pad fills */
    hipStream_t result_stream = nullptr; /* packed results -> host (never behind a staging in progress) */
    std::mutex launch_mu;              /* spx_launch may be called from several threads (pipelined callers) */
    uint8_t *d_ref4 = nullptr;
    int64_t ref_bytes = 0;
    spx::RefIndex ref;
    void *d_refidx = nullptr;     /* device copy of the contig table + ambiguous-base index (spxl::RefView) */
    spxl::RefView d_rv = {};
    double *d_tables = nullptr; /* qthr[102] | match[256] | mis[256] */
    /* device arenas of finished work lists are kept for the next one (hipMalloc of several GB costs ~0.2 s) */
    std::vector<std::pair<void *, size_t>> arena_cache;
    std::vector<std::pair<void *, size_t>> pinned_cache; /* hipHostMalloc'ed staging buffers */
    std::mutex arena_mu;
    size_t hbm_bytes = (size_t)256 << 30; /* the device's total memory (hipMemGetInfo at spx_create) */
    std::condition_variable arena_cv; /* signalled when a work list gives device memory back (arena_put) */
    std::atomic<bool> hbm_tight{false}; /* an allocation has failed once: no more head room on new blocks */
    size_t arena_in_use = 0;            /* bytes of blocks handed out by arena_get (under arena_mu) */
    /* Staging ring: record batches go to HBM through a few pinned chunks that are filled by a thread pool and copied
     * asynchronously, one behind the other (a pinned buffer per batch would mean pinning gigabytes anew whenever a batch
     * is larger than any before -- ~0.2 s per GB -- and again after every spx_trim) */
    static const int SPX_PIN_CHUNKS = 4;
    void *pin_chunk[SPX_PIN_CHUNKS] = {};
    hipEvent_t pin_done[SPX_PIN_CHUNKS] = {};
    bool pin_busy[SPX_PIN_CHUNKS] = {};
    size_t pin_bytes = 0;
    int pin_next = 0;
    std::thread warm; /* allocates the ring behind spx_create's back (pinning 512 MB takes ~0.15 s: off the first staging's path) */
    std::mutex stage_mu; /* one staging at a time per context */
    std::unique_ptr<spx::Pool> stage_pool;
    /* work-list preparation on the device: its own stream (it overlaps the DP kernels of the previous list), pools
     * that only live during a preparation and are shared by all of them (prep_mu serialises preparations) */
    int prep_cus = 0; /* CUs reserved for the preparation streams (0: no CU masks) */
    struct DevBuf {
        void *p = nullptr;
        size_t cap = 0;
    };
    /* Preparation LANES (round 3): the preparation kernels are chains of dependent loads -- one lane walks one alignment --
     * that leave most of the chip idle, and their duration is set by the longest alignment of the batch, not by the number
     * of groups.  So the preparations of several batches run SIDE BY SIDE, each on its own stream with its own pools
     * (a lane's mutex serialises the preparations that share it; work list w uses lane w->lane). */
    static const int SPX_N_PREP = 12; /* lanes that exist; n_prep of them are used (SPX_PREP_LANES, default 4).  A lane keeps its pools: ~29 GB for lists of
                                       * 16 384 ONT groups, so six lanes + five such lists in flight over-commit the device (measured: 424 instead of 325 ms
                                       * per step); workloads with small lists gain from more lanes (mixed 2-100 kb, 16 384 groups: +3 % with six) */
    int n_prep = 4;
    struct PrepLane {
        spx_ctx *owner = nullptr;
        hipStream_t stream = nullptr;
        std::mutex mu;
        DevBuf pool_ops, pool_conf, pool_mm, pool_garena, pool_keys, pool_sort, pool_heavy;
        spx_prep_totals *d_tot = nullptr, *h_tot = nullptr; /* device / pinned host */
        spxl::PlanBase *d_bounds = nullptr, *h_bounds = nullptr; /* DP slices: SPX_MAX_SLICES + 1 prefix records */
        DevBuf pool_bins;                                   /* 2 passes x 3 x slices x SPX_N_CLASSES x 1024 int32 */
        spx_order_segs *d_segs = nullptr, *h_segs = nullptr; /* 2 x SPX_MAX_SLICES (forward, backward): device / pinned host */
    } lane[SPX_N_PREP];
    std::atomic<unsigned> lane_rr{0};
    /* the scratch slack the consensus rounds of some list of this context needed (1, 4, 16 ..): later lists start with it instead of finding it
     * out again by an overflow and a second run of the whole group phase (round 5: two of seven mixed batches paid that at EVERY preparation) */
    std::atomic<int> slack_hint{1};
    std::atomic<int> work_arenas{0}, work_arenas_most{0}; /* staged work lists that hold their arena now / the most so far (or what a pipeline announced) */
    /* two-tier DP: what the latest list that looked decided (fast_classes: -1 nobody has looked yet, 1 tiers, 0 exact kernels only).  The lists of a run are
     * alike, and the decision is only known after the counting phase, which already needs the row size (4 rows of slots per wanted row with the tiers, 2
     * without): lists that follow a "0" are counted with 2 and do not look again, except every 32nd (the mixed 2-100 kb workload: its lists had twice the
     * scratch for nothing, and at eight lists in flight the step went from 79 to 171 ms) */
    std::atomic<int> tiers_hint{-1};
    std::atomic<unsigned> tiers_looks{0};
    std::atomic<int> tiers_epoch{0};
};

/* grow-only device buffer of a preparation lane; the caller holds the lane's mutex.  Kernels of an earlier preparation
 * may still read the old allocation, so the lane's stream is drained before it is released. */
static void arena_flush(spx_ctx *c);
static int ensure_pool(spx_ctx::PrepLane &PL, spx_ctx::DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return SPX_OK;
    if (b.p) {
        if (hipStreamSynchronize(PL.stream) != hipSuccess) return SPX_EHIP;
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 4 + (1u << 20);
    {
        /* (not the device's last bytes: see arena_get) */
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            if (free_b < want + total_b / 64 && PL.owner) { arena_flush(PL.owner); want = bytes + (1u << 20); (void)hipMemGetInfo(&free_b, &total_b); }
            if (free_b < want + total_b / 128) return SPX_ENOMEM;
        } else (void)hipGetLastError();
    }
    const double tp0 = now_s();
    const hipError_t pe = hipMalloc(&b.p, want);
    if (timing_on()) fprintf(stderr, "[spx timing] preparation pool of %.2f GB (lane %d): hipMalloc %.3f s%s\n", want / 1e9, PL.owner ? (int)(&PL - PL.owner->lane) : -1, now_s() - tp0, pe == hipSuccess ? "" : " (failed)");
    if (pe != hipSuccess) {
        /* the blocks the context keeps for re-use go back to the driver, then once more without the head room */
        (void)hipGetLastError();
        if (PL.owner) arena_flush(PL.owner);
        want = bytes + (1u << 20);
        if (hipMalloc(&b.p, want) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; return SPX_ENOMEM; }
    }
    b.cap = want;
    return SPX_OK;
}

/* work lists of one pipeline take the memory of their lists in ticket order */
struct spx_alloc_gate {
    std::mutex mu;
    std::condition_variable cv;
    int64_t turn = 0;
};
extern "C" spx_alloc_gate *spx_internal_gate_create(void) { return new spx_alloc_gate(); }
extern "C" void spx_internal_lists_in_flight(spx_ctx *c, int n);
extern "C" void spx_internal_gate_free(spx_alloc_gate *g) { delete g; }
/* a ticket that will never reach the allocation (its job failed earlier) is passed over */
extern "C" void spx_internal_gate_skip(spx_alloc_gate *g, int64_t ticket)
{
    if (!g) return;
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [&] { return g->turn >= ticket; });
    if (g->turn == ticket) { ++g->turn; g->cv.notify_all(); }
}

struct spx_work {
    spx::HostBatch hb;
    int32_t n_groups_in = 0;
    /* ---- device-prepared work lists: the staged records (part A, resident until the list is freed) ---- */
    spx_ctx *owner = nullptr;
    bool staged = false;        /* spx_stage has run: records are in HBM */
    int stage_threads = 1;
    bool prepared = false;      /* spx_prepare_staged has run: the work list exists */
    spx::Stage stage;
    void *h_stage = nullptr;    /* pinned host copy of the staged buffer */
    size_t h_stage_cap = 0;
    void *in_arena = nullptr;   /* device: staged buffer | recoded SEQ | per-alignment / per-group state */
    size_t in_cap = 0;
    size_t o_code = 0, o_ast = 0, o_gc = 0, o_ac = 0, o_gab = 0, o_gao = 0, o_base = 0, o_mkb = 0, o_scan = 0;
    spx_prep_args pa;
    spx_prep_totals tot;
    hipEvent_t ev_ready = nullptr; /* recorded on the preparation stream when the list may be launched */
    hipEvent_t ev_staged = nullptr; /* recorded on the copy stream when the records are in HBM */
    hipEvent_t ev_done = nullptr;  /* recorded on the main stream behind the last kernel of the latest launch */
    int32_t n_dgroups = 0;      /* groups that passed the dispatch filter (device arrays have one entry each) */
    std::vector<spx_group_info> info; /* pulled back by spx_collect */
    spx_group_out *d_results = nullptr;
    spx_group_info *d_info = nullptr;
    int32_t *d_rfe = nullptr, *d_rfs = nullptr, *d_atid = nullptr, *d_mk_first = nullptr, *d_mk_ref_pos = nullptr;
    int32_t *d_qe[5] = {};
    int32_t *d_row_expect = nullptr;
    int64_t n_rows_dev = 0, n_mk_dev = 0, n_prob_dev = 0;
    bool mirrors_markers = false, mirrors_qe = false;
    void *arena = nullptr;
    size_t arena_bytes = 0, arena_cap = 0;
    spx_dev_batch cls_batch[SPX_N_CLASSES];
    int cls_used[SPX_N_CLASSES] = {};
    /* DP slices (round 4): the list's problems in K ranges of consecutive groups; forward -> backward -> MAP run slice by
     * slice over ONE scratch area (1/s of every DP row + the saved forward rows) sized for the largest slice, so that a
     * list can be LONG (the preparation's fixed latency amortised) without being LARGE.  Empty: one launch over everything. */
    struct Slice {
        int64_t r0 = 0, r1 = 0; /* its wanted rows */
        spx_dev_batch cls_batch[SPX_N_CLASSES];
        hipEvent_t ev_start = nullptr, ev_f0 = nullptr, ev_f1 = nullptr, ev_b1 = nullptr;
    };
    std::vector<Slice> slices;
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
    /* two-tier DP: per problem tier + 4 counters (device), the launch-uniform constants, whether this list uses the tiers */
    int32_t *d_tier = nullptr, *d_tier_counts = nullptr;
    spx_fast_consts fk;
    bool fast = false, any_fast_cls = false, any_exact_cls = false;
    bool tiers_look = false; /* this list was counted with the fast tier's row size and decides for itself (spx_ctx::tiers_hint) */
    bool cls_fast[SPX_N_CLASSES] = {}; /* the classes of THIS list that take the fast tier: those with a fast kernel that hold a share of the list's band
                                        * cells worth it (SPX_FAST_MIN_SHARE percent, default 2) -- every class with a fast tier costs a re-run launch per
                                        * slice, and a launch lasts a wave's lifetime however few problems it has (the mixed workload's nine classes) */
    float fast_d = 0, fast_e = 0; /* host-built lists (spx_probaln_batch): the parameters of problem 0; problems with others take the exact tier */
    int fast_set_q = 0;
    int64_t n_launches_counted = 0; /* launches since the counters were last reset by a preparation */
    std::vector<hipEvent_t> rr_ev;  /* events of the re-run fan-out (one start event per slice + one per fast class), created on demand */
    spx_stats st;
    spx_params par;
    bool launched = false;
    std::vector<int64_t> launch_ids; /* this work list's launches since its last spx_collect (indices into the ctx event ring) */
    std::vector<uint8_t> posmin_host; /* filled by spx_collect: per first-of-position marker, min quality */
    std::vector<uint8_t> bq_host;     /* filled by spx_apply_quals: BAQ value of every wanted row */
    std::atomic<int> in_pipe{0};      /* submitted to a pipeline and not yet delivered: no second submission, no release */
    int lane = 0;                     /* preparation lane of the latest spx_prepare_staged */
    /* a pipeline's allocation gate: work lists take their device memory in submission order (see spx_pipe.cpp) */
    spx_alloc_gate *gate = nullptr;
    int64_t gate_ticket = 0;
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
/* internal: lets the pipeline hand a worker thread's error text to the thread that asks for the results */
extern "C" void spx_internal_set_error(const char *msg) { g_err = msg ? msg : ""; }

extern "C" void spx_internal_work_gate(spx_work *w, spx_alloc_gate *g, int64_t ticket)
{
    if (!w) return;
    w->gate = g;
    w->gate_ticket = ticket;
}

extern "C" int spx_internal_work_claim(spx_work *w, int claim)
{
    if (!w) return SPX_EINVAL;
    if (!claim) { w->in_pipe = 0; return SPX_OK; }
    int expect = 0;
    return w->in_pipe.compare_exchange_strong(expect, 1) ? SPX_OK : SPX_EINVAL;
}

extern "C" int spx_internal_ctx_device(spx_ctx *c) { return c ? c->device : -1; }

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
    /* HIP multiplexes streams onto hardware queues (4 by default): two streams that share one run their kernels one
     * after the other.  Ask for 16 (main, six side, copy, unpack, result and up to six preparation lanes' streams) before the runtime initialises; a process that has initialised HIP already (e.g.
     * after importing torch) must have set the variable itself -- bench.py and the command line do. */
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    int n = 0;
    const double tc0 = now_s();
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SPX_ENODEVICE, "hipGetDeviceCount found no device");
    if (device < 0 || device >= n) return fail(SPX_ENODEVICE, "device index out of range");
    HIPCHK(hipSetDevice(device));
    /* host threads that wait for the device SLEEP (the runtime's default is to spin): the waits of the pipeline workers,
     * of the staging ring and of the inflate workers would otherwise burn the CPU time the container is short of (the boxes
     * give it 16 cores; a device-inflated chunk cost 25 ms of CPU, most of it spinning beside a 20 ms kernel) */
    if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    const double tc1 = now_s();
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SPX_ENODEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    spx_ctx *c = new spx_ctx();
    c->device = device;
    if (prop.totalGlobalMem > 0) c->hbm_bytes = (size_t)prop.totalGlobalMem;
    /* Optional spatial partition of the chip (SPX_PREP_CUS=k, off by default): k CUs are reserved for the preparation
     * stream and masked out of the DP streams.  Measured on MI355X (round 2): the DP kernels fill every SIMD's register
     * file, so preparation waves launched beside them wait for the DP launch to drain -- but the preparation kernels
     * are scattered-load bound and need far more than 16-32 CUs to finish within a DP step, so reserving CUs loses
     * (127 / 82 ms per step with 16 / 32 CUs against 63 ms without masks).  Kept for experiments. */
    int prep_cus = 0;
    if (const char *e = getenv("SPX_PREP_CUS")) prep_cus = atoi(e);
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> m_prep((size_t)(ncu + 31) / 32, 0u), m_dp((size_t)(ncu + 31) / 32, 0u);
    bool masked = prep_cus > 0 && prep_cus < ncu / 2;
    if (masked) {
        /* queue CU masks on a multi-XCD part: bit i names CU i / 8 of XCD i % 8 (the driver deals the bits round-robin
         * over the XCDs), so a block of 8k consecutive bits takes k CUs from EVERY XCD -- anything else would leave one
         * XCD short and the DP kernels, whose workgroups are dealt evenly over the XCDs, would wait for it */
        prep_cus = (prep_cus + 7) / 8 * 8;
        for (int i = 0; i < ncu; ++i) (i < prep_cus ? m_prep : m_dp)[(size_t)i / 32] |= 1u << (i % 32);
    }
    /* SPX_DP_EXCLUDE_CUS=k (round 5 experiment): only the DP streams are masked -- k CUs (a multiple of 8: one or more from every XCD) never
     * run a DP wave, the preparation streams stay unmasked and find free registers there whatever the DP kernels are doing */
    bool dp_only = false;
    if (const char *e = getenv("SPX_DP_EXCLUDE_CUS")) {
        int k = (atoi(e) + 7) / 8 * 8;
        if (!masked && k > 0 && k < ncu / 2) {
            for (int i = 0; i < ncu; ++i) if (i >= k) m_dp[(size_t)i / 32] |= 1u << (i % 32);
            masked = dp_only = true;
        }
    }
    auto mk_stream = [&](hipStream_t *st, const std::vector<uint32_t> &mask) -> hipError_t {
        if (masked && !(dp_only && &mask == &m_prep)) {
            hipError_t e = hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data());
            if (e == hipSuccess) return e;
            (void)hipGetLastError();
            masked = false; /* not supported here: plain streams for everything created from now on */
        }
        return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
    if (const char *e = getenv("SPX_PREP_LANES")) c->n_prep = std::max(1, std::min(atoi(e), (int)spx_ctx::SPX_N_PREP));
    for (int l = 0; l < c->n_prep; ++l) {
        spx_ctx::PrepLane &PL = c->lane[l];
        PL.owner = c;
        HIPCHK(mk_stream(&PL.stream, m_prep));
        HIPCHK(hipMalloc((void **)&PL.d_tot, sizeof(spx_prep_totals)));
        HIPCHK(hipHostMalloc((void **)&PL.h_tot, sizeof(spx_prep_totals), hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&PL.d_bounds, sizeof(spxl::PlanBase) * (SPX_MAX_SLICES + 1)));
        HIPCHK(hipHostMalloc((void **)&PL.h_bounds, sizeof(spxl::PlanBase) * (SPX_MAX_SLICES + 1), hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&PL.d_segs, sizeof(spx_order_segs) * 2 * SPX_MAX_SLICES));
        HIPCHK(hipHostMalloc((void **)&PL.h_segs, sizeof(spx_order_segs) * 2 * SPX_MAX_SLICES, hipHostMallocDefault));
    }
    c->prep_cus = masked ? prep_cus : 0;
    HIPCHK(mk_stream(&c->stream, m_dp));
    for (int r = 0; r < spx_ctx::SPX_EV_RING; ++r)
        for (int i = 0; i < 8; ++i) HIPCHK(hipEventCreate(&c->evr[r][i]));
    if (const char *e = getenv("SPX_SIDE_STREAMS")) c->n_side = std::max(1, std::min((int)spx_ctx::SPX_N_SIDE, atoi(e)));
    for (int i = 0; i < c->n_side; ++i) {
        HIPCHK(mk_stream(&c->side_stream[i], m_dp));
        HIPCHK(hipEventCreateWithFlags(&c->side_done[i], hipEventDisableTiming | hipEventBlockingSync));
    }
    if (const char *e = getenv("SPX_RERUN_STREAMS")) c->n_rerun = std::max(0, std::min((int)spx_ctx::SPX_N_RERUN, atoi(e)));
    for (int i = 0; i < c->n_rerun; ++i) HIPCHK(mk_stream(&c->rerun_stream[i], m_dp));
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->unpack_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->result_stream, hipStreamNonBlocking));
    {
        std::vector<double> t(102 + 512);
        spx::phred_thresholds(t.data());
        spx::score_tables(t.data() + 102, t.data() + 102 + 256);
        HIPCHK(hipMalloc((void **)&c->d_tables, t.size() * sizeof(double)));
        HIPCHK(hipMemcpy(c->d_tables, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    {
        size_t mb = 128;
        if (const char *e = getenv("SPX_PIN_MB")) mb = (size_t)std::max(1, atoi(e));
        c->pin_bytes = mb << 20;
        if (!getenv("SPX_NO_WARM"))
            c->warm = std::thread([c] { /* (a chunk that cannot be had here is tried again by the staging that needs it) */
                std::lock_guard<std::mutex> sl(c->stage_mu);
                if (hipSetDevice(c->device) != hipSuccess) return;
                for (int k = 0; k < spx_ctx::SPX_PIN_CHUNKS; ++k) {
                    if (c->pin_chunk[k]) continue;
                    if (hipHostMalloc(&c->pin_chunk[k], c->pin_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->pin_chunk[k] = nullptr; return; }
                    if (hipEventCreateWithFlags(&c->pin_done[k], hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) {
                        (void)hipGetLastError();
                        (void)hipHostFree(c->pin_chunk[k]);
                        c->pin_chunk[k] = nullptr;
                        return;
                    }
                }
            });
    }
    if (timing_on()) {
        const double tc2 = now_s();
        (void)hipFree(nullptr);
        fprintf(stderr, "[spx timing] spx_create: device count + properties %.3f s, streams / events / first allocations %.3f s\n", tc1 - tc0, tc2 - tc1);
    }
    *out = c;
    return SPX_OK;
}

extern "C" void spx_destroy(spx_ctx *c)
{
    if (!c) return;
    if (c->warm.joinable()) c->warm.join();
    (void)hipSetDevice(c->device);
    if (c->d_ref4) (void)hipFree(c->d_ref4);
    if (c->d_refidx) (void)hipFree(c->d_refidx);
    if (c->d_tables) (void)hipFree(c->d_tables);
    for (auto &a : c->arena_cache) (void)hipFree(a.first);
    for (auto &a : c->pinned_cache) (void)hipHostFree(a.first);
    for (int k = 0; k < spx_ctx::SPX_PIN_CHUNKS; ++k) {
        if (c->pin_chunk[k]) (void)hipHostFree(c->pin_chunk[k]);
        if (c->pin_done[k]) (void)hipEventDestroy(c->pin_done[k]);
    }
    c->stage_pool.reset();
    for (int l = 0; l < spx_ctx::SPX_N_PREP; ++l) {
        spx_ctx::PrepLane &PL = c->lane[l];
        for (spx_ctx::DevBuf *b : {&PL.pool_ops, &PL.pool_conf, &PL.pool_mm, &PL.pool_garena, &PL.pool_keys, &PL.pool_sort, &PL.pool_heavy})
            if (b->p) (void)hipFree(b->p);
        if (PL.d_tot) (void)hipFree(PL.d_tot);
        if (PL.h_tot) (void)hipHostFree(PL.h_tot);
        if (PL.d_bounds) (void)hipFree(PL.d_bounds);
        if (PL.h_bounds) (void)hipHostFree(PL.h_bounds);
        if (PL.pool_bins.p) (void)hipFree(PL.pool_bins.p);
        if (PL.d_segs) (void)hipFree(PL.d_segs);
        if (PL.h_segs) (void)hipHostFree(PL.h_segs);
        if (PL.stream) (void)hipStreamDestroy(PL.stream);
    }
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < spx_ctx::SPX_EV_RING; ++r)
            if (c->evr[r][i]) (void)hipEventDestroy(c->evr[r][i]);
    for (int i = 0; i < spx_ctx::SPX_N_SIDE; ++i) {
        if (c->side_done[i]) (void)hipEventDestroy(c->side_done[i]);
        if (c->side_stream[i]) (void)hipStreamDestroy(c->side_stream[i]);
        if (i < spx_ctx::SPX_N_RERUN && c->rerun_stream[i]) (void)hipStreamDestroy(c->rerun_stream[i]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->unpack_stream) (void)hipStreamDestroy(c->unpack_stream);
    if (c->result_stream) (void)hipStreamDestroy(c->result_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int spx_set_reference(spx_ctx *c, const spx_ref *ref)
{
    if (!c || !ref) return fail(SPX_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    const int nc = ref->n_contigs;
    c->ref.build(ref);
    int64_t nib = spx::kRefLeadNibbles;
    for (int i = 0; i < nc; ++i) nib += (c->ref.len[i] + 1) & ~(int64_t)1;
    std::vector<uint8_t> packed((size_t)(nib / 2) + spx::kRefTailBytes, 0); /* slack: ... and past a window */
    { /* characters -> 4-bit codes, in pieces of 4 Mbp on threads (htslib's seq_nt16_table / seq_nt16_int, ptMarker.c:744) */
        struct Piece { int contig; int64_t k0, k1; };
        std::vector<Piece> pieces;
        const int64_t step = (int64_t)4 << 20; /* even: a piece starts on a byte boundary */
        for (int i = 0; i < nc; ++i)
            for (int64_t k = 0; k < c->ref.len[i]; k += step) pieces.push_back({i, k, std::min(c->ref.len[i], k + step)});
        static const struct Tbl { uint8_t code[256]; Tbl() { for (int x = 0; x < 256; ++x) code[x] = kNt16Int[kNt16Table[x]]; } } T;
        std::atomic<size_t> next(0);
        auto work = [&]() {
            for (;;) {
                const size_t q = next.fetch_add(1);
                if (q >= pieces.size()) break;
                const Piece &pc = pieces[q];
                const unsigned char *s = (const unsigned char *)ref->bases + ref->seq_off[pc.contig];
                uint8_t *d = packed.data() + c->ref.nib_off[pc.contig] / 2;
                int64_t k = pc.k0;
                for (; k + 1 < pc.k1; k += 2) d[k >> 1] = (uint8_t)(T.code[s[k]] | (T.code[s[k + 1]] << 4));
                if (k < pc.k1) d[k >> 1] = T.code[s[k]];
            }
        };
        const unsigned nthr = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(32, std::thread::hardware_concurrency()), pieces.size()));
        if (nthr <= 1) work();
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nthr; ++t) th.emplace_back(work);
            for (auto &t : th) t.join();
        }
    }
    if (c->d_ref4) { (void)hipFree(c->d_ref4); c->d_ref4 = nullptr; }
    c->ref_bytes = (int64_t)packed.size();
    HIPCHK(hipMalloc((void **)&c->d_ref4, packed.size()));
    HIPCHK(hipMemcpy(c->d_ref4, packed.data(), packed.size(), hipMemcpyHostToDevice));
    { /* contig table + ambiguous-base index for the preparation kernels */
        if (c->d_refidx) { (void)hipFree(c->d_refidx); c->d_refidx = nullptr; }
        const size_t b0 = (size_t)nc * 8, b1 = (size_t)nc * 8, b2 = ((size_t)nc + 1) * 8, b3 = c->ref.npos.size() * 4 + 16;
        HIPCHK(hipMalloc(&c->d_refidx, b0 + b1 + b2 + b3 + 64));
        char *d = (char *)c->d_refidx;
        if (nc) {
            HIPCHK(hipMemcpy(d, c->ref.nib_off.data(), b0, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(d + b0, c->ref.len.data(), b1, hipMemcpyHostToDevice));
        }
        HIPCHK(hipMemcpy(d + b0 + b1, c->ref.npos_off.data(), b2, hipMemcpyHostToDevice));
        if (!c->ref.npos.empty()) HIPCHK(hipMemcpy(d + b0 + b1 + b2, c->ref.npos.data(), c->ref.npos.size() * 4, hipMemcpyHostToDevice));
        c->d_rv.n_contigs = nc;
        c->d_rv.nib_off = (const int64_t *)d;
        c->d_rv.len = (const int64_t *)(d + b0);
        c->d_rv.npos_off = (const int64_t *)(d + b0 + b1);
        c->d_rv.npos = (const int32_t *)(d + b0 + b1 + b2);
    }
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

static void *arena_get(spx_ctx *c, size_t bytes, size_t *cap);

/* ---- two-tier DP switch and constants ---- */
static std::atomic<int> g_dp_tiers{-1};
static std::atomic<int> g_dp_tiers_epoch{0}; /* bumped by spx_set_dp_tiers: contexts forget what their earlier lists decided (spx_ctx::tiers_hint) */
extern "C" int spx_get_dp_tiers(void)
{
    int v = g_dp_tiers.load();
    if (v < 0) {
        const char *e = getenv("SPX_DP_TIERS");
        v = (e && (!strcmp(e, "0") || !strcmp(e, "off") || !strcmp(e, "exact"))) ? 0 : 1;
        g_dp_tiers.store(v);
    }
    return v;
}
extern "C" int spx_set_dp_tiers(int on)
{
    g_dp_tiers.store(on < 0 ? 0 : (on > 2 ? 1 : on)); /* 2 = test mode: the fast tier runs but certifies nothing (every problem is re-run) */
    g_dp_tiers_epoch.fetch_add(1);
    return SPX_OK;
}
static std::atomic<int64_t> g_last_tier[5];
extern "C" int spx_last_tier_stats(int64_t *out /* 5: fast problems, re-run by certificate / model / range, rows not certified */)
{
    if (!out) return fail(SPX_EINVAL, "NULL argument");
    for (int k = 0; k < 5; ++k) out[k] = g_last_tier[k].load();
    return SPX_OK;
}
/* which band classes of a list take the fast tier (spx_work::cls_fast) */
static void fast_classes(spx_work *w)
{
    int share = 2;
    if (const char *e = getenv("SPX_FAST_MIN_SHARE")) share = std::max(0, std::min(100, atoi(e)));
    int64_t tot = 0;
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls) tot += w->cls_used[cls] ? w->cls_cells[cls] : 0;
    /* ... and the list as a whole takes the tiers only if those classes hold most of its cells (SPX_FAST_MIN_TOTAL percent, default 92): the mixed
     * 2-100 kb workload spreads its cells over nine classes, its lists are small, and the fast tier's extra launches (MAP twice, re-runs) cost it more
     * than its faster forward kernel gains (measured at four lists in flight: 195 k groups/s either way; at eight lists in flight on six preparation
     * lanes, the configuration that is fastest for it: 117 k with the tiers -- twice the scratch per list -- and 213 k without; its fast classes hold
     * ~87 % of its cells, those of the HiFi and ONT presets 97-98 %) */
    int min_total = 92;
    if (const char *e = getenv("SPX_FAST_MIN_TOTAL")) min_total = std::max(0, std::min(100, atoi(e)));
    int64_t in_fast = 0;
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls)
        if (w->fast && w->cls_used[cls] && spx_fast_class(cls) && w->cls_cells[cls] * 100 >= tot * share) in_fast += w->cls_cells[cls];
    if (w->fast && in_fast * 100 < tot * min_total) w->fast = false;
    w->any_fast_cls = w->any_exact_cls = false;
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls) {
        w->cls_fast[cls] = w->fast && w->cls_used[cls] && spx_fast_class(cls) && w->cls_cells[cls] * 100 >= tot * share;
        if (w->cls_used[cls]) (w->cls_fast[cls] ? w->any_fast_cls : w->any_exact_cls) = true;
    }
}
/* constants every problem of a launch shares (spx_device.h spx_fast_consts), from the list's (d, e, set_q) exactly as
 * spxl::hmm_constants forms them (float expressions promoted), without the (1 - sM) factors.
 * range_bits + 16 rows x mu_bits + 100 (constant factors between the carried rows and the exact tier's M, I, D) must stay
 * below 1022: a row of the exact tier, normalised to sum 1, then keeps every value in the normal FP64 range between two checks. */
static void fast_constants(float d, float e, int set_q, spx_fast_consts *K)
{
    memset(K, 0, sizeof *K);
    double h[SPX_H_N];
    spx::hmm_constants(100, 100, d, e, set_q, h); /* (for e_match / e_mis / m6 / m8: they do not depend on the lengths) */
    K->m0h = (double)((1 - d) - d);
    K->m1h = (double)d;
    K->m3h = (double)(1 - e);
    K->m4h = (double)e;
    K->m6 = h[SPX_H_M6];
    K->m8 = h[SPX_H_M8];
    K->e_match = h[SPX_H_EMATCH];
    K->e_mis = h[SPX_H_EMIS];
    K->pw[0] = 1.0;
    for (int c = 1; c <= SPX_FAST_MAXC; ++c) K->pw[c] = K->pw[c - 1] * K->m8;
    K->ups = K->m6 * K->m1h;
    K->gam = 0.25 * K->m1h;
    K->emU = K->e_match * K->ups; K->exU = K->e_mis * K->ups;
    K->cU0 = K->m0h / K->ups; K->cU1 = (K->m3h * K->gam) / K->ups;
    K->c4 = 0.25 * K->m4h;
    K->emB = K->e_match * K->m0h; K->exB = K->e_mis * K->m0h;
    K->cB1 = (K->gam * K->m3h) / K->m0h; K->cB2 = (K->m1h * K->m6) / K->m0h;
    K->rho = (K->gam * K->m3h) / K->m0h;
    static const int rb = [] { const char *e2 = getenv("SPX_FAST_RANGE_BITS"); const int v = e2 ? atoi(e2) : 600; return v < 64 ? 64 : (v > 700 ? 700 : v); }();
    K->range_bits = spx_get_dp_tiers() == 2 ? -1 : rb; /* (test mode: every range check fails) */
    K->mu_bits = (1000 - 100 - rb) / 16;
}

static int build_device_batch(spx_ctx *c, spx_work *w, bool want_state_q, bool allow_fast = true)
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
    const bool fast_order = allow_fast && spx_get_dp_tiers() != 0; /* the fast forward kernel stops at the LAST wanted row: its order goes by that */
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
            const uint32_t len = (uint32_t)(pass ? brows(p) : (fast_order && hb.n_rows[p] > 0 ? hb.rows[hb.row_off[p] + hb.n_rows[p] - 1] : hb.L[p]));
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
    const int row_mult = (allow_fast && spx_get_dp_tiers() != 0) ? 4 : 2; /* two-tier DP: forward and backward rows side by side (spxl::Params::row_mult) */
    std::vector<int64_t> s_off(np), fsave_off(np);
    std::vector<int32_t> prob_slots(np), row_prob(nr);
    int64_t s_tot = 0, f_tot = 0;
    for (size_t p = 0; p < np; ++p) {
        /* whole 64-byte lines behind a lead pad of one line: the one-lane forward kernel writes 1/s[] eight rows at a time */
        s_off[p] = s_tot + 8;
        s_tot += 8 + ((hb.L[p] + 2 + 7) & ~7);
        fsave_off[p] = f_tot;
        prob_slots[p] = spx::class_slots(spx::band_class(2 * hb.bw[p] + 1));
        f_tot += (int64_t)hb.n_rows[p] * row_mult * prob_slots[p];
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
    const size_t o_tier = cv.take<int32_t>(np + 8);
    w->arena_bytes = cv.off + 256;
    w->arena = arena_get(c, w->arena_bytes, &w->arena_cap);
    if (!w->arena) return fail(SPX_ENOMEM, "device memory for the work list");
    char *base = (char *)w->arena;
    w->fast = allow_fast && spx_get_dp_tiers() != 0 && np > 0;
    w->d_tier = (int32_t *)(base + o_tier);
    w->d_tier_counts = w->d_tier + np;
    w->n_launches_counted = 0;
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls) w->cls_used[cls] = !order[cls].empty();
    fast_classes(w); /* (may decide against the tiers for this list) */
    if (w->fast) {
        fast_constants(w->fast_d, w->fast_e, w->fast_set_q, &w->fk);
        HIPCHK(hipMemsetAsync(w->d_tier, 0xff, (np + 8) * sizeof(int32_t), c->stream));
    }
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
        B.tier = w->fast ? w->d_tier : nullptr;
        B.tier_counts = w->fast ? w->d_tier_counts : nullptr;
        B.tier_want = SPX_TIER_ALL;
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

/* ------------------------------------------------------------------ */
/* device allocations of finished work lists are kept for the next one (hipMalloc of several GB costs ~0.2 s) */
/* every block waiting for re-use goes back to the driver */
static void arena_flush(spx_ctx *c)
{
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(c->arena_mu);
        for (auto &a : c->arena_cache) drop.push_back(a.first);
        c->arena_cache.clear();
    }
    for (void *d : drop) (void)hipFree(d);
}
extern "C" void spx_internal_lists_in_flight(spx_ctx *c, int n)
{
    if (!c) return;
    int h = c->work_arenas_most.load();
    while (h < n && !c->work_arenas_most.compare_exchange_weak(h, n)) {}
}
static void work_arena_taken(spx_ctx *c) { spx_internal_lists_in_flight(c, ++c->work_arenas); }
static void work_arena_back(spx_ctx *c) { --c->work_arenas; }

static void *arena_get(spx_ctx *c, size_t bytes, size_t *cap)
{
    /* When HBM is full -- several large work lists in flight: 16 384 ONT groups need ~70 GB of saved rows -- the caller
     * WAITS for an older list to be collected and freed instead of failing: in a pipeline the oldest list always holds
     * its memory already and finishes.  Gives up after two minutes (a single list that cannot fit). */
    const double t_end = now_s() + 120.0;
    for (;;) {
        {
            std::lock_guard<std::mutex> lk(c->arena_mu);
            int best = -1;
            /* best fit, and never a block more than half again as large as asked for: the image of the staged records
             * and the work list are two size classes, and a small request that takes a large block sends the next large
             * request to hipMalloc (~25 ms per GB, with the device idle meanwhile) */
            for (size_t i = 0; i < c->arena_cache.size(); ++i)
                if (c->arena_cache[i].second >= bytes && c->arena_cache[i].second <= bytes + bytes / 2 + ((size_t)64 << 20) &&
                    (best < 0 || c->arena_cache[i].second < c->arena_cache[best].second))
                    best = (int)i;
            if (best >= 0) {
                void *p = c->arena_cache[best].first;
                *cap = c->arena_cache[best].second;
                c->arena_cache.erase(c->arena_cache.begin() + best);
                c->arena_in_use += *cap;
                return p;
            }
        }
        void *p = nullptr;
        /* head room so that the next, slightly larger list fits -- until memory has been tight once */
        *cap = bytes + (c->hbm_tight.load() ? 0 : bytes / 8) + 4096;
        bool starve = false;
        {
            /* blocks in use + blocks waiting for re-use + this one stay under ~7/8 of the device: the OLDEST waiting blocks
             * go back to the driver first (a failed hipMalloc costs a flush of the whole cache and, with several lists in
             * flight, sends the pipeline into a slow mode: mixed config, 6 lists of 27 GB: 157 k or 56 k groups/s) */
            std::vector<void *> drop;
            {
                std::lock_guard<std::mutex> lk(c->arena_mu);
                size_t held = 0;
                for (auto &a : c->arena_cache) held += a.second;
                const size_t budget = c->hbm_bytes - c->hbm_bytes / 8;
                /* ... and what the driver reports as free covers the request with 1/32 of the device to spare: the preparation pools of the
                 * lanes, the staged records and other processes' memory are not in arena_in_use (round 5: with six lanes the blocks of two
                 * earlier 4 096-group ONT lists waited for re-use while the pipeline's lists failed their hipMalloc -- 424 instead of 325 ms per step) */
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = ~(size_t)0 >> 1; }
                while (!c->arena_cache.empty() && (c->arena_in_use + held + *cap > budget || free_b < *cap + c->hbm_bytes / 32)) {
                    free_b += c->arena_cache.front().second;
                    held -= c->arena_cache.front().second;
                    drop.push_back(c->arena_cache.front().first);
                    c->arena_cache.erase(c->arena_cache.begin());
                }
                /* never the device's last bytes while older lists hold memory they will give back: the HIP runtime allocates from the same
                 * memory (kernel scratch, queues) and ends the process with HSA_STATUS_ERROR_OUT_OF_RESOURCES when it finds none (six
                 * preparation lanes' pools beside four 65 536-group mixed lists) */
                starve = c->arena_in_use > 0 && free_b < *cap + c->hbm_bytes / 64;
            }
            for (void *d : drop) (void)hipFree(d);
        }
        if (starve && now_s() < t_end) {
            c->hbm_tight = true;
            std::unique_lock<std::mutex> lk(c->arena_mu);
            c->arena_cv.wait_for(lk, std::chrono::milliseconds(50));
            continue;
        }
        const double tm0 = now_s();
        const hipError_t me = hipMalloc(&p, *cap);
        if (timing_on()) {
            size_t held = 0;
            std::lock_guard<std::mutex> lk(c->arena_mu);
            for (auto &a : c->arena_cache) held += a.second;
            fprintf(stderr, "[spx timing] hipMalloc of %.2f GB: %.3f s%s (in use %.1f GB, %zu blocks / %.1f GB waiting for re-use)\n", *cap / 1e9,
                    now_s() - tm0, me == hipSuccess ? "" : " (failed)", c->arena_in_use / 1e9, c->arena_cache.size(), held / 1e9);
        }
        if (me == hipSuccess) {
            std::lock_guard<std::mutex> lk(c->arena_mu);
            c->arena_in_use += *cap;
            return p;
        }
        (void)hipGetLastError();
        c->hbm_tight = true;
        std::unique_lock<std::mutex> lk(c->arena_mu);
        /* give cached blocks (all too small) back to the driver and try once more, without the head room */
        for (auto &a : c->arena_cache) (void)hipFree(a.first);
        c->arena_cache.clear();
        *cap = bytes + 4096;
        if (hipMalloc(&p, *cap) == hipSuccess) { c->arena_in_use += *cap; return p; }
        (void)hipGetLastError();
        if (now_s() >= t_end) return nullptr;
        c->arena_cv.wait_for(lk, std::chrono::milliseconds(250));
    }
}
static void arena_put(spx_ctx *c, void *p, size_t cap)
{
    if (!p) return;
    if (c) {
        std::vector<void *> drop;
        {
            /* at most 8 blocks and half of the device's memory wait for re-use; the OLDEST waiting blocks make room (the
             * sizes in demand change with the workload: a cache full of blocks nobody asks for any more would send every
             * request to hipMalloc) */
            std::lock_guard<std::mutex> lk(c->arena_mu);
            c->arena_in_use -= std::min(c->arena_in_use, cap);
            if (cap <= c->hbm_bytes / 2) {
                size_t held = cap;
                for (auto &a : c->arena_cache) held += a.second;
                while (!c->arena_cache.empty() && (c->arena_cache.size() >= 8 || held > c->hbm_bytes / 2 ||
                                                   c->arena_in_use + held > c->hbm_bytes - c->hbm_bytes / 8)) {
                    held -= c->arena_cache.front().second;
                    drop.push_back(c->arena_cache.front().first);
                    c->arena_cache.erase(c->arena_cache.begin());
                }
                c->arena_cache.emplace_back(p, cap);
                p = nullptr;
            }
        }
        for (void *d : drop) (void)hipFree(d);
        if (p) (void)hipFree(p);
        c->arena_cv.notify_all();
        return;
    }
    (void)hipFree(p);
}
static void *pinned_get(spx_ctx *c, size_t bytes, size_t *cap)
{
    {
        std::lock_guard<std::mutex> lk(c->arena_mu);
        int best = -1;
        for (size_t i = 0; i < c->pinned_cache.size(); ++i)
            if (c->pinned_cache[i].second >= bytes && (best < 0 || c->pinned_cache[i].second < c->pinned_cache[best].second))
                best = (int)i;
        if (best >= 0) {
            void *p = c->pinned_cache[best].first;
            *cap = c->pinned_cache[best].second;
            c->pinned_cache.erase(c->pinned_cache.begin() + best);
            return p;
        }
    }
    void *p = nullptr;
    *cap = bytes + bytes / 8 + 4096;
    if (hipHostMalloc(&p, *cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
static void pinned_put(spx_ctx *c, void *p, size_t cap)
{
    if (!p) return;
    if (c) {
        std::lock_guard<std::mutex> lk(c->arena_mu);
        if (c->pinned_cache.size() < 4) { c->pinned_cache.emplace_back(p, cap); return; }
    }
    (void)hipHostFree(p);
}

/* host half: dispatch filter, payload sizes, which records repeat another one (no device call: the pipeline runs it
 * for submission k+1 while the records of submission k are on their way over PCIe) */
extern "C" int spx_internal_stage_begin(spx_ctx *c, const spx_batch *const *bts, int32_t n_batches, const spx_params *par, int host_threads,
                                        spx_work **out)
{
    if (!c || !bts || n_batches <= 0 || !par || !out) return fail(SPX_EINVAL, "NULL argument");
    if (!c->d_ref4) return fail(SPX_ENOREF, "spx_set_reference has not been called");
    *out = nullptr;
    spx_work *w = new spx_work();
    memset(&w->st, 0, sizeof w->st);
    memset(&w->pa, 0, sizeof w->pa);
    memset(&w->tot, 0, sizeof w->tot);
    w->par = *par;
    w->owner = c;
    const double t0 = now_s();
    int nthr = host_threads > 0 ? host_threads : spx::effective_cpus();
    nthr = std::max(1, std::min(nthr, 128));
    int rc = spx::stage_measure(bts, n_batches, nthr, w->stage);
    if (rc) { delete w; return fail(rc, "invalid batch"); }
    const spx::StageLayout &L = w->stage.lay;
    w->n_groups_in = (int32_t)L.n_groups_in;
    w->n_dgroups = (int32_t)L.n_dgroups;
    w->hb.grp_error = w->stage.grp_error;
    if (L.n_slots > SPX_MAX_STAGE_SLOTS || L.n_dgroups > SPX_MAX_STAGE_SLOTS) {
        delete w;
        return fail(SPX_EINVAL, "more than 2^20 alignments in one work list: stage fewer groups at a time");
    }
    w->st.prep_seconds = now_s() - t0;
    w->st.n_groups = w->n_groups_in;
    w->stage_threads = nthr;
    *out = w;
    return SPX_OK;
}

/* device half: memory for the image, the payload through the ring of pinned chunks, unpack / rebuild kernels.  On failure
 * the caller frees the list. */
extern "C" int spx_internal_stage_finish(spx_ctx *c, spx_work *w)
{
    if (!c || !w || w->staged) return fail(SPX_EINVAL, "work list is not waiting for its records");
    HIPCHK(hipSetDevice(c->device));
    const spx::StageLayout &L = w->stage.lay;
    const int nthr = w->stage_threads;
    const double t0 = now_s();
    /* part A in HBM: staged records | recoded SEQ | per-alignment state | per-group counts and offsets */
    const double t_meas = now_s();
    Carver cv;
    const size_t ns = (size_t)L.n_slots, ng = (size_t)L.n_dgroups;
    const size_t o_in = cv.take<char>(L.bytes + 64);
    w->o_code = cv.take<char>((size_t)(spx::kCodeLeadBytes + L.seq_bytes + spx::kCodeTailBytes));
    w->o_ast = cv.take<spxl::AlnState>(ns + 1);
    w->o_gc = cv.take<spxl::GroupCount>(ng + 1);
    w->o_ac = cv.take<spxl::GroupCount>(ns + 1);
    w->o_gab = cv.take<int64_t>(ng + 1);
    w->o_gao = cv.take<int64_t>(ng + 1);
    w->o_base = cv.take<spxl::PlanBase>(ns + 1);
    w->o_mkb = cv.take<int64_t>(ng + 2);
    w->o_scan = cv.take<int64_t>(5 * (std::max(ns, ng) + 8) + 5 * 1024 + 16);
    /* packed SEQ / QUAL as they cross PCIe (aliased secondaries left out); unpacked into the image's pools on the device */
    const size_t o_pkseq = cv.take<char>((size_t)L.pk_seq_bytes + 64), o_pkqual = cv.take<char>((size_t)L.pk_qual_bytes + 64);
    (void)o_in;
    w->in_arena = arena_get(c, cv.off + 256, &w->in_cap);
    if (!w->in_arena) return fail(SPX_ENOMEM, "device memory for the staged records");
    char *base = (char *)w->in_arena;
    const double t_arena = now_s();
    {
        /* through the ring of pinned chunks: fill chunk k on the pool while chunk k-1 is on its way over PCIe */
        std::lock_guard<std::mutex> sl(c->stage_mu);
        if (!c->stage_pool || c->stage_pool->size() < nthr) c->stage_pool.reset(new spx::Pool(nthr));
        spx::Pool *pool = c->stage_pool.get();
        const std::function<void(int64_t, int64_t, const std::function<void(int64_t, int64_t)> &)> par_for =
            [pool](int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)> &fn) { pool->parallel_for(n, grain, fn); };
        int rc2 = SPX_OK;
        auto chunk_get = [&](char **p) -> int { /* next chunk of the ring, free again */
            const int k = c->pin_next;
            c->pin_next = (k + 1) % spx_ctx::SPX_PIN_CHUNKS;
            if (!c->pin_chunk[k]) {
                if (hipHostMalloc(&c->pin_chunk[k], c->pin_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->pin_chunk[k] = nullptr; return -1; }
                if (hipEventCreateWithFlags(&c->pin_done[k], hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) return -1;
            }
            if (c->pin_busy[k] && hipEventSynchronize(c->pin_done[k]) != hipSuccess) return -1;
            c->pin_busy[k] = false;
            *p = (char *)c->pin_chunk[k];
            return k;
        };
        auto chunk_send = [&](int k, char *dev_dst, size_t bytes) -> bool {
            if (hipMemcpyAsync(dev_dst, c->pin_chunk[k], bytes, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess) return false;
            if (hipEventRecord(c->pin_done[k], c->copy_stream) != hipSuccess) return false;
            c->pin_busy[k] = true;
            return true;
        };
        /* the small arrays: record descriptors, group -> first alignment, group -> input group */
        struct Raw { const void *src; size_t bytes, off; };
        const Raw raws[3] = {{w->stage.recs.data(), w->stage.recs.size() * sizeof(spxl::Rec), L.o_recs},
                             {w->stage.slot0.data(), w->stage.slot0.size() * 4, L.o_slot0},
                             {w->stage.grp_index.data(), w->stage.grp_index.size() * 4, L.o_gidx}};
        for (const Raw &r : raws)
            for (size_t o = 0; o < r.bytes && rc2 == SPX_OK; o += c->pin_bytes) {
                char *h = nullptr;
                const int k = chunk_get(&h);
                if (k < 0) { rc2 = SPX_ENOMEM; break; }
                const size_t n = std::min(c->pin_bytes, r.bytes - o);
                const char *src = (const char *)r.src + o;
                if (n > ((size_t)8 << 20)) pool->parallel_for((int64_t)n, (int64_t)4 << 20, [&](int64_t a, int64_t b) { memcpy(h + a, src + a, (size_t)(b - a)); });
                else memcpy(h, src, n);
                if (!chunk_send(k, base + r.off + o, n)) rc2 = SPX_EHIP;
            }
        static const int kSections[4] = {0, 4, 5, 3}; /* CIGAR words, packed SEQ, packed QUAL, tag text */
        for (int si = 0; si < 4 && rc2 == SPX_OK; ++si) {
            const int sec = kSections[si];
            const int64_t tot = spx::stage_section_bytes(w->stage, sec);
            const size_t doff = sec == 4 ? o_pkseq : sec == 5 ? o_pkqual : spx::stage_section_offset(w->stage, sec);
            for (int64_t b0 = 0; b0 < tot && rc2 == SPX_OK; b0 += (int64_t)c->pin_bytes) {
                const int64_t b1 = std::min<int64_t>(tot, b0 + (int64_t)c->pin_bytes);
                char *h = nullptr;
                const int k = chunk_get(&h);
                if (k < 0) { rc2 = SPX_ENOMEM; break; }
                spx::stage_fill(w->stage, sec, b0, b1, h, par_for);
                if (!chunk_send(k, base + doff + b0, (size_t)(b1 - b0))) rc2 = SPX_EHIP;
            }
        }
        if (rc2 != SPX_OK) return fail(rc2, rc2 == SPX_ENOMEM ? "pinned staging chunk" : "copy of the staged records");
    }
    {
        hipEvent_t ev_copied = nullptr; /* the unpack stream goes on behind the last chunk */
        HIPCHK(hipEventCreateWithFlags(&ev_copied, hipEventDisableTiming | hipEventBlockingSync));
        hipError_t e = hipEventRecord(ev_copied, c->copy_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->unpack_stream, ev_copied, 0);
        (void)hipEventDestroy(ev_copied); /* (released by the runtime once it has completed) */
        HIPCHK(e);
    }
    HIPCHK(spx_stage_expand(base + L.o_recs, (int32_t)L.n_slots, (const uint8_t *)base + o_pkseq, (const uint8_t *)base + o_pkqual,
                            (uint8_t *)base + L.o_seq, (uint8_t *)base + L.o_qual, L.n_aliased > 0, c->unpack_stream));
    const double t1 = now_s();
    /* the recoded-SEQ pool: recode_kernel writes every word behind the lead pad, only the pads need a value (they are read,
     * never used); zeroing the whole pool was a 3.7 GB fill per 131 072 HiFi groups */
    HIPCHK(hipMemsetAsync(base + w->o_code, 0, (size_t)spx::kCodeLeadBytes, c->unpack_stream));
    HIPCHK(hipMemsetAsync(base + w->o_code + spx::kCodeLeadBytes + ((L.seq_bytes + 3) & ~(int64_t)3), 0, (size_t)spx::kCodeTailBytes - 8, c->unpack_stream));
    HIPCHK(hipEventCreateWithFlags(&w->ev_staged, hipEventDisableTiming | hipEventBlockingSync));
    HIPCHK(hipEventRecord(w->ev_staged, c->unpack_stream));
    w->st.prep_seconds += t1 - t0;
    w->st.bytes_h2d = (int64_t)L.bytes - (L.seq_bytes - L.pk_seq_bytes) - (L.qual_bytes - L.pk_qual_bytes); /* what really crossed PCIe */
    w->staged = true;
    if (timing_on())
        fprintf(stderr, "[spx timing] stage: %d threads, measure %.3f s, device memory %.3f s, fill + copy %.3f s, %.1f MB\n", nthr,
                w->st.prep_seconds - (t1 - t0), t_arena - t_meas, t1 - t_arena, w->st.bytes_h2d / 1e6);
    return SPX_OK;
}

/* Step 1 of a work list: dispatch filter + staging of the dispatched groups' records into pinned memory (host
 * threads), copies into HBM.  Several record batches (e.g. the blocks a reader thread hands over) become
 * ONE work list; group g of batch b is reported at index (groups of batches < b) + g. */
extern "C" int spx_stage(spx_ctx *c, const spx_batch *const *bts, int32_t n_batches, const spx_params *par, int host_threads,
                         spx_work **out)
{
    if (out) *out = nullptr;
    spx_work *w = nullptr;
    int rc = spx_internal_stage_begin(c, bts, n_batches, par, host_threads, &w);
    if (rc != SPX_OK) return rc;
    rc = spx_internal_stage_finish(c, w);
    if (rc != SPX_OK) { spx_work_free(c, w); return rc; }
    *out = w;
    return SPX_OK;
}

/* ---- records staged BY THE DEVICE (spx_devin.cpp: BGZF blocks inflated in HBM, record chain / fields / tags / dispatch
 * filter / gather as kernels).  begin: a work list with the image's layout from the device-side counts and the memory of
 * part A; the caller's kernels then write the image (O points into it) and `info_bytes` of host-bound data behind it.
 * finish: the host-side vectors from the dispatch flags, pad fills and the staged event on the caller's stream. ---- */
struct spx_devstage_sizes {
    int64_t n_groups_in, n_dgroups, n_slots, cigar_words, seq_bytes, qual_bytes, text_bytes, ops_bound, conf_bound, mm_bound, info_bytes;
};
extern "C" int spx_internal_devstage_begin(spx_ctx *c, const spx_params *par, const spx_devstage_sizes *sz, spx_work **out, spx_din_out *O, char **d_info)
{
    if (!c || !par || !sz || !out || !O || !d_info) return fail(SPX_EINVAL, "NULL argument");
    /* (the reference is not needed to STAGE records: the input pipelines may start while it is still on its way to the device;
     * spx_prepare_staged asks for it) */
    *out = nullptr;
    if (sz->n_slots > SPX_MAX_STAGE_SLOTS || sz->n_dgroups > SPX_MAX_STAGE_SLOTS || sz->n_groups_in > 0x7fffffff)
        return fail(SPX_EINVAL, "more than 2^20 alignments in one work list");
    HIPCHK(hipSetDevice(c->device));
    spx_work *w = new spx_work();
    memset(&w->st, 0, sizeof w->st);
    memset(&w->pa, 0, sizeof w->pa);
    memset(&w->tot, 0, sizeof w->tot);
    w->par = *par;
    w->owner = c;
    spx::StageLayout &L = w->stage.lay;
    L = spx::StageLayout();
    L.n_groups_in = sz->n_groups_in; L.n_dgroups = sz->n_dgroups; L.n_slots = sz->n_slots;
    L.cigar_words = sz->cigar_words; L.seq_bytes = sz->seq_bytes; L.qual_bytes = sz->qual_bytes; L.text_bytes = sz->text_bytes;
    L.pk_seq_bytes = sz->seq_bytes; L.pk_qual_bytes = sz->qual_bytes;
    L.ops_bound = sz->ops_bound; L.conf_bound = sz->conf_bound; L.mm_bound = sz->mm_bound;
    spx::stage_layout_offsets(L);
    w->n_groups_in = (int32_t)L.n_groups_in;
    w->n_dgroups = (int32_t)L.n_dgroups;
    w->st.n_groups = w->n_groups_in;
    w->stage_threads = 1;
    Carver cv;
    const size_t ns = (size_t)L.n_slots, ng = (size_t)L.n_dgroups;
    (void)cv.take<char>(L.bytes + 64);
    w->o_code = cv.take<char>((size_t)(spx::kCodeLeadBytes + L.seq_bytes + spx::kCodeTailBytes));
    w->o_ast = cv.take<spxl::AlnState>(ns + 1);
    w->o_gc = cv.take<spxl::GroupCount>(ng + 1);
    w->o_ac = cv.take<spxl::GroupCount>(ns + 1);
    w->o_gab = cv.take<int64_t>(ng + 1);
    w->o_gao = cv.take<int64_t>(ng + 1);
    w->o_base = cv.take<spxl::PlanBase>(ns + 1);
    w->o_mkb = cv.take<int64_t>(ng + 2);
    w->o_scan = cv.take<int64_t>(5 * (std::max(ns, ng) + 8) + 5 * 1024 + 16);
    const size_t o_info = cv.take<char>((size_t)sz->info_bytes + 64);
    w->in_arena = arena_get(c, cv.off + 256, &w->in_cap);
    if (!w->in_arena) { delete w; return fail(SPX_ENOMEM, "device memory for the staged records"); }
    char *base = (char *)w->in_arena;
    memset(O, 0, sizeof *O);
    O->recs = (spxl::Rec *)(base + L.o_recs);
    O->slot0 = (int32_t *)(base + L.o_slot0);
    O->gidx = (int32_t *)(base + L.o_gidx);
    O->cigar = (uint32_t *)(base + L.o_cigar);
    O->seq = (uint8_t *)(base + L.o_seq);
    O->qual = (uint8_t *)(base + L.o_qual);
    O->text = base + L.o_text;
    *d_info = base + o_info;
    *out = w;
    return SPX_OK;
}

extern "C" int spx_internal_devstage_finish(spx_ctx *c, spx_work *w, const uint8_t *grp_disp, hipStream_t st)
{
    if (!c || !w || w->staged || !w->in_arena) return fail(SPX_EINVAL, "work list is not waiting for its records");
    HIPCHK(hipSetDevice(c->device));
    const spx::StageLayout &L = w->stage.lay;
    w->stage.grp_error.assign((size_t)L.n_groups_in, 1);
    w->stage.grp_index.clear();
    w->stage.grp_index.reserve((size_t)L.n_dgroups);
    for (int64_t g = 0; g < L.n_groups_in; ++g)
        if (grp_disp[g]) { w->stage.grp_error[(size_t)g] = 0; w->stage.grp_index.push_back((int32_t)g); }
    if ((int64_t)w->stage.grp_index.size() != L.n_dgroups) return fail(SPX_EINVAL, "dispatch flags do not match the device's count");
    w->hb.grp_error = w->stage.grp_error;
    char *base = (char *)w->in_arena;
    HIPCHK(hipMemsetAsync(base + w->o_code, 0, (size_t)spx::kCodeLeadBytes, st));
    HIPCHK(hipMemsetAsync(base + w->o_code + spx::kCodeLeadBytes + ((L.seq_bytes + 3) & ~(int64_t)3), 0, (size_t)spx::kCodeTailBytes - 8, st));
    HIPCHK(hipEventCreateWithFlags(&w->ev_staged, hipEventDisableTiming | hipEventBlockingSync));
    HIPCHK(hipEventRecord(w->ev_staged, st));
    w->st.bytes_h2d = (int64_t)L.bytes;
    w->staged = true;
    return SPX_OK;
}

static void fill_prep_args(spx_ctx *c, spx_work *w)
{
    const spx::StageLayout &L = w->stage.lay;
    char *base = (char *)w->in_arena;
    spx_prep_args &A = w->pa;
    memset(&A, 0, sizeof A);
    A.n_slots = (int32_t)L.n_slots;
    A.n_dgroups = (int32_t)L.n_dgroups;
    A.recs = (const spxl::Rec *)(base + L.o_recs);
    A.slot0 = (const int32_t *)(base + L.o_slot0);
    A.ast = (spxl::AlnState *)(base + w->o_ast);
    A.P.cigar = (const uint32_t *)(base + L.o_cigar);
    A.P.qual = (const uint8_t *)(base + L.o_qual);
    A.P.text = (const char *)(base + L.o_text);
    A.P.code4 = (const uint8_t *)(base + w->o_code);
    A.P.code_lead_bytes = spx::kCodeLeadBytes;
    A.code4_w = (uint8_t *)(base + w->o_code);
    spx_ctx::PrepLane &PL = c->lane[w->lane];
    A.P.ops = (spxl::Op *)PL.pool_ops.p;
    A.P.conf = (spxl::Blk *)PL.pool_conf.p;
    A.P.mm = (spxl::MM *)PL.pool_mm.p;
    A.rv = c->d_rv;
    A.par = spx::logic_params(&w->par);
    /* two-tier DP: a wanted row holds the fast tier's forward AND backward rows -- unless the lists in front of this one decided against the tiers */
    if (c->tiers_epoch.exchange(g_dp_tiers_epoch.load()) != g_dp_tiers_epoch.load()) c->tiers_hint.store(-1);
    w->tiers_look = spx_get_dp_tiers() != 0 && (c->tiers_hint.load() != 0 || (c->tiers_looks.fetch_add(1) % 32) == 31);
    if (w->tiers_look) A.par.row_mult = 4;
    A.gc = (spxl::GroupCount *)(base + w->o_gc);
    A.ac = (spxl::GroupCount *)(base + w->o_ac);
    A.ga_bytes = (int64_t *)(base + w->o_gab);
    A.ga_off = (int64_t *)(base + w->o_gao);
    A.arena = (char *)PL.pool_garena.p;
    A.arena_cap = (int64_t)PL.pool_garena.cap;
    A.slack = c->slack_hint.load(std::memory_order_relaxed);
    A.scan_stride = (int64_t)std::max(L.n_slots, L.n_dgroups) + 8;
    A.scan_v = (int64_t *)(base + w->o_scan);
    A.scan_tile = A.scan_v + 5 * A.scan_stride;
    A.scan_grand = A.scan_tile + 5 * 1024;
    A.tot = PL.d_tot;
}

/* Step 2: the work list is built ON THE DEVICE from the staged records (spx_prep_kernels.hip).  One small copy of the
 * sizes comes back in the middle (scratch and output arrays are carved to measure); everything else is asynchronous on
 * the context's preparation stream.  May be called again on the same staged records (bench.py re-prepares resident
 * batches: the timed step then covers the whole path, records -> decisions). */
extern "C" int spx_prepare_staged(spx_ctx *c, spx_work *w)
{
    if (!c || !w || !w->staged) return fail(SPX_EINVAL, "work list has not been staged");
    if (!c->d_ref4) return fail(SPX_ENOREF, "spx_set_reference has not been called");
    HIPCHK(hipSetDevice(c->device));
    const spx::StageLayout &L = w->stage.lay;
    const size_t ns = (size_t)L.n_slots, ng = (size_t)L.n_dgroups;
    /* whatever happens below, this list's turn at the pipeline's allocation gate is used up when we leave */
    struct GateGuard {
        spx_work *w;
        bool passed = false;
        void wait()
        {
            if (!w->gate || passed) return;
            std::unique_lock<std::mutex> lk(w->gate->mu);
            w->gate->cv.wait(lk, [&] { return w->gate->turn >= w->gate_ticket; });
        }
        void pass()
        {
            if (!w->gate || passed) return;
            passed = true;
            std::lock_guard<std::mutex> lk(w->gate->mu);
            if (w->gate->turn == w->gate_ticket) { ++w->gate->turn; w->gate->cv.notify_all(); }
        }
        ~GateGuard() { if (w->gate && !passed) { wait(); pass(); } }
    } gate{w};
    w->lane = (int)(c->lane_rr.fetch_add(1) % (unsigned)c->n_prep);
    if (w->gate) {
        w->lane = (int)(w->gate_ticket % c->n_prep);
        /* Lists t and t - n_prep share a lane.  The older one waits at the allocation gate while it HOLDS the lane; were the
         * younger one to take the lane first (the older one preempted on its way to the lock), it would wait at the gate
         * for a turn the older one can never pass.  So a list asks for its lane only once the list n_prep tickets ahead
         * has passed the gate -- from there on that one waits for nothing this one holds. */
        std::unique_lock<std::mutex> gl(w->gate->mu);
        w->gate->cv.wait(gl, [&] { return w->gate->turn > w->gate_ticket - c->n_prep; });
    }
    spx_ctx::PrepLane &PL = c->lane[w->lane];
    std::lock_guard<std::mutex> lk(PL.mu);
    const double t0 = now_s();
    if (w->arena) { /* re-preparation: the previous list of this work must have left the device */
        if (w->ev_done) HIPCHK(hipEventSynchronize(w->ev_done));
        else HIPCHK(hipStreamSynchronize(c->stream));
        arena_put(c, w->arena, w->arena_cap);
        work_arena_back(c);
        w->arena = nullptr;
    }
    w->prepared = false;
    /* tables of the per-alignment pass: by the per-alignment bounds of spxl::aln_caps (the device carves them the same
     * way and parses every tag once); SPX_PREP_EXACT=1 sizes them with the counting pass instead (two parses) */
    size_t ops_bound = (size_t)L.ops_bound + 16, conf_bound = (size_t)L.conf_bound + 16, mm_bound = (size_t)L.mm_bound + 16;
    int rc;
    auto size_pools = [&]() -> int {
        int r_;
        if ((r_ = ensure_pool(PL, PL.pool_ops, ops_bound * sizeof(spxl::Op))) || (r_ = ensure_pool(PL, PL.pool_conf, conf_bound * sizeof(spxl::Blk))) ||
            (r_ = ensure_pool(PL, PL.pool_mm, mm_bound * sizeof(spxl::MM))))
            return r_;
        return 0;
    };
    if ((rc = size_pools())) return fail(rc, "device memory for the preparation pools");
    if ((rc = ensure_pool(PL, PL.pool_garena, (size_t)(32u << 20) + ns * 4096))) return fail(rc, "device memory for the group scratch");
    fill_prep_args(c, w);
    spx_prep_args &A = w->pa;
    A.ops_cap = (int64_t)ops_bound; A.conf_cap = (int64_t)conf_bound; A.mm_cap = (int64_t)mm_bound;
    A.exact_counts = getenv("SPX_PREP_EXACT") ? 1 : 0;
    A.tight_caps = getenv("SPX_PREP_TIGHT") ? 1 : 0;
    char *base = (char *)w->in_arena;
    spxl::PlanBase *d_base = (spxl::PlanBase *)(base + w->o_base);
    int64_t *d_mkb = (int64_t *)(base + w->o_mkb);
    if (w->ev_staged) HIPCHK(hipStreamWaitEvent(PL.stream, w->ev_staged, 0));
    HIPCHK(hipMemsetAsync(PL.d_tot, 0, sizeof(spx_prep_totals), PL.stream));
    /* the heaviest alignments / groups of the list walk alone in waves of their own (spx_prep_dev.h): SPX_PREP_HEAVY = how many alignments at
     * most (default 1024; half as many groups; 0: none), SPX_PREP_HEAVY_MIN = the work estimate (tag characters + 4 x CIGAR operations) from
     * which an item counts as heavy (default 2048: a HiFi read of 15 kb stays where it is, an ONT-like read of 20 kb moves out) */
    static const int heavy = [] { const char *e = getenv("SPX_PREP_HEAVY"); return e ? std::max(0, atoi(e)) : 1024; }();
    static const int heavy_min = [] { const char *e = getenv("SPX_PREP_HEAVY_MIN"); return e ? std::max(1, atoi(e)) : 2048; }();
    static const int share = [] { const char *e = getenv("SPX_PREP_SHARE"); return e ? std::max(0, atoi(e)) : 32768; }();
    if (heavy > 0 && ns >= 64) {
        const size_t n = std::max(ns, ng), tb = spx_prep_heavy_temp_bytes((int32_t)n);
        A.n_heavy_slots = (int32_t)std::min<size_t>((size_t)heavy, ns / 16);
        A.n_heavy_groups = (int32_t)std::min<size_t>((size_t)heavy / 2, ng / 16);
        /* the kernels that SHARE an extracted item among the 64 lanes of its wave extract many more (SPX_PREP_SHARE, default 32 768, at most a quarter of the list) */
        A.n_share_slots = (int32_t)std::max<size_t>((size_t)A.n_heavy_slots, std::min<size_t>((size_t)share, ns / 4));
        A.n_share_groups = (int32_t)std::max<size_t>((size_t)A.n_heavy_groups, std::min<size_t>((size_t)share / 2, ng / 4));
        Carver pc;
        const size_t o_keys = pc.take<int32_t>(3 * n), o_vals = pc.take<int32_t>(2 * n), o_sh = pc.take<int32_t>(ns + 1), o_gh = pc.take<int32_t>(ng + 1),
                     o_sf = pc.take<uint8_t>(ns + 1), o_gf = pc.take<uint8_t>(ng + 1), o_tmp = pc.take<char>(tb + 256),
                     o_hp = pc.take<spxl::PlanBase>((size_t)A.n_share_slots * 64 + 64);
        if ((rc = ensure_pool(PL, PL.pool_heavy, pc.off + 256))) return fail(rc, "device memory for the list of heavy alignments");
        char *pb = (char *)PL.pool_heavy.p;
        HIPCHK(spx_prep_heavy(&A, (int32_t *)(pb + o_keys), (int32_t *)(pb + o_vals), pb + o_tmp, tb, (int32_t *)(pb + o_sh), (int32_t *)(pb + o_gh),
                              (uint8_t *)(pb + o_sf), (uint8_t *)(pb + o_gf), heavy_min, PL.stream));
        A.slot_heavy = (const int32_t *)(pb + o_sh);
        A.group_heavy = (const int32_t *)(pb + o_gh);
        A.slot_flag = (const uint8_t *)(pb + o_sf);
        A.group_flag = (const uint8_t *)(pb + o_gf);
        A.heavy_plan = (spxl::PlanBase *)(pb + o_hp);
    }
    bool phase1 = true;
    for (int attempt = 0;; ++attempt) {
        /* (the per-alignment state is initialised by the first kernel of phase 1, not by a fill launch) */
        if (phase1) HIPCHK(spx_prep_phase1(&A, (const uint32_t *)(base + L.o_seq), (L.seq_bytes + 3) / 4, PL.stream));
        HIPCHK(spx_prep_phase2(&A, d_base, d_mkb, PL.stream));
        HIPCHK(hipMemcpyAsync(PL.h_tot, PL.d_tot, sizeof(spx_prep_totals), hipMemcpyDeviceToHost, PL.stream));
        HIPCHK(hipStreamSynchronize(PL.stream));
        w->tot = *PL.h_tot;
        const int ov = w->tot.overflow;
        if (!ov) break;
        if (attempt >= 8) return fail(SPX_ENOMEM, "preparation scratch keeps overflowing");
        phase1 = (ov & (4 | 8)) != 0;
        if (ov & 8) { /* an alignment outgrew the length bounds: count exactly, then carve */
            A.exact_counts = 1;
        } else if (ov & 4) { /* exact sizes are known now (the prefix sums of the counting pass) */
            ops_bound = std::max(ops_bound, (size_t)w->tot.n_ops + 16);
            conf_bound = std::max(conf_bound, (size_t)w->tot.n_conf + 16);
            mm_bound = std::max(mm_bound, (size_t)w->tot.n_mm + 16);
            if ((rc = size_pools())) return fail(rc, "device memory for the preparation pools");
            A.P.ops = (spxl::Op *)PL.pool_ops.p; A.P.conf = (spxl::Blk *)PL.pool_conf.p; A.P.mm = (spxl::MM *)PL.pool_mm.p;
            A.ops_cap = (int64_t)ops_bound; A.conf_cap = (int64_t)conf_bound; A.mm_cap = (int64_t)mm_bound;
        } else if (ov & 2) {
            A.slack *= 4;
            { const int want = std::min(A.slack, 64); int h = c->slack_hint.load(std::memory_order_relaxed); while (h < want && !c->slack_hint.compare_exchange_weak(h, want)) {} }
            if ((rc = ensure_pool(PL, PL.pool_garena, (size_t)w->tot.arena_bytes * 4 + 4096))) return fail(rc, "device memory for the group scratch");
            A.arena = (char *)PL.pool_garena.p;
            A.arena_cap = (int64_t)PL.pool_garena.cap;
        } else {
            if ((rc = ensure_pool(PL, PL.pool_garena, (size_t)w->tot.arena_bytes + 4096))) return fail(rc, "device memory for the group scratch");
            A.arena = (char *)PL.pool_garena.p;
            A.arena_cap = (int64_t)PL.pool_garena.cap;
        }
        HIPCHK(hipMemsetAsync(PL.d_tot, 0, sizeof(spx_prep_totals), PL.stream));
    }
    if (getenv("SPX_DEBUG_GC")) { /* debugging aid: the per-group and per-alignment state the device arrived at */
        std::vector<spxl::GroupCount> hg(ng);
        std::vector<spxl::AlnState> ha(ns);
        (void)hipMemcpy(hg.data(), A.gc, ng * sizeof(spxl::GroupCount), hipMemcpyDeviceToHost);
        (void)hipMemcpy(ha.data(), A.ast, ns * sizeof(spxl::AlnState), hipMemcpyDeviceToHost);
        for (size_t k = 0; k < ng; ++k) {
            fprintf(stderr, "[spx debug] group %zu: err %d scored %d n_cols %d n_prob %d n_rows %d |", k, hg[k].err, hg[k].scored, hg[k].n_cols,
                    hg[k].n_prob, hg[k].n_rows);
            for (int32_t q = w->stage.slot0[k]; q < w->stage.slot0[k + 1]; ++q)
                fprintf(stderr, " [ops %d visit %d conf %d mm %d err %d rds %d rde %d rfs %d rfe %d]", ha[q].n_ops, ha[q].n_visit, ha[q].n_conf, ha[q].n_mm,
                        ha[q].err, ha[q].rds, ha[q].rde, ha[q].rfs, ha[q].rfe);
            fprintf(stderr, "\n");
        }
    }
    const double t1 = now_s();
    /* ---- part B: the work list, its scratch and its outputs, carved to measure ---- */
    const spx_prep_totals &T = w->tot;
    const size_t np = (size_t)T.n_prob, nr = (size_t)T.n_rows, nq = (size_t)T.n_qe, nm = (size_t)T.n_mk;
    if (np > 0x7ffffff0u || nr > 0x7ffffff0u || nm > 0x7ffffff0u) return fail(SPX_EINVAL, "work list too large: stage fewer groups at a time");
    /* ---- DP slices: K ranges of consecutive groups that share ONE scratch area.  K from the scratch the list
     * would need (SPX_DP_SLICE_GB per slice, default 16, 24 with the two-tier DP; SPX_DP_SLICES forces a count); >= 256 groups per slice ---- */
    int K = 1;
    {
        const double scratch_gb = ((double)T.s_tot + (double)T.f_tot) * 8.0 / 1e9;
        /* (two-tier DP: a wanted row holds four rows of slots instead of two; HiFi lists of 131 072 groups: 4 slices at 16 GB, 3 at 24 -- 1.19 /
         * 1.20 M groups/s on one box; 2 at 32 GB measured 1.23 M once and 0.87 / 1.21 M against 1.32 / 1.31 M later: four lists of that size wait for memory.  With more than five lists holding memory at once the slices stay at 16 GB: the mixed leg's nine lists of
         * two 17 GB slices each filled the device and the pipeline waited for memory, 118 k groups/s instead of 212 k) */
        const int lists = std::max(c->work_arenas_most.load(), c->work_arenas.load() + 1);
        double budget = (A.par.row_mult > 2 && lists <= 5) ? 24.0 : 16.0;
        if (const char *e = getenv("SPX_DP_SLICE_GB")) budget = std::max(0.001, atof(e));
        if (scratch_gb > budget) K = (int)ceil(scratch_gb / budget);
        if (const char *e = getenv("SPX_DP_SLICES")) K = atoi(e);
        K = std::max(1, std::min(K, std::min(SPX_MAX_SLICES, (int)std::max<size_t>(1, ng / 256))));
    }
    std::vector<spxl::PlanBase> bnd((size_t)K + 1);
    bnd[0] = spxl::PlanBase{0, 0, 0, 0, 0};
    bnd[(size_t)K] = spxl::PlanBase{(int64_t)np, (int64_t)nr, (int64_t)nq, T.s_tot, T.f_tot};
    if (K > 1) {
        HIPCHK(spx_prep_slice_bounds(A.slot0, d_base, (int32_t)ng, K, &bnd[(size_t)K], PL.d_bounds, PL.stream));
        HIPCHK(hipMemcpyAsync(PL.h_bounds, PL.d_bounds, sizeof(spxl::PlanBase) * ((size_t)K + 1), hipMemcpyDeviceToHost, PL.stream));
        HIPCHK(hipStreamSynchronize(PL.stream));
        for (int k = 0; k <= K; ++k) bnd[(size_t)k] = PL.h_bounds[k];
        for (int k = 0; k < K; ++k)
            if (bnd[(size_t)k + 1].prob < bnd[(size_t)k].prob || bnd[(size_t)k + 1].s_off < bnd[(size_t)k].s_off || bnd[(size_t)k + 1].f_off < bnd[(size_t)k].f_off)
                return fail(SPX_EINVAL, "work-list prefix sums are not monotonic");
    }
    int64_t max_s = 0, max_f = 0;
    for (int k = 0; k < K; ++k) {
        max_s = std::max(max_s, bnd[(size_t)k + 1].s_off - bnd[(size_t)k].s_off);
        max_f = std::max(max_f, bnd[(size_t)k + 1].f_off - bnd[(size_t)k].f_off);
    }
    std::vector<spx_order_segs> sfk((size_t)K), sbk((size_t)K);
    size_t order_f_n = 0, order_b_n = 0;
    w->main_cls = -1;
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls) {
        w->cls_cells[cls] = T.cls_cells[cls];
        w->st.problems_per_class[cls] = T.cls_prob[cls];
        w->cls_used[cls] = T.cls_prob[cls] > 0;
        if (w->cls_used[cls] && (w->main_cls < 0 || w->cls_cells[cls] > w->cls_cells[w->main_cls])) w->main_cls = cls;
    }
    for (int k = 0; k < K; ++k)
        for (int cls = 0; cls < SPX_N_CLASSES; ++cls) {
            /* a class segment holds its problems plus, per band width that can occur in the class, less than one wave of padding
             * (a slice: at most all the problems of the slice, at most all the problems of the class) */
            const int ppw_f = 64 / spx::class_lanes(cls), ppw_b = 64 / spx::class_lanes_bwd(cls);
            const int wmax = spx::class_slots(cls);
            const size_t widths = (size_t)std::min(1024, wmax / 2 + 1);
            const size_t most = (size_t)std::min<int64_t>(T.cls_prob[cls], bnd[(size_t)k + 1].prob - bnd[(size_t)k].prob);
            spx_order_segs &sf = sfk[(size_t)k], &sb = sbk[(size_t)k];
            sf.off[cls] = (int64_t)order_f_n;
            sf.cap[cls] = most > 0 ? (int64_t)((most + widths * ppw_f + 63) & ~(size_t)63) : 0;
            order_f_n += (size_t)sf.cap[cls];
            sb.off[cls] = (int64_t)order_b_n;
            sb.cap[cls] = most > 0 ? (int64_t)((most + widths * ppw_b + 63) & ~(size_t)63) : 0;
            order_b_n += (size_t)sb.cap[cls];
        }
    Carver cv;
    const size_t o_ref_nib = cv.take<int64_t>(np), o_qry_nib = cv.take<int64_t>(np), o_L = cv.take<int32_t>(np), o_R = cv.take<int32_t>(np),
                 o_bw = cv.take<int32_t>(np), o_hmm = cv.take<double>(np * SPX_H_N), o_row_off = cv.take<int32_t>(np),
                 o_n_rows = cv.take<int32_t>(np), o_s_off = cv.take<int64_t>(np), o_fs_off = cv.take<int64_t>(np),
                 o_prob_slots = cv.take<int32_t>(np), o_hasn = cv.take<uint8_t>(np + 16), o_rows = cv.take<int32_t>(nr), o_expect = cv.take<int32_t>(nr),
                 o_rawq = cv.take<uint8_t>(nr + 16), o_row_prob = cv.take<int32_t>(nr), o_rr = cv.take<spxl::RowRec>(nr + 1), o_qe = cv.take<int32_t>(5 * nq + 4),
                 o_order_f = cv.take<int32_t>(order_f_n + 64), o_order_b = cv.take<int32_t>(order_b_n + 64),
                 o_tier = cv.take<int32_t>(np + 8), /* (behind the launch orders: ONE 0xff fill covers orders, tiers and tier counters) */
                 o_mk_first = cv.take<int32_t>(ng + 2), o_markers = cv.take<spx_dev_marker>(nm + 1), o_mkref = cv.take<int32_t>(nm + 1),
                 o_naln = cv.take<uint8_t>(ng + 16), o_sec = cv.take<uint16_t>(ng + 8), o_rfe = cv.take<int32_t>(ng * 10 + 10),
                 o_rfs = cv.take<int32_t>(ng * 10 + 10), o_atid = cv.take<int32_t>(ng * 10 + 10), o_info = cv.take<spx_group_info>(ng + 1);
    const size_t in_bytes = cv.off;
    const size_t o_sinv = cv.take<double>((size_t)max_s + 16), o_fsave = cv.take<double>((size_t)max_f + 16), o_bq = cv.take<uint8_t>(nr + 16),
                 o_posmin = cv.take<uint8_t>(nm + 16), o_score = cv.take<double>(ng * 10 + 10), o_prim = cv.take<uint8_t>(ng + 16),
                 o_max = cv.take<uint8_t>(ng + 16), o_pass = cv.take<uint8_t>(ng + 16), o_tie = cv.take<uint16_t>(ng + 8),
                 o_results = cv.take<spx_group_out>(ng + 1);
    (void)in_bytes;
    w->arena_bytes = cv.off + 256;
    gate.wait(); /* lists of one pipeline take their memory in submission order: the oldest can always finish */
    w->arena = arena_get(c, w->arena_bytes, &w->arena_cap);
    gate.pass();
    if (!w->arena) return fail(SPX_ENOMEM, "device memory for the work list");
    work_arena_taken(c);
    char *B0 = (char *)w->arena;
    /* temporaries of the launch-order sort (context pools) */
    const size_t sort_tmp = spx_order_temp_bytes((int32_t)np);
    if ((rc = ensure_pool(PL, PL.pool_keys, (np + 16) * (3 * 8 + 2 * 4))) || (rc = ensure_pool(PL, PL.pool_sort, sort_tmp + 256)))
        return fail(rc, "device memory for the launch-order sort");
    spx_emit_args E;
    memset(&E, 0, sizeof E);
    E.base = d_base;
    E.mk_base = d_mkb;
    E.out.ref_nib = (int64_t *)(B0 + o_ref_nib); E.out.qry_nib = (int64_t *)(B0 + o_qry_nib);
    E.out.L = (int32_t *)(B0 + o_L); E.out.R = (int32_t *)(B0 + o_R); E.out.bw = (int32_t *)(B0 + o_bw);
    E.out.row_off = (int32_t *)(B0 + o_row_off); E.out.n_rows = (int32_t *)(B0 + o_n_rows); E.out.prob_slots = (int32_t *)(B0 + o_prob_slots);
    E.out.has_n = (uint8_t *)(B0 + o_hasn); E.hmm = (double *)(B0 + o_hmm); E.n_prob = (int32_t)np; E.out.s_off = (int64_t *)(B0 + o_s_off); E.out.fsave_off = (int64_t *)(B0 + o_fs_off);
    E.out.rows = (int32_t *)(B0 + o_rows); E.out.row_expect = (int32_t *)(B0 + o_expect); E.out.row_prob = (int32_t *)(B0 + o_row_prob);
    E.out.row_rawq = (uint8_t *)(B0 + o_rawq);
    E.out.rr = (spxl::RowRec *)(B0 + o_rr);
    E.n_rows = (int64_t)nr;
    int32_t *qe = (int32_t *)(B0 + o_qe);
    E.out.qe_rec = qe; E.out.qe_pos = qe + nq; E.out.qe_len = qe + 2 * nq; E.out.qe_row0 = qe + 3 * nq; E.out.qe_batch = qe + 4 * nq;
    for (int k = 0; k < 5; ++k) w->d_qe[k] = qe + (size_t)k * nq;
    E.markers = (spx_dev_marker *)(B0 + o_markers);
    E.mk_ref_pos = (int32_t *)(B0 + o_mkref);
    E.mk_first = (int32_t *)(B0 + o_mk_first);
    E.n_aln = (uint8_t *)(B0 + o_naln);
    E.sec_mask = (uint16_t *)(B0 + o_sec);
    E.rfe = (int32_t *)(B0 + o_rfe); E.rfs = (int32_t *)(B0 + o_rfs); E.atid = (int32_t *)(B0 + o_atid);
    E.info = (spx_group_info *)(B0 + o_info);
    HIPCHK(spx_prep_emit(&A, &E, PL.stream)); /* (group_finish_kernel writes every entry of mk_first) */
    /* launch orders: ONE fill for both order arrays (consecutive in the arena), one for the bins of both passes, one pair of sorts for all
     * DP slices (round 4: two fills + a pair of sorts + a fill per slice) */
    HIPCHK(hipMemsetAsync(B0 + o_order_f, 0xff, (o_tier - o_order_f) + (np + 8) * 4, PL.stream));
    w->fast = w->tiers_look && np > 0;
    w->d_tier = (int32_t *)(B0 + o_tier);
    w->d_tier_counts = w->d_tier + np;
    w->n_launches_counted = 0;
    if (w->fast) {
        fast_constants((float)w->par.conf_d, (float)w->par.conf_e, w->par.set_q, &w->fk);
    }
    fast_classes(w);
    if (w->tiers_look && np > 0) c->tiers_hint.store(w->fast ? 1 : 0);
    if (np) {
        const size_t nbins = (size_t)K * SPX_N_CLASSES * 1024;
        if ((rc = ensure_pool(PL, PL.pool_bins, nbins * 6 * sizeof(int32_t) + 256))) return fail(rc, "device memory for the launch-order bins");
        int32_t *bins = (int32_t *)PL.pool_bins.p;
        HIPCHK(hipMemsetAsync(bins, 0xff, nbins * 2 * sizeof(int32_t), PL.stream)); /* bin_start of both passes */
        spx_order_args O;
        memset(&O, 0, sizeof O);
        O.n_prob = (int32_t)np;
        O.fwd_by_last_row = w->fast ? 1 : 0;
        O.bw = E.out.bw; O.L = E.out.L; O.n_rows = E.out.n_rows; O.row_off = E.out.row_off; O.rows = E.out.rows;
        char *kp = (char *)PL.pool_keys.p;
        O.key_f = (uint64_t *)kp; O.key_b = O.key_f + (np + 2); O.key_sorted = O.key_b + (np + 2);
        O.val = (int32_t *)(O.key_sorted + (np + 2)); O.val_sorted = O.val + (np + 2);
        O.bin_start = bins; O.bin_end = bins + 2 * nbins; O.pad_base = bins + 4 * nbins;
        O.temp = PL.pool_sort.p; O.temp_bytes = sort_tmp;
        O.order_f = (int32_t *)(B0 + o_order_f); O.order_b = (int32_t *)(B0 + o_order_b);
        O.n_slices = K;
        for (int k = 0; k <= K; ++k) O.slice_prob[k] = (int32_t)bnd[(size_t)k].prob;
        /* (the pinned copy of the segment tables is free again: this call has synchronised the lane's stream since the last list used it) */
        for (int k = 0; k < K; ++k) { PL.h_segs[k] = sfk[(size_t)k]; PL.h_segs[SPX_MAX_SLICES + k] = sbk[(size_t)k]; }
        HIPCHK(hipMemcpyAsync(PL.d_segs, PL.h_segs, sizeof(spx_order_segs) * 2 * SPX_MAX_SLICES, hipMemcpyHostToDevice, PL.stream));
        O.segs_f = PL.d_segs; O.segs_b = PL.d_segs + SPX_MAX_SLICES;
        HIPCHK(spx_prep_orders(&O, PL.stream));
    }
    if (!w->ev_ready) HIPCHK(hipEventCreateWithFlags(&w->ev_ready, hipEventDisableTiming | hipEventBlockingSync));
    HIPCHK(hipEventRecord(w->ev_ready, PL.stream));
    /* ---- kernel argument blocks ---- */
    w->d_bq = (uint8_t *)(B0 + o_bq);
    w->d_posmin = (uint8_t *)(B0 + o_posmin);
    w->d_state = nullptr;
    w->d_q = nullptr;
    w->d_score = (double *)(B0 + o_score);
    w->d_prim = (uint8_t *)(B0 + o_prim);
    w->d_max = (uint8_t *)(B0 + o_max);
    w->d_pass = (uint8_t *)(B0 + o_pass);
    w->d_tie = (uint16_t *)(B0 + o_tie);
    w->d_grp_index = (int32_t *)(base + L.o_gidx);
    w->d_results = (spx_group_out *)(B0 + o_results);
    w->d_info = E.info;
    w->d_rfe = E.rfe; w->d_rfs = E.rfs; w->d_atid = E.atid; w->d_mk_first = E.mk_first; w->d_mk_ref_pos = E.mk_ref_pos;
    w->d_row_expect = E.out.row_expect;
    w->n_rows_dev = (int64_t)nr; w->n_mk_dev = (int64_t)nm; w->n_prob_dev = (int64_t)np;
    w->mirrors_markers = w->mirrors_qe = false;
    for (spx_work::Slice &sl : w->slices) {
        for (hipEvent_t e : {sl.ev_start, sl.ev_f0, sl.ev_f1, sl.ev_b1}) if (e) (void)hipEventDestroy(e);
    }
    w->slices.clear();
    if (K > 1) w->slices.resize((size_t)K);
    for (int k = 0; k < K; ++k)
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls) {
        const spx_order_segs &sf = sfk[(size_t)k], &sb = sbk[(size_t)k];
        spx_dev_batch &B = K > 1 ? w->slices[(size_t)k].cls_batch[cls] : w->cls_batch[cls];
        memset(&B, 0, sizeof B);
        B.order = (const int32_t *)(B0 + o_order_f) + sf.off[cls];
        B.order_bwd = (const int32_t *)(B0 + o_order_b) + sb.off[cls];
        B.n_order = (int32_t)sf.cap[cls];
        B.n_order_bwd = (int32_t)sb.cap[cls];
        B.ref_nib = E.out.ref_nib; B.qry_nib = E.out.qry_nib; B.L = E.out.L; B.R = E.out.R; B.bw = E.out.bw; B.hmm = E.hmm;
        B.row_off = E.out.row_off; B.n_rows = E.out.n_rows; B.s_off = E.out.s_off;
        B.ref4 = c->d_ref4;
        B.qry4 = A.P.code4;
        B.rows = E.out.rows; B.row_expect = E.out.row_expect; B.row_rawq = E.out.row_rawq;
        /* scratch shared by the slices: a slice's offsets start at its prefix (s_off / fsave_off stay list-wide) */
        B.sinv = (double *)(B0 + o_sinv) - bnd[(size_t)k].s_off;
        B.s_raw = nullptr;
        B.fsave = (double *)(B0 + o_fsave) - bnd[(size_t)k].f_off;
        B.row_prob = E.out.row_prob; B.prob_slots = E.out.prob_slots;
        B.fsave_stride = 2 * spx::class_slots(cls);
        B.fsave_off = E.out.fsave_off;
        B.out_bq = w->d_bq; B.out_state = nullptr; B.out_q = nullptr;
        B.qthr = c->d_tables;
        B.tier = w->fast ? w->d_tier : nullptr;
        B.tier_counts = w->fast ? w->d_tier_counts : nullptr;
        B.tier_want = SPX_TIER_ALL;
        if (K > 1 && k == 0) w->cls_batch[cls] = B; /* (what spx_work_export / the mirrors read: the per-problem arrays) */
    }
    for (int k = 0; k < K && K > 1; ++k) {
        spx_work::Slice &sl = w->slices[(size_t)k];
        sl.r0 = bnd[(size_t)k].row;
        sl.r1 = bnd[(size_t)k + 1].row;
        for (hipEvent_t *e : {&sl.ev_start, &sl.ev_f0, &sl.ev_f1, &sl.ev_b1}) HIPCHK(hipEventCreate(e));
    }
    spx_dev_groups &G = w->dg;
    memset(&G, 0, sizeof G);
    G.n_groups = (int32_t)ng;
    G.mk_first = E.mk_first;
    G.markers = E.markers;
    G.n_aln = E.n_aln;
    G.sec_mask = E.sec_mask;
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
    w->st.n_dispatched = (int64_t)T.n_ok;
    w->st.n_problems = (int64_t)np;
    w->st.n_rows = (int64_t)nr;
    w->st.dp_cells = T.cells;
    w->st.n_markers = (int64_t)nm;
    w->st.h2d_seconds = 0;
    w->prepared = true;
    w->launched = false;
    w->launch_ids.clear();
    if (timing_on())
        fprintf(stderr, "[spx timing] device prepare: counts %.3f s, carve+emit+orders enqueued %.3f s; %zu problems, %zu rows, list %.2f GB\n",
                t1 - t0, now_s() - t1, np, nr, w->arena_bytes / 1e9);
    return SPX_OK;
}

/* gives the prepared list (work list, scratch, outputs) back and keeps the staged records: the next
 * spx_prepare_staged builds it anew.  Waits for the list's kernels. */
extern "C" int spx_work_release(spx_ctx *c, spx_work *w)
{
    if (!c || !w) return fail(SPX_EINVAL, "NULL argument");
    if (w->in_pipe.load()) return fail(SPX_EINVAL, "work list is in flight in a pipeline");
    if (!w->staged || !w->arena) return SPX_OK;
    HIPCHK(hipSetDevice(c->device));
    if (w->ev_ready) HIPCHK(hipEventSynchronize(w->ev_ready));
    if (w->ev_done) HIPCHK(hipEventSynchronize(w->ev_done));
    arena_put(c, w->arena, w->arena_cap);
    work_arena_back(c);
    w->arena = nullptr;
    w->prepared = false;
    w->launched = false;
    return SPX_OK;
}

extern "C" int spx_prepare_many(spx_ctx *c, const spx_batch *const *bts, int32_t n_batches, const spx_params *par,
                                int host_threads, spx_work **out)
{
    if (!out) return fail(SPX_EINVAL, "NULL argument");
    spx_work *w = nullptr;
    int rc = spx_stage(c, bts, n_batches, par, host_threads, &w);
    if (rc) return rc;
    rc = spx_prepare_staged(c, w);
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
    if (w->staged && !w->prepared) return fail(SPX_EINVAL, "work list has been staged but not prepared");
    std::lock_guard<std::mutex> lk(c->launch_mu);
    if (w->ev_ready) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_ready, 0)); /* the list is built on the preparation stream */
    if (w->launched && w->ev_done) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_done, 0)); /* a replay: behind the scoring kernels of the launch before (result stream) */
    const int64_t launch_id = c->n_launch.load();
    hipEvent_t *ev = c->evr[launch_id % spx_ctx::SPX_EV_RING];
    w->launch_ids.push_back(launch_id);
    if (w->launch_ids.size() > (size_t)spx_ctx::SPX_EV_RING) w->launch_ids.erase(w->launch_ids.begin());
    c->n_launch++;
    HIPCHK(hipEventRecord(ev[0], c->stream));
    /* the class holding most of the band cells runs on the main stream (its forward and backward kernels are
     * bracketed by events); the others run beside it on the side streams, heaviest first, each on the stream that
     * has the least work so far */
    const int mc = w->main_cls;
    static const bool serial = getenv("SPX_SERIAL") != nullptr; /* diagnostics: one class after the other */
    /* What follows the last backward kernel of a list with groups -- the MAP kernel (of its last slice), the marker filter / score / decision /
     * result kernels -- goes to the RESULT stream (round 5): on the main stream it stood between this list's DP kernels and the next list's
     * heaviest forward kernel.  MAP streams the z rows at ~2 TB/s with the ALUs idle (2.8 ms per HiFi slice, 3.8 ms per mixed list); in the
     * score kernel one lane adds up the 45 000 marker positions of a 100 kb read's group in order (4 ms per mixed list).  The lists have
     * scratch of their own, so the next list's DP kernels may run beside them; the SLICES of one list share theirs and stay in order. */
    static const bool score_main = getenv("SPX_SCORE_MAIN") != nullptr; /* (experiment switch: the round-4 placement) */
    hipStream_t tail = (w->have_groups && !score_main) ? c->result_stream : c->stream;
    bool tail_open = false;
    auto open_tail = [&]() -> int { /* ev[0] .. ev[6]: the DP kernels on the main and side streams; ev[7] .. ev[1]: the last MAP kernel where it runs */
        if (tail_open) return SPX_OK;
        tail_open = true;
        HIPCHK(hipEventRecord(ev[6], c->stream));
        if (tail != c->stream) HIPCHK(hipStreamWaitEvent(tail, ev[6], 0));
        HIPCHK(hipEventRecord(ev[7], tail));
        return SPX_OK;
    };
    /* Two-tier DP: a class with a fast tier runs the fast kernels (which sort their problems into tiers), any other class the exact ones.
     * What follows the last backward kernel of a slice -- finish_rows: (1) fast MAP + certificate over the rows of fast-tier problems,
     * (2) exact MAP over the rows of the other classes, (3) the exact forward / backward kernels once more, restricted to the problems
     * that were not certified (SPX_TIER_RERUN; a launch of the class' whole order in which every other wave leaves at once), (4) exact
     * MAP over their rows. */
    const bool fast = w->fast;
    auto launch_cls = [&](int cls, int phase, const spx_dev_batch *Bc, hipStream_t st) -> hipError_t {
        if (w->cls_fast[cls]) return spx_launch_fast(cls, phase, Bc, &w->fk, st);
        return spx_launch_baq(cls, phase, Bc, st);
    };
    int fast_slots = 0; /* the widest fast class of the list: which instantiation of the fast MAP kernel */
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls)
        if (w->cls_fast[cls]) fast_slots = std::max(fast_slots, (int)spx::class_slots(cls));
    auto finish_rows = [&](const spx_dev_batch *cb /* [SPX_N_CLASSES] of the slice */, int64_t r0, int64_t r1, bool wide_first, hipStream_t st) -> int {
        const int32_t nrows = (int32_t)(r1 - r0);
        if (nrows <= 0) return SPX_OK;
        spx_dev_batch Bm = cb[0];
        Bm.row_base = (int32_t)r0;
        if (!fast) { HIPCHK(spx_launch_map(&Bm, nrows, wide_first, st)); return SPX_OK; }
        if (w->any_fast_cls) HIPCHK(spx_launch_fast_map(&Bm, &w->fk, nrows, fast_slots, st));
        if (w->any_exact_cls) { Bm.tier_want = SPX_TIER_EXACT; HIPCHK(spx_launch_map(&Bm, nrows, wide_first, st)); }
        return SPX_OK;
    };
    /* The re-run of the problems the fast tier did not certify waits until the LAST slice's rows are done and then runs on the tail stream,
     * slice by slice (they share one scratch area), beside the next list's DP kernels: a launch with a single problem still lasts a wave's
     * lifetime (2-3 ms for a full-length window), so re-running per slice and class on the main stream cost 10-16 ms per slice. */
    size_t rr_next = 0; /* next unused event of w->rr_ev */
    auto rr_event = [&](hipEvent_t *out) -> int {
        if (rr_next == w->rr_ev.size()) {
            hipEvent_t e = nullptr;
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            w->rr_ev.push_back(e);
        }
        *out = w->rr_ev[rr_next++];
        return SPX_OK;
    };
    auto rerun_slice = [&](const spx_dev_batch *cb, int64_t r0, int64_t r1, bool wide_first, hipStream_t st) -> int {
        if (!fast || !w->any_fast_cls) return SPX_OK;
        /* the classes' re-runs side by side on the side streams (each launch lasts a wave's lifetime even for one problem): fan out behind
         * an event on st, join before the MAP kernel.  The slices of a list share their scratch, so slice k+1's re-run starts behind slice k's MAP. */
        hipEvent_t ev_go = nullptr;
        { const int rc = rr_event(&ev_go); if (rc) return rc; }
        HIPCHK(hipEventRecord(ev_go, st));
        int si = 0;
        for (int cls = 0; cls < SPX_N_CLASSES; ++cls) {
            if (!w->cls_fast[cls] || cb[cls].n_order <= 0) continue;
            spx_dev_batch Br = cb[cls];
            Br.tier_want = SPX_TIER_RERUN;
            hipStream_t ss = serial ? st : (c->n_rerun > 0 ? c->rerun_stream[si % c->n_rerun] : c->side_stream[si % c->n_side]);
            if (ss != st) HIPCHK(hipStreamWaitEvent(ss, ev_go, 0));
            HIPCHK(spx_launch_baq(cls, 2, &Br, ss));
            if (ss != st) {
                hipEvent_t ev_d = nullptr;
                { const int rc = rr_event(&ev_d); if (rc) return rc; }
                HIPCHK(hipEventRecord(ev_d, ss));
                HIPCHK(hipStreamWaitEvent(st, ev_d, 0));
            }
            ++si;
        }
        if (r1 > r0) {
            spx_dev_batch Bm = cb[0];
            Bm.row_base = (int32_t)r0;
            Bm.tier_want = SPX_TIER_RERUN;
            HIPCHK(spx_launch_map(&Bm, (int32_t)(r1 - r0), wide_first, st));
        }
        return SPX_OK;
    };
    /* With the tiers on, the last slice's MAP kernels stay on the main stream (SPX_LAST_MAP_MAIN=0: on the tail stream like the exact MAP
     * kernel): beside the next list's DP kernels the fast MAP kernel of 131 072 HiFi groups took 33 ms instead of 4, and the re-runs and
     * the scoring kernels wait for it. */
    static const bool last_map_main = [] { const char *e = getenv("SPX_LAST_MAP_MAIN"); return !e || atoi(e) != 0; }(); /* default on (round 6: +2 %) */
    auto last_map = [&](const spx_dev_batch *cb, int64_t r0, int64_t r1, bool wide_first) -> int {
        if (fast && last_map_main) { const int rc = finish_rows(cb, r0, r1, wide_first, c->stream); if (rc) return rc; return open_tail(); }
        { const int rc = open_tail(); if (rc) return rc; }
        return finish_rows(cb, r0, r1, wide_first, tail);
    };
    int order[SPX_N_CLASSES], no = 0;
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls)
        if (w->cls_used[cls] && cls != mc) order[no++] = cls;
    std::sort(order, order + no, [&](int a, int b) { return w->cls_cells[a] != w->cls_cells[b] ? w->cls_cells[a] > w->cls_cells[b] : a < b; });
    if (w->slices.empty()) {
    int64_t load[spx_ctx::SPX_N_SIDE] = {};
    bool used_side[spx_ctx::SPX_N_SIDE] = {};
    for (int k = 0; k < no; ++k) {
        const int cls = order[k];
        if (serial) { HIPCHK(launch_cls(cls, 2, &w->cls_batch[cls], c->stream)); continue; }
        int sidx = 0;
        for (int t = 1; t < c->n_side; ++t) if (load[t] < load[sidx]) sidx = t;
        if (!used_side[sidx]) { HIPCHK(hipStreamWaitEvent(c->side_stream[sidx], ev[0], 0)); used_side[sidx] = true; }
        HIPCHK(launch_cls(cls, 2, &w->cls_batch[cls], c->side_stream[sidx]));
        load[sidx] += w->cls_cells[cls] + 1;
    }
    for (int t = 0; t < c->n_side; ++t)
        if (used_side[t]) HIPCHK(hipEventRecord(c->side_done[t], c->side_stream[t]));
    if (mc >= 0) {
        HIPCHK(hipEventRecord(ev[3], c->stream));
        HIPCHK(launch_cls(mc, 0, &w->cls_batch[mc], c->stream));
        HIPCHK(hipEventRecord(ev[4], c->stream));
        HIPCHK(launch_cls(mc, 1, &w->cls_batch[mc], c->stream));
        HIPCHK(hipEventRecord(ev[5], c->stream));
    }
    for (int t = 0; t < c->n_side; ++t)
        if (used_side[t]) HIPCHK(hipStreamWaitEvent(c->stream, c->side_done[t], 0));
    {
        int64_t narrow = 0, wide = 0;
        for (int cls = 0; cls < SPX_N_CLASSES; ++cls) (spx::class_slots(cls) <= 48 ? narrow : wide) += w->st.problems_per_class[cls];
        const int64_t nrows = w->staged ? w->n_rows_dev : (int64_t)w->hb.rows.size();
        { const int rc = last_map(w->cls_batch, 0, nrows, wide > narrow); if (rc) return rc; }
        { const int rc = rerun_slice(w->cls_batch, 0, nrows, wide > narrow, tail); if (rc) return rc; }
    }
    } else {
        /* DP slices: forward -> backward -> MAP of one slice after the other over the shared scratch; inside a slice the band
         * classes run side by side as above.  A slice starts when the MAP kernel of the one before has read the scratch. */
        int64_t narrow = 0, wide = 0;
        for (int cls = 0; cls < SPX_N_CLASSES; ++cls) (spx::class_slots(cls) <= 48 ? narrow : wide) += w->st.problems_per_class[cls];
        for (size_t k = 0; k < w->slices.size(); ++k) {
            spx_work::Slice &sl = w->slices[k];
            HIPCHK(hipEventRecord(sl.ev_start, c->stream));
            int64_t load[spx_ctx::SPX_N_SIDE] = {};
            bool used_side[spx_ctx::SPX_N_SIDE] = {};
            for (int q = 0; q < no; ++q) {
                const int cls = order[q];
                if (sl.cls_batch[cls].n_order <= 0) continue;
                if (serial) { HIPCHK(launch_cls(cls, 2, &sl.cls_batch[cls], c->stream)); continue; }
                int sidx = 0;
                for (int t = 1; t < c->n_side; ++t) if (load[t] < load[sidx]) sidx = t;
                if (!used_side[sidx]) { HIPCHK(hipStreamWaitEvent(c->side_stream[sidx], sl.ev_start, 0)); used_side[sidx] = true; }
                HIPCHK(launch_cls(cls, 2, &sl.cls_batch[cls], c->side_stream[sidx]));
                load[sidx] += w->cls_cells[cls] + 1;
            }
            for (int t = 0; t < c->n_side; ++t)
                if (used_side[t]) HIPCHK(hipEventRecord(c->side_done[t], c->side_stream[t]));
            if (mc >= 0) {
                HIPCHK(hipEventRecord(sl.ev_f0, c->stream));
                HIPCHK(launch_cls(mc, 0, &sl.cls_batch[mc], c->stream));
                HIPCHK(hipEventRecord(sl.ev_f1, c->stream));
                HIPCHK(launch_cls(mc, 1, &sl.cls_batch[mc], c->stream));
                HIPCHK(hipEventRecord(sl.ev_b1, c->stream));
            }
            for (int t = 0; t < c->n_side; ++t)
                if (used_side[t]) HIPCHK(hipStreamWaitEvent(c->stream, c->side_done[t], 0));
            if (sl.r1 > sl.r0 || k + 1 == w->slices.size()) {
                if (k + 1 == w->slices.size()) { const int rc = last_map(sl.cls_batch, sl.r0, sl.r1, wide > narrow); if (rc) return rc; }
                else { const int rc = finish_rows(sl.cls_batch, sl.r0, sl.r1, wide > narrow, c->stream); if (rc) return rc; }
            }
        }
        for (size_t k = 0; k < w->slices.size(); ++k) {
            spx_work::Slice &sl = w->slices[k];
            const int rc = rerun_slice(sl.cls_batch, sl.r0, sl.r1, wide > narrow, tail);
            if (rc) return rc;
        }
        if (mc >= 0) { /* (the ring's slots stay defined for readers that expect them) */
            HIPCHK(hipEventRecord(ev[3], c->stream));
            HIPCHK(hipEventRecord(ev[4], c->stream));
            HIPCHK(hipEventRecord(ev[5], c->stream));
        }
    }
    { const int rc = open_tail(); if (rc) return rc; } /* (a last slice without rows launched no MAP kernel: the tail still starts behind the DP kernels) */
    HIPCHK(hipEventRecord(ev[1], tail));
    if (w->have_groups) {
        const int64_t nmk = w->staged ? w->n_mk_dev : (int64_t)w->hb.markers.size();
        HIPCHK(spx_launch_score(&w->dg, (int32_t)nmk, w->d_posmin, tail));
        if (w->staged) HIPCHK(spx_launch_results(&w->dg, w->d_info, w->d_rfe, w->d_results, tail));
    }
    HIPCHK(hipEventRecord(ev[2], tail));
    if (!w->ev_done) HIPCHK(hipEventCreateWithFlags(&w->ev_done, hipEventDisableTiming | hipEventBlockingSync));
    HIPCHK(hipEventRecord(w->ev_done, tail));
    w->launched = true;
    w->n_launches_counted++;
    return SPX_OK;
}

extern "C" int spx_pack_decisions(spx_ctx *c, spx_work *w, int32_t group_base, void *device_out, int64_t capacity)
{
    if (!c || !w || !device_out) return fail(SPX_EINVAL, "NULL argument");
    const int64_t ng = w->staged ? (int64_t)w->n_dgroups : (int64_t)w->hb.grp_index.size();
    if (capacity < ng) return fail(SPX_EINVAL, "decision buffer too small");
    HIPCHK(hipSetDevice(c->device));
    /* behind the work list's own kernels only (not behind lists launched later), on the result stream; returns when the
     * records are in the buffer, so the caller can hand it to a collective on any stream */
    if (w->ev_done) HIPCHK(hipStreamWaitEvent(c->result_stream, w->ev_done, 0));
    else HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(spx_launch_pack(&w->dg, w->d_grp_index, group_base, (spx_decision *)device_out, c->result_stream));
    HIPCHK(hipStreamSynchronize(c->result_stream));
    return (int)ng;
}

/* Gives the device and pinned memory the context keeps for re-use (arenas of freed work lists, preparation pools)
 * back to the driver.  The next work list allocates afresh. */
extern "C" int spx_trim(spx_ctx *c)
{
    if (!c) return fail(SPX_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    for (int l = 0; l < c->n_prep; ++l) {
        spx_ctx::PrepLane &PL = c->lane[l];
        std::lock_guard<std::mutex> pl(PL.mu);
        HIPCHK(hipStreamSynchronize(PL.stream));
        for (spx_ctx::DevBuf *b : {&PL.pool_ops, &PL.pool_conf, &PL.pool_mm, &PL.pool_garena, &PL.pool_keys, &PL.pool_sort, &PL.pool_heavy, &PL.pool_bins})
            if (b->p) { (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->result_stream));
    {
        std::lock_guard<std::mutex> lk(c->arena_mu);
        for (auto &a : c->arena_cache) (void)hipFree(a.first);
        c->arena_cache.clear();
        for (auto &a : c->pinned_cache) (void)hipHostFree(a.first);
        c->pinned_cache.clear();
    }
    c->arena_cv.notify_all();
    return SPX_OK;
}

extern "C" int spx_sync(spx_ctx *c)
{
    if (!c) return fail(SPX_EINVAL, "NULL argument");
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->result_stream)); /* (the scoring kernels of the launched lists) */
    return SPX_OK;
}

extern "C" int spx_collect(spx_ctx *c, spx_work *w, spx_group_out *out)
{
    if (!c || !w || !out) return fail(SPX_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    /* wait for THIS work list only: lists launched after it keep running (pipelined callers) */
    if (w->ev_done) HIPCHK(hipEventSynchronize(w->ev_done));
    else HIPCHK(hipStreamSynchronize(c->stream));
    if (w->launched) {
        /* averages over THIS work list's launches since its previous collect (those whose events are still in the ring);
         * other work lists launched in between have their own slots.  Timing is diagnostics: a failing event query
         * zeroes the figures instead of failing the collect */
        double baq = 0, sc = 0, fw = 0, bw = 0, crit = 0, tailsp = 0;
        int n = 0;
        bool ok = true;
        for (int64_t l : w->launch_ids) {
            if (c->n_launch.load() - l > spx_ctx::SPX_EV_RING) continue; /* slot reused since */
            hipEvent_t *ev = c->evr[l % spx_ctx::SPX_EV_RING];
            float ms = 0;
            ok = ok && hipEventElapsedTime(&ms, ev[0], ev[6]) == hipSuccess; /* the DP kernels ... */
            baq += ms;
            crit += ms;
            ok = ok && hipEventElapsedTime(&ms, ev[7], ev[2]) == hipSuccess;
            tailsp += ms;
            ok = ok && hipEventElapsedTime(&ms, ev[7], ev[1]) == hipSuccess; /* ... + the last MAP kernel where it ran (not the time it waited for the stream) */
            baq += ms;
            ok = ok && hipEventElapsedTime(&ms, ev[1], ev[2]) == hipSuccess;
            sc += ms;
            if (w->main_cls >= 0 && w->slices.empty()) {
                ok = ok && hipEventElapsedTime(&ms, ev[3], ev[4]) == hipSuccess;
                fw += ms;
                ok = ok && hipEventElapsedTime(&ms, ev[4], ev[5]) == hipSuccess;
                bw += ms;
            }
            ++n;
        }
        w->launch_ids.clear();
        if (w->main_cls >= 0 && !w->slices.empty() && n > 0) {
            /* sliced list: the main class' forward / backward time = the sum over the slices of its LATEST launch, counted once
             * per averaged launch */
            double f1 = 0, b1 = 0;
            for (spx_work::Slice &sl : w->slices) {
                float ms = 0;
                ok = ok && hipEventElapsedTime(&ms, sl.ev_f0, sl.ev_f1) == hipSuccess;
                f1 += ms;
                ok = ok && hipEventElapsedTime(&ms, sl.ev_f1, sl.ev_b1) == hipSuccess;
                b1 += ms;
            }
            fw = f1 * n;
            bw = b1 * n;
        }
        if (!ok) { (void)hipGetLastError(); n = 0; }
        const double dn = n > 0 ? (double)n : 1.0;
        if (n == 0) baq = sc = fw = bw = crit = tailsp = 0;
        w->st.dp_critical_ms = crit / dn;
        w->st.tail_span_ms = tailsp / dn;
        w->st.baq_kernel_ms = baq / dn;
        w->st.score_kernel_ms = sc / dn;
        w->st.n_launches_averaged = n;
        w->st.dp_slices = w->slices.empty() ? 1 : (int32_t)w->slices.size();
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
    w->st.tier_fast_problems = w->st.tier_rerun_certificate = w->st.tier_rerun_model = w->st.tier_rerun_range = w->st.tier_rows_uncertified = 0;
    if (w->fast && w->launched && w->d_tier_counts) {
        int32_t cnt[4] = {-1, -1, -1, -1}; /* (the counters start at -1: the launch orders' 0xff fill) */
        if (hipMemcpy(cnt, w->d_tier_counts, sizeof cnt, hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError();
        const int64_t nl = std::max<int64_t>(1, w->n_launches_counted);
        for (int cls = 0; cls < SPX_N_CLASSES; ++cls)
            if (w->cls_fast[cls]) w->st.tier_fast_problems += w->st.problems_per_class[cls];
        w->st.tier_rerun_certificate = ((int64_t)cnt[0] + 1) / nl;
        w->st.tier_rerun_model = ((int64_t)cnt[1] + 1) / nl;
        w->st.tier_rerun_range = ((int64_t)cnt[2] + 1) / nl;
        w->st.tier_rows_uncertified = ((int64_t)cnt[3] + 1) / nl;
        const int64_t v[5] = {w->st.tier_fast_problems, w->st.tier_rerun_certificate, w->st.tier_rerun_model, w->st.tier_rerun_range, w->st.tier_rows_uncertified};
        for (int k = 0; k < 5; ++k) g_last_tier[k].store(v[k]);
    } else if (w->launched)
        for (int k = 0; k < 5; ++k) g_last_tier[k].store(0);
    if (w->staged) {
        /* device-prepared list: one packed record per dispatched group (results_kernel), one copy */
        const size_t ng = (size_t)w->n_dgroups;
        double t0 = now_s();
        size_t cap = 0;
        spx_group_out *h = ng ? (spx_group_out *)pinned_get(c, ng * sizeof(spx_group_out), &cap) : nullptr;
        if (ng && !h) return fail(SPX_ENOMEM, "pinned result buffer");
        if (ng) {
            /* the list's own kernels are done (ev_done above).  A SYNCHRONOUS copy: on this runtime an asynchronous
             * device-to-host copy of these 19 MB is served by the copy kernel (whatever the stream), which then queues for
             * CUs behind the DP kernels of the next lists -- 90 ms in the kernel trace for a 0.4 ms transfer */
            hipError_t e = hipMemcpy(h, w->d_results, ng * sizeof(spx_group_out), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { pinned_put(c, h, cap); return fail(SPX_EHIP, std::string("result copy: ") + hipGetErrorString(e)); }
        }
        w->st.d2h_seconds = now_s() - t0;
        w->st.bytes_d2h = (int64_t)(ng * sizeof(spx_group_out));
        for (int32_t g = 0; g < w->n_groups_in; ++g) {
            memset(&out[g], 0, sizeof out[g]);
            out[g].prim_idx = out[g].max_idx = out[g].best_idx = -1;
        }
        for (size_t k = 0; k < ng; ++k) {
            const int32_t g = w->stage.grp_index[k];
            out[g] = h[k];
            if (h[k].n_aln < 0) w->hb.grp_error[(size_t)g] = h[k].n_aln;
        }
        pinned_put(c, h, cap);
        return SPX_OK;
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

/* host mirrors of a device-prepared list, pulled on demand: the per-group / per-marker arrays the BED bookkeeping
 * reads, and the quality edits of an all-rows list */
static int pull_marker_mirrors(spx_ctx *c, spx_work *w)
{
    if (!w->staged || w->mirrors_markers) return SPX_OK;
    HIPCHK(hipSetDevice(c->device));
    if (w->ev_done) HIPCHK(hipEventSynchronize(w->ev_done)); /* this list's kernels, not whatever follows on the stream */
    else HIPCHK(hipStreamSynchronize(c->stream));
    spx::HostBatch &hb = w->hb;
    const size_t ng = (size_t)w->n_dgroups, nm = (size_t)w->n_mk_dev;
    hb.grp_index = w->stage.grp_index;
    hb.rfe.assign(ng * 10, 0); hb.rfs.assign(ng * 10, 0); hb.atid.assign(ng * 10, -1);
    hb.mk_first.assign(ng + 1, 0); hb.mk_ref_pos.assign(nm, 0); hb.n_aln.assign(ng, 0);
    w->posmin_host.assign(nm, 0);
    std::vector<spx_group_info> info(ng);
    if (ng) {
        HIPCHK(hipMemcpy(hb.rfe.data(), w->d_rfe, ng * 40, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hb.rfs.data(), w->d_rfs, ng * 40, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hb.atid.data(), w->d_atid, ng * 40, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hb.mk_first.data(), w->d_mk_first, (ng + 1) * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(info.data(), w->d_info, ng * sizeof(spx_group_info), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < ng; ++k) hb.n_aln[k] = (uint8_t)(info[k].err ? 0 : info[k].n_aln);
    }
    if (nm) {
        HIPCHK(hipMemcpy(hb.mk_ref_pos.data(), w->d_mk_ref_pos, nm * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(w->posmin_host.data(), w->d_posmin, nm, hipMemcpyDeviceToHost));
    }
    w->mirrors_markers = true;
    return SPX_OK;
}

static int pull_qe_mirrors(spx_ctx *c, spx_work *w)
{
    if (!w->staged || w->mirrors_qe) return SPX_OK;
    if (w->ev_done) HIPCHK(hipEventSynchronize(w->ev_done));
    HIPCHK(hipStreamSynchronize(c->stream));
    spx::HostBatch &hb = w->hb;
    const size_t nq = (size_t)w->tot.n_qe, nr = (size_t)w->n_rows_dev;
    hb.qe_rec.assign(nq, 0); hb.qe_pos.assign(nq, 0); hb.qe_len.assign(nq, 0); hb.qe_row0.assign(nq, 0); hb.qe_batch.assign(nq, 0);
    hb.row_expect.assign(nr, 0);
    hb.rows.assign(nr, 0); /* only its size is consulted */
    if (nq) {
        HIPCHK(hipMemcpy(hb.qe_rec.data(), w->d_qe[0], nq * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hb.qe_pos.data(), w->d_qe[1], nq * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hb.qe_len.data(), w->d_qe[2], nq * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hb.qe_row0.data(), w->d_qe[3], nq * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hb.qe_batch.data(), w->d_qe[4], nq * 4, hipMemcpyDeviceToHost));
    }
    if (nr) HIPCHK(hipMemcpy(hb.row_expect.data(), w->d_row_expect, nr * 4, hipMemcpyDeviceToHost));
    w->mirrors_qe = true;
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
    if (w->ev_done) HIPCHK(hipEventSynchronize(w->ev_done)); /* (the last MAP kernel runs on the result stream) */
    HIPCHK(hipStreamSynchronize(c->stream));
    { int rc = pull_qe_mirrors(c, w); if (rc) return rc; }
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

extern "C" int64_t spx_work_device_bytes(const spx_work *w, int32_t *n_slices)
{
    if (!w) return 0;
    if (n_slices) *n_slices = w->slices.empty() ? 1 : (int32_t)w->slices.size();
    return (int64_t)w->arena_bytes;
}

extern "C" void spx_work_free(spx_ctx *c, spx_work *w)
{
    if (!w) return;
    if (c) (void)hipSetDevice(c->device);
    if (c && (w->arena || w->in_arena)) { /* nothing of this work list may still be running */
        if (w->ev_staged) (void)hipEventSynchronize(w->ev_staged);
        if (w->ev_ready) (void)hipEventSynchronize(w->ev_ready);
        else if (w->staged) (void)hipStreamSynchronize(c->lane[w->lane].stream);
        if (w->ev_done) (void)hipEventSynchronize(w->ev_done);
        else (void)hipStreamSynchronize(c->stream);
    }
    if (w->arena) { arena_put(c, w->arena, w->arena_cap); if (w->staged) work_arena_back(c); }
    if (w->in_arena) arena_put(c, w->in_arena, w->in_cap);
    if (w->h_stage) pinned_put(c, w->h_stage, w->h_stage_cap);
    if (w->ev_ready) (void)hipEventDestroy(w->ev_ready);
    if (w->ev_staged) (void)hipEventDestroy(w->ev_staged);
    if (w->ev_done) (void)hipEventDestroy(w->ev_done);
    for (spx_work::Slice &sl : w->slices)
        for (hipEvent_t e : {sl.ev_start, sl.ev_f0, sl.ev_f1, sl.ev_b1}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : w->rr_ev) if (e) (void)hipEventDestroy(e);
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

/* one value of the stream (spx_gather.cpp holds the decision rule that consumes them) */
extern "C" int spx_finalizer_draw(spx_finalizer *f, int32_t *out)
{
    if (!f || !out) return SPX_EINVAL;
    random_r(&f->rd, out);
    return SPX_OK;
}

/* advances the stream by n values without using them (a rank of a multi-GPU run passes over the draws of the groups
 * that other ranks decide; ~1 ns per value) */
extern "C" int spx_finalizer_skip(spx_finalizer *f, int64_t n)
{
    if (!f || n < 0) return SPX_EINVAL;
    int32_t v;
    for (int64_t k = 0; k < n; ++k) random_r(&f->rd, &v);
    return SPX_OK;
}

/* BED bookkeeping of relabelled reads (src/secphase.c:201-212): extents of the old primary and of the promoted
 * secondary (count 1 each), and the reference positions of their surviving markers */
extern "C" int spx_bedset_add(spx_bedset *b, const char *contig, int32_t start, int32_t end, int32_t count);
extern "C" int spx_relabel_blocks(const spx_work *w, const spx_ref *ref, const spx_group_out *out,
                                  spx_bedset *modified_blocks, spx_bedset *marker_blocks)
{
    if (!w || !ref || !out) return fail(SPX_EINVAL, "NULL argument");
    const double t_rb0 = now_s();
    if (w->staged) {
        if (!w->owner) return fail(SPX_EINVAL, "work list has no context");
        int rc = pull_marker_mirrors(w->owner, const_cast<spx_work *>(w));
        if (rc) return rc;
    }
    const double t_rb1 = now_s();
    const spx::HostBatch &hb = w->hb;
    if (marker_blocks && !w->staged && w->posmin_host.size() != hb.markers.size()) return fail(SPX_EINVAL, "spx_collect has not run");
    int n = 0;
    std::vector<int32_t> pts;
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
            if (na <= 0) continue;
            pts.clear();
            for (int32_t m = hb.mk_first[k]; m < hb.mk_first[k + 1]; m += na) {
                if (w->posmin_host[m] <= w->par.min_q) continue; /* position removed by filter_lowq_markers */
                pts.push_back(hb.mk_ref_pos[m + a]);
            }
            spx_bedset_add_points(marker_blocks, contig, pts.data(), (int32_t)pts.size());
        }
    }
    if (timing_on()) fprintf(stderr, "[spx timing] relabel blocks: marker arrays back from the device %.3f s, sets %.3f s\n", t_rb1 - t_rb0, now_s() - t_rb1);
    return n;
}

/* the records print_alignment_scores (src/secphase.c:32-57) + the header lines (:194-200) would write for the relabelled
 * groups of a finalized batch, formatted in slices on several threads (tens of thousands of relabelled groups per batch:
 * one thread of fprintf was a third of the command line's loop) and returned in file order */
static void format_relabel_text(const spx_batch *bt, const spx_ref *ref, const spx_group_out *out, std::vector<std::string> &text)
{
    const int32_t n = bt->n_groups;
    int nthr = (int)std::thread::hardware_concurrency();
    nthr = std::max(1, std::min(nthr, 16));
    if (n < 4096) nthr = 1;
    text.assign((size_t)nthr, std::string());
    auto slice = [&](int t) {
        const int32_t g0 = (int32_t)((int64_t)n * t / nthr), g1 = (int32_t)((int64_t)n * (t + 1) / nthr);
        std::string &s = text[(size_t)t];
        char line[512];
        for (int32_t g = g0; g < g1; ++g) {
            const spx_group_out &o = out[g];
            if (!o.relabel) continue;
            s += "#MARKER SCORE\n$\t";
            s += bt->qnames + bt->qname_off[g];
            s += '\n';
            int i = 0;
            for (int a = bt->grp_first[g]; a < bt->grp_first[g + 1]; ++a) {
                if (bt->flag[a] & SPX_FUNMAP) continue;
                if (i >= o.n_aln) break;
                const char *tag = !(bt->flag[a] & SPX_FSECONDARY) ? "*" : (i == o.best_idx ? "@" : "!");
                const int m = snprintf(line, sizeof line, "%s\t%.2f\t%s\t%ld\t%d\n", tag, o.score[i], ref->names + ref->name_off[bt->tid[a]],
                                       (long)bt->pos[a], o.rfe[i]);
                if (m > 0) s.append(line, (size_t)std::min<int>(m, (int)sizeof line - 1));
                ++i;
            }
            s += '\n';
        }
    };
    if (nthr == 1) slice(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; ++t) th.emplace_back(slice, t);
        for (auto &x : th) x.join();
    }
}

extern "C" int spx_write_relabel_log(const char *path, const char *mode, const spx_batch *bt, const spx_ref *ref,
                                     const spx_group_out *out)
{
    if (!path || !bt || !ref || !out) return fail(SPX_EINVAL, "NULL argument");
    FILE *f = fopen(path, mode && *mode ? mode : "w");
    if (!f) return fail(SPX_EINVAL, std::string("cannot open ") + path);
    std::vector<std::string> text;
    format_relabel_text(bt, ref, out, text);
    bool ok = true;
    for (const std::string &t : text)
        if (!t.empty() && fwrite(t.data(), 1, t.size(), f) != t.size()) ok = false;
    if (fclose(f) != 0) ok = false;
    return ok ? SPX_OK : fail(SPX_EINVAL, std::string("write error on ") + path);
}

/* the same text into memory (malloc'ed, free it with spx_free_text): a rank of a multi-GPU run formats the fragment of
 * the list that its own groups contribute; rank 0 only appends the fragments in rank order */
extern "C" int spx_format_relabel_text(const spx_batch *bt, const spx_ref *ref, const spx_group_out *out, char **text_out, int64_t *len_out)
{
    if (!bt || !ref || !out || !text_out || !len_out) return fail(SPX_EINVAL, "NULL argument");
    std::vector<std::string> text;
    format_relabel_text(bt, ref, out, text);
    size_t n = 0;
    for (const std::string &t : text) n += t.size();
    char *buf = (char *)malloc(n + 1);
    if (!buf) return fail(SPX_ENOMEM, "relabel text");
    size_t at = 0;
    for (const std::string &t : text) { memcpy(buf + at, t.data(), t.size()); at += t.size(); }
    buf[n] = 0;
    *text_out = buf;
    *len_out = (int64_t)n;
    return SPX_OK;
}
extern "C" void spx_free_text(char *text) { free(text); }

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
    hb = spx::HostBatch();
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
            hb.hmm[hb.hmm.size() - SPX_H_N + SPX_H_TDROP] = spxl::terminal_drop(spx::terminal_guard(), L, R, bw) ? 1.0 : 0.0;
        }
        hb.dp_cells += spx::band_cells(L, R, bw);
    }
    ref4.resize(ref4.size() + spx::kRefTailBytes, 0);
    uint8_t *d_ref = nullptr, *saved = c->d_ref4;
    HIPCHK(hipMalloc((void **)&d_ref, ref4.size()));
    HIPCHK(hipMemcpy(d_ref, ref4.data(), ref4.size(), hipMemcpyHostToDevice));
    c->d_ref4 = d_ref;
    if (n > 0) { w->fast_d = pars[0].d; w->fast_e = pars[0].e; w->fast_set_q = set_q[0]; }
    int rc = build_device_batch(c, w, true, /*allow_fast=*/!post_scale && !pr_out);
    c->d_ref4 = saved;
    if (!rc) rc = spx_launch(c, w);
    if (!rc && (hipStreamSynchronize(c->stream) != hipSuccess || hipStreamSynchronize(c->result_stream) != hipSuccess)) rc = fail(SPX_EHIP, "kernel execution failed");
    if (!rc) {
        float ms = 0;
        hipEvent_t *ev = c->evr[(c->n_launch.load() - 1) % spx_ctx::SPX_EV_RING];
        (void)hipEventElapsedTime(&ms, ev[0], ev[1]); /* (no groups: everything on the main stream) */
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
        if (w->fast) {
            int32_t cnt[4] = {-1, -1, -1, -1};
            if (hipMemcpy(cnt, w->d_tier_counts, sizeof cnt, hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError();
            int64_t nf = 0;
            for (int32_t p = 0; p < n; ++p) nf += w->cls_fast[spx::band_class(2 * hb.bw[p] + 1)] ? 1 : 0;
            g_last_tier[0].store(nf);
            for (int k = 0; k < 4; ++k) g_last_tier[k + 1].store((int64_t)cnt[k] + 1);
        } else
            for (int k = 0; k < 5; ++k) g_last_tier[k].store(0);
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

extern "C" hipError_t spx_launch_probaln_general(const uint8_t *d_ref, int32_t l_ref, const uint8_t *d_query, int32_t l_query, const float *d_qual, int32_t bw,
                                                 const double *hmm13, double *d_f, double *d_b, double *d_s, int64_t i_dim, int32_t *d_state, uint8_t *d_q,
                                                 const double *d_thr, int32_t drop_last_column, hipStream_t st);

extern "C" int spx_set_terminal_guard(int reading)
{
    if (reading != SPX_GUARD_BAND && reading != SPX_GUARD_ROW) return fail(SPX_EINVAL, "terminal guard: SPX_GUARD_BAND or SPX_GUARD_ROW");
    spx::set_terminal_guard(reading);
    return SPX_OK;
}
extern "C" int spx_get_terminal_guard(void) { return spx::terminal_guard(); }

/* per-base qualities (htslib's iqual[i]; samtools' BAQ passes them, secphase never does): the general kernel of
 * spx_probaln_general.hip, one problem at a time.  Returns the phred-scaled likelihood or INT_MIN. */
static int probaln_per_base(spx_ctx *c, const uint8_t *ref, int l_ref, const uint8_t *query, int l_query, const uint8_t *iqual,
                            const spx_probaln_par *cpar, int *state, uint8_t *q)
{
    if (hipSetDevice(c->device) != hipSuccess) return INT_MIN;
    const int bw = spx::effective_bw(l_ref, l_query, cpar->bw);
    const int64_t i_dim = 3 * (2 * (int64_t)bw + 1) + 6, cells = ((int64_t)l_query + 1) * i_dim + 8;
    double h[SPX_H_N];
    spx::hmm_constants(l_ref, l_query, cpar->d, cpar->e, 30, h); /* (the emission entries are not used: they come per base) */
    const double hmm13[13] = {h[SPX_H_M0], h[SPX_H_M1], h[SPX_H_M2], h[SPX_H_M3], h[SPX_H_M4], 0., h[SPX_H_M6], 0., h[SPX_H_M8],
                              h[SPX_H_BM], h[SPX_H_BI], h[SPX_H_SM], h[SPX_H_SI]};
    std::vector<float> qual((size_t)l_query);
    for (int i = 0; i < l_query; ++i) qual[(size_t)i] = (float)pow(10, -iqual[i] / 10.); /* htslib's qual[] (host libm) */
    char *blk = nullptr;
    Carver cv;
    const size_t o_f = cv.take<double>((size_t)cells), o_b = cv.take<double>((size_t)cells), o_s = cv.take<double>((size_t)l_query + 2),
                 o_qual = cv.take<float>((size_t)l_query), o_ref = cv.take<uint8_t>((size_t)l_ref), o_qry = cv.take<uint8_t>((size_t)l_query),
                 o_state = cv.take<int32_t>((size_t)l_query), o_q = cv.take<uint8_t>((size_t)l_query);
    if (hipMalloc((void **)&blk, cv.off + 256) != hipSuccess) { (void)hipGetLastError(); fail(SPX_ENOMEM, "device memory for the DP matrices"); return INT_MIN; }
    int pr = INT_MIN;
    std::vector<double> s((size_t)l_query + 2);
    std::vector<int32_t> st32((size_t)l_query);
    hipError_t e = hipMemsetAsync(blk, 0, cv.off + 256, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(blk + o_qual, qual.data(), qual.size() * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(blk + o_ref, ref, (size_t)l_ref, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(blk + o_qry, query, (size_t)l_query, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = spx_launch_probaln_general((const uint8_t *)(blk + o_ref), l_ref, (const uint8_t *)(blk + o_qry), l_query, (const float *)(blk + o_qual), bw, hmm13,
                                       (double *)(blk + o_f), (double *)(blk + o_b), (double *)(blk + o_s), i_dim, (int32_t *)(blk + o_state),
                                       (uint8_t *)(blk + o_q), c->d_tables, spxl::terminal_drop(spx::terminal_guard(), l_query, l_ref, bw), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s.data(), blk + o_s, s.size() * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(st32.data(), blk + o_state, st32.size() * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(q, blk + o_q, (size_t)l_query, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(blk);
    if (e != hipSuccess) { fail(SPX_EHIP, hipGetErrorString(e)); return INT_MIN; }
    for (int i = 0; i < l_query; ++i) state[i] = st32[(size_t)i];
    double pp = 1., Pr1 = 0.;
    for (int i = 0; i <= l_query + 1; ++i) {
        pp *= s[(size_t)i];
        if (pp < 1e-100) { Pr1 += -4.343 * log(pp); pp = 1.; }
    }
    Pr1 += -4.343 * log(pp * l_ref * l_query);
    const double v = Pr1 + .499;
    pr = (v > -2147483649.0 && v < 2147483648.0) ? (int)v : INT_MIN;
    return pr;
}

extern "C" int spx_probaln_glocal(const uint8_t *ref, int l_ref, const uint8_t *query, int l_query, const uint8_t *iqual,
                                  const spx_probaln_par *cpar, int *state, uint8_t *q)
{
    if (l_ref <= 0 || l_query <= 0) return 0; /* as htslib */
    if (!ref || !query || !cpar || !state || !q) return INT_MIN;
    int sq = iqual ? iqual[0] : 30;
    bool per_base = false;
    if (iqual)
        for (int i = 1; i < l_query; ++i)
            if (iqual[i] != sq) { per_base = true; break; }
    std::lock_guard<std::mutex> lk(g_single_mu);
    if (!g_single && spx_create(0, &g_single) != SPX_OK) return INT_MIN;
    if (per_base) return probaln_per_base(g_single, ref, l_ref, query, l_query, iqual, cpar, state, q);
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

extern "C" int spx_effective_cpus(void) { return spx::effective_cpus(); }

extern "C" void spx_internal_cpu_add(int kind, double seconds)
{
    if (kind >= 0 && kind < spx::CPU_N && spx::cpu_acc_on()) spx::cpu_acc()[kind].fetch_add((int64_t)(seconds * 1e9), std::memory_order_relaxed);
}
/* diagnostics (SPX_TIMING): core-seconds of the host side by kind of work, all threads, since the process started */
extern "C" void spx_internal_cpu_report(FILE *f)
{
    static const char *const names[spx::CPU_N] = {"host inflate", "CRC-32", "device-inflate chunks (copies in / out, waiting)", "mapping the file",
                                                  "record chain (walker thread)", "fields / tags / CIGAR copies", "staging: sizes + repeated SEQ/QUAL",
                                                  "staging: copies into pinned memory", "BED bookkeeping"};
    struct timespec ts;
    clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
    fprintf(f, "[spx timing] CPU time of the process %.3f core-s, of which:", ts.tv_sec + 1e-9 * ts.tv_nsec);
    for (int k = 0; k < spx::CPU_N; ++k) fprintf(f, " %s %.3f;", names[k], 1e-9 * (double)spx::cpu_acc()[k].load());
    fprintf(f, "\n");
}

/* diagnostics (host only): what staging these batches would put on the wire.  out[0] alignments of dispatched groups,
 * out[1] of them aliased to their group's primary (SEQ / QUAL not transferred), out[2] SEQ + QUAL bytes of all of them,
 * out[3] SEQ + QUAL bytes really transferred */
extern "C" int spx_stage_transfer_stats(const spx_batch *const *bts, int32_t n_batches, int host_threads, int64_t *out)
{
    if (!bts || n_batches <= 0 || !out) return fail(SPX_EINVAL, "NULL argument");
    spx::Stage st;
    int rc = spx::stage_measure(bts, n_batches, host_threads > 0 ? host_threads : 1, st);
    if (rc) return fail(rc, "invalid batch");
    out[0] = st.lay.n_slots; out[1] = st.lay.n_aliased;
    out[2] = st.lay.seq_bytes + st.lay.qual_bytes; out[3] = st.lay.pk_seq_bytes + st.lay.pk_qual_bytes;
    return SPX_OK;
}

/* ------------------------------------------------------------------ */
/* BGZF blocks inflated on the device (spx_inflate_kernels.hip): `file` holds n_blocks consecutive BGZF blocks starting at
 * block_off[0] (block_off has n_blocks + 1 entries: the starts and the end).  The inflated bytes of the blocks are written
 * back to back into out (host); status[b] = 0, -1 corrupt DEFLATE data, -2 / -3 size mismatch, -4 CRC mismatch.
 * Returns the number of inflated bytes or SPX_E*; *kernel_ms (may be NULL) receives the kernel's duration. */
struct SpxBgzfDesc { int64_t in_off, out_off; uint32_t clen, ulen, crc, pad; };
extern "C" int64_t spx_inflate_bgzf_device(spx_ctx *c, const uint8_t *file, const int64_t *block_off, int32_t n_blocks, uint8_t *out,
                                           int64_t out_cap, int32_t *status, double *kernel_ms)
{
    if (!c || !file || !block_off || n_blocks < 0 || (!out && out_cap > 0) || !status) return fail(SPX_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    std::vector<SpxBgzfDesc> desc((size_t)n_blocks);
    const int64_t base = n_blocks ? block_off[0] : 0;
    int64_t utot = 0;
    for (int32_t b = 0; b < n_blocks; ++b) {
        const uint8_t *p = file + block_off[b];
        const int64_t total = block_off[b + 1] - block_off[b];
        if (total < 26 || p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return fail(SPX_EINVAL, "not a BGZF block");
        const int64_t xlen = p[10] | (p[11] << 8);
        if (12 + xlen + 8 > total) return fail(SPX_EINVAL, "corrupt BGZF block");
        const uint8_t *t = p + total - 8;
        SpxBgzfDesc &d = desc[(size_t)b];
        d.in_off = block_off[b] - base + 12 + xlen;
        d.clen = (uint32_t)(total - 12 - xlen - 8);
        d.crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        d.ulen = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        d.out_off = utot;
        d.pad = 0;
        if (d.ulen > 65536) return fail(SPX_EINVAL, "corrupt BGZF block (ISIZE)");
        utot += d.ulen;
    }
    if (utot > out_cap) return fail(SPX_EINVAL, "output buffer too small");
    if (n_blocks == 0) return 0;
    const size_t cbytes = (size_t)(block_off[n_blocks] - base);
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    SpxBgzfDesc *d_desc = nullptr;
    int32_t *d_status = nullptr;
    void *d_scratch = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SPX_OK;
    auto cleanup = [&]() {
        if (d_comp) (void)hipFree(d_comp);
        if (d_out) (void)hipFree(d_out);
        if (d_desc) (void)hipFree(d_desc);
        if (d_status) (void)hipFree(d_status);
        if (d_scratch) (void)hipFree(d_scratch);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define ZCHK(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); cleanup(); return fail(SPX_EHIP, #x); } } while (0)
    ZCHK(hipMalloc((void **)&d_comp, cbytes + 64));
    ZCHK(hipMalloc((void **)&d_out, (size_t)utot + 64));
    ZCHK(hipMalloc((void **)&d_desc, desc.size() * sizeof(SpxBgzfDesc)));
    ZCHK(hipMalloc((void **)&d_status, (size_t)n_blocks * 4));
    ZCHK(hipMalloc(&d_scratch, spx_bgzf_inflate_scratch_bytes(n_blocks) + 64));
    ZCHK(hipMemset(d_comp + cbytes, 0, 64));
    ZCHK(hipMemcpy(d_comp, file + base, cbytes, hipMemcpyHostToDevice));
    ZCHK(hipMemcpy(d_desc, desc.data(), desc.size() * sizeof(SpxBgzfDesc), hipMemcpyHostToDevice));
    ZCHK(hipEventCreate(&e0));
    ZCHK(hipEventCreate(&e1));
    ZCHK(hipEventRecord(e0, c->stream));
    ZCHK(spx_launch_bgzf_inflate2(d_comp, d_desc, n_blocks, d_out, d_status, 1, d_scratch, c->stream));
    ZCHK(hipEventRecord(e1, c->stream));
    ZCHK(hipStreamSynchronize(c->stream));
    if (kernel_ms) { float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); *kernel_ms = ms; }
    ZCHK(hipMemcpy(status, d_status, (size_t)n_blocks * 4, hipMemcpyDeviceToHost));
    if (utot) ZCHK(hipMemcpy(out, d_out, (size_t)utot, hipMemcpyDeviceToHost));
#undef ZCHK
    cleanup();
    (void)rc;
    return utot;
}

/* ---- device inflate workers for the BAM reader (spx_bam_attach_device_inflate) ---- */
struct spx_inflater {
    spx_ctx *c = nullptr;
    /* copies have streams of their own, shared by the workers: a stream that has run a kernel gets its copies served by the
     * runtime's copy kernel instead of the DMA engines (see spx_ctx::copy_stream) */
    hipStream_t st_h2d = nullptr, st_d2h = nullptr;
    std::mutex mu_h2d, mu_d2h; /* a copy and the event behind it go in together */
    struct Worker {
        hipStream_t st = nullptr; /* the kernel */
        hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_out = nullptr;
        uint8_t *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
        SpxBgzfDesc *h_desc = nullptr, *d_desc = nullptr;
        int32_t *h_status = nullptr, *d_status = nullptr;
        void *d_scratch = nullptr; /* the inflate kernels' (spx_bgzf_inflate_scratch_bytes) */
        size_t in_cap = 0, out_cap = 0, desc_cap = 0;
        std::mutex mu;
    };
    std::vector<Worker> w;
};

extern "C" int spx_inflater_create(spx_ctx *c, int32_t n_workers, spx_inflater **out)
{
    if (!c || !out || n_workers < 1 || n_workers > 32) return fail(SPX_EINVAL, "invalid argument");
    HIPCHK(hipSetDevice(c->device));
    spx_inflater *inf = new spx_inflater();
    inf->c = c;
    inf->w = std::vector<spx_inflater::Worker>((size_t)n_workers);
    bool ok = hipStreamCreateWithFlags(&inf->st_h2d, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&inf->st_d2h, hipStreamNonBlocking) == hipSuccess;
    for (auto &k : inf->w)
        ok = ok && hipStreamCreateWithFlags(&k.st, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&k.ev_in, hipEventDisableTiming | hipEventBlockingSync) == hipSuccess &&
             hipEventCreateWithFlags(&k.ev_k, hipEventDisableTiming | hipEventBlockingSync) == hipSuccess &&
             hipEventCreateWithFlags(&k.ev_out, hipEventDisableTiming | hipEventBlockingSync) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); spx_inflater_free(inf); return fail(SPX_EHIP, "streams of the inflater"); }
    *out = inf;
    return SPX_OK;
}

extern "C" void spx_inflater_free(spx_inflater *inf)
{
    if (!inf) return;
    if (inf->c) (void)hipSetDevice(inf->c->device);
    for (auto &k : inf->w) {
        if (k.st) { (void)hipStreamSynchronize(k.st); (void)hipStreamDestroy(k.st); }
        if (k.ev_in) (void)hipEventDestroy(k.ev_in);
        if (k.ev_k) (void)hipEventDestroy(k.ev_k);
        if (k.ev_out) (void)hipEventDestroy(k.ev_out);
        if (k.h_in) (void)hipHostFree(k.h_in);
        if (k.h_out) (void)hipHostFree(k.h_out);
        if (k.h_desc) (void)hipHostFree(k.h_desc);
        if (k.h_status) (void)hipHostFree(k.h_status);
        if (k.d_in) (void)hipFree(k.d_in);
        if (k.d_out) (void)hipFree(k.d_out);
        if (k.d_desc) (void)hipFree(k.d_desc);
        if (k.d_status) (void)hipFree(k.d_status);
        if (k.d_scratch) (void)hipFree(k.d_scratch);
    }
    if (inf->st_h2d) { (void)hipStreamSynchronize(inf->st_h2d); (void)hipStreamDestroy(inf->st_h2d); }
    if (inf->st_d2h) { (void)hipStreamSynchronize(inf->st_d2h); (void)hipStreamDestroy(inf->st_d2h); }
    delete inf;
}

/* one chunk: the file range of its blocks -> pinned -> HBM, descriptors, kernel, inflated bytes -> pinned -> dst */
extern "C" int spx_inflater_run(void *user, int32_t worker, const uint8_t *file, int64_t file_bytes, const spx_bgzf_block *blocks,
                                int32_t n_blocks, uint8_t *dst, int64_t dst_bytes, int32_t check_crc)
{
    spx_inflater *inf = (spx_inflater *)user;
    if (!inf || worker < 0 || (size_t)worker >= inf->w.size() || !file || !blocks || n_blocks <= 0 || !dst) return -1;
    spx_inflater::Worker &W = inf->w[(size_t)worker];
    std::lock_guard<std::mutex> lk(W.mu);
    if (hipSetDevice(inf->c->device) != hipSuccess) return -1;
    int64_t lo = blocks[0].data_off, hi = 0, umax = 0;
    for (int32_t b = 0; b < n_blocks; ++b) {
        lo = std::min<int64_t>(lo, blocks[b].data_off);
        hi = std::max<int64_t>(hi, blocks[b].data_off + blocks[b].clen);
        umax = std::max<int64_t>(umax, (int64_t)blocks[b].uoff + blocks[b].ulen);
        if (blocks[b].ulen > 65536) return 1;
    }
    lo &= ~(int64_t)3; /* the kernel reads aligned dwords */
    if (lo < 0 || hi > file_bytes || umax > dst_bytes) return -1;
    const size_t in_bytes = (size_t)(hi - lo), out_bytes = (size_t)umax;
#define WCHK(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); return -1; } } while (0)
    if (in_bytes + 64 > W.in_cap) {
        if (W.h_in) (void)hipHostFree(W.h_in);
        if (W.d_in) (void)hipFree(W.d_in);
        W.h_in = nullptr; W.d_in = nullptr;
        W.in_cap = in_bytes + in_bytes / 4 + ((size_t)1 << 20);
        WCHK(hipHostMalloc((void **)&W.h_in, W.in_cap, hipHostMallocDefault));
        WCHK(hipMalloc((void **)&W.d_in, W.in_cap));
    }
    if (out_bytes + 64 > W.out_cap) {
        if (W.h_out) (void)hipHostFree(W.h_out);
        if (W.d_out) (void)hipFree(W.d_out);
        W.h_out = nullptr; W.d_out = nullptr;
        W.out_cap = out_bytes + out_bytes / 4 + ((size_t)1 << 20);
        WCHK(hipHostMalloc((void **)&W.h_out, W.out_cap, hipHostMallocDefault));
        WCHK(hipMalloc((void **)&W.d_out, W.out_cap));
    }
    if ((size_t)n_blocks > W.desc_cap) {
        if (W.h_desc) (void)hipHostFree(W.h_desc);
        if (W.d_desc) (void)hipFree(W.d_desc);
        if (W.h_status) (void)hipHostFree(W.h_status);
        if (W.d_status) (void)hipFree(W.d_status);
        if (W.d_scratch) (void)hipFree(W.d_scratch);
        W.h_desc = nullptr; W.d_desc = nullptr; W.h_status = nullptr; W.d_status = nullptr; W.d_scratch = nullptr;
        W.desc_cap = (size_t)n_blocks + 1024;
        WCHK(hipHostMalloc((void **)&W.h_desc, W.desc_cap * sizeof(SpxBgzfDesc), hipHostMallocDefault));
        WCHK(hipMalloc((void **)&W.d_desc, W.desc_cap * sizeof(SpxBgzfDesc)));
        WCHK(hipHostMalloc((void **)&W.h_status, W.desc_cap * 4, hipHostMallocDefault));
        WCHK(hipMalloc((void **)&W.d_status, W.desc_cap * 4));
        WCHK(hipMalloc(&W.d_scratch, spx_bgzf_inflate_scratch_bytes((int32_t)W.desc_cap) + 64));
    }
    memcpy(W.h_in, file + lo, in_bytes);
    memset(W.h_in + in_bytes, 0, 64);
    for (int32_t b = 0; b < n_blocks; ++b) {
        SpxBgzfDesc &d = W.h_desc[b];
        d.in_off = blocks[b].data_off - lo;
        d.out_off = blocks[b].uoff;
        d.clen = blocks[b].clen; d.ulen = blocks[b].ulen; d.crc = blocks[b].crc; d.pad = 0;
    }
    {
        std::lock_guard<std::mutex> cl(inf->mu_h2d);
        WCHK(hipMemcpyAsync(W.d_in, W.h_in, in_bytes + 64, hipMemcpyHostToDevice, inf->st_h2d));
        WCHK(hipMemcpyAsync(W.d_desc, W.h_desc, (size_t)n_blocks * sizeof(SpxBgzfDesc), hipMemcpyHostToDevice, inf->st_h2d));
        WCHK(hipEventRecord(W.ev_in, inf->st_h2d));
    }
    WCHK(hipStreamWaitEvent(W.st, W.ev_in, 0));
    WCHK(spx_launch_bgzf_inflate2(W.d_in, W.d_desc, n_blocks, W.d_out, W.d_status, check_crc, W.d_scratch, W.st));
    WCHK(hipEventRecord(W.ev_k, W.st));
    /* the HOST waits for the kernel (asleep): a copy stream that waits for a kernel's event is served by the copy kernel
     * from then on, like one that has run a kernel itself */
    WCHK(hipEventSynchronize(W.ev_k));
    {
        std::lock_guard<std::mutex> cl(inf->mu_d2h);
        WCHK(hipMemcpyAsync(W.h_status, W.d_status, (size_t)n_blocks * 4, hipMemcpyDeviceToHost, inf->st_d2h));
        WCHK(hipMemcpyAsync(W.h_out, W.d_out, out_bytes, hipMemcpyDeviceToHost, inf->st_d2h));
        WCHK(hipEventRecord(W.ev_out, inf->st_d2h));
    }
    WCHK(hipEventSynchronize(W.ev_out));
#undef WCHK
    int rc = 0;
    for (int32_t b = 0; b < n_blocks; ++b) {
        if (W.h_status[b] == -4) rc = rc ? rc : 2;
        else if (W.h_status[b] != 0) rc = 1;
    }
    memcpy(dst, W.h_out, out_bytes);
    return rc;
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
    ri.build(ref);
    spx_plan *p = new spx_plan();
    int nthr = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    int rc = spx::host_plan(&bt, 1, ri, par, nthr, p->hb);
    if (rc) { delete p; return fail(rc, "host plan failed"); }
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

/* diagnostics: the work list the DEVICE built, copied back in the shape of a host plan, so that a test can compare
 * the two field by field (groups with an error are left out, as in the host plan) */
extern "C" int spx_work_export(spx_ctx *c, spx_work *w, spx_plan **out)
{
    if (!c || !w || !out) return fail(SPX_EINVAL, "NULL argument");
    if (!w->staged || !w->prepared) return fail(SPX_EINVAL, "not a device-prepared work list");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->lane[w->lane].stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->result_stream));
    spx_plan *p = new spx_plan();
    spx::HostBatch &hb = p->hb;
    const size_t np = (size_t)w->n_prob_dev, nr = (size_t)w->n_rows_dev, nm = (size_t)w->n_mk_dev, ng = (size_t)w->n_dgroups,
                 nq = (size_t)w->tot.n_qe;
    const spx_dev_batch &B = w->cls_batch[0];
    auto pull = [&](auto &vec, const void *src, size_t n) -> bool {
        vec.resize(n);
        return n == 0 || hipMemcpy(vec.data(), src, n * sizeof(vec[0]), hipMemcpyDeviceToHost) == hipSuccess;
    };
    bool ok = pull(hb.ref_nib, B.ref_nib, np) && pull(hb.qry_nib, B.qry_nib, np) && pull(hb.L, B.L, np) && pull(hb.R, B.R, np) &&
              pull(hb.bw, B.bw, np) && pull(hb.row_off, B.row_off, np) && pull(hb.n_rows, B.n_rows, np) && pull(hb.hmm, B.hmm, np * SPX_H_N) &&
              pull(hb.rows, B.rows, nr) && pull(hb.row_expect, B.row_expect, nr) && pull(hb.row_rawq, B.row_rawq, nr) &&
              pull(hb.qe_rec, w->d_qe[0], nq) && pull(hb.qe_pos, w->d_qe[1], nq) && pull(hb.qe_len, w->d_qe[2], nq) &&
              pull(hb.qe_row0, w->d_qe[3], nq) && pull(hb.qe_batch, w->d_qe[4], nq);
    const spx::StageLayout &L = w->stage.lay;
    ok = ok && pull(hb.qry4, (const char *)w->in_arena + w->o_code, (size_t)(spx::kCodeLeadBytes + L.seq_bytes + spx::kCodeTailBytes));
    hb.qry_nibbles = (int64_t)hb.qry4.size() * 2;
    std::vector<spx_dev_marker> mk;
    std::vector<int32_t> mkref, mkfirst, rfe, rfs, atid;
    std::vector<uint8_t> naln;
    std::vector<uint16_t> sec;
    std::vector<spx_group_info> info;
    ok = ok && pull(mk, w->dg.markers, nm) && pull(mkref, w->d_mk_ref_pos, nm) && pull(mkfirst, w->d_mk_first, ng + 1) &&
         pull(rfe, w->d_rfe, ng * 10) && pull(rfs, w->d_rfs, ng * 10) && pull(atid, w->d_atid, ng * 10) && pull(naln, w->dg.n_aln, ng) &&
         pull(sec, w->dg.sec_mask, ng) && pull(info, w->d_info, ng);
    if (!ok) { delete p; return fail(SPX_EHIP, "copy back failed"); }
    /* reference window of a problem from its nibble address */
    hb.ref_tid.assign(np, -1);
    hb.ref_rfs.assign(np, 0);
    for (size_t q = 0; q < np; ++q)
        for (size_t t = 0; t < c->ref.nib_off.size(); ++t)
            if (hb.ref_nib[q] >= c->ref.nib_off[t] && hb.ref_nib[q] < c->ref.nib_off[t] + c->ref.len[t]) {
                hb.ref_tid[q] = (int32_t)t;
                hb.ref_rfs[q] = (int32_t)(hb.ref_nib[q] - c->ref.nib_off[t]);
                break;
            }
    hb.grp_error = w->stage.grp_error;
    hb.mk_first.assign(1, 0);
    for (size_t k = 0; k < ng; ++k) {
        if (info[k].err) { hb.grp_error[(size_t)w->stage.grp_index[k]] = info[k].err; continue; }
        hb.grp_index.push_back(w->stage.grp_index[k]);
        for (int32_t m = mkfirst[k]; m < mkfirst[k + 1]; ++m) { hb.markers.push_back(mk[(size_t)m]); hb.mk_ref_pos.push_back(mkref[(size_t)m]); }
        hb.mk_first.push_back((int32_t)hb.markers.size());
        hb.n_aln.push_back(naln[k]);
        hb.sec_mask.push_back(sec[k]);
        for (int i = 0; i < 10; ++i) { hb.rfe.push_back(rfe[k * 10 + i]); hb.rfs.push_back(rfs[k * 10 + i]); hb.atid.push_back(atid[k * 10 + i]); }
        hb.grp_problems.push_back(info[k].n_prob);
        hb.grp_cells.push_back(info[k].cells);
        hb.dp_cells += info[k].cells;
    }
    for (const spx_dev_marker &m : hb.markers) {
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
