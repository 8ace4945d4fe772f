/*
 * secphase_main.cpp -- `secphase` command line on top of libspx (MI355X scoring path).
 *
 * Same options, presets, defaults and output files as the reference's main()
 * (/root/reference/programs/src/secphase.c:387-743): -i/--inputBam, -f/--inputFasta, -o/--outDir,
 * -P/--prefix, -x/--hifi, -y/--ont, -q -c -d -e -b -t -s -m -p -r -n, -@/--threads, --flankMargin.
 * Marker mode only: -v/--inputVcf, -B, -g, -G (variant mode) are refused with a message; -M (disable marker
 * mode) leaves nothing to do.  -w/--writeBam writes <prefix>.quality_modified.out.bam exactly as the reference
 * does: SAM text (sam_open(path, "w"), src/secphase.c:643-652) holding the records of every dispatched group with
 * the qualities calc_local_baq left in them (src/secphase.c:182-189).  All six output files the WDLs glob for
 * (wdls/workflows/secphase.wdl:100-107) are created; the three variant-mode BEDs stay empty, as they do in
 * the reference when no VCF is given.
 *
 * Pipeline: the reader (spx_io.cpp) inflates BGZF blocks on a pool of -@ threads and cuts the name-grouped BAM into
 * batches of --groupsPerBatch groups; batches are dealt round-robin to the devices of --devices (one scoring context
 * and one in-order pipeline per GPU: staging into pinned memory, copy to HBM, device preparation, DP + scoring
 * kernels, one packed result copy back); results are taken in FILE order, finalised with one rand() stream
 * (= the reference at -@1) and appended to <prefix>.out.log; BED bookkeeping and the release of device memory run
 * on a helper thread.  Start-up is overlapped: HIP initialisation per device, the FASTA parse and the first batches
 * of the BAM run concurrently.
 * Multi-GPU: reads shard over the devices batch by batch; what comes back from every device is the packed decision
 * record of each group (spx_group_out, one device-to-host copy per batch) -- the consumer of the decisions is the
 * host-side finalizer, so the gather IS that copy: no RCCL hop through another GPU is needed inside one process
 * (bench.py's one-process-per-GPU job gathers over RCCL, spx_gather.cpp).
 */
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spx.h"

extern "C" void spx_internal_cpu_report(FILE *f);              /* SPX_TIMING: core-seconds of the host side by kind of work */
extern "C" void spx_internal_cpu_add(int kind, double seconds);

static double thread_cpu_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static const char *timestamp()
{
    static char buf[64];
    time_t t = time(NULL);
    struct tm *tm = localtime(&t);
    snprintf(buf, sizeof buf, "%04d-%02d-%02d %02d:%02d:%02d", tm->tm_year + 1900, tm->tm_mon + 1, tm->tm_mday, tm->tm_hour,
             tm->tm_min, tm->tm_sec);
    return buf;
}

static struct option long_options[] = {{"inputBam", required_argument, NULL, 'i'},
                                       {"inputFasta", required_argument, NULL, 'f'},
                                       {"inputVcf", required_argument, NULL, 'v'},
                                       {"disableMarkerMode", no_argument, NULL, 'M'},
                                       {"baq", no_argument, NULL, 'q'},
                                       {"gapOpen", required_argument, NULL, 'd'},
                                       {"gapExt", required_argument, NULL, 'e'},
                                       {"bandwidth", required_argument, NULL, 'b'},
                                       {"consensus", no_argument, NULL, 'c'},
                                       {"indelThreshold", required_argument, NULL, 't'},
                                       {"initQ", required_argument, NULL, 's'},
                                       {"minQ", required_argument, NULL, 'm'},
                                       {"primMarginScore", required_argument, NULL, 'p'},
                                       {"primMarginRandom", required_argument, NULL, 'r'},
                                       {"minScore", required_argument, NULL, 'n'},
                                       {"hifi", no_argument, NULL, 'x'},
                                       {"ont", no_argument, NULL, 'y'},
                                       {"minVariantMargin", required_argument, NULL, 'g'},
                                       {"prefix", required_argument, NULL, 'P'},
                                       {"outDir", required_argument, NULL, 'o'},
                                       {"variantBed", required_argument, NULL, 'B'},
                                       {"minGQ", required_argument, NULL, 'G'},
                                       {"threads", required_argument, NULL, '@'},
                                       {"writeBam", no_argument, NULL, 'w'},
                                       {"flankMargin", required_argument, NULL, 'F'},
                                       {"groupsPerBatch", required_argument, NULL, 1001},
                                       {"device", required_argument, NULL, 1002},
                                       {"devices", required_argument, NULL, 1003},
                                       {"gpuInflate", required_argument, NULL, 1004},
                                       {"hostInput", no_argument, NULL, 1005},
                                       {"segmentMB", required_argument, NULL, 1006},
                                       {NULL, 0, NULL, 0}};

static void usage(const char *prog)
{
    fprintf(stderr, "\nUsage: %s  -i <INPUT_BAM> -f <FASTA> \n", prog);
    fprintf(stderr,
            "Options (as in secphase; marker mode only):\n"
            "         --inputBam, -i         Input BAM file (grouped by read name, cs tag present)\n"
            "         --inputFasta, -f       Input FASTA file\n"
            "         --outDir, -o           Output dir [secphase_out_dir]\n"
            "         --prefix, -P           Prefix of the output files [secphase]\n"
            "         --hifi, -x             [-q -c -t10 -d 1e-4 -e 0.1 -b20 -m10 -s40 -p40 -r0 -n -10]\n"
            "         --ont, -y              [-q -c -t20 -d 1e-3 -e 0.1 -b20 -m10 -s20 -p20 -r0 -n -10]\n"
            "         --baq -q, --gapOpen -d, --gapExt -e, --bandwidth -b, --consensus -c, --indelThreshold -t,\n"
            "         --initQ -s, --minQ -m, --primMarginScore -p, --primMarginRandom -r, --minScore -n, --flankMargin\n"
            "         --threads, -@          host threads for BGZF inflation and group preparation [4]\n"
            "         --groupsPerBatch       read groups per GPU work list [16384]\n"
            "         --device               GPU index [0]\n"
            "         --devices              several GPUs, e.g. 0-7 or 0,2,5: batches are dealt round-robin, the output is\n"
            "                                the same file-order list\n"
            "         --gpuInflate           (with --hostInput) BGZF inflate workers per GPU beside the host threads [8]; 0: host only\n"
            "         --segmentMB            device-resident input: inflated MB per segment = per GPU work list [1024]\n"
            "         --hostInput            read, inflate and parse the BAM on the host (the round-3 reader) instead of on the\n"
            "                                device(s); implied by --writeBam\n"
            "         --writeBam, -w         Write <prefix>.quality_modified.out.bam (SAM text, as the reference does) with\n"
            "                                the base qualities modified by BAQ\n"
            "Not supported by this build: --inputVcf/-v, --variantBed/-B, -g, -G (variant mode)\n");
}

int main(int argc, char *argv[])
{
    setenv("GPU_MAX_HW_QUEUES", "16", 0); /* one hardware queue per stream of the scoring context */
    /* the command line's work lists are one ~1 GB input segment each (18 k HiFi / 8 k ONT groups: preparation pools of a few GB per lane):
     * six preparation lanes instead of the library's four (scoring loop of 524 288 HiFi groups 1.377 -> 1.325 s) */
    setenv("SPX_PREP_LANES", "6", 0);
    /* defaults: src/secphase.c:420-449 */
    spx_params par;
    memset(&par, 0, sizeof par);
    par.baq_flag = 0; par.consensus = 0; par.indel_threshold = 10; par.min_q = 10; par.min_score = -10;
    par.prim_margin_score = 40; par.prim_margin_random = 0; par.set_q = 40; par.conf_d = 1e-4; par.conf_e = 0.1;
    par.conf_b = 20; par.flank_margin = 500;
    std::string inputPath, fastaPath, prefix = "secphase", dirPath = "secphase_out_dir";
    bool preset_ont = false, preset_hifi = false, marker_mode = true, write_bam = false, batch_given = false;
    bool host_input = getenv("SPX_HOST_INPUT") != nullptr;
    int segment_mb = 0;
    int threads = 4, groups_per_batch = 16384, gpu_inflate = 8, c;
    if (const char *e = getenv("SPX_GPU_INFLATE")) gpu_inflate = atoi(e);
    std::vector<int> devices;
    auto parse_devices = [&](const char *txt) { /* "0-3", "0,2,5", "1" */
        devices.clear();
        const char *q = txt;
        while (*q) {
            char *e = nullptr;
            long a = strtol(q, &e, 10), b = a;
            if (e == q) return false;
            if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); if (e == q) return false; }
            for (long d = a; d <= b && devices.size() < 64; ++d) devices.push_back((int)d);
            q = e;
            if (*q == ',') ++q; else if (*q) return false;
        }
        return !devices.empty();
    };
    const char *prog = strrchr(argv[0], '/') ? strrchr(argv[0], '/') + 1 : argv[0];
    while (~(c = getopt_long(argc, argv, "i:p:P:G:o:f:v:qd:e:b:n:r:m:ct:s:B:g:@:wxyMh", long_options, NULL))) {
        switch (c) {
        case 'i': inputPath = optarg; break;
        case 'f': fastaPath = optarg; break;
        case '@': threads = atoi(optarg); break;
        case 'P': prefix = optarg; break;
        case 'o': dirPath = optarg; break;
        case 'x': /* src/secphase.c:477-490 */
            preset_hifi = true;
            par.baq_flag = 1; par.consensus = 1; par.indel_threshold = 10; par.conf_d = 1e-4; par.conf_e = 0.1; par.conf_b = 20;
            par.min_q = 10; par.set_q = 40; par.prim_margin_score = 40; par.prim_margin_random = 0; par.min_score = -10;
            break;
        case 'y': /* src/secphase.c:491-504 */
            preset_ont = true;
            par.baq_flag = 1; par.consensus = 1; par.indel_threshold = 20; par.conf_d = 1e-3; par.conf_e = 0.1; par.conf_b = 20;
            par.min_q = 10; par.set_q = 20; par.prim_margin_score = 20; par.prim_margin_random = 0; par.min_score = -10;
            break;
        case 'q': par.baq_flag = 1; break;
        case 'd': par.conf_d = atof(optarg); break;
        case 'e': par.conf_e = atof(optarg); break;
        case 'b': par.conf_b = atof(optarg); break;
        case 'c': par.consensus = 1; break;
        case 't': par.indel_threshold = atoi(optarg); break;
        case 's': par.set_q = atoi(optarg); break;
        case 'm': par.min_q = atoi(optarg); break;
        case 'p': par.prim_margin_score = atof(optarg); break;
        case 'r': par.prim_margin_random = atof(optarg); break;
        case 'n': par.min_score = atoi(optarg); break;
        case 'F': par.flank_margin = atoi(optarg); break;
        case 'M': marker_mode = false; break;
        case 1001: groups_per_batch = atoi(optarg); batch_given = true; break;
        case 1004: gpu_inflate = atoi(optarg); break;
        case 1005: host_input = true; break;
        case 1006: segment_mb = atoi(optarg); break;
        case 1002: case 1003:
            if (!parse_devices(optarg)) { fprintf(stderr, "[%s] cannot parse the device list %s\n", timestamp(), optarg); return 1; }
            break;
        case 'v': case 'B': case 'g': case 'G':
            fprintf(stderr, "[%s] variant mode (-v/-B/-g/-G) is not part of this build: marker mode only\n", timestamp());
            return 2;
        case 'w': write_bam = true; break;
        default:
            if (c != 'h') fprintf(stderr, "[E::%s] undefined option %c\n", __func__, c);
            usage(prog);
            return 1;
        }
    }
    if (inputPath.empty() || fastaPath.empty()) { usage(prog); return 1; }
    if (preset_ont && preset_hifi) {
        fprintf(stderr, "[%s] Presets --hifi and --ont cannot be enabled at the same time. Select only one of them!\n", timestamp());
        return EXIT_FAILURE;
    }
    if (groups_per_batch < 1) groups_per_batch = 1;
    /* the device-side scans of a work list handle 2^20 alignments (SPX_MAX_STAGE_SLOTS): 11 records per group at most */
    if (groups_per_batch > 95000) groups_per_batch = 95000;
    if (devices.empty()) devices.push_back(0);
    /* with -w every base of every realigned window keeps its forward row in HBM until the MAP kernel has run
     * (~0.7 KB per base): smaller work lists */
    if (write_bam && !batch_given) groups_per_batch = 1024;
    if (write_bam) par.flags |= SPX_PAR_ALL_ROWS;
    struct stat st;
    if (stat(dirPath.c_str(), &st) == -1) mkdir(dirPath.c_str(), 0777);
    auto out_path = [&](const char *suffix) { return dirPath + "/" + prefix + suffix; };

    /* files that only variant mode fills: created empty (src/secphase.c:619-630,715-727) */
    for (const char *sfx : {".initial_variant_blocks.bed", ".modified_read_blocks.variants.bed", ".variant_blocks.bed"}) {
        FILE *f = fopen(out_path(sfx).c_str(), "w");
        if (f) fclose(f);
    }
    const std::string log_path = out_path(".out.log");
    { FILE *f = fopen(log_path.c_str(), "w"); if (!f) { fprintf(stderr, "cannot write %s\n", log_path.c_str()); return 1; } fclose(f); }

    const double t_proc0 = now_s();
    const int n_dev = (int)devices.size();
    int depth = 3; /* batches in flight per device */
    if (const char *e = getenv("SPX_DEPTH")) depth = std::max(1, std::min(8, atoi(e)));
    /* ---- start-up, overlapped: (a) one thread per device initialises HIP and creates the scoring context, (b) the BAM
     * reader starts inflating and cutting batches right away, (c) this thread parses the FASTA; then the reference goes
     * to every device and the loop starts with batches already waiting ---- */
    /* (device-resident input stages nothing on the host: the contexts need not pin their 512 MB staging ring behind our back -- 0.15 s of
     * pinning that the input pipelines' own pinned buffers queued behind, and as much again when the contexts are destroyed; a host-side
     * staging that happens after all allocates its chunks when it needs them) */
    if (marker_mode && !write_bam && !host_input) setenv("SPX_NO_WARM", "1", 0);
    std::vector<spx_ctx *> ctxs((size_t)n_dev, nullptr);
    std::vector<int> ctx_rc((size_t)n_dev, SPX_OK);
    std::vector<std::string> ctx_err((size_t)n_dev);
    std::vector<std::thread> ctx_threads;
    for (int d = 0; d < n_dev; ++d)
        ctx_threads.emplace_back([&, d] {
            ctx_rc[(size_t)d] = spx_create(devices[(size_t)d], &ctxs[(size_t)d]);
            if (ctx_rc[(size_t)d] != SPX_OK) ctx_err[(size_t)d] = spx_last_error();
        });
    auto join_ctx = [&] { for (auto &t : ctx_threads) if (t.joinable()) t.join(); };
    spx_bam_options bo;
    spx_bam_default_options(&bo);
    bo.threads = threads;
    /* (tests: SPX_GPU_INFLATE_FIRST holds the reader back until the device inflate workers are attached, so that even a tiny
     * file goes through them) */
    const bool inflate_first = getenv("SPX_GPU_INFLATE_FIRST") != nullptr;
    bo.batch_groups = (marker_mode && !inflate_first) ? groups_per_batch : 0;
    bo.ahead_batches = 2;
    bo.keep_batches = n_dev * (depth + 1) + 2 * n_dev + 6;
    bo.ahead_batches = 3; /* the reader keeps cutting batches while the devices start up */
    if (const char *e = getenv("SPX_BAM_AHEAD")) bo.ahead_batches = std::max(1, atoi(e));
    /* The input side.  Default: DEVICE-RESIDENT -- the host sends compressed BGZF blocks, inflate / record chain / fields /
     * name groups / dispatch filter / staging run on the device(s), one input pipeline per GPU (spx_devin.cpp).  --hostInput
     * (and -w, whose output needs every record's bytes on the host): the round-3 host reader. */
    const bool dev_input = marker_mode && !write_bam && !host_input;
    auto die = [](int code) { fflush(NULL); _exit(code); }; /* (reader / pool / context threads are running: no static destructors under them) */
    spx_bam_reader *bam = nullptr;
    spx_dbam *dbam = nullptr;
    if (dev_input) {
        spx_dbam_options dopt;
        spx_dbam_default_options(&dopt);
        dopt.threads = std::max(1, std::min(threads, spx_effective_cpus()));
        if (batch_given) dopt.max_groups = groups_per_batch;
        if (segment_mb > 0) dopt.segment_bytes = (int64_t)segment_mb << 20;
        dopt.ahead = 2;
        if (spx_dbam_open(inputPath.c_str(), &dopt, &dbam) != SPX_OK) { fprintf(stderr, "[%s] %s\n", timestamp(), spx_last_error()); join_ctx(); die(1); }
        bam = spx_dbam_header(dbam);
    } else if (spx_bam_open_opts(inputPath.c_str(), &bo, &bam) != SPX_OK) { fprintf(stderr, "[%s] %s\n", timestamp(), spx_io_last_error()); join_ctx(); return 1; }
    const double t_bam_open = now_s();
    spx_fasta *fa = nullptr;
    if (spx_fasta_load(fastaPath.c_str(), &fa) != SPX_OK) { fprintf(stderr, "[%s] %s\n", timestamp(), spx_io_last_error()); join_ctx(); die(1); }
    const spx_ref *ref = spx_fasta_ref(fa);
    const double t_fasta = now_s();
    int missing = spx_bam_bind_reference(bam, ref);
    if (missing > 0) fprintf(stderr, "[%s] warning: %d BAM target(s) are not in the FASTA; reads on them are skipped\n", timestamp(), missing);

    spx_sam_writer *sam = nullptr;
    if (write_bam && spx_sam_open(out_path(".quality_modified.out.bam").c_str(), bam, &sam) != SPX_OK) {
        fprintf(stderr, "[%s] %s\n", timestamp(), spx_io_last_error());
        join_ctx();
        die(1);
    }

    join_ctx();
    const double t_ctx = now_s();
    int rc = SPX_OK;
    for (int d = 0; d < n_dev; ++d)
        if (ctx_rc[(size_t)d] != SPX_OK) { fprintf(stderr, "[%s] device %d: %s: %s\n", timestamp(), devices[(size_t)d], spx_strerror(ctx_rc[(size_t)d]), ctx_err[(size_t)d].c_str()); die(1); }
    /* the input pipelines start NOW: compressed blocks go up and are inflated / parsed / staged while the assembly is still
     * being packed and copied to the device(s) (staging does not need it; the first spx_pipe_submit comes after it) */
    if (dev_input && (rc = spx_dbam_start(dbam, ctxs.data(), n_dev, &par)) != SPX_OK) {
        fprintf(stderr, "[%s] %s: %s\n", timestamp(), spx_strerror(rc), spx_last_error());
        die(1);
    }
    const double t_dstart = now_s();
    { /* the assembly into every device's HBM, in parallel */
        std::vector<std::thread> th;
        for (int d = 0; d < n_dev; ++d)
            th.emplace_back([&, d] {
                ctx_rc[(size_t)d] = spx_set_reference(ctxs[(size_t)d], ref);
                if (ctx_rc[(size_t)d] != SPX_OK) ctx_err[(size_t)d] = spx_last_error();
            });
        for (auto &t : th) t.join();
        for (int d = 0; d < n_dev; ++d)
            if (ctx_rc[(size_t)d] != SPX_OK) { fprintf(stderr, "[%s] device %d: %s: %s\n", timestamp(), devices[(size_t)d], spx_strerror(ctx_rc[(size_t)d]), ctx_err[(size_t)d].c_str()); die(1); }
    }
    const double t_ref = now_s();
    /* BGZF inflate on the device(s) beside the host pool: the host pool claims chunks from the front of the reader's queue, idle device
     * workers from its back */
    struct InflateRoute { std::vector<spx_inflater *> inf; int per_dev = 0; } route;
    if (gpu_inflate > 0 && marker_mode && !dev_input) {
        /* the reader takes at most 32 device workers in all: with many devices each contributes fewer */
        route.per_dev = std::max(1, std::min(std::min(gpu_inflate, 16), 32 / n_dev));
        for (int d = 0; d < n_dev && (int)route.inf.size() * route.per_dev + route.per_dev <= 32; ++d) {
            spx_inflater *inf = nullptr;
            if (spx_inflater_create(ctxs[(size_t)d], route.per_dev, &inf) == SPX_OK) route.inf.push_back(inf);
        }
        if (!route.inf.empty()) {
            auto fn = [](void *user, int32_t worker, const uint8_t *file, int64_t file_bytes, const spx_bgzf_block *blocks, int32_t n_blocks,
                         uint8_t *dst, int64_t dst_bytes, int32_t check_crc) -> int {
                InflateRoute *R = (InflateRoute *)user;
                const size_t nd = R->inf.size();
                return spx_inflater_run(R->inf[(size_t)worker % nd], (int32_t)((size_t)worker / nd), file, file_bytes, blocks, n_blocks, dst, dst_bytes,
                                        check_crc);
            };
            if (spx_bam_attach_device_inflate(bam, fn, &route, (int32_t)(route.inf.size() * (size_t)route.per_dev)) != SPX_OK) {
                fprintf(stderr, "[%s] warning: the device inflate workers could not be attached; BGZF blocks are inflated on the host only\n", timestamp());
                for (spx_inflater *inf : route.inf) spx_inflater_free(inf);
                route.inf.clear();
            }
        }
    }

    spx_finalizer *fin = nullptr;
    spx_finalizer_create(1, &fin); /* unseeded rand() == srand(1) */
    spx_bedset *bed_mod = nullptr, *bed_mk = nullptr;
    spx_bedset_create(&bed_mod);
    spx_bedset_create(&bed_mk);

    fprintf(stderr, "[%s] Started parsing alignments\n", timestamp());
    if (getenv("SPX_TIMING"))
        fprintf(stderr, "[%s] start-up %.3f s: BAM open (header) %.3f, FASTA parse %.3f, waiting for the device context(s) %.3f, input pipelines started %.3f, reference to HBM %.3f\n",
                timestamp(), t_ref - t_proc0, t_bam_open - t_proc0, t_fasta - t_bam_open, t_ctx - t_fasta, t_dstart - t_ctx, t_ref - t_dstart);
    long long n_alns = 0, n_reads = 0, n_modified = 0, n_rejected = 0;
    double t_read = 0, t_wait = 0, t_out = 0, t_start = now_s(), t_hostprep = 0, t_kernel = 0;
    double t_fin = 0;
    /* the reference hands every group to a pool thread and serialises the output with a mutex; here whole batches
     * flow through in-order pipelines (one per device): while batch k is on a GPU, batch k+1 is staged and copied,
     * batch k+2 is inflated by the reader, and the results of batch k-1 are written -- in file order */
    std::vector<spx_pipe *> pipes((size_t)n_dev, nullptr);
    /* staging copies into pinned memory: threads beyond the container's CPU quota only get the process throttled */
    const int stage_threads = std::max(1, std::min(threads, spx_effective_cpus()) / n_dev);
    for (int d = 0; d < n_dev && marker_mode; ++d)
        if ((rc = spx_pipe_create(ctxs[(size_t)d], &par, depth, stage_threads, &pipes[(size_t)d])) != SPX_OK) {
            fprintf(stderr, "[%s] %s: %s\n", timestamp(), spx_strerror(rc), spx_last_error());
            die(1);
        }
    /* BED bookkeeping (marker arrays come back from the device) and the release of a work list's device memory happen on
     * a helper thread: the sets are order-independent (sorted and merged at the end), only the counts come back */
    struct Post { spx_work *w; int lane; const spx_batch *bt; std::vector<spx_group_out> out; };
    std::mutex post_mu;
    std::condition_variable post_cv;
    std::deque<Post> post_q;
    bool post_stop = false;
    long long post_modified = 0;
    int post_pending = 0;
    bool post_failed = false;
    double t_bed = 0, t_free = 0, t_log = 0;
    std::thread post_thread([&] {
        for (;;) {
            Post ps;
            {
                std::unique_lock<std::mutex> lk(post_mu);
                post_cv.wait(lk, [&] { return post_stop || !post_q.empty(); });
                if (post_q.empty()) return;
                ps = std::move(post_q.front());
                post_q.pop_front();
            }
            /* the queue is FIFO and this is its only consumer: the list grows in file order */
            const double t9 = now_s();
            if (spx_write_relabel_log(log_path.c_str(), "a", ps.bt, ref, ps.out.data()) != SPX_OK) post_failed = true;
            if (dev_input) spx_dbam_release(dbam, ps.bt);
            else spx_bam_release_batch(bam, ps.bt); /* its share of the inflate arena is recycled */
            const double ta = now_s();
            const double cpu0 = thread_cpu_s();
            const int nm = spx_relabel_blocks(ps.w, ref, ps.out.data(), bed_mod, bed_mk);
            spx_internal_cpu_add(8 /* BED bookkeeping */, thread_cpu_s() - cpu0);
            const double tb = now_s();
            spx_work_free(ctxs[(size_t)ps.lane], ps.w);
            const double tc = now_s();
            std::lock_guard<std::mutex> lk(post_mu);
            if (nm > 0) post_modified += nm;
            t_log += ta - t9;
            t_bed += tb - ta;
            t_free += tc - tb;
            --post_pending;
            post_cv.notify_all();
        }
    });
    auto stop_post = [&] {
        {
            std::lock_guard<std::mutex> lk(post_mu);
            post_stop = true;
        }
        post_cv.notify_all();
        if (post_thread.joinable()) post_thread.join();
    };
    bool eof = false;
    long long submitted = 0, received = 0;
    std::vector<spx_group_out> out;
    int fail_rc = 0;
    struct Sub { const spx_batch *bt; int lane; int ng; };
    std::deque<Sub> order; /* submissions in file order: which pipeline holds them */
    /* the next batch, fetched but not yet submitted: the device input deals segments to whichever GPU is free, so several
     * lists in a row may belong to ONE pipeline -- a full pipeline is drained (oldest submission first, in file order)
     * before the batch goes in, instead of blocking in spx_pipe_submit with nobody left to take results */
    struct { bool have = false; const spx_batch *bt = nullptr; spx_work *staged = nullptr; int32_t lane = 0; int ng = 0; } nxt;
    for (;;) {
        if (!nxt.have && !eof && submitted - received < (long long)n_dev * (depth + 1)) {
            nxt.lane = (int32_t)(submitted % n_dev);
            nxt.staged = nullptr;
            double t0 = now_s();
            nxt.ng = dev_input ? spx_dbam_next(dbam, &nxt.staged, &nxt.lane, &nxt.bt) : spx_bam_next_batch(bam, groups_per_batch, &nxt.bt);
            t_read += now_s() - t0;
            if (nxt.ng < 0) { fprintf(stderr, "[%s] BAM read error: %s\n", timestamp(), dev_input ? spx_last_error() : spx_io_last_error()); fail_rc = 1; break; }
            if (nxt.ng == 0) { eof = true; continue; }
            n_alns += nxt.bt->n_alns;
            n_reads += nxt.ng;
            if (!marker_mode) { spx_bam_release_batch(bam, nxt.bt); continue; }
            nxt.have = true;
        }
        if (nxt.have && spx_pipe_pending(pipes[(size_t)nxt.lane]) < depth + 1) {
            rc = dev_input ? spx_pipe_submit(pipes[(size_t)nxt.lane], nullptr, 0, nxt.staged, nxt.ng, (void *)nxt.bt)
                           : spx_pipe_submit(pipes[(size_t)nxt.lane], &nxt.bt, 1, nullptr, 0, (void *)nxt.bt);
            if (rc != SPX_OK) {
                fprintf(stderr, "[%s] %s: %s\n", timestamp(), spx_strerror(rc), spx_last_error());
                fail_rc = 1;
                break;
            }
            order.push_back(Sub{nxt.bt, nxt.lane, nxt.ng});
            ++submitted;
            nxt.have = false;
            continue;
        }
        if (submitted == received) {
            if (eof && !nxt.have) break;
            continue;
        }
        const Sub sub = order.front();
        order.pop_front();
        const int lane = sub.lane;
        spx_work *w = nullptr;
        void *tag = nullptr;
        double t0 = now_s();
        out.resize((size_t)sub.ng + 1);
        const int ng = spx_pipe_next(pipes[(size_t)lane], out.data(), sub.ng, &w, &tag);
        t_wait += now_s() - t0;
        ++received;
        if (ng < 0) { fprintf(stderr, "[%s] %s: %s\n", timestamp(), spx_strerror(ng), spx_last_error()); fail_rc = 1; break; }
        out.resize((size_t)ng);
        const spx_batch *bt = (const spx_batch *)tag;
        { spx_stats st; spx_work_stats(w, &st); t_hostprep += st.prep_seconds; t_kernel += st.kernel_seconds; }
        for (int g = 0; g < ng; ++g)
            if (out[(size_t)g].n_aln == SPX_ENOTAG) { /* the reference stops here: cigar_it.c:64-67 */
                fprintf(stderr, "At least one of the MD or CS tags should be present!\n");
                fail_rc = 1;
            }
        if (fail_rc) break;
        t0 = now_s();
        if (sam) { /* src/secphase.c:182-189: written before the decision, file order = the reference at -@1 */
            /* calc_local_baq edits the record's quality array in place (ptMarker.c:706,759,763); so do we: the batch's bytes
             * live in the reader's arena, belong to this batch alone and are not read again after staging */
            uint8_t *qv = const_cast<uint8_t *>(bt->qual);
            if ((rc = spx_apply_quals(ctxs[(size_t)lane], w, 0, bt, qv)) != SPX_OK) { fprintf(stderr, "[%s] %s: %s\n", timestamp(), spx_strerror(rc), spx_last_error()); fail_rc = 1; break; }
            for (int g = 0; g < ng; ++g)
                if (spx_group_is_dispatched(bt, g) && spx_sam_write_group_of(sam, bam, bt, g, qv) < 0) {
                    fprintf(stderr, "[%s] %s\n", timestamp(), spx_io_last_error());
                    fail_rc = 1;
                    break;
                }
            if (fail_rc) break;
        }
        const double ta = now_s();
        spx_finalizer_apply(fin, &par, out.data(), ng);
        const double tb = now_s();
        for (int g = 0; g < ng; ++g) if (out[(size_t)g].n_aln < 0) ++n_rejected;
        long long shown;
        {
            /* (the helper may lag a few batches behind: their work lists, and the slots of the reader they point into,
             * stay alive meanwhile -- not without bound) */
            std::unique_lock<std::mutex> lk(post_mu);
            post_cv.wait(lk, [&] { return post_pending < 2 * n_dev + 2; });
            post_q.push_back(Post{w, lane, bt, std::move(out)});
            ++post_pending;
            shown = post_modified;
        }
        post_cv.notify_all();
        out = std::vector<spx_group_out>();
        t_out += now_s() - t0;
        t_fin += tb - ta;
        fprintf(stderr, "[%s] #parsed alignments = %lld, #parsed reads = %lld, #modifed by phased variants = 0, #modifed by markers = %lld\n",
                timestamp(), n_alns, n_reads, shown);
    }
    stop_post();
    n_modified = post_modified;
    for (spx_pipe *p : pipes) if (p) spx_pipe_destroy(p);
    if (fail_rc || post_failed) die(1);
    fprintf(stderr, "[%s] time in the scoring loop: %.3f s (BAM read+inflate not hidden by the read-ahead %.3f, waiting for results %.3f, "
                    "finalise+write %.3f); on pipeline threads: staging %.3f; GPU kernels %.3f; %d device(s)\n", timestamp(), now_s() - t_start, t_read, t_wait, t_out, t_hostprep, t_kernel, n_dev);
    if (getenv("SPX_TIMING"))
        fprintf(stderr, "[%s] finalise: draws %.3f; on the helper thread: relabel list %.3f, BED bookkeeping %.3f, work free %.3f\n", timestamp(), t_fin, t_log, t_bed, t_free);
    if (n_rejected) fprintf(stderr, "[%s] %lld read group(s) use constructs the reference leaves undefined and were skipped\n", timestamp(), n_rejected);
    fprintf(stderr, "[%s] Number of reads modified by phased variants = 0\n", timestamp());
    fprintf(stderr, "[%s] Number of reads modified by marker score = %lld\n", timestamp(), n_modified);
    const double t_end0 = now_s();
    /* device input: the input pipelines and the scoring contexts are done -- their device and pinned memory goes back on a
     * thread of its own while the BED sets are merged and written (0.1-0.3 s of host work that needs none of it) */
    std::thread teardown;
    double t_teardown = 0, t_teardown_in = 0;
    int64_t dbam_nseg = 0, dbam_up = 0;
    double dbam_sec[7] = {0};
    if (dev_input) spx_dbam_stats(dbam, &dbam_nseg, &dbam_up, dbam_sec);
    const bool early_teardown = dev_input && !(getenv("SPX_QUICK_EXIT") && atoi(getenv("SPX_QUICK_EXIT")) != 0);
    if (early_teardown)
        teardown = std::thread([&] {
            const double t0 = now_s();
            spx_dbam_close(dbam);
            t_teardown_in = now_s() - t0;
            for (spx_ctx *c_ : ctxs) spx_destroy(c_);
            t_teardown = now_s() - t0;
        });
    const int rc_bed1 = spx_bedset_save(bed_mod, out_path(".modified_read_blocks.markers.bed").c_str(), 1);
    const int rc_bed2 = spx_bedset_save(bed_mk, out_path(".marker_blocks.bed").c_str(), 0);
    if (rc_bed1 != SPX_OK || rc_bed2 != SPX_OK) {
        fprintf(stderr, "[%s] could not write the BED outputs: %s\n", timestamp(), spx_last_error());
        if (teardown.joinable()) teardown.join(); /* (never leave main with the teardown thread running) */
        fflush(NULL);
        if (getenv("SPX_PROFILER")) return 1; /* (a profiler writes its files from an exit handler) */
        _exit(1); /* (reader, pool and context threads are still running: no static destructors under them) */
    }
    spx_bedset_free(bed_mod);
    spx_bedset_free(bed_mk);
    spx_finalizer_free(fin);
    const bool sam_failed = sam && spx_sam_close(sam) != SPX_OK;
    if (teardown.joinable()) teardown.join(); /* before any path that leaves main */
    if (sam_failed) { fprintf(stderr, "[%s] could not finish the quality-modified output\n", timestamp()); return 1; }
    const double t_end1 = now_s();
    if (getenv("SPX_TIMING") && !dev_input) {
        int64_t ch = 0, cd = 0;
        spx_bam_inflate_counts(bam, &ch, &cd);
        fprintf(stderr, "[%s] inflate chunks: %lld on the host pool, %lld on the device(s)\n", timestamp(), (long long)ch, (long long)cd);
    }
    if (getenv("SPX_TIMING") && dev_input) {
        int64_t nseg = dbam_nseg, up = dbam_up;
        double *sec = dbam_sec;
        fprintf(stderr, "[%s] device input: %lld segments, %.2f GB of compressed bytes uploaded; summed over the lanes' threads: upload (copies into pinned memory + "
                        "enqueue) %.3f s, parsing %.3f s = waiting for the carry %.3f + for inflate %.3f + record chain %.3f + fields / groups %.3f + image %.3f\n",
                timestamp(), (long long)nseg, up / 1e9, sec[0], sec[1], sec[2], sec[3], sec[4], sec[5], sec[6]);
    }
    /* How to leave.  Host input: every output is closed, what is left is tens of GB of inflate arena and file mapping to give
     * back page by page, only for the process to end right after -- the pages go back in parallel on the pool and the process
     * _exit()s.  Device input: the host side is small and the DRIVER is what takes long with a process that dies holding
     * gigabytes of device and pinned memory (~0.4 s until the parent's wait returns, against ~0.16 s for freeing them here). */
    const bool quick_exit = getenv("SPX_QUICK_EXIT") ? atoi(getenv("SPX_QUICK_EXIT")) != 0 : (!dev_input && !getenv("SPX_TIDY_EXIT"));
    if (quick_exit) {
        const double t_drop0 = now_s();
        spx_bam_drop_pages(bam); /* (device input: the header reader owns the mapping) */ /* in parallel, instead of by the kernel's single-threaded teardown while the parent waits */
        const double t_drop = now_s() - t_drop0;
        /* Every output file is complete and closed.  What is left is giving back memory -- tens of GB of inflate arena, pinned
         * staging chunks, device arenas -- page by page (munmap, hipHostFree, hipFree: 0.5-0.7 s), only for the process to
         * end right after; the kernel and the driver reclaim all of it at exit anyway.  SPX_TIDY_EXIT=1 runs the orderly
         * shutdown (leak checkers, tests of the close paths). */
        if (getenv("SPX_TIMING")) {
            fprintf(stderr, "[%s] wind-down: BED merge + write %.3f s, reader pages returned %.3f s; whole process %.3f s (the rest of the memory is left to process exit)\n",
                    timestamp(), t_end1 - t_end0, t_drop, now_s() - t_proc0);
            spx_internal_cpu_report(stderr);
            struct timespec rt;
            clock_gettime(CLOCK_REALTIME, &rt); /* (for a parent that times exec -> exit: what lies before main and after _exit) */
            fprintf(stderr, "[spx timing] main() entered at %.3f, leaving at %.3f (epoch seconds)\n", rt.tv_sec + 1e-9 * rt.tv_nsec - (now_s() - t_proc0),
                    rt.tv_sec + 1e-9 * rt.tv_nsec);
        }
        fflush(NULL);
        _exit(0);
    }
    if (dev_input) { if (!early_teardown) spx_dbam_close(dbam); /* (closes its header reader) */ }
    else spx_bam_close(bam);
    for (spx_inflater *inf : route.inf) spx_inflater_free(inf);
    const double t_end2 = now_s();
    if (!early_teardown) for (spx_ctx *c_ : ctxs) spx_destroy(c_);
    const double t_end3 = now_s();
    if (early_teardown && getenv("SPX_TIMING"))
        fprintf(stderr, "[%s] input pipelines and contexts closed beside the BED merge: %.3f s (input pipelines %.3f, contexts %.3f)\n", timestamp(), t_teardown,
                t_teardown_in, t_teardown - t_teardown_in);
    spx_fasta_free(fa);
    if (getenv("SPX_TIMING")) {
        fprintf(stderr, "[%s] wind-down: BED merge + write %.3f s, closing the reader %.3f s, the context(s) %.3f s; whole process %.3f s\n", timestamp(),
                t_end1 - t_end0, t_end2 - t_end1, t_end3 - t_end2, now_s() - t_proc0);
        spx_internal_cpu_report(stderr);
        struct timespec rt;
        clock_gettime(CLOCK_REALTIME, &rt);
        fprintf(stderr, "[spx timing] main() entered at %.3f, leaving at %.3f (epoch seconds)\n", rt.tv_sec + 1e-9 * rt.tv_nsec - (now_s() - t_proc0),
                rt.tv_sec + 1e-9 * rt.tv_nsec);
    }
    fflush(NULL);
    /* everything is freed and closed; no static destructors of the HIP runtime under a finished program -- unless a profiler needs
     * its exit handlers (rocprofv3 writes its files from one): SPX_PROFILER=1 */
    if (dev_input && !getenv("SPX_PROFILER")) _exit(0);
    return 0;
}
