/*
 * spx_pipe.cpp -- in-order scoring pipeline over one context (include/spx.h, spx_pipe_*).
 *
 * The reference hands every read group to a thread pool and serialises the output with a mutex
 * (/root/reference/programs/src/secphase.c:230-351 dispatch, :74-228 worker).  Here a submission is a whole batch of
 * groups; `depth` submissions are in flight at once, each driven by one worker thread through
 *     stage (host threads pack the records, one copy to HBM)  ->  device preparation  ->  DP + scoring kernels  ->
 *     one copy of the packed results back,
 * so that the copy of batch k+2, the preparation of batch k+1 and the DP kernels of batch k overlap on the device (copy,
 * preparation and main streams of the context).  Results come back in SUBMISSION order, which is what the rand()
 * replay of the decision (spx_finalizer_apply) and the relabel list need: the output equals the reference at -@1.
 */
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spx.h"
#include "spx_pool.h"

extern "C" void spx_internal_set_error(const char *msg);
extern "C" int spx_internal_work_claim(spx_work *w, int claim); /* 1: mark in flight (fails when it already is), 0: clear */
struct spx_alloc_gate;
extern "C" void spx_internal_lists_in_flight(spx_ctx *c, int n); /* this many work lists will hold device memory at once */
extern "C" spx_alloc_gate *spx_internal_gate_create(void);
extern "C" void spx_internal_gate_free(spx_alloc_gate *g);
extern "C" void spx_internal_gate_skip(spx_alloc_gate *g, int64_t ticket);
extern "C" void spx_internal_work_gate(spx_work *w, spx_alloc_gate *g, int64_t ticket);
/* spx_stage in its two halves: host only (sizes, dispatch filter, repeated SEQ / QUAL) | device memory + copies */
extern "C" int spx_internal_stage_begin(spx_ctx *c, const spx_batch *const *bts, int32_t n_batches, const spx_params *par, int host_threads,
                                        spx_work **out);
extern "C" int spx_internal_stage_finish(spx_ctx *c, spx_work *w);

namespace {

struct Job {
    std::vector<const spx_batch *> batches;
    spx_work *work = nullptr; /* staged by the caller (records resident in HBM) or created here */
    bool own_work = false;
    void *tag = nullptr;
    int32_t n_groups = 0;
    int64_t ticket = 0; /* submission number: device memory is taken in this order */
    std::vector<spx_group_out> out;
    int rc = 0;
    std::string err;
    bool done = false;
};

} // namespace

struct spx_pipe {
    spx_ctx *ctx = nullptr;
    spx_params par;
    int depth = 2, stage_threads = 1;
    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_room;
    std::deque<Job *> todo;     /* submitted, not yet picked up */
    std::deque<Job *> inflight; /* submission order, delivered from the front */
    bool stop = false;
    std::vector<std::thread> workers;
    /* Device memory is handed out in SUBMISSION order: a job stages (takes the HBM of its records) only after every
     * older job has staged, and its work list takes its memory only after every older job holds its list (the allocation
     * gate inside spx_prepare_staged).  Results are delivered in order and a list is freed by the caller after delivery,
     * so the oldest job in flight can always finish; a younger job that took memory first could leave it waiting for
     * ever (the lists of 16 384 ONT groups take ~70 GB each).  The device PREPARATIONS of several jobs run side by side
     * (preparation lanes of the context: chains of dependent loads that leave the chip idle); the DP launches go out in
     * submission order again. */
    int64_t next_ticket = 0, stage_turn = 0, launch_turn = 0;
    std::condition_variable cv_turn;
    spx_alloc_gate *gate = nullptr;
};

static double pipe_now()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}
static void run_job(spx_pipe *p, Job *j)
{
    int rc = SPX_OK;
    static const bool timing = getenv("SPX_TIMING") != nullptr;
    const double t_a = pipe_now();
    double t_b = 0, t_c = 0, t_d = 0, t_e = 0;
    const bool stage_here = !j->work;
    if (stage_here) { /* the host half runs beside the copies of the submission in front */
        rc = spx_internal_stage_begin(p->ctx, j->batches.data(), (int32_t)j->batches.size(), &p->par, p->stage_threads, &j->work);
        j->own_work = rc == SPX_OK;
    }
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv_turn.wait(lk, [&] { return p->stage_turn == j->ticket; });
    }
    if (stage_here && rc == SPX_OK) rc = spx_internal_stage_finish(p->ctx, j->work);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        ++p->stage_turn;
    }
    p->cv_turn.notify_all();
    t_b = pipe_now();
    if (rc == SPX_OK) {
        spx_internal_work_gate(j->work, p->gate, j->ticket);
        rc = spx_prepare_staged(p->ctx, j->work);
        spx_internal_work_gate(j->work, nullptr, 0);
    }
    t_c = pipe_now();
    spx_internal_gate_skip(p->gate, j->ticket); /* (no-op when the preparation went through the gate) */
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv_turn.wait(lk, [&] { return p->launch_turn == j->ticket; });
    }
    t_d = pipe_now();
    if (rc == SPX_OK) rc = spx_launch(p->ctx, j->work);
    t_e = pipe_now();
    {
        std::lock_guard<std::mutex> lk(p->mu);
        ++p->launch_turn;
    }
    p->cv_turn.notify_all();
    if (rc == SPX_OK) {
        j->out.resize((size_t)(j->n_groups > 0 ? j->n_groups : 1));
        rc = spx_collect(p->ctx, j->work, j->out.data());
    }
    if (timing)
        fprintf(stderr, "[spx timing] pipe job %lld: start %.3f | staged +%.3f | prepared (host) +%.3f | launch turn +%.3f | launched +%.3f | collected +%.3f s\n", (long long)j->ticket,
                t_a, t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d, pipe_now() - t_e);
    if (rc != SPX_OK) j->err = spx_last_error();
    j->rc = rc;
}

static void worker_main(spx_pipe *p)
{
    for (;;) {
        Job *j = nullptr;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_work.wait(lk, [&] { return p->stop || !p->todo.empty(); });
            if (p->todo.empty()) return; /* stop requested and nothing left */
            j = p->todo.front();
            p->todo.pop_front();
        }
        run_job(p, j);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            j->done = true;
        }
        p->cv_done.notify_all();
    }
}

extern "C" int spx_pipe_create(spx_ctx *ctx, const spx_params *par, int depth, int host_threads, spx_pipe **out)
{
    if (!ctx || !par || !out) return SPX_EINVAL;
    spx_pipe *p = new spx_pipe();
    p->ctx = ctx;
    p->par = *par;
    p->depth = depth < 1 ? 1 : (depth > 16 ? 16 : depth);
    int ht = host_threads > 0 ? host_threads : spx::effective_cpus();
    p->stage_threads = ht > 0 ? ht : 1; /* stagings run one after the other (ticket order): each gets all the threads */
    p->gate = spx_internal_gate_create();
    spx_internal_lists_in_flight(ctx, p->depth + 1);
    for (int t = 0; t < p->depth; ++t) p->workers.emplace_back(worker_main, p);
    *out = p;
    return SPX_OK;
}

extern "C" int spx_pipe_submit(spx_pipe *p, const spx_batch *const *batches, int32_t n_batches, spx_work *staged, int32_t n_groups_staged,
                               void *tag)
{
    if (!p || (!staged && (!batches || n_batches <= 0))) return SPX_EINVAL;
    Job *j = new Job();
    j->tag = tag;
    if (staged) {
        /* the result buffer is sized from the list itself; a list may be in flight once */
        spx_stats st;
        if (spx_work_stats(staged, &st) != SPX_OK || st.n_groups != (int64_t)n_groups_staged) {
            delete j;
            spx_internal_set_error("n_groups_staged does not match the staged work list");
            return SPX_EINVAL;
        }
        if (spx_internal_work_claim(staged, 1) != SPX_OK) {
            delete j;
            spx_internal_set_error("work list is already in flight");
            return SPX_EINVAL;
        }
        j->work = staged;
        j->n_groups = n_groups_staged;
    } else {
        int64_t n = 0;
        for (int32_t b = 0; b < n_batches; ++b) {
            if (!batches[b]) { delete j; return SPX_EINVAL; }
            j->batches.push_back(batches[b]);
            n += batches[b]->n_groups;
        }
        j->n_groups = (int32_t)n;
    }
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv_room.wait(lk, [&] { return (int)p->inflight.size() < p->depth + 1; }); /* one waiting beside `depth` running */
        j->ticket = p->next_ticket++;
        p->todo.push_back(j);
        p->inflight.push_back(j);
    }
    p->cv_work.notify_one();
    return SPX_OK;
}

extern "C" int spx_pipe_pending(spx_pipe *p)
{
    if (!p) return 0;
    std::lock_guard<std::mutex> lk(p->mu);
    return (int)p->inflight.size();
}

extern "C" int spx_pipe_next(spx_pipe *p, spx_group_out *out, int32_t capacity, spx_work **work, void **tag)
{
    if (!p || !out) return SPX_EINVAL;
    Job *j = nullptr;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        if (p->inflight.empty()) return SPX_EINVAL;
        j = p->inflight.front();
        p->cv_done.wait(lk, [&] { return j->done; });
        p->inflight.pop_front();
    }
    p->cv_room.notify_all();
    int rc = j->rc;
    if (rc != SPX_OK) spx_internal_set_error(j->err.c_str());
    if (tag) *tag = j->tag;
    if (rc == SPX_OK && capacity < j->n_groups) rc = SPX_EINVAL;
    if (rc == SPX_OK) {
        for (int32_t g = 0; g < j->n_groups; ++g) out[g] = j->out[(size_t)g];
        rc = j->n_groups;
    }
    if (!j->own_work && j->work) spx_internal_work_claim(j->work, 0);
    if (work) *work = j->work; /* the caller frees it (or keeps its own staged list) */
    else if (j->own_work && j->work) spx_work_free(p->ctx, j->work);
    delete j;
    return rc;
}

extern "C" void spx_pipe_destroy(spx_pipe *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_work.notify_all();
    for (auto &t : p->workers) t.join();
    for (Job *j : p->inflight) {
        if (!j->own_work && j->work) spx_internal_work_claim(j->work, 0);
        if (j->own_work && j->work) spx_work_free(p->ctx, j->work);
        delete j;
    }
    spx_internal_gate_free(p->gate);
    delete p;
}
