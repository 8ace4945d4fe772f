/*
 * spx_pool.h -- persistent worker pool shared by the BAM reader (inflate, field / tag passes) and the staging of record
 * batches into pinned memory.  Internal.
 */
#ifndef SPX_POOL_H
#define SPX_POOL_H

#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace spx {

/* CPUs this process may really use: the online count, cut by the affinity mask and by the container's CPU-time quota
 * (cgroup v2 cpu.max / v1 cpu.cfs_quota_us).  The MI355X boxes show 256 CPUs to a container that may use 16: threads
 * beyond the quota only get the whole group throttled (staging into pinned memory: 72 GB/s on 64 threads, 118 on 16). */
inline int effective_cpus()
{
    static const int n = [] {
        int hw = (int)std::thread::hardware_concurrency();
        if (hw < 1) hw = 1;
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) {
            const int k = CPU_COUNT(&set);
            if (k > 0 && k < hw) hw = k;
        }
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else {
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
        }
        if (quota > 0 && period > 0) {
            const int k = (int)((quota + period - 1) / period);
            if (k >= 1 && k < hw) hw = k;
        }
        return hw;
    }();
    return n;
}

class Pool {
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    bool stop_ = false;

public:
    explicit Pool(int n)
    {
        for (int t = 0; t < std::max(1, n); ++t)
            th_.emplace_back([this] {
                for (;;) {
                    std::function<void()> f;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                        if (q_.empty()) return;
                        f = std::move(q_.front());
                        q_.pop_front();
                    }
                    f();
                }
            });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return (int)th_.size(); }
    void submit(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            q_.push_back(std::move(f));
        }
        cv_.notify_one();
    }
    /* f(k0, k1) over [0, n) in pieces of `grain`; the caller works too and returns when every piece is done */
    template <class F>
    void parallel_for(int64_t n, int64_t grain, F f)
    {
        if (n <= 0) return;
        grain = std::max<int64_t>(1, grain);
        const int64_t pieces = (n + grain - 1) / grain;
        if (pieces <= 1) { f((int64_t)0, n); return; }
        struct St {
            std::atomic<int64_t> next{0}, done{0};
            std::mutex mu;
            std::condition_variable cv;
        };
        auto st = std::make_shared<St>();
        F *fp = &f; /* late helpers find no piece left and never touch it */
        auto run = [st, fp, n, grain, pieces] {
            for (;;) {
                const int64_t k = st->next.fetch_add(1);
                if (k >= pieces) return;
                (*fp)(k * grain, std::min(n, (k + 1) * grain));
                if (st->done.fetch_add(1) + 1 == pieces) {
                    std::lock_guard<std::mutex> lk(st->mu);
                    st->cv.notify_all();
                }
            }
        };
        const int helpers = (int)std::min<int64_t>(size(), pieces - 1);
        for (int t = 0; t < helpers; ++t) submit(run);
        run();
        std::unique_lock<std::mutex> lk(st->mu);
        st->cv.wait(lk, [&] { return st->done.load() == pieces; });
    }
};


} // namespace spx
#endif
