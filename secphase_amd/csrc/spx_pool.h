/*
 * spx_pool.h -- persistent worker pool shared by the BAM reader (inflate, field / tag passes) and the staging of record
 * batches into pinned memory.  Internal.
 */
#ifndef SPX_POOL_H
#define SPX_POOL_H

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
