/*
 * spx_cpuacc.h -- diagnostics (SPX_TIMING): CPU time of the host side by category, summed over all threads.  On the MI355X
 * boxes the container's CPU quota (16 cores) is what bounds the path from a BAM file, so wall-clock phase times say little:
 * what counts is how many core-seconds every kind of work takes.  Internal.
 */
#ifndef SPX_CPUACC_H
#define SPX_CPUACC_H
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

#include <atomic>

namespace spx {
enum { CPU_INFLATE = 0, CPU_CRC, CPU_DEV_CHUNK, CPU_POPULATE, CPU_WALK, CPU_PARSE, CPU_MEASURE, CPU_FILL, CPU_BED, CPU_N };
inline std::atomic<int64_t> *cpu_acc()
{
    static std::atomic<int64_t> a[CPU_N];
    return a;
}
inline bool cpu_acc_on()
{
    static const bool on = getenv("SPX_TIMING") != nullptr;
    return on;
}
inline int64_t thread_cpu_ns()
{
    struct timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec;
}
struct CpuScope {
    int k;
    int64_t t0;
    explicit CpuScope(int kind) : k(kind), t0(cpu_acc_on() ? thread_cpu_ns() : 0) {}
    ~CpuScope() { if (cpu_acc_on()) cpu_acc()[k].fetch_add(thread_cpu_ns() - t0, std::memory_order_relaxed); }
};
} // namespace spx
#endif
